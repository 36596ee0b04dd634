#!/usr/bin/env python3
"""ISA check of the hand-managed list ring of forces_tile_kernel (gpusph_amd/csrc/forces.hip, "AccRing").

The list batches in flight live in the accumulation registers a0..a15, written and read by asm statements only: the compiler
never sees a value that has not landed.  That holds as long as the compiler itself leaves those registers alone; every ring
statement lists them as clobbered, so it cannot keep a value in them across one, but nothing in the language forbids it to use
them in between.  This script verifies that it does not, on the compiler's output, for every instantiation that uses the ring:

  * no instruction outside the ring's own statements (marked ACCRING) names one of a0..a15 while the ring or the requests
    made one tile ahead are alive (the compiler does use the accumulation registers as a cheap spill space: above a15 at any
    time, a0..a15 in the stretch between two pair phases, when the ring's contents are parked in ordinary registers);
  * the kernel descriptor allocates a0..a15.

  python scripts/check_ring_isa.py [file.s ...]      (default: compiles the five forces parts to assembly; ~4 min)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpusph_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
         "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only", "-S"]
PARTS = [["-DSPHX_FORCES_PART=%d" % k] for k in (1, 2, 3, 4)] + [["-DSPHX_FORCES_PART=3", "-DSPHX_FORCES_SA"]]
ALLOWED_READERS = ("v_and_b32", "v_lshrrev_b32", "v_bfe_u32", "v_and_b32_sdwa", "v_lshrrev_b32_sdwa", "v_add_u32", "v_lshl_add_u32",
                   "v_and_or_b32", "v_add_lshl_u32", "v_mad_u32_u24", "v_mul_u32_u24", "v_lshlrev_b32")


def kernels(text):
    """name -> (lines, metadata dict)"""
    out = {}
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w*forces_tile_kernel\w*):\s*(;.*)?$", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if ".end_amdhsa_kernel" in line:
                out[name] = cur
                cur = None
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:\s+\.\w+:.*\n)+)", text):
        d = dict(re.findall(r"\.(\w+):\s+(\S+)", m.group(2)))
        meta[m.group(1)] = d
    # metadata blocks list the fields in alphabetical order around .name: collect per kernel by a wider match
    for blk in re.split(r"\n\s+- \.", text):
        n = re.search(r"\.name:\s+(\S+)", blk)
        if n and "forces_tile_kernel" in n.group(1):
            meta[n.group(1)] = dict(re.findall(r"\.(\w+):\s+(\S+)", blk))
    return out, meta


def regs_of(operand):
    m = re.match(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", operand)
    return [int(m.group(1))] if m else []


def check(text, label):
    ks, meta = kernels(text)
    bad, checked = [], 0
    accre = re.compile(r"(?<![\w.])a(\d+)\b|(?<![\w.])a\[(\d+):(\d+)\]")

    def ring_regs(code):
        """accumulation registers 0..15 named by an instruction"""
        hit = set()
        for m in accre.finditer(code):
            if m.group(1) is not None:
                lo = hi = int(m.group(1))
            else:
                lo, hi = int(m.group(2)), int(m.group(3))
            hit.update(r for r in range(lo, hi + 1) if r < 16)
        return hit

    for name, lines in ks.items():
        if not any("ACCRING" in l for l in lines):
            continue
        checked += 1
        md = meta.get(name, {})
        m = re.search(r"\.amdhsa_accum_offset\s+(\d+)", "\n".join(lines))
        nfv = re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", "\n".join(lines))
        if not m or not nfv or int(nfv.group(1)) - int(m.group(1)) < 16:
            bad.append("%s: the kernel descriptor does not cover a0..a15 (next_free_vgpr %s, accum_offset %s)"
                       % (name, nfv and nfv.group(1), m and m.group(1)))
        # where the ring registers hold nothing: before the first and after the last ring statement of the kernel, and from the
        # statement that fetches a8..a15 into ordinary registers at the top of a tile (it ends with the read of a15) to the next
        # ring statement (the requests for the next tile / the start of this tile's ring).  Text order stands for program order
        # here; a block the compiler placed elsewhere makes the check stricter, not weaker
        marks = [i for i, l in enumerate(lines) if "ACCRING" in l]
        dead = [(0, marks[0]), (marks[-1], len(lines))]
        for k, i in enumerate(marks):
            if "v_accvgpr_read_b32" in lines[i] and re.search(r"\ba15\b", lines[i]) and k + 1 < len(marks):
                dead.append((i, marks[k + 1]))
        for i, l in enumerate(lines):
            code = l.split(";")[0]
            if "ACCRING" in l or not code.strip() or code.strip().startswith("."):
                continue
            if ring_regs(code) and not any(lo < i < hi for lo, hi in dead):
                bad.append("%s: line %d touches a register of the ring while it is in use: %s" % (name, i, code.strip()))
    return checked, bad


def main():
    files = sys.argv[1:]
    texts = []
    if files:
        for f in files:
            texts.append((f, open(f).read()))
    else:
        tmp = tempfile.mkdtemp(prefix="ringisa_")
        procs = []
        for k, extra in enumerate(PARTS):
            out = os.path.join(tmp, "part%d.s" % k)
            procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc"] + FLAGS + extra + [os.path.join(CSRC, "forces.hip"), "-o", out],
                                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
        for out, pr in procs:
            if pr.wait() != 0:
                print("compile failed for", out)
                return 2
            texts.append((out, open(out).read()))
    total, allbad = 0, []
    for label, t in texts:
        n, bad = check(t, label)
        total += n
        allbad += ["%s: %s" % (os.path.basename(label), b) for b in bad]
    print("%d instantiations with a hand-managed ring checked, %d findings" % (total, len(allbad)))
    for b in allbad:
        print("  " + b)
    return 1 if allbad or not total else 0


if __name__ == "__main__":
    sys.exit(main())
