"""The translation loop of tile_lists_kernel (gpusph_amd/csrc/forces.hip) keeps its list loads in flight.

Round 4 found that the compiler had put `s_waitcnt vmcnt(0)` in front of every one of the sixteen two-byte loads of that loop in
some builds and not in others, depending on edits elsewhere in the kernel (7.4 against 11.3 ms per launch at 32 M particles from the
same source).  The loads are unconditional now; this test holds the compiler's output to it: in the loop body the loads are issued
back to back, and the waits that follow count down instead of draining the queue before each load.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpusph_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_tile_list_builder_keeps_its_loads_in_flight():
    tmp = tempfile.mkdtemp(prefix="tl_isa_")
    try:
        out = os.path.join(tmp, "forces.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
                        "-ffp-contract=off", "-fno-slp-vectorize", "--cuda-device-only", "-S", os.path.join(CSRC, "forces.hip"), "-o", out],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        text = open(out).read()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    m = re.search(r"^_Z17tile_lists_kernel\w*:.*?\.end_amdhsa_kernel", text, re.S | re.M)
    assert m, "tile_lists_kernel not found in the compiler's output"
    lines = [l.split(";")[0].strip() for l in m.group(0).splitlines()]
    loads = [i for i, l in enumerate(lines) if l.startswith("global_load_ushort")]
    assert len(loads) >= 16, "the translation loop should issue its sixteen list loads (found %d)" % len(loads)
    # the longest stretch of loads with no full drain of the memory queue and no branch in between
    best = run = 1
    for a, b in zip(loads, loads[1:]):
        between = lines[a + 1:b]
        broken = any(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in between) or any(l.startswith("s_cbranch") for l in between)
        run = 1 if broken else run + 1
        best = max(best, run)
    assert best >= 12, "the list loads of the translation loop are serialised again (longest run in flight: %d)" % best
