#!/usr/bin/env python3
"""hipcc with the accumulation-register budget of the device kernels stated instead of guessed.

The tiled forces kernel keeps its list ring in a0..a15 by inline asm (forces.hip, "AccRing").  LLVM's register budget for
gfx90a+ is one file of VGPRs and AGPRs; a kernel whose IR does not say how many AGPRs it needs gets HALF of the budget as
AGPRs as soon as anything names one (SIRegisterInfo::getMaxNumVectorRegs: "amdgpu-agpr-alloc" absent -> MaxVectorRegs / 2).
With two waves per SIMD that is 128 + 128, the kernel's ~220 live values do not fit 128 and ~90 of them live in AGPR spill
slots behind v_accvgpr_read / _write.  The function attribute "amdgpu-agpr-alloc"="N" is the documented way to state the
need, but clang has no source spelling for it and this compiler's attributor only infers "0" (AAAMDGPUNoAGPR).  So the
attribute is written into the device bitcode between the two stages of the device compilation:

    hipcc -### -save-temps ...   ->  the driver's own command list, run one by one here;
    after the "-emit-llvm-bc" device stage: llvm-dis | add the attribute to every amdgpu_kernel | llvm-link (as assembler).

usage: hipcc_agpr_alloc.py N <hipcc> <hipcc arguments ... -c file.hip -o out.o>     (run in a scratch directory)
"""
import os
import re
import shlex
import subprocess
import sys


def main():
    n = int(sys.argv[1])
    hipcc = sys.argv[2]
    args = sys.argv[3:]
    out = subprocess.run([hipcc, "-###", "-save-temps"] + args, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    if out.returncode != 0:
        sys.stderr.write(out.stderr)
        sys.exit(out.returncode)
    cmds = [shlex.split(l.strip()) for l in out.stderr.splitlines() if l.startswith(' "')]
    if not cmds:
        sys.exit("hipcc_agpr_alloc: no commands in the driver's -### output")
    llvm_bin = os.path.dirname(cmds[0][0])
    patched = 0
    for cmd in cmds:
        subprocess.check_call(cmd)
        if "-emit-llvm-bc" in cmd and "amdgcn-amd-amdhsa" in cmd[cmd.index("-triple") + 1]:
            bc = cmd[cmd.index("-o") + 1]
            ll = subprocess.check_output([os.path.join(llvm_bin, "llvm-dis"), bc, "-o", "-"], text=True)
            # attribute groups used by amdgpu_kernel definitions
            groups = set(re.findall(r"^define [^\n]*amdgpu_kernel[^\n]*?#(\d+)", ll, flags=re.M))
            if not groups:
                sys.exit("hipcc_agpr_alloc: no amdgpu_kernel in " + bc)

            def add(m):
                if m.group(1) not in groups:
                    return m.group(0)
                body = re.sub(r'"amdgpu-agpr-alloc"="[^"]*" ?', "", m.group(2))
                return 'attributes #%s = { "amdgpu-agpr-alloc"="%d" %s}' % (m.group(1), n, body)
            ll, k = re.subn(r"^attributes #(\d+) = \{ ([^\n]*)\}", add, ll, flags=re.M)
            # (the image has no llvm-as; llvm-link of one text module is the same thing)
            with open(bc + ".ll", "w") as f:
                f.write(ll)
            subprocess.check_call([os.path.join(llvm_bin, "llvm-link"), bc + ".ll", "-o", bc])
            patched += len(groups)
    if not patched:
        sys.exit("hipcc_agpr_alloc: the device bitcode stage was not found in the driver's command list")


if __name__ == "__main__":
    main()
