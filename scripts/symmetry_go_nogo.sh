#!/bin/bash
# the two measurements behind the pair-symmetry decision of DESIGN.md 5.2
cd "$(dirname "$0")/.."
python scripts/tile_pair_fraction.py 2e6 2>&1 | grep -v amdgpu.ids
scripts/ubench/lds_pair_symmetry
