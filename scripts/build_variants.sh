#!/bin/bash
# Cross-compiles the experiment libraries under gpusph_amd/variants/ (git-ignored, they travel with a gpurun snapshot) from the
# patches in scripts/experiments/, for scripts/ab_forces.sh to measure against the committed build on the GPU:
#   libsphx_ringbase.so       ring_running_base.patch
#   libsphx_ringbase_sps.so   + sps_share_descriptor.patch
#   libsphx_ringbase_hiw.so   ring_running_base.patch + walk_per_simd_slot.patch
# ~12 min on this container's cores.  The tree itself is not touched (the kernel sources are keyed to the committed profile set).
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
W=$(mktemp -d)
mkdir -p $W/gpusph_amd gpusph_amd/variants
cp -r gpusph_amd/csrc $W/gpusph_amd/ && cp -r include $W/ && rm -f $W/gpusph_amd/csrc/*.o
(cd $W && patch -p0 < $ROOT/scripts/experiments/ring_running_base.patch)
make -C $W/gpusph_amd/csrc -j10 ../libsphx.so > /dev/null
cp $W/gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_ringbase.so
(cd $W && patch -p0 < $ROOT/scripts/experiments/sps_share_descriptor.patch)
make -C $W/gpusph_amd/csrc -j10 ../libsphx.so > /dev/null
cp $W/gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_ringbase_sps.so
# the third on top of the first only: a fresh copy
rm -rf $W/gpusph_amd/csrc && cp -r gpusph_amd/csrc $W/gpusph_amd/ && rm -f $W/gpusph_amd/csrc/*.o
(cd $W && patch -p0 < $ROOT/scripts/experiments/ring_running_base.patch && patch -p0 < $ROOT/scripts/experiments/walk_per_simd_slot.patch)
make -C $W/gpusph_amd/csrc -j10 ../libsphx.so > /dev/null
cp $W/gpusph_amd/libsphx.so gpusph_amd/variants/libsphx_ringbase_hiw.so
rm -rf $W
ls -la gpusph_amd/variants/
