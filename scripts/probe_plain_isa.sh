#!/bin/bash
# the ISA of the plain forces pass alone (forces_tile_kernel<WENDLAND, ARTIFICIAL, COLAGROSSI, false>), in /tmp/probe_plain: seconds instead
# of the minutes a whole part takes
cd "$(dirname "$0")/../gpusph_amd/csrc"
rm -rf /tmp/probe_plain && mkdir -p /tmp/probe_plain && cd /tmp/probe_plain
/opt/rocm/bin/hipcc -save-temps --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/root/repo/include -I/root/repo/gpusph_amd/csrc -Wall -Wno-unused-function \
  -ffp-contract=off -fno-slp-vectorize -DSPHX_FORCES_PART=3 -DSPHX_PROBE_PLAIN $EXTRA -c /root/repo/gpusph_amd/csrc/forces.hip -o probe.o
ls -la /tmp/probe_plain/*.s
