#!/bin/bash
# tests/test_sa_io_hostemu.py on an address- and UB-sanitised build of the emulated kernels (tests/hostemu/README): out-of-bounds
# accesses of the open-boundary kernels on exact-size buffers abort the run.
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
SPHX_HOSTEMU_ASAN=1 LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
	python -m pytest tests/test_sa_io_hostemu.py tests/test_engine_sa_io.py -x -q -p no:cacheprovider "$@"
