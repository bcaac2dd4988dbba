#!/bin/bash
# `make -C hitl_slam_amd/csrc hostcheck` ends here: the CPU tests that drive GPU-free entry points of the C ABI, the hostcheck
# driver (one rank) and the multi-rank sessions (2-3 processes joined by the all-gather callback), against the AddressSanitizer +
# UBSan build of the library's host code (stub device layer underneath).  CPU only.
# usage: hostcheck_run.sh /abs/path/libhitlgpu_hostcheck.so
set -e
LIB=${1:?path of libhitlgpu_hostcheck.so}
cd "$(dirname "$0")/.."
export HITL_LIBHITLGPU=$LIB
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1      # (the interpreter itself is not leak-clean; leaks of the library show in the driver's own count below)
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
export OMP_NUM_THREADS=2
python3 tests/hostcheck_driver.py
python3 tests/hostcheck_ranks.py
python3 -m pytest -q -x -p no:cacheprovider tests/test_io_formats.py tests/test_abi.py -k "not no_silent_cpu_fallback" 2>&1 | tail -3
python3 -m pytest -q -x -p no:cacheprovider tests/test_cycle_oracle.py -k "order" 2>&1 | tail -3
echo "hostcheck: clean"
