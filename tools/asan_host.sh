#!/bin/bash
# The CPU test suite against an AddressSanitizer + UBSan build of the pybind11 host extension (the C++ host classes); the GPU pool has
# no sanitizer runs, the host side can.  Rebuilds the normal extension afterwards.
set -e
cd "$(dirname "$0")/../kaldi_hmm_gmm_amd/csrc"
EXT=../_kaldi_hmm_gmm_amd$(python3-config --extension-suffix)
SRC="khg_pybind.cpp khg_py_host.cpp khg_py_hmm.cpp khg_py_align.cpp khg_host_gmm.cpp khg_host_hmm.cpp khg_host_align.cpp khg_host_fst.cpp"
g++ -O1 -g -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden -fsanitize=address,undefined -fno-omit-frame-pointer \
    $(python3 -m pybind11 --includes) $SRC -o $EXT -L.. -lkhg_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath-link,/opt/rocm/lib
cd ../..
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" python -m pytest tests -x -q -m "not gpu" || true
touch kaldi_hmm_gmm_amd/csrc/khg_pybind.cpp
make -C kaldi_hmm_gmm_amd/csrc
