#!/bin/bash
# Knock-out builds of the default K1 for tools/ab_k1.sh (timing only: their results are wrong): tools/bin/libkhg_ko<mask>.so with
# K1S_KO=<mask> (khg_k1_f16x2s.hip.inc).  usage: tools/build_k1_variants.sh 1 2 4 8 16 ...
set -e
cd "$(dirname "$0")/../kaldi_hmm_gmm_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/bin build
for ko in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-c++20-extensions -DK1S_KO=$ko -c -o build/khg_k1_ko$ko.o khg_k1.hip 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o ../../tools/bin/libkhg_ko$ko.so build/khg_ctx_model.o build/khg_utts.o build/khg_k1_ko$ko.o build/khg_k2.o build/khg_k3.o build/khg_c1.o build/khg_k4.o build/khg_host.o -ldl -Wl,-rpath,/opt/rocm/lib && echo "built ko$ko" ) &
done
wait
