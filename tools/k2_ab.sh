#!/bin/bash
# A/B builds of the K2 kernels ON THE BOX (measurement only; the tree's .so is overwritten there, not here):
#   tools/k2_ab.sh "-DK2_WAVES_PER_EU=6" "<other -D flags>" ...   -> ms per step, K1 / K2 / K3 kernel ms per variant, base first
cd "$GRAFT_REPO_ROOT"
BASEFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-c++20-extensions"
line() { timeout 240 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']
print('$1', 'step', round(d['ms_per_step'],2), 'k1', k.get('k1_loglikes'), 'k2', k.get('k2_viterbi_dp'), 'k3', k.get('k3_accumulate'))"; }
build() { touch kaldi_hmm_gmm_amd/csrc/khg_k2.hip; make -C kaldi_hmm_gmm_amd/csrc CXXFLAGS="$BASEFLAGS $1" 2>&1 | grep -i "error" ; }
line base; line base
for v in "$@"; do build "$v"; line "[$v]"; line "[$v]"; done
