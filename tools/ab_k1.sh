#!/bin/bash
# A/B of K1 builds on ONE box: tools/bin/libkhg_<name>.so variants are copied over the library in turn, each timed by a short bench
# run (K1 kernel ms by HIP events); "base" = the library as built.  usage: tools/ab_k1.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT"
cp kaldi_hmm_gmm_amd/libkhg_hip.so /tmp/libkhg_base.so
for round in 1 2; do
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/libkhg_base.so kaldi_hmm_gmm_amd/libkhg_hip.so; else cp tools/bin/libkhg_$v.so kaldi_hmm_gmm_amd/libkhg_hip.so; fi
  python bench.py --steps 3 --warmup 1 --no-fp32-line --no-cpu-baseline --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$v', 'k1 %.2f  step %.2f  ll %.6f' % (k['k1_loglikes'], d['ms_per_step'], d['check']['avg_loglike_per_frame']))"
done; done
cp /tmp/libkhg_base.so kaldi_hmm_gmm_amd/libkhg_hip.so
