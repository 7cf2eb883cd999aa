#!/bin/bash
# round 4, GPU session C: the > 1000-state decoder test against a build WITHOUT the bucket re-ranking (must fail: the test has teeth),
# then the long-transcript fallback timing
cd "$GRAFT_REPO_ROOT"
cp kaldi_hmm_gmm_amd/libkhg_hip.so /tmp/libkhg_base.so
cp tools/bin/libkhg_norerank.so kaldi_hmm_gmm_amd/libkhg_hip.so
echo "== no re-ranking (expected to FAIL) =="
timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -q -x -k "above_1000" 2>&1 | tail -8
cp /tmp/libkhg_base.so kaldi_hmm_gmm_amd/libkhg_hip.so
echo "== product =="


timeout 600 python -m pytest tests/test_gpu_api.py -m gpu -q -x -k "above_1000" 2>&1 | tail -8
