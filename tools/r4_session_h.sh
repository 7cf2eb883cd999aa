#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "beyond_1024 or align" 2>&1 | tail -3
timeout 1000 python tools/k2_long.py 6 2>&1 | grep -v amdgpu.ids
