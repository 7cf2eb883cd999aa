"""Command-line form of tests/fuzzlib.validate_large (tests/test_gpu_fuzz.py asserts on it and writes the report):
python tests/manual/validate_large.py [utterances]"""
import json
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import fuzzlib
from kaldi_hmm_gmm_amd import Context

print(json.dumps(fuzzlib.validate_large(Context(0), int(sys.argv[1]) if len(sys.argv) > 1 else 300), indent=1))
