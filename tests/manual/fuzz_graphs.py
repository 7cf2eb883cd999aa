"""Open-ended form of tests/fuzzlib.fuzz_graphs (a seeded, time-boxed slice runs under pytest -m gpu):
python tests/manual/fuzz_graphs.py [seconds] [seed]"""
import os, sys
_R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "tests"))
import fuzzlib
from kaldi_hmm_gmm_amd import Context

r = fuzzlib.fuzz_graphs(Context(0), float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
print("graph fuzz ok:", r)
