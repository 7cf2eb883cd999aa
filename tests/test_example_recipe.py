"""BASELINE.json configs[0] (the egs/yesno plumbing): the whole monophone recipe -- init, training-graph
compile, equal-align, {align, acc-stats, est} with mixing up -- runs end to end on the GPU path and
recovers every transcript."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_monophone_recipe_end_to_end():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_mono_synthetic.py"), "--utts", "24", "--iters", "8"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    likes = [float(line.split("per frame")[1].split()[0]) for line in r.stdout.splitlines() if "avg log-like per frame" in line]
    assert len(likes) == 8 and likes[-1] > likes[0] + 5.0, likes      # EM must raise the likelihood
    assert "24/24 utterances aligned to their transcript" in r.stdout


@pytest.mark.gpu
def test_monophone_recipe_resident_end_to_end():
    """The same schedule through ResidentEm (everything in HBM, device M-step)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_mono_synthetic.py"), "--utts", "24", "--iters", "8",
                        "--resident"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    likes = [float(line.split("per frame")[1].split()[0]) for line in r.stdout.splitlines() if "avg log-like per frame" in line]
    assert len(likes) == 8 and likes[-1] > likes[0] + 5.0, likes
    assert "24/24 utterances aligned to their transcript" in r.stdout
