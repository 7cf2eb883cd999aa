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


@pytest.mark.gpu
def test_decode_example_word_loop_graph_vs_oracle(ctx):
    """egs/yesno/decode.py's flow on the synthetic task: a unigram word-loop decoding graph (cycles, in-degree 7,
    out-degree 8), beam 13.  The HIP decode must equal the oracle's FasterDecoder on the same graph -- words AND
    alignment -- and recover the transcripts."""
    import types

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import decode_synthetic as dx
    from kaldi_hmm_gmm_amd import _gpu
    from oracle import oracle as orc

    _gpu.set_default_context(ctx)
    args = types.SimpleNamespace(utts=200, test_utts=30, iters=80, dim=23, seed=3)
    tm, tree, am, lexicon, test_utts = dx.train(args, log=lambda *a: None)
    errs, nref, graph, res = dx.decode(tm, tree, am, lexicon, test_utts, log=lambda *a: None)
    c = graph.to_csr()
    assert int((c["ilabel"] == 0).sum()) == 0 and np.bincount(c["nextstate"]).max() > 6      # not a training-graph shape
    go, gc, w, miv, iv = am.flat()
    om = orc.OModel(go, gc, miv, iv)
    og = orc.OGraph(c["start"], c["arc_off"], c["ilabel"], c["olabel"], c["weight"], c["nextstate"], c["final"])
    id2pdf = np.asarray(tm.transition_id_to_pdf_array(), np.int32)
    for u, r in zip(test_utts, res):
        want = orc.align_utterance(og, om, id2pdf, u[2], acoustic_scale=0.1, beam=13.0, retry_beam=0.0)
        assert want["status"] == 0 and r["ok"]
        assert r["alignment"] == want["ali"].tolist() and r["words"] == want["words"].tolist()
    # reproducible run (seeded split perturbations, ordered K3 reductions): the 80-pass schedule on 200 utterances recovers every transcript
    assert errs <= 0.05 * nref, (errs, nref)
