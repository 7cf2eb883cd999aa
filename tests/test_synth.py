"""The synthetic workload generator (kaldi_hmm_gmm_amd/synth.py): the transcript laws bench.py / the K3 tests draw from."""
import numpy as np

from kaldi_hmm_gmm_amd import synth


def _pdf_sets(ut, n):
    fo = ut.frame_off
    return [set(np.unique(ut.frame_pdf[fo[u]: fo[u + 1]]).tolist()) for u in range(n)]


def test_zipf_stream_is_reproducible_and_in_range():
    a = synth.zipf_phone_stream(np.random.default_rng(3), 50, 10000)
    b = synth.zipf_phone_stream(np.random.default_rng(3), 50, 10000)
    assert a.shape == (10000,) and np.array_equal(a, b) and a.min() >= 0 and a.max() < 50
    sil = float((a == 0).mean())
    assert 0.02 < sil < 0.08          # one silence per ~5 words of ~5 phones
    # frequent words repeat: the most common 4-gram of phones occurs far more often than in independent phones
    grams = np.lib.stride_tricks.sliding_window_view(a, 4)
    _, cnt = np.unique(grams, axis=0, return_counts=True)
    assert cnt.max() > 20


def test_transcript_laws_change_the_overlap_and_the_skew_not_the_graph_shape():
    m = synth.make_model(600, 2, 4, seed=1)
    uts = {tr: synth.make_utts(m, 300, seed=5, feats=False, transcripts=tr) for tr in ("uniform", "zipf", "skew")}
    shared = {}
    for tr, ut in uts.items():
        # same utterance lengths in phones whatever the law; chain graphs of 3 states per phone + the final state
        assert np.array_equal(ut.num_phones, uts["uniform"].num_phones)
        assert np.array_equal(np.diff(ut.graphs["state_off"]), 3 * ut.num_phones + 1)
        assert ut.frame_pdf.min() >= 0 and ut.frame_pdf.max() < 600
        sets = _pdf_sets(ut, 100)
        shared[tr] = np.mean([len(sets[i] & sets[j]) for i in range(0, 100, 2) for j in range(1, 100, 2)])
    assert shared["zipf"] > 1.3 * shared["uniform"]
    per = np.bincount(uts["skew"].frame_pdf, minlength=600)
    assert per[:3].min() > 30 * np.median(per)        # half of all phones are phone 0


def test_mismatched_model_trades_whole_pdfs_and_nothing_else():
    """The scoring model of the flat-start line and the hard-regime tests: a seeded fraction of the pdfs trade ALL their parameters
    among themselves (no pdf keeps its own, none is lost), everything else -- transition-ids, the other pdfs -- is untouched."""
    m = synth.make_model(60, 4, 5, seed=3)
    mm = synth.mismatched_model(m, 0.3, seed=7)
    G = 4
    rows = lambda a, p: a[p * G:(p + 1) * G]
    moved = [p for p in range(60) if not np.array_equal(rows(mm.means_invvars, p), rows(m.means_invvars, p))]
    assert 5 <= len(moved) <= 35                                   # ~ 0.3 of 60
    for p in range(60):
        src = [q for q in range(60) if np.array_equal(rows(mm.means_invvars, p), rows(m.means_invvars, q))]
        assert len(src) == 1
        q = src[0]
        assert (q != p) == (p in moved)
        for name in ("weights", "inv_vars", "means", "vars"):
            assert np.array_equal(rows(getattr(mm, name), p), rows(getattr(m, name), q)), name
    # a permutation of the moved set: every traded pdf's parameters are still somewhere
    srcs = sorted(next(q for q in range(60) if np.array_equal(rows(mm.means_invvars, p), rows(m.means_invvars, q))) for p in moved)
    assert srcs == sorted(moved)
    assert np.array_equal(mm.gauss_off, m.gauss_off) and np.array_equal(mm.id2pdf, m.id2pdf)
    again = synth.mismatched_model(m, 0.3, seed=7)
    assert np.array_equal(again.means_invvars, mm.means_invvars)
    assert np.array_equal(synth.mismatched_model(m, 0.0, seed=7).means_invvars, m.means_invvars)
