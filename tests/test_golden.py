"""tests/golden/em_small.npz: (CPU) the oracle still reproduces the committed vectors;
(GPU) the HIP path matches them: alignments / words / status bit-exact on K1's own scores for this
fixture, statistics and M-step output within the stated fp32 tolerance."""
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "em_small.npz"))


def _graphs():
    return {k[2:]: G[k] for k in G.files if k.startswith("g_")}


def test_oracle_reproduces_golden_alignment():
    om = orc.OModel(G["gauss_off"], G["gconsts"], G["means_invvars"], G["inv_vars"])
    g = _graphs()
    w = np.where(g["ilabel"] >= 1, g["weight"] + G["trans_cost"][g["ilabel"]], g["weight"]).astype(np.float32)
    g = dict(g, weight=w)
    fo = G["frame_off"]
    for tag, beam, retry in (("wide", 200.0, 0.0), ("narrow", 2.0, 6.0)):
        for u in range(len(fo) - 1):
            r = orc.align_utterance(orc.OGraph.from_set(g, u), om, G["id2pdf"], G["feats"][fo[u]: fo[u + 1]], acoustic_scale=0.1,
                                    beam=beam, retry_beam=retry)
            assert r["status"] == G[f"status_{tag}"][u]
            if (r["status"] & 1) == 0:
                assert (r["ali"] == G[f"ali_{tag}"][fo[u]: fo[u + 1]]).all()
                assert r["like"] == G[f"like_{tag}"][u]


def test_golden_has_interesting_cases():
    assert (G["status_wide"] == 0).all()
    assert G["words_wide"].size > 0            # epsilon-input arcs with word labels are on the best paths
    assert G["trans_acc"].sum() == G["frame_off"][-1]
    assert (np.diff(G["new_gauss_off"]) <= np.diff(G["gauss_off"])).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,beam,retry", [("wide", 200.0, 0.0), ("narrow", 2.0, 6.0)])
def test_gpu_alignment_matches_golden(ctx, tag, beam, retry):
    from kaldi_hmm_gmm_amd import DeviceModel, DeviceTransitions, UtteranceSet

    dm = DeviceModel(ctx, G["gauss_off"], G["gconsts"], G["means_invvars"], G["inv_vars"])
    tm = DeviceTransitions(ctx, G["id2pdf"])
    tm.set_trans_cost(G["trans_cost"])
    us = UtteranceSet(ctx, tm, G["frame_off"], G["feats"], graphs=_graphs())
    us.loglikes(dm)
    res = us.align(tm, beam=beam, retry_beam=retry, acoustic_scale=0.1)
    assert ((res["status"] & 3) == G[f"status_{tag}"]).all()
    assert (res["ali"] == G[f"ali_{tag}"]).all()
    assert (res["words"] == G[f"words_{tag}"]).all() and (res["words_off"] == G[f"words_off_{tag}"]).all()
    np.testing.assert_allclose(res["like"], G[f"like_{tag}"], rtol=2e-5)


@pytest.mark.gpu
def test_gpu_stats_and_m_step_match_golden(ctx):
    import kaldi_hmm_gmm_amd as khg
    from kaldi_hmm_gmm_amd import DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet

    dm = DeviceModel(ctx, G["gauss_off"], G["gconsts"], G["means_invvars"], G["inv_vars"])
    tm = DeviceTransitions(ctx, G["id2pdf"])
    us = UtteranceSet(ctx, None, G["frame_off"], G["feats"])
    us.upload_ali(G["ali_wide"])
    accs = DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs)
    st = accs.download()
    assert (st["trans_acc"] == G["trans_acc"]).all() and st["total_frames"] == G["total_frames"]
    assert st["total_log_like"] == pytest.approx(float(G["total_log_like"]), rel=2e-6)
    np.testing.assert_allclose(st["occ"], G["occ"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(st["mean_acc"], G["mean_acc"], rtol=2e-5, atol=2e-6 * np.abs(G["mean_acc"]).max())
    np.testing.assert_allclose(st["var_acc"], G["var_acc"], rtol=2e-5, atol=2e-6 * np.abs(G["var_acc"]).max())
    # host M-step (C++) on the GPU statistics vs the oracle M-step on the oracle statistics
    from kaldi_hmm_gmm_amd.mle import _flat_update
    r = _flat_update(khg.MleDiagGmmOptions(min_gaussian_occupancy=3.0), G["gauss_off"], st["occ"], st["mean_acc"], st["var_acc"], 0xF,
                     0x7, G["weights"], G["means_invvars"], G["inv_vars"])
    assert (r[0] == G["new_gauss_off"]).all()
    np.testing.assert_allclose(r[1], G["new_weights"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(r[3], G["new_means_invvars"], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(r[4], G["new_inv_vars"], rtol=2e-3)
    assert r[5] == pytest.approx(float(G["objf_change"]), rel=1e-3)
    # the same M-step on the device (K4), from the accumulators where K3 left them
    dm.set_weights(G["weights"])
    rd = dm.mle_update(accs, khg.MleDiagGmmOptions(min_gaussian_occupancy=3.0), 0x7)
    d = dm.download()
    assert (d["gauss_off"] == G["new_gauss_off"]).all()
    assert np.array_equal(d["weights"], r[1]) and np.array_equal(d["inv_vars"], r[4]) and np.array_equal(d["means_invvars"], r[3])
    np.testing.assert_allclose(d["weights"], G["new_weights"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(d["means_invvars"], G["new_means_invvars"], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(d["inv_vars"], G["new_inv_vars"], rtol=2e-3)
    assert rd["objf_change"] == pytest.approx(float(G["objf_change"]), rel=1e-3)
