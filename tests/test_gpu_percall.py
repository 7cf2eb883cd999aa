"""The reference's own call pattern (egs/yesno/train.py:170-202): gmm_align_compiled + gmm_acc_stats_ali once per utterance, with
the reference's keyword arguments, against the batched device path on the same utterances -- at a model far above the recipe's
(2000 pdfs x 32 Gaussians x 40 dims: one upload of it per call would dominate everything).

What is checked: identical alignments; statistics within the tolerance of tests/test_gpu_parity.py; the per-call log-likelihoods;
the transition counts; that the device state cached on the host objects follows their mutations (a changed pdf is seen by the
next call, a copy made by gmm_boost_silence is its own model) and that every host-side reader of AccumAmDiagGmm sees the
statistics that were still on the device."""
import pickle

import numpy as np
import pytest

from kaldi_hmm_gmm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def khg(ctx):
    import kaldi_hmm_gmm_amd as k
    from kaldi_hmm_gmm_amd import _gpu
    _gpu.set_default_context(ctx)
    return k


def _batched(khg, ctx, m, am, ut, n, cost, beam, retry_beam):
    go, gc, _, miv, iv = am.flat()
    dm = khg.DeviceModel(ctx, go, gc, miv, iv)
    tm = khg.DeviceTransitions(ctx, m.id2pdf)
    tm.set_trans_cost(cost)
    fo = ut.frame_off[: n + 1]
    g = ut.graphs
    so = g["state_off"][: n + 1]
    ao = g["arc_off"][: so[-1] + 1]
    sub = {"state_off": so, "start": g["start"][:n], "arc_off": ao, "ilabel": g["ilabel"][: ao[-1]], "olabel": g["olabel"][: ao[-1]],
           "weight": g["weight"][: ao[-1]], "nextstate": g["nextstate"][: ao[-1]], "final": g["final"][: so[-1]]}
    us = khg.UtteranceSet(ctx, tm, fo, ut.feats[: fo[-1]], graphs=sub)
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=beam, retry_beam=retry_beam, acoustic_scale=0.1)
    accs = khg.DeviceAccs(ctx, dm, tm)
    us.acc_stats(dm, tm, accs)
    st = accs.download()
    accs.close(); us.close(); tm.close(); dm.close()
    return res, st


@pytest.mark.parametrize("beam,retry_beam", [(200.0, 0.0), (6.0, 40.0)])
def test_per_utterance_calls_equal_the_batched_path(khg, ctx, beam, retry_beam):
    P, G, D, n = 2000, 32, 40, 24
    m = synth.make_model(P, G, D, seed=77)
    ut = synth.make_utts(m, n, seed=9, min_phones=4, max_phones=30)
    am, tm = synth.host_objects(m)
    assert tm.num_transition_ids == 2 * 3 * (P // 3)
    np.testing.assert_array_equal(np.asarray(tm.transition_id_to_pdf_array())[1:], m.id2pdf[1: tm.num_transition_ids + 1])
    cfg = khg.AlignConfig(beam=beam, retry_beam=retry_beam, careful=False)
    gmm_accs = khg.AccumAmDiagGmm()
    gmm_accs.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    transition_accs = None
    alis, lls, likes, frames = [], [], [], 0
    for u in range(n):
        feats = np.ascontiguousarray(ut.feats[ut.frame_off[u]: ut.frame_off[u + 1]])
        ans = khg.gmm_align_compiled(am_gmm=am, transition_model=tm, utt=f"utt{u}", fst=synth.utt_fst(ut.graphs, u), feats=feats, align_config=cfg,
                                     acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
        assert ans["num_done"] == 1 and ans["frame_count"] == feats.shape[0]
        alis.append(np.asarray(ans["alignment"], np.int32)); likes.append(ans["tot_like"])
        log_like, transition_accs = khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=gmm_accs, transition_model=tm, feats=feats, ali=ans["alignment"],
                                                          transition_accs=transition_accs)
        lls.append(log_like); frames += feats.shape[0]
    assert gmm_accs._has_device_stats            # nothing has crossed PCIe but the scalars
    cost = np.asarray(tm.scaled_trans_cost(1.0, 0.1), np.float32)
    cost_full = np.zeros(m.num_tids + 1, np.float32); cost_full[: cost.shape[0]] = cost
    res, st = _batched(khg, ctx, m, am, ut, n, cost_full, beam, retry_beam)
    np.testing.assert_array_equal(np.concatenate(alis), res["ali"])
    np.testing.assert_allclose(likes, res["like"], rtol=1e-6)
    # host-side readers see the device statistics
    assert gmm_accs.tot_count == pytest.approx(frames)
    assert not gmm_accs._has_device_stats
    occ = np.concatenate([np.asarray(gmm_accs.get_acc(p).occupancy) for p in range(P)])
    mean = np.concatenate([np.asarray(gmm_accs.get_acc(p).mean_accumulator) for p in range(P)])
    var = np.concatenate([np.asarray(gmm_accs.get_acc(p).variance_accumulator) for p in range(P)])
    np.testing.assert_allclose(occ, st["occ"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(mean, st["mean_acc"], rtol=2e-5, atol=2e-6 * np.abs(st["mean_acc"]).max())
    np.testing.assert_allclose(var, st["var_acc"], rtol=2e-5, atol=2e-6 * np.abs(st["var_acc"]).max())
    assert sum(lls) == pytest.approx(st["total_log_like"], rel=1e-9)
    assert gmm_accs.tot_log_like == pytest.approx(st["total_log_like"], rel=1e-6)
    want_t = np.bincount(np.concatenate(alis), minlength=tm.num_transition_ids + 1).astype(np.float64)
    np.testing.assert_array_equal(transition_accs, want_t)
    # the update reads the flushed accumulators
    r = khg.mle_am_diag_gmm_update(config=khg.MleDiagGmmOptions(), amdiag_gmm_acc=gmm_accs, flags=khg.GmmUpdateFlags.kGmmAll, am_gmm=am)
    assert r[1] == pytest.approx(frames, rel=1e-5)


def test_cached_device_model_follows_the_host_object(khg, ctx):
    P, G, D = 60, 8, 13
    m = synth.make_model(P, G, D, seed=5)
    ut = synth.make_utts(m, 3, seed=2, min_phones=3, max_phones=6)
    am, tm = synth.host_objects(m)
    feats = np.ascontiguousarray(ut.feats[ut.frame_off[0]: ut.frame_off[1]])
    ali = [int(x) for x in ut.ref_ali[ut.frame_off[0]: ut.frame_off[1]]]

    def run(model):
        a = khg.AccumAmDiagGmm(); a.init(model=model, flags=khg.GmmUpdateFlags.kGmmAll)
        ll, _ = khg.gmm_acc_stats_ali(am_gmm=model, gmm_accs=a, transition_model=tm, feats=feats, ali=ali)
        return ll, a

    ll0, _ = run(am)
    ll0b, _ = run(am)
    assert ll0 == ll0b
    # a mutation through the reference-returning get_pdf(i) must reach the device
    p0 = int(ut.frame_pdf[ut.frame_off[0]])
    g = am.get_pdf(p0)
    g.set_means(np.asarray(g.means) + np.float32(0.5))
    g.compute_gconsts()
    ll1, _ = run(am)
    assert ll1 != ll0
    fresh = khg.AmDiagGmm(); fresh.copy_from_am_diag_gmm(am)
    ll1f, _ = run(fresh)
    assert ll1f == ll1
    # gmm_boost_silence returns a COPY with other weights: its own device model, the original's stays what it was
    boosted = khg.gmm_boost_silence(am_gmm=am, transition_model=tm, silence_phones=[p0 // 3 + 1], boost=3.0)
    ll2, _ = run(boosted)
    assert ll2 != ll1
    ll1c, _ = run(am)
    assert ll1c == ll1
    # pickling the accumulators reads the device statistics; an AccumAmDiagGmm copy is a deep copy of them
    _, a = run(am)
    assert a._has_device_stats
    b = khg.AccumAmDiagGmm(); b.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    b.add(1.0, a)
    assert b.tot_count == pytest.approx(len(ali)) and a.tot_count == pytest.approx(len(ali))
    am2 = pickle.loads(pickle.dumps(am))
    ll3, _ = run(am2)
    assert ll3 == pytest.approx(ll1, rel=1e-6)


def test_align_after_the_band_model_changed_or_died_is_an_error(khg, ctx):
    """khg_loglikes_band keeps pointers into the model for khg_align's repair launch: a model destroyed or updated in between must
    give an error, not a use-after-free; a model whose fp16 image was merely re-packed (another set lowered the shared feature
    exponents) makes khg_align score the set again, with identical alignments."""
    P, G, D = 90, 24, 40
    m = synth.make_model(P, G, D, seed=3)
    ut = synth.make_utts(m, 12, seed=4, min_phones=3, max_phones=8)
    import kaldi_hmm_gmm_amd as k
    gc = np.zeros(m.weights.shape[0], np.float32)
    from kaldi_hmm_gmm_amd import _lib
    import ctypes as C
    _lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(m.gauss_off, C.c_int32), _lib.ptr(m.weights, C.c_float), _lib.ptr(m.inv_vars, C.c_float),
                                            _lib.ptr(m.means_invvars, C.c_float), _lib.ptr(gc, C.c_float), None))
    tm = k.DeviceTransitions(ctx, m.id2pdf)

    def new_set(scale=1.0):
        return k.UtteranceSet(ctx, tm, ut.frame_off, (ut.feats * np.float32(scale)).astype(np.float32), graphs=ut.graphs)

    dm = k.DeviceModel(ctx, m.gauss_off, gc, m.means_invvars, m.inv_vars)
    us = new_set()
    us.loglikes(dm, reachable_only=True, band=True)
    want = us.align(tm, beam=200.0, acoustic_scale=0.1)
    # (1) the image re-packed for another set's (larger) features: align must still give the same answer, at a beam that sends
    # every utterance through the repair
    us.loglikes(dm, reachable_only=True, band=True)
    big = new_set(scale=3.0)
    big.loglikes(dm, reachable_only=True, band=True)
    got = us.align(tm, beam=1e-3, retry_beam=200.0, acoustic_scale=0.1)
    np.testing.assert_array_equal(got["ali"], want["ali"])
    big.close()
    # (2) updated in place
    us.loglikes(dm, reachable_only=True, band=True)
    dm.set_weights(m.weights)
    dm.scale_weights(np.asarray([0, 1], np.int32), 1.5)
    with pytest.raises(k.KhgError, match="destroyed or updated"):
        us.align(tm, beam=200.0, acoustic_scale=0.1)
    # (3) destroyed
    us.loglikes(dm, reachable_only=True, band=True)
    dm.close()
    with pytest.raises(k.KhgError, match="destroyed or updated"):
        us.align(tm, beam=200.0, acoustic_scale=0.1)
    us.close(); tm.close()


@pytest.mark.parametrize("P,G,D,ragged", [(30, 8, 13, False), (45, 20, 39, True), (60, 70, 40, True), (24, 40, 80, False), (21, 130, 23, True)])
def test_per_utterance_calls_across_model_shapes(khg, ctx, P, G, D, ragged):
    """Every K1 / K3 form the shapes select (packed small pdfs, the wave forms, the block form at D = 80, the VALU form above 128
    Gaussians) through the per-utterance API with arena scratch, against the batched path; plus what the reference's wrapper does with
    an utterance it cannot align and with an empty one."""
    m = synth.make_model(P, G, D, seed=100 + P, ragged=ragged)
    n = 10
    ut = synth.make_utts(m, n, seed=P, min_phones=2, max_phones=6)
    am, tm = synth.host_objects(m)
    cfg = khg.AlignConfig(beam=200.0, retry_beam=0.0, careful=False)
    accs = khg.AccumAmDiagGmm(); accs.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    tacc, alis = None, []
    for u in range(n):
        feats = np.ascontiguousarray(ut.feats[ut.frame_off[u]: ut.frame_off[u + 1]])
        ans = khg.gmm_align_compiled(am_gmm=am, transition_model=tm, utt=str(u), fst=synth.utt_fst(ut.graphs, u), feats=feats, align_config=cfg,
                                     acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
        assert ans["num_done"] == 1
        alis.append(np.asarray(ans["alignment"], np.int32))
        _, tacc = khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=accs, transition_model=tm, feats=feats, ali=ans["alignment"], transition_accs=tacc)
    cost = np.zeros(m.num_tids + 1, np.float32)
    c = np.asarray(tm.scaled_trans_cost(1.0, 0.1), np.float32); cost[: c.shape[0]] = c
    res, st = _batched(khg, ctx, m, am, ut, n, cost, 200.0, 0.0)
    np.testing.assert_array_equal(np.concatenate(alis), res["ali"])
    occ = np.concatenate([np.asarray(accs.get_acc(p).occupancy) for p in range(P)])
    mean = np.concatenate([np.asarray(accs.get_acc(p).mean_accumulator) for p in range(P)])
    var = np.concatenate([np.asarray(accs.get_acc(p).variance_accumulator) for p in range(P)])
    np.testing.assert_allclose(occ, st["occ"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(mean, st["mean_acc"], rtol=2e-5, atol=2e-6 * np.abs(st["mean_acc"]).max())
    np.testing.assert_allclose(var, st["var_acc"], rtol=2e-5, atol=2e-6 * np.abs(st["var_acc"]).max())
    # an utterance too short for its graph: num_error, empty alignment, counters as decoder-wrappers.cc:35-107 leaves them
    u = int(np.argmax(np.diff(ut.graphs["state_off"])))
    short = np.ascontiguousarray(ut.feats[ut.frame_off[u]: ut.frame_off[u] + 2])
    ans = khg.gmm_align_compiled(am_gmm=am, transition_model=tm, utt="short", fst=synth.utt_fst(ut.graphs, u), feats=short, align_config=cfg,
                                 acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1, num_done=3, num_error=1, tot_like=-5.0, frame_count=7)
    assert (ans["num_done"], ans["num_error"], ans["tot_like"], ans["frame_count"], ans["alignment"], ans["words"]) == (3, 2, -5.0, 7, [], [])
    # no frames at all: nothing accumulated, log_like 0
    before = accs.tot_count
    ll, tacc2 = khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=accs, transition_model=tm, feats=np.zeros((0, D), np.float32), ali=[], transition_accs=tacc.copy())
    assert ll == 0.0 and accs.tot_count == before and np.array_equal(tacc2, tacc)
    with pytest.raises(khg.KhgError):
        khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=accs, transition_model=tm, feats=short, ali=[1], transition_accs=None)       # len(ali) != frames
    with pytest.raises(khg.KhgError):
        khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=accs, transition_model=tm, feats=short, ali=[0, 10 ** 6], transition_accs=None)


def test_scratch_block_is_reused_beside_a_longer_lived_small_set(khg, ctx):
    """The per-call scratch block (DESIGN.md "per-utterance calls") is a stack: a small set that stays alive -- a held-out utterance
    kept resident, say -- must not make every later per-call set add to it until the block is full (the calls would silently fall
    back to hipMalloc).  Results of the resident set stay right while its neighbours' bytes are reused."""
    P, G, D = 30, 8, 13
    m = synth.make_model(P, G, D, seed=5)
    ut = synth.make_utts(m, 6, seed=3, min_phones=3, max_phones=8)
    go, gc, _, miv, iv = synth.host_objects(m)[0].flat()
    dm = khg.DeviceModel(ctx, go, gc, miv, iv)
    tm = khg.DeviceTransitions(ctx, m.id2pdf)

    def one(u):
        f0, f1 = int(ut.frame_off[u]), int(ut.frame_off[u + 1])
        return khg.UtteranceSet(ctx, None, np.array([0, f1 - f0], np.int64), np.ascontiguousarray(ut.feats[f0:f1]))

    assert ctx.get_option("scratch_blocks") == 0
    keep = one(0)
    keep.loglikes(dm)
    ref = np.array(keep.download_loglikes(), copy=True)
    base_bytes, base_blocks = ctx.get_option("scratch_bytes"), ctx.get_option("scratch_blocks")
    assert base_blocks > 0 and base_bytes > 256
    peak = 0
    for it in range(300):
        us = one(1 + it % 5)
        us.loglikes(dm)
        got = us.download_loglikes()
        assert np.isfinite(np.asarray(got)).all()
        peak = max(peak, ctx.get_option("scratch_bytes"))
        us.close()
        assert ctx.get_option("scratch_bytes") == base_bytes and ctx.get_option("scratch_blocks") == base_blocks, it
    assert peak < base_bytes + (4 << 20), "one per-call set's worth above the resident one, not 300"
    np.testing.assert_array_equal(np.asarray(keep.download_loglikes()), ref)
    keep.close()
    assert ctx.get_option("scratch_blocks") == 0 and ctx.get_option("scratch_bytes") == 256
    tm.close(); dm.close()


def test_per_frame_accumulate_is_the_same_device_block(khg, ctx):
    """The loop body of the reference's own scripts/gmm_acc_stats_ali.py:46-56 -- accumulate_for_gmm(model, data=feats[i],
    gmm_index=pdf, weight=1.0) once per frame -- against one _acc_stats_ali call per utterance: same statistics, same log-likes,
    no model upload per frame (the device block stays pending until something reads it), and the two ways mix."""
    P, G, D, n = 300, 16, 40, 3
    m = synth.make_model(P, G, D, seed=21)
    ut = synth.make_utts(m, n, seed=4, min_phones=3, max_phones=6)
    am, tm = synth.host_objects(m)

    def aligned(u):
        f0, f1 = int(ut.frame_off[u]), int(ut.frame_off[u + 1])
        feats = np.ascontiguousarray(ut.feats[f0:f1])
        r = khg.gmm_align_compiled(am_gmm=am, transition_model=tm, utt=str(u), fst=synth.utt_fst(ut.graphs, u), feats=feats,
                                   align_config=khg.AlignConfig(beam=200.0, retry_beam=0.0, careful=False), acoustic_scale=0.1,
                                   transition_scale=1.0, self_loop_scale=0.1)
        return feats, np.asarray(r["alignment"], np.int32)

    a = khg.AccumAmDiagGmm(); a.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    b = khg.AccumAmDiagGmm(); b.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    tot_a = tot_b = 0.0
    for u in range(n):
        feats, ali = aligned(u)
        ll_b, _ = khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=b, transition_model=tm, feats=feats, ali=ali.tolist(), transition_accs=None)
        tot_b += ll_b
        per_frame = []
        for t in range(feats.shape[0]):
            pdf = tm.transition_id_to_pdf(int(ali[t]))
            per_frame.append(a.accumulate_for_gmm(model=am, data=feats[t], gmm_index=pdf, weight=1.0))
        assert a._has_device_stats, "per-frame calls leave their sums on the device"
        tot_a += float(np.sum(per_frame))
        # each return value is that frame's log-likelihood under its pdf
        want = [am.get_pdf(tm.transition_id_to_pdf(int(ali[t]))).log_likelihood(data=feats[t]) for t in range(0, feats.shape[0], 17)]
        np.testing.assert_allclose(per_frame[::17], want, rtol=2e-5, atol=2e-4)
        if u == 1:       # and the per-utterance entry point into the same accumulator (another transition-id table: the block is re-made)
            khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=a, transition_model=tm, feats=feats, ali=ali.tolist(), transition_accs=None)
            khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=b, transition_model=tm, feats=feats, ali=ali.tolist(), transition_accs=None)
    assert tot_a == pytest.approx(tot_b, rel=2e-6)
    assert a.tot_count == pytest.approx(b.tot_count, rel=1e-12) and a.tot_log_like == pytest.approx(b.tot_log_like, rel=2e-6)
    assert not a._has_device_stats
    for p in range(0, P, 7):
        x, y = a.get_acc(p), b.get_acc(p)
        np.testing.assert_allclose(x.occupancy, y.occupancy, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(x.mean_accumulator, y.mean_accumulator, rtol=2e-5, atol=1e-4)
        np.testing.assert_allclose(x.variance_accumulator, y.variance_accumulator, rtol=2e-5, atol=1e-3)
    # weight 0 adds nothing and still returns the likelihood; a fractional weight scales the sums
    c = khg.AccumAmDiagGmm(); c.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    feats, ali = aligned(0)
    ll0 = c.accumulate_for_gmm(model=am, data=feats[0], gmm_index=5, weight=0.0)
    assert ll0 == pytest.approx(am.get_pdf(5).log_likelihood(data=feats[0]), abs=2e-4) and c.tot_count == 0.0
    llh = c.accumulate_for_gmm(model=am, data=feats[0], gmm_index=5, weight=0.5)
    assert llh == pytest.approx(ll0, abs=2e-4)
    assert c.tot_count == pytest.approx(0.5) and c.tot_log_like == pytest.approx(0.5 * llh, rel=1e-5)
    assert float(np.sum(c.get_acc(5).occupancy)) == pytest.approx(0.5, rel=1e-5)


def test_a_closed_context_is_an_error_and_loses_nothing(khg, ctx):
    """The host objects cache device handles (the model, the transition table, the statistics block) made on the default context.
    Closing that context under them must neither crash nor lose pending statistics: entry points refuse a destroyed context
    (RuntimeError), the statistics block is plain device memory that the next default context can still read, the model is
    uploaded again for the new context, and a small set that outlives its context keeps its scratch until it goes."""
    from kaldi_hmm_gmm_amd import _gpu
    P, G, D = 90, 8, 13
    m = synth.make_model(P, G, D, seed=31)
    ut = synth.make_utts(m, 2, seed=8, min_phones=3, max_phones=6)
    am, tm = synth.host_objects(m)
    cfg = khg.AlignConfig(beam=200.0, retry_beam=0.0, careful=False)

    def one(u, accs):
        f0, f1 = int(ut.frame_off[u]), int(ut.frame_off[u + 1])
        feats = np.ascontiguousarray(ut.feats[f0:f1])
        r = khg.gmm_align_compiled(am_gmm=am, transition_model=tm, utt=str(u), fst=synth.utt_fst(ut.graphs, u), feats=feats, align_config=cfg,
                                   acoustic_scale=0.1, transition_scale=1.0, self_loop_scale=0.1)
        khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=accs, transition_model=tm, feats=feats, ali=r["alignment"], transition_accs=None)
        return r["alignment"], f1 - f0

    want = khg.AccumAmDiagGmm(); want.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    ali_want = [one(u, want)[0] for u in range(2)]

    ctx2 = khg.Context(0)
    try:
        _gpu.set_default_context(ctx2)
        accs = khg.AccumAmDiagGmm(); accs.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
        ali0, n0 = one(0, accs)
        assert accs._has_device_stats
        orphan = khg.UtteranceSet(ctx2, None, np.array([0, 40], np.int64), np.ascontiguousarray(ut.feats[:40]))   # a small set left alive
        ctx2.close()
        with pytest.raises(RuntimeError):
            one(1, accs)                                   # the default context is gone: an error, not a use after free
    finally:
        _gpu.set_default_context(ctx)
    ali1, n1 = one(1, accs)                                # new context: the model goes up again, the pending block is read through it
    assert ali0 == ali_want[0] and ali1 == ali_want[1]
    assert accs.tot_count == pytest.approx(n0 + n1) and accs.tot_count == pytest.approx(want.tot_count)
    assert accs.tot_log_like == pytest.approx(want.tot_log_like, rel=1e-6)
    for p in range(0, P, 11):
        np.testing.assert_allclose(accs.get_acc(p).mean_accumulator, want.get_acc(p).mean_accumulator, rtol=2e-5, atol=1e-4)
    before = ctx.get_option("scratch_blocks")
    orphan.close()                                         # its scratch belonged to the closed context: released there, not in this one
    assert ctx.get_option("scratch_blocks") == before


def test_small_set_staged_alignment_reaches_the_device_before_plain_copies(khg, ctx):
    """A small set's khg_ali_upload only stages the alignment in the arena's pinned mirror.  Every path that touches the block with a
    plain copy or memset must flush first: upload -> download with no launch in between returns what was uploaded, and khg_align
    after an upload is not overwritten by the stale staging (its cleared block + the new answer survive)."""
    P, G, D = 30, 8, 13
    m = synth.make_model(P, G, D, seed=5)
    ut = synth.make_utts(m, 3, seed=3, min_phones=3, max_phones=8)
    go, gc, _, miv, iv = synth.host_objects(m)[0].flat()
    dm = khg.DeviceModel(ctx, go, gc, miv, iv)
    tm = khg.DeviceTransitions(ctx, m.id2pdf)
    us = khg.UtteranceSet(ctx, tm, ut.frame_off, ut.feats, graphs=ut.graphs)
    assert ctx.get_option("scratch_blocks") > 0, "3 utterances: a small (arena-backed) set"
    N = int(ut.frame_off[-1])
    fake = (np.arange(N, dtype=np.int32) % m.num_tids) + 1
    us.upload_ali(fake)
    np.testing.assert_array_equal(np.asarray(us.download_ali()), fake)
    fake2 = fake[::-1].copy()
    us.upload_ali(fake2)                    # staged again ...
    us.loglikes(dm, reachable_only=True)
    res = us.align(tm, beam=200.0, acoustic_scale=0.1)       # ... and replaced by the decoder's answer
    assert not np.any(res["status"] & 1)
    got = np.asarray(us.download_ali())
    np.testing.assert_array_equal(got, res["ali"])
    assert not np.array_equal(got, fake2)
    us.close(); tm.close(); dm.close()


def test_small_set_is_refused_on_a_foreign_context(khg, ctx):
    """A small set's scratch belongs to its creator's arena (flushes and the destroy-time wait follow that context): launching on it
    through another live context is KHG_E_ARG -> RuntimeError, not stale reads."""
    import ctypes as C
    from kaldi_hmm_gmm_amd import _lib
    P, G, D = 30, 8, 13
    m = synth.make_model(P, G, D, seed=5)
    ut = synth.make_utts(m, 2, seed=3, min_phones=3, max_phones=8)
    go, gc, _, miv, iv = synth.host_objects(m)[0].flat()
    ctx2 = khg.Context(0)
    try:
        dm2 = khg.DeviceModel(ctx2, go, gc, miv, iv)
        us = khg.UtteranceSet(ctx, None, ut.frame_off, ut.feats)
        rc = _lib.lib.khg_loglikes(C.c_void_p(ctx2.h), C.c_void_p(dm2.h), C.c_void_p(us.h))
        assert rc == -1 and b"context that created it" in _lib.lib.khg_last_error()      # KHG_E_ARG
        ali = np.ones(int(ut.frame_off[-1]), np.int32)
        rc = _lib.lib.khg_ali_upload(C.c_void_p(ctx2.h), C.c_void_p(us.h), _lib.ptr(ali, C.c_int32))
        assert rc == -1
        us.close(); dm2.close()
    finally:
        ctx2.close()


def test_set_zero_with_partial_flags_keeps_the_pending_device_sums(khg, ctx):
    """csrc/mle-am-diag-gmm.cc:35-39: SetZero(flags) zeroes the flagged statistics only.  Statistics still resident on the device
    are folded in first: after set_zero(kGmmMeans) the occupancies, variance sums and the totals are those accumulated so far."""
    P, G, D = 60, 8, 13
    m = synth.make_model(P, G, D, seed=9)
    ut = synth.make_utts(m, 2, seed=4, min_phones=3, max_phones=6)
    am, tm = synth.host_objects(m)
    a = khg.AccumAmDiagGmm(); a.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    b = khg.AccumAmDiagGmm(); b.init(model=am, flags=khg.GmmUpdateFlags.kGmmAll)
    for acc in (a, b):
        for u in range(2):
            f0, f1 = int(ut.frame_off[u]), int(ut.frame_off[u + 1])
            khg.gmm_acc_stats_ali(am_gmm=am, gmm_accs=acc, transition_model=tm, feats=np.ascontiguousarray(ut.feats[f0:f1]),
                                  ali=ut.ref_ali[f0:f1].tolist(), transition_accs=None)
    assert a._has_device_stats
    a.set_zero(khg.GmmUpdateFlags.kGmmMeans)
    assert a.tot_count == b.tot_count == float(ut.frame_off[2]) and a.tot_log_like == b.tot_log_like
    seen = 0.0
    for p in range(P):
        x, y = a.get_acc(p), b.get_acc(p)
        np.testing.assert_array_equal(x.occupancy, y.occupancy)
        np.testing.assert_array_equal(x.variance_accumulator, y.variance_accumulator)
        assert not np.any(x.mean_accumulator)
        seen += float(np.sum(y.occupancy))
    assert seen == pytest.approx(float(ut.frame_off[2]), rel=1e-6)
