"""Synthetic HMM-GMM workloads: model, transition tables, per-utterance
training graphs and features sampled from the model.  Used by tests and bench.py.

Graph shape follows what the reference's TrainingGraphCompiler emits for a linear transcript
with reorder=true (csrc/hmm-utils.cc:293-369): a chain g_0 .. g_S where
the arc g_i -> g_{i+1} carries the FORWARD transition-id of HMM-state i and g_{i+1} carries the
self-loop of HMM-state i, so a state held d frames emits [forward, self-loop x (d-1)].
"""
from dataclasses import dataclass

import numpy as np

CONFIGS = {
    # name: (num_pdfs, gauss_per_pdf, dim)   -- BASELINE.json configs[1..4]
    "mono100x8": (100, 8, 39),
    "tri2000x32": (2000, 32, 40),
    "tri5000x64": (5000, 64, 40),
    "stress10000x128": (10000, 128, 80),
    # probe shapes (not BASELINE configs): config #5's pdfs with a model small enough for the Infinity Cache
    "probe1000x128": (1000, 128, 80),
}


@dataclass
class SynthModel:
    gauss_off: np.ndarray       # int32 [P+1]
    weights: np.ndarray         # float32 [sumG]
    means: np.ndarray           # float32 [sumG, D]
    vars: np.ndarray            # float32 [sumG, D]
    inv_vars: np.ndarray        # float32 [sumG, D]
    means_invvars: np.ndarray   # float32 [sumG, D]
    # transition tables of a 3-state left-to-right monophone-style model
    # (csrc/transition-model.cc:254-303,318-359): tstate = pdf+1, tids (2*pdf+1: self-loop, 2*pdf+2: forward)
    id2pdf: np.ndarray          # int32 [num_tids+1]
    id2state: np.ndarray        # int32 [num_tids+1]
    is_self_loop: np.ndarray    # uint8 [num_tids+1]
    state2id: np.ndarray        # int32 [num_tstates+2]
    self_loop_of: np.ndarray    # int32 [num_tstates+1]
    log_probs: np.ndarray       # float32 [num_tids+1]
    non_self_loop_log_probs: np.ndarray  # float32 [num_tstates+1]

    @property
    def num_pdfs(self):
        return self.gauss_off.shape[0] - 1

    @property
    def dim(self):
        return self.means.shape[1]

    @property
    def num_tids(self):
        return self.id2pdf.shape[0] - 1


def make_model(num_pdfs, gauss, dim, seed=20230414, ragged=False, self_loop_prob=0.75, mean_scale=3.0, gauss_counts=None):
    """mean_scale: spread of the Gaussian means (3.0 = a trained-like model, Gaussians ~27 sigma apart at D = 40).
    A small spread (0.1 .. 0.3) gives CONFUSABLE pdfs, the regime of a recipe's first realign passes: the best path leaves a narrow
    beam now and then, the reference's pruning decides the answer and khg_align must follow it token for token."""
    rng = np.random.default_rng(seed)
    if gauss_counts is not None:           # explicit Gaussians per pdf (a model some of whose pdfs a split has grown)
        g = np.asarray(gauss_counts, np.int64)
        assert g.shape == (num_pdfs,) and (g >= 1).all()
    elif ragged:
        g = rng.integers(max(1, gauss // 2), gauss + 1, size=num_pdfs)
    else:
        g = np.full(num_pdfs, gauss)
    gauss_off = np.concatenate([[0], np.cumsum(g)]).astype(np.int32)
    sumG = int(gauss_off[-1])
    means = (mean_scale * rng.standard_normal((sumG, dim))).astype(np.float32)
    var = rng.uniform(0.5, 2.0, size=(sumG, dim)).astype(np.float32)
    w = rng.uniform(0.5, 1.5, size=sumG).astype(np.float32)
    for p in range(num_pdfs):
        sl = slice(gauss_off[p], gauss_off[p + 1])
        w[sl] /= w[sl].sum()
    inv_vars = (1.0 / var).astype(np.float32)
    means_invvars = (means * inv_vars).astype(np.float32)
    nt = 2 * num_pdfs
    tid = np.arange(nt + 1)
    id2pdf = np.where(tid > 0, (tid - 1) // 2, 0).astype(np.int32)
    id2state = (id2pdf + 1).astype(np.int32)
    id2state[0] = 0
    is_self_loop = ((tid % 2) == 1).astype(np.uint8)
    is_self_loop[0] = 0
    state2id = (2 * np.arange(num_pdfs + 2) - 1).astype(np.int32)  # state2id[ts] = 2(ts-1)+1
    state2id[0] = 0
    self_loop_of = state2id[: num_pdfs + 1].copy()
    self_loop_of[0] = 0
    log_probs = np.zeros(nt + 1, np.float32)
    log_probs[1::2] = np.log(np.float32(self_loop_prob))
    log_probs[2::2] = np.log(np.float32(1.0 - self_loop_prob))
    nsl = np.zeros(num_pdfs + 1, np.float32)
    nsl[1:] = np.log(np.float32(1.0) - np.exp(np.log(np.float32(self_loop_prob))))
    return SynthModel(gauss_off, w, means, var, inv_vars, means_invvars, id2pdf, id2state, is_self_loop, state2id,
                      self_loop_of, log_probs, nsl)


@dataclass
class SynthUtts:
    frame_off: np.ndarray   # int64 [U+1]
    feats: object           # float32 [N, D] numpy, or None when generated on the device
    ref_ali: np.ndarray     # int32 [N] generating transition-id sequence
    frame_pdf: np.ndarray   # int32 [N]
    graphs: dict            # CSR arrays (see device.UtteranceSet)
    num_phones: np.ndarray  # [U]


def zipf_phone_stream(rng, nphones, n, vocab=20000, exponent=1.0, sil_prob=0.2):
    """n phones of running text: words drawn from a lexicon of `vocab` words (2..8 phones each, fixed per word) with Zipf
    frequencies rank^-exponent, phone 0 (silence) between two words with probability sil_prob.  Utterances cut from this stream
    share pdfs the way real transcripts do (frequent words, silence), unlike independent uniform phones."""
    wlen = rng.integers(2, 9, size=vocab)
    woff = np.concatenate([[0], np.cumsum(wlen)])
    wphones = rng.integers(1 if nphones > 1 else 0, nphones, size=int(woff[-1]))
    p = 1.0 / np.arange(1, vocab + 1) ** exponent
    nw = n // 2 + 2                                       # words of >= 2 phones: enough for n phones
    words = rng.choice(vocab, size=nw, p=p / p.sum())
    sil = rng.random(nw) < sil_prob
    seg = wlen[words] + sil                               # phones per word incl. its leading silence
    start = np.concatenate([[0], np.cumsum(seg)])[:-1]
    total = int(seg.sum())
    pos = np.arange(total) - np.repeat(start, seg)        # position inside the segment
    w_of = np.repeat(words, seg)
    s_of = np.repeat(sil, seg)
    idx = woff[w_of] + np.maximum(pos - s_of, 0)
    out = np.where(s_of & (pos == 0), 0, wphones[idx])
    return out[:n]


def make_utts(model: SynthModel, n_utt, seed=1, min_phones=10, max_phones=40, leave_prob=0.25, feats=True,
              shuffle_phones=True, transcripts="uniform"):
    """Utterances of L ~ U{min..max} phones x 3 HMM states; state durations 1 + Geom(leave_prob).
    transcripts: "uniform" = independent uniform phones; "zipf" = running text from a Zipf lexicon (zipf_phone_stream);
    "skew" = uniform, but every second phone is phone 0."""
    rng = np.random.default_rng(seed)
    P = model.num_pdfs
    nphones = P // 3
    assert nphones >= 1, "need at least 3 pdfs"
    L = rng.integers(min_phones, max_phones + 1, size=n_utt)
    nst = 3 * L                                   # emitting HMM states per utterance
    tot_states = int(nst.sum())
    if transcripts == "zipf":
        phones = zipf_phone_stream(rng, nphones, int(L.sum()))
    else:
        phones = rng.integers(0, nphones, size=int(L.sum()))
        if transcripts == "skew":                  # half of all phones are phone 0: its pdfs hold many times the average pdf's frames
            phones = np.where(rng.random(phones.shape[0]) < 0.5, 0, phones)
    state_pdf = (3 * np.repeat(phones, 3) + np.tile(np.arange(3), phones.shape[0])).astype(np.int32)
    dur = rng.geometric(leave_prob, size=tot_states).astype(np.int64)  # >= 1
    st_off = np.concatenate([[0], np.cumsum(nst)])
    T = np.add.reduceat(dur, st_off[:-1])
    frame_off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    N = int(frame_off[-1])
    frame_pdf = np.repeat(state_pdf, dur)
    # generating alignment: forward tid on the first frame of each state, self-loop afterwards
    first = np.zeros(N, bool)
    first[np.concatenate([[0], np.cumsum(dur)[:-1]])] = True
    ref_ali = np.where(first, 2 * frame_pdf + 2, 2 * frame_pdf + 1).astype(np.int32)

    # graphs: per utterance S = nst+1 states, arcs: g_0:[fwd0]; g_i:[fwd_i, loop_{i-1}]; g_S:[loop_{S-1}]
    S = nst + 1
    state_off = np.concatenate([[0], np.cumsum(S)]).astype(np.int64)
    NS = int(state_off[-1])
    narcs_state = np.full(NS, 2, np.int64)
    narcs_state[state_off[:-1]] = 1
    narcs_state[state_off[1:] - 1] = 1
    arc_off = np.concatenate([[0], np.cumsum(narcs_state)]).astype(np.int64)
    NA = int(arc_off[-1])
    ilabel = np.zeros(NA, np.int32)
    nextstate = np.zeros(NA, np.int32)
    local = np.arange(NS) - np.repeat(state_off[:-1], S)           # local state index
    # emitting-state index (into state_pdf) of local state i of utterance u is st_off[u] + i
    base = np.repeat(st_off[:-1], S)
    is_last = local == np.repeat(nst, S)
    is_first = local == 0
    # forward arc (first arc of every non-last state)
    fwd_states = np.nonzero(~is_last)[0]
    a = arc_off[fwd_states]
    ilabel[a] = 2 * state_pdf[base[fwd_states] + local[fwd_states]] + 2
    nextstate[a] = local[fwd_states] + 1
    # self-loop arc (last arc of every non-first state)
    loop_states = np.nonzero(~is_first)[0]
    a = arc_off[loop_states + 1] - 1
    ilabel[a] = 2 * state_pdf[base[loop_states] + local[loop_states] - 1] + 1
    nextstate[a] = local[loop_states]
    final = np.full(NS, np.inf, np.float32)
    final[state_off[1:] - 1] = 0.0
    graphs = {
        "state_off": state_off,
        "start": np.zeros(n_utt, np.int32),
        "arc_off": arc_off,
        "ilabel": ilabel,
        "olabel": np.zeros(NA, np.int32),
        "weight": np.zeros(NA, np.float32),
        "nextstate": nextstate,
        "final": final,
    }
    x = sample_feats(model, frame_pdf, rng) if feats else None
    return SynthUtts(frame_off, x, ref_ali, frame_pdf.astype(np.int32), graphs, L)


def sample_feats(model: SynthModel, frame_pdf, rng, chunk=1 << 18):
    """x ~ N(mean, var) of a component drawn by weight from the frame's pdf (numpy, host)."""
    N = frame_pdf.shape[0]
    D = model.dim
    out = np.empty((N, D), np.float32)
    go = model.gauss_off
    cw = np.cumsum(model.weights.astype(np.float64))
    base = np.concatenate([[0.0], cw])[go[:-1]]   # cumulative weight before each pdf's first Gaussian
    for s in range(0, N, chunk):
        p = frame_pdf[s: s + chunk]
        r = rng.random(p.shape[0]) * (cw[go[p + 1] - 1] - base[p]) + base[p]
        comp = np.searchsorted(cw, r, side="right")
        comp = np.minimum(np.maximum(comp, go[p]), go[p + 1] - 1)
        z = rng.standard_normal((p.shape[0], D)).astype(np.float32)
        out[s: s + chunk] = model.means[comp] + np.sqrt(model.vars[comp]) * z
    return out


def sample_feats_torch(model: SynthModel, frame_pdf, seed, device, chunk=1 << 22, keep=None):
    """Same law as sample_feats, generated directly in HBM with torch (bench-sized sets).
    keep: optional boolean mask over the frames -- only those rows are stored (a rank's shard of the ONE global
    set: every rank draws the whole stream chunk by chunk, so frame i has the same bits whoever owns it)."""
    import torch

    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    N = frame_pdf.shape[0]
    D = model.dim
    n_out = N if keep is None else int(np.count_nonzero(keep))
    out = torch.empty((n_out, D), dtype=torch.float32, device=device)
    go = torch.as_tensor(model.gauss_off.astype(np.int64), device=device)
    cw = torch.cumsum(torch.as_tensor(model.weights, device=device).double(), 0)
    base = torch.cat([torch.zeros(1, device=device, dtype=torch.float64), cw])[go[:-1]]
    means = torch.as_tensor(model.means, device=device)
    std = torch.sqrt(torch.as_tensor(model.vars, device=device))
    fp = torch.as_tensor(frame_pdf.astype(np.int64), device=device)
    cur = 0
    for s in range(0, N, chunk):
        p = fp[s: s + chunk]
        r = torch.rand(p.shape[0], device=device, dtype=torch.float64, generator=gen)
        r = r * (cw[go[p + 1] - 1] - base[p]) + base[p]
        comp = torch.searchsorted(cw, r, right=True)
        comp = torch.minimum(torch.maximum(comp, go[p]), go[p + 1] - 1)
        z = torch.randn((p.shape[0], D), device=device, dtype=torch.float32, generator=gen)
        if keep is None:
            out[s: s + chunk] = means[comp] + std[comp] * z
        else:
            k = torch.as_tensor(keep[s: s + chunk], device=device)
            n = int(k.sum())
            if n:
                ck = comp[k]
                out[cur: cur + n] = means[ck] + std[ck] * z[k]
            cur += n
    return out


def mismatched_model(model: SynthModel, fraction, seed=0):
    """The model a recipe's EARLY realign passes work with, in caricature: a copy of `model` (uniform Gaussians per pdf) in which
    a random `fraction` of the pdfs have traded their parameters among themselves -- sure of itself and wrong there.  Features
    sampled from `model` and aligned with the copy put path costs hundreds apart wherever a traded pdf is on the graph: a narrow
    beam then prunes the best path, the reference's answer depends on its pruning order, retries and failures occur."""
    import dataclasses
    rng = np.random.default_rng(seed)
    P = model.num_pdfs
    G = int(model.gauss_off[1] - model.gauss_off[0])
    assert (np.diff(model.gauss_off) == G).all(), "uniform Gaussians per pdf"
    sel = np.nonzero(rng.random(P) < fraction)[0]
    perm = np.arange(P)
    if sel.size > 1:
        perm[sel] = np.roll(rng.permutation(sel), 1)
    idx = (perm[:, None] * G + np.arange(G)[None, :]).reshape(-1)
    return dataclasses.replace(model, weights=model.weights[idx].copy(), means=model.means[idx].copy(), vars=model.vars[idx].copy(),
                               inv_vars=model.inv_vars[idx].copy(), means_invvars=model.means_invvars[idx].copy())


def host_objects(model: SynthModel):
    """The host classes the reference's scripts take (AmDiagGmm, TransitionModel) for a synthetic model: P // 3 phones (ids 1..)
    of three left-to-right states (self-loop 0.75 / forward 0.25, scripts/prepare_lang.py:514-560), a monophone tree -- its
    transition-ids are make_model's (2 pdf + 1 self-loop, 2 pdf + 2 forward; pdf = 3 (phone - 1) + state).  Pdfs beyond
    3 (P // 3) exist in the AmDiagGmm only."""
    from . import _lib
    import ctypes as C
    from .context_dep import monophone_context_dependency
    from .diag_gmm import AmDiagGmm, DiagGmm
    from .hmm_topology import HmmTopology
    from .transition_model import TransitionModel
    P, D = model.num_pdfs, model.dim
    nph = P // 3
    s = "<Topology> <TopologyEntry> <ForPhones> " + " ".join(str(i) for i in range(1, nph + 1)) + "\n</ForPhones> "
    for i in range(3):
        s += f"<State> {i} <PdfClass> {i} <Transition> {i} 0.75 <Transition> {i + 1} 0.25 </State> "
    s += "<State> 3 </State> </TopologyEntry> </Topology>"
    topo = HmmTopology()
    topo.read(s)
    tree = monophone_context_dependency(list(range(1, nph + 1)), topo.get_phone_to_num_pdf_classes())
    tm = TransitionModel(ctx_dep=tree, hmm_topo=topo)
    gc = np.zeros(model.weights.shape[0], np.float32)
    _lib.check(_lib.lib.khg_compute_gconsts(P, D, _lib.ptr(model.gauss_off, C.c_int32), _lib.ptr(model.weights, C.c_float),
                                            _lib.ptr(model.inv_vars, C.c_float), _lib.ptr(model.means_invvars, C.c_float),
                                            _lib.ptr(gc, C.c_float), None))
    am = AmDiagGmm()
    am.init(DiagGmm(nmix=1, dim=D), P)
    am.set_flat(model.gauss_off, model.weights, gc, model.means_invvars, model.inv_vars)
    return am, tm


def utt_fst(graphs, u):
    """Utterance u of a CSR graph set as the StdVectorFst the reference's scripts pass around."""
    from .fst import StdVectorFst
    so = graphs["state_off"]
    s0, s1 = int(so[u]), int(so[u + 1])
    ao = graphs["arc_off"][s0: s1 + 1]
    a0, a1 = int(ao[0]), int(ao[-1])
    return StdVectorFst.from_csr(int(graphs["start"][u]), (ao - a0).astype(np.int64), graphs["ilabel"][a0:a1], graphs["olabel"][a0:a1],
                                 graphs["weight"][a0:a1], graphs["nextstate"][a0:a1], graphs["final"][s0:s1])
