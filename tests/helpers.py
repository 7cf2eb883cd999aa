"""Shared helpers for the parity tests: build oracle / device objects from a synthetic set."""
import numpy as np

from kaldi_hmm_gmm_amd import synth
from oracle import oracle as orc


def build(num_pdfs, gauss, dim, n_utt, seed=1, ragged=False, min_phones=2, max_phones=6, tscale=1.0, slscale=0.1):
    m = synth.make_model(num_pdfs, gauss, dim, seed=20230414 + seed, ragged=ragged)
    gc = orc.model_gconsts(m.gauss_off, m.weights, m.inv_vars, m.means_invvars)
    om = orc.OModel(m.gauss_off, gc, m.means_invvars, m.inv_vars)
    ut = synth.make_utts(m, n_utt, seed=seed, min_phones=min_phones, max_phones=max_phones)
    # what AddTransitionProbs adds per tid, from the oracle (hmm-utils.cc:442-463)
    il = np.arange(m.num_tids + 1, dtype=np.int32)
    cost = orc.add_transition_probs(il, np.zeros(m.num_tids + 1, np.float32), m.log_probs, m.non_self_loop_log_probs,
                                    m.id2state, m.is_self_loop, tscale, slscale)
    return m, gc, om, ut, cost


def oracle_graph(ut, u, cost):
    g = dict(ut.graphs)
    w = g["weight"].copy()
    il = g["ilabel"]
    w = np.where(il >= 1, w + cost[np.maximum(il, 0)], w).astype(np.float32)
    g["weight"] = w
    return orc.OGraph.from_set(g, u)


def utt_feats(ut, u):
    return ut.feats[ut.frame_off[u]: ut.frame_off[u + 1]]


def exact_loglikes(m, gc, feats, pdfs):
    """float64 evaluation of decodable-am-diag-gmm.cc:55-61 and the magnitude bound
    B = max_g(|gconst| + sum|M x| + 0.5 sum|V x^2|) that fp32 rounding error scales with."""
    x = feats.astype(np.float64)
    out = np.zeros((len(pdfs), x.shape[0]))
    bound = np.zeros_like(out)
    for j, p in enumerate(pdfs):
        a, b = m.gauss_off[p], m.gauss_off[p + 1]
        miv = m.means_invvars[a:b].astype(np.float64)
        iv = m.inv_vars[a:b].astype(np.float64)
        g = gc[a:b].astype(np.float64)
        ll = g[None, :] + x @ miv.T - 0.5 * (x * x) @ iv.T
        mx = ll.max(1, keepdims=True)
        out[j] = (mx + np.log(np.exp(ll - mx).sum(1, keepdims=True)))[:, 0]
        bb = np.abs(g)[None, :] + np.abs(x) @ np.abs(miv).T + 0.5 * (x * x) @ iv.T
        bound[j] = bb.max(1)
    return out, bound


def token_path_cost(graphs, u, ali, ll_u, pdf_list, cost, id2pdf, acoustic_scale):
    """Walk utterance u's graph along the transition-id sequence `ali` (out-arcs of a state carry distinct ids in the
    compiled training graphs): -> (is an accepting path, its cost in the token arithmetic of faster-decoder.h:119-137:
    double(prev) + float(arc weight + AddTransitionProbs) + float(-(scale * ll)), left to right, + final weight)."""
    g = graphs
    s0 = int(g["state_off"][u])
    ao = g["arc_off"]
    col = {int(p): j for j, p in enumerate(pdf_list)}
    st = int(g["start"][u])
    scale = np.float32(acoustic_scale)
    terms = np.empty(2 * len(ali), np.float64)
    for t, tid in enumerate(ali):
        a0, a1 = int(ao[s0 + st]), int(ao[s0 + st + 1])
        k = np.nonzero(g["ilabel"][a0:a1] == tid)[0]
        if k.size != 1:
            return False, np.inf
        a = a0 + int(k[0])
        terms[2 * t] = np.float32(g["weight"][a] + cost[tid])                              # AddTransitionProbs: float add
        terms[2 * t + 1] = np.float32(-1) * (scale * ll_u[col[int(id2pdf[tid])], t])        # decodable-am-diag-gmm.h:96
        st = int(g["nextstate"][a])
    fin = g["final"][s0 + st]
    if not np.isfinite(fin):
        return False, np.inf
    return True, float(np.add.accumulate(terms)[-1] + np.float64(fin))
