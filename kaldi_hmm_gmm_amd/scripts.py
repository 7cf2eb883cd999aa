"""The four "binary-level" entry points of the reference (scripts/gmm_init_mono.py,
gmm_align_compiled.py, gmm_acc_stats_ali.py, gmm_est.py) with the same names, keyword arguments
and return conventions, plus batched variants that keep a whole shard on the GPU."""
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _gpu
from ._lib import KhgError
from .align import AlignConfig, DecodableAmDiagGmmScaled, add_transition_probs, align_batch, align_utterance_wrapper
from .context_dep import monophone_context_dependency, monophone_context_dependency_shared
from .device import DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet
from .diag_gmm import AmDiagGmm, DiagGmm
from .fst import StdVectorFst
from .hmm_topology import HmmTopology
from .mle import (AccumAmDiagGmm, GmmUpdateFlags, MleDiagGmmOptions, mle_am_diag_gmm_update, str_to_gmm_flags)
from .transition_model import MleTransitionUpdateConfig, TransitionModel


def gmm_init_mono(topo: HmmTopology, cuts, shared_phones: Optional[List[List[int]]] = None,
                  perturb_factor: float = 0.0):
    """scripts/gmm_init_mono.py:10-73.  `cuts` is anything with compute_global_feature_stats()
    (lhotse CutSet), a dict with norm_means / norm_stds, or a [N, D] feature matrix."""
    if hasattr(cuts, "compute_global_feature_stats"):
        stats = cuts.compute_global_feature_stats()
        means, stds = np.asarray(stats["norm_means"]), np.asarray(stats["norm_stds"])
    elif isinstance(cuts, dict):
        means, stds = np.asarray(cuts["norm_means"]), np.asarray(cuts["norm_stds"])
    else:
        x = np.asarray(cuts, np.float64)
        means, stds = x.mean(0), x.std(0)
    means = means.astype(np.float32)[None, :]
    variances = np.square(stds.astype(np.float32))[None, :]
    feat_dim = means.shape[1]
    p2n = topo.get_phone_to_num_pdf_classes()
    tree = (monophone_context_dependency(topo.phones, p2n) if shared_phones is None
            else monophone_context_dependency_shared(shared_phones, p2n))
    g = DiagGmm(nmix=1, dim=feat_dim)
    g.set_weights(np.ones(1, np.float32))
    g.set_means(means)
    g.set_invvars(1 / variances)
    g.compute_gconsts()
    am = AmDiagGmm()
    for _ in range(tree.num_pdfs):
        am.add_pdf(g)
    if perturb_factor != 0:
        for i in range(tree.num_pdfs):
            am.get_pdf(i).perturb(perturb_factor)
    return TransitionModel(ctx_dep=tree, hmm_topo=topo), tree, am


def gmm_info(am_gmm: AmDiagGmm, transition_model: TransitionModel) -> Dict[str, int]:
    """scripts/gmm_info.py:9-29 (the key "feature_dimensition" is spelled as the reference spells it)."""
    return {"number_of_phones": len(transition_model.phones), "number_of_pdfs": transition_model.num_pdfs,
            "number_of_transition_ids": transition_model.num_transition_ids,
            "number_of_transition_states": transition_model.num_transition_states, "feature_dimensition": am_gmm.dim,
            "number_of_gaussians": am_gmm.num_gauss}


def gmm_align_compiled(am_gmm: AmDiagGmm, transition_model: TransitionModel, utt: str, fst: StdVectorFst, feats,
                       align_config: AlignConfig, acoustic_scale: float = 1.0, transition_scale: float = 1.0,
                       self_loop_scale: float = 1.0, num_done: int = 0, num_error: int = 0, num_retried: int = 0,
                       tot_like: float = 0, frame_count: int = 0) -> Dict[str, Any]:
    """scripts/gmm_align_compiled.py:10-79 (mutates `fst` like the reference: callers pass a copy)."""
    add_transition_probs(trans_model=transition_model, transition_scale=transition_scale,
                         self_loop_scale=self_loop_scale, fst=fst)
    dec = DecodableAmDiagGmmScaled(am=am_gmm, tm=transition_model, feats=feats, scale=acoustic_scale)
    (num_done, num_error, num_retried, tot_like, frame_count, alignment, words) = align_utterance_wrapper(
        config=align_config, utt=utt, acoustic_scale=acoustic_scale, fst=fst, decodable=dec, num_done=num_done,
        num_error=num_error, num_retried=num_retried, tot_like=tot_like, frame_count=frame_count)
    return {"num_done": num_done, "num_error": num_error, "num_retried": num_retried, "tot_like": tot_like,
            "frame_count": frame_count, "alignment": alignment, "words": words}


def gmm_align_compiled_batch(am_gmm: AmDiagGmm, transition_model: TransitionModel, utts: Sequence[str],
                             fsts: Sequence[StdVectorFst], feats: Sequence[np.ndarray], align_config: AlignConfig,
                             acoustic_scale: float = 1.0, transition_scale: float = 1.0, self_loop_scale: float = 1.0):
    """All utterances of a shard in one GPU pass.  Graphs are NOT mutated: the per-transition-id cost
    AddTransitionProbs would add is applied on the device.  Returns the same counters as the loop of
    single calls would, plus per-utterance alignments / words."""
    cost = transition_model.scaled_trans_cost(transition_scale, self_loop_scale)
    res = align_batch(am_gmm, transition_model, list(fsts), list(feats), align_config, acoustic_scale, trans_cost=cost)
    out = {"num_done": 0, "num_error": 0, "num_retried": 0, "tot_like": 0.0, "frame_count": 0, "alignment": [],
           "words": [], "utts": list(utts)}
    for r in res:
        out["num_retried"] += int(r["retried"])
        if r["ok"]:
            out["num_done"] += 1
            out["tot_like"] += r["like"]
            out["frame_count"] += r["num_frames"]
        else:
            out["num_error"] += 1
        out["alignment"].append(r["alignment"])
        out["words"].append(r["words"])
    return out


def gmm_acc_stats_ali(am_gmm: AmDiagGmm, gmm_accs: AccumAmDiagGmm, transition_model: TransitionModel, feats,
                      ali: List[int], transition_accs: Optional[np.ndarray] = None):
    """scripts/gmm_acc_stats_ali.py:9-58 -> (log_like, transition_accs); gmm_accs is updated in place.
    One K3 pass over the utterance; the model is on the device already (cached on am_gmm, uploaded again only after it changed) and
    the statistics STAY on the device between calls (gmm_accs adds them into its host accumulators when something reads those:
    get_acc, tot_count, mle_am_diag_gmm_update, pickling) -- the reference's caller makes this call once per utterance."""
    feats = np.ascontiguousarray(feats, np.float32)
    return gmm_accs._acc_stats_ali(am_gmm, transition_model, feats, ali, transition_accs)


def gmm_acc_stats_ali_batch(am_gmm: AmDiagGmm, gmm_accs: AccumAmDiagGmm, transition_model: TransitionModel,
                            feats: Sequence[np.ndarray], alis: Sequence[Sequence[int]],
                            transition_accs: Optional[np.ndarray] = None):
    """Several utterances, same contract: -> (total log_like, transition_accs)."""
    tot = 0.0
    if transition_accs is None:
        transition_accs = transition_model.init_stats()
    for f, a in zip(feats, alis):
        ll, transition_accs = gmm_acc_stats_ali(am_gmm, gmm_accs, transition_model, f, a, transition_accs)
        tot += ll
    return tot, transition_accs


def gmm_est(am_gmm: AmDiagGmm, gmm_accs: AccumAmDiagGmm, transition_model: TransitionModel, transition_accs,
            tcfg: MleTransitionUpdateConfig, gmm_opts: MleDiagGmmOptions, mixup: int = 0, mixdown: int = 0,
            perturb_factor: float = 0.01, power: float = 0.2, min_count: float = 20.0, update_flags: str = "mvwt",
            verbose: bool = True, randn=None) -> Dict[str, float]:
    """scripts/gmm_est.py:8-96.  Returns the printed statistics as a dict as well."""
    flags = str_to_gmm_flags(update_flags)
    info = {}
    if int(flags) & int(GmmUpdateFlags.kGmmTransitions):
        objf_impr, count = transition_model.mle_update(transition_accs, tcfg)
        info["transition_objf_impr"], info["transition_count"] = objf_impr, count
        if verbose:
            print("Transition model update: Overall", objf_impr / count, "log-like improvement per frame over", count, "frames.")
    tot_like, tot_t = gmm_accs.tot_log_like, gmm_accs.tot_count
    objf_impr, count = mle_am_diag_gmm_update(config=gmm_opts, amdiag_gmm_acc=gmm_accs, flags=flags, am_gmm=am_gmm)
    info.update(gmm_objf_impr=objf_impr, gmm_count=count, avg_like=tot_like / tot_t if tot_t else float("nan"), frames=tot_t)
    if verbose:
        print("GMM update: Overall", objf_impr / count, "objective function improvement per frame over", count, "frames")
        print("GMM update: Overall avg like per frame =", tot_like / tot_t, "over", tot_t, "frames.")
    if mixup != 0 or mixdown != 0:
        pdf_occs = np.asarray([gmm_accs.get_acc(i).occupancy.sum() for i in range(gmm_accs.num_accs)], np.float32)
        if mixdown != 0:
            am_gmm.merge_by_count(state_occs=pdf_occs, target_components=mixdown, power=power, min_count=min_count)
        if mixup != 0:
            am_gmm.split_by_count(state_occs=pdf_occs, target_components=mixup, perturb_factor=perturb_factor,
                                  power=power, min_count=min_count, randn=randn)
    return info


def gmm_boost_silence(am_gmm: AmDiagGmm, transition_model: TransitionModel, silence_phones: List[int], boost: float = 1.5,
                      verbose: bool = False) -> AmDiagGmm:
    """scripts/gmm_boost_silence.py:10-45: a COPY of am_gmm with the weights of the silence phones' pdfs scaled by `boost`
    (gconsts recomputed); the argument is left untouched, as in the reference, whose recipe does
    `am = gmm_boost_silence(am_gmm=am, ...)` (egs/yesno/train.py:158).  silence_phones is sorted in place like there."""
    from .transition_model import get_pdfs_for_phones
    if len(silence_phones) == 0:
        raise KhgError("gmm_boost_silence: no silence phones")
    silence_phones.sort()
    is_unique, pdfs = get_pdfs_for_phones(transition_model, silence_phones)
    if not is_unique and verbose:
        print("The pdfs for the silence phones may be shared by other phones (note: this probably does not matter.)")
    dgm = AmDiagGmm()
    dgm.copy_from_am_diag_gmm(am_gmm)
    for pdf in pdfs:
        g = dgm.get_pdf(pdf)
        g.set_weights(g.weights * np.float32(boost))
        g.compute_gconsts()
    if verbose:
        print("Boosted weights for", len(pdfs), "pdfs, by factor of", boost)
    return dgm
