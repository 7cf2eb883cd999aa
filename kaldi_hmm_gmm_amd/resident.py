"""ResidentEm: the EM loop of the reference's recipe (egs/yesno/train.py:131-222 -- gmm_boost_silence,
gmm_align_compiled, gmm_acc_stats_ali, gmm_est per pass) for a fixed shard of utterances whose features,
decoding graphs, alignments, accumulators AND model stay in HBM across passes.

The per-call scripts (scripts.py) mirror the reference's function signatures and therefore re-upload their
arguments on every call; this class is what a multi-pass recipe uses instead -- the configuration BASELINE's
metric is quoted on ("model, graphs and features resident in HBM").  Per pass:

    align()       K1 (reachable cells only) + K2; alignments stay on the device
    accumulate()  K3 into the fp64 block, then ONE all-reduce of the block when torch.distributed is
                  initialised (utterances are sharded over ranks, csrc/mle-am-diag-gmm.cc:119-128 == gmm-sum-accs)
    update()      transition update on the host (a few kB come down), GMM update on the device (K4,
                  khg_model_mle_update), mixing up on the device too (khg_model_split: the per-pdf occupancies come
                  down, the normal deviates DiagGmm::Split needs go up; scripts/gmm_est.py:66-84); every rank holds the
                  same all-reduced sums and draws the same deviates, so every rank computes the same new model and
                  nothing is broadcast.

Utterances whose alignment fails contribute no statistics until they align again."""
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _gpu
from ._lib import KhgError
from .align import AlignConfig, FasterDecoderOptions
from .device import ALIGN_ERROR, ALIGN_RETRIED, DeviceAccs, DeviceModel, DeviceTransitions, UtteranceSet
from .diag_gmm import AmDiagGmm
from .fst import StdVectorFst, concat_graphs
from .mle import GmmUpdateFlags, MleDiagGmmOptions, get_split_targets, str_to_gmm_flags
from .transition_model import MleTransitionUpdateConfig, TransitionModel, get_pdfs_for_phones


class ResidentEm:
    def __init__(self, am_gmm: AmDiagGmm, transition_model: TransitionModel, fsts: Sequence[StdVectorFst],
                 feats: Sequence[np.ndarray], acoustic_scale: float = 1.0, transition_scale: float = 1.0,
                 self_loop_scale: float = 1.0, ctx=None, split_seed: int = 0, sharded_mstep: bool = False):
        if len(fsts) != len(feats):
            raise KhgError("ResidentEm: one decoding graph per utterance")
        self.am, self.tm = am_gmm, transition_model
        self.acoustic_scale, self.transition_scale, self.self_loop_scale = acoustic_scale, transition_scale, self_loop_scale
        self.ctx = ctx or _gpu.default_context()
        feats = [np.asarray(f, np.float32).reshape(-1, am_gmm.dim) for f in feats]
        self.frame_off = np.concatenate([[0], np.cumsum([f.shape[0] for f in feats])]).astype(np.int64)
        allf = np.concatenate(feats) if feats else np.zeros((0, am_gmm.dim), np.float32)
        self.dt = DeviceTransitions(self.ctx, np.asarray(transition_model.transition_id_to_pdf_array(), np.int32))
        self._set_trans_cost()
        self.us = UtteranceSet(self.ctx, self.dt, self.frame_off, allf, graphs=concat_graphs(list(fsts)))
        self.dm: Optional[DeviceModel] = None
        self.accs: Optional[DeviceAccs] = None
        self._upload_model()
        self.host_in_sync = True      # am_gmm holds the device model's parameters
        self.split_seed, self._updates = int(split_seed), 0
        self._comm, self._comm_made = None, False
        # SURVEY 8f-3 as written (multi-GPU, nccl): the statistics are REDUCED by pdf range to their owner instead of all-reduced,
        # every rank updates its own pdfs and broadcasts the rows (khg_model_mle_update_sharded): ~25 % fewer bytes over xGMI at
        # N = 8.  Default: the block is all-reduced (pipelined behind K3) and every rank runs the whole 0.5 ms update.
        self.sharded_mstep = bool(sharded_mstep)

    # -- plumbing ------------------------------------------------------------------------------
    def _set_trans_cost(self):
        self.dt.set_trans_cost(self.tm.scaled_trans_cost(self.transition_scale, self.self_loop_scale))

    def _upload_model(self):
        go, gc, w, miv, iv = self.am.flat()
        if self.accs is not None:
            self.accs.close()
        if self.dm is not None:
            self.dm.close()
        self.dm = DeviceModel(self.ctx, go, gc, miv, iv, weights=w)
        self.accs = DeviceAccs(self.ctx, self.dm, self.dt)

    def sync_host(self) -> AmDiagGmm:
        """Bring am_gmm up to date with the device model (needed before writing or mixing up the model)."""
        if not self.host_in_sync:
            d = self.dm.download()
            self.am.set_flat(d["gauss_off"], d["weights"], d["gconsts"], d["means_invvars"], d["inv_vars"])
            self.host_in_sync = True
        return self.am

    @property
    def num_gauss(self) -> int:
        return int(self.dm.gauss_off[-1])

    # -- the pass ------------------------------------------------------------------------------
    def set_alignments(self, alis: Sequence[Sequence[int]]):
        """E.g. the equal_align start of egs/yesno/train.py:86-108."""
        a = np.concatenate([np.asarray(x, np.int32) for x in alis]) if len(alis) else np.zeros(0, np.int32)
        if a.shape[0] != self.frame_off[-1]:
            raise KhgError("ResidentEm.set_alignments: one transition-id per frame")
        self.us.upload_ali(a)

    def alignments(self) -> List[List[int]]:
        """The resident alignments, downloaded (empty list for an utterance that failed to align)."""
        a = self.us.download_ali()
        out = []
        for u in range(len(self.frame_off) - 1):
            x = a[self.frame_off[u]: self.frame_off[u + 1]]
            out.append(x.tolist() if x.size and x.all() else [])
        return out

    def boost_silence(self, silence_phones: List[int], boost: float = 1.5):
        """scripts/gmm_boost_silence.py:10-45 on the device model (in place, like the script mutates am_gmm)."""
        _, pdfs = get_pdfs_for_phones(self.tm, sorted(silence_phones))
        self.dm.scale_weights(sorted(pdfs), boost)
        self.host_in_sync = False

    def align(self, config: AlignConfig, decoder_opts: FasterDecoderOptions = None) -> Dict[str, float]:
        """gmm_align_compiled over the shard (scripts/gmm_align_compiled.py:10-79 + AlignUtteranceWrapper,
        python/csrc/decoder-wrappers.cc:25-47) -> the counters the reference threads through its calls."""
        if (config.retry_beam != 0 and config.retry_beam <= config.beam) or config.beam <= 0.0:
            raise KhgError(f"Beams do not make sense: beam {config.beam}, retry-beam {config.retry_beam}")
        if config.careful:
            raise KhgError("ResidentEm.align: careful alignment changes the resident graphs; use align_batch")
        o = decoder_opts or FasterDecoderOptions()
        # only the cells a decoder token can read; at a wide beam (few failed beam certificates to repair) also not what only tokens
        # past any accepting path read (khg_loglikes_band: identical alignments)
        self.us.loglikes(self.dm, reachable_only=True, band=config.beam >= 100.0)
        r = self.us.align(self.dt, beam=config.beam, retry_beam=config.retry_beam, acoustic_scale=self.acoustic_scale,
                          max_active=o.max_active, min_active=o.min_active, beam_delta=o.beam_delta, hash_ratio=o.hash_ratio,
                          download="summary")
        ok = (r["status"] & ALIGN_ERROR) == 0
        T = np.diff(self.frame_off)
        return {"num_done": int(ok.sum()), "num_error": int((~ok).sum()),
                "num_retried": int(((r["status"] & ALIGN_RETRIED) != 0).sum()),
                "tot_like": float(sum(float(x) for x in r["like"][ok])), "frame_count": int(T[ok].sum())}

    def accumulate(self, weight: float = 1.0) -> Dict[str, float]:
        """gmm_acc_stats_ali over the shard (scripts/gmm_acc_stats_ali.py:9-58) + the cross-rank sum."""
        self.accs.zero()
        comm = self._rccl_comm()
        if comm is not None and self.sharded_mstep:
            # only the transition counts and scalars are summed here; the Gaussian rows go to their owners inside update()
            self.us.acc_stats(self.dm, self.dt, self.accs, weight)
            self.accs.allreduce_range(self.dm, -1, 0, comm)
        elif comm is not None:        # C1 pipelined behind K3 by pdf ranges, on the library's second stream (khg_acc_stats_reduce)
            self.us.acc_stats_reduce(self.dm, self.dt, self.accs, weight, comm, 4)
        else:
            self.us.acc_stats(self.dm, self.dt, self.accs, weight)
            self._sum_over_ranks()
        self._tr = self.accs.download_trans()
        return {"total_log_like": self._tr["total_log_like"], "total_frames": self._tr["total_frames"]}

    def _rccl_comm(self):
        """The library's communicator when torch.distributed runs more than one rank over nccl (one GPU per rank), else None."""
        try:
            import torch.distributed as dist
        except ImportError:
            return None
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and dist.get_backend() == "nccl"):
            return None
        if not self._comm_made:
            from .dist import make_comm
            self._comm, self._comm_made = make_comm(self.ctx), True
        return self._comm

    def _sum_over_ranks(self):
        """C1 when torch.distributed is initialised with more than one rank: the library's own RCCL all-reduce on the
        context's stream (backend nccl: one GPU per rank), or -- for groups that cannot carry device buffers (gloo, e.g.
        several ranks sharing one GPU in the tests) -- the block summed on the host."""
        try:
            import torch.distributed as dist
        except ImportError:
            return
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        if dist.get_backend() == "nccl":
            if not self._comm_made:
                from .dist import make_comm
                self._comm, self._comm_made = make_comm(self.ctx), True
            self.accs.allreduce(self._comm)
        else:
            import torch
            buf = np.zeros(self.accs.size, np.float64)
            from ._lib import check, lib, ptr
            import ctypes as C
            check(lib.khg_accs_download(self.ctx.h, self.accs.h, ptr(buf, C.c_double)))
            t = torch.from_numpy(buf)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.accs.upload(buf)

    def update(self, tcfg: MleTransitionUpdateConfig = None, gmm_opts: MleDiagGmmOptions = None, mixup: int = 0,
               mixdown: int = 0, perturb_factor: float = 0.01, power: float = 0.2, min_count: float = 20.0, update_flags: str = "mvwt",
               randn=None) -> Dict[str, float]:
        """gmm_est (scripts/gmm_est.py:8-96) from the resident accumulators; returns its printed statistics."""
        flags = str_to_gmm_flags(update_flags)
        gmm_opts = gmm_opts or MleDiagGmmOptions()
        self._updates += 1
        if randn is None:
            # DiagGmm::Split draws from the process-global rand() in the reference (csrc/diag-gmm.cc:823); here every
            # rank must perturb the SAME way (the model is replicated, only statistics are exchanged), so the default
            # stream is a counter-keyed generator: identical on all ranks, reproducible across runs
            _rng = np.random.default_rng([self.split_seed, self._updates])
            randn = lambda d: _rng.standard_normal(d).astype(np.float32)  # noqa: E731
        tr = self._tr
        info: Dict[str, float] = {}
        if int(flags) & int(GmmUpdateFlags.kGmmTransitions):
            objf_impr, count = self.tm.mle_update(tr["trans_acc"], tcfg or MleTransitionUpdateConfig())
            info["transition_objf_impr"], info["transition_count"] = objf_impr, count
            self._set_trans_cost()
        pdf_occs = None
        sharded = self.sharded_mstep and self._rccl_comm() is not None
        go_before = np.asarray(self.dm.gauss_off).copy()
        if (mixup != 0 or mixdown != 0) and not sharded:      # per-pdf occupancies of the statistics, before the update re-lays the block
            occ = self.accs.download_occ()
            pdf_occs = np.asarray([occ[go_before[p]: go_before[p + 1]].sum() for p in range(self.dm.num_pdfs)], np.float32)
        if sharded:
            r = self.dm.mle_update_sharded(self.accs, gmm_opts, int(flags) & 0x7, self._comm)
            if mixup != 0 or mixdown != 0:  # the occupancies were all-reduced inside; the block still has the old layout
                occ = self.accs.download_occ()
                pdf_occs = np.asarray([occ[go_before[p]: go_before[p + 1]].sum() for p in range(len(go_before) - 1)], np.float32)
        else:
            r = self.dm.mle_update(self.accs, gmm_opts, int(flags) & 0x7)
        if r["removed"]:
            self.accs.relayout(self.dm)
        self.host_in_sync = False
        tot_like, tot_t = np.float32(tr["total_log_like"]), np.float32(tr["total_frames"])
        info.update(gmm_objf_impr=r["objf_change"], gmm_count=r["count"], frames=float(tot_t),
                    avg_like=float(tot_like / tot_t) if tot_t else float("nan"), removed=r["removed"])
        if mixdown != 0:
            # AmDiagGmm::MergeByCount (csrc/am-diag-gmm.cc:91-108) on the device model (khg_model_merge), before mixing up
            # like scripts/gmm_est.py:70-84
            targets = np.asarray(get_split_targets(pdf_occs, mixdown, power, min_count), np.int64)
            cur = np.diff(self.dm.gauss_off)
            tgt = np.minimum(np.where(targets == 0, 1, targets), cur).astype(np.int32)      # "can't merge below 1"
            if (tgt < cur).any():
                self.dm.merge(tgt)
                self.accs.relayout(self.dm)
        if mixup != 0:
            # AmDiagGmm::SplitByCount (csrc/am-diag-gmm.cc:72-90) on the device model (khg_model_split): nothing comes down
            # but the occupancies; the normal deviates are drawn here in the order DiagGmm::Split would consume them
            targets = get_split_targets(pdf_occs, mixup, power, min_count)
            cur = np.diff(self.dm.gauss_off)
            tgt = np.maximum(np.asarray(targets, np.int64), cur).astype(np.int32)
            n_new = int((tgt - cur).sum())
            if n_new > 0:
                D = self.dm.dim
                normals = np.concatenate([np.asarray(randn(D), np.float32).reshape(D) for _ in range(n_new)]).reshape(n_new, D)
                self.dm.split(tgt, perturb_factor, normals)
                self.accs.relayout(self.dm)
        return info

    def close(self):
        if self._comm is not None:
            self._comm.close()
            self._comm = None
        for o in (self.accs, self.us, self.dm, self.dt):
            if o is not None:
                o.close()
        self.accs = self.us = self.dm = self.dt = None
