"""Multi-GPU: utterances shard across ranks -- they are independent in align and acc-stats --
the model is replicated, and the only exchange per EM iteration is ONE sum all-reduce of the fp64 accumulator
block: the device-side form of AccumAmDiagGmm::Add (csrc/mle-am-diag-gmm.cc:119-128, Kaldi's gmm-sum-accs).

One process per GPU.  The data-path collective is the library's own khg_accs_allreduce (RCCL called from the
C-ABI on the context's stream, include/khg_hip.h "C1"); torch.distributed is only the rendezvous that ships the
128-byte communicator id (and, in bench.py, the barrier / max-over-ranks timing)."""
from typing import Dict, List, Sequence, Tuple

import numpy as np


def shard_utterances(num_frames: Sequence[int], world_size: int) -> List[np.ndarray]:
    """Deal utterances to ranks balancing total frames: longest first onto the lightest rank
    (SURVEY.md 8e).  Returns, per rank, the sorted utterance indices it owns."""
    num_frames = np.asarray(num_frames, np.int64)
    order = np.argsort(-num_frames, kind="stable")
    load = np.zeros(world_size, np.int64)
    owner = np.zeros(num_frames.shape[0], np.int64)
    for u in order:
        r = int(np.argmin(load))
        owner[u] = r
        load[r] += num_frames[u]
    return [np.nonzero(owner == r)[0] for r in range(world_size)]


def take_utterances(frame_off, graphs: Dict[str, np.ndarray], idx) -> Tuple[np.ndarray, Dict[str, np.ndarray], np.ndarray]:
    """The sub-set `idx` (increasing utterance indices) of a concatenated utterance set: -> (frame_off of the
    sub-set, its graphs in the same CSR-by-source-state layout UtteranceSet takes, the global frame index of each
    of its frames).  Pure index arithmetic, no per-utterance Python loop (bench-sized sets)."""
    frame_off = np.asarray(frame_off, np.int64)
    idx = np.asarray(idx, np.int64)
    T = frame_off[idx + 1] - frame_off[idx]
    fo = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)

    def ranges(starts, lens):
        """concatenation of arange(starts[i], starts[i] + lens[i])"""
        tot = int(lens.sum())
        if tot == 0:
            return np.zeros(0, np.int64)
        off = np.concatenate([[0], np.cumsum(lens)[:-1]])
        return np.repeat(starts - off, lens) + np.arange(tot, dtype=np.int64)

    frames = ranges(frame_off[idx], T)
    if graphs is None:
        return fo, None, frames
    so = np.asarray(graphs["state_off"], np.int64)
    ao = np.asarray(graphs["arc_off"], np.int64)
    S = so[idx + 1] - so[idx]
    states = ranges(so[idx], S)
    narc = ao[states + 1] - ao[states]
    arcs = ranges(ao[states], narc)
    g = {"state_off": np.concatenate([[0], np.cumsum(S)]).astype(np.int64),
         "start": np.asarray(graphs["start"])[idx],
         "arc_off": np.concatenate([[0], np.cumsum(narc)]).astype(np.int64),
         "final": np.asarray(graphs["final"])[states]}
    for k in ("ilabel", "olabel", "weight", "nextstate"):
        g[k] = np.asarray(graphs[k])[arcs]
    return fo, g, frames


def make_comm(ctx):
    """The library's RCCL communicator over the ranks of the initialised torch.distributed group (any backend:
    the group only carries the 128-byte id from rank 0).  None for a one-rank job."""
    import torch.distributed as dist

    from .device import Comm

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    box = [Comm.unique_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return Comm(ctx, dist.get_world_size(), dist.get_rank(), box[0])


def allreduce_accs(acc_tensor):
    """torch.distributed form of the same sum (a torch view of the block, DeviceAccs.as_torch(), or a CPU tensor
    under gloo): in place.  The product path uses DeviceAccs.allreduce(comm) instead."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc_tensor, op=dist.ReduceOp.SUM)
    return acc_tensor
