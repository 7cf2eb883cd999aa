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


def make_comm(ctx, one_rank=False, timeout_s=None):
    """The library's RCCL communicator over the ranks of the initialised torch.distributed group (any backend:
    the group only carries the 128-byte id from rank 0).  None for a one-rank job, unless ``one_rank`` asks for a
    one-rank communicator (the collective code path on a one-GPU box).  ``timeout_s``: communicator formation is a collective
    that blocks while a rank is missing; past the timeout a TimeoutError is raised in THIS process (the caller decides how to
    leave: a process that has touched the GPU must exit, not re-exec)."""
    import torch.distributed as dist

    from .device import Comm

    up = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size() if up else 1
    if world == 1 and not one_rank:
        return None
    rank = dist.get_rank() if up else 0
    box = [Comm.unique_id() if rank == 0 else None]
    if up and world > 1:
        dist.broadcast_object_list(box, src=0)
    if timeout_s is None:
        return Comm(ctx, world, rank, box[0])
    import threading
    out = {}

    def form():
        try:
            out["comm"] = Comm(ctx, world, rank, box[0])      # releases the GIL while ncclCommInitRank blocks
        except BaseException as ex:                           # noqa: B036  (handed to the caller's thread)
            out["error"] = ex
    th = threading.Thread(target=form, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        raise TimeoutError(f"rank {rank}: ncclCommInitRank over {world} ranks did not return within {timeout_s:g} s")
    if "error" in out:
        raise out["error"]
    return out["comm"]


def allreduce_accs(acc_tensor):
    """torch.distributed form of the same sum (a torch view of the block, DeviceAccs.as_torch(), or a CPU tensor
    under gloo): in place.  The product path uses DeviceAccs.allreduce(comm) instead."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc_tensor, op=dist.ReduceOp.SUM)
    return acc_tensor
