"""Multi-GPU: utterances shard across ranks (they are independent in align and acc-stats); the
only exchange per EM iteration is ONE sum all-reduce of the fp64 accumulator block -- the
device-side form of AccumAmDiagGmm::Add (csrc/mle-am-diag-gmm.cc:119-128, Kaldi's gmm-sum-accs).
backend "nccl" is RCCL over xGMI on the GPU box; "gloo" runs the same code on CPU in the tests."""
from typing import List, Sequence

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


def allreduce_accs(acc_tensor):
    """Sum the accumulator block over all ranks, in place (torch tensor on the rank's device)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc_tensor, op=dist.ReduceOp.SUM)
    return acc_tensor
