"""Batch sharding across the GPUs of one node: utterances are independent, so the batch
is split by utterance, every rank runs its rows on its own GPU with a full weight replica, and nothing
is exchanged on the data path.  ``gather_results`` is the optional result gather the north star names
(``torch.distributed`` all_gather_object over RCCL/gloo); throughput runs do not call it.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split of ``n`` utterances: the first ``n % world_size`` ranks get one extra."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def balanced_order(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time assignment of utterances to ranks by phoneme count (cost ~ frames ~ Tx)."""
    order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))
    loads = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += int(lengths[i])
    for r in range(world_size):
        out[r].sort()
    return out


def shard_feed(feed: Dict[str, np.ndarray], world_size: int, rank: int, balance: bool = False
               ) -> Tuple[Dict[str, np.ndarray], np.ndarray]:
    """Rows of an ``onnx_model.run`` feed this rank owns, plus their global row indices."""
    ids = np.asarray(feed["input"])
    B = ids.shape[0]
    if balance:
        rows = np.asarray(balanced_order(np.asarray(feed["input_lengths"]).tolist(), world_size)[rank], dtype=np.int64)
    else:
        lo, hi = shard_bounds(B, world_size, rank)
        rows = np.arange(lo, hi, dtype=np.int64)
    out = {"input": ids[rows], "input_lengths": np.asarray(feed["input_lengths"])[rows], "scales": feed["scales"]}
    if "sid" in feed:
        out["sid"] = np.asarray(feed["sid"])[rows]
    return out, rows


def gather_results(local_audio: List[np.ndarray], rows: np.ndarray, total: int, group=None) -> Optional[List[np.ndarray]]:
    """Optional gather of per-utterance waveforms onto rank 0, restoring global order."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    payload = (rows.tolist(), local_audio)
    gathered: List = [None] * world if dist.get_rank(group) == 0 else None
    dist.gather_object(payload, gathered, dst=0, group=group)
    if dist.get_rank(group) != 0:
        return None
    out: List[Optional[np.ndarray]] = [None] * total
    for idx, auds in gathered:
        for i, a in zip(idx, auds):
            out[i] = a
    if any(a is None for a in out):
        raise RuntimeError("gather_results: missing utterances")
    return out  # type: ignore[return-value]
