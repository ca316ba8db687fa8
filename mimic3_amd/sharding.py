"""Batch sharding across the GPUs of one node: utterances are independent, so the batch
is split by utterance, every rank runs its rows on its own GPU with a full weight replica, and nothing
is exchanged on the data path.  ``gather_pcm`` is the optional result gather the north star names: a collective
gather of padded int16 ``[B/N, L_max]`` blocks plus their lengths (RCCL on the GPUs — the blocks go HBM -> HBM over
xGMI straight from the engines' result buffers, ``device_pcm_block`` — or gloo on CPU in the tests); throughput runs do
not call it, and when the audio has to reach the host anyway per-GPU D2H is faster.
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


def device_pcm_block(engine):
    """The int16 result of the engine's last run as a torch tensor ``[B, L_max]`` that ALIASES the engine's result buffer
    in HBM (no copy; valid until the engine's next run) plus its valid lengths ``[B]`` (host, int64).  On the CPU model
    of the kernels (tests) the same call gives a CPU tensor."""
    import ctypes

    import torch

    d = engine.device_result()
    B, L = d["batch"], d["row_stride"]
    if "gfx950" in engine.native.version():
        class _Dev:  # what torch.as_tensor needs to wrap foreign device memory
            __cuda_array_interface__ = {"shape": (B, L), "typestr": "<i2", "data": (int(d["pcm"]), False), "version": 3,
                                        "strides": None}

        pcm = torch.as_tensor(_Dev(), device=torch.device("cuda", int(d["device"])))
        lens = torch.empty(B, dtype=torch.int32, device=pcm.device)

        class _DevL:
            __cuda_array_interface__ = {"shape": (B,), "typestr": "<i4", "data": (int(d["lengths"]), False), "version": 3,
                                        "strides": None}

        lens = torch.as_tensor(_DevL(), device=pcm.device).to(torch.int64).cpu()
    else:
        buf = (ctypes.c_int16 * (B * L)).from_address(int(d["pcm"]))
        pcm = torch.from_numpy(np.frombuffer(buf, dtype=np.int16).reshape(B, L))
        lb = (ctypes.c_int32 * B).from_address(int(d["lengths"]))
        lens = torch.from_numpy(np.frombuffer(lb, dtype=np.int32).astype(np.int64))
    return pcm, lens


def gather_pcm(pcm_block, lengths, rows, total: int, group=None, dst: int = 0) -> Optional[List[np.ndarray]]:
    """Optional result gather (north star): every rank contributes its int16 block ``[n_local, L_local]`` (torch tensor —
    on the GPU for RCCL, CPU for gloo — or numpy) with the valid ``lengths`` and the global ``rows`` of its utterances;
    rank ``dst`` gets the ``total`` utterances back in global order as int16 arrays, the others ``None``.

    One small all-reduce agrees on the padded block shape, then ONE ``gather`` moves ``[n_max, L_max]`` int16 per rank
    plus an int64 ``[n_max, 2]`` (row, length) table — tensors, not pickles."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    blk = pcm_block if isinstance(pcm_block, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pcm_block, dtype=np.int16))
    if blk.dim() != 2 or blk.dtype != torch.int16:
        raise ValueError("pcm_block must be int16 [n_local, L_local]")
    dev = blk.device
    n_local, L_local = int(blk.shape[0]), int(blk.shape[1])
    rows = np.asarray(rows, dtype=np.int64).reshape(-1)
    lengths = np.asarray(lengths.cpu() if isinstance(lengths, torch.Tensor) else lengths, dtype=np.int64).reshape(-1)
    if rows.shape[0] != n_local or lengths.shape[0] != n_local:
        raise ValueError("rows / lengths must have one entry per local utterance")
    shape = torch.tensor([n_local, L_local], dtype=torch.int64, device=dev)
    dist.all_reduce(shape, op=dist.ReduceOp.MAX, group=group)
    n_max, L_max = int(shape[0]), int(shape[1])
    padded = torch.zeros((n_max, L_max), dtype=torch.int16, device=dev)
    padded[:n_local, :L_local] = blk
    table = torch.full((n_max, 2), -1, dtype=torch.int64)
    table[:n_local, 0] = torch.from_numpy(rows)
    table[:n_local, 1] = torch.from_numpy(lengths)
    table = table.to(dev)
    wire = padded.view(torch.uint8)  # same bytes; every backend moves uint8 (gloo has no int16 gather)
    if rank == dst:
        blocks = [torch.empty_like(wire) for _ in range(world)]
        tables = [torch.empty_like(table) for _ in range(world)]
    else:
        blocks = tables = None
    dist.gather(wire, blocks, dst=dst, group=group)
    dist.gather(table, tables, dst=dst, group=group)
    if rank != dst:
        return None
    host = torch.stack(blocks).view(torch.int16).cpu().numpy()      # one D2H of [world, n_max, L_max]
    tab = torch.stack(tables).cpu().numpy()
    out: List[Optional[np.ndarray]] = [None] * total
    for r in range(world):
        for i in range(n_max):
            g, n = int(tab[r, i, 0]), int(tab[r, i, 1])
            if g < 0:
                continue
            if g >= total or out[g] is not None or n > L_max:
                raise RuntimeError("gather_pcm: inconsistent row table")
            out[g] = host[r, i, :n].copy()
    if any(a is None for a in out):
        raise RuntimeError("gather_pcm: missing utterances")
    return out  # type: ignore[return-value]


def gather_results(local_audio: List[np.ndarray], rows: np.ndarray, total: int, group=None) -> Optional[List[np.ndarray]]:
    """Gather of per-utterance int16 waveforms (a list of ragged arrays) onto rank 0, restoring global order — packs
    them into one padded block and calls :func:`gather_pcm`."""
    import torch
    import torch.distributed as dist

    n = len(local_audio)
    L = max([len(a) for a in local_audio] + [1])
    blk = np.zeros((n, L), np.int16)
    for i, a in enumerate(local_audio):
        blk[i, : len(a)] = np.asarray(a, dtype=np.int16)
    t = torch.from_numpy(blk)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    return gather_pcm(t, [len(a) for a in local_audio], rows, total, group)
