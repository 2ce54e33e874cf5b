"""Multi-GPU layer: one process per GPU, segments sharded across ranks, and ONE exchange
step — the all-gather of fixed-width (digest, size) records for cross-file duplicate
detection. Chunking and hashing need no collective: segments are
independent streams (fresh chunker state, forced end cut), so the path shards at segment
granularity with no data-path communication.

``torch.distributed`` is plumbing here (RCCL over xGMI with backend "nccl" on the GPU box,
gloo in the CPU tests); the records are ~48 B per ~4 MiB chunk (≈12 MB per TiB), so the
exchange is latency-bound and a single all_gather of padded record arrays is enough.
"""
from __future__ import annotations

import numpy as np

from ._lib import RECORD_DTYPE


def shard_segments(lengths, world_size: int):
    """Greedy longest-first assignment of segment indices to ranks (balanced bytes).

    Returns a list (per rank) of ascending segment indices. Deterministic on every rank."""
    lengths = np.asarray(lengths, dtype=np.uint64)
    order = np.argsort(-lengths.astype(np.int64), kind="stable")
    load = np.zeros(world_size, dtype=np.uint64)
    out = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        out[r].append(int(i))
        load[r] += lengths[i]
    return [sorted(x) for x in out]


def allgather_records(records: np.ndarray, device=None, group=None) -> np.ndarray:
    """All-gather every rank's record array (variable length) -> concatenation in rank order.

    One all_gather of the counts and one of the padded payload (uint8 view of the 48-byte
    records). `device` = torch device for the communication buffers ("cuda:N" with RCCL,
    None/"cpu" with gloo)."""
    import torch
    import torch.distributed as dist

    ws = dist.get_world_size(group)
    recs = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    cnt = torch.tensor([recs.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    payload = np.zeros(cap * RECORD_DTYPE.itemsize, dtype=np.uint8)
    payload[: recs.size * RECORD_DTYPE.itemsize] = recs.view(np.uint8).reshape(-1)
    mine = torch.from_numpy(payload).to(dev)
    bufs = [torch.empty_like(mine) for _ in range(ws)]
    dist.all_gather(bufs, mine, group=group)
    parts = []
    for r in range(ws):
        a = bufs[r].cpu().numpy()[: counts[r] * RECORD_DTYPE.itemsize]
        parts.append(a.view(RECORD_DTYPE).copy())
    return np.concatenate(parts) if parts else np.zeros(0, dtype=RECORD_DTYPE)


def global_dedup(engine, local_records: np.ndarray, device=None, group=None):
    """Digest-set reduce: gather all ranks' records, then duplicate detection on this rank's
    GPU (sort by digest prefix + compare, libpbsgpu). Returns (dup flags, stats, all records)."""
    allrecs = allgather_records(local_records, device=device, group=group)
    dup, stats = engine.dedup(allrecs)
    return dup, stats, allrecs
