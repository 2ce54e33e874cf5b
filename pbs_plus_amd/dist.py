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


# ---- one stream split over several GPUs -----------------------------------------------------------
def split_plan(total_len: int, world_size: int, max_chunk: int):
    """Byte range each rank must hold for a stream split evenly over the ranks.

    Rank r OWNS [r*S, min((r+1)*S, T)) and must have the bytes [lo, hi) resident, where lo reaches 63 bytes
    to the left (window halo for the candidates it reports) and hi reaches max_chunk to the right (so that
    every chunk STARTING in its range is fully local). Returns [(own_start, own_end, lo, hi)]."""
    S = -(-total_len // world_size)
    S = (S + 7) & ~7
    plan = []
    for r in range(world_size):
        a, b = min(r * S, total_len), min((r + 1) * S, total_len)
        plan.append((a, b, max(0, a - 63), min(total_len, b + max_chunk)))
    return plan


def _allgather_var(arr: np.ndarray, device=None, group=None) -> list:
    """all_gather of variable-length 1-D uint8-viewable arrays -> list (per rank) of numpy arrays."""
    import torch
    import torch.distributed as dist

    ws = dist.get_world_size(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    cnt = torch.tensor([raw.size], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    pad = np.zeros(cap, dtype=np.uint8)
    pad[: raw.size] = raw
    mine = torch.from_numpy(pad).to(dev)
    bufs = [torch.empty_like(mine) for _ in range(ws)]
    dist.all_gather(bufs, mine, group=group)
    return [bufs[r].cpu().numpy()[: counts[r]].copy() for r in range(ws)]


def split_stream_chunk_and_digest(engine, local, total_len: int, device=None, group=None) -> np.ndarray:
    """Cut + hash ONE stream of `total_len` bytes that is split over the ranks (second row).

    `local` = this rank's device-resident bytes [lo, hi) as given by split_plan. Data path: every rank scans
    its own bytes; exchange steps: (1) all-gather of the candidate END offsets (a few per MiB), (2) all-gather
    of the digests. The cut chain is resolved identically on every rank from the gathered list (the resolve
    kernel), then each rank hashes the chunks that START in its range. Returns the full record list."""
    import torch.distributed as dist

    rank, ws = dist.get_rank(group), dist.get_world_size(group)
    cfg = engine.config
    own_a, own_b, lo, hi = split_plan(total_len, ws, cfg.MaxSize)[rank]
    ptr, nbytes = engine._dev(local)
    assert nbytes >= hi - lo, "local buffer shorter than split_plan asks for"
    if hi > lo:
        ends = engine.candidates(ptr, hi - lo) + np.uint64(lo)       # stream coordinates
        ends = ends[(ends > own_a) & (ends <= own_b)]                 # report each candidate exactly once
    else:
        ends = np.zeros(0, dtype=np.uint64)
    parts = _allgather_var(ends, device=device, group=group)
    allc = np.concatenate([p.view(np.uint64) for p in parts]) if parts else ends
    recs = engine.resolve_candidates(allc, total_len)
    starts = recs["end"] - recs["size"].astype(np.uint64)
    mine = np.flatnonzero((starts >= own_a) & (starts < own_b))
    digs = np.zeros((0, 32), dtype=np.uint8)
    if mine.size:
        segs = np.stack([starts[mine] - np.uint64(lo), recs["size"][mine].astype(np.uint64)], axis=1)
        digs = engine.sha256_many(ptr, segs, nbytes=hi - lo)
    gathered = _allgather_var(digs, device=device, group=group)
    alld = np.concatenate([g.reshape(-1, 32) for g in gathered]) if gathered else digs
    assert alld.shape[0] == recs.size, (alld.shape, recs.size)
    recs["digest"] = alld  # ranks own contiguous, ascending runs of chunks -> rank order == stream order
    return recs
