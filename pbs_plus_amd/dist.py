"""Multi-GPU layer: one process per GPU, segments sharded across ranks, and ONE exchange
step — the all-gather of fixed-width (digest, size) records for cross-file duplicate
detection. Chunking and hashing need no collective: segments are
independent streams (fresh chunker state, forced end cut), so the path shards at segment
granularity with no data-path communication.

``torch.distributed`` is plumbing here (RCCL over xGMI with backend "nccl" on the GPU box,
gloo in the CPU tests); the records are ~48 B per ~4 MiB chunk (≈12 MB per TiB), so the
exchange is latency-bound and a single all_gather of padded record arrays is enough.

The same reduce exists BEHIND THE C ABI (pbsgpu_comm_* / pbsgpu_digest_allgather_dedup, csrc/comm.cpp: libpbsgpu
resolves RCCL itself) for hosts without Python — ``make_comm`` / ``comm_dedup`` drive it from here; the torch path
stays because gloo (CPU tests, world size > 1 without GPUs) cannot run RCCL.
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


def _gather_rows(records: np.ndarray, dev, group, cap_records):
    """ONE all_gather_into_tensor of [count | padded 48-byte records] rows; returns (rows tensor on `dev`, row bytes,
    world size, capacity). The rows tensor stays on the device with RCCL."""
    import torch
    import torch.distributed as dist

    ws = dist.get_world_size(group)
    recs = np.ascontiguousarray(records, dtype=RECORD_DTYPE)
    isz = RECORD_DTYPE.itemsize
    if cap_records is None:
        cnt = torch.tensor([recs.size], dtype=torch.int64, device=dev)
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(ws)]
        dist.all_gather(counts, cnt, group=group)
        cap_records = max(max(int(c.item()) for c in counts), 1)
    if recs.size > cap_records:
        raise ValueError(f"{recs.size} records exceed the agreed capacity {cap_records}")
    row = 16 + cap_records * isz                       # 16-byte header keeps the records 16-byte aligned
    # only the bytes that exist are uploaded: the row is assembled on the device (the padding is never copied H2D)
    mine = torch.zeros(row, dtype=torch.uint8, device=dev)
    head = np.zeros(16, dtype=np.uint8)
    head[:8] = np.frombuffer(np.uint64(recs.size).tobytes(), dtype=np.uint8)
    mine[:16] = torch.from_numpy(head).to(dev)
    if recs.size:
        mine[16:16 + recs.size * isz] = torch.from_numpy(recs.view(np.uint8).reshape(-1)).to(dev)
    out = torch.empty(ws * row, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    return out.view(ws, row), row, ws, cap_records


def allgather_records(records: np.ndarray, device=None, group=None, cap_records: int | None = None) -> np.ndarray:
    """All-gather every rank's record array (variable length) -> concatenation in rank order, on the HOST.

    With `cap_records` (an upper bound every rank agrees on, e.g. bytes / min chunk size) this is ONE collective:
    an all_gather_into_tensor of [count | padded 48-byte records]; without it the counts are exchanged first.
    `device` = torch device of the communication buffers ("cuda:N" with RCCL, None/"cpu" with gloo)."""
    import torch

    dev = torch.device(device) if device is not None else torch.device("cpu")
    rows, row, ws, _ = _gather_rows(records, dev, group, cap_records)
    host = rows.cpu().numpy()
    isz = RECORD_DTYPE.itemsize
    parts = []
    for r in range(ws):
        n = int(host[r, :8].view(np.uint64)[0])
        parts.append(host[r, 16:16 + n * isz].view(RECORD_DTYPE).copy())
    return np.concatenate(parts) if parts else np.zeros(0, dtype=RECORD_DTYPE)


def global_dedup(engine, local_records: np.ndarray, device=None, group=None, cap_records: int | None = None,
                 want_records: bool = True):
    """Digest-set reduce: gather all ranks' records (one collective), then duplicate detection on this rank's GPU
    (radix sort by digest prefix + compare, libpbsgpu). Returns (dup flags, stats, all records or None).

    With RCCL (`device` = a cuda device) the gathered set never leaves the device: the rows of the all-gather are
    compacted with one device gather and handed to pbsgpu_dedup_device; only the per-rank counts (8 bytes each), the
    flags and the four statistics come back to the host. `want_records=False` skips the download of the gathered
    records (the benchmark only needs the statistics).
    Every rank runs the (sub-millisecond) dedup over the whole set: with ~10 MB per TiB of corpus the exchange is
    latency-bound, and a rank-0-only dedup would pay a second collective to broadcast four numbers."""
    import torch

    dev = torch.device(device) if device is not None else torch.device("cpu")
    if dev.type != "cuda" or not hasattr(engine, "dedup_device"):
        allrecs = allgather_records(local_records, device=device, group=group, cap_records=cap_records)
        dup, stats = engine.dedup(allrecs)
        return dup, stats, allrecs
    rows, row, ws, _ = _gather_rows(local_records, dev, group, cap_records)
    isz = RECORD_DTYPE.itemsize
    counts = rows[:, :8].contiguous().cpu().numpy().view(np.uint64).reshape(-1)      # ws x 8 bytes
    total = int(counts.sum())
    if total == 0:
        dup, stats = engine.dedup(np.zeros(0, dtype=RECORD_DTYPE))
        return dup, stats, (np.zeros(0, dtype=RECORD_DTYPE) if want_records else None)
    packed = torch.cat([rows[r, 16:16 + int(counts[r]) * isz] for r in range(ws) if counts[r]])
    # the engine's own HIP stream reads the tensor next: wait for THIS stream only. (A device-wide synchronize would
    # never return while a page ring's persistent SHA-256 service is running.)
    torch.cuda.current_stream(dev).synchronize()
    dup, stats = engine.dedup_device(packed.data_ptr(), total)
    allrecs = packed.cpu().numpy().view(RECORD_DTYPE).copy() if want_records else None
    return dup, stats, allrecs


def make_comm(engine, device=None, group=None):
    """The C ABI's own communicator for the digest-set reduce (pbsgpu_comm_*, engine.Comm): libpbsgpu resolves RCCL itself
    and moves the records with ONE ncclAllGather — the path a Go host binds, with no torch in it. torch.distributed only
    carries the 128-byte id from rank 0 to the others here (a Go host ships it over its own RPC). Collective."""
    import torch
    import torch.distributed as dist

    from .engine import Comm

    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    idt = torch.zeros(Comm.ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt = torch.frombuffer(bytearray(Comm.unique_id()), dtype=torch.uint8).to(dev)
    dist.broadcast(idt, src=0, group=group)
    return Comm(engine, bytes(idt.cpu().numpy().tobytes()), rank, ws)


def comm_dedup(comm, local_records: np.ndarray, cap_records: int, want_flags: bool = False):
    """global_dedup through the C ABI: (dup flags of THIS rank's records or None, stats of the union)."""
    return comm.dedup(local_records, cap_records, want_flags=want_flags)


def ingest_corpus(engine, lengths, make_batch, device=None, group=None, max_batch_bytes: int = 64 << 30):
    """End-to-end corpus ingest over all ranks: shard the segments (files) over the ranks, cut + hash
    this rank's share batch by batch on its GPU, then ONE digest-set reduce (all-gather of the 48-byte records) and
    duplicate detection. No data-path collective: segments are independent streams.

    lengths      byte length of every segment of the corpus (same on every rank)
    make_batch   callback(list of global segment indices) -> (device buffer, [(offset, length)] in that buffer);
                 the caller owns how bytes reach HBM (generated, read from disk, received from agents). Two batches
                 are in flight: make_batch must NOT reuse the buffer of the previous call (it is still being read)
    Returns (records of this rank with GLOBAL segment ids, dedup stats over the whole corpus, all ranks' records)."""
    import torch.distributed as dist

    ws = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lengths = np.asarray(lengths, dtype=np.uint64)
    plan = shard_segments(lengths, ws)
    mine = plan[rank]
    # batches of <= max_batch_bytes, in segment order
    batches, cur, cur_bytes = [], [], 0
    for g in mine:
        n = int(lengths[g])
        if cur and cur_bytes + n > max_batch_bytes:
            batches.append(cur)
            cur, cur_bytes = [], 0
        cur.append(g)
        cur_bytes += n
    if cur:
        batches.append(cur)
    parts, pending = [], []
    for ids in batches:                      # two batches in flight: the next is generated while one hashes
        buf, segs = make_batch(ids)
        # a raw device pointer needs the byte extent of its segments (they need not start at 0 nor be gap-free)
        nbytes = None if not isinstance(buf, int) else max((int(o) + int(n) for o, n in segs), default=0)
        # `buf` stays referenced until its ticket has been collected: device pointers are borrowed until then (a torch
        # tensor dropped here could be handed to the next batch by the caching allocator while this one is still read)
        pending.append((ids, engine.submit(buf, segs, nbytes), buf))
        if len(pending) == 2:
            ids0, t0, _keep = pending.pop(0)
            parts.append(_collect_global(engine, ids0, t0))
    while pending:
        ids0, t0, _keep = pending.pop(0)
        parts.append(_collect_global(engine, ids0, t0))
    local = np.concatenate(parts) if parts else np.zeros(0, dtype=RECORD_DTYPE)
    if ws > 1:
        cap = max(int(sum(int(lengths[g]) for g in p) // max(1, engine.config.MinSize)) + 2 * len(p) + 16 for p in plan)
        dup, stats, allrecs = global_dedup(engine, local, device=device, group=group, cap_records=cap)
    else:
        dup, stats = engine.dedup(local)
        allrecs = local
    return local, stats, allrecs


def _collect_global(engine, ids, ticket):
    recs = engine.collect(ticket)
    recs["segment"] = np.asarray(ids, dtype=np.uint32)[recs["segment"]]   # batch-local -> global segment id
    return recs


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
