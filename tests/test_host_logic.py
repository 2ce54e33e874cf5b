"""Host-side logic that runs without a GPU: segment sharding for the multi-GPU path and the
world_size-2 digest-set exchange over gloo (the RCCL path on the GPU box uses the same code
with backend nccl)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_segments_balanced_and_complete():
    from pbs_plus_amd.dist import shard_segments

    rng = np.random.default_rng(3)
    lens = rng.integers(1, 1 << 26, 1000)
    for ws in (1, 2, 4, 8):
        shards = shard_segments(lens, ws)
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(1000))
        loads = [int(lens[s].sum()) for s in shards]
        assert max(loads) - min(loads) <= int(lens.max())
        assert all(s == sorted(s) for s in shards)
    assert shard_segments([5, 5], 4) == [[0], [1], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from pbs_plus_amd import RECORD_DTYPE
    from pbs_plus_amd.dist import allgather_records, shard_segments

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        lens = [1000 + 37 * i for i in range(11)]
        mine = shard_segments(lens, ws)[rank]
        recs = np.zeros(len(mine) + rank, dtype=RECORD_DTYPE)  # ragged on purpose (rank 0 may differ)
        for j, seg in enumerate(mine):
            recs[j]["segment"], recs[j]["end"], recs[j]["size"] = seg, lens[seg], lens[seg]
            recs[j]["digest"][:] = seg % 5  # duplicates across ranks
        for j in range(len(mine), recs.size):
            recs[j]["segment"], recs[j]["size"] = 1000 + rank, 1
            recs[j]["digest"][:] = 200 + rank
        allr = allgather_records(recs)
        q.put((rank, allr.tobytes(), recs.tobytes()))
    finally:
        dist.destroy_process_group()


def test_allgather_records_world_size_2_gloo():
    import torch.multiprocessing as mp

    from pbs_plus_amd import RECORD_DTYPE

    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(ws))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all0 = np.frombuffer(got[0][1], dtype=RECORD_DTYPE)
    all1 = np.frombuffer(got[1][1], dtype=RECORD_DTYPE)
    assert all0.tobytes() == all1.tobytes()  # every rank sees the same global set
    local = [np.frombuffer(g[2], dtype=RECORD_DTYPE) for g in got]
    assert all0.tobytes() == np.concatenate(local).tobytes()  # rank order, nothing lost or padded in
    segs = sorted(int(s) for s in all0["segment"] if s < 1000)
    assert segs == list(range(11))
    # duplicate structure the device dedup must find: digests seg%5 -> 5 unique among the 11
    uniq = {bytes(d) for d in all0["digest"]}
    assert len(uniq) == 5 + 1  # + rank 1's filler record


def test_split_plan_covers_stream_with_halo_and_overlap():
    from pbs_plus_amd.dist import split_plan

    T, mx = 1_000_003, 16384
    for ws in (1, 2, 3, 8):
        plan = split_plan(T, ws, mx)
        assert plan[0][0] == 0 and plan[-1][1] == T
        for (a, b, lo, hi), nxt in zip(plan, plan[1:] + [None]):
            assert lo == max(0, a - 63) and hi == min(T, b + mx) and a <= b
            if nxt is not None:
                assert nxt[0] == b


class _OracleEngine:
    """Engine stand-in for the CPU tests of the multi-rank driver: same submit/collect/dedup surface, answers with
    the oracle (test infrastructure; the product path never does this)."""

    def __init__(self, avg):
        from oracle import oracle as O
        from pbs_plus_amd import buzhash

        self.O, self.cfg, self.config = O, O.new_config(avg), buzhash.NewConfig(avg)
        self.t, self.tickets = 0, {}

    def submit(self, buf, segments=None, nbytes=None):
        self.t += 1
        self.tickets[self.t] = self.O.chunk_and_digest(self.cfg, buf, [(int(o), int(n)) for o, n in segments])
        return self.t

    def collect(self, t):
        return self.tickets.pop(t)

    def dedup(self, records):
        seen, dup, ub = set(), np.zeros(records.size, dtype=np.uint8), 0
        for i, d in enumerate(records["digest"]):
            k = d.tobytes()
            dup[i] = k in seen
            if k not in seen:
                ub += int(records["size"][i])
            seen.add(k)
        return dup, {"nrecords": int(records.size), "nunique": len(seen), "total_bytes": int(records["size"].sum()),
                     "unique_bytes": ub}


def _corpus(nseg=24, seg=1 << 20, dup_pct=40, seed=11):
    """content id per segment: dup_pct % copy an earlier segment (deterministic)"""
    rng = np.random.default_rng(seed)
    root = np.arange(nseg)
    for i in range(1, nseg):
        if rng.integers(0, 100) < dup_pct:
            root[i] = root[int(rng.integers(0, i))]
    return root, [seg + 4096 * (i % 3) for i in range(nseg)]


def _ingest_worker(rank, ws, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from pbs_plus_amd.dist import ingest_corpus

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        eng = _OracleEngine(65536)
        root, lens = _corpus()

        def make_batch(ids):     # "bytes reach HBM": here a host array, segments back to back
            parts, segs, off = [], [], 0
            for g in ids:
                parts.append(eng.O.fill(lens[g], 1000 + int(root[g]), 0)[: lens[g]])
                segs.append((off, lens[g]))
                off += lens[g]
            return np.concatenate(parts), segs

        local, stats, allrecs = ingest_corpus(eng, lens, make_batch, max_batch_bytes=5 << 20)
        q.put((rank, local.tobytes(), {k: int(v) for k, v in stats.items()}, allrecs.tobytes()))
    finally:
        dist.destroy_process_group()


def test_ingest_corpus_two_ranks_finds_the_planted_duplicates():
    """dist.ingest_corpus end to end over gloo, world size 2: shard -> per-rank batches (several per rank) -> ONE
    all-gather of the records -> dedup. The duplicate bytes found equal the bytes of the planted duplicate segments
    (whole segments are copied and cuts are content-defined, so the match is exact, not statistical)."""
    import torch.multiprocessing as mp

    from pbs_plus_amd import RECORD_DTYPE

    ws, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ingest_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(ws))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root, lens = _corpus()
    assert got[0][2] == got[1][2] and got[0][3] == got[1][3]          # every rank: same stats, same global record set
    stats = got[0][2]
    locals_ = [np.frombuffer(g[1], dtype=RECORD_DTYPE) for g in got]
    allr = np.frombuffer(got[0][3], dtype=RECORD_DTYPE)
    assert allr.size == sum(x.size for x in locals_)
    assert sorted(set(int(s) for s in allr["segment"])) == list(range(len(lens)))   # GLOBAL segment ids, all present
    assert set(int(s) for s in locals_[0]["segment"]).isdisjoint(int(s) for s in locals_[1]["segment"])
    first = {}
    dup_bytes = 0
    for g in range(len(lens)):
        if int(root[g]) in first and lens[first[int(root[g])]] == lens[g]:
            dup_bytes += lens[g]
        else:
            first.setdefault(int(root[g]), g)
    assert stats["total_bytes"] == sum(lens)
    # segments with equal content id but different length share a prefix: their common leading chunks dedup as well
    assert stats["total_bytes"] - stats["unique_bytes"] >= dup_bytes > 0
    frac = 1 - stats["unique_bytes"] / stats["total_bytes"]
    assert 0.2 < frac < 0.6


def _ingest_worker_uneven(rank, ws, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from pbs_plus_amd.dist import ingest_corpus

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        eng = _OracleEngine(65536)
        # three segments over four ranks: one rank gets nothing (an empty shard must not break the collective), the
        # others get very different byte counts
        lens = [3 << 20, (1 << 20) + 77, 70000]
        root = [0, 1, 1]
        calls = []

        def make_batch(ids):
            calls.append(list(ids))
            parts, segs, off = [], [], 0
            for g in ids:
                parts.append(eng.O.fill(lens[g], 500 + root[g], 0)[: lens[g]])
                segs.append((off, lens[g]))
                off += lens[g]
            return np.concatenate(parts), segs

        local, stats, allrecs = ingest_corpus(eng, lens, make_batch, max_batch_bytes=2 << 20)
        q.put((rank, int(local.size), {k: int(v) for k, v in stats.items()}, allrecs.tobytes(), calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ws", [4, 8])
def test_ingest_corpus_more_ranks_than_segments(ws):
    """world size 4 and 8 over gloo with 3 segments: empty ranks, uneven shards, one segment larger than the batch limit.
    Every rank ends with the same global record set and statistics; segment 2's content is a prefix of segment 1's."""
    import torch.multiprocessing as mp

    from pbs_plus_amd import RECORD_DTYPE

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ingest_worker_uneven, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(ws))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(1 for g in got if g[1] == 0) == ws - 3          # the ranks without a segment hold no records
    assert all(g[2] == got[0][2] and g[3] == got[0][3] for g in got)
    allr = np.frombuffer(got[0][3], dtype=RECORD_DTYPE)
    assert sorted(set(int(x) for x in allr["segment"])) == [0, 1, 2]
    st = got[0][2]
    assert st["total_bytes"] == (3 << 20) + (1 << 20) + 77 + 70000 and st["nrecords"] == allr.size
    assert st["total_bytes"] - st["unique_bytes"] > 0          # the shared prefix of segments 1 and 2 dedups


def test_split_plan_behind_the_c_abi_equals_the_python_plan():
    """pbsgpu_split_plan (round 6: the single-stream split lives behind the C ABI, a Go host needs no Python for BASELINE
    configs[1] at N > 1) against dist.split_plan, the arithmetic the gloo tests of rounds 2-5 exercised: owned ranges tile the
    stream, held ranges reach 63 bytes to the left and one maximum chunk to the right."""
    from pbs_plus_amd import _lib
    from pbs_plus_amd.dist import split_plan as py_plan
    from pbs_plus_amd.engine import split_plan

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    for total in (0, 1, 63, 64, 1000, (64 << 30) + 12345, 1 << 40):
        for world in (1, 2, 3, 8):
            want = py_plan(total, world, 16 << 20)
            got = [split_plan(total, world, r, 16 << 20) for r in range(world)]
            assert got == [tuple(int(x) for x in w) for w in want], (total, world)
            assert got[0][0] == 0 and got[-1][1] == total and all(got[r][1] == got[r + 1][0] for r in range(world - 1))
