"""bench.py's output contract, exercised on CPU: the engine is replaced by a stand-in that answers with the
oracle's records (tests may use the oracle; the product path never does), torch.cuda calls are stubbed, and the
one JSON line is checked for every field the driver and the judge read. This is a test of bench.py's host
logic (workload construction, slot scheduling, distinct data per slot, JSON assembly), not of the kernels
(those are the -m gpu tests)."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeBuf:
    _next = 1 << 20

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = _FakeBuf._next            # fake, 4 KiB aligned "device address"
        _FakeBuf._next += (self.nbytes + 8191) & ~4095
        self.host = np.zeros(self.nbytes, dtype=np.uint8)   # the stand-in's "HBM"
        self.freed = False

    def free(self):
        self.freed = True

    def download(self, offset=0, nbytes=None):
        n = self.nbytes - offset if nbytes is None else nbytes
        return self.host[offset:offset + n].copy()


class _FakeEngine:
    """Same surface as pbs_plus_amd.Engine as far as bench.py uses it; bytes live in host arrays."""

    def __init__(self, config, device=0, inflight=2):
        from oracle import oracle as O

        self.O, self.avg, self.inflight = O, config.AvgSize, inflight
        self.tickets, self.next, self.bufs = {}, 1, []
        self.in_flight_bufs = set()

    def alloc(self, nbytes):
        b = _FakeBuf(nbytes)
        self.bufs.append(b)
        return b

    def _find(self, ptr):
        for b in self.bufs:
            if b.ptr <= ptr < b.ptr + max(b.nbytes, 1):
                return b, ptr - b.ptr
        raise AssertionError("fill/submit outside any allocation")

    def fill(self, dptr, nbytes, seed, kind=0, stream_off=0):
        b, off = self._find(dptr)
        assert off + nbytes <= b.nbytes
        self.O.fill(nbytes, seed, kind, stream_off, out=b.host[off:off + nbytes])

    def gather(self, src, dst, items):
        for so, do, ln in np.asarray(items, dtype=np.uint64).reshape(-1, 3):
            dst.host[int(do):int(do + ln)] = src.host[int(so):int(so + ln)]

    def _cut(self, buf, segments, nbytes):
        segs = [(0, nbytes)] if segments is None else [(int(o), int(n)) for o, n in segments]
        return self.O.chunk_and_digest(self.O.new_config(self.avg), buf.host[:nbytes], segs)

    def chunk_and_digest(self, buf, segments=None, nbytes=None):
        return self._cut(buf, segments, nbytes if nbytes is not None else buf.nbytes)

    def submit(self, buf, segments=None, nbytes=None):
        assert len(self.tickets) < self.inflight, "bench submitted more batches than engine slots"
        assert id(buf) not in self.in_flight_bufs, "a batch was resubmitted before its previous pass was collected"
        t = self.next
        self.next += 1
        self.tickets[t] = (self._cut(buf, segments, nbytes), id(buf))
        self.in_flight_bufs.add(id(buf))
        return t

    def timing(self, t):
        return {"h2d_ms": 0.0, "scan_ms": 2.0, "resolve_ms": 0.5, "sha_ms": 40.0, "total_ms": 42.5,
                "ncandidates": 10, "nrecords": int(self.tickets[t][0].size), "retries": 0}

    def done(self, t):
        return t == max(self.tickets)   # the newest ticket "finishes" first: out of order, always progress

    def collect(self, t):
        recs, bid = self.tickets.pop(t)
        self.in_flight_bufs.discard(bid)
        return recs

    def dedup(self, records):
        """first-occurrence duplicate flags + stats, like pbsgpu_dedup_host"""
        seen, dup = set(), np.zeros(records.size, dtype=np.uint8)
        ub = 0
        for i, d in enumerate(records["digest"]):
            k = d.tobytes()
            dup[i] = k in seen
            if k not in seen:
                ub += int(records["size"][i])
            seen.add(k)
        return dup, {"nrecords": int(records.size), "nunique": len(seen), "total_bytes": int(records["size"].sum()),
                     "unique_bytes": ub}

    def close(self):
        assert not self.tickets, "bench left tickets uncollected"


class _RereadEngine(_FakeEngine):
    def submit(self, buf, segments=None, nbytes=None):   # the round-1 protocol re-reads one buffer on purpose
        self.in_flight_bufs.clear()
        return super().submit(buf, segments, nbytes)


def _patch(monkeypatch, engine_cls=_FakeEngine):
    import torch

    import pbs_plus_amd

    sys.path.insert(0, ROOT)
    monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(pbs_plus_amd, "Engine", engine_cls)
    monkeypatch.setattr(pbs_plus_amd, "PageRing", _FakeRing)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)


class _FakeRing:
    """Stand-in for pbs_plus_amd.PageRing as far as bench.py drives it: a stream's bytes are generated on the host by the
    oracle's twin of the device generator, a fraction per fill() call (pages trickle in), and cut + hashed by the oracle
    when the stream ends. Records come out in two instalments so that bench.py's poll loop is exercised."""
    PAGE = 1 << 16

    def __init__(self, eng, arena_bytes=0, page_bytes=0, max_streams=0, sha_cus=0, round_pages=0):
        self.eng, self.streams, self.next, self.rounds, self.launches = eng, {}, 0, 0, 0
        self.page_bytes = self.PAGE
        self.running, self.bytes = False, 0
        self.svc_bytes0 = 0

    def open(self):
        self.next += 1
        self.streams[self.next] = {"parts": [], "final": False, "recs": None, "given": 0}
        return self.next

    def fill(self, sid, seed, kind, nbytes, final=False):
        st = self.streams[sid]
        take = min(nbytes, 3 * self.PAGE)          # a few pages per call: the bench has to come back
        if take < nbytes:
            take -= take % self.PAGE
        off = sum(p.size for p in st["parts"])
        st["parts"].append(self.eng.O.fill(take, seed, kind, stream_off=off))
        if take == nbytes and final:
            st["final"] = True
        self.bytes += take
        return take

    def fill_pieces(self, sid, pieces, nbytes, final=False):
        st = self.streams[sid]
        if pieces is not None:
            st["table"] = np.asarray(pieces, dtype=np.uint64).reshape(-1, 4)
            assert int(st["table"][0, 0]) == 0 and not (st["table"][:, :3] & 15).any()
        take = min(nbytes, 3 * self.PAGE)
        if take < nbytes:
            take -= take % self.PAGE
        off = sum(p.size for p in st["parts"])
        out = np.empty(take, dtype=np.uint8)
        for dst, ln, so, seed in st["table"]:                 # the stand-in's "device generator": the oracle's twin, piece by piece
            lo, hi = max(int(dst), off), min(int(dst + ln), off + take)
            if lo < hi:
                self.eng.O.fill(hi - lo, int(seed), 4, stream_off=int(so) + lo - int(dst), out=out[lo - off:hi - off])
        st["parts"].append(out)
        if take == nbytes and final:
            st["final"] = True
        self.bytes += take
        return take

    def pump(self):
        if not self.running:
            self.running = True
            self.launches += 1
        self.rounds += 1
        for st in self.streams.values():
            if st["final"] and st["recs"] is None:
                data = np.concatenate(st["parts"]) if st["parts"] else np.zeros(0, np.uint8)
                st["recs"] = self.eng.O.chunk_and_digest(self.eng.O.new_config(self.eng.avg), data, [(0, data.size)])
                st["parts"] = []

    def poll(self, sid, cap=4096):
        st = self.streams[sid]
        if st["recs"] is None:
            return np.zeros(0, dtype=st_dtype()), False
        lo = st["given"]
        hi = min(st["recs"].size, lo + max(1, st["recs"].size // 2 + 1), lo + cap)
        st["given"] = hi
        return st["recs"][lo:hi].copy(), hi == st["recs"].size

    def poll_any(self, cap=16384, fcap=4096):
        parts, fin = [], []
        for sid in list(self.streams):
            recs, done = self.poll(sid, cap)
            if recs.size:
                recs = recs.copy()
                recs["segment"] = sid
                parts.append(recs)
            if done and not self.streams[sid].get("reported"):
                self.streams[sid]["reported"] = True
                fin.append(sid)
        return (np.concatenate(parts) if parts else np.zeros(0, dtype=st_dtype())), np.array(fin, dtype=np.uint32)

    def close_stream(self, sid):
        assert self.streams[sid]["given"] == self.streams[sid]["recs"].size
        del self.streams[sid]

    def quiesce(self):
        self.running = False
        self.svc_last, self.svc_bytes0 = self.bytes - self.svc_bytes0, self.bytes

    def stats(self):
        return {"page_bytes": self.PAGE, "pages_total": 64, "sha_cus": 208, "rounds": self.rounds, "service_launches": self.launches,
                "service_ms_last": 5.0, "service_bytes_last": getattr(self, "svc_last", 0), "chunks": 0, "bytes_enqueued": self.bytes}

    def express(self):
        return (16, 13 << 20)

    def probe(self):
        self._pn = getattr(self, "_pn", 0) + 1
        n = 4096 * self._pn
        return {"pair_steps": n, "pair_cycles": n * 4000, "pair_ticks": n * 178, "express_steps": n, "express_cycles": n * 5800,
                "express_ticks": n * 256}

    @staticmethod
    def probe_delta(a, b):
        import pbs_plus_amd.engine as E
        return E.PageRing.probe_delta(a, b)

    def close(self):
        pass


def st_dtype():
    from pbs_plus_amd import RECORD_DTYPE
    return RECORD_DTYPE


def _run(monkeypatch, argv):
    import bench

    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    lines = [ln for ln in buf.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def _check_common(d, steps, warmup):
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert d["unit"] == "GiB/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "hbm", "latency_bound", "kernels"):
        assert key in r, key
    # the path is bound by integer VALU issue (SHA-256), not by HBM and not by MFMA: the contract names it
    assert r["bound"] in ("hbm", "mfma", "valu") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["hbm"]["frac"] - r["achieved"] / r["hbm"]["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes_per_step"]
    assert {"k_sha256_pair<RecordSource>", "k_scan3<34,4>", "resolve_chain"} <= set(r["kernels"])
    lb = r["latency_bound"]
    assert lb["resident_bytes"] == d["config"]["resident_bytes_per_gpu"] and lb["bound_GiBps"] > 0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0
    assert c["records_match_gpu"] is True and c["records_checked"] > 0
    # parity is checked across every resident slot (restart points), not only at the front of batch 0
    sp = c["whole_batch_restart_points"]
    assert sp["ok"] is True and sp["slots"] == d["config"]["resident_batches"] and sp["records_checked"] > 0
    assert sp["max_offset"] >= 0.9 * d["config"]["bytes_per_batch"]   # the tail of a slot was part of a span


@pytest.mark.parametrize("slots,steps,collect", [(4, 9, "any"), (1, 2, "fifo"), (3, 7, "fifo")])
def test_bench_default_workload_contract(monkeypatch, slots, steps, collect):
    _patch(monkeypatch)
    d = _run(monkeypatch, ["--workload", "stream64g", "--gib", str(16 / 1024), "--avg", "65536", "--steps", str(steps), "--warmup", "1",
                           "--slots", str(slots), "--cpu-sample-gib", str(8 / 1024), "--collect", collect])
    _check_common(d, steps, 1)
    assert d["scaling"] == "weak"
    assert d["config"]["resident_batches"] == slots and d["config"]["inflight_batches"] == slots
    assert d["config"]["distinct_data_per_slot"] is True and d["config"]["collect"] == collect
    assert d["config"]["resident_bytes_per_gpu"] == slots * d["config"]["bytes_per_batch"]


def test_bench_default_line_carries_the_other_configs(monkeypatch):
    """configs[2..4] ride in the default line as short legs, each with its own oracle check (here at reduced shapes; the
    host-fed legs need the real stream writer and are covered by the -m gpu suite)"""
    _patch(monkeypatch)
    d = _run(monkeypatch, ["--workload", "stream64g", "--gib", str(16 / 1024), "--avg", "65536", "--steps", "4", "--warmup", "1", "--cpu-sample-gib",
                           str(8 / 1024), "--extras-gib", str(32 / 1024), "--extras-file-mib", "1",
                           "--extras", "manyfiles,corpus_dup,rechunk"])
    _check_common(d, 4, 1)
    w = d["workloads"]
    assert set(w) == {"manyfiles", "corpus_dup", "rechunk", "total_seconds"}
    for name in ("manyfiles", "corpus_dup", "rechunk"):
        leg = w[name]
        assert "error" not in leg, leg
        assert leg["value"] > 0 and leg["steps"] == 4 and leg["records_match_gpu"] is True and leg["records_checked"] > 0
    dd = w["corpus_dup"]["results"]["dedup"]
    assert abs(dd["duplicate_bytes_frac"] - dd["expected_duplicate_frac"]) < 0.02
    assert w["rechunk"]["results"]["reused_chunk_bytes_frac"] > 0.3


def test_bench_ring_workload_is_the_default_and_keeps_the_contract(monkeypatch):
    """The default line = configs[1] through the page ring: a step is one whole file, every file new data, the roofline
    entry is the persistent service launch, parity = prefix + restart points over EVERY timed file on regenerated bytes;
    the batch path rides along in `workloads`."""
    _patch(monkeypatch)
    d = _run(monkeypatch, ["--gib", str(24 / 1024), "--avg", "65536", "--steps", "5", "--warmup", "2", "--cpu-sample-gib",
                           str(8 / 1024), "--ring-streams", "3", "--extras-gib", str(16 / 1024), "--extras-file-mib", "1",
                           "--extras", "batch,manyfiles"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["steps"] == 5 and d["warmup"] == 2 and d["n_gpus"] == 1 and d["unit"] == "GiB/s" and d["value"] > 0
    assert "page ring" in d["config"]["path"] and d["config"]["files_in_flight"] == 3
    assert d["config"]["bytes_per_step"] == 24 << 20 and d["config"]["distinct_data_per_step"] is True
    r = d["roofline"]
    for key in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "valu", "path", "single_file"):
        assert key in r, key
    assert "RingSource" in r["kernel"] and r["service_launch_bytes"] == 5 * (24 << 20)
    # the contract's form: the dominant kernel's algorithmic bytes / its duration against the HBM peak; the integer-issue
    # ceiling that actually binds SHA-256 sits beside it
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["valu"]["frac"] - r["achieved"] / r["valu"]["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["records_match_gpu"] is True and c["kind"] == "port" and c["cores"] == 1
    sp = c["whole_file_restart_points"]
    assert sp["ok"] is True and sp["files"] == 6 and sp["max_offset"] == 24 << 20   # 5 timed files + the single-file pass
    w = d["workloads"]
    assert "error" not in w["batch_path_stream64g"] and w["batch_path_stream64g"]["value"] > 0
    assert w["batch_path_stream64g"]["records_match_gpu"] is True and w["manyfiles"]["records_match_gpu"] is True


def test_bench_reread_protocol_is_labelled(monkeypatch):
    _patch(monkeypatch, _RereadEngine)
    d = _run(monkeypatch, ["--workload", "stream64g", "--gib", str(16 / 1024), "--avg", "65536", "--steps", "6", "--warmup", "1", "--reread", "3",
                           "--cpu-sample-gib", str(8 / 1024)])
    _check_common(d, 6, 1)
    assert d["config"]["distinct_data_per_slot"] is False and d["config"]["inflight_batches"] == 3
    assert "REREAD" in d["config"]["workload"]


def test_bench_manyfiles_contract(monkeypatch):
    _patch(monkeypatch)
    d = _run(monkeypatch, ["--workload", "manyfiles", "--gib", str(24 / 1024), "--file-mib", "1.5", "--avg", "65536",
                           "--steps", "4", "--warmup", "1", "--cpu-sample-gib", str(6 / 1024)])
    _check_common(d, 4, 1)
    assert d["config"]["files_per_batch"] == 16 and d["config"]["resident_batches"] == 2
    assert "entropy_classes" in d["config"]


def test_bench_corpus_dup_reports_expected_duplicate_fraction(monkeypatch):
    _patch(monkeypatch)
    d = _run(monkeypatch, ["--workload", "corpus_dup", "--gib", str(64 / 1024), "--file-mib", "1", "--avg", "65536",
                           "--steps", "4", "--warmup", "2", "--cpu-sample-gib", str(4 / 1024)])
    _check_common(d, 4, 2)
    dd = d["results"]["dedup"]
    assert dd["records"] > dd["unique"] > 0
    # whole segments are duplicated, chunk boundaries are content-defined: the duplicate BYTES match exactly
    assert abs(dd["duplicate_bytes_frac"] - dd["expected_duplicate_frac"]) < 1e-9
    assert 0.25 < dd["expected_duplicate_frac"] < 0.55


def test_bench_rechunk_reuses_most_chunks(monkeypatch):
    _patch(monkeypatch)
    d = _run(monkeypatch, ["--workload", "rechunk", "--gib", str(64 / 1024), "--file-mib", "4", "--avg", "16384",
                           "--steps", "4", "--warmup", "1", "--cpu-sample-gib", str(8 / 1024)])
    _check_common(d, 4, 1)
    # 2 % of the bytes edited, ~16 KiB chunks: the vast majority of chunks must be found again
    assert 0.80 < d["results"]["reused_chunk_bytes_frac"] < 0.995


def _patched_main(argv):
    """bench.main() with the GPU pieces stubbed (entry point of the 2-rank subprocesses below)."""
    import torch

    import pbs_plus_amd

    sys.path.insert(0, ROOT)
    import bench

    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.device_count = lambda: 1
    pbs_plus_amd.Engine = _FakeEngine
    pbs_plus_amd.PageRing = _FakeRing
    if os.environ.get("PBS_BENCH_REDUCE") == "cabi":
        # stand-in for pbs_plus_amd.Comm (pbsgpu_comm_create needs RCCL and one GPU per rank): the same collective contract —
        # dedup(records, cap) on every rank, statistics of the union — over the process group that is already up
        import pbs_plus_amd.dist as pdist

        class _FakeComm:
            def __init__(self, eng):
                self.eng, self.calls = eng, 0

            def dedup(self, recs, cap_records, want_flags=False):
                import torch.distributed as dist
                assert recs.size <= cap_records
                parts = [None] * dist.get_world_size()
                dist.all_gather_object(parts, recs)
                self.calls += 1
                dup, stats = self.eng.dedup(np.concatenate(parts))
                return None, stats

            def close(self):
                assert self.calls >= 2          # first contact + at least one timed reduce went through here

        pdist.make_comm = lambda eng, device=None, group=None: _FakeComm(eng)
    sys.argv = ["bench.py"] + list(argv)
    bench.main()


def _two_ranks(args, extra_env=None):
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_bench_contract import _patched_main\n"
            "_patched_main(%r)\n" % (ROOT, os.path.join(ROOT, "tests"), args))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PBS_BENCH_BACKEND="gloo", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    json_lines = [[ln for ln in o.splitlines() if ln.startswith("{")] for o, _ in outs]
    assert len(json_lines[0]) == 1 and len(json_lines[1]) == 0, json_lines   # rank 0 only
    return json.loads(json_lines[0][0])


def test_bench_two_ranks_gloo_prints_one_aggregate_line():
    """The N>1 branch of bench.py (barrier, per-step digest-set all-gather, MAX-over-ranks timing, rank 0 prints)
    with world_size 2 over gloo on CPU — the launch shape the driver uses with RCCL on the GPU node."""
    d = _two_ranks(["--gpus", "2", "--workload", "stream64g", "--gib", str(16 / 1024), "--avg", "65536", "--steps", "4", "--warmup", "1",
                    "--slots", "3", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    one_rank_bytes = d["config"]["bytes_per_batch"]
    # whole-job aggregate: both ranks' bytes over the (max over ranks) wall time
    assert abs(d["value"] - 2 * 4 * one_rank_bytes / (1 << 30) / (d["ms_per_step"] * 4e-3)) / d["value"] < 0.02
    assert "cpu_baseline" not in d  # rank 0 at N=1 only
    assert d["results"]["dedup_last_step"]["nrecords"] > 0


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_corpus_dup_two_ranks_digest_set_reduce(scaling):
    """configs[3] over two ranks (gloo): segments sharded, ONE all-gather of the records per pass, dedup over the
    union; the duplicate fraction found equals the fraction planted (40 % of the segments, +- the seeded draw)."""
    d = _two_ranks(["--gpus", "2", "--workload", "corpus_dup", "--gib", str(64 / 1024), "--file-mib", "1", "--avg",
                    "65536", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--scaling", scaling])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling
    dd = d["results"]["dedup"]
    assert abs(dd["duplicate_bytes_frac"] - dd["expected_duplicate_frac"]) < 1e-9
    assert 0.25 < dd["expected_duplicate_frac"] < 0.55
    if scaling == "strong":
        assert d["config"]["corpus_segments"] == 64 and d["config"]["segments_per_gpu"] == 32
    else:
        assert d["config"]["corpus_segments"] == 128 and d["config"]["segments_per_gpu"] == 64


def test_bench_ring_two_ranks_gloo_reduces_in_step_order():
    """The DEFAULT workload with world size 2 over gloo: files may finish in any order on a rank, the per-file digest-set
    reduce is issued in step order on every rank (a collective sequence that differs between ranks would hang), the
    first contact of the reduce happens before the ring starts."""
    d = _two_ranks(["--gpus", "2", "--gib", str(12 / 1024), "--avg", "65536", "--steps", "5", "--warmup", "1",
                    "--ring-streams", "3", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["value"] > 0 and d["scaling"] == "weak"
    assert "page ring" in d["config"]["path"] and "2 rank" in d["config"]["parallelism"]
    st = d["results"]["dedup_last_step"]
    assert st["nrecords"] > 0 and st["nunique"] == st["nrecords"]        # two different files: no shared chunks


def test_bench_two_ranks_time_the_reduce_of_the_c_abi_communicator():
    """Round 6: the digest-set reduce INSIDE the timed region of an N-rank line goes through the C ABI's own communicator
    (pbsgpu_comm_create + pbsgpu_digest_allgather_dedup: what a Go host binds) whenever every rank has its own GPU; the
    torch.distributed path is the cross-check outside the timed region (rounds 3-5: the other way round). World size 2 over
    gloo with a stand-in communicator that keeps the collective contract: every timed reduce and the first contact go through
    it, the line says so, and both paths report the same union."""
    d = _two_ranks(["--gpus", "2", "--gib", str(12 / 1024), "--avg", "65536", "--steps", "4", "--warmup", "1",
                    "--ring-streams", "2", "--no-cpu-baseline"], extra_env={"PBS_BENCH_REDUCE": "cabi"})
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0
    c = d["results"]["c_abi_digest_reduce"]
    assert c["ok"] is True and c["timed_through_c_abi"] is True and c["equals_torch_path"] is True, c
    assert "pbsgpu_digest_allgather_dedup" in c["timed_path"]
    assert d["results"]["dedup_last_step"]["nrecords"] == c["stats"]["nrecords"] or d["results"]["dedup_last_step"]["nrecords"] > 0
    # ... and the batch workloads take the same switch
    d2 = _two_ranks(["--gpus", "2", "--workload", "corpus_dup", "--gib", str(64 / 1024), "--file-mib", "1", "--avg", "65536",
                     "--steps", "4", "--warmup", "2", "--no-cpu-baseline"], extra_env={"PBS_BENCH_REDUCE": "cabi"})
    assert "pbsgpu_digest_allgather_dedup" in d2["results"]["digest_reduce_path"]
    dd = d2["results"]["dedup"]
    assert abs(dd["duplicate_bytes_frac"] - dd["expected_duplicate_frac"]) < 1e-9


@pytest.mark.parametrize("mode", ["ring_manyfiles", "ring_corpus_dup", "ring_rechunk"])
def test_bench_many_files_through_the_ring(monkeypatch, mode):
    """configs[2] / configs[3] with one ring stream per file (poll_any): whole sampled files vs the oracle, every file tiled
    by its records, planted duplicates found exactly."""
    _patch(monkeypatch)
    d = _run(monkeypatch, ["--workload", mode, "--gib", str(12 / 1024), "--file-mib", "0.5", "--avg", "16384", "--steps", "3",
                           "--warmup", "1", "--cpu-sample-gib", str(2 / 1024)])
    assert d["steps"] == 3 and d["value"] > 0 and d["config"]["files_per_step"] == 24
    c = d["cpu_baseline"]
    assert c["records_match_gpu"] is True and c["every_file_tiled_by_its_records"] is True and c["records_checked"] > 50
    if mode == "ring_corpus_dup":
        dd = d["results"]["dedup"]
        assert abs(dd["duplicate_bytes_frac"] - dd["expected_duplicate_frac"]) < 1e-9 and dd["expected_duplicate_frac"] > 0.15
    if mode == "ring_rechunk":                # configs[4]: the edited corpus re-uses most chunks of the snapshot ingested before
        rr = d["results"]
        assert 0.3 < rr["reused_chunk_bytes_frac"] < 1.0 and rr["base_snapshot_chunks"] > 50
        assert abs(rr["edited_bytes"] - 3 * 24 * (512 << 10)) < 0.05 * 3 * 24 * (512 << 10)


def test_bench_gpus_flag_spawns_its_own_ranks_and_every_rank_is_checked():
    """`bench.py --gpus N` used to be parsed and ignored — without torchrun it ran ONE rank and printed
    n_gpus: 1. Now it starts the N ranks itself (rendezvous on 127.0.0.1); the line says n_gpus = N, and rank 0's
    cpu_baseline carries the oracle's verdict over EVERY rank's files (an N-rank record without parity evidence is
    worth nothing). Over gloo on CPU here; the driver's torchrun launch takes the other branch (WORLD_SIZE set)."""
    import subprocess

    args = ["--gpus", "2", "--gib", str(12 / 1024), "--avg", "65536", "--steps", "3", "--warmup", "1", "--ring-streams", "2",
            "--cpu-sample-gib", str(4 / 1024)]
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_bench_contract import _patched_main\n"
            "_patched_main(%r)\n" % (ROOT, os.path.join(ROOT, "tests"), args))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(PBS_BENCH_BACKEND="gloo", PBS_BENCH_SPAWN_CMD=json.dumps([sys.executable, "-c", code]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0
    c = d["cpu_baseline"]
    assert c["records_match_gpu"] is True and c["all_ranks"]["ranks"] == 2 and c["all_ranks"]["records_match_gpu"] is True
    assert c["all_ranks"]["files_checked"] == 2 * (3 + 1)            # every timed file + the single-file pass, on both ranks
    assert c["all_ranks"]["records_checked"] > c["front_of_file"]["records_checked"]
    # a launcher whose world size disagrees with --gpus is refused, not silently re-labelled
    bad = subprocess.run([sys.executable, "-c", code], env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2 and "WORLD_SIZE=3" in bad.stderr
