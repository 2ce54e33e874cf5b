#!/usr/bin/env python3
"""bench.py — GiB/s ingested through Buzhash CDC + per-chunk SHA-256 on MI355X.

A "step" = one full pass of the hot path over one HBM-resident batch: candidate scan ->
compaction/resolve -> SHA-256 of every chunk -> (end, digest) records on the host. Chunker
parameters are the reference's production ones, buzhash.NewConfig(4 << 20)
(internal/pxarmount/commit_orchestrate.go:144, internal/tapeio/converter.go:248).

Every in-flight slot owns DIFFERENT bytes: the corpus is cut into `--slots` resident batches
(4 x 64 GiB for the default workload, i.e. the whole HBM) and a batch is resubmitted only after its
previous pass has been collected, exactly as a deployment would refill a buffer. (`--reread N`
reproduces round 1's protocol of N overlapping passes over the same bytes, for comparison only.)

Workloads (BASELINE.json configs):
  stream64g   configs[1]  one 64 GiB random stream per batch                         (default)
  manyfiles   configs[2]  2048 x 64 MiB files per 128 GiB device batch, entropy class = file % 4
  corpus_dup  configs[3]  this GPU's 128 GiB share of the 1 TiB corpus, 40 % of the 64 MiB segments are
                          copies of an earlier segment; digest-set reduce (all-gather + device dedup) per pass
  rechunk     configs[4]  the same share after 2 % byte edits (overwrite / insert / delete extents),
                          re-chunked; reports the re-used chunk fraction vs the base corpus
  hostfeed    the drop-in's real feed: host buffers written through pbsgpu_stream_write by several
              producer threads (PCIe-inclusive; never the headline value)
  verify      A8/A9: whole-file SHA-256 / XXH3-64 batches (verification.HashFile)

N > 1 (torch.distributed.run, one rank per GPU, RCCL): every rank owns its own batches (segments and
streams are independent, no data-path collective); the only exchange is the digest-set all-gather.

Prints ONE JSON line on rank 0 with `roofline` and `cpu_baseline` (the C oracle, SHA-NI, one host
thread, on a bounded sample of the same workload, which also re-checks the GPU records bit for bit).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per in-flight batch: ROCm's default of 4 would make engine streams share queues
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

import numpy as np  # noqa: E402
import torch  # noqa: E402

GiB = 1 << 30
MiB = 1 << 20
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_BYTES = 288e9           # spec capacity
# Integer VALU issue ceiling for the SHA-256 instruction mix (v_alignbit / v_bitop3 / v_add3 / v_bfi, all VOP3):
# 620 G wave64-instructions/s chip-wide at 8 waves/SIMD (profiles/r01_ubench_int_valu_issue.log) = 39.7 T lane-ops/s
VALU_PEAK_TOPS = 39.7
# VALU instructions per 64-byte block that gfx950's ISA cannot go below: 64 rounds x 14 + 48 schedule words x 11
# (sigma0/sigma1 = 2 v_alignbit + v_lshrrev + v_bitop3 each, three adds incl. K) + 16 K adds + 16 v_perm + 8 = 1464.
# The kernels emit 906 (consumer) + 744 (producer, incl. addressing) = 1650: profiles/r02_sha_isa_counts.txt
SHA_OPS_PER_BYTE = 1464 / 64.0
SHA_VALU_GBS = VALU_PEAK_TOPS * 1e3 / SHA_OPS_PER_BYTE
CHAIN_US_PER_BLOCK = 1.655  # measured serial chain of the wave-pair kernel: 64 rounds x 14 instr x 4.25 cycles at 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 20; hostfeed 48: its final drain is one chunk chain, ~0.6 s, whatever the length)")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="ring",
                    choices=("ring", "ring_manyfiles", "ring_corpus_dup", "ring_rechunk", "stream64g", "manyfiles", "corpus_dup", "rechunk",
                             "hostfeed", "verify"))
    ap.add_argument("--arena-gib", type=float, default=None, help="ring: device memory of the page arena (default: free HBM - 12 GiB)")
    ap.add_argument("--ring-streams", type=int, default=4, help="ring: files in flight at once")
    ap.add_argument("--ring-sha-cus", type=int, default=0, help="ring: CUs of the SHA-256 service (0 = library default)")
    ap.add_argument("--ring-round-pages", type=int, default=0)
    ap.add_argument("--ring-kind", type=int, default=4, help="ring: synthetic generator (4 = cheap ARX, 0 = splitmix64)")
    ap.add_argument("--gib", type=float, default=None, help="bytes per batch in GiB (default: 64; manyfiles 128)")
    ap.add_argument("--slots", type=int, default=None, help="resident batches = batches in flight (default 4; manyfiles 2)")
    ap.add_argument("--file-mib", type=float, default=None,
                    help="segment size of the many-file / corpus workloads (default 64) and of verify (default 1)")
    ap.add_argument("--avg", type=int, default=4 << 20)
    ap.add_argument("--reread", type=int, default=0,
                    help="round-1 protocol: this many overlapping passes over ONE resident batch (not the default)")
    ap.add_argument("--collect", choices=("fifo", "any"), default="any")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="corpus workloads at N > 1: weak = every rank owns a full share; strong = one share split over the ranks")
    ap.add_argument("--producers", type=int, default=8, help="hostfeed: producer threads (one stream each)")
    ap.add_argument("--tee", action="store_true", help="hostfeed: every 1 GiB step is one file with the XXH3-64 tee on")
    ap.add_argument("--archives", type=int, default=1,
                    help="hostfeed: archives per producer, written back to back (finish_begin: an archive's last chunks are "
                         "hashed while the next one is being written)")
    ap.add_argument("--cpu-sample-gib", type=float, default=2.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spread-check", action="store_true", help="skip the whole-batch restart-point parity check")
    ap.add_argument("--spread-points", type=int, default=40, help="restart points per resident slot")
    ap.add_argument("--no-extras", action="store_true",
                    help="default workload only: skip the short configs[2..4] / host-fed legs folded into the line")
    ap.add_argument("--extras", default="batch,manyfiles,corpus_dup,rechunk,ring_manyfiles,ring_corpus_dup,ring_rechunk,hostfeed1,hostfeed8",
                    help="which short legs the default line carries (they run when --gib is left at its default, or when "
                         "--extras-gib names a reduced shape)")
    ap.add_argument("--extras-gib", type=float, default=None, help="bytes per device batch of the short legs (tests)")
    ap.add_argument("--extras-file-mib", type=float, default=64.0)
    ap.add_argument("--seed", type=int, default=2)
    a = ap.parse_args()
    a.brief = False
    if a.steps is None:
        a.steps = 48 if a.workload == "hostfeed" else 20
    if a.file_mib is None:
        a.file_mib = 1.0 if a.workload == "verify" else 64.0
    return a


# ---------------------------------------------------------------------------------------------------
# corpus description: every batch is a list of fills (offset, nbytes, seed, kind) so that the oracle can
# regenerate any sampled range on the host (oracle.fill is the CPU twin of pbsgpu_fill_device)
# ---------------------------------------------------------------------------------------------------
class Batch:
    def __init__(self, buf, nbytes, segs=None, fills=None, label=""):
        self.buf, self.nbytes, self.label = buf, int(nbytes), label
        self.segs = None if segs is None else np.ascontiguousarray(segs, dtype=np.uint64).reshape(-1, 2)
        self.fills = fills or []
        self.ticket = None
        self.last = None      # records of the most recent collected pass

    @property
    def nseg(self):
        return 1 if self.segs is None else int(self.segs.shape[0])


def splitmix64(x):
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def alloc_batches(eng, nbytes, want):
    """`want` device buffers of nbytes each; fewer when HBM (288 GB) cannot hold them (reported in config)."""
    from pbs_plus_amd import PbsGpuError
    bufs = []
    for _ in range(want):
        try:
            bufs.append(eng.alloc(nbytes))
        except PbsGpuError:
            break
    if not bufs:
        raise SystemExit("bench: could not allocate even one batch")
    return bufs


class Workload:
    name = "?"
    scaling = "weak"

    def __init__(self, a, eng, rank, world):
        self.a, self.eng, self.rank, self.world = a, eng, rank, world
        self.batches = []
        self.extra = {}

    def describe(self):
        return {}

    def after_collect(self, batch, recs, ctx):
        pass

    def finish(self, ctx):
        pass

    def cpu_sample(self):
        """(host bytes, segments, gpu records of exactly those segments or None) for the oracle leg."""
        raise NotImplementedError

    def free(self):
        for b in self.batches:
            if hasattr(b.buf, "free"):
                b.buf.free()


class Stream64g(Workload):
    name = "stream64g"

    def __init__(self, a, eng, rank, world):
        super().__init__(a, eng, rank, world)
        gib = 64.0 if a.gib is None else a.gib
        nbytes = int(gib * GiB) & ~7
        slots = 1 if a.reread else (4 if a.slots is None else a.slots)
        for i, buf in enumerate(alloc_batches(eng, nbytes, slots)):
            seed = a.seed + 1000 * rank + i
            eng.fill(buf.ptr, nbytes, seed=seed, kind=0)
            self.batches.append(Batch(buf, nbytes, None, [(0, nbytes, seed, 0)], f"stream{i}"))
        self.gib = gib

    def describe(self):
        return {"workload": f"single {self.gib:g} GiB random stream per batch, Buzhash CDC avg {self.a.avg >> 20} MiB "
                            f"(min avg/4, max 4*avg) + SHA-256 per chunk (BASELINE.json configs[1])"}

    def after_collect(self, batch, recs, ctx):
        if ctx.dist is not None:   # cross-stream duplicate detection over all ranks' files of this step
            stats = reduce_records(ctx, self.eng, recs)
            self.extra["dedup_last_step"] = {k: int(v) for k, v in stats.items()}

    def cpu_sample(self):
        from oracle import oracle as O
        b = self.batches[0]
        n = int(min(self.a.cpu_sample_gib * GiB, b.nbytes)) & ~7
        return O.fill(n, b.fills[0][2], 0), [(0, n)], b.last, "prefix"


class ManyFiles(Workload):
    """configs[2]: 10 000 x 64 MiB files streamed through 128 GiB device batches; two batches (4096 files,
    256 GiB) are resident and alternate. Entropy class by file index % 4: random / all-zero /
    repeating 4 KiB block / random with 30 % zero extents. Fresh chunker state per file, forced cut at file end."""
    name = "manyfiles"

    def __init__(self, a, eng, rank, world):
        super().__init__(a, eng, rank, world)
        gib = 128.0 if a.gib is None else a.gib
        fbytes = int(a.file_mib * MiB) & ~7
        nfiles = max(1, int(gib * GiB) // fbytes)
        nbytes = nfiles * fbytes
        slots = 2 if a.slots is None else a.slots
        for i, buf in enumerate(alloc_batches(eng, nbytes, slots)):
            fills, segs = [], np.zeros((nfiles, 2), dtype=np.uint64)
            for f in range(nfiles):
                g = (rank * slots + i) * nfiles + f           # global file index
                seed, kind = a.seed + 7919 * g + 1, g % 4
                eng.fill(buf.ptr + f * fbytes, fbytes, seed=seed, kind=kind)
                fills.append((f * fbytes, fbytes, seed, kind))
                segs[f] = (f * fbytes, fbytes)
            self.batches.append(Batch(buf, nbytes, segs, fills, f"files{i}"))
        self.nfiles, self.fbytes = nfiles, fbytes

    def describe(self):
        return {"workload": f"{self.nfiles} x {self.fbytes / MiB:g} MiB mixed-entropy files per device batch "
                            f"(10 000-file job streamed batch by batch; BASELINE.json configs[2])",
                "files_per_batch": self.nfiles, "file_bytes": self.fbytes,
                "distinct_files_resident": self.nfiles * len(self.batches),
                "note": "the resident batches (distinct files each) are resubmitted unchanged step after step: HBM holds "
                        "4096 of the 10 000 files, the throughput of a batch does not depend on which files it holds",
                "entropy_classes": "file % 4: random / zeros / repeating 4 KiB / random with 30 % zero extents"}

    def cpu_sample(self):
        return sample_segments(self, self.batches[0], per_class=max(1, int(self.a.cpu_sample_gib * GiB / self.fbytes) // 4))


def sample_segments(w, b, per_class=4, classes=4):
    """Regenerate a few segments of batch b on the host (oracle.fill) and pick the GPU's records for them."""
    from oracle import oracle as O
    idx = []
    for c in range(classes):
        idx += [i for i in range(c, b.nseg, classes)][:per_class]
    idx = sorted(set(idx))
    total = sum(b.fills[i][1] for i in idx)
    host = np.empty(total, dtype=np.uint8)
    segs, off = [], 0
    for i in idx:
        _, n, seed, kind = b.fills[i]
        O.fill(n, seed, kind, out=host[off:off + n])
        segs.append((off, n))
        off += n
    gpu = None
    if b.last is not None:
        parts = []
        for k, i in enumerate(idx):
            r = b.last[b.last["segment"] == i].copy()
            r["segment"] = k
            parts.append(r)
        gpu = np.concatenate(parts) if parts else None
    return host, segs, gpu, "segments"


def dup_roots(nseg_total, seed, pct=40):
    """Content id of every corpus segment: `pct` % of the segments are exact copies of an EARLIER segment chosen
    by a seeded hash; copies of copies resolve to the original."""
    g = np.arange(nseg_total, dtype=np.uint64)
    r = splitmix64(g * np.uint64(2) + np.uint64(seed) * np.uint64(0x10001))
    is_dup = ((r % np.uint64(100)) < np.uint64(pct)) & (g > 0)
    src = splitmix64(r) % np.maximum(g, np.uint64(1))
    root = g.copy()
    for i in range(nseg_total):          # ascending: the source's root is already final
        if is_dup[i]:
            root[i] = root[int(src[i])]
    return root


class CorpusDup(Workload):
    """configs[3]: 1 TiB = 16 384 x 64 MiB segments, 40 % exact copies; this rank's share (128 GiB = 2048
    segments) is cut into resident batches. Digest-set reduce once per pass over the share."""
    name = "corpus_dup"

    def __init__(self, a, eng, rank, world):
        super().__init__(a, eng, rank, world)
        self.scaling = a.scaling
        fbytes = int(a.file_mib * MiB) & ~7
        share_gib = 128.0 if a.gib is None else a.gib
        share = max(1, int(share_gib * GiB) // fbytes)           # segments per rank (weak) / in total (strong)
        if a.scaling == "strong":
            total = share
            mine = np.arange(rank, total, world)                 # round-robin sharding
        else:
            total = share * world
            mine = np.arange(rank * share, (rank + 1) * share)
        self.root = dup_roots(total, a.seed + 4)
        slots = 2 if a.slots is None else a.slots
        per = -(-len(mine) // slots)
        for i in range(slots):
            ids = mine[i * per:(i + 1) * per]
            if len(ids) == 0:
                continue
            nbytes = len(ids) * fbytes
            buf = alloc_batches(eng, nbytes, 1)[0]
            fills, segs = [], np.zeros((len(ids), 2), dtype=np.uint64)
            for f, g in enumerate(ids):
                seed = a.seed + 104729 * int(self.root[g]) + 5
                eng.fill(buf.ptr + f * fbytes, fbytes, seed=seed, kind=0)
                fills.append((f * fbytes, fbytes, seed, 0))
                segs[f] = (f * fbytes, fbytes)
            bt = Batch(buf, nbytes, segs, fills, f"share{i}")
            bt.ids = ids
            self.batches.append(bt)
        self.total, self.mine, self.fbytes = total, mine, fbytes
        self.pass_recs = {}
        # expected duplicate bytes among the segments the job holds (all ranks): a segment is a duplicate when an
        # earlier segment of the job has the same content id
        held = np.arange(total)
        _, first = np.unique(self.root[held], return_index=True)
        self.expected_dup_frac = 1.0 - len(first) / len(held)

    def describe(self):
        return {"workload": f"{len(self.mine)} x {self.fbytes / MiB:g} MiB segments per GPU of a {self.total}-segment corpus, "
                            f"40 % of the segments exact copies of an earlier one; digest-set all-gather + device dedup "
                            f"once per pass (BASELINE.json configs[3])",
                "segments_per_gpu": int(len(self.mine)), "corpus_segments": int(self.total)}

    def after_collect(self, batch, recs, ctx):
        self.pass_recs[id(batch)] = recs
        if len(self.pass_recs) == len(self.batches):     # one pass over the share is complete: digest-set reduce
            local = np.concatenate([self.pass_recs[id(b)] for b in self.batches])
            self.pass_recs = {}
            if ctx.dist is not None:
                stats = reduce_records(ctx, self.eng, local, ctx.rec_cap * len(self.batches))
            else:
                _, stats = self.eng.dedup(local)
            tb = max(int(stats["total_bytes"]), 1)
            self.extra["dedup"] = {"records": int(stats["nrecords"]), "unique": int(stats["nunique"]),
                                   "duplicate_bytes_frac": round(1.0 - int(stats["unique_bytes"]) / tb, 4),
                                   "expected_duplicate_frac": round(self.expected_dup_frac, 4)}

    def cpu_sample(self):
        return sample_segments(self, self.batches[0], per_class=max(1, int(self.a.cpu_sample_gib * GiB / self.fbytes)),
                               classes=1)


def edit_plan(seg_len, rng, frac=0.02, align=8):
    """Piece table of one edited segment: extents with log-uniform length 4 KiB-4 MiB, 1/3 overwrite / insert /
    delete, totalling ~frac of the bytes. Offsets and lengths are multiples of `align` (8: the
    device gather's word; 16: the generator block of the ring's piece-table producer).
    Returns [(kind, src_off, len)] with kind 0 = copy from the base segment, 1 = new random bytes."""
    budget = int(seg_len * frac)
    edits = []
    m = ~(align - 1)
    while budget > 0:
        ln = int(np.exp(rng.uniform(np.log(4096), np.log(4 << 20)))) & m
        ln = max(align, min(ln, (budget + align) & m))
        pos = int(rng.integers(0, max(1, seg_len - ln))) & m
        edits.append((pos, ln, int(rng.integers(0, 3))))
        budget -= ln
    edits.sort()
    pieces, cur = [], 0
    for pos, ln, kind in edits:
        if pos < cur:
            continue                                   # overlapping extent: skipped
        if pos > cur:
            pieces.append((0, cur, pos - cur))
        if kind == 0:                                  # overwrite
            pieces.append((1, 0, ln))
            cur = pos + ln
        elif kind == 1:                                # insert
            pieces.append((1, 0, ln))
            cur = pos
        else:                                          # delete
            cur = pos + ln
    if cur < seg_len:
        pieces.append((0, cur, seg_len - cur))
    return pieces


class Rechunk(Workload):
    """configs[4]: this GPU's share of the corpus after edits totalling 2 % of the bytes, re-chunked. The base share
    (64 GiB by default so that base + edited batches fit in HBM together with 4 distinct in-flight batches) is chunked
    once outside the timed region; the timed steps pass over the EDITED batches; `reused_chunk_bytes_frac` = share of
    the edited corpus' bytes that fall in chunks whose digest already exists in the base snapshot."""
    name = "rechunk"

    def __init__(self, a, eng, rank, world):
        super().__init__(a, eng, rank, world)
        fbytes = int(a.file_mib * MiB) & ~7
        share_gib = 128.0 if a.gib is None else a.gib
        nseg = max(1, int(share_gib * GiB) // fbytes)
        slots = 4 if a.slots is None else a.slots
        per = -(-nseg // slots)
        rng = np.random.default_rng(a.seed + 5 + 977 * rank)
        base = alloc_batches(eng, per * fbytes, 1)[0]     # one base batch at a time: build, chunk, derive the edit
        newbuf = alloc_batches(eng, max(8, int(per * fbytes * 0.03)) + (8 << 20), 1)[0]
        self.base_digests = []
        self.base_bytes = 0
        for i in range(slots):
            n = min(per, nseg - i * per)
            if n <= 0:
                break
            bsegs = np.zeros((n, 2), dtype=np.uint64)
            fills = []
            for f in range(n):
                g = (rank * slots + i) * per + f
                seed = a.seed + 15485863 * g + 9
                eng.fill(base.ptr + f * fbytes, fbytes, seed=seed, kind=0)
                bsegs[f] = (f * fbytes, fbytes)
                fills.append((seed, fbytes))
            brecs = eng.chunk_and_digest(base, bsegs, nbytes=n * fbytes)
            self.base_digests.append(brecs["digest"].copy())
            self.base_bytes += n * fbytes
            # edited batch: piece table -> device gather; new bytes come from a random pool
            eng.fill(newbuf.ptr, newbuf.nbytes & ~7, seed=a.seed + 31 * (rank * slots + i) + 77, kind=0)
            items, esegs, plan, pos, npos = [], np.zeros((n, 2), dtype=np.uint64), [], 0, 0
            for f in range(n):
                start = pos
                pieces = edit_plan(fbytes, rng)
                for kind, so, ln in pieces:
                    if kind == 0:
                        items.append((0, f * fbytes + so, pos, ln))
                    else:
                        items.append((1, npos, pos, ln))
                        npos += ln
                    pos += ln
                esegs[f] = (start, pos - start)
                plan.append(pieces)
            assert npos <= newbuf.nbytes, (npos, newbuf.nbytes)
            ebuf = alloc_batches(eng, pos + 64, 1)[0]
            it = np.array(items, dtype=np.uint64).reshape(-1, 4)
            for kind, src in ((0, base), (1, newbuf)):
                sel = it[it[:, 0] == kind][:, 1:]
                if len(sel):
                    eng.gather(src, ebuf, sel)
            bt = Batch(ebuf, pos, esegs, None, f"edited{i}")
            bt.plan, bt.base_fills, bt.new_seed = plan, fills, a.seed + 31 * (rank * slots + i) + 77
            self.batches.append(bt)
        base.free()
        newbuf.free()
        self.nseg, self.fbytes = nseg, fbytes
        self.base_set = None

    def describe(self):
        return {"workload": f"{self.nseg} x {self.fbytes / MiB:g} MiB segments per GPU after byte edits totalling 2 % "
                            f"(log-uniform 4 KiB-4 MiB extents, 1/3 overwrite / insert / delete), re-chunked "
                            f"(BASELINE.json configs[4])", "segments_per_gpu": int(self.nseg)}

    def finish(self, ctx):
        if self.base_set is None:
            d = np.concatenate(self.base_digests)
            self.base_set = set(map(bytes, d))
        reused = total = 0
        for b in self.batches:
            if b.last is None:
                continue
            hit = np.fromiter((bytes(x) in self.base_set for x in b.last["digest"]), dtype=bool, count=b.last.size)
            reused += int(b.last["size"][hit].sum())
            total += int(b.last["size"].sum())
        if total:
            self.extra["reused_chunk_bytes_frac"] = round(reused / total, 4)
            self.extra["edited_bytes"] = total

    def cpu_sample(self):
        """Rebuild a few edited segments on the host from their piece tables and run the oracle on them."""
        from oracle import oracle as O
        b = self.batches[0]
        k = max(1, min(b.nseg, int(self.a.cpu_sample_gib * GiB / self.fbytes)))
        newpool_needed = 0
        for f in range(b.nseg):       # offsets into the new-bytes pool are cumulative over the whole batch
            for kind, so, ln in b.plan[f]:
                if kind == 1:
                    newpool_needed += ln
            if f + 1 == k:
                break
        pool = O.fill((newpool_needed + 7) & ~7, b.new_seed, 0)
        parts, segs, off, npos = [], [], 0, 0
        for f in range(k):
            seed, n = b.base_fills[f]
            base = O.fill(n, seed, 0)
            seg = []
            for kind, so, ln in b.plan[f]:
                if kind == 0:
                    seg.append(base[so:so + ln])
                else:
                    seg.append(pool[npos:npos + ln])
                    npos += ln
            s = np.concatenate(seg)
            parts.append(s)
            segs.append((off, s.size))
            off += s.size
        gpu = None
        if b.last is not None:
            gpu = b.last[b.last["segment"] < k].copy()
        return np.concatenate(parts), segs, gpu, "segments"


WORKLOADS = {"stream64g": Stream64g, "manyfiles": ManyFiles, "corpus_dup": CorpusDup, "rechunk": Rechunk}


class Ctx:
    dist = None
    comm_dev = None
    rec_cap = 0
    cabi = None        # what the digest-set reduce of the timed region went through + its cross-check (N > 1)
    cabi_comm = None   # the C ABI's communicator (setup_reduce), when the timed reduce goes through libpbsgpu itself
    reduce_path = "torch.distributed (pbs_plus_amd.dist.global_dedup)"
    reduce_error = None
    hard_exit = False  # a watchdog gave up on a collective: leave with os._exit once the line is out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (what `python -m torch.distributed.run
    --nproc-per-node N` would do), one per GPU, rendezvous on 127.0.0.1. Rank 0 inherits stdout (its ONE JSON line is this
    command's output), a rank that dies takes the others down, the exit code is the first non-zero one."""
    import signal
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = json.loads(os.environ["PBS_BENCH_SPAWN_CMD"]) if os.environ.get("PBS_BENCH_SPAWN_CMD") else [sys.executable] + sys.argv
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in alive:      # our own children, by PID
                        q.send_signal(signal.SIGTERM)
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and not (world == 1 and a.gpus <= 1):
        # the launcher's world size and --gpus disagree: a line that says n_gpus = WORLD_SIZE while the caller asked for
        # --gpus N would be scored against the wrong N
        print(f"bench: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    ctx = Ctx()
    backend = os.environ.get("PBS_BENCH_BACKEND", "nccl")  # "nccl" = RCCL over xGMI; gloo only for CPU tests / 1-GPU debugging
    # PBS_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, barriers, collectives, FIFO collection) even
    # with ONE rank — the only way to exercise the RCCL ("nccl") calls on a single-GPU box
    if world > 1 or os.environ.get("PBS_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ngpu0 = torch.cuda.device_count()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(ngpu0, 1))
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank % max(ngpu0, 1)}"))
            # first collective now, before the workload takes 256 GiB of the 288 GB: RCCL allocates its channel buffers
            # lazily, and a communicator that first runs out of HBM inside the timed region would take the job down
            warm = torch.zeros(1, dtype=torch.int64, device=f"cuda:{local_rank % max(ngpu0, 1)}")
            dist.all_reduce(warm, op=dist.ReduceOp.MAX)
            dist.barrier()
            torch.cuda.synchronize()
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        ctx.dist = dist
    ngpu = torch.cuda.device_count()
    if world > 1 and backend != "nccl":
        local_rank = local_rank % max(ngpu, 1)  # debug: several ranks share one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    ctx.comm_dev = dev if backend == "nccl" else torch.device("cpu")

    if a.workload == "ring":
        out = ring_run(a, rank, local_rank, world, ctx)
        if rank == 0:
            if world == 1 and not a.no_extras and (a.gib is None or a.extras_gib is not None):
                out["workloads"] = extras(_copy_args(a, workload="stream64g"), rank, local_rank, world, ctx, with_batch=True)
                mc = (out.get("cpu_baseline") or {}).get("many_core") or {}
                if mc.get("value"):
                    # the honest statement for the PCIe-fed path: how many times the host's OWN cores (all that scale) it is worth
                    out["workloads"]["gpu_over_host_cores"] = {
                        k: round(v["value"] / mc["value"], 2) for k, v in out["workloads"].items()
                        if isinstance(v, dict) and k.startswith("hostfeed") and v.get("value")}
                    out["workloads"]["gpu_over_host_cores"]["host_many_core_GiBps"] = mc["value"]
                    out["workloads"]["gpu_over_host_cores"]["host_threads"] = mc.get("cores")
            print(json.dumps(out), flush=True)
    elif a.workload in ("ring_manyfiles", "ring_corpus_dup", "ring_rechunk"):
        out = ring_files_run(a, rank, local_rank, world, ctx, a.workload[5:])
        if rank == 0:
            print(json.dumps(out), flush=True)
    elif a.workload == "hostfeed":
        outj = hostfeed_run(a, rank, local_rank, world, ctx)
        if rank == 0:
            print(json.dumps(outj), flush=True)
    elif a.workload == "verify":
        return verify_main(a, rank, local_rank, world, ctx)
    else:
        out = run_batch(a, rank, local_rank, world, ctx)
        if rank == 0:
            if (a.workload == "stream64g" and world == 1 and not a.no_extras and not a.reread
                    and (a.gib is None or a.extras_gib is not None)):
                out["workloads"] = extras(a, rank, local_rank, world, ctx)
            print(json.dumps(out), flush=True)
    if ctx.hard_exit:
        sys.stdout.flush()
        os._exit(0)
    if ctx.dist is not None:
        ctx.dist.destroy_process_group()


def _copy_args(a, **kw):
    import copy
    b = copy.copy(a)
    for k, v in kw.items():
        setattr(b, k, v)
    return b


def cu_time_budget(regime):
    """Where the CU-seconds of one GiB go: the pair service's figure is THIS run's (regime probe), scan and refill are static
    (profiles/r06_kernel_trace_bench_default.csv.gz: kernel time x the cut side's 64 CUs / bytes, one box)."""
    svc = (regime or {}).get("pair_cu_ms_per_GiB")
    scan, refill = 45, 16
    out = {"service_cu_ms_per_GiB": svc, "service_source": "measured in this run (regime.pair_cu_ms_per_GiB)",
           "scan_cu_ms_per_GiB": scan, "refill_cu_ms_per_GiB": refill,
           "scan_refill_source": "static: profiles/r06_kernel_trace_bench_default.csv.gz (round 6, one box), NOT measured in this run"}
    if svc:
        tot = svc + scan + refill
        out["chip_ceiling_GiBps"] = round(256 * 1e3 / tot)
        out["note"] = ("%.0f CU-ms per GiB over the three kernels -> 256 CUs / that = the feed phase's ceiling however the CUs are "
                       "split (refill: bench only); SHA-256 is 14 VALU instructions per round and lane in the consumer and as many "
                       "issue slots in the producer, both SIMDs of a pair busy (profiles/r06_dense_service_and_cu_split.log)" % tot)
    return out


def regime_block(ring, p0, p_fed, p1, p2, pair_cus, xp_cus):
    """roofline.regime: ns per block step of the chains under load and the shader clock they ran at, per phase of the line."""
    try:
        D = type(ring).probe_delta
        out = {"feed_phase": D(p0, p_fed) if p_fed else None, "drain": D(p_fed, p1) if p_fed else None,
               "timed_region": D(p0, p1), "single_file": D(p1, p2)}
        pr = (out["timed_region"] or {}).get("pair")
        if pr:   # a pair-service CU carries 128 chains: 128 x 64 B per block step
            out["pair_GiBps_per_cu"] = round(128 * 64 / pr["ns_per_block_step"] * 1e9 / GiB, 3)
            out["pair_cu_ms_per_GiB"] = round(1e3 / out["pair_GiBps_per_cu"], 1)
        xp = (out["timed_region"] or {}).get("express")
        if xp:   # an express CU carries 64 chunks on lane pairs
            out["express_GiBps_per_cu"] = round(64 * 64 / xp["ns_per_block_step"] * 1e9 / GiB, 3)
        out["note"] = ("ns_per_block_step: wall time per 64-byte block of one chain, averaged over every service wave that was busy "
                       "throughout a 4096-step interval (pair form floor 1.655-1.75 us alone on an idle chip; express 1.28); "
                       "sclk_mhz: shader cycles / wall time of the same intervals. A 'slow' box shows up HERE: a lower sclk_mhz "
                       "or the same clock with more ns per step (see DESIGN.md 6.3 / profiles/r06_slow_regime.log)")
        return out
    except Exception as exc:   # (a probe must never take the line down)
        return {"error": repr(exc)}


def extras(a, rank, local_rank, world, ctx, with_batch=False):
    """Short legs of the other BASELINE.json configs and of the host-fed path, folded into the DEFAULT line so that the
    driver's clock sees them too (each at its full single-GPU shape, a few steps, its own oracle check). A leg that fails
    reports the error instead of taking the headline line down."""
    import copy
    res = {}
    t_all = time.perf_counter()
    legs = [x.strip() for x in a.extras.split(",") if x.strip()]
    names = (("stream64g",) if with_batch and "batch" in legs else ()) + ("manyfiles", "corpus_dup", "rechunk")
    for name in names:
        if name not in legs and name != "stream64g":
            continue
        b = copy.copy(a)
        b.workload, b.steps, b.warmup, b.gib, b.slots, b.file_mib = name, 4, 2, a.extras_gib, None, a.extras_file_mib
        if name == "stream64g":  # the batch path on configs[1] (resident batches, batch-granular release): round 2's headline
            b.steps, b.warmup, b.gib = 8, 4, a.gib
        b.cpu_sample_gib, b.spread_points, b.brief = (0.25 if a.extras_gib is None else a.extras_gib / 8), 24, True
        t0 = time.perf_counter()
        try:
            o = run_batch(b, rank, local_rank, world, ctx)
            res["batch_path_stream64g" if name == "stream64g" else name] = {
                         "value": o["value"], "unit": o["unit"], "steps": o["steps"], "ms_per_step": o["ms_per_step"],
                         "workload": o["config"]["workload"], "resident_bytes_per_gpu": o["config"]["resident_bytes_per_gpu"],
                         "records_match_gpu": o.get("cpu_baseline", {}).get("records_match_gpu"),
                         "records_checked": o.get("cpu_baseline", {}).get("records_checked"),
                         "results": o.get("results"), "leg_seconds": round(time.perf_counter() - t0, 1)}
            if name == "stream64g":   # one 64 GiB file alone through the batch path (scan on the whole chip)
                res["batch_path_stream64g"]["single_file_ms"] = o.get("serial_step_ms", {}).get("total")
        except BaseException as exc:  # noqa: BLE001
            res["batch_path_stream64g" if name == "stream64g" else name] = {"error": repr(exc)}
    for name in ("ring_manyfiles", "ring_corpus_dup", "ring_rechunk"):
        if name not in legs or not with_batch:
            continue
        b = copy.copy(a)
        b.workload, b.steps, b.warmup, b.gib, b.file_mib = name, 6, 1, a.extras_gib, a.extras_file_mib
        b.cpu_sample_gib, b.brief = (0.25 if a.extras_gib is None else a.extras_gib / 8), True
        t0 = time.perf_counter()
        try:
            o = ring_files_run(b, rank, local_rank, world, ctx, name[5:])
            res[name] = {"value": o["value"], "unit": o["unit"], "steps": o["steps"], "ms_per_step": o["ms_per_step"],
                         "workload": o["config"]["workload"], "feed_phase": o["roofline"]["feed_phase"],
                         "service_cus": {"pair": o["config"].get("sha_service_cus"), "express": o["config"].get("express_cus")},
                         "records_match_gpu": o.get("cpu_baseline", {}).get("records_match_gpu"),
                         "records_checked": o.get("cpu_baseline", {}).get("records_checked"),
                         "results": o.get("results"), "leg_seconds": round(time.perf_counter() - t0, 1)}
        except BaseException as exc:  # noqa: BLE001
            res[name] = {"error": repr(exc)}
    # (The host-fed legs ran last for a reason until round 4: with GPU_MAX_HW_QUEUES=24 a process that had used more than ~20
    # hardware queues was time-sliced by the hardware scheduler from then on and every later leg lost 20-30 %,
    # profiles/r03_extras_leg_order.log. With 20 the effect is gone and the order is free.)
    for label, key, producers, gib_steps, archives in (("hostfeed_1_writer", "hostfeed1", 1, 96, 1),
                                                       ("hostfeed_1_writer_4_archives", "hostfeed1", 1, 96, 4),
                                                       ("hostfeed_8_writers", "hostfeed8", 8, 32, 1)):
        if key not in legs:
            continue
        b = copy.copy(a)
        b.workload, b.producers, b.steps, b.warmup, b.gib, b.tee = "hostfeed", producers, gib_steps, 4, None, False
        b.archives = archives
        t0 = time.perf_counter()
        try:
            # Each host-fed leg in a process of its own: ONE engine per process is what a host looks like (the Go agent keeps
            # one for its lifetime), and a process that has created and destroyed several engines before — as this line's
            # earlier legs do — feeds up to 15 % slower (the same leg: 42.2 GiB/s alone, 36.8 as the third engine of its
            # process, 34.6 as the eighth; profiles/r05_hostfeed_leg_isolation.log). PBS_BENCH_HF_INPROC=1: in this process.
            if os.environ.get("PBS_BENCH_HF_INPROC"):
                o = hostfeed_run(b, rank, local_rank, world, ctx)
            else:
                o = _hostfeed_subprocess(b, local_rank)
            res[label] = {"value": o["value"], "unit": o["unit"], "producers": producers, "archives_per_producer": archives,
                          "bytes": int(o["config"]["bytes"]),
                          "frac_of_measured_h2d": o["roofline"]["frac_of_measured_h2d"],
                          "measured_h2d_GBps": o["roofline"]["measured_h2d_GBps"],
                          "write_phase": {k: o["write_phase"][k] for k in ("seconds", "GiBps", "drain_seconds")},
                          "write_phase_frac_of_measured_h2d": None if not o["roofline"]["measured_h2d_GBps"] else round(
                              o["write_phase"]["GiBps"] * GiB / 1e9 / o["roofline"]["measured_h2d_GBps"], 3),
                          "records_match_oracle": o["stream_records_match_oracle"],
                          "in_subprocess": not os.environ.get("PBS_BENCH_HF_INPROC"),  # (since round 5: a fresh process per leg)
                          "leg_seconds": round(time.perf_counter() - t0, 1)}
        except BaseException as exc:  # noqa: BLE001
            res[label] = {"error": repr(exc)}
    res["total_seconds"] = round(time.perf_counter() - t_all, 1)
    return res


def _hostfeed_subprocess(b, local_rank):
    """One host-fed leg as `bench.py --workload hostfeed ...` in a fresh process on the same GPU; returns its JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "hostfeed", "--gpus", "1", "--producers", str(b.producers),
           "--steps", str(b.steps), "--warmup", str(b.warmup), "--archives", str(b.archives), "--avg", str(b.avg),
           "--seed", str(b.seed), "--no-cpu-baseline"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PBS_BENCH_FORCE_DIST"):
        env.pop(k, None)
    # the child sees ONE device: the parent rank's (the local_rank-th entry of the parent's own mask, if it has one)
    mask = [x for x in env.get("HIP_VISIBLE_DEVICES", "").split(",") if x.strip()]
    if mask:
        env["HIP_VISIBLE_DEVICES"] = mask[local_rank] if local_rank < len(mask) else mask[-1]
    elif local_rank:
        env["HIP_VISIBLE_DEVICES"] = str(local_rank)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if os.environ.get("PBS_BENCH_HF_TRACE"):
        sys.stderr.write(r.stderr)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            return json.loads(line)
    raise RuntimeError(f"host-fed leg failed (rc {r.returncode}): {r.stderr[-400:]}")


def setup_reduce(ctx, eng):
    """Which digest-set reduce the TIMED region uses. Since round 6: the C ABI's own communicator (pbsgpu_comm_create +
    pbsgpu_digest_allgather_dedup: libpbsgpu dlopens RCCL, ONE ncclAllGather of [count | records] slots + device dedup — what a
    Go host binds, no torch in it) whenever the ranks sit on their own GPUs (backend nccl); the torch.distributed path
    (pbs_plus_amd.dist.global_dedup) otherwise — gloo: CPU tests, several ranks sharing one GPU, where RCCL cannot build a
    communicator. PBS_BENCH_REDUCE=torch|cabi forces one. Collective; a watchdog keeps a communicator that does not come up
    from taking the line down (every rank then falls back together)."""
    ctx.cabi_comm, ctx.reduce_path = None, "torch.distributed (pbs_plus_amd.dist.global_dedup)"
    if ctx.dist is None:
        return
    want = os.environ.get("PBS_BENCH_REDUCE", "torch" if os.environ.get("PBS_BENCH_NO_CABI_REDUCE") else "auto")
    if want == "torch" or (want == "auto" and (ctx.comm_dev is None or ctx.comm_dev.type != "cuda")):
        return
    from pbs_plus_amd.dist import make_comm
    res = {}

    def work():
        try:
            res["comm"] = make_comm(eng, device=ctx.comm_dev)
        except BaseException as exc:  # noqa: BLE001
            res["error"] = repr(exc)

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(120)
    ok = 0 if (th.is_alive() or "comm" not in res) else 1
    if th.is_alive():
        ctx.hard_exit = True
    import torch
    flag = torch.tensor([ok], dtype=torch.int64, device=ctx.comm_dev)
    ctx.dist.all_reduce(flag, op=ctx.dist.ReduceOp.MIN)          # every rank or none
    if int(flag.item()) == 1:
        ctx.cabi_comm = res["comm"]
        ctx.reduce_path = ("C ABI: pbsgpu_comm_create + pbsgpu_digest_allgather_dedup (libpbsgpu dlopens RCCL: one ncclAllGather of "
                           "[count | records] slots + device dedup)")
    else:
        ctx.reduce_error = res.get("error", "timeout after 120 s")
        if "comm" in res:
            res["comm"].close()


def reduce_records(ctx, eng, recs, cap_records=None):
    """the digest-set reduce of the timed region (collective): statistics of the union over all ranks"""
    cap = ctx.rec_cap if cap_records is None else cap_records
    if getattr(ctx, "cabi_comm", None) is not None:
        _, stats = ctx.cabi_comm.dedup(recs, cap, want_flags=False)
        return stats
    from pbs_plus_amd.dist import global_dedup
    _, stats, _ = global_dedup(eng, recs, device=ctx.comm_dev, cap_records=cap, want_records=False)
    return stats


def reduce_cross_check(ctx, eng, recs):
    """outside the timed region: the OTHER path over the same records (collective) — both must report the same union"""
    if ctx.dist is None:
        return None
    from pbs_plus_amd.dist import global_dedup
    keys = ("nrecords", "nunique", "total_bytes", "unique_bytes")
    out = {"timed_path": ctx.reduce_path, "timed_through_c_abi": getattr(ctx, "cabi_comm", None) is not None}
    if getattr(ctx, "reduce_error", None):
        out["c_abi_error"] = ctx.reduce_error
    try:
        t0 = time.perf_counter()
        mine = reduce_records(ctx, eng, recs)
        out["ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        out["stats"] = {k: int(mine[k]) for k in keys}
        if out["timed_through_c_abi"]:
            _, ref, _ = global_dedup(eng, recs, device=ctx.comm_dev, cap_records=ctx.rec_cap, want_records=False)
            out["equals_torch_path"] = bool(all(int(mine[k]) == int(ref[k]) for k in keys))
        out["ok"] = True
    except BaseException as exc:  # noqa: BLE001
        out.update(ok=False, error=repr(exc))
    return out


def ring_run(a, rank, local_rank, world, ctx):
    """BASELINE.json configs[1] through the PAGE RING (pbsgpu_ring_*): `--ring-streams` 64 GiB files in flight at once,
    a STEP = one file completely ingested (its final record delivered). Every file is new data (its own seed): pages
    are refilled by the generator as soon as the SHA-256 service has read them — nothing is resident longer than the
    chunks that touch its page take to hash. Timed region = K files from an idle ring to an idle ring (service stopped)."""
    import pbs_plus_amd
    from pbs_plus_amd import buzhash

    cfg = buzhash.NewConfig(a.avg)
    eng = pbs_plus_amd.Engine(cfg, device=local_rank, inflight=1)
    file_bytes = int((64.0 if a.gib is None else a.gib) * GiB) & ~15
    arena = 0 if a.arena_gib is None else int(a.arena_gib * GiB)
    ring = pbs_plus_amd.PageRing(eng, arena_bytes=arena, max_streams=max(8, a.ring_streams) * 4, sha_cus=a.ring_sha_cus,
                                 round_pages=a.ring_round_pages)
    state = {"S": max(1, a.ring_streams), "next_file": 0, "timed": False, "first_timed": 0, "next_reduce": 0}
    # pages a stream may take per turn: a quarter of a round with 4 files in flight (steady state is bounded by the pages that
    # come back, not by this; at the start of a region it lets the first rounds be full ones, which the ring cuts ahead of the
    # service at full chip width)
    quota = max(16, 256 // max(1, a.ring_streams)) * int(ring.page_bytes)
    max_open = max(8, a.ring_streams) * 4
    kind = a.ring_kind
    pending_reduce, extra, marks = {}, {}, {}

    def seed_of(fidx):
        return a.seed + 1000 * rank + 7919 * fidx

    def reduce_in_order(step, recs):
        # with several ranks every rank issues its collectives in the same sequence: step k's digest-set reduce goes out
        # once steps 0..k-1 have been reduced, whatever order the files finished in on this rank
        if ctx.dist is None or not state["timed"]:
            return
        pending_reduce[step] = recs
        while state["next_reduce"] in pending_reduce:
            r = pending_reduce.pop(state["next_reduce"])
            stats = reduce_records(ctx, eng, r)
            extra["dedup_last_step"] = {k: int(v) for k, v in stats.items()}
            state["next_reduce"] += 1

    trace_free_all = []
    trace_done, delivered = [], [0, 0, 0]

    def run_files(nfiles, keep):
        """nfiles whole files through the ring, S at a time; returns {file index: records} of the kept ones"""
        out, active, opened, done = {}, {}, 0, 0
        marks.clear()
        loopstat = [0, 0, 0]   # feeder loop: iterations, fill calls, fill calls that got no page (arena empty or backlog gate)
        marks["loopstat"] = loopstat
        del trace_free_all[:]
        if state["timed"] and nfiles > 1:
            del trace_done[:]
            delivered[:] = [0, 0, 0]
        trace_free = trace_free_all if (os.environ.get("PBS_BENCH_RING_TRACE") and state["timed"] and nfiles > 1) else None
        base = state["next_file"]
        enq0 = ring.stats()["bytes_enqueued"]
        t_last = time.perf_counter()
        while done < nfiles:
            # S files are being FED at any time; a file whose bytes are all in (its last chunks still hashing, up to ~0.45 s)
            # does not hold a feeding slot — otherwise every generation of S files would end with an idle feed
            while opened < nfiles and sum(1 for st in active.values() if st[1]) < state["S"] and len(active) < max_open:
                sid = ring.open()
                active[sid] = [base + opened, file_bytes, []]
                opened += 1
            # free pages are dealt out evenly: a stream gets at most `quota` pages per turn, so every round carries pages
            # of all the files in flight (one stream taking every free page would feed the files one after the other)
            # (a file that is ALONE gets a whole round's worth per turn: with 16 pages per turn one 64 GiB file went through
            # 253 small rounds and ~100 ms of per-round latency — and never qualified for the ring's lone-stream cut-ahead)
            q_turn = quota if (state["S"] > 1 or len(active) > 1) else 256 * int(ring.page_bytes)
            loopstat[0] += 1
            for sid, st in active.items():
                if st[1]:
                    want = min(st[1], q_turn)
                    got = ring.fill(sid, seed_of(st[0]), kind, want, final=(want == st[1]))
                    st[1] -= got
                    loopstat[1] += 1
                    loopstat[2] += 1 if got == 0 else 0
            ring.pump()
            if trace_free is not None and loopstat[0] % 128 == 0 and "t_fed" not in marks:    # diagnostic: what refuses the feeder, the gate or the arena?
                trace_free.append(ring.stats()["pages_free"])
            if (opened == nfiles and "t_fed" not in marks and not any(st[1] for st in active.values())
                    and ring.stats()["bytes_enqueued"] >= enq0 + nfiles * file_bytes):
                marks["t_fed"] = time.perf_counter()     # every byte of the last file is in a cut round
                marks["loopstat_fed"] = list(loopstat)
                if os.environ.get("PBS_BENCH_RING_DEBUG"):
                    marks["raw_fed"] = ring.debug()
                marks["probe_fed"] = ring.probe()
                if os.environ.get("PBS_BENCH_RING_TRACE") == "2":   # queue state at the end of the feed phase (diagnostic; pbsgpu_ring_debug
                    # copies on the null stream and was seen to stall beside the services: profiles/r06_lanes_service.log)
                    marks["fed_state"] = (ring.debug().splitlines()[0], ring.stats())
            for sid in list(active):
                recs, fin = ring.poll(sid, 8192)
                if recs.size:
                    active[sid][2].append(recs.copy())
                    t_last = time.perf_counter()
                    if trace_free is not None:   # diagnostic: bytes of delivered records against time (the services' real rate)
                        delivered[0] += int(recs["size"].sum())
                        tier = recs["segment"] >> 28   # (PBSGPU_RING_TIER_TAG=1: 0 main queue, 1 long, 2 short; else all 0)
                        for k_ in (1, 2):
                            delivered[k_] += int(recs["size"][tier == k_].sum())
                        trace_done.append((t_last, delivered[0], delivered[1], delivered[2]))
                if fin:
                    fidx, _, parts = active.pop(sid)
                    ring.close_stream(sid)
                    allr = np.concatenate(parts) if parts else np.zeros(0, dtype=pbs_plus_amd.RECORD_DTYPE)
                    allr["segment"] = 0          # (the ring reports its stream id there; a file is one segment)
                    if keep:
                        out[fidx] = allr
                    reduce_in_order(fidx - state["first_timed"], allr)
                    done += 1
            if time.perf_counter() - t_last > 60:
                raise SystemExit(f"ring made no progress for 60 s: {ring.stats()}\n{ring.debug() if hasattr(ring, 'debug') else ''}")
        state["next_file"] = base + nfiles
        return out

    ctx.rec_cap = file_bytes // max(cfg.MinSize, 65) + 64
    if ctx.dist is not None:
        rc = torch.tensor([ctx.rec_cap], dtype=torch.int64, device=ctx.comm_dev)
        ctx.dist.all_reduce(rc, op=ctx.dist.ReduceOp.MAX)
        ctx.rec_cap = int(rc.item())
    if ctx.dist is not None:
        # first contact of the digest-set reduce BEFORE the persistent service starts: RCCL's lazy channel buffers, torch's
        # communication tensors and the engine's dedup work buffers all take their final size here (at the agreed record
        # capacity), so nothing inside the timed region allocates, frees or waits for the whole device
        setup_reduce(ctx, eng)
        dummy = np.zeros(ctx.rec_cap, dtype=pbs_plus_amd.RECORD_DTYPE)
        dummy["digest"][:, :8] = np.arange(ctx.rec_cap, dtype=np.uint64).view(np.uint8).reshape(-1, 8)
        dummy["size"] = 1
        reduce_records(ctx, eng, dummy)
        ctx.dist.barrier()
    if a.warmup:
        run_files(a.warmup, False)
    ring.quiesce()
    if ctx.dist is not None:
        ctx.dist.barrier()
    torch.cuda.synchronize()
    state["timed"], state["first_timed"] = True, state["next_file"]
    st0 = ring.stats()
    probe0 = ring.probe()
    raw0 = ring.debug() if os.environ.get("PBS_BENCH_RING_DEBUG") else None
    t0 = time.perf_counter()
    kept = run_files(a.steps, True)
    t_fed = marks.get("t_fed", None)
    probe_fed = marks.get("probe_fed", None)
    if os.environ.get("PBS_BENCH_RING_DEBUG") and rank == 0 and t_fed and "loopstat_fed" in marks:
        it, nf, nz = marks["loopstat_fed"]
        print("[feeder] feed phase %.3f s: %d loop iterations (%.0f us each), %d fill calls, %d of them got no page (%.1f %%)"
              % (t_fed - t0, it, (t_fed - t0) * 1e6 / max(it, 1), nf, nz, 100.0 * nz / max(nf, 1)), file=sys.stderr, flush=True)
    if raw0 and "raw_fed" in marks and rank == 0:   # lane occupancy of the probe waves in the feed phase (pbsgpu_ring_debug's raw counters)
        import re as _re

        def _raw(t):
            m = _re.search(r"probe raw: pair_steps=(\d+) pair_active=(\d+) lanes_steps=(\d+) lanes_active=(\d+)", t)
            return [int(x) for x in m.groups()] if m else None
        a0, a1 = _raw(raw0), _raw(marks["raw_fed"])
        if a0 and a1:
            d = [y - x for x, y in zip(a0, a1)]
            print("[occupancy] feed phase: pair probe wave %.3f of its lanes busy (%d steps)%s" % (
                d[1] / (64.0 * d[0]) if d[0] else 0.0, d[0],
                ", lanes-service probe wave %.3f (%d steps)" % (d[3] / (64.0 * d[2]), d[2]) if d[2] else ""), file=sys.stderr, flush=True)
    if trace_free_all and rank == 0:
        trace_free = list(trace_free_all)
        tf = np.array(trace_free[len(trace_free) // 4:], dtype=np.float64)   # the steady state: without the first quarter
        print("[ring trace] free pages seen by the feeder (every 128th loop iteration, last 3/4 of the run): min %d median %d mean %.0f "
              "max %d of %d; share of samples with fewer than 64 free: %.3f" % (tf.min(), np.median(tf), tf.mean(), tf.max(),
              ring.stats()["pages_total"], float((tf < 64).mean())), file=sys.stderr, flush=True)
    if trace_done and t_fed and rank == 0:
        # slope of delivered bytes over the steady state: from 35 % of the feed phase to its end (records are delivered in stream
        # order, so the curve lags the hashing by up to one long chain — a constant lag in the steady state)
        td = np.array(trace_done, dtype=np.float64)
        a_, b_ = t0 + 0.35 * (t_fed - t0), t_fed
        m_ = (td[:, 0] >= a_) & (td[:, 0] <= b_)
        if m_.sum() > 8:
            sl = [np.polyfit(td[m_, 0], td[m_, c_], 1)[0] / GiB for c_ in (1, 2, 3)]
            print("[ring trace] delivered records, slope over the last 65 %% of the feed phase: %.1f GiB/s (%d samples); by queue "
                  "(PBSGPU_RING_TIER_TAG): main %.1f, long %.1f, short %.1f" % (sl[0], int(m_.sum()), sl[0] - sl[1] - sl[2], sl[1], sl[2]),
                  file=sys.stderr, flush=True)
    if "fed_state" in marks and rank == 0:
        print("[ring trace] at end of feed:", marks["fed_state"][0], "| pages_free", marks["fed_state"][1]["pages_free"],
              "of", marks["fed_state"][1]["pages_total"], file=sys.stderr, flush=True)
    ring.quiesce()
    torch.cuda.synchronize()
    if ctx.dist is not None:
        ctx.dist.barrier()
    elapsed = time.perf_counter() - t0
    st1 = ring.stats()
    probe1 = ring.probe()
    total_bytes = float(a.steps) * file_bytes
    if ctx.dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctx.comm_dev)
        ctx.dist.all_reduce(tt, op=ctx.dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tb = torch.tensor([total_bytes], dtype=torch.float64, device=ctx.comm_dev)
        ctx.dist.all_reduce(tb, op=ctx.dist.ReduceOp.SUM)
        total_bytes = float(tb.item())
    state["timed"] = False
    if ctx.dist is not None:
        # the reduce of the timed region once more on the last timed file of every rank, and (when it went through the C ABI)
        # the torch.distributed path beside it: both must report the same union
        ctx.cabi = reduce_cross_check(ctx, eng, kept[max(kept)])
    # one file alone through an idle ring: single-file latency (outside the timed region)
    ts0 = time.perf_counter()
    s_saved, state["S"] = state["S"], 1
    single = run_files(1, True)
    single_fed_s = marks.get("t_fed", ts0) - ts0     # every byte of the file in a cut round (then: the chains of its last chunks)
    ring.quiesce()
    single_s = time.perf_counter() - ts0
    probe2 = ring.probe()
    if os.environ.get("PBS_BENCH_RING_DEBUG"):
        dbg = ring.debug().splitlines()
        sys.stderr.write("[ring debug] " + dbg[1] + "\n")
        for ln in dbg:
            if ln.startswith("control kernel"):
                sys.stderr.write("[ring debug] " + ln + "\n")
    state["S"] = s_saved
    out = None
    if rank == 0:
        value = total_bytes / GiB / elapsed
        gbs = total_bytes / elapsed / 1e9 / world
        svc_ms = float(st1["service_ms_last"])
        svc_bytes = float(st1["service_bytes_last"])
        svc_gbs = svc_bytes / max(svc_ms, 1e-9) / 1e6
        tr = load_traffic_ring()
        recs0 = kept[min(kept)]
        # which regime did THIS run land in? (pbsgpu_ring_get_probe)
        regime = regime_block(ring, probe0, probe_fed, probe1, probe2, int(st1["sha_cus"]), ring.express()[0])
        out = {
            "metric": "GiB/s ingested through CDC+SHA-256",
            "value": round(value, 2), "unit": "GiB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32",
            "data": "synthetic (generated on device page by page as pages come free: every file is new bytes, seed per file; "
                    "generator kind %d)" % kind,
            "config": {
                "workload": f"single {file_bytes / GiB:g} GiB random stream per step, {state['S']} files in flight through the "
                            f"page ring, Buzhash CDC avg {a.avg >> 20} MiB (min avg/4, max 4*avg) + SHA-256 per chunk "
                            f"(BASELINE.json configs[1])",
                "path": "page ring: page-granular memory release + persistent cross-stream SHA-256 service (pbsgpu_ring_*)",
                "bytes_per_step": file_bytes, "files_in_flight": state["S"], "avg_chunk": a.avg,
                "arena_pages": int(st1["pages_total"]), "page_bytes": int(st1["page_bytes"]),
                "resident_bytes_per_gpu": int(st1["pages_total"]) * int(st1["page_bytes"]),
                "sha_service_cus": int(st1["sha_cus"]), "express_cus": ring.express()[0], "express_from_bytes": ring.express()[1],
                "chunks_per_step": int(recs0.size),
                "rounds_in_timed_region": int(st1["rounds"] - st0["rounds"]),
                "distinct_data_per_step": True,
                "parallelism": (f"{world} rank(s), one per GPU, files independent, digest-set all-gather per file"
                                if world > 1 else "1 GPU")},
            "roofline": {
                "kernel": "k_sha256_pair<RingSource,false> (the persistent SHA-256 service: ONE launch spans the timed region, the "
                          "express service k_sha256_xpair<RingSource> beside it on its own CUs; the cut rounds — k_ring_stage, "
                          "k_ring_fill, k_scan3, k_ring_control — run on the CUs the services leave free)",
                # the contract's roofline: algorithmic bytes of the dominant kernel / its duration against the HBM peak
                # (BASELINE.json quotes % of the HBM roofline). What actually bounds SHA-256 on this chip — integer issue
                # slots, and the serial chain inside a chunk — is in `valu` beside it.
                "bound": "hbm", "achieved": round(svc_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(svc_gbs / HBM_PEAK_GBS, 4),
                "achieved_note": "bytes hashed by the service launch (1 algorithmic byte per input byte, SURVEY.md 8d) / its "
                                 "duration, HIP events on the service's own stream (launch -> end of kernel); the launch starts "
                                 "with the first round of the timed region and ends when the last chunk is hashed",
                "service_launch_ms": round(svc_ms, 3), "service_launch_bytes": int(svc_bytes),
                "valu": {"peak": round(SHA_VALU_GBS, 1), "frac": round(svc_gbs / round(SHA_VALU_GBS, 1), 4), "unit": "GB/s",
                         "note": "the ceiling that binds first: %.1f T integer lane-ops/s (620 G wave64 VOP3 instr/s measured, "
                                 "profiles/r01_ubench_int_valu_issue.log) / %.1f ops per byte of SHA-256; HBM would allow 4.6x more"
                                 % (VALU_PEAK_TOPS, SHA_OPS_PER_BYTE)},
                "feed_phase": None if t_fed is None else {
                    "seconds": round(t_fed - t0, 4), "GiBps": round(a.steps * file_bytes / GiB / max(t_fed - t0, 1e-9), 1),
                    "drain_seconds": round(elapsed - (t_fed - t0), 4),
                    "note": "timed region = feed phase (idle ring -> last page of the last file in a cut round; pages are cut, "
                            "hashed and recycled all the while) + drain (the last chunks' serial SHA-256 chains with no new work — "
                            "a max-size chunk on an express pair ~0.34 s, a chunk just under the express threshold on a pair lane "
                            "~0.37 s: a fixed cost per timed region whatever its length)"},
                "path": {"achieved": round(gbs, 1), "frac_of_valu_peak": round(gbs / SHA_VALU_GBS, 4),
                         "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                         "note": "whole timed region incl. ramp-up from an idle ring and the drain of the last chunks"},
                # which regime did THIS run land in? (pbsgpu_ring_get_probe: every service wave samples the shader clock and the
                # wall clock once per 4096 block steps; only intervals in which the wave carried a block in every step count)
                "regime": regime,
                "cu_time_budget": cu_time_budget(regime),
                "traffic": None if tr is None else int(tr["ratio"] * svc_bytes),
                "traffic_note": None if tr is None else tr["note"],
                "algorithmic_bytes_per_launch": int(svc_bytes),
                "single_file": {"ms": round(single_s * 1e3, 1), "GiBps": round(file_bytes / GiB / single_s, 1),
                                "cut_ms": round(single_fed_s * 1e3, 1),
                                "note": "one file alone through an idle ring, first page filled to last record delivered"},
            },
            "serial_value": round(file_bytes / GiB / single_s, 2),
        }
        if extra:
            out["results"] = extra
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = ring_cpu_baseline(a, kept, single, file_bytes, kind, seed_of)
    if not a.no_cpu_baseline and world > 1:
        # N ranks: EVERY rank checks its own files against the oracle (prefix + restart points, regenerated from the files'
        # seeds), the verdicts meet on rank 0 — an N-GPU line carries parity evidence for all N GPUs' records. The one-core
        # CPU figure is rank 0's (the contract times it at N = 1; here it is context only).
        mine = ring_cpu_baseline(_copy_args(a, brief=True, spread_points=max(8, a.spread_points // 2)), kept, single, file_bytes,
                                 kind, seed_of)
        v = torch.tensor([1 if mine["records_match_gpu"] else 0, -int(mine["records_checked"]),
                          -int(mine["whole_file_restart_points"]["files"])], dtype=torch.int64, device=ctx.comm_dev)
        lo = v.clone()
        ctx.dist.all_reduce(lo, op=ctx.dist.ReduceOp.MIN)          # all ranks matched <=> the minimum is 1
        ctx.dist.all_reduce(v, op=ctx.dist.ReduceOp.SUM)
        if rank == 0:
            mine["all_ranks"] = {"ranks": world, "records_match_gpu": bool(int(lo[0].item()) == 1),
                                 "records_checked": int(-v[1].item()), "files_checked": int(-v[2].item())}
            mine["records_match_gpu"] = mine["all_ranks"]["records_match_gpu"]
            mine["records_checked"] = mine["all_ranks"]["records_checked"]
            out["cpu_baseline"] = mine
    if ctx.dist is not None and rank == 0 and ctx.cabi is not None:
        out.setdefault("results", {})["c_abi_digest_reduce"] = ctx.cabi
    if getattr(ctx, "cabi_comm", None) is not None:
        ctx.cabi_comm.close()
        ctx.cabi_comm = None
    ring.close()
    eng.close()
    return out


def ring_files_run(a, rank, local_rank, world, ctx, mode):
    """BASELINE.json configs[2] / configs[3] through the PAGE RING: every 64 MiB file (or corpus segment) is its own ring
    stream — the many-small-stream packing the ring exists for. A step = `files_per_step` files (128 GiB); the files of
    all timed steps flow through the ring back to back (pages are recycled as they are read), then the ring drains once.
    mode "manyfiles": entropy class = file % 4; mode "corpus_dup": 40 % of the segments are exact copies of an earlier
    one (the copy carries the original's generator seed), duplicate bytes found by the device dedup vs planted."""
    import pbs_plus_amd
    from pbs_plus_amd import buzhash

    cfg = buzhash.NewConfig(a.avg)
    eng = pbs_plus_amd.Engine(cfg, device=local_rank, inflight=1)
    fbytes = int(a.file_mib * MiB) & ~15
    per_step = max(1, int((128.0 if a.gib is None else a.gib) * GiB) // fbytes)
    arena = 0 if a.arena_gib is None else int(a.arena_gib * GiB)
    feeding = max(8, a.ring_streams * 8)
    ring = pbs_plus_amd.PageRing(eng, arena_bytes=arena, max_streams=4096, sha_cus=a.ring_sha_cus, round_pages=a.ring_round_pages)
    total_files = per_step * (a.steps + a.warmup)
    root = dup_roots(per_step * a.steps, a.seed + 4) if mode == "corpus_dup" else None
    edited = {"on": False}          # rechunk: False while the BASE snapshot is ingested, True for the edited corpus
    plans = {}

    def spec(g):
        """(seed, kind) of global file index g (g < 0: warm-up files)"""
        if mode == "corpus_dup" and g >= 0:
            return a.seed + 104729 * int(root[g]) + 5 + 1000003 * rank, 0
        if mode == "rechunk":
            return a.seed + 15485863 * (g + 10 ** 6 * (rank + 1)) + 9, 4
        return a.seed + 7919 * (g + 10 ** 6 * (rank + 1)) + 1, ((g % 4) if mode == "manyfiles" else 0)

    def plan_of(g):
        """rechunk: (piece table rows (dst_off, len, src_off, seed), edited length) of file g — ~2 % of its bytes edited in
        log-uniform 4 KiB-4 MiB extents, 1/3 overwrite / insert / delete; kept extents come from the base file's generator
        stream, new bytes from a second stream of the file"""
        if g not in plans:
            base_seed = spec(g)[0]
            new_seed = base_seed ^ 0x5DEECE66D
            rng = np.random.default_rng(a.seed + 5 + 977 * rank + 31 * (g + 10 ** 6))
            rows, pos, npos = [], 0, 0
            for kind, so, ln in edit_plan(fbytes, rng, align=16):
                if kind == 0:
                    rows.append((pos, ln, so, base_seed))
                else:
                    rows.append((pos, ln, npos, new_seed))
                    npos += ln
                pos += ln
            plans[g] = (np.array(rows, dtype=np.uint64).reshape(-1, 4), pos)
        return plans[g]

    def host_bytes(g):
        """the file's bytes rebuilt on the host (the oracle's twin of the device generator)"""
        from oracle import oracle as O
        if mode == "rechunk" and edited["on"]:
            rows, n = plan_of(g)
            out = np.empty(n, dtype=np.uint8)
            for dst, ln, so, seed in rows:
                O.fill(int(ln), int(seed), 4, stream_off=int(so), out=out[int(dst):int(dst + ln)])
            return out
        seed, kind = spec(g)
        return O.fill(fbytes, seed, kind)

    quota = 4 * int(ring.page_bytes)
    recs_of = {}

    def run(first, count, keep):
        nxt, done, active, t_last = first, 0, {}, time.perf_counter()
        marks.clear()
        enq0 = ring.stats()["bytes_enqueued"]
        fed = {"bytes": 0}
        while done < count:
            nfeed = sum(1 for st in active.values() if st[1])
            while nxt < first + count and nfeed < feeding and len(active) < 4000:
                sid = ring.open()
                active[sid] = [nxt, plan_of(nxt)[1] if edited["on"] else fbytes, [], True]
                fed["bytes"] += active[sid][1]
                nxt += 1
                nfeed += 1
            for sid, st in active.items():
                if st[1]:
                    want = min(st[1], quota)
                    if edited["on"]:
                        st[1] -= ring.fill_pieces(sid, plan_of(st[0])[0] if st[3] else None, want, final=(want == st[1]))
                        st[3] = False
                    else:
                        seed, kind = spec(st[0])
                        st[1] -= ring.fill(sid, seed, kind, want, final=(want == st[1]))
            ring.pump()
            if (nxt == first + count and "t_fed" not in marks and not any(st[1] for st in active.values())
                    and ring.stats()["bytes_enqueued"] >= enq0 + fed["bytes"]):
                marks["t_fed"] = time.perf_counter()
            recs, fin = ring.poll_any()
            if recs.size:
                t_last = time.perf_counter()
                if keep:
                    for sid in np.unique(recs["segment"]):
                        active[int(sid)][2].append(recs[recs["segment"] == sid])
            for sid in fin:
                g, _, parts, _ = active.pop(int(sid))
                ring.close_stream(int(sid))
                if keep:
                    r = np.concatenate(parts) if parts else np.zeros(0, dtype=pbs_plus_amd.RECORD_DTYPE)
                    r["segment"] = 0
                    recs_of[g] = r
                done += 1
            if time.perf_counter() - t_last > 60:
                raise SystemExit(f"ring made no progress for 60 s: {ring.stats()}\n{ring.debug() if hasattr(ring, 'debug') else ''}")

    marks = {}
    base_digests = None
    if mode == "rechunk":
        # the PREVIOUS snapshot: the unedited files of every timed step through the same ring, outside the timed region;
        # its digest set is what the edited corpus' chunks are looked up in (re-used chunk fraction)
        run(0, per_step * a.steps, True)
        ring.quiesce()
        base_digests = set()
        for r in recs_of.values():
            base_digests.update(map(bytes, r["digest"]))
        recs_of.clear()
        edited["on"] = True
    if a.warmup:
        run(-per_step * a.warmup, per_step * a.warmup, False)
    ring.quiesce()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(0, per_step * a.steps, True)
    t_fed = marks.get("t_fed")
    ring.quiesce()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st1 = ring.stats()
    total_bytes = float(sum(plan_of(g)[1] for g in range(per_step * a.steps))) if mode == "rechunk" else float(per_step * a.steps) * fbytes
    out = {
        "metric": "GiB/s ingested through CDC+SHA-256", "value": round(total_bytes / GiB / elapsed, 2), "unit": "GiB/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32",
        "data": "synthetic (generated on device page by page; one ring stream per file)",
        "config": {"workload": (f"{per_step} x {fbytes / MiB:g} MiB files per step through the page ring, one stream per file "
                                + ("(entropy class = file % 4; BASELINE.json configs[2])" if mode == "manyfiles" else
                                   "(every file after byte edits totalling 2 %: log-uniform 4 KiB-4 MiB extents, 1/3 overwrite / "
                                   "insert / delete, produced page by page from its piece table; the unedited snapshot went "
                                   "through the ring before the timed region; BASELINE.json configs[4])" if mode == "rechunk" else
                                   "(40 % of the segments exact copies of an earlier one, device dedup over the pass; "
                                   "BASELINE.json configs[3])")),
                   "files_per_step": per_step, "file_bytes": fbytes, "streams_feeding": feeding,
                   "resident_bytes_per_gpu": int(st1["pages_total"]) * int(st1["page_bytes"]),
                   # the split between the pair and the express service follows the share of bytes in long chunks the ring
                   # has seen (decided when a service starts: here after the warm-up step)
                   "sha_service_cus": int(st1["sha_cus"]), "express_cus": ring.express()[0]},
        "roofline": {"kernel": "k_sha256_pair<RingSource,false>", "bound": "valu",
                     "achieved": round(float(st1["service_bytes_last"]) / max(float(st1["service_ms_last"]), 1e-9) / 1e6, 1),
                     "peak": round(SHA_VALU_GBS, 1), "unit": "GB/s",
                     "frac": round(float(st1["service_bytes_last"]) / max(float(st1["service_ms_last"]), 1e-9) / 1e6 / SHA_VALU_GBS, 4),
                     "traffic": None,
                     "feed_phase": None if t_fed is None else {"seconds": round(t_fed - t0, 4),
                                                               "GiBps": round(total_bytes / GiB / max(t_fed - t0, 1e-9), 1),
                                                               "drain_seconds": round(elapsed - (t_fed - t0), 4)}},
    }
    if mode == "corpus_dup":
        allr = np.concatenate([recs_of[g] for g in sorted(recs_of)])
        _, stats = eng.dedup(allr)
        _, first = np.unique(root, return_index=True)
        tb = max(int(stats["total_bytes"]), 1)
        out["results"] = {"dedup": {"records": int(stats["nrecords"]), "unique": int(stats["nunique"]),
                                    "duplicate_bytes_frac": round(1.0 - int(stats["unique_bytes"]) / tb, 4),
                                    "expected_duplicate_frac": round(1.0 - len(first) / len(root), 4)}}
    if mode == "rechunk":
        reused = tot = 0
        for r in recs_of.values():
            hit = np.fromiter((bytes(x) in base_digests for x in r["digest"]), dtype=bool, count=r.size)
            reused += int(r["size"][hit].sum())
            tot += int(r["size"].sum())
        out["results"] = {"reused_chunk_bytes_frac": round(reused / max(tot, 1), 4), "edited_bytes": int(tot),
                          "base_snapshot_chunks": len(base_digests)}
    if not a.no_cpu_baseline:
        # oracle on whole files sampled uniformly over the timed files (regenerated from their seeds), 32 threads
        from oracle import oracle as O
        O.build()
        ocfg = O.new_config(a.avg)
        nsamp = max(4, min(len(recs_of), int(a.cpu_sample_gib * GiB * 8 / fbytes)))
        ids = sorted(recs_of)[:: max(1, len(recs_of) // nsamp)][:nsamp]
        res, lock = {"ok": True, "records": 0, "bad": None}, threading.Lock()

        def check(g):
            hb = host_bytes(g)
            w = O.chunk_and_digest(ocfg, hb, [(0, hb.size)], impl=1)
            r = recs_of[g]
            same = bool(r.size == w.size and np.array_equal(r["end"], w["end"]) and np.array_equal(r["digest"], w["digest"]))
            with lock:
                res["records"] += int(w.size)
                if not same and res["ok"]:
                    res["ok"], res["bad"] = False, int(g)

        t1 = time.perf_counter()
        it, itl = iter(ids), threading.Lock()

        def worker():
            while True:
                with itl:
                    g = next(it, None)
                if g is None:
                    return
                check(g)
        ths = [threading.Thread(target=worker) for _ in range(max(1, min(32, os.cpu_count() or 1)))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t1
        flen = (lambda g: plan_of(g)[1]) if mode == "rechunk" else (lambda g: fbytes)
        tiles = all(int(r["end"][-1]) == flen(g) and int(r["size"].astype(np.int64).sum()) == flen(g) for g, r in recs_of.items())
        out["cpu_baseline"] = {"value": round(len(ids) * fbytes / GiB / dt, 3), "unit": "GiB/s", "cores": len(ths), "kind": "port",
                               "sample": f"{len(ids)} whole files spread over the {len(recs_of)} timed files, regenerated from "
                                         f"their seeds, oracle chunk_and_digest on {len(ths)} threads",
                               "records_match_gpu": bool(res["ok"] and tiles), "records_checked": res["records"],
                               "first_bad_file": res["bad"], "every_file_tiled_by_its_records": bool(tiles)}
    ring.close()
    eng.close()
    return out


def load_traffic_ring():
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json"):   # PMC passes on the ring's own kernels (newest round first)
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tj = json.load(f)
            return {"ratio": float(tj["ring"]["hbm_bytes_per_algorithmic_byte"]), "note": tj["ring"]["note"]}
        except Exception:
            continue
    return None


def ring_cpu_baseline(a, kept, single, file_bytes, kind, seed_of):
    """Oracle leg of the ring workload. The ring keeps no file resident (pages are refilled as they come free), so the
    checker REGENERATES the bytes of any range from the file's seed (oracle.fill is the CPU twin of the device
    generator): (1) the first --cpu-sample-gib of the first timed file, cut and hashed by the oracle on one core (the CPU
    figure), must equal the GPU's records; (2) restart points (oracle/restart_check.py) spread over EVERY timed file and
    the single-file pass — beyond 2^32 and 2^35, up to each file's final chunk."""
    from oracle import oracle as O
    from oracle import restart_check as RC

    O.build()
    cfg = O.new_config(a.avg)

    def regen(fidx):
        def download(off, n):
            lo = off & ~15
            buf = O.fill(n + (off - lo), seed_of(fidx), kind, stream_off=lo)
            return buf[off - lo:]
        return download

    f0 = min(kept)
    n = int(min(a.cpu_sample_gib * GiB, file_bytes)) & ~15
    host = O.fill(n, seed_of(f0), kind)
    t0 = time.perf_counter()
    want = O.chunk_and_digest(cfg, host, [(0, n)], impl=1)
    dt = time.perf_counter() - t0
    g = kept[f0]
    k = want.size - 1 if n < file_bytes else want.size
    same = bool(k > 0 and g.size >= k and np.array_equal(want["end"][:k], g["end"][:k])
                and np.array_equal(want["digest"][:k], g["digest"][:k]))
    spread = {"points": 0, "records_checked": 0, "bytes_checked": 0, "ok": True, "files": 0, "max_offset": 0, "mismatch": None}
    if not a.no_spread_check:
        files = dict(kept)
        files.update(single)
        per_file = max(4, int(a.spread_points * 4 / max(len(files), 1)))
        nth = max(1, min(32, os.cpu_count() or 1))
        for fidx, recs in sorted(files.items()):
            tiles = bool(recs.size and int(recs["end"][-1]) == file_bytes
                         and int(recs["size"].astype(np.int64).sum()) == file_bytes)
            if not tiles:
                spread["ok"] = False
                spread["mismatch"] = spread["mismatch"] or {"file": int(fidx), "what": "records do not tile the file"}
                continue
            r = RC.check_batch(regen(fidx), None, recs, a.avg, nbytes=file_bytes, k=per_file, span=64 << 20, threads=nth)
            spread["files"] += 1
            for key in ("points", "records_checked", "bytes_checked"):
                spread[key] += r[key]
            spread["max_offset"] = max(spread["max_offset"], r["max_offset"])
            if not r["ok"]:
                spread["ok"] = False
                spread["mismatch"] = spread["mismatch"] or dict(r["mismatch"], file=int(fidx))
    many = None
    try:
        many = cpu_many_core(a, O, cfg)
    except Exception as exc:  # pragma: no cover
        many = {"error": repr(exc)}
    return {
        "many_core": many,
        "value": round(n / GiB / dt, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
        "sample": f"{n / GiB:.3g} GiB prefix of the first timed file, oracle chunk_and_digest (byte-serial Buzhash + SHA-NI), "
                  f"{os.cpu_count()} host cores present",
        "records_match_gpu": bool(same and spread["ok"]),
        "records_checked": int(k) + int(spread["records_checked"]),
        "front_of_file": {"records_checked": int(k), "match": same},
        "whole_file_restart_points": spread,
    }


def host_cpu_limits():
    """what this process may use of the host: logical CPUs, its affinity mask, the cgroup CPU quota (v2 cpu.max / v1 cfs)"""
    info = {"cpu_count": os.cpu_count(), "affinity": None, "cgroup_cpu_max": None, "cgroup_cpus": None}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        info["cgroup_cpu_max"] = f"{q} {per}"
        if q != "max":
            info["cgroup_cpus"] = round(int(q) / int(per), 2)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            info["cgroup_cpu_max"] = f"{q} {per}"
            if q > 0:
                info["cgroup_cpus"] = round(q / per, 2)
        except Exception:
            pass
    try:
        info["loadavg_1m"] = round(os.getloadavg()[0], 1)
    except Exception:
        pass
    return info


def cpu_many_core(a, O, cfg):
    """The same port on MANY host cores at once — what the host's own cores reach on this path, the honest comparison for the
    PCIe-fed GPU figure. Persistent worker threads (started once, released per trial: no thread start-up inside a timed
    trial), >= 1 GiB of work per thread and trial (independent random streams; ctypes drops the GIL), thread counts doubled
    until the rate stops growing (< 5 % per doubling) or the time budget is spent. Reports the whole curve, the thread count
    that saturates, and what limits this process (affinity mask, cgroup CPU quota) — a figure that stops scaling at 64 of
    256 cores is either a CPU-capped container or memory bandwidth, and the line should say which it could be."""
    if a.brief:
        return None
    lim = host_cpu_limits()
    usable = lim["affinity"] or lim["cpu_count"] or 1
    if lim["cgroup_cpus"]:
        usable = max(1, min(usable, int(lim["cgroup_cpus"] + 0.999)))
    per = int(min(1 << 30, max(1 << 20, int(a.cpu_sample_gib * GiB) // 2)))     # 1 GiB per thread at the default sample size
    piece = min(per, 128 << 20)
    reps = max(1, per // piece)
    per = reps * piece
    bufs = [O.fill(piece, a.seed + 100 + i, 0) for i in range(min(8, max(1, usable)))]
    counts, n = [], 1
    while n < usable:
        counts.append(n)
        n *= 2
    counts.append(usable)
    maxn = counts[-1]
    go = [threading.Event() for _ in range(maxn)]
    done = threading.Semaphore(0)
    stop = {"flag": False}

    def worker(i):
        while True:
            go[i].wait()
            go[i].clear()
            if stop["flag"]:
                return
            for r in range(reps):
                O.chunk_and_digest(cfg, bufs[(i + r) % len(bufs)], [(0, piece)], impl=1)
            done.release()

    ths = [threading.Thread(target=worker, args=(i,), daemon=True) for i in range(maxn)]
    for t in ths:
        t.start()
    curve, best, sat = {}, None, None
    t_budget = time.perf_counter() + float(os.environ.get("PBS_BENCH_MANYCORE_BUDGET_S", "40"))
    prev = None
    for n in counts:
        t1 = time.perf_counter()
        for i in range(n):
            go[i].set()
        for _ in range(n):
            done.acquire()
        v = n * per / GiB / (time.perf_counter() - t1)
        curve[str(n)] = round(v, 2)
        if best is None or v > best[1]:
            best = (n, v)
        if prev is not None and sat is None and v < 1.05 * prev[1]:
            sat = prev[0]
        prev = (n, v)
        if sat is not None and n >= 2 * sat:
            break                                   # one more doubling past saturation confirms it
        if time.perf_counter() > t_budget:
            break
    stop["flag"] = True
    for e in go:
        e.set()
    return {"value": round(best[1], 2), "unit": "GiB/s", "cores": best[0],
            "curve_GiBps_by_threads": curve, "saturates_at_threads": sat,
            "per_thread_bytes": per, "host": lim,
            "sample": f"{per >> 20} MiB per thread and trial ({reps} x {piece >> 20} MiB independent random streams), persistent worker "
                      f"threads, thread counts doubled from 1 to {maxn} until the rate stops growing; best = {best[0]} threads",
            "note": "one thread = one serial Buzhash + SHA-NI stream, as the reference's one goroutine per archive; the curve shows "
                    "how far the HOST's cores carry this path when archives are independent"}


def run_batch(a, rank, local_rank, world, ctx):
    import pbs_plus_amd
    from pbs_plus_amd import buzhash

    cfg = buzhash.NewConfig(a.avg)
    want_slots = {"stream64g": 4, "manyfiles": 2, "corpus_dup": 2, "rechunk": 4}[a.workload]
    nslots = max(1, a.reread) if a.reread else (want_slots if a.slots is None else a.slots)
    eng = pbs_plus_amd.Engine(cfg, device=local_rank, inflight=min(16, max(nslots, 1)))
    w = WORKLOADS[a.workload](a, eng, rank, world)
    torch.cuda.synchronize()
    batches = w.batches
    ctx.rec_cap = max(b.nbytes // max(cfg.MinSize, 65) + 2 * b.nseg + 16 for b in batches)
    if ctx.dist is not None:       # the one-collective all-gather needs a record capacity every rank agrees on
        rc = torch.tensor([ctx.rec_cap], dtype=torch.int64, device=ctx.comm_dev)
        ctx.dist.all_reduce(rc, op=ctx.dist.ReduceOp.MAX)
        ctx.rec_cap = int(rc.item())
        setup_reduce(ctx, eng)      # (the C ABI's own communicator when the ranks sit on their own GPUs)
    inflight = max(1, a.reread) if a.reread else len(batches)

    def collect(b, t, timings):
        if timings is not None:
            timings.append(eng.timing(t))
        recs = eng.collect(t)
        b.last = recs
        w.after_collect(b, recs, ctx)
        return recs

    # with several ranks every rank must issue its collectives in the same sequence: first-in-first-out there
    collect_any = a.collect == "any" and ctx.dist is None

    def run_steps(k, timings=None):
        """k passes; passes overlap on separate HIP streams, one resident batch per slot. A batch is resubmitted
        only after its previous pass has been collected (its buffer is "refilled")."""
        pending = []                 # (batch, ticket) in submission order
        idle = list(batches)         # batches not in flight
        nbytes_done = 0

        def pop_one():
            i = 0
            if collect_any and len(pending) > 1:
                while True:          # whichever pass has finished on the device
                    done = [j for j, (_, t) in enumerate(pending) if eng.done(t)]
                    if done:
                        i = done[0]
                        break
                    time.sleep(0.0003)
            pb, t = pending.pop(i)
            collect(pb, t, timings)
            return pb

        for _ in range(k):
            if a.reread:
                if len(pending) == inflight:
                    pop_one()
                b = batches[0]
            else:
                b = idle.pop(0) if idle else pop_one()
            pending.append((b, eng.submit(b.buf, b.segs, b.nbytes)))
            nbytes_done += b.nbytes
        while pending:
            pb = pop_one()
            if not a.reread:
                idle.append(pb)
        return nbytes_done

    run_steps(a.warmup)
    if ctx.dist is not None:
        ctx.dist.barrier()
    torch.cuda.synchronize()
    timings = []
    t0 = time.perf_counter()
    my_bytes = run_steps(a.steps, timings)
    torch.cuda.synchronize()
    if ctx.dist is not None:
        ctx.dist.barrier()
    elapsed = time.perf_counter() - t0
    total_bytes = float(my_bytes)
    if ctx.dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctx.comm_dev)
        ctx.dist.all_reduce(tt, op=ctx.dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tb = torch.tensor([total_bytes], dtype=torch.float64, device=ctx.comm_dev)
        ctx.dist.all_reduce(tb, op=ctx.dist.ReduceOp.SUM)
        total_bytes = float(tb.item())
    w.finish(ctx)

    # one strictly serial pass (uncontended kernel times, single-batch latency), outside the timed region
    torch.cuda.synchronize()
    b0 = batches[0]
    ts0 = time.perf_counter()
    tk = eng.submit(b0.buf, b0.segs, b0.nbytes)
    serial_timing = eng.timing(tk)
    b0.last = eng.collect(tk)
    serial_s = time.perf_counter() - ts0

    out = None
    if rank == 0:
        out = assemble(a, w, cfg, world, inflight, elapsed, total_bytes, timings, serial_timing, serial_s)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(a, w)
        if ctx.dist is not None:
            out.setdefault("results", {})["digest_reduce_path"] = ctx.reduce_path
    if getattr(ctx, "cabi_comm", None) is not None:
        ctx.cabi_comm.close()
        ctx.cabi_comm = None
    w.free()
    eng.close()
    return out


def load_traffic():
    """HBM bytes per launch / algorithmic bytes, from the committed PMC passes (FETCH_SIZE, gfx950 x2 correction)."""
    for name in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tj = json.load(f)
            k = tj["kernels"]
            return {"sha": float(k["k_sha256_pair<RecordSource>"]["read_ratio_vs_algorithmic"]),
                    "scan": float(k["k_scan3<34,4>"]["read_ratio_vs_algorithmic"]), "file": "profiles/" + name}
        except Exception:
            continue
    return None


def assemble(a, w, cfg, world, inflight, elapsed, total_bytes, timings, serial_timing, serial_s):
    b0 = w.batches[0]
    nb0 = b0.nbytes
    value = total_bytes / GiB / elapsed
    agg_gbs = total_bytes / elapsed / 1e9
    sha_ms = float(np.mean([t["sha_ms"] for t in timings]))
    scan_ms = float(np.mean([t["scan_ms"] for t in timings]))
    resolve_ms = float(np.mean([t["resolve_ms"] for t in timings]))
    mean_bytes = total_bytes / world / max(len(timings), 1)
    tr = load_traffic()
    longest = int(max(int(b.last["size"].max()) for b in w.batches if b.last is not None and b.last.size))
    # the serial chain of THIS box (2.29-2.40 GHz sustained, box to box): in the strictly serial pass the hash launch of a
    # chain-bound batch lasts exactly as long as its longest chunk, so that launch measures the chain
    longest0 = int(b0.last["size"].max()) if b0.last is not None and b0.last.size else longest
    chain_bound = serial_timing["sha_ms"] > 0 and not sha_dense_pays_host(nb0, longest0)
    chain_us = (serial_timing["sha_ms"] * 1e3 / (longest0 / 64 + 1)) if chain_bound else CHAIN_US_PER_BLOCK
    chain_ms = longest / 64 * chain_us * 1e-3
    resident = float(sum(b.nbytes for b in w.batches))
    lat_s = (serial_timing["total_ms"] * 1e-3) or serial_s
    cfgd = {"bytes_per_batch": nb0, "resident_batches": len(w.batches), "resident_bytes_per_gpu": int(resident),
            "avg_chunk": a.avg, "chunks_per_batch": int(b0.last.size), "inflight_batches": inflight, "collect": a.collect,
            "distinct_data_per_slot": not a.reread,
            "parallelism": f"{world} rank(s), one per GPU, batches independent, digest-set all-gather" if world > 1 else "1 GPU"}
    cfgd.update(w.describe())
    if a.reread:
        cfgd["workload"] += f" — REREAD protocol: {a.reread} overlapping passes over the same resident batch"
    out = {
        "metric": "GiB/s ingested through CDC+SHA-256",
        "value": round(value, 2),
        "unit": "GiB/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": w.scaling,
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic (splitmix64 bytes generated on device, resident in HBM; every in-flight slot owns different bytes)",
        "config": cfgd,
        "roofline": {
            # path level: all input bytes of the timed region / wall time. What bounds it is integer VALU issue
            # (SHA-256, ~22 ops/B; the scan adds 3.5 ops/B), never HBM — and, below that ceiling, the serial chain of
            # the longest chunk times the bytes HBM can hold (latency_bound)
            "kernel": "whole path (k_scan3 + resolve + k_sha256_pair); dominant kernel k_sha256_pair<RecordSource>: "
                      "%.0f%% of device time" % (100.0 * sha_ms / max(sha_ms + scan_ms + resolve_ms, 1e-9)),
            "bound": "valu",
            "achieved": round(agg_gbs / world, 1),
            "peak": round(SHA_VALU_GBS, 1),
            "unit": "GB/s",
            "frac": round(agg_gbs / world / round(SHA_VALU_GBS, 1), 4),
            "peak_note": "%.1f T integer lane-ops/s (620 G wave64 VOP3 instr/s measured, profiles/r01_ubench_int_valu_issue.log) "
                         "/ %.1f ops per byte of SHA-256" % (VALU_PEAK_TOPS, SHA_OPS_PER_BYTE),
            "hbm": {"peak": HBM_PEAK_GBS, "frac": round(agg_gbs / world / HBM_PEAK_GBS, 4),
                    "note": "same achieved figure against the HBM3E peak (BASELINE.json quotes % of HBM roofline); the path "
                            "reads every byte twice (scan, SHA-256): real traffic = 2 x achieved"},
            "traffic": None if tr is None else int((tr["sha"] + tr["scan"]) * mean_bytes),
            "traffic_note": None if tr is None else
                "HBM read bytes per step = (scan %.2fx + SHA %.2fx) x algorithmic bytes, FETCH_SIZE PMC passes (%s)" % (
                    tr["scan"], tr["sha"], tr["file"]),
            "algorithmic_bytes_per_step": int(mean_bytes),
            "latency_bound": {
                "longest_chunk": longest, "chain_ms": round(chain_ms, 1),
                "serial_pass_ms": round(lat_s * 1e3, 1),
                "resident_bytes": int(resident),
                "bound_GiBps": round(resident / GiB / max(lat_s, 1e-9), 1),
                "frac_of_bound": round(value / world / max(resident / GiB / max(lat_s, 1e-9), 1e-9), 3),
                "chain_us_per_block": round(chain_us, 4), "chain_us_per_block_measured_this_run": bool(chain_bound),
                "note": "SHA-256 is serial inside a chunk (max-size chunk = %d compressions x %.3f us): a batch cannot "
                        "complete sooner than serial_pass_ms, and with batch-granular buffer release its bytes stay "
                        "resident that long: throughput <= resident bytes / pass latency. The ring workload "
                        "(--workload ring) releases memory per page instead" % (longest // 64, chain_us)},
            "kernels": {
                "k_sha256_pair<RecordSource>": {
                    "bound": "valu (serial chain per chunk)", "kernel_ms": round(sha_ms, 3),
                    "achieved": round(mean_bytes / (sha_ms * 1e-3) / 1e9, 1), "unit": "GB/s",
                    "valu_frac": round(mean_bytes / (sha_ms * 1e-3) / 1e9 / SHA_VALU_GBS, 4),
                    "hbm_frac": round(mean_bytes / (sha_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "uncontended_ms": round(serial_timing["sha_ms"], 3),
                    "chain_frac": round(chain_ms / max(serial_timing["sha_ms"], 1e-9), 3),
                    "traffic": None if tr is None else int(tr["sha"] * mean_bytes)},
                "k_scan3<34,4>": {
                    "bound": "hbm", "kernel_ms": round(scan_ms, 3),
                    "achieved": round(mean_bytes / (scan_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(mean_bytes / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "uncontended_ms": round(serial_timing["scan_ms"], 3),
                    "uncontended_frac": round(nb0 / (serial_timing["scan_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "traffic": None if tr is None else int(tr["scan"] * mean_bytes)},
                "resolve_chain": {"kernel_ms": round(resolve_ms, 3), "uncontended_ms": round(serial_timing["resolve_ms"], 3)},
                "note": "kernel_ms = mean per launch over the timed region, HIP events on the engine's own streams while %d "
                        "passes overlap (a launch then also waits for CUs); uncontended_ms = one strictly serial pass" % inflight,
            },
        },
        "serial_value": round(nb0 / GiB / serial_s, 2),
        "serial_step_ms": {"total": round(serial_s * 1e3, 2), "scan": round(serial_timing["scan_ms"], 3),
                           "resolve": round(serial_timing["resolve_ms"], 3), "sha256": round(serial_timing["sha_ms"], 3)},
    }
    if w.extra:
        out["results"] = w.extra
    return out


def sha_dense_pays_host(nbytes, longest_bytes, num_cus=256, pct=150):
    """host-side twin of pbsk::sha256_dense_pays (is a hash launch issue-bound rather than chain-bound?)"""
    return (nbytes // 64) * 100 > pct * (longest_bytes // 64 + 1) * 128 * num_cus


def spread_check(a, w, per_slot=40, span=64 << 20):
    """Oracle parity ACROSS every resident batch, not only at its front: restart points (oracle/restart_check.py)
    spread uniformly over each slot's bytes — beyond 2^32 and 2^35, the last scan tile, every slot — on the bytes
    DOWNLOADED from the device buffer the GPU chunked."""
    from oracle import restart_check as RC
    tot = {"points": 0, "records_checked": 0, "bytes_checked": 0, "ok": True, "slots": 0, "max_offset": 0, "mismatch": None}
    nth = max(1, min(32, os.cpu_count() or 1))
    for b in w.batches:
        if b.last is None or not hasattr(b.buf, "download"):
            continue
        segs = None if b.segs is None else [(int(x), int(y)) for x, y in b.segs]
        r = RC.check_batch(lambda off, n, b=b: b.buf.download(off, n), segs, b.last, a.avg, nbytes=b.nbytes,
                           k=per_slot, span=span, threads=nth)
        tot["slots"] += 1
        for key in ("points", "records_checked", "bytes_checked"):
            tot[key] += r[key]
        tot["max_offset"] = max(tot["max_offset"], r["max_offset"])
        if not r["ok"]:
            tot["ok"] = False
            tot["mismatch"] = tot["mismatch"] or dict(r["mismatch"], slot=b.label)
    return tot


def cpu_baseline(a, w):
    """The oracle (C restatement; SHA-NI like Go's crypto/sha256) on ONE host thread — the reference's writer is a
    single goroutine (internal/tapeio/converter.go:672-680) — over a bounded sample of the same workload. The same
    sample re-checks the GPU's records bit for bit (prefix causality for a stream, whole segments otherwise)."""
    from oracle import oracle as O

    O.build()
    host, segs, gpu, how = w.cpu_sample()
    cfg = O.new_config(a.avg)
    t0 = time.perf_counter()
    recs = O.chunk_and_digest(cfg, host, segs, impl=1)
    dt = time.perf_counter() - t0
    n = int(sum(s[1] for s in segs))
    same, k = False, 0
    if gpu is not None:
        if how == "prefix":      # cuts before the prefix end depend only on the prefix -> all but the tail must match
            k = recs.size - 1
            same = bool(k > 0 and gpu.size >= k and np.array_equal(recs["end"][:k], gpu["end"][:k])
                        and np.array_equal(recs["digest"][:k], gpu["digest"][:k]))
        else:
            k = recs.size
            same = bool(gpu.size == k and np.array_equal(recs["end"], gpu["end"]) and np.array_equal(recs["digest"], gpu["digest"])
                        and np.array_equal(recs["segment"], gpu["segment"]))
    spread = None
    if not a.no_spread_check:
        try:
            spread = spread_check(a, w, per_slot=a.spread_points)
        except Exception as exc:  # pragma: no cover
            spread = {"ok": False, "error": repr(exc), "records_checked": 0}
    many = None
    try:
        many = cpu_many_core(a, O, cfg)
    except Exception as exc:  # pragma: no cover
        many = {"error": repr(exc)}
    return {
        "many_core": many,
        "value": round(n / GiB / dt, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
        "sample": f"{n / GiB:.3g} GiB of the same workload ({how}: {len(segs)} segment(s)), oracle chunk_and_digest "
                  f"(byte-serial Buzhash + SHA-NI), {os.cpu_count()} host cores present",
        "records_match_gpu": bool(same and (spread is None or spread.get("ok", False))),
        "records_checked": int(k) + int(0 if spread is None else spread.get("records_checked", 0)),
        "front_of_batch": {"records_checked": int(k), "match": same},
        "whole_batch_restart_points": spread,
    }


# ---------------------------------------------------------------------------------------------------
# host-fed ingest: what the cgo drop-in actually sees (io.Reader bytes arrive in host memory)
# ---------------------------------------------------------------------------------------------------
def hostfeed_run(a, rank, local_rank, world, ctx):
    import pbs_plus_amd
    from pbs_plus_amd import buzhash

    cfg = buzhash.NewConfig(a.avg)
    eng = pbs_plus_amd.Engine(cfg, device=local_rank, inflight=2)
    P = max(1, a.producers)
    per_gib = 1.0 if a.gib is None else a.gib            # bytes each producer writes per step
    per = int(per_gib * GiB) & ~7
    # host source bytes: generated by the engine's own device generator and copied back (no oracle in the fed path)
    src = []
    gen = eng.alloc(min(per, 1 << 30))
    for i in range(P):
        eng.fill(gen.ptr, gen.nbytes, seed=a.seed + 17 * (rank * P + i), kind=0)
        src.append(gen.download())
    gen.free()
    wsize = 32 << 20
    h2d = eng.h2d_bandwidth(1 << 30) if hasattr(eng, "h2d_bandwidth") else None

    # one long-lived stream per producer (an archive being written): its device window ring and pinned staging are
    # allocated while it warms up, the timed phase is steady-state ingest and ends with finish() (all records delivered)
    warm_steps = max(1, a.warmup // 2)
    gate = threading.Barrier(P + 1)
    out = [None] * P
    nfiles = [0] * P
    t_written = [0.0] * P
    errs = []

    A = max(1, min(getattr(a, "archives", 1) or 1, a.steps))   # archives per producer, written back to back
    trace_hf = bool(os.environ.get("PBS_BENCH_HF_TRACE"))

    def producer(i):
        try:
            draining = []        # archives whose input is closed (finish_begin) and whose last chunks are still being hashed
            tot = {"nrec": 0, "bytes": 0}

            def reap(block):
                for st in list(draining):
                    if block:
                        st.finish()
                    tot["nrec"] += st.poll(1 << 16).size
                    nfiles[i] += len(st.poll_files())
                    if block or st.done():
                        tot["nrec"] += st.poll(1 << 16).size
                        nfiles[i] += len(st.poll_files())
                        st.close()               # its window buffers go back to the engine's pool: the next archive takes them
                        draining.remove(st)

            def archive(steps, count):
                tr = {"t0": time.perf_counter(), "max_write_ms": 0.0, "max_at_gib": 0.0, "reap_ms": 0.0, "poll_ms": 0.0}
                st = pbs_plus_amd.PayloadStream(eng, 256 << 20)
                tr["create_ms"] = (time.perf_counter() - tr["t0"]) * 1e3
                done_bytes = 0
                for _ in range(steps):
                    off = 0
                    if a.tee:
                        st.begin_file()
                    while off < per:
                        o = off % src[i].size
                        n = min(wsize, per - off, src[i].size - o)
                        tw = time.perf_counter()
                        st.write(src[i][o:o + n])
                        tw = (time.perf_counter() - tw) * 1e3
                        if tw > tr["max_write_ms"]:
                            tr["max_write_ms"], tr["max_at_gib"] = tw, (done_bytes + off) / GiB
                        off += n
                        if (off // wsize) % 8 == 0:
                            tp = time.perf_counter()
                            tot["nrec"] += st.poll(4096).size
                            tr["poll_ms"] += (time.perf_counter() - tp) * 1e3
                            tp = time.perf_counter()
                            reap(False)
                            tr["reap_ms"] += (time.perf_counter() - tp) * 1e3
                    if a.tee:
                        st.end_file()
                        nfiles[i] += len(st.poll_files())
                    done_bytes += per
                if count:
                    tot["bytes"] += st.bytes_written()
                tf = time.perf_counter()
                st.finish_begin()    # the goroutine goes on with its next archive; this one drains beside it
                tr["finish_begin_ms"] = (time.perf_counter() - tf) * 1e3
                tr["total_s"] = time.perf_counter() - tr["t0"]
                if trace_hf and count:
                    print(f"[hostfeed trace] producer {i}: archive of {steps * per / GiB:g} GiB in {tr['total_s']:.3f} s "
                          f"({steps * per / GiB / tr['total_s']:.1f} GiB/s): create {tr['create_ms']:.2f} ms, longest write "
                          f"{tr['max_write_ms']:.1f} ms at {tr['max_at_gib']:.2f} GiB, polls {tr['poll_ms']:.1f} ms, reaps "
                          f"{tr['reap_ms']:.1f} ms, finish_begin {tr['finish_begin_ms']:.2f} ms", file=sys.stderr, flush=True)
                draining.append(st)

            for k in range(2 if A > 1 else 1):      # warm-up (two overlapping archives when the timed phase overlaps them too)
                archive(max(1, warm_steps // (2 if A > 1 else 1)), False)
            reap(True)
            tot["nrec"] = 0
            gate.wait()          # warm-up written and drained
            gate.wait()          # timed phase starts
            for k in range(A):
                archive(a.steps // A + (1 if k < a.steps % A else 0), True)
            t_written[i] = time.perf_counter()   # the last archive's last byte has been accepted; what follows is its drain
            reap(True)
            out[i] = (tot["nrec"], tot["bytes"])
            gate.wait()          # every record of every archive delivered: end of the timed region
        except Exception as exc:  # noqa: BLE001
            errs.append(repr(exc))
            gate.abort()

    ths = [threading.Thread(target=producer, args=(i,)) for i in range(P)]
    for t in ths:
        t.start()
    gate.wait()
    if ctx.dist is not None:
        ctx.dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gate.wait()
    gate.wait()
    torch.cuda.synchronize()
    if ctx.dist is not None:
        ctx.dist.barrier()
    elapsed = time.perf_counter() - t0
    for t in ths:
        t.join()
    if errs:
        raise SystemExit("hostfeed producer failed: " + "; ".join(errs))
    total = float(sum(o[1] for o in out))
    if ctx.dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctx.comm_dev)
        ctx.dist.all_reduce(tt, op=ctx.dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        tb = torch.tensor([total], dtype=torch.float64, device=ctx.comm_dev)
        ctx.dist.all_reduce(tb, op=ctx.dist.ReduceOp.SUM)
        total = float(tb.item())
    # checker + cpu_baseline leg (the only place this workload touches the oracle): one producer's first 256 MiB through a
    # fresh stream must give the oracle's records; the oracle's time on that sample is the one-core CPU figure
    from oracle import oracle as O
    O.build()
    chk = pbs_plus_amd.PayloadStream(eng, 64 << 20)
    n_chk = min(src[0].size, 256 << 20)
    chk.write(src[0][:n_chk])
    chk.finish()
    srecs = chk.poll()
    chk.close()
    tc0 = time.perf_counter()
    orecs = O.chunk_and_digest(O.new_config(a.avg), src[0][:n_chk], [(0, n_chk)], impl=1)
    cpu_dt = time.perf_counter() - tc0
    same = bool(srecs.size == orecs.size and np.array_equal(srecs["end"], orecs["end"])
                and np.array_equal(srecs["digest"], orecs["digest"]))
    outj = None
    if rank == 0:
        gbs = total / elapsed / 1e9
        outj = {
            "metric": "GiB/s ingested through CDC+SHA-256 (host-fed, PCIe-inclusive)",
            "value": round(total / GiB / elapsed, 2), "unit": "GiB/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32",
            "data": "synthetic random bytes in HOST memory, written through pbsgpu_stream_write (pinned staging -> H2D)",
            "config": {"workload": f"host-fed payload streams: {P} producer threads x {per_gib:g} GiB per step, "
                                   f"32 MiB writes, 256 MiB device windows (the WriteEntryReader seam)"
                                   + (f"; {A} archives per producer back to back, each archive's drain overlapped with the next "
                                      f"one (pbsgpu_stream_finish_begin)" if A > 1 else ""),
                       "archives_per_producer": A,
                       "producers": P, "avg_chunk": a.avg, "records": int(sum(o[0] for o in out)), "bytes": int(total),
                       "xxh3_tee_files": int(sum(nfiles)) if a.tee else None},
            "roofline": {"kernel": "H2D copy engine (PCIe Gen5 x16)", "bound": "pcie", "achieved": round(gbs / world, 1),
                         "peak": 63.0, "unit": "GB/s", "frac": round(gbs / world / 63.0, 4), "traffic": None,
                         "measured_h2d_GBps": h2d,
                         "frac_of_measured_h2d": None if not h2d else round(gbs / world / h2d, 3)},
            "write_phase": {"seconds": round(max(t_written) - t0, 4),
                            "GiBps": round(total / world / GiB / max(max(t_written) - t0, 1e-9), 2),
                            "drain_seconds": round(elapsed - (max(t_written) - t0), 4),
                            "note": "until the last writer's last byte was accepted (this rank); the rest is finish(): the last "
                                    "windows' cut + the serial SHA-256 chain of their longest chunk, a fixed ~0.5-0.7 s per "
                                    "archive whatever its length"},
            "stream_records_match_oracle": same,
            "cpu_baseline": {"value": round(n_chk / GiB / cpu_dt, 4), "unit": "GiB/s", "cores": 1, "kind": "port",
                             "sample": f"{n_chk >> 20} MiB of one producer's bytes, oracle chunk_and_digest (byte-serial Buzhash + "
                                       f"SHA-NI), {os.cpu_count()} host cores present", "records_match_gpu": same},
        }
    eng.close()
    return outj if rank == 0 else None


# ---------------------------------------------------------------------------------------------------
# verification batches (A8 / A9): whole-file SHA-256 and XXH3-64
# ---------------------------------------------------------------------------------------------------
def verify_main(a, rank, local_rank, world, ctx):
    import hashlib

    import pbs_plus_amd
    from pbs_plus_amd import buzhash

    eng = pbs_plus_amd.Engine(buzhash.NewConfig(a.avg), device=local_rank, inflight=2)
    fbytes = int(a.file_mib * MiB) & ~7
    gib = 16.0 if a.gib is None else a.gib
    nfiles = max(1, int(gib * GiB) // fbytes)
    buf = eng.alloc(nfiles * fbytes)
    eng.fill(buf.ptr, nfiles * fbytes, seed=a.seed + rank, kind=0)
    segs = np.array([(i * fbytes, fbytes) for i in range(nfiles)], dtype=np.uint64)
    res = {}
    for name, fn in (("sha256", eng.sha256_many), ("xxh3", eng.xxh3_many)):
        fn(buf, segs[: max(1, nfiles // 8)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            outv = fn(buf, segs)
        dt = (time.perf_counter() - t0) / a.steps
        res[name] = {"GiBps": round(nfiles * fbytes / GiB / dt, 2), "ms": round(dt * 1e3, 2), "files": nfiles}
        res[name + "_out0"] = bytes(outv[0]).hex() if name == "sha256" else int(outv[0])
    # the reference keeps 4 files in flight (internal/server/verification/job.go:493): 4 files, one GPU lane each for
    # SHA-256 (serial inside a file: 64 B per 1.655 us = 38.7 MB/s per file whatever the file size), one wave each for XXH3
    big = min(fbytes, 64 << 20) & ~7
    segs4 = np.array([(i * big, big) for i in range(4)], dtype=np.uint64)
    for name, fn in (("sha256_4_files_in_flight", eng.sha256_many), ("xxh3_4_files_in_flight", eng.xxh3_many)):
        t0 = time.perf_counter()
        fn(buf, segs4)
        dt = time.perf_counter() - t0
        res[name] = {"GiBps": round(4 * big / GiB / dt, 3), "ms": round(dt * 1e3, 1), "file_bytes": big}
    # CPU side on one core: hashlib (OpenSSL, SHA-NI) and the xxhash C library, and the check of file 0
    host = buf.download(0, fbytes)
    t0 = time.perf_counter()
    want = hashlib.sha256(host.tobytes()).hexdigest()
    cpu_dt = time.perf_counter() - t0
    res["sha256_file0_matches_hashlib"] = (want == res["sha256_out0"])
    try:
        import xxhash
        t0 = time.perf_counter()
        wx = xxhash.xxh3_64_intdigest(host.tobytes())
        res["cpu_xxh3_one_core_GiBps"] = round(fbytes / GiB / (time.perf_counter() - t0), 2)
        res["xxh3_file0_matches_xxhash"] = (wx == res["xxh3_out0"])
    except ImportError:
        pass
    cpu_sha = fbytes / GiB / cpu_dt
    res["cpu_sha256_one_core_GiBps"] = round(cpu_sha, 3)
    lane = 64 / 1.655e-6 / GiB
    res["sha256_crossover_files_per_cpu_core"] = round(cpu_sha / lane, 1)
    res["note"] = ("SHA-256 is serial inside a file: the GPU hashes one file per lane at %.3f GiB/s, a SHA-NI core at %.2f GiB/s; "
                   "a batch beats C host cores only with more than ~%.0f x C files in flight (the reference's verify job "
                   "keeps 4), XXH3 (block-parallel inside a file, one wave per file) wins from the first file" % (lane, cpu_sha, cpu_sha / lane))
    if rank == 0:
        outj = {"metric": "GiB/s whole-file SHA-256 (verification.HashFile batches)", "value": res["sha256"]["GiBps"],
                "unit": "GiB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": res["sha256"]["ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u32", "data": "synthetic random files resident in HBM",
                "config": {"workload": f"{nfiles} files x {fbytes / MiB:g} MiB hashed whole (A8/A9), plus 4 files x {big / MiB:g} MiB "
                                       f"(the reference's 4-in-flight verify job)"},
                "roofline": {"kernel": "k_sha256_pair<SegmentSource>", "bound": "valu",
                             "achieved": round(res["sha256"]["GiBps"] * 1.073741824, 1), "peak": round(SHA_VALU_GBS, 1),
                             "unit": "GB/s", "frac": round(res["sha256"]["GiBps"] * 1.073741824 / SHA_VALU_GBS, 4), "traffic": None},
                "cpu_baseline": {"value": round(fbytes / GiB / cpu_dt, 3), "unit": "GiB/s", "cores": 1, "kind": "port",
                                 "sample": f"hashlib.sha256 (OpenSSL SHA-NI) of one {fbytes / MiB:g} MiB file"},
                "results": res}
        print(json.dumps(outj), flush=True)
    buf.free()
    eng.close()
    if ctx.dist is not None:
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
