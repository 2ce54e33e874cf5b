#!/usr/bin/env python3
"""Scale checks at BASELINE sizes: (1) 128 GiB single batch (configs[3] per-GPU share) as 2048 x 64 MiB
segments of mixed entropy; (2) 10 000 small segments. Properties only (no 128 GiB oracle)."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pbs_plus_amd import Engine, buzhash

eng = Engine(buzhash.NewConfig(4 << 20), inflight=1)
SEG = 64 << 20
nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
total = nseg * SEG
buf = eng.alloc(total)
t0 = time.perf_counter()
for kind in range(4):  # entropy class by seg % 4: fill each class region-wise (4 fills per 256 MiB stripe is slow; do per segment)
    pass
for i in range(nseg):
    eng.fill(buf.ptr + i * SEG, SEG, seed=1000 + i, kind=i % 4)
print(f"filled {total >> 30} GiB in {time.perf_counter() - t0:.1f} s", flush=True)
segs = np.stack([np.arange(nseg, dtype=np.uint64) * np.uint64(SEG), np.full(nseg, SEG, dtype=np.uint64)], axis=1)
t0 = time.perf_counter()
t = eng.submit(buf, segs, nbytes=total)
tm = eng.timing(t)
recs = eng.collect(t)
dt = time.perf_counter() - t0
print(f"{nseg} x 64 MiB = {total >> 30} GiB: {recs.size} records in {dt:.3f} s = {total / dt / 2**30:.1f} GiB/s; timing {tm}", flush=True)
# properties per segment
seg_ids = recs["segment"]
assert np.all(np.diff(seg_ids.astype(np.int64)) >= 0)
for s in (0, 1, 2, 3, nseg // 2, nseg - 1):
    r = recs[seg_ids == s]
    ends = r["end"].astype(np.int64)
    assert ends[-1] == SEG and np.all(np.diff(ends) > 0), s
    sizes = r["size"].astype(np.int64)
    assert np.array_equal(np.diff(np.concatenate([[0], ends])), sizes)
    assert sizes.max() <= 16 << 20 and (len(sizes) == 1 or sizes[:-1].min() >= 1 << 20)
    if s % 4 == 1:
        assert sizes.tolist() == [16 << 20] * 4, sizes  # zero file: max-size cuts only
    i = int(np.flatnonzero(seg_ids == s)[0])
    data = buf.download(s * SEG + int(ends[0] - sizes[0]), int(sizes[0]))
    assert bytes(recs["digest"][i]) == hashlib.sha256(data.tobytes()).digest(), s
# identical zero files -> identical digests; dedup on device
dup, stats = eng.dedup(recs)
print("dedup:", stats, "dup fraction by bytes %.3f" % (1 - stats["unique_bytes"] / stats["total_bytes"]), flush=True)
assert stats["total_bytes"] == total
buf.free()
# (2) 10 000 small ragged segments in one batch
rng = np.random.default_rng(2)
lens = rng.integers(0, 3 << 20, 10_000).astype(np.uint64)
offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
tot2 = int(lens.sum())
b2 = eng.alloc(tot2 + 8)
eng.fill(b2.ptr, (tot2 + 7) & ~7, seed=9, kind=0)
t0 = time.perf_counter()
r2 = eng.chunk_and_digest(b2, np.stack([offs, lens], axis=1), nbytes=tot2)
print(f"10000 ragged segments ({tot2 >> 20} MiB): {r2.size} records in {time.perf_counter() - t0:.3f} s", flush=True)
per_seg = np.bincount(r2["segment"], weights=r2["size"].astype(np.float64), minlength=10_000)
assert np.array_equal(per_seg.astype(np.uint64), lens)
print("scale-ok")
