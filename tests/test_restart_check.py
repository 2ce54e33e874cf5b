"""The whole-batch parity checker (oracle/restart_check.py) itself, on the CPU: fed with the oracle's own records it
must accept, cover the tail and every segment kind, and it must reject a single wrong digest, a shifted cut, a missing
final record — otherwise a green `records_match_gpu` in the bench line would mean nothing."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import restart_check as RC


@pytest.fixture(scope="module")
def corpus():
    O.build()
    rng = np.random.default_rng(11)
    n = 3 << 20
    data = rng.integers(0, 256, n, dtype=np.uint8)
    data[(1 << 20):(1 << 20) + 200000] = 0            # a zero run: max-size cuts
    segs = [(0, 1 << 20), (1 << 20, (1 << 20) + 12345), ((2 << 20) + 12345, n - (2 << 20) - 12345)]
    cfg = O.new_config(4096)
    recs = O.chunk_and_digest(cfg, data, segs)
    return data, segs, recs


def _dl(data):
    return lambda off, n: data[off:off + n]


def test_accepts_the_oracles_own_records_and_reaches_the_tail(corpus):
    data, segs, recs = corpus
    r = RC.check_batch(_dl(data), segs, recs, 4096, k=24, span=96 << 10, threads=4)
    assert r["ok"], r
    assert r["points"] >= 20 and r["records_checked"] > 200
    assert r["max_offset"] == data.size                      # the last segment's final chunk was part of a span


def test_single_segment_form(corpus):
    data, _, _ = corpus
    cfg = O.new_config(4096)
    recs = O.chunk_and_digest(cfg, data, [(0, data.size)])
    r = RC.check_batch(_dl(data), None, recs, 4096, nbytes=data.size, k=16, span=128 << 10, threads=2)
    assert r["ok"] and r["max_offset"] == data.size, r


@pytest.mark.parametrize("how", ["digest", "end", "drop_last", "extra"])
def test_rejects_wrong_records(corpus, how):
    data, segs, recs = corpus
    bad = recs.copy()
    if how == "digest":
        bad["digest"][bad.size // 2, 7] ^= 1
    elif how == "end":
        j = bad.size // 3
        bad["end"][j] += 1
        bad["size"][j] += 1
    elif how == "drop_last":
        bad = bad[:-1]
    else:
        extra = bad[-1:].copy()
        extra["end"] += 10
        bad = np.concatenate([bad, extra])
    # every record is covered when the points are dense enough
    r = RC.check_batch(_dl(data), segs, bad, 4096, k=400, span=96 << 10, threads=4)
    assert not r["ok"], (how, r)
    assert r["mismatch"] is not None
