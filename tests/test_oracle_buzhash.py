"""Pins the oracle's Buzhash chunker (oracle/buzhash_oracle.c).

PARITY UNPINNED vs github.com/pbs-plus/pxar v0.34.0 (absent from /root/reference; the
reference's own tests pin no boundary — SURVEY.md §8c). What is checked here:
 * parameter derivation of buzhash.NewConfig (reference commit_orchestrate.go:143-149 fixes
   only avg = 4 << 20; commit_walk_test.go:25 uses 4096),
 * the invariant upstream's own chunker test checks: 1-byte feeds == whole-buffer feeds,
 * chunk sizes recalled from upstream Proxmox's chunker test on its LE-u32-counter buffer
   (hash-determined, so they corroborate table + scan semantics),
 * structural properties (min/max, zero runs) and the candidate/resolve decomposition the
   GPU engine relies on,
 * the committed golden fixture tests/golden/chunks_v1.json.
"""
import numpy as np
import pytest

from helpers import (golden_case_data, golden_files, golden_records, golden_suggested_data, load_golden, records_equal,
                     resolve_model)


def test_config_derivation(O):
    c = O.new_config(4 << 20)
    assert (c.avg, c.min, c.max, c.window) == (4 << 20, 1 << 20, 16 << 20, 64)
    assert (c.mask, c.break_min) == (0x7FFFFF, 0x7FFFFD)
    c = O.new_config(4096)
    assert (c.min, c.max, c.mask, c.break_min) == (1024, 16384, 0x1FFF, 0x1FFD)


@pytest.mark.parametrize("bad", [0, 1, 255, 3000, 4097, (1 << 28) + 1, 1 << 29])
def test_config_rejects_non_power_of_two_or_out_of_range(O, bad):
    with pytest.raises(ValueError):
        O.new_config(bad)


def test_default_table_is_balanced(O):
    """casync's table has a balanced bit distribution per bit position (128 ones each);
    a single mistyped word would almost surely break this."""
    t = O.default_table()
    assert len(set(t.tolist())) == 256
    for bit in range(32):
        assert int(((t >> np.uint32(bit)) & np.uint32(1)).sum()) == 128, bit


def test_streaming_equals_whole_buffer(O):
    """Upstream test_chunker1's invariant: feeding single bytes == feeding the whole buffer."""
    cfg = O.new_config(4096)
    data = O.fill(300_000, 3)
    whole = O.chunk_stream(cfg, data)
    for step in (1, 7, 64, 1000):
        ch = O.StreamingChunker(cfg)
        ends, pos, base = [], 0, 0
        while pos < data.size:
            n = min(step, data.size - pos)
            off = 0
            while off < n:
                k = ch.scan(data[pos + off: pos + n])
                if k == 0:
                    break
                off += k
                ends.append(pos + off)
            pos += n
        if not ends or ends[-1] != data.size:
            ends.append(data.size)
        assert np.array_equal(np.asarray(ends, dtype=np.uint64), whole), step


def test_upstream_counter_buffer_chunk_sizes(O):
    """Buffer = LE u32 counters 0..262143 (upstream Proxmox chunker tests), avg 64 KiB.
    Recalled upstream expectation with suggested boundaries: sizes
    [32768, 110609, 229376, 32768, 262144, 262144, 118767]; the hash-determined numbers are
    the cut 110609 bytes after a boundary at 32768 and the 118767 tail."""
    buf = np.arange(256 * 1024, dtype="<u4").view(np.uint8)
    cfg = O.new_config(64 * 1024)
    sizes = np.diff(np.concatenate([[0], O.chunk_stream(cfg, buf)])).astype(int).tolist()
    assert sizes == [143377, 262144, 262144, 262144, 118767]
    assert 143377 == 32768 + 110609
    after = np.diff(np.concatenate([[0], O.chunk_stream(cfg, buf[32768:])])).astype(int).tolist()
    assert after[0] == 110609


def test_upstream_suggested_boundary_vector(O):
    """Payload chunker (ChunkerImpl + suggested boundaries) on the same counter buffer. The recalled upstream
    expectation [32768, 110609, 229376, 32768, 262144, 262144, 118767] is reproduced — for 1-byte feeds, whole-buffer
    feeds and 4 KiB feeds alike, which is the invariant upstream's test asserts — with the boundary list
    {32768 (twice: the second yields an empty chunk and is dropped), 0 (in the past), 372753, 405521 (aligned with
    the regular max-size cut at 143377 + 262144)}. 372753 is implied by the expected sizes themselves (a 32768-byte
    chunk ending on the regular boundary 405521 needs a cut there that no hash/max rule produces); without it the
    same code gives [32768, 110609, 262144, 262144, 262144, 118767]. EXTERNAL / recalled: corroboration, not a pin."""
    buf = np.arange(256 * 1024, dtype="<u4").view(np.uint8)
    cfg = O.new_config(64 * 1024)
    for feed in (1, 0, 4096, 77777):
        e = O.chunk_stream_suggested(cfg, buf, [32 * 1024, 32 * 1024, 0, 372753, 405521], feed)
        assert np.diff(np.concatenate([[0], e])).astype(int).tolist() == \
            [32768, 110609, 229376, 32768, 262144, 262144, 118767], feed
        e = O.chunk_stream_suggested(cfg, buf, [32 * 1024, 32 * 1024, 0, 405521], feed)
        assert np.diff(np.concatenate([[0], e])).astype(int).tolist() == \
            [32768, 110609, 262144, 262144, 262144, 118767], feed
    # no suggestions == the plain chunker
    assert np.array_equal(O.chunk_stream_suggested(cfg, buf, [], 1), O.chunk_stream(cfg, buf))


@pytest.mark.parametrize("avg,n,kind", [(256, 60_000, 0), (4096, 900_000, 0), (4096, 500_000, 1), (4096, 700_000, 3),
                                        (65536, 5_000_000, 0)])
def test_suggested_boundaries_as_a_second_candidate_list(O, avg, n, kind):
    """The engine's formulation (tests/helpers.py::resolve_model_suggested = kernels.hip k_resolve) equals the serial
    payload chunker fed byte by byte, including boundaries closer than min (dropped), farther than max (left to the
    hash scan and re-examined from the next chunk), duplicates, and boundaries at / past the end."""
    from helpers import resolve_model_suggested

    cfg = O.new_config(avg)
    data = O.fill(n, 2000 + avg + n, kind)
    rng = np.random.default_rng(avg + n)
    cands = O.candidates(cfg, data)
    for dens in (3, 40, 400):
        sg = sorted(set(int(x) for x in rng.integers(1, n + 10, dens)) | {cfg.min, cfg.min - 1, cfg.max, cfg.max + 1, n, 64, 65})
        sg = sg + sg[:2]
        want = O.chunk_stream_suggested(cfg, data, sorted(sg), 1)
        got = resolve_model_suggested(cands, sg, data.size, cfg.min, cfg.max)
        assert np.array_equal(got, want), (dens, got[:6], want[:6])


def test_suggested_boundary_feed_dependence_is_documented(O):
    """Upstream's scan lets a suggested boundary inside the CURRENT buffer win over an earlier hash cut in the same
    buffer, so its result depends on how much data one scan call sees; the engine implements the byte-serial limit
    (earlier hash cut wins). Find an input where the two differ and pin both behaviours."""
    cfg = O.new_config(4096)
    data = O.fill(200_000, 99, 0)
    plain = O.chunk_stream(cfg, data)
    first = int(plain[0])
    b = first + 700                      # a boundary shortly after the first hash cut, within [min, max] of offset 0
    assert cfg.min <= b <= cfg.max
    serial = O.chunk_stream_suggested(cfg, data, [b], 1)
    whole = O.chunk_stream_suggested(cfg, data, [b], 0)
    assert int(serial[0]) == first       # byte-serial: the hash cut comes first, the boundary is then too close (< min)
    assert int(whole[0]) == b            # whole-buffer feed: the boundary pre-empts the scan


def test_chunk_size_bounds_and_zero_runs(O):
    cfg = O.new_config(4096)
    data = O.fill(2_000_000, 8)
    sizes = np.diff(np.concatenate([[0], O.chunk_stream(cfg, data)]))
    assert sizes[:-1].min() >= cfg.min and sizes.max() <= cfg.max
    # constant bytes hash to 0 in every window: only max-size cuts
    for byte in (0, 0x5A):
        z = np.full(5 * cfg.max + 123, byte, dtype=np.uint8)
        zs = np.diff(np.concatenate([[0], O.chunk_stream(cfg, z)])).astype(int).tolist()
        assert zs == [cfg.max] * 5 + [123]
    assert O.candidates(cfg, np.zeros(100_000, np.uint8)).size == 0


def test_min_equals_window_edge(O):
    """avg 256 -> min 64 == window: the break test first runs at chunk_size 65."""
    cfg = O.new_config(256)
    data = O.fill(200_000, 12)
    ends = O.chunk_stream(cfg, data)
    sizes = np.diff(np.concatenate([[0], ends]))
    assert sizes[:-1].min() >= 65
    assert np.array_equal(resolve_model(O.candidates(cfg, data), data.size, cfg.min, cfg.max), ends)


@pytest.mark.parametrize("avg,n,kind", [(256, 100_000, 0), (4096, 1_500_000, 0), (4096, 900_000, 3),
                                        (4096, 700_000, 2), (65536, 6_000_000, 0), (4096, 64, 0),
                                        (4096, 65, 0), (4096, 1, 0), (4096, 1024, 0), (4096, 1025, 0)])
def test_candidate_resolve_decomposition_equals_serial(O, avg, n, kind):
    """The engine's parallel formulation: (all-position window hash candidates) + (min/max
    resolution over the sorted list) must equal the serial chunker."""
    cfg = O.new_config(avg)
    data = O.fill(n, 1000 + avg + n, kind)
    cands = O.candidates(cfg, data)
    assert np.all(np.diff(cands.astype(np.int64)) > 0)
    assert np.array_equal(resolve_model(cands, n, cfg.min, cfg.max), O.chunk_stream(cfg, data))


def test_candidate_density_matches_three_in_2avg(O):
    cfg = O.new_config(4096)
    n = 8 << 20
    c = O.candidates(cfg, O.fill(n, 77))
    expect = 3 * n / (2 * 4096)
    assert abs(c.size - expect) < 6 * expect ** 0.5


def test_segments_are_independent_streams(O):
    cfg = O.new_config(4096)
    a, b = O.fill(200_000, 1), O.fill(123_457, 2)
    both = np.concatenate([a, b])
    recs = O.chunk_and_digest(cfg, both, [(0, a.size), (a.size, b.size)])
    ra = O.chunk_and_digest(cfg, a)
    rb = O.chunk_and_digest(cfg, b)
    assert np.array_equal(recs["end"], np.concatenate([ra["end"], rb["end"]]))
    assert np.array_equal(recs["digest"], np.concatenate([ra["digest"], rb["digest"]]))
    assert recs["segment"].tolist() == [0] * ra.size + [1] * rb.size
    assert int(recs["size"].sum()) == both.size


@pytest.mark.parametrize("fixture", golden_files())
def test_golden_fixture(O, fixture):
    """tests/golden/chunks_v1.json was produced by tests/golden/make_golden.py (the oracle's own output: a regression
    pin, not a parity pin); tests/golden/chunks_go.json — when a maintainer has run `make golden-go` — comes from the REAL
    github.com/pbs-plus/pxar module and IS the parity pin: the oracle must reproduce every record of it, the suggested-
    boundary cuts at every reader-buffer size, and its Config derivation is compared with the module's field dump."""
    g = load_golden(fixture)
    assert g["schema"] in ("pbsgpu-golden-v1", "pbsgpu-golden-v2")
    for case in g["cases"]:
        cfg = O.new_config(case["avg"])
        data, table = golden_case_data(O, case)
        got = O.chunk_and_digest(cfg, data, table, impl=1)
        want = golden_records(case, O.RECORD_DTYPE)
        assert records_equal(got, want), (fixture, case["name"])
    for case in g.get("suggested", []):
        cfg = O.new_config(case["avg"])
        data = golden_suggested_data(O, case)
        ends = O.chunk_stream_suggested(cfg, data, sorted(case["suggested"]), feed=int(case["feed"]))
        assert ends.tolist() == [int(e) for e in case["ends"]], (fixture, case["name"], case["feed"])
    for avg, fields in (g.get("config") or {}).items():   # the module's Config, as reflection saw it: name what differs
        cfg = O.new_config(int(avg))
        ours = {"min": cfg.min, "max": cfg.max, "avg": cfg.avg, "window": cfg.window, "mask": cfg.mask, "break_min": cfg.break_min}
        flat = {str(k).lower().replace("_", ""): v for k, v in fields.items() if isinstance(v, (int, float))}
        for key, val in ours.items():
            hits = [v for k, v in flat.items() if key.replace("_", "") in k]
            assert not hits or int(val) in [int(h) for h in hits], (fixture, avg, key, val, fields)


def test_fill_is_offset_consistent(O):
    for kind in range(5):
        whole = O.fill(300_000, 5, kind)
        for off in ((16, 4096, 65536 + 16, 131072) if kind == 4 else (8, 4096, 65536 + 8, 131072)):   # kind 4 works in 16-byte blocks
            assert np.array_equal(O.fill(1000, 5, kind, stream_off=off), whole[off:off + 1000]), (kind, off)
    arx = O.fill(4 << 20, 5, 4)      # the ring bench's refill bytes: every byte value about equally often
    hist = np.bincount(arx, minlength=256)
    assert hist.min() > 0.9 * arx.size / 256 and hist.max() < 1.1 * arx.size / 256
    assert not O.fill(1 << 16, 5, 1).any()
    z = O.fill(8 << 20, 5, 3)
    frac = 1.0 - np.count_nonzero(z.reshape(-1, 65536).any(axis=1)) / (z.size / 65536)
    assert 0.15 < frac < 0.45


@pytest.mark.parametrize("avg,n,kind,strip", [(4096, 70_000, 0, 4352), (256, 9_000, 0, 512), (4096, 50_000, 3, 512),
                                              (65536, 300_000, 0, 4352)])
def test_scan_kernel_algebra_model_equals_rolling_hash(O, avg, n, kind, strip):
    """The three identities the scan kernel rests on (prefix ring, unrotated leaving term, pre-rotated table
    with one unsigned compare), stated in numpy (tests/helpers.py::scan_lane_model) and checked against the
    oracle's rolling hash — so kernel edits can be validated against the math without a GPU."""
    from helpers import scan_lane_model

    cfg = O.new_config(avg)
    data = O.fill(n, 31 + avg, kind)
    got = scan_lane_model(data, O.default_table(), cfg.mask, cfg.break_min, strip=strip)
    assert np.array_equal(got, O.candidates(cfg, data))


@pytest.mark.parametrize("avg,n,kind,lines", [(256, 40_000, 0, 1), (4096, 90_001, 0, 4), (4096, 70_000, 3, 2)])
def test_coop_scan_dataflow_model_equals_rolling_hash(O, avg, n, kind, lines):
    """k_scan3's dataflow (quad-cooperative pieces, staged transpose, rolling hash over two alternating rings of
    table values, warm-up as half-line -1) restated in numpy and checked against the oracle: the index algebra
    of the default scan kernel can be validated without a GPU."""
    from helpers import scan_coop_model

    cfg = O.new_config(avg)
    data = O.fill(n, 77 + avg, kind)
    got = scan_coop_model(data, O.default_table(), cfg.mask, cfg.break_min, lines=lines)
    assert np.array_equal(got, O.candidates(cfg, data))
