"""Chunk-reuse planner (host logic, no GPU): the reference's own table tests restated against the
C-ABI functions — TestLookupDynamicEntries / TestShouldReuse / the rangeEnd arithmetic of
internal/pxarmount/commit_bottleneck_test.go:795-908 (synthetic DIDX: n chunks of equal size,
digest[0] = i, digest[1] = i >> 8, commit_bottleneck_test.go:773-793)."""
import numpy as np
import pytest

from pbs_plus_amd import RECORD_DTYPE
from pbs_plus_amd import reuse


def synthetic_didx(num_chunks, chunk_size):
    idx = np.zeros(num_chunks, dtype=RECORD_DTYPE)
    for i in range(num_chunks):
        idx[i]["end"] = (i + 1) * chunk_size
        idx[i]["size"] = chunk_size
        idx[i]["digest"][0] = i & 0xFF
        idx[i]["digest"][1] = (i >> 8) & 0xFF
    return idx


LOOKUP_CASES = [
    ("full_range", 0, 500, 5, 0, 0),
    ("aligned_first_chunk", 0, 100, 2, 0, 100),
    ("aligned_last_chunk", 400, 500, 1, 0, 0),
    ("middle_two_chunks", 100, 300, 3, 0, 100),
    ("misaligned_start", 50, 300, 4, 50, 100),
    ("misaligned_end", 100, 350, 3, 0, 50),
    ("misaligned_both", 50, 350, 4, 50, 50),
    ("tiny_range_in_first", 10, 20, 1, 10, 80),
    ("empty_range", 100, 100, 0, 0, 0),
    ("past_end", 600, 700, 0, 0, 0),
]


@pytest.mark.parametrize("name,rs,re_,nchunks,spad,epad", LOOKUP_CASES, ids=[c[0] for c in LOOKUP_CASES])
def test_lookup_dynamic_entries(name, rs, re_, nchunks, spad, epad):
    chunks, sp, ep = reuse.lookup_dynamic_entries(synthetic_didx(5, 100), rs, re_)
    assert (len(chunks), sp, ep) == (nchunks, spad, epad)
    if chunks:
        assert chunks[0]["padding"] >= sp and chunks[-1]["padding"] >= ep
        assert all(c["size"] == 100 for c in chunks)
        assert sum(c["padding"] for c in chunks) == sp + ep


def test_should_reuse_reference_cases():
    idx = synthetic_didx(10, 1000)
    assert reuse.should_reuse(idx, [(i * 10, 10) for i in range(1000)]) is True      # aligned_full_chunks
    assert reuse.should_reuse(idx, [(0, 900)]) is True                                # single_file_aligned (8.4 % padding)
    assert reuse.should_reuse(idx, [(500, 10)]) is False                              # 97 % of the chunk would be padding
    assert reuse.should_reuse(None, [(0, 100)]) is True                               # nil index
    assert reuse.should_reuse(idx, []) is True


def test_range_end_includes_payload_header():
    # TestRangeHoleDetection: sortKey 100, fileSize 200 -> batchRangeEnd 316 (= 100 + 200 + HeaderSize)
    assert reuse.range_end(100, 200) == 316
    assert reuse.pending_refs_range([(100, 200), (316, 50)]) == (100, 382)


def test_saved_chunk_discount():
    """keepLastChunk: when the first chunk of the range is the chunk saved from the previous batch,
    the bytes already used from it do not count as padding (commit_reuse.go:166-174)."""
    idx = synthetic_didx(10, 1000)
    chunks, sp, ep = reuse.lookup_dynamic_entries(idx, 600, 1900)
    assert (sp, ep, len(chunks)) == (600, 100, 2)
    refs = [(600, 1300 - 16)]
    assert reuse.should_reuse(idx, refs) is False  # 700 / 2000 padding
    saved = {"size": 1000, "padding": 400, "endOffset": 1000, "digest": bytes(idx[0]["digest"])}
    assert reuse.should_reuse(idx, refs, saved=saved) is True  # 600 used bytes discounted -> 100 / 1400


def test_records_from_the_engine_feed_the_planner_roundtrip():
    """DIDX bytes -> records -> planner (the index the planner reads is what the engine writes)."""
    from pbs_plus_amd.engine import didx_decode

    idx = synthetic_didx(4, 4096)
    magic = bytes([28, 145, 78, 165, 25, 186, 179, 205])
    blob = bytearray(4096)
    blob[:8] = magic
    body = b"".join(int(r["end"]).to_bytes(8, "little") + bytes(r["digest"]) for r in idx)
    back, _, _ = didx_decode(bytes(blob) + body)
    chunks, sp, ep = reuse.lookup_dynamic_entries(back, 5000, 9000)
    assert [c["endOffset"] for c in chunks] == [8192, 12288] and (sp, ep) == (904, 3288)


# ---- randomised cross-check against a line-by-line Python statement of the reference functions -------------------
def _model_lookup(ends, digests, rs, re_):
    """lookupDynamicEntries, internal/pxarmount/commit_reuse.go:84-135, over (End, Digest) lists."""
    n = len(ends)
    if n == 0 or rs >= re_:
        return [], 0, 0
    start = next((i for i in range(n) if ends[i] > rs), None)     # DynamicIndexReader.ChunkFromOffset
    if start is None:
        return [], 0, 0
    prev_end = ends[start - 1] if start > 0 else 0
    spad, epad, chunks = rs - prev_end, 0, []
    for i in range(start, n):
        chunks.append({"size": ends[i] - prev_end, "padding": 0, "endOffset": ends[i], "digest": digests[i]})
        prev_end = ends[i]
        if re_ < ends[i]:
            epad = ends[i] - re_
            break
    if chunks:
        chunks[0]["padding"] += spad
        chunks[-1]["padding"] += epad
    return chunks, spad, epad


def _model_should(ends, digests, refs, saved, threshold=reuse.CHUNK_PADDING_THRESHOLD):
    """shouldReuse, commit_reuse.go:152-183 (pendingRefsRange :137-150; rangeEnd commit_types.go:24-32)."""
    if not ends or not refs:
        return True
    rs = refs[0][0]
    re_ = max(k + n + reuse.HEADER_SIZE for k, n in refs)
    if re_ <= rs:
        return True
    chunks, spad, epad = _model_lookup(ends, digests, rs, re_)
    if not chunks:
        return True
    padding = spad + epad
    if saved is not None and saved["digest"] == chunks[0]["digest"] and saved["endOffset"] == chunks[0]["endOffset"]:
        used = saved["size"] - saved["padding"]
        padding = 0 if used > padding else padding - used
    total = (re_ - rs) + padding
    return True if total == 0 else padding / total <= threshold


@pytest.mark.parametrize("seed", range(6))
def test_planner_equals_the_reference_statement_on_random_indexes(seed):
    rng = np.random.default_rng(seed)
    for _ in range(60):
        n = int(rng.integers(0, 40))
        sizes = rng.integers(1, 5000, n)
        ends = [int(x) for x in np.cumsum(sizes)]
        idx = np.zeros(n, dtype=RECORD_DTYPE)
        digests = []
        for i in range(n):
            d = rng.integers(0, 256, 32, dtype=np.uint8)
            idx[i]["end"], idx[i]["size"], idx[i]["digest"] = ends[i], sizes[i], d
            digests.append(bytes(d))
        total = ends[-1] if n else 1000
        for _ in range(25):
            # ranges of every kind: inside one chunk, ending exactly on a chunk end, empty, reversed, past the index
            rs = int(rng.integers(0, total + 200))
            kind = int(rng.integers(0, 5))
            if kind == 0 and n:
                re_ = ends[int(rng.integers(0, n))]
            elif kind == 1:
                re_ = rs
            elif kind == 2:
                re_ = max(0, rs - int(rng.integers(0, 50)))
            else:
                re_ = rs + int(rng.integers(1, total + 300))
            got = reuse.lookup_dynamic_entries(idx, rs, re_)
            want = _model_lookup(ends, digests, rs, re_)
            assert got == want, (seed, ends, rs, re_)
            # shouldReuse over pending refs that span [rs, ...): with and without a matching saved chunk
            k = int(rng.integers(1, 4))
            refs, pos = [], rs
            for _ in range(k):
                fs = int(rng.integers(0, 3000))
                refs.append((pos, fs))
                pos += fs + reuse.HEADER_SIZE + int(rng.integers(0, 40))
            saved = None
            chunks = _model_lookup(ends, digests, refs[0][0], max(a + b + reuse.HEADER_SIZE for a, b in refs))[0]
            pick = int(rng.integers(0, 3))
            if chunks and pick == 0:      # the chunk the previous flush kept (sameIndexedChunkAs: digest + endOffset)
                c = chunks[0]
                saved = {"size": c["size"], "padding": int(rng.integers(0, c["size"] + 1)), "endOffset": c["endOffset"],
                         "digest": c["digest"]}
            elif chunks and pick == 1:    # same digest, other offset: must NOT be discounted
                c = chunks[0]
                saved = {"size": c["size"], "padding": 0, "endOffset": c["endOffset"] + 1, "digest": c["digest"]}
            assert reuse.should_reuse(idx, refs, saved) == _model_should(ends, digests, refs, saved), (seed, ends, refs, saved)
