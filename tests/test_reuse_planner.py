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
