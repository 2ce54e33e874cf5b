"""Pins oracle/xxh3_oracle.py (XXH3-64 restatement) against the independent `xxhash` C library —
the digest the reference records per new file (xxh3.New() tee, commit_reuse.go:450-461)."""
import numpy as np
import xxhash

from oracle import xxh3_oracle as X


def test_every_length_class_matches_xxhash():
    rng = np.random.default_rng(8)
    blob = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    for n in list(range(0, 300)) + [511, 512, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 4096, 4999]:
        assert X.xxh3_64(blob[:n]) == xxhash.xxh3_64_intdigest(blob[:n]), n


def test_known_answers():
    assert X.xxh3_64(b"") == 0x2D06800538D394C2
    assert xxhash.xxh3_64_intdigest(b"") == 0x2D06800538D394C2
    assert X.xxh3_64(b"a" * 100_000) == xxhash.xxh3_64_intdigest(b"a" * 100_000)
