"""The crafted candidate-dense inputs of the GPU parity tests (tests/dense_inputs.py) checked against the CPU oracle: the
patterns really are what the GPU tests assume (every position a candidate / none at all), and the decomposition the engine
relies on for such data — "the cut rule only needs the first candidate at or behind s + max(min, 65)" — equals the serial
chunker on them."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dense_inputs as D  # noqa: E402
from helpers import resolve_model  # noqa: E402


@pytest.fixture(scope="module")
def allp(O):
    return D.all_candidate_pattern(O.default_table())


def test_window_hash_restatement_and_all_candidate_pattern(O, allp):
    T = O.default_table()
    assert D.window_hash(T, allp) == 0xFFFFFFFF
    for ph in (1, 17, 63):                                          # the hash of a period's window only rotates with the phase
        assert D.window_hash(T, np.roll(allp, -ph)) == 0xFFFFFFFF
    for avg in (256, 4096, 65536, 4 << 20):
        cfg = O.new_config(avg)
        c = O.candidates(cfg, np.tile(allp, 40))
        assert np.array_equal(c, np.arange(64, 40 * 64 + 1, dtype=c.dtype))   # every END offset from the first full window on
        ends = O.chunk_stream(cfg, np.tile(allp, (6 * cfg.min) // 64 + 3))
        assert (np.diff(np.r_[0, ends])[:-1] == max(cfg.min, 65)).all()       # cut at the minimum every time


def test_periods_that_divide_32_have_no_candidate_at_all(O, allp):
    cfg = O.new_config(4096)
    for period in (1, 2, 4, 8, 16, 32):
        assert O.candidates(cfg, np.tile(allp[:period], 20000 // period)).size == 0


def test_first_candidate_rule_equals_the_serial_chunker_on_crafted_streams(O, allp):
    for avg, total, unit in ((256, 60_000, 900), (4096, 700_000, 20_000)):
        cfg = O.new_config(avg)
        onep = D.one_phase_pattern(O, cfg)
        data = D.crafted_stream(O, cfg, total, 3, allp, onep, unit)
        cands = O.candidates(cfg, data)
        assert cands.size > total // 8                              # really dense
        want = O.chunk_stream(cfg, data)
        assert np.array_equal(resolve_model(cands, data.size, cfg.min, cfg.max), want)
