"""The lane-pair SHA-256 formulation behind k_sha256_xpair, executed lane by lane on the CPU: lane A carries
e,f,g,h, lane B a,b,c,d two slot-rounds behind; every slot is ONE operation on both lanes (per-lane shift registers, the ROLE
mask, v_xad, the DPP exchange of X1). It must equal hashlib for every padding case — the kernel is a transcription of this."""
import hashlib
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sim():
    spec = importlib.util.spec_from_file_location("r4_xpair_sim", os.path.join(ROOT, "scripts", "r4_xpair_sim.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # (runs its own self-check on import)
    return mod


def test_lane_pair_formulation_matches_hashlib():
    sim = _sim()
    for n in (0, 1, 3, 55, 56, 57, 63, 64, 65, 119, 120, 127, 128, 129, 1000):
        data = bytes((i * 131 + n) & 0xFF for i in range(n))
        assert sim.sha(data) == hashlib.sha256(data).digest(), n


def test_both_lanes_execute_the_same_nine_slots():
    """The point of the formulation: no slot is specific to a lane. block_pair's inner loop is written once and executed
    for ('A','B') and ('B','A') with per-lane constants only (SH, ROLE, the constant 1 in place of K+W)."""
    sim = _sim()
    assert sim.SH == {"A": (6, 11, 25), "B": (2, 13, 22)} and sim.ROLE == {"A": 0, "B": 0xFFFFFFFF}
