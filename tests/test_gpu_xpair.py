"""The EXPRESS SHA-256 form (k_sha256_xpair: two lanes per chunk — lane A carries e,f,g,h, lane B a,b,c,d two rounds behind,
DPP exchanges inside the round; the producer's two lanes of a chunk expand consecutive blocks) against the CPU oracle and
hashlib, through the C ABI:
  * as the ONLY hash kernel of the batch path (PBSGPU_SHA_MODE=xpair, read once per process -> subprocesses): every chunk of
    every size class goes through it — chunks of 1, 2, 3 blocks (odd and even block counts decide which lane holds the last
    block), unaligned starts, padding that spills into an extra block, 16 MiB chains;
  * as the ring's express SERVICE beside the pair service (PBSGPU_RING_XP_CUS): long chunks take the long-chunk queue and the
    two-lane kernel, everything else the pair kernel; chunks crossing pages, page release from either lane of the pair.
The formulation itself was checked lane by lane against hashlib on the CPU first: scripts/r4_xpair_sim.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, env, timeout=240):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT)
    assert "xp-ok" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    return out.stdout


def test_simulation_of_the_lane_pair_rounds_matches_hashlib():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r4_xpair_sim.py")], capture_output=True, text=True,
                         timeout=120)
    assert "matches hashlib" in out.stdout, out.stdout + out.stderr


def test_batch_path_with_the_express_kernel_only(gpu_lib):
    code = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import oracle as O\n"
        "from pbs_plus_amd import Engine, buzhash\n"
        "from tests.helpers import records_equal\n"
        # chunkers from 256 B average (1-16 block chunks, ragged tails) to the production 4 MiB (zero run: 16 MiB chains)
        "for avg, n, kind, seed in ((256, 700_001, 0, 3), (4096, 3_000_001, 3, 5), (65536, 40_000_003, 0, 7),\n"
        "                           (4 << 20, 80 << 20, 0, 9), (4 << 20, (48 << 20) + 13, 1, 11)):\n"
        "    eng = Engine(buzhash.NewConfig(avg))\n"
        "    data = O.fill(n, seed, kind)\n"
        "    got = eng.chunk_and_digest(data)\n"
        "    assert records_equal(got, O.chunk_and_digest(O.new_config(avg), data)), avg\n"
        "    # many ragged segments in one batch (empty, 1..130 bytes, misaligned starts)\n"
        "    segs, o = [], 0\n"
        "    for i in range(300):\n"
        "        ln = (i * 7) %% 131\n"
        "        segs.append((o, ln)); o += ln + (i %% 3)\n"
        "    got = eng.chunk_and_digest(data[:o + 200], segs)\n"
        "    assert records_equal(got, O.chunk_and_digest(O.new_config(avg), data[:o + 200], segs)), ('ragged', avg)\n"
        "    eng.close()\n"
        # whole-segment hashing (pbsgpu_sha256_many): every length 0..300 at every alignment class, vs hashlib
        "eng = Engine(buzhash.NewConfig(4096))\n"
        "blob = O.fill(1 << 20, 21, 0)\n"
        "segs = [((17 * i) %% 4096 + i, i %% 301) for i in range(1200)] + [(5, 70_001), (100_003, 300_000), (3, 64), (64, 55), (1, 56)]\n"
        "dig = eng.sha256_many(blob, segs)\n"
        "for (o, n), d in zip(segs, dig):\n"
        "    assert bytes(d) == hashlib.sha256(blob[o:o + n].tobytes()).digest(), (o, n)\n"
        "eng.close()\n"
        "print('xp-ok')\n" % ROOT)
    _run(code, dict(PBSGPU_SHA_MODE="xpair"))


RING_CODE = (
    "import sys, threading, numpy as np\n"
    "sys.path.insert(0, %r)\n"
    "from oracle import oracle as O\n"
    "from pbs_plus_amd import Engine, PageRing, buzhash\n"
    "def same(g, w, what):\n"
    "    assert g.size == w.size, (what, g.size, w.size)\n"
    "    for f in ('end', 'size', 'digest'):\n"
    "        assert np.array_equal(g[f], w[f]), (what, f)\n"
    "def run(avg, opt, jobs, conc=None):\n"
    "    eng = Engine(buzhash.NewConfig(avg), device=0, inflight=1)\n"
    "    ring = PageRing(eng, **opt)\n"
    "    got = ring.ingest_synthetic(jobs, timeout_s=90.0, concurrent=conc)\n"
    "    ring.quiesce()\n"
    "    st = ring.stats()\n"
    "    cfg = O.new_config(avg)\n"
    "    for i, (seed, kind, n) in enumerate(jobs):\n"
    "        w = O.chunk_and_digest(cfg, O.fill(n, seed, kind), [(0, n)]) if n else np.zeros(0, dtype=O.RECORD_DTYPE)\n"
    "        same(got[i], w, (avg, i))\n"
    "    assert st['pages_free'] == st['pages_total'] and st['pages_recycled'] == st['pages_enqueued'], st\n"
    "    long_chunks = sum(int((g['size'] >= LONG).sum()) for g in got)\n"
    "    ring.close(); eng.close()\n"
    "    return long_chunks\n" % ROOT)


def test_ring_with_an_express_service_small_pages(gpu_lib):
    """avg 4 KiB chunker (max 16 KiB), 64 KiB pages, long = 10 KiB: ~4 % of the chunks of random data and every chunk of the
    zero / periodic streams go through the express service (2 CUs) — many of them crossing pages — while the pair service
    (4 CUs) takes the rest; the arena of 24 pages turns over many times."""
    code = RING_CODE + (
        "LONG = 10240\n"
        "nl = run(4096, dict(arena_bytes=24 * (65536 + 256), page_bytes=65536, max_streams=8, sha_cus=4, round_pages=6),\n"
        "         [(21, 0, (1 << 20) + 5), (22, 1, 300 * 1024), (23, 3, 700 * 1024 + 3), (24, 0, 0), (25, 0, 63), (26, 2, 65536),\n"
        "          (27, 0, 65536 * 3), (28, 4, 65536 * 2 + 1), (29, 0, 1), (30, 1, 1 << 20)])\n"
        "assert nl > 50, nl\n"
        "print('xp-ok', nl)\n")
    _run(code, dict(PBSGPU_RING_XP_CUS="2", PBSGPU_RING_LONG_BYTES="10240", PBSGPU_RING_IDLE_TIMEOUT_S="5"))


def test_ring_with_an_express_service_production_chunker(gpu_lib):
    """avg 4 MiB: default pages (16.2 MiB), long = 10 MiB (the default 5/8 of the maximum): random, 30 % zero extents and an
    all-zero stream (every chunk 16 MiB = express) through 3 GiB of pages."""
    code = RING_CODE + (
        "LONG = 10 << 20\n"
        "G = 1 << 30\n"
        "nl = run(4 << 20, dict(arena_bytes=3 * G, max_streams=4, sha_cus=48, round_pages=64),\n"
        "         [(51, 0, 3 * G // 2 + 56), (52, 3, 3 * G // 2), (53, 1, G // 2 + 4096), (54, 4, G + 24)])\n"
        "assert nl > 30, nl\n"
        "print('xp-ok', nl)\n")
    _run(code, dict(PBSGPU_RING_XP_CUS="16", PBSGPU_RING_IDLE_TIMEOUT_S="5"))
