"""The evidence under profiles/ must stay reproducible from what is committed beside it (no GPU needed):
the measured HBM traffic ratio the bench line quotes is recomputed from the committed counter CSVs, the kernel-trace anatomy
script runs on the committed trace of the driver's command, and bench.py picks the newest round's traffic summary."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def test_traffic_summary_is_what_the_counter_passes_say(tmp_path):
    runs = os.path.join(PROF, "r05_ring_pmc_runs.jsonl")
    one = tmp_path / "run.json"
    one.write_text([l for l in open(runs) if l.startswith("{")][1])   # the FETCH_SIZE pass's run line
    out = tmp_path / "traffic.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r5_traffic.py"),
                    os.path.join(PROF, "r05_pmc_fetch_size_ring_both_services.csv"),
                    os.path.join(PROF, "r05_pmc_write_size_ring_both_services.csv"), str(one), str(out)], check=True, timeout=60,
                   capture_output=True)
    got, want = json.load(open(out)), json.load(open(os.path.join(PROF, "r05_traffic.json")))
    assert got["algorithmic_bytes"] == want["algorithmic_bytes"] == 4 * 48 * (1 << 30)
    assert got["ring"]["hbm_bytes_per_algorithmic_byte"] == want["ring"]["hbm_bytes_per_algorithmic_byte"]
    r = got["ring"]["hbm_bytes_per_algorithmic_byte"]
    assert 3.0 <= r <= 3.1, r      # refill write 1 + scan read ~1.04 + service read 1 (algorithmic = 1 B/B)
    ks = got["kernels"]
    svc = [v for k, v in ks.items() if k.startswith("k_sha256_pair")][0]
    assert abs(svc["read_ratio_vs_algorithmic"] - 1.0) < 0.002, svc   # the service reads every byte exactly once


def test_trace_anatomy_script_runs_on_the_committed_trace():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r5_trace_regimes.py"),
                        os.path.join(PROF, "r05_kernel_trace_bench_default.csv.gz")], check=True, timeout=120, capture_output=True,
                       text=True)
    out = p.stdout
    assert "service launch 20" in out, out[:400]                       # the timed launch of the driver's command: ~2.06 s
    assert "overlap:" in out and "rounds in the window" in out
    rounds = int(out.split("rounds in the window ")[1].split(":")[0])
    assert rounds > 1000, out                                          # steady state: one gate quantum per round


def test_bench_quotes_the_newest_measured_traffic():
    sys.path.insert(0, ROOT)
    import importlib

    bench = importlib.import_module("bench")
    t = bench.load_traffic_ring()
    want = json.load(open(os.path.join(PROF, "r06_traffic.json")))["ring"]["hbm_bytes_per_algorithmic_byte"]
    assert t and abs(t["ratio"] - want) < 1e-9 and "round 6" in t["note"]
