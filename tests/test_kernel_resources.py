"""Build-quality guard for the gfx950 kernels (no GPU needed: hipcc cross-compiles and reports per-kernel resources).
What the design relies on and a careless edit would silently break:
  * no kernel spills to scratch (a spilling SHA-256 round loop or scan loop costs far more than any tuning gains);
  * the sparse SHA-256 form keeps exactly one workgroup per CU (its static LDS + the launch's 16 KB padding > half of
    the 160 KB LDS) with one wave per SIMD, the dense form holds four pairs (eight waves) in one workgroup per CU;
  * register budgets that the occupancy assumptions in DESIGN.md section 5 are built on."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LDS_PER_CU = 160 * 1024


@pytest.fixture(scope="module")
def usage():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "pbs_plus_amd", "csrc"), "-s", "usage"], capture_output=True,
                         text=True, timeout=600)
    text = out.stdout + out.stderr
    kernels = {}
    cur = None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("agprs", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)")):
            mm = re.search(pat, line)
            if mm:
                cur[key] = int(mm.group(1))
    assert len(kernels) >= 25, (len(kernels), text[-2000:])
    return kernels


def _find(kernels, *needles):
    hit = [k for k in kernels if all(n in k for n in needles)]
    assert hit, needles
    return hit


def test_no_kernel_spills(usage):
    ours = {k: v for k, v in usage.items() if k.startswith("_ZN4pbsk")}
    assert len(ours) >= 25, len(ours)
    # (k_ring_control — the page ring's one-workgroup control kernel — keeps ~60 kernel arguments live across its phases:
    # a few SGPRs parked in VGPR lanes are fine there, it runs once per cut round and carries no bytes; the same holds for
    # k_resolve, one latency-bound wave per segment, since round 6 also carries the DenseTiles arguments of the on-demand
    # re-scan: uniform values kept in VGPR lanes, never memory)
    spilled = {k: v for k, v in ours.items()
               if v.get("vgpr_spill", 0) or (v.get("sgpr_spill", 0) and "k_ring_control" not in k and "k_resolveILb" not in k)}
    assert not spilled, spilled
    # scratch memory only where a kernel indexes a small private array on purpose (k_compact sorts <= 48 slots of a tile
    # in one thread); never in the kernels that carry the bytes
    hot = [k for k in ours if any(n in k for n in ("k_scan", "k_sha256", "k_xxh3", "k_resolve", "k_par_", "k_order", "k_pack",
                                                  "k_gather", "k_publish", "k_fill"))]
    assert len(hot) >= 20, len(hot)
    scratchy = {k: ours[k]["scratch"] for k in hot if ours[k].get("scratch", 0)}
    assert not scratchy, scratchy
    allowed = {k: v["scratch"] for k, v in ours.items() if v.get("scratch", 0)}
    assert all("k_compact" in k or "k_ring_control" in k for k in allowed), allowed  # (the control kernel compacts tiles too)


def test_sha256_pair_forms_keep_their_cu_residency(usage):
    sparse = _find(usage, "k_sha256_pair", "RecordSource", "Lb0E")
    dense = _find(usage, "k_sha256_pair", "RecordSource", "Lb1E")
    for k in sparse:
        r = usage[k]
        # one workgroup per CU: two of them (with the launch's 16 KB dynamic padding each) must not fit
        assert 2 * (r["lds"] + (16 << 10)) > LDS_PER_CU and r["lds"] + (16 << 10) <= LDS_PER_CU, r
        assert r["vgprs"] <= 256 and r["occupancy"] >= 1, r
    for k in dense:
        r = usage[k]
        assert 2 * r["lds"] > LDS_PER_CU and r["lds"] <= LDS_PER_CU, r   # one 8-wave workgroup per CU, no padding needed
        assert r["vgprs"] <= 256 and r["occupancy"] >= 2, r              # two waves per SIMD must fit
    # every source type has both forms
    for src in ("RecordSource", "SegmentSource"):
        assert _find(usage, "k_sha256_pair", src, "Lb0E") and _find(usage, "k_sha256_pair", src, "Lb1E")


def test_single_wave_sha_and_scan_register_budgets(usage):
    for k in _find(usage, "k_sha256INS"):
        assert usage[k]["vgprs"] <= 128, usage[k]      # four waves per SIMD (the refuted lanes form runs there)
    scans = [k for k in usage if "k_scan3" in k]
    assert scans
    for k in scans:
        r = usage[k]
        assert r["vgprs"] <= 256 and r["occupancy"] >= 2, r   # two waves per SIMD: the scan's latency hiding
        assert r["lds"] <= LDS_PER_CU, r
