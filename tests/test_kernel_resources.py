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


@pytest.fixture(scope="module")
def sha_isa(tmp_path_factory):
    """The device ISA of kernels.hip, one text per SHA-256 kernel (hipcc -S --offload-device-only; ~1 min)."""
    d = tmp_path_factory.mktemp("isa")
    src = os.path.join(ROOT, "pbs_plus_amd", "csrc", "kernels.hip")
    out = os.path.join(str(d), "kernels.s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                        "--offload-device-only", "-S", src, "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    kernels = {}
    for m in re.finditer(r"^(_ZN4pbsk\w*k_sha256\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        kernels[m.group(1)] = m.group(2)
    assert len(kernels) >= 10, list(kernels)
    return kernels


def test_sha256_kernels_read_and_write_through_global_instructions_only(sha_isa):
    """a flat_load counts in lgkmcnt, so the producer's wait in front of every s_barrier awaited the block requested in
    the same step (rounds 3-5: no prefetch under load); and ONE flat_store anywhere in the kernel leaves a pending-flat state that
    turns every VMEM wait of the producer's loop into vmcnt(0). No flat memory instruction may come back."""
    bad = {k: re.findall(r"^\s*(flat_(?:load|store|atomic)\w*)", v, re.M)[:3] for k, v in sha_isa.items()}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, bad


def test_sha256_producers_keep_the_other_slot_in_flight(sha_isa):
    """Each block slot's sixteen v_perm are awaited with vmcnt(9)..vmcnt(5): the other slot's five requests stay outstanding. A
    vmcnt(0) in front of them means the prefetch is gone again (a branch around the requests, an exit in the middle of the loop,
    message words sunk behind the refill: kernels.hip, k_sha256_pair)."""
    checked = 0
    for k, v in sha_isa.items():
        if "k_sha256_pair" not in k and "k_sha256_xpair" not in k:
            continue
        lines = v.splitlines()
        first = [i for i, l in enumerate(lines) if re.match(r"\s*v_perm_b32 v\d+, v\d+, v\d+, v\d+\s*$", l)
                 and not re.match(r"\s*v_perm_b32 v\d+, v\d+, v\d+, v\d+\s*$", lines[i - 1])]
        waits = [lines[i - 1].strip() for i in first]
        loop = [w for w in waits if w.startswith("s_waitcnt vmcnt(")]
        depth = [int(re.search(r"vmcnt\((\d+)\)", w).group(1)) for w in loop]
        assert len(depth) >= 2 and min(depth) >= 5, (k, waits)   # never fewer than the other slot's five requests outstanding
        assert depth.count(5) >= 2, (k, waits)                   # both slots
        checked += 1
    assert checked >= 8, checked
