"""The C-ABI shared library: it builds, loads without a GPU, exports every symbol
include/pbsgpu.h declares, validates arguments, and refuses to run without a device
(no CPU fallback). No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from pbs_plus_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "pbsgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pbsgpu_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(L):
    from pbs_plus_amd import _lib

    declared = header_symbols()
    assert len(declared) >= 35
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (pbsgpu_[a-z0-9_]+)", out))
    missing = [s for s in declared if s not in exported]
    assert not missing, f"declared in include/pbsgpu.h but not exported: {missing}"
    unbound = [s for s in declared if s not in _lib.SYMBOLS]
    assert not unbound, f"not bound in pbs_plus_amd/_lib.py: {unbound}"


def test_abi_version_and_strerror(L):
    assert L.pbsgpu_abi_version() == 5
    assert L.pbsgpu_strerror(0) == b"ok"
    assert b"device" in L.pbsgpu_strerror(-2)


def test_record_layout_matches_header():
    from pbs_plus_amd import RECORD_DTYPE

    assert RECORD_DTYPE.itemsize == 48
    assert RECORD_DTYPE.fields["end"][1] == 0 and RECORD_DTYPE.fields["digest"][1] == 8
    assert RECORD_DTYPE.fields["segment"][1] == 40 and RECORD_DTYPE.fields["size"][1] == 44


def test_newconfig_matches_oracle_and_reference_call_sites(L, O):
    from pbs_plus_amd import buzhash

    for avg in (256, 4096, 65536, 4 << 20, 1 << 28):
        c, o = buzhash.NewConfig(avg), O.new_config(avg)
        assert (c.AvgSize, c.MinSize, c.MaxSize, c.WindowSize, c.BreakTestMask, c.BreakTestMinimum) == (
            o.avg, o.min, o.max, o.window, o.mask, o.break_min)
        assert np.array_equal(c.Table, O.default_table())
    c = buzhash.NewConfig(4 << 20)  # commit_orchestrate.go:144, converter.go:248
    assert (c.MinSize, c.MaxSize, c.BreakTestMask) == (1 << 20, 16 << 20, 0x7FFFFF)


@pytest.mark.parametrize("bad", [0, 100, 255, 3 << 20, (1 << 28) + 1, 1 << 40, -1])
def test_newconfig_error_return(L, bad):
    from pbs_plus_amd import buzhash

    with pytest.raises(buzhash.ConfigError):
        buzhash.NewConfig(bad)


def test_injected_table(L):
    from pbs_plus_amd import buzhash

    t = np.arange(256, dtype=np.uint32) * np.uint32(2654435761)
    c = buzhash.NewConfig(4096, table=t)
    assert np.array_equal(c.Table, t)
    with pytest.raises(buzhash.ConfigError):
        buzhash.NewConfig(4096, table=np.zeros(10, np.uint32))


def test_engine_refuses_without_device_or_bad_config(L):
    from pbs_plus_amd import Engine, PbsGpuError, _lib, buzhash

    cfg = buzhash.NewConfig(4096)
    if L.pbsgpu_device_count() == 0:
        with pytest.raises(PbsGpuError) as ei:
            Engine(cfg)
        assert ei.value.status == _lib.E_NO_DEVICE
    bad = _lib.Config.from_buffer_copy(cfg._c)
    bad.mask = 0x1FFE  # not 2^k - 1
    h = C.c_void_p()
    assert L.pbsgpu_engine_create(0, C.byref(bad), 1, C.byref(h)) == _lib.E_INVALID
    bad = _lib.Config.from_buffer_copy(cfg._c)
    bad.min = 32  # < window
    assert L.pbsgpu_engine_create(0, C.byref(bad), 1, C.byref(h)) == _lib.E_INVALID
    assert L.pbsgpu_engine_create(0, None, 1, C.byref(h)) == _lib.E_INVALID


def test_didx_decode_rejects_garbage_and_roundtrips_layout(L):
    from pbs_plus_amd import RECORD_DTYPE, PbsGpuError
    from pbs_plus_amd.engine import didx_decode

    with pytest.raises(PbsGpuError):
        didx_decode(b"\0" * 4096)
    # hand-built image: magic | uuid | ctime | csum | pad, then 40-byte entries
    magic = bytes([28, 145, 78, 165, 25, 186, 179, 205])
    hdr = bytearray(4096)
    hdr[:8] = magic
    hdr[24:32] = (1_700_000_000).to_bytes(8, "little")
    hdr[32:64] = bytes(range(32))
    ents = b"".join((e).to_bytes(8, "little") + bytes([i]) * 32 for i, e in enumerate([100, 300, 316]))
    recs, ctime, csum = didx_decode(bytes(hdr) + ents)
    assert ctime == 1_700_000_000 and csum == bytes(range(32))
    assert recs["end"].tolist() == [100, 300, 316] and recs["size"].tolist() == [100, 200, 16]
    assert bytes(recs["digest"][2]) == bytes([2]) * 32
    with pytest.raises(PbsGpuError):
        didx_decode(bytes(hdr) + ents[:-1])
    n = C.c_uint64()
    assert L.pbsgpu_didx_size(3, C.byref(n)) == 0 and n.value == 4096 + 120


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under pbs_plus_amd/ or include/ may import,
    include, link or mention it."""
    bad = []
    for base in ("pbs_plus_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            dn[:] = [d for d in dn if d not in ("lib", "__pycache__")]
            for f in fn:
                if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                    text = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"\boracle\b", text) and not f.endswith(".md"):
                        hits = [l for l in text.splitlines() if re.search(r"\boracle\b", l)]
                        # comments that merely cite the oracle file as the checker are fine; code is not
                        code = [l for l in hits if re.search(r"import|include|from|dlopen|CDLL|-l", l)]
                        if code:
                            bad.append((f, code[:2]))
    assert not bad, bad


def test_library_load_sets_hw_queue_default_but_never_overrides():
    """libpbsgpu.so's load-time constructor exports GPU_MAX_HW_QUEUES=20 for hosts that bind the C ABI directly
    (the engine overlaps up to 16 batches on separate HIP streams; ROCm's default of 4 queues serialises them),
    and leaves an explicit setting of the host alone."""
    import subprocess
    import sys

    from pbs_plus_amd._lib import LIB_PATH

    code = ("import ctypes, sys\n"
            "ctypes.CDLL(%r)\n"
            "g = ctypes.CDLL(None).getenv; g.restype = ctypes.c_char_p\n"
            "print((g(b'GPU_MAX_HW_QUEUES') or b'').decode())\n" % LIB_PATH)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.stdout.strip() == "20", out.stdout + out.stderr
    out = subprocess.run([sys.executable, "-c", code], env=dict(env, GPU_MAX_HW_QUEUES="6"), capture_output=True,
                         text=True, timeout=120)
    assert out.stdout.strip() == "6", out.stdout + out.stderr


def test_option_structs_have_the_layout_the_bindings_assume(L, tmp_path):
    """ABI v5 moved the tuning values out of the environment into pbsgpu_engine_options / pbsgpu_ring_options: the ctypes
    mirrors (and through test_binding_sources the Go ones) must agree with what a C compiler makes of include/pbsgpu.h —
    sizes and the offset of every field."""
    import subprocess

    from pbs_plus_amd import _lib

    fields = {"pbsgpu_ring_options": _lib.RingOptions, "pbsgpu_engine_options": _lib.EngineOptions,
              "pbsgpu_ring_probe": _lib.RingProbe, "pbsgpu_ring_stats": _lib.RingStats}
    lines = []
    for cname, cls in fields.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "pbsgpu.h"\nint main(void) {\n' + "\n".join(lines) +
                   "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.rsplit(" ", 1) for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, cls in fields.items():
        assert int(got[cname + " size"]) == C.sizeof(cls), (cname, got[cname + " size"], C.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_the_library_reads_its_tuning_from_options_not_from_the_environment():
    """51 getenv() sites made split, thresholds, poll period, staging and priorities process-global
    state, invisible to a host with two engines. Since ABI v5 they are fields of the two options structs; what is left of
    getenv are two override TABLES (one per struct, for A/B runs of an unmodified binary) and a handful of process-level
    debug switches."""
    import re

    sites = []
    csrc = os.path.join(ROOT, "pbs_plus_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".cpp", ".hip", ".inc", ".h")):
            text = open(os.path.join(csrc, name)).read()
            text = re.sub(r"//[^\n]*", "", text)
            sites += [(name, m.start()) for m in re.finditer(r"\bgetenv\s*\(", text)]
    assert len(sites) <= 10, sites
