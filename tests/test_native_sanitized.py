"""Builds the host-only translation units of libpbsgpu (hostonly.cpp, reuse.cpp) together with a C test
driver under AddressSanitizer + UndefinedBehaviorSanitizer and runs it — the native-code counterpart of the
reference's `go test -race` discipline (.github/workflows/go-tests.yml:24-33)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_only_code_under_asan_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "pbs_plus_amd", "csrc")
    exe = str(tmp_path / "test_hostonly")
    flags = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-Wall", "-Wextra"]
    objs = []
    for src in ("hostonly.cpp", "reuse.cpp"):
        o = str(tmp_path / (src + ".o"))
        subprocess.run(["g++", "-std=c++17", *flags, "-c", os.path.join(csrc, src), "-o", o], check=True)
        objs.append(o)
    o = str(tmp_path / "driver.o")
    subprocess.run(["gcc", "-std=c11", *flags, "-c", os.path.join(ROOT, "tests", "native", "test_hostonly.c"), "-o", o], check=True)
    subprocess.run(["g++", *flags, o, *objs, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1"))
    assert out.returncode == 0 and "native-host-ok" in out.stdout, out.stdout + out.stderr


def test_cpp_mirror_reader_under_asan(tmp_path):
    """include/pbsgpu.hpp, host-only half (NewConfig error return, ParseDynamicIndex, ChunkFromOffset)."""
    csrc = os.path.join(ROOT, "pbs_plus_amd", "csrc")
    exe = str(tmp_path / "test_cpp_reader")
    flags = ["-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra"]
    subprocess.run(["g++", *flags, os.path.join(ROOT, "tests", "native", "test_cpp_reader.cpp"),
                    os.path.join(csrc, "hostonly.cpp"), os.path.join(csrc, "reuse.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "cpp-reader-ok" in out.stdout, out.stdout + out.stderr


def test_oracle_under_asan_ubsan(tmp_path):
    """The parity checker itself, memory- and UB-clean on exact-size buffers."""
    exe = str(tmp_path / "test_oracle_asan")
    flags = ["-std=c11", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra"]
    srcs = [os.path.join(ROOT, "tests", "native", "test_oracle_asan.c"), os.path.join(ROOT, "oracle", "buzhash_oracle.c"),
            os.path.join(ROOT, "oracle", "sha256_oracle.c")]
    subprocess.run(["gcc", *flags, *srcs, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "oracle-asan-ok" in out.stdout, out.stdout + out.stderr
