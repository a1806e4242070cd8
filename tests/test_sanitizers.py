"""Sanitizer builds of the CPU-side code (the reference has the same two opt-in builds,
libzen/CMakeLists.txt:108-133, README.md:140-147).  Host code only: no GPU sanitizer exists on this pool.

CPU tier : oracle/zen_oracle.c under ASAN+UBSAN and under UBSAN alone (oracle/san_driver.c drives every
           entry point, including the in-place shifts and the read past size() of the offline driver that
           restate reference behaviour); the plain, ASAN and UBSAN builds must print one checksum.
           zen_amd/cli/wav.h under ASAN+UBSAN on truncated / malformed RIFF headers.
GPU tier : the C++ host mirror (zen_amd/libzen/hps.cpp) and the C++ reference suites (tests/cpp/test_libzen.cpp)
           built with -fsanitize=address,undefined against the uninstrumented libzen_hip.so and run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


def _run(cmd, env=None, timeout=600):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    return r.returncode, r.stdout, r.stderr


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(["make", "-s", "-C", ORACLE, "san_driver_asan", "san_driver_ubsan"])
    plain = str(tmp_path / "san_plain")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-o", plain,
                           os.path.join(ORACLE, "san_driver.c"), os.path.join(ORACLE, "zen_oracle.c"), "-lm"])
    procs = [subprocess.Popen([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
             for exe in (plain, os.path.join(ORACLE, "san_driver_asan"), os.path.join(ORACLE, "san_driver_ubsan"))]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
        assert "runtime error" not in err and "AddressSanitizer" not in err, err[-3000:]
    sums = [[ln for ln in out.splitlines() if ln.startswith("checksum")] for out, _ in outs]
    assert sums[0] and sums[0] == sums[1] == sums[2]
    assert outs[0][0] == outs[1][0] == outs[2][0]          # the refusals (bad sizes) print the same codes too


def test_wav_reader_rejects_malformed_headers_under_asan(tmp_path):
    exe = str(tmp_path / "test_wav")
    subprocess.check_call(["g++", "-std=c++17"] + SAN + [os.path.join(ROOT, "tests", "cpp", "test_wav.cpp"), "-o", exe])
    rc, out, err = _run([exe, str(tmp_path)])
    assert rc == 0 and "passed" in out, out + err[-3000:]


@pytest.mark.gpu
def test_cpp_host_mirror_under_asan_and_ubsan(tmp_path):
    from oracle import oracle as o
    from zen_amd import build
    o.build()
    build.build()
    exe = str(tmp_path / "test_libzen_san")
    zdir = os.path.join(ROOT, "zen_amd")
    obj = str(tmp_path / "zen_oracle_san.o")
    subprocess.check_call(["gcc", "-std=c99", "-ffp-contract=off"] + SAN +
                          ["-c", os.path.join(ORACLE, "zen_oracle.c"), "-o", obj])
    subprocess.check_call(["g++", "-std=c++17", "-ffp-contract=off"] + SAN +
                          ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(zdir, "libzen"),
                           os.path.join(ROOT, "tests", "cpp", "test_libzen.cpp"),
                           os.path.join(zdir, "libzen", "hps.cpp"), obj,
                           "-o", exe, "-L", zdir, "-lzen_hip", "-Wl,-rpath," + zdir, "-lm"])
    # README.md:140-147 of the reference: the GPU runtime maps memory the shadow must not guard
    env = dict(os.environ, ASAN_OPTIONS="protect_shadow_gap=0:replace_intrin=0:detect_leaks=0",
               UBSAN_OPTIONS="print_stacktrace=1")
    rc, out, err = _run([exe], env=env, timeout=900)
    assert rc == 0, out[-2000:] + err[-3000:]
    assert "runtime error" not in err and "AddressSanitizer" not in err, err[-3000:]
