"""CPU: the host-side C++ of libpgo (graph I/O, injector, synthetic generator, shard structure, halo plan) built with
AddressSanitizer + UndefinedBehaviorSanitizer and exercised by tests/native/host_sanitize_main.cpp.  (GPU ASan is not
available on the pool; the device code is covered by the parity tests.)"""
import os
import subprocess

from conftest import DATA, ROOT


def test_host_code_under_asan_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "toy-robust-backend-slam_amd", "csrc")
    exe = str(tmp_path / "host_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
           os.path.join(csrc, "host_graph.cpp"), os.path.join(csrc, "structure.cpp"),
           os.path.join(ROOT, "tests", "native", "host_sanitize_main.cpp"), "-o", exe]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, DATA], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "host sanitizer run ok" in p.stdout
