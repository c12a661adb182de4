"""The host-side native code of the library (smrt_amd/csrc/host_assemble.cpp: output assembly, finiteness sweep) under
AddressSanitizer + UndefinedBehaviorSanitizer: compiled with g++ next to a small driver and run on the CPU (GPU
sanitizers are not available on this pool; host code is where they can run)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_entry_points_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_driver")
    # the sanitizer runtimes are linked statically so that nothing depends on their place in the library load order
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-static-libasan", "-static-libubsan",
           os.path.join(ROOT, "tests", "native", "host_driver.cpp"),
           os.path.join(ROOT, "smrt_amd", "csrc", "host_assemble.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and "host driver ok" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]
