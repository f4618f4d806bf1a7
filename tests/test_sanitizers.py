"""CPU sanitizer run of the library's threaded host code (the reference has no race detection; GPU
sanitizers are not available on the pool): tests/sanitize/Makefile builds host_path.hip (staging ring + copy pool),
blmm_multi.hip (one worker thread per device, every gather mode) and readers.hip (CSV / Helium parsers) as plain C++ against
a CPU stand-in for the HIP runtime, once with -fsanitize=address,undefined and once with -fsanitize=thread, and runs both."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("make") is None, reason="no g++ / make")
def test_host_code_under_asan_ubsan_and_tsan(tmp_path):
    r = subprocess.run(["make", "-C", os.path.join(HERE, "sanitize"), "check", f"TMP={tmp_path}"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("sanitize: copy pool, multi-device workers, readers ok") == 2
