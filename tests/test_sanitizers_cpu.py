"""CPU-side sanitizer runs: the C oracle and the library's host-only code under AddressSanitizer + UBSan.
Never on the GPU box's device code (GPU ASan / XNACK builds are not available on this pool); these run wherever gcc does."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


def _need(tool):
    if shutil.which(tool) is None:
        pytest.skip(tool + " not installed")


def _clean(r):
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]


def test_oracle_golden_suite_under_asan_ubsan(tmp_path):
    """oracle/bg_oracle.c compiled with -fsanitize=address,undefined (oracle/Makefile: libbg_oracle_asan.so) replays the
    tests.cpp known answers, fixtures G1-G5 (1 380 edge calls, 200 random and 24 greedy reference games) and 800 env steps
    of the lane driver: no out-of-bounds access, no undefined shift / overflow in the restatement the parity tests trust."""
    _need("gcc"); _need("make")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libbg_oracle_asan.so"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, **SAN_ENV, LD_PRELOAD=libasan, BG_ORACLE_LIB=os.path.join(ROOT, "oracle", "libbg_oracle_asan.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "sanitize", "oracle_driver.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    _clean(r)
    assert r.stdout.strip().endswith("OK")


def test_library_host_code_under_asan_ubsan(tmp_path):
    """bgamd_td_stream_schedule is the library's one piece of host-only logic (csrc/bg_schedule.h, plain C++): compiled with the
    sanitizers and driven over 300 random rounds (ragged lengths, empty lanes, more slots than games, invalid arguments)."""
    _need("g++")
    exe = tmp_path / "schedule_driver"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(ROOT, "backgammon-engine_amd", "csrc"),
                           os.path.join(ROOT, "tests", "sanitize", "schedule_driver.cpp"), "-o", str(exe)])
    r = subprocess.run([str(exe)], env=dict(os.environ, **SAN_ENV), capture_output=True, text=True, timeout=300)
    _clean(r)
    assert r.stdout.startswith("OK")
