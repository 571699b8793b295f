"""Host sanitizer target (CPU only): the library's translation units compiled for the host under AddressSanitizer +
UBSan and linked with a stand-in HIP runtime, driven through the C ABI over every order 1 .. 12 x 2^14 .. 2^22 samples x both
precisions x 1 / 4 / 16 / 64 records (tests/sanitize/walk.cpp has the list of what is checked).  Needs hipcc (the build
container has it; the run takes about a minute); a GPU is neither needed nor used."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="the host sanitizer build needs hipcc")
def test_host_code_under_asan_ubsan_over_every_layout():
    build = os.path.join(SAN, "_build")
    make = subprocess.run(["make", "-C", SAN, f"-j{min(8, os.cpu_count() or 1)}", f"HIPCC={HIPCC}"], capture_output=True, text=True, timeout=1200)
    assert make.returncode == 0, make.stdout[-2000:] + make.stderr[-4000:]
    tables = os.path.join(build, "tables.bin")
    gen = subprocess.run([sys.executable, os.path.join(SAN, "gen_tables.py"), tables], capture_output=True, text=True, timeout=600)
    assert gen.returncode == 0, gen.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([os.path.join(build, "walk"), tables], capture_output=True, text=True, timeout=1500, env=env)
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-6000:]
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr and "LeakSanitizer" not in run.stderr
    rec = json.loads(run.stdout.strip().splitlines()[-1])
    # 12 orders x 9 lengths x 2 precisions x 4 batch sizes, eight transform calls each (+ the atoms bank on a few)
    assert rec["ok"] and rec["plans"] == 864 and rec["calls"] >= 8 * 864 and rec["scratch_regions_checked"] > 60000
    # (every table of these shapes is one for the native engines, but float64 at 2^14 samples: hipFFT engine by choice)
    assert rec["plans_on_native_engines"] == 864 - 12 * 4
