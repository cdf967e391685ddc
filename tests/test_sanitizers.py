"""CPU hardening: the whole non-GPU suite again, against the AddressSanitizer + UBSan builds of the host objects
(`make -C raytracinginrust_amd/csrc asan`: C-ABI, flattener, JPEG + OBJ ingest) and of the oracle (`make -C oracle asan`).  The
libraries are swapped in through RT_AMD_LIB / ORC_LIB and the sanitizer runtimes are preloaded into the child interpreter; any
report aborts the child.  CPU only: GPU sanitizers are not available on the MI355X pool."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _runtime(name):
    p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(os.environ.get("RT_SANITIZER_RUN") == "1", reason="already inside the sanitizer run")
def test_cpu_suite_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "raytracinginrust_amd", "csrc"), "asan"])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ)
    env.update({
        "LD_PRELOAD": f"{asan} {ubsan}",
        "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1",          # CPython itself leaks by design; everything else is fatal
        "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1",
        "RT_AMD_LIB": os.path.join(ROOT, "raytracinginrust_amd", "csrc", "_asan", "librt_amd_asan.so"),
        "ORC_LIB": os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so"),
        "RT_SANITIZER_RUN": "1",
    })
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-m", "not gpu", "-x", "-q", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout
