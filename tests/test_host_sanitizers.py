"""SURVEY.md section 5 "sanitizer build": the library's host-side algorithms compiled with AddressSanitizer +
UndefinedBehaviorSanitizer and run on seeded random inputs with functional checks (tests/host_asan.cpp): the slab
partition, the SpMV marching-unit planner (its records executed by a host emulation of the kernel's data flow and compared
with the stencil applied row by row), the geometry device functions of the capacity kernels (pg_geom.h) and the oracle's C
restatement of the Krylov loops.  CPU only: GPU sanitizers are not available on the pool."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "tests" / "_build"
FLAGS = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_algorithms_under_asan_ubsan():
    BUILD.mkdir(exist_ok=True)
    exe = BUILD / "host_asan"
    subprocess.run(["gcc", "-std=gnu11", *FLAGS, "-c", str(ROOT / "oracle" / "krylov_ref.c"), "-o", str(BUILD / "krylov_ref_asan.o")],
                   check=True, cwd=ROOT)
    subprocess.run(["g++", "-std=c++17", *FLAGS, "-c", str(ROOT / "tests" / "host_asan.cpp"), "-o", str(BUILD / "host_asan.o")],
                   check=True, cwd=ROOT)
    subprocess.run(["g++", "-fsanitize=address,undefined", str(BUILD / "host_asan.o"), str(BUILD / "krylov_ref_asan.o"), "-o", str(exe),
                    "-lm"], check=True, cwd=ROOT)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "all checks passed" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
