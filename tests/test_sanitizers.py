"""CPU-side native code under AddressSanitizer + UndefinedBehaviorSanitizer (container only; GPU sanitizers are not available on
the pool): the C oracle (oracle/*.c - what every parity test trusts) is rebuilt with -fsanitize=address,undefined and the
oracle-versus-reference tests are run against that build in a child interpreter; csrc/fsq_x87.h's host check likewise."""
import os
import shutil
import subprocess
import sys

import pytest

from _util import ROOT

SAN = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(shutil.which("gcc") is None or _libasan() is None, reason="no gcc / libasan here")
def test_oracle_under_asan_ubsan(tmp_path):
    src = [os.path.join(ROOT, "oracle", f) for f in ("fsq_oracle.c", "fsq_refmath.c", "fsq_register_oracle.c", "fsq_track_oracle.c")]
    lib = str(tmp_path / "libfsq_oracle_asan.so")
    subprocess.check_call(["gcc"] + SAN + ["-fPIC", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fopenmp",
                                           "-Wall", "-Wno-unused-function", "-shared", "-o", lib] + src + ["-lm"])
    env = dict(os.environ, LD_PRELOAD=_libasan(), ASAN_OPTIONS="detect_leaks=0", FSQ_ORACLE_LIB=lib)
    # the golden-vector tests of the oracle: LM fits of three fields, candidates, consolidation, the KATs, registration tuples,
    # both trackers (every branch of their bin / pair / window logic), the libm restatement
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "-x",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"), os.path.join(ROOT, "tests", "test_tracking.py"),
                        os.path.join(ROOT, "tests", "test_refmath.py"), "-k", "not x87"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0 and " passed" in p.stdout, tail
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, tail


@pytest.mark.skipif(shutil.which("g++") is None or _libasan() is None, reason="no g++ / libasan here")
def test_x87_restatement_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "x87_check_san")
    subprocess.check_call(["g++"] + SAN + ["-o", exe, os.path.join(ROOT, "tests", "x87_check.cpp")])
    p = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"), timeout=600)
    assert p.returncode == 0 and "bad=0" in p.stdout, p.stdout + p.stderr
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr
