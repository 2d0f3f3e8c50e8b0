"""tools/validate_real.py - the one-command check a user runs the day REAL artefacts are at hand (none
exist offline: SURVEY.md fact 5) - on synthetic stand-ins: a checkpoint with the reference's key layout, a
bank in the reference's npz schema and a generated ``spherical_harmonics_ylm.py`` in the generator's
syntax.  Exit 0 = loaded through load_model and inside the test tolerances against the oracle; exit 2 =
a shape the kernels do not cover, named."""
import os
import subprocess
import sys

import pytest

from tools import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    return subprocess.run([sys.executable, os.path.join(REPO, "tools", "validate_real.py"), *args],
                          capture_output=True, text=True, timeout=600)


def test_validate_real_on_synthetic_artefacts(tmp_path):
    L = 20
    ck = synth.write_checkpoint(str(tmp_path / "e.ckpt"), L=L, hidden=256, seed=4)
    db = synth.write_bank(str(tmp_path / "db.npz"), 3000, seed=8)
    ylm = synth.write_ylm_source(str(tmp_path / "spherical_harmonics_ylm.py"), L)
    r = _run(ck, db, ylm)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "validate_real: OK" in r.stdout and "capacity H=256" in r.stdout
    # without the generated file: the regenerated table, checked against the oracle's reference-shaped evaluation
    r = _run(ck, db)
    assert r.returncode == 0 and "validate_real: OK" in r.stdout, r.stdout + r.stderr


def test_validate_real_widths_between_the_kernels_and_beyond(tmp_path):
    db = synth.write_bank(str(tmp_path / "db.npz"), 1500, seed=8)
    # a hidden width no kernel exists for runs zero-padded as the next one that has - and says so
    ck = synth.write_checkpoint(str(tmp_path / "e576.ckpt"), L=10, hidden=576, seed=4)
    r = _run(ck, db)
    assert r.returncode == 0 and "runs zero-padded as the kernel width 768" in r.stdout, r.stdout + r.stderr
    # beyond 1024: refused, loudly, with the limit named
    ck = synth.write_checkpoint(str(tmp_path / "e1088.ckpt"), L=10, hidden=1088, seed=4)
    r = _run(ck, db)
    assert r.returncode == 2 and "UNSUPPORTED" in r.stderr and "1024" in r.stderr, r.stdout + r.stderr


def test_c_abi_from_a_plain_program(tmp_path):
    """librange_hip.so from a program without Python or torch (tests/native/abi_smoke.cpp): hipMalloc'ed
    buffers, its own stream, status codes and range_last_error - the boundary as a maintainer of another
    host language would bind it."""
    import shutil
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "abi_smoke")
    lib_dir = os.path.join(REPO, "range_amd")
    b = subprocess.run(["hipcc", "-O1", "-std=c++17", "-o", exe, os.path.join(REPO, "tests", "native", "abi_smoke.cpp"),
                        "-I" + os.path.join(REPO, "include"), "-L" + lib_dir, "-lrange_hip", "-Wl,-rpath," + lib_dir],
                       capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "abi_smoke ok" in r.stdout, r.stdout + r.stderr
