"""The host build of the lane programs under UBSan + ASan: fp_inv (signed-limb division steps) against a * inv(a) == 1 on 2 000
random field elements and 0, hash_to_g2 against its two-lane form.  CPU only; skipped where the sanitizer runtimes are missing."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp_inv_and_message_stage_under_sanitizers(tmp_path):
    src = os.path.join(ROOT, "tests", "sanitize", "fp_inv_h2c_probe.cpp")
    exe = str(tmp_path / "probe")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=undefined,address", "-fno-sanitize-recover=undefined",
                         "-Wno-unknown-pragmas", "-I" + os.path.join(ROOT, "ethereum_consensus_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                         src, "-o", exe], capture_output=True, text=True)
    if cc.returncode != 0 and ("cannot find -lubsan" in cc.stderr or "cannot find -lasan" in cc.stderr or "libasan" in cc.stderr):
        pytest.skip("sanitizer runtimes not installed")
    assert cc.returncode == 0, cc.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "bad=0" in run.stdout, run.stdout[-500:] + run.stderr[-2000:]
