"""The experiment switches live in a second library (make exp -> libheadroom_hip_exp.so, -DHR_EXPERIMENTS); the shipped
library contains none of them (tests/test_abi_host.py::test_shipped_library_has_no_experiment_switches) and an
HR_ABLATE in the environment changes nothing in it."""
import json
import os
import subprocess
import sys

import pytest

import headroom_amd as hr
import signals

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EXP_LIB = os.path.join(os.path.dirname(hr.LIB_PATH), "libheadroom_hip_exp.so")


def run_worker(fs, **env):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, os.path.join(HERE, "exp_worker.py"), str(fs)], capture_output=True, text=True, env=e, timeout=110)
    assert p.returncode == 0, p.stderr[-2000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def reference(oracle, fs):
    pcm = signals.seeded_noise(31 + fs, fs * 9 + 321, 2, -12.0)
    pcm[:fs] *= 0.003
    return oracle.measure(pcm, fs)


def test_ablation_switch_does_nothing_in_the_shipped_library(oracle):
    fs = 44100
    ref = reference(oracle, fs)
    for val in ("3", "18", "257"):  # data-movement-only build, no DMA, never list candidates: all would corrupt the result
        got = run_worker(fs, HR_ABLATE=val, HR_FUSED_SPLIT="1", HR_WAVE_MAP="7")
        assert got["status"] == 0 and got["n_blocks"] == ref["n_blocks"] and got["n_gated"] == ref["n_gated"]
        assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
        assert got["peak_per_ch"] == ref["peak_per_ch"]


@pytest.mark.parametrize("fs", [44100, 48000])
def test_split_wave_kernel_of_the_experiments_library_is_bit_identical(oracle, fs):
    """HR_FUSED_SPLIT=1 in libheadroom_hip_exp.so (k_fused_split: K-weighting and true peak in separate waves)."""
    if not os.path.exists(EXP_LIB):
        pytest.skip("libheadroom_hip_exp.so is not built (make -C headroom_amd/csrc exp)")
    ref = reference(oracle, fs)
    base = run_worker(fs)
    got = run_worker(fs, HEADROOM_HIP_LIB=EXP_LIB, HR_FUSED_SPLIT="1")
    assert got["integrated_lufs"] == base["integrated_lufs"] and got["peak_per_ch"] == base["peak_per_ch"] == ref["peak_per_ch"]
    assert abs(got["integrated_lufs"] - ref["integrated_lufs"]) < 1e-6
    # and the ablation switch does act there (this is what the shipped library must not contain)
    wrong = run_worker(fs, HEADROOM_HIP_LIB=EXP_LIB, HR_ABLATE="3")
    assert wrong["peak_per_ch"] != ref["peak_per_ch"] or wrong["integrated_lufs"] != base["integrated_lufs"]
