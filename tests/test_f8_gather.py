"""conv_igemm_f8_kernel (fp32, 8 waves, LDS-DMA weights; opt-in with SAE_F8=1, see csrc/conv2d.hip): small shapes forced
onto it (SAE_F8_MIN_TILES=1) against the oracle, on the emulator (CPU) and on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(which):
    env = dict(os.environ, SAE_F8="1", SAE_F8_MIN_TILES="1", SAE_CONV_MATH="f32")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "f8_worker.py"), which], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "f8-ok" in out.stdout


def test_f8_kernel_on_the_emulator():
    _run("emu")


@pytest.mark.gpu
def test_f8_kernel_on_the_gpu():
    _run("gpu")
