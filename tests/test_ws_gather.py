"""conv_igemm_ws_kernel (csrc/conv2d.hip): the 64 x 256 tile of the 3x3 stride-1 gather with four consumer waves (MFMAs only) and
two producer waves (LDS-DMA staging into a second buffer) keeps the arithmetic and its order: results BIT-identical to
conv_igemm_kernel.  Two subprocesses (SAE_WS is read once per process), on the emulator and on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(which, ws):
    env = dict(os.environ, SAE_CONV_MATH="f32", SAE_TRACE_DISPATCH="1", SAE_WS=ws)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ws_worker.py"), which], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "ws-done" in out.stdout
    assert ("sae-dispatch ws " in out.stderr) == (ws == "1"), out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l.startswith("(")], out.stderr.count("sae-dispatch ws ")


def _compare(which):
    ws, launches = _run(which, "1")
    plain, _ = _run(which, "0")
    assert len(ws) == len(plain) and len(ws) > 0
    assert launches >= len(ws) // 2, launches       # most of the launches really took the kernel
    diff = [(a, b) for a, b in zip(ws, plain) if a != b]
    assert not diff, diff[:4]


def test_wave_specialised_gather_is_bit_identical_on_the_emulator():
    _compare("emu")


@pytest.mark.gpu
def test_wave_specialised_gather_is_bit_identical_on_the_gpu():
    _compare("gpu")
