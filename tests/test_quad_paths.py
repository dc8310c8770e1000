"""The quad-staged gather / weight-gradient kernels and the LDS-transposed epilogue (csrc/conv2d.hip, HISTORY.md 4.0b) change
how operands reach LDS and how results leave the registers, not the arithmetic or its order: their results must be
BIT-identical to the dword-staged kernels.  The same holds for the second-generation transposed gather
(conv_igemm_tr2_kernel: quad staging, 16- or 8-channel chunks, LDS-transposed row stores) against conv_igemm_tr_kernel.  Two subprocesses (the knobs are read once per process), on the emulator and on
the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(which, **env):
    # (SAE_TR_SPLITK=0: the K split of conv_igemm_tr_kernel changes the summation order by design)
    # (SAE_WGRAD16=0: conv_wgrad16_kernel sums four pixels per instruction -- another order by design, checked against the
    # oracle in the kernel tests; this file holds the first-generation weight-gradient kernels, which remain the fallback)
    e = dict(os.environ, SAE_CONV_MATH="f32", SAE_CONV_THIN="0", SAE_TRACE_DISPATCH="1", SAE_TR_SPLITK="0", SAE_WGRAD16="0", **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "quad_worker.py"), which], cwd=ROOT, env=e,
                         capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "quad-done" in out.stdout
    # the stride-2 data gradients really took the kernel the knob names (tuning builds trace their dispatch)
    assert ("sae-dispatch tr2" in out.stderr) == (env.get("SAE_TR2") != "0"), out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l.startswith("(")]


def _compare(which):
    # (SAE_TR2_FLAT=2: flat tiles on every odd grid up to 65 wide, not only the 17-wide ones the product takes them for)
    quad = _run(which, SAE_IGEMM_QUAD="1", SAE_WGRAD_QUAD="1", SAE_IGEMM_VEC_STORE="2", SAE_TR2="3", SAE_TR2_FLAT="2")
    plain = _run(which, SAE_IGEMM_QUAD="0", SAE_WGRAD_QUAD="0", SAE_IGEMM_VEC_STORE="0", SAE_TR2="0")
    assert len(quad) == len(plain) and len(quad) > 0
    diff = [(a, b) for a, b in zip(quad, plain) if a != b]
    assert not diff, diff[:4]
    tr8 = _run(which, SAE_IGEMM_QUAD="1", SAE_WGRAD_QUAD="1", SAE_IGEMM_VEC_STORE="2", SAE_TR2="2")   # 8-channel chunks
    diff = [(a, b) for a, b in zip(tr8, plain) if a != b]
    assert not diff, diff[:4]


def test_quad_staging_is_bit_identical_on_the_emulator():
    _compare("emu")


@pytest.mark.gpu
def test_quad_staging_is_bit_identical_on_the_gpu():
    _compare("gpu")
