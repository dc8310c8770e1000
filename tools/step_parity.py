"""Observed step-level deviations from the reference's golden run (micro preset; D, G, D + R1, G steps).

    python tools/step_parity.py cpu  > profiles/r2_step_parity_cpu.json     (build container: oracle, emulator, controls)
    python tools/step_parity.py gpu  > gpurun_out/r2_step_parity_gpu.json   (GPU box: f32 and bf16x6 kernels)

Back ends: the CPU oracle (double accumulation), the emulator build of the product's kernels (the same fp32
arithmetic the GPU runs), the real library under both conv arithmetics.  Controls: the ORACLE back end with
1e-7 / 1e-6 relative noise on the input images — one implementation, fp32-level input noise — which measures how
much of the later-step deviation is the conditioning of Adam(beta1 = 0) rather than an implementation difference."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_common as P  # noqa: E402
from swapping_autoencoder_pytorch_amd import hip_lib  # noqa: E402
from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary  # noqa: E402


def slim(m):
    return {"max_loss_dev_per_step": [r["max_loss_dev"] for r in m["steps"]],
            "max_grad_norm_dev_per_step": [r["max_grad_norm_dev"] for r in m["steps"]],
            "worst_grad_per_step": [r["worst_grad"] for r in m["steps"]],
            "max_param_norm_dev_after": m["max_param_norm_dev_after"]}


def main(where):
    out = {}
    if where == "cpu":
        ora = SaeLibrary(os.path.join(ROOT, "oracle", "libsae_oracle.so"), prefix="oracle_", device_only=False)
        with P.backend(ora):
            out["oracle"] = slim(P.measure_micro_steps("cpu"))
            out["oracle_input_noise_1e-7"] = slim(P.measure_micro_steps("cpu", perturb=1e-7))
            out["oracle_input_noise_1e-6"] = slim(P.measure_micro_steps("cpu", perturb=1e-6))
        from emu import build_emu
        emu = SaeLibrary(build_emu.build(), prefix="sae_", device_only=False)
        with P.backend(emu):
            out["emulated_hip_kernels_f32"] = slim(P.measure_micro_steps("cpu"))
    else:
        for mode in ("f32", "bf16x6"):
            hip_lib.set_conv_math(mode)
            out["gpu_" + mode] = slim(P.measure_micro_steps("cuda:0"))
            full = P.measure_micro_steps("cuda:0")
            out["gpu_" + mode + "_loss_dev_by_key"] = [r["loss_dev"] for r in full["steps"]]
        hip_lib.set_conv_math("f32")
        out["gpu_f32_torch_adam"] = slim(P.measure_micro_steps("cuda:0", fused_adam=False))
        hip_lib.set_conv_math("f32")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "cpu")
