"""Multi-tensor Adam (csrc/adam.hip, fused_adam.FusedAdam): the oracle is pinned to torch.optim.Adam — what the
reference constructs at optimizers/swapping_autoencoder_optimizer.py:34-42 — the emulated kernel and (``-m gpu``)
the real kernel to the oracle, and the optimizer class to torch.optim.Adam's trajectories and state_dict layout."""
import copy

import numpy as np
import pytest
import torch

import abi_harness as H
import parity_common as P

SIZES = [1, 3, 4, 5, 1023, 4096, 65536, 65537, 200001, 0, 7]          # body / tail / chunk-boundary / empty tensors


def _problem(sizes, seed):
    rng = np.random.default_rng(seed)
    ps = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    gs = [(rng.standard_normal(n) * 10.0 ** rng.uniform(-4, 1)).astype(np.float32) for n in sizes]
    ms = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in sizes]
    vs = [(0.01 * rng.random(n)).astype(np.float32) for n in sizes]
    steps = [int(rng.integers(1, 40)) for _ in sizes]
    return ps, gs, ms, vs, steps


def _torch_adam(ps, gs, ms, vs, steps, lr, b1, b2, eps, scale):
    out = []
    for p, g, m, v, t in zip(ps, gs, ms, vs, steps):
        tp = torch.nn.Parameter(torch.from_numpy(p.copy()).double())
        opt = torch.optim.Adam([tp], lr=lr, betas=(b1, b2), eps=eps)
        tp.grad = torch.from_numpy(g.copy()).double() * scale
        if p.size:
            opt.state[tp] = {"step": torch.tensor(float(t - 1)), "exp_avg": torch.from_numpy(m.copy()).double(),
                             "exp_avg_sq": torch.from_numpy(v.copy()).double()}
            opt.step()
            st = opt.state[tp]
            out.append((tp.detach().numpy(), st["exp_avg"].numpy(), st["exp_avg_sq"].numpy()))
        else:
            out.append((p, m, v))
    return out


@pytest.mark.parametrize("hp", [(0.002, 0.0, 0.99, 1e-8, 1.0), (0.00188, 0.0, 0.9905, 1e-8, 0.125), (1e-3, 0.9, 0.999, 1e-8, 1.0)], ids=str)
def test_oracle_is_torch_adam(oracle_lib, hp):
    """oracle_adam_multi_f32 == torch.optim.Adam (double precision run of torch) to fp32 round-off of the outputs."""
    lr, b1, b2, eps, scale = hp
    ps, gs, ms, vs, steps = _problem(SIZES, 1)
    want = _torch_adam(ps, gs, ms, vs, steps, lr, b1, b2, eps, scale)
    got_p, got_m, got_v = H.adam_multi(oracle_lib, ps, gs, ms, vs, steps, lr, b1, b2, eps, scale)
    for i, (wp, wm, wv) in enumerate(want):
        assert np.allclose(got_p[i], wp, rtol=2e-7, atol=1e-9), i
        assert np.allclose(got_m[i], wm, rtol=2e-7, atol=1e-12), i
        assert np.allclose(got_v[i], wv, rtol=2e-7, atol=1e-20), i


def _check_against_oracle(lib, oracle_lib, device, sizes, offset):
    lr, b1, b2, eps, scale = 0.00188, 0.0, 0.9905, 1e-8, 0.5
    ps, gs, ms, vs, steps = _problem(sizes, 2)
    o_p, o_m, o_v = H.adam_multi(oracle_lib, ps, gs, ms, vs, steps, lr, b1, b2, eps, scale)
    k_p, k_m, k_v = H.adam_multi(lib, ps, gs, ms, vs, steps, lr, b1, b2, eps, scale, device=device, offset_elems=offset)
    for i in range(len(sizes)):
        # fp32 evaluation of the same formula: a few ulp on m and v; the parameter moves by <= lr per step, so its
        # error is bounded by lr * (relative error of m / denom)
        assert np.allclose(k_m[i], o_m[i], rtol=5e-7, atol=1e-12), i
        assert np.allclose(k_v[i], o_v[i], rtol=5e-7, atol=1e-20), i
        assert np.abs(k_p[i] - o_p[i]).max(initial=0.0) <= 1e-6 * lr + 2e-7 * np.abs(o_p[i]).max(initial=0.0), i


@pytest.mark.parametrize("offset", [0, 1])
def test_emulated_kernel_vs_oracle(emu_lib, oracle_lib, offset):
    _check_against_oracle(emu_lib, oracle_lib, None, SIZES, offset)


def test_emulated_kernel_many_tensors(emu_lib, oracle_lib):
    """more tensors than one argument block holds (24) and more chunks than one launch holds (224 x 64 Ki)"""
    _check_against_oracle(emu_lib, oracle_lib, None, [257] * 60 + [3], 0)


def _device_steps_equal_host_steps(lib, device, sizes, exact):
    """sae_adam_multi_dev_f32 (counts in device memory, bias corrections formed by the kernel) against sae_adam_multi_f32 (counts
    as host arguments) on the same problem; the counts come back advanced by one."""
    lr, b1, b2, eps, scale = 0.00188, 0.0, 0.9905, 1e-8, 0.5
    ps, gs, ms, vs, steps = _problem(sizes, 4)
    h_p, h_m, h_v = H.adam_multi(lib, ps, gs, ms, vs, steps, lr, b1, b2, eps, scale, device=device)
    d_p, d_m, d_v, after = H.adam_multi(lib, ps, gs, ms, vs, steps, lr, b1, b2, eps, scale, device=device, dev_steps=True)
    assert after == steps
    for i in range(len(sizes)):
        assert np.array_equal(d_m[i], h_m[i]) and np.array_equal(d_v[i], h_v[i]), i      # no bias correction in the moments
        if exact:
            assert np.array_equal(d_p[i], h_p[i]), i
        else:   # pow() of the device library against the host's: the two fp32 scalars may differ in their last bit
            assert np.abs(d_p[i] - h_p[i]).max(initial=0.0) <= 2e-7 * lr + 1e-7 * np.abs(h_p[i]).max(initial=0.0), i


def test_device_step_counts_on_the_emulator_and_the_oracle(emu_lib, oracle_lib):
    _device_steps_equal_host_steps(emu_lib, None, SIZES, exact=True)
    _device_steps_equal_host_steps(emu_lib, None, [257] * 400 + [3], exact=True)      # more counters than one advance launch holds
    _device_steps_equal_host_steps(oracle_lib, None, SIZES, exact=True)


def test_fused_adam_with_device_step_counts(oracle_lib):
    """FusedAdam.use_device_steps(): same trajectory as the host-count form, state_dict() reports the device counts, a
    checkpoint loads back into them."""
    from swapping_autoencoder_pytorch_amd.fused_adam import FusedAdam
    torch.manual_seed(1)
    net_a = torch.nn.Sequential(torch.nn.Linear(5, 9), torch.nn.Tanh(), torch.nn.Linear(9, 2))
    net_b = copy.deepcopy(net_a)
    x = torch.randn(16, 5)
    with P.backend(oracle_lib):
        host = FusedAdam(list(net_a.parameters()), lr=0.002, betas=(0.0, 0.99))
        dev = FusedAdam(list(net_b.parameters()), lr=0.002, betas=(0.0, 0.99))

        def step_both():
            for net, opt in ((net_a, host), (net_b, dev)):
                opt.zero_grad()
                net(x).pow(2).mean().backward()
                opt.step()
            for pa, pb in zip(net_a.parameters(), net_b.parameters()):
                assert torch.equal(pa, pb)

        step_both()
        dev.use_device_steps()               # switches after the first update: the counts are carried over
        for _ in range(3):
            step_both()
        sd = dev.state_dict()
        assert [float(v["step"]) for v in sd["state"].values()] == [4.0] * 4
        assert [float(v["step"]) for v in host.state_dict()["state"].values()] == [4.0] * 4
        again = FusedAdam(list(net_b.parameters()), lr=0.002, betas=(0.0, 0.99))
        again.use_device_steps()
        again.load_state_dict(copy.deepcopy(sd))
        dev = again
        step_both()
        assert [float(v["step"]) for v in dev.state_dict()["state"].values()] == [5.0] * 4


def test_bad_arguments_are_refused(oracle_lib, emu_lib):
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeError
    ps, gs, ms, vs, steps = _problem([8], 3)
    for lib in (oracle_lib, emu_lib):
        with pytest.raises(SaeError):
            H.adam_multi(lib, ps, gs, ms, vs, [0], 1e-3, 0.0, 0.99, 1e-8)          # step counts are 1-based
        with pytest.raises(SaeError):
            H.adam_multi(lib, ps, gs, ms, vs, steps, 1e-3, 1.0, 0.99, 1e-8)        # beta1 = 1


def test_fused_adam_follows_torch_adam_and_shares_its_state_dict(oracle_lib):
    from swapping_autoencoder_pytorch_amd.fused_adam import FusedAdam
    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    net_b = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    net_b.load_state_dict(net_a.state_dict())
    frozen = torch.nn.Parameter(torch.ones(4))                       # never gets a gradient: skipped, step stays 0
    ref = torch.optim.Adam(list(net_a.parameters()), lr=0.002, betas=(0.0, 0.99))
    with P.backend(oracle_lib):
        ours = FusedAdam(list(net_b.parameters()) + [frozen], lr=0.002, betas=(0.0, 0.99))
        x = torch.randn(32, 6)

        def same_gradient_step(opt_a, opt_b):
            """both optimisers see the SAME gradients (those of net_a): compares the update rule itself, without the
            feedback through which Adam(beta1 = 0) turns round-off in tiny gradients into diverging trajectories"""
            opt_a.zero_grad()
            net_a(x).pow(2).mean().backward()
            for pa, pb in zip(net_a.parameters(), net_b.parameters()):
                pb.grad = pa.grad.clone()
            opt_a.step()
            opt_b.step()
            for pa, pb in zip(net_a.parameters(), net_b.parameters()):
                assert torch.allclose(pa, pb, rtol=0, atol=2e-7), float((pa - pb).abs().max())

        for it in range(5):
            same_gradient_step(ref, ours)
        assert frozen not in ours.state or len(ours.state[frozen]) == 0
        # an optimiser checkpoint of torch.optim.Adam loads into FusedAdam (and back) and continues identically
        fresh = FusedAdam(list(net_b.parameters()), lr=0.002, betas=(0.0, 0.99))
        fresh.load_state_dict(copy.deepcopy(ref.state_dict()))     # (load_state_dict shares the tensors it is given)
        for it in range(2):
            same_gradient_step(ref, fresh)
        back = torch.optim.Adam(list(net_b.parameters()), lr=0.002, betas=(0.0, 0.99))
        back.load_state_dict(copy.deepcopy(fresh.state_dict()))
        same_gradient_step(ref, back)
    with pytest.raises(Exception):
        FusedAdam(list(net_b.parameters()), lr=0.002, weight_decay=0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("offset", [0, 1, 3])
def test_gpu_kernel_vs_oracle(oracle_lib, offset):
    from swapping_autoencoder_pytorch_amd import hip_lib
    _check_against_oracle(hip_lib.get(), oracle_lib, "cuda:0", SIZES + [3_000_001], offset)


@pytest.mark.gpu
def test_gpu_device_step_counts():
    from swapping_autoencoder_pytorch_amd import hip_lib
    _device_steps_equal_host_steps(hip_lib.get(), "cuda:0", SIZES + [3_000_001], exact=False)
    _device_steps_equal_host_steps(hip_lib.get(), "cuda:0", [257] * 400 + [3], exact=False)


@pytest.mark.gpu
def test_gpu_kernel_many_tensors(oracle_lib):
    from swapping_autoencoder_pytorch_amd import hip_lib
    _check_against_oracle(hip_lib.get(), oracle_lib, "cuda:0", [257] * 60 + [65536 * 230 + 5, 3], 0)
