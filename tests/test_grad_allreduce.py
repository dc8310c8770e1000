"""CPU, world_size = 2 over gloo: the bucketed gradient all-reduce of grad_allreduce.py gives every
rank the global-batch gradient, for both alternating parameter groups, with a parameter that gets
no gradient, and for the R1 pattern (autograd.grad(create_graph=True) followed by backward())."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16))   # "G" group
        self.b = torch.nn.Sequential(torch.nn.Linear(16, 12), torch.nn.Tanh(), torch.nn.Linear(12, 1))   # "D" group
        self.unused = torch.nn.Parameter(torch.zeros(3))                                                # never gets a grad

    def forward(self, x):
        return self.b(self.a(x))


def _loss(net, x):
    return net(x).pow(2).mean(dim=1)      # per-sample, like the reference's losses


def _r1(net, x):
    x = x.clone().requires_grad_()
    g, = torch.autograd.grad(net(x).sum(), x, create_graph=True)
    return g.pow(2).sum(dim=1)


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from swapping_autoencoder_pytorch_amd.grad_allreduce import GradAllReducer, broadcast_parameters
    torch.manual_seed(100 + rank)            # deliberately different replicas before the broadcast
    net = Net()
    broadcast_parameters(net)
    torch.manual_seed(0)
    full = torch.randn(8, 6)
    shard = full[rank * 4:(rank + 1) * 4]
    group_a = list(net.a.parameters())
    group_b = list(net.b.parameters()) + [net.unused]
    red_a = GradAllReducer(group_a, bucket_bytes=256)      # tiny buckets -> several collectives
    red_b = GradAllReducer(group_b, bucket_bytes=256)
    assert red_a.enabled and len(red_a.buckets) > 1
    # a rank of a multi-rank job runs the step on ONE stream unless told otherwise (streams.enabled)
    from swapping_autoencoder_pytorch_amd import streams
    os.environ.pop("SAE_TWO_STREAMS", None)
    assert not streams.enabled()
    os.environ["SAE_TWO_STREAMS"] = "1"
    assert streams.enabled()
    os.environ.pop("SAE_TWO_STREAMS")
    launched_in_backward = []
    results = {}
    for tag, params, frozen, red, fn in (("a", group_a, group_b, red_a, _loss), ("b", group_b, group_a, red_b, _loss),
                                         ("b_r1", group_b, group_a, red_b, _r1)):
        for p in params:
            p.requires_grad_(True)
            p.grad = None
        for p in frozen:
            p.requires_grad_(False)
        red.arm()
        fn(net, shard).mean().backward()
        launched_in_backward.append(sum(b.work is not None for b in red.buckets))
        red.finish()
        results[tag] = [None if p.grad is None else p.grad.clone() for p in params]
    # complete buckets are in flight when backward returns (launched from the grad-ready hooks); the one holding the parameter that
    # never gets a gradient is launched by finish()
    assert max(launched_in_backward) > 0, launched_in_backward
    state = {k: v.clone() for k, v in net.state_dict().items()}
    # finish_into: the multi-tensor Adam reads the summed buckets in place (oracle bound behind the C-ABI: test only)
    from swapping_autoencoder_pytorch_amd import hip_lib
    from swapping_autoencoder_pytorch_amd.fused_adam import FusedAdam
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hip_lib._LIB = SaeLibrary(os.path.join(root, "oracle", "libsae_oracle.so"), prefix="oracle_", device_only=False)
    for p in group_a:
        p.requires_grad_(True)
        p.grad = None
    for p in group_b:
        p.requires_grad_(False)
    adam = FusedAdam(group_a, lr=0.01, betas=(0.0, 0.99))
    for it in range(2):
        adam.zero_grad()
        red_a.arm()
        _loss(net, shard).mean().backward()
        red_a.finish_into(adam)
    torch.save({"state": state, "grads": results, "after_adam": net.state_dict()}, os.path.join(tmp, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    # replicas were synchronised by the broadcast
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), k
    # single-process global-batch reference
    net = Net()
    net.load_state_dict(r0["state"])
    torch.manual_seed(0)
    full = torch.randn(8, 6)
    group_a = list(net.a.parameters())
    group_b = list(net.b.parameters()) + [net.unused]
    for tag, params, fn in (("a", group_a, _loss), ("b", group_b, _loss), ("b_r1", group_b, _r1)):
        for p in net.parameters():
            p.grad = None
            p.requires_grad_(True)
        fn(net, full).mean().backward()
        for i, p in enumerate(params):
            for r in (r0, r1):
                got = r["grads"][tag][i]
                if p.grad is None:      # the never-used parameter: zeros (or nothing) on every rank
                    assert got is None or float(got.abs().max()) == 0.0
                else:
                    assert torch.allclose(got, p.grad, rtol=1e-5, atol=1e-6), (tag, i)


    # two Adam steps through finish_into == torch.optim.Adam on the global-batch gradient, identical on both ranks
    net.load_state_dict(r0["state"])
    for p in net.parameters():
        p.grad = None
        p.requires_grad_(True)
    ref = torch.optim.Adam(group_a, lr=0.01, betas=(0.0, 0.99))
    for it in range(2):
        ref.zero_grad()
        for p in group_b:
            p.grad = None
        _loss(net, full).mean().backward()
        ref.step()
    for k, v in net.state_dict().items():
        assert torch.equal(r0["after_adam"][k], r1["after_adam"][k]), k
        assert torch.allclose(r0["after_adam"][k], v, rtol=1e-5, atol=2e-6), k


def test_single_process_is_a_noop(monkeypatch):
    from swapping_autoencoder_pytorch_amd import streams
    from swapping_autoencoder_pytorch_amd.grad_allreduce import GradAllReducer
    monkeypatch.delenv("SAE_TWO_STREAMS", raising=False)
    assert streams.enabled()                    # single rank: the step's branches on two streams
    monkeypatch.setenv("SAE_TWO_STREAMS", "0")
    assert not streams.enabled()
    monkeypatch.delenv("SAE_TWO_STREAMS")
    net = Net()
    red = GradAllReducer(list(net.parameters()))
    assert not red.enabled
    red.arm()
    _loss(net, torch.randn(4, 6)).mean().backward()
    red.finish()
    assert net.a[0].weight.grad is not None


class _ConvNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        from swapping_autoencoder_pytorch_amd.stylegan2_layers import ConvLayer
        self.g = torch.nn.Sequential(ConvLayer(3, 8, 3), ConvLayer(8, 8, 3))
        self.d = torch.nn.Sequential(ConvLayer(8, 8, 3), ConvLayer(8, 1, 1, activate=False))


def _dropin_worker(rank, world, port, tmp, fused):
    """The reference's loop shape (zero_grad -> backward -> step, optimizers/swapping_autoencoder_optimizer.py:69-107) driven through
    dropin.attach_gradient_allreduce: from the second iteration on the conv weights' gradients must still be produced inside
    their bucket slots (arming right after step() found every .grad alive and registered no destination)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from swapping_autoencoder_pytorch_amd import dropin, grad_allreduce, hip_lib
    from swapping_autoencoder_pytorch_amd.fused_adam import FusedAdam
    from swapping_autoencoder_pytorch_amd.hip_lib import SaeLibrary
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hip_lib._LIB = SaeLibrary(os.path.join(root, "oracle", "libsae_oracle.so"), prefix="oracle_", device_only=False)
    torch.manual_seed(7 + rank)
    net = _ConvNet()
    gp, dp = list(net.g.parameters()), list(net.d.parameters())
    make = (lambda ps: FusedAdam(ps, lr=0.01, betas=(0.0, 0.99))) if fused else (lambda ps: torch.optim.Adam(ps, lr=0.01, betas=(0.0, 0.99)))
    host = types.SimpleNamespace(Gparams=gp, Dparams=dp, optimizer_G=make(gp), optimizer_D=make(dp),
                                 model=types.SimpleNamespace(singlegpu_model=net), save=lambda *a, **k: None)
    dropin.attach_gradient_allreduce(host)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    torch.manual_seed(0)
    full = torch.randn(4, 3, 8, 8)
    shard = full[rank * 2:(rank + 1) * 2]
    claimed = []
    for it in range(3):
        for params, frozen, opt in ((dp, gp, host.optimizer_D), (gp, dp, host.optimizer_G)):
            for p in params:
                p.requires_grad_(True)
            for p in frozen:
                p.requires_grad_(False)
            before = grad_allreduce.CLAIMED[0]
            opt.zero_grad()
            net.d(net.g(shard)).square().mean().backward()
            opt.step()
            claimed.append(grad_allreduce.CLAIMED[0] - before)
    torch.save({"state": state, "after": net.state_dict(), "claimed": claimed}, os.path.join(tmp, "rank%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("fused", [True, False])
def test_two_rank_dropin_loop_keeps_writing_conv_gradients_into_the_buckets(tmp_path, fused, oracle_lib):
    world, port = 2, _free_port()
    mp.spawn(_dropin_worker, args=(world, port, str(tmp_path), fused), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    # every pass of every iteration hands out the two conv-weight slots of its group: [D, G] x 3 iterations
    assert r0["claimed"] == [2] * 6 and r1["claimed"] == [2] * 6, (r0["claimed"], r1["claimed"])
    for k in r0["after"]:
        assert torch.equal(r0["after"][k], r1["after"][k]), k
    # the same three iterations in one process on the global batch
    from parity_common import backend
    with backend(oracle_lib):
        net = _ConvNet()
        net.load_state_dict(r0["state"])
        gp, dp = list(net.g.parameters()), list(net.d.parameters())
        og, od = torch.optim.Adam(gp, lr=0.01, betas=(0.0, 0.99)), torch.optim.Adam(dp, lr=0.01, betas=(0.0, 0.99))
        torch.manual_seed(0)
        full = torch.randn(4, 3, 8, 8)
        for it in range(3):
            for params, frozen, opt in ((dp, gp, od), (gp, dp, og)):
                for p in params:
                    p.requires_grad_(True)
                for p in frozen:
                    p.requires_grad_(False)
                opt.zero_grad()
                net.d(net.g(full)).square().mean().backward()
                opt.step()
        for k, v in net.state_dict().items():
            assert torch.allclose(r0["after"][k], v, rtol=2e-4, atol=2e-5), k
