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
        red.finish()
        results[tag] = [None if p.grad is None else p.grad.clone() for p in params]
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


def test_single_process_is_a_noop():
    from swapping_autoencoder_pytorch_amd.grad_allreduce import GradAllReducer
    net = Net()
    red = GradAllReducer(list(net.parameters()))
    assert not red.enabled
    red.arm()
    _loss(net, torch.randn(4, 6)).mean().backward()
    red.finish()
    assert net.a[0].weight.grad is not None
