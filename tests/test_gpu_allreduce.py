"""GPU: the gradient all-reduce path on the real backend (RCCL = torch.distributed "nccl"), rehearsed on ONE
rank: SAE_FORCE_ALLREDUCE=1 keeps the bucketing, the post-accumulate hooks, the asynchronous collectives and
finish() active at world_size 1, where the averaged gradient must equal the local one.  (The 2-rank arithmetic
is covered on CPU/gloo in test_grad_allreduce.py; the driver runs the 2/4/8-GPU benchmark.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from swapping_autoencoder_pytorch_amd.grad_allreduce import GradAllReducer, broadcast_parameters
from swapping_autoencoder_pytorch_amd.stylegan2_layers import ConvLayer, ResBlock
torch.manual_seed(0)
net = torch.nn.Sequential(ConvLayer(3, 32, 3), ResBlock(32, 64), ResBlock(64, 96)).cuda()
broadcast_parameters(net)
params = list(net.parameters())
x = torch.randn(4, 3, 64, 64, device="cuda")
net(x).square().mean().backward()
want = [p.grad.clone() for p in params]
for p in params: p.grad = None
red = GradAllReducer(params, bucket_bytes=64 * 1024)      # small buckets: several collectives in flight
assert red.enabled and len(red.buckets) > 2, (red.enabled, len(red.buckets))
for it in range(2):                                        # twice: hooks re-arm, flat buffers are reused
    for p in params: p.grad = None
    red.arm()
    net(x).square().mean().backward()
    # gradients as bucket views: the conv weight gradients were written by their kernels straight into the flat buckets
    in_place = sum(int(p.grad.data_ptr() == red.buckets[red._where[p][0]].flat[red.buckets[red._where[p][0]].offsets[red._where[p][1]]:].data_ptr())
                   for p in params if p.dim() == 4)
    assert in_place == sum(1 for p in params if p.dim() == 4), in_place
    red.finish()
    for p, w in zip(params, want):
        assert torch.equal(p.grad, w), (it, p.shape)
    for b in red.buckets:                                  # slots start on 256-byte boundaries
        assert all(off %% 64 == 0 for off in b.offsets)
torch.cuda.synchronize()
dist.destroy_process_group()
print("ALLREDUCE_OK")
''' % ROOT


def test_forced_single_rank_allreduce_over_rccl():
    env = dict(os.environ, SAE_FORCE_ALLREDUCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ALLREDUCE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_device_prefetcher_delivers_batches_in_order():
    """swapping_autoencoder_pytorch_amd/data_prefetch.py: every batch arrives on the GPU, in order, bit-identical to
    the host batch, through pinned staging on a side stream (the consumer only waits on the copy's event)."""
    import torch
    from swapping_autoencoder_pytorch_amd.data_prefetch import DevicePrefetcher
    host = [{"real_A": torch.rand(4, 3, 64, 64) * 2 - 1, "path_A": ["p%d" % i] * 4, "idx": i} for i in range(7)]
    seen = 0
    for i, batch in enumerate(DevicePrefetcher(iter(host), device="cuda:0", depth=3)):
        assert batch["real_A"].is_cuda and batch["idx"] == i and batch["path_A"] == host[i]["path_A"]
        y = batch["real_A"] * 2.0                                   # consume on the current stream
        assert torch.equal(y.cpu(), host[i]["real_A"] * 2.0)
        seen += 1
    assert seen == len(host)
    with pytest.raises(RuntimeError):
        DevicePrefetcher(iter(host), device="cpu")
