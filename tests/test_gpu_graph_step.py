"""GPU: the train step replayed as hipGraphs (swapping_autoencoder_pytorch_amd/hip_graph.py) computes what the eager step
computes -- the loss dictionaries of every call (incl. lazy-R1 iterations, which stay eager between replays), every parameter
and both Adam states after the updates, bit for bit: every kernel is deterministic, the graph holds the same launches in the
same order, the random draws advance the same Philox offsets, and Adam's update counts live in device memory in both runs.
Replaces the eager loop of the reference's train.py:22-28 / optimizers/swapping_autoencoder_optimizer.py:59-111."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(graph, iters, preset="tiny32", batch=4, r1_every=3, two_streams=True):
    from swapping_autoencoder_pytorch_amd import hip_graph
    from swapping_autoencoder_pytorch_amd.options import make_options
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_model import create_model
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer
    os.environ["SAE_TWO_STREAMS"] = "1" if two_streams else "0"
    os.environ.pop("SAE_HIP_GRAPH", None)
    try:
        opt = make_options(preset, batch_size=batch, num_gpus=1)
        opt.R1_once_every = r1_every
        torch.manual_seed(0)
        model = create_model(opt)
        optimizer = create_optimizer(opt, model)
        assert optimizer.graphs is not None            # single rank, GPU, FusedAdam: graph mode is the default
        if not graph:
            optimizer.graphs = None                    # eager calls; Adam keeps its device-resident update counts
        torch.manual_seed(1)
        g = torch.Generator(device="cuda").manual_seed(2)
        out = []
        for i in range(iters):
            for _ in range(2):
                x = torch.rand(batch, 3, opt.crop_size, opt.crop_size, device="cuda", generator=g) * 2 - 1
                losses = optimizer.train_one_step({"real_A": x}, i)
                out.append({k: float(v) for k, v in losses.items()})
        torch.cuda.synchronize()
        if graph:
            assert optimizer.graphs.captured() == ["discriminator", "generator"]
            assert iters > hip_graph.WARMUP_CALLS + 1      # ... and at least one pure replay after the capture call
        state = {k: v.detach().clone() for k, v in model.singlegpu_model.state_dict().items()}
        adam = optimizer.state_dict()
        return out, state, adam
    finally:
        os.environ.pop("SAE_TWO_STREAMS", None)


def _same(a, b):
    (la, sa, aa), (lb, sb, ab) = a, b
    assert len(la) == len(lb)
    for i, (x, y) in enumerate(zip(la, lb)):
        assert x == y, (i, x, y)
    assert all(v == v for call in la for v in call.values())          # no NaN
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    for name in ("optimizer_G", "optimizer_D"):
        for (ia, ea), (ib, eb) in zip(sorted(aa[name]["state"].items()), sorted(ab[name]["state"].items())):
            assert ia == ib and float(ea["step"]) == float(eb["step"]), (name, ia)
            assert torch.equal(ea["exp_avg"], eb["exp_avg"]) and torch.equal(ea["exp_avg_sq"], eb["exp_avg_sq"]), (name, ia)
    assert int(aa["discriminator_iter_counter"]) == int(ab["discriminator_iter_counter"])


@pytest.mark.parametrize("two_streams", [True, False], ids=["two_streams", "one_stream"])
def test_graph_replay_equals_the_eager_step(two_streams):
    iters = 7                                          # calls 1-2 eager, 3 captured + replayed, 4-7 replayed; R1 at 3 and 6
    eager = _run(False, iters, two_streams=two_streams)
    graph = _run(True, iters, two_streams=two_streams)
    assert any("D_R1" in call for call in eager[0])
    # a loss dictionary of a non-R1 iteration must not carry the previous R1 iteration's D_R1 (static outputs are re-wrapped)
    assert [sorted(c) for c in eager[0]] == [sorted(c) for c in graph[0]]
    _same(eager, graph)


def test_graph_replay_at_a_mid_size_preset():
    """the same at 64 x 64 crops of the church networks' widths (Winograd route and stride-2 kernels in the graph)"""
    from swapping_autoencoder_pytorch_amd.options import PRESETS
    if "church256" not in PRESETS:
        pytest.skip("no church256 preset")
    eager = _run(False, 4, preset="church256", batch=2, r1_every=16)
    graph = _run(True, 4, preset="church256", batch=2, r1_every=16)
    _same(eager, graph)


def test_graphs_are_off_for_a_rank_of_a_multi_rank_job_and_by_the_switch(monkeypatch):
    from swapping_autoencoder_pytorch_amd import hip_graph
    from swapping_autoencoder_pytorch_amd.fused_adam import FusedAdam
    p = [torch.nn.Parameter(torch.zeros(4, device="cuda"))]
    o = [FusedAdam(p)]
    assert hip_graph.wanted(p, o)
    monkeypatch.setenv("SAE_HIP_GRAPH", "0")
    assert not hip_graph.wanted(p, o)
    monkeypatch.delenv("SAE_HIP_GRAPH")
    assert not hip_graph.wanted(p, [torch.optim.Adam(p)])
    import torch.distributed as dist
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda *a: 2)
    assert not hip_graph.wanted(p, o)
