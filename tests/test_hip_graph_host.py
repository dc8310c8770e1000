"""Host logic of the hipGraph replay (swapping_autoencoder_pytorch_amd/hip_graph.py) that needs no GPU: when graph mode is
chosen, the warm-up / capture / replay bookkeeping of StepGraphs (with the capture itself stubbed out), and the re-wrapping of
the static outputs.  The capture and the bit-identity with the eager step are tests/test_gpu_graph_step.py (-m gpu).
Replaces the eager loop of the reference's train.py:22-28."""
import torch


def test_graph_mode_is_not_chosen_without_a_gpu_or_the_fused_adam(monkeypatch):
    from swapping_autoencoder_pytorch_amd import hip_graph
    p = [torch.nn.Parameter(torch.zeros(3))]
    assert not hip_graph.wanted(p, [torch.optim.Adam(p)])          # CPU parameters (and ATen's Adam)
    assert not hip_graph.wanted([], [])
    monkeypatch.setenv("SAE_HIP_GRAPH", "0")
    assert not hip_graph.wanted(p, [])


def test_step_graphs_warm_up_then_capture_once_then_replay(monkeypatch):
    from swapping_autoencoder_pytorch_amd import hip_graph
    made = []

    class FakeCaptured:
        def __init__(self, body, images):
            made.append(tuple(images.shape))
            self.out = body(images)                # "capture": the body runs once more on the static input
            self.replays = 0

        def replay(self, images):
            self.replays += 1
            return hip_graph._fresh_containers(self.out)

    monkeypatch.setattr(hip_graph, "_Captured", FakeCaptured)
    calls = []

    def body(images):
        calls.append(tuple(images.shape))
        return {"loss": images.sum()}, {"metric": images.mean()}

    g = hip_graph.StepGraphs()
    x = torch.ones(4, 3, 8, 8)
    for _ in range(hip_graph.WARMUP_CALLS):
        g.run("discriminator", x, body)
    assert len(calls) == hip_graph.WARMUP_CALLS and not made                   # eager calls first
    a = g.run("discriminator", x, body)                                         # the capture call: body once, then one replay
    b = g.run("discriminator", x, body)
    assert len(made) == 1 and len(calls) == hip_graph.WARMUP_CALLS + 1
    assert g.graphs[("discriminator", (4, 3, 8, 8), x.device)].replays == 2
    # static tensors, fresh containers: the caller adds keys to the loss dictionary it gets back
    a[0]["D_R1"] = torch.zeros(())
    assert "D_R1" not in b[0] and a[0]["loss"] is b[0]["loss"]
    # another call kind, another batch shape: their own warm-up and capture
    g.run("generator", x, body)
    g.run("discriminator", torch.ones(2, 3, 8, 8), body)
    assert len(made) == 1 and g.captured() == ["discriminator"]


def test_bench_rehearsal_of_several_ranks_on_one_gpu_lowers_the_hardware_queues(monkeypatch):
    """bench.py --gpus N --same-device: N processes x 8 hardware queues on one GPU oversubscribe the device's queues (a two-rank
    run did not finish in round 6): the relaunch sets 4 per process."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(os, "execv", lambda exe, argv: seen.update(exe=exe, argv=argv, queues=os.environ.get("GPU_MAX_HW_QUEUES")))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--same-device"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    try:
        bench.main()
    except BaseException:      # after the (stubbed) execv the single-process path goes on and stops at "needs a GPU"
        pass
    assert seen.get("queues") == "4" and "--same-device" in seen["argv"]
