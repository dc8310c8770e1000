"""bench.py's launch contract (no GPU needed): `python bench.py --gpus N` with N > 1 and no launcher environment re-executes
itself as N ranks under torch.distributed.run on 127.0.0.1; under a launcher (WORLD_SIZE / RANK set) it is the worker."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def _args(bench, argv):
    old = sys.argv
    sys.argv = ["bench.py"] + argv
    try:
        return bench.parse()
    finally:
        sys.argv = old


def test_single_gpu_runs_in_process(monkeypatch):
    bench = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert bench.self_launch_command(_args(bench, ["--gpus", "1"]), ["--gpus", "1"]) is None


def test_multi_gpu_relaunches_itself_as_n_ranks(monkeypatch):
    bench = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd = bench.self_launch_command(_args(bench, argv), argv)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv


def test_under_a_launcher_it_is_the_worker(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "3")
    assert bench.self_launch_command(_args(bench, ["--gpus", "8"]), ["--gpus", "8"]) is None


def test_dominant_kernel_filter_follows_the_dispatch_rule():
    """bench.py counts a launch towards `roofline` only when the library runs it on the named instantiation
    (csrc/conv2d.hip fwd_shape / launch_igemm): 128 x 128 tile, quad staging."""
    import bench
    f = bench._runs_quad_main_kernel
    assert f(128, 256, 256, 256, 1)            # D 128 -> 128 @ 256^2
    assert f(512, 16, 16, 16, 1)               # 512 @ 16^2: one 16 x 8 tile per image half
    assert not f(512, 8, 8, 8, 1)              # 8 x 8 images: two per tile, 80 quads per channel -> dword staging
    assert not f(64, 256, 256, 256, 1)         # 64 output channels: the 64 x 256 tile
    assert not f(128, 258, 256, 256, 0)        # reflection-padded rows (258 floats): dword staging
    assert not f(409, 64, 64, 64, 1)           # 409 channels pad 8 % less on the 64-row tile
