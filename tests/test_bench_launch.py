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
