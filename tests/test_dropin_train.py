"""The drop-in runner (swapping_autoencoder_pytorch_amd/dropin.py): device pinning for one-rank-per-GPU launches,
the reference's unmodified train.py end to end, and its optimizer under the 2-rank gradient all-reduce.
CPU only; the two tests that need the reference checkout are skipped where it is absent (the GPU box)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dropin_ref_worker.py")
needs_reference = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")


def test_pin_rank_device_maps_local_rank_to_cuda0():
    from swapping_autoencoder_pytorch_amd.dropin import pin_rank_device
    assert pin_rank_device({"WORLD_SIZE": "1"}, reexec=False) is None
    env = {"WORLD_SIZE": "8", "LOCAL_RANK": "5"}
    assert pin_rank_device(env, reexec=False) == "5" and env["HIP_VISIBLE_DEVICES"] == "5"
    assert pin_rank_device(env, reexec=False) == "5"                 # idempotent (a re-exec'd process keeps its pin)
    env = {"WORLD_SIZE": "4", "LOCAL_RANK": "2", "HIP_VISIBLE_DEVICES": "4,5,6,7"}
    assert pin_rank_device(env, reexec=False) == "6" and env["HIP_VISIBLE_DEVICES"] == "6"


def test_pin_rank_device_honours_cuda_visible_devices():
    from swapping_autoencoder_pytorch_amd.dropin import pin_rank_device
    env = {"WORLD_SIZE": "4", "LOCAL_RANK": "1", "CUDA_VISIBLE_DEVICES": "4,5,6,7"}
    assert pin_rank_device(env, reexec=False) == "5"
    assert env["HIP_VISIBLE_DEVICES"] == "5" and "CUDA_VISIBLE_DEVICES" not in env
    env = {"WORLD_SIZE": "4", "LOCAL_RANK": "3", "HIP_VISIBLE_DEVICES": "2,3", "CUDA_VISIBLE_DEVICES": "0,1,2,3"}
    with pytest.raises(RuntimeError):          # more ranks than visible devices: refuse instead of landing on GPU 3
        pin_rank_device(env, reexec=False)


def test_data_path_stubs_fail_with_the_package_name():
    from swapping_autoencoder_pytorch_amd.dropin import _MissingDataPackage
    m = _MissingDataPackage("lmdb")
    with pytest.raises(ImportError, match="lmdb"):
        m.open("x")


def test_prefetched_loader_delegates_and_restarts(monkeypatch):
    """PrefetchedLoader around a reference-style loader (data/__init__.py:81-129), with the staging object replaced by a
    recorder (no GPU here): attribute / len delegation, lazy start, restart on set_phase and iter(), StopIteration."""
    from swapping_autoencoder_pytorch_amd import data_prefetch, dropin

    class Loader:
        def __init__(self):
            self.phase, self.items, self.length, self.underlying_dataset = "train", iter(range(5)), 5, "ds"

        def set_phase(self, phase):
            if phase != self.phase:
                self.phase, self.items = phase, iter(range(100, 103))

        def __iter__(self):
            self.items = iter(range(5))
            return self

        def __len__(self):
            return self.length

        def __next__(self):
            return next(self.items)

    made = []

    class Recorder:
        def __init__(self, iterable, device, depth):
            made.append((device, depth))
            self.it = iter(iterable)

        def __next__(self):
            return next(self.it)

    monkeypatch.setattr(data_prefetch, "DevicePrefetcher", Recorder)
    pl = dropin.PrefetchedLoader(Loader(), device="cuda:0", depth=2)
    assert len(pl) == 5 and pl.underlying_dataset == "ds" and pl.phase == "train" and made == []
    assert [next(pl), next(pl)] == [0, 1] and made == [("cuda:0", 2)]
    pl.set_phase("train")
    assert next(pl) == 2 and len(made) == 1                   # same phase: staging continues
    pl.set_phase("test")
    assert [next(pl) for _ in range(3)] == [100, 101, 102] and len(made) == 2
    with pytest.raises(StopIteration):
        next(pl)
    assert [x for x in pl] == [0, 1, 2, 3, 4] and len(made) == 3
    pl.length = 7                                              # attribute writes reach the wrapped loader
    assert len(pl) == 7


def test_device_is_pinned_before_torch_is_imported():
    """The runner's entry point narrows HIP_VISIBLE_DEVICES before `import torch` happens at all (the HIP runtime
    reads the variable once, at initialisation)."""
    code = ("import os, sys\n"
            "os.environ.update(WORLD_SIZE='2', LOCAL_RANK='1', RANK='1')\n"
            "import swapping_autoencoder_pytorch_amd.dropin as d\n"
            "assert 'torch' not in sys.modules, 'importing the runner must not import torch'\n"
            "d.pin_rank_device()\n"
            "assert os.environ['HIP_VISIBLE_DEVICES'] == '1'\n"
            "assert 'torch' not in sys.modules\n"
            "try:\n"
            "    d.main([])\n"                                           # main() pins first, then checks its arguments
            "except SystemExit:\n"
            "    pass\n"
            "assert 'torch' not in sys.modules\n"
            "print('pinned-before-torch')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "pinned-before-torch" in out.stdout


@needs_reference
def test_reference_train_py_runs_end_to_end(tmp_path):
    """`python train.py --dataset_mode synthetic ...` of the reference, byte-identical, on our operators: option
    parser, data loader, D / G / R1 steps, loss log, checkpoint."""
    out = subprocess.run([sys.executable, WORKER, "train", str(tmp_path)], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "Training finished." in out.stdout
    run = tmp_path / "dropin_e2e"
    assert (run / "loss_log.txt").exists() and (run / "iter.txt").exists()
    assert (run / "latest_checkpoint.pth").exists()
    log = (run / "loss_log.txt").read_text()
    for key in ("D_R1", "D_real", "G_L1", "G_GAN_mix", "PatchD_real"):
        assert key in log, key
    state = torch.load(run / "latest_checkpoint.pth", map_location="cpu")
    assert any(k.startswith("G.") for k in state) and any(k.startswith("Dpatch.") for k in state)


@needs_reference
@pytest.mark.parametrize("adam", ["torch", "fused"])
def test_reference_optimizer_under_two_rank_allreduce(tmp_path, adam):
    """attach_gradient_allreduce on the reference's own optimizer, world_size 2 over gloo: replicas that start
    different and see different data end bit-identical (broadcast + averaged gradients), and one checkpoint.
    "torch": the reference's torch.optim.Adam behind step hooks (finish -> step -> arm); "fused": FusedAdam substituted as
    dropin.main does, its step() replaced by GradAllReducer.finish_into (per-bucket updates on the summed gradients)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, WORKER, "ddp", str(tmp_path), str(port), adam], cwd=ROOT, capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    moved = 0
    for k in r0["end"]:
        assert torch.equal(r0["end"][k], r1["end"][k]), k             # same weights on both ranks after 4 steps
        assert torch.equal(r0["start"][k], r1["start"][k]), k         # rank 0's initial weights were broadcast
        moved += int(not torch.equal(r0["start"][k], r0["end"][k]))
    assert moved > 50                                                  # ... and they were actually trained
    ckpts = [f for f in os.listdir(tmp_path / "dropin_ddp") if f.endswith("checkpoint.pth")]
    assert sorted(ckpts) == ["0k_checkpoint.pth", "latest_checkpoint.pth"]


@needs_reference
def test_aten_cpu_path_matches_reference_discriminator():
    """Pin of oracle/aten_cpu_path.py (bench.py's cpu_baseline) to the reference's own CPU path."""
    out = subprocess.run([sys.executable, WORKER, "aten_pin"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "aten-cpu-path-pinned" in out.stdout
