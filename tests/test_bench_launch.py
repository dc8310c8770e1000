"""bench.py's launch contract (no GPU needed): `python bench.py --gpus N` with N > 1 and no launcher environment re-executes
itself as N ranks under torch.distributed.run on 127.0.0.1; under a launcher (WORLD_SIZE / RANK set) it is the worker."""
import importlib
import os
import sys

import pytest

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
    (csrc/conv2d.hip fwd_shape / launch_igemm): the quad-staged 64 x 256 tile that wide layers take on maps >= 128 wide;
    the 128 x 128 tile of the smaller maps is its own class."""
    import bench
    f = bench._quad_gather_tile
    assert f(128, 256, 256, 256, 1) == "64x256"          # D 128 -> 128 @ 256^2
    assert f(256, 128, 128, 128, 1) == "64x256"
    assert f(512, 64, 64, 64, 1) == "128x128"            # under 128 wide: the 128-row tile
    assert f(512, 16, 16, 16, 1) == "128x128"            # 512 @ 16^2: one 16 x 8 tile per image half
    assert f(512, 8, 8, 8, 1) is None                    # 8 x 8 images: two per tile, 80 quads per channel -> dword staging
    assert f(64, 256, 256, 256, 1) is None               # 64 output channels: a narrow layer
    assert f(128, 258, 256, 256, 0) is None              # reflection-padded rows (258 floats): dword staging
    assert f(409, 64, 64, 64, 1) == "64x256"             # 409 channels pad 8 % less on the 64-row tile


def test_pmc_record_of_another_kernel_is_refused(tmp_path, monkeypatch):
    """roofline.traffic comes from the committed counter record of the dominant kernel; a record that names another kernel
    (stale after a kernel change) stops the bench instead of describing the wrong kernel."""
    import json
    bench = _bench()
    rec = bench.load_pmc_dominant()
    assert rec["kernel"].replace(" ", "").startswith(bench.DOMINANT_KERNEL) and rec["read_bytes"] > 0 and rec["write_bytes"] > 0
    # (the one-kernel Winograd convolution reads x once per 64-channel output block: 8 passes over x at 512 channels, through the
    # 256 MB Infinity Cache -- the request-level figure is L2 -> fabric traffic, 4.3x the algorithmic bytes at the very least)
    assert 0.9 < (rec["read_bytes"] + rec["write_bytes"]) / rec["algorithmic_bytes"] < 12.0
    other = tmp_path / "pmc.json"
    other.write_text(json.dumps(dict(rec, kernel="conv_igemm_kernel<3,1,2,2,2,2,8,false,false>")))
    monkeypatch.setattr(bench, "PMC_DOMINANT_FILE", str(other))
    with pytest.raises(SystemExit):
        bench.load_pmc_dominant()


def test_kernel_class_of_a_launch(oracle_lib):
    """The event timer's classes (roofline_by_kernel) follow the routing: a launch winograd.route() takes is bracketed at the
    route (class None here); what it leaves to the direct kernels is classed by the library's dispatch -- the two quad-staged
    gather tiles, the weight gradients, the stride-2 classes by operation."""
    bench = _bench()
    from parity_common import backend
    from swapping_autoencoder_pytorch_amd.stylegan2_op import conv2d_gemm as cg, winograd as wino
    t = bench.DominantKernelTimer()
    t.active = True
    g = cg._Geom(16, 128, 256, 256, 128, 3, 1, 1, False, 1.0)
    small = cg._Geom(16, 512, 64, 64, 512, 3, 1, 1, False, 1.0)       # maps under 128 wide keep the 128 x 128 tile
    narrow = cg._Geom(128, 32, 128, 128, 32, 3, 1, 1, False, 1.0)
    s2 = cg._Geom(16, 128, 257, 257, 256, 3, 2, 0, False, 1.0)
    with backend(oracle_lib):
        with wino.override(enabled=True):
            assert wino.route(g, wino.FWD) == "fused" and wino.route(small, wino.DGRAD) == "fused" and wino.route(g, wino.WGRAD) == "fused"
            assert wino.route(narrow, wino.FWD) is None and wino.route(s2, wino.FWD) is None
            assert wino.route(cg._Geom(4, 64, 32, 32, 64, 3, 1, 1, False, 1.0), wino.FWD) is None      # the small presets stay direct
            for op in (cg.SAE_CONV_FWD, cg.SAE_CONV_DGRAD, cg.SAE_CONV_WGRAD):
                assert t.classify(cg, wino, op, g, False) is None          # bracketed at winograd.conv / wgrad instead
            assert [t.classify(cg, wino, op, s2, False) for op in (cg.SAE_CONV_FWD, cg.SAE_CONV_DGRAD, cg.SAE_CONV_WGRAD)] == [
                "s2_fwd", "s2_dgrad", "s2_wgrad"]
        with wino.override(enabled=False):
            assert t.classify(cg, wino, cg.SAE_CONV_FWD, g, False) == "s1_gather_256"
            assert t.classify(cg, wino, cg.SAE_CONV_DGRAD, g, False) == "s1_gather_256"
            assert t.classify(cg, wino, cg.SAE_CONV_FWD, small, False) == "s1_gather_128"
            assert t.classify(cg, wino, cg.SAE_CONV_FWD, g, True) is None and t.classify(cg, wino, cg.SAE_CONV_WGRAD, g, True) == "s1_wgrad"
            assert t.classify(cg, wino, cg.SAE_CONV_FWD, narrow, False) is None and t.classify(cg, wino, cg.SAE_CONV_WGRAD, narrow, False) is None
            assert t.classify(cg, wino, cg.SAE_CONV_FWD, cg._Geom(16, 128, 64, 64, 256, 1, 1, 0, False, 1.0), False) is None
            t.active = False
            assert t.classify(cg, wino, cg.SAE_CONV_FWD, g, False) is None
    assert bench._quad_gather_tile(409, 128, 128, 128, 1) == "64x256" and bench._quad_gather_tile(409, 64, 64, 64, 1) == "64x256"
    assert bench._quad_gather_tile(128, 130, 128, 128, 1) is None          # rows not a multiple of four floats
