"""Checkpoint with optimiser state: a run that saves after two steps and resumes in a fresh
process-state continues bit-identically to an uninterrupted run (CPU, oracle back end, micro preset)."""
import torch

import parity_common as P
from param_recipe import uniform_images


def _run(opt_over, steps, oracle_lib, resume=False, save_at=None):
    from swapping_autoencoder_pytorch_amd.swapping_autoencoder_optimizer import create_optimizer
    with P.backend(oracle_lib), P.cpu_random_stream("cpu"):
        opt, model, net = P.build_micro("cpu", **opt_over)
        if resume:
            opt.continue_train = True
            net.load()        # weights + the discriminator iteration buffer (after build_micro's recipe fill)
        optimizer = create_optimizer(opt, model)
        out = []
        for it in steps:
            torch.manual_seed(1000 + it)
            out.append(optimizer.train_one_step({"real_A": uniform_images(4, 32, 600 + it)}, it))
            if save_at is not None and it == save_at:
                optimizer.save(it)
        return out, {k: v.clone() for k, v in net.state_dict().items()}, optimizer


def test_resume_continues_identically(oracle_lib, tmp_path):
    over = dict(checkpoints_dir=str(tmp_path), name="resume")
    full_losses, full_state, _ = _run(over, range(4), oracle_lib)
    _run(over, range(2), oracle_lib, save_at=1)
    assert (tmp_path / "resume" / "latest_optimizer.pth").exists() and (tmp_path / "resume" / "latest_checkpoint.pth").exists()
    tail_losses, tail_state, optimizer = _run(over, range(2, 4), oracle_lib, resume=True)
    assert optimizer.discriminator_iter_counter == 2 and optimizer.train_mode_counter == 0
    for a, b in zip(full_losses[2:], tail_losses):
        assert set(a) == set(b)
        for k in a:
            assert float(a[k]) == float(b[k]), (k, float(a[k]), float(b[k]))
    for k in full_state:
        assert torch.equal(full_state[k], tail_state[k]), k
