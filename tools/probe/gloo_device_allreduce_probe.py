"""Two ranks on ONE GPU over gloo: does an asynchronous all-reduce of DEVICE tensors complete, by message size?
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 \
        tools/probe/gloo_device_allreduce_probe.py
Prints one line per size from rank 0; a size that stalls is reported after 20 s by the watchdog (the process then exits).
Background: bench.py --gpus 2 --same-device (the one-GPU rehearsal of the multi-rank path) stalled in work.wait() on the 32 MB
gradient buckets in round 6, while the 3 MB buckets of tests/test_ddp_fullmodel.py pass; the real multi-GPU path is RCCL."""
import os
import sys
import threading
import time

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    state = {"size": None, "t0": time.time()}

    def watchdog():
        while True:
            time.sleep(1.0)
            if state["size"] is not None and time.time() - state["t0"] > 20.0:
                print("rank %d: STALLED at %d MB x 4 in flight (20 s)" % (rank, state["size"]), flush=True)
                os._exit(3)

    threading.Thread(target=watchdog, daemon=True).start()
    for mb in (1, 4, 8, 16, 32, 64):
        n = mb * 1024 * 1024 // 4
        ts = [torch.full((n,), float(rank + 1 + i), device="cuda:0") for i in range(4)]
        torch.cuda.synchronize()
        state["size"], state["t0"] = mb, time.time()
        works = [dist.all_reduce(t, async_op=True) for t in ts]
        for w in works:
            w.wait()
        torch.cuda.synchronize()
        ok = all(float(t[0]) == sum(r + 1 + i for r in range(world)) for i, t in enumerate(ts))
        if rank == 0:
            print("%3d MB x 4 asynchronous device all-reduces: %.2f s, values %s" % (mb, time.time() - state["t0"], "ok" if ok else "WRONG"),
                  flush=True)
        state["size"] = None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
