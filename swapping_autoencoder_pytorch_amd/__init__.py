"""MI355X-native hot path of the Swapping-Autoencoder GAN training step."""
__all__ = ["hip_lib"]


def _reserve_hardware_queues():
    """The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share
    one run one after the other.  The step runs its independent branches on two streams (streams.py); once RCCL has created its
    own streams the side stream landed on the main stream's queue and the overlap was gone (one rank, communicator initialised:
    281.6 -> 290.6 ms per step; with 8 queues 281.2 -- profiles/r4_ab_hw_queues.txt).  The variable is read when the runtime
    initialises, so it is set here, at import, unless the user chose a value; a process whose HIP runtime is already up keeps
    what it has (export GPU_MAX_HW_QUEUES=8 in that case)."""
    import os
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


_reserve_hardware_queues()
