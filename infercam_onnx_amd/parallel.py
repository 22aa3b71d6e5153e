"""Multi-GPU layout of the hot path: one process per GPU, one camera stream per rank.

The path shards by independent streams (the reference has no cross-frame state: the model is
immutable, `run(&self)`, infer_server/src/nn.rs:178-186), so there is NO data-path collective.
The only exchange is the start-up broadcast of the packed weight blob (+ priors) from rank 0
(RCCL over xGMI on the GPU box: torch.distributed backend "nccl"; gloo in the CPU tests), and
the max-over-ranks reduction of the benchmark timer.
"""
import numpy as np


def broadcast_weights(weights, dist=None, device=None, src=0):
    """weights: np.float32 array on rank `src` (ignored elsewhere: pass None or any array of the
    right size).  Returns the same bytes on every rank."""
    import torch

    from . import topology as T

    n = T.total_weight_floats()
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return np.ascontiguousarray(weights, np.float32)
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(weights, np.float32).copy())
        assert t.numel() == n
    else:
        t = torch.empty(n, dtype=torch.float32)
    if device is not None:
        t = t.to(device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def stream_for_rank(rank, world_size, num_streams=None):
    """Camera streams owned by `rank`: stream i -> GPU i mod world_size."""
    num_streams = world_size if num_streams is None else num_streams
    return [s for s in range(num_streams) if s % world_size == rank]


def max_over_ranks(seconds, dist=None, device=None):
    """Benchmark timer contract: the job's time is the slowest rank's."""
    import torch

    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_fps(frames_per_rank, world_size, seconds):
    """Whole-job throughput: all ranks' frames / slowest rank's time."""
    return frames_per_rank * world_size / seconds
