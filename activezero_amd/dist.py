"""Data-parallel plumbing shared by bench.py and the tests: one process per GPU,
torch.distributed over RCCL (backend "nccl" on ROCm); the same helpers run on CPU
with backend "gloo" for the world-size-2 tests.

The path shards by stereo pair (independent units; train-mode BatchNorm statistics stay
per-rank exactly as in the reference, train.py:536-539 has no SyncBN).  The only data-path
collective is DistributedDataParallel's gradient all-reduce: 5 224 768 fp32 = 20.9 MB, kept
in ONE bucket (bucket_cap_mb=32) so that a single RCCL call moves it over xGMI.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
            int(os.environ.get("WORLD_SIZE", 1)))


def init(backend=None):
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, init_method="env://")
    return rank, local_rank, world


def wrap(model, device=None):
    """DDP wrap with the single-bucket settings; identity for world size 1."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    ids = [device.index] if device is not None and device.type == "cuda" else None
    return torch.nn.parallel.DistributedDataParallel(
        model, device_ids=ids, bucket_cap_mb=32, gradient_as_bucket_view=True)


def fence():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(seconds, device="cpu"):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def rank_seed(base, rank):
    """Each rank draws its own pairs (DistributedSampler-style sharding of synthetic data)."""
    return base + rank


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
