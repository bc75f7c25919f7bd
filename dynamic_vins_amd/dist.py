"""Multi-GPU plumbing (one process per GPU, torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).

SURVEY 8(e): the realistic scaling axis of this path is REPLICAS — independent sequences, one per GPU, no data-path
collective (a single window has ~300 landmarks / ~5 k residual blocks: sharding it over 8 GPUs buys nothing).
bench.py therefore runs one independent sequence per rank and only synchronises for timing.

The one exchange step the path does have when a single window IS sharded by landmark (north_star: "RCCL all-reduce
of the reduced camera-pose Hessian") is provided as `allreduce_reduced_system`: every rank contributes its partial
[S_g | g_g | cost_g]; the sum is formed in RANK ORDER after an all-gather, so all ranks hold bitwise identical
systems regardless of the collective's internal algorithm (ring / tree / direct over the 7 xGMI links)."""
import os

import torch
import torch.distributed as dist


def init(prefer_gpu=True):
    """-> (rank, world, local_rank).  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* as set by torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        use_gpu = prefer_gpu and torch.cuda.is_available()
        dist.init_process_group(backend="nccl" if use_gpu else "gloo", rank=rank, world_size=world)
    return rank, world, local_rank


def finalize():
    if dist.is_initialized():
        dist.destroy_process_group()


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX over ranks of a python float (the bench's step time)"""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def sequence_phase(rank):
    """start-time offset of rank's independent sequence along the trajectory (replica mode)"""
    return 1.7 * rank


def shard_landmarks(nlm, rank, world):
    """round-robin landmark partition: landmark l lives on rank l % world"""
    return list(range(rank, nlm, world))


def allreduce_reduced_system(partial):
    """partial: 1-D fp64 tensor [S (n*n) | g (n) | cost] of this rank -> the rank-ordered sum, identical bits on every rank"""
    if not dist.is_initialized():
        return partial.clone()
    world = dist.get_world_size()
    parts = [torch.empty_like(partial) for _ in range(world)]
    dist.all_gather(parts, partial.contiguous())
    out = parts[0].clone()
    for p in parts[1:]:
        out += p
    return out


def whole_job_rate(units_per_rank, world, seconds_max):
    """BASELINE metric for N ranks: units all ranks processed / max-over-ranks time"""
    return world * units_per_rank / seconds_max
