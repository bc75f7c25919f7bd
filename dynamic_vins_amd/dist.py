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
    """the landmark partition of the sharded window (be_api.hip: contiguous ranges of cap = ceil(nlm / world)): the landmarks of `rank`"""
    cap = max(1, -(-nlm // world))
    lo = min(nlm, rank * cap)
    return list(range(lo, min(nlm, lo + cap)))


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


def shard_window(ctx, rank=None, world=None, transport="auto"):
    """Turn ctx (frontend.Context) into rank `rank` of a landmark-sharded window solve (include/dvins.h: dv_dist_init_*).  Every rank must then hand the
    same problem to dv_ba_solve / dv_est_process.
      transport "rccl": ncclAllGather on the BA stream; the ncclUniqueId is created on rank 0 and broadcast through torch.distributed
      transport "host": the exchange vector is staged through pinned host memory and all-gathered by torch.distributed (gloo or nccl) — tests, 1-GPU boxes
      transport "peer": the one-shot exchange — every rank writes its vector into the windows its peers expose (hipIpc over the direct xGMI links); the
                        64-byte window handles are all-gathered once through torch.distributed
      "auto": rccl when the process group's backend is nccl, host otherwise."""
    import ctypes as C
    import numpy as np
    from . import _abi
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if transport == "auto":
        # a lone rank has nothing to exchange: the host transport needs no librccl; rccl only where the process group really is RCCL
        transport = "rccl" if (dist.is_initialized() and world > 1 and dist.get_backend() == "nccl") else "host"
    lib = ctx.lib
    if transport == "peer":
        mine = np.zeros(64, np.uint8)
        if lib.dv_dist_peer_prepare(ctx.h, rank, world, mine.ctypes.data) != 0:
            raise _abi.DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
        handles = np.zeros((world, 64), np.uint8)
        handles[rank] = mine
        if dist.is_initialized() and world > 1:
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            parts = [torch.empty(64, dtype=torch.uint8, device=dev) for _ in range(world)]
            dist.all_gather(parts, torch.from_numpy(mine).to(dev))
            handles = np.stack([t.cpu().numpy() for t in parts])
        handles = np.ascontiguousarray(handles)
        if lib.dv_dist_init_peer(ctx.h, handles.ctypes.data) != 0:
            raise _abi.DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
        return None
    if transport == "rccl":
        uid = np.zeros(128, np.uint8)
        if rank == 0 and lib.dv_dist_unique_id(uid.ctypes.data) != 0:
            raise _abi.DvinsError(ctx.lib.dv_last_error(None).decode())          # be_shard.hip reports this one through the global (ctx-less) error slot
        if dist.is_initialized() and world > 1:
            dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            t = torch.from_numpy(uid).to(dev)
            dist.broadcast(t, 0)
            uid = t.cpu().numpy().copy()
        if lib.dv_dist_init_rccl(ctx.h, rank, world, uid.ctypes.data) != 0:
            raise _abi.DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
        return None

    def allgather(user, send, recv, nbytes):
        try:
            n = nbytes // 8
            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_double)), shape=(n,))
            dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_double)), shape=(world, n))
            if world == 1 or not dist.is_initialized():
                dst[0] = src
                return 0
            if dist.get_backend() == "nccl":
                parts = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(world)]
                dist.all_gather(parts, torch.from_numpy(src.copy()).cuda())
                for r in range(world):
                    dst[r] = parts[r].cpu().numpy()
            else:
                parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(parts, torch.from_numpy(src.copy()))
                for r in range(world):
                    dst[r] = parts[r].numpy()
            return 0
        except Exception:            # never unwind through the C frame
            import traceback
            traceback.print_exc()
            return 1
    cb = _abi.ALLGATHER_FN(allgather)
    if lib.dv_dist_init_host(ctx.h, rank, world, C.cast(cb, C.c_void_p), None) != 0:
        raise _abi.DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
    ctx._dist_cb = cb          # keep the trampoline alive as long as the ctx
    return cb


def dist_info(ctx):
    import ctypes as C
    r, w, t, e = C.c_int(0), C.c_int(0), C.c_int(0), C.c_longlong(0)
    if ctx.lib.dv_dist_info(ctx.h, C.byref(r), C.byref(w), C.byref(t), C.byref(e)) != 0:
        from . import _abi
        raise _abi.DvinsError(ctx.lib.dv_last_error(ctx.h).decode())
    n = C.c_int(0)
    ctx.lib.dv_dist_rccl_ranks(ctx.h, C.byref(n))
    return dict(rank=r.value, world=w.value, transport={0: "none", 1: "rccl", 2: "host", 3: "peer"}[t.value], exchanges=e.value, rccl_ranks=n.value)


def exchange_bytes(ctx, n_landmarks):
    """bytes one rank contributes to the exchanges of a sharded window solve: (system: per linearisation, independent of the landmark count; cost: per cost-only slot; depth: once per solve)"""
    import ctypes as C
    a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
    ctx.lib.dv_dist_exchange_bytes(ctx.h, int(n_landmarks), C.byref(a), C.byref(b), C.byref(c))
    return dict(system=a.value, cost=b.value, depth=c.value)


def whole_job_rate(units_per_rank, world, seconds_max):
    """BASELINE metric for N ranks: units all ranks processed / max-over-ranks time"""
    return world * units_per_rank / seconds_max
