"""One process per GPU: bootstrap the library's RCCL communicators through torch.distributed.

torch.distributed is plumbing only (rendezvous, the id broadcast, barriers, the max-over-ranks
of timings).  The data path -- halo planes and dot-product all-reduces -- runs inside
libstorm_hip.so on its own RCCL communicators over xGMI (csrc/comm.hip).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: Optional[str] = None):
    """Initialise torch.distributed from the environment (MASTER_ADDR defaults to 127.0.0.1)."""
    import torch
    import torch.distributed as td

    rank, local_rank, world = env_rank()
    if world == 1 and "MASTER_ADDR" not in os.environ:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not td.is_initialized():
        td.init_process_group(backend=backend, rank=rank, world_size=world)
    return td


def broadcast_bytes(payload: Optional[bytes], n: int, src: int = 0) -> bytes:
    """Broadcast ``n`` bytes from ``src`` over the default process group (any backend)."""
    import torch
    import torch.distributed as td

    dev = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
    if td.get_rank() == src:
        t = torch.tensor(list(payload), dtype=torch.uint8, device=dev)
    else:
        t = torch.zeros(n, dtype=torch.uint8, device=dev)
    td.broadcast(t, src=src)
    return bytes(t.cpu().numpy().tobytes())


def connect(ctx) -> None:
    """Give ``ctx`` its RCCL communicators: rank 0 draws the unique id, everyone joins."""
    import torch.distributed as td

    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        ctx.comm_init(None, 1, 0)
        return
    from .api import Context

    rank, world = td.get_rank(), td.get_world_size()
    uid = Context.comm_unique_id() if rank == 0 else None
    uid = broadcast_bytes(uid, 128, src=0)
    ctx.comm_init(uid, world, rank)


def connect_host_staged(ctx) -> None:
    """Give ``ctx`` the host-staged transport over the default torch.distributed group (any backend that moves
    CPU tensors, i.e. gloo): halo planes and reduction scalars travel through host memory.  Lets several ranks
    share ONE device (tests/test_gpu_two_ranks.py) and serves hosts without a usable RCCL."""
    import torch
    import torch.distributed as td

    rank, world = td.get_rank(), td.get_world_size()

    def allreduce(buf: np.ndarray) -> None:
        t = torch.from_numpy(buf)  # shares memory: reduced in place
        td.all_reduce(t, op=td.ReduceOp.SUM)

    def exchange(nbr_rank, sends, recvs) -> None:
        reqs = []
        for q, peer in enumerate(nbr_rank):
            if recvs[q].size:
                reqs.append(td.irecv(torch.from_numpy(recvs[q]), src=peer, tag=17))
        for q, peer in enumerate(nbr_rank):
            if sends[q].size:
                reqs.append(td.isend(torch.from_numpy(np.ascontiguousarray(sends[q])), dst=peer, tag=17))
        for r in reqs:
            r.wait()

    ctx.comm_init_host(world, rank, allreduce, exchange)


def connect_ipc(ctx, window_bytes: int = 0) -> None:
    """Give ``ctx`` the peer-window transport (``storm_hip_ctx_comm_ipc_export`` / ``_init_ipc``): every rank's window
    handle is all-gathered over the default torch.distributed group (any backend), then mapped.  Halo planes and
    reduction scalars then move by direct stores into the receiver's window -- between GPUs over xGMI, or between
    processes that share one device (tests/test_gpu_two_ranks.py)."""
    import torch
    import torch.distributed as td

    rank, world = td.get_rank(), td.get_world_size()
    mine = ctx.comm_ipc_export(world, rank, window_bytes)
    dev = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
    parts = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(world)]
    td.all_gather(parts, torch.tensor(list(mine), dtype=torch.uint8, device=dev))
    ctx.comm_init_ipc(b"".join(bytes(p.cpu().numpy().tobytes()) for p in parts))
    td.barrier()


def allreduce_max(value: float) -> float:
    import torch
    import torch.distributed as td

    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return value
    dev = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    import torch.distributed as td

    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        td.barrier()


def all_gather_object(obj) -> list:
    """Every rank's ``obj`` (anything picklable), in rank order, on every rank."""
    import torch.distributed as td

    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return [obj]
    out = [None] * td.get_world_size()
    td.all_gather_object(out, obj)
    return out
