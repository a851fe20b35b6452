"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm,
"gloo" for the CPU tests).  Clips are independent, so extraction needs no collective; the only
exchange is the all-gather that assembles the retrieval gallery."""
import math
import os

import torch
import torch.distributed as td


def world_size():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def rank():
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def shard_range(n, rank_=None, nshard=None):
    """Contiguous block rule of preprocess/extract_gestsync_feats.py:366-370:
    num_per_shard = ceil(n / nshard); items [rank*n_per : (rank+1)*n_per]."""
    r = rank() if rank_ is None else rank_
    w = world_size() if nshard is None else nshard
    per = math.ceil(n / w)
    return min(n, r * per), min(n, (r + 1) * per)


def init_from_env(backend=None):
    """Initialise from torchrun-style env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 or (td.is_available() and td.is_initialized()):
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    td.init_process_group(backend=backend)


def _backend():
    return td.get_backend() if td.is_available() and td.is_initialized() else None


backend = _backend


def all_gather_rows(x):
    """Gather ragged row blocks (n_r, D) from every rank in rank order.
    Returns (full (sum n_r, D), row offset of this rank's block).
    nccl (= RCCL over xGMI): one all_gather_into_tensor of the padded blocks, device to device.  gloo (the CPU tests, or
    several ranks sharing one GPU): device tensors are staged through host memory."""
    ws = world_size()
    if ws == 1:
        return x, 0
    dev = x.device
    host_staged = _backend() == "gloo" and x.is_cuda
    if host_staged:
        x = x.cpu()
    n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    counts = [torch.zeros_like(n) for _ in range(ws)]
    td.all_gather(counts, n)
    counts = [int(c) for c in counts]
    m = max(counts)
    pad = torch.zeros((m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[:x.shape[0]] = x
    out = torch.empty((ws * m,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if x.is_cuda and hasattr(td, "all_gather_into_tensor"):
        td.all_gather_into_tensor(out, pad)
    else:
        _gather_list(out, pad, ws, m)
    parts = [out[r * m: r * m + counts[r]] for r in range(ws)]
    full = torch.cat(parts, 0)
    return (full.to(dev) if host_staged else full), sum(counts[:rank()])


def _gather_list(out, pad, ws, m):
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    td.all_gather(bufs, pad)
    for r in range(ws):
        out[r * m:(r + 1) * m] = bufs[r]


def all_reduce_sum(t):
    if world_size() > 1:
        if _backend() == "gloo" and t.is_cuda:
            c = t.cpu()
            td.all_reduce(c, op=td.ReduceOp.SUM)
            t.copy_(c)
        else:
            td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def all_reduce_max(t):
    if world_size() > 1:
        if _backend() == "gloo" and t.is_cuda:
            c = t.cpu()
            td.all_reduce(c, op=td.ReduceOp.MAX)
            t.copy_(c)
        else:
            td.all_reduce(t, op=td.ReduceOp.MAX)
    return t


def barrier():
    if world_size() > 1:
        td.barrier()
