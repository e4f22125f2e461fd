"""Multi-GPU sharding of a batched rollout: one process per GPU, contiguous env shards, zero data-path exchange while
stepping; one all-gather of the rollout tensors per chunk (RCCL over xGMI when the tensors live on GPUs, gloo on CPU).

The reference's only parallelism is one OS process per env over mp.Pipe (research/wrappers/async_vector_env.py:98-109);
envs are independent, so sharding is by env id and the only collective is the concatenation of `lcd`/`full_state`
(SURVEY.md §8e).
"""
import os
import numpy as np


def shard_range(n_total, rank, world_size):
  """Contiguous shard [lo, hi) of env ids owned by `rank` (sizes differ by at most one)."""
  base, rem = divmod(int(n_total), int(world_size))
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
  """torch.distributed init from RANK/WORLD_SIZE/MASTER_* (torchrun contract).  Returns (rank, world_size, local_rank)."""
  import torch
  import torch.distributed as dist
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local = int(os.environ.get('LOCAL_RANK', str(rank)))
  if world > 1 and not dist.is_initialized():
    if backend is None:
      backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if backend == 'nccl':
      torch.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
  return rank, world, local


def all_gather_shards(t, n_total, dim=0):
  """Concatenate per-rank shards (possibly uneven) along `dim` into the full [n_total, ...] tensor on every rank."""
  import torch
  import torch.distributed as dist
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return t
  world = dist.get_world_size()
  sizes = [shard_range(n_total, r, world) for r in range(world)]
  maxn = max(hi - lo for lo, hi in sizes)
  t = t.movedim(dim, 0).contiguous()
  if t.shape[0] < maxn:   # pad to equal size: all_gather_into_tensor needs equal shards
    pad = torch.zeros((maxn - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    t = torch.cat([t, pad], 0)
  out = torch.empty((world * maxn,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
  dist.all_gather_into_tensor(out, t)
  parts = [out[r * maxn:r * maxn + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
  return torch.cat(parts, 0).movedim(0, dim)


def max_over_ranks(x):
  import torch
  import torch.distributed as dist
  if not dist.is_initialized() or dist.get_world_size() == 1:
    return float(x)
  dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
  t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


def barrier():
  import torch.distributed as dist
  if dist.is_initialized() and dist.get_world_size() > 1:
    dist.barrier()
