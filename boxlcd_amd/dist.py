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


class ChunkGatherer:
  """All-gather of one rollout chunk's result tensors (`lcd[Tc, N, H, W]`, `full_state[Tc, N, obs]`) into
  preallocated `[world, Tc, N, ...]` buffers, on a side stream so that the next chunk's stepping overlaps the transfer.

  Equal shards (the bench's weak-scaling layout) go through `all_gather_into_tensor` straight into the destination - no
  pad, no concatenate.  uint8 0/1 tensors (LCD frames: the reference's bool arrays) cross the links at ONE BIT per pixel
  (`blcd_pack_bits` before, `blcd_unpack_bits` after the collective, both on the side stream) and are delivered as uint8
  again - 8x less xGMI traffic for the tensor that is 94 % of a chunk.  Destination buffers are double-buffered; collectives
  are ordered by the side stream, so a buffer is only rewritten after the gather that last filled it has finished.
  `producer` is the stream the chunk was written on (the handle's stream); the returned event marks the end of the gather:
  make the producer wait on it before it rewrites the source region (`producer.wait_event(ev)`)."""

  def __init__(self, world, templates, nbuf=2, binary=None, mode='all', rank=0):
    """`binary[i]` = True declares tensor i to hold only 0/1 bytes (LCD frames in mode '1'): it is bit-packed for the wire.
    Packing keeps bit 0 of every byte, so it is strictly opt-in - RGB frames or LCDs scaled to 255 must not be declared."""
    import torch
    self.world = world
    # mode 'all': every rank receives every rank's chunk (all-gather, north_star's wording); 'consumer': only rank 0 does
    # (gather to the consuming rank - what a single trainer process needs; the other ranks only send)
    assert mode in ('all', 'consumer')
    self.mode, self.rank = mode, rank
    self.cuda = templates[0].is_cuda
    binary = [False] * len(templates) if binary is None else list(binary)
    assert len(binary) == len(templates)
    for t, b in zip(templates, binary):
      assert not b or (t.dtype == torch.uint8 and t.numel() % 8 == 0), 'binary tensors must be uint8 with a multiple of 8 elements'
    recv = mode == 'all' or rank == 0
    self.bufs = [[torch.empty((world if recv else 0,) + tuple(t.shape), dtype=t.dtype, device=t.device) for t in templates] for _ in range(nbuf)]
    self.k = 0
    self.stream = torch.cuda.Stream(device=templates[0].device) if self.cuda else None
    # packed staging for the uint8 tensors (GPU only: the kernels live in the HIP library)
    self.packed = [self.cuda and b for b in binary]
    self.pk_src = [torch.empty(t.numel() // 8, dtype=torch.uint8, device=t.device) if p else None for t, p in zip(templates, self.packed)]
    self.pk_dst = [torch.empty((world if recv else 0) * t.numel() // 8, dtype=torch.uint8, device=t.device) if p else None for t, p in zip(templates, self.packed)]

  def gather(self, srcs, producer=None):
    import torch
    import torch.distributed as dist
    dst = self.bufs[self.k % len(self.bufs)]
    self.k += 1
    consumer = self.mode == 'consumer'
    if not self.cuda:                       # gloo rehearsal on CPU: synchronous
      for d, s in zip(dst, srcs):
        if consumer:
          dist.gather(s.contiguous(), list(d.unbind(0)) if self.rank == 0 else None, dst=0)
        else:
          dist.all_gather(list(d.unbind(0)), s.contiguous())
      return None
    from . import _lib
    ready = torch.cuda.Event()
    ready.record(producer if producer is not None else torch.cuda.current_stream())
    with torch.cuda.stream(self.stream):
      self.stream.wait_event(ready)
      sp = self.stream.cuda_stream
      for i, (d, s) in enumerate(zip(dst, srcs)):
        assert s.is_contiguous(), 'gather sources must be contiguous'
        if self.packed[i]:
          assert s.data_ptr() % 8 == 0, 'bit-packed sources must be 8-byte aligned'
          _lib.pack_bits(s, self.pk_src[i], sp)
          if consumer:
            dist.gather(self.pk_src[i], list(self.pk_dst[i].view(self.world, -1).unbind(0)) if self.rank == 0 else None, dst=0)
          else:
            dist.all_gather_into_tensor(self.pk_dst[i], self.pk_src[i])
          if not consumer or self.rank == 0:
            _lib.unpack_bits(self.pk_dst[i], d, sp)
        elif consumer:
          dist.gather(s, list(d.unbind(0)) if self.rank == 0 else None, dst=0)
        else:
          dist.all_gather_into_tensor(d.flatten(0, 1), s)   # [world*Tc, N, ...] = concatenation along dim 0 (NCCL and gloo)
      done = torch.cuda.Event()
      done.record(self.stream)
    return done

  def last(self):
    return self.bufs[(self.k - 1) % len(self.bufs)]

  def finish(self):
    if self.cuda:
      self.stream.synchronize()


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
