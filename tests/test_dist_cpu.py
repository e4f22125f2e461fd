"""N>1 path on CPU: world_size-2 gloo processes shard env ids contiguously and all-gather rollout tensors."""
import os
import socket
import numpy as np
import torch
import torch.multiprocessing as mp
from boxlcd_amd import dist as bdist


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, q):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  r, w, _ = bdist.init_from_env('gloo')
  lo, hi = bdist.shard_range(n_total, r, w)
  # each rank "rolls out" its shard: here the tensor content encodes the global env id
  lcd = torch.arange(lo, hi, dtype=torch.uint8).view(-1, 1, 1).repeat(1, 16, 16)
  obs = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1).repeat(1, 4)
  full_lcd = bdist.all_gather_shards(lcd, n_total)
  full_obs = bdist.all_gather_shards(obs, n_total)
  # per-chunk gather of the FULL chunk tensors [Tc, N, ...] (what bench.py --gpus N moves), equal shards, double-buffered
  Tc, n_eq = 3, 5
  g = bdist.ChunkGatherer(w, [torch.empty((Tc, n_eq, 16, 16), dtype=torch.uint8), torch.empty((Tc, n_eq, 4))])
  chunk_ok = True
  for chunk in range(3):
    cl = torch.full((Tc, n_eq, 16, 16), 10 * chunk + r, dtype=torch.uint8)
    co = torch.full((Tc, n_eq, 4), float(10 * chunk + r))
    g.gather([cl, co])
    gl, go = g.last()
    chunk_ok = chunk_ok and gl.shape == (w, Tc, n_eq, 16, 16) and all((gl[k] == 10 * chunk + k).all() and (go[k] == 10 * chunk + k).all() for k in range(w))
  g.finish()
  # consumer mode: only rank 0 receives (gather to the consuming rank), the others only send
  gc = bdist.ChunkGatherer(w, [torch.empty((Tc, n_eq, 16, 16), dtype=torch.uint8), torch.empty((Tc, n_eq, 4))], mode='consumer', rank=r)
  gc.gather([torch.full((Tc, n_eq, 16, 16), 50 + r, dtype=torch.uint8), torch.full((Tc, n_eq, 4), float(50 + r))])
  gl, go = gc.last()
  if r == 0:
    chunk_ok = chunk_ok and gl.shape == (w, Tc, n_eq, 16, 16) and all((gl[k] == 50 + k).all() and (go[k] == 50 + k).all() for k in range(w))
  else:
    chunk_ok = chunk_ok and gl.shape[0] == 0
  gc.finish()
  t = bdist.max_over_ranks(1.0 + rank)
  bdist.barrier()
  ok = full_lcd.shape == (n_total, 16, 16) and (full_lcd[:, 0, 0].numpy() == np.arange(n_total)).all() and \
      (full_obs[:, 0].numpy() == np.arange(n_total)).all() and t == float(world) and bool(chunk_ok)
  q.put((rank, bool(ok)))
  torch.distributed.destroy_process_group()


def test_shard_ranges_partition():
  for n, w in [(100000, 8), (7, 2), (5, 8), (50001, 4)]:
    spans = [bdist.shard_range(n, r, w) for r in range(w)]
    assert spans[0][0] == 0 and spans[-1][1] == n and all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def test_gloo_two_ranks_gather_uneven_shards():
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, 37, q)) for r in range(2)]
  for p in procs: p.start()
  res = [q.get(timeout=120) for _ in procs]
  for p in procs: p.join(timeout=60)
  assert sorted(res) == [(0, True), (1, True)]
