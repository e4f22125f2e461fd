"""Drop-in boundary on the GPU: the gym-style single env and the batched (vector-env-shaped) env."""
import numpy as np
import pytest
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
from oracle import pyb2o

pytestmark = pytest.mark.gpu


def test_single_env_api_like_reference():
  env = B.envs.Urchin()
  env.seed(0)
  obs = env.reset()
  assert set(obs) == {'full_state', 'proprio', 'lcd'}
  assert obs['full_state'].dtype == np.float64 and obs['full_state'].shape == (16,) and obs['lcd'].dtype == bool
  assert obs['lcd'].shape == (16, 32) and obs['proprio'].shape == (16,)
  assert np.abs(obs['full_state']).max() <= 1.0 + 1e-9
  for t in range(env.G.ep_len):
    obs, rew, done, info = env.step(env.action_space.sample())
  assert rew == 0.0 and done is True and info == {'timeout': True}
  assert (env.render() == obs['lcd']).all() and (env.lcd_render() == obs['lcd']).all()
  env.close()


def test_reset_with_state_renders_that_state():
  env = B.envs.LuxoBall()
  env.seed(1)
  o1 = env.reset()
  for _ in range(5):
    o1 = env.step(np.zeros(3, np.float32))[0]
  o2 = env.reset(full_state=o1['full_state'])
  assert np.abs(o2['full_state'] - o1['full_state']).max() < 1e-5
  assert (o2['lcd'] != o1['lcd']).sum() <= 2        # poses survive a float32 round trip through the normalised state
  o3 = env.reset(proprio=o1['proprio'])
  assert np.abs(o3['proprio'] - o1['proprio']).max() < 1e-5
  env.close()


def test_batched_env_shapes_and_torch_zero_copy():
  import torch
  n = 512
  venv = B.BatchedWorldEnv('LuxoBall', n, seed=0)
  obs = venv.reset()
  assert obs['full_state'].shape == (n, 20) and obs['full_state'].dtype == np.float32
  assert obs['lcd'].shape == (n, 16, 24) and obs['lcd'].dtype == bool and obs['proprio'].shape == (n, 16)
  obs, rew, done, infos = venv.step(venv.sample_actions())
  assert rew.shape == (n,) and done.shape == (n,) and len(infos) == n and not done.any()
  # device-resident rollout: actions and outputs are torch tensors on the GPU, no host copies
  T = 8
  acts = torch.as_tensor(venv.sample_actions(T)).cuda()
  lcd = torch.empty((T, n, 16, 24), dtype=torch.uint8, device='cuda')
  fs = torch.empty((T, n, 20), dtype=torch.float32, device='cuda')
  h = venv._handle()
  h.rollout(acts, T, lcd, fs)
  torch.cuda.synchronize()
  host_fs, host_lcd = h.get_obs(np.float32)
  assert (lcd[-1].cpu().numpy() == host_lcd).all() and (fs[-1].cpu().numpy() == host_fs).all()
  # state -> LCD renderer used by the world models
  frames = venv.render_states(host_fs[:32].astype(np.float64))
  assert frames.shape == (32, 16, 24) and (frames != host_lcd[:32].astype(bool)).reshape(32, -1).sum(1).max() <= 3
  venv.close()


def test_barrel_writer_matches_reference_layout(tmp_path):
  """research/data.py:36-79 layout: entry j = observation before action j; loadable the way RolloutDataset does it."""
  from boxlcd_amd.data import fill_barrels
  paths = fill_barrels('Urchin', 1, tmp_path, 'train', {'ep_len': 12}, seed=3, barrel_size=200, stamp='20260101T000000')
  assert paths[0].name == '20260101T000000-12.barrel.npz' and list((tmp_path / 'train').glob('*.barrel.npz')) == paths
  b = np.load(paths[0], allow_pickle=True)
  assert set(b.keys()) == {'action', 'full_state', 'proprio', 'lcd'}
  assert b['action'].shape == (200, 12, 3) and b['action'].dtype == np.float64
  assert b['full_state'].shape == (200, 12, 16) and b['full_state'].dtype == np.float32
  assert b['proprio'].shape == (200, 12, 16) and b['lcd'].shape == (200, 12, 16, 32) and b['lcd'].dtype == bool
  # replay episode 7 with the single-env API semantics: obs_{j+1} = step(action_j)
  venv = B.BatchedWorldEnv('Urchin', 200, {'ep_len': 12}, seed=3)
  obs = venv.reset()
  assert (obs['lcd'][7] == b['lcd'][7, 0]).all() and np.allclose(obs['full_state'][7], b['full_state'][7, 0])
  for j in range(11):
    obs, *_ = venv.step(b['action'][:, j].astype(np.float32))
    assert (obs['lcd'][7] == b['lcd'][7, j + 1]).all() and np.allclose(obs['full_state'][7], b['full_state'][7, j + 1], atol=1e-6)
  venv.close()
  import torch
  elems = {k: torch.as_tensor(b[k], dtype=torch.float32) for k in b.keys()}     # research/data.py:149
  assert elems['lcd'].max() <= 1.0 and elems['lcd'].min() >= 0.0


def test_barrel_contents_equal_the_oracle_rollout(tmp_path):
  """A default-size barrel (1000 episodes, research/data.py:19 BARREL_SIZE) of LuxoBall: every array of the file against the
  CPU oracle stepping the same start poses with the same action tape - lcd bit-exact, full_state/proprio to float32 rounding."""
  from oracle import pyb2o
  from boxlcd_amd.data import fill_barrels, BARREL_SIZE
  assert BARREL_SIZE == 1000
  G = {'ep_len': 9}
  paths = fill_barrels('LuxoBall', 1, tmp_path, 'train', G, seed=11, stamp='20260101T000000')
  b = np.load(paths[0])
  assert b['lcd'].shape == (1000, 9, 16, 24) and b['action'].shape == (1000, 9, 3)
  venv = B.BatchedWorldEnv('LuxoBall', BARREL_SIZE, G, seed=11)      # same Philox streams as the writer
  poses, sel = venv.sample_initial(BARREL_SIZE)
  d = venv.scene.desc
  for e in range(0, BARREL_SIZE, 37):
    o = pyb2o.OracleEnv(d)
    o.reset(poses[e], sel[e])
    for j in range(9):
      assert (b['lcd'][e, j] == o.render().astype(bool)).all(), (e, j)
      fs = o.obs()
      assert np.abs(b['full_state'][e, j] - fs).max() < 1e-6 and np.abs(b['proprio'][e, j] - fs[venv.pobs_idxs]).max() < 1e-6
      o.step(b['action'][e, j].astype(np.float32))
  venv.close()


def test_env_fault_is_returned_by_step():
  """BLCD_ERR_ENV_FAULT: a poisoned environment (NaN pose) makes blcd_step / blcd_rollout report it, the others keep running."""
  from boxlcd_amd._lib import EnvFaultError
  venv = B.BatchedWorldEnv('Object2', 64, seed=2)
  venv.reset()
  h = venv._handle()
  h.step(None, 2)
  bad = h.get_poses()[5:6, :, :3].copy()
  bad[0, 0, 0] = np.nan
  h.set_poses(np.array([5], np.int32), bad, None)
  with pytest.raises(EnvFaultError):
    h.step(None, 1)
  f = h.faults()
  assert f[5] != 0 and (np.delete(f, 5) == 0).all()
  with pytest.raises(EnvFaultError):
    h.rollout(None, 3)
  venv.reset(idxs=[5])
  h.step(None, 1)                       # cleared by the reset
  assert not h.faults().any()
  venv.close()


@pytest.mark.parametrize('variant', [1, 2])
@pytest.mark.parametrize('name', ['Dropbox', 'Bounce2', 'Object2', 'Urchin', 'LuxoBall', 'UrchinCubes', 'Crab'])
def test_render_poses_ex_equals_pillow_and_oracle(name, variant):
  """blcd_render_poses_ex = lcd_render(width, height, lcd_mode) (reference world_env.py:460-512) on the device: the 8x RGB
  human view, native-size RGB, an odd non-proportional RGB canvas and a larger mode-'1' canvas, against Pillow goldens
  (tools/gen_pillow_rgb_goldens.py) and the oracle's sequential renderer."""
  from oracle import pyb2o
  g = np.load('tests/golden/pillow_rgb.npz')
  env = B.BatchedWorldEnv(name, 1, raster_variant=variant)
  d = env.scene.desc
  poses, sel = g[name + '_poses'], g[name + '_sel']
  h = Handle(d, 1, 0)
  for si in range(4):
    w, hh, rgb = g[f'{name}_size{si}'].tolist()
    got = h.render_poses_ex(poses, sel, w, hh, 'RGB' if rgb else '1')
    if variant == 1:     # the goldens are Pillow 12.2's; variant 2 (the recordings' rule, the default) is checked against the oracle
      assert (got == g[f'{name}_frames{si}']).all(), (name, si)
    assert (got == pyb2o.render_poses_ex(d, poses, sel, w, hh, 'RGB' if rgb else '1')).all()
  # the native LCD size too: thin links truncate to one-pixel-high polygons there (the case the variants differ on)
  got = h.render_poses_ex(poses, sel, d.lcd_w, d.lcd_h, '1')
  assert (got == pyb2o.render_poses_ex(d, poses, sel, d.lcd_w, d.lcd_h, '1')).all()
  assert (got == h.render_poses(poses, sel)).all()     # and it is the 1-bit rasteriser's frame
  h.close()


def test_lcd_render_rgb_and_human_view_of_a_live_env():
  """WorldEnv.lcd_render(w, h, 'RGB') / render(mode='human', return_pyglet_view=True): the 8x RGB half of the UrchinBall
  recording, frame by frame, from the HIP path (same replay as tests/test_oracle_replay.py)."""
  import replay as R
  env = B.envs.UrchinBall(raster_variant=2)
  rgb, lcd = R.fixtures('UrchinBall')
  env.seed(7)
  env.reset()
  rs = np.random.RandomState(4)
  for t in range(40):
    env.step(rs.uniform(-1, 1, env.act_size))
    view = env.render(mode='human', return_pyglet_view=True)
    W8 = rgb.shape[2]
    assert (view[:, :W8] == rgb[t]).all(), t
    assert ((view[::8, W8 + 1::8, 0] > 127) == (lcd[t] > 0)).all(), t
  assert env.lcd_render(50, 30, '1').shape == (30, 50) and env.lcd_render(50, 30, 'RGB').shape == (30, 50, 3)
  env.close()


def test_device_resident_vector_env():
  """step_torch / reset_torch: CUDA tensors in and out (what research/rl/ppo.py:127-133-style loops need to run at kernel
  speed); same numbers as the host-copy API stepping an identically seeded twin."""
  import torch
  n = 300
  a = B.BatchedWorldEnv('LuxoBall', n, {'ep_len': 7}, seed=8)
  b = B.BatchedWorldEnv('LuxoBall', n, {'ep_len': 7}, seed=8)
  oa, ob = a.reset_torch(), b.reset()
  assert oa['full_state'].is_cuda and oa['lcd'].dtype == torch.uint8 and oa['proprio'].shape == (n, a.pobs_size)
  assert (oa['full_state'].cpu().numpy() == ob['full_state']).all() and (oa['lcd'].cpu().numpy().astype(bool) == ob['lcd']).all()
  acts = torch.as_tensor(b.sample_actions(7)).cuda()
  for t in range(7):
    o, rew, done, timeout = a.step_torch(acts[t])
    o2, rew2, done2, infos = b.step(acts[t].cpu().numpy())
    assert rew.is_cuda and done.dtype == torch.bool and (done.cpu().numpy() == done2).all() and float(rew.sum()) == 0.0
    assert (o['full_state'].cpu().numpy() == o2['full_state']).all() and (o['lcd'].cpu().numpy().astype(bool) == o2['lcd']).all()
    assert (o['proprio'].cpu().numpy() == o2['proprio']).all()
  assert bool(done.all())
  idx = torch.tensor([2, 5, 9])
  o = a.reset_torch(idx)
  b.reset(idx.numpy())
  assert (o['full_state'].cpu().numpy() == b._obs()['full_state']).all()
  a.close(); b.close()


def test_device_tensors_are_ordered_behind_torchs_stream():
  """`step_torch(policy(obs))`: when the call is made the policy's kernels have been LAUNCHED on torch's stream, not finished, and the
  library steps on its own non-blocking stream - so the handle's stream must first wait for torch's (boxlcd_amd/_lib.py
  Handle._after_torch).  Here the action tensor is the end of a long matmul chain and holds zeros until that chain has run; a step that
  does not wait reads the zeros."""
  import torch
  n = 4096
  a = B.BatchedWorldEnv('Urchin', n, seed=3)
  b = B.BatchedWorldEnv('Urchin', n, seed=3)
  a.reset_torch(); b.reset_torch()
  big = torch.randn(3072, 3072, device='cuda')
  acts = torch.zeros((n, a.act_size), device='cuda')
  for t in range(4):
    want = torch.empty_like(acts).uniform_(-1, 1)
    torch.cuda.synchronize()
    x = big
    for _ in range(30):
      x = (x @ big) * 1e-3                            # tens of milliseconds of queued work
    acts.copy_(want + 0.0 * x[:1, :1].nan_to_num())   # the actions exist only once the chain has run
    oa, *_ = a.step_torch(acts)                       # called while the chain is still running
    fa = oa['full_state'].clone()
    torch.cuda.synchronize()
    acts.zero_()
    ob, *_ = b.step_torch(want)
    assert (fa == ob['full_state']).all(), t
  a.close(); b.close()


def test_episode_clocks_of_the_device_loop():
  """step_torch keeps the episode clocks lazily (a constant `done` tensor while every environment shares a clock, per-environment
  clocks after a partial reset, the host array brought up to date on demand): the `done` flags of a mixed sequence of torch steps,
  numpy steps, partial and full resets equal those of a twin driven through the numpy API alone."""
  import torch
  n, L = 100, 6
  a = B.BatchedWorldEnv('Dropbox', n, {'ep_len': L}, seed=5)
  b = B.BatchedWorldEnv('Dropbox', n, {'ep_len': L}, seed=5)
  a.reset_torch(); b.reset()
  z = np.zeros((n, a.act_size), np.float32)
  zt = torch.zeros((n, a.act_size), device='cuda')
  rs = np.random.RandomState(0)
  script = ['t', 't', 'p', 't', 'n', 't', 'T', 'P', 'I', 't', 'I', 'f', 't', 'T', 'n', 'I', 't', 'p', 'I', 'P', 't', 'T', 'I', 't', 'F', 'I', 't']
  for k, op in enumerate(script):
    if op in 'tTIn':
      if op == 'n':
        _, _, da, _ = a.step(z)
      else:
        _, _, da, _ = a.step_torch(zt, sync={'t': True, 'T': False, 'I': 'inline'}[op])
        da = da.cpu().numpy()
      _, _, db, _ = b.step(z)
      assert (np.asarray(da) == db).all(), (k, op)
      assert (a.ep_t + a._ep_lag == b.ep_t).all(), (k, op)
    elif op in 'pP':
      idx = np.sort(rs.choice(n, 17, replace=False))
      if op == 'p':
        a.reset_torch(torch.as_tensor(idx))
      else:
        a.reset(idx)                                  # a partial reset through the numpy API restarts the device clocks too
      b.reset(idx)
    else:
      if op == 'f':
        a.reset_torch()
      else:
        a.reset()
      b.reset()
  a.close(); b.close()


@pytest.mark.parametrize('mode', [False, 'inline'])
@pytest.mark.parametrize('name,n,T', [('Bounce', 70_000, 130), ('Urchin', 300, 40), ('Object2', 66_000, 25)])
def test_asynchronous_step_loop_equals_the_synchronous_one(name, n, T, mode):
  """step_torch(sync=False) = blcd_step_obs_async: no host synchronisation anywhere in the loop - the step is ordered on the device
  between the torch kernels that produce its actions and those that consume its outputs.  A 'policy' that reads the previous
  observation (so outputs feed inputs) drives an asynchronous env and a synchronous twin: same observations at every step, same
  final state, across a re-bin boundary of the oversubscribed batches (Bounce: every 100 steps)."""
  import torch
  a = B.BatchedWorldEnv(name, n, seed=21)
  b = B.BatchedWorldEnv(name, n, seed=21)
  oa, ob = a.reset_torch(), b.reset_torch()
  W = torch.randn(a.obs_size, max(1, a.act_size), device='cuda')
  policy = lambda o: torch.tanh(o['full_state'].nan_to_num() @ W)[:, :a.act_size].contiguous()
  snaps = []
  for t in range(T):
    oa, _, _, _ = a.step_torch(policy(oa), sync=mode)           # 'inline': queued ON torch's stream (blcd_set_async_stream)
    if mode == 'inline' and t == T // 2:
      assert not a.faults().any()                                # a synchronising call on the handle's own stream in between
    snaps.append((oa['full_state'].clone(), oa['lcd'].clone()))           # queued behind the step on torch's stream
  for t in range(T):
    ob, _, _, _ = b.step_torch(policy(ob))
    assert (snaps[t][0] == ob['full_state']).all() and (snaps[t][1] == ob['lcd']).all(), t
  assert not a.faults().any()
  da, db = a._handle().debug_dump(), b._handle().debug_dump()
  for x, y in zip(da, db):
    assert (x == y).all()
  # the synchronous entry points keep working on a handle that has asynchronous work behind it
  oa, _, _, _ = a.step_torch(policy(oa))
  ob, _, _, _ = b.step_torch(policy(ob))
  assert (oa['full_state'] == ob['full_state']).all()
  a.close(); b.close()


@pytest.mark.parametrize('name,n', [('Bounce', 70_000), ('Dropbox', 1000), ('Object2', 333), ('Urchin', 100), ('Crab', 40)])
def test_step_obs_equals_step_then_get_obs(name, n):
  """blcd_step_obs (one call, one synchronisation: the step kernel writes the observation row and the frame itself) against
  blcd_step + blcd_get_obs on an identically started twin, every step of 30, host and device buffers; final state vs the oracle.
  Bounce-70k is a re-binned two-cohort batch, Crab the 32-row class (separate raster kernel for classes <= 7 bodies only)."""
  import torch
  T = 30
  env = B.BatchedWorldEnv(name, n, seed=17)
  poses, sel = env.sample_initial(n)
  acts = env.sample_actions(T)
  d = env.scene.desc
  ha, hb = Handle(d, n, 0), Handle(d, n, 0)
  ha.reset(None, poses, sel); hb.reset(None, poses, sel)
  fs = np.zeros((n, d.n_obs), np.float32); lcd = np.zeros((n, d.lcd_h, d.lcd_w), np.uint8)
  fs_t = torch.zeros((n, d.n_obs), dtype=torch.float32, device='cuda'); lcd_t = torch.zeros((n, d.lcd_h, d.lcd_w), dtype=torch.uint8, device='cuda')
  for t in range(T):
    if t % 2 == 0:
      ha.step_obs(acts[t], fs, lcd)
    else:
      ha.step_obs(torch.as_tensor(acts[t]).cuda(), fs_t, lcd_t)
      fs, lcd = fs_t.cpu().numpy(), lcd_t.cpu().numpy()
    hb.step(acts[t], 1)
    fs2, lcd2 = hb.get_obs(np.float32)
    assert (fs == fs2).all() and (lcd == lcd2).all(), t
    fs, lcd = np.zeros_like(fs2), np.zeros_like(lcd2)
  assert (ha.debug_dump()[0] == hb.debug_dump()[0]).all()
  idx = np.arange(min(n, 64))
  _, _, olcd, ost = pyb2o.rollout(d, poses[idx], sel[idx], acts[:, idx], T, threads=8)
  assert (ost == ha.debug_dump()[0][idx]).all() and (olcd == lcd2[idx]).all()
  ha.step_obs(None, None, None)      # no outputs: a plain step
  ha.close(); hb.close()


def test_bit_transport_of_lcd_frames():
  """blcd_pack_bits / blcd_unpack_bits (the 1-bit wire format of the multi-GPU gather): exact round trip, numpy bit order."""
  import torch
  from boxlcd_amd import _lib
  _lib.load()
  x = (torch.rand((7, 33, 16, 24), device='cuda') < 0.3).to(torch.uint8)
  p = torch.empty(x.numel() // 8, dtype=torch.uint8, device='cuda')
  y = torch.empty_like(x)
  s = torch.cuda.current_stream().cuda_stream
  _lib.pack_bits(x, p, s)
  _lib.unpack_bits(p, y, s)
  torch.cuda.synchronize()
  assert (x == y).all()
  assert (p.cpu().numpy() == np.packbits(x.cpu().numpy().reshape(-1), bitorder='little')).all()


def _gather_rank(rank, world, port, q):
  import os, torch
  from boxlcd_amd import dist as bdist
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  bdist.init_from_env('gloo')                      # both ranks share cuda:0 here; on a node it is nccl, one GPU per rank
  torch.cuda.set_device(0)
  Tc, n = 4, 96
  g = bdist.ChunkGatherer(world, [torch.empty((Tc, n, 16, 16), dtype=torch.uint8, device='cuda'), torch.empty((Tc, n, 4), device='cuda')], binary=[True, False])
  ok = True
  for chunk in range(3):
    gen = torch.Generator(device='cuda'); gen.manual_seed(100 * chunk + rank)
    lcd = (torch.rand((Tc, n, 16, 16), device='cuda', generator=gen) < 0.4).to(torch.uint8)
    obs = torch.rand((Tc, n, 4), device='cuda', generator=gen)
    g.gather([lcd, obs], producer=torch.cuda.current_stream())
    g.finish()
    gl, go = g.last()
    for r in range(world):
      gen2 = torch.Generator(device='cuda'); gen2.manual_seed(100 * chunk + r)
      el = (torch.rand((Tc, n, 16, 16), device='cuda', generator=gen2) < 0.4).to(torch.uint8)
      eo = torch.rand((Tc, n, 4), device='cuda', generator=gen2)
      ok = ok and bool((gl[r] == el).all()) and bool((go[r] == eo).all())
  q.put((rank, ok))
  torch.distributed.destroy_process_group()


def test_chunk_gather_of_full_rollout_tensors_two_ranks():
  """The N>1 data path of bench.py --gpus N: every rank receives every rank's full chunk tensors; LCD frames travel at one bit
  per pixel and arrive as uint8.  Two processes on this one GPU (gloo); on a node the same code runs over RCCL."""
  import socket
  import torch.multiprocessing as mp
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_gather_rank, args=(r, 2, port, q)) for r in range(2)]
  for p in procs: p.start()
  res = [q.get(timeout=240) for _ in procs]
  for p in procs: p.join(timeout=60)
  assert sorted(res) == [(0, True), (1, True)]


def _rccl_rank(port, mode, q):
  import os, torch
  from boxlcd_amd import dist as bdist
  os.environ.update(RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  try:
    torch.cuda.set_device(0)
    # a ONE-rank RCCL communicator (torch backend 'nccl'): the same ChunkGatherer calls bench.py --gpus N issues, on the real backend
    torch.distributed.init_process_group(backend='nccl', rank=0, world_size=1, init_method=f'tcp://127.0.0.1:{port}')
    Tc, n = 4, 96
    g = bdist.ChunkGatherer(1, [torch.empty((Tc, n, 16, 16), dtype=torch.uint8, device='cuda'), torch.empty((Tc, n, 4), device='cuda')],
                            binary=[True, False], mode=mode, rank=0)
    ok = True
    for chunk in range(3):
      gen = torch.Generator(device='cuda'); gen.manual_seed(chunk)
      lcd = (torch.rand((Tc, n, 16, 16), device='cuda', generator=gen) < 0.4).to(torch.uint8)
      obs = torch.rand((Tc, n, 4), device='cuda', generator=gen)
      g.gather([lcd, obs], producer=torch.cuda.current_stream())
      g.finish()
      gl, go = g.last()
      ok = ok and bool((gl[0] == lcd).all()) and bool((go[0] == obs).all())
    t = torch.tensor([3.5], dtype=torch.float64, device='cuda')
    ok = ok and bdist.max_over_ranks(3.5) == 3.5
    bdist.barrier()
    q.put(('ok', ok))
    torch.distributed.destroy_process_group()
  except Exception as ex:      # report instead of hanging the parent on the queue
    q.put(('error', repr(ex)))


@pytest.mark.parametrize('mode', ['all', 'consumer'])
def test_chunk_gather_on_the_rccl_backend_single_rank(mode):
  """The gather path on its REAL backend: torch.distributed 'nccl' (= RCCL on ROCm) with a one-rank communicator on this one GPU -
  `all_gather_into_tensor` / `gather` of bit-packed frames and float32 observations on the side stream, `max_over_ranks`, `barrier`,
  exactly as `bench.py --gpus N` issues them.  (Two ranks cannot share a GPU under RCCL; the multi-rank data movement is the
  driver's 8-GPU run, the two-rank logic is covered on gloo above.)"""
  import socket
  import torch.multiprocessing as mp
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  p = ctx.Process(target=_rccl_rank, args=(port, mode, q))
  p.start()
  res = q.get(timeout=240)
  p.join(timeout=60)
  assert res == ('ok', True), res


@pytest.mark.parametrize('name,n', [('Bounce', 4096), ('Dropbox', 1000), ('LuxoBall', 300), ('Urchin', 257), ('Crab', 64)])
def test_rollout_bits_unpacks_to_the_uint8_frames(name, n):
  """blcd_rollout_bits: the row masks the raster holds, one bit per pixel (north_star's "1-bit framebuffer"; `lcd` is a bool array
  in the reference, world_env.py:508-509).  numpy.unpackbits(bitorder='little') of it equals blcd_rollout's uint8 frames for
  16-, 24-, 32- and 64-pixel-wide LCDs, full and ragged waves; observations and final state are those of blcd_rollout."""
  T = 25
  env = B.BatchedWorldEnv(name, n, seed=77)
  poses, sel = env.sample_initial(n)
  acts = env.sample_actions(T)
  d = env.scene.desc
  outs = []
  for bits in (False, True):
    h = Handle(d, n, 0)
    h.reset(None, poses, sel)
    obs = np.zeros((T, n, d.n_obs), np.float32)
    if bits:
      lcd = np.full((T, n, d.lcd_h, d.lcd_w // 8), 0xAA, np.uint8)
      h.rollout_bits(acts, T, lcd, obs)
      lcd = np.unpackbits(lcd, axis=-1, bitorder='little')
    else:
      lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
      h.rollout(acts, T, lcd, obs)
    outs.append((lcd, obs, h.debug_dump()[0].copy()))
    h.close()
  assert outs[0][0].shape == outs[1][0].shape and (outs[0][0] == outs[1][0]).all()
  assert (outs[0][1] == outs[1][1]).all() and (outs[0][2] == outs[1][2]).all()


@pytest.mark.parametrize('name', ['Dropbox', 'Bounce', 'Bounce2', 'Object2', 'Object3', 'Urchin', 'Luxo', 'UrchinBall', 'LuxoBall', 'UrchinCube',
                                  'LuxoCube', 'UrchinBalls', 'LuxoBalls', 'UrchinCubes', 'LuxoCubes', 'Crab', 'CrabCube', 'SpiderCube'])
def test_device_reset_sampler_equals_its_host_mirror(name):
  """blcd_reset_sampled (what reset() / reset_torch() run: poses are never built on the host) against the numpy restatement of
  the same counter-based stream and the same float64 arithmetic (BatchedWorldEnv.mirror_poses): body poses and shape choices
  bit for bit, for a full reset, a second full reset (reset count 1) and a partial reset."""
  n = 3000
  env = B.BatchedWorldEnv(name, n, seed=123)
  env.reset()
  h = env._handle()
  p0, s0 = env.mirror_poses(np.arange(n), np.zeros(n, np.int64))
  assert (h.get_poses()[:, :, :3] == p0).all() and (h.shape_sel() == s0).all()
  env.reset()
  p1, s1 = env.mirror_poses(np.arange(n), np.ones(n, np.int64))
  assert (h.get_poses()[:, :, :3] == p1).all() and (p1 != p0).any()
  idxs = np.array([5, 77, 2999, 1024], np.int32)
  env.reset(idxs)
  got = h.get_poses()[:, :, :3]
  p2, _ = env.mirror_poses(idxs, np.full(len(idxs), 2))
  keep = np.setdiff1d(np.arange(n), idxs)
  assert (got[idxs] == p2).all() and (got[keep] == p1[keep]).all()
  # ... and the oracle started from the mirror's poses steps exactly like the device-reset batch
  acts = env.sample_actions(5)
  for t in range(5):
    env.step(acts[t])
  st = h.debug_dump()[0]
  pm = p1.copy(); pm[idxs] = p2
  sm = h.shape_sel()
  sub = np.array([0, 5, 77, 1500, 2999])
  _, _, _, ost = pyb2o.rollout(env.scene.desc, pm[sub], sm[sub], acts[:, sub], 5, threads=4)
  assert (ost == st[sub]).all()
  env.close()


@pytest.mark.parametrize('name', ['Object2', 'LuxoBall'])
def test_shards_with_one_seed_are_one_batch(name):
  """Sharded batches: every rank uses the SAME seed and env_id_base = rank * envs-per-rank; the device sampler keys
  its Philox counter with the global env id, so the shards' starts - poses and 'random' shape choices - are exactly the starts
  of one batch of world x n environments, at the first and at later resets, and a partial reset on one shard draws what the
  whole batch would draw for those environments."""
  n, world = 1000, 2
  whole = B.BatchedWorldEnv(name, n * world, seed=77)
  whole.reset()
  whole.reset(np.array([3, n + 3], np.int32))
  wp, ws = whole._handle().get_poses()[:, :, :3], whole._handle().shape_sel()
  for r in range(world):
    shard = B.BatchedWorldEnv(name, n, seed=77, env_id_base=r * n)
    shard.reset()
    shard.reset(np.array([3], np.int32))
    h = shard._handle()
    assert (h.get_poses()[:, :, :3] == wp[r * n:(r + 1) * n]).all() and (h.shape_sel() == ws[r * n:(r + 1) * n]).all()
    pm, sm = shard.mirror_poses(np.arange(n), np.where(np.arange(n) == 3, 1, 0))      # the host mirror follows the base too
    assert (pm == wp[r * n:(r + 1) * n]).all() and (sm == ws[r * n:(r + 1) * n]).all()
    shard.close()
  assert (wp[:n] != wp[n:]).any()
  whole.close()


def test_snapshot_carries_the_reset_counters_and_duplicate_indices_are_refused():
  """blcd_get_state / blcd_set_state include the device sampler's per-environment reset counts (a resumed run draws the same
  reset stream); an index list naming an environment twice is refused (two threads would rebuild one world)."""
  env = B.BatchedWorldEnv('Object2', 512, seed=3)
  env.reset()
  env.reset(np.array([7, 9], np.int32))
  h = env._handle()
  blob = h.get_state()
  env.reset()
  after = (h.get_poses().copy(), h.shape_sel().copy())
  env.reset()                       # moves the counters on
  h.set_state(blob)
  env.reset()                       # ... and from the restored snapshot the same draw comes again
  assert (h.get_poses() == after[0]).all() and (h.shape_sel() == after[1]).all()
  bad = blob.copy(); bad[4] ^= 1    # another library version
  with pytest.raises(RuntimeError, match='version'):
    h.set_state(bad)
  for call in (lambda: env.reset(np.array([5, 6, 5], np.int32)), lambda: h.set_poses(np.array([1, 1], np.int32), np.zeros((2, 2, 3), np.float32)),
               lambda: h.reset(np.array([2, 2], np.int32), np.zeros((2, 2, 3), np.float32))):
    with pytest.raises(RuntimeError, match='twice'):
      call()
  env.close()


@pytest.mark.parametrize('name,n', [('Dropbox', 1024), ('Dropbox', 200)])
def test_environments_at_rest_reemit_identical_outputs(name, n):
  """The general one-body class skips the world steps of an environment whose body is asleep and re-emits the frame rows /
  observation row it computed a step earlier (step_kernel).  A fused rollout long enough for most boxes to fall
  asleep - in waves that mix sleeping and moving environments, full (1024) and ragged (200) - must equal the unfused path (one
  blcd_step + blcd_get_obs per env-step: no reuse there) frame for frame, observation for observation and in its final state,
  for uint8 frames, 1-bit frames, and with either output missing."""
  T = 150
  env = B.BatchedWorldEnv(name, n, seed=5)
  poses, sel = env.sample_initial(n)
  d = env.scene.desc
  h = Handle(d, n, 0)
  h.reset(None, poses, sel)
  ref_lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
  ref_obs = np.zeros((T, n, d.n_obs), np.float32)
  for t in range(T):
    h.step(None, 1)
    fs, lcd = h.get_obs(np.float32)
    ref_obs[t], ref_lcd[t] = fs, lcd
  ref_state = h.debug_dump()[0].copy()
  asleep = ref_state[:, 0, 7] == 0.0
  assert asleep.mean() > 0.5     # the boxes fall asleep one by one during the rollout: sleeping and moving environments share waves on the way
  h.close()
  for variant in ('u8', 'bits', 'lcd_only', 'obs_only'):
    h = Handle(d, n, 0)
    h.reset(None, poses, sel)
    obs = np.zeros((T, n, d.n_obs), np.float32) if variant != 'lcd_only' else None
    if variant == 'bits':
      lcd = np.zeros((T, n, d.lcd_h, d.lcd_w // 8), np.uint8)
      h.rollout_bits(None, T, lcd, obs)
      lcd = np.unpackbits(lcd, axis=-1, bitorder='little')
    else:
      lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8) if variant != 'obs_only' else None
      h.rollout(None, T, lcd, obs)
    if lcd is not None:
      assert (lcd == ref_lcd).all(), variant
    if obs is not None:
      assert (obs == ref_obs).all(), variant
    assert (h.debug_dump()[0] == ref_state).all(), variant
    h.close()
