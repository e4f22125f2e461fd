"""Parity tests proper: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs, world step by
world step, bit-exact on every state word (positions, velocities, sleep timers, broad-phase AABBs, manifolds, accumulated
impulses, joint limit states); LCD frames bit-exact; observations within 1e-6.  Plus size-independent properties at the
BASELINE sizes where the oracle would be too slow."""
import numpy as np
import pytest
import parity
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
from oracle import pyb2o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name,n,steps', [('Dropbox', 256, 200), ('Bounce', 256, 200), ('Object2', 256, 200),
                                          ('Urchin', 128, 200), ('LuxoBall', 128, 200)])
def test_baseline_configs_bit_exact_per_world_step(name, n, steps):
  cnt, msgs = parity.run_substep_parity(name, n, steps, seed=11)
  assert cnt == steps * 3 and not msgs, msgs[:5]


@pytest.mark.parametrize('name', ['Bounce2', 'Object3', 'Luxo', 'UrchinCube', 'LuxoCube', 'UrchinBall', 'UrchinBalls', 'LuxoBalls',
                                  'UrchinCubes', 'LuxoCubes', 'Crab', 'CrabCube', 'SpiderCube'])
def test_rest_of_catalogue_bit_exact(name):
  cnt, msgs = parity.run_substep_parity(name, 32, 60, seed=5)
  assert not msgs, msgs[:5]


def test_device_sincos_equals_oracle():
  rng = np.random.RandomState(0)
  # every float within 2 000 ulps of each range boundary of the routine (the device folds some ranges into one pass)
  edges = []
  for t in (2.0 ** -27, 2.0 ** -5, np.pi / 4, 9 * np.pi / 4, 2.0 ** 23):
    u = np.float32(t).view(np.uint32)
    w = (u + np.arange(-2000, 2001)).astype(np.uint32).view(np.float32)
    edges += [w, -w]
  x = np.concatenate([rng.uniform(-130, 130, 1 << 20), rng.uniform(-1, 1, 1 << 18), rng.uniform(-1e-2, 1e-2, 4096),
                      rng.uniform(-1e6, 1e6, 4096), np.concatenate(edges),
                      [0.0, -0.0, 0.75, 0.7853982, 119.99, 120.0, 1e9]]).astype(np.float32)
  from boxlcd_amd import _lib
  s, c = np.zeros_like(x), np.zeros_like(x)
  _lib._check(_lib.load().blcd_debug_sincos(_lib._ptr(x), x.size, _lib._ptr(s), _lib._ptr(c), 0))
  so, co = pyb2o.sincos(x)
  assert (s == so).all() and (c == co).all()


def test_mass_data_equals_oracle():
  from boxlcd_amd import _lib
  for name in ['Dropbox', 'Bounce', 'LuxoBall', 'Crab']:
    d = getattr(B.envs, name)().scene.desc
    for sh in range(d.n_shapes):
      out = np.zeros(24, np.float32)
      _lib._check(_lib.load().blcd_debug_mass_data(d, sh, 0.37, _lib._ptr(out)))
      assert (out == pyb2o.mass_data(d, sh, 0.37)).all(), (name, sh)


@pytest.mark.parametrize('name,variant', [('Urchin', 0), ('Urchin', 1), ('Urchin', 2), ('LuxoBall', 1), ('LuxoBall', 2), ('Crab', 1), ('Crab', 2),
                                          ('Object2', 0)])
def test_render_poses_equals_oracle_and_pillow(name, variant):
  env = B.BatchedWorldEnv(name, 1, raster_variant=variant)
  g = np.load('tests/golden/pillow_render.npz')
  poses, sel = g[name + '_poses'], g[name + '_sel']
  h = Handle(env.scene.desc, 1, 0)
  got = h.render_poses(poses, sel)
  assert (got == pyb2o.render_poses(env.scene.desc, poses, sel)).all()
  if variant == 1:
    exp = np.unpackbits(g[name + '_frames'], axis=-1, bitorder='little')[..., :env.scene.desc.lcd_w]
    assert (got == exp).all()
  h.close()


def test_recordings_replayed_on_device_from_recorder_inputs():
  """The reference's published recordings reproduced by the HIP path itself from the recorder's own inputs (env.seed(S) reset
  sample + RandomState(A) action tape, tests/replay.py; no fitted pose): every LCD frame of the eight recordings the oracle
  reproduces exactly, and the same frame counts as the oracle on the four open ones (tests/test_oracle_replay.py)."""
  import replay as R
  for gif, (cls, force_sel, seed, aseed) in R.GIFS.items():
    env = getattr(B.envs, cls)(raster_variant=2)
    d = env.scene.desc
    _, exp = R.fixtures(gif)
    T = len(exp)
    P, sel = R.recorder_start(env, seed)
    if force_sel is not None:
      sel = np.array(force_sel, np.int32)
    rs = np.random.RandomState(aseed)
    acts = np.stack([rs.uniform(-1, 1, env.act_size) for _ in range(T)]).astype(np.float32)[:, None, :]
    h = Handle(d, 1, 0)
    h.reset(None, P[None].astype(np.float32), np.asarray(sel, np.int32)[None])
    lcd = np.zeros((T, 1, d.lcd_h, d.lcd_w), np.uint8)
    h.rollout(acts, T, lcd_out=lcd)
    bad = [int((lcd[t, 0] != exp[t]).sum()) for t in range(T)]
    o = pyb2o.OracleEnv(d)
    o.reset(np.asarray(P, np.float32), sel)
    obad = []
    for t in range(T):
      o.step(acts[t, 0])
      obad.append(int((o.render() != exp[t]).sum()))
    assert bad == obad, gif                                   # device == oracle, frame by frame
    # north_star's pose tolerance (1e-4 on positions / angles) against the oracle at the end of the recording: met with zero error
    dev, ora = h.debug_dump()[0][0], o.dump()[0]
    assert np.abs(dev[:, :3] - ora[:, :3]).max() <= 1e-4 and (dev == ora).all(), gif
    if gif in ('Dropbox', 'Bounce', 'Bounce2', 'Object2', 'Object2_circles', 'Object2_cubes', 'UrchinBall', 'UrchinCube'):
      assert sum(bad) == 0, (gif, R.summary(bad))
    # the four open recordings: every LCD frame of the physics-exact prefix, incl. the one-pixel-high foot polygon of Luxo frame 37 /
    # LuxoBall frame 38 (raster variant 2's scan-position rule; tests/test_oracle_replay.py)
    prefix = {'Urchin': 17, 'Luxo': 66, 'LuxoBall': 57, 'LuxoCube': 33}.get(gif)
    if prefix is not None:
      assert sum(bad[:prefix]) == 0, (gif, [i for i, b in enumerate(bad[:prefix]) if b])
    h.close()


def test_full_size_properties_bounce_100k():
  """BASELINE configs[1] size: determinism, shard-invariance, snapshot/restore, and physical sanity (no oracle)."""
  n, T = 100_000, 60
  env = B.BatchedWorldEnv('Bounce', n, seed=1)
  poses, sel = env.sample_initial(n)
  h = Handle(env.scene.desc, n, 0)
  h.reset(None, poses, sel)
  h.step(None, 20)
  snap = h.get_state()
  h.step(None, T - 20)
  b1 = h.debug_dump()[0]
  _, lcd1 = h.get_obs(None)
  assert np.isfinite(b1).all() and not h.faults().any()
  assert (b1[:, 0, 1] > 0.45).all() and (b1[:, 0, 1] < 4.6).all() and (b1[:, 0, 0] > 0.45).all() and (b1[:, 0, 0] < 4.55).all()
  assert (lcd1.reshape(n, -1).min(1) == 0).all()                       # every ball is visible
  assert np.all((16 * 16 - lcd1.reshape(n, -1).sum(1)) <= 25)         # r=0.5 ball covers at most a 5x5 pattern
  # restore + continue == uninterrupted
  h.set_state(snap)
  h.step(None, T - 20)
  assert (h.debug_dump()[0] == b1).all()
  h.close()
  # two half-size handles (the multi-GPU sharding unit) == one full handle
  half = n // 2
  for lo, hi in [(0, half), (half, n)]:
    hh = Handle(env.scene.desc, hi - lo, 0)
    hh.reset(None, poses[lo:hi], sel[lo:hi])
    hh.step(None, T)
    assert (hh.debug_dump()[0] == b1[lo:hi]).all()
    hh.close()
  # a sample of the full-size batch against the oracle
  idx = np.random.RandomState(0).choice(n, 64, replace=False)
  _, _, olcd, ost = pyb2o.rollout(env.scene.desc, poses[idx], sel[idx], None, T, threads=8)
  assert (ost == b1[idx]).all() and (olcd == lcd1[idx]).all()


def test_partial_reset_and_set_poses():
  n = 64
  env = B.BatchedWorldEnv('Object2', n, seed=4)
  poses, sel = env.sample_initial(n)
  h = Handle(env.scene.desc, n, 0)
  h.reset(None, poses, sel)
  h.step(None, 15)
  before = h.debug_dump()[0].copy()
  idxs = np.array([3, 17, 40], np.int32)
  h.reset(idxs, poses[idxs], sel[idxs])
  after = h.debug_dump()[0]
  keep = np.setdiff1d(np.arange(n), idxs)
  assert (after[keep] == before[keep]).all()
  oras = []
  for e in idxs:
    o = pyb2o.OracleEnv(env.scene.desc); o.reset(poses[e], sel[e]); oras.append(o)
  assert all((after[e] == o.dump()[0]).all() for e, o in zip(idxs, oras))
  # SetTransform path (reset(full_state=), world_env.py:333-380): FindNewContacts after every setter, as Box2D 2.3.0 does -
  # the whole state (bodies incl. fat AABBs, pairs, contact order through the following steps) equals the oracle's
  newp = poses[idxs].copy(); newp[..., 0] = 2.5; newp[:, 0, 1] = 1.0; newp[:, 1, 1] = 3.0; newp[..., 2] = 0.3
  h.set_poses(idxs, newp, None)
  for o, p in zip(oras, newp):
    o.set_poses(p)
  _, lcd = h.get_obs(None)
  db, _, dp = h.debug_dump()
  for e, o in zip(idxs, oras):
    ob, _, op = o.dump()
    assert (lcd[e] == o.render()).all() and (db[e] == ob).all() and (dp[e][:len(op)] == op).all()
  h.close()


@pytest.mark.parametrize('name', ['Object3', 'UrchinBalls', 'LuxoCubes'])
def test_state_injection_right_after_reset_keeps_contact_order(name):
  """reset(full_state=) as the reference runs it: fresh bodies (proxies still in the move buffer), then two SetTransforms per
  body with FindNewContacts after each.  Bodies are injected on top of each other / of the floor so that several contacts
  appear at t = 0; the order they are created in decides the solver order, so 30 env-steps later every state word must still
  equal the oracle's."""
  n = 48
  env = B.BatchedWorldEnv(name, n, seed=21)
  poses, sel = env.sample_initial(n)
  h = Handle(env.scene.desc, n, 0)
  h.reset(None, poses, sel)
  inj = poses.copy()
  nb = inj.shape[1]
  rng = np.random.RandomState(3)
  inj[:, :, 0] = 2.0 + 0.35 * np.arange(nb)[None, :] + rng.uniform(-0.05, 0.05, (n, nb))   # a touching row of bodies...
  inj[:, :, 1] = 0.55 + rng.uniform(0.0, 0.1, (n, nb))                                     # ...resting on the floor
  if env.scene.desc.n_joints:                                                                # robots: keep links on their anchors
    inj[:, :env.scene.desc.n_joints + 1] = poses[:, :env.scene.desc.n_joints + 1]
    inj[:, env.scene.desc.n_joints + 1:, 0] = poses[:, 0:1, 0] + 0.4 * (1 + np.arange(nb - env.scene.desc.n_joints - 1))[None, :]
  h.set_poses(None, inj, None)
  acts = env.sample_actions(30)
  oras = []
  for e in range(n):
    o = pyb2o.OracleEnv(env.scene.desc)
    o.reset(poses[e], sel[e])
    o.set_poses(inj[e])
    oras.append(o)
  db, dj, dp = h.debug_dump()
  for e, o in enumerate(oras):
    ob, oj, op = o.dump()
    assert (db[e] == ob).all() and (dp[e][:len(op)] == op).all(), e
  created = sum(int((o.dump()[2][:, 0] > 0).sum()) for o in oras)
  assert created >= 3 * n                                   # several contacts per environment exist before the first step
  for t in range(30):
    h.step(acts[t], 1)
    for e, o in enumerate(oras):
      o.step(acts[t, e])
  db, dj, dp = h.debug_dump()
  for e, o in enumerate(oras):
    ob, oj, op = o.dump()
    assert (db[e] == ob).all() and (dj[e][:len(oj)] == oj).all() and (dp[e][:len(op)] == op).all(), e
  assert not h.faults().any()
  h.close()


@pytest.mark.parametrize('name,n,T', [('Urchin', 50_000, 200), ('LuxoBall', 50_000, 200), ('Object2', 200_000, 200),
                                     ('Dropbox', 100_000, 200), ('Bounce', 100_000, 200)])
def test_full_size_baseline_batches(name, n, T):
  """Every BASELINE workload at its full batch size AND its stated length (200 env-steps; one GPU's worth - the 8-GPU configs
  shard exactly this): fused rollout, no faults, physical bounds, every environment drawn, and a 64-environment sample of the
  batch against the oracle at the final step."""
  env = B.BatchedWorldEnv(name, n, seed=31)
  poses, sel = env.sample_initial(n)
  acts = env.sample_actions(T)
  d = env.scene.desc
  h = Handle(d, n, 0)
  h.reset(None, poses, sel)
  lcd_last = np.zeros((1, n, d.lcd_h, d.lcd_w), np.uint8)
  h.rollout(acts[:T - 1], T - 1)
  h.rollout(acts[T - 1:], 1, lcd_out=lcd_last)
  st = h.debug_dump()[0]
  assert np.isfinite(st).all() and not h.faults().any()
  assert (st[:, :, 0] > -0.5).all() and (st[:, :, 0] < d.world_w + 0.5).all() and (st[:, :, 1] > -0.5).all() and (st[:, :, 1] < d.world_h + 0.5).all()
  assert (lcd_last[0].reshape(n, -1).min(1) == 0).all()
  idx = np.random.RandomState(1).choice(n, 64, replace=False)
  _, _, olcd, ost = pyb2o.rollout(d, poses[idx], sel[idx], acts[:, idx], T, threads=8)
  assert (ost == st[idx]).all() and (olcd == lcd_last[0][idx]).all()
  h.close()


@pytest.mark.parametrize('name', sorted(B.env_map))
def test_every_frame_of_a_fused_rollout_equals_the_oracle(name):
  """The soak of tools/soak.py inside the suite: every env class of the catalogue, 256 environments, 200
  env-steps with random actions through blcd_rollout - the FUSED path the bench times (chunked launches, frames and observation
  rows written by step_kernel itself) - and EVERY one of the 200 x 256 LCD frames bit for bit, every observation row to 1e-6
  (float64 sin/cos: ocml vs glibc) and the final body state bit for bit against the oracle stepped one env-step at a time."""
  n, T = 256, 200
  env = B.BatchedWorldEnv(name, n, seed=2024)
  poses, sel = env.sample_initial(n)
  acts = env.sample_actions(T)
  d = env.scene.desc
  h = Handle(d, n, 0)
  h.reset(None, poses, sel)
  lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
  obs = np.zeros((T, n, d.n_obs), np.float32)
  h.rollout(acts, T, lcd, obs)
  state = h.debug_dump()[0]
  assert not h.faults().any()
  h.close()
  _, oobs, olcd, ost = pyb2o.rollout_frames(d, poses, sel, acts, T, threads=16)
  bad = np.argwhere(~(lcd == olcd).reshape(T, n, -1).all(2))
  assert len(bad) == 0, f'{len(bad)} of {T * n} frames differ, first (t, env) = {bad[0]}'
  assert np.abs(obs - oobs).max() < 1e-6
  assert (state == ost).all()


@pytest.mark.parametrize('name,n', [('UrchinBall', 100), ('Bounce', 1000), ('Object2', 77)])
def test_fused_rollout_ragged_batch_matches_oracle(name, n):
  """blcd_rollout (fused chunks, re-binning between chunks, per-step LCD/obs rows) on batch sizes that are not multiples of the
  wave width: final state + every per-step LCD frame of a few envs against the oracle stepped one env-step at a time."""
  T = 45
  env = B.BatchedWorldEnv(name, n, seed=9)
  poses, sel = env.sample_initial(n)
  acts = env.sample_actions(T)
  h = Handle(env.scene.desc, n, 0)
  h.reset(None, poses, sel)
  lcd = np.zeros((T, n, env.scene.desc.lcd_h, env.scene.desc.lcd_w), np.uint8)
  obs = np.zeros((T, n, env.scene.desc.n_obs), np.float32)
  h.rollout(acts, T, lcd, obs)
  state = h.debug_dump()[0]
  _, oobs, olcd, ost = pyb2o.rollout(env.scene.desc, poses, sel, acts, T, threads=8)
  assert (state == ost).all() and (lcd[-1] == olcd).all() and np.abs(obs[-1] - oobs).max() < 1e-6
  for e in (0, n // 2, n - 1):           # per-step rows of three envs, including the last (ragged) lane
    o = pyb2o.OracleEnv(env.scene.desc)
    o.reset(poses[e], sel[e])
    for t in range(T):
      o.step(acts[t, e])
      assert (lcd[t, e] == o.render()).all() and np.abs(obs[t, e] - o.obs()).max() < 1e-6, (e, t)
  assert not h.faults().any()
  h.close()


@pytest.mark.parametrize('name', ['Urchin', 'LuxoBall', 'Object2'])
def test_rollout_results_do_not_depend_on_the_launch_chunking(name, monkeypatch):
  """blcd_rollout sizes its fused launches itself (joint-free scenes: 20 env-steps + re-binning; jointed scenes: up to 200,
  from the previous rollout's time per step).  Whatever the chunking, frames, observations and final state are the same."""
  n, T = 192, 60
  env, poses, sel = parity.make_batch(name, n, 5)
  d = env.scene.desc
  acts = np.random.RandomState(6).uniform(-1, 1, (T, n, max(1, d.n_act))).astype(np.float32)[:, :, :d.n_act]
  outs = []
  for chunk in (None, '7', '1'):
    if chunk is None:
      monkeypatch.delenv('BLCD_CHUNK', raising=False)
    else:
      monkeypatch.setenv('BLCD_CHUNK', chunk)
    h = Handle(d, n, 0)
    res = []
    for rep in range(2):      # the second rollout of the adaptive handle uses the measured time per step
      h.reset(None, poses, sel)
      lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
      obs = np.zeros((T, n, d.n_obs), np.float32)
      h.rollout(acts, T, lcd, obs)
      res.append((lcd, obs, [x.copy() for x in h.debug_dump()]))
    h.close()
    outs.append(res)
  ref = outs[0][0]
  for res in outs:
    for lcd, obs, dump in res:
      assert (lcd == ref[0]).all() and (obs == ref[1]).all()
      for a, b in zip(dump, ref[2]):
        assert (a == b).all()


def test_cohorts_of_an_oversubscribed_batch_change_nothing(monkeypatch):
  """A joint-free batch larger than 64 environments per SIMD is stepped as two slot ranges on two streams, each re-binned within
  itself (no chunk-boundary barrier across the whole batch).  Frames, observations and final state equal the single-range run."""
  n, T = 70_000, 60
  outs = []
  for name in ('Bounce', 'Object2'):
    env, poses, sel = parity.make_batch(name, n, 9)
    d = env.scene.desc
    res = []
    # + the round-4 placement knobs: three cohorts, settling-phase chunks without the sort, cohorts on exchanged streams
    for cohorts, extra in (('1', {}), ('2', {}), ('3', {}), ('2', {'BLCD_CHUNK0': '30:7'}), ('2', {'BLCD_CHUNK0': '40:10:0', 'BLCD_COHORT_SWAP': '1'})):
      for k in ('BLCD_CHUNK0', 'BLCD_COHORT_SWAP'):
        monkeypatch.delenv(k, raising=False)
      for k, v in extra.items():
        monkeypatch.setenv(k, v)
      monkeypatch.setenv('BLCD_COHORTS', cohorts)
      h = Handle(d, n, 0)
      h.reset(None, poses, sel)
      lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
      obs = np.zeros((T, n, d.n_obs), np.float32)
      h.rollout(None, T, lcd, obs)
      h.step(None, 3)                                   # the single-stream path right after a cohort rollout
      res.append((lcd, obs, [x.copy() for x in h.debug_dump()], h.faults().copy()))
      h.close()
    a = res[0]
    for b in res[1:]:
      assert (a[0] == b[0]).all() and (a[1] == b[1]).all(), name
      for x, y in zip(a[2], b[2]):
        assert (x == y).all(), name
      assert (a[3] == 0).all() and (b[3] == 0).all()


@pytest.mark.parametrize('name,n', [('Object2', 10), ('Object2', 64), ('Object2', 65), ('Object2', 130), ('Object2', 255),
                                    ('Dropbox', 10), ('Dropbox', 130), ('Bounce', 65), ('Bounce', 1000)])
def test_rebin_and_cohort_knobs_on_small_batches_change_nothing(monkeypatch, name, n):
  """BLCD_REBIN=1 / BLCD_COHORTS=4 are documented as placement-only knobs: on batches too small to give every cohort a whole wave
  the handle falls back to fewer cohorts (never an empty or out-of-range slot range) and results equal the default run.  On the
  one-body scenes a forced re-bin also brings the narrow waves of the unsorted first chunk (8 lanes + 56 shadow lanes each) and of
  the two-width launches to batches of a few waves."""
  T = 40
  env, poses, sel = parity.make_batch(name, n, 3)
  d = env.scene.desc
  res = []
  for knobs in ({}, {'BLCD_REBIN': '1'}, {'BLCD_REBIN': '1', 'BLCD_COHORTS': '4'}, {'BLCD_REBIN': '1', 'BLCD_LANES': '64', 'BLCD_CHUNK': '7', 'BLCD_TWO_WIDTHS': '5'},
                {'BLCD_REBIN': '1', 'BLCD_LANES': '64', 'BLCD_UNSORTED_SPREAD': '0', 'BLCD_CHUNK': '10'}, {'BLCD_REBIN': '1', 'BLCD_LANES': '64'}):
    # (BLCD_LANES=64: small batches otherwise run 16-lane waves, which switches the two-width paths off)
    for k in ('BLCD_REBIN', 'BLCD_COHORTS', 'BLCD_CHUNK', 'BLCD_TWO_WIDTHS', 'BLCD_UNSORTED_SPREAD', 'BLCD_LANES'):
      monkeypatch.delenv(k, raising=False)
    for k, v in knobs.items():
      monkeypatch.setenv(k, v)
    h = Handle(d, n, 0)
    h.reset(None, poses, sel)
    lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
    obs = np.zeros((T, n, d.n_obs), np.float32)
    h.rollout(None, T, lcd, obs)
    res.append((lcd, obs, [x.copy() for x in h.debug_dump()]))
    assert not h.faults().any()
    h.close()
  for b in res[1:]:
    assert (res[0][0] == b[0]).all() and (res[0][1] == b[1]).all()
    for x, y in zip(res[0][2], b[2]):
      assert (x == y).all()


def _needs_sched_build():
  """the three environment-level schedulers are opt-in at build time (BLCD_DEFS=-DBLCD_SCHED; DESIGN.md 4.4: bit-neutral, slower)"""
  from boxlcd_amd import _lib
  if not _lib.features() & _lib.FEATURE_SCHED:
    pytest.skip('library built without -DBLCD_SCHED (the default): the rejected schedulers are compiled out')


def test_default_build_refuses_the_scheduler_knobs(monkeypatch):
  from boxlcd_amd import _lib
  if _lib.features() & _lib.FEATURE_SCHED:
    pytest.skip('scheduler build')
  env, poses, sel = parity.make_batch('Dropbox', 256, 1)
  for k, v in (('BLCD_ASYNC', '3'), ('BLCD_WAVE_BATCH', '4'), ('BLCD_YIELD_PASSES', '2')):
    monkeypatch.setenv(k, v)
    with pytest.raises(RuntimeError, match='BLCD_SCHED'):
      Handle(env.scene.desc, 256, 0)
    monkeypatch.delenv(k)
  monkeypatch.setenv('BLCD_YIELD_PASSES', '1')       # "off" is accepted
  Handle(env.scene.desc, 256, 0).close()


@pytest.mark.parametrize('name,n', [('Dropbox', 5000), ('Object2', 5000), ('Object3', 3000), ('Bounce2', 3000)])
def test_suspending_straggler_environments_changes_nothing(monkeypatch, name, n):
  """Environment-level scheduling of fused chunks: a lane whose joint-free island has not converged after 24 velocity sweeps may
  suspend its environment; a later pass of the same chunk resumes it at that sweep.  Placement in time only: frames,
  observations and the full state equal the single-pass run, whatever the policy (never / at most 32 lanes / always / at most 8 lanes), and
  equal the oracle on a sample."""
  _needs_sched_build()
  T = 60
  env, poses, sel = parity.make_batch(name, n, 21)
  d = env.scene.desc
  res = []
  for knobs in ({'BLCD_YIELD_PASSES': '1'}, {'BLCD_YIELD_PASSES': '2'}, {'BLCD_YIELD_PASSES': '5', 'BLCD_YIELD_LANES': '64'}, {'BLCD_YIELD_PASSES': '2', 'BLCD_YIELD_LANES': '8'}):
    for k in ('BLCD_YIELD_PASSES', 'BLCD_YIELD_LANES'):
      monkeypatch.delenv(k, raising=False)
    for k, v in knobs.items():
      monkeypatch.setenv(k, v)
    h = Handle(d, n, 0)
    h.reset(None, poses, sel)
    lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
    obs = np.zeros((T, n, d.n_obs), np.float32)
    h.rollout(None, T, lcd, obs)
    res.append((lcd, obs, [x.copy() for x in h.debug_dump()]))
    assert not h.faults().any()
    h.close()
  for b in res[1:]:
    assert (res[0][0] == b[0]).all() and (res[0][1] == b[1]).all()
    for x, y in zip(res[0][2], b[2]):
      assert (x == y).all()
  idx = np.random.RandomState(2).choice(n, 64, replace=False)
  _, _, olcd, ost = pyb2o.rollout(d, poses[idx], sel[idx], None, T, threads=8)
  assert (ost == res[1][2][0][idx]).all() and (olcd == res[1][0][-1][idx]).all()


@pytest.mark.parametrize('name,n', [('Urchin', 2000), ('LuxoBall', 2000), ('UrchinBalls', 700), ('Object2', 4000), ('Dropbox', 4000), ('Luxo', 333)])
def test_asynchronous_rollouts_change_nothing(monkeypatch, name, n):
  """BLCD_ASYNC=<k>: no chunk boundaries - every launch advances every unfinished environment by at most k world steps from its own
  position; an environment suspends at velocity sweep 24 of a joint-free island, at position iteration 12 of a staged island or at
  its first TOI event when few lanes of its wave are in the same situation, and pays what it owes in a later launch, sorted next
  to its like.  Scheduling only: frames, observations and the full state equal the plain rollout and the oracle."""
  _needs_sched_build()
  T = 40
  env, poses, sel = parity.make_batch(name, n, 33)
  acts = env.sample_actions(T)
  d = env.scene.desc
  res = []
  for knobs in ({}, {'BLCD_ASYNC': '1'}, {'BLCD_ASYNC': '3', 'BLCD_YIELD_LANES': '64'}, {'BLCD_ASYNC': '7', 'BLCD_YIELD_LANES': '8'},
                {'BLCD_WAVE_BATCH': '12'}, {'BLCD_WAVE_BATCH': '3', 'BLCD_YIELD_LANES': '40'}):   # in-wave batching: resumption inside the launch
    for k in ('BLCD_ASYNC', 'BLCD_YIELD_LANES', 'BLCD_WAVE_BATCH'):
      monkeypatch.delenv(k, raising=False)
    for k, v in knobs.items():
      monkeypatch.setenv(k, v)
    h = Handle(d, n, 0)
    h.reset(None, poses, sel)
    lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
    obs = np.zeros((T, n, d.n_obs), np.float32)
    h.rollout(acts, T, lcd, obs)
    h.step(acts[0], 1)                                   # the plain path right after an asynchronous rollout
    res.append((lcd, obs, [x.copy() for x in h.debug_dump()]))
    assert not h.faults().any()
    if 'BLCD_ASYNC' in knobs:
      s = h.sched_stats()
      assert s['later_suspended'] > 0, s                 # the policy did suspend environments
    h.close()
  for b in res[1:]:
    assert (res[0][0] == b[0]).all() and (res[0][1] == b[1]).all()
    for x, y in zip(res[0][2], b[2]):
      assert (x == y).all()
  idx = np.random.RandomState(2).choice(n, 48, replace=False)
  _, _, olcd, ost = pyb2o.rollout(d, poses[idx], sel[idx], acts[:, idx], T, threads=8)
  assert (olcd == res[1][0][-1][idx]).all()


@pytest.mark.parametrize('name,n', [('Bounce', 70_000), ('Dropbox', 70_000), ('Object2', 90_000)])
def test_two_wave_widths_change_nothing(monkeypatch, name, n):
  """Re-binned batches: once the environments that are not asleep no longer fill the SIMDs they are stepped in narrower waves (the
  sleeping tail keeps full waves) - a launch then holds waves of two widths.  Placement only: frames, observations and the final
  state equal the one-width run."""
  T = 100
  env, poses, sel = parity.make_batch(name, n, 13)
  d = env.scene.desc
  res = []
  for tw in ('0', '16', '8', '5'):   # 5: widths that are not a divisor of anything
    monkeypatch.setenv('BLCD_TWO_WIDTHS', tw)
    monkeypatch.setenv('BLCD_UNSORTED_SPREAD', '0' if tw == '16' else '1')
    h = Handle(d, n, 0)
    h.reset(None, poses, sel)
    lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
    obs = np.zeros((T, n, d.n_obs), np.float32)
    h.rollout(None, T, lcd, obs)
    h.rollout(None, 20)
    res.append((lcd, obs, [x.copy() for x in h.debug_dump()]))
    assert not h.faults().any()
    h.close()
  for b in res[1:]:
    assert (res[0][0] == b[0]).all() and (res[0][1] == b[1]).all()
    for x, y in zip(res[0][2], b[2]):
      assert (x == y).all()


def _random_convex(rng, k, r0, r1):
  """k points on a jittered circle, CCW: a convex polygon b2PolygonShape::Set keeps as it is (up to its own vertex order)"""
  ang = np.sort(rng.uniform(0, 2 * np.pi, k))
  while np.diff(np.concatenate([ang, [ang[0] + 2 * np.pi]])).min() < 0.25:
    ang = np.sort(rng.uniform(0, 2 * np.pi, k))
  r = rng.uniform(r0, r1)
  return np.stack([r * np.cos(ang), r * np.sin(ang)], 1)


def _shape_spec(rng, kind):
  sp = np.zeros(34, np.float32)
  if kind == 'circle':
    sp[:2] = [0, rng.uniform(0.2, 0.8)]
    reach = sp[1]
  elif kind == 'box':
    sp[:3] = [1, rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8)]
    reach = float(np.hypot(sp[1], sp[2]))
  else:
    k = int(kind[4:])
    v = _random_convex(rng, k, 0.3, 0.8)
    sp[0], sp[1] = 3, k
    sp[2:2 + 2 * k] = v.reshape(-1)
    reach = float(np.abs(v).max())
  return sp, reach


@pytest.mark.parametrize('pair', ['circle-circle', 'box-circle', 'poly5-circle', 'box-box', 'poly6-box', 'poly8-poly5', 'wall-circle', 'wall-box',
                                  'wall-poly7', 'edge-circle', 'edge-poly6'])
def test_device_narrow_phase_equals_the_oracle_on_random_configurations(pair):
  """The product's narrow phase (wall-specialised edge routines with host-built per-wall constants, rewritten body-body routines,
  its own world-manifold code) against the oracle's upstream-shaped generic routines, manifold by manifold and bit for bit, on
  8 000 random near-contact configurations per shape pair - vertex regions, polygon reference faces, clipped and dropped points,
  deep overlaps and near misses that a rollout visits rarely.  (The oracle's routines are themselves checked against geometry in
  tests/test_oracle_narrowphase.py.)  'wall' = one of the arena's four edges, 'edge' = an arbitrary segment; both at pose 0."""
  from boxlcd_amd import _lib
  import ctypes as C
  n = 8000
  import zlib
  rng = np.random.RandomState(zlib.crc32(pair.encode()))
  ka, kb = pair.split('-')
  SA, PA, SB, PB = np.zeros((n, 34), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 34), np.float32), np.zeros((n, 3), np.float32)
  for i in range(n):
    sb, rb = _shape_spec(rng, kb)
    pb = np.array([rng.uniform(1, 4), rng.uniform(1, 4), rng.uniform(-3.2, 3.2)])
    if ka in ('wall', 'edge'):
      W, H = rng.choice([5.0, 7.5, 10.0]), 5.0
      if ka == 'wall':
        a, b = [((0, 0), (W, 0)), ((0, 0), (0, H)), ((W, 0), (W, H)), ((0, H), (W, H))][rng.randint(4)]
      else:
        a = rng.uniform(0, 5, 2)
        th = rng.uniform(0, 2 * np.pi)
        b = a + rng.uniform(1.0, 6.0) * np.array([np.cos(th), np.sin(th)])
      a, b = np.asarray(a, float), np.asarray(b, float)
      SA[i, :5] = [2, a[0], a[1], b[0], b[1]]
      t = rng.uniform(-0.15, 1.15)                       # along the segment, a little beyond both ends (vertex regions)
      e = (b - a) / np.linalg.norm(b - a)
      nrm = np.array([-e[1], e[0]]) * rng.choice([-1.0, 1.0])
      inner = min(sb[1], sb[2]) if kb == 'box' else (sb[1] if kb == 'circle' else 0.3)
      dist = rng.uniform(0.5 * inner, rb + 0.03) if rng.rand() < 0.8 else rng.uniform(-0.05, rb + 0.08)
      pb[:2] = a + t * (b - a) + dist * nrm
    else:
      sa, ra = _shape_spec(rng, ka)
      SA[i] = sa
      PA[i] = [rng.uniform(1, 4), rng.uniform(1, 4), rng.uniform(-3.2, 3.2)]
      th = rng.uniform(0, 2 * np.pi)
      dist = rng.uniform(0.4, 1.0) * (ra + rb) + rng.uniform(-0.03, 0.05)
      pb[:2] = PA[i, :2] + dist * np.array([np.cos(th), np.sin(th)])
    SB[i], PB[i] = sb, pb
  out = np.zeros((n, 24), np.float32)
  A18, B18 = np.ascontiguousarray(SA[:, :18]), np.ascontiguousarray(SB[:, :18])     # BLCD_COLLIDE_SPEC_FLOATS (polygons of up to 8 vertices)
  _lib._check(_lib.load().blcd_debug_collide(0, n, _lib._ptr(A18), _lib._ptr(PA), _lib._ptr(B18), _lib._ptr(PB), _lib._ptr(out)))
  olib = pyb2o.load()
  olib.b2o_collide.restype = C.c_int32
  ref = np.zeros((n, 24), np.float32)
  for i in range(n):
    assert olib.b2o_collide(pyb2o._p(SA[i]), pyb2o._p(PA[i]), pyb2o._p(SB[i]), pyb2o._p(PB[i]), pyb2o._p(ref[i])) >= 0
  ref[:, [16, 19]] = 0.0                                 # the separations: the product's world-manifold code does not compute them
  touching = ref[:, 0] > 0
  ref[~touching, 2:6] = 0.0                              # without a contact the routines leave localNormal / localPoint as they found them
  out[~touching, 2:6] = 0.0                              # (the oracle's hook hands them an uninitialised manifold)
  assert 0.25 < touching.mean() < 0.98, touching.mean()  # the sample straddles the contact boundary
  if kb != 'circle':
    assert (ref[:, 0] == 2).mean() > 0.1 and (ref[:, 0] == 1).mean() > 0.01
    assert (ref[touching, 1] == 2).mean() > 0.02         # polygon reference faces (e_faceB) occur
  bad = np.nonzero((out.view(np.uint32) != ref.view(np.uint32)).any(1))[0]
  # the manifold keys are floats holding integers; compare bit patterns so that signs of zeros count too
  assert len(bad) == 0, (pair, len(bad), bad[:5], out[bad[:2]], ref[bad[:2]])
