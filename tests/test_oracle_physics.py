"""CPU oracle pinned against analytic known answers and libm (the reference's recordings: tests/test_oracle_replay.py)."""
import numpy as np
import pytest
import boxlcd_amd as B


def _env(oracle, name, pose, variant=0):
  env = getattr(B.envs, name)(raster_variant=variant)
  o = oracle.OracleEnv(env.scene.desc)
  o.reset(np.asarray(pose, np.float32))
  return env, o


def test_free_fall_increments(oracle):
  """Semi-implicit Euler, 3 sub-steps of 1/30 s: env-step k moves by -(g/900)(9k-3)."""
  _, o = _env(oracle, 'Dropbox', [[2.5, 3.6, 0.3]])
  y = [3.6]
  for k in range(1, 5):
    o.step(np.zeros(1, np.float32))
    y.append(float(o.dump()[0][0, 1]))
    assert abs((y[-1] - y[-2]) + (9.81 / 900.0) * (9 * k - 3)) < 2e-6


def test_box_rest_height_and_sleep(oracle):
  _, o = _env(oracle, 'Dropbox', [[2.5, 1.5, 0.0]])
  for _ in range(40):
    o.step(np.zeros(1, np.float32))
  b = o.dump()[0][0]
  assert abs(b[1] - (0.7 + 0.015)) < 2e-3          # half size + 2*polygonRadius - linearSlop
  assert b[7] == 0.0 and b[3] == 0.0 and b[4] == 0.0  # asleep, velocities zeroed
  assert o.render()[11:16, :].min() == 0


def test_bounce_apex_ratio(oracle):
  """restitution mixes as max(0.8, 0) -> apex heights above rest shrink by 0.8^2 per bounce."""
  _, o = _env(oracle, 'Bounce', [[2.5, 4.0, 0.0]])
  ys = []
  for _ in range(120):
    o.step(np.zeros(1, np.float32))
    ys.append(float(o.dump()[0][0, 1]))
  ys = np.array(ys)
  apex = [ys[i] for i in range(1, len(ys) - 1) if ys[i] >= ys[i - 1] and ys[i] > ys[i + 1] and ys[i] > 0.8]
  rest = 0.5 + 0.01
  ratios = [(apex[i + 1] - rest) / (apex[i] - rest) for i in range(min(2, len(apex) - 1))]
  assert len(ratios) >= 1 and all(abs(r - 0.64) < 0.08 for r in ratios), (apex, ratios)


def test_urchin_limits_and_motors(oracle):
  env = B.envs.Urchin()
  benv = B.BatchedWorldEnv('Urchin', 1, seed=3)
  poses, sel = benv.sample_initial(1)
  o = oracle.OracleEnv(env.scene.desc)
  o.reset(poses[0], sel[0])
  for t in range(60):
    o.step(np.array([1.0, -1.0, 1.0], np.float32))
  b, j, p = o.dump()
  assert np.isfinite(b).all() and np.isfinite(j).all()
  assert (b[:, 7] == 1.0).all()                      # SetMotorSpeed keeps the robot awake
  # joint angle within limits +- slack: aB - aA - ref in [-1, 1]
  ref = poses[0, 1:, 2] - poses[0, 0, 2]
  ang = b[1:, 2] - b[0, 2] - ref
  assert (ang > -1.5).all() and (ang < 1.5).all(), ang   # soft limits under a 150 N m motor
  assert set(j[:, 4].astype(int).tolist()) <= {0, 1, 2}


def test_mass_data_known_answers(oracle):
  env = B.envs.Dropbox()
  m = oracle.mass_data(env.scene.desc, 0, 0.1)
  assert abs(m[0] - 0.1 * 1.4 * 1.4) < 1e-6          # box mass = rho * (2h)^2
  assert abs(m[3] - m[0] * (1.4**2 + 1.4**2) / 12.0) < 1e-6
  env = B.envs.Bounce()
  m = oracle.mass_data(env.scene.desc, 0, 0.1)
  assert abs(m[0] - 0.1 * np.pi * 0.25) < 1e-6 and abs(m[3] - m[0] * 0.125) < 1e-6
  env = B.envs.Luxo()
  m = oracle.mass_data(env.scene.desc, 0, 0.1)        # luxo root: Set() hull is CCW from the right-most lowest vertex
  n = int(m[4]); v = m[5:5 + 2 * n].reshape(n, 2)
  assert n == 4 and v[0, 0] == v[:, 0].max() and v[0, 1] == v[v[:, 0] == v[:, 0].max(), 1].min()
  area2 = sum(v[i, 0] * v[(i + 1) % n, 1] - v[(i + 1) % n, 0] * v[i, 1] for i in range(n))
  assert area2 > 0 and abs(m[0] - 0.1 * area2 / 2) < 1e-6 and abs(m[1]) > 1e-3   # non-zero centroid


def test_sincos_variants(oracle):
  """Default sincos = glibc <= 2.27's algorithm (the libm the exactly-reproduced recordings were made with): double-precision
  Chebyshev polynomials, correctly rounded on all but ~1e-7 of inputs.  Variant 0 = glibc >= 2.28's algorithm, checked against
  this container's glibc (it differs only through glibc's FMA ifunc build, ~1e-8 of inputs)."""
  rng = np.random.RandomState(0)
  x = np.concatenate([rng.uniform(-130, 130, 2_000_000), rng.uniform(-1e-3, 1e-3, 1000), rng.uniform(-0.04, 0.04, 100000),
                      [0.0, -0.0, 0.75, 119.99, 120.0, 1e5, -8e6, 0.7853981, 0.7853982, 7.0685835, 7.068584]]).astype(np.float32)
  s, c = oracle.sincos(x)
  rs, rc = np.sin(x.astype(np.float64)).astype(np.float32), np.cos(x.astype(np.float64)).astype(np.float32)
  assert (s != rs).sum() <= 3 and (c != rc).sum() <= 3, ((s != rs).sum(), (c != rc).sum())
  assert np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64))).max() < 6e-8
  with oracle.variants(sincos=0):
    s, c = oracle.sincos(x)
  import ctypes, ctypes.util
  libm = ctypes.CDLL(ctypes.util.find_library('m'))
  libm.sinf.restype = ctypes.c_float; libm.sinf.argtypes = [ctypes.c_float]
  libm.cosf.restype = ctypes.c_float; libm.cosf.argtypes = [ctypes.c_float]
  idx = rng.randint(0, len(x), 20000).tolist() + list(range(len(x) - 1011, len(x)))
  bad = sum((libm.sinf(float(x[i])) != s[i]) or (libm.cosf(float(x[i])) != c[i]) for i in idx)
  assert bad <= 1
  assert np.abs(s - np.sin(x.astype(np.float64))).max() < 6e-8 and np.abs(c - np.cos(x.astype(np.float64))).max() < 6e-8


def test_oracle_is_deterministic_and_threaded_rollout_matches(oracle):
  benv = B.BatchedWorldEnv('Object2', 24, seed=5)
  poses, sel = benv.sample_initial(24)
  d = benv.scene.desc
  _, o1, l1, s1 = oracle.rollout(d, poses, sel, None, 30, threads=1)
  _, o2, l2, s2 = oracle.rollout(d, poses, sel, None, 30, threads=4)
  assert (o1 == o2).all() and (l1 == l2).all() and (s1 == s2).all()


@pytest.mark.parametrize('name', ['Bounce', 'Dropbox', 'Object2', 'Urchin', 'LuxoBall', 'UrchinCube'])
def test_velocities_never_carry_a_negative_zero(oracle, name):
  """The product folds the wall side out of the contact sweeps (RegIsland::warmStartContactT): X - (+0) - (+-0) == X bit for bit
  unless X is -0, and X = v + t is -0 only if a body velocity is.  Every write of v / w in the step is a sum or difference
  with the old value, a product with a positive factor or +0, so -0 cannot arise; checked here after every world step."""
  n, T = 12, 80
  benv = B.BatchedWorldEnv(name, n, seed=11)
  poses, sel = benv.sample_initial(n)
  acts = benv.sample_actions(T)
  for e in range(n):
    o = oracle.OracleEnv(benv.scene.desc)
    o.reset(poses[e], sel[e])
    for t in range(T):
      o.set_motor_speeds(acts[t, e])
      for _ in range(3):
        o.world_step()
        vel = o.dump()[0][:, 3:6]
        assert not (np.signbit(vel) & (vel == 0.0)).any(), (name, e, t)
