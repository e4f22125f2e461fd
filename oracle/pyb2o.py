"""ORACLE — TEST INFRASTRUCTURE ONLY (ctypes wrapper of oracle/_build/libb2oracle.so).

Importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Never boxlcd_amd/.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# B2O_LIB: an alternative build of the same sources (tools/replay_gifs.py --builds: FMA-contracting experiments)
LIB_PATH = os.environ.get('B2O_LIB') or os.path.join(_HERE, '_build', 'libb2oracle.so')
BODY_F, JOINT_F, PAIR_F = 12, 5, 18
_lib = None
_ellipse_rgb_lut = None


def build():
  subprocess.check_call(['make', '-s', '-C', _HERE])


def load():
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      build()
    _lib = C.CDLL(LIB_PATH)
    _lib.b2o_create.restype = C.c_void_p
    _lib.b2o_rollout.restype = C.c_double
    _lib.b2o_rollout_frames.restype = C.c_double
    _lib.b2o_num_pairs.restype = C.c_int32
    _lib.b2o_contact_order.restype = C.c_int32
    global _ellipse_rgb_lut
    _ellipse_rgb_lut = np.fromfile(os.path.join(_HERE, 'ellipse_rgb_lut.bin'), np.uint8)   # Pillow span table (DATA)
    amax = 0
    while (amax + 1) * 5 * (amax + 3) * 6 < _ellipse_rgb_lut.size:
      amax += 1
    assert (amax + 1) * 5 * (amax + 3) * 6 == _ellipse_rgb_lut.size
    _lib.b2o_set_ellipse_rgb_lut(_p(_ellipse_rgb_lut), amax)
  return _lib


def _p(x):
  return None if x is None else x.ctypes.data_as(C.c_void_p)


class OracleEnv:
  """One CPU-oracle environment built from a compiled scene descriptor (same struct layout as blcd_scene_desc)."""

  def __init__(self, desc):
    self.lib = load()
    self.desc = desc
    self.nb, self.nj, self.n_obs, self.n_act = desc.n_bodies, desc.n_joints, desc.n_obs, desc.n_act
    self.h, self.w = desc.lcd_h, desc.lcd_w
    self._e = C.c_void_p(self.lib.b2o_create(C.byref(desc)))
    self.n_pairs = 0

  def __del__(self):
    try:
      self.lib.b2o_destroy(self._e)
    except Exception:
      pass

  def reset(self, poses, shape_sel=None):
    poses = np.ascontiguousarray(poses, np.float32).reshape(self.nb, 3)
    sel = None if shape_sel is None else np.ascontiguousarray(shape_sel, np.int32).reshape(self.nb)
    self.lib.b2o_reset(self._e, _p(poses), _p(sel))
    self.n_pairs = self.lib.b2o_num_pairs(self._e)

  def set_poses(self, poses, mask=None):
    poses = np.ascontiguousarray(poses, np.float32).reshape(self.nb, 3)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    self.lib.b2o_set_poses(self._e, _p(poses), _p(m))

  def step(self, action=None):
    a = None if action is None else np.ascontiguousarray(action, np.float32).reshape(self.n_act)
    self.lib.b2o_env_step(self._e, _p(a))

  def world_step(self):
    self.lib.b2o_world_step(self._e)

  def set_motor_speeds(self, action):
    a = np.ascontiguousarray(action, np.float32).reshape(self.n_act)
    self.lib.b2o_set_motor_speeds(self._e, _p(a))

  def obs(self):
    o = np.zeros(self.n_obs, np.float64)
    self.lib.b2o_get_obs(self._e, _p(o))
    return o

  def render(self):
    img = np.zeros((self.h, self.w), np.uint8)
    self.lib.b2o_render(self._e, _p(img))
    return img

  def render_ex(self, width, height, mode='1'):
    """lcd_render(width, height, lcd_mode): mode '1' -> uint8 [H, W]; 'RGB' -> uint8 [H, W, 3]"""
    rgb = mode.upper() == 'RGB'
    img = np.zeros((height, width, 3) if rgb else (height, width), np.uint8)
    rc = self.lib.b2o_render_ex(self._e, int(width), int(height), int(rgb), _p(img))
    if rc != 0:
      raise ValueError('ellipse bbox outside the span table')
    return img

  def pair_table(self):
    t = np.zeros((self.n_pairs, 2), np.int32)
    self.lib.b2o_pair_table(self._e, _p(t))
    return t

  def dump(self):
    b = np.zeros((self.nb, BODY_F), np.float32)
    j = np.zeros((max(self.nj, 1), JOINT_F), np.float32)
    p = np.zeros((max(self.n_pairs, 1), PAIR_F), np.float32)
    self.lib.b2o_dump(self._e, _p(b), _p(j), _p(p))
    return b, j[:self.nj], p[:self.n_pairs]

  def nudge(self, body, field, ulps):
    self.lib.b2o_nudge(self._e, int(body), int(field), int(ulps))

  def body_xf(self):
    """(xf [nb,4] = p.x,p.y,q.s,q.c ; shapes list of ('circle', r) | ('poly', world verts [k,2])) as lcd_render reads them."""
    xf = np.zeros((self.nb, 4), np.float32)
    vs = np.zeros((self.nb, 33), np.float32)
    self.lib.b2o_body_xf(self._e, _p(xf), _p(vs))
    shapes = []
    for i in range(self.nb):
      k = int(vs[i, 0])
      shapes.append(('circle', float(vs[i, 1])) if k == 0 else ('poly', vs[i, 1:1 + 2 * k].reshape(k, 2).copy()))
    return xf, shapes

  def stats(self):
    s = np.zeros(6, np.int64)
    self.lib.b2o_stats(self._e, _p(s))
    return dict(zip(['steps', 'toi_events', 'toi_calls', 'islands', 'contacts_created', 'contacts_destroyed'], s.tolist()))

  def contact_order(self):
    o = np.zeros(64, np.int32)
    n = self.lib.b2o_contact_order(self._e, _p(o), 64)
    return o[:n]


VARIANT_KEYS = {'sincos': 0, 'damping': 1, 'advance': 2, 'polygons': 3, 'massref': 4}
VARIANT_DEFAULTS = {'sincos': 2, 'damping': 1, 'advance': 1, 'polygons': 1, 'massref': 0}


class variants:
  """with pyb2o.variants(damping=0, ...): ...  — temporarily switch oracle variants (see b2o_math.h); restores defaults."""

  def __init__(self, **kw):
    self.kw = kw

  def __enter__(self):
    lib = load()
    for k, v in self.kw.items():
      lib.b2o_set_variant(VARIANT_KEYS[k], int(v))
    return self

  def __exit__(self, *a):
    lib = load()
    for k, v in VARIANT_DEFAULTS.items():
      lib.b2o_set_variant(VARIANT_KEYS[k], int(v))


def render_poses(desc, poses, shape_sel=None):
  lib = load()
  poses = np.ascontiguousarray(poses, np.float32)
  n = poses.shape[0]
  sel = None if shape_sel is None else np.ascontiguousarray(shape_sel, np.int32)
  img = np.zeros((n, desc.lcd_h, desc.lcd_w), np.uint8)
  lib.b2o_render_poses(C.byref(desc), _p(poses), _p(sel), n, _p(img))
  return img


def render_poses_ex(desc, poses, shape_sel, width, height, mode='1'):
  lib = load()
  poses = np.ascontiguousarray(poses, np.float32)
  n = poses.shape[0]
  sel = None if shape_sel is None else np.ascontiguousarray(shape_sel, np.int32)
  rgb = mode.upper() == 'RGB'
  img = np.zeros((n, height, width, 3) if rgb else (n, height, width), np.uint8)
  if lib.b2o_render_poses_ex(C.byref(desc), _p(poses), _p(sel), n, int(width), int(height), int(rgb), _p(img)) != 0:
    raise ValueError('ellipse bbox outside the span table')
  return img


def raster_polygon(xy, w, h, variant):
  lib = load()
  xy = np.ascontiguousarray(xy, np.int32)
  img = np.ones((h, w), np.uint8)
  lib.b2o_raster_polygon(_p(xy), len(xy) // 2 if xy.ndim == 1 else xy.shape[0], w, h, variant, _p(img))
  return img


def raster_ellipse(x0, y0, x1, y1, w, h):
  lib = load()
  img = np.ones((h, w), np.uint8)
  lib.b2o_raster_ellipse(int(x0), int(y0), int(x1), int(y1), w, h, _p(img))
  return img


def sincos(x):
  lib = load()
  x = np.ascontiguousarray(x, np.float32)
  s, c = np.zeros_like(x), np.zeros_like(x)
  lib.b2o_sincos(_p(x), C.c_int64(x.size), _p(s), _p(c))
  return s, c


def mass_data(desc, shape, density):
  lib = load()
  out = np.zeros(24, np.float32)
  lib.b2o_mass_data(C.byref(desc), int(shape), C.c_float(density), _p(out))
  return out


def rollout(desc, poses, shape_sel, actions, T, threads=1, want_obs=True, want_lcd=True, want_state=True, render_every_step=False):
  """Returns (seconds, obs f32 [n,obs], lcd u8 [n,h,w], state f32 [n,nb,12])."""
  lib = load()
  poses = np.ascontiguousarray(poses, np.float32)
  n = poses.shape[0]
  sel = None if shape_sel is None else np.ascontiguousarray(shape_sel, np.int32)
  act = None if actions is None else np.ascontiguousarray(actions, np.float32)
  obs = np.zeros((n, desc.n_obs), np.float32) if want_obs else None
  lcd = np.zeros((n, desc.lcd_h, desc.lcd_w), np.uint8) if want_lcd else None
  st = np.zeros((n, desc.n_bodies, BODY_F), np.float32) if want_state else None
  sec = lib.b2o_rollout(C.byref(desc), n, int(T), int(threads), _p(poses), _p(sel), _p(act), _p(obs), _p(lcd), _p(st),
                        int(bool(render_every_step)))
  return sec, obs, lcd, st


def rollout_frames(desc, poses, shape_sel, actions, T, threads=1):
  """Every env-step's outputs: (seconds, obs f32 [T,n,obs], lcd u8 [T,n,h,w], final state f32 [n,nb,12])."""
  lib = load()
  poses = np.ascontiguousarray(poses, np.float32)
  n = poses.shape[0]
  sel = None if shape_sel is None else np.ascontiguousarray(shape_sel, np.int32)
  act = None if actions is None else np.ascontiguousarray(actions, np.float32)
  obs = np.zeros((T, n, desc.n_obs), np.float32)
  lcd = np.zeros((T, n, desc.lcd_h, desc.lcd_w), np.uint8)
  st = np.zeros((n, desc.n_bodies, BODY_F), np.float32)
  sec = lib.b2o_rollout_frames(C.byref(desc), n, int(T), int(threads), _p(poses), _p(sel), _p(act), _p(obs), _p(lcd), _p(st))
  return sec, obs, lcd, st
