"""`WorldEnv` (single gym-style env) and `BatchedWorldEnv` (N envs, AsyncVectorEnv call shapes) over the HIP C-ABI.

Mirrors the reference's `boxLCD/world_env.py` API: constructor + config (:47-61), obs/action specification
(:69-142), properties (:144-166), `seed` (:168-170), `reset(full_state=None, proprio=None)` (:306-385),
`step(action)` (:431-458), `lcd_render` (:460-512), `render` (:514-535), `close` (:185-188).
The physics (`b2World.Step`), `_get_obs` and the PIL raster run on the GPU in libboxlcd_hip.so; this file keeps the
host-side glue the reference has in Python: the float64 numpy sampling of initial poses and the
normalised-state -> pose conversion of `reset(full_state=)`.

There is no CPU fallback: constructing an env only needs the host tables, but `reset/step/render` raise
`RuntimeError` if the HIP library or a GPU is missing.
"""
import hashlib
import os
import struct
import numpy as np
from . import utils
from . import spaces as _spaces
from .scene import compile_scene, fill_robots, KIND_OBJECT, KIND_ROOT, KIND_LINK

A = utils.A


# --- gym 0.17.3 `gym.utils.seeding.np_random` restated (reference world_env.py:169 uses it) -----------------------
def _bigint_from_bytes(b):
  sizeof_int = 4
  padding = sizeof_int - len(b) % sizeof_int
  b += b'\0' * padding
  int_count = int(len(b) / sizeof_int)
  unpacked = struct.unpack('{}I'.format(int_count), b)
  accum = 0
  for i, val in enumerate(unpacked):
    accum += 2**(sizeof_int * 8 * i) * val
  return accum


def np_random(seed=None):
  if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
    raise ValueError('Seed must be a non-negative integer or omitted, not {}'.format(seed))
  if seed is None:
    seed = _bigint_from_bytes(os.urandom(8))
  seed = int(seed) % 2**(8 * 8)
  h = hashlib.sha512(str(seed).encode('utf8')).digest()
  big = _bigint_from_bytes(h[:8])
  ints = []
  if big == 0:
    ints = [0]
  while big > 0:
    big, mod = divmod(big, 2**32)
    ints.append(mod)
  rng = np.random.RandomState()
  rng.seed(ints)
  return rng, seed


def _f32(x):
  return np.asarray(x, dtype=np.float64).astype(np.float32)


class _EnvSpec:
  """Everything about an env class+config that does not need the GPU (shared by WorldEnv and BatchedWorldEnv)."""

  ENV_DG = utils.AttrDict()
  ENV_DG.base_dim = 5
  ENV_DG.lcd_base = 16
  ENV_DG.wh_ratio = 2.0
  ENV_DG.ep_len = 100
  ENV_DG.angular_offset = 0
  ENV_DG.root_offset = 0
  ENV_DG.compact_obs = 0
  ENV_DG.use_speed = 1
  ENV_DG.all_corners = 0
  ENV_DG.walls = 1
  ENV_DG.debug = 0
  ENV_DG.fps = 10

  def _init_spec(self, world_def, G, raster_variant=None):
    self.G = utils.AttrDict(self.ENV_DG)
    if not isinstance(G, dict):
      G = G.__dict__
    for key in G:
      self.G[key] = G[key]
    for key, want in (('walls', 1), ('use_speed', 1), ('all_corners', 0), ('root_offset', 0), ('angular_offset', 0), ('compact_obs', 0)):
      if self.G[key] != want:
        # the reference's non-default branches are broken; the defaults are the contract
        raise NotImplementedError(f'G.{key}={self.G[key]} is outside the supported contract (default {want})')
    if raster_variant is None:
      # Polygon scan-conversion rule (Pillow changed it between releases; they differ only on sub-pixel-thin links):
      #   2 = the Pillow that rendered the reference's published recordings - pinned frame for frame by eight of them,
      #       incl. 150 frames of thin-limbed UrchinBall and the one-pixel-high polygons of the Luxo recordings (DEFAULT: the
      #       only variant with evidence from the reference itself; by its behaviour the 9.0.x line the reference pins)
      #   1 = Pillow 12.2 (pinned by tests/golden/pillow_*.npz)      0 = variant 1 without corner joining (no fixture of its own)
      raster_variant = int(os.environ.get('BOXLCD_RASTER_VARIANT', '2'))
    self.raster_variant = raster_variant
    self.world_def = fill_robots(world_def, self.G)
    self.scene = compile_scene(self.world_def, self.G, self.WIDTH, self.HEIGHT, raster_variant)
    self.obs_info = self.scene.obs_info
    self.act_info = self.scene.act_info
    self.obs_size = len(self.obs_info)
    self.obs_keys = list(self.obs_info.keys())
    self.pobs_keys = utils.nfiltlist(self.obs_keys, 'object')
    self.pobs_size = len(self.pobs_keys)
    self.pobs_idxs = [self.obs_keys.index(x) for x in self.pobs_keys]
    sp = {}
    sp['full_state'] = _spaces.Box(-1, +1, (self.obs_size,), dtype=np.float32)
    sp['proprio'] = _spaces.Box(-1, +1, (self.pobs_size if self.pobs_size else 1,), dtype=np.float32)
    sp['lcd'] = _spaces.Box(0, 1, (self.G.lcd_base, int(self.G.lcd_base * self.G.wh_ratio)), dtype=np.bool_)
    self.observation_space = _spaces.Dict(sp)
    self.act_size = len(self.act_info)
    self.act_keys = list(self.act_info.keys())
    self.action_space = _spaces.Box(-1, +1, (self.act_size,), dtype=np.float32)

  @property
  def WIDTH(self):
    return int(self.G.wh_ratio * self.G.base_dim)

  @property
  def HEIGHT(self):
    return self.G.base_dim

  @property
  def VIEWPORT_H(self):
    return 30 * self.HEIGHT

  @property
  def VIEWPORT_W(self):
    return 30 * self.WIDTH

  @property
  def FPS(self):
    return self.G.fps

  @property
  def SCALE(self):
    from .world_defs import SCALE
    return SCALE

  # ---- initial-state sampling: reference world_env.py:172-175 (_sample) and :197-304 (_reset_bodies) ------------
  def _sample_poses(self, uniform, n, randint=None):
    """Vectorised over n envs.  `uniform(lo, hi)` -> float64 [n] draws IN THE REFERENCE'S ORDER; returns
    (poses float32 [n, nb, 3], shape_sel int32 [n, nb])."""
    nb = len(self.scene.bodies)
    poses = np.zeros((n, nb, 3), np.float32)
    sel = np.zeros((n, nb), np.int32)
    W, H = self.WIDTH, self.HEIGHT
    info = self.obs_info

    def S(key, lr=-1.0, ur=None):
      if ur is None:
        ur = -lr
      return utils.mapto(uniform(lr, ur), info[key])

    bodies = self.scene.bodies
    pos32 = {}
    for robot in self.world_def.robots:
      name = robot.name + ':root'
      rangex = 1 - (2 * robot.bound / W)
      rangey = 1 - (2 * robot.bound / H)
      rx = S(name + ':x:p', -rangex, rangex)
      ry = S(name + ':y:p', -rangey, -rangey)
      s_ = S(name + ':sin')
      c_ = S(name + ':cos')
      root_angle = np.arctan2(s_, c_)
      if not robot.rand_angle:
        root_angle = np.zeros(n)
      root = bodies[self.scene.body_index[name]]
      pos32[name] = np.stack([_f32(rx), _f32(ry)], -1)          # b2Vec2: float32 at the SWIG boundary
      poses[:, root.index, :2] = pos32[name]
      poses[:, root.index, 2] = _f32(root_angle)
      parent_angles = {name: root_angle}
      for jname, joint in robot.joints.items():
        lname = robot.name + ':' + jname
        pname = robot.name + ':' + joint.parent
        mangle = root_angle + joint.angle
        mangle = np.arctan2(np.sin(mangle), np.cos(mangle))
        parent_angles[lname] = mangle
        pangle = parent_angles[pname]
        ax, ay = joint.anchorA
        aa = np.stack([np.cos(pangle) * ax + -np.sin(pangle) * ay, np.sin(pangle) * ax + np.cos(pangle) * ay], -1)
        bx, by = joint.anchorB
        ab = np.stack([np.cos(mangle) * bx + -np.sin(mangle) * by, np.sin(mangle) * bx + np.cos(mangle) * by], -1)
        # `b2Vec2 + ndarray - ndarray` runs in pybox2d's float32 b2Vec2 arithmetic (world_env.py:250) [upstream]
        p = (pos32[pname] + _f32(aa)).astype(np.float32) - _f32(ab)
        pos32[lname] = p.astype(np.float32)
        link = bodies[self.scene.body_index[lname]]
        poses[:, link.index, :2] = pos32[lname]
        poses[:, link.index, 2] = _f32(mangle)
    for obj in self.world_def.objects:
      ob = bodies[self.scene.body_index[obj.name]]
      if obj.shape == 'random':
        sel[:, ob.index] = randint(2, n) if randint is not None else np.random.randint(2, size=n)
      rangex = 1 - (2 * obj.size / W)
      rangey = 1 - (2 * obj.size / H)
      x = S(obj.name + ':x:p', -rangex, rangex)
      if len(self.world_def.robots) == 0:
        y = S(obj.name + ':y:p', -rangey, rangey)
      else:
        y = S(obj.name + ':y:p', -rangey, -0.25)
      if obj.rand_angle:
        s_ = S(obj.name + ':sin')
        c_ = S(obj.name + ':cos')
        angle = np.arctan2(s_, c_)
      else:
        angle = np.zeros(n)
      poses[:, ob.index, 0] = _f32(x)
      poses[:, ob.index, 1] = _f32(y)
      poses[:, ob.index, 2] = _f32(angle)
    return poses, sel

  # ---- normalised full_state -> body poses: reference world_env.py:323-380 ---------------------------------
  def _state_to_poses(self, full_state):
    """full_state float64 [n, obs] -> (poses float32 [n, nb, 3]); every body is overwritten (objects, roots, links)."""
    fs = utils.NamedArray(np.asarray(full_state, dtype=np.float64), self.obs_info)
    n = fs.arr.shape[0]
    nb = len(self.scene.bodies)
    poses = np.zeros((n, nb, 3), np.float32)
    for b in self.scene.bodies:
      name = b.name
      xy = fs[f'{name}:x:p', f'{name}:y:p']
      ang = np.arctan2(fs[name + ':sin'], fs[name + ':cos'])
      poses[:, b.index, :2] = _f32(xy)
      poses[:, b.index, 2] = _f32(ang)
    return poses


class WorldEnv(_EnvSpec):
  """One environment with the reference's gym-style API, backed by a 1-env GPU handle."""
  metadata = {'render.modes': ['human', 'rgb_array']}

  def __init__(self, world_def, G={}, device=0, raster_variant=None):
    self._init_spec(world_def, G, raster_variant)
    self._device = device
    self._h = None
    self.viewer = None
    self.scroll = 0.0
    self.ep_t = 0
    self.seed()

  def seed(self, seed=None):
    self.np_random, seed = np_random(seed)
    return [seed]

  def _handle(self):
    if self._h is None:
      from ._lib import Handle
      self._h = Handle(self.scene.desc, 1, self._device)
    return self._h

  def close(self):
    if self._h is not None:
      self._h.close()
      self._h = None

  def reset(self, full_state=None, proprio=None):
    h = self._handle()
    self.ep_t = 0
    poses, sel = self._sample_poses(lambda lo, hi: np.array([self.np_random.uniform(lo, hi)]), 1)
    self._sel = sel
    h.reset(None, poses, sel)
    if proprio is not None:
      pshape = self.observation_space.spaces['proprio'].shape
      assert np.shape(proprio)[-1] == pshape[-1], f'invalid shape for proprio {np.shape(proprio)} {pshape}'
      full_state = np.zeros(self.observation_space.spaces['full_state'].shape)
      full_state[self.pobs_idxs] = proprio
    if full_state is not None:
      fs = np.array(full_state).astype(np.float64)[None]
      h.set_poses(None, self._state_to_poses(fs), None)
    return self._get_obs()

  def _get_obs(self):
    fs, lcd = self._handle().get_obs(np.float64)
    full_state = fs[0]
    proprio = full_state[self.pobs_idxs] if self.pobs_size != 0 else np.zeros(1)
    return {'full_state': full_state, 'proprio': proprio, 'lcd': lcd[0].astype(bool)}

  def step(self, action):
    self.ep_t += 1
    a = np.asarray(action, dtype=np.float32).reshape(1, self.act_size)
    self._handle().step(a, 1)
    reward = 0.0
    done = self.ep_t >= self.G.ep_len
    info = {'timeout': done}
    return self._get_obs(), reward, done, info

  def lcd_render(self, width=None, height=None, lcd_mode='1'):
    """reference world_env.py:460-512: any canvas size, lcd_mode '1' (bool [H, W]) or 'RGB' (uint8 [H, W, 3])."""
    lcd_mode = lcd_mode.upper()
    assert lcd_mode in ['1', 'RGB'], 'lcd_mode must be in one of these PIL supported modes'
    dw, dh = int(self.G.lcd_base * self.G.wh_ratio), self.G.lcd_base
    if width is None and height is None:
      width, height = dw, dh
    h = self._handle()
    if lcd_mode == '1' and (width, height) == (dw, dh):
      _, lcd = h.get_obs(None)
      return lcd[0].astype(bool)
    img = h.render_poses_ex(h.get_poses()[:, :, :3], self._sel, int(width), int(height), lcd_mode)[0]
    return img.astype(bool) if lcd_mode == '1' else img

  def render(self, mode='rgb_array', lcd_mode='1', return_pyglet_view=False):
    """reference world_env.py:514-535.  mode='human' composes the same side-by-side frame (8x RGB view | separator | LCD x8)
    the reference hands to its pyglet viewer; no window is opened here - pass return_pyglet_view=True to get the frame."""
    lcd_mode = lcd_mode.upper()
    width, height = int(self.G.lcd_base * self.G.wh_ratio), self.G.lcd_base
    lcd = self.lcd_render(width, height, lcd_mode=lcd_mode)
    if mode == 'rgb_array':
      return lcd
    if mode != 'human':
      raise ValueError(mode)
    high_res = self.lcd_render(width * 8, height * 8, lcd_mode='RGB').astype(np.uint8)
    if lcd_mode == 'RGB':
      low_res = lcd.astype(np.uint8).repeat(8, 0).repeat(8, 1)
    else:
      low_res = 255 * lcd.astype(np.uint8)[..., None].repeat(8, 0).repeat(8, 1).repeat(3, 2)
    img = np.concatenate([high_res, np.zeros_like(low_res)[:, :1], low_res], axis=1)
    return img if return_pyglet_view else lcd


class BatchedWorldEnv(_EnvSpec):
  """N independent envs of one class on one GPU, with the call shapes of the reference's vector env
  (research/wrappers/async_vector_env.py: reset(idxs, **kwargs) :131-189, step(actions) :191-242) — one kernel launch
  per step instead of one pipe round-trip per env."""

  def __init__(self, env_cls, num_envs, G={}, device=0, seed=0, raster_variant=None, env_id_base=0):
    if isinstance(env_cls, str):
      from . import envs as _envs
      env_cls = getattr(_envs, env_cls)
    proto = env_cls(G, raster_variant=raster_variant)
    self.ENV_DG = proto.ENV_DG
    self._init_spec(proto.world_def, proto.G, proto.raster_variant)
    self.env_cls = env_cls
    self.num_envs = int(num_envs)
    self.single_observation_space = self.observation_space
    self.single_action_space = self.action_space
    self._device = device
    self._h = None
    # Sharded batches: this object's environment i is environment env_id_base + i of the whole batch.  With the SAME seed on
    # every rank and env_id_base = rank * num_envs, the gathered shards are the batch one object of world x num_envs
    # environments would have sampled (tests/test_gpu_api.py::test_shards_with_one_seed_are_one_batch).
    self.env_id_base = int(env_id_base)
    self.ep_t = np.zeros(self.num_envs, np.int64)
    # Episode clocks of the device-resident loop (step_torch): while every environment is at the same step of its episode (after a full
    # reset, until a partial one) `done` is a cached constant tensor and neither clock costs a kernel or an O(N) host pass per step:
    # _ep_same = that common step count (None once the clocks differ), _ep_lag = step_torch increments not yet added to the host array
    self._ep_same = 0
    self._ep_lag = 0
    self.seed(seed)

  def seed(self, seed=0):
    self._seed = int(seed) & 0xffffffffffffffff
    self._mirror_episode = np.zeros(self.num_envs, np.int64)   # reset counts of sample_initial(), the host-side mirror (the device keeps its own)
    # action tapes of sample_actions(): one stream per shard (shards with one seed must not replay each other's actions)
    self._act_rng = np.random.Generator(np.random.Philox(key=int(seed) + 1 if self.env_id_base == 0 else [int(seed) + 1, self.env_id_base]))
    if self._h is not None:
      self._h.sample_reseed()
    return [seed]

  def _handle(self):
    if self._h is None:
      from ._lib import Handle
      self._h = Handle(self.scene.desc, self.num_envs, self._device)
      self._h.sample_set_base(self.env_id_base)
    return self._h

  def close(self):
    if self._h is not None:
      self._h.close()
      self._h = None

  # ---- initial states: a counter-based stream keyed by (seed, env id, reset count, variable) ------------------------------
  # BASELINE.md §3 ("per-env counter-based RNG, seed = env id"): an environment's start never depends on the batch size or on
  # which rank holds it.  The PRODUCT samples on the device (blcd_reset_sampled: reset() / reset_torch() never build poses on
  # the host); `mirror_poses` is the numpy restatement of the same stream and the same float64 arithmetic - the checker of
  # the device sampler and the source of explicit (poses, shape_sel) pairs for the parity tests and the bench.
  @staticmethod
  def _philox_u01(seed, env_ids, episodes, var):
    """Philox4x32-10, key = seed (lo, hi), counter = (env id, reset count, var, 0) -> float64 in [0, 1) from 53 bits"""
    M = np.uint64(0xffffffff)
    gid = np.asarray(env_ids, np.uint64)
    c0 = gid & M
    c1 = np.asarray(episodes, np.uint64) & M
    c2 = np.full_like(c0, var)
    c3 = gid >> np.uint64(32)
    k0, k1 = np.uint64(seed & 0xffffffff), np.uint64((seed >> 32) & 0xffffffff)
    for _ in range(10):
      p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
      c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ k0, p1 & M, (p0 >> np.uint64(32)) ^ c3 ^ k1, p0 & M
      k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return ((c0 >> np.uint64(5)).astype(np.float64) * 67108864.0 + (c1 >> np.uint64(6)).astype(np.float64)) * (1.0 / 9007199254740992.0)

  def mirror_poses(self, env_ids, episodes):
    """(poses float32 [n, nb, 3], shape_sel int32 [n, nb]) the device sampler gives this object's environment `env_ids[i]` at
    its `episodes[i]`-th reset."""
    env_ids, episodes = np.asarray(env_ids, np.int64) + self.env_id_base, np.asarray(episodes, np.int64)
    var = [0]

    def u01():
      v = var[0]
      var[0] += 1
      return self._philox_u01(self._seed, env_ids, episodes, v)
    return self._sample_poses(lambda lo, hi: lo + (hi - lo) * u01(), len(env_ids), randint=lambda k, m: np.where(u01() < 0.5, 0, 1).astype(np.int32))

  def sample_initial(self, n, idxs=None):
    """Host-side mirror of what `reset(idxs)` samples on the device: (poses, shape_sel) of environments idxs (default 0..n-1) at
    their next reset in the mirror's own count."""
    ids = np.arange(n, dtype=np.int64) if idxs is None else np.asarray(idxs, np.int64)
    out = self.mirror_poses(ids, self._mirror_episode[ids])
    self._mirror_episode[ids] += 1
    return out

  def sample_program(self):
    """The scene's sampling program for blcd_reset_sampled (include/boxlcd.h), in the reference's draw order
    (world_env.py:197-304) - the data twin of `_sample_poses`."""
    if getattr(self, '_prog', None) is not None:
      return self._prog
    from ._lib import SampleOp
    ops = []
    W, H, info, bodies = self.WIDTH, self.HEIGHT, self.obs_info, self.scene.bodies

    def op(kind, d=0, a=0, b=0, body=0, parent=0, f=()):
      o = SampleOp()
      o.kind, o.d, o.a, o.b, o.body, o.parent = kind, d, a, b, body, parent
      for i, x in enumerate(f):
        o.f[i] = float(x)
      ops.append(o)

    def draw(d, key, lr=-1.0, ur=None):
      ur = -lr if ur is None else ur
      lo, hi = info[key]
      op(0, d=d, f=(lr, ur, lo, hi))

    for robot in self.world_def.robots:
      name = robot.name + ':root'
      rangex, rangey = 1 - (2 * robot.bound / W), 1 - (2 * robot.bound / H)
      draw(0, name + ':x:p', -rangex, rangex)
      draw(1, name + ':y:p', -rangey, -rangey)
      draw(2, name + ':sin')
      draw(3, name + ':cos')
      op(2, d=4, a=2, b=3) if robot.rand_angle else op(3, d=4)
      root = bodies[self.scene.body_index[name]].index
      op(4, d=4, a=0, b=1, body=root)
      for jname, joint in robot.joints.items():
        link = bodies[self.scene.body_index[robot.name + ':' + jname]].index
        parent = bodies[self.scene.body_index[robot.name + ':' + joint.parent]].index
        op(5, a=root, body=link, parent=parent, f=(joint.angle, joint.anchorA[0], joint.anchorA[1], joint.anchorB[0], joint.anchorB[1]))
    for obj in self.world_def.objects:
      ob = bodies[self.scene.body_index[obj.name]].index
      if obj.shape == 'random':
        op(1, body=ob)
      rangex, rangey = 1 - (2 * obj.size / W), 1 - (2 * obj.size / H)
      draw(0, obj.name + ':x:p', -rangex, rangex)
      draw(1, obj.name + ':y:p', -rangey, rangey if len(self.world_def.robots) == 0 else -0.25)
      if obj.rand_angle:
        draw(2, obj.name + ':sin')
        draw(3, obj.name + ':cos')
        op(2, d=4, a=2, b=3)
      else:
        op(3, d=4)
      op(4, d=4, a=0, b=1, body=ob)
    self._prog = ops
    return ops

  def _reset_on_device(self, ii):
    """reset environments ii (int32 numpy) from the device sampler.  Reset counts and shape choices live on the device only
    (the render calls read the shapes back with blcd_get_shape_sel; snapshots carry the counts)."""
    h = self._handle()
    full = len(ii) == self.num_envs and (ii == np.arange(self.num_envs)).all()
    h.reset_sampled(None if full else ii, self._seed, self.sample_program())
    self._flush_ep()
    self.ep_t[ii] = 0
    tb = getattr(self, '_tb', None)
    if full:
      self._ep_same = 0
    elif self._ep_same is not None and self._ep_same != 0:
      if tb is not None:
        tb['ep_t'].fill_(self._ep_same)      # the device clocks become per-environment from here on
      self._ep_same = None
    if self._ep_same is None and tb is not None:
      import torch
      tb['ep_t'][torch.as_tensor(np.asarray(ii, np.int64), device=tb['ep_t'].device)] = 0

  def _flush_ep(self):
    if self._ep_lag:
      self.ep_t += self._ep_lag
      self._ep_lag = 0

  def sample_actions(self, T=None):
    shape = (self.num_envs, self.act_size) if T is None else (T, self.num_envs, self.act_size)
    return self._act_rng.uniform(-1, 1, shape).astype(np.float32)

  def reset(self, idxs=None, full_state=None, proprio=None):
    h = self._handle()
    idxs = np.arange(self.num_envs, dtype=np.int32) if idxs is None else np.asarray(idxs, dtype=np.int32)
    n = len(idxs)
    self._reset_on_device(idxs)
    if proprio is not None:
      fs = np.zeros((n, self.obs_size))
      fs[:, self.pobs_idxs] = np.asarray(proprio, dtype=np.float64)
      full_state = fs
    if full_state is not None:
      h.set_poses(idxs, self._state_to_poses(np.asarray(full_state, dtype=np.float64)), None)
    return self._obs()

  def _obs(self):
    fs, lcd = self._handle().get_obs(np.float32)
    proprio = fs[:, self.pobs_idxs] if self.pobs_size != 0 else np.zeros((self.num_envs, 1), np.float32)
    return {'full_state': fs, 'proprio': proprio, 'lcd': lcd.astype(bool)}

  def _step_handle(self, a, fs=None, lcd=None):
    """blcd_step / blcd_step_obs complete the step for EVERY environment and then report BLCD_ERR_ENV_FAULT if any environment
    carries a device fault flag (contact-slot overflow, non-finite state).  One bad environment must not abort a 100k batch: the
    bookkeeping goes on, the flags stay readable through `faults()` / `infos[i]['fault']` until those envs are reset.
    With fs / lcd buffers the observation comes back with the step (one call, one stream synchronisation)."""
    from ._lib import EnvFaultError
    try:
      if fs is None and lcd is None:
        self._handle().step(a, 1)
      else:
        self._handle().step_obs(a, fs, lcd)
      self._any_fault = False
    except EnvFaultError:
      self._any_fault = True

  def faults(self):
    """uint32 [N] device fault flags (0 = healthy); sticky until the environment is reset."""
    return self._handle().faults()

  def step(self, actions):
    a = np.ascontiguousarray(np.asarray(actions, dtype=np.float32).reshape(self.num_envs, self.act_size))
    d = self.scene.desc
    fs = np.zeros((self.num_envs, self.obs_size), np.float32)
    lcd = np.zeros((self.num_envs, d.lcd_h, d.lcd_w), np.uint8)
    self._step_handle(a, fs, lcd)
    self._flush_ep()
    self.ep_t += 1
    if self._ep_same is not None:
      self._ep_same += 1
    else:
      tb = getattr(self, '_tb', None)
      if tb is not None:
        tb['ep_t'] += 1
    done = self.ep_t >= self.G.ep_len
    infos = [{'timeout': bool(d_)} for d_ in done]
    if self._any_fault:
      for i in np.nonzero(self.faults())[0]:
        infos[int(i)]['fault'] = True
    proprio = fs[:, self.pobs_idxs] if self.pobs_size != 0 else np.zeros((self.num_envs, 1), np.float32)
    return {'full_state': fs, 'proprio': proprio, 'lcd': lcd.astype(bool)}, np.zeros(self.num_envs, np.float64), done, infos

  # ---- device-resident surface: torch CUDA tensors in and out, no host copies, no per-env Python objects -----------------
  # What the reference's GPU-side consumers loop over (research/rl/ppo.py:127-133 `o, r, d, info = env.step(a)`,
  # rl/sac.py:200-214, data.py:50-77), with the vector env's call shapes but tensors that stay in HBM.
  def _torch_bufs(self):
    import torch
    if getattr(self, '_tb', None) is None:
      dev = torch.device('cuda', self._device)
      d = self.scene.desc
      self._tb = {'full_state': torch.empty((self.num_envs, self.obs_size), dtype=torch.float32, device=dev),
                  'lcd': torch.empty((self.num_envs, d.lcd_h, d.lcd_w), dtype=torch.uint8, device=dev),
                  'ep_t': torch.zeros(self.num_envs, dtype=torch.int64, device=dev),
                  'rew': torch.zeros(self.num_envs, dtype=torch.float64, device=dev),
                  'pidx': torch.as_tensor(np.asarray(self.pobs_idxs, np.int64), device=dev),
                  'all_true': torch.ones(self.num_envs, dtype=torch.bool, device=dev),
                  'all_false': torch.zeros(self.num_envs, dtype=torch.bool, device=dev),
                  'proprio0': torch.zeros((self.num_envs, 1), dtype=torch.float32, device=dev)}
      if self._ep_same is None:
        self._flush_ep()
        self._tb['ep_t'].copy_(torch.as_tensor(self.ep_t, device=dev))
    return self._tb

  def _obs_torch(self):
    tb = self._torch_bufs()
    self._handle().get_obs_into(tb['full_state'], tb['lcd'])
    fs = tb['full_state']
    proprio = fs[:, tb['pidx']] if self.pobs_size != 0 else tb['proprio0']
    return {'full_state': fs, 'proprio': proprio, 'lcd': tb['lcd']}

  def reset_torch(self, idxs=None):
    """reset(idxs) returning CUDA tensors: {'full_state' f32 [N, obs], 'proprio' f32 [N, pobs], 'lcd' uint8 [N, H, W] (0/1)}.
    The returned tensors are views of per-env buffers that the next call overwrites (clone to keep)."""
    import torch
    tb = self._torch_bufs()
    h = self._handle()
    ii = np.arange(self.num_envs, dtype=np.int32) if idxs is None else np.asarray(torch.as_tensor(idxs).cpu() if hasattr(idxs, 'cpu') else idxs, dtype=np.int32)
    self._reset_on_device(ii)          # sampled on the device: no pose ever crosses PCIe (and the episode clocks of ii restart)
    return self._obs_torch()

  def step_torch(self, actions, sync=True):
    """step(actions) on CUDA tensors: actions f32 [N, act] on the device -> (obs dict of tensors, rew f64 [N] zeros,
    done bool [N], timeout bool [N]) - ONE call (blcd_step_obs: the step kernel writes the observation row and the frame itself),
    one stream synchronisation, nothing crosses PCIe.
    sync=False: no host synchronisation at all (blcd_step_obs_async) - the step is ordered between the torch work before and after it
    on the device, the call returns while it runs; device faults are then only seen through `faults()`.
    sync='inline': as False, with the step queued ON torch's current stream (blcd_set_async_stream) - no hand-off between streams either;
    meant for the small scene classes (a stream's hardware queue reserves scratch for the largest kernel it has run)."""
    tb = self._torch_bufs()
    a = actions.contiguous() if actions.dtype == tb['full_state'].dtype else actions.float().contiguous()
    assert a.is_cuda and tuple(a.shape) == (self.num_envs, self.act_size)
    if not (sync is False or (isinstance(sync, str) and sync == 'inline')):
      self._step_handle(a, tb['full_state'], tb['lcd'])        # blcd_step_obs; a faulted environment does not abort the batch: see faults()
    else:
      self._handle().step_obs_async(a, tb['full_state'], tb['lcd'], inline=isinstance(sync, str))
    self._ep_lag += 1                  # the host clocks are brought up to date when something reads them (_flush_ep)
    if self._ep_same is not None:      # every environment at the same step of its episode: a constant, no kernel
      self._ep_same += 1
      done = tb['all_true'] if self._ep_same >= int(self.G.ep_len) else tb['all_false']
    else:
      tb['ep_t'] += 1
      done = tb['ep_t'] >= int(self.G.ep_len)
    fs = tb['full_state']
    proprio = fs[:, tb['pidx']] if self.pobs_size != 0 else tb['proprio0']
    return {'full_state': fs, 'proprio': proprio, 'lcd': tb['lcd']}, tb['rew'], done, done

  def lcd_render(self, width=None, height=None, lcd_mode='1'):
    """Batched reference world_env.py:460-512: bool [N, H, W] for '1', uint8 [N, H, W, 3] for 'RGB', any canvas size."""
    lcd_mode = lcd_mode.upper()
    assert lcd_mode in ['1', 'RGB'], 'lcd_mode must be in one of these PIL supported modes'
    dw, dh = int(self.G.lcd_base * self.G.wh_ratio), self.G.lcd_base
    if width is None and height is None:
      width, height = dw, dh
    h = self._handle()
    if lcd_mode == '1' and (width, height) == (dw, dh):
      return h.get_obs(None)[1].astype(bool)
    img = h.render_poses_ex(h.get_poses()[:, :, :3], h.shape_sel(), int(width), int(height), lcd_mode)
    return img.astype(bool) if lcd_mode == '1' else img

  def render_states(self, full_state):
    """state -> LCD for M normalised states (the `env.reset(proprio=s)['lcd']` use), without touching env state."""
    fs = np.asarray(full_state, dtype=np.float64)
    return self._handle().render_poses(self._state_to_poses(fs), None).astype(bool)
