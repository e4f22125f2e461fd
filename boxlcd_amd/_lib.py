"""ctypes binding of libboxlcd_hip.so (include/boxlcd.h).  Fails loudly: no library or no GPU -> RuntimeError."""
import ctypes as C
import os
import numpy as np
from .scene import SceneDesc

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLCD_LIB: a side-by-side diagnostic build (__graft_entry__ BLCD_VARIANT); the product library otherwise
LIB_PATH = os.path.join(_HERE, os.environ.get('BLCD_LIB', 'libboxlcd_hip.so'))
BODY_F, JOINT_F, PAIR_F = 12, 5, 18

SYMBOLS = {
    'blcd_version': (C.c_int, []),
    'blcd_build_features': (C.c_int, []),
    'blcd_last_error': (C.c_char_p, []),
    'blcd_device_count': (C.c_int, []),
    'blcd_create': (C.c_int, [C.POINTER(SceneDesc), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    'blcd_destroy': (C.c_int, [C.c_void_p]),
    'blcd_num_envs': (C.c_int, [C.c_void_p]),
    'blcd_num_pairs': (C.c_int, [C.c_void_p]),
    'blcd_pair_table': (C.c_int, [C.c_void_p, C.c_void_p]),
    'blcd_reset': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'blcd_set_poses': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'blcd_step': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'blcd_step_obs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'blcd_step_obs_async': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'blcd_set_async_stream': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'blcd_reset_sampled': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p, C.c_int32]),
    'blcd_sample_reseed': (C.c_int, [C.c_void_p]),
    'blcd_sample_set_base': (C.c_int, [C.c_void_p, C.c_uint64]),
    'blcd_get_shape_sel': (C.c_int, [C.c_void_p, C.c_void_p]),
    'blcd_rollout': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'blcd_rollout_bits': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'blcd_goal_set': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'blcd_goal_seed': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'blcd_goal_eval': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'blcd_get_obs': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'blcd_render_poses': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'blcd_set_ellipse_rgb_lut': (C.c_int, [C.c_void_p, C.c_int32]),
    'blcd_render_poses_ex': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'blcd_pack_bits': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'blcd_unpack_bits': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'blcd_get_poses': (C.c_int, [C.c_void_p, C.c_void_p]),
    'blcd_get_state': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]),
    'blcd_set_state': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'blcd_get_faults': (C.c_int, [C.c_void_p, C.c_void_p]),
    'blcd_sync': (C.c_int, [C.c_void_p]),
    'blcd_stream': (C.c_void_p, [C.c_void_p]),
    'blcd_last_kernel_ms': (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    'blcd_sched_stats': (C.c_int, [C.c_void_p, C.c_void_p]),
    'blcd_debug_wave_times': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    'blcd_debug_world_step': (C.c_int, [C.c_void_p, C.c_int32]),
    'blcd_debug_set_motor_speeds': (C.c_int, [C.c_void_p, C.c_void_p]),
    'blcd_debug_dump': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'blcd_debug_sincos': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]),
    'blcd_debug_mass_data': (C.c_int, [C.POINTER(SceneDesc), C.c_int32, C.c_float, C.c_void_p]),
    'blcd_debug_collide': (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None
FEATURE_WAVETIMES, FEATURE_SCHED = 1, 2


def features():
  """bit mask of the optional parts compiled into the library (include/boxlcd.h BLCD_FEATURE_*)"""
  return load().blcd_build_features()


def load():
  """dlopen the HIP library and bind every symbol include/boxlcd.h declares.  Raises RuntimeError if it is missing."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       '(hipcc --offload-arch=gfx950). boxlcd_amd has no CPU fallback.')
  try:
    import torch  # noqa: F401  torch bundles its own HIP runtime; it has to be the first one the process loads, or
  except ImportError:   # `import torch` AFTER this library reports "No HIP GPUs are available" (two libamdhip64 with one soname)
    pass
  lib = C.CDLL(LIB_PATH)
  for name, (res, args) in SYMBOLS.items():
    fn = getattr(lib, name)  # AttributeError if the library does not export it
    fn.restype, fn.argtypes = res, args
  _lib = lib
  # Pillow's ellipse fill/outline span table (DATA, tools/gen_ellipse_rgb_lut.py) for lcd_render(width, height, 'RGB')
  lut = np.fromfile(os.path.join(os.path.dirname(LIB_PATH), 'ellipse_rgb_lut.bin'), np.uint8)
  amax = 0
  while (amax + 1) * 5 * (amax + 3) * 6 < lut.size:
    amax += 1
  if (amax + 1) * 5 * (amax + 3) * 6 != lut.size:
    raise RuntimeError('ellipse_rgb_lut.bin has an unexpected size')
  if lib.blcd_set_ellipse_rgb_lut(lut.ctypes.data_as(C.c_void_p), amax) != 0:
    raise RuntimeError('blcd_set_ellipse_rgb_lut failed')
  return lib


def pack_bits(src, dst, stream_ptr=None):
  """uint8 0/1 CUDA tensor -> packed CUDA tensor (numel/8 bytes), asynchronously on the given hipStream_t"""
  _check(load().blcd_pack_bits(src.data_ptr(), dst.data_ptr(), src.numel(), stream_ptr))


def unpack_bits(src, dst, stream_ptr=None):
  _check(load().blcd_unpack_bits(src.data_ptr(), dst.data_ptr(), dst.numel(), stream_ptr))


class EnvFaultError(RuntimeError):
  """BLCD_ERR_ENV_FAULT: the call completed, but at least one environment carries a device fault flag (Handle.faults())."""


def _check(rc):
  if rc != 0:
    msg = load().blcd_last_error()
    text = f'boxlcd_hip error {rc}: {msg.decode() if msg else "?"}'
    raise (EnvFaultError if rc == -5 else RuntimeError)(text)


def _ptr(x):
  """numpy array -> host pointer; torch tensor -> its data_ptr (device or host); None -> NULL."""
  if x is None:
    return None
  if hasattr(x, 'data_ptr'):
    return C.c_void_p(x.data_ptr())
  return x.ctypes.data_as(C.c_void_p)


class SampleOp(C.Structure):
  """blcd_sample_op (include/boxlcd.h): one instruction of a scene's reset-sampling program"""
  _fields_ = [('kind', C.c_int32), ('d', C.c_int32), ('a', C.c_int32), ('b', C.c_int32), ('body', C.c_int32), ('parent', C.c_int32),
              ('f', C.c_double * 5)]


class GoalDesc(C.Structure):
  """blcd_goal_desc (include/boxlcd.h)"""
  _fields_ = [('mode', C.c_int32), ('diff_delt', C.c_int32), ('n_idx', C.c_int32), ('idxs', C.c_int32 * 96),
              ('thresh', C.c_double), ('rew_scale', C.c_double)]


class Handle:
  """One blcd_handle: n_envs environments of one scene on one GPU."""

  def __init__(self, desc, n_envs, device=0):
    self.lib = load()
    self.desc = desc
    self.n = int(n_envs)
    self.nb, self.nj = desc.n_bodies, desc.n_joints
    self.n_obs, self.n_act = desc.n_obs, desc.n_act
    self.h, self.w = desc.lcd_h, desc.lcd_w
    out = C.c_void_p()
    _check(self.lib.blcd_create(C.byref(desc), self.n, int(device), C.byref(out)))
    self._h = out
    self.n_pairs = self.lib.blcd_num_pairs(self._h)

  def close(self):
    if self._h is not None:
      self.lib.blcd_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  _ext = None

  def _after_torch(self, *xs):
    """Order the call behind torch.  The library launches on its OWN (non-blocking) stream; a CUDA tensor handed to a call was produced -
    and an output buffer may still be read - by work queued on torch's current stream (`step_torch(policy(obs))`: the policy's kernels
    have been launched, not finished).  So the handle's stream first waits, on the device, for torch's current stream (an event +
    hipStreamWaitEvent, no host synchronisation).  The other direction needs nothing: every call returns after synchronising its stream."""
    for x in xs:
      if x is not None and getattr(x, 'is_cuda', False):
        import torch
        if self._ext is None:
          self._ext = (torch.cuda.ExternalStream(int(self.stream()), device=x.device), torch.cuda.Event())
        ext, ev = self._ext
        ev.record(torch.cuda.current_stream(x.device))   # one cached event: each record supersedes the last
        ext.wait_event(ev)
        return

  def pair_table(self):
    t = np.zeros((self.n_pairs, 2), np.int32)
    _check(self.lib.blcd_pair_table(self._h, _ptr(t)))
    return t

  def reset(self, idxs, poses, shape_sel=None):
    n = self.n if idxs is None else len(idxs)
    if idxs is not None:
      idxs = np.ascontiguousarray(idxs, dtype=np.int32)
    if not hasattr(poses, 'data_ptr'):
      poses = np.ascontiguousarray(poses, dtype=np.float32)
      assert poses.shape == (n, self.nb, 3), poses.shape
    if shape_sel is not None and not hasattr(shape_sel, 'data_ptr'):
      shape_sel = np.ascontiguousarray(shape_sel, dtype=np.int32)
    self._after_torch(poses, shape_sel)
    _check(self.lib.blcd_reset(self._h, _ptr(idxs), n, _ptr(poses), _ptr(shape_sel)))

  def set_poses(self, idxs, poses, mask=None):
    n = self.n if idxs is None else len(idxs)
    if idxs is not None:
      idxs = np.ascontiguousarray(idxs, dtype=np.int32)
    poses = np.ascontiguousarray(poses, dtype=np.float32)
    assert poses.shape == (n, self.nb, 3), poses.shape
    if mask is not None:
      mask = np.ascontiguousarray(mask, dtype=np.uint8)
    _check(self.lib.blcd_set_poses(self._h, _ptr(idxs), n, _ptr(poses), _ptr(mask)))

  def step(self, actions=None, n_steps=1):
    if actions is not None and not hasattr(actions, 'data_ptr'):
      actions = np.ascontiguousarray(actions, dtype=np.float32)
      assert actions.shape == (self.n, self.n_act), actions.shape
    self._after_torch(actions)
    _check(self.lib.blcd_step(self._h, _ptr(actions), int(n_steps)))

  def step_obs(self, actions, fs, lcd):
    """blcd_step_obs: one env-step + float32 observations + LCD into fs / lcd (numpy or torch, host or device), one sync"""
    if actions is not None and not hasattr(actions, 'data_ptr'):
      actions = np.ascontiguousarray(actions, dtype=np.float32)
      assert actions.shape == (self.n, self.n_act), actions.shape
    self._after_torch(actions, fs, lcd)
    _check(self.lib.blcd_step_obs(self._h, _ptr(actions), _ptr(fs), _ptr(lcd)))

  _async_stream = None

  def step_obs_async(self, actions, fs, lcd, inline=False):
    """blcd_step_obs_async: the same step queued on the handle's stream, no host synchronisation; CUDA tensors only.  The handle's stream
    waits for torch's current stream before the step and torch's current stream waits for the handle's after it, so torch work queued
    behind this call sees the outputs (`.cpu()` / `.item()` / `torch.cuda.synchronize()` wait for the step like for any torch kernel)."""
    import torch
    assert all(x is None or x.is_cuda for x in (actions, fs, lcd))
    if inline:
      # the step is queued ON torch's current stream (blcd_set_async_stream): ordered like any torch kernel, no hand-off between streams
      cur = int(torch.cuda.current_stream(actions.device).cuda_stream)   # 0 = the device's default stream (torch's, unless told otherwise)
      if self._async_stream != cur:
        _check(self.lib.blcd_set_async_stream(self._h, C.c_void_p(cur), 1))
        self._async_stream = cur
      _check(self.lib.blcd_step_obs_async(self._h, _ptr(actions), _ptr(fs), _ptr(lcd)))
      return
    if self._async_stream is not None:
      _check(self.lib.blcd_set_async_stream(self._h, None, 0))
      self._async_stream = None
    self._after_torch(actions, fs, lcd)
    _check(self.lib.blcd_step_obs_async(self._h, _ptr(actions), _ptr(fs), _ptr(lcd)))
    ext, ev = self._ext
    torch.cuda.current_stream(ext.device).wait_stream(ext)

  def reset_sampled(self, idxs, seed, ops):
    """blcd_reset_sampled: reset environments idxs (None = all) from the device-side counter-based sampler"""
    n = self.n if idxs is None else len(idxs)
    if idxs is not None and not hasattr(idxs, 'data_ptr'):
      idxs = np.ascontiguousarray(idxs, dtype=np.int32)
    arr = (SampleOp * len(ops))(*ops)
    self._after_torch(idxs)
    _check(self.lib.blcd_reset_sampled(self._h, _ptr(idxs), n, C.c_uint64(int(seed)), arr, len(ops)))

  def sample_reseed(self):
    _check(self.lib.blcd_sample_reseed(self._h))

  def sample_set_base(self, env_id_base):
    """global id of this handle's environment 0 (sharded batches: rank * envs per rank)"""
    _check(self.lib.blcd_sample_set_base(self._h, C.c_uint64(int(env_id_base))))

  def shape_sel(self, out=None):
    """int32 [n, nb]: the shape every body currently has ('random' objects choose one per reset)"""
    out = np.zeros((self.n, self.nb), np.int32) if out is None else out
    self._after_torch(out)
    _check(self.lib.blcd_get_shape_sel(self._h, _ptr(out)))
    return out

  def rollout(self, actions, T, lcd_out=None, obs_out=None):
    if actions is not None and not hasattr(actions, 'data_ptr'):
      actions = np.ascontiguousarray(actions, dtype=np.float32)
      assert actions.shape == (T, self.n, self.n_act), actions.shape
    self._after_torch(actions, lcd_out, obs_out)
    _check(self.lib.blcd_rollout(self._h, _ptr(actions), int(T), _ptr(lcd_out), _ptr(obs_out)))

  def rollout_bits(self, actions, T, lcd_bits_out=None, obs_out=None):
    """rollout() with frames at one bit per pixel: lcd_bits_out uint8 [T, n, lcd_h, lcd_w // 8], numpy bitorder='little'"""
    if actions is not None and not hasattr(actions, 'data_ptr'):
      actions = np.ascontiguousarray(actions, dtype=np.float32)
      assert actions.shape == (T, self.n, self.n_act), actions.shape
    self._after_torch(actions, lcd_bits_out, obs_out)
    _check(self.lib.blcd_rollout_bits(self._h, _ptr(actions), int(T), _ptr(lcd_bits_out), _ptr(obs_out)))

  def goal_set(self, mode, idxs_cols, thresh, rew_scale, diff_delt, goal_full_state, goal_lcd=None, env_idxs=None):
    g = GoalDesc()
    g.mode, g.diff_delt, g.n_idx = int(mode), int(bool(diff_delt)), len(idxs_cols)
    for i, c in enumerate(idxs_cols):
      g.idxs[i] = int(c)
    g.thresh, g.rew_scale = float(thresh), float(rew_scale)
    if not hasattr(goal_full_state, 'data_ptr'):
      goal_full_state = np.ascontiguousarray(goal_full_state, dtype=np.float64)
    if goal_lcd is not None and not hasattr(goal_lcd, 'data_ptr'):
      goal_lcd = np.ascontiguousarray(goal_lcd, dtype=np.uint8)
    n = self.n if env_idxs is None else len(env_idxs)
    if env_idxs is not None:
      env_idxs = np.ascontiguousarray(env_idxs, dtype=np.int32)
    self._after_torch(goal_full_state, goal_lcd)
    _check(self.lib.blcd_goal_set(self._h, C.byref(g), _ptr(env_idxs), n, _ptr(goal_full_state), _ptr(goal_lcd)))

  def goal_seed(self, env_idxs=None):
    n = self.n if env_idxs is None else len(env_idxs)
    if env_idxs is not None:
      env_idxs = np.ascontiguousarray(env_idxs, dtype=np.int32)
    _check(self.lib.blcd_goal_seed(self._h, _ptr(env_idxs), n))

  def goal_eval(self, rew=None, done=None, delta=None):
    """-> (rew f64[N], done bool[N], delta f64[N]); pass torch tensors to keep the results on the device."""
    rew = np.zeros(self.n, np.float64) if rew is None else rew
    done = np.zeros(self.n, np.uint8) if done is None else done
    delta = np.zeros(self.n, np.float64) if delta is None else delta
    self._after_torch(rew, done, delta)
    _check(self.lib.blcd_goal_eval(self._h, _ptr(rew), _ptr(done), _ptr(delta)))
    return rew, done, delta

  def get_obs(self, dtype=np.float32, lcd=True):
    fs = None
    if dtype is not None:
      fs = np.zeros((self.n, self.n_obs), dtype)
    img = np.zeros((self.n, self.h, self.w), np.uint8) if lcd else None
    _check(self.lib.blcd_get_obs(self._h, _ptr(fs), 1 if dtype == np.float64 else 0, _ptr(img)))
    return fs, img

  def get_obs_into(self, fs, lcd):
    """device (torch) or host buffers, float32 obs."""
    self._after_torch(fs, lcd)
    _check(self.lib.blcd_get_obs(self._h, _ptr(fs), 0, _ptr(lcd)))

  def render_poses(self, poses, shape_sel=None):
    poses = np.ascontiguousarray(poses, dtype=np.float32)
    m = poses.shape[0]
    assert poses.shape == (m, self.nb, 3)
    if shape_sel is not None:
      shape_sel = np.ascontiguousarray(shape_sel, dtype=np.int32)
    img = np.zeros((m, self.h, self.w), np.uint8)
    _check(self.lib.blcd_render_poses(self._h, _ptr(poses), _ptr(shape_sel), m, _ptr(img)))
    return img

  def render_poses_ex(self, poses, shape_sel, width, height, mode='1'):
    """lcd_render(width, height, lcd_mode) for m pose sets: '1' -> uint8 [m, H, W]; 'RGB' -> uint8 [m, H, W, 3]"""
    poses = np.ascontiguousarray(poses, dtype=np.float32)
    m = poses.shape[0]
    assert poses.shape == (m, self.nb, 3)
    if shape_sel is not None:
      shape_sel = np.ascontiguousarray(shape_sel, dtype=np.int32)
    rgb = mode.upper() == 'RGB'
    img = np.zeros((m, height, width, 3) if rgb else (m, height, width), np.uint8)
    _check(self.lib.blcd_render_poses_ex(self._h, _ptr(poses), _ptr(shape_sel), m, int(width), int(height), int(rgb), _ptr(img)))
    return img

  def get_poses(self):
    p = np.zeros((self.n, self.nb, 4), np.float32)
    _check(self.lib.blcd_get_poses(self._h, _ptr(p)))
    return p

  def get_state(self):
    size = C.c_size_t(0)
    _check(self.lib.blcd_get_state(self._h, None, C.byref(size)))
    blob = np.zeros(size.value, np.uint8)
    _check(self.lib.blcd_get_state(self._h, _ptr(blob), C.byref(size)))
    return blob

  def set_state(self, blob):
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    _check(self.lib.blcd_set_state(self._h, _ptr(blob), blob.size))

  def faults(self):
    f = np.zeros(self.n, np.int32)
    _check(self.lib.blcd_get_faults(self._h, _ptr(f)))
    return f

  def sync(self):
    _check(self.lib.blcd_sync(self._h))

  def stream(self):
    return self.lib.blcd_stream(self._h)

  def last_kernel_ms(self):
    ms, n = C.c_float(0), C.c_int32(0)
    _check(self.lib.blcd_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
    return ms.value, n.value

  def sched_stats(self):
    """counters of the environment-level scheduler since the last call (include/boxlcd.h blcd_sched_stats)"""
    out = np.zeros(8, np.uint64)
    _check(self.lib.blcd_sched_stats(self._h, _ptr(out)))
    return dict(first_live=int(out[0]), first_suspended=int(out[1]), first_waves=int(out[2]), passes=int(out[3]),
                later_live=int(out[4]), later_suspended=int(out[5]), later_waves=int(out[6]), max_lanes=int(out[7]))

  def debug_wave_times(self):
    out = np.zeros((self.n, 9), np.uint64)
    n = self.lib.blcd_debug_wave_times(self._h, _ptr(out), self.n * 9)
    if n < 0:
      _check(n)
    return out[:n]

  # parity hooks
  def debug_world_step(self, n=1):
    _check(self.lib.blcd_debug_world_step(self._h, int(n)))

  def debug_set_motor_speeds(self, actions):
    actions = np.ascontiguousarray(actions, dtype=np.float32)
    _check(self.lib.blcd_debug_set_motor_speeds(self._h, _ptr(actions)))

  def debug_dump(self):
    b = np.zeros((self.n, self.nb, BODY_F), np.float32)
    j = np.zeros((self.n, max(self.nj, 1), JOINT_F), np.float32)
    p = np.zeros((self.n, max(self.n_pairs, 1), PAIR_F), np.float32)
    _check(self.lib.blcd_debug_dump(self._h, _ptr(b), _ptr(j), _ptr(p)))
    return b, j[:, :self.nj], p[:, :self.n_pairs]
