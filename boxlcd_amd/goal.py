"""Goal-conditioned wrappers with the reward / done epilogue evaluated on the GPU.

Mirror of research/wrappers/body_goal.py (BodyGoalEnv) and research/wrappers/cube_goal.py (CubeGoalEnv): same
constructor `(env, G)`, same observation keys ('goal:lcd', 'goal:proprio', and for the cube variant 'goal:full_state',
'goal:object'), same reward and done rules (include/boxlcd.h: blcd_goal_*).  `env` is a WorldEnv (one environment) or a
BatchedWorldEnv (N environments, vector-env call shapes: rew f64[N], done bool[N])."""
import copy
import re

import numpy as np

from .world_env import BatchedWorldEnv


def _filtlist(keys, phrase):          # research/utils.py:38
  return [k for k in keys if re.match(phrase, k) is not None]


def _get(G, key, default):
  try:
    return G[key] if key in G else default
  except TypeError:
    return getattr(G, key, default)


class _GoalBase:
  def __init__(self, env, G):
    self._env = env
    self.SCALE = 2
    self.G = G
    self._batched = isinstance(env, BatchedWorldEnv)

  def seed(self, *args):
    return self._env.seed(*args)

  @property
  def action_space(self):
    return self._env.action_space

  def render(self, *args, **kwargs):
    return self._env.render(*args, **kwargs)

  def __getattr__(self, name):        # obs_keys, pobs_keys, num_envs, ... of the wrapped env
    return getattr(self._env, name)

  # -- helpers ------------------------------------------------------------------------------------
  def _handle(self):
    return self._env._handle()

  def _snapshot_goal(self):
    """float64 full_state + LCD of the current (goal) state, as the per-process reference wrapper sees them"""
    self._goal64 = self._handle().get_obs(np.float64)

  def _install(self, mode, cols, thresh, diff_delt, idxs=None):
    fs64, lcd = self._goal64
    if idxs is not None:
      fs64, lcd = fs64[idxs], lcd[idxs]
    self._handle().goal_set(mode, cols, thresh, _get(self.G, 'rew_scale', 1.0), diff_delt, fs64, lcd, env_idxs=idxs)

  # Batched envs: the reference gives every sub-environment its own wrapper, and AsyncVectorEnv.reset(idxs) only touches the
  # listed workers (research/wrappers/async_vector_env.py:131-189).  So a goal is sampled - and, for the cube variant, settled for
  # `settle` zero-action env-steps - in a scratch handle holding just those environments; the running ones keep their state,
  # goal and last delta.
  def _goal_from_scratch(self, idxs, settle):
    env = self._env
    n = len(idxs)
    poses, sel = env.sample_initial(n)
    from ._lib import Handle
    key = n
    if getattr(self, '_scratch', None) is None or self._scratch[0] != key:
      if getattr(self, '_scratch', None) is not None:
        self._scratch[1].close()
      self._scratch = (key, Handle(env.scene.desc, n, env._device))
    sh = self._scratch[1]
    sh.reset(None, poses, sel)
    if settle:
      sh.step(None, settle)
    fs64, lcd = sh.get_obs(np.float64)
    if getattr(self, '_goal64', None) is None or self._goal64[0].shape[0] != env.num_envs:
      self._goal64 = (np.zeros((env.num_envs, env.obs_size), np.float64), np.zeros((env.num_envs,) + lcd.shape[1:], np.uint8))
      self.goal = {'full_state': np.zeros((env.num_envs, env.obs_size), np.float32), 'lcd': np.zeros((env.num_envs,) + lcd.shape[1:], bool),
                   'proprio': np.zeros((env.num_envs, max(env.pobs_size, 1)), np.float32)}
    self._goal64[0][idxs], self._goal64[1][idxs] = fs64, lcd
    self.goal['full_state'][idxs] = fs64.astype(np.float32)
    self.goal['lcd'][idxs] = lcd.astype(bool)
    if env.pobs_size:
      self.goal['proprio'][idxs] = fs64[:, env.pobs_idxs].astype(np.float32)

  def _batched_reset(self, idxs, settle, mode, cols, thresh, kwargs):
    env = self._env
    idxs = np.arange(env.num_envs, dtype=np.int32) if idxs is None else np.asarray(idxs, dtype=np.int32)
    self._goal_from_scratch(idxs, settle)
    obs = env.reset(idxs, **kwargs)
    self._install(mode, cols, thresh, _get(self.G, 'diff_delt', 0), idxs)
    self._handle().goal_seed(idxs)
    return obs

  def close(self):
    if getattr(self, '_scratch', None) is not None:
      self._scratch[1].close()
      self._scratch = None
    self._env.close()

  def _eval(self):
    rew, done, delta = self._handle().goal_eval()
    if self._batched:
      return rew, done.astype(bool), delta
    return float(rew[0]), bool(done[0]), float(delta[0])


class BodyGoalEnv(_GoalBase):
  """research/wrappers/body_goal.py:15-104.  G: state_rew, diff_delt, goal_thresh, rew_scale."""

  def __init__(self, env, G):
    super().__init__(env, G)
    keys = _filtlist(env.pobs_keys, '.*(x|y):p')                       # body_goal.py:63
    pidx = [env.pobs_keys.index(k) for k in keys]                      # :64 (indices into proprio)
    self._cols = [int(env.pobs_idxs[i]) for i in pidx]                 # the same entries as columns of full_state

  @property
  def observation_space(self):
    base_space = self._env.observation_space
    base_space.spaces['goal:lcd'] = base_space.spaces['lcd']
    base_space.spaces['goal:proprio'] = base_space.spaces['proprio']
    return base_space

  def _mode(self):
    return 0 if _get(self.G, 'state_rew', 1) else 1

  def reset(self, *args, **kwargs):
    thresh = _get(self.G, 'goal_thresh', 0.05) if self._mode() == 0 else 0.70
    if self._batched:                                                  # reset(idxs=None, **kwargs): only the listed environments
      idxs = args[0] if args else kwargs.pop('idxs', None)
      obs = self._batched_reset(idxs, 0, self._mode(), self._cols, thresh, kwargs)
    else:
      self.goal = self._env.reset()                                    # body_goal.py:36: a fresh random state is the goal
      self._snapshot_goal()
      obs = self._env.reset(*args, **kwargs)
      self._install(self._mode(), self._cols, thresh, _get(self.G, 'diff_delt', 0))
      self._handle().goal_seed()
    obs['goal:lcd'] = np.array(self.goal['lcd'])
    obs['goal:proprio'] = np.array(self.goal['proprio'])
    self.last_obs = copy.deepcopy(obs)
    return obs

  def step(self, action):
    obs, rew, done, info = self._env.step(action)
    obs['goal:lcd'] = np.array(self.goal['lcd'])
    obs['goal:proprio'] = np.array(self.goal['proprio'])
    rew, _done, delta = self._eval()                                   # comp_rew_done (:58-88) * rew_scale (:98), on device
    if self._batched:
      for i, inf in enumerate(info):
        inf['delta'] = delta[i]
        if _done[i]:
          inf['success'] = True
      done = np.logical_or(done, _done)
    else:
      info['delta'] = delta
      if _done:
        info['success'] = True
      done = done or _done
    self.last_obs = copy.deepcopy(obs)
    return obs, rew, done, info


class CubeGoalEnv(_GoalBase):
  """research/wrappers/cube_goal.py:7-89.  G: diff_delt, rew_scale; success threshold 0.05 (:80)."""

  def __init__(self, env, G):
    super().__init__(env, G)
    self.keys = _filtlist(env.obs_keys, 'object.*(x|y):p')             # cube_goal.py:12
    self.idxs = [env.obs_keys.index(x) for x in self.keys]
    self.rootkeys = _filtlist(env.obs_keys, '.*root.*(x|y):p')
    self.root_idxs = [env.obs_keys.index(x) for x in self.rootkeys]

  @property
  def observation_space(self):
    base_space = self._env.observation_space
    base_space.spaces['goal:lcd'] = copy.deepcopy(base_space.spaces['lcd'])
    base_space.spaces['goal:proprio'] = copy.deepcopy(base_space.spaces['proprio'])
    base_space.spaces['goal:object'] = copy.deepcopy(base_space.spaces['proprio'])
    base_space.spaces['goal:object'].shape = (2,)
    base_space.spaces['goal:full_state'] = copy.deepcopy(base_space.spaces['full_state'])
    return base_space

  def _add_goal(self, obs):
    obs['goal:lcd'] = np.array(self.goal['lcd'])
    obs['goal:full_state'] = np.array(self.goal['full_state'])
    obs['goal:proprio'] = np.array(self.goal['proprio'])
    obs['goal:object'] = np.array(self.goal['full_state'][..., self.idxs])

  def reset(self, *args, **kwargs):
    if self._batched:
      idxs = args[0] if args else kwargs.pop('idxs', None)
      obs = self._batched_reset(idxs, 10, 0, self.idxs, 0.05, kwargs)  # cube_goal.py:36-37: the goal scene settles for 10 steps
    else:
      self.goal = self._env.reset()
      zeros = np.zeros(self._env.action_space.shape)
      for i in range(10):                                              # cube_goal.py:36-37: let the goal scene settle
        self.goal = self._env.step(zeros)[0]
      self._snapshot_goal()
      obs = self._env.reset(*args, **kwargs)
      self._install(0, self.idxs, 0.05, _get(self.G, 'diff_delt', 0))
      self._handle().goal_seed()
    self._add_goal(obs)
    self.last_obs = copy.deepcopy(obs)
    return obs

  def step(self, action):
    obs, rew, done, info = self._env.step(action)
    self._add_goal(obs)
    rew, _done, delta = self._eval()                                   # comp_rew_done (:64-86) * rew_scale (:59), on device
    if self._batched:
      for i, inf in enumerate(info):
        inf['delta'] = delta[i]
      done = np.logical_or(done, _done)
    else:
      info['delta'] = delta
      done = done or _done
    self.last_obs = copy.deepcopy(obs)
    return obs, rew, done, info
