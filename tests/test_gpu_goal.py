"""Goal wrappers: the device epilogue against the numpy restatement of the reference's rules."""
import numpy as np
import pytest
import boxlcd_amd as B
from boxlcd_amd.goal import BodyGoalEnv, CubeGoalEnv
from oracle import goal_ref

pytestmark = pytest.mark.gpu


def _per_env_obs(venv, k, fs64, lcd, goal64, goal_lcd):
  pidx = venv.pobs_idxs
  return {'full_state': fs64[k], 'proprio': fs64[k][pidx], 'lcd': lcd[k].astype(bool), 'goal:full_state': goal64[k],
          'goal:proprio': goal64[k][pidx], 'goal:lcd': goal_lcd[k].astype(bool)}


@pytest.mark.parametrize('name,state_rew,diff_delt', [('Urchin', 1, 0), ('Urchin', 1, 1), ('Luxo', 0, 0), ('Crab', 1, 1)])
def test_body_goal_matches_reference_rule(name, state_rew, diff_delt):
  n, T = 96, 12
  G = {'state_rew': state_rew, 'diff_delt': diff_delt, 'goal_thresh': 0.35, 'rew_scale': 0.5}
  venv = B.BatchedWorldEnv(name, n, seed=3)
  env = BodyGoalEnv(venv, G)
  obs = env.reset()
  assert obs['goal:lcd'].shape == obs['lcd'].shape and obs['goal:proprio'].shape == obs['proprio'].shape
  goal64, goal_lcd = env._goal64
  h = venv._handle()
  fs64, lcd = h.get_obs(np.float64)
  last = [_per_env_obs(venv, k, fs64, lcd, goal64, goal_lcd) for k in range(n)]
  n_done = 0
  for t in range(T):
    obs, rew, done, info = env.step(venv.sample_actions())
    fs64, lcd = h.get_obs(np.float64)
    for k in range(n):
      cur = _per_env_obs(venv, k, fs64, lcd, goal64, goal_lcd)
      r, d, inf = goal_ref.body_comp_rew_done(cur, last[k], venv.pobs_keys, state_rew, diff_delt, 0.35, 0.5)
      assert rew[k] == r and bool(done[k]) == (d or info[k]['timeout']) and info[k]['delta'] == inf['delta'], (t, k, rew[k], r)
      n_done += d
      last[k] = cur
  assert n_done > 0          # the thresholds are chosen so that the success branch is exercised


@pytest.mark.parametrize('name,diff_delt', [('UrchinCube', 1), ('LuxoCube', 0), ('UrchinCubes', 1)])
def test_cube_goal_matches_reference_rule(name, diff_delt):
  n, T = 64, 10
  G = {'diff_delt': diff_delt, 'rew_scale': 2.0}
  venv = B.BatchedWorldEnv(name, n, seed=5)
  env = CubeGoalEnv(venv, G)
  obs = env.reset()
  assert obs['goal:object'].shape == (n, len(env.idxs)) and len(env.idxs) >= 2
  goal64, goal_lcd = env._goal64
  h = venv._handle()
  fs64, lcd = h.get_obs(np.float64)
  last = [_per_env_obs(venv, k, fs64, lcd, goal64, goal_lcd) for k in range(n)]
  for t in range(T):
    obs, rew, done, info = env.step(venv.sample_actions())
    fs64, lcd = h.get_obs(np.float64)
    for k in range(n):
      cur = _per_env_obs(venv, k, fs64, lcd, goal64, goal_lcd)
      r, d, inf = goal_ref.cube_comp_rew_done(cur, last[k], env.idxs, diff_delt, 2.0)
      assert rew[k] == r and bool(done[k]) == (d or info[k]['timeout']), (t, k, rew[k], r)
      last[k] = cur


def test_single_env_goal_wrapper():
  env = BodyGoalEnv(B.envs.Luxo(), {'state_rew': 1, 'diff_delt': 1, 'goal_thresh': 0.05, 'rew_scale': 1.0})
  env.seed(2)
  obs = env.reset()
  assert set(obs) >= {'full_state', 'proprio', 'lcd', 'goal:lcd', 'goal:proprio'}
  o2, rew, done, info = env.step(env.action_space.sample())
  assert isinstance(rew, float) and isinstance(done, bool) and 'delta' in info
  last = {'proprio': obs['proprio'], 'goal:proprio': obs['goal:proprio']}
  r, d, inf = goal_ref.body_comp_rew_done(o2, last, env.pobs_keys, 1, 1, 0.05, 1.0)
  assert rew == r and info['delta'] == inf['delta']
  env.close()
