"""Goal wrappers: the device epilogue (blcd_goal_set / _seed / _eval) against fixtures computed in the
authoring container from the reference's rules (research/wrappers/body_goal.py:58-88, cube_goal.py:64-86) on the CPU oracle's
observations - tools/gen_goal_fixtures.py -> tests/golden/goal_fixtures.npz.  Nothing here imports the rules themselves."""
import ast
import numpy as np
import pytest
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
from boxlcd_amd.goal import BodyGoalEnv, CubeGoalEnv

pytestmark = pytest.mark.gpu
FIX = np.load('tests/golden/goal_fixtures.npz')


@pytest.mark.parametrize('ci', range(7))
def test_goal_epilogue_equals_reference_rules(ci):
  k = f'case{ci}'
  kind, name, G = FIX[k + '_meta'].tolist()
  G = ast.literal_eval(G)
  venv = B.BatchedWorldEnv(name, 1)
  wrap = (BodyGoalEnv if kind == 'body' else CubeGoalEnv)(venv, G)
  gposes, gsel, poses, sel, acts = (FIX[k + s] for s in ('_gposes', '_gsel', '_poses', '_sel', '_acts'))
  n, T = poses.shape[0], acts.shape[0]
  h = Handle(venv.scene.desc, n, 0)
  h.reset(None, gposes, gsel)                                   # the goal: a fresh state (cube: settled for 10 zero-action steps)
  if kind == 'cube':
    h.step(None, 10)
  gfs, glcd = h.get_obs(np.float64)
  if kind == 'body':
    mode = 0 if G['state_rew'] else 1
    cols, thresh = wrap._cols, (G['goal_thresh'] if mode == 0 else 0.70)
  else:
    mode, cols, thresh = 0, wrap.idxs, 0.05
  h.reset(None, poses, sel)
  h.goal_set(mode, cols, thresh, G['rew_scale'], G['diff_delt'], gfs, glcd)
  h.goal_seed()
  for t in range(T):
    h.step(acts[t], 1)
    rew, done, delta = h.goal_eval()
    assert (rew == FIX[k + '_rew'][t]).all(), (t, np.abs(rew - FIX[k + '_rew'][t]).max())
    assert (done.astype(bool) == FIX[k + '_done'][t]).all()
    exp = FIX[k + '_delta'][t]
    assert (delta[~np.isnan(exp)] == exp[~np.isnan(exp)]).all()
  h.close()


def test_wrapper_api_and_partial_reset_keeps_running_envs():
  """reset(idxs) of a batched goal env touches only the listed environments (reference: one wrapper per worker,
  async_vector_env.py:131-189): the others keep state, goal and last delta - their next rewards equal an undisturbed twin's."""
  n, G = 48, {'state_rew': 1, 'diff_delt': 1, 'goal_thresh': 0.05, 'rew_scale': 1.0}
  for cls, name in ((BodyGoalEnv, 'Urchin'), (CubeGoalEnv, 'UrchinCube')):
    a = cls(B.BatchedWorldEnv(name, n, seed=4), G)
    b = cls(B.BatchedWorldEnv(name, n, seed=4), G)
    oa, ob = a.reset(), b.reset()
    assert set(oa) >= {'full_state', 'proprio', 'lcd', 'goal:lcd', 'goal:proprio'} and (oa['goal:lcd'] == ob['goal:lcd']).all()
    acts = a._env.sample_actions(6)
    for t in range(3):
      ra, rb = a.step(acts[t]), b.step(acts[t])
      assert (ra[1] == rb[1]).all()
    idxs = np.array([1, 7, 30], np.int32)
    keep = np.setdiff1d(np.arange(n), idxs)
    goal_before = a.goal['lcd'].copy()
    o = a.reset(idxs)
    assert (a.goal['lcd'][keep] == goal_before[keep]).all() and (o['goal:lcd'][keep] == goal_before[keep]).all()
    for t in range(3, 6):
      ra, rb = a.step(acts[t]), b.step(acts[t])
      assert (ra[1][keep] == rb[1][keep]).all() and (ra[0]['full_state'][keep] == rb[0]['full_state'][keep]).all()
      assert all(ra[3][i]['delta'] == rb[3][i]['delta'] for i in keep)
    a.close(); b.close()


def test_single_env_goal_wrapper():
  env = BodyGoalEnv(B.envs.Luxo(), {'state_rew': 1, 'diff_delt': 1, 'goal_thresh': 0.05, 'rew_scale': 1.0})
  env.seed(2)
  obs = env.reset()
  assert set(obs) >= {'full_state', 'proprio', 'lcd', 'goal:lcd', 'goal:proprio'}
  o2, rew, done, info = env.step(np.array([0.3, -0.2, 0.9], np.float32))
  assert isinstance(rew, float) and isinstance(done, bool) and 'delta' in info
  # body_goal.py:58-75 by hand: mean |goal - proprio| over the x/y entries, reward -0.05 + 10 (last - now)
  cols = [env.pobs_keys.index(k) for k in env.pobs_keys if k.endswith(('x:p', 'y:p'))]
  now = np.abs(o2['goal:proprio'] - o2['proprio'])[cols].mean()
  last = np.abs(obs['goal:proprio'] - obs['proprio'])[cols].mean()
  exp = -0.05 + 10 * (last - now) + (1.0 if now < 0.05 else 0.0)      # + success bonus below goal_thresh (body_goal.py:76-79)
  assert info['delta'] == now and rew == exp * 1.0 and done == bool(now < 0.05 or info['timeout'])
  env.close()
