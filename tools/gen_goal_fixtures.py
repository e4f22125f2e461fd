"""Goal-wrapper fixtures (tests/golden/goal_fixtures.npz): per case the inputs of a short goal-conditioned rollout (goal start
poses, episode start poses, shapes, action tape) and the rewards / done flags / deltas the reference's rules give for it.

Rules: the reference's OWN functions - `comp_rew_done` of research/wrappers/body_goal.py:58-88 and cube_goal.py:64-86 are lifted
out of the reference's source files with `ast` at generation time (authoring container only, nothing is copied into this
repository), compiled with numpy + the reference's own `filtlist` (research/utils.py:38, lifted the same way) and called on a
stub `self` (G, _env.pobs_keys, idxs, last_obs); `rew * rew_scale` as body_goal.py:98 / cube_goal.py:59 do.  They are
evaluated on the CPU oracle's observations: the goal is a fresh reset state (the cube variant lets it settle for 10
zero-action env-steps, cube_goal.py:36-37); every compared entry is an x/y position or an LCD pixel, which the HIP path
reproduces bit for bit, so the expected float64 values are exact for the device epilogue.
Run:  python tools/gen_goal_fixtures.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import boxlcd_amd as B
from boxlcd_amd.goal import BodyGoalEnv, CubeGoalEnv
from oracle import pyb2o
import ast, re, types

REF = '/root/reference/research'


def lift(path, name, cls=None, glob=None):
  """Compile ONE function of a reference source file (optionally a method of `cls`) in a namespace of our choosing."""
  tree = ast.parse(open(path).read())
  body = tree.body
  if cls is not None:
    body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
  fn = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
  mod = ast.Module(body=[fn], type_ignores=[])
  ns = dict(glob or {})
  exec(compile(mod, path, 'exec'), ns)
  return ns[name]


_utils = types.SimpleNamespace(filtlist=lift(f'{REF}/utils.py', 'filtlist', glob={'re': re}))
_body_rule = lift(f'{REF}/wrappers/body_goal.py', 'comp_rew_done', 'BodyGoalEnv', {'np': np, 'utils': _utils})
_cube_rule = lift(f'{REF}/wrappers/cube_goal.py', 'comp_rew_done', 'CubeGoalEnv', {'np': np, 'utils': _utils})


class _AttrDict(dict):
  __getattr__ = dict.__getitem__


def ref_rule(kind, obs, last_obs, pobs_keys, idxs, G):
  """(rew * rew_scale, done, info) from the reference's own comp_rew_done"""
  me = types.SimpleNamespace(G=_AttrDict(G), _env=types.SimpleNamespace(pobs_keys=list(pobs_keys)), idxs=idxs, last_obs=last_obs)
  info = {}
  rew, done = (_body_rule if kind == 'body' else _cube_rule)(me, obs, info)
  return rew * G['rew_scale'], done, info

CASES = [('body', 'Urchin', dict(state_rew=1, diff_delt=0, goal_thresh=0.35, rew_scale=0.5)),
         ('body', 'Urchin', dict(state_rew=1, diff_delt=1, goal_thresh=0.35, rew_scale=0.5)),
         ('body', 'Luxo', dict(state_rew=0, diff_delt=0, goal_thresh=0.35, rew_scale=0.5)),
         ('body', 'Crab', dict(state_rew=1, diff_delt=1, goal_thresh=0.35, rew_scale=0.5)),
         ('cube', 'UrchinCube', dict(diff_delt=1, rew_scale=2.0)), ('cube', 'LuxoCube', dict(diff_delt=0, rew_scale=2.0)),
         ('cube', 'UrchinCubes', dict(diff_delt=1, rew_scale=2.0))]
N, T = 32, 12


def obs_of(env, o, goal_fs, goal_lcd):
  fs = o.obs()
  return {'full_state': fs, 'proprio': fs[env.pobs_idxs], 'lcd': o.render().astype(bool), 'goal:full_state': goal_fs,
          'goal:proprio': goal_fs[env.pobs_idxs], 'goal:lcd': goal_lcd.astype(bool)}


def main():
  out = {}
  for ci, (kind, name, G) in enumerate(CASES):
    venv = B.BatchedWorldEnv(name, N, seed=100 + ci)
    wrap = (BodyGoalEnv if kind == 'body' else CubeGoalEnv)(venv, G)
    d = venv.scene.desc
    gposes, gsel = venv.sample_initial(N)
    poses, sel = venv.sample_initial(N)
    acts = venv.sample_actions(T)
    rew = np.zeros((T, N)); done = np.zeros((T, N), bool); delta = np.zeros((T, N))
    for e in range(N):
      g = pyb2o.OracleEnv(d)
      g.reset(gposes[e], gsel[e])
      if kind == 'cube':
        for _ in range(10):
          g.step(np.zeros(venv.act_size, np.float32))
      gfs, glcd = g.obs(), g.render()
      o = pyb2o.OracleEnv(d)
      o.reset(poses[e], sel[e])
      last = obs_of(venv, o, gfs, glcd)
      for t in range(T):
        o.step(acts[t, e])
        cur = obs_of(venv, o, gfs, glcd)
        r, dn, inf = ref_rule(kind, cur, last, venv.pobs_keys, getattr(wrap, 'idxs', None), G)
        rew[t, e], done[t, e], delta[t, e] = r, dn, inf['delta'] if 'delta' in inf else np.nan
        last = cur
    k = f'case{ci}'
    out[k + '_meta'] = np.array([kind, name, repr(G)])
    out[k + '_gposes'], out[k + '_gsel'], out[k + '_poses'], out[k + '_sel'], out[k + '_acts'] = gposes, gsel, poses, sel, acts
    out[k + '_rew'], out[k + '_done'], out[k + '_delta'] = rew, done, delta
    print(k, kind, name, G, 'successes', int(done.sum()))
  np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'goal_fixtures.npz'), **out)


if __name__ == '__main__':
  main()
