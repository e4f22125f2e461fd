"""Fit the initial poses of the reference's two-object GIFs (Bounce2, Object2-circles) so that the CPU oracle reproduces their
LCD frames.  The 8x RGB half of each GIF frame gives object centres to ~0.02 world units; a Nelder-Mead fit of the initial
poses against that trajectory is then checked against the LCD half bit for bit.  (one-off tool; results are constants in
tests/test_oracle_physics.py)"""
import sys
sys.path.insert(0, '.')
import numpy as np
from PIL import Image, ImageSequence
from scipy import ndimage, optimize
import boxlcd_amd as B
from oracle import pyb2o

def centres(name):
  im = Image.open(f'/root/reference/assets/envs/{name}.gif')
  out = []
  for fr in ImageSequence.Iterator(im):
    f = np.asarray(fr.convert('RGB'))[:, :128].astype(int)
    fill = (np.abs(f - np.array([128, 102, 230])).sum(-1) < 40) | (np.abs(f - np.array([77, 77, 128])).sum(-1) < 40)
    lab, n = ndimage.label(fill)
    cs = []
    for k in range(1, n + 1):
      ys, xs = np.nonzero(lab == k)
      if len(ys) > 80: cs.append(((xs.mean() + 0.5) / 25.6, (128 - ys.mean() - 0.5) / 25.6, len(ys)))
    out.append(cs)
  return out

def run(env_name, gif_key, sel, order, with_angle=False, angle_starts=((0.0, 0.0),), max_trials=6):
  gif = np.unpackbits(np.load('tests/golden/gif_lcd_frames.npz')[gif_key], axis=-1)[:, :, :16]
  cs = centres(gif_key.replace('_', '-'))
  env = getattr(B.envs, env_name)()
  T = len(gif)
  def rollout(p):
    o = pyb2o.OracleEnv(env.scene.desc)
    poses = np.array([[p[0], p[1], p[4] if with_angle else 0.0], [p[2], p[3], p[5] if with_angle else 0.0]], np.float32)
    o.reset(poses, sel)
    traj, bad = [], 0
    for t in range(T):
      o.step(np.zeros(1, np.float32))
      b = o.dump()[0]
      traj.append(b[:, :2].copy())
      bad += int((o.render() != gif[t]).sum())
    return np.array(traj), bad
  # target trajectory: match components to bodies by nearest neighbour from the frame-0 guess
  c0 = cs[0]
  guess = [c0[order[0]][0], c0[order[0]][1] + 0.0654, c0[order[1]][0], c0[order[1]][1] + 0.0654]
  def cost(p):
    traj, bad = rollout(p)
    err = 0.0
    for t in range(T):
      if len(cs[t]) != 2: continue      # merged blobs while touching
      pts = np.array([[c[0], c[1]] for c in cs[t]])
      d = np.linalg.norm(traj[t][:, None, :] - pts[None], axis=-1)
      err += min(d[0, 0] + d[1, 1], d[0, 1] + d[1, 0]) ** 2
    return err + 1e-3 * bad
  best = None
  nd = 6 if with_angle else 4
  trial = -1
  for astart in angle_starts:
   for rep in range(max_trials):
    trial += 1
    x0 = np.array(guess + (list(astart) if with_angle else []))
    if rep: x0[:4] += np.random.RandomState(trial).uniform(-0.02, 0.02, 4)
    if rep and with_angle: x0[4:] += np.random.RandomState(trial + 99).uniform(-0.03, 0.03, 2)
    r = optimize.minimize(cost, x0, method='Nelder-Mead', options=dict(xatol=1e-4, fatol=1e-6, maxiter=600 if with_angle else 400))
    traj, bad = rollout(r.x)
    print(env_name, gif_key, 'trial', trial, 'x', np.round(r.x, 4), 'cost', round(r.fun, 5), 'LCD mismatched px', bad)
    if best is None or bad < best[0]: best = (bad, r.x)
    if bad == 0: break
   if best[0] == 0: break
  # local random refinement on the pixel objective
  rng = np.random.RandomState(0)
  bad, x = best
  for it in range(1500):
    if bad == 0: break
    cand = x + rng.normal(0, 0.004, nd) * (np.array([1, 1, 1, 1, 5, 5][:nd]))
    _, b2 = rollout(cand)
    if b2 < bad: bad, x = b2, cand; print('  refine', it, bad, np.round(x, 4))
  print('RESULT', env_name, gif_key, 'sel', sel, 'order', order, 'bad px', bad, 'x', [float(v) for v in np.round(x, 5)])

if __name__ == '__main__':
  which = sys.argv[1] if len(sys.argv) > 1 else 'Bounce2'
  if which == 'Bounce2':
    run('Bounce2', 'Bounce2', [0, 0], (0, 1))
  elif which == 'circles':
    run('Object2', 'Object2_circles', [0, 0], (0, 1))
    run('Object2', 'Object2_circles', [0, 0], (1, 0))
  elif which == 'cubes':      # orientations measured from the RGB half (min-area rectangle), modulo pi/2
    run('Object2', 'Object2_cubes', [1, 1], (0, 1), True, [(1.375, 0.59), (1.375 - np.pi / 2, 0.59)], 8)
  elif which == 'cubes_swapped':
    run('Object2', 'Object2_cubes', [1, 1], (1, 0), True, [(0.59, 1.375), (0.59, 1.375 - np.pi / 2)], 8)
  elif which == 'mixed':
    run('Object2', 'Object2', [1, 0], (0, 1), True, [(1.295, 0.0), (1.295 - np.pi / 2, 0.0)], 8)
