"""Fit two-object GIFs containing boxes using the 8x RGB half: per-frame centres AND box orientations (min-area rectangle,
modulo pi/2) constrain a Nelder-Mead fit of the start poses; success = LCD half reproduced bit for bit.  One-off tool."""
import sys
sys.path.insert(0, '.')
import numpy as np
from PIL import Image, ImageSequence
from scipy import ndimage, optimize
import boxlcd_amd as B
from oracle import pyb2o

def measure(name):
  im = Image.open(f'/root/reference/assets/envs/{name}.gif')
  res = []
  for fr in ImageSequence.Iterator(im):
    f = np.asarray(fr.convert('RGB'))[:, :128].astype(int)
    fill = (np.abs(f - np.array([128, 102, 230])).sum(-1) < 40) | (np.abs(f - np.array([77, 77, 128])).sum(-1) < 40)
    lab, n = ndimage.label(fill)
    row = []
    for k in range(1, n + 1):
      ys, xs = np.nonzero(lab == k)
      if len(ys) < 80: continue
      pts = np.stack([xs + 0.5, 128 - ys - 0.5], -1) / 25.6
      best = None
      for a in np.arange(0, np.pi / 2, 0.01):
        c, s_ = np.cos(a), np.sin(a)
        u = pts[:, 0] * c + pts[:, 1] * s_; v = -pts[:, 0] * s_ + pts[:, 1] * c
        area = (u.max() - u.min()) * (v.max() - v.min())
        if best is None or area < best[0]: best = (area, a)
      row.append((pts[:, 0].mean(), pts[:, 1].mean(), best[1], len(ys)))
    res.append(row)
  return res

def angdiff(a, b):   # difference modulo pi/2
  d = (a - b) % (np.pi / 2)
  return min(d, np.pi / 2 - d)

def main(key, sel, x0s):
  gif = np.unpackbits(np.load('tests/golden/gif_lcd_frames.npz')[key], axis=-1)[:, :, :16]
  meas = measure(key.replace('_', '-'))
  env = B.envs.Object2()
  T = len(gif)
  isbox = [s == 1 for s in sel]
  def rollout(p):
    o = pyb2o.OracleEnv(env.scene.desc)
    o.reset(np.array([[p[0], p[1], p[2]], [p[3], p[4], p[5]]], np.float32), sel)
    traj, bad = [], 0
    for t in range(T):
      o.step(None)
      traj.append(o.dump()[0][:, :3].copy())
      bad += int((o.render() != gif[t]).sum())
    return np.array(traj), bad
  def cost(p, upto=T):
    traj, bad = rollout(p)
    err = 0.0
    for t in range(upto):
      if len(meas[t]) != 2: continue
      m = meas[t]
      def pair(i, j):
        e = (traj[t][0, 0] - m[i][0])**2 + (traj[t][0, 1] - m[i][1])**2 + (traj[t][1, 0] - m[j][0])**2 + (traj[t][1, 1] - m[j][1])**2
        if isbox[0]: e += 0.05 * angdiff(traj[t][0, 2], m[i][2])**2
        if isbox[1]: e += 0.05 * angdiff(traj[t][1, 2], m[j][2])**2
        return e
      err += min(pair(0, 1), pair(1, 0))
    return err + 1e-4 * bad
  best = (10**9, None)
  for k, x0 in enumerate(x0s):
    for upto in (10, 20, T):   # progressively longer horizons
      r = optimize.minimize(lambda p: cost(p, upto), x0, method='Nelder-Mead', options=dict(xatol=2e-5, fatol=1e-8, maxiter=900, initial_simplex=None))
      x0 = r.x
    traj, bad = rollout(x0)
    print(key, 'start', k, 'x', np.round(x0, 4).tolist(), 'cost', round(r.fun, 5), 'LCD mismatched px', bad, flush=True)
    if bad < best[0]: best = (bad, x0)
    if bad == 0: break
  bad, x = best
  rng = np.random.RandomState(1)
  for it in range(4000):
    if bad == 0: break
    cand = x + rng.normal(0, 0.0015, 6)
    _, b2 = rollout(cand)
    if b2 < bad: bad, x = b2, cand; print('  refine', it, bad, np.round(x, 5).tolist(), flush=True)
  print('RESULT', key, sel, bad, [float(v) for v in np.round(x, 5)])

if __name__ == '__main__':
  w = sys.argv[1]
  if w == 'mixed':
    main('Object2', [1, 0], [[1.604, 4.176, 1.295, 2.48, 3.014, 0.0], [1.604, 4.176, 1.295 - np.pi / 2, 2.48, 3.014, 0.0]])
  elif w == 'cubes':
    main('Object2_cubes', [1, 1], [[1.895, 4.42, 1.375, 0.874, 2.342, 0.59], [1.895, 4.42, 1.375 - np.pi / 2, 0.874, 2.342, 0.59],
                                    [0.874, 2.342, 0.59, 1.895, 4.42, 1.375], [0.874, 2.342, 0.59, 1.895, 4.42, 1.375 - np.pi / 2]])
