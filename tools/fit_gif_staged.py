"""Staged fit of a two-object GIF: (1) grid-search each object's start pose on the free-fall frames (objects do not interact
before first contact), (2) joint refinement of the survivors on all frames.  One-off tool."""
import sys, itertools
sys.path.insert(0, '.')
import numpy as np
import boxlcd_amd as B
from oracle import pyb2o

def main(key, sel, guess, pre, ang_ranges):
  gif = np.unpackbits(np.load('tests/golden/gif_lcd_frames.npz')[key], axis=-1)[:, :, :16]
  env = B.envs.Object2()
  T = len(gif)
  def rollout(p, upto=T):
    o = pyb2o.OracleEnv(env.scene.desc)
    o.reset(np.array([[p[0], p[1], p[2]], [p[3], p[4], p[5]]], np.float32), sel)
    per = []
    for t in range(upto):
      o.step(None)
      per.append(int((o.render() != gif[t]).sum()))
    return per
  # stage 1: per-object grids on the first `pre` frames
  cands = []
  for obj in (0, 1):
    lst = []
    xs = np.arange(guess[3 * obj] - 0.05, guess[3 * obj] + 0.0501, 0.01)
    ys = np.arange(guess[3 * obj + 1] - 0.05, guess[3 * obj + 1] + 0.0501, 0.01)
    angs = ang_ranges[obj]
    for x, y, a in itertools.product(xs, ys, angs):
      p = list(guess)
      p[3 * obj:3 * obj + 3] = [x, y, a]
      per = rollout(p, pre)
      lst.append((sum(per), x, y, a))
    lst.sort()
    best = lst[0][0]
    keep = [l for l in lst if l[0] <= best][:400]
    print('object', obj, 'best pre-contact mismatch', best, 'survivors', len(keep), 'e.g.', [round(v, 3) for v in keep[0][1:]])
    cands.append(keep)
  # stage 2: joint random search over survivors + jitter
  rng = np.random.RandomState(0)
  best = (10**9, None)
  for it in range(6000):
    a = cands[0][rng.randint(len(cands[0]))]
    b = cands[1][rng.randint(len(cands[1]))]
    p = np.array([a[1], a[2], a[3], b[1], b[2], b[3]]) + rng.uniform(-0.005, 0.005, 6)
    s = sum(rollout(p))
    if s < best[0]:
      best = (s, p)
      print('  it', it, 'mismatch', s, np.round(p, 4).tolist())
      if s == 0: break
  s, x = best
  for it in range(3000):
    if s == 0: break
    cand = x + rng.normal(0, 0.002, 6)
    s2 = sum(rollout(cand))
    if s2 < s: s, x = s2, cand; print('  refine', it, s, np.round(x, 5).tolist())
  print('RESULT', key, sel, s, [float(v) for v in np.round(x, 5)])

if __name__ == '__main__':
  which = sys.argv[1]
  if which == 'mixed':
    main('Object2', [1, 0], [1.604, 4.176, 1.295, 2.48, 3.014, 0.0], 6,
         [np.concatenate([np.arange(1.25, 1.345, 0.005), np.arange(1.25, 1.345, 0.005) - np.pi / 2]), [0.0]])
  else:
    main('Object2_cubes', [1, 1], [1.895, 4.42, 1.375, 0.874, 2.342, 0.59], 4,
         [np.concatenate([np.arange(1.33, 1.42, 0.005), np.arange(1.33, 1.42, 0.005) - np.pi / 2]),
          np.concatenate([np.arange(0.545, 0.635, 0.005), np.arange(0.545, 0.635, 0.005) + np.pi / 2])])
