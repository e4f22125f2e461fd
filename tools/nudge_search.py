"""Diagnostic: where does a replay depart from a recording?  Re-run the seed-7 replay of one GIF with a 1-ulp nudge of one
state scalar at the start of env-step k and report how many further 8x RGB frames become exact.  (Authoring container only.)
usage: python tools/nudge_search.py Urchin 0 20 60"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import replay as R
import boxlcd_amd as B
from oracle import pyb2o

name, k0, k1, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ULPS = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else (-1, 1)
env = getattr(B.envs, name)(raster_variant=2)
rgb, lcd = R.load_gif(name)
P, sel = R.seed7_start(env)
acts = np.random.RandomState(4).uniform(-1, 1, (len(lcd), env.act_size)).astype(np.float32)

def run(nudge=None):
  o = pyb2o.OracleEnv(env.scene.desc)
  o.reset(np.asarray(P, np.float32), sel)
  bad = []
  for t in range(T):
    if nudge is not None and nudge[0] == t:
      o.nudge(*nudge[1:])
    o.step(acts[t])
    bad.append(int((R.pil_rgb(env, o) != rgb[t]).any(-1).sum()))
  return bad

base = run()
print('base', sum(b == 0 for b in base), [(i, b) for i, b in enumerate(base) if b][:12])
nb = len(env.scene.bodies)
best = []
for k in range(k0, k1):
  for body in range(nb):
    for field in range(6):
      for ulps in ULPS:
        bad = run((k, body, field, ulps))
        exact = sum(b == 0 for b in bad)
        best.append((exact, sum(bad), k, body, field, ulps))
  best.sort(key=lambda r: (-r[0], r[1]))
  print('after k', k, 'top', best[:5], flush=True)
