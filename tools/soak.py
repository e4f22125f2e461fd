"""Soak test (GPU box): every env of the catalogue, fused 200-step rollouts on a few thousand environments, final body state +
last LCD frame against the CPU oracle.  Larger and longer than tests/test_gpu_parity.py; prints one line per env."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
from oracle import pyb2o

T = int(os.environ.get('SOAK_T', 200))
bad_total = 0
only = [x for x in os.environ.get('SOAK_ENVS', '').split(',') if x]
for name in (only or sorted(B.env_map)):
  env1 = B.env_map[name]()
  nb = env1.scene.desc.n_bodies
  n = int(os.environ.get('SOAK_N', 0)) or (4096 if nb <= 2 else 1024 if nb <= 7 else 256)
  env = B.BatchedWorldEnv(name, n, seed=123)
  poses, sel = env.sample_initial(n)
  acts = env.sample_actions(T)
  h = Handle(env.scene.desc, n, 0)
  h.reset(None, poses, sel)
  lcd = np.zeros((T, n, env.scene.desc.lcd_h, env.scene.desc.lcd_w), np.uint8)
  t0 = time.time()
  h.rollout(acts, T, lcd, None)
  tg = time.time() - t0
  state = h.debug_dump()[0]
  faults = h.faults()
  t0 = time.time()
  _, _, olcd, ost = pyb2o.rollout(env.scene.desc, poses, sel, acts, T, threads=16)
  to = time.time() - t0
  bad_env = int((~(state == ost).reshape(n, -1).all(1)).sum())
  bad_lcd = int((~(lcd[-1] == olcd).reshape(n, -1).all(1)).sum())
  bad_total += bad_env + bad_lcd + int((faults != 0).sum())
  print(f'{name:12s} n={n:5d} T={T}: state mismatches {bad_env}, lcd mismatches {bad_lcd}, faults {int((faults != 0).sum())} '
        f'(gpu {tg:.2f} s, oracle {to:.1f} s)', flush=True)
  h.close()
print('SOAK', 'OK' if bad_total == 0 else f'FAILED ({bad_total})')
sys.exit(0 if bad_total == 0 else 1)
