"""Every-frame soak (GPU box): every env of the catalogue, fused 200-step rollouts, EVERY LCD frame and observation row of every
environment and step + the final state against the CPU oracle (tests/test_gpu_parity.py does this with 256 environments per class;
this is the larger run behind profiles/r04_soak_every_frame.log).  usage: SOAK_N=2048 python tools/soak_frames.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
from oracle import pyb2o

T = int(os.environ.get('SOAK_T', 200))
n = int(os.environ.get('SOAK_N', 2048))
bad_total = 0
for name in sorted(B.env_map):
  env = B.BatchedWorldEnv(name, n, seed=321)
  poses, sel = env.sample_initial(n)
  acts = env.sample_actions(T)
  d = env.scene.desc
  h = Handle(d, n, 0)
  h.reset(None, poses, sel)
  lcd = np.zeros((T, n, d.lcd_h, d.lcd_w), np.uint8)
  obs = np.zeros((T, n, d.n_obs), np.float32)
  h.rollout(acts, T, lcd, obs)
  state = h.debug_dump()[0]
  faults = int((h.faults() != 0).sum())
  h.close()
  t0 = time.time()
  _, oobs, olcd, ost = pyb2o.rollout_frames(d, poses, sel, acts, T, threads=16)
  bad_frames = int((~(lcd == olcd).reshape(T * n, -1).all(1)).sum())
  bad_state = int((~(state == ost).reshape(n, -1).all(1)).sum())
  err = float(np.abs(obs - oobs).max())
  bad_total += bad_frames + bad_state + faults + (err >= 1e-6)
  print(f'{name:12s} n={n} T={T}: frame mismatches {bad_frames} of {T * n}, state mismatches {bad_state}, max |obs diff| {err:.2e}, faults {faults} (oracle {time.time() - t0:.1f} s)', flush=True)
print('SOAK', 'OK' if bad_total == 0 else f'FAILED ({bad_total})')
sys.exit(0 if bad_total == 0 else 1)
