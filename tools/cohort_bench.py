"""Experiment behind DESIGN.md 8 item 1: the same batch as K independent cohorts (one handle + one private stream + one host
thread each), so that no launch has to wait for the slowest wave of the WHOLE batch.  usage: python tools/cohort_bench.py Bounce 100000 2"""
import os, sys, time, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
name = sys.argv[1]; N = int(sys.argv[2]); K = int(sys.argv[3]); T = 200; R = 4
os.environ['BLCD_PRIVATE_STREAM'] = '1'
if K > 1:
  os.environ['BLCD_REBIN'] = '1'; os.environ['BLCD_LANES'] = '64'
dev = torch.device('cuda', 0)
hs = []
n = N // K
for k in range(K):
  venv = B.BatchedWorldEnv(name, n, seed=1000 + k)
  d = venv.scene.desc
  poses, sel = venv.sample_initial(n)
  h = Handle(d, n, 0)
  hs.append((h, torch.as_tensor(poses).to(dev), torch.as_tensor(sel).to(dev),
             torch.empty((T, n, d.lcd_h, d.lcd_w), dtype=torch.uint8, device=dev), torch.empty((T, n, d.n_obs), dtype=torch.float32, device=dev)))
def work(item, reps):
  h, p, s, lcd, obs = item
  for _ in range(reps):
    h.reset(None, p, s); h.rollout(None, T, lcd, obs)
def run(reps):
  th = [threading.Thread(target=work, args=(it, reps)) for it in hs]
  [t.start() for t in th]; [t.join() for t in th]
run(1); torch.cuda.synchronize()
t0 = time.perf_counter(); run(R); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'{name} N={N} cohorts={K}: {R*T*N/dt:.4g} env-steps/s')
