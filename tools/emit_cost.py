"""How much of a fused rollout is observation / LCD emission?  (diagnostic; GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
name, n, T = sys.argv[1], int(sys.argv[2]), 200
env = B.BatchedWorldEnv(name, n, seed=1)
poses, sel = env.sample_initial(n)
d = env.scene.desc
h = Handle(d, n, 0)
lcd = torch.empty((T, n, d.lcd_h, d.lcd_w), dtype=torch.uint8, device='cuda')
obs = torch.empty((T, n, d.n_obs), dtype=torch.float32, device='cuda')
for label, l, o in [('lcd+obs', lcd, obs), ('lcd only', lcd, None), ('obs only', None, obs), ('neither', None, None)]:
  ts = []
  for rep in range(3):
    h.reset(None, poses, sel)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.rollout(None, T, l, o)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
  print(f'{name} {label:9s}: {min(ts) * 1e3:.2f} ms per {T}-step rollout ({n * T / min(ts):.4g} env-steps/s)')
