"""One line per (env, batch): env-steps/s and mean step_kernel launch time of a fused rollout.  usage: python tools/quick_bench.py Urchin 50000 [T] [rollouts]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
name, N = sys.argv[1], int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
R = int(sys.argv[4]) if len(sys.argv) > 4 else 2
venv = B.BatchedWorldEnv(name, N, seed=1000)
d = venv.scene.desc
h = Handle(d, N, 0)
poses, sel = venv.sample_initial(N)
dev = torch.device('cuda', 0)
poses_t, sel_t = torch.as_tensor(poses).to(dev), torch.as_tensor(sel).to(dev)
acts = torch.as_tensor(venv.sample_actions(T)).to(dev)
lcd = torch.empty((T, N, d.lcd_h, d.lcd_w), dtype=torch.uint8, device=dev)
obs = torch.empty((T, N, d.n_obs), dtype=torch.float32, device=dev)
NORESET = bool(os.environ.get('QB_NORESET'))   # timed rollouts continue from the warm-up rollout's final state (e.g. a batch at rest)
def roll(reset=True):
  if reset: h.reset(None, poses_t, sel_t)
  h.rollout(acts, T, lcd, obs); return h.last_kernel_ms()
roll(); torch.cuda.synchronize()
t0 = time.perf_counter(); ms = 0.0; nl = 0
for _ in range(R):
  m, n = roll(not NORESET); ms += m; nl += n
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'{name} N={N} T={T} lanes={os.environ.get("BLCD_LANES","64")} chunk={os.environ.get("BLCD_CHUNK","20")}: {R*T*N/dt:.4g} env-steps/s, {ms/nl:.3f} ms/launch x {nl//R} launches/rollout, faults {int((h.faults()!=0).sum())}')
