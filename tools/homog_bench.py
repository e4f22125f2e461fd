import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
name, N, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]); G = int(sys.argv[4])   # G distinct envs, replicated
venv = B.BatchedWorldEnv(name, N, seed=1000)
d = venv.scene.desc
h = Handle(d, N, 0)
poses, sel = venv.sample_initial(N)
acts = venv.sample_actions(T)
if G > 0:
  # groups of 64 consecutive slots hold copies of one env -> every wave is perfectly homogeneous
  idx = (np.arange(N) // 64) % G
  poses, sel, acts = poses[idx], sel[idx], acts[:, idx]
dev = torch.device('cuda', 0)
poses_t, sel_t, acts_t = torch.as_tensor(poses).to(dev), torch.as_tensor(sel).to(dev), torch.as_tensor(np.ascontiguousarray(acts)).to(dev)
lcd = torch.empty((T, N, d.lcd_h, d.lcd_w), dtype=torch.uint8, device=dev); obs = torch.empty((T, N, d.n_obs), dtype=torch.float32, device=dev)
def roll():
  h.reset(None, poses_t, sel_t); h.rollout(acts_t, T, lcd, obs); return h.last_kernel_ms()
roll(); m, n = roll()
print(f'{name} N={N} T={T} groups={G}: {m/n:.2f} ms/launch ({n} launches)')
