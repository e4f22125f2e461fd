"""What would straggler compaction buy?  Times a fused rollout with the position / velocity iteration caps lowered (results are
then NOT the reference's - this is a bound, not a mode): the difference to the real caps is what the slowest lanes cost today.
usage: python tools/sched_bound.py Urchin 50000 [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
name, N = sys.argv[1], int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
venv = B.BatchedWorldEnv(name, N, seed=1000)
poses, sel = venv.sample_initial(N)
dev = torch.device('cuda', 0)
poses_t, sel_t = torch.as_tensor(poses).to(dev), torch.as_tensor(sel).to(dev)
acts = torch.as_tensor(venv.sample_actions(T)).to(dev)
for vi, pi in ((180, 60), (180, 8), (180, 4), (180, 0), (16, 60), (16, 8), (8, 8)):
  d = venv.scene.desc
  d.vel_iters, d.pos_iters = vi, pi
  h = Handle(d, N, 0)
  def roll():
    h.reset(None, poses_t, sel_t); h.rollout(acts, T); return h.last_kernel_ms()
  roll(); torch.cuda.synchronize()
  t0 = time.perf_counter(); roll(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
  print(f'{name} N={N} T={T} vel_iters={vi} pos_iters={pi}: {T*N/dt:.4g} env-steps/s', flush=True)
  h.close()
