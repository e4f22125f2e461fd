import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
name, N, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
venv = B.BatchedWorldEnv(name, N, seed=1000)
d = venv.scene.desc
h = Handle(d, N, 0)
poses, sel = venv.sample_initial(N)
h.reset(None, poses, sel)
acts = venv.sample_actions(40 + T)
h.rollout(acts[:40], 40)      # settle a bit (untraced interest)
torch.cuda.synchronize()
h.sched_stats()
import time
t0 = time.perf_counter()
h.rollout(np.ascontiguousarray(acts[40:]), T)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
s = h.sched_stats()
print(f'{name} N={N} T={T} chunk={os.environ.get("BLCD_CHUNK")} passes={s["passes"]} lanes<={s["max_lanes"]}: {N*T/dt:.3g} env-steps/s; first passes: {s["first_live"]} lanes in {s["first_waves"]} waves, {s["first_suspended"]} suspended ({100.0*s["first_suspended"]/max(1,s["first_live"]):.1f} %); later passes: {s["later_live"]} lanes in {s["later_waves"]} waves ({s["later_live"]/max(1,s["later_waves"]):.1f} per wave), {s["later_suspended"]} suspended again')
