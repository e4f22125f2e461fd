"""Where a wave's TOI phase goes (BLCD_PROF_TOI + BLCD_WAVETIMES builds of the class): shader cycles per wave per T env-steps in
0 per-step reset of the TOI bookkeeping | 1 phase 1 (pending scan, early-outs, the routine) | 4 phase 2 (minimum over the list) |
5 event handling (advance, sub-step island, re-synchronisation).  usage: python tools/toi_phases.py Bounce 100000 20"""
import os, sys
os.environ['BLCD_WAVETIMES'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
name, N, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
venv = B.BatchedWorldEnv(name, N, seed=1000)
d = venv.scene.desc
poses, sel = venv.sample_initial(N)
acts = venv.sample_actions(2 * T)
h = Handle(d, N, 0)
h.reset(None, poses, sel)
h.rollout(acts[:T], T)
h.debug_wave_times()
h.rollout(acts[T:], T)
wt = h.debug_wave_times().astype(np.float64)
ms = h.last_kernel_ms()[0]
m = wt.mean(0) / 1e3
print(f'{name}: kernel {ms:.2f} ms / {T} env-steps; mean kcycles per wave: reset {m[1]:.0f} phase1 {m[2]:.0f} [toi all {m[3]:.0f}] ? {m[4]:.0f} phase2 {m[5]:.0f} events {m[6]:.0f}; '
      f'routine {m[7]:.0f} kcyc in {wt[:,8].mean():.0f} wave-level runs; wave ms mean {wt[:,0].mean()*10e-6:.2f} max {wt[:,0].max()*10e-6:.2f}')
