"""BLCD_PROF_TOI build only: cycles per part of solveTOI (0 reset, 1 phase-1 scan + TOI routine, 4 phase-2 minimum, 5 event part)."""
import os, sys
os.environ['BLCD_WAVETIMES'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
name, N, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
venv = B.BatchedWorldEnv(name, N, seed=1000)
poses, sel = venv.sample_initial(N); acts = venv.sample_actions(2 * T)
h = Handle(venv.scene.desc, N, 0); h.reset(None, poses, sel); h.rollout(acts[:T], T); h.debug_wave_times(); h.rollout(acts[T:], T)
wt = h.debug_wave_times().astype(np.float64)
print(name, 'kernel ms', h.last_kernel_ms()[0], 'mean kcycles per wave: prof[0..7] =', (wt[:, 1:9].mean(0) / 1e3).round(0).tolist())
