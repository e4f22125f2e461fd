"""Per-phase shader cycles (collide / solve / TOI) of a fused chunk, for a mixed batch and for a batch whose waves hold 64 copies
of one environment (the bound a perfect re-binning would reach).  usage: BLCD_WAVETIMES=1 python tools/phase_breakdown.py Urchin 50000 20"""
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
for label, G in (('mixed', 0), ('64 copies per wave', 1)):
  p, s, a = poses, sel, acts
  if G:
    idx = (np.arange(N) // 64)
    p, s, a = poses[idx], sel[idx], np.ascontiguousarray(acts[:, idx])
  h = Handle(d, N, 0)
  h.reset(None, p, s)
  h.rollout(a[:T], T)
  h.debug_wave_times()
  h.rollout(a[T:], T)
  wt = h.debug_wave_times().astype(np.float64)
  ms = h.last_kernel_ms()[0]
  print(f'{name} {label}: kernel {ms:.2f} ms / {T} env-steps; mean kcycles per wave: collide {wt[:,1].mean()/1e3:.0f} solve {wt[:,2].mean()/1e3:.0f} toi {wt[:,3].mean()/1e3:.0f} (toi-event {wt[:,4].mean()/1e3:.0f}) [prof4 {wt[:,5].mean()/1e3:.0f} prof5 {wt[:,6].mean()/1e3:.0f} toi-routine {wt[:,7].mean()/1e3:.0f} kcyc in {wt[:,8].mean():.0f} wave-level runs]; wave ticks mean {wt[:,0].mean()*10e-6:.2f} ms max {wt[:,0].max()*10e-6:.2f} ms')
  h.close()
