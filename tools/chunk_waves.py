"""Per-wave times of the first chunks of a rollout from reset (needs the -DBLCD_WAVETIMES variant: BLCD_LIB=libboxlcd_hip_wt.so).
usage: BLCD_LIB=libboxlcd_hip_wt.so python tools/chunk_waves.py Dropbox 100000 50 [chunks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['BLCD_WAVETIMES'] = '1'
import numpy as np
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
name, n, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
chunks = int(sys.argv[4]) if len(sys.argv) > 4 else 2
os.environ['BLCD_CHUNK'] = str(chunk)
env = B.BatchedWorldEnv(name, n, seed=1000)
poses, sel = env.sample_initial(n)
h = Handle(env.scene.desc, n, 0)
for rep in range(2):
  h.reset(None, poses, sel)
  h.debug_wave_times()
  for c in range(chunks):
    h.rollout(None, chunk, None, None)
    wt = h.debug_wave_times().astype(np.float64)
    if rep == 0:
      continue
    tot = wt[:, 0] * 10e-6
    ms = h.last_kernel_ms()[0]
    q = np.percentile(tot, [10, 50, 90, 99])
    print(f'{name} chunk {c} ({chunk} steps): kernel {ms:.3f} ms; waves {len(tot)}; sum/1024 {tot.sum() / 1024:.3f} ms; mean {tot.mean():.3f} p10/50/90/99 {q.round(3).tolist()} max {tot.max():.3f}')
    k = len(tot) // 8
    print('   mean wave ms by eighth of the block order:', [round(float(tot[i * k:(i + 1) * k].mean()), 3) for i in range(8)])
    print('   mean kcycles collide/solve/toi/toi-event', (wt[:, 1:5].mean(0) / 1e3).round(1), ' max-wave', (wt[np.argmax(tot), 1:5] / 1e3).round(1))
    # 5 #toi calls (lane max) 6 #events (lane max) | 7 cycles inside the full TOI routine, 8 wave-level executions of it (wave sums)
    print('   mean per wave: TOI calls (lane max) %.1f, events (lane max) %.1f, full-routine executions %.1f, kcycles inside them %.1f' % (wt[:, 5].mean(), wt[:, 6].mean(), wt[:, 8].mean(), wt[:, 7].mean() / 1e3))
    if os.environ.get('CW_RAW'):   # diagnostic builds that give the eight slots other meanings (-DBLCD_PROF_TOI: 0 init 1 phase 1 2 all of SolveTOI 3 events 4 phase 2 5 event rounds' rest)
      print('   raw slot means (k):', (wt[:, 1:9].mean(0) / 1e3).round(1))
