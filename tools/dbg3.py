import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, parity
from boxlcd_amd._lib import Handle
for name in ['Bounce', 'Dropbox']:
  for n in [1, 64, 4096, 100000]:
    env, poses, sel = parity.make_batch(name, n, 0)
    h = Handle(env.scene.desc, n, 0); h.reset(None, poses, sel)
    ts = []
    for t in range(200):
        h.step(None, 1); ts.append(h.last_kernel_ms()[0])
    ts = np.array(ts)
    print(name, 'N', n, 'per-step kernel ms: mean', ts.mean().round(4), 'max', ts.max().round(4), 'min', ts.min().round(4), 'p50', np.median(ts).round(4), 'first10', ts[:10].round(3))
    h.close()
