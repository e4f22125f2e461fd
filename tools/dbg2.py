import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, parity
from boxlcd_amd._lib import Handle
name, n = 'Bounce', 100000
for vi in [180, 20, 1]:
    env, poses, sel = parity.make_batch(name, n, 0)
    env.scene.desc.vel_iters = vi
    h = Handle(env.scene.desc, n, 0); h.reset(None, poses, sel)
    ts = []
    for t in range(200):
        h.step(None, 1); ts.append(h.last_kernel_ms()[0])
    ts = np.array(ts)
    print('vel_iters', vi, 'total ms', ts.sum(), 'per-step ms at t=0,10,20,40,80,120,199:', ts[[0,10,20,40,80,120,199]].round(2), 'awake', h.get_poses()[:,:,3].mean())
    h.close()
