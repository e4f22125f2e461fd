import sys, os; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ['BLCD_WAVETIMES'] = '1'
import numpy as np, parity
from boxlcd_amd._lib import Handle
name, n = sys.argv[1], int(sys.argv[2])
env, poses, sel = parity.make_batch(name, n, 0)
h = Handle(env.scene.desc, n, 0); h.reset(None, poses, sel)
h.debug_wave_times()
for t in range(200):
    h.step(None, 1)
    wt = h.debug_wave_times().astype(np.float64)
    if t in (1, 5, 10, 20, 40, 80, 150):
        ms = h.last_kernel_ms()[0]
        tot = wt[:, 0] * 10e-6
        i = np.argsort(tot)[-3:]
        print(name, 't', t, 'kernel ms', round(ms, 3), 'wave ms mean', tot.mean().round(4), 'max', tot.max().round(4))
        for j in i:
            cyc = wt[j, 1:5]
            print('    slow wave', j, 'ms', tot[j].round(4), 'kcycles collide/solve/toi(all)/toi-event', (cyc/1e3).round(1), 'lane-max #toiCalls', wt[j,5], '#events', wt[j,6], 'sweeps', wt[j,7], 'posIters', wt[j,8])
        print('    mean over waves kcycles', (wt[:,1:5].mean(0)/1e3).round(1), 'mean lane-max calls/events/sweeps/pos', wt[:,5:].mean(0).round(2))
