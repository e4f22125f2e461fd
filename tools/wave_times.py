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

# fused-rollout view: per 50-step chunk, how far is the launch from a perfectly balanced schedule?
import torch
h2 = Handle(env.scene.desc, n, 0); h2.reset(None, poses, sel)
h2.debug_wave_times()
for chunk in range(4):
    h2.rollout(None, 50, None, None)
    wt = h2.debug_wave_times().astype(np.float64)
    tot = wt[:, 0] * 10e-6
    ms = h2.last_kernel_ms()[0]
    srt = np.sort(tot)[::-1]
    print(f'chunk {chunk}: kernel {ms:.3f} ms; waves {len(tot)}; sum/1024 {tot.sum()/1024:.3f} ms; max {tot.max():.3f}; p50 {np.median(tot):.3f}; '
          f'top-1024 mean {srt[:1024].mean():.3f}; first 8 waves {tot[:8].round(3).tolist()} last 4 {tot[-4:].round(3).tolist()}')
    print('   mean kcycles collide/solve/toi/toi-event', (wt[:, 1:5].mean(0) / 1e3).round(1), ' max-wave', (wt[np.argmax(tot), 1:5] / 1e3).round(1),
          ' TOI routine: mean kcycles/wave', (wt[:, 7].mean() / 1e3).round(1), 'wave-level executions/wave', wt[:, 8].mean().round(1), 'lane-max calls', wt[:, 5].mean().round(1))

# wave time (last launch of the rollout) by position in the sorted order
tot = wt[:, 0] * 10e-6
nz = tot[:len(tot)]
print('wave ms by position (every 64th):', nz[::64].round(3).tolist())
