import sys, time; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import parity
from boxlcd_amd._lib import Handle
for name in ['Bounce','LuxoBall']:
    env, poses, sel = parity.make_batch(name, 4, 0)
    h = Handle(env.scene.desc, 4, 0); h.reset(None, poses, sel)
    from oracle import pyb2o
    o = pyb2o.OracleEnv(env.scene.desc); o.reset(poses[0], sel[0])
    acts = np.random.RandomState(1).uniform(-1,1,(40,4,env.scene.desc.n_act)).astype(np.float32)
    for t in range(40):
        h.step(acts[t],1); o.step(acts[t][0])
    b,j,p = h.debug_dump(); ob,oj,op = o.dump()
    print(name, 'gpu body0', b[0,0,:8]); print(name,'ora body0', ob[0,:8]); print('pairs gpu', p[0][:, :4].tolist()); print('pairs ora', op[:, :4].tolist()); print('joints', j[0].tolist(), oj.tolist())
    print(o.stats())
for name, n, steps in [('Dropbox',256,200),('Bounce',256,200),('Object2',256,200),('Urchin',128,200),('LuxoBall',128,200)]:
    t0=time.time(); cnt, msgs = parity.run_substep_parity(name, n, steps, seed=3)
    print(name, n, steps, 'world steps', cnt, 'mismatches', len(msgs), 'sec', round(time.time()-t0,1)); [print('   ',m) for m in msgs[:5]]
# timing
for name, n in [('Bounce', 100000), ('Dropbox', 100000), ('Urchin', 50000), ('LuxoBall', 50000), ('Object2', 200000)]:
    env, poses, sel = parity.make_batch(name, n, 0)
    h = Handle(env.scene.desc, n, 0); h.reset(None, poses, sel)
    acts = np.random.RandomState(1).uniform(-1,1,(n,env.scene.desc.n_act)).astype(np.float32)
    h.step(acts, 2)
    t0=time.time(); T=20; h.step(acts, T); dt=time.time()-t0
    ms,_ = h.last_kernel_ms()
    print(name, n, 'env-steps/s', n*T/dt, 'kernel ms', ms, 'faults', int((h.faults()!=0).sum()))
    h.close()
