"""Shared helpers for the parity tests: drive the HIP product (through the C-ABI) and the CPU oracle on the same
seeded inputs and compare the canonical state dumps.  -0.0 == +0.0 counts as equal (value comparison), NaNs never do."""
import numpy as np
import boxlcd_amd as B
from boxlcd_amd._lib import Handle
from oracle import pyb2o

BODY_FIELDS = ['cx', 'cy', 'a', 'vx', 'vy', 'w', 'sleepTime', 'awake', 'fat.lo.x', 'fat.lo.y', 'fat.hi.x', 'fat.hi.y']


def make_batch(env_name, n, seed=0, G=None, raster_variant=0):
  env = B.BatchedWorldEnv(env_name, n, G or {}, seed=seed, raster_variant=raster_variant)
  poses, sel = env.sample_initial(n)
  return env, poses, sel


def first_diff(a, b):
  bad = ~(a == b)
  if not bad.any():
    return None
  idx = np.argwhere(bad)[0]
  return tuple(int(i) for i in idx), float(a[tuple(idx)]), float(b[tuple(idx)])


def compare_dumps(gpu, ora_list, tag=''):
  """gpu = (bodies[N,nb,12], joints[N,nj,5], pairs[N,np,18]); ora_list = list of per-env (b, j, p). Returns list of messages."""
  msgs = []
  gb, gj, gp = gpu
  for e, (ob, oj, op) in enumerate(ora_list):
    for name, g, o in (('body', gb[e], ob), ('joint', gj[e], oj), ('pair', gp[e], op)):
      d = first_diff(g, o)
      if d is not None:
        msgs.append(f'{tag} env {e} {name}{d[0]}: gpu={d[1]!r} oracle={d[2]!r}')
  return msgs


def run_substep_parity(env_name, n, steps, seed=0, actions=None, device=0, stop_on_first=True, G=None):
  """Steps GPU and oracle world-step by world-step; returns (n_compared_substeps, messages)."""
  env, poses, sel = make_batch(env_name, n, seed, G)
  desc = env.scene.desc
  h = Handle(desc, n, device)
  h.reset(None, poses, sel)
  oras = [pyb2o.OracleEnv(desc) for _ in range(n)]
  for e, o in enumerate(oras):
    o.reset(poses[e], sel[e])
  assert (h.pair_table() == oras[0].pair_table()).all(), 'pair-slot tables differ'
  if actions is None:
    actions = np.random.RandomState(seed + 1).uniform(-1, 1, (steps, n, desc.n_act)).astype(np.float32)
  msgs, count = [], 0
  msgs += compare_dumps(h.debug_dump(), [o.dump() for o in oras], 'after reset')
  for t in range(steps):
    h.debug_set_motor_speeds(actions[t])
    for o, a in zip(oras, actions[t]):
      o.set_motor_speeds(a)
    for k in range(desc.substeps):
      h.debug_world_step(1)
      for o in oras:
        o.world_step()
      count += 1
      msgs += compare_dumps(h.debug_dump(), [o.dump() for o in oras], f'step {t} sub {k}')
      if msgs and stop_on_first:
        h.close()
        return count, msgs
  # observation + LCD parity at the end
  fs, lcd = h.get_obs(np.float64)
  for e, o in enumerate(oras):
    if not np.allclose(fs[e], o.obs(), rtol=0, atol=1e-6):
      msgs.append(f'obs env {e}: max|d|={np.abs(fs[e] - o.obs()).max()}')
    if not (lcd[e] == o.render()).all():
      msgs.append(f'lcd env {e}: {(lcd[e] != o.render()).sum()} px differ')
  f = h.faults()
  if f.any():
    msgs.append(f'fault flags set: {np.unique(f)}')
  h.close()
  return count, msgs


if __name__ == '__main__':
  import sys
  name = sys.argv[1] if len(sys.argv) > 1 else 'Dropbox'
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
  steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
  cnt, msgs = run_substep_parity(name, n, steps)
  print(f'{name}: compared {cnt} world steps x {n} envs; {len(msgs)} mismatches')
  for m in msgs[:20]:
    print('  ', m)
  sys.exit(1 if msgs else 0)
