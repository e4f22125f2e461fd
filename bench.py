#!/usr/bin/env python
"""bench.py — env-steps/s of the batched boxLCD hot path (step + obs + LCD render) on N MI355X GPUs of one node.

Workload at N=1 = BASELINE.json configs[1]: envs.Bounce() 16x16, 100 000 parallel envs, 200 env-steps from reset.
One bench "step" = one such rollout (reset -> 200 x [step_kernel + obs/raster kernel], per-step LCD and observation
tensors written to HBM).  Inputs (initial poses, action tape) are resident in HBM before the timed region.
Weak scaling: every rank (one process per GPU) owns `envs` environments; the only collective is an all-gather of the
final frame/observation tensors per rollout (RCCL), as the reference's collectors concatenate per-env results.

Prints ONE JSON line on rank 0 (see the driver contract in the task description): metric/value/unit, roofline block
(HBM, algorithmic bytes per env-step from SURVEY.md §8d x envs per launch / average step_kernel launch time measured with
hipEvents on the handle's stream) and cpu_baseline block (CPU oracle, all host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {'Dropbox': 564, 'Bounce': 564, 'Object2': 924, 'Urchin': 1836, 'LuxoBall': 2236}
HBM_PEAK_GBS = 8000.0                                                                            # MI355X_MICROARCH.md


def cpu_baseline(env_name, T, target_s=12.0):
  """CPU oracle ("port") on all host cores, bounded sample of the same workload (same sampler, same action tape law)."""
  import numpy as np
  import boxlcd_amd as B
  from oracle import pyb2o
  try:
    cores = len(os.sched_getaffinity(0))
  except AttributeError:
    cores = os.cpu_count() or 1
  cores = max(1, min(cores, 16))      # a 1-GPU box's CPU share is 16 cores
  probe = 16 * cores
  env = B.BatchedWorldEnv(env_name, probe, seed=12345)
  poses, sel = env.sample_initial(probe)
  acts = env.sample_actions(T)
  sec, *_ = pyb2o.rollout(env.scene.desc, poses, sel, acts, T, threads=cores, render_every_step=True,
                          want_obs=False, want_lcd=False, want_state=False)
  n = int(max(probe, min(200000, probe * target_s / max(sec, 1e-3))))
  env = B.BatchedWorldEnv(env_name, n, seed=12345)
  poses, sel = env.sample_initial(n)
  acts = env.sample_actions(T)
  sec, *_ = pyb2o.rollout(env.scene.desc, poses, sel, acts, T, threads=cores, render_every_step=True,
                          want_obs=False, want_lcd=False, want_state=False)
  return {'value': n * T / sec, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
          'sample': f'{n} {env_name} envs x {T} env-steps from reset, obs+LCD every step, {cores} threads, {sec:.1f} s'}


def parity_sample(desc, poses, sel, acts, T, gpu_state, gpu_lcd, idx, cores):
  """BASELINE.json's second metric ("LCD frame bit-match %", positions within 1e-4): the oracle as checker on a sample of the
  environments the timed rollout just advanced (same start poses, same action tape)."""
  import numpy as np
  from oracle import pyb2o
  a = None if acts is None else acts[:, idx]
  _, _, olcd, ost = pyb2o.rollout(desc, poses[idx], sel[idx], a, T, threads=cores)
  return {'envs_compared': int(len(idx)), 'env_steps': int(T),
          'lcd_frame_bit_match_pct': 100.0 * float((gpu_lcd[idx] == olcd).reshape(len(idx), -1).all(1).mean()),
          'max_abs_pose_diff': float(np.abs(gpu_state[idx][:, :, :3] - ost[:, :, :3]).max())}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3, help='timed rollouts (each = envs x rollout_len env-steps)')
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--env', default='Bounce')
  ap.add_argument('--envs', type=int, default=100000, help='environments per GPU')
  ap.add_argument('--rollout-len', type=int, default=200)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  args = ap.parse_args()

  import numpy as np
  import torch
  import boxlcd_amd as B
  from boxlcd_amd import dist as bdist
  from boxlcd_amd._lib import Handle

  # BENCH_BACKEND=gloo BENCH_DEVICE=0: rehearse the N>1 path on a box with fewer GPUs than ranks (never used by the driver)
  rank, world, local = bdist.init_from_env(os.environ.get('BENCH_BACKEND'))
  if 'BENCH_DEVICE' in os.environ:
    local = int(os.environ['BENCH_DEVICE'])
  if world != args.gpus:
    if rank == 0:
      print(f'warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE', file=sys.stderr)
  if not torch.cuda.is_available():
    raise RuntimeError('bench.py needs a GPU: boxlcd_amd has no CPU path')
  torch.cuda.set_device(local)
  dev = torch.device('cuda', local)
  N, T = args.envs, args.rollout_len

  venv = B.BatchedWorldEnv(args.env, N, seed=1000 + rank)
  d = venv.scene.desc
  h = Handle(d, N, local)
  poses_np, sel_np = venv.sample_initial(N)
  poses = torch.as_tensor(poses_np).to(dev)
  sel = torch.as_tensor(sel_np).to(dev)
  acts = torch.as_tensor(venv.sample_actions(T)).to(dev)                   # [T, N, act] resident in HBM
  lcd = torch.empty((T, N, d.lcd_h, d.lcd_w), dtype=torch.uint8, device=dev)
  obs = torch.empty((T, N, d.n_obs), dtype=torch.float32, device=dev)
  torch.cuda.synchronize()

  kernel_ms, launches = [], []

  def one_rollout():
    h.reset(None, poses, sel)
    h.rollout(acts, T, lcd, obs)
    ms, n = h.last_kernel_ms()
    kernel_ms.append(ms)
    launches.append(n)
    if world > 1:                                                            # concatenate the rollout's final tensors
      bdist.all_gather_shards(lcd[-1], N * world)
      bdist.all_gather_shards(obs[-1], N * world)

  for _ in range(args.warmup):
    one_rollout()
  kernel_ms.clear(); launches.clear()
  bdist.barrier(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    one_rollout()
  torch.cuda.synchronize(); bdist.barrier()
  dt = bdist.max_over_ranks(time.perf_counter() - t0)

  faults = int((h.faults() != 0).sum())
  awake_frac = float(h.get_poses()[:, :, 3].mean())
  if rank == 0:
    total_env_steps = float(args.steps) * T * N * world
    avg_launch_s = (sum(kernel_ms) / max(sum(launches), 1)) / 1e3
    # one launch advances N envs by (T / launches-per-rollout) env-steps (fused rollout chunks)
    steps_per_launch = T * args.steps / max(sum(launches), 1)
    bytes_per_launch = ALG_BYTES.get(args.env, 0) * N * steps_per_launch
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    traffic = None   # HBM bytes per step_kernel launch from the committed rocprofv3 PMC passes of this same command
    try:
      prof = json.load(open(os.path.join(ROOT, 'profiles', 'r01_final4_bounce100k_pmc.json')))
      if args.env == 'Bounce' and N == 100000 and T == 200:
        k = [x for x in prof if 'step_kernel' in x][0]
        traffic = (prof[k]['FETCH_SIZE']['mean'] + prof[k]['WRITE_SIZE']['mean']) * 1024.0
    except Exception:
      traffic = None
    out = {
        'metric': 'env_steps_per_sec', 'value': total_env_steps / dt, 'unit': 'env-steps/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'envs.{args.env}() {d.lcd_h}x{d.lcd_w}, {N} envs/GPU, {T} env-steps per rollout from reset, '
                               'U(-1,1) actions, obs + LCD rendered every step', 'envs_per_gpu': N, 'rollout_len': T,
                   'parallelism': f'env-sharded x{world}', 'raster_variant': 'legacy', 'faulted_envs': faults,
                   'awake_fraction_at_end': awake_frac},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': 'profiles/r01_final4_bounce100k_pmc.json (FETCH_SIZE + WRITE_SIZE, KB -> B, per launch)', 'kernel': 'step_kernel',
                     'avg_launch_ms': avg_launch_s * 1e3, 'env_steps_per_env_per_launch': steps_per_launch, 'alg_bytes_per_env_step': ALG_BYTES.get(args.env, 0),
                     'note': 'path is VALU/latency-bound (SURVEY.md §8d): HBM fraction is reported as required, not the limiter'},
    }
    if not args.no_cpu_baseline and world == 1:
      out['cpu_baseline'] = cpu_baseline(args.env, T)
      idx = np.random.RandomState(0).choice(N, min(N, 1024), replace=False)
      out['parity'] = parity_sample(d, poses_np, sel_np, acts.cpu().numpy(), T, h.debug_dump()[0], lcd[-1].cpu().numpy(), idx,
                                    out['cpu_baseline']['cores'])
    print(json.dumps(out))
  h.close()
  if world > 1:
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
