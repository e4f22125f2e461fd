#!/usr/bin/env python
"""bench.py — env-steps/s of the batched boxLCD hot path (step + obs + LCD render) on N MI355X GPUs of one node.

Workload at N=1 = BASELINE.json configs[1]: envs.Bounce() 16x16, 100 000 parallel envs, 200-env-step rollouts from reset.
One bench "step" = `--rollouts-per-step` (default 20) such rollouts = 4 000 env-steps of every environment (reset -> 200 x
[world steps + obs + LCD raster], per-step LCD and observation tensors written to HBM), so that the driver's `--steps 20`
times ~5 s of GPU work.  Inputs (initial poses, action tape) are resident in HBM before the timed region.

Weak scaling (`--gpus N`, one process per GPU, launched by torchrun): every rank owns `--envs` environments and steps them
with no data-path collective - with EXACTLY the N=1 stepping path (one `blcd_rollout` per rollout, cohorts and all); the
finished rollout's FULL result tensors `lcd[T, N, 16, W]` and `full_state[T, N, obs]` are then gathered over RCCL on a side
stream in 20-step chunks while the next rollout steps into the other output buffer.  `--gather all`
(default; north_star's all-gather) or `--gather consumer` (only rank 0 receives).  The N>1 line also carries
`stepping_only`: the same rollouts without any transfer, so that a scaling curve separates kernel scaling from the gather.

Prints ONE JSON line on rank 0: metric/value/unit, `roofline` (HBM; algorithmic bytes per env-step from SURVEY.md §8d x envs
x env-steps per launch / mean step_kernel launch time from hipEvents on the handle's stream), `cpu_baseline` (the CPU oracle
on the host cores over a bounded sample), `parity` (BASELINE's second metric on a sample of the timed batch) and `configs`
(the other single-GPU BASELINE workloads incl. north_star's Dropbox-100k, each with its own roofline block).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {'Dropbox': 564, 'Bounce': 564, 'Object2': 924, 'Urchin': 1836, 'LuxoBall': 2236,
             'Crab': 2 * (17 * 32 + 16 * 16 + 68 * 28) + 4 * 12 + 4 * 68 + 32 * 64}             # same rule (2 x state + 4 act + 4 obs + lcd bytes): 17 bodies, 16 joints, 68 wall pair slots, 32x64 LCD = 7776 B
HBM_PEAK_GBS = 8000.0                                                                            # MI355X_MICROARCH.md
RASTER_NAMES = {0: 'pillow-9.0.x (inferred, no fixture)', 1: 'pillow-12.2 (goldens)', 2: "recordings' Pillow (default; pinned by the reference's GIF frames)"}
# committed PMC summaries (tools/pmc_summary.py) per workload: HBM traffic and VALU figures are PROFILE-DERIVED (measured with
# rocprofv3 --pmc on the same command on an earlier box, not in this run) and labelled so in the JSON line
PMC_PROFILES = {('Bounce', 100000): 'profiles/r04_bounce100k_pmc.json', ('Dropbox', 100000): 'profiles/r04_dropbox100k_pmc.json',
                ('Urchin', 50000): 'profiles/r04_urchin50k_pmc.json', ('LuxoBall', 50000): 'profiles/r04_luxoball50k_pmc.json',
                ('Object2', 200000): 'profiles/r04_object2_200k_pmc.json', ('Crab', 20000): 'profiles/r04_crab20k_pmc.json'}


FETCH_FACTOR = 2.0   # MI355X_MICROARCH.md "HBM": gfx950 FETCH_SIZE tallies 128-B requests at 64 B - exactly half the bytes of wide coalesced reads


def pmc_blocks(env_name, n_envs, dispatches_per_chunk=1, steps_per_launch=None):
  """(traffic bytes per launch or None, traffic detail dict, valu block or None) from the committed PMC summary of this workload.
  The summary must carry the revision stamp of the device code this process runs (tools/csrc_rev.py) - a kernel change that
  was not re-profiled yields None, not stale counters - and, where launches of several lengths were profiled together (jointed
  classes: 50-step launches in the first rollout, 200-step launches after it), the group that matches `steps_per_launch`."""
  from tools.csrc_rev import csrc_rev
  for table in (PMC_PROFILES,):
    path = table.get((env_name, n_envs))
    if not path or not os.path.exists(os.path.join(ROOT, path)):
      continue
    try:
      prof = json.load(open(os.path.join(ROOT, path)))
      meta = prof.get('_meta', {})
      if meta.get('csrc_rev') != csrc_rev():
        return None, {'source': path, 'dropped': f"profile taken on device code {meta.get('csrc_rev')}, this run is {csrc_rev()}: re-profile (tools/profile_all.sh)"}, None
      k = [x for x in prof if 'step_kernel' in x][0]
      c = prof[k]
      groups = c.get('by_env_steps')
      basis = 'all step_kernel dispatches of the profile'
      if groups and steps_per_launch is not None:
        g = str(int(round(steps_per_launch)))
        if g not in groups:
          return None, {'source': path, 'dropped': f'no profiled launches of {g} env-steps (profile has {sorted(groups)})'}, None
        c, basis = groups[g], f'the profile\'s {g}-env-step launches only'
      traffic = detail = None
      if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        f, w = c['FETCH_SIZE']['mean'] * 1024.0 * dispatches_per_chunk, c['WRITE_SIZE']['mean'] * 1024.0 * dispatches_per_chunk
        traffic = FETCH_FACTOR * f + w
        detail = {'source': path, 'basis': basis, 'dispatches_per_chunk': dispatches_per_chunk, 'fetch_size_raw': f, 'write_size': w,
                  'traffic_uncorrected': f + w,
                  'note': 'rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE (separate passes; KB -> B) per step_kernel dispatch; profile-derived, not measured in '
                          'this run.  `traffic` = 2 x FETCH_SIZE + WRITE_SIZE: gfx950 tallies a 128-B read request at 64 B (MI355X_MICROARCH.md); '
                          'calibrated on this path\'s own patterns (profiles/r04_fetch_calib.txt: FETCH_SIZE = 0.5000 x true bytes for 4 B/lane '
                          'field-major state loads and for 16 B/lane loads alike, WRITE_SIZE = 1.0000 x true bytes)'}
      valu = None
      if 'SQ_WAVE_CYCLES' in c and 'SQ_ACTIVE_INST_VALU' in c:
        w = c['SQ_WAVE_CYCLES']['mean']
        valu = {'valu_busy_frac_of_wave_cycles': c['SQ_ACTIVE_INST_VALU']['mean'] / w,
                'waiting_frac_of_wave_cycles': c['SQ_WAIT_ANY']['mean'] / w if 'SQ_WAIT_ANY' in c else None,
                'lanes_per_valu_inst': (c['SQ_THREAD_CYCLES_VALU']['mean'] / c['SQ_ACTIVE_INST_VALU']['mean']) if 'SQ_THREAD_CYCLES_VALU' in c else None,
                'waves_per_dispatch': c['SQ_WAVES']['mean'] if 'SQ_WAVES' in c else None,
                'lanes_note': 'active lanes per VALU instruction INCLUDING shadow lanes: lanes beyond a narrow wave\'s width run as copies of its last environment (no stores) so that every wave keeps the coalesced frame path; an under-subscribed batch of w environments per wave carries w / 64 of this figure',
                'source': f'{path} ({basis}; rocprofv3 --pmc SQ_* pass of the same command; profile-derived)'}
      return traffic, detail, valu
    except Exception as ex:
      return None, {'source': path, 'dropped': f'unreadable summary: {ex}'}, None
  return None, None, None


def cpu_baseline(env_name, T, target_s=12.0):
  """CPU oracle ("port") on the host cores, bounded sample of the same workload (same sampler, same action tape law)."""
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


class Workload:
  """One env class x batch size on one GPU: resident inputs, output tensors, and the rollout loop."""

  def __init__(self, env_name, N, T, local, dev, seed, chunk=20, env_id_base=0):
    import torch
    import boxlcd_amd as B
    from boxlcd_amd._lib import Handle
    self.name, self.N, self.T, self.chunk = env_name, N, T, chunk
    self.venv = B.BatchedWorldEnv(env_name, N, seed=seed, env_id_base=env_id_base)
    self.d = d = self.venv.scene.desc
    self.h = Handle(d, N, local)
    self.poses_np, self.sel_np = self.venv.sample_initial(N)
    self.poses = torch.as_tensor(self.poses_np).to(dev)
    self.sel = torch.as_tensor(self.sel_np).to(dev)
    self.acts = torch.as_tensor(self.venv.sample_actions(T)).to(dev)                 # [T, N, act] resident in HBM
    self.lcd = torch.empty((T, N, d.lcd_h, d.lcd_w), dtype=torch.uint8, device=dev)
    self.obs = torch.empty((T, N, d.n_obs), dtype=torch.float32, device=dev)
    self.kernel_ms, self.launches = 0.0, 0
    self.stream = torch.cuda.ExternalStream(self.h.stream(), device=dev)
    self.gatherer, self.gather_done, self.bufs, self.k = None, {}, [(self.lcd, self.obs)], 0
    self.faulted = False

  def enable_gather(self, world, mode, rank):
    """Second output buffer (a rollout steps into one while the previous one's tensors travel) + the chunk gatherer."""
    import torch
    from boxlcd_amd import dist as bdist
    c = min(self.chunk, self.T)
    self.bufs.append((torch.empty_like(self.lcd), torch.empty_like(self.obs)))
    self.gatherer = bdist.ChunkGatherer(world, [self.lcd[:c], self.obs[:c]], binary=[True, False], mode=mode, rank=rank)   # mode-1 LCD frames are 0/1 bytes

  def rollout(self, gather=True):
    from boxlcd_amd._lib import EnvFaultError
    h = self.h
    lcd, obs = self.bufs[self.k % len(self.bufs)]
    self.k += 1
    self.lcd, self.obs = lcd, obs
    h.reset(None, self.poses, self.sel)
    if self.gatherer is not None and id(lcd) in self.gather_done:   # the gather that last read this buffer must be through
      self.stream.wait_event(self.gather_done[id(lcd)])
    try:
      h.rollout(self.acts, self.T, lcd, obs)          # the SAME call at every N: one fused rollout, cohorts and all
    except EnvFaultError:
      self.faulted = True                              # the rollout completed; the flags are reported as `faulted_envs`
    ms, n = h.last_kernel_ms()
    self.kernel_ms += ms
    self.launches += n
    if self.gatherer is not None and gather:
      for t0 in range(0, self.T - self.chunk + 1, self.chunk):
        t1 = t0 + self.chunk
        self.gather_done[id(lcd)] = self.gatherer.gather([lcd[t0:t1], obs[t0:t1]], producer=self.stream)

  def roofline(self, rollouts):
    avg_launch_s = (self.kernel_ms / max(self.launches, 1)) / 1e3
    steps_per_launch = self.T * rollouts / max(self.launches, 1)     # one launch advances N envs by this many env-steps
    bpe = ALG_BYTES.get(self.name, 0)
    achieved = bpe * self.N * steps_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    cohorts = self.d.n_joints == 0 and self.N > 65536 and os.environ.get('BLCD_COHORTS', '2') != '1'
    traffic, tsrc, valu = pmc_blocks(self.name, self.N, 2 if cohorts and os.environ.get('BLCD_COHORTS', '2') == '2' else 1, steps_per_launch)
    if any(k.startswith('BLCD_') and k != 'BLCD_LAUNCH_LOG' for k in os.environ) or self.T != 200:
      traffic = valu = None                            # the committed profiles describe the default knobs only
      tsrc = {'dropped': 'BLCD_* overrides or a non-default rollout length: the committed profiles do not describe this run'}
    return {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': traffic, 'traffic_detail': tsrc,
            'traffic_over_algorithmic': (traffic / (bpe * self.N * steps_per_launch)) if traffic and bpe else None, 'valu': valu, 'kernel': 'step_kernel', 'avg_launch_ms': avg_launch_s * 1e3,
            'env_steps_per_env_per_launch': steps_per_launch, 'alg_bytes_per_env_step': bpe,
            'note': 'path is VALU/latency-bound (SURVEY.md §8d): HBM fraction is reported as required, not the limiter'
                    + ('; this batch is stepped as two cohorts on two streams: a "launch" here is one chunk of the WHOLE batch (two '
                       'concurrent step_kernel launches), avg_launch_ms = the overlapped sequence (re-bin kernels included) / chunks'
                       if cohorts else '')}

  def close(self):
    self.h.close()


def step_loop(env_name, N, local, steps, warmup=5):
  """The call shape of the reference's policy-in-the-loop consumers (research/rl/ppo.py:127-133 `o, r, d, _ = env.step(a)`,
  rl/sac.py:200-214): ONE env-step per call through BatchedWorldEnv.step_torch - device-resident actions in, observation dict
  (full_state, proprio, lcd) of CUDA tensors out, nothing crosses PCIe.  The 'policy' is a fresh U(-1,1) action tensor per step
  (torch.rand on the device, inside the timed loop, as a policy's forward pass would be)."""
  import torch
  import boxlcd_amd as B
  env = B.BatchedWorldEnv(env_name, N, device=local, seed=4242)
  env.reset_torch()
  dev = torch.device('cuda', local)
  a = torch.empty((N, env.act_size), dtype=torch.float32, device=dev)
  for _ in range(warmup):
    env.step_torch(a.uniform_(-1, 1))
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    obs, rew, done, _ = env.step_torch(a.uniform_(-1, 1))
  torch.cuda.synchronize()
  sec = time.perf_counter() - t0
  # the same loop without any host synchronisation (blcd_step_obs_async): the step is ordered between the torch work around it on the device
  for _ in range(warmup):
    env.step_torch(a.uniform_(-1, 1), sync=False)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    obs, rew, done, _ = env.step_torch(a.uniform_(-1, 1), sync=False)
  torch.cuda.synchronize()
  sec_async = time.perf_counter() - t0
  # ... and with the step queued on torch's own stream (blcd_set_async_stream): no hand-off between streams either
  for _ in range(warmup):
    env.step_torch(a.uniform_(-1, 1), sync='inline')
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    obs, rew, done, _ = env.step_torch(a.uniform_(-1, 1), sync='inline')
  torch.cuda.synchronize()
  sec_inline = time.perf_counter() - t0
  faults = int((env.faults() != 0).sum())
  env.close()
  return {'value': steps * N / sec, 'unit': 'env-steps/s', 'env_steps_per_call': 1, 'calls': steps, 'ms_per_call': sec / steps * 1e3,
          'async': {'value': steps * N / sec_async, 'ms_per_call': sec_async / steps * 1e3, 'faults': faults,
                    'note': 'step_torch(sync=False) = blcd_step_obs_async: no host synchronisation, stream-ordered against torch in both directions'},
          'inline': {'value': steps * N / sec_inline, 'ms_per_call': sec_inline / steps * 1e3,
                     'note': "step_torch(sync='inline'): the asynchronous step queued ON torch's current stream (blcd_set_async_stream)"},
          'note': 'BatchedWorldEnv.step_torch = blcd_step_obs: one call, one launch (step_kernel writes the observation row and the frame), one stream '
                  'synchronisation, device tensors in and out (ordered behind torch\'s stream); round 3 issued blcd_step + blcd_get_obs (two launches, four synchronisations)'}


def wire_model(world, N, T, d, mode, step_s_per_rollout, weak=True):
  """Bytes a rollout puts on xGMI and what that predicts, so that the first real SCALE run can be checked against a number:
  LCD frames travel at 1 bit per pixel, observations as float32.  xGMI is point-to-point, 7 links x ~153 GB/s per GPU
  (MI355X_MICROARCH.md): a DIRECT all-gather uses all 7 links at once (each peer's shard arrives over its own link); a single
  RING all-gather moves (world-1) shards over one link.  Gather overlaps the next rollout's stepping, so the predicted
  weak-scaling efficiency is stepping time / max(stepping time, gather time)."""
  LINK = 153e9
  per_rank = N * T * (d.lcd_h * d.lcd_w // 8 + 4 * d.n_obs)
  recv = (world - 1) * per_rank if mode == 'all' else 0
  t_direct = per_rank / LINK
  t_ring = (world - 1) * per_rank / LINK
  eff = lambda tg: step_s_per_rollout / max(step_s_per_rollout, tg) if step_s_per_rollout > 0 else None
  out = {'bytes_sent_per_rank_per_rollout': per_rank, 'bytes_received_per_rank_per_rollout': recv if mode == 'all' else '(world-1) x per_rank on rank 0 only',
         'link_GBps': LINK / 1e9, 'gather_s_direct_all_links': t_direct, 'gather_s_single_ring': t_ring, 'stepping_s_per_rollout_measured': step_s_per_rollout,
         'predicted_efficiency_direct': eff(t_direct), 'predicted_efficiency_single_ring': eff(t_ring)}
  if weak:   # weak scaling keeps the shard size: the same shard's gather over 7 peers against the same stepping time
    out['predicted_at_8_ranks'] = {'gather_s_direct': per_rank / LINK, 'gather_s_single_ring': 7 * per_rank / LINK,
                                   'efficiency_direct': eff(per_rank / LINK), 'efficiency_single_ring': eff(7 * per_rank / LINK)}
  return out


def time_rollouts(w, rollouts, warmup, bdist, torch, gather=True):
  for _ in range(warmup):
    w.rollout(gather)
  if w.gatherer is not None:
    w.gatherer.finish()
  w.kernel_ms, w.launches = 0.0, 0
  bdist.barrier(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(rollouts):
    w.rollout(gather)
  if w.gatherer is not None:
    w.gatherer.finish()
  torch.cuda.synchronize(); bdist.barrier()
  return bdist.max_over_ranks(time.perf_counter() - t0)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20, help='timed bench steps (each = --rollouts-per-step rollouts)')
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--rollouts-per-step', type=int, default=20, help='rollouts of --rollout-len env-steps per bench step')
  ap.add_argument('--env', default='Bounce')
  ap.add_argument('--envs', type=int, default=100000, help='environments per GPU')
  ap.add_argument('--rollout-len', type=int, default=200)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-configs', action='store_true', help='skip the extra single-GPU workloads of the `configs` block')
  ap.add_argument('--gather', choices=['all', 'consumer'], default='all', help='N>1: all-gather to every rank, or gather to rank 0 only')
  args = ap.parse_args()

  world_env = int(os.environ.get('WORLD_SIZE', '1'))
  if args.gpus > 1 and world_env == 1 and 'RANK' not in os.environ:
    # not under torchrun: start one process per GPU BEFORE anything touches the GPU, and exit with the launcher's code
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', os.environ.get('MASTER_PORT', '29533'), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))

  # stdout carries exactly ONE line, the JSON: libraries that chat on fd 1 (RCCL's start-up banner, gloo's "[Gloo] Rank ..."
  # lines) are sent to stderr for the whole run, and rank 0 writes the result to the saved descriptor at the end
  sys.stdout.flush()
  json_fd = os.dup(1)
  os.dup2(2, 1)

  import numpy as np
  import torch
  from boxlcd_amd import dist as bdist

  # BENCH_BACKEND=gloo BENCH_DEVICE=0: rehearse the N>1 path on a box with fewer GPUs than ranks (never used by the driver)
  rank, world, local = bdist.init_from_env(os.environ.get('BENCH_BACKEND'))
  if 'BENCH_DEVICE' in os.environ:
    local = int(os.environ['BENCH_DEVICE'])
  if world != args.gpus:
    raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
  if not torch.cuda.is_available():
    raise RuntimeError('bench.py needs a GPU: boxlcd_amd has no CPU path')
  torch.cuda.set_device(local)
  dev = torch.device('cuda', local)
  N, T = args.envs, args.rollout_len
  overrides = {k: v for k, v in os.environ.items() if k.startswith('BLCD_') or k.startswith('BOXLCD_')}

  # every rank: the SAME seed; the reset sampler keys its counter with the global env id (BatchedWorldEnv env_id_base), the
  # mirror used here samples environments rank*N .. rank*N+N-1 of one world*N batch
  w = Workload(args.env, N, T, local, dev, seed=1000, env_id_base=rank * N)
  d = w.d
  if world > 1:
    w.enable_gather(world, args.gather, rank)
  torch.cuda.synchronize()
  rollouts = args.steps * args.rollouts_per_step
  dt = time_rollouts(w, rollouts, args.warmup * args.rollouts_per_step, bdist, torch)
  stepping_only = None
  if world > 1:   # the same rollouts with no transfer at all (outside the timed region above): kernel scaling by itself
    r2 = max(1, min(rollouts, 2 * args.rollouts_per_step))
    stepping_only = {'value': float(r2) * T * N * world / time_rollouts(w, r2, 0, bdist, torch, gather=False), 'unit': 'env-steps/s',
                     'rollouts': r2, 'note': 'identical stepping path, gather skipped; max over ranks, barrier + synchronize on both sides'}

  faults = int((w.h.faults() != 0).sum())
  awake_frac = float(w.h.get_poses()[:, :, 3].mean())
  if rank == 0:
    total_env_steps = float(rollouts) * T * N * world
    roof = w.roofline(rollouts)
    out = {
        'metric': 'env_steps_per_sec', 'value': total_env_steps / dt, 'unit': 'env-steps/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'envs.{args.env}() {d.lcd_h}x{d.lcd_w}, {N} envs/GPU, one step = {args.rollouts_per_step} rollouts of {T} env-steps '
                               'from reset, U(-1,1) actions, obs + LCD rendered every env-step', 'envs_per_gpu': N, 'rollout_len': T,
                   'rollouts_per_step': args.rollouts_per_step, 'parallelism': f'env-sharded x{world}',
                   'collective': None if world == 1 else ('all-gather' if args.gather == 'all' else 'gather to rank 0') + f' of every chunk\'s lcd[{w.chunk},{N},{d.lcd_h},{d.lcd_w}] u8 (1 bit per pixel on the wire, delivered as u8) + full_state[{w.chunk},{N},{d.n_obs}] f32 on a side stream, overlapped with the next rollout',
                   'raster_variant': RASTER_NAMES.get(w.venv.raster_variant, str(w.venv.raster_variant)), 'overrides': overrides,
                   'faulted_envs': faults, 'awake_fraction_at_end': awake_frac},
        'roofline': roof,
    }
    if stepping_only is not None:
      out['stepping_only'] = stepping_only
      out['wire'] = wire_model(world, N, T, d, args.gather, T * N * world / stepping_only['value'])
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(args.env, T)
      idx = np.random.RandomState(0).choice(N, min(N, 1024), replace=False)
      out['parity'] = parity_sample(d, w.poses_np, w.sel_np, w.acts.cpu().numpy(), T, w.h.debug_dump()[0], w.lcd[-1].cpu().numpy(), idx,
                                    out['cpu_baseline']['cores'])
  w.close()
  del w
  torch.cuda.empty_cache()
  if world > 1 and not args.no_configs and args.env == 'Bounce':
    # BASELINE configs[3] and [4] AS STATED: the 50 000-env LuxoBall batch and the 200 000-env Object2 batch cut into `world`
    # contiguous shards (strong scaling of a fixed batch), each rank stepping its shard and the finished rollout gathered over
    # RCCL; every rank takes part (collectives), rank 0 reports.  `value` = whole-batch env-steps/s with the gather, max over ranks.
    cfgs = {}
    for name, n_total, rolls in (('LuxoBall', 50000, 2), ('Object2', 200000, 3)):
      n_r = n_total // world                       # equal shards (all_gather_into_tensor); the remainder of an uneven cut is dropped and reported
      try:                                         # an error that every rank hits alike (allocation, a Python slip) must not cost the main line
        ww = Workload(name, n_r, T, local, dev, seed=1000, env_id_base=rank * n_r)
        ww.enable_gather(world, args.gather, rank)
        sec = time_rollouts(ww, rolls, 1, bdist, torch)
        sec2 = time_rollouts(ww, rolls, 0, bdist, torch, gather=False)
        if rank == 0:
          cfgs[f'{name}-{n_total}-sharded-x{world}'] = {
              'value': rolls * T * n_r * world / sec, 'unit': 'env-steps/s', 'scaling': 'strong', 'envs_total': n_r * world, 'envs_per_gpu': n_r, 'rollouts': rolls,
              'seconds': sec, 'gather': args.gather, 'stepping_only': {'value': rolls * T * n_r * world / sec2, 'unit': 'env-steps/s', 'seconds': sec2},
              'wire': wire_model(world, n_r, T, ww.d, args.gather, sec2 / rolls, weak=False), 'faulted_envs': int((ww.h.faults() != 0).sum())}
        ww.close()
        del ww
      except Exception as ex:
        cfgs[f'{name}-{n_total}-sharded-x{world}'] = {'error': repr(ex)}
      torch.cuda.empty_cache()
    if rank == 0:
      out['configs'] = cfgs
  if rank == 0 and world == 1 and not args.no_configs and args.env == 'Bounce':
    # the other BASELINE workloads, driver-run on this one GPU: north_star's Dropbox-100k target, configs[2] Urchin-50k, and the
    # whole batches of configs[3] LuxoBall-50k and configs[4] Object2-200k (they fit one MI355X)
    cfgs = {}
    for name, n_envs, rolls in (('Dropbox', 100000, 20), ('Urchin', 50000, 3), ('LuxoBall', 50000, 2), ('Object2', 200000, 2), ('Crab', 20000, 1)):
      try:
        ww = Workload(name, n_envs, T, local, dev, seed=1000)
        sec = time_rollouts(ww, rolls, 0 if name == 'Crab' else 1, bdist, torch)     # the largest class: one cold rollout (seconds each)
        cfgs[f'{name}-{n_envs}'] = {'value': rolls * T * n_envs / sec, 'unit': 'env-steps/s', 'rollouts': rolls, 'seconds': sec,
                                    'faulted_envs': int((ww.h.faults() != 0).sum()), 'roofline': ww.roofline(rolls)}
        ww.close()
        del ww
      except Exception as ex:      # an extra workload must not cost the main line
        cfgs[f'{name}-{n_envs}'] = {'error': repr(ex)}
      torch.cuda.empty_cache()
    out['configs'] = cfgs
    # (last: its 'inline' leg queues step kernels on torch's stream, i.e. on another hardware queue)
    # policy-in-the-loop call shape (one env-step per call, device tensors): what PPO / SAC consumers see
    try:
      out['step_loop'] = {'Bounce-100000': step_loop('Bounce', 100000, local, 200), 'Urchin-50000': step_loop('Urchin', 50000, local, 40)}
    except Exception as ex:
      out['step_loop'] = {'error': repr(ex)}
  if rank == 0:
    os.write(json_fd, (json.dumps(out) + '\n').encode())
  if world > 1:
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
