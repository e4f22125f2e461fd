"""Fused-rollout throughput under different environment-level scheduling policies (passes per chunk x lanes threshold).
usage: python tools/yield_bench.py Dropbox 100000 [T]"""
import os, sys, time, subprocess
if len(sys.argv) > 1 and sys.argv[1] == '--one':
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import numpy as np, torch
  import boxlcd_amd as B
  from boxlcd_amd._lib import Handle
  name, N, T = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
  venv = B.BatchedWorldEnv(name, N, seed=1000)
  d = venv.scene.desc
  h = Handle(d, N, 0)
  poses, sel = venv.sample_initial(N)
  dev = torch.device('cuda', 0)
  poses_t, sel_t = torch.as_tensor(poses).to(dev), torch.as_tensor(sel).to(dev)
  acts = torch.as_tensor(venv.sample_actions(T)).to(dev)
  lcd = torch.empty((T, N, d.lcd_h, d.lcd_w), dtype=torch.uint8, device=dev)
  obs = torch.empty((T, N, d.n_obs), dtype=torch.float32, device=dev)
  def roll():
    h.reset(None, poses_t, sel_t); h.rollout(acts, T, lcd, obs)
  roll(); roll(); torch.cuda.synchronize()
  R = 3
  t0 = time.perf_counter()
  for _ in range(R): roll()
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  print(f'{name} N={N} passes={os.environ.get("BLCD_YIELD_PASSES","def")} lanes={os.environ.get("BLCD_YIELD_LANES","def")} chunk={os.environ.get("BLCD_CHUNK","def")}: {R*T*N/dt:.4g} env-steps/s', flush=True)
else:
  name, N = sys.argv[1], sys.argv[2]
  T = sys.argv[3] if len(sys.argv) > 3 else '200'
  grid = [('1', None, None)] + [('2', l, c) for c in ('1', '2', '3', '5', '10', '20') for l in ('32',)] + [('2', '16', '3'), ('2', '64', '3'), ('2', '48', '5'), ('3', '32', '5')]
  for passes, lanes, chunk in grid:
    env = dict(os.environ)
    if passes: env['BLCD_YIELD_PASSES'] = passes
    if lanes: env['BLCD_YIELD_LANES'] = lanes
    if chunk: env['BLCD_CHUNK'] = chunk
    rc = subprocess.call([sys.executable, os.path.abspath(__file__), '--one', name, N, T], env=env)
    if rc: sys.exit(rc)
