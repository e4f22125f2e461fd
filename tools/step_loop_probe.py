"""One env-step per call through step_torch(sync=False) for a few hundred steps: for `rocprofv3 --kernel-trace --stats` (what the
single-step launch of a batch costs on the device, next to the loop's own torch kernels).  usage: python tools/step_loop_probe.py Bounce 100000 300"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import boxlcd_amd as B
name, N, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
env = B.BatchedWorldEnv(name, N, seed=4242)
env.reset_torch()
a = torch.empty((N, env.act_size), dtype=torch.float32, device='cuda')
for sync in (True, False, 'inline'):
  for _ in range(10):
    env.step_torch(a.uniform_(-1, 1), sync=sync)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    env.step_torch(a.uniform_(-1, 1), sync=sync)
  host = time.perf_counter() - t0
  torch.cuda.synchronize()
  sec = time.perf_counter() - t0
  print(f'{name}-{N} sync={sync}: {sec / steps * 1e6:.1f} us per step ({steps * N / sec:.4g} env-steps/s); host loop alone {host / steps * 1e6:.1f} us per step')
