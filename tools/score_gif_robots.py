"""Score the robot GIFs with the CPU oracle under each raster variant (seed-7 start and fitted start)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, boxlcd_amd as B
from oracle import pyb2o
z = np.load('tests/golden/gif_lcd_frames.npz')
for name in sys.argv[1:]:
  fitp = f'gpurun_out/fit_robot_{name}.json'
  fit = json.load(open(fitp)) if os.path.exists(fitp) else None
  for variant in (0, 2):
    env = getattr(B.envs, name)(raster_variant=variant); W = env.scene.desc.lcd_w
    gif = np.unpackbits(z[name], axis=-1)[:, :, :W]
    env.seed(7); poses, sel = env._sample_poses(lambda lo, hi: np.array([env.np_random.uniform(lo, hi)]), 1)
    cases = [('seed7', poses[0])] + ([('fit', np.array(fit['poses'], np.float32))] if fit else [])
    for label, P in cases:
      o = pyb2o.OracleEnv(env.scene.desc); o.reset(P, [0] * len(P)); rs = np.random.RandomState(4)
      bad = []
      for t in range(len(gif)):
        o.step(rs.uniform(-1, 1, env.act_size).astype(np.float32)); bad.append(int((o.render() != gif[t]).sum()))
      print(name, 'variant', variant, label, sum(bad), [(i, b) for i, b in enumerate(bad) if b][:25])
