"""Robot GIF pins: the reference's demo recorder (research/scripts/evaluations/demo_imgs.py:60-72) seeds the env with 7 and
draws actions from np.random.RandomState(4).uniform(-1, 1, act_dim).  With that action tape the oracle reproduces
assets/envs/{Urchin,Luxo,...}.gif almost exactly from the seed-7 start; this tool scans the few free start parameters
(root x, root angle; object x/y/angle) around the seed-7 sample for an exact match.  CPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import boxlcd_amd as B
from oracle import pyb2o

Z = np.load('tests/golden/gif_lcd_frames.npz')

def seed7_sample(env, fs_delta=None):
  env.seed(7)
  return env._sample_poses(lambda lo, hi: np.array([env.np_random.uniform(lo, hi)]), 1)

def score(env, gif, poses, sel, T=None):
  o = pyb2o.OracleEnv(env.scene.desc)
  o.reset(poses, sel)
  rs = np.random.RandomState(4)
  bad = []
  for t in range(T or len(gif)):
    o.step(rs.uniform(-1, 1, env.act_size).astype(np.float32))
    bad.append(int((o.render() != gif[t]).sum()))
  return bad

if __name__ == '__main__':
  for name in sys.argv[1:]:
    env = getattr(B.envs, name)()
    gif = np.unpackbits(Z[name], axis=-1)[:, :, :env.scene.desc.lcd_w]
    poses, sel = seed7_sample(env)
    bad = score(env, gif, poses[0], sel[0])
    print(name, 'seed-7 sample:', sum(bad), bad)
