"""Barrel writer: random-action rollouts from the batched GPU env into the reference's `*.barrel.npz` format, so
`research/data.py:RolloutDataset` (:123-165) and everything in research/nets train on them unchanged.

Reference writer: research/data.py:36-79 (`fill_barrels`): per barrel 1000 episodes of `ep_len` entries, entry j = the
observation BEFORE action j (entry 0 is the reset observation) and the sampled action j; keys `action` float64
[1000, ep_len, act], `full_state` float32 [1000, ep_len, obs], `proprio` float32, `lcd` bool [1000, ep_len, 16, W];
file name `<timestamp>-<ep_len>.barrel.npz`.  The reference fills a barrel with 1000/num_envs sequential vector-env
episodes; here one barrel is ONE fused device rollout of 1000 environments.
"""
import pathlib
from datetime import datetime
import numpy as np

BARREL_SIZE = int(1e3)


def fill_barrels(env_name, num_barrels, logdir, prefix='train', G=None, seed=0, device=0, barrel_size=BARREL_SIZE, stamp=None):
  """Write `num_barrels` barrels to <logdir>/<prefix>/ and return their paths."""
  from .world_env import BatchedWorldEnv
  venv = BatchedWorldEnv(env_name, barrel_size, G or {}, device=device, seed=seed)
  ep_len = int(venv.G.ep_len)
  h = venv._handle()
  out_dir = pathlib.Path(logdir) / prefix
  out_dir.mkdir(parents=True, exist_ok=True)
  paths = []
  for ti in range(num_barrels):
    obs0 = venv.reset()                                           # entry 0: reset observation
    acts = venv.sample_actions(ep_len)                            # [ep_len, N, act] float32 in (-1, 1); the last one is never applied
    lcd = np.zeros((ep_len - 1, barrel_size, venv.scene.desc.lcd_h, venv.scene.desc.lcd_w), np.uint8)
    fs = np.zeros((ep_len - 1, barrel_size, venv.obs_size), np.float32)
    if ep_len > 1:
      h.rollout(acts[:-1], ep_len - 1, lcd, fs)
    full_state = np.concatenate([obs0['full_state'][None], fs], 0).transpose(1, 0, 2)
    lcd_all = np.concatenate([obs0['lcd'][None], lcd.astype(bool)], 0).transpose(1, 0, 2, 3)
    proprio = full_state[..., venv.pobs_idxs] if venv.pobs_size else np.zeros(full_state.shape[:2] + (1,), np.float32)
    ts = stamp or datetime.now().strftime('%Y%m%dT%H%M%S')
    path = out_dir / f'{ts}-{ti:03d}-{ep_len}.barrel.npz' if num_barrels > 1 else out_dir / f'{ts}-{ep_len}.barrel.npz'
    np.savez_compressed(path, action=acts.transpose(1, 0, 2).astype(np.float64), full_state=np.ascontiguousarray(full_state),
                        proprio=np.ascontiguousarray(proprio.astype(np.float32)), lcd=np.ascontiguousarray(lcd_all))
    paths.append(path)
  venv.close()
  return paths
