"""Replay of the reference's published recordings from the RECORDER'S OWN INPUTS (no fitted pose anywhere).

Recorder: reference research/scripts/evaluations/demo_imgs.py:58-72 — `env.seed(S)`, `env.reset()`, then per frame
`env.step(np.random.RandomState(A).uniform(-1, 1, act_dim))` and `env.render(mode='human', return_pyglet_view=True)`
(world_env.py:521-535): the GIF's left half is `lcd_render(8W, 8H, 'RGB')`, the right half the LCD x8.  S = 7 and A = 4
in the script; three recordings were made with other seeds, found by matching FRAME 0 ONLY over seeds 0..2999
(Object2-circles: S=1, Object2-cubes: S=6) and over (S, A) in 400 x 40 (LuxoCube: A=3) — the remaining 49..149 frames of
those recordings are then predictions, not fits.  'random' object shapes come from the GLOBAL np.random
(world_env.py:274), which the recorder does not seed: they are read off the recording.

Fixtures: tests/golden/gif_lcd_frames.npz (LCD half) and gif_rgb8_frames.npz (8x RGB half as palette indices), both
decoded from /root/reference/assets/envs/*.gif by tools/gen_gif_fixtures.py.

Two scores per recording: LCD frames (3.2-3.4 px/unit, the oracle's own raster, variant 2 = the recordings' Pillow) and the
8x RGB view (25.6-27.4 px/unit) drawn with the installed Pillow replaying world_env.py:475-511 on the oracle's body
transforms — that score does not depend on our raster restatement (thick shapes only).
"""
import os
import numpy as np
import boxlcd_amd as B

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
# recording -> (env class, shapes read off the recording or None, env seed, action-tape seed)
GIFS = {
    'Dropbox': ('Dropbox', None, 7, 4), 'Bounce': ('Bounce', None, 7, 4), 'Bounce2': ('Bounce2', None, 7, 4),
    'Object2': ('Object2', [1, 0], 7, 4), 'Object2_circles': ('Object2', [0, 0], 1, 4), 'Object2_cubes': ('Object2', [1, 1], 6, 4),
    'Urchin': ('Urchin', None, 7, 4), 'UrchinBall': ('UrchinBall', None, 7, 4), 'UrchinCube': ('UrchinCube', None, 7, 4),
    'Luxo': ('Luxo', None, 7, 4), 'LuxoBall': ('LuxoBall', None, 7, 4), 'LuxoCube': ('LuxoCube', None, 7, 3),
}
ROBOT_COL = ((0.9, 0.4, 0.4), (0.5, 0.3, 0.5))     # reference world_env.py:201
OBJ_COL = ((0.5, 0.4, 0.9), (0.3, 0.3, 0.5))       # reference world_env.py:303
_cache = {}


def fixtures(gif):
  """(rgb uint8 [T, 8H, 8W, 3], lcd uint8 [T, H, W])"""
  if 'lcd' not in _cache:
    _cache['lcd'] = np.load(os.path.join(GOLDEN, 'gif_lcd_frames.npz'))
    _cache['rgb'] = np.load(os.path.join(GOLDEN, 'gif_rgb8_frames.npz'))
  idx = _cache['rgb'][gif]
  rgb = _cache['rgb']['palette'][idx]
  lcd = np.unpackbits(_cache['lcd'][gif], axis=-1)[:, :, :idx.shape[2] // 8]
  return rgb, lcd


def recorder_start(env, seed):
  """env.seed(seed); env.reset() — the reference's own sampling order (boxlcd_amd.world_env._sample_poses)."""
  env.seed(seed)
  poses, sel = env._sample_poses(lambda lo, hi: np.array([env.np_random.uniform(lo, hi)]), 1)
  return poses[0], sel[0]


def pil_rgb(env, o, scale=8):
  """reference world_env.py:460-512 with lcd_mode='RGB', width=8W, height=8H, drawn with PIL on the oracle's transforms."""
  from PIL import Image, ImageDraw
  d = env.scene.desc
  width, height, WIDTH = d.lcd_w * scale, d.lcd_h * scale, float(d.world_w)
  image = Image.new('RGB', (width, height))
  draw = ImageDraw.Draw(image)
  draw.rectangle([0, 0, width, height], fill=(1, 1, 1))
  xf, shapes = o.body_xf()
  for i, spec in enumerate(env.scene.bodies):
    c1, c2 = OBJ_COL if spec.kind == 0 else ROBOT_COL
    color = tuple(int(255.0 * (1 - x)) for x in c1)
    outline = tuple(int(255.0 * (1 - x)) for x in c2)
    kind, val = shapes[i]
    if kind == 'circle':
      pos = xf[i, :2].astype(np.float64)
      rad = np.float64(np.float32(val))
      tl = (pos - rad) / WIDTH * width
      br = (pos + rad) / WIDTH * width
      draw.ellipse(tl.tolist() + br.tolist(), fill=color, outline=outline)
    else:
      pts = val.astype(np.float64) / WIDTH
      pts = tuple(tuple(xy) for xy in (width * pts).tolist())
      draw.polygon(pts, fill=color, outline=outline)
  image = image.transpose(method=Image.FLIP_TOP_BOTTOM)
  return 255 - np.asarray(image)


def replay(gif, oracle, want_rgb=True, frames=None, nudge=None):
  """Returns (bad LCD pixels per frame, bad 8x-RGB pixels per frame) of the oracle replayed from the recorder's inputs."""
  cls, force_sel, seed, aseed = GIFS[gif]
  env = getattr(B.envs, cls)(raster_variant=2)
  rgb, lcd = fixtures(gif)
  P, sel = recorder_start(env, seed)
  if force_sel is not None:
    sel = np.array(force_sel, np.int32)
  o = oracle.OracleEnv(env.scene.desc)
  o.reset(np.asarray(P, np.float32), sel)
  rs = np.random.RandomState(aseed)
  bad_lcd, bad_rgb = [], []
  for t in range(len(lcd) if frames is None else frames):
    a = rs.uniform(-1, 1, env.act_size)          # float64, as the recorder feeds it; the glue is float64 (world_env.py:441)
    if nudge is not None and nudge[0] == t:
      o.nudge(*nudge[1:])
    o.step(a.astype(np.float32))
    bad_lcd.append(int((o.render() != lcd[t]).sum()))
    if want_rgb:
      bad_rgb.append(int((pil_rgb(env, o) != rgb[t]).any(-1).sum()))
  return bad_lcd, bad_rgb


def summary(bad):
  miss = [i for i, b in enumerate(bad) if b]
  return {'exact': sum(b == 0 for b in bad), 'frames': len(bad), 'first_miss': miss[0] if miss else None, 'px': int(sum(bad))}
