"""Pillow goldens for `lcd_render(width, height, lcd_mode)` at arbitrary sizes and in RGB mode (tests/golden/pillow_rgb.npz).

For seeded random poses of several env classes this replays the reference's PIL call sequence (boxLCD/world_env.py:475-511:
Image.new / draw.rectangle(bg) / draw.ellipse(fill, outline) / draw.polygon(fill, outline) / FLIP_TOP_BOTTOM / 255 - img) with the
installed Pillow on the body transforms the oracle reports (`trans * v` in Box2D's vertex order - the outline's Bresenham
lines depend on edge direction).  Sizes: the 8x human view, the native LCD size in RGB, an odd non-proportional size, and a
larger mode-'1' canvas.  Run in the authoring container:  python tools/gen_pillow_rgb_goldens.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from PIL import Image, ImageDraw, __version__ as PILV
import boxlcd_amd as B
from oracle import pyb2o
import replay as R

ENVS = [('Dropbox', None), ('Bounce2', None), ('Object2', [1, 0]), ('Urchin', None), ('LuxoBall', None), ('UrchinCubes', None), ('Crab', None)]
NPOSE = 10


def pil_render(env, o, width, height, mode):
  d = env.scene.desc
  WIDTH = float(d.world_w)
  rgb = mode == 'RGB'
  image = Image.new(mode, (width, height))
  draw = ImageDraw.Draw(image)
  draw.rectangle([0, 0, width, height], fill=(1, 1, 1) if rgb else 1)
  xf, shapes = o.body_xf()
  for i, spec in enumerate(env.scene.bodies):
    c1, c2 = R.OBJ_COL if spec.kind == 0 else R.ROBOT_COL
    color = tuple(int(255.0 * (1 - x)) for x in c1) if rgb else 0
    outline = tuple(int(255.0 * (1 - x)) for x in c2) if rgb else None
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
  a = np.asarray(image)
  return (255 - a) if rgb else a.astype(np.uint8)


def main():
  out = {}
  rng = np.random.RandomState(7)
  for name, sel in ENVS:
    env = getattr(B.envs, name)(raster_variant=1)
    d = env.scene.desc
    nb = d.n_bodies
    W, H = float(d.world_w), float(d.world_h)
    poses = np.zeros((NPOSE, nb, 3), np.float32)
    poses[..., 0] = rng.uniform(-0.3, W + 0.3, (NPOSE, nb))
    poses[..., 1] = rng.uniform(-0.3, H + 0.3, (NPOSE, nb))
    poses[..., 2] = rng.uniform(-np.pi, np.pi, (NPOSE, nb))
    poses[0, :, 2] = 0.0                                  # axis-aligned case
    sels = np.zeros((NPOSE, nb), np.int32) if sel is None else np.tile(np.array(sel, np.int32), (NPOSE, 1))
    sizes = [(8 * d.lcd_w, 8 * d.lcd_h, 'RGB'), (d.lcd_w, d.lcd_h, 'RGB'), (3 * d.lcd_w + 1, 2 * d.lcd_h + 3, 'RGB'), (5 * d.lcd_w, 4 * d.lcd_h, '1')]
    out[name + '_poses'], out[name + '_sel'] = poses, sels
    for si, (w, h, mode) in enumerate(sizes):
      frames = []
      for k in range(NPOSE):
        o = pyb2o.OracleEnv(d)
        o.reset(poses[k], sels[k])
        frames.append(pil_render(env, o, w, h, mode))
      out[f'{name}_size{si}'] = np.array([w, h, int(mode == 'RGB')], np.int32)
      out[f'{name}_frames{si}'] = np.stack(frames)
  out['pillow_version'] = np.array(PILV)
  np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'pillow_rgb.npz'), **out)
  print('Pillow', PILV, 'written', sum(v.nbytes for v in out.values()) // 1024, 'KiB raw')


if __name__ == '__main__':
  main()
