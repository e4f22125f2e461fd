"""Decode the reference's published demo GIFs into LCD-frame fixtures (tests/golden/gif_lcd_frames.npz).

Source: /root/reference/assets/envs/*.gif (README.md:74-85) — recordings of env.render(mode='human', return_pyglet_view=True)
(boxLCD/world_env.py:521-535): left half = 8x RGB render, 2-px separator, right half = LCD upscaled x8.
Decode rule: LCD pixel (r, c) = gif[8r, Wh + 2 + 8c] > 127 with Wh = width/2.
These are DATA produced by the reference stack (pybox2d 2.3.10 + Pillow of early 2021), the only reference outputs in its tree.
Run in the authoring container only:  python tools/gen_gif_fixtures.py
"""
import glob, os
import numpy as np
from PIL import Image, ImageSequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PALETTE = [(254, 254, 254), (128, 102, 230), (77, 77, 128), (230, 102, 102), (128, 77, 128)]   # bg, object fill/outline, robot fill/outline
out = {}
rgb8 = {}
for path in sorted(glob.glob('/root/reference/assets/envs/*.gif')):
  name = os.path.splitext(os.path.basename(path))[0].replace('-', '_')
  im = Image.open(path)
  frames = []
  halves = []
  for fr in ImageSequence.Iterator(im):
    c = np.asarray(fr.convert('RGB'))
    half = c[:, :c.shape[1] // 2]
    idx = np.full(half.shape[:2], 255, np.uint8)
    for k, col in enumerate(PALETTE):
      idx[(half == np.array(col, np.uint8)).all(-1)] = k
    assert (idx != 255).all(), (name, 'unexpected colour in the 8x view')
    halves.append(idx)
    g = np.asarray(fr.convert('L'))
    H, Wg = g.shape
    Wh = Wg // 2
    lw = Wh // 8
    lcd = g[0::8, Wh + 2::8][:16, :lw] > 127
    assert lcd.shape == (16, lw), (name, lcd.shape)
    # block uniformity check (each LCD pixel is an 8x8 block; last column clipped to 6 px)
    blk = g[:, Wh + 2:]
    for r in (0, 5, 15):
      for c in (0, lw // 2, lw - 1):
        b = blk[8 * r:8 * r + 8, 8 * c:8 * c + 6] > 127
        assert b.all() or not b.any(), (name, r, c)
    frames.append(np.packbits(lcd, axis=-1))
  out[name] = np.stack(frames)
  rgb8[name] = np.stack(halves)
  print(name, out[name].shape, rgb8[name].shape)
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'gif_lcd_frames.npz'), **out)
# the left half of every frame = lcd_render(8W, 8H, 'RGB') of the same state (world_env.py:525), as palette indices
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'gif_rgb8_frames.npz'), palette=np.array(PALETTE, np.uint8), **rgb8)
