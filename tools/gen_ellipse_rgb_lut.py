"""Dump Pillow's ellipse span tables for near-round bounding boxes (fill, and the 1-px outline) into a binary table.

`lcd_render(width, height, 'RGB')` draws circles with `draw.ellipse(bbox, fill=color, outline=outline)` (reference
boxLCD/world_env.py:493-499, 481-483): Pillow fills the ellipse and then draws its outline with width 1.  The bbox is the
int-truncation of (pos -+ rad) / WIDTH * width on both axes, so b = y1-y0 differs from a = x1-x0 by at most one; the pixel
pattern depends only on (a, b) and is confined to the bbox.  For 0 <= a <= AMAX and b-a in {-2..+2} (a bbox that straddles pixel 0 on one axis truncates toward zero there)
this script records per row: the fill span and the (at most two) outline spans, all inclusive, offsets from x0; 255 = none.
Layout: uint8 [AMAX+1][5][AMAX+3][6] = (fill_s, fill_t, o1_s, o1_t, o2_s, o2_t).  It is DATA (Pillow's output), used by both
the oracle and the HIP product (each loads the file; neither includes the other's code).

Run:  python tools/gen_ellipse_rgb_lut.py   (needs Pillow; writes boxlcd_amd/ellipse_rgb_lut.bin and oracle/ellipse_rgb_lut.bin)
"""
import os
import numpy as np
from PIL import Image, ImageDraw, __version__ as PILV

AMAX = 100
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pattern(a, b, fill, x0=4, y0=4):
  size = AMAX + 14
  im = Image.new('L', (size, size))
  d = ImageDraw.Draw(im)
  if fill:
    d.ellipse([x0, y0, x0 + a, y0 + b], fill=255)
  else:
    d.ellipse([x0, y0, x0 + a, y0 + b], fill=None, outline=255)
  p = np.asarray(im) > 0
  ys, xs = np.nonzero(p)
  if len(ys):
    assert ys.min() >= y0 and ys.max() <= y0 + b and xs.min() >= x0 and xs.max() <= x0 + a, (a, b)
  return p[y0:y0 + b + 1, x0:x0 + a + 1]


def runs(row):
  out, x = [], 0
  n = len(row)
  while x < n:
    if row[x]:
      s = x
      while x < n and row[x]:
        x += 1
      out.append((s, x - 1))
    else:
      x += 1
  return out


def main():
  lut = np.full((AMAX + 1, 5, AMAX + 3, 6), 255, np.uint8)
  for a in range(AMAX + 1):
    for k, db in enumerate((-2, -1, 0, 1, 2)):
      b = a + db
      if b < 0:
        continue
      pf, po = pattern(a, b, True), pattern(a, b, False)
      for r in range(b + 1):
        rf, ro = runs(pf[r]), runs(po[r])
        assert len(rf) <= 1 and len(ro) <= 2, (a, b, r, rf, ro)
        if rf:
          lut[a, k, r, 0:2] = rf[0]
        for q, (s, t) in enumerate(ro):
          lut[a, k, r, 2 + 2 * q:4 + 2 * q] = (s, t)
  # translation invariance + fill/outline composition on a few random placements
  rng = np.random.RandomState(0)
  for _ in range(200):
    a = int(rng.randint(0, AMAX + 1)); db = int(rng.randint(-2, 3)); b = max(a + db, 0)
    x0, y0 = int(rng.randint(0, 9)), int(rng.randint(0, 9))
    im = Image.new("L", (AMAX + 14, AMAX + 14)); d = ImageDraw.Draw(im)
    d.ellipse([x0, y0, x0 + a, y0 + b], fill=100, outline=200)
    got = np.asarray(im)
    exp = np.zeros_like(got)
    for r in range(b + 1):
      e = lut[a, b - a + 2, r]
      if e[0] != 255: exp[y0 + r, x0 + e[0]:x0 + e[1] + 1] = 100
      for q in (2, 4):
        if e[q] != 255: exp[y0 + r, x0 + e[q]:x0 + e[q + 1] + 1] = 200
    assert (got == exp).all(), (a, b, x0, y0)
  for d_ in ('boxlcd_amd', 'oracle'):
    lut.tofile(os.path.join(ROOT, d_, 'ellipse_rgb_lut.bin'))
  print('Pillow', PILV, 'AMAX', AMAX, 'bytes', lut.nbytes)


if __name__ == '__main__':
  main()
