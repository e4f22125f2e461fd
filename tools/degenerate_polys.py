"""Every polygon of the exactly-replayed recording frames that truncates to a zero-area (degenerate) integer polygon, with what the
recording's Pillow did with it.

For each recording: replay the oracle from the recorder's inputs (tests/replay.py); for every frame inside the physics-exact prefix
(8x RGB view exact) and every polygon body: LCD-scale vertices float64(float32 world) / WIDTH * lcd_w truncated toward zero
(world_env.py:500-505).  Degenerate = all integer vertices on one row or on one column.  "drawn" = the pixels of
the polygon's horizontal (vertical) extent that no OTHER body covers are set in the recording's LCD frame.
NOTE: the table in profiles/r04_param_sweep.md was generated BEFORE raster variant 2 learnt the rule this table led to (horizontal edges are
drawn from a per-row scan position that starts at 0).  "Pixels no other body covers" are taken from the oracle's own variant-2 frame, which now
contains the two drawn feet - so today the tool lists the nine polygons the rule leaves undrawn and no longer the two it draws.
usage: python tools/degenerate_polys.py [--md file]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import replay as R
import boxlcd_amd as B
from oracle import pyb2o


def run(gif):
  cls, force_sel, seed, aseed = R.GIFS[gif]
  env = getattr(B.envs, cls)(raster_variant=2)
  rgb, lcd = R.fixtures(gif)
  P, sel = R.recorder_start(env, seed)
  if force_sel is not None:
    sel = np.array(force_sel, np.int32)
  d = env.scene.desc
  o = pyb2o.OracleEnv(d)
  o.reset(np.asarray(P, np.float32), sel)
  rs = np.random.RandomState(aseed)
  W, H, WIDTH = d.lcd_w, d.lcd_h, float(d.world_w)
  rows = []
  for t in range(len(lcd)):
    o.step(rs.uniform(-1, 1, env.act_size).astype(np.float32))
    if (R.pil_rgb(env, o) != rgb[t]).any():
      break                                   # physics-exact prefix only: beyond it the poses are not the recording's
    xf, shapes = o.body_xf()
    ours = o.render()
    for i, (kind, val) in enumerate(shapes):
      if kind != 'poly':
        continue
      pts = (val.astype(np.float64) / WIDTH * W)
      ip = np.trunc(pts).astype(int)
      horiz, vert = len(set(ip[:, 1])) == 1, len(set(ip[:, 0])) == 1
      if not (horiz or vert):
        continue
      # pixels of the extent, in image coordinates after FLIP_TOP_BOTTOM
      xs, ys = ip[:, 0], ip[:, 1]
      px = [(x, H - 1 - ys[0]) for x in range(xs.min(), xs.max() + 1)] if horiz else [(xs[0], H - 1 - y) for y in range(ys.min(), ys.max() + 1)]
      px = [(x, y) for x, y in px if 0 <= x < W and 0 <= y < H]
      # pixels no other body covers: render the frame without this body by comparing with the oracle's variant-2 frame (which draws
      # nothing for a degenerate polygon): a pixel of the extent that the oracle leaves background is covered by nobody else
      free = [(x, y) for x, y in px if ours[y, x] == 1]
      if not free:
        continue
      drawn = sum(int(lcd[t][y, x] == 0) for x, y in free)
      rows.append(dict(gif=gif, frame=t, body=env.scene.bodies[i].name if hasattr(env.scene.bodies[i], 'name') else i, verts=[tuple(v) for v in ip.tolist()],
                       kind='horizontal' if horiz and not vert else ('vertical' if vert and not horiz else 'point'), free_px=len(free), drawn_px=drawn,
                       first_at_origin=tuple(ip[0]) == (0, 0), touches_x0=int(xs.min()) == 0, touches_y0=int(ys.min()) == 0,
                       raw=[(round(float(a), 3), round(float(b), 3)) for a, b in pts]))
  return rows, t


if __name__ == '__main__':
  md = sys.argv[sys.argv.index('--md') + 1] if '--md' in sys.argv else None
  allrows = []
  for gif in R.GIFS:
    rows, t = run(gif)
    allrows += rows
    print(f'{gif}: exact prefix {t} frames, {len(rows)} degenerate polygons with uncovered pixels', flush=True)
  lines = ['| recording | frame | body | truncated vertices (Pillow order) | kind | uncovered px | of them set in the recording | first vertex (0,0) | xmin = 0 | ymin = 0 (bottom row) | LCD-scale vertices |', '|---|---|---|---|---|---|---|---|---|---|---|']
  for r in allrows:
    lines.append(f"| {r['gif']} | {r['frame']} | {r['body']} | {r['verts']} | {r['kind']} | {r['free_px']} | {r['drawn_px']} | {r['first_at_origin']} | {r['touches_x0']} | {r['touches_y0']} | {r['raw']} |")
  print('\n'.join(lines))
  if md:
    open(md, 'a').write('\n## Degenerate polygons of the exactly-replayed frames\n\n' + __doc__.split('usage:')[0].strip() + '\n\n' + '\n'.join(lines) + '\n')
