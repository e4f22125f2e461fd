"""Score the CPU oracle against the reference's recordings replayed from the recorder's own inputs (tests/replay.py), under
the default oracle variant set and — with --table — under each single-switch alternative (the table in DESIGN.md §2).
usage: python tools/replay_gifs.py [--table] [-v] [names...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import replay as R
from oracle import pyb2o


def line(gif, **variants):
  with pyb2o.variants(**variants):
    bl, br = R.replay(gif, pyb2o)
  sl, sr = R.summary(bl), R.summary(br)
  return sl, sr, bl, br


if __name__ == '__main__':
  names = [a for a in sys.argv[1:] if not a.startswith('-')] or list(R.GIFS)
  if '--table' in sys.argv:
    sets = [('default (Box2D 2.3.0 forms, glibc<=2.27 sincos)', {}), ('sincos: glibc>=2.28', {'sincos': 0}),
            ('damping: Pade (>=2.3.1)', {'damping': 0}), ('Sweep::Advance: increment form (>=2.3.1)', {'advance': 0}),
            ('polygons: brute-force FindMaxSeparation + k_tol (>=2.3.1)', {'polygons': 0}),
            ('round-1 oracle (all four of the above)', {'sincos': 0, 'damping': 0, 'advance': 0, 'polygons': 0})]
    print('| variant set | ' + ' | '.join(names) + ' |')
    print('|---|' + '---|' * len(names))
    for label, kw in sets:
      cells = []
      for g in names:
        sl, sr, _, _ = line(g, **kw)
        cells.append(f"{sl['exact']}/{sl['frames']} · {sr['exact']}/{sr['frames']}")
      print(f'| {label} | ' + ' | '.join(cells) + ' |', flush=True)
  else:
    for g in names:
      sl, sr, bl, br = line(g)
      print(f"{g:16s} LCD exact {sl['exact']}/{sl['frames']} first-miss {sl['first_miss']} px {sl['px']} | "
            f"RGB8x exact {sr['exact']}/{sr['frames']} first-miss {sr['first_miss']} px {sr['px']}")
      if '-v' in sys.argv:
        print('   lcd', [(i, b) for i, b in enumerate(bl) if b][:20])
        print('   rgb', [(i, b) for i, b in enumerate(br) if b][:20])
