"""Bounded sweep of SCENE PARAMETERS against the four recordings the oracle reproduces only for a prefix.

The in-tree recorder (reference research/scripts/evaluations/demo_imgs.py:22-39) cannot construct a Luxo env at all, so the three
Luxo recordings were made by another script, possibly against other `make_luxo` / object constants (reference
boxLCD/world_defs.py:33-41,97-124; envs.py:63-64 - which still carries two commented-out cube settings and one commented-out ball).
Geometry is pinned by frame 0 and the exact prefix; what can differ without moving the first 33-66 frames is everything that acts
only later: joint limits / motor torque / motor speed, densities, friction, restitution, damping, gravity.

Score of a candidate = number of leading frames whose 8x RGB view (25.6-27.4 px/unit, Pillow on the oracle's transforms,
tests/replay.py) is pixel-exact, and the LCD / RGB exact-frame totals.  A candidate "extends" a recording if its prefix beats the
default's.  usage: python tools/param_sweep.py [--md profiles/r04_param_sweep.md]
"""
import os
import sys
import ctypes
import itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import replay as R
import boxlcd_amd as B
from oracle import pyb2o


def clone(d):
  return type(d).from_buffer_copy(d)


def score(gif, mod=None, act_scale=1.0, full=False):
  """(rgb prefix, rgb exact, lcd exact, frames) of the oracle on `gif` with the scene description modified by mod(desc, env)"""
  cls, force_sel, seed, aseed = R.GIFS[gif]
  env = getattr(B.envs, cls)(raster_variant=2)
  rgb, lcd = R.fixtures(gif)
  P, sel = R.recorder_start(env, seed)
  if force_sel is not None:
    sel = np.array(force_sel, np.int32)
  d = clone(env.scene.desc)
  if mod is not None:
    mod(d, env)
  o = pyb2o.OracleEnv(d)
  o.reset(np.asarray(P, np.float32), sel)
  rs = np.random.RandomState(aseed)
  prefix, rgb_ok, lcd_ok, broken = 0, 0, 0, False
  T = len(lcd)
  for t in range(T):
    a = rs.uniform(-1, 1, env.act_size) * act_scale
    o.step(a.astype(np.float32))
    l_ok = bool((o.render() == lcd[t]).all())
    r_ok = bool((R.pil_rgb(env, o) == rgb[t]).all())
    lcd_ok += l_ok
    rgb_ok += r_ok
    if r_ok and not broken:
      prefix += 1
    else:
      if not broken and not full and t > 8 and prefix <= t - 8:
        pass
      broken = True
    if broken and not full and t - prefix >= 12:
      break          # 12 frames past the first miss: the totals of clearly broken candidates are not needed
  return prefix, rgb_ok, lcd_ok, T


# ---- candidate modifications -----------------------------------------------------------------------------------------------
def joints(d):
  return [d.joints[i] for i in range(d.n_joints)]


def robot_bodies(d):
  return [d.bodies[i] for i in range(d.n_bodies) if d.bodies[i].kind != 0]


def objects(d):
  return [d.bodies[i] for i in range(d.n_bodies) if d.bodies[i].kind == 0]


def cands(gif):
  """(label, mod) pairs: one parameter family at a time, then pairs of the families that did not break the prefix"""
  C = []

  def add(label, fn):
    C.append((label, fn))

  nj = 3
  for v in (2, 4, 5, 6, 7, 7.5, 8.5, 9, 10, 12, 16):
    add(f'Joint.speed={v} (all joints)', lambda d, e, v=v: [setattr(j, 'speed', v) for j in joints(d)])
  for v in (20, 50, 75, 100, 125, 140, 160, 175, 200, 300, 1000):
    add(f'Joint.torque={v} (all joints)', lambda d, e, v=v: [setattr(j, 'max_motor_torque', v) for j in joints(d)])
  for k in range(nj):
    for v in (50, 100, 200, 300):
      add(f'joint[{k}].torque={v}', lambda d, e, k=k, v=v: setattr(joints(d)[k], 'max_motor_torque', v))
  # limits: the robot builders' values are (lhip +-0.1, lknee +-0.9, lfoot -0.5..0.9) for luxo, +-1.0 for urchin
  for k in range(nj):
    for dl, du in itertools.product((-0.2, -0.1, -0.05, 0.0, 0.05, 0.1, 0.2), repeat=2):
      if dl == 0.0 and du == 0.0:
        continue
      add(f'joint[{k}].limits += ({dl:+.2f}, {du:+.2f})', lambda d, e, k=k, dl=dl, du=du: (setattr(joints(d)[k], 'lower', joints(d)[k].lower + dl), setattr(joints(d)[k], 'upper', joints(d)[k].upper + du)))
    add(f'joint[{k}].limited=False', lambda d, e, k=k: setattr(joints(d)[k], 'enable_limit', 0))
  for s in (0.5, 0.8, 0.9, 1.1, 1.25, 1.5, 2.0):
    add(f'all limits x {s}', lambda d, e, s=s: [(setattr(j, 'lower', j.lower * s), setattr(j, 'upper', j.upper * s)) for j in joints(d)])
  # densities / friction / restitution of the robot's parts
  for v in (0.05, 0.2, 0.25, 0.5, 1.0):
    add(f'root density={v}', lambda d, e, v=v: setattr(robot_bodies(d)[0], 'density', v))
  for v in (0.1, 0.5, 0.8, 1.25, 2.0):
    add(f'link density={v} (all links)', lambda d, e, v=v: [setattr(b, 'density', v) for b in robot_bodies(d)[1:]])
  for k in range(1, 4):
    for v in (0.5, 2.0):
      add(f'link[{k}] density={v}', lambda d, e, k=k, v=v: setattr(robot_bodies(d)[k], 'density', v))
  for v in (0.2, 0.5, 0.8, 1.5, 2.0):
    add(f'robot friction={v}', lambda d, e, v=v: [setattr(b, 'friction', v) for b in robot_bodies(d)])
  for v in (0.1, 0.2, 0.5):
    add(f'robot restitution={v}', lambda d, e, v=v: [setattr(b, 'restitution', v) for b in robot_bodies(d)])
  for v in (0.01, 0.05, 0.1, 0.5):
    add(f'robot angularDamping={v}', lambda d, e, v=v: [setattr(b, 'angular_damping', v) for b in robot_bodies(d)])
    add(f'robot linearDamping={v}', lambda d, e, v=v: [setattr(b, 'linear_damping', v) for b in robot_bodies(d)])
    add(f'root angularDamping={v}', lambda d, e, v=v: setattr(robot_bodies(d)[0], 'angular_damping', v))
    add(f'root linearDamping={v}', lambda d, e, v=v: setattr(robot_bodies(d)[0], 'linear_damping', v))
  # collision filter: links that also collide with each other / root that does not see objects
  add('robot maskBits=0x031 (self-collision)', lambda d, e: [setattr(b, 'mask_bits', 0x031) for b in robot_bodies(d)])
  add('root maskBits=0x001 (urchin-style: root ignores objects)', lambda d, e: setattr(robot_bodies(d)[0], 'mask_bits', 0x001))
  # objects (reference envs.py:61-64: two commented-out cube settings, one commented-out ball)
  if objects(env_probe(gif)):
    for v in (0.05, 0.1, 0.25, 0.3, 0.4, 0.5, 1.0):
      add(f'object density={v}', lambda d, e, v=v: [setattr(b, 'density', v) for b in objects(d)])
    for v in (0.0, 0.5, 0.7, 0.9, 1.0):
      add(f'object restitution={v}', lambda d, e, v=v: [setattr(b, 'restitution', v) for b in objects(d)])
    for v in (0.2, 0.3, 1.0):
      add(f'object friction={v}', lambda d, e, v=v: [setattr(b, 'friction', v) for b in objects(d)])
    for lin, ang in ((0.0, 0.0), (0.5, 0.2), (1.0, 0.0), (1.0, 1.0), (2.0, 0.2), (5.0, 1.0), (1.0, 0.1), (1.0, 0.5)):
      add(f'object damping=({lin}, {ang})', lambda d, e, lin=lin, ang=ang: [(setattr(b, 'linear_damping', lin), setattr(b, 'angular_damping', ang)) for b in objects(d)])
    add('cube as commented-out setting 1 (density .25, damping 1.0/0.2)', lambda d, e: [(setattr(b, 'density', 0.25), setattr(b, 'linear_damping', 1.0), setattr(b, 'angular_damping', 0.2)) for b in objects(d)])
    add('cube as commented-out setting 2 (density .1, damping 5.0/1.0)', lambda d, e: [(setattr(b, 'density', 0.1), setattr(b, 'linear_damping', 5.0), setattr(b, 'angular_damping', 1.0)) for b in objects(d)])
  # world
  for g in (-9.8, -9.80665, -10.0, -9.0):
    add(f'gravity y={g}', lambda d, e, g=g: d.gravity.__setitem__(1, g))
  for it in (30, 60, 90, 120, 150, 179, 181, 200):
    add(f'velocity iterations={it}', lambda d, e, it=it: setattr(d, 'vel_iters', it))
  for it in (1, 3, 10, 20, 30):
    add(f'position iterations={it}', lambda d, e, it=it: setattr(d, 'pos_iters', it))
  return C


_probe = {}


def env_probe(gif):
  if gif not in _probe:
    _probe[gif] = getattr(B.envs, R.GIFS[gif][0])(raster_variant=2).scene.desc
  return _probe[gif]


if __name__ == '__main__':
  md = sys.argv[sys.argv.index('--md') + 1] if '--md' in sys.argv else None
  gifs = [a for a in sys.argv[1:] if a in R.GIFS] or ['Urchin', 'Luxo', 'LuxoBall', 'LuxoCube']
  out = []
  for gif in gifs:
    base = score(gif, full=True)
    print(f'== {gif}: default prefix {base[0]}, RGB exact {base[1]}/{base[3]}, LCD exact {base[2]}/{base[3]}', flush=True)
    rows = []
    for label, fn in cands(gif):
      s = score(gif, fn)
      rows.append((label, s))
      if s[0] > base[0]:
        s = score(gif, fn, full=True)
        rows[-1] = (label, s)
        print(f'   EXTENDS: {label}: prefix {s[0]} rgb {s[1]} lcd {s[2]}', flush=True)
    # action scale (world_env.py:441: speed * clip(a)); a recorder that fed other action magnitudes
    for sc in (0.5, 0.9, 1.1, 2.0):
      s = score(gif, None, act_scale=sc)
      rows.append((f'actions x {sc}', s))
    out.append((gif, base, rows))
    better = [(l, s) for l, s in rows if s[0] > base[0]]
    print(f'   {len(rows)} candidates, {len(better)} extend the prefix; best prefix {max(s[0] for _, s in rows)}', flush=True)
  if md:
    with open(md, 'w') as f:
      f.write('# Scene-parameter sweep against the four open recordings (round 4)\n\n')
      f.write(__doc__.split('usage:')[0].strip() + '\n\n')
      for gif, base, rows in out:
        f.write(f'## {gif}: default prefix {base[0]} frames (8x RGB exact {base[1]}/{base[3]}, LCD exact {base[2]}/{base[3]})\n\n')
        better = [(l, s) for l, s in rows if s[0] > base[0]]
        same = [(l, s) for l, s in rows if s[0] == base[0]]
        f.write(f'{len(rows)} candidates: {len(better)} extend the prefix, {len(same)} leave it unchanged, {len(rows) - len(better) - len(same)} shorten it.\n\n')
        f.write('| candidate | exact prefix (8x RGB) | note |\n|---|---|---|\n')
        for l, s in sorted(rows, key=lambda x: -x[1][0]):
          note = 'EXTENDS' if s[0] > base[0] else ('unchanged' if s[0] == base[0] else '')
          tot = f' (RGB {s[1]}/{s[3]}, LCD {s[2]}/{s[3]})' if s[0] > base[0] else ''
          f.write(f'| {l} | {s[0]}{tot} | {note} |\n')
        f.write('\n')
