"""World description types and the robot catalogue.

Same public names and field meanings as the reference's `boxLCD/world_defs.py` (Object :11-23, Body :26-31,
Joint :33-41, Robot :43-52, WorldDef :55-59, ROBOT_FILLER :63-70, robot builders :78-445) so that
`WorldDef(robots=[Robot(type='urchin', name='urchin0')], objects=[Object('object0', ...)])` keeps working.

Differences in construction (not in content): there is no Box2D here, so shapes are plain records
(`CircleShape`, `PolygonShape`) that the scene compiler (boxlcd_amd/scene.py) lowers into the C-ABI's
`blcd_scene_desc`; robots are declared as compact part tables rather than one function per robot.
All dimensions reproduce the reference's float64 expressions operation by operation (they cross into the
float32 world only inside the scene compiler, like pybox2d's SWIG layer does).
"""
from typing import NamedTuple, List, Tuple, Dict
import numpy as np

SCALE = 30.0  # reference world_defs.py:8


class CircleShape(NamedTuple):
  radius: float


class PolygonShape(NamedTuple):
  box: Tuple[float, float] = None        # polygonShape(box=(hx, hy))  -> b2PolygonShape::SetAsBox
  vertices: Tuple = None                 # polygonShape(vertices=[..]) -> b2PolygonShape::Set (hull + centroid)


def circleShape(radius, pos=(0, 0)):
  return CircleShape(float(radius))


def polygonShape(box=None, vertices=None):
  if box is not None:
    return PolygonShape(box=(float(box[0]), float(box[1])))
  return PolygonShape(vertices=tuple((float(x), float(y)) for x, y in vertices))


class Object(NamedTuple):
  name: str
  shape: str = 'box'
  size: float = 0.5
  linearDamping: float = 0.0
  angularDamping: float = 0.0
  density: float = 1.0
  friction: float = 0.5
  restitution: float = 0.0
  categoryBits: int = 0x0110
  rand_angle: int = 1
  rangex: Tuple[float, float] = None
  rangey: Tuple[float, float] = None


class Body(NamedTuple):
  shape: object
  density: float = 1
  maskBits: int = 0x001
  categoryBits: int = 0x0020
  friction: float = 1.0


class Joint(NamedTuple):
  parent: str
  angle: float
  anchorA: list
  anchorB: list
  limits: List[float]
  limited: bool = True
  speed: float = 8
  torque: float = 150


class Robot(NamedTuple):
  type: str
  name: str
  root_body: Body = None
  bodies: Dict[str, Body] = None
  joints: Dict[str, Joint] = None
  rand_angle: int = 0
  angularDamping: float = 0
  linearDamping: float = 0
  bound: float = 1.5


class WorldDef(NamedTuple):
  robots: List[Robot] = []
  objects: List[Object] = []
  gravity: List[float] = [0, -9.81]
  forcetorque: int = 0


ROBOT_FILLER = {}


def register(name):
  def _reg(func):
    ROBOT_FILLER[name] = func
    return func
  return _reg


def _assemble(robot, root, parts, **kw):
  """parts: {link: (Body, Joint)} in creation order."""
  return Robot(type=robot.type, name=robot.name, root_body=root,
               bodies={k: b for k, (b, _) in parts.items()}, joints={k: j for k, (_, j) in parts.items()}, **kw)


def _box(w, h):
  return polygonShape(box=(w, h))


def _radial(robot, angles, rand_angle, bound, limited=True):
  """circle hub + identical legs hinged at the hub centre (urchin / quad / legs; reference :78-95, :129-165)."""
  LEG_W, LEG_H = 8 / SCALE, 40 / SCALE
  leg = _box(LEG_W / 2, LEG_H / 2)
  parts = {name: (Body(leg, maskBits=0x011, density=1.0), Joint('root', ang, (0, 0), (0, LEG_H / 2), [-1.0, 1.0], limited=limited))
           for name, ang in angles.items()}
  return _assemble(robot, Body(circleShape(radius=0.8 * LEG_W)), parts, rand_angle=rand_angle, bound=bound)


@register('urchin')
def make_urchin(robot, G):
  return _radial(robot, {'aleg': 0.0, 'bleg': 2.0, 'cleg': 4.2}, rand_angle=1, bound=1.25)


@register('quad')
def make_quad(robot, G):
  return _radial(robot, {'aleg': 0.0, 'bleg': 2.0, 'cleg': 4.2}, rand_angle=0, bound=1.5)


@register('legs')
def make_legs(robot, G):
  return _radial(robot, {'aleg': -1.0, 'bleg': 1.0}, rand_angle=0, bound=1.5)


@register('luxo')
def make_luxo(robot, G):
  """reference :97-124"""
  VERT, SIDE = 10 / SCALE, 5 / SCALE
  LEG_W, LEG_H, LL_H = 8 / SCALE, 24 / SCALE, 20 / SCALE
  poly = np.array([(-15, +15), (+20, +25), (+20, -25), (-15, -15)]) * 0.8
  root = Body(polygonShape(vertices=[(x / SCALE, y / SCALE) for x, y in poly]), density=0.1, maskBits=0x011)
  parts = {
      'lhip': (Body(_box(LEG_W / 2, LEG_H / 2), maskBits=0x011), Joint('root', -0.5, (-SIDE, -VERT), (0, LEG_H / 2), [-0.1, 0.1])),
      'lknee': (Body(_box(0.8 * LEG_W / 2, LL_H / 2), maskBits=0x011), Joint('lhip', 0.5, (0, -LEG_H / 2), (0, LL_H / 2), [-0.9, 0.9])),
      'lfoot': (Body(_box(LEG_H, LEG_W / 2), maskBits=0x011), Joint('lknee', 0.0, (0, -LEG_H / 2), (0, LEG_W / 2), [-0.5, 0.9])),
  }
  return _assemble(robot, root, parts, bound=2.0)


_HEX = [(-25, +0), (-20, +16), (+20, +16), (+25, +0), (+20, -16), (-20, -16)]


@register('crab')
def make_crab(robot, G):
  """reference :169-248 (16 links: 2 two-segment legs, 2 two-segment arms, 2x2 two-segment claws)"""
  VERT, SIDE = 12 / SCALE, 20 / SCALE
  LEG_W, LEG_H, LL_H = 8 / SCALE, 20 / SCALE, 20 / SCALE
  ARM_W, ARM_H = 8 / SCALE, 20 / SCALE
  CLAW_W, CLAW_H = 4 / SCALE, 16 / SCALE
  poly = 0.9 * np.array(_HEX)
  hip, knee = _box(LEG_W / 2, LEG_H / 2), _box(0.8 * LEG_W / 2, LL_H / 2)
  arm, claw = _box(ARM_W / 2, ARM_H / 2), _box(CLAW_W / 2, CLAW_H / 2)
  base, cm = 0x001, 0x011
  B = lambda s, m: Body(s, maskBits=m)
  bodies = {
      'lhip': B(hip, base), 'lknee': B(knee, base), 'rhip': B(hip, base), 'rknee': B(knee, base),
      'lshoulder': B(arm, cm), 'lelbow': B(arm, cm), 'rshoulder': B(arm, cm), 'relbow': B(arm, cm),
      'llclaw0': B(claw, cm), 'llclaw1': B(claw, cm), 'lrclaw0': B(claw, cm), 'lrclaw1': B(claw, cm),
      'rlclaw0': B(claw, cm), 'rlclaw1': B(claw, cm), 'rrclaw0': B(claw, cm), 'rrclaw1': B(claw, cm),
  }
  top, bot = (0, ARM_H / 2), (0, -ARM_H / 2)
  ctop, cbot = (0, CLAW_H / 2), (0, -CLAW_H / 2)
  joints = {
      'lhip': Joint('root', -0.5, (-SIDE, -VERT), (0, LEG_H / 2), [-1.5, 0.5]),
      'rhip': Joint('root', 0.5, (SIDE, -VERT), (0, LEG_H / 2), [0.5, 1.5]),
      'lknee': Joint('lhip', 0.5, (0, -LEG_H / 2), (0, LL_H / 2), [-0.5, 0.5]),
      'rknee': Joint('rhip', -0.5, (0, -LEG_H / 2), (0, LL_H / 2), [-0.5, 0.5]),
      'lshoulder': Joint('root', 2.0, (-SIDE, VERT), bot, [-3.0, 3.0], limited=False),
      'rshoulder': Joint('root', -2.0, (SIDE, VERT), bot, [-3.0, 3.0], limited=False),
      'lelbow': Joint('lshoulder', 3.0, top, bot, [-2.0, 2.0], limited=False),
      'relbow': Joint('rshoulder', -3.0, top, bot, [-2.0, 2.0], limited=False),
      'llclaw0': Joint('lelbow', 2.25, top, cbot, [-2.0, 1.0]),
      'llclaw1': Joint('llclaw0', 3.75, ctop, cbot, [0.0, 0.0]),
      'lrclaw0': Joint('lelbow', -2.25, top, cbot, [-1.0, 2.0]),
      'lrclaw1': Joint('lrclaw0', -3.75, ctop, cbot, [0.0, 0.0]),
      'rlclaw0': Joint('relbow', 2.25, top, cbot, [-2.0, 1.0]),
      'rlclaw1': Joint('rlclaw0', 3.75, ctop, cbot, [0.0, 0.0]),
      'rrclaw0': Joint('relbow', -2.25, top, cbot, [-1.0, 2.0]),
      'rrclaw1': Joint('rrclaw0', -3.75, ctop, cbot, [0.0, 0.0]),
  }
  root = Body(polygonShape(vertices=[(x / SCALE, y / SCALE) for x, y in poly]), density=1.0, maskBits=base, categoryBits=0x0020)
  # NB: the reference's `bodies` and `joints` dicts are ordered differently; creation follows the JOINT order (world_env.py:230)
  return Robot(type=robot.type, name=robot.name, root_body=root, bodies=bodies, joints=joints, bound=2.0)


@register('walker')
def make_walker(robot, G):
  """reference :251-297"""
  LEG_DOWN = -6 / SCALE
  LEG_W, LEG_H = 10 / SCALE, 24 / SCALE
  ARM_W, ARM_H = 8 / SCALE, 20 / SCALE
  CLAW_W, CLAW_H = 6 / SCALE, 16 / SCALE
  poly = 0.8 * np.array([(-30, +9), (+6, +9), (+34, +1), (+34, -8), (-30, -8)])
  hip, knee = _box(LEG_W / 2, LEG_H / 2), _box(0.8 * LEG_W / 2, LEG_H / 2)
  arm, claw = _box(ARM_W / 2, ARM_H / 2), _box(CLAW_W / 2, CLAW_H / 2)
  bodies = {
      'lhip': Body(hip), 'lknee': Body(knee), 'rhip': Body(hip), 'rknee': Body(knee),
      'shoulder': Body(arm, maskBits=0x001, density=0.1), 'elbow': Body(arm, maskBits=0x001, density=0.1),
      'lclaw0': Body(claw, maskBits=0x011, density=0.1), 'lclaw1': Body(claw, maskBits=0x011, density=0.1),
      'rclaw0': Body(claw, maskBits=0x011, density=0.1), 'rclaw1': Body(claw, maskBits=0x011, density=0.1),
  }
  top, bot = (0, ARM_H / 2), (0, -ARM_H / 2)
  ctop, cbot = (0, CLAW_H / 2), (0, -CLAW_H / 2)
  joints = {
      'lhip': Joint('root', 0.05, (0.0, LEG_DOWN), (0, LEG_H / 2), [-0.8, 1.1]),
      'lknee': Joint('lhip', 0.05, (0, -LEG_H / 2), (0, LEG_H / 2), [-1.6, -0.1]),
      'rhip': Joint('root', -0.05, (0.0, LEG_DOWN), (0, LEG_H / 2), [-0.8, 1.1]),
      'rknee': Joint('rhip', -0.05, (0, -LEG_H / 2), (0, LEG_H / 2), [-1.6, -0.1]),
      'shoulder': Joint('root', 2.0, (0, 5 / SCALE), bot, [-3.0, 3.0], limited=False),
      'elbow': Joint('shoulder', 3.0, top, bot, [-2.0, 2.0], limited=False),
      'lclaw0': Joint('elbow', 2.25, top, cbot, [-2.0, 1.0]),
      'lclaw1': Joint('lclaw0', 3.75, ctop, cbot, [0.0, 0.0]),
      'rclaw0': Joint('elbow', -2.25, top, cbot, [-1.0, 2.0]),
      'rclaw1': Joint('rclaw0', -3.75, ctop, cbot, [0.0, 0.0]),
  }
  root = Body(polygonShape(vertices=[(x / SCALE, y / SCALE) for x, y in poly]))
  return Robot(type=robot.type, name=robot.name, root_body=root, bodies=bodies, joints=joints)


@register('gingy')
def make_gingy(robot, G):
  """reference :300-335"""
  VERT, SIDE = 10 / SCALE, 2 / SCALE
  BODY_W, BODY_H = 8 / SCALE, 25 / SCALE
  ARM_W, ARM_H = 8 / SCALE, 25 / SCALE
  LEG_W, LEG_H = 8 / SCALE, 30 / SCALE
  trunk, arm, leg = _box(BODY_W / 2, BODY_H / 2), _box(ARM_W / 2, ARM_H / 2), _box(LEG_W / 2, LEG_H / 2)
  parts = {
      'body': (Body(trunk, density=1.0), Joint('root', 0.0, (0, -VERT), (0, BODY_H / 2), [-0.1, 0.1])),
      'larm': (Body(arm, maskBits=0x011), Joint('body', 1.5, (-SIDE, +VERT), (0, ARM_H / 2), [-1.5, 0.8])),
      'rarm': (Body(arm, maskBits=0x011), Joint('body', -1.5, (SIDE, +VERT), (0, ARM_H / 2), [-1.5, 0.8])),
      'llarm': (Body(arm, maskBits=0x011), Joint('larm', 1.5, (0, -ARM_H / 2), (0, ARM_H / 2), [-1.5, 1.5])),
      'rlarm': (Body(arm, maskBits=0x011), Joint('rarm', -1.5, (0, -ARM_H / 2), (0, ARM_H / 2), [-1.5, 1.5])),
      'lleg': (Body(leg, density=1.0), Joint('body', 0.8, (-SIDE, -VERT), (0, LEG_H / 2), [-0.2, 0.4])),
      'rleg': (Body(leg, density=1.0), Joint('body', -0.8, (SIDE, -VERT), (0, LEG_H / 2), [-0.4, 0.2])),
  }
  return _assemble(robot, Body(circleShape(radius=10 / SCALE), density=0.01), parts)


@register('octo')
def make_octo(robot, G):
  """reference :337-364"""
  LEG_W, LEG_H = 8 / SCALE, 25 / SCALE
  leg = _box(LEG_W / 2, LEG_H / 2)
  parts = {}
  for k, ang in zip('abcd', (0.0, 1.0, 2.0, 3.0)):
    parts[f'{k}leg1'] = (Body(leg, maskBits=0x011, density=1.0), Joint('root', ang, (0, 0), (0, LEG_H / 2), [-1.0, 1.0], limited=False))
  for k, ang in zip('abcd', (0.0, 1.0, 2.0, 3.0)):
    parts[f'{k}leg2'] = (Body(leg, maskBits=0x011, density=1.0),
                         Joint(f'{k}leg1', ang, (0, -LEG_H / 2), (0, LEG_H / 2), [-1.0, 1.0], limited=False))
  return _assemble(robot, Body(circleShape(radius=1.5 * LEG_W), density=0.1), parts, rand_angle=1)


@register('spider')
def make_spider(robot, G):
  """reference :367-445 (bodies without a joint — shoulder/elbow/claws — are never instantiated: world_env.py:230)"""
  VERT, SIDE = 8 / SCALE, 8 / SCALE
  LEG_W, LEG_H, LL_H = 6 / SCALE, 20 / SCALE, 20 / SCALE
  ARM_W, ARM_H = 6 / SCALE, 26 / SCALE
  CLAW_W, CLAW_H = 4 / SCALE, 22 / SCALE
  hip, knee = _box(LEG_W / 2, LEG_H / 2), _box(0.8 * LEG_W / 2, LL_H / 2)
  arm, claw = _box(ARM_W / 2, ARM_H / 2), _box(CLAW_W / 2, CLAW_H / 2)
  cm = 0x011
  bodies = {
      'lhip': Body(hip, maskBits=0x001), 'lknee': Body(knee, maskBits=0x001),
      'rhip': Body(hip, maskBits=0x001), 'rknee': Body(knee, maskBits=0x001),
      'ulhip': Body(arm, maskBits=cm, density=0.1), 'ulknee': Body(arm, maskBits=cm, density=0.1),
      'urhip': Body(arm, maskBits=cm, density=0.1), 'urknee': Body(arm, maskBits=cm, density=0.1),
      'shoulder': Body(arm, maskBits=cm, density=0.5), 'elbow': Body(arm, maskBits=cm, density=0.1),
      'lclaw0': Body(claw, maskBits=cm, density=0.1), 'rclaw0': Body(claw, maskBits=cm, density=0.1),
  }
  joints = {
      'lhip': Joint('root', -1.0, (-SIDE, -VERT), (0, LEG_H / 2), [-1.5, 0.5]),
      'rhip': Joint('root', 1.0, (SIDE, -VERT), (0, LEG_H / 2), [0.5, 1.5]),
      'lknee': Joint('lhip', 0.5, (0, -LEG_H / 2), (0, LL_H / 2), [-0.5, 0.5]),
      'rknee': Joint('rhip', -0.5, (0, -LEG_H / 2), (0, LL_H / 2), [-0.5, 0.5]),
      'ulhip': Joint('root', 1.5, (-SIDE, VERT), (0, -LEG_H / 2), [-1.5, 0.5]),
      'urhip': Joint('root', -1.5, (SIDE, VERT), (0, -LEG_H / 2), [0.5, 1.5]),
      'ulknee': Joint('ulhip', -0.5, (0, LEG_H / 2), (0, LL_H / 2), [-0.5, 0.5]),
      'urknee': Joint('urhip', 0.5, (0, LEG_H / 2), (0, LL_H / 2), [-0.5, 0.5]),
  }
  root = Body(circleShape(radius=10 / SCALE), density=1.0, maskBits=cm, categoryBits=0x0020)
  return Robot(type=robot.type, name=robot.name, root_body=root, bodies=bodies, joints=joints, bound=1.3)
