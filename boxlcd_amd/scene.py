"""Scene compiler: lowers a boxLCD `WorldDef` + config `G` into the C-ABI's `blcd_scene_desc` (include/boxlcd.h).

It restates, as data, what the reference does imperatively when it builds a world:
  * observation / action key tables                         — reference world_env.py:69-141
  * body creation order, fixture parameters, filter bits    — reference world_env.py:197-304
  * joint definitions                                        — reference world_env.py:255-267, world_defs.py:33-41
  * world size / LCD size / time step                        — reference world_env.py:144-166, 446-450, 467-469
Every float crosses into float32 here, which is where pybox2d's SWIG layer does it in the reference.
"""
import ctypes as C
import numpy as np
from . import utils
from .world_defs import CircleShape, PolygonShape, ROBOT_FILLER

MAX_POLY_VERTS, MAX_BODIES, MAX_JOINTS, MAX_SHAPES, MAX_OBS = 8, 20, 20, 24, 96
OBS_X, OBS_Y, OBS_COS_BODY, OBS_SIN_BODY, OBS_COS_XF, OBS_SIN_XF = range(6)
KIND_OBJECT, KIND_ROOT, KIND_LINK = 0, 1, 2


class ShapeDef(C.Structure):
  _fields_ = [('type', C.c_int32), ('n_verts', C.c_int32), ('radius', C.c_float), ('is_box', C.c_int32),
              ('verts', (C.c_float * 2) * MAX_POLY_VERTS)]


class BodyDef(C.Structure):
  _fields_ = [('n_choices', C.c_int32), ('shape', C.c_int32 * 2), ('density', C.c_float), ('friction', C.c_float),
              ('restitution', C.c_float), ('category_bits', C.c_uint32), ('mask_bits', C.c_uint32),
              ('linear_damping', C.c_float), ('angular_damping', C.c_float), ('kind', C.c_int32), ('_pad', C.c_int32)]


class JointDef(C.Structure):
  _fields_ = [('body_a', C.c_int32), ('body_b', C.c_int32), ('anchor_a', C.c_float * 2), ('anchor_b', C.c_float * 2),
              ('enable_limit', C.c_int32), ('lower', C.c_float), ('upper', C.c_float), ('max_motor_torque', C.c_float),
              ('speed', C.c_float), ('action_index', C.c_int32)]


class ObsDef(C.Structure):
  _fields_ = [('kind', C.c_int32), ('body', C.c_int32), ('lo', C.c_float), ('hi', C.c_float)]


class SceneDesc(C.Structure):
  _fields_ = [('n_bodies', C.c_int32), ('n_joints', C.c_int32), ('n_shapes', C.c_int32), ('n_obs', C.c_int32),
              ('n_act', C.c_int32), ('lcd_w', C.c_int32), ('lcd_h', C.c_int32), ('raster_variant', C.c_int32),
              ('world_w', C.c_float), ('world_h', C.c_float), ('gravity', C.c_float * 2), ('dt', C.c_float),
              ('substeps', C.c_int32), ('vel_iters', C.c_int32), ('pos_iters', C.c_int32),
              ('shapes', ShapeDef * MAX_SHAPES), ('bodies', BodyDef * MAX_BODIES), ('joints', JointDef * MAX_JOINTS),
              ('obs', ObsDef * MAX_OBS)]


class BodySpec:
  """Host-side record of one dynamic body in creation order (= Box2D proxy order after the 4 walls)."""

  def __init__(self, name, kind, index):
    self.name, self.kind, self.index = name, kind, index
    self.parent = None        # link: parent BodySpec
    self.joint = None         # link: world_defs.Joint
    self.robot = None
    self.obj = None


class CompiledScene:
  """`desc` (ctypes, passed to blcd_create) + the Python-side tables the env API needs."""

  def __init__(self):
    self.desc = SceneDesc()
    self.bodies = []          # [BodySpec]
    self.obs_info = {}
    self.act_info = {}
    self.obs_keys = []
    self.act_keys = []

  @property
  def body_index(self):
    return {b.name: b.index for b in self.bodies}


def fill_robots(world_def, G):
  """ROBOT_FILLER expansion (reference world_env.py:82-84); returns a new WorldDef with filled robots."""
  robots = [ROBOT_FILLER[r.type](r, G) if r.root_body is None else r for r in world_def.robots]
  return world_def._replace(robots=robots)


def _add_shape(desc, table, shape):
  key = shape
  if key in table:
    return table[key]
  i = len(table)
  if i >= MAX_SHAPES:
    raise ValueError('too many distinct shapes for blcd_scene_desc')
  sd = desc.shapes[i]
  if isinstance(shape, CircleShape):
    sd.type, sd.radius = 0, shape.radius
  else:
    sd.type = 1
    if shape.box is not None:
      sd.is_box, sd.n_verts = 1, 4
      sd.verts[0][0], sd.verts[0][1] = shape.box
    else:
      if len(shape.vertices) > MAX_POLY_VERTS:
        raise ValueError('polygon has too many vertices')
      sd.n_verts = len(shape.vertices)
      for k, (x, y) in enumerate(shape.vertices):
        sd.verts[k][0], sd.verts[k][1] = x, y
  table[key] = i
  return i


def compile_scene(world_def, G, WIDTH, HEIGHT, raster_variant=0):
  """world_def must already have its robots filled (fill_robots)."""
  from .world_defs import circleShape, polygonShape
  cs = CompiledScene()
  d = cs.desc
  shapes = {}
  obs_info, act_info = {}, {}
  # --- observation/action tables (reference world_env.py:72-117) ------------------------------------------
  for obj in world_def.objects:
    obs_info[f'{obj.name}:x:p'] = utils.A[0, WIDTH]
    obs_info[f'{obj.name}:y:p'] = utils.A[0, HEIGHT]
    obs_info[f'{obj.name}:cos'] = utils.A[-1, 1]
    obs_info[f'{obj.name}:sin'] = utils.A[-1, 1]
  for robot in world_def.robots:
    obs_info[f'{robot.name}:root:x:p'] = utils.A[0, WIDTH]
    obs_info[f'{robot.name}:root:y:p'] = utils.A[0, HEIGHT]
    obs_info[f'{robot.name}:root:cos'] = utils.A[-1, 1]
    obs_info[f'{robot.name}:root:sin'] = utils.A[-1, 1]
    for jname, joint in robot.joints.items():
      obs_info[f'{robot.name}:{jname}:x:p'] = utils.A[0, WIDTH]
      obs_info[f'{robot.name}:{jname}:y:p'] = utils.A[0, HEIGHT]
      obs_info[f'{robot.name}:{jname}:cos'] = utils.A[-1, 1]
      obs_info[f'{robot.name}:{jname}:sin'] = utils.A[-1, 1]
      if joint.limits[0] != joint.limits[1]:
        act_info[f'{robot.name}:{jname}:speed'] = utils.A[-1, 1]
  if len(world_def.robots) == 0:
    act_info['dummy'] = utils.A[-1, 1]
  cs.obs_info = utils.sortdict(obs_info)
  cs.act_info = utils.sortdict(act_info)
  cs.obs_keys = list(cs.obs_info.keys())
  cs.act_keys = list(cs.act_info.keys())

  # --- bodies in creation order: per robot root then links in JOINT order; then objects (world_env.py:200-304) ---
  def new_body(name, kind):
    if len(cs.bodies) >= MAX_BODIES:
      raise ValueError('too many bodies for blcd_scene_desc')
    b = BodySpec(name, kind, len(cs.bodies))
    cs.bodies.append(b)
    return b

  nj = 0
  for robot in world_def.robots:
    rb = robot.root_body
    root = new_body(robot.name + ':root', KIND_ROOT)
    root.robot = robot
    bd = d.bodies[root.index]
    bd.n_choices, bd.shape[0] = 1, _add_shape(d, shapes, rb.shape)
    bd.density = 1.0 if rb.density is None else rb.density
    bd.friction, bd.restitution = 1.0, 0.0                      # world_env.py:203 (friction hard-coded)
    bd.category_bits, bd.mask_bits = rb.categoryBits, rb.maskBits
    bd.linear_damping, bd.angular_damping, bd.kind = robot.linearDamping, robot.angularDamping, KIND_ROOT
    byname = {'root': root}
    for jname, joint in robot.joints.items():
      body = robot.bodies[jname]
      link = new_body(robot.name + ':' + jname, KIND_LINK)
      link.robot, link.joint, link.parent = robot, joint, byname[joint.parent]
      byname[jname] = link
      bd = d.bodies[link.index]
      bd.n_choices, bd.shape[0] = 1, _add_shape(d, shapes, body.shape)
      bd.density, bd.friction, bd.restitution = 1.0, body.friction, 0.0   # world_env.py:238 (density hard-coded)
      bd.category_bits, bd.mask_bits, bd.kind = body.categoryBits, body.maskBits, KIND_LINK
      if nj >= MAX_JOINTS:
        raise ValueError('too many joints for blcd_scene_desc')
      jd = d.joints[nj]
      jd.body_a, jd.body_b = link.parent.index, link.index
      jd.anchor_a[0], jd.anchor_a[1] = joint.anchorA
      jd.anchor_b[0], jd.anchor_b[1] = joint.anchorB
      jd.enable_limit = int(bool(joint.limited))
      jd.lower, jd.upper = joint.limits
      jd.max_motor_torque, jd.speed = joint.torque, joint.speed
      akey = f'{robot.name}:{jname}:speed'
      jd.action_index = cs.act_keys.index(akey) if akey in cs.act_info else -1
      nj += 1
  for obj in world_def.objects:
    ob = new_body(obj.name, KIND_OBJECT)
    ob.obj = obj
    bd = d.bodies[ob.index]
    choices = {'circle': circleShape(radius=obj.size, pos=(0, 0)), 'box': polygonShape(box=(obj.size, obj.size))}
    names = list(choices.keys()) if obj.shape == 'random' else [obj.shape]   # world_env.py:273-274
    bd.n_choices = len(names)
    for k, nm in enumerate(names):
      bd.shape[k] = _add_shape(d, shapes, choices[nm])
    bd.density, bd.friction, bd.restitution = obj.density, obj.friction, obj.restitution
    bd.category_bits, bd.mask_bits = obj.categoryBits, 0xFFFF                # fixtureDef default maskBits
    bd.linear_damping, bd.angular_damping, bd.kind = obj.linearDamping, obj.angularDamping, KIND_OBJECT

  # --- observation lowering (reference world_env.py:387-425) ----------------------------------------------
  bidx = cs.body_index
  if len(cs.obs_keys) > MAX_OBS:
    raise ValueError('too many observation keys for blcd_scene_desc')
  for i, key in enumerate(cs.obs_keys):
    parts = key.split(':')
    if parts[0] in bidx:                                     # object key: '<obj>:x:p' / '<obj>:cos'
      bname, field, is_link = parts[0], parts[1], False
    else:
      bname, field = parts[0] + ':' + parts[1], parts[2]
      is_link = parts[1] != 'root'
    od = d.obs[i]
    od.body = bidx[bname]
    lo, hi = cs.obs_info[key]
    od.lo, od.hi = float(lo), float(hi)
    if field == 'x':
      od.kind = OBS_X
    elif field == 'y':
      od.kind = OBS_Y
    elif field == 'cos':
      od.kind = OBS_COS_XF if is_link else OBS_COS_BODY     # links: transform.angle, roots/objects: body.angle
    elif field == 'sin':
      od.kind = OBS_SIN_XF if is_link else OBS_SIN_BODY
    else:
      raise ValueError(key)

  d.n_bodies, d.n_joints, d.n_shapes = len(cs.bodies), nj, len(shapes)
  d.n_obs, d.n_act = len(cs.obs_keys), len(cs.act_keys)
  d.lcd_h = int(G.lcd_base)
  d.lcd_w = int(G.lcd_base * G.wh_ratio)
  d.raster_variant = int(raster_variant)
  d.world_w, d.world_h = float(WIDTH), float(HEIGHT)
  d.gravity[0], d.gravity[1] = world_def.gravity
  if G.fps < 30:                                              # world_env.py:446-452
    d.dt, d.substeps = 1.0 / (G.fps * 3), 3
  else:
    d.dt, d.substeps = 1.0 / G.fps, 1
  d.vel_iters, d.pos_iters = 6 * 30, 2 * 30
  return cs
