"""Named environments: `envs.Dropbox()`, `envs.Bounce()`, `envs.Urchin()`, `envs.LuxoBall()`, `envs.Object2()` ...

Same constructors and per-env default overrides as the reference's `boxLCD/envs.py` (cc decorator :5-14, classes
:17-137); here the catalogue is a table and the classes are generated from it.  Every class takes `G` (dict or
Namespace) overriding `ENV_DG` keys, exactly like `WorldEnv.__init__` (reference world_env.py:47-61).
"""
from .world_env import WorldEnv
from .world_defs import WorldDef, Object, Robot
from . import utils


def cc(**kwargs):
  """class decorator: custom default config (reference envs.py:5-14)."""
  def decorator(Cls):
    class CustomWorldEnv(Cls):
      ENV_DG = utils.AttrDict(WorldEnv.ENV_DG)
      for key in kwargs:
        ENV_DG[key] = kwargs[key]
    CustomWorldEnv.__name__ = Cls.__name__
    CustomWorldEnv.__qualname__ = Cls.__qualname__
    return CustomWorldEnv
  return decorator


_BALL = dict(shape='circle', size=0.5, density=0.2, restitution=0.8)                            # reference envs.py:64
_CUBE = dict(shape='box', size=0.4, density=0.5, linearDamping=1.0, angularDamping=0.2)         # reference envs.py:63
_BOUNCY = dict(size=0.5, density=0.1, restitution=0.8)

# name: (ENV_DG overrides, robot type or None, [object kwargs])
_CATALOGUE = {
    'Dropbox': (dict(ep_len=25, wh_ratio=1.0), None, [dict(shape='box', size=0.7, density=0.1)]),
    'Bounce': (dict(ep_len=50, wh_ratio=1.0), None, [dict(shape='circle', **_BOUNCY)]),
    'Bounce2': (dict(ep_len=50, wh_ratio=1.0), None, [dict(shape='circle', **_BOUNCY)] * 2),
    'Object2': (dict(ep_len=50, wh_ratio=1.0), None, [dict(shape='random', **_BOUNCY)] * 2),
    'Object3': (dict(ep_len=50, wh_ratio=1.0), None, [dict(shape='random', **_BOUNCY)] * 3),
    'Urchin': (dict(ep_len=100), 'urchin', []),
    'Luxo': (dict(ep_len=100), 'luxo', []),
    'UrchinCube': (dict(ep_len=150, wh_ratio=1.5), 'urchin', [_CUBE]),
    'LuxoCube': (dict(ep_len=150, wh_ratio=1.5), 'luxo', [_CUBE]),
    'UrchinBall': (dict(ep_len=150, wh_ratio=1.5), 'urchin', [_BALL]),
    'LuxoBall': (dict(ep_len=150, wh_ratio=1.5), 'luxo', [_BALL]),
    'UrchinBalls': (None, 'urchin', [_BALL] * 3),
    'LuxoBalls': (None, 'luxo', [_BALL] * 3),
    'UrchinCubes': (None, 'urchin', [_CUBE] * 3),
    'LuxoCubes': (None, 'luxo', [_CUBE] * 3),
    'Crab': (dict(lcd_base=32), 'crab', []),
    'CrabCube': (dict(lcd_base=32), 'crab', [dict(shape='box', size=0.4, density=1.0, friction=1.0)]),
    'SpiderCube': (dict(lcd_base=32), 'spider', [dict(shape='box', size=0.3, density=0.1, friction=1.0)]),
}


def _make(name, overrides, robot, objs):
  def __init__(self, G={}, **backend_kw):
    w = WorldDef(robots=[Robot(type=robot, name=f'{robot}0')] if robot else [],
                 objects=[Object(f'object{i}', **kw) for i, kw in enumerate(objs)])
    WorldEnv.__init__(self, w, G, **backend_kw)
  cls = type(name, (WorldEnv,), {'__init__': __init__, '__doc__': f'boxLCD env {name} (reference boxLCD/envs.py)'})
  return cc(**overrides)(cls) if overrides is not None else cls


for _name, (_ov, _robot, _objs) in _CATALOGUE.items():
  globals()[_name] = _make(_name, _ov, _robot, _objs)
del _name, _ov, _robot, _objs
