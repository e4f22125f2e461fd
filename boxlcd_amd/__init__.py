"""boxlcd_amd — MI355X-native batched Box2D-2.3-equivalent stepper + 1-bit LCD rasteriser behind boxLCD's env API.

Drop-in for the hot path of matwilso/boxLCD (`WorldEnv.step()/render()`); see DESIGN.md for scope.
Mirrors the reference package surface (boxLCD/__init__.py:9-17): `envs`, `env_map`, `ENV_DG`, `WorldEnv`,
`WorldDef`, `Object`, `Robot`, `AttrDict`, plus the batched `BatchedWorldEnv`.
"""
import inspect as _inspect
from .__version__ import __version__
from .world_env import WorldEnv, BatchedWorldEnv
from .world_defs import WorldDef, Object, Robot
from . import envs
from .utils import AttrDict
from . import goal  # BodyGoalEnv / CubeGoalEnv (research/wrappers) with the reward epilogue on device

ENV_DG = AttrDict(WorldEnv.ENV_DG)
env_map = {name: obj for name, obj in _inspect.getmembers(envs)
           if _inspect.isclass(obj) and issubclass(obj, WorldEnv) and obj is not WorldEnv}
