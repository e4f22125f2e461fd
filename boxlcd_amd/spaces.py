"""Minimal stand-ins for `gym.spaces.Box` / `gym.spaces.Dict` (gym 0.17.3 is the reference's dependency,
requirements.txt:2; it is not installed here).  Only what boxLCD's callers touch is provided: `.shape`, `.dtype`,
`.low/.high`, `.sample()`, `.contains()`, `Dict.spaces`, and equality (reference uses: world_env.py:128-141,
research/wrappers/async_vector_env.py:297-307, research/data.py:57).  If gym is importable it is used instead.
"""
import numpy as np

try:  # pragma: no cover - gym is absent in the build container
  from gym.spaces import Box, Dict  # type: ignore
except Exception:

  class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
      self.dtype = np.dtype(dtype)
      self.shape = tuple(shape) if shape is not None else np.shape(low)
      self.low = np.full(self.shape, low, dtype=self.dtype if self.dtype != np.bool_ else np.float32)
      self.high = np.full(self.shape, high, dtype=self.dtype if self.dtype != np.bool_ else np.float32)
      self.np_random = np.random.RandomState()

    def seed(self, seed=None):
      self.np_random = np.random.RandomState(seed)
      return [seed]

    def sample(self):
      if self.dtype == np.bool_:
        return self.np_random.randint(0, 2, self.shape).astype(np.bool_)
      return self.np_random.uniform(self.low, self.high, self.shape).astype(self.dtype)

    def contains(self, x):
      x = np.asarray(x)
      return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def __eq__(self, other):
      return isinstance(other, Box) and self.shape == other.shape and self.dtype == other.dtype and \
          np.allclose(self.low, other.low) and np.allclose(self.high, other.high)

    def __repr__(self):
      return f'Box{self.shape}'

  class Dict:
    def __init__(self, spaces):
      self.spaces = dict(spaces)

    def seed(self, seed=None):
      return [s.seed(seed) for s in self.spaces.values()]

    def sample(self):
      return {k: s.sample() for k, s in self.spaces.items()}

    def contains(self, x):
      return isinstance(x, dict) and all(k in x and s.contains(x[k]) for k, s in self.spaces.items())

    def __getitem__(self, key):
      return self.spaces[key]

    def __eq__(self, other):
      return isinstance(other, Dict) and self.spaces == other.spaces

    def __repr__(self):
      return 'Dict(' + ', '.join(f'{k}:{s}' for k, s in self.spaces.items()) + ')'
