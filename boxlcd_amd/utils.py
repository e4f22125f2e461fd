"""Host-side helpers the hot path uses, with the semantics of the reference's `boxLCD/utils.py` (AttrDict :5-7,
A[...] :18-31, NamedArray :33-101, sortdict/nfiltlist :104-111, mapto :117, rmapto :119).

The arithmetic of `mapto` / `rmapto` is part of the hot path's contract (observation normalisation and the
action -> motorSpeed mapping): float64, in exactly this operation order.
"""
import re
import numpy as np


class AttrDict(dict):
  """dict whose items are also attributes (reference utils.py:5-7)."""
  __setattr__ = dict.__setitem__
  __getattr__ = dict.__getitem__


class _ArrayMaker:
  """`A[1, 2, 3]` -> np.array([1, 2, 3]) (reference utils.py:18-31)."""

  def __getitem__(self, stuff):
    return np.array(stuff)


A = _ArrayMaker()


def mapto(a, lowhigh):
  """[-1, 1] -> [low, high]   (reference utils.py:117; keep the operation order)."""
  return ((a + 1.0) / (2.0) * (lowhigh[1] - lowhigh[0])) + lowhigh[0]


def rmapto(a, lowhigh):
  """[low, high] -> [-1, 1]   (reference utils.py:119; keep the operation order)."""
  return ((a - lowhigh[0]) / (lowhigh[1] - lowhigh[0]) * (2)) + -1


class NamedArray:
  """Name-indexed, optionally range-normalised view of the last axis of `arr` (reference utils.py:33-101)."""

  def __init__(self, arr, arr_info, do_map=True):
    self.arr = arr
    self.arr_info = arr_info
    self.do_map = do_map
    self._index = {k: i for i, k in enumerate(arr_info)}

  def _name2idx(self, name):
    return self._index[name]

  def _bounds(self, key):
    return np.array([self.arr_info[k] for k in key]).T

  def todict(self):
    return {key: self[key] for key in self.arr_info}

  def __call__(self, key):
    return self[key]

  def __getitem__(self, key):
    if isinstance(key, str):
      idx = self._name2idx(key)
      return mapto(self.arr[..., idx], self.arr_info[key]) if self.do_map else self.arr[..., idx]
    if isinstance(key, (list, tuple)):
      idx = [self._name2idx(k) for k in key]
      return mapto(self.arr[..., idx], self._bounds(key)) if self.do_map else self.arr[..., idx]
    raise NotImplementedError

  def __setitem__(self, key, item):
    if isinstance(key, str):
      idx = self._name2idx(key)
      self.arr[..., idx] = rmapto(item, self.arr_info[key]) if self.do_map else item
    elif isinstance(key, (list, tuple)):
      idx = [self._name2idx(k) for k in key]
      self.arr[..., idx] = rmapto(item, self._bounds(key)) if self.do_map else item
    else:
      raise NotImplementedError


def sortdict(x): return {key: x[key] for key in sorted(x)}
def nfiltlist(l, phrase): return [i for i in l if re.match(phrase, i) is None]
