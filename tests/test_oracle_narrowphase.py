"""The narrow phase against GEOMETRY, not against a copy of itself.

The device narrow phase (`blcd_collide.h`) is compared bit for bit with the oracle's (`oracle/b2o_collide.h`) by the parity
suite, but the two share most of their text (both restate Box2D 2.3.0's b2CollideCircle / b2CollidePolygon / b2CollideEdge):
that comparison shows two compilers agree.  This file checks the ALGORITHM: the manifolds the oracle's five routines produce for
tens of thousands of random near-contact configurations are compared with independent float64 geometry (SAT separations,
point-to-polygon / point-to-segment distances) - contact exactly when the shapes' skins overlap, reported separations equal to
the true ones, world points on the mid-surface between the two skins, normals pointing from A to B, results invariant under a
common rigid motion."""
import ctypes as C
import numpy as np
import pytest
from oracle import pyb2o

R_POLY = 0.01          # b2_polygonRadius = 2 * b2_linearSlop
TOL = 2e-5


def collide(specA, poseA, specB, poseB):
  lib = pyb2o.load()
  lib.b2o_collide.restype = C.c_int32
  out = np.zeros(24, np.float32)
  f = lambda a: np.ascontiguousarray(a, np.float32)
  a, pa, b, pb = f(specA), f(poseA), f(specB), f(poseB)
  n = lib.b2o_collide(pyb2o._p(a), pyb2o._p(pa), pyb2o._p(b), pyb2o._p(pb), pyb2o._p(out))
  assert n >= 0
  return dict(n=n, type=int(out[1]), ln=out[2:4].astype(float), lp=out[4:6].astype(float), wn=out[12:14].astype(float),
              wp=[out[14 + 3 * j:16 + 3 * j].astype(float) for j in range(n)], sep=[float(out[16 + 3 * j]) for j in range(n)], swapped=bool(out[20]))


def rot(a):
  a = float(np.float32(a))
  return np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])


def box_world(hx, hy, pose):
  v = np.array([[-hx, -hy], [hx, -hy], [hx, hy], [-hx, hy]], float)
  return v @ rot(pose[2]).T + np.array(pose[:2], float)


def normals(P):
  e = np.roll(P, -1, 0) - P
  n = np.stack([e[:, 1], -e[:, 0]], 1)
  return n / np.linalg.norm(n, axis=1, keepdims=True)


def face_seps(P, Q):
  """per face of convex CCW polygon P: min over vertices of Q of the signed distance to the face line"""
  n = normals(P)
  return ((Q[None, :, :] - P[:, None, :]) * n[:, None, :]).sum(-1).min(1)


def dist_point_seg(c, a, b):
  ab = b - a
  t = np.clip(np.dot(c - a, ab) / np.dot(ab, ab), 0.0, 1.0)
  return np.linalg.norm(c - (a + t * ab))


def dist_point_poly(c, P):
  s = ((c[None, :] - P) * normals(P)).sum(-1)
  if (s <= 0).all():
    return s.max()
  return min(dist_point_seg(c, P[i], P[(i + 1) % len(P)]) for i in range(len(P)))


def test_circle_circle():
  rng = np.random.RandomState(0)
  for _ in range(4000):
    ra, rb = rng.uniform(0.2, 0.8, 2)
    pa = np.array([rng.uniform(1, 4), rng.uniform(1, 4), rng.uniform(-3, 3)])
    d = ra + rb + rng.uniform(-0.05, 0.05)
    th = rng.uniform(0, 2 * np.pi)
    pb = np.array([pa[0] + d * np.cos(th), pa[1] + d * np.sin(th), rng.uniform(-3, 3)])
    m = collide([0, ra], pa, [0, rb], pb)
    true = np.linalg.norm(np.float32(pb[:2]).astype(float) - np.float32(pa[:2]).astype(float)) - np.float32(ra) - np.float32(rb)
    if abs(true) > 1e-5:
      assert (m['n'] == 1) == (true < 0), (true, m)
    if m['n']:
      assert abs(m['sep'][0] - true) < TOL
      u = (pb[:2] - pa[:2]) / np.linalg.norm(pb[:2] - pa[:2])
      assert np.abs(m['wn'] - u).max() < 1e-4
      mid = pa[:2] + u * (np.float32(ra) + true / 2)
      assert np.abs(m['wp'][0] - mid).max() < 1e-4


def test_polygon_circle_and_edge_circle():
  rng = np.random.RandomState(1)
  for _ in range(6000):
    r = rng.uniform(0.2, 0.6)
    edge = rng.rand() < 0.4
    if edge:
      a, b = np.array([0.0, 0.0]), np.array([rng.uniform(3, 8), 0.0]) if rng.rand() < 0.5 else np.array([0.0, rng.uniform(3, 8)])
      specA, poseA = [2, a[0], a[1], b[0], b[1]], [0.0, 0.0, 0.0]
      t = rng.uniform(-0.1, 1.1)
      base = a + t * (b - a)
      nrm = np.array([-(b - a)[1], (b - a)[0]]) / np.linalg.norm(b - a) * rng.choice([-1, 1])
      c = base + nrm * (r + R_POLY + rng.uniform(-0.05, 0.05))
      true = dist_point_seg(np.float32(c).astype(float), a, b) - np.float32(r) - R_POLY
    else:
      hx, hy = rng.uniform(0.2, 0.8, 2)
      poseA = [rng.uniform(1, 4), rng.uniform(1, 4), rng.uniform(-3, 3)]
      specA = [1, hx, hy]
      P = box_world(np.float32(hx), np.float32(hy), np.float32(poseA))
      th = rng.uniform(0, 2 * np.pi)
      far = np.array(poseA[:2]) + 3.0 * np.array([np.cos(th), np.sin(th)])
      # walk in from far away to the wanted clearance
      lo_, hi_ = 0.0, 1.0
      want = np.float32(r) + R_POLY + rng.uniform(-0.05, 0.05)
      for _k in range(40):
        mid = 0.5 * (lo_ + hi_)
        c = far + mid * (np.array(poseA[:2]) - far)
        if dist_point_poly(c, P) > want: lo_ = mid
        else: hi_ = mid
      c = far + lo_ * (np.array(poseA[:2]) - far)
      true = dist_point_poly(np.float32(c).astype(float), P) - np.float32(r) - R_POLY
    m = collide(specA, poseA, [0, r], [c[0], c[1], rng.uniform(-3, 3)])
    if abs(true) > 5e-5:
      assert (m['n'] == 1) == (true < 0), (edge, true, m)
    if m['n']:
      assert abs(m['sep'][0] - true) < 5e-5, (edge, true, m)
      assert abs(np.linalg.norm(m['wn']) - 1) < 1e-5
      assert np.dot(m['wn'], np.float32(c).astype(float) - m['wp'][0]) > 0          # the normal points from A to the circle


def _sat(P, Q):
  return max(face_seps(P, Q).max(), face_seps(Q, P).max())


def test_polygon_polygon_and_edge_polygon():
  rng = np.random.RandomState(2)
  bad_clip = 0
  total = 0
  for _ in range(8000):
    edge = rng.rand() < 0.4
    hx, hy = rng.uniform(0.2, 0.8, 2)
    angB = rng.choice([rng.uniform(-3, 3), rng.choice([0, np.pi / 2, np.pi]) + rng.normal(0, 0.02)])    # generic and nearly flat
    if edge:
      L = rng.uniform(4, 8)
      horizontal = rng.rand() < 0.5
      a, b = np.array([0.0, 0.0]), (np.array([L, 0.0]) if horizontal else np.array([0.0, L]))
      specA, poseA = [2, a[0], a[1], b[0], b[1]], [0.0, 0.0, 0.0]
      nrm = np.array([-(b - a)[1], (b - a)[0]]) / L * rng.choice([-1, 1])
      base = a + rng.uniform(0.15, 0.85) * (b - a)
      # place the box so that its lowest vertex along -nrm is at the wanted clearance
      P0 = box_world(np.float32(hx), np.float32(hy), np.float32([0, 0, angB]))
      depth = (P0 @ nrm).min()
      clear = 2 * R_POLY + rng.uniform(-0.03, 0.03)
      cB = base + nrm * (clear - depth)
      poseB = [cB[0], cB[1], angB]
      Q = box_world(np.float32(hx), np.float32(hy), np.float32(poseB))
      side = np.sign(np.dot(Q.mean(0) - a, nrm)) * nrm                 # the edge's normal on the polygon's side (b2EPCollider m_front)
      s_edge = ((Q - a) @ side).min()
      nQ = normals(Q)
      s_poly = np.minimum(((Q - a) * -nQ).sum(1), ((Q - b) * -nQ).sum(1))   # -n_i separation of the two edge points, as the collider tests it
      s_true, Rt = s_edge, 2 * R_POLY
      separated = s_edge > Rt + 5e-5 or (s_poly > Rt + 5e-5).any()
      touching = s_edge < Rt - 5e-5 and (s_poly < Rt - 5e-5).all()
      specB = [1, hx, hy]
    else:
      hx2, hy2 = rng.uniform(0.2, 0.8, 2)
      poseA = [rng.uniform(2, 4), rng.uniform(2, 4), rng.choice([rng.uniform(-3, 3), 0.0])]
      specA = [1, hx2, hy2]
      P = box_world(np.float32(hx2), np.float32(hy2), np.float32(poseA))
      th = rng.uniform(0, 2 * np.pi)
      far = np.array(poseA[:2]) + 4.0 * np.array([np.cos(th), np.sin(th)])
      want = 2 * R_POLY + rng.uniform(-0.03, 0.03)
      lo_, hi_ = 0.0, 1.0
      for _k in range(40):
        mid = 0.5 * (lo_ + hi_)
        cB = far + mid * (np.array(poseA[:2]) - far)
        if _sat(P, box_world(np.float32(hx), np.float32(hy), np.float32([cB[0], cB[1], angB]))) > want: lo_ = mid
        else: hi_ = mid
      cB = far + lo_ * (np.array(poseA[:2]) - far)
      poseB = [cB[0], cB[1], angB]
      Q = box_world(np.float32(hx), np.float32(hy), np.float32(poseB))
      s_true, Rt = _sat(P, Q), 2 * R_POLY
      separated, touching = s_true > Rt + 5e-5, s_true < Rt - 5e-5
      specB = [1, hx, hy]
    m = collide(specA, poseA, specB, poseB)
    total += 1
    if separated:
      assert m['n'] == 0, (edge, s_true, m)            # never a contact between shapes whose skins are apart
    if touching and m['n'] == 0:
      bad_clip += 1                                    # SAT overlap but the clipped incident edge missed the reference face's side planes
    if m['n']:
      assert not separated
      assert abs(np.linalg.norm(m['wn']) - 1) < 1e-5
      # the deepest reported point is as deep as the true separation, up to the routine's own face-preference slack
      # (it keeps face A unless B's is better by 2 % + 0.001)
      smin = min(m['sep'])
      assert smin <= s_true - Rt + 0.02 * abs(s_true) + 0.001 + 5e-5 and smin >= s_true - Rt - 0.03 - 5e-5, (edge, s_true - Rt, m)
      cA = np.array([0.0, 0.0]) + (np.array([L, 0.0]) if edge and horizontal else np.array([0.0, L]) if edge else 0) * 0.5 if edge else P.mean(0)
      if abs(s_true) < 0.02:                           # shallow: the normal points from A towards B's centroid
        assert np.dot(m['wn'], Q.mean(0) - m['wp'][0]) > 0
      # the world points sit on the mid-surface between the two skins: measured from the REFERENCE face (a face of A for a
      # face-A manifold, of B for face-B; found here as the face whose outward normal is the manifold normal) a point with
      # separation s lies at radius + s / 2
      ref = Q if m['type'] == 2 else (np.stack([a, b]) if edge else P)
      nref = m['wn'] if m['type'] != 2 else -m['wn']
      if edge and m['type'] != 2:
        dplane = lambda p: np.dot(p - a, nref)
      else:
        nn = normals(ref)
        k = int(np.argmax(nn @ nref))
        assert nn[k] @ nref > 1 - 1e-5
        dplane = lambda p: np.dot(p - ref[k], nn[k])
      for p, s in zip(m['wp'], m['sep']):
        assert abs(dplane(p) - (R_POLY + s / 2)) < 2e-4, (edge, dplane(p), s, m)
  assert bad_clip <= 0.002 * total, bad_clip


def test_manifolds_are_invariant_under_a_common_rigid_motion():
  rng = np.random.RandomState(3)
  for _ in range(1500):
    specA = [1, *rng.uniform(0.2, 0.8, 2)]
    specB = [1, *rng.uniform(0.2, 0.8, 2)] if rng.rand() < 0.5 else [0, rng.uniform(0.2, 0.6)]
    poseA = np.array([0.0, 0.0, rng.uniform(-1, 1)])
    poseB = np.array([rng.uniform(0.3, 1.2), rng.uniform(-0.6, 0.6), rng.uniform(-3, 3)])
    m0 = collide(specA, poseA, specB, poseB)
    th, t = rng.uniform(-3, 3), rng.uniform(1, 4, 2)
    Rm = rot(th)
    mv = lambda p: np.array([*(Rm @ p[:2] + t), p[2] + th])
    m1 = collide(specA, mv(poseA), specB, mv(poseB))
    if m0['n'] == m1['n'] and m0['n'] > 0 and m0['type'] == m1['type'] and min(m0['sep']) < -1e-3:
      assert np.abs(m0['ln'] - m1['ln']).max() < 1e-4 or m0['n'] == 2       # the local manifold does not see the motion
      assert abs(min(m0['sep']) - min(m1['sep'])) < 1e-4
