// blcd_collide_wall.h — narrow phase against the arena walls, written for what a boxLCD wall IS.
//
// Every static fixture of a boxLCD world is one of the four isolated edges made in boxLCD/world_env.py:309-312; the body that
// carries it sits at the origin with angle 0 and never moves.  b2CollideEdgeAndCircle / b2EPCollider::Collide (upstream
// b2CollideEdge.cpp, reached through b2Contact::Update for every (wall, body) pair) spend a large part of their instructions
// on things that depend on the wall alone - two b2Vec2::Normalize (a correctly rounded sqrt and divide each), 1 / |e|^2, the
// identity transform - and on keeping both orientations of generic data structures alive.  Here
//   * everything that is a function of the wall only is evaluated once per launch (WallK, same expressions, same order),
//   * the wall body's transform is folded away where that cannot change a result, and multiplied through where a sign of a
//     zero could (MulT(xfA, xfB) below),
//   * contact ids are built as the 32-bit keys they are stored as, clip vertices are scalars.
// The arithmetic that produces a manifold is upstream's, operation for operation; the CPU oracle keeps upstream's generic
// routines and the parity suite compares manifolds bit for bit after every world step.
#pragma once
#include "blcd_collide.h"

namespace blcd {

// What the two routines need to know about one wall (an edge from a to b with radius rad), as upstream computes it per call
struct WallK {
  Vec2 a, b;       // m_vertex1, m_vertex2
  Vec2 e;          // b - a
  float invDen;    // 1 / Dot(e, e)
  Vec2 nLeft;      // Normalize((-e.y, e.x)): the face normal of b2CollideEdgeAndCircle before its orientation test
  Vec2 t1;         // Normalize(e)                     (b2EPCollider: edge1)
  Vec2 n1;         // (t1.y, -t1.x)                    (b2EPCollider: m_normal1)
  float rad;
};
BLCD_HD static inline WallK MakeWallK(Vec2 v1, Vec2 v2, float radius) {
  WallK w;
  w.a = v1;
  w.b = v2;
  w.e = v2 - v1;
  const float den = Dot(w.e, w.e);
  w.invDen = 1.0f / den;
  w.nLeft = V2(-w.e.y, w.e.x);
  Normalize(w.nLeft);
  w.t1 = w.e;
  Normalize(w.t1);
  w.n1 = V2(w.t1.y, -w.t1.x);
  w.rad = radius;
  return w;
}

// (wall, circle): b2CollideEdgeAndCircle for an isolated edge.  Q = MulT(xfA, Mul(xfB, c)) with the identity xfA is Mul(xfB, c)
// up to the sign of a zero, and Q only enters differences, dot products and comparisons.  -Normalize(n) == Normalize(-n) bit for
// bit (squares and products are sign-symmetric), so the oriented face normal is +-nLeft.
BLCD_HD static inline void CollideWallCircle(Manifold* m, const WallK& w, Vec2 centre, float circleRadius, const Transform& xfB) {
  m->pointCount = 0;
  const Vec2 Q = Mul(xfB, centre);
  const float u = Dot(w.e, w.b - Q);
  const float v = Dot(w.e, Q - w.a);
  const float reach = w.rad + circleRadius;
  // Voronoi region of Q: vertex a (v <= 0), vertex b (u <= 0), or the face.  Closest point P, contact id and manifold kind per region
  const bool atA = v <= 0.0f, atB = !atA && u <= 0.0f;
  Vec2 P;
  if (atA) {
    P = w.a;
  } else if (atB) {
    P = w.b;
  } else {
    P = w.invDen * (u * w.a + v * w.b);
  }
  const Vec2 d = Q - P;
  if (Dot(d, d) > reach * reach) return;
  m->pointCount = 1;
  m->points[0].localPoint = centre;
  if (atA || atB) {
    m->type = kManifoldCircles;
    m->localNormal = V2(0.0f, 0.0f);
    m->localPoint = P;
    m->points[0].id.key = FeatureKey(atB ? 1 : 0, 0, kFeatureVertex, kFeatureVertex);
  } else {
    const bool flip = Dot(V2(-w.e.y, w.e.x), Q - w.a) < 0.0f;
    m->type = kManifoldFaceA;
    m->localNormal = flip ? -w.nLeft : w.nLeft;
    m->localPoint = w.a;
    m->points[0].id.key = FeatureKey(0, 0, kFeatureFace, kFeatureVertex);
  }
}

// (wall, polygon): b2EPCollider::Collide for an isolated edge (no adjacent vertices: lower/upper limit = -m_normal)
BLCD_HD static inline void CollideWallPolygon(Manifold* m, const WallK& w, const Shape* poly, const Transform& xfB) {
  // m_xf = b2MulT(xfA, xfB) with xfA = identity, multiplied through: q = (1 sB - 0 cB, 1 cB + 0 sB), p = (1 dx + 0 dy, -0 dx + 1 dy)
  Transform xf;
  xf.q.s = xfB.q.s - 0.0f * xfB.q.c;
  xf.q.c = xfB.q.c + 0.0f * xfB.q.s;
  xf.p.x = xfB.p.x + 0.0f * xfB.p.y;
  xf.p.y = -0.0f * xfB.p.x + xfB.p.y;
  const Vec2 centroidB = Mul(xf, poly->centroid);
  const bool front = Dot(w.n1, centroidB - w.a) >= 0.0f;
  const Vec2 faceN = front ? w.n1 : -w.n1;     // m_normal
  const Vec2 limit = -faceN;                   // m_lowerLimit == m_upperLimit for an isolated edge
  const int count = poly->count;
  Vec2 pv[kShapeVerts], pn[kShapeVerts];       // the polygon in the wall's frame
#pragma unroll
  for (int i = 0; i < kShapeVerts; ++i) {
    const bool live = i < count;
    pv[i] = live ? Mul(xf, poly->v[i]) : V2(0.0f, 0.0f);
    pn[i] = live ? Mul(xf.q, poly->n[i]) : V2(0.0f, 0.0f);
  }
  const float reach = 2.0f * kPolygonRadius;
  m->pointCount = 0;

  // separation along the wall normal (ComputeEdgeSeparation)
  float edgeSep = FLT_MAX;
#pragma unroll
  for (int i = 0; i < kShapeVerts; ++i)
    if (i < count) edgeSep = Min(Dot(faceN, pv[i] - w.a), edgeSep);   // s < best ? s : best
  if (edgeSep > reach) return;

  // separation along the polygon's face normals (ComputePolygonSeparation): first face that separates outright, else the
  // deepest admissible one
  int polyIdx = -1;
  float polySep = -FLT_MAX;
  {
    bool stop = false;
#pragma unroll
    for (int i = 0; i < kShapeVerts; ++i) {
      if (i >= count || stop) continue;
      const Vec2 n = -pn[i];
      const float s = Min(Dot(n, pv[i] - w.a), Dot(n, pv[i] - w.b));
      if (s > reach) {
        polyIdx = i;
        polySep = s;
        stop = true;
        continue;
      }
      // adjacency test: for an isolated edge both branches of upstream's `Dot(n, perp) >= 0` read the same limit
      if (Dot(n - limit, faceN) < -kAngularSlop) continue;
      if (s > polySep) {
        polyIdx = i;
        polySep = s;
      }
    }
  }
  if (polyIdx >= 0 && polySep > reach) return;

  const bool polyFace = polyIdx >= 0 && polySep > 0.98f * edgeSep + 0.001f;   // k_relativeTol, k_absoluteTol
  // incident edge (two clip vertices) and reference face
  Vec2 c0, c1, rv1, rv2, rn;
  uint32_t k0, k1;
  int ri1, ri2;
  if (!polyFace) {
    m->type = kManifoldFaceA;
    int best = 0;
    float bestDot = Dot(faceN, pn[0]);
#pragma unroll
    for (int i = 1; i < kShapeVerts; ++i) {
      if (i < count) {
        const float value = Dot(faceN, pn[i]);
        if (value < bestDot) {
          bestDot = value;
          best = i;
        }
      }
    }
    const int i1 = best, i2 = best + 1 < count ? best + 1 : 0;
    c0 = SelVec(pv, i1);
    c1 = SelVec(pv, i2);
    k0 = FeatureKey(0, i1, kFeatureFace, kFeatureVertex);
    k1 = FeatureKey(0, i2, kFeatureFace, kFeatureVertex);
    ri1 = front ? 0 : 1;
    ri2 = front ? 1 : 0;
    rv1 = front ? w.a : w.b;
    rv2 = front ? w.b : w.a;
    rn = faceN;
  } else {
    m->type = kManifoldFaceB;
    c0 = w.a;
    c1 = w.b;
    k0 = k1 = FeatureKey(0, polyIdx, kFeatureVertex, kFeatureFace);
    ri1 = polyIdx;
    ri2 = polyIdx + 1 < count ? polyIdx + 1 : 0;
    rv1 = SelVec(pv, ri1);
    rv2 = SelVec(pv, ri2);
    rn = SelVec(pn, ri1);
  }
  const Vec2 side1 = V2(rn.y, -rn.x), side2 = -side1;
  if (ClipPair(c0, k0, c1, k1, side1, Dot(side1, rv1), ri1) < kMaxManifoldPoints) return;
  if (ClipPair(c0, k0, c1, k1, side2, Dot(side2, rv2), ri2) < kMaxManifoldPoints) return;
  if (!polyFace) {
    m->localNormal = rn;
    m->localPoint = rv1;
  } else {
    m->localNormal = poly->n[ri1];
    m->localPoint = poly->v[ri1];
  }
  int kept = 0;
#pragma unroll
  for (int i = 0; i < kMaxManifoldPoints; ++i) {
    const Vec2 c = i == 0 ? c0 : c1;
    const uint32_t key = i == 0 ? k0 : k1;
    if (Dot(rn, c - rv1) <= reach) {
      ManifoldPoint cp;
      cp.normalImpulse = 0.0f;
      cp.tangentImpulse = 0.0f;
      cp.localPoint = polyFace ? c : MulT(xf, c);
      cp.id.key = polyFace ? SwapFeatureSides(key) : key;
      PutManifoldPoint(m, kept, cp);
      ++kept;
    }
  }
  m->pointCount = kept;
}

}  // namespace blcd
