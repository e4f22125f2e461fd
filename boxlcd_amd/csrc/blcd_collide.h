// blcd_collide.h — shapes + narrow phase of the HIP product (host+device): b2PolygonShape::Set/SetAsBox/ComputeMass,
// b2CollideCircles, b2CollidePolygonAndCircle, b2CollidePolygons; the two edge routines are blcd_collide_wall.h
// (Box2D 2.3.x semantics).  Shapes come from boxLCD/world_env.py:273,311-314 and
// boxLCD/world_defs.py:82-83,103-108.
#pragma once
#include <utility>
#include "blcd_math.h"

namespace blcd {

enum ShapeType { kCircle = 0, kPolygon = 1, kEdge = 2 };

struct Shape {
  int type;
  float radius;   // circle radius; polygon/edge: b2_polygonRadius
  int count;      // polygon vertex count; edge: 2
  Vec2 v[kShapeVerts];  // polygon vertices (CCW hull) / edge v1,v2 / circle: v[0] = m_p
  Vec2 n[kShapeVerts];
  Vec2 centroid;
};

struct MassData {
  float mass;
  Vec2 center;
  float I;
};

// b2PolygonShape::SetAsBox (b2PolygonShape.cpp)
BLCD_HD static inline void ShapeSetAsBox(Shape* s, float hx, float hy) {
  s->type = kPolygon;
  s->radius = kPolygonRadius;
  s->count = 4;
  s->v[0] = V2(-hx, -hy);
  s->v[1] = V2(hx, -hy);
  s->v[2] = V2(hx, hy);
  s->v[3] = V2(-hx, hy);
  s->n[0] = V2(0.0f, -1.0f);
  s->n[1] = V2(1.0f, 0.0f);
  s->n[2] = V2(0.0f, 1.0f);
  s->n[3] = V2(-1.0f, 0.0f);
  s->centroid = V2(0.0f, 0.0f);
}

// ComputeCentroid (b2PolygonShape.cpp)
BLCD_HD static inline Vec2 ComputeCentroid(const Vec2* vs, int count) {
  // area-weighted mean of the centroids of the fan of triangles (origin, vs[i], vs[i+1]); the origin's zero terms stay in the sums
  const Vec2 origin = V2(0.0f, 0.0f);
  const float third = 1.0f / 3.0f;
  Vec2 sum = V2(0.0f, 0.0f);
  float area = 0.0f;
  for (int i = 0; i < count; ++i) {
    const Vec2 a = vs[i], b = vs[i + 1 < count ? i + 1 : 0];
    const float tri = 0.5f * Cross(a - origin, b - origin);
    area += tri;
    sum += tri * third * (origin + a + b);
  }
  sum *= 1.0f / area;
  return sum;
}

// b2PolygonShape::Set (b2PolygonShape.cpp, 2.3.1: weld, gift-wrap hull from the right-most point, normals, centroid)
BLCD_HD static inline void ShapeSetPolygon(Shape* s, const Vec2* vertices, int count) {
  s->type = kPolygon;
  s->radius = kPolygonRadius;
  if (count < 3) {
    ShapeSetAsBox(s, 1.0f, 1.0f);
    return;
  }
  int n = count < kShapeVerts ? count : kShapeVerts;
  Vec2 ps[kShapeVerts];
  int tempCount = 0;
  for (int i = 0; i < n; ++i) {
    Vec2 v = vertices[i];
    bool unique = true;
    for (int j = 0; j < tempCount; ++j) {
      if (DistanceSquared(v, ps[j]) < 0.5f * kLinearSlop) {
        unique = false;
        break;
      }
    }
    if (unique) ps[tempCount++] = v;
  }
  n = tempCount;
  if (n < 3) {
    ShapeSetAsBox(s, 1.0f, 1.0f);
    return;
  }
  int i0 = 0;
  float x0 = ps[0].x;
  for (int i = 1; i < n; ++i) {
    float x = ps[i].x;
    if (x > x0 || (x == x0 && ps[i].y < ps[i0].y)) {
      i0 = i;
      x0 = x;
    }
  }
  int hull[kShapeVerts];
  int m = 0;
  int ih = i0;
  for (;;) {
    hull[m] = ih;
    int ie = 0;
    for (int j = 1; j < n; ++j) {
      if (ie == ih) {
        ie = j;
        continue;
      }
      Vec2 r = ps[ie] - ps[hull[m]];
      Vec2 v = ps[j] - ps[hull[m]];
      float c = Cross(r, v);
      if (c < 0.0f) ie = j;
      if (c == 0.0f && LengthSquared(v) > LengthSquared(r)) ie = j;
    }
    ++m;
    ih = ie;
    if (ie == i0) break;
  }
  if (m < 3) {
    ShapeSetAsBox(s, 1.0f, 1.0f);
    return;
  }
  s->count = m;
  for (int i = 0; i < m; ++i) s->v[i] = ps[hull[i]];
  for (int i = 0; i < m; ++i) {
    int i1 = i;
    int i2 = i + 1 < m ? i + 1 : 0;
    Vec2 edge = s->v[i2] - s->v[i1];
    s->n[i] = Cross(edge, 1.0f);
    Normalize(s->n[i]);
  }
  s->centroid = ComputeCentroid(s->v, m);
}

BLCD_HD static inline void ShapeSetCircle(Shape* s, float radius) {
  s->type = kCircle;
  s->radius = radius;
  s->count = 1;
  s->v[0] = V2(0.0f, 0.0f);
  s->centroid = V2(0.0f, 0.0f);
}

BLCD_HD static inline void ShapeSetEdge(Shape* s, Vec2 v1, Vec2 v2) {
  s->type = kEdge;
  s->radius = kPolygonRadius;
  s->count = 2;
  s->v[0] = v1;
  s->v[1] = v2;
  s->centroid = V2(0.0f, 0.0f);
}

// b2{Circle,Polygon,Edge}Shape::ComputeMass
BLCD_HD static inline void ShapeComputeMass(const Shape* s, MassData* md, float density) {
  if (s->type == kCircle) {
    md->mass = density * kPi * s->radius * s->radius;
    md->center = s->v[0];
    md->I = md->mass * (0.5f * s->radius * s->radius + Dot(s->v[0], s->v[0]));
    return;
  }
  if (s->type == kEdge) {
    md->mass = 0.0f;
    md->center = 0.5f * (s->v[0] + s->v[1]);
    md->I = 0.0f;
    return;
  }
  Vec2 center = V2(0.0f, 0.0f);
  float area = 0.0f;
  float I = 0.0f;
  Vec2 sref = V2(0.0f, 0.0f);
  for (int i = 0; i < s->count; ++i) sref += s->v[i];
  sref *= 1.0f / s->count;
  const float k_inv3 = 1.0f / 3.0f;
  for (int i = 0; i < s->count; ++i) {
    Vec2 e1 = s->v[i] - sref;
    Vec2 e2 = i + 1 < s->count ? s->v[i + 1] - sref : s->v[0] - sref;
    float D = Cross(e1, e2);
    float triangleArea = 0.5f * D;
    area += triangleArea;
    center += triangleArea * k_inv3 * (e1 + e2);
    float ex1 = e1.x, ey1 = e1.y;
    float ex2 = e2.x, ey2 = e2.y;
    float intx2 = ex1 * ex1 + ex2 * ex1 + ex2 * ex2;
    float inty2 = ey1 * ey1 + ey2 * ey1 + ey2 * ey2;
    I += (0.25f * k_inv3 * D) * (intx2 + inty2);
  }
  md->mass = density * area;
  center *= 1.0f / area;
  md->center = center + sref;
  md->I = density * I;
  md->I += md->mass * (Dot(md->center, md->center) - Dot(center, center));
}

// b2{Circle,Polygon,Edge}Shape::ComputeAABB
BLCD_HD static inline void CircleComputeAABB(const Shape* s, AABB* aabb, const Transform& xf) {
  Vec2 p = xf.p + Mul(xf.q, s->v[0]);
  aabb->lo = V2(p.x - s->radius, p.y - s->radius);
  aabb->hi = V2(p.x + s->radius, p.y + s->radius);
}
// An edge is treated as the two-vertex polygon it is: Min / Max over (v1, v2) is the loop's first step.
BLCD_HD static inline void ShapeComputeAABB(const Shape* s, AABB* aabb, const Transform& xf) {
  if (s->type == kCircle) {
    CircleComputeAABB(s, aabb, xf);
    return;
  }
  const int nv = s->type == kEdge ? 2 : s->count;
  Vec2 lo = Mul(xf, s->v[0]), hi = lo;
#pragma unroll
  for (int i = 1; i < kShapeVerts; ++i) {
    if (i < nv) {
      const Vec2 w = Mul(xf, s->v[i]);
      lo = Min(lo, w);
      hi = Max(hi, w);
    }
  }
  aabb->lo = lo - V2(s->radius, s->radius);
  aabb->hi = hi + V2(s->radius, s->radius);
}

// ----------------------------------------------------------------------------------------------
// b2Collision.h
// ----------------------------------------------------------------------------------------------
enum { kFeatureVertex = 0, kFeatureFace = 1 };
struct ContactFeature {
  uint8_t indexA, indexB, typeA, typeB;
};
union ContactID {
  ContactFeature cf;
  uint32_t key;
};
struct ManifoldPoint {
  Vec2 localPoint;
  float normalImpulse, tangentImpulse;
  ContactID id;
};
enum ManifoldType { kManifoldCircles = 0, kManifoldFaceA = 1, kManifoldFaceB = 2 };
struct Manifold {
  ManifoldPoint points[kMaxManifoldPoints];
  Vec2 localNormal, localPoint;
  int type;
  int pointCount;
};

// b2ContactFeature as stored: indexA | indexB << 8 | typeA << 16 | typeB << 24
BLCD_HD static inline uint32_t FeatureKey(int indexA, int indexB, int typeA, int typeB) {
  return (uint32_t)(indexA & 0xff) | ((uint32_t)(indexB & 0xff) << 8) | ((uint32_t)typeA << 16) | ((uint32_t)typeB << 24);
}
BLCD_HD static inline uint32_t SwapFeatureSides(uint32_t k) {   // (indexA, typeA) <-> (indexB, typeB)
  return ((k & 0x00ff00ffu) << 8) | ((k >> 8) & 0x00ff00ffu);
}

// cond ? a : b on the four words of a transform.  (A conditional expression on the objects themselves is a select of
// ADDRESSES, which keeps both operands - and whatever struct they are members of - in memory.)
BLCD_HD static inline Transform PickXf(bool first, const Transform& a, const Transform& b) {
  Transform t;
  t.p.x = first ? a.p.x : b.p.x;
  t.p.y = first ? a.p.y : b.p.y;
  t.q.s = first ? a.q.s : b.q.s;
  t.q.c = first ? a.q.c : b.q.c;
  return t;
}

// World-space contact normal and mid-surface points of a manifold (b2WorldManifold::Initialize, b2Collision.cpp; the solver
// never reads the separations it also computes).  The two face cases are one computation with the roles of the bodies
// exchanged: R carries the reference face, I the incident points; 0.5 (cA + cB) does not care which of the two is which.
struct WorldManifold {
  Vec2 normal;
  Vec2 points[kMaxManifoldPoints];
  BLCD_HD void Initialize(const Manifold* manifold, const Transform& xfA, float radiusA, const Transform& xfB, float radiusB) {
    if (manifold->pointCount == 0) return;
    if (manifold->type == kManifoldCircles) {
      const Vec2 onA = Mul(xfA, manifold->localPoint), onB = Mul(xfB, manifold->points[0].localPoint);
      normal = V2(1.0f, 0.0f);
      if (DistanceSquared(onA, onB) > kEpsilon * kEpsilon) {
        normal = onB - onA;
        Normalize(normal);
      }
      points[0] = 0.5f * ((onA + radiusA * normal) + (onB - radiusB * normal));
      return;
    }
    const bool refIsA = manifold->type == kManifoldFaceA;
    const Transform xfR = PickXf(refIsA, xfA, xfB), xfI = PickXf(refIsA, xfB, xfA);
    const float radiusR = refIsA ? radiusA : radiusB, radiusI = refIsA ? radiusB : radiusA;
    const Vec2 n = Mul(xfR.q, manifold->localNormal);
    const Vec2 plane = Mul(xfR, manifold->localPoint);
#pragma unroll
    for (int i = 0; i < kMaxManifoldPoints; ++i) {
      if (i >= manifold->pointCount) break;
      const Vec2 clip = Mul(xfI, manifold->points[i].localPoint);
      const Vec2 onRef = clip + (radiusR - Dot(clip - plane, n)) * n;
      const Vec2 onInc = clip - radiusI * n;
      points[i] = 0.5f * (onRef + onInc);
    }
    normal = refIsA ? n : -n;
  }
};

// manifold->points[idx] = {localPoint, id} without a run-time indexed store (impulses are left as they are, like Box2D)
BLCD_HD static inline void PutManifoldPoint(Manifold* m, int idx, const ManifoldPoint& cp) {
  if (idx == 0) {
    m->points[0].localPoint = cp.localPoint;
    m->points[0].id = cp.id;
  } else {
    m->points[1].localPoint = cp.localPoint;
    m->points[1].id = cp.id;
  }
}

// one b2ClipSegmentToLine (b2Collision.cpp) on scalar clip vertices (p0, k0), (p1, k1), in place; returns the number of output
// points.  When the segment crosses the line exactly one end is inside and the interpolated vertex becomes output 1; when both
// are inside the output is the input - the selection upstream's `vOut[numOut++] = ...` sequence makes, without indexed stores.
BLCD_HD static inline int ClipPair(Vec2& p0, uint32_t& k0, Vec2& p1, uint32_t& k1, Vec2 normal, float offset, int vertexIndexA) {
  const float s0 = Dot(normal, p0) - offset;
  const float s1 = Dot(normal, p1) - offset;
  int n = (s0 <= 0.0f ? 1 : 0) + (s1 <= 0.0f ? 1 : 0);
  Vec2 q1 = p1;
  uint32_t j1 = k1;
  if (s0 * s1 < 0.0f) {
    const float interp = s0 / (s0 - s1);
    q1 = p0 + interp * (p1 - p0);
    j1 = FeatureKey(vertexIndexA, (int)((k0 >> 8) & 0xffu), kFeatureVertex, kFeatureFace);
    ++n;
  }
  if (!(s0 <= 0.0f)) {
    p0 = p1;
    k0 = k1;
  }
  p1 = q1;
  k1 = j1;
  return n;
}

// (circle, circle): b2CollideCircles (b2CollideCircle.cpp)
BLCD_HD static inline void CollideCircles(Manifold* m, const Shape* circleA, const Transform& xfA, const Shape* circleB, const Transform& xfB) {
  m->pointCount = 0;
  const Vec2 gap = Mul(xfB, circleB->v[0]) - Mul(xfA, circleA->v[0]);
  const float reach = circleA->radius + circleB->radius;
  if (Dot(gap, gap) > reach * reach) return;
  m->type = kManifoldCircles;
  m->localNormal = V2(0.0f, 0.0f);
  m->localPoint = circleA->v[0];
  m->points[0].localPoint = circleB->v[0];
  m->points[0].id.key = 0;
  m->pointCount = 1;
}

// (polygon, circle): b2CollidePolygonAndCircle (b2CollideCircle.cpp).  The face of deepest penetration is found first (a face
// that separates outright ends the routine); then the circle centre is in one of four situations - inside the polygon, beyond
// the face's first or second vertex, or over the face - each of which yields (normal, point) for the same one-point manifold.
BLCD_HD static inline void CollidePolygonAndCircle(Manifold* m, const Shape* polygonA, const Transform& xfA, const Shape* circleB,
                                                   const Transform& xfB) {
  m->pointCount = 0;
  const Vec2 centre = MulT(xfA, Mul(xfB, circleB->v[0]));   // the circle in the polygon's frame
  const float reach = polygonA->radius + circleB->radius;
  const int count = polygonA->count;
  int face = 0;
  float deepest = -kMaxFloat;
  for (int i = 0; i < count; ++i) {
    const float s = Dot(polygonA->n[i], centre - polygonA->v[i]);
    if (s > reach) return;
    if (s > deepest) {
      deepest = s;
      face = i;
    }
  }
  const Vec2 v1 = polygonA->v[face], v2 = polygonA->v[face + 1 < count ? face + 1 : 0];
  const Vec2 faceN = polygonA->n[face];
  Vec2 normal, point;
  if (deepest < kEpsilon) {
    normal = faceN;
    point = 0.5f * (v1 + v2);
  } else {
    const float u1 = Dot(centre - v1, v2 - v1);
    const float u2 = Dot(centre - v2, v1 - v2);
    if (u1 <= 0.0f || u2 <= 0.0f) {
      const Vec2 corner = u1 <= 0.0f ? v1 : v2;
      if (DistanceSquared(centre, corner) > reach * reach) return;
      normal = centre - corner;
      Normalize(normal);
      point = corner;
    } else {
      const Vec2 mid = 0.5f * (v1 + v2);
      if (Dot(centre - mid, faceN) > reach) return;
      normal = faceN;
      point = mid;
    }
  }
  m->pointCount = 1;
  m->type = kManifoldFaceA;
  m->localNormal = normal;
  m->localPoint = point;
  m->points[0].localPoint = circleB->v[0];
  m->points[0].id.key = 0;
}

// b2FindMaxSeparation + b2EdgeSeparation (b2CollidePolygon.cpp, Box2D 2.3.0): separation of poly2 from edge e of poly1 =
// Dot(xf2*v2[support] - xf1*v1[e], n1World[e]) with the support vertex chosen in poly2's frame; the edge is found by a hill
// climb that starts at the edge whose normal faces poly2's centroid.  The separation of an edge is a pure function of the
// pose, so this kernel-side formulation evaluates all (<= 8) edges branch-free first - poly2's world vertices once - and
// then replays the climb's decisions on that table; the oracle evaluates lazily like upstream.
// Run-time indexed reads of the two small tables below are compile-time unrolled compare/selects (fold expressions): written as
// loops they are re-rolled by LLVM into indexed loads, the tables then live in scratch, and every step of the climb becomes a
// dependent ~450-cycle scratch load at one wave per SIMD.
template <typename T, size_t... K>
BLCD_HD static inline T pickOf8(const T (&a)[kShapeVerts], int i, std::index_sequence<K...>) {
  T r = a[0];
  ((r = (i == (int)(K + 1)) ? a[K + 1] : r), ...);
  return r;
}
BLCD_HD static inline float FindMaxSeparation(int* edgeIndex, const Shape* poly1, const Transform& xf1, const Shape* poly2,
                                      const Transform& xf2) {
  static_assert(kShapeVerts == 8, "pickOf8");
  const int count1 = poly1->count, count2 = poly2->count;
  float w2x[kShapeVerts], w2y[kShapeVerts], v2x[kShapeVerts], v2y[kShapeVerts];
#pragma unroll
  for (int j = 0; j < kShapeVerts; ++j) {
    const Vec2 pv = poly2->v[j < count2 ? j : 0];
    const Vec2 pw = Mul(xf2, pv);
    v2x[j] = pv.x;
    v2y[j] = pv.y;
    w2x[j] = pw.x;
    w2y[j] = pw.y;
  }
  float sep[kShapeVerts];
#pragma unroll
  for (int e = 0; e < kShapeVerts; ++e) {
    sep[e] = 0.0f;
    if (e >= count1) continue;
    const Vec2 nW = Mul(xf1.q, poly1->n[e]);
    const Vec2 nL = MulT(xf2.q, nW);
    int support = 0;
    float lowest = kMaxFloat;
#pragma unroll
    for (int j = 0; j < kShapeVerts; ++j) {
      if (j >= count2) continue;
      const float d = Dot(V2(v2x[j], v2y[j]), nL);
      if (d < lowest) {
        lowest = d;
        support = j;
      }
    }
    const Vec2 far = V2(pickOf8(w2x, support, std::make_index_sequence<kShapeVerts - 1>{}), pickOf8(w2y, support, std::make_index_sequence<kShapeVerts - 1>{}));
    sep[e] = Dot(far - Mul(xf1, poly1->v[e]), nW);
  }
  auto at = [&](int e) { return pickOf8(sep, e, std::make_index_sequence<kShapeVerts - 1>{}); };
  // starting edge: normal with the largest projection on the centroid offset (poly1 frame)
  const Vec2 dLocal1 = MulT(xf1.q, Mul(xf2, poly2->centroid) - Mul(xf1, poly1->centroid));
  int edge = 0;
  float maxDot = -kMaxFloat;
#pragma unroll
  for (int i = 0; i < kShapeVerts; ++i) {
    if (i >= count1) continue;
    const float dot = Dot(poly1->n[i], dLocal1);
    if (dot > maxDot) {
      maxDot = dot;
      edge = i;
    }
  }
  const int prevEdge = edge - 1 >= 0 ? edge - 1 : count1 - 1;
  const int nextEdge = edge + 1 < count1 ? edge + 1 : 0;
  const float s = at(edge), sPrev = at(prevEdge), sNext = at(nextEdge);
  int best, step;
  float bestSep;
  if (sPrev > s && sPrev > sNext) {
    step = -1;
    best = prevEdge;
    bestSep = sPrev;
  } else if (sNext > s) {
    step = 1;
    best = nextEdge;
    bestSep = sNext;
  } else {
    *edgeIndex = edge;
    return s;
  }
  for (int it = 0; it < kShapeVerts; ++it) {   // the climb visits each edge at most once
    const int e = step < 0 ? (best - 1 >= 0 ? best - 1 : count1 - 1) : (best + 1 < count1 ? best + 1 : 0);
    const float se = at(e);
    if (!(se > bestSep)) break;
    best = e;
    bestSep = se;
  }
  *edgeIndex = best;
  return bestSep;
}

// (polygon, polygon): b2CollidePolygons (b2CollidePolygon.cpp, Box2D 2.3.0 reference-face rule).  Polygon 1 carries the
// reference face, polygon 2 the incident edge (b2FindIncidentEdge: the face of 2 most anti-parallel to the reference normal);
// the incident edge is clipped against the side planes of the reference face and what stays within reach becomes the manifold.
BLCD_HD static inline void CollidePolygons(Manifold* m, const Shape* polyA, const Transform& xfA, const Shape* polyB, const Transform& xfB) {
  m->pointCount = 0;
  const float reach = polyA->radius + polyB->radius;
  int edgeA = 0, edgeB = 0;
  const float sepA = FindMaxSeparation(&edgeA, polyA, xfA, polyB, xfB);
  if (sepA > reach) return;
  const float sepB = FindMaxSeparation(&edgeB, polyB, xfB, polyA, xfA);
  if (sepB > reach) return;
  const bool refIsB = sepB > 0.98f * sepA + 0.001f;   // k_relativeTol, k_absoluteTol
  const Shape* poly1 = refIsB ? polyB : polyA;
  const Shape* poly2 = refIsB ? polyA : polyB;
  const Transform xf1 = PickXf(refIsB, xfB, xfA), xf2 = PickXf(refIsB, xfA, xfB);
  const int e1 = refIsB ? edgeB : edgeA;
  m->type = refIsB ? kManifoldFaceB : kManifoldFaceA;
  // incident edge
  const Vec2 refNormalIn2 = MulT(xf2.q, Mul(xf1.q, poly1->n[e1]));
  const int count2 = poly2->count;
  int inc = 0;
  float least = kMaxFloat;
  for (int i = 0; i < count2; ++i) {
    const float d = Dot(refNormalIn2, poly2->n[i]);
    if (d < least) {
      least = d;
      inc = i;
    }
  }
  const int inc2 = inc + 1 < count2 ? inc + 1 : 0;
  Vec2 c0 = Mul(xf2, poly2->v[inc]), c1 = Mul(xf2, poly2->v[inc2]);
  uint32_t k0 = FeatureKey(e1, inc, kFeatureFace, kFeatureVertex), k1 = FeatureKey(e1, inc2, kFeatureFace, kFeatureVertex);
  // reference face
  const int count1 = poly1->count;
  const int e2 = e1 + 1 < count1 ? e1 + 1 : 0;
  const Vec2 a1 = poly1->v[e1], a2 = poly1->v[e2];
  Vec2 localTangent = a2 - a1;
  Normalize(localTangent);
  const Vec2 tangent = Mul(xf1.q, localTangent);
  const Vec2 normal = Cross(tangent, 1.0f);
  const Vec2 w1 = Mul(xf1, a1), w2 = Mul(xf1, a2);
  const float frontOffset = Dot(normal, w1);
  if (ClipPair(c0, k0, c1, k1, -tangent, -Dot(tangent, w1) + reach, e1) < 2) return;
  if (ClipPair(c0, k0, c1, k1, tangent, Dot(tangent, w2) + reach, e2) < 2) return;
  m->localNormal = Cross(localTangent, 1.0f);
  m->localPoint = 0.5f * (a1 + a2);
  int kept = 0;
#pragma unroll
  for (int i = 0; i < kMaxManifoldPoints; ++i) {
    const Vec2 c = i == 0 ? c0 : c1;
    const uint32_t key = i == 0 ? k0 : k1;
    if (Dot(normal, c) - frontOffset <= reach) {
      ManifoldPoint cp;
      cp.normalImpulse = 0.0f;
      cp.tangentImpulse = 0.0f;
      cp.localPoint = MulT(xf2, c);
      cp.id.key = refIsB ? SwapFeatureSides(key) : key;
      PutManifoldPoint(m, kept, cp);
      ++kept;
    }
  }
  m->pointCount = kept;
}

// (wall, circle) and (wall, polygon) - b2CollideEdgeAndCircle, b2EPCollider - live in blcd_collide_wall.h

// pv[idx] for a run-time idx as a compare/select chain (keeps the array in registers)
template <size_t... I>
BLCD_HD static inline Vec2 SelVecImpl(const Vec2 (&a)[kShapeVerts], int idx, std::index_sequence<I...>) {
  float x = a[0].x, y = a[0].y;
  ((x = (idx == (int)(I + 1)) ? a[I + 1].x : x, y = (idx == (int)(I + 1)) ? a[I + 1].y : y), ...);
  return Vec2{x, y};
}
BLCD_HD static inline Vec2 SelVec(const Vec2 (&a)[kShapeVerts], int idx) {
  return SelVecImpl(a, idx, std::make_index_sequence<kShapeVerts - 1>{});
}

}  // namespace blcd
