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
  Vec2 c = V2(0.0f, 0.0f);
  float area = 0.0f;
  Vec2 pRef = V2(0.0f, 0.0f);
  const float inv3 = 1.0f / 3.0f;
  for (int i = 0; i < count; ++i) {
    Vec2 p1 = pRef;
    Vec2 p2 = vs[i];
    Vec2 p3 = i + 1 < count ? vs[i + 1] : vs[0];
    Vec2 e1 = p2 - p1;
    Vec2 e2 = p3 - p1;
    float D = Cross(e1, e2);
    float triangleArea = 0.5f * D;
    area += triangleArea;
    c += triangleArea * inv3 * (p1 + p2 + p3);
  }
  c *= 1.0f / area;
  return c;
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
BLCD_HD static inline void ShapeComputeAABB(const Shape* s, AABB* aabb, const Transform& xf) {
  if (s->type == kCircle) {
    CircleComputeAABB(s, aabb, xf);
    return;
  }
  if (s->type == kEdge) {
    Vec2 v1 = Mul(xf, s->v[0]);
    Vec2 v2 = Mul(xf, s->v[1]);
    Vec2 lower = Min(v1, v2);
    Vec2 upper = Max(v1, v2);
    Vec2 r = V2(s->radius, s->radius);
    aabb->lo = lower - r;
    aabb->hi = upper + r;
    return;
  }
  Vec2 lower = Mul(xf, s->v[0]);
  Vec2 upper = lower;
#pragma unroll
  for (int i = 1; i < kShapeVerts; ++i) {
    if (i < s->count) {
      Vec2 v = Mul(xf, s->v[i]);
      lower = Min(lower, v);
      upper = Max(upper, v);
    }
  }
  Vec2 r = V2(s->radius, s->radius);
  aabb->lo = lower - r;
  aabb->hi = upper + r;
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
struct ClipVertex {
  Vec2 v;
  ContactID id;
};

struct WorldManifold {
  Vec2 normal;
  Vec2 points[kMaxManifoldPoints];
  float separations[kMaxManifoldPoints];
  // b2WorldManifold::Initialize (b2Collision.cpp)
  BLCD_HD void Initialize(const Manifold* manifold, const Transform& xfA, float radiusA, const Transform& xfB, float radiusB) {
    if (manifold->pointCount == 0) return;
    switch (manifold->type) {
      case kManifoldCircles: {
        normal = V2(1.0f, 0.0f);
        Vec2 pointA = Mul(xfA, manifold->localPoint);
        Vec2 pointB = Mul(xfB, manifold->points[0].localPoint);
        if (DistanceSquared(pointA, pointB) > kEpsilon * kEpsilon) {
          normal = pointB - pointA;
          Normalize(normal);
        }
        Vec2 cA = pointA + radiusA * normal;
        Vec2 cB = pointB - radiusB * normal;
        points[0] = 0.5f * (cA + cB);
        separations[0] = Dot(cB - cA, normal);
      } break;
      case kManifoldFaceA: {
        normal = Mul(xfA.q, manifold->localNormal);
        Vec2 planePoint = Mul(xfA, manifold->localPoint);
#pragma unroll
        for (int i = 0; i < kMaxManifoldPoints; ++i) {
          if (i >= manifold->pointCount) break;
          Vec2 clipPoint = Mul(xfB, manifold->points[i].localPoint);
          Vec2 cA = clipPoint + (radiusA - Dot(clipPoint - planePoint, normal)) * normal;
          Vec2 cB = clipPoint - radiusB * normal;
          points[i] = 0.5f * (cA + cB);
          separations[i] = Dot(cB - cA, normal);
        }
      } break;
      case kManifoldFaceB: {
        normal = Mul(xfB.q, manifold->localNormal);
        Vec2 planePoint = Mul(xfB, manifold->localPoint);
#pragma unroll
        for (int i = 0; i < kMaxManifoldPoints; ++i) {
          if (i >= manifold->pointCount) break;
          Vec2 clipPoint = Mul(xfA, manifold->points[i].localPoint);
          Vec2 cB = clipPoint + (radiusB - Dot(clipPoint - planePoint, normal)) * normal;
          Vec2 cA = clipPoint - radiusA * normal;
          points[i] = 0.5f * (cA + cB);
          separations[i] = Dot(cA - cB, normal);
        }
        normal = -normal;
      } break;
    }
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

// b2ClipSegmentToLine (b2Collision.cpp)
BLCD_HD static inline int ClipSegmentToLine(ClipVertex vOut[2], const ClipVertex vIn[2], Vec2 normal, float offset, int vertexIndexA) {
  // Same selection as Box2D's `vOut[numOut++] = ...` sequence, written without run-time indexed stores (register
  // residency): when the segment crosses the line exactly one end point is inside, so the interpolated vertex lands in
  // vOut[1]; when both are inside vOut = vIn.
  float distance0 = Dot(normal, vIn[0].v) - offset;
  float distance1 = Dot(normal, vIn[1].v) - offset;
  const bool in0 = distance0 <= 0.0f, in1 = distance1 <= 0.0f;
  int numOut = (in0 ? 1 : 0) + (in1 ? 1 : 0);
  ClipVertex o0 = in0 ? vIn[0] : vIn[1];
  ClipVertex o1 = vIn[1];
  if (distance0 * distance1 < 0.0f) {
    float interp = distance0 / (distance0 - distance1);
    o1.v = vIn[0].v + interp * (vIn[1].v - vIn[0].v);
    o1.id.cf.indexA = (uint8_t)vertexIndexA;
    o1.id.cf.indexB = vIn[0].id.cf.indexB;
    o1.id.cf.typeA = kFeatureVertex;
    o1.id.cf.typeB = kFeatureFace;
    ++numOut;
  }
  vOut[0] = o0;
  vOut[1] = o1;
  return numOut;
}

// b2CollideCircles (b2CollideCircle.cpp)
BLCD_HD static inline void CollideCircles(Manifold* manifold, const Shape* circleA, const Transform& xfA, const Shape* circleB,
                                  const Transform& xfB) {
  manifold->pointCount = 0;
  Vec2 pA = Mul(xfA, circleA->v[0]);
  Vec2 pB = Mul(xfB, circleB->v[0]);
  Vec2 d = pB - pA;
  float distSqr = Dot(d, d);
  float rA = circleA->radius, rB = circleB->radius;
  float radius = rA + rB;
  if (distSqr > radius * radius) return;
  manifold->type = kManifoldCircles;
  manifold->localPoint = circleA->v[0];
  manifold->localNormal = V2(0.0f, 0.0f);
  manifold->pointCount = 1;
  manifold->points[0].localPoint = circleB->v[0];
  manifold->points[0].id.key = 0;
}

// b2CollidePolygonAndCircle (b2CollideCircle.cpp)
BLCD_HD static inline void CollidePolygonAndCircle(Manifold* manifold, const Shape* polygonA, const Transform& xfA,
                                           const Shape* circleB, const Transform& xfB) {
  manifold->pointCount = 0;
  Vec2 c = Mul(xfB, circleB->v[0]);
  Vec2 cLocal = MulT(xfA, c);
  int normalIndex = 0;
  float separation = -kMaxFloat;
  float radius = polygonA->radius + circleB->radius;
  int vertexCount = polygonA->count;
  const Vec2* vertices = polygonA->v;
  const Vec2* normals = polygonA->n;
  for (int i = 0; i < vertexCount; ++i) {
    float s = Dot(normals[i], cLocal - vertices[i]);
    if (s > radius) return;
    if (s > separation) {
      separation = s;
      normalIndex = i;
    }
  }
  int vertIndex1 = normalIndex;
  int vertIndex2 = vertIndex1 + 1 < vertexCount ? vertIndex1 + 1 : 0;
  Vec2 v1 = vertices[vertIndex1];
  Vec2 v2 = vertices[vertIndex2];
  if (separation < kEpsilon) {
    manifold->pointCount = 1;
    manifold->type = kManifoldFaceA;
    manifold->localNormal = normals[normalIndex];
    manifold->localPoint = 0.5f * (v1 + v2);
    manifold->points[0].localPoint = circleB->v[0];
    manifold->points[0].id.key = 0;
    return;
  }
  float u1 = Dot(cLocal - v1, v2 - v1);
  float u2 = Dot(cLocal - v2, v1 - v2);
  if (u1 <= 0.0f) {
    if (DistanceSquared(cLocal, v1) > radius * radius) return;
    manifold->pointCount = 1;
    manifold->type = kManifoldFaceA;
    manifold->localNormal = cLocal - v1;
    Normalize(manifold->localNormal);
    manifold->localPoint = v1;
    manifold->points[0].localPoint = circleB->v[0];
    manifold->points[0].id.key = 0;
  } else if (u2 <= 0.0f) {
    if (DistanceSquared(cLocal, v2) > radius * radius) return;
    manifold->pointCount = 1;
    manifold->type = kManifoldFaceA;
    manifold->localNormal = cLocal - v2;
    Normalize(manifold->localNormal);
    manifold->localPoint = v2;
    manifold->points[0].localPoint = circleB->v[0];
    manifold->points[0].id.key = 0;
  } else {
    Vec2 faceCenter = 0.5f * (v1 + v2);
    float sep = Dot(cLocal - faceCenter, normals[vertIndex1]);
    if (sep > radius) return;
    manifold->pointCount = 1;
    manifold->type = kManifoldFaceA;
    manifold->localNormal = normals[vertIndex1];
    manifold->localPoint = faceCenter;
    manifold->points[0].localPoint = circleB->v[0];
    manifold->points[0].id.key = 0;
  }
}

// b2FindMaxSeparation + b2EdgeSeparation (b2CollidePolygon.cpp, Box2D 2.3.0): separation of poly2 from edge e of poly1 =
// Dot(xf2*v2[support] - xf1*v1[e], n1World[e]) with the support vertex chosen in poly2's frame; the edge is found by a hill
// climb that starts at the edge whose normal faces poly2's centroid.  The separation of an edge is a pure function of the
// pose, so this kernel-side formulation evaluates all (<= 8) edges branch-free first - poly2's world vertices once - and
// then replays the climb's decisions on that table; the oracle evaluates lazily like upstream.
BLCD_HD static inline float FindMaxSeparation(int* edgeIndex, const Shape* poly1, const Transform& xf1, const Shape* poly2,
                                      const Transform& xf2) {
  const int count1 = poly1->count, count2 = poly2->count;
  Vec2 w2[kShapeVerts];
  for (int j = 0; j < kShapeVerts; ++j)
    if (j < count2) w2[j] = Mul(xf2, poly2->v[j]);
  float sep[kShapeVerts];
  for (int e = 0; e < kShapeVerts; ++e) {
    if (e >= count1) continue;
    const Vec2 nW = Mul(xf1.q, poly1->n[e]);
    const Vec2 nL = MulT(xf2.q, nW);
    int support = 0;
    float lowest = kMaxFloat;
    for (int j = 0; j < kShapeVerts; ++j) {
      if (j >= count2) continue;
      const float d = Dot(poly2->v[j], nL);
      if (d < lowest) {
        lowest = d;
        support = j;
      }
    }
    Vec2 far = w2[0];
    for (int j = 1; j < kShapeVerts; ++j)
      if (j == support) far = w2[j];
    sep[e] = Dot(far - Mul(xf1, poly1->v[e]), nW);
  }
  auto at = [&](int e) {
    float r = sep[0];
    for (int k = 1; k < kShapeVerts; ++k)
      if (k == e) r = sep[k];
    return r;
  };
  // starting edge: normal with the largest projection on the centroid offset (poly1 frame)
  const Vec2 dLocal1 = MulT(xf1.q, Mul(xf2, poly2->centroid) - Mul(xf1, poly1->centroid));
  int edge = 0;
  float maxDot = -kMaxFloat;
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

// b2FindIncidentEdge (b2CollidePolygon.cpp)
BLCD_HD static inline void FindIncidentEdge(ClipVertex c[2], const Shape* poly1, const Transform& xf1, int edge1, const Shape* poly2,
                                    const Transform& xf2) {
  const Vec2* normals1 = poly1->n;
  int count2 = poly2->count;
  const Vec2* vertices2 = poly2->v;
  const Vec2* normals2 = poly2->n;
  Vec2 normal1 = MulT(xf2.q, Mul(xf1.q, normals1[edge1]));
  int index = 0;
  float minDot = kMaxFloat;
  for (int i = 0; i < count2; ++i) {
    float dot = Dot(normal1, normals2[i]);
    if (dot < minDot) {
      minDot = dot;
      index = i;
    }
  }
  int i1 = index;
  int i2 = i1 + 1 < count2 ? i1 + 1 : 0;
  c[0].v = Mul(xf2, vertices2[i1]);
  c[0].id.cf.indexA = (uint8_t)edge1;
  c[0].id.cf.indexB = (uint8_t)i1;
  c[0].id.cf.typeA = kFeatureFace;
  c[0].id.cf.typeB = kFeatureVertex;
  c[1].v = Mul(xf2, vertices2[i2]);
  c[1].id.cf.indexA = (uint8_t)edge1;
  c[1].id.cf.indexB = (uint8_t)i2;
  c[1].id.cf.typeA = kFeatureFace;
  c[1].id.cf.typeB = kFeatureVertex;
}

// b2CollidePolygons (b2CollidePolygon.cpp)
BLCD_HD static inline void CollidePolygons(Manifold* manifold, const Shape* polyA, const Transform& xfA, const Shape* polyB,
                                   const Transform& xfB) {
  manifold->pointCount = 0;
  float totalRadius = polyA->radius + polyB->radius;
  int edgeA = 0;
  float separationA = FindMaxSeparation(&edgeA, polyA, xfA, polyB, xfB);
  if (separationA > totalRadius) return;
  int edgeB = 0;
  float separationB = FindMaxSeparation(&edgeB, polyB, xfB, polyA, xfA);
  if (separationB > totalRadius) return;

  const Shape* poly1;
  const Shape* poly2;
  Transform xf1, xf2;
  int edge1;
  uint8_t flip;
  if (separationB > 0.98f * separationA + 0.001f) {   // Box2D 2.3.0: k_relativeTol, k_absoluteTol
    poly1 = polyB;
    poly2 = polyA;
    xf1 = xfB;
    xf2 = xfA;
    edge1 = edgeB;
    manifold->type = kManifoldFaceB;
    flip = 1;
  } else {
    poly1 = polyA;
    poly2 = polyB;
    xf1 = xfA;
    xf2 = xfB;
    edge1 = edgeA;
    manifold->type = kManifoldFaceA;
    flip = 0;
  }
  ClipVertex incidentEdge[2];
  FindIncidentEdge(incidentEdge, poly1, xf1, edge1, poly2, xf2);
  int count1 = poly1->count;
  const Vec2* vertices1 = poly1->v;
  int iv1 = edge1;
  int iv2 = edge1 + 1 < count1 ? edge1 + 1 : 0;
  Vec2 v11 = vertices1[iv1];
  Vec2 v12 = vertices1[iv2];
  Vec2 localTangent = v12 - v11;
  Normalize(localTangent);
  Vec2 localNormal = Cross(localTangent, 1.0f);
  Vec2 planePoint = 0.5f * (v11 + v12);
  Vec2 tangent = Mul(xf1.q, localTangent);
  Vec2 normal = Cross(tangent, 1.0f);
  v11 = Mul(xf1, v11);
  v12 = Mul(xf1, v12);
  float frontOffset = Dot(normal, v11);
  float sideOffset1 = -Dot(tangent, v11) + totalRadius;
  float sideOffset2 = Dot(tangent, v12) + totalRadius;
  ClipVertex clipPoints1[2];
  ClipVertex clipPoints2[2];
  int np;
  np = ClipSegmentToLine(clipPoints1, incidentEdge, -tangent, sideOffset1, iv1);
  if (np < 2) return;
  np = ClipSegmentToLine(clipPoints2, clipPoints1, tangent, sideOffset2, iv2);
  if (np < 2) return;
  manifold->localNormal = localNormal;
  manifold->localPoint = planePoint;
  int pointCount = 0;
#pragma unroll
  for (int i = 0; i < kMaxManifoldPoints; ++i) {
    float separation = Dot(normal, clipPoints2[i].v) - frontOffset;
    if (separation <= totalRadius) {
      ManifoldPoint cp;
      cp.normalImpulse = 0.0f;
      cp.tangentImpulse = 0.0f;
      cp.localPoint = MulT(xf2, clipPoints2[i].v);
      cp.id = clipPoints2[i].id;
      if (flip) {
        ContactFeature cf = cp.id.cf;
        cp.id.cf.indexA = cf.indexB;
        cp.id.cf.indexB = cf.indexA;
        cp.id.cf.typeA = cf.typeB;
        cp.id.cf.typeB = cf.typeA;
      }
      PutManifoldPoint(manifold, pointCount, cp);
      ++pointCount;
    }
  }
  manifold->pointCount = pointCount;
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
