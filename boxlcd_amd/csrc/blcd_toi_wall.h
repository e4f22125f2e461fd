// blcd_toi_wall.h — register-resident time of impact for (static wall edge A, moving circle/polygon B).
//
// Same algorithm and the same float operations as blcd_toi.h's generic b2Distance / b2TimeOfImpact (Box2D 2.3.x
// b2Distance.cpp, b2TimeOfImpact.cpp), restructured for CDNA: without bullets boxLCD only ever asks
// for the TOI of a dynamic shape against one of the four wall edges, whose sweep is the identity at every time, so
//   * the wall transform drops out (x*1 - y*0 + 0 == x in value; zero signs are irrelevant downstream),
//   * the simplex, the GJK cache and both distance proxies live in named registers: every array access of the generic
//     code (`vertices[m_count]`, `cache->indexA[i]`, `proxy->GetVertex(i)`) becomes a short compare-select chain, so nothing
//     is spilled to scratch and no dependent global loads sit inside the GJK / root-finder loops.
// Measured before this file: ~54 k cycles per TimeOfImpact call in the slowest waves (scratch latency), >50 % of step time.
#pragma once
#include "blcd_toi.h"

namespace blcd {

template <int MAXV>
struct ProxyR {  // B's core shape in registers
  float vx[MAXV], vy[MAXV];
  int count;
  float radius;
  __device__ __forceinline__ void load(const Shape* s) {
    count = s->type == kCircle ? 1 : s->count;
    radius = s->radius;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      bool live = i < count;
      vx[i] = live ? s->v[i].x : 0.0f;
      vy[i] = live ? s->v[i].y : 0.0f;
    }
  }
  __device__ __forceinline__ Vec2 vertex(int idx) const {
    float x = vx[0], y = vy[0];
#pragma unroll
    for (int i = 1; i < MAXV; ++i) {
      x = idx == i ? vx[i] : x;
      y = idx == i ? vy[i] : y;
    }
    return V2(x, y);
  }
  __device__ __forceinline__ int support(Vec2 d) const {
    int bestIndex = 0;
    float bestValue = vx[0] * d.x + vy[0] * d.y;
#pragma unroll
    for (int i = 1; i < MAXV; ++i) {
      float value = vx[i] * d.x + vy[i] * d.y;
      bool better = i < count && value > bestValue;
      bestIndex = better ? i : bestIndex;
      bestValue = better ? value : bestValue;
    }
    return bestIndex;
  }
};

struct EdgeR {  // wall edge (A): two vertices, identity transform
  Vec2 a0, a1;
  float radius;
  __device__ __forceinline__ Vec2 vertex(int idx) const { return idx == 0 ? a0 : a1; }
  __device__ __forceinline__ int support(Vec2 d) const {
    float v0 = Dot(a0, d), v1 = Dot(a1, d);
    return v1 > v0 ? 1 : 0;
  }
};

struct SVtx {
  Vec2 wA, wB, w;
  float a;
  int indexA, indexB;
};

struct CacheR {
  float metric;
  int count;
  int iA0, iA1, iA2, iB0, iB1, iB2;
};

// DEAD: the caller guarantees a one-vertex proxy at the body origin with the centre of mass there (see xfAt)
template <int MAXV, bool DEAD = false>
struct TOIWall {
  EdgeR A;
  ProxyR<MAXV> B;
  Sweep sweepB;

  // ---- b2Simplex in registers ----
  SVtx v1, v2, v3;
  int count;

  __device__ __forceinline__ float metric() const {
    if (count == 1) return 0.0f;
    if (count == 2) return Distance(v1.w, v2.w);
    return Cross(v2.w - v1.w, v3.w - v1.w);
  }
  __device__ __forceinline__ void makeVertex(SVtx& v, int ia, int ib, const Transform& xfB) const {
    v.indexA = ia;
    v.indexB = ib;
    v.wA = A.vertex(ia);             // identity transform
    v.wB = DEAD ? xfB.p : Mul(xfB, B.vertex(ib));   // DEAD: the vertex is the body origin (q*0 + p == p)
    v.w = v.wB - v.wA;
  }
  __device__ __forceinline__ void readCache(const CacheR& c, const Transform& xfB) {
    count = c.count;
    if (count > 0) { makeVertex(v1, c.iA0, c.iB0, xfB); v1.a = 0.0f; }
    if (count > 1) { makeVertex(v2, c.iA1, c.iB1, xfB); v2.a = 0.0f; }
    if (MAXV > 1 && count > 2) { makeVertex(v3, c.iA2, c.iB2, xfB); v3.a = 0.0f; }
    if (count > 1) {
      float metric1 = c.metric;
      float metric2 = metric();
      if (metric2 < 0.5f * metric1 || 2.0f * metric1 < metric2 || metric2 < kEpsilon) count = 0;
    }
    if (count == 0) {
      makeVertex(v1, 0, 0, xfB);
      v1.a = 1.0f;
      count = 1;
    }
  }
  __device__ __forceinline__ void writeCache(CacheR& c) const {
    c.metric = metric();
    c.count = count;
    c.iA0 = v1.indexA; c.iB0 = v1.indexB;
    if (count > 1) { c.iA1 = v2.indexA; c.iB1 = v2.indexB; }
    if (count > 2) { c.iA2 = v3.indexA; c.iB2 = v3.indexB; }
  }
  __device__ __forceinline__ void solve2() {
    Vec2 w1 = v1.w, w2 = v2.w;
    Vec2 e12 = w2 - w1;
    float d12_2 = -Dot(w1, e12);
    if (d12_2 <= 0.0f) { v1.a = 1.0f; count = 1; return; }
    float d12_1 = Dot(w2, e12);
    if (d12_1 <= 0.0f) { v2.a = 1.0f; count = 1; v1 = v2; return; }
    float inv_d12 = 1.0f / (d12_1 + d12_2);
    v1.a = d12_1 * inv_d12;
    v2.a = d12_2 * inv_d12;
    count = 2;
  }
  __device__ __forceinline__ void solve3() {
    Vec2 w1 = v1.w, w2 = v2.w, w3 = v3.w;
    Vec2 e12 = w2 - w1;
    float w1e12 = Dot(w1, e12), w2e12 = Dot(w2, e12);
    float d12_1 = w2e12, d12_2 = -w1e12;
    Vec2 e13 = w3 - w1;
    float w1e13 = Dot(w1, e13), w3e13 = Dot(w3, e13);
    float d13_1 = w3e13, d13_2 = -w1e13;
    Vec2 e23 = w3 - w2;
    float w2e23 = Dot(w2, e23), w3e23 = Dot(w3, e23);
    float d23_1 = w3e23, d23_2 = -w2e23;
    float n123 = Cross(e12, e13);
    float d123_1 = n123 * Cross(w2, w3);
    float d123_2 = n123 * Cross(w3, w1);
    float d123_3 = n123 * Cross(w1, w2);
    if (d12_2 <= 0.0f && d13_2 <= 0.0f) { v1.a = 1.0f; count = 1; return; }
    if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f) {
      float inv_d12 = 1.0f / (d12_1 + d12_2);
      v1.a = d12_1 * inv_d12; v2.a = d12_2 * inv_d12; count = 2; return;
    }
    if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f) {
      float inv_d13 = 1.0f / (d13_1 + d13_2);
      v1.a = d13_1 * inv_d13; v3.a = d13_2 * inv_d13; count = 2; v2 = v3; return;
    }
    if (d12_1 <= 0.0f && d23_2 <= 0.0f) { v2.a = 1.0f; count = 1; v1 = v2; return; }
    if (d13_1 <= 0.0f && d23_1 <= 0.0f) { v3.a = 1.0f; count = 1; v1 = v3; return; }
    if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f) {
      float inv_d23 = 1.0f / (d23_1 + d23_2);
      v2.a = d23_1 * inv_d23; v3.a = d23_2 * inv_d23; count = 2; v1 = v3; return;
    }
    float inv_d123 = 1.0f / (d123_1 + d123_2 + d123_3);
    v1.a = d123_1 * inv_d123; v2.a = d123_2 * inv_d123; v3.a = d123_3 * inv_d123; count = 3;
  }

  // b2Distance (useRadii = false); returns the distance, updates the cache
  __device__ __forceinline__ float distance(CacheR& cache, const Transform& xfB) {
    readCache(cache, xfB);
    const int k_maxIters = 20;
    int iter = 0;
    while (iter < k_maxIters) {
      int saveCount = count;
      int sA0 = v1.indexA, sB0 = v1.indexB, sA1 = v2.indexA, sB1 = v2.indexB, sA2 = v3.indexA, sB2 = v3.indexB;
      // a one-vertex proxy against a two-vertex edge can never hold three distinct simplex vertices
      if (count == 2) solve2();
      else if (MAXV > 1 && count == 3) solve3();
      if (MAXV > 1 && count == 3) break;
      Vec2 d;
      if (count == 1) {
        d = -v1.w;
      } else {
        Vec2 e12 = v2.w - v1.w;
        float sgn = Cross(e12, -v1.w);
        d = sgn > 0.0f ? Cross(1.0f, e12) : Cross(e12, 1.0f);
      }
      if (LengthSquared(d) < kEpsilon * kEpsilon) break;
      SVtx nv;
      int ia = A.support(-d);                    // MulT(identity, -d) == -d
      int ib = B.support(MulT(xfB.q, d));
      makeVertex(nv, ia, ib, xfB);
      nv.a = 0.0f;
      ++iter;
      bool duplicate = (saveCount > 0 && ia == sA0 && ib == sB0) || (saveCount > 1 && ia == sA1 && ib == sB1) ||
                       (saveCount > 2 && ia == sA2 && ib == sB2);
      // the new vertex is written at vertices[m_count] even when it turns out to be a duplicate (the generic code does
      // the same; the slot is beyond m_count so it is never read)
      if (count == 1) v2 = nv; else v3 = nv;
      if (duplicate) break;
      ++count;
    }
    Vec2 pA, pB;
    if (count == 1) {
      pA = v1.wA;
      pB = v1.wB;
    } else if (count == 2) {
      pA = v1.a * v1.wA + v2.a * v2.wA;
      pB = v1.a * v1.wB + v2.a * v2.wB;
    } else {
      pA = v1.a * v1.wA + v2.a * v2.wA + v3.a * v3.wA;
      pB = pA;
    }
    float dist = Distance(pA, pB);
    writeCache(cache);
    return dist;
  }

  // ---- exact memo of sweepB.GetTransform(t): the TOI iteration asks for the same t repeatedly (t1 in the outer loop, the
  // separation function's Initialize/Evaluate; tMax in every FindMinSeparation; the root it just found).  The sweep is
  // fixed during one TOI query, so GetTransform is a pure function of t and replaying a stored result is bit-identical.
  bool qDead;               // see xfAt
  float mkA, mkB, mkC;      // keys (NaN = empty)
  Transform mxA, mxB, mxC;  // A: current t1, B: tMax, C: most recent other t
  __device__ __forceinline__ void memoReset() {
    mkA = mkB = mkC = __builtin_nanf("");
  }
  __device__ __forceinline__ Transform xfAt(float t) {
    if (DEAD) {  // two multiply-adds per coordinate: cheaper than consulting the memo
      Transform xf;
      xf.p = (1.0f - t) * sweepB.c0 + t * sweepB.c;
      xf.q.s = 0.0f;
      xf.q.c = 1.0f;
      return xf;
    }
    if (t == mkA) return mxA;
    if (t == mkB) return mxB;
    if (t == mkC) return mxC;
    Transform xf;
    if (DEAD || (MAXV == 1 && qDead)) {
      // circle whose centre is the body origin (and the centre of mass): every use of q below multiplies the zero vector
      // (vertex (0,0), localCenter (0,0); the support of a 1-vertex proxy is always 0), so sincosf is skipped.  Values are
      // unchanged; only the sign of an exact zero could differ.
      xf.p = (1.0f - t) * sweepB.c0 + t * sweepB.c;
      xf.q.s = 0.0f;
      xf.q.c = 1.0f;
    } else {
      sweepB.GetTransform(&xf, t);
    }
    mkC = t;
    mxC = xf;
    return xf;
  }
  __device__ __forceinline__ void memoPinT1(float t, const Transform& xf) {
    mkA = t;
    mxA = xf;
  }
  __device__ __forceinline__ void memoPinMax(float t, const Transform& xf) {
    mkB = t;
    mxB = xf;
  }

  // ---- b2SeparationFunction ----
  int sfType;  // 0 points, 1 faceA, 2 faceB
  Vec2 sfLocalPoint, sfAxis;

  __device__ __forceinline__ void sfInitialize(const CacheR& cache, float t1) {
    Transform xfB = xfAt(t1);
    if (cache.count == 1) {
      sfType = 0;
      Vec2 pointA = A.vertex(cache.iA0);
      Vec2 pointB = Mul(xfB, B.vertex(cache.iB0));
      sfAxis = pointB - pointA;
      Normalize(sfAxis);
    } else if (MAXV > 1 && cache.iA0 == cache.iA1) {   // two distinct points on B: impossible for a one-vertex proxy
      sfType = 2;
      Vec2 localPointB1 = B.vertex(cache.iB0);
      Vec2 localPointB2 = B.vertex(cache.iB1);
      sfAxis = Cross(localPointB2 - localPointB1, 1.0f);
      Normalize(sfAxis);
      Vec2 normal = Mul(xfB.q, sfAxis);
      sfLocalPoint = 0.5f * (localPointB1 + localPointB2);
      Vec2 pointB = Mul(xfB, sfLocalPoint);
      Vec2 pointA = A.vertex(cache.iA0);
      float s = Dot(pointA - pointB, normal);
      if (s < 0.0f) sfAxis = -sfAxis;
    } else {
      sfType = 1;
      Vec2 localPointA1 = A.vertex(cache.iA0);
      Vec2 localPointA2 = A.vertex(cache.iA1);
      sfAxis = Cross(localPointA2 - localPointA1, 1.0f);
      Normalize(sfAxis);
      Vec2 normal = sfAxis;  // Mul(identity.q, axis)
      sfLocalPoint = 0.5f * (localPointA1 + localPointA2);
      Vec2 pointA = sfLocalPoint;
      Vec2 pointB = Mul(xfB, B.vertex(cache.iB0));
      float s = Dot(pointB - pointA, normal);
      if (s < 0.0f) sfAxis = -sfAxis;
    }
  }
  __device__ __forceinline__ float sfFindMinSeparation(int* indexA, int* indexB, float t) {
    Transform xfB = xfAt(t);
    if (sfType == 0) {
      *indexA = A.support(sfAxis);
      *indexB = B.support(MulT(xfB.q, -sfAxis));
      Vec2 pointA = A.vertex(*indexA);
      Vec2 pointB = DEAD ? xfB.p : Mul(xfB, B.vertex(*indexB));
      return Dot(pointB - pointA, sfAxis);
    } else if (MAXV == 1 || sfType == 1) {
      Vec2 normal = sfAxis;
      Vec2 pointA = sfLocalPoint;
      *indexA = -1;
      *indexB = B.support(MulT(xfB.q, -normal));
      Vec2 pointB = DEAD ? xfB.p : Mul(xfB, B.vertex(*indexB));
      return Dot(pointB - pointA, normal);
    } else {
      Vec2 normal = Mul(xfB.q, sfAxis);
      Vec2 pointB = Mul(xfB, sfLocalPoint);
      *indexB = -1;
      *indexA = A.support(-normal);
      Vec2 pointA = A.vertex(*indexA);
      return Dot(pointA - pointB, normal);
    }
  }
  __device__ __forceinline__ float sfEvaluate(int indexA, int indexB, float t) {
    Transform xfB = xfAt(t);
    if (sfType == 0) {
      Vec2 pointA = A.vertex(indexA);
      Vec2 pointB = DEAD ? xfB.p : Mul(xfB, B.vertex(indexB));
      return Dot(pointB - pointA, sfAxis);
    } else if (MAXV == 1 || sfType == 1) {
      Vec2 normal = sfAxis;
      Vec2 pointA = sfLocalPoint;
      Vec2 pointB = DEAD ? xfB.p : Mul(xfB, B.vertex(indexB));
      return Dot(pointB - pointA, normal);
    } else {
      Vec2 normal = Mul(xfB.q, sfAxis);
      Vec2 pointB = Mul(xfB, sfLocalPoint);
      Vec2 pointA = A.vertex(indexA);
      return Dot(pointA - pointB, normal);
    }
  }

  // b2TimeOfImpact with tMax = 1
  __device__ __forceinline__ void run(TOIOutput* output, const Sweep& sweepB_in) {
    const float tMax = 1.0f;
    output->state = kTOIUnknown;
    output->t = tMax;
    sweepB = sweepB_in;
    sweepB.Normalize();
    qDead = MAXV == 1 && B.vx[0] == 0.0f && B.vy[0] == 0.0f && sweepB.localCenter.x == 0.0f && sweepB.localCenter.y == 0.0f;
    float totalRadius = A.radius + B.radius;
    float target = Max(kLinearSlop, totalRadius - 3.0f * kLinearSlop);
    float tolerance = 0.25f * kLinearSlop;
    float t1 = 0.0f;
    const int k_maxIterations = 20;
    int iter = 0;
    CacheR cache;
    cache.count = 0;
    cache.metric = 0.0f;
    cache.iA0 = cache.iA1 = cache.iA2 = cache.iB0 = cache.iB1 = cache.iB2 = 0;
    v1.indexA = v1.indexB = v2.indexA = v2.indexB = v3.indexA = v3.indexB = 0;
    memoReset();
    for (;;) {
      Transform xfB = xfAt(t1);
      memoPinT1(t1, xfB);
      float dist = distance(cache, xfB);
      if (dist <= 0.0f) {
        output->state = kTOIOverlapped;
        output->t = 0.0f;
        break;
      }
      if (dist < target + tolerance) {
        output->state = kTOITouching;
        output->t = t1;
        break;
      }
      sfInitialize(cache, t1);
      bool done = false;
      float t2 = tMax;
      if (!(mkB == tMax)) {
        Transform xm = xfAt(tMax);
        memoPinMax(tMax, xm);
      }
      int pushBackIter = 0;
      for (;;) {
        int indexA, indexB;
        float s2 = sfFindMinSeparation(&indexA, &indexB, t2);
        if (s2 > target + tolerance) {
          output->state = kTOISeparated;
          output->t = tMax;
          done = true;
          break;
        }
        if (s2 > target - tolerance) {
          t1 = t2;
          break;
        }
        float s1 = sfEvaluate(indexA, indexB, t1);
        if (s1 < target - tolerance) {
          output->state = kTOIFailed;
          output->t = t1;
          done = true;
          break;
        }
        if (s1 <= target + tolerance) {
          output->state = kTOITouching;
          output->t = t1;
          done = true;
          break;
        }
        int rootIterCount = 0;
        float a1 = t1, a2 = t2;
        for (;;) {
          float t;
          if (rootIterCount & 1) t = a1 + (target - s1) * (a2 - a1) / (s2 - s1);
          else t = 0.5f * (a1 + a2);
          ++rootIterCount;
          float s = sfEvaluate(indexA, indexB, t);
          if (Abs(s - target) < tolerance) {
            t2 = t;
            break;
          }
          if (s > target) {
            a1 = t;
            s1 = s;
          } else {
            a2 = t;
            s2 = s;
          }
          if (rootIterCount == 50) break;
        }
        ++pushBackIter;
        if (pushBackIter == kMaxPolygonVertices) break;
      }
      ++iter;
      if (done) break;
      if (iter == k_maxIterations) {
        output->state = kTOIFailed;
        output->t = t1;
        break;
      }
    }
  }
};

}  // namespace blcd
