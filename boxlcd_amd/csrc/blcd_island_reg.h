// blcd_island_reg.h — staged b2Island::Solve for small islands (<= NJR joints, <= NCR contacts): constraints in registers,
// body rows and the contacts' sweep constants in registers (2-3 bodies) or in the wave's LDS (>= 4 bodies).
//
// The 540 Gauss-Seidel velocity sweeps per env step are the hot loop of the path (b2Island::Solve,
// b2ContactSolver::SolveVelocityConstraints, b2RevoluteJoint::SolveVelocityConstraints; reference call site
// boxLCD/world_env.py:448-450).  In the generic Env the constraint arrays are indexed with run-time values (island order is
// dynamic), which keeps them in scratch: measured 27 k cycles per sweep for an Urchin (3 joints + ~3 contacts).  Here the
// island's working set is staged once per world step into statically indexed structures, in island (DFS) order, so that
// the sweeps run out of VGPRs and LDS:
//   * constraints k = 0..n-1 are unrolled (static index), each remembers its two body ids;
//   * body velocities/positions/masses are rows addressed by body id: compare-select chains over small register arrays for
//     2-3 bodies, an LDS block [word][lane] for more (see RegIsland below; 17 k -> 11 k cycles per sweep on the final build).
// Arithmetic and operation order are exactly those of the generic path (and of Box2D); the parity tests compare both
// against the CPU oracle.  Islands that do not fit fall back to the generic path.
#pragma once
#include "blcd_collide.h"

namespace blcd {

// sweeps during which the short-cycle detector of the joint-free islands watches (reference rows after sweeps 1, 2, 4, ..); afterwards the
// sweeps run untracked (velocitySweeps)
#ifndef BLCD_CYC_WATCH
#define BLCD_CYC_WATCH 48
#endif

// always-select array access (compile-time unrolled word selects; see selGet in blcd_world.h for why not a loop)
template <typename T, int N, size_t... K>
__device__ __forceinline__ void rWordsGet(const T (&a)[N], int i, int t, uint32_t* r, std::index_sequence<K...>) {
  uint32_t w[sizeof...(K)];
  __builtin_memcpy(w, &a[t], sizeof(T));
  ((r[K] = (i == t) ? w[K] : r[K]), ...);
}
template <typename T, int N, size_t... Ts>
__device__ __forceinline__ T rGetImpl(const T (&a)[N], int i, std::index_sequence<Ts...>) {
  constexpr size_t W = sizeof(T) / 4;
  uint32_t r[W];
  __builtin_memcpy(r, &a[0], sizeof(T));
  (rWordsGet(a, i, (int)(Ts + 1), r, std::make_index_sequence<W>{}), ...);
  T out;
  __builtin_memcpy(&out, r, sizeof(T));
  return out;
}
template <int N, typename T>
__device__ __forceinline__ T rGet(const T (&a)[N], int i) {
  return rGetImpl(a, i, std::make_index_sequence<N - 1>{});
}
template <typename T, int N, size_t... K>
__device__ __forceinline__ void rWordsSet(T (&a)[N], int i, int t, const uint32_t* v, std::index_sequence<K...>) {
  uint32_t w[sizeof...(K)];
  __builtin_memcpy(w, &a[t], sizeof(T));
  ((w[K] = (i == t) ? v[K] : w[K]), ...);
  __builtin_memcpy(&a[t], w, sizeof(T));
}
template <typename T, int N, size_t... Ts>
__device__ __forceinline__ void rSetImpl(T (&a)[N], int i, const T& val, std::index_sequence<Ts...>) {
  constexpr size_t W = sizeof(T) / 4;
  uint32_t v[W];
  __builtin_memcpy(v, &val, sizeof(T));
  (rWordsSet(a, i, (int)Ts, v, std::make_index_sequence<W>{}), ...);
}
template <int N, typename T>
__device__ __forceinline__ void rSet(T (&a)[N], int i, const T& v) {
  rSetImpl(a, i, v, std::make_index_sequence<N>{});
}

struct BodyVel {
  Vec2 v;
  float w;
};
struct BodyPos {
  Vec2 c;
  float a;
};
struct BodyMass {
  float invMass, invI;
  Vec2 lc;
};

struct RJoint {  // one revolute joint in island order
  int A, B;      // dynamic-body indices
  Vec2 anchorA, anchorB;
  int enableLimit;
  float lower, upper, maxMotorTorque, ref, speed;
  Vec3 imp;
  float motor;
  int limit;
  Vec2 rA, rB;
  Mat33 mass;
  float motorMass;
  // sweep-invariant parts of b2Mat33::Solve33 / Solve22 (the mass matrix is fixed during the velocity iterations): same
  // expressions as upstream, evaluated once per world step instead of once per sweep
  Vec3 cyz;          // b2Cross(ey, ez)
  float det33, det22;  // the reciprocal determinants (0 stays 0)
};

struct RPoint {
  Vec2 rA, rB;
  float normalImpulse, tangentImpulse, normalMass, tangentMass, velocityBias;
};
struct RContact {  // velocity + position constraint of one contact in island order
  RPoint points[2];
  Vec2 normal;
  Mat22 normalMass, K;
  int pA, pB;  // proxy ids (0..3 walls, 4+i bodies)
  float friction, restitution;
  int pointCount;
  // position constraint (copy of the manifold: b2ContactPositionConstraint)
  int mtype, mcount;
  Vec2 localNormal, localPoint, lp0, lp1;
  float radiusA, radiusB;
};

// Body rows (velocity, position, mass data) are addressed with run-time body ids.  Two homes for them:
//  * LDS = false: small register arrays behind compare-select chains (1-2 selects per word for 2-3 bodies);
//  * LDS = true : the wave's LDS block, laid out [word][lane] (word = 10 * body + field, 64 lanes per word: every access is
//    bank-conflict free whatever body each lane asks for).  For >= 4 bodies this replaces ~550 v_cndmask, ~260
//    v_readlane/v_writelane (hoisted compare masks spilling out of the SGPR file) and part of the AGPR shuffling per
//    velocity sweep by a handful of ds_read/ds_write (ISA of the (4,3,16) class), and frees 10 * NB VGPRs.
// Pure data movement either way: the arithmetic is untouched.
//  * CLDS = true (with LDS): the sweep-invariant part of every contact's velocity constraint (lever arms, effective masses,
//    bias, normal, block matrices: 22 words per contact) is parked in a second LDS block [word][lane] with compile-time
//    offsets, and only the accumulated impulses stay in registers.  The hot set of a 3-joint + 4-contact island is ~260
//    words against 256 VGPRs, so without this ~400 v_accvgpr moves per sweep shuttle it through the AGPR half.
template <int NB, int NJR, int NCR, bool LDS = false, bool CLDS = false>
struct RegIsland {
  static constexpr int kBodyWords = 10;   // v.x v.y w | c.x c.y a | invMass invI lc.x lc.y
  static constexpr int kLdsWords = kBodyWords * NB * 64;
  static constexpr int kCtWords = 22;     // per point {rA rB normalMass tangentMass velocityBias} x2 | normal | normalMass(3) | K(3)
  static constexpr int kCtLdsWords = kCtWords * NCR * 64;
  float* C;        // CLDS mode: this lane's column of the contact-constant block
  BodyVel vel[LDS ? 1 : NB];
  BodyPos pos[LDS ? 1 : NB];
  BodyMass mass[LDS ? 1 : NB];
  float* L;        // LDS mode: this lane's column of the block (word k at L[64 * k])
  RJoint jt[NJR > 0 ? NJR : 1];
  RContact ct[NCR];
  int nj, nc;
  uint32_t deadQ;  // Env::deadQ

  // ---- body rows by dynamic-body index ----
  __device__ __forceinline__ BodyVel getVel(int i) const {
    if constexpr (LDS) {
      const float* q = L + 64 * kBodyWords * i;
      BodyVel r;
      r.v.x = q[0];
      r.v.y = q[64];
      r.w = q[128];
      return r;
    } else {
      return rGet(vel, i);
    }
  }
  __device__ __forceinline__ void setVel(int i, const BodyVel& x) {
    if constexpr (LDS) {
      float* q = L + 64 * kBodyWords * i;
      q[0] = x.v.x;
      q[64] = x.v.y;
      q[128] = x.w;
    } else {
      rSet(vel, i, x);
    }
  }
  __device__ __forceinline__ BodyPos getPos(int i) const {
    if constexpr (LDS) {
      const float* q = L + 64 * (kBodyWords * i + 3);
      BodyPos r;
      r.c.x = q[0];
      r.c.y = q[64];
      r.a = q[128];
      return r;
    } else {
      return rGet(pos, i);
    }
  }
  __device__ __forceinline__ void setPos(int i, const BodyPos& x) {
    if constexpr (LDS) {
      float* q = L + 64 * (kBodyWords * i + 3);
      q[0] = x.c.x;
      q[64] = x.c.y;
      q[128] = x.a;
    } else {
      rSet(pos, i, x);
    }
  }
  __device__ __forceinline__ BodyMass getMass(int i) const {
    if constexpr (LDS) {
      const float* q = L + 64 * (kBodyWords * i + 6);
      BodyMass r;
      r.invMass = q[0];
      r.invI = q[64];
      r.lc.x = q[128];
      r.lc.y = q[192];
      return r;
    } else {
      return rGet(mass, i);
    }
  }
  __device__ __forceinline__ void setMass(int i, const BodyMass& x) {
    if constexpr (LDS) {
      float* q = L + 64 * (kBodyWords * i + 6);
      q[0] = x.invMass;
      q[64] = x.invI;
      q[128] = x.lc.x;
      q[192] = x.lc.y;
    } else {
      rSet(mass, i, x);
    }
  }

  // ---- body accessors by proxy id (0..3 = the static walls: zero rows, writes dropped) ----
  __device__ __forceinline__ BodyVel V(int p) const {
    BodyVel z;
    z.v = V2(0.0f, 0.0f);
    z.w = 0.0f;
    BodyVel r = getVel(p < 4 ? 0 : p - 4);
    return p < 4 ? z : r;
  }
  __device__ __forceinline__ void setV(int p, const BodyVel& x) {
    if (p >= 4) setVel(p - 4, x);
  }
  __device__ __forceinline__ BodyPos P(int p) const {
    BodyPos z;
    z.c = V2(0.0f, 0.0f);
    z.a = 0.0f;
    BodyPos r = getPos(p < 4 ? 0 : p - 4);
    return p < 4 ? z : r;
  }
  __device__ __forceinline__ void setP(int p, const BodyPos& x) {
    if (p >= 4) setPos(p - 4, x);
  }
  __device__ __forceinline__ BodyMass M(int p) const {
    BodyMass z;
    z.invMass = 0.0f;
    z.invI = 0.0f;
    z.lc = V2(0.0f, 0.0f);
    BodyMass r = getMass(p < 4 ? 0 : p - 4);
    return p < 4 ? z : r;
  }

  // sweep-invariant velocity-constraint data of one contact (b2ContactVelocityConstraint minus the accumulated impulses)
  struct SweepC {
    Vec2 rA[2], rB[2];
    float normalMass[2], tangentMass[2], velocityBias[2];
    Vec2 normal;
    Mat22 nm, K;
    float friction;
  };
  template <bool WALLA = false>   // WALLA: the caller never reads rA (see sweepContact)
  __device__ __forceinline__ SweepC loadSweepC(int k, const RContact& c_) const {
    SweepC q;
    if constexpr (CLDS) {
      const float* p = C + 64 * kCtWords * k;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        q.rA[j] = WALLA ? V2(0.0f, 0.0f) : V2(p[64 * (7 * j)], p[64 * (7 * j + 1)]);
        q.rB[j] = V2(p[64 * (7 * j + 2)], p[64 * (7 * j + 3)]);
        q.normalMass[j] = p[64 * (7 * j + 4)];
        q.tangentMass[j] = p[64 * (7 * j + 5)];
        q.velocityBias[j] = p[64 * (7 * j + 6)];
      }
      q.normal = V2(p[64 * 14], p[64 * 15]);
      q.nm.ex = V2(p[64 * 16], p[64 * 17]);
      q.nm.ey = V2(p[64 * 17], p[64 * 18]);     // inverse of a symmetric matrix: ex.y == ey.x bit for bit (b2Mat22::GetInverse)
      q.K.ex = V2(p[64 * 19], p[64 * 20]);
      q.K.ey = V2(p[64 * 20], p[64 * 21]);
      q.friction = c_.friction;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        q.rA[j] = c_.points[j].rA;
        q.rB[j] = c_.points[j].rB;
        q.normalMass[j] = c_.points[j].normalMass;
        q.tangentMass[j] = c_.points[j].tangentMass;
        q.velocityBias[j] = c_.points[j].velocityBias;
      }
      q.normal = c_.normal;
      q.nm = c_.normalMass;
      q.K = c_.K;
      q.friction = c_.friction;
    }
    return q;
  }
  __device__ __forceinline__ void storeSweepC(int k, RContact& c_, const SweepC& q) {
    if constexpr (CLDS) {
      float* p = C + 64 * kCtWords * k;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        p[64 * (7 * j)] = q.rA[j].x;
        p[64 * (7 * j + 1)] = q.rA[j].y;
        p[64 * (7 * j + 2)] = q.rB[j].x;
        p[64 * (7 * j + 3)] = q.rB[j].y;
        p[64 * (7 * j + 4)] = q.normalMass[j];
        p[64 * (7 * j + 5)] = q.tangentMass[j];
        p[64 * (7 * j + 6)] = q.velocityBias[j];
      }
      p[64 * 14] = q.normal.x;
      p[64 * 15] = q.normal.y;
      p[64 * 16] = q.nm.ex.x;
      p[64 * 17] = q.nm.ex.y;
      p[64 * 18] = q.nm.ey.y;
      p[64 * 19] = q.K.ex.x;
      p[64 * 20] = q.K.ex.y;
      p[64 * 21] = q.K.ey.y;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        c_.points[j].rA = q.rA[j];
        c_.points[j].rB = q.rB[j];
        c_.points[j].normalMass = q.normalMass[j];
        c_.points[j].tangentMass = q.tangentMass[j];
        c_.points[j].velocityBias = q.velocityBias[j];
      }
      c_.normal = q.normal;
      c_.normalMass = q.nm;
      c_.K = q.K;
    }
  }

  // ---- b2ContactSolver::InitializeVelocityConstraints for contact k (k static) ----
  __device__ __forceinline__ void initContact(int k, RContact& c_, const Manifold& manifold) {
    const int pA = c_.pA, pB = c_.pB;
    BodyMass mAs = M(pA), mBs = M(pB);
    float mA = mAs.invMass, mB = mBs.invMass, iA = mAs.invI, iB = mBs.invI;
    BodyPos pa_ = P(pA), pb_ = P(pB);
    BodyVel va_ = V(pA), vb_ = V(pB);
    Vec2 cA = pa_.c, cB = pb_.c, vA = va_.v, vB = vb_.v;
    float aA = pa_.a, aB = pb_.a, wA = va_.w, wB = vb_.w;
    Transform xfA, xfB;
    xfA.q = rotDead(deadQ, pA, aA);
    xfB.q = rotDead(deadQ, pB, aB);
    xfA.p = cA - Mul(xfA.q, mAs.lc);
    xfB.p = cB - Mul(xfB.q, mBs.lc);
    WorldManifold worldManifold;
    worldManifold.Initialize(&manifold, xfA, c_.radiusA, xfB, c_.radiusB);
    SweepC q;
    q.normal = worldManifold.normal;
    q.friction = c_.friction;
    q.K.ex = q.K.ey = V2(0.0f, 0.0f);
    q.nm.ex = q.nm.ey = V2(0.0f, 0.0f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      q.rA[j] = q.rB[j] = V2(0.0f, 0.0f);
      q.normalMass[j] = q.tangentMass[j] = q.velocityBias[j] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j >= c_.pointCount) break;
      q.rA[j] = worldManifold.points[j] - cA;
      q.rB[j] = worldManifold.points[j] - cB;
      float rnA = Cross(q.rA[j], q.normal);
      float rnB = Cross(q.rB[j], q.normal);
      float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
      q.normalMass[j] = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
      Vec2 tangent = Cross(q.normal, 1.0f);
      float rtA = Cross(q.rA[j], tangent);
      float rtB = Cross(q.rB[j], tangent);
      float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
      q.tangentMass[j] = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
      q.velocityBias[j] = 0.0f;
      float vRel = Dot(q.normal, vB + Cross(wB, q.rB[j]) - vA - Cross(wA, q.rA[j]));
      if (vRel < -kVelocityThreshold) q.velocityBias[j] = -c_.restitution * vRel;
    }
    if (c_.pointCount == 2) {
      float rn1A = Cross(q.rA[0], q.normal);
      float rn1B = Cross(q.rB[0], q.normal);
      float rn2A = Cross(q.rA[1], q.normal);
      float rn2B = Cross(q.rB[1], q.normal);
      float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
      float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
      float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
      const float k_maxConditionNumber = 1000.0f;
      if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12)) {
        q.K.ex = V2(k11, k12);
        q.K.ey = V2(k12, k22);
        q.nm = q.K.GetInverse();
      } else {
        c_.pointCount = 1;
      }
    }
    storeSweepC(k, c_, q);
  }

  // Wall-side folding (WALLA).  Proxy A of a (wall, body) contact is a static body: invMass = invI = 0, v = w = +0, and
  // nothing ever writes them (setV drops wall rows).  Then, for finite operands and bit for bit:
  //   vA -= mA * P, wA -= iA * (...)          : +0 - (+-0) = +0                       -> the wall row stays +0, no need to compute it
  //   dv = vB + Cross(wB, rB) - vA - Cross(wA, rA): X - (+0) = X; X - (+-0) = X unless X is -0 and the subtrahend is -0.  X = vB + t is
  //       -0 only if vB is -0, and a body velocity is never -0: every write of v / w is a sum or difference with the old value
  //       (x + y and x - y give -0 only from (-0, -0) / (-0, +0)), a product with a positive factor, or +0 (reset, sleep), and
  //       every world step adds h * (+0) to v.x (tests/test_oracle_physics.py::test_velocities_never_carry_a_negative_zero).
  // So the wall terms are dropped from the 180 sweeps (a third of a contact's arithmetic, 4 of its 22 constant words) whenever
  // every lane that sweeps contact k has a wall on side A - always in scenes without body-body pairs, wave by wave otherwise.
  template <bool WALLA>
  __device__ __forceinline__ void warmStartContactT(int k, RContact& c_) {
    const int pA = c_.pA, pB = c_.pB;
    const SweepC q = loadSweepC<WALLA>(k, c_);
    BodyMass mAs = WALLA ? BodyMass{0.0f, 0.0f, V2(0.0f, 0.0f)} : M(pA), mBs = WALLA ? getMass(pB - 4) : M(pB);   // B is never a wall
    float mA = mAs.invMass, iA = mAs.invI, mB = mBs.invMass, iB = mBs.invI;
    BodyVel va_ = WALLA ? BodyVel{V2(0.0f, 0.0f), 0.0f} : V(pA), vb_ = WALLA ? getVel(pB - 4) : V(pB);
    Vec2 vA = va_.v, vB = vb_.v;
    float wA = va_.w, wB = vb_.w;
    Vec2 normal = q.normal;
    Vec2 tangent = Cross(normal, 1.0f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j >= c_.pointCount) break;
      RPoint* vcp = c_.points + j;
      Vec2 P_ = vcp->normalImpulse * normal + vcp->tangentImpulse * tangent;
      if constexpr (!WALLA) {
        wA -= iA * Cross(q.rA[j], P_);
        vA -= mA * P_;
      }
      wB += iB * Cross(q.rB[j], P_);
      vB += mB * P_;
    }
    va_.v = vA; va_.w = wA; vb_.v = vB; vb_.w = wB;
    if constexpr (WALLA) {
      setVel(pB - 4, vb_);
    } else {
      setV(pA, va_);
      setV(pB, vb_);
    }
  }
  __device__ __forceinline__ void warmStartContact(int k, RContact& c_) {
    if (__ballot(c_.pA >= 4) == 0) warmStartContactT<true>(k, c_);
    else warmStartContactT<false>(k, c_);
  }

  // b2ContactSolver::SolveVelocityConstraints for one contact; returns true iff a non-zero impulse was applied
  // TRACK = false: the caller never looks at the result (jointed islands run all their sweeps, see velocitySweeps)
  template <bool WALLA, bool TRACK = true>
  __device__ __forceinline__ bool sweepContactT(int k, RContact& c_) {
    bool changed = false;
    const int pA = c_.pA, pB = c_.pB;
    const SweepC q = loadSweepC<WALLA>(k, c_);
    BodyMass mAs = WALLA ? BodyMass{0.0f, 0.0f, V2(0.0f, 0.0f)} : M(pA), mBs = WALLA ? getMass(pB - 4) : M(pB);   // B is never a wall
    float mA = mAs.invMass, iA = mAs.invI, mB = mBs.invMass, iB = mBs.invI;
    int pointCount = c_.pointCount;
    BodyVel va_ = WALLA ? BodyVel{V2(0.0f, 0.0f), 0.0f} : V(pA), vb_ = WALLA ? getVel(pB - 4) : V(pB);
    Vec2 vA = va_.v, vB = vb_.v;
    float wA = va_.w, wB = vb_.w;
    Vec2 normal = q.normal;
    Vec2 tangent = Cross(normal, 1.0f);
    float friction = q.friction;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j >= pointCount) break;
      RPoint* vcp = c_.points + j;
      Vec2 dv = vB + Cross(wB, q.rB[j]);
      if constexpr (!WALLA) dv = dv - vA - Cross(wA, q.rA[j]);
      float vt = Dot(dv, tangent) - 0.0f;
      float lambda = q.tangentMass[j] * (-vt);
      float maxFriction = friction * vcp->normalImpulse;
      float newImpulse = Clamp(vcp->tangentImpulse + lambda, -maxFriction, maxFriction);
      lambda = newImpulse - vcp->tangentImpulse;
      vcp->tangentImpulse = newImpulse;
      if constexpr (TRACK) changed = changed || (lambda != 0.0f);
      Vec2 P_ = lambda * tangent;
      if constexpr (!WALLA) {
        vA -= mA * P_;
        wA -= iA * Cross(q.rA[j], P_);
      }
      vB += mB * P_;
      wB += iB * Cross(q.rB[j], P_);
    }
    if (pointCount == 1) {
      RPoint* vcp = c_.points + 0;
      Vec2 dv = vB + Cross(wB, q.rB[0]);
      if constexpr (!WALLA) dv = dv - vA - Cross(wA, q.rA[0]);
      float vn = Dot(dv, normal);
      float lambda = -q.normalMass[0] * (vn - q.velocityBias[0]);
      float newImpulse = Max(vcp->normalImpulse + lambda, 0.0f);
      lambda = newImpulse - vcp->normalImpulse;
      vcp->normalImpulse = newImpulse;
      if constexpr (TRACK) changed = changed || (lambda != 0.0f);
      Vec2 P_ = lambda * normal;
      if constexpr (!WALLA) {
        vA -= mA * P_;
        wA -= iA * Cross(q.rA[0], P_);
      }
      vB += mB * P_;
      wB += iB * Cross(q.rB[0], P_);
    } else {
      RPoint* cp1 = c_.points + 0;
      RPoint* cp2 = c_.points + 1;
      Vec2 a_ = V2(cp1->normalImpulse, cp2->normalImpulse);
      Vec2 dv1 = vB + Cross(wB, q.rB[0]);
      if constexpr (!WALLA) dv1 = dv1 - vA - Cross(wA, q.rA[0]);
      Vec2 dv2 = vB + Cross(wB, q.rB[1]);
      if constexpr (!WALLA) dv2 = dv2 - vA - Cross(wA, q.rA[1]);
      float vn1 = Dot(dv1, normal);
      float vn2 = Dot(dv2, normal);
      Vec2 b;
      b.x = vn1 - q.velocityBias[0];
      b.y = vn2 - q.velocityBias[1];
      b -= Mul(q.K, a_);
      Vec2 x;
      bool solved = false;
      x = -Mul(q.nm, b);
      if (x.x >= 0.0f && x.y >= 0.0f) solved = true;
      if (!solved) {
        x.x = -q.normalMass[0] * b.x;
        x.y = 0.0f;
        vn2 = q.K.ex.y * x.x + b.y;
        if (x.x >= 0.0f && vn2 >= 0.0f) solved = true;
      }
      if (!solved) {
        x.x = 0.0f;
        x.y = -q.normalMass[1] * b.y;
        vn1 = q.K.ey.x * x.y + b.x;
        if (x.y >= 0.0f && vn1 >= 0.0f) solved = true;
      }
      if (!solved) {
        x.x = 0.0f;
        x.y = 0.0f;
        vn1 = b.x;
        vn2 = b.y;
        if (vn1 >= 0.0f && vn2 >= 0.0f) solved = true;
      }
      if (solved) {
        Vec2 d = x - a_;
        if constexpr (TRACK) changed = changed || (d.x != 0.0f) || (d.y != 0.0f);
        Vec2 P1 = d.x * normal;
        Vec2 P2 = d.y * normal;
        if constexpr (!WALLA) {
          vA -= mA * (P1 + P2);
          wA -= iA * (Cross(q.rA[0], P1) + Cross(q.rA[1], P2));
        }
        vB += mB * (P1 + P2);
        wB += iB * (Cross(q.rB[0], P1) + Cross(q.rB[1], P2));
        cp1->normalImpulse = x.x;
        cp2->normalImpulse = x.y;
      }
    }
    va_.v = vA; va_.w = wA; vb_.v = vB; vb_.w = wB;
    if constexpr (WALLA) {
      setVel(pB - 4, vb_);
    } else {
      setV(pA, va_);
      setV(pB, vb_);
    }
    return changed;
  }
  template <bool TRACK = true>
  __device__ __forceinline__ bool sweepContact(int k, RContact& c_) {
    if constexpr (NB == 1) return sweepContactT<true, TRACK>(k, c_);
    if (__ballot(c_.pA >= 4) == 0) return sweepContactT<true, TRACK>(k, c_);   // wave-uniform: every sweeping lane has a wall on side A
    return sweepContactT<false, TRACK>(k, c_);
  }

  // b2PositionSolverManifold + one b2ContactSolver::SolvePositionConstraints pass over one contact; returns min separation
  // WALLA (see warmStartContactT): the wall row (c = 0, a = 0, no mass) is a constant; K = mA + mB + iA rnA^2 + iB rnB^2 loses its
  // two zero terms exactly (0 + mB = mB, mB + (+-0) = mB), the wall's own updates are +0 - (+-0) = +0.  The identity rotation is
  // still multiplied through (1 * x - 0 * y), so every value - signs of zeros included - is the generic path's.
  template <bool TOI, bool WALLA>   // TOI: b2ContactSolver::SolveTOIPositionConstraints (b2_toiBaugarte; only the TOI body has mass)
  __device__ __forceinline__ float positionContactT(const RContact& c_, float minSeparation) {
    const int pA = c_.pA, pB = c_.pB;
    BodyMass mAs = WALLA ? BodyMass{0.0f, 0.0f, V2(0.0f, 0.0f)} : M(pA), mBs = WALLA ? getMass(pB - 4) : M(pB);
    float mA = mAs.invMass, iA = mAs.invI, mB = mBs.invMass, iB = mBs.invI;
    Vec2 localCenterA = mAs.lc, localCenterB = mBs.lc;
    float radiusA = c_.radiusA, radiusB = c_.radiusB;
    BodyPos pa_ = WALLA ? BodyPos{V2(0.0f, 0.0f), 0.0f} : P(pA), pb_ = WALLA ? getPos(pB - 4) : P(pB);
    Vec2 cA = pa_.c, cB = pb_.c;
    float aA = pa_.a, aB = pb_.a;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j >= c_.mcount) break;
      Transform xfA, xfB;
      if constexpr (WALLA) {
        xfA.q.s = 0.0f;
        xfA.q.c = 1.0f;
        xfA.p = V2(0.0f, 0.0f);
      } else {
        xfA.q = rotDead(deadQ, c_.pA, aA);
        xfA.p = cA - Mul(xfA.q, localCenterA);
      }
      xfB.q = rotDead(deadQ, c_.pB, aB);
      xfB.p = cB - Mul(xfB.q, localCenterB);
      Vec2 normal, point;
      float separation;
      Vec2 lpj = j == 0 ? c_.lp0 : c_.lp1;
      if (c_.mtype == kManifoldCircles) {
        Vec2 pointA = Mul(xfA, c_.localPoint);
        Vec2 pointB = Mul(xfB, c_.lp0);
        normal = pointB - pointA;
        Normalize(normal);
        point = 0.5f * (pointA + pointB);
        separation = Dot(pointB - pointA, normal) - radiusA - radiusB;
      } else if (c_.mtype == kManifoldFaceA) {
        normal = Mul(xfA.q, c_.localNormal);
        Vec2 planePoint = Mul(xfA, c_.localPoint);
        Vec2 clipPoint = Mul(xfB, lpj);
        separation = Dot(clipPoint - planePoint, normal) - radiusA - radiusB;
        point = clipPoint;
      } else {
        normal = Mul(xfB.q, c_.localNormal);
        Vec2 planePoint = Mul(xfB, c_.localPoint);
        Vec2 clipPoint = Mul(xfA, lpj);
        separation = Dot(clipPoint - planePoint, normal) - radiusA - radiusB;
        point = clipPoint;
        normal = -normal;
      }
      Vec2 rA = point - cA;
      Vec2 rB = point - cB;
      minSeparation = Min(minSeparation, separation);
      float C = Clamp((TOI ? kToiBaumgarte : kBaumgarte) * (separation + kLinearSlop), -kMaxLinearCorrection, 0.0f);
      float rnA = Cross(rA, normal);
      float rnB = Cross(rB, normal);
      float K = WALLA ? mB + iB * rnB * rnB : mA + mB + iA * rnA * rnA + iB * rnB * rnB;
      float impulse = K > 0.0f ? -C / K : 0.0f;
      Vec2 P_ = impulse * normal;
      if constexpr (!WALLA) {
        cA -= mA * P_;
        aA -= iA * Cross(rA, P_);
      }
      cB += mB * P_;
      aB += iB * Cross(rB, P_);
    }
    pa_.c = cA; pa_.a = aA; pb_.c = cB; pb_.a = aB;
    if constexpr (WALLA) {
      setPos(pB - 4, pb_);
    } else {
      setP(pA, pa_);
      setP(pB, pb_);
    }
    return minSeparation;
  }
  template <bool TOI = false>
  __device__ __forceinline__ float positionContact(const RContact& c_, float minSeparation) {
    if constexpr (NB == 1) return positionContactT<TOI, true>(c_, minSeparation);   // a one-body island: every contact is (wall, the body)
    if (__ballot(c_.pA >= 4) == 0) return positionContactT<TOI, true>(c_, minSeparation);
    return positionContactT<TOI, false>(c_, minSeparation);
  }

  // ---- b2RevoluteJoint ----
  __device__ __forceinline__ void initJoint(RJoint& J, float dtRatio) {
    BodyMass mAs = getMass(J.A), mBs = getMass(J.B);
    BodyPos pa_ = getPos(J.A), pb_ = getPos(J.B);
    BodyVel va_ = getVel(J.A), vb_ = getVel(J.B);
    float aA = pa_.a, aB = pb_.a;
    Vec2 vA = va_.v, vB = vb_.v;
    float wA = va_.w, wB = vb_.w;
    Rot qA = MakeRot(aA), qB = MakeRot(aB);
    J.rA = Mul(qA, J.anchorA - mAs.lc);
    J.rB = Mul(qB, J.anchorB - mBs.lc);
    Vec2 rA = J.rA, rB = J.rB;
    float mA = mAs.invMass, mB = mBs.invMass, iA = mAs.invI, iB = mBs.invI;
    bool fixedRotation = (iA + iB == 0.0f);
    Mat33& Mx = J.mass;
    Mx.ex.x = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
    Mx.ey.x = -rA.y * rA.x * iA - rB.y * rB.x * iB;
    Mx.ez.x = -rA.y * iA - rB.y * iB;
    Mx.ex.y = Mx.ey.x;
    Mx.ey.y = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
    Mx.ez.y = rA.x * iA + rB.x * iB;
    Mx.ex.z = Mx.ez.x;
    Mx.ey.z = Mx.ez.y;
    Mx.ez.z = iA + iB;
    {
      J.cyz = Cross(Mx.ey, Mx.ez);
      float det = Dot(Mx.ex, J.cyz);
      if (det != 0.0f) det = 1.0f / det;
      J.det33 = det;
      const float a11 = Mx.ex.x, a12 = Mx.ey.x, a21 = Mx.ex.y, a22 = Mx.ey.y;
      float d2 = a11 * a22 - a12 * a21;
      if (d2 != 0.0f) d2 = 1.0f / d2;
      J.det22 = d2;
    }
    float motorMass = iA + iB;
    if (motorMass > 0.0f) motorMass = 1.0f / motorMass;
    J.motorMass = motorMass;
    if (fixedRotation) J.motor = 0.0f;
    if (J.enableLimit && fixedRotation == false) {
      float jointAngle = aB - aA - J.ref;
      if (Abs(J.upper - J.lower) < 2.0f * kAngularSlop) {
        J.limit = 3;
      } else if (jointAngle <= J.lower) {
        if (J.limit != 1) J.imp.z = 0.0f;
        J.limit = 1;
      } else if (jointAngle >= J.upper) {
        if (J.limit != 2) J.imp.z = 0.0f;
        J.limit = 2;
      } else {
        J.limit = 0;
        J.imp.z = 0.0f;
      }
    } else {
      J.limit = 0;
    }
    // warm starting is always on in b2World::Step
    J.imp *= dtRatio;
    J.motor *= dtRatio;
    Vec2 P_ = V2(J.imp.x, J.imp.y);
    vA -= mA * P_;
    wA -= iA * (Cross(rA, P_) + J.motor + J.imp.z);
    vB += mB * P_;
    wB += iB * (Cross(rB, P_) + J.motor + J.imp.z);
    va_.v = vA; va_.w = wA; vb_.v = vB; vb_.w = wB;
    setVel(J.A, va_);
    setVel(J.B, vb_);
  }

  template <bool TRACK = true>
  __device__ __forceinline__ bool sweepJoint(RJoint& J, float dt) {
    bool changed = false;
    BodyMass mAs = getMass(J.A), mBs = getMass(J.B);
    BodyVel va_ = getVel(J.A), vb_ = getVel(J.B);
    Vec2 vA = va_.v, vB = vb_.v;
    float wA = va_.w, wB = vb_.w;
    float mA = mAs.invMass, mB = mBs.invMass, iA = mAs.invI, iB = mBs.invI;
    Vec2 rA = J.rA, rB = J.rB;
    const Mat33& Mx = J.mass;
    bool fixedRotation = (iA + iB == 0.0f);
    int limitState = J.limit;
    if (limitState != 3 && fixedRotation == false) {
      float Cdot = wB - wA - J.speed;
      float impulse = -J.motorMass * Cdot;
      float oldImpulse = J.motor;
      float maxImpulse = dt * J.maxMotorTorque;
      J.motor = Clamp(J.motor + impulse, -maxImpulse, maxImpulse);
      impulse = J.motor - oldImpulse;
      if constexpr (TRACK) changed = changed || (impulse != 0.0f);
      wA -= iA * impulse;
      wB += iB * impulse;
    }
    {
      // Limit and point constraint of b2RevoluteJoint::SolveVelocityConstraints as ONE predicated path.  Upstream has two
      // branches: limit active -> 3x3 solve with the z impulse clamped (on violation: "reduced" 2x2 solve with
      // rhs = -Cdot1 + impulse.z * ez.xy, z impulse reset to -accumulated); limit inactive -> plain 2x2 solve of -Cdot.
      // The second branch IS the clamp case of the first with an accumulated z impulse of zero (it is zero whenever the limit
      // is inactive): same Solve22, P applied the same way, and `+ (-0.0f)` on the angular term changes no value.  So lanes
      // without an active limit take the clamp case (with rhs = -Cdot1 exactly), and only lanes with one run Solve33.
      const bool lim = J.enableLimit && limitState != 0 && fixedRotation == false;
      const Vec2 Cdot1 = vB + Cross(wB, rB) - vA - Cross(wA, rA);
      Vec3 acc = J.imp;
      Vec3 impulse = Vec3{0.0f, 0.0f, 0.0f};
      if (lim) {
        const float Cdot2 = wB - wA;
        const Vec3 b3 = Vec3{Cdot1.x, Cdot1.y, Cdot2};
        Vec3 x;                                   // b2Mat33::Solve33 with the hoisted Cross(ey, ez) and 1/det
        x.x = J.det33 * Dot(b3, J.cyz);
        x.y = J.det33 * Dot(Mx.ex, Cross(b3, Mx.ez));
        x.z = J.det33 * Dot(Mx.ex, Cross(Mx.ey, b3));
        impulse = -x;
      }
      const float newImpulse = acc.z + impulse.z;
      const bool clampCase = !lim || (limitState == 1 && newImpulse < 0.0f) || (limitState == 2 && newImpulse > 0.0f);
      Vec2 rhs = -Cdot1;
      if (lim) rhs = -Cdot1 + acc.z * V2(Mx.ez.x, Mx.ez.y);
      Vec2 reduced;                               // b2Mat33::Solve22 with the hoisted 1/det
      reduced.x = J.det22 * (Mx.ey.y * rhs.x - Mx.ey.x * rhs.y);
      reduced.y = J.det22 * (Mx.ex.x * rhs.y - Mx.ex.y * rhs.x);
      Vec3 accPlain = acc;
      accPlain += impulse;
      const Vec3 impClamp = Vec3{reduced.x, reduced.y, lim ? -acc.z : -0.0f};
      const Vec3 accClamp = Vec3{acc.x + reduced.x, acc.y + reduced.y, lim ? 0.0f : acc.z};
      impulse.x = clampCase ? impClamp.x : impulse.x;
      impulse.y = clampCase ? impClamp.y : impulse.y;
      impulse.z = clampCase ? impClamp.z : impulse.z;
      acc.x = clampCase ? accClamp.x : accPlain.x;
      acc.y = clampCase ? accClamp.y : accPlain.y;
      acc.z = clampCase ? accClamp.z : accPlain.z;
      if constexpr (TRACK)
        changed = changed || (impulse.x != 0.0f) || (impulse.y != 0.0f) || (impulse.z != 0.0f) || (acc.x != J.imp.x) ||
                  (acc.y != J.imp.y) || (acc.z != J.imp.z);
      J.imp = acc;
      Vec2 P_ = V2(impulse.x, impulse.y);
      vA -= mA * P_;
      wA -= iA * (Cross(rA, P_) + impulse.z);
      vB += mB * P_;
      wB += iB * (Cross(rB, P_) + impulse.z);
    }
    va_.v = vA; va_.w = wA; vb_.v = vB; vb_.w = wB;
    setVel(J.A, va_);
    setVel(J.B, vb_);
    return changed;
  }

  __device__ __forceinline__ bool positionJoint(const RJoint& J) {
    BodyMass mAs = getMass(J.A), mBs = getMass(J.B);
    BodyPos pa_ = getPos(J.A), pb_ = getPos(J.B);
    Vec2 cA = pa_.c, cB = pb_.c;
    float aA = pa_.a, aB = pb_.a;
    float mA = mAs.invMass, mB = mBs.invMass, iA = mAs.invI, iB = mBs.invI;
    float angularError = 0.0f;
    float positionError = 0.0f;
    bool fixedRotation = (iA + iB == 0.0f);
    int limitState = J.limit;
    if (J.enableLimit && limitState != 0 && fixedRotation == false) {
      float angle = aB - aA - J.ref;
      float limitImpulse = 0.0f;
      if (limitState == 3) {
        float C = Clamp(angle - J.lower, -kMaxAngularCorrection, kMaxAngularCorrection);
        limitImpulse = -J.motorMass * C;
        angularError = Abs(C);
      } else if (limitState == 1) {
        float C = angle - J.lower;
        angularError = -C;
        C = Clamp(C + kAngularSlop, -kMaxAngularCorrection, 0.0f);
        limitImpulse = -J.motorMass * C;
      } else if (limitState == 2) {
        float C = angle - J.upper;
        angularError = C;
        C = Clamp(C - kAngularSlop, 0.0f, kMaxAngularCorrection);
        limitImpulse = -J.motorMass * C;
      }
      aA -= iA * limitImpulse;
      aB += iB * limitImpulse;
    }
    {
      Rot qA = MakeRot(aA), qB = MakeRot(aB);
      Vec2 rA = Mul(qA, J.anchorA - mAs.lc);
      Vec2 rB = Mul(qB, J.anchorB - mBs.lc);
      Vec2 C = cB + rB - cA - rA;
      positionError = Length(C);
      Mat22 K;
      K.ex.x = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
      K.ex.y = -iA * rA.x * rA.y - iB * rB.x * rB.y;
      K.ey.x = K.ex.y;
      K.ey.y = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
      Vec2 impulse = -K.Solve(C);
      cA -= mA * impulse;
      aA -= iA * Cross(rA, impulse);
      cB += mB * impulse;
      aB += iB * Cross(rB, impulse);
    }
    pa_.c = cA; pa_.a = aA; pb_.c = cB; pb_.a = aB;
    setPos(J.A, pa_);
    setPos(J.B, pb_);
    return positionError <= kLinearSlop && angularError <= kAngularSlop;
  }

  // state row for the short-cycle detector (joint-free islands): all body velocities + contact impulses
  static constexpr int kCycP = NB <= 3 ? 4 : 2;
  static constexpr int kCycW = 3 * NB + 4 * NCR;
#ifndef BLCD_CYC_REF_LDS
#define BLCD_CYC_REF_LDS 1
#endif
#ifndef BLCD_TOI1_REF_LDS
#define BLCD_TOI1_REF_LDS 0   // the one-body TOI mini-island (Env::toiIslandReg) of a class keeps its reference row in LDS too (set per class by the build)
#endif
  static constexpr bool kCycRefLds = BLCD_CYC_REF_LDS && (NB == 2 || (NB == 1 && BLCD_TOI1_REF_LDS));   // Brent's reference row in LDS (velocitySweeps); the three-body class loses 5 % with it (Object3-100k 2.44e7 -> 2.31e7)
  static __device__ __forceinline__ float* cycRefLds() {
    __shared__ float blk[kCycRefLds ? kCycW * 64 : 1];
    return blk;
  }
  struct CycRow {
    float v[kCycW];
  };
  struct CycDig {
    uint32_t d;
  };
  __device__ __forceinline__ void cycPack(CycRow& r) const {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const BodyVel bv = getVel(i);
      r.v[3 * i] = bv.v.x;
      r.v[3 * i + 1] = bv.v.y;
      r.v[3 * i + 2] = bv.w;
    }
#pragma unroll
    for (int k = 0; k < NCR; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bool live = k < nc && j < ct[k].pointCount;
        r.v[3 * NB + 4 * k + 2 * j] = live ? ct[k].points[j].normalImpulse : 0.0f;
        r.v[3 * NB + 4 * k + 2 * j + 1] = live ? ct[k].points[j].tangentImpulse : 0.0f;
      }
  }
  __device__ __forceinline__ void cycUnpack(const CycRow& r) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      BodyVel bv;
      bv.v.x = r.v[3 * i];
      bv.v.y = r.v[3 * i + 1];
      bv.w = r.v[3 * i + 2];
      setVel(i, bv);
    }
#pragma unroll
    for (int k = 0; k < NCR; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (k < nc && j < ct[k].pointCount) {
          ct[k].points[j].normalImpulse = r.v[3 * NB + 4 * k + 2 * j];
          ct[k].points[j].tangentImpulse = r.v[3 * NB + 4 * k + 2 * j + 1];
        }
  }

  // all velocity sweeps, with the bit-safe early exits of the generic path (fixed point; short cycle for joint-free islands)
  // startIt / yieldAt / yieldMaxLanes: environment-level scheduling (Env::islandSolve): a joint-free island that is still
  // sweeping after `yieldAt` sweeps while at most yieldMaxLanes lanes of the wave are, stops there and sets *yielded
  __device__ __forceinline__ int velocitySweeps(int velIters, float dt, unsigned long long* waveIters = nullptr, int startIt = 0,
                                                int yieldAt = 0, int yieldMaxLanes = 0, bool* yielded = nullptr) {
    int done = 0;
#ifndef BLCD_NO_UNTRACKED_SWEEPS
    if constexpr (NJR > 0) {
      // Jointed islands practically never reach a fixed point (oracle statistics: none in 2 400 Urchin solves), and the exits
      // only skip sweeps that would change nothing: running all of them gives the same bits.  When every sweeping lane of
      // the wave holds a jointed island the bookkeeping of the exits (a compare + scalar or per applied impulse, ~110
      // instructions per sweep of a 3-joint + 4-contact island) is dropped.
      if (startIt == 0 && yieldAt == 0 && __ballot(nj == 0) == 0) {
        for (int it = 0; it < velIters; ++it) {
#pragma unroll
          for (int k = 0; k < NJR; ++k)
            if (k < nj) sweepJoint<false>(jt[k], dt);
#pragma unroll
          for (int k = 0; k < NCR; ++k)
            if (k < nc) sweepContact<false>(k, ct[k]);
        }
        return velIters;
      }
    }
#endif
    const bool watch = nj == 0 && nc > 0;
#ifndef BLCD_CYC_WINDOW
    constexpr int kCycWatch = BLCD_CYC_WATCH;
#if BLCD_CYC_REF_LDS
    // The reference row lives in the wave's LDS ([word][lane]) for the two-body classes: it is written after sweeps 1, 2, 4,
    // .. and read only when a digest matches, but as kCycW (22) registers that stay live across the whole watched loop it
    // was what the 256-register build of the two-body class spilled - 44 scratch instructions in every one of the first 48
    // sweeps (kept ISA, tools/spill_loops.py).  The main island and the TOI mini-island never sweep at the same time: one block.
    float* const refL = kCycRefLds ? cycRefLds() + threadIdx.x : nullptr;
#endif
    CycRow ref;
    uint32_t refDig = 0;
    int refIt = -1, last = velIters - 1;
    bool cycling = false;
#else
    CycRow cyc[kCycP];
    CycDig cycDig[kCycP];
#endif
    for (int it = startIt; it < velIters; ++it) {
      bool changed = false;
#ifdef BLCD_PROF_TOI2
      if (waveIters && (int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) *waveIters += 1;   // wave-level iterations
#endif
      if constexpr (NJR > 0) {
#pragma unroll
        for (int k = 0; k < NJR; ++k)
          if (k < nj) changed = sweepJoint(jt[k], dt) || changed;
      }
#pragma unroll
      for (int k = 0; k < NCR; ++k)
        if (k < nc) changed = sweepContact(k, ct[k]) || changed;
      ++done;
      if (!changed) break;
#ifndef BLCD_CYC_WINDOW
      // Short-cycle detection, Brent's way: ONE reference row (taken after sweeps 1, 2, 4, 8, 16, 32), a digest of every sweep's
      // state compared with the reference's, the rows compared word by word only when the digests agree.  State(it) == state(ref)
      // makes the sequence periodic with period p = it - ref from ref on, so the state after the last sweep is the state
      // (velIters - 1 - it) mod p sweeps from here: the lane sweeps that many more times and stops.  Against the four-row window
      // this replaced: no ring of rows behind select chains (the window cost 28 instructions per state word and
      // sweep, then 7 with the digest; this costs 4), periods up to 16 and cycles that begin as late as sweep 32 are caught too;
      // a cycle is noticed up to twice as late, and its tail is swept instead of restored.  Which sweeps are skipped is the only
      // thing that changes: every exit still leaves exactly the state that all velIters sweeps would.
      if (it >= last) break;
      if (watch && !cycling && it < kCycWatch) {
        CycRow cur;
        cycPack(cur);
        uint32_t dig = 0;      // + 0.0f maps -0 to +0 like the float comparison does; 31 is odd, so no word's contribution is shifted out
#pragma unroll
        for (int q = 0; q < kCycW; ++q) dig = dig * 31u + __float_as_uint(cur.v[q] + 0.0f);
        if (refIt >= 0 && dig == refDig) {
          bool same = true;
#if BLCD_CYC_REF_LDS
          if constexpr (kCycRefLds) {
#pragma unroll
            for (int q = 0; q < kCycW; ++q) same = same && (refL[64 * q] == cur.v[q]);
          } else
#endif
          {
#pragma unroll
            for (int q = 0; q < kCycW; ++q) same = same && (ref.v[q] == cur.v[q]);
          }
          if (same) {
            cycling = true;
            last = it + (velIters - 1 - it) % (it - refIt);
            if (last == it) break;
          }
        }
        if (((it + 1) & it) == 0) {
#if BLCD_CYC_REF_LDS
          if constexpr (kCycRefLds) {
#pragma unroll
            for (int q = 0; q < kCycW; ++q) refL[64 * q] = cur.v[q];
          } else
#endif
          {
            ref = cur;
          }
          refDig = dig;
          refIt = it;
        }
      }
#ifndef BLCD_NO_UNTRACKED_TAIL
      if (it == kCycWatch - 1 && yieldAt == 0) {
        // Whoever is still here after the watch window either owes the tail of a detected cycle (`last`) or is a straggler that
        // will, with rare exceptions, run to the end: the rest is swept without the exits' bookkeeping (a lane that reaches a fixed
        // point on the way just sweeps on - the exits only ever skipped no-op sweeps).
        for (int it2 = it + 1; it2 <= last; ++it2) {
          if constexpr (NJR > 0) {
#pragma unroll
            for (int k = 0; k < NJR; ++k)
              if (k < nj) sweepJoint<false>(jt[k], dt);
          }
#pragma unroll
          for (int k = 0; k < NCR; ++k)
            if (k < nc) sweepContact<false>(k, ct[k]);
          ++done;
        }
        break;
      }
#endif
#else
      if (watch && it < 24) {
        CycRow cur;
        cycPack(cur);
        // A digest of the row filters the candidates: the stored rows are fetched (a select chain per word) and compared word by
        // word only where the digests agree, and the verdict is the word-by-word comparison's alone.  (+ 0.0f maps -0 to +0 like
        // the float comparison does; 31 is odd, so no word's contribution is shifted out.)
        uint32_t dig = 0;
#pragma unroll
        for (int q = 0; q < kCycW; ++q) dig = dig * 31u + __float_as_uint(cur.v[q] + 0.0f);
        bool found = false;
#pragma unroll
        for (int p = 1; p <= kCycP; ++p) {
          if (found || p > it) continue;
          if (rGet(cycDig, (it - p) & (kCycP - 1)).d != dig) continue;
          const CycRow old = rGet(cyc, (it - p) & (kCycP - 1));
          bool same = true;
#pragma unroll
          for (int q = 0; q < kCycW; ++q) same = same && (old.v[q] == cur.v[q]);
          if (same) {
            int r = (velIters - 1 - it) % p;
            if (r != 0) cycUnpack(rGet(cyc, (it - p + r) & (kCycP - 1)));
            found = true;
          }
        }
        if (found) break;
        rSet(cyc, it & (kCycP - 1), cur);
        rSet(cycDig, it & (kCycP - 1), CycDig{dig});
      }
#endif
      if (yieldAt > 0 && it == yieldAt - 1 && velIters > yieldAt) {
        if (__popcll(__ballot(1)) <= yieldMaxLanes) {
          *yielded = true;
          break;
        }
      }
    }
    return done;
  }

  // position iterations; returns positionSolved
  // startIt / yieldAt / yieldMaxLanes / yielded: as for velocitySweeps - an island still unsolved after `yieldAt` iterations
  // while at most yieldMaxLanes lanes of the wave are, stops there (the positions reached so far are the whole state needed)
  __device__ __forceinline__ bool positionIterations(int posIters, int* itersDone, int startIt = 0, int yieldAt = 0, int yieldMaxLanes = 0,
                                                     bool* yielded = nullptr) {
    bool positionSolved = false;
    int n = 0;
    for (int it = startIt; it < posIters; ++it) {
      float minSeparation = 0.0f;
#pragma unroll
      for (int k = 0; k < NCR; ++k)
        if (k < nc) minSeparation = positionContact(ct[k], minSeparation);
      ++n;
      bool contactsOkay = minSeparation >= -3.0f * kLinearSlop;
      bool jointsOkay = true;
      if constexpr (NJR > 0) {
#pragma unroll
        for (int k = 0; k < NJR; ++k)
          if (k < nj) {
            bool ok = positionJoint(jt[k]);
            jointsOkay = jointsOkay && ok;
          }
      }
      if (contactsOkay && jointsOkay) {
        positionSolved = true;
        break;
      }
      if (yieldAt > 0 && it == yieldAt - 1 && posIters > yieldAt) {
        if (__popcll(__ballot(1)) <= yieldMaxLanes) {
          *yielded = true;
          break;
        }
      }
    }
    *itersDone = n;
    return positionSolved;
  }
};

}  // namespace blcd
