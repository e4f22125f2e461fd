// blcd_math.h — float32 math layer of the HIP product (host+device), Box2D 2.3.x b2Math.h semantics.
// Reference path: the arithmetic under `b2World.Step` called at boxLCD/world_env.py:448-450 (Box2D is un-vendored).
// Compile with -ffp-contract=off: Box2D's x86-64 builds have no FMA contraction and parity is bit-exact.
// Version forms follow Box2D 2.3.0, the snapshot pybox2d 2.3.10 bundles (established by replaying the reference's recordings,
// DESIGN.md §2).  sincosf is our own restatement of glibc <= 2.27's algorithm, so that host setup code, device kernels and the
// parity oracle all agree bit for bit.
#define BLCD_HD __host__ __device__
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>

namespace blcd {

constexpr float kPi = 3.14159265359f;          // b2_pi
constexpr float kEpsilon = FLT_EPSILON;        // b2_epsilon
constexpr float kMaxFloat = FLT_MAX;           // b2_maxFloat
constexpr float kLinearSlop = 0.005f;
constexpr float kAngularSlop = 2.0f / 180.0f * kPi;
constexpr float kPolygonRadius = 2.0f * kLinearSlop;
constexpr float kAabbExtension = 0.1f;
constexpr float kAabbMultiplier = 2.0f;
constexpr int kMaxManifoldPoints = 2;
constexpr int kMaxPolygonVertices = 16;        // pybox2d builds Box2D with 16 (stock is 8): b2TimeOfImpact push-back bound
constexpr int kShapeVerts = 8;                 // storage bound of a polygon (BLCD_MAX_POLY_VERTS)
constexpr int kMaxSubSteps = 8;
constexpr int kMaxTOIContacts = 32;
constexpr float kVelocityThreshold = 1.0f;
constexpr float kMaxLinearCorrection = 0.2f;
constexpr float kMaxAngularCorrection = 8.0f / 180.0f * kPi;
constexpr float kMaxTranslation = 2.0f;
constexpr float kMaxTranslationSquared = kMaxTranslation * kMaxTranslation;
constexpr float kMaxRotation = 0.5f * kPi;
constexpr float kMaxRotationSquared = kMaxRotation * kMaxRotation;
constexpr float kBaumgarte = 0.2f;
constexpr float kToiBaumgarte = 0.75f;
constexpr float kTimeToSleep = 0.5f;
constexpr float kLinearSleepTolerance = 0.01f;
constexpr float kAngularSleepTolerance = 2.0f / 180.0f * kPi;

// ---------------------------------------------------------------------------------------------
// sincosf = sinf/cosf of glibc <= 2.27 (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c: the C form of the x86-64 assembly glibc
// used from 2.16): double-precision Chebyshev polynomials on |t| <= pi/4 after subtracting a multiple of pi/2.  This is the
// libm under the reference recordings that replay exactly from the recorder's inputs; results are correctly
// rounded for all but ~1e-7 of inputs.  Table-free and branch-light so the same source runs on host and device.
// Domain: |y| < 2^23 (body angles never leave it); beyond that both results are y - y.
// ---------------------------------------------------------------------------------------------
BLCD_HD static inline uint32_t asuint(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
#endif
}

BLCD_HD static inline double sc_sin_poly(double t, double t2) {   // t + t^3 (S0 + t^2 (S1 + t^2 (S2 + t^2 (S3 + t^2 S4))))
  double p = 0x1.71d7264e6b5b4p-19 + t2 * -0x1.a947e1674b58ap-26;
  p = -0x1.a019f8b4bd1f9p-13 + t2 * p;
  p = 0x1.1111110c2688bp-7 + t2 * p;
  p = -0x1.5555555551cd9p-3 + t2 * p;
  return t + t * t2 * p;
}
BLCD_HD static inline double sc_cos_poly(double t2) {             // 1 + t^2 (C0 + t^2 (C1 + t^2 (C2 + t^2 (C3 + t^2 C4))))
  double p = 0x1.a00eb9ac43ccp-16 + t2 * -0x1.23c97dd8844d7p-22;
  p = -0x1.6c16b348b6874p-10 + t2 * p;
  p = 0x1.55555545c50c7p-5 + t2 * p;
  p = -0x1.ffffffffe98aep-2 + t2 * p;
  return 1.0 + t2 * p;
}

// Out of line unless BLCD_SINCOS_INLINE: the circles-only step kernels need sincosf only on cold paths, and dozens of
// inlined copies were a third of a kernel that is already far larger than the instruction cache (measured: Bounce +6 %).
// Kernels that call it in hot loops (general classes: position solver, TOI of polygons) define BLCD_SINCOS_INLINE - out of
// line cost them 5-12 %.
#if defined(BLCD_SINCOS_INLINE)
#define BLCD_SINCOS_ATTR inline
#else
#define BLCD_SINCOS_ATTR __attribute__((noinline))
#endif
BLCD_HD static BLCD_SINCOS_ATTR void blcd_sincosf(float y, float* sinp, float* cosp) {
  // Same bits as the table-driven routine, with its argument ranges folded so that a wave whose lanes hold angles from
  // different ranges runs ONE polynomial pass instead of three:
  //  * [2^-5, pi/4) is the reduced path with n = 1, k = 0 (t = |y| - 0 = |y| exactly; the odd sine polynomial commutes with
  //    the sign, which is then applied from signbit(y) like everywhere else);
  //  * the two reductions ([pi/4, 9pi/4): one table constant; beyond: hi/lo parts) become a select.
  // Only |y| < 2^-5 (short polynomials / tiny arguments) and |y| >= 2^23 keep their own (rare) branches.
  const double kPio4 = 0x1.921fb54442d18p-1, kPio2 = 0x1.921fb54442d18p+0;
  const double theta = y;
  const double a = theta < 0.0 ? -theta : theta;
  if (a < 0x1p-5) {
    const double t2 = theta * theta;
    if (a >= 0x1p-27) {
      *sinp = (float)(theta + theta * t2 * (-0x1.555555543d49dp-3 + t2 * 0x1.110f475cec8c5p-7));
      *cosp = (float)(1.0 + t2 * (-0x1.fffffff5cc6fdp-2 + t2 * 0x1.55514b178dac5p-5));
    } else {
      *sinp = theta != 0.0 ? (float)(theta - theta * 0x1p-50) : y;
      *cosp = (float)(1.0 - a);
    }
    return;
  }
  if (!(a < 0x1p+23)) {
    *sinp = *cosp = y - y;
    return;
  }
  const unsigned n = (unsigned)(a * 0x1.45f306dc9c883p+0) + 1u;     // |y| * 4/pi, +1  (1 below pi/4)
  const double k = (double)(n >> 1);
  const double tNear = a - k * kPio2;                                            // pio2_table[n / 2]
  const double tFar = (a - k * 0x1.921fb544p+0) - k * 0x1.0b4611a626332p-34;     // PI_2_hi, PI_2_lo
  const double t = a < 9 * kPio4 ? tNear : tFar;
  const double t2 = t * t;
  const double sp = sc_sin_poly(t, t2), cp = sc_cos_poly(t2);
  // sinf: polynomial by (n & 2), sign by ((n >> 2) & 1) ^ signbit; cosf: the same with n + 2 and no signbit
  const unsigned m = n + 2u;
  const double sv = (n & 2u) ? cp : sp, cv = (m & 2u) ? cp : sp;
  const bool sneg = (((n >> 2) & 1u) != 0u) != (y < 0.0f), cneg = ((m >> 2) & 1u) != 0u;
  *sinp = (float)(sneg ? -sv : sv);
  *cosp = (float)(cneg ? -cv : cv);
}

// ---------------------------------------------------------------------------------------------
// b2Math.h restated
// ---------------------------------------------------------------------------------------------
// BLCD_VEC2_NATIVE makes Vec2 the compiler's own two-float vector: +, -, scalar * vector become element-wise built-ins (IEEE per
// element, nothing contracted - the GPU suite stays bit-equal) and on gfx950 v_pk_add_f32 / v_pk_mul_f32 with the cross products'
// swizzles in op_sel / neg modifiers: 14-18 % fewer VALU instructions in the solver loops of the (4,3,16) class.  Measured
// (round 3): no gain - Bounce-100k -1.5 %, Dropbox-100k -4.5 %, Object2-200k -12 %, Urchin-50k 0 %, LuxoBall-50k -1 % - because
// the even-aligned register pairs cost 30-50 more registers per kernel and more than twice the v_accvgpr traffic in the sweeps
// (100 -> 232 per sweep).  The struct stays the default.
#ifdef BLCD_VEC2_NATIVE
typedef float Vec2 __attribute__((ext_vector_type(2)));
#else
struct Vec2 {
  float x, y;
};
#endif
struct Vec3 {
  float x, y, z;
};
BLCD_HD static inline Vec2 V2(float x, float y) { return Vec2{x, y}; }
#ifndef BLCD_VEC2_NATIVE
BLCD_HD static inline Vec2 operator+(Vec2 a, Vec2 b) { return Vec2{a.x + b.x, a.y + b.y}; }
BLCD_HD static inline Vec2 operator-(Vec2 a, Vec2 b) { return Vec2{a.x - b.x, a.y - b.y}; }
BLCD_HD static inline Vec2 operator-(Vec2 a) { return Vec2{-a.x, -a.y}; }
BLCD_HD static inline Vec2 operator*(float s, Vec2 a) { return Vec2{s * a.x, s * a.y}; }
BLCD_HD static inline void operator+=(Vec2& a, Vec2 b) {
  a.x += b.x;
  a.y += b.y;
}
BLCD_HD static inline void operator-=(Vec2& a, Vec2 b) {
  a.x -= b.x;
  a.y -= b.y;
}
BLCD_HD static inline void operator*=(Vec2& a, float s) {
  a.x *= s;
  a.y *= s;
}
#endif
BLCD_HD static inline float Dot(Vec2 a, Vec2 b) { return a.x * b.x + a.y * b.y; }
BLCD_HD static inline float Cross(Vec2 a, Vec2 b) { return a.x * b.y - a.y * b.x; }
BLCD_HD static inline Vec2 Cross(Vec2 a, float s) { return Vec2{s * a.y, -s * a.x}; }
BLCD_HD static inline Vec2 Cross(float s, Vec2 a) { return Vec2{-s * a.y, s * a.x}; }
BLCD_HD static inline float LengthSquared(Vec2 a) { return a.x * a.x + a.y * a.y; }
BLCD_HD static inline float Length(Vec2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
BLCD_HD static inline float Normalize(Vec2& a) {
  float length = Length(a);
  if (length < kEpsilon) return 0.0f;
  float invLength = 1.0f / length;
  a.x *= invLength;
  a.y *= invLength;
  return length;
}
BLCD_HD static inline float Distance(Vec2 a, Vec2 b) { return Length(a - b); }
BLCD_HD static inline float DistanceSquared(Vec2 a, Vec2 b) {
  Vec2 c = a - b;
  return Dot(c, c);
}
BLCD_HD static inline float Min(float a, float b) { return a < b ? a : b; }
BLCD_HD static inline float Max(float a, float b) { return a > b ? a : b; }
BLCD_HD static inline Vec2 Min(Vec2 a, Vec2 b) { return Vec2{Min(a.x, b.x), Min(a.y, b.y)}; }
BLCD_HD static inline Vec2 Max(Vec2 a, Vec2 b) { return Vec2{Max(a.x, b.x), Max(a.y, b.y)}; }
BLCD_HD static inline float Abs(float a) { return a > 0.0f ? a : -a; }
BLCD_HD static inline float Clamp(float a, float lo, float hi) { return Max(lo, Min(a, hi)); }

BLCD_HD static inline Vec3 operator-(Vec3 a) { return Vec3{-a.x, -a.y, -a.z}; }
BLCD_HD static inline void operator+=(Vec3& a, Vec3 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
}
BLCD_HD static inline void operator*=(Vec3& a, float s) {
  a.x *= s;
  a.y *= s;
  a.z *= s;
}
BLCD_HD static inline float Dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
BLCD_HD static inline Vec3 Cross(Vec3 a, Vec3 b) {
  return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct Rot {
  float s, c;
  BLCD_HD void Set(float angle) { blcd_sincosf(angle, &s, &c); }
};
BLCD_HD static inline Rot MakeRot(float angle) {
  Rot q;
  q.Set(angle);
  return q;
}
BLCD_HD static inline Vec2 Mul(Rot q, Vec2 v) { return Vec2{q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y}; }
BLCD_HD static inline Vec2 MulT(Rot q, Vec2 v) { return Vec2{q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y}; }
BLCD_HD static inline Rot MulT(Rot q, Rot r) {
  Rot qr;
  qr.s = q.c * r.s - q.s * r.c;
  qr.c = q.c * r.c + q.s * r.s;
  return qr;
}

struct Transform {
  Vec2 p;
  Rot q;
};
BLCD_HD static inline Vec2 Mul(const Transform& T, Vec2 v) {
  float x = (T.q.c * v.x - T.q.s * v.y) + T.p.x;
  float y = (T.q.s * v.x + T.q.c * v.y) + T.p.y;
  return Vec2{x, y};
}
BLCD_HD static inline Vec2 MulT(const Transform& T, Vec2 v) {
  float px = v.x - T.p.x;
  float py = v.y - T.p.y;
  float x = (T.q.c * px + T.q.s * py);
  float y = (-T.q.s * px + T.q.c * py);
  return Vec2{x, y};
}
BLCD_HD static inline Transform MulT(const Transform& A, const Transform& B) {
  Transform C;
  C.q = MulT(A.q, B.q);
  C.p = MulT(A.q, B.p - A.p);
  return C;
}

struct Mat22 {
  Vec2 ex, ey;
  BLCD_HD Mat22 GetInverse() const {
    float a = ex.x, b = ey.x, c = ex.y, d = ey.y;
    Mat22 B;
    float det = a * d - b * c;
    if (det != 0.0f) det = 1.0f / det;
    B.ex.x = det * d;
    B.ey.x = -det * b;
    B.ex.y = -det * c;
    B.ey.y = det * a;
    return B;
  }
  BLCD_HD Vec2 Solve(Vec2 b) const {
    float a11 = ex.x, a12 = ey.x, a21 = ex.y, a22 = ey.y;
    float det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    Vec2 x;
    x.x = det * (a22 * b.x - a12 * b.y);
    x.y = det * (a11 * b.y - a21 * b.x);
    return x;
  }
};
BLCD_HD static inline Vec2 Mul(const Mat22& A, Vec2 v) { return Vec2{A.ex.x * v.x + A.ey.x * v.y, A.ex.y * v.x + A.ey.y * v.y}; }

struct Mat33 {
  Vec3 ex, ey, ez;
  BLCD_HD Vec3 Solve33(Vec3 b) const {
    float det = Dot(ex, Cross(ey, ez));
    if (det != 0.0f) det = 1.0f / det;
    Vec3 x;
    x.x = det * Dot(b, Cross(ey, ez));
    x.y = det * Dot(ex, Cross(b, ez));
    x.z = det * Dot(ex, Cross(ey, b));
    return x;
  }
  BLCD_HD Vec2 Solve22(Vec2 b) const {
    float a11 = ex.x, a12 = ey.x, a21 = ex.y, a22 = ey.y;
    float det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    Vec2 x;
    x.x = det * (a22 * b.x - a12 * b.y);
    x.y = det * (a11 * b.y - a21 * b.x);
    return x;
  }
};

// see Env::rotFor
BLCD_HD static inline Rot rotDead(uint32_t deadMask, int p, float angle) {
  Rot r;
  if ((deadMask >> p) & 1u) {
    r.s = 0.0f;
    r.c = 1.0f;
  } else {
    r.Set(angle);
  }
  return r;
}

struct Sweep {
  Vec2 localCenter, c0, c;
  float a0, a, alpha0;
  BLCD_HD void GetTransform(Transform* xf, float beta) const {
    xf->p = (1.0f - beta) * c0 + beta * c;
    float angle = (1.0f - beta) * a0 + beta * a;
    xf->q.Set(angle);
    xf->p -= Mul(xf->q, localCenter);
  }
  BLCD_HD void Advance(float alpha) {
    float beta = (alpha - alpha0) / (1.0f - alpha0);
    c0 = (1.0f - beta) * c0 + beta * c;      // Box2D 2.3.0 form (>= 2.3.1 writes c0 += beta * (c - c0))
    a0 = (1.0f - beta) * a0 + beta * a;
    alpha0 = alpha;
  }
  BLCD_HD void Normalize() {
    float twoPi = 2.0f * kPi;
    float d = twoPi * floorf(a0 / twoPi);
    a0 -= d;
    a -= d;
  }
};

struct AABB {
  Vec2 lo, hi;
  BLCD_HD bool Contains(const AABB& b) const {
    bool r = true;
    r = r && lo.x <= b.lo.x;
    r = r && lo.y <= b.lo.y;
    r = r && b.hi.x <= hi.x;
    r = r && b.hi.y <= hi.y;
    return r;
  }
};
BLCD_HD static inline bool TestOverlap(const AABB& a, const AABB& b) {
  Vec2 d1 = b.lo - a.hi, d2 = a.lo - b.hi;
  if (d1.x > 0.0f || d1.y > 0.0f) return false;
  if (d2.x > 0.0f || d2.y > 0.0f) return false;
  return true;
}

}  // namespace blcd
