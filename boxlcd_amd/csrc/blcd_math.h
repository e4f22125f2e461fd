// blcd_math.h — float32 math layer of the HIP product (host+device), Box2D 2.3.x b2Math.h semantics.
// Reference path: the arithmetic under `b2World.Step` called at boxLCD/world_env.py:448-450 (Box2D is un-vendored).
// Compile with -ffp-contract=off: Box2D's x86-64 builds have no FMA contraction and parity is bit-exact.
// sincosf is our own double-precision-polynomial implementation (same published algorithm as glibc >= 2.28), so that
// host setup code, device kernels and the parity oracle all agree bit for bit.
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
// sincosf: double-precision polynomial evaluation after a 2/pi reduction (the published algorithm glibc >= 2.28 and
// the ARM optimized routines use).  Written table-free so the same source runs on host and device: the second
// coefficient table of that algorithm is the first with the cosine polynomial negated, and the quadrant sign
// multiplies an odd polynomial, so both reduce to exact sign flips of the results.
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
BLCD_HD static inline uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }

// sin and cos polynomials on the reduced argument (|x| <= pi/4), x2 = x*x
BLCD_HD static inline void sincos_poly(double x, double x2, float* sinv, float* cosv) {
  const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
               c4 = 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  double x3, x4, x5, x6, s, c, c1_, c2_, s1_;
  x4 = x2 * x2;
  x3 = x2 * x;
  c2_ = c3 + x2 * c4;
  s1_ = s2 + x2 * s3;
  c1_ = c0 + x2 * c1;
  x5 = x3 * x2;
  x6 = x4 * x2;
  s = x + x3 * s1;
  c = c1_ + x4 * c2;
  *sinv = (float)(s + x5 * s1_);
  *cosv = (float)(c + x6 * c2_);
}

// Out of line unless BLCD_SINCOS_INLINE: the circles-only step kernels need sincosf only on cold paths, and 58 inlined
// copies (~250 instructions each) were a third of a kernel that is already far larger than the instruction cache
// (measured: Bounce +6 %).  Kernels that call it in hot loops (general classes: position solver, TOI of polygons) define
// BLCD_SINCOS_INLINE - out of line costs them 5-12 %.
#if defined(BLCD_SINCOS_INLINE)
#define BLCD_SINCOS_ATTR inline
#else
#define BLCD_SINCOS_ATTR __attribute__((noinline))
#endif
BLCD_HD static BLCD_SINCOS_ATTR void blcd_sincosf(float y, float* sinp, float* cosp) {
  double x = y;
  if (abstop12(y) < 0x3f4u) {          // |y| < 0x1.921FB6p-1f's top-12 class
    if (abstop12(y) < 0x398u) {        // |y| < 2^-12
      *sinp = y;
      *cosp = 1.0f;
      return;
    }
    sincos_poly(x, x * x, sinp, cosp);
    return;
  }
  int n;
  int sign = 0;
  if (abstop12(y) < 0x42fu) {          // |y| < 120
    double r = x * 0x1.45F306DC9C883p+23;
    n = ((int32_t)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;
  } else if (abstop12(y) < 0x7f8u) {   // finite
    const uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
                                   0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                   0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
                                   0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    uint32_t xi = asuint(y);
    sign = (int)(xi >> 31);
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
    int shift = (xi >> 23) & 7;
    uint64_t nn, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = xi * arr[0];
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    nn = (res0 + (1ULL << 61)) >> 62;
    res0 -= nn << 62;
    x = (double)(int64_t)res0;
    n = (int)nn;
    x = x * 0x1.921FB54442D18p-62;
  } else {
    *sinp = *cosp = y - y;
    return;
  }
  float sv, cv;
  sincos_poly(x, x * x, &sv, &cv);
  int q = n + sign;                    // quadrant used for the signs
  // sign[q & 3] = {+,-,-,+} multiplies x (odd sine polynomial => flips sv; cosine polynomial even => unchanged)
  if (((q & 3) == 1) || ((q & 3) == 2)) sv = -sv;
  if (q & 2) cv = -cv;                 // second table = negated cosine polynomial
  if (n & 1) {                         // odd quadrant: swap
    *sinp = cv;
    *cosp = sv;
  } else {
    *sinp = sv;
    *cosp = cv;
  }
}

// ---------------------------------------------------------------------------------------------
// b2Math.h restated
// ---------------------------------------------------------------------------------------------
struct Vec2 {
  float x, y;
};
struct Vec3 {
  float x, y, z;
};
BLCD_HD static inline Vec2 V2(float x, float y) { return Vec2{x, y}; }
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
    c0 += beta * (c - c0);
    a0 += beta * (a - a0);
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
