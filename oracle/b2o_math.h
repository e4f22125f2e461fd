// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under boxlcd_amd/ may include, link or call this.
//
// CPU restatement of the float32 math layer of Box2D 2.3.x (b2Math.h / b2Math.cpp), which is the
// arithmetic underneath boxLCD's `b2World.Step` (reference call site: boxLCD/world_env.py:448-450;
// Box2D itself is an un-vendored dependency, `Box2D==2.3.10` in requirements.txt:17).
//
// Every expression keeps Box2D's operand order; the file must be compiled with -ffp-contract=off so that
// no a*b+c is fused (the pybox2d wheels are plain x86-64 SSE2 builds: no FMA).
//
// sinf/cosf: Box2D calls libm (b2Rot::Set).  To make CPU oracle == GPU product bit-for-bit we do not call
// libm here; `b2o_sincosf` restates glibc >= 2.28's published sincosf algorithm (double-precision minimax
// polynomials after a 2/pi range reduction — the ARM optimized-routines implementation glibc adopted).
// tests/test_oracle_math.py measures its agreement with this container's glibc.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <cstdio>
#include <cstdlib>

namespace b2o {

constexpr float kPi = 3.14159265359f;          // b2_pi
constexpr float kEpsilon = FLT_EPSILON;        // b2_epsilon
constexpr float kMaxFloat = FLT_MAX;           // b2_maxFloat
constexpr float kLinearSlop = 0.005f;
constexpr float kAngularSlop = 2.0f / 180.0f * kPi;
constexpr float kPolygonRadius = 2.0f * kLinearSlop;
constexpr float kAabbExtension = 0.1f;
constexpr float kAabbMultiplier = 2.0f;
constexpr int kMaxManifoldPoints = 2;
constexpr int kMaxPolygonVertices = 16;        // pybox2d builds Box2D with 16 (stock is 8)
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
// sincosf (glibc sysdeps/ieee754/flt-32/s_sincosf.{c,h}, sincosf_data.c) restated.
// ---------------------------------------------------------------------------------------------
struct SinCosTab {
  double sign[4];
  double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3;
};
static const SinCosTab kSinCosTab[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2,
     0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2,
     -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static const uint32_t kInvPio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
                                      0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                      0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
                                      0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};

static inline uint32_t asuint(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
static inline uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }

static inline void sincosf_poly(double x, double x2, const SinCosTab* p, int n, float* sinp, float* cosp) {
  double x3, x4, x5, x6, s, c, c1, c2, s1;
  x4 = x2 * x2;
  x3 = x2 * x;
  c2 = p->c3 + x2 * p->c4;
  s1 = p->s2 + x2 * p->s3;
  float* tmp = (n & 1 ? cosp : sinp);
  cosp = (n & 1 ? sinp : cosp);
  sinp = tmp;
  c1 = p->c0 + x2 * p->c1;
  x5 = x3 * x2;
  x6 = x4 * x2;
  s = x + x3 * p->s1;
  c = c1 + x4 * p->c2;
  *sinp = (float)(s + x5 * s1);
  *cosp = (float)(c + x6 * c2);
}


// The same routine as glibc >= 2.28 selects through ifunc on x86-64 CPUs with FMA (sysdeps/x86_64/fpu/multiarch: the C
// file compiled with -mfma -mavx2, GCC contracting every a + b * c): variant 3 of the sincos switch.
static inline void sincosf_poly_fma(double x, double x2, const SinCosTab* p, int n, float* sinp, float* cosp) {
  double x3, x4, x5, x6, s, c, c1, c2, s1;
  x4 = x2 * x2;
  x3 = x2 * x;
  c2 = std::fma(x2, p->c4, p->c3);
  s1 = std::fma(x2, p->s3, p->s2);
  float* tmp = (n & 1 ? cosp : sinp);
  cosp = (n & 1 ? sinp : cosp);
  sinp = tmp;
  c1 = std::fma(x2, p->c1, p->c0);
  x5 = x3 * x2;
  x6 = x4 * x2;
  s = std::fma(x3, p->s1, x);
  c = std::fma(x4, p->c2, c1);
  *sinp = (float)std::fma(x5, s1, s);
  *cosp = (float)std::fma(x6, c2, c);
}

// ---------------------------------------------------------------------------------------------
// sinf / cosf of glibc <= 2.27 (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, s_sincosf.h of 2.26/2.27 = the C form of the
// x86_64 assembly used since 2.16): double-precision Chebyshev polynomials after a pi/2 table reduction.  This is the
// libm of the Ubuntu-18.04-era stack the reference's recordings were made with (replay table).
// ---------------------------------------------------------------------------------------------
namespace g227 {
static const double C0 = -0x1.ffffffffe98aep-2, C1 = 0x1.55555545c50c7p-5, C2 = -0x1.6c16b348b6874p-10,
                    C3 = 0x1.a00eb9ac43ccp-16, C4 = -0x1.23c97dd8844d7p-22;
static const double S0 = -0x1.5555555551cd9p-3, S1 = 0x1.1111110c2688bp-7, S2 = -0x1.a019f8b4bd1f9p-13,
                    S3 = 0x1.71d7264e6b5b4p-19, S4 = -0x1.a947e1674b58ap-26;
static const double SS0 = -0x1.555555543d49dp-3, SS1 = 0x1.110f475cec8c5p-7;
static const double CC0 = -0x1.fffffff5cc6fdp-2, CC1 = 0x1.55514b178dac5p-5;
static const double PI_2_hi = 0x1.921fb544p+0, PI_2_lo = 0x1.0b4611a626332p-34;
static const double SMALL = 0x1p-50, inv_PI_4 = 0x1.45f306dc9c883p+0;
static const double PIO4 = 0x1.921fb54442d18p-1;   // M_PI_4
static const double PIO2 = 0x1.921fb54442d18p+0;   // M_PI_2
static inline double pio2_table(unsigned k) { return (double)k * PIO2; }   // {0,1,2,3,4,5} * M_PI_2, each product exact-rounded as in the table
static inline double sin_poly(double t, double t2) {
  double cx = S3 + t2 * S4;
  cx = S2 + t2 * cx;
  cx = S1 + t2 * cx;
  cx = S0 + t2 * cx;
  return t + t * t2 * cx;
}
static inline double cos_poly(double t2) {
  double cx = C3 + t2 * C4;
  cx = C2 + t2 * cx;
  cx = C1 + t2 * cx;
  cx = C0 + t2 * cx;
  return 1.0 + t2 * cx;
}
static inline float reduced_sin(double theta, unsigned n, unsigned signbit) {
  const double theta2 = theta * theta;
  double sign = (((n >> 2) & 1) ^ signbit) ? -1.0 : 1.0;
  double sx = (n & 2) == 0 ? sin_poly(theta, theta2) : cos_poly(theta2);
  return (float)(sign * sx);
}
static inline float reduced_cos(double theta, unsigned n) {
  const double theta2 = theta * theta;
  n += 2;
  double sign = ((n >> 2) & 1) ? -1.0 : 1.0;
  double cx = (n & 2) == 0 ? sin_poly(theta, theta2) : cos_poly(theta2);
  return (float)(sign * cx);
}
// shared range reduction for PI/4 <= |x| < 2^23; returns theta, sets n
static inline double reduce(double abstheta, unsigned* n_out) {
  if (abstheta < 9 * PIO4) {
    unsigned n = (unsigned)(abstheta * inv_PI_4) + 1;
    *n_out = n;
    return abstheta - pio2_table(n / 2);
  }
  unsigned n = ((unsigned)(abstheta * inv_PI_4)) + 1;
  double x = (double)(n / 2);
  *n_out = n;
  return (abstheta - x * PI_2_hi) - x * PI_2_lo;
}
}  // namespace g227

static inline float b2o_sinf_g227(float x) {
  using namespace g227;
  double theta = x, abstheta = theta < 0 ? -theta : theta;
  if (abstheta < PIO4) {
    if (abstheta >= 0x1p-5) return (float)sin_poly(theta, theta * theta);
    if (abstheta >= 0x1p-27) {
      const double theta2 = theta * theta;
      double cx = SS0 + theta2 * SS1;
      return (float)(theta + theta * theta2 * cx);
    }
    return theta != 0.0 ? (float)(theta - theta * SMALL) : x;
  }
  if (!(abstheta < 0x1p+23)) return x - x;   // outside the restated domain (body angles never reach it)
  unsigned n;
  double t = reduce(abstheta, &n);
  return reduced_sin(t, n, x < 0.0f ? 1u : 0u);
}
static inline float b2o_cosf_g227(float x) {
  using namespace g227;
  double theta = x, abstheta = theta < 0 ? -theta : theta;
  if (abstheta < PIO4) {
    if (abstheta >= 0x1p-5) return (float)cos_poly(theta * theta);
    if (abstheta >= 0x1p-27) {
      const double theta2 = theta * theta;
      double cx = CC0 + theta2 * CC1;
      return (float)(1.0 + theta2 * cx);
    }
    return (float)(1.0 - abstheta);
  }
  if (!(abstheta < 0x1p+23)) return x - x;
  unsigned n;
  double t = reduce(abstheta, &n);
  return reduced_cos(t, n);
}

// ---------------------------------------------------------------------------------------------
// Variant switches.  The DEFAULTS are the set that reproduces the reference's recordings from the recorder's own inputs
// (tests/test_oracle_replay.py, DESIGN.md "replay table"): pybox2d 2.3.10 bundles Box2D 2.3.0, run on a glibc <= 2.27 libm.
// The alternatives are kept so that the table in DESIGN.md can be regenerated (tools/replay_gifs.py --table).
//   0 sincos      2 = glibc<=2.27 (default)  0 = glibc>=2.28   3 = glibc>=2.28 as built for FMA CPUs   1 = (float)sin((double)x)
//   1 damping     1 = Box2D 2.3.0 `v *= clamp(1 - h*c, 0, 1)` (default)   0 = >=2.3.1 Pade `v *= 1/(1 + h*c)`
//   2 advance     1 = Box2D 2.3.0 `c0 = (1-beta)*c0 + beta*c` (default)   0 = >=2.3.1 `c0 += beta*(c - c0)`
//   3 polygons    1 = Box2D 2.3.0 b2FindMaxSeparation hill climb + 0.98/0.001 face rule (default)   0 = >=2.3.1 brute force + k_tol
//   4 polygon mass reference point   0 = vertex mean (default)   1 = origin   2 = Box2D 2.2.1 formula
// ---------------------------------------------------------------------------------------------
enum { kVarSinCos = 0, kVarDamping = 1, kVarAdvance = 2, kVarPolygons = 3, kVarMassRef = 4, kNumVariants = 8 };
inline int g_variant[kNumVariants] = {2, 1, 1, 1, 0, 0, 0, 0};
static inline int b2o_variant(int k) { return g_variant[k]; }
static inline void b2o_sincosf(float y, float* sinp, float* cosp) {
  if (b2o_variant(0) == 1) { *sinp = (float)sin((double)y); *cosp = (float)cos((double)y); return; }
  if (b2o_variant(0) == 2) { *sinp = b2o_sinf_g227(y); *cosp = b2o_cosf_g227(y); return; }
  const bool fmaPoly = b2o_variant(0) == 3;
  auto poly = [&](double px, double px2, const SinCosTab* pp, int pn, float* ps, float* pc) {
    if (fmaPoly) sincosf_poly_fma(px, px2, pp, pn, ps, pc);
    else sincosf_poly(px, px2, pp, pn, ps, pc);
  };
  double x = y;
  double s;
  int n;
  const SinCosTab* p = &kSinCosTab[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    double x2 = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) {
      *sinp = y;
      *cosp = 1.0f;
      return;
    }
    poly(x, x2, p, 0, sinp, cosp);
  } else if (abstop12(y) < abstop12(120.0f)) {
    double r = x * p->hpi_inv;
    n = ((int32_t)r + 0x800000) >> 24;
    x = fmaPoly ? std::fma(-(double)n, p->hpi, x) : x - n * p->hpi;
    s = p->sign[n & 3];
    if (n & 2) p = &kSinCosTab[1];
    poly(x * s, x * x, p, n, sinp, cosp);
  } else if (abstop12(y) < abstop12(INFINITY)) {
    uint32_t xi = asuint(y);
    int sign = xi >> 31;
    const uint32_t* arr = &kInvPio4[(xi >> 26) & 15];
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
    s = p->sign[(n + sign) & 3];
    if ((n + sign) & 2) p = &kSinCosTab[1];
    poly(x * s, x * x, p, n, sinp, cosp);
  } else {
    *sinp = *cosp = y - y;
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
static inline Vec2 V2(float x, float y) { return Vec2{x, y}; }
static inline Vec2 operator+(Vec2 a, Vec2 b) { return Vec2{a.x + b.x, a.y + b.y}; }
static inline Vec2 operator-(Vec2 a, Vec2 b) { return Vec2{a.x - b.x, a.y - b.y}; }
static inline Vec2 operator-(Vec2 a) { return Vec2{-a.x, -a.y}; }
static inline Vec2 operator*(float s, Vec2 a) { return Vec2{s * a.x, s * a.y}; }
static inline void operator+=(Vec2& a, Vec2 b) {
  a.x += b.x;
  a.y += b.y;
}
static inline void operator-=(Vec2& a, Vec2 b) {
  a.x -= b.x;
  a.y -= b.y;
}
static inline void operator*=(Vec2& a, float s) {
  a.x *= s;
  a.y *= s;
}
static inline float Dot(Vec2 a, Vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float Cross(Vec2 a, Vec2 b) { return a.x * b.y - a.y * b.x; }
static inline Vec2 Cross(Vec2 a, float s) { return Vec2{s * a.y, -s * a.x}; }
static inline Vec2 Cross(float s, Vec2 a) { return Vec2{-s * a.y, s * a.x}; }
static inline float LengthSquared(Vec2 a) { return a.x * a.x + a.y * a.y; }
static inline float Length(Vec2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
static inline float Normalize(Vec2& a) {
  float length = Length(a);
  if (length < kEpsilon) return 0.0f;
  float invLength = 1.0f / length;
  a.x *= invLength;
  a.y *= invLength;
  return length;
}
static inline float Distance(Vec2 a, Vec2 b) { return Length(a - b); }
static inline float DistanceSquared(Vec2 a, Vec2 b) {
  Vec2 c = a - b;
  return Dot(c, c);
}
static inline float Min(float a, float b) { return a < b ? a : b; }
static inline float Max(float a, float b) { return a > b ? a : b; }
static inline Vec2 Min(Vec2 a, Vec2 b) { return Vec2{Min(a.x, b.x), Min(a.y, b.y)}; }
static inline Vec2 Max(Vec2 a, Vec2 b) { return Vec2{Max(a.x, b.x), Max(a.y, b.y)}; }
static inline float Abs(float a) { return a > 0.0f ? a : -a; }
static inline float Clamp(float a, float lo, float hi) { return Max(lo, Min(a, hi)); }

static inline Vec3 operator-(Vec3 a) { return Vec3{-a.x, -a.y, -a.z}; }
static inline void operator+=(Vec3& a, Vec3 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
}
static inline void operator*=(Vec3& a, float s) {
  a.x *= s;
  a.y *= s;
  a.z *= s;
}
static inline float Dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline Vec3 Cross(Vec3 a, Vec3 b) {
  return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct Rot {
  float s, c;
  void Set(float angle) { b2o_sincosf(angle, &s, &c); }
};
static inline Rot MakeRot(float angle) {
  Rot q;
  q.Set(angle);
  return q;
}
static inline Vec2 Mul(Rot q, Vec2 v) { return Vec2{q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y}; }
static inline Vec2 MulT(Rot q, Vec2 v) { return Vec2{q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y}; }
static inline Rot MulT(Rot q, Rot r) {
  Rot qr;
  qr.s = q.c * r.s - q.s * r.c;
  qr.c = q.c * r.c + q.s * r.s;
  return qr;
}

struct Transform {
  Vec2 p;
  Rot q;
};
static inline Vec2 Mul(const Transform& T, Vec2 v) {
  float x = (T.q.c * v.x - T.q.s * v.y) + T.p.x;
  float y = (T.q.s * v.x + T.q.c * v.y) + T.p.y;
  return Vec2{x, y};
}
static inline Vec2 MulT(const Transform& T, Vec2 v) {
  float px = v.x - T.p.x;
  float py = v.y - T.p.y;
  float x = (T.q.c * px + T.q.s * py);
  float y = (-T.q.s * px + T.q.c * py);
  return Vec2{x, y};
}
static inline Transform MulT(const Transform& A, const Transform& B) {
  Transform C;
  C.q = MulT(A.q, B.q);
  C.p = MulT(A.q, B.p - A.p);
  return C;
}

struct Mat22 {
  Vec2 ex, ey;
  Mat22 GetInverse() const {
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
  Vec2 Solve(Vec2 b) const {
    float a11 = ex.x, a12 = ey.x, a21 = ex.y, a22 = ey.y;
    float det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    Vec2 x;
    x.x = det * (a22 * b.x - a12 * b.y);
    x.y = det * (a11 * b.y - a21 * b.x);
    return x;
  }
};
static inline Vec2 Mul(const Mat22& A, Vec2 v) { return Vec2{A.ex.x * v.x + A.ey.x * v.y, A.ex.y * v.x + A.ey.y * v.y}; }

struct Mat33 {
  Vec3 ex, ey, ez;
  Vec3 Solve33(Vec3 b) const {
    float det = Dot(ex, Cross(ey, ez));
    if (det != 0.0f) det = 1.0f / det;
    Vec3 x;
    x.x = det * Dot(b, Cross(ey, ez));
    x.y = det * Dot(ex, Cross(b, ez));
    x.z = det * Dot(ex, Cross(ey, b));
    return x;
  }
  Vec2 Solve22(Vec2 b) const {
    float a11 = ex.x, a12 = ey.x, a21 = ex.y, a22 = ey.y;
    float det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    Vec2 x;
    x.x = det * (a22 * b.x - a12 * b.y);
    x.y = det * (a11 * b.y - a21 * b.x);
    return x;
  }
};

struct Sweep {
  Vec2 localCenter, c0, c;
  float a0, a, alpha0;
  void GetTransform(Transform* xf, float beta) const {
    xf->p = (1.0f - beta) * c0 + beta * c;
    float angle = (1.0f - beta) * a0 + beta * a;
    xf->q.Set(angle);
    xf->p -= Mul(xf->q, localCenter);
  }
  void Advance(float alpha) {
    float beta = (alpha - alpha0) / (1.0f - alpha0);
    if (b2o_variant(2) == 1) {  // Box2D 2.3.0: lerp form
      c0 = (1.0f - beta) * c0 + beta * c;
      a0 = (1.0f - beta) * a0 + beta * a;
    } else {                    // Box2D >= 2.3.1
      c0 += beta * (c - c0);
      a0 += beta * (a - a0);
    }
    alpha0 = alpha;
  }
  void Normalize() {
    float twoPi = 2.0f * kPi;
    float d = twoPi * floorf(a0 / twoPi);
    a0 -= d;
    a -= d;
  }
};

struct AABB {
  Vec2 lo, hi;
  bool Contains(const AABB& b) const {
    bool r = true;
    r = r && lo.x <= b.lo.x;
    r = r && lo.y <= b.lo.y;
    r = r && b.hi.x <= hi.x;
    r = r && b.hi.y <= hi.y;
    return r;
  }
};
static inline bool TestOverlap(const AABB& a, const AABB& b) {
  Vec2 d1 = b.lo - a.hi, d2 = a.lo - b.hi;
  if (d1.x > 0.0f || d1.y > 0.0f) return false;
  if (d2.x > 0.0f || d2.y > 0.0f) return false;
  return true;
}

}  // namespace b2o
