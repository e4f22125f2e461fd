// blcd_world.h — device-side world step of the HIP product: one thread advances one environment.
//
// Replaces `b2World.Step(dt, 180, 60)` as called 3x per env step at boxLCD/world_env.py:448-450 (Box2D 2.3.x
// semantics: b2World::Step / b2ContactManager::Collide / b2World::Solve / b2Island::Solve / b2ContactSolver /
// b2RevoluteJoint / b2World::SolveTOI / b2Island::SolveTOI; SURVEY.md §8 a3.*).
//
// Design (not Box2D's): boxLCD scenes are tiny and their topology is static, so instead of dynamic trees, lists and
// allocators an environment is a fixed set of *pair slots* — every (proxy A, proxy B) combination that Box2D's
// AddPair could ever accept for the scene (not both static, not joint-connected, category/mask filter), enumerated in
// (A,B)-sorted order on the host.  Box2D's order-defining structures collapse to:
//   * world contact list / body contact-edge lists  -> one newest-first array of slot ids (`wl`)
//   * broad-phase move buffer                        -> a bit mask of dynamic bodies
//   * islands                                        -> bit masks + a slot list in DFS order
// Walls (proxies 0..3) are constants: identity transform, zero velocity, zero inverse mass.
// State lives SoA in HBM ([field][env], env fastest => coalesced 256 B lines per wave) and is staged into
// per-thread storage for the duration of a launch.
#pragma once
#include <type_traits>
#include <utility>
#include "blcd_toi.h"
#include "blcd_toi_wall.h"
#include "blcd_collide_wall.h"
#include "blcd_island_reg.h"

#ifndef BLCD_REG_MAXNB
#define BLCD_REG_MAXNB 7   // largest scene class that uses the staged register island (blcd_island_reg.h); see DESIGN.md §4
#endif
#ifndef BLCD_REG_LDS
#define BLCD_REG_LDS 1     // classes with >= 4 bodies keep the staged island's body rows in LDS instead of select chains
#endif
#ifndef BLCD_GEN_LDS
#define BLCD_GEN_LDS 1     // largest class: the generic island's per-body arrays live in LDS
#endif
#ifndef BLCD_REG_CLDS
#define BLCD_REG_CLDS 1    // ... and (<= 5 bodies) the contacts' sweep-invariant constants too
#endif

namespace blcd {

constexpr int kMaxPairs = 100;
constexpr int kBodyFields = 18;   // c.xy a v.xy w c0.xy a0 xf.p.xy sleepTime awake fat.lo.xy fat.hi.xy sel
constexpr int kPairFields = 17;   // flags type|count ln.xy lp.xy {lp.xy ni ti id}x2
constexpr int kJointFields = 7;   // impulse.xyz motorImpulse limitState motorSpeed referenceAngle
constexpr int kWorldFields = 4;   // inv_dt0 moveMask flags nc   (+ wl packed 4 slots per word)

enum { PF_EXISTS = 1, PF_TOUCHING = 2, PF_ENABLED = 4, PF_ISLAND = 8, PF_TOI = 16, PF_TOISKIP = 32 };
enum { WF_NEWFIXTURE = 1 };
enum { FAULT_NAN = 1, FAULT_ELLIPSE = 2, FAULT_OVERFLOW = 4 };
enum { kInactiveLimit = 0, kAtLowerLimit = 1, kAtUpperLimit = 2, kEqualLimits = 3 };

struct DevVariant {
  int shape;
  float mass, invMass, I, invI;
  Vec2 localCenter;
};
struct DevBody {
  int nChoices;
  DevVariant var[2];
  float friction, restitution, linearDamping, angularDamping;
  uint32_t cat, mask;
  int nJoints;
  int joints[8];  // joint-edge list, newest first (b2Body::m_jointList order)
};
struct DevJoint {
  int bodyA, bodyB;  // dynamic-body indices
  Vec2 anchorA, anchorB;
  int enableLimit;
  float lower, upper, maxMotorTorque, speed;
  int actionIndex;
};
struct DevPair {
  int a, b;  // proxy ids, a < b; a < 4 => wall
  float friction, restitution;
};
struct DevObs {
  int kind, body;
  float lo, hi;
};
struct DevScene {
  int nb, nj, np, nobs, nact, lcdW, lcdH, rasterVariant;
  float worldW, worldH;
  Vec2 gravity;
  float dt;
  int substeps, velIters, posIters;
  int nShapes;
  int dbgSkip;  // only read by -DBLCD_ABLATION builds (BLCD_DEBUG_SKIP: 1 collide, 2 solve, 4 TOI, 8 everything; results are wrong when set)
  Shape wallShape[4];
  AABB wallFat[4];
  WallK wallK[4];                 // blcd_collide_wall.h: what the wall narrow phase needs of each wall (filled on the host)
  Vec2 wallNrm[4], wallTan[4];    // (edge x 1)/|edge| and edge/|edge|^2: the conservative TOI early-out
  Shape shapes[24];
  DevBody bodies[20];
  int bodyKind[20];   // 0 object, 1 robot root, 2 robot link: colours of the RGB render only
  DevJoint joints[20];
  DevPair pairs[kMaxPairs];
  DevObs obs[96];
};

// Scheduling words, appended after everything else (so no other offset moves): two progress words (env-step and sub-step
// reached inside the current rollout / chunk, what the environment is suspended at, bodies that were in an island this world
// step) and, per body, the velocity a suspended island's sweeps had reached (v.x v.y w).  See Env::solve.
BLCD_HD static inline int schedWordOffset(int nb, int nj, int np) {
  return nb * kBodyFields + np * kPairFields + nj * kJointFields + kWorldFields + (np + 3) / 4;
}
#ifdef BLCD_SCHED
BLCD_HD static inline int stateWords(int nb, int nj, int np) { return schedWordOffset(nb, nj, np) + 2 + 3 * nb; }
#else
// default build: the environment-level schedulers (built, bit-neutral, measured slower) are compiled out
// (BLCD_DEFS=-DBLCD_SCHED brings them back) and no slot carries their words
BLCD_HD static inline int stateWords(int nb, int nj, int np) { return schedWordOffset(nb, nj, np); }
#endif
// word 0: env-step reached (16 bits) | sub-step (2) << 16 | suspended at a TOI event << 18
// word 1: seeds of the islands suspended in their velocity sweeps (7 bits) | in their position iterations << 7 | islanded bodies << 14
constexpr uint32_t kProgPendingMask1 = 0x3fffu;
// A lane whose joint-free island has not converged after this many velocity sweeps (= the window of the short-cycle detector)
// may suspend its environment instead of dragging the wave through the remaining <= 156 sweeps (scheduling only: the
// environment resumes in a later pass of the same chunk at exactly this sweep, with exactly this solver state)
constexpr int kYieldSweeps = 24;
// ... and an island whose position constraints are still unsolved after this many of the <= 60 position iterations
// (mean 8; 7 % of the jointed islands never meet the tolerance and run all of them)
constexpr int kYieldPosIters = 12;

struct VCPoint {
  Vec2 rA, rB;
  float normalImpulse, tangentImpulse, normalMass, tangentMass, velocityBias;
};
struct VC {
  VCPoint points[2];
  Vec2 normal;
  Mat22 normalMass, K;
  int slot, pA, pB;  // proxy ids of fixture A / B
  float friction, restitution;
  int pointCount;
};

// Register residency for small scenes: a per-thread array that is indexed with a run-time value lives in scratch (private
// memory, ~500+ cycles per dependent access at one or two waves per SIMD).  For arrays of at most 4 elements the helpers
// below turn a run-time index into a compare/select chain over statically indexed elements, so the array can be
// scalarised into VGPRs; larger arrays fall back to plain indexing.
// (A loop / struct-assignment formulation gets re-rolled into an indexed scratch access by LLVM; selecting 32-bit words with
// compile-time unrolled fold expressions does not.)
template <typename T, int N, size_t... K>
__device__ __forceinline__ void selWordsGet(const T (&a)[N], int i, int t, uint32_t* r, std::index_sequence<K...>) {
  uint32_t w[sizeof...(K)];
  __builtin_memcpy(w, &a[t], sizeof(T));
  ((r[K] = (i == t) ? w[K] : r[K]), ...);
}
template <typename T, int N, size_t... Ts>
__device__ __forceinline__ T selGetImpl(const T (&a)[N], int i, std::index_sequence<Ts...>) {
  constexpr size_t W = sizeof(T) / 4;
  static_assert(sizeof(T) % 4 == 0, "selGet needs whole 32-bit words");
  uint32_t r[W];
  __builtin_memcpy(r, &a[0], sizeof(T));
  (selWordsGet(a, i, (int)(Ts + 1), r, std::make_index_sequence<W>{}), ...);
  T out;
  __builtin_memcpy(&out, r, sizeof(T));
  return out;
}
template <int N, typename T>
__device__ __forceinline__ T selGet(const T (&a)[N], int i) {
  if constexpr (N <= 4) return selGetImpl(a, i, std::make_index_sequence<N - 1>{});
  else return a[i];
}
template <typename T, int N, size_t... K>
__device__ __forceinline__ void selWordsSet(T (&a)[N], int i, int t, const uint32_t* v, std::index_sequence<K...>) {
  uint32_t w[sizeof...(K)];
  __builtin_memcpy(w, &a[t], sizeof(T));
  ((w[K] = (i == t) ? v[K] : w[K]), ...);
  __builtin_memcpy(&a[t], w, sizeof(T));
}
template <typename T, int N, size_t... Ts>
__device__ __forceinline__ void selSetImpl(T (&a)[N], int i, const T& val, std::index_sequence<Ts...>) {
  constexpr size_t W = sizeof(T) / 4;
  uint32_t v[W];
  __builtin_memcpy(v, &val, sizeof(T));
  (selWordsSet(a, i, (int)Ts, v, std::make_index_sequence<W>{}), ...);
}
template <int N, typename T>
__device__ __forceinline__ void selSet(T (&a)[N], int i, const T& v) {
  if constexpr (N <= 4) selSetImpl(a, i, v, std::make_index_sequence<N>{});
  else a[i] = v;
}
// Per-body state arrays of the small scene classes.  Most accesses have a compile-time index (unrolled loops over bodies), but the
// TOI phase, the contact update and the island bookkeeping reach a body through a run-time index (bi(proxy)), and ONE such access
// keeps the whole array - and with it every access - in scratch (~450 cycles per dependent load at one wave per SIMD).  For 2-4
// bodies `arr[i]` therefore goes through the compare/select chains above (a proxy object, so the call sites keep their array
// syntax; a constant index folds to a plain register); larger classes keep plain arrays.  Pure data movement.
template <typename T, int N>
struct SelArr {
  T e[N];
  struct Ref {
    T (&e)[N];
    int i;
    __device__ __forceinline__ operator T() const { return selGet(e, i); }
    __device__ __forceinline__ const Ref& operator=(const T& x) const {
      selSet(e, i, x);
      return *this;
    }
    __device__ __forceinline__ const Ref& operator=(const Ref& r) const {
      const T x = r;
      selSet(e, i, x);
      return *this;
    }
    template <typename U>
    __device__ __forceinline__ const Ref& operator+=(const U& x) const {
      T t = selGet(e, i);
      t += x;
      selSet(e, i, t);
      return *this;
    }
    template <typename U>
    __device__ __forceinline__ const Ref& operator-=(const U& x) const {
      T t = selGet(e, i);
      t -= x;
      selSet(e, i, t);
      return *this;
    }
    template <typename U>
    __device__ __forceinline__ const Ref& operator*=(const U& x) const {
      T t = selGet(e, i);
      t *= x;
      selSet(e, i, t);
      return *this;
    }
  };
  __device__ __forceinline__ Ref operator[](int i) { return Ref{e, i}; }
  __device__ __forceinline__ T operator[](int i) const { return selGet(e, i); }
};
template <typename T, int N>
struct PlainArr {
  T e[N];
  __device__ __forceinline__ T& operator[](int i) { return e[i]; }
  __device__ __forceinline__ const T& operator[](int i) const { return e[i]; }
};
#ifndef BLCD_BODY_SEL
#define BLCD_BODY_SEL 1
#endif
template <typename T, int N>
using BodyArr = std::conditional_t<(BLCD_BODY_SEL && N >= 2 && N <= 4), SelArr<T, N>, PlainArr<T, N>>;

// Small integers (slot ids, flag bytes, counters) indexed with run-time values.  Up to 4 entries share one register; up to 32
// are packed four to a word in (N+3)/4 registers and a run-time index becomes a compare/select over the WORDS plus a shift -
// a byte array indexed at run time would live in scratch, and the contact-list loops (collide, island DFS, the TOI scan) read
// these vectors several times per list position: at one wave per SIMD every such read was a ~500-cycle dependent scratch
// load (measured: 28 M of the 40 M cycles a wave of Urchins spent in the TOI phase per 20 env-steps were this scan).
template <int N>
struct ByteVec {
  static constexpr int kWords = N <= 32 ? (N + 3) / 4 : 1;
  uint32_t w;                        // N <= 4
  uint32_t ww[N > 4 && N <= 32 ? kWords : 1];
  uint8_t arr[N > 32 ? N : 1];
  template <size_t... K>
  __device__ __forceinline__ uint32_t wordAt(int wi, std::index_sequence<K...>) const {
    uint32_t x = ww[0];
    ((x = (wi == (int)(K + 1)) ? ww[K + 1] : x), ...);
    return x;
  }
  template <size_t... K>
  __device__ __forceinline__ void wordSet(int wi, uint32_t v, std::index_sequence<K...>) {
    ((ww[K] = (wi == (int)K) ? v : ww[K]), ...);
  }
  __device__ __forceinline__ void clear() {
    w = 0;
#pragma unroll
    for (int k = 0; k < (N > 4 && N <= 32 ? kWords : 1); ++k) ww[k] = 0;
    if constexpr (N > 32)
      for (int k = 0; k < N; ++k) arr[k] = 0;
  }
  __device__ __forceinline__ int get(int i) const {
    if constexpr (N <= 4) return (int)((w >> (8 * i)) & 0xffu);
    else if constexpr (N <= 32) return (int)((wordAt(i >> 2, std::make_index_sequence<kWords - 1>{}) >> (8 * (i & 3))) & 0xffu);
    else return arr[i];
  }
  __device__ __forceinline__ void set(int i, int v) {
    if constexpr (N <= 4) {
      w = (w & ~(0xffu << (8 * i))) | ((uint32_t)(v & 0xff) << (8 * i));
    } else if constexpr (N <= 32) {
      const int wi = i >> 2, sh = 8 * (i & 3);
      const uint32_t old = wordAt(wi, std::make_index_sequence<kWords - 1>{});
      wordSet(wi, (old & ~(0xffu << sh)) | ((uint32_t)(v & 0xff) << sh), std::make_index_sequence<kWords>{});
    } else {
      arr[i] = (uint8_t)v;
    }
  }
  __device__ __forceinline__ void orBits(int i, int bits) { set(i, get(i) | bits); }
  __device__ __forceinline__ void clearBits(int i, int bits) { set(i, get(i) & ~bits); }
  __device__ __forceinline__ void insertFront(int count, int v) {  // shift [0,count) up by one, put v at 0
    if constexpr (N <= 4) {
      w = (w << 8) | (uint32_t)(v & 0xff);
    } else if constexpr (N <= 32) {
#pragma unroll
      for (int k = kWords - 1; k > 0; --k) ww[k] = (ww[k] << 8) | (ww[k - 1] >> 24);
      ww[0] = (ww[0] << 8) | (uint32_t)(v & 0xff);
    } else {
      for (int k = count; k > 0; --k) arr[k] = arr[k - 1];
      arr[0] = (uint8_t)v;
    }
  }
};

// SH = shape set of the scene, known when the handle is created: 0 = anything, 1 = every dynamic body is a circle.
// Circles-only scenes (Bounce, Bounce2) get kernels without the polygon routines (edge-polygon / polygon-polygon collide,
// 2-point manifolds and the block solver, polygon TOI proxies, polygon raster): far fewer live registers.
// Per-body working arrays of the generic island (positions / velocities of the island's bodies, indexed with run-time body
// ids).  Small classes: plain per-thread arrays.  The largest class (17-20 bodies): columns of an LDS block [body][lane],
// so that a run-time index is an LDS address (~100 cycles) instead of a scratch access behind 25 KB per lane of private
// memory (the class's generic solver is bound by that latency).
// Lanes per wave of the largest class.  Its waves are bound by the latency of the scratch-resident generic solver (26 KB of
// private memory per lane, one dependent ~500-cycle access per constraint row), so it wants SEVERAL waves per SIMD to hide
// that latency - and its 210 registers allow two.  What stood in the way was the LDS block of the island's body columns
// (6 x NB x 64 words = 30 KB per wave: four waves per CU): with at most BLCD_BIG_LANES environments per wave the block is
// 6 x NB x 32 words = 15 KB, eight waves fit a CU (the host never launches this class with wider waves).
#ifndef BLCD_BIG_LANES
#define BLCD_BIG_LANES 32
#endif
template <typename T, int N, bool LDS, int LL = 64>
struct BodyCol {
  T a[N];
  __device__ __forceinline__ T get(int i) const { return a[i]; }
  __device__ __forceinline__ void set(int i, const T& v) { a[i] = v; }
};
typedef __attribute__((address_space(3))) float LdsFloat;
template <int N, int LL>
struct BodyCol<float, N, true, LL> {
  LdsFloat* p;   // this lane's element of row 0; rows are LL lanes apart
  __device__ __forceinline__ float get(int i) const { return p[i * LL]; }
  __device__ __forceinline__ void set(int i, float v) const { p[i * LL] = v; }
};
template <int N, int LL>
struct BodyCol<Vec2, N, true, LL> {
  LdsFloat* p;   // this lane's x of row 0; a row is LL lanes x (x, y)
  __device__ __forceinline__ Vec2 get(int i) const { return V2(p[i * 2 * LL], p[i * 2 * LL + 1]); }
  __device__ __forceinline__ void set(int i, const Vec2& v) const {
    p[i * 2 * LL] = v.x;
    p[i * 2 * LL + 1] = v.y;
  }
};

// SCHED: the environment-level scheduler's kernels (suspend / resume).  It is a compile-time switch because the
// tuned one-body kernels lose 8-29 % when the suspension paths are merely present (registers, loop shape).
template <int NB, int NJ, int NP, int SH = 0, bool SCHED = false>
struct Env {
  static constexpr bool kCirc = SH == 1;
  static constexpr int kMP = kCirc ? 1 : 2;  // manifold points a contact can have
  // Island contact capacity.  A single body in the walled arena can touch at most two (adjacent) walls at once, so the
  // one-body configuration keeps two constraint slots in registers; an island that would need more raises FAULT_OVERFLOW
  // (reported through blcd_get_faults) instead of silently dropping physics.
  static constexpr int kMaxC = NB == 1 ? (NP < 2 ? NP : 2) : NP;
  // One-body classes: side A of every contact is a wall (pairAOf) - its zero-mass, never-moving row is folded out of the hot
  // loops; the argument that this is bit-neutral is spelled out at RegIsland::warmStartContactT (blcd_island_reg.h).
  static constexpr bool kWallA = NB == 1;
  static constexpr int kU = kMaxC <= 4 ? kMaxC : 1;   // unroll factor of the constraint loops (full unroll => static vc[] indices)
  static constexpr int kUS = NP <= 4 ? NP : 1;        // unroll factor of the per-slot load/store loops
  const DevScene* S;
  // --- dynamic bodies ---
  BodyArr<Vec2, NB> c, v, c0, xfp;
  BodyArr<float, NB> a, w, a0, sleepTime, alpha0;
  BodyArr<Rot, NB> q;
  BodyArr<AABB, NB> fat;
  BodyArr<int, NB> sel;
  BodyArr<float, NB> rmaxV;   // largest distance of a shape vertex from the body's centre of mass (TOI early-out)
  BodyArr<int, NB> shapeIx;   // S->bodies[i].var[sel[i]].shape, read once per launch: shapeOf() is then ONE dependent global load, not two
  uint32_t awakeMask;
  BodyArr<float, NB> invMass, invI;
  BodyArr<Vec2, NB> lc;
  // --- pair slots ---
  ByteVec<NP> wl;        // world contact list, newest first (slot ids)
  int nc;
  ByteVec<NP> pflags;
  ByteVec<NP> toiCount;
  float toi[NP];
  // Contact manifolds (15 state words per pair slot), indexed with run-time slot ids.  Three homes:
  //  * one-body classes: the wave's LDS block [word][lane] (word = 15 * slot + field, 15 KB per wave): a slot id is an address - no
  //    48 / 64 v_cndmask select chain per read / write - and 64 fewer long-lived vector registers: the general one-body kernel
  //    drops from 512 to 414 registers and its velocity sweep from 286 to 238 instructions (80 -> 16 v_accvgpr_read: the loop's
  //    constants were being parked in AGPRs), the circles-only one from 363 to 212, i.e. no AGPR traffic at all
  //    (Dropbox-100k +10 %);
  //  * BLCD_MAN_LDS=2 adds the two-body classes (34.5 KB per wave; measured: Object2-200k -4 % - fifteen ds_read against four
  //    scratch_load_dwordx4 per manifold - so off by default);
  //  * otherwise: NP <= 4 registers behind select chains (selGet / selSet), larger classes plain indexing (scratch).
#ifndef BLCD_MAN_LDS
#define BLCD_MAN_LDS 1
#endif
  static constexpr bool kManLds = BLCD_MAN_LDS >= 1 && NB <= BLCD_MAN_LDS;
  static constexpr int kManWords = 15;
  Manifold man[kManLds ? 1 : NP];
  float* ML;   // kManLds: this lane's column of the manifold block
  static __device__ __forceinline__ float* manLdsBase() {
    __shared__ float blk[kManWords * NP * 64];
    return blk;
  }
  __device__ __forceinline__ Manifold manGet(int s) const {
    if constexpr (kManLds) {
      const float* p = ML + 64 * kManWords * s;
      Manifold m;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        m.points[k].localPoint = V2(p[64 * (5 * k)], p[64 * (5 * k + 1)]);
        m.points[k].normalImpulse = p[64 * (5 * k + 2)];
        m.points[k].tangentImpulse = p[64 * (5 * k + 3)];
        m.points[k].id.key = __float_as_uint(p[64 * (5 * k + 4)]);
      }
      m.localNormal = V2(p[64 * 10], p[64 * 11]);
      m.localPoint = V2(p[64 * 12], p[64 * 13]);
      const int tc = __float_as_int(p[64 * 14]);
      m.type = tc & 0xff;
      m.pointCount = tc >> 8;
      return m;
    } else {
      return selGet(man, s);
    }
  }
  __device__ __forceinline__ void manSet(int s, const Manifold& m) {
    if constexpr (kManLds) {
      float* p = ML + 64 * kManWords * s;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        p[64 * (5 * k)] = m.points[k].localPoint.x;
        p[64 * (5 * k + 1)] = m.points[k].localPoint.y;
        p[64 * (5 * k + 2)] = m.points[k].normalImpulse;
        p[64 * (5 * k + 3)] = m.points[k].tangentImpulse;
        p[64 * (5 * k + 4)] = __uint_as_float(m.points[k].id.key);
      }
      p[64 * 10] = m.localNormal.x;
      p[64 * 11] = m.localNormal.y;
      p[64 * 12] = m.localPoint.x;
      p[64 * 13] = m.localPoint.y;
      p[64 * 14] = __int_as_float((m.type & 0xff) | (m.pointCount << 8));
    } else {
      selSet(man, s, m);
    }
  }
  __device__ __forceinline__ void manClearCount(int s) {   // a new contact starts with no points; everything else stays as it was
    if constexpr (kManLds) {
      float* p = ML + 64 * kManWords * s;
      p[64 * 14] = __int_as_float(__float_as_int(p[64 * 14]) & 0xff);
    } else {
      man[s].pointCount = 0;
    }
  }
  // --- joints ---
  Vec3 jimp[NJ > 0 ? NJ : 1];
  float jmotor[NJ > 0 ? NJ : 1], jspeed[NJ > 0 ? NJ : 1], jref[NJ > 0 ? NJ : 1];
  int jlimit[NJ > 0 ? NJ : 1];
  // joint solver temporaries
  Vec2 jrA[NJ > 0 ? NJ : 1], jrB[NJ > 0 ? NJ : 1];
  Mat33 jmass[NJ > 0 ? NJ : 1];
  float jmotorMass[NJ > 0 ? NJ : 1];
  // --- world ---
  float inv_dt0;
  uint32_t moveMask;
  uint32_t wflags;
  int fault;
  float wallAlpha0[4];
  // Scene constants the hot loops ask for with per-lane indices, copied to registers once per launch: a per-lane-indexed read
  // of DevScene is a vector global load (hundreds of cycles at one wave per SIMD) in the middle of a dependent chain.
  Vec2 wallV0[4], wallV1[4];                 // the four wall edges (wave-uniform)
  float wallRad[4];
  Vec2 wallFatLo[4], wallFatHi[4];           // their fat AABBs
  Vec2 wallNrm[4], wallTan[4];               // (edge x 1)/|edge| and edge/|edge|^2: for the conservative TOI early-out only
  WallK wallK[4];                            // what the wall narrow phase needs of each wall, evaluated once per launch
  static constexpr bool kPairRegs = NP <= 4;   // ByteVec packs up to 4 entries into one register
  ByteVec<4> pairA_, pairB_;                 // pair table (proxy ids), one-body classes
  float crad[kCirc ? NB : 1];                // circles-only scenes: the bodies' shapes
  Vec2 cctr[kCirc ? NB : 1];
  // diagnostic per-wave cycle accounting (BLCD_WAVETIMES): 0 collide 1 solve 2 toi (all) 3 toi-event 4 #toi calls 5 #events | summed over lanes: 6 cycles inside the TOI routine, 7 wave-level executions of it
  unsigned long long prof[8];
  bool profOn;
  // --- island scratch ---
  static constexpr bool kGenLds = BLCD_GEN_LDS && NB > 7;
  static constexpr int kMaxLanes = NB > 7 ? BLCD_BIG_LANES : 64;   // environments per wave the kernel's LDS blocks are sized for
  BodyCol<Vec2, NB, kGenLds, kMaxLanes> pc, pv;
  BodyCol<float, NB, kGenLds, kMaxLanes> pa, pw;
  VC vc[kMaxC];
  ByteVec<NP> ic;
  // short-cycle detector for the velocity sweeps (islands without joints only)
  static constexpr int kCycP = 4;                       // longest period looked for
  static constexpr int kCycNB = NB < 3 ? NB : 3;        // eligible islands: at most this many dynamic bodies
  static constexpr int kCycNC = kMaxC < 6 ? kMaxC : 6;  //                   and this many contacts
  static constexpr int kCycW = 3 * kCycNB + 4 * kCycNC;
  static constexpr int kCycSweeps = 24;                 // stop looking after this many sweeps

  uint8_t ij[NJ > 0 ? NJ : 1];

  // --- environment-level scheduling (fused rollouts) ---
  static constexpr bool kCanYield = SCHED && NB <= 7;   // suspension points: velocity sweep 24 of a joint-free island, position iteration 12 of a staged island, the first TOI event
  float* gst;            // this slot's column of the state array (word f at gst[f * gN]): suspended velocities go straight there
  int gN;
  uint32_t velMask;      // seeds of the islands suspended in their velocity sweeps
  uint32_t posMask;      // seeds of the islands suspended in their position iterations
  bool toiPending;       // the world step is suspended at its first TOI event (SolveTOI restarts from scratch: nothing it did so far is kept)
  uint32_t islandedMask; // bodies that were in an island when the world step was suspended (SynchronizeFixtures still owed)
  int yieldMaxLanes;     // suspend only when at most this many lanes of the wave are still sweeping (0 = never)

  // ------------------------------------------------------------------------------------------------
  // SoA state <-> thread
  // ------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void load(const DevScene* scene, const float* __restrict__ st, int N, int e) {
    S = scene;
    if constexpr (kManLds) ML = manLdsBase() + threadIdx.x;
    else ML = nullptr;
    if constexpr (SCHED) {
      gst = const_cast<float*>(st) + e;
      gN = N;
    }
    velMask = 0;
    posMask = 0;
    toiPending = false;
    islandedMask = 0;
    yieldMaxLanes = 0;
    if constexpr (kGenLds) {   // 6 words x NB bodies x kMaxLanes lanes (15 KB for NB = 20, 32 lanes)
      __shared__ float blk[6 * NB * kMaxLanes];
      LdsFloat* base = (LdsFloat*)blk;
      pc.p = base + 2 * threadIdx.x;
      pv.p = base + 2 * NB * kMaxLanes + 2 * threadIdx.x;
      pa.p = base + 4 * NB * kMaxLanes + threadIdx.x;
      pw.p = base + 5 * NB * kMaxLanes + threadIdx.x;
    }
    const int nb = S->nb, nj = S->nj, np = S->np;
    awakeMask = 0;
    deadQ = 15u;
    wl.clear();
    pflags.clear();
    toiCount.clear();
    ic.clear();
#pragma unroll   // must unroll: a rolled loop indexes the member arrays at run time, which sends the whole Env to scratch
    for (int k = 0; k < 4; ++k) {
      wallV0[k] = S->wallShape[k].v[0];
      wallV1[k] = S->wallShape[k].v[1];
      wallRad[k] = S->wallShape[k].radius;
      wallNrm[k] = S->wallNrm[k];
      wallTan[k] = S->wallTan[k];
      wallFatLo[k] = S->wallFat[k].lo;
      wallFatHi[k] = S->wallFat[k].hi;
      wallK[k] = S->wallK[k];
    }
    pairA_.w = pairB_.w = 0;
    if (kPairRegs) {
#pragma unroll
      for (int s2 = 0; s2 < (kPairRegs ? NP : 0); ++s2) {
        pairA_.set(s2, s2 < np ? S->pairs[s2].a : 0);
        pairB_.set(s2, s2 < np ? S->pairs[s2].b : 0);
      }
    }
    for (int i = 0; i < NB; ++i) {
      if (i >= nb) break;
      const float* p = st + (size_t)(i * kBodyFields) * N + e;
      c[i] = V2(p[0], p[(size_t)1 * N]);
      a[i] = p[(size_t)2 * N];
      v[i] = V2(p[(size_t)3 * N], p[(size_t)4 * N]);
      w[i] = p[(size_t)5 * N];
      c0[i] = V2(p[(size_t)6 * N], p[(size_t)7 * N]);
      a0[i] = p[(size_t)8 * N];
      xfp[i] = V2(p[(size_t)9 * N], p[(size_t)10 * N]);
      sleepTime[i] = p[(size_t)11 * N];
      if (p[(size_t)12 * N] != 0.0f) awakeMask |= 1u << i;
      {
        AABB fb_;
        fb_.lo = V2(p[(size_t)13 * N], p[(size_t)14 * N]);
        fb_.hi = V2(p[(size_t)15 * N], p[(size_t)16 * N]);
        fat[i] = fb_;
      }
      sel[i] = __float_as_int(p[(size_t)17 * N]);
      alpha0[i] = 0.0f;
      const DevVariant& var = S->bodies[i].var[sel[i]];
      invMass[i] = var.invMass;
      invI[i] = var.invI;
      lc[i] = var.localCenter;
      shapeIx[i] = var.shape;
      const Shape& shp = S->shapes[var.shape];
      if constexpr (!kCirc) {
        float rm_ = 0.0f;
        const int nv_ = shp.type == kCircle ? 1 : shp.count;
        for (int k2 = 0; k2 < kShapeVerts; ++k2)
          if (k2 < nv_) rm_ = Max(rm_, Length(shp.v[k2] - var.localCenter));
        rmaxV[i] = rm_ * 1.0000005f;   // rounded up: the bound must not come out larger than with the exact radius
      } else {
        rmaxV[i] = 0.0f;
      }
      if (kCirc) {
        crad[i] = shp.radius;
        cctr[i] = shp.v[0];
      }
      if (shp.type == kCircle && shp.v[0].x == 0.0f && shp.v[0].y == 0.0f && var.localCenter.x == 0.0f && var.localCenter.y == 0.0f &&
          S->bodies[i].nJoints == 0)   // a joint anchor would make the rotation matter
        deadQ |= 16u << i;
      q[i] = rotFor(4 + i, a[i]);   // the stored rotation is itself only consumed by contact code (see rotFor)
    }
    const float* pp = st + (size_t)(nb * kBodyFields) * N + e;
#pragma unroll kUS
    for (int s = 0; s < NP; ++s) {
      if (s >= np) break;
      const float* p = pp + (size_t)(s * kPairFields) * N;
      int fl = __float_as_int(p[0]);
      pflags.set(s, fl);
      int tc = __float_as_int(p[(size_t)1 * N]);
      Manifold m;
      m.type = tc & 0xff;
      m.pointCount = tc >> 8;
      m.localNormal = V2(p[(size_t)2 * N], p[(size_t)3 * N]);
      m.localPoint = V2(p[(size_t)4 * N], p[(size_t)5 * N]);
      for (int k = 0; k < 2; ++k) {
        const float* r = p + (size_t)(6 + 5 * k) * N;
        m.points[k].localPoint = V2(r[0], r[(size_t)1 * N]);
        m.points[k].normalImpulse = r[(size_t)2 * N];
        m.points[k].tangentImpulse = r[(size_t)3 * N];
        m.points[k].id.key = __float_as_uint(r[(size_t)4 * N]);
      }
      if constexpr (kManLds) manSet(s, m);
      else man[s] = m;
      toiCount.set(s, 0);
      selSet(toi, s, (float)(1.0f));
    }
    const float* jp = pp + (size_t)(np * kPairFields) * N;
    for (int j = 0; j < NJ; ++j) {
      if (j >= nj) break;
      const float* p = jp + (size_t)(j * kJointFields) * N;
      jimp[j] = Vec3{p[0], p[(size_t)1 * N], p[(size_t)2 * N]};
      jmotor[j] = p[(size_t)3 * N];
      jlimit[j] = __float_as_int(p[(size_t)4 * N]);
      jspeed[j] = p[(size_t)5 * N];
      jref[j] = p[(size_t)6 * N];
    }
    const float* wp = jp + (size_t)(nj * kJointFields) * N;
    inv_dt0 = wp[0];
    moveMask = __float_as_uint(wp[(size_t)1 * N]);
    uint32_t fl = __float_as_uint(wp[(size_t)2 * N]);
    wflags = fl & 0xff;
    fault = (int)(fl >> 8);
    nc = __float_as_int(wp[(size_t)3 * N]);
    for (int k = 0; k < (NP + 3) / 4; ++k) {
      if (4 * k >= np) break;
      uint32_t word = __float_as_uint(wp[(size_t)(4 + k) * N]);
      for (int t = 0; t < 4; ++t)
        if (4 * k + t < NP) wl.set(4 * k + t, (int)((word >> (8 * t)) & 0xffu));
    }
    for (int k = 0; k < 4; ++k) wallAlpha0[k] = 0.0f;
    profOn = false;
    for (int k = 0; k < 8; ++k) prof[k] = 0;
  }

  __device__ __forceinline__ void store(float* __restrict__ st, int N, int e) {
    const int nb = S->nb, nj = S->nj, np = S->np;
    for (int i = 0; i < NB; ++i) {
      if (i >= nb) break;
      float* p = st + (size_t)(i * kBodyFields) * N + e;
      const Vec2 c_ = c[i], v_ = v[i], c0_ = c0[i], xfp_ = xfp[i];
      const AABB fat_ = fat[i];
      p[0] = c_.x;
      p[(size_t)1 * N] = c_.y;
      p[(size_t)2 * N] = a[i];
      p[(size_t)3 * N] = v_.x;
      p[(size_t)4 * N] = v_.y;
      p[(size_t)5 * N] = w[i];
      p[(size_t)6 * N] = c0_.x;
      p[(size_t)7 * N] = c0_.y;
      p[(size_t)8 * N] = a0[i];
      p[(size_t)9 * N] = xfp_.x;
      p[(size_t)10 * N] = xfp_.y;
      p[(size_t)11 * N] = sleepTime[i];
      p[(size_t)12 * N] = (awakeMask >> i) & 1 ? 1.0f : 0.0f;
      p[(size_t)13 * N] = fat_.lo.x;
      p[(size_t)14 * N] = fat_.lo.y;
      p[(size_t)15 * N] = fat_.hi.x;
      p[(size_t)16 * N] = fat_.hi.y;
      // sel is immutable during stepping
    }
    float* pp = st + (size_t)(nb * kBodyFields) * N + e;
#pragma unroll kUS
    for (int s = 0; s < NP; ++s) {
      if (s >= np) break;
      float* p = pp + (size_t)(s * kPairFields) * N;
      p[0] = __int_as_float((int)(pflags.get(s) & (PF_EXISTS | PF_TOUCHING | PF_ENABLED)));
      Manifold m;
      if constexpr (kManLds) m = manGet(s);
      else m = man[s];
      p[(size_t)1 * N] = __int_as_float((m.type & 0xff) | (m.pointCount << 8));
      p[(size_t)2 * N] = m.localNormal.x;
      p[(size_t)3 * N] = m.localNormal.y;
      p[(size_t)4 * N] = m.localPoint.x;
      p[(size_t)5 * N] = m.localPoint.y;
      for (int k = 0; k < 2; ++k) {
        float* r = p + (size_t)(6 + 5 * k) * N;
        r[0] = m.points[k].localPoint.x;
        r[(size_t)1 * N] = m.points[k].localPoint.y;
        r[(size_t)2 * N] = m.points[k].normalImpulse;
        r[(size_t)3 * N] = m.points[k].tangentImpulse;
        r[(size_t)4 * N] = __uint_as_float(m.points[k].id.key);
      }
    }
    float* jp = pp + (size_t)(np * kPairFields) * N;
    for (int j = 0; j < NJ; ++j) {
      if (j >= nj) break;
      float* p = jp + (size_t)(j * kJointFields) * N;
      p[0] = jimp[j].x;
      p[(size_t)1 * N] = jimp[j].y;
      p[(size_t)2 * N] = jimp[j].z;
      p[(size_t)3 * N] = jmotor[j];
      p[(size_t)4 * N] = __int_as_float(jlimit[j]);
      p[(size_t)5 * N] = jspeed[j];
      // jref immutable
    }
    float* wp = jp + (size_t)(nj * kJointFields) * N;
    wp[0] = inv_dt0;
    wp[(size_t)1 * N] = __uint_as_float(moveMask);
    wp[(size_t)2 * N] = __uint_as_float((wflags & 0xff) | ((uint32_t)fault << 8));
    wp[(size_t)3 * N] = __int_as_float(nc);
    for (int k = 0; k < (NP + 3) / 4; ++k) {
      if (4 * k >= np) break;
      uint32_t word = 0;
      for (int t = 0; t < 4; ++t)
        if (4 * k + t < NP) word |= (uint32_t)wl.get(4 * k + t) << (8 * t);
      wp[(size_t)(4 + k) * N] = __uint_as_float(word);
    }
  }

  // ------------------------------------------------------------------------------------------------
  // proxy accessors (proxy p: 0..3 wall, 4+i dynamic body i)
  // ------------------------------------------------------------------------------------------------
  static __device__ __forceinline__ int bi(int p) { return NB == 1 ? 0 : p - 4; }  // body index of a dynamic proxy
  __device__ __forceinline__ Transform xfOf(int p) const {
    Transform t;
    if (p < 4) {
      t.p = V2(0.0f, 0.0f);
      t.q.s = 0.0f;
      t.q.c = 1.0f;
    } else {
      t.p = xfp[bi(p)];
      t.q = q[bi(p)];
    }
    return t;
  }
  // Rotation used by the CONTACT code for proxy p at angle `angle`.  Walls sit at angle 0 (sincosf(0) == (0, 1) exactly).
  // For a circle centred on its body origin, with the centre of mass there too, every contact-side use of q multiplies the
  // zero vector (xf.p = c - q*localCenter; manifold local points are the circle centre), so sincosf is skipped: values are
  // unchanged, only the sign of an exact zero could differ.  This includes the body's stored rotation q[i]: it is not part
  // of the persistent state (observations and the raster recompute sin/cos from the angle) and is read by contact code only.
  uint32_t deadQ;
  __device__ __forceinline__ Rot rotFor(int p, float angle) const {
    if (kCirc) {  // circles-only classes are picked only for origin-centred circles without joints: every bit is set
      Rot r;
      r.s = 0.0f;
      r.c = 1.0f;
      return r;
    }
    return rotDead(deadQ, p, angle);
  }
  // One-body classes: every pair slot is (wall, the body) - static-static pairs do not exist - so B is proxy 4 and A < 4.
  __device__ __forceinline__ int pairAOf(int s) const {
    int a_ = kPairRegs ? pairA_.get(s) : S->pairs[s].a;
    if constexpr (NB == 1) a_ &= 3;
    return a_;
  }
  __device__ __forceinline__ int pairBOf(int s) const {
    if constexpr (NB == 1) return 4;
    return kPairRegs ? pairB_.get(s) : S->pairs[s].b;
  }
  __device__ __forceinline__ Shape circShapeReg(int i) const {  // kCirc only
    Shape c_{};
    c_.type = kCircle;
    c_.count = 1;
    c_.radius = crad[NB == 1 ? 0 : i];
    c_.v[0] = cctr[NB == 1 ? 0 : i];
    return c_;
  }
  __device__ __forceinline__ float radiusOf(int p) const {
    if (p < 4) return selGet(wallRad, p);
    if (kCirc) return circShapeReg(bi(p)).radius;
    return shapeOf(p)->radius;
  }
  __device__ __forceinline__ int typeOf(int p) const { return p < 4 ? (int)kEdge : (kCirc ? (int)kCircle : (int)shapeOf(p)->type); }
  __device__ __forceinline__ const Shape* shapeOf(int p) const {
    if (p < 4) return &S->wallShape[p];
    return &S->shapes[shapeIx[bi(p)]];
  }
  __device__ __forceinline__ AABB fatOf(int p) const {
    // value selects on purpose: `cond ? global : member` would become a select of POINTERS, which makes the member's
    // address escape as a flat pointer and keeps the whole Env object in scratch
    AABB w;
    w.lo = selGet(wallFatLo, p < 4 ? p : 0);
    w.hi = selGet(wallFatHi, p < 4 ? p : 0);
    const AABB d = fat[p < 4 ? 0 : bi(p)];
    AABB r;
    r.lo.x = p < 4 ? w.lo.x : d.lo.x;
    r.lo.y = p < 4 ? w.lo.y : d.lo.y;
    r.hi.x = p < 4 ? w.hi.x : d.hi.x;
    r.hi.y = p < 4 ? w.hi.y : d.hi.y;
    return r;
  }
  __device__ __forceinline__ bool awakeDyn(int i) const { return (awakeMask >> i) & 1; }
  __device__ __forceinline__ void wake(int p) {  // b2Body::SetAwake(true); wall flags are never consulted
    if (p < 4) return;
    int i = bi(p);
    if (!((awakeMask >> i) & 1)) {
      awakeMask |= 1u << i;
      sleepTime[i] = 0.0f;
    }
  }
  __device__ __forceinline__ void sleepBody(int i) {  // b2Body::SetAwake(false)
    awakeMask &= ~(1u << i);
    sleepTime[i] = 0.0f;
    v[i] = V2(0.0f, 0.0f);
    w[i] = 0.0f;
  }
  __device__ __forceinline__ void syncTransform(int i) {  // b2Body::SynchronizeTransform
    q[i] = rotFor(4 + i, a[i]);
    xfp[i] = c[i] - Mul(q[i], lc[i]);
  }
  // fixture order of a slot after b2Contact::Create's type normalisation
  __device__ __forceinline__ void slotAB(int s, int* pA, int* pB) const {
    int pa_ = pairAOf(s), pb_ = pairBOf(s);
    if (pa_ >= 4) {
      int ta = typeOf(pa_), tb = typeOf(pb_);
      // rank: polygon(1) before circle(0)  <=> swap when A is a circle and B a polygon
      if (ta == kCircle && tb == kPolygon) {
        int t = pa_;
        pa_ = pb_;
        pb_ = t;
      }
    }
    *pA = pa_;
    *pB = pb_;
  }

  // ------------------------------------------------------------------------------------------------
  // broad phase: b2Fixture::Synchronize + b2DynamicTree::MoveProxy, b2BroadPhase::UpdatePairs, AddPair
  // ------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void synchronizeProxy(int i, const Transform& xf1, const Transform& xf2) {
    AABB aabb1, aabb2, aabb;
    if (kCirc) {
      const Shape cs = circShapeReg(i);
      CircleComputeAABB(&cs, &aabb1, xf1);
      CircleComputeAABB(&cs, &aabb2, xf2);
    } else {
      const Shape* sh = shapeOf(4 + i);
      ShapeComputeAABB(sh, &aabb1, xf1);
      ShapeComputeAABB(sh, &aabb2, xf2);
    }
    aabb.lo = Min(aabb1.lo, aabb2.lo);
    aabb.hi = Max(aabb1.hi, aabb2.hi);
    Vec2 displacement = xf2.p - xf1.p;
    const AABB fatOld_ = fat[i];
    if (fatOld_.Contains(aabb)) return;
    AABB fb = aabb;
    Vec2 r = V2(kAabbExtension, kAabbExtension);
    fb.lo = fb.lo - r;
    fb.hi = fb.hi + r;
    Vec2 d = kAabbMultiplier * displacement;
    if (d.x < 0.0f) fb.lo.x += d.x; else fb.hi.x += d.x;
    if (d.y < 0.0f) fb.lo.y += d.y; else fb.hi.y += d.y;
    fat[i] = fb;
    moveMask |= 1u << i;
  }
  __device__ __forceinline__ void synchronizeFixtures(int i) {  // b2Body::SynchronizeFixtures
    Transform xf1;
    xf1.q = rotFor(4 + i, a0[i]);
    xf1.p = c0[i] - Mul(xf1.q, lc[i]);
    synchronizeProxy(i, xf1, xfOf(4 + i));
  }
  __device__ __forceinline__ void findNewContacts(bool allMoved) {
    const int np = S->np;
    uint32_t mm = moveMask;
    moveMask = 0;
    if (!allMoved && mm == 0) return;
#pragma unroll kUS
    for (int s = 0; s < NP; ++s) {
      if (s >= np) break;
      if (pflags.get(s) & PF_EXISTS) continue;
      int pa_ = pairAOf(s), pb_ = pairBOf(s);
      bool moved = allMoved || (pa_ >= 4 && ((mm >> (pa_ - 4)) & 1)) || ((mm >> (pb_ - 4)) & 1);
      if (!moved) continue;
      if (!TestOverlap(fatOf(pa_), fatOf(pb_))) continue;
      // b2ContactManager::AddPair: new contact at the front of the world list; wake both bodies
      wl.insertFront(nc, s);
      ++nc;
      pflags.set(s, PF_EXISTS | PF_ENABLED);
      manClearCount(s);
      toiCount.set(s, 0);
      selSet(toi, s, (float)(1.0f));
      wake(pa_);
      wake(pb_);
    }
  }

  // ------------------------------------------------------------------------------------------------
  // b2Contact::Update (+ Evaluate dispatch), b2ContactManager::Collide
  // ------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void updateContact(int s) {
    int pA, pB;
    slotAB(s, &pA, &pB);
    Manifold m = manGet(s);
    Manifold oldManifold = m;
    pflags.orBits(s, PF_ENABLED);
    bool wasTouching = (pflags.get(s) & PF_TOUCHING) != 0;
    Transform xfA = xfOf(pA), xfB = xfOf(pB);
    int ta = typeOf(pA), tb = typeOf(pB);
    if (ta == kEdge) {   // a wall: blcd_collide_wall.h
      const WallK wA = selGet(wallK, pA);
      if (kCirc) {
        CollideWallCircle(&m, wA, cctr[bi(pB)], crad[bi(pB)], xfB);
      } else if (tb == kCircle) {
        const Shape* cB = shapeOf(pB);
        CollideWallCircle(&m, wA, cB->v[0], cB->radius, xfB);
      } else {
        CollideWallPolygon(&m, wA, shapeOf(pB), xfB);
      }
    } else if (ta == kPolygon) {
      if (tb == kCircle) CollidePolygonAndCircle(&m, shapeOf(pA), xfA, shapeOf(pB), xfB);
      else CollidePolygons(&m, shapeOf(pA), xfA, shapeOf(pB), xfB);
    } else if (kCirc) {
      const Shape cA = circShapeReg(bi(pA)), cB = circShapeReg(bi(pB));
      CollideCircles(&m, &cA, xfA, &cB, xfB);
    } else {
      CollideCircles(&m, shapeOf(pA), xfA, shapeOf(pB), xfB);
    }
    bool touching = m.pointCount > 0;
#pragma unroll
    for (int i = 0; i < kMP; ++i) {
      if (i >= m.pointCount) break;
      ManifoldPoint* mp2 = m.points + i;
      mp2->normalImpulse = 0.0f;
      mp2->tangentImpulse = 0.0f;
      uint32_t key = mp2->id.key;
#pragma unroll
      for (int j = 0; j < kMP; ++j) {
        if (j >= oldManifold.pointCount) break;
        const ManifoldPoint* mp1 = oldManifold.points + j;
        if (mp1->id.key == key) {
          mp2->normalImpulse = mp1->normalImpulse;
          mp2->tangentImpulse = mp1->tangentImpulse;
          break;
        }
      }
    }
    if (touching != wasTouching) {
      wake(pA);
      wake(pB);
    }
    if (touching) pflags.orBits(s, PF_TOUCHING); else pflags.clearBits(s, PF_TOUCHING);
    manSet(s, m);
  }

  __device__ __forceinline__ void collide() {
    int n = nc, out = 0;
    for (int k = 0; k < n; ++k) {
      int s = wl.get(k);
      int pa_ = pairAOf(s), pb_ = pairBOf(s);
      bool activeA = pa_ >= 4 && awakeDyn(pa_ - 4);
      bool activeB = awakeDyn(pb_ - 4);
      if (activeA || activeB) {
        if (!TestOverlap(fatOf(pa_), fatOf(pb_))) {
          // b2ContactManager::Destroy + b2Contact::Destroy
          Manifold dm = manGet(s);
          if (dm.pointCount > 0) {
            wake(pa_);
            wake(pb_);
          }
          pflags.set(s, 0);
          dm.pointCount = 0;
          manSet(s, dm);
          continue;
        }
        updateContact(s);
      }
      wl.set(out++, s);
    }
    nc = out;
  }

  // ------------------------------------------------------------------------------------------------
  // island state accessors: dynamic bodies from pc/pa/pv/pw, walls constant
  // ------------------------------------------------------------------------------------------------
  __device__ __forceinline__ Vec2 Pc(int p) const { return p < 4 ? V2(0.0f, 0.0f) : pc.get(bi(p)); }
  __device__ __forceinline__ float Pa(int p) const { return p < 4 ? 0.0f : pa.get(bi(p)); }
  __device__ __forceinline__ Vec2 Pv(int p) const { return p < 4 ? V2(0.0f, 0.0f) : pv.get(bi(p)); }
  __device__ __forceinline__ float Pw(int p) const { return p < 4 ? 0.0f : pw.get(bi(p)); }
  __device__ __forceinline__ float mOf(int p) const { return p < 4 ? 0.0f : invMass[bi(p)]; }
  __device__ __forceinline__ float iOf(int p) const { return p < 4 ? 0.0f : invI[bi(p)]; }
  __device__ __forceinline__ Vec2 lcOf(int p) const { return p < 4 ? V2(0.0f, 0.0f) : lc[bi(p)]; }
  __device__ __forceinline__ void setVel(int p, Vec2 vv, float ww) {
    if (p >= 4) {
      pv.set(bi(p), vv);
      pw.set(bi(p), ww);
    }
  }
  __device__ __forceinline__ void setPos(int p, Vec2 cc, float aa) {
    if (p >= 4) {
      pc.set(bi(p), cc);
      pa.set(bi(p), aa);
    }
  }

  // ------------------------------------------------------------------------------------------------
  // b2ContactSolver
  // ------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void csInit(int count, bool warmStarting, float dtRatio) {
#ifndef BLCD_NO_KILL
    // End the live ranges of the previous island's constraint rows here.  The loops below define rows / points conditionally
    // (i < count, j < pointCount), so without this every word of the register-resident vc[] stays live from one island solve
    // to the next - through collide and the TOI root finder - as the "else" value of a phi: the general one-body kernel needed
    // 512 registers (+ 18 spills) and 668 spills under a 256-register bound; with the rows value-initialised here it needs 411,
    // and 177 spills (none in a hot loop) under the 256-register bound that lets two waves share a SIMD.
    // Bit-neutral: every word read after this point is written first (it was before, or the old code read a stale row).
    // (not for the circles-only class: it fits two waves anyway - 212 registers - and its islands settle in 2-5 sweeps, so the
    // ~70 extra moves per island solve cost Bounce-100k 3 %)
    if constexpr (kMaxC <= 4 && !kCirc) {
#pragma unroll
      for (int i = 0; i < kMaxC; ++i) vc[i] = VC{};
    }
#endif
#pragma unroll kU
    for (int i = 0; i < kMaxC; ++i) {
      if (i >= count) break;
      int s = ic.get(i);
      VC& c_ = vc[i];
      const Manifold m = manGet(s);
      c_.slot = s;
      slotAB(s, &c_.pA, &c_.pB);
      c_.friction = S->pairs[s].friction;
      c_.restitution = S->pairs[s].restitution;
      c_.pointCount = m.pointCount;
      c_.K.ex = c_.K.ey = V2(0.0f, 0.0f);
      c_.normalMass.ex = c_.normalMass.ey = V2(0.0f, 0.0f);
#pragma unroll
      for (int j = 0; j < kMP; ++j) {
        if (j >= m.pointCount) break;
        VCPoint& p = c_.points[j];
        if (warmStarting) {
          p.normalImpulse = dtRatio * m.points[j].normalImpulse;
          p.tangentImpulse = dtRatio * m.points[j].tangentImpulse;
        } else {
          p.normalImpulse = 0.0f;
          p.tangentImpulse = 0.0f;
        }
        p.rA = V2(0.0f, 0.0f);
        p.rB = V2(0.0f, 0.0f);
        p.normalMass = 0.0f;
        p.tangentMass = 0.0f;
        p.velocityBias = 0.0f;
      }
    }
  }

  __device__ __forceinline__ void csInitVelocityConstraints(int count) {
#pragma unroll kU
    for (int i = 0; i < kMaxC; ++i) {
      if (i >= count) break;
      VC& c_ = vc[i];
      const Manifold mcopy = manGet(c_.slot);
      const Manifold* manifold = &mcopy;
      int pA = c_.pA, pB = c_.pB;
      float radiusA = radiusOf(pA), radiusB = radiusOf(pB);
      float mA = mOf(pA), mB = mOf(pB), iA = iOf(pA), iB = iOf(pB);
      Vec2 localCenterA = lcOf(pA), localCenterB = lcOf(pB);
      Vec2 cA = Pc(pA);
      float aA = Pa(pA);
      Vec2 vA = Pv(pA);
      float wA = Pw(pA);
      Vec2 cB = Pc(pB);
      float aB = Pa(pB);
      Vec2 vB = Pv(pB);
      float wB = Pw(pB);
      Transform xfA, xfB;
      xfA.q = rotFor(pA, aA);
      xfB.q = rotFor(pB, aB);
      xfA.p = cA - Mul(xfA.q, localCenterA);
      xfB.p = cB - Mul(xfB.q, localCenterB);
      WorldManifold worldManifold;
      worldManifold.Initialize(manifold, xfA, radiusA, xfB, radiusB);
      c_.normal = worldManifold.normal;
      int pointCount = c_.pointCount;
#pragma unroll
      for (int j = 0; j < kMP; ++j) {
        if (j >= pointCount) break;
        VCPoint* vcp = c_.points + j;
        vcp->rA = worldManifold.points[j] - cA;
        vcp->rB = worldManifold.points[j] - cB;
        float rnA = Cross(vcp->rA, c_.normal);
        float rnB = Cross(vcp->rB, c_.normal);
        float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
        vcp->normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
        Vec2 tangent = Cross(c_.normal, 1.0f);
        float rtA = Cross(vcp->rA, tangent);
        float rtB = Cross(vcp->rB, tangent);
        float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
        vcp->tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
        vcp->velocityBias = 0.0f;
        float vRel = Dot(c_.normal, vB + Cross(wB, vcp->rB) - vA - Cross(wA, vcp->rA));
        if (vRel < -kVelocityThreshold) vcp->velocityBias = -c_.restitution * vRel;
      }
      if (kMP == 2 && c_.pointCount == 2) {
        VCPoint* vcp1 = c_.points + 0;
        VCPoint* vcp2 = c_.points + 1;
        float rn1A = Cross(vcp1->rA, c_.normal);
        float rn1B = Cross(vcp1->rB, c_.normal);
        float rn2A = Cross(vcp2->rA, c_.normal);
        float rn2B = Cross(vcp2->rB, c_.normal);
        float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
        float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
        float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
        const float k_maxConditionNumber = 1000.0f;
        if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12)) {
          c_.K.ex = V2(k11, k12);
          c_.K.ey = V2(k12, k22);
          c_.normalMass = c_.K.GetInverse();
        } else {
          c_.pointCount = 1;
        }
      }
    }
  }

  __device__ __forceinline__ void csWarmStart(int count) {
#pragma unroll kU
    for (int i = 0; i < kMaxC; ++i) {
      if (i >= count) break;
      VC& c_ = vc[i];
      int pA = c_.pA, pB = c_.pB;
      float mA = mOf(pA), iA = iOf(pA), mB = mOf(pB), iB = iOf(pB);
      Vec2 vA = Pv(pA);
      float wA = Pw(pA);
      Vec2 vB = Pv(pB);
      float wB = Pw(pB);
      Vec2 normal = c_.normal;
      Vec2 tangent = Cross(normal, 1.0f);
#pragma unroll
      for (int j = 0; j < kMP; ++j) {
        if (j >= c_.pointCount) break;
        VCPoint* vcp = c_.points + j;
        Vec2 P = vcp->normalImpulse * normal + vcp->tangentImpulse * tangent;
        if constexpr (!kWallA) {
          wA -= iA * Cross(vcp->rA, P);
          vA -= mA * P;
        }
        wB += iB * Cross(vcp->rB, P);
        vB += mB * P;
      }
      if constexpr (!kWallA) setVel(pA, vA, wA);
      setVel(pB, vB, wB);
    }
  }

  // Returns true iff some constraint applied a non-zero impulse in this sweep.  A sweep is a deterministic function of
  // (velocities, accumulated impulses); Box2D applies lambda = clamp(acc + d) - acc, so "no accumulator moved" means
  // every applied impulse was exactly zero and the state is a fixed point: all remaining sweeps are no-ops.
  template <bool TRACK = true>   // TRACK = false: the caller does not look at the result (velocitySweeps' tail)
  __device__ __forceinline__ bool csSolveVelocityConstraints(int count) {
    bool changed = false;
#pragma unroll kU
    for (int i = 0; i < kMaxC; ++i) {
      if (i >= count) break;
      VC& c_ = vc[i];
      int pA = c_.pA, pB = c_.pB;
      float mA = mOf(pA), iA = iOf(pA), mB = mOf(pB), iB = iOf(pB);
      int pointCount = c_.pointCount;
      Vec2 vA = Pv(pA);
      float wA = Pw(pA);
      Vec2 vB = Pv(pB);
      float wB = Pw(pB);
      Vec2 normal = c_.normal;
      Vec2 tangent = Cross(normal, 1.0f);
      float friction = c_.friction;
#pragma unroll
      for (int j = 0; j < kMP; ++j) {
        if (j >= pointCount) break;
        VCPoint* vcp = c_.points + j;
        Vec2 dv = vB + Cross(wB, vcp->rB);
        if constexpr (!kWallA) dv = dv - vA - Cross(wA, vcp->rA);
        float vt = Dot(dv, tangent) - 0.0f;
        float lambda = vcp->tangentMass * (-vt);
        float maxFriction = friction * vcp->normalImpulse;
        float newImpulse = Clamp(vcp->tangentImpulse + lambda, -maxFriction, maxFriction);
        lambda = newImpulse - vcp->tangentImpulse;
        vcp->tangentImpulse = newImpulse;
        if constexpr (TRACK) changed = changed || (lambda != 0.0f);
        Vec2 P = lambda * tangent;
        if constexpr (!kWallA) {
          vA -= mA * P;
          wA -= iA * Cross(vcp->rA, P);
        }
        vB += mB * P;
        wB += iB * Cross(vcp->rB, P);
      }
      if (kMP == 1 || pointCount == 1) {
        VCPoint* vcp = c_.points + 0;
        Vec2 dv = vB + Cross(wB, vcp->rB);
        if constexpr (!kWallA) dv = dv - vA - Cross(wA, vcp->rA);
        float vn = Dot(dv, normal);
        float lambda = -vcp->normalMass * (vn - vcp->velocityBias);
        float newImpulse = Max(vcp->normalImpulse + lambda, 0.0f);
        lambda = newImpulse - vcp->normalImpulse;
        vcp->normalImpulse = newImpulse;
        if constexpr (TRACK) changed = changed || (lambda != 0.0f);
        Vec2 P = lambda * normal;
        if constexpr (!kWallA) {
          vA -= mA * P;
          wA -= iA * Cross(vcp->rA, P);
        }
        vB += mB * P;
        wB += iB * Cross(vcp->rB, P);
      } else {
        VCPoint* cp1 = c_.points + 0;
        VCPoint* cp2 = c_.points + 1;
        Vec2 a_ = V2(cp1->normalImpulse, cp2->normalImpulse);
        Vec2 dv1 = vB + Cross(wB, cp1->rB);
        Vec2 dv2 = vB + Cross(wB, cp2->rB);
        if constexpr (!kWallA) {
          dv1 = dv1 - vA - Cross(wA, cp1->rA);
          dv2 = dv2 - vA - Cross(wA, cp2->rA);
        }
        float vn1 = Dot(dv1, normal);
        float vn2 = Dot(dv2, normal);
        Vec2 b;
        b.x = vn1 - cp1->velocityBias;
        b.y = vn2 - cp2->velocityBias;
        b -= Mul(c_.K, a_);
        Vec2 x;
        bool solved = false;
        // case 1
        x = -Mul(c_.normalMass, b);
        if (x.x >= 0.0f && x.y >= 0.0f) solved = true;
        if (!solved) {  // case 2
          x.x = -cp1->normalMass * b.x;
          x.y = 0.0f;
          vn2 = c_.K.ex.y * x.x + b.y;
          if (x.x >= 0.0f && vn2 >= 0.0f) solved = true;
        }
        if (!solved) {  // case 3
          x.x = 0.0f;
          x.y = -cp2->normalMass * b.y;
          vn1 = c_.K.ey.x * x.y + b.x;
          if (x.y >= 0.0f && vn1 >= 0.0f) solved = true;
        }
        if (!solved) {  // case 4
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
          if constexpr (!kWallA) {
            vA -= mA * (P1 + P2);
            wA -= iA * (Cross(cp1->rA, P1) + Cross(cp2->rA, P2));
          }
          vB += mB * (P1 + P2);
          wB += iB * (Cross(cp1->rB, P1) + Cross(cp2->rB, P2));
          cp1->normalImpulse = x.x;
          cp2->normalImpulse = x.y;
        }
      }
      if constexpr (!kWallA) setVel(pA, vA, wA);
      setVel(pB, vB, wB);
    }
    return changed;
  }

  __device__ __forceinline__ void csStoreImpulses(int count) {
#pragma unroll kU
    for (int i = 0; i < kMaxC; ++i) {
      if (i >= count) break;
      VC& c_ = vc[i];
      Manifold m = manGet(c_.slot);
#pragma unroll
      for (int j = 0; j < kMP; ++j) {
        if (j < c_.pointCount) {
          m.points[j].normalImpulse = c_.points[j].normalImpulse;
          m.points[j].tangentImpulse = c_.points[j].tangentImpulse;
        }
      }
      manSet(c_.slot, m);
    }
  }

  // b2PositionSolverManifold::Initialize + one b2ContactSolver position iteration (baumgarte/limit differ for TOI)
  __device__ __forceinline__ float csSolvePosition(int count, bool toiMode, int toiBody) {
    float minSeparation = 0.0f;
#pragma unroll kU
    for (int i = 0; i < kMaxC; ++i) {
      if (i >= count) break;
      VC& c_ = vc[i];
      const Manifold m = manGet(c_.slot);
      int pA = c_.pA, pB = c_.pB;
      Vec2 localCenterA = lcOf(pA), localCenterB = lcOf(pB);
      float mA = mOf(pA), iA = iOf(pA), mB = mOf(pB), iB = iOf(pB);
      if (toiMode) {  // only the TOI pair moves (the wall of the pair has zero mass anyway)
        if (pA != toiBody) {
          mA = 0.0f;
          iA = 0.0f;
        }
        if (pB != toiBody) {
          mB = 0.0f;
          iB = 0.0f;
        }
      }
      float radiusA = radiusOf(pA), radiusB = radiusOf(pB);
      int pointCount = m.pointCount;
      Vec2 cA = Pc(pA);
      float aA = Pa(pA);
      Vec2 cB = Pc(pB);
      float aB = Pa(pB);
#pragma unroll
      for (int j = 0; j < kMP; ++j) {
        if (j >= pointCount) break;
        Transform xfA, xfB;
        xfA.q = rotFor(pA, aA);
        xfB.q = rotFor(pB, aB);
        xfA.p = cA - Mul(xfA.q, localCenterA);
        xfB.p = cB - Mul(xfB.q, localCenterB);
        Vec2 normal, point;
        float separation;
        if (m.type == kManifoldCircles) {
          Vec2 pointA = Mul(xfA, m.localPoint);
          Vec2 pointB = Mul(xfB, m.points[0].localPoint);
          normal = pointB - pointA;
          Normalize(normal);
          point = 0.5f * (pointA + pointB);
          separation = Dot(pointB - pointA, normal) - radiusA - radiusB;
        } else if (m.type == kManifoldFaceA) {
          normal = Mul(xfA.q, m.localNormal);
          Vec2 planePoint = Mul(xfA, m.localPoint);
          Vec2 clipPoint = Mul(xfB, m.points[j].localPoint);
          separation = Dot(clipPoint - planePoint, normal) - radiusA - radiusB;
          point = clipPoint;
        } else {
          normal = Mul(xfB.q, m.localNormal);
          Vec2 planePoint = Mul(xfB, m.localPoint);
          Vec2 clipPoint = Mul(xfA, m.points[j].localPoint);
          separation = Dot(clipPoint - planePoint, normal) - radiusA - radiusB;
          point = clipPoint;
          normal = -normal;
        }
        Vec2 rA = point - cA;
        Vec2 rB = point - cB;
        minSeparation = Min(minSeparation, separation);
        float C = Clamp((toiMode ? kToiBaumgarte : kBaumgarte) * (separation + kLinearSlop), -kMaxLinearCorrection, 0.0f);
        float rnA = Cross(rA, normal);
        float rnB = Cross(rB, normal);
        float K = kWallA ? mB + iB * rnB * rnB : mA + mB + iA * rnA * rnA + iB * rnB * rnB;
        float impulse = K > 0.0f ? -C / K : 0.0f;
        Vec2 P = impulse * normal;
        if constexpr (!kWallA) {
          cA -= mA * P;
          aA -= iA * Cross(rA, P);
        }
        cB += mB * P;
        aB += iB * Cross(rB, P);
      }
      if constexpr (!kWallA) setPos(pA, cA, aA);
      setPos(pB, cB, aB);
    }
    return minSeparation;
  }

  // ------------------------------------------------------------------------------------------------
  // b2RevoluteJoint
  // ------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void jointInit(int j, bool warmStarting, float dtRatio) {
    const DevJoint& J = S->joints[j];
    int A = J.bodyA, B = J.bodyB;
    float aA = pa.get(A);
    Vec2 vA = pv.get(A);
    float wA = pw.get(A);
    float aB = pa.get(B);
    Vec2 vB = pv.get(B);
    float wB = pw.get(B);
    Rot qA = MakeRot(aA), qB = MakeRot(aB);
    jrA[j] = Mul(qA, J.anchorA - lc[A]);
    jrB[j] = Mul(qB, J.anchorB - lc[B]);
    Vec2 rA = jrA[j], rB = jrB[j];
    float mA = invMass[A], mB = invMass[B];
    float iA = invI[A], iB = invI[B];
    bool fixedRotation = (iA + iB == 0.0f);
    Mat33& M = jmass[j];
    M.ex.x = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
    M.ey.x = -rA.y * rA.x * iA - rB.y * rB.x * iB;
    M.ez.x = -rA.y * iA - rB.y * iB;
    M.ex.y = M.ey.x;
    M.ey.y = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
    M.ez.y = rA.x * iA + rB.x * iB;
    M.ex.z = M.ez.x;
    M.ey.z = M.ez.y;
    M.ez.z = iA + iB;
    float motorMass = iA + iB;
    if (motorMass > 0.0f) motorMass = 1.0f / motorMass;
    jmotorMass[j] = motorMass;
    if (fixedRotation) jmotor[j] = 0.0f;
    if (J.enableLimit && fixedRotation == false) {
      float jointAngle = aB - aA - jref[j];
      if (Abs(J.upper - J.lower) < 2.0f * kAngularSlop) {
        jlimit[j] = kEqualLimits;
      } else if (jointAngle <= J.lower) {
        if (jlimit[j] != kAtLowerLimit) jimp[j].z = 0.0f;
        jlimit[j] = kAtLowerLimit;
      } else if (jointAngle >= J.upper) {
        if (jlimit[j] != kAtUpperLimit) jimp[j].z = 0.0f;
        jlimit[j] = kAtUpperLimit;
      } else {
        jlimit[j] = kInactiveLimit;
        jimp[j].z = 0.0f;
      }
    } else {
      jlimit[j] = kInactiveLimit;
    }
    if (warmStarting) {
      jimp[j] *= dtRatio;
      jmotor[j] *= dtRatio;
      Vec2 P = V2(jimp[j].x, jimp[j].y);
      vA -= mA * P;
      wA -= iA * (Cross(rA, P) + jmotor[j] + jimp[j].z);
      vB += mB * P;
      wB += iB * (Cross(rB, P) + jmotor[j] + jimp[j].z);
    } else {
      jimp[j] = Vec3{0.0f, 0.0f, 0.0f};
      jmotor[j] = 0.0f;
    }
    pv.set(A, vA);
    pw.set(A, wA);
    pv.set(B, vB);
    pw.set(B, wB);
  }

  __device__ __forceinline__ bool jointSolveVelocity(int j, float dt) {
    bool changed = false;
    const DevJoint& J = S->joints[j];
    int A = J.bodyA, B = J.bodyB;
    Vec2 vA = pv.get(A);
    float wA = pw.get(A);
    Vec2 vB = pv.get(B);
    float wB = pw.get(B);
    float mA = invMass[A], mB = invMass[B];
    float iA = invI[A], iB = invI[B];
    Vec2 rA = jrA[j], rB = jrB[j];
    const Mat33& M = jmass[j];
    bool fixedRotation = (iA + iB == 0.0f);
    int limitState = jlimit[j];
    if (limitState != kEqualLimits && fixedRotation == false) {  // enableMotor is always true (world_env.py:260)
      float Cdot = wB - wA - jspeed[j];
      float impulse = -jmotorMass[j] * Cdot;
      float oldImpulse = jmotor[j];
      float maxImpulse = dt * J.maxMotorTorque;
      jmotor[j] = Clamp(jmotor[j] + impulse, -maxImpulse, maxImpulse);
      impulse = jmotor[j] - oldImpulse;
      changed = changed || (impulse != 0.0f);
      wA -= iA * impulse;
      wB += iB * impulse;
    }
    if (J.enableLimit && limitState != kInactiveLimit && fixedRotation == false) {
      Vec2 Cdot1 = vB + Cross(wB, rB) - vA - Cross(wA, rA);
      float Cdot2 = wB - wA;
      Vec3 Cdot = Vec3{Cdot1.x, Cdot1.y, Cdot2};
      Vec3 impulse = -M.Solve33(Cdot);
      Vec3 acc = jimp[j];
      if (limitState == kEqualLimits) {
        acc += impulse;
      } else if (limitState == kAtLowerLimit) {
        float newImpulse = acc.z + impulse.z;
        if (newImpulse < 0.0f) {
          Vec2 rhs = -Cdot1 + acc.z * V2(M.ez.x, M.ez.y);
          Vec2 reduced = M.Solve22(rhs);
          impulse.x = reduced.x;
          impulse.y = reduced.y;
          impulse.z = -acc.z;
          acc.x += reduced.x;
          acc.y += reduced.y;
          acc.z = 0.0f;
        } else {
          acc += impulse;
        }
      } else if (limitState == kAtUpperLimit) {
        float newImpulse = acc.z + impulse.z;
        if (newImpulse > 0.0f) {
          Vec2 rhs = -Cdot1 + acc.z * V2(M.ez.x, M.ez.y);
          Vec2 reduced = M.Solve22(rhs);
          impulse.x = reduced.x;
          impulse.y = reduced.y;
          impulse.z = -acc.z;
          acc.x += reduced.x;
          acc.y += reduced.y;
          acc.z = 0.0f;
        } else {
          acc += impulse;
        }
      }
      changed = changed || (impulse.x != 0.0f) || (impulse.y != 0.0f) || (impulse.z != 0.0f) || (acc.x != jimp[j].x) ||
                (acc.y != jimp[j].y) || (acc.z != jimp[j].z);
      jimp[j] = acc;
      Vec2 P = V2(impulse.x, impulse.y);
      vA -= mA * P;
      wA -= iA * (Cross(rA, P) + impulse.z);
      vB += mB * P;
      wB += iB * (Cross(rB, P) + impulse.z);
    } else {
      Vec2 Cdot = vB + Cross(wB, rB) - vA - Cross(wA, rA);
      Vec2 impulse = M.Solve22(-Cdot);
      changed = changed || (impulse.x != 0.0f) || (impulse.y != 0.0f);
      jimp[j].x += impulse.x;
      jimp[j].y += impulse.y;
      vA -= mA * impulse;
      wA -= iA * Cross(rA, impulse);
      vB += mB * impulse;
      wB += iB * Cross(rB, impulse);
    }
    pv.set(A, vA);
    pw.set(A, wA);
    pv.set(B, vB);
    pw.set(B, wB);
    return changed;
  }

  __device__ __forceinline__ bool jointSolvePosition(int j) {
    const DevJoint& J = S->joints[j];
    int A = J.bodyA, B = J.bodyB;
    Vec2 cA = pc.get(A);
    float aA = pa.get(A);
    Vec2 cB = pc.get(B);
    float aB = pa.get(B);
    float mA = invMass[A], mB = invMass[B];
    float iA = invI[A], iB = invI[B];
    float angularError = 0.0f;
    float positionError = 0.0f;
    bool fixedRotation = (iA + iB == 0.0f);
    int limitState = jlimit[j];
    if (J.enableLimit && limitState != kInactiveLimit && fixedRotation == false) {
      float angle = aB - aA - jref[j];
      float limitImpulse = 0.0f;
      if (limitState == kEqualLimits) {
        float C = Clamp(angle - J.lower, -kMaxAngularCorrection, kMaxAngularCorrection);
        limitImpulse = -jmotorMass[j] * C;
        angularError = Abs(C);
      } else if (limitState == kAtLowerLimit) {
        float C = angle - J.lower;
        angularError = -C;
        C = Clamp(C + kAngularSlop, -kMaxAngularCorrection, 0.0f);
        limitImpulse = -jmotorMass[j] * C;
      } else if (limitState == kAtUpperLimit) {
        float C = angle - J.upper;
        angularError = C;
        C = Clamp(C - kAngularSlop, 0.0f, kMaxAngularCorrection);
        limitImpulse = -jmotorMass[j] * C;
      }
      aA -= iA * limitImpulse;
      aB += iB * limitImpulse;
    }
    {
      Rot qA = MakeRot(aA), qB = MakeRot(aB);
      Vec2 rA = Mul(qA, J.anchorA - lc[A]);
      Vec2 rB = Mul(qB, J.anchorB - lc[B]);
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
    pc.set(A, cA);
    pa.set(A, aA);
    pc.set(B, cB);
    pa.set(B, aB);
    return positionError <= kLinearSlop && angularError <= kAngularSlop;
  }


  // ------------------------------------------------------------------------------------------------
  // The velocity sweeps.  Box2D always runs `velIters` sweeps; here the loop ends early when that provably cannot change
  // the result (the parity oracle runs all of them):
  //  * fixed point: a sweep applied exactly zero impulse everywhere => every later sweep is a no-op;
  //  * short cycle: the sweep is a deterministic function of (velocities, accumulated impulses); if the state after sweep k
  //    equals the state after sweep k-p the sequence is p-periodic from there on, so the state after the last sweep is the
  //    stored state at index k-p+((velIters-1-k) mod p).  (1-ulp rounding ping-pong, period 2-4, is what keeps ~1 % of
  //    solves from ever reaching a fixed point.)
  // ------------------------------------------------------------------------------------------------
  struct CycRow {
    float v[kCycW];
  };
  struct CycDig {
    uint32_t d;
  };
  // static layout: body i at [3*i..3*i+2] (bodies outside the island stay 0), contact k at [3*kCycNB + 4*k ..]
  __device__ __forceinline__ void cycPack(uint32_t ibmask, int nic, CycRow& dst) const {
#pragma unroll
    for (int q = 0; q < kCycW; ++q) dst.v[q] = 0.0f;
#pragma unroll
    for (int i = 0; i < kCycNB; ++i) {
      if ((ibmask >> i) & 1) {
        const Vec2 pvi_ = pv.get(i);
        dst.v[3 * i] = pvi_.x;
        dst.v[3 * i + 1] = pvi_.y;
        dst.v[3 * i + 2] = pw.get(i);
      }
    }
#pragma unroll
    for (int k = 0; k < kCycNC; ++k) {
      if (k < nic) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bool live = j < kMP && j < vc[k].pointCount;
          dst.v[3 * kCycNB + 4 * k + 2 * j] = live ? vc[k].points[j].normalImpulse : 0.0f;
          dst.v[3 * kCycNB + 4 * k + 2 * j + 1] = live ? vc[k].points[j].tangentImpulse : 0.0f;
        }
      }
    }
  }
  __device__ __forceinline__ void cycUnpack(uint32_t ibmask, int nic, const CycRow& src) {
#pragma unroll
    for (int i = 0; i < kCycNB; ++i) {
      if ((ibmask >> i) & 1) {
        pv.set(i, V2(src.v[3 * i], src.v[3 * i + 1]));
        pw.set(i, src.v[3 * i + 2]);
      }
    }
#pragma unroll
    for (int k = 0; k < kCycNC; ++k) {
      if (k < nic) {
#pragma unroll
        for (int j = 0; j < kMP; ++j) {
          if (j < vc[k].pointCount) {
            vc[k].points[j].normalImpulse = src.v[3 * kCycNB + 4 * k + 2 * j];
            vc[k].points[j].tangentImpulse = src.v[3 * kCycNB + 4 * k + 2 * j + 1];
          }
        }
      }
    }
  }
  // returns true when the lane suspends at sweep kYieldSweeps (only asked of joint-free islands, see kCanYield)
  // Start a fresh live range in a VGPR right here (empty asm, value unchanged).  The sweep loop's constants are defined long before
  // the loop (constraint initialisation) and the register allocator - 512 registers, half of them AGPRs that no VALU instruction
  // can read - parks them in AGPRs in favour of state the loop never touches: 80 of the 286 instructions of a one-body sweep were
  // v_accvgpr_read.  A value that is born at the loop's door and dies behind it gets a register of its own.
#ifndef BLCD_NO_PIN
#define BLCD_PIN(x_) asm volatile("" : "+v"(x_))
#else
#define BLCD_PIN(x_) do {} while (0)
#endif
  __device__ __forceinline__ void pinSweepConstants() {
    if constexpr (NB == 1) {
#pragma unroll
      for (int k = 0; k < kMaxC; ++k) {
        VC& c_ = vc[k];
        BLCD_PIN(c_.normal.x); BLCD_PIN(c_.normal.y); BLCD_PIN(c_.friction);
        if constexpr (kMP == 2) {
          BLCD_PIN(c_.K.ex.x); BLCD_PIN(c_.K.ex.y); BLCD_PIN(c_.K.ey.x); BLCD_PIN(c_.K.ey.y);
          BLCD_PIN(c_.normalMass.ex.x); BLCD_PIN(c_.normalMass.ex.y); BLCD_PIN(c_.normalMass.ey.x); BLCD_PIN(c_.normalMass.ey.y);
        }
#pragma unroll
        for (int j = 0; j < kMP; ++j) {
          VCPoint& p_ = c_.points[j];
          BLCD_PIN(p_.rB.x); BLCD_PIN(p_.rB.y); BLCD_PIN(p_.normalMass); BLCD_PIN(p_.tangentMass); BLCD_PIN(p_.velocityBias);
        }
      }
      BLCD_PIN(invMass[0]); BLCD_PIN(invI[0]);
    }
  }
  __device__ __forceinline__ bool velocitySweeps(uint32_t ibmask, int nic, int nij, float h, int startIt = 0, bool mayYield = false) {
    const int velIters = S->velIters;
    pinSweepConstants();
    bool watch = nij == 0 && nic > 0 && nic <= kCycNC && (ibmask >> kCycNB) == 0;
#ifndef BLCD_CYC_WINDOW
    constexpr int kCycWatch = BLCD_CYC_WATCH;
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
      for (int k = 0; k < nij; ++k) changed = jointSolveVelocity(ij[k], h) || changed;
      changed = csSolveVelocityConstraints(nic) || changed;
      if (!changed) break;
#ifndef BLCD_CYC_WINDOW
      // Brent's cycle detection with one reference row and a digest per sweep: see RegIsland::velocitySweeps (blcd_island_reg.h)
      if (it >= last) break;
      if (watch && !cycling && it < kCycWatch) {
        CycRow cur;
        cycPack(ibmask, nic, cur);
        uint32_t dig = 0;
#pragma unroll
        for (int q = 0; q < kCycW; ++q) dig = dig * 31u + __float_as_uint(cur.v[q] + 0.0f);
        if (refIt >= 0 && dig == refDig) {
          bool same = true;
#pragma unroll
          for (int q = 0; q < kCycW; ++q) same = same && (ref.v[q] == cur.v[q]);
          if (same) {
            cycling = true;
            last = it + (velIters - 1 - it) % (it - refIt);
            if (last == it) break;
          }
        }
        if (((it + 1) & it) == 0) {
          ref = cur;
          refDig = dig;
          refIt = it;
        }
      }
      if (it == kCycWatch - 1 && !mayYield && nij == 0) {
        // the rest without the exits' bookkeeping: see RegIsland::velocitySweeps
        for (int it2 = it + 1; it2 <= last; ++it2) csSolveVelocityConstraints<false>(nic);
        break;
      }
#else
      if (watch && it < kCycSweeps) {
        CycRow cur;
        cycPack(ibmask, nic, cur);
        // digest filter: see RegIsland::velocitySweeps (blcd_island_reg.h) - rows are fetched and compared only where digests agree
        uint32_t dig = 0;
#pragma unroll
        for (int q = 0; q < kCycW; ++q) dig = dig * 31u + __float_as_uint(cur.v[q] + 0.0f);
        bool found = false;
#pragma unroll
        for (int p = 1; p <= kCycP; ++p) {
          if (found || p > it) continue;
          if (selGet(cycDig, (it - p) & (kCycP - 1)).d != dig) continue;
          const CycRow old = selGet(cyc, (it - p) & (kCycP - 1));
          bool same = true;
#pragma unroll
          for (int q = 0; q < kCycW; ++q) same = same && (old.v[q] == cur.v[q]);
          if (same) {
            int r = (velIters - 1 - it) % p;
            if (r != 0) cycUnpack(ibmask, nic, selGet(cyc, (it - p + r) & (kCycP - 1)));
            found = true;
          }
        }
        if (found) break;
        selSet(cyc, it & (kCycP - 1), cur);
        selSet(cycDig, it & (kCycP - 1), CycDig{dig});
      }
#endif
      if (mayYield && it == kYieldSweeps - 1 && velIters > kYieldSweeps) {
        // the lanes that reach this line are the wave's stragglers (everyone else left the loop at a fixed point / short cycle)
        if (__popcll(__ballot(1)) <= yieldMaxLanes) return true;
      }
    }
    return false;
  }

  // suspended-velocity words of body i (schedWordOffset + 1 + 3 i ..)
  __device__ __forceinline__ float* susWords(int i) const { return gst + (size_t)(schedWordOffset(S->nb, S->nj, S->np) + 2 + 3 * i) * gN; }

  // ------------------------------------------------------------------------------------------------
  // b2Island::Solve for the island {bodies in ibmask, contacts ic[0..nic), joints ij[0..nij)}
  // ------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void integratePositions(uint32_t ibmask, float h) {
    const int nb = S->nb;
    for (int i = 0; i < NB; ++i) {
      if (i >= nb) break;
      if (!((ibmask >> i) & 1)) continue;
      Vec2 c_ = pc.get(i);
      float a_ = pa.get(i);
      Vec2 v_ = pv.get(i);
      float w_ = pw.get(i);
      Vec2 translation = h * v_;
      if (Dot(translation, translation) > kMaxTranslationSquared) {
        float ratio = kMaxTranslation / Length(translation);
        v_ *= ratio;
      }
      float rotation = h * w_;
      if (rotation * rotation > kMaxRotationSquared) {
        float ratio = kMaxRotation / Abs(rotation);
        w_ *= ratio;
      }
      c_ += h * v_;
      a_ += h * w_;
      pc.set(i, c_);
      pa.set(i, a_);
      pv.set(i, v_);
      pw.set(i, w_);
    }
  }

  // ------------------------------------------------------------------------------------------------
  // register-resident island solve (blcd_island_reg.h) for multi-body scenes: islands with <= kRegC contacts
  // ------------------------------------------------------------------------------------------------
  static constexpr bool kUseReg = (NB > 1) && (NB <= BLCD_REG_MAXNB) && (NJ <= 4);   // NB = 7: re-enabled in round 2 (parity incl. a -ftrivial-auto-var-init=pattern build; UrchinBalls +26 %), see DESIGN.md
#ifndef BLCD_REGC7
#define BLCD_REGC7 6
#endif
  // contacts per staged island.  Robot + one object: 5 or more contacts in 0.03 % of solves; robot + three objects (7-body
  // class): 1.0-1.6 % (measured on the CPU side), and ONE such lane sends the whole wave through the generic scratch-resident solver
  static constexpr int kRegC = (NB >= 7 || NB == 3) ? BLCD_REGC7 : 4;   // three free objects: 0.5 % of their islands hold 5-6 contacts
  static constexpr bool kRegLds = kUseReg && BLCD_REG_LDS && NB >= 4;   // body rows of the staged island live in LDS (see RegIsland)
  static constexpr bool kRegCtLds = kUseReg && NB >= 4 && BLCD_REG_CLDS && (10 * NB + 22 * kRegC) * 256 <= 40960;   // + the contacts' sweep constants.  LDS budget = 40 KB per wave
  // (four waves per CU): body rows 10 KB x NB/4 + 22 KB of contact constants; the frame-store staging rows of step_kernel
  // (4.3 KB) live in the contact block, which is dead while a frame is written (ldsFrameRows)
  using RegI = RegIsland<NB, NJ, kRegC, kRegLds, kRegCtLds>;
  // this lane's column of the wave's staged-island LDS block (one block per kernel: the main solve and the TOI
  // mini-islands never overlap in time)
  __device__ __forceinline__ float* regIslandLds() {
    if constexpr (kRegLds) {
      __shared__ float blk[RegI::kLdsWords];
      return blk + threadIdx.x;
    } else {
      return nullptr;
    }
  }
  static __device__ __forceinline__ float* ctLdsBase() {
    __shared__ float blk[kRegCtLds ? RegI::kCtLdsWords : kFrameLdsWords];
    return blk;
  }
  __device__ __forceinline__ float* regContactLds() {
    if constexpr (kRegCtLds) return ctLdsBase() + threadIdx.x;
    else return nullptr;
  }
  // step_kernel's wave-coalesced 16x16 frame stores: 16 rows x (64 + 4) row masks, [row][lane] (row stride 68: a lane's four
  // consecutive frames are one aligned 16-byte read, and the 64 lanes of a store instruction - 16 rows x 4 frames - hit 64 different
  // banks), then the wave's 64 environment ids
  static constexpr int kFrameRowStride = 68;
  static constexpr int kFrameLdsWords = 16 * kFrameRowStride + 64;
  static __device__ __forceinline__ uint32_t* ldsFrameRows() {
    static_assert(!kRegCtLds || RegI::kCtLdsWords >= kFrameLdsWords, "frame rows must fit the contact block");
    return reinterpret_cast<uint32_t*>(ctLdsBase());
  }
  // MODE 0: the whole island solve; 1: resumed inside the velocity sweeps (see islandSolve); 2: resumed inside the position
  // iterations - velocities, impulses and the integrated positions are final and already in place, so only the position rows and
  // the position-constraint halves of the contacts / joints are staged and the loop continues at iteration kYieldPosIters
  template <int MODE = 0>
  __device__ __forceinline__ bool islandSolveReg(uint32_t ibmask, int nic, int nij, float h, float dtRatio, int seed = 0, bool mayYield = false) {
    constexpr bool RESUME = MODE == 1;
#ifdef BLCD_PROF_SOLVE
    const unsigned long long psE_ = __builtin_amdgcn_s_memtime();
#endif
    const int nb = S->nb;
    Vec2 gravity = S->gravity;
    RegI R;
    R.L = regIslandLds();
    R.C = regContactLds();
    R.nc = nic;
    R.nj = nij;
    R.deadQ = deadQ;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      bool in = i < nb && ((ibmask >> i) & 1);
      Vec2 v_ = v[i];
      float w_ = w[i];
      if (in && MODE != 2) {
        c0[i] = c[i];
        a0[i] = a[i];
        v_ += h * (1.0f * gravity + invMass[i] * V2(0.0f, 0.0f));
        w_ += h * invI[i] * 0.0f;
        v_ *= Clamp(1.0f - h * S->bodies[i].linearDamping, 0.0f, 1.0f);    // Box2D 2.3.0 (first-order; >= 2.3.1 is Pade)
        w_ *= Clamp(1.0f - h * S->bodies[i].angularDamping, 0.0f, 1.0f);
      }
      R.setPos(i, BodyPos{c[i], a[i]});
      R.setVel(i, BodyVel{v_, w_});
      R.setMass(i, BodyMass{invMass[i], invI[i], lc[i]});
    }
    // contacts in island order: b2ContactSolver ctor + InitializeVelocityConstraints
    Manifold mans[kRegC];
#pragma unroll
    for (int k = 0; k < kRegC; ++k) {
      if (k < nic) {
        int s = ic.get(k);
        RContact& c_ = R.ct[k];
        const Manifold m = manGet(s);
        mans[k] = m;
        slotAB(s, &c_.pA, &c_.pB);
        c_.friction = S->pairs[s].friction;
        c_.restitution = S->pairs[s].restitution;
        c_.pointCount = m.pointCount;
        c_.K.ex = c_.K.ey = V2(0.0f, 0.0f);
        c_.normalMass.ex = c_.normalMass.ey = V2(0.0f, 0.0f);
        c_.mtype = m.type;
        c_.mcount = m.pointCount;
        c_.localNormal = m.localNormal;
        c_.localPoint = m.localPoint;
        c_.lp0 = m.points[0].localPoint;
        c_.lp1 = m.points[1].localPoint;
        c_.radiusA = radiusOf(c_.pA);
        c_.radiusB = radiusOf(c_.pB);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          RPoint& p = c_.points[j];
          p.normalImpulse = dtRatio * m.points[j].normalImpulse;
          p.tangentImpulse = dtRatio * m.points[j].tangentImpulse;
          p.rA = V2(0.0f, 0.0f);
          p.rB = V2(0.0f, 0.0f);
          p.normalMass = 0.0f;
          p.tangentMass = 0.0f;
          p.velocityBias = 0.0f;
        }
      }
    }
    if constexpr (MODE != 2) {
#pragma unroll
    for (int k = 0; k < kRegC; ++k)
      if (k < nic) R.initContact(k, R.ct[k], mans[k]);
    }
    if constexpr (MODE == 2) {
    } else if constexpr (RESUME && kCanYield) {
      // the sweeps' own state as it was at suspension: island velocities, accumulated impulses (stored raw in the manifolds)
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (i < nb && ((ibmask >> i) & 1)) {
          const float* q_ = susWords(i);
          R.setVel(i, BodyVel{V2(q_[0], q_[(size_t)1 * gN]), q_[(size_t)2 * gN]});
        }
      }
#pragma unroll
      for (int k = 0; k < kRegC; ++k) {
        if (k < nic) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            R.ct[k].points[j].normalImpulse = mans[k].points[j].normalImpulse;
            R.ct[k].points[j].tangentImpulse = mans[k].points[j].tangentImpulse;
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < kRegC; ++k)
        if (k < nic) R.warmStartContact(k, R.ct[k]);
    }
    if constexpr (NJ > 0) {
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        if (k < nij) {
          int j = ij[k];
          const DevJoint& J = S->joints[j];
          RJoint& r = R.jt[k];
          r.A = J.bodyA;
          r.B = J.bodyB;
          r.anchorA = J.anchorA;
          r.anchorB = J.anchorB;
          r.enableLimit = J.enableLimit;
          r.lower = J.lower;
          r.upper = J.upper;
          r.maxMotorTorque = J.maxMotorTorque;
          r.ref = jref[j];
          r.speed = jspeed[j];
          r.imp = jimp[j];
          r.motor = jmotor[j];
          r.limit = jlimit[j];
          if constexpr (MODE == 2) {   // b2RevoluteJoint::SolvePositionConstraints reads m_motorMass: same expression as initJoint
            float motorMass = invI[r.A] + invI[r.B];
            if (motorMass > 0.0f) motorMass = 1.0f / motorMass;
            r.motorMass = motorMass;
          } else {
            R.initJoint(r, dtRatio);
          }
        }
      }
    }
#ifdef BLCD_PROF_SOLVE
    unsigned long long ps0_ = __builtin_amdgcn_s_memtime();
#ifndef BLCD_PROF_SOLVE2
    prof[3] += ps0_ - psE_;     // staging + constraint initialisation + warm start
#endif
#endif
    bool yielded = false;
    if constexpr (MODE != 2) {
    int sweeps = R.velocitySweeps(S->velIters, h, nullptr, RESUME ? kYieldSweeps : 0,
                                  (kCanYield && !RESUME && mayYield && nij == 0) ? kYieldSweeps : 0, yieldMaxLanes, &yielded);
#ifdef BLCD_PROF_SOLVE
    unsigned long long ps1_ = __builtin_amdgcn_s_memtime();
    prof[4] += ps1_ - ps0_;
#endif
    if constexpr (kCanYield) {
      if (yielded) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          if (i < nb && ((ibmask >> i) & 1)) {
            const BodyVel bv_ = R.getVel(i);
            float* q_ = susWords(i);
            q_[0] = bv_.v.x;
            q_[(size_t)1 * gN] = bv_.v.y;
            q_[(size_t)2 * gN] = bv_.w;
          }
        }
#pragma unroll
        for (int k = 0; k < kRegC; ++k) {
          if (k < nic) {
            int s = ic.get(k);
            Manifold m = mans[k];
#pragma unroll
            for (int j = 0; j < 2; ++j)
              if (j < R.ct[k].pointCount) {
                m.points[j].normalImpulse = R.ct[k].points[j].normalImpulse;
                m.points[j].tangentImpulse = R.ct[k].points[j].tangentImpulse;
              }
            manSet(s, m);
          }
        }
        velMask |= 1u << seed;
        return true;
      }
    }
    // b2ContactSolver::StoreImpulses
#pragma unroll
    for (int k = 0; k < kRegC; ++k) {
      if (k < nic) {
        int s = ic.get(k);
        Manifold m = mans[k];
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (j < R.ct[k].pointCount) {
            m.points[j].normalImpulse = R.ct[k].points[j].normalImpulse;
            m.points[j].tangentImpulse = R.ct[k].points[j].tangentImpulse;
          }
        manSet(s, m);
      }
    }
    // integrate positions
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (i < nb && ((ibmask >> i) & 1)) {
        const BodyPos bp_ = R.getPos(i);
        const BodyVel bv_ = R.getVel(i);
        Vec2 c_ = bp_.c;
        float a_ = bp_.a;
        Vec2 v_ = bv_.v;
        float w_ = bv_.w;
        Vec2 translation = h * v_;
        if (Dot(translation, translation) > kMaxTranslationSquared) {
          float ratio = kMaxTranslation / Length(translation);
          v_ *= ratio;
        }
        float rotation = h * w_;
        if (rotation * rotation > kMaxRotationSquared) {
          float ratio = kMaxRotation / Abs(rotation);
          w_ *= ratio;
        }
        c_ += h * v_;
        a_ += h * w_;
        R.setPos(i, BodyPos{c_, a_});
        R.setVel(i, BodyVel{v_, w_});
      }
    }
    }   // MODE != 2
    int pit = 0;
#ifdef BLCD_PROF_SOLVE
    unsigned long long ps2_ = __builtin_amdgcn_s_memtime();
#endif
    bool posYielded = false;
    bool positionSolved = R.positionIterations(S->posIters, &pit, MODE == 2 ? kYieldPosIters : 0,
                                               (kCanYield && MODE != 2 && mayYield) ? kYieldPosIters : 0, yieldMaxLanes, &posYielded);
#ifdef BLCD_PROF_SOLVE
    prof[5] += __builtin_amdgcn_s_memtime() - ps2_;
#endif
    // copy back (a position-suspended island too: what the iterations reached so far IS their state; its velocities are final)
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (i < nb && ((ibmask >> i) & 1)) {
        const BodyPos bp_ = R.getPos(i);
        c[i] = bp_.c;
        a[i] = bp_.a;
        if constexpr (MODE != 2) {
          const BodyVel bv_ = R.getVel(i);
          v[i] = bv_.v;
          w[i] = bv_.w;
        }
        if (!posYielded) syncTransform(i);
      }
    }
    if constexpr (NJ > 0 && MODE != 2) {
#pragma unroll
      for (int k = 0; k < NJ; ++k) {
        if (k < nij) {
          int j = ij[k];
          jimp[j] = R.jt[k].imp;
          jmotor[j] = R.jt[k].motor;
          jlimit[j] = R.jt[k].limit;
        }
      }
    }
    if constexpr (kCanYield) {
      if (posYielded) {
        posMask |= 1u << seed;
        return true;
      }
    }
    float minSleepTime = kMaxFloat;
    const float linTolSqr = kLinearSleepTolerance * kLinearSleepTolerance;
    const float angTolSqr = kAngularSleepTolerance * kAngularSleepTolerance;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (i < nb && ((ibmask >> i) & 1)) {
        if (w[i] * w[i] > angTolSqr || Dot(v[i], v[i]) > linTolSqr) {
          sleepTime[i] = 0.0f;
          minSleepTime = 0.0f;
        } else {
          sleepTime[i] += h;
          minSleepTime = Min(minSleepTime, sleepTime[i]);
        }
      }
    }
    if (minSleepTime >= kTimeToSleep && positionSolved) {
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (i < nb && ((ibmask >> i) & 1)) sleepBody(i);
    }
    return false;
  }

  // b2Island::SolveTOI for the mini-island {body b + its touching wall contacts ic[0..nic)} on the staged register island:
  // up to 20 TOI position iterations, velocity constraints initialised WITH restitution and without warm starting, the
  // velocity sweeps for the rest of the step (same bit-safe early exits as islandSolveReg), position integration.
  // Every contact of a TOI island here is (wall, b) - other dynamic bodies are skipped when it is built (no bullets).
#ifndef BLCD_TOI_ONEBODY
#define BLCD_TOI_ONEBODY 0
#endif
  // BLCD_TOI_ONEBODY (per class): the TOI mini-island is staged as what it is - ONE moving body and its wall contacts - on a
  // one-body RegIsland (body b is its body 0, every contact is (wall, proxy 4)): no select chains over the class's NB bodies, the
  // wall side folded out statically, cycle rows of 3 + 4 kRegC words instead of 3 NB + 4 kRegC.  Same arithmetic in the same order
  // (the other bodies of the full-size island never moved: no contact of a TOI island names them).
  static constexpr bool kToi1 = BLCD_TOI_ONEBODY && NB >= 2 && !SCHED;
  using RegT = RegIsland<kToi1 ? 1 : NB, kToi1 ? 0 : NJ, kRegC, kToi1 ? false : kRegLds, kToi1 ? false : kRegCtLds>;
  __device__ __forceinline__ void toiIslandReg(int b, int nic, float h) {
    RegT R;
    if constexpr (kToi1) {
      R.L = nullptr;
      R.C = nullptr;
    } else {
      R.L = regIslandLds();
      R.C = regContactLds();
    }
    R.nc = nic;
    R.nj = 0;
    const int b0 = kToi1 ? 0 : b;            // the moving body's index inside R
    if constexpr (kToi1) {
      R.deadQ = (deadQ & 0xfu) | (((deadQ >> (4 + b)) & 1u) << 4);
      R.setPos(0, BodyPos{c[b], a[b]});
      R.setVel(0, BodyVel{v[b], w[b]});
      R.setMass(0, BodyMass{invMass[b], invI[b], lc[b]});
    } else {
      R.deadQ = deadQ;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        R.setPos(i, BodyPos{c[i], a[i]});
        R.setVel(i, BodyVel{v[i], w[i]});
        R.setMass(i, BodyMass{invMass[i], invI[i], lc[i]});
      }
    }
    Manifold mans[kRegC];
#pragma unroll
    for (int k = 0; k < kRegC; ++k) {
      if (k < nic) {
        int s = ic.get(k);
        auto& c_ = R.ct[k];
        const Manifold m = manGet(s);
        mans[k] = m;
        slotAB(s, &c_.pA, &c_.pB);
        c_.friction = S->pairs[s].friction;
        c_.restitution = S->pairs[s].restitution;
        c_.pointCount = m.pointCount;
        c_.K.ex = c_.K.ey = V2(0.0f, 0.0f);
        c_.normalMass.ex = c_.normalMass.ey = V2(0.0f, 0.0f);
        c_.mtype = m.type;
        c_.mcount = m.pointCount;
        c_.localNormal = m.localNormal;
        c_.localPoint = m.localPoint;
        c_.lp0 = m.points[0].localPoint;
        c_.lp1 = m.points[1].localPoint;
        c_.radiusA = radiusOf(c_.pA);
        c_.radiusB = radiusOf(c_.pB);
        if constexpr (kToi1) c_.pB = 4;    // side B of every contact of a TOI island is the moving body (side A a wall): body 0 of R
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          RPoint& p = c_.points[j];
          p.normalImpulse = 0.0f;
          p.tangentImpulse = 0.0f;
          p.rA = V2(0.0f, 0.0f);
          p.rB = V2(0.0f, 0.0f);
          p.normalMass = 0.0f;
          p.tangentMass = 0.0f;
          p.velocityBias = 0.0f;
        }
      }
    }
#ifdef BLCD_PROF_TOI2
    unsigned long long qa_ = __builtin_amdgcn_s_memtime();
    const bool qrec_ = (int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1;
#endif
    for (int it = 0; it < 20; ++it) {
      float minSeparation = 0.0f;
#pragma unroll
      for (int k = 0; k < kRegC; ++k)
        if (k < nic) minSeparation = R.template positionContact<true>(R.ct[k], minSeparation);
      if (minSeparation >= -1.5f * kLinearSlop) break;
#ifdef BLCD_PROF_TOI2
#endif
    }
#ifdef BLCD_PROF_TOI2
    { unsigned long long qb_ = __builtin_amdgcn_s_memtime(); if (qrec_) prof[3] += qb_ - qa_; }
#endif
    const BodyPos pb0 = R.getPos(b0);
    c0[b] = pb0.c;                                   // "leap of faith to new safe state"
    a0[b] = pb0.a;
#pragma unroll
    for (int k = 0; k < kRegC; ++k)
      if (k < nic) R.initContact(k, R.ct[k], mans[k]);
#ifdef BLCD_PROF_TOI2
    R.velocitySweeps(S->velIters, h, &prof[7]);
    if (qrec_) prof[4] += 1000;   // event rounds (x1000: the tool prints thousands)
#else
    R.velocitySweeps(S->velIters, h);
#endif
    BodyPos pp = R.getPos(b0);
    BodyVel vv = R.getVel(b0);
    {
      Vec2 translation = h * vv.v;
      if (Dot(translation, translation) > kMaxTranslationSquared) {
        float ratio = kMaxTranslation / Length(translation);
        vv.v *= ratio;
      }
      float rotation = h * vv.w;
      if (rotation * rotation > kMaxRotationSquared) {
        float ratio = kMaxRotation / Abs(rotation);
        vv.w *= ratio;
      }
      pp.c += h * vv.v;
      pp.a += h * vv.w;
    }
    c[b] = pp.c;
    a[b] = pp.a;
    v[b] = vv.v;
    w[b] = vv.w;
    syncTransform(b);
  }

  // RESUME: the island was suspended at sweep kYieldSweeps in an earlier pass.  Everything up to the sweeps is recomputed from
  // the untouched world-step-start state (same inputs, same arithmetic, same values - the bias terms need the pre-solve
  // velocities, which is why those stay in place), then the sweeps' own state - island velocities and accumulated impulses - is
  // put back and the loop continues at sweep kYieldSweeps.  Returns true when the island suspends (never when RESUME).
  template <int MODE = 0>   // 0 fresh, 1 resume in the velocity sweeps, 2 resume in the position iterations (staged islands only)
  __device__ __forceinline__ bool islandSolve(uint32_t ibmask, int nic, int nij, float h, float dtRatio, int seed = 0, bool mayYield = false) {
    constexpr bool RESUME = MODE == 1;
    if constexpr (kUseReg) {
      if (MODE == 2 || nic <= kRegC) return islandSolveReg<MODE>(ibmask, nic, nij, h, dtRatio, seed, mayYield);
    }
    const int nb = S->nb;
    Vec2 gravity = S->gravity;
    for (int i = 0; i < NB; ++i) {
      if (i >= nb) break;
      if (!((ibmask >> i) & 1)) continue;
      Vec2 v_ = v[i];
      float w_ = w[i];
      c0[i] = c[i];
      a0[i] = a[i];
      v_ += h * (1.0f * gravity + invMass[i] * V2(0.0f, 0.0f));
      w_ += h * invI[i] * 0.0f;
      v_ *= Clamp(1.0f - h * S->bodies[i].linearDamping, 0.0f, 1.0f);      // Box2D 2.3.0 (first-order; >= 2.3.1 is Pade)
      w_ *= Clamp(1.0f - h * S->bodies[i].angularDamping, 0.0f, 1.0f);
      pc.set(i, c[i]);
      pa.set(i, a[i]);
      pv.set(i, v_);
      pw.set(i, w_);
    }
    csInit(nic, true, dtRatio);
    csInitVelocityConstraints(nic);
    if constexpr (RESUME && kCanYield) {
      for (int i = 0; i < NB; ++i) {
        if (i >= nb) break;
        if (!((ibmask >> i) & 1)) continue;
        const float* q_ = susWords(i);
        pv.set(i, V2(q_[0], q_[(size_t)1 * gN]));
        pw.set(i, q_[(size_t)2 * gN]);
      }
#pragma unroll kU
      for (int k = 0; k < kMaxC; ++k) {
        if (k >= nic) break;
        const Manifold m = manGet(vc[k].slot);
#pragma unroll
        for (int j = 0; j < kMP; ++j) {
          vc[k].points[j].normalImpulse = m.points[j].normalImpulse;     // stored raw at suspension (no dtRatio)
          vc[k].points[j].tangentImpulse = m.points[j].tangentImpulse;
        }
      }
    } else {
      csWarmStart(nic);
    }
    for (int k = 0; k < nij; ++k) jointInit(ij[k], true, dtRatio);
    if (velocitySweeps(ibmask, nic, nij, h, RESUME ? kYieldSweeps : 0, kCanYield && !RESUME && mayYield && nij == 0)) {
      if constexpr (kCanYield) {
        for (int i = 0; i < NB; ++i) {
          if (i >= nb) break;
          if (!((ibmask >> i) & 1)) continue;
          float* q_ = susWords(i);
          const Vec2 pvi_ = pv.get(i);
          q_[0] = pvi_.x;
          q_[(size_t)1 * gN] = pvi_.y;
          q_[(size_t)2 * gN] = pw.get(i);
        }
        csStoreImpulses(nic);
        velMask |= 1u << seed;
      }
      return true;
    }
    csStoreImpulses(nic);
    integratePositions(ibmask, h);
    bool positionSolved = false;
    const int posIters = S->posIters;
    for (int it = 0; it < posIters; ++it) {
      float minSeparation = csSolvePosition(nic, false, -1);
      bool contactsOkay = minSeparation >= -3.0f * kLinearSlop;
      bool jointsOkay = true;
      for (int k = 0; k < nij; ++k) {
        bool jointOkay = jointSolvePosition(ij[k]);
        jointsOkay = jointsOkay && jointOkay;
      }
      if (contactsOkay && jointsOkay) {
        positionSolved = true;
        break;
      }
    }
    for (int i = 0; i < NB; ++i) {
      if (i >= nb) break;
      if (!((ibmask >> i) & 1)) continue;
      c[i] = pc.get(i);
      a[i] = pa.get(i);
      v[i] = pv.get(i);
      w[i] = pw.get(i);
      syncTransform(i);
    }
    float minSleepTime = kMaxFloat;
    const float linTolSqr = kLinearSleepTolerance * kLinearSleepTolerance;
    const float angTolSqr = kAngularSleepTolerance * kAngularSleepTolerance;
    for (int i = 0; i < NB; ++i) {
      if (i >= nb) break;
      if (!((ibmask >> i) & 1)) continue;
      if (w[i] * w[i] > angTolSqr || Dot(v[i], v[i]) > linTolSqr) {
        sleepTime[i] = 0.0f;
        minSleepTime = 0.0f;
      } else {
        sleepTime[i] += h;
        minSleepTime = Min(minSleepTime, sleepTime[i]);
      }
    }
    if (minSleepTime >= kTimeToSleep && positionSolved) {
      for (int i = 0; i < NB; ++i) {
        if (i >= nb) break;
        if ((ibmask >> i) & 1) sleepBody(i);
      }
    }
    return false;
  }

  // island discovery of b2World::Solve from one seed body: depth-first over contact edges (world contact-list order filtered
  // by body) then joint edges; fills ic[0..nic) / ij[0..nij) in island order and marks what it visited
  __device__ __forceinline__ void islandDFS(int seed, uint32_t& bodyIsland, uint32_t& jointIsland, uint32_t& ibmask, int& nic, int& nij) {
    int stack[NB + 4];
    uint32_t wallIsland = 0;
    int sp = 0;
    stack[sp++] = 4 + seed;
    bodyIsland |= 1u << seed;
    while (sp > 0) {
      int p = stack[--sp];
      if (p < 4) continue;  // static bodies join the island but are not expanded
      int b = p - 4;
      ibmask |= 1u << b;
      wake(p);
      for (int k = 0; k < nc; ++k) {  // contact-edge list of b == world list filtered by b (same relative order)
        int s = wl.get(k);
        int pa_ = pairAOf(s), pb_ = pairBOf(s);
        if (pa_ != p && pb_ != p) continue;
        int fl = pflags.get(s);
        if (fl & PF_ISLAND) continue;
        if (!(fl & PF_ENABLED) || !(fl & PF_TOUCHING)) continue;
        ic.set(nic++, s);
        pflags.set(s, fl | PF_ISLAND);
        int other = pa_ == p ? pb_ : pa_;
        if (other < 4) {
          if ((wallIsland >> other) & 1) continue;
          wallIsland |= 1u << other;
          stack[sp++] = other;
        } else {
          if ((bodyIsland >> (other - 4)) & 1) continue;
          bodyIsland |= 1u << (other - 4);
          stack[sp++] = other;
        }
      }
      const DevBody& db = S->bodies[b];
      for (int k = 0; k < db.nJoints; ++k) {
        int j = db.joints[k];
        if ((jointIsland >> j) & 1) continue;
        int other = S->joints[j].bodyA == b ? S->joints[j].bodyB : S->joints[j].bodyA;
        ij[nij++] = (uint8_t)j;
        jointIsland |= 1u << j;
        if ((bodyIsland >> other) & 1) continue;
        bodyIsland |= 1u << other;
        stack[sp++] = 4 + other;
      }
    }
  }

  // b2World::Solve.  Returns true when an island suspended (scheduler kernels only): in its velocity sweeps (velMask) or in its
  // position iterations (posMask).  The rest of the world step - that island's remaining solve, SynchronizeFixtures of every
  // islanded body, FindNewContacts, SolveTOI - is then owed, and solve<true> pays it in a later launch: it re-discovers the
  // suspended islands from their seeds (same seed, same contact list, same flags => same DFS order), resumes each where it
  // stopped, and runs the tail.  A resumed island can suspend again (velocity sweeps -> position iterations).
  template <bool RESUME = false>
  __device__ __forceinline__ bool solve(float h, float dtRatio, bool mayYield = false) {
    if constexpr (NB == 1) {
      // single dynamic body: the only possible island is {body 0} with its touching contacts in contact-list order
      for (int k = 0; k < nc; ++k) pflags.clearBits(wl.get(k), PF_ISLAND);
      if (RESUME || awakeDyn(0)) {
        int nic = 0;
        for (int k = 0; k < nc; ++k) {
          int s = wl.get(k);
          int fl = pflags.get(s);
          if (!(fl & PF_ENABLED) || !(fl & PF_TOUCHING)) continue;
          if (nic == kMaxC) {
            fault |= FAULT_OVERFLOW;
            continue;
          }
          ic.set(nic++, s);
          pflags.set(s, fl | PF_ISLAND);
        }
        velMask = 0;
        if (islandSolve<RESUME ? 1 : 0>(1u, nic, 0, h, dtRatio, 0, mayYield)) {
          islandedMask = 1u;
          return true;
        }
        synchronizeFixtures(0);
      }
      findNewContacts(false);
      return false;
    }
#ifndef BLCD_NO_KILL
    // the generic island's per-body working arrays (register-resident for <= 7 bodies) are written for island members only:
    // value-initialise them here so that last step's values do not stay live through collide and the TOI pass (see csInit)
    if constexpr (!kGenLds && !SCHED) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        pc.a[i] = V2(0.0f, 0.0f);
        pv.a[i] = V2(0.0f, 0.0f);
        pa.a[i] = 0.0f;
        pw.a[i] = 0.0f;
      }
    }
#endif
    const int nb = S->nb;
    uint32_t bodyIsland = 0, jointIsland = 0;
    bool anyYield = false;
    for (int k = 0; k < nc; ++k) pflags.clearBits(wl.get(k), PF_ISLAND);
    if constexpr (RESUME) {
      uint32_t vseeds = velMask, pseeds = posMask;
      velMask = 0;
      posMask = 0;
      while (vseeds | pseeds) {
        const int seed = 31 - __clz((int)(vseeds | pseeds));
        const bool inPos = (pseeds >> seed) & 1;
        vseeds &= ~(1u << seed);
        pseeds &= ~(1u << seed);
        uint32_t ibmask = 0;
        int nic = 0, nij = 0;
        islandDFS(seed, bodyIsland, jointIsland, ibmask, nic, nij);
        if (inPos) anyYield = islandSolve<2>(ibmask, nic, nij, h, dtRatio, seed, false) || anyYield;
        else anyYield = islandSolve<1>(ibmask, nic, nij, h, dtRatio, seed, mayYield) || anyYield;
      }
      bodyIsland = islandedMask;
    } else if constexpr (NJ > 0) {
      // Islands are disjoint, so the ORDER in which they are solved changes nothing - but in a wave it decides whether the
      // lanes' expensive (jointed) islands run in the same loop iteration.  With the seed loop alone, an environment whose
      // free object touches the robot solves the robot at the object's seed and its neighbour lane solves it one seed later:
      // the wave pays the 180 jointed sweeps twice.  So: discover first (each island keeps Box2D's own seed = its first body
      // in body-list order, hence its own DFS order), then solve the jointed islands of all lanes together, then the rest.
      uint32_t jointedSeeds = 0, plainSeeds = 0;
      for (int seed = NB - 1; seed >= 0; --seed) {
        if (seed >= nb) continue;
        if ((bodyIsland >> seed) & 1) continue;
        if (!awakeDyn(seed)) continue;
        uint32_t ibmask = 0;
        int nic = 0, nij = 0;
        islandDFS(seed, bodyIsland, jointIsland, ibmask, nic, nij);
        if (nij > 0) jointedSeeds |= 1u << seed;
        else plainSeeds |= 1u << seed;
      }
      for (int k = 0; k < nc; ++k) pflags.clearBits(wl.get(k), PF_ISLAND);
      const uint32_t allBodies = bodyIsland;
      bodyIsland = 0;
      jointIsland = 0;
      for (int pass = 0; pass < 2; ++pass) {
        uint32_t seeds = pass == 0 ? jointedSeeds : plainSeeds;
        while (seeds) {
          const int seed = 31 - __clz((int)seeds);   // body-list order = descending index
          seeds &= ~(1u << seed);
          uint32_t ibmask = 0;
          int nic = 0, nij = 0;
          islandDFS(seed, bodyIsland, jointIsland, ibmask, nic, nij);
          anyYield = islandSolve(ibmask, nic, nij, h, dtRatio, seed, mayYield) || anyYield;
        }
      }
      bodyIsland = allBodies;
    } else {
      for (int seed = NB - 1; seed >= 0; --seed) {
        if (seed >= nb) continue;
        if ((bodyIsland >> seed) & 1) continue;
        if (!awakeDyn(seed)) continue;
        uint32_t ibmask = 0;
        int nic = 0, nij = 0;
        islandDFS(seed, bodyIsland, jointIsland, ibmask, nic, nij);
        anyYield = islandSolve(ibmask, nic, nij, h, dtRatio, seed, mayYield) || anyYield;
      }
    }
    if (anyYield) {
      if constexpr (!RESUME) islandedMask = bodyIsland;
      return true;
    }
#ifdef BLCD_PROF_SOLVE2
    const unsigned long long psS_ = __builtin_amdgcn_s_memtime();
#endif
    for (int i = NB - 1; i >= 0; --i) {
      if (i >= nb) continue;
      if ((bodyIsland >> i) & 1) synchronizeFixtures(i);
    }
    findNewContacts(false);
#ifdef BLCD_PROF_SOLVE2
    prof[3] += __builtin_amdgcn_s_memtime() - psS_;   // SynchronizeFixtures + FindNewContacts
#endif
    return false;
  }

  // ------------------------------------------------------------------------------------------------
  template <int MAXV>
  static __device__ __forceinline__ void toiWallRun(TOIOutput* out, Vec2 e0, Vec2 e1, float er, const Shape* shB, const Sweep& sw) {
    TOIWall<MAXV, kCirc> tw;
    tw.A.a0 = e0;
    tw.A.a1 = e1;
    tw.A.radius = er;
    tw.B.load(shB);
    tw.run(out, sw);
  }

  // b2World::SolveTOI (+ b2Island::SolveTOI).  Without bullets only dynamic-vs-wall contacts are eligible, so a TOI
  // island is one dynamic body plus the walls it touches at the time of impact.
  // ------------------------------------------------------------------------------------------------
  __device__ __forceinline__ Sweep sweepOf(int p) const {
    Sweep sw;
    if (p < 4) {
      sw.localCenter = V2(0.0f, 0.0f);
      sw.c0 = sw.c = V2(0.0f, 0.0f);
      sw.a0 = sw.a = 0.0f;
      sw.alpha0 = selGet(wallAlpha0, p < 4 ? p : 0);
    } else {
      int i = bi(p);
      sw.localCenter = lc[i];
      sw.c0 = c0[i];
      sw.c = c[i];
      sw.a0 = a0[i];
      sw.a = a[i];
      sw.alpha0 = alpha0[i];
    }
    return sw;
  }
  __device__ __forceinline__ void advanceBody(int i, float alpha) {  // b2Body::Advance
    Sweep sw = sweepOf(4 + i);
    sw.Advance(alpha);
    c0[i] = sw.c0;
    a0[i] = sw.a0;
    alpha0[i] = sw.alpha0;
    c[i] = c0[i];
    a[i] = a0[i];
    q[i] = rotFor(4 + i, a[i]);
    xfp[i] = c[i] - Mul(q[i], lc[i]);
  }

  // mayYield (scheduler kernels): returns true - with NOTHING of what it did so far kept - when there is a TOI event to process
  // and the caller should suspend the environment; the resuming launch calls solveTOI again from the same state, which finds
  // the same event first (everything up to it is a pure function of the state: the TOI cache, flags and counters live only
  // inside one SolveTOI)
  __device__ __forceinline__ bool solveTOI(float dt, bool mayYield = false) {
#ifdef BLCD_PROF_TOI
#define PT(k_) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof[k_] += n_ - pt_; pt_ = n_; } while (0)
    unsigned long long pt_ = __builtin_amdgcn_s_memtime();
#else
#define PT(k_) do {} while (0)
#endif
    const int nb = S->nb;
    for (int i = 0; i < NB; ++i)
      if (i < nb) alpha0[i] = 0.0f;
    for (int k = 0; k < 4; ++k) wallAlpha0[k] = 0.0f;
    for (int k = 0; k < nc; ++k) {
      int s = wl.get(k);
      pflags.clearBits(s, PF_TOI | PF_ISLAND | PF_TOISKIP);
      toiCount.set(s, 0);
      selSet(toi, s, (float)(1.0f));
    }
    PT(0);
    for (int guard = 0; guard < 64 * (NP + 1); ++guard) {
      for (int k = 0; k < nc; ++k) {
        int s = wl.get(k);
        if (pflags.get(s) & PF_TOISKIP) pflags.clearBits(s, PF_TOI | PF_TOISKIP);
      }
      // Phase 1: every contact whose TOI is not cached gets it computed, in contact-list order per environment.  Each
      // lane jumps straight to ITS next pending contact, so a wave runs the (long) TOI routine max-over-lanes(#pending)
      // times instead of once per list position at which any lane is pending.  Same per-environment operation order as
      // the reference's single loop: cached entries are only read, and they are read in phase 2.
      // (a cursor per lane: everything in front of it has its verdict for this pass - nothing in phase 1 clears PF_TOI - so the
      // search for the next pending contact resumes where the last one ended instead of rescanning the list: nc list reads per
      // pass instead of nc^2 / 2)
      int kNext = 0;
      for (;;) {
        int s = -1, fl = 0;
        while (kNext < nc) {
          const int s2 = wl.get(kNext);
          const int f2 = pflags.get(s2);
          ++kNext;
          if ((f2 & PF_ENABLED) && !(f2 & PF_TOI) && toiCount.get(s2) <= kMaxSubSteps) {
            s = s2;
            fl = f2;
            break;
          }
        }
        if (s < 0) break;
        float alpha = 1.0f;
        int pa_ = pairAOf(s), pb_ = pairBOf(s);
        int b = pa_ >= 4 ? 0 : bi(pb_);
        if (pa_ >= 4 || !awakeDyn(b)) {
          // two non-bullet dynamic bodies, or nothing awake: the reference skips the contact without caching anything.
          // The verdict is remembered for THIS pass only (so the scan moves on) and dropped before the next pass.
          selSet(toi, s, 1.0f);
          pflags.set(s, fl | PF_TOI | PF_TOISKIP);
          continue;
        }
        // put the sweeps onto the same time interval (the wall's alpha0 is part of the state, see b2World::SolveTOI)
        float alpha0_ = selGet(wallAlpha0, pa_);
        if (alpha0_ < alpha0[b]) {
          alpha0_ = alpha0[b];
          selSet(wallAlpha0, pa_, alpha0_);  // wall sweep Advance: c0 = c = 0 stays, alpha0 moves
        } else if (alpha0[b] < alpha0_) {
          Sweep sw = sweepOf(pb_);
          sw.Advance(alpha0_);
          c0[b] = sw.c0;
          a0[b] = sw.a0;
          alpha0[b] = sw.alpha0;
        }
        TOIOutput output;
        // Exact early-out for circles centred on their body origin (deadQ): b2TimeOfImpact's first iteration returns
        // e_separated, t = tMax when (1) b2Distance at t1 = 0 is >= target + tolerance and (2) the separation at tMax along
        // the axis it then picks is > target + tolerance.  For a one-vertex proxy against a wall whose segment the centre
        // projects well inside, GJK ends on the face (both edge vertices), the axis is the wall normal, and both numbers are
        // the centre's signed distance to the wall line at c0 and at c.  If both clear the threshold by a margin (5e-4,
        // ~500x the float error of either evaluation) on the same side, the routine's answer is known without running it.
        bool knownSeparated = false;
        if ((deadQ >> pb_) & 1u) {
          const Vec2 w0 = selGet(wallV0, pa_);
          const float totalRadius = selGet(wallRad, pa_) + radiusOf(pb_);
          const float thr = Max(kLinearSlop, totalRadius - 3.0f * kLinearSlop) + 0.25f * kLinearSlop + 5.0e-4f;
          const Vec2 nrm = selGet(wallNrm, pa_), tan_ = selGet(wallTan, pa_);   // per-launch constants (see load)
          const Vec2 r0 = c0[b] - w0, r1 = c[b] - w0;
          const float d0 = -Dot(r0, nrm), d1 = -Dot(r1, nrm);     // signed distances to the wall line (either orientation)
          const float u = Dot(r0, tan_);                           // projection parameter of the start point on the segment
          knownSeparated = u > 0.01f && u < 0.99f && ((d0 > thr && d1 > thr) || (d0 < -thr && d1 < -thr));
        } else if constexpr (!kCirc) {
          // Exact early-out for every other shape (polygons, circles that carry joints), by a lower bound instead of the
          // routine's own numbers.  b2TimeOfImpact can only answer e_touching (alpha < 1) at a time t1 at which either the
          // b2Distance of the core shapes or the separation of a vertex pair along the axis built AT t1 is below
          // target + tolerance; that axis separates the shapes at t1, so both numbers are >= the true distance at t1, which
          // is >= the distance of the moving vertices to the wall's LINE.  A vertex v moves as c(t) + R(a(t)) (v - lc) with c, a
          // linear in t: its signed distance is a chord plus an arc term whose second derivative is bounded by |r| da^2, so
          // it stays above min(D(0), D(1)) - |r| da^2 / 8.  If that bound clears the threshold for every vertex (margin 5e-4
          // + 2e-5 for the fast sin/cos of da used here) the answer is "not touching": alpha = 1, nothing else changes.
          const Sweep sw = sweepOf(pb_);
          const float da = sw.a - sw.a0;
          if (da > -1.0f && da < 1.0f) {
            const Vec2 w0 = selGet(wallV0, pa_);
            const Vec2 nrm = selGet(wallNrm, pa_);
            const Shape* shB = shapeOf(pb_);
            const float totalRadius = selGet(wallRad, pa_) + shB->radius;
            const float thr = Max(kLinearSlop, totalRadius - 3.0f * kLinearSlop) + 0.25f * kLinearSlop + 5.0e-4f + 2.0e-5f;
            const Rot q1 = q[b];
            const float sd = __sinf(da), cd = __cosf(da);
            Rot q0;                                          // R(a0) = R(a) R(-da)
            q0.c = q1.c * cd + q1.s * sd;
            q0.s = q1.s * cd - q1.c * sd;
            const float lin0 = Dot(sw.c0 - w0, nrm), lin1 = Dot(sw.c - w0, nrm);
            const float sgn = lin1 >= 0.0f ? 1.0f : -1.0f;   // the side the body ends on; a crossing fails the test below
            const float curv = 0.125f * da * da;
            const int nv = shB->type == kCircle ? 1 : shB->count;
            float lowest = kMaxFloat;
#pragma unroll
            for (int k2 = 0; k2 < kShapeVerts; ++k2) {
              if (k2 >= nv) break;
              const Vec2 r = shB->v[k2] - sw.localCenter;
              const float D0 = sgn * (lin0 + Dot(Mul(q0, r), nrm));
              const float D1 = sgn * (lin1 + Dot(Mul(q1, r), nrm));
              lowest = Min(lowest, Min(D0, D1));
            }
            // the arc term with the largest vertex radius of the shape (one number per body, taken at load) instead of |r| per
            // vertex: a slightly weaker bound (still a lower bound), eight correctly rounded square roots fewer per test
            knownSeparated = lowest - curv * rmaxV[b] > thr;
          }
        }
        if (knownSeparated) {
          output.state = kTOISeparated;
          output.t = 1.0f;
        } else {
          unsigned long long tq0 = profOn ? __builtin_amdgcn_s_memtime() : 0;
          // wall edge vs moving shape, everything in registers (blcd_toi_wall.h); the proxy width follows the shape so
          // that circles and boxes do not pay for 8-vertex select chains
          const Vec2 e0 = selGet(wallV0, pa_), e1 = selGet(wallV1, pa_);
          const float er = selGet(wallRad, pa_);
          if (kCirc) {
            const Shape cB = circShapeReg(b);
            toiWallRun<1>(&output, e0, e1, er, &cB, sweepOf(pb_));
          } else {
            const Shape* shB = shapeOf(pb_);
            const int nv = shB->type == kCircle ? 1 : shB->count;
            if (nv == 1) toiWallRun<1>(&output, e0, e1, er, shB, sweepOf(pb_));
            else if (nv <= 4) toiWallRun<4>(&output, e0, e1, er, shB, sweepOf(pb_));
            else toiWallRun<kShapeVerts>(&output, e0, e1, er, shB, sweepOf(pb_));
          }
#if !defined(BLCD_PROF_TOI) && !defined(BLCD_PROF_SOLVE) && !defined(BLCD_PROF_TOI2)
          prof[4] += 1;
#endif
#ifndef BLCD_PROF_TOI2
          if (profOn && (int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) {  // once per wave-level execution
            prof[6] += __builtin_amdgcn_s_memtime() - tq0;
            prof[7] += 1;
          }
#endif
        }
        float beta = output.t;
        if (output.state == kTOITouching) alpha = Min(alpha0_ + (1.0f - alpha0_) * beta, 1.0f);
        else alpha = 1.0f;
        selSet(toi, s, (float)(alpha));
        pflags.set(s, fl | PF_TOI);
      }
      PT(1);
      // Phase 2: the minimum over the contact list (first minimum wins, as in the reference's `alpha < minAlpha`)
      int minSlot = -1;
      float minAlpha = 1.0f;
      for (int k = 0; k < nc; ++k) {
        int s = wl.get(k);
        int fl = pflags.get(s);
        if (!(fl & PF_ENABLED)) continue;
        if (toiCount.get(s) > kMaxSubSteps) continue;
        if (fl & PF_TOISKIP) continue;
        float alpha = selGet(toi, s);
        if (alpha < minAlpha) {
          minSlot = s;
          minAlpha = alpha;
        }
      }
      PT(4);
      if (minSlot < 0 || 1.0f - 10.0f * kEpsilon < minAlpha) break;
      if constexpr (kCanYield) {
        if (mayYield && guard == 0 && __popcll(__ballot(1)) <= yieldMaxLanes) return true;
      }

#ifdef BLCD_PROF_TOI2
#define QT(k_) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) prof[k_] += n_ - qt_; qt_ = n_; } while (0)
      unsigned long long qt_ = __builtin_amdgcn_s_memtime();
      if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) prof[7] += 1;
#else
#define QT(k_) do {} while (0)
#endif
      const int wA = pairAOf(minSlot);      // wall (fixture A)
      const int pB = pairBOf(minSlot);
      const int b = bi(pB);
      // backups
      float backupWallAlpha = selGet(wallAlpha0, wA);
      Vec2 bc0 = c0[b], bc = c[b];
      float ba0 = a0[b], ba = a[b], balpha0 = alpha0[b];
      // bA->Advance(minAlpha) for the wall, bB->Advance(minAlpha)
      selSet(wallAlpha0, wA, minAlpha);
      advanceBody(b, minAlpha);
      updateContact(minSlot);
      pflags.clearBits(minSlot, PF_TOI);
      toiCount.set(minSlot, toiCount.get(minSlot) + 1);
      if (!(pflags.get(minSlot) & PF_ENABLED) || !(pflags.get(minSlot) & PF_TOUCHING)) {
        pflags.clearBits(minSlot, PF_ENABLED);
        selSet(wallAlpha0, wA, backupWallAlpha);
        c0[b] = bc0;
        c[b] = bc;
        a0[b] = ba0;
        a[b] = ba;
        alpha0[b] = balpha0;
        syncTransform(b);
        continue;
      }
      wake(pB);
      QT(0);
#if !defined(BLCD_PROF_TOI) && !defined(BLCD_PROF_SOLVE) && !defined(BLCD_PROF_TOI2)
      prof[5] += 1;
#endif
      unsigned long long e0_ = profOn ? __builtin_amdgcn_s_memtime() : 0;
      // build the TOI island: contact list of the dynamic body, static others only
      uint32_t wallIsland = 1u << wA;
      int nic = 0;
      ic.set(nic++, minSlot);
      pflags.orBits(minSlot, PF_ISLAND);
      for (int k = 0; k < nc; ++k) {
        if (nic == kMaxTOIContacts) break;
        int s = wl.get(k);
        int pa_ = pairAOf(s), pb_ = pairBOf(s);
        if (pa_ != pB && pb_ != pB) continue;
        if (pflags.get(s) & PF_ISLAND) continue;
        int other = pa_ == pB ? pb_ : pa_;
        if (other >= 4) continue;  // only static (no bullets)
        float backup = selGet(wallAlpha0, other);
        if (!((wallIsland >> other) & 1)) selSet(wallAlpha0, other, minAlpha);  // other->Advance(minAlpha)
        updateContact(s);
        if (!(pflags.get(s) & PF_ENABLED) || !(pflags.get(s) & PF_TOUCHING)) {
          selSet(wallAlpha0, other, backup);
          continue;
        }
        if (nic == kMaxC) {  // only reachable in the reduced-capacity one-body configuration
          fault |= FAULT_OVERFLOW;
          selSet(wallAlpha0, other, backup);
          continue;
        }
        pflags.orBits(s, PF_ISLAND);
        ic.set(nic++, s);
        wallIsland |= 1u << other;
      }
      QT(1);
      // b2Island::SolveTOI
      float h = (1.0f - minAlpha) * dt;
      bool toiDone = false;
      if constexpr (kUseReg) {
        if (nic <= kRegC) {
          QT(4);
          toiIslandReg(b, nic, h);
          QT(5);
          toiDone = true;
        }
      }
      if (!toiDone) {
      pc.set(b, c[b]);
      pa.set(b, a[b]);
      pv.set(b, v[b]);
      pw.set(b, w[b]);
      csInit(nic, false, 1.0f);
      for (int it = 0; it < 20; ++it) {
        float minSeparation = csSolvePosition(nic, true, pB);
        if (minSeparation >= -1.5f * kLinearSlop) break;
      }
      c0[b] = pc.get(b);
      a0[b] = pa.get(b);
      QT(4);
      csInitVelocityConstraints(nic);
      velocitySweeps(1u << b, nic, 0, h);
      QT(5);
      integratePositions(1u << b, h);
      c[b] = pc.get(b);
      a[b] = pa.get(b);
      v[b] = pv.get(b);
      w[b] = pw.get(b);
      syncTransform(b);
      }
      // reset island flags, synchronize the broad phase, invalidate the body's contact TOIs
      synchronizeFixtures(b);
      for (int k = 0; k < nc; ++k) {
        int s = wl.get(k);
        if (pairAOf(s) == pB || pairBOf(s) == pB) pflags.clearBits(s, PF_TOI | PF_ISLAND | PF_TOISKIP);
      }
      findNewContacts(false);
      QT(6);
#ifndef BLCD_PROF_TOI2
#ifndef BLCD_PROF_SOLVE2
      if (profOn) prof[3] += __builtin_amdgcn_s_memtime() - e0_;
#endif
#endif
      PT(5);
    }
    PT(5);
#undef PT
#undef QT
    return false;
  }

  // b2World::Step
  // the owed part of a suspended world step (see solve / solveTOI); returns true when it suspends again at a later point
  __device__ __forceinline__ bool worldStepResume(bool mayYield) {
    const float dt = S->dt;
    const float inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
    if (velMask | posMask) {
      if (solve<true>(dt, inv_dt0 * dt, mayYield)) return true;
      islandedMask = 0;
    }
    const bool atToi = toiPending;   // resumed AT its TOI event: that event is processed now, whatever the wave looks like
    toiPending = false;
    if (solveTOI(dt, mayYield && !atToi)) {
      toiPending = true;
      return true;
    }
    inv_dt0 = inv_dt;
    return false;
  }

  // returns true when the environment suspended (mayYield only; the caller resumes it with worldStepResume in a later pass)
  __device__ __forceinline__ bool worldStep(bool mayYield = false) {
    if (wflags & WF_NEWFIXTURE) {
      findNewContacts(true);
      wflags &= ~WF_NEWFIXTURE;
    }
    float dt = S->dt;
    float inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
    float dtRatio = inv_dt0 * dt;
#if defined(BLCD_CT_SKIP)
    const int skip = BLCD_CT_SKIP;   // compile-time phase removal: register-pressure experiments only
#elif defined(BLCD_ABLATION)
    const int skip = S->dbgSkip;     // run-time phase removal (BLCD_DEBUG_SKIP): ablation builds only, never the shipped library
#else
    constexpr int skip = 0;
#endif
    unsigned long long c0_ = profOn ? __builtin_amdgcn_s_memtime() : 0;
    if (!(skip & 1)) collide();
    unsigned long long c1_ = profOn ? __builtin_amdgcn_s_memtime() : 0;
    if (!(skip & 2)) {
      if (solve(dt, dtRatio, mayYield)) return true;
    }
    unsigned long long c2_ = profOn ? __builtin_amdgcn_s_memtime() : 0;
    if (!(skip & 4)) {
      if (solveTOI(dt, mayYield)) {
        toiPending = true;
        return true;
      }
    }
    if (profOn) {
      unsigned long long c3_ = __builtin_amdgcn_s_memtime();
#if !defined(BLCD_PROF_TOI) && !defined(BLCD_PROF_TOI2)
      prof[0] += c1_ - c0_;
      prof[1] += c2_ - c1_;
#endif
      prof[2] += c3_ - c2_;
    }
    inv_dt0 = inv_dt;
    return false;
  }

  // b2World::Step is the identity on an environment whose bodies are all asleep: b2ContactManager::Collide skips contacts without
  // an active body, b2World::Solve seeds islands from awake bodies only (none: no velocity, position or sleep-timer update, no
  // SynchronizeFixtures, an empty move buffer for FindNewContacts), b2World::SolveTOI skips contacts without an awake body.  Of
  // the stored state only m_inv_dt0 is assigned, and it already holds this step's value once one step has been taken; a pending
  // e_newFixture pass (right after a reset / state injection) is excluded.  Flag bits that a step toggles on the way
  // (island / TOI marks) are not part of the stored state.
  __device__ __forceinline__ bool atRest() const {
    const float dt = S->dt;
    const float inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
    return awakeMask == 0 && !(wflags & WF_NEWFIXTURE) && inv_dt0 == inv_dt && fault == 0;
  }

  // action -> joint.motorSpeed (boxLCD/utils.py:117 mapto, world_env.py:441); b2RevoluteJoint::SetMotorSpeed wakes both bodies
  __device__ __forceinline__ void setMotorSpeeds(const float* __restrict__ actions, int N, int e) {
    const int nj = S->nj, nact = S->nact;
    for (int j = 0; j < NJ; ++j) {
      if (j >= nj) break;
      const DevJoint& J = S->joints[j];
      if (J.actionIndex < 0) continue;
      double act = actions ? (double)actions[(size_t)e * nact + J.actionIndex] : 0.0;
      double m = ((act + 1.0) / 2.0 * (double)(1 - (-1))) + (double)(-1);
      double cl = m < -1.0 ? -1.0 : (m > 1.0 ? 1.0 : m);
      jspeed[j] = (float)((double)J.speed * cl);
      wake(4 + J.bodyA);
      wake(4 + J.bodyB);
    }
  }

  __device__ __forceinline__ void checkFault() {
    const int nb = S->nb;
    for (int i = 0; i < NB; ++i) {
      if (i >= nb) break;
      const Vec2 cc_ = c[i], vv_ = v[i];
      float t = cc_.x + cc_.y + a[i] + vv_.x + vv_.y + w[i];
      if (!(t == t) || Abs(t) > 3.0e38f) fault |= FAULT_NAN;
    }
  }
};

}  // namespace blcd
