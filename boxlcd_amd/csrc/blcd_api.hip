// blcd_api.hip — kernels and the C ABI (include/boxlcd.h) of libboxlcd_hip.so.  gfx950 only, no CPU fallback.
//
// Host side: scene lowering (hull/centroid/mass data/pair-slot table: b2PolygonShape::Set, ComputeMass,
// b2Body::ResetMassData, the static half of b2ContactManager::AddPair), device-state ownership, staging of host
// buffers, launches on the handle's HIP stream, hipEvent timing of the step launches.
// Device side: one thread per environment (see blcd_world.h / blcd_raster.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/boxlcd.h"
#include "blcd_emit.h"
#include "blcd_cfg_launch.h"
#include "blcd_render_ex.h"
#include <vector>
#include <mutex>

using namespace blcd;

// ---------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess) return fail(BLCD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// ---------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------

// world construction for the listed envs: b2World::CreateBody + CreateFixture (proxy: tight AABB +- aabbExtension, buffered
// as moved, e_newFixture), revolute joints with referenceAngle = bodyB.angle - bodyA.angle.
__global__ void reset_kernel(const DevScene* __restrict__ S, float* __restrict__ st, int N, const int* __restrict__ slotOf,
                             const int* __restrict__ idxs, int n, const float* __restrict__ poses,
                             const int* __restrict__ shapeSel) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int e = idxs ? idxs[k] : k;
  if (e < 0 || e >= N) return;
  e = slotOf[e];  // from here on `e` is the slot that stores this environment
  const int nb = S->nb, nj = S->nj, np = S->np;
  float ang[BLCD_MAX_BODIES];
  for (int i = 0; i < nb; ++i) {
    const float* p = poses + ((size_t)k * nb + i) * 3;
    int sel = shapeSel ? shapeSel[(size_t)k * nb + i] : 0;
    if (sel < 0 || sel >= S->bodies[i].nChoices) sel = 0;
    const DevVariant& var = S->bodies[i].var[sel];
    Transform xf;
    xf.p = V2(p[0], p[1]);
    xf.q.Set(p[2]);
    ang[i] = p[2];
    Vec2 c = Mul(xf, var.localCenter);  // ResetMassData: sweep.c0 = sweep.c = b2Mul(xf, localCenter)
    AABB aabb;
    ShapeComputeAABB(&S->shapes[var.shape], &aabb, xf);
    float* o = st + (size_t)(i * kBodyFields) * N + e;
    o[0] = c.x;
    o[(size_t)1 * N] = c.y;
    o[(size_t)2 * N] = p[2];
    o[(size_t)3 * N] = 0.0f;
    o[(size_t)4 * N] = 0.0f;
    o[(size_t)5 * N] = 0.0f;
    o[(size_t)6 * N] = c.x;
    o[(size_t)7 * N] = c.y;
    o[(size_t)8 * N] = p[2];
    o[(size_t)9 * N] = xf.p.x;
    o[(size_t)10 * N] = xf.p.y;
    o[(size_t)11 * N] = 0.0f;
    o[(size_t)12 * N] = 1.0f;
    o[(size_t)13 * N] = aabb.lo.x - kAabbExtension;
    o[(size_t)14 * N] = aabb.lo.y - kAabbExtension;
    o[(size_t)15 * N] = aabb.hi.x + kAabbExtension;
    o[(size_t)16 * N] = aabb.hi.y + kAabbExtension;
    o[(size_t)17 * N] = __int_as_float(sel);
  }
  float* pp = st + (size_t)(nb * kBodyFields) * N + e;
  for (int s = 0; s < np * kPairFields; ++s) pp[(size_t)s * N] = 0.0f;
  float* jp = pp + (size_t)(np * kPairFields) * N;
  for (int j = 0; j < nj; ++j) {
    float* o = jp + (size_t)(j * kJointFields) * N;
    for (int f = 0; f < 6; ++f) o[(size_t)f * N] = 0.0f;
    o[(size_t)6 * N] = ang[S->joints[j].bodyB] - ang[S->joints[j].bodyA];
  }
  float* wp = jp + (size_t)(nj * kJointFields) * N;
  wp[0] = 0.0f;                                                   // m_inv_dt0
  wp[(size_t)1 * N] = __uint_as_float(nb >= 32 ? 0xffffffffu : ((1u << nb) - 1u));  // every proxy is in the move buffer
  wp[(size_t)2 * N] = __uint_as_float((uint32_t)WF_NEWFIXTURE);
  wp[(size_t)3 * N] = __int_as_float(0);
  for (int k2 = 0; k2 < (np + 3) / 4; ++k2) wp[(size_t)(4 + k2) * N] = 0.0f;
}

template <int H, typename RowT, typename ObsT>
__global__ __launch_bounds__(kBlock) void obs_kernel(const DevScene* __restrict__ S, const float* st, int N,
                                                     const int* __restrict__ eid, ObsT* __restrict__ obs,
                                                     uint8_t* __restrict__ lcd, float* stw) {
  int e = blockIdx.x * kBlock + threadIdx.x;  // slot
  if (e >= N) return;
  const int env = eid[e];
  const int nb = S->nb;
  auto body = [&](int i, Vec2* p, float* a, int* sel) {
    const float* q = st + (size_t)(i * kBodyFields) * N + e;
    *p = V2(q[(size_t)9 * N], q[(size_t)10 * N]);
    *a = q[(size_t)2 * N];
    *sel = __float_as_int(q[(size_t)17 * N]);
  };
  bool ok = emit_env<H, RowT, ObsT>(S, body, obs ? obs + (size_t)env * S->nobs : nullptr,
                                    lcd ? lcd + (size_t)env * H * S->lcdW : nullptr);
  if (!ok && stw) {
    size_t off = (size_t)(nb * kBodyFields + S->np * kPairFields + S->nj * kJointFields + 2) * N + e;
    stw[off] = __uint_as_float(__float_as_uint(stw[off]) | ((uint32_t)FAULT_ELLIPSE << 8));
  }
}

template <int H, typename RowT>
__global__ void render_poses_kernel(const DevScene* __restrict__ S, int m, const float* __restrict__ poses,
                                    const int* __restrict__ shapeSel, uint8_t* __restrict__ lcd) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m) return;
  const int nb = S->nb;
  Raster<H, RowT> r;
  r.clear(S->lcdW, S->rasterVariant);
  for (int i = 0; i < nb; ++i) {
    const float* p = poses + ((size_t)k * nb + i) * 3;
    Transform xf;
    xf.p = V2(p[0], p[1]);
    xf.q.Set(p[2]);
    int sel = shapeSel ? shapeSel[(size_t)k * nb + i] : 0;
    if (sel < 0 || sel >= S->bodies[i].nChoices) sel = 0;
    r.drawBody(&S->shapes[S->bodies[i].var[sel].shape], xf, (double)S->worldW, (double)S->lcdW);
  }
  r.write(lcd + (size_t)k * H * S->lcdW);
}

__global__ void poses_kernel(const DevScene* __restrict__ S, const float* __restrict__ st, int N, const int* __restrict__ eid,
                             float* __restrict__ out) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const int env = eid[e];
  for (int i = 0; i < S->nb; ++i) {
    const float* p = st + (size_t)(i * kBodyFields) * N + e;
    float* o = out + ((size_t)env * S->nb + i) * 4;
    o[0] = p[(size_t)9 * N];
    o[1] = p[(size_t)10 * N];
    o[2] = p[(size_t)2 * N];
    o[3] = p[(size_t)12 * N];
  }
}

__global__ void shape_sel_kernel(const DevScene* __restrict__ S, const float* __restrict__ st, int N, const int* __restrict__ eid,
                                 int* __restrict__ out) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const int env = eid[e];
  for (int i = 0; i < S->nb; ++i) out[(size_t)env * S->nb + i] = __float_as_int(st[(size_t)(i * kBodyFields + 17) * N + e]);
}

__global__ void faults_kernel(const DevScene* __restrict__ S, const float* __restrict__ st, int N, const int* __restrict__ eid,
                              int* __restrict__ out) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  size_t off = (size_t)(S->nb * kBodyFields + S->np * kPairFields + S->nj * kJointFields + 2) * N + e;
  out[eid[e]] = (int)(__float_as_uint(st[off]) >> 8);
}

// canonical dump (same layout as the parity oracle's dump)
__global__ void dump_kernel(const DevScene* __restrict__ S, const float* __restrict__ st, int N, const int* __restrict__ eid,
                            float* __restrict__ bodies, float* __restrict__ joints, float* __restrict__ pairs) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  const int env = eid[e];
  const int nb = S->nb, nj = S->nj, np = S->np;
  const int bmap[12] = {0, 1, 2, 3, 4, 5, 11, 12, 13, 14, 15, 16};
  for (int i = 0; i < nb; ++i) {
    const float* p = st + (size_t)(i * kBodyFields) * N + e;
    float* o = bodies + ((size_t)env * nb + i) * BLCD_BODY_STATE_FLOATS;
    for (int f = 0; f < 12; ++f) o[f] = p[(size_t)bmap[f] * N];
  }
  const float* pp = st + (size_t)(nb * kBodyFields) * N + e;
  for (int s = 0; s < np; ++s) {
    const float* p = pp + (size_t)(s * kPairFields) * N;
    float* o = pairs + ((size_t)env * np + s) * BLCD_PAIR_STATE_FLOATS;
    for (int f = 0; f < BLCD_PAIR_STATE_FLOATS; ++f) o[f] = 0.0f;
    int fl = __float_as_int(p[0]);
    if (!(fl & PF_EXISTS)) continue;
    int tc = __float_as_int(p[(size_t)1 * N]);
    int cnt = tc >> 8;
    o[0] = 1.0f;
    o[1] = (fl & PF_TOUCHING) ? 1.0f : 0.0f;
    o[3] = (float)cnt;
    if (cnt > 0) {
      o[2] = (float)(tc & 0xff);
      for (int f = 0; f < 4; ++f) o[4 + f] = p[(size_t)(2 + f) * N];
      for (int k = 0; k < cnt; ++k) {
        const float* r = p + (size_t)(6 + 5 * k) * N;
        for (int f = 0; f < 4; ++f) o[8 + 4 * k + f] = r[(size_t)f * N];
        uint32_t key = __float_as_uint(r[(size_t)4 * N]);
        o[16 + k] = (float)((key & 0xff) + 16 * ((key >> 8) & 0xff) + 256 * ((key >> 16) & 0xff) + 512 * ((key >> 24) & 0xff));
      }
    }
  }
  const float* jp = pp + (size_t)(np * kPairFields) * N;
  for (int j = 0; j < nj; ++j) {
    const float* p = jp + (size_t)(j * kJointFields) * N;
    float* o = joints + ((size_t)env * nj + j) * BLCD_JOINT_STATE_FLOATS;
    o[0] = p[0];
    o[1] = p[(size_t)1 * N];
    o[2] = p[(size_t)2 * N];
    o[3] = p[(size_t)3 * N];
    o[4] = (float)__float_as_int(p[(size_t)4 * N]);
  }
}


// ---------------------------------------------------------------------------------------------------------
// Re-binning: environments are independent, so WHERE an env's state lives is free.  Waves are 64 consecutive slots;
// mixing asleep / free-flight / near-wall / touching envs in one wave leaves ~5 of 64 lanes active (measured, SQ PMC).
// A stable counting sort of slots by work class makes waves homogeneous.  Stable => deterministic.
//   work 0 asleep | 1 awake, no contact slot | 2 awake, contact slots, none touching | 3.. touching (by count)
// The sort key is kBins-1-work: the heaviest slots come FIRST.  Workgroups are dispatched in blockIdx order and a launch
// has ~1.5x more waves than the chip has wave slots (occupancy 1), so heavy-first is longest-processing-time-first
// scheduling: light waves backfill behind the heavy ones instead of the heavy ones starting last and forming the tail.
// ---------------------------------------------------------------------------------------------------------
// One-body scenes (Bounce, Dropbox) are additionally sorted by WHEN the body will next reach a wall: the expensive phases
// (time of impact + TOI sub-step) run for a whole wave whenever one lane needs them, so a wave whose 64 environments hit a
// wall at the same world step pays for them once instead of at almost every step.  The prediction integrates the free
// flight with the solver's own semi-implicit Euler step (no damping, bounding circle); it only decides placement, never
// results.  Keys: 0..31 = impact predicted in that many world steps, 32 = none within 32 steps, 33 = resting on a wall,
// 40..46 = generic work classes (heaviest first), 47 = asleep.
constexpr int kBins = 51;   // + bins 0..2 (ahead of everything): environments suspended at a TOI event / in position iterations / in velocity sweeps
constexpr int kPredictSteps = 32;
constexpr int kRebinBlock = 256;
// ints between two cohorts' count / offset tables (a multiple of four: the scan reads and writes int4)
static inline size_t binTableStride(int nBlocksAll) { return ((size_t)nBlocksAll * kBins + 3) & ~(size_t)3; }

__device__ inline int work_class1(const DevScene* __restrict__ S, const float* __restrict__ st, int N, int slot, int mode);
__device__ inline int work_class(const DevScene* __restrict__ S, const float* __restrict__ st, int N, int slot, int mode, int tEnd) {
  // suspended environments first and together, by what they are suspended at: the launch that resumes them then runs what they
  // owe in dense, homogeneous waves; environments that have finished the rollout (tEnd > 0) go last
#ifdef BLCD_SCHED
  const size_t po = (size_t)schedWordOffset(S->nb, S->nj, S->np);
  const uint32_t prog0 = __float_as_uint(st[po * N + slot]), prog1 = __float_as_uint(st[(po + 1) * N + slot]);
  if ((prog0 >> 18) & 1u) return 0;
  if ((prog1 >> 7) & 0x7fu) return 1;
  if (prog1 & 0x7fu) return 2;
  if (tEnd > 0 && (int)(prog0 & 0xffffu) >= tEnd) return kBins - 1;
#else
  (void)tEnd;
#endif
  if (mode < 0) return 3;       // batches that are not sorted by work class: only the suspended ones move (stable sort)
  return 3 + work_class1(S, st, N, slot, mode);
}
__device__ inline int work_class1(const DevScene* __restrict__ S, const float* __restrict__ st, int N, int slot, int mode) {
  constexpr int kAll = kBins;          // this function's own key space: 3 + (0 .. kAll - 4)
  constexpr int kBins = kAll - 3;
  const int nb = S->nb, nj = S->nj, np = S->np;
  bool anyAwake = false;
  for (int i = 0; i < nb; ++i) anyAwake = anyAwake || st[(size_t)(i * kBodyFields + 12) * N + slot] != 0.0f;
  if (!anyAwake) return kBins - 1;
  const float* pp = st + (size_t)(nb * kBodyFields) * N + slot;
  int touching = 0;
  for (int s2 = 0; s2 < np; ++s2)
    if (__float_as_int(pp[(size_t)(s2 * kPairFields) * N]) & PF_TOUCHING) ++touching;
  if (nb == 1 && nj == 0 && mode != 0) {
    const float* bp = st + slot;
    Vec2 p = V2(bp[0], bp[(size_t)1 * N]);
    Vec2 v = V2(bp[(size_t)3 * N], bp[(size_t)4 * N]);
    const int sel = __float_as_int(bp[(size_t)17 * N]);
    const Shape& sh = S->shapes[S->bodies[0].var[sel].shape];
    float R = sh.radius;
    if (sh.type == kCircle) {
      R += Length(sh.v[0]);
    } else {
      float m = 0.0f;
      for (int i = 0; i < sh.count; ++i) m = Max(m, Length(sh.v[i] - S->bodies[0].var[sel].localCenter));
      R += m;
    }
    const float slack = 0.02f, W = S->worldW, H = S->worldH, dt = S->dt;
    if (touching > 0 && (mode == 2 || Dot(v, v) < 0.25f)) return kPredictSteps + 1;   // resting / rolling: the TOI early-out handles it
    int n = 0;
    for (; n < kPredictSteps; ++n) {
      v += dt * S->gravity;
      p += dt * v;
      if (p.x - R < slack || p.x + R > W - slack || p.y - R < slack || p.y + R > H - slack) break;
    }
    return n;   // 0..31, or 32 = no wall within the horizon
  }
  int nc = __float_as_int(pp[(size_t)(np * kPairFields + nj * kJointFields + 3) * N]);
  if (nc == 0) return kBins - 2;
  int c = 2 + touching;
  return kBins - 1 - (c < 8 ? c : 7);
}

// The three re-bin kernels sort ONE slot range (a cohort, or all slots): st / keys / eid point at the range's first slot, N stays
// the stride between state fields, n is the number of slots in the range and lo its first slot (slotOf holds absolute slots).
// Every workgroup of the three is ONE wave (kRebinBlock slots = kRebinBlock / 64 per lane).  A cohort's sort runs while the other
// cohort's step_kernel holds the SIMDs with 512-register waves: a multi-wave workgroup then has to wait for a CU with room on
// all its SIMDs at once - rocprofv3 showed the 256-thread histogram of a Dropbox-100k cohort taking 4.1 ms (572 us on average,
// 11 % of the GPU time of a rollout) against 9 us on an idle GPU, and the cohort behind it idle meanwhile; a single wave fits
// wherever one SIMD has a free slot (profiles/r04_dropbox100k_timeline_*.txt).
__global__ __launch_bounds__(64) void rebin_hist_kernel(const DevScene* __restrict__ S, const float* __restrict__ st,
                                                        int N, int n, uint8_t* __restrict__ keys, int* __restrict__ counts, int mode, int tEnd) {
  __shared__ int h[kBins];
  static_assert(kBins <= 64, "one lane per bin");
  if (threadIdx.x < kBins) h[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll 1
  for (int q = 0; q < kRebinBlock / 64; ++q) {
    const int slot = blockIdx.x * kRebinBlock + q * 64 + threadIdx.x;
    if (slot < n) {
      const int k = work_class(S, st, N, slot, mode, tEnd);
      keys[slot] = (uint8_t)k;
      atomicAdd(&h[k], 1);
    }
  }
  __syncthreads();
  if (threadIdx.x < kBins) counts[threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];  // bin-major
}

// exclusive scan of counts[bin][block] in bin-major order: one wave, lane l owns a contiguous run of entries (a multiple of four,
// read and written as int4: the run is walked twice and every access is a dependent L1 hit otherwise)
__global__ __launch_bounds__(64) void rebin_scan_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int n) {
  const int lane = threadIdx.x, per = (((n + 63) / 64) + 3) & ~3;
  const int lo = lane * per, hi = lo + per < n ? lo + per : n;
  int sum = 0;
  int i = lo;
  for (; i + 4 <= hi; i += 4) {
    const int4 c = *reinterpret_cast<const int4*>(counts + i);
    sum += (c.x + c.y) + (c.z + c.w);
  }
  for (; i < hi; ++i) sum += counts[i];
  int incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  int acc = incl - sum;
  i = lo;
  for (; i + 4 <= hi; i += 4) {
    const int4 c = *reinterpret_cast<const int4*>(counts + i);
    int4 o;
    o.x = acc;
    o.y = o.x + c.x;
    o.z = o.y + c.y;
    o.w = o.z + c.z;
    acc = o.w + c.w;
    *reinterpret_cast<int4*>(offsets + i) = o;
  }
  for (; i < hi; ++i) {
    const int c = counts[i];
    offsets[i] = acc;
    acc += c;
  }
}

// stable scatter: new slot = offset[bin][block] + rank of this slot among same-bin slots of the block (in slot order).
// One wave per 64 slots (grid = blocks x kRebinBlock / 64): the same-bin slots of the block's EARLIER quarters are counted from
// their keys (a coalesced 64-byte read per quarter), so the four quarters of a block move in parallel and nothing goes through LDS
// (which a batch at two waves per SIMD leaves no room for: 8 x 19.7 KB of a CU's 160).
__global__ __launch_bounds__(64) void rebin_move_kernel(const float* __restrict__ st, float* __restrict__ st2, int N, int n, int lo,
                                                        int words, const uint8_t* __restrict__ keys,
                                                        const int* __restrict__ offsets, const int* __restrict__ eid,
                                                        int* __restrict__ eid2, int* __restrict__ slotOf) {
  constexpr int Q = kRebinBlock / 64;
  const int lane = threadIdx.x, blk = blockIdx.x / Q, q = blockIdx.x % Q, nBlocks = gridDim.x / Q;
  const int slot = blk * kRebinBlock + q * 64 + lane;
  const int key = slot < n ? (int)keys[slot] : -1;
  int prevKey[Q - 1];
#pragma unroll
  for (int p = 0; p < Q - 1; ++p) {
    const int s2 = blk * kRebinBlock + p * 64 + lane;
    prevKey[p] = (p < q && s2 < n) ? (int)keys[s2] : -2;
  }
  int dst = -1;
  unsigned long long todo = __ballot(key >= 0);
  while (todo) {   // one round per distinct key of these 64 slots (waves of a sorted batch hold one to three)
    const int k = __shfl(key, __ffsll((long long)todo) - 1, 64);
    const unsigned long long m = __ballot(key == k);
    int before = 0;
#pragma unroll
    for (int p = 0; p < Q - 1; ++p) before += __popcll(__ballot(prevKey[p] == k));
    if (key == k) dst = offsets[k * nBlocks + blk] + before + __popcll(m & ((1ull << lane) - 1ull));
    todo &= ~m;
  }
  if (slot < n) {
    for (int f = 0; f < words; ++f) st2[(size_t)f * N + dst] = st[(size_t)f * N + slot];
    const int e = eid[slot];
    eid2[dst] = e;
    slotOf[e] = lo + dst;
  }
}

__global__ void invert_kernel(const int* __restrict__ eid, int* __restrict__ slotOf, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slotOf[eid[i]] = i;
}

__global__ void iota_kernel(int* __restrict__ a, int* __restrict__ b, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    a[i] = i;
    b[i] = i;
  }
}

__global__ void sincos_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ s, float* __restrict__ c) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) blcd_sincosf(x[i], &s[i], &c[i]);
}

// ---------------------------------------------------------------------------------------------------------
// host: scene lowering
// ---------------------------------------------------------------------------------------------------------
static Shape build_shape(const blcd_shape_def& d) {
  Shape s;
  std::memset(&s, 0, sizeof(s));
  if (d.type == 0) {
    ShapeSetCircle(&s, d.radius);
  } else if (d.is_box) {
    ShapeSetAsBox(&s, d.verts[0][0], d.verts[0][1]);
  } else {
    Vec2 vs[BLCD_MAX_POLY_VERTS];
    int n = d.n_verts < BLCD_MAX_POLY_VERTS ? d.n_verts : BLCD_MAX_POLY_VERTS;
    for (int i = 0; i < n; ++i) vs[i] = V2(d.verts[i][0], d.verts[i][1]);
    ShapeSetPolygon(&s, vs, n);
  }
  return s;
}

// b2Body::ResetMassData for a single-fixture dynamic body
static void mass_variant(const Shape& s, float density, DevVariant* v) {
  float mass = 0.0f, I = 0.0f;
  Vec2 localCenter = V2(0.0f, 0.0f);
  if (density != 0.0f) {
    MassData md;
    ShapeComputeMass(&s, &md, density);
    mass += md.mass;
    localCenter += md.mass * md.center;
    I += md.I;
  }
  float invMass, invI;
  if (mass > 0.0f) {
    invMass = 1.0f / mass;
    localCenter *= invMass;
  } else {
    mass = 1.0f;
    invMass = 1.0f;
  }
  if (I > 0.0f) {
    I -= mass * Dot(localCenter, localCenter);
    invI = 1.0f / I;
  } else {
    I = 0.0f;
    invI = 0.0f;
  }
  v->mass = mass;
  v->invMass = invMass;
  v->I = I;
  v->invI = invI;
  v->localCenter = localCenter;
}

static int lower_scene(const blcd_scene_desc& d, DevScene* S) {
  if (d.n_bodies < 1 || d.n_bodies > BLCD_MAX_BODIES) return fail(BLCD_ERR_INVALID, "n_bodies out of range");
  if (d.n_joints < 0 || d.n_joints > BLCD_MAX_JOINTS) return fail(BLCD_ERR_INVALID, "n_joints out of range");
  if (d.n_shapes < 1 || d.n_shapes > BLCD_MAX_SHAPES) return fail(BLCD_ERR_INVALID, "n_shapes out of range");
  if (d.n_obs < 0 || d.n_obs > BLCD_MAX_OBS) return fail(BLCD_ERR_INVALID, "n_obs out of range");
  if (!((d.lcd_h == 16 && d.lcd_w <= 32) || (d.lcd_h == 32 && d.lcd_w <= 64)) || d.lcd_w % 8 != 0)
    return fail(BLCD_ERR_UNSUPPORTED, "LCD size must be 16x{16,24,32} or 32x{32,48,64}");
  std::memset(S, 0, sizeof(*S));
  S->nb = d.n_bodies;
  S->nj = d.n_joints;
  S->nobs = d.n_obs;
  S->nact = d.n_act;
  S->lcdW = d.lcd_w;
  S->lcdH = d.lcd_h;
  S->rasterVariant = d.raster_variant;
  S->worldW = d.world_w;
  S->worldH = d.world_h;
  S->gravity = V2(d.gravity[0], d.gravity[1]);
  S->dt = d.dt;
  S->substeps = d.substeps;
  S->velIters = d.vel_iters;
  S->posIters = d.pos_iters;
  S->nShapes = d.n_shapes;
#ifdef BLCD_ABLATION
  S->dbgSkip = getenv("BLCD_DEBUG_SKIP") ? atoi(getenv("BLCD_DEBUG_SKIP")) : 0;
#else
  S->dbgSkip = 0;
#endif
  // walls: boxLCD/world_env.py:311-314 — bottom, left, right, top
  float W = d.world_w, H = d.world_h;
  Vec2 ev[4][2] = {{V2(0, 0), V2(W, 0)}, {V2(0, 0), V2(0, H)}, {V2(W, 0), V2(W, H)}, {V2(0, H), V2(W, H)}};
  Transform ident;
  ident.p = V2(0.0f, 0.0f);
  ident.q.Set(0.0f);
  for (int i = 0; i < 4; ++i) {
    ShapeSetEdge(&S->wallShape[i], ev[i][0], ev[i][1]);
    AABB aabb;
    ShapeComputeAABB(&S->wallShape[i], &aabb, ident);
    S->wallFat[i].lo = aabb.lo - V2(kAabbExtension, kAabbExtension);
    S->wallFat[i].hi = aabb.hi + V2(kAabbExtension, kAabbExtension);
    // per-wall constants of the wall narrow phase and of the TOI early-out: evaluated here once (IEEE sqrt / divide, no contraction:
    // the host's float arithmetic is the device's), read by the kernels through the scalar path - as VALU results computed at
    // kernel entry they occupied ~70 vector registers per lane for wave-uniform values
    S->wallK[i] = MakeWallK(S->wallShape[i].v[0], S->wallShape[i].v[1], S->wallShape[i].radius);
    {
      const Vec2 ed = S->wallShape[i].v[1] - S->wallShape[i].v[0];
      const float len2 = Dot(ed, ed);
      const float inv = 1.0f / sqrtf(len2);
      S->wallNrm[i] = V2(ed.y * inv, -ed.x * inv);   // Cross(ed, r) * inv == Dot(r, wallNrm)
      S->wallTan[i] = V2(ed.x / len2, ed.y / len2);
    }
  }
  for (int i = 0; i < d.n_shapes; ++i) S->shapes[i] = build_shape(d.shapes[i]);
  for (int i = 0; i < d.n_bodies; ++i) {
    const blcd_body_def& bd = d.bodies[i];
    DevBody& B = S->bodies[i];
    if (bd.n_choices < 1 || bd.n_choices > 2) return fail(BLCD_ERR_INVALID, "body n_choices must be 1 or 2");
    B.nChoices = bd.n_choices;
    for (int k = 0; k < bd.n_choices; ++k) {
      if (bd.shape[k] < 0 || bd.shape[k] >= d.n_shapes) return fail(BLCD_ERR_INVALID, "body shape index out of range");
      B.var[k].shape = bd.shape[k];
      mass_variant(S->shapes[bd.shape[k]], bd.density, &B.var[k]);
    }
    if (bd.n_choices == 1) B.var[1] = B.var[0];
    B.friction = bd.friction;
    B.restitution = bd.restitution;
    B.linearDamping = bd.linear_damping;
    B.angularDamping = bd.angular_damping;
    B.cat = bd.category_bits;
    B.mask = bd.mask_bits;
    B.nJoints = 0;
    S->bodyKind[i] = bd.kind;
  }
  for (int j = 0; j < d.n_joints; ++j) {
    const blcd_joint_def& jd = d.joints[j];
    if (jd.body_a < 0 || jd.body_a >= d.n_bodies || jd.body_b < 0 || jd.body_b >= d.n_bodies || jd.body_a == jd.body_b)
      return fail(BLCD_ERR_INVALID, "joint body index out of range");
    DevJoint& J = S->joints[j];
    J.bodyA = jd.body_a;
    J.bodyB = jd.body_b;
    J.anchorA = V2(jd.anchor_a[0], jd.anchor_a[1]);
    J.anchorB = V2(jd.anchor_b[0], jd.anchor_b[1]);
    J.enableLimit = jd.enable_limit;
    J.lower = jd.lower;
    J.upper = jd.upper;
    J.maxMotorTorque = jd.max_motor_torque;
    J.speed = jd.speed;
    J.actionIndex = jd.action_index;
    if (jd.action_index >= d.n_act) return fail(BLCD_ERR_INVALID, "joint action_index out of range");
  }
  // joint-edge lists, newest first
  for (int j = d.n_joints - 1; j >= 0; --j) {
    int ends[2] = {S->joints[j].bodyA, S->joints[j].bodyB};
    for (int k = 0; k < 2; ++k) {
      DevBody& B = S->bodies[ends[k]];
      if (B.nJoints >= 8) return fail(BLCD_ERR_UNSUPPORTED, "more than 8 joints on one body");
      B.joints[B.nJoints++] = j;
    }
  }
  // pair slots: (A,B)-sorted proxies that AddPair can accept (b2Body::ShouldCollide + b2ContactFilter::ShouldCollide)
  int np = 0;
  int total = 4 + d.n_bodies;
  for (int a = 0; a < total; ++a)
    for (int b = a + 1; b < total; ++b) {
      if (b < 4) continue;  // both static
      uint32_t catA = a < 4 ? 0x0001u : S->bodies[a - 4].cat, maskA = a < 4 ? 0xFFFFu : S->bodies[a - 4].mask;
      uint32_t catB = S->bodies[b - 4].cat, maskB = S->bodies[b - 4].mask;
      float fricA = a < 4 ? 0.2f : S->bodies[a - 4].friction, restA = a < 4 ? 0.0f : S->bodies[a - 4].restitution;
      float fricB = S->bodies[b - 4].friction, restB = S->bodies[b - 4].restitution;
      bool jointed = false;
      if (a >= 4)
        for (int j = 0; j < d.n_joints; ++j) {
          int ja = S->joints[j].bodyA + 4, jb = S->joints[j].bodyB + 4;
          if ((ja == a && jb == b) || (ja == b && jb == a)) jointed = true;
        }
      if (jointed) continue;
      if (!((maskA & catB) != 0 && (catA & maskB) != 0)) continue;
      if (np >= kMaxPairs) return fail(BLCD_ERR_UNSUPPORTED, "scene has more than 100 collidable pairs");
      DevPair& P = S->pairs[np++];
      P.a = a;
      P.b = b;
      P.friction = sqrtf(fricA * fricB);                 // b2MixFriction (argument order: fixtureA, fixtureB)
      P.restitution = restA > restB ? restA : restB;     // b2MixRestitution
    }
  S->np = np;
  for (int i = 0; i < d.n_obs; ++i) {
    const blcd_obs_def& od = d.obs[i];
    if (od.body < 0 || od.body >= d.n_bodies || od.kind < 0 || od.kind > 5) return fail(BLCD_ERR_INVALID, "bad obs entry");
    S->obs[i].kind = od.kind;
    S->obs[i].body = od.body;
    S->obs[i].lo = od.lo;
    S->obs[i].hi = od.hi;
  }
  return BLCD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------------------------
struct blcd_handle_s {
  int device = 0;
  int N = 0;
  DevScene hostScene;
  DevScene* dScene = nullptr;
  float* st = nullptr;
  float* st2 = nullptr;     // second state buffer for re-binning
  int *eid = nullptr, *eid2 = nullptr, *slotOf = nullptr;
  uint8_t* keys = nullptr;
  int *binCounts = nullptr, *binOffsets = nullptr;
  int rebinEvery = 0;       // 0 = never; k = after every k-th env step
  static constexpr int kMaxCohortsDecl = 4;
  int lanes = 64;           // environments per wave in step_kernel (BLCD_LANES)
  // environment-level scheduling of fused chunks (joint-free classes): passes per chunk (1 = off) and the most
  // lanes of a wave that may still be sweeping for those lanes to be suspended (BLCD_YIELD_PASSES, BLCD_YIELD_LANES)
  int yieldPasses = 1, yieldMaxLanes = 0;
  // asynchronous rollouts (BLCD_ASYNC=<world steps per launch>): every launch advances every unfinished environment by at most
  // that many world steps from wherever it stands, suspended ones pay what they owe in dense waves, a re-bin follows every launch
  int asyncBudget = 0;
  long long asyncLaunches = 0;
  // in-wave batching (BLCD_WAVE_BATCH=<lanes>): the plain chunked rollout on the scheduler's kernel - a suspended lane resumes
  // inside the same launch, together with the other lanes of its wave that are owed the same kind of work
  int waveBatch = 0;
  // two wave widths per launch for re-binned batches (BLCD_TWO_WIDTHS=0 turns it off): the awake slots in narrower waves once they
  // no longer fill the SIMDs; sortedBlocks[c] = blocks of cohort c's last slot sort (0 = the current order is not a sorted one)
  bool cohortsBusy = false;    // work has been queued on a cohort's own stream since the last join_cohort_stream
  // Asynchronous steps (blcd_step_obs_async): the stream holding queued work that no call has synchronised yet (nullptr: none) - the
  // handle's own stream, or the caller's stream adopted with blcd_set_async_stream; evExt orders one behind the other on the device
  bool pending = false;               // pendingStream holds queued asynchronous work (the stream itself may be the null stream)
  hipStream_t pendingStream = nullptr;
  bool asyncAdopt = false;            // blcd_set_async_stream: asynchronous steps go on asyncStream (which may be the null stream)
  hipStream_t asyncStream = nullptr;
  hipEvent_t evExt = nullptr;
  int unsortedSpread = 1;   // BLCD_UNSORTED_SPREAD=0: ranges that no sort has ordered keep full waves
  int twoWidths = 16, nSimds = 0, twSlots = 0;   // twSlots: wave slots the awake region of a two-width launch is spread over (BLCD_TW_SLOTS; default nSimds)
  int sortedBlocks[kMaxCohortsDecl] = {0, 0, 0, 0};
  // goal epilogue (blcd_goal_*): device-resident goals, previous deltas and scratch observation buffers
  blcd_goal_desc goal{};
  bool goalSet = false;
  blcd_goal_desc* dGoal = nullptr;
  double *goalFs = nullptr, *goalLast = nullptr, *goalObs = nullptr;
  uint8_t *goalLcd = nullptr, *goalCurLcd = nullptr;
  int binMode = 1;          // work_class: 0 generic classes only, 1 impact-time sorting for one-body scenes (BLCD_BINMODE)
  int rolloutChunk = 20;    // env steps per fused rollout launch (BLCD_CHUNK; 0 = one launch per step + separate obs kernel)
  bool chunkFixed = false;  // BLCD_CHUNK given: no adaptation
  float estMsPerStep = 0.0f;  // step-kernel time per env step of the last fused rollout (sizes the chunks of jointed scenes)
  unsigned long long* waveTimes = nullptr;  // per-wave duration of the last step launch (diagnostic, BLCD_WAVETIMES=1)
  int stepsSinceRebin = 0;
  // Re-binned batches right after a FULL reset: every environment is falling / landing / settling, every wave holds a lane that
  // runs all its sweeps, so all waves of a chunk last about equally long and a batch of 1.5 x as many waves as SIMDs needs two
  // rounds per chunk - unless the chunks are short enough for the two cohorts to interleave.  For the first
  // phase0Steps env-steps after a full reset the chunks are phase0Chunk long (0 = no such phase).  BLCD_CHUNK0=<steps>:<len>.
  int phase0Steps = 0, phase0Chunk = 0, phase0NoSort = 1;
  int cohortSwap = 0;
  long long stepsSinceFullReset = 1 << 30;
  // Cohorts: an oversubscribed joint-free batch is stepped as two slot ranges on two streams, each re-binned within itself, so
  // that neither waits at a chunk boundary for the slowest wave of the whole batch.  cohortLo[c] ..
  // cohortLo[c + 1] are cohort c's slots; cohort 0 runs on the handle's stream, cohort c > 0 on cstream[c].
  static constexpr int kMaxCohorts = 4;
  int nCohorts = 1;
  int cohortLo[kMaxCohorts + 1] = {0, 0, 0, 0, 0};
  hipStream_t cstream[kMaxCohorts] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t evJoin[kMaxCohorts] = {nullptr, nullptr, nullptr, nullptr};
  size_t words = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<hipEvent_t> evPool;  // pairs of events, one pair per step launch of a rollout
  float lastMs = 0.0f;
  int lastLaunches = 0;
  // staging
  void* stage[4] = {nullptr, nullptr, nullptr, nullptr};
  int* dFaultAny = nullptr;  // flag raised by step_kernel when an environment is faulted: device view of ...
  volatile int* hFaultAny = nullptr;   // ... one int of pinned, host-coherent memory: the host reads it after a stream synchronisation instead of copying it back (one round trip less per call)
  unsigned long long* dSchedStats = nullptr;   // blcd_sched_stats
  int* dEpisode = nullptr;                     // blcd_reset_sampled: per-environment reset count; staging of the sampled poses / shapes
  uint64_t envIdBase = 0;                      // blcd_sample_set_base: global id of this handle's environment 0
  std::vector<uint8_t> seen;                   // duplicate check of host index lists
  blcd_sample_op* dSampleOps = nullptr;
  float* dSamplePoses = nullptr;
  int* dSampleSel = nullptr;
  uint8_t* dLut = nullptr;   // Pillow's ellipse span table for blcd_render_poses_ex (uploaded on first use)
  int lutAmax = -1;
  int* dErr = nullptr;
  size_t stageBytes[4] = {0, 0, 0, 0};
  int cfg = -1;
};

static inline hipStream_t cohort_stream(blcd_handle h, int c) {
  if (h->cohortSwap && h->nCohorts >= 2 && c <= 1) c = 1 - c;
  if (c > 0) h->cohortsBusy = true;   // every user of a cohort's own stream queues work on it
  return c > 0 ? h->cstream[c] : h->stream;
}

static int ensure_stage(blcd_handle h, int k, size_t bytes) {
  if (h->stageBytes[k] >= bytes) return BLCD_OK;
  if (h->stage[k]) HIPCHK(hipFree(h->stage[k]));
  h->stage[k] = nullptr;
  h->stageBytes[k] = 0;
  HIPCHK(hipMalloc(&h->stage[k], bytes));
  h->stageBytes[k] = bytes;
  return BLCD_OK;
}

static bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// returns a device pointer holding `bytes` of input data (staged if `p` is host memory)
static int in_ptr(blcd_handle h, int k, const void* p, size_t bytes, const void** out) {
  if (!p) {
    *out = nullptr;
    return BLCD_OK;
  }
  if (is_device_ptr(p)) {
    *out = p;
    return BLCD_OK;
  }
  int rc = ensure_stage(h, k, bytes);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(h->stage[k], p, bytes, hipMemcpyHostToDevice, h->stream));
  *out = h->stage[k];
  return BLCD_OK;
}
// returns a device pointer to write `bytes` into; if `p` is host memory the caller must call out_done afterwards
static int out_ptr(blcd_handle h, int k, void* p, size_t bytes, void** out) {
  if (!p) {
    *out = nullptr;
    return BLCD_OK;
  }
  if (is_device_ptr(p)) {
    *out = p;
    return BLCD_OK;
  }
  int rc = ensure_stage(h, k, bytes);
  if (rc) return rc;
  *out = h->stage[k];
  return BLCD_OK;
}
static int out_done(blcd_handle h, int k, void* p, size_t bytes, void* dev) {
  if (!p || dev == p) return BLCD_OK;
  HIPCHK(hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return BLCD_OK;
}

static FILE* g_launchLog = nullptr;   // diagnostic: see launch_step
// the stream cohort c runs on: cohort 0 on the handle's stream, cohort c > 0 on cstream[c] (BLCD_COHORT_SWAP: experiment, cohorts 0 and 1 exchanged)
static inline hipStream_t cohort_stream(blcd_handle h, int c);
constexpr int kBigLanes = BLCD_BIG_LANES;   // blcd_world.h

// ---- template dispatch over (max bodies, max joints, max pair slots) -------------------------------------
struct Cfg {
  int nb, nj, np, sh;
};
static const Cfg kCfgs[] = {
#define X(a, b, c, d) {a, b, c, d},
    BLCD_CONFIGS(X)
#undef X
};
static int pick_cfg(const DevScene& S) {
  // circles-only classes: every variant of every dynamic body is a circle centred on the body origin (so is its centre of
  // mass) and nothing is jointed - the conditions under which contact code never needs a body's rotation (Env::rotFor)
  bool circlesOnly = S.nj == 0;
  for (int i = 0; i < S.nb; ++i)
    for (int k = 0; k < S.bodies[i].nChoices; ++k) {
      const Shape& sh = S.shapes[S.bodies[i].var[k].shape];
      const Vec2 lc = S.bodies[i].var[k].localCenter;
      circlesOnly = circlesOnly && sh.type == kCircle && sh.v[0].x == 0.0f && sh.v[0].y == 0.0f && lc.x == 0.0f && lc.y == 0.0f;
    }
  for (size_t i = 0; i < sizeof(kCfgs) / sizeof(kCfgs[0]); ++i)
    if (S.nb <= kCfgs[i].nb && S.nj <= kCfgs[i].nj && S.np <= kCfgs[i].np && (kCfgs[i].sh == 0 || circlesOnly)) return (int)i;
  return -1;
}

// cohort < 0: all slots on the handle's stream, bracketed by the events (the handle's own pair if none are given);
// cohort >= 0: that cohort's slot range on its stream, no events (the caller times the whole sequence)
// pass / nPasses: a fused chunk is stepped in nPasses launches (environment-level scheduling): in all but the last
// one a lane may suspend a straggling environment, which the next pass resumes; e0 is recorded before the first pass and e1
// after the last, so a "launch" of the timing code stays one chunk of the batch
static int launch_step(blcd_handle h, const float* dActions, int nEnvSteps, int nWorldSteps, int setMotors,
                       hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, long long actStride = 0, uint8_t* lcdOut = nullptr,
                       float* obsOut = nullptr, int cohort = -1, int pass = 0, int nPasses = 1, int lcdBits = 0, int stepBudget = 0) {
  int lanes = h->lanes;
  const int lo = cohort < 0 ? 0 : h->cohortLo[cohort];
  const int n = cohort < 0 ? h->N : h->cohortLo[cohort + 1] - lo;
  hipStream_t stream = cohort < 0 ? h->stream : cohort_stream(h, cohort);
  const bool widthsFree = h->twoWidths && nEnvSteps > 0 && nPasses == 1 && stepBudget == 0 && h->waveBatch == 0 && lanes == 64 && !h->waveTimes &&
                          (cohort >= 0 || h->nCohorts == 1);
  // a slot range that no sort has ordered yet (the first chunk after a full reset of a circles-only scene) and that is smaller
  // than one full wave per wave slot: the same spreading as the awake region of a two-width launch, by the plain `lanes` argument
  if (widthsFree && h->rebinEvery > 0 && h->sortedBlocks[cohort < 0 ? 0 : cohort] <= 0 && h->unsortedSpread && h->twSlots > 0 && (long long)n < 64LL * h->twSlots) {
    const int l = (n + h->twSlots - 1) / h->twSlots;
    lanes = l < h->twoWidths ? h->twoWidths : l;
  }
  dim3 grid((n + lanes - 1) / lanes), block(kBlock);
  // two wave widths (see step_kernel): only for the plain kernel on a slot order that the last sort produced for exactly this range
  const int* heavyEnd = nullptr;
  const int cIdx = cohort < 0 ? 0 : cohort;
  if (widthsFree && h->sortedBlocks[cIdx] > 0) {
    const int nBlocksAll = (h->N + kRebinBlock - 1) / kRebinBlock + blcd_handle_s::kMaxCohorts;
    heavyEnd = h->binOffsets + (size_t)cIdx * binTableStride(nBlocksAll) + (size_t)(kBins - 1) * h->sortedBlocks[cIdx];   // start of the asleep bin
    grid = dim3(h->twSlots + 2 + (n + 63) / 64);   // heavy blocks <= max(nSimds, heavy / 64) + 1, light blocks <= light / 64 + 1
  }
  if (cohort < 0) {
    if (!e0) {
      e0 = h->ev0;
      e1 = h->ev1;
    }
    if (pass == 0 && stepBudget == 0) HIPCHK(hipEventRecord(e0, stream));
  }
  StepArgs A{h->dScene, h->st + lo, h->N, n, h->eid + lo, dActions, nEnvSteps, nWorldSteps, setMotors, lanes, h->waveTimes, actStride, lcdOut, obsOut, h->dFaultAny,
             pass, (pass + 1 < nPasses || stepBudget > 0 || (h->waveBatch > 0 && nEnvSteps > 0)) ? h->yieldMaxLanes : 0, h->dSchedStats, lcdBits,
             (nPasses > 1 || stepBudget > 0 || (h->waveBatch > 0 && nEnvSteps > 0)) ? 1 : 0, stepBudget, heavyEnd, (heavyEnd ? h->twSlots : h->nSimds) | (h->twoWidths << 16),
             stepBudget > 0 || nPasses > 1 ? 0 : h->waveBatch};
  int idx = 0;
#define X(a, b, c, d) \
  if (h->cfg == idx) launch_step_##a##_##b##_##c##_##d(grid, stream, A); \
  ++idx;
  BLCD_CONFIGS(X)
#undef X
  HIPCHK(hipGetLastError());
  if (g_launchLog) {   // BLCD_LAUNCH_LOG=<file>: one line per step_kernel dispatch, in dispatch order (tools/pmc_summary.py groups counters by launch shape)
    fprintf(g_launchLog, "%d %d %d %d %d\n", nEnvSteps, nWorldSteps, n, cohort, (int)grid.x);
    fflush(g_launchLog);
  }
  if (cohort < 0 && pass + 1 == nPasses && stepBudget == 0) {
    HIPCHK(hipEventRecord(e1, stream));
    h->lastLaunches += 1;
  }
  return BLCD_OK;
}

static int launch_set_poses(blcd_handle h, const int* dIdx, int n, const float* dPoses, const uint8_t* dMask) {
  dim3 grid((n + 63) / 64), block(64);
  SetPosesArgs A{h->dScene, h->st, h->N, h->slotOf, dIdx, n, dPoses, dMask};
  int idx = 0;
#define X(a, b, c, d) \
  if (h->cfg == idx) launch_set_poses_##a##_##b##_##c##_##d(grid, h->stream, A); \
  ++idx;
  BLCD_CONFIGS(X)
#undef X
  HIPCHK(hipGetLastError());
  return BLCD_OK;
}

// stable counting sort of slots by work class, within each cohort and on that cohort's stream; swaps the state buffers
// midChunk: between the passes of a fused chunk (suspended environments to the front; the rest keeps its work-class order, or - in
// batches that are not sorted by work class at all - its place)
static int launch_rebin(blcd_handle h, bool midChunk = false, int tEnd = 0) {
  const int nBlocksAll = (h->N + kRebinBlock - 1) / kRebinBlock + blcd_handle_s::kMaxCohorts;   // room for every cohort's block counts (rounding)
  for (int c = 0; c < h->nCohorts; ++c) {
    const int lo = h->cohortLo[c], n = h->cohortLo[c + 1] - lo;
    hipStream_t stream = cohort_stream(h, c);
    const int nBlocks = (n + kRebinBlock - 1) / kRebinBlock;
    int* counts = h->binCounts + (size_t)c * binTableStride(nBlocksAll);
    int* offsets = h->binOffsets + (size_t)c * binTableStride(nBlocksAll);
    h->sortedBlocks[c] = midChunk ? 0 : nBlocks;
    hipLaunchKernelGGL(rebin_hist_kernel, dim3(nBlocks), dim3(64), 0, stream, h->dScene, h->st + lo, h->N, n, h->keys + lo, counts,
                       midChunk && h->rebinEvery <= 0 ? -1 : h->binMode, tEnd);
    hipLaunchKernelGGL(rebin_scan_kernel, dim3(1), dim3(64), 0, stream, counts, offsets, nBlocks * kBins);
    hipLaunchKernelGGL(rebin_move_kernel, dim3(nBlocks * (kRebinBlock / 64)), dim3(64), 0, stream, h->st + lo, h->st2 + lo, h->N, n, lo, (int)h->words,
                       h->keys + lo, offsets, h->eid + lo, h->eid2 + lo, h->slotOf);
  }
  HIPCHK(hipGetLastError());
  std::swap(h->st, h->st2);
  std::swap(h->eid, h->eid2);
  if (!midChunk || h->rebinEvery > 0) h->stepsSinceRebin = 0;
  return BLCD_OK;
}
// host-side wait for asynchronous steps that are still queued (on whichever stream they were put)
static int drain_async(blcd_handle h) {
  if (h->pending) {
    HIPCHK(hipStreamSynchronize(h->pendingStream));
    h->pending = false;
  }
  return BLCD_OK;
}
// every entry point that works on the handle's own stream: asynchronous steps queued on an adopted stream come first (device-side)
static int enter(blcd_handle h) {
  HIPCHK(hipSetDevice(h->device));
  if (h->pending && h->pendingStream != h->stream) {
    if (!h->evExt) HIPCHK(hipEventCreateWithFlags(&h->evExt, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->evExt, h->pendingStream));
    HIPCHK(hipStreamWaitEvent(h->stream, h->evExt, 0));
    h->pendingStream = h->stream;   // what is pending now sits behind that wait on the handle's own stream
  }
  return BLCD_OK;
}
// everything outside the fused rollout runs on the handle's stream: let it see cohort 1's re-bin
static int join_cohort_stream(blcd_handle h) {
  if (!h->cohortsBusy) return BLCD_OK;   // nothing has been queued on them since the last join (a step loop pays this check per call)
  for (int c = 1; c < h->nCohorts; ++c) HIPCHK(hipStreamSynchronize(h->cstream[c]));
  h->cohortsBusy = false;
  return BLCD_OK;
}
// Re-bin lazily, right before a launch, once a chunk's worth of env steps has passed since the last sort - whoever cut the
// steps into calls (one blcd_rollout, a caller that feeds 20-step rollouts to overlap a gather, or blcd_step in a loop).
static int rebin_if_due(blcd_handle h, int chunkNow = 0) {
  if (h->rebinEvery <= 0) return BLCD_OK;
  // inside the settling phase after a full reset the slots keep their (impact-time sorted) order: sorting by work class there
  // packs the few environments that sweep to the end into the same waves, which then take 0.5 ms per env-step while the mixed
  // waves take 0.15 (per-wave timers, profiles/r04_dropbox100k_chunk_waves.txt) - and a chunk lasts as long as its slowest wave
  if (h->phase0Chunk > 0 && h->phase0NoSort && h->stepsSinceFullReset > 0 && h->stepsSinceFullReset < h->phase0Steps) return BLCD_OK;
  int interval = chunkNow > 0 ? chunkNow : (h->rolloutChunk > 0 ? h->rolloutChunk : 20);
  if (interval < h->rebinEvery) interval = h->rebinEvery;
  if (h->stepsSinceRebin >= interval) {
    // an asynchronous step (blcd_step_obs_async) may still be running on the handle's stream, and the sort runs on the cohorts'
    if (int rc = drain_async(h)) return rc;
    return launch_rebin(h);
  }
  return BLCD_OK;
}

template <typename ObsT>
static int launch_obs(blcd_handle h, ObsT* dObs, uint8_t* dLcd) {
  dim3 grid((h->N + kBlock - 1) / kBlock), block(kBlock);
  if (h->hostScene.lcdH == 16)
    hipLaunchKernelGGL((obs_kernel<16, uint32_t, ObsT>), grid, block, 0, h->stream, h->dScene, h->st, h->N, h->eid, dObs, dLcd, h->st);
  else
    hipLaunchKernelGGL((obs_kernel<32, uint64_t, ObsT>), grid, block, 0, h->stream, h->dScene, h->st, h->N, h->eid, dObs, dLcd, h->st);
  HIPCHK(hipGetLastError());
  return BLCD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Goal epilogue kernels (research/wrappers/body_goal.py:58-88, cube_goal.py:64-86): one thread per environment.
// np.mean over a contiguous float64 vector is numpy's pairwise sum divided by n (n < 8: sequential from 0.0; n <= 128:
// eight interleaved accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail sequentially) - checked
// against numpy on random vectors of 2..100 elements (tests/test_gpu_goal.py compares bit for bit).
// ---------------------------------------------------------------------------------------------------------
template <typename F>
__device__ inline double np_pairwise_sum(F a, int n) {  // a(i) -> double, n <= 128
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += a(i);
    return res;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a(j);
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a(i + j);
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a(i);
  return res;
}
template <typename F>
__device__ inline double np_mean(F a, int n) {
  return np_pairwise_sum(a, n) / (double)n;
}

__global__ void goal_scatter_kernel(const int* __restrict__ idxs, int n, int N, int nobs, int lcdBytes,
                                    const double* __restrict__ fs, const uint8_t* __restrict__ lcd, double* __restrict__ goalFs,
                                    uint8_t* __restrict__ goalLcd) {
  int k = blockIdx.x;
  if (k >= n) return;
  int e = idxs ? idxs[k] : k;
  if (e < 0 || e >= N) return;
  for (int i = threadIdx.x; i < nobs; i += blockDim.x) goalFs[(size_t)e * nobs + i] = fs[(size_t)k * nobs + i];
  if (lcd)
    for (int i = threadIdx.x; i < lcdBytes; i += blockDim.x) goalLcd[(size_t)e * lcdBytes + i] = lcd[(size_t)k * lcdBytes + i];
}

// seedOnly: last_delta := delta for the listed envs; otherwise the full evaluation for every env
__global__ void goal_eval_kernel(const blcd_goal_desc* __restrict__ G, int N, int nobs, int lcdBytes, const int* __restrict__ idxs,
                                 int n, int seedOnly, const double* __restrict__ obs, const uint8_t* __restrict__ lcd,
                                 const double* __restrict__ goalFs, const uint8_t* __restrict__ goalLcd,
                                 double* __restrict__ last, double* __restrict__ rew, uint8_t* __restrict__ done,
                                 double* __restrict__ deltaOut) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int e = idxs ? idxs[k] : k;
  if (e < 0 || e >= N) return;
  double delta, r;
  bool d = false;
  if (G->mode == 0) {
    const double* o = obs + (size_t)e * nobs;
    const double* g = goalFs + (size_t)e * nobs;
    delta = np_mean([&](int i) { int c = G->idxs[i]; return fabs(g[c] - o[c]); }, G->n_idx);
    if (seedOnly) {
      last[e] = delta;
      return;
    }
    if (G->diff_delt) r = -0.05 + 10.0 * (last[e] - delta);
    else r = -delta;
    if (delta < G->thresh) {
      r += 1.0;
      d = true;
    }
    last[e] = delta;
  } else {
    if (seedOnly) return;
    const uint8_t* a = lcd + (size_t)e * lcdBytes;
    const uint8_t* b = goalLcd + (size_t)e * lcdBytes;
    int both = 0, zero = 0;
    for (int i = 0; i < lcdBytes; ++i) {
      zero += a[i] == 0;
      both += (a[i] == 0) && (a[i] == b[i]);
    }
    delta = ((double)both / (double)lcdBytes) / ((double)zero / (double)lcdBytes);   // two np.mean of bool arrays
    r = -1.0 + delta;
    if (delta > G->thresh) {
      r = 0.0;
      d = true;
    }
  }
  r = r * G->rew_scale;
  if (rew) rew[e] = r;
  if (done) done[e] = d ? 1 : 0;
  if (deltaOut) deltaOut[e] = delta;
}

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
extern "C" {

int blcd_version(void) { return BLCD_VERSION; }
int blcd_build_features(void) {
  int f = 0;
#ifdef BLCD_WAVETIMES
  f |= BLCD_FEATURE_WAVETIMES;
#endif
#ifdef BLCD_SCHED
  f |= BLCD_FEATURE_SCHED;
#endif
  return f;
}
const char* blcd_last_error(void) { return g_err.c_str(); }
int blcd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// All handles of a device share ONE stream (created on first use, kept for the life of the process).  Every HIP stream maps
// onto one of the runtime's hardware queues and each queue reserves scratch for the largest kernel it has run; with handles
// spread over several queues the 17-body class (12 KB of scratch per lane) was measured to drop onto a per-dispatch
// scratch path - 0.33 s per launch instead of 8 ms (GPU_MAX_HW_QUEUES=1 made it disappear).  Calls on different handles are
// therefore serialised on the device; each call still ends with a stream synchronisation, as before.
static std::mutex g_streamMutex;
static std::vector<std::pair<int, hipStream_t>> g_deviceStream;
static int acquire_stream(int device, hipStream_t* out) {
  std::lock_guard<std::mutex> lock(g_streamMutex);
  if (getenv("BLCD_PRIVATE_STREAM")) {   // experiments only (cohorts on separate streams); see the note above about queues
    HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return BLCD_OK;
  }
  for (auto& e : g_deviceStream)
    if (e.first == device) {
      *out = e.second;
      return BLCD_OK;
    }
  HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  g_deviceStream.emplace_back(device, *out);
  return BLCD_OK;
}
static void release_stream(int, hipStream_t) {}

int blcd_create(const blcd_scene_desc* scene, int32_t n_envs, int32_t device, blcd_handle* out) {
  if (!scene || !out || n_envs < 1) return fail(BLCD_ERR_INVALID, "blcd_create: bad arguments");
#ifndef BLCD_WAVETIMES
  // the per-wave timers are compiled out of the product kernels (they cost every class ~20 registers): diagnostic builds only
  if (getenv("BLCD_WAVETIMES")) return fail(BLCD_ERR_UNSUPPORTED, "BLCD_WAVETIMES needs a library built with BLCD_DEFS=-DBLCD_WAVETIMES");
#endif
#ifndef BLCD_SCHED
  // the environment-level schedulers are compiled out of the default build
  for (const char* knob : {"BLCD_ASYNC", "BLCD_WAVE_BATCH", "BLCD_YIELD_PASSES"})
    if (const char* ev = getenv(knob))
      if (atoi(ev) > (knob[5] == 'Y' ? 1 : 0))
        return fail(BLCD_ERR_UNSUPPORTED, "BLCD_ASYNC / BLCD_WAVE_BATCH / BLCD_YIELD_PASSES need a library built with BLCD_DEFS=-DBLCD_SCHED");
#endif
  int ndev = blcd_device_count();
  if (ndev <= 0) return fail(BLCD_ERR_NO_DEVICE, "no HIP device available (boxlcd_hip has no CPU path)");
  if (device < 0 || device >= ndev) return fail(BLCD_ERR_NO_DEVICE, "device index out of range");
  if (!g_launchLog)
    if (const char* path = getenv("BLCD_LAUNCH_LOG")) g_launchLog = fopen(path, "a");
  blcd_handle h = new blcd_handle_s();
  int rc = lower_scene(*scene, &h->hostScene);
  if (rc) {
    delete h;
    return rc;
  }
  h->cfg = pick_cfg(h->hostScene);
  if (h->cfg < 0) {
    delete h;
    return fail(BLCD_ERR_UNSUPPORTED, "scene exceeds the compiled kernel configurations");
  }
  h->device = device;
  h->N = n_envs;
  h->words = (size_t)stateWords(h->hostScene.nb, h->hostScene.nj, h->hostScene.np);
  HIPCHK(hipSetDevice(device));
  { int rcS = acquire_stream(device, &h->stream); if (rcS) return rcS; }
  HIPCHK(hipEventCreate(&h->ev0));
  {
    void* hp = nullptr;
    HIPCHK(hipHostMalloc(&hp, sizeof(int), hipHostMallocMapped));
    *(int*)hp = 0;
    h->hFaultAny = (volatile int*)hp;
    HIPCHK(hipHostGetDevicePointer((void**)&h->dFaultAny, hp, 0));
  }
  HIPCHK(hipMalloc((void**)&h->dSchedStats, 8 * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(h->dSchedStats, 0, 8 * sizeof(unsigned long long), h->stream));
  HIPCHK(hipEventCreate(&h->ev1));
  HIPCHK(hipMalloc((void**)&h->dScene, sizeof(DevScene)));
  HIPCHK(hipMemcpy(h->dScene, &h->hostScene, sizeof(DevScene), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc((void**)&h->st, h->words * (size_t)n_envs * sizeof(float)));
  HIPCHK(hipMemset(h->st, 0, h->words * (size_t)n_envs * sizeof(float)));
  HIPCHK(hipMalloc((void**)&h->st2, h->words * (size_t)n_envs * sizeof(float)));
  HIPCHK(hipMalloc((void**)&h->eid, (size_t)n_envs * sizeof(int)));
  HIPCHK(hipMalloc((void**)&h->eid2, (size_t)n_envs * sizeof(int)));
  HIPCHK(hipMalloc((void**)&h->slotOf, (size_t)n_envs * sizeof(int)));
  HIPCHK(hipMalloc((void**)&h->keys, (size_t)n_envs));
  {
    int nBlocks = (n_envs + kRebinBlock - 1) / kRebinBlock + blcd_handle_s::kMaxCohorts;
    HIPCHK(hipMalloc((void**)&h->binCounts, (size_t)blcd_handle_s::kMaxCohorts * binTableStride(nBlocks) * sizeof(int)));
    HIPCHK(hipMalloc((void**)&h->binOffsets, (size_t)blcd_handle_s::kMaxCohorts * binTableStride(nBlocks) * sizeof(int)));
  }
  hipLaunchKernelGGL(iota_kernel, dim3((n_envs + 255) / 256), dim3(256), 0, h->stream, h->eid, h->slotOf, n_envs);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  if (const char* ev = getenv("BLCD_BINMODE")) h->binMode = atoi(ev);
#ifdef BLCD_WAVETIMES
  if (getenv("BLCD_WAVETIMES")) HIPCHK(hipMalloc((void**)&h->waveTimes, (size_t)n_envs * 9 * sizeof(unsigned long long)));
#endif
  if (const char* ev = getenv("BLCD_CHUNK")) {
    h->rolloutChunk = atoi(ev);
    h->chunkFixed = true;
  }
  {
    // Environments per wave.  A wave costs the UNION of its lanes' code paths (which contact slots exist, 1- or 2-point
    // manifolds, limit states, position iterations up to the slowest lane), and these kernels run one wave per SIMD, so when
    // the batch has fewer than 64 environments per SIMD the waves are made narrower instead of leaving SIMDs idle:
    // Urchin-50k measured 79.1 ms per 20 env-steps with 782 full waves, 74.8 ms with 1021 waves of 49 lanes.
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    const int simds = prop.multiProcessorCount * 4;
    h->nSimds = simds;
    // narrowest wave of the awake region (0 = one width).  Round 3 (tools/tw_bench.sh): Object2-200k +6 %, Bounce2-100k +9 %,
    // Object3-100k +12 % at 16, while the one-body classes lost (Bounce-100k -4 %, Dropbox-100k -1 %) because a narrow wave gave up
    // the wave-coalesced frame store.  Round 4: the lanes beyond a narrow wave's width shadow its last environment (step_kernel), so
    // every wave keeps that path, and the one-body classes gain twice: a cohort's 50 000 awake slots become 1 021 waves of 49 lanes -
    // with two cohorts exactly two waves on every SIMD instead of 1 563 full waves spread 2 / 1 - and the awake eighth of a late
    // chunk is spread over all SIMDs.  Bounce-100k 2.15e9 -> 2.31 / 2.34 / 2.31 / 2.31 / 2.25 / 2.22e9 at 4 / 6 / 8 / 12 / 16 / 32,
    // Dropbox-100k 1.32e9 -> 1.39e9 (8, 16); Object3-100k 2.51e7 -> 2.73e7 (the shadow lanes alone, at 16; 2.78 at 8, 2.56 at 32);
    // Bounce2-100k 9.57e7 / 9.82e7 / 1.05e8 at 8 / 16 / 32; Object2-200k 9.50 / 9.47 / 9.32e7 at 8 / 16 / 32.
    // Spreading over 2 048 wave slots per cohort instead of 1 024 (BLCD_TW_SLOTS) loses everywhere (Bounce-100k 2.21e9, Object3-100k 2.24e7).
    h->twoWidths = h->hostScene.nb == 1 ? 8 : (h->hostScene.nb == 2 && kCfgs[h->cfg].sh == 1 ? 32 : 16);
    h->twSlots = simds;
    if (const char* ev = getenv("BLCD_TW_SLOTS")) {
      int q = atoi(ev);
      if (q >= 64 && q <= 32768) h->twSlots = q;
    }
    // ... and the same spreading for a range no sort has ordered yet (launch_step): Bounce-100k 2.32e9 -> 2.37e9; the one-wave-per-SIMD
    // classes lose (Bounce2-100k 1.04e8 -> 1.00e8: two cohorts of 1 021 waves are two rounds of waves either way) and keep full waves
    h->unsortedSpread = h->hostScene.nb == 1;
    if (const char* ev = getenv("BLCD_UNSORTED_SPREAD")) h->unsortedSpread = atoi(ev) != 0;
    if (const char* ev = getenv("BLCD_TWO_WIDTHS")) {
      int q = atoi(ev);
      h->twoWidths = q <= 0 ? 0 : (q == 1 ? 16 : (q > 64 ? 64 : q));
    }
    if (simds > 0 && (long long)n_envs < 64LL * simds) {
      int l = (n_envs + simds - 1) / simds;
      h->lanes = l < 16 ? 16 : (l > 64 ? 64 : l);   // below ~16 lanes the scratch footprint per useful lane costs more than the narrower union saves (Urchin-4096: 76 / 66 / 97 ms at 64 / 16 / 4 lanes)
    }
    if (const char* ev = getenv("BLCD_LANES")) {   // experiments: overrides the automatic choice (placement only, results unchanged)
      int l = atoi(ev);
      if (l >= 1 && l <= 64) h->lanes = l;
    }
    // The largest class is bound by the traffic of its scratch-resident constraint rows (26 KB per lane), and a scratch line
    // holds one dword of each of a wave's 64 lanes: narrow waves waste most of every line.  Measured (Crab, ms per env-step of
    // the batch): 20 000 envs as 1000 x 20 lanes 35.6, 625 x 32 lanes 27.7, 313 x 64 lanes (old 64-lane blocks) 29.0; 40 000
    // envs 40.4-40.8 either way.  So: the widest waves the class's LDS blocks hold (kBigLanes = 32) once the batch gives every
    // fourth SIMD such a wave; smaller batches keep the narrower-waves rule above (Crab-4096: 2.02e5 at 16 lanes, 1.87e5 at 32).
    if (h->hostScene.nb > 7) {
      if (!getenv("BLCD_LANES") && (long long)n_envs >= (long long)kBigLanes * (simds / 4)) h->lanes = kBigLanes;
      if (h->lanes > kBigLanes) h->lanes = kBigLanes;
    }
    // Re-binning by work class.  It pays when environments sleep / fly freely (no joints keep them awake) AND the batch
    // oversubscribes the SIMDs, so that total wave time is what counts: Bounce-100k 1.66e9 with, 1.09e9 without.  With at most
    // one wave per SIMD the launch lasts as long as its slowest wave, a wave's cost is convex in its number of heavy lanes, and
    // concentrating them makes that wave slower: Bounce-50k +9 %, Dropbox-50k +12 %, Object2-50k +22 %, Object3-50k +30 %
    // WITHOUT re-binning (and then the launches can be long, see blcd_rollout).  BLCD_REBIN overrides (0 = off).
    h->rebinEvery = (h->hostScene.nj == 0 && simds > 0 && (long long)n_envs > 64LL * simds) ? 1 : 0;
  }
  if (const char* ev = getenv("BLCD_REBIN")) h->rebinEvery = atoi(ev);
  // Chunk length of the re-binned one-body batches.  Every chunk boundary costs a sort of the slots and the wait for the chunk's
  // slowest wave; since the one-body kernels lost a quarter of their instructions (round 3: wall-side folding, LDS manifolds)
  // those fixed costs weigh more than the better grouping of a 20-step interval buys (200-step rollouts, tools/chunk_sweep.sh):
  // Bounce-100k 1.84e9 (20) -> 2.02e9 (50) -> 2.12e9 (100); Dropbox-100k 6.4e8 (20) -> 6.6e8 (40, 50) -> 6.4e8 (100).
  // Multi-body batches keep 20 (Object2-200k: 6.3e7 / 6.1e7 / 5.8e7 at 10 / 20 / 40).  Results do not depend on the chunking.
  // Round 4: with two co-resident waves per SIMD the general one-body class no longer runs its settling phase in two rounds, and the
  // longer chunk wins there too: Dropbox-100k 1.01 / 1.15 / 1.20 / 1.24 / 1.26 / 1.30e9 at 30 / 40 / 50 / 67 / 100 / 200 (the circles-only
  // class keeps 100: 2.04 / 2.10 / 1.94e9 at 50 / 100 / 200).
  if (!h->chunkFixed && h->rebinEvery > 0 && h->hostScene.nb == 1) h->rolloutChunk = kCfgs[h->cfg].sh == 1 ? 100 : 200;
  // The general two-body class after the cheaper cycle detection (round 3): Object2-200k 7.65e7 / 7.62e7 / 7.50e7 / 7.31e7 at
  // 5 / 10 / 15 / 20 - its waves are so uneven (5 of 64 lanes busy on average) that regrouping them twice as often pays; the circles
  // two-body class and the three-body class show no trend (Bounce2-100k, Object3-100k within noise or best at 20) and keep 20.
  if (!h->chunkFixed && h->rebinEvery > 0 && h->hostScene.nb == 2 && kCfgs[h->cfg].sh == 0) h->rolloutChunk = 10;
  // the circles two-body class (round 4, on the cheaper re-bin kernels): Bounce2-100k 8.75 / 9.21 / 9.26 / 9.52 / 8.44e7 at 10 / 20 / 50 / 100 / 200
  if (!h->chunkFixed && h->rebinEvery > 0 && h->hostScene.nb == 2 && kCfgs[h->cfg].sh == 1) h->rolloutChunk = 100;
  if (const char* ev = getenv("BLCD_COHORT_SWAP")) h->cohortSwap = atoi(ev) != 0;
  if (const char* ev = getenv("BLCD_CHUNK0")) {
    int a = 0, b = 0;
    int c = 1;
    if (sscanf(ev, "%d:%d:%d", &a, &b, &c) >= 2 && a >= 0 && b >= 0) h->phase0Steps = a, h->phase0Chunk = b, h->phase0NoSort = c;
  }
#ifdef BLCD_SCHED
  if (h->hostScene.nb <= 7) {
    if (const char* ev = getenv("BLCD_ASYNC")) h->asyncBudget = atoi(ev) > 0 ? atoi(ev) : 0;
    if (h->asyncBudget > 0) {
      h->yieldMaxLanes = 32;
      if (const char* ev = getenv("BLCD_YIELD_LANES")) h->yieldMaxLanes = atoi(ev);
      if (h->yieldMaxLanes < 1) h->yieldMaxLanes = 1;
      if (h->yieldMaxLanes > 64) h->yieldMaxLanes = 64;
    }
  }
  if (h->hostScene.nb <= 7 && h->asyncBudget == 0) {
    if (const char* ev = getenv("BLCD_WAVE_BATCH")) h->waveBatch = atoi(ev) > 0 ? atoi(ev) : 0;
    if (h->waveBatch > 0) {
      h->yieldMaxLanes = 16;
      if (const char* ev = getenv("BLCD_YIELD_LANES")) h->yieldMaxLanes = atoi(ev);
      if (h->yieldMaxLanes < 1) h->yieldMaxLanes = 1;
      if (h->yieldMaxLanes > 64) h->yieldMaxLanes = 64;
    }
  }
  if (h->asyncBudget == 0 && h->waveBatch == 0 && h->hostScene.nj == 0 && h->hostScene.nb <= 7 && h->hostScene.velIters > kYieldSweeps) {
    // measured: chunk-level passes do not pay - a suspended environment's remaining chunk is as long a critical
    // path in the resuming pass as it was in the first - so the default is OFF; the mechanism stays (it is what the
    // schedule-invariance test and the asynchronous scheduler build on): BLCD_YIELD_PASSES=2 BLCD_YIELD_LANES=32 turns it on
    h->yieldPasses = 1;
    h->yieldMaxLanes = 32;
    if (const char* ev = getenv("BLCD_YIELD_PASSES")) {
      int q = atoi(ev);
      h->yieldPasses = q < 1 ? 1 : (q > 8 ? 8 : q);
    }
    if (const char* ev = getenv("BLCD_YIELD_LANES")) {
      int q = atoi(ev);
      h->yieldMaxLanes = q < 0 ? 0 : (q > 64 ? 64 : q);
    }
    if (h->yieldPasses == 1 || h->yieldMaxLanes == 0) h->yieldPasses = 1, h->yieldMaxLanes = 0;
  }
#endif
  for (int c = 1; c <= blcd_handle_s::kMaxCohorts; ++c) h->cohortLo[c] = n_envs;
  {
    int k = h->rebinEvery > 0 && !h->waveTimes && h->asyncBudget == 0 ? 2 : 1;     // the re-binned (oversubscribed, joint-free) batches
    if (const char* ev = getenv("BLCD_COHORTS")) {
      int q = atoi(ev);
      k = (q >= 2 && h->rebinEvery > 0 && !h->waveTimes && h->asyncBudget == 0) ? (q > blcd_handle_s::kMaxCohorts ? blcd_handle_s::kMaxCohorts : q) : 1;
    }
    // every cohort needs at least one whole wave of slots: small batches (BLCD_REBIN=1 / BLCD_COHORTS on a few environments)
    // get fewer cohorts, down to the single range
    while (k >= 2 && (long long)n_envs < 64LL * k) --k;
    if (k >= 2) {
      h->nCohorts = k;
      for (int c = 1; c < k; ++c) {
        long long lo = (((long long)n_envs * c / k + 63) / 64) * 64;                // whole waves per cohort
        h->cohortLo[c] = (int)(lo < n_envs ? lo : n_envs);
        HIPCHK(hipStreamCreateWithFlags(&h->cstream[c], hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&h->evJoin[c]));
      }
    }
  }
  *out = h;
  return BLCD_OK;
}

int blcd_destroy(blcd_handle h) {
  if (!h) return BLCD_OK;
  (void)hipSetDevice(h->device);
  if (h->pending) (void)hipStreamSynchronize(h->pendingStream);   // asynchronous steps still queued (possibly on the caller's stream)
  if (h->evExt) (void)hipEventDestroy(h->evExt);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (int k = 0; k < 4; ++k)
    if (h->stage[k]) (void)hipFree(h->stage[k]);
  for (void* q : {(void*)h->dGoal, (void*)h->goalFs, (void*)h->goalLast, (void*)h->goalObs, (void*)h->goalLcd, (void*)h->goalCurLcd})
    if (q) (void)hipFree(q);
  for (int c = 1; c < blcd_handle_s::kMaxCohorts; ++c) {
    if (h->cstream[c]) {
      (void)hipStreamSynchronize(h->cstream[c]);
      (void)hipStreamDestroy(h->cstream[c]);
    }
    if (h->evJoin[c]) (void)hipEventDestroy(h->evJoin[c]);
  }
  if (h->hFaultAny) (void)hipHostFree((void*)h->hFaultAny);
  if (h->dSchedStats) (void)hipFree(h->dSchedStats);
  for (void* q : {(void*)h->dEpisode, (void*)h->dSampleOps, (void*)h->dSamplePoses, (void*)h->dSampleSel})
    if (q) (void)hipFree(q);
  if (h->dLut) (void)hipFree(h->dLut);
  if (h->dErr) (void)hipFree(h->dErr);
  if (h->st) (void)hipFree(h->st);
  if (h->st2) (void)hipFree(h->st2);
  if (h->eid) (void)hipFree(h->eid);
  if (h->eid2) (void)hipFree(h->eid2);
  if (h->slotOf) (void)hipFree(h->slotOf);
  if (h->keys) (void)hipFree(h->keys);
  if (h->binCounts) (void)hipFree(h->binCounts);
  if (h->binOffsets) (void)hipFree(h->binOffsets);
  if (h->dScene) (void)hipFree(h->dScene);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (hipEvent_t ev : h->evPool) (void)hipEventDestroy(ev);
  if (h->waveTimes) (void)hipFree(h->waveTimes);
  if (h->stream) release_stream(h->device, h->stream);
  delete h;
  return BLCD_OK;
}

int blcd_num_envs(blcd_handle h) { return h ? h->N : 0; }
int blcd_num_pairs(blcd_handle h) { return h ? h->hostScene.np : 0; }
int blcd_pair_table(blcd_handle h, int32_t* pairs) {
  if (!h || !pairs) return fail(BLCD_ERR_INVALID, "blcd_pair_table: bad arguments");
  for (int i = 0; i < h->hostScene.np; ++i) {
    pairs[2 * i] = h->hostScene.pairs[i].a;
    pairs[2 * i + 1] = h->hostScene.pairs[i].b;
  }
  return BLCD_OK;
}

// An index list that names an environment twice would have two threads rebuild the same world (and, in the sampler, bump its
// reset count twice): host lists are checked here; device lists are the caller's responsibility.
static int check_idxs(blcd_handle h, const int32_t* idxs, int n, const char* who) {
  if (!idxs || is_device_ptr(idxs)) return BLCD_OK;
  h->seen.assign((size_t)h->N, 0);
  for (int k = 0; k < n; ++k) {
    const int e = idxs[k];
    if (e < 0 || e >= h->N) return fail(BLCD_ERR_INVALID, std::string(who) + ": environment index out of range");
    if (h->seen[e]) return fail(BLCD_ERR_INVALID, std::string(who) + ": an environment index appears twice");
    h->seen[e] = 1;
  }
  return BLCD_OK;
}

int blcd_reset(blcd_handle h, const int32_t* idxs, int32_t n, const float* poses, const int32_t* shape_sel) {
  if (!h || !poses || n < 1) return fail(BLCD_ERR_INVALID, "blcd_reset: bad arguments");
  if (!idxs && n != h->N) return fail(BLCD_ERR_INVALID, "blcd_reset: idxs == NULL requires n == n_envs");
  if (int rcEnter = enter(h)) return rcEnter;
  if (int rcI = check_idxs(h, idxs, n, "blcd_reset")) return rcI;
  const int nb = h->hostScene.nb;
  const void *dIdx, *dPoses, *dSel;
  int rc;
  if ((rc = in_ptr(h, 0, idxs, (size_t)n * sizeof(int32_t), &dIdx))) return rc;
  if ((rc = in_ptr(h, 1, poses, (size_t)n * nb * 3 * sizeof(float), &dPoses))) return rc;
  if ((rc = in_ptr(h, 2, shape_sel, (size_t)n * nb * sizeof(int32_t), &dSel))) return rc;
  hipLaunchKernelGGL(reset_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->dScene, h->st, h->N, h->slotOf, (const int*)dIdx, n,
                     (const float*)dPoses, (const int*)dSel);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int& b : h->sortedBlocks) b = 0;   // reset environments are awake wherever their slot is: the order is no longer a sorted one
  if (!idxs) h->stepsSinceFullReset = 0;
  // A full reset of a circles-only scene restarts the re-bin clock: its first chunk runs best in sampled (well mixed) order
  // (Bounce-100k 1.67e9 vs 1.51e9 when the fresh states are sorted by predicted impact first).  Scenes with polygons keep the
  // clock running, so a sort that is due happens before their first chunk (Dropbox-100k 4.4e8 -> 5.0e8).
  if (!idxs && kCfgs[h->cfg].sh == 1) h->stepsSinceRebin = 0;
  return BLCD_OK;
}

// ---- device-side reset sampling (include/boxlcd.h blcd_reset_sampled) ------------------------------------------------------
__device__ inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* o0, uint32_t* o1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  *o0 = c0;
  *o1 = c1;
}
constexpr int kMaxSampleOps = 160;
__global__ void sample_kernel(const DevScene* __restrict__ S, const blcd_sample_op* __restrict__ ops, int nOps, const int* __restrict__ idxs, int n,
                              int N, uint32_t k0, uint32_t k1, unsigned long long idBase, int* __restrict__ episode, float* __restrict__ poses,
                              int* __restrict__ sel) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int e = idxs ? idxs[k] : k;
  if (e < 0 || e >= N) return;
  const int nb = S->nb;
  const uint32_t epi = (uint32_t)episode[e];
  episode[e] = (int)(epi + 1u);
  const unsigned long long gid = idBase + (unsigned long long)e;   // the environment's id in the whole (possibly sharded) batch
  double r[16];
  float px[20], py[20];      // b2Vec2 position of every body placed so far (float32, as behind the SWIG boundary)
  double ang[20];            // its angle as the reference holds it (float64)
  for (int i = 0; i < 16; ++i) r[i] = 0.0;
  for (int i = 0; i < nb; ++i) sel[(size_t)k * nb + i] = 0;
  uint32_t var = 0;
  auto uniform01 = [&]() {
    uint32_t x0, x1;
    philox4x32_10(k0, k1, (uint32_t)gid, epi, var++, (uint32_t)(gid >> 32), &x0, &x1);
    return ((double)(x0 >> 5) * 67108864.0 + (double)(x1 >> 6)) * (1.0 / 9007199254740992.0);
  };
  for (int q = 0; q < nOps; ++q) {
    const blcd_sample_op op = ops[q];
    if (op.kind == 0) {
      const double a = op.f[0] + (op.f[1] - op.f[0]) * uniform01();
      r[op.d] = ((a + 1.0) / (2.0) * (op.f[3] - op.f[2])) + op.f[2];     // utils.py:117 mapto
    } else if (op.kind == 1) {
      sel[(size_t)k * nb + op.body] = uniform01() < 0.5 ? 0 : 1;
    } else if (op.kind == 2) {
      r[op.d] = atan2(r[op.a], r[op.b]);
    } else if (op.kind == 3) {
      r[op.d] = 0.0;
    } else if (op.kind == 4) {
      px[op.body] = (float)r[op.a];
      py[op.body] = (float)r[op.b];
      ang[op.body] = r[op.d];
    } else if (op.kind == 5) {
      double mangle = ang[op.a] + op.f[0];
      mangle = atan2(sin(mangle), cos(mangle));
      const double pangle = ang[op.parent];
      const double cp = cos(pangle), sp = sin(pangle), cm = cos(mangle), sm = sin(mangle);
      const double aax = cp * op.f[1] + -sp * op.f[2], aay = sp * op.f[1] + cp * op.f[2];
      const double abx = cm * op.f[3] + -sm * op.f[4], aby = sm * op.f[3] + cm * op.f[4];
      px[op.body] = (px[op.parent] + (float)aax) - (float)abx;
      py[op.body] = (py[op.parent] + (float)aay) - (float)aby;
      ang[op.body] = mangle;
    }
    if (op.kind == 4 || op.kind == 5) {
      float* o = poses + ((size_t)k * nb + op.body) * 3;
      o[0] = px[op.body];
      o[1] = py[op.body];
      o[2] = (float)ang[op.body];
    }
  }
}

static int sampler_buffers(blcd_handle h) {
  if (h->dEpisode) return BLCD_OK;
  const int nb = h->hostScene.nb;
  HIPCHK(hipMalloc((void**)&h->dEpisode, (size_t)h->N * sizeof(int)));
  HIPCHK(hipMemsetAsync(h->dEpisode, 0, (size_t)h->N * sizeof(int), h->stream));
  HIPCHK(hipMalloc((void**)&h->dSampleOps, kMaxSampleOps * sizeof(blcd_sample_op)));
  HIPCHK(hipMalloc((void**)&h->dSamplePoses, (size_t)h->N * nb * 3 * sizeof(float)));
  HIPCHK(hipMalloc((void**)&h->dSampleSel, (size_t)h->N * nb * sizeof(int)));
  return BLCD_OK;
}

int blcd_sample_set_base(blcd_handle h, uint64_t env_id_base) {
  if (!h) return fail(BLCD_ERR_INVALID, "blcd_sample_set_base: bad handle");
  h->envIdBase = env_id_base;
  return BLCD_OK;
}

int blcd_sample_reseed(blcd_handle h) {
  if (!h) return fail(BLCD_ERR_INVALID, "blcd_sample_reseed: bad handle");
  if (int rcEnter = enter(h)) return rcEnter;
  if (h->dEpisode) HIPCHK(hipMemsetAsync(h->dEpisode, 0, (size_t)h->N * sizeof(int), h->stream));
  return BLCD_OK;
}

int blcd_reset_sampled(blcd_handle h, const int32_t* idxs, int32_t n, uint64_t seed, const blcd_sample_op* ops, int32_t n_ops) {
  if (!h || !ops || n < 1 || n_ops < 1 || n_ops > kMaxSampleOps) return fail(BLCD_ERR_INVALID, "blcd_reset_sampled: bad arguments");
  if (!idxs && n != h->N) return fail(BLCD_ERR_INVALID, "blcd_reset_sampled: idxs == NULL requires n == n_envs");
  const int nb = h->hostScene.nb;
  for (int q = 0; q < n_ops; ++q) {
    const blcd_sample_op& op = ops[q];
    const bool bodyOk = op.body >= 0 && op.body < nb, regOk = op.d >= 0 && op.d < 16 && op.a >= 0 && op.b >= 0;
    if (op.kind < 0 || op.kind > 5 || !regOk || ((op.kind == 1 || op.kind >= 4) && !bodyOk) || (op.kind <= 4 && op.kind != 1 && (op.a >= 16 || op.b >= 16)) ||
        (op.kind == 5 && (op.parent < 0 || op.parent >= nb || op.a >= nb)))
      return fail(BLCD_ERR_INVALID, "blcd_reset_sampled: malformed sampling program");
  }
  if (int rcEnter = enter(h)) return rcEnter;
  int rc;
  if ((rc = check_idxs(h, idxs, n, "blcd_reset_sampled"))) return rc;
  if ((rc = sampler_buffers(h))) return rc;
  const void* dIdx;
  if ((rc = in_ptr(h, 0, idxs, (size_t)n * sizeof(int32_t), &dIdx))) return rc;
  HIPCHK(hipMemcpyAsync(h->dSampleOps, ops, (size_t)n_ops * sizeof(blcd_sample_op), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(sample_kernel, dim3((n + 127) / 128), dim3(128), 0, h->stream, h->dScene, h->dSampleOps, n_ops, (const int*)dIdx, n, h->N,
                     (uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), (unsigned long long)h->envIdBase, h->dEpisode, h->dSamplePoses, h->dSampleSel);
  HIPCHK(hipGetLastError());
  hipLaunchKernelGGL(reset_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->dScene, h->st, h->N, h->slotOf, (const int*)dIdx, n,
                     (const float*)h->dSamplePoses, (const int*)h->dSampleSel);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int& b : h->sortedBlocks) b = 0;   // reset environments are awake wherever their slot is: the order is no longer a sorted one
  if (!idxs) h->stepsSinceFullReset = 0;
  if (!idxs && kCfgs[h->cfg].sh == 1) h->stepsSinceRebin = 0;
  return BLCD_OK;
}

int blcd_set_poses(blcd_handle h, const int32_t* idxs, int32_t n, const float* poses, const uint8_t* mask) {
  if (!h || !poses || n < 1) return fail(BLCD_ERR_INVALID, "blcd_set_poses: bad arguments");
  if (!idxs && n != h->N) return fail(BLCD_ERR_INVALID, "blcd_set_poses: idxs == NULL requires n == n_envs");
  if (int rcEnter = enter(h)) return rcEnter;
  if (int rcI = check_idxs(h, idxs, n, "blcd_set_poses")) return rcI;
  const int nb = h->hostScene.nb;
  const void *dIdx, *dPoses, *dMask;
  int rc;
  if ((rc = in_ptr(h, 0, idxs, (size_t)n * sizeof(int32_t), &dIdx))) return rc;
  if ((rc = in_ptr(h, 1, poses, (size_t)n * nb * 3 * sizeof(float), &dPoses))) return rc;
  if ((rc = in_ptr(h, 2, mask, (size_t)nb, &dMask))) return rc;
  if ((rc = launch_set_poses(h, (const int*)dIdx, n, (const float*)dPoses, (const uint8_t*)dMask))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return BLCD_OK;
}

// after the streams have drained (the caller has synchronised them): did any environment end the launch sequence with a fault flag?
// (the flag is re-armed)
static int fault_status(blcd_handle h) {
  if (!*h->hFaultAny) return BLCD_OK;
  *h->hFaultAny = 0;
  return fail(BLCD_ERR_ENV_FAULT, "an environment tripped a device guard (NaN state / ellipse outside the span table / island overflow): "
                                  "the step completed; read blcd_get_faults and reset the flagged environments");
}

int blcd_step(blcd_handle h, const float* actions, int32_t n_steps) {
  if (!h || n_steps < 0) return fail(BLCD_ERR_INVALID, "blcd_step: bad arguments");
  if (n_steps == 0) return BLCD_OK;
  if (int rcEnter = enter(h)) return rcEnter;
  const void* dAct;
  int rc;
  if ((rc = in_ptr(h, 0, actions, (size_t)h->N * h->hostScene.nact * sizeof(float), &dAct))) return rc;
  h->lastLaunches = 0;
  if ((rc = rebin_if_due(h))) return rc;
  if ((rc = join_cohort_stream(h))) return rc;
  if ((rc = launch_step(h, (const float*)dAct, n_steps, 0, 0))) return rc;
  h->stepsSinceRebin += n_steps;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipEventElapsedTime(&h->lastMs, h->ev0, h->ev1));
  return fault_status(h);
}

// One env-step and its observation in ONE call with ONE stream synchronisation: the call shape of a policy in the loop
// (research/rl/ppo.py:127-133 `o, r, d, _ = env.step(a)`; async_vector_env.py:191-242 returns the observations with the step).
int blcd_step_obs(blcd_handle h, const float* actions, float* full_state, uint8_t* lcd) {
  if (!h) return fail(BLCD_ERR_INVALID, "blcd_step_obs: bad handle");
  if (int rcEnter = enter(h)) return rcEnter;
  const size_t obsBytes = (size_t)h->N * h->hostScene.nobs * sizeof(float), lcdBytes = (size_t)h->N * h->hostScene.lcdH * h->hostScene.lcdW;
  const void* dAct;
  void *dObs, *dLcd;
  int rc;
  if ((rc = in_ptr(h, 0, actions, (size_t)h->N * h->hostScene.nact * sizeof(float), &dAct))) return rc;
  if ((rc = out_ptr(h, 1, full_state, obsBytes, &dObs))) return rc;
  if ((rc = out_ptr(h, 2, lcd, lcdBytes, &dLcd))) return rc;
  h->lastLaunches = 0;
  if ((rc = rebin_if_due(h))) return rc;
  if ((rc = join_cohort_stream(h))) return rc;
  const bool fused = (h->hostScene.lcdH == 16 || (h->hostScene.lcdH == 32 && h->hostScene.nb > 7)) && h->rolloutChunk > 0 && (dObs || dLcd);
  if (fused) {
    // the rollout's kernel path with T = 1: step_kernel writes the observation row and the frame itself
    if ((rc = launch_step(h, (const float*)dAct, 1, 0, 0, nullptr, nullptr, (long long)h->N * h->hostScene.nact, (uint8_t*)dLcd, (float*)dObs))) return rc;
  } else {
    if ((rc = launch_step(h, (const float*)dAct, 1, 0, 0))) return rc;
    if ((dObs || dLcd) && (rc = launch_obs<float>(h, (float*)dObs, (uint8_t*)dLcd))) return rc;
  }
  h->stepsSinceRebin += 1;
  if (full_state && dObs != full_state) HIPCHK(hipMemcpyAsync(full_state, dObs, obsBytes, hipMemcpyDeviceToHost, h->stream));
  if (lcd && dLcd != lcd) HIPCHK(hipMemcpyAsync(lcd, dLcd, lcdBytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->pending = false;
  HIPCHK(hipEventElapsedTime(&h->lastMs, h->ev0, h->ev1));
  return fault_status(h);
}

int blcd_set_async_stream(blcd_handle h, void* stream, int32_t adopt) {
  if (!h) return fail(BLCD_ERR_INVALID, "blcd_set_async_stream: bad handle");
  HIPCHK(hipSetDevice(h->device));
  if (int rc = drain_async(h)) return rc;
  h->asyncAdopt = adopt != 0;
  h->asyncStream = adopt ? (hipStream_t)stream : nullptr;
  return BLCD_OK;
}

int blcd_step_obs_async(blcd_handle h, const float* actions, float* full_state, uint8_t* lcd) {
  if (!h) return fail(BLCD_ERR_INVALID, "blcd_step_obs_async: bad handle");
  HIPCHK(hipSetDevice(h->device));
  for (const void* p : {(const void*)actions, (const void*)full_state, (const void*)lcd})
    if (p && !is_device_ptr(p)) return fail(BLCD_ERR_INVALID, "blcd_step_obs_async: device buffers only (host buffers need the synchronising blcd_step_obs)");
  // the stream this step is queued on: the handle's own, or the caller's (blcd_set_async_stream) - then the step sits between the
  // caller's kernels with no hand-off between streams at all
  hipStream_t own = h->stream, T = h->asyncAdopt ? h->asyncStream : own;
  if (h->pending && h->pendingStream != T) {   // asynchronous work queued on the other stream first
    if (!h->evExt) HIPCHK(hipEventCreateWithFlags(&h->evExt, hipEventDisableTiming));
    HIPCHK(hipEventRecord(h->evExt, h->pendingStream));
    HIPCHK(hipStreamWaitEvent(T, h->evExt, 0));
  }
  h->pendingStream = T;
  h->pending = true;
  h->lastLaunches = 0;
  h->stream = T;                 // launch_step / launch_obs / cohort 0's sort queue on h->stream
  int rc = rebin_if_due(h);      // (a due sort first drains T on the host: the other cohort sorts on its own stream)
  if (!rc) rc = join_cohort_stream(h);
  const bool fused = (h->hostScene.lcdH == 16 || (h->hostScene.lcdH == 32 && h->hostScene.nb > 7)) && h->rolloutChunk > 0 && (full_state || lcd);
  if (!rc) {
    if (fused) {
      rc = launch_step(h, actions, 1, 0, 0, nullptr, nullptr, (long long)h->N * h->hostScene.nact, lcd, full_state);
    } else {
      rc = launch_step(h, actions, 1, 0, 0);
      if (!rc && (full_state || lcd)) rc = launch_obs<float>(h, full_state, lcd);
    }
  }
  h->stream = own;
  if (rc) return rc;
  h->stepsSinceRebin += 1;
  h->pendingStream = T;          // (rebin_if_due may have drained it: this step is queued behind that)
  h->pending = true;
  return BLCD_OK;   // queued; a fault this step raises is reported by the next synchronising call
}

static int rollout_impl(blcd_handle h, const float* actions, int32_t T, uint8_t* lcd_out, float* obs_out, int lcdBits);
int blcd_rollout(blcd_handle h, const float* actions, int32_t T, uint8_t* lcd_out, float* obs_out) {
  return rollout_impl(h, actions, T, lcd_out, obs_out, 0);
}
int blcd_rollout_bits(blcd_handle h, const float* actions, int32_t T, uint8_t* lcd_bits_out, float* obs_out) {
  if (h && (h->hostScene.lcdW % 8) != 0) return fail(BLCD_ERR_UNSUPPORTED, "blcd_rollout_bits: the LCD width must be a multiple of 8");
  if (h && !((h->hostScene.lcdH == 16 || (h->hostScene.lcdH == 32 && h->hostScene.nb > 7)) && h->rolloutChunk > 0))
    return fail(BLCD_ERR_UNSUPPORTED, "blcd_rollout_bits: only the fused rollout path (16-row frames, or 32-row frames of the large class) packs frames");
  return rollout_impl(h, actions, T, lcd_bits_out, obs_out, 1);
}
static int rollout_impl(blcd_handle h, const float* actions, int32_t T, uint8_t* lcd_out, float* obs_out, int lcdBits) {
  if (!h || T < 1) return fail(BLCD_ERR_INVALID, "blcd_rollout: bad arguments");
  if (int rcEnter = enter(h)) return rcEnter;
  if (int rc = drain_async(h)) return rc;   // the cohorts' streams are not ordered behind an asynchronous step that is still queued
  const int nact = h->hostScene.nact, nobs = h->hostScene.nobs;
  const size_t lcdStep = (size_t)h->N * h->hostScene.lcdH * h->hostScene.lcdW / (lcdBits ? 8 : 1);
  const size_t obsStep = (size_t)h->N * nobs;
  const void* dAct;
  void *dLcd, *dObs;
  int rc;
  if ((rc = in_ptr(h, 0, actions, (size_t)T * h->N * nact * sizeof(float), &dAct))) return rc;
  if ((rc = out_ptr(h, 1, lcd_out, lcdStep * T, &dLcd))) return rc;
  if ((rc = out_ptr(h, 2, obs_out, obsStep * T * sizeof(float), &dObs))) return rc;
  h->lastLaunches = 0;
  while ((int)h->evPool.size() < 2 * T) {
    hipEvent_t ev;
    HIPCHK(hipEventCreate(&ev));
    h->evPool.push_back(ev);
  }
  int nLaunch = 0, fusedSteps = 0;
  bool cohortTimed = false;
  if ((h->hostScene.lcdH == 16 || (h->hostScene.lcdH == 32 && h->hostScene.nb > 7)) && h->rolloutChunk > 0) {
    // fused path: `chunk` env steps per launch, every wave runs its envs through the whole chunk and emits obs/LCD itself;
    // slots are re-binned by work class between chunks
    // Chunk length.  Batches that are re-binned between chunks (joint-free scenes that oversubscribe the SIMDs: work classes
    // drift with impacts and sleep) keep 20.  The others are never re-binned, and every launch ends with all SIMDs waiting for
    // the slowest wave, so longer launches average that tail out (Urchin-50k +8 %, LuxoBall-50k +17 % at one launch per 200-step rollout): as long as the
    // previous rollout's time per env step allows, up to ~1.5 s per launch.  Results do not depend on the chunking.
    int chunk = h->rolloutChunk;
    if (!h->chunkFixed && h->rebinEvery == 0) {
      chunk = 50;
      if (h->estMsPerStep > 0.0f) {
        float c = 1500.0f / h->estMsPerStep;
        chunk = c < 20.0f ? 20 : (c > 200.0f ? 200 : (int)c);
      }
    }
#ifdef BLCD_SCHED
    // the scheduler's progress word keeps the env-step in 16 bits: longer chunks / rollouts would overflow into the sub-step bits
    if ((h->asyncBudget > 0 || h->waveBatch > 0 || h->yieldPasses > 1) && (chunk > 60000 || (h->asyncBudget > 0 && T > 60000)))
      return fail(BLCD_ERR_UNSUPPORTED, "blcd_rollout: the environment-level schedulers handle at most 60000 env-steps per chunk / asynchronous rollout");
#endif
    if (h->asyncBudget > 0) {
      // Asynchronous rollout: no chunk boundaries at all.  Launch after launch, every unfinished environment
      // advances by at most asyncBudget world steps from its own position; the slot sort after each launch puts the suspended
      // ones together.  The loop ends when the device counter of unfinished environments reads zero.
      if ((rc = join_cohort_stream(h))) return rc;
      const size_t po = (size_t)schedWordOffset(h->hostScene.nb, h->hostScene.nj, h->hostScene.np);
      HIPCHK(hipMemsetAsync(h->st + po * (size_t)h->N, 0, 2 * (size_t)h->N * sizeof(float), h->stream));   // progress words: everyone at step 0
      HIPCHK(hipEventRecord(h->evPool[0], h->stream));
      const int minLaunches = (T * h->hostScene.substeps + h->asyncBudget - 1) / h->asyncBudget;
      unsigned long long unfinished = 1;
      int launches = 0;
      while (unfinished) {
        HIPCHK(hipMemsetAsync(h->dSchedStats + 3, 0, sizeof(unsigned long long), h->stream));
        if ((rc = launch_step(h, (const float*)dAct, T, 0, 0, nullptr, nullptr, (long long)h->N * nact, (uint8_t*)dLcd, (float*)dObs, -1, 1, 1, lcdBits,
                              h->asyncBudget)))
          return rc;
        ++launches;
        if (launches >= minLaunches) {
          HIPCHK(hipMemcpyAsync(&unfinished, h->dSchedStats + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
          HIPCHK(hipStreamSynchronize(h->stream));
          if (launches > 64 * minLaunches + 64) return fail(BLCD_ERR_HIP, "blcd_rollout: the asynchronous scheduler made no progress");
        }
        if (unfinished && (rc = launch_rebin(h, true, T))) return rc;
      }
      HIPCHK(hipEventRecord(h->evPool[1], h->stream));
      h->asyncLaunches += launches;
      h->lastLaunches = 1;
      nLaunch = 1;
      cohortTimed = true;
    } else if (h->nCohorts > 1) {
      // two cohorts, two streams, no barrier between chunks: cohort k's chunk i + 1 follows its own chunk i (and its own
      // re-bin) only.  Timed as ONE sequence on the handle's stream: e0 before the first launch, e1 after the other streams have joined.
      HIPCHK(hipEventRecord(h->evPool[0], h->stream));
      for (int k = 1; k < h->nCohorts; ++k) HIPCHK(hipStreamWaitEvent(h->cstream[k], h->evPool[0], 0));   // staged inputs (copied on the handle's stream) first
      for (int t = 0, c = 0; t < T; t += c) {
        const int chunkNow = (h->phase0Chunk > 0 && h->stepsSinceFullReset < h->phase0Steps) ? h->phase0Chunk : chunk;
        c = T - t < chunkNow ? T - t : chunkNow;
        const float* a = dAct ? (const float*)dAct + (size_t)t * h->N * nact : nullptr;
        if ((rc = rebin_if_due(h, c))) return rc;
        h->stepsSinceFullReset += c;
        for (int pass = 0; pass < h->yieldPasses; ++pass) {
          if (pass > 0 && (rc = launch_rebin(h, true))) return rc;      // suspended environments to the front, in dense waves
          for (int k = 0; k < h->nCohorts; ++k)
            if ((rc = launch_step(h, a, c, 0, 0, nullptr, nullptr, (long long)h->N * nact, dLcd ? (uint8_t*)dLcd + lcdStep * t : nullptr,
                                  dObs ? (float*)dObs + obsStep * t : nullptr, k, pass, h->yieldPasses, lcdBits)))
              return rc;
        }
        h->stepsSinceRebin += c;
        ++nLaunch;
      }
      for (int k = 1; k < h->nCohorts; ++k) {
        HIPCHK(hipEventRecord(h->evJoin[k], h->cstream[k]));
        HIPCHK(hipStreamWaitEvent(h->stream, h->evJoin[k], 0));
      }
      HIPCHK(hipEventRecord(h->evPool[1], h->stream));
      h->lastLaunches = nLaunch;
      cohortTimed = true;
    } else {
    for (int t = 0; t < T; t += chunk) {
      int c = T - t < chunk ? T - t : chunk;
      const float* a = dAct ? (const float*)dAct + (size_t)t * h->N * nact : nullptr;
      if ((rc = rebin_if_due(h))) return rc;
      for (int pass = 0; pass < h->yieldPasses; ++pass) {
        if (pass > 0 && (rc = launch_rebin(h, true))) return rc;
        if ((rc = launch_step(h, a, c, 0, 0, h->evPool[2 * nLaunch], h->evPool[2 * nLaunch + 1], (long long)h->N * nact,
                              dLcd ? (uint8_t*)dLcd + lcdStep * t : nullptr, dObs ? (float*)dObs + obsStep * t : nullptr, -1, pass, h->yieldPasses, lcdBits)))
          return rc;
      }
      h->stepsSinceRebin += c;
      ++nLaunch;
    }
    }
    fusedSteps = T;
  } else {
    for (int t = 0; t < T; ++t) {
      const float* a = dAct ? (const float*)dAct + (size_t)t * h->N * nact : nullptr;
      if ((rc = rebin_if_due(h))) return rc;
      if ((rc = join_cohort_stream(h))) return rc;
      if ((rc = launch_step(h, a, 1, 0, 0, h->evPool[2 * t], h->evPool[2 * t + 1]))) return rc;
      h->stepsSinceRebin += 1;
      ++nLaunch;
      if (dLcd || dObs) {
        if ((rc = launch_obs<float>(h, dObs ? (float*)dObs + obsStep * t : nullptr, dLcd ? (uint8_t*)dLcd + lcdStep * t : nullptr)))
          return rc;
      }
    }
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int c = 1; c < h->nCohorts; ++c) HIPCHK(hipStreamSynchronize(h->cstream[c]));
  h->cohortsBusy = false;
  float total = 0.0f;  // step-kernel time only: each event pair brackets one step_kernel launch on this stream
  if (cohortTimed) {   // cohorts: the whole overlapped sequence (both cohorts' launches and their re-bins), start to join
    HIPCHK(hipEventElapsedTime(&total, h->evPool[0], h->evPool[1]));
  } else {
    for (int t = 0; t < nLaunch; ++t) {
      float ms = 0.0f;
      HIPCHK(hipEventElapsedTime(&ms, h->evPool[2 * t], h->evPool[2 * t + 1]));
      total += ms;
    }
  }
  h->lastMs = total;
  if (fusedSteps > 0) h->estMsPerStep = total / (float)fusedSteps;
  if ((rc = out_done(h, 1, lcd_out, lcdStep * T, dLcd))) return rc;
  if ((rc = out_done(h, 2, obs_out, obsStep * T * sizeof(float), dObs))) return rc;
  return fault_status(h);
}

int blcd_get_obs(blcd_handle h, void* full_state, int32_t dtype, uint8_t* lcd) {
  if (!h) return fail(BLCD_ERR_INVALID, "blcd_get_obs: bad handle");
  if (dtype != 0 && dtype != 1) return fail(BLCD_ERR_INVALID, "blcd_get_obs: dtype must be 0 (f32) or 1 (f64)");
  if (int rcEnter = enter(h)) return rcEnter;
  const size_t obsBytes = (size_t)h->N * h->hostScene.nobs * (dtype ? 8 : 4);
  const size_t lcdBytes = (size_t)h->N * h->hostScene.lcdH * h->hostScene.lcdW;
  void *dObs, *dLcd;
  int rc;
  if ((rc = out_ptr(h, 1, full_state, obsBytes, &dObs))) return rc;
  if ((rc = out_ptr(h, 2, lcd, lcdBytes, &dLcd))) return rc;
  if (dtype) rc = launch_obs<double>(h, (double*)dObs, (uint8_t*)dLcd);
  else rc = launch_obs<float>(h, (float*)dObs, (uint8_t*)dLcd);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  if ((rc = out_done(h, 1, full_state, obsBytes, dObs))) return rc;
  if ((rc = out_done(h, 2, lcd, lcdBytes, dLcd))) return rc;
  return BLCD_OK;
}

static int goal_buffers(blcd_handle h) {
  const size_t nobs = h->hostScene.nobs, lcdBytes = (size_t)h->hostScene.lcdH * h->hostScene.lcdW;
  if (!h->dGoal) {
    HIPCHK(hipMalloc(&h->dGoal, sizeof(blcd_goal_desc)));
    HIPCHK(hipMalloc(&h->goalFs, (size_t)h->N * nobs * sizeof(double)));
    HIPCHK(hipMalloc(&h->goalObs, (size_t)h->N * nobs * sizeof(double)));
    HIPCHK(hipMalloc(&h->goalLast, (size_t)h->N * sizeof(double)));
    HIPCHK(hipMalloc(&h->goalLcd, (size_t)h->N * lcdBytes));
    HIPCHK(hipMalloc(&h->goalCurLcd, (size_t)h->N * lcdBytes));
    HIPCHK(hipMemsetAsync(h->goalFs, 0, (size_t)h->N * nobs * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->goalLast, 0, (size_t)h->N * sizeof(double), h->stream));
    HIPCHK(hipMemsetAsync(h->goalLcd, 0, (size_t)h->N * lcdBytes, h->stream));
  }
  return BLCD_OK;
}

int blcd_goal_set(blcd_handle h, const blcd_goal_desc* g, const int32_t* idxs, int32_t n, const double* goal_full_state,
                  const uint8_t* goal_lcd) {
  if (!h || !g || !goal_full_state || n < 1 || n > h->N) return fail(BLCD_ERR_INVALID, "blcd_goal_set: bad arguments");
  if (g->mode != 0 && g->mode != 1) return fail(BLCD_ERR_INVALID, "blcd_goal_set: mode must be 0 or 1");
  if (g->mode == 0 && (g->n_idx < 1 || g->n_idx > BLCD_MAX_OBS))
    return fail(BLCD_ERR_INVALID, "blcd_goal_set: n_idx out of range");
  for (int i = 0; i < (g->mode == 0 ? g->n_idx : 0); ++i)
    if (g->idxs[i] < 0 || g->idxs[i] >= h->hostScene.nobs) return fail(BLCD_ERR_INVALID, "blcd_goal_set: index outside full_state");
  if (g->mode == 1 && !goal_lcd) return fail(BLCD_ERR_INVALID, "blcd_goal_set: mode 1 needs goal_lcd");
  if (int rcEnter = enter(h)) return rcEnter;
  int rc;
  if ((rc = goal_buffers(h))) return rc;
  h->goal = *g;
  HIPCHK(hipMemcpyAsync(h->dGoal, &h->goal, sizeof(blcd_goal_desc), hipMemcpyHostToDevice, h->stream));
  const size_t nobs = h->hostScene.nobs, lcdBytes = (size_t)h->hostScene.lcdH * h->hostScene.lcdW;
  const void *dIdx, *dFs, *dLcd;
  if ((rc = in_ptr(h, 0, idxs, (size_t)n * sizeof(int32_t), &dIdx))) return rc;
  if ((rc = in_ptr(h, 1, goal_full_state, (size_t)n * nobs * sizeof(double), &dFs))) return rc;
  if ((rc = in_ptr(h, 2, goal_lcd, (size_t)n * lcdBytes, &dLcd))) return rc;
  hipLaunchKernelGGL(goal_scatter_kernel, dim3(n), dim3(64), 0, h->stream, (const int*)dIdx, n, h->N, (int)nobs, (int)lcdBytes,
                     (const double*)dFs, (const uint8_t*)dLcd, h->goalFs, h->goalLcd);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  h->goalSet = true;
  return BLCD_OK;
}

static int goal_launch(blcd_handle h, const int* dIdx, int n, int seedOnly, double* dRew, uint8_t* dDone, double* dDelta) {
  int rc;
  const bool needLcd = h->goal.mode == 1;
  if ((rc = launch_obs<double>(h, needLcd ? nullptr : h->goalObs, needLcd ? h->goalCurLcd : nullptr))) return rc;
  const int lcdBytes = h->hostScene.lcdH * h->hostScene.lcdW;
  hipLaunchKernelGGL(goal_eval_kernel, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->dGoal, h->N, h->hostScene.nobs, lcdBytes, dIdx,
                     n, seedOnly, h->goalObs, h->goalCurLcd, h->goalFs, h->goalLcd, h->goalLast, dRew, dDone, dDelta);
  HIPCHK(hipGetLastError());
  return BLCD_OK;
}

// ---- 1-bit transport of LCD frames (the frames are bool arrays; eight pixels travel as one byte) -----------
__global__ void pack_bits_kernel(const uint2* __restrict__ src, uint8_t* __restrict__ dst, long long n8) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint2 v = src[i];   // 8 pixels, one byte each (0 / 1); bit k of the result = pixel k
  const uint32_t lo = ((v.x & 0x01010101u) * 0x10204080u) >> 28, hi = ((v.y & 0x01010101u) * 0x10204080u) >> 28;
  dst[i] = (uint8_t)(lo | (hi << 4));
}
__global__ void unpack_bits_kernel(const uint8_t* __restrict__ src, uint2* __restrict__ dst, long long n8) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint32_t b = src[i];
  uint2 v;
  v.x = ((b & 0xfu) * 0x00204081u) & 0x01010101u;
  v.y = ((b >> 4) * 0x00204081u) & 0x01010101u;
  dst[i] = v;
}
int blcd_pack_bits(const uint8_t* src, uint8_t* dst, int64_t n_pixels, void* stream) {
  if (!src || !dst || n_pixels < 8 || (n_pixels & 7) || ((uintptr_t)src & 7)) return fail(BLCD_ERR_INVALID, "blcd_pack_bits: bad arguments");
  const long long n8 = n_pixels / 8;
  hipLaunchKernelGGL(pack_bits_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint2*)src, dst, n8);
  HIPCHK(hipGetLastError());
  return BLCD_OK;
}
int blcd_unpack_bits(const uint8_t* src, uint8_t* dst, int64_t n_pixels, void* stream) {
  if (!src || !dst || n_pixels < 8 || (n_pixels & 7) || ((uintptr_t)dst & 7)) return fail(BLCD_ERR_INVALID, "blcd_unpack_bits: bad arguments");
  const long long n8 = n_pixels / 8;
  hipLaunchKernelGGL(unpack_bits_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (uint2*)dst, n8);
  HIPCHK(hipGetLastError());
  return BLCD_OK;
}

// ---- lcd_render(width, height, lcd_mode) --------------------------------------------------------------------------
static std::vector<uint8_t> g_lutHost;   // process-wide: Pillow's ellipse fill/outline span table (DATA, boxlcd_amd/ellipse_rgb_lut.bin)
static int g_lutAmax = -1;
static std::mutex g_lutMutex;

int blcd_set_ellipse_rgb_lut(const uint8_t* lut, int32_t amax) {
  if (!lut || amax < 0 || amax > 254) return fail(BLCD_ERR_INVALID, "blcd_set_ellipse_rgb_lut: bad arguments");
  std::lock_guard<std::mutex> lock(g_lutMutex);
  g_lutHost.assign(lut, lut + (size_t)(amax + 1) * 5 * (amax + 3) * 6);
  g_lutAmax = amax;
  return BLCD_OK;
}

int blcd_render_poses_ex(blcd_handle h, const float* poses, const int32_t* shape_sel, int32_t m, int32_t width, int32_t height,
                         int32_t mode, uint8_t* out) {
  if (!h || !poses || !out || m < 1 || width < 1 || height < 1 || width > 4096 || height > 4096 || (mode != 0 && mode != 1))
    return fail(BLCD_ERR_INVALID, "blcd_render_poses_ex: bad arguments");
  if (int rcEnter = enter(h)) return rcEnter;
  {
    std::lock_guard<std::mutex> lock(g_lutMutex);
    if (g_lutAmax < 0) return fail(BLCD_ERR_INVALID, "blcd_render_poses_ex: call blcd_set_ellipse_rgb_lut first (boxlcd_amd/ellipse_rgb_lut.bin)");
    if (h->lutAmax != g_lutAmax) {
      if (h->dLut) (void)hipFree(h->dLut);
      h->dLut = nullptr;
      HIPCHK(hipMalloc((void**)&h->dLut, g_lutHost.size()));
      HIPCHK(hipMemcpy(h->dLut, g_lutHost.data(), g_lutHost.size(), hipMemcpyHostToDevice));
      h->lutAmax = g_lutAmax;
    }
  }
  if (!h->dErr) HIPCHK(hipMalloc((void**)&h->dErr, sizeof(int)));
  HIPCHK(hipMemsetAsync(h->dErr, 0, sizeof(int), h->stream));
  const int nb = h->hostScene.nb;
  const size_t bytes = (size_t)m * height * width * (mode ? 3 : 1);
  const void *dPoses, *dSel;
  void* dOut;
  int rc;
  if ((rc = in_ptr(h, 0, poses, (size_t)m * nb * 3 * sizeof(float), &dPoses))) return rc;
  if ((rc = in_ptr(h, 1, shape_sel, (size_t)m * nb * sizeof(int32_t), &dSel))) return rc;
  if ((rc = out_ptr(h, 2, out, bytes, &dOut))) return rc;
  hipLaunchKernelGGL(render_ex_kernel, dim3((unsigned)m), dim3(64), 0, h->stream, h->dScene, m, (const float*)dPoses, (const int*)dSel,
                     width, height, mode, (const uint8_t*)h->dLut, h->lutAmax, (uint8_t*)dOut, h->dErr);
  HIPCHK(hipGetLastError());
  int err = 0;
  HIPCHK(hipMemcpyAsync(&err, h->dErr, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if ((rc = out_done(h, 2, out, bytes, dOut))) return rc;
  if (err) return fail(BLCD_ERR_INVALID, "blcd_render_poses_ex: a circle's bounding box is outside Pillow's span table (too large a canvas)");
  return BLCD_OK;
}

int blcd_goal_seed(blcd_handle h, const int32_t* idxs, int32_t n) {
  if (!h || !h->goalSet) return fail(BLCD_ERR_INVALID, "blcd_goal_seed: no goal installed (blcd_goal_set)");
  if (!idxs) n = h->N;
  if (n < 1 || n > h->N) return fail(BLCD_ERR_INVALID, "blcd_goal_seed: bad count");
  if (int rcEnter = enter(h)) return rcEnter;
  const void* dIdx;
  int rc;
  if ((rc = in_ptr(h, 0, idxs, (size_t)n * sizeof(int32_t), &dIdx))) return rc;
  if ((rc = goal_launch(h, (const int*)dIdx, n, 1, nullptr, nullptr, nullptr))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return BLCD_OK;
}

int blcd_goal_eval(blcd_handle h, double* rew, uint8_t* done, double* delta) {
  if (!h || !h->goalSet) return fail(BLCD_ERR_INVALID, "blcd_goal_eval: no goal installed (blcd_goal_set)");
  if (int rcEnter = enter(h)) return rcEnter;
  void *dRew, *dDone, *dDelta;
  int rc;
  if ((rc = out_ptr(h, 0, rew, (size_t)h->N * sizeof(double), &dRew))) return rc;
  if ((rc = out_ptr(h, 1, done, (size_t)h->N, &dDone))) return rc;
  if ((rc = out_ptr(h, 2, delta, (size_t)h->N * sizeof(double), &dDelta))) return rc;
  if ((rc = goal_launch(h, nullptr, h->N, 0, (double*)dRew, (uint8_t*)dDone, (double*)dDelta))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  if ((rc = out_done(h, 0, rew, (size_t)h->N * sizeof(double), dRew))) return rc;
  if ((rc = out_done(h, 1, done, (size_t)h->N, dDone))) return rc;
  if ((rc = out_done(h, 2, delta, (size_t)h->N * sizeof(double), dDelta))) return rc;
  return BLCD_OK;
}

int blcd_render_poses(blcd_handle h, const float* poses, const int32_t* shape_sel, int32_t m, uint8_t* lcd) {
  if (!h || !poses || !lcd || m < 1) return fail(BLCD_ERR_INVALID, "blcd_render_poses: bad arguments");
  if (int rcEnter = enter(h)) return rcEnter;
  const int nb = h->hostScene.nb;
  const size_t lcdBytes = (size_t)m * h->hostScene.lcdH * h->hostScene.lcdW;
  const void *dPoses, *dSel;
  void* dLcd;
  int rc;
  if ((rc = in_ptr(h, 0, poses, (size_t)m * nb * 3 * sizeof(float), &dPoses))) return rc;
  if ((rc = in_ptr(h, 1, shape_sel, (size_t)m * nb * sizeof(int32_t), &dSel))) return rc;
  if ((rc = out_ptr(h, 2, lcd, lcdBytes, &dLcd))) return rc;
  dim3 grid((m + 63) / 64), block(64);
  if (h->hostScene.lcdH == 16)
    hipLaunchKernelGGL((render_poses_kernel<16, uint32_t>), grid, block, 0, h->stream, h->dScene, m, (const float*)dPoses,
                       (const int*)dSel, (uint8_t*)dLcd);
  else
    hipLaunchKernelGGL((render_poses_kernel<32, uint64_t>), grid, block, 0, h->stream, h->dScene, m, (const float*)dPoses,
                       (const int*)dSel, (uint8_t*)dLcd);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return out_done(h, 2, lcd, lcdBytes, dLcd);
}

int blcd_get_poses(blcd_handle h, float* poses) {
  if (!h || !poses) return fail(BLCD_ERR_INVALID, "blcd_get_poses: bad arguments");
  if (int rcEnter = enter(h)) return rcEnter;
  const size_t bytes = (size_t)h->N * h->hostScene.nb * 4 * sizeof(float);
  void* d;
  int rc;
  if ((rc = out_ptr(h, 1, poses, bytes, &d))) return rc;
  hipLaunchKernelGGL(poses_kernel, dim3((h->N + 63) / 64), dim3(64), 0, h->stream, h->dScene, h->st, h->N, h->eid, (float*)d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return out_done(h, 1, poses, bytes, d);
}

int blcd_get_shape_sel(blcd_handle h, int32_t* shape_sel) {
  if (!h || !shape_sel) return fail(BLCD_ERR_INVALID, "blcd_get_shape_sel: bad arguments");
  if (int rcEnter = enter(h)) return rcEnter;
  const size_t bytes = (size_t)h->N * h->hostScene.nb * sizeof(int32_t);
  void* d;
  int rc;
  if ((rc = out_ptr(h, 1, shape_sel, bytes, &d))) return rc;
  hipLaunchKernelGGL(shape_sel_kernel, dim3((h->N + 63) / 64), dim3(64), 0, h->stream, h->dScene, h->st, h->N, h->eid, (int*)d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return out_done(h, 1, shape_sel, bytes, d);
}

// snapshot = header | state in slot order | slot -> env id table | reset counts of the device sampler (per env id)
struct StateHeader {
  uint32_t magic, version;
  int32_t n, nb, nj, np;
  uint64_t words;
  uint64_t envIdBase;
};
int blcd_get_state(blcd_handle h, void* blob, size_t* size) {
  if (!h || !size) return fail(BLCD_ERR_INVALID, "blcd_get_state: bad arguments");
  const size_t stBytes = h->words * (size_t)h->N * sizeof(float), idBytes = (size_t)h->N * sizeof(int);
  size_t need = sizeof(StateHeader) + stBytes + 2 * idBytes;
  if (!blob) {
    *size = need;
    return BLCD_OK;
  }
  if (*size < need) return fail(BLCD_ERR_INVALID, "blcd_get_state: buffer too small");
  if (int rcEnter = enter(h)) return rcEnter;
  StateHeader hd = {0x44434c42u, BLCD_VERSION, h->N, h->hostScene.nb, h->hostScene.nj, h->hostScene.np, (uint64_t)h->words, h->envIdBase};
  std::memcpy(blob, &hd, sizeof(hd));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy((char*)blob + sizeof(hd), h->st, stBytes, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy((char*)blob + sizeof(hd) + stBytes, h->eid, idBytes, hipMemcpyDeviceToHost));
  if (h->dEpisode) HIPCHK(hipMemcpy((char*)blob + sizeof(hd) + stBytes + idBytes, h->dEpisode, idBytes, hipMemcpyDeviceToHost));
  else std::memset((char*)blob + sizeof(hd) + stBytes + idBytes, 0, idBytes);
  *size = need;
  return BLCD_OK;
}
int blcd_set_state(blcd_handle h, const void* blob, size_t size) {
  if (!h || !blob) return fail(BLCD_ERR_INVALID, "blcd_set_state: bad arguments");
  const size_t stBytes = h->words * (size_t)h->N * sizeof(float), idBytes = (size_t)h->N * sizeof(int);
  size_t need = sizeof(StateHeader) + stBytes + 2 * idBytes;
  if (size < sizeof(StateHeader)) return fail(BLCD_ERR_INVALID, "blcd_set_state: size mismatch");
  StateHeader hd;
  std::memcpy(&hd, blob, sizeof(hd));
  if (hd.magic != 0x44434c42u) return fail(BLCD_ERR_INVALID, "blcd_set_state: not a boxlcd snapshot");
  if (hd.version != BLCD_VERSION) return fail(BLCD_ERR_INVALID, "blcd_set_state: snapshot written by another library version");
  if (size != need || hd.n != h->N || hd.nb != h->hostScene.nb || hd.nj != h->hostScene.nj || hd.np != h->hostScene.np || hd.words != h->words)
    return fail(BLCD_ERR_INVALID, "blcd_set_state: snapshot belongs to a different scene/batch/build");
  if (int rcEnter = enter(h)) return rcEnter;
  int rc;
  if ((rc = sampler_buffers(h))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(h->st, (const char*)blob + sizeof(hd), stBytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->eid, (const char*)blob + sizeof(hd) + stBytes, idBytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->dEpisode, (const char*)blob + sizeof(hd) + stBytes + idBytes, idBytes, hipMemcpyHostToDevice));
  h->envIdBase = hd.envIdBase;
  hipLaunchKernelGGL(invert_kernel, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, h->eid, h->slotOf, h->N);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int& b : h->sortedBlocks) b = 0;
  return BLCD_OK;
}

int blcd_get_faults(blcd_handle h, int32_t* flags) {
  if (!h || !flags) return fail(BLCD_ERR_INVALID, "blcd_get_faults: bad arguments");
  if (int rcEnter = enter(h)) return rcEnter;
  const size_t bytes = (size_t)h->N * sizeof(int32_t);
  int rc = ensure_stage(h, 3, bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(faults_kernel, dim3((h->N + 63) / 64), dim3(64), 0, h->stream, h->dScene, h->st, h->N, h->eid, (int*)h->stage[3]);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(flags, h->stage[3], bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return BLCD_OK;
}

int blcd_sync(blcd_handle h) {
  if (!h) return fail(BLCD_ERR_INVALID, "blcd_sync: bad handle");
  if (int rcEnter = enter(h)) return rcEnter;
  HIPCHK(hipStreamSynchronize(h->stream));
  h->pending = false;              // enter() has put whatever was pending behind the handle's stream
  return BLCD_OK;
}
void* blcd_stream(blcd_handle h) { return h ? (void*)h->stream : nullptr; }

int blcd_last_kernel_ms(blcd_handle h, float* ms, int32_t* launches) {
  if (!h) return fail(BLCD_ERR_INVALID, "blcd_last_kernel_ms: bad handle");
  if (ms) *ms = h->lastMs;
  if (launches) *launches = h->lastLaunches;
  return BLCD_OK;
}

int blcd_sched_stats(blcd_handle h, uint64_t* out8) {
  if (!h || !out8) return fail(BLCD_ERR_INVALID, "blcd_sched_stats: bad arguments");
  if (int rcEnter = enter(h)) return rcEnter;
  int rc;
  if ((rc = join_cohort_stream(h))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(out8, h->dSchedStats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(h->dSchedStats, 0, 8 * sizeof(uint64_t)));
  out8[3] = (uint64_t)h->yieldPasses;
  out8[7] = (uint64_t)h->yieldMaxLanes;
  return BLCD_OK;
}

int blcd_debug_wave_times(blcd_handle h, uint64_t* out, int32_t cap) {
  if (!h || !out) return fail(BLCD_ERR_INVALID, "blcd_debug_wave_times: bad arguments");
  if (!h->waveTimes) return fail(BLCD_ERR_UNSUPPORTED, "build with BLCD_DEFS=-DBLCD_WAVETIMES and set BLCD_WAVETIMES=1 before blcd_create");
  int nw = (h->N + h->lanes - 1) / h->lanes;
  if (nw * 9 > cap) nw = cap / 9;
  if (int rcEnter = enter(h)) return rcEnter;
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(out, h->waveTimes, (size_t)nw * 9 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIPCHK(hipMemset(h->waveTimes, 0, (size_t)nw * 9 * sizeof(uint64_t)));
  return nw;
}
int blcd_debug_world_step(blcd_handle h, int32_t n) {
  if (!h || n < 0) return fail(BLCD_ERR_INVALID, "blcd_debug_world_step: bad arguments");
  if (int rcEnter = enter(h)) return rcEnter;
  int rc = launch_step(h, nullptr, 0, n, 0);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return BLCD_OK;
}
int blcd_debug_set_motor_speeds(blcd_handle h, const float* actions) {
  if (!h) return fail(BLCD_ERR_INVALID, "blcd_debug_set_motor_speeds: bad handle");
  if (int rcEnter = enter(h)) return rcEnter;
  const void* dAct;
  int rc;
  if ((rc = in_ptr(h, 0, actions, (size_t)h->N * h->hostScene.nact * sizeof(float), &dAct))) return rc;
  if ((rc = launch_step(h, (const float*)dAct, 0, 0, 1))) return rc;
  HIPCHK(hipStreamSynchronize(h->stream));
  return BLCD_OK;
}
int blcd_debug_dump(blcd_handle h, float* bodies, float* joints, float* pairs) {
  if (!h || !bodies || !joints || !pairs) return fail(BLCD_ERR_INVALID, "blcd_debug_dump: bad arguments");
  if (int rcEnter = enter(h)) return rcEnter;
  const DevScene& S = h->hostScene;
  size_t bb = (size_t)h->N * S.nb * BLCD_BODY_STATE_FLOATS * 4;
  size_t jb = (size_t)h->N * (S.nj > 0 ? S.nj : 1) * BLCD_JOINT_STATE_FLOATS * 4;
  size_t pb = (size_t)h->N * (S.np > 0 ? S.np : 1) * BLCD_PAIR_STATE_FLOATS * 4;
  int rc;
  if ((rc = ensure_stage(h, 1, bb))) return rc;
  if ((rc = ensure_stage(h, 2, jb))) return rc;
  if ((rc = ensure_stage(h, 3, pb))) return rc;
  hipLaunchKernelGGL(dump_kernel, dim3((h->N + 63) / 64), dim3(64), 0, h->stream, h->dScene, h->st, h->N, h->eid, (float*)h->stage[1],
                     (float*)h->stage[2], (float*)h->stage[3]);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(bodies, h->stage[1], bb, hipMemcpyDeviceToHost, h->stream));
  if (S.nj > 0) HIPCHK(hipMemcpyAsync(joints, h->stage[2], (size_t)h->N * S.nj * BLCD_JOINT_STATE_FLOATS * 4, hipMemcpyDeviceToHost, h->stream));
  if (S.np > 0) HIPCHK(hipMemcpyAsync(pairs, h->stage[3], (size_t)h->N * S.np * BLCD_PAIR_STATE_FLOATS * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return BLCD_OK;
}

int blcd_debug_sincos(const float* x, int64_t n, float* s, float* c, int32_t device) {
  if (!x || !s || !c || n < 1) return fail(BLCD_ERR_INVALID, "blcd_debug_sincos: bad arguments");
  if (blcd_device_count() <= 0) return fail(BLCD_ERR_NO_DEVICE, "no HIP device available");
  HIPCHK(hipSetDevice(device));
  float *dx, *ds, *dc;
  HIPCHK(hipMalloc((void**)&dx, n * 4));
  HIPCHK(hipMalloc((void**)&ds, n * 4));
  HIPCHK(hipMalloc((void**)&dc, n * 4));
  HIPCHK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(sincos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, n, ds, dc);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(s, ds, n * 4, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(c, dc, n * 4, hipMemcpyDeviceToHost));
  (void)hipFree(dx);
  (void)hipFree(ds);
  (void)hipFree(dc);
  return BLCD_OK;
}

// The narrow phase by itself: n independent (shape A, pose A, shape B, pose B) configurations through the DEVICE routines, dispatched
// like b2Contact's s_registers.  tests/test_gpu_parity.py compares the manifolds bit for bit with the oracle's generic routines on
// tens of thousands of random near-contact configurations - branches (vertex regions, polygon reference faces, clipped points)
// that rollouts visit rarely.  An edge stands for an arena wall: its body must sit at the origin with angle 0 (what
// blcd_collide_wall.h is written for).
struct CollideCase {
  Shape a, b;
  Transform xa, xb;
  int swapped;
};
__global__ void collide_kernel(const CollideCase* __restrict__ cases, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const CollideCase& c = cases[i];
  Manifold m;
  m.pointCount = 0;
  m.type = 0;
  m.localNormal = m.localPoint = V2(0.0f, 0.0f);
  for (int j = 0; j < 2; ++j) {
    m.points[j].localPoint = V2(0.0f, 0.0f);
    m.points[j].normalImpulse = m.points[j].tangentImpulse = 0.0f;
    m.points[j].id.key = 0;
  }
  if (c.a.type == kCircle) CollideCircles(&m, &c.a, c.xa, &c.b, c.xb);
  else if (c.a.type == kPolygon && c.b.type == kCircle) CollidePolygonAndCircle(&m, &c.a, c.xa, &c.b, c.xb);
  else if (c.a.type == kPolygon) CollidePolygons(&m, &c.a, c.xa, &c.b, c.xb);
  else {
    const WallK w = MakeWallK(c.a.v[0], c.a.v[1], c.a.radius);
    if (c.b.type == kCircle) CollideWallCircle(&m, w, c.b.v[0], c.b.radius, c.xb);
    else CollideWallPolygon(&m, w, &c.b, c.xb);
  }
  float* o = out + (size_t)i * 24;
  for (int k = 0; k < 24; ++k) o[k] = 0.0f;
  o[0] = (float)m.pointCount;
  o[1] = (float)m.type;
  o[2] = m.localNormal.x; o[3] = m.localNormal.y; o[4] = m.localPoint.x; o[5] = m.localPoint.y;
  for (int j = 0; j < m.pointCount && j < 2; ++j) {
    o[6 + 3 * j] = m.points[j].localPoint.x; o[7 + 3 * j] = m.points[j].localPoint.y; o[8 + 3 * j] = (float)m.points[j].id.key;
  }
  if (m.pointCount > 0) {
    WorldManifold wm;
    wm.Initialize(&m, c.xa, c.a.radius, c.xb, c.b.radius);
    o[12] = wm.normal.x; o[13] = wm.normal.y;
    for (int j = 0; j < m.pointCount && j < 2; ++j) {
      o[14 + 3 * j] = wm.points[j].x; o[15 + 3 * j] = wm.points[j].y;   // [16 + 3 j]: separation, not computed by the product
    }
  }
  o[20] = (float)c.swapped;
}
static bool shape_from_spec(const float* sp, Shape* s) {   // {0, r} circle | {1, hx, hy} box | {2, x1, y1, x2, y2} edge | {3, n, x0, y0, ...} polygon
  const int kind = (int)sp[0];
  *s = Shape{};
  if (kind == 0) ShapeSetCircle(s, sp[1]);
  else if (kind == 1) ShapeSetAsBox(s, sp[1], sp[2]);
  else if (kind == 2) ShapeSetEdge(s, V2(sp[1], sp[2]), V2(sp[3], sp[4]));
  else if (kind == 3 && (int)sp[1] >= 3 && (int)sp[1] <= kShapeVerts) {
    Vec2 vs[kShapeVerts];
    const int nv = (int)sp[1];
    for (int i = 0; i < nv; ++i) vs[i] = V2(sp[2 + 2 * i], sp[3 + 2 * i]);
    ShapeSetPolygon(s, vs, nv);
  } else {
    return false;
  }
  return true;
}
int blcd_debug_collide(int32_t device, int32_t n, const float* specA, const float* poseA, const float* specB, const float* poseB, float* out) {
  if (n < 1 || !specA || !poseA || !specB || !poseB || !out) return fail(BLCD_ERR_INVALID, "blcd_debug_collide: bad arguments");
  if (blcd_device_count() <= 0) return fail(BLCD_ERR_NO_DEVICE, "no HIP device available");
  std::vector<CollideCase> cases((size_t)n);
  for (int i = 0; i < n; ++i) {
    CollideCase& c = cases[(size_t)i];
    if (!shape_from_spec(specA + (size_t)i * BLCD_COLLIDE_SPEC_FLOATS, &c.a) || !shape_from_spec(specB + (size_t)i * BLCD_COLLIDE_SPEC_FLOATS, &c.b))
      return fail(BLCD_ERR_INVALID, "blcd_debug_collide: malformed shape spec");
    c.xa.p = V2(poseA[3 * i], poseA[3 * i + 1]);
    c.xa.q.Set(poseA[3 * i + 2]);
    c.xb.p = V2(poseB[3 * i], poseB[3 * i + 1]);
    c.xb.q.Set(poseB[3 * i + 2]);
    auto rank = [](const Shape& s) { return s.type == kEdge ? 0 : (s.type == kPolygon ? 1 : 2); };
    c.swapped = 0;
    if (rank(c.a) > rank(c.b)) {
      std::swap(c.a, c.b);
      std::swap(c.xa, c.xb);
      c.swapped = 1;
    }
    if (c.a.type == kEdge && (c.b.type == kEdge || c.xa.p.x != 0.0f || c.xa.p.y != 0.0f || c.xa.q.s != 0.0f || c.xa.q.c != 1.0f))
      return fail(BLCD_ERR_UNSUPPORTED, "blcd_debug_collide: an edge is an arena wall - its body sits at the origin with angle 0, and walls do not meet walls");
  }
  HIPCHK(hipSetDevice(device));
  CollideCase* dCases = nullptr;
  float* dOut = nullptr;
  HIPCHK(hipMalloc((void**)&dCases, (size_t)n * sizeof(CollideCase)));
  HIPCHK(hipMalloc((void**)&dOut, (size_t)n * 24 * sizeof(float)));
  HIPCHK(hipMemcpy(dCases, cases.data(), (size_t)n * sizeof(CollideCase), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(collide_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, (const CollideCase*)dCases, (int)n, dOut);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpy(out, dOut, (size_t)n * 24 * sizeof(float), hipMemcpyDeviceToHost));
  (void)hipFree(dCases);
  (void)hipFree(dOut);
  return BLCD_OK;
}

int blcd_debug_mass_data(const blcd_scene_desc* scene, int32_t shape, float density, float* out) {
  if (!scene || !out || shape < 0 || shape >= scene->n_shapes) return fail(BLCD_ERR_INVALID, "blcd_debug_mass_data: bad arguments");
  Shape s = build_shape(scene->shapes[shape]);
  MassData md;
  ShapeComputeMass(&s, &md, density);
  out[0] = md.mass;
  out[1] = md.center.x;
  out[2] = md.center.y;
  out[3] = md.I;
  out[4] = (float)s.count;
  for (int i = 0; i < s.count && i < 8; ++i) {
    out[5 + 2 * i] = s.v[i].x;
    out[6 + 2 * i] = s.v[i].y;
  }
  return BLCD_OK;
}

}  // extern "C"
