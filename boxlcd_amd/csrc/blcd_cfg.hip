// blcd_cfg.hip — the two kernels that hold a whole environment in registers, for ONE scene-size class
// (compile with -DBLCD_NB=<max bodies> -DBLCD_NJ=<max joints> -DBLCD_NP=<max pair slots> -DBLCD_SH=<shape set>; see
// blcd_cfg_launch.h).
#include <hip/hip_runtime.h>
#if !defined(BLCD_SH) || BLCD_SH == 0
#define BLCD_SINCOS_INLINE 1   // general classes call sincosf in their hot loops (position solver, TOI): keep it inlined there
#endif
#include "blcd_emit.h"
#include "blcd_cfg_launch.h"

#if !defined(BLCD_NB) || !defined(BLCD_NJ) || !defined(BLCD_NP) || !defined(BLCD_SH)
#error "compile with -DBLCD_NB=.. -DBLCD_NJ=.. -DBLCD_NP=.. -DBLCD_SH=.."
#endif

#ifndef BLCD_WAVES_PER_EU
#define BLCD_WAVES_PER_EU 1   // waves per SIMD the register allocator must leave room for (build-time tuning knob)
#endif

namespace blcd {

template <int NB, int NJ, int NP, int SH>
__global__ __launch_bounds__(kBlock, BLCD_WAVES_PER_EU) void step_kernel(const DevScene* __restrict__ S, float* __restrict__ st, int N, int nSlots,
                                                      const int* __restrict__ eid, const float* __restrict__ actions,
                                                      int nEnvSteps, int nWorldSteps, int setMotors, int lanes,
                                                      unsigned long long* __restrict__ waveTimes, long long actStride,
                                                      uint8_t* __restrict__ lcdOut, float* __restrict__ obsOut, int* __restrict__ faultAny) {
  uint32_t* const ldsRows = Env<NB, NJ, NP, SH>::ldsFrameRows();   // LCD row masks of the wave's 64 environments (stride 17: conflict-free); shares LDS with the staged island's contact block
  unsigned long long t0 = waveTimes ? __builtin_amdgcn_s_memrealtime() : 0ull;  // diagnostic only (BLCD_WAVETIMES)
  // `lanes` (<= 64) environments per wave: the path is bound by per-wave serial latency and lane divergence, not by
  // VALU throughput, so partially filled waves (more, shorter waves) can finish a launch sooner.
  if ((int)threadIdx.x >= lanes) return;
  int slot = blockIdx.x * lanes + threadIdx.x;   // state is stored in slot order; eid[slot] is the environment it holds
  if (slot >= nSlots) return;
  const int e = eid[slot];
  Env<NB, NJ, NP, SH> env;
  env.load(S, st, N, slot);
  env.profOn = waveTimes != nullptr;
#ifdef BLCD_ABLATION
  if (S->dbgSkip & 8) nEnvSteps = nWorldSteps = 0;
#endif
  if (nEnvSteps > 0) {
    // Fused rollout: this wave advances its environments through all nEnvSteps on its own (no grid-wide barrier between
    // env steps: environments are independent), writing the per-step LCD frame / observation rows as it goes.
    const size_t lcdRow = (size_t)S->lcdH * S->lcdW;
    for (int t = 0; t < nEnvSteps; ++t) {
      env.setMotorSpeeds(actions ? actions + (size_t)t * actStride : nullptr, N, e);
      for (int k = 0; k < S->substeps; ++k) env.worldStep();
      if (lcdOut || obsOut) {
        auto body = [&](int i, Vec2* p, float* a, int* sel) {
          const int bi = NB == 1 ? 0 : i;  // static index for single-body scenes (keeps env in registers)
          *p = env.xfp[bi];
          *a = env.a[bi];
          *sel = env.sel[bi];
        };
        bool ok;
        if (lcdOut && S->lcdW == 16 && __ballot(1) == ~0ull) {   // full waves only: a frame needs all 64 lanes to write it
          // 16x16 frames: 256 B = one dword per lane.  Writing each lane's own frame row by row makes every store touch 64
          // different cache lines; instead the lanes park their 16 row masks in LDS and the wave writes one whole frame per
          // store instruction (fully coalesced), frame k being the environment held by lane k.
          uint32_t rows[16];
          ok = emit_env<16, uint32_t, float, SH == 1>(S, body, obsOut ? obsOut + ((size_t)t * N + e) * S->nobs : nullptr, nullptr, rows);
          const int lane = (int)threadIdx.x;
#pragma unroll
          for (int y = 0; y < 16; ++y) ldsRows[lane * 17 + y] = rows[y];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          uint8_t* frames = lcdOut + (size_t)t * N * lcdRow;
          const int row = lane >> 2, x0 = (lane & 3) * 4;   // this lane's 4 pixels of any frame
          for (int k = 0; k < 64; ++k) {
            const int ek = __builtin_amdgcn_readlane(e, k);
            const uint32_t m = ldsRows[k * 17 + row];
            const uint32_t px = ((((m >> x0) & 0xfu) * 0x00204081u) & 0x01010101u) ^ 0x01010101u;
            *reinterpret_cast<uint32_t*>(frames + (size_t)ek * 256 + 4 * lane) = px;
          }
          __builtin_amdgcn_wave_barrier();
        } else {
          float* obsRowOut = obsOut ? obsOut + ((size_t)t * N + e) * S->nobs : nullptr;
          uint8_t* lcdRowOut = lcdOut ? lcdOut + ((size_t)t * N + e) * lcdRow : nullptr;
          bool tall = false;
          if constexpr (NB > 7) tall = S->lcdH == 32;   // 32-row LCDs (Crab, CrabCube, SpiderCube) only occur in the largest class
          if (tall) ok = emit_env<32, uint64_t, float, false>(S, body, obsRowOut, lcdRowOut);
          else ok = emit_env<16, uint32_t, float, SH == 1>(S, body, obsRowOut, lcdRowOut);
        }
        if (!ok) env.fault |= FAULT_ELLIPSE;
      }
    }
  } else {
    if (setMotors) env.setMotorSpeeds(actions, N, e);
    for (int k = 0; k < nWorldSteps; ++k) env.worldStep();
  }
  env.checkFault();
  if (env.fault && faultAny) *faultAny = 1;
  env.store(st, N, slot);
  if (waveTimes) {
    // wave total in 100 MHz ticks + per-phase shader cycles / event counts (lane maxima via cross-lane max)
    unsigned long long* o = waveTimes + (size_t)blockIdx.x * 9;
    if (threadIdx.x == 0) o[0] = __builtin_amdgcn_s_memrealtime() - t0;
#ifdef BLCD_PROF_TOI2
    for (int k = 0; k < 8; ++k) atomicAdd(&o[1 + k], env.prof[k]);   // wave totals (one lane records each interval)
#else
    for (int k = 0; k < 6; ++k) atomicMax(&o[1 + k], env.prof[k]);
    for (int k = 6; k < 8; ++k) atomicAdd(&o[1 + k], env.prof[k]);
#endif
  }
}

// b2Body::SetTransform per masked body: position first, then angle (two calls, like `body.position=`; `body.angle=`),
// each followed by proxy synchronisation with zero displacement and - Box2D 2.3.0 - m_contactManager.FindNewContacts(), which
// also consumes the moves buffered when the bodies were created (reset), so contacts appear in the reference's order.
template <int NB, int NJ, int NP, int SH>
__global__ void set_poses_kernel(const DevScene* __restrict__ S, float* __restrict__ st, int N, const int* __restrict__ slotOf,
                                 const int* __restrict__ idxs, int n, const float* __restrict__ poses,
                                 const uint8_t* __restrict__ mask) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int e = idxs ? idxs[k] : k;
  if (e < 0 || e >= N) return;
  e = slotOf[e];
  Env<NB, NJ, NP, SH> env;
  env.load(S, st, N, e);
  bool createdMoves = (env.wflags & WF_NEWFIXTURE) != 0;   // proxies created by reset are still in the move buffer
  for (int i = 0; i < S->nb; ++i) {
    if (mask && !mask[i]) continue;
    const float* p = poses + ((size_t)k * S->nb + i) * 3;
    for (int pass = 0; pass < 2; ++pass) {
      float angle = pass == 0 ? env.a[i] : p[2];
      Vec2 pos = pass == 0 ? V2(p[0], p[1]) : env.xfp[i];
      env.q[i] = env.rotFor(4 + i, angle);
      env.xfp[i] = pos;
      Transform xf = env.xfOf(4 + i);
      env.c[i] = Mul(xf, env.lc[i]);
      env.a[i] = angle;
      env.c0[i] = env.c[i];
      env.a0[i] = angle;
      env.synchronizeProxy(i, xf, xf);
      env.findNewContacts(createdMoves);
      createdMoves = false;
    }
  }
  env.store(st, N, e);
}

#define BLCD_PASTE5(p, a, b, c, d) p##a##_##b##_##c##_##d
#define BLCD_NAME(p, a, b, c, d) BLCD_PASTE5(p, a, b, c, d)

void BLCD_NAME(launch_step_, BLCD_NB, BLCD_NJ, BLCD_NP, BLCD_SH)(dim3 grid, hipStream_t stream, const StepArgs& A) {
  hipLaunchKernelGGL((step_kernel<BLCD_NB, BLCD_NJ, BLCD_NP, BLCD_SH>), grid, dim3(kBlock), 0, stream, A.S, A.st, A.N, A.nSlots, A.eid, A.actions,
                     A.nEnvSteps, A.nWorldSteps, A.setMotors, A.lanes, A.waveTimes, A.actStride, A.lcdOut, A.obsOut, A.faultAny);
}
void BLCD_NAME(launch_set_poses_, BLCD_NB, BLCD_NJ, BLCD_NP, BLCD_SH)(dim3 grid, hipStream_t stream, const SetPosesArgs& A) {
  hipLaunchKernelGGL((set_poses_kernel<BLCD_NB, BLCD_NJ, BLCD_NP, BLCD_SH>), grid, dim3(64), 0, stream, A.S, A.st, A.N, A.slotOf, A.idxs, A.n,
                     A.poses, A.mask);
}

}  // namespace blcd
