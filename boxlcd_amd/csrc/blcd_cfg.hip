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

#ifndef BLCD_REST_REUSE
#define BLCD_REST_REUSE 0   // 1: environments at rest skip their world steps and re-emit the unchanged frame / observation (step_kernel; set per class by the build)
#endif

namespace blcd {

// observation row of step t := row of step t - 1 (an environment at rest: same poses, same float64 glue, same values)
__device__ __forceinline__ void copyObsRow(float* __restrict__ obsOut, int t, int N, int e, int nobs) {
  if (!obsOut) return;
  const float* src = obsOut + ((size_t)(t - 1) * N + e) * nobs;
  float* dst = obsOut + ((size_t)t * N + e) * nobs;
  for (int i = 0; i < nobs; ++i) dst[i] = src[i];
}

// Per-wave timers / event counters (blcd_debug_wave_times) are a BUILD-time feature (BLCD_DEFS=-DBLCD_WAVETIMES): as a run-time
// switch they kept 16 counter registers and their updates alive in every product kernel.
#ifdef BLCD_WAVETIMES
constexpr bool kWaveTimes = true;
#else
constexpr bool kWaveTimes = false;
#endif

template <int NB, int NJ, int NP, int SH, bool SCHED>
__global__ __launch_bounds__(kBlock, BLCD_WAVES_PER_EU) void step_kernel(const DevScene* __restrict__ S, float* __restrict__ st, int N, int nSlots,
                                                      const int* __restrict__ eid, const float* __restrict__ actions,
                                                      int nEnvSteps, int nWorldSteps, int setMotors, int lanes,
                                                      unsigned long long* __restrict__ waveTimes, long long actStride,
                                                      uint8_t* __restrict__ lcdOut, float* __restrict__ obsOut, int* __restrict__ faultAny,
                                                      int pass, int yieldMaxLanes, unsigned long long* __restrict__ schedStats, int lcdBits, int stepBudget, int resumeBatch, const int* __restrict__ heavyEnd, int nSimds) {
  uint32_t* const ldsRows = Env<NB, NJ, NP, SH, SCHED>::ldsFrameRows();   // LCD row masks of the wave's 64 environments (stride 17: conflict-free); shares LDS with the staged island's contact block
  unsigned long long t0 = (kWaveTimes && waveTimes) ? __builtin_amdgcn_s_memrealtime() : 0ull;  // diagnostic builds only (BLCD_WAVETIMES)
  // `lanes` (<= 64) environments per wave: the path is bound by per-wave serial latency and lane divergence, not by
  // VALU throughput, so partially filled waves (more, shorter waves) can finish a launch sooner.
  int slot;
  bool valid = true;   // false: a shadow lane (computes what the wave's last environment computes, never stores)
  // this wave holds the slots [first, first + width) below `end`
  int first, width, end = nSlots;
  if (heavyEnd) {
    // Two wave widths in one launch (re-binned batches).  The slot sort puts the environments that are not asleep first
    // (*heavyEnd of them, heaviest first).  While they oversubscribe the SIMDs, full waves are best (total wave time counts); once
    // most of the batch sleeps, the launch lasts as long as ONE awake wave's dependent chain while most SIMDs idle - then the
    // awake environments are spread over all SIMDs in narrower waves, whose union of code paths (who hits a wall this step,
    // who needs the TOI sub-step) is smaller.  Placement only.  The sleeping tail keeps full waves.
    int he = *heavyEnd;
    he = he < 0 ? 0 : (he > nSlots ? nSlots : he);
    const int wmin = nSimds >> 16, ns = nSimds & 0xffff;   // narrowest width in the high half
    const int wh = he >= 64 * ns ? 64 : (he <= wmin * ns ? wmin : (he + ns - 1) / ns);
    const int heavyBlocks = (he + wh - 1) / wh;
    if ((int)blockIdx.x < heavyBlocks) {
      first = blockIdx.x * wh;
      width = wh;
      end = he;
    } else {
      first = he + ((int)blockIdx.x - heavyBlocks) * 64;
      width = 64;
    }
  } else {
    first = blockIdx.x * lanes;   // state is stored in slot order; eid[slot] is the environment it holds
    width = lanes;
  }
  if (first >= end) return;
  slot = first + (int)threadIdx.x;
  {
    const int last = (first + width < end ? first + width : end) - 1;
    if ((int)threadIdx.x >= width || slot > last) {
      // Lanes without an environment of their own: the ragged last wave of a launch (100 000 environments = 1 562 waves + 32
      // lanes) and every lane beyond the width of a narrow wave.  They used to exit here, which sent the wave down the per-lane
      // emission path (a frame needs all 64 lanes to write it, see below): full raster + float64 observation glue on EVERY step,
      // no reuse for environments at rest - one such wave was the whole 1.7 ms of a sleeping Dropbox chunk (0.47 ms without it;
      // profiles/r04_dropbox100k_rest_timeline.txt), and the narrow waves of the two-width launches paid it on every step.
      // Instead they shadow the wave's last environment: same inputs, same arithmetic (no extra code path for the wave), no
      // store (`valid`), so every wave is a full one.  The scheduler's kernel keeps the early exit, and so do lanes for which
      // the class has no LDS column (the host never asks for more lanes than the class's LDS blocks hold).
      if (SCHED || (int)threadIdx.x >= Env<NB, NJ, NP, SH, SCHED>::kMaxLanes) return;
      slot = last;
      valid = false;
    }
  }
  const int e = eid[slot];
  using EnvT = Env<NB, NJ, NP, SH, SCHED>;
  if constexpr (!SCHED) {
    // ---- the plain kernel: every lane runs its environment through the whole launch in lock step ----
    EnvT env;
    env.load(S, st, N, slot);
    env.profOn = kWaveTimes && waveTimes != nullptr;
#ifdef BLCD_ABLATION
    if (S->dbgSkip & 8) nEnvSteps = nWorldSteps = 0;
#endif
    if (nEnvSteps > 0) {
      // Fused rollout: this wave advances its environments through all nEnvSteps on its own (no grid-wide barrier between
      // env steps: environments are independent), writing the per-step LCD frame / observation rows as it goes.
      const size_t lcdRow = (size_t)S->lcdH * S->lcdW / (lcdBits ? 8 : 1);   // bytes per frame (lcdBits: one bit per pixel)
      // Environments at rest.  When every body of a joint-free environment is asleep, b2World::Step changes nothing: contacts of
      // sleeping bodies are skipped, no island forms, no proxy moves, no TOI candidate exists (Env::atRest spells out the state
      // words involved) - so the three world steps are skipped, and what _get_obs / lcd_render would compute from the unchanged
      // poses is what they computed last step: the frame rows are still in LDS (or, bit-packed, in the previous output row) and
      // the observation row is copied from the previous step's.  Every frame and observation is still written, bit for bit the
      // same.  It pays where whole launches are at rest (Dropbox: the batch sleeps from step ~50 on) and is compiled in per class
      // (-DBLCD_REST_REUSE=1, __graft_entry__.CLASS_FLAGS): elsewhere the launch is as long as its awake waves and the extra
      // registers cost more than the idle SIMDs gain.
      constexpr bool kRest = NJ == 0 && BLCD_REST_REUSE;
      bool emitted = false;   // this lane has emitted a frame + observation in this launch
      for (int t = 0; t < nEnvSteps; ++t) {
        env.setMotorSpeeds(actions ? actions + (size_t)t * actStride : nullptr, N, e);
        const bool rest = kRest && env.atRest();
        if (!rest)
          for (int k = 0; k < S->substeps; ++k) env.worldStep();
        const bool reuse = rest && emitted && (!obsOut || t > 0);
        if (lcdOut || obsOut) {
          auto body = [&](int i, Vec2* p, float* a, int* sel) {
            const int bi = NB == 1 ? 0 : i;  // static index for single-body scenes (keeps env in registers)
            *p = env.xfp[bi];
            *a = env.a[bi];
            *sel = env.sel[bi];
          };
          bool ok;
          float* const obsRowV = (obsOut && valid) ? obsOut + ((size_t)t * N + e) * S->nobs : nullptr;   // this lane's observation row (shadow lanes: none)
          if (lcdOut && lcdBits && S->lcdW == 16) {
            // 16x16 frames at one bit per pixel: 32 B per environment, straight from the row masks (complemented: 1 = background)
            uint4* o = reinterpret_cast<uint4*>(lcdOut + ((size_t)t * N + e) * 32);
            ok = true;
            if (reuse) {
              if (valid) {
                const uint4* prev = reinterpret_cast<const uint4*>(lcdOut + ((size_t)(t - 1) * N + e) * 32);
                o[0] = prev[0];
                o[1] = prev[1];
                copyObsRow(obsOut, t, N, e, S->nobs);
              }
            } else {
              uint32_t rows[16];
              ok = emit_env<16, uint32_t, float, SH == 1>(S, body, obsRowV, nullptr, rows);
              uint32_t w[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) w[j] = (~rows[2 * j] & 0xffffu) | (~rows[2 * j + 1] << 16);
              if (valid) {
                o[0] = make_uint4(w[0], w[1], w[2], w[3]);
                o[1] = make_uint4(w[4], w[5], w[6], w[7]);
              }
            }
            emitted = true;
          } else if (lcdOut && S->lcdW == 16 && __ballot(1) == ~0ull) {   // full waves only (shadow lanes included): a frame needs all 64 lanes to write it
            // 16x16 frames: 256 B = one dword per lane.  Writing each lane's own frame row by row makes every store touch 64
            // different cache lines; instead the lanes park their 16 row masks in LDS and the wave writes one whole frame per
            // store instruction (fully coalesced), frame k being the environment held by lane k.
            // Round 4: the masks are parked [row][lane] (row stride 68) and every store instruction writes FOUR whole frames: lane l
            // owns row l & 15 of frame 4 g + (l >> 4), expands its 16 mask bits to 16 bytes and stores them as one dwordx4 - 16
            // LDS reads (conflict-free) and 16 stores of 1 KB per env-step instead of 64 + 64 of 256 B (LCD emission was 1.3 ms
            // of a 9.3 ms Bounce-100k rollout, tools/emit_cost.py).
            const int lane = (int)threadIdx.x;
            constexpr int RS = EnvT::kFrameRowStride;
            uint32_t* const ldsE = ldsRows + 16 * RS;          // the wave's environment ids (rewritten every step: the block is shared with the solver in some classes)
            ok = true;
            if (reuse) {   // this lane's rows of the previous step are still in LDS (nothing else lives in that block for joint-free classes)
              if (valid) copyObsRow(obsOut, t, N, e, S->nobs);
            } else {
              uint32_t rows[16];
              ok = emit_env<16, uint32_t, float, SH == 1>(S, body, obsRowV, nullptr, rows);
#pragma unroll
              for (int y = 0; y < 16; ++y) ldsRows[y * RS + lane] = rows[y];
            }
            ldsE[lane] = (uint32_t)e;
            emitted = true;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            uint8_t* frames = lcdOut + (size_t)t * N * lcdRow;
            const int row = lane & 15, sub = lane >> 4;        // this lane's row of frame 4 g + sub
            const int nFrames = __popcll(__ballot(valid));      // shadow lanes are the wave's last lanes: their frames are not written
#pragma unroll 4
            for (int g = 0; g < 16; ++g) {
              const int k = 4 * g + sub;
              const uint32_t m = ldsRows[row * RS + k];
              const uint32_t ek = ldsE[k];
              uint4 px;
              px.x = ((((m >> 0) & 0xfu) * 0x00204081u) & 0x01010101u) ^ 0x01010101u;
              px.y = ((((m >> 4) & 0xfu) * 0x00204081u) & 0x01010101u) ^ 0x01010101u;
              px.z = ((((m >> 8) & 0xfu) * 0x00204081u) & 0x01010101u) ^ 0x01010101u;
              px.w = ((((m >> 12) & 0xfu) * 0x00204081u) & 0x01010101u) ^ 0x01010101u;
              if (k < nFrames) *reinterpret_cast<uint4*>(frames + (size_t)ek * 256 + 16 * row) = px;
            }
            __builtin_amdgcn_wave_barrier();
          } else if (valid) {
            float* obsRowOut = obsOut ? obsOut + ((size_t)t * N + e) * S->nobs : nullptr;
            uint8_t* lcdRowOut = lcdOut ? lcdOut + ((size_t)t * N + e) * lcdRow : nullptr;
            bool tall = false;
            if constexpr (NB > 7) tall = S->lcdH == 32;   // 32-row LCDs (Crab, CrabCube, SpiderCube) only occur in the largest class
            if (tall) ok = emit_env<32, uint64_t, float, false>(S, body, obsRowOut, lcdRowOut, nullptr, lcdBits != 0);
            else ok = emit_env<16, uint32_t, float, SH == 1>(S, body, obsRowOut, lcdRowOut, nullptr, lcdBits != 0);
          } else {
            ok = true;
          }
          if (!ok) env.fault |= FAULT_ELLIPSE;
        }
      }
    } else {
      if (setMotors) env.setMotorSpeeds(actions, N, e);
      for (int k = 0; k < nWorldSteps; ++k) env.worldStep();
    }
    env.checkFault();
    if (!valid) return;   // shadow lane: the environment's own lane stores
    if (env.fault && faultAny) *faultAny = 1;
    env.store(st, N, slot);
    if (kWaveTimes && waveTimes) {
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
    return;
  } else {
  // ---- the scheduler's kernel ----
  // Every environment carries its own progress (env-step t, sub-step) and, when it is suspended, what it is suspended at
  // (state words schedWordOffset.. ).  A launch advances every unfinished environment by at most `stepBudget`
  // world steps from wherever it stands: a suspended one first pays what it owes (worldStepResume).  A lane whose environment
  // suspends idles until the next launch - by then the host has sorted the suspended environments together, so that what they
  // owe (the rest of 180 sweeps / 60 position iterations, a TOI event with its sub-step) runs in dense waves instead of keeping
  // 63 finished lanes waiting.  pass 0 of a chunk (or the host's memset in asynchronous rollouts) starts everyone at step 0.
  float* const progWord = st + (size_t)schedWordOffset(S->nb, S->nj, S->np) * N + slot;
  const bool fresh = nEnvSteps <= 0 || pass == 0;
  const uint32_t prog = fresh ? 0u : __float_as_uint(progWord[0]);
  const uint32_t prog1 = fresh ? 0u : __float_as_uint(progWord[N]);
  int t = (int)(prog & 0xffffu), sub = (int)((prog >> 16) & 3u);
  bool live = t < nEnvSteps;
  if (nEnvSteps > 0 && !__any(live)) return;     // the whole wave has finished the rollout / chunk
  const bool fullWave = __ballot(1) == ~0ull;    // every lane of the wave holds an environment (the coalesced frame store needs all 64)
  EnvT env;
  env.load(S, st, N, slot);
  env.profOn = kWaveTimes && waveTimes != nullptr;
  env.toiPending = (prog >> 18) & 1u;
  env.velMask = prog1 & 0x7fu;
  env.posMask = (prog1 >> 7) & 0x7fu;
  env.islandedMask = (prog1 >> 14) & 0x7fu;
  env.yieldMaxLanes = yieldMaxLanes;
#ifdef BLCD_ABLATION
  if (S->dbgSkip & 8) nEnvSteps = nWorldSteps = 0;
#endif
  if (nEnvSteps > 0) {
    const size_t lcdRow = (size_t)S->lcdH * S->lcdW / (lcdBits ? 8 : 1);   // bytes per frame (lcdBits: one bit per pixel)
    const bool mayYield = EnvT::kCanYield && yieldMaxLanes > 0;
    bool pending = env.toiPending || (env.velMask | env.posMask) != 0;
    const bool enteredLive = live;
    int budget = stepBudget > 0 ? stepBudget : 0x7fffffff;
    while (__any(live)) {
      bool emitNow = false, stop = false;
      // In-wave batching (resumeBatch > 0): a suspended lane does not wait for the next launch but for company - the wave runs
      // the resumption code once for everyone who is owed the same kind of work (the rest of a solve / a TOI event) as soon as
      // resumeBatch lanes are, or when nobody has anything else to do; until then the others carry on with their world steps.
      // Lanes of one wave may therefore stand at different env-steps: phases are code, not time.
      bool runResume = pending, runStep = !pending;
      if (resumeBatch > 0) {
        const bool wantSolve = live && pending && !env.toiPending, wantToi = live && pending && env.toiPending;
        const int nSolve = __popcll(__ballot(wantSolve)), nToi = __popcll(__ballot(wantToi));
        const bool anyStep = __any(live && !pending);
        const bool pickSolve = nSolve > 0 && nSolve >= nToi && (nSolve >= resumeBatch || !anyStep);
        const bool pickToi = !pickSolve && nToi > 0 && (nToi >= resumeBatch || !anyStep);
        runResume = (pickSolve && wantSolve) || (pickToi && wantToi);
        runStep = !(pickSolve || pickToi) && !pending;
      }
      if (live && (runResume || runStep)) {
        bool suspended;
        if (runResume) {
          suspended = env.worldStepResume(mayYield);
        } else {
          if (sub == 0) env.setMotorSpeeds(actions ? actions + (size_t)t * actStride : nullptr, N, e);
          suspended = env.worldStep(mayYield);
        }
        pending = suspended;
        stop = --budget <= 0;
        if (suspended) {
          if (resumeBatch <= 0) live = false;     // waits for the next launch
        } else if (++sub == S->substeps) {
          sub = 0;
          emitNow = true;
        }
      }
      if ((lcdOut || obsOut) && __any(emitNow)) {
        auto body = [&](int i, Vec2* p, float* a, int* sel) {
          const int bi = NB == 1 ? 0 : i;  // static index for single-body scenes (keeps env in registers)
          *p = env.xfp[bi];
          *a = env.a[bi];
          *sel = env.sel[bi];
        };
        bool ok = true;
        if (lcdOut && lcdBits && S->lcdW == 16) {
          // 16x16 frames at one bit per pixel: 32 B per environment, straight from the row masks (complemented: 1 = background)
          if (emitNow) {
            uint32_t rows[16];
            ok = emit_env<16, uint32_t, float, SH == 1>(S, body, obsOut ? obsOut + ((size_t)t * N + e) * S->nobs : nullptr, nullptr, rows);
            uint32_t w[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = (~rows[2 * j] & 0xffffu) | (~rows[2 * j + 1] << 16);
            uint4* o = reinterpret_cast<uint4*>(lcdOut + ((size_t)t * N + e) * 32);
            o[0] = make_uint4(w[0], w[1], w[2], w[3]);
            o[1] = make_uint4(w[4], w[5], w[6], w[7]);
          }
        } else if (lcdOut && S->lcdW == 16 && fullWave) {
          // 16x16 frames: 256 B = one dword per lane.  The emitting lanes park their 16 row masks in LDS and the WHOLE wave writes
          // one frame per store instruction (fully coalesced): frame k = the environment of lane k, at that lane's own env-step.
          const int lane = (int)threadIdx.x;
          if (emitNow) {
            uint32_t rows[16];
            ok = emit_env<16, uint32_t, float, SH == 1>(S, body, obsOut ? obsOut + ((size_t)t * N + e) * S->nobs : nullptr, nullptr, rows);
#pragma unroll
            for (int y = 0; y < 16; ++y) ldsRows[lane * 17 + y] = rows[y];
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          const int row = lane >> 2, x0 = (lane & 3) * 4;   // this lane's 4 pixels of any frame
          unsigned long long em = __ballot(emitNow);
          while (em) {
            const int k = __ffsll((long long)em) - 1;
            em &= em - 1;
            const int ek = __builtin_amdgcn_readlane(e, k);
            const int tk = __builtin_amdgcn_readlane(t, k);
            const uint32_t m = ldsRows[k * 17 + row];
            const uint32_t px = ((((m >> x0) & 0xfu) * 0x00204081u) & 0x01010101u) ^ 0x01010101u;
            *reinterpret_cast<uint32_t*>(lcdOut + ((size_t)tk * N + ek) * 256 + 4 * lane) = px;
          }
          __builtin_amdgcn_wave_barrier();
        } else if (emitNow) {
          float* obsRowOut = obsOut ? obsOut + ((size_t)t * N + e) * S->nobs : nullptr;
          uint8_t* lcdRowOut = lcdOut ? lcdOut + ((size_t)t * N + e) * lcdRow : nullptr;
          bool tall = false;
          if constexpr (NB > 7) tall = S->lcdH == 32;   // 32-row LCDs (Crab, CrabCube, SpiderCube) only occur in the largest class
          if (tall) ok = emit_env<32, uint64_t, float, false>(S, body, obsRowOut, lcdRowOut, nullptr, lcdBits != 0);
          else ok = emit_env<16, uint32_t, float, SH == 1>(S, body, obsRowOut, lcdRowOut, nullptr, lcdBits != 0);
        }
        if (!ok) env.fault |= FAULT_ELLIPSE;
      }
      if (emitNow && ++t >= nEnvSteps) live = false;
      if (stop) live = false;
    }
    progWord[0] = __uint_as_float((uint32_t)t | ((uint32_t)sub << 16) | ((uint32_t)env.toiPending << 18));
    progWord[N] = __uint_as_float(env.velMask | (env.posMask << 7) | (env.islandedMask << 14));
    if (schedStats) {
      // [0..2] / [4..6]: first / later passes: lanes that entered live, lanes that left suspended, waves; [3]: environments that
      // have not reached the last env-step yet (the host zeroes it before a launch of an asynchronous rollout)
      const bool susp = env.toiPending || (env.velMask | env.posMask) != 0;
      const unsigned long long in_ = __ballot(1), sus_ = __ballot(susp), lv_ = __ballot(enteredLive), un_ = __ballot(t < nEnvSteps);
      if ((int)threadIdx.x == __ffsll((long long)in_) - 1) {
        unsigned long long* o = schedStats + (pass > 0 ? 4 : 0);
        atomicAdd(o, (unsigned long long)__popcll(lv_));
        atomicAdd(o + 1, (unsigned long long)__popcll(sus_));
        atomicAdd(o + 2, 1ull);
        if (un_) atomicAdd(schedStats + 3, (unsigned long long)__popcll(un_));
      }
    }
  } else {
    if (setMotors) env.setMotorSpeeds(actions, N, e);
    for (int k = 0; k < nWorldSteps; ++k) env.worldStep();
  }
  env.checkFault();
  if (env.fault && faultAny) *faultAny = 1;
  env.store(st, N, slot);
  if (kWaveTimes && waveTimes) {
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
  }   // SCHED
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
  // the scheduler's kernel exists for the classes it can act on (see Env::kCanYield), runs only when asked for, and is compiled
  // only into BLCD_DEFS=-DBLCD_SCHED builds (blcd_create refuses the scheduler knobs otherwise): the three schedulers were
  // measured 25-60 % slower than the plain rollout and doubled the device code of every class
#ifdef BLCD_SCHED
  if constexpr (BLCD_NB <= 7) {
    if (A.sched) {
      hipLaunchKernelGGL((step_kernel<BLCD_NB, BLCD_NJ, BLCD_NP, BLCD_SH, true>), grid, dim3(kBlock), 0, stream, A.S, A.st, A.N, A.nSlots, A.eid, A.actions,
                         A.nEnvSteps, A.nWorldSteps, A.setMotors, A.lanes, A.waveTimes, A.actStride, A.lcdOut, A.obsOut, A.faultAny, A.pass, A.yieldMaxLanes, A.schedStats, A.lcdBits, A.stepBudget, A.resumeBatch, A.heavyEnd, A.nSimds);
      return;
    }
  }
#endif
  hipLaunchKernelGGL((step_kernel<BLCD_NB, BLCD_NJ, BLCD_NP, BLCD_SH, false>), grid, dim3(kBlock), 0, stream, A.S, A.st, A.N, A.nSlots, A.eid, A.actions,
                     A.nEnvSteps, A.nWorldSteps, A.setMotors, A.lanes, A.waveTimes, A.actStride, A.lcdOut, A.obsOut, A.faultAny, A.pass, A.yieldMaxLanes, A.schedStats, A.lcdBits, A.stepBudget, A.resumeBatch, A.heavyEnd, A.nSimds);
}
void BLCD_NAME(launch_set_poses_, BLCD_NB, BLCD_NJ, BLCD_NP, BLCD_SH)(dim3 grid, hipStream_t stream, const SetPosesArgs& A) {
  hipLaunchKernelGGL((set_poses_kernel<BLCD_NB, BLCD_NJ, BLCD_NP, BLCD_SH>), grid, dim3(64), 0, stream, A.S, A.st, A.N, A.slotOf, A.idxs, A.n,
                     A.poses, A.mask);
}

}  // namespace blcd
