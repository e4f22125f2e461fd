// blcd_cfg_launch.h — host-side launchers of the kernels that are specialised per scene-size class
// (max bodies, max joints, max pair slots).  Each class is compiled in its own translation unit (blcd_cfg.hip with
// -DBLCD_NB/-DBLCD_NJ/-DBLCD_NP) so that the classes build in parallel; blcd_api.hip dispatches on the handle's class.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "blcd_world.h"

// X(max bodies, max joints, max pair slots, shape set): shape set 1 = circles-only scenes (see Env's SH parameter).
// Order matters: the first class that fits the scene is used.
#define BLCD_CONFIGS(X) \
  X(1, 0, 4, 1) X(2, 0, 9, 1) X(1, 0, 4, 0) X(2, 0, 9, 0) X(3, 0, 15, 0) X(4, 3, 16, 0) X(5, 3, 24, 0) X(7, 3, 44, 0) X(20, 20, 100, 0)

namespace blcd {

struct StepArgs {
  const DevScene* S;
  float* st;      // state base of the launch's slot range (= full base + first slot): field f of slot k at st[f * N + k]
  int N;          // environments of the handle = stride between state fields
  int nSlots;     // slots this launch covers (a cohort of a large batch, or all N)
  const int* eid; // eid[k] = environment held by slot k of the range
  const float* actions;
  int nEnvSteps, nWorldSteps, setMotors, lanes;
  unsigned long long* waveTimes;
  long long actStride;
  uint8_t* lcdOut;
  float* obsOut;
  int* faultAny;   // set to 1 by any lane whose environment carries a fault flag after the launch
  int pass;           // fused chunks are stepped in passes: 0 starts the chunk, > 0 resumes it (progress word of every slot)
  int yieldMaxLanes;  // > 0: a lane may suspend its environment when at most this many lanes of its wave still sweep (never in the last pass)
  unsigned long long* schedStats;   // 8 counters (see blcd_sched_stats) or null
  int lcdBits;        // lcdOut holds frames at one bit per pixel (blcd_rollout_bits)
  int sched;          // launch the scheduler's kernel (step_kernel<..., true>): passes / suspension / resumption
  int stepBudget;     // scheduler kernel: world steps an environment may advance in this launch (0 = to the end)
  const int* heavyEnd;  // device: number of leading slots that are not asleep after the last slot sort (null = one wave width, `lanes`)
  int nSimds;           // SIMDs of the device (wave-width choice for the awake slots)
  int resumeBatch;    // scheduler kernel: > 0 = suspended lanes resume inside the launch, as soon as this many of the wave wait for the same kind of work
};
struct SetPosesArgs {
  const DevScene* S;
  float* st;
  int N;
  const int* slotOf;
  const int* idxs;
  int n;
  const float* poses;
  const uint8_t* mask;
};

#define X(a, b, c, d)                                                                       \
  void launch_step_##a##_##b##_##c##_##d(dim3 grid, hipStream_t stream, const StepArgs& A);    \
  void launch_set_poses_##a##_##b##_##c##_##d(dim3 grid, hipStream_t stream, const SetPosesArgs& A);
BLCD_CONFIGS(X)
#undef X

}  // namespace blcd
