#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../boxlcd_amd/csrc/blcd_math.h"
using namespace blcd;
__global__ void k(int reps, float base, unsigned long long* cyc, float* out) {
  float x = base + 1e-3f * threadIdx.x, acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    float s, c;
    blcd_sincosf(x, &s, &c);
    acc += s * c;
    x += 1e-4f * (s + 1.5f);   // dependent chain: next argument depends on this result (like the root finder)
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = acc;
}
__global__ void kt(int reps, unsigned long long* cyc, float* out) {   // Sweep::GetTransform chain
  Sweep sw; sw.localCenter = V2(0.01f, 0.02f); sw.c0 = V2(1, 2); sw.c = V2(1.1f, 1.7f); sw.a0 = 0.3f + 1e-3f * threadIdx.x; sw.a = 0.5f; sw.alpha0 = 0;
  float t = 0.3f, acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    Transform xf; sw.GetTransform(&xf, t);
    acc += xf.p.x + xf.q.s;
    t = 0.5f * (t + 0.5f + 1e-3f * xf.q.c);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  out[threadIdx.x] = acc;
}
int main() {
  unsigned long long* dc; hipMalloc(&dc, 8); float* dout; hipMalloc(&dout, 256);
  for (float base : {0.01f, 0.5f, 1.3f, 40.0f, 200.0f}) {
    k<<<1, 64>>>(1000, base, dc, dout); k<<<1, 64>>>(1000, base, dc, dout); hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("sincosf base %.2f: %.0f cycles/call (dependent chain)\n", base, c / 1000.0);
  }
  kt<<<1, 64>>>(1000, dc, dout); kt<<<1, 64>>>(1000, dc, dout); hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  printf("Sweep::GetTransform: %.0f cycles/call\n", c / 1000.0);
}
