// micro-benchmark: cycles per velocity sweep of the register-resident island solver, in isolation (no surrounding Env)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../boxlcd_amd/csrc/blcd_world.h"
using namespace blcd;

template <int NB, int NJ, int NC>
__global__ __launch_bounds__(64) void k(const float* in, int nc, int sweeps, unsigned long long* cyc, float* out) {
  int lane = threadIdx.x;
  RegIsland<NB, NJ, NC> R;
  R.nj = 0;
  R.nc = nc;
  R.deadQ = 0;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    R.vel[i].v = V2(in[lane] * 0.3f + i, -1.0f - in[64 + lane]);
    R.vel[i].w = in[128 + lane];
    R.pos[i].c = V2(1.0f + i, 0.7f);
    R.pos[i].a = 0.1f * i;
    R.mass[i].invMass = 5.0f;
    R.mass[i].invI = 8.0f;
    R.mass[i].lc = V2(0, 0);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    RContact& ct = R.ct[c];
    ct.pA = c == NC - 1 && NB > 1 ? 4 : 0;   // last contact: body-body, others wall-body
    ct.pB = 4 + (c % NB);
    ct.pointCount = 2;
    ct.normal = V2(0.0f, 1.0f);
    ct.friction = 0.5f;
    ct.restitution = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ct.points[j].rA = V2(0.3f * j, -0.2f);
      ct.points[j].rB = V2(-0.4f + 0.8f * j + 0.01f * in[lane], -0.7f);
      ct.points[j].normalImpulse = 0.01f;
      ct.points[j].tangentImpulse = 0.0f;
      ct.points[j].normalMass = 0.1f;
      ct.points[j].tangentMass = 0.1f;
      ct.points[j].velocityBias = 0.0f;
    }
    ct.K.ex = V2(10.0f, 2.0f);
    ct.K.ey = V2(2.0f, 10.0f);
    ct.normalMass = ct.K.GetInverse();
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int done = R.velocitySweeps(sweeps, 1.0f / 30.0f);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = done; }
  float acc = 0;
#pragma unroll
  for (int i = 0; i < NB; ++i) acc += R.vel[i].v.x + R.vel[i].v.y + R.vel[i].w;
  out[lane] = acc;
}

int main() {
  float hin[192];
  for (int i = 0; i < 192; ++i) hin[i] = (float)((i * 37) % 101) / 101.0f;
  float* din; hipMalloc(&din, sizeof(hin)); hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  unsigned long long* dc; hipMalloc(&dc, 16); float* dout; hipMalloc(&dout, 256);
  for (int nc = 1; nc <= 4; ++nc) {
    for (int rep = 0; rep < 2; ++rep) k<2, 0, 4><<<1, 64>>>(din, nc, 180, dc, dout);
    hipDeviceSynchronize();
    unsigned long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    printf("RegIsland<2,0,4> contacts %d: %llu sweeps, %.0f cycles/sweep, %.0f cycles/contact-sweep\n", nc, c[1], (double)c[0] / c[1], (double)c[0] / c[1] / nc);
  }
  return 0;
}
