// micro-benchmark: cycles per TimeOfImpact call (register-resident wall version vs generic), uniform vs divergent lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../boxlcd_amd/csrc/blcd_toi_wall.h"
using namespace blcd;

__global__ void k(const Shape* shapes, int mode, int reps, unsigned long long* cyc, float* out) {
  int lane = threadIdx.x;
  Sweep sw;
  sw.localCenter = V2(0, 0);
  float y0 = 0.9f + (mode & 1 ? 0.01f * lane : 0.0f);
  sw.c0 = V2(2.0f, y0);
  sw.c = V2(2.0f + (mode & 1 ? 0.003f * lane : 0.0f), y0 - 0.6f);
  sw.a0 = 0.3f; sw.a = 0.3f + (mode & 1 ? 0.01f * lane : 0.02f);
  sw.alpha0 = 0.0f;
  Shape edge; ShapeSetEdge(&edge, V2(0, 0), V2(5, 0));
  float acc = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    TOIOutput o;
    if (mode & 2) {
      DistanceProxy pa, pb; pa.Set(&edge); pb.Set(&shapes[(mode >> 2) & 1]);
      Sweep sa; sa.localCenter = sa.c0 = sa.c = V2(0, 0); sa.a0 = sa.a = 0; sa.alpha0 = 0;
      TimeOfImpact(&o, &pa, sa, &pb, sw, 1.0f);
    } else {
      TOIWall<8> tw;
      tw.A.a0 = V2(0, 0); tw.A.a1 = V2(5, 0); tw.A.radius = kPolygonRadius;
      tw.B.load(&shapes[(mode >> 2) & 1]);
      tw.run(&o, sw);
    }
    acc += o.t + o.state;
    sw.c.y -= 1e-6f;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) cyc[0] = t1 - t0;
  out[lane] = acc;
}

int main() {
  Shape hs[2];
  ShapeSetCircle(&hs[0], 0.5f);
  ShapeSetAsBox(&hs[1], 0.7f, 0.7f);
  Shape* ds; hipMalloc(&ds, sizeof(hs)); hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
  unsigned long long* dc; hipMalloc(&dc, 8); float* dout; hipMalloc(&dout, 256);
  const char* names[] = {"wall/uniform/circle", "wall/divergent/circle", "generic/uniform/circle", "generic/divergent/circle",
                         "wall/uniform/box", "wall/divergent/box", "generic/uniform/box", "generic/divergent/box"};
  for (int mode = 0; mode < 8; ++mode) {
    int reps = 200;
    k<<<1, 64>>>(ds, mode, reps, dc, dout);
    k<<<1, 64>>>(ds, mode, reps, dc, dout);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    float o[64]; hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost);
    printf("%-28s cycles/call %8.0f   (lane0 acc %.4f lane63 %.4f)\n", names[mode], (double)c / reps, o[0] / reps, o[63] / reps);
  }
  return 0;
}
