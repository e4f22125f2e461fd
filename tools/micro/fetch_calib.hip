// FETCH_SIZE / WRITE_SIZE calibration for THIS path's access patterns (MI355X_MICROARCH.md "HBM": gfx950 FETCH_SIZE reports
// exactly 1/2 of the bytes of a 16 B/lane coalesced streaming read; "other access widths are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Three kernels over a buffer well past the 256 MiB Infinity Cache, each reading or
// writing a KNOWN byte count, run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/fetch_calib.sh):
//   read4   one dword per lane, field-major like step_kernel's state loads (a wave reads 256 contiguous bytes per field)
//   read16  one dwordx4 per lane (the guide's calibrated case: expect counter = bytes / 2)
//   write4 / write16  the store side
// hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calib.hip -o tools/micro/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void read4(const float* __restrict__ a, size_t nPerField, int fields, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPerField) return;
  float s = 0.0f;
  for (int f = 0; f < fields; ++f) s += a[(size_t)f * nPerField + i];   // [field][slot], slot fastest: the state layout
  if (s == 123.456f) out[0] = s;
}
__global__ void read16(const float4* __restrict__ a, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = a[i];
  if (v.x + v.y + v.z + v.w == 123.456f) out[0] = v.x;
}
__global__ void write4(float* __restrict__ a, size_t nPerField, int fields) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nPerField) return;
  for (int f = 0; f < fields; ++f) a[(size_t)f * nPerField + i] = (float)f;
}
__global__ void write16(float4* __restrict__ a, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  a[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
  const size_t bytes = (size_t)1 << 30;   // 1 GiB: 4x the Infinity Cache
  const int fields = 64;
  const size_t nPerField = bytes / 4 / fields, n16 = bytes / 16;
  float *a, *out;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&out, 4));
  CK(hipMemset(a, 0, bytes));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read4, dim3((nPerField + 255) / 256), dim3(256), 0, 0, a, nPerField, fields, out);
    hipLaunchKernelGGL(read16, dim3((n16 + 255) / 256), dim3(256), 0, 0, (const float4*)a, n16, out);
    hipLaunchKernelGGL(write4, dim3((nPerField + 255) / 256), dim3(256), 0, 0, a, nPerField, fields);
    hipLaunchKernelGGL(write16, dim3((n16 + 255) / 256), dim3(256), 0, 0, (float4*)a, n16);
  }
  CK(hipDeviceSynchronize());
  printf("bytes per kernel: %zu\n", bytes);
  return 0;
}
