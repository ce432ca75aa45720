#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 64) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, unsigned long long* clk) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 64) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float* dOut; unsigned long long* dClk; CK(hipMalloc(&dOut, 1 << 24)); CK(hipMalloc(&dClk, 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* nm, int nacc, int wps, double flopPer) {
    int iters = 8192 / nacc;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, s, dOut, iters, dClk);
    CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, s, dOut, iters, dClk);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    unsigned long long hc[2]; CK(hipMemcpy(hc, dClk, 16, hipMemcpyDeviceToHost));
    double fl = 256.0 * wps * 4 * iters * nacc * flopPer;
    printf("%-10s nacc=%d waves/simd=%d  %7.1f TF/s  clk %.2f GHz  cycles/MFMA/SIMD %.1f\n", nm, nacc, wps, fl / (ms * 1e-3) / 1e12,
           (double)hc[0] / hc[1] * 0.1, (double)hc[0] / (iters * nacc) / wps);
  };
  for (int wps = 1; wps <= 8; wps *= 2) {
    run(k32<1>, "32x32x2", 1, wps, 4096); run(k32<2>, "32x32x2", 2, wps, 4096); run(k32<4>, "32x32x2", 4, wps, 4096);
    run(k16<2>, "16x16x4", 2, wps, 2048); run(k16<4>, "16x16x4", 4, wps, 2048); run(k16<8>, "16x16x4", 8, wps, 2048);
  }
  return 0;
}
