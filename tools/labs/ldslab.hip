#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// MODE 0: read pair -> 2 MFMA (compiler order); MODE 1: software pipelined (next operands read before current MFMAs)
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float As[2][16 * 64];
  __shared__ __attribute__((aligned(16))) float Bs[2][16 * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1, i = lane & 31, half = lane >> 5;
  for (int q = tid; q < 2 * 16 * 64; q += 256) { (&As[0][0])[q] = q * 0.001f; (&Bs[0][0])[q] = 1.0f + q * 0.002f; }
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
      const int cur = it & 1;
      const float* A = &As[cur][half * 64 + wr * 32 + i];
      const float* Bm = &Bs[cur][half * 64 + wc * 32 + i];
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = MFMA32(A[s * 128], Bm[s * 128], acc);
    }
  } else {
    float av[8], bv[8], an[8], bn[8];
    {
      const float* A = &As[0][half * 64 + wr * 32 + i];
      const float* Bm = &Bs[0][half * 64 + wc * 32 + i];
#pragma unroll
      for (int s = 0; s < 8; ++s) { av[s] = A[s * 128]; bv[s] = Bm[s * 128]; }
    }
    for (int it = 0; it < iters; ++it) {
      const int nxt = (it + 1) & 1;
      const float* A = &As[nxt][half * 64 + wr * 32 + i];
      const float* Bm = &Bs[nxt][half * 64 + wc * 32 + i];
#pragma unroll
      for (int s = 0; s < 8; ++s) { an[s] = A[s * 128]; bn[s] = Bm[s * 128]; }
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = MFMA32(av[s], bv[s], acc);
#pragma unroll
      for (int s = 0; s < 8; ++s) { av[s] = an[s]; bv[s] = bn[s]; }
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  float* dOut; CK(hipMalloc(&dOut, 1 << 24));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* nm, int wps) {
    int iters = 1024;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, s, dOut, iters);
    CK(hipStreamSynchronize(s)); CK(hipEventRecord(e0, s));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, s, dOut, iters);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    double fl = 256.0 * wps * 4 * iters * 8 * 4096.0;
    printf("%-28s waves/simd=%d  %7.1f TF/s\n", nm, wps, fl / (ms * 1e-3) / 1e12);
  };
  for (int wps : {1, 2, 4, 6, 8}) { run(k_lds<0>, "lds->mfma compiler order", wps); run(k_lds<1>, "lds->mfma sw pipelined", wps); }
  return 0;
}
