// gates_lab.hip - accuracy of the step kernels' fast gate non-linearities (sigmoid16 / tanh16 of matgcn_node16.hip)
// against double precision, next to the libm forms they replace.
//   hipcc -O3 --offload-arch=gfx950 -I multistgraph_amd/csrc -o tools/labs/gates_lab tools/labs/gates_lab.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA16BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ unsigned int bf16_rne(float f) { unsigned int u = __float_as_uint(f); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
#include "matgcn_node16.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_eval(const float* x, float* sf, float* sp, float* tf, float* tp, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  sf[i] = sigmoid16(x[i]); sp[i] = 1.0f / (1.0f + expf(-x[i]));
  tf[i] = tanh16(x[i]); tp[i] = tanhf(x[i]);
}

static double ulp_of(double ref) { float f = (float)fabs(ref); int e; frexpf(f, &e); return ldexp(1.0, e - 24); }

int main() {
  std::vector<float> x;
  for (double v = -30.0; v <= 30.0; v += 1e-4) x.push_back((float)v);
  for (double e = -20; e <= 1.0; e += 0.001) { x.push_back((float)pow(10.0, e)); x.push_back(-(float)pow(10.0, e)); }
  const int n = (int)x.size();
  float *dx, *d[4]; CK(hipMalloc(&dx, n * 4)); for (auto& p : d) CK(hipMalloc(&p, n * 4));
  CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_eval, dim3((n + 255) / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], n);
  CK(hipDeviceSynchronize());
  std::vector<float> r[4]; for (int k = 0; k < 4; ++k) { r[k].resize(n); CK(hipMemcpy(r[k].data(), d[k], n * 4, hipMemcpyDeviceToHost)); }
  const char* nm[4] = {"sigmoid16 (v_exp, v_rcp)", "1/(1+expf(-x))", "tanh16 (series | (1-t)/(1+t))", "tanhf"};
  for (int k = 0; k < 4; ++k) {
    double maxUlp = 0, maxAbs = 0, atU = 0, atA = 0;
    for (int i = 0; i < n; ++i) {
      const double xi = x[i], ref = k < 2 ? 1.0 / (1.0 + exp(-xi)) : tanh(xi);
      const double err = fabs((double)r[k][i] - ref);
      if (ref != 0 && err / ulp_of(ref) > maxUlp) { maxUlp = err / ulp_of(ref); atU = xi; }
      if (err > maxAbs) { maxAbs = err; atA = xi; }
    }
    printf("%-32s max error %.2f ulp (at x = %g), max abs error %.3g (at x = %g) over %d points\n", nm[k], maxUlp, atU, maxAbs, atA, n);
  }
  return 0;
}
