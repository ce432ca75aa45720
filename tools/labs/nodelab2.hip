// nodelab2.hip - timing lab for the pipelined node kernels (k_gate16 / k_update16 of matgcn_node16.hip) at Baltimore
// shapes (Ks = 3 dense slots, B = 64), swept over the number of nodes = workgroups.
//   hipcc -O3 --offload-arch=gfx950 -I multistgraph_amd/csrc [-DNODE_LAB_NO_MFMA] [-DNODE_LAB_NO_WEIGHTS] -o tools/nodelab2 tools/nodelab2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#include "matgcn_node16.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty(float* p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 1.f; }

int main(int argc, char** argv) {
  const int NMAX = 1024, Np = 1024, Ks = 3, B = 64;
  const int nG = 4 * (1 + Ks);
  hipStream_t s; CK(hipStreamCreate(&s));
  auto dalloc = [&](size_t floats, float val) { float* p; CK(hipMalloc(&p, floats * 4)); std::vector<float> h(floats, val);
    for (size_t i = 0; i < floats; i += 97) h[i] = 0.001f * (i % 1000); CK(hipMemcpy(p, h.data(), floats * 4, hipMemcpyHostToDevice)); return p; };
  float* S = dalloc((size_t)B * Np * 64, 0.1f);
  float* G = dalloc((size_t)NMAX * B * Ks * 64, 0.1f);
  float* Wg = dalloc((size_t)NMAX * nG * 16 * 128, 0.01f);
  float* Wu = dalloc((size_t)NMAX * nG * 16 * 64, 0.01f);
  float* PX = dalloc((size_t)NMAX * NODE_PX_BLOCK, 0.1f);
  float* ZH = dalloc((size_t)B * Np * 64, 0.f);
  float* R = dalloc((size_t)NMAX * NODE_R_BLOCK, 0.5f);
  float* H = dalloc((size_t)B * Np * 64, 0.1f);
  float* XT = dalloc((size_t)B * Np * 64, 0.1f);
  float* RG = dalloc((size_t)8 * 8 * 64 * 4, 0.01f);
  float* RU = dalloc((size_t)8 * 4 * 64 * 4, 0.01f);
  float* BIAS = dalloc(256, 0.1f);
  float* SEQ = dalloc((size_t)B * Np * 64, 0.f);
  const int ldsG = 3 * 4096 * 4, ldsU = 4 * 4096 * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gate16<false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsG));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsU));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsU));
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_gate16<false>, 512, ldsG));
  printf("occupancy: k_gate16 %d blocks/CU", nb);
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_update16<1, false>, 512, ldsU));
  printf(", k_update16<1> %d blocks/CU\n", nb);
  Node16Args a; memset(&a, 0, sizeof(a));
  a.s = S; a.g = G; a.w = Wg; a.px = PX; a.rows = B; a.N = NMAX; a.Np = Np; a.Ks = Ks; a.zh = ZH; a.r = R;
  Node16Args u = a; u.w = Wu; u.h = H; u.hout = H; u.xt = XT; u.xRowStride = (long)Np * 64; u.C = 64; u.Cpad = 64;
  u.rg = RG; u.rgb = BIAS; u.ru = RU; u.rub = BIAS; u.blend = BIAS; u.seq = SEQ; u.seqRowStride = (long)Np * 64;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* nm, int nodes, double flops, double bytes, auto&& launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 20; ++i) launch();
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    const double us = best * 1e3 / 20;
    printf("%-22s N=%4d %7.2f us/launch  %6.1f TF/s  %5.2f TB/s\n", nm, nodes, us, flops / us / 1e6, bytes / us / 1e6);
  };
  timeit("empty kernel", 403, 0, 0, [&]() { hipLaunchKernelGGL(k_empty, dim3(403), dim3(512), 0, s, ZH); });
  const int sweep[] = {64, 128, 256, 403, 512, 768, 1024};
  for (int nodes : sweep) {
    // bytes per node: weights + G + s + PX + outputs
    const double bg = (double)nG * 16 * 128 * 4 + B * Ks * 256.0 + B * 256.0 + B * 128 * 4.0 + B * 256.0 * 2;
    timeit("gate16", nodes, 2.0 * B * nodes * (64.0 * (1 + Ks)) * 128, bg * nodes,
           [&]() { hipLaunchKernelGGL(k_gate16<false>, dim3(nodes), dim3(512), ldsG, s, a); });
  }
  for (int nodes : sweep) {
    const double bu = (double)nG * 16 * 64 * 4 + B * Ks * 256.0 + B * 256.0 * 5 + B * 64 * 4.0;
    timeit("update16<1>", nodes, 2.0 * B * nodes * (64.0 * (1 + Ks)) * 64 + 2.0 * B * nodes * 128.0 * 192, bu * nodes,
           [&]() { hipLaunchKernelGGL((k_update16<1, false>), dim3(nodes), dim3(512), ldsU, s, u); });
  }
  for (int nodes : {256, 403}) {
    const double bu = (double)nG * 16 * 64 * 4 + B * Ks * 256.0 + B * 256.0 * 4 + B * 64 * 4.0;
    timeit("update16<0>", nodes, 2.0 * B * nodes * (64.0 * (1 + Ks)) * 64, bu * nodes,
           [&]() { hipLaunchKernelGGL((k_update16<0, false>), dim3(nodes), dim3(512), ldsU, s, u); });
  }
  return 0;
}
