// nodelab.hip - timing lab for the node-wise kernels (k_gate16 / k_update16) at Baltimore shapes.
// hipcc -O3 --offload-arch=gfx950 -I multistgraph_amd/csrc -o tools/nodelab tools/nodelab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#include "matgcn_node16.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// pure weight-stream probes: same launch geometry as k_gate16<false>, no LDS, no MFMA
template <int PATTERN>
__global__ __launch_bounds__(512) void k_stream(const float* __restrict__ w, float* __restrict__ out, int nG) {
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float4* wp = PATTERN == 0 ? reinterpret_cast<const float4*>(w) + ((size_t)n * nG * 8 + wv) * 64 + lane      // [g][ct] interleaved
                   : PATTERN == 1 ? reinterpret_cast<const float4*>(w) + ((size_t)(n * 8 + wv) * nG) * 64 + lane     // wave-contiguous
                   : reinterpret_cast<const float4*>(w) + (size_t)n * (nG * 8 * 64 + (PATTERN == 2 ? 16 : 80)) + wv * 64 + lane;  // padded node stride
  const size_t gs = PATTERN == 1 ? 64 : 8 * 64;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 r[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) r[i] = wp[(size_t)i * gs];
  for (int g0 = 0; g0 < nG; g0 += 10) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const float4 v = r[i];
      r[i] = wp[(size_t)min(g0 + i + 10, nG - 1) * gs];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (acc.x == 123.456f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// K-loop probes: 8 waves per workgroup, 20 k-groups x 16 MFMAs per wave, features added one at a time
//   0: operands in registers   1: A fragments from the swizzled LDS tile   2: + weights from a global ring
//   3: + the runtime (g < nG) guard of the product kernel
template <int MODE>
__global__ __launch_bounds__(512, 4) void k_probe(const float* __restrict__ wsrc, float* __restrict__ out, int nG, int Ks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds; float* Gs = lds + 64 * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  for (int i = tid; i < 64 * 64 * (1 + Ks); i += 512) lds[i] = 0.001f * (i & 255);
  const float4* wp = reinterpret_cast<const float4*>(wsrc) + ((size_t)blockIdx.x * nG * 8 + w) * 64 + lane;
  float4 wr[N16_RING];
#pragma unroll
  for (int r = 0; r < N16_RING; ++r) wr[r] = (MODE >= 2) ? wp[(size_t)min(r, nG - 1) * 8 * 64] : make_float4(0.1f, 0.2f, 0.3f, 0.4f);
  f32x4 acc[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  float4 creg[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) creg[rt] = make_float4(0.5f + rt, 0.25f, 0.125f, 1.0f);
  if (MODE == 4) {
    float4 avA[4], avB[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) avA[rt] = a_frag(Hs, Gs, Ks, rt, 0, j, kq);
    auto mm = [&](float4 (&av)[4], const float4& wv) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].x, wv.x, acc[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].y, wv.y, acc[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].z, wv.z, acc[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].w, wv.w, acc[rt]);
    };
    for (int g0 = 0; g0 < nG; g0 += N16_RING) {
#pragma unroll
      for (int r = 0; r < N16_RING; r += 2) {
        const int g = g0 + r;
        const float4 w0 = wr[r], w1 = wr[r + 1];
        wr[r] = wp[(size_t)min(g + N16_RING, nG - 1) * 8 * 64];
        wr[r + 1] = wp[(size_t)min(g + 1 + N16_RING, nG - 1) * 8 * 64];
        if (g < nG) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) avB[rt] = a_frag(Hs, Gs, Ks, rt, min(g + 1, nG - 1), j, kq);
          mm(avA, w0);
        }
        if (g + 1 < nG) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) avA[rt] = a_frag(Hs, Gs, Ks, rt, min(g + 2, nG - 1), j, kq);
          mm(avB, w1);
        }
      }
    }
  } else
  for (int g0 = 0; g0 < nG; g0 += N16_RING) {
#pragma unroll
    for (int r = 0; r < N16_RING; ++r) {
      const int g = g0 + r;
      const float4 wv = wr[r];
      if (MODE >= 2) wr[r] = wp[(size_t)min(g + N16_RING, nG - 1) * 8 * 64];
      if (MODE < 3 || g < nG) {
        float4 av[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) av[rt] = (MODE >= 1) ? a_frag(Hs, Gs, Ks, rt, (MODE < 3) ? min(g, nG - 1) : g, j, kq) : creg[rt];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].x, wv.x, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].y, wv.y, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].z, wv.z, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].w, wv.w, acc[rt]);
      }
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) sum += acc[rt][0] + acc[rt][1] + acc[rt][2] + acc[rt][3];
  if (sum == 123.456f) out[tid] = sum;
}

int main(int argc, char** argv) {
  const int N = 403, Np = 416, Ks = 4, B = 64;
  const int nG = 4 * (1 + Ks);
  hipStream_t s; CK(hipStreamCreate(&s));
  auto dalloc = [&](size_t floats, float val) { float* p; CK(hipMalloc(&p, floats * 4)); std::vector<float> h(floats, val);
    for (size_t i = 0; i < floats; i += 97) h[i] = 0.001f * (i % 1000); CK(hipMemcpy(p, h.data(), floats * 4, hipMemcpyHostToDevice)); return p; };
  float* S = dalloc((size_t)B * Np * 64, 0.1f);
  float* G = dalloc((size_t)N * B * Ks * 64, 0.1f);
  float* Wg = dalloc((size_t)512 * (24 * 16 * 128 + 1024), 0.01f);
  float* Wu = dalloc((size_t)N * nG * 16 * 64, 0.01f);
  float* PX = dalloc((size_t)N * B * 192, 0.1f);
  float* ZH = dalloc((size_t)B * Np * 64, 0.f);
  float* R = dalloc((size_t)N * B * 64, 0.5f);
  float* H = dalloc((size_t)B * Np * 64, 0.1f);
  float* XT = dalloc((size_t)B * 24 * Np * 64, 0.1f);
  float* RG = dalloc((size_t)8 * 8 * 64 * 4, 0.01f);
  float* RU = dalloc((size_t)8 * 4 * 64 * 4, 0.01f);
  float* BIAS = dalloc(256, 0.1f);
  float* SEQ = dalloc((size_t)B * 24 * Np * 64, 0.f);
  // a second weight set so consecutive launches alternate (as gate/update of two layers do)
  float* Wg2 = dalloc((size_t)N * nG * 16 * 128, 0.01f);
  const int lds = (64 * 64 + 64 * 64 * 4) * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gate16), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int nb = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_gate16<false>, 512, lds));
  printf("occupancy query: k_gate16 %d blocks/CU at %d B LDS\n", nb, lds);
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_update16<1, false>, 512, lds));
  printf("occupancy query: k_update16<1> %d blocks/CU\n", nb);
  Node16Args a; memset(&a, 0, sizeof(a));
  a.s = S; a.g = G; a.w = Wg; a.px = PX; a.rows = B; a.N = N; a.Np = Np; a.Ks = Ks; a.zh = ZH; a.r = R;
  Node16Args u = a; u.w = Wu; u.h = H; u.hout = H; u.xt = XT; u.xRowStride = (long)24 * Np * 64; u.C = 64; u.Cpad = 64;
  u.rg = RG; u.rgb = BIAS; u.ru = RU; u.rub = BIAS; u.blend = BIAS; u.seq = SEQ; u.seqRowStride = (long)24 * Np * 64;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* nm, double flops, auto&& launch) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 20; ++i) launch(i);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-36s %8.2f us/launch  %7.1f TF/s\n", nm, ms * 1e3 / 20, flops / (ms / 20 * 1e-3) / 1e12);
    }
  };
  const double fg = 2.0 * B * N * 320.0 * 128, fu = 2.0 * B * N * 320.0 * 64 + 2.0 * B * N * 128.0 * 192;
  if (argc > 1) {   // profiling mode: a few launches of one kernel
    for (int i = 0; i < 5; ++i) {
      if (argv[1][0] == 'g') hipLaunchKernelGGL(k_gate16<false>, dim3(N), dim3(512), lds, s, a);
      else hipLaunchKernelGGL((k_update16<1, false>), dim3(N), dim3(512), lds, s, u);
    }
    CK(hipStreamSynchronize(s));
    return 0;
  }
  timeit("stream W only, gate pattern", 66e6 * 4 / 4, [&](int) { hipLaunchKernelGGL(k_stream<0>, dim3(N), dim3(512), 0, s, Wg, ZH, nG); });
  timeit("stream W only, wave-contiguous", 66e6, [&](int) { hipLaunchKernelGGL(k_stream<1>, dim3(N), dim3(512), 0, s, Wg, ZH, nG); });
  timeit("stream W, node stride +256 B", 66e6, [&](int) { hipLaunchKernelGGL(k_stream<2>, dim3(N), dim3(512), 0, s, Wg, ZH, nG); });
  timeit("stream W, node stride +1280 B", 66e6, [&](int) { hipLaunchKernelGGL(k_stream<3>, dim3(N), dim3(512), 0, s, Wg, ZH, nG); });
  timeit("stream W, nG=16 (128 KB stride)", 53e6, [&](int) { hipLaunchKernelGGL(k_stream<0>, dim3(N), dim3(512), 0, s, Wg, ZH, 16); });
  timeit("stream W, nG=16 stride +256 B", 53e6, [&](int) { hipLaunchKernelGGL(k_stream<2>, dim3(N), dim3(512), 0, s, Wg, ZH, 16); });
  timeit("stream G only (26 MB)", 26e6, [&](int) { hipLaunchKernelGGL(k_stream<1>, dim3(N), dim3(512), 0, s, G, ZH, 8); });
  timeit("stream W alternating 2 sets", 66e6, [&](int i) { hipLaunchKernelGGL(k_stream<0>, dim3(N), dim3(512), 0, s, (i & 1) ? Wg2 : Wg, ZH, nG); });
  {
    const double fp = 256.0 * 8 * 24 * 16 * 2048.0;   // nG = 24 = 3 ring cycles, no guard skips
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    timeit("probe 0: register operands", fp, [&](int) { hipLaunchKernelGGL(k_probe<0>, dim3(256), dim3(512), lds, s, Wg, ZH, 24, 4); });
    timeit("probe 1: + A from LDS", fp, [&](int) { hipLaunchKernelGGL(k_probe<1>, dim3(256), dim3(512), lds, s, Wg, ZH, 24, 4); });
    timeit("probe 2: + W global ring", fp, [&](int) { hipLaunchKernelGGL(k_probe<2>, dim3(256), dim3(512), lds, s, Wg, ZH, 24, 4); });
    timeit("probe 3: + runtime guard", fp, [&](int) { hipLaunchKernelGGL(k_probe<3>, dim3(256), dim3(512), lds, s, Wg, ZH, 24, 4); });
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    timeit("probe 4: ping-pong A prefetch", fp, [&](int) { hipLaunchKernelGGL(k_probe<4>, dim3(256), dim3(512), lds, s, Wg, ZH, 24, 4); });
    timeit("probe 4 x2 blocks/CU", 2 * fp, [&](int) { hipLaunchKernelGGL(k_probe<4>, dim3(512), dim3(512), lds, s, Wg, ZH, 24, 4); });
    timeit("probe 3 x2 blocks/CU", 2 * fp, [&](int) { hipLaunchKernelGGL(k_probe<3>, dim3(512), dim3(512), lds, s, Wg, ZH, 24, 4); });
    timeit("probe 1 x2 blocks/CU", 2 * fp, [&](int) { hipLaunchKernelGGL(k_probe<1>, dim3(512), dim3(512), lds, s, Wg, ZH, 24, 4); });
    timeit("probe 0 x2 blocks/CU", 2 * fp, [&](int) { hipLaunchKernelGGL(k_probe<0>, dim3(512), dim3(512), lds, s, Wg, ZH, 24, 4); });
  }
  timeit("gate16 (same W every launch)", fg, [&](int) { hipLaunchKernelGGL(k_gate16<false>, dim3(N), dim3(512), lds, s, a); });
  timeit("gate16 (alternating W sets)", fg, [&](int i) { Node16Args b2 = a; b2.w = (i & 1) ? Wg2 : Wg; hipLaunchKernelGGL(k_gate16<false>, dim3(N), dim3(512), lds, s, b2); });
  timeit("update16<1> (update+res)", fu, [&](int) { hipLaunchKernelGGL((k_update16<1, false>), dim3(N), dim3(512), lds, s, u); });
  timeit("update16<0> (update only)", fu, [&](int) { hipLaunchKernelGGL((k_update16<0, false>), dim3(N), dim3(512), lds, s, u); });
  {
    const int rows = 4 * B, RB = rows / 64;
    float* X4 = dalloc((size_t)rows * Np * 64, 0.1f);
    float* G4 = dalloc((size_t)N * rows * Ks * 64, 0.1f);
    float* Wp = dalloc((size_t)N * nG * 12 * 256, 0.01f);
    float* Bp = dalloc((size_t)N * 192, 0.1f);
    float* PX4 = dalloc((size_t)4 * N * B * 192, 0.f);
    Px16Args p; p.x = X4; p.g = G4; p.w = Wp; p.bias = Bp; p.pxOut = PX4; p.rows = rows; p.N = N; p.Np = Np; p.Ks = Ks; p.B = B;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px16), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    timeit("px16 (4-step chunk)", 2.0 * rows * N * 320.0 * 192, [&](int) { hipLaunchKernelGGL(k_px16, dim3(((N + 7) / 8) * 8 * RB), dim3(512), lds, s, p); });
  }
  { Node16Args b2 = a; b2.N = 256; timeit("gate16 256 nodes only", fg * 256 / N, [&](int) { hipLaunchKernelGGL(k_gate16<false>, dim3(256), dim3(512), lds, s, b2); }); }
  { Node16Args b2 = a; timeit("gate16 128 nodes only", fg * 128 / N, [&](int) { hipLaunchKernelGGL(k_gate16<false>, dim3(128), dim3(512), lds, s, b2); }); }
  return 0;
}
