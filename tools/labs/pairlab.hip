// pairlab.hip - round 4: do the step kernels of the two chains hide beside each other?
// The wavefront forward runs the two layers' chains on two streams; in-situ events say every kernel is stretched (k_mix 41 ->
// 75 us on average) and the wall is the sum of the stretched kernels of one chain.  A graph mix is MFMA-bound with HBM idle,
// a node kernel streams weights with the matrix pipe a third busy: this lab times the PAIRS at Baltimore shapes (N = 403,
// B = 64, Ks = 3) - each kernel alone, then two kernels launched together on two streams (wall until both are done):
//   mix || mix,   mix || gate,   mix || update,   gate || update,   gate || gate
//   hipcc -O3 --offload-arch=gfx950 -I multistgraph_amd/csrc -o tools/labs/pairlab tools/labs/pairlab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "matgcn_internal.h"
#include "matgcn_kernels.hip"
#include "matgcn_node16.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
  const int N = 403, Np = 416, Ks = 3, B = 64, H = 64;
  const int nG = 4 * (1 + Ks);
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  auto dalloc = [&](size_t floats, float val) { float* p; CK(hipMalloc(&p, floats * 4)); std::vector<float> h(floats, val);
    for (size_t i = 0; i < floats; i += 97) h[i] = 0.001f * (i % 1000); CK(hipMemcpy(p, h.data(), floats * 4, hipMemcpyHostToDevice)); return p; };
  // two independent sets (chain A, chain B)
  struct Set { float *St, *X, *G, *S, *Wg, *Wu, *PX, *ZH, *R, *Hs, *XT, *SEQ; } set[2];
  float* RG = dalloc((size_t)8 * 8 * 64 * 4, 0.01f);
  float* RU = dalloc((size_t)8 * 4 * 64 * 4, 0.01f);
  float* BIAS = dalloc(256, 0.1f);
  for (auto& q : set) {
    q.St = dalloc((size_t)Np * Ks * Np, 0.001f);
    q.X = dalloc((size_t)B * Np * H, 0.1f);
    q.G = dalloc((size_t)N * B * Ks * H, 0.1f);
    q.S = dalloc((size_t)B * Np * H, 0.1f);
    q.Wg = dalloc((size_t)N * nG * 16 * 128, 0.01f);
    q.Wu = dalloc((size_t)N * nG * 16 * 64, 0.01f);
    q.PX = dalloc((size_t)N * NODE_PX_BLOCK, 0.1f);
    q.ZH = dalloc((size_t)B * Np * H, 0.f);
    q.R = dalloc((size_t)N * NODE_R_BLOCK, 0.5f);
    q.Hs = dalloc((size_t)B * Np * H, 0.1f);
    q.XT = dalloc((size_t)B * Np * H, 0.1f);
    q.SEQ = dalloc((size_t)B * Np * H, 0.f);
  }
  const int ldsG = 3 * 4096 * 4, ldsU = 4 * 4096 * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gate16<false, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsG));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<1, false, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsU));
  auto mix = [&](const Set& q, hipStream_t s) {
    MixArgs a;
    a.St = q.St; a.ldS = Ks * Np; a.X = q.X; a.xTileStride = (long)Np * H; a.ldX = H;
    a.out = q.G; a.sN = (long)B * Ks * H; a.sK = H; a.sT = (long)Ks * H; a.outFloats = (long)(N - 1) * a.sN + (long)B * Ks * H;
    a.Np = Np; a.N = N; a.Ks = Ks; a.nK = Np / 16; a.nColTiles = B; a.nRowTiles = (Ks * Np + 63) / 64;
    hipLaunchKernelGGL(k_mix<1>, dim3((unsigned)(a.nRowTiles * a.nColTiles)), dim3(256), 0, s, a);
  };
  auto node_args = [&](const Set& q) {
    Node16Args a; memset(&a, 0, sizeof(a));
    a.s = q.S; a.g = q.G; a.w = q.Wg; a.px = q.PX; a.rows = B; a.N = N; a.Np = Np; a.Ks = Ks; a.zh = q.ZH; a.r = q.R;
    return a;
  };
  auto gate = [&](const Set& q, hipStream_t s) {
    Node16Args a = node_args(q);
    hipLaunchKernelGGL((k_gate16<false, 64>), dim3(node_items(N, B, 64)), dim3(512), ldsG, s, a);
  };
  auto update = [&](const Set& q, hipStream_t s) {
    Node16Args u = node_args(q);
    u.w = q.Wu; u.h = q.Hs; u.hout = q.Hs; u.xt = q.XT; u.xRowStride = (long)Np * 64; u.C = 64; u.Cpad = 64;
    u.rg = RG; u.rgb = BIAS; u.ru = RU; u.rub = BIAS; u.blend = BIAS; u.seq = q.SEQ; u.seqRowStride = (long)Np * 64;
    hipLaunchKernelGGL((k_update16<1, false, 64>), dim3(node_items(N, B, 64)), dim3(512), ldsU, s, u);
  };
  hipEvent_t e0, e1, f; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&f, hipEventDisableTiming));
  // one kernel alone: chain of `reps` launches on one stream
  auto alone = [&](const char* nm, auto&& launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s1));
      for (int i = 0; i < 20; ++i) launch(set[0], s1);
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    printf("%-10s alone            %7.2f us per launch\n", nm, best * 1e3 / 20);
    return best * 1e3f / 20;
  };
  // a pair: both launched at the same moment on two streams, 20 rounds; a round ends when BOTH are done (s1 waits for s2)
  auto pair = [&](const char* na, const char* nb, auto&& la, auto&& lb, float ta, float tb) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      for (int i = 0; i < 20; ++i) {
        CK(hipEventRecord(f, s1)); CK(hipStreamWaitEvent(s2, f, 0));     // both start behind the previous round
        la(set[0], s1); lb(set[1], s2);
        CK(hipEventRecord(f, s2)); CK(hipStreamWaitEvent(s1, f, 0));
      }
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    const float t = best * 1e3f / 20;
    printf("%-10s || %-10s   %7.2f us per pair   (sum of the two alone %.2f, the longer alone %.2f: hidden %.0f %% of the shorter)\n",
           na, nb, t, ta + tb, ta > tb ? ta : tb, 100.f * (ta + tb - t) / (ta < tb ? ta : tb));
  };
  const float tm = alone("k_mix<1>", mix), tg = alone("k_gate16", gate), tu = alone("k_update16", update);
  pair("k_mix<1>", "k_mix<1>", mix, mix, tm, tm);
  pair("k_mix<1>", "k_gate16", mix, gate, tm, tg);
  pair("k_mix<1>", "k_update16", mix, update, tm, tu);
  pair("k_gate16", "k_update16", gate, update, tg, tu);
  pair("k_gate16", "k_gate16", gate, gate, tg, tg);
  return 0;
}
