// mixlab.hip - standalone lab for the graph-mix GEMM: isolates MFMA issue, LDS feed and global staging.
// hipcc -O3 --offload-arch=gfx950 -o mixlab mixlab.hip && ./mixlab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---- V_reg: pure MFMA issue, NACC independent accumulators per wave, WPS waves per SIMD via grid/block
template <int NACC>
__global__ __launch_bounds__(256) void k_reg(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = MFMA32(a, b, acc[j]);
    a += 0.5f;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 64) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

struct MixArgs {
  const float* St; int ldS; const float* X; long xTileStride; int ldX;
  float* out; long sN, sK, sT; int Np, N, Ks, nK, nColTiles, nRowTiles;
};
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---- V0: the shipped kernel (64x64 tile, 4 waves x one 32x32 accumulator, BK=16)
template <int MODE>  // 0 full, 1 no global loads in loop (reuse first tile), 2 no epilogue stores
__global__ __launch_bounds__(256) void k_mix0(MixArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dynsm[];
  float (*As)[16 * 64] = reinterpret_cast<float (*)[16 * 64]>(dynsm);
  float (*Bs)[16 * 64] = reinterpret_cast<float (*)[16 * 64]>(dynsm + 2 * 16 * 64);
  const int id = blockIdx.x;
  int colTile, rowTile;
  if ((a.nColTiles & 7) == 0) { const int xcd = id & 7, j = id >> 3, cpx = a.nColTiles >> 3; rowTile = j % a.nRowTiles; colTile = xcd * cpx + j / a.nRowTiles; }
  else { rowTile = id % a.nRowTiles; colTile = id / a.nRowTiles; }
  const int row0 = rowTile * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1, i = lane & 31, half = lane >> 5;
  const int kk = tid >> 4, sg = tid & 15;
  const float* ap = a.St + (size_t)kk * a.ldS + row0 + sg * 4;
  const float* bp = a.X + (size_t)colTile * a.xTileStride + (size_t)kk * a.ldX + sg * 4;
  float4 ra = *reinterpret_cast<const float4*>(ap);
  float4 rb = *reinterpret_cast<const float4*>(bp);
  *reinterpret_cast<float4*>(&As[0][kk * 64 + sg * 4]) = ra;
  *reinterpret_cast<float4*>(&Bs[0][kk * 64 + sg * 4]) = rb;
  __syncthreads();
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int it = 0; it < a.nK; ++it) {
    const int cur = it & 1;
    const bool more = (it + 1) < a.nK;
    if (more && MODE != 1) {
      ra = *reinterpret_cast<const float4*>(ap + (size_t)(it + 1) * 16 * a.ldS);
      rb = *reinterpret_cast<const float4*>(bp + (size_t)(it + 1) * 16 * a.ldX);
    }
    const float* A = &As[cur][half * 64 + wr * 32 + i];
    const float* Bm = &Bs[cur][half * 64 + wc * 32 + i];
    if (MODE == 3) {
      float av = ra.x, bv = rb.y;
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = MFMA32(av, bv, acc);
    } else if (MODE == 5) {
      float av[8], bv[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) { av[s] = A[s * 128]; bv[s] = Bm[s * 128]; }
#pragma unroll
      for (int s = 0; s < 8; ++s) { asm volatile("" : "+v"(av[s]), "+v"(bv[s])); }
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = MFMA32(av[s], bv[s], acc);
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = MFMA32(A[s * 128], Bm[s * 128], acc);
    }
    if (more) {
      *reinterpret_cast<float4*>(&As[cur ^ 1][kk * 64 + sg * 4]) = ra;
      *reinterpret_cast<float4*>(&Bs[cur ^ 1][kk * 64 + sg * 4]) = rb;
    }
    if (MODE != 4) __syncthreads();
  }
  float* obase = a.out + (size_t)colTile * a.sT + wc * 32 + i;
  if (MODE == 2) { float s = 0; for (int r = 0; r < 16; ++r) s += acc[r]; if (s == 12345.678f) obase[0] = s; return; }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + wr * 32 + acc_row(r, half);
    const int k = row / a.Np, n = row - k * a.Np;
    if (k < a.Ks && n < a.N) obase[(size_t)n * a.sN + (size_t)k * a.sK] = acc[r];
  }
}

// ---- V1: 128x64 tile (rows x cols), 4 waves, each wave 64x32?? -> wave (wr 0..1, wc 0..1): 64 rows x 32 cols = 2 accumulators
// A tile 16 x 128 (8 KB), B tile 16 x 64 (4 KB).  2 independent accumulators per wave share the B fragment.
template <int BK>
__global__ __launch_bounds__(256) void k_mix1(MixArgs a) {
  __shared__ __attribute__((aligned(16))) float As[2][BK * 128];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * 64];
  const int id = blockIdx.x;
  int colTile, rowTile;
  if ((a.nColTiles & 7) == 0) { const int xcd = id & 7, j = id >> 3, cpx = a.nColTiles >> 3; rowTile = j % a.nRowTiles; colTile = xcd * cpx + j / a.nRowTiles; }
  else { rowTile = id % a.nRowTiles; colTile = id / a.nRowTiles; }
  const int row0 = rowTile * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1, i = lane & 31, half = lane >> 5;
  // A: BK rows x 128 floats = BK*32 float4; B: BK rows x 64 = BK*16 float4
  constexpr int NA = BK * 32 / 256, NB = BK * 16 / 256;
  float4 ra[NA], rb[NB > 0 ? NB : 1];
  auto gload = [&](int it) {
#pragma unroll
    for (int u = 0; u < NA; ++u) { int idx = tid + 256 * u; int kk = idx >> 5, sg = idx & 31;
      ra[u] = *reinterpret_cast<const float4*>(a.St + (size_t)(it * BK + kk) * a.ldS + row0 + sg * 4); }
#pragma unroll
    for (int u = 0; u < NB; ++u) { int idx = tid + 256 * u; int kk = idx >> 4, sg = idx & 15;
      rb[u] = *reinterpret_cast<const float4*>(a.X + (size_t)colTile * a.xTileStride + (size_t)(it * BK + kk) * a.ldX + sg * 4); }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NA; ++u) { int idx = tid + 256 * u; *reinterpret_cast<float4*>(&As[buf][idx * 4]) = ra[u]; }
#pragma unroll
    for (int u = 0; u < NB; ++u) { int idx = tid + 256 * u; *reinterpret_cast<float4*>(&Bs[buf][idx * 4]) = rb[u]; }
  };
  gload(0); sstore(0);
  __syncthreads();
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  const int nIt = a.nK * 16 / BK;
  for (int it = 0; it < nIt; ++it) {
    const int cur = it & 1;
    const bool more = (it + 1) < nIt;
    if (more) gload(it + 1);
    const float* A0 = &As[cur][half * 128 + wr * 64 + i];
    const float* Bm = &Bs[cur][half * 64 + wc * 32 + i];
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const float b = Bm[s * 128];
      acc0 = MFMA32(A0[s * 256], b, acc0);
      acc1 = MFMA32(A0[s * 256 + 32], b, acc1);
    }
    if (more) sstore(cur ^ 1);
    __syncthreads();
  }
  float* obase = a.out + (size_t)colTile * a.sT + wc * 32 + i;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = row0 + wr * 64 + acc_row(r, half);
    int k = row / a.Np, n = row - k * a.Np;
    if (k < a.Ks && n < a.N) obase[(size_t)n * a.sN + (size_t)k * a.sK] = acc0[r];
    row += 32; k = row / a.Np; n = row - k * a.Np;
    if (k < a.Ks && n < a.N) obase[(size_t)n * a.sN + (size_t)k * a.sK] = acc1[r];
  }
}

int main() {
  const int N = 403, Np = 416, Ks = 4, B = 64, H = 64;
  const int Mp = 1664;
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<float> hSt((size_t)Np * Mp), hX((size_t)B * Np * H);
  srand(1);
  for (auto& v : hSt) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f);
  for (int m = N; m < Np; ++m) for (int c = 0; c < Mp; ++c) hSt[(size_t)m * Mp + c] = 0.f;
  float *dSt, *dX, *dG, *dOut; unsigned long long* dClk;
  CK(hipMalloc(&dSt, hSt.size() * 4)); CK(hipMalloc(&dX, hX.size() * 4));
  CK(hipMalloc(&dG, (size_t)N * B * Ks * H * 4)); CK(hipMalloc(&dOut, 1 << 24)); CK(hipMalloc(&dClk, 1024));
  CK(hipMemcpy(dSt, hSt.data(), hSt.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double flops, int reps, auto&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %8.2f us/launch  %7.1f TF/s\n", name, ms * 1e3 / reps, flops / (ms / reps * 1e-3) / 1e12);
  };
  // pure MFMA issue: 256 CUs x WPS blocks
  for (int wps = 1; wps <= 4; wps *= 2) {
    int iters = 4096;
    auto run = [&](auto kern, int nacc, const char* nm) {
      char buf[64]; snprintf(buf, 64, "reg nacc=%d waves/simd=%d", nacc, wps);
      timeit(buf, (double)256 * wps * 4 * iters * nacc * 4096.0, 10, [&] { hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, s, dOut, iters, dClk); });
      unsigned long long hc[2]; CK(hipMemcpy(hc, dClk, 16, hipMemcpyDeviceToHost));
      printf("    in-kernel clock %.3f GHz, cycles/MFMA/wave %.1f\n", (double)hc[0] / hc[1] * 0.1, (double)hc[0] / (iters * nacc));
    };
    run(k_reg<1>, 1, ""); run(k_reg<2>, 2, ""); run(k_reg<4>, 4, "");
  }
  MixArgs a;
  a.St = dSt; a.ldS = Mp; a.X = dX; a.xTileStride = (long)Np * H; a.ldX = H; a.out = dG;
  a.sN = (long)B * Ks * H; a.sK = H; a.sT = (long)Ks * H; a.Np = Np; a.N = N; a.Ks = Ks; a.nK = Np / 16; a.nColTiles = B;
  const double fl = 2.0 * Mp * Np * B * H;
  a.nRowTiles = Mp / 64;
  timeit("mix0 64x64 full", fl, 50, [&] { hipLaunchKernelGGL(k_mix0<0>, dim3(a.nRowTiles * B), dim3(256), 16384, s, a); });
  timeit("mix0 64x64 no-global-in-loop", fl, 50, [&] { hipLaunchKernelGGL(k_mix0<1>, dim3(a.nRowTiles * B), dim3(256), 16384, s, a); });
  timeit("mix0 64x64 no-epilogue", fl, 50, [&] { hipLaunchKernelGGL(k_mix0<2>, dim3(a.nRowTiles * B), dim3(256), 16384, s, a); });
  timeit("mix0 no-LDS-reads (reg operands)", fl, 50, [&] { hipLaunchKernelGGL(k_mix0<3>, dim3(a.nRowTiles * B), dim3(256), 16384, s, a); });
  timeit("mix0 no-barrier", fl, 50, [&] { hipLaunchKernelGGL(k_mix0<4>, dim3(a.nRowTiles * B), dim3(256), 16384, s, a); });
  timeit("mix0 all-reads-first", fl, 50, [&] { hipLaunchKernelGGL(k_mix0<5>, dim3(a.nRowTiles * B), dim3(256), 16384, s, a); });
  for (int lds : {16384, 18432, 20480, 23040, 26624, 32768, 40960}) {
    char nm[64]; snprintf(nm, 64, "mix0 full, LDS/WG=%d (max %d WG/CU)", lds, 163840 / lds);
    timeit(nm, fl, 50, [&] { hipLaunchKernelGGL(k_mix0<0>, dim3(a.nRowTiles * B), dim3(256), lds, s, a); });
  }
  a.nRowTiles = Mp / 128;
  timeit("mix1 128x64 2acc BK16", fl, 50, [&] { hipLaunchKernelGGL(k_mix1<16>, dim3(a.nRowTiles * B), dim3(256), 0, s, a); });
  timeit("mix1 128x64 2acc BK32", fl, 50, [&] { hipLaunchKernelGGL(k_mix1<32>, dim3(a.nRowTiles * B), dim3(256), 0, s, a); });
  // big launch (24x columns) to look at steady state
  return 0;
}
