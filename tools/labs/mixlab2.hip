// mixlab2.hip - lab for the state-resident graph-mix kernel: one workgroup per (support k, batch item b),
// X_b resident in LDS, supports streamed in 16x16x4-MFMA A-fragment order straight into registers.
// hipcc -O3 --offload-arch=gfx950 -o mixlab2 mixlab2.hip && ./mixlab2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct MixResArgs {
  const float* Sf;   // [Ks][nRt][nG][64][4]
  const float* X;    // [units][Np][64]
  float* out;        // G[n*sN + b*sT + k*64 + f]
  long sN, sT;
  int Np, N, Ks, nRt, nG;
};

#define MAXNP 416
template <int NT, int MODE>
__global__ __launch_bounds__(512) void k_mix_res(MixResArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[(MAXNP + 16) * 64];
  const int u = blockIdx.x;
  const int k = u % a.Ks, b = u / a.Ks;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int rh = w >> 2, ct = w & 3, j = lane & 15, kq = lane >> 4;
  // A fragment pointers: row tile rt = rh + 2t
  // no runtime conditions around loads / MFMAs (they would serialise the pipeline): tiles past the end
  // are clamped to the last valid tile and simply not stored
  const float4* sf = reinterpret_cast<const float4*>(a.Sf) + ((size_t)k * a.nRt * a.nG) * 64 + lane;
  const int nMine = (a.nRt - rh + 1) >> 1;
  unsigned tofs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tofs[t] = (unsigned)min(rh + 2 * t, a.nRt - 1) * a.nG * 64;
  float4 a0[NT], a1[NT];
  auto loadA = [&](float4 (&dst)[NT], int g) {
#pragma unroll
    for (int t = 0; t < NT; ++t) dst[t] = sf[tofs[t] + g * 64];
  };
  loadA(a0, 0);
  // stage X_b
  {
    const float4* xb = reinterpret_cast<const float4*>(a.X + (size_t)b * a.Np * 64);
    const int total = a.Np * 16;
    for (int idx = tid; idx < total; idx += 512) {
      const int m = idx >> 4, c4 = idx & 15;
      const int c4s = c4 ^ (((m >> 2) & 1) << 2);
      *reinterpret_cast<float4*>(&Xs[m * 64 + c4s * 4]) = xb[idx];
    }
    for (int idx = total + tid; idx < a.nG * 256; idx += 512)   // zero rows of the padding group
      *reinterpret_cast<float4*>(&Xs[idx * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const float* bbase = &Xs[(4 * kq) * 64 + ((ct * 16 + j) ^ ((kq & 1) << 4))];
  auto compute = [&](float4 (&av)[NT], int g) {
    const float* bp = bbase + g * 16 * 64;
    const float b0 = bp[0], b1 = bp[64], b2 = bp[128], b3 = bp[192];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = MFMA16(av[t].x, b0, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = MFMA16(av[t].y, b1, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = MFMA16(av[t].z, b2, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = MFMA16(av[t].w, b3, acc[t]);
  };
  for (int g = 0; g < a.nG; g += 2) {
    if (MODE == 1) { compute(a0, g); compute(a0, g + 1); continue; }
    loadA(a1, g + 1);
    if (MODE != 2) compute(a0, g);
    loadA(a0, min(g + 2, a.nG - 1));
    if (MODE != 2) compute(a1, g + 1);
    else { for (int t = 0; t < NT; ++t) { acc[t][0] += a0[t].x + a1[t].y; } }
  }
  float* obase = a.out + (size_t)b * a.sT + (size_t)k * 64 + ct * 16 + j;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nMine) {
      const int n0 = (rh + 2 * t) * 16 + 4 * kq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + r;
        if (n < a.N) obase[(size_t)n * a.sN] = acc[t][r];
      }
    }
  }
}

// ---- variant (c): supports staged through LDS (one 16-wide k-group per stage, double buffered), X_b resident
struct MixRes2Args {
  const float* Sg;   // [Ks][nG][nRt][64][4]   group-major fragment order
  const float* X;    // [units][Np][64]
  float* out;
  long sN, sT;
  int Np, N, Ks, nRt, nG;
};
template <int NT, int MODE>
__global__ __launch_bounds__(512) void k_mix_res2(MixRes2Args a) {
  __shared__ __attribute__((aligned(16))) float Xs[MAXNP * 64];
  __shared__ __attribute__((aligned(16))) float As[2][(MAXNP / 16) * 256];
  const int u = blockIdx.x;
  const int k = u % a.Ks, b = u / a.Ks;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int rh = w >> 2, ct = w & 3, j = lane & 15, kq = lane >> 4;
  const int nMine = (a.nRt - rh + 1) >> 1;
  const int nA4 = a.nRt * 64;                       // float4 per k-group
  const float4* sg = reinterpret_cast<const float4*>(a.Sg) + (size_t)k * a.nG * nA4;
  int ldIdx[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) ldIdx[q] = min(tid + 512 * q, nA4 - 1);
  float4 ra[4];
  auto loadA = [&](int g) {
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = sg[(size_t)g * nA4 + ldIdx[q]];
  };
  auto storeA = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (tid + 512 * q < nA4) *reinterpret_cast<float4*>(&As[buf][(tid + 512 * q) * 4]) = ra[q];
  };
  loadA(0);
  {
    const float4* xb = reinterpret_cast<const float4*>(a.X + (size_t)b * a.Np * 64);
    const int total = a.Np * 16;
    for (int idx = tid; idx < total; idx += 512) {
      const int m = idx >> 4, c4 = idx & 15;
      const int c4s = c4 ^ (((m >> 2) & 1) << 2);
      *reinterpret_cast<float4*>(&Xs[m * 64 + c4s * 4]) = xb[idx];
    }
  }
  storeA(0);
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  int aofs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) aofs[t] = (min(rh + 2 * t, a.nRt - 1) * 64 + lane) * 4;
  const float* bbase = &Xs[(4 * kq) * 64 + ((ct * 16 + j) ^ ((kq & 1) << 4))];
  __syncthreads();
  float b0 = bbase[0], b1 = bbase[64], b2 = bbase[128], b3 = bbase[192];
  for (int g = 0; g < a.nG; ++g) {
    const int cur = g & 1;
    const int gn = min(g + 1, a.nG - 1);
    if (MODE != 1) loadA(gn);
    float4 av[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) av[t] = *reinterpret_cast<const float4*>(&As[cur][aofs[t]]);
    const float* bp = bbase + gn * 16 * 64;
    const float n0 = bp[0], n1 = bp[64], n2 = bp[128], n3 = bp[192];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = MFMA16(av[t].x, b0, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = MFMA16(av[t].y, b1, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = MFMA16(av[t].z, b2, acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = MFMA16(av[t].w, b3, acc[t]);
    b0 = n0; b1 = n1; b2 = n2; b3 = n3;
    if (MODE != 1) storeA(cur ^ 1);
    __syncthreads();
  }
  float* obase = a.out + (size_t)b * a.sT + (size_t)k * 64 + ct * 16 + j;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nMine) {
      const int n0 = (rh + 2 * t) * 16 + 4 * kq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + r;
        if (n < a.N) obase[(size_t)n * a.sN] = acc[t][r];
      }
    }
  }
}

// ---- variant (f): 4 waves (one per SIMD), wave = 13 row tiles x 2 col tiles (26 accumulators), A direct
template <int NT, int MODE>
__global__ __launch_bounds__(256) void k_mix_res3(MixResArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[(MAXNP + 16) * 64];
  const int u = blockIdx.x;
  const int k = u % a.Ks, b = u / a.Ks;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int rh = w >> 1, cp = w & 1, j = lane & 15, kq = lane >> 4;
  const float4* sf = reinterpret_cast<const float4*>(a.Sf) + ((size_t)k * a.nRt * a.nG) * 64 + lane;
  const int nMine = (a.nRt - rh + 1) >> 1;
  unsigned tofs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) tofs[t] = (unsigned)min(rh + 2 * t, a.nRt - 1) * a.nG * 64;
  float4 a0[NT], a1[NT];
  auto loadA = [&](float4 (&dst)[NT], int g) {
#pragma unroll
    for (int t = 0; t < NT; ++t) dst[t] = sf[tofs[t] + g * 64];
  };
  loadA(a0, 0);
  {
    const float4* xb = reinterpret_cast<const float4*>(a.X + (size_t)b * a.Np * 64);
    const int total = a.Np * 16;
    for (int idx = tid; idx < total; idx += 256) {
      const int m = idx >> 4, c4 = idx & 15;
      const int c4s = c4 ^ (((m >> 2) & 1) << 2);
      *reinterpret_cast<float4*>(&Xs[m * 64 + c4s * 4]) = xb[idx];
    }
    for (int idx = total + tid; idx < a.nG * 256; idx += 256)
      *reinterpret_cast<float4*>(&Xs[idx * 4]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f32x4 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  __syncthreads();
  const float* bb0 = &Xs[(4 * kq) * 64 + ((cp * 32 + j) ^ ((kq & 1) << 4))];
  const float* bb1 = &Xs[(4 * kq) * 64 + ((cp * 32 + 16 + j) ^ ((kq & 1) << 4))];
  auto compute = [&](float4 (&av)[NT], int g) {
    const float* p0 = bb0 + g * 16 * 64;
    const float* p1 = bb1 + g * 16 * 64;
    const float x0 = p0[0], x1 = p0[64], x2 = p0[128], x3 = p0[192];
    const float y0 = p1[0], y1 = p1[64], y2 = p1[128], y3 = p1[192];
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = MFMA16(av[t].x, x0, acc[t][0]); acc[t][1] = MFMA16(av[t].x, y0, acc[t][1]); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = MFMA16(av[t].y, x1, acc[t][0]); acc[t][1] = MFMA16(av[t].y, y1, acc[t][1]); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = MFMA16(av[t].z, x2, acc[t][0]); acc[t][1] = MFMA16(av[t].z, y2, acc[t][1]); }
#pragma unroll
    for (int t = 0; t < NT; ++t) { acc[t][0] = MFMA16(av[t].w, x3, acc[t][0]); acc[t][1] = MFMA16(av[t].w, y3, acc[t][1]); }
  };
  for (int g = 0; g < a.nG; g += 2) {
    if (MODE == 1) { compute(a0, g); compute(a0, g + 1); continue; }
    loadA(a1, g + 1);
    compute(a0, g);
    loadA(a0, min(g + 2, a.nG - 1));
    compute(a1, g + 1);
  }
  float* obase = a.out + (size_t)b * a.sT + (size_t)k * 64 + cp * 32 + j;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nMine) {
      const int n0 = (rh + 2 * t) * 16 + 4 * kq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + r;
        if (n < a.N) { obase[(size_t)n * a.sN] = acc[t][0][r]; obase[(size_t)n * a.sN + 16] = acc[t][1][r]; }
      }
    }
  }
}

// ---- variant 4: 8 waves; wave owns OWN row tiles x all 4 col tiles (A unique per wave) + LEFT leftover
// (row tile, col tile) units; two passes over K so the first pass's stores overlap the second pass's MFMAs;
// X_b staged in 128-row chunks so the first MFMAs start before the whole slab has landed.
template <int NR, int NL>
struct PassRegs {
  f32x4 acc[NR > 0 ? NR : 1][4];
  f32x4 lacc[NL > 0 ? NL : 1];
};

template <int OWN, int LEFT, int MODE>
__global__ __launch_bounds__(512) void k_mix_res4(MixResArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[(MAXNP + 16) * 64];
  const int u = blockIdx.x;
  const int k = u % a.Ks, b = u / a.Ks;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const float4* sf = reinterpret_cast<const float4*>(a.Sf) + ((size_t)k * a.nRt * a.nG) * 64 + lane;
  const int nLeftTiles = a.nRt - 8 * OWN;   // leftover row tiles (0..7)
  // stage X_b: 13 sweeps of 32 rows, all loads issued up front
  constexpr int NSW = (MAXNP + 31) / 32;
  float4 xr[NSW];
  {
    const float4* xb = reinterpret_cast<const float4*>(a.X + (size_t)b * a.Np * 64);
    const int total = a.Np * 16;
#pragma unroll
    for (int q = 0; q < NSW; ++q) xr[q] = xb[min(tid + 512 * q, total - 1)];
  }
  const int xm = tid >> 4, xc4 = tid & 15;
  auto stageRows = [&](int q) {   // sweep q -> rows 32q .. 32q+31
    const int m = 32 * q + xm;
    const int c4s = xc4 ^ (((m >> 2) & 1) << 2);
    float4 v = xr[q];
    if (m >= a.Np) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.nG * 16) *reinterpret_cast<float4*>(&Xs[m * 64 + c4s * 4]) = v;
  };
  const float* bbase = &Xs[(4 * kq) * 64];
  const int sw = (kq & 1) << 4;
  float* obase = a.out + (size_t)b * a.sT + (size_t)k * 64 + j;

  auto run_pass = [&](auto nrTag, auto nlTag, int ownFirst, bool chunked) {
    constexpr int NR = decltype(nrTag)::value, NL = decltype(nlTag)::value;
    f32x4 acc[NR > 0 ? NR : 1][4];
    f32x4 lacc[NL > 0 ? NL : 1];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NL; ++q) lacc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned aofs[NR > 0 ? NR : 1], lofs[NL > 0 ? NL : 1];
    int lcol[NL > 0 ? NL : 1];
#pragma unroll
    for (int r = 0; r < NR; ++r) aofs[r] = (unsigned)(w * OWN + ownFirst + r) * a.nG * 64;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int lu = w + 8 * q;
      const int lrt = min(lu >> 2, max(nLeftTiles - 1, 0));
      lofs[q] = (unsigned)min(8 * OWN + lrt, a.nRt - 1) * a.nG * 64;
      lcol[q] = (((lu & 3) * 16 + j) ^ sw);
    }
    float4 a0[NR > 0 ? NR : 1], a1[NR > 0 ? NR : 1], l0[NL > 0 ? NL : 1], l1[NL > 0 ? NL : 1];
    auto loadA = [&](float4 (&d)[NR > 0 ? NR : 1], float4 (&dl)[NL > 0 ? NL : 1], int g) {
#pragma unroll
      for (int r = 0; r < NR; ++r) d[r] = sf[aofs[r] + g * 64];
#pragma unroll
      for (int q = 0; q < NL; ++q) dl[q] = sf[lofs[q] + g * 64];
    };
    auto compute = [&](float4 (&av)[NR > 0 ? NR : 1], float4 (&lv)[NL > 0 ? NL : 1], int g) {
      const float* bp = bbase + g * 16 * 64;
      float bv[4][4];
      if (NR > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) bv[c][s2] = bp[s2 * 64 + ((c * 16 + j) ^ sw)];
      }
      float lb[NL > 0 ? NL : 1][4];
#pragma unroll
      for (int q = 0; q < NL; ++q)
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) lb[q][s2] = bp[s2 * 64 + lcol[q]];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const float av1 = s2 == 0 ? av[r].x : s2 == 1 ? av[r].y : s2 == 2 ? av[r].z : av[r].w;
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[r][c] = MFMA16(av1, bv[c][s2], acc[r][c]);
        }
#pragma unroll
        for (int q = 0; q < NL; ++q) {
          const float lv1 = s2 == 0 ? lv[q].x : s2 == 1 ? lv[q].y : s2 == 2 ? lv[q].z : lv[q].w;
          lacc[q] = MFMA16(lv1, lb[q][s2], lacc[q]);
        }
      }
    };
    loadA(a0, l0, 0);
    if (chunked) {
      // chunk c = sweeps 4c..4c+3 = rows 128c..128c+127 = groups 8c..8c+7
      constexpr int NCH = (NSW + 3) / 4;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int q = 4 * c; q < 4 * c + 4 && q < NSW; ++q) stageRows(q);
        __syncthreads();
        const int gEnd = min(8 * c + 8, a.nG);
        for (int g = 8 * c; g < gEnd; g += 2) {
          loadA(a1, l1, g + 1);
          compute(a0, l0, g);
          loadA(a0, l0, min(g + 2, a.nG - 1));
          compute(a1, l1, g + 1);
        }
      }
    } else {
      for (int g = 0; g < a.nG; g += 2) {
        loadA(a1, l1, g + 1);
        compute(a0, l0, g);
        loadA(a0, l0, min(g + 2, a.nG - 1));
        compute(a1, l1, g + 1);
      }
    }
    // stores
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int n0 = (w * OWN + ownFirst + r) * 16 + 4 * kq;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + e;
        if (n < a.N) {
#pragma unroll
          for (int c = 0; c < 4; ++c) obase[(size_t)n * a.sN + c * 16] = acc[r][c][e];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int lu = w + 8 * q;
      if ((lu >> 2) < nLeftTiles) {
        const int n0 = (8 * OWN + (lu >> 2)) * 16 + 4 * kq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = n0 + e;
          if (n < a.N) obase[(size_t)n * a.sN + (lu & 3) * 16] = lacc[q][e];
        }
      }
    }
  };
  constexpr int OA = (OWN + 1) / 2;
  if (OA > 0) {
    run_pass(std::integral_constant<int, OA>{}, std::integral_constant<int, 0>{}, 0, true);
    run_pass(std::integral_constant<int, OWN - OA>{}, std::integral_constant<int, LEFT>{}, OA, false);
  } else {
    run_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, LEFT>{}, 0, true);
  }
}

// ---- variant 5: row-tile-major passes.  Each wave walks its row tiles one at a time over the whole K
// (4 accumulators = 4 col tiles), so every pass's stores drain under the next pass's MFMAs.  X_b lives in LDS
// as row-quads Xs4[m/4][col] = {X[4q..4q+3][col]} so a B fragment (4 k-steps) is one ds_read_b128.
template <int OWN, int LEFT, int ST>
__global__ __launch_bounds__(512) void k_mix_res5(MixResArgs a) {
  __shared__ float4 Xs4[(MAXNP / 4) * 64];
  const int u = blockIdx.x;
  const int k = u % a.Ks, b = u / a.Ks;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const float4* sf = reinterpret_cast<const float4*>(a.Sf) + ((size_t)k * a.nRt * a.nG) * 64 + lane;
  const int nLeftTiles = a.nRt - 8 * OWN;
  const float4* xb = reinterpret_cast<const float4*>(a.X + (size_t)b * a.Np * 64);
  const int rqL = tid >> 4, c4 = tid & 15;
  const int nChunks = (a.Np + 127) >> 7;
  // chunk c: rows 128c..128c+127 (32 row-quads x 16 column quads = one 4x4 block per thread)
  float4 xv[4];
  auto chunkLoad = [&](int c) {
    const int rq = 32 * c + rqL;
#pragma unroll
    for (int e = 0; e < 4; ++e) xv[e] = xb[min(4 * rq + e, a.Np - 1) * 16 + c4];
  };
  auto chunkStore = [&](int c) {
    const int rq = 32 * c + rqL;
    if (4 * rq < a.Np) {
      float4* d = &Xs4[rq * 64 + 4 * c4];
      d[0] = make_float4(xv[0].x, xv[1].x, xv[2].x, xv[3].x);
      d[1] = make_float4(xv[0].y, xv[1].y, xv[2].y, xv[3].y);
      d[2] = make_float4(xv[0].z, xv[1].z, xv[2].z, xv[3].z);
      d[3] = make_float4(xv[0].w, xv[1].w, xv[2].w, xv[3].w);
    }
  };
  const float4* bbase = &Xs4[kq * 64 + j];
  float* obase = a.out + (size_t)b * a.sT + (size_t)k * 64 + j;

  // one pass: NA accumulators.  FULLROW: one row tile x 4 col tiles (A shared); else NA independent units.
  auto run_pass = [&](auto naTag, auto fullTag, int rowTile, bool chunked) {
    constexpr int NA = decltype(naTag)::value;
    constexpr bool FULLROW = decltype(fullTag)::value;
    constexpr int NAF = FULLROW ? 1 : NA;     // A fragments per group
    f32x4 acc[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned aofs[NAF];
    int bcol[NA], urow[NA];
    bool uvalid[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      if (FULLROW) { bcol[q] = q * 16; urow[q] = rowTile; uvalid[q] = true; }
      else {
        const int lu = w + 8 * q;
        uvalid[q] = (lu >> 2) < nLeftTiles;
        urow[q] = min(8 * OWN + (lu >> 2), a.nRt - 1);
        bcol[q] = (lu & 3) * 16;
      }
    }
#pragma unroll
    for (int q = 0; q < NAF; ++q) aofs[q] = (unsigned)urow[q] * a.nG * 64;
    float4 ac[NAF], an[NAF], ann[NAF];
    float4 bc[NA], bn[NA];
    auto loadA = [&](float4 (&d)[NAF], int g) {
#pragma unroll
      for (int q = 0; q < NAF; ++q) d[q] = sf[aofs[q] + g * 64];
    };
    auto loadB = [&](float4 (&d)[NA], int g) {
#pragma unroll
      for (int q = 0; q < NA; ++q) d[q] = bbase[g * 256 + bcol[q]];
    };
    auto mfmas = [&]() {
#pragma unroll
      for (int q = 0; q < NA; ++q) acc[q] = MFMA16(ac[FULLROW ? 0 : q].x, bc[q].x, acc[q]);
#pragma unroll
      for (int q = 0; q < NA; ++q) acc[q] = MFMA16(ac[FULLROW ? 0 : q].y, bc[q].y, acc[q]);
#pragma unroll
      for (int q = 0; q < NA; ++q) acc[q] = MFMA16(ac[FULLROW ? 0 : q].z, bc[q].z, acc[q]);
#pragma unroll
      for (int q = 0; q < NA; ++q) acc[q] = MFMA16(ac[FULLROW ? 0 : q].w, bc[q].w, acc[q]);
    };
    auto rotate = [&]() {
#pragma unroll
      for (int q = 0; q < NAF; ++q) { ac[q] = an[q]; an[q] = ann[q]; }
#pragma unroll
      for (int q = 0; q < NA; ++q) bc[q] = bn[q];
    };
    loadA(ac, 0);
    loadA(an, min(1, a.nG - 1));
    if (chunked) {
      chunkLoad(0);
      chunkStore(0);
      if (nChunks > 1) chunkLoad(1);
      __syncthreads();
      loadB(bc, 0);
      for (int c = 0; c < nChunks; ++c) {
        const int gEnd = min(8 * c + 8, a.nG);
        for (int g = 8 * c; g < gEnd; ++g) {
          loadA(ann, min(g + 2, a.nG - 1));
          if (g + 1 < gEnd) loadB(bn, g + 1);
          mfmas();
          rotate();
        }
        if (c + 1 < nChunks) {
          chunkStore(c + 1);
          if (c + 2 < nChunks) chunkLoad(c + 2);
          __syncthreads();
          loadB(bc, 8 * c + 8);
        }
      }
    } else {
      loadB(bc, 0);
      for (int g = 0; g < a.nG; ++g) {
        loadA(ann, min(g + 2, a.nG - 1));
        loadB(bn, min(g + 1, a.nG - 1));
        mfmas();
        rotate();
      }
    }
    if (ST == 1 && FULLROW) {   // timing experiment: same bytes, one full 256-byte row per instruction
      float* ob2 = a.out + (size_t)b * a.sT + (size_t)k * 64 + lane;
#pragma unroll
      for (int q = 0; q < NA; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int n = urow[q] * 16 + 4 * q + e;
          if (n < a.N) ob2[(size_t)n * a.sN] = acc[q][e];
        }
    } else {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      if (uvalid[q]) {
        const int n0 = urow[q] * 16 + 4 * kq;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n0 + e < a.N) obase[(size_t)(n0 + e) * a.sN + bcol[q]] = acc[q][e];
      }
    }
    }
  };
  using T4 = std::integral_constant<int, 4>;
  using TL = std::integral_constant<int, LEFT>;
  if (OWN > 0) {
    run_pass(T4{}, std::true_type{}, w * OWN, true);
    for (int o = 1; o < OWN; ++o) run_pass(T4{}, std::true_type{}, w * OWN + o, false);
    if (LEFT > 0) run_pass(TL{}, std::false_type{}, 0, false);
  } else {
    run_pass(TL{}, std::false_type{}, 0, true);
  }
}

template <int NACC>
__global__ __launch_bounds__(512) void k_reg16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < NACC; ++j) acc[j] = MFMA16(a, b, acc[j]);
  }
  float s = 0;
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  const int N = 403, Np = 416, Ks = 4, B = 64, H = 64;
  const int nRt = Np / 16, nG = (Np / 16 + 1) & ~1;
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<float> S((size_t)Ks * N * N), hX((size_t)B * Np * H, 0.f);
  srand(1);
  for (auto& v : S) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  for (int b = 0; b < B; ++b) for (int m = 0; m < N; ++m) for (int f = 0; f < H; ++f)
    hX[((size_t)b * Np + m) * H + f] = (rand() / (float)RAND_MAX - 0.5f);
  std::vector<float> hSf((size_t)Ks * nRt * nG * 256, 0.f);
  for (int k = 0; k < Ks; ++k) for (int rt = 0; rt < nRt; ++rt) for (int g = 0; g < nG; ++g)
    for (int lane = 0; lane < 64; ++lane) for (int q = 0; q < 4; ++q) {
      const int i = lane & 15, kq = lane >> 4;
      const int n = rt * 16 + i, m = g * 16 + 4 * kq + q;
      float v = (n < N && m < N) ? S[((size_t)k * N + n) * N + m] : 0.f;
      hSf[(((size_t)(k * nRt + rt) * nG + g) * 64 + lane) * 4 + q] = v;
    }
  float *dSf, *dX, *dG;
  CK(hipMalloc(&dSf, hSf.size() * 4)); CK(hipMalloc(&dX, hX.size() * 4));
  const size_t gElems = (size_t)N * B * Ks * H;
  CK(hipMalloc(&dG, gElems * 4));
  CK(hipMemcpy(dSf, hSf.data(), hSf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dG, 0xff, gElems * 4));
  MixResArgs a;
  a.Sf = dSf; a.X = dX; a.out = dG; a.sN = (long)B * Ks * H; a.sT = (long)Ks * H;
  a.Np = Np; a.N = N; a.Ks = Ks; a.nRt = nRt; a.nG = nG;
  hipLaunchKernelGGL((k_mix_res<13, 0>), dim3(Ks * B), dim3(512), 0, s, a);
  CK(hipStreamSynchronize(s));
  std::vector<float> hG(gElems);
  CK(hipMemcpy(hG.data(), dG, gElems * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0;
  srand(7);
  for (int it = 0; it < 20000; ++it) {
    int n = rand() % N, b = rand() % B, k = rand() % Ks, f = rand() % H;
    if (it < 64) { n = (it < 32) ? it * 13 % N : N - 1 - (it - 32); }
    double ref = 0;
    for (int m = 0; m < N; ++m) ref += (double)S[((size_t)k * N + n) * N + m] * hX[((size_t)b * Np + m) * H + f];
    double got = hG[(size_t)n * a.sN + (size_t)b * a.sT + k * 64 + f];
    maxerr = fmax(maxerr, fabs(got - ref)); maxref = fmax(maxref, fabs(ref));
  }
  printf("check: max abs err %.3e (max |ref| %.3e) -> %s\n", maxerr, maxref, maxerr < 1e-5 * fmax(1.0, maxref) * 10 ? "OK" : "FAIL");
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double fl = 2.0 * Ks * N * (double)N * B * H;
  auto timeit = [&](const char* nm, auto&& launch) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < 50; ++i) launch();
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-28s %8.2f us/launch  %7.1f TF/s (algorithmic, unpadded)\n", nm, ms * 1e3 / 50, fl / (ms / 50 * 1e-3) / 1e12);
    }
  };
  {
    const int iters = 1000;
    const double flr = 256.0 * 8 * iters * 4 * 13 * 2048.0;
    auto t2 = [&](const char* nm, double f2, auto&& launch) {
      CK(hipEventRecord(e0, s)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-28s %8.2f us/launch  %7.1f TF/s\n", nm, ms * 1e3 / 10, f2 / (ms / 10 * 1e-3) / 1e12);
    };
    t2("reg16 13acc 2w/simd", flr, [&] { hipLaunchKernelGGL((k_reg16<13>), dim3(256), dim3(512), 0, s, dG, iters); });
    t2("reg16 13acc 2w/simd", flr, [&] { hipLaunchKernelGGL((k_reg16<13>), dim3(256), dim3(512), 0, s, dG, iters); });
    t2("reg16 4acc 2w/simd", flr * 4 / 13, [&] { hipLaunchKernelGGL((k_reg16<4>), dim3(256), dim3(512), 0, s, dG, iters); });
  }
  {
    const int nG1 = Np / 16;
    std::vector<float> hSg((size_t)Ks * nG1 * nRt * 256, 0.f);
    for (int k = 0; k < Ks; ++k) for (int g = 0; g < nG1; ++g) for (int rt = 0; rt < nRt; ++rt)
      for (int lane = 0; lane < 64; ++lane) for (int q = 0; q < 4; ++q) {
        const int i = lane & 15, kq = lane >> 4;
        const int n = rt * 16 + i, m = g * 16 + 4 * kq + q;
        hSg[(((size_t)(k * nG1 + g) * nRt + rt) * 64 + lane) * 4 + q] = (n < N && m < N) ? S[((size_t)k * N + n) * N + m] : 0.f;
      }
    float* dSg; CK(hipMalloc(&dSg, hSg.size() * 4));
    CK(hipMemcpy(dSg, hSg.data(), hSg.size() * 4, hipMemcpyHostToDevice));
    MixRes2Args a2; a2.Sg = dSg; a2.X = dX; a2.out = dG; a2.sN = a.sN; a2.sT = a.sT; a2.Np = Np; a2.N = N; a2.Ks = Ks; a2.nRt = nRt; a2.nG = nG1;
    CK(hipMemset(dG, 0xff, gElems * 4));
    hipLaunchKernelGGL((k_mix_res2<13, 0>), dim3(Ks * B), dim3(512), 0, s, a2);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(hG.data(), dG, gElems * 4, hipMemcpyDeviceToHost));
    double me = 0; srand(7);
    for (int it = 0; it < 20000; ++it) {
      int n = rand() % N, b = rand() % B, k = rand() % Ks, f = rand() % H;
      if (it < 64) { n = (it < 32) ? it * 13 % N : N - 1 - (it - 32); }
      double ref = 0;
      for (int m = 0; m < N; ++m) ref += (double)S[((size_t)k * N + n) * N + m] * hX[((size_t)b * Np + m) * H + f];
      me = fmax(me, fabs(hG[(size_t)n * a.sN + (size_t)b * a.sT + k * 64 + f] - ref));
    }
    printf("res2 check: max abs err %.3e -> %s\n", me, me < 1e-5 ? "OK" : "FAIL");
    timeit("res2 (A via LDS) full", [&] { hipLaunchKernelGGL((k_mix_res2<13, 0>), dim3(Ks * B), dim3(512), 0, s, a2); });
    timeit("res2 no A global loads", [&] { hipLaunchKernelGGL((k_mix_res2<13, 1>), dim3(Ks * B), dim3(512), 0, s, a2); });
  }
  CK(hipMemset(dG, 0xff, gElems * 4));
  hipLaunchKernelGGL((k_mix_res3<13, 0>), dim3(Ks * B), dim3(256), 0, s, a);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(hG.data(), dG, gElems * 4, hipMemcpyDeviceToHost));
  { double me = 0; srand(7);
    for (int it = 0; it < 20000; ++it) {
      int n = rand() % N, b = rand() % B, k = rand() % Ks, f = rand() % H;
      if (it < 64) { n = (it < 32) ? it * 13 % N : N - 1 - (it - 32); }
      double ref = 0;
      for (int m = 0; m < N; ++m) ref += (double)S[((size_t)k * N + n) * N + m] * hX[((size_t)b * Np + m) * H + f];
      me = fmax(me, fabs(hG[(size_t)n * a.sN + (size_t)b * a.sT + k * 64 + f] - ref));
    }
    printf("res3 check: max abs err %.3e -> %s\n", me, me < 1e-5 ? "OK" : "FAIL"); }
  timeit("res3 (4 waves, 26 acc) full", [&] { hipLaunchKernelGGL((k_mix_res3<13, 0>), dim3(Ks * B), dim3(256), 0, s, a); });
  timeit("res3 no A loads", [&] { hipLaunchKernelGGL((k_mix_res3<13, 1>), dim3(Ks * B), dim3(256), 0, s, a); });
  { MixResArgs af = a; af.nG = 2;
    timeit("fixed cost: nG=2 (8 waves)", [&] { hipLaunchKernelGGL((k_mix_res<13, 1>), dim3(Ks * B), dim3(512), 0, s, af); });
    timeit("fixed cost: nG=2 (4 waves)", [&] { hipLaunchKernelGGL((k_mix_res3<13, 1>), dim3(Ks * B), dim3(256), 0, s, af); });
    af.N = 0;
    timeit("fixed, nG=2, no stores", [&] { hipLaunchKernelGGL((k_mix_res<13, 1>), dim3(Ks * B), dim3(512), 0, s, af); });
  }
  CK(hipMemset(dG, 0xff, gElems * 4));
  hipLaunchKernelGGL((k_mix_res4<3, 1, 0>), dim3(Ks * B), dim3(512), 0, s, a);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(hG.data(), dG, gElems * 4, hipMemcpyDeviceToHost));
  { double me = 0; srand(7);
    for (int it = 0; it < 40000; ++it) {
      int n = rand() % N, b = rand() % B, k = rand() % Ks, f = rand() % H;
      if (it < 64) { n = (it < 32) ? it * 13 % N : N - 1 - (it - 32); }
      double ref = 0;
      for (int m = 0; m < N; ++m) ref += (double)S[((size_t)k * N + n) * N + m] * hX[((size_t)b * Np + m) * H + f];
      me = fmax(me, fabs(hG[(size_t)n * a.sN + (size_t)b * a.sT + k * 64 + f] - ref));
    }
    printf("res4 check: max abs err %.3e -> %s\n", me, me < 1e-5 ? "OK" : "FAIL"); }
  timeit("res4 (own 3 + left 1, 2 pass)", [&] { hipLaunchKernelGGL((k_mix_res4<3, 1, 0>), dim3(Ks * B), dim3(512), 0, s, a); });
  {
    // un-padded group count for variant 5 (its Sf uses nG = Np/16 exactly)
    const int nG5 = Np / 16;
    std::vector<float> h5((size_t)Ks * nRt * nG5 * 256, 0.f);
    for (int k = 0; k < Ks; ++k) for (int rt = 0; rt < nRt; ++rt) for (int g = 0; g < nG5; ++g)
      for (int lane = 0; lane < 64; ++lane) for (int q = 0; q < 4; ++q) {
        const int i = lane & 15, kq = lane >> 4;
        const int n = rt * 16 + i, m = g * 16 + 4 * kq + q;
        h5[(((size_t)(k * nRt + rt) * nG5 + g) * 64 + lane) * 4 + q] = (n < N && m < N) ? S[((size_t)k * N + n) * N + m] : 0.f;
      }
    float* d5; CK(hipMalloc(&d5, h5.size() * 4)); CK(hipMemcpy(d5, h5.data(), h5.size() * 4, hipMemcpyHostToDevice));
    MixResArgs a5 = a; a5.Sf = d5; a5.nG = nG5;
    CK(hipMemset(dG, 0xff, gElems * 4));
    hipLaunchKernelGGL((k_mix_res5<3, 1, 0>), dim3(Ks * B), dim3(512), 0, s, a5);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(hG.data(), dG, gElems * 4, hipMemcpyDeviceToHost));
    double me = 0; srand(7);
    for (int it = 0; it < 40000; ++it) {
      int n = rand() % N, b = rand() % B, k = rand() % Ks, f = rand() % H;
      if (it < 64) { n = (it < 32) ? it * 13 % N : N - 1 - (it - 32); }
      double ref = 0;
      for (int m = 0; m < N; ++m) ref += (double)S[((size_t)k * N + n) * N + m] * hX[((size_t)b * Np + m) * H + f];
      me = fmax(me, fabs(hG[(size_t)n * a.sN + (size_t)b * a.sT + k * 64 + f] - ref));
    }
    printf("res5 check: max abs err %.3e -> %s\n", me, me < 1e-5 ? "OK" : "FAIL");
    timeit("res5 (row-tile passes)", [&] { hipLaunchKernelGGL((k_mix_res5<3, 1, 0>), dim3(Ks * B), dim3(512), 0, s, a5); });
    timeit("res5 full-row stores (timing only)", [&] { hipLaunchKernelGGL((k_mix_res5<3, 1, 1>), dim3(Ks * B), dim3(512), 0, s, a5); });
    MixResArgs af = a5; af.N = 0;
    timeit("res5 no stores", [&] { hipLaunchKernelGGL((k_mix_res5<3, 1, 0>), dim3(Ks * B), dim3(512), 0, s, af); });
  }
  { MixResArgs af = a; af.nG = 2;
    timeit("res4 fixed: nG=2", [&] { hipLaunchKernelGGL((k_mix_res4<3, 1, 0>), dim3(Ks * B), dim3(512), 0, s, af); });
    af = a; af.N = 0;
    timeit("res4 no stores", [&] { hipLaunchKernelGGL((k_mix_res4<3, 1, 0>), dim3(Ks * B), dim3(512), 0, s, af); });
  }
  timeit("full", [&] { hipLaunchKernelGGL((k_mix_res<13, 0>), dim3(Ks * B), dim3(512), 0, s, a); });
  timeit("no A loads in loop", [&] { hipLaunchKernelGGL((k_mix_res<13, 1>), dim3(Ks * B), dim3(512), 0, s, a); });
  timeit("loads only, no MFMA", [&] { hipLaunchKernelGGL((k_mix_res<13, 2>), dim3(Ks * B), dim3(512), 0, s, a); });
  return 0;
}
