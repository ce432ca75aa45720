// matgcn_bwd_kernels.hip - kernels of the backward pass: autograd of
// MultiATGCN.forward (MultiATGCN.py:76-128,142-150,194-212,363-420) as explicit HIP kernels.
//
// Round-1 shape of the backward: correctness first.  Every contraction of the backward is an instance of ONE
// strided, two-level-batched fp32 GEMM on the exact-fp32 matrix cores (k_bgemm, v_mfma_f32_16x16x4_f32, 64 x 64
// tiles through LDS); the pointwise algebra of the two GRU cells, the blend, the softmaxes and the head fusion
// lives in small element-wise kernels.  The schedule (matgcn_bwd.hip) keeps the time-sequential chain minimal -
// only the recurrent (h) columns are back-propagated step by step - and batches everything else over all T steps:
// the x columns, every weight gradient and the gradient of the adaptive adjacency.
#ifndef MATGCN_BWD_KERNELS_HIP
#define MATGCN_BWD_KERNELS_HIP

// f32x4 / MFMA16 / sigmoid_f / cheb_scalar / StackMap come from matgcn_kernels.hip (same translation unit)

// ---- generic strided batched GEMM ------------------------------------------------------------------------
//   C[b1][b2][m][n] (+)= alpha * sum_{k2 < K2} sum_{k < K} A[b1][b2][m][k2][k] * B[b1][b2][k2][k][n]
// element (m, k2, k) of A sits at A + b1*bA1 + b2*bA2 + m*sAm + k2*sAk2 + k*sAk (B, C alike): transposes, slices
// and interleaved layouts are all strides.  mode 0: C = alpha*acc + beta*C; mode 1: atomicAdd(C, alpha*acc) with the
// (k2, k) range additionally split over `split` workgroups (C must hold its initial value beforehand).
struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, K2;
  long sAm, sAk, sAk2, sBk, sBn, sBk2, sCm, sCn;
  int nb2;                       // inner batch count: blockIdx.z = (b1 * nb2 + b2) * split + part
  long bA1, bA2, bB1, bB2, bC1, bC2;
  float alpha, beta;
  int mode, split;
  long offA[8], offB[8], offC[8];  // k_bgemm only, nOff > 0: outer batch item b1 < nOff sits at these offsets (floats)
  int nOff;                        // instead of b1 * bA1 / bB1 / bC1 (batch items that are not equally spaced: the pools)
  const float* scaleC;           // k_bgemm only: C = scaleC[b1*bS1 + b2*bS2 + m*sSm + n*sSn] * (alpha A.B) (the dropout mask of
  long sSm, sSn, bS1, bS2;       // the head, applied where the gradient of the head's input is produced)
  float* colsumA;                // k_bgemm_tn only: [M] += column sums of A over the whole reduction (nn.Linear bias
                                 // gradients: A = the pre-activation gradients), added by the workgroups of column tile 0
};

#define BG_LD 80   // LDS row pitch (floats): 80 = 16 mod 32, so k and k + 1 sit half a bank row apart
// column swizzle of the staged tiles: element (k, x) lives in column x ^ (8 * ((k >> 2) & 3)).  ds_write_b32 and
// ds_read_b32 bank by (address / 4) mod 32 within each 32-lane half of the wave (MI355X_MICROARCH.md, LDS).  A thread that
// loaded four CONSECUTIVE k of one row (K-contiguous operands: pre-activation gradients, plain weights) writes rows k,
// k+1, ..; the four threads that share a row write rows 4 apart - 320 floats apart at this pitch, i.e. the SAME bank -
// and a 32-lane half holds 8 rows x 4 such threads: XOR-ing 8 * (k quad) into the column spreads them over the 32 banks
// (round 2 first used 16 * (k quad): exact for 64 banks, still two-way for the 32 of a b32 access - PMC showed 30-47 % of
// the LDS-active cycles as conflicts).  The fragment reads (k = 4 s + kq: the term is 8 * (s & 3), wave-uniform; kq = 0 / 1
// of a half differ by one row = 16 banks) stay conflict-free.
#define BG_SWZ(k) ((((k) >> 2) & 3) << 3)
#ifndef BG_KH
#define BG_KH 1    // 16-wide K halves per staged tile (2 was measured: no gain at the backward's shapes, twice the LDS)
#endif
#define BG_KT (16 * BG_KH)

// ROLE only names the instantiation, so that profilers list the call sites of the backward on separate lines
enum { BG_GENERIC = 0, BG_CHAIN_DENSE, BG_CHAIN_NODE, BG_CHAIN_MIX, BG_X_NODE, BG_X_MIX, BG_WGRAD, BG_ADJ, BG_LINEAR,
       BG_POOL, BG_HEAD, BG_ROLES };
template <int ROLE>
__global__ __launch_bounds__(256) void k_bgemm(GemmArgs g) {
  __shared__ float As[2][BG_KT][BG_LD];
  __shared__ float Bs[2][BG_KT][BG_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  int z = blockIdx.z;
  const int part = z % g.split;
  z /= g.split;
  const int b2 = z % g.nb2, b1 = z / g.nb2;
  const float* A = g.A + (g.nOff ? (size_t)g.offA[b1] : (size_t)b1 * g.bA1) + (size_t)b2 * g.bA2;
  const float* B = g.B + (g.nOff ? (size_t)g.offB[b1] : (size_t)b1 * g.bB1) + (size_t)b2 * g.bB2;
  float* C = g.C + (g.nOff ? (size_t)g.offC[b1] : (size_t)b1 * g.bC1) + (size_t)b2 * g.bC2;
  // element of the 64 x 16 A tile (16 x 64 B tile) this thread loads in sweep i: the unit-stride axis runs fastest.
  // When pointer and strides are 16-byte friendly the whole tile is ONE float4 per thread along the unit-stride axis
  // (vecA / vecB, wave-uniform); ragged edges fall back to scalar loads per element of the quad.
  const bool aK = g.sAk == 1, bN = g.sBn == 1;
  const bool aM = g.sAm == 1, bK = g.sBk == 1;
  auto aligned4 = [](const float* p, long s0, long s1, long s2) {
    return ((reinterpret_cast<size_t>(p) & 15) == 0) && (s0 & 3) == 0 && (s1 & 3) == 0 && (s2 & 3) == 0;
  };
  const bool vecA = (aK && aligned4(A, g.sAm, g.sAk2, 0)) || (aM && !aK && aligned4(A, g.sAk, g.sAk2, 0));
  const bool vecB = (bN && aligned4(B, g.sBk, g.sBk2, 0)) || (bK && !bN && aligned4(B, g.sBn, g.sBk2, 0));
  int am[4], ak[4], bk[4], bn[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (vecA) {
      if (aK) { am[i] = tid >> 2; ak[i] = (tid & 3) * 4 + i; } else { ak[i] = tid >> 4; am[i] = (tid & 15) * 4 + i; }
    } else if (aK) { ak[i] = tid & 15; am[i] = (tid >> 4) + 16 * i; } else { am[i] = tid & 63; ak[i] = (tid >> 6) + 4 * i; }
    if (vecB) {
      if (bN) { bk[i] = tid >> 4; bn[i] = (tid & 15) * 4 + i; } else { bn[i] = tid >> 2; bk[i] = (tid & 3) * 4 + i; }
    } else if (bN) { bn[i] = tid & 63; bk[i] = (tid >> 6) + 4 * i; } else { bk[i] = tid & 15; bn[i] = (tid >> 4) + 16 * i; }
  }
  const int kTiles = (g.K + BG_KT - 1) / BG_KT;
  const int total = g.K2 * kTiles;                     // K tiles over (k2, k): far below 2^31 (checked by the launcher)
  const int per = (total + g.split - 1) / g.split;
  const int tBeg = part * per, tEnd = tBeg + per < total ? tBeg + per : total;
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // two register sets: K-tiles t+1 and t+2 are in flight while tile t is multiplied (one tile ahead - 16 MFMAs per wave -
  // cannot cover a memory round trip; the loop is unrolled by two so that the sets stay named registers)
  float ra[2][BG_KH][4], rb[2][BG_KH][4];
  int fk2 = tBeg / kTiles, fkt = tBeg - fk2 * kTiles;   // (k2, k tile) of the next fetch: advanced, never divided again
  auto fetch = [&](float (&fa)[BG_KH][4], float (&fb)[BG_KH][4]) __attribute__((always_inline)) {
    const int k2 = fk2, kBase = fkt * BG_KT;
    if (++fkt == kTiles) { fkt = 0; ++fk2; }
    const float* Ap = A + (size_t)k2 * g.sAk2;
    const float* Bp = B + (size_t)k2 * g.sBk2;
#pragma unroll
    for (int h = 0; h < BG_KH; ++h) {
      const int k0 = kBase + 16 * h;
      // the quad of this thread is contiguous in memory along its unit-stride axis: elements 0 and 3 bound it
      const bool fullA = vecA && m0 + am[3] < g.M && k0 + ak[3] < g.K;
      if (fullA) {
        const float4 v = *reinterpret_cast<const float4*>(Ap + (size_t)(m0 + am[0]) * g.sAm + (size_t)(k0 + ak[0]) * g.sAk);
        fa[h][0] = v.x; fa[h][1] = v.y; fa[h][2] = v.z; fa[h][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + am[i], k = k0 + ak[i];
          fa[h][i] = (m < g.M && k < g.K) ? Ap[(size_t)m * g.sAm + (size_t)k * g.sAk] : 0.f;
        }
      }
      const bool fullB = vecB && n0 + bn[3] < g.N && k0 + bk[3] < g.K;
      if (fullB) {
        const float4 v = *reinterpret_cast<const float4*>(Bp + (size_t)(k0 + bk[0]) * g.sBk + (size_t)(n0 + bn[0]) * g.sBn);
        fb[h][0] = v.x; fb[h][1] = v.y; fb[h][2] = v.z; fb[h][3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = n0 + bn[i], kb = k0 + bk[i];
          fb[h][i] = (n < g.N && kb < g.K) ? Bp[(size_t)kb * g.sBk + (size_t)n * g.sBn] : 0.f;
        }
      }
    }
  };
  // (operands that are unit-stride along M / N hold four consecutive columns of ONE k row: a single 16-byte store)
  const bool rowA = vecA && !aK, rowB = vecB && bN;
  auto stash = [&](int buf, const float (&fa)[BG_KH][4], const float (&fb)[BG_KH][4]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < BG_KH; ++h) {
      if (rowA) {
        *reinterpret_cast<float4*>(&As[buf][16 * h + ak[0]][am[0] ^ BG_SWZ(ak[0])]) =
            make_float4(fa[h][0], fa[h][1], fa[h][2], fa[h][3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) As[buf][16 * h + ak[i]][am[i] ^ BG_SWZ(ak[i])] = fa[h][i];
      }
      if (rowB) {
        *reinterpret_cast<float4*>(&Bs[buf][16 * h + bk[0]][bn[0] ^ BG_SWZ(bk[0])]) =
            make_float4(fb[h][0], fb[h][1], fb[h][2], fb[h][3]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) Bs[buf][16 * h + bk[i]][bn[i] ^ BG_SWZ(bk[i])] = fb[h][i];
      }
    }
  };
  auto mma = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4 * BG_KH; ++s) {
      const int k = 4 * s + kq;
      float av[2], bv[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) av[a] = As[buf][k][(wm * 32 + a * 16 + j) ^ ((s & 3) << 3)];
#pragma unroll
      for (int b = 0; b < 2; ++b) bv[b] = Bs[buf][k][(wn * 32 + b * 16 + j) ^ ((s & 3) << 3)];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = MFMA16(av[a], bv[b], acc[a][b]);
    }
  };
  if (tBeg < tEnd) {
    fetch(ra[0], rb[0]);                               // tile tBeg
    stash(0, ra[0], rb[0]);
    if (tBeg + 1 < tEnd) fetch(ra[1], rb[1]);          // tile tBeg + 1
    __syncthreads();
    for (int tile = tBeg; tile < tEnd; tile += 2) {
      if (tile + 2 < tEnd) fetch(ra[0], rb[0]);        // tile + 2
      mma(0);
      if (tile + 1 < tEnd) stash(1, ra[1], rb[1]);     // tile + 1
      __syncthreads();
      if (tile + 1 < tEnd) {
        if (tile + 3 < tEnd) fetch(ra[1], rb[1]);      // tile + 3
        mma(1);
        if (tile + 2 < tEnd) stash(0, ra[0], rb[0]);   // tile + 2
        __syncthreads();
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 32 + a * 16 + 4 * kq + e, n = n0 + wn * 32 + b * 16 + j;
        if (m >= g.M || n >= g.N) continue;
        float* dst = C + (size_t)m * g.sCm + (size_t)n * g.sCn;
        float v = g.alpha * acc[a][b][e];
        if (g.scaleC) v *= g.scaleC[(size_t)b1 * g.bS1 + (size_t)b2 * g.bS2 + (size_t)m * g.sSm + (size_t)n * g.sSn];
        if (g.mode == 1) unsafeAtomicAdd(dst, v);
        else *dst = (g.beta != 0.f) ? v + g.beta * *dst : v;
      }
}

// ---- the same GEMM, fast path for operands that are unit-stride along M (A) and along N (B) --------------------------
// Conditions (checked by the launcher, tn_eligible): sAm == 1, sBn == 1, M and N multiples of 64, every other stride and
// both base pointers multiples of 4 floats.  That is the graph-mix kernel's operand layout - a K-tile of A is 16 rows
// of 256 contiguous bytes - so it gets the graph-mix kernel's pipeline instead of the generic one: K-tiles wait in
// registers TWO tiles ahead of the MFMAs (the generic kernel runs one ahead: 16 MFMAs per wave cannot cover a memory
// round trip), one float4 per thread and operand, 16-byte LDS stores into rows rotated by 16 * (k & 3) floats
// (conflict-free fragment reads), loads unconditional (row index clamped, rows past the end of K zeroed by a select).
// Call sites: the node-adaptive weight gradients (K = T*B rows per node and slot) and the residual nn.Linear
// gradients (K = every (t, b, n) row, split over workgroups with atomics) - 6.8 of the backward's 28.7 ms of kernels.
template <int ROLE>
__global__ __launch_bounds__(256) void k_bgemm_tn(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][16 * 64];
  __shared__ __attribute__((aligned(16))) float Bs[2][16 * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const int wr = w >> 1, wc = w & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  int z = blockIdx.z;
  const int part = z % g.split;
  z /= g.split;
  const int b2 = z % g.nb2, b1 = z / g.nb2;
  const int kk = tid >> 4, sg = tid & 15;
  const float* A = g.A + (size_t)b1 * g.bA1 + (size_t)b2 * g.bA2 + m0 + sg * 4;
  const float* B = g.B + (size_t)b1 * g.bB1 + (size_t)b2 * g.bB2 + n0 + sg * 4;
  float* C = g.C + (size_t)b1 * g.bC1 + (size_t)b2 * g.bC2;
  const int kTiles = (g.K + 15) >> 4;
  const int total = g.K2 * kTiles;
  const int per = (total + g.split - 1) / g.split;
  const int tBeg = part * per, tEnd = tBeg + per < total ? tBeg + per : total;
  const int stPos = kk * 64 + (((sg + 4 * (kk & 3)) & 15) << 2);
  // the fetch stream walks the (k2, k-tile) pairs of this part in order; past the end it re-reads the last tile, zeroed
  int fk2 = tBeg / kTiles, fkt = tBeg - fk2 * kTiles, fetched = tBeg;
  const int lastK2 = (tEnd - 1) / kTiles, lastKt = (tEnd - 1) - lastK2 * kTiles;
  float4 asum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](float4& ra, float4& rb) {
    const bool live = fetched < tEnd;
    const int k2 = live ? fk2 : lastK2, kt = live ? fkt : lastKt;
    ++fetched;
    if (++fkt == kTiles) { fkt = 0; ++fk2; }
    const int k = kt * 16 + kk, kc = min(k, g.K - 1);
    const float4 va = *reinterpret_cast<const float4*>(A + (size_t)k2 * g.sAk2 + (size_t)kc * g.sAk);
    const float4 vb = *reinterpret_cast<const float4*>(B + (size_t)k2 * g.sBk2 + (size_t)kc * g.sBk);
    const bool ok = live && k < g.K;
    ra = make_float4(ok ? va.x : 0.f, ok ? va.y : 0.f, ok ? va.z : 0.f, ok ? va.w : 0.f);
    rb = make_float4(ok ? vb.x : 0.f, ok ? vb.y : 0.f, ok ? vb.z : 0.f, ok ? vb.w : 0.f);
    asum = make_float4(asum.x + ra.x, asum.y + ra.y, asum.z + ra.z, asum.w + ra.w);   // every row of A passes here once
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tBeg < tEnd) {
    float4 ra0, rb0, ra1, rb1;
    fetch(ra0, rb0);
    *reinterpret_cast<float4*>(&As[0][stPos]) = ra0;
    *reinterpret_cast<float4*>(&Bs[0][stPos]) = rb0;
    fetch(ra0, rb0);
    fetch(ra1, rb1);
    __syncthreads();
    const int rotA0 = (wr * 32 + j + 16 * kq) & 63, rotA1 = (wr * 32 + 16 + j + 16 * kq) & 63;
    const int rotB0 = (wc * 32 + j + 16 * kq) & 63, rotB1 = (wc * 32 + 16 + j + 16 * kq) & 63;
    auto mma = [&](int cur) {
      const float* At = &As[cur][kq * 64];
      const float* Bt = &Bs[cur][kq * 64];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float a0 = At[s * 256 + rotA0], a1 = At[s * 256 + rotA1];
        const float b0 = Bt[s * 256 + rotB0], b1 = Bt[s * 256 + rotB1];
        acc[0][0] = MFMA16(a0, b0, acc[0][0]);
        acc[0][1] = MFMA16(a0, b1, acc[0][1]);
        acc[1][0] = MFMA16(a1, b0, acc[1][0]);
        acc[1][1] = MFMA16(a1, b1, acc[1][1]);
      }
    };
    const int nT = tEnd - tBeg;
    for (int it = 0; it < nT; it += 2) {
      mma(0);
      *reinterpret_cast<float4*>(&As[1][stPos]) = ra0;
      *reinterpret_cast<float4*>(&Bs[1][stPos]) = rb0;
      fetch(ra0, rb0);
      __syncthreads();
      if (it + 1 < nT) {
        mma(1);
        *reinterpret_cast<float4*>(&As[0][stPos]) = ra1;
        *reinterpret_cast<float4*>(&Bs[0][stPos]) = rb1;
        fetch(ra1, rb1);
        __syncthreads();
      }
    }
    if (g.colsumA && blockIdx.x == 0) {   // the 16 threads that hold the same 4 columns (one per k row of a tile) meet in LDS
      *reinterpret_cast<float4*>(&As[0][kk * 64 + sg * 4]) = asum;
      __syncthreads();
      if (tid < 64) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) v += As[0][q * 64 + tid];
        unsafeAtomicAdd(g.colsumA + m0 + tid, v);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wr * 32 + p * 16 + 4 * kq + e, n = n0 + wc * 32 + q * 16 + j;
        float* dst = C + (size_t)m * g.sCm + (size_t)n * g.sCn;
        const float v = g.alpha * acc[p][q][e];
        if (g.mode == 1) unsafeAtomicAdd(dst, v);
        else *dst = (g.beta != 0.f) ? v + g.beta * *dst : v;
      }
}

// ---- plain (row-major) folded node-adaptive weights for the backward GEMMs ------------------------------------
// Wp[n][s][i][o] = the weights the forward node kernels contract with, slot s = 0 (identity, with the diagonal
// supports folded in) or a dense slot: g_k * sum_d E[n][d] Wpool[d][k][i][o]  (MultiATGCN.py:102-105; StackMap)
struct PlainPrep {
  const float* E;
  const float* wpool;
  const float* wg;       // weights_g or null
  float* out;            // [N][S][I][O]
  int d, I, O, N, S;
  StackMap map;
};

__device__ __forceinline__ float stack_gain(const float* wg, int Kt, int k) {
  if (!wg) return 1.f;
  float mx = -3.0e38f, sum = 0.f;
  for (int q = 0; q < Kt; ++q) mx = fmaxf(mx, wg[q]);
  for (int q = 0; q < Kt; ++q) sum += expf(wg[q] - mx);
  return expf(wg[k] - mx) / sum;
}

// One thread per (slot, i, o) and PP_NODES consecutive nodes: the d pool values of its weight entry are read ONCE into
// registers and serve all of them (one thread per node re-read them from L2 for every node: 2 GB of L2 traffic per call,
// 205 us for a 105 MB result; the node embeddings are uniform per workgroup: scalar loads).  d > PP_MAXD takes the
// node-at-a-time path.
#define PP_NODES 8
#define PP_MAXD 32
__global__ __launch_bounds__(256) void k_prep_plain(PlainPrep a) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t per = (size_t)a.S * a.I * a.O;
  if (idx >= per) return;
  const int o = idx % a.O, i = (idx / a.O) % a.I, s = idx / ((size_t)a.O * a.I);
  const size_t dstride = (size_t)a.map.KtotOrig * a.I * a.O;
  const int n0 = blockIdx.y * PP_NODES, nEnd = min(a.N, n0 + PP_NODES);
  float acc[PP_NODES];
#pragma unroll
  for (int q = 0; q < PP_NODES; ++q) acc[q] = 0.f;
  const int terms = s == 0 ? 1 + a.map.nDiag : 1;     // the identity slot also carries the folded diagonal supports
  for (int term = 0; term < terms; ++term) {
    const int k = term == 0 ? a.map.keepK[s] : a.map.diagK[term - 1];
    const float gain = stack_gain(a.wg, a.map.KtotOrig, k);
    const float* wp = a.wpool + ((size_t)k * a.I + i) * a.O + o;
    if (a.d <= PP_MAXD) {
      float p[PP_MAXD];
#pragma unroll
      for (int dd = 0; dd < PP_MAXD; ++dd) p[dd] = dd < a.d ? wp[dd * dstride] : 0.f;
#pragma unroll
      for (int q = 0; q < PP_NODES; ++q) {
        const int n = n0 + q;
        if (n >= nEnd) break;
        const float* e = a.E + (size_t)n * a.d;
        float part = 0.f;
#pragma unroll
        for (int dd = 0; dd < PP_MAXD; ++dd)
          if (dd < a.d) part = fmaf(e[dd], p[dd], part);
        const float f = term == 0 ? 1.f
                                  : cheb_scalar(a.map.diagSrc[term - 1][(size_t)n * (a.map.N + 1)], a.map.diagOrder[term - 1]);
        acc[q] = fmaf(f * gain, part, acc[q]);
      }
    } else {
      for (int q = 0; q < PP_NODES && n0 + q < nEnd; ++q) {
        const int n = n0 + q;
        const float* e = a.E + (size_t)n * a.d;
        float part = 0.f;
        for (int dd = 0; dd < a.d; ++dd) part = fmaf(e[dd], wp[dd * dstride], part);
        const float f = term == 0 ? 1.f
                                  : cheb_scalar(a.map.diagSrc[term - 1][(size_t)n * (a.map.N + 1)], a.map.diagOrder[term - 1]);
        acc[q] = fmaf(f * gain, part, acc[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < PP_NODES; ++q)
    if (n0 + q < nEnd) a.out[(size_t)(n0 + q) * per + idx] = acc[q];
}

// EK[e][n][d] = g_k * f_e[n] * E[n][d] for stack entry e with pool index k: f = 1 for kept slots, the Chebyshev value of
// the diagonal for folded ones;  FK[e][n] = f_e[n]  (left operands of the pool-gradient GEMMs)
__global__ __launch_bounds__(256) void k_scaled_emb(const float* __restrict__ E, const float* __restrict__ wg,
                                                    StackMap map, StackEntries ent, int d, float* __restrict__ EK,
                                                    float* __restrict__ FK) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int e = blockIdx.y;
  if (idx >= map.N * d) return;
  const int n = idx / d;
  const int q = ent.diag[e];
  const float f = q >= 0 ? cheb_scalar(map.diagSrc[q][(size_t)n * (map.N + 1)], map.diagOrder[q]) : 1.f;
  EK[(size_t)e * map.N * d + idx] = stack_gain(wg, map.KtotOrig, ent.pool[e]) * f * E[idx];
  if (idx % d == 0) FK[(size_t)e * map.N + n] = f;
}

// The same plain weights on the matrix cores (round 3; the embedding contraction of k_prep_mfma with another choice of
// the 16 A rows): a quad is 16 consecutive output columns o of ONE weight row (s, i), so the accumulator of a lane - rows
// 4 (l >> 4) + e, column l & 15 - is a float4 of four consecutive o for node l & 15, stored straight into [n][s][i][o].
// A wave keeps 8 quads (128 consecutive elements of the [S*I][O] matrix) and walks 16-node tiles.  The stack gain is
// folded into the A values as the forward's weight streams have it, so both hold the same numbers.
template <int DS>
__global__ __launch_bounds__(256) void k_prep_plain_mfma(PlainPrep a, int tilesPerBlock) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, ka = lane >> 4;
  const size_t per = (size_t)a.S * a.I * a.O;
  const size_t base = (size_t)blockIdx.x * 128;
  const size_t dstride = (size_t)a.map.KtotOrig * a.I * a.O;
  float A[8][DS], Aq[8][DS];
  bool ident = false;
#pragma unroll
  for (int q8 = 0; q8 < 8; ++q8) {
    const size_t el = min(base + 16 * q8 + r, per - 1);
    const int o = (int)(el % a.O), row = (int)(el / a.O), i = row % a.I, sl = row / a.I;
    const int k = a.map.keepK[sl];
    const float gain = stack_gain(a.wg, a.map.KtotOrig, k);
    const float* src = a.wpool + ((size_t)k * a.I + i) * a.O + o;
    const bool id = sl == 0 && a.map.nDiag > 0;
    ident = ident || id;
    const float* srcq = a.wpool + ((size_t)a.map.diagK[0] * a.I + i) * a.O + o;
#pragma unroll
    for (int st = 0; st < DS; ++st) {
      const int dd = 4 * st + ka;
      const float v = src[(size_t)min(dd, a.d - 1) * dstride];
      A[q8][st] = dd < a.d ? v * gain : 0.f;
      const float vq = id ? srcq[(size_t)min(dd, a.d - 1) * dstride] : 0.f;
      Aq[q8][st] = (dd < a.d && id) ? vq : 0.f;
    }
  }
  const bool anyIdent = __ballot(ident) != 0ull;
  const int nTiles = (a.N + 15) >> 4;
  const int tile0 = blockIdx.y * tilesPerBlock, tile1 = min(tile0 + tilesPerBlock, nTiles);
  const int nodeL = lane & 15, kb = lane >> 4;
  auto load_b = [&](int tile, float (&Bf)[DS]) {
    const float* e = a.E + (size_t)min(tile * 16 + nodeL, a.N - 1) * a.d;
#pragma unroll
    for (int st = 0; st < DS; ++st) {
      const int dd = 4 * st + kb;
      const float v = e[min(dd, a.d - 1)];
      Bf[st] = dd < a.d ? v : 0.f;
    }
  };
  float Bf[DS], Bn[DS];
  if (tile0 + w < tile1) load_b(tile0 + w, Bf);
  for (int tile = tile0 + w; tile < tile1; tile += 4) {
    load_b(min(tile + 4, tile1 - 1), Bn);
    const int n = tile * 16 + nodeL;
    f32x4 acc[8];
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8) acc[q8] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < DS; ++st)
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8) acc[q8] = MFMA16(A[q8][st], Bf[st], acc[q8]);
    if (anyIdent) {
      const int nc = min(n, a.N - 1);
      for (int q = 0; q < a.map.nDiag; ++q) {
        f32x4 part[8];
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) part[q8] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q == 0) {
#pragma unroll
          for (int st = 0; st < DS; ++st)
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) part[q8] = MFMA16(Aq[q8][st], Bf[st], part[q8]);
        } else {   // further diagonal supports: their pool values are re-read (rare)
#pragma unroll
          for (int q8 = 0; q8 < 8; ++q8) {
            const size_t el = min(base + 16 * q8 + r, per - 1);
            const int o = (int)(el % a.O), row = (int)(el / a.O), i = row % a.I, sl = row / a.I;
#pragma unroll
            for (int st = 0; st < DS; ++st) {
              const int dd = 4 * st + ka;
              const float v = a.wpool[(size_t)min(dd, a.d - 1) * dstride + ((size_t)a.map.diagK[q] * a.I + i) * a.O + o];
              part[q8] = MFMA16((dd < a.d && sl == 0) ? v : 0.f, Bf[st], part[q8]);
            }
          }
        }
        const float t = stack_gain(a.wg, a.map.KtotOrig, a.map.diagK[q]) *
                        cheb_scalar(a.map.diagSrc[q][(size_t)nc * (a.map.N + 1)], a.map.diagOrder[q]);
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[q8][e] = fmaf(t, part[q8][e], acc[q8][e]);
      }
    }
    if (n < a.N) {
      float* dst = a.out + (size_t)n * per + base + 4 * kb;
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8)
        if (base + 16 * q8 + 4 * kb < per)
          *reinterpret_cast<float4*>(dst + 16 * q8) = make_float4(acc[q8][0], acc[q8][1], acc[q8][2], acc[q8][3]);
    }
#pragma unroll
    for (int st = 0; st < DS; ++st) Bf[st] = Bn[st];
  }
}

// ---- the two pool-gradient products on the matrix cores (round 3) ---------------------------------------------------
// Both have the embedding dimension d (<= 32) as one side of the product: on 64 x 64 GEMM tiles (k_bgemm<BG_POOL>) two
// thirds of every tile were padding and the plain weight gradients dWp (250 MB per backward) streamed through the scalar
// path of the generic kernel.  The 16x16x4 fp32 MFMA takes d as 2 tiles of 16 and reads dWp as whole float4s:
//
// (1) dWpool[dd][pool(e)][io] = sum_n EK[e][n][dd] * dWp[n][slot(e)][io]           (transpose of k_prep_mfma)
//     A = EK^T (rows dd, reduction n), B = dWp (reduction n, columns io).  A lane's float4 of a node's row holds the
//     SAME column of four column tiles (column 4 (l & 15) + ct of the wave's 64), so one 256-byte row piece per node
//     feeds eight MFMAs, and the accumulators leave as float4 stores of four consecutive columns.
//     One wave per (entry, 64 columns); entries with distinct pool indices only (cheb_order = 1 keeps the GEMMs).
__global__ __launch_bounds__(256) void k_pool_grad_mfma(const float* __restrict__ EK, const float* __restrict__ dWp,
                                                        StackEntries ent, int N, int d, long IO, int S, int Kt,
                                                        float* __restrict__ dPool) {
  const int lane = threadIdx.x & 63, kq = lane >> 4, j = lane & 15;
  const int e = blockIdx.y;
  const long col0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (col0 >= IO) return;
  const float* ek = EK + (size_t)e * N * d;
  const float* src = dWp + (size_t)ent.slot[e] * IO + col0 + 4 * j;
  const size_t nodeStride = (size_t)S * IO;
  f32x4 acc[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[r][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int steps = (N + 3) >> 2;
  auto load = [&](int st, float4& b, float& a0, float& a1) {
    const int node = 4 * st + kq, nc = min(node, N - 1);
    b = *reinterpret_cast<const float4*>(src + (size_t)nc * nodeStride);
    const float v0 = ek[(size_t)nc * d + min(j, d - 1)], v1 = ek[(size_t)nc * d + min(16 + j, d - 1)];
    a0 = (node < N && j < d) ? v0 : 0.f;
    a1 = (node < N && 16 + j < d) ? v1 : 0.f;
  };
  float4 b[4];
  float a0[4], a1[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) load(min(u, steps - 1), b[u], a0[u], a1[u]);
  for (int st = 0; st < steps; st += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 bv = b[u];
      const float x0 = st + u < steps ? a0[u] : 0.f, x1 = st + u < steps ? a1[u] : 0.f;
      load(min(st + u + 4, steps - 1), b[u], a0[u], a1[u]);
      acc[0][0] = MFMA16(x0, bv.x, acc[0][0]); acc[1][0] = MFMA16(x1, bv.x, acc[1][0]);
      acc[0][1] = MFMA16(x0, bv.y, acc[0][1]); acc[1][1] = MFMA16(x1, bv.y, acc[1][1]);
      acc[0][2] = MFMA16(x0, bv.z, acc[0][2]); acc[1][2] = MFMA16(x1, bv.z, acc[1][2]);
      acc[0][3] = MFMA16(x0, bv.w, acc[0][3]); acc[1][3] = MFMA16(x1, bv.w, acc[1][3]);
    }
  }
  float* dst = dPool + (size_t)ent.pool[e] * IO + col0 + 4 * j;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int dd = 16 * r + 4 * kq + q;
      if (dd < d)
        *reinterpret_cast<float4*>(dst + (size_t)dd * Kt * IO) =
            make_float4(acc[r][0][q], acc[r][1][q], acc[r][2][q], acc[r][3][q]);
    }
}

// (2) TmpK[e][n][dd] += sum_io dWp[n][slot(e)][io] * Wpool[dd][pool(e)][io]
//     A = dWp (rows n, reduction io), B = Wpool^T: both lie K-contiguous, so a lane's float4 is four MFMA steps of its
//     row (the node kernels' operand trick).  One wave per (entry, 16 nodes, 1/splits of the io range), fp32 atomics.
__global__ __launch_bounds__(256) void k_pool_emb_mfma(const float* __restrict__ dWp, const float* __restrict__ wpool,
                                                       StackEntries ent, int N, int d, long IO, int S, int Kt, int splits,
                                                       float* __restrict__ TmpK) {
  const int lane = threadIdx.x & 63, kq = lane >> 4, j = lane & 15;
  const int e = blockIdx.y;
  const int tile = blockIdx.x / splits, part = (blockIdx.x - tile * splits) * 4 + (threadIdx.x >> 6), parts = splits * 4;
  const long groups = IO >> 4, per = (groups + parts - 1) / parts;
  const long g0 = part * per, g1 = min(g0 + per, groups);
  const int node = min(tile * 16 + j, N - 1);
  const float* ap = dWp + (size_t)node * S * IO + (size_t)ent.slot[e] * IO + 4 * kq;
  const float* bp0 = wpool + ((size_t)min(j, d - 1) * Kt + ent.pool[e]) * IO + 4 * kq;
  const float* bp1 = wpool + ((size_t)min(16 + j, d - 1) * Kt + ent.pool[e]) * IO + 4 * kq;
  const bool ok0 = j < d, ok1 = 16 + j < d;
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  for (long g = g0; g < g1; g += 2) {
    const long gb = min(g + 1, g1 - 1);
    const float4 a0 = *reinterpret_cast<const float4*>(ap + g * 16), a1 = *reinterpret_cast<const float4*>(ap + gb * 16);
    float4 p0 = *reinterpret_cast<const float4*>(bp0 + g * 16), p1 = *reinterpret_cast<const float4*>(bp1 + g * 16);
    float4 q0 = *reinterpret_cast<const float4*>(bp0 + gb * 16), q1 = *reinterpret_cast<const float4*>(bp1 + gb * 16);
    if (!ok0) { p0 = make_float4(0.f, 0.f, 0.f, 0.f); q0 = p0; }
    if (!ok1) { p1 = make_float4(0.f, 0.f, 0.f, 0.f); q1 = p1; }
    if (g + 1 >= g1) { q0 = make_float4(0.f, 0.f, 0.f, 0.f); q1 = q0; }
    acc[0] = MFMA16(a0.x, p0.x, acc[0]); acc[1] = MFMA16(a0.x, p1.x, acc[1]);
    acc[0] = MFMA16(a0.y, p0.y, acc[0]); acc[1] = MFMA16(a0.y, p1.y, acc[1]);
    acc[0] = MFMA16(a0.z, p0.z, acc[0]); acc[1] = MFMA16(a0.z, p1.z, acc[1]);
    acc[0] = MFMA16(a0.w, p0.w, acc[0]); acc[1] = MFMA16(a0.w, p1.w, acc[1]);
    acc[0] = MFMA16(a1.x, q0.x, acc[0]); acc[1] = MFMA16(a1.x, q1.x, acc[1]);
    acc[0] = MFMA16(a1.y, q0.y, acc[0]); acc[1] = MFMA16(a1.y, q1.y, acc[1]);
    acc[0] = MFMA16(a1.z, q0.z, acc[0]); acc[1] = MFMA16(a1.z, q1.z, acc[1]);
    acc[0] = MFMA16(a1.w, q0.w, acc[0]); acc[1] = MFMA16(a1.w, q1.w, acc[1]);
  }
  if (g0 >= g1) return;
  float* dst = TmpK + (size_t)e * N * d;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = tile * 16 + 4 * kq + q, dd = 16 * r + j;
      if (n < N && dd < d) unsafeAtomicAdd(dst + (size_t)n * d + dd, acc[r][q]);
    }
}

// bias = E . bias_pool (MultiATGCN.py:105), both gradients in one launch (two 64 x 64-tile GEMMs of M = d resp. N = d
// took 70-120 us per AGCN):  dBpool[dd][o] = sum_n E[n][dd] dBias[n][o]  (blocks 0 .. d-1: block dd, thread o, four
// partial sums over the nodes);  dE[n][dd] += sum_o dBias[n][o] bpool[dd][o]  (the remaining blocks: one thread per (n, dd))
__global__ __launch_bounds__(256) void k_bias_pool_grad(const float* __restrict__ E, const float* __restrict__ dBias,
                                                        const float* __restrict__ bpool, int N, int d, int O,
                                                        float* __restrict__ dBpool, float* __restrict__ dE) {
  __shared__ float red[256];
  if ((int)blockIdx.x < d) {
    const int dd = blockIdx.x, o = threadIdx.x % O, part = threadIdx.x / O, parts = 256 / O;
    float s = 0.f;
    for (int n = part; n < N; n += parts) s = fmaf(E[(size_t)n * d + dd], dBias[(size_t)n * O + o], s);
    red[threadIdx.x] = s;
    __syncthreads();
    if (part == 0) {
      for (int q = 1; q < parts; ++q) s += red[q * O + o];
      dBpool[(size_t)dd * O + o] = s;
    }
    return;
  }
  if (!dE) return;
  const int idx = (blockIdx.x - d) * 256 + threadIdx.x;
  if (idx >= N * d) return;
  const int n = idx / d, dd = idx - n * d;
  const float4* a = reinterpret_cast<const float4*>(dBias + (size_t)n * O);
  const float4* b = reinterpret_cast<const float4*>(bpool + (size_t)dd * O);
  float s = 0.f;
  for (int q = 0; q < O / 4; ++q) {
    const float4 x = a[q], y = b[q];
    s = fmaf(x.x, y.x, s); s = fmaf(x.y, y.y, s); s = fmaf(x.z, y.z, s); s = fmaf(x.w, y.w, s);
  }
  dE[idx] += s;
}

// dE[n][d] += sum_e g_k(e) f_e[n] TmpK[e][n][d];   dgain[k(e)] += sum_{n,d} f_e[n] E[n][d] TmpK[e][n][d]
__global__ __launch_bounds__(256) void k_emb_grad(const float* __restrict__ TmpK, const float* __restrict__ FK,
                                                  const float* __restrict__ E, const float* __restrict__ wg, int Kt,
                                                  StackEntries ent, int N, int d, float* __restrict__ dE,
                                                  float* __restrict__ dgain) {
  __shared__ float red[256];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int e = blockIdx.y, k = ent.pool[e];
  float part = 0.f;
  if (idx < N * d) {
    const int n = idx / d;
    const float f = FK[(size_t)e * N + n], v = TmpK[(size_t)e * N * d + idx];
    if (dE) unsafeAtomicAdd(&dE[idx], stack_gain(wg, Kt, k) * f * v);
    part = f * E[idx] * v;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && dgain) unsafeAtomicAdd(&dgain[k], red[0]);
}

// softmax backward of a short vector: dw[k] = g_k * (dg[k] - sum_j g_j dg[j])   (weights_g, weight_tsg)
__global__ void k_softmax_bwd_small(const float* __restrict__ w, const float* __restrict__ dg, int K,
                                    float* __restrict__ dw) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float dot = 0.f;
  for (int k = 0; k < K; ++k) dot += stack_gain(w, K, k) * dg[k];
  for (int k = 0; k < K; ++k) dw[k] = stack_gain(w, K, k) * (dg[k] - dot);
}

// ---- adaptive adjacency backward (MultiATGCN.py:80-83): dA (N,N) -> gradient of the pre-relu logits ------------
// one workgroup per row n: p = softmax(relu(l)); dl[m] = p[m] * (dA[n][m] - sum_j dA[n][j] p[j]) * (l[m] > 0)
__global__ __launch_bounds__(256) void k_adaptive_adj_bwd(const float* __restrict__ e1, const float* __restrict__ e2,
                                                          int rank, int bidir, int N, const float* __restrict__ dA,
                                                          float* __restrict__ dL) {
  __shared__ float red[256];
  __shared__ float erow[64];
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int r = tid; r < rank; r += 256) erow[r] = e1[(size_t)n * rank + r];
  __syncthreads();
  auto raw = [&](int m) {
    float s = 0.f;
    if (bidir) { for (int r = 0; r < rank; ++r) s = fmaf(erow[r], e1[(size_t)m * rank + r], s); }
    else { for (int r = 0; r < rank; ++r) s = fmaf(erow[r], e2[(size_t)r * N + m], s); }
    return s;
  };
  auto reduce = [&](float v, bool isMax) {
    red[tid] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] = isMax ? fmaxf(red[tid], red[tid + s]) : red[tid] + red[tid + s];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  float mx = 0.f;
  for (int m = tid; m < N; m += 256) mx = fmaxf(mx, fmaxf(raw(m), 0.f));
  mx = reduce(mx, true);
  float sum = 0.f;
  for (int m = tid; m < N; m += 256) sum += expf(fmaxf(raw(m), 0.f) - mx);
  const float inv = 1.0f / reduce(sum, false);
  float dot = 0.f;
  for (int m = tid; m < N; m += 256) dot += dA[(size_t)n * N + m] * expf(fmaxf(raw(m), 0.f) - mx) * inv;
  dot = reduce(dot, false);
  for (int m = tid; m < N; m += 256) {
    const float l = raw(m);
    const float p = expf(fmaxf(l, 0.f) - mx) * inv;
    dL[(size_t)n * N + m] = l > 0.f ? p * (dA[(size_t)n * N + m] - dot) : 0.f;
  }
}

// ---- element-wise pieces of the recurrent chain -------------------------------------------------------------
// All state-like tensors are [B][Np][64] slabs ("S layout"); idx runs over B*Np*64.
//
// step (l, t), part 1: blend + residual cell output algebra (MultiATGCN.py:148-150, 205-208)
//   h' = g ha + (1-g) res,  res = r2 ha + (1-r2) hc2,  ha = r h + (1-r) hc
//   in : dhp = dSeq_l[t] (+ dcarry), saved z?,r,hc,z2,r2,hc2, h = h_{t-1}
//   out: dHa (partial: blend + r2 paths), dpu2 = dhc2 * (1 - hc2^2), dr2 -> kept in DR2, blend-scalar gradient
struct ChainArgs {
  const float* dseq;     // dSeq_l[t]
  const float* dcarry;   // dh carried from step t+1 (null at t = T-1)
  const float* hprev;    // h_{t-1} (null at t = 0: zeros)
  const float *z, *r, *hc, *z2, *r2, *hc2;
  const float* blend;    // &weights_gru[l][t]
  float* dblend;         // &dweights_gru[l][t]
  float* dha;            // [B][Np][64]
  float* dpu2;           // DPU2[t]  [B][Np][64]
  float* dpg2;           // DPG2[t]  [B][Np][128]
  float* dpu;            // DPU[t]   [B][Np][64]
  float* dpg;            // DPG[t]   [B][Np][128]
  const float* dzh2;     // [B][Np][64] gradient of z2*ha  (GEMM result)
  const float* dzhA;     // dA buffer of the update AGCN [B][S][Np][64] (slot 0 = direct z*h gradient)
  const float* dzhMix;   // transposed-mix result [B][Np][64] (null when Ks = 0)
  const float* dhA;      // dA buffer of the gate AGCN
  const float* dhMix;
  float* dh;             // running dh_{t-1}: written by part 3, extended by part 4; the next step's fused kernel adds
                         // slot 0 of the gate AGCN's dA and its transposed mix to form the carry
  float* dr;             // scratch [B][Np][64]
  int mixParts;          // the transposed mixes arrive as this many partial results (split by support slot) ...
  long mixPartStride;    // ... this many floats apart: whoever reads dzhMix / dhMix / carryMix adds them up
  int B, N, Np, S;
  int dense;             // gcn_off: the layer IS a dense GRU cell on (x, h): "ha" is h_{t-1}, no blend (blend == null)
};

__device__ __forceinline__ float ha_of(const ChainArgs& a, size_t idx) {
  const float h = a.hprev ? a.hprev[idx] : 0.f;
  if (a.dense) return h;
  const float r = a.r[idx];
  return r * h + (1.f - r) * a.hc[idx];
}

__global__ __launch_bounds__(256) void k_chain_res_out(ChainArgs a) {
  __shared__ float red[256];
  const size_t total = (size_t)a.B * a.Np * 64;
  const float g = a.blend ? sigmoid_f(a.blend[0]) : 0.f;
  float part = 0.f;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int n = (idx >> 6) % a.Np;
    const float dhp = a.dseq[idx] + (a.dcarry ? a.dcarry[idx] : 0.f);
    const float ha = ha_of(a, idx), r2 = a.r2[idx], hc2 = a.hc2[idx];
    const float res = r2 * ha + (1.f - r2) * hc2;
    if (n < a.N) part += dhp * (ha - res);
    const float dres = (1.f - g) * dhp;
    a.dha[idx] = g * dhp + dres * r2;
    a.dpu2[idx] = dres * (1.f - r2) * (1.f - hc2 * hc2);
    a.dr[idx] = dres * (ha - hc2);     // dr2, consumed by k_chain_res_gate
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && a.dblend) unsafeAtomicAdd(a.dblend, red[0] * g * (1.f - g));
}

// ---- parts 1-3 (and the carry of the step before) in one kernel -------------------------------------------------
// Everything between the incoming gradient of h'_t and the pre-activation gradient of the update AGCN is local to a
// row (b, n): blend, residual-cell algebra, the two nn.Linear contractions with the h columns of the residual
// weights (shared by all rows: staged in LDS), graph-cell output algebra.  One workgroup owns 64 rows; the two
// contractions run on the 16x16x4 MFMA with the A tile (the freshly computed pre-activation gradients) in LDS.
//   incoming: dseq (+ carry of step t+1 = dh + slot 0 of the gate AGCN's dA + its transposed mix)
//   outgoing: dpu2, dpg2 (kept for the batched part), dpu (update AGCN), dr, dh (partial), blend-scalar gradient
#define CF_LD 80
struct FusedResArgs {
  ChainArgs c;
  const float* carryA;   // dA of the gate AGCN of step t+1 ([B][S][Np][64], slot 0 is read) or null
  const float* carryMix; // its transposed mix [B][Np][64] or null
  const float* ruh;      // RU + C: RU[o][C + i] at ruh[o*ldW + i]
  const float* rgh;      // RG + C
  int ldW;
};

__device__ __forceinline__ float4 ld4(const float* p, size_t idx) { return *reinterpret_cast<const float4*>(p + idx); }
// 32-bit BYTE offsets from a wave-uniform base: global_load / global_store with an SGPR base and one offset VGPR
__device__ __forceinline__ float4 ld4b(const float* p, unsigned byteOfs) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p) + byteOfs);
}
#ifndef CHAIN_STORE_WT
#define CHAIN_STORE_WT 1   // the fused chain kernels' outputs are the next chain kernel's inputs: written through (sc1) like the
#endif                     // forward's step outputs, instead of dirty L2 lines flushed at the end of the kernel
__device__ __forceinline__ void st4b(float* p, unsigned byteOfs, const float4& v) {
#if CHAIN_STORE_WT
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7ffffff0, 0x00020000);   // p is wave-uniform (a kernel argument)
  const u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, (int)byteOfs, 0, 16);            // aux 16 = sc1
#else
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(p) + byteOfs) = v;
#endif
}
__device__ __forceinline__ float get4(const float4& v, int x) { return x == 0 ? v.x : x == 1 ? v.y : x == 2 ? v.z : v.w; }

// ROWS rows per workgroup (64, or 32: twice the workgroups - the kernel streams ~120 MB per launch and 416 workgroups of
// 64 rows are 1.6 per CU, too few to keep enough loads in flight).  The staged tile is k-major; row x of reduction
// index k lives in column (x + 4 * (k >> 2)) & 63: the threads of phase 1 hold FOUR consecutive k of one row and the
// 16 threads that share a row hold k-quads 16 rows apart - without the rotation all of them write into one bank.
// (A ds_write_b32 banks by 32 within each 32-lane half, so quads q and q + 8 still meet - PMC: a third of the LDS-active
// cycles; a rotation of 2 per quad removes that and was measured SLOWER, 47.4 vs 43.4 us: the kernel is not LDS-bound.)
#define CF_ROT(k) (4 * ((k) >> 2))
template <int ROWS>
__global__ __launch_bounds__(256) void k_chain_res_fused(FusedResArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Ps = lds;                    // [128 k][CF_LD] k-major A tile (element (k, local row)); the weights (B operands,
                                      // 48 KB shared by all workgroups) come straight from L2 so that several
                                      // workgroups fit a CU
  __shared__ float red[256];
  constexpr int NP = ROWS / 32;       // 16-row sub-tiles per wave: wave (wm, wn) owns rows wm*(ROWS/2) + 16 p
  const ChainArgs& a = f.c;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const int wm = w >> 1, wn = w & 1;
  const size_t rows = (size_t)a.B * a.Np, row0 = (size_t)blockIdx.x * ROWS;
  const float g = sigmoid_f(a.blend[0]);
  // phase 1 (one thread per row and 4 columns, 16-byte accesses): everything that does not need a contraction -
  // dpu2 and the r half of dpg2 -> global + A tile; dha so far and ha*z2*(1-z2) -> row scratch; blend partial sum
  float part = 0.f;
#pragma unroll
  for (int it = 0; it < ROWS / 16; ++it) {
    const int q = tid + 256 * it, lr = q >> 4, c4 = (q & 15) * 4;
    float dpu2[4] = {0.f, 0.f, 0.f, 0.f}, drr[4] = {0.f, 0.f, 0.f, 0.f};
    if (row0 + lr < rows) {
      const size_t row = row0 + lr, idx = row * 64 + c4;
      float4 dhp = ld4(a.dseq, idx);
      if (a.dcarry) {
        const size_t b = row / a.Np, n = row - b * a.Np;
        const float4 c0 = ld4(a.dcarry, idx), c1 = ld4(f.carryA, ((b * a.S) * a.Np + n) * 64 + c4);
        dhp = make_float4(dhp.x + c0.x + c1.x, dhp.y + c0.y + c1.y, dhp.z + c0.z + c1.z, dhp.w + c0.w + c1.w);
        if (f.carryMix) {
          for (int pt = 0; pt < a.mixParts; ++pt) {
            const float4 c2 = ld4(f.carryMix + (size_t)pt * a.mixPartStride, idx);
            dhp = make_float4(dhp.x + c2.x, dhp.y + c2.y, dhp.z + c2.z, dhp.w + c2.w);
          }
        }
      }
      const float4 hp = a.hprev ? ld4(a.hprev, idx) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 r4 = ld4(a.r, idx), hc4 = ld4(a.hc, idx), r24 = ld4(a.r2, idx), z24 = ld4(a.z2, idx);
      const float4 hc24 = ld4(a.hc2, idx);
      float dha0[4], c1v[4];
      const bool real = (int)(row % a.Np) < a.N;
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float d = get4(dhp, x), h = get4(hp, x), r = get4(r4, x), hc = get4(hc4, x);
        const float r2 = get4(r24, x), z2 = get4(z24, x), hc2 = get4(hc24, x);
        const float ha = r * h + (1.f - r) * hc;
        const float res = r2 * ha + (1.f - r2) * hc2;
        if (real) part += d * (ha - res);
        const float dres = (1.f - g) * d;
        dha0[x] = g * d + dres * r2;
        dpu2[x] = dres * (1.f - r2) * (1.f - hc2 * hc2);
        drr[x] = dres * (ha - hc2) * r2 * (1.f - r2);
        c1v[x] = ha * z2 * (1.f - z2);
      }
      *reinterpret_cast<float4*>(a.dpu2 + idx) = make_float4(dpu2[0], dpu2[1], dpu2[2], dpu2[3]);
      *reinterpret_cast<float4*>(a.dpg2 + row * 128 + 64 + c4) = make_float4(drr[0], drr[1], drr[2], drr[3]);
      *reinterpret_cast<float4*>(a.dha + idx) = make_float4(dha0[0], dha0[1], dha0[2], dha0[3]);
      *reinterpret_cast<float4*>(const_cast<float*>(a.dzh2) + idx) = make_float4(c1v[0], c1v[1], c1v[2], c1v[3]);
    }
    const int pos = (lr + CF_ROT(c4)) & 63;          // the four k of this thread share a k-quad: one rotation
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      Ps[(c4 + x) * CF_LD + pos] = dpu2[x];
      Ps[(64 + c4 + x) * CF_LD + pos] = drr[x];       // (64 + c4) >> 2 = 16 + (c4 >> 2): the same rotation mod 64
    }
  }
  red[tid] = part;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) unsafeAtomicAdd(a.dblend, red[0] * g * (1.f - g));
  // phase 2: d(z2*ha) = dpu2 . RU[:, C:]
  f32x4 acc[NP][2];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int s = 0; s < 16; ++s) {
    const int k = 4 * s + kq;                        // k >> 2 = s: the rotation is 4 s, wave-uniform
    float av[NP], bv[2];
#pragma unroll
    for (int p = 0; p < NP; ++p) av[p] = Ps[k * CF_LD + ((wm * (ROWS / 2) + p * 16 + j + 4 * s) & 63)];
#pragma unroll
    for (int q = 0; q < 2; ++q) bv[q] = f.ruh[(size_t)k * f.ldW + wn * 32 + q * 16 + j];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[p][q] = MFMA16(av[p], bv[q], acc[p][q]);
  }
  __syncthreads();   // every wave is done reading the dpu2 rows of the tile
  // phase 3: the z half of dpg2 -> global + A tile; dha so far moves into the accumulators of the second contraction
  // (the row scratch written in phase 1 is read back by other threads of this workgroup: ordered by the barriers)
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const int lr = wm * (ROWS / 2) + p * 16 + 4 * kq + e4, col = wn * 32 + q * 16 + j;
        float dz = 0.f, dha = 0.f;
        if (row0 + lr < rows) {
          const size_t idx = (row0 + lr) * 64 + col;
          const float dzh2 = acc[p][q][e4];
          dha = a.dha[idx] + dzh2 * a.z2[idx];
          dz = dzh2 * a.dzh2[idx];
          a.dpg2[(row0 + lr) * 128 + col] = dz;
        }
        Ps[col * CF_LD + ((lr + CF_ROT(col)) & 63)] = dz;
        acc[p][q][e4] = dha;
      }
  __syncthreads();
  // phase 4: dha += dpg2 . RG[:, C:]
#pragma unroll 8
  for (int s = 0; s < 32; ++s) {
    const int k = 4 * s + kq;
    float av[NP], bv[2];
#pragma unroll
    for (int p = 0; p < NP; ++p) av[p] = Ps[k * CF_LD + ((wm * (ROWS / 2) + p * 16 + j + 4 * s) & 63)];
#pragma unroll
    for (int q = 0; q < 2; ++q) bv[q] = f.rgh[(size_t)k * f.ldW + wn * 32 + q * 16 + j];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[p][q] = MFMA16(av[p], bv[q], acc[p][q]);
  }
  // phase 5: graph cell output algebra (MultiATGCN.py:127)
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const int lr = wm * (ROWS / 2) + p * 16 + 4 * kq + e4, col = wn * 32 + q * 16 + j;
        if (row0 + lr >= rows) continue;
        const size_t idx = (row0 + lr) * 64 + col;
        const float dha = acc[p][q][e4];
        const float h = a.hprev ? a.hprev[idx] : 0.f, r = a.r[idx], hc = a.hc[idx];
        a.dr[idx] = dha * (h - hc);
        a.dh[idx] = dha * r;
        a.dpu[idx] = dha * (1.f - r) * (1.f - hc * hc);
      }
}

// part 2: gradient of z2*ha arrived -> dz2, dha += dzh2 * z2, gate pre-activation gradient of the residual cell
__global__ __launch_bounds__(256) void k_chain_res_gate(ChainArgs a) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)a.B * a.Np * 64) return;
  const size_t row = idx >> 6;
  const int o = idx & 63;
  const float ha = ha_of(a, idx), z2 = a.z2[idx], r2 = a.r2[idx], dzh2 = a.dzh2[idx];
  a.dha[idx] += dzh2 * z2;
  a.dpg2[row * 128 + o] = dzh2 * ha * z2 * (1.f - z2);
  a.dpg2[row * 128 + 64 + o] = a.dr[idx] * r2 * (1.f - r2);
}

// part 3: graph cell output algebra (MultiATGCN.py:127): ha = r h + (1-r) hc
__global__ __launch_bounds__(256) void k_chain_cell_out(ChainArgs a) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)a.B * a.Np * 64) return;
  const float h = a.hprev ? a.hprev[idx] : 0.f, r = a.r[idx], hc = a.hc[idx], dha = a.dha[idx];
  a.dr[idx] = dha * (h - hc);
  a.dh[idx] = dha * r;
  a.dpu[idx] = dha * (1.f - r) * (1.f - hc * hc);
}

// part 4: gradient of z*h arrived (slot 0 of the update AGCN's dA + transposed mix of its dense slots)
__global__ __launch_bounds__(256) void k_chain_cell_gate(ChainArgs a) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)a.B * a.Np * 64) return;
  const size_t row = idx >> 6;
  const int o = idx & 63;
  const size_t b = row / a.Np, n = row - b * a.Np;
  float dzh = a.dzhA[((b * a.S) * a.Np + n) * 64 + o];
  if (a.dzhMix)
    for (int pt = 0; pt < a.mixParts; ++pt) dzh += a.dzhMix[idx + (size_t)pt * a.mixPartStride];
  const float h = a.hprev ? a.hprev[idx] : 0.f, z = a.z[idx], r = a.r[idx];
  a.dh[idx] += dzh * z;
  a.dpg[row * 128 + o] = dzh * h * z * (1.f - z);
  a.dpg[row * 128 + 64 + o] = a.dr[idx] * r * (1.f - r);
}

// gradient of a layer's initial state, unpadded (B, N, 64): dh + slot 0 of the gate AGCN's dA of step 0 + its transposed
// mix - exactly what the fused chain kernel adds up as the carry into an earlier step (dA / mix null: dense GRU layer)
__global__ __launch_bounds__(256) void k_dh0_out(const float* __restrict__ dh, const float* __restrict__ dA,
                                                 const float* __restrict__ mix, int mixParts, long mixPartStride,
                                                 float* __restrict__ out, int B, int N, int Np, int S) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)B * N * 64) return;
  const int o = idx & 63;
  const size_t row = idx >> 6, b = row / N, n = row - b * N;
  const size_t p = (b * Np + n) * 64 + o;
  float v = dh[p];
  if (dA) v += dA[((b * S) * Np + n) * 64 + o];
  if (mix)
    for (int pt = 0; pt < mixParts; ++pt) v += mix[p + (size_t)pt * mixPartStride];
  out[idx] = v;
}

// ---- round 4: the row-local part of the chain AND the update block's node contraction, one workgroup per node ---------
// k_chain_res_fused (above) walked the rows [b][n] in blocks of 64 consecutive rows.  Its element-wise phases between the
// two nn.Linear contractions ran in the MFMA accumulator layout: 4-byte accesses behind per-element row guards, i.e. one
// EXPOSED memory round trip per element and phase (the ISA waits on vmcnt(0) 16 times per phase), 26 slab passes, 8 % of
// the matrix pipe, 44 us per launch - and the node contraction of the update block then re-read dpu as a launch of its own.
// Here a workgroup owns ONE node and 64 batch rows (the forward node kernels' work item): every global access is a float4
// of a 256-byte row piece in the row layout (thread = row x 16-byte slot), requested in one batch per phase; the two
// nn.Linear products and the node contraction take their A operands from swizzled LDS tiles (ds_read_b128, a float4 = four
// reduction steps) and hand their results back to the row layout through LDS; what a row needs across the phases (dha,
// z2, h, r, hc, ..) stays in registers.  The B operands of the nn.Linear products are fragment-ordered transposes made
// once per parameter update (k_prep_linear_t16), those of the node contraction the plain folded weights of the node.
//   in : dseq (+ carry of step t+1 = dh + slot 0 of the gate block's dA + its transposed mix), hprev, r, hc, r2, z2, hc2
//   out: dpu2, dpg2 (kept for the batched part), dpu (kept), dr, dh (partial), blend-scalar gradient,
//        dA_u[b][s][n][0:64] = dpu[b][n][:] . WpU[n][s][C + i][:]   for every kept slot s
// Rows of the padding nodes are never touched: the caller keeps them zero where a consumer sums over all rows (DPU2, DPG2).
struct ChainResNodeArgs {
  ChainArgs c;
  const float* carryA;   // dA of the gate AGCN of step t+1 ([B][S][Np][64], slot 0 is read) or null
  const float* carryMix; // its transposed mix [parts][B][Np][64] or null
  const float* ruf;      // [4 g][4 ct][64][4]  B fragments of d(z2 ha) = dpu2 . RU[:, C:]   (k = o, n = i)
  const float* rgf;      // [8 g][4 ct][64][4]  B fragments of dha    += dpg2 . RG[:, C:]
  const float* Wp;       // plain folded weights of the update AGCN [N][S][I][64]
  float* dA;             // [B][S][Np][64]
  int I, iOfs;
};

// nn.Linear weight W (O, I), columns C.. = the 64 hidden inputs: out[g][ct][lane][s] = W[16 g + 4 (lane >> 4) + s][C + 16 ct + (lane & 15)]
__global__ __launch_bounds__(256) void k_prep_linear_t16(const float* __restrict__ W, int I, int C, int O, float* __restrict__ out) {
  const int unit = blockIdx.x * 256 + threadIdx.x;
  if (unit >= (O / 16) * 4 * 64) return;
  const int lane = unit & 63, ct = (unit >> 6) & 3, g = unit >> 8;
  float v[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) v[s4] = W[(size_t)(16 * g + 4 * (lane >> 4) + s4) * I + C + 16 * ct + (lane & 15)];
  *reinterpret_cast<float4*>(out + (size_t)unit * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

#define CRN_LDS (3 * 4096 * (int)sizeof(float))
__device__ __forceinline__ float4 f4_add(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_mul(const float4& a, const float4& b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// CARRY: step t+1 exists (dcarry, carryA and PARTS partial transposed mixes are added to the incoming gradient); HPREV: h_{t-1}
// exists.  Template parameters, not runtime branches: a request behind a branch ends a basic block and the compiler drains the
// whole queue (vmcnt(0)) where the paths meet - phase 1 is ONE batch of 8 + 2 + PARTS requests per row sweep.
template <bool CARRY, bool HPREV, int PARTS>
__global__ __launch_bounds__(512, 4) void k_chain_res_node(ChainResNodeArgs f) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* R0 = lds;             // dpu2 tile -> z half of dpg2 -> dpu tile
  float* R1 = lds + 4096;      // r half of dpg2 -> output tile 0 of the node contraction
  float* R2 = lds + 8192;      // GEMM results on their way back to the row layout -> output tile 1
  __shared__ float red[8];
  const ChainArgs& a = f.c;
  const int n = blockIdx.y, rowBase = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
  const int ct = w & 3, rh = w >> 2;
  const int srow = tid >> 4, sq = tid & 15;
  const float g = sigmoid_f(a.blend[0]);
  // ---- phase 1, row layout (one batch of requests per row sweep; 32-bit byte offsets: SGPR base + one VGPR) ----
  float4 dha0[2], c1v[2], z2k[2];
  float part = 0.f;
  {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int b = rowBase + srow + 32 * it;
      const bool ok = b < a.B;
      const unsigned bb = (unsigned)min(b, a.B - 1);
      const unsigned ob = ((bb * a.Np + n) * 64 + sq * 4) * 4u;       // byte offset of this thread's float4 in a [B][Np][64] slab
      float4 d = ld4b(a.dseq, ob);
      const float4 hv = HPREV ? ld4b(a.hprev, ob) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 rv = ld4b(a.r, ob), hcv = ld4b(a.hc, ob), r2v = ld4b(a.r2, ob), hc2v = ld4b(a.hc2, ob);
      z2k[it] = ld4b(a.z2, ob);
      if constexpr (CARRY) {
        const float4 cry = ld4b(a.dcarry, ob), cA = ld4b(f.carryA, (((bb * a.S) * a.Np + n) * 64 + sq * 4) * 4u);
        float4 cM[PARTS > 0 ? PARTS : 1];
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) cM[pt] = ld4b(f.carryMix + (size_t)pt * a.mixPartStride, ob);
        d = f4_add(d, f4_add(cry, cA));
#pragma unroll
        for (int pt = 0; pt < PARTS; ++pt) d = f4_add(d, cM[pt]);
      }
      float dpu2[4], drr[4], dh0[4], c1[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const float dd = get4(d, x), h = get4(hv, x), r = get4(rv, x), hc = get4(hcv, x);
        const float r2 = get4(r2v, x), z2 = get4(z2k[it], x), hc2 = get4(hc2v, x);
        const float ha = r * h + (1.f - r) * hc;
        const float res = r2 * ha + (1.f - r2) * hc2;
        if (ok) part += dd * (ha - res);
        const float dres = (1.f - g) * dd;
        dh0[x] = g * dd + dres * r2;
        dpu2[x] = dres * (1.f - r2) * (1.f - hc2 * hc2);
        drr[x] = dres * (ha - hc2) * r2 * (1.f - r2);
        c1[x] = ha * z2 * (1.f - z2);
      }
      dha0[it] = make_float4(dh0[0], dh0[1], dh0[2], dh0[3]);
      c1v[it] = make_float4(c1[0], c1[1], c1[2], c1[3]);
      const float4 v2 = make_float4(dpu2[0], dpu2[1], dpu2[2], dpu2[3]), vr = make_float4(drr[0], drr[1], drr[2], drr[3]);
      const int lr = srow + 32 * it, pos = (lr * 16 + (sq ^ (lr & 15))) * 4;
      *reinterpret_cast<float4*>(&R0[pos]) = v2;
      *reinterpret_cast<float4*>(&R1[pos]) = vr;
      if (ok) {
        st4b(a.dpu2, ob, v2);
        st4b(a.dpg2, 2 * (ob - sq * 16) + 256 + sq * 16, vr);
      }
    }
    // blend-scalar gradient: wave sums now, the workgroup's sum behind the next barriers
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) part += __shfl_xor(part, m, 64);
    if (lane == 0) red[w] = part;
  }
  // B fragments of the nn.Linear products (shared by every workgroup: L2): the first product's before the first barrier,
  // the second product's behind phase 2 (all twelve at once spilled phase 1's row values)
  __builtin_amdgcn_sched_barrier(0);
  float4 bu[4], bgf[8];
  {
    const float4* pu = reinterpret_cast<const float4*>(f.ruf) + (size_t)ct * 64 + lane;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) bu[gq] = pu[gq * 256];
  }
  __syncthreads();
  if (tid == 0 && a.dblend) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += red[q];
    unsafeAtomicAdd(a.dblend, sum * g * (1.f - g));
  }
  // ---- phase 2: d(z2 ha) = dpu2 . RU[:, C:]  (wave = column tile ct x row half rh) ----
  f32x4 acc[2];
  auto frag = [&](const float* tile, int rt, int gq) {
    return *reinterpret_cast<const float4*>(&tile[((rt * 16 + j) * 16 + ((4 * gq + kq) ^ j)) * 4]);
  };
  auto to_tile = [&](float* tile) {   // the wave's 2 x (16 x 16) results -> row-major tile
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[swz((2 * rh + q) * 16 + 4 * kq + e, 16 * ct + j, 16)] = acc[q][e];
  };
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const float4 a0 = frag(R0, 2 * rh, gq), a1 = frag(R0, 2 * rh + 1, gq), bv = bu[gq];
    acc[0] = MFMA16(a0.x, bv.x, acc[0]); acc[1] = MFMA16(a1.x, bv.x, acc[1]);
    acc[0] = MFMA16(a0.y, bv.y, acc[0]); acc[1] = MFMA16(a1.y, bv.y, acc[1]);
    acc[0] = MFMA16(a0.z, bv.z, acc[0]); acc[1] = MFMA16(a1.z, bv.z, acc[1]);
    acc[0] = MFMA16(a0.w, bv.w, acc[0]); acc[1] = MFMA16(a1.w, bv.w, acc[1]);
  }
  to_tile(R2);
  {
    const float4* pg = reinterpret_cast<const float4*>(f.rgf) + (size_t)ct * 64 + lane;
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) bgf[gq] = pg[gq * 256];
  }
  __syncthreads();
  // ---- phase 3, row layout: dha += dzh2 z2; the z half of dpg2 -> global + tile (over the dpu2 tile: phase 2 is through) ----
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int lr = srow + 32 * it, pos = (lr * 16 + (sq ^ (lr & 15))) * 4;
    const float4 dzh2 = *reinterpret_cast<const float4*>(&R2[pos]);
    dha0[it] = f4_add(dha0[it], f4_mul(dzh2, z2k[it]));
    const float4 dz = f4_mul(dzh2, c1v[it]);
    *reinterpret_cast<float4*>(&R0[pos]) = dz;
    const int b = rowBase + lr;
    if (b < a.B) st4b(a.dpg2, (((unsigned)b * a.Np + n) * 128 + sq * 4) * 4u, dz);
  }
  const int nCt = 4 * a.S;
  auto wload = [&](int tile, int gq) -> float4 {
    const int slot = tile >> 2, i0 = (tile & 3) * 16;
    const size_t wrow = ((size_t)n * a.S + slot) * f.I + f.iOfs + i0 + j;
    return (reinterpret_cast<const float4*>(f.Wp + wrow * 64) + kq)[gq * 4];
  };
  // what phase 5 needs again of the saved state (read in phase 1: L2 hits), in flight under phase 4
  float4 hk[2], rk[2], hck[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const unsigned ob = (((unsigned)min(rowBase + srow + 32 * it, a.B - 1) * a.Np + n) * 64 + sq * 4) * 4u;
    hk[it] = HPREV ? ld4b(a.hprev, ob) : make_float4(0.f, 0.f, 0.f, 0.f);
    rk[it] = ld4b(a.r, ob); hck[it] = ld4b(a.hc, ob);
  }
  __syncthreads();
  // ---- phase 4: dha += dpg2 . RG[:, C:]  (k 0..63 the z half, 64..127 the r half) ----
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
#pragma unroll
  for (int gq = 0; gq < 8; ++gq) {
    const float* tile = gq < 4 ? R0 : R1;
    const float4 a0 = frag(tile, 2 * rh, gq & 3), a1 = frag(tile, 2 * rh + 1, gq & 3), bv = bgf[gq];
    acc[0] = MFMA16(a0.x, bv.x, acc[0]); acc[1] = MFMA16(a1.x, bv.x, acc[1]);
    acc[0] = MFMA16(a0.y, bv.y, acc[0]); acc[1] = MFMA16(a1.y, bv.y, acc[1]);
    acc[0] = MFMA16(a0.z, bv.z, acc[0]); acc[1] = MFMA16(a1.z, bv.z, acc[1]);
    acc[0] = MFMA16(a0.w, bv.w, acc[0]); acc[1] = MFMA16(a1.w, bv.w, acc[1]);
  }
  to_tile(R2);   // (phase 3 read R2 before the barrier above)
  // the node contraction's weights of this wave's first column tile land under phase 5
  float4 wt[4];
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) wt[gq] = wload(min(w, nCt - 1), gq);
  __syncthreads();
  // ---- phase 5, row layout: graph cell output algebra (MultiATGCN.py:127); dpu -> global + tile ----
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int lr = srow + 32 * it, pos = (lr * 16 + (sq ^ (lr & 15))) * 4;
    const float4 dha = f4_add(dha0[it], *reinterpret_cast<const float4*>(&R2[pos]));
    float dr4[4], dh4[4], dpu4[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float d = get4(dha, x), h = get4(hk[it], x), r = get4(rk[it], x), hc = get4(hck[it], x);
      dr4[x] = d * (h - hc);
      dh4[x] = d * r;
      dpu4[x] = d * (1.f - r) * (1.f - hc * hc);
    }
    const float4 vpu = make_float4(dpu4[0], dpu4[1], dpu4[2], dpu4[3]);
    *reinterpret_cast<float4*>(&R0[pos]) = vpu;
    const int b = rowBase + lr;
    if (b < a.B) {
      const unsigned ob = (((unsigned)b * a.Np + n) * 64 + sq * 4) * 4u;
      st4b(a.dr, ob, make_float4(dr4[0], dr4[1], dr4[2], dr4[3]));
      st4b(a.dh, ob, make_float4(dh4[0], dh4[1], dh4[2], dh4[3]));
      st4b(a.dpu, ob, vpu);
    }
  }
  __syncthreads();
  // ---- phase 6: dA_u = dpu . WpU[n]^T, column tiles w, w + 8, .. of the 4 S (slot tile >> 2, hidden columns 16 (tile & 3) ..) ----
  const int passes = (nCt + 7) >> 3;
  for (int pass = 0; pass < passes; ++pass) {
    const int tile = w + 8 * pass;
    if (tile < nCt) {                               // wave-uniform
      f32x4 ac[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) ac[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int nextTile = min(tile + 8, nCt - 1);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        float4 av[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) av[rt] = frag(R0, rt, gq);
        const float4 wv = wt[gq];
        wt[gq] = wload(nextTile, gq);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) ac[rt] = MFMA16(av[rt].x, wv.x, ac[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) ac[rt] = MFMA16(av[rt].y, wv.y, ac[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) ac[rt] = MFMA16(av[rt].z, wv.z, ac[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) ac[rt] = MFMA16(av[rt].w, wv.w, ac[rt]);
      }
      float* tileOut = (w >> 2) ? R2 : R1;
      const int col = (w & 3) * 16 + j;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) tileOut[swz(rt * 16 + 4 * kq + e, col, 16)] = ac[rt][e];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int id2 = tid + 512 * q, sl = id2 >> 10, lb = (id2 >> 4) & 63, s4 = id2 & 15;
      const int slot = 2 * pass + sl, b = rowBase + lb;
      if (slot < a.S && b < a.B) {
        const float4 v = *reinterpret_cast<const float4*>(&(sl ? R2 : R1)[(lb * 16 + (s4 ^ (lb & 15))) * 4]);
        st4b(f.dA, ((((unsigned)b * a.S + slot) * a.Np + n) * 64 + 4 * s4) * 4u, v);
      }
    }
    if (pass + 1 < passes) __syncthreads();
  }
}

// ---- the chain's node contraction, dedicated kernel ----------------------------------------------------------------
//   dA[b][s][n][i] (+)= sum_o dPre[b][n][o] * Wp[n][s][iOfs + i][o]        for the 64 hidden columns i of every slot s
// the transpose of the forward's node-wise contraction (MultiATGCN.py:108), with the forward kernel's structure: one
// workgroup per (node, 64-row block), the A tile - the pre-activation gradients of the node's rows - in XOR-swizzled
// LDS, the plain weights read straight from global memory (a lane's B fragment is four consecutive o: one aligned
// float4 of the plain layout), 16x16x4 MFMA, 4 row tiles x (4 S) column tiles.  GATE: the tile is not read but COMPUTED
// in the prologue - the gate algebra of the graph cell (what k_chain_cell_gate did as a launch of its own:
// dzh = slot 0 of the update block's dA + its transposed mix; dh += dzh z; dpg = [dzh h z (1-z) | dr r (1-r)]) - and
// stored to DPG for the batched part on the way.  Replaces k_chain_cell_gate + two generic k_bgemm launches per step.
struct ChainNodeArgs {
  ChainArgs c;           // GATE: operands of the gate algebra (dzhA, dzhMix, hprev, z, r, dr, dh, dpg)
  const float* dPre;     // !GATE: the pre-activation gradients [rows][Np][O]   (O = 192: the 128 gate columns ...
  const float* dPre2;    //                                                      ... and here the 64 update columns)
  const float* Wp;       // plain folded weights [N][S][I][O]                   (O = 192: of the gate AGCN, O = 128 ...
  const float* Wp2;      //                                                      ... and of the update AGCN, O = 64)
  float* dA;             // [rows][S][Np][64]
  int I, iOfs, rows, N, Np, S;
  float beta;            // 1: add to what dA holds (the x-column gradient of the layer above rides in the gate block)
};

// (grid: x = 64-row block, y = node - the row blocks of one node are neighbours in launch order, so the batched
// x-column calls, which have 23 row blocks per node, re-read a node's weights from L2.)
template <bool GATE, int O>
__global__ __launch_bounds__(512, 4) void k_chain_node(ChainNodeArgs p) {
  constexpr int NG = O / 16;
  static_assert(!GATE || O == 128, "the gate algebra produces the 128 gate columns");
  __shared__ __attribute__((aligned(16))) float As[(O / 64) * 4096];   // [O/64 chunks][64 rows][16 slots], swizzled
  const int n = blockIdx.y, rowBase = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const int srow = tid >> 4, sq = tid & 15;
  // ---- weights of this wave's first column tile (ct = w): they depend on nothing this kernel computes, so they are
  // requested before the A tile is built and land under the gate algebra ----
  const int nCt = 4 * p.S;
  constexpr int O1 = O == 192 ? 128 : O;     // columns of the first operand pair (O = 192: gate 128 | update 64)
  auto wload1 = [&](int ct, int g) -> float4 {
    const int slot = ct >> 2, i0 = (ct & 3) * 16;
    const size_t wrow = ((size_t)n * p.S + slot) * p.I + p.iOfs + i0 + j;
    if (O == 192 && g >= O1 / 16) return (reinterpret_cast<const float4*>(p.Wp2 + wrow * 64) + kq)[(g - O1 / 16) * 4];
    return (reinterpret_cast<const float4*>(p.Wp + wrow * O1) + kq)[g * 4];
  };
  // ONE register set of weights: group g of the next column tile is requested as soon as group g of this one has been
  // consumed, so the loads of tile ct + 8 fly under the MFMAs of tile ct (two whole sets - 96 registers at O = 192 - spill)
  float4 wt[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) wt[g] = wload1(min(w, nCt - 1), g);
  // ---- A tile ----
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int lr = srow + 32 * it, b = rowBase + lr;
    const bool ok = b < p.rows;
    const size_t row = (size_t)min(b, p.rows - 1) * p.Np + n, idx = row * 64 + sq * 4;
    const int pos = (lr * 16 + (sq ^ (lr & 15))) * 4;
    if (GATE) {
      const ChainArgs& a = p.c;
      float4 dzh = ld4(a.dzhA, (((size_t)min(b, p.rows - 1) * a.S) * a.Np + n) * 64 + sq * 4);
      if (a.dzhMix)
        for (int pt = 0; pt < a.mixParts; ++pt) {
          const float4 m = ld4(a.dzhMix + (size_t)pt * a.mixPartStride, idx);
          dzh = make_float4(dzh.x + m.x, dzh.y + m.y, dzh.z + m.z, dzh.w + m.w);
        }
      const float4 h = a.hprev ? ld4(a.hprev, idx) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 z = ld4(a.z, idx), r = ld4(a.r, idx), dr = ld4(a.dr, idx), dh = ld4(a.dh, idx);
      const float4 gz = make_float4(dzh.x * h.x * z.x * (1.f - z.x), dzh.y * h.y * z.y * (1.f - z.y),
                                    dzh.z * h.z * z.z * (1.f - z.z), dzh.w * h.w * z.w * (1.f - z.w));
      const float4 gr = make_float4(dr.x * r.x * (1.f - r.x), dr.y * r.y * (1.f - r.y), dr.z * r.z * (1.f - r.z),
                                    dr.w * r.w * (1.f - r.w));
      if (ok) {
        *reinterpret_cast<float4*>(a.dh + idx) = make_float4(dh.x + dzh.x * z.x, dh.y + dzh.y * z.y, dh.z + dzh.z * z.z,
                                                             dh.w + dzh.w * z.w);
        *reinterpret_cast<float4*>(a.dpg + row * 128 + sq * 4) = gz;
        *reinterpret_cast<float4*>(a.dpg + row * 128 + 64 + sq * 4) = gr;
      }
      *reinterpret_cast<float4*>(&As[pos]) = gz;             // rows past the batch: garbage in, discarded out (row-local)
      *reinterpret_cast<float4*>(&As[4096 + pos]) = gr;
    } else {
      *reinterpret_cast<float4*>(&As[pos]) = ld4(p.dPre, row * O1 + sq * 4);
      if (O1 == 128) *reinterpret_cast<float4*>(&As[4096 + pos]) = ld4(p.dPre, row * O1 + 64 + sq * 4);
      if (O == 192) *reinterpret_cast<float4*>(&As[2 * 4096 + pos]) = ld4(p.dPre2, row * 64 + sq * 4);
    }
  }
  __syncthreads();
  // ---- contraction: column tiles ct = w, w + 8, .. of the 4 S tiles (slot ct >> 2, hidden columns 16 (ct & 3) ..) ----
  auto contract = [&](int ct, f32x4 (&acc)[4]) {
    const int nextCt = min(ct + 8, nCt - 1);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float* buf = As + (g >> 2) * 4096;
      float4 av[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        av[rt] = *reinterpret_cast<const float4*>(&buf[((rt * 16 + j) * 16 + ((4 * (g & 3) + kq) ^ j)) * 4]);
      const float4 wg = wt[g];
      wt[g] = wload1(nextCt, g);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].x, wg.x, acc[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].y, wg.y, acc[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].z, wg.z, acc[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].w, wg.w, acc[rt]);
      __builtin_amdgcn_sched_barrier(0);        // keeps the A reads of later groups from being hoisted (they spilled)
    }
  };
  if constexpr (O <= 128) {
    // Round 3: the eight tiles of a pass are two whole slots of the node's 64 rows - [2][64 rows][64 columns] - and leave
    // through LDS as 256-byte rows (float4 per thread, read-modify-write where the block already holds the x-column
    // gradient of the layer above) instead of as 32 scalar stores / read-modify-writes per lane in 64-byte pieces: the
    // next kernel of the chain (the transposed mix) reads exactly these rows.
    __shared__ __attribute__((aligned(16))) float Out[2 * 4096];
    const int passes = (nCt + 7) >> 3;
    for (int pass = 0; pass < passes; ++pass) {
      const int ct = w + 8 * pass;
      if (ct < nCt) {                             // wave-uniform
        f32x4 acc[4];
        contract(ct, acc);
        float* tileOut = Out + (w >> 2) * 4096;
        const int i0 = (w & 3) * 16;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int lb = rt * 16 + 4 * kq + e, col = i0 + j;
            tileOut[(lb * 16 + ((col >> 2) ^ (lb & 15))) * 4 + (col & 3)] = acc[rt][e];
          }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = tid + 512 * q, sl = idx >> 10, lb = (idx >> 4) & 63, s4 = idx & 15;
        const int slot = 2 * pass + sl, b = rowBase + lb;
        if (slot < p.S && b < p.rows) {
          float4 v = *reinterpret_cast<const float4*>(&Out[sl * 4096 + (lb * 16 + (s4 ^ (lb & 15))) * 4]);
          float* dst = p.dA + (((size_t)b * p.S + slot) * p.Np + n) * 64 + 4 * s4;
          if (p.beta != 0.f) {
            const float4 o = *reinterpret_cast<const float4*>(dst);
            v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
          }
          *reinterpret_cast<float4*>(dst) = v;
        }
      }
      if (pass + 1 < passes) __syncthreads();
    }
  } else {
    for (int ct = w; ct < nCt; ct += 8) {
      f32x4 acc[4];
      contract(ct, acc);
      const int slot = ct >> 2, i0 = (ct & 3) * 16;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int b = rowBase + rt * 16 + 4 * kq + e;
          if (b >= p.rows) continue;
          float* dst = p.dA + (((size_t)b * p.S + slot) * p.Np + n) * 64 + i0 + j;
          *dst = p.beta != 0.f ? *dst + acc[rt][e] : acc[rt][e];
        }
    }
  }
}

// ---- round 4: the gate block of the chain with its requests batched -------------------------------------------------
// k_chain_node<true, 128> above, restated: its prologue looped over the partial transposed mixes and guarded its stores -
// four or five dependent waits (vmcnt(0)) per row sweep - and its read-modify-write epilogue (the block already holds the
// x-column gradient of the layer above) waited for every old value right before its store, 40 us per launch for 2 GFLOP.
// Here the PARTS partial mixes and the presence of old values are template parameters, h_{t-1} always exists (the launcher
// passes a zero slab at t = 0), every request of a row sweep leaves in one batch with 32-bit byte offsets from SGPR bases,
// and the old values of a pass are requested before its MFMAs.
template <int PARTS, bool BETA>
__global__ __launch_bounds__(512, 4) void k_chain_gate_node(ChainNodeArgs p) {
  constexpr int NG = 8;
  __shared__ __attribute__((aligned(16))) float As[2 * 4096];    // [gz | gr][64 rows][16 slots], swizzled
  __shared__ __attribute__((aligned(16))) float Out[2 * 4096];
  const ChainArgs& a = p.c;
  const int n = blockIdx.y, rowBase = blockIdx.x * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
  const int srow = tid >> 4, sq = tid & 15;
  const int nCt = 4 * p.S;
  auto wload1 = [&](int ct, int g) -> float4 {
    const int slot = ct >> 2, i0 = (ct & 3) * 16;
    const size_t wrow = ((size_t)n * p.S + slot) * p.I + p.iOfs + i0 + j;
    return (reinterpret_cast<const float4*>(p.Wp + wrow * 128) + kq)[g * 4];
  };
  float4 wt[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) wt[g] = wload1(min(w, nCt - 1), g);
  // ---- A tile = the gate algebra of the graph cell (MultiATGCN.py:122-125 transposed), row layout ----
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int lr = srow + 32 * it, b = rowBase + lr;
    const bool ok = b < p.rows;
    const unsigned bb = (unsigned)min(b, p.rows - 1);
    const unsigned ob = ((bb * p.Np + n) * 64 + sq * 4) * 4u;
    float4 dzh = ld4b(a.dzhA, (((bb * a.S) * a.Np + n) * 64 + sq * 4) * 4u);
    float4 mx[PARTS > 0 ? PARTS : 1];
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) mx[pt] = ld4b(a.dzhMix + (size_t)pt * a.mixPartStride, ob);
    const float4 h = ld4b(a.hprev, ob), z = ld4b(a.z, ob), r = ld4b(a.r, ob), dr = ld4b(a.dr, ob), dh = ld4b(a.dh, ob);
#pragma unroll
    for (int pt = 0; pt < PARTS; ++pt) dzh = f4_add(dzh, mx[pt]);
    const float4 gz = make_float4(dzh.x * h.x * z.x * (1.f - z.x), dzh.y * h.y * z.y * (1.f - z.y),
                                  dzh.z * h.z * z.z * (1.f - z.z), dzh.w * h.w * z.w * (1.f - z.w));
    const float4 gr = make_float4(dr.x * r.x * (1.f - r.x), dr.y * r.y * (1.f - r.y), dr.z * r.z * (1.f - r.z),
                                  dr.w * r.w * (1.f - r.w));
    const int pos = (lr * 16 + (sq ^ (lr & 15))) * 4;
    *reinterpret_cast<float4*>(&As[pos]) = gz;             // rows past the batch: garbage in, discarded out (row-local)
    *reinterpret_cast<float4*>(&As[4096 + pos]) = gr;
    if (ok) {
      st4b(a.dh, ob, f4_add(dh, f4_mul(dzh, z)));
      st4b(a.dpg, 2 * (ob - sq * 16) + sq * 16, gz);
      st4b(a.dpg, 2 * (ob - sq * 16) + 256 + sq * 16, gr);
    }
  }
  __syncthreads();
  const int passes = (nCt + 7) >> 3;
  for (int pass = 0; pass < passes; ++pass) {
    const int ct = w + 8 * pass;
    // what the block already holds for this pass' rows (the x-column gradient of the layer above): in flight under the MFMAs
    float4 old[4];
    if constexpr (BETA) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = tid + 512 * q, sl = idx >> 10, lb = (idx >> 4) & 63, s4 = idx & 15;
        const unsigned slot = (unsigned)min(2 * pass + sl, p.S - 1), b = (unsigned)min(rowBase + lb, p.rows - 1);
        old[q] = ld4b(p.dA, (((b * p.S + slot) * p.Np + n) * 64 + 4 * s4) * 4u);
      }
    }
    if (ct < nCt) {                             // wave-uniform
      f32x4 acc[4];
      const int nextCt = min(ct + 8, nCt - 1);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float* buf = As + (g >> 2) * 4096;
        float4 av[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          av[rt] = *reinterpret_cast<const float4*>(&buf[((rt * 16 + j) * 16 + ((4 * (g & 3) + kq) ^ j)) * 4]);
        const float4 wg = wt[g];
        wt[g] = wload1(nextCt, g);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].x, wg.x, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].y, wg.y, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].z, wg.z, acc[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].w, wg.w, acc[rt]);
        __builtin_amdgcn_sched_barrier(0);        // keeps the A reads of later groups from being hoisted (they spilled)
      }
      float* tileOut = Out + (w >> 2) * 4096;
      const int col = (w & 3) * 16 + j;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) tileOut[swz(rt * 16 + 4 * kq + e, col, 16)] = acc[rt][e];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 512 * q, sl = idx >> 10, lb = (idx >> 4) & 63, s4 = idx & 15;
      const int slot = 2 * pass + sl, b = rowBase + lb;
      if (slot < p.S && b < p.rows) {
        float4 v = *reinterpret_cast<const float4*>(&Out[sl * 4096 + (lb * 16 + (s4 ^ (lb & 15))) * 4]);
        if constexpr (BETA) v = f4_add(v, old[q]);
        st4b(p.dA, ((((unsigned)b * p.S + slot) * p.Np + n) * 64 + 4 * s4) * 4u, v);
      }
    }
    if (pass + 1 < passes) __syncthreads();
  }
}

// ---- node-adaptive weight gradients of 64-channel rows: every slot of a node in ONE workgroup ----------------------
//   dW[n][s][iOfs + i][o] += sum_{(t, b)} XA[t][b][n][s][i] * dPre[t][b][n][o]          (i < 64, o < O)
// XA slot 0 = the rows themselves (U: [T*B][Np][64]), slots 1.. = their graph mixes where the forward left them: per
// step t a block g[t] with the node's rows at g[t] + n*gNode[t] + b*Ks*64 (the chunk blocks of the hoisted x part have
// different sizes, the recurrent rows of a lower layer sit one step off in the blocks of the layer above - a table per
// step covers all of them; g[t] = null: the mix of that step is zero).
// As batched GEMMs (one per chunk block, identity slot apart: 47 launches a step at BM) each 64 x 64 tile re-read its
// pre-activation rows per slot and column tile and the K loops were 64..384 rows short: 43 TFLOP/s.  Here a workgroup
// owns a node and a range of steps: S*64 x O outputs (wave = slot x 64 columns, 4 x 4 MFMA tiles), both operands staged
// once per 16 rows through LDS (k-major as they lie in memory: no transposition), partial sums added atomically.
#define WG_KT 16
struct WgradNodeArgs {
  const float* U;
  const float* dPre;
  float* dW;                    // at [n = 0][s = 0][iOfs][0]
  const float* g[64];           // MAX_STEPS
  long gNode[64];
  int T, B, N, Np, S, Ks, I, stepsPerPart;
  float* dBias;                 // [N][O] += column sums of dPre over the rows (the AGCN's bias gradient), or null
};
__device__ __forceinline__ void wg_multiply(const float* Ab, const float* Bb, int AM, int BM, f32x4 (&acc)[4][4]) {
#pragma unroll
  for (int kk = 0; kk < WG_KT / 4; ++kk) {
    float av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      av[q] = Ab[4 * kk * AM + 16 * q];
      bv[q] = Bb[4 * kk * BM + 16 * q];
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA16(av[mt], bv[nt], acc[mt][nt]);
    __builtin_amdgcn_sched_barrier(0);   // the operand reads of later k groups stay behind these MFMAs (hoisted, they spill)
  }
}
template <int O, int S, int MINW>   // S * O threads (a wave per slot and 64 columns); MINW waves per SIMD the register budget allows
__global__ __launch_bounds__(S * O, MINW) void k_wgrad_node(WgradNodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float wgLds[];
  // the per-step table is looked up with a per-lane step: from LDS (indexing the kernel argument with a vector index
  // would make the compiler copy the whole struct to scratch)
  __shared__ const float* sg[64];
  __shared__ long sgNode[64];
  for (int t = 0; t < a.T; ++t)
    if (threadIdx.x == 0) { sg[t] = a.g[t]; sgNode[t] = a.gNode[t]; }
  constexpr int AM = S * 64 + 16, BM = O + 16, NT = S * O;
  float* As = wgLds;                       // [2][WG_KT][AM]   (row stride = 16 mod 64 floats: the four k rows of an
  float* Bs = wgLds + 2 * WG_KT * AM;      // [2][WG_KT][BM]    operand read land in disjoint banks)
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const int ms = w % S, ns = w / S;
  const int tBeg = blockIdx.y * a.stepsPerPart, tEnd = min(a.T, tBeg + a.stepsPerPart);
  const int rows = (tEnd - tBeg) * a.B, nTiles = (rows + WG_KT - 1) / WG_KT;
  constexpr int NA = 256 / O;              // float4 units of A per thread and tile: 16 rows * S*16 units / (S * O threads)
  constexpr int nbUnits = WG_KT * O / 4;   // of B: 16 rows * O/4 units, ceil(4 / S) per thread
  constexpr int NB = (4 + S - 1) / S;
  // two register sets: the rows of tiles t+1 and t+2 are in flight while tile t is multiplied (every row of a node lies
  // in another page - 106 KB apart - so a load takes long; one tile ahead left the MFMAs waiting: 55 TFLOP/s)
  float4 ra[2][NA], rb[2][NB], bsum[NB];
#pragma unroll
  for (int q = 0; q < NB; ++q) bsum[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int tile, float4 (&fa)[NA], float4 (&fb)[NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int u = tid + q * NT, k = u / (S * 16), r = u - k * (S * 16), slot = r >> 4, c4 = r & 15;
      const int row = tile * WG_KT + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef WGN_LAB_NO_LOAD
      if (false) {
#else
      if (row < rows) {
#endif
        const int t = tBeg + row / a.B, b = row - (row / a.B) * a.B;
        if (slot == 0) v = *reinterpret_cast<const float4*>(a.U + (((size_t)t * a.B + b) * a.Np + n) * 64 + c4 * 4);
        else if (sg[t]) v = *reinterpret_cast<const float4*>(sg[t] + (size_t)n * sgNode[t] + ((size_t)b * a.Ks + slot - 1) * 64 + c4 * 4);
      }
      fa[q] = v;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int u = tid + q * NT, k = u / (O / 4), c4 = u - k * (O / 4);
      const int row = tile * WG_KT + k;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef WGN_LAB_NO_LOAD
      if (false) {
#else
      if (u < nbUnits && row < rows) {
#endif
        const int t = tBeg + row / a.B, b = row - (row / a.B) * a.B;
        v = *reinterpret_cast<const float4*>(a.dPre + (((size_t)t * a.B + b) * a.Np + n) * O + c4 * 4);
      }
      fb[q] = v;
    }
  };
  auto stash = [&](int buf, const float4 (&fa)[NA], const float4 (&fb)[NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int u = tid + q * NT, k = u / (S * 16), r = u - k * (S * 16);
      *reinterpret_cast<float4*>(&As[(buf * WG_KT + k) * AM + r * 4]) = fa[q];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int u = tid + q * NT, k = u / (O / 4), c4 = u - k * (O / 4);
      if (u < nbUnits) *reinterpret_cast<float4*>(&Bs[(buf * WG_KT + k) * BM + c4 * 4]) = fb[q];
      // every pre-activation row passes through here exactly once: its column sums are the node's bias gradient
      // (k_node_colsum's extra pass over DPG / DPU - 0.34 ms a step - is gone)
      bsum[q] = make_float4(bsum[q].x + fb[q].x, bsum[q].y + fb[q].y, bsum[q].z + fb[q].z, bsum[q].w + fb[q].w);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (fetch of a tile past the end returns zeros without touching memory)
  __syncthreads();
  fetch(0, ra[0], rb[0]);
  stash(0, ra[0], rb[0]);
  fetch(1, ra[1], rb[1]);
  __syncthreads();
  const float* Ab = As + ms * 64 + j + kq * AM;
  const float* Bb = Bs + ns * 64 + j + kq * BM;
  for (int tile = 0; tile < nTiles; tile += 2) {
    fetch(tile + 2, ra[0], rb[0]);
#ifndef WGN_LAB_NO_MFMA
    wg_multiply(Ab, Bb, AM, BM, acc);
#endif
    stash(1, ra[1], rb[1]);            // tile + 1
    __syncthreads();
    fetch(tile + 3, ra[1], rb[1]);
#ifndef WGN_LAB_NO_MFMA
    if (tile + 1 < nTiles) wg_multiply(Ab + WG_KT * AM, Bb + WG_KT * BM, AM, BM, acc);
#endif
    stash(0, ra[0], rb[0]);            // tile + 2
    __syncthreads();
  }
  if (a.dBias) {   // the 16 threads that hold the same 4 columns (one per k row of a tile) meet in LDS (free after the loop)
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int u = tid + q * NT, k = u / (O / 4), c4 = u - k * (O / 4);
      if (u < nbUnits) *reinterpret_cast<float4*>(&Bs[k * BM + c4 * 4]) = bsum[q];
    }
    __syncthreads();
    if (tid < O) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < WG_KT; ++k) v += Bs[k * BM + tid];
      unsafeAtomicAdd(a.dBias + (size_t)n * O + tid, v);
    }
  }
  float* dst = a.dW + ((size_t)n * S + ms) * a.I * O + ns * 64 + j;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#ifdef WGN_LAB_NO_ATOMIC
        if (acc[mt][nt][e] == 12345.678f)
#endif
          unsafeAtomicAdd(dst + (size_t)(16 * mt + 4 * kq + e) * O + 16 * nt, acc[mt][nt][e]);
}

// ---- gradient of the learned support (round 3): dT[n][m] += sum_r sum_i dA[r][n][i] * U[r][m][i] -------------------
// Both operands are [rows][nodes][64] blocks with the reduction index i CONTIGUOUS (a node's 256-byte row): for the
// 16x16x4 fp32 MFMA a lane's float4 of its row IS four reduction steps (the node kernels' operand trick), for the A and
// for the B side alike - nothing has to be transposed on the way.  The generic GEMM turned both tiles into k-major LDS
// images first (scalar LDS stores, 51 % MFMA busy, 470 us for 32 GFLOP); here a 64-node x 32-i half of a row block goes
// from global memory to LDS as whole float4s (16-byte slots XOR-swizzled by the row) and comes back as ds_read_b128
// fragments: 32 MFMAs per wave and stage behind 8 LDS reads.  64 x 64 output tiles, 2 x 2 MFMA tiles per wave; the
// 32-row halves of the last tile that lie beyond the padded node count are skipped (403 nodes: 6.5 x 6.5 tiles of work
// instead of 7 x 7); split over the row blocks, fp32 atomics into dT.
struct AdjGradArgs {
  const float* A;        // dA + slot offset: row block r at A + r * aStride, node n at + n * 64
  const float* B;        // U: row block r at B + r * bStride
  long aStride, bStride;
  int R, N, Np;          // row blocks, nodes, padded nodes (rows of a block that exist)
  float* dT;             // [N][N]
};

__global__ __launch_bounds__(256) void k_adj_grad(AdjGradArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][64 * 32];
  __shared__ __attribute__((aligned(16))) float Bs[2][64 * 32];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const int wr = w >> 1, wc = w & 1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int per = (g.R + gridDim.z - 1) / gridDim.z;
  const int r0 = blockIdx.z * per, r1 = min(r0 + per, g.R);
  if (r0 >= r1) return;
  const bool live = m0 + wr * 32 < g.Np && n0 + wc * 32 < g.Np;     // wave-uniform
  // staging: thread -> rows (tid >> 3) and (tid >> 3) + 32 of the tile, 16-byte slot tid & 7 of the 32-i half
  const int srow = tid >> 3, sq = tid & 7;
  const size_t aOff0 = (size_t)min(m0 + srow, g.Np - 1) * 64 + sq * 4, aOff1 = (size_t)min(m0 + srow + 32, g.Np - 1) * 64 + sq * 4;
  const size_t bOff0 = (size_t)min(n0 + srow, g.Np - 1) * 64 + sq * 4, bOff1 = (size_t)min(n0 + srow + 32, g.Np - 1) * 64 + sq * 4;
  const int st0 = (srow * 8 + (sq ^ (srow & 7))) * 4, st1 = ((srow + 32) * 8 + (sq ^ (srow & 7))) * 4;
  const int stages = 2 * (r1 - r0);
  struct Stage { float4 a0, a1, b0, b1; };
  auto fetch = [&](int st) {
    const int sc = min(st, stages - 1);
    const size_t r = (size_t)(r0 + (sc >> 1)), h = (size_t)(sc & 1) * 32;
    Stage v;
    v.a0 = *reinterpret_cast<const float4*>(g.A + r * g.aStride + aOff0 + h);
    v.a1 = *reinterpret_cast<const float4*>(g.A + r * g.aStride + aOff1 + h);
    v.b0 = *reinterpret_cast<const float4*>(g.B + r * g.bStride + bOff0 + h);
    v.b1 = *reinterpret_cast<const float4*>(g.B + r * g.bStride + bOff1 + h);
    return v;
  };
  auto stash = [&](int buf, const Stage& v) {
    *reinterpret_cast<float4*>(&As[buf][st0]) = v.a0;
    *reinterpret_cast<float4*>(&As[buf][st1]) = v.a1;
    *reinterpret_cast<float4*>(&Bs[buf][st0]) = v.b0;
    *reinterpret_cast<float4*>(&Bs[buf][st1]) = v.b1;
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  Stage cur = fetch(0);
  stash(0, cur);
  cur = fetch(1);         // stage 1 waits in `cur`, stage 2 in `nxt`
  Stage nxt = fetch(2);
  __syncthreads();
  // fragment rows of this wave: A rows wr*32 + 16 p + j, B rows wc*32 + 16 q + j; slot 4 gq + kq, swizzled by the row
  const int ra0 = wr * 32 + j, ra1 = ra0 + 16, rb0 = wc * 32 + j, rb1 = rb0 + 16;
  for (int st = 0; st < stages; ++st) {
    const int buf = st & 1;
    // stage st + 1 into the other buffer (its last readers passed the barrier at the end of stage st - 1)
    stash(buf ^ 1, cur);
    cur = nxt;
    nxt = fetch(st + 3);
    if (live) {
#pragma unroll
      for (int gq = 0; gq < 2; ++gq) {
        const int sl = 4 * gq + kq;
        const float4 a0 = *reinterpret_cast<const float4*>(&As[buf][(ra0 * 8 + (sl ^ (ra0 & 7))) * 4]);
        const float4 a1 = *reinterpret_cast<const float4*>(&As[buf][(ra1 * 8 + (sl ^ (ra1 & 7))) * 4]);
        const float4 b0 = *reinterpret_cast<const float4*>(&Bs[buf][(rb0 * 8 + (sl ^ (rb0 & 7))) * 4]);
        const float4 b1 = *reinterpret_cast<const float4*>(&Bs[buf][(rb1 * 8 + (sl ^ (rb1 & 7))) * 4]);
        acc[0][0] = MFMA16(a0.x, b0.x, acc[0][0]); acc[0][1] = MFMA16(a0.x, b1.x, acc[0][1]);
        acc[1][0] = MFMA16(a1.x, b0.x, acc[1][0]); acc[1][1] = MFMA16(a1.x, b1.x, acc[1][1]);
        acc[0][0] = MFMA16(a0.y, b0.y, acc[0][0]); acc[0][1] = MFMA16(a0.y, b1.y, acc[0][1]);
        acc[1][0] = MFMA16(a1.y, b0.y, acc[1][0]); acc[1][1] = MFMA16(a1.y, b1.y, acc[1][1]);
        acc[0][0] = MFMA16(a0.z, b0.z, acc[0][0]); acc[0][1] = MFMA16(a0.z, b1.z, acc[0][1]);
        acc[1][0] = MFMA16(a1.z, b0.z, acc[1][0]); acc[1][1] = MFMA16(a1.z, b1.z, acc[1][1]);
        acc[0][0] = MFMA16(a0.w, b0.w, acc[0][0]); acc[0][1] = MFMA16(a0.w, b1.w, acc[0][1]);
        acc[1][0] = MFMA16(a1.w, b0.w, acc[1][0]); acc[1][1] = MFMA16(a1.w, b1.w, acc[1][1]);
      }
    }
    __syncthreads();
  }
  if (!live) return;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wr * 32 + 16 * p + 4 * kq + e, n = n0 + wc * 32 + 16 * q + j;
        if (m < g.N && n < g.N) unsafeAtomicAdd(g.dT + (size_t)m * g.N + n, acc[p][q][e]);
      }
}

// the same gradient for layer 0's NARROW x part with two input channels: dT[n][m] += sum_{r,c} dA[n][r][c] * x[r][m][c],
// dA node-major (a node's rows*2 values contiguous: the A operand's float4 is four reduction steps), x time-major
// [r][Np][2] (the four steps of a lane are the two channels of two consecutive rows: two float2).  1 GFLOP: the generic
// GEMM (K = 2 per batch item, 1 536 batch items) took 154 us for it; 32 x 32 output tiles per wave, K cut over waves and
// workgroups, fp32 atomics.
__global__ __launch_bounds__(256) void k_adj_grad_narrow2(const float* __restrict__ dA, const float* __restrict__ x,
                                                          int rows, int N, int Np, int splits, float* __restrict__ dT) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.y * 32, m0 = blockIdx.x * 32;
  const int K = rows * 2, groups = K >> 4, parts = splits * 4, part = blockIdx.z * 4 + w;
  const int per = (groups + parts - 1) / parts, g0 = part * per, g1 = min(g0 + per, groups);
  const float* a0 = dA + (size_t)min(n0 + j, N - 1) * K + 4 * kq;
  const float* a1 = dA + (size_t)min(n0 + 16 + j, N - 1) * K + 4 * kq;
  const size_t bm0 = (size_t)min(m0 + j, N - 1) * 2, bm1 = (size_t)min(m0 + 16 + j, N - 1) * 2;
  f32x4 acc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int g = g0; g < g1; ++g) {
    const float4 av0 = *reinterpret_cast<const float4*>(a0 + 16 * g), av1 = *reinterpret_cast<const float4*>(a1 + 16 * g);
    const size_t r = (size_t)(8 * g + 2 * kq) * Np * 2;
    const float2 b00 = *reinterpret_cast<const float2*>(x + r + bm0), b01 = *reinterpret_cast<const float2*>(x + r + (size_t)Np * 2 + bm0);
    const float2 b10 = *reinterpret_cast<const float2*>(x + r + bm1), b11 = *reinterpret_cast<const float2*>(x + r + (size_t)Np * 2 + bm1);
    acc[0][0] = MFMA16(av0.x, b00.x, acc[0][0]); acc[0][1] = MFMA16(av0.x, b10.x, acc[0][1]);
    acc[1][0] = MFMA16(av1.x, b00.x, acc[1][0]); acc[1][1] = MFMA16(av1.x, b10.x, acc[1][1]);
    acc[0][0] = MFMA16(av0.y, b00.y, acc[0][0]); acc[0][1] = MFMA16(av0.y, b10.y, acc[0][1]);
    acc[1][0] = MFMA16(av1.y, b00.y, acc[1][0]); acc[1][1] = MFMA16(av1.y, b10.y, acc[1][1]);
    acc[0][0] = MFMA16(av0.z, b01.x, acc[0][0]); acc[0][1] = MFMA16(av0.z, b11.x, acc[0][1]);
    acc[1][0] = MFMA16(av1.z, b01.x, acc[1][0]); acc[1][1] = MFMA16(av1.z, b11.x, acc[1][1]);
    acc[0][0] = MFMA16(av0.w, b01.y, acc[0][0]); acc[0][1] = MFMA16(av0.w, b11.y, acc[0][1]);
    acc[1][0] = MFMA16(av1.w, b01.y, acc[1][0]); acc[1][1] = MFMA16(av1.w, b11.y, acc[1][1]);
  }
  if (g0 >= g1) return;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + 16 * p + 4 * kq + e, m = m0 + 16 * q + j;
        if (n < N && m < N) unsafeAtomicAdd(dT + (size_t)n * N + m, acc[p][q][e]);
      }
}

// ---- node-adaptive weight gradients of layer 0's NARROW x rows (C0 = 2..16 input channels) ------------------------
//   dWpG[n][s][c][o] += sum_rows XA[rows][n][s][c] * dpg[rows][n][o]      (o < 128; dWpU with dpu alike)
// XA slot 0 = the input rows themselves (time-major x0), slots 1.. = the fold's plain matrix MX0 [(k, n)][ld] with
// column (b*T + t)*C0 + c.  As a GEMM this has M = C0: a 64 x 64 tile is 97 % padding (the generic kernel spent 1.7 ms
// on it, at the exposed tail of the backward).  One thread per (node, output column o of gate | update (192), row
// stream), S*C0 accumulators each; the pre-activation gradients stream through once, coalesced.
#define WN_MAXACC 36
#define WN_PARTS 8        // workgroups per node (blockIdx.y), each with WN_GROUPS row groups of 192 threads
#define WN_GROUPS 4
template <int C0, int S>
__global__ __launch_bounds__(192 * WN_GROUPS) void k_wgrad_narrow(const float* __restrict__ x0tm, const float* __restrict__ mx0,
                                                                  long ld, const float* __restrict__ dpg,
                                                                  const float* __restrict__ dpu, float* __restrict__ dWpG,
                                                                  float* __restrict__ dWpU, int T, int B, int N, int Np, int I) {
  static_assert(S * C0 <= WN_MAXACC, "accumulators live in registers");
  __shared__ float part[WN_GROUPS - 1][S * C0][192];
  const int n = blockIdx.x, o = threadIdx.x % 192, rg = threadIdx.x / 192;
  const bool gate = o < 128;
  const int oc = gate ? o : o - 128, O = gate ? 128 : 64;
  const float* dp = gate ? dpg : dpu;
  float acc[S][C0];
#pragma unroll
  for (int sl = 0; sl < S; ++sl)
#pragma unroll
    for (int c = 0; c < C0; ++c) acc[sl][c] = 0.f;
  // the T*B rows are dealt out round robin over (workgroup part, row group): 32 independent streams per node keep
  // enough loads in flight (one stream per node - 1536 dependent iterations - took 1.06 ms)
  const int rows = T * B;
  for (int r = blockIdx.y * WN_GROUPS + rg; r < rows; r += WN_PARTS * WN_GROUPS) {
    const int t = r / B, b = r - t * B;
    const float d = dp[((size_t)r * Np + n) * O + oc];
    const float* xs = x0tm + ((size_t)r * Np + n) * C0;
    const float* ms = mx0 + (size_t)n * ld + ((size_t)b * T + t) * C0;
#pragma unroll
    for (int c = 0; c < C0; ++c) acc[0][c] = fmaf(xs[c], d, acc[0][c]);
#pragma unroll
    for (int sl = 1; sl < S; ++sl)
#pragma unroll
      for (int c = 0; c < C0; ++c) acc[sl][c] = fmaf(ms[(size_t)(sl - 1) * Np * ld + c], d, acc[sl][c]);
  }
  if (rg > 0) {
#pragma unroll
    for (int sl = 0; sl < S; ++sl)
#pragma unroll
      for (int c = 0; c < C0; ++c) part[rg - 1][sl * C0 + c][o] = acc[sl][c];
  }
  __syncthreads();
  if (rg > 0) return;
  float* dst = gate ? dWpG : dWpU;
#pragma unroll
  for (int sl = 0; sl < S; ++sl)
#pragma unroll
    for (int c = 0; c < C0; ++c) {
      float v = acc[sl][c];
#pragma unroll
      for (int q = 0; q < WN_GROUPS - 1; ++q) v += part[q][sl * C0 + c][o];
      unsafeAtomicAdd(&dst[(((size_t)n * S + sl) * I + c) * O + oc], v);
    }
}

// The same gradients on the matrix cores (round 3; S * C0 <= 16).  Per node a [S*C0 x rows] . [rows x 192] product:
// the (slot, channel) pairs are the 16 rows of ONE MFMA tile, the 192 gradient columns its 12 column tiles.  A lane's
// float4 of a gradient row is the same column of FOUR column tiles (column 64 og + 4 (l & 15) + ct), so a step of four
// rows is three 1 KB loads per wave feeding twelve MFMAs, eight steps in flight - bound by the 475 MB it streams instead
// of by 48 dependent iterations per thread (0.30 ms alone, 1.0 ms beside the adjacency gradients in the tail of the
// backward).  Waves = row streams (WNM_PARTS x 4 per node); the four of a workgroup meet in LDS, one atomic per output.
#define WNM_PARTS 8
template <int C0, int S>
__global__ __launch_bounds__(256) void k_wgrad_narrow_mfma(const float* __restrict__ x0tm, const float* __restrict__ mx0,
                                                           long ld, const float* __restrict__ dpg,
                                                           const float* __restrict__ dpu, float* __restrict__ dWpG,
                                                           float* __restrict__ dWpU, int T, int B, int N, int Np, int I) {
  static_assert(S * C0 <= 16, "one MFMA row tile");
  __shared__ float part[3][S * C0][192];
  const int n = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, kq = lane >> 4, j = lane & 15;
  const int rows = T * B, stream = blockIdx.y * 4 + w, streams = WNM_PARTS * 4;
  const int steps = (rows + 3) >> 2, per = (steps + streams - 1) / streams;
  const int st0 = stream * per, st1 = min(st0 + per, steps);
  const int sl = j / C0, ch = j - sl * C0;          // this lane's A row: (slot, channel)
  const bool aLive = j < S * C0;
  f32x4 acc[3][4];
#pragma unroll
  for (int og = 0; og < 3; ++og)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[og][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load = [&](int st, float4 (&bv)[3], float& av) {
    const int r = 4 * st + kq, rc = min(r, rows - 1);
    const size_t rowN = (size_t)rc * Np + n;
    bv[0] = *reinterpret_cast<const float4*>(dpg + rowN * 128 + 4 * j);
    bv[1] = *reinterpret_cast<const float4*>(dpg + rowN * 128 + 64 + 4 * j);
    bv[2] = *reinterpret_cast<const float4*>(dpu + rowN * 64 + 4 * j);
    const int t = rc / B, b = rc - t * B;
    const float* src = sl == 0 ? x0tm + rowN * C0 + ch
                               : mx0 + (size_t)(sl - 1) * Np * ld + (size_t)n * ld + ((size_t)b * T + t) * C0 + ch;
    const float v = aLive ? *src : 0.f;
    av = r < rows ? v : 0.f;
  };
  constexpr int DEPTH = 4;
  float4 bq[DEPTH][3];
  float aq[DEPTH];
  if (st0 < st1) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) load(min(st0 + u, st1 - 1), bq[u], aq[u]);
    for (int st = st0; st < st1; st += DEPTH) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const float4 b0 = bq[u][0], b1 = bq[u][1], b2 = bq[u][2];
        const float a = st + u < st1 ? aq[u] : 0.f;
        load(min(st + u + DEPTH, st1 - 1), bq[u], aq[u]);
        acc[0][0] = MFMA16(a, b0.x, acc[0][0]); acc[0][1] = MFMA16(a, b0.y, acc[0][1]);
        acc[0][2] = MFMA16(a, b0.z, acc[0][2]); acc[0][3] = MFMA16(a, b0.w, acc[0][3]);
        acc[1][0] = MFMA16(a, b1.x, acc[1][0]); acc[1][1] = MFMA16(a, b1.y, acc[1][1]);
        acc[1][2] = MFMA16(a, b1.z, acc[1][2]); acc[1][3] = MFMA16(a, b1.w, acc[1][3]);
        acc[2][0] = MFMA16(a, b2.x, acc[2][0]); acc[2][1] = MFMA16(a, b2.y, acc[2][1]);
        acc[2][2] = MFMA16(a, b2.z, acc[2][2]); acc[2][3] = MFMA16(a, b2.w, acc[2][3]);
      }
    }
  }
  // accumulator rows 4 kq + e = (slot, channel) pairs, column 64 og + 4 j + ct
  if (w > 0) {
#pragma unroll
    for (int og = 0; og < 3; ++og)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * kq + e < S * C0) part[w - 1][4 * kq + e][64 * og + 4 * j + ct] = acc[og][ct][e];
  }
  __syncthreads();
  if (w > 0) return;
#pragma unroll
  for (int og = 0; og < 3; ++og)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = 4 * kq + e;
        if (m >= S * C0) continue;
        const int o = 64 * og + 4 * j + ct;
        const float v = acc[og][ct][e] + part[0][m][o] + part[1][m][o] + part[2][m][o];
        const int s2 = m / C0, c2 = m - s2 * C0;
        if (og < 2) unsafeAtomicAdd(&dWpG[(((size_t)n * S + s2) * I + c2) * 128 + o], v);
        else unsafeAtomicAdd(&dWpU[(((size_t)n * S + s2) * I + c2) * 64 + (o - 128)], v);
      }
}

// ---- x-column gradients of layer 0's NARROW input (the transposed counterpart of k_wgrad_narrow) -----------------
//   dA[s][n][row][c] = sum_o dpg[row][n][o] WpG[n][s][c][o] + sum_o dpu[row][n][o] WpU[n][s][c][o]        (c < C0)
// node-major output (the transposed mix then is ONE GEMM with rows*C0 columns).  As a GEMM this has N = C0 output
// columns per slot: the generic kernel spent 2 x 0.58 ms on 97 % padding.  Here: one thread per (node, row), the S*C0
// weights of every o broadcast from LDS, the 192 pre-activation gradients of the row read once as float4.
template <int C0, int S>
__global__ __launch_bounds__(256) void k_xcol_narrow(const float* __restrict__ dpg, const float* __restrict__ dpu,
                                                     const float* __restrict__ WpG, const float* __restrict__ WpU,
                                                     float* __restrict__ dA, int rows, int N, int Np, int I) {
  constexpr int SC = S * C0, SCP = (SC + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float wl[192 * SCP];
  const int n = blockIdx.y, tid = threadIdx.x;
  for (int e = tid; e < 192 * SC; e += 256) {
    const int o = e % 192, sc = e / 192, sl = sc / C0, c = sc - sl * C0;
    const size_t wrow = ((size_t)n * S + sl) * I + c;
    wl[o * SCP + sc] = o < 128 ? WpG[wrow * 128 + o] : WpU[wrow * 64 + o - 128];
  }
  __syncthreads();
  const int row = blockIdx.x * 256 + tid;
  if (row >= rows) return;
  float acc[SC];
#pragma unroll
  for (int i = 0; i < SC; ++i) acc[i] = 0.f;
  const float4* g4 = reinterpret_cast<const float4*>(dpg + ((size_t)row * Np + n) * 128);
  const float4* u4 = reinterpret_cast<const float4*>(dpu + ((size_t)row * Np + n) * 64);
  auto fold = [&](float d, int o) {
#pragma unroll
    for (int i = 0; i < SC; ++i) acc[i] = fmaf(d, wl[o * SCP + i], acc[i]);
  };
#pragma unroll 4
  for (int q = 0; q < 48; ++q) {
    const float4 d = q < 32 ? g4[q] : u4[q - 32];
    fold(d.x, 4 * q); fold(d.y, 4 * q + 1); fold(d.z, 4 * q + 2); fold(d.w, 4 * q + 3);
  }
#pragma unroll
  for (int sl = 0; sl < S; ++sl)
#pragma unroll
    for (int c = 0; c < C0; ++c) dA[(((size_t)sl * Np + n) * rows + row) * C0 + c] = acc[sl * C0 + c];
}

// ---- x columns of the residual cell for a NARROW input (layer 0) ---------------------------------------------------
//   dX[row][c] += sum_o dpu2[row][o] RU[o][c] + sum_o dpg2[row][o] RG[o][c]        (c < C0; RU (64, I), RG (128, I))
// As GEMMs these have N = C0 output columns (two launches of the generic kernel, 0.40 ms of 97 % padding on the main
// stream, in the tail of the backward).  Here the 192 pre-activation gradients of a row stream through once, coalesced:
// 16 lanes share a row (three float4 each), every lane keeps its 12 x C0 weights in registers, the 16 partial sums
// meet in a butterfly.  Bound by the 768 bytes per row it reads.
template <int C0>
__global__ __launch_bounds__(256) void k_res_xcol_narrow(const float* __restrict__ dpu2, const float* __restrict__ dpg2,
                                                         const float* __restrict__ RU, const float* __restrict__ RG, int I,
                                                         float* __restrict__ dX, long rows) {
  const int lane = threadIdx.x & 63, sub = lane & 15, rw = lane >> 4;
  float wu[4][C0], wg[8][C0];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int c = 0; c < C0; ++c) {
      wu[e][c] = RU[(size_t)(4 * sub + e) * I + c];
      wg[e][c] = RG[(size_t)(4 * sub + e) * I + c];
      wg[4 + e][c] = RG[(size_t)(64 + 4 * sub + e) * I + c];
    }
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (long)gridDim.x * 4;
  for (long r0 = wave * 4; r0 < rows; r0 += waves * 4) {
    const long row = min(r0 + rw, rows - 1);
    const float4 u = *reinterpret_cast<const float4*>(dpu2 + row * 64 + 4 * sub);
    const float4 ga = *reinterpret_cast<const float4*>(dpg2 + row * 128 + 4 * sub);
    const float4 gb = *reinterpret_cast<const float4*>(dpg2 + row * 128 + 64 + 4 * sub);
    float acc[C0];
#pragma unroll
    for (int c = 0; c < C0; ++c) {
      float v = u.x * wu[0][c];
      v = fmaf(u.y, wu[1][c], v); v = fmaf(u.z, wu[2][c], v); v = fmaf(u.w, wu[3][c], v);
      v = fmaf(ga.x, wg[0][c], v); v = fmaf(ga.y, wg[1][c], v); v = fmaf(ga.z, wg[2][c], v); v = fmaf(ga.w, wg[3][c], v);
      v = fmaf(gb.x, wg[4][c], v); v = fmaf(gb.y, wg[5][c], v); v = fmaf(gb.z, wg[6][c], v); v = fmaf(gb.w, wg[7][c], v);
      acc[c] = v;
    }
#pragma unroll
    for (int c = 0; c < C0; ++c) {
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) acc[c] += __shfl_xor(acc[c], m, 16);
    }
    if (sub == 0 && r0 + rw < rows) {
#pragma unroll
      for (int c = 0; c < C0; ++c) dX[(r0 + rw) * C0 + c] += acc[c];
    }
  }
}

// ---- residual nn.Linear weight gradients of a NARROW input's columns (layer 0) ---------------------------------------
//   dRG[o][c] += sum_rows dpg2[row][o] x[row][c]   (o < 128),   dRU[o][c] += sum_rows dpu2[row][o] x[row][c]   (o < 64)
// As GEMMs: N = C0 columns, 2 x 0.25 ms of the generic kernel.  The lane map of k_res_xcol_narrow: 16 lanes share a row,
// 12 gradient columns per lane, C0 accumulators each; the partial sums of a workgroup meet in LDS and leave as one
// atomic add per (o, c).
template <int C0>
__global__ __launch_bounds__(256) void k_res_wgrad_narrow(const float* __restrict__ dpu2, const float* __restrict__ dpg2,
                                                          const float* __restrict__ x, int I, float* __restrict__ dRU,
                                                          float* __restrict__ dRG, long rows) {
  __shared__ float part[4][192 * C0];
  const int lane = threadIdx.x & 63, sub = lane & 15, rw = lane >> 4, w = threadIdx.x >> 6;
  float acc[12][C0];
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int c = 0; c < C0; ++c) acc[i][c] = 0.f;
  const long wave = (long)blockIdx.x * 4 + w, waves = (long)gridDim.x * 4;
  for (long r0 = wave * 4; r0 < rows; r0 += waves * 4) {
    const long row = min(r0 + rw, rows - 1);
    const float live = r0 + rw < rows ? 1.f : 0.f;
    const float4 u = *reinterpret_cast<const float4*>(dpu2 + row * 64 + 4 * sub);
    const float4 ga = *reinterpret_cast<const float4*>(dpg2 + row * 128 + 4 * sub);
    const float4 gb = *reinterpret_cast<const float4*>(dpg2 + row * 128 + 64 + 4 * sub);
    const float d[12] = {u.x, u.y, u.z, u.w, ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
#pragma unroll
    for (int c = 0; c < C0; ++c) {
      const float xv = x[row * C0 + c] * live;
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i][c] = fmaf(d[i], xv, acc[i][c]);
    }
  }
  // the four rows of a wave, then the four waves
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int c = 0; c < C0; ++c) {
      float v = acc[i][c];
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      acc[i][c] = v;
    }
  // slot of gradient column i of this lane: 0..63 dpu2 column 4 sub + e, 64..191 dpg2 column
  if (rw == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int o = i < 4 ? 4 * sub + i : (i < 8 ? 64 + 4 * sub + (i - 4) : 128 + 4 * sub + (i - 8));
#pragma unroll
      for (int c = 0; c < C0; ++c) part[w][o * C0 + c] = acc[i][c];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 192 * C0; e += 256) {
    const float v = part[0][e] + part[1][e] + part[2][e] + part[3][e];
    const int o = e / C0, c = e - o * C0;
    if (o < 64) unsafeAtomicAdd(dRU + (size_t)o * I + c, v);
    else unsafeAtomicAdd(dRG + (size_t)(o - 64) * I + c, v);
  }
}

// The same x-column gradients on the matrix cores (round 3; S * C0 <= 16): per node a [rows x 192] . [192 x S*C0] product.
// Both operands lie K-contiguous (a gradient row's 192 columns; a weight row's 192 columns), so a lane's float4 is four
// reduction steps on either side; the (slot, channel) pairs are the 16 columns of ONE MFMA tile, the node's 12 weight
// fragments stay in registers, and a wave walks 16-row tiles with the next tile's 12 row pieces in flight.  The VALU
// kernel above reads the same rows 16 bytes per lane and instruction (189 us).
template <int C0, int S>
__global__ __launch_bounds__(256) void k_xcol_narrow_mfma(const float* __restrict__ dpg, const float* __restrict__ dpu,
                                                          const float* __restrict__ WpG, const float* __restrict__ WpU,
                                                          float* __restrict__ dA, int rows, int N, int Np, int I,
                                                          int tilesPerWave) {
  static_assert(S * C0 <= 16, "one MFMA column tile");
  const int n = blockIdx.y, lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
  const int nTiles = (rows + 15) >> 4;
  const int tile0 = (blockIdx.x * 4 + w) * tilesPerWave, tile1 = min(tile0 + tilesPerWave, nTiles);
  if (tile0 >= tile1) return;
  // B fragments: column j = (slot, channel), groups 0..7 the gate AGCN's 128 columns, 8..11 the update AGCN's 64
  const int sl = min(j / C0, S - 1), ch = j - (j / C0) * C0;
  const bool colLive = j < S * C0;
  const size_t wrow = ((size_t)n * S + sl) * I + ch;
  float4 bf[12];
#pragma unroll
  for (int g = 0; g < 12; ++g) {
    const float4 v = g < 8 ? *reinterpret_cast<const float4*>(WpG + wrow * 128 + 16 * g + 4 * kq)
                           : *reinterpret_cast<const float4*>(WpU + wrow * 64 + 16 * (g - 8) + 4 * kq);
    bf[g] = colLive ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto load = [&](int tile, float4 (&a)[12]) {
    const size_t rowN = (size_t)min(tile * 16 + j, rows - 1) * Np + n;
#pragma unroll
    for (int g = 0; g < 8; ++g) a[g] = *reinterpret_cast<const float4*>(dpg + rowN * 128 + 16 * g + 4 * kq);
#pragma unroll
    for (int g = 8; g < 12; ++g) a[g] = *reinterpret_cast<const float4*>(dpu + rowN * 64 + 16 * (g - 8) + 4 * kq);
  };
  float4 cur[12], nxt[12];
  load(tile0, cur);
  for (int tile = tile0; tile < tile1; ++tile) {
    load(min(tile + 1, tile1 - 1), nxt);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      acc = MFMA16(cur[g].x, bf[g].x, acc);
      acc = MFMA16(cur[g].y, bf[g].y, acc);
      acc = MFMA16(cur[g].z, bf[g].z, acc);
      acc = MFMA16(cur[g].w, bf[g].w, acc);
    }
    if (colLive) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = tile * 16 + 4 * kq + e;
        if (row < rows) dA[(((size_t)sl * Np + n) * rows + row) * C0 + ch] = acc[e];
      }
    }
#pragma unroll
    for (int g = 0; g < 12; ++g) cur[g] = nxt[g];
  }
}

// k_res_xcol_narrow and k_res_wgrad_narrow in ONE pass over the residual cell's gradients (two input channels: the 24
// weights and the 24 accumulators of a lane fit the registers): dX += [dpu2 | dpg2] . [RU | RG][:, 0:2] and
// dRU / dRG[:, 0:2] += [dpu2 | dpg2]^T . x.  The caller has cleared dRU / dRG (whole tensors) on this stream.
__global__ __launch_bounds__(256) void k_res_narrow2(const float* __restrict__ dpu2, const float* __restrict__ dpg2,
                                                     const float* __restrict__ RU, const float* __restrict__ RG,
                                                     const float* __restrict__ x, int I, float* __restrict__ dX,
                                                     float* __restrict__ dRU, float* __restrict__ dRG, long rows) {
  __shared__ float part[4][192 * 2];
  const int lane = threadIdx.x & 63, sub = lane & 15, rw = lane >> 4, w = threadIdx.x >> 6;
  float wt[12][2], acc[12][2];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      wt[e][c] = RU[(size_t)(4 * sub + e) * I + c];
      wt[4 + e][c] = RG[(size_t)(4 * sub + e) * I + c];
      wt[8 + e][c] = RG[(size_t)(64 + 4 * sub + e) * I + c];
    }
#pragma unroll
  for (int i = 0; i < 12; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; }
  const long wave = (long)blockIdx.x * 4 + w, waves = (long)gridDim.x * 4;
  for (long r0 = wave * 4; r0 < rows; r0 += waves * 4) {
    const long row = min(r0 + rw, rows - 1);
    const bool liveRow = r0 + rw < rows;
    const float4 u = *reinterpret_cast<const float4*>(dpu2 + row * 64 + 4 * sub);
    const float4 ga = *reinterpret_cast<const float4*>(dpg2 + row * 128 + 4 * sub);
    const float4 gb = *reinterpret_cast<const float4*>(dpg2 + row * 128 + 64 + 4 * sub);
    const float2 xv = *reinterpret_cast<const float2*>(x + row * 2);
    const float d[12] = {u.x, u.y, u.z, u.w, ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
    float s0 = 0.f, s1 = 0.f;
    const float x0 = liveRow ? xv.x : 0.f, x1 = liveRow ? xv.y : 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      s0 = fmaf(d[i], wt[i][0], s0); s1 = fmaf(d[i], wt[i][1], s1);
      acc[i][0] = fmaf(d[i], x0, acc[i][0]); acc[i][1] = fmaf(d[i], x1, acc[i][1]);
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) { s0 += __shfl_xor(s0, m, 16); s1 += __shfl_xor(s1, m, 16); }
    if (sub == 0 && liveRow) {
      float2* dst = reinterpret_cast<float2*>(dX + (r0 + rw) * 2);
      const float2 o = *dst;
      *dst = make_float2(o.x + s0, o.y + s1);
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float v = acc[i][c];
      v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
      acc[i][c] = v;
    }
  if (rw == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int o = i < 4 ? 4 * sub + i : (i < 8 ? 64 + 4 * sub + (i - 4) : 128 + 4 * sub + (i - 8));
      part[w][o * 2] = acc[i][0]; part[w][o * 2 + 1] = acc[i][1];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 192 * 2; e += 256) {
    const float v = part[0][e] + part[1][e] + part[2][e] + part[3][e];
    const int o = e >> 1, c = e & 1;
    if (o < 64) unsafeAtomicAdd(dRU + (size_t)o * I + c, v);
    else unsafeAtomicAdd(dRG + (size_t)(o - 64) * I + c, v);
  }
}

// ---- x columns of the residual cell for a 64-channel input (the layers above the first) ---------------------------
//   dX[row][0:64] += dpu2[row][0:64] . RU[:, 0:64] + dpg2[row][0:128] . RG[:, 0:64]          (RU (64, I), RG (128, I))
// Two launches of the generic GEMM per x-column chunk (K-contiguous A operands: its scalar staging path, 46 us each, ten
// a step) re-read the residual cell's gradients twice and read-modify-wrote dX twice.  One pass: a 32-row tile of both
// gradients goes to LDS as whole rows (16-byte slots XOR-swizzled by the row), wave w contracts it with its 16 output
// columns - the 192 x 16 weight fragments stay in registers, a lane's float4 of a row is four reduction steps - and the
// 32 x 64 result is turned through LDS into float4 read-modify-writes of dX.  The next tile's rows are in flight meanwhile.
__global__ __launch_bounds__(256) void k_res_xcol64(const float* __restrict__ dpu2, const float* __restrict__ dpg2,
                                                    const float* __restrict__ RU, const float* __restrict__ RG, int I,
                                                    float* __restrict__ dX, long rows) {
  __shared__ __attribute__((aligned(16))) float As[32 * 192];
  __shared__ __attribute__((aligned(16))) float Out[32 * 68];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  float4 bf[12];
#pragma unroll
  for (int g = 0; g < 12; ++g) {
    const float* W = g < 4 ? RU + (size_t)(16 * g + 4 * kq) * I : RG + (size_t)(16 * (g - 4) + 4 * kq) * I;
    bf[g] = make_float4(W[16 * w + j], W[(size_t)I + 16 * w + j], W[2 * (size_t)I + 16 * w + j], W[3 * (size_t)I + 16 * w + j]);
  }
  const long tiles = (rows + 31) >> 5;
  struct Six { float4 v0, v1, v2, v3, v4, v5; };
  auto one = [&](long tile, int q) {
    const int u = tid + 256 * q, r = u / 48, slot = u - r * 48;
    const long row = min(tile * 32 + r, rows - 1);
    return slot < 16 ? *reinterpret_cast<const float4*>(dpu2 + row * 64 + 4 * slot)
                     : *reinterpret_cast<const float4*>(dpg2 + row * 128 + 4 * (slot - 16));
  };
  auto fetch = [&](long tile) {
    Six s6;
    s6.v0 = one(tile, 0); s6.v1 = one(tile, 1); s6.v2 = one(tile, 2); s6.v3 = one(tile, 3); s6.v4 = one(tile, 4); s6.v5 = one(tile, 5);
    return s6;
  };
  auto put = [&](int q, const float4& v) {
    const int u = tid + 256 * q, r = u / 48, slot = u - r * 48;
    *reinterpret_cast<float4*>(&As[(r * 48 + ((slot & ~7) | ((slot ^ r) & 7))) * 4]) = v;
  };
  Six cur = fetch(min((long)blockIdx.x, tiles - 1)), nxt;
  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    put(0, cur.v0); put(1, cur.v1); put(2, cur.v2); put(3, cur.v3); put(4, cur.v4); put(5, cur.v5);
    __syncthreads();
    nxt = fetch(min(tile + (long)gridDim.x, tiles - 1));
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      const int slot = 4 * g + kq;
      const float4 a0 = *reinterpret_cast<const float4*>(&As[(j * 48 + ((slot & ~7) | ((slot ^ j) & 7))) * 4]);
      const float4 a1 = *reinterpret_cast<const float4*>(&As[((16 + j) * 48 + ((slot & ~7) | ((slot ^ (16 + j)) & 7))) * 4]);
      acc[0] = MFMA16(a0.x, bf[g].x, acc[0]); acc[1] = MFMA16(a1.x, bf[g].x, acc[1]);
      acc[0] = MFMA16(a0.y, bf[g].y, acc[0]); acc[1] = MFMA16(a1.y, bf[g].y, acc[1]);
      acc[0] = MFMA16(a0.z, bf[g].z, acc[0]); acc[1] = MFMA16(a1.z, bf[g].z, acc[1]);
      acc[0] = MFMA16(a0.w, bf[g].w, acc[0]); acc[1] = MFMA16(a1.w, bf[g].w, acc[1]);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) Out[(16 * p + 4 * kq + e) * 68 + 16 * w + j] = acc[p][e];
    __syncthreads();
    {
      const int r = tid >> 3, c8 = (tid & 7) * 8;
      const long row = tile * 32 + r;
      if (row < rows) {
        float4* dst = reinterpret_cast<float4*>(dX + row * 64 + c8);
        const float4 o0 = *reinterpret_cast<const float4*>(&Out[r * 68 + c8]), o1 = *reinterpret_cast<const float4*>(&Out[r * 68 + c8 + 4]);
        const float4 d0 = dst[0], d1 = dst[1];
        dst[0] = make_float4(d0.x + o0.x, d0.y + o0.y, d0.z + o0.z, d0.w + o0.w);
        dst[1] = make_float4(d1.x + o1.x, d1.y + o1.y, d1.z + o1.z, d1.w + o1.w);
      }
    }
    cur = nxt;
  }
}

// plain copy of the support stack for the transposed graph mix: StP[kk][m] = St[m][kk] (m < N, zero beyond), i.e. row
// kk = k*Np + n holds S_k[n][.] - the A operand of k_mix when the reduction runs over (k, n)
__global__ __launch_bounds__(256) void k_stack_plain(const float* __restrict__ St, int ldS, int N, int rowsKK, int ldP,
                                                     float* __restrict__ StP) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx: kk block, by: m block
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int m = by + j, kk = bx + tx;
    tile[j][tx] = (m < N && kk < rowsKK) ? St[(size_t)m * ldS + kk] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int kk = bx + j, m = by + tx;
    if (kk < rowsKK && m < ldP) StP[(size_t)kk * ldP + m] = tile[tx][j];
  }
}

// ---- small helpers -------------------------------------------------------------------------------------------
// dst[r][n][c] += src[r][slot 0][n][c]   (src rows hold S slabs of Np*C)
__global__ __launch_bounds__(256) void k_add_slot0(float* __restrict__ dst, const float* __restrict__ src, size_t rows,
                                                   int Np, int C, int S) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t per = (size_t)Np * C;
  if (idx >= rows * per) return;
  const size_t r = idx / per, q = idx - r * per;
  dst[idx] += src[r * S * per + q];
}

// narrow x columns (layer 0): dst[row][n][c] = mix[n][row*C + c] + slot0[n][row*C + c]  (node-major -> row-major)
__global__ __launch_bounds__(256) void k_narrow_gather(const float* __restrict__ mix, const float* __restrict__ slot0,
                                                       float* __restrict__ dst, size_t rows, int N, int Np, int C) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * Np * C) return;
  const int c = idx % C;
  const int n = (idx / C) % Np;
  const size_t row = idx / ((size_t)C * Np);
  const size_t src = ((size_t)n * rows + row) * C + c;
  dst[idx] = n < N ? (mix ? mix[src] : 0.f) + slot0[src] : 0.f;
}

// dst += alpha * src
__global__ __launch_bounds__(256) void k_axpy(float* __restrict__ dst, const float* __restrict__ src, float alpha,
                                              size_t n) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < n) dst[idx] += alpha * src[idx];
}

// out[t][b][n][c] = a[t][b][n][c] * b[t][b][n][c]  (z * h_{t-1} for all steps)
__global__ __launch_bounds__(256) void k_mul(const float* __restrict__ x, const float* __restrict__ y,
                                             float* __restrict__ out, size_t n) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < n) out[idx] = x[idx] * y[idx];
}

// ha = r h + (1-r) hc and z2*ha for all steps (inputs of the residual cell's weight gradients)
__global__ __launch_bounds__(256) void k_ha_all(const float* __restrict__ r, const float* __restrict__ hprev,
                                                const float* __restrict__ hc, const float* __restrict__ z2,
                                                float* __restrict__ ha, float* __restrict__ z2ha, size_t n) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const float v = r[idx] * hprev[idx] + (1.f - r[idx]) * hc[idx];
  ha[idx] = v;
  z2ha[idx] = z2[idx] * v;
}

// x0p [B][T][Np][C] (batch-major) -> time-major [T][B][Np][C]
__global__ __launch_bounds__(256) void k_x0_time_major(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                       int T, int Np, int C) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t per = (size_t)Np * C;
  if (idx >= (size_t)T * B * per) return;
  const size_t q = idx % per;
  const int b = (idx / per) % B;
  const int t = idx / (per * B);
  dst[idx] = src[((size_t)b * T + t) * per + q];
}

// column sums over the rows of a [rows][Np][O] tensor, per node: out[n][o] += sum_rows src[row][n][o]; the rows are
// split over gridDim.y workgroups that meet in `out` with atomics (out holds its initial value beforehand)
__global__ __launch_bounds__(256) void k_node_colsum(const float* __restrict__ src, size_t rows, int N, int Np, int O,
                                                     float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * O) return;
  const int n = idx / O, o = idx - n * O;
  const size_t per = (rows + gridDim.y - 1) / gridDim.y;
  const size_t r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
  float s = 0.f;
  for (size_t r = r0; r < r1; ++r) s += src[(r * Np + n) * O + o];
  unsafeAtomicAdd(&out[idx], s);
}

// column sums over rows AND nodes: out[o] += sum_{row, n < N} src[row][n][o]   (nn.Linear / Conv2d bias gradients).
// A workgroup walks items (row, n) with 256/Opad of them in flight, lanes along o (coalesced); partial sums meet in
// LDS and then in `out` with one atomic per column and workgroup (out holds its initial value beforehand).
__global__ __launch_bounds__(256) void k_colsum_all(const float* __restrict__ src, size_t rows, int N, int Np, int O,
                                                    int Opad, float* __restrict__ out) {
  __shared__ float red[256];
  const int o = threadIdx.x % Opad, item = threadIdx.x / Opad, per = 256 / Opad;
  const size_t total = rows * N;
  float s = 0.f;
  if (o < O)
    for (size_t q = (size_t)blockIdx.x * per + item; q < total; q += (size_t)gridDim.x * per) {
      const size_t r = q / N, n = q - r * N;
      s += src[(r * Np + n) * O + o];
    }
  red[threadIdx.x] = s;
  __syncthreads();
  if (item == 0 && o < O) {
    for (int q = 1; q < per; ++q) s += red[q * Opad + o];
    unsafeAtomicAdd(&out[o], s);
  }
}

// ---- dropout in front of the head (MultiATGCN.py:416, training mode): the mask comes from the caller's RNG as a
// (B, T, N, 64) tensor of 0 / 1/(1-p);  dst[t][b][n][h] = src[t][b][n][h] * mask[b][t][n][h]  (dst may alias src)
__global__ __launch_bounds__(256) void k_apply_mask(const float* __restrict__ src, const float* __restrict__ mask,
                                                    float* __restrict__ dst, int B, int T, int N, int Np) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)T * B * Np * 64) return;
  const int h = idx & 63;
  const int n = (idx >> 6) % Np;
  const int b = (idx / ((size_t)64 * Np)) % B;
  const int t = idx / ((size_t)64 * Np * B);
  dst[idx] = n < N ? src[idx] * mask[(((size_t)b * T + t) * N + n) * 64 + h] : 0.f;
}

// ---- output head backward (MultiATGCN.py:416-418): dOut (B, out, N, od) -> plain [B][Np][CH] with oc = o*od + d
__global__ __launch_bounds__(256) void k_dout_rows(const float* __restrict__ dout, float* __restrict__ dst, int B,
                                                   int outSteps, int N, int Np, int od) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int CH = outSteps * od;
  if (idx >= (size_t)B * Np * CH) return;
  const int oc = idx % CH;
  const int n = (idx / CH) % Np;
  const size_t b = idx / ((size_t)CH * Np);
  const int o = oc / od, d = oc - o * od;
  dst[idx] = n < N ? dout[((b * outSteps + o) * N + n) * od + d] : 0.f;
}

// ---- head fusion backward (MultiATGCN.py:365-402) -----------------------------------------------------------------
// dx0 time-major [T][B][Np][C0]; one thread per (head, t, n, c): reduces over the batch
//   dweight_ts[h][t][n][c] = g_h * sum_b dx0 * X ;  dgain[h] += sum dx0 * X * weight_ts[h]
struct FuseBwdArgs {
  const float* X;        // windows (B, xSteps, N, F), or the raw series (steps, N, F) when labelStart != null
  const int* labelStart; // device (B) label starts, or null
  int rel[256];          // series mode: row offsets relative to the label start (MATGCN_MAX_XSTEPS)
  long seriesSteps;      // series mode: rows of the series (out-of-range rows are clamped and counted)
  const float* dx0;
  const float* tsg;
  const float* ts[8];
  float* dts[8];
  float* dgain;          // [nTs] accumulators (zeroed by the caller)
  int B, T, N, Np, C0, od, F, xSteps, startDim, nHeads, nTs;
  int headBegin[8];
};
__global__ __launch_bounds__(256) void k_fuse_heads_bwd(FuseBwdArgs a) {
  __shared__ float red[256];
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int h = blockIdx.y;
  const size_t per = (size_t)a.T * a.N * a.od;
  float gpart = 0.f;
  if (idx < per) {
    const int c = idx % a.od;
    const int n = (idx / a.od) % a.N;
    const int t = idx / ((size_t)a.od * a.N);
    float s = 0.f;
    for (int b = 0; b < a.B; ++b) {
      const size_t xrow = a.labelStart ? series_row((long)a.labelStart[b] + a.rel[a.headBegin[h] + t], a.seriesSteps)
                                       : (size_t)b * a.xSteps + a.headBegin[h] + t;
      const float xv = a.X[(xrow * a.N + n) * a.F + a.startDim + c];
      s = fmaf(a.dx0[(((size_t)t * a.B + b) * a.Np + n) * a.C0 + c], xv, s);
    }
    a.dts[h][idx] = stack_gain(a.tsg, a.nTs, h) * s;
    gpart = s * a.ts[h][idx];
  }
  red[threadIdx.x] = gpart;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) unsafeAtomicAdd(&a.dgain[h], red[0]);
}

#endif
