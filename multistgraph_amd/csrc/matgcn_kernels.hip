// matgcn_kernels.hip - hand-written gfx950 (CDNA4, wave64) kernels of the Multi-ATGCN forward path.
//
// Every kernel cites the reference lines (libcity/model/traffic_flow_prediction/MultiATGCN.py) whose
// arithmetic it implements.  All matrix contractions run on the exact-fp32 matrix cores:
//   v_mfma_f32_16x16x4_f32 (graph mix, node kernels): lane l supplies A[row = l&15][k = l>>4], B[k = l>>4][col = l&15];
//     its 4 accumulator registers hold C[row = 4*(l>>4) + e][col = l&15];
//   v_mfma_f32_32x32x2_f32 (output head): A[row = l&31][k = l>>5], B[k = l>>5][col = l&31];
//     its 16 accumulator registers hold C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
// This file: support stack, weight-stream preparation, layout helpers, head fusion, the graph-mix GEMM, the output
// head and the loss epilogue; the node-wise step kernels live in matgcn_node16.hip (same translation unit).
//
// Data layout (all fp32, Np = N rounded up to 16, H = 64; DESIGN.md section 3 has the full tables):
//   St   [Np][Mp]              transposed dense support stack, column k*Np+n holds S_k[n][.]   (mix A operand)
//   Hx   [rows][Np][64]        recurrent state / any per-row node features                    (mix B operand)
//   G    [N][rows][Ks][64]     graph-mixed features, node-major                                (node GEMM A operand)
//   W*   [N][K/16][OT][64][4]  node-adaptive weights in 16x16x4 B-fragment order               (node GEMM B operand)
//   PX   [T][N][B][192]        hoisted x-part of gate|update pre-activations (+bias), layers >= 1
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "matgcn_internal.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// =================================================================================================
// 1. support stack
// =================================================================================================
// Adaptive adjacency, one workgroup per row n (MultiATGCN.py:80-83):
//   A[n][m] = softmax_m(relu(sum_r E1[n][r] * E2[r][m]))      (unidirection)
//   A[n][m] = softmax_m(relu(sum_d E[n][d]  * E[m][d]))       (bidirection)
// Every logit is computed ONCE (kept in LDS: `lds` holds N floats) and the row is written contiguously into the plain
// matrix P[n][ldP]; k_static_transpose then moves it into its transposed stack slot with tiled, coalesced accesses
// (round 1 recomputed the logits in all three softmax passes and scattered 4-byte stores down a column of St).
__global__ __launch_bounds__(256) void k_adaptive_adj(const float* __restrict__ e1, const float* __restrict__ e2,
                                                      int rank, int bidir, int N, float* __restrict__ plain, int ldP) {
  extern __shared__ float logits[];      // N floats
  __shared__ float red[256];
  __shared__ float erow[64];
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int r = tid; r < rank; r += 256) erow[r] = e1[(size_t)n * rank + r];
  __syncthreads();
  float mx = 0.f;  // relu output is >= 0
  for (int m = tid; m < N; m += 256) {
    float s = 0.f;
    if (bidir) {
      for (int r = 0; r < rank; ++r) s = fmaf(erow[r], e1[(size_t)m * rank + r], s);
    } else {
      for (int r = 0; r < rank; ++r) s = fmaf(erow[r], e2[(size_t)r * N + m], s);
    }
    s = fmaxf(s, 0.f);
    logits[m] = s;
    mx = fmaxf(mx, s);
  }
  red[tid] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int m = tid; m < N; m += 256) {
    const float e = expf(logits[m] - mx);
    logits[m] = e;
    sum += e;
  }
  red[tid] = sum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float inv = 1.0f / red[0];
  for (int m = tid; m < N; m += 256) plain[(size_t)n * ldP + m] = logits[m] * inv;
}

// static first-order supports (model.supports[s][1], MultiATGCN.py:269-283) -> transposed slots
// (accumulate: add into the slot instead - cheb_order = 1 sums its dense supports, see StackMap)
// (ldSrc: leading dimension of S - N for the reference's supports, the plain buffer's pitch for the adaptive adjacency)
__global__ __launch_bounds__(256) void k_static_transpose(const float* __restrict__ S, int N, int ldSrc,
                                                          float* __restrict__ St, int ldS, int col0,
                                                          float* __restrict__ plain, int ldP, int accumulate) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx: m block, by: n block
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int n = by + j, m = bx + tx;
    const float v = (n < N && m < N) ? S[(size_t)n * ldSrc + m] : 0.f;
    tile[j][tx] = v;
    if (plain && plain != S && n < N && m < N) plain[(size_t)n * ldP + m] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int m = bx + j, n = by + tx;
    if (m < N && n < N) {
      float* dst = &St[(size_t)m * ldS + col0 + n];
      *dst = accumulate ? *dst + tile[tx][j] : tile[tx][j];
    }
  }
}

// Chebyshev recursion T_k = 2 S T_{k-1} - T_{k-2} (MultiATGCN.py:98-99): prod = S T_{k-1} comes from k_mix
// (plain output); this kernel forms T_k, stores it transposed into its stack slot and plain for the next order.
__global__ __launch_bounds__(256) void k_cheb_combine(const float* __restrict__ prod, const float* __restrict__ prev2,
                                                      int prev2_is_identity, int N, int ldP, float* __restrict__ St,
                                                      int ldS, int col0, float* __restrict__ plain_out) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int n = by + j, m = bx + tx;
    float v = 0.f;
    if (n < N && m < N) {
      const float p2 = prev2_is_identity ? (n == m ? 1.f : 0.f) : prev2[(size_t)n * ldP + m];
      v = 2.0f * prod[(size_t)n * ldP + m] - p2;
      plain_out[(size_t)n * ldP + m] = v;
    }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int m = bx + j, n = by + tx;
    if (m < N && n < N) St[(size_t)m * ldS + col0 + n] = tile[tx][j];
  }
}

// =================================================================================================
// 2. node-adaptive weights in MFMA fragment order
// =================================================================================================
// W[n][k][i][o] = g_k * sum_d E[n][d] * Wpool[d][k][i][o], bias[n][o] = sum_d E[n][d] * bpool[d][o]
// (MultiATGCN.py:102-105; g = softmax(weights_g) is folded here instead of scaling the stack; diagonal supports
// are folded into the identity slot, see StackMap).  One thread produces, for PREP_NB consecutive nodes, the
// float4 a lane feeds to 4 consecutive MFMAs, so every pool element it reads serves PREP_NB nodes.
#define PREP_NB 8

// Chebyshev value t_order(s): t_0 = 1, t_1 = s, t_j = 2 s t_{j-1} - t_{j-2}   (MultiATGCN.py:98-99 on a diagonal)
__device__ __forceinline__ float cheb_scalar(float s, int order) {
  float t0 = 1.f, t1 = s;
  for (int j = 2; j <= order; ++j) { const float t2 = 2.f * s * t1 - t0; t0 = t1; t1 = t2; }
  return t1;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_prep_stream(PrepStream a) {
  const int nBase = blockIdx.y * PREP_NB;
  const int unit = blockIdx.x * 256 + threadIdx.x;
  const int OT = a.O >> 4;
  if (unit >= a.groups * OT * 64) return;
  const int lane = unit & 63, ct = (unit >> 6) % OT, g = (unit >> 6) / OT;
  const int o = 16 * ct + (lane & 15);
  // decode the 4 rows of this lane's float4
  int slot[4], chan[4];   // kept slot (-1: zero row, -2: bias row) and input channel
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (KIND == 0) {
      const int kk = 16 * g + 4 * (lane >> 4) + s;
      slot[s] = kk >> 6; chan[s] = a.iOfs + (kk & 63);
    } else if (KIND == 1) {
      const int kk = 16 * g + 4 * (lane >> 4) + s;
      const int nx = a.map.nKeep * a.C0;
      if (kk < nx) { slot[s] = kk / a.C0; chan[s] = kk - slot[s] * a.C0; }
      else { slot[s] = kk == nx ? -2 : -1; chan[s] = 0; }
    }
  }
  // softmax over the stack weights (tiny)
  float gmax = -3.0e38f, gsum = 1.f;
  if (a.wg) {
    gsum = 0.f;
    for (int k = 0; k < a.map.KtotOrig; ++k) gmax = fmaxf(gmax, a.wg[k]);
    for (int k = 0; k < a.map.KtotOrig; ++k) gsum += expf(a.wg[k] - gmax);
  }
  auto gk = [&](int k) { return a.wg ? expf(a.wg[k] - gmax) / gsum : 1.f; };
  const size_t dstride = (size_t)a.map.KtotOrig * a.I * a.O;
  float acc[PREP_NB][4];
#pragma unroll
  for (int b = 0; b < PREP_NB; ++b)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[b][s] = 0.f;
  // kept slots (and the bias row): one pool offset per row
  size_t rofs[4];
  float scale[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    rofs[s] = 0; scale[s] = 0.f;
    if (slot[s] >= 0) {
      const int k = a.map.keepK[slot[s]];
      rofs[s] = ((size_t)k * a.I + chan[s]) * a.O + o;
      scale[s] = gk(k);
    } else if (slot[s] == -2) {
      scale[s] = 1.f;
    }
  }
#pragma unroll 4
  for (int dd = 0; dd < a.d; ++dd) {
    float pv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float v = 0.f;
      if (slot[s] >= 0) v = a.wpool[(size_t)dd * dstride + rofs[s]];
      else if (slot[s] == -2) v = a.bpool[(size_t)dd * a.O + o];
      pv[s] = v * scale[s];
    }
#pragma unroll
    for (int b = 0; b < PREP_NB; ++b) {
      const float e = a.E[(size_t)min(nBase + b, a.N - 1) * a.d + dd];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[b][s] = fmaf(e, pv[s], acc[b][s]);
    }
  }
  // folded diagonal supports: identity-slot rows also take t_order(s_n) * g_k * W[n][k]
  const bool anyIdent = slot[0] == 0 || slot[1] == 0 || slot[2] == 0 || slot[3] == 0;
  if (anyIdent) {
    for (int q = 0; q < a.map.nDiag; ++q) {
      const int k = a.map.diagK[q];
      const float gq = gk(k);
      float part[PREP_NB][4];
#pragma unroll
      for (int b = 0; b < PREP_NB; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) part[b][s] = 0.f;
      for (int dd = 0; dd < a.d; ++dd) {
        float pv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
          pv[s] = slot[s] == 0 ? a.wpool[(size_t)dd * dstride + ((size_t)k * a.I + chan[s]) * a.O + o] : 0.f;
#pragma unroll
        for (int b = 0; b < PREP_NB; ++b) {
          const float e = a.E[(size_t)min(nBase + b, a.N - 1) * a.d + dd];
#pragma unroll
          for (int s = 0; s < 4; ++s) part[b][s] = fmaf(e, pv[s], part[b][s]);
        }
      }
#pragma unroll
      for (int b = 0; b < PREP_NB; ++b) {
        const int n = min(nBase + b, a.N - 1);
        const float t = gq * cheb_scalar(a.map.diagSrc[q][(size_t)n * (a.map.N + 1)], a.map.diagOrder[q]);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[b][s] = fmaf(t, part[b][s], acc[b][s]);
      }
    }
  }
  size_t frag;
  if (a.OTdst > 0) frag = ((size_t)(g * a.OTdst + a.otOfs + ct) * 64 + lane) * 4;
  else frag = (size_t)unit * 4;
#pragma unroll
  for (int b = 0; b < PREP_NB; ++b) {
    const int n = nBase + b;
    if (n < a.N)
      *reinterpret_cast<float4*>(a.out + (size_t)n * a.nodeStride + a.baseOfs + frag) =
          make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
  }
}

// Round 3: the same streams on the matrix cores.  W[n][row][o] = sum_d pool[d][row][o] * E[n][d] is a GEMM with the embedding
// dimension as its (short) reduction: A = 16 pool elements x d, B = d x 16 nodes, and the 16 A rows are chosen so that
// the accumulator of a lane - rows 4 (l >> 4) + e, column l & 15 of the 16x16x4 fp32 MFMA - is exactly one float4 of the
// fragment stream of node l & 15: rows 4 q + s of a "quad" are the four weight rows s of the float4 in slot j = 4 j4 + q
// of lane group kq.  A wave keeps the A fragments of 8 quads (half a fragment row: 2 lane groups x 4 j4) in registers and
// walks 16-node tiles; per tile it reads 4 ceil(d / 4) embedding values, issues 8 ceil(d / 4) MFMAs and stores 8 float4
// per lane (64-byte pieces that the L2 merges into the node's 1 KB rows).  The exact-fp32 MFMA accumulates k = 0..3 in
// order, so the sums are the fma chains of the kernel above.  DS = ceil(d / 4) steps are compiled in (3 / 5 / 8:
// embeddings of up to 12 / 20 / 32; wider ones take the first kernel).  `groups` k-groups of kind-0 rows are followed by
// `groupsX` k-groups of kind-1 rows (layer 0: the folded x rows and the bias row sit behind the recurrent rows of the same
// node stream, so one launch writes both).  Measured at N = 403, d = 20 (profiles/r03_prep_lab.log): 27 us per launch
// against 49 us (+ 35 us for the kind-1 launch) of the first kernel, which re-reads every pool value per 8 nodes and
// spends 2 d VALU fmas per output; a VALU variant with the pool values in registers and a node loop was slower (68 us).
template <int DS>
__global__ __launch_bounds__(256) void k_prep_mfma(PrepStream a, int tilesPerBlock) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int OT = a.O >> 4;
  const int pair = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int ct = pair % OT, g = pair / OT;
  const int r = lane & 15, ka = lane >> 4;      // A operand: row r of the quad (slot r >> 2, weight row s = r & 3), d index ka
  float gmax = -3.0e38f, gsum = 1.f;
  if (a.wg) {
    gsum = 0.f;
    for (int k = 0; k < a.map.KtotOrig; ++k) gmax = fmaxf(gmax, a.wg[k]);
    for (int k = 0; k < a.map.KtotOrig; ++k) gsum += expf(a.wg[k] - gmax);
  }
  auto gk = [&](int k) { return a.wg ? expf(a.wg[k] - gmax) / gsum : 1.f; };
  const size_t dstride = (size_t)a.map.KtotOrig * a.I * a.O;
  float A[8][DS], Aq[8][DS];
  bool ident = false;
#pragma unroll
  for (int q8 = 0; q8 < 8; ++q8) {
    const int kq = 2 * half + (q8 >> 2), j = 4 * (q8 & 3) + (r >> 2), sr = r & 3;
    const int o = 16 * ct + j;
    int slot, chan;
    if (g < a.groups) {
      const int kk = 16 * g + 4 * kq + sr;
      slot = kk >> 6; chan = a.iOfs + (kk & 63);
    } else {
      const int kk = 16 * (g - a.groups) + 4 * kq + sr;
      const int nx = a.map.nKeep * a.C0;
      if (kk < nx) { slot = kk / a.C0; chan = kk - slot * a.C0; }
      else { slot = kk == nx ? -2 : -1; chan = 0; }
    }
    const float* src = a.wpool; size_t sstep = 0; float scale = 0.f;
    if (slot >= 0) {
      const int k = a.map.keepK[slot];
      src = a.wpool + ((size_t)k * a.I + chan) * a.O + o; sstep = dstride; scale = gk(k);
    } else if (slot == -2) {
      src = a.bpool + o; sstep = (size_t)a.O; scale = 1.f;
    }
    const bool id = slot == 0 && a.map.nDiag > 0;
    ident = ident || id;
    const float* srcq = id ? a.wpool + ((size_t)a.map.diagK[0] * a.I + chan) * a.O + o : a.wpool;
#pragma unroll
    for (int st = 0; st < DS; ++st) {
      const int dd = 4 * st + ka;
      const float v = src[(size_t)min(dd, a.d - 1) * sstep];
      A[q8][st] = (dd < a.d && slot != -1) ? v * scale : 0.f;
      const float vq = srcq[(size_t)min(dd, a.d - 1) * (id ? dstride : 0)];
      Aq[q8][st] = (dd < a.d && id) ? vq : 0.f;
    }
  }
  const bool anyIdent = __ballot(ident) != 0ull;
  const int nTiles = (a.N + 15) >> 4;
  const int tile0 = blockIdx.y * tilesPerBlock, tile1 = min(tile0 + tilesPerBlock, nTiles);
  const int nodeL = lane & 15, kb = lane >> 4;   // B operand / accumulator: node of the tile, d index; slot quarter kb
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
          const int k = a.map.diagK[q];
#pragma unroll
          for (int q8 = 0; q8 < 8; ++q8) {
            const int kq = 2 * half + (q8 >> 2), j = 4 * (q8 & 3) + (r >> 2), sr = r & 3;
            int slot, chan;
            if (g < a.groups) { const int kk = 16 * g + 4 * kq + sr; slot = kk >> 6; chan = a.iOfs + (kk & 63); }
            else {
              const int kk = 16 * (g - a.groups) + 4 * kq + sr;
              if (kk < a.map.nKeep * a.C0) { slot = kk / a.C0; chan = kk - slot * a.C0; } else { slot = -1; chan = 0; }
            }
#pragma unroll
            for (int st = 0; st < DS; ++st) {
              const int dd = 4 * st + ka;
              const float v = a.wpool[(size_t)min(dd, a.d - 1) * dstride + ((size_t)k * a.I + chan) * a.O + 16 * ct + j];
              part[q8] = MFMA16((dd < a.d && slot == 0) ? v : 0.f, Bf[st], part[q8]);
            }
          }
        }
        const float t = gk(a.map.diagK[q]) *
                        cheb_scalar(a.map.diagSrc[q][(size_t)nc * (a.map.N + 1)], a.map.diagOrder[q]);
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[q8][e] = fmaf(t, part[q8][e], acc[q8][e]);
      }
    }
    if (n < a.N) {
      float* dst = a.out + (size_t)n * a.nodeStride + a.baseOfs;
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8) {
        const int kq = 2 * half + (q8 >> 2), j = 4 * (q8 & 3) + kb;
        const size_t frag = ((size_t)(a.OTdst > 0 ? g * a.OTdst + a.otOfs + ct : g * OT + ct) * 64 + kq * 16 + j) * 4;
        *reinterpret_cast<float4*>(dst + frag) = make_float4(acc[q8][0], acc[q8][1], acc[q8][2], acc[q8][3]);
      }
    }
#pragma unroll
    for (int st = 0; st < DS; ++st) Bf[st] = Bn[st];
  }
}

// bias[n][colOfs + o] = E[n] . bpool[:, o]   (hoisted-PX layers)
__global__ __launch_bounds__(256) void k_prep_bias(const float* __restrict__ E, const float* __restrict__ bpool, int d,
                                                   int O, int N, float* __restrict__ out, int ldo, int colOfs) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * O) return;
  const int n = idx / O, o = idx - n * O;
  float s = 0.f;
  for (int dd = 0; dd < d; ++dd) s = fmaf(E[(size_t)n * d + dd], bpool[(size_t)dd * O + o], s);
  out[(size_t)n * ldo + colOfs + o] = s;
}

// nn.Linear / Conv2d weight (O, I) -> fragment order [jg][OT][64][4] of B[j][o] = W[o][in(j)].
// Rows j < Cpad map to input j (zero beyond C); rows j >= Cpad map to input C + (j - Cpad).
__global__ __launch_bounds__(256) void k_prep_linear(const float* __restrict__ W, int I, int O, int C, int Cpad,
                                                     int rows, int OT, float* __restrict__ out) {
  const int unit = blockIdx.x * 256 + threadIdx.x;
  const int units = (rows >> 3) * OT * 64;
  if (unit >= units) return;
  const int lane = unit & 63, ot = (unit >> 6) % OT, jg = (unit >> 6) / OT;
  const int o = ot * 32 + (lane & 31);
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = jg * 8 + 4 * (lane >> 5) + q;
    int in = -1;
    if (j < Cpad) { if (j < C) in = j; } else { in = C + (j - Cpad); }
    v[q] = (in >= 0 && in < I && o < O) ? W[(size_t)o * I + in] : 0.f;
  }
  *reinterpret_cast<float4*>(out + (size_t)unit * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

// =================================================================================================
// 3. layout helpers (all tiny, HBM-trivial)
// =================================================================================================
// user (rows, N, C) -> padded [rows][Np][C] (pad rows zero)
__global__ __launch_bounds__(256) void k_pack_rows(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                   int N, int Np, int C) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)rows * Np * C;
  if (idx >= total) return;
  const int c = idx % C;
  const int n = (idx / C) % Np;
  const size_t r = idx / ((size_t)C * Np);
  dst[idx] = (src && n < N) ? src[(r * N + n) * C + c] : 0.f;
}

__global__ __launch_bounds__(256) void k_unpack_rows(const float* __restrict__ src, float* __restrict__ dst, int rows,
                                                     int N, int Np, int C) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)rows * N * C;
  if (idx >= total) return;
  const int c = idx % C;
  const int n = (idx / C) % N;
  const size_t r = idx / ((size_t)C * N);
  dst[idx] = src[(r * Np + n) * C + c];
}

// user (B, T, N, C) <-> padded time-major [T][B][Np][C]
__global__ __launch_bounds__(256) void k_pack_seq_tm(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                     int T, int N, int Np, int C) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)T * B * Np * C;
  if (idx >= total) return;
  const int c = idx % C;
  const int n = (idx / C) % Np;
  const int b = (idx / ((size_t)C * Np)) % B;
  const int t = idx / ((size_t)C * Np * B);
  dst[idx] = (n < N) ? src[(((size_t)b * T + t) * N + n) * C + c] : 0.f;
}

__global__ __launch_bounds__(256) void k_unpack_seq_tm(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                       int T, int N, int Np, int C) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)B * T * N * C;
  if (idx >= total) return;
  const int c = idx % C;
  const int n = (idx / C) % N;
  const int t = (idx / ((size_t)C * N)) % T;
  const int b = idx / ((size_t)C * N * T);
  dst[idx] = src[(((size_t)t * B + b) * Np + n) * C + c];
}

// zero rows n in [N, Np) of a [rows][Np][C] buffer
__global__ __launch_bounds__(256) void k_zero_pad_rows(float* __restrict__ buf, int rows, int N, int Np, int C) {
  const int per = (Np - N) * C;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)rows * per) return;
  const size_t r = idx / per;
  const int q = idx - r * per;
  buf[(r * Np + N) * C + q] = 0.f;
}

// the same for up to 8 buffers of one shape in ONE launch (blockIdx.y picks the buffer), 16 bytes per thread (C % 4 == 0):
// the backward opens with six of them in front of its first chain
struct PadRowBufs { float* p[8]; };
__global__ __launch_bounds__(256) void k_zero_pad_rows_multi(PadRowBufs bufs, int rows, int N, int Np, int C) {
  const int per4 = (Np - N) * C / 4;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)rows * per4) return;
  const size_t r = idx / per4;
  const int q = (int)(idx - r * per4);
  *reinterpret_cast<float4*>(bufs.p[blockIdx.y] + (r * Np + N) * C + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// =================================================================================================
// 4. temporal-head fusion prologue (MultiATGCN.py:365-402)
// =================================================================================================
// x0[b][t][n][c<od]  = sum_h softmax(weight_tsg)[h] * X[b][begin_h + t][n][start+c] * weight_ts[h][t][n][c]
// x0[b][t][n][od+j]  = X[b][t][n][ext_src[j]]           (time of day / dynamic channels)
// Series mode: row label_start[b] + offset of the device-resident series.  The caller guarantees the range
// (include/matgcn.h); a violated contract must not fault the device, so the row is clamped into the series and the
// violation counted (matgcn_series_violations reads and clears the counter).
__device__ unsigned long long g_series_violations = 0;
__device__ __forceinline__ size_t series_row(long row, long steps) {
  if (row < 0 || row >= steps) {
    atomicAdd(&g_series_violations, 1ull);
    row = row < 0 ? 0 : steps - 1;
  }
  return (size_t)row;
}

__global__ __launch_bounds__(256) void k_fuse_heads(FuseArgs a) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)a.B * a.T * a.N;
  if (idx >= total) return;
  const int n = idx % a.N;
  const int t = (idx / a.N) % a.T;
  const int b = idx / ((size_t)a.N * a.T);
  float gmax = -3.0e38f, gsum = 0.f;
  for (int h = 0; h < a.nTs; ++h) gmax = fmaxf(gmax, a.tsg[h]);
  for (int h = 0; h < a.nTs; ++h) gsum += expf(a.tsg[h] - gmax);
  // row `step` of sample b: a window row, or - series mode - the series row at label start + rel[step]
  auto xrow = [&](int step) -> const float* {
    const size_t r = a.labelStart ? series_row((long)a.labelStart[b] + a.rel[step], a.seriesSteps)
                                  : (size_t)b * a.xSteps + step;
    return a.X + (r * a.N + n) * a.F;
  };
  float* dst = a.x0 + (((size_t)b * a.T + t) * a.Np + n) * a.C0;
  for (int c = 0; c < a.od; ++c) {
    float acc = 0.f;
    for (int h = 0; h < a.nHeads; ++h) {
      const float g = expf(a.tsg[h] - gmax) / gsum;
      const float xv = xrow(a.headBegin[h] + t)[a.startDim + c];
      const float wv = a.ts[h][((size_t)t * a.N + n) * a.od + c];
      acc += g * xv * wv;
    }
    dst[c] = acc;
  }
  for (int j = 0; j < a.C0 - a.od; ++j)
    dst[a.od + j] = xrow(t)[a.extSrc[j]];
}

// layer-0 encoder input as a plain matrix for the mix GEMM: X0m[m][(row*C0 + c)] = x0p[row][m][c]
__global__ __launch_bounds__(256) void k_x0_to_matrix(const float* __restrict__ x0p, float* __restrict__ X0m, int rows,
                                                      int Np, int C0, int ld) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)Np * ld;
  if (idx >= total) return;
  const int col = idx % ld;
  const int m = idx / ld;
  float v = 0.f;
  if (col < rows * C0) {
    const int row = col / C0, c = col - row * C0;
    v = x0p[((size_t)row * Np + m) * C0 + c];
  }
  X0m[idx] = v;
}

// folded x-part of the layer-0 node GEMM: XA0[t][n][b][Kx] = [x0 (k=0) | mix_k(x0), k<Ks | 1.0 | 0...]
// (the 1.0 column meets the bias row of the folded weights)
__global__ __launch_bounds__(256) void k_build_xa0(const float* __restrict__ x0p, const float* __restrict__ MX0,
                                                   float* __restrict__ XA0, int B, int T, int N, int Np, int C0,
                                                   int Ks, int Kx, int ld) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)T * N * B * Kx;
  if (idx >= total) return;
  const int j = idx % Kx;
  const int b = (idx / Kx) % B;
  const int n = (idx / ((size_t)Kx * B)) % N;
  const int t = idx / ((size_t)Kx * B * N);
  const int row = b * T + t;
  float v = 0.f;
  const int nx = (Ks + 1) * C0;
  if (j < C0) v = x0p[((size_t)row * Np + n) * C0 + j];
  else if (j < nx) {
    const int k = j / C0 - 1, c = j % C0;
    v = MX0[((size_t)k * Np + n) * ld + (size_t)row * C0 + c];
  } else if (j == nx) v = 1.0f;
  XA0[idx] = v;
}

// =================================================================================================
// 5. graph mix GEMM (MultiATGCN.py:106):  out[(k,n)][col] = sum_m S_k[n][m] * X[m][col]
// =================================================================================================
// 64 x 64 output tile per workgroup, 4 waves, each a 32x32 tile as 2x2 accumulators of the 16x16x4 MFMA; K-step 16 staged through
// LDS (double buffer, fed from registers that run two tiles ahead of the MFMAs; one barrier per step).  A = St (k-major, so the tile is 16 rows of
// 256 contiguous bytes), B = 64 feature columns of one state row (256-byte lines).  Workgroups that share an
// XCD (id % 8) sweep the row tiles of one column tile back to back, so St and that X slice stay in its L2.
// ROLE only names the instantiation (0: pre-passes and Chebyshev products, 1: the recurrent step's mix of h / z*h),
// so that profilers list the roofline kernel - the per-step launch - on its own line.
// FLUSH (round 4): the accumulators are one fp32 fma chain over the WHOLE reduction - 4 096 links at N = 4 096, where the
// reference's own prediction sits 4.7e-7 from its float64 run and this kernel's single chain left the path at 1.2e-6
// (tests/golden/fp64_gap.npz; a blocked CPU sgemm sums K in blocks of a few hundred).  With FLUSH the chain is cut every
// MIX_FLUSH_TILES K-tiles (256 reduction indices): partial sums go to a second accumulator set.  Launched for nK > 64 only,
// so every graph of at most 1 024 nodes - the headline's 403 - keeps its kernel and its bits.
// (round 4, measured and rejected: rotating the wave priority with the K-tile index by dispatch round so that the five
//  workgroups of a CU leave together - they do, and every one is slower: 43.0 vs 41.3 us, profiles/r04_mix_stamps_lab.log)
#ifndef MIX_FLUSH_TILES
#define MIX_FLUSH_TILES 16
#endif
template <int ROLE, bool FLUSH = false>
__global__ __launch_bounds__(256) void k_mix(MixArgs a) {
  // (FLUSH costs 16 registers, 72 in all: seven workgroups per CU instead of eight, +3.7 % per launch at N = 4 096; forced
  //  into 64 registers the compiler spilled a pointer inside the K loop, +6.5 %: profiles/r04_mix_lab.log)
  __shared__ __attribute__((aligned(16))) float As[2][16 * 64];
  __shared__ __attribute__((aligned(16))) float Bs[2][16 * 64];
  const int id = blockIdx.x;
  int colTile, rowTile;
  if ((a.nColTiles & 7) == 0) {
    const int xcd = id & 7, j = id >> 3, cpx = a.nColTiles >> 3;
    rowTile = j % a.nRowTiles;
    colTile = xcd * cpx + j / a.nRowTiles;
  } else {
    rowTile = id % a.nRowTiles;
    colTile = id / a.nRowTiles;
  }
  const int row0 = rowTile * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1, j = lane & 15, kq = lane >> 4;
  const int kk = tid >> 4, sg = tid & 15;
#ifdef MIX_LAB_STAGGER   // LAB: the workgroups that share a CU (ids 256 apart) start MIX_LAB_STAGGER x 64 cycles apart
  for (int d = 0; d < (int)(blockIdx.x >> 8); ++d) __builtin_amdgcn_s_sleep(MIX_LAB_STAGGER);
#endif
  const int part = blockIdx.y;     // split reduction (MixArgs.parts): 0 unless the launch has a second grid dimension
  NODE_STAMP_DECL
  NODE_STAMP(0);
  const float* ap = a.St + (size_t)part * a.aPartStride + (size_t)kk * a.ldS + row0 + sg * 4;
  const float* bp = a.X + (size_t)part * a.xPartStride + (size_t)colTile * a.xTileStride + (size_t)kk * a.ldX + sg * 4;
  // LDS image of a K-tile: row kk (one reduction index, 64 values) rotated by 16*(kk&3) floats, so that the four
  // k rows one 16x16x4 MFMA step reads (kq = 0..3) sit in four different bank quarters
  const int stPos = kk * 64 + (((sg + 4 * (kk & 3)) & 15) << 2);
  // K-tile t+1 waits in registers while tile t is multiplied, and tile t+2 is already requested: two register
  // sets alternate (the loop is unrolled by two so that they stay named registers), loads are unconditional
  // (clamped to the last tile) so the compiler keeps counting them instead of draining the queue at a branch
  const int last = a.nK - 1;
  auto ldA = [&](int t) { return *reinterpret_cast<const float4*>(ap + (size_t)min(t, last) * 16 * a.ldS); };
  auto ldB = [&](int t) { return *reinterpret_cast<const float4*>(bp + (size_t)min(t, last) * 16 * a.ldX); };
  // All six requests of the prologue leave BEFORE the first wait (round 4: at 48 registers the compiler had serialised
  // them to save two - load A0, wait, store, load B0, wait, store, and only then tiles 1 and 2: two exposed memory round
  // trips in front of every launch's first MFMA; the sched_barrier pins the order, the stores then wait with vmcnt(5) / (4))
  float4 ra0, rb0, ra1, rb1;
  {
    const float4 a0 = ldA(0), b0 = ldB(0);
#ifndef MIX_LAB_SERIAL_PROLOGUE
    __builtin_amdgcn_sched_barrier(0);    // tile 0 first: its stores wait for the two OLDEST requests only
#endif
    ra0 = ldA(1); rb0 = ldB(1); ra1 = ldA(2); rb1 = ldB(2);
#ifndef MIX_LAB_SERIAL_PROLOGUE
    __builtin_amdgcn_sched_barrier(0);
#endif
    NODE_STAMP(1);   // six requests issued
    *reinterpret_cast<float4*>(&As[0][stPos]) = a0;
    NODE_STAMP(2);   // A tile 0 arrived
    *reinterpret_cast<float4*>(&Bs[0][stPos]) = b0;
    NODE_STAMP(3);   // B tile 0 arrived
  }
  __syncthreads();
  NODE_STAMP(4);
  // the wave's 32x32 output tile = 2x2 accumulators of v_mfma_f32_16x16x4_f32: 20 independent accumulator chains
  // per SIMD at 5 resident workgroups per CU, enough to keep the matrix pipe issuing back to back
  f32x4 acc[2][2], tot[FLUSH ? 2 : 1][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (FLUSH) tot[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  const int rotA0 = (wr * 32 + j + 16 * kq) & 63, rotA1 = (wr * 32 + 16 + j + 16 * kq) & 63;
  const int rotB0 = (wc * 32 + j + 16 * kq) & 63, rotB1 = (wc * 32 + 16 + j + 16 * kq) & 63;
  auto mma = [&](int cur) {
    const float* A = &As[cur][kq * 64];
    const float* Bm = &Bs[cur][kq * 64];
#pragma unroll
    for (int s = 0; s < 4; ++s) {      // reduction indices 4s + kq of this K-tile
      const float a0 = A[s * 256 + rotA0], a1 = A[s * 256 + rotA1];
      const float b0 = Bm[s * 256 + rotB0], b1 = Bm[s * 256 + rotB1];
      acc[0][0] = MFMA16(a0, b0, acc[0][0]);
      acc[0][1] = MFMA16(a0, b1, acc[0][1]);
      acc[1][0] = MFMA16(a1, b0, acc[1][0]);
      acc[1][1] = MFMA16(a1, b1, acc[1][1]);
    }
  };
#ifdef MIX_LAB_NK   // LAB: only the first MIX_LAB_NK K-tiles (results are garbage: timing of the fixed part only)
  const int nKrun = a.nK < MIX_LAB_NK ? a.nK : MIX_LAB_NK;
#else
  const int nKrun = a.nK;
#endif
  for (int it = 0; it < nKrun; it += 2) {
    mma(0);                                               // tile it
    *reinterpret_cast<float4*>(&As[1][stPos]) = ra0;      // tile it+1 (a clamped copy past the end: unused)
    *reinterpret_cast<float4*>(&Bs[1][stPos]) = rb0;
    ra0 = ldA(it + 3); rb0 = ldB(it + 3);
    __syncthreads();
    if (it + 1 < nKrun) {
      mma(1);                                             // tile it+1
      *reinterpret_cast<float4*>(&As[0][stPos]) = ra1;    // tile it+2
      *reinterpret_cast<float4*>(&Bs[0][stPos]) = rb1;
      ra1 = ldA(it + 4); rb1 = ldB(it + 4);
      __syncthreads();
    }
    if constexpr (FLUSH) {
      if (((it + 2) & (MIX_FLUSH_TILES - 1)) == 0) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tot[p][q][e] += acc[p][q][e];
            acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      }
    }
  }
  if constexpr (FLUSH) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[p][q][e] += tot[p][q][e];
  }
  // Epilogue: each wave turns its 32x32 accumulator tile through LDS (the K-loop buffers are free after the last
  // barrier; 16-byte slots XOR-swizzled by row so both the scalar writes and the b128 reads are conflict-free)
  // and writes whole 128-byte row segments with 16-byte WRITE-THROUGH stores (sc1): the 20-75 MB this kernel
  // produces then leave the L2 while it is still computing instead of as one dirty-line flush at its end, which
  // the next kernel of the chain would otherwise wait for.
  NODE_STAMP(5);   // K loop issued
  // (round 4: the last workgroup of a CU runs this stretch alone, one wave per SIMD - 8.5 k cycles by the stamps of
  //  tools/labs/stamps_mix_r04.py: sixteen scalar LDS writes with seven address instructions each, then four times read ->
  //  wait -> 64-bit divide -> guarded store.  Now the sixteen offsets are three adds from precomputed pieces, the four
  //  rows are read in one batch, and a row outside the output is dropped by the buffer's range check instead of a branch.)
  float* stg = (w < 2 ? &As[0][0] : &Bs[0][0]) + (w & 1) * 1024;
  {
    const int jq = j >> 2, jr = j & 3, kb = kq & 1;
    int xe[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xe[e] = e * 32 + ((jq ^ e) << 2) + jr;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int base = (p * 16 + 4 * kq) * 32 + ((q ^ kb) << 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) stg[base + xe[e]] = acc[p][q][e];
      }
  }
  NODE_STAMP(6);   // accumulators -> LDS
  const bool wt = a.outFloats > 0 && a.outFloats < (1L << 29);   // 32-bit byte offsets
  float* outp = a.out + (size_t)part * a.outPartStride;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, wt ? (int)(a.outFloats * 4) : 0, 0x00020000);
  float4 v4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int lrow = u * 8 + (lane >> 3), q = lane & 7;
    v4[u] = *reinterpret_cast<const float4*>(&stg[lrow * 32 + ((q ^ (lrow & 7)) << 2)]);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int lrow = u * 8 + (lane >> 3), q = lane & 7;
    const float4 v = v4[u];
    const int row = row0 + wr * 32 + lrow;
    const int k = row / a.Np, n = row - k * a.Np;
#ifdef MIX_LAB_NOSTORE   // LAB: no output (a never-true condition keeps the accumulators alive)
    const bool ok = k < a.Ks && n < a.N && v.x == 1.2345e-30f;
#else
    const bool ok = k < a.Ks && n < a.N;
#endif
    const size_t off = (size_t)colTile * a.sT + (size_t)n * a.sN + (size_t)k * a.sK + wc * 32 + q * 4;
    if (wt) {
      const u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      // a row outside the output: an offset beyond num_records - the store is dropped by the range check, no branch
      __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, ok ? (int)(off * 4) : (int)0x7ffffff0, 0, 16);   // aux 16 = sc1
    } else if (ok) {
      *reinterpret_cast<float4*>(outp + off) = v;
    }
  }
  NODE_STAMP(7);   // stores issued
#ifdef NODE_LAB_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  NODE_STAMP(8);   // stores acknowledged
#endif
  NODE_STAMP_FLUSH(a);
}

// -------------------------------------------------------------------------------------------------
// 5a''. the same mix with 64-row x 32-column tiles: small batches
// -------------------------------------------------------------------------------------------------
// The reference ships batch_size 16 (MultiATGCN.json:12).  There k_mix has 20 x 16 = 320 workgroups for 256 CUs - 64 CUs get
// two, the launch takes 20.6 us for a quarter of the B = 64 work (41 us).  With half-width column tiles there are 640
// workgroups (2.5 per CU, five waves per CU on average as before, but the longest CU holds 3 halves instead of 2 wholes).
// Same pipeline as k_mix (K-step 16 through LDS, two K-tiles ahead in registers, rotated rows); a wave owns 32 rows x 16
// columns = 2 x 1 accumulators, the B tile uses columns 0..31 of its 64-wide LDS rows (both halves of the workgroup request
// it - duplicate stores of equal values).  Launched for at most 32 column tiles (B <= 32).
template <int ROLE>
__global__ __launch_bounds__(256) void k_mix_c32(MixArgs a) {
  __shared__ __attribute__((aligned(16))) float As[2][16 * 64];
  __shared__ __attribute__((aligned(16))) float Bs[2][16 * 64];
  const int id = blockIdx.x, nCt2 = 2 * a.nColTiles;
  int ct2, rowTile;
  if ((nCt2 & 7) == 0) {
    const int xcd = id & 7, jj = id >> 3, cpx = nCt2 >> 3;
    rowTile = jj % a.nRowTiles;
    ct2 = xcd * cpx + jj / a.nRowTiles;
  } else {
    rowTile = id % a.nRowTiles;
    ct2 = id / a.nRowTiles;
  }
  const int colTile = ct2 >> 1, half = ct2 & 1;
  const int row0 = rowTile * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1, j = lane & 15, kq = lane >> 4;
  const int kk = tid >> 4, sg = tid & 15, sgB = sg & 7;
  const int part = blockIdx.y;
  const float* ap = a.St + (size_t)part * a.aPartStride + (size_t)kk * a.ldS + row0 + sg * 4;
  const float* bp = a.X + (size_t)part * a.xPartStride + (size_t)colTile * a.xTileStride + (size_t)kk * a.ldX + half * 32 + sgB * 4;
  const int stPosA = kk * 64 + (((sg + 4 * (kk & 3)) & 15) << 2);
  const int stPosB = kk * 64 + (((sgB + 4 * (kk & 3)) & 15) << 2);
  const int last = a.nK - 1;
  auto ldA = [&](int t) { return *reinterpret_cast<const float4*>(ap + (size_t)min(t, last) * 16 * a.ldS); };
  auto ldB = [&](int t) { return *reinterpret_cast<const float4*>(bp + (size_t)min(t, last) * 16 * a.ldX); };
  float4 ra0, rb0, ra1, rb1;
  {
    const float4 a0 = ldA(0), b0 = ldB(0);
    __builtin_amdgcn_sched_barrier(0);
    ra0 = ldA(1); rb0 = ldB(1); ra1 = ldA(2); rb1 = ldB(2);
    __builtin_amdgcn_sched_barrier(0);
    *reinterpret_cast<float4*>(&As[0][stPosA]) = a0;
    *reinterpret_cast<float4*>(&Bs[0][stPosB]) = b0;
  }
  __syncthreads();
  f32x4 acc[2];
  acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
  const int rotA0 = (wr * 32 + j + 16 * kq) & 63, rotA1 = (wr * 32 + 16 + j + 16 * kq) & 63;
  const int rotB0 = (wc * 16 + j + 16 * kq) & 63;
  auto mma = [&](int cur) {
    const float* A = &As[cur][kq * 64];
    const float* Bm = &Bs[cur][kq * 64];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float a0 = A[s4 * 256 + rotA0], a1 = A[s4 * 256 + rotA1];
      const float b0 = Bm[s4 * 256 + rotB0];
      acc[0] = MFMA16(a0, b0, acc[0]);
      acc[1] = MFMA16(a1, b0, acc[1]);
    }
  };
  for (int it = 0; it < a.nK; it += 2) {
    mma(0);
    *reinterpret_cast<float4*>(&As[1][stPosA]) = ra0;
    *reinterpret_cast<float4*>(&Bs[1][stPosB]) = rb0;
    ra0 = ldA(it + 3); rb0 = ldB(it + 3);
    __syncthreads();
    if (it + 1 < a.nK) {
      mma(1);
      *reinterpret_cast<float4*>(&As[0][stPosA]) = ra1;
      *reinterpret_cast<float4*>(&Bs[0][stPosB]) = rb1;
      ra1 = ldA(it + 4); rb1 = ldB(it + 4);
      __syncthreads();
    }
  }
  // epilogue: the wave's 32 x 16 tile through LDS ([32 rows][4 slots], slot ^ (row & 3)), 16-byte write-through stores
  float* stg = (w < 2 ? &As[0][0] : &Bs[0][0]) + (w & 1) * 1024;
  {
    const int jq = j >> 2, jr = j & 3;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lrow = p * 16 + 4 * kq + e;
        stg[lrow * 16 + ((jq ^ (lrow & 3)) << 2) + jr] = acc[p][e];
      }
  }
  const bool wt = a.outFloats > 0 && a.outFloats < (1L << 29);
  float* outp = a.out + (size_t)part * a.outPartStride;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, wt ? (int)(a.outFloats * 4) : 0, 0x00020000);
  float4 v2[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int lrow = u * 16 + (lane >> 2), q = lane & 3;
    v2[u] = *reinterpret_cast<const float4*>(&stg[lrow * 16 + ((q ^ (lrow & 3)) << 2)]);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int lrow = u * 16 + (lane >> 2), q = lane & 3;
    const float4 v = v2[u];
    const int row = row0 + wr * 32 + lrow;
    const int k = row / a.Np, n = row - k * a.Np;
    const bool ok = k < a.Ks && n < a.N;
    const size_t off = (size_t)colTile * a.sT + (size_t)n * a.sN + (size_t)k * a.sK + half * 32 + wc * 16 + q * 4;
    if (wt) {
      const u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, ok ? (int)(off * 4) : (int)0x7ffffff0, 0, 16);   // sc1; dropped when out of range
    } else if (ok) {
      *reinterpret_cast<float4*>(outp + off) = v;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// 5a'. the same mix with a 32-row x 128-column workgroup tile (the backward's transposed mix)
// -------------------------------------------------------------------------------------------------
// The transposed mix has N = 403 output rows: 7 row tiles of 64 are 10 % padding and 7 x 64 x 3 parts = 1344 workgroups
// are 5.25 per CU (one CU in four runs a sixth round).  32-row tiles cover 416 rows (3 %), and with 128 columns - two
// batch rows - per workgroup 13 x 32 x 3 = 1248 workgroups are 4.9 per CU.  Each of the 4 waves owns 32 of the 128 columns
// and all 32 rows: the same 2 x 2 accumulators, K-tile 16, two tiles ahead in registers, rotated LDS rows (16 floats per odd
// k: the two k rows a 32-lane half reads sit half a bank row apart) and write-through row-store epilogue as k_mix.
// Needs an even number of column tiles (the launcher falls back to k_mix<2> otherwise).
__global__ __launch_bounds__(256) void k_mix_n32(MixArgs a) {
  __shared__ __attribute__((aligned(16))) float As[2][16 * 32];
  __shared__ __attribute__((aligned(16))) float Bs[2][16 * 128];
  const int id = blockIdx.x;
  const int nPairs = a.nColTiles >> 1;
  int colPair, rowTile;
  if ((nPairs & 7) == 0) {
    const int xcd = id & 7, jj = id >> 3, cpx = nPairs >> 3;
    rowTile = jj % a.nRowTiles;
    colPair = xcd * cpx + jj / a.nRowTiles;
  } else {
    rowTile = id % a.nRowTiles;
    colPair = id / a.nRowTiles;
  }
  const int row0 = rowTile * 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int part = blockIdx.y;
  // staging: A tile 16 k x 32 rows = 128 float4 (threads t and t + 128 fetch and store the same one: no branch in the
  // pipeline), B tile 16 k x 128 columns = two float4 per thread (column tiles 2 colPair and 2 colPair + 1)
  const int akk = (tid & 127) >> 3, asg = tid & 7;
  const int bkk = tid >> 4, bsg = tid & 15;
  const float* ap = a.St + (size_t)part * a.aPartStride + (size_t)akk * a.ldS + row0 + asg * 4;
  const float* bp0 = a.X + (size_t)part * a.xPartStride + (size_t)(2 * colPair) * a.xTileStride + (size_t)bkk * a.ldX + bsg * 4;
  const float* bp1 = bp0 + a.xTileStride;
  const int aPos = akk * 32 + ((asg * 4 + 16 * (akk & 1)) & 31);
  const int bPos0 = bkk * 128 + ((bsg * 4 + 16 * (bkk & 1)) & 127);
  const int bPos1 = bkk * 128 + ((64 + bsg * 4 + 16 * (bkk & 1)) & 127);
  const int last = a.nK - 1;
  auto ldA = [&](int t) { return *reinterpret_cast<const float4*>(ap + (size_t)min(t, last) * 16 * a.ldS); };
  auto ldB0 = [&](int t) { return *reinterpret_cast<const float4*>(bp0 + (size_t)min(t, last) * 16 * a.ldX); };
  auto ldB1 = [&](int t) { return *reinterpret_cast<const float4*>(bp1 + (size_t)min(t, last) * 16 * a.ldX); };
  float4 ra0, rb00, rb01, ra1, rb10, rb11;   // (every request of the prologue before its first wait: see k_mix)
  {
    const float4 a0 = ldA(0), b0 = ldB0(0), b1 = ldB1(0);
#ifndef MIX_LAB_SERIAL_PROLOGUE
    __builtin_amdgcn_sched_barrier(0);
#endif
    ra0 = ldA(1); rb00 = ldB0(1); rb01 = ldB1(1); ra1 = ldA(2); rb10 = ldB0(2); rb11 = ldB1(2);
#ifndef MIX_LAB_SERIAL_PROLOGUE
    __builtin_amdgcn_sched_barrier(0);
#endif
    *reinterpret_cast<float4*>(&As[0][aPos]) = a0;
    *reinterpret_cast<float4*>(&Bs[0][bPos0]) = b0;
    *reinterpret_cast<float4*>(&Bs[0][bPos1]) = b1;
  }
  __syncthreads();
  f32x4 acc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment of reduction index k = 4 s + kq: its row is rotated by 16 * (k & 1) = 16 * (kq & 1)
  const int rot = 16 * (kq & 1);
  const int cA0 = (j + rot) & 31, cA1 = (16 + j + rot) & 31;
  const int cB0 = (w * 32 + j + rot) & 127, cB1 = (w * 32 + 16 + j + rot) & 127;
  auto mma = [&](int cur) {
    const float* A = &As[cur][kq * 32];
    const float* Bm = &Bs[cur][kq * 128];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float a0 = A[s * 128 + cA0], a1 = A[s * 128 + cA1];
      const float b0 = Bm[s * 512 + cB0], b1 = Bm[s * 512 + cB1];
      acc[0][0] = MFMA16(a0, b0, acc[0][0]);
      acc[0][1] = MFMA16(a0, b1, acc[0][1]);
      acc[1][0] = MFMA16(a1, b0, acc[1][0]);
      acc[1][1] = MFMA16(a1, b1, acc[1][1]);
    }
  };
  for (int it = 0; it < a.nK; it += 2) {
    mma(0);                                               // tile it
    *reinterpret_cast<float4*>(&As[1][aPos]) = ra0;       // tile it+1 (a clamped copy past the end: unused)
    *reinterpret_cast<float4*>(&Bs[1][bPos0]) = rb00;
    *reinterpret_cast<float4*>(&Bs[1][bPos1]) = rb01;
    ra0 = ldA(it + 3); rb00 = ldB0(it + 3); rb01 = ldB1(it + 3);
    __syncthreads();
    if (it + 1 < a.nK) {
      mma(1);                                             // tile it+1
      *reinterpret_cast<float4*>(&As[0][aPos]) = ra1;     // tile it+2
      *reinterpret_cast<float4*>(&Bs[0][bPos0]) = rb10;
      *reinterpret_cast<float4*>(&Bs[0][bPos1]) = rb11;
      ra1 = ldA(it + 4); rb10 = ldB0(it + 4); rb11 = ldB1(it + 4);
      __syncthreads();
    }
  }
  // epilogue as k_mix (batched, branch-free: round 4): the wave's 32 x 32 tile through LDS (Bs is free after the last
  // barrier), 16-byte write-through stores
  float* stg = &Bs[0][0] + w * 1024;
  {
    const int jq = j >> 2, jr = j & 3, kb = kq & 1;
    int xe[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xe[e] = e * 32 + ((jq ^ e) << 2) + jr;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int base = (p * 16 + 4 * kq) * 32 + ((q ^ kb) << 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) stg[base + xe[e]] = acc[p][q][e];
      }
  }
  const bool wt = a.outFloats > 0 && a.outFloats < (1L << 29);   // 32-bit byte offsets
  float* outp = a.out + (size_t)part * a.outPartStride;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(outp, 0, wt ? (int)(a.outFloats * 4) : 0, 0x00020000);
  const int colTile = 2 * colPair + (w >> 1), wc = w & 1;
  float4 v4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int lrow = u * 8 + (lane >> 3), q = lane & 7;
    v4[u] = *reinterpret_cast<const float4*>(&stg[lrow * 32 + ((q ^ (lrow & 7)) << 2)]);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int lrow = u * 8 + (lane >> 3), q = lane & 7;
    const float4 v = v4[u];
    const int row = row0 + lrow;
    const int k = row / a.Np, n = row - k * a.Np;
    const bool ok = k < a.Ks && n < a.N;
    const size_t off = (size_t)colTile * a.sT + (size_t)n * a.sN + (size_t)k * a.sK + wc * 32 + q * 4;
    if (wt) {
      const u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
      __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, ok ? (int)(off * 4) : (int)0x7ffffff0, 0, 16);   // sc1; dropped when out of range
    } else if (ok) {
      *reinterpret_cast<float4*>(outp + off) = v;
    }
  }
}

// -------------------------------------------------------------------------------------------------
// 5b. the same graph mix with bf16 OPERANDS (BASELINE config 3's dtype): fp32 accumulation, fp32 inputs and outputs
// -------------------------------------------------------------------------------------------------
// Opt-in variant (matgcn_set_mix_precision(1); inference only, never the headline: narrower than the reference's fp32).
// The supports and the state stay fp32 in memory; they are rounded to bf16 (round to nearest even) on their way into
// LDS, two consecutive reduction indices packed into one 32-bit word, so that a lane's v_mfma_f32_16x16x16_bf16
// fragment (4 consecutive k of one row / column) is two ds_read_b32.  K-tile 32 = 16 packed rows of 64 words; packed
// row p is rotated by 16 * ((p >> 1) & 3) words, which puts the four k-quarters of a wave on four different bank
// quarters.  Everything else - 64 x 64 tile, 2 x 2 accumulators per wave, register prefetch two tiles ahead, XCD-aware
// tile order, the write-through row-store epilogue - is k_mix's.
typedef short bf16x4_t __attribute__((ext_vector_type(4)));
#define MFMA16BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned int bf16_rne(float x) {     // upper 16 bits of x, rounded to nearest even
  const unsigned int u = __float_as_uint(x);
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ uint4 pack_bf16_rows(const float4& k0, const float4& k1) {   // word j = (k0[j], k1[j])
  return make_uint4(bf16_rne(k0.x) | (bf16_rne(k1.x) << 16), bf16_rne(k0.y) | (bf16_rne(k1.y) << 16),
                    bf16_rne(k0.z) | (bf16_rne(k1.z) << 16), bf16_rne(k0.w) | (bf16_rne(k1.w) << 16));
}

template <int ROLE>
__global__ __launch_bounds__(256) void k_mix_bf16(MixArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned int As[2][16 * 64];
  __shared__ __attribute__((aligned(16))) unsigned int Bs[2][16 * 64];
  const int id = blockIdx.x;
  int colTile, rowTile;
  if ((a.nColTiles & 7) == 0) {
    const int xcd = id & 7, jj = id >> 3, cpx = a.nColTiles >> 3;
    rowTile = jj % a.nRowTiles;
    colTile = xcd * cpx + jj / a.nRowTiles;
  } else {
    rowTile = id % a.nRowTiles;
    colTile = id / a.nRowTiles;
  }
  const int row0 = rowTile * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w >> 1, wc = w & 1, j = lane & 15, kq = lane >> 4;
  const int pr = tid >> 4, sg = tid & 15;                  // staging: packed row pr (k = 2 pr, 2 pr + 1), 4-word segment sg
  const int kLast = 16 * a.nK - 1;                         // reduction indices 0 .. Np-1 (a.nK = Np / 16)
  const int nT = (a.nK + 1) >> 1;                          // K-tiles of 32
  const int stPos = pr * 64 + (((sg + 4 * ((pr >> 1) & 3)) & 15) << 2);
  const float* ap = a.St + row0 + sg * 4;
  const float* bp = a.X + (size_t)colTile * a.xTileStride + sg * 4;
  // loads are unconditional (row index clamped, values past the last reduction index zeroed by a select)
  auto ld = [&](const float* base, size_t ld_, int t, float4& r0, float4& r1) {
    const int k0 = 32 * min(t, nT - 1) + 2 * pr, k1 = k0 + 1;
    const float4 v0 = *reinterpret_cast<const float4*>(base + (size_t)min(k0, kLast) * ld_);
    const float4 v1 = *reinterpret_cast<const float4*>(base + (size_t)min(k1, kLast) * ld_);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    r0 = k0 <= kLast ? v0 : z;
    r1 = k1 <= kLast ? v1 : z;
  };
  float4 a0, a1, b0, b1, a2, a3, b2, b3;
  ld(ap, a.ldS, 0, a0, a1); ld(bp, a.ldX, 0, b0, b1);
  *reinterpret_cast<uint4*>(&As[0][stPos]) = pack_bf16_rows(a0, a1);
  *reinterpret_cast<uint4*>(&Bs[0][stPos]) = pack_bf16_rows(b0, b1);
  ld(ap, a.ldS, 1, a0, a1); ld(bp, a.ldX, 1, b0, b1);      // tile t+1 waits in (a0, a1, b0, b1) / (a2, a3, b2, b3)
  ld(ap, a.ldS, 2, a2, a3); ld(bp, a.ldX, 2, b2, b3);
  __syncthreads();
  f32x4 acc[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  // word of packed row (8 s + 2 kq + h) for row / column x: rotation 16 * kq
  const int rotA0 = (wr * 32 + j + 16 * kq) & 63, rotA1 = (wr * 32 + 16 + j + 16 * kq) & 63;
  const int rotB0 = (wc * 32 + j + 16 * kq) & 63, rotB1 = (wc * 32 + 16 + j + 16 * kq) & 63;
  auto frag = [&](const unsigned int* T, int s, int rot) {
    const unsigned int lo = T[(8 * s + 2 * kq) * 64 + rot], hi = T[(8 * s + 2 * kq + 1) * 64 + rot];
    bf16x4_t v;
    v[0] = (short)(lo & 0xffffu); v[1] = (short)(lo >> 16); v[2] = (short)(hi & 0xffffu); v[3] = (short)(hi >> 16);
    return v;
  };
  auto mma = [&](int cur) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {      // the two 16-wide k-steps of this K-tile
      const bf16x4_t fa0 = frag(As[cur], s, rotA0), fa1 = frag(As[cur], s, rotA1);
      const bf16x4_t fb0 = frag(Bs[cur], s, rotB0), fb1 = frag(Bs[cur], s, rotB1);
      acc[0][0] = MFMA16BF(fa0, fb0, acc[0][0]);
      acc[0][1] = MFMA16BF(fa0, fb1, acc[0][1]);
      acc[1][0] = MFMA16BF(fa1, fb0, acc[1][0]);
      acc[1][1] = MFMA16BF(fa1, fb1, acc[1][1]);
    }
  };
  for (int it = 0; it < nT; it += 2) {
    mma(0);
    *reinterpret_cast<uint4*>(&As[1][stPos]) = pack_bf16_rows(a0, a1);
    *reinterpret_cast<uint4*>(&Bs[1][stPos]) = pack_bf16_rows(b0, b1);
    ld(ap, a.ldS, it + 3, a0, a1); ld(bp, a.ldX, it + 3, b0, b1);
    __syncthreads();
    if (it + 1 < nT) {
      mma(1);
      *reinterpret_cast<uint4*>(&As[0][stPos]) = pack_bf16_rows(a2, a3);
      *reinterpret_cast<uint4*>(&Bs[0][stPos]) = pack_bf16_rows(b2, b3);
      ld(ap, a.ldS, it + 4, a2, a3); ld(bp, a.ldX, it + 4, b2, b3);
      __syncthreads();
    }
  }
  // epilogue: as k_mix (accumulator tile turned through LDS into 16-byte write-through row stores)
  float* stg = reinterpret_cast<float*>(w < 2 ? &As[0][0] : &Bs[0][0]) + (w & 1) * 1024;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lrow = p * 16 + 4 * kq + e, lcol = q * 16 + j;
        stg[lrow * 32 + (((lcol >> 2) ^ (lrow & 7)) << 2) + (lcol & 3)] = acc[p][q][e];
      }
  const bool wt = a.outFloats > 0 && a.outFloats < (1L << 29);   // 32-bit byte offsets
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, wt ? (int)(a.outFloats * 4) : 0, 0x00020000);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int lrow = u * 8 + (lane >> 3), q = lane & 7;
    const float4 v = *reinterpret_cast<const float4*>(&stg[lrow * 32 + ((q ^ (lrow & 7)) << 2)]);
    const int row = row0 + wr * 32 + lrow;
    const int k = row / a.Np, n = row - k * a.Np;
    if (k < a.Ks && n < a.N) {
      const size_t off = (size_t)colTile * a.sT + (size_t)n * a.sN + (size_t)k * a.sK + wc * 32 + q * 4;
      if (wt) {
        const u32x4 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, (int)(off * 4), 0, 16);   // aux 16 = sc1
      } else {
        *reinterpret_cast<float4*>(a.out + off) = v;
      }
    }
  }
}

// =================================================================================================
// 8. output head (MultiATGCN.py:416-418): Conv2d(T -> out*od, (1,H)) == [B*N x T*H] . [T*H x CH]
// =================================================================================================
// One workgroup of 4 waves per (b, 32-node tile): the waves split the T steps of the reduction (t = w, w+4, ..)
// so four times as many loads are in flight per tile, and add their partial tiles in LDS in a fixed order.
// A fragments come straight from the padded time-major sequence [T][B][Np][64] (each lane walks its node's
// 256-byte rows), B = fragment-ordered conv weight from L2.  out[b][o][n][dd] with channel ch = o*od + dd.
__global__ __launch_bounds__(256) void k_head(HeadArgs a) {
  __shared__ f32x16 part[3][2][64];      // partial accumulators of waves 1..3
  const int tilesPerB = (a.N + 31) >> 5;
  const int b = blockIdx.x / tilesPerB, n0 = (blockIdx.x % tilesPerB) * 32;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 31, half = lane >> 5;
  const int node = min(n0 + i, a.Np - 1);  // pad rows are zero / in bounds
  f32x16 acc[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
  const int nt = a.NTc;
  for (int t = w; t < a.T; t += 4) {
    const float* rowp = a.seq + (((size_t)t * a.B + b) * a.Np + node) * 64 + half * 4;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const float4 a4 = *reinterpret_cast<const float4*>(rowp + g * 8);
      const int jg = t * 8 + g;
      for (int tt = 0; tt < nt; ++tt) {
        const float4 w4 = *reinterpret_cast<const float4*>(a.w + ((size_t)(jg * nt + tt) * 64 + lane) * 4);
        acc[tt] = MFMA32(a4.x, w4.x, acc[tt]);
        acc[tt] = MFMA32(a4.y, w4.y, acc[tt]);
        acc[tt] = MFMA32(a4.z, w4.z, acc[tt]);
        acc[tt] = MFMA32(a4.w, w4.w, acc[tt]);
      }
    }
  }
  if (w > 0) { part[w - 1][0][lane] = acc[0]; part[w - 1][1][lane] = acc[1]; }
  __syncthreads();
  if (w > 0) return;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const f32x16 p0 = part[q][0][lane], p1 = part[q][1][lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] += p0[r]; acc[1][r] += p1[r]; }
  }
  // accumulator rows = nodes, cols = channels
  for (int tt = 0; tt < nt; ++tt) {
    const int ch = tt * 32 + i;
    if (ch >= a.CH) continue;
    const float bias = a.bias[ch];
    const int o = ch / a.od, dd = ch - o * a.od;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + acc_row(r, half);
      if (n < a.N) a.out[(((size_t)b * (a.CH / a.od) + o) * a.N + n) * a.od + dd] = acc[tt][r] + bias;
    }
  }
}

// =================================================================================================
// 9. loss / metric epilogue (MultiATGCN.py:422-427, loss.py:17-29, traffic_state_evaluator.py:87-104)
// =================================================================================================
// stage 1: one workgroup per (b, horizon): sum |p-l|*mask and sum mask over the N*od values of that slab
// (labelStart != null: y is the raw series (steps, N, yFeat), label row (b, o) = y[labelStart[b] + o])
__global__ __launch_bounds__(256) void k_mae_partial(const float* __restrict__ pred, const float* __restrict__ y,
                                                     const int* __restrict__ labelStart, int outSteps, int N, int od,
                                                     int ySteps, int yFeat, int yStart, float mean, float std,
                                                     float nullVal, float minS, float* __restrict__ partials) {
  __shared__ float sAbs[256], sCnt[256];
  const int b = blockIdx.x / outSteps, o = blockIdx.x - b * outSteps;
  const float* pp = pred + ((size_t)b * outSteps + o) * N * od;
  const size_t yrow = labelStart ? series_row((long)labelStart[b] + o, ySteps) : (size_t)b * ySteps + o;
  const float* yp = y + yrow * N * yFeat + yStart;
  const bool nanMask = nullVal != nullVal;
  float sa = 0.f, sc = 0.f;
  for (int idx = threadIdx.x; idx < N * od; idx += 256) {
    const int n = idx / od, c = idx - n * od;
    float l = yp[(size_t)n * yFeat + c] * std + mean;
    const float p = pp[idx] * std + mean;
    if (fabsf(l) < minS) l = 0.f;                       // loss.py:18 (the reference does this in place)
    const float m = nanMask ? (l == l ? 1.f : 0.f) : (l != nullVal ? 1.f : 0.f);
    const float d = fabsf(p - l) * m;
    sa += (d == d) ? d : 0.f;                           // nan -> 0 (loss.py:27)
    sc += m;
  }
  sAbs[threadIdx.x] = sa; sCnt[threadIdx.x] = sc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sAbs[threadIdx.x] += sAbs[threadIdx.x + s]; sCnt[threadIdx.x] += sCnt[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = sAbs[0]; partials[2 * blockIdx.x + 1] = sCnt[0]; }
}

// stage 2: one workgroup: thread o sums horizon o over the batch in fp64, thread 0 then the total
__global__ __launch_bounds__(64) void k_mae_final(const float* __restrict__ partials, int B, int outSteps,
                                                  float* __restrict__ result) {
  __shared__ double hAbs[64], hCnt[64];
  const int o = threadIdx.x;
  double sa = 0.0, sc = 0.0;
  if (o < outSteps)
    for (int b = 0; b < B; ++b) { sa += partials[2 * (b * outSteps + o)]; sc += partials[2 * (b * outSteps + o) + 1]; }
  hAbs[o] = sa; hCnt[o] = sc;
  if (o < outSteps) result[1 + o] = sc > 0.0 ? (float)(sa / sc) : 0.f;
  __syncthreads();
  if (o == 0) {
    double ta = 0.0, tc = 0.0;
    for (int k = 0; k < outSteps; ++k) { ta += hAbs[k]; tc += hCnt[k]; }
    result[0] = tc > 0.0 ? (float)(ta / tc) : 0.f;      // all-masked: mask/mean(mask) is nan -> 0 (loss.py:24-27)
    const_cast<float*>(partials)[2 * B * outSteps] = (float)tc;   // kept for the gradient (k_mae_grad)
  }
}

// ---- the evaluator's metric table (traffic_state_evaluator.py:46-121; group-std re-transform of
// traffic_state_executor.py:293-322) ---------------------------------------------------------------------------------
// One pass over prediction and label gives, per horizon, the MATGCN_METRIC_SUMS sums every metric of the table is a
// ratio of; they are accumulated in fp64 (R2 / EVAR subtract squares of means).  Element-wise arithmetic is fp32, like
// the reference's tensors.  stage 1: one workgroup per (b, horizon) -> partials[b][o][.]; stage 2 adds the batch up
// in a fixed order (run-to-run identical) and, on request, onto the running sums of earlier batches.
struct MetricArgs {
  const float* pred;     // (B, out, N, od)
  const float* y;        // (B, ySteps, N, yFeat) windows, or the raw series (ySteps, N, yFeat) when labelStart != null
  const int* labelStart;
  const float* mean; const float* std;     // first affine x*std + mean: one value, or N values (perNode); null = identity
  const float* mean2; const float* std2;   // second affine, per node (the group-std re-transform), or null
  int perNode;
  float clampMin;        // prediction below -> clampMin (NaN: off)
  float truthMin;        // only elements whose label exceeds it take part (NaN: off)
  float minS;            // |label| < minS -> 0 (loss.py:18; < 0: off)
  int outSteps, N, od, ySteps, yFeat, yStart;
  double* partials;      // [B*out][MATGCN_METRIC_SUMS]
};
constexpr int METRIC_SUMS = 14;
// sums: 0 elements, 1 non-NaN labels, 2 |d|, 3 d^2, 4 |d/l| (over 1), 5 labels != 0, 6-8 the same three over 5,
//       9 l, 10 l^2, 11 p, 12 p^2, 13 l-p
__global__ __launch_bounds__(256) void k_metric_partial(MetricArgs a) {
  __shared__ double red[4][METRIC_SUMS];
  const int b = blockIdx.x / a.outSteps, o = blockIdx.x - b * a.outSteps;
  const float* pp = a.pred + ((size_t)b * a.outSteps + o) * a.N * a.od;
  const size_t yrow = a.labelStart ? series_row((long)a.labelStart[b] + o, a.ySteps) : (size_t)b * a.ySteps + o;
  const float* yp = a.y + yrow * a.N * a.yFeat + a.yStart;
  double s[METRIC_SUMS];
#pragma unroll
  for (int k = 0; k < METRIC_SUMS; ++k) s[k] = 0.0;
  const bool clamp = a.clampMin == a.clampMin, pick = a.truthMin == a.truthMin;
  for (int idx = threadIdx.x; idx < a.N * a.od; idx += 256) {
    const int n = idx / a.od, c = idx - n * a.od;
    float l = yp[(size_t)n * a.yFeat + c], p = pp[idx];
    if (a.std) { const float sd = a.std[a.perNode ? n : 0], mu = a.mean[a.perNode ? n : 0]; l = l * sd + mu; p = p * sd + mu; }
    if (a.std2) { l = l * a.std2[n] + a.mean2[n]; p = p * a.std2[n] + a.mean2[n]; }
    if (clamp && p < a.clampMin) p = a.clampMin;
    if (pick && !(l > a.truthMin)) continue;
    if (a.minS >= 0.f && fabsf(l) < a.minS) l = 0.f;
    const float d = p - l, ad = fabsf(d), sq = d * d, ap = fabsf(d / l);
    const bool valid = l == l, nz = l != 0.f;          // NaN != 0 holds, as in labels.ne(0)
    s[0] += 1.0;
    if (valid) { s[1] += 1.0; s[2] += ad == ad ? ad : 0.f; s[3] += sq == sq ? sq : 0.f; s[4] += ap == ap ? ap : 0.f; }
    if (nz) { s[5] += 1.0; s[6] += ad == ad ? ad : 0.f; s[7] += sq == sq ? sq : 0.f; s[8] += ap == ap ? ap : 0.f; }
    const double ld = l, pd = p;
    s[9] += ld; s[10] += ld * ld; s[11] += pd; s[12] += pd * pd; s[13] += ld - pd;
  }
#pragma unroll
  for (int k = 0; k < METRIC_SUMS; ++k) {
    double v = s[k];
    for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_down(v, sh, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < METRIC_SUMS)
    a.partials[(size_t)blockIdx.x * METRIC_SUMS + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// stage 2: block o, thread k: sums[o][k] (+)= sum_b partials[b][o][k]
__global__ __launch_bounds__(64) void k_metric_accumulate(const double* __restrict__ partials, int B, int outSteps,
                                                          int accumulate, double* __restrict__ sums) {
  const int o = blockIdx.x, k = threadIdx.x;
  if (k >= METRIC_SUMS) return;
  double v = 0.0;
  for (int b = 0; b < B; ++b) v += partials[((size_t)b * outSteps + o) * METRIC_SUMS + k];
  sums[o * METRIC_SUMS + k] = (accumulate ? sums[o * METRIC_SUMS + k] : 0.0) + v;
}

// table[mode][o][m]: mode 0 "single" = horizon o alone, mode 1 "average" = horizons 0..o (traffic_state_evaluator.py:
// 46-121); m: MAE, MAPE, MSE, RMSE, masked_MAE, masked_MAPE, masked_MSE, masked_RMSE, R2, EVAR (the order of
// TrafficStateEvaluator.json).  swap: R2 / EVAR with prediction and truth exchanged, as traffic_state_executor.py:
// 318-319 hands them to sklearn.
__global__ __launch_bounds__(64) void k_metric_table(const double* __restrict__ sums, int outSteps, int swap,
                                                     double* __restrict__ table) {
  const int o = threadIdx.x;
  if (o >= outSteps) return;
  for (int mode = 0; mode < 2; ++mode) {
    double s[METRIC_SUMS];
    for (int k = 0; k < METRIC_SUMS; ++k) s[k] = 0.0;
    for (int q = mode ? 0 : o; q <= o; ++q)
      for (int k = 0; k < METRIC_SUMS; ++k) s[k] += sums[q * METRIC_SUMS + k];
    double* t = table + ((size_t)mode * outSteps + o) * 10;
    const double n = s[0];
    // mask / mean(mask) then mean: sum over the kept elements / their count; nothing kept -> nan -> 0 (loss.py:24-27)
    t[0] = s[1] > 0 ? s[2] / s[1] : 0.0;
    t[1] = s[1] > 0 ? s[4] / s[1] : 0.0;
    t[2] = s[1] > 0 ? s[3] / s[1] : 0.0;
    t[3] = sqrt(t[2]);
    t[4] = s[5] > 0 ? s[6] / s[5] : 0.0;
    t[5] = s[5] > 0 ? s[8] / s[5] : 0.0;
    t[6] = s[5] > 0 ? s[7] / s[5] : 0.0;
    t[7] = sqrt(t[6]);
    // sklearn r2_score(y_true, y_pred) = 1 - sum (t-p)^2 / sum (t - mean t)^2, explained_variance_score = 1 - Var(t-p)/Var(t)
    // (the residual sum leaves NaN labels out; sklearn refuses such input anyway)
    const double s1 = swap ? s[11] : s[9], s2 = swap ? s[12] : s[10];
    const double ssTot = n > 0 ? s2 - s1 * s1 / n : 0.0, res = s[3];
    const double varD = n > 0 ? res / n - (s[13] / n) * (s[13] / n) : 0.0;
    const double varT = n > 0 ? ssTot / n : 0.0;
    t[8] = ssTot != 0.0 ? 1.0 - res / ssTot : (res == 0.0 ? 1.0 : 0.0);
    t[9] = varT != 0.0 ? 1.0 - varD / varT : (varD == 0.0 ? 1.0 : 0.0);
  }
}

// gradient of result[0] w.r.t. pred: upstream * std * sign(p - l) * mask / sum(mask)   (torch: d|x| = sign(x), 0 at 0;
// terms the forward replaced by 0 - NaN differences - get no gradient)
__global__ __launch_bounds__(256) void k_mae_grad(const float* __restrict__ pred, const float* __restrict__ y,
                                                  const int* __restrict__ labelStart, int outSteps, int N, int od,
                                                  int ySteps, int yFeat, int yStart, float mean, float std,
                                                  float nullVal, float minS,
                                                  const float* __restrict__ count, const float* __restrict__ upstream,
                                                  size_t total, float* __restrict__ dpred) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = idx % od;
  const int n = (idx / od) % N;
  const int o = (idx / ((size_t)od * N)) % outSteps;
  const size_t b = idx / ((size_t)od * N * outSteps);
  const size_t yrow = labelStart ? series_row((long)labelStart[b] + o, ySteps) : b * ySteps + o;
  float l = y[(yrow * N + n) * yFeat + yStart + c] * std + mean;
  const float p = pred[idx] * std + mean;
  if (fabsf(l) < minS) l = 0.f;
  const bool nanMask = nullVal != nullVal;
  const float m = nanMask ? (l == l ? 1.f : 0.f) : (l != nullVal ? 1.f : 0.f);
  const float d = p - l;
  const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);      // NaN compares false both ways: 0
  const float cnt = count[0];
  dpred[idx] = cnt > 0.f ? upstream[0] * std * sgn * m / cnt : 0.f;
}
