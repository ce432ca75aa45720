// matgcn_node16.hip - node-wise contraction kernels of the recurrent step (included by matgcn_capi.hip).
//
// One workgroup (8 waves) per node n.  For that node the step needs
//     Y[b][o] = sum_kk A[b][kk] * W_n[kk][o],   A[b] = [ s[b][n][0:64] | G[n][b][0:Ks][0:64] ]
// (s = h for the gate AGCN, z*h for the update AGCN; G = graph-mixed s; MultiATGCN.py:106-108 restricted to the
// recurrent rows - the x rows live in the hoisted pre-activation PX).  All 64 batch rows of the node sit in LDS
// (80 KB at Ks = 4, so two workgroups share a CU and one's staging hides under the other's MFMAs); the
// node-adaptive weights - the one big stream, 172 KB per node and step - go straight from L2/Infinity Cache into
// each wave's registers in v_mfma_f32_16x16x4_f32 B-fragment order, four k-groups ahead, and are read exactly
// once per workgroup.
//
// MFMA 16x16x4 f32 operand maps: A lane l -> A[row = l&15][k = l>>4], B lane l -> B[k = l>>4][col = l&15],
// C/D lane l, reg e -> C[row = 4*(l>>4) + e][col = l&15].  A k-group is 16 reduction indices; MFMA step s of a
// group uses k = 16g + 4*(l>>4) + s, so both fragments of a group are one aligned float4 per lane.
//
// LDS tile layout: rows of 16-byte slots; slot q of row r is stored at position q ^ (r & 15) inside its
// 16-slot block, which makes the ds_read_b128 of lane (row, kq) conflict-free without padding.
#ifndef MATGCN_NODE16_HIP
#define MATGCN_NODE16_HIP

// f32x4 / MFMA16 come from matgcn_kernels.hip (same translation unit)

#ifndef N16_RING
#define N16_RING 10   // k-groups of weights in flight per wave (10 KB): covers an Infinity-Cache round trip
#endif
#ifndef U16_RING
#define U16_RING 6    // the same for k_update16: its waves stream half as many bytes per k-group, and the
                      // residual-cell operands need the registers (10 spills)
#endif

struct Node16Args {
  const float* s;        // [rows][Np][64]: h (gate / res-only) or z*h (update)
  const float* g;        // [N][rows][Ks][64] graph-mixed s
  long gNodeStride;      // floats between the nodes of g; 0 = rows*Ks*64 (the block may sit inside a larger one)
  const float* w;        // [N][nG][OT][64][4] recurrent rows of the node-adaptive weights (fragment order)
  const float* px;       // [N][rows][192] hoisted pre-activation of this step (x rows + bias): gate 0:128, update
                         // 128:192 - layers >= 1; null for layer 0, whose narrow x part is contracted in the kernel:
  const float* xa;       // [N][rows][16*nGx] folded x rows of this step [x | mix_k(x) | 1 | 0..] (layer 0) or null
  int nGx;               // k-groups of the x part (weights: groups nG .. nG+nGx-1 of the node's stream)
  int rows, N, Np, Ks;
  // gate
  float* zh;             // out [rows][Np][64]  z*h
  float* r;              // gate: out / update: in  [N][rows][64]
  float* raw;            // optional (rows, N, 128) pre-activation dump (unit entry point)
  // update
  const float* h;        // [rows][Np][64] previous state (blend input)
  float* hout;           // [rows][Np][64] new state (may alias h)
  // residual GRU cell + blend (MultiATGCN.py:142-150, 205-208)
  const float* xt;       // x_t rows: xt[b*xRowStride + n*C + c]
  long xRowStride;
  int C, Cpad;           // input channels of the layer, padded to 16
  const float* rg;       // [K1/16][8][64][4] fragment-ordered res gate weight, rows [x (Cpad) | h' (64)]
  const float* rgb;      // (128)
  const float* ru;       // [K1/16][4][64][4]
  const float* rub;      // (64)
  const float* blend;    // &weights_gru[l][t] or null
  float* seq;            // Seq_l[:, t] or null: seq[b*seqRowStride + n*64 + o]
  long seqRowStride;
  // training (SAVE instantiations): activations of this step kept for the backward, each [rows][Np][64]
  float *svZ, *svR, *svHC, *svZ2, *svR2, *svHC2;
};

__device__ __forceinline__ float sigmoid16(float x) { return 1.0f / (1.0f + expf(-x)); }

// keep v where ok, else zeros - element-wise, so the float4 stays in registers (a ?: on the structs would
// select between their addresses and push them to scratch)
__device__ __forceinline__ float4 keep4(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

typedef unsigned int u32x4_n16 __attribute__((ext_vector_type(4)));

// 16-byte WRITE-THROUGH (sc1) store: the outputs of a step kernel are the next kernel's inputs, so they should
// leave the L2 while this kernel still runs instead of as a dirty-line flush at its end (scalar sc1 stores would
// cost a fabric write each: outputs are therefore turned into whole 256-byte rows in LDS first)
// `base` must be wave-uniform (the descriptor lives in SGPRs); `off` is this lane's float offset (< 2^29)
__device__ __forceinline__ void store_wt16(float* base, size_t off, const float4& v) {
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);
  const u32x4_n16 bits = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(bits, rsrc, (int)(off * 4), 0, 16);   // aux 16 = sc1
}

// position (in floats) of element (row, col) of a swizzled [.][16*blocks slots] tile with `spr` slots per row
__device__ __forceinline__ int swz(int row, int col, int spr) {
  const int slot = col >> 2;
  return (row * spr + ((slot & ~15) | ((slot ^ row) & 15))) * 4 + (col & 3);
}

// stage the 64-row A tile of node n: Hs <- s rows (16 slots), Gs <- G rows (16*Ks slots); rows >= a.rows are zero.
// All loads of a round are issued before the first LDS write and none sits behind a branch (a predicated load
// would make the compiler wait for each one separately): out-of-range rows are clamped and zeroed by a select.
__device__ __forceinline__ void stage_node_tile(const Node16Args& a, int n, int rowBase, float* Hs, float* Gs) {
  const int tid = threadIdx.x;
  const int row = tid >> 4, q = tid & 15;          // 32 rows x 16 slots per sweep
  const int rA = row, rB = row + 32;
  const bool vA = rowBase + rA < a.rows, vB = rowBase + rB < a.rows;
  const int gA = min(rowBase + rA, a.rows - 1), gB = min(rowBase + rB, a.rows - 1);
  float4 hA = *reinterpret_cast<const float4*>(a.s + ((size_t)gA * a.Np + n) * 64 + q * 4);
  float4 hB = *reinterpret_cast<const float4*>(a.s + ((size_t)gB * a.Np + n) * 64 + q * 4);
  const int spr = 16 * a.Ks;
  const float4* gsrc = reinterpret_cast<const float4*>(
      a.g + (size_t)n * (a.gNodeStride ? (size_t)a.gNodeStride : (size_t)a.rows * spr * 4));
  const int pA = q ^ (rA & 15), pB = q ^ (rB & 15);
  if (a.Ks == 0) {   // no dense support left (all folded): only the s slots
    *reinterpret_cast<float4*>(&Hs[(rA * 16 + pA) * 4]) = keep4(vA, hA);
    *reinterpret_cast<float4*>(&Hs[(rB * 16 + pB) * 4]) = keep4(vB, hB);
  }
  for (int k0 = 0; k0 < a.Ks; k0 += 4) {
    float4 vAk[4], vBk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = min(k0 + kk, a.Ks - 1);
      vAk[kk] = gsrc[(size_t)gA * spr + k * 16 + q];
      vBk[kk] = gsrc[(size_t)gB * spr + k * 16 + q];
    }
    if (k0 == 0) {
      *reinterpret_cast<float4*>(&Hs[(rA * 16 + pA) * 4]) = keep4(vA, hA);
      *reinterpret_cast<float4*>(&Hs[(rB * 16 + pB) * 4]) = keep4(vB, hB);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (k0 + kk < a.Ks) {
        *reinterpret_cast<float4*>(&Gs[(rA * spr + (k0 + kk) * 16 + pA) * 4]) = keep4(vA, vAk[kk]);
        *reinterpret_cast<float4*>(&Gs[(rB * spr + (k0 + kk) * 16 + pB) * 4]) = keep4(vB, vBk[kk]);
      }
    }
  }
}

// A fragment of row tile rt for k-group g (g < 4: the s slots, else the mixed slots)
__device__ __forceinline__ float4 a_frag(const float* Hs, const float* Gs, int Ks, int rt, int g, int i, int kq) {
  const int row = rt * 16 + i;
  if (g < 4) return *reinterpret_cast<const float4*>(&Hs[(row * 16 + ((4 * g + kq) ^ i)) * 4]);
  const int q = 4 * (g - 4) + kq;
  return *reinterpret_cast<const float4*>(&Gs[(row * 16 * Ks + ((q & ~15) | ((q ^ i) & 15))) * 4]);
}

// the 16 MFMAs of one k-group: 4 row tiles x 4 k-steps, accumulators rotate so that a chain is revisited every
// fourth instruction
__device__ __forceinline__ void mfma_group(const float4 (&av)[4], const float4& wv, f32x4 (&acc)[4]) {
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].x, wv.x, acc[rt]);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].y, wv.y, acc[rt]);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].z, wv.z, acc[rt]);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc[rt] = MFMA16(av[rt].w, wv.w, acc[rt]);
}

// layer-0 x part: acc[rt] += XA[rows of tile rt][16 gx .. +16] . Wx[gx]; A fragments come straight from global
// memory (a row of XA is 64*nGx bytes, a 16-row tile is contiguous), weights from the tail of the node's stream
__device__ __forceinline__ void x_groups(const Node16Args& a, int n, int rowBase, const float4* wx, int gStride, int i,
                                         int kq, f32x4 (&acc)[4]) {
  const int kx = 16 * a.nGx;
  const float* base = a.xa + (size_t)n * a.rows * kx + kq * 4;
  for (int gx = 0; gx < a.nGx; ++gx) {
    const float4 wv = wx[(size_t)gx * gStride];
    float4 av[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
      av[rt] = *reinterpret_cast<const float4*>(base + (size_t)min(rowBase + rt * 16 + i, a.rows - 1) * kx + gx * 16);
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

// ---- gate AGCN + sigmoid + z*h (MultiATGCN.py:122-125) -----------------------------------------------------
template <bool SAVE>
__global__ __launch_bounds__(512, 4) void k_gate16(Node16Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds;               // [64][16 slots]
  float* Gs = lds + 64 * 64;     // [64][16*Ks slots]
  const int n = blockIdx.x, rowBase = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const int nG = 4 * (1 + a.Ks);
  // weight stream of this wave: column tile w (of 8)
  const float4* wp = reinterpret_cast<const float4*>(a.w) + ((size_t)n * (nG + a.nGx) * 8 + w) * 64 + lane;
  stage_node_tile(a, n, rowBase, Hs, Gs);   // requested first: the MFMAs cannot start without the tile
  float4 wr[N16_RING];
#pragma unroll
  for (int r = 0; r < N16_RING; ++r) wr[r] = wp[(size_t)min(r, nG - 1) * 8 * 64];
  // accumulators start from the hoisted pre-activation (x rows + bias), fetched while the tile lands;
  // layer 0 instead contracts its narrow x part here, straight from global memory, before the tile is needed
  const int o = 16 * w + j;
  f32x4 acc[4];
  if (a.px) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int b = min(rowBase + rt * 16 + 4 * kq + e, a.rows - 1);
        acc[rt][e] = a.px[((size_t)n * a.rows + b) * 192 + o];
      }
  } else {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    x_groups(a, n, rowBase, wp + (size_t)nG * 8 * 64, 8 * 64, j, kq, acc);
  }
  __syncthreads();
  // k-groups in pairs: the A fragments of a group are read from LDS while the MFMAs of the group before it run
  // (two named fragment sets ping-pong; nG is even)
  float4 avA[4], avB[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) avA[rt] = a_frag(Hs, Gs, a.Ks, rt, 0, j, kq);
  for (int g0 = 0; g0 < nG; g0 += N16_RING) {
#pragma unroll
    for (int r = 0; r < N16_RING; r += 2) {
      const int g = g0 + r;
      const float4 w0 = wr[r], w1 = wr[r + 1];
      wr[r] = wp[(size_t)min(g + N16_RING, nG - 1) * 8 * 64];
      wr[r + 1] = wp[(size_t)min(g + 1 + N16_RING, nG - 1) * 8 * 64];
      if (g < nG) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) avB[rt] = a_frag(Hs, Gs, a.Ks, rt, g + 1, j, kq);
        mfma_group(avA, w0, acc);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) avA[rt] = a_frag(Hs, Gs, a.Ks, rt, min(g + 2, nG - 1), j, kq);
        mfma_group(avB, w1, acc);
      }
    }
  }
  // epilogue: zr = sigmoid(.), z*h and r gathered as a [64 rows][z*h 64 | r 64] tile in LDS (the mixed slots are
  // dead once every wave has left the K loop), then written out as whole 256-byte rows
  __syncthreads();
  float* Out = Gs;                                  // [64][32 slots], same XOR swizzle as the tiles
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int lb = rt * 16 + 4 * kq + e, b = rowBase + lb;
      const float v = acc[rt][e];
      if (a.raw && b < a.rows) a.raw[((size_t)b * a.N + n) * 128 + o] = v;
      const float sg = sigmoid16(v);
      if (SAVE && b < a.rows) ((w < 4) ? a.svZ : a.svR)[((size_t)b * a.Np + n) * 64 + (o & 63)] = sg;
      Out[swz(lb, o, 32)] = (w < 4) ? sg * Hs[swz(lb, o, 16)] : sg;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {                  // 64 rows x 32 slots = 2048 float4 over 512 threads
    const int idx = tid + 512 * it;
    const int lb = idx >> 5, q = idx & 31, b = rowBase + lb;
    if (b >= a.rows) continue;
    const float4 v = *reinterpret_cast<const float4*>(&Out[(lb * 32 + ((q & ~15) | ((q ^ lb) & 15))) * 4]);
    if (q < 16) store_wt16(a.zh, ((size_t)b * a.Np + n) * 64 + q * 4, v);
    else store_wt16(a.r, ((size_t)n * a.rows + b) * 64 + (q - 16) * 4, v);
  }
}

// ---- hoisted x part of layers >= 1: PX[t][n][b][0:192] = bias[n] + [x | mix_k(x)] . Wx[n] --------------------
// (MultiATGCN.py:106-108 restricted to the x rows; gate columns 0:128, update columns 128:192.)  Same structure as
// the gate kernel - 64-row tile [x | G] of one node in LDS, weights streamed once per workgroup - with 12 column
// tiles over 8 waves: waves 0-3 take two tiles, waves 4-7 one, i.e. three per SIMD.  rows = B * (steps of the
// chunk), row -> (t, b) t-major; workgroup ids of one node's row blocks are 8 apart (same XCD, same time: the
// second and later blocks read the node's weights from that XCD's L2).
#ifndef PX16_RING
#define PX16_RING 4   // two rings (two column tiles per wave); 6 spills to scratch (-11 %)
#endif
struct Px16Args {
  const float* x;        // [rows][Np][64] input rows of the chunk (layer below, time-major)
  const float* g;        // [N][rows][Ks][64] graph-mixed input rows
  const float* w;        // [N][nG][12][64][4] x rows of both AGCNs (fragment order)
  const float* bias;     // [N][192]
  float* pxOut;          // [Tc][N][B][192] slice of PX
  int rows, N, Np, Ks, B;
};
__global__ __launch_bounds__(512, 4) void k_px16(Px16Args p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds;
  float* Gs = lds + 64 * 64;
  const int RB = (p.rows + 63) >> 6;
  const int id = blockIdx.x;
  const int grp = id / (8 * RB), rem = id - grp * 8 * RB;
  const int n = grp * 8 + (rem & 7), rowBase = (rem >> 3) * 64;
  if (n >= p.N) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const int nG = 4 * (1 + p.Ks);
  const bool two = w < 4;                       // this wave also owns column tile w + 8
  const float4* wp0 = reinterpret_cast<const float4*>(p.w) + ((size_t)n * nG * 12 + w) * 64 + lane;
  const float4* wp1 = reinterpret_cast<const float4*>(p.w) + ((size_t)n * nG * 12 + min(w + 8, 11)) * 64 + lane;
  Node16Args a;                                 // staging helper speaks Node16Args
  a.s = p.x; a.g = p.g; a.gNodeStride = 0; a.rows = p.rows; a.Np = p.Np; a.Ks = p.Ks;
  stage_node_tile(a, n, rowBase, Hs, Gs);
  float4 wr0[PX16_RING], wr1[PX16_RING];
#pragma unroll
  for (int r = 0; r < PX16_RING; ++r) {
    wr0[r] = wp0[(size_t)min(r, nG - 1) * 12 * 64];
    wr1[r] = wp1[(size_t)min(r, nG - 1) * 12 * 64];
  }
  const int o0 = 16 * w + j, o1 = 16 * (w + 8) + j;
  const float b0 = p.bias[(size_t)n * 192 + o0], b1 = two ? p.bias[(size_t)n * 192 + o1] : 0.f;
  f32x4 acc0[4], acc1[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) { acc0[rt] = f32x4{b0, b0, b0, b0}; acc1[rt] = f32x4{b1, b1, b1, b1}; }
  __syncthreads();
  for (int g0 = 0; g0 < nG; g0 += PX16_RING) {
#pragma unroll
    for (int r = 0; r < PX16_RING; ++r) {
      const int g = g0 + r;
      const float4 wv0 = wr0[r], wv1 = wr1[r];
      wr0[r] = wp0[(size_t)min(g + PX16_RING, nG - 1) * 12 * 64];
      wr1[r] = wp1[(size_t)min(g + PX16_RING, nG - 1) * 12 * 64];
      if (g < nG) {
        float4 av[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) av[rt] = a_frag(Hs, Gs, p.Ks, rt, g, j, kq);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc0[rt] = MFMA16(av[rt].x, wv0.x, acc0[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc0[rt] = MFMA16(av[rt].y, wv0.y, acc0[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc0[rt] = MFMA16(av[rt].z, wv0.z, acc0[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc0[rt] = MFMA16(av[rt].w, wv0.w, acc0[rt]);
        if (two) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) acc1[rt] = MFMA16(av[rt].x, wv1.x, acc1[rt]);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) acc1[rt] = MFMA16(av[rt].y, wv1.y, acc1[rt]);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) acc1[rt] = MFMA16(av[rt].z, wv1.z, acc1[rt]);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) acc1[rt] = MFMA16(av[rt].w, wv1.w, acc1[rt]);
        }
      }
    }
  }
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = rowBase + rt * 16 + 4 * kq + e;
      if (row >= p.rows) continue;
      const int t = row / p.B, b = row - t * p.B;
      float* dst = p.pxOut + (((size_t)t * p.N + n) * p.B + b) * 192;
      dst[o0] = acc0[rt][e];
      if (two) dst[o1] = acc1[rt][e];
    }
}

// ---- update AGCN + tanh + GRU blend, fused with the residual GRU cell and the per-step blend ------------------
// MODE 0: ATGRU update only (h' out); 1: update + residual cell (+ blend); 2: residual cell only on s (unit entry)
//
// Waves: (ct = w&3, kh = w>>2).  The update GEMM (O = 64: 4 column tiles) splits K in two halves over the wave
// pairs so that every weight fragment is still fetched exactly once; the halves meet in LDS.  The residual cell
// then runs two small GEMMs on tiles that never leave LDS.  Every global operand of a later phase is requested
// before the barrier of the phase in front of it, so its latency hides under that phase.
template <int MODE, bool SAVE>
__global__ __launch_bounds__(512, 4) void k_update16(Node16Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds;               // [64][16 slots]: z*h during the update GEMM, then h'
  float* Gs = lds + 64 * 64;     // [64][16*Ks slots]; reused afterwards (>= 64 KB is allocated):
  float* Red = Gs;               //   [4 ct][4 rt][64 lanes][4]  K-half partial sums          16 KB
  float* ZH2 = Gs + 4096;        //   [64][16 slots] z2*h'                                      16 KB
  float* R2 = Gs + 2 * 4096;     //   [64][16 slots] r2                                         16 KB
  float* XT = Gs + 3 * 4096;     //   [64][16 slots] x_t (zero padded)                          16 KB
  const int n = blockIdx.x, rowBase = blockIdx.y * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kq = lane >> 4;
  const int ct = w & 3, kh = w >> 2;
  const int srow = tid >> 4, sq = tid & 15;   // staging coordinates: 32 rows x 16 slots per sweep

  f32x4 acc[4];
  if (MODE != 2) {
    const int nG = 4 * (1 + a.Ks), nGh = nG >> 1;   // nG is even: each K half is nGh groups
    const int gBeg = kh * nGh;
    const float4* wp = reinterpret_cast<const float4*>(a.w) + ((size_t)n * (nG + a.nGx) * 4 + ct) * 64 + lane;
    stage_node_tile(a, n, rowBase, Hs, Gs);   // requested first: the MFMAs cannot start without the tile
    float4 wr[U16_RING];
#pragma unroll
    for (int r = 0; r < U16_RING; ++r) wr[r] = wp[(size_t)(gBeg + min(r, nGh - 1)) * 4 * 64];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!a.px && kh == 0) x_groups(a, n, rowBase, wp + (size_t)nG * 4 * 64, 4 * 64, j, kq, acc);   // layer 0
    __syncthreads();
    // k-groups in pairs, A fragments one group ahead of the MFMAs (see k_gate16); nGh is even
    float4 avA[4], avB[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) avA[rt] = a_frag(Hs, Gs, a.Ks, rt, gBeg, j, kq);
    for (int g0 = 0; g0 < nGh; g0 += U16_RING) {
#pragma unroll
      for (int r = 0; r < U16_RING; r += 2) {
        const int gl = g0 + r;
        const float4 w0 = wr[r], w1 = wr[r + 1];
        wr[r] = wp[(size_t)(gBeg + min(gl + U16_RING, nGh - 1)) * 4 * 64];
        wr[r + 1] = wp[(size_t)(gBeg + min(gl + 1 + U16_RING, nGh - 1)) * 4 * 64];
        if (gl < nGh) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) avB[rt] = a_frag(Hs, Gs, a.Ks, rt, gBeg + gl + 1, j, kq);
          mfma_group(avA, w0, acc);
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) avA[rt] = a_frag(Hs, Gs, a.Ks, rt, gBeg + min(gl + 2, nGh - 1), j, kq);
          mfma_group(avB, w1, acc);
        }
      }
    }
  }

  // ---- operands of the next phases, requested now ----
  const int o4 = 16 * ct + j;                      // column of this lane in a 64-wide tile
  float pxv[4][4], rv[4][4], hv[4][4];             // blend operands (kh == 0 waves use them)
  if (MODE != 2) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int b = min(rowBase + rt * 16 + 4 * kq + e, a.rows - 1);
        pxv[rt][e] = a.px ? a.px[((size_t)n * a.rows + b) * 192 + 128 + o4] : 0.f;
        rv[rt][e] = a.r[((size_t)n * a.rows + b) * 64 + o4];
        hv[rt][e] = a.h[((size_t)b * a.Np + n) * 64 + o4];
      }
  }
  const int ngx = a.Cpad >> 4;                     // x groups of the residual GEMMs (1 or 4)
  const int nG1 = ngx + 4;                         // <= 8
  float4 xv[2];
  float xs[2][4];
  if (MODE != 0) {
    if (a.C == 64) {
#pragma unroll
      for (int it = 0; it < 2; ++it)
        xv[it] = *reinterpret_cast<const float4*>(a.xt + (size_t)min(rowBase + srow + 32 * it, a.rows - 1) * a.xRowStride +
                                                  (size_t)n * 64 + sq * 4);
    } else {
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int c = min(sq * 4 + cc, a.C - 1);
          xs[it][cc] = a.xt[(size_t)min(rowBase + srow + 32 * it, a.rows - 1) * a.xRowStride + (size_t)n * a.C + c];
        }
    }
  }

  if (MODE != 2) {
    __syncthreads();   // every wave is done with the mixed slots: Gs becomes scratch
    if (kh == 1) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        *reinterpret_cast<f32x4*>(&Red[((ct * 4 + rt) * 64 + lane) * 4]) = acc[rt];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(&Red[((ct * 4 + rt) * 64 + lane) * 4]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int lb = rt * 16 + 4 * kq + e, b = rowBase + lb;
          const float hc = tanhf(acc[rt][e] + p[e] + pxv[rt][e]);
          const float rr = rv[rt][e];
          float hn = rr * hv[rt][e] + (1.0f - rr) * hc;   // (MultiATGCN.py:127: r blends, z gated the candidate)
          if (SAVE && b < a.rows) a.svHC[((size_t)b * a.Np + n) * 64 + o4] = hc;
          if (b >= a.rows) hn = 0.f;
          if (MODE == 0) { if (b < a.rows) a.hout[((size_t)b * a.Np + n) * 64 + o4] = hn; }
          else Hs[swz(lb, o4, 16)] = hn;                  // h' tile for the residual cell (z*h no longer needed)
        }
      }
    }
    if (MODE == 0) return;
  } else {
    // residual cell only: h' := s rows
    float4 sv[2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
      sv[it] = *reinterpret_cast<const float4*>(a.s + ((size_t)min(rowBase + srow + 32 * it, a.rows - 1) * a.Np + n) * 64 + sq * 4);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int rr = srow + 32 * it;
      *reinterpret_cast<float4*>(&Hs[(rr * 16 + (sq ^ (rr & 15))) * 4]) = keep4(rowBase + rr < a.rows, sv[it]);
    }
  }

  // ---- residual GRU cell on [x_t | h'] (MultiATGCN.py:142-150) ----
#pragma unroll
  for (int it = 0; it < 2; ++it) {   // x_t tile (zero padded to Cpad)
    const int rr = srow + 32 * it;
    const bool ok = rowBase + rr < a.rows;
    if (a.C == 64) {
      *reinterpret_cast<float4*>(&XT[(rr * 16 + (sq ^ (rr & 15))) * 4]) = keep4(ok, xv[it]);
    } else if (sq < (a.Cpad >> 2)) {
      float e4[4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) e4[cc] = (ok && sq * 4 + cc < a.C) ? xs[it][cc] : 0.f;
      *reinterpret_cast<float4*>(&XT[(rr * 16 + (sq ^ (rr & 15))) * 4]) = make_float4(e4[0], e4[1], e4[2], e4[3]);
    }
  }
  // weights of both residual GEMMs (shared by all nodes, L2-resident), requested before the tile barrier
  const int rp = kh;
  float4 rgv[8], ruv[8];
  {
    const float4* rgp = reinterpret_cast<const float4*>(a.rg) + (size_t)w * 64 + lane;
#pragma unroll
    for (int g = 0; g < 8; ++g) rgv[g] = rgp[(size_t)min(g, nG1 - 1) * 8 * 64];
    const float4* rup = reinterpret_cast<const float4*>(a.ru) + (size_t)ct * 64 + lane;
#pragma unroll
    for (int g = 0; g < 8; ++g) ruv[g] = rup[(size_t)min(g, nG1 - 1) * 4 * 64];
  }
  const float bg = a.rgb[16 * w + j], bu = a.rub[o4];
  const float gate = a.blend ? sigmoid16(a.blend[0]) : 0.f;   // g = sigmoid(weights_gru[l][t]) (:208)
  __syncthreads();
  // GEMM 1: zr2 = sigmoid([x|h'] Wg + bg): wave w = column tile w (of 8), 4 row tiles
  f32x4 acc1[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) acc1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g < nG1) {
      const float* T = (g < ngx) ? XT : Hs;
      const int gg = (g < ngx) ? g : g - ngx;
      const float4 wv = rgv[g];
      float4 av[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) av[rt] = *reinterpret_cast<const float4*>(&T[((rt * 16 + j) * 16 + ((4 * gg + kq) ^ j)) * 4]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc1[rt] = MFMA16(av[rt].x, wv.x, acc1[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc1[rt] = MFMA16(av[rt].y, wv.y, acc1[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc1[rt] = MFMA16(av[rt].z, wv.z, acc1[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc1[rt] = MFMA16(av[rt].w, wv.w, acc1[rt]);
    }
  }
  {
    const int o = 16 * w + j;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lb = rt * 16 + 4 * kq + e;
        const float sg = sigmoid16(acc1[rt][e] + bg);
        if (SAVE && rowBase + lb < a.rows)
          ((w < 4) ? a.svZ2 : a.svR2)[((size_t)(rowBase + lb) * a.Np + n) * 64 + (o & 63)] = sg;
        if (w < 4) ZH2[swz(lb, o, 16)] = sg * Hs[swz(lb, o, 16)];
        else R2[swz(lb, o - 64, 16)] = sg;
      }
  }
  __syncthreads();
  // GEMM 2: hc2 = tanh([x | z2*h'] Wu + bu): wave (ct, rp) -> column tile ct, row tiles 2rp, 2rp+1
  f32x4 acc2[2];
  acc2[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  acc2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g < nG1) {
      const float* T = (g < ngx) ? XT : ZH2;
      const int gg = (g < ngx) ? g : g - ngx;
      const float4 wv = ruv[g];
      float4 av[2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
        av[q] = *reinterpret_cast<const float4*>(&T[(((2 * rp + q) * 16 + j) * 16 + ((4 * gg + kq) ^ j)) * 4]);
#pragma unroll
      for (int q = 0; q < 2; ++q) acc2[q] = MFMA16(av[q].x, wv.x, acc2[q]);
#pragma unroll
      for (int q = 0; q < 2; ++q) acc2[q] = MFMA16(av[q].y, wv.y, acc2[q]);
#pragma unroll
      for (int q = 0; q < 2; ++q) acc2[q] = MFMA16(av[q].z, wv.z, acc2[q]);
#pragma unroll
      for (int q = 0; q < 2; ++q) acc2[q] = MFMA16(av[q].w, wv.w, acc2[q]);
    }
  }
  // the new state is gathered as a [64][64] tile in LDS (over x_t, dead once every wave has left GEMM 2) and
  // written to the state and to Seq_l[t] as whole 256-byte rows
  __syncthreads();
  float* Out = XT;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int lb = (2 * rp + q) * 16 + 4 * kq + e;
      const float hc = tanhf(acc2[q][e] + bu);
      if (SAVE && rowBase + lb < a.rows) a.svHC2[((size_t)(rowBase + lb) * a.Np + n) * 64 + o4] = hc;
      const float hp = Hs[swz(lb, o4, 16)];
      const float rr = R2[swz(lb, o4, 16)];
      const float res = rr * hp + (1.0f - rr) * hc;
      Out[swz(lb, o4, 16)] = a.blend ? (gate * hp + (1.0f - gate) * res) : res;
    }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {                  // 64 rows x 16 slots = 1024 float4 over 512 threads
    const int lb = srow + 32 * it, b = rowBase + lb;
    if (b >= a.rows) continue;
    const float4 v = *reinterpret_cast<const float4*>(&Out[(lb * 16 + (sq ^ (lb & 15))) * 4]);
    store_wt16(a.hout, ((size_t)b * a.Np + n) * 64 + sq * 4, v);
    if (a.seq) store_wt16(a.seq, (size_t)b * a.seqRowStride + (size_t)n * 64 + sq * 4, v);
  }
}

// ---- parameter-only: the node-adaptive weight streams are written by k_prep_stream (matgcn_kernels.hip) ----

// nn.Linear weight (O, I) -> [g][ct][lane][4] of B[kk][o] = W[o][in(kk)]: rows kk < Cpad map to input kk (zero
// beyond C), rows kk >= Cpad map to input C + (kk - Cpad)
__global__ __launch_bounds__(256) void k_prep_linear16(const float* __restrict__ W, int I, int O, int C, int Cpad,
                                                       int nG, float* __restrict__ out) {
  const int unit = blockIdx.x * 256 + threadIdx.x;
  const int OT = O >> 4;
  if (unit >= nG * OT * 64) return;
  const int lane = unit & 63, ct = (unit >> 6) % OT, g = (unit >> 6) / OT;
  const int o = 16 * ct + (lane & 15);
  float v[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int kk = 16 * g + 4 * (lane >> 4) + s;
    int in = -1;
    if (kk < Cpad) { if (kk < C) in = kk; } else { in = C + (kk - Cpad); }
    v[s] = (in >= 0 && in < I) ? W[(size_t)o * I + in] : 0.f;
  }
  *reinterpret_cast<float4*>(out + (size_t)unit * 4) = make_float4(v[0], v[1], v[2], v[3]);
}

#endif
