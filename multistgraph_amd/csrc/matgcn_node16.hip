// matgcn_node16.hip - node-wise contraction kernels of the recurrent step (included by matgcn_capi.hip).
//
// One workgroup (8 waves) per node n and 64-row block of the batch.  For that node the step needs
//     Y[b][o] = sum_kk A[b][kk] * W_n[kk][o],   A[b] = [ s[b][n][0:64] | G[n][b][0:Ks][0:64] ]
// (s = h for the gate AGCN, z*h for the update AGCN; G = graph-mixed s; MultiATGCN.py:106-108 restricted to the
// recurrent rows - the x rows live in the hoisted pre-activation PX).
//
// The kernels sit at the ridge of the machine (32 FLOP per weight byte; ~100 MB per launch against ~2 GFLOP), so
// what decides their time is whether the weight stream and the matrix pipe run AT THE SAME TIME.  Round 1 staged the
// whole 64 x 256 A tile (64 KB) plus a ten-deep weight ring before the first MFMA: memory time and MFMA time added
// up (21 + 17 us measured, tools/nodelab.hip).  Now the K range is walked in CHUNKS of 64 (one stack slot):
//   chunk 0 = the 64 state rows (kept in LDS to the end: z*h / the blend need them), chunks 1..Ks = the mixed slots,
//   ping-ponging through two 16 KB LDS buffers that are fed from registers two chunks ahead of the MFMAs;
//   the node-adaptive weights - the one big stream, 131 / 65 KB per node and step - go straight from L2 / Infinity
//   Cache into each wave's registers in v_mfma_f32_16x16x4_f32 B-fragment order, eight k-groups (two chunks) ahead.
// The first MFMA issues after 24 KB have landed instead of 144 KB; what the epilogue adds (PX, R, previous state) is
// requested before the last chunk; 48 / 64 KB of LDS.
//
// MFMA 16x16x4 f32 operand maps: A lane l -> A[row = l&15][k = l>>4], B lane l -> B[k = l>>4][col = l&15],
// C/D lane l, reg e -> C[row = 4*(l>>4) + e][col = l&15].  A k-group is 16 reduction indices; MFMA step s of a
// group uses k = 16g + 4*(l>>4) + s, so both fragments of a group are one aligned float4 per lane.
//
// LDS tile layout: rows of 16-byte slots; slot q of row r is stored at position q ^ (r & 15) inside its
// 16-slot block, which makes the ds_read_b128 of lane (row, kq) conflict-free without padding.
//
// FRAGMENT-ORDERED intermediates: what one node kernel hands to the next for the SAME (node, row block) - the hoisted
// pre-activation PX and the reset-blend gate R - is stored exactly as the producer's accumulators hold it,
// [column tile][row tile][lane][4]: the producer's float4 IS the consumer's float4 (row 16 rt + 4 (l>>4) + e,
// column 16 ct + (l&15)), so both sides move whole 1 KB wave rows and nothing is transposed.
#ifndef MATGCN_NODE16_HIP
#define MATGCN_NODE16_HIP

// f32x4 / MFMA16 come from matgcn_kernels.hip (same translation unit)

#ifndef NODE_MIN_WAVES
#define NODE_MIN_WAVES 4   // __launch_bounds__ second argument: 4 waves per SIMD = a budget of 128 VGPRs (no spills).
                           // 5 (96 VGPRs: two node waves per SIMD would fit beside five 64-VGPR graph-mix waves of the
                           // other layer's chain) was measured in round 2: the forward got SLOWER (7.37 vs 7.17 ms) -
                           // the spills cost more than co-residency gives, the two chains phase-lock anyway
#endif
// (the weight ring of the K loop runs one 64-wide K chunk = 4 k-groups ahead of the MFMAs: 32-64 MFMAs per wave and
// chunk, shared by up to four waves per SIMD, cover the round trip)
#ifndef NODE_MIN_WAVES_32
#define NODE_MIN_WAVES_32 4   // the 32-row instantiations of k_gate16 / k_update16
#endif
#ifndef NODE_XT_LATE
#define NODE_XT_LATE 1              // k_update16: request the residual cell's x_t rows before the last K chunk
#endif
#ifndef NODE_SAVE_WT
#define NODE_SAVE_WT 1    // the activations forward_train saves leave through write-through (sc1) stores like the outputs
#endif
#ifndef PX16_RING
#define PX16_RING 4        // k_px16: two rings (two column tiles per wave)
#endif
#ifndef PX16_OUT_WT
#define PX16_OUT_WT 1      // k_px16 writes PX through (sc1)
#endif
#ifndef PX16_PIPELINE
#define PX16_PIPELINE 1    // k_px16 stages its A tile in K chunks under the MFMAs (0: whole tile first, the round-3 form)
#endif
#ifndef NODE_ROWS
#define NODE_ROWS 64       // rows (batch items) of one (node, row block) work item of k_gate16 / k_update16: 64 or 32.
                           // The fragment-ordered PX / R blocks stay 64-row blocks either way (a 32-row item is the
                           // lower or upper pair of row tiles of its block).
#endif
#define NODE_PX_BLOCK 12288   // floats of one fragment-ordered PX block: 12 column tiles x 4 row tiles x 64 lanes x 4
#define NODE_R_BLOCK 4096     // floats of one fragment-ordered R block: 4 column tiles x 4 row tiles x 64 lanes x 4

struct Node16Args {
  const float* s;        // [rows][Np][64]: h (gate / res-only) or z*h (update)
  const float* g;        // [N][rows][Ks][64] graph-mixed s
  long gNodeStride;      // floats between the nodes of g; 0 = rows*Ks*64 (the block may sit inside a larger one)
  const float* w;        // [N][nG][OT][64][4] recurrent rows of the node-adaptive weights (fragment order)
  const float* px;       // [N][RB][12][4][64][4] hoisted pre-activation of this step (x rows + bias), fragment order:
                         // column tiles 0..7 gate, 8..11 update - layers >= 1; null for layer 0, whose narrow x part
                         // is contracted in the kernel:
  const float* xa;       // [N][rows][16*nGx] folded x rows of this step [x | mix_k(x) | 1 | 0..] (layer 0) or null
  int nGx;               // k-groups of the x part (weights: groups nG .. nG+nGx-1 of the node's stream)
  int rows, N, Np, Ks;
  // gate
  float* zh;             // out [rows][Np][64]  z*h
  float* r;              // gate: out / update: in  [N][RB][4][4][64][4] reset-blend gate, fragment order
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
  // training, top layer: the dropout in front of the head (MultiATGCN.py:416) applied where the sequence is written -
  // seqDrop[b*seqRowStride + n*64 + o] = seq value * dropMask[b*dropRowStride + n*64 + o]   (null: no dropout here)
  const float* dropMask;
  long dropRowStride;
  float* seqDrop;
#ifdef NODE_LAB_STAMPS   // tools/labs/stamps_r04.py: per-wave phase stamps of this launch (NODE_STAMPS words per wave)
  unsigned int* stamps;
#endif
};

// (in-kernel phase stamps of the lab builds: NODE_STAMP* macros, matgcn_internal.h)

// Gate non-linearities of the step kernels.  The node kernels' epilogues are VALU-bound stretches in which the matrix
// pipe idles (round 4, tools/labs/stamps_r04.py: the 16 sigmoids per lane of k_gate16 took 8 200 cycles with two
// workgroups on a CU): the IEEE expf + division of `1 / (1 + expf(-x))` is 26 VALU instructions, tanhf 30.  These forms
// are 4 and 15: v_exp_f32 / v_rcp_f32 (1 ulp each); tanh keeps its RELATIVE accuracy near 0 with the odd series below
// |x| = 0.25 (next term 9e-9 relative there) and (1 - t) / (1 + t), t = exp(-2|x|), above it (<= 3 ulp).
// NODE_PRECISE_GATES=1 restores the libm forms (A/B builds).
#ifndef NODE_PRECISE_GATES
#define NODE_PRECISE_GATES 0
#endif
__device__ __forceinline__ float sigmoid16(float x) {
#if NODE_PRECISE_GATES
  return 1.0f / (1.0f + expf(-x));
#else
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
#endif
}
__device__ __forceinline__ float tanh16(float x) {
#if NODE_PRECISE_GATES
  return tanhf(x);
#else
  const float ax = fabsf(x), x2 = x * x;
  const float t = __builtin_amdgcn_exp2f(-2.88539008177792681472f * ax);
  const float big = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  const float small = ax * (1.0f + x2 * (-0.333333333f + x2 * (0.133333333f + x2 * (-0.0539682540f + x2 * 0.0218694885f))));
  return copysignf(ax < 0.25f ? small : big, x);
#endif
}

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

// a saved activation (forward_train): nothing reads it before the backward, milliseconds later.  Written through as well
// (NODE_SAVE_WT): a plain store leaves a dirty line in this XCD's L2 for the flush at the end of the kernel
__device__ __forceinline__ void save16(float* base, size_t off, const float4& v) {
#if NODE_SAVE_WT
  store_wt16(base, off, v);
#else
  *reinterpret_cast<float4*>(base + off) = v;
#endif
}

// position (in floats) of element (row, col) of a swizzled [.][16*blocks slots] tile with `spr` slots per row
__device__ __forceinline__ int swz(int row, int col, int spr) {
  const int slot = col >> 2;
  return (row * spr + ((slot & ~15) | ((slot ^ row) & 15))) * 4 + (col & 3);
}

// ---- operand precision of the K loop ----------------------------------------------------------------------------
// BF = false: fp32 operands on v_mfma_f32_16x16x4_f32 (the product path, bit-for-bit an fp32 fma chain).
// BF = true (matgcn_set_mix_precision(2), inference only, a side line with its own tolerance): the node-adaptive weights
// arrive as a bf16 copy of the same fragment stream - HALF the bytes of the one big stream of these kernels - and the A
// rows are rounded to bf16 on their way into LDS; one v_mfma_f32_16x16x16_bf16 per k-group, fp32 accumulation.  A lane's
// four bf16 of a k-group are exactly the four fp32 of its float4 in the fp32 stream (k = 16 g + 4 (l >> 4) + s), so both
// streams share one indexing.  State, PX, R, the residual cell and every epilogue stay fp32.
template <bool BF> struct NodeOp { typedef float4 T; };
template <> struct NodeOp<true> { typedef uint2 T; };
__device__ __forceinline__ uint2 to_bf16x4(const float4& v) {
  return make_uint2(bf16_rne(v.x) | (bf16_rne(v.y) << 16), bf16_rne(v.z) | (bf16_rne(v.w) << 16));
}
__device__ __forceinline__ bf16x4_t as_bf16x4(const uint2& v) {
  bf16x4_t r;
  r[0] = (short)(v.x & 0xffffu); r[1] = (short)(v.x >> 16); r[2] = (short)(v.y & 0xffffu); r[3] = (short)(v.y >> 16);
  return r;
}

// ---- K-chunk pipeline shared by k_gate16 and k_update16 --------------------------------------------------------
// staging coordinates of a thread: rows srow and srow + 32 of a chunk, 16-byte slot sq
// a ROWS-row chunk is ROWS x 16 slots = ROWS/32 float4 per thread of the 512: sweep it of a thread is row srow + 32 it
template <int ROWS>
struct ChunkStage {
  static constexpr int NS = ROWS / 32;
  int srow, sq;
  int g[NS];                 // the global rows (clamped into range: loads are never predicated, a predicated load
  bool v[NS];                //                  would make the compiler wait for each one separately)
  const float4* gsrc;        // this node's mixed rows [rows][Ks][16 slots]
  int spr;                   // slots per mixed row = 16 * Ks
};

template <int ROWS>
__device__ __forceinline__ ChunkStage<ROWS> chunk_stage(const Node16Args& a, int n, int rowBase) {
  ChunkStage<ROWS> c;
  c.srow = threadIdx.x >> 4; c.sq = threadIdx.x & 15;
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    c.v[it] = rowBase + c.srow + 32 * it < a.rows;
    c.g[it] = min(rowBase + c.srow + 32 * it, a.rows - 1);
  }
  c.spr = 16 * a.Ks;
  c.gsrc = reinterpret_cast<const float4*>(
      a.g + (size_t)n * (a.gNodeStride ? (size_t)a.gNodeStride : (size_t)a.rows * c.spr * 4));
  return c;
}

// request chunk c (1..Ks) = mixed slot c-1 of this node's rows
template <int ROWS>
__device__ __forceinline__ void chunk_load(const ChunkStage<ROWS>& c, int Ks, int chunk, float4 (&r)[ROWS / 32]) {
  const int k = max(min(chunk, Ks) - 1, 0);
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) r[it] = c.gsrc[(size_t)c.g[it] * c.spr + k * 16 + c.sq];
}

// registers -> one [ROWS][16 slots] swizzled LDS chunk; rows beyond a.rows are zero
template <int ROWS, bool BF = false>
__device__ __forceinline__ void chunk_store(float* buf, const ChunkStage<ROWS>& c, const float4 (&r)[ROWS / 32]) {
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {
    const int rr = c.srow + 32 * it;
    if constexpr (BF) *reinterpret_cast<uint2*>(&buf[(rr * 16 + (c.sq ^ (rr & 15))) * 2]) = to_bf16x4(keep4(c.v[it], r[it]));
    else *reinterpret_cast<float4*>(&buf[(rr * 16 + (c.sq ^ (rr & 15))) * 4]) = keep4(c.v[it], r[it]);
  }
}

// The four k-groups of one chunk for NRT row tiles starting at tile rt0.  The weights of a chunk wait in one HALF of
// the ring (wr[PAR]); the four k-groups of the NEXT chunk are requested into the other half BEFORE this chunk's first
// MFMA (a refill of the half being read could only issue after the MFMAs that read it: round 2 measured exactly that
// - the loads bunched up at the end of the chunk and the next chunk began by waiting for them).  A fragments come from
// the LDS chunk; accumulators rotate so that a chain is revisited every NRT-th instruction.
template <int NRT, int PAR, bool PREFETCH, bool BF = false>
__device__ __forceinline__ void chunk_mfma(const float* buf, int rt0, int j, int kq, typename NodeOp<BF>::T (&wr)[2][4],
                                           const typename NodeOp<BF>::T* wp, size_t gStride, int gNext, int gLast,
                                           f32x4 (&acc)[NRT]) {
  typedef typename NodeOp<BF>::T Op;
  if (PREFETCH) {
#ifdef NODE_LAB_NO_WEIGHTS   // tools/labs/nodelab2.hip: every k-group re-reads the node's first one (an L1 hit)
    gNext = 0; gLast = 0;
#endif
#pragma unroll
    for (int gl = 0; gl < 4; ++gl) wr[PAR ^ 1][gl] = wp[(size_t)min(gNext + gl, gLast) * gStride];
    // the scheduler would otherwise sink these loads below the chunk's MFMAs to shorten their live ranges
    __builtin_amdgcn_sched_barrier(0);
  }
  const Op* tile = reinterpret_cast<const Op*>(buf);
#pragma unroll
  for (int gl = 0; gl < 4; ++gl) {
    Op av[NRT];
#pragma unroll
    for (int q = 0; q < NRT; ++q) av[q] = tile[((rt0 + q) * 16 + j) * 16 + ((4 * gl + kq) ^ j)];
    const Op wv = wr[PAR][gl];
    if constexpr (BF) {
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16BF(as_bf16x4(av[q]), as_bf16x4(wv), acc[q]);
    } else {
#ifdef NODE_LAB_NO_MFMA   // tools/labs/nodelab2.hip: the same operand traffic without the matrix pipe
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q][0] += av[q].x * wv.x + av[q].y * wv.y + av[q].z * wv.z + av[q].w * wv.w;
#else
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16(av[q].x, wv.x, acc[q]);
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16(av[q].y, wv.y, acc[q]);
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16(av[q].z, wv.z, acc[q]);
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16(av[q].w, wv.w, acc[q]);
#endif
    }
  }
}

// The whole recurrent K range of one node: acc[q] = [s | G][rows of tiles rt0..rt0+NRT-1] . W_n[:, this wave's tile]
// Hs: chunk 0 (left in place), Gb: the two ping-pong buffers.  Every thread of the workgroup must call it (barriers);
// on return every wave may still be inside the last chunk's MFMAs.
// What is requested before the first MFMA decides how long the matrix pipe idles at the start of the launch (all
// workgroups of a launch start together): only the 16 KB of state rows and the first k-groups of weights come first;
// the mixed chunks follow, and everything the epilogue adds (PX, R, the previous state, the narrow x part of layer 0)
// is requested by `late()` just before the LAST chunk's MFMAs (one call site: the last chunk is peeled).
//   schedule   request s rows, the weights of chunk 0, chunks 1 and 2;  Hs <- s;  barrier;
//              chunk 0 from Hs (weights of chunk 1 requested first);  Gb[0] <- chunk 1, Gb[1] <- chunk 2, request
//              chunks 3 and 4;  barrier;
//              for c = 1..Ks-1:  request the weights of chunk c+1, chunk c from Gb[(c-1)&1];  barrier;
//                                Gb[(c-1)&1] <- chunk c+2, request chunk c+4
//              late();  chunk Ks
template <int ROWS, int NRT, bool BF, typename Late>
__device__ __forceinline__ void node_k_loop(const Node16Args& a, int n, int rowBase, float* Hs, float* Gb, int rt0, int j,
                                            int kq, const typename NodeOp<BF>::T* wp, size_t gStride, f32x4 (&acc)[NRT],
                                            Late&& late NODE_STAMP_PARAM) {
  constexpr int NS = ROWS / 32, CH = ROWS * 64;      // float4 per thread and chunk; floats of one LDS chunk
  const int Ks = a.Ks, gLast = 4 * (1 + Ks) - 1;
  const ChunkStage<ROWS> cs = chunk_stage<ROWS>(a, n, rowBase);
  float4 hS[NS];
#pragma unroll
  for (int it = 0; it < NS; ++it)
    hS[it] = *reinterpret_cast<const float4*>(a.s + ((size_t)cs.g[it] * a.Np + n) * 64 + cs.sq * 4);
  typename NodeOp<BF>::T wr[2][4];       // weight ring: chunk c reads half c & 1
#pragma unroll
  for (int r = 0; r < 4; ++r) wr[0][r] = wp[(size_t)min(r, gLast) * gStride];
  // staging registers: chunk c waits in st[c & 1].  Every request below is UNCONDITIONAL (the chunk index is clamped, a
  // request past the last chunk re-reads it from L2): a load behind a branch makes the number of loads in flight depend
  // on the path, and the compiler then drains the whole queue (vmcnt(0)) where the paths meet - at the loop head
  float4 st[2][NS];
  chunk_load<ROWS>(cs, Ks, 1, st[1]);
  chunk_load<ROWS>(cs, Ks, 2, st[0]);
  // (round 4 lab: requesting these two chunks only after the state rows are in LDS - the waves sit 6-8 k cycles in the
  //  vector-memory issue queue before this point - changed nothing: 6.75 vs 6.74 ms, profiles/r04_node_epilogue_lab.log)
#pragma unroll
  for (int q = 0; q < NRT; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  NODE_STAMP(1);   // requests issued
  chunk_store<ROWS, BF>(Hs, cs, hS);
  NODE_STAMP(2);   // state rows arrived and stored
  __syncthreads();
  NODE_STAMP(3);
  if (Ks > 0) {
    chunk_mfma<NRT, 0, true, BF>(Hs, rt0, j, kq, wr, wp, gStride, 4, gLast, acc);
    NODE_STAMP(4);   // chunk 0 issued
    chunk_store<ROWS, BF>(Gb, cs, st[1]);
    if (Ks > 1) chunk_store<ROWS, BF>(Gb + CH, cs, st[0]);
    chunk_load<ROWS>(cs, Ks, 3, st[1]);
    chunk_load<ROWS>(cs, Ks, 4, st[0]);
    NODE_STAMP(5);   // chunks 1, 2 arrived and stored
    __syncthreads();
    NODE_STAMP(6);
    for (int c = 1; c < Ks; c += 2) {
      // odd chunk c (not the last) in Gb[0]; afterwards Gb[0] <- chunk c+2 (waiting in st[1])
      chunk_mfma<NRT, 1, true, BF>(Gb, rt0, j, kq, wr, wp, gStride, 4 * (c + 1), gLast, acc);
      NODE_STAMP(7);   // (the last odd chunk's)
      __syncthreads();
      NODE_STAMP(8);
      if (c + 2 <= Ks) chunk_store<ROWS, BF>(Gb, cs, st[1]);
      chunk_load<ROWS>(cs, Ks, c + 4, st[1]);
      if (c + 1 < Ks) {  // even chunk c+1 (not the last) in Gb[1]; afterwards Gb[1] <- chunk c+3 (waiting in st[0])
        chunk_mfma<NRT, 0, true, BF>(Gb + CH, rt0, j, kq, wr, wp, gStride, 4 * (c + 2), gLast, acc);
        NODE_STAMP(9);
        __syncthreads();
        NODE_STAMP(10);
        if (c + 3 <= Ks) chunk_store<ROWS, BF>(Gb + CH, cs, st[0]);
        chunk_load<ROWS>(cs, Ks, c + 5, st[0]);
      }
    }
  }
  late();
  NODE_STAMP(11);  // late requests issued
  if (Ks & 1) chunk_mfma<NRT, 1, false, BF>(Gb, rt0, j, kq, wr, wp, gStride, 0, gLast, acc);               // odd last chunk
  else chunk_mfma<NRT, 0, false, BF>(Ks > 0 ? Gb + CH : Hs, rt0, j, kq, wr, wp, gStride, 0, gLast, acc);  // even (or chunk 0)
  NODE_STAMP(12);  // last chunk issued
}

// layer-0 x part: acc[q] += XA[rows of tile rt0+q][16 gx .. +16] . Wx[gx]; A fragments come straight from global
// memory (a row of XA is 64*nGx bytes, a 16-row tile is contiguous), weights from the tail of the node's stream
template <int NRT, bool BF = false>
__device__ __forceinline__ void x_groups(const Node16Args& a, int n, int rowBase, int rt0, const typename NodeOp<BF>::T* wx,
                                         size_t gStride, int i, int kq, f32x4 (&acc)[NRT]) {
  const int kx = 16 * a.nGx;
  const float* base = a.xa + (size_t)n * a.rows * kx + kq * 4;
  for (int gx = 0; gx < a.nGx; ++gx) {
    const typename NodeOp<BF>::T wv = wx[(size_t)gx * gStride];
    float4 av[NRT];
#pragma unroll
    for (int q = 0; q < NRT; ++q)
      av[q] = *reinterpret_cast<const float4*>(base + (size_t)min(rowBase + (rt0 + q) * 16 + i, a.rows - 1) * kx + gx * 16);
    if constexpr (BF) {
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16BF(as_bf16x4(to_bf16x4(av[q])), as_bf16x4(wv), acc[q]);
    } else {
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16(av[q].x, wv.x, acc[q]);
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16(av[q].y, wv.y, acc[q]);
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16(av[q].z, wv.z, acc[q]);
#pragma unroll
      for (int q = 0; q < NRT; ++q) acc[q] = MFMA16(av[q].w, wv.w, acc[q]);
    }
  }
}

__device__ __forceinline__ f32x4 as_f32x4(const float4& v) { return f32x4{v.x, v.y, v.z, v.w}; }

// work item id -> (node, row block): the row blocks of one node get ids 8 apart - the same XCD under round-robin
// dispatch and adjacent in time, so every block after the first finds the node's weights in that XCD's L2
__device__ __forceinline__ bool node_item(int id, int blocks, int N, int& n, int& rbr) {
  const int grp = id / (8 * blocks), rem = id - grp * 8 * blocks;
  n = grp * 8 + (rem & 7); rbr = rem >> 3;
  return n < N;
}
inline unsigned node_items(int N, int rows, int blockRows) {
  return (unsigned)((N + 7) / 8 * 8 * ((rows + blockRows - 1) / blockRows));
}

// ---- gate AGCN + sigmoid + z*h (MultiATGCN.py:122-125) -----------------------------------------------------
// wave w = column tile w of 8 (0..3: z, 4..7: r), all ROWS/16 row tiles.  LDS 3 chunks of ROWS x 64 floats: Hs | Gb[2]
#ifndef NODE_MIN_WAVES_GATE
#define NODE_MIN_WAVES_GATE NODE_MIN_WAVES
#endif
template <bool SAVE, int ROWS, bool BF = false>
__global__ __launch_bounds__(512, ROWS == 64 ? NODE_MIN_WAVES_GATE : NODE_MIN_WAVES_32) void k_gate16(Node16Args a) {
  typedef typename NodeOp<BF>::T Op;
  static_assert(!(SAVE && BF), "the training forward runs fp32 operands");
  constexpr int NRT = ROWS / 16, CH = ROWS * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds;               // [ROWS][16 slots] the state rows: chunk 0, and the h of z*h
  float* Gb = lds + CH;          // 2 x [ROWS][16 slots] mixed-slot chunks; afterwards the z*h output tile
  int n, rbr;
  if (!node_item(blockIdx.x, (a.rows + ROWS - 1) / ROWS, a.N, n, rbr)) return;
  const int rowBase = rbr * ROWS;
  const int RB = (a.rows + 63) >> 6, rb = rowBase >> 6, rtb = (rowBase & 63) >> 4;   // 64-row block of PX / R, first row tile in it
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
  const int nG = 4 * (1 + a.Ks);
  const size_t gStride = 8 * 64;
  const Op* wp = reinterpret_cast<const Op*>(a.w) + ((size_t)n * (nG + a.nGx) * 8 + w) * 64 + lane;
  f32x4 acc[NRT];
  float4 pxv[NRT];               // hoisted pre-activation (x rows + bias) in fragment order, added in the epilogue
  float hz[BF ? NRT : 1][4];     // BF: the fp32 state values of z*h (the LDS copy is rounded to bf16), requested late
  NODE_STAMP_DECL
  NODE_STAMP(0);
  node_k_loop<ROWS, NRT, BF>(a, n, rowBase, Hs, Gb, 0, j, kq, wp, gStride, acc, [&]() {
    if constexpr (BF) {
      if (w < 4) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            hz[rt][e] = a.s[((size_t)min(rowBase + rt * 16 + 4 * kq + e, a.rows - 1) * a.Np + n) * 64 + 16 * w + j];
      }
    }
    if (a.px) {
      const float4* pf = reinterpret_cast<const float4*>(a.px) + ((((size_t)n * RB + rb) * 12 + w) * 4 + rtb) * 64 + lane;
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) pxv[rt] = pf[rt * 64];
    } else {   // layer 0 contracts its narrow x part here, straight from global memory
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) pxv[rt] = make_float4(0.f, 0.f, 0.f, 0.f);
      x_groups<NRT, BF>(a, n, rowBase, 0, wp + (size_t)nG * gStride, gStride, j, kq, acc);
    }
  } NODE_STAMP_ARG);
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    acc[rt][0] += pxv[rt].x; acc[rt][1] += pxv[rt].y; acc[rt][2] += pxv[rt].z; acc[rt][3] += pxv[rt].w;
  }
  NODE_STAMP(13);  // accumulators + PX ready
  // epilogue: zr = sigmoid(.);  r leaves in fragment order straight from the accumulators (the update kernel of
  // this node reads it back the same way); z is gathered as a [ROWS][64] tile in LDS (the chunk buffers are dead once
  // every wave has left the K loop), and z*h leaves as whole 256-byte rows of the next mix's operand: the row sweep
  // multiplies the z tile with the state tile, float4 by float4.
  // (Round 4: this stretch used to read h and write z*h element by element behind exec-masked branches on the wave
  // index - one exposed LDS round trip per element; the wave index is now provably uniform, the 16 sigmoids of a lane
  // are independent, and the LDS sees 16 scalar writes and 2 wide reads per thread.  The training instantiation keeps
  // an r tile too and saves z and r as float4 rows instead of 64-byte pieces.)
  __syncthreads();
  NODE_STAMP(14);
  float* Zt = Gb;                // z tile (BF: z*h, the LDS state copy is bf16 there)
  float* Rt = Gb + CH;           // SAVE: r tile
  const int o = 16 * w + j;
  if (a.raw) {                   // unit entry point: pre-activation dump
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int b = rowBase + rt * 16 + 4 * kq + e;
        if (b < a.rows) a.raw[((size_t)b * a.N + n) * 128 + o] = acc[rt][e];
      }
  }
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[rt][e] = sigmoid16(acc[rt][e]);
  if (w < 4) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lb = rt * 16 + 4 * kq + e;
        if constexpr (BF) Zt[swz(lb, o, 16)] = acc[rt][e] * hz[rt][e];
        else Zt[swz(lb, o, 16)] = acc[rt][e];
      }
  } else {
    const size_t base = (((((size_t)n * RB + rb) * 4 + (w - 4)) * 4) + rtb) * 256 + (size_t)lane * 4;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
      store_wt16(a.r, base + (size_t)rt * 256, make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]));
    if constexpr (SAVE) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) Rt[swz(rt * 16 + 4 * kq + e, o - 64, 16)] = acc[rt][e];
    }
  }
  NODE_STAMP(15);  // sigmoids done, tiles in LDS
  __syncthreads();
  NODE_STAMP(16);
#pragma unroll
  for (int it = 0; it < ROWS / 32; ++it) {          // ROWS rows x 16 slots float4 over 512 threads
    const int lb = (tid >> 4) + 32 * it, q = tid & 15, b = rowBase + lb;
    if (b >= a.rows) continue;
    const int at = (lb * 16 + (q ^ (lb & 15))) * 4;
    const float4 z = *reinterpret_cast<const float4*>(&Zt[at]);
    float4 v = z;
    if constexpr (!BF) {
      const float4 h = *reinterpret_cast<const float4*>(&Hs[at]);
      v = make_float4(z.x * h.x, z.y * h.y, z.z * h.z, z.w * h.w);
    }
    store_wt16(a.zh, ((size_t)b * a.Np + n) * 64 + q * 4, v);
    if constexpr (SAVE) {
      save16(a.svZ, ((size_t)b * a.Np + n) * 64 + q * 4, z);
      save16(a.svR, ((size_t)b * a.Np + n) * 64 + q * 4, *reinterpret_cast<const float4*>(&Rt[at]));
    }
  }
  NODE_STAMP(17);  // stores issued
#ifdef NODE_LAB_STAMPS
  __builtin_amdgcn_s_waitcnt(0);   // every store of this wave acknowledged
  NODE_STAMP(18);
#endif
  NODE_STAMP_FLUSH(a);
}

// ---- hoisted x part of layers >= 1: PX[t][n][rb] = bias[n] + [x | mix_k(x)] . Wx[n], fragment order ---------------
// (MultiATGCN.py:106-108 restricted to the x rows; gate column tiles 0..7, update column tiles 8..11.)  64-row tile
// [x | G] of one (node, step, row block) in LDS, weights streamed once per workgroup, 12 column tiles over 8 waves:
// waves 0-3 take two tiles, waves 4-7 one, i.e. three per SIMD.  Workgroup ids of one node's blocks are 8 apart
// (same XCD, same time: the second and later blocks read the node's weights from that XCD's L2).
struct Px16Args {
  const float* x;        // [steps*B][Np][64] input rows of the chunk (layer below, time-major)
  const float* g;        // [N][steps*B][Ks][64] graph-mixed input rows
  const float* w;        // [N][nG][12][64][4] x rows of both AGCNs (fragment order)
  const float* bias;     // [N][192]
  float* pxOut;          // [steps][N][RB][12][4][64][4] slice of PX
  int steps, N, Np, Ks, B;
};

// stage the whole 64-row A tile of a (node, step, row block): Hs <- x rows (16 slots), Gs <- G rows (16*Ks slots)
__device__ __forceinline__ void stage_node_tile(const Node16Args& a, int n, int rowBase, float* Hs, float* Gs) {
  const ChunkStage<64> cs = chunk_stage<64>(a, n, rowBase);
  float4 hS[2];
  hS[0] = *reinterpret_cast<const float4*>(a.s + ((size_t)cs.g[0] * a.Np + n) * 64 + cs.sq * 4);
  hS[1] = *reinterpret_cast<const float4*>(a.s + ((size_t)cs.g[1] * a.Np + n) * 64 + cs.sq * 4);
  const int rA = cs.srow, rB = cs.srow + 32;
  const int pA = cs.sq ^ (rA & 15), pB = cs.sq ^ (rB & 15);
  const int spr = cs.spr;
  if (a.Ks == 0) chunk_store<64>(Hs, cs, hS);
  for (int k0 = 0; k0 < a.Ks; k0 += 4) {
    float4 vAk[4], vBk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = min(k0 + kk, a.Ks - 1);
      vAk[kk] = cs.gsrc[(size_t)cs.g[0] * spr + k * 16 + cs.sq];
      vBk[kk] = cs.gsrc[(size_t)cs.g[1] * spr + k * 16 + cs.sq];
    }
    if (k0 == 0) chunk_store<64>(Hs, cs, hS);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (k0 + kk < a.Ks) {
        *reinterpret_cast<float4*>(&Gs[(rA * spr + (k0 + kk) * 16 + pA) * 4]) = keep4(cs.v[0], vAk[kk]);
        *reinterpret_cast<float4*>(&Gs[(rB * spr + (k0 + kk) * 16 + pB) * 4]) = keep4(cs.v[1], vBk[kk]);
      }
    }
  }
}

// A fragment of row tile rt for k-group g (g < 4: the x slots, else the mixed slots) of the whole-tile layout
__device__ __forceinline__ float4 a_frag(const float* Hs, const float* Gs, int Ks, int rt, int g, int i, int kq) {
  const int row = rt * 16 + i;
  if (g < 4) return *reinterpret_cast<const float4*>(&Hs[(row * 16 + ((4 * g + kq) ^ i)) * 4]);
  const int q = 4 * (g - 4) + kq;
  return *reinterpret_cast<const float4*>(&Gs[(row * 16 * Ks + ((q & ~15) | ((q ^ i) & 15))) * 4]);
}

// NRT: 16-row tiles of the 64-row block that hold batch rows (4; 2 / 1 for batches of at most 32 / 16 rows - the reference
// ships batch_size 16, where three of the four row tiles were padding: 64 us per launch at ANY batch size until round 4)
template <int NRT, bool BF = false>
__global__ __launch_bounds__(512, 4) void k_px16(Px16Args p) {
  typedef typename NodeOp<BF>::T Op;   // BF (precision mode 2): the bf16 copy of the weight stream, A rows rounded into LDS
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds;
  float* Gs = lds + 64 * 64;
  const int RB = (p.B + 63) >> 6, blocks = p.steps * RB;
  const int id = blockIdx.x;
  const int grp = id / (8 * blocks), rem = id - grp * 8 * blocks;
  const int n = grp * 8 + (rem & 7), blk = rem >> 3;
  if (n >= p.N) return;
  const int tl = blk / RB, rb = blk - tl * RB, rowBase = rb * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
  const int nG = 4 * (1 + p.Ks);
  const bool two = w < 4;                       // this wave also owns column tile w + 8
  const Op* wp0 = reinterpret_cast<const Op*>(p.w) + ((size_t)n * nG * 12 + w) * 64 + lane;
  const Op* wp1 = reinterpret_cast<const Op*>(p.w) + ((size_t)n * nG * 12 + min(w + 8, 11)) * 64 + lane;
  Node16Args a;                                 // the staging helper speaks Node16Args: the B rows of step tl
  a.s = p.x + (size_t)tl * p.B * p.Np * 64;
  a.g = p.g + (size_t)tl * p.B * p.Ks * 64; a.gNodeStride = (long)p.steps * p.B * p.Ks * 64;
  a.rows = p.B; a.Np = p.Np; a.Ks = p.Ks;
  // the first weight groups and the bias depend on nothing this kernel stages: requested BEFORE the A tile, they are in
  // flight while the tile's 64 KB go global -> registers -> LDS (round 4: no measurable gain, 69.6 us either way; a ring of
  // 6 groups instead of 4 is slower, 72 us: profiles/r04_small_batch_lab.log)
  Op wr0[PX16_RING], wr1[PX16_RING];
#pragma unroll
  for (int r = 0; r < PX16_RING; ++r) {
    wr0[r] = wp0[(size_t)min(r, nG - 1) * 12 * 64];
    wr1[r] = wp1[(size_t)min(r, nG - 1) * 12 * 64];
  }
  const int o0 = 16 * w + j, o1 = 16 * (w + 8) + j;
  const float b0 = p.bias[(size_t)n * 192 + o0], b1 = two ? p.bias[(size_t)n * 192 + o1] : 0.f;
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc0[NRT], acc1[NRT];
#if PX16_PIPELINE
  // the A tile goes through LDS in 16 KB K chunks (chunk 0 = the x rows, chunk c = mixed slot c-1) on the schedule of
  // node_k_loop: chunk c+2 is written to LDS and chunk c+4 requested while chunk c feeds the matrix pipe - the tile's
  // 64 KB no longer arrive before the first MFMA.  Both weight rings keep their one-chunk lead (refilled as consumed).
  const int Ks = p.Ks;
  float* Gb = Gs;
  constexpr int CH = 64 * 64;
  const ChunkStage<64> cs = chunk_stage<64>(a, n, rowBase);
  float4 hS[2], st[2][2];
#pragma unroll
  for (int it = 0; it < 2; ++it)
    hS[it] = *reinterpret_cast<const float4*>(a.s + ((size_t)cs.g[it] * a.Np + n) * 64 + cs.sq * 4);
  chunk_load<64>(cs, Ks, 1, st[1]);
  chunk_load<64>(cs, Ks, 2, st[0]);
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) { acc0[rt] = f32x4{b0, b0, b0, b0}; acc1[rt] = f32x4{b1, b1, b1, b1}; }
  auto chunk = [&](const float* buf, int c) {
    const Op* tile = reinterpret_cast<const Op*>(buf);
#pragma unroll
    for (int gl = 0; gl < 4; ++gl) {
      const int g = 4 * c + gl;
      const Op wv0 = wr0[gl], wv1 = wr1[gl];
      wr0[gl] = wp0[(size_t)min(g + 4, nG - 1) * 12 * 64];
      wr1[gl] = wp1[(size_t)min(g + 4, nG - 1) * 12 * 64];
      Op av[NRT];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) av[rt] = tile[(rt * 16 + j) * 16 + ((4 * gl + kq) ^ j)];
      if constexpr (BF) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc0[rt] = MFMA16BF(as_bf16x4(av[rt]), as_bf16x4(wv0), acc0[rt]);
        if (two) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16BF(as_bf16x4(av[rt]), as_bf16x4(wv1), acc1[rt]);
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc0[rt] = MFMA16(av[rt].x, wv0.x, acc0[rt]);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc0[rt] = MFMA16(av[rt].y, wv0.y, acc0[rt]);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc0[rt] = MFMA16(av[rt].z, wv0.z, acc0[rt]);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc0[rt] = MFMA16(av[rt].w, wv0.w, acc0[rt]);
        if (two) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].x, wv1.x, acc1[rt]);
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].y, wv1.y, acc1[rt]);
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].z, wv1.z, acc1[rt]);
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].w, wv1.w, acc1[rt]);
        }
      }
    }
  };
  chunk_store<64, BF>(Hs, cs, hS);
  __syncthreads();
  if (Ks > 0) {
    chunk(Hs, 0);
    chunk_store<64, BF>(Gb, cs, st[1]);
    if (Ks > 1) chunk_store<64, BF>(Gb + CH, cs, st[0]);
    chunk_load<64>(cs, Ks, 3, st[1]);
    chunk_load<64>(cs, Ks, 4, st[0]);
    __syncthreads();
    for (int c = 1; c < Ks; c += 2) {
      chunk(Gb, c);
      __syncthreads();
      if (c + 2 <= Ks) chunk_store<64, BF>(Gb, cs, st[1]);
      chunk_load<64>(cs, Ks, c + 4, st[1]);
      if (c + 1 < Ks) {
        chunk(Gb + CH, c + 1);
        __syncthreads();
        if (c + 3 <= Ks) chunk_store<64, BF>(Gb + CH, cs, st[0]);
        chunk_load<64>(cs, Ks, c + 5, st[0]);
      }
    }
  }
  if (Ks & 1) chunk(Gb, Ks);
  else chunk(Ks > 0 ? Gb + CH : Hs, Ks);
#else
  static_assert(!BF, "the bf16 form exists for the chunked A tile only");
  stage_node_tile(a, n, rowBase, Hs, Gs);
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) { acc0[rt] = f32x4{b0, b0, b0, b0}; acc1[rt] = f32x4{b1, b1, b1, b1}; }
  __syncthreads();
  for (int g0 = 0; g0 < nG; g0 += PX16_RING) {
#pragma unroll
    for (int r = 0; r < PX16_RING; ++r) {
      const int g = g0 + r;
      const float4 wv0 = wr0[r], wv1 = wr1[r];
      wr0[r] = wp0[(size_t)min(g + PX16_RING, nG - 1) * 12 * 64];
      wr1[r] = wp1[(size_t)min(g + PX16_RING, nG - 1) * 12 * 64];
      if (g < nG) {
        float4 av[NRT];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) av[rt] = a_frag(Hs, Gs, p.Ks, rt, g, j, kq);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc0[rt] = MFMA16(av[rt].x, wv0.x, acc0[rt]);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc0[rt] = MFMA16(av[rt].y, wv0.y, acc0[rt]);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc0[rt] = MFMA16(av[rt].z, wv0.z, acc0[rt]);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc0[rt] = MFMA16(av[rt].w, wv0.w, acc0[rt]);
        if (two) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].x, wv1.x, acc1[rt]);
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].y, wv1.y, acc1[rt]);
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].z, wv1.z, acc1[rt]);
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].w, wv1.w, acc1[rt]);
        }
      }
    }
  }
#endif
  // the accumulators leave as they are: one 1 KB wave row per (column tile, row tile)
#if PX16_OUT_WT
  // written through (sc1) like every other producer -> consumer buffer of the step kernels: the 39 MB a chunk's launch
  // produces leave the L2s while it runs, not as dirty lines at its end (this (node, step, row block)'s 48 KB block is the
  // wave-uniform base, so the lane offsets stay small at any N)
  float* pxb = p.pxOut + (((size_t)tl * p.N + n) * RB + rb) * NODE_PX_BLOCK;
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    store_wt16(pxb, (((size_t)w * 4 + rt) * 64 + lane) * 4, make_float4(acc0[rt][0], acc0[rt][1], acc0[rt][2], acc0[rt][3]));
    if (two)
      store_wt16(pxb, (((size_t)(w + 8) * 4 + rt) * 64 + lane) * 4, make_float4(acc1[rt][0], acc1[rt][1], acc1[rt][2], acc1[rt][3]));
  }
#else
  float4* dst = reinterpret_cast<float4*>(p.pxOut) + (((size_t)tl * p.N + n) * RB + rb) * (NODE_PX_BLOCK / 4) + lane;
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    dst[((size_t)w * 4 + rt) * 64] = make_float4(acc0[rt][0], acc0[rt][1], acc0[rt][2], acc0[rt][3]);
    if (two) dst[((size_t)(w + 8) * 4 + rt) * 64] = make_float4(acc1[rt][0], acc1[rt][1], acc1[rt][2], acc1[rt][3]);
  }
#endif
}

// ---- update AGCN + tanh + GRU blend, fused with the residual GRU cell and the per-step blend ------------------
// MODE 0: ATGRU update only (h' out); 1: update + residual cell (+ blend); 2: residual cell only on s (unit entry)
//
// Waves: (ct = w&3, rh = w>>2) = column tile ct of 4, row tiles of half rh (ROWS/32 of them), the whole K range (the
// two waves of a column tile request the same weight fragments: the second request is an L1 / L2 hit, HBM sees each
// byte once).  The residual cell then runs two small GEMMs on tiles that never leave LDS.
// LDS 4 chunks of ROWS x 64 floats: Hs | Gb[2] | X
template <int MODE, bool SAVE, int ROWS, bool BF = false>
__global__ __launch_bounds__(512, ROWS == 64 ? NODE_MIN_WAVES : NODE_MIN_WAVES_32) void k_update16(Node16Args a) {
  typedef typename NodeOp<BF>::T Op;
  constexpr int NRT = ROWS / 16, NR2 = ROWS / 32, NS = ROWS / 32, CH = ROWS * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Hs = lds;               // [ROWS][16 slots]: z*h (chunk 0) during the update GEMM, then h'
  float* Gb = lds + CH;          // 2 x [ROWS][16 slots] mixed-slot chunks; reused by the residual cell:
  float* ZH2 = Gb;               //   [ROWS][16 slots] z2*h'
  float* R2 = Gb + CH;           //   [ROWS][16 slots] r2
  float* XT = lds + 3 * CH;      // [ROWS][16 slots] x_t (zero padded); afterwards the output tile
  float* SV = lds + 4 * CH;      // SAVE only (80 KB of LDS): the tile an activation passes through on its way to the
                                 // training buffer - hc, then z2 - so that it is saved as float4 rows (round 4; scalar
                                 // stores from the accumulator layout were 64-byte pieces, +1.1 ms per training forward)
  static_assert(!(SAVE && BF), "the training forward runs fp32 operands");
  int n, rbr;
  if (!node_item(blockIdx.x, (a.rows + ROWS - 1) / ROWS, a.N, n, rbr)) return;
  const int rowBase = rbr * ROWS;
  const int RB = (a.rows + 63) >> 6, rb = rowBase >> 6, rtb = (rowBase & 63) >> 4;   // 64-row block of PX / R, first row tile in it
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
  const int ct = w & 3, rh = w >> 2;
  const int srow = tid >> 4, sq = tid & 15;   // staging coordinates: 32 rows x 16 slots per sweep
  const int o4 = 16 * ct + j;                 // column of this lane in a 64-wide tile
  NODE_STAMP_DECL
  // the residual cell's x_t rows of this thread's staging slots (zero padded to Cpad); MODE 1 requests them before the
  // last K chunk - requested after the update's epilogue, their whole memory latency sat between two barriers with the
  // matrix pipe idle (round 4 stamps: 10 k cycles for that stretch)
  float4 xv[NS];
  auto request_xt = [&]() {
#pragma unroll
    for (int it = 0; it < NS; ++it) {
      const int rr = srow + 32 * it;
      const size_t xrow = (size_t)min(rowBase + rr, a.rows - 1) * a.xRowStride;
      if (a.C == 64) {
        xv[it] = *reinterpret_cast<const float4*>(a.xt + xrow + (size_t)n * 64 + sq * 4);
      } else {
        float e4[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          const int c = min(sq * 4 + cc, a.C - 1);
          const float v = a.xt[xrow + (size_t)n * a.C + c];
          e4[cc] = (sq * 4 + cc < a.C) ? v : 0.f;
        }
        xv[it] = make_float4(e4[0], e4[1], e4[2], e4[3]);
      }
    }
  };

  if (MODE != 2) {
    const int nG = 4 * (1 + a.Ks);
    const size_t gStride = 4 * 64;
    const Op* wp = reinterpret_cast<const Op*>(a.w) + ((size_t)n * (nG + a.nGx) * 4 + ct) * 64 + lane;
    f32x4 acc[NR2];
    // epilogue operands, requested just before the last chunk: PX and r in fragment order (r as k_gate16 left it),
    // the previous state row-major
    float4 pxv[NR2], rv[NR2];
    float hv[NR2][4];
    NODE_STAMP(0);
    node_k_loop<ROWS, NR2, BF>(a, n, rowBase, Hs, Gb, NR2 * rh, j, kq, wp, gStride, acc, [&]() {
      if (a.px) {
        const float4* pf = reinterpret_cast<const float4*>(a.px) +
                           ((((size_t)n * RB + rb) * 12 + 8 + ct) * 4 + rtb + NR2 * rh) * 64 + lane;
#pragma unroll
        for (int q = 0; q < NR2; ++q) pxv[q] = pf[q * 64];
      } else {
#pragma unroll
        for (int q = 0; q < NR2; ++q) pxv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        x_groups<NR2, BF>(a, n, rowBase, NR2 * rh, wp + (size_t)nG * gStride, gStride, j, kq, acc);
      }
      const float4* rf = reinterpret_cast<const float4*>(a.r) + ((((size_t)n * RB + rb) * 4 + ct) * 4 + rtb + NR2 * rh) * 64 + lane;
#pragma unroll
      for (int q = 0; q < NR2; ++q) rv[q] = rf[q * 64];
#pragma unroll
      for (int q = 0; q < NR2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int b = min(rowBase + (NR2 * rh + q) * 16 + 4 * kq + e, a.rows - 1);
          hv[q][e] = a.h[((size_t)b * a.Np + n) * 64 + o4];
        }
#if NODE_XT_LATE
      if (MODE == 1) request_xt();
#endif
    } NODE_STAMP_ARG);
#pragma unroll
    for (int q = 0; q < NR2; ++q) {
      acc[q][0] += pxv[q].x; acc[q][1] += pxv[q].y; acc[q][2] += pxv[q].z; acc[q][3] += pxv[q].w;
    }
    NODE_STAMP(13);
    __syncthreads();   // every wave is out of the K loop: Hs (z*h) may be overwritten by h'
    NODE_STAMP(14);
#pragma unroll
    for (int q = 0; q < NR2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lb = (NR2 * rh + q) * 16 + 4 * kq + e, b = rowBase + lb;
        const float hc = tanh16(acc[q][e]);
        const float rr = e == 0 ? rv[q].x : e == 1 ? rv[q].y : e == 2 ? rv[q].z : rv[q].w;
        float hn = rr * hv[q][e] + (1.0f - rr) * hc;   // (MultiATGCN.py:127: r blends, z gated the candidate)
        if constexpr (SAVE) SV[swz(lb, o4, 16)] = hc;   // saved as float4 rows behind the next barrier
        if (b >= a.rows) hn = 0.f;
        if (MODE == 0) { if (b < a.rows) a.hout[((size_t)b * a.Np + n) * 64 + o4] = hn; }
        else Hs[swz(lb, o4, 16)] = hn;                  // h' tile for the residual cell (z*h no longer needed)
      }
    if (MODE == 0) return;
  } else {
    // residual cell only: h' := s rows
    float4 sv[NS];
#pragma unroll
    for (int it = 0; it < NS; ++it)
      sv[it] = *reinterpret_cast<const float4*>(a.s + ((size_t)min(rowBase + srow + 32 * it, a.rows - 1) * a.Np + n) * 64 + sq * 4);
#pragma unroll
    for (int it = 0; it < NS; ++it) {
      const int rr = srow + 32 * it;
      *reinterpret_cast<float4*>(&Hs[(rr * 16 + (sq ^ (rr & 15))) * 4]) = keep4(rowBase + rr < a.rows, sv[it]);
    }
  }

  // ---- residual GRU cell on [x_t | h'] (MultiATGCN.py:142-150) ----
  const int ngx = a.Cpad >> 4;                     // x groups of the residual GEMMs (1 or 4)
  const int nG1 = ngx + 4;                         // <= 8
  if (MODE != 1 || !NODE_XT_LATE) request_xt();
#pragma unroll
  for (int it = 0; it < NS; ++it) {   // x_t tile (zero padded to Cpad)
    const int rr = srow + 32 * it;
    if (a.C == 64 || sq < (a.Cpad >> 2))
      *reinterpret_cast<float4*>(&XT[(rr * 16 + (sq ^ (rr & 15))) * 4]) = keep4(rowBase + rr < a.rows, xv[it]);
  }
  // weights of both residual GEMMs (shared by all nodes, L2-resident), requested before the tile barrier
  const int rp = rh;
  float4 rgv[8];
  {
    const float4* rgp = reinterpret_cast<const float4*>(a.rg) + (size_t)w * 64 + lane;
#pragma unroll
    for (int g = 0; g < 8; ++g) rgv[g] = rgp[(size_t)min(g, nG1 - 1) * 8 * 64];
  }
  const float bg = a.rgb[16 * w + j], bu = a.rub[o4];
  const float gate = a.blend ? sigmoid_f(a.blend[0]) : 0.f;   // g = sigmoid(weights_gru[l][t]) (:208); the backward's form
  NODE_STAMP(15);  // tanh + blend done, h' and x_t tiles stored
  __syncthreads();
  NODE_STAMP(16);
  if constexpr (SAVE && MODE == 1) {   // hc rows -> training buffer (SV is rewritten only behind the NEXT barrier)
#pragma unroll
    for (int it = 0; it < NS; ++it) {
      const int lb = srow + 32 * it, b = rowBase + lb;
      if (b < a.rows)
        save16(a.svHC, ((size_t)b * a.Np + n) * 64 + sq * 4, *reinterpret_cast<const float4*>(&SV[(lb * 16 + (sq ^ (lb & 15))) * 4]));
    }
  }
  // GEMM 1: zr2 = sigmoid([x|h'] Wg + bg): wave w = column tile w (of 8), all row tiles
  f32x4 acc1[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) acc1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g < nG1) {
      const float* T = (g < ngx) ? XT : Hs;
      const int gg = (g < ngx) ? g : g - ngx;
      const float4 wv = rgv[g];
      float4 av[NRT];
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) av[rt] = *reinterpret_cast<const float4*>(&T[((rt * 16 + j) * 16 + ((4 * gg + kq) ^ j)) * 4]);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].x, wv.x, acc1[rt]);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].y, wv.y, acc1[rt]);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].z, wv.z, acc1[rt]);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) acc1[rt] = MFMA16(av[rt].w, wv.w, acc1[rt]);
    }
  }
  // the weights of GEMM 2 are requested now (the registers of GEMM 1's weights are free): they land under the sigmoids
  float4 ruv[8];
  {
    const float4* rup = reinterpret_cast<const float4*>(a.ru) + (size_t)ct * 64 + lane;
#pragma unroll
    for (int g = 0; g < 8; ++g) ruv[g] = rup[(size_t)min(g, nG1 - 1) * 4 * 64];
  }
  {
    // (round 4: the h' values a z2 wave multiplies by are read in one batch - 16 exposed LDS round trips before)
    const int o = 16 * w + j;
    float hp1[NRT][4];
    if (w < 4) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) hp1[rt][e] = Hs[swz(rt * 16 + 4 * kq + e, o, 16)];
    }
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc1[rt][e] = sigmoid16(acc1[rt][e] + bg);
    if (w < 4) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) ZH2[swz(rt * 16 + 4 * kq + e, o, 16)] = acc1[rt][e] * hp1[rt][e];
    } else {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) R2[swz(rt * 16 + 4 * kq + e, o - 64, 16)] = acc1[rt][e];
    }
  }
  NODE_STAMP(17);  // residual GEMM 1 + sigmoids
  __syncthreads();
  NODE_STAMP(18);
  if constexpr (SAVE) {   // every thread is past its hc rows: SV <- z2 (kept in the z2 waves' accumulators)
    if (w < 4) {
      const int o = 16 * w + j;
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int e = 0; e < 4; ++e) SV[swz(rt * 16 + 4 * kq + e, o, 16)] = acc1[rt][e];
    }
  }
  // GEMM 2: hc2 = tanh([x | z2*h'] Wu + bu): wave (ct, rp) -> column tile ct, the row tiles of half rp
  f32x4 acc2[NR2];
#pragma unroll
  for (int q = 0; q < NR2; ++q) acc2[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    if (g < nG1) {
      const float* T = (g < ngx) ? XT : ZH2;
      const int gg = (g < ngx) ? g : g - ngx;
      const float4 wv = ruv[g];
      float4 av[NR2];
#pragma unroll
      for (int q = 0; q < NR2; ++q)
        av[q] = *reinterpret_cast<const float4*>(&T[(((NR2 * rp + q) * 16 + j) * 16 + ((4 * gg + kq) ^ j)) * 4]);
#pragma unroll
      for (int q = 0; q < NR2; ++q) acc2[q] = MFMA16(av[q].x, wv.x, acc2[q]);
#pragma unroll
      for (int q = 0; q < NR2; ++q) acc2[q] = MFMA16(av[q].y, wv.y, acc2[q]);
#pragma unroll
      for (int q = 0; q < NR2; ++q) acc2[q] = MFMA16(av[q].z, wv.z, acc2[q]);
#pragma unroll
      for (int q = 0; q < NR2; ++q) acc2[q] = MFMA16(av[q].w, wv.w, acc2[q]);
    }
  }
  // the new state is gathered as a [ROWS][64] tile in LDS (over x_t, dead once every wave has left GEMM 2) and
  // written to the state and to Seq_l[t] as whole 256-byte rows
  NODE_STAMP(19);  // residual GEMM 2 issued
  __syncthreads();
  float* Out = XT;
  float* HC2t = ZH2;             // SAVE: hc2 tile (z2*h' is dead once every wave has left GEMM 2)
  {
    float hp2[NR2][4], rr2[NR2][4];   // read in one batch (round 4), then the tanhs, then the writes
#pragma unroll
    for (int q = 0; q < NR2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lb = (NR2 * rp + q) * 16 + 4 * kq + e;
        hp2[q][e] = Hs[swz(lb, o4, 16)];
        rr2[q][e] = R2[swz(lb, o4, 16)];
      }
#pragma unroll
    for (int q = 0; q < NR2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc2[q][e] = tanh16(acc2[q][e] + bu);
#pragma unroll
    for (int q = 0; q < NR2; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lb = (NR2 * rp + q) * 16 + 4 * kq + e;
        const float hc = acc2[q][e], hp = hp2[q][e], rr = rr2[q][e];
        if constexpr (SAVE) HC2t[swz(lb, o4, 16)] = hc;
        const float res = rr * hp + (1.0f - rr) * hc;
        Out[swz(lb, o4, 16)] = a.blend ? (gate * hp + (1.0f - gate) * res) : res;
      }
  }
  NODE_STAMP(20);  // tanh + blends
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NS; ++it) {                 // ROWS rows x 16 slots float4 over 512 threads
    const int lb = srow + 32 * it, b = rowBase + lb;
    if (b >= a.rows) continue;
    const int at = (lb * 16 + (sq ^ (lb & 15))) * 4;
    const float4 v = *reinterpret_cast<const float4*>(&Out[at]);
    store_wt16(a.hout, ((size_t)b * a.Np + n) * 64 + sq * 4, v);
    if (a.seq) store_wt16(a.seq, (size_t)b * a.seqRowStride + (size_t)n * 64 + sq * 4, v);
    if constexpr (SAVE) {
      const size_t sat = ((size_t)b * a.Np + n) * 64 + sq * 4;
      save16(a.svZ2, sat, *reinterpret_cast<const float4*>(&SV[at]));
      save16(a.svR2, sat, *reinterpret_cast<const float4*>(&R2[at]));
      save16(a.svHC2, sat, *reinterpret_cast<const float4*>(&HC2t[at]));
      if (a.seqDrop) {
        const float4 m = *reinterpret_cast<const float4*>(a.dropMask + (size_t)b * a.dropRowStride + (size_t)n * 64 + sq * 4);
        save16(a.seqDrop, (size_t)b * a.seqRowStride + (size_t)n * 64 + sq * 4,
               make_float4(v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w));
      }
    }
  }
#ifdef NODE_LAB_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  NODE_STAMP(21);
#endif
  NODE_STAMP_FLUSH(a);
}

// bf16 copy of a fragment-ordered weight stream (same indexing, half the bytes): 8 floats per thread
__global__ __launch_bounds__(256) void k_stream_to_bf16(const float* __restrict__ src, unsigned int* __restrict__ dst,
                                                        size_t octets) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= octets) return;
  const float4 v0 = *reinterpret_cast<const float4*>(src + i * 8), v1 = *reinterpret_cast<const float4*>(src + i * 8 + 4);
  const uint2 p0 = to_bf16x4(v0), p1 = to_bf16x4(v1);
  *reinterpret_cast<uint4*>(dst + i * 4) = make_uint4(p0.x, p0.y, p1.x, p1.y);
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
