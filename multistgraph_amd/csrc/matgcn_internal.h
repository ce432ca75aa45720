// matgcn_internal.h - kernel argument blocks shared by the kernels and the C-ABI host code.
#ifndef MATGCN_INTERNAL_H
#define MATGCN_INTERNAL_H
#include <stddef.h>
#include <stdint.h>

#define MATGCN_MAX_STACK 16   /* entries of the reference's support stack (identity included) this build maps */

// How the reference's support stack [I, S_1 orders.., S_2 orders.., ...] (MultiATGCN.py:94-103) maps onto what the
// kernels see.  Supports that are diagonal matrices (e.g. the similarity Laplacian -I when there are no static
// features, MultiATGCN.py:244-250) need no graph mix at all: S x = diag(s) x, so their weight rows are folded into
// the identity slot, scaled per node by the Chebyshev value t_order(s_n), and they disappear from the stack.
//
// cheb_order = 1 (MultiATGCN.py:65-70,94-108): weights_g / weights_pool have ONE entry along k, yet the stack still is
// [I, S_1, S_2, ...] (the Chebyshev loop at :98 is empty, :100 appends every first-order support) and einsum
// 'bnki,nkio->bno' broadcasts the single weight over all of them: y = ((I + sum_s S_s) x) W.  Here: the dense
// supports are SUMMED into one dense slot, the diagonal ones fold into the identity slot as usual, and every entry
// reads pool index 0 - several stack entries may alias one pool index, which is why the backward walks entries.
struct StackMap {
  int KtotOrig;                       // entries of weights_g / weights_pool along k
  int nKeep;                          // 1 (identity) + dense slots
  int keepK[MATGCN_MAX_STACK];        // pool index k of every kept slot (keepK[0] = 0)
  int nDiag;                          // folded (diagonal) slots
  int diagK[MATGCN_MAX_STACK];        // pool index k of every folded slot
  int diagOrder[MATGCN_MAX_STACK];    // Chebyshev order of that slot (1 = the support itself)
  const float* diagSrc[MATGCN_MAX_STACK];  // (N,N) first-order support whose diagonal feeds it
  int N;
};

// What the pool gradients of the backward walk: one entry per stack entry that draws weights from the pools
struct StackEntries {
  int n;
  int pool[2 * MATGCN_MAX_STACK];     // pool index k the entry reads
  int slot[2 * MATGCN_MAX_STACK];     // node-GEMM slot whose weight gradient it takes (0 = identity)
  int diag[2 * MATGCN_MAX_STACK];     // index into StackMap.diag* for a folded diagonal entry, else -1
};

// One launch of k_prep_stream writes, for a group of nodes, one piece of their weight streams:
//   val(n, row, o) = sum_d E[n][d] * g_k * Wpool[d][k][i][o]   with (k, i) decoded from the row by `kind`
struct PrepStream {
  const float* E;        // node_emb (N, d)
  const float* wpool;    // (d, KtotOrig, I, O)
  const float* bpool;    // (d, O) - bias row of kind 1
  const float* wg;       // weights_g (KtotOrig) or null (no stack scaling)
  float* out;
  long nodeStride;       // floats between nodes in the destination
  long baseOfs;          // float offset of this piece inside a node's stream
  int d, I, O, N;
  int kind;              // 0: recurrent rows, 16x16x4 order   rows kk -> slot kk/64, channel iOfs + kk%64
                         // 1: layer-0 folded x rows, 16x16x4 order   rows kk -> slot kk/C0, channel kk%C0, then bias row
  int iOfs, C0;
  int groups;            // k-groups of 16 rows
  int groupsX;           // k_prep_mfma: kind-1 k-groups appended behind the `groups` kind-0 ones (same launch)
  int OTdst, otOfs;      // OTdst > 0: tiles per fragment row in the destination and first tile of this piece
                         // (the 192-column x-part stream takes the gate in tiles 0..7 and the update in 8..11)
  StackMap map;
};

struct FuseArgs {
  const float* X;        // windows (B, xSteps, N, F), or the raw series (steps, N, F) when labelStart != null
  const int* labelStart; // device (B) label starts, or null
  int rel[256];          // series mode: row offsets relative to the label start (MATGCN_MAX_XSTEPS)
  long seriesSteps;      // series mode: rows of the series (an out-of-range row is clamped and counted, never read)
  float* x0;           // [B][T][Np][C0]
  const float* tsg;
  const float* ts[8];
  int B, T, N, Np, C0, od, F, xSteps, startDim, nHeads, nTs;
  int headBegin[8];
  int extSrc[16];
};

struct MixArgs {
  const float* St;
  int ldS;
  const float* X;
  long xTileStride;
  int ldX;
  float* out;
  long sN, sK, sT;
  long outFloats;        // extent of `out` (bounds the write-through buffer descriptor; 0: plain stores)
  int Np, N, Ks, nK, nColTiles, nRowTiles;
  // split reduction (blockIdx.y = part): part p multiplies reduction rows [p*nK*16, ..) of St with the matching rows
  // of X and writes its own partial result - the transposed mix of the backward splits by support slot
  int parts = 1;
  long aPartStride = 0, xPartStride = 0, outPartStride = 0;
#ifdef NODE_LAB_STAMPS
  unsigned int* stamps = nullptr;   // lab builds: per-wave phase stamps of this launch
#endif
};

struct HeadArgs {
  const float* seq;      // [B][T][Np][64]
  const float* w;        // fragment-ordered [T*8][NTc][64][4]
  const float* bias;
  float* out;
  int B, T, N, Np, CH, od, NTc;   // seq is time-major [T][B][Np][64]
};

// ---- lab only: in-kernel phase stamps (s_memtime at wave granularity, kept in SGPRs, written once at the end) ----
#ifdef NODE_LAB_STAMPS
#define NODE_STAMPS 28
struct NodeStamps {
  unsigned int t[NODE_STAMPS];
  __device__ __forceinline__ void at(int i) {
    t[i] = (unsigned int)__builtin_amdgcn_s_memtime();
    if (i == 0) t[24] = (unsigned int)__builtin_amdgcn_s_memrealtime();
  }
  __device__ __forceinline__ void flush(unsigned int* out) {
    if (!out) return;
    t[25] = (unsigned int)__builtin_amdgcn_s_memrealtime();
    t[26] = __builtin_amdgcn_s_getreg(63492);   // HW_REG_HW_ID
    t[27] = __builtin_amdgcn_s_getreg(63508);   // HW_REG_XCC_ID
    if ((threadIdx.x & 63) == 0) {
      unsigned int* o = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * NODE_STAMPS;
#pragma unroll
      for (int i = 0; i < NODE_STAMPS; ++i) o[i] = t[i];
    }
  }
};
#define NODE_STAMP_DECL NodeStamps nst; for (int i_ = 0; i_ < NODE_STAMPS; ++i_) nst.t[i_] = 0;
#define NODE_STAMP(i) nst.at(i)
#define NODE_STAMP_ARG , nst
#define NODE_STAMP_PARAM , NodeStamps& nst
#define NODE_STAMP_FLUSH(a) nst.flush((a).stamps)
#else
#define NODE_STAMP_DECL
#define NODE_STAMP(i)
#define NODE_STAMP_ARG
#define NODE_STAMP_PARAM
#define NODE_STAMP_FLUSH(a)
#endif

#endif
