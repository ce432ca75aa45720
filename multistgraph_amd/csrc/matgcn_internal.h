// matgcn_internal.h - kernel argument blocks shared by the kernels and the C-ABI host code.
#ifndef MATGCN_INTERNAL_H
#define MATGCN_INTERNAL_H
#include <stddef.h>
#include <stdint.h>

struct PrepAgcn {
  const float* E;      // node_emb (N, d)
  const float* wpool;  // (d, Ktot, I, O)
  const float* bpool;  // (d, O)
  const float* wg;     // weights_g (Ktot) or null (no stack scaling)
  float* out;
  long nodeStride;     // floats between nodes in the destination stream
  long streamOfs;      // float offset of this part inside a node's stream
  int d, Ktot, I, O;
  int iOfs;            // first input channel of this part inside I
  int mode;            // 0: rows j -> (k = j / Cw, i = iOfs + j % Cw); 1: folded x rows + bias row
  int Cw;              // channels per support slot in this part
  int rows;            // padded row count (multiple of 8)
  int OTsrc;           // O / 32
  int OTdst, otOfs;    // tiles per fragment row in the destination, first tile of this part
};

struct FuseArgs {
  const float* X;
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
  int Np, N, Ks, nK, nColTiles, nRowTiles;
};

struct NodeArgs {
  const float* xa;       // folded x-part rows (layer 0) or null
  long xaNodeStride, xaRowStride;
  int xaLen;
  const float* ident;    // identity slot rows: [row][Np][64] (+ n*64)
  long identRowStride;
  const float* g;        // mixed slots [N][rows][Ks][64]
  int Ks;
  const float* w;        // fragment-ordered weights, per node
  long wNodeStride;
  int rows;              // rows per node (B, or B*T for k_px)
  int N, Np, T;
  const float* px;       // hoisted pre-activations of this step: [N][rows][192], or null
  float* raw;            // optional (B,N,128) pre-activation dump (unit entry point)
  float* zh;             // k_gate: z*h out [rows][Np][64]
  float* r;              // k_gate: r out / k_update: r in  [N][rows][64]
  float* hstate;         // k_update: h in / h' out [rows][Np][64]
  const float* bias;     // k_px: [N][192]
  float* pxOut;          // k_px: [T][N][B][192]
};

struct ResArgs {
  const float* x;        // x_t rows: x[b*xRowStride + n*C + c]
  long xRowStride;
  int C, Cpad;
  const float* h;        // h' [B][Np][64]
  float* hout;           // [B][Np][64]
  float* seq;            // Seq_l at step t (or null): seq[b*seqRowStride + n*64 + o]
  long seqRowStride;
  const float* wg;       // fragment-ordered gate weight  [K/8][4][64][4]
  const float* bg;       // (128)
  const float* wu;       // fragment-ordered update weight [K/8][2][64][4]
  const float* bu;       // (64)
  const float* blend;    // &weights_gru[l][t] or null (plain GRUCell output)
  int B, N, Np;
};

struct HeadArgs {
  const float* seq;      // [B][T][Np][64]
  const float* w;        // fragment-ordered [T*8][NTc][64][4]
  const float* bias;
  float* out;
  int B, T, N, Np, CH, od, NTc;
};

#endif
