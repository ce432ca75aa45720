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

struct NodeArgs {        // k_px: hoisted x-part of layers >= 1
  const float* xa;       // unused (kept for the shared main loop): folded rows or null
  long xaNodeStride, xaRowStride;
  int xaLen;
  const float* ident;    // identity slot rows: [row][Np][64] (+ n*64)
  long identRowStride;
  const float* g;        // mixed slots [N][rows][Ks][64]
  int Ks;
  const float* w;        // fragment-ordered (32x32x2) x-part weights, per node
  long wNodeStride;
  int rows;              // rows per node = B * (steps of the chunk)
  int N, Np, B;          // B: batch rows per step (row -> (t, b), t-major)
  const float* bias;     // [N][192]
  float* pxOut;          // [Tc][N][B][192] slice of PX
};

struct HeadArgs {
  const float* seq;      // [B][T][Np][64]
  const float* w;        // fragment-ordered [T*8][NTc][64][4]
  const float* bias;
  float* out;
  int B, T, N, Np, CH, od, NTc;   // seq is time-major [T][B][Np][64]
};

#endif
