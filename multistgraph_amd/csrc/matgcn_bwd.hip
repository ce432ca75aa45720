// matgcn_bwd.hip - host schedule of the training step (included at the end of matgcn_capi.hip).
//
// matgcn_forward_train = matgcn_forward that also keeps z, r, hc of the graph cell and z2, r2, hc2 of the residual
// cell of every (layer, step); matgcn_backward back-propagates d_out through the head, the encoder and the head
// fusion down to every parameter of the reference state_dict (autograd of MultiATGCN.py:363-420).
//
// Schedule of the encoder backward, per layer from the top:
//   chain  t = T-1 .. 0   only what is sequential in time: residual cell and graph cell algebra, the h columns of
//                         both AGCNs (node GEMM with the transposed weights, transposed graph mix); the
//                         pre-activation gradients of every step are kept
//   batch  over all T     x columns -> gradient of the layer's input sequence; weight gradients as per-node GEMMs
//                         with K = T*B; gradient of the adaptive adjacency; residual nn.Linear gradients
// and at the end the parameter-only part: node-adaptive weights -> pools / node_emb / weights_g, adaptive adjacency
// -> node_vec1/2 (or node_emb), head fusion -> weight_ts / weight_tsg.
//
// Every configuration of the forward trains: any cheb_order (the recursion is back-propagated into the adaptive
// adjacency), gcn_off (dense GRU layers), fnn_off (head over the last step), node_specific_off (node_emb frozen).

namespace {

GemmArgs gemm_args(const float* A, const float* B, float* C, int M, int N, int K) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.K2 = 1;
  g.nb2 = 1; g.alpha = 1.f; g.beta = 0.f; g.mode = 0; g.split = 1;
  return g;
}

template <int ROLE>
void launch_bgemm(const dim3& grid, hipStream_t s, const GemmArgs& g) {
  hipLaunchKernelGGL(k_bgemm<ROLE>, grid, dim3(256), 0, s, g);
}

// the fast path of k_bgemm_tn: A unit-stride along M, B unit-stride along N, whole 64 x 64 tiles, 16-byte friendly
bool tn_eligible(const GemmArgs& g) {
  auto m4 = [](long v) { return (v & 3) == 0; };
  return g.sAm == 1 && g.sBn == 1 && g.M % 64 == 0 && g.N % 64 == 0 && m4(g.sAk) && m4(g.sAk2) && m4(g.sBk) &&
         m4(g.sBk2) && m4(g.bA1) && m4(g.bA2) && m4(g.bB1) && m4(g.bB2) &&
         (reinterpret_cast<size_t>(g.A) & 15) == 0 && (reinterpret_cast<size_t>(g.B) & 15) == 0;
}

template <int ROLE>
void launch_bgemm_tn(const dim3& grid, hipStream_t s, const GemmArgs& g) {
  hipLaunchKernelGGL(k_bgemm_tn<ROLE>, grid, dim3(256), 0, s, g);
}

int gemm(const GemmArgs& g, int nb1, hipStream_t s, int role = BG_GENERIC) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.K2 <= 0 || nb1 <= 0) return MATGCN_OK;
  const long gx = (g.N + 63) / 64, gy = (g.M + 63) / 64, gz = (long)nb1 * g.nb2 * g.split;
  if (gy > 65535 || gz > 65535 || (long)g.K2 * ((g.K + BG_KT - 1) / BG_KT) >= (1L << 30)) return MATGCN_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)gz);
  if (tn_eligible(g)) {
    switch (role) {
      case BG_WGRAD: launch_bgemm_tn<BG_WGRAD>(grid, s, g); return launch_ok();
      case BG_LINEAR: launch_bgemm_tn<BG_LINEAR>(grid, s, g); return launch_ok();
      case BG_GENERIC: launch_bgemm_tn<BG_GENERIC>(grid, s, g); return launch_ok();   // matgcn_debug_gemm: the test's way in
      default: break;
    }
  }
  switch (role) {
    case BG_CHAIN_DENSE: launch_bgemm<BG_CHAIN_DENSE>(grid, s, g); break;
    case BG_CHAIN_NODE: launch_bgemm<BG_CHAIN_NODE>(grid, s, g); break;
    case BG_CHAIN_MIX: launch_bgemm<BG_CHAIN_MIX>(grid, s, g); break;
    case BG_X_NODE: launch_bgemm<BG_X_NODE>(grid, s, g); break;
    case BG_X_MIX: launch_bgemm<BG_X_MIX>(grid, s, g); break;
    case BG_WGRAD: launch_bgemm<BG_WGRAD>(grid, s, g); break;
    case BG_ADJ: launch_bgemm<BG_ADJ>(grid, s, g); break;
    case BG_LINEAR: launch_bgemm<BG_LINEAR>(grid, s, g); break;
    case BG_POOL: launch_bgemm<BG_POOL>(grid, s, g); break;
    case BG_HEAD: launch_bgemm<BG_HEAD>(grid, s, g); break;
    default: launch_bgemm<BG_GENERIC>(grid, s, g); break;
  }
  return launch_ok();
}

struct Bwd {
  Ctx c;
  const float* X;
  const float* dropMask;
  const matgcn_grads* g;
  float* tr;
  const matgcn_series* src = nullptr;   // series mode: X rows are gathered from the raw series
  bool hasH0 = false;     // the forward started from a caller-supplied state (its padded copy sits at tr + R.oH0)
  float* dH0 = nullptr;   // (L, B, N, H) gradient w.r.t. that state, or null
};

// the stack entries whose weights come out of the pools: kept slots first, then the folded diagonal ones (which share
// the identity slot); with cheb_order = 1 several entries alias pool index 0 (StackMap)
StackEntries stack_entries(const StackMap& map) {
  StackEntries e;
  memset(&e, 0, sizeof(e));
  for (int s2 = 0; s2 < map.nKeep; ++s2) { e.pool[e.n] = map.keepK[s2]; e.slot[e.n] = s2; e.diag[e.n] = -1; ++e.n; }
  for (int q = 0; q < map.nDiag; ++q) { e.pool[e.n] = map.diagK[q]; e.slot[e.n] = 0; e.diag[e.n] = q; ++e.n; }
  return e;
}

// dst[rows][m][i] = sum_kk StP[kk][m] * src[rows][slot 1..][kk][i]: the transposed graph mix of the dense slots of a
// [rows][S][Np][Cc] gradient.  With 64 feature columns this is the forward's graph-mix kernel run on the plain stack
// (reduction over (k, n), one column tile per row); narrower inputs (layer 0) take the generic GEMM.
// split: the reduction is cut by support slot - Ks times the workgroups, each with a K loop of the forward's length -
// and slot k's partial result lands Ks-th part k of dst (parts `partStride` floats apart; the consumers add them up).
int mix_transposed(const Bwd& b, const float* src, int rows, int Cc, float* dst, bool split = false, long partStride = 0) {
  const Plan& P = b.c.P;
  if (P.Ks <= 0) return MATGCN_OK;
  const int S = b.c.R.S;
  if (Cc == H) {
    MixArgs a;
    a.St = b.tr + b.c.R.oStP; a.ldS = P.NpC;
    a.X = src + (size_t)P.Np * H; a.xTileStride = (long)S * P.Np * H; a.ldX = H;
    a.out = dst; a.sN = H; a.sK = 0; a.sT = (long)P.Np * H;
    const long outFloats = (long)rows * P.Np * H;
    a.outFloats = outFloats < (1L << 29) ? outFloats : 0;
    a.Np = P.NpC; a.N = P.N; a.Ks = 1; a.nK = P.Ks * P.Np / 16; a.nColTiles = rows;
    a.nRowTiles = P.NpC / 64;
    if (split && P.Ks > 1) {
      a.parts = P.Ks; a.nK = P.Np / 16;
      a.aPartStride = (long)P.Np * P.NpC; a.xPartStride = (long)P.Np * H; a.outPartStride = partStride;
    }
    if ((rows & 1) == 0) {   // 32-row x 128-column tiles: 3 % row padding instead of 10 %, 4.9 workgroups per CU (k_mix_n32)
      a.nRowTiles = (P.N + 31) / 32;
      hipLaunchKernelGGL(k_mix_n32, dim3((unsigned)(a.nRowTiles * (rows / 2)), (unsigned)a.parts), dim3(256), 0, b.c.s, a);
    } else {
      hipLaunchKernelGGL(k_mix<2>, dim3((unsigned)(a.nRowTiles * rows), (unsigned)a.parts), dim3(256), 0, b.c.s, a);
    }
    return launch_ok();
  }
  GemmArgs g = gemm_args(b.c.prep + P.oSt, src + (size_t)P.Np * Cc, dst, P.N, Cc, P.Ks * P.Np);
  g.sAm = P.Mp; g.sAk = 1;
  g.sBk = Cc; g.sBn = 1; g.bB1 = (long)S * P.Np * Cc;
  g.sCm = Cc; g.sCn = 1; g.bC1 = (long)P.Np * Cc;
  return gemm(g, rows, b.c.s, rows > P.B ? BG_X_MIX : BG_CHAIN_MIX);
}

// dA[rows][s][n][i] (+)= dPre[rows][n][0:O] . Wp[n][s][iOfs + i][0:O]^T for every node and slot
// (nodeMajor: dA is laid out [s][n][rows][Cc] instead - used for the narrow x columns of layer 0, whose transposed
// mix then is ONE GEMM with rows*Cc columns instead of `rows` GEMMs with Cc columns)
int node_gemm_transposed(const Bwd& b, const float* dPre, int O, const float* Wp, int I, int iOfs, int Cc, int rows,
                         float* dA, float beta, bool nodeMajor = false) {
  const Plan& P = b.c.P;
  const int S = b.c.R.S;
  GemmArgs g = gemm_args(dPre, Wp + (size_t)iOfs * O, dA, rows, Cc, O);
  g.sAm = (long)P.Np * O; g.sAk = 1; g.bA1 = O;
  g.sBk = 1; g.sBn = O; g.bB1 = (long)S * I * O; g.bB2 = (long)I * O;
  g.sCm = (long)S * P.Np * Cc; g.sCn = 1; g.bC1 = Cc; g.bC2 = (long)P.Np * Cc;
  if (nodeMajor) { g.sCm = Cc; g.bC1 = (long)rows * Cc; g.bC2 = (long)P.Np * rows * Cc; }
  g.nb2 = S; g.beta = beta;
  return gemm(g, P.N, b.c.s, rows > P.B ? BG_X_NODE : BG_CHAIN_NODE);
}

// the same contraction on the dedicated node kernel (64 hidden / input columns per slot, whole 16-byte rows):
// dA[rows][s][n][0:64] (+)= dPre[rows][n][0:O] . Wp[n][s][iOfs .. iOfs+63][0:O]^T
// (dPreG / WpG: the gate AGCN's 128 columns, dPreU / WpU: the update AGCN's 64 - ONE contraction over all 192, so the
// output block - 627 MB for the 23 x-column steps of a layer at BM / B = 64 - is written once instead of written, read
// and written again)
int node_contract(const Bwd& b, const float* dPreG, const float* WpG, const float* dPreU, const float* WpU, int I,
                  int iOfs, int rows, float* dA, float beta) {
  const Plan& P = b.c.P;
  ChainNodeArgs cn;
  memset(&cn, 0, sizeof(cn));
  cn.dPre = dPreG; cn.Wp = WpG; cn.dPre2 = dPreU; cn.Wp2 = WpU; cn.dA = dA; cn.I = I; cn.iOfs = iOfs; cn.rows = rows;
  cn.N = P.N; cn.Np = P.Np; cn.S = b.c.R.S; cn.beta = beta;
  const dim3 grid((unsigned)((rows + 63) / 64), (unsigned)P.N);
  hipLaunchKernelGGL((k_chain_node<false, 192>), grid, dim3(512), 0, b.c.s, cn);
  return launch_ok();
}

// where the forward left the graph-mixed rows G[s'][n][(k2, k)][i] of a range of steps: strides in floats
struct MixedRows {
  const float* G;
  long sNode, sSlot, sK2, sK;
  int K2, K;             // (k2, k) enumerate the rows of the range in the order of dPre's rows
};

// dWp[n][s][iOfs + i][o] += sum_rows [U | mix(U)][rows][n][s][i] * dPre[rows][n][o] over `rows` consecutive (t, b) rows
// starting at row0; U rows are [Np][Cc] slabs, the mixed rows come as the forward stored them
// (parts: 1 = identity slot only, 2 = dense slots only, 3 = both)
int node_weight_grad(const Bwd& b, const float* U, const MixedRows& mr, int Cc, const float* dPre, int O, int I, int iOfs,
                     long row0, int rows, float* dWp, int parts = 3) {
  const Plan& P = b.c.P;
  const int S = b.c.R.S;
  const float* dP = dPre + (size_t)row0 * P.Np * O;
  if (parts & 1) {  // identity slot: the rows themselves
    GemmArgs g = gemm_args(U + (size_t)row0 * P.Np * Cc, dP, dWp + (size_t)iOfs * O, Cc, O, rows);
    g.sAm = 1; g.sAk = (long)P.Np * Cc; g.bA1 = Cc;
    g.sBk = (long)P.Np * O; g.sBn = 1; g.bB1 = O;
    g.sCm = O; g.sCn = 1; g.bC1 = (long)S * I * O;
    g.beta = 1.f;
    RETURN_IF(gemm(g, P.N, b.c.s, BG_WGRAD));
  }
  if ((parts & 2) && P.Ks > 0 && mr.K > 0) {  // dense slots: the mixed rows
    GemmArgs g = gemm_args(mr.G, dP, dWp + (size_t)I * O + (size_t)iOfs * O, Cc, O, mr.K);
    g.K2 = mr.K2;
    g.sAm = 1; g.sAk = mr.sK; g.sAk2 = mr.sK2; g.bA1 = mr.sNode; g.bA2 = mr.sSlot;
    g.sBk = (long)P.Np * O; g.sBk2 = (long)mr.K * P.Np * O; g.sBn = 1; g.bB1 = O; g.bB2 = 0;
    g.sCm = O; g.sCn = 1; g.bC1 = (long)S * I * O; g.bC2 = (long)I * O;
    g.nb2 = P.Ks; g.beta = 1.f;
    RETURN_IF(gemm(g, P.N, b.c.s, BG_WGRAD));
  }
  return MATGCN_OK;
}

// where the forward left the graph mixes of every step of a sequence (see k_wgrad_node): a block per step
struct StepBlocks {
  const float* g[MAX_STEPS];
  long gNode[MAX_STEPS];
};
// private blocks [T][N][B][Ks][64] (recurrent rows of the top layer, z*h rows)
StepBlocks uniform_blocks(const Plan& P, const float* base) {
  StepBlocks sb;
  const long gStep = (long)P.N * P.B * P.Ks * H;
  for (int t = 0; t < P.T; ++t) { sb.g[t] = base + (size_t)t * gStep; sb.gNode[t] = (long)P.B * P.Ks * H; }
  return sb;
}
// the chunk blocks of a hoisted x part: [chunk][N][nt*B][Ks][64]
StepBlocks chunk_blocks(const Plan& P, const float* base) {
  StepBlocks sb;
  const long gStep = (long)P.N * P.B * P.Ks * H;
  for (int t0 = 0; t0 < P.T;) {
    const int nt = chunk_steps(P, t0);
    for (int t = t0; t < t0 + nt; ++t) {
      sb.g[t] = base + (size_t)t0 * gStep + (size_t)(t - t0) * P.B * P.Ks * H;
      sb.gNode[t] = (long)nt * P.B * P.Ks * H;
    }
    t0 += nt;
  }
  return sb;
}
// all slots of a node in one workgroup (k_wgrad_node); false = shape outside what the kernel covers
bool node_wgrad_fast(const Bwd& b, const float* U, const StepBlocks& sb, const float* dPre, int O, int I, int iOfs,
                     float* dWp, int* rc, float* dBias = nullptr) {
  const Plan& P = b.c.P;
  const int S = b.c.R.S;
  if (S != P.Ks + 1 || S > 5 || S == 3 || S < 2 || P.T > MAX_STEPS || (O != 128 && O != 64)) return false;
  WgradNodeArgs a;
  memset(&a, 0, sizeof(a));
  a.U = U; a.dPre = dPre; a.dW = dWp + (size_t)iOfs * O; a.dBias = dBias;
  for (int t = 0; t < P.T; ++t) { a.g[t] = sb.g[t]; a.gNode[t] = sb.gNode[t]; }
  a.T = P.T; a.B = P.B; a.N = P.N; a.Np = P.Np; a.S = S; a.Ks = P.Ks; a.I = I;
  // parts: enough workgroups for a few full rounds of the chip (2 resident per CU at O = 128, 3 at O = 64) without a
  // mostly empty last one - at BM (403 nodes, 24 steps): 4 x 403 = 3.1 rounds of 512, 5 x 403 = 2.6 rounds of 768
  const int parts = O == 128 ? 4 : 5;
  a.stepsPerPart = P.T >= 8 ? (P.T + parts - 1) / parts : P.T;
  const dim3 grid((unsigned)P.N, (unsigned)((P.T + a.stepsPerPart - 1) / a.stepsPerPart));
  const size_t lds = (size_t)2 * WG_KT * ((S * 64 + 16) + (O + 16)) * sizeof(float);
  const dim3 block((unsigned)(S * O));
#define WGN(O_, S_, W_) hipLaunchKernelGGL((k_wgrad_node<O_, S_, W_>), grid, block, lds, b.c.s, a)
  if (O == 128) {
    if (S == 4) WGN(128, 4, 4);        // 512 threads, 128 registers: two workgroups per compute unit
    else if (S == 5) WGN(128, 5, 2);
    else if (S == 2) WGN(128, 2, 3);
    else return false;
  } else {
    if (S == 4) WGN(64, 4, 3);
    else if (S == 5) WGN(64, 5, 3);
    else if (S == 2) WGN(64, 2, 3);
    else return false;
  }
#undef WGN
  *rc = launch_ok();
  return true;
}

// dT_j[n][m] += sum_{rows, i} dA[rows][slot 1 + j][n][i] * U[rows][m][i] for the `per` Chebyshev orders of the adaptive
// adjacency (first-order support 0 is never diagonal: its orders are the dense slots 0 .. per-1)
int adaptive_grad(const Bwd& b, const float* dA, const float* U, int rows, int Cc, float* dT, bool nodeMajor = false) {
  const Plan& P = b.c.P;
  const int S = b.c.R.S;
  for (int j = 0; j < P.per; ++j) {
    const size_t slot = nodeMajor ? (size_t)(1 + j) * P.Np * rows * Cc : (size_t)(1 + j) * P.Np * Cc;
#ifndef ADJ_LAB_GENERIC
    if (!nodeMajor && Cc == H && rows >= 8) {   // both operands in 256-byte node rows: the dedicated kernel (k_adj_grad)
      AdjGradArgs q;
      q.A = dA + slot; q.B = U; q.aStride = (long)S * P.Np * H; q.bStride = (long)P.Np * H;
      q.R = rows; q.N = P.N; q.Np = P.Np; q.dT = dT + (size_t)j * P.N * P.N;
      const unsigned tiles = (unsigned)((P.Np + 63) / 64);
      // one resident set of workgroups (5 per CU at 32 KB of LDS): 1 280 / tiles^2 slices of the row blocks
      int split = (int)(1280 / (tiles * tiles));
      if (split < 1) split = 1;
      if (split > rows) split = rows;
      hipLaunchKernelGGL(k_adj_grad, dim3(tiles, tiles, (unsigned)split), dim3(256), 0, b.c.s, q);
      CHECK_LAUNCH();
      continue;
    }
#endif
    if (nodeMajor && Cc == 2 && rows % 8 == 0) {   // layer 0's two-channel x part: dedicated small kernel
      const unsigned t32 = (unsigned)((P.N + 31) / 32);
      hipLaunchKernelGGL(k_adj_grad_narrow2, dim3(t32, t32, 2), dim3(256), 0, b.c.s, dA + slot, U, rows, P.N, P.Np, 2,
                         dT + (size_t)j * P.N * P.N);
      CHECK_LAUNCH();
      continue;
    }
    GemmArgs g = gemm_args(dA + slot, U, dT + (size_t)j * P.N * P.N, P.N, P.N, Cc);
    g.K2 = rows;
    g.sAm = Cc; g.sAk = 1; g.sAk2 = (long)S * P.Np * Cc;
    if (nodeMajor) { g.sAm = (long)rows * Cc; g.sAk2 = Cc; }
    g.sBk = 1; g.sBn = Cc; g.sBk2 = (long)P.Np * Cc;
    g.sCm = P.N; g.sCn = 1;
    g.mode = 1; g.split = 48;
    RETURN_IF(gemm(g, 1, b.c.s, BG_ADJ));
  }
  return MATGCN_OK;
}

// nn.Linear weight gradient: dW[o][iOfs + i] = sum_rows dPre[rows][o] * In[rows][i]   (rows = every (t, b, n))
// (bias / biasDone: the bias gradient = the column sums of dPre falls out of the fast kernel's operand stream; *biasDone
// says whether this call took it along - the caller runs k_colsum_all otherwise)
int linear_weight_grad(const Bwd& b, const float* dPre, int O, const float* In, int Cc, long rows, int I, int iOfs,
                       float* dW, float* bias = nullptr, bool* biasDone = nullptr) {
  GemmArgs g = gemm_args(dPre, In, dW + iOfs, O, Cc, (int)rows);   // rows < 2^31 is checked by the caller
  g.sAm = 1; g.sAk = O;
  g.sBk = Cc; g.sBn = 1;
  g.sCm = I; g.sCn = 1;
  g.mode = 1; g.split = 256;
  if (bias && tn_eligible(g)) { g.colsumA = bias; *biasDone = true; }
  return gemm(g, 1, b.c.s, BG_LINEAR);
}

// ---- one backward pass: what its stages share ---------------------------------------------------------------
struct Pass {
  Bwd b, bw, bx;             // bw: the same context on the second stream (weight gradients), bx: on the x-column stream
  hipStream_t s, ws, xs;
  int chunk;                 // steps per x-column chunk of the layers above the first (see bwd_x_chunk)
  bool twoStreams, adp;
  int hT, tOff;              // fnn_off: the head sees the last step only (MultiATGCN.py:412)
  int fusedLds;
  StackMap map;
  StackEntries ent;
};

// per-layer views of the scratch (index l & 1: the weight gradients of layer l run on the second stream while the
// chain of layer l-1 already fills the other set)
struct LayerBufs {
  int l, C, I, par;
  const float* seq;          // Seq_l
  const float* h0;           // h_{-1} of the layer [B][Np][64], or null = zeros
  float* dSeqCur;            // gradient of Seq_l
  float* dXall;              // gradient of the layer's input sequence (dSeq of the layer below / dX0)
  float *DPU, *DPG, *DPU2, *DPG2, *DAg, *DAu, *DAx;
  const float *WpG, *WpU, *RG, *RU;
  bool mergeAbove;           // the gate block already holds the x-column gradient of the layer above
  bool narrow;               // layer 0 with a handful of input channels
};

#define PASS_LOCALS(q)                                                                                             \
  [[maybe_unused]] const Bwd& b = (q).b; [[maybe_unused]] const Bwd& bw = (q).bw;                                   \
  [[maybe_unused]] const Ctx& c = (q).b.c; [[maybe_unused]] const Plan& P = (q).b.c.P;                              \
  [[maybe_unused]] const TrainPlan& R = (q).b.c.R; [[maybe_unused]] const matgcn_dims* D = (q).b.c.D;               \
  [[maybe_unused]] const matgcn_params* prm = (q).b.c.prm; [[maybe_unused]] const matgcn_grads* g = (q).b.g;        \
  [[maybe_unused]] float* tr = (q).b.tr; [[maybe_unused]] hipStream_t s = (q).s; [[maybe_unused]] hipStream_t ws = (q).ws; \
  [[maybe_unused]] const int S = R.S, T = P.T, B = P.B, Np = P.Np, N = P.N;                                         \
  [[maybe_unused]] const long slab = (long)B * Np * H; [[maybe_unused]] const int rowsTB = T * B;                   \
  [[maybe_unused]] const bool twoStreams = (q).twoStreams, adp = (q).adp;                                           \
  [[maybe_unused]] const int hT = (q).hT, tOff = (q).tOff, fusedLds = (q).fusedLds;                                 \
  [[maybe_unused]] const StackMap& map = (q).map; [[maybe_unused]] const StackEntries& ent = (q).ent;               \
  [[maybe_unused]] float* dT = tr + R.oDT

#define LAYER_LOCALS(L)                                                                                            \
  [[maybe_unused]] const int l = (L).l, C = (L).C, I = (L).I, par = (L).par;                                        \
  [[maybe_unused]] const float* seq = (L).seq; [[maybe_unused]] float* dSeqCur = (L).dSeqCur;                       \
  [[maybe_unused]] const float* h0 = (L).h0;                                                                        \
  [[maybe_unused]] float* dXall = (L).dXall;                                                                        \
  [[maybe_unused]] float *DPU = (L).DPU, *DPG = (L).DPG, *DPU2 = (L).DPU2, *DPG2 = (L).DPG2, *DAg = (L).DAg,        \
                         *DAu = (L).DAu, *DAx = (L).DAx;                                                            \
  [[maybe_unused]] const float *WpG = (L).WpG, *WpU = (L).WpU, *RG = (L).RG, *RU = (L).RU;                          \
  [[maybe_unused]] const bool mergeAbove = (L).mergeAbove, narrow = (L).narrow;                                     \
  [[maybe_unused]] float* DH = tr + R.oDH[par]; [[maybe_unused]] float* DHa = tr + R.oDHa[par];                     \
  [[maybe_unused]] float* TMP = tr + R.oTmp[par]; [[maybe_unused]] float* MixOut = tr + R.oMixOut[par]

#ifndef CHAIN_FUSE_RES_NODE
#define CHAIN_FUSE_RES_NODE 1   // 0: round 3's pair k_chain_res_fused + k_chain_node<false, 64> (A/B builds)
#endif
// scratch and outputs start from zero (only what is accumulated into, or what the GEMMs leave untouched)
int bwd_clear(Pass& pass) {
  PASS_LOCALS(pass);
  // (only what is accumulated into, or what the GEMMs leave untouched: the rows of the padding nodes)
  // (the accumulators of the node-adaptive weight gradients - 250 MB - are written and read on the weight-gradient
  // stream only: cleared there, behind its fork, instead of in front of the first chain)
  for (int l = 0; l < P.L; ++l)
    for (int part = 0; part < 2; ++part) {
      const long O = part == 0 ? 128 : 64, I = P.Cl[l] + H;
      RETURN_IF(zero_async(tr + R.oDWp[l][part], (long)N * S * I * O, ws));
      RETURN_IF(zero_async(tr + R.oDBias[l][part], (long)N * O, ws));
    }
  if (hT < T) {
    RETURN_IF(zero_async(tr + R.oDSeq[0], (long)T * slab, s));
  } else if (Np != N) {   // the head GEMM writes every step of every real node: only the padding rows need clearing
    hipLaunchKernelGGL(k_zero_pad_rows, dim3(blocks_for((size_t)rowsTB * (Np - N) * H)), dim3(256), 0, s, tr + R.oDSeq[0],
                       rowsTB, N, Np, H);
    CHECK_LAUNCH();
  }
  for (int q = 0; q < (P.L > 1 ? 2 : 1); ++q) RETURN_IF(zero_async(tr + R.oMixOut[q], slab * (P.Ks > 1 ? P.Ks : 1), s));
  RETURN_IF(zero_async(tr + R.oDT, (long)P.per * N * N, s));
  if (Np != N) {   // one launch for the padding rows of both scratch sets
    PadRowBufs bufs;
    int nb = 0;
    for (int q = 0; q < (P.L > 1 ? 2 : 1); ++q) {
      bufs.p[nb++] = tr + R.oDAg[q]; bufs.p[nb++] = tr + R.oDAu[q]; bufs.p[nb++] = tr + R.oDAx[q];
    }
    hipLaunchKernelGGL(k_zero_pad_rows_multi, dim3(blocks_for((size_t)rowsTB * S * (Np - N) * H / 4), (unsigned)nb),
                       dim3(256), 0, s, bufs, rowsTB * S, N, Np, H);
    CHECK_LAUNCH();
  }
  if (CHAIN_FUSE_RES_NODE && !P.gcnOff) RETURN_IF(zero_async(tr + R.oZeroSlab, slab, s));
  if (CHAIN_FUSE_RES_NODE && Np != N && !P.gcnOff) {
    // k_chain_res_node writes the rows of the real nodes only; DPU2 / DPG2 are summed over ALL rows by the residual
    // nn.Linear gradients (column sums, weight GEMMs): the rows of the padding nodes must read as zero
    for (int q = 0; q < (P.L > 1 ? 2 : 1); ++q) {
      hipLaunchKernelGGL(k_zero_pad_rows, dim3(blocks_for((size_t)rowsTB * (Np - N) * H)), dim3(256), 0, s, tr + R.oDPU2[q],
                         rowsTB, N, Np, H);
      CHECK_LAUNCH();
      hipLaunchKernelGGL(k_zero_pad_rows, dim3(blocks_for((size_t)rowsTB * (Np - N) * 2 * H)), dim3(256), 0, s, tr + R.oDPG2[q],
                         rowsTB, N, Np, 2 * H);
      CHECK_LAUNCH();
    }
  }
  auto zero_grad = [&](float* p, long n) { return p ? zero_async(p, n, s) : MATGCN_OK; };
  RETURN_IF(zero_grad(g->node_emb, (long)N * P.d));
  RETURN_IF(zero_grad(g->weights_gru, (long)P.L * T));
  if ((!P.gcnOff && !g->weights_gru) || !g->weight_tsg || !g->end_conv_weight || !g->end_conv_bias)
    return MATGCN_ERR_NULL;

  return MATGCN_OK;
}

// plain copies the backward GEMMs contract with: the support stack and the folded node-adaptive weights.  Parameter-only
// work: matgcn_forward_train runs it on a side stream beside the forward (it used to open every backward: 0.8 ms on the
// critical path), matgcn_backward finds the copies in the train buffer.
int plain_operands(const Ctx& c, float* tr, hipStream_t s) {
  const Plan& P = c.P;
  const TrainPlan& R = c.R;
  const StackMap map = build_stack_map(P, c.D, c.prm);
  if (P.Ks > 0) {  // plain copy of the support stack for the transposed mixes
    hipLaunchKernelGGL(k_stack_plain, dim3((unsigned)((P.Ks * P.Np + 31) / 32), (unsigned)((P.NpC + 31) / 32)), dim3(256),
                       0, s, c.prep + P.oSt, P.Mp, P.N, P.Ks * P.Np, P.NpC, tr + R.oStP);
    CHECK_LAUNCH();
  }
  // plain folded weights of both AGCNs of every layer
  for (int l = 0; l < P.L && !P.gcnOff; ++l)
    for (int part = 0; part < 2; ++part) {
      const matgcn_agcn_params& ap = part == 0 ? c.prm->gate[l] : c.prm->update[l];
      PlainPrep q;
      memset(&q, 0, sizeof(q));
      q.E = c.prm->node_emb; q.wpool = ap.weights_pool; q.wg = c.D->scale_by_g ? ap.weights_g : nullptr;
      q.out = tr + R.oWp[l][part];
      q.d = P.d; q.I = P.Cl[l] + H; q.O = part == 0 ? 128 : 64; q.N = P.N; q.S = R.S; q.map = map;
      const size_t perNode = (size_t)R.S * q.I * q.O;
      if (P.d <= 32 && perNode % 16 == 0) {   // the embedding contraction on the matrix cores (k_prep_mfma's, plain layout)
        const int nTiles = (P.N + 15) / 16, tilesPerBlock = 16;
        const dim3 grid((unsigned)((perNode + 127) / 128), (unsigned)((nTiles + tilesPerBlock - 1) / tilesPerBlock));
        if (P.d <= 12) hipLaunchKernelGGL(k_prep_plain_mfma<3>, grid, dim3(256), 0, s, q, tilesPerBlock);
        else if (P.d <= 20) hipLaunchKernelGGL(k_prep_plain_mfma<5>, grid, dim3(256), 0, s, q, tilesPerBlock);
        else hipLaunchKernelGGL(k_prep_plain_mfma<8>, grid, dim3(256), 0, s, q, tilesPerBlock);
      } else {
        hipLaunchKernelGGL(k_prep_plain, dim3(blocks_for(perNode), (unsigned)((P.N + PP_NODES - 1) / PP_NODES)), dim3(256),
                           0, s, q);
      }
      CHECK_LAUNCH();
    }
  // fragment-ordered transposes of the residual cell's nn.Linear weights (hidden columns): k_chain_res_node's B operands
  for (int l = 0; l < P.L && !P.gcnOff; ++l) {
    const int I = P.Cl[l] + H;
    hipLaunchKernelGGL(k_prep_linear_t16, dim3(4), dim3(256), 0, s, c.prm->res_update[l].weight, I, P.Cl[l], 64, tr + R.oRUf[l]);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(k_prep_linear_t16, dim3(8), dim3(256), 0, s, c.prm->res_gate[l].weight, I, P.Cl[l], 128, tr + R.oRGf[l]);
    CHECK_LAUNCH();
  }
  return MATGCN_OK;
}

int bwd_head(Pass& pass, const float* dOut) {
  PASS_LOCALS(pass);
  // ---- output head (MultiATGCN.py:416-418) ----
  float* dOutRows = tr + R.oDOutRows;
  hipLaunchKernelGGL(k_dout_rows, dim3(blocks_for((size_t)B * Np * P.CH)), dim3(256), 0, s, dOut, dOutRows, B,
                     P.CH / P.od, N, Np, P.od);
  CHECK_LAUNCH();
  auto pad_pow2 = [](int v) { int p2 = 1; while (p2 < v) p2 <<= 1; return p2; };
  RETURN_IF(zero_async(g->end_conv_bias, P.CH, s));
  hipLaunchKernelGGL(k_colsum_all, dim3(256), dim3(256), 0, s, dOutRows, (size_t)B, N, Np, P.CH, pad_pow2(P.CH),
                     g->end_conv_bias);
  CHECK_LAUNCH();
  const float* seqTop = b.dropMask ? tr + R.oSeqDrop : c.ws + P.oSeq[P.L - 1];   // what the head convolved
  float* dSeq = tr + R.oDSeq[0];
  {
    GemmArgs q = gemm_args(dOutRows, prm->end_conv_weight, dSeq + (size_t)tOff * slab, N, H, P.CH);
    q.sAm = P.CH; q.sAk = 1; q.bA2 = (long)Np * P.CH;
    q.sBk = (long)hT * H; q.sBn = 1; q.bB1 = H;
    q.sCm = H; q.sCn = 1; q.bC1 = slab; q.bC2 = (long)Np * H;
    q.nb2 = B;
    if (b.dropMask) {   // the dropout mask (B, hT, N, H) of the steps the head saw, applied where the gradient is produced
      q.scaleC = b.dropMask; q.bS1 = (long)N * H; q.bS2 = (long)hT * N * H; q.sSm = H; q.sSn = 1;
    }
    RETURN_IF(gemm(q, hT, s, BG_HEAD));   // (a row-per-16-lanes VALU kernel for this K = CH product was measured: 205 vs 154 us)
    // the head's weight gradient feeds nothing in this pass: with two streams it runs on the second one (behind the
    // operand preparation, joined with the last layer's weight gradients) instead of in front of the first chain
    hipStream_t hs = s;
    if (pass.twoStreams) {
      HIP_OK(hipEventRecord(g_wf.auxFork, s));          // dOutRows is there
      HIP_OK(hipStreamWaitEvent(pass.ws, g_wf.auxFork, 0));
      hs = pass.ws;
    }
    RETURN_IF(zero_async(g->end_conv_weight, (long)P.CH * hT * H, hs));
    GemmArgs w = gemm_args(dOutRows, seqTop + (size_t)tOff * slab, g->end_conv_weight, P.CH, H, N);
    w.K2 = B;
    w.sAm = 1; w.sAk = P.CH; w.sAk2 = (long)Np * P.CH;
    w.sBk = H; w.sBn = 1; w.sBk2 = (long)Np * H; w.bB1 = slab;
    w.sCm = (long)hT * H; w.sCn = 1; w.bC1 = H;
    w.mode = 1; w.split = 16;
    RETURN_IF(gemm(w, hT, hs, BG_HEAD));
  }

  return MATGCN_OK;
}

// gcn_off: a layer of dense GRU cells
int bwd_dense_layer(Pass& pass, const LayerBufs& L) {
  PASS_LOCALS(pass);
  LAYER_LOCALS(L);
  {
    // ablation: the layer is one dense GRU cell per step on (x_t, h) (MultiATGCN.py:142-150,187-192,204); its
    // nn.Linear parameters travel in the res_* fields.  z, r, hc were saved in the residual-cell slots.
    float* carry[2] = {DH, DHa};
    for (int t = T - 1; t >= 0; --t) {
      const size_t at = (size_t)t * slab;
      ChainArgs a;
      memset(&a, 0, sizeof(a));
      a.dense = 1;
      a.dseq = dSeqCur + at; a.dcarry = (t == T - 1) ? nullptr : carry[(t + 1) & 1];
      a.hprev = t > 0 ? seq + at - slab : h0;
      a.z2 = tr + R.oZ2[l] + at; a.r2 = tr + R.oR2[l] + at; a.hc2 = tr + R.oHC2[l] + at;
      a.dha = carry[t & 1]; a.dpu2 = DPU2 + at; a.dpg2 = DPG2 + 2 * at; a.dzh2 = TMP; a.dr = tr + R.oDR[par];
      a.B = B; a.N = N; a.Np = Np; a.S = S;
      const dim3 eg(blocks_for((size_t)slab));
      hipLaunchKernelGGL(k_chain_res_out, dim3(eg.x < 512 ? eg.x : 512), dim3(256), 0, s, a);
      CHECK_LAUNCH();
      GemmArgs q = gemm_args(DPU2 + at, RU + C, TMP, B * Np, H, H);
      q.sAm = H; q.sAk = 1; q.sBk = I; q.sBn = 1; q.sCm = H; q.sCn = 1;
      RETURN_IF(gemm(q, 1, s, BG_CHAIN_DENSE));
      hipLaunchKernelGGL(k_chain_res_gate, eg, dim3(256), 0, s, a);
      CHECK_LAUNCH();
      GemmArgs q2 = gemm_args(DPG2 + 2 * at, RG + C, carry[t & 1], B * Np, H, 128);
      q2.sAm = 128; q2.sAk = 1; q2.sBk = I; q2.sBn = 1; q2.sCm = H; q2.sCn = 1; q2.beta = 1.f;
      RETURN_IF(gemm(q2, 1, s, BG_CHAIN_DENSE));
    }
    const float* Xall;
    if (l == 0) {
      float* X0tm = tr + R.oX0tm;
      hipLaunchKernelGGL(k_x0_time_major, dim3(blocks_for((size_t)T * B * Np * P.C0)), dim3(256), 0, s,
                         c.ws + P.oX0p, X0tm, B, T, Np, P.C0);
      CHECK_LAUNCH();
      Xall = X0tm;
    } else {
      Xall = c.ws + P.oSeq[l - 1];
    }
    if (b.dH0) {   // the carry left by step 0 is the gradient of the layer's initial state
      hipLaunchKernelGGL(k_dh0_out, dim3(blocks_for((size_t)B * N * H)), dim3(256), 0, s, carry[0], nullptr, nullptr, 1, 0L,
                         b.dH0 + (size_t)l * B * N * H, B, N, Np, S);
      CHECK_LAUNCH();
    }
    float* Hprev = tr + R.oHprev[par]; float* Z2H = tr + R.oZ2HA[par];
    if (h0) HIP_OK(hipMemcpyAsync(Hprev, h0, (size_t)slab * sizeof(float), hipMemcpyDeviceToDevice, s));
    else RETURN_IF(zero_async(Hprev, slab, s));
    if (T > 1)
      HIP_OK(hipMemcpyAsync(Hprev + slab, seq, (size_t)(T - 1) * slab * sizeof(float), hipMemcpyDeviceToDevice, s));
    const size_t seqN = (size_t)T * slab;
    hipLaunchKernelGGL(k_mul, dim3(blocks_for(seqN)), dim3(256), 0, s, tr + R.oZ2[l], Hprev, Z2H, seqN);
    CHECK_LAUNCH();
    {
      GemmArgs q = gemm_args(DPU2, RU, dXall, rowsTB * Np, C, H);
      q.sAm = H; q.sAk = 1; q.sBk = I; q.sBn = 1; q.sCm = C; q.sCn = 1;
      RETURN_IF(gemm(q, 1, s, BG_X_NODE));
      GemmArgs q2 = gemm_args(DPG2, RG, dXall, rowsTB * Np, C, 128);
      q2.sAm = 128; q2.sAk = 1; q2.sBk = I; q2.sBn = 1; q2.sCm = C; q2.sCn = 1; q2.beta = 1.f;
      RETURN_IF(gemm(q2, 1, s, BG_X_NODE));
    }
    const matgcn_linear_grads& gg = g->res_gate[l];
    const matgcn_linear_grads& gu = g->res_update[l];
    if (!gg.weight || !gg.bias || !gu.weight || !gu.bias) return MATGCN_ERR_NULL;
    RETURN_IF(zero_async(gg.weight, 128L * I, s));
    RETURN_IF(zero_async(gu.weight, 64L * I, s));
    const long rows = (long)rowsTB * Np;
    RETURN_IF(linear_weight_grad(b, DPG2, 128, Xall, C, rows, I, 0, gg.weight));
    RETURN_IF(linear_weight_grad(b, DPG2, 128, Hprev, H, rows, I, C, gg.weight));
    RETURN_IF(linear_weight_grad(b, DPU2, 64, Xall, C, rows, I, 0, gu.weight));
    RETURN_IF(linear_weight_grad(b, DPU2, 64, Z2H, H, rows, I, C, gu.weight));
    RETURN_IF(zero_async(gg.bias, 128, s));
    RETURN_IF(zero_async(gu.bias, 64, s));
    hipLaunchKernelGGL(k_colsum_all, dim3(1024), dim3(256), 0, s, DPG2, (size_t)rowsTB, N, Np, 128, 128, gg.bias);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(k_colsum_all, dim3(1024), dim3(256), 0, s, DPU2, (size_t)rowsTB, N, Np, 64, 64, gu.bias);
    CHECK_LAUNCH();
  }
  return MATGCN_OK;
}

// x columns of both AGCNs and of the residual cell of a layer ABOVE the first, for the steps [t0, t1) its chain has
// just finished: gradient of the layer's input sequence = of the output of the layer below.  Steps 0..T-2 go into the
// gate block of the layer below at step t+1 (see mergeAbove); only the sequence's last step, which no step of the
// layer below follows, is mixed back here.  Runs on the x-column stream beside the rest of the chain (it used to follow
// the chain on the caller's stream: 2 ms of the critical path at BM / B = 64); the chain of the layer below waits chunk
// by chunk (bwd_chain).
int bwd_x_chunk(Pass& pass, const LayerBufs& L, int t0, int t1) {
  PASS_LOCALS(pass);
  LAYER_LOCALS(L);
  const Bwd& bx = pass.bx;
  hipStream_t xs = pass.xs;
  float* DAgBelow = tr + R.oDAg[(l - 1) & 1];
  const size_t last = (size_t)(T - 1) * B;
  if (t1 == T) {   // the first chunk processed: what the whole layer needs once
    if (twoStreams && l + 1 < P.L) HIP_OK(hipStreamWaitEvent(xs, g_wf.step[1][l + 1], 0));   // the block's readers are done
    RETURN_IF(zero_async(DAgBelow, slab * S, xs));                                          // step 0: nothing from above
    if (Np != N) {
      hipLaunchKernelGGL(k_zero_pad_rows, dim3(blocks_for((size_t)B * S * (Np - N) * H)), dim3(256), 0, xs, DAx, B * S, N,
                         Np, H);
      CHECK_LAUNCH();
    }
  }
  const size_t r0 = (size_t)t0 * B;
  const int rows = (t1 - t0) * B;
  RETURN_IF(zero_async(dXall + r0 * Np * C, (long)rows * Np * C, xs));
  const int tEnd = t1 < T - 1 ? t1 : T - 1;
  if (tEnd > t0)
    RETURN_IF(node_contract(bx, DPG + r0 * Np * 128, WpG, DPU + r0 * Np * 64, WpU, I, 0, (tEnd - t0) * B,
                            DAgBelow + slab * S * (t0 + 1), 0.f));
  if (t1 == T) {
    RETURN_IF(node_contract(bx, DPG + last * Np * 128, WpG, DPU + last * Np * 64, WpU, I, 0, B, DAx, 0.f));
    RETURN_IF(mix_transposed(bx, DAx, B, C, dXall + last * Np * C));
    hipLaunchKernelGGL(k_add_slot0, dim3(blocks_for((size_t)B * Np * C)), dim3(256), 0, xs, dXall + last * Np * C, DAx,
                       (size_t)B, Np, C, S);
    CHECK_LAUNCH();
  }
  if (C == H) {   // residual cell x columns: one pass over both gradient blocks of the chunk (k_res_xcol64)
    const long rr = (long)rows * Np, tiles = (rr + 31) / 32;
    hipLaunchKernelGGL(k_res_xcol64, dim3((unsigned)(tiles < 1024 ? tiles : 1024)), dim3(256), 0, xs, DPU2 + r0 * Np * H,
                       DPG2 + r0 * Np * 128, RU, RG, I, dXall + r0 * Np * C, rr);
    CHECK_LAUNCH();
  } else {  // residual cell x columns
    GemmArgs q = gemm_args(DPU2 + r0 * Np * H, RU, dXall + r0 * Np * C, rows * Np, C, H);
    q.sAm = H; q.sAk = 1; q.sBk = I; q.sBn = 1; q.sCm = C; q.sCn = 1; q.beta = 1.f;
    RETURN_IF(gemm(q, 1, xs));
    GemmArgs q2 = gemm_args(DPG2 + r0 * Np * 128, RG, dXall + r0 * Np * C, rows * Np, C, 128);
    q2.sAm = 128; q2.sAk = 1; q2.sBk = I; q2.sBn = 1; q2.sCm = C; q2.sCn = 1; q2.beta = 1.f;
    RETURN_IF(gemm(q2, 1, xs));
  }
  return MATGCN_OK;
}

// the part of a graph layer that is sequential in time
int bwd_chain(Pass& pass, const LayerBufs& L) {
  PASS_LOCALS(pass);
  LAYER_LOCALS(L);
  // ---------------- chain ----------------
  for (int t = T - 1; t >= 0; --t) {
    const size_t at = (size_t)t * slab;
    ChainArgs a;
    memset(&a, 0, sizeof(a));
    a.dseq = dSeqCur + at; a.dcarry = (t == T - 1) ? nullptr : DH; a.hprev = t > 0 ? seq + at - slab : h0;
    a.z = tr + R.oZ[l] + at; a.r = tr + R.oR[l] + at; a.hc = tr + R.oHC[l] + at;
    a.z2 = tr + R.oZ2[l] + at; a.r2 = tr + R.oR2[l] + at; a.hc2 = tr + R.oHC2[l] + at;
    a.blend = prm->weights_gru + (size_t)l * T + t; a.dblend = g->weights_gru + (size_t)l * T + t;
    a.dha = DHa; a.dpu2 = DPU2 + at; a.dpg2 = DPG2 + 2 * at; a.dpu = DPU + at; a.dpg = DPG + 2 * at;
    a.dzh2 = TMP; a.dzhA = DAu + at * S; a.dzhMix = P.Ks > 0 ? MixOut : nullptr;
    a.dhA = DAg + at * S; a.dhMix = P.Ks > 0 ? MixOut : nullptr;
    a.dh = DH; a.dr = tr + R.oDR[par];
    a.B = B; a.N = N; a.Np = Np; a.S = S;
    a.mixParts = P.Ks > 1 ? P.Ks : 1; a.mixPartStride = slab;     // the transposed mixes arrive split by slot
    const dim3 eg(blocks_for((size_t)slab));
    if (twoStreams && l + 1 < P.L) {
      // step t reads the gradient of its output - x columns of the chunk of the layer above that holds t: awaited on
      // entering the chunk from its top - and, in its gate block, what rides from step t-1 of the layer above: the
      // chunk below, awaited at the chunk's lowest step
      if (t == T - 1 || (t + 1) % pass.chunk == 0) HIP_OK(hipStreamWaitEvent(s, g_wf.bxcol[l + 1][t / pass.chunk * pass.chunk], 0));
      if (t > 0 && t % pass.chunk == 0) HIP_OK(hipStreamWaitEvent(s, g_wf.bxcol[l + 1][t - pass.chunk], 0));
    }
    ChainNodeArgs cn;
    memset(&cn, 0, sizeof(cn));
    cn.c = a; cn.I = I; cn.iOfs = C; cn.rows = B; cn.N = N; cn.Np = Np; cn.S = S;
    const dim3 ngrid((unsigned)((B + 63) / 64), (unsigned)N);
    const bool fused = CHAIN_FUSE_RES_NODE && a.mixParts <= 4;
    if (fused && !a.hprev) { a.hprev = tr + R.oZeroSlab; cn.c.hprev = a.hprev; }
    if (fused) {
      // blend + residual cell + graph-cell output algebra of step t, the carry of step t+1, and the update block's node
      // contraction dA_u = dpu . WpU^T (h columns), one workgroup per node (round 4; more than four partial mixes - more
      // than four dense stack slots - take round 3's pair of kernels below)
      ChainResNodeArgs f;
      f.c = a;
      f.c.dcarry = (t == T - 1) ? nullptr : DH;
      f.carryA = (t == T - 1) ? nullptr : DAg + (at + slab) * S;
      f.carryMix = (t == T - 1 || P.Ks <= 0) ? nullptr : MixOut;
      f.ruf = tr + R.oRUf[l]; f.rgf = tr + R.oRGf[l];
      f.Wp = WpU; f.dA = DAu + at * S; f.I = I; f.iOfs = C;
      const bool carry = t != T - 1;
      const int parts = (carry && f.carryMix) ? a.mixParts : 0;
#define CRN_LAUNCH(C_, P_) hipLaunchKernelGGL((k_chain_res_node<C_, true, P_>), ngrid, dim3(512), CRN_LDS, s, f)
      if (!carry) CRN_LAUNCH(false, 0);
      else switch (parts) {
        case 0: CRN_LAUNCH(true, 0); break; case 1: CRN_LAUNCH(true, 1); break; case 2: CRN_LAUNCH(true, 2); break;
        case 3: CRN_LAUNCH(true, 3); break; default: CRN_LAUNCH(true, 4); break;
      }
#undef CRN_LAUNCH
      CHECK_LAUNCH();
    } else {
      // blend + residual cell + graph-cell output algebra of step t, and the carry of step t+1, in one kernel
      FusedResArgs f;
      f.c = a;
      f.c.dcarry = (t == T - 1) ? nullptr : DH;
      f.carryA = (t == T - 1) ? nullptr : DAg + (at + slab) * S;
      f.carryMix = (t == T - 1 || P.Ks <= 0) ? nullptr : MixOut;
      f.ruh = RU + C; f.rgh = RG + C; f.ldW = I;
      hipLaunchKernelGGL(k_chain_res_fused<64>, dim3((unsigned)(((long)B * Np + 63) / 64)), dim3(256), fusedLds, s, f);
      CHECK_LAUNCH();
      // update block: dA_u = dpu . WpU^T (h columns), then its transposed mix
      cn.dPre = DPU + at; cn.Wp = WpU; cn.dA = DAu + at * S; cn.beta = 0.f;
      hipLaunchKernelGGL((k_chain_node<false, 64>), ngrid, dim3(512), 0, s, cn);
      CHECK_LAUNCH();
    }
    RETURN_IF(mix_transposed(b, DAu + at * S, B, H, MixOut, true, slab));
    // gate block: the gate algebra (prologue) and dA_g = dpg . WpG^T in one kernel, then its transposed mix
    // (below the top layer the block already holds the x-column gradient of the layer above for the step before:
    // same mix input h_{t-1}, so both ride the same transposed mix, carry and adjacency gradient)
    cn.dPre = nullptr; cn.Wp = WpG; cn.dA = DAg + at * S; cn.beta = mergeAbove ? 1.f : 0.f;
    if (fused) {
      const int parts = P.Ks > 0 ? a.mixParts : 0;
#define CGN_LAUNCH(P_)                                                                                     \
  do {                                                                                                     \
    if (mergeAbove) hipLaunchKernelGGL((k_chain_gate_node<P_, true>), ngrid, dim3(512), 0, s, cn);         \
    else hipLaunchKernelGGL((k_chain_gate_node<P_, false>), ngrid, dim3(512), 0, s, cn);                   \
  } while (0)
      switch (parts) {
        case 0: CGN_LAUNCH(0); break; case 1: CGN_LAUNCH(1); break; case 2: CGN_LAUNCH(2); break;
        case 3: CGN_LAUNCH(3); break; default: CGN_LAUNCH(4); break;
      }
#undef CGN_LAUNCH
    } else {
      hipLaunchKernelGGL((k_chain_node<true, 128>), ngrid, dim3(512), 0, s, cn);
    }
    CHECK_LAUNCH();
    RETURN_IF(mix_transposed(b, DAg + at * S, B, H, MixOut, true, slab));   // the carry itself is formed by the next step's kernel
    if (l > 0 && t % pass.chunk == 0) {   // a chunk of steps is through: its x columns start beside the rest of the chain
      const int t1 = t + pass.chunk < T ? t + pass.chunk : T;
      if (twoStreams) {
        HIP_OK(hipEventRecord(g_wf.bready[l][t], s));
        HIP_OK(hipStreamWaitEvent(pass.xs, g_wf.bready[l][t], 0));
      }
      RETURN_IF(bwd_x_chunk(pass, L, t, t1));
      if (twoStreams) HIP_OK(hipEventRecord(g_wf.bxcol[l][t], pass.xs));
    }
  }
  if (b.dH0) {   // what step 0 would carry into a step before it: dh + slot 0 of the gate AGCN's dA + its transposed mix
    hipLaunchKernelGGL(k_dh0_out, dim3(blocks_for((size_t)B * N * H)), dim3(256), 0, s, DH, DAg, P.Ks > 0 ? MixOut : nullptr,
                       P.Ks > 1 ? P.Ks : 1, (long)slab, b.dH0 + (size_t)l * B * N * H, B, N, Np, S);
    CHECK_LAUNCH();
  }
  return MATGCN_OK;
}

inline bool prep_hoisted(const Pass& pass, int l) { return pass.twoStreams && l >= pass.b.c.P.L - 2; }
// layer 0 with two input channels: bwd_x_columns takes the x-column blocks of the residual nn.Linear weight gradients
// along (k_res_narrow2).  It reads the time-major input, which exists that early only when the operands were prepared at
// the start of the backward (event auxDone on the second stream).
inline bool res_narrow_fused(const Pass& pass, const LayerBufs& L) { return L.narrow && L.C == 2 && prep_hoisted(pass, L.l); }

// x columns of both AGCNs and of the residual cell -> gradient of the layer's input sequence
int bwd_x_columns(Pass& pass, const LayerBufs& L) {
  PASS_LOCALS(pass);
  LAYER_LOCALS(L);
  if (l > 0) return MATGCN_OK;   // the layers above the first did theirs chunk by chunk beside the chain (bwd_x_chunk)
  // ---------------- everything that batches over the T steps ----------------
  // x columns of both AGCNs -> gradient of the input sequence of this layer
  if (narrow) {
    // node-major [s][n][rows][C]: the transposed mix is one GEMM with rows*C columns
    RETURN_IF(zero_async(DAx, (long)rowsTB * S * Np * C, s));
    bool narrowDone = true;     // the same (C0, S) set as k_wgrad_narrow; any other shape takes the generic GEMM
#define XN_LAUNCH(C0_, S_)                                                                                               \
  hipLaunchKernelGGL((k_xcol_narrow<C0_, S_>), dim3((unsigned)((rowsTB + 255) / 256), (unsigned)N), dim3(256), 0, s, DPG, \
                     DPU, WpG, WpU, DAx, rowsTB, N, Np, I)
    const int xnTpw = 6;   // 16-row tiles per wave of the matrix-core kernel
#define XNM_LAUNCH(C0_, S_)                                                                                              \
  hipLaunchKernelGGL((k_xcol_narrow_mfma<C0_, S_>),                                                                      \
                     dim3((unsigned)(((rowsTB + 15) / 16 + 4 * xnTpw - 1) / (4 * xnTpw)), (unsigned)N), dim3(256), 0, s,  \
                     DPG, DPU, WpG, WpU, DAx, rowsTB, N, Np, I, xnTpw)
    if (C == 2 && S == 4) XNM_LAUNCH(2, 4);
    else if (C == 2 && S == 5) XNM_LAUNCH(2, 5);
    else if (C == 2 && S == 2) XNM_LAUNCH(2, 2);
    else if (C == 2 && S == 1) XNM_LAUNCH(2, 1);
    else if (C == 9 && S == 4) XN_LAUNCH(9, 4);
    else narrowDone = false;
#undef XN_LAUNCH
#undef XNM_LAUNCH
    if (narrowDone) {
      CHECK_LAUNCH();
    } else {
      RETURN_IF(node_gemm_transposed(b, DPG, 128, WpG, I, 0, C, rowsTB, DAx, 0.f, true));
      RETURN_IF(node_gemm_transposed(b, DPU, 64, WpU, I, 0, C, rowsTB, DAx, 1.f, true));
    }
    const long cols = (long)rowsTB * C;
    float* MixN = tr + R.oMixN;
    if (P.Ks > 0 && cols % 64 == 0 && (long)N * cols < (1L << 29)) {
      // the graph-mix kernel on the plain stack, 64 of the rows*C columns per tile (as the forward folds x0): the generic
      // GEMM ran this 3 GFLOP product at 13 TFLOP/s - in the tail of the backward, where nothing hides it
      MixArgs a;
      a.St = tr + R.oStP; a.ldS = P.NpC;
      a.X = DAx + (size_t)Np * cols; a.xTileStride = 64; a.ldX = (int)cols;
      a.out = MixN; a.sN = cols; a.sK = 0; a.sT = 64;
      a.outFloats = (long)N * cols;
      a.Np = P.NpC; a.N = N; a.Ks = 1; a.nK = P.Ks * Np / 16; a.nColTiles = (int)(cols / 64);
      a.nRowTiles = P.NpC / 64;
      hipLaunchKernelGGL(k_mix<2>, dim3((unsigned)(a.nRowTiles * a.nColTiles), 1u), dim3(256), 0, s, a);
      CHECK_LAUNCH();
    } else if (P.Ks > 0) {
      GemmArgs q = gemm_args(c.prep + P.oSt, DAx + (size_t)Np * cols, MixN, N, (int)cols, P.Ks * Np);
      q.sAm = P.Mp; q.sAk = 1; q.sBk = cols; q.sBn = 1; q.sCm = cols; q.sCn = 1;
      RETURN_IF(gemm(q, 1, s, BG_X_MIX));
    }
    hipLaunchKernelGGL(k_narrow_gather, dim3(blocks_for((size_t)rowsTB * Np * C)), dim3(256), 0, s,
                       P.Ks > 0 ? MixN : nullptr, DAx, dXall, (size_t)rowsTB, N, Np, C);
    CHECK_LAUNCH();
  } else if (l == 0) {   // a 64-channel input layer: nothing below to ride with
    RETURN_IF(node_contract(b, DPG, WpG, DPU, WpU, I, 0, rowsTB, DAx, 0.f));
    RETURN_IF(zero_async(dXall, (long)rowsTB * Np * C, s));
    RETURN_IF(mix_transposed(b, DAx, rowsTB, C, dXall));
    hipLaunchKernelGGL(k_add_slot0, dim3(blocks_for((size_t)rowsTB * Np * C)), dim3(256), 0, s, dXall, DAx,
                       (size_t)rowsTB, Np, C, S);
    CHECK_LAUNCH();
  }
  const long rrows = (long)rowsTB * Np;
  if (res_narrow_fused(pass, L)) {
    HIP_OK(hipStreamWaitEvent(s, g_wf.auxDone, 0));     // the hoisted operand preparation (X0tm) is done
    // two input channels: the x columns AND the x-column blocks of the residual nn.Linear weight gradients in one pass over
    // the residual cell's gradients (bwd_layer_other_grads, later on this stream, finds its weight tensors cleared and those
    // blocks done: res_narrow_fused)
    const matgcn_linear_grads& gg = g->res_gate[l];
    const matgcn_linear_grads& gu = g->res_update[l];
    if (!gg.weight || !gu.weight) return MATGCN_ERR_NULL;
    RETURN_IF(zero_async(gg.weight, 128L * I, s));
    RETURN_IF(zero_async(gu.weight, 64L * I, s));
    const float* Xall = tr + R.oX0tm;
    const dim3 grid((unsigned)((rrows + 15) / 16 < 2048 ? (rrows + 15) / 16 : 2048));
    hipLaunchKernelGGL(k_res_narrow2, grid, dim3(256), 0, s, DPU2, DPG2, RU, RG, Xall, I, dXall, gu.weight, gg.weight, rrows);
    CHECK_LAUNCH();
  } else if (narrow && (C == 2 || C == 9)) {   // residual cell x columns of a narrow input: one pass over the 192 gradients per row
    const dim3 grid((unsigned)((rrows + 15) / 16 < 4096 ? (rrows + 15) / 16 : 4096));
    if (C == 2) hipLaunchKernelGGL(k_res_xcol_narrow<2>, grid, dim3(256), 0, s, DPU2, DPG2, RU, RG, I, dXall, rrows);
    else hipLaunchKernelGGL(k_res_xcol_narrow<9>, grid, dim3(256), 0, s, DPU2, DPG2, RU, RG, I, dXall, rrows);
    CHECK_LAUNCH();
  } else {  // residual cell x columns
    GemmArgs q = gemm_args(DPU2, RU, dXall, rowsTB * Np, C, H);
    q.sAm = H; q.sAk = 1; q.sBk = I; q.sBn = 1; q.sCm = C; q.sCn = 1; q.beta = 1.f;
    RETURN_IF(gemm(q, 1, s));
    GemmArgs q2 = gemm_args(DPG2, RG, dXall, rowsTB * Np, C, 128);
    q2.sAm = 128; q2.sAk = 1; q2.sBk = I; q2.sBn = 1; q2.sCm = C; q2.sCn = 1; q2.beta = 1.f;
    RETURN_IF(gemm(q2, 1, s));
  }
  return MATGCN_OK;
}

int bwd_pools_layer(Pass& pass, int l, hipStream_t onStream);

// The "other" parameter gradients of a layer - adaptive adjacency, residual nn.Linear weights and biases - on the
// stream of `bx` (inputs: Hprev / ZH / HA / Z2HA as bwd_layer_weights prepared them, DAx from bwd_x_columns).
int bwd_layer_other_grads(Pass& pass, const LayerBufs& L, const Bwd& bx, const float* Xall) {
  PASS_LOCALS(pass);
  LAYER_LOCALS(L);
  hipStream_t xs = bx.c.s;
  float* Hprev = tr + R.oHprev[par]; float* ZH = tr + R.oZH[par]; float* HA = tr + R.oHA[par];
  float* Z2HA = tr + R.oZ2HA[par];
  if (adp) {
    RETURN_IF(adaptive_grad(bx, DAg, Hprev, rowsTB, H, dT));
    RETURN_IF(adaptive_grad(bx, DAu, ZH, rowsTB, H, dT));
    if (narrow) RETURN_IF(adaptive_grad(bx, DAx, Xall, rowsTB, C, dT, true));
    else if (l == 0) RETURN_IF(adaptive_grad(bx, DAx, Xall, rowsTB, C, dT));
    else RETURN_IF(adaptive_grad(bx, DAx, Xall + (size_t)(T - 1) * slab, B, C, dT));   // the other steps ride below
  }
  // residual nn.Linear gradients (MultiATGCN.py:139-150)
  const matgcn_linear_grads& gg = g->res_gate[l];
  const matgcn_linear_grads& gu = g->res_update[l];
  if (!gg.weight || !gg.bias || !gu.weight || !gu.bias) return MATGCN_ERR_NULL;
  const bool resNarrowFused = res_narrow_fused(pass, L);   // bwd_x_columns cleared the tensors and took the x-column blocks along
  if (!resNarrowFused) {
    RETURN_IF(zero_async(gg.weight, 128L * I, xs));
    RETURN_IF(zero_async(gu.weight, 64L * I, xs));
  }
  const long rows = (long)rowsTB * Np;
  RETURN_IF(zero_async(gg.bias, 128, xs));
  RETURN_IF(zero_async(gu.bias, 64, xs));
  bool biasG = false, biasU = false;   // the h-column GEMMs (64 input channels: fast kernel) take the bias sums along
  if (resNarrowFused) {
    // (done by k_res_narrow2 in bwd_x_columns)
  } else if (narrow && (C == 2 || C == 9)) {   // both x-column blocks in one pass over the residual cell's gradients
    const dim3 grid((unsigned)((rows + 15) / 16 < 1024 ? (rows + 15) / 16 : 1024));
    if (C == 2) hipLaunchKernelGGL(k_res_wgrad_narrow<2>, grid, dim3(256), 0, xs, DPU2, DPG2, Xall, I, gu.weight, gg.weight, rows);
    else hipLaunchKernelGGL(k_res_wgrad_narrow<9>, grid, dim3(256), 0, xs, DPU2, DPG2, Xall, I, gu.weight, gg.weight, rows);
    CHECK_LAUNCH();
  } else {
    RETURN_IF(linear_weight_grad(bx, DPG2, 128, Xall, C, rows, I, 0, gg.weight));
    RETURN_IF(linear_weight_grad(bx, DPU2, 64, Xall, C, rows, I, 0, gu.weight));
  }
  RETURN_IF(linear_weight_grad(bx, DPG2, 128, HA, H, rows, I, C, gg.weight, gg.bias, &biasG));
  RETURN_IF(linear_weight_grad(bx, DPU2, 64, Z2HA, H, rows, I, C, gu.weight, gu.bias, &biasU));
  if (!biasG) {   // (the rows of the padding nodes are zero in DPG2 / DPU2: summing them with the operand stream is exact)
    hipLaunchKernelGGL(k_colsum_all, dim3(1024), dim3(256), 0, xs, DPG2, (size_t)rowsTB, N, Np, 128, 128, gg.bias);
    CHECK_LAUNCH();
  }
  if (!biasU) {
    hipLaunchKernelGGL(k_colsum_all, dim3(1024), dim3(256), 0, xs, DPU2, (size_t)rowsTB, N, Np, 64, 64, gu.bias);
    CHECK_LAUNCH();
  }
  return MATGCN_OK;
}

// Operands of a layer's weight gradients that depend on the FORWARD only: h_{t-1} as one contiguous sequence, z*h, the
// residual cell's inputs, layer 0's time-major input.  With two streams they are built at the very start of the backward,
// beside the head and the first chain (the weight-gradient stream idles there), for the two layers whose scratch sets are
// free then; deeper encoders prepare the remaining layers where they always did, right before the layer's gradients.
int bwd_prep_operands(Pass& pass, const LayerBufs& L, hipStream_t on) {
  PASS_LOCALS(pass);
  LAYER_LOCALS(L);
  if (l == 0) {
    hipLaunchKernelGGL(k_x0_time_major, dim3(blocks_for((size_t)T * B * Np * P.C0)), dim3(256), 0, on,
                       c.ws + P.oX0p, tr + R.oX0tm, B, T, Np, P.C0);
    CHECK_LAUNCH();
  }
  float* Hprev = tr + R.oHprev[par]; float* ZH = tr + R.oZH[par]; float* HA = tr + R.oHA[par];
  float* Z2HA = tr + R.oZ2HA[par];
  if (h0) HIP_OK(hipMemcpyAsync(Hprev, h0, (size_t)slab * sizeof(float), hipMemcpyDeviceToDevice, on));
  else RETURN_IF(zero_async(Hprev, slab, on));
  if (T > 1)
    HIP_OK(hipMemcpyAsync(Hprev + slab, seq, (size_t)(T - 1) * slab * sizeof(float), hipMemcpyDeviceToDevice, on));
  const size_t seqN = (size_t)T * slab;
  hipLaunchKernelGGL(k_mul, dim3(blocks_for(seqN)), dim3(256), 0, on, tr + R.oZ[l], Hprev, ZH, seqN);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(k_ha_all, dim3(blocks_for(seqN)), dim3(256), 0, on, tr + R.oR[l], Hprev, tr + R.oHC[l],
                     tr + R.oZ2[l], HA, Z2HA, seqN);
  CHECK_LAUNCH();
  return MATGCN_OK;
}

// Weight gradients of a graph layer, on the second stream when there is one (forked by the caller right after the
// layer's chain: event step[0][l]; the x columns of the layer run on the main stream meanwhile and signal mixed[0][l]).
// tailOnMain (the LAST layer processed, nothing left for the main stream to overlap with): the second stream keeps the
// node-adaptive weight gradients and the pools, the other gradients (bwd_layer_other_grads) go to the main stream.
int bwd_layer_weights(Pass& pass, const LayerBufs& L, bool tailOnMain) {
  PASS_LOCALS(pass);
  LAYER_LOCALS(L);
  if (twoStreams) HIP_OK(hipStreamWaitEvent(ws, g_wf.step[0][l], 0));
  const float* Xall = l == 0 ? tr + R.oX0tm : c.ws + P.oSeq[l - 1];
  float* Hprev = tr + R.oHprev[par]; float* ZH = tr + R.oZH[par];
  if (!prep_hoisted(pass, l)) RETURN_IF(bwd_prep_operands(pass, L, ws));
  // node-adaptive weight gradients (plain folded layout) and biases; the graph-mixed rows are the forward's
  float* dWpG = tr + R.oDWp[l][0];
  float* dWpU = tr + R.oDWp[l][1];
  // x rows of layer 0 with a dedicated narrow kernel (see below): returns false when the shape has none
  const long ldx = rup((long)rowsTB * P.C0, 64);
  auto narrow_wgrad = [&](hipStream_t on) -> bool {
#define WN_LAUNCH(C0_, S_)                                                                                               \
  hipLaunchKernelGGL((k_wgrad_narrow<C0_, S_>), dim3((unsigned)N, WN_PARTS), dim3(192 * WN_GROUPS), 0, on, Xall,          \
                     c.ws + P.oMX0, ldx, DPG, DPU, dWpG, dWpU, T, B, N, Np, I)
#define WNM_LAUNCH(C0_, S_)                                                                                              \
  hipLaunchKernelGGL((k_wgrad_narrow_mfma<C0_, S_>), dim3((unsigned)N, WNM_PARTS), dim3(256), 0, on, Xall, c.ws + P.oMX0,  \
                     ldx, DPG, DPU, dWpG, dWpU, T, B, N, Np, I)
    if (P.C0 == 2 && S == 4) WNM_LAUNCH(2, 4);
    else if (P.C0 == 2 && S == 5) WNM_LAUNCH(2, 5);
    else if (P.C0 == 2 && S == 2) WNM_LAUNCH(2, 2);
    else if (P.C0 == 2 && S == 1) WNM_LAUNCH(2, 1);
    else if (P.C0 == 9 && S == 4) WN_LAUNCH(9, 4);
    else return false;
#undef WN_LAUNCH
#undef WNM_LAUNCH
    return true;
  };
  const bool narrowShape = l == 0 && ((P.C0 == 2 && (S == 4 || S == 5 || S == 2 || S == 1)) || (P.C0 == 9 && S == 4));
  bool narrowOnMain = false;
  if (twoStreams && tailOnMain) {   // the main stream takes the other gradients once these operands exist
    HIP_OK(hipEventRecord(g_wf.xdone[0][l], ws));
    HIP_OK(hipStreamWaitEvent(s, g_wf.xdone[0][l], 0));
    // (behind that event the accumulators are cleared too.)  The narrow x-row gradients go FIRST on the main stream: the
    // weight-gradient stream is the longer of the two in the tail, and its pools need them last
    if (narrowShape) {
      narrow_wgrad(s);
      CHECK_LAUNCH();
      HIP_OK(hipEventRecord(g_wf.bfork, s));      // (bfork: recorded once at the start of the pass, free again by now)
      narrowOnMain = true;
    }
    RETURN_IF(bwd_layer_other_grads(pass, L, b, Xall));
  }
  const long gStep = (long)N * B * P.Ks * H;
  int rc = MATGCN_OK;
  // recurrent rows h_{t-1} and their mixes
  bool fastH = false;
  if (P.Ks > 0) {
    StepBlocks sb;
    if (l + 1 < P.L) {
      // the mix of h_{t-1} was written into the chunk blocks of the layer above (its x-part mix of step t-1, see
      // shared_mix_slot): one step off; the mix of the zero state at t = 0 contributes nothing, unless the forward
      // started from a state: its mix at t = 0 stayed in the workspace block G_l
      const StepBlocks above = chunk_blocks(P, tr + R.oGX[l + 1]);
      for (int t = 1; t < T; ++t) { sb.g[t] = above.g[t - 1]; sb.gNode[t] = above.gNode[t - 1]; }
      sb.g[0] = h0 ? c.ws + P.oG[l] : nullptr; sb.gNode[0] = (long)B * P.Ks * H;
    } else {
      sb = uniform_blocks(P, tr + R.oGH[l]);   // the top layer: private blocks [T][N][B][Ks][64]
    }
    fastH = node_wgrad_fast(bw, Hprev, sb, DPG, 128, I, C, dWpG, &rc, tr + R.oDBias[l][0]);   // + the gate's bias gradient
    RETURN_IF(rc);
  }
  if (fastH) {
  } else if (l + 1 < P.L && P.Ks > 0) {
    // (batched GEMMs: identity slot over all rows, dense slots chunk by chunk)
    MixedRows none = {nullptr, 0, 0, 0, 0, 0, 0};
    RETURN_IF(node_weight_grad(bw, Hprev, none, H, DPG, 128, I, C, 0, rowsTB, dWpG, 1));
    if (h0) {
      MixedRows m0 = {c.ws + P.oG[l], (long)B * P.Ks * H, H, 0, (long)P.Ks * H, 1, B};
      RETURN_IF(node_weight_grad(bw, Hprev, m0, H, DPG, 128, I, C, 0, B, dWpG, 2));
    }
    for (int t0 = 0; t0 < T;) {
      const int nt = chunk_steps(P, t0);
      const int ntm = T - 1 - t0 < nt ? T - 1 - t0 : nt;   // the block's last slot of the sequence feeds nobody here
      MixedRows mh = {tr + R.oGX[l + 1] + (size_t)t0 * gStep, (long)nt * B * P.Ks * H, H, 0, (long)P.Ks * H, 1, ntm * B};
      RETURN_IF(node_weight_grad(bw, Hprev, mh, H, DPG, 128, I, C, (long)(t0 + 1) * B, ntm * B, dWpG, 2));
      t0 += nt;
    }
  } else {
    MixedRows mh = {tr + R.oGH[l], (long)B * P.Ks * H, H, gStep, (long)P.Ks * H, T, B};
    RETURN_IF(node_weight_grad(bw, Hprev, mh, H, DPG, 128, I, C, 0, rowsTB, dWpG));
  }
  // z * h rows of the candidate's AGCN
  const bool fastZ = P.Ks > 0 && node_wgrad_fast(bw, ZH, uniform_blocks(P, tr + R.oGZH[l]), DPU, 64, I, C, dWpU, &rc,
                                                 tr + R.oDBias[l][1]);   // + the candidate's bias gradient
  if (!fastZ) {
    MixedRows mz = {tr + R.oGZH[l], (long)B * P.Ks * H, H, gStep, (long)P.Ks * H, T, B};
    RETURN_IF(node_weight_grad(bw, ZH, mz, H, DPU, 64, I, C, 0, rowsTB, dWpU));
  }
  RETURN_IF(rc);
  if (l == 0) {  // x rows of layer 0: the plain matrix of the fold, [(s, n)][ld] with column (b*T + t)*C0 + c
    // narrow x rows: M = C0 is no GEMM shape; the common widths (flow + time of day, + day of week) and stack sizes
    // (multi-graph: identity + 3 or 4 dense slots, single graph: + 1) have a dedicated kernel (k_wgrad_narrow)
    if (narrowOnMain) {
      HIP_OK(hipStreamWaitEvent(ws, g_wf.bfork, 0));   // launched on the main stream above: the pools below read them
    } else if (narrow_wgrad(ws)) {
      CHECK_LAUNCH();
    } else {
      MixedRows mx = {c.ws + P.oMX0, ldx, (long)Np * ldx, P.C0, (long)T * P.C0, T, B};
      RETURN_IF(node_weight_grad(bw, Xall, mx, C, DPG, 128, I, 0, 0, rowsTB, dWpG));
      RETURN_IF(node_weight_grad(bw, Xall, mx, C, DPU, 64, I, 0, 0, rowsTB, dWpU));
    }
  } else if (P.Ks > 0 && node_wgrad_fast(bw, Xall, chunk_blocks(P, tr + R.oGX[l]), DPG, 128, I, 0, dWpG, &rc)) {
    // x rows of deeper layers (64 channels): the chunk blocks of the forward's hoisted x part
    RETURN_IF(rc);
    if (!node_wgrad_fast(bw, Xall, chunk_blocks(P, tr + R.oGX[l]), DPU, 64, I, 0, dWpU, &rc)) return MATGCN_ERR_UNSUPPORTED;
    RETURN_IF(rc);
  } else {       // (batched GEMMs: one node-major block per x-part chunk)
    for (int t0 = 0; t0 < T;) {
      const int nt = chunk_steps(P, t0);
      MixedRows mx = {tr + R.oGX[l] + (size_t)t0 * gStep, (long)nt * B * P.Ks * H, H, 0, (long)P.Ks * H, 1, nt * B};
      RETURN_IF(node_weight_grad(bw, Xall, mx, C, DPG, 128, I, 0, (long)t0 * B, nt * B, dWpG));
      RETURN_IF(node_weight_grad(bw, Xall, mx, C, DPU, 64, I, 0, (long)t0 * B, nt * B, dWpU));
      t0 += nt;
    }
  }
  if (!fastH) {   // (k_wgrad_node summed the columns of the rows it streamed; the GEMM path needs the extra pass)
    hipLaunchKernelGGL(k_node_colsum, dim3(blocks_for((size_t)N * 128), 24), dim3(256), 0, ws, DPG, (size_t)rowsTB, N,
                       Np, 128, tr + R.oDBias[l][0]);
    CHECK_LAUNCH();
  }
  if (!fastZ) {
    hipLaunchKernelGGL(k_node_colsum, dim3(blocks_for((size_t)N * 64), 24), dim3(256), 0, ws, DPU, (size_t)rowsTB, N, Np,
                       64, tr + R.oDBias[l][1]);
    CHECK_LAUNCH();
  }
  if (!(twoStreams && tailOnMain)) {
    if (twoStreams) HIP_OK(hipStreamWaitEvent(ws, g_wf.mixed[0][l], 0));   // DAx comes from the x columns (main stream)
    RETURN_IF(bwd_layer_other_grads(pass, L, bw, Xall));
  }
  RETURN_IF(bwd_pools_layer(pass, l, ws));   // node-adaptive weight gradients of this layer -> pools, node_emb, weights_g
  if (twoStreams) HIP_OK(hipEventRecord(g_wf.step[1][l], ws));
  return MATGCN_OK;
}

int bwd_fuse_heads(Pass& pass) {
  PASS_LOCALS(pass);
  // ---- head fusion (MultiATGCN.py:365-402) ----
  {
    float* dgain = tr + R.oDGain;
    RETURN_IF(zero_async(dgain, 64, s));
    FuseBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.X = b.src ? b.src->series : b.X; a.dx0 = tr + R.oDX0; a.tsg = prm->weight_tsg; a.dgain = dgain;
    if (b.src) {
      a.labelStart = b.src->label_start; a.seriesSteps = (long)b.src->series_steps;
      for (int s2 = 0; s2 < D->x_steps; ++s2) a.rel[s2] = b.src->rel_steps[s2];
    }
    for (int h = 0; h < D->n_heads; ++h) {
      if (!g->weight_ts[h]) return MATGCN_ERR_NULL;
      a.ts[h] = prm->weight_ts[h]; a.dts[h] = g->weight_ts[h]; a.headBegin[h] = D->head_begin[h];
    }
    a.B = B; a.T = T; a.N = N; a.Np = Np; a.C0 = P.C0; a.od = P.od; a.F = D->x_feat; a.xSteps = D->x_steps;
    a.startDim = D->start_dim; a.nHeads = D->n_heads; a.nTs = D->n_ts;
    hipLaunchKernelGGL(k_fuse_heads_bwd, dim3(blocks_for((size_t)T * N * P.od), (unsigned)D->n_heads), dim3(256), 0, s,
                       a);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(k_softmax_bwd_small, dim3(1), dim3(64), 0, s, prm->weight_tsg, dgain, D->n_ts, g->weight_tsg);
    CHECK_LAUNCH();
  }

  return MATGCN_OK;
}

// parameter-only part of one layer: node-adaptive weight gradients -> pools, node_emb, weights_g (MultiATGCN.py:102-105).
// Runs on the stream of the layer's weight gradients, right behind them: the pool GEMMs of the upper layers then hide
// under the chain of the layer below instead of queueing up behind the last chain (40 small dependent launches).
int bwd_pools_layer(Pass& pass, int l, hipStream_t onStream) {
  PASS_LOCALS(pass);
  float* EK = tr + R.oEK; float* FK = tr + R.oFK; float* TmpK = tr + R.oTmpK; float* dgain = tr + R.oDPoolGain;
  const int Kt = P.KtotOrig;
  if (P.gcnOff) return MATGCN_OK;
  {
    hipStream_t s = onStream;   // shadows the pass's stream: every launch below goes behind the layer's weight gradients
    for (int part = 0; part < 2; ++part) {
      const matgcn_agcn_params& ap = part == 0 ? prm->gate[l] : prm->update[l];
      const matgcn_agcn_grads& ag = part == 0 ? g->gate[l] : g->update[l];
      if (!ag.weights_pool || !ag.bias_pool || (D->scale_by_g && !ag.weights_g)) return MATGCN_ERR_NULL;
      const int O = part == 0 ? 128 : 64, I = P.Cl[l] + H;
      const long IO = (long)I * O;
      const float* wg = D->scale_by_g ? ap.weights_g : nullptr;
      const float* dWp = tr + R.oDWp[l][part];
      const float* dBias = tr + R.oDBias[l][part];
      hipLaunchKernelGGL(k_scaled_emb, dim3(blocks_for((size_t)N * P.d), (unsigned)ent.n), dim3(256), 0, s,
                         prm->node_emb, wg, map, ent, P.d, EK, FK);
      CHECK_LAUNCH();
      RETURN_IF(zero_async(TmpK, (long)ent.n * N * P.d, s));
      RETURN_IF(zero_async(dgain, 64, s));
      // entries with distinct pool indices (every cheb_order but 1) go as TWO batched launches - 2 x ent.n dependent 25 us
      // launches were latency, not work; the batch items are not equally spaced, hence the offset tables
      bool distinct = ent.n <= 8;
      for (int e2 = 0; e2 < ent.n && distinct; ++e2)
        for (int e3 = 0; e3 < e2; ++e3) distinct = distinct && ent.pool[e3] != ent.pool[e2];
      if (distinct && ent.n > 0 && P.d <= 32 && IO % 64 == 0) {
        // both products on the matrix cores with d as two 16-wide tiles (k_pool_grad_mfma / k_pool_emb_mfma)
        hipLaunchKernelGGL(k_pool_grad_mfma, dim3((unsigned)((IO / 64 + 3) / 4), (unsigned)ent.n), dim3(256), 0, s, EK, dWp,
                           ent, N, P.d, IO, S, Kt, ag.weights_pool);
        CHECK_LAUNCH();
        const int splits = 4;
        hipLaunchKernelGGL(k_pool_emb_mfma, dim3((unsigned)(((N + 15) / 16) * splits), (unsigned)ent.n), dim3(256), 0, s,
                           dWp, ap.weights_pool, ent, N, P.d, IO, S, Kt, splits, TmpK);
        CHECK_LAUNCH();
      } else if (distinct && ent.n > 0) {
        GemmArgs q = gemm_args(EK, dWp, ag.weights_pool, P.d, (int)IO, N);
        q.sAm = 1; q.sAk = P.d; q.sBk = (long)S * IO; q.sBn = 1; q.sCm = (long)Kt * IO; q.sCn = 1;
        GemmArgs e = gemm_args(dWp, ap.weights_pool, TmpK, N, P.d, (int)IO);
        e.sAm = (long)S * IO; e.sAk = 1; e.sBk = 1; e.sBn = (long)Kt * IO; e.sCm = P.d; e.sCn = 1;
        e.mode = 1; e.split = 32;
        q.nOff = e.nOff = ent.n;
        for (int e2 = 0; e2 < ent.n; ++e2) {
          q.offA[e2] = (long)e2 * N * P.d; q.offB[e2] = (long)ent.slot[e2] * IO; q.offC[e2] = (long)ent.pool[e2] * IO;
          e.offA[e2] = (long)ent.slot[e2] * IO; e.offB[e2] = (long)ent.pool[e2] * IO; e.offC[e2] = (long)e2 * N * P.d;
        }
        RETURN_IF(gemm(q, ent.n, s, BG_POOL));
        RETURN_IF(gemm(e, ent.n, s, BG_POOL));
      }
      bool written[MATGCN_MAX_STACK] = {false};
      for (int e2 = 0; e2 < ent.n && !distinct; ++e2) {
        const int k = ent.pool[e2];
        const float* src = dWp + (size_t)ent.slot[e2] * IO;
        GemmArgs q = gemm_args(EK + (size_t)e2 * N * P.d, src, ag.weights_pool + (size_t)k * IO, P.d, (int)IO, N);
        q.sAm = 1; q.sAk = P.d; q.sBk = (long)S * IO; q.sBn = 1; q.sCm = (long)Kt * IO; q.sCn = 1;
        q.beta = written[k] ? 1.f : 0.f;   // entries that alias one pool index (cheb_order = 1) add up
        written[k] = true;
        RETURN_IF(gemm(q, 1, s, BG_POOL));
        GemmArgs e = gemm_args(src, ap.weights_pool + (size_t)k * IO, TmpK + (size_t)e2 * N * P.d, N, P.d, (int)IO);
        e.sAm = (long)S * IO; e.sAk = 1; e.sBk = 1; e.sBn = (long)Kt * IO; e.sCm = P.d; e.sCn = 1;
        e.mode = 1; e.split = 32;
        RETURN_IF(gemm(e, 1, s, BG_POOL));
      }
      hipLaunchKernelGGL(k_emb_grad, dim3(blocks_for((size_t)N * P.d), (unsigned)ent.n), dim3(256), 0, s, TmpK, FK,
                         prm->node_emb, wg, Kt, ent, N, P.d, g->node_emb, dgain);
      CHECK_LAUNCH();
      if (D->scale_by_g) {
        hipLaunchKernelGGL(k_softmax_bwd_small, dim3(1), dim3(64), 0, s, ap.weights_g, dgain, Kt, ag.weights_g);
        CHECK_LAUNCH();
      }
      if (O == 64 || O == 128) {   // bias = E . bias_pool: both gradients in one small launch
        hipLaunchKernelGGL(k_bias_pool_grad, dim3((unsigned)(P.d + blocks_for((size_t)N * P.d))), dim3(256), 0, s,
                           prm->node_emb, dBias, ap.bias_pool, N, P.d, O, ag.bias_pool, g->node_emb);
        CHECK_LAUNCH();
      } else {
        GemmArgs q = gemm_args(prm->node_emb, dBias, ag.bias_pool, P.d, O, N);
        q.sAm = 1; q.sAk = P.d; q.sBk = O; q.sBn = 1; q.sCm = O; q.sCn = 1;
        RETURN_IF(gemm(q, 1, s));
        if (g->node_emb) {
          GemmArgs e = gemm_args(dBias, ap.bias_pool, g->node_emb, N, P.d, O);
          e.sAm = O; e.sAk = 1; e.sBk = 1; e.sBn = O; e.sCm = P.d; e.sCn = 1; e.beta = 1.f;
          RETURN_IF(gemm(e, 1, s));
        }
      }
    }
  }
  return MATGCN_OK;
}

int bwd_adaptive_adjacency(Pass& pass) {
  PASS_LOCALS(pass);
  // ---- adaptive adjacency (MultiATGCN.py:80-83) ----
  if (adp) {
    // Chebyshev orders of the adaptive adjacency A: T_o = 2 A T_{o-1} - T_{o-2} (MultiATGCN.py:98-99), back to A:
    //   dA += 2 dT_o T_{o-1}^T;  dT_{o-1} += 2 A^T dT_o;  dT_{o-2} -= dT_o          (plain T_j = rows j*Np.. of StP)
    const float* StP = tr + R.oStP;
    const long NN = (long)N * N;
    for (int o = P.per; o >= 2; --o) {
      float* dTo = dT + (size_t)(o - 1) * NN;
      const float* Tprev = StP + (size_t)(o - 2) * Np * P.NpC;
      GemmArgs q = gemm_args(dTo, Tprev, dT, N, N, N);
      q.sAm = N; q.sAk = 1; q.sBk = 1; q.sBn = P.NpC; q.sCm = N; q.sCn = 1; q.alpha = 2.f; q.beta = 1.f;
      RETURN_IF(gemm(q, 1, s, BG_ADJ));
      GemmArgs e = gemm_args(StP, dTo, dT + (size_t)(o - 2) * NN, N, N, N);
      e.sAm = 1; e.sAk = P.NpC; e.sBk = N; e.sBn = 1; e.sCm = N; e.sCn = 1; e.alpha = 2.f; e.beta = 1.f;
      RETURN_IF(gemm(e, 1, s, BG_ADJ));
      if (o - 2 >= 1) {
        hipLaunchKernelGGL(k_axpy, dim3(blocks_for((size_t)NN)), dim3(256), 0, s, dT + (size_t)(o - 3) * NN, dTo, -1.f,
                           (size_t)NN);
        CHECK_LAUNCH();
      }
    }
    const bool bi = D->adp_mode == MATGCN_ADP_BI;
    const int rank = bi ? D->embed_dim : D->adj_rank;
    float* dL = tr + R.oDL;
    hipLaunchKernelGGL(k_adaptive_adj_bwd, dim3((unsigned)N), dim3(256), 0, s, bi ? prm->node_emb : prm->node_vec1,
                       bi ? nullptr : prm->node_vec2, rank, bi ? 1 : 0, N, dT, dL);
    CHECK_LAUNCH();
    if (!bi) {
      if (!g->node_vec1 || !g->node_vec2) return MATGCN_ERR_NULL;
      GemmArgs q = gemm_args(dL, prm->node_vec2, g->node_vec1, N, rank, N);
      q.sAm = N; q.sAk = 1; q.sBk = 1; q.sBn = N; q.sCm = rank; q.sCn = 1;
      RETURN_IF(gemm(q, 1, s));
      GemmArgs e = gemm_args(prm->node_vec1, dL, g->node_vec2, rank, N, N);
      e.sAm = 1; e.sAk = rank; e.sBk = N; e.sBn = 1; e.sCm = N; e.sCn = 1;
      RETURN_IF(gemm(e, 1, s));
    } else if (g->node_emb) {
      GemmArgs q = gemm_args(dL, prm->node_emb, g->node_emb, N, rank, N);
      q.sAm = N; q.sAk = 1; q.sBk = rank; q.sBn = 1; q.sCm = rank; q.sCn = 1; q.beta = 1.f;
      RETURN_IF(gemm(q, 1, s));
      q.sAm = 1; q.sAk = N;   // transposed logits gradient
      RETURN_IF(gemm(q, 1, s));
    }
  }
  return MATGCN_OK;
}

int backward_impl(Bwd& b, const float* dOut) {
  const Plan& P = b.c.P;
  const TrainPlan& R = b.c.R;
  float* tr = b.tr;
  if ((long)P.T * P.B * P.Np >= (1L << 31)) return MATGCN_ERR_UNSUPPORTED;
  Pass q;
  q.b = b;
  q.s = b.c.s;
  // the adaptive adjacency is first-order support 0 and never diagonal: it is dense slot 0 (node-GEMM slot 1)
  q.adp = b.c.D->adp_mode != MATGCN_ADP_NONE && !P.gcnOff;
  q.hT = P.headT; q.tOff = P.T - P.headT;
  q.map = build_stack_map(P, b.c.D, b.c.prm);
  q.ent = stack_entries(q.map);
  // the weight gradients of a layer (big GEMMs) run on a library stream while the caller's stream already walks the
  // chain of the layer below (small dependent launches); matgcn_set_wavefront(0) keeps everything on one stream
  RETURN_IF(wavefront_ready());
  q.fusedLds = 128 * CF_LD * (int)sizeof(float);
  {
    static bool optedIn[MAX_DEVICES] = {false};   // dynamic LDS above 64 KB must be opted into once per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    if (!optedIn[dev]) {
      HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chain_res_fused<64>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, q.fusedLds));
      optedIn[dev] = true;
    }
  }
  q.twoStreams = g_wavefront_mode != 0 && P.L > 1 && !P.gcnOff;
  q.ws = q.twoStreams ? g_wf.chain[1] : q.s;
  q.bw = b;
  q.bw.c.s = q.ws;
  q.xs = q.twoStreams ? g_wf.xcol : q.s;
  q.bx = b;
  q.bx.c.s = q.xs;
#ifndef BX_CHUNKS
#define BX_CHUNKS 4   // x-column chunks per sequence (lab switch; 2 / 8 / 12 measured: 16.3-16.4 / 16.4 / 16.2 ms against 16.1)
#endif
  q.chunk = P.T >= 8 ? (P.T + BX_CHUNKS - 1) / BX_CHUNKS : P.T;

  LayerBufs LB[MATGCN_MAX_LAYERS];
  int cur = 0;   // which of the two sequence-gradient buffers holds the gradient of the current layer's output
  for (int l = P.L - 1; l >= 0; --l) {
    LayerBufs& L = LB[l];
    L.l = l; L.C = P.Cl[l]; L.I = L.C + H; L.par = P.L > 1 ? (l & 1) : 0;
    L.seq = b.c.ws + P.oSeq[l];
    L.h0 = b.hasH0 ? tr + R.oH0 + (size_t)l * P.B * P.Np * H : nullptr;
    L.dSeqCur = tr + R.oDSeq[cur];
    L.dXall = (l == 0) ? tr + R.oDX0 : tr + R.oDSeq[cur ^ 1];
    L.DPU = tr + R.oDPU[L.par]; L.DPG = tr + R.oDPG[L.par]; L.DPU2 = tr + R.oDPU2[L.par]; L.DPG2 = tr + R.oDPG2[L.par];
    L.DAg = tr + R.oDAg[L.par]; L.DAu = tr + R.oDAu[L.par]; L.DAx = tr + R.oDAx[L.par];
    L.WpG = tr + R.oWp[l][0]; L.WpU = tr + R.oWp[l][1];
    L.RG = b.c.prm->res_gate[l].weight;     // (128, I)
    L.RU = b.c.prm->res_update[l].weight;   // (64, I)
    // x_t of layer l+1 IS h_t of layer l: the gradient of [x | mix(x)] of the layer above (steps 0..T-2) was written
    // into this layer's gate block of step t+1 and is back-propagated together with it
    L.mergeAbove = !P.gcnOff && l + 1 < P.L;
    L.narrow = L.C != H;   // layer 0: a handful of input channels
    if (l > 0) cur ^= 1;
  }
  if (q.twoStreams) {   // forward-only operands of the first two layers' weight gradients: now, on the idle second stream
    HIP_OK(hipEventRecord(g_wf.fork, q.s));
    HIP_OK(hipStreamWaitEvent(q.ws, g_wf.fork, 0));
    for (int l = P.L - 1; l >= 0; --l)
      if (prep_hoisted(q, l)) RETURN_IF(bwd_prep_operands(q, LB[l], q.ws));
    HIP_OK(hipEventRecord(g_wf.auxDone, q.ws));   // (k_res_narrow2 on the chain's stream reads the time-major input)
  }
  RETURN_IF(bwd_clear(q));
  RETURN_IF(bwd_head(q, dOut));
  // A wavefront over the layers, as in the forward: step t of layer l needs the x-column gradient of layer l+1 for the
  // chunk of steps that holds t (and t-1) only - not its whole chain.  Every second layer therefore walks its chain (and
  // everything else that used to sit on the caller's stream for it) on a library stream, beside the chain of the layer
  // above; the chunk events of bwd_x_chunk are the only ties between the two.  The chains' scratch exists twice.
  hipStream_t callerS = q.s;
  const bool layerWave = q.twoStreams && P.L > 1;
  if (layerWave) {
    HIP_OK(hipEventRecord(g_wf.bfork, callerS));                   // scratch cleared, head done
    HIP_OK(hipStreamWaitEvent(g_wf.bchain, g_wf.bfork, 0));
  }
  for (int l = P.L - 1; l >= 0; --l) {
    const LayerBufs& L = LB[l];
    q.s = (layerWave && ((P.L - 1 - l) & 1)) ? g_wf.bchain : callerS;
    q.b.c.s = q.s;
    if (q.twoStreams && l + 2 < P.L) HIP_OK(hipStreamWaitEvent(q.s, g_wf.step[1][l + 2], 0));   // scratch set is free again
    if (P.gcnOff) {
      RETURN_IF(bwd_dense_layer(q, L));
    } else {
      RETURN_IF(bwd_chain(q, L));
      if (q.twoStreams) HIP_OK(hipEventRecord(g_wf.step[0][l], q.s));    // the weight-gradient stream forks here
      RETURN_IF(bwd_x_columns(q, L));
      if (l == 0) RETURN_IF(bwd_fuse_heads(q));   // dX0 is complete on this stream: the head-fusion gradients need nothing else
      if (q.twoStreams) {   // DAx is complete (layers above the first: with the top chunk, on the x-column stream)
        if (l > 0) HIP_OK(hipStreamWaitEvent(q.s, g_wf.bxcol[l][(P.T - 1) / q.chunk * q.chunk], 0));
        HIP_OK(hipEventRecord(g_wf.mixed[0][l], q.s));
      }
      RETURN_IF(bwd_layer_weights(q, L, l == 0));
    }
  }
  q.s = callerS;
  q.b.c.s = callerS;
  if (layerWave) {
    HIP_OK(hipEventRecord(g_wf.bjoin, g_wf.bchain));
    HIP_OK(hipStreamWaitEvent(callerS, g_wf.bjoin, 0));
  }
  if (q.twoStreams)
    for (int l = 0; l < P.L; ++l) HIP_OK(hipStreamWaitEvent(q.s, g_wf.step[1][l], 0));   // join
  if (P.gcnOff) RETURN_IF(bwd_fuse_heads(q));   // (graph layers: done right behind layer 0's x columns, see above)
  return bwd_adaptive_adjacency(q);
}

}  // namespace

extern "C" {

int matgcn_train_bytes(const matgcn_dims* dims, size_t* bytes) {
  if (!bytes) return MATGCN_ERR_NULL;
  Plan P;
  RETURN_IF(make_plan(dims, &P));
  TrainPlan R;
  RETURN_IF(make_train_plan(P, &R));
  *bytes = (size_t)R.floats * sizeof(float);
  return MATGCN_OK;
}

static int forward_train_impl(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* X,
                              const matgcn_series* src, const float* h0, const float* drop_mask, float* out,
                              void* workspace, size_t workspace_bytes, void* train, size_t train_bytes, void* stream) {
  if (!prepared || (!X && !src) || !out || !train) return MATGCN_ERR_NULL;
  if (src) RETURN_IF(check_series(dims, src->series, src->series_steps, src->label_start, src->rel_steps));
  Ctx c;
  // (a lazy matgcn_prepare: the encoder's chains wait for the weight streams they read, as in the inference forward;
  // what the side stream below reads of `prepared` - the support stack - was written on the caller's stream)
  RETURN_IF(make_ctx(&c, dims, params, prepared, workspace, workspace_bytes, stream, false));
  if (!params->weight_tsg || !params->end_conv_bias) return MATGCN_ERR_NULL;
  for (int h = 0; h < dims->n_heads; ++h) if (!params->weight_ts[h]) return MATGCN_ERR_NULL;
  RETURN_IF(check_layer_params(dims, params));
  const Plan& P = c.P;
  RETURN_IF(make_train_plan(P, &c.R));
  if (train_bytes < (size_t)c.R.floats * sizeof(float)) return MATGCN_ERR_SMALL_BUFFER;
  c.train = (float*)train;
  // the forward kernels write the rows of the real nodes only: the rows of the padding nodes must read as zero.  The
  // saved tensors are contiguous [T][B][Np][64] blocks, so one launch clears the padding rows of all of them.
  if (P.Np != P.N) {
    const size_t rows = (size_t)(c.R.savedFloats / ((long)P.Np * H));
    hipLaunchKernelGGL(k_zero_pad_rows, dim3(blocks_for(rows * (P.Np - P.N) * H)), dim3(256), 0, c.s, c.train,
                       (int)rows, P.N, P.Np, H);
    CHECK_LAUNCH();
  }
  // parameter-only operands of the backward (plain support stack, plain folded weights): on a side stream beside the
  // forward (joined into the caller's stream before this call returns its last kernel)
  RETURN_IF(wavefront_ready());
  const bool side = g_wavefront_mode != 0;
  hipStream_t aux = side ? g_wf.aux : c.s;
  if (side) {
    HIP_OK(hipEventRecord(g_wf.auxFork, c.s));
    HIP_OK(hipStreamWaitEvent(aux, g_wf.auxFork, 0));
  }
  RETURN_IF(plain_operands(c, c.train, aux));
  if (side) HIP_OK(hipEventRecord(g_wf.auxDone, aux));
  float* x0p = c.ws + P.oX0p;
  if (src) RETURN_IF(fuse_padded(c, src->series, x0p, src->label_start, src->rel_steps, src->series_steps));
  else RETURN_IF(fuse_padded(c, X, x0p));
  // graph layers: the top layer's update kernel writes the dropped-out sequence beside the plain one (a pass of its own
  // over 3 x 163 MB used to sit between the encoder and the head, 0.14 ms that nothing hides)
  const bool fusedDrop = drop_mask != nullptr && !P.gcnOff;
  if (fusedDrop) c.dropMask = drop_mask;
  RETURN_IF(encoder_padded(c, x0p, h0, nullptr));
  if (side) HIP_OK(hipStreamWaitEvent(c.s, g_wf.auxDone, 0));
  const float* seqTop = c.ws + P.oSeq[P.L - 1];
  if (fusedDrop) {
    seqTop = c.train + c.R.oSeqDrop;
  } else if (drop_mask) {   // mask (B, headT, N, H) on the steps the head convolves (fnn_off: the last one)
    const size_t ofs = (size_t)(P.T - P.headT) * P.B * P.Np * H;
    float* dropped = c.train + c.R.oSeqDrop;
    hipLaunchKernelGGL(k_apply_mask, dim3(blocks_for((size_t)P.headT * P.B * P.Np * H)), dim3(256), 0, c.s, seqTop + ofs,
                       drop_mask, dropped + ofs, P.B, P.headT, P.N, P.Np);
    CHECK_LAUNCH();
    seqTop = dropped;
  }
  return head_padded(c, seqTop, out);
}

// a failure between a fork onto the library streams and their join (side stream of the plain operands, the layers'
// chains) joins them into the caller's stream before the error code is returned (join_library_streams)
static int forward_train_entry(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* X,
                               const matgcn_series* src, const float* h0, const float* drop_mask, float* out, void* workspace,
                               size_t workspace_bytes, void* train, size_t train_bytes, void* stream) {
  JOINED(forward_train_impl(dims, params, prepared, X, src, h0, drop_mask, out, workspace, workspace_bytes, train,
                            train_bytes, stream), stream);
}

int matgcn_forward_train(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* X,
                         const matgcn_series* src, const float* h0, const float* drop_mask, float* out, void* workspace,
                         size_t workspace_bytes, void* train, size_t train_bytes, void* stream) {
  return on_main_stream(stream, [&](void* s) {
    return forward_train_entry(dims, params, prepared, X, src, h0, drop_mask, out, workspace, workspace_bytes, train,
                               train_bytes, s);
  });
}

static int backward_entry(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* X,
                          const matgcn_series* src, const float* h0, const float* drop_mask, const float* d_out,
                          const matgcn_grads* grads, float* d_h0, void* workspace, size_t workspace_bytes, void* train,
                          size_t train_bytes, void* stream) {
  if (!prepared || (!X && !src) || !d_out || !grads || !train) return MATGCN_ERR_NULL;
  if (src) RETURN_IF(check_series(dims, src->series, src->series_steps, src->label_start, src->rel_steps));
  Bwd b;
  RETURN_IF(make_ctx(&b.c, dims, params, prepared, workspace, workspace_bytes, stream));
  RETURN_IF(check_layer_params(dims, params));
  RETURN_IF(make_train_plan(b.c.P, &b.c.R));
  if (train_bytes < (size_t)b.c.R.floats * sizeof(float)) return MATGCN_ERR_SMALL_BUFFER;
  b.X = X; b.dropMask = drop_mask; b.g = grads; b.tr = (float*)train;
  b.hasH0 = h0 != nullptr; b.dH0 = d_h0; b.src = src;
  JOINED(backward_impl(b, d_out), stream);
}

int matgcn_backward(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* X,
                    const matgcn_series* src, const float* h0, const float* drop_mask, const float* d_out,
                    const matgcn_grads* grads, float* d_h0, void* workspace, size_t workspace_bytes, void* train,
                    size_t train_bytes, void* stream) {
  return on_main_stream(stream, [&](void* s) {
    return backward_entry(dims, params, prepared, X, src, h0, drop_mask, d_out, grads, d_h0, workspace, workspace_bytes,
                          train, train_bytes, s);
  });
}

int matgcn_debug_gemm(const float* A, const float* B, float* C, const int64_t* desc, float alpha, float beta,
                      void* stream) {
  if (!A || !B || !C || !desc) return MATGCN_ERR_NULL;
  GemmArgs g = gemm_args(A, B, C, (int)desc[0], (int)desc[1], (int)desc[2]);
  g.K2 = (int)desc[3];
  g.sAm = desc[4]; g.sAk = desc[5]; g.sAk2 = desc[6];
  g.sBk = desc[7]; g.sBn = desc[8]; g.sBk2 = desc[9];
  g.sCm = desc[10]; g.sCn = desc[11];
  const int nb1 = (int)desc[12];
  g.nb2 = (int)desc[13];
  g.bA1 = desc[14]; g.bA2 = desc[15]; g.bB1 = desc[16]; g.bB2 = desc[17]; g.bC1 = desc[18]; g.bC2 = desc[19];
  g.mode = (int)desc[20]; g.split = (int)desc[21];
  g.alpha = alpha; g.beta = beta;
  if (g.nb2 < 1 || g.split < 1 || nb1 < 1 || g.mode < 0 || g.mode > 1) return MATGCN_ERR_BAD_ARG;
  return gemm(g, nb1, (hipStream_t)stream, BG_GENERIC);
}

}  // extern "C"
