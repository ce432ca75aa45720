// matgcn_capi.hip - host side of libmatgcn.so: the C ABI declared in include/matgcn.h.
//
// Pure launch orchestration: no allocation, no synchronisation, every kernel goes onto the caller's stream.
// The call sequence of one forward mirrors MultiATGCN.forward (reference MultiATGCN.py:363-420) with the
// parameter-only work hoisted into matgcn_prepare and the x-part of every graph convolution hoisted out of
// the recurrence ("schedule").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/matgcn.h"
#include "matgcn_internal.h"

// single translation unit: the kernels are compiled together with their launchers
#include "matgcn_kernels.hip"
#include "matgcn_node16.hip"
#include "matgcn_bwd_kernels.hip"

namespace {

constexpr int H = 64;
constexpr int MAX_STEPS = 64;     // in_steps the wavefront scheduler keeps events for
#ifndef X_CHUNK_STEPS
#define X_CHUNK_STEPS 2
#endif
constexpr int X_CHUNK = X_CHUNK_STEPS;   // steps per hoisted x-part chunk of layers >= 1 (round 3: 1 / 2 / 3 equal, 4 +1 %, 8 +3 %)

inline long rup(long v, long m) { return (v + m - 1) / m * m; }

// ---- derived sizes ---------------------------------------------------------------------------------
struct Plan {
  int B, N, Np, T, L, C0, Ks, Ktot, nFirst, Mp, Kx, d, CH, NTc, od, Tc;   // Ks/Ktot: DENSE slots (+ identity)
  int per, KtotOrig, nDenseFirst;
  int sumDense;               // cheb_order = 1: the dense first-order supports share one weight and are summed into one slot
  int gcnOff, headT;          // ablations: dense GRU cells instead of graph cells; head over the last step only
  int denseFirst[4];          // first-order supports that are mixed (index into [adaptive?, statics...])
  int diagFirst[4], nDiagFirst;
  int nc0, nc0p;              // layer-0 plain-matrix columns (B*T*C0) and padded to 64
  int NpC;                    // N rounded up to 64: plain N x N scratch leading dimension
  // prepared offsets (floats)
  long oSt, oPlainA, oPlainB, oPlainC;
  long oWg[MATGCN_MAX_LAYERS], oWu[MATGCN_MAX_LAYERS], oWx[MATGCN_MAX_LAYERS], oBx[MATGCN_MAX_LAYERS];
  long wgFloats[MATGCN_MAX_LAYERS], wuFloats[MATGCN_MAX_LAYERS];   // floats of the two recurrent weight streams
  long oW16g[MATGCN_MAX_LAYERS], oW16u[MATGCN_MAX_LAYERS];         // workspace: their bf16 copies (precision mode 2)
  long oW16x[MATGCN_MAX_LAYERS];                                   // and of the hoisted x part's stream (layers >= 1)
  long oRg[MATGCN_MAX_LAYERS], oRu[MATGCN_MAX_LAYERS], oHead;
  long wxStride;
  int Cl[MATGCN_MAX_LAYERS], Cpad[MATGCN_MAX_LAYERS], nGx[MATGCN_MAX_LAYERS];
  int nodeLds;                // dynamic LDS bytes of k_px16 (three 16 KB K chunks of its 64-row tile)
  int RB;                     // 64-row blocks of the batch: the unit of the fragment-ordered PX / R blocks
  long preparedFloats;
  // workspace offsets (floats); state buffers are per layer so that layers can run concurrently
  long oX0p, oX0m, oMX0, oXA0;
  long oHx[MATGCN_MAX_LAYERS], oZHx[MATGCN_MAX_LAYERS], oG[MATGCN_MAX_LAYERS], oR[MATGCN_MAX_LAYERS];
  long oSeq[MATGCN_MAX_LAYERS], oGX[MATGCN_MAX_LAYERS], oPX[MATGCN_MAX_LAYERS];
  long workspaceFloats;
  long workspaceFloatsBf16;   // with the bf16 weight-stream copies of precision mode 2 behind everything else
};

int make_plan(const matgcn_dims* D, Plan* P) {
  if (!D || !P) return MATGCN_ERR_NULL;
  if (D->batch < 1 || D->nodes < 1 || D->layers < 1 || D->layers > MATGCN_MAX_LAYERS) return MATGCN_ERR_BAD_ARG;
  if (D->hidden != H) return MATGCN_ERR_UNSUPPORTED;
  if (D->in_steps < 1 || D->x_steps < D->in_steps || D->x_feat < 1) return MATGCN_ERR_BAD_ARG;
  if (D->in_steps > MAX_STEPS) return MATGCN_ERR_UNSUPPORTED;
  if (D->out_dim < 1 || D->out_channels < 1 || D->out_channels % D->out_dim) return MATGCN_ERR_BAD_ARG;
  if (D->out_channels > 64) return MATGCN_ERR_UNSUPPORTED;
  if (D->feat_in < D->out_dim || D->feat_in - D->out_dim > MATGCN_MAX_EXT || D->feat_in > 64) return MATGCN_ERR_BAD_ARG;
  if (D->embed_dim < 1 || D->cheb_k < 1 || D->n_static < 0 || D->n_static > 3) return MATGCN_ERR_BAD_ARG;
  if (D->adp_mode < 0 || D->adp_mode > 2) return MATGCN_ERR_BAD_ARG;
  if (D->adp_mode == MATGCN_ADP_UNI && (D->adj_rank < 1 || D->adj_rank > 64)) return MATGCN_ERR_BAD_ARG;
  if (D->adp_mode == MATGCN_ADP_BI && D->embed_dim > 64) return MATGCN_ERR_BAD_ARG;
  if (D->n_heads < 1 || D->n_heads > MATGCN_MAX_HEADS || D->n_ts < D->n_heads || D->n_ts > MATGCN_MAX_HEADS)
    return MATGCN_ERR_BAD_ARG;
  for (int h = 0; h < D->n_heads; ++h)
    if (D->head_begin[h] < 0 || D->head_begin[h] + D->in_steps > D->x_steps) return MATGCN_ERR_BAD_ARG;
  for (int j = 0; j < D->feat_in - D->out_dim; ++j)
    if (D->ext_src[j] < 0 || D->ext_src[j] >= D->x_feat) return MATGCN_ERR_BAD_ARG;
  if (D->start_dim < 0 || D->start_dim + D->out_dim > D->x_feat) return MATGCN_ERR_BAD_ARG;
  memset(P, 0, sizeof(*P));
  P->B = D->batch; P->N = D->nodes; P->T = D->in_steps; P->L = D->layers; P->C0 = D->feat_in;
  P->d = D->embed_dim; P->CH = D->out_channels; P->od = D->out_dim;
  P->Np = (int)rup(P->N, 16);
  P->NpC = (int)rup(P->N, 64);
  // the write-through row stores of the node kernels (store_wt16) address a step's state slab [B][Np][64] and the
  // reset-gate block [N][B][64] with 32-bit byte offsets: keep a slab below 2^29 floats (B * Np < 8.4 M)
  if ((long)P->B * P->Np * H >= (1L << 29)) return MATGCN_ERR_UNSUPPORTED;
  const int adp = D->adp_mode != MATGCN_ADP_NONE ? 1 : 0;
  P->gcnOff = D->gcn_off ? 1 : 0;
  P->headT = D->fnn_off ? 1 : P->T;
  P->nFirst = adp + D->n_static;
  if (P->nFirst < 1 && !P->gcnOff) return MATGCN_ERR_BAD_ARG;
  // cheb_order = 1: one weight entry broadcast over [I, S_1, S_2, ..] (MultiATGCN.py:65-70,94-108; StackMap)
  P->sumDense = D->cheb_k == 1 ? 1 : 0;
  P->per = P->sumDense ? 1 : D->cheb_k - 1;
  P->KtotOrig = P->sumDense ? 1 : 1 + P->nFirst * P->per;
  if (P->KtotOrig > MATGCN_MAX_STACK) return MATGCN_ERR_UNSUPPORTED;
  if (D->diag_static_mask < 0 || D->diag_static_mask >= (1 << D->n_static)) return MATGCN_ERR_BAD_ARG;
  for (int f = 0; f < P->nFirst; ++f) {
    const bool diag = f >= adp && ((D->diag_static_mask >> (f - adp)) & 1);
    if (diag) P->diagFirst[P->nDiagFirst++] = f;
    else P->denseFirst[P->nDenseFirst++] = f;
  }
  P->Ks = P->sumDense ? (P->nDenseFirst > 0 ? 1 : 0) : P->nDenseFirst * P->per;
  if (P->gcnOff) { P->Ks = 0; P->nDenseFirst = 0; P->nDiagFirst = 0; }
  P->Ktot = P->Ks + 1;
  P->Mp = (int)rup((long)P->Ks * P->Np, 64);
  P->Kx = (int)rup((long)P->Ktot * P->C0 + 1, 16);   // folded x rows of layer 0 (+ bias row), whole k-groups
  P->NTc = (P->CH + 31) / 32;
  P->Tc = P->T < X_CHUNK ? P->T : X_CHUNK;
  P->nc0 = P->B * P->T * P->C0;
  P->nc0p = (int)rup(P->nc0, 64);
  long o = 0;
  auto take = [&](long n) { long at = o; o += rup(n, 64); return at; };
  P->oSt = take((long)P->Np * P->Mp);
  const long plainN = (D->cheb_k > 2) ? (long)P->Np * P->NpC : 0;
  // plainA also stages the adaptive adjacency (written row-wise, then transposed into its stack slot)
  P->oPlainA = take((D->cheb_k > 2 || adp) ? (long)P->Np * P->NpC : 0);
  P->oPlainB = take(plainN); P->oPlainC = take(plainN);
  for (int l = 0; l < P->L; ++l) {
    P->Cl[l] = (l == 0) ? P->C0 : H;
    P->Cpad[l] = (int)rup(P->Cl[l], 16);
    P->nGx[l] = (l == 0) ? P->Kx / 16 : 0;
    const long kt = P->gcnOff ? 0 : (long)P->Ktot * H + 16L * P->nGx[l];
    P->wgFloats[l] = (long)P->N * kt * 128; P->wuFloats[l] = (long)P->N * kt * 64;
    P->oWg[l] = take(P->wgFloats[l]);
    P->oWu[l] = take(P->wuFloats[l]);
    if (l > 0 && !P->gcnOff) {
      P->wxStride = (long)P->Ktot * H * 192;
      P->oWx[l] = take((long)P->N * P->wxStride);
      P->oBx[l] = take((long)P->N * 192);
    }
    P->oRg[l] = take((long)(P->Cpad[l] + H) * 128);
    P->oRu[l] = take((long)(P->Cpad[l] + H) * 64);
  }
  P->oHead = take((long)P->headT * H * 32 * P->NTc);
  P->preparedFloats = o;
#if PX16_PIPELINE
  P->nodeLds = 3 * 64 * 64 * (int)sizeof(float);   // the x-row chunk + two ping-pong chunks of mixed rows
#else
  P->nodeLds = (64 * 64 + 64 * 64 * (P->Ks > 1 ? P->Ks : 1)) * (int)sizeof(float);
#endif
  P->RB = (P->B + 63) / 64;
  // workspace
  o = 0;
  const long rowsBT = (long)P->B * P->T;
  P->oX0p = take(rowsBT * P->Np * P->C0);
  P->oX0m = take((long)P->Np * P->nc0p);
  P->oMX0 = take((long)P->Mp * P->nc0p);
  P->oXA0 = take((long)P->T * P->N * P->B * P->Kx);
  for (int l = 0; l < P->L; ++l) {
    P->oHx[l] = take((long)P->B * P->Np * H);
    P->oZHx[l] = take((long)P->B * P->Np * H);
    P->oG[l] = take((long)P->N * P->B * P->Ks * H);
    P->oR[l] = take((long)P->N * P->RB * NODE_R_BLOCK);
    P->oSeq[l] = take(rowsBT * P->Np * H);
    if (l > 0 && !P->gcnOff) {
      P->oGX[l] = take((long)P->T * P->N * P->B * P->Ks * H);   // every chunk keeps its own block [N][nt*B][Ks][64]
      P->oPX[l] = take((long)P->T * P->N * P->RB * NODE_PX_BLOCK);
    }
  }
  P->workspaceFloats = o;
  // optional tail, precision mode 2 only (half the bytes of the fp32 weight streams - +125 MB at N = 403,
  // several hundred MB at N = 4096 - that the fp32 product path and training never touch): matgcn_workspace_bytes counts
  // it only while mode 2 is set, and a mode-2 forward on a workspace without it returns MATGCN_ERR_SMALL_BUFFER
  for (int l = 0; l < P->L; ++l) {
    P->oW16g[l] = take((P->wgFloats[l] + 1) / 2);   // bf16: two values per float slot
    P->oW16u[l] = take((P->wuFloats[l] + 1) / 2);
    P->oW16x[l] = (l > 0 && !P->gcnOff) ? take(((long)P->N * P->wxStride + 1) / 2) : 0;
  }
  P->workspaceFloatsBf16 = o;
  return MATGCN_OK;
}

// ---- training buffer: activations saved by matgcn_forward_train + scratch of matgcn_backward -------------
struct TrainPlan {
  int S;                                   // slots the node GEMMs see: identity + dense
  // saved per layer, each [T][B][Np][64]
  long oZ[MATGCN_MAX_LAYERS], oR[MATGCN_MAX_LAYERS], oHC[MATGCN_MAX_LAYERS];
  long oZ2[MATGCN_MAX_LAYERS], oR2[MATGCN_MAX_LAYERS], oHC2[MATGCN_MAX_LAYERS];
  long oSeqDrop;                           // the top sequence after dropout (what the head saw), [T][B][Np][64]
  long oH0;                                // [L][B][Np][64] the initial state as the forward packed it (zeros without h0)
  long savedFloats;                        // [0, savedFloats) is zeroed by forward_train
  // graph-mixed rows of every step as the forward wrote them, [T][N][B][Ks][64] (x part: per chunk [N][nt*B][Ks][64],
  // which also holds the recurrent mix of the layer below; oGH exists for the top layer only);
  // written in full by forward_train, never zeroed
  long oGH[MATGCN_MAX_LAYERS], oGZH[MATGCN_MAX_LAYERS], oGX[MATGCN_MAX_LAYERS];
  long keepFloats;                         // [keepFloats, floats) is zeroed by backward
  long oWp[MATGCN_MAX_LAYERS][2], oDWp[MATGCN_MAX_LAYERS][2], oDBias[MATGCN_MAX_LAYERS][2];
  long oZeroSlab;                          // [B][Np][64] zeros: h_{-1} of a layer without a caller-supplied state (the chain
                                           // kernels read h_{t-1} unconditionally)
  long oRUf[MATGCN_MAX_LAYERS], oRGf[MATGCN_MAX_LAYERS];   // fragment-ordered transposes of the residual nn.Linear weights' hidden
                                                           // columns (B operands of k_chain_res_node; parameter-only)
  // per-layer scratch exists twice (index l & 1): the weight gradients of layer l run on a second stream while the
  // chain of layer l-1 already fills the other set
  long oDPU[2], oDPG[2], oDPU2[2], oDPG2[2];   // pre-activation gradients of every step
  long oDSeq[2];                           // gradient of a layer's output sequence (ping-pong)
  long oDAg[2], oDAu[2], oDAx[2];          // [T][B][S][Np][C] gradient of [s | mix(s)] of both AGCNs (h / x columns)
  long oDH[2], oDHa[2], oDR[2], oTmp[2], oMixOut[2];   // [B][Np][64] scratch of a chain (two sets: the chains of two
                                           // consecutive layers run side by side on two streams)
  long oX0tm, oHprev[2], oZH[2], oHA[2], oZ2HA[2], oDX0;
  long oMixN;                              // [Np][T*B*C0] transposed mix of the narrow layer-0 x columns
  long oDT, oDL, oEK, oFK, oTmpK, oDGain, oDPoolGain, oDOutRows;
  long oStP;                               // [Ks*Np][NpC] plain support stack (A operand of the transposed mix)
  long floats;
};

int make_train_plan(const Plan& P, TrainPlan* R) {
  memset(R, 0, sizeof(*R));
  R->S = P.Ktot;
  long o = 0;
  auto take = [&](long n) { long at = o; o += rup(n, 64); return at; };
  const long slab = (long)P.B * P.Np * H, seq = slab * P.T;
  for (int l = 0; l < P.L; ++l) {
    R->oZ[l] = take(seq); R->oR[l] = take(seq); R->oHC[l] = take(seq);
    R->oZ2[l] = take(seq); R->oR2[l] = take(seq); R->oHC2[l] = take(seq);
  }
  R->oSeqDrop = take(seq);
  R->oH0 = take(slab * P.L);
  R->savedFloats = o;
  const long gAll = (long)P.T * P.N * P.B * P.Ks * H;
  for (int l = 0; l < P.L; ++l) {
    // the recurrent mix of a layer below the top one lives in the chunk blocks of the layer above (shared_mix_slot)
    const bool sharedUp = l + 1 < P.L && !P.gcnOff && P.Ks > 0;
    R->oGH[l] = sharedUp ? 0 : take(gAll);
    R->oGZH[l] = take(gAll);
    if (l > 0) R->oGX[l] = take(gAll);
  }
  R->keepFloats = o;
  for (int l = 0; l < P.L; ++l)
    for (int part = 0; part < 2; ++part) {
      const long O = part == 0 ? 128 : 64, I = P.Cl[l] + H;
      R->oWp[l][part] = take((long)P.N * R->S * I * O);
      R->oDWp[l][part] = take((long)P.N * R->S * I * O);
      R->oDBias[l][part] = take((long)P.N * O);
    }
  for (int l = 0; l < P.L; ++l) { R->oRUf[l] = take(4 * 4 * 64 * 4); R->oRGf[l] = take(8 * 4 * 64 * 4); }
  R->oZeroSlab = take(slab);
  for (int q = 0; q < (P.L > 1 ? 2 : 1); ++q) {
    R->oDPU[q] = take(seq); R->oDPG[q] = take(2 * seq); R->oDPU2[q] = take(seq); R->oDPG2[q] = take(2 * seq);
    R->oDAg[q] = take(seq * R->S); R->oDAu[q] = take(seq * R->S); R->oDAx[q] = take(seq * R->S);
    R->oHprev[q] = take(seq); R->oZH[q] = take(seq); R->oHA[q] = take(seq); R->oZ2HA[q] = take(seq);
  }
  R->oDSeq[0] = take(seq); R->oDSeq[1] = take(seq);
  for (int q = 0; q < 2; ++q) {
    if (q == 1 && P.L < 2) {
      R->oDH[1] = R->oDH[0]; R->oDHa[1] = R->oDHa[0]; R->oDR[1] = R->oDR[0]; R->oTmp[1] = R->oTmp[0];
      R->oMixOut[1] = R->oMixOut[0];
      break;
    }
    R->oDH[q] = take(slab); R->oDHa[q] = take(slab); R->oDR[q] = take(slab); R->oTmp[q] = take(slab);
    R->oMixOut[q] = take(slab * (P.Ks > 1 ? P.Ks : 1));   // one partial result per dense slot (split transposed mix)
  }
  R->oX0tm = take((long)P.T * P.B * P.Np * P.C0);
  R->oDX0 = take((long)P.T * P.B * P.Np * P.C0);
  R->oMixN = take((long)P.T * P.B * P.Np * P.C0);
  R->oDT = take((long)P.per * P.N * P.N); R->oDL = take((long)P.N * P.N);   // dT: one (N,N) per Chebyshev order
  const long nEnt = P.KtotOrig + 4;   // stack entries the pool gradients walk (cheb_order = 1: up to 2 + 3 on one pool index)
  R->oEK = take(nEnt * P.N * P.d); R->oFK = take(nEnt * P.N);
  R->oTmpK = take(nEnt * P.N * P.d); R->oDGain = take(64); R->oDPoolGain = take(64);
  R->oDOutRows = take((long)P.B * P.Np * P.CH);
  R->oStP = take((long)P.Ks * P.Np * P.NpC);
  R->floats = o;
  return MATGCN_OK;
}

// ---- optional in-situ launch timing (matgcn_profile_*) ------------------------------------------------
struct Prof {
  int mask = 0, cap = 0, used = 0;
  hipEvent_t* ev = nullptr;
  int* kinds = nullptr;
};
Prof g_prof;

struct ProfScope {  // records an event pair around one launch when that kernel kind is selected
  hipStream_t s;
  int slot = -1;
  ProfScope(int kind, hipStream_t stream) : s(stream) {
    if ((g_prof.mask & kind) && g_prof.used < g_prof.cap) {
      slot = g_prof.used++;
      g_prof.kinds[slot] = kind;
      (void)hipEventRecord(g_prof.ev[2 * slot], s);
    }
  }
  ~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], s);
  }
};

inline int launch_ok() { return hipGetLastError() == hipSuccess ? MATGCN_OK : MATGCN_ERR_LAUNCH; }
// MATGCN_DEBUG_SYNC=1: synchronise after every checked launch and name the source line (fault hunting only)
inline bool debug_sync() {
  static const int on = []() { const char* e = getenv("MATGCN_DEBUG_SYNC"); return e && e[0] == '1' ? 1 : 0; }();
  return on != 0;
}
#define CHECK_LAUNCH()                                   \
  do {                                                   \
    if (hipGetLastError() != hipSuccess) return MATGCN_ERR_LAUNCH; \
    if (debug_sync()) {                                  \
      fprintf(stderr, "[matgcn] launch at line %d ...", __LINE__); fflush(stderr); \
      const hipError_t e__ = hipDeviceSynchronize();     \
      fprintf(stderr, " %s\n", hipGetErrorString(e__)); fflush(stderr); \
    }                                                    \
  } while (0)
#define RETURN_IF(x)            \
  do {                          \
    int rc__ = (x);             \
    if (rc__ != MATGCN_OK) return rc__; \
  } while (0)
#define HIP_OK(x)                                      \
  do {                                                 \
    if ((x) != hipSuccess) return MATGCN_ERR_LAUNCH;   \
  } while (0)

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

// ---- layer wavefront: one stream per recurrent chain and per x-part pre-pass ----------------------------
// Layer l+1 only needs step t of layer l to start its own step t (MultiATGCN.py:194-212 runs the layers one
// after the other, but the data dependence is the diagonal), so the chains of the layers run on separate HIP
// streams, tied together by events.  While one chain sits in a memory-bound phase (weight stream of a node
// kernel) the other can own the matrix cores (graph mix), and the tails of one kernel fill with the other's
// workgroups.  Streams and events are created once, on first use; no call creates or destroys them afterwards.
int g_stream_pool = 0;        // matgcn_set_stream_pool: 1 = library streams in a hardware-queue pool of their own
struct Wavefront {
  bool ready = false;
  hipStream_t main;                                // what the hot entry points use INSTEAD of the caller's stream (on_main_stream)
  hipEvent_t mainFork, mainJoin;
  hipStream_t chain[MATGCN_MAX_LAYERS];            // [0] unused: layer 0 runs on the caller's stream
  hipStream_t xpart[MATGCN_MAX_LAYERS];
  hipEvent_t fork, done[MATGCN_MAX_LAYERS];
  hipStream_t aux;                                 // parameter-only side work of forward_train (plain copies for the backward)
  hipStream_t xcol;                                // backward: x-column gradients of a layer, chunk by chunk beside its chain
  hipStream_t bchain;                              // backward: the chain of every second layer (beside the chain of the layer above)
  hipEvent_t bfork, bjoin;
  hipEvent_t bready[MATGCN_MAX_LAYERS][MAX_STEPS]; // backward: the chain of layer l has finished step t (a chunk's lowest)
  hipEvent_t bxcol[MATGCN_MAX_LAYERS][MAX_STEPS];  // backward: the x columns of layer l's chunk starting at step t are done
  hipEvent_t auxFork, auxDone;
  hipEvent_t step[MATGCN_MAX_LAYERS][MAX_STEPS];   // layer l finished step t
  hipEvent_t xdone[MATGCN_MAX_LAYERS][MAX_STEPS];  // x-part chunk starting at step t of layer l is in PX
  hipEvent_t mixed[MATGCN_MAX_LAYERS][MAX_STEPS];  // layer l has mixed h_{t-1} (phase 0 of its step t)
  hipEvent_t mixz[MATGCN_MAX_LAYERS][MAX_STEPS];   // layer l has mixed z*h (phase 2 of its step t): the mix token's second stop
  hipEvent_t bail[2 * MATGCN_MAX_LAYERS + 4];      // error exits: one per library stream (join_library_streams)
};
// one set per device ordinal: a HIP stream / event belongs to the device that was current when it was created, so a
// process that drives several GPUs (or the rehearsal runs that put two ranks on one box) must not share them.
// (One host thread per device at a time, like the rest of the library: the reference is single-threaded too.)
constexpr int MAX_DEVICES = 64;
// Two sets per device: the batch-split forward (matgcn_set_batch_split) runs the two halves of the batch as two
// independent forwards side by side, each with its own layer wavefront; g_wf_set says which set the code below sees.
Wavefront g_wfs[MAX_DEVICES][2];
int g_wf_set = 0;
inline Wavefront& wf_current() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
  return g_wfs[dev][g_wf_set];
}
struct SplitStreams {            // the second half's "caller stream" and the fork / join events around it
  bool ready = false;
  hipStream_t s1;
  hipEvent_t fork, done;
};
SplitStreams g_split[MAX_DEVICES];
int g_batch_split = 0;          // matgcn_set_batch_split: 0 / 1 off, 2 = two halves

// Lazy prepare (matgcn_set_lazy_prepare(1); off by default: the plain contract is "prepared is complete in stream order
// when matgcn_prepare returns").  The node-adaptive weight streams - 250 MB, most of matgcn_prepare's time - are written
// on two library streams; with the option on, matgcn_prepare does NOT join them into the caller's stream but leaves four
// events behind: the weights of layer 0 (needed by the first gate kernel, ~100 us into a forward) and of the layers
// above (needed when their chains start, later still), each on both streams.  Every entry point that takes `prepared`
// orders its stream behind all four first (make_ctx) - except the wavefront forward, which lets every chain wait for
// exactly what it reads, so the weight preparation runs beside head fusion, the layer-0 fold and the first graph mix.
struct PrepEvents {
  bool ready = false, pending = false;
  hipEvent_t l0[2], l1[2];
};
PrepEvents g_prep[MAX_DEVICES];
int g_lazy_prepare = 0;
inline PrepEvents& prep_current() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
  return g_prep[dev];
}
int prep_events_ready() {
  PrepEvents& E = prep_current();
  if (E.ready) return MATGCN_OK;
  for (int i = 0; i < 2; ++i) {
    HIP_OK(hipEventCreateWithFlags(&E.l0[i], hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&E.l1[i], hipEventDisableTiming));
  }
  E.ready = true;
  return MATGCN_OK;
}
// stream s waits for what the last lazy matgcn_prepare may still be writing: which & 1 - the weight streams of layer 0,
// which & 2 - those of the layers above.  (Waiting on an event that has completed costs nothing on the device.)
int prep_wait(hipStream_t s, int which) {
  PrepEvents& E = prep_current();
  if (!E.pending) return MATGCN_OK;
  for (int i = 0; i < 2; ++i) {
    if (which & 1) HIP_OK(hipStreamWaitEvent(s, E.l0[i], 0));
    if (which & 2) HIP_OK(hipStreamWaitEvent(s, E.l1[i], 0));
  }
  return MATGCN_OK;
}


#define g_wf (wf_current())
int g_wavefront_mode = 1;     // matgcn_set_wavefront: 0 serial, 1 free-running chains
int g_mix_precision = 0;      // matgcn_set_mix_precision: 0 fp32 operands, 1 bf16 operands for the inference graph mixes,
                              // 2 bf16 operands for the graph mixes AND the node-wise contractions (bf16 weight streams)
bool g_mix_bf16_now = false;  // set for the duration of an inference forward only (MixPrecisionScope)
bool g_node_bf16_now = false;
struct MixPrecisionScope {
  explicit MixPrecisionScope(bool inferenceForward) {
    g_mix_bf16_now = inferenceForward && g_mix_precision >= 1;
    g_node_bf16_now = inferenceForward && g_mix_precision == 2;
  }
  ~MixPrecisionScope() { g_mix_bf16_now = false; g_node_bf16_now = false; }
};

int split_ready() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
  SplitStreams& S = g_split[dev];
  if (S.ready) return MATGCN_OK;
  HIP_OK(hipStreamCreateWithFlags(&S.s1, hipStreamNonBlocking));
  HIP_OK(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
  S.ready = true;
  return MATGCN_OK;
}

int wavefront_ready() {
  if (g_wf.ready) return MATGCN_OK;
  // The runtime hands hardware queues (GPU_MAX_HW_QUEUES of them, default 4) to the streams of a process round robin in
  // creation order, from one pool per stream PRIORITY.  Which of the library's streams share a queue decides how well the
  // chains overlap (round 4, profiles/r04_rccl_queues_lab.log): two chains on one queue serialise - forward 8.0 instead of
  // 6.8 ms - while the main chain sharing one with the x-part stream is the GOOD arrangement.
  //  * default (matgcn_set_stream_pool(0)): default-priority streams in the order below, chain 0 on the caller's stream.
  //    Measured best for a single process; but the pool is shared with every other stream of the process, so an RCCL
  //    communicator created first moves everything to other queues (the 8.0 ms above).
  //  * own pool (matgcn_set_stream_pool(1), for data-parallel jobs): every library stream at the device's highest priority
  //    - a pool no other stream of the process touches - and the hot entry points on the library's `main` stream instead of
  //    the caller's (on_main_stream), so that every busy stream's queue is decided HERE: the busy streams are created at
  //    chosen places of the round robin (main 1, chain[1] 0, xpart[1] 1, aux 2, xcol 3, bchain 1 of q queues: the best of the
  //    16 xcol x bchain arrangements searched), the streams of layers 2-3 and unused ones fill the gaps.  Identical
  //    behaviour with and without a process group; 1 % slower forward than the default without one (fork / join).
  if (!g_stream_pool) {
    for (int l = 1; l < MATGCN_MAX_LAYERS; ++l) {
      HIP_OK(hipStreamCreateWithFlags(&g_wf.chain[l], hipStreamNonBlocking));
      HIP_OK(hipStreamCreateWithFlags(&g_wf.xpart[l], hipStreamNonBlocking));
    }
    HIP_OK(hipStreamCreateWithFlags(&g_wf.aux, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&g_wf.xcol, hipStreamNonBlocking));
    HIP_OK(hipStreamCreateWithFlags(&g_wf.bchain, hipStreamNonBlocking));
    g_wf.main = nullptr;
  } else {
    int prLeast = 0, prGreatest = 0;
    HIP_OK(hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest));
    auto make_stream = [&](hipStream_t* st) { return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prGreatest); };
    int q = 4;
    if (const char* e = getenv("GPU_MAX_HW_QUEUES")) { const int v = atoi(e); if (v >= 1 && v <= 64) q = v; }
#ifndef WF_RES
#define WF_RES {1, 0, 1, 2, 3, 1}
#endif
    const int want[6] = WF_RES;        // main, chain[1], xpart[1], aux, xcol, bchain
    hipStream_t* tgt[6] = {&g_wf.main, &g_wf.chain[1], &g_wf.xpart[1], &g_wf.aux, &g_wf.xcol, &g_wf.bchain};
    bool made[6] = {false, false, false, false, false, false};
    hipStream_t* fill[4] = {&g_wf.chain[2], &g_wf.xpart[2], &g_wf.chain[3], &g_wf.xpart[3]};
    int nFill = 0, left = 6;
    for (int p = 0; left > 0 && p < 512; ++p) {
      int pick = -1;
      for (int i = 0; i < 6; ++i) if (!made[i] && want[i] % q == p % q) { pick = i; break; }
      if (pick >= 0) { HIP_OK(make_stream(tgt[pick])); made[pick] = true; --left; }
      else if (nFill < 4) { HIP_OK(make_stream(fill[nFill++])); }
      else { hipStream_t unused; HIP_OK(make_stream(&unused)); }
    }
    while (nFill < 4) HIP_OK(make_stream(fill[nFill++]));
  }
  HIP_OK(hipEventCreateWithFlags(&g_wf.fork, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&g_wf.mainFork, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&g_wf.mainJoin, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&g_wf.bfork, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&g_wf.bjoin, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&g_wf.auxFork, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&g_wf.auxDone, hipEventDisableTiming));
  for (int i = 0; i < 2 * MATGCN_MAX_LAYERS + 4; ++i) HIP_OK(hipEventCreateWithFlags(&g_wf.bail[i], hipEventDisableTiming));
  for (int l = 0; l < MATGCN_MAX_LAYERS; ++l) {
    HIP_OK(hipEventCreateWithFlags(&g_wf.done[l], hipEventDisableTiming));
    for (int t = 0; t < MAX_STEPS; ++t) {
      HIP_OK(hipEventCreateWithFlags(&g_wf.step[l][t], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&g_wf.xdone[l][t], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&g_wf.mixed[l][t], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&g_wf.mixz[l][t], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&g_wf.bready[l][t], hipEventDisableTiming));
      HIP_OK(hipEventCreateWithFlags(&g_wf.bxcol[l][t], hipEventDisableTiming));
    }
  }

  g_wf.ready = true;
  return MATGCN_OK;
}

// Error exits of the entry points that fork work onto the library streams (encoder wavefront, forward_train's side
// stream, the backward's chain / weight-gradient / x-column streams): a failure between fork and join must not return to
// the caller with those streams still writing into the caller's buffers (workspace, train buffer, gradient bucket - the
// caller may free or reuse them as soon as ITS stream is idle).  Whatever is in flight on every library stream is joined
// into the caller's stream; the error code of the failed stage is what the call returns.
void join_one_set(Wavefront& W, hipStream_t caller);
void join_library_streams(hipStream_t caller) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
  for (int set = 0; set < 2; ++set) join_one_set(g_wfs[dev][set], caller);
  if (g_split[dev].ready && g_split[dev].s1 != caller &&
      hipEventRecord(g_split[dev].done, g_split[dev].s1) == hipSuccess)
    (void)hipStreamWaitEvent(caller, g_split[dev].done, 0);
  (void)hipGetLastError();
}
void join_one_set(Wavefront& W, hipStream_t caller) {
  if (!W.ready) return;
  hipStream_t all[2 * MATGCN_MAX_LAYERS + 4];
  int n = 0;
  for (int l = 1; l < MATGCN_MAX_LAYERS; ++l) { all[n++] = W.chain[l]; all[n++] = W.xpart[l]; }
  all[n++] = W.aux; all[n++] = W.xcol; all[n++] = W.bchain;
  if (W.main) all[n++] = W.main;
  for (int i = 0; i < n; ++i) {
    if (all[i] == caller) continue;
    if (hipEventRecord(W.bail[i], all[i]) == hipSuccess) (void)hipStreamWaitEvent(caller, W.bail[i], 0);
  }
  (void)hipGetLastError();
}
// runs `call`, joins the library streams into the caller's stream when it failed
#define JOINED(call, stream)                                              \
  do {                                                                    \
    const int rc_ = (call);                                               \
    if (rc_ != MATGCN_OK) join_library_streams((hipStream_t)(stream));    \
    return rc_;                                                           \
  } while (0)

// The hot entry points (forward, forward_series, forward_train, backward) run on the library's `main` stream, forked from
// the caller's stream and joined back into it before they return - success or failure - so the caller still sees one
// in-order stream - in the own-pool mode (matgcn_set_stream_pool(1)) only.  See wavefront_ready() for why.  (matgcn_set_wavefront(0), the one-stream schedule for kernel timing,
// keeps the caller's stream.)
int wavefront_ready();
extern int g_wavefront_mode;
template <class Body>
int on_main_stream(void* callerStream, Body&& body) {
  if (!g_stream_pool || g_wavefront_mode == 0) return body(callerStream);
  hipStream_t caller = (hipStream_t)callerStream;
  RETURN_IF(wavefront_ready());
  Wavefront& W = g_wf;
  HIP_OK(hipEventRecord(W.mainFork, caller));
  HIP_OK(hipStreamWaitEvent(W.main, W.mainFork, 0));
  const int rc = body((void*)W.main);
  if (hipEventRecord(W.mainJoin, W.main) == hipSuccess) (void)hipStreamWaitEvent(caller, W.mainJoin, 0);
  else (void)hipGetLastError();
  return rc;
}

#ifdef NODE_LAB_STAMPS   // lab builds only (tools/labs/stamps_r04.py): where the node kernels' in-kernel stamps go
unsigned int* g_lab_stamps = nullptr;
size_t g_lab_stamp_words = 0;
int g_lab_stamp_launch = 0;
int g_lab_stamp_kinds = 1;   // 1: node kernels (k_gate16 / k_update16), 2: k_mix<1>
extern "C" int matgcn_lab_stamps(void* buf, size_t words) {
  g_lab_stamps = static_cast<unsigned int*>(buf); g_lab_stamp_words = words; g_lab_stamp_launch = 0;
  return MATGCN_OK;
}
extern "C" int matgcn_lab_stamp_launches(void) { return g_lab_stamp_launch; }
extern "C" int matgcn_lab_stamp_kinds(int kinds) { g_lab_stamp_kinds = kinds; return MATGCN_OK; }
static unsigned int* lab_stamp_slot(unsigned gridBlocks, int wavesPerBlock = 8) {
  const size_t per = (size_t)gridBlocks * wavesPerBlock * NODE_STAMPS;
  if (!g_lab_stamps || (size_t)(g_lab_stamp_launch + 1) * per > g_lab_stamp_words) return nullptr;
  return g_lab_stamps + (size_t)(g_lab_stamp_launch++) * per;
}
#endif

// out[(k,n)][col] = sum_m S_k[n][m] X[m][col]; see k_mix
int launch_mix(const Plan& P, const float* St, const float* X, long xTileStride, int ldX, int nColTiles,
               float* out, long sN, long sK, long sT, int Ks, int rowsM, hipStream_t s, bool stepRole = false,
               long outFloats = 0) {
  if (Ks <= 0 || rowsM <= 0) return MATGCN_OK;   // every support folded away: nothing to mix
  MixArgs a;
  a.St = St; a.ldS = P.Mp; a.X = X; a.xTileStride = xTileStride; a.ldX = ldX;
  a.out = out; a.sN = sN; a.sK = sK; a.sT = sT; a.outFloats = outFloats;
  a.Np = P.Np; a.N = P.N; a.Ks = Ks; a.nK = P.Np / 16; a.nColTiles = nColTiles;
  a.nRowTiles = (int)(rup(rowsM, 64) / 64);
  ProfScope prof(stepRole ? MATGCN_PROF_MIX : MATGCN_PROF_MIX_PRE, s);
  const dim3 grid((unsigned)(a.nRowTiles * nColTiles));
#ifdef NODE_LAB_STAMPS
  if (stepRole && (g_lab_stamp_kinds & 2)) a.stamps = lab_stamp_slot(grid.x, 4);
#endif
  if (g_mix_bf16_now) {   // opt-in bf16-operand variant of the inference forward (fp32 accumulate, fp32 in / out)
    if (stepRole) hipLaunchKernelGGL(k_mix_bf16<1>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_mix_bf16<0>, grid, dim3(256), 0, s, a);
#ifndef MIX_FLUSH_MIN_NK
#define MIX_FLUSH_MIN_NK 64
#endif
  } else if (a.nK > MIX_FLUSH_MIN_NK) {   // more than 1 024 reduction indices: partial sums every 256 (k_mix's FLUSH)
    if (stepRole) hipLaunchKernelGGL((k_mix<1, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_mix<0, true>), grid, dim3(256), 0, s, a);
#ifndef MIX_C32_MAX_COLTILES
#define MIX_C32_MAX_COLTILES 16    // column tiles (batch rows of a step mix) up to which the 64 x 32-tile kernel runs; 0: never
                                   // (round 4, BM 403: B = 16 20.8 -> 18.1 us per launch, forward 3.69 -> 3.56 ms; B = 32 no gain per
                                   //  launch and the forward 2 % SLOWER; DC 237 at B = 16 unchanged: profiles/r04_small_batch_lab.log)
#endif
  } else if (nColTiles <= MIX_C32_MAX_COLTILES && xTileStride % 4 == 0 && ldX >= 64) {
    // small batches (the reference ships batch_size 16): twice the workgroups of half the width
    const dim3 g2((unsigned)(a.nRowTiles * nColTiles * 2));
    if (stepRole) hipLaunchKernelGGL(k_mix_c32<1>, g2, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_mix_c32<0>, g2, dim3(256), 0, s, a);
  } else if (stepRole) {
    hipLaunchKernelGGL(k_mix<1>, grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_mix<0>, grid, dim3(256), 0, s, a);
  }
  return launch_ok();
}

// mix of `rows` contiguous [Np][64] slabs into the node-major buffer G [N][rows][Ks][64]
// (nodeStride: floats between the nodes of G when the `rows` rows are a slice of a larger node-major block)
int mix_rows(const Plan& P, const float* St, const float* X, int rows, float* G, hipStream_t s, bool stepRole = false,
             long nodeStride = 0) {
  const long sN = nodeStride ? nodeStride : (long)rows * P.Ks * H;
  return launch_mix(P, St, X, (long)P.Np * H, H, rows, G, sN, H, (long)P.Ks * H, P.Ks, P.Ks * P.Np, s, stepRole,
                    (long)(P.N - 1) * sN + (long)rows * P.Ks * H);
}

// the stack entries the kernels see (StackMap): kept slots (identity + dense) and folded diagonal ones
StackMap build_stack_map(const Plan& P, const matgcn_dims* D, const matgcn_params* params) {
  StackMap map;
  memset(&map, 0, sizeof(map));
  map.KtotOrig = P.KtotOrig; map.N = P.N;
  map.keepK[0] = 0; map.nKeep = 1;
  const int adp = D->adp_mode != MATGCN_ADP_NONE ? 1 : 0;
  if (P.sumDense) {
    if (P.nDenseFirst > 0) map.keepK[map.nKeep++] = 0;   // the summed dense slot reads the one weight entry
  } else {
    for (int fd = 0; fd < P.nDenseFirst; ++fd)
      for (int j = 0; j < P.per; ++j) map.keepK[map.nKeep++] = 1 + P.denseFirst[fd] * P.per + j;
  }
  for (int q = 0; q < P.nDiagFirst; ++q)
    for (int j = 0; j < P.per; ++j) {
      map.diagK[map.nDiag] = P.sumDense ? 0 : 1 + P.diagFirst[q] * P.per + j;
      map.diagOrder[map.nDiag] = j + 1;
      map.diagSrc[map.nDiag] = params->static_supports + (size_t)(P.diagFirst[q] - adp) * P.N * P.N;
      ++map.nDiag;
    }
  return map;
}

struct Ctx {
  Plan P;
  const matgcn_dims* D;
  const matgcn_params* prm;
  const float* prep;
  float* ws;
  size_t wsBytes = 0;
  hipStream_t s;
  float* train = nullptr;     // matgcn_forward_train: the training buffer (activations are saved into it)
  TrainPlan R;
  const float* dropMask = nullptr;   // matgcn_forward_train: (B, headT, N, H) dropout mask of the head's input, applied by the
                                     // top layer's update kernel as it writes the sequence (graph layers)
  size_t h0LayerStride = 0;   // floats between the layers of the caller's h0 (0: B*N*H; the batch-split halves see the
                              // caller's full-batch layout)
};

// dynamic LDS: k_gate16 48 KB (state chunk + two ping-pong chunks), k_update16 64 KB (+ the x_t tile of the residual
// cell), k_px16 the whole tile; sizes above 64 KB must be opted into once per kernel
constexpr int GATE_LDS = 3 * NODE_ROWS * 64 * (int)sizeof(float);
constexpr int UPDATE_LDS = 4 * NODE_ROWS * 64 * (int)sizeof(float);
constexpr int UPDATE_SAVE_LDS = 5 * NODE_ROWS * 64 * (int)sizeof(float);   // training: + the tile saved activations pass through
int node_kernels_ready(int ldsBytes) {
  static int readyOn[MAX_DEVICES] = {0};   // function attributes are per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
  int& ready = readyOn[dev];
  if (ready >= ldsBytes) return MATGCN_OK;
  const hipFuncAttribute at = hipFuncAttributeMaxDynamicSharedMemorySize;
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gate16<false, NODE_ROWS>), at, GATE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gate16<true, NODE_ROWS>), at, GATE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<0, false, NODE_ROWS>), at, UPDATE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<1, false, NODE_ROWS>), at, UPDATE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<1, true, NODE_ROWS>), at, UPDATE_SAVE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<2, false, NODE_ROWS>), at, UPDATE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<2, true, NODE_ROWS>), at, UPDATE_SAVE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gate16<false, NODE_ROWS, true>), at, GATE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<1, false, NODE_ROWS, true>), at, UPDATE_LDS));
  // 32-row work items for batches of at most 32 rows (the halves of the batch-split forward)
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gate16<false, 32>), at, GATE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<1, false, 32>), at, UPDATE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gate16<true, 32>), at, GATE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_update16<1, true, 32>), at, UPDATE_SAVE_LDS));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px16<4, false>), at, ldsBytes));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px16<2, false>), at, ldsBytes));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px16<1, false>), at, ldsBytes));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px16<4, true>), at, ldsBytes));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px16<2, true>), at, ldsBytes));
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_px16<1, true>), at, ldsBytes));
  ready = ldsBytes;
  return MATGCN_OK;
}

// layer 0: fold the x part of all Tq steps into XA0[t][n][b][Kx] = [x | mix_k(x) | 1 | 0..]; the node kernels
// contract it with the x rows of the layer-0 weights (k-groups nG.. of the node stream).  xin: [B][Tq][Np][C0].
int fold_x0(const Ctx& c, const float* xin, int Tq, hipStream_t s) {
  const Plan& P = c.P;
  const int rows = P.B * Tq;
  const float* St = c.prep + P.oSt;
  const int ld = (int)rup((long)rows * P.C0, 64);
  float* X0m = c.ws + P.oX0m;
  float* MX0 = c.ws + P.oMX0;
  hipLaunchKernelGGL(k_x0_to_matrix, dim3(blocks_for((size_t)P.Np * ld)), dim3(256), 0, s, xin, X0m, rows, P.Np, P.C0,
                     ld);
  CHECK_LAUNCH();
  RETURN_IF(launch_mix(P, St, X0m, 64, ld, ld / 64, MX0, (long)ld, (long)P.Np * ld, 64, P.Ks, P.Ks * P.Np, s));
  hipLaunchKernelGGL(k_build_xa0, dim3(blocks_for((size_t)Tq * P.N * P.B * P.Kx)), dim3(256), 0, s, xin, MX0,
                     c.ws + P.oXA0, P.B, Tq, P.N, P.Np, P.C0, P.Ks, P.Kx, ld);
  return launch_ok();
}

// x-part chunks of layers >= 1: the first are short (1, 1, 2 steps) so that a layer starts one step behind the one below
inline int chunk_steps(const Plan& P, int t) {
  int nt = (t < 2) ? 1 : (t < 4 ? 2 : P.Tc);
  if (nt > P.Tc) nt = P.Tc;
  if (t + nt > P.T) nt = P.T - t;
  return nt;
}

// layers >= 1: hoisted x part of steps [t0, t0+nt) -> PX_l[t0..].  xin: time-major rows [nt*B][Np][64] of the
// layer below (MultiATGCN.py:106-108 restricted to the x rows, + bias)
// mixedSteps: how many of the chunk's first steps already have their mixed rows in place - written there by the layer
// below, whose recurrent mix of h_t IS the mix of this layer's input x_t (see shared_mix_slot)
int hoist_x(const Ctx& c, int l, const float* xin, int t0, int nt, hipStream_t s, int mixedSteps = 0) {
  const Plan& P = c.P;
  const int rows = P.B * nt;
  // training keeps the mixed rows of every chunk for the weight gradients
  float* GX = (c.train ? c.train + c.R.oGX[l] : c.ws + P.oGX[l]) + (size_t)t0 * P.N * P.B * P.Ks * H;
  if (mixedSteps < nt) {
    const int r0 = mixedSteps * P.B;
    RETURN_IF(mix_rows(P, c.prep + P.oSt, xin + (size_t)mixedSteps * P.B * P.Np * H, rows - r0,
                       GX + (size_t)r0 * P.Ks * H, s, false, mixedSteps ? (long)rows * P.Ks * H : 0));
  }
  Px16Args a;
  a.x = xin; a.g = GX; a.w = c.prep + P.oWx[l]; a.bias = c.prep + P.oBx[l];
  a.pxOut = c.ws + P.oPX[l] + (size_t)t0 * P.N * P.RB * NODE_PX_BLOCK;
  a.steps = nt; a.N = P.N; a.Np = P.Np; a.Ks = P.Ks; a.B = P.B;
  ProfScope prof(MATGCN_PROF_PX, s);
  const dim3 grid((unsigned)(rup(P.N, 8) * nt * P.RB));
  // precision mode 2 (inference forwards only): the bf16 copy of the stream, A rows rounded on their way into LDS
  const bool bf = g_node_bf16_now && !c.train;
  if (bf) a.w = c.ws + P.oW16x[l];
#define PX16_LAUNCH(NRT)                                                                                   \
  do {                                                                                                     \
    if (bf) hipLaunchKernelGGL((k_px16<NRT, true>), grid, dim3(512), P.nodeLds, s, a);                     \
    else hipLaunchKernelGGL((k_px16<NRT, false>), grid, dim3(512), P.nodeLds, s, a);                       \
  } while (0)
  if (P.B <= 16) PX16_LAUNCH(1);         // row tiles that hold batch rows
  else if (P.B <= 32) PX16_LAUNCH(2);
  else PX16_LAUNCH(4);
#undef PX16_LAUNCH
  return launch_ok();
}

// The recurrent mix of layer l at step t+1 (phase 0: mix of h_t) equals the x-part mix of layer l+1 at step t, so in
// layer l writes it straight into layer l+1's chunk block and reads it from there: slot of step t inside the block of
// the chunk that holds it (in the workspace, or in the training buffer, which keeps every block for the backward).
// Returns false when the mix stays private (last layer, gcn_off, nothing to mix).
bool shared_mix_slot(const Ctx& c, int l, int t, float** g, long* nodeStride) {
  const Plan& P = c.P;
  if (P.gcnOff || l + 1 >= P.L || t < 0 || t >= P.T || P.Ks <= 0) return false;
  int t0 = 0;
  while (t0 + chunk_steps(P, t0) <= t) t0 += chunk_steps(P, t0);
  const int nt = chunk_steps(P, t0);
  const size_t gStep = (size_t)P.N * P.B * P.Ks * H;
  *g = (c.train ? c.train + c.R.oGX[l + 1] : c.ws + P.oGX[l + 1]) + t0 * gStep + (size_t)(t - t0) * P.B * P.Ks * H;
  *nodeStride = (long)nt * P.B * P.Ks * H;
  return true;
}

// residual-cell operands of layer l at step t for the fused update kernel
void fill_res_args(const Ctx& c, int l, const float* xt, long xRowStride, const float* blend, float* seq_t,
                   Node16Args* a) {
  const Plan& P = c.P;
  a->xt = xt; a->xRowStride = xRowStride; a->C = P.Cl[l]; a->Cpad = P.Cpad[l];
  a->rg = c.prep + P.oRg[l]; a->rgb = c.prm->res_gate[l].bias;
  a->ru = c.prep + P.oRu[l]; a->rub = c.prm->res_update[l].bias;
  a->blend = blend; a->seq = seq_t; a->seqRowStride = (long)P.Np * H;
}


// phase 0: mix(h) -> G;  1: gate;  2: mix(z*h) -> G;  3: update [+ residual cell + blend when `res` is set]
int cell_phase(const Ctx& c, int l, int t, int phase, float* raw, const Node16Args* res, hipStream_t s) {
  const Plan& P = c.P;
  const float* St = c.prep + P.oSt;
  float* Hx = c.ws + P.oHx[l];
  float* ZHx = c.ws + P.oZHx[l];
  float* G = c.ws + P.oG[l];
  long gNodeStride = 0;
  // the mix of h_{t-1} doubles as the next layer's x-part mix of step t-1; otherwise training keeps the mixed rows of
  // every step in private per-step blocks (weight gradients of the backward)
  const bool shared = res && phase < 2 && shared_mix_slot(c, l, t - 1, &G, &gNodeStride);
  // (a layer that shares upwards has no private block for its recurrent mix: at t = 0, where nothing is shared, the
  // mix of the zero state stays in the workspace - the backward knows it contributes nothing)
  const bool sharesUp = !P.gcnOff && l + 1 < P.L && P.Ks > 0;
  if (!shared && c.train && res && !(phase < 2 && sharesUp))
    G = c.train + (phase < 2 ? c.R.oGH[l] : c.R.oGZH[l]) + (size_t)t * P.N * P.B * P.Ks * H;
  float* R = c.ws + P.oR[l];
  if (phase == 0) return mix_rows(P, St, Hx, P.B, G, s, true, gNodeStride);
  if (phase == 2) return mix_rows(P, St, ZHx, P.B, G, s, true);
  Node16Args a;
  memset(&a, 0, sizeof(a));
  a.g = G; a.gNodeStride = phase == 1 ? gNodeStride : 0; a.rows = P.B; a.N = P.N; a.Np = P.Np; a.Ks = P.Ks;
  if (l == 0) { a.xa = c.ws + P.oXA0 + (size_t)t * P.N * P.B * P.Kx; a.nGx = P.nGx[0]; }
  else a.px = c.ws + P.oPX[l] + (size_t)t * P.N * P.RB * NODE_PX_BLOCK;
  const bool save = c.train != nullptr && res != nullptr;
  // a batch of at most 32 rows (the halves of the batch-split forward) runs the 32-row instantiations: a 64-row tile
  // would be half padding
  // ... and so does a graph of at most NODE_ROWS32_MAX_NODES nodes at any batch size: 237 (node, 64-row) items leave half of
  // the chip's 512 workgroup slots empty; as (node, 32-row) items they fill 474 of them (round 3 measured the compile-time
  // variant at N = 237: gate 19.4 -> 18.5 us, update 24.4 -> 22.5 us; round 4 picks it at run time)
#ifndef NODE_ROWS32_MAX_NODES
#define NODE_ROWS32_MAX_NODES 256
#endif
  // (round 4: the training instantiations too - at the shipped batch size 16 the training forward ran 64-row tiles)
  const bool rows32 = NODE_ROWS == 64 && (P.B <= 32 || P.N <= NODE_ROWS32_MAX_NODES) && res != nullptr && !raw &&
                      !g_node_bf16_now;
  const dim3 grid(node_items(P.N, P.B, rows32 ? 32 : NODE_ROWS));   // (node, row block) work items, XCD-paired per node
#ifdef NODE_LAB_STAMPS
  a.stamps = (g_lab_stamp_kinds & 1) ? lab_stamp_slot(grid.x) : nullptr;
#endif
  if (save) {
    const size_t at = (size_t)t * P.B * P.Np * H;
    a.svZ = c.train + c.R.oZ[l] + at; a.svR = c.train + c.R.oR[l] + at; a.svHC = c.train + c.R.oHC[l] + at;
    a.svZ2 = c.train + c.R.oZ2[l] + at; a.svR2 = c.train + c.R.oR2[l] + at; a.svHC2 = c.train + c.R.oHC2[l] + at;
  }
  // precision mode 2 (inference forwards only): bf16 copies of the weight streams, made by encoder_chains
  const bool bf = g_node_bf16_now && !save && res != nullptr && !raw;
  if (phase == 1) {
    a.s = Hx; a.w = bf ? c.ws + P.oW16g[l] : c.prep + P.oWg[l]; a.zh = ZHx; a.r = R; a.raw = raw;
    ProfScope prof(MATGCN_PROF_GATE, s);
    if (rows32 && save) hipLaunchKernelGGL((k_gate16<true, 32>), grid, dim3(512), GATE_LDS / 2, s, a);
    else if (rows32) hipLaunchKernelGGL((k_gate16<false, 32>), grid, dim3(512), GATE_LDS / 2, s, a);
    else if (bf) hipLaunchKernelGGL((k_gate16<false, NODE_ROWS, true>), grid, dim3(512), GATE_LDS, s, a);
    else if (save) hipLaunchKernelGGL((k_gate16<true, NODE_ROWS>), grid, dim3(512), GATE_LDS, s, a);
    else hipLaunchKernelGGL((k_gate16<false, NODE_ROWS>), grid, dim3(512), GATE_LDS, s, a);
    return launch_ok();
  }
  a.s = ZHx; a.w = bf ? c.ws + P.oW16u[l] : c.prep + P.oWu[l]; a.r = R; a.h = Hx; a.hout = Hx;
  ProfScope prof(MATGCN_PROF_UPDATE, s);
  if (res) {
    a.xt = res->xt; a.xRowStride = res->xRowStride; a.C = res->C; a.Cpad = res->Cpad;
    a.rg = res->rg; a.rgb = res->rgb; a.ru = res->ru; a.rub = res->rub;
    a.blend = res->blend; a.seq = res->seq; a.seqRowStride = res->seqRowStride;
    if (save && c.dropMask && l == P.L - 1 && t >= P.T - P.headT && a.seq) {   // the head's dropout rides in the sequence store
      a.dropMask = c.dropMask + (size_t)(t - (P.T - P.headT)) * P.N * H;
      a.dropRowStride = (long)P.headT * P.N * H;
      a.seqDrop = c.train + c.R.oSeqDrop + (a.seq - (c.ws + P.oSeq[l]));
    }
    if (rows32 && save) hipLaunchKernelGGL((k_update16<1, true, 32>), grid, dim3(512), UPDATE_SAVE_LDS / 2, s, a);
    else if (rows32) hipLaunchKernelGGL((k_update16<1, false, 32>), grid, dim3(512), UPDATE_LDS / 2, s, a);
    else if (bf) hipLaunchKernelGGL((k_update16<1, false, NODE_ROWS, true>), grid, dim3(512), UPDATE_LDS, s, a);
    else if (save) hipLaunchKernelGGL((k_update16<1, true, NODE_ROWS>), grid, dim3(512), UPDATE_SAVE_LDS, s, a);
    else hipLaunchKernelGGL((k_update16<1, false, NODE_ROWS>), grid, dim3(512), UPDATE_LDS, s, a);
  } else {
    hipLaunchKernelGGL((k_update16<0, false, NODE_ROWS>), grid, dim3(512), UPDATE_LDS, s, a);
  }
  return launch_ok();
}

// One recurrent step of layer l at step t on the layer's state Hx_l:
//   mix(h) -> gate -> mix(z*h) -> update [+ residual GRU cell + blend when `res` is set]   (MultiATGCN.py:120-128,
//   142-150, 205-208).  raw: optional (B,N,128) dump of the gate pre-activation; gateOnly stops after the gate.
int cell_step(const Ctx& c, int l, int t, float* raw, bool gateOnly, const Node16Args* res, hipStream_t s) {
  RETURN_IF(cell_phase(c, l, t, 0, raw, res, s));
  RETURN_IF(cell_phase(c, l, t, 1, raw, res, s));
  if (gateOnly) return MATGCN_OK;
  RETURN_IF(cell_phase(c, l, t, 2, nullptr, res, s));
  return cell_phase(c, l, t, 3, nullptr, res, s);
}

// residual GRU cell alone (unit entry point): Hx_l <- cell(x_t, Hx_l)
int res_step(const Ctx& c, int l, const float* xt, long xRowStride) {
  const Plan& P = c.P;
  Node16Args a;
  memset(&a, 0, sizeof(a));
  a.s = c.ws + P.oHx[l]; a.hout = c.ws + P.oHx[l];
  a.rows = P.B; a.N = P.N; a.Np = P.Np; a.Ks = P.Ks;
  fill_res_args(c, l, xt, xRowStride, nullptr, nullptr, &a);
  ProfScope prof(MATGCN_PROF_RES, c.s);
  hipLaunchKernelGGL((k_update16<2, false, NODE_ROWS>), dim3(node_items(P.N, P.B, NODE_ROWS)), dim3(512), UPDATE_LDS, c.s, a);
  return launch_ok();
}

int zero_async(float* p, long floats, hipStream_t s) {
  return hipMemsetAsync(p, 0, (size_t)floats * sizeof(float), s) == hipSuccess ? MATGCN_OK : MATGCN_ERR_LAUNCH;
}

// the encoder over padded buffers: x0p [B][T][Np][C0] -> Seq_{L-1} (time-major [T][B][Np][64]); finalsUser
// (L,B,N,H) optional.  Layers run as a wavefront over streams (see Wavefront).
int encoder_chains(const Ctx& c, const float* x0p, const float* h0User, float* finalsUser);
// the encoder: its chains fork onto library streams; a failure between fork and join joins them before returning
int encoder_padded(const Ctx& c, const float* x0p, const float* h0User, float* finalsUser) {
  const int rc = encoder_chains(c, x0p, h0User, finalsUser);
  if (rc != MATGCN_OK) join_library_streams(c.s);
  return rc;
}
int encoder_chains(const Ctx& c, const float* x0p, const float* h0User, float* finalsUser) {
  const Plan& P = c.P;
  RETURN_IF(node_kernels_ready(P.nodeLds));
  RETURN_IF(wavefront_ready());
  Wavefront& W = g_wf;
  const bool multi = P.L > 1 && g_wavefront_mode != 0;
  const bool lazyPrep = prep_current().pending;
  // one stream, the bf16 copies (they read every stream) and the dense-GRU ablation take everything up front
  if (lazyPrep && (!multi || g_node_bf16_now || P.gcnOff)) RETURN_IF(prep_wait(c.s, 3));
  if (g_node_bf16_now && !P.gcnOff && !c.train) {
    if (c.wsBytes < (size_t)P.workspaceFloatsBf16 * sizeof(float)) return MATGCN_ERR_SMALL_BUFFER;   // sized without mode 2
    // precision mode 2: bf16 copies of the recurrent weight streams into the workspace, once per forward and in front
    // of the fork (every chain reads them); 240 MB of traffic, part of what the side line's time includes
    for (int l = 0; l < P.L; ++l) {
      const size_t og = (size_t)(P.wgFloats[l] / 8), ou = (size_t)(P.wuFloats[l] / 8);
      hipLaunchKernelGGL(k_stream_to_bf16, dim3(blocks_for(og)), dim3(256), 0, c.s, c.prep + P.oWg[l],
                         reinterpret_cast<unsigned int*>(c.ws + P.oW16g[l]), og);
      CHECK_LAUNCH();
      hipLaunchKernelGGL(k_stream_to_bf16, dim3(blocks_for(ou)), dim3(256), 0, c.s, c.prep + P.oWu[l],
                         reinterpret_cast<unsigned int*>(c.ws + P.oW16u[l]), ou);
      CHECK_LAUNCH();
      if (l > 0) {   // the hoisted x part's stream (k_px16<.., true>)
        const size_t ox = (size_t)((long)P.N * P.wxStride / 8);
        hipLaunchKernelGGL(k_stream_to_bf16, dim3(blocks_for(ox)), dim3(256), 0, c.s, c.prep + P.oWx[l],
                           reinterpret_cast<unsigned int*>(c.ws + P.oW16x[l]), ox);
        CHECK_LAUNCH();
      }
    }
  }
  if (multi) {
    HIP_OK(hipEventRecord(W.fork, c.s));
    for (int l = 1; l < P.L; ++l) {
      HIP_OK(hipStreamWaitEvent(W.chain[l], W.fork, 0));
      HIP_OK(hipStreamWaitEvent(W.xpart[l], W.fork, 0));
      if (lazyPrep) {   // the upper layers' chains and x-part streams read the upper layers' weight streams
        RETURN_IF(prep_wait(W.chain[l], 2));
        RETURN_IF(prep_wait(W.xpart[l], 2));
      }
    }
  }
  if (!P.gcnOff) RETURN_IF(fold_x0(c, x0p, P.T, c.s));
  const long stepRows = (long)P.B * P.Np * H;     // one step of a time-major sequence
  auto chain_stream = [&](int l) { return (l == 0 || !multi) ? c.s : W.chain[l]; };
  // ---- per layer: state, padding rows ----
  for (int l = 0; l < P.L; ++l) {
    hipStream_t cs = chain_stream(l);
    RETURN_IF(zero_async(c.ws + P.oZHx[l], (long)P.B * P.Np * H, cs));
    if (P.Np != P.N) {
      hipLaunchKernelGGL(k_zero_pad_rows, dim3(blocks_for((size_t)P.B * P.T * (P.Np - P.N) * H)), dim3(256), 0, cs,
                         c.ws + P.oSeq[l], P.B * P.T, P.N, P.Np, H);
      CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_pack_rows, dim3(blocks_for((size_t)P.B * P.Np * H)), dim3(256), 0, cs,
                       h0User ? h0User + (size_t)l * (c.h0LayerStride ? c.h0LayerStride : (size_t)P.B * P.N * H) : nullptr,
                       c.ws + P.oHx[l], P.B, P.N, P.Np, H);
    CHECK_LAUNCH();
    if (c.train && h0User) {   // the backward needs h_{-1} of every layer (gate algebra, weight gradients of step 0)
      hipLaunchKernelGGL(k_pack_rows, dim3(blocks_for((size_t)P.B * P.Np * H)), dim3(256), 0, cs,
                         h0User + (size_t)l * P.B * P.N * H, c.train + c.R.oH0 + (size_t)l * P.B * P.Np * H, P.B, P.N,
                         P.Np, H);
      CHECK_LAUNCH();
    }
  }
  // ---- the mix token (matgcn_set_wavefront(2), round 4) -------------------------------------------------------------
  // In the free-running wavefront every kernel of a chain is stretched by whatever the other chain happens to run beside
  // it (in-situ events: k_mix 41 -> 75 us on average, the wall is the sum of one chain's stretched kernels).  A graph mix
  // is MFMA-bound with HBM idle, a node kernel streams weights with the matrix pipe a third busy, and one node workgroup
  // fits beside the five mix workgroups of a CU (5 x 56 + 2 x 112 registers): they are complementary - two mixes, or two
  // node kernels, are not.  With the token the graph mixes of ALL chains form one global order (every mix waits for the mix
  // enqueued before it, on whatever stream that was); the node kernels stay free on their chains.  So a mix never runs
  // beside another mix, and the node kernel that follows a mix runs beside the next chain's mix.  The steps are enqueued
  // in global order (layer l runs `lag` steps behind layer l-1) so that every event is recorded before it is waited for.
  const bool token = multi && g_wavefront_mode == 2 && !P.gcnOff && P.Ks > 0;
  hipEvent_t lastMix = nullptr;
  hipStream_t lastMixStream = nullptr;
  auto mix_phase = [&](int l, int t, int phase, const Node16Args* res, hipStream_t cs) -> int {
    if (token && lastMix && lastMixStream != cs) HIP_OK(hipStreamWaitEvent(cs, lastMix, 0));
    RETURN_IF(cell_phase(c, l, t, phase, nullptr, res, cs));
    if (phase == 0 && ((multi && l + 1 < P.L) || token)) HIP_OK(hipEventRecord(W.mixed[l][t], cs));
    if (phase == 2 && token) HIP_OK(hipEventRecord(W.mixz[l][t], cs));
    if (token) { lastMix = phase == 0 ? W.mixed[l][t] : W.mixz[l][t]; lastMixStream = cs; }
    return MATGCN_OK;
  };
  int nextChunk[MATGCN_MAX_LAYERS] = {0};
  // ---- one step of one layer ----
  auto enqueue_step = [&](int l, int t) -> int {
    hipStream_t cs = chain_stream(l);
    hipStream_t xs = multi ? W.xpart[l] : c.s;
    const float* below = (l == 0) ? nullptr : c.ws + P.oSeq[l - 1];
    float* seq = c.ws + P.oSeq[l];
    if (P.gcnOff) {
      // ablation: the layer is a plain GRU cell on (x_t, h) (MultiATGCN.py:187-192,204): one launch per step
      if (multi && l > 0) HIP_OK(hipStreamWaitEvent(cs, W.step[l - 1][t], 0));
      Node16Args a;
      memset(&a, 0, sizeof(a));
      a.s = c.ws + P.oHx[l]; a.hout = c.ws + P.oHx[l];
      a.rows = P.B; a.N = P.N; a.Np = P.Np; a.Ks = 0;
      if (l == 0) fill_res_args(c, l, x0p + (long)t * P.Np * P.C0, (long)P.T * P.Np * P.C0, nullptr, seq + t * stepRows, &a);
      else fill_res_args(c, l, below + t * stepRows, (long)P.Np * H, nullptr, seq + t * stepRows, &a);
      {
        ProfScope prof(MATGCN_PROF_RES, cs);
        const dim3 grid(node_items(P.N, P.B, NODE_ROWS));
        if (c.train) {   // training keeps z, r, hc of the dense cell (slots of the residual cell)
          const size_t at = (size_t)t * P.B * P.Np * H;
          a.svZ2 = c.train + c.R.oZ2[l] + at; a.svR2 = c.train + c.R.oR2[l] + at; a.svHC2 = c.train + c.R.oHC2[l] + at;
          hipLaunchKernelGGL((k_update16<2, true, NODE_ROWS>), grid, dim3(512), UPDATE_SAVE_LDS, cs, a);
        } else {
          hipLaunchKernelGGL((k_update16<2, false, NODE_ROWS>), grid, dim3(512), UPDATE_LDS, cs, a);
        }
      }
      CHECK_LAUNCH();
      if (multi && l + 1 < P.L) HIP_OK(hipEventRecord(W.step[l][t], cs));
      return MATGCN_OK;
    }
    if (l > 0 && t == nextChunk[l]) {
      // x-part chunk [t, t+nt) of this layer, as soon as the layer below has produced those steps; the first
      // chunks are short (1, 1, 2 steps) so that this layer starts one step behind the layer below
      const int nt = chunk_steps(P, t);
      nextChunk[l] = t + nt;
      // steps whose mixed rows the layer below has already written into this chunk's block (its recurrent mix of
      // h_t at its step t+1): all but the sequence's last step
      float* slot; long stride;
      int mixedSteps = 0;
      if (shared_mix_slot(c, l - 1, t, &slot, &stride)) mixedSteps = P.T - 1 - t < nt ? P.T - 1 - t : nt;
      if (multi) {
        if (mixedSteps == nt) HIP_OK(hipStreamWaitEvent(xs, W.mixed[l - 1][t + nt], 0));
        else HIP_OK(hipStreamWaitEvent(xs, W.step[l - 1][t + nt - 1], 0));
      }
      RETURN_IF(hoist_x(c, l, below + t * stepRows, t, nt, xs, mixedSteps));
      if (multi) {
        HIP_OK(hipEventRecord(W.xdone[l][t], xs));
        HIP_OK(hipStreamWaitEvent(cs, W.xdone[l][t], 0));
      }
    }
    Node16Args res;
    if (l == 0)
      fill_res_args(c, l, x0p + (long)t * P.Np * P.C0, (long)P.T * P.Np * P.C0,
                    c.prm->weights_gru + (size_t)l * P.T + t, seq + t * stepRows, &res);
    else
      fill_res_args(c, l, below + t * stepRows, (long)P.Np * H, c.prm->weights_gru + (size_t)l * P.T + t,
                    seq + t * stepRows, &res);
    RETURN_IF(mix_phase(l, t, 0, &res, cs));
    // layer 0's weight streams are first read here: head fusion, the fold of x0 and the first mix ran beside their
    // preparation (lazy prepare)
    if (lazyPrep && multi && l == 0 && t == 0) RETURN_IF(prep_wait(cs, 1));
    RETURN_IF(cell_phase(c, l, t, 1, nullptr, &res, cs));
    RETURN_IF(mix_phase(l, t, 2, &res, cs));
    RETURN_IF(cell_phase(c, l, t, 3, nullptr, &res, cs));
    if (multi && l + 1 < P.L) HIP_OK(hipEventRecord(W.step[l][t], cs));
    return MATGCN_OK;
  };
  // ---- global order: layer l runs `lag` steps behind layer l-1 (lag = T: layer after layer, the free-running wavefront's
  // enqueue order - its events only tie a layer to the one below).  The x-part chunk of layer l that starts at step t waits
  // for step t + nt (<= t + X_CHUNK) of the layer below, which must have been ENQUEUED by then: lag = X_CHUNK + 1. ----
  const int lag = token ? X_CHUNK + 1 : P.T;
  for (int g = 0; g < P.T + lag * (P.L - 1); ++g)
    for (int l = 0; l < P.L; ++l) {
      const int t = g - lag * l;
      if (t >= 0 && t < P.T) RETURN_IF(enqueue_step(l, t));
    }
  // ---- per layer: final states, join ----
  for (int l = 0; l < P.L; ++l) {
    hipStream_t cs = chain_stream(l);
    if (finalsUser) {
      hipLaunchKernelGGL(k_unpack_rows, dim3(blocks_for((size_t)P.B * P.N * H)), dim3(256), 0, cs, c.ws + P.oHx[l],
                         finalsUser + (size_t)l * P.B * P.N * H, P.B, P.N, P.Np, H);
      CHECK_LAUNCH();
    }
    if (multi && l > 0) HIP_OK(hipEventRecord(W.done[l], cs));
  }
  if (multi)
    for (int l = 1; l < P.L; ++l) HIP_OK(hipStreamWaitEvent(c.s, W.done[l], 0));   // join
  return MATGCN_OK;
}

int fuse_padded(const Ctx& c, const float* X, float* x0p, const int32_t* labelStart = nullptr,
                const int32_t* relSteps = nullptr, int64_t seriesSteps = 0) {
  const Plan& P = c.P;
  const matgcn_dims* D = c.D;
  RETURN_IF(zero_async(x0p, (long)P.B * P.T * P.Np * P.C0, c.s));
  FuseArgs a;
  memset(&a, 0, sizeof(a));
  a.X = X; a.x0 = x0p; a.tsg = c.prm->weight_tsg;
  a.labelStart = labelStart; a.seriesSteps = (long)seriesSteps;
  if (labelStart)
    for (int s2 = 0; s2 < D->x_steps; ++s2) a.rel[s2] = relSteps[s2];
  for (int h = 0; h < D->n_heads; ++h) { a.ts[h] = c.prm->weight_ts[h]; a.headBegin[h] = D->head_begin[h]; }
  for (int j = 0; j < P.C0 - P.od; ++j) a.extSrc[j] = D->ext_src[j];
  a.B = P.B; a.T = P.T; a.N = P.N; a.Np = P.Np; a.C0 = P.C0; a.od = P.od; a.F = D->x_feat;
  a.xSteps = D->x_steps; a.startDim = D->start_dim; a.nHeads = D->n_heads; a.nTs = D->n_ts;
  hipLaunchKernelGGL(k_fuse_heads, dim3(blocks_for((size_t)P.B * P.T * P.N)), dim3(256), 0, c.s, a);
  return launch_ok();
}

// seqp: time-major padded sequence [T][B][Np][64]
int head_padded(const Ctx& c, const float* seqp, float* out) {
  const Plan& P = c.P;
  HeadArgs a;
  // fnn_off: the head convolves the last step only (MultiATGCN.py:412)
  a.seq = seqp + (size_t)(P.T - P.headT) * P.B * P.Np * H; a.w = c.prep + P.oHead;
  a.bias = c.prm->end_conv_bias; a.out = out;
  a.B = P.B; a.T = P.headT; a.N = P.N; a.Np = P.Np; a.CH = P.CH; a.od = P.od; a.NTc = P.NTc;
  ProfScope prof(MATGCN_PROF_HEAD, c.s);
  hipLaunchKernelGGL(k_head, dim3((unsigned)(P.B * ((P.N + 31) / 32))), dim3(256), 0, c.s, a);
  return launch_ok();
}

// joinPrepare = false: the caller (the wavefront forward) orders its streams behind a lazy matgcn_prepare itself
int make_ctx(Ctx* c, const matgcn_dims* dims, const matgcn_params* params, const void* prepared, void* workspace,
             size_t workspace_bytes, void* stream, bool joinPrepare = true) {
  if (!dims || !params || !workspace) return MATGCN_ERR_NULL;
  RETURN_IF(make_plan(dims, &c->P));
  if (workspace_bytes < (size_t)c->P.workspaceFloats * sizeof(float)) return MATGCN_ERR_SMALL_BUFFER;
  c->D = dims; c->prm = params; c->prep = (const float*)prepared; c->ws = (float*)workspace;
  c->wsBytes = workspace_bytes;
  c->s = (hipStream_t)stream;
  if (prepared && joinPrepare) RETURN_IF(prep_wait(c->s, 3));
  return MATGCN_OK;
}

int check_layer_params(const matgcn_dims* D, const matgcn_params* p) {
  if (!D->gcn_off && (!p->node_emb || !p->weights_gru)) return MATGCN_ERR_NULL;
  for (int l = 0; l < D->layers; ++l) {
    if (!p->res_gate[l].weight || !p->res_gate[l].bias || !p->res_update[l].weight || !p->res_update[l].bias)
      return MATGCN_ERR_NULL;
    if (D->gcn_off) continue;
    if (!p->gate[l].weights_pool || !p->gate[l].bias_pool || !p->update[l].weights_pool || !p->update[l].bias_pool)
      return MATGCN_ERR_NULL;
    if (D->scale_by_g && (!p->gate[l].weights_g || !p->update[l].weights_g)) return MATGCN_ERR_NULL;
    if (!p->res_gate[l].weight || !p->res_gate[l].bias || !p->res_update[l].weight || !p->res_update[l].bias)
      return MATGCN_ERR_NULL;
  }
  return MATGCN_OK;
}

// shared by the three single-step entry points: stage x (B,N,C_l) as a one-step sequence, h as the layer's state,
// and the x part of that step (XA0 for layer 0, PX_l[0] for deeper layers)
int stage_single_step(const Ctx& c, int layer, const float* x, const float* h, bool xpart, float** xin_out) {
  const Plan& P = c.P;
  if (layer < 0 || layer >= P.L) return MATGCN_ERR_BAD_ARG;
  if (P.gcnOff && xpart) return MATGCN_ERR_UNSUPPORTED;   // no graph cell exists in this ablation
  RETURN_IF(node_kernels_ready(P.nodeLds));
  // layer 0 stages into x0p, deeper layers into step 0 of the sequence of the layer below
  float* xin = (layer == 0) ? c.ws + P.oX0p : c.ws + P.oSeq[layer - 1];
  hipLaunchKernelGGL(k_pack_rows, dim3(blocks_for((size_t)P.B * P.Np * P.Cl[layer])), dim3(256), 0, c.s, x, xin, P.B,
                     P.N, P.Np, P.Cl[layer]);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(k_pack_rows, dim3(blocks_for((size_t)P.B * P.Np * H)), dim3(256), 0, c.s, h, c.ws + P.oHx[layer],
                     P.B, P.N, P.Np, H);
  CHECK_LAUNCH();
  RETURN_IF(zero_async(c.ws + P.oZHx[layer], (long)P.B * P.Np * H, c.s));
  if (xpart) {
    if (layer == 0) RETURN_IF(fold_x0(c, xin, 1, c.s));
    else RETURN_IF(hoist_x(c, layer, xin, 0, 1, c.s));
  }
  *xin_out = xin;
  return MATGCN_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

int matgcn_abi_version(void) { return MATGCN_ABI_VERSION; }

int matgcn_masked_mae(const float* pred, const float* y, const int32_t* label_start, int batch, int out_steps, int nodes,
                      int out_dim, int y_steps, int y_feat, int y_start, float mean, float std, float null_val,
                      float min_s, float* partials, float* result, void* stream) {
  if (!pred || !y || !partials || !result) return MATGCN_ERR_NULL;
  if (batch < 1 || out_steps < 1 || out_steps > 64 || nodes < 1 || out_dim < 1 || y_steps < out_steps ||
      y_start < 0 || y_start + out_dim > y_feat)
    return MATGCN_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mae_partial, dim3((unsigned)(batch * out_steps)), dim3(256), 0, s, pred, y, label_start, out_steps,
                     nodes, out_dim, y_steps, y_feat, y_start, mean, std, null_val, min_s, partials);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(k_mae_final, dim3(1), dim3(64), 0, s, partials, batch, out_steps, result);
  return launch_ok();
}

int matgcn_masked_mae_grad(const float* pred, const float* y, const int32_t* label_start, int batch, int out_steps,
                           int nodes, int out_dim, int y_steps, int y_feat, int y_start, float mean, float std,
                           float null_val, float min_s, const float* partials, const float* upstream, float* d_pred,
                           void* stream) {
  if (!pred || !y || !partials || !upstream || !d_pred) return MATGCN_ERR_NULL;
  if (batch < 1 || out_steps < 1 || out_steps > 64 || nodes < 1 || out_dim < 1 || y_steps < out_steps ||
      y_start < 0 || y_start + out_dim > y_feat)
    return MATGCN_ERR_BAD_ARG;
  const size_t total = (size_t)batch * out_steps * nodes * out_dim;
  hipLaunchKernelGGL(k_mae_grad, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, pred, y, label_start,
                     out_steps, nodes, out_dim, y_steps, y_feat, y_start, mean, std, null_val, min_s,
                     partials + 2 * (size_t)batch * out_steps, upstream, total, d_pred);
  return launch_ok();
}

int matgcn_metric_sums(const float* pred, const float* y, const int32_t* label_start, int batch, int out_steps, int nodes,
                       int out_dim, int y_steps, int y_feat, int y_start, const matgcn_metric_scale* scale,
                       double* partials, double* sums, int accumulate, void* stream) {
  if (!pred || !y || !scale || !partials || !sums) return MATGCN_ERR_NULL;
  if (batch < 1 || out_steps < 1 || out_steps > 64 || nodes < 1 || out_dim < 1 || y_steps < out_steps ||
      y_start < 0 || y_start + out_dim > y_feat)
    return MATGCN_ERR_BAD_ARG;
  if ((scale->mean == nullptr) != (scale->std == nullptr) || (scale->mean2 == nullptr) != (scale->std2 == nullptr))
    return MATGCN_ERR_BAD_ARG;
  MetricArgs a;
  a.pred = pred; a.y = y; a.labelStart = label_start;
  a.mean = scale->mean; a.std = scale->std; a.mean2 = scale->mean2; a.std2 = scale->std2; a.perNode = scale->per_node;
  a.clampMin = scale->clamp_min; a.truthMin = scale->truth_min; a.minS = scale->min_s;
  a.outSteps = out_steps; a.N = nodes; a.od = out_dim; a.ySteps = y_steps; a.yFeat = y_feat; a.yStart = y_start;
  a.partials = partials;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_metric_partial, dim3((unsigned)(batch * out_steps)), dim3(256), 0, s, a);
  CHECK_LAUNCH();
  hipLaunchKernelGGL(k_metric_accumulate, dim3((unsigned)out_steps), dim3(64), 0, s, partials, batch, out_steps,
                     accumulate, sums);
  return launch_ok();
}

int matgcn_metric_table(const double* sums, int out_steps, int swap_r2, double* table, void* stream) {
  if (!sums || !table) return MATGCN_ERR_NULL;
  if (out_steps < 1 || out_steps > 64) return MATGCN_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_metric_table, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, out_steps, swap_r2, table);
  return launch_ok();
}

int matgcn_set_lazy_prepare(int enabled) {
  const int prev = g_lazy_prepare;
  g_lazy_prepare = enabled ? 1 : 0;
  return prev;
}

int matgcn_prepare_join(void* stream) {
  return prep_wait((hipStream_t)stream, 3);
}

int matgcn_set_batch_split(int parts) {
  const int prev = g_batch_split;
  g_batch_split = parts == 2 ? 2 : 0;
  return prev;
}

int matgcn_series_violations(int64_t* count, int reset) {
  if (!count) return MATGCN_ERR_NULL;
  unsigned long long v = 0;
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_series_violations), sizeof(v)));
  *count = (int64_t)v;
  if (reset && v) {
    v = 0;
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_series_violations), &v, sizeof(v)));
  }
  return MATGCN_OK;
}

int matgcn_set_mix_precision(int mode) {
  const int prev = g_mix_precision;
  g_mix_precision = (mode == 1 || mode == 2) ? mode : 0;
  return prev;
}

int matgcn_set_stream_pool(int own) {
  for (int d = 0; d < MAX_DEVICES; ++d)
    for (int set = 0; set < 2; ++set)
      if (g_wfs[d][set].ready) return own ? (g_stream_pool ? MATGCN_OK : MATGCN_ERR_BAD_ARG) : (g_stream_pool ? MATGCN_ERR_BAD_ARG : MATGCN_OK);
  g_stream_pool = own ? 1 : 0;
  return MATGCN_OK;
}

int matgcn_set_wavefront(int mode) {
  const int prev = g_wavefront_mode;
  g_wavefront_mode = (mode == 1 || mode == 2) ? mode : (mode != 0 ? 1 : 0);
  return prev;
}

const char* matgcn_error_string(int status) {
  switch (status) {
    case MATGCN_OK: return "ok";
    case MATGCN_ERR_NULL: return "required pointer is NULL";
    case MATGCN_ERR_BAD_ARG: return "inconsistent or out-of-range dims";
    case MATGCN_ERR_UNSUPPORTED: return "configuration not supported by this build";
    case MATGCN_ERR_SMALL_BUFFER: return "prepared/workspace buffer too small";
    case MATGCN_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown matgcn status";
  }
}

int matgcn_prepared_bytes(const matgcn_dims* dims, size_t* bytes) {
  if (!bytes) return MATGCN_ERR_NULL;
  Plan P;
  RETURN_IF(make_plan(dims, &P));
  *bytes = (size_t)P.preparedFloats * sizeof(float);
  return MATGCN_OK;
}

int matgcn_workspace_bytes(const matgcn_dims* dims, size_t* bytes) {
  if (!bytes) return MATGCN_ERR_NULL;
  Plan P;
  RETURN_IF(make_plan(dims, &P));
  size_t need = (size_t)(g_mix_precision == 2 ? P.workspaceFloatsBf16 : P.workspaceFloats) * sizeof(float);
  if (dims->batch >= 2 && !(dims->batch & 1)) {     // room for the two half-batch plans of the batch-split forward
    matgcn_dims half = *dims;
    half.batch = dims->batch / 2;
    Plan Q;
    if (make_plan(&half, &Q) == MATGCN_OK && 2 * (size_t)Q.workspaceFloats * sizeof(float) > need)
      need = 2 * (size_t)Q.workspaceFloats * sizeof(float);
  }
  *bytes = need;
  return MATGCN_OK;
}

int matgcn_supports_layout(const matgcn_dims* dims, int64_t out[4]) {
  if (!out) return MATGCN_ERR_NULL;
  Plan P;
  RETURN_IF(make_plan(dims, &P));
  out[0] = P.oSt; out[1] = P.Mp; out[2] = P.Np; out[3] = P.Ks;
  return MATGCN_OK;
}

int matgcn_weights_layout(const matgcn_dims* dims, int layer, int part, int64_t out[4]) {
  if (!out) return MATGCN_ERR_NULL;
  Plan P;
  RETURN_IF(make_plan(dims, &P));
  if (layer < 0 || layer >= P.L || part < 0 || part > 1 || P.gcnOff) return MATGCN_ERR_BAD_ARG;
  const int O = part == 0 ? 128 : 64, nG = 4 * P.Ktot;
  out[0] = part == 0 ? P.oWg[layer] : P.oWu[layer];
  out[1] = (long)(nG + P.nGx[layer]) * 16 * O;
  out[2] = nG; out[3] = O / 16;
  return MATGCN_OK;
}

static int prepare_impl(const matgcn_dims* dims, const matgcn_params* params, void* prepared, size_t prepared_bytes,
                        void* workspace, size_t workspace_bytes, void* stream);
// (a failure between the fork onto the library streams and their join - eager or lazy - joins them into the caller's
// stream before the error code is returned, like every other forking entry point: the caller may free or reuse
// `prepared` as soon as its own stream is idle)
int matgcn_prepare(const matgcn_dims* dims, const matgcn_params* params, void* prepared, size_t prepared_bytes,
                   void* workspace, size_t workspace_bytes, void* stream) {
  JOINED(prepare_impl(dims, params, prepared, prepared_bytes, workspace, workspace_bytes, stream), stream);
}
static int prepare_impl(const matgcn_dims* dims, const matgcn_params* params, void* prepared, size_t prepared_bytes,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!prepared) return MATGCN_ERR_NULL;
  Ctx c;
  RETURN_IF(make_ctx(&c, dims, params, prepared, workspace, workspace_bytes, stream));
  const Plan& P = c.P;
  if (prepared_bytes < (size_t)P.preparedFloats * sizeof(float)) return MATGCN_ERR_SMALL_BUFFER;
  RETURN_IF(check_layer_params(dims, params));
  if (!P.gcnOff) {
    if (dims->adp_mode == MATGCN_ADP_UNI && (!params->node_vec1 || !params->node_vec2)) return MATGCN_ERR_NULL;
    if (dims->n_static > 0 && !params->static_supports) return MATGCN_ERR_NULL;   // also the diagonal ones
  }
  if (!params->end_conv_weight || !params->end_conv_bias) return MATGCN_ERR_NULL;
  float* prep = (float*)prepared;
  float* St = prep + P.oSt;
  const int per = P.per;  // stack slots per first-order support
  if (P.Mp > 0) RETURN_IF(zero_async(St, (long)P.Np * P.Mp, c.s));
  const bool cheb = dims->cheb_k > 2 && P.nDenseFirst > 0;
  const bool hasAdp = dims->adp_mode != MATGCN_ADP_NONE && !P.gcnOff;
  float* plainA = (cheb || hasAdp) ? prep + P.oPlainA : nullptr;  // T_{k-1}; staging of the adaptive adjacency
  float* plainB = cheb ? prep + P.oPlainB : nullptr;  // T_{k-2} / product scratch
  float* plainC = cheb ? prep + P.oPlainC : nullptr;
  if (cheb || hasAdp) RETURN_IF(zero_async(plainA, (long)P.Np * P.NpC, c.s));
  if (cheb) {
    RETURN_IF(zero_async(plainB, (long)P.Np * P.NpC, c.s));
    RETURN_IF(zero_async(plainC, (long)P.Np * P.NpC, c.s));
  }
  const dim3 tgrid((unsigned)((P.N + 31) / 32), (unsigned)((P.N + 31) / 32));
  for (int fd = 0; fd < P.nDenseFirst; ++fd) {   // diagonal supports are folded into the weights instead
    const int f = P.denseFirst[fd];
    const int slot0 = P.sumDense ? 0 : fd * per;   // cheb_order = 1: every dense support adds into the one slot
    const int col0 = slot0 * P.Np;
    const int accumulate = P.sumDense && fd > 0 ? 1 : 0;
    const bool adaptive = (dims->adp_mode != MATGCN_ADP_NONE) && f == 0;   // always the first dense support
    if (adaptive) {
      const bool bi = dims->adp_mode == MATGCN_ADP_BI;
      hipLaunchKernelGGL(k_adaptive_adj, dim3(P.N), dim3(256), (size_t)P.N * sizeof(float), c.s,
                         bi ? params->node_emb : params->node_vec1, bi ? nullptr : params->node_vec2,
                         bi ? dims->embed_dim : dims->adj_rank, bi ? 1 : 0, P.N, plainA, P.NpC);
      CHECK_LAUNCH();
      hipLaunchKernelGGL(k_static_transpose, tgrid, dim3(256), 0, c.s, plainA, P.N, P.NpC, St, P.Mp, col0, plainA, P.NpC,
                         accumulate);
    } else {
      const int sidx = f - (dims->adp_mode != MATGCN_ADP_NONE ? 1 : 0);
      hipLaunchKernelGGL(k_static_transpose, tgrid, dim3(256), 0, c.s,
                         params->static_supports + (size_t)sidx * P.N * P.N, P.N, P.N, St, P.Mp, col0,
                         cheb ? plainA : nullptr, P.NpC, accumulate);
    }
    CHECK_LAUNCH();
    // Chebyshev orders 2..cheb_k-1 of this support: T_k = 2 S T_{k-1} - T_{k-2} (T_0 = I, T_1 = S).
    // Three plain N x N buffers rotate; the product S.T_{k-1} is combined in place.
    float* buf[3] = {plainA, plainB, plainC};
    int iPrev1 = 0, iPrev2 = -1;
    for (int k = 2; k < dims->cheb_k; ++k) {
      const int iOut = (iPrev2 < 0) ? 1 : 3 - iPrev1 - iPrev2;
      RETURN_IF(launch_mix(P, St + col0, buf[iPrev1], 64, P.NpC, P.NpC / 64, buf[iOut], (long)P.NpC, 0, 64, 1, P.Np,
                           c.s));
      hipLaunchKernelGGL(k_cheb_combine, tgrid, dim3(256), 0, c.s, buf[iOut], iPrev2 < 0 ? nullptr : buf[iPrev2],
                         iPrev2 < 0 ? 1 : 0, P.N, P.NpC, St, P.Mp, (slot0 + k - 1) * P.Np, buf[iOut]);
      CHECK_LAUNCH();
      iPrev2 = iPrev1;
      iPrev1 = iOut;
    }
  }
  // node-adaptive weights: independent of the support stack and of each other, so the per-(layer, part) pieces
  // go to the library's internal streams (forked from / joined into the caller's stream) and overlap
  RETURN_IF(wavefront_ready());
  Wavefront& W = g_wf;
  hipStream_t pool[3] = {c.s, W.chain[1], W.xpart[1]};
  const bool fork = g_wavefront_mode != 0;
  const bool lazy = fork && g_lazy_prepare != 0;   // leave the weight streams running behind events (see PrepEvents)
  PrepEvents& E = prep_current();
  if (lazy) RETURN_IF(prep_events_ready());
  E.pending = false;                               // what an earlier lazy prepare left behind was joined by make_ctx above
  if (fork) {
    HIP_OK(hipEventRecord(W.fork, c.s));
    HIP_OK(hipStreamWaitEvent(pool[1], W.fork, 0));
    HIP_OK(hipStreamWaitEvent(pool[2], W.fork, 0));
  }
  int piece = 0;
  const StackMap map = build_stack_map(P, dims, params);
  const unsigned nodeGroups = (unsigned)((P.N + PREP_NB - 1) / PREP_NB);
#ifndef PREP_KERNEL
#define PREP_KERNEL 2      // 0: k_prep_stream (round 1, lab builds), 2: k_prep_mfma
#endif
#ifndef PREP_TPW
#define PREP_TPW 4         // k_prep_mfma: 16-node tiles per wave (a workgroup = 4 waves walks 4 * PREP_TPW tiles)
#endif
  const bool fast = PREP_KERNEL == 2 && P.d <= 32;
  const int nTiles = (P.N + 15) / 16, tilesPerBlock = 4 * PREP_TPW;
  const unsigned tileBlocks = (unsigned)((nTiles + tilesPerBlock - 1) / tilesPerBlock);
  auto launch_fast = [&](const PrepStream& q, int O, hipStream_t ws) {
    const dim3 grid((unsigned)((q.groups + q.groupsX) * (O / 16) * 2), tileBlocks);
    if (P.d <= 12) hipLaunchKernelGGL(k_prep_mfma<3>, grid, dim3(256), 0, ws, q, tilesPerBlock);
    else if (P.d <= 20) hipLaunchKernelGGL(k_prep_mfma<5>, grid, dim3(256), 0, ws, q, tilesPerBlock);
    else hipLaunchKernelGGL(k_prep_mfma<8>, grid, dim3(256), 0, ws, q, tilesPerBlock);
  };
  for (int l = 0; l < P.L; ++l) {
    const int I = P.Cl[l] + H;
    for (int part = 0; part < 2 && !P.gcnOff; ++part) {  // 0 gate (O=128), 1 update (O=64)
      const matgcn_agcn_params& ap = part == 0 ? params->gate[l] : params->update[l];
      const int O = part == 0 ? 128 : 64;
      const int nG = 4 * P.Ktot;
      // lazy: the pieces alternate between the two library streams only (layer 0 first on both, so that its events come
      // early; the gate piece is twice the update piece: the order flips per layer to balance the streams)
      hipStream_t ws = !fork ? c.s : lazy ? pool[1 + ((part + l) & 1)] : pool[(++piece) % 3];
      PrepStream q;
      memset(&q, 0, sizeof(q));
      q.E = params->node_emb; q.wpool = ap.weights_pool; q.bpool = ap.bias_pool;
      q.wg = dims->scale_by_g ? ap.weights_g : nullptr;
      q.d = P.d; q.I = I; q.O = O; q.N = P.N; q.C0 = P.C0; q.map = map;
      // recurrent (h) rows, 16x16x4 fragment order: groups 0..nG-1 of the node stream
      q.out = prep + (part == 0 ? P.oWg[l] : P.oWu[l]);
      q.nodeStride = (long)(nG + P.nGx[l]) * 16 * O;
      q.baseOfs = 0; q.kind = 0; q.iOfs = P.Cl[l]; q.groups = nG;
      if (fast) {     // layer 0: the folded x rows + bias row sit behind the recurrent rows - one launch writes both
        q.groupsX = l == 0 ? P.nGx[0] : 0;
        launch_fast(q, O, ws);
        q.groupsX = 0;
      } else {
        hipLaunchKernelGGL(k_prep_stream<0>, dim3(blocks_for((size_t)q.groups * (O / 16) * 64), nodeGroups), dim3(256), 0,
                           ws, q);
      }
      CHECK_LAUNCH();
      if (l == 0) {  // folded x rows + bias row, appended to the node stream
        if (!fast) {
          q.baseOfs = (long)nG * 16 * O; q.kind = 1; q.iOfs = 0; q.groups = P.nGx[0];
          hipLaunchKernelGGL(k_prep_stream<1>, dim3(blocks_for((size_t)q.groups * (O / 16) * 64), nodeGroups), dim3(256), 0,
                             ws, q);
          CHECK_LAUNCH();
        }
      } else {
        // hoisted x part: gate column tiles 0..7, update tiles 8..11 of a 192-wide fragment row (k_px16)
        q.out = prep + P.oWx[l]; q.nodeStride = P.wxStride; q.baseOfs = 0;
        q.kind = 0; q.iOfs = 0; q.groups = nG; q.OTdst = 12; q.otOfs = part == 0 ? 0 : 8;
        if (fast)
          launch_fast(q, O, ws);
        else
          hipLaunchKernelGGL(k_prep_stream<0>, dim3(blocks_for((size_t)q.groups * (O / 16) * 64), nodeGroups), dim3(256), 0,
                             ws, q);
        CHECK_LAUNCH();
        hipLaunchKernelGGL(k_prep_bias, dim3(blocks_for((size_t)P.N * O)), dim3(256), 0, ws, params->node_emb,
                           ap.bias_pool, P.d, O, P.N, prep + P.oBx[l], 192, part == 0 ? 0 : 128);
        CHECK_LAUNCH();
      }
    }
    if (lazy && l == 0) {   // layer 0's weight streams are complete on both library streams
      HIP_OK(hipEventRecord(E.l0[0], pool[1]));
      HIP_OK(hipEventRecord(E.l0[1], pool[2]));
    }
    const int nG1 = (P.Cpad[l] + H) / 16;
    hipLaunchKernelGGL(k_prep_linear16, dim3(blocks_for((size_t)nG1 * 8 * 64)), dim3(256), 0, c.s,
                       params->res_gate[l].weight, I, 128, P.Cl[l], P.Cpad[l], nG1, prep + P.oRg[l]);
    CHECK_LAUNCH();
    hipLaunchKernelGGL(k_prep_linear16, dim3(blocks_for((size_t)nG1 * 4 * 64)), dim3(256), 0, c.s,
                       params->res_update[l].weight, I, 64, P.Cl[l], P.Cpad[l], nG1, prep + P.oRu[l]);
    CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_prep_linear, dim3(blocks_for((size_t)(P.headT * H / 8) * P.NTc * 64)), dim3(256), 0, c.s,
                     params->end_conv_weight, P.headT * H, P.CH, 0, 0, P.headT * H, P.NTc, prep + P.oHead);
  CHECK_LAUNCH();
  if (lazy) {          // no join: every consumer waits for what it reads (make_ctx, encoder_chains)
    HIP_OK(hipEventRecord(E.l1[0], pool[1]));
    HIP_OK(hipEventRecord(E.l1[1], pool[2]));
    E.pending = true;
  } else if (fork) {
    HIP_OK(hipEventRecord(W.done[1], pool[1]));
    HIP_OK(hipEventRecord(W.done[2], pool[2]));
    HIP_OK(hipStreamWaitEvent(c.s, W.done[1], 0));
    HIP_OK(hipStreamWaitEvent(c.s, W.done[2], 0));
  }
  return MATGCN_OK;
}

// one whole inference forward of `dims->batch` samples on stream c.s with the current wavefront set: head fusion
// (from windows X, or - series != null - gathered from the resident series), encoder, head
static int forward_once(Ctx& c, const float* X, const float* series, int64_t seriesSteps, const int32_t* labelStart,
                        const int32_t* relSteps, const float* h0, float* out) {
  const Plan& P = c.P;
  float* x0p = c.ws + P.oX0p;
  if (series) RETURN_IF(fuse_padded(c, series, x0p, labelStart, relSteps, seriesSteps));
  else RETURN_IF(fuse_padded(c, X, x0p));
  RETURN_IF(encoder_padded(c, x0p, h0, nullptr));
  return head_padded(c, c.ws + P.oSeq[P.L - 1], out);
}

// The batch-split forward (matgcn_set_batch_split(2)): the two halves of the batch are two independent forwards of
// B / 2 samples - no arithmetic ties them, MultiATGCN.py:363-420 is per sample - run side by side: half 0 on the caller's
// stream with wavefront set 0, half 1 on a library stream with set 1, each in its own half of the workspace, joined at
// the end.  Four chains of half-size kernels instead of two: their fixed launch costs hide behind each other's work and
// their smaller grids leave room to co-reside.  Same kernels, same per-sample arithmetic - the
// results are those of two forwards of B / 2.
static int forward_split(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* X,
                         const float* series, int64_t seriesSteps, const int32_t* labelStart, const int32_t* relSteps,
                         const float* h0, float* out, void* workspace, size_t workspace_bytes, void* stream,
                         bool* done) {
  *done = false;
  if (g_batch_split != 2 || g_wavefront_mode == 0 || g_mix_precision != 0 || dims->batch < 2 || (dims->batch & 1))
    return MATGCN_OK;
  matgcn_dims half = *dims;
  half.batch = dims->batch / 2;
  Ctx c0, c1;
  Plan probe;
  RETURN_IF(make_plan(&half, &probe));
  const size_t halfBytes = (size_t)probe.workspaceFloats * sizeof(float);
  if (2 * halfBytes > workspace_bytes) return MATGCN_OK;          // does not fit: the caller runs the plain forward
  RETURN_IF(split_ready());
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
  SplitStreams& S = g_split[dev];
  RETURN_IF(make_ctx(&c0, &half, params, prepared, workspace, halfBytes, stream, false));
  RETURN_IF(make_ctx(&c1, &half, params, prepared, (char*)workspace + halfBytes, halfBytes, S.s1, false));
  const size_t layerStride = (size_t)dims->batch * dims->nodes * H;
  c0.h0LayerStride = c1.h0LayerStride = layerStride;
  const int hb = half.batch;
  const size_t xHalf = (size_t)hb * dims->x_steps * dims->nodes * dims->x_feat;
  const size_t outHalf = (size_t)hb * dims->out_channels * dims->nodes;
  const size_t h0Half = (size_t)hb * dims->nodes * H;
  HIP_OK(hipEventRecord(S.fork, c0.s));
  HIP_OK(hipStreamWaitEvent(S.s1, S.fork, 0));
  g_wf_set = 0;
  int rc = forward_once(c0, X, series, seriesSteps, labelStart, relSteps, h0, out);
  if (rc == MATGCN_OK) {
    g_wf_set = 1;
    rc = forward_once(c1, X ? X + xHalf : nullptr, series, seriesSteps, labelStart ? labelStart + hb : nullptr, relSteps,
                      h0 ? h0 + h0Half : nullptr, out + outHalf);
  }
  g_wf_set = 0;
  if (rc != MATGCN_OK) { join_library_streams(c0.s); return rc; }
  HIP_OK(hipEventRecord(S.done, S.s1));
  HIP_OK(hipStreamWaitEvent(c0.s, S.done, 0));
  *done = true;
  return MATGCN_OK;
}

static int forward_entry(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* X,
                         const float* h0, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!prepared || !X || !out) return MATGCN_ERR_NULL;
  Ctx c;
  RETURN_IF(make_ctx(&c, dims, params, prepared, workspace, workspace_bytes, stream, false));   // chains wait per layer
  if (!params->weight_tsg || !params->end_conv_bias) return MATGCN_ERR_NULL;
  for (int h = 0; h < dims->n_heads; ++h) if (!params->weight_ts[h]) return MATGCN_ERR_NULL;
  RETURN_IF(check_layer_params(dims, params));
  bool done = false;
  RETURN_IF(forward_split(dims, params, prepared, X, nullptr, 0, nullptr, nullptr, h0, out, workspace, workspace_bytes,
                          stream, &done));
  if (done) return MATGCN_OK;
  MixPrecisionScope mixScope(true);
  return forward_once(c, X, nullptr, 0, nullptr, nullptr, h0, out);
}

int matgcn_forward(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* X,
                   const float* h0, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  return on_main_stream(stream, [&](void* s) {
    return forward_entry(dims, params, prepared, X, h0, out, workspace, workspace_bytes, s);
  });
}

// the host-visible part of the series range contract: no window row may start before the series
static int check_series(const matgcn_dims* dims, const float* series, int64_t series_steps, const int32_t* label_start,
                 const int32_t* rel_steps) {
  if (!series || !label_start || !rel_steps) return MATGCN_ERR_NULL;
  if (dims->x_steps > MATGCN_MAX_XSTEPS) return MATGCN_ERR_UNSUPPORTED;
  int lo = 0;
  for (int s2 = 0; s2 < dims->x_steps; ++s2) lo = rel_steps[s2] < lo ? rel_steps[s2] : lo;
  if (series_steps < 1 || -(int64_t)lo >= series_steps) return MATGCN_ERR_BAD_ARG;
  return MATGCN_OK;
}

static int forward_series_entry(const matgcn_dims* dims, const matgcn_params* params, const void* prepared,
                                const float* series, int64_t series_steps, const int32_t* label_start,
                                const int32_t* rel_steps, const float* h0, float* out, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!prepared || !series || !label_start || !rel_steps || !out) return MATGCN_ERR_NULL;
  Ctx c;
  RETURN_IF(make_ctx(&c, dims, params, prepared, workspace, workspace_bytes, stream, false));   // chains wait per layer
  if (!params->weight_tsg || !params->end_conv_bias) return MATGCN_ERR_NULL;
  for (int h = 0; h < dims->n_heads; ++h) if (!params->weight_ts[h]) return MATGCN_ERR_NULL;
  RETURN_IF(check_layer_params(dims, params));
  RETURN_IF(check_series(dims, series, series_steps, label_start, rel_steps));
  bool done = false;
  RETURN_IF(forward_split(dims, params, prepared, nullptr, series, series_steps, label_start, rel_steps, h0, out, workspace,
                          workspace_bytes, stream, &done));
  if (done) return MATGCN_OK;
  MixPrecisionScope mixScope(true);
  return forward_once(c, nullptr, series, series_steps, label_start, rel_steps, h0, out);
}

int matgcn_forward_series(const matgcn_dims* dims, const matgcn_params* params, const void* prepared,
                          const float* series, int64_t series_steps, const int32_t* label_start,
                          const int32_t* rel_steps, const float* h0, float* out, void* workspace,
                          size_t workspace_bytes, void* stream) {
  return on_main_stream(stream, [&](void* s) {
    return forward_series_entry(dims, params, prepared, series, series_steps, label_start, rel_steps, h0, out, workspace,
                                workspace_bytes, s);
  });
}

int matgcn_fuse_heads(const matgcn_dims* dims, const matgcn_params* params, const float* X, float* x0,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!X || !x0) return MATGCN_ERR_NULL;
  Ctx c;
  RETURN_IF(make_ctx(&c, dims, params, nullptr, workspace, workspace_bytes, stream));
  if (!params->weight_tsg) return MATGCN_ERR_NULL;
  for (int h = 0; h < dims->n_heads; ++h) if (!params->weight_ts[h]) return MATGCN_ERR_NULL;
  const Plan& P = c.P;
  float* x0p = c.ws + P.oX0p;
  RETURN_IF(fuse_padded(c, X, x0p));
  hipLaunchKernelGGL(k_unpack_rows, dim3(blocks_for((size_t)P.B * P.T * P.N * P.C0)), dim3(256), 0, c.s, x0p, x0,
                     P.B * P.T, P.N, P.Np, P.C0);
  return launch_ok();
}

int matgcn_agcn_gate_fwd(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, int layer,
                         const float* x, const float* h, float* y, void* workspace, size_t workspace_bytes,
                         void* stream) {
  if (!prepared || !x || !h || !y) return MATGCN_ERR_NULL;
  Ctx c;
  RETURN_IF(make_ctx(&c, dims, params, prepared, workspace, workspace_bytes, stream));
  float* xin;
  RETURN_IF(stage_single_step(c, layer, x, h, true, &xin));
  return cell_step(c, layer, 0, y, true, nullptr, c.s);
}

int matgcn_atgru_cell_fwd(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, int layer,
                          const float* x, const float* h, float* h_out, void* workspace, size_t workspace_bytes,
                          void* stream) {
  if (!prepared || !x || !h || !h_out) return MATGCN_ERR_NULL;
  Ctx c;
  RETURN_IF(make_ctx(&c, dims, params, prepared, workspace, workspace_bytes, stream));
  const Plan& P = c.P;
  float* xin;
  RETURN_IF(stage_single_step(c, layer, x, h, true, &xin));
  RETURN_IF(cell_step(c, layer, 0, nullptr, false, nullptr, c.s));
  hipLaunchKernelGGL(k_unpack_rows, dim3(blocks_for((size_t)P.B * P.N * H)), dim3(256), 0, c.s, c.ws + P.oHx[layer],
                     h_out, P.B, P.N, P.Np, H);
  return launch_ok();
}

int matgcn_res_cell_fwd(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, int layer,
                        const float* x, const float* h, float* h_out, void* workspace, size_t workspace_bytes,
                        void* stream) {
  if (!prepared || !x || !h || !h_out) return MATGCN_ERR_NULL;
  Ctx c;
  RETURN_IF(make_ctx(&c, dims, params, prepared, workspace, workspace_bytes, stream));
  const Plan& P = c.P;
  float* xin;
  RETURN_IF(stage_single_step(c, layer, x, h, false, &xin));
  RETURN_IF(res_step(c, layer, xin, (long)P.Np * P.Cl[layer]));
  hipLaunchKernelGGL(k_unpack_rows, dim3(blocks_for((size_t)P.B * P.N * H)), dim3(256), 0, c.s, c.ws + P.oHx[layer],
                     h_out, P.B, P.N, P.Np, H);
  return launch_ok();
}

int matgcn_encoder_fwd(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* x0,
                       const float* h0, float* seq, float* finals, void* workspace, size_t workspace_bytes,
                       void* stream) {
  if (!prepared || !x0) return MATGCN_ERR_NULL;
  Ctx c;
  RETURN_IF(make_ctx(&c, dims, params, prepared, workspace, workspace_bytes, stream));
  RETURN_IF(check_layer_params(dims, params));
  const Plan& P = c.P;
  float* x0p = c.ws + P.oX0p;
  hipLaunchKernelGGL(k_pack_rows, dim3(blocks_for((size_t)P.B * P.T * P.Np * P.C0)), dim3(256), 0, c.s, x0, x0p,
                     P.B * P.T, P.N, P.Np, P.C0);
  CHECK_LAUNCH();
  RETURN_IF(encoder_padded(c, x0p, h0, finals));
  if (seq) {
    hipLaunchKernelGGL(k_unpack_seq_tm, dim3(blocks_for((size_t)P.B * P.T * P.N * H)), dim3(256), 0, c.s,
                       c.ws + P.oSeq[P.L - 1], seq, P.B, P.T, P.N, P.Np, H);
    CHECK_LAUNCH();
  }
  return MATGCN_OK;
}

int matgcn_output_head(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* seq,
                       float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!prepared || !seq || !out) return MATGCN_ERR_NULL;
  Ctx c;
  RETURN_IF(make_ctx(&c, dims, params, prepared, workspace, workspace_bytes, stream));
  if (!params->end_conv_bias) return MATGCN_ERR_NULL;
  const Plan& P = c.P;
  float* seqp = c.ws + P.oSeq[P.L - 1];
  hipLaunchKernelGGL(k_pack_seq_tm, dim3(blocks_for((size_t)P.B * P.T * P.Np * H)), dim3(256), 0, c.s, seq, seqp,
                     P.B, P.T, P.N, P.Np, H);
  CHECK_LAUNCH();
  return head_padded(c, seqp, out);
}

int matgcn_profile_disable(void) {
  if (g_prof.ev) {
    for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
    delete[] g_prof.kinds;
  }
  g_prof = Prof();
  return MATGCN_OK;
}

int matgcn_profile_enable(int kind_mask, int max_launches) {
  if (max_launches < 1 || max_launches > (1 << 20)) return MATGCN_ERR_BAD_ARG;
  matgcn_profile_disable();
  g_prof.ev = new hipEvent_t[2 * (size_t)max_launches];
  g_prof.kinds = new int[max_launches];
  for (int i = 0; i < 2 * max_launches; ++i)
    if (hipEventCreate(&g_prof.ev[i]) != hipSuccess) return MATGCN_ERR_LAUNCH;
  g_prof.cap = max_launches;
  g_prof.mask = kind_mask;
  return MATGCN_OK;
}

int matgcn_profile_collect(float* ms, int* kinds, int capacity, int* count) {
  if (!ms || !count) return MATGCN_ERR_NULL;
  const int n = g_prof.used < capacity ? g_prof.used : capacity;
  for (int i = 0; i < n; ++i) {
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return MATGCN_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms[i], g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return MATGCN_ERR_LAUNCH;
    if (kinds) kinds[i] = g_prof.kinds[i];
  }
  *count = n;
  g_prof.used = 0;
  return MATGCN_OK;
}

}  // extern "C"

#include "matgcn_bwd.hip"
