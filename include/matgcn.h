/*
 * matgcn.h - C ABI of the MI355X-native Multi-ATGCN forward hot path (libmatgcn.so).
 *
 * This is the drop-in boundary for ONE path of SonghuaHu-UMD/MultiSTGraph: MultiATGCN.forward()
 * (reference libcity/model/traffic_flow_prediction/MultiATGCN.py:363-420) and the pieces it is
 * made of.  The reference has no FFI of its own (it is 100 % Python/torch), so each entry point
 * names the reference symbol it replaces; the Python class that binds them through ctypes is
 * multistgraph_amd/model.py (the binding a maintainer would add is shown in INTEGRATION.md).
 *
 * Conventions
 *   - every function returns MATGCN_OK (0) or a negative matgcn_status; no exceptions, no device
 *     allocation, no implicit synchronisation: work is enqueued on the caller's hipStream_t
 *     (passed as void*) - internally forked onto library-owned streams that join back before the
 *     call's last kernel - and the caller owns every buffer including `prepared` and `workspace`.
 *   - every pointer in matgcn_params / X / out / prepared / workspace is a DEVICE pointer to
 *     contiguous row-major fp32, 16-byte aligned (torch allocations are 256-byte aligned).
 *   - matgcn_dims is a plain host struct, read at call time.
 *   - shapes use the reference's names: B batch, T = input_window (24), N nodes, H = rnn_units,
 *     F = feature dim of batch['X'], K = stacked supports including the identity.
 *   - an entry point that fails between forking onto the library streams and joining them joins them into the caller's
 *     stream before it returns its error code: the caller's buffers are quiet once its own stream is.
 *
 * ABI 10 (round 3) over ABI 9: matgcn_metric_sums / matgcn_metric_table (the evaluator's table and the group-std
 * re-transform on the device), matgcn_series_violations (range contract of the series entry points: rows are clamped
 * and counted, never read out of bounds), matgcn_set_mix_precision(2) (bf16 operands for the node-wise contractions),
 * matgcn_set_batch_split, matgcn_set_lazy_prepare / matgcn_prepare_join; matgcn_workspace_bytes also covers two
 * half-batch plans and the bf16 copies of the weight streams.  No signature of ABI 9 changed.
 */
#ifndef MATGCN_H
#define MATGCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MATGCN_ABI_VERSION 11

typedef enum matgcn_status {
  MATGCN_OK = 0,
  MATGCN_ERR_NULL = -1,        /* a required pointer is NULL */
  MATGCN_ERR_BAD_ARG = -2,     /* inconsistent / out-of-range dims */
  MATGCN_ERR_UNSUPPORTED = -3, /* valid in the reference, not built yet */
  MATGCN_ERR_SMALL_BUFFER = -4,/* prepared / workspace smaller than *_bytes() reports */
  MATGCN_ERR_LAUNCH = -5       /* hipGetLastError() != hipSuccess after a launch */
} matgcn_status;

/* adaptive adjacency mode = config['adpadj'] (MultiATGCN.py:80-85) */
enum { MATGCN_ADP_NONE = 0, MATGCN_ADP_UNI = 1, MATGCN_ADP_BI = 2 };

#define MATGCN_MAX_LAYERS 4
#define MATGCN_MAX_HEADS 8
#define MATGCN_MAX_EXT 16
#define MATGCN_MAX_XSTEPS 256 /* x_steps accepted by matgcn_forward_series */

typedef struct matgcn_dims {
  int32_t batch;        /* B */
  int32_t nodes;        /* N */
  int32_t in_steps;     /* T = input_window; the reference hard-codes 24-step heads (:373-393) */
  int32_t x_steps;      /* steps of batch['X'] = len_closeness+len_period+len_trend (96) */
  int32_t x_feat;       /* F */
  int32_t out_channels; /* output_window * output_dim = Conv2d out channels (:340) */
  int32_t out_dim;      /* output_dim = end_dim - start_dim (:320) */
  int32_t start_dim;    /* first flow channel of X (:365) */
  int32_t hidden;       /* H = rnn_units; this build: 64 */
  int32_t layers;       /* num_layers */
  int32_t feat_in;      /* feature_final = out_dim + ext channels fed to layer 0 (:321) */
  int32_t embed_dim;    /* d = node_emb columns */
  int32_t adj_rank;     /* r = node_vec1 columns (unidirection) */
  int32_t adp_mode;     /* MATGCN_ADP_* */
  int32_t n_static;     /* first-order static supports used by the stack (0..3), (:87-93) */
  int32_t cheb_k;       /* cheb_order (>= 1; 1 = one weight broadcast over [I, S_1, S_2, ..], :65-70,94-108) */
  int32_t scale_by_g;   /* 1 iff adjtype == 'multi': stack *= softmax(weights_g) (:102-103) */
  int32_t n_heads;      /* temporal heads fused (2 if output_window < 6 else 4, :371-393) */
  int32_t n_ts;         /* len(weight_tsg) = len_ts (:328-332) */
  int32_t diag_static_mask; /* bit s set: static support s is a diagonal matrix (host-checked, e.g. the
                           * similarity Laplacian -I without static features, :244-250).  Such supports are
                           * folded into the identity slot of the node-adaptive weights and never mixed. */
  int32_t gcn_off;      /* 1: ablation without graph convolution - encoder.agru_cells are dense GRU cells, no
                         * residual cells, no blend (:177-192,205-208); their nn.Linear weights are passed in
                         * matgcn_params.res_gate / res_update, the AGCN / support fields are ignored */
  int32_t fnn_off;      /* 1: ablation of the temporal head - end_conv sees the last step only (:342-344,412) */
  int32_t head_begin[MATGCN_MAX_HEADS]; /* first X step of head h (trend head never advances) */
  int32_t ext_src[MATGCN_MAX_EXT];      /* X channel copied into encoder channel out_dim+j */
} matgcn_dims;

typedef struct matgcn_agcn_params { /* one AGCN (MultiATGCN.py:71-73) */
  const float* weights_g;    /* (K,1,1) */
  const float* weights_pool; /* (d, K, I, O) */
  const float* bias_pool;    /* (d, O) */
} matgcn_agcn_params;

typedef struct matgcn_linear_params { /* one nn.Linear of the residual GRUCell (:139-140) */
  const float* weight; /* (O, I) */
  const float* bias;   /* (O) */
} matgcn_linear_params;

typedef struct matgcn_params { /* device pointers, names = the reference state_dict keys */
  const float* node_emb;        /* (N, d) */
  const float* node_vec1;       /* (N, r)  or NULL */
  const float* node_vec2;       /* (r, N)  or NULL */
  const float* static_supports; /* (n_static, N, N): model.supports[s][1] (:264-283) or NULL */
  const float* weight_tsg;      /* (n_ts) */
  const float* weight_ts[MATGCN_MAX_HEADS]; /* each (1,24,N,out_dim) */
  const float* weights_gru;     /* encoder.weights_gru (L, T) */
  matgcn_agcn_params gate[MATGCN_MAX_LAYERS];     /* encoder.agru_cells.l.gate   (O = 2H) */
  matgcn_agcn_params update[MATGCN_MAX_LAYERS];   /* encoder.agru_cells.l.update (O = H)  */
  matgcn_linear_params res_gate[MATGCN_MAX_LAYERS];   /* encoder.res_cells.l.gate   */
  matgcn_linear_params res_update[MATGCN_MAX_LAYERS]; /* encoder.res_cells.l.update */
  const float* end_conv_weight; /* (out_channels, T, 1, H); (out_channels, 1, 1, H) with fnn_off */
  const float* end_conv_bias;   /* (out_channels) */
} matgcn_params;

/* ---- introspection ------------------------------------------------------------------------ */
int matgcn_abi_version(void);
const char* matgcn_error_string(int status);

/* Bytes of the two caller-owned device buffers for `dims`.  matgcn_workspace_bytes depends on ONE library setting: while
 * matgcn_set_mix_precision(2) is in force it also counts the bf16 copies of the recurrent weight streams (half the bytes of
 * the fp32 streams, behind everything else); a mode-2 forward on a workspace sized without them returns
 * MATGCN_ERR_SMALL_BUFFER (ask again and re-allocate).  The fp32 product path and training never pay for them. */
int matgcn_prepared_bytes(const matgcn_dims* dims, size_t* bytes);
int matgcn_workspace_bytes(const matgcn_dims* dims, size_t* bytes);

/* Where the transposed support stack lives inside `prepared` (for tests / debugging):
 * out[0] = float offset, out[1] = leading dimension, out[2] = padded node count Np,
 * out[3] = number of DENSE non-identity slots Ks (diagonal supports are folded away, in stack order
 * otherwise).  Element S_k[n][m] of dense slot k is at prepared[out[0] + m*out[1] + k*Np + n]. */
int matgcn_supports_layout(const matgcn_dims* dims, int64_t out[4]);

/* Where the RECURRENT rows of a node-adaptive weight stream live inside `prepared` (for tests: what matgcn_prepare
 * really wrote, including diagonal supports folded into the identity slot and softmax(weights_g)).
 * part 0 = agru_cells[layer].gate (O = 128), 1 = .update (O = 64).  out[0] = float offset of node 0, out[1] = floats
 * between nodes, out[2] = k-groups of 16 rows (= 4 * (1 + Ks)), out[3] = column tiles O/16.  Row kk = 64*slot + c is
 * hidden channel c of node-GEMM slot `slot` (0 = identity, then the dense slots); W_n[kk][o] is at
 *   out[0] + n*out[1] + (((kk/16)*out[3] + o/16)*64 + (o%16) + 16*((kk%16)/4))*4 + kk%4   (v_mfma_f32_16x16x4 B order). */
int matgcn_weights_layout(const matgcn_dims* dims, int layer, int part, int64_t out[4]);

/* ---- parameter-only work, once per parameter update ----------------------------------------
 * Replaces what AGCN.forward rebuilds on every call (MultiATGCN.py:78-105): the adaptive adjacency
 * softmax(relu(E1 E2)) / softmax(relu(E E^T)), the Chebyshev support stack, the node-adaptive
 * weights einsum('nd,dkio->nkio') and bias, with softmax(weights_g) folded into the weights; plus
 * MFMA-fragment-ordered copies of the residual-GRU and Conv2d-head weights. */
int matgcn_prepare(const matgcn_dims* dims, const matgcn_params* params, void* prepared,
                   size_t prepared_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* ---- the path ------------------------------------------------------------------------------
 * MultiATGCN.forward / predict (MultiATGCN.py:363-420, eval mode):
 * X (B, x_steps, N, F) -> out (B, output_window, N, output_dim).
 * h0: initial state of the encoder, (L, B, N, H) device, or NULL for the zero state of init_hidden (:214-218, :405).
 * With static features the reference starts every layer and sample from static_initial_gru(static @ v) (:406-409):
 * that (N, H) embedding is host-side torch (a PCA and one nn.Linear), expanded to (L, B, N, H) by the caller. */
int matgcn_forward(const matgcn_dims* dims, const matgcn_params* params, const void* prepared,
                   const float* X, const float* h0, float* out, void* workspace, size_t workspace_bytes,
                   void* stream);

/* The same forward fed from the raw series instead of materialised windows (replaces the window build of
 * MTHDataset._generate_input_data, libcity/data/dataset/dataset_subclass/mth_dataset.py:110-160, and the
 * per-batch host copy of data/utils.py:68-72 + batch.py:43-57): series (series_steps, N, F) resident on the
 * device, label_start (B) device int32 - the first target step of each sample -, rel_steps (x_steps) HOST
 * int32 - offset of every window row relative to its label start (multistgraph_amd/windows.py).
 * Row s of sample b is series[label_start[b] + rel_steps[s]]; the caller guarantees they are in range
 * (multistgraph_amd/windows.py::check_label_starts validates a table on the host where it is built).  The kernels
 * never read outside the series all the same: an out-of-range row is clamped to the first / last row and counted,
 * see matgcn_series_violations. */
int matgcn_forward_series(const matgcn_dims* dims, const matgcn_params* params, const void* prepared,
                          const float* series, int64_t series_steps, const int32_t* label_start,
                          const int32_t* rel_steps, const float* h0, float* out, void* workspace,
                          size_t workspace_bytes, void* stream);

/* The batch as B label starts into the device-resident raw series instead of materialised windows: what
 * matgcn_forward_series takes as separate arguments, as one descriptor for the training entry points. */
typedef struct matgcn_series {
  const float* series;        /* (series_steps, N, F) device */
  int64_t series_steps;
  const int32_t* label_start; /* (B) device: first target step of every sample */
  const int32_t* rel_steps;   /* (x_steps) HOST: offset of every window row relative to its label start */
} matgcn_series;

/* Diagnostic for the series entry points (matgcn_forward_series, the matgcn_series source of the training calls,
 * matgcn_masked_mae{,_grad} with label_start): number of row accesses on the CURRENT device, since the last reset,
 * whose index label_start[b] + offset fell outside the series and was clamped.  0 = the range contract held.
 * Synchronises the device (a debugging / validation call, not part of a step); reset != 0 clears the counter. */
int matgcn_series_violations(int64_t* count, int reset);

/* ---- the pieces (same kernels, exposed for parity tests against the reference's modules) ----
 * temporal-head fusion + channel concat (MultiATGCN.py:365-402): X -> x0 (B, T, N, feat_in) */
int matgcn_fuse_heads(const matgcn_dims* dims, const matgcn_params* params, const float* X,
                      float* x0, void* workspace, size_t workspace_bytes, void* stream);

/* AGCN.forward of agru_cells[layer].gate on cat(x, h) (MultiATGCN.py:76-109):
 * x (B,N,C_l), h (B,N,H) -> y (B,N,2H), no activation. */
int matgcn_agcn_gate_fwd(const matgcn_dims* dims, const matgcn_params* params,
                         const void* prepared, int layer, const float* x, const float* h,
                         float* y, void* workspace, size_t workspace_bytes, void* stream);

/* ATGRUCell.forward of agru_cells[layer] (MultiATGCN.py:120-128): x, h -> h_out (B,N,H) */
int matgcn_atgru_cell_fwd(const matgcn_dims* dims, const matgcn_params* params,
                          const void* prepared, int layer, const float* x, const float* h,
                          float* h_out, void* workspace, size_t workspace_bytes, void* stream);

/* GRUCell.forward of res_cells[layer] (MultiATGCN.py:142-150): x, h -> h_out (B,N,H) */
int matgcn_res_cell_fwd(const matgcn_dims* dims, const matgcn_params* params,
                        const void* prepared, int layer, const float* x, const float* h,
                        float* h_out, void* workspace, size_t workspace_bytes, void* stream);

/* ATGRUEncoder.forward (MultiATGCN.py:194-212): x0 (B,T,N,feat_in), h0 (L,B,N,H) or NULL (zeros)
 * -> seq (B,T,N,H) of the last layer and finals (L,B,N,H); either output may be NULL. */
int matgcn_encoder_fwd(const matgcn_dims* dims, const matgcn_params* params, const void* prepared,
                       const float* x0, const float* h0, float* seq, float* finals,
                       void* workspace, size_t workspace_bytes, void* stream);

/* end_conv + reshape/permute (MultiATGCN.py:416-418, eval): seq (B,T,N,H) -> out */
int matgcn_output_head(const matgcn_dims* dims, const matgcn_params* params, const void* prepared,
                       const float* seq, float* out, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- loss / metric epilogue ---------------------------------------------------------------------
 * De-scale, mask and reduce on the device (MultiATGCN.calculate_loss, MultiATGCN.py:422-427, with
 * loss.masked_mae_torch, libcity/model/loss.py:17-29; TrafficStateEvaluator "single" mode MAE@k,
 * libcity/evaluator/traffic_state_evaluator.py:87-104).  For an affine scaler x -> x*std + mean:
 *   l = y*std + mean, p = pred*std + mean;  l := 0 where |l| < min_s;
 *   mask = (l != null_val)   [null_val NaN: mask = !isnan(l)]
 *   result[0]     = sum(|p-l|*mask) / sum(mask)              (the masked-MAE loss over all horizons)
 *   result[1 + k] = the same restricted to horizon k          (MAE@k+1)
 * pred (B, out, N, od) contiguous; y (B, y_steps >= out, N, y_feat), channels y_start .. y_start+od-1.
 * label_start != NULL (device, B int32): `y` is the raw series (y_steps, N, y_feat) instead and the label rows of
 * sample b are y[label_start[b] + o], o < out - the targets MTHDataset._generate_input_data materialises
 * (mth_dataset.py:110-160) are gathered here, on the device.
 * partials: caller-owned device scratch of 2*B*out + 1 floats (the last one keeps sum(mask) for the gradient);
 * result: device, 1+out floats.  Two launches, fixed summation order (no atomics): results are run-to-run identical. */
int matgcn_masked_mae(const float* pred, const float* y, const int32_t* label_start, int batch, int out_steps, int nodes,
                      int out_dim, int y_steps, int y_feat, int y_start, float mean, float std, float null_val,
                      float min_s, float* partials, float* result, void* stream);

/* Gradient of result[0] (the calculate_loss value) w.r.t. pred, for the training step: d_pred (B, out, N, od) =
 * upstream[0] * std * sign(p - l) * mask / sum(mask), with `partials` as left by the matching matgcn_masked_mae call
 * and `upstream` the device scalar autograd hands down (d loss / d loss = 1 for a plain loss.backward()). */
int matgcn_masked_mae_grad(const float* pred, const float* y, const int32_t* label_start, int batch, int out_steps,
                           int nodes, int out_dim, int y_steps, int y_feat, int y_start, float mean, float std,
                           float null_val, float min_s, const float* partials, const float* upstream, float* d_pred,
                           void* stream);

/* ---- the evaluator's metric table on the device ------------------------------------
 * TrafficStateEvaluator.collect (libcity/evaluator/traffic_state_evaluator.py:34-121: ten metrics per horizon, modes
 * "single" and "average") and the group-std re-transform table of TrafficStateExecutor.evaluate
 * (libcity/executor/traffic_state_executor.py:293-322) without copying predictions to the host: one pass over
 * prediction and label reduces, per horizon, the MATGCN_METRIC_SUMS sums every metric is a ratio of (fp64), and
 * matgcn_metric_table turns (accumulated) sums into the table.
 *   l = y, p = pred;  first affine x*std + mean (scalar, or one pair per node: per_node) - the scaler's
 *   inverse_transform (:268-273);  second affine per node - prediction_t = prediction * All_std + All_m (:307-308);
 *   p := clamp_min where p < clamp_min (:312);  elements with l <= truth_min are left out (:316-317);
 *   l := 0 where |l| < min_s (loss.py:18, 53, 71; min_s < 0: off - the *_np functions of the re-transform do not).
 * sums per horizon: 0 elements, 1 non-NaN labels, 2 sum|p-l|, 3 sum (p-l)^2, 4 sum |(p-l)/l| (NaN terms -> 0, as
 * loss.py does) over those, 5 labels != 0, 6-8 the same three over those, 9 sum l, 10 sum l^2, 11 sum p, 12 sum p^2,
 * 13 sum (l-p).  Geometry arguments as matgcn_masked_mae (label_start != NULL: y is the raw series).
 * partials: caller-owned scratch of batch*out_steps*MATGCN_METRIC_SUMS doubles; sums: out_steps*MATGCN_METRIC_SUMS
 * doubles, overwritten, or added to when accumulate != 0 (several batches = one collect() over their concatenation,
 * which is how the executor evaluates, :289).  Fixed summation order: run-to-run identical. */
#define MATGCN_METRIC_SUMS 14
#define MATGCN_METRICS 10 /* MAE, MAPE, MSE, RMSE, masked_MAE, masked_MAPE, masked_MSE, masked_RMSE, R2, EVAR */
typedef struct matgcn_metric_scale {
  const float* mean;   /* device: 1 value, or N with per_node; NULL (with std) = identity */
  const float* std;
  int32_t per_node;
  const float* mean2;  /* device (N) or NULL: second, per-node affine (group-std re-transform) */
  const float* std2;
  float clamp_min;     /* NaN: off */
  float truth_min;     /* NaN: off */
  float min_s;         /* < 0: off */
} matgcn_metric_scale;
int matgcn_metric_sums(const float* pred, const float* y, const int32_t* label_start, int batch, int out_steps, int nodes,
                       int out_dim, int y_steps, int y_feat, int y_start, const matgcn_metric_scale* scale,
                       double* partials, double* sums, int accumulate, void* stream);
/* table (2, out_steps, MATGCN_METRICS) doubles, device: [0] "single" mode (horizon o alone), [1] "average" mode
 * (horizons 0..o), metric order as MATGCN_METRICS above (= TrafficStateEvaluator.json).  swap_r2 != 0: R2 / EVAR with
 * prediction and truth exchanged, the way traffic_state_executor.py:318-319 calls sklearn. */
int matgcn_metric_table(const double* sums, int out_steps, int swap_r2, double* table, void* stream);

/* ---- training step: forward that keeps activations + backward -----------
 * Replaces torch autograd through MultiATGCN.forward as driven by TrafficStateExecutor._train_epoch
 * (libcity/executor/traffic_state_executor.py:411-422: loss = calculate_loss(batch); loss.backward()).
 * matgcn_grads mirrors matgcn_params field by field (same shapes); every non-NULL gradient is OVERWRITTEN.
 * node_emb may be NULL (node_specific_off freezes it); node_vec1/2 are required iff adp_mode == UNI.
 * `train` is one more caller-owned device buffer of matgcn_train_bytes(): forward_train saves z, r, hc of the
 * graph cell, z2, r2, hc2 of the residual cell, the initial state and the graph-mixed rows of every (layer, step) into it, backward uses
 * the rest as scratch.  matgcn_backward must see the SAME workspace and train buffers, untouched, that the matching
 * matgcn_forward_train call used (the sequences of every layer live in the workspace).
 * d_out (B, output_window, N, output_dim) is the gradient of the loss w.r.t. the forward's output.
 * drop_mask: NULL (eval-mode forward), or the training-mode dropout in front of end_conv (MultiATGCN.py:416,
 * F.dropout p = 0.1) as a (B, T, N, H) device tensor of multipliers 0 or 1/(1-p) drawn by the caller's RNG
 * (torch: F.dropout(torch.ones(B,T,N,H))); the same mask goes to the matching matgcn_backward.
 * Reductions that meet in one address use fp32 atomics: gradients are reproducible to rounding, not bitwise.
 * With gcn_off the dense cells' nn.Linear gradients come back in res_gate / res_update (as their weights go in);
 * with fnn_off drop_mask is (B, 1, N, H). */
typedef struct matgcn_agcn_grads {
  float* weights_g;
  float* weights_pool;
  float* bias_pool;
} matgcn_agcn_grads;

typedef struct matgcn_linear_grads {
  float* weight;
  float* bias;
} matgcn_linear_grads;

typedef struct matgcn_grads {
  float* node_emb;
  float* node_vec1;
  float* node_vec2;
  float* static_supports; /* unused: the static supports are constants */
  float* weight_tsg;
  float* weight_ts[MATGCN_MAX_HEADS];
  float* weights_gru;
  matgcn_agcn_grads gate[MATGCN_MAX_LAYERS];
  matgcn_agcn_grads update[MATGCN_MAX_LAYERS];
  matgcn_linear_grads res_gate[MATGCN_MAX_LAYERS];
  matgcn_linear_grads res_update[MATGCN_MAX_LAYERS];
  float* end_conv_weight;
  float* end_conv_bias;
} matgcn_grads;

int matgcn_train_bytes(const matgcn_dims* dims, size_t* bytes);
/* src != NULL: the batch comes from the device-resident series (X is ignored and may be NULL), as in
 * matgcn_forward_series; the matching matgcn_backward must get the same descriptor. */
int matgcn_forward_train(const matgcn_dims* dims, const matgcn_params* params, const void* prepared,
                         const float* X, const matgcn_series* src, const float* h0, const float* drop_mask, float* out,
                         void* workspace, size_t workspace_bytes, void* train, size_t train_bytes, void* stream);
/* h0: the pointer the matching matgcn_forward_train saw (NULL = zero initial state; the backward reads the padded copy
 * forward_train kept in `train`, the pointer only says that there was one).  d_h0: (L, B, N, H) gradient of the loss
 * w.r.t. the initial state, or NULL when the caller does not need it (it is overwritten, not accumulated). */
int matgcn_backward(const matgcn_dims* dims, const matgcn_params* params, const void* prepared, const float* X,
                    const matgcn_series* src, const float* h0, const float* drop_mask, const float* d_out,
                    const matgcn_grads* grads, float* d_h0, void* workspace, size_t workspace_bytes, void* train,
                    size_t train_bytes, void* stream);

/* The one contraction kernel of the backward, exposed for its parity test: a strided, two-level-batched fp32
 * GEMM  C[b1][b2] (+)= alpha * sum_{k2,k} A[b1][b2][m][k2][k] B[b1][b2][k2][k][n].  desc (22 x int64, host):
 * M, N, K, K2, sAm, sAk, sAk2, sBk, sBn, sBk2, sCm, sCn, nb1, nb2, bA1, bA2, bB1, bB2, bC1, bC2,
 * mode (0: C = alpha*acc + beta*C; 1: atomic add into C), split (K ranges per output tile, mode 1). */
int matgcn_debug_gemm(const float* A, const float* B, float* C, const int64_t* desc, float alpha, float beta,
                      void* stream);

/* ---- scheduling option ------------------------------------------------------------------------
 * The encoder runs the recurrent chains of the layers as a wavefront on internal HIP streams (created once, on
 * first use; forked from and joined back into the caller's stream with events, so the caller still sees one
 * in-order stream).  matgcn_set_wavefront(0) serialises everything on the caller's stream instead - same kernels,
 * same results; used to time one kernel alone.  (A lock-step pairing of the chains through per-kernel events was
 * measured and rejected: the cross-stream waits cost more than the pairing gains.)  Returns the previous setting.
 * matgcn_set_wavefront(2) is a lab switch (round 4, measured and rejected: 8.3 against 6.8 ms): the graph mixes of all
 * chains form one global order so that a mix only ever runs beside the other chain's node kernel; same results. */
int matgcn_set_wavefront(int enabled);

/* Hardware queues.  The HIP runtime maps the streams of a process onto GPU_MAX_HW_QUEUES (default 4) hardware queues per
 * stream priority, round robin in creation order, and which of the library's streams share a queue decides how well the
 * layers' chains overlap.  By default the library creates ordinary streams and runs the first chain on the caller's
 * stream: the best measured for a single process - but the pool is shared with every other stream of the process, and an
 * RCCL communicator created before the library's streams moves them onto each other's queues (forward 8.0 instead of
 * 6.8 ms at the headline shape, identical kernel durations: profiles/r04_rccl_queues_lab.log).
 * matgcn_set_stream_pool(1) - for data-parallel jobs - makes the library create its streams at the device's highest
 * priority (a queue pool nothing else in the process uses) at chosen places of the round robin, and run
 * matgcn_forward / _forward_series / _forward_train / _backward on a library stream forked from and joined back into the
 * caller's: the same timings with and without a process group, about 1 % behind the default without one.  Must be called
 * before the first of those entry points creates the streams (MATGCN_ERR_BAD_ARG afterwards, unless the mode asked for is
 * the one in use).  A scheduling change only: the forward's results are bit-identical in both modes, gradients agree up to the
 * order of the backward's atomic accumulations (as two runs in one mode do). */
int matgcn_set_stream_pool(int own);

/* Lazy prepare.  By default `prepared` is complete, in stream order, when matgcn_prepare returns.  With
 * matgcn_set_lazy_prepare(1) matgcn_prepare returns while the node-adaptive weight streams (most of its work) are still
 * being written on two library streams, and every entry point of this library that takes `prepared` orders its stream
 * behind them itself - the wavefront forward per layer: each chain waits for exactly the weights it reads, so the weight
 * preparation runs beside head fusion, the layer-0 fold and the first graph mix of matgcn_forward{,_series}.  A caller
 * that enables it promises (a) to keep `prepared` and the parameter tensors alive and unmodified until a later call of
 * this library on the same stream has been enqueued, and (b) to call matgcn_prepare_join(stream) before it reads or
 * frees `prepared` itself.  Meant for hosts that own `prepared` for the lifetime of the model (the plugin class does).
 * Both return MATGCN_OK / the previous setting. */
int matgcn_set_lazy_prepare(int enabled);
int matgcn_prepare_join(void* stream);

/* matgcn_set_batch_split(2): the inference forwards (matgcn_forward, matgcn_forward_series) run the two halves of an
 * even batch as two independent forwards of B / 2 samples side by side - half 0 on the caller's stream, half 1 on a
 * library stream, each with its own layer wavefront and its own half of the workspace - and join them before the call's
 * last event.  The samples of a batch never interact (MultiATGCN.py:363-420), so the result is that of two forwards of
 * B / 2: the same kernels, the same arithmetic per sample.  Needs the wavefront on, fp32 operands and a workspace that
 * holds two B / 2 plans (matgcn_workspace_bytes of the full batch does); otherwise the plain forward runs.  0 / 1: off
 * (default).  Returns the previous setting. */
int matgcn_set_batch_split(int parts);

/* ---- precision option (BASELINE config 3's dtype; a side line, never the headline) -----------------------------
 * matgcn_set_mix_precision(1): the graph mixes of matgcn_forward / matgcn_forward_series round their operands - the
 * support stack and the state rows - to bf16 on the way into LDS and run on v_mfma_f32_16x16x16_bf16 with fp32
 * accumulation; inputs, outputs, the recurrent state, the node-wise contractions and everything in memory stay fp32.
 * matgcn_set_mix_precision(2): additionally the node-wise contractions of the recurrent step (MultiATGCN.py:108) and,
 * since round 4, of the hoisted x part of layers >= 1 take bf16 operands: the node-adaptive weights are streamed from a
 * bf16 copy of their fragment streams (made once per forward in the workspace - half the bytes of the largest stream of
 * a step), the rows [s | mix(s)] / [x | mix(x)] are rounded on their way into LDS, fp32 accumulation; the state, the
 * layer-0 x part, the residual cell and every epilogue stay fp32.
 * Both are NARROWER than the reference's fp32 arithmetic: measured max-normalised deviation from the fp32 path <= 3e-3
 * (mode 1) at N = 403 (tests/test_hip_parity.py::test_bf16_mix_variant holds both modes to 5e-3).  matgcn_prepare, the
 * training entry points and the unit entry points always use fp32 operands.  Returns the previous setting; 0 (default)
 * = fp32. */
int matgcn_set_mix_precision(int mode);

/* ---- measurement hooks (bench.py; not on the hot path) ---------------------------------------
 * Time individual kernel launches in situ with HIP events recorded on the caller's stream.
 * enable() creates 2*max_launches events (the only allocation in the library, outside any forward)
 * and selects which kernels are bracketed (bit mask of MATGCN_PROF_*); while enabled every selected
 * launch records an event pair until max_launches is reached.  collect() synchronises the recorded
 * events and writes per-launch milliseconds and kernel kinds; disable() destroys the events. */
enum { MATGCN_PROF_MIX = 1,      /* k_mix<1>: the recurrent step's graph mix (the roofline kernel) */
       MATGCN_PROF_GATE = 2, MATGCN_PROF_UPDATE = 4, MATGCN_PROF_RES = 8, MATGCN_PROF_PX = 16,
       MATGCN_PROF_HEAD = 32,
       MATGCN_PROF_MIX_PRE = 64, /* k_mix<0>: pre-pass mixes (layer-0 fold, x-part chunks) */
       MATGCN_PROF_ALL = 127 };
int matgcn_profile_enable(int kind_mask, int max_launches);
int matgcn_profile_collect(float* ms, int* kinds, int capacity, int* count);
int matgcn_profile_disable(void);

#ifdef __cplusplus
}
#endif
#endif /* MATGCN_H */
