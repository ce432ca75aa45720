"""CPU oracle for the Multi-ATGCN forward path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  ``multistgraph_amd`` never does.

It restates, in plain functional torch-on-CPU (fp32, or fp64 when ``dtype=torch.float64``), what
``/root/reference/libcity/model/traffic_flow_prediction/MultiATGCN.py`` computes; every function
cites the reference lines it follows.  Parity is PINNED: ``tests/golden/*.npz`` were produced by
importing the reference model itself in the build container (``tests/golden/make_golden.py``) and
``tests/test_oracle_golden.py`` checks this restatement against them (<= 1e-5 max-normalised).

Two evaluation orders are provided:
  * ``faithful=True``  - rebuilds the support stack and the node-adaptive weights inside every
    AGCN call exactly like the reference (MultiATGCN.py:76-109).  This is "the reference CPU
    path" timed by bench.py's cpu_baseline.
  * ``faithful=False`` - hoists them out of the time loop (same arithmetic per element).

Parameters are passed as a flat ``dict`` keyed by the reference's state_dict names
(SURVEY.md section 8b).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# one-off host graph preparation (reference MultiATGCN.py:15-56, 238-283)
# ----------------------------------------------------------------------------------------------
def scaled_laplacian(adj: np.ndarray) -> np.ndarray:
    """L~ = 2 L / lambda_max - I with lambda_max = 2, L = I - D^-1/2 A^T D^-1/2.

    Follows calculate_normalized_laplacian / calculate_scaled_laplacian (MultiATGCN.py:15-38):
    d = row sums of A, zero-degree rows give 0, directed graph (no symmetrisation), so the result
    is simply  -D^-1/2 A^T D^-1/2  (+ exact zeros from the I - I cancellation).
    """
    a = np.asarray(adj, dtype=np.float32)
    deg = a.sum(axis=1)
    with np.errstate(divide="ignore"):
        dis = np.power(deg, -0.5)
    dis[np.isinf(dis)] = 0.0
    # (A . diag(dis))^T . diag(dis)  ==  diag(dis) A^T diag(dis)
    norm = (a * dis[None, :]).T * dis[None, :]
    n = a.shape[0]
    lap = np.eye(n, dtype=np.float64) - norm.astype(np.float64)
    return (lap - np.eye(n, dtype=np.float64)).astype(np.float32)


def od_adjacency(adj_mx: np.ndarray) -> np.ndarray:
    """A / diag(A) broadcast over the LAST axis (column j divided by A[j,j]), clamped to <= 1
    (MultiATGCN.py:238-241)."""
    a = torch.as_tensor(np.asarray(adj_mx), dtype=torch.float32)
    a = a / torch.diagonal(a)
    a = torch.where(a > 1, torch.ones_like(a), a)
    return a.numpy()


def parse_coordinates(coordinate) -> np.ndarray:
    """(N, 2) [lon, lat] ordered by geo_id, from the "[lon, lat]" strings (MultiATGCN.py:253-255).
    The reference's pivot on geo_id sorts the ids (``:260``), hence the argsort."""
    ids = np.asarray(coordinate["geo_id"])
    lonlat = np.array([[float(v) for v in s.strip()[1:-1].split(",")]
                       for s in coordinate["coordinates"]], dtype=np.float64)
    return lonlat[np.argsort(ids, kind="stable")]


def distance_adjacency(lonlat: np.ndarray, eps: float = 0.1) -> np.ndarray:
    """Haversine km -> exp(-(d/std)^2), entries < eps zeroed (MultiATGCN.py:41-56, 259-261)."""
    lon = np.radians(lonlat[:, 0])
    lat = np.radians(lonlat[:, 1])
    dlat = lat[None, :] - lat[:, None]
    dlon = lon[None, :] - lon[:, None]
    s = np.sin(dlat * 0.5) ** 2 + np.cos(lat[:, None]) * np.cos(lat[None, :]) * np.sin(dlon * 0.5) ** 2
    dist = 2.0 * 6371.0 * np.arcsin(np.sqrt(s))
    std = dist[~np.isinf(dist)].std()
    w = np.exp(-np.square(dist / std))
    w[w < eps] = 0.0
    return w.astype(np.float32)


def similarity_adjacency(static: Optional[np.ndarray], n: int) -> np.ndarray:
    """1 / euclid(static_i, static_j) with zero distances -> 1; identity when no static table
    (MultiATGCN.py:244-250)."""
    if static is None:
        return np.eye(n, dtype=np.float32)
    s = np.asarray(static, dtype=np.float64)
    diff = s[:, None, :] - s[None, :, :]
    euc = np.sqrt((diff * diff).sum(-1))
    euc[euc == 0] = 1.0
    return (1.0 / euc).astype(np.float32)


def static_supports(adj_mx, coordinate, static, adjtype: str) -> List[np.ndarray]:
    """The non-identity static supports in stack order (MultiATGCN.py:264-283)."""
    n = np.asarray(adj_mx).shape[0]
    if adjtype == "identity":
        return [np.eye(n, dtype=np.float32)]
    od = od_adjacency(adj_mx)
    if adjtype == "od":
        return [scaled_laplacian(od)]
    dis = distance_adjacency(parse_coordinates(coordinate))
    if adjtype == "dist":
        return [scaled_laplacian(dis)]
    cos = similarity_adjacency(static, n)
    if adjtype == "cosine":
        return [scaled_laplacian(cos)]
    if adjtype == "multi":
        return [scaled_laplacian(od), scaled_laplacian(dis), scaled_laplacian(cos)]
    raise ValueError("unknown adjtype %r" % adjtype)


# ----------------------------------------------------------------------------------------------
# AGCN (reference MultiATGCN.py:76-109)
# ----------------------------------------------------------------------------------------------
def adaptive_adjacency(p: Dict[str, Tensor], adpadj: str) -> Optional[Tensor]:
    """softmax_row(relu(E1 E2)) or softmax_row(relu(E E^T)) (MultiATGCN.py:80-83)."""
    if adpadj == "unidirection":
        return F.softmax(F.relu(p["node_vec1"] @ p["node_vec2"]), dim=1)
    if adpadj == "bidirection":
        return F.softmax(F.relu(p["node_emb"] @ p["node_emb"].T), dim=1)
    if adpadj == "none":
        return None
    raise ValueError("unknown adpadj %r" % adpadj)


def support_stack(p: Dict[str, Tensor], statics: Sequence[Tensor], adjtype: str, adpadj: str,
                  cheb_k: int, weights_g: Optional[Tensor]) -> Tensor:
    """(K, N, N): [I, chebyshev terms of each support ...], optionally scaled per slot by
    softmax(weights_g) when adjtype == 'multi' (MultiATGCN.py:84-103).  Note the reference only
    keeps the static supports next to the adaptive one in 'multi' mode (``:90-93``)."""
    n = p["node_emb"].shape[0]
    eye = torch.eye(n, dtype=p["node_emb"].dtype)
    adp = adaptive_adjacency(p, adpadj)
    if adp is None:
        firsts = list(statics)
    elif adjtype == "multi":
        firsts = [adp] + list(statics)
    else:
        firsts = [adp]
    slots = [eye]
    for s in firsts:
        terms = [eye, s]
        for _ in range(2, cheb_k):
            terms.append(torch.matmul(2 * terms[1], terms[-1]) - terms[-2])
        slots.extend(terms[1:])
    stack = torch.stack(slots, dim=0)
    if adjtype == "multi" and weights_g is not None:
        stack = F.softmax(weights_g, dim=0) * stack
    return stack


def agcn(x: Tensor, p: Dict[str, Tensor], prefix: str, statics, adjtype, adpadj, cheb_k,
         hoisted=None) -> Tensor:
    """x (B, N, I) -> (B, N, O): graph mix with every support, then the node-adaptive contraction
    (MultiATGCN.py:104-109).  ``hoisted`` = (stack, weights, bias) skips the rebuild."""
    if hoisted is None:
        stack = support_stack(p, statics, adjtype, adpadj, cheb_k, p[prefix + "weights_g"])
        weights = torch.einsum("nd,dkio->nkio", p["node_emb"], p[prefix + "weights_pool"])
        bias = p["node_emb"] @ p[prefix + "bias_pool"]
    else:
        stack, weights, bias = hoisted
    mixed = torch.einsum("knm,bmc->bknc", stack, x).permute(0, 2, 1, 3)
    return torch.einsum("bnki,nkio->bno", mixed, weights) + bias


# ----------------------------------------------------------------------------------------------
# recurrent cells (reference MultiATGCN.py:120-128, 142-150)
# ----------------------------------------------------------------------------------------------
def atgru_cell(x: Tensor, h: Tensor, p, prefix: str, statics, adjtype, adpadj, cheb_k,
               hoisted=None) -> Tensor:
    """Graph GRU step as written in the reference: z gates the candidate input, r blends."""
    hid = h.shape[-1]
    hg = None if hoisted is None else hoisted["gate"]
    hu = None if hoisted is None else hoisted["update"]
    zr = torch.sigmoid(agcn(torch.cat((x, h), -1), p, prefix + "gate.", statics, adjtype, adpadj,
                            cheb_k, hg))
    z, r = zr[..., :hid], zr[..., hid:]
    hc = torch.tanh(agcn(torch.cat((x, z * h), -1), p, prefix + "update.", statics, adjtype,
                         adpadj, cheb_k, hu))
    return r * h + (1 - r) * hc


def dense_gru_cell(x: Tensor, h: Tensor, p, prefix: str) -> Tensor:
    """The residual GRU (two nn.Linear) with the same gate algebra (MultiATGCN.py:142-150)."""
    hid = h.shape[-1]
    zr = torch.sigmoid(F.linear(torch.cat((x, h), -1), p[prefix + "gate.weight"], p[prefix + "gate.bias"]))
    z, r = zr[..., :hid], zr[..., hid:]
    hc = torch.tanh(F.linear(torch.cat((x, z * h), -1), p[prefix + "update.weight"],
                             p[prefix + "update.bias"]))
    return r * h + (1 - r) * hc


# ----------------------------------------------------------------------------------------------
# encoder, head fusion, output head (reference MultiATGCN.py:194-212, 363-420)
# ----------------------------------------------------------------------------------------------
def _hoist(p, statics, adjtype, adpadj, cheb_k, layers):
    out = []
    for l in range(layers):
        entry = {}
        for nm in ("gate", "update"):
            pre = "encoder.agru_cells.%d.%s." % (l, nm)
            stack = support_stack(p, statics, adjtype, adpadj, cheb_k, p[pre + "weights_g"])
            w = torch.einsum("nd,dkio->nkio", p["node_emb"], p[pre + "weights_pool"])
            b = p["node_emb"] @ p[pre + "bias_pool"]
            entry[nm] = (stack, w, b)
        out.append(entry)
    return out


def encoder(x: Tensor, init: Tensor, p, statics, adjtype, adpadj, cheb_k, layers,
            faithful: bool = True, gcn_off: bool = False):
    """Layer-major, then time; per-(l,t) scalar residual blend sigma(weights_gru[l,t])
    (MultiATGCN.py:194-212).  Returns (sequence of the last layer (B,T,N,H), final states)."""
    steps = x.shape[1]
    gates = torch.sigmoid(p["encoder.weights_gru"])
    hoisted = None if (faithful or gcn_off) else _hoist(p, statics, adjtype, adpadj, cheb_k, layers)
    cur = x
    finals = []
    for l in range(layers):
        h = init[l]
        seq = []
        for t in range(steps):
            xt = cur[:, t]
            if gcn_off:
                # agru_cells then hold plain GRU cells (MultiATGCN.py:187-192)
                h = dense_gru_cell(xt, h, p, "encoder.agru_cells.%d." % l)
            else:
                h = atgru_cell(xt, h, p, "encoder.agru_cells.%d." % l, statics, adjtype, adpadj,
                               cheb_k, None if hoisted is None else hoisted[l])
                res = dense_gru_cell(xt, h, p, "encoder.res_cells.%d." % l)
                g = gates[l][t]
                h = g * h + (1 - g) * res
            seq.append(h)
        finals.append(h)
        cur = torch.stack(seq, dim=1)
    return cur, finals


def fuse_heads(xb: Tensor, p, cfg) -> Tensor:
    """Temporal-head fusion + concat of time-of-day / dynamic channels (MultiATGCN.py:365-402).

    ``cfg`` keys: start_dim, end_dim, len_closeness, len_period, len_trend (hours), output_window,
    input_window, add_time_in_day, add_day_in_week, load_dynamic.
    The trend loop never advances its window (``:389-393``) - restated as written.
    """
    s0, s1 = cfg.get("start_dim", 0), cfg.get("end_dim", 1)
    src = xb[..., s0:s1]
    gate = F.softmax(p["weight_tsg"], dim=0)
    lc, lp, lt = cfg["len_closeness"], cfg["len_period"], cfg["len_trend"]
    outw = cfg["output_window"]
    acc = 0.0
    slot = 0
    for kk in range(lc // 24):
        acc = acc + gate[slot] * src[:, 24 * kk:24 * kk + 24] * p["weight_ts.%d" % slot]
        slot += 1
    if lp > 0 and outw >= 6:
        for kk in range(lp // 24):
            b0 = lc + 24 * kk
            acc = acc + gate[slot] * src[:, b0:b0 + 24] * p["weight_ts.%d" % slot]
            slot += 1
    if lt > 0 and outw >= 6:
        for kk in range(lt // 24):
            b0 = lc + lp
            acc = acc + gate[slot] * src[:, b0:b0 + 24] * p["weight_ts.%d" % slot]
            slot += 1
    tid = 0
    if cfg.get("add_time_in_day", False):
        tid = 8 if cfg.get("add_day_in_week", False) else 1
    win = cfg["input_window"]
    out = acc
    if cfg.get("add_time_in_day", False):
        out = torch.cat((out, xb[:, 0:win, :, s1:s1 + tid]), dim=-1)
    if cfg.get("load_dynamic", False):
        out = torch.cat((out, xb[:, 0:win, :, s1 + tid:]), dim=-1)
    return out


def output_head(seq: Tensor, p, out_window: int, out_dim: int) -> Tensor:
    """Conv2d(T -> out*od, kernel (1,H)) == contraction over (t,h), then the reshape/permute to
    (B, out, N, od) (MultiATGCN.py:416-418; dropout is identity in eval)."""
    b, t, n, h = seq.shape
    w = p["end_conv.weight"].reshape(-1, t * h)              # (out*od, T*H)
    flat = seq.permute(0, 2, 1, 3).reshape(b, n, t * h)      # (B, N, T*H)
    conv = flat @ w.T + p["end_conv.bias"]                   # (B, N, out*od)
    return conv.permute(0, 2, 1).reshape(b, out_window, out_dim, n).permute(0, 1, 3, 2)


def static_initial_state(static: Tensor, v: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """(N, H) initial state from static features: relu(Linear(static @ v)) with v the PCA basis the reference draws with
    torch.pca_lowrank inside forward (MultiATGCN.py:336-338, 406-408).  The PCA itself is randomised in the reference;
    the fixtures record the v it drew (tests/golden/make_golden.py)."""
    return F.relu(F.linear(static @ v, p["static_initial_gru.embd.weight"], p["static_initial_gru.embd.bias"]))


def forward(xb: Tensor, p: Dict[str, Tensor], statics: Sequence[Tensor], cfg: dict,
            faithful: bool = True, return_stages: bool = False, h0: Optional[Tensor] = None):
    """MultiATGCN.forward (MultiATGCN.py:363-420).  h0 (N, H): the static-feature initial state of every layer and
    sample (``init_state = static_embedding.expand(L, B, -1, -1)``, :409); None = the zero state of init_hidden."""
    adjtype, adpadj = cfg["adjtype"], cfg["adpadj"]
    cheb_k, layers, hid = cfg.get("cheb_order", 2), cfg.get("num_layers", 2), cfg.get("rnn_units", 64)
    x0 = fuse_heads(xb, p, cfg)
    bsz, _, n, _ = x0.shape
    init = torch.zeros(layers, bsz, n, hid, dtype=x0.dtype)
    if h0 is not None:
        init = h0.expand(layers, bsz, -1, -1)
    seq, finals = encoder(x0, init, p, statics, adjtype, adpadj, cheb_k, layers, faithful,
                          cfg.get("gcn_off", False))
    if cfg.get("fnn_off", False):
        seq = seq[:, -1:]
    od = cfg.get("end_dim", 1) - cfg.get("start_dim", 0)
    y = output_head(seq, p, cfg["output_window"], od)
    if return_stages:
        return y, {"x0": x0, "seq": seq, "finals": torch.stack(finals, 0)}
    return y


# ----------------------------------------------------------------------------------------------
# loss / metric epilogue (reference loss.py:17-29, traffic_state_evaluator.py:87-104)
# ----------------------------------------------------------------------------------------------
def masked_mae(pred: Tensor, label: Tensor, null_val=float("nan"), min_s: float = 1e-4) -> Tensor:
    """masked_mae_torch: labels with |y| < min_s are zeroed (the reference does it in place),
    mask = y != null_val (or not-NaN), mean-normalised mask, NaN -> 0."""
    label = torch.where(label.abs() < min_s, torch.zeros_like(label), label)
    if isinstance(null_val, float) and math.isnan(null_val):
        mask = ~torch.isnan(label)
    else:
        mask = label.ne(null_val)
    mask = mask.to(pred.dtype)
    mask = mask / mask.mean()
    mask = torch.where(torch.isnan(mask), torch.zeros_like(mask), mask)
    loss = (pred - label).abs() * mask
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)
    return loss.mean()


def calculate_loss(xb: Tensor, yb: Tensor, p, statics, cfg, mean=0.0, std=1.0, faithful=True, h0=None):
    """MultiATGCN.calculate_loss (MultiATGCN.py:422-427): de-scale both, masked MAE, null 0."""
    s0, s1 = cfg.get("start_dim", 0), cfg.get("end_dim", 1)
    pred = forward(xb, p, statics, cfg, faithful, h0=h0) * std + mean
    true = yb[..., s0:s1] * std + mean
    return masked_mae(pred, true, 0.0)


def horizon_mae(pred: Tensor, true: Tensor, horizon: int) -> Tensor:
    """MAE@horizon in evaluator 'single' mode (traffic_state_evaluator.py:102-104)."""
    return masked_mae(pred[:, horizon - 1], true[:, horizon - 1])


# ----------------------------------------------------------------------------------------------
# helpers for tests / bench
# ----------------------------------------------------------------------------------------------
def to_tensors(state: Dict[str, np.ndarray], dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in state.items()}


def supports_as_tensors(mats: Sequence[np.ndarray], dtype=torch.float32) -> List[Tensor]:
    return [torch.as_tensor(np.asarray(m)).to(dtype) for m in mats]


# ----------------------------------------------------------------------------------------------
# evaluator metrics: TrafficStateEvaluator.collect and the group-std re-transform
# ----------------------------------------------------------------------------------------------
EVAL_METRICS = ("MAE", "MAPE", "MSE", "RMSE", "masked_MAE", "masked_MAPE", "masked_MSE", "masked_RMSE", "R2", "EVAR")


def _masked_mean(loss: Tensor, label: Tensor, null_val) -> Tensor:
    """mask = label != null_val (not-NaN for a NaN null value), mask /= mean(mask), NaN -> 0, mean (loss.py:19-29)."""
    if isinstance(null_val, float) and math.isnan(null_val):
        mask = ~torch.isnan(label)
    else:
        mask = label.ne(null_val)
    mask = mask.to(loss.dtype)
    mask = mask / mask.mean()
    mask = torch.where(torch.isnan(mask), torch.zeros_like(mask), mask)
    loss = loss * mask
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)
    return loss.mean()


def _zero_small(label: Tensor, min_s: float) -> Tensor:
    """labels[|labels| < min_s] = 0 (loss.py:18,53,71,85 - in place in the reference, so every later metric of the same
    collect() sees the zeroed labels)."""
    return torch.where(label.abs() < min_s, torch.zeros_like(label), label)


def masked_mse(pred: Tensor, label: Tensor, null_val=float("nan"), min_s: float = 1e-4) -> Tensor:
    """masked_mse_torch (loss.py:70-82)"""
    label = _zero_small(label, min_s)
    return _masked_mean(torch.square(pred - label), label, null_val)


def masked_rmse(pred: Tensor, label: Tensor, null_val=float("nan"), min_s: float = 1e-4) -> Tensor:
    """masked_rmse_torch (loss.py:85-88): zeroes with min_s, then masked_mse_torch with ITS default min_s = 1e-4"""
    return torch.sqrt(masked_mse(pred, _zero_small(label, min_s), null_val))


def masked_mape(pred: Tensor, label: Tensor, null_val=float("nan"), min_s: float = 1e-4) -> Tensor:
    """masked_mape_torch with eps = 0 (loss.py:52-67): |(p-l)/l|; a zero label gives inf (kept) or, with p = l, NaN (-> 0)"""
    label = _zero_small(label, min_s)
    return _masked_mean(torch.abs((pred - label) / label), label, null_val)


def r2_score(truth: Tensor, pred: Tensor) -> float:
    """sklearn.metrics.r2_score(y_true, y_pred) on the flattened values (loss.py:91-94): 1 - SS_res / SS_tot, with
    sklearn's conventions for a constant truth (1 if the residual is zero too, else 0)."""
    t, p = truth.double().flatten(), pred.double().flatten()
    res = float(((t - p) ** 2).sum())
    tot = float(((t - t.mean()) ** 2).sum())
    return 1.0 - res / tot if tot != 0.0 else (1.0 if res == 0.0 else 0.0)


def explained_variance(truth: Tensor, pred: Tensor) -> float:
    """sklearn.metrics.explained_variance_score(y_true, y_pred) (loss.py:97-100): 1 - Var(t - p) / Var(t)"""
    t, p = truth.double().flatten(), pred.double().flatten()
    num = float(((t - p) - (t - p).mean()).pow(2).mean())
    den = float((t - t.mean()).pow(2).mean())
    return 1.0 - num / den if den != 0.0 else (1.0 if num == 0.0 else 0.0)


def evaluator_table(y_pred: Tensor, y_true: Tensor, mode: str = "single", min_s: float = 1e-4) -> Dict[str, float]:
    """TrafficStateEvaluator.collect + evaluate for ONE collected batch (traffic_state_evaluator.py:46-131): every
    metric of EVAL_METRICS at every horizon i, over y[:, i-1] ("single") or y[:, :i] ("average").  The masked_*
    metrics use null value 0, the plain ones NaN; R2 / EVAR see the labels the loss functions before them zeroed in
    place (the order of TrafficStateEvaluator.json)."""
    out: Dict[str, float] = {}
    for i in range(1, y_true.shape[1] + 1):
        sl = slice(i - 1, i) if mode == "single" else slice(0, i)
        p, t = y_pred[:, sl], y_true[:, sl]
        tz = _zero_small(t, min_s)
        vals = {"MAE": masked_mae(p, t, float("nan"), min_s), "MAPE": masked_mape(p, t, float("nan"), min_s),
                "MSE": masked_mse(p, t, float("nan"), min_s), "RMSE": masked_rmse(p, t, float("nan"), min_s),
                "masked_MAE": masked_mae(p, t, 0.0, min_s), "masked_MAPE": masked_mape(p, t, 0.0, min_s),
                "masked_MSE": masked_mse(p, t, 0.0, min_s), "masked_RMSE": masked_rmse(p, t, 0.0, min_s),
                "R2": r2_score(tz, p), "EVAR": explained_variance(tz, p)}
        for k, v in vals.items():
            out["%s@%d" % (k, i)] = float(v)
    return out


def groupstd_table(y_pred: Tensor, y_true: Tensor, all_m, all_std, s_small: float = 10.0) -> Dict[str, List[float]]:
    """The re-transform table of TrafficStateExecutor.evaluate (traffic_state_executor.py:293-322) on de-scaled
    (B, out, N, od) values: x_t = x * All_std[n] + All_m[n] (:307-308), prediction_t < 0 -> 0 (:312), per ahead step
    the elements with truth_t > s_small (:316-317), then MAE / MSE / RMSE (the *_np functions, NaN null value: plain
    means), r2_score(pr, tr) and explained_variance_score(pr, tr) - prediction FIRST, as the reference writes it
    (:318-319) - and MAPE."""
    m = torch.as_tensor(all_m, dtype=y_pred.dtype).reshape(1, 1, -1, 1)
    s = torch.as_tensor(all_std, dtype=y_pred.dtype).reshape(1, 1, -1, 1)
    pt = y_pred * s + m
    tt = y_true * s + m
    pt = torch.where(pt < 0, torch.zeros_like(pt), pt)
    cols: Dict[str, List[float]] = {k: [] for k in ("MAE", "MSE", "RMSE", "R2", "EVAR", "MAPE")}
    for rr in range(y_pred.shape[1]):
        keep = tt[:, rr] > s_small
        pr, tr = pt[:, rr][keep], tt[:, rr][keep]
        cols["MAE"].append(float((pr - tr).abs().mean()))
        cols["MSE"].append(float(torch.square(pr - tr).mean()))
        cols["RMSE"].append(float(torch.sqrt(torch.square(pr - tr).mean())))
        cols["R2"].append(r2_score(pr, tr))
        cols["EVAR"].append(explained_variance(pr, tr))
        cols["MAPE"].append(float(((pr - tr) / tr).abs().mean()))
    return cols
