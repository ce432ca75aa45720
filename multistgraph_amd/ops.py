"""Host-side handle on the HIP hot path: owns the device buffers and calls the C ABI.

``PathSpec`` describes one model configuration (everything in matgcn_dims except the batch size);
``HotPath`` binds a spec + batch size to a ``prepared`` and a ``workspace`` buffer and exposes one
method per entry point of include/matgcn.h.  Tensors go in and out as torch CUDA tensors; only raw
device pointers and the current HIP stream cross the boundary.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib


@dataclass
class PathSpec:
    nodes: int
    out_window: int
    out_dim: int = 1
    start_dim: int = 0
    in_steps: int = 24
    x_steps: int = 96
    x_feat: int = 2
    hidden: int = 64
    layers: int = 2
    feat_in: int = 2
    embed_dim: int = 20
    adj_rank: int = 20
    adpadj: str = "unidirection"
    adjtype: str = "multi"
    cheb_k: int = 2
    n_static: int = 3
    head_begin: Sequence[int] = (0, 24, 48, 72)
    n_ts: int = 4
    ext_src: Sequence[int] = (1,)
    diag_static_mask: int = 0       # bit s: static support s is a diagonal matrix (folded, never mixed)
    gcn_off: bool = False           # ablation: dense GRU cells instead of graph cells (MultiATGCN.py:177-192)
    fnn_off: bool = False           # ablation: head over the last step only (:342-344,412)

    @property
    def scale_by_g(self) -> bool:
        return self.adjtype == "multi"

    @property
    def n_first(self) -> int:
        return (0 if self.adpadj == "none" else 1) + self.n_static

    @property
    def k_total(self) -> int:
        return 1 + self.n_first * (self.cheb_k - 1)

    def dims(self, batch: int) -> _lib.Dims:
        d = _lib.Dims()
        d.batch, d.nodes, d.in_steps, d.x_steps, d.x_feat = batch, self.nodes, self.in_steps, self.x_steps, self.x_feat
        d.out_channels, d.out_dim, d.start_dim = self.out_window * self.out_dim, self.out_dim, self.start_dim
        d.hidden, d.layers, d.feat_in = self.hidden, self.layers, self.feat_in
        d.embed_dim, d.adj_rank = self.embed_dim, self.adj_rank
        d.adp_mode = _lib.ADP_CODES[self.adpadj]
        d.n_static, d.cheb_k, d.scale_by_g = self.n_static, self.cheb_k, int(self.scale_by_g)
        d.n_heads, d.n_ts = len(self.head_begin), self.n_ts
        d.diag_static_mask = int(self.diag_static_mask)
        d.gcn_off, d.fnn_off = int(self.gcn_off), int(self.fnn_off)
        for i, v in enumerate(self.head_begin):
            d.head_begin[i] = int(v)
        for i, v in enumerate(self.ext_src):
            d.ext_src[i] = int(v)
        return d


def diagonal_mask(static_supports) -> int:
    """Bit s set when static support s (host array / CPU or GPU tensor, (S,N,N)) has no off-diagonal entry.
    Host-side check made once at model construction; such supports are folded into the weights."""
    if static_supports is None:
        return 0
    st = torch.as_tensor(static_supports).detach().cpu()
    mask = 0
    for i in range(st.shape[0]):
        m = st[i]
        if torch.count_nonzero(m - torch.diag(torch.diagonal(m))).item() == 0:
            mask |= 1 << i
    return mask


def spec_from_config(config, data_feature, num_nodes: int, adj_rank: int, n_static: int,
                     diag_static_mask: int = 0, hidden: Optional[int] = None) -> PathSpec:
    """Derive the path description from the reference's config / data_feature keys
    (MultiATGCN.py:224-235, 264-265, 310-332; head windows :371-393).  ``config`` only needs ``get`` (LibCity's
    ConfigParser has no ``keys()``, so it is never copied); ``hidden`` overrides ``rnn_units`` (the padded width the
    kernels run, hidden_pad.py)."""
    out_window = config.get("output_window", 1)
    start_dim, end_dim = config.get("start_dim", 0), config.get("end_dim", 1)
    od = end_dim - start_dim
    tid = 0
    if config.get("add_time_in_day", False):
        tid = 8 if config.get("add_day_in_week", False) else 1
    lc = data_feature.get("len_closeness", 0)
    lp = data_feature.get("len_period", 0)
    lt = data_feature.get("len_trend", 0)
    heads: List[int] = []
    for kk in range(lc // 24):
        heads.append(24 * kk)
    if lp > 0 and out_window >= 6:
        for kk in range(lp // 24):
            heads.append(lc + 24 * kk)
    if lt > 0 and out_window >= 6:
        for kk in range(lt // 24):
            heads.append(lc + lp)  # the reference never advances the trend window (:389-393)
    ext_dim = data_feature.get("ext_dim", 1)
    x_feat = int(data_feature.get("feature_dim", od + ext_dim))
    ext: List[int] = []
    if config.get("add_time_in_day", False):
        ext += [end_dim + j for j in range(tid)]
    if config.get("load_dynamic", False):
        ext += list(range(end_dim + tid, x_feat))
    feat_in = od + ext_dim  # feature_final (:321); must equal od + len(ext) for cat() to line up
    if feat_in != od + len(ext):
        raise ValueError("feature_final=%d but forward() concatenates %d channels" % (feat_in, od + len(ext)))
    node_specific_off = config.get("node_specific_off", False)
    return PathSpec(
        nodes=num_nodes, out_window=out_window, out_dim=od, start_dim=start_dim,
        in_steps=config.get("input_window", 1), x_steps=lc + lp + lt, x_feat=x_feat,
        hidden=config.get("rnn_units", 64) if hidden is None else hidden, layers=config.get("num_layers", 2), feat_in=feat_in,
        embed_dim=1 if node_specific_off else config.get("embed_dim_node", 10), adj_rank=adj_rank,
        adpadj=config.get("adpadj", "bidirection"), adjtype=config.get("adjtype", "od"),
        cheb_k=config.get("cheb_order", 2), n_static=n_static, head_begin=tuple(heads),
        n_ts=int((lp + lt + lc) / 24), ext_src=tuple(ext), diag_static_mask=diag_static_mask,
        gcn_off=bool(config.get("gcn_off", False)), fnn_off=bool(config.get("fnn_off", False)))


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _check_tensor(t: torch.Tensor, name: str, shape=None) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.MatgcnError("%s must live on the GPU: the hot path is HIP-only (no CPU fallback)" % name)
    if t.dtype != torch.float32:
        raise _lib.MatgcnError("%s must be float32, got %s" % (name, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise _lib.MatgcnError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t.contiguous()


def debug_gemm(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, desc, alpha=1.0, beta=0.0) -> None:
    """matgcn_debug_gemm on raw buffers (tests): desc = the 22 integers of include/matgcn.h."""
    lib = _lib.load()
    arr = (C.c_int64 * 22)(*[int(v) for v in desc])
    stream = C.c_void_p(torch.cuda.current_stream(c.device).cuda_stream)
    _lib.check(lib.matgcn_debug_gemm(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr()), arr,
                                     C.c_float(alpha), C.c_float(beta), stream), "matgcn_debug_gemm")


def _label_geometry(pred, y, label_start):
    """(y, label_start, y_steps, y_feat) checked: y is (B, y_steps, N, F) windows, or - with label_start (B) int32 -
    the raw series (T, N, F) whose rows label_start[b] .. +out are sample b's targets"""
    b, out, n, od = pred.shape
    y = _check_tensor(y, "y")
    if label_start is None:
        if y.dim() != 4 or y.shape[0] != b or y.shape[2] != n:
            raise _lib.MatgcnError("y has shape %s, incompatible with pred %s" % (tuple(y.shape), tuple(pred.shape)))
        return y, None, int(y.shape[1]), int(y.shape[3])
    if y.dim() != 3 or y.shape[1] != n:
        raise _lib.MatgcnError("series has shape %s, expected (T, %d, F)" % (tuple(y.shape), n))
    if not label_start.is_cuda or label_start.dtype != torch.int32 or tuple(label_start.shape) != (b,):
        raise _lib.MatgcnError("label_start must be a CUDA int32 tensor of shape (%d,)" % b)
    return y, label_start.contiguous(), int(y.shape[0]), int(y.shape[2])


def _mae_call(pred, y, y_start, mean, std, null_val, min_s, label_start=None):
    lib = _lib.load()
    pred = _check_tensor(pred, "pred")
    b, out, n, od = pred.shape
    y, label_start, y_steps, y_feat = _label_geometry(pred, y, label_start)
    partials = torch.empty(2 * b * out + 1, dtype=torch.float32, device=pred.device)
    result = torch.empty(1 + out, dtype=torch.float32, device=pred.device)
    stream = C.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream)
    _lib.check(lib.matgcn_masked_mae(C.c_void_p(pred.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(_ptr(label_start)),
                                     b, out, n, od, y_steps, y_feat, int(y_start), float(mean), float(std),
                                     float(null_val), float(min_s), C.c_void_p(partials.data_ptr()),
                                     C.c_void_p(result.data_ptr()), stream), "matgcn_masked_mae")
    return pred, y, partials, result, label_start


def masked_mae_device(pred: torch.Tensor, y: torch.Tensor, y_start: int, mean: float, std: float,
                      null_val: float = float("nan"), min_s: float = 1e-4,
                      label_start: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(1 + out,) tensor [masked-MAE over all horizons, MAE@1 .. MAE@out] computed on the device by
    matgcn_masked_mae (de-scale + mask + reduce; reference loss.py:17-29, traffic_state_evaluator.py:87-104).
    With label_start, y is the raw series and the targets are gathered on the device."""
    return _mae_call(pred, y, y_start, mean, std, null_val, min_s, label_start)[3]


class _MaskedMAE(torch.autograd.Function):
    """calculate_loss on the device with its gradient (matgcn_masked_mae / matgcn_masked_mae_grad)."""

    @staticmethod
    def forward(ctx, pred, y, y_start, mean, std, null_val, min_s, label_start):
        pred_c, y_c, partials, result, ls = _mae_call(pred.detach(), y, y_start, mean, std, null_val, min_s, label_start)
        ctx.save_for_backward(pred_c, y_c, partials)
        ctx.label_start = ls
        ctx.args = (int(y_start), float(mean), float(std), float(null_val), float(min_s))
        return result[0].clone()

    @staticmethod
    def backward(ctx, upstream):
        pred, y, partials = ctx.saved_tensors
        y_start, mean, std, null_val, min_s = ctx.args
        b, out, n, od = pred.shape
        ls = ctx.label_start
        y_steps, y_feat = (int(y.shape[1]), int(y.shape[3])) if ls is None else (int(y.shape[0]), int(y.shape[2]))
        d_pred = torch.empty_like(pred)
        up = upstream.detach().to(torch.float32).reshape(1).contiguous()
        stream = C.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream)
        _lib.check(_lib.load().matgcn_masked_mae_grad(
            C.c_void_p(pred.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(_ptr(ls)), b, out, n, od, y_steps, y_feat,
            y_start, mean, std, null_val, min_s, C.c_void_p(partials.data_ptr()), C.c_void_p(up.data_ptr()),
            C.c_void_p(d_pred.data_ptr()), stream), "matgcn_masked_mae_grad")
        return d_pred, None, None, None, None, None, None, None


def masked_mae_loss(pred: torch.Tensor, y: torch.Tensor, y_start: int, mean: float, std: float,
                    null_val: float = float("nan"), min_s: float = 1e-4,
                    label_start: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The calculate_loss scalar with autograd support: the device reduction forward, matgcn_masked_mae_grad backward.
    With label_start, y is the raw series and the targets are gathered on the device."""
    return _MaskedMAE.apply(pred, y, y_start, mean, std, null_val, min_s, label_start)


def _affine_arg(v, n, device, name):
    """scalar or length-N vector -> (float32 device tensor, per_node flag)"""
    t = torch.as_tensor(v, dtype=torch.float32).reshape(-1).to(device)
    if t.numel() not in (1, n):
        raise _lib.MatgcnError("%s has %d entries, expected 1 or %d (one per node)" % (name, t.numel(), n))
    return t.contiguous(), int(t.numel() == n and n > 1)


def metric_sums(pred: torch.Tensor, y: torch.Tensor, y_start: int = 0, mean=None, std=None, group_mean=None,
                group_std=None, clamp_min: float = float("nan"), truth_min: float = float("nan"), min_s: float = 1e-4,
                label_start: Optional[torch.Tensor] = None, sums: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(out, 14) float64 device tensor of the per-horizon sums the evaluator's metrics are ratios of
    (matgcn_metric_sums, include/matgcn.h): de-scale (mean / std: the scaler, scalar or per node; group_mean /
    group_std: the per-node group-std re-transform), clamp, select, zero tiny labels and reduce on the device.
    ``sums``: running sums of earlier batches to add to (several batches = one collect over their concatenation)."""
    lib = _lib.load()
    pred = _check_tensor(pred, "pred")
    b, out, n, od = pred.shape
    y, label_start, y_steps, y_feat = _label_geometry(pred, y, label_start)
    sc = _lib.MetricScale()
    keep = []
    if (mean is None) != (std is None) or (group_mean is None) != (group_std is None):
        raise _lib.MatgcnError("mean / std (and group_mean / group_std) come in pairs")
    if mean is not None:
        m, pn = _affine_arg(mean, n, pred.device, "mean")
        sd, pn2 = _affine_arg(std, n, pred.device, "std")
        if pn != pn2:       # broadcast the scalar one
            m, sd = (m.expand(n).contiguous() if not pn else m), (sd.expand(n).contiguous() if not pn2 else sd)
        keep += [m, sd]
        sc.mean, sc.std, sc.per_node = m.data_ptr(), sd.data_ptr(), int(pn or pn2)
    if group_mean is not None:
        gm, _ = _affine_arg(group_mean, n, pred.device, "group_mean")
        gs, _ = _affine_arg(group_std, n, pred.device, "group_std")
        gm, gs = gm.expand(n).contiguous(), gs.expand(n).contiguous()
        keep += [gm, gs]
        sc.mean2, sc.std2 = gm.data_ptr(), gs.data_ptr()
    sc.clamp_min, sc.truth_min, sc.min_s = float(clamp_min), float(truth_min), float(min_s)
    partials = torch.empty(b * out * _lib.METRIC_SUMS, dtype=torch.float64, device=pred.device)
    accumulate = sums is not None
    if sums is None:
        sums = torch.empty(out, _lib.METRIC_SUMS, dtype=torch.float64, device=pred.device)
    elif tuple(sums.shape) != (out, _lib.METRIC_SUMS) or sums.dtype != torch.float64 or not sums.is_cuda:
        raise _lib.MatgcnError("sums must be a CUDA float64 tensor of shape (%d, %d)" % (out, _lib.METRIC_SUMS))
    stream = C.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream)
    _lib.check(lib.matgcn_metric_sums(C.c_void_p(pred.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(_ptr(label_start)),
                                      b, out, n, od, y_steps, y_feat, int(y_start), C.byref(sc),
                                      C.c_void_p(partials.data_ptr()), C.c_void_p(sums.data_ptr()), int(accumulate),
                                      stream), "matgcn_metric_sums")
    del keep
    return sums


def metric_table(sums: torch.Tensor, swap_r2: bool = False) -> torch.Tensor:
    """(2, out, 10) float64 device tensor from (accumulated) metric sums: [0] the evaluator's "single" mode, [1] its
    "average" mode; metric order _lib.METRICS (matgcn_metric_table)."""
    out = int(sums.shape[0])
    table = torch.empty(2, out, len(_lib.METRICS), dtype=torch.float64, device=sums.device)
    stream = C.c_void_p(torch.cuda.current_stream(sums.device).cuda_stream)
    _lib.check(_lib.load().matgcn_metric_table(C.c_void_p(sums.data_ptr()), out, int(bool(swap_r2)),
                                               C.c_void_p(table.data_ptr()), stream), "matgcn_metric_table")
    return table


class HotPath:
    """One (spec, batch) binding.  Not thread-safe; uses torch's current stream at call time."""

    def __init__(self, spec: PathSpec, batch: int, device: torch.device):
        self.lib = _lib.load()
        self.spec, self.batch, self.device = spec, batch, torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MatgcnError("HotPath needs a GPU device, got %s" % self.device)
        self.dims = spec.dims(batch)
        nbytes = C.c_size_t()
        _lib.check(self.lib.matgcn_prepared_bytes(C.byref(self.dims), C.byref(nbytes)), "matgcn_prepared_bytes")
        self.prepared = torch.empty(nbytes.value // 4, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.matgcn_workspace_bytes(C.byref(self.dims), C.byref(nbytes)), "matgcn_workspace_bytes")
        self.workspace = torch.empty(nbytes.value // 4, dtype=torch.float32, device=self.device)
        self.params = _lib.Params()
        self._static = None
        self._keep: List[torch.Tensor] = []
        self._prepared_ok = False
        self._train = None
        self.train_generation = 0
        self.grad_bucket: Optional[torch.Tensor] = None   # flat buffer the gradients of the LAST backward() are views of

    # ---- plumbing ------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ws(self):
        # every call that uses the workspace invalidates what an earlier forward_train left there for its backward
        self.train_generation += 1
        return C.c_void_p(self.workspace.data_ptr()), C.c_size_t(self.workspace.numel() * 4)

    def _with_workspace(self, call, what: str) -> None:
        """Run ``call(ws_ptr, ws_bytes)``; a workspace that is too small for the CURRENT library mode - precision mode 2
        was switched on after this binding sized it: the bf16 weight-stream copies are counted only while that mode is
        set - is re-sized once and the call repeated."""
        ws, wsb = self._ws()
        status = call(ws, wsb)
        if status == -4:      # MATGCN_ERR_SMALL_BUFFER
            nbytes = C.c_size_t()
            _lib.check(self.lib.matgcn_workspace_bytes(C.byref(self.dims), C.byref(nbytes)), "matgcn_workspace_bytes")
            if nbytes.value > self.workspace.numel() * 4:
                self.workspace = torch.empty(nbytes.value // 4, dtype=torch.float32, device=self.device)
                ws, wsb = self._ws()
                status = call(ws, wsb)
        _lib.check(status, what)

    def bind(self, state: Dict[str, torch.Tensor], static_supports: Optional[torch.Tensor]):
        """Point matgcn_params at the tensors of a reference-named state dict."""
        s, p = self.spec, self.params
        keep: List[torch.Tensor] = []

        def dev(name, shape=None):
            t = _check_tensor(state[name], name, shape)
            keep.append(t)
            return t.data_ptr()

        n, d, h = s.nodes, s.embed_dim, s.hidden
        p.node_emb = dev("node_emb", (n, d))
        if s.adpadj == "unidirection":
            p.node_vec1 = dev("node_vec1", (n, s.adj_rank))
            p.node_vec2 = dev("node_vec2", (s.adj_rank, n))
        else:
            p.node_vec1 = p.node_vec2 = None
        if s.n_static > 0:
            st = _check_tensor(static_supports, "static_supports", (s.n_static, n, n))
            keep.append(st)
            self._static = st
            p.static_supports = st.data_ptr()
        else:
            p.static_supports = None
        p.weight_tsg = dev("weight_tsg", (s.n_ts,))
        for i in range(len(s.head_begin)):
            p.weight_ts[i] = dev("weight_ts.%d" % i, (1, 24, n, s.out_dim))
        p.weights_gru = dev("encoder.weights_gru", (s.layers, s.in_steps))
        kt = s.k_total
        for l in range(s.layers):
            cin = (s.feat_in if l == 0 else h) + h
            if not s.gcn_off:
                for nm, o, dst in (("gate", 2 * h, p.gate), ("update", h, p.update)):
                    pre = "encoder.agru_cells.%d.%s." % (l, nm)
                    dst[l].weights_g = dev(pre + "weights_g", (kt, 1, 1))
                    dst[l].weights_pool = dev(pre + "weights_pool", (d, kt, cin, o))
                    dst[l].bias_pool = dev(pre + "bias_pool", (d, o))
            # gcn_off: encoder.agru_cells hold the dense GRU cells; they travel in the res_* fields
            cells = "encoder.agru_cells" if s.gcn_off else "encoder.res_cells"
            for nm, o, dst in (("gate", 2 * h, p.res_gate), ("update", h, p.res_update)):
                pre = "%s.%d.%s." % (cells, l, nm)
                dst[l].weight = dev(pre + "weight", (o, cin))
                dst[l].bias = dev(pre + "bias", (o,))
        p.end_conv_weight = dev("end_conv.weight",
                                (s.out_window * s.out_dim, 1 if s.fnn_off else s.in_steps, 1, h))
        p.end_conv_bias = dev("end_conv.bias", (s.out_window * s.out_dim,))
        self._keep = keep
        self._prepared_ok = False

    # ---- entry points --------------------------------------------------------------------------
    def prepare(self):
        ws, wsb = self._ws()
        _lib.check(self.lib.matgcn_prepare(C.byref(self.dims), C.byref(self.params),
                                           C.c_void_p(self.prepared.data_ptr()),
                                           C.c_size_t(self.prepared.numel() * 4), ws, wsb, self._stream()),
                   "matgcn_prepare")
        self._prepared_ok = True

    def _need_prepared(self):
        if not self._prepared_ok:
            self.prepare()

    def prepare_join(self):
        """the current stream waits for whatever a lazy matgcn_prepare still writes (include/matgcn.h): needed only by
        code that reads ``self.prepared`` itself instead of through the library"""
        _lib.check(self.lib.matgcn_prepare_join(self._stream()), "matgcn_prepare_join")

    def __del__(self):
        # `prepared` goes back to torch's allocator when this object dies: nothing of the library may still write it.
        # The lazy prepare's events live per DEVICE (the library picks them by hipGetDevice()), and the allocator reuses
        # the block in order of the stream it was allocated on: join on this binding's device and on that stream, and
        # additionally tell the allocator about the joining stream when it is another one (a join issued
        # under another current device / stream context landed on the wrong ones).
        try:
            if self._prepared_ok and torch.cuda.is_available():
                with torch.cuda.device(self.device):
                    cur = torch.cuda.current_stream(self.device)
                    self.lib.matgcn_prepare_join(C.c_void_p(cur.cuda_stream))
                    self.prepared.record_stream(cur)
        except Exception as exc:   # interpreter shutdown: torch may already be gone - say so instead of hiding it
            try:
                import warnings
                warnings.warn("HotPath.__del__: matgcn_prepare_join failed (%r)" % (exc,), RuntimeWarning)
            except Exception:
                pass

    def _source(self, x):
        """The batch of a training step: a windows tensor X (B, x_steps, N, F), or a (series, label_start, rel_steps)
        triple - the raw series (T, N, F) resident on the device, B int32 label starts, the x_steps row offsets.
        Returns (X pointer or None, matgcn_series or None, objects to keep alive)."""
        s = self.spec
        if isinstance(x, torch.Tensor):
            x = _check_tensor(x, "X", (self.batch, s.x_steps, s.nodes, s.x_feat))
            return x.data_ptr(), None, (x,)
        series, label_start, rel_steps = x
        if series.dim() != 3 or tuple(series.shape[1:]) != (s.nodes, s.x_feat):
            raise _lib.MatgcnError("series has shape %s, expected (T, %d, %d)" % (tuple(series.shape), s.nodes, s.x_feat))
        series = _check_tensor(series, "series")
        if not label_start.is_cuda or label_start.dtype != torch.int32 or tuple(label_start.shape) != (self.batch,):
            raise _lib.MatgcnError("label_start must be a CUDA int32 tensor of shape (%d,)" % self.batch)
        rel = [int(v) for v in rel_steps]
        if len(rel) != s.x_steps:
            raise _lib.MatgcnError("rel_steps has %d entries, x_steps is %d" % (len(rel), s.x_steps))
        label_start = label_start.contiguous()
        rel_c = (C.c_int32 * len(rel))(*rel)
        src = _lib.Series(series.data_ptr(), int(series.shape[0]), label_start.data_ptr(), rel_c)
        return None, src, (series, label_start, rel_c)

    def _h0(self, h0):
        """initial encoder state (L, B, N, H) or None = zeros (MultiATGCN.py:405-409)"""
        if h0 is None:
            return None
        s = self.spec
        return _check_tensor(h0, "h0", (s.layers, self.batch, s.nodes, s.hidden))

    def forward(self, x: torch.Tensor, h0: Optional[torch.Tensor] = None) -> torch.Tensor:
        s = self.spec
        x = _check_tensor(x, "X", (self.batch, s.x_steps, s.nodes, s.x_feat))
        h0 = self._h0(h0)
        self._need_prepared()
        out = torch.empty(self.batch, s.out_window, s.nodes, s.out_dim, dtype=torch.float32, device=self.device)
        self._with_workspace(lambda ws, wsb: self.lib.matgcn_forward(
            C.byref(self.dims), C.byref(self.params), C.c_void_p(self.prepared.data_ptr()), C.c_void_p(x.data_ptr()),
            C.c_void_p(_ptr(h0)), C.c_void_p(out.data_ptr()), ws, wsb, self._stream()), "matgcn_forward")
        return out

    # ---- training step ---------------------------------------------------
    def _train_buffer(self) -> torch.Tensor:
        if self._train is None:
            nbytes = C.c_size_t()
            _lib.check(self.lib.matgcn_train_bytes(C.byref(self.dims), C.byref(nbytes)), "matgcn_train_bytes")
            self._train = torch.empty(nbytes.value // 4, dtype=torch.float32, device=self.device)
        return self._train

    def _mask(self, drop_mask):
        if drop_mask is None:
            return None
        s = self.spec
        return _check_tensor(drop_mask, "drop_mask", (self.batch, 1 if s.fnn_off else s.in_steps, s.nodes, s.hidden))

    def forward_train(self, x, drop_mask: Optional[torch.Tensor] = None,
                      h0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """matgcn_forward that keeps the activations the backward needs (in the train buffer and the workspace).
        x: the windows tensor X, or a (series, label_start, rel_steps) triple (see _source).
        drop_mask: the (B, T, N, H) multipliers of the dropout in front of end_conv (training mode) or None;
        h0: initial encoder state (L, B, N, H) or None."""
        s = self.spec
        xp, src, _keep = self._source(x)
        drop_mask = self._mask(drop_mask)
        h0 = self._h0(h0)
        self._need_prepared()
        tr = self._train_buffer()
        out = torch.empty(self.batch, s.out_window, s.nodes, s.out_dim, dtype=torch.float32, device=self.device)
        ws, wsb = self._ws()
        _lib.check(self.lib.matgcn_forward_train(C.byref(self.dims), C.byref(self.params),
                                                 C.c_void_p(self.prepared.data_ptr()), C.c_void_p(xp),
                                                 C.byref(src) if src is not None else None,
                                                 C.c_void_p(_ptr(h0)), C.c_void_p(_ptr(drop_mask)),
                                                 C.c_void_p(out.data_ptr()), ws, wsb, C.c_void_p(tr.data_ptr()),
                                                 C.c_size_t(tr.numel() * 4), self._stream()), "matgcn_forward_train")
        return out

    D_H0 = "__d_h0__"   # key of the initial-state gradient in backward()'s result

    def _grad_views(self, state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Every gradient matgcn_backward returns is a view of ONE flat fp32 buffer (``grad_bucket``, 15.5 MB at
        N = 403), so that the gradient exchange of data-parallel training is a single all-reduce of that buffer with no
        concatenation or copy-back (torch autograd adopts the views as ``p.grad``).  Every call gets a buffer of its
        own from torch's caching allocator - gradients handed out earlier stay valid for as long as somebody holds
        them (gradient accumulation, zero_grad(set_to_none=False), callers that compare two backward results), and
        once they are dropped the allocator hands the same block back, so a training loop reuses one block anyway."""
        names = [k for k in state if not k.startswith("static_initial")]
        offs, total = {}, 0
        for k in names:
            offs[k] = total
            total += (state[k].numel() + 63) // 64 * 64        # every view starts on a 256-byte boundary
        bucket = self.grad_bucket = torch.zeros(total, dtype=torch.float32, device=self.device)
        return {k: bucket[offs[k]:offs[k] + state[k].numel()].view(state[k].shape) for k in names}

    def backward(self, x, d_out: torch.Tensor, state: Dict[str, torch.Tensor],
                 drop_mask: Optional[torch.Tensor] = None, h0: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """Gradients of every tensor of `state` (the dict bind() saw) that the loss depends on, keyed by the same
        names; must directly follow the matching forward_train (same workspace, same train buffer).  With h0 (the
        tensor forward_train saw) the result also holds its gradient (L, B, N, H) under HotPath.D_H0."""
        s = self.spec
        xp, src, _keep = self._source(x)
        d_out = _check_tensor(d_out, "d_out", (self.batch, s.out_window, s.nodes, s.out_dim))
        drop_mask = self._mask(drop_mask)
        h0 = self._h0(h0)
        d_h0 = torch.empty_like(h0) if h0 is not None else None
        grads: Dict[str, torch.Tensor] = {}
        g = _lib.Params()
        views = self._grad_views(state)

        def new(name):
            grads[name] = views[name]
            return views[name].data_ptr()

        if not s.gcn_off:
            if state["node_emb"].requires_grad or not isinstance(state["node_emb"], torch.nn.Parameter):
                g.node_emb = new("node_emb")
            if s.adpadj == "unidirection":
                g.node_vec1, g.node_vec2 = new("node_vec1"), new("node_vec2")
            g.weights_gru = new("encoder.weights_gru")
        g.weight_tsg = new("weight_tsg")
        for i in range(len(s.head_begin)):
            g.weight_ts[i] = new("weight_ts.%d" % i)
        for l in range(s.layers):
            if not s.gcn_off:
                for nm, dst in (("gate", g.gate), ("update", g.update)):
                    pre = "encoder.agru_cells.%d.%s." % (l, nm)
                    dst[l].weights_g = new(pre + "weights_g")
                    dst[l].weights_pool = new(pre + "weights_pool")
                    dst[l].bias_pool = new(pre + "bias_pool")
            # gcn_off: encoder.agru_cells hold the dense GRU cells; they travel in the res_* fields (see bind)
            cells = "encoder.agru_cells" if s.gcn_off else "encoder.res_cells"
            for nm, dst in (("gate", g.res_gate), ("update", g.res_update)):
                pre = "%s.%d.%s." % (cells, l, nm)
                dst[l].weight = new(pre + "weight")
                dst[l].bias = new(pre + "bias")
        g.end_conv_weight = new("end_conv.weight")
        g.end_conv_bias = new("end_conv.bias")
        if not s.scale_by_g and not s.gcn_off:   # the stack is not scaled: weights_g does not reach the output
            for l in range(s.layers):
                for nm in ("gate", "update"):
                    grads["encoder.agru_cells.%d.%s.weights_g" % (l, nm)].zero_()
        for name, t in state.items():    # tensors the forward never reads (unused heads, node_vec* without
            if name.startswith("static_initial"):   # host-side torch layers (MultiATGCN.py:288-290, 336-338): not ours
                continue
            if name not in grads and (t.requires_grad or not isinstance(t, torch.nn.Parameter)):   # adaptive adjacency)
                grads[name] = views[name].zero_()
        tr = self._train_buffer()
        ws, wsb = self._ws()
        _lib.check(self.lib.matgcn_backward(C.byref(self.dims), C.byref(self.params),
                                            C.c_void_p(self.prepared.data_ptr()), C.c_void_p(xp),
                                            C.byref(src) if src is not None else None,
                                            C.c_void_p(_ptr(h0)), C.c_void_p(_ptr(drop_mask)),
                                            C.c_void_p(d_out.data_ptr()), C.byref(g), C.c_void_p(_ptr(d_h0)), ws, wsb,
                                            C.c_void_p(tr.data_ptr()), C.c_size_t(tr.numel() * 4), self._stream()),
                   "matgcn_backward")
        if d_h0 is not None:
            grads[self.D_H0] = d_h0
        return grads

    def forward_series(self, series: torch.Tensor, label_start: torch.Tensor, rel_steps,
                       h0: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Forward fed from the device-resident series (T, N, F): sample b's window rows are gathered by the
        head-fusion prologue from series[label_start[b] + rel_steps[s]] (multistgraph_amd/windows.py)."""
        s = self.spec
        if series.dim() != 3 or tuple(series.shape[1:]) != (s.nodes, s.x_feat):
            raise _lib.MatgcnError("series has shape %s, expected (T, %d, %d)" % (tuple(series.shape), s.nodes, s.x_feat))
        series = _check_tensor(series, "series")
        if not label_start.is_cuda or label_start.dtype != torch.int32 or tuple(label_start.shape) != (self.batch,):
            raise _lib.MatgcnError("label_start must be a CUDA int32 tensor of shape (%d,)" % self.batch)
        rel = [int(v) for v in rel_steps]
        if len(rel) != s.x_steps:
            raise _lib.MatgcnError("rel_steps has %d entries, x_steps is %d" % (len(rel), s.x_steps))
        h0 = self._h0(h0)
        self._need_prepared()
        out = torch.empty(self.batch, s.out_window, s.nodes, s.out_dim, dtype=torch.float32, device=self.device)
        rel_c = (C.c_int32 * len(rel))(*rel)
        self._with_workspace(lambda ws, wsb: self.lib.matgcn_forward_series(
            C.byref(self.dims), C.byref(self.params), C.c_void_p(self.prepared.data_ptr()),
            C.c_void_p(series.data_ptr()), C.c_int64(series.shape[0]), C.c_void_p(label_start.contiguous().data_ptr()),
            rel_c, C.c_void_p(_ptr(h0)), C.c_void_p(out.data_ptr()), ws, wsb, self._stream()), "matgcn_forward_series")
        return out

    def fuse_heads(self, x: torch.Tensor) -> torch.Tensor:
        s = self.spec
        x = _check_tensor(x, "X", (self.batch, s.x_steps, s.nodes, s.x_feat))
        out = torch.empty(self.batch, s.in_steps, s.nodes, s.feat_in, dtype=torch.float32, device=self.device)
        ws, wsb = self._ws()
        _lib.check(self.lib.matgcn_fuse_heads(C.byref(self.dims), C.byref(self.params), C.c_void_p(x.data_ptr()),
                                              C.c_void_p(out.data_ptr()), ws, wsb, self._stream()),
                   "matgcn_fuse_heads")
        return out

    def _single(self, fn, name, layer, x, h, out_cols):
        s = self.spec
        cl = s.feat_in if layer == 0 else s.hidden
        x = _check_tensor(x, "x", (self.batch, s.nodes, cl))
        h = _check_tensor(h, "h", (self.batch, s.nodes, s.hidden))
        self._need_prepared()
        out = torch.empty(self.batch, s.nodes, out_cols, dtype=torch.float32, device=self.device)
        ws, wsb = self._ws()
        _lib.check(fn(C.byref(self.dims), C.byref(self.params), C.c_void_p(self.prepared.data_ptr()), layer,
                      C.c_void_p(x.data_ptr()), C.c_void_p(h.data_ptr()), C.c_void_p(out.data_ptr()), ws, wsb,
                      self._stream()), name)
        return out

    def agcn_gate(self, layer, x, h):
        return self._single(self.lib.matgcn_agcn_gate_fwd, "matgcn_agcn_gate_fwd", layer, x, h, 2 * self.spec.hidden)

    def atgru_cell(self, layer, x, h):
        return self._single(self.lib.matgcn_atgru_cell_fwd, "matgcn_atgru_cell_fwd", layer, x, h, self.spec.hidden)

    def res_cell(self, layer, x, h):
        return self._single(self.lib.matgcn_res_cell_fwd, "matgcn_res_cell_fwd", layer, x, h, self.spec.hidden)

    def encoder(self, x0: torch.Tensor, h0: Optional[torch.Tensor] = None):
        s = self.spec
        x0 = _check_tensor(x0, "x0", (self.batch, s.in_steps, s.nodes, s.feat_in))
        if h0 is not None:
            h0 = _check_tensor(h0, "h0", (s.layers, self.batch, s.nodes, s.hidden))
        self._need_prepared()
        seq = torch.empty(self.batch, s.in_steps, s.nodes, s.hidden, dtype=torch.float32, device=self.device)
        fin = torch.empty(s.layers, self.batch, s.nodes, s.hidden, dtype=torch.float32, device=self.device)
        ws, wsb = self._ws()
        _lib.check(self.lib.matgcn_encoder_fwd(C.byref(self.dims), C.byref(self.params),
                                               C.c_void_p(self.prepared.data_ptr()), C.c_void_p(x0.data_ptr()),
                                               C.c_void_p(_ptr(h0)), C.c_void_p(seq.data_ptr()),
                                               C.c_void_p(fin.data_ptr()), ws, wsb, self._stream()),
                   "matgcn_encoder_fwd")
        return seq, fin

    def output_head(self, seq: torch.Tensor) -> torch.Tensor:
        s = self.spec
        seq = _check_tensor(seq, "seq", (self.batch, s.in_steps, s.nodes, s.hidden))
        self._need_prepared()
        out = torch.empty(self.batch, s.out_window, s.nodes, s.out_dim, dtype=torch.float32, device=self.device)
        ws, wsb = self._ws()
        _lib.check(self.lib.matgcn_output_head(C.byref(self.dims), C.byref(self.params),
                                               C.c_void_p(self.prepared.data_ptr()), C.c_void_p(seq.data_ptr()),
                                               C.c_void_p(out.data_ptr()), ws, wsb, self._stream()),
                   "matgcn_output_head")
        return out

    def supports(self) -> torch.Tensor:
        """(Ks, N, N): the DENSE non-identity slots of the support stack as matgcn_prepare built them, in stack order
        (tests).  Diagonal supports never reach the stack - they are folded into the identity slot of the node-adaptive
        weights (read those back with node_weights) -, and cheb_order = 1 holds the SUM of its dense supports."""
        self._need_prepared()
        self.prepare_join()
        lay = (C.c_int64 * 4)()
        _lib.check(self.lib.matgcn_supports_layout(C.byref(self.dims), C.byref(lay)), "matgcn_supports_layout")
        off, ld, npad, ks = (int(v) for v in lay)
        n = self.spec.nodes
        if ks == 0:
            return torch.empty(0, n, n, dtype=torch.float32, device=self.device)
        st = self.prepared[off:off + npad * ld].view(npad, ld)
        return torch.stack([st[:n, k * npad:k * npad + n].t() for k in range(ks)], 0).contiguous()

    def node_weights(self, layer: int, part: int) -> torch.Tensor:
        """(N, 1 + Ks, 64, O): the recurrent (hidden-channel) rows of the node-adaptive weights of
        agru_cells[layer].gate (part 0, O = 128) / .update (part 1, O = 64), decoded from the MFMA-fragment-ordered
        stream matgcn_prepare wrote into `prepared` (tests: softmax(weights_g) and the diagonal-support fold included)."""
        self._need_prepared()
        self.prepare_join()
        lay = (C.c_int64 * 4)()
        _lib.check(self.lib.matgcn_weights_layout(C.byref(self.dims), layer, part, C.byref(lay)),
                   "matgcn_weights_layout")
        off, stride, groups, ot = (int(v) for v in lay)
        n, o_dim = self.spec.nodes, ot * 16
        idx = torch.arange(n, device=self.device)[:, None] * stride + off + \
            torch.arange(groups * ot * 256, device=self.device)[None, :]
        frag = self.prepared[idx].view(n, groups, ot, 4, 16, 4)      # [n][g][ct][kq][j][s]
        # row kk = 16 g + 4 kq + s, column o = 16 ct + j
        w = frag.permute(0, 1, 3, 5, 2, 4).reshape(n, groups * 16, o_dim)
        return w.view(n, groups // 4, 64, o_dim).contiguous()
