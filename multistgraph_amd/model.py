"""Drop-in ``MultiATGCN`` plugin whose forward runs on the HIP hot path (libmatgcn.so).

Mirrors the reference's model-plugin surface for this path - same class name, constructor
signature, config / data_feature keys, parameter names and shapes (= checkpoint ABI), and the
``forward / predict / calculate_loss`` methods the LibCity executor calls
(reference libcity/model/traffic_flow_prediction/MultiATGCN.py:221-430,
libcity/model/abstract_traffic_state_model.py:4-29).  The arithmetic of ``forward`` is NOT here:
it is one call into the C ABI (include/matgcn.h); this module only owns parameters and plumbing.

There is no CPU or PyTorch fallback: a CPU tensor, a missing library or an unsupported option
raises.  See INTEGRATION.md for the one-line change that makes the reference's registry resolve
this class.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from logging import getLogger
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from . import graph_prep
from . import hidden_pad
from .ops import HotPath, PathSpec, diagonal_mask, masked_mae_device, masked_mae_loss, spec_from_config

try:  # inside a LibCity checkout: subclass the real plugin base so isinstance checks hold
    from libcity.model.abstract_traffic_state_model import AbstractTrafficStateModel  # type: ignore
except Exception:  # standalone: same two-level base (abstract_model.py:4-7, abstract_traffic_state_model.py:4-9)
    class AbstractModel(nn.Module):
        def __init__(self, config, data_feature):
            nn.Module.__init__(self)

        def predict(self, batch):
            raise NotImplementedError

        def calculate_loss(self, batch):
            raise NotImplementedError

    class AbstractTrafficStateModel(AbstractModel):
        def __init__(self, config, data_feature):
            self.data_feature = data_feature
            super().__init__(config, data_feature)


def masked_mae(preds: torch.Tensor, labels: torch.Tensor, null_val=float("nan"), min_s: float = 1e-4):
    """loss.masked_mae_torch (reference libcity/model/loss.py:17-29), including its in-place
    zeroing of small labels."""
    labels[torch.abs(labels) < min_s] = 0
    mask = ~torch.isnan(labels) if (isinstance(null_val, float) and math.isnan(null_val)) else labels.ne(null_val)
    mask = mask.float()
    mask = mask / torch.mean(mask)
    mask = torch.where(torch.isnan(mask), torch.zeros_like(mask), mask)
    loss = torch.abs(preds - labels) * mask
    loss = torch.where(torch.isnan(loss), torch.zeros_like(loss), loss)
    return torch.mean(loss)


class _GraphConvParams(nn.Module):
    """Parameter holder named like the reference AGCN (MultiATGCN.py:59-73)."""

    def __init__(self, dim_in, dim_out, k_total, embed_dim):
        super().__init__()
        self.weights_g = nn.Parameter(torch.empty(k_total, 1, 1))
        self.weights_pool = nn.Parameter(torch.empty(embed_dim, k_total, dim_in, dim_out))
        self.bias_pool = nn.Parameter(torch.empty(embed_dim, dim_out))


class _GraphCellParams(nn.Module):
    """ATGRUCell parameter tree: .gate / .update (MultiATGCN.py:112-118)."""

    def __init__(self, dim_in, hidden, k_total, embed_dim):
        super().__init__()
        self.gate = _GraphConvParams(dim_in + hidden, 2 * hidden, k_total, embed_dim)
        self.update = _GraphConvParams(dim_in + hidden, hidden, k_total, embed_dim)


class _DenseCellParams(nn.Module):
    """Residual GRUCell parameter tree: two nn.Linear (MultiATGCN.py:134-140)."""

    def __init__(self, dim_in, hidden):
        super().__init__()
        self.gate = nn.Linear(dim_in + hidden, 2 * hidden)
        self.update = nn.Linear(dim_in + hidden, hidden)


class _EncoderParams(nn.Module):
    """ATGRUEncoder parameter tree (MultiATGCN.py:156-186)."""

    def __init__(self, layers, in_steps, feat_in, hidden, k_total, embed_dim, gcn_off=False):
        super().__init__()
        self.agru_cells = nn.ModuleList()
        self.res_cells = nn.ModuleList()
        self.weights_gru = nn.Parameter(torch.empty(layers, in_steps))
        for l in range(layers):
            cin = feat_in if l == 0 else hidden
            if gcn_off:   # dense GRU cells take the place of the graph cells, no residual cells (:187-192)
                self.agru_cells.append(_DenseCellParams(cin, hidden))
            else:
                self.agru_cells.append(_GraphCellParams(cin, hidden, k_total, embed_dim))
                self.res_cells.append(_DenseCellParams(cin, hidden))


def _is_series_batch(batch) -> bool:
    """a batch of the resident-series loader (multistgraph_amd.dataset.ResidentBatch), or a plain dict with the same
    two entries: the device-resident raw series and B label starts instead of materialised windows.  (No ``in`` on
    foreign objects: the reference's Batch has no __contains__ and raises KeyError on the fallback protocol.)"""
    from .dataset import ResidentBatch
    if isinstance(batch, ResidentBatch):
        return True
    return isinstance(batch, dict) and "label_start" in batch and "series" in batch


class _TrainStep(torch.autograd.Function):
    """autograd node of one training-mode forward: matgcn_forward_train / matgcn_backward.
    The parameters ride along as inputs so that autograd routes their gradients; X gets none (the reference
    never differentiates w.r.t. the batch)."""

    @staticmethod
    def forward(ctx, path, x, drop_mask, h0, names, *params):
        h0 = None if h0 is None else h0.detach()
        out = path.forward_train(x, drop_mask, h0)
        ctx.path, ctx.x, ctx.mask, ctx.names, ctx.h0 = path, x, drop_mask, names, h0
        ctx.params = params
        ctx.generation = path.train_generation
        return out

    @staticmethod
    def backward(ctx, d_out):
        path = ctx.path
        if path.train_generation != ctx.generation:
            raise RuntimeError("MultiATGCN backward: another forward ran on this model between this forward and its "
                               "backward; the saved activations live in the shared workspace (one graph at a time)")
        state = {k: p for k, p in zip(ctx.names, ctx.params)}
        grads = path.backward(ctx.x, d_out.contiguous(), state, ctx.mask, ctx.h0)
        # the initial state (static features, :406-409) gets its gradient back: torch autograd carries it on through
        # expand() and static_initial_gru
        return (None, None, None, grads.get(path.D_H0), None) + tuple(grads.get(k) if p.requires_grad else None
                                                                      for k, p in zip(ctx.names, ctx.params))


class MultiATGCN(AbstractTrafficStateModel):
    def __init__(self, config, data_feature):
        super().__init__(config, data_feature)
        get = config.get
        self.num_nodes = data_feature.get("num_nodes", 1)
        self.input_window = get("input_window", 1)
        self.output_window = get("output_window", 1)
        self.add_time_in_day = get("add_time_in_day", False)
        self.add_day_in_week = get("add_day_in_week", False)
        self.node_specific_off = get("node_specific_off", False)
        self.fnn_off = get("fnn_off", False)
        self.gcn_off = get("gcn_off", False)
        self.batch_size = get("batch_size", 64)
        self.device = get("device", torch.device("cpu"))
        config["num_nodes"] = self.num_nodes  # the reference writes this back (:233)
        self.embed_dim_node = get("embed_dim_node", 10)
        self.embed_dim_adj = get("embed_dim_adj", 10)
        self.adpadj = get("adpadj", "bidirection")
        self.adjtype = get("adjtype", "od")
        self.cheb_order = get("cheb_order", 2)
        self.start_dim = get("start_dim", 0)
        self.end_dim = get("end_dim", 1)
        self.load_dynamic = get("load_dynamic", False)
        self.hidden_dim = get("rnn_units", 64)
        if self.hidden_dim > hidden_pad.WIDTH:
            raise NotImplementedError("rnn_units = %d: the HIP kernels hold 64 hidden channels; narrower models run on "
                                      "them zero-padded (hidden_pad.py), wider ones are not built" % self.hidden_dim)
        self._padded = self.hidden_dim < hidden_pad.WIDTH   # parameters keep the reference's shapes; the kernels see 64
        self.num_layers = get("num_layers", 2)
        assert self.num_layers >= 1, "At least one recurrent layer in the encoder"
        if self.add_day_in_week and not self.add_time_in_day:
            raise ValueError("add_day_in_week without add_time_in_day is undefined in the reference (:313-318)")
        static = data_feature.get("static", None)
        if self.input_window != 24:
            raise ValueError("the reference fuses 24-step heads (:373-393): input_window must be 24")
        embed_dim_cfg = self.embed_dim_node
        if self.node_specific_off:
            self.embed_dim_node = 1     # what the encoder's AGCN pools see (:166); node_emb itself shrinks after init

        # ---- one-off host graph prep (:238-283) -> first-order static supports (fp32, host)
        mats = graph_prep.build_static_supports(data_feature.get("adj_mx"), data_feature.get("coordinate"),
                                                None if static is None else np.asarray(static), self.adjtype)
        use_static = self.adpadj == "none" or self.adjtype == "multi"  # (:87-93)
        self._static_host = torch.from_numpy(np.stack(mats, 0)) if use_static else None
        self._static_dev: Optional[torch.Tensor] = None

        # ---- parameters, registered in the reference's order with the reference's names (:285-344)
        n = self.num_nodes
        rank = min(n, self.embed_dim_adj)  # torch.svd(adj)[..][:, :embed_dim_adj] (:299-304)
        # Every step below that consumes torch's global RNG does so in the reference's order, so that the same
        # torch.manual_seed gives bit-identical initial weights: the randn of (:296) - drawn at the CONFIG width even
        # under node_specific_off (the reference shrinks its own embed_dim_node only at :351) -, the nn.Linear /
        # Conv2d constructors, then _init_parameters over parameters() in registration order.
        # With static features (add_static, :286-294): static_initial_node is registered first (it stays in the
        # state_dict although forward never uses it), torch.pca_lowrank draws from the generator, and node_emb starts
        # from static_initial_node(static @ v) - only to be re-initialised by _init_parameters like everything else.
        self._static_q = min(n, embed_dim_cfg)
        if static is not None:
            st = torch.as_tensor(np.asarray(static), dtype=torch.float32)
            self.register_buffer("static", st, persistent=False)   # follows model.to(device); not in the state_dict
            self.static_initial_node = nn.Sequential(OrderedDict(
                [("embd", nn.Linear(self._static_q, embed_dim_cfg, bias=True)), ("relu1", nn.ReLU())]))
            _, _, v = torch.pca_lowrank(st, q=self._static_q)
            with torch.no_grad():
                self.node_emb = nn.Parameter(self.static_initial_node(torch.matmul(st, v)))
        else:
            self.static = None
            self.node_emb = nn.Parameter(torch.randn(n, embed_dim_cfg))
        self.node_vec1 = nn.Parameter(torch.empty(n, rank))
        self.node_vec2 = nn.Parameter(torch.empty(rank, n))
        self.spec: PathSpec = spec_from_config(config, data_feature, n, rank, 0 if not use_static else len(mats),
                                               diagonal_mask(self._static_host), hidden=hidden_pad.WIDTH)
        self.output_dim = self.spec.out_dim
        self.feature_final = self.spec.feat_in
        self.len_ts = self.spec.n_ts
        self.weight_ts = nn.ParameterList(
            [nn.Parameter(torch.empty(1, 24, n, self.output_dim)) for _ in range(self.len_ts)])
        self.weight_tsg = nn.Parameter(torch.empty(self.len_ts))
        if static is not None:   # initial state of the encoder from the static features (:335-338)
            self.static_initial_gru = nn.Sequential(OrderedDict(
                [("embd", nn.Linear(self._static_q, self.hidden_dim, bias=True)), ("relu1", nn.ReLU())]))
        self.encoder = _EncoderParams(self.num_layers, self.input_window, self.feature_final, self.hidden_dim,
                                      self.spec.k_total, self.embed_dim_node, self.gcn_off)
        self.end_conv = nn.Conv2d(self.input_window, self.output_window * self.output_dim,
                                  kernel_size=(1, self.hidden_dim), bias=True)   # (:340-341)
        if self.fnn_off:   # the reference builds the full head first, then replaces it (:342-344): two sets of draws
            self.end_conv = nn.Conv2d(1, self.output_window * self.output_dim, kernel_size=(1, self.hidden_dim),
                                      bias=True)
        self._logger = getLogger()
        self._scaler = data_feature.get("scaler")
        self._init_parameters()
        if self.node_specific_off:  # (:350-354)
            self.node_emb = nn.Parameter(torch.ones(n, 1), requires_grad=False)
        self._paths: Dict[int, HotPath] = {}
        self._valid_label_tables = {}
        self._prepared_key = None
        self.cache_prepared = True

    def _init_parameters(self):
        """xavier_uniform on every >=2-D parameter, U(0,1) on every 1-D one (:356-361)."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
            else:
                nn.init.uniform_(p)

    # ---- hot path plumbing ----------------------------------------------------------------------
    def _state(self) -> Dict[str, torch.Tensor]:
        state = {k: v.detach() for k, v in self.named_parameters()}
        return hidden_pad.pad_state(state, self.hidden_dim, self.feature_final) if self._padded else state

    def _initial_state(self, batch: int) -> Optional[torch.Tensor]:
        """(L, B, N, H) initial encoder state from the static features, or None (zeros): a PCA of the static table and
        one nn.Linear + ReLU, expanded over layers and samples (:405-409).  Host-side torch - tiny, and the PCA is
        torch.pca_lowrank exactly as in the reference (randomised: it draws from torch's generator on every forward)."""
        if self.static is None:
            return None
        _, _, v = torch.pca_lowrank(self.static, q=self._static_q)
        emb = self.static_initial_gru(torch.matmul(self.static, v))
        return emb.expand(self.num_layers, batch, -1, -1)

    def _params_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _path_for(self, x: torch.Tensor) -> HotPath:
        if not x.is_cuda:
            raise RuntimeError("MultiATGCN.forward runs on the HIP hot path only: batch['X'] is on %s. "
                               "Move the model and the batch to the GPU (config['device'])." % x.device)
        return self._path_for_batch(x.shape[0], x.device)

    def _path_for_batch(self, batch: int, device) -> HotPath:
        hp = self._paths.get(batch)
        if hp is None or hp.device != device:
            hp = HotPath(self.spec, batch, device)
            self._paths[batch] = hp
            self._prepared_key = None
        if self._static_host is not None and (self._static_dev is None or self._static_dev.device != device):
            self._static_dev = self._static_host.to(device).contiguous()
        key = (id(hp),) + self._params_key()
        if key != self._prepared_key or not self.cache_prepared:
            hp.bind(self._state(), self._static_dev)
            hp.prepare()
            self._prepared_key = key
        return hp

    # ---- the plugin surface the executor calls ---------------------------------------------------
    def _run(self, source, batch: int, device):
        """One forward on the HIP path.  source: the windows tensor X, or a (series, label_start, rel_steps) triple
        (device-resident raw series).  Inference under no_grad / eval; with gradients
        enabled the training form (activations kept, HIP backward behind torch autograd)."""
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        h0 = self._initial_state(batch)
        h0 = None if h0 is None else (hidden_pad.pad_last(h0) if self._padded else h0.contiguous())
        hp = self._path_for_batch(batch, device)
        if not needs_grad:
            if self.training:
                raise NotImplementedError("training-mode forward (dropout, :416) without autograd is not built: "
                                          "call model.eval() for inference")
            if isinstance(source, torch.Tensor):
                return hp.forward(source, h0)
            return hp.forward_series(source[0], source[1], source[2], h0)
        # training step: HIP forward that keeps its activations + HIP backward behind torch autograd
        mask = None
        if self.training:   # F.dropout(output, p=0.1) in front of end_conv (:416), drawn from torch's generator
            mask = nn.functional.dropout(torch.ones(batch, 1 if self.fnn_off else self.input_window,
                                                    self.num_nodes, self.hidden_dim, device=device),
                                         p=0.1, training=True)   # fnn_off keeps the last step only (:412)
            if self._padded:
                mask = hidden_pad.pad_last(mask)
        # static_initial_* are host-side torch layers: their gradients flow through h0, not through the HIP backward
        named = [(k, p) for k, p in self.named_parameters() if not k.startswith("static_initial")]
        if self._padded:   # differentiable zero-padding: autograd slices the gradients back into the real parameters
            padded = hidden_pad.pad_state(dict(named), self.hidden_dim, self.feature_final)
            named = [(k, padded[k]) for k, _ in named]
        return _TrainStep.apply(hp, source, mask, h0, tuple(k for k, _ in named), *[p for _, p in named])

    def forward(self, batch):
        if _is_series_batch(batch):    # MTHDatasetResident: windows and labels are gathered on the device (f-2)
            src = self._batch_source(batch)
            return self._run(src, int(src[1].shape[0]), src[0].device)
        x = batch["X"]
        assert x.shape[2] == self.num_nodes  # (:195)
        if not x.is_cuda:
            raise RuntimeError("MultiATGCN.forward runs on the HIP hot path only: batch['X'] is on %s. "
                               "Move the model and the batch to the GPU (config['device'])." % x.device)
        if x.dtype != torch.float32:
            x = x.float()
        return self._run(x.contiguous(), int(x.shape[0]), x.device)

    def predict(self, batch):
        return self.forward(batch)

    # ---- the same surface fed from the device-resident raw series (no windows, no labels materialised) ----------
    def _batch_source(self, batch):
        """the (series, label_start, rel_steps) triple of a resident-series batch; its label starts were validated on
        the host when the loader's table was built (ResidentBatch.range_checked), a plain dict is checked here"""
        rel = batch.get("rel_steps") if hasattr(batch, "get") else None
        return self._series_source(batch["series"], batch["label_start"], rel,
                                   trusted=bool(getattr(batch, "range_checked", False)))

    def _series_source(self, series: torch.Tensor, label_start, rel_steps, trusted: bool = False):
        """(series, int32 device label starts, rel_steps), with the range contract of the series entry points checked
        on the host: every window row and every target row of every sample inside the series (windows.
        check_label_starts).  A host table (numpy / CPU tensor) is checked before it is uploaded; a device tensor costs
        one min / max read-back the first time it is seen (tables validated once - e.g. by MTHDatasetResident or
        windows.epoch_batches - are registered with ``mark_label_starts_valid`` and cost nothing per batch)."""
        from . import windows
        if rel_steps is None:
            rel_steps = windows.window_offsets(self.input_window)
        if not series.is_cuda:
            raise RuntimeError("the series must live on the GPU (HIP path only)")
        steps = int(series.shape[0])
        if not isinstance(label_start, torch.Tensor) or not label_start.is_cuda:
            host = np.asarray(label_start.cpu() if isinstance(label_start, torch.Tensor) else label_start)
            windows.check_label_starts(host, rel_steps, self.output_window, steps)
            return (series, torch.as_tensor(host.astype(np.int32)).to(series.device), rel_steps)
        ls = label_start.to(torch.int32)
        if trusted:
            return (series, ls, rel_steps)
        # Only tables that were registered as OBJECTS are trusted (a key built from data_ptr() survives
        # the tensor - the caching allocator hands the same address to the next batch, whose _version is 0 again - so a
        # later out-of-range batch could hit the stale key and skip the check).  A row of a registered table is a view of
        # it (``_base``); anything else - a fresh tensor per batch, the int32 copy of an int64 tensor - is checked on every
        # call (one min / max read-back).
        base = ls._base if ls._base is not None else ls
        rel_key = tuple(int(v) for v in (min(rel_steps), max(rel_steps)))
        if self._label_table_is_valid(base, steps, rel_key):
            return (series, ls, rel_steps)
        lo, hi = (int(v) for v in torch.aminmax(ls))
        windows.check_label_starts(np.array([lo, hi]), rel_steps, self.output_window, steps)
        if base is not ls:      # a view of a longer-lived table: validate the table once, then trust its other rows
            lo, hi = (int(v) for v in torch.aminmax(base))
            try:
                windows.check_label_starts(np.array([lo, hi]), rel_steps, self.output_window, steps)
                self.mark_label_starts_valid(base, steps, rel_steps)
            except ValueError:
                pass
        return (series, ls, rel_steps)

    def _label_table_is_valid(self, table: torch.Tensor, steps: int, rel_key) -> bool:
        ent = self._valid_label_tables.get(id(table))
        if ent is None:
            return False
        ref, version, st, rk = ent
        if ref() is not table:      # the id belongs to a dead tensor: forget it
            del self._valid_label_tables[id(table)]
            return False
        return version == table._version and st == steps and rk == rel_key

    def mark_label_starts_valid(self, table: torch.Tensor, series_steps: int, rel_steps) -> None:
        """Register a device table of label starts that has been validated on the host (windows.check_label_starts):
        batches that are views of it skip the per-call range check.  The table is remembered as an object (weak
        reference + ``_version``): writing into it, or its death, ends the trust."""
        import weakref
        if len(self._valid_label_tables) > 64:
            self._valid_label_tables = {k: v for k, v in self._valid_label_tables.items() if v[0]() is not None}
        self._valid_label_tables[id(table)] = (weakref.ref(table), table._version, int(series_steps),
                                               tuple(int(v) for v in (min(rel_steps), max(rel_steps))))

    def forward_series(self, series: torch.Tensor, label_start: torch.Tensor, rel_steps=None):
        """forward() without materialised windows: ``series`` (T, N, F) float32 resident on the GPU, ``label_start``
        (B) first-target indices; the window rows are gathered on the device (windows.window_offsets gives
        ``rel_steps``; the default is the reference's 2 x 24 h closeness + 1-week + 4-week heads).  Replaces
        MTHDataset._generate_input_data + the per-batch host copy (mth_dataset.py:110-160, data/utils.py:68-72,
        batch.py:43-57); trains like forward() when gradients are enabled."""
        src = self._series_source(series, label_start, rel_steps)
        return self._run(src, int(src[1].shape[0]), series.device)

    def predict_series(self, series: torch.Tensor, label_start: torch.Tensor, rel_steps=None):
        return self.forward_series(series, label_start, rel_steps)

    def calculate_loss_series(self, series: torch.Tensor, label_start: torch.Tensor, rel_steps=None, _checked=False):
        """calculate_loss (:422-427) of the batch given by ``label_start``: the prediction from the series-fed forward,
        the targets series[label_start[b] + o] gathered by the loss kernel on the device.  With gradients enabled this is
        the executor's training step (traffic_state_executor.py:411-422) without any host-side window or label."""
        src = self._series_source(series, label_start, rel_steps, trusted=_checked)
        pred = self._run(src, int(src[1].shape[0]), series.device)
        ls = src[1]
        affine = self._affine_scaler()
        if affine is not None and series.dtype == torch.float32:
            if pred.requires_grad:
                return masked_mae_loss(pred, series, self.start_dim, affine[0], affine[1], null_val=0.0, label_start=ls)
            return masked_mae_device(pred, series, self.start_dim, affine[0], affine[1], null_val=0.0, label_start=ls)[0]
        rows = ls.long()[:, None] + torch.arange(self.output_window, device=series.device)[None, :]
        y_true = self._scaler.inverse_transform(series[rows][..., self.start_dim:self.end_dim].clone())
        return masked_mae(self._scaler.inverse_transform(pred), y_true, 0)

    def horizon_mae_series(self, series: torch.Tensor, label_start: torch.Tensor, rel_steps=None, _checked=False):
        """(out,) MAE@1..MAE@out of the batch given by ``label_start`` (evaluator "single" mode), all on the device."""
        affine = self._affine_scaler()
        if affine is None:
            raise NotImplementedError("horizon_mae_series needs an affine scaler (StandardScaler / NoneScaler)")
        src = self._series_source(series, label_start, rel_steps, trusted=_checked)
        pred = self._run(src, int(src[1].shape[0]), series.device)
        return masked_mae_device(pred, series, self.start_dim, affine[0], affine[1], label_start=src[1])[1:]

    def collect_metrics(self, evaluator, batch):
        """One test batch of TrafficStateExecutor.evaluate (traffic_state_executor.py:264-273, 289) without leaving the
        GPU: predict, de-scale prediction and label with the scaler's affine inside the metric kernel, and reduce into
        ``evaluator`` (multistgraph_amd.evaluator.DeviceEvaluator).  ``batch`` is a window batch (``X``, ``y``) or a
        resident-series batch (``series``, ``label_start``: MTHDatasetResident).  Returns the prediction (scaled)."""
        affine = self._affine_scaler()
        if affine is None:
            raise NotImplementedError("collect_metrics needs an affine scaler (StandardScaler / NoneScaler)")
        if _is_series_batch(batch):
            src = self._batch_source(batch)
            pred = self._run(src, int(src[1].shape[0]), src[0].device)
            evaluator.collect_scaled(pred, src[0], self.start_dim, affine[0], affine[1], label_start=src[1])
        else:
            pred = self.predict(batch)
            evaluator.collect_scaled(pred, batch["y"], self.start_dim, affine[0], affine[1])
        return pred

    def gradient_exchange(self):
        """(bucket, leftovers) for the gradient exchange of data-parallel training after ``loss.backward()``:
        ``bucket`` = the flat fp32 buffer the gradients of the HIP-path parameters are views of (one all-reduce, no
        copy), or None when some of them live elsewhere (rnn_units < 64: autograd slices the padded gradients back
        into tensors of their own; gradients accumulated over several backward calls); ``leftovers`` = every other
        ``.grad`` that is not None - with a bucket these are the host-side ``static_initial_*`` layers of the
        static-feature path, whose gradients arrive through d_h0 and torch autograd, without one ALL gradients.
        sharding.allreduce_model_grads_(model) is the entry point that reduces both."""
        with_grad = [(k, p) for k, p in self.named_parameters() if p.requires_grad and p.grad is not None]
        for hp in self._paths.values():
            b = hp.grad_bucket
            if b is None:
                continue
            lo, hi = b.data_ptr(), b.data_ptr() + b.numel() * 4
            mine = [p for k, p in self.named_parameters() if p.requires_grad and not k.startswith("static_initial")]
            if mine and all(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in mine):
                return b, [p.grad for _, p in with_grad if not lo <= p.grad.data_ptr() < hi]
        return None, [p.grad for _, p in with_grad]

    def gradient_bucket(self) -> Optional[torch.Tensor]:
        """The flat fp32 buffer that holds the gradient of EVERY parameter after ``loss.backward()`` (their ``.grad``
        are views of it) - or None as soon as any gradient lives outside it (static-feature layers, rnn_units < 64,
        accumulated gradients): reducing the buffer alone would then leave replicas diverging.  Data-parallel loops
        call sharding.allreduce_model_grads_(model), which handles both cases."""
        bucket, rest = self.gradient_exchange()
        return bucket if not rest else None

    def _affine_scaler(self):
        """(mean, std) when the scaler de-scales as x*std + mean with scalar parameters - LibCity's StandardScaler,
        NoneScaler, NormalScaler, MinMax01Scaler, MinMax11Scaler (libcity/utils/normalization.py:20-113) all do -,
        else None (LogScaler).  Found by probing inverse_transform at three points, so any scaler object works."""
        sc = self._scaler
        if sc is None:
            return None
        try:
            probe = sc.inverse_transform(torch.tensor([0.0, 1.0, 2.0], dtype=torch.float64))
            f0, f1, f2 = (float(v) for v in probe)
        except Exception:
            return None
        mean, std = f0, f1 - f0
        if not all(math.isfinite(v) for v in (f0, f1, f2)) or abs(f2 - (mean + 2.0 * std)) > 1e-9 * max(1.0, abs(f2)):
            return None
        return mean, std

    def calculate_loss(self, batch):
        """de-scale prediction and label, masked MAE with null value 0 (:422-427).  With an affine scaler the
        de-scale + mask + reduction run fused on the device; any other scaler takes the reference's torch
        arithmetic on the HIP prediction."""
        if _is_series_batch(batch):
            src = self._batch_source(batch)
            return self.calculate_loss_series(src[0], src[1], src[2], _checked=True)
        y_true = batch["y"]
        y_predicted = self.predict(batch)
        affine = self._affine_scaler()
        if affine is not None and y_true.is_cuda and y_true.dtype == torch.float32 and y_predicted.requires_grad:
            return masked_mae_loss(y_predicted, y_true, self.start_dim, affine[0], affine[1], null_val=0.0)
        if affine is not None and y_true.is_cuda and y_true.dtype == torch.float32:
            return masked_mae_device(y_predicted, y_true, self.start_dim, affine[0], affine[1], null_val=0.0)[0]
        y_true = self._scaler.inverse_transform(y_true[..., self.start_dim:self.end_dim])
        y_predicted = self._scaler.inverse_transform(y_predicted)
        return masked_mae(y_predicted, y_true, 0)

    def horizon_mae(self, batch):
        """(out,) device tensor MAE@1..MAE@out in the evaluator's "single" mode on de-scaled values
        (traffic_state_executor.py:268-273, traffic_state_evaluator.py:87-104), without leaving the GPU."""
        affine = self._affine_scaler()
        if affine is None:
            raise NotImplementedError("horizon_mae needs an affine scaler (StandardScaler / NoneScaler)")
        if _is_series_batch(batch):
            src = self._batch_source(batch)
            return self.horizon_mae_series(src[0], src[1], src[2], _checked=True)
        pred = self.predict(batch)
        return masked_mae_device(pred, batch["y"], self.start_dim, affine[0], affine[1])[1:]
