"""rnn_units below 64 on the 64-wide kernels (the commented sweep of reference run_model_parameter.py:11: 16, 32).

The HIP kernels are built around 64 hidden channels (MultiATGCN.py:322 default).  A narrower model runs on them EXACTLY by
zero-padding every hidden axis of its parameters to 64: a padded channel has zero incoming weights and a zero bias, so its
gates are sigmoid(0), its candidate is tanh(0) = 0 and - starting from a zero (or zero-padded) state - its state stays
r*0 + (1-r)*0 = 0 through the graph cell, the residual cell and the blend (MultiATGCN.py:120-128,142-150,204-210); its
outgoing weights (the hidden ROWS of every later contraction, the Conv2d head's kernel) are zero as well, so nothing it
holds reaches a real channel.  The padding is a handful of differentiable torch ops on ~1 MB of parameters: in training
autograd carries the gradients of the padded tensors back into the reference-shaped parameters (the checkpoint ABI is
untouched).  Layouts: an output axis of width 2H is [z | r] -> each half padded on its own; an input axis is [x | h] ->
layer 0 (x = the fused input channels) pads at the end, deeper layers (x = the hidden state below) pad both halves.
"""
from __future__ import annotations

import re
from typing import Dict

import torch

WIDTH = 64

_CELL = re.compile(r"^encoder\.(agru_cells|res_cells)\.(\d+)\.(gate|update)\.(weights_pool|bias_pool|weight|bias)$")


def _end(t: torch.Tensor, axis: int, width: int = WIDTH) -> torch.Tensor:
    extra = width - t.shape[axis]
    if extra == 0:
        return t
    shape = list(t.shape)
    shape[axis] = extra
    return torch.cat([t, t.new_zeros(shape)], axis)


def _halves(t: torch.Tensor, axis: int, h: int) -> torch.Tensor:
    return torch.cat([_end(t.narrow(axis, 0, h), axis), _end(t.narrow(axis, h, h), axis)], axis)


def _in_axis(t: torch.Tensor, axis: int, layer: int, feat_in: int, h: int) -> torch.Tensor:
    if layer == 0:
        return _end(t, axis, feat_in + WIDTH)
    return _halves(t, axis, h)


def pad_last(t: torch.Tensor) -> torch.Tensor:
    """states / dropout masks (..., H) -> (..., 64)"""
    return _end(t, t.dim() - 1).contiguous()


def pad_state(state: Dict[str, torch.Tensor], hidden: int, feat_in: int) -> Dict[str, torch.Tensor]:
    """reference-named tensors of a model with ``hidden`` < 64 channels -> the same names at 64 channels"""
    if not 0 < hidden < WIDTH:
        raise ValueError("pad_state is for 0 < rnn_units < 64, got %d" % hidden)
    out = {}
    for name, t in state.items():
        m = _CELL.match(name)
        if m:
            layer, part, kind = int(m.group(2)), m.group(3), m.group(4)
            o_pad = (lambda v, ax: _halves(v, ax, hidden)) if part == "gate" else (lambda v, ax: _end(v, ax))
            if kind == "weights_pool":      # (d, K, I, O)
                t = o_pad(_in_axis(t, 2, layer, feat_in, hidden), 3)
            elif kind == "bias_pool":       # (d, O)
                t = o_pad(t, 1)
            elif kind == "weight":          # nn.Linear (O, I)
                t = o_pad(_in_axis(t, 1, layer, feat_in, hidden), 0)
            else:                           # nn.Linear bias (O)
                t = o_pad(t, 0)
        elif name == "end_conv.weight":     # (out, T or 1, 1, H)
            t = _end(t, 3)
        out[name] = t.contiguous()
    return out
