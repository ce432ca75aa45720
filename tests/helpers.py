"""Shared test plumbing: golden cases -> inputs for the oracle and for the HIP path."""
import json
import os

import numpy as np
import torch

from multistgraph_amd import synthetic as syn

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN_DIR, "index.json")) as fh:
    INDEX = json.load(fh)

TINY = sorted(k for k in INDEX if k.startswith("tiny_"))
BIG = sorted(k for k in INDEX if INDEX[k].get("big"))      # N = 4096: prediction only, static supports subsampled
HID = sorted(k for k in INDEX if k.startswith("hid"))      # rnn_units < 64: through the plugin class only (hidden_pad)
FULL = sorted(k for k in INDEX if not k.startswith("tiny_") and k not in BIG and k not in HID)


class Case:
    def __init__(self, name):
        self.name = name
        self.meta = INDEX[name]
        self.gold = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        m = self.meta
        self.n, self.b, self.out, self.feat = m["nodes"], m["batch"], m["out"], m["feat"]
        self.adjtype, self.adpadj, self.cheb, self.seed = m["adjtype"], m["adpadj"], m["cheb"], m["seed"]
        self.flags = dict(m.get("flags", {}))
        self.k_total = syn.k_total_for(self.adjtype, self.adpadj, self.cheb)
        self.static_dim = m.get("static_dim", 0)
        self.lens = tuple(m.get("lens", (2, 1, 1)))      # (len_closeness, len_period, len_trend) in 24-step blocks
        self.tid = bool(m.get("tid", True))               # add_time_in_day
        self.data_feature = syn.make_data_feature(self.n, self.seed, m.get("city", "DC"), ext_dim=self.feat - 1,
                                                  static_dim=self.static_dim, lens=self.lens)
        self.static = self.data_feature["static"]
        self.shapes = syn.param_shapes(self.n, out_steps=self.out, feat_in=self.feat, k_total=self.k_total,
                                       static=self.static_dim > 0, len_ts=sum(self.lens), **self.flags)
        self.state = syn.closed_form_state(self.shapes, self.seed)
        self.x, self.y = syn.make_batch_arrays(self.b, self.n, self.out, self.seed, feat=self.feat,
                                               x_steps=24 * sum(self.lens))

    def config(self, device="cpu"):
        cfg = dict(input_window=24, output_window=self.out, add_time_in_day=self.tid, add_day_in_week=False,
                   load_dynamic=self.feat > 2, adjtype=self.adjtype, adpadj=self.adpadj, cheb_order=self.cheb,
                   embed_dim_node=20, embed_dim_adj=20, rnn_units=64, num_layers=2, device=torch.device(device),
                   batch_size=self.b)
        cfg.update(self.flags)
        return cfg

    def oracle_cfg(self):
        cfg = dict(adjtype=self.adjtype, adpadj=self.adpadj, cheb_order=self.cheb, num_layers=2, rnn_units=64,
                   len_closeness=24 * self.lens[0], len_period=24 * self.lens[1], len_trend=24 * self.lens[2],
                   output_window=self.out, input_window=24, add_time_in_day=self.tid, add_day_in_week=False, load_dynamic=self.feat > 2, start_dim=0,
                   end_dim=1)
        cfg.update(self.flags)
        return cfg

    def h0(self, layers=2, batch=None):
        """(L, B, N, H) initial state of the static-feature cases as the reference computed it, else None"""
        if self.static_dim == 0:
            return None
        e = torch.from_numpy(self.gold["h0"])
        return e.expand(layers, self.b if batch is None else batch, -1, -1).contiguous()

    def checksums_ok(self):
        xs = float(self.x.astype(np.float64).sum())
        ps = sum(float(np.abs(v.astype(np.float64)).sum()) for v in self.state.values())
        return (abs(xs - float(self.gold["x_checksum"])) <= 1e-9 * max(1.0, abs(xs)) and
                abs(ps - float(self.gold["param_checksum"])) <= 1e-9 * max(1.0, abs(ps)))


def max_norm_err(a, b):
    """max |a-b| / max |b| - the 'rel fp32' measure the north star quotes (1e-4)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def elementwise_excess(a, b, rtol=1e-4, floor=1e-6):
    """max over the elements of |a-b| / (rtol*|b| + floor*max|b|): <= 1 means EVERY element is within ``rtol`` relative
    of the reference value, with an absolute floor of ``floor`` x the largest reference magnitude for the elements near
    zero (fp32 rounding of O(max|b|) intermediates leaves an absolute, not a relative, error there).  The element-wise
    companion of max_norm_err for end-to-end outputs - the north star's "1e-4 rel"."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / (rtol * np.abs(b) + floor * max(np.abs(b).max(), 1e-30))).max())
