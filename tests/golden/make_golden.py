#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running THE REFERENCE MODEL ITSELF on CPU.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is stored is data only: the case description (shapes, seeds, flags), per-stage outputs of the
reference (support stacks, one AGCN call, one ATGRU / residual-GRU cell call, the head-fusion
output, the encoder sequence / final states, the final prediction, calculate_loss, MAE@k from the
reference's own TrafficStateEvaluator).  Inputs and parameters are regenerated from the seeds by
multistgraph_amd.synthetic (checksums of both are stored so a drift is detected).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from multistgraph_amd import synthetic as syn  # noqa: E402

import libcity.model.traffic_flow_prediction.MultiATGCN  # noqa: E402,F401  (reference)
REF = sys.modules["libcity.model.traffic_flow_prediction.MultiATGCN"]
from libcity.evaluator.traffic_state_evaluator import TrafficStateEvaluator  # noqa: E402


def build_reference(case):
    n = case["nodes"]
    df = syn.make_data_feature(n, case["seed"], case.get("city", "DC"),
                               static_dim=case.get("static_dim", 0), ext_dim=case["feat"] - 1,
                               lens=case.get("lens", (2, 1, 1)))
    cfg = dict(input_window=24, output_window=case["out"], add_time_in_day=case.get("tid", True),
               add_day_in_week=False, load_dynamic=case["feat"] > 2,
               adjtype=case["adjtype"], adpadj=case["adpadj"], cheb_order=case["cheb"],
               embed_dim_node=20, embed_dim_adj=20, rnn_units=64, num_layers=2,
               device=torch.device("cpu"), batch_size=case["batch"])
    cfg.update(case.get("flags", {}))        # ablation switches: gcn_off / fnn_off / node_specific_off
    torch.manual_seed(0)
    model = REF.MultiATGCN(cfg, df).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    state = syn.closed_form_state(shapes, case["seed"])
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    return model, df, cfg, state


class pinned_pca:
    """The reference draws a fresh randomised PCA basis with torch.pca_lowrank inside EVERY forward (MultiATGCN.py:407).
    To make its outputs reproducible elsewhere, the basis v of one real call (seeded) is recorded and replayed for
    every forward made under this context; the fixture stores that v."""

    def __init__(self, model, seed):
        self.v = None
        if model.static is not None:
            torch.manual_seed(seed + 4242)
            _, _, self.v = torch.pca_lowrank(model.static, q=min(model.num_nodes, model.embed_dim_node))
        self.real = torch.pca_lowrank

    def __enter__(self):
        if self.v is not None:
            torch.pca_lowrank = lambda A, q=None, center=True, niter=2: (None, None, self.v)
        return self

    def __exit__(self, *exc):
        torch.pca_lowrank = self.real


def run_case(case):
    model, df, cfg, state = build_reference(case)
    with pinned_pca(model, case["seed"]) as pca:
        out = _run_case(case, model, df, cfg, state)
        if pca.v is not None:
            out["pca_v"] = pca.v.numpy()
            with torch.no_grad():   # the (N, H) initial state every layer and sample starts from (:406-409)
                out["h0"] = model.static_initial_gru(torch.matmul(model.static, pca.v)).numpy()
    return out


def _run_case(case, model, df, cfg, state):
    n, b = case["nodes"], case["batch"]
    x, y = syn.make_batch_arrays(b, n, case["out"], case["seed"], feat=case["feat"],
                                 x_steps=24 * sum(case.get("lens", (2, 1, 1))))
    xb, yb = torch.from_numpy(x), torch.from_numpy(y)
    out = {}
    out["x_checksum"] = np.float64(x.astype(np.float64).sum())
    out["param_checksum"] = np.float64(sum(float(np.abs(v.astype(np.float64)).sum()) for v in state.values()))

    # static supports as the reference stores them (plain attributes, MultiATGCN.py:264-283)
    statics = np.stack([s[1].numpy() for s in model.supports], 0)
    if case.get("big", False):
        # 3 x N x N does not fit a fixture at N = 4096: every BIG_SUB-th element + per-support [sum, sum |.|]
        out["static_sub"] = statics.reshape(statics.shape[0], -1)[:, ::BIG_SUB].copy()
        out["static_sums"] = np.stack([statics.astype(np.float64).sum((1, 2)), np.abs(statics.astype(np.float64)).sum((1, 2))], 1)
    else:
        out["static_supports"] = statics

    # --- capture the support stack of every AGCN and the fused encoder input
    captured = {"stacks": [], "x0": None}
    real_einsum = torch.einsum

    def spy(eq, *ops):
        if eq == "knm,bmc->bknc" and len(captured["stacks"]) < 4:
            captured["stacks"].append(ops[0].detach().clone().numpy())
        return real_einsum(eq, *ops)

    def grab(mod, args):
        captured["x0"] = args[0].detach().clone().numpy()

    hook = model.encoder.register_forward_pre_hook(grab)
    torch.einsum = spy
    try:
        with torch.no_grad():
            pred = model.predict({"X": xb})
    finally:
        torch.einsum = real_einsum
        hook.remove()
    out["pred"] = pred.numpy()
    out["x0"] = captured["x0"]
    if case.get("stages", False):
        # order of first four AGCN calls: l0.gate, l0.update (t=0) ... both again at t=1
        out["stack_l0_gate"] = captured["stacks"][0]
        out["stack_l0_update"] = captured["stacks"][1]
        rng = np.random.default_rng(case["seed"] + 99)
        cell = model.encoder.agru_cells[0]
        res = model.encoder.res_cells[0]
        c0 = case["feat"]
        xs = rng.standard_normal((b, n, c0)).astype(np.float32)
        hs = np.tanh(rng.standard_normal((b, n, 64))).astype(np.float32)
        with torch.no_grad():
            xin = torch.cat((torch.from_numpy(xs), torch.from_numpy(hs)), -1)
            out["agcn_gate_l0"] = cell.gate(xin, model.node_emb, model.node_vec1, model.node_vec2,
                                            model.supports).numpy()
            out["cell_l0"] = cell(torch.from_numpy(xs), torch.from_numpy(hs), model.node_emb,
                                  model.node_vec1, model.node_vec2, model.supports).numpy()
            out["res_l0"] = res(torch.from_numpy(xs), torch.from_numpy(hs), model.node_emb,
                                model.node_vec1, model.node_vec2, model.supports).numpy()
            cell1 = model.encoder.agru_cells[1]
            xs1 = np.tanh(rng.standard_normal((b, n, 64))).astype(np.float32)
            out["cell_l1"] = cell1(torch.from_numpy(xs1), torch.from_numpy(hs), model.node_emb,
                                   model.node_vec1, model.node_vec2, model.supports).numpy()
            init = model.encoder.init_hidden(b)
            seq, finals = model.encoder(torch.from_numpy(captured["x0"]), init, model.node_emb,
                                        model.node_vec1, model.node_vec2, model.supports)
        out["stage_x"] = xs
        out["stage_h"] = hs
        out["stage_x1"] = xs1
        out["enc_seq"] = seq.numpy()
        out["enc_finals"] = torch.stack(finals, 0).numpy()

    # loss + evaluator metrics (reference MultiATGCN.py:422-427, traffic_state_evaluator.py:87-104)
    with torch.no_grad():
        loss = model.calculate_loss({"X": xb, "y": yb.clone()})
    out["loss"] = np.float64(loss.item())
    ev = TrafficStateEvaluator({"metrics": ["MAE"], "evaluator_mode": "single"})
    ev.collect({"y_true": yb[..., 0:1].clone(), "y_pred": pred})
    res = ev.evaluate()
    out["mae_at"] = np.array([res["MAE@%d" % (i + 1)] for i in range(case["out"])], dtype=np.float64)
    return out


BIG_SUB = 4099   # prime stride of the static-support subsample of the N = 4096 case

CASES = []
_modes = [("multi", "unidirection"), ("multi", "bidirection"), ("multi", "none"),
          ("od", "unidirection"), ("od", "none"), ("identity", "none"), ("dist", "none"),
          ("cosine", "none")]
for (adjt, adp) in _modes:
    for cheb in (2, 3):
        CASES.append(dict(name="tiny_%s_%s_c%d" % (adjt, adp[:3], cheb), nodes=21, batch=2, out=3,
                          feat=2, adjtype=adjt, adpadj=adp, cheb=cheb, seed=10, stages=True))
# cheb_order = 1, the first value of the sweep the reference ships enabled (run_model_parameter.py:13): ONE weight
# entry broadcast by einsum over the stack [I, S_1, S_2, ..] (MultiATGCN.py:65-70,94-108)
for (adjt, adp) in [("multi", "unidirection"), ("multi", "bidirection"), ("multi", "none"), ("od", "unidirection"),
                    ("od", "none"), ("identity", "none")]:
    CASES.append(dict(name="tiny_%s_%s_c1" % (adjt, adp[:3]), nodes=21, batch=2, out=3, feat=2, adjtype=adjt,
                      adpadj=adp, cheb=1, seed=10, stages=True))
# add_static (MultiATGCN.py:244-250,286-296,335-338,406-409): static features give the 1/euclid similarity adjacency
# (a dense third support), static_initial_node / static_initial_gru, and the encoder's initial state
CASES.append(dict(name="tiny_multi_uni_c2_static", nodes=21, batch=2, out=3, feat=2, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=10, stages=True, static_dim=24))
CASES.append(dict(name="tiny_cosine_non_c3_static", nodes=21, batch=3, out=6, feat=2, adjtype="cosine",
                  adpadj="none", cheb=3, seed=100, stages=True, static_dim=30))
CASES.append(dict(name="tiny_multi_uni_out12", nodes=21, batch=3, out=12, feat=2, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=100, stages=True))
CASES.append(dict(name="tiny_multi_uni_dyn7", nodes=19, batch=2, out=6, feat=7, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=1000, stages=True))
CASES.append(dict(name="dc237_out3", nodes=237, batch=4, out=3, feat=2, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=0, city="DC"))
CASES.append(dict(name="dc237_out12", nodes=237, batch=4, out=12, feat=2, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=10, city="DC"))
CASES.append(dict(name="bm403_out24", nodes=403, batch=4, out=24, feat=2, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=0, city="BM"))
CASES.append(dict(name="bm403_out24_bi", nodes=403, batch=2, out=24, feat=2, adjtype="multi",
                  adpadj="bidirection", cheb=2, seed=100, city="BM"))


# temporal-head layouts of the reference's first sweep (run_model_parameter.py:6-7: len_closeness / len_period /
# len_trend), incl. no closeness at all (time of day then comes from the first block there is, MultiATGCN.py:397) and
# the trend loop that never advances its window (:389-393: three trend heads read the SAME 24 rows); and the
# add_time_in_day = False corner of the channel ablations (:11-12: one input channel)
for lens, out, seed in (((1, 0, 0), 3, 10), ((0, 0, 1), 6, 100), ((3, 3, 1), 12, 1000), ((1, 1, 3), 6, 0),
                        ((0, 1, 1), 6, 10)):
    CASES.append(dict(name="tiny_heads_%d%d%d" % lens, nodes=21, batch=2, out=out, feat=2, adjtype="multi",
                      adpadj="unidirection", cheb=2, seed=seed, stages=True, lens=list(lens)))
CASES.append(dict(name="tiny_notid_c2", nodes=21, batch=2, out=6, feat=1, adjtype="multi", adpadj="unidirection",
                  cheb=2, seed=10, stages=True, tid=False))

# BASELINE config 5's graph: synthetic 4096 nodes, in 24 -> out 24 (per-GPU share of the batch cut to 2 so that the
# reference finishes in minutes on CPU); prediction, loss and MAE@k only, static supports as a subsample
CASES.append(dict(name="synth4096_out24", nodes=4096, batch=2, out=24, feat=2, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=1000, city="BM", big=True))

# ablation switches of the reference (run_model_parameter.py:6-15), final outputs only
for nm, flags in (("gcnoff", {"gcn_off": True}), ("fnnoff", {"fnn_off": True}),
                  ("nodeoff", {"node_specific_off": True}),
                  ("gcnfnnoff", {"gcn_off": True, "fnn_off": True})):
    CASES.append(dict(name="abl_%s" % nm, nodes=21, batch=3, out=6, feat=2, adjtype="multi",
                      adpadj="unidirection", cheb=2, seed=10, flags=flags))

# rnn_units below the default 64 (MultiATGCN.py:322; the commented sweep of run_model_parameter.py:11 lists 16, 32, 64,
# 72): the plugin runs them zero-padded on the 64-wide kernels (multistgraph_amd/hidden_pad.py)
CASES.append(dict(name="hid32_multi_uni_c2", nodes=21, batch=2, out=3, feat=2, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=10, flags={"rnn_units": 32}))
CASES.append(dict(name="hid16_od_non_c3", nodes=21, batch=3, out=6, feat=2, adjtype="od",
                  adpadj="none", cheb=3, seed=100, flags={"rnn_units": 16}))
CASES.append(dict(name="hid32_multi_uni_c2_static", nodes=21, batch=2, out=3, feat=2, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=10, static_dim=24, flags={"rnn_units": 32}))
CASES.append(dict(name="hid48_gcnoff", nodes=21, batch=3, out=6, feat=2, adjtype="multi",
                  adpadj="unidirection", cheb=2, seed=10, flags={"rnn_units": 48, "gcn_off": True}))
CASES.append(dict(name="hid32_fnnoff_dyn7", nodes=19, batch=2, out=6, feat=7, adjtype="multi",
                  adpadj="bidirection", cheb=2, seed=1000, flags={"rnn_units": 32, "fnn_off": True}))


def main():
    only = set(sys.argv[1:])
    index = {}
    idx_path = os.path.join(HERE, "index.json")
    if os.path.exists(idx_path) and only:
        index = json.load(open(idx_path))
    for case in CASES:
        if only and case["name"] not in only:
            continue
        res = run_case(case)
        np.savez_compressed(os.path.join(HERE, case["name"] + ".npz"), **res)
        index[case["name"]] = case
        print("%-28s pred %s  |pred|max %.4f  loss %.6f" % (
            case["name"], res["pred"].shape, np.abs(res["pred"]).max(), res["loss"]))
    json.dump(index, open(idx_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
