#!/usr/bin/env python3
"""Gradient fixtures for the training step, produced by THE REFERENCE MODEL ITSELF:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_grad_golden.py

For a few tiny cases of make_golden.py the reference runs one training-mode step on CPU -
loss = model.calculate_loss(batch); loss.backward() (traffic_state_executor.py:411-422) - with its dropout
(MultiATGCN.py:416) fed from a stored mask, so that the step is reproducible elsewhere.  Stored (data only):
the dropout mask (packed bits), the loss, d loss / d prediction and the gradient of every parameter - whole for
tensors up to 20000 elements, as every 17th element of the flattened tensor plus [sum, sum of |.|] otherwise.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference)

NAMES = ["tiny_multi_uni_c2", "tiny_multi_bid_c2", "tiny_od_non_c2", "tiny_multi_uni_dyn7", "tiny_multi_uni_c1",
         "tiny_identity_non_c1", "tiny_multi_uni_c2_static", "tiny_cosine_non_c3_static",
         "hid32_multi_uni_c2", "hid32_multi_uni_c2_static", "hid48_gcnoff",
         # the headline shapes (B = 4): the padding paths of the backward (403 -> 416 / 448 rows, K = T*B per node)
         "dc237_out12", "bm403_out24",
         # head layouts of run_model_parameter.py:6-7 and the one-channel input
         "tiny_heads_100", "tiny_heads_001", "tiny_heads_331", "tiny_heads_113", "tiny_heads_011", "tiny_notid_c2"]
SUB = 17


def run(case):
    model, df, cfg, state = mg.build_reference(case)
    model.train()
    n, b = case["nodes"], case["batch"]
    x, y = mg.syn.make_batch_arrays(b, n, case["out"], case["seed"], feat=case["feat"],
                                    x_steps=24 * sum(case.get("lens", (2, 1, 1))))
    rng = np.random.default_rng(case["seed"] + 5)
    hid = case.get("flags", {}).get("rnn_units", 64)   # the 64-wide draw is kept for the older fixtures: same bits
    mask = ((rng.random((b, 24, n, hid)) >= 0.1).astype(np.float32) / np.float32(0.9)).astype(np.float32)
    held = {}
    real_dropout = mg.REF.F.dropout

    def fixed_dropout(inp, p=0.5, training=True, inplace=False):
        assert training and abs(p - 0.1) < 1e-12 and tuple(inp.shape) == mask.shape
        return inp * torch.from_numpy(mask)

    real_predict = model.predict

    def spy(batch):
        out = real_predict(batch)
        out.retain_grad()
        held["pred"] = out
        return out

    mg.REF.F.dropout = fixed_dropout
    model.predict = spy
    try:
        with mg.pinned_pca(model, case["seed"]):     # the same basis v the forward fixture stores
            loss = model.calculate_loss({"X": torch.from_numpy(x), "y": torch.from_numpy(y).clone()})
            loss.backward()
    finally:
        mg.REF.F.dropout = real_dropout
    out = {"drop_bits": np.packbits(mask > 0), "drop_shape": np.array(mask.shape), "loss": np.float64(loss.item()),
           "d_out": held["pred"].grad.numpy(), "pred": held["pred"].detach().numpy()}
    for k, p in model.named_parameters():
        g = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        if g.size <= 20000:
            out["grad." + k] = g
        else:
            out["gsub." + k] = g.reshape(-1)[::SUB].copy()
            out["gsum." + k] = np.array([g.astype(np.float64).sum(), np.abs(g.astype(np.float64)).sum()])
    return out


def main():
    cases = {c["name"]: c for c in mg.CASES}
    only = set(sys.argv[1:])
    for name in NAMES:
        if only and name not in only:
            continue
        res = run(cases[name])
        np.savez_compressed(os.path.join(HERE, "grad_%s.npz" % name), **res)
        print("%-24s loss %.6f  |d_out|max %.3e  %d gradients" % (
            name, res["loss"], np.abs(res["d_out"]).max(), sum(k.startswith("grad.") or k.startswith("gsub.") for k in res)))


if __name__ == "__main__":
    main()
