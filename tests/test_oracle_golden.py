"""The CPU oracle (oracle/matgcn_oracle.py) against vectors produced by the reference model itself.

This is what pins the oracle: every stage the reference exposes is compared on every golden case.
Tolerance 1e-5 max-normalised (observed: bit-exact to 1e-7).
"""
import numpy as np
import pytest
import torch

from helpers import FULL, TINY, Case, max_norm_err
from oracle import matgcn_oracle as O

TOL = 1e-5


@pytest.mark.parametrize("name", TINY + FULL)
def test_inputs_regenerate_identically(name):
    assert Case(name).checksums_ok()


@pytest.mark.parametrize("name", TINY + FULL)
def test_static_supports(name):
    c = Case(name)
    mats = O.static_supports(c.data_feature["adj_mx"], c.data_feature["coordinate"], None, c.adjtype)
    got = np.stack(mats, 0)
    assert got.shape == c.gold["static_supports"].shape
    assert np.abs(got - c.gold["static_supports"]).max() <= 1e-6


def test_laplacian_known_answer():
    # L~ == -D^-1/2 A^T D^-1/2 with D = row sums; zero-degree rows give zeros
    rng = np.random.default_rng(3)
    a = rng.random((9, 9)).astype(np.float32)
    a[4] = 0.0
    d = a.sum(1)
    dis = np.where(d > 0, d ** -0.5, 0.0)
    want = -(dis[:, None] * a.T * dis[None, :])
    assert np.abs(O.scaled_laplacian(a) - want).max() <= 1e-6


def test_od_normalisation_is_per_column():
    a = np.array([[2.0, 1.0], [4.0, 8.0]], dtype=np.float32)
    got = O.od_adjacency(a)
    assert np.allclose(got, [[1.0, 0.125], [1.0, 1.0]])


@pytest.mark.parametrize("name", TINY)
def test_stages_tiny(name):
    c = Case(name)
    g = c.gold
    p = O.to_tensors(c.state)
    st = O.supports_as_tensors(g["static_supports"])
    use_static = st if (c.adpadj == "none" or c.adjtype == "multi") else []
    for nm in ("gate", "update"):
        stack = O.support_stack(p, st, c.adjtype, c.adpadj, c.cheb, p["encoder.agru_cells.0.%s.weights_g" % nm])
        assert stack.shape[0] == c.k_total
        assert max_norm_err(stack.numpy(), g["stack_l0_%s" % nm]) <= TOL
    xs, hs, xs1 = (torch.from_numpy(g[k]) for k in ("stage_x", "stage_h", "stage_x1"))
    y = O.agcn(torch.cat((xs, hs), -1), p, "encoder.agru_cells.0.gate.", st, c.adjtype, c.adpadj, c.cheb)
    assert max_norm_err(y.numpy(), g["agcn_gate_l0"]) <= TOL
    h1 = O.atgru_cell(xs, hs, p, "encoder.agru_cells.0.", st, c.adjtype, c.adpadj, c.cheb)
    assert max_norm_err(h1.numpy(), g["cell_l0"]) <= TOL
    h2 = O.atgru_cell(xs1, hs, p, "encoder.agru_cells.1.", st, c.adjtype, c.adpadj, c.cheb)
    assert max_norm_err(h2.numpy(), g["cell_l1"]) <= TOL
    r = O.dense_gru_cell(xs, hs, p, "encoder.res_cells.0.")
    assert max_norm_err(r.numpy(), g["res_l0"]) <= TOL
    for faithful in (True, False):
        pred, stg = O.forward(torch.from_numpy(c.x), p, st, c.oracle_cfg(), faithful, True)
        assert max_norm_err(stg["x0"].numpy(), g["x0"]) <= TOL
        assert max_norm_err(stg["seq"].numpy(), g["enc_seq"]) <= TOL
        assert max_norm_err(stg["finals"].numpy(), g["enc_finals"]) <= TOL
        assert max_norm_err(pred.numpy(), g["pred"]) <= TOL
    del use_static


@pytest.mark.parametrize("name", TINY + FULL)
def test_prediction_loss_and_mae(name):
    c = Case(name)
    g = c.gold
    p = O.to_tensors(c.state)
    st = O.supports_as_tensors(g["static_supports"])
    pred = O.forward(torch.from_numpy(c.x), p, st, c.oracle_cfg(), faithful=False)
    assert pred.shape == g["pred"].shape
    assert max_norm_err(pred.numpy(), g["pred"]) <= TOL
    loss = O.calculate_loss(torch.from_numpy(c.x), torch.from_numpy(c.y), p, st, c.oracle_cfg(), faithful=False)
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    ytrue = torch.from_numpy(c.y)[..., 0:1]
    for i in range(c.out):
        mae = O.horizon_mae(torch.from_numpy(g["pred"]), ytrue, i + 1).item()
        assert abs(mae - g["mae_at"][i]) <= 1e-5 * abs(g["mae_at"][i])


def test_fp64_gap_is_small():
    # the tolerance budget: fp32 oracle vs fp64 oracle on a tiny case
    c = Case("tiny_multi_uni_c2")
    st32 = O.supports_as_tensors(c.gold["static_supports"])
    st64 = O.supports_as_tensors(c.gold["static_supports"], torch.float64)
    a = O.forward(torch.from_numpy(c.x), O.to_tensors(c.state), st32, c.oracle_cfg(), False)
    b = O.forward(torch.from_numpy(c.x).double(), O.to_tensors(c.state, torch.float64), st64, c.oracle_cfg(), False)
    assert max_norm_err(a.numpy(), b.numpy()) <= 2e-5
