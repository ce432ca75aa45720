"""The CPU oracle (oracle/matgcn_oracle.py) against vectors produced by the reference model itself.

This is what pins the oracle: every stage the reference exposes is compared on every golden case.
Tolerance 1e-5 max-normalised (observed: bit-exact to 1e-7).
"""
import os

import numpy as np
import pytest
import torch

from helpers import FULL, GOLDEN_DIR, HID, TINY, Case, max_norm_err
from oracle import matgcn_oracle as O

TOL = 1e-5


def _static_h0(c, p, dtype=torch.float32):
    """initial state of the static-feature cases from the fixture's PCA basis (None for the others)"""
    if c.static_dim == 0:
        return None
    return O.static_initial_state(torch.as_tensor(c.static).to(dtype), torch.as_tensor(c.gold["pca_v"]).to(dtype), p)


@pytest.mark.parametrize("name", TINY + FULL + HID)
def test_inputs_regenerate_identically(name):
    assert Case(name).checksums_ok()


@pytest.mark.parametrize("name", TINY + FULL + HID)
def test_static_supports(name):
    c = Case(name)
    mats = O.static_supports(c.data_feature["adj_mx"], c.data_feature["coordinate"], c.static, c.adjtype)
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
        # cheb_order = 1 keeps one stack entry per first-order support but a single weight entry (:65-70,94-100)
        assert stack.shape[0] == (c.k_total if c.cheb > 1 else g["stack_l0_%s" % nm].shape[0]) and \
            p["encoder.agru_cells.0.%s.weights_pool" % nm].shape[1] == c.k_total
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
    h0 = _static_h0(c, p)
    if h0 is not None:      # the (N, H) initial state itself: relu(Linear(static @ v)) with the recorded PCA basis
        assert max_norm_err(h0.numpy(), g["h0"]) <= TOL
    for faithful in (True, False):
        # the encoder stages of the fixtures start from the zero state (init_hidden), the prediction from h0
        _, stg = O.forward(torch.from_numpy(c.x), p, st, c.oracle_cfg(), faithful, True)
        assert max_norm_err(stg["x0"].numpy(), g["x0"]) <= TOL
        assert max_norm_err(stg["seq"].numpy(), g["enc_seq"]) <= TOL
        assert max_norm_err(stg["finals"].numpy(), g["enc_finals"]) <= TOL
        pred = O.forward(torch.from_numpy(c.x), p, st, c.oracle_cfg(), faithful, h0=h0)
        assert max_norm_err(pred.numpy(), g["pred"]) <= TOL
    del use_static


@pytest.mark.parametrize("name", TINY + FULL + HID)
def test_prediction_loss_and_mae(name):
    c = Case(name)
    g = c.gold
    p = O.to_tensors(c.state)
    st = O.supports_as_tensors(g["static_supports"])
    h0 = _static_h0(c, p)
    pred = O.forward(torch.from_numpy(c.x), p, st, c.oracle_cfg(), faithful=False, h0=h0)
    assert pred.shape == g["pred"].shape
    assert max_norm_err(pred.numpy(), g["pred"]) <= TOL
    loss = O.calculate_loss(torch.from_numpy(c.x), torch.from_numpy(c.y), p, st, c.oracle_cfg(), faithful=False, h0=h0)
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


GRAD_CASES = sorted(f[5:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("grad_"))


@pytest.mark.parametrize("name", GRAD_CASES)
def test_oracle_autograd_matches_reference_gradients(name):
    """the gradient oracle of tests/test_backward_gpu.py (torch autograd through the restatement) against the
    reference's own training-mode step (tests/golden/make_grad_golden.py), dropout fed from the stored mask"""
    from oracle import matgcn_oracle as orc
    c = Case(name)
    gold = np.load(os.path.join(GOLDEN_DIR, "grad_%s.npz" % name))
    shape = tuple(int(v) for v in gold["drop_shape"])
    mask = np.unpackbits(gold["drop_bits"])[:int(np.prod(shape))].reshape(shape).astype(np.float64) / 0.9
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in c.state.items()}
    use_static = c.adpadj == "none" or c.adjtype == "multi"
    statics = orc.supports_as_tensors(c.gold["static_supports"], torch.float64) if use_static else []
    cfg = c.oracle_cfg()
    x0 = orc.fuse_heads(torch.tensor(c.x, dtype=torch.float64), p, cfg)
    init = torch.zeros(2, c.b, c.n, c.flags.get("rnn_units", 64), dtype=torch.float64)
    h0 = _static_h0(c, p, torch.float64)       # its gradient reaches static_initial_gru.embd.*
    if h0 is not None:
        init = h0.expand(2, c.b, -1, -1)
    seq, _ = orc.encoder(x0, init, p, statics, cfg["adjtype"], cfg["adpadj"], cfg["cheb_order"], 2, faithful=False,
                         gcn_off=bool(cfg.get("gcn_off", False)))
    y = orc.output_head(seq * torch.tensor(mask), p, c.out, 1)
    assert max_norm_err(y.detach().numpy(), gold["pred"]) <= 1e-5
    (y * torch.tensor(gold["d_out"], dtype=torch.float64)).sum().backward()
    for k, v in p.items():
        g = v.grad.numpy() if v.grad is not None else np.zeros(v.shape)
        if "grad." + k in gold:
            got, w = g, gold["grad." + k]
        else:
            got, w = g.reshape(-1)[::17], gold["gsub." + k]
        if np.abs(w).max() == 0.0:
            assert np.abs(got).max() <= 1e-9, k
        else:
            assert max_norm_err(got, w) <= 2e-5, k
