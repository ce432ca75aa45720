"""rnn_units below 64 (MultiATGCN.py:322; the commented sweep of run_model_parameter.py:11) on the 64-wide kernels:
multistgraph_amd/hidden_pad.py zero-pads every hidden axis.  CPU: the padding is exact (the oracle on the padded state at
64 channels = the oracle on the real state = the reference's golden prediction) and its autograd hands back slices.
GPU: the plugin class against the reference's fixtures, forward and training step."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR, HID, Case, max_norm_err
from multistgraph_amd import hidden_pad


def _padded_oracle_inputs(c):
    from oracle import matgcn_oracle as O
    p = O.to_tensors(c.state)
    hid = c.flags["rnn_units"]
    keep = {k: v for k, v in p.items() if not k.startswith("static_initial")}
    wide = hidden_pad.pad_state(keep, hid, c.feat)
    cfg = dict(c.oracle_cfg(), rnn_units=64)
    h0 = None
    if c.static_dim:
        h0n = O.static_initial_state(torch.as_tensor(c.static), torch.as_tensor(c.gold["pca_v"]), p)
        h0 = hidden_pad.pad_last(h0n)
    return O, p, wide, cfg, h0


@pytest.mark.parametrize("name", HID)
def test_padding_is_exact_on_the_oracle(name):
    c = Case(name)
    O, p, wide, cfg, h0 = _padded_oracle_inputs(c)
    st = O.supports_as_tensors(c.gold["static_supports"])
    pred = O.forward(torch.from_numpy(c.x), wide, st, cfg, faithful=False, h0=h0)
    assert max_norm_err(pred.numpy(), c.gold["pred"]) <= 1e-5
    from multistgraph_amd import synthetic as syn
    want = syn.param_shapes(c.n, out_steps=c.out, feat_in=c.feat, k_total=c.k_total, static=False,
                            **dict(c.flags, rnn_units=64))
    for k, v in wide.items():     # shapes are those of the same model at 64 channels
        assert tuple(v.shape) == tuple(want[k]), k


def test_padding_backpropagates_slices():
    c = Case("hid32_multi_uni_c2")
    real = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in c.state.items()}
    wide = hidden_pad.pad_state(real, 32, c.feat)
    rng = np.random.default_rng(0)
    probe = {k: torch.from_numpy(rng.standard_normal(tuple(v.shape))) for k, v in wide.items()}
    sum((wide[k] * probe[k]).sum() for k in wide).backward()
    k = "encoder.agru_cells.1.gate.weights_pool"       # (d, K, [x 32 | h 32], [z 32 | r 32]) inside (d, K, 128, 128)
    g, q = real[k].grad, probe[k]
    assert torch.equal(g[:, :, :32, :32], q[:, :, :32, :32]) and torch.equal(g[:, :, 32:, 32:], q[:, :, 64:96, 64:96])
    assert torch.equal(g[:, :, :32, 32:], q[:, :, :32, 64:96]) and torch.equal(g[:, :, 32:, :32], q[:, :, 64:96, :32])
    k = "encoder.agru_cells.0.update.weights_pool"     # layer 0: (d, K, [x 2 | h 32], 32) inside (d, K, 2 + 64, 64)
    assert torch.equal(real[k].grad, probe[k][:, :, :34, :32])
    k = "end_conv.weight"
    assert torch.equal(real[k].grad, probe[k][..., :32])
    assert torch.equal(real["node_emb"].grad, probe["node_emb"])


def test_wider_than_the_kernels_is_refused():
    from multistgraph_amd.model import MultiATGCN
    c = Case("tiny_multi_uni_c2")
    with pytest.raises(NotImplementedError):
        MultiATGCN(dict(c.config(), rnn_units=72), c.data_feature)


def _model(c, dev):
    from multistgraph_amd.model import MultiATGCN
    m = MultiATGCN(c.config("cuda:0"), c.data_feature).to(dev)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})   # the reference's shapes: checkpoint ABI
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("name", HID)
def test_plugin_forward(name, lib_built, monkeypatch):
    from oracle import matgcn_oracle as O
    c = Case(name)
    dev = torch.device("cuda:0")
    m = _model(c, dev).eval()
    if c.static_dim:
        v = torch.from_numpy(c.gold["pca_v"]).to(dev)
        monkeypatch.setattr(torch, "pca_lowrank", lambda A, q=None, center=True, niter=2: (None, None, v))
    batch = {"X": torch.from_numpy(c.x).to(dev), "y": torch.from_numpy(c.y).to(dev)}
    with torch.no_grad():
        pred = m.predict(batch)
        loss = m.calculate_loss(batch)
    assert tuple(pred.shape) == c.gold["pred"].shape
    assert max_norm_err(pred.cpu().numpy(), c.gold["pred"]) <= 1e-4
    assert abs(loss.item() - float(c.gold["loss"])) <= 1e-4 * abs(float(c.gold["loss"]))
    ytrue = torch.from_numpy(c.y)[..., 0:1]
    for i in range(c.out):
        assert abs(O.horizon_mae(pred.cpu(), ytrue, i + 1).item() - c.gold["mae_at"][i]) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(f[5:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("grad_hid")))
def test_plugin_training_step(name, lib_built, monkeypatch):
    """loss and the gradient of EVERY reference-shaped parameter against the reference's own training step"""
    from test_backward_gpu import _check_against_fixture, _fixture_mask
    c = Case(name)
    gold = np.load(os.path.join(GOLDEN_DIR, "grad_%s.npz" % name))
    dev = torch.device("cuda:0")
    m = _model(c, dev)
    if c.static_dim:
        v = torch.from_numpy(c.gold["pca_v"]).to(dev)
        monkeypatch.setattr(torch, "pca_lowrank", lambda A, q=None, center=True, niter=2: (None, None, v))
    m.train()
    mask = torch.from_numpy(_fixture_mask(gold)).to(dev)
    monkeypatch.setattr(torch.nn.functional, "dropout", lambda inp, p=0.5, training=True, inplace=False: inp * mask)
    batch = {"X": torch.from_numpy(c.x).to(dev), "y": torch.from_numpy(c.y).to(dev)}
    loss = m.calculate_loss(batch)
    assert abs(float(loss) - float(gold["loss"])) <= 1e-4 * abs(float(gold["loss"]))
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in m.named_parameters()}
    assert all(tuple(g.shape) == tuple(c.state[k].shape) for k, g in grads.items())
    bad = _check_against_fixture(gold, grads)
    assert not bad, bad
    # and it trains: an optimizer step on the reference-shaped parameters lowers the loss
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.zero_grad()
    l2 = m.calculate_loss(batch)
    l2.backward()
    opt.step()
    assert float(m.calculate_loss(batch)) < float(l2)
