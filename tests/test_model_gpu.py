"""The drop-in MultiATGCN class end to end on the GPU: predict, calculate_loss, MAE@k."""
import pytest
import torch

from helpers import FULL, Case, max_norm_err

pytestmark = pytest.mark.gpu


def _model(c):
    from multistgraph_amd.model import MultiATGCN
    dev = torch.device("cuda:0")
    m = MultiATGCN(c.config("cuda:0"), c.data_feature).to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
    return m, dev


@pytest.mark.parametrize("name", ["tiny_multi_uni_c2", "tiny_od_non_c2", "tiny_multi_uni_dyn7"] + FULL)
def test_predict_loss_mae(name, lib_built):
    from oracle import matgcn_oracle as O
    c = Case(name)
    m, dev = _model(c)
    batch = {"X": torch.from_numpy(c.x).to(dev), "y": torch.from_numpy(c.y).to(dev)}
    with torch.no_grad():
        pred = m.predict(batch)
        loss = m.calculate_loss(batch)
    assert max_norm_err(pred.cpu().numpy(), c.gold["pred"]) <= 1e-4
    assert abs(loss.item() - float(c.gold["loss"])) <= 1e-4 * abs(float(c.gold["loss"]))
    ytrue = torch.from_numpy(c.y)[..., 0:1]
    for i in range(c.out):
        mae = O.horizon_mae(pred.cpu(), ytrue, i + 1).item()
        assert abs(mae - c.gold["mae_at"][i]) <= 1e-4  # "MAE within 1e-4 of reference"


def test_parameter_update_invalidates_prepared(lib_built):
    c = Case("tiny_multi_uni_c2")
    m, dev = _model(c)
    batch = {"X": torch.from_numpy(c.x).to(dev)}
    with torch.no_grad():
        a = m.predict(batch).clone()
        m.node_emb.mul_(1.5)
        b = m.predict(batch)
    assert (a - b).abs().max().item() > 1e-4


def test_cpu_input_fails_loudly(lib_built):
    c = Case("tiny_multi_uni_c2")
    m, dev = _model(c)
    with torch.no_grad(), pytest.raises(RuntimeError):
        m.predict({"X": torch.from_numpy(c.x)})


def test_training_mode_without_autograd_is_refused(lib_built):
    c = Case("tiny_multi_uni_c2")
    m, dev = _model(c)
    m.train()
    with torch.no_grad(), pytest.raises(NotImplementedError):   # dropout without autograd: not an inference mode
        m.predict({"X": torch.from_numpy(c.x).to(dev)})


def test_fused_loss_and_horizon_mae_with_a_real_scaler(lib_built):
    """matgcn_masked_mae (de-scale + mask + reduce on the device) vs the torch arithmetic of the reference
    (loss.py:17-29) with Baltimore's flow statistics as scaler and some exactly-zero / tiny labels."""
    from multistgraph_amd import synthetic as syn
    from multistgraph_amd.model import MultiATGCN, masked_mae
    c = Case("tiny_multi_uni_out12")
    dev = torch.device("cuda:0")
    scaler = syn.PlainScaler(14.41, 29.3)
    m = MultiATGCN(c.config("cuda:0"), dict(c.data_feature, scaler=scaler)).to(dev).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
    y = torch.from_numpy(c.y).clone()
    y[0, :, :3, 0] = -14.41 / 29.3            # de-scales to exactly 0 -> masked out of the loss
    y[1, 2, 5, 0] = (5e-5 - 14.41) / 29.3     # |label| < 1e-4 after de-scaling -> zeroed, then masked
    batch = {"X": torch.from_numpy(c.x).to(dev), "y": y.to(dev)}
    with torch.no_grad():
        pred = m.predict(batch)
        loss = m.calculate_loss(batch)
        mae = m.horizon_mae(batch)
    p = scaler.inverse_transform(pred.cpu())
    want = masked_mae(p, scaler.inverse_transform(y[..., 0:1].clone()), 0)
    assert abs(loss.item() - want.item()) <= 2e-6 * abs(want.item())
    for k in range(c.out):
        w = masked_mae(p[:, k], scaler.inverse_transform(y[:, k, :, 0:1].clone()))
        assert abs(mae[k].item() - w.item()) <= 2e-6 * abs(w.item())
    # a scaler that is not affine falls back to the reference's torch arithmetic
    class Cubic:
        def inverse_transform(self, d):
            return d * d * d
    m._scaler = Cubic()
    with torch.no_grad():
        l2 = m.calculate_loss(batch)
    assert torch.isfinite(l2)


def test_fused_loss_gradient_matches_torch_autograd(lib_built):
    """matgcn_masked_mae_grad vs autograd through the reference's torch arithmetic (loss.py:17-29) on the same
    prediction: real scaler (mean 14.41, std 29.3), labels that de-scale to exactly 0 (masked out), a prediction that
    equals its label (sign 0), and an upstream factor other than 1"""
    from multistgraph_amd.model import masked_mae
    from multistgraph_amd.ops import masked_mae_loss
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    b, out, n = 5, 6, 33
    mean, std = 14.41, 29.3
    pred = torch.randn(b, out, n, 1, generator=g)
    y = torch.randn(b, out + 2, n, 3, generator=g)
    y[0, :, :4, 0] = -mean / std                 # de-scales to 0 -> masked
    pred[1, 2, 7, 0] = y[1, 2, 7, 0]              # |p - l| = 0 exactly -> sign 0
    p1 = pred.to(dev).requires_grad_(True)
    loss = masked_mae_loss(p1, y.to(dev), 0, mean, std, null_val=0.0)
    (3.0 * loss).backward()
    p2 = pred.clone().requires_grad_(True)
    want = masked_mae(p2 * std + mean, y[:, :out, :, 0:1].clone() * std + mean, 0.0)
    (3.0 * want).backward()
    assert abs(float(loss) - float(want)) <= 2e-6 * abs(float(want))
    assert max_norm_err(p1.grad.cpu().numpy(), p2.grad.numpy()) <= 1e-6
    assert float(p1.grad[0, :, :4].abs().max()) == 0.0 and float(p1.grad[1, 2, 7, 0]) == 0.0


def test_lazy_prepare_is_bitwise_the_eager_prepare(lib_built):
    """matgcn_set_lazy_prepare(1) (on in this binding): matgcn_prepare leaves its weight streams running behind events
    and every consumer waits for what it reads.  The same sequence of calls - parameter updates between forwards, two
    batch sizes (two `prepared` buffers) interleaved, a training step in between, the read-back helpers, serial and
    wavefront schedules - must give bitwise the results of the eager prepare (the C default)."""
    from multistgraph_amd import _lib
    from multistgraph_amd.model import MultiATGCN
    c = Case("tiny_multi_uni_c2")
    dev = torch.device("cuda:0")
    lib = _lib.load()

    def run(lazy):
        prev = lib.matgcn_set_lazy_prepare(1 if lazy else 0)
        try:
            torch.manual_seed(3)
            m = MultiATGCN(c.config("cuda:0"), c.data_feature).to(dev)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
            m.cache_prepared = False                       # prepare in front of EVERY forward, as the bench does
            x2 = torch.from_numpy(c.x).to(dev)
            x6 = torch.cat([x2, x2.flip(0), x2 * 0.5], 0).contiguous()
            y2 = torch.from_numpy(c.y).to(dev)
            outs = []
            m.eval()
            with torch.no_grad():
                outs.append(m.predict({"X": x2}).clone())
                outs.append(m.predict({"X": x6}).clone())                  # a second HotPath / prepared buffer
                m.node_emb.mul_(1.25)                                       # parameter update right after a forward
                outs.append(m.predict({"X": x2}).clone())
                hp = m._paths[2]
                outs.append(hp.supports().clone())                          # read-back helpers join first
                outs.append(hp.node_weights(1, 0).clone())
                lib.matgcn_set_wavefront(0)
                outs.append(m.predict({"X": x6}).clone())
                lib.matgcn_set_wavefront(1)
            m.train()
            torch.manual_seed(5)
            loss = m.calculate_loss({"X": x2, "y": y2})
            loss.backward()
            outs.append(loss.detach().clone())
            outs.append(m.node_emb.grad.clone())
            with torch.no_grad():
                m.encoder.weights_gru.add_(0.1)
            m.eval()
            with torch.no_grad():
                outs.append(m.predict({"X": x2}).clone())
            torch.cuda.synchronize()
            return outs
        finally:
            lib.matgcn_set_wavefront(1)
            lib.matgcn_set_lazy_prepare(prev)

    eager, lazy = run(False), run(True)
    assert len(eager) == len(lazy)
    for i, (a, b) in enumerate(zip(eager, lazy)):
        if i == 7:      # the node-embedding gradient accumulates with atomics (split-K): equal to rounding, not bitwise
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), i
        else:
            assert torch.equal(a, b), i
