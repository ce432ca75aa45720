"""The drop-in MultiATGCN class end to end on the GPU: predict, calculate_loss, MAE@k."""
import numpy as np
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


def test_grad_mode_is_refused(lib_built):
    c = Case("tiny_multi_uni_c2")
    m, dev = _model(c)
    with pytest.raises(NotImplementedError):
        m.predict({"X": torch.from_numpy(c.x).to(dev)})
