import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from multistgraph_amd import graph_prep, synthetic as syn, _lib
from multistgraph_amd.ops import HotPath, diagonal_mask, spec_from_config
from oracle import matgcn_oracle as orc
from helpers import max_norm_err
n, b, layers = 21, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
df = syn.make_data_feature(n, 3, "DC", ext_dim=1)
cfg = dict(input_window=24, output_window=6, add_time_in_day=True, add_day_in_week=False, load_dynamic=False,
           adjtype="multi", adpadj="unidirection", cheb_order=2, embed_dim_node=20, embed_dim_adj=20, rnn_units=64,
           num_layers=layers, device=torch.device("cpu"), batch_size=b)
mats = graph_prep.build_static_supports(df["adj_mx"], df["coordinate"], None, "multi")
st = torch.from_numpy(np.stack(mats, 0))
shapes = syn.param_shapes(n, out_steps=6, feat_in=2, k_total=5, layers=layers)
state_np = syn.closed_form_state(shapes, 3)
x_np, _ = syn.make_batch_arrays(b, n, 6, 3, feat=2)
rng = np.random.default_rng(9)
d_out = rng.standard_normal((b, 6, n, 1)).astype(np.float32)
p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in state_np.items()}
ocfg = dict(adjtype="multi", adpadj="unidirection", cheb_order=2, num_layers=layers, rnn_units=64, len_closeness=48,
            len_period=24, len_trend=24, output_window=6, input_window=24, add_time_in_day=True,
            add_day_in_week=False, load_dynamic=False, start_dim=0, end_dim=1)
y = orc.forward(torch.tensor(x_np, dtype=torch.float64), p, [m.double() for m in st], ocfg, faithful=False)
(y * torch.tensor(d_out, dtype=torch.float64)).sum().backward()
for mode in (0, 1):
    _lib.load().matgcn_set_wavefront(mode)
    spec = spec_from_config(cfg, df, n, min(n, 20), 3, diagonal_mask(st))
    hp = HotPath(spec, b, dev)
    state = {k: torch.from_numpy(v).to(dev) for k, v in state_np.items()}
    hp.bind(state, st.to(dev))
    x = torch.from_numpy(x_np).to(dev)
    hp.forward_train(x)
    grads = hp.backward(x, torch.from_numpy(d_out).to(dev), state)
    print("wavefront", mode)
    for k, v in p.items():
        w = v.grad.numpy() if v.grad is not None else np.zeros(v.shape)
        if np.abs(w).max() > 0:
            e = max_norm_err(grads[k].cpu().numpy(), w)
            if e > 1e-5: print("   %-48s %.3e" % (k, e))
