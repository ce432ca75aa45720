import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import TINY, Case
from multistgraph_amd.ops import HotPath, diagonal_mask, spec_from_config
dev = torch.device("cuda:0")
for name in TINY:
    c = Case(name)
    use_static = c.adpadj == "none" or c.adjtype == "multi"
    st = torch.from_numpy(c.gold["static_supports"]).to(dev) if use_static else None
    mask = diagonal_mask(st)
    spec = spec_from_config(c.config(), c.data_feature, c.n, min(c.n, 20), st.shape[0] if use_static else 0, mask)
    print(name, "N", c.n, "mask", mask, "n_static", spec.n_static, flush=True)
    hp = HotPath(spec, c.b, dev)
    hp.bind({k: torch.from_numpy(v).to(dev) for k, v in c.state.items()}, st)
    hp.prepare()
    torch.cuda.synchronize()
    print("  prepare ok", flush=True)
    s = hp.supports()
    torch.cuda.synchronize()
    print("  supports ok", tuple(s.shape), flush=True)
