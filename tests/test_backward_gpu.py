"""Training step: matgcn_forward_train + matgcn_backward vs torch autograd
through the CPU oracle (oracle/matgcn_oracle.py, itself pinned to the reference's golden vectors; the
gradient fixtures tests/golden/grad_*.npz come from autograd through the reference model itself).

Tolerance: 1e-4 max-normalised per parameter tensor (fp32; the oracle side runs in fp64).
"""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN_DIR, TINY, Case, max_norm_err

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-4
from helpers import FULL  # noqa: E402
ABLATIONS = sorted(n for n in FULL if n.startswith("abl_"))


def _path(c, fold=True):
    from multistgraph_amd.ops import HotPath, diagonal_mask, spec_from_config
    dev = torch.device("cuda:0")
    use_static = c.adpadj == "none" or c.adjtype == "multi"
    st = torch.from_numpy(c.gold["static_supports"]).to(dev) if use_static else None
    spec = spec_from_config(c.config(), c.data_feature, c.n, min(c.n, 20), st.shape[0] if use_static else 0,
                            diagonal_mask(st) if fold else 0)
    hp = HotPath(spec, c.b, dev)
    state = {k: torch.from_numpy(v).to(dev) for k, v in c.state.items()}
    hp.bind(state, st)
    return hp, dev, state


D_H0 = "__d_h0__"


def _oracle_grads(c, d_out):
    """d(sum(out * d_out))/d(param) by autograd through the oracle in fp64.  Static-feature cases: the initial state
    (L, B, N, H) is a leaf of its own (key D_H0) - what the C ABI returns; the host-side static_initial_* layers are
    not the HIP path's business and are left out."""
    from oracle import matgcn_oracle as orc
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in c.state.items()}
    use_static = c.adpadj == "none" or c.adjtype == "multi"
    statics = orc.supports_as_tensors(c.gold["static_supports"], torch.float64) if use_static else []
    h0 = c.h0()
    cfg = c.oracle_cfg()
    if h0 is None:
        y = orc.forward(torch.tensor(c.x, dtype=torch.float64), p, statics, cfg, faithful=False)
    else:
        h0 = h0.double().requires_grad_(True)
        x0 = orc.fuse_heads(torch.tensor(c.x, dtype=torch.float64), p, cfg)
        seq, _ = orc.encoder(x0, h0, p, statics, cfg["adjtype"], cfg["adpadj"], cfg["cheb_order"], 2, faithful=False)
        y = orc.output_head(seq, p, c.out, 1)
    (y * torch.tensor(d_out, dtype=torch.float64)).sum().backward()
    want = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in p.items()
            if not k.startswith("static_initial")}
    if h0 is not None:
        want[D_H0] = h0.grad.numpy()
    return y.detach().numpy(), want


def _reference_gemm(a, b, c, desc, alpha, beta):
    (M, N, K, K2, sAm, sAk, sAk2, sBk, sBn, sBk2, sCm, sCn, nb1, nb2, bA1, bA2, bB1, bB2, bC1, bC2, mode, split) = desc
    out = c.astype(np.float64).copy()
    m, n, k, k2 = np.arange(M), np.arange(N), np.arange(K), np.arange(K2)
    for b1 in range(nb1):
        for b2 in range(nb2):
            ia = b1 * bA1 + b2 * bA2 + m[:, None, None] * sAm + k2[None, :, None] * sAk2 + k[None, None, :] * sAk
            ib = b1 * bB1 + b2 * bB2 + k2[:, None, None] * sBk2 + k[None, :, None] * sBk + n[None, None, :] * sBn
            acc = np.einsum("mqk,qkn->mn", a[ia].astype(np.float64), b[ib].astype(np.float64))
            ic = b1 * bC1 + b2 * bC2 + m[:, None] * sCm + n[None, :] * sCn
            out[ic] = alpha * acc + (out[ic] if mode == 1 else beta * out[ic])
    return out


@pytest.mark.parametrize("case", [
    # M, N, K, K2, sAm, sAk, sAk2, sBk, sBn, sBk2, sCm, sCn, nb1, nb2, bA1.., mode, split
    dict(M=70, N=33, K=45, K2=1, ta=False, tb=False, nb1=1, nb2=1, mode=0, split=1, beta=0.0),
    dict(M=64, N=64, K=16, K2=1, ta=True, tb=True, nb1=3, nb2=2, mode=0, split=1, beta=0.5),
    dict(M=5, N=130, K=7, K2=6, ta=True, tb=False, nb1=2, nb2=1, mode=0, split=1, beta=1.0),
    dict(M=100, N=20, K=37, K2=9, ta=False, tb=True, nb1=1, nb2=3, mode=1, split=5, beta=0.0),
    dict(M=1, N=1, K=1, K2=1, ta=False, tb=False, nb1=1, nb2=1, mode=1, split=3, beta=0.0),
    # 16-byte friendly strides: the float4 tile loads, with ragged edges falling back per quad
    dict(M=128, N=64, K=48, K2=2, ta=False, tb=False, nb1=2, nb2=1, mode=0, split=1, beta=0.0, pad4=True),
    dict(M=70, N=34, K=45, K2=3, ta=True, tb=True, nb1=1, nb2=2, mode=0, split=1, beta=1.0, pad4=True),
    dict(M=66, N=130, K=18, K2=1, ta=True, tb=False, nb1=1, nb2=1, mode=1, split=2, beta=0.0, pad4=True),
    dict(M=64, N=64, K=64, K2=1, ta=False, tb=True, nb1=3, nb2=1, mode=0, split=1, beta=0.0, pad4=True),
    # k_bgemm_tn fast path (A unit-stride along M, B along N, whole 64 x 64 tiles): K tails inside every k2 block,
    # two-level K, both batch levels, accumulation, split-K with atomics, K smaller than a tile, a single tile
    dict(M=64, N=128, K=45, K2=3, ta=True, tb=False, nb1=2, nb2=2, mode=0, split=1, beta=1.0, pad4=True),
    dict(M=128, N=64, K=130, K2=1, ta=True, tb=False, nb1=1, nb2=1, mode=1, split=3, beta=0.0, pad4=True),
    dict(M=64, N=64, K=5, K2=7, ta=True, tb=False, nb1=1, nb2=3, mode=0, split=1, beta=0.0, pad4=True),
    dict(M=64, N=64, K=16, K2=1, ta=True, tb=False, nb1=1, nb2=1, mode=0, split=1, beta=0.5, pad4=True),
    dict(M=192, N=64, K=100, K2=4, ta=True, tb=False, nb1=1, nb2=1, mode=1, split=7, beta=0.0, pad4=True),
])
def test_strided_batched_gemm(case, lib_built):
    from multistgraph_amd.ops import debug_gemm
    rng = np.random.default_rng(7)
    M, N, K, K2, nb1, nb2 = (case[k] for k in ("M", "N", "K", "K2", "nb1", "nb2"))
    # A element (m, k2, k) and B element (k2, k, n) inside padded per-batch blocks; ta/tb swap the fast axis
    r4 = (lambda v: (v + 3) // 4 * 4) if case.get("pad4") else (lambda v: v)
    if case["ta"]:
        sAk, sAm = r4(M + 3), 1
    else:
        sAm, sAk = r4(K + 2), 1
    sAk2 = r4((K * sAk if case["ta"] else M * sAm) + 8)
    blockA = r4(sAk2 * K2 + 5)
    if case["tb"]:
        sBn, sBk = r4(K + 1), 1
    else:
        sBk, sBn = r4(N + 4), 1
    sBk2 = r4((N * sBn if case["tb"] else K * sBk) + 8)
    blockB = r4(sBk2 * K2 + 3)
    sCm, sCn = N + 2, 1
    blockC = M * sCm + 1
    bA2, bA1 = blockA, blockA * nb2
    bB2, bB1 = blockB, blockB * nb2
    bC2, bC1 = blockC, blockC * nb2
    a = rng.standard_normal(blockA * nb1 * nb2).astype(np.float32)
    b = rng.standard_normal(blockB * nb1 * nb2).astype(np.float32)
    c0 = rng.standard_normal(blockC * nb1 * nb2).astype(np.float32)
    desc = (M, N, K, K2, sAm, sAk, sAk2, sBk, sBn, sBk2, sCm, sCn, nb1, nb2, bA1, bA2, bB1, bB2, bC1, bC2,
            case["mode"], case["split"])
    dev = torch.device("cuda:0")
    ta, tb, tc = (torch.from_numpy(v).to(dev) for v in (a, b, c0))
    debug_gemm(ta, tb, tc, desc, alpha=1.25, beta=case["beta"])
    want = _reference_gemm(a, b, c0, desc, 1.25, case["beta"])
    assert max_norm_err(tc.cpu().numpy(), want) <= 2e-6


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("name", TINY + ABLATIONS)
def test_backward_matches_oracle_autograd(name, fold, lib_built):
    c = Case(name)
    if not fold and (c.adjtype not in ("multi", "cosine", "identity") or c.flags.get("gcn_off")):
        pytest.skip("no diagonal support to fold in this mode")
    hp, dev, state = _path(c, fold)
    rng = np.random.default_rng(c.seed + 11)
    d_out = rng.standard_normal((c.b, c.out, c.n, 1)).astype(np.float32)
    want_y, want = _oracle_grads(c, d_out)
    x = torch.from_numpy(c.x).to(dev)
    h0 = c.h0()
    h0 = None if h0 is None else h0.to(dev)
    y = hp.forward_train(x, None, h0)
    assert max_norm_err(y.cpu().numpy(), want_y) <= 1e-4
    assert torch.equal(y, hp.forward(x, h0))                   # the saving instantiations compute the same forward
    hp.forward_train(x, None, h0)
    grads = hp.backward(x, torch.from_numpy(d_out).to(dev), state, None, h0)
    torch.cuda.synchronize()
    assert set(grads) == set(want)
    worst = {}
    for k, g in grads.items():
        w = want[k]
        if np.abs(w).max() == 0.0:
            assert float(g.abs().max()) <= 1e-6, k
            continue
        worst[k] = max_norm_err(g.cpu().numpy(), w)
    bad = {k: v for k, v in worst.items() if v > GRAD_TOL}
    assert not bad, bad


@pytest.mark.parametrize("name", ["tiny_multi_uni_c2", "tiny_multi_uni_c2_static", "tiny_od_non_c3", "tiny_multi_uni_c1"])
def test_backward_one_stream_equals_three_streams(name, lib_built):
    """matgcn_set_wavefront(0) runs the same kernels on ONE stream (the configuration kernel durations are measured in);
    the default schedule spreads them over three with events in between: both must give the same gradients (not
    bit-for-bit: split-K and the bias by-products accumulate with atomics)."""
    c = Case(name)
    hp, dev, state = _path(c)
    rng = np.random.default_rng(c.seed + 3)
    d_out = torch.from_numpy(rng.standard_normal((c.b, c.out, c.n, 1)).astype(np.float32)).to(dev)
    x = torch.from_numpy(c.x).to(dev)
    h0 = c.h0()
    h0 = None if h0 is None else h0.to(dev)
    res = {}
    prev = hp.lib.matgcn_set_wavefront(1)
    try:
        for mode in (1, 0):
            hp.lib.matgcn_set_wavefront(mode)
            hp.forward_train(x, None, h0)
            res[mode] = {k: v.clone() for k, v in hp.backward(x, d_out, state, None, h0).items()}
            torch.cuda.synchronize()
    finally:
        hp.lib.matgcn_set_wavefront(prev)
    for k, g in res[1].items():
        scale = max(float(g.abs().max()), 1e-30)
        assert float((g - res[0][k]).abs().max()) <= 2e-6 * scale + 1e-9, k


# (grad_hid*: rnn_units < 64 runs through the plugin class only - tests/test_hidden_pad.py)
GRAD_CASES = sorted(f[5:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("grad_") and not f.startswith("grad_hid"))


def _check_against_fixture(gold, grads, tol=GRAD_TOL):
    bad = {}
    for k, g in grads.items():
        g = g.detach().cpu().numpy() if isinstance(g, torch.Tensor) else np.asarray(g)
        if "grad." + k in gold:
            got, w = g, gold["grad." + k]
        else:   # large tensors: every 17th element + [sum, sum |.|]
            got, w = g.reshape(-1)[::17], gold["gsub." + k]
            sums = gold["gsum." + k]
            if abs(float(g.astype(np.float64).sum()) - sums[0]) > 2e-4 * max(sums[1], 1e-30):
                bad[k + " (sum)"] = float(g.astype(np.float64).sum()), float(sums[0])
        if np.abs(w).max() == 0.0:
            if float(np.abs(got).max()) > 1e-6:
                bad[k] = "expected zero"
        elif max_norm_err(got, w) > tol:
            bad[k] = max_norm_err(got, w)
    return bad


def _fixture_mask(gold):
    shape = tuple(int(v) for v in gold["drop_shape"])
    bits = np.unpackbits(gold["drop_bits"])[:int(np.prod(shape))].reshape(shape)
    return (bits.astype(np.float32) / np.float32(0.9)).astype(np.float32)


# the headline shapes also on ONE stream (matgcn_set_wavefront(0)): the schedule kernel durations are measured in
@pytest.mark.parametrize("name,wavefront", [(n, 1) for n in GRAD_CASES] + [("bm403_out24", 0), ("dc237_out12", 0)])
def test_backward_matches_reference_autograd(name, wavefront, lib_built):
    """one training-mode step of the reference itself (calculate_loss(batch).backward() with its dropout fed from
    the stored mask; tests/golden/make_grad_golden.py): prediction and every parameter gradient.  grad_dc237_out12 /
    grad_bm403_out24 are the headline graphs (B = 4): N = 237 / 403 are no multiples of the 16 / 32 / 64-row tiles, so
    the padding paths of the transposed mix (403 -> 416 rows), of the adjacency-gradient GEMM (403 -> 448) and the
    per-step tables of the node weight gradients (K = T*B rows per node) meet the reference's own numbers here."""
    c = Case(name)
    gold = np.load(os.path.join(GOLDEN_DIR, "grad_%s.npz" % name))
    hp, dev, state = _path(c)
    prev = hp.lib.matgcn_set_wavefront(wavefront)
    try:
        _backward_vs_fixture(c, gold, hp, dev, state)
    finally:
        hp.lib.matgcn_set_wavefront(prev)


def _backward_vs_fixture(c, gold, hp, dev, state):
    x = torch.from_numpy(c.x).to(dev)
    mask = torch.from_numpy(_fixture_mask(gold)).to(dev)
    h0 = c.h0()
    h0 = None if h0 is None else h0.to(dev)
    y = hp.forward_train(x, mask, h0)
    assert max_norm_err(y.cpu().numpy(), gold["pred"]) <= 1e-4
    grads = hp.backward(x, torch.from_numpy(gold["d_out"]).to(dev), state, mask, h0)
    d_h0 = grads.pop(D_H0, None)
    if d_h0 is not None:
        # the reference's gradients of static_initial_gru.embd.* are d_h0 pushed through expand + ReLU + nn.Linear:
        # redo those three host-side steps in torch and compare with the fixture
        w = torch.tensor(c.state["static_initial_gru.embd.weight"], dtype=torch.float64, requires_grad=True)
        bvec = torch.tensor(c.state["static_initial_gru.embd.bias"], dtype=torch.float64, requires_grad=True)
        z = torch.tensor(c.static, dtype=torch.float64) @ torch.tensor(c.gold["pca_v"], dtype=torch.float64)
        emb = torch.relu(torch.nn.functional.linear(z, w, bvec))
        emb.expand(2, c.b, -1, -1).backward(d_h0.double().cpu())
        grads["static_initial_gru.embd.weight"], grads["static_initial_gru.embd.bias"] = w.grad, bvec.grad
    bad = _check_against_fixture(gold, grads)
    assert not bad, bad


@pytest.mark.parametrize("name", ["tiny_multi_bid_c2", "tiny_multi_uni_c2", "tiny_multi_uni_c1", "bm403_out24",
                                  "dc237_out12", "tiny_heads_100", "tiny_heads_001", "tiny_heads_331", "tiny_heads_113",
                                  "tiny_heads_011", "tiny_notid_c2"])
def test_plugin_training_step(name, lib_built, monkeypatch):
    """the plugin surface as TrafficStateExecutor._train_epoch drives it (traffic_state_executor.py:411-422):
    model.train(); loss = model.calculate_loss(batch); loss.backward() -> p.grad of every parameter"""
    from multistgraph_amd.model import MultiATGCN
    c = Case(name)
    gold = np.load(os.path.join(GOLDEN_DIR, "grad_%s.npz" % name))
    dev = torch.device("cuda:0")
    model = MultiATGCN(c.config("cuda:0"), c.data_feature).to(dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
    model.train()
    mask = torch.from_numpy(_fixture_mask(gold)).to(dev)
    monkeypatch.setattr(torch.nn.functional, "dropout", lambda inp, p=0.5, training=True, inplace=False: inp * mask)
    batch = {"X": torch.from_numpy(c.x).to(dev), "y": torch.from_numpy(c.y).to(dev)}
    loss = model.calculate_loss(batch)
    assert abs(float(loss) - float(gold["loss"])) <= 1e-4 * abs(float(gold["loss"]))
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
    bad = _check_against_fixture(gold, grads)
    assert not bad, bad
    # a second forward between a forward and its backward is refused, not silently wrong
    l1 = model.calculate_loss(batch)
    model.calculate_loss(batch)
    with pytest.raises(RuntimeError):
        l1.backward()
    # an optimizer step changes the parameters -> the next step re-prepares and the loss moves
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt.zero_grad()
    l2 = model.calculate_loss(batch)
    l2.backward()
    opt.step()
    l3 = model.calculate_loss(batch)
    assert float(l3) < float(l2)


def test_gradient_bucket_views_accumulation_and_zeroing(lib_built, monkeypatch):
    """Every HIP-path gradient of a backward is a view of ONE flat buffer (one all-reduce in data-parallel training, no
    copy).  Gradients handed out earlier must stay valid: gradient accumulation and zero_grad(set_to_none=False) give
    the same numbers as with separate tensors."""
    from multistgraph_amd.model import MultiATGCN
    c = Case("tiny_multi_uni_c2")
    dev = torch.device("cuda:0")
    model = MultiATGCN(c.config("cuda:0"), c.data_feature).to(dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
    model.train()
    monkeypatch.setattr(torch.nn.functional, "dropout", lambda inp, p=0.5, training=True, inplace=False: inp)
    batch = {"X": torch.from_numpy(c.x).to(dev), "y": torch.from_numpy(c.y).to(dev)}
    model.calculate_loss(batch).backward()
    bucket = model.gradient_bucket()
    assert bucket is not None and bucket.dim() == 1
    lo, hi = bucket.data_ptr(), bucket.data_ptr() + bucket.numel() * 4
    assert all(lo <= p.grad.data_ptr() < hi for p in model.parameters())
    first = {k: p.grad.clone() for k, p in model.named_parameters()}

    def close(a, b):
        return float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-12)

    model.calculate_loss(batch).backward()                      # accumulate: .grad still lives in the bucket
    assert all(close(p.grad, 2 * first[k]) for k, p in model.named_parameters())
    for p in model.parameters():
        p.grad.zero_()                                          # zero_grad(set_to_none=False)
    model.calculate_loss(batch).backward()
    assert all(close(p.grad, first[k]) for k, p in model.named_parameters())
    for p in model.parameters():
        p.grad = None                                           # zero_grad() of torch 2
    model.calculate_loss(batch).backward()
    again = model.gradient_bucket()
    assert again is not None and again.numel() == bucket.numel()
    lo, hi = again.data_ptr(), again.data_ptr() + again.numel() * 4
    assert all(lo <= p.grad.data_ptr() < hi for p in model.parameters())
    assert all(close(p.grad, first[k]) for k, p in model.named_parameters())


def test_plugin_training_step_with_static_features(lib_built, monkeypatch):
    """add_static through the plugin surface (MultiATGCN.py:244-250,286-296,335-338,406-409): the randomised
    torch.pca_lowrank of the reference's forward is replayed from the fixture's recorded basis; loss and the gradient
    of EVERY parameter - static_initial_gru.embd.* through the HIP backward's d_h0 included - against the reference's
    own training step; static_initial_node gets none (forward never uses it)."""
    from multistgraph_amd.model import MultiATGCN
    name = "tiny_multi_uni_c2_static"
    c = Case(name)
    gold = np.load(os.path.join(GOLDEN_DIR, "grad_%s.npz" % name))
    dev = torch.device("cuda:0")
    model = MultiATGCN(c.config("cuda:0"), c.data_feature).to(dev)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
    assert model.static.device.type == "cuda"                       # the buffer follows model.to(device)
    v = torch.from_numpy(c.gold["pca_v"]).to(dev)
    monkeypatch.setattr(torch, "pca_lowrank", lambda A, q=None, center=True, niter=2: (None, None, v))
    batch = {"X": torch.from_numpy(c.x).to(dev), "y": torch.from_numpy(c.y).to(dev)}
    model.eval()
    with torch.no_grad():
        assert max_norm_err(model.predict(batch).cpu().numpy(), c.gold["pred"]) <= 1e-4
    model.train()
    mask = torch.from_numpy(_fixture_mask(gold)).to(dev)
    monkeypatch.setattr(torch.nn.functional, "dropout", lambda inp, p=0.5, training=True, inplace=False: inp * mask)
    loss = model.calculate_loss(batch)
    assert abs(float(loss) - float(gold["loss"])) <= 1e-4 * abs(float(gold["loss"]))
    loss.backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in model.named_parameters()}
    assert float(grads["static_initial_gru.embd.weight"].abs().max()) > 0
    bad = _check_against_fixture(gold, grads)
    assert not bad, bad


def test_dropout_mask_of_the_last_step_head(lib_built):
    """fnn_off: the head and its dropout see the last step only (MultiATGCN.py:412-416): mask (B, 1, N, H)"""
    from oracle import matgcn_oracle as orc
    c = Case("abl_fnnoff")
    hp, dev, state = _path(c)
    rng = np.random.default_rng(3)
    mask = ((rng.random((c.b, 1, c.n, 64)) >= 0.1) / 0.9).astype(np.float32)
    d_out = rng.standard_normal((c.b, c.out, c.n, 1)).astype(np.float32)
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in c.state.items()}
    statics = orc.supports_as_tensors(c.gold["static_supports"], torch.float64)
    cfg = c.oracle_cfg()
    x0 = orc.fuse_heads(torch.tensor(c.x, dtype=torch.float64), p, cfg)
    seq, _ = orc.encoder(x0, torch.zeros(2, c.b, c.n, 64, dtype=torch.float64), p, statics, cfg["adjtype"],
                         cfg["adpadj"], cfg["cheb_order"], 2, faithful=False)
    y = orc.output_head(seq[:, -1:] * torch.tensor(mask, dtype=torch.float64), p, c.out, 1)
    (y * torch.tensor(d_out, dtype=torch.float64)).sum().backward()
    x = torch.from_numpy(c.x).to(dev)
    tm = torch.from_numpy(mask).to(dev)
    got_y = hp.forward_train(x, tm)
    assert max_norm_err(got_y.cpu().numpy(), y.detach().numpy()) <= 1e-4
    grads = hp.backward(x, torch.from_numpy(d_out).to(dev), state, tm)
    for k, v in p.items():
        w = v.grad.numpy() if v.grad is not None else np.zeros(v.shape)
        if np.abs(w).max() == 0.0:
            assert float(grads[k].abs().max()) <= 1e-6, k
        else:
            assert max_norm_err(grads[k].cpu().numpy(), w) <= GRAD_TOL, k


def _full_size(batch=64):
    """Baltimore-size path (N = 403, multi + unidirection) with the golden case's parameters and a batch of 64"""
    from multistgraph_amd.ops import HotPath, diagonal_mask, spec_from_config
    c = Case("bm403_out24")
    dev = torch.device("cuda:0")
    st = torch.from_numpy(c.gold["static_supports"]).to(dev)
    cfg = dict(c.config(), batch_size=batch)
    spec = spec_from_config(cfg, c.data_feature, c.n, 20, st.shape[0], diagonal_mask(st))
    hp = HotPath(spec, batch, dev)
    state = {k: torch.from_numpy(v).to(dev) for k, v in c.state.items()}
    hp.bind(state, st)
    rng = np.random.default_rng(5)
    reps = (batch + c.b - 1) // c.b
    x = np.tile(c.x, (reps, 1, 1, 1))[:batch].copy()
    x[..., 0] += 0.05 * rng.standard_normal(x.shape[:-1]).astype(np.float32)
    return c, hp, dev, state, torch.from_numpy(x).to(dev), rng


def test_full_size_backward_is_linear_in_d_out(lib_built):
    """size-independent property at BASELINE size (B=64, N=403): the backward is a linear map of d_out"""
    c, hp, dev, state, x, rng = _full_size()
    d1, d2 = (torch.from_numpy(rng.standard_normal((64, c.out, c.n, 1)).astype(np.float32)).to(dev) for _ in range(2))
    hp.forward_train(x)
    g1 = hp.backward(x, d1, state)
    g2 = hp.backward(x, d2, state)
    g3 = hp.backward(x, 0.5 * d1 - 2.0 * d2, state)
    for k in g1:
        want = (0.5 * g1[k].double() - 2.0 * g2[k].double()).cpu().numpy()
        scale = max(float(g1[k].abs().max()), float(g2[k].abs().max()), 1e-30)
        assert np.abs(g3[k].double().cpu().numpy() - want).max() <= 2e-4 * scale, k
        assert torch.isfinite(g3[k]).all(), k


def test_full_size_directional_derivative(lib_built):
    """size-independent property at BASELINE size: <grad, v> equals the central difference of L = sum(out * d_out)
    along random directions v of a few parameter tensors (fp32 forward: 2 % tolerance)"""
    c, hp, dev, state, x, rng = _full_size()
    d_out = torch.from_numpy(rng.standard_normal((64, c.out, c.n, 1)).astype(np.float32)).to(dev)
    hp.forward_train(x)
    grads = hp.backward(x, d_out, state)

    def loss():
        hp.prepare()
        return float((hp.forward(x).double() * d_out.double()).sum())

    for name in ("node_emb", "node_vec1", "encoder.agru_cells.1.gate.weights_pool", "encoder.weights_gru",
                 "encoder.res_cells.0.update.weight", "weight_ts.0", "encoder.agru_cells.0.update.weights_g"):
        p = state[name]
        v = torch.from_numpy(rng.standard_normal(tuple(p.shape)).astype(np.float32)).to(dev)
        eps = 2e-3 * float(p.abs().max()) / max(float(v.abs().max()), 1e-30) * 3.0
        base = p.clone()
        p.copy_(base + eps * v)
        lp = loss()
        p.copy_(base - eps * v)
        lm = loss()
        p.copy_(base)
        fd = (lp - lm) / (2 * eps)
        an = float((grads[name].double() * v.double()).sum())
        assert abs(fd - an) <= 2e-2 * max(abs(an), abs(fd)) + 1e-3, (name, fd, an)
    hp.prepare()


@pytest.mark.parametrize("n,b,layers,flags", [
    (16, 70, 2, {}), (32, 5, 2, {}), (21, 3, 1, {}), (21, 3, 3, {}), (16, 2, 4, {}), (21, 1, 2, {}),
    # the shipped batch size and its double (MultiATGCN.json:12): 32-row node work items in the training forward, the steps of an
    # x-part chunk in one k_px16 tile
    (21, 16, 2, {}), (19, 32, 3, {}),
    (21, 2, 3, {"gcn_off": True}), (21, 2, 3, {"fnn_off": True}), (21, 2, 1, {"gcn_off": True, "fnn_off": True}),
    (19, 2, 3, {"cheb_order": 3}),
    # cheb_order = 1 (run_model_parameter.py:13): one weight entry over I + sum of the supports (MultiATGCN.py:94-108)
    (21, 2, 3, {"cheb_order": 1}), (16, 3, 1, {"cheb_order": 1, "adjtype": "od", "adpadj": "none"}),
    (21, 2, 2, {"cheb_order": 1, "adjtype": "identity", "adpadj": "none"}),
    (19, 5, 2, {"cheb_order": 1, "adjtype": "multi", "adpadj": "bidirection"}),
    (21, 2, 3, {"adjtype": "od", "adpadj": "bidirection"}), (21, 2, 1, {"adjtype": "dist", "adpadj": "none"}),
    (16, 3, 3, {"adjtype": "identity", "adpadj": "none", "cheb_order": 3}), (21, 2, 3, {"adjtype": "multi", "adpadj": "none"}),
    (21, 2, 3, {"adjtype": "cosine", "adpadj": "unidirection", "cheb_order": 3}),
    (21, 3, 2, {"end_dim": 2}),     # two flow channels: output_dim = 2 (MultiATGCN.py:320)
    (21, 2, 2, {"ext": 8}),         # add_day_in_week: time of day + 7 day-of-week channels (:313-318)
    # the widest input of the reference's channel sweep (run_model_parameter.py:11-12, [True, True, True, ..]): time of
    # day + 7 day-of-week channels + 5 dynamic variables = 14 input channels
    (21, 2, 2, {"ext": 13}),
    (16, 3, 1, {"ext": 0}),         # add_time_in_day off: the flow channel alone
    # node embeddings of other widths than the default 20 (run_model_parameter.py sweeps embed_dim): 8 and 24 take the
    # 3- and 8-step instantiations of the matrix-core prepare / pool-gradient kernels, 40 the kernels for wide embeddings
    (21, 2, 2, {"embed": 8}), (21, 3, 2, {"embed": 24}), (16, 2, 2, {"embed": 40}),
    # round 4: the values the reference's embed_dim_node sweep actually runs (run_model_parameter.py:14,
    # [1, 5, 10, 20, 30, 50]) - 1 / 5 / 10 take k_prep_mfma<3>, 30 the <8> instantiation next to its boundary (32), 50 the
    # kernels for wide embeddings
    (21, 2, 2, {"embed": 1}), (21, 2, 2, {"embed": 5}), (19, 3, 2, {"embed": 10}), (21, 2, 2, {"embed": 30}),
    (16, 2, 2, {"embed": 50}),
    # temporal-head layouts of the reference's first sweep that have no reference fixture (run_model_parameter.py:6-7):
    # closeness + trend without a period block, closeness + period without trend, two period heads
    (21, 2, 2, {"lens": (1, 0, 1)}), (21, 3, 2, {"lens": (1, 1, 0)}), (19, 2, 2, {"lens": (1, 2, 1)}),
])
def test_backward_on_synthetic_shapes_outside_the_golden_set(n, b, layers, flags, lib_built):
    """N a multiple of 16 (no padding rows anywhere), a batch that is not a multiple of the 64-row tile, and 1 / 3 / 4
    encoder layers (the golden cases all have 2): synthetic cases outside the golden set, forward and HIP gradients vs
    fp64 autograd through the oracle"""
    from multistgraph_amd import graph_prep, synthetic as syn
    from multistgraph_amd.ops import HotPath, diagonal_mask, spec_from_config
    from oracle import matgcn_oracle as orc
    dev = torch.device("cuda:0")
    cheb = flags.get("cheb_order", 2)
    adjtype, adpadj = flags.get("adjtype", "multi"), flags.get("adpadj", "unidirection")
    od, ext = flags.get("end_dim", 1), flags.get("ext", 1)
    emb = flags.get("embed", 20)
    lens = flags.get("lens", (2, 1, 1))
    abl = {k: v for k, v in flags.items() if k not in ("cheb_order", "adjtype", "adpadj", "end_dim", "ext", "embed", "lens")}
    cfg = dict(input_window=24, output_window=6, add_time_in_day=ext > 0, add_day_in_week=ext in (8, 13),
               load_dynamic=ext == 13,
               adjtype=adjtype, adpadj=adpadj, cheb_order=cheb, embed_dim_node=emb, embed_dim_adj=20, rnn_units=64,
               num_layers=layers, device=torch.device("cpu"), batch_size=b, start_dim=0, end_dim=od, **abl)
    df = dict(syn.make_data_feature(n, 3, "DC", ext_dim=ext, lens=lens), output_dim=od, feature_dim=od + ext)
    mats = graph_prep.build_static_supports(df["adj_mx"], df["coordinate"], None, adjtype)
    use_static = adpadj == "none" or adjtype == "multi"
    st = torch.from_numpy(np.stack(mats, 0))
    shapes = syn.param_shapes(n, out_steps=6, feat_in=od + ext, out_dim=od, k_total=syn.k_total_for(adjtype, adpadj, cheb),
                              layers=layers, embed_dim_node=emb, len_ts=sum(lens), **abl)
    state_np = syn.closed_form_state(shapes, 3)
    x_np, _ = syn.make_batch_arrays(b, n, 6, 3, feat=od + ext, x_steps=24 * sum(lens))
    if od > 1:   # channels [flow 0 .. flow od-1 | time of day]
        x_np = np.ascontiguousarray(np.concatenate([x_np[..., :1], x_np[..., 2:], x_np[..., 1:2]], -1))
    spec = spec_from_config(cfg, df, n, min(n, 20), st.shape[0] if use_static else 0,
                            diagonal_mask(st) if use_static else 0)
    hp = HotPath(spec, b, dev)
    state = {k: torch.from_numpy(v).to(dev) for k, v in state_np.items()}
    hp.bind(state, st.to(dev) if use_static else None)
    rng = np.random.default_rng(9)
    d_out = rng.standard_normal((b, 6, n, od)).astype(np.float32)
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in state_np.items()}
    ocfg = dict(adjtype=adjtype, adpadj=adpadj, cheb_order=cheb, num_layers=layers, rnn_units=64, len_closeness=24 * lens[0],
                len_period=24 * lens[1], len_trend=24 * lens[2], output_window=6, input_window=24, add_time_in_day=ext > 0,
                add_day_in_week=ext in (8, 13), load_dynamic=ext == 13, start_dim=0, end_dim=od, **abl)
    y = orc.forward(torch.tensor(x_np, dtype=torch.float64), p, [m.double() for m in st] if use_static else [], ocfg,
                    faithful=False)
    (y * torch.tensor(d_out, dtype=torch.float64)).sum().backward()
    x = torch.from_numpy(x_np).to(dev)
    got_y = hp.forward_train(x)
    assert max_norm_err(got_y.cpu().numpy(), y.detach().numpy()) <= 1e-4
    assert max_norm_err(hp.forward(x).cpu().numpy(), y.detach().numpy()) <= 1e-4
    grads = hp.backward(x, torch.from_numpy(d_out).to(dev), (hp.forward_train(x), state)[1])
    bad = {}
    for k, v in p.items():
        w = v.grad.numpy() if v.grad is not None else np.zeros(v.shape)
        if np.abs(w).max() == 0.0:
            if float(grads[k].abs().max()) > 1e-6:
                bad[k] = "expected zero"
        elif max_norm_err(grads[k].cpu().numpy(), w) > GRAD_TOL:
            bad[k] = max_norm_err(grads[k].cpu().numpy(), w)
    assert not bad, bad
