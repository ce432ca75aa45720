"""HIP hot path vs the reference's golden vectors and vs the CPU oracle, stage by stage.

Everything here calls through the C ABI (multistgraph_amd.ops.HotPath -> libmatgcn.so).
Tolerance: the north star's 1e-4 max-normalised fp32 for end-to-end outputs; single stages are held
to 2e-5 (they differ from the reference only by fp32 summation order).
"""
import numpy as np
import pytest
import torch

from helpers import FULL, TINY, Case, elementwise_excess, max_norm_err

pytestmark = pytest.mark.gpu

STAGE_TOL = 2e-5
E2E_TOL = 1e-4


def _path(c, lib_built, fold=True):
    """fold=True is the product configuration: static supports that are diagonal matrices (the similarity
    Laplacian without static features, the identity mode) are folded into the weights; fold=False mixes
    them as dense matrices like any other support."""
    from multistgraph_amd.ops import HotPath, diagonal_mask, spec_from_config
    dev = torch.device("cuda:0")
    use_static = c.adpadj == "none" or c.adjtype == "multi"
    st = torch.from_numpy(c.gold["static_supports"]).to(dev) if use_static else None
    spec = spec_from_config(c.config(), c.data_feature, c.n, min(c.n, 20), st.shape[0] if use_static else 0,
                            diagonal_mask(st) if fold else 0)
    hp = HotPath(spec, c.b, dev)
    hp.bind({k: torch.from_numpy(v).to(dev) for k, v in c.state.items()}, st)
    return hp, dev


def _unscaled_stack_all(c, which):
    """reference stack (K,N,N), identity included, with softmax(weights_g) divided out -> the raw supports"""
    stack = c.gold["stack_l0_%s" % which].astype(np.float64)
    if c.adjtype == "multi":
        g = c.state["encoder.agru_cells.0.%s.weights_g" % which].reshape(-1).astype(np.float64)
        g = np.exp(g - g.max())
        g /= g.sum()
        stack = stack / g[:, None, None]
    return stack


def _first_order(c, spec):
    """[(is_diagonal, [reference stack indices of its Chebyshev orders])] per first-order support, in stack order
    (MultiATGCN.py:94-100: [I, orders of S_1.., orders of S_2.., ..]; cheb_order = 1 still appends S itself)"""
    per = max(1, c.cheb - 1)
    adp = 0 if c.adpadj == "none" else 1
    out = []
    for f in range(spec.n_first):
        diag = f >= adp and bool((spec.diag_static_mask >> (f - adp)) & 1)
        out.append((diag, [1 + f * per + j for j in range(per)]))
    return out


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("name", TINY)
def test_support_stack(name, fold, lib_built):
    """the DENSE slots the device built (adaptive adjacency softmax(relu(.)), static Laplacians, Chebyshev orders;
    cheb_order = 1: their sum) against the reference's stack; folded diagonal supports are not in the stack - they are
    checked where they went, in test_prepared_weights_carry_gains_and_folded_diagonals"""
    c = Case(name)
    hp, _ = _path(c, lib_built, fold)
    got = hp.supports().cpu().numpy()
    ref = _unscaled_stack_all(c, "gate")
    dense = [idx for diag, idx in _first_order(c, hp.spec) if not diag]
    if c.cheb == 1:
        want = [sum(ref[i[0]] for i in dense)] if dense else []
    else:
        want = [ref[i] for idx in dense for i in idx]
    assert got.shape[0] == len(want)
    if want:
        assert max_norm_err(got, np.stack(want, 0)) <= STAGE_TOL


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("name", TINY)
def test_prepared_weights_carry_gains_and_folded_diagonals(name, fold, lib_built):
    """Read the node-adaptive weight streams back from `prepared` (what the node kernels really contract with) and
    compare with einsum('nd,dkio->nkio') of the reference (MultiATGCN.py:104) in fp64, with softmax(weights_g) folded in
    (:102-103) and every diagonal support folded into the identity slot scaled by its Chebyshev value t_j(s_n)."""
    c = Case(name)
    hp, _ = _path(c, lib_built, fold)
    spec = hp.spec
    adp = 0 if c.adpadj == "none" else 1
    firsts = _first_order(c, spec)
    emb = c.state["node_emb"].astype(np.float64)
    for layer in range(2):
        cin = c.feat if layer == 0 else 64
        for part, nm in enumerate(("gate", "update")):
            pre = "encoder.agru_cells.%d.%s." % (layer, nm)
            pool = c.state[pre + "weights_pool"].astype(np.float64)[:, :, cin:cin + 64, :]      # hidden-channel rows
            w = np.einsum("nd,dkio->nkio", emb, pool)
            kt = pool.shape[1]
            gain = np.ones(kt)
            if c.adjtype == "multi":
                wg = c.state[pre + "weights_g"].reshape(-1).astype(np.float64)
                gain = np.exp(wg - wg.max()) / np.exp(wg - wg.max()).sum()
            pk = (lambda k: 0) if c.cheb == 1 else (lambda k: k)     # cheb_order = 1: every entry reads pool index 0
            ident = gain[0] * w[:, 0]
            slots = []
            for f, (diag, idx) in enumerate(firsts):
                if diag:
                    sn = np.diagonal(c.gold["static_supports"][f - adp]).astype(np.float64)
                    t0, t1 = np.ones_like(sn), sn
                    for k in idx:
                        ident = ident + gain[pk(k)] * t1[:, None, None] * w[:, pk(k)]
                        t0, t1 = t1, 2 * sn * t1 - t0
                elif c.cheb > 1:
                    slots += [gain[k] * w[:, k] for k in idx]
            if c.cheb == 1 and any(not d for d, _ in firsts):
                slots = [gain[0] * w[:, 0]]
            want = np.stack([ident] + slots, 1)
            got = hp.node_weights(layer, part).cpu().numpy()
            assert got.shape == want.shape, (layer, nm)
            assert max_norm_err(got, want) <= STAGE_TOL, (layer, nm)


@pytest.mark.parametrize("name", TINY)
def test_fuse_heads(name, lib_built):
    c = Case(name)
    hp, dev = _path(c, lib_built)
    got = hp.fuse_heads(torch.from_numpy(c.x).to(dev)).cpu().numpy()
    assert max_norm_err(got, c.gold["x0"]) <= STAGE_TOL


@pytest.mark.parametrize("name", TINY)
def test_agcn_gate(name, lib_built):
    c = Case(name)
    hp, dev = _path(c, lib_built)
    x, h = (torch.from_numpy(c.gold[k]).to(dev) for k in ("stage_x", "stage_h"))
    got = hp.agcn_gate(0, x, h).cpu().numpy()
    assert max_norm_err(got, c.gold["agcn_gate_l0"]) <= STAGE_TOL


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("name", TINY)
def test_atgru_cells(name, fold, lib_built):
    c = Case(name)
    hp, dev = _path(c, lib_built, fold)
    x, h, x1 = (torch.from_numpy(c.gold[k]).to(dev) for k in ("stage_x", "stage_h", "stage_x1"))
    assert max_norm_err(hp.atgru_cell(0, x, h).cpu().numpy(), c.gold["cell_l0"]) <= STAGE_TOL
    assert max_norm_err(hp.atgru_cell(1, x1, h).cpu().numpy(), c.gold["cell_l1"]) <= STAGE_TOL
    assert max_norm_err(hp.res_cell(0, x, h).cpu().numpy(), c.gold["res_l0"]) <= STAGE_TOL


@pytest.mark.parametrize("name", TINY)
def test_encoder_and_head(name, lib_built):
    c = Case(name)
    hp, dev = _path(c, lib_built)
    assert hp.lib.matgcn_set_wavefront(1) in (0, 1)
    seq, fin = hp.encoder(torch.from_numpy(c.gold["x0"]).to(dev))
    assert max_norm_err(seq.cpu().numpy(), c.gold["enc_seq"]) <= E2E_TOL
    assert max_norm_err(fin.cpu().numpy(), c.gold["enc_finals"]) <= E2E_TOL
    out = hp.output_head(torch.from_numpy(c.gold["enc_seq"]).to(dev)).cpu().numpy()
    if c.static_dim == 0:
        assert max_norm_err(out, c.gold["pred"]) <= STAGE_TOL
    else:   # the fixture's encoder stage starts from the zero state, its prediction from the static embedding
        from oracle import matgcn_oracle as O
        want = O.output_head(torch.from_numpy(c.gold["enc_seq"]), O.to_tensors(c.state), c.out, 1).numpy()
        assert max_norm_err(out, want) <= STAGE_TOL


@pytest.mark.parametrize("fold", [True, False])
@pytest.mark.parametrize("name", TINY + FULL)
def test_forward(name, fold, lib_built):
    c = Case(name)
    hp, dev = _path(c, lib_built, fold)
    h0 = c.h0()      # static-feature cases start every layer and sample from the reference's static embedding (:406-409)
    got = hp.forward(torch.from_numpy(c.x).to(dev), None if h0 is None else h0.to(dev)).cpu().numpy()
    assert got.shape == c.gold["pred"].shape
    assert max_norm_err(got, c.gold["pred"]) <= E2E_TOL
    # element-wise: |got - ref| <= 1e-4 |ref| + 1e-6 max|ref| for EVERY prediction
    assert elementwise_excess(got, c.gold["pred"]) <= 1.0, elementwise_excess(got, c.gold["pred"])


def _fp64_gap_check(name, got, ref32, factor=2.0):
    """|got - ref64| against the reference's OWN |ref32 - ref64| on the elements fp64_gap.npz holds: the worst element and
    the r.m.s. within ``factor``.  Returns (gap_ref max, gap_hip max)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp64_gap.npz"))
    sub = int(z[name + "_sub"])
    p64 = z[name + "_pred64"]
    g = np.asarray(got, dtype=np.float64).reshape(-1)[::sub]
    r = np.asarray(ref32, dtype=np.float64).reshape(-1)[::sub]
    assert g.shape == p64.shape == r.shape
    e_ref, e_hip = np.abs(r - p64), np.abs(g - p64)
    rms_ref, rms_hip = float(np.sqrt((e_ref ** 2).mean())), float(np.sqrt((e_hip ** 2).mean()))
    print("%s: reference fp32-vs-fp64 gap max %.3e rms %.3e | HIP-vs-fp64 max %.3e rms %.3e (max|y| %.4f)" % (
        name, e_ref.max(), rms_ref, e_hip.max(), rms_hip, np.abs(p64).max()))
    assert e_hip.max() <= factor * e_ref.max(), (e_hip.max(), e_ref.max())
    assert rms_hip <= factor * rms_ref, (rms_hip, rms_ref)
    return float(e_ref.max()), float(e_hip.max())


def test_forward_is_as_close_to_float64_as_the_reference_at_403_nodes(lib_built):
    """SURVEY 8(c)'s "fp32-vs-fp64 gap 8e-7" as a fixture instead of a sentence: the reference's float64 prediction of
    bm403_out24 (B = 4), and the HIP path within twice the reference's own fp32 distance from it"""
    c = Case("bm403_out24")
    hp, dev = _path(c, lib_built)
    got = hp.forward(torch.from_numpy(c.x).to(dev)).cpu().numpy()
    gap_ref, _ = _fp64_gap_check("bm403_out24", got, c.gold["pred"])
    assert 5e-7 <= gap_ref / float(np.abs(c.gold["pred"]).max()) <= 1.2e-6      # the "8e-7" of the survey


BF16_TOL = 5e-3


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("name", ["tiny_multi_uni_c2", "tiny_od_non_c3", "tiny_multi_uni_c1", "tiny_multi_uni_c2_static",
                                  "tiny_heads_331", "tiny_multi_uni_dyn7", "dc237_out12", "bm403_out24"])
def test_bf16_mix_variant(name, mode, lib_built):
    """BASELINE config 3's dtype as an opt-in side line (matgcn_set_mix_precision(1)): bf16 OPERANDS for the graph
    mixes, fp32 accumulation, fp32 state and node-wise contractions.  Narrower than the reference's fp32, so it has
    its own tolerance: 5e-3 max-normalised against the reference's fp32 prediction (measured <= 3e-3); the fp32 path
    must come back bit-identical when the option is switched off again."""
    c = Case(name)
    hp, dev = _path(c, lib_built)
    x = torch.from_numpy(c.x).to(dev)
    h0 = c.h0()
    h0 = None if h0 is None else h0.to(dev)
    exact = hp.forward(x, h0).clone()
    prev = hp.lib.matgcn_set_mix_precision(mode)
    try:
        got = hp.forward(x, h0).clone()
        if mode == 2:
            hp.lib.matgcn_set_mix_precision(1)
            assert not torch.equal(hp.forward(x, h0), got)          # mode 2 is more than mode 1
    finally:
        hp.lib.matgcn_set_mix_precision(prev)
    assert not torch.equal(got, exact)                              # the variant really ran
    err = max_norm_err(got.cpu().numpy(), c.gold["pred"])
    assert err <= BF16_TOL, err
    assert max_norm_err(exact.cpu().numpy(), c.gold["pred"]) <= E2E_TOL
    assert torch.equal(hp.forward(x, h0), exact)                    # and the default is untouched afterwards


def test_encoder_nonzero_initial_state_vs_oracle(lib_built):
    # h0 != 0 is not in the golden set: check against the oracle directly
    from oracle import matgcn_oracle as O
    c = Case("tiny_multi_uni_c2")
    hp, dev = _path(c, lib_built)
    rng = np.random.default_rng(5)
    h0 = np.tanh(rng.standard_normal((2, c.b, c.n, 64))).astype(np.float32)
    seq, fin = hp.encoder(torch.from_numpy(c.gold["x0"]).to(dev), torch.from_numpy(h0).to(dev))
    p = O.to_tensors(c.state)
    st = O.supports_as_tensors(c.gold["static_supports"])
    want_seq, want_fin = O.encoder(torch.from_numpy(c.gold["x0"]), torch.from_numpy(h0), p, st, c.adjtype, c.adpadj,
                                   c.cheb, 2, faithful=False)
    assert max_norm_err(seq.cpu().numpy(), want_seq.numpy()) <= E2E_TOL
    assert max_norm_err(fin.cpu().numpy(), torch.stack(want_fin, 0).numpy()) <= E2E_TOL


@pytest.mark.parametrize("batch", [1, 5, 16, 32, 48, 64, 70])
def test_ragged_batches_vs_oracle(batch, lib_built):
    # batch sizes around the 64-row tile of the node-wise contraction, N not a multiple of 16; 16 and 32: the steps of an x-part
    # chunk share one tile (k_px16 merge), 32-row node work items
    from multistgraph_amd import synthetic as syn
    from oracle import matgcn_oracle as O
    c = Case("tiny_multi_uni_c2")
    c.b = batch
    c.x, c.y = syn.make_batch_arrays(batch, c.n, c.out, 77, feat=c.feat)
    hp, dev = _path(c, lib_built)
    got = hp.forward(torch.from_numpy(c.x).to(dev)).cpu().numpy()
    p = O.to_tensors(c.state)
    st = O.supports_as_tensors(c.gold["static_supports"])
    want = O.forward(torch.from_numpy(c.x), p, st, c.oracle_cfg(), faithful=False).numpy()
    assert max_norm_err(got, want) <= E2E_TOL


def test_config2_dc237_batch64_vs_oracle(lib_built):
    """BASELINE config 2 at its full batch: DC 237 nodes, in 24 h -> out 12 h, B = 64, fp32 - against the CPU oracle
    (hoisted order; the golden case dc237_out12 pins the same configuration to the reference at B = 4)"""
    from multistgraph_amd import synthetic as syn
    from oracle import matgcn_oracle as O
    c = Case("dc237_out12")
    c.b = 64
    c.x, c.y = syn.make_batch_arrays(64, c.n, c.out, 2024, feat=c.feat)
    hp, dev = _path(c, lib_built)
    got = hp.forward(torch.from_numpy(c.x).to(dev)).cpu().numpy()
    want = O.forward(torch.from_numpy(c.x), O.to_tensors(c.state), O.supports_as_tensors(c.gold["static_supports"]),
                     c.oracle_cfg(), faithful=False).numpy()
    assert got.shape == (64, 12, 237, 1)
    assert max_norm_err(got, want) <= E2E_TOL
    assert elementwise_excess(got, want) <= 1.0, elementwise_excess(got, want)


def test_linearity_of_graph_mix(lib_built):
    # size-independent property at the Baltimore shape: the gate pre-activation is affine in (x, h):
    # f(a) + f(b) - f(0) == f(a + b)
    c = Case("bm403_out24")
    hp, dev = _path(c, lib_built)
    g = torch.Generator(device="cpu").manual_seed(1)
    xa, xb = (torch.randn(c.b, c.n, 2, generator=g).to(dev) for _ in range(2))
    ha, hb = (torch.randn(c.b, c.n, 64, generator=g).to(dev) for _ in range(2))
    z = hp.agcn_gate(0, torch.zeros_like(xa), torch.zeros_like(ha))
    lhs = hp.agcn_gate(0, xa, ha) + hp.agcn_gate(0, xb, hb) - z
    rhs = hp.agcn_gate(0, xa + xb, ha + hb)
    assert max_norm_err(lhs.cpu().numpy(), rhs.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("name", ["dc237_out12", "tiny_multi_uni_c2_static", "tiny_identity_non_c1", "tiny_multi_uni_dyn7"])
def test_wavefront_and_serial_schedules_agree_bitwise(name, lib_built):
    # the layer wavefront only reorders independent launches: results must be identical to the serial schedule
    c = Case(name)
    hp, dev = _path(c, lib_built)
    x = torch.from_numpy(c.x).to(dev)
    h0 = c.h0()
    h0 = None if h0 is None else h0.to(dev)
    prev = hp.lib.matgcn_set_wavefront(1)
    try:
        a = hp.forward(x, h0).cpu().numpy()
        hp.lib.matgcn_set_wavefront(0)
        b = hp.forward(x, h0).cpu().numpy()
    finally:
        hp.lib.matgcn_set_wavefront(prev)
    assert np.array_equal(a, b)


def test_full_size_batch_properties(lib_built):
    """BASELINE-size properties that need no reference run (Baltimore 403 nodes, B = 64, wavefront schedule):
    the forward is run-to-run bit-identical, and batch items are independent - permuting the batch permutes the
    predictions bit for bit, and a sample's prediction does not depend on what else is in the batch."""
    from multistgraph_amd import synthetic as syn
    c = Case("bm403_out24")
    c.b = 64
    c.x, c.y = syn.make_batch_arrays(64, c.n, c.out, 123, feat=c.feat)
    hp, dev = _path(c, lib_built)
    x = torch.from_numpy(c.x).to(dev)
    a = hp.forward(x).clone()
    b = hp.forward(x)
    assert torch.equal(a, b)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(1)).to(dev)
    p = hp.forward(x[perm].contiguous())
    assert torch.equal(p, a[perm])
    x2 = x.clone()
    x2[1:] = torch.roll(x2[1:], 1, dims=0)          # sample 0 keeps its place, the rest of the batch changes
    assert torch.equal(hp.forward(x2)[0], a[0])
    assert torch.isfinite(a).all()


# ---- BASELINE config 5's graph: synthetic 4096 nodes (reference-generated fixture synth4096_out24) ----------------
def _big_path(c, batch, lib_built):
    """N = 4096: the static supports come from the plugin's own host graph prep (3 x N x N is no fixture); the test
    first pins that prep to the reference's subsample, then hands it to the HIP path"""
    from multistgraph_amd import graph_prep
    from multistgraph_amd.ops import HotPath, diagonal_mask, spec_from_config
    mats = np.stack(graph_prep.build_static_supports(c.data_feature["adj_mx"], c.data_feature["coordinate"], None,
                                                     c.adjtype), 0)
    sub = mats.reshape(mats.shape[0], -1)[:, ::4099]
    assert sub.shape == c.gold["static_sub"].shape
    assert np.abs(sub - c.gold["static_sub"]).max() <= 1e-6
    sums = np.stack([mats.astype(np.float64).sum((1, 2)), np.abs(mats.astype(np.float64)).sum((1, 2))], 1)
    assert np.abs(sums - c.gold["static_sums"]).max() <= 1e-5 * np.abs(c.gold["static_sums"]).max()
    dev = torch.device("cuda:0")
    st = torch.from_numpy(mats).to(dev)
    cfg = dict(c.config(), batch_size=batch)
    spec = spec_from_config(cfg, c.data_feature, c.n, 20, 3, diagonal_mask(torch.from_numpy(mats)))
    hp = HotPath(spec, batch, dev)
    state = {k: torch.from_numpy(v).to(dev) for k, v in c.state.items()}
    hp.bind(state, st)
    return hp, dev, state, mats


def test_forward_synth4096(lib_built):
    """BASELINE config 5 (4096-node graph, in 24 -> out 24) against the reference's own prediction, loss and MAE@k
    at B = 2: Np = 4096, St = [4096][12288], 192 row tiles per column tile, every 32-bit offset at its largest"""
    from multistgraph_amd.ops import masked_mae_device
    c = Case("synth4096_out24")
    assert c.checksums_ok()
    hp, dev, _, _ = _big_path(c, c.b, lib_built)
    got = hp.forward(torch.from_numpy(c.x).to(dev))
    assert got.shape == c.gold["pred"].shape
    assert max_norm_err(got.cpu().numpy(), c.gold["pred"]) <= E2E_TOL
    # Element-wise, against a MEASURED floor (round 4; round 3 scaled the floor by sqrt(N / 403), an argument the
    # measurement below refutes): the reference model was run once more in float64 on this very input
    # (tests/golden/make_fp64_golden.py -> fp64_gap.npz, every 7th element).  Its own fp32 result - the golden vector -
    # sits gap_ref = max|ref32 - ref64| away from it (9.1e-7 of max|y| at N = 4096, 8.6e-7 at N = 403: the gap does NOT
    # grow with N).  The HIP result must be as close to the float64 truth as twice that, at the worst element and in
    # the r.m.s.; two fp32 results that both sit within (2 gap_ref, gap_ref) of the truth differ by at most 3 gap_ref,
    # which is the absolute floor of the element-wise check against the golden vector.
    gap_ref, _ = _fp64_gap_check("synth4096_out24", got.cpu().numpy(), c.gold["pred"])
    floor = 3.0 * gap_ref / float(np.abs(c.gold["pred"]).max())
    assert floor < 3.2e-6
    excess = elementwise_excess(got.cpu().numpy(), c.gold["pred"], floor=floor)
    assert excess <= 1.0, excess
    res = masked_mae_device(got, torch.from_numpy(c.y).to(dev), 0, 0.0, 1.0, null_val=0.0).cpu().numpy()
    assert abs(res[0] - float(c.gold["loss"])) <= 1e-4 * abs(float(c.gold["loss"]))
    res = masked_mae_device(got, torch.from_numpy(c.y).to(dev), 0, 0.0, 1.0).cpu().numpy()
    assert np.abs(res[1:] - c.gold["mae_at"]).max() <= 1e-4 * np.abs(c.gold["mae_at"]).max()


def test_backward_synth4096_vs_fp64_oracle(lib_built):
    """one gradient check at N = 4096, B = 1: matgcn_forward_train + matgcn_backward against fp64 autograd through the
    oracle (hoisted order) on the GPU box's host cores"""
    from oracle import matgcn_oracle as orc
    c = Case("synth4096_out24")
    hp, dev, state, mats = _big_path(c, 1, lib_built)
    x_np = c.x[:1].copy()
    rng = np.random.default_rng(31)
    d_out = rng.standard_normal((1, c.out, c.n, 1)).astype(np.float32)
    p = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in c.state.items()}
    y = orc.forward(torch.tensor(x_np, dtype=torch.float64), p, [torch.from_numpy(m).double() for m in mats],
                    c.oracle_cfg(), faithful=False)
    (y * torch.tensor(d_out, dtype=torch.float64)).sum().backward()
    x = torch.from_numpy(x_np).to(dev)
    got_y = hp.forward_train(x)
    assert max_norm_err(got_y.cpu().numpy(), y.detach().numpy()) <= E2E_TOL
    assert max_norm_err(got_y.cpu().numpy(), c.gold["pred"][:1]) <= E2E_TOL     # batch items are independent
    grads = hp.backward(x, torch.from_numpy(d_out).to(dev), state)
    bad = {}
    for k, v in p.items():
        w = v.grad.numpy() if v.grad is not None else np.zeros(v.shape)
        if np.abs(w).max() == 0.0:
            if float(grads[k].abs().max()) > 1e-6:
                bad[k] = "expected zero"
        elif max_norm_err(grads[k].cpu().numpy(), w) > 1e-4:
            bad[k] = max_norm_err(grads[k].cpu().numpy(), w)
    assert not bad, bad


@pytest.mark.parametrize("name", ["tiny_multi_uni_c2", "tiny_multi_uni_c2_static", "tiny_heads_331", "tiny_od_non_c3",
                                  "tiny_multi_uni_dyn7", "dc237_out12", "bm403_out24"])
def test_batch_split_forward_equals_two_half_batch_forwards(name, lib_built):
    """matgcn_set_batch_split(2): the two halves of the batch as two independent forwards side by side (two streams,
    two wavefront sets, two halves of the workspace).  Samples never interact, so the result must be BITWISE that of
    forwards of B / 2 samples each, and within the end-to-end tolerance of the reference's prediction."""
    from multistgraph_amd.ops import HotPath
    c = Case(name)
    if c.b % 2:
        pytest.skip("odd batch: the split does not apply")
    hp, dev = _path(c, lib_built)
    x = torch.from_numpy(c.x).to(dev)
    h0 = c.h0()
    h0 = None if h0 is None else h0.to(dev)
    plain = hp.forward(x, h0).clone()
    prev = hp.lib.matgcn_set_batch_split(2)
    try:
        split = hp.forward(x, h0).clone()
        again = hp.forward(x, h0).clone()
    finally:
        hp.lib.matgcn_set_batch_split(prev)
    assert torch.equal(split, again)
    assert max_norm_err(split.cpu().numpy(), c.gold["pred"]) <= E2E_TOL
    assert elementwise_excess(split.cpu().numpy(), c.gold["pred"]) <= 1.0
    # the same as forwards of the half batch (32-row work items for B / 2 <= 32; same arithmetic per sample)
    hb = c.b // 2
    half = HotPath(hp.spec, hb, dev)
    half.bind({k: torch.from_numpy(v).to(dev) for k, v in c.state.items()}, hp._static)
    for i in range(2):
        hh = None if h0 is None else h0[:, i * hb:(i + 1) * hb].contiguous()
        want = half.forward(x[i * hb:(i + 1) * hb].contiguous(), hh)
        assert torch.equal(split[i * hb:(i + 1) * hb], want), i
    assert max_norm_err(split.cpu().numpy(), plain.cpu().numpy()) <= 1e-5     # and equal to the unsplit forward to rounding
