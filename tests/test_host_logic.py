"""CPU-side checks: host graph prep, the plugin's parameter tree, config -> path description,
and that libmatgcn.so builds, loads and exports every symbol include/matgcn.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from helpers import FULL, TINY, Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", TINY[:6] + FULL[:2] + ["tiny_multi_uni_c2_static", "tiny_cosine_non_c3_static"])
def test_graph_prep_matches_reference(name):
    from multistgraph_amd import graph_prep
    c = Case(name)
    mats = graph_prep.build_static_supports(c.data_feature["adj_mx"], c.data_feature["coordinate"], c.static, c.adjtype)
    assert np.abs(np.stack(mats, 0) - c.gold["static_supports"]).max() <= 1e-6


def test_graph_prep_unsorted_geo_ids():
    # the reference pivots on geo_id, i.e. distances follow sorted ids whatever the row order
    from multistgraph_amd import graph_prep, synthetic as syn
    co = syn.make_coordinates(9, 3)
    perm = np.random.default_rng(0).permutation(9)
    shuffled = co.iloc[perm].reset_index(drop=True)
    a = graph_prep.haversine_km(graph_prep.lonlat_table(co))
    b = graph_prep.haversine_km(graph_prep.lonlat_table(shuffled))
    assert np.allclose(a, b)


@pytest.mark.parametrize("name", ["tiny_multi_uni_c2", "tiny_od_non_c3", "tiny_multi_non_c3", "tiny_multi_uni_dyn7",
                                  "dc237_out12", "abl_gcnoff", "abl_fnnoff", "abl_nodeoff", "abl_gcnfnnoff",
                                  "tiny_multi_uni_c1", "tiny_multi_uni_c2_static", "tiny_cosine_non_c3_static"])
def test_parameter_tree_is_the_checkpoint_abi(name):
    from multistgraph_amd.model import MultiATGCN
    c = Case(name)
    m = MultiATGCN(c.config(), c.data_feature)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got) == list(c.shapes)          # same names, same registration order
    assert got == {k: tuple(v) for k, v in c.shapes.items()}
    m.load_state_dict({k: torch.from_numpy(v) for k, v in c.state.items()})
    # the reference init law: xavier for >= 2-D, U(0,1) for 1-D
    m2 = MultiATGCN(c.config(), c.data_feature)
    assert 0.0 <= float(m2.weight_tsg.min()) and float(m2.weight_tsg.max()) <= 1.0
    if c.flags.get("node_specific_off"):
        assert float(m2.node_emb.min()) == 1.0 == float(m2.node_emb.max()) and not m2.node_emb.requires_grad
    else:
        bound = (6.0 / (c.n + 20)) ** 0.5
        assert float(m2.node_emb.abs().max()) <= bound + 1e-6


def test_spec_from_config_heads_and_channels():
    from multistgraph_amd.ops import spec_from_config
    c = Case("tiny_multi_uni_c2")
    s3 = spec_from_config(c.config(), c.data_feature, c.n, 20, 3)
    assert tuple(s3.head_begin) == (0, 24) and s3.n_ts == 4 and s3.k_total == 5   # out=3 < 6: closeness only
    c12 = Case("tiny_multi_uni_out12")
    s12 = spec_from_config(c12.config(), c12.data_feature, c12.n, 20, 3)
    assert tuple(s12.head_begin) == (0, 24, 48, 72)
    c7 = Case("tiny_multi_uni_dyn7")
    s7 = spec_from_config(c7.config(), c7.data_feature, c7.n, 19, 3)
    assert s7.feat_in == 7 and tuple(s7.ext_src) == (1, 2, 3, 4, 5, 6) and s7.x_feat == 7
    cfg = dict(c.config(), adjtype="od", adpadj="none", cheb_order=3)
    assert spec_from_config(cfg, c.data_feature, c.n, 20, 1).k_total == 3


class ConfigParserLike:
    """The surface of LibCity's ConfigParser (libcity/config/config_parser.py:134-151): get / [] / []= / in / iter over
    keys - and NO keys(), so dict(config) or {**config} raise."""

    def __init__(self, d):
        self.config = dict(d)

    def get(self, key, default=None):
        return self.config.get(key, default)

    def __getitem__(self, key):
        if key in self.config:
            return self.config[key]
        raise KeyError("{} is not in the config".format(key))

    def __setitem__(self, key, value):
        self.config[key] = value

    def __contains__(self, key):
        return key in self.config

    def __iter__(self):
        return self.config.__iter__()


@pytest.mark.parametrize("name", ["tiny_multi_uni_c2", "hid32_multi_uni_c2", "tiny_multi_uni_c2_static", "abl_nodeoff"])
def test_model_builds_from_a_config_parser_like_object(name):
    """the pipeline hands the constructor a ConfigParser, not a dict (pipeline.py:30,52): the constructor must only use
    get / [] / []= (and it writes num_nodes back, MultiATGCN.py:233)"""
    from multistgraph_amd.model import MultiATGCN
    c = Case(name)
    cfg = ConfigParserLike(c.config())
    with pytest.raises((TypeError, ValueError)):
        dict(cfg, rnn_units=64)       # what a copy of the config would do
    m = MultiATGCN(cfg, c.data_feature)
    assert cfg["num_nodes"] == c.n
    ref = MultiATGCN(c.config(), c.data_feature)
    assert m.spec == ref.spec and m.spec.hidden == 64
    assert [k for k, _ in m.named_parameters()] == [k for k, _ in ref.named_parameters()]


def test_unsupported_options_fail_loudly():
    from multistgraph_amd import synthetic as syn
    from multistgraph_amd.model import MultiATGCN
    c = Case("tiny_multi_uni_c2")
    m = MultiATGCN(c.config(), c.data_feature).eval()
    with torch.no_grad(), pytest.raises(RuntimeError):
        m.predict({"X": torch.from_numpy(c.x)})      # CPU tensor: no fallback


def test_library_exports_every_declared_symbol(lib_built):
    from multistgraph_amd import _lib
    header = open(os.path.join(ROOT, "include", "matgcn.h")).read()
    declared = set(re.findall(r"\b(matgcn_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    lib = _lib.load()
    for sym in declared:
        assert hasattr(lib, sym)
    assert lib.matgcn_abi_version() == _lib.ABI_VERSION == 11
    assert lib.matgcn_error_string(-3) == b"configuration not supported by this build"


def test_size_queries_and_argument_checks(lib_built):
    from multistgraph_amd import _lib
    from multistgraph_amd.ops import spec_from_config
    lib = _lib.load()
    c = Case("bm403_out24")
    spec = spec_from_config(c.config(), c.data_feature, c.n, 20, 3)
    d = spec.dims(64)
    nb = C.c_size_t()
    assert lib.matgcn_prepared_bytes(C.byref(d), C.byref(nb)) == 0
    assert 250e6 < nb.value < 400e6       # ~300 MB of node-adaptive weights at N=403
    assert lib.matgcn_workspace_bytes(C.byref(d), C.byref(nb)) == 0
    assert 1e9 < nb.value < 3e9
    lay = (C.c_int64 * 4)()
    assert lib.matgcn_supports_layout(C.byref(d), C.byref(lay)) == 0
    assert lay[2] == 416 and lay[3] == 4
    d.hidden = 32
    assert lib.matgcn_prepared_bytes(C.byref(d), C.byref(nb)) == -3     # unsupported
    d = spec.dims(0)
    assert lib.matgcn_prepared_bytes(C.byref(d), C.byref(nb)) == -2     # bad arg
    assert lib.matgcn_prepared_bytes(None, C.byref(nb)) == -1           # null


def test_diagonal_supports_are_detected_on_the_host():
    # no static features: the similarity Laplacian is exactly -I (MultiATGCN.py:244-250) -> bit 2 of the mask
    from multistgraph_amd.model import MultiATGCN
    from multistgraph_amd.ops import diagonal_mask
    c = Case("tiny_multi_uni_c2")
    m = MultiATGCN(c.config(), c.data_feature)
    assert m.spec.n_static == 3 and m.spec.diag_static_mask == 0b100
    assert diagonal_mask(torch.stack([torch.eye(4), torch.ones(4, 4), -2 * torch.eye(4)])) == 0b101
    assert diagonal_mask(None) == 0
    ci = Case("tiny_identity_non_c2")
    assert MultiATGCN(ci.config(), ci.data_feature).spec.diag_static_mask == 1


def test_training_entry_points_check_their_arguments(lib_built):
    # no GPU needed: size query and the argument checks that come before any launch
    from multistgraph_amd import _lib
    from multistgraph_amd.ops import spec_from_config
    lib = _lib.load()
    c = Case("bm403_out24")
    spec = spec_from_config(c.config(), c.data_feature, c.n, 20, 3)
    d = spec.dims(64)
    nb = C.c_size_t()
    assert lib.matgcn_train_bytes(C.byref(d), C.byref(nb)) == 0
    assert 8e9 < nb.value < 16e9          # saved activations + mixed rows + two scratch sets at B=64, N=403
    d1 = spec.dims(1)
    nb1 = C.c_size_t()
    assert lib.matgcn_train_bytes(C.byref(d1), C.byref(nb1)) == 0 and nb1.value < nb.value / 10
    assert lib.matgcn_train_bytes(C.byref(d), None) == -1
    p = _lib.Params()
    assert lib.matgcn_forward_train(C.byref(d), C.byref(p), None, None, None, None, None, None, None, 0, None, 0,
                                    None) == -1
    assert lib.matgcn_backward(C.byref(d), C.byref(p), None, None, None, None, None, None, C.byref(p), None, None, 0,
                               None, 0, None) == -1
    assert lib.matgcn_debug_gemm(None, None, None, None, 1.0, 0.0, None) == -1


REFERENCE_ROOT = "/root/reference"


def _reference_model_class():
    """The reference's own MultiATGCN class - build container only (the reference never travels)."""
    import sys
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "libcity")):
        pytest.skip("reference checkout not present (GPU box): same-seed check runs in the build container")
    sys.dont_write_bytecode = True      # the reference tree is read-only
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
    import libcity.model.traffic_flow_prediction.MultiATGCN  # noqa: F401
    return sys.modules["libcity.model.traffic_flow_prediction.MultiATGCN"].MultiATGCN


@pytest.mark.parametrize("name,extra", [
    ("tiny_multi_uni_c2", {}), ("tiny_od_non_c3", {}), ("tiny_multi_bid_c2", {}), ("tiny_multi_uni_dyn7", {}),
    ("abl_gcnoff", {}), ("abl_fnnoff", {}), ("abl_nodeoff", {}), ("abl_gcnfnnoff", {}),
    ("tiny_multi_uni_c2", {"cheb_order": 1}), ("tiny_od_non_c2", {"cheb_order": 1}),
    ("tiny_multi_uni_c2", {"static_dim": 24}), ("tiny_cosine_non_c2", {"static_dim": 30}),
    ("tiny_multi_uni_c2", {"rnn_units": 32}), ("tiny_od_non_c2", {"rnn_units": 16, "static_dim": 24}),
])
def test_same_seed_gives_the_reference_initial_weights(name, extra):
    """_init_parameters (MultiATGCN.py:356-361) AND the RNG stream in front of it (:296 randn,
    nn.Linear / Conv2d constructors, :291 pca_lowrank with static features) - the same torch.manual_seed must give
    bit-identical state_dicts in the reference class and in the plugin."""
    from multistgraph_amd import synthetic as syn
    from multistgraph_amd.model import MultiATGCN
    ref_cls = _reference_model_class()
    c = Case(name)
    cfg = dict(c.config())
    static_dim = extra.get("static_dim", 0)
    cfg.update({k: v for k, v in extra.items() if k != "static_dim"})
    for seed in (0, 10):
        dfs = [syn.make_data_feature(c.n, c.seed, ext_dim=c.feat - 1, static_dim=static_dim) for _ in range(2)]
        torch.manual_seed(seed)
        ref = ref_cls(dict(cfg), dfs[0])
        torch.manual_seed(seed)
        own = MultiATGCN(dict(cfg), dfs[1])
        rs, os_ = ref.state_dict(), own.state_dict()
        assert list(rs) == list(os_)
        for k in rs:
            assert rs[k].shape == os_[k].shape, k
            assert torch.equal(rs[k], os_[k]), "%s differs for seed %d" % (k, seed)


def test_header_is_plain_c_and_links_against_the_library(tmp_path, lib_built):
    """include/matgcn.h is the drop-in boundary: it must compile as plain C (C99, -pedantic) and a C program that only
    includes it must link against libmatgcn.so and run the GPU-free entry points (status codes, sizes, argument checks)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi.c"
    src.write_text("""
#include <stdio.h>
#include <string.h>
#include "matgcn.h"
int main(void) {
  matgcn_dims d;
  size_t bytes = 0;
  memset(&d, 0, sizeof(d));
  if (matgcn_abi_version() != MATGCN_ABI_VERSION) return 1;
  if (matgcn_prepared_bytes(&d, &bytes) == MATGCN_OK) return 2;            /* all-zero dims are refused */
  if (matgcn_prepared_bytes(&d, NULL) != MATGCN_ERR_NULL) return 3;
  if (!matgcn_error_string(MATGCN_ERR_BAD_ARG)) return 4;
  if (matgcn_set_batch_split(0) != 0 || matgcn_set_mix_precision(0) != 0) return 5;
  if (matgcn_set_stream_pool(0) != MATGCN_OK) return 6;                     /* the default mode, before any stream exists */
  printf("abi %d ok\\n", matgcn_abi_version());
  return 0;
}
""")
    exe = tmp_path / "abi"
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(lib_built)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, str(src), "-o", str(exe),
                    "-L", libdir, "-lmatgcn", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, env=env)
    assert "abi 11 ok" in out.stdout


def test_bench_flop_models_are_consistent():
    """the executed-FLOP models of bench.py (what `whole_forward.frac_mfma` and `backward_frac_mfma` divide by): the
    executed forward is below the SURVEY formula (dense supports only, x columns mixed once), and the backward lies
    between 1.3 x and 2 x the executed forward (two GEMMs per forward GEMM, minus the adjacency gradients the static
    supports do not have and the shared x-part mixes)"""
    import bench
    for n, b, out in ((403, 64, 24), (237, 64, 12), (4096, 32, 24)):
        units = b * 24 * n
        survey = bench.algorithmic_flops_per_unit(n, out=out) * units
        fwd = bench.executed_flops_per_unit(n, 3, out=out) * units
        bwd = bench.backward_executed_flops(n, b, 3, 5, out=out)
        assert 0.5 * survey < fwd < survey
        assert 1.3 * fwd < bwd < 2.0 * fwd, (n, bwd / fwd)
    m = bench.step_kernel_models(403, 416, 64, 3)
    assert abs(m["k_mix"]["flops"] - 2 * 3 * 403 * 403 * 64 * 64) < 1


def test_bench_finds_its_kernels_in_the_committed_pmc_summary():
    """bench.py replays the PMC figures of profiles/r*_pmc_kernels.json by kernel NAME: every name it looks for must be in
    the newest summary (round 4: a template argument added to k_mix silently turned the roofline kernel's traffic into null)"""
    import glob
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    found = sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_kernels.json")), reverse=True)
    assert found, "no PMC summary committed"
    names = list(json.load(open(found[0]))["kernels"])
    for kind, needle in bench.PMC_KERNEL_NAMES.items():
        hits = [n for n in names if needle in n]
        assert hits, (kind, needle, names)
        if kind == "k_mix":
            assert all("bf16" not in n for n in hits), hits
