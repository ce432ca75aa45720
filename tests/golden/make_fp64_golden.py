#!/usr/bin/env python3
"""The reference's OWN fp32-vs-fp64 gap at the headline graphs, as a fixture (build container only).

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fp64_golden.py

For bm403_out24 (B = 4) and synth4096_out24 (B = 2) the reference model is run once more with every parameter, buffer and
input in float64 (`model.double()` + its plain-attribute supports) on the inputs and parameters of the fp32 fixture; what
is stored is the float64 prediction (N = 4096: every FP64_SUB-th element) next to nothing else - the fp32 prediction of the
reference is already in <case>.npz.  tests compare |HIP - fp64| with |reference fp32 - fp64| on the same elements, so the
tolerance of the N = 4096 case rests on a measurement instead of a scaling argument.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (imports the reference)
from multistgraph_amd import synthetic as syn  # noqa: E402

FP64_SUB = 7   # stride of the stored subsample at N = 4096


def run64(case):
    model, df, cfg, state = mg.build_reference(case)
    model = model.double()
    # the static supports are plain attributes [name, tensor] (MultiATGCN.py:264-283), not buffers: cast them too
    model.supports = [[s[0], s[1].double()] for s in model.supports]
    x, _ = syn.make_batch_arrays(case["batch"], case["nodes"], case["out"], case["seed"], feat=case["feat"])
    with torch.no_grad():
        pred = model.predict({"X": torch.from_numpy(x).double()})
    assert pred.dtype == torch.float64
    return pred.numpy()


def main():
    out = {}
    for name in ("bm403_out24", "synth4096_out24"):
        case = next(c for c in mg.CASES if c["name"] == name)
        t0 = time.time()
        p64 = run64(case)
        p32 = np.load(os.path.join(HERE, name + ".npz"))["pred"]
        gap = np.abs(p32.astype(np.float64) - p64)
        print("%s: fp64 forward %.0f s; reference fp32-vs-fp64 gap max %.3e (%.3e of max|y|), rms %.3e" % (
            name, time.time() - t0, gap.max(), gap.max() / np.abs(p64).max(), np.sqrt((gap ** 2).mean())), flush=True)
        if case.get("big", False):
            out[name + "_sub"] = np.int64(FP64_SUB)
            out[name + "_pred64"] = p64.reshape(-1)[::FP64_SUB].copy()
        else:
            out[name + "_sub"] = np.int64(1)
            out[name + "_pred64"] = p64.reshape(-1).copy()
    np.savez_compressed(os.path.join(HERE, "fp64_gap.npz"), **out)


if __name__ == "__main__":
    main()
