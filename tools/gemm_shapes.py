"""Times k_bgemm (matgcn_debug_gemm) on the shapes the backward launches at Baltimore size (B=64, N=403)."""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multistgraph_amd.ops import debug_gemm
dev = torch.device("cuda:0")
N, Np, B, S, T, Ks = 403, 416, 64, 4, 24, 3
buf = lambda n: torch.randn(int(n), device=dev)
shapes = {
    # name: (desc builder, sizes of A, B, C, flops)
    "chain node gate  dA=dPG.WpT  (403x4 of 64x64x128)": dict(
        M=B, N=64, K=128, K2=1, sA=(Np * 128, 1, 0), sB=(1, 128, 0), sC=(S * Np * 64, 1), nb=(N, S),
        bA=(128, 0), bB=(S * 192 * 128, 192 * 128), bC=(64, Np * 64), sizes=(B * Np * 128, N * S * 192 * 128, B * S * Np * 64)),
    "chain node update dA=dPU.WpT (403x4 of 64x64x64)": dict(
        M=B, N=64, K=64, K2=1, sA=(Np * 64, 1, 0), sB=(1, 64, 0), sC=(S * Np * 64, 1), nb=(N, S),
        bA=(64, 0), bB=(S * 192 * 64, 192 * 64), bC=(64, Np * 64), sizes=(B * Np * 64, N * S * 192 * 64, B * S * Np * 64)),
    "wgrad h rows gate (403x3 of 64x128x(24x64))": dict(
        M=64, N=128, K=B, K2=T, sA=(1, Ks * 64, N * B * Ks * 64), sB=(Np * 128, 1, B * Np * 128), sC=(128, 1), nb=(N, Ks),
        bA=(B * Ks * 64, 64), bB=(128, 0), bC=(S * 128 * 128, 128 * 128),
        sizes=(T * N * B * Ks * 64, T * B * Np * 128, N * S * 128 * 128)),
    "adaptive grad (403x403x(1536x64), split 48)": dict(
        M=N, N=N, K=64, K2=T * B, sA=(64, 1, S * Np * 64), sB=(1, 64, Np * 64), sC=(N, 1), nb=(1, 1),
        bA=(0, 0), bB=(0, 0), bC=(0, 0), sizes=(T * B * S * Np * 64, T * B * Np * 64, N * N), mode=1, split=48),
    "x node gate dAx=dPG.WpT (403x4 of 1536x64x128)": dict(
        M=T * B, N=64, K=128, K2=1, sA=(Np * 128, 1, 0), sB=(1, 128, 0), sC=(S * Np * 64, 1), nb=(N, S),
        bA=(128, 0), bB=(S * 192 * 128, 192 * 128), bC=(64, Np * 64),
        sizes=(T * B * Np * 128, N * S * 192 * 128, T * B * S * Np * 64)),
}
for name, d in shapes.items():
    a, b, c = (buf(n) for n in d["sizes"])
    desc = (d["M"], d["N"], d["K"], d["K2"], d["sA"][0], d["sA"][1], d["sA"][2], d["sB"][0], d["sB"][1], d["sB"][2],
            d["sC"][0], d["sC"][1], d["nb"][0], d["nb"][1], d["bA"][0], d["bA"][1], d["bB"][0], d["bB"][1],
            d["bC"][0], d["bC"][1], d.get("mode", 0), d.get("split", 1))
    for _ in range(3):
        debug_gemm(a, b, c, desc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        debug_gemm(a, b, c, desc)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    flops = 2.0 * d["M"] * d["N"] * d["K"] * d["K2"] * d["nb"][0] * d["nb"][1]
    print("%-58s %8.1f us  %6.1f TF/s" % (name, us, flops / us / 1e6))
