"""How much of a mid-size kernel with fused BatchNorm statistics is the tail of same-address fp64 atomics?  The pointwise forward with and without its statistics
epilogue, and the backward reduction kernel, at the pixel counts of levels 1-4 (bs 8, 800 x 1104), HIP events.

    python tools/stats_tail_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcct_amd._lib import lib
from tools.kbench import timeit


def main():
    dev = 'cuda'
    spin = torch.empty(8, 800, 1104, 32, device=dev, dtype=torch.bfloat16)
    for _ in range(120):
        spin.copy_(spin)
    for lv, (K, N) in ((1, (64, 32)), (2, (96, 32)), (3, (128, 32)), (4, (160, 32)), (3, (128, 128)), (2, (96, 96))):
        M = 8 * (800 >> lv) * (1104 >> lv)
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev) * 0.05
        y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        sums = torch.zeros(2 * N, device=dev, dtype=torch.float64)
        t_plain = timeit(lambda: lib.pw_fwd(x, w, None, y, M, K, N, 0, 1), iters=50, warm=5)
        t_stats = timeit(lambda: lib.pw_fwd_bnstats(x, w, None, y, M, K, N, sums, 0), iters=50, warm=5)
        mb = M * (K + N) * 2 / 1e6
        print(f'level {lv}: pointwise {K:3d} -> {N:3d}, M = {M:7d} ({mb:6.1f} MB): plain {t_plain * 1e3:6.1f} us, with statistics {t_stats * 1e3:6.1f} us (incl. a 0.5 KB memset launch)', flush=True)
        mr = torch.zeros(2 * N, device=dev); mr[N:] = 1
        ab = torch.ones(2 * N, device=dev)
        dy = torch.randn(M, N, device=dev).to(torch.bfloat16)
        t_red = timeit(lambda: lib.bn_bwd_reduce(y, dy, M, N, mr, ab, 0, 0, sums, 1), iters=50, warm=5)
        print(f'          BatchNorm backward reduction over {N} channels ({M * N * 4 / 1e6:6.1f} MB): {t_red * 1e3:6.1f} us', flush=True)


if __name__ == '__main__':
    main()
