"""One-pass backward of the first layers (k_c3_bn_bwd<2, POST, PF>) at the bench shapes: the wave-private kernel (4), the block-tile kernel with 1 / 3 tiles requested ahead and its round-5 launch (0) (tcct_c3_bn_bwd_prefetch), HIP events.

    python tools/c3bwd_bench.py     (gpurun: redirect into gpurun_out/)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcct_amd._lib import lib
from tools.kbench import timeit


def main():
    torch.manual_seed(0)
    B, H, W = 8, 800, 1104
    dev = 'cuda'
    x4 = torch.zeros(B, H, W, 4, device=dev, dtype=torch.bfloat16)
    x4[..., :3] = torch.randn(B, H, W, 3, device=dev).to(torch.bfloat16)
    w = torch.randn(32, 3, 3, 3, device=dev) * 0.2
    spin = torch.empty(B, H, W, 32, device=dev, dtype=torch.bfloat16)
    for _ in range(120):
        spin.copy_(spin)
    for stride, post, name in ((1, 0, 'CNN cnn.0 -> cnn.1 (stride 1, no activation)'), (2, 2, 'ViT stem[0] (stride 2, Hardswish)')):
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        M = B * Ho * Wo
        bias = torch.randn(32, device=dev) * 0.1 if stride == 1 else None
        gamma, beta = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
        rm, rv, nbt = torch.zeros(32, device=dev), torch.ones(32, device=dev), torch.zeros((), device=dev, dtype=torch.int64)
        z = torch.empty(B, Ho, Wo, 32, device=dev, dtype=torch.bfloat16)
        sums = torch.zeros(64, device=dev, dtype=torch.float64)
        mean_rstd, ab = torch.empty(64, device=dev), torch.empty(64, device=dev)
        lib.c3_bn_fwd_train(x4, w, bias, z, B, H, W, stride, sums, gamma, beta, 1e-5, 0.1, rm, rv, nbt, mean_rstd, ab, post)
        dz = torch.randn(B, Ho, Wo, 32, device=dev).to(torch.bfloat16)
        outs = {}
        for pf in (1, 4, 0, 3, 4, 1, 4):
            lib.c3_bn_bwd_prefetch(pf)
            work = torch.zeros(4160, device=dev)
            s96 = torch.zeros(96, device=dev, dtype=torch.float64)
            dw, dbias, dg, db_ = torch.empty(32, 3, 3, 3, device=dev), (torch.empty(32, device=dev) if bias is not None else None), torch.empty(32, device=dev), torch.empty(32, device=dev)

            def run():
                work.zero_(); s96.zero_()
                lib.c3_bn_bwd_onepass(x4, w, bias, dz, B, H, W, stride, mean_rstd, ab, work, s96, dw, dbias, dg, db_, post)
            ms = timeit(run, iters=20, warm=3)
            mb = (M * 64 + B * H * W * 8) / 1e6
            print(f'{name}: form {pf}: {ms:.3f} ms (incl. two tiny memsets)  {mb / ms / 1e3:.2f} TB/s on dz + image ({mb:.0f} MB)', flush=True)
            torch.cuda.synchronize()
            if pf in outs:
                continue
            outs[pf] = (dw.clone(), dg.clone(), db_.clone())
        for pf in (3, 4):
            for a, b, nm in zip(outs[1], outs[pf], ('dw', 'dgamma', 'dbeta')):
                err = float((a - b).abs().max() / (a.abs().max() + 1e-12))
                print(f'    form {pf} vs 1: {nm} max rel diff {err:.2e}')
                assert err < 2e-3, (pf, nm, err)
    lib.c3_bn_bwd_prefetch(1)


if __name__ == '__main__':
    main()
