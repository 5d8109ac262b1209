"""The deep-supervision Dice criterion at the bench shape (bs 8, 800 x 1104, 5 classes; heads at 1/1, 1/2, 1/4, 1/8), kernel by kernel (HIP events): forward sums and
the gradient kernels of each head, and the whole fused forward (tcct_dice_ds_fwd).  Prints the results of the forward so that two builds can be compared.

    python tools/dice_bench.py      (gpurun: redirect into gpurun_out/; TCCT_LIB_PATH=ab/libtcct_REV.so for the other arm)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcct_amd._lib import lib
from tools.kbench import timeit


def main():
    torch.manual_seed(0)
    B, H, W, C = 8, 800, 1104, 5
    dev = 'cuda'
    logits = torch.randn(B, H, W, C, device=dev) * 2
    lab = torch.randint(0, C, (B, H, W), device=dev, dtype=torch.uint8)
    lows = [(torch.randn(B, H // s, W // s, C, device=dev) * 2).contiguous() for s in (2, 4, 8)]
    spin = torch.empty(B, H, W, 32, device=dev, dtype=torch.bfloat16)
    for _ in range(120):
        spin.copy_(spin)
    sums = torch.zeros(4 * 3 * C, device=dev, dtype=torch.float64)
    loss = torch.zeros((), device=dev)
    g = torch.ones((), device=dev)
    tot_f = tot_b = 0.0
    ms = timeit(lambda: lib.softmax_dice_fwd(logits, lab, B * H * W, C, sums[:3 * C], loss, 0), iters=20, warm=3)
    print(f'head 0 (full resolution) forward  {ms * 1e3:7.1f} us   loss {float(loss):.6f}', flush=True)
    tot_f += ms
    d0 = torch.empty_like(logits)
    ms = timeit(lambda: lib.softmax_dice_bwd(logits, lab, B * H * W, C, sums[:3 * C], g, 1.0, d0, 0), iters=20, warm=3)
    print(f'head 0 (full resolution) backward {ms * 1e3:7.1f} us   |d| {float(d0.abs().sum()):.6e}', flush=True)
    tot_b += ms
    for i, low in enumerate(lows):
        _, h, w, _ = low.shape
        sm = sums[(i + 1) * 3 * C:(i + 2) * 3 * C]
        ms = timeit(lambda: lib.updice_fwd(low, lab, B, h, w, H, W, C, sm, loss), iters=20, warm=3)
        print(f'head {i + 1} (scale {H // h}) forward           {ms * 1e3:7.1f} us   loss {float(loss):.6f}', flush=True)
        tot_f += ms
        ws = torch.empty(B, H, w, C, device=dev)
        d = torch.empty_like(low)
        ms = timeit(lambda: lib.updice_bwd(low, lab, B, h, w, H, W, C, sm, g, 0.5, ws, d), iters=20, warm=3)
        print(f'head {i + 1} (scale {H // h}) backward          {ms * 1e3:7.1f} us   |d| {float(d.abs().sum()):.6e}  d[0,1,1] {d[0, 1, 1].tolist()}', flush=True)
        tot_b += ms
    ms = timeit(lambda: lib.dice_ds_fwd(logits, 0, lab, B, H, W, C, lows[0], lows[0].shape[1], lows[0].shape[2], lows[1], lows[1].shape[1], lows[1].shape[2],
                                        lows[2], lows[2].shape[1], lows[2].shape[2], 0.5, sums, loss), iters=20, warm=3)
    print(f'tcct_dice_ds_fwd (four heads, one finalisation) {ms * 1e3:7.1f} us   loss {float(loss):.6f}')
    print(f'sum of the separate calls: forward {tot_f * 1e3:.1f} us (each with its memset + finalisation), backward {tot_b * 1e3:.1f} us')


if __name__ == '__main__':
    main()
