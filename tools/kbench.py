"""Micro-benchmark of individual kernels at the level-0 bench shapes (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcct_amd import ops
from tcct_amd._lib import lib, BF16

B, H, W = 8, 800, 1104
dt = torch.bfloat16


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    which = sys.argv[1:] or ['conv']
    x = torch.randn(B, H, W, 32, device='cuda').to(dt)
    dy = torch.randn(B, H, W, 32, device='cuda').to(dt)
    gb = x.numel() * 2 / 1e9
    for _ in range(120):        # spin-up past the clock transient of the first ~10 ms of GPU activity (tools/dbg_iters.py)
        dy.copy_(x)
    dy = torch.randn(B, H, W, 32, device='cuda').to(dt)
    for (kh, kw) in [(3, 3), (1, 13), (13, 1)]:
        w = torch.randn(32, 32, kh, kw, device='cuda') * 0.05
        b = torch.zeros(32, device='cuda')
        y = torch.empty_like(x)
        wp = torch.empty(kh * kw * 1024, device='cuda', dtype=dt)
        lib.conv32_pack_weights(w, wp, kh, kw, 0)
        fl = 2 * kh * kw * 32 * 32 * B * H * W / 1e12
        if 'conv' in which:
            ms = timeit(lambda: lib.conv32_fwd(x, wp, b, y, B, H, W, kh, kw, (kh - 1) // 2, (kw - 1) // 2))
            print(f'conv32_fwd {kh}x{kw}: {ms:.3f} ms  {fl / ms * 1e3:.1f} TFLOP/s  {2 * gb / ms * 1e3:.0f} GB/s algorithmic')
            st = torch.zeros(64, device='cuda', dtype=torch.float64)
            ms = timeit(lambda: lib.conv32_fwd_bnstats(x, wp, b, y, B, H, W, kh, kw, (kh - 1) // 2, (kw - 1) // 2, st, 1))
            print(f'conv32_fwd_bnstats {kh}x{kw}: {ms:.3f} ms')
        if 'wgrad' in which:
            dw = torch.empty_like(w)
            db = torch.empty(32, device='cuda')
            ms = timeit(lambda: lib.conv2d_wgrad(x, dy, dw, db, B, H, W, 32, 32, 32, kh, kw, 1, (kh - 1) // 2, (kw - 1) // 2, 1, 1), iters=3, warm=1)
            print(f'conv2d_wgrad(valu) {kh}x{kw}: {ms:.3f} ms  {fl / ms * 1e3:.1f} TFLOP/s')
        if 'wgrad32' in which:
            dw = torch.empty_like(w)
            db = torch.empty(32, device='cuda')
            ms = timeit(lambda: lib.conv32_wgrad(x, dy, dw, db, B, H, W, kh, kw, (kh - 1) // 2, (kw - 1) // 2))
            print(f'conv32_wgrad {kh}x{kw}: {ms:.3f} ms  {fl / ms * 1e3:.1f} TFLOP/s  {2 * gb / ms * 1e3:.0f} GB/s algorithmic')
    if 'ew' in which:
        y = torch.empty_like(x)
        ms = timeit(lambda: lib.act_fwd(x, y, x.numel(), 3, 1))
        print(f'act_fwd gelu: {ms:.3f} ms {2 * gb / ms * 1e3:.0f} GB/s')
        ab = torch.ones(64, device='cuda')
        ms = timeit(lambda: lib.bn_apply(x, y, B * H * W, 32, ab, 1, 0, 1))
        print(f'bn_apply: {ms:.3f} ms {2 * gb / ms * 1e3:.0f} GB/s')
        sums = torch.empty(64, device='cuda', dtype=torch.float64)
        ms = timeit(lambda: lib.bn_stats(x, B * H * W, 32, 1, sums, 1))
        print(f'bn_stats: {ms:.3f} ms {gb / ms * 1e3:.0f} GB/s')
        mr = torch.zeros(64, device='cuda'); mr[32:] = 1
        ms = timeit(lambda: lib.bn_bwd_reduce(x, dy, B * H * W, 32, mr, ab, 1, 0, sums, 1))
        print(f'bn_bwd_reduce: {ms:.3f} ms {2 * gb / ms * 1e3:.0f} GB/s')
        dg = torch.empty(32, device='cuda'); db = torch.empty(32, device='cuda')
        ms = timeit(lambda: lib.bn_bwd_apply(x, dy, y, B * H * W, 32, mr, ab, ab, sums, 1, 0, dg, db, 1))
        print(f'bn_bwd_apply: {ms:.3f} ms {3 * gb / ms * 1e3:.0f} GB/s')


def misc_bench():
    import torch.nn.functional as F
    for (C, H, W, Ho, Wo, al, tdt) in [(32, 400, 552, 800, 1104, 1, dt), (5, 400, 552, 800, 1104, 0, torch.float32),
                                        (5, 200, 276, 800, 1104, 0, torch.float32), (5, 100, 138, 800, 1104, 0, torch.float32)]:
        code = 1 if tdt == torch.bfloat16 else 0
        x = torch.randn(8, H, W, C, device='cuda').to(tdt); y = torch.empty(8, Ho, Wo, C, device='cuda', dtype=tdt)
        dy = torch.randn(8, Ho, Wo, C, device='cuda').to(tdt); dx = torch.empty_like(x)
        gb = (x.numel() + y.numel()) * x.element_size() / 1e9
        m1 = timeit(lambda: lib.bilinear_fwd(x, y, 8, H, W, C, Ho, Wo, al, code))
        m2 = timeit(lambda: lib.bilinear_bwd(dy, dx, 8, H, W, C, Ho, Wo, al, code))
        print(f'bilinear C={C} {H}x{W}->{Ho}x{Wo}: fwd {m1:.3f} ms {gb / m1 * 1e3:.0f} GB/s | bwd {m2:.3f} ms {gb / m2 * 1e3:.0f} GB/s')
        if tdt == torch.float32:
            ws = torch.empty(8 * Ho * W * C, device='cuda')
            m3 = timeit(lambda: lib.bilinear_bwd_separable(dy, dx, ws, 8, H, W, C, Ho, Wo, al))
            print(f'   separable bwd {m3:.3f} ms')
    x = torch.randn(8, 220800, 64, device='cuda').to(dt); y = torch.empty_like(x)
    gb = 2 * x.numel() * 2 / 1e9
    m1 = timeit(lambda: lib.metapool_fwd(x, y, 8, 220800, 64, 1)); m2 = timeit(lambda: lib.metapool_bwd(x, y, 8, 220800, 64, 1))
    print(f'metapool 8x220800x64: fwd {m1:.3f} ms {gb / m1 * 1e3:.0f} GB/s | bwd {m2:.3f} ms {gb / m2 * 1e3:.0f} GB/s')
    x = torch.randn(8, 800, 1104, 32, device='cuda').to(dt); y = torch.empty(8, 400, 552, 32, device='cuda', dtype=dt); dx = torch.empty_like(x)
    gb = (x.numel() + y.numel()) * 2 / 1e9
    m1 = timeit(lambda: lib.maxpool2_fwd(x, y, 8, 800, 1104, 32, 1)); m2 = timeit(lambda: lib.maxpool2_bwd(x, y, dx, 8, 800, 1104, 32, 1))
    print(f'maxpool2 L0: fwd {m1:.3f} ms {gb / m1 * 1e3:.0f} GB/s | bwd {m2:.3f} ms')
    dl = torch.randn(8, 800, 1104, 5, device='cuda'); w = torch.randn(5, 32, 1, 1, device='cuda'); dx = torch.empty(8, 800, 1104, 32, device='cuda', dtype=dt)
    m1 = timeit(lambda: lib.conv2d_dgrad(dl, w, dx, 8, 800, 1104, 32, 5, 1, 1, 0, 0, 0, 1))
    print(f'aux dgrad 5->32 @L0: {m1:.3f} ms {(dl.numel() * 4 + dx.numel() * 2) / 1e9 / m1 * 1e3:.0f} GB/s')


def dw_bench():
    for (N, H, W, C, st) in [(8, 400, 552, 64, 1), (8, 400, 552, 96, 2), (8, 200, 276, 96, 1), (8, 800, 1104, 4, 1)]:
        x = torch.randn(N, H, W, C, device='cuda').to(dt)
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        y = torch.empty(N, Ho, Wo, C, device='cuda', dtype=dt)
        dy = torch.randn(N, Ho, Wo, C, device='cuda').to(dt)
        dx = torch.empty_like(x)
        w = torch.randn(C, 1, 3, 3, device='cuda')
        b = torch.zeros(C, device='cuda')
        dw = torch.empty_like(w); db = torch.empty_like(b)
        gx, gy = x.numel() * 2 / 1e9, y.numel() * 2 / 1e9
        m1 = timeit(lambda: lib.dwconv3x3_fwd(x, w, b, y, N, H, W, C, st, 0, 1))
        m2 = timeit(lambda: lib.dwconv3x3_dgrad(dy, w, dx, N, H, W, C, st, 0, 1))
        m3 = timeit(lambda: lib.dwconv3x3_wgrad(x, dy, dw, db, N, H, W, C, st, 1))
        print(f'dw {N}x{H}x{W}x{C} s{st}: fwd {m1:.3f} ms {(gx + gy) / m1 * 1e3:.0f} GB/s | dgrad {m2:.3f} ms {(gx + gy) / m2 * 1e3:.0f} GB/s | wgrad {m3:.3f} ms {(gx + gy) / m3 * 1e3:.0f} GB/s')


if __name__ == "__main__" and not ({"pw", "bwd", "ln", "c3", "bnpool", "junction"} & set(sys.argv[1:])):
    if 'dw' in sys.argv[1:]:
        dw_bench()
        sys.exit(0)
    if 'misc' in sys.argv[1:]:
        misc_bench()
        sys.exit(0)
    main()


def pw_bench():
    """pointwise GEMM shapes of the ViT branch at the bench batch"""
    shapes = [(8 * 400 * 552, 64, 64), (8 * 400 * 552, 128, 96), (8 * 400 * 552, 96, 32), (8 * 200 * 276, 96, 96), (8 * 200 * 276, 192, 128),
              (8 * 800 * 1104, 32, 32)]
    for (M, K, N) in shapes:
        x = torch.randn(M, K, device='cuda').to(dt)
        dy = torch.randn(M, N, device='cuda').to(dt)
        w = torch.randn(N, K, device='cuda') * 0.1
        b = torch.zeros(N, device='cuda')
        y = torch.empty(M, N, device='cuda', dtype=dt)
        dw = torch.empty_like(w)
        db = torch.empty(N, device='cuda')
        gb = M * (K + N) * 2 / 1e9
        ms = timeit(lambda: lib.pw_fwd(x, w, b, y, M, K, N, 0, 1))
        ms2 = timeit(lambda: lib.pw_wgrad(x, dy, dw, db, M, K, N))
        print(f'pw M={M} K={K} N={N}: fwd {ms:.3f} ms {gb / ms * 1e3:.0f} GB/s | wgrad {ms2:.3f} ms {gb / ms2 * 1e3:.0f} GB/s')


if 'pw' in sys.argv[1:]:
    pw_bench()


def bwd_bench():
    """fused conv32 3x3 backward vs the separate input-gradient + weight-gradient kernels; fused pointwise backward vs its parts"""
    x = torch.randn(B, H, W, 32, device='cuda').to(dt); dy = torch.randn(B, H, W, 32, device='cuda').to(dt)
    w = torch.randn(32, 32, 3, 3, device='cuda') * 0.05
    wp = torch.empty(9 * 1024, device='cuda', dtype=dt); lib.conv32_pack_weights(w, wp, 3, 3, 1)
    dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(32, device='cuda')
    t1 = timeit(lambda: lib.conv32_fwd(dy, wp, None, dx, B, H, W, 3, 3, 1, 1))
    t2 = timeit(lambda: lib.conv32_wgrad(x, dy, dw, db, B, H, W, 3, 3, 1, 1))
    t3 = timeit(lambda: lib.conv32_bwd3x3(x, dy, wp, None, dx, dw, db, B, H, W))
    gb = x.numel() * 2 / 1e9
    print(f'conv32 3x3 @L0: dgrad {t1:.3f} + wgrad {t2:.3f} = {t1 + t2:.3f} ms | fused {t3:.3f} ms ({3 * gb / t3 * 1e3:.0f} GB/s algorithmic)')
    for (M, K, N) in [(8 * 400 * 552, 64, 64), (8 * 400 * 552, 128, 96), (8 * 400 * 552, 96, 32), (8 * 800 * 1104, 32, 32), (8 * 200 * 276, 96, 96), (8 * 200 * 276, 128, 128), (8 * 100 * 138, 128, 128), (8 * 100 * 138, 96, 96)]:
        x = torch.randn(M, K, device='cuda').to(dt); dy = torch.randn(M, N, device='cuda').to(dt)
        w = torch.randn(N, K, device='cuda') * 0.1
        dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(N, device='cuda')
        t1 = timeit(lambda: lib.pw_fwd(dy, w, None, dx, M, N, K, 1, 1))
        t2 = timeit(lambda: lib.pw_wgrad(x, dy, dw, db, M, K, N))
        t3 = timeit(lambda: lib.pw_bwd(x, dy, w, None, dx, dw, db, M, K, N))
        print(f'pw M={M} K={K} N={N}: dgrad {t1:.3f} + wgrad {t2:.3f} = {t1 + t2:.3f} ms | fused {t3:.3f} ms ({M * (2 * K + N) * 2 / 1e9 / t3 * 1e3:.0f} GB/s algorithmic)')


if 'bwd' in sys.argv[1:]:
    bwd_bench()


def ln_bench():
    for (M, C) in [(8 * 400 * 552, 64), (8 * 200 * 276, 96), (8 * 100 * 138, 128)]:
        x = torch.randn(M, C, device='cuda').to(dt); dy = torch.randn(M, C, device='cuda').to(dt); y = torch.empty_like(x); dx = torch.empty_like(x)
        g = torch.ones(C, device='cuda'); b = torch.zeros(C, device='cuda'); mr = torch.empty(2 * M, device='cuda')
        dg = torch.zeros(C, device='cuda'); db = torch.zeros(C, device='cuda')
        t1 = timeit(lambda: lib.layernorm_fwd(x, y, M, C, g, b, 1e-6, mr, 1))
        t2 = timeit(lambda: lib.layernorm_bwd(x, dy, dx, M, C, g, mr, dg, db, 1))
        t3 = timeit(lambda: lib.layernorm_bwd_add(x, dy, y, dx, M, C, g, mr, dg, db, 1))
        gb = M * C * 2 / 1e9
        print(f'layernorm M={M} C={C}: fwd {t1:.3f} ms {2 * gb / t1 * 1e3:.0f} GB/s | bwd {t2:.3f} ms {3 * gb / t2 * 1e3:.0f} GB/s | bwd+res {t3:.3f} ms {4 * gb / t3 * 1e3:.0f} GB/s')


if 'ln' in sys.argv[1:]:
    ln_bench()


def c3_bench():
    """first layers: direct 3 -> 32 channel 3x3 convolution (tcct_c3_fwd / tcct_c3_wgrad) against im2col + pointwise GEMM"""
    for (B, H, W, st) in [(8, 800, 1104, 1), (8, 800, 1104, 2)]:
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        x4 = torch.randn(B, H, W, 4, device='cuda').to(torch.bfloat16)
        w = torch.randn(32, 3, 3, 3, device='cuda') * 0.2
        b = torch.randn(32, device='cuda')
        y = torch.empty(B, Ho, Wo, 32, device='cuda', dtype=torch.bfloat16)
        dy = torch.randn_like(y)
        sums = torch.zeros(64, device='cuda', dtype=torch.float64)
        dw, db = torch.zeros_like(w), torch.zeros_like(b)
        t1 = timeit(lambda: lib.c3_fwd(x4, w, b, y, B, H, W, st, sums, 0, None, 0, 0))
        t2 = timeit(lambda: lib.c3_fwd(x4, w, b, y, B, H, W, st, None, 0, None, 0, 0))
        t3 = timeit(lambda: lib.c3_wgrad(x4, dy, dw, db, B, H, W, st))
        pat = torch.empty(B, Ho, Wo, 32, device='cuda', dtype=torch.bfloat16)
        w2 = torch.randn(32, 32, device='cuda') * 0.2
        dw2 = torch.zeros_like(w2)
        M = B * Ho * Wo
        t4 = timeit(lambda: lib.im2col3x3_c3(x4, pat, B, H, W, st, BF16))
        t5 = timeit(lambda: lib.pw_fwd_bnstats(pat, w2, b, y, M, 32, 32, sums, 0))
        t6 = timeit(lambda: lib.pw_wgrad(pat, dy, dw2, db, M, 32, 32))
        gb = y.numel() * 2 / 1e9
        print(f'c3 {B}x{H}x{W} stride {st}: fwd+stats {t1:.3f} ms ({gb / t1 * 1e3:.0f} GB/s written) | fwd {t2:.3f} | wgrad {t3:.3f} ms || '
              f'im2col {t4:.3f} + gemm+stats {t5:.3f} | gemm wgrad {t6:.3f}')


if 'c3' in sys.argv[1:]:
    c3_bench()


def bnpool_bench():
    """last BatchNorm of an encoder level + MaxPool2d(2): fused kernels against bn_apply + maxpool | maxpool_bwd_add + bn_bwd_reduce + bn_bwd_apply"""
    for (H_, W_) in [(800, 1104), (400, 552)]:
        C, M = 32, B * H_ * W_
        x = torch.randn(B, H_, W_, C, device='cuda').to(dt)
        z, dsk, dx, dz = torch.empty_like(x), torch.randn_like(x), torch.empty_like(x), torch.empty_like(x)
        pooled = torch.empty(B, H_ // 2, W_ // 2, C, device='cuda', dtype=dt)
        dpool = torch.randn_like(pooled)
        amax = torch.empty(B, H_ // 2, W_ // 2, C // 4, device='cuda', dtype=torch.uint8)
        sums = torch.zeros(2 * C, device='cuda', dtype=torch.float64); sums[C:] = M
        g, bta = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
        mr, ab = torch.empty(2 * C, device='cuda'), torch.empty(2 * C, device='cuda')
        s2 = torch.zeros(2 * C, device='cuda', dtype=torch.float64)
        dg, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        t1 = timeit(lambda: lib.bn_pool_fwd_train(x, z, pooled, amax, B, H_, W_, C, sums, g, bta, 1e-5, 0.1, None, None, None, mr, ab, 1, 0, BF16))
        t2 = timeit(lambda: lib.bn_pool_bwd(x, dpool, dsk, amax, dx, B, H_, W_, C, mr, ab, 1, 0, s2, dg, db, BF16))
        u1 = timeit(lambda: lib.bn_apply_train(x, None, z, M, C, sums, g, bta, 1e-5, 0.1, None, None, None, mr, ab, 1, 0, BF16))
        u2 = timeit(lambda: lib.maxpool2_fwd(z, pooled, B, H_, W_, C, BF16))
        u3 = timeit(lambda: lib.maxpool2_bwd_add(z, dpool, dsk, dz, B, H_, W_, C, BF16))
        u4 = timeit(lambda: lib.bn_bwd_reduce(x, dz, M, C, mr, ab, 1, 0, s2, BF16))
        u5 = timeit(lambda: lib.bn_bwd_apply(x, dz, dx, M, C, mr, ab, g, s2, 1, 0, dg, db, BF16))
        print(f'bn+pool {B}x{H_}x{W_}x{C}: fused fwd {t1:.3f} bwd {t2:.3f} ms || separate fwd {u1:.3f}+{u2:.3f} bwd {u3:.3f}+{u4:.3f}+{u5:.3f} ms')


if 'bnpool' in sys.argv[1:]:
    bnpool_bench()


def junction_bench():
    """CrossCNN junction backward (two BatchNorms + GELU): the reduction and the apply pass at levels 0 / 1"""
    for (H_, W_) in [(800, 1104), (400, 552)]:
        C, M = 32, B * H_ * W_
        xa, xb, dyj = (torch.randn(B, H_, W_, C, device='cuda').to(dt) for _ in range(3))
        dxa, dxb = torch.empty_like(xa), torch.empty_like(xa)
        mr = torch.zeros(2 * C, device='cuda'); mr[C:] = 1
        ab = torch.ones(2 * C, device='cuda'); ab[C:] = 0
        s4 = torch.zeros(4 * C, device='cuda', dtype=torch.float64)
        dgs = [torch.empty(C, device='cuda') for _ in range(4)]
        t1 = timeit(lambda: lib.bn2_add_act_bwd_reduce(xa, xb, dyj, M, C, mr, ab, mr, ab, 1, 3, s4, BF16), iters=20)
        t2 = timeit(lambda: lib.bn2_add_act_bwd_apply(xa, xb, dyj, dxa, dxb, M, C, mr, ab, mr, ab, s4, 1, 3, dgs[0], dgs[1], dgs[2], dgs[3], BF16), iters=20)
        gb = xa.numel() * 2 / 1e9
        print(f'junction bwd {B}x{H_}x{W_}x{C}: reduce {t1:.3f} ms ({3 * gb / t1:.2f} TB/s)  apply {t2:.3f} ms ({5 * gb / t2:.2f} TB/s)')


if 'junction' in sys.argv[1:]:
    junction_bench()
