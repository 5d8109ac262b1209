"""conv3x3 -> conv3x3 as ONE launch (tcct_conv32_chain33, csrc/conv_chain.hip) against the two launches it replaces: bit-identity on a few shapes, then timing
at the level-0 / level-1 bench shapes, both arms interleaved on one box.   python tools/chain_bench.py [check|time|all]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tcct_amd._lib import lib
what = sys.argv[1] if len(sys.argv) > 1 else 'all'
dev = 'cuda'


def packs(seed):
    g = torch.Generator().manual_seed(seed)
    w = (torch.randn(32, 32, 3, 3, generator=g) / 17).to(dev)
    b = torch.randn(32, generator=g).to(dev)
    wp = torch.empty(9 * 1024, device=dev, dtype=torch.bfloat16)
    lib.conv32_pack_weights(w, wp, 3, 3, 0)
    return wp, b


def two(x, p1, p2, mode, res=None):
    N, H, W, _ = x.shape
    mid, y = torch.empty_like(x), torch.empty_like(x)
    sums = torch.zeros(64, device=dev, dtype=torch.float64)
    lib.conv32_fwd(x, p1[0], p1[1], mid, N, H, W, 3, 3, 1, 1)
    if mode == 'stats':
        lib.conv32_fwd_bnstats(mid, p2[0], p2[1], y, N, H, W, 3, 3, 1, 1, sums, 1)
    elif mode == 'res':
        lib.conv32_fwd_add(mid, p2[0], p2[1], res, y, N, H, W, 3, 3, 1, 1)
    else:
        lib.conv32_fwd(mid, p2[0], p2[1], y, N, H, W, 3, 3, 1, 1)
    return mid, y, sums


def one(x, p1, p2, mode, res=None):
    N, H, W, _ = x.shape
    mid, y = torch.empty_like(x), torch.empty_like(x)
    sums = torch.zeros(64, device=dev, dtype=torch.float64)
    lib.conv32_chain33(x, p1[0], p1[1], mid, p2[0], p2[1], y, res if mode == 'res' else None, N, H, W, sums if mode == 'stats' else None)
    return mid, y, sums


if what in ('check', 'all'):
    p1, p2 = packs(1), packs(2)
    for shape in [(2, 40, 70), (1, 5, 30), (3, 33, 61), (1, 64, 96), (2, 100, 138), (8, 50, 69), (8, 400, 552), (8, 800, 1104)]:
        N, H, W = shape
        g = torch.Generator().manual_seed(H * W)
        x = torch.randn(N, H, W, 32, generator=g).to(dev).bfloat16()
        res = torch.randn(N, H, W, 32, generator=g).to(dev).bfloat16()
        for mode in ('plain', 'stats', 'res'):
            ma, ya, sa = two(x, p1, p2, mode, res)
            mb, yb, sb = one(x, p1, p2, mode, res)
            torch.cuda.synchronize()
            ok_m, ok_y = torch.equal(ma, mb), torch.equal(ya, yb)
            ds = ((sa - sb).abs() / sa.abs().clamp_min(1.0)).max().item()
            print(f'{shape} {mode:5s}: mid equal {ok_m}  y equal {ok_y}  stats rel diff {ds:.1e}', flush=True)
            if not (ok_m and ok_y and ds < 2e-6):
                bad = (ya != yb).any(-1).nonzero()
                badm = (ma != mb).any(-1).nonzero()
                print('   first mismatching y pixels', bad[:6].tolist(), 'of', bad.shape[0], ' mid', badm[:6].tolist(), 'of', badm.shape[0])
                sys.exit(1)

if what in ('time', 'all'):
    p1, p2 = packs(1), packs(2)
    for SC in (1, 2, 4, 8, 16):
        N, H, W = 8, 800 // SC, 1104 // SC
        x = torch.randn(N, H, W, 32, device=dev).bfloat16()
        res = torch.randn(N, H, W, 32, device=dev).bfloat16()
        mid, y = torch.empty_like(x), torch.empty_like(x)
        sums = torch.zeros(64, device=dev, dtype=torch.float64)
        for _ in range(60): y.copy_(x)

        def t(fn, iters=30):
            for _ in range(5): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(iters): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters
        arms = {
            'two plain': lambda: (lib.conv32_fwd(x, p1[0], p1[1], mid, N, H, W, 3, 3, 1, 1), lib.conv32_fwd(mid, p2[0], p2[1], y, N, H, W, 3, 3, 1, 1)),
            'chain plain': lambda: lib.conv32_chain33(x, p1[0], p1[1], mid, p2[0], p2[1], y, None, N, H, W, None),
            'two stats': lambda: (lib.conv32_fwd(x, p1[0], p1[1], mid, N, H, W, 3, 3, 1, 1), lib.conv32_fwd_bnstats(mid, p2[0], p2[1], y, N, H, W, 3, 3, 1, 1, sums, 1)),
            'chain stats': lambda: lib.conv32_chain33(x, p1[0], p1[1], mid, p2[0], p2[1], y, None, N, H, W, sums),
            'two res': lambda: (lib.conv32_fwd(x, p1[0], p1[1], mid, N, H, W, 3, 3, 1, 1), lib.conv32_fwd_add(mid, p2[0], p2[1], res, y, N, H, W, 3, 3, 1, 1)),
            'chain res': lambda: lib.conv32_chain33(x, p1[0], p1[1], mid, p2[0], p2[1], y, res, N, H, W, None),
        }
        for rep in range(3):
            print(f'{N}x{H}x{W}: ' + '   '.join(f'{k} {t(f):.4f}' for k, f in arms.items()), flush=True)
