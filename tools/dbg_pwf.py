import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcct_amd._lib import lib
torch.manual_seed(0)
bad = 0
for (K, N) in [(64, 64), (96, 96), (128, 128), (160, 160), (128, 96), (192, 128), (256, 160), (320, 160), (96, 32), (128, 32), (160, 32), (32, 32), (32, 64)]:
    for M in [8, 32, 100, 128, 512, 2048, 5000]:
        x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
        dy = torch.randn(M, N, device='cuda')
        y = torch.full((M, N), float('nan'), device='cuda'); dx = torch.full((M, K), float('nan'), device='cuda')
        lib.pwf_fwd(x, w, b, y, M, K, N, 0)
        lib.pwf_fwd(dy, w, None, dx, M, N, K, 1)
        yr = (x.double() @ w.double().t() + b.double()); dxr = dy.double() @ w.double()
        e1 = (y.double() - yr).abs().max().item() / yr.abs().max().item(); e2 = (dx.double() - dxr).abs().max().item() / dxr.abs().max().item()
        e3 = e4 = 0.0
        if N <= 160:
            dw = torch.full((N, K), float('nan'), device='cuda'); db = torch.full((N,), float('nan'), device='cuda')
            lib.pwf_wgrad(x, dy, dw, db, M, K, N)
            dwr = dy.double().t() @ x.double(); dbr = dy.double().sum(0)
            e3 = (dw.double() - dwr).abs().max().item() / dwr.abs().max().item(); e4 = (db.double() - dbr).abs().max().item() / dbr.abs().max().item()
        flag = '' if max(e1, e2, e3, e4) < 1e-5 else '   <<<<<<'
        bad += bool(flag)
        print(f'K={K} N={N} M={M}: fwd {e1:.1e} dgrad {e2:.1e} wgrad {e3:.1e} dbias {e4:.1e}{flag}')
print('bad cases:', bad)
