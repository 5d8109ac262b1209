import os, sys, torch
sys.path.insert(0, '.')
from tcct_amd._lib import lib
for (B, H, W) in [(8, 800, 1104), (2, 400, 552), (1, 200, 64), (1, 60, 32)]:
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn((B, H, W, 32), device='cuda', generator=g).bfloat16()
    w = torch.randn((32, 32, 3, 3), device='cuda', generator=g) / 17
    b = torch.randn(32, device='cuda', generator=g)
    wp = torch.empty(9 * 1024, device='cuda', dtype=torch.bfloat16)
    lib.conv32_pack_weights(w, wp, 3, 3, 0)
    ys = []
    for mode in (1, 2):
        lib.conv32_fwd_mode(mode)
        y = torch.full((B, H, W, 32), 777.0, device='cuda', dtype=torch.bfloat16)
        lib.conv32_fwd(x, wp, b, y, B, H, W, 3, 3, 1, 1)
        torch.cuda.synchronize()
        ys.append(y)
    lib.conv32_fwd_mode(0)
    bad = (ys[0] != ys[1]).any(dim=3)
    print((B, H, W), 'bad pixels', int(bad.sum()), 'unwritten', int((ys[1] == 777.0).all(dim=3).sum()))
    if bad.any():
        idx = bad.nonzero()
        print(' first', idx[:5].tolist(), 'last', idx[-3:].tolist())
        rows = idx[:, 1].unique(); cols = idx[:, 2].unique()
        print(' rows', rows[:20].tolist(), len(rows), ' cols', cols[:40].tolist(), len(cols))
        n0, r0, c0 = idx[0].tolist()
        print(' vals', ys[0][n0, r0, c0, :4].tolist(), ys[1][n0, r0, c0, :4].tolist())
