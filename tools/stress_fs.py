"""stress of the row-stream convolution's counted waits: full-size bit-identity against the tiled kernel, repeated, with a second stream hammering HBM"""
import sys, torch
sys.path.insert(0, '.')
from tcct_amd._lib import lib
torch.manual_seed(0)
side = torch.cuda.Stream()
big = torch.randn(64 * 1024 * 1024, device='cuda'); big2 = torch.empty_like(big)
bad_total = 0
for (B, H, W, stat) in [(8, 800, 1104, None), (8, 800, 1100, 1), (8, 400, 550, None), (3, 800, 1072, 0), (8, 800, 1104, 1)]:
    x = torch.randn((B, H, W, 32), device='cuda').bfloat16()
    w = torch.randn((32, 32, 3, 3), device='cuda') / 17
    b = torch.randn(32, device='cuda')
    wp = torch.empty(9 * 1024, device='cuda', dtype=torch.bfloat16)
    lib.conv32_pack_weights(w, wp, 3, 3, 0)
    sums = torch.zeros(64, device='cuda', dtype=torch.float64)
    def run(y):
        if stat is None: lib.conv32_fwd(x, wp, b, y, B, H, W, 3, 3, 1, 1)
        else: lib.conv32_fwd_bnstats(x, wp, b, y, B, H, W, 3, 3, 1, 1, sums, stat)
    lib.conv32_fwd_mode(1)
    ref = torch.empty_like(x); run(ref); torch.cuda.synchronize()
    lib.conv32_fwd_mode(2)
    nbad = 0
    for it in range(25):
        y = torch.full_like(x, 777.0)
        if it % 2:
            with torch.cuda.stream(side):
                for _ in range(6): big2.copy_(big)
        run(y)
        torch.cuda.synchronize()
        nbad += int((y != ref).any(dim=3).sum())
    print((B, H, W, stat), 'mismatching pixels over 25 runs:', nbad, flush=True)
    bad_total += nbad
lib.conv32_fwd_mode(0)
print('TOTAL', bad_total)
sys.exit(1 if bad_total else 0)
