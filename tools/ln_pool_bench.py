"""LayerNorm + token mixer + residual: the one-pass kernels (csrc/ln_pool.hip) against the two-kernel path, forward and backward, at the stage shapes of the bench
configuration (bs 8, 400 x 552 tokens x 64 channels; 200 x 276 x 96):   python tools/ln_pool_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tcct_amd import ops


def timeit(f, iters=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (B, N, C) in ((8, 400 * 552, 64), (8, 200 * 276, 96), (8, 100 * 138, 128)):
    t = torch.randn(B, N, C, device='cuda').bfloat16().requires_grad_(True)
    g = (1 + 0.1 * torch.randn(C, device='cuda')).requires_grad_(True)
    b = (0.1 * torch.randn(C, device='cuda')).requires_grad_(True)
    gy = torch.randn(B, N, C, device='cuda').bfloat16()
    sc = torch.full((B,), 1.1, device='cuda')
    res = {}
    for fused in (True, False):
        def fwd():
            if fused:
                return ops.ln_metapool_residual(t, g, b, 1e-6, sc)
            cur, al = ops.layernorm_fork(t, g, b, 1e-6)
            return ops.metapool_residual(cur, al, sc)
        with torch.no_grad():
            tf = timeit(fwd)
        y = fwd()
        def bwd():
            t.grad = None
            y.backward(gy, retain_graph=True)
        tb = timeit(bwd)
        res[fused] = (tf, tb)
    mb = B * N * C * 2 / 1e6
    print(f'[{B} x {N} x {C}] ({mb:.0f} MB per tensor)  one pass: fwd {res[True][0]:.4f} ms, bwd {res[True][1]:.4f} ms   two kernels: fwd {res[False][0]:.4f} ms, bwd {res[False][1]:.4f} ms', flush=True)
