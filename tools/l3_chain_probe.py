"""Does a producer -> consumer chain run faster when it is cut into image chunks small enough for the intermediate to stay in the 256 MiB Infinity Cache?
A chain of DEPTH plain 3x3 32->32 convolutions at the level-0 bench shape (8 x 800 x 1104 x 32 bf16 = 452 MB per tensor), every link a separate launch:
  full   : conv_1 over the whole batch, then conv_2 over the whole batch, ...   (what the step does today: every intermediate is read back from HBM)
  chunk c: for each group of c images: conv_1, conv_2, ... on that group only   (intermediate of c x 56.5 MB between a store and its re-read)
python tools/l3_chain_probe.py [depth [streams]]        streams 2: even / odd chunks on two HIP streams (the tails of short launches overlap)"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tcct_amd._lib import lib, launch_on
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B, H, W = 8, 800, 1104
t = [torch.randn(B, H, W, 32, device='cuda').bfloat16() for _ in range(DEPTH + 1)]
w = torch.randn(32, 32, 3, 3, device='cuda') / 17
b = torch.randn(32, device='cuda')
wp = torch.empty(9 * 1024, device='cuda', dtype=torch.bfloat16)
lib.conv32_pack_weights(w, wp, 3, 3, 0)
streams = [torch.cuda.Stream() for _ in range(NS)]


def chain(c):
    for i, n0 in enumerate(range(0, B, c)):
        if NS > 1:
            with launch_on(streams[i % NS]):
                for d in range(DEPTH):
                    lib.conv32_fwd(t[d][n0:n0 + c], wp, b, t[d + 1][n0:n0 + c], c, H, W, 3, 3, 1, 1)
        else:
            for d in range(DEPTH):
                lib.conv32_fwd(t[d][n0:n0 + c], wp, b, t[d + 1][n0:n0 + c], c, H, W, 3, 3, 1, 1)


def timed(c, iters=10):
    for _ in range(3): chain(c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if NS > 1:
        for s in streams: s.wait_stream(torch.cuda.current_stream())
    e0.record()
    if NS > 1:
        for s in streams: s.wait_event(e0)
    for _ in range(iters): chain(c)
    if NS > 1:
        for s in streams: torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


lib.conv32_fwd_mode(2)          # the row-stream kernel whatever the block count
ref = None
for rep in range(3):
    line = []
    for c in (8, 4, 2, 1):
        ms = timed(c)
        line.append(f'chunk {c}: {ms:.4f} ms ({ms / DEPTH:.4f} per conv)')
    print(f'depth {DEPTH} streams {NS}:  ' + '   '.join(line), flush=True)
chain(8); full = t[DEPTH].clone(); chain(1)
print('chunked == full:', torch.equal(full, t[DEPTH]))
lib.conv32_fwd_mode(0)
