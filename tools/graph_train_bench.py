"""Training throughput on the reference's real crop size (256x256, data/octgen.py:8-19): eager launches vs hipGraph replay of the step.
usage: python tools/graph_train_bench.py [--height 256 --width 256 --bs 8]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tcct_amd.graph import GraphedTrainStep
if not any(x.startswith('--height') for x in sys.argv):
    sys.argv += ['--height=256', '--width=256']
a = bench.parse()
for graphed in (False, True):
    k, ds, args = bench.build_trainer(a, 1)
    k.model.train()
    img, lab, _, _ = ds.parse(ds.make_batch(a.bs, 2023))
    img, lab = img.contiguous(), lab.contiguous()
    step = GraphedTrainStep(k) if graphed else k.train_step
    for _ in range(6):
        step(img, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        loss = step(img, lab)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'bs={a.bs} {a.height}x{a.width} --los={a.los}: {"hipGraph replay" if graphed else "eager launches"}: {1e3 * dt:.2f} ms/step = {a.bs / dt:.0f} B-scans/s, loss {loss.item():.4f}')
