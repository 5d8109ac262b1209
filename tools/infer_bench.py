"""Inference throughput of the eval path: KiteSeg.predict = eval-mode forward + argmax mask, bs x 1x800x1100 bf16.
usage: python tools/infer_bench.py [--bs 8] [--steps 20] [--unfused]   (unfused: BatchNorm/activation passes run as separate kernels)"""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tcct_amd import ops

p = argparse.ArgumentParser()
p.add_argument('--bs', type=int, default=8)
p.add_argument('--steps', type=int, default=20)
p.add_argument('--unfused', action='store_true')
p.add_argument('--dtype', default='bf16')
a = p.parse_args()
sys.argv = ['bench.py', f'--bs={a.bs}', f'--dtype={a.dtype}']
k, ds, args = bench.build_trainer(bench.parse(), 1)
k.model.eval()
if a.bs <= 2:            # the launch-bound regime is measured with hipGraph replay below: captures and the nested stage fork exclude each other in one process
    ops.graphs_exclude_stage_fork('tools/infer_bench.py --bs<=2')
img, lab, _, _ = ds.parse(ds.make_batch(a.bs, 2023))
res = {}
for fuse in ([False] if a.unfused else [True, False]):
    ops.INFER_FUSE = fuse
    for _ in range(3):
        m = k.predict(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m = k.predict(img)
    b = m.boundaries()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    res[fuse] = (m.index.clone(), dt)
    print(f'predict (eval forward + argmax mask), bs={a.bs} {a.dtype}, epilogue fusion {"on" if fuse else "off"}: {1e3 * dt:.2f} ms/batch = {a.bs / dt:.1f} B-scans/s; '
          f'boundaries {tuple(b.shape)}')
if a.bs <= 2:            # launch-bound regime: hipGraph replay (default) vs eager launches
    ops.INFER_FUSE = True
    for graph in (True, False):
        k.use_graph = graph
        for _ in range(3):
            m = k.predict(img)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            m = k.predict(img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f'predict bs={a.bs}: hipGraph replay {"on" if graph else "off"}: {1e3 * dt:.2f} ms/batch = {a.bs / dt:.1f} B-scans/s')
if len(res) == 2:
    agree = (res[True][0] == res[False][0]).float().mean().item()
    print(f'mask agreement fused vs op-by-op: {100 * agree:.4f} %')
