"""Is the step host-bound?  Compare enqueue time (no sync) with wall time per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ['bench.py'] + sys.argv[1:]
a = bench.parse()
k, ds, args = bench.build_trainer(a, 1)
k.model.train()
img, lab, _, _ = ds.parse(ds.make_batch(a.bs, 2023))
for _ in range(3):
    k.train_step(img, lab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    k.train_step(img, lab)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'bs={a.bs}: enqueue {1e3*(t1-t0)/5:.1f} ms/step, wall {1e3*(t2-t0)/5:.1f} ms/step')
