"""cProfile of the host side of the training step (enqueue only) -- where does the Python/driver time go?"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = bench.parse()
k, ds, args = bench.build_trainer(a, 1)
k.model.train()
img, lab, _, _ = ds.parse(ds.make_batch(a.bs, 2023))
for _ in range(4):
    k.train_step(img, lab)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(4):
    k.train_step(img, lab)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'enqueue {1e3 * (t1 - t0) / 4:.1f} ms/step (under cProfile), wall {1e3 * (t2 - t0) / 4:.1f}')
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
