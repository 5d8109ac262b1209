"""Find what stalls single steps for 0.6-2 s: time Python GC passes and per-phase host time of every step."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tcct_amd import ops
a = bench.parse()
k, ds, args = bench.build_trainer(a, 1)
k.model.train()
img, lab, _, _ = ds.parse(ds.make_batch(a.bs, 2023))
gcs = []
def cb(phase, info):
    if phase == 'start':
        cb.t = time.perf_counter()
    else:
        gcs.append((info['generation'], 1e3 * (time.perf_counter() - cb.t), info['collected']))
gc.callbacks.append(cb)
for it in range(a.steps):
    t0 = time.perf_counter()
    k.optimG.zero_grad(set_to_none=True)
    ops.begin_step(k.device)
    tot, _ = k.calc_loss(img, lab, want_log=False)
    t1 = time.perf_counter()
    tot.backward()
    t2 = time.perf_counter()
    ops.end_step()
    k.optimG.step()
    t3 = time.perf_counter()
    big = [g for g in gcs if g[1] > 5]
    ms = torch.cuda.memory_stats()
    print(f'step {it}: fwd {1e3*(t1-t0):.1f} bwd {1e3*(t2-t1):.1f} opt {1e3*(t3-t2):.1f} ms; gc slow {big}; segments {ms["num_device_alloc"]} frees {ms["num_device_free"]} reserved {ms["reserved_bytes.all.current"] / 2**30:.1f} GB retries {ms["num_alloc_retries"]}', flush=True)
    gcs.clear()
torch.cuda.synchronize()
print('objects tracked by gc:', len(gc.get_objects()), 'thresholds', gc.get_threshold(), 'mem reserved GB', torch.cuda.memory_reserved() / 2**30)
