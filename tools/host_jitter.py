"""Per-step host enqueue time over many steps: how often does the host stall, and is the cyclic garbage collector the cause?  Three arms of 150 steps: default gc, after
gc.collect() + gc.freeze() (the objects alive so far -- the model, the trainer, the packed weights -- leave the collector's generations), and gc disabled.

    python tools/host_jitter.py"""
import argparse
import gc
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def run(k, img, lab, n):
    ts = []
    torch.cuda.synchronize()
    w0, c0 = time.perf_counter(), time.process_time()
    for _ in range(n):
        t0 = time.perf_counter()
        k.train_step(img, lab)
        ts.append(1e3 * (time.perf_counter() - t0))
    c1 = time.process_time()
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - w0) / n
    ts.sort()
    # CPU time of the process (all threads: the enqueueing thread + autograd's): what the step costs the host, free of the waits on a full launch queue that the
    # wall-clock enqueue time contains whenever the host is ahead of the GPU
    return wall, 1e3 * (c1 - c0) / n, statistics.median(ts), ts[int(0.99 * n)], ts[-1]


def main():
    ba = argparse.Namespace(los='di', bs=8, height=800, width=1100, dtype='bf16', att='pool')
    k, ds, _ = bench.build_trainer(ba, 1)
    img, lab, _, _ = ds.parse(ds.make_batch(8, seed=2023))
    img, lab = img.contiguous(), lab.contiguous()
    k.model.train()
    for _ in range(10):
        k.train_step(img, lab)
    n = 150
    print('arm: wall ms/step, CPU ms/step (process time, all threads), host enqueue wall median / p99 / max ms; gc collections per generation')
    c0 = gc.get_stats()
    print('default gc      ', ' '.join(f'{v:8.2f}' for v in run(k, img, lab, n)), [s['collections'] - c['collections'] for s, c in zip(gc.get_stats(), c0)])
    gc.collect()
    gc.freeze()
    c0 = gc.get_stats()
    print('gc.freeze()     ', ' '.join(f'{v:8.2f}' for v in run(k, img, lab, n)), [s['collections'] - c['collections'] for s, c in zip(gc.get_stats(), c0)])
    gc.disable()
    print('gc disabled     ', ' '.join(f'{v:8.2f}' for v in run(k, img, lab, n)))
    gc.enable()


if __name__ == '__main__':
    main()
