"""Where does the HOST spend its time enqueueing a training step?  cProfile over N steps of the bench trainer (the GPU runs asynchronously; the step is host-bound whenever
the enqueue time approaches the GPU time -- bench.py reports both as config.step_ms_host_enqueue / step_ms_gpu).

    python tools/host_profile.py [--steps 20] [--los di]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--steps', type=int, default=20)
    p.add_argument('--los', default='di')
    a = p.parse_args()
    ba = argparse.Namespace(los=a.los, bs=8, height=800, width=1100, dtype='bf16', att='pool')
    k, ds, _ = bench.build_trainer(ba, 1)
    img, lab, _, _ = ds.parse(ds.make_batch(8, seed=2023))
    img, lab = img.contiguous(), lab.contiguous()
    k.model.train()
    for _ in range(5):
        k.train_step(img, lab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        k.train_step(img, lab)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{a.steps} steps: host enqueue {1e3 * (t1 - t0) / a.steps:.2f} ms per step, wall {1e3 * (t2 - t0) / a.steps:.2f} ms per step (no profiler)')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        k.train_step(img, lab)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s)
    st.sort_stats('tottime').print_stats(35)
    print(s.getvalue()[:9000])
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(30)
    print(s.getvalue()[:7000])


if __name__ == '__main__':
    main()
