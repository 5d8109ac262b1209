"""Coarse GPU timeline of the MULTI-STREAM training step without a profiler (rocprofv3 slows the host to ~28 ms per step, so its multi-stream timeline is the
host's, not the GPU's): HIP events recorded on whatever stream is current when a level of either encoder / a decoder block finishes its forward, and -- through
tensor hooks, which autograd runs in the stream context of the node they belong to -- when the backward pass reaches the same tensor.  Median over the timed steps,
milliseconds from the first event of the step.

    python tools/level_timeline.py [--los di] [--steps 8]   ->  table on stdout (gpurun: redirect into gpurun_out/)"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--los', default='di')
    p.add_argument('--bs', type=int, default=8)
    p.add_argument('--steps', type=int, default=8)
    p.add_argument('--set', type=str, default='')
    a = p.parse_args()
    ba = argparse.Namespace(los=a.los, bs=a.bs, height=800, width=1100, dtype='bf16', att='pool')
    from tcct_amd import ops
    for kv in filter(None, a.set.split(',')):
        k_, v_ = kv.split('=')
        cur_ = getattr(ops, k_)
        setattr(ops, k_, int(v_) if isinstance(cur_, int) and not isinstance(cur_, bool) else bool(int(v_)))
    k, ds, _ = bench.build_trainer(ba, 1)
    img, lab, _, _ = ds.parse(ds.make_batch(a.bs, seed=2023))
    img, lab = img.contiguous(), lab.contiguous()
    k.model.train()
    marks = []          # (label, event) of the running step

    def mark(label):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        marks.append((label, ev))

    def first_tensor(o):
        if torch.is_tensor(o):
            return o
        if isinstance(o, (list, tuple)):
            for e in o:
                t = first_tensor(e)
                if t is not None:
                    return t
        return None

    def hook(name):
        def fwd(m, args, out):
            mark('fwd ' + name)
            t = first_tensor(out)
            if t is not None and t.requires_grad:
                t.register_hook(lambda g, n=name: (mark('bwd ' + n), None)[1])
        return fwd
    base = k.model.base
    for i, m in enumerate(base.base_cnn.path_estan):
        m.register_forward_hook(hook(f'CNN L{i}'))
    for i, m in enumerate(base.base_vit.mhca_stages):
        m.register_forward_hook(hook(f'ViT stage {i} (L{i + 1})'))
    for i, m in enumerate(base.base_vit.patch_embed_stages):
        m.register_forward_hook(hook(f'ViT patch-embed {i}'))
    for n in ('dec1', 'dec2', 'dec3'):
        getattr(base, n).register_forward_hook(hook(n))
    base.register_forward_hook(hook('FTC (all heads)'))
    runs = []
    for s in range(3 + a.steps):
        marks.clear()
        mark('step start')
        loss = k.train_step(img, lab)
        mark('step end (optimizer issued)')
        torch.cuda.synchronize()
        if s >= 3:
            t0 = marks[0][1]
            runs.append([(lbl, t0.elapsed_time(ev)) for lbl, ev in marks])
    labels = [l_ for l_, _ in runs[0]]
    rows = []
    for i, l_ in enumerate(labels):
        v = [r[i][1] for r in runs if i < len(r) and r[i][0] == l_]
        rows.append((statistics.median(v), l_))
    rows.sort()
    print(f'# multi-stream step timeline, --los={a.los}, bs {a.bs}, bf16; median of {a.steps} steps, ms from the start of the step (HIP events, no profiler); --set {a.set or "-"}')
    print(f'# loss {float(loss):.4f}; step (start -> optimizer issued on the main stream) {rows[-1][0]:.2f} ms')
    for t, l_ in rows:
        print(f'{t:8.3f}  {l_}')


if __name__ == '__main__':
    main()
