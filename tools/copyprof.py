"""Which call sites issue device-to-device copies / torch elementwise kernels during one training step (GPU box)."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def main():
    sys.argv = sys.argv[:1]
    a = bench.parse()
    k, ds, args = bench.build_trainer(a, 1)
    k.model.train()
    img, lab, _, _ = ds.parse(ds.make_batch(a.bs, seed=2023))
    img, lab = img.contiguous(), lab.contiguous()
    for _ in range(3):
        k.train_step(img, lab)
    torch.cuda.synchronize()
    counts = collections.Counter()
    sizes = collections.Counter()
    real_copy = torch.Tensor.copy_
    def copy_(self, src, *a, **kw):
        st = traceback.extract_stack(limit=6)[:-1]
        key = ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st[-3:]))
        counts[key] += 1
        sizes[key] += self.numel() * self.element_size()
        return real_copy(self, src, *a, **kw)
    torch.Tensor.copy_ = copy_
    # optimizer copy branch census
    opt = k.optimG
    k.train_step(img, lab)
    torch.Tensor.copy_ = real_copy
    for key, c in counts.most_common(20):
        print(c, sizes[key], key)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
        k.train_step(img, lab)
        torch.cuda.synchronize()
    agg = collections.Counter(); tim = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            continue
        n = e.name
        if n.startswith('aten::') and e.cpu_parent is None or (e.cpu_parent is not None and not e.cpu_parent.name.startswith('aten::') and n.startswith('aten::')):
            dev = sum(k_.duration for k_ in e.kernels) if hasattr(e, 'kernels') else 0
            par = e.cpu_parent.name if e.cpu_parent is not None else '-'
            stack = [s for s in (e.stack or []) if 'tcct_amd' in s or 'bench.py' in s][:2]
            key = (n, par, ' | '.join(s.split('/')[-1] for s in stack))
            agg[key] += 1; tim[key] += dev
    for key, t in tim.most_common(40):
        print(f'{t:9.0f} us  x{agg[key]:4d}  {key}')

if __name__ == '__main__':
    main()
