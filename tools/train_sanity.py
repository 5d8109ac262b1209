"""Longer training run on one synthetic batch set: loss must fall and stay finite (bf16, full loss, realistic learning rate)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = bench.parse()
k, ds, args = bench.build_trainer(a, 1)
k.model.train()
for g in k.optimG.param_groups:
    g['lr'] = 2e-3
batches = [ds.parse(ds.make_batch(a.bs, 100 + i))[:2] for i in range(4)]
hist = []
t0 = time.time()
for it in range(a.steps):
    img, lab = batches[it % 4]
    hist.append(k.train_step(img, lab))
    if (it + 1) % 50 == 0:
        v = torch.stack(hist[-50:]).float()
        print(f'steps {it - 48:4d}-{it + 1:4d}: mean loss {v.mean().item():.4f}  max {v.max().item():.4f}  finite {bool(torch.isfinite(v).all())}', flush=True)
k.model.eval()
with torch.no_grad():
    from tcct_amd.kite.losses import MDiceLoss
    img, lab = batches[0]
    sc = MDiceLoss.scorem(k.predict(img), lab, start_idx=1).item()
print(f'{a.steps} steps in {time.time() - t0:.1f} s; train-batch Dice (classes 1..) after training: {sc:.4f}; reserved {torch.cuda.memory_reserved() / 2**30:.1f} GB')
