"""(a) the SAME trained weights evaluated in fp32 and in bf16: validation Dice difference (inference parity of the benchmarked precision);
(b) two fp32 training runs from the same start (the weight-gradient atomics reorder sums): the run-to-run floor of 'Dice after N steps';
(c) fp32 vs bf16 training runs."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tcct_amd.nets import stc_tt, RegNet
from tcct_amd.kite import KiteSeg
from tcct_amd.kite.losses import MDiceLoss
from tcct_amd.data import SynthOCT

H, lr, steps = 128, float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3, int(sys.argv[2]) if len(sys.argv) > 2 else 60
torch.manual_seed(0)
sd0 = {k: v.clone() for k, v in RegNet(stc_tt(5), con='cos', out_channels=5).state_dict().items()}
ds = SynthOCT(height=H, width=H, device='cuda', n_train=2 * steps, n_val=16)


def make(dt, sd):
    model = RegNet(stc_tt(5, compute_dtype=dt), con='cos', out_channels=5)
    model.load_state_dict(sd)
    args = argparse.Namespace(los='di', lr=lr, gpu='0', pl=False, bs=2, coff_ds=1, udh=False, reg=False, epl=False, coff_udh=1, coff_reg=.1, coff_epl=.1, bug=False)
    k = KiteSeg(model=model, dataset=ds, root='/tmp/dice_probe2', args=args)
    k.model.base.base_vit.drop_probs = [0.0] * 4
    for g in k.optimG.param_groups:
        g['lr'] = lr
    return k


def train(k):
    k.model.train()
    for i, b in enumerate(ds.trainSet(bs=2)):
        img, lab, _, _ = ds.parse(b)
        k.train_step(img, lab)
        if i + 1 == steps:
            break
    return k


def dice(k):
    k.model.eval()
    tot, n = 0.0, 0
    with torch.no_grad():
        for b in ds.valSet(bs=1):
            img, lab, _, _ = ds.parse(b)
            tot += MDiceLoss.scorem(k.predict(img), lab, start_idx=1).item(); n += 1
    return tot / n


ka, kb, kc = train(make(torch.float32, sd0)), train(make(torch.float32, sd0)), train(make(torch.bfloat16, sd0))
da, db, dc = dice(ka), dice(kb), dice(kc)
print(f'lr {lr} steps {steps}: fp32 run A {da:.5f}  fp32 run B {db:.5f}  bf16 run {dc:.5f}   |A-B| {abs(da - db):.2e}  |A-bf16| {abs(da - dc):.2e}')
sdt = {k_: v.clone() for k_, v in ka.model.state_dict().items()}
d32, d16 = dice(make(torch.float32, sdt)), dice(make(torch.bfloat16, sdt))
print(f'same fp32-trained weights: eval fp32 {d32:.5f}  eval bf16 {d16:.5f}  |delta| {abs(d32 - d16):.2e}')
sdt = {k_: v.clone() for k_, v in kc.model.state_dict().items()}
d32, d16 = dice(make(torch.float32, sdt)), dice(make(torch.bfloat16, sdt))
print(f'same bf16-trained weights: eval fp32 {d32:.5f}  eval bf16 {d16:.5f}  |delta| {abs(d32 - d16):.2e}')
