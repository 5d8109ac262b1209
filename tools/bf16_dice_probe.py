"""30 training steps in fp32 and in bf16 from the same initial weights on the same batches, then the validation Dice
(MDiceLoss.scorem(start_idx=1), eval mode) of both: the 'Dice within 1e-3' criterion of BASELINE.json measured on the HIP path."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from tcct_amd.nets import stc_tt, RegNet
from tcct_amd.kite import KiteSeg
from tcct_amd.kite.losses import MDiceLoss
from tcct_amd.data import SynthOCT

H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
torch.manual_seed(0)
ref = RegNet(stc_tt(5), con='cos', out_channels=5)
sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
ds = SynthOCT(height=H, width=H, device='cuda', n_train=2 * steps, n_val=8)
scores = {}
for dt in (torch.float32, torch.bfloat16):
    model = RegNet(stc_tt(5, compute_dtype=dt), con='cos', out_channels=5)
    model.load_state_dict(sd0)
    args = argparse.Namespace(los='di', lr=lr, gpu='0', pl=False, bs=2, coff_ds=1, udh=False, reg=False, epl=False, coff_udh=1, coff_reg=.1, coff_epl=.1, bug=False)
    k = KiteSeg(model=model, dataset=ds, root='/tmp/dice_probe', args=args)
    k.model.base.base_vit.drop_probs = [0.0] * 4
    for g in k.optimG.param_groups:
        g['lr'] = lr
    k.model.train()
    losses = []
    for i, b in enumerate(ds.trainSet(bs=2)):
        img, lab, _, _ = ds.parse(b)
        losses.append(k.train_step(img, lab).item())
        if i + 1 == steps:
            break
    k.model.eval()
    tot, n = 0.0, 0
    with torch.no_grad():
        for b in ds.valSet(bs=1):
            img, lab, _, _ = ds.parse(b)
            m = k.predict(img)
            tot += MDiceLoss.scorem(m, lab, start_idx=1).item(); n += 1
    scores[dt] = tot / n
    print(dt, 'loss first/last', losses[0], losses[-1], 'val dice', scores[dt])
print('H', H, 'lr', lr, 'steps', steps, 'delta dice', abs(scores[torch.float32] - scores[torch.bfloat16]))
