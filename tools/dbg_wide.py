import sys, json, os, numpy as np, torch
sys.path.insert(0, 'oracle')
import tcct_oracle as O
from tcct_amd import nets
z = np.load('tests/golden/variants_2x32x64.npz')
own = json.load(open('tests/golden/variants_keys.json'))
base = json.load(open('tests/golden/state_dict_keys.json'))
img = torch.from_numpy(z['img'])
def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / max(1.0, b.abs().max().item())).item()
for name in ('cnnu', 'stc_tb', 'pnnu'):
    keys = [(k, tuple(s)) for k, s in (own[name] if name in own else base)]
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in O.formula_state_dict(keys).items()}
    kw = {}
    model = nets.RegNet(getattr(nets, name)(5), con='cos', out_channels=5)
    model.load_state_dict(O.formula_state_dict(keys), strict=True)
    model = model.cuda().train()
    model.base.base_vit.drop_probs = [0.0] * 4
    with torch.no_grad():
        tr = model(img.cuda())
    ref = z[f'{name}_train']
    if name == 'stc_tb':
        with torch.no_grad():
            o64, _ = O.ftc_forward(sd64, img.double().repeat(1, 3, 1, 1) if img.shape[1] == 1 else img.double(), True, None)
        print(name, 'hip vs f64', [rel(tr[i], o64[i]) for i in range(4)], 'ref vs f64', [rel(torch.from_numpy(ref[i]), o64[i]) for i in range(4)])
    print(name, 'hip vs ref', [rel(tr[i], torch.from_numpy(ref[i])) for i in range(4)])
    # encoder features level by level (CNN branch) vs fp64 oracle if available
