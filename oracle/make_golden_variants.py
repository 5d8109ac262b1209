"""TEST INFRASTRUCTURE (build container only, needs /root/reference): fixtures for the sibling variants of stc_tt that the
reference defines next to it (nets/tcct.py:1048-1061 gtc_tt / gtc_tb, :1097-1102 stc_tb, :1117-1134 pnnu / cnnu / vitu).  Formula weights (oracle.formula_state_dict),
formula input, REAL reference forward: eval-mode logits of all four heads for every variant, train-mode logits (batch-statistics
BatchNorm, DropPath off) for all of them; GateFusion's four random alpha fields (torch.rand, nets/tcct.py:925) are recorded as inputs.  Variants whose parameter
shapes differ from stc_tt's (wide CNN encoder, 3-tap PlainCNNBlock) get their state_dict key/shape list in variants_keys.json."""
import contextlib, io, json, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport
import tcct_oracle as O

if __name__ == '__main__':
    _refimport.install()
    import nets
    img, lab = O.synth_batch(2, 32, 64, seed=11)
    out = {'img': img.numpy()}
    keys_out = {}
    for name in ('gtc_tt', 'cnnu', 'vitu', 'stc_tb', 'gtc_tb', 'pnnu'):
        with contextlib.redirect_stdout(io.StringIO()):
            model = nets.RegNet(getattr(nets, name)(5), con='cos', out_channels=5)
        keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        model.load_state_dict(O.formula_state_dict(keys), strict=True)
        if name in ('stc_tb', 'gtc_tb', 'pnnu'):
            keys_out[name] = [[k, list(s)] for k, s in keys]
        for m in model.modules():
            if isinstance(m, _refimport.DropPath):
                m.drop_prob = 0.
        model.eval()
        with torch.no_grad():
            ev = model(img)
        out[f'{name}_eval'] = np.stack([o.numpy() for o in ev])
        model.train()
        draws = []
        real_rand = torch.rand

        def rec_rand(*a, **k):             # GateFusion draws its alpha field with torch.rand (nets/tcct.py:925): record the draws
            r = real_rand(*a, **k)
            draws.append(r.clone())
            return r
        torch.rand = rec_rand
        try:
            with torch.no_grad():
                tr = model(img)
        finally:
            torch.rand = real_rand
        out[f'{name}_train'] = np.stack([o.numpy() for o in tr])
        if name.startswith('gtc'):
            assert len(draws) == 4
            for j, d in enumerate(draws):
                out[f'{name}_field{j}'] = d.numpy()
        print(name, 'eval logit range', float(ev[0].min()), float(ev[0].max()))
    path = os.path.join(HERE, '..', 'tests', 'golden', 'variants_2x32x64.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, 'KiB')
    json.dump(keys_out, open(os.path.join(HERE, '..', 'tests', 'golden', 'variants_keys.json'), 'w'))
