"""TEST INFRASTRUCTURE (build container only, needs /root/reference): fixtures for the sibling variants of stc_tt that the
reference defines next to it (nets/tcct.py:1048-1053 gtc_tt, :1120-1134 cnnu / vitu).  Formula weights (oracle.formula_state_dict),
formula input, REAL reference forward: eval-mode logits of all four heads for every variant, train-mode logits (batch-statistics
BatchNorm, DropPath off) for all three; GateFusion's four random alpha fields (torch.rand, nets/tcct.py:925) are recorded as inputs."""
import contextlib, io, os, sys
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
    for name in ('gtc_tt', 'cnnu', 'vitu'):
        with contextlib.redirect_stdout(io.StringIO()):
            model = nets.RegNet(getattr(nets, name)(5), con='cos', out_channels=5)
        keys = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
        model.load_state_dict(O.formula_state_dict(keys), strict=True)
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
        if name == 'gtc_tt':
            assert len(draws) == 4
            for j, d in enumerate(draws):
                out[f'gtc_tt_field{j}'] = d.numpy()
        print(name, 'eval logit range', float(ev[0].min()), float(ev[0].max()))
    path = os.path.join(HERE, '..', 'tests', 'golden', 'variants_2x32x64.npz')
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, 'KiB')
