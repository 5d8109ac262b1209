"""Synthetic GOALS-shaped OCT dataset implementing the reference's dataset protocol (data/octgen.py:60-62,81-93,124-127):
`.out_channels`, `.trainSet(bs)`, `.valSet(bs)`, `.testSet(bs)` -> iterables of {'img','lab','tag'}, `.parse(batch)`.

A B-scan is 1 x 800 x 1100 (BASELINE.json).  The network needs 3 channels and H, W % 16 == 0 (reference tcct.py:873 and
the 138-vs-137 mismatch at level 3), so `parse` does the loader-side prep: img stays 1-channel (replicated 1->3 inside the
NHWC conversion kernel, same values as cv2.IMREAD_COLOR of a gray PNG, data/octnpy.py:119) and W is zero-padded on the right
to the next multiple of 16 (labels padded with class 0).  Batches are generated on the device: nothing here is the product,
it only feeds the kernels."""
import torch


def _pad16(n):
    return (n + 15) // 16 * 16


class SynthBatches:
    def __init__(self, ds, bs, n_batches, seed):
        self.ds, self.bs, self.n, self.seed = ds, bs, n_batches, seed

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.ds.make_batch(self.bs, self.seed + i)


class SynthOCT:
    def __init__(self, dbname='synth', height=800, width=1100, n_class=5, n_train=16, n_val=4, device=None, seed=2023):
        self.__name__ = dbname
        self.out_channels = n_class
        self.H, self.W = height, width
        self.n_train, self.n_val = n_train, n_val
        self.device = torch.device(device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu'))
        self.seed = seed

    def make_batch(self, bs, seed):
        """img [bs,1,H,W] fp32 in [0,1): speckle x layer-intensity profile; lab [bs,H,W] int64: 4 sorted smooth boundaries
        per column (min thickness >= H/50 px) -> 5 classes each >> 32 px."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        H, W, C, dev = self.H, self.W, self.out_channels, self.device
        xs = torch.arange(W, device=dev, dtype=torch.float32) / max(W - 1, 1)
        ph = torch.rand((bs, C - 1, 1), generator=g, device=dev) * 6.2832
        fr = 1.0 + 3.0 * torch.rand((bs, C - 1, 1), generator=g, device=dev)
        amp = 0.02 + 0.03 * torch.rand((bs, C - 1, 1), generator=g, device=dev)
        base = torch.linspace(0.15, 0.8, C - 1, device=dev).view(1, C - 1, 1)
        bnd = (base + amp * torch.sin(6.2832 * fr * xs.view(1, 1, W) + ph)) * H
        bnd, _ = torch.sort(bnd, dim=1)
        rows = torch.arange(H, device=dev, dtype=torch.float32).view(1, 1, H, 1)
        lab = (rows >= bnd.unsqueeze(2)).sum(1)                                  # [bs,H,W] int64
        inten = 0.2 + 0.15 * lab.float()
        img = (inten * (0.5 + 0.5 * torch.rand((bs, H, W), generator=g, device=dev))).clamp_(0, 0.999).unsqueeze(1)
        return {'img': img, 'lab': lab, 'tag': [f'synth_{seed}_{i}' for i in range(bs)]}

    def trainSet(self, bs=8, data='train'):
        return SynthBatches(self, bs, max(1, self.n_train // bs), self.seed)

    def valSet(self, bs=1, data='val'):
        return SynthBatches(self, bs, max(1, self.n_val // bs), self.seed + 100003)

    def testSet(self, bs=1, data='test'):
        return SynthBatches(self, bs, max(1, self.n_val // bs), self.seed + 200003)

    def parse(self, pics):
        """-> (img [B,1,H,Wp] fp32, lab [B,H,Wp] int64, tag, None) with Wp = W rounded up to a multiple of 16"""
        img, lab = pics['img'], pics['lab']
        Hp, Wp = _pad16(img.shape[-2]), _pad16(img.shape[-1])
        if (Hp, Wp) != tuple(img.shape[-2:]):
            img = torch.nn.functional.pad(img, (0, Wp - img.shape[-1], 0, Hp - img.shape[-2]))
            lab = torch.nn.functional.pad(lab, (0, Wp - lab.shape[-1], 0, Hp - lab.shape[-2]))
        return img, lab, pics['tag'], None
