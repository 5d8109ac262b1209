"""Reference checkpoints (SURVEY 8(b) "Checkpoints", 8(f)3): `torch.save(state_dict)` files with the reference's key names, loaded
the way the reference loads them (`load_state_dict(strict=False)`, kite/loopback.py:82, kite/main.py:70, onnx/onnx_save.py:34-38).

Two layouts exist among the files the reference ships (task1/onnx/*.pt):
  * current (`nets/tcct.py`, e.g. tcct_duke.pt): 514 model keys incl. `base.t321..t324`;
  * legacy (`onnx/tcct_goals.py:949-1036`, e.g. tcct_goals.pt / tcct_hcms.pt / tcct_heg.pt): no `base.t32x`, aux heads on the decoder
    outputs -> build the model with `legacy_heads=True`.
Keys of training-time augmentation modules (`aug.*`) and of RegNet variants the model does not have are reported as unexpected, never
loaded.  Host-side only: tensors are copied into the parameter holders, the kernels read them from there."""
import numpy as np
import torch

from ._lib import TcctError


def read_checkpoint(src):
    """src: path to a torch .pt/.pth state_dict, path to an .npz written by oracle/make_golden_ckpt.py (bf16 bit patterns under
    'w::<key>', integer buffers under 'i::<key>'), or a dict -> {key: CPU tensor}"""
    if isinstance(src, dict):
        return {k: (v if torch.is_tensor(v) else torch.as_tensor(v)) for k, v in src.items()}
    if str(src).endswith('.npz'):
        z = np.load(src, allow_pickle=False)
        sd = {}
        for k in z.files:
            if k.startswith('w::'):
                sd[k[3:]] = torch.from_numpy(z[k].view(np.int16).copy()).view(torch.bfloat16).to(torch.float32)
            elif k.startswith('i::'):
                sd[k[3:]] = torch.from_numpy(np.asarray(z[k]).copy())
        if not sd:
            raise TcctError(f'{src}: no checkpoint tensors (w::/i:: keys) inside')
        return sd
    sd = torch.load(src, map_location='cpu', weights_only=True)
    if not isinstance(sd, dict):
        raise TcctError(f'{src}: expected a state_dict')
    return sd


def describe(sd):
    """-> dict(n_class, legacy_heads) inferred from the tensors"""
    if 'base.aux0.weight' not in sd:
        raise TcctError("not a RegNet(stc_tt) checkpoint: key 'base.aux0.weight' is missing")
    return dict(n_class=int(sd['base.aux0.weight'].shape[0]), legacy_heads='base.t321.weight' not in sd)


def load_reference_checkpoint(model, src):
    """strict=False load with the reference's semantics; returns (missing_keys, unexpected_keys).  Raises when the checkpoint layout
    (current / legacy heads) or the class count does not match the model, instead of silently leaving heads at their random init."""
    sd = read_checkpoint(src)
    info = describe(sd)
    base = model.base if hasattr(model, 'base') else model
    if bool(getattr(base, 'legacy_heads', False)) != info['legacy_heads']:
        raise TcctError(f"checkpoint layout is {'legacy (no t32x)' if info['legacy_heads'] else 'current'}: build the model with "
                        f"legacy_heads={info['legacy_heads']} (tcct_amd.checkpoint.model_from_checkpoint does)")
    if base.aux0.weight.shape[0] != info['n_class']:
        raise TcctError(f"checkpoint has {info['n_class']} classes, the model {base.aux0.weight.shape[0]}")
    own = model.state_dict()
    bad = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    if bad:
        raise TcctError(f'shape mismatch for {bad[:4]}')
    msg = model.load_state_dict(sd, strict=False)
    return sorted(msg.missing_keys), sorted(msg.unexpected_keys)


def model_from_checkpoint(src, compute_dtype=torch.float32, device='cuda'):
    """RegNet(stc_tt(n_class)) with the layout the checkpoint needs, loaded, on `device`, in eval mode"""
    from .nets import stc_tt, RegNet
    sd = read_checkpoint(src)
    info = describe(sd)
    net = RegNet(stc_tt(info['n_class'], compute_dtype=compute_dtype, legacy_heads=info['legacy_heads']), out_channels=info['n_class'])
    missing, unexpected = load_reference_checkpoint(net, sd)
    net = net.to(device).eval()
    net.load_report = dict(missing=missing, unexpected=unexpected, **info)
    return net
