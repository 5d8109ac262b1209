"""TEST INFRASTRUCTURE (runs only in the build container, needs /root/reference): real-weights fixtures.

The reference ships trained checkpoints next to its ONNX scripts (task1/onnx/tcct_duke.pt: current `nets/tcct.py` layout, 9 classes;
task1/onnx/tcct_goals.pt: the older layout of task1/onnx/tcct_goals.py:949-1036 -- no t32x convolutions, aux heads on the decoder
outputs -- 5 classes) and the B-scan its own inference script reads (task1/onnx/oct_duke.png, onnx_infer.py:37-41: the [:160,:160] crop).
This script loads each checkpoint into the REAL reference model exactly as the reference does (RegNet(stc_tt(n)), strict=False,
onnx_save.py:34-38 / tcct_goals.py:1160-1165), runs the eval forward on that crop and stores, as data: the weights (rounded to bf16 and stored as
uint16 bit patterns so the file stays ~2 MB -- fp16 overflows on some BatchNorm running variances; the reference model is run with the
SAME rounded weights), the uint8 input crop, the fp32 logits of the main
head, the argmax masks of all four heads and the missing / unexpected key lists of the reference's own strict=False load."""
import importlib.util, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport
ONNX = '/root/reference/task1/onnx'
OUT = os.path.join(HERE, '..', 'tests', 'golden')


def rounded(sd):
    out = {}
    for k, v in sd.items():
        out[k] = v.to(torch.bfloat16).to(torch.float32) if v.is_floating_point() else v.clone()
    return out


def run(name, ckpt, make_model, n_class):
    sd = rounded(torch.load(os.path.join(ONNX, ckpt), map_location='cpu', weights_only=False))
    torch.manual_seed(0)
    net = make_model(n_class)
    msg = net.load_state_dict(sd, strict=False)
    net.eval()
    from PIL import Image
    crop = np.array(Image.open(os.path.join(ONNX, 'oct_duke.png')).convert('RGB'))[:160, :160]
    x = torch.from_numpy(crop).permute(2, 0, 1)[None].float() / 255
    with torch.no_grad():
        outs = net(x)
    masks = np.stack([o.softmax(1).argmax(1)[0].numpy().astype(np.uint8) for o in outs])
    arrays = {('w::' if v.is_floating_point() else 'i::') + k: (v.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16) if v.is_floating_point() else v.numpy())
              for k, v in sd.items()}
    arrays.update(input_u8=crop, logits0=outs[0][0].numpy().astype(np.float32), masks=masks,
                  missing=np.array(sorted(msg.missing_keys)), unexpected=np.array(sorted(msg.unexpected_keys)), n_class=np.int64(n_class))
    path = os.path.join(OUT, f'ckpt_{name}.npz')
    np.savez_compressed(path, **arrays)
    print(name, 'classes in mask0:', np.unique(masks[0]), 'missing', len(msg.missing_keys), 'unexpected', len(msg.unexpected_keys),
          'logit range', float(outs[0].min()), float(outs[0].max()), '->', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    _refimport.install()
    import nets
    run('duke', 'tcct_duke.pt', lambda n: nets.RegNet(nets.stc_tt(n), out_channels=n), 9)
    spec = importlib.util.spec_from_file_location('tcct_goals_legacy', os.path.join(ONNX, 'tcct_goals.py'))
    legacy = importlib.util.module_from_spec(spec)
    sys.modules['pandas'] = sys.modules.get('pandas') or __import__('pandas')
    spec.loader.exec_module(legacy)
    run('goals_legacy', 'tcct_goals.pt', lambda n: legacy.RegNet(legacy.stc_tt(n), out_channels=n), 5)
