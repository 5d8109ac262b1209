"""GOALS B-scan preprocessing on the GPU: what reference data/octnpy.py (`goals` branch: rows 0:608, nearest resize
to 608x512 and back to 608x1100, label gray level = class * 30) and data/octgen.py (256x256 crops, flips, ToTensor) do with OpenCV and
albumentations on the CPU, as one uint8 gather kernel (tcct_u8_gather2d) per step -- real B-scans can feed the kernels without cv2.
Inputs are what cv2.imread returns: uint8 HWC images / HW gray label maps, batched on the device.  Random draws (crop corner, flip
flags) are arguments: the reference's albumentations RNG stream is not restated."""
import torch

from .._lib import lib, TcctError

ROWS = (0, 608)
PREP_HW = (608, 512)
POST_HW = (608, 1100)
CANVAS_HW = (800, 1100)
DIVIDE = 30


def _gather(src, dst_hw, src_roi=None, dst_roi=None, flipy=False, flipx=False, mul=1, div=1, fill=0):
    if not (src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous() and src.dim() in (3, 4)):
        raise TcctError('goals preprocessing expects a contiguous CUDA uint8 tensor [B,H,W] or [B,H,W,C] (no CPU fallback)')
    B, SH, SW = src.shape[:3]
    C = src.shape[3] if src.dim() == 4 else 1
    DH, DW = dst_hw
    sy0, sx0, sh, sw = src_roi if src_roi is not None else (0, 0, SH, SW)
    dy0, dx0, dh, dw = dst_roi if dst_roi is not None else (0, 0, DH, DW)
    out = torch.empty((B, DH, DW) + ((C,) if src.dim() == 4 else ()), device=src.device, dtype=torch.uint8)
    lib.u8_gather2d(src, out, B, SH, SW, C, sy0, sx0, sh, sw, DH, DW, dy0, dx0, dh, dw, int(flipy), int(flipx), mul, div, fill)
    return out


def prep(img, lab):
    """readPair: img uint8 [B,H,W,3], lab uint8 gray [B,H,W] -> (img [B,608,512,3], class map [B,608,512])"""
    r0, r1 = ROWS
    H, W = img.shape[1], img.shape[2]
    if H < r1 or lab.shape[1:3] != (H, W):
        raise TcctError(f'GOALS B-scans are at least {r1} rows high and image/label sizes must match (got {tuple(img.shape)}, {tuple(lab.shape)})')
    return (_gather(img, PREP_HW, src_roi=(r0, 0, r1 - r0, W)),
            _gather(lab, PREP_HW, src_roi=(r0, 0, r1 - r0, W), div=DIVIDE))


def post(mask):
    """postprocess: class map uint8 [B,608,512] -> gray-level label image [B,800,1100] (class * 30 in rows 0:608, 0 below)"""
    if tuple(mask.shape[1:]) != PREP_HW:
        raise TcctError(f'post expects [B,{PREP_HW[0]},{PREP_HW[1]}] class maps')
    return _gather(mask, CANVAS_HW, dst_roi=(ROWS[0], 0, POST_HW[0], POST_HW[1]), mul=DIVIDE)


def crop_flip(a, y0, x0, h, w, flipx=False, flipy=False):
    """albumentations crop at a given corner + HorizontalFlip / VerticalFlip (draws are the caller's)"""
    return _gather(a, (h, w), src_roi=(y0, x0, h, w), flipy=flipy, flipx=flipx)


def to_model_input(img_u8):
    """ToTensor of data/octgen.py:124-126: uint8 [B,H,W,3] -> float32 [B,3,H,W] in [0,1] (plumbing; the model's own input kernel
    converts it to the NHWC compute layout)"""
    return img_u8.permute(0, 3, 1, 2).float().div_(255).clamp_(0, 1).contiguous()
