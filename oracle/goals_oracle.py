"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the GOALS image preprocessing of the reference (data/octnpy.py:82-85 `goals`
branch, :91-112 postprocess, :119-130 readPair; data/octgen.py:9-24 crop / flips).  Only tests/ may import this file.

The arithmetic lives in third-party code that is absent from /root/reference and from this image: OpenCV (`cv2.resize(...,
interpolation=cv2.INTER_NEAREST)`, called through albumentations.Resize; versions not pinned anywhere in the reference,
README.md:22-25) and albumentations (PadIfNeeded / CropNonEmptyMaskIfExists / flips).  PARITY UNPINNED: the nearest-neighbour rule
below is OpenCV 4.x `resizeNN` (modules/imgproc/src/resize.cpp): x_ofs[x] = min(cvFloor(x * ifx), ssize.width - 1) with
ifx = 1 / inv_scale_x, inv_scale_x = (double)dsize.width / ssize.width, rows likewise; crops and flips are index arithmetic with the
random draws (crop corner, flip flags) supplied by the caller, so no RNG stream is restated.
"""
import numpy as np

GOALS_ROWS = (0, 608)          # data/octnpy.py:83  height_stt, height_end
GOALS_PREP = (608, 512)        # data/octnpy.py:84  prep_tran: Resize(height=608, width=512, INTER_NEAREST)
GOALS_POST = (608, 1100)       # data/octnpy.py:85  post_tran: Resize(height=608, width=1100, INTER_NEAREST)
DIVIDE = 30                    # data/octnpy.py:118 label gray level = class * 30


def nn_index(dn, sn):
    """cv2.INTER_NEAREST source index for every destination index (OpenCV resizeNN)."""
    inv_scale = float(dn) / float(sn)
    ifx = 1.0 / inv_scale
    return np.minimum(np.floor(np.arange(dn, dtype=np.float64) * ifx).astype(np.int64), sn - 1)


def resize_nearest(img, dh, dw):
    """img [..., H, W] or [..., H, W, C] given as (array, has_channels) -> nearest-neighbour resize of the two spatial axes"""
    a, ch = img
    ax = a.ndim - (3 if ch else 2)
    ys, xs = nn_index(dh, a.shape[ax]), nn_index(dw, a.shape[ax + 1])
    return np.take(np.take(a, ys, axis=ax), xs, axis=ax + 1)


def goals_prep(img, lab):
    """readPair (data/octnpy.py:119-130): img uint8 [B,H,W,3], lab uint8 gray [B,H,W] -> img [B,608,512,3], class map [B,608,512]"""
    r0, r1 = GOALS_ROWS
    img, lab = img[:, r0:r1], (lab // DIVIDE)[:, r0:r1]
    return resize_nearest((img, True), *GOALS_PREP), resize_nearest((lab, False), *GOALS_PREP)


def goals_post(mask, canvas_hw=(800, 1100)):
    """postprocess (data/octnpy.py:95-112): class map [B,608,512] -> gray-level label image [B,800,1100] (rows 0:608 filled)"""
    g = (mask.astype(np.int64) * DIVIDE).astype(np.uint8)
    g = resize_nearest((g, False), *GOALS_POST)
    out = np.zeros((mask.shape[0],) + tuple(canvas_hw), np.uint8)
    out[:, GOALS_ROWS[0]:GOALS_ROWS[1], :] = g
    return out


def crop_flip(a, has_ch, y0, x0, h, w, flipx, flipy):
    """albumentations crop (corner given) followed by HorizontalFlip / VerticalFlip (data/octgen.py:9-24)"""
    ax = a.ndim - (3 if has_ch else 2)
    sl = [slice(None)] * a.ndim
    sl[ax], sl[ax + 1] = slice(y0, y0 + h), slice(x0, x0 + w)
    o = a[tuple(sl)]
    if flipx:
        o = np.flip(o, axis=ax + 1)
    if flipy:
        o = np.flip(o, axis=ax)
    return np.ascontiguousarray(o)
