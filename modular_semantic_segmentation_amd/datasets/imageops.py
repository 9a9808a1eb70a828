"""Image file reading and geometric resampling for the dataset readers, on numpy + PIL.

The reference reads and resamples with OpenCV (`cityscapes.py:159-184`, `augmentation.py:183-198`);
cv2 is not a dependency here, so the few operations the readers need are restated with cv2's
conventions: BGR channel order, any-depth single channel reads, half-pixel-centre bilinear
(`INTER_LINEAR`) and floor-index nearest (`INTER_NEAREST`) resizing, inverse-mapped affine warps with
a zero border.  cv2 is absent from the build image, so these are convention-level restatements, not
bit-pinned against cv2 (its fixed-point bilinear weights can differ in the last bit).
"""
import numpy as np


def imread_bgr(filename):
    """uint8 [H,W,3] in B,G,R order, like `cv2.imread(filename)`."""
    from PIL import Image
    with Image.open(filename) as im:
        return np.ascontiguousarray(np.asarray(im.convert('RGB'))[:, :, ::-1])


def imread_anydepth(filename):
    """Single-channel image at its stored depth, like `cv2.imread(filename, cv2.IMREAD_ANYDEPTH)`:
    uint16 for 16-bit PNGs (disparity / depth), uint8 for 8-bit ones (label ids)."""
    from PIL import Image
    with Image.open(filename) as im:
        if im.mode in ('I;16', 'I;16B', 'I;16L', 'I'):
            return np.asarray(im).astype(np.uint16)
        return np.asarray(im.convert('L'))


def _restore_dtype(values, like):
    if np.issubdtype(like.dtype, np.integer):
        info = np.iinfo(like.dtype)
        return np.clip(np.rint(values), info.min, info.max).astype(like.dtype)
    return values.astype(like.dtype)


def resize_nearest(image, out_h, out_w):
    """`cv2.resize(..., interpolation=INTER_NEAREST)`: source index = floor(dst * in/out)."""
    h, w = image.shape[:2]
    rows = np.minimum((np.arange(out_h) * (h / out_h)).astype(np.int64), h - 1)
    cols = np.minimum((np.arange(out_w) * (w / out_w)).astype(np.int64), w - 1)
    return image[rows][:, cols]


def resize_linear(image, out_h, out_w):
    """`cv2.resize(..., interpolation=INTER_LINEAR)`: half-pixel centres, edge-clamped, no
    anti-aliasing."""
    h, w = image.shape[:2]

    def taps(n_out, n_in):
        src = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        lo = np.floor(src)
        frac = src - lo
        lo = lo.astype(np.int64)
        return np.clip(lo, 0, n_in - 1), np.clip(lo + 1, 0, n_in - 1), frac

    r0, r1, fr = taps(out_h, h)
    c0, c1, fc = taps(out_w, w)
    img = image.astype(np.float64)
    extra = (1,) * (img.ndim - 2)
    fr = fr.reshape((-1, 1) + extra)
    fc = fc.reshape((1, -1) + extra)
    top = img[r0][:, c0] * (1 - fc) + img[r0][:, c1] * fc
    bottom = img[r1][:, c0] * (1 - fc) + img[r1][:, c1] * fc
    return _restore_dtype(top * (1 - fr) + bottom * fr, image)


def scale_image(image, factor, nearest):
    """`cv2.resize(image, None, fx=factor, fy=factor)`: output size is round(size * factor)."""
    h, w = image.shape[:2]
    out_h, out_w = int(round(h * factor)), int(round(w * factor))
    return (resize_nearest if nearest else resize_linear)(image, out_h, out_w)


def warp_affine(image, matrix, out_w, out_h):
    """`cv2.warpAffine(image, matrix, (out_w, out_h), flags=INTER_LINEAR)`: `matrix` [2,3] maps source
    to destination coordinates (x, y); pixels that fall outside the source read 0."""
    full = np.vstack([np.asarray(matrix, dtype=np.float64), [0.0, 0.0, 1.0]])
    inv = np.linalg.inv(full)
    ys, xs = np.mgrid[0:out_h, 0:out_w].astype(np.float64)
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    h, w = image.shape[:2]
    x0 = np.floor(sx).astype(np.int64)
    y0 = np.floor(sy).astype(np.int64)
    fx = sx - x0
    fy = sy - y0
    img = image.astype(np.float64)
    extra = (1,) * (img.ndim - 2)

    def fetch(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        vals = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return vals * ok.reshape(ok.shape + extra)

    fx = fx.reshape(fx.shape + extra)
    fy = fy.reshape(fy.shape + extra)
    out = (fetch(y0, x0) * (1 - fx) + fetch(y0, x0 + 1) * fx) * (1 - fy) \
        + (fetch(y0 + 1, x0) * (1 - fx) + fetch(y0 + 1, x0 + 1) * fx) * fy
    return _restore_dtype(out, image)
