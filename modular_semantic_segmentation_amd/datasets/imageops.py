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


def png_channels(filename):
    """Decode a non-interlaced 8- or 16-bit grey / RGB / RGBA PNG to [H,W,channels] at full bit depth.

    SYNTHIA stores class ids in the first channel of 16-bit RGB PNGs, which PIL truncates to the high
    byte; the reference decodes them with pypng (`synthia.py:215-228` `one_channel_image_reader`).
    This is the PNG specification's inflate + per-row unfilter, nothing else."""
    import struct
    import zlib
    with open(filename, 'rb') as f:
        raw = f.read()
    if raw[:8] != b'\x89PNG\r\n\x1a\n':
        raise ValueError('%s is not a PNG file' % filename)
    pos, idat, header = 8, [], None
    while pos < len(raw):
        length, kind = struct.unpack('>I4s', raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + length]
        pos += 12 + length
        if kind == b'IHDR':
            header = struct.unpack('>IIBBBBB', body)
        elif kind == b'IDAT':
            idat.append(body)
        elif kind == b'IEND':
            break
    width, height, depth, colour, _, _, interlace = header
    channels = {0: 1, 2: 3, 4: 2, 6: 4}.get(colour)
    if channels is None or depth not in (8, 16) or interlace:
        raise ValueError('unsupported PNG layout in %s (colour type %d, depth %d, interlace %d)'
                         % (filename, colour, depth, interlace))
    bpp = channels * depth // 8
    stride = width * bpp
    data = np.frombuffer(zlib.decompress(b''.join(idat)), dtype=np.uint8).reshape(height, stride + 1)
    out = np.zeros((height, stride), dtype=np.uint8)
    above = np.zeros(stride, dtype=np.int64)
    for y in range(height):
        kind, line = int(data[y, 0]), data[y, 1:].astype(np.int64)
        if kind == 0:
            row = line
        elif kind == 2:
            row = (line + above) & 255
        elif kind == 1:
            # each byte lane (stride bpp) is a running sum
            row = (np.cumsum(line.reshape(width, bpp), axis=0) & 255).reshape(-1)
        elif kind in (3, 4):
            row = np.zeros(stride, dtype=np.int64)
            cur, up = row.tolist(), above.tolist()
            src = line.tolist()
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = up[i]
                if kind == 3:
                    pred = (a + b) >> 1
                else:
                    c = up[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (src[i] + pred) & 255
            row = np.asarray(cur, dtype=np.int64)
        else:
            raise ValueError('bad PNG filter type %d in %s' % (kind, filename))
        out[y] = row
        above = row
    if depth == 16:
        out = out.reshape(height, width, channels, 2)
        return (out[..., 0].astype(np.uint16) << 8) | out[..., 1]
    return out.reshape(height, width, channels)


def one_channel_image_reader(filename, datatype):
    """First channel of a PNG as `datatype`: SYNTHIA's class-id and depth encoding (synthia.py:215-228)."""
    return png_channels(filename)[:, :, 0].astype(datatype)
