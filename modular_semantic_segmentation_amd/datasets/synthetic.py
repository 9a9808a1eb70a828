"""Procedural RGB-D segmentation data in the reference's data contract (xview/datasets/data_baseclass.py:33-55,
cityscapes.py:170-183): 'rgb' float32 [N,H,W,3] raw 0..255 (BGR), 'depth' float32 [N,H,W,1] raw uint16 range,
'labels' int32 [N,H,W] with 0 = void.

No dataset can be downloaded here, so the accuracy evidence of bench.py / tests (mIoU of the MI355X path against the
fp32 oracle on TRAINED weights) runs on this learnable stand-in: overlapping rectangles and ellipses whose class
determines colour AND depth, built so that neither modality suffices alone --

  * classes (2,3), (4,5), (6,7) have nearly the same colour (20 grey levels apart in one channel, against a pixel
    noise of sigma 30 and an illumination gain of +-15 %) and clearly different depths,
  * depths sit on a ladder 5000 apart (noise sigma 1500, ramp +-800), so the depth expert confuses ladder neighbours
    -- (4,6), (6,3), (3,8), (8,5), (5,7), (7,9), ... -- all of which have clearly different colours, while the
    look-alike colour pairs are at least 20000 apart in depth,
  * per-pixel sensor noise, a per-image illumination gain and a smooth depth ramp on top,

so a single expert confuses its look-alike pairs and the Bayes / Dirichlet fusion of both has something to gain,
as on SYNTHIA (experts 0.72, fusion 0.75-0.77).  Shapes are at least 48 px wide: the FCN
decodes at 1/8 resolution (simple_fcn.py:129-133)."""
import numpy as np

NUM_CLASSES = 12

# BGR colours / depths per class; class 0 (void) is rendered as a dark noisy region at a depth of its own
_COLOUR = np.array([[20, 20, 20], [200, 160, 120],
                    [60, 60, 190], [60, 60, 210], [60, 190, 60], [60, 210, 60], [190, 60, 60], [210, 60, 60],
                    [40, 180, 220], [220, 40, 180], [120, 120, 120], [240, 240, 80]], np.float32)
_DEPTH = np.array([13000, 60000,
                   8000, 28000, 18000, 38000, 23000, 43000,
                   33000, 48000, 53000, 3000], np.float32)


def make_rgbd_shapes(n, h, w, seed=0, num_classes=NUM_CLASSES, rgb_noise=30.0, depth_noise=1500.0):
    """n samples as a dict of arrays.  Deterministic in (n, h, w, seed)."""
    if num_classes != NUM_CLASSES:
        raise ValueError('the procedural palette has %d classes' % NUM_CLASSES)
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    labels = np.empty((n, h, w), np.int32)
    rgb = np.empty((n, h, w, 3), np.float32)
    depth = np.empty((n, h, w, 1), np.float32)
    for i in range(n):
        lab = np.ones((h, w), np.int32)                         # class 1: the backdrop
        for _ in range(int(rng.integers(14, 23))):
            c = int(rng.integers(0, num_classes))               # 0 paints an unlabelled ('void') patch
            sh, sw = int(rng.integers(48, max(49, h // 2))), int(rng.integers(48, max(49, w // 3)))
            cy, cx = int(rng.integers(0, h)), int(rng.integers(0, w))
            y0, y1 = max(0, cy - sh // 2 - 1), min(h, cy + sh // 2 + 2)      # the shape's bounding window
            x0, x1 = max(0, cx - sw // 2 - 1), min(w, cx + sw // 2 + 2)
            wy, wx = yy[y0:y1, x0:x1], xx[y0:y1, x0:x1]
            if rng.random() < 0.5:
                m = (np.abs(wy - cy) <= sh / 2) & (np.abs(wx - cx) <= sw / 2)
            else:
                m = ((wy - cy) / (sh / 2)) ** 2 + ((wx - cx) / (sw / 2)) ** 2 <= 1.0
            lab[y0:y1, x0:x1][m] = c
        gain = rng.uniform(0.85, 1.15)
        ramp = rng.uniform(-800, 800) * (yy / h - 0.5) + rng.uniform(-800, 800) * (xx / w - 0.5)
        img = _COLOUR[lab] * gain + rng.normal(0, rgb_noise, (h, w, 3))
        dep = _DEPTH[lab] + ramp + rng.normal(0, depth_noise, (h, w))
        labels[i] = lab
        rgb[i] = np.clip(np.rint(img), 0, 255)
        depth[i, ..., 0] = np.clip(np.rint(dep), 0, 65535)
    return {'rgb': rgb, 'depth': depth, 'labels': labels}


def data_description(h=None, w=None, num_classes=NUM_CLASSES):
    """(dtypes, shapes, num_classes) as DataBaseclass.get_data_description (data_baseclass.py:33-55)."""
    return ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
            {'rgb': (h, w, 3), 'depth': (h, w, 1), 'labels': (h, w)}, num_classes)


def augmented_stream(clean, modality, seed=0, rgb_noise=30.0, depth_noise=1500.0):
    """Endless stream of per-sample dicts {modality, 'labels'} for `fit` (the tf.data generator dataset of the reference,
    data_baseclass.py:57-126, with the augmentations of augmentation.py:143-262 in spirit: flips and shifts) from
    NOISE-FREE renderings `clean` = make_rgbd_shapes(..., rgb_noise=0, depth_noise=0) held as torch tensors on the
    training device: every sample is a random image, flipped, cyclically shifted, with fresh sensor noise -- so a
    15 M-parameter FCN cannot memorise the few dozen layouts and has to learn colour / depth -> class."""
    import torch
    x, lab = clean[modality], clean['labels']
    n, h, w = lab.shape
    g = torch.Generator(device=x.device).manual_seed(seed)
    sigma = rgb_noise if modality == 'rgb' else depth_noise
    top = 255.0 if modality == 'rgb' else 65535.0
    while True:
        i = int(torch.randint(0, n, (1,), generator=g, device=x.device))
        dy, dx = (int(v) for v in torch.randint(0, max(h, w), (2,), generator=g, device=x.device))
        flips = [d for d, f in zip((0, 1), torch.rand(2, generator=g, device=x.device) < 0.5) if bool(f)]
        img, la = x[i], lab[i]
        if flips:
            img, la = torch.flip(img, flips), torch.flip(la, flips)
        img, la = torch.roll(img, (dy % h, dx % w), (0, 1)), torch.roll(la, (dy % h, dx % w), (0, 1))
        noise = torch.randn(img.shape, generator=g, device=x.device) * sigma
        yield {modality: torch.clamp(torch.round(img + noise), 0, top), 'labels': la}
