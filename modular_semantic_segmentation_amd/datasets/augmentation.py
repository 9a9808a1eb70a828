"""Training-time augmentation of an image blob and the multiple-of-16 crop of the data contract.

Host-side counterpart of the reference's `xview/datasets/augmentation.py` (`augmentate` :143-241,
`crop_multiple` :244-262) on numpy only: cv2 resampling is restated in `imageops`, the two imgaug
operators the reference uses are the plain formulas below.  The order of the transforms, their
argument convention (`[probability, low, high]` lists, `False` = off) and which random generator
each draw comes from (`random` vs `numpy.random`) follow the reference, so a seeded run takes the
same decisions.
"""
import math
import random

import numpy as np

from . import imageops


def crop_multiple(data, multiple_of=16):
    """Cut the two leading axes down to multiples of `multiple_of` (augmentation.py:244-262); things
    without a shape pass through."""
    shape = getattr(data, 'shape', None)
    if shape is None or len(shape) < 2:
        return data
    h, w = (int(d) - int(d) % multiple_of for d in shape[:2])
    if (h, w) == tuple(shape[:2]):
        return data
    return data[:h, :w, ...]


def _rotated_canvas(image, degrees):
    """Rotate about the centre onto a canvas that holds the whole rotated image (augmentation.py:8-76).
    Every modality, labels included, is resampled bilinearly there; kept."""
    h, w = image.shape[:2]
    rad = math.radians(degrees)
    a, b = math.cos(rad), math.sin(rad)
    cx, cy = w / 2.0, h / 2.0
    rot = np.array([[a, b, (1 - a) * cx - b * cy],
                    [-b, a, b * cx + (1 - a) * cy]])
    new_w = int(abs(w * a) + abs(h * b))
    new_h = int(abs(w * b) + abs(h * a))
    rot[0, 2] += int(new_w * 0.5 - w * 0.5)
    rot[1, 2] += int(new_h * 0.5 - h * 0.5)
    return imageops.warp_affine(image, rot, new_w, new_h)


def inscribed_rect(w, h, radians):
    """Width and height of the axis-aligned rectangle the reference cuts out of a rotated w x h image
    (`largest_rotated_rect`, augmentation.py:79-116).  The reference's third angle is atan2(bb_w, bb_w),
    i.e. always 45 degrees, whatever the aspect ratio; reproduced because it decides the crop size."""
    quadrant = int(math.floor(radians / (math.pi / 2))) & 3
    alpha = radians if quadrant % 2 == 0 else math.pi - radians
    alpha = (alpha % math.pi + math.pi) % math.pi
    box_w = w * math.cos(alpha) + h * math.sin(alpha)
    box_h = w * math.sin(alpha) + h * math.cos(alpha)
    gamma = math.pi / 4
    delta = math.pi - alpha - gamma
    longest = max(w, h)
    a = longest * math.cos(alpha) * math.sin(alpha) / math.sin(delta)
    y = a * math.cos(gamma)
    x = y * math.tan(gamma)
    return box_w - 2 * x, box_h - 2 * y


def _centre_crop(image, width, height):
    h, w = image.shape[:2]
    width, height = min(width, w), min(height, h)
    cx, cy = int(w * 0.5), int(h * 0.5)
    return image[int(cy - height * 0.5):int(cy + height * 0.5), int(cx - width * 0.5):int(cx + width * 0.5)]


def _shear(image, degrees):
    """Horizontal shear about the image centre, zero border (imgaug `Affine(shear=...)`)."""
    h, w = image.shape[:2]
    t = math.tan(math.radians(degrees))
    mat = np.array([[1.0, -t, t * h / 2.0], [0.0, 1.0, 0.0]])
    return imageops.warp_affine(image, mat, w, h)


def _as_uint8(values):
    return np.clip(np.rint(values), 0, 255).astype(np.uint8)


def flip_labels(labels, c1, c2, prob=0.5):
    """Map c1 onto c2 with probability `prob`, else c2 onto c1 (augmentation.py:132-140)."""
    if np.random.rand() < prob:
        labels[labels == c1] = c2
    else:
        labels[labels == c2] = c1
    return labels


def augmentate(blob, scale=False, crop=False, hflip=False, vflip=False, gamma=False, contrast=False,
               brightness=False, rotate=False, shear=False, label_flip=False, label_merge=False):
    """Augment all modalities of one sample consistently (augmentation.py:143-241).

    scale [p, lo, hi]: resize by a factor from [max(lo, crop/min side), hi], only when cropping;
    crop [p, size]: random size x size window; rotate [p, lo_deg, hi_deg]; shear [p, lo, hi]
    (fractions of the width, only when cropping); hflip / vflip: probability (the reference's hflip
    reverses axis 0 and vflip axis 1, each halved by a second coin; kept); gamma [p, lo, hi],
    contrast [p, lo, hi], brightness [p, lo, hi]: rgb only; label_flip [c1, c2]; label_merge [keep, drop].
    """
    modalities = list(blob.keys())
    do_crop = bool(crop) and crop[0] > random.random()

    if scale and do_crop and scale[0] > random.random():
        h, w = blob[modalities[0]].shape[:2]
        k = random.uniform(max(crop[1] / float(min(h, w)), scale[1]), scale[2])
        for m in modalities:
            blob[m] = imageops.scale_image(blob[m], k, nearest=(m != 'rgb'))

    if rotate and rotate[0] > random.random():
        h, w = blob[modalities[0]].shape[:2]
        degrees = np.random.randint(rotate[1], rotate[2])
        rect = inscribed_rect(w, h, math.radians(degrees))
        for m in modalities:
            blob[m] = _centre_crop(_rotated_canvas(blob[m], degrees), *rect)

    if shear and do_crop and shear[0] > random.random():
        h, w = blob[modalities[0]].shape[:2]
        amount = np.random.randint(shear[1] * w, shear[2] * w) * np.random.choice([-1, 1])
        for m in modalities:
            blob[m] = _shear(blob[m], amount)

    if do_crop:
        h, w = blob[modalities[0]].shape[:2]
        top = random.randint(0, h - crop[1])
        left = random.randint(0, w - crop[1])
        for m in modalities:
            blob[m] = blob[m][top:top + crop[1], left:left + crop[1], ...]

    if hflip and hflip > random.random() and np.random.choice([0, 1]):
        for m in modalities:
            blob[m] = np.flip(blob[m], axis=0)

    if vflip and vflip > random.random() and np.random.choice([0, 1]):
        for m in modalities:
            blob[m] = np.flip(blob[m], axis=1)

    if contrast and 'rgb' in modalities and contrast[0] > np.random.rand():
        alpha = np.random.uniform(contrast[1], contrast[2])
        blob['rgb'] = _as_uint8(128.0 + alpha * (blob['rgb'].astype(np.float64) - 128.0))

    if brightness and 'rgb' in modalities and brightness[0] > np.random.rand():
        offset = np.random.randint(brightness[1], brightness[2] + 1)
        blob['rgb'] = _as_uint8(blob['rgb'].astype(np.float64) + offset)

    if gamma and 'rgb' in modalities and gamma[0] > random.random():
        k = random.uniform(gamma[1], gamma[2])
        lut = (((np.arange(256) / 255.0) ** (1 / k)) * 255).astype('uint8')
        blob['rgb'] = lut[blob['rgb']]

    if label_flip:
        blob['labels'] = flip_labels(blob['labels'], *label_flip)

    if label_merge:
        blob['labels'][blob['labels'] == label_merge[1]] = label_merge[0]

    return blob
