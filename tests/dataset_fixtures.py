"""Synthetic miniature Cityscapes / SYNTHIA directory trees for the dataset-reader tests and for
`tests/golden/make_dataset_golden.py` (which runs the reference's readers over the same trees)."""
import json
import os

import numpy as np

CITYSCAPES_TRAIN = {'aachen': 14, 'bremen': 13, 'zurich': 12, 'atlantis': 3}     # atlantis: not a train city
CITYSCAPES_VAL = {'munster': 4, 'frankfurt': 3, 'lindau': 2, 'elsewhere': 2}
IMAGE_HW = (36, 52)                                                               # crops to 32 x 48
SYNTHIA_TRAIN, SYNTHIA_TEST = 20, 9


def _sample_arrays(seed, image_hw=IMAGE_HW):
    rng = np.random.default_rng(seed)
    h, w = image_hw
    return (rng.integers(0, 256, (h, w, 3), dtype=np.uint8),
            rng.integers(0, 30000, (h, w)).astype(np.uint16),
            rng.integers(0, 34, (h, w), dtype=np.uint8))


def _write_png16(filename, array):
    from PIL import Image
    Image.fromarray(array.astype(np.uint16)).save(filename)


def build_cityscapes_tree(root, image_hw=IMAGE_HW):
    """<root>/{leftImg8bit,gtFine,disparity}_trainvaltest/.../<set>/<city>/<city>_<seq>_<frame>_<kind>.png"""
    from PIL import Image
    dirs = {'rgb': ('leftImg8bit_trainvaltest/leftImg8bit', 'leftImg8bit'),
            'labels': ('gtFine_trainvaltest/gtFine', 'gtFine_labelIds'),
            'depth': ('disparity_trainvaltest/disparity', 'disparity')}
    seed = 0
    for fileset, cities in (('train', CITYSCAPES_TRAIN), ('val', CITYSCAPES_VAL)):
        for city, count in cities.items():
            for m in dirs:
                os.makedirs(os.path.join(root, dirs[m][0], fileset, city), exist_ok=True)
            for i in range(count):
                stem = '{}_{:06d}_{:06d}'.format(city, i, 19 * i)
                rgb, depth, ids = _sample_arrays(seed, image_hw)
                seed += 1
                name = {m: os.path.join(root, dirs[m][0], fileset, city, '{}_{}.png'.format(stem, dirs[m][1]))
                        for m in dirs}
                Image.fromarray(rgb).save(name['rgb'])                  # file holds R,G,B
                _write_png16(name['depth'], depth)
                Image.fromarray(ids).save(name['labels'])
    return root


def cityscapes_sample(index):
    """(rgb as R,G,B, disparity, labelIds) written for the index-th image in creation order."""
    return _sample_arrays(index)


def build_synthia_tree(root, image_hw=IMAGE_HW):
    """<root>/RAND_CITYSCAPES/{RGB,Depth,GT/LABELS_NPY}/Stereo_Right/Omni_F/<name> + train_test_split.json"""
    from PIL import Image
    base = os.path.join(root, 'RAND_CITYSCAPES')
    sub = {m: os.path.join(base, folder, 'Stereo_Right/Omni_F') for m, folder in
           (('rgb', 'RGB'), ('depth', 'Depth'), ('labels', 'GT/LABELS_NPY'))}
    for d in sub.values():
        os.makedirs(d, exist_ok=True)
    names = ['{:07d}'.format(i) for i in range(SYNTHIA_TRAIN + SYNTHIA_TEST)]
    for i, name in enumerate(names):
        rgb, depth, _ = _sample_arrays(1000 + i, image_hw)
        labels = np.random.default_rng(2000 + i).integers(0, 23, image_hw).astype(np.uint8)
        Image.fromarray(rgb).save(os.path.join(sub['rgb'], name + '.png'))
        _write_png16(os.path.join(sub['depth'], name + '.png'), depth)
        np.save(os.path.join(sub['labels'], name + '.npy'), labels)
    with open(os.path.join(base, 'train_test_split.json'), 'w') as f:
        json.dump({'trainset': names[:SYNTHIA_TRAIN], 'testset': names[SYNTHIA_TRAIN:]}, f)
    return root


def synthia_labels(index):
    return np.random.default_rng(2000 + index).integers(0, 23, IMAGE_HW).astype(np.uint8)
