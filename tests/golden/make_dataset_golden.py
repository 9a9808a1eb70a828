#!/usr/bin/env python3
"""Generate tests/golden/datasets.json + augmentation.npz from the reference's dataset readers.

TEST INFRASTRUCTURE, build container only (needs /root/reference).  The reference's readers are
imported in place with the absent third-party modules (tensorflow, cv2, imgaug, png, scipy.misc)
stubbed, and run over the synthetic directory trees of tests/dataset_fixtures.py.  Only their outputs
are stored:
  datasets.json      Cityscapes (xview/datasets/cityscapes.py:65-157): label lookup table and the
                     train / validation / measure / test item lists; SynthiaCityscapes
                     (synthia_cityscapes.py:91-167): the item lists and the remap of ids 0..22 with and
                     without lane markings.  `os.listdir` is wrapped to return sorted names while the
                     reference builds its lists, the order the build's readers define.
  augmentation.npz   augmentation.py: `augmentate` on a seeded blob for the numpy-only transforms
                     (crop, flips, gamma, label merge / flip), `largest_rotated_rect`,
                     `crop_around_center`, `crop_multiple`.
Image decoding (cv2.imread) is stubbed out, so pixel values are not part of these vectors.

Usage:  python tests/golden/make_dataset_golden.py     (from the repo root)
"""
import importlib
import json
import os
import random
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True
import dataset_fixtures as fx  # noqa: E402


class _Any(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        m = _Any(self.__name__ + '.' + k)
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return _Any('call')


def _import_reference(data_root):
    for n in ['tensorflow', 'cv2', 'imgaug', 'png', 'scipy.misc', 'tqdm']:
        sys.modules[n] = _Any(n)
    sys.modules['tqdm'].tqdm = lambda x, *a, **k: x
    settings = types.ModuleType('xview.settings')
    settings.DATA_BASEPATH = data_root
    sys.modules['xview.settings'] = settings
    for pkg, p in [('xview', '/root/reference/xview'), ('xview.datasets', '/root/reference/xview/datasets')]:
        m = types.ModuleType(pkg)
        m.__path__ = [p]
        sys.modules[pkg] = m
    return (importlib.import_module('xview.datasets.cityscapes'),
            importlib.import_module('xview.datasets.synthia_cityscapes'),
            importlib.import_module('xview.datasets.augmentation'))


def _paths(items, key):
    return sorted(i[key] for i in items)


def main():
    out = {}
    with tempfile.TemporaryDirectory() as root:
        fx.build_cityscapes_tree(os.path.join(root, 'cityscapes'))
        fx.build_synthia_tree(os.path.join(root, 'synthia'))
        cs, sc, aug = _import_reference(root)

        real_listdir = os.listdir
        cs.listdir = lambda p: sorted(real_listdir(p))
        data = cs.Cityscapes(base_path=os.path.join(root, 'cityscapes'))
        out['cityscapes'] = {'label_lookup': [int(i) for i in data.label_lookup],
                             'labelinfo': {str(k): v for k, v in data.labelinfo.items()},
                             'trainset': _paths(data.trainset, 'image_path'),
                             'validation_set': _paths(data.validation_set, 'image_path'),
                             'measureset': _paths(data.measureset, 'image_path'),
                             'testset': _paths(data.testset, 'image_path')}

        ids = np.arange(23, dtype=np.uint8).reshape(1, 23)
        remaps = {}
        for lanes in (False, True):
            data = sc.SynthiaCityscapes(base_path=os.path.join(root, 'synthia'), labels={'lanemarkings': lanes})
            label_file = os.path.join(root, 'synthia/RAND_CITYSCAPES/GT/LABELS_NPY/Stereo_Right/Omni_F/0000000.npy')
            np.save(label_file, ids)
            remaps[str(lanes)] = data._load_data('0000000')['labels'].ravel().astype(int).tolist()
            if not lanes:
                out['synthia_cityscapes'] = {'trainset': _paths(data.trainset, 'image_name'),
                                             'validation_set': _paths(data.validation_set, 'image_name'),
                                             'measureset': _paths(data.measureset, 'image_name'),
                                             'testset': _paths(data.testset, 'image_name')}
            else:
                out['synthia_cityscapes']['labelinfo_lanes'] = {str(k): v for k, v in data.labelinfo.items()}
        out['synthia_cityscapes']['remap'] = remaps

    with open(os.path.join(HERE, 'datasets.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)

    # ---- augmentation: numpy-only transforms on a seeded blob ------------------------------------
    vec = {}
    rng = np.random.default_rng(5)
    blob0 = {'rgb': rng.integers(0, 256, (40, 56, 3), dtype=np.uint8),
             'depth': rng.integers(0, 30000, (40, 56, 1)).astype(np.uint16),
             'labels': rng.integers(0, 12, (40, 56)).astype(np.int32)}
    cases = {'crop_flip_gamma': dict(crop=[1, 24], hflip=.95, vflip=.95, gamma=[1, 0.3, 1.2]),
             'labels': dict(label_flip=[3, 4], label_merge=[1, 2]),
             'nothing': dict(crop=[0, 24], gamma=[0, .3, 1.2])}
    for case_no, (name, kwargs) in enumerate(cases.items()):
        for seed in range(4):
            random.seed(10 * case_no + seed)
            np.random.seed(10 * case_no + seed)
            got = aug.augmentate({k: v.copy() for k, v in blob0.items()}, **kwargs)
            for m, v in got.items():
                vec['{}/{}/{}'.format(name, seed, m)] = np.ascontiguousarray(v)
    for m, v in blob0.items():
        vec['blob/' + m] = v
    sizes = [(64, 48), (48, 64), (100, 100), (1280, 760)]
    degs = [-13, -7, -1, 0, 3, 12, 45, 100]
    vec['rect_sizes'] = np.array(sizes)
    vec['rect_degs'] = np.array(degs)
    vec['rect'] = np.array([[aug.largest_rotated_rect(w, h, np.deg2rad(d)) for d in degs] for w, h in sizes])
    img = np.arange(30 * 44).reshape(30, 44)
    vec['centre_crop'] = aug.crop_around_center(img, 17.6, 11.2)
    vec['centre_crop_big'] = aug.crop_around_center(img, 100.0, 12.0)
    vec['crop_multiple_shape'] = np.array(aug.crop_multiple(np.zeros((37, 50, 3))).shape)
    np.savez_compressed(os.path.join(HERE, 'augmentation.npz'), **vec)
    print('wrote datasets.json, augmentation.npz')


if __name__ == '__main__':
    main()
