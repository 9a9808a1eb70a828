"""Dataset readers, augmentation and the data contract against vectors made by
the reference's own readers (tests/golden/make_dataset_golden.py) over the same synthetic trees."""
import json
import os
import random

import numpy as np
import pytest

import dataset_fixtures as fx
from modular_semantic_segmentation_amd import datasets
from modular_semantic_segmentation_amd.datasets import augmentation as aug, imageops
from modular_semantic_segmentation_amd.datasets.cityscapes import Cityscapes, label_lookup_table
from modular_semantic_segmentation_amd.datasets.synthia_cityscapes import SynthiaCityscapes, remap_labels

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def golden():
    with open(os.path.join(GOLDEN, 'datasets.json')) as f:
        return json.load(f)


@pytest.fixture(scope='module')
def trees(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('data'))
    fx.build_cityscapes_tree(os.path.join(root, 'cityscapes'))
    fx.build_synthia_tree(os.path.join(root, 'synthia'))
    return root


def _names(items, key):
    return sorted(i[key] for i in items)


def test_cityscapes_splits_and_label_table_match_reference(trees, golden):
    data = Cityscapes(base_path=os.path.join(trees, 'cityscapes'))
    g = golden['cityscapes']
    assert label_lookup_table() == g['label_lookup']
    assert {str(k): v for k, v in data.labelinfo.items()} == g['labelinfo']
    for split in ('trainset', 'validation_set', 'measureset', 'testset'):
        assert _names(getattr(data, split), 'image_path') == g[split], split
    assert not any('atlantis' in p or 'elsewhere' in p for s in g for p in (g[s] if s.endswith('set') else []))


def test_synthia_cityscapes_splits_and_remap_match_reference(trees, golden):
    g = golden['synthia_cityscapes']
    data = SynthiaCityscapes(base_path=os.path.join(trees, 'synthia'))
    for split in ('trainset', 'validation_set', 'measureset', 'testset'):
        assert _names(getattr(data, split), 'image_name') == g[split], split
    ids = np.arange(23, dtype=np.uint8).reshape(1, 23)
    assert remap_labels(ids, False).ravel().tolist() == g['remap']['False']
    assert remap_labels(ids, True).ravel().tolist() == g['remap']['True']
    lanes = SynthiaCityscapes(base_path=os.path.join(trees, 'synthia'), labels={'lanemarkings': True})
    assert {str(k): v for k, v in lanes.labelinfo.items()} == g['labelinfo_lanes']


def test_cityscapes_sample_contract(trees):
    """rgb = raw B,G,R bytes, depth = raw uint16 values with a channel axis, labels remapped, all cropped to
    multiples of 16 and typed float32 / int32 (cityscapes.py:159-184, data_baseclass.py:64-80)."""
    data = Cityscapes(base_path=os.path.join(trees, 'cityscapes'))
    batch = data._get_batch([{'image_path': 'train/aachen/aachen_000002_000038'}])
    rgb, depth, ids = fx.cityscapes_sample(2)
    assert batch['rgb'].shape == (1, 32, 48, 3) and batch['rgb'].dtype == np.float32
    assert batch['depth'].shape == (1, 32, 48, 1) and batch['depth'].dtype == np.float32
    assert batch['labels'].shape == (1, 32, 48) and batch['labels'].dtype == np.int32
    assert np.array_equal(batch['rgb'][0], rgb[:32, :48, ::-1].astype(np.float32))
    assert np.array_equal(batch['depth'][0, :, :, 0], depth[:32, :48].astype(np.float32))
    assert np.array_equal(batch['labels'][0], np.asarray(label_lookup_table())[ids[:32, :48]])
    mask = data.get_ego_vehicle_mask('train/aachen/aachen_000002_000038')['labels']
    assert np.array_equal(mask, (ids == 1).astype(np.int32))
    resized = Cityscapes(base_path=os.path.join(trees, 'cityscapes'), resize=True)
    blob = resized._get_data('train/aachen/aachen_000002_000038')
    assert blob['rgb'].shape == (384, 768, 3) and blob['depth'].shape == (384, 768, 1)
    assert blob['labels'].shape == (384, 768) and set(np.unique(blob['labels'])) <= set(range(12))


def test_streams_and_batches(trees):
    data = SynthiaCityscapes(base_path=os.path.join(trees, 'synthia'))
    assert data.get_data_description() == ({'rgb': 'float32', 'depth': 'float32', 'labels': 'int32'},
                                           SynthiaCityscapes._data_shape_description, 12)
    assert SynthiaCityscapes.get_data_description(num_classes=13)[2] == 13
    stream = data.get_testset()
    samples = list(stream)
    assert len(samples) == len(stream) == len(data.testset) and len(list(stream)) == len(samples)   # re-iterable
    batch = data.get_testset(tf_dataset=False)
    assert batch['rgb'].shape == (len(samples), 32, 48, 3)
    for i, s in enumerate(samples):
        for m in ('rgb', 'depth', 'labels'):
            assert np.array_equal(s[m], batch[m][i])
    assert data.get_testset(num_items=2, tf_dataset=False)['labels'].shape[0] == 2
    assert len(data.get_validation_set(num_items=3)) == 3 and len(data.get_measureset()) == len(data.measureset)
    index = int(data.testset[0]['image_name'])
    assert np.array_equal(samples[0]['labels'], remap_labels(fx.synthia_labels(index))[:32, :48])
    in_mem = SynthiaCityscapes(base_path=os.path.join(trees, 'synthia'), in_memory=True)
    assert np.array_equal(in_mem.get_testset(tf_dataset=False)['depth'], batch['depth'])
    colours = data.coloured_labels(batch['labels'][0])
    assert colours.shape == (32, 48, 3) and colours.dtype == np.uint8
    assert np.array_equal(colours[batch['labels'][0] == 3][0], [128, 64, 128])
    # training format: augmented 16 x 16 crops, same keys and dtypes
    small = SynthiaCityscapes(base_path=os.path.join(trees, 'synthia'),
                              augmentation={'crop': [1, 16], 'scale': [.5, .7, 1.5], 'vflip': .3, 'hflip': False,
                                            'gamma': [.4, .3, 1.2], 'rotate': [.4, -13, 13], 'shear': [.3, .01, .1],
                                            'contrast': [.3, .5, 1.5], 'brightness': [.2, -40, 40]})
    random.seed(0)
    np.random.seed(0)
    for s in small.get_trainset():
        assert s['rgb'].shape == (16, 16, 3) and s['depth'].shape == (16, 16, 1) and s['labels'].shape == (16, 16)
        assert s['labels'].min() >= 0 and s['labels'].max() < 12 and 0 <= s['rgb'].min() and s['rgb'].max() <= 255


def test_registry():
    assert datasets.get_dataset('cityscapes') is Cityscapes and datasets.get_dataset('cityscapes_c') is Cityscapes
    assert datasets.get_dataset('synthia_cityscapes') is SynthiaCityscapes
    with pytest.raises(UserWarning):
        datasets.get_dataset('pascalvoc')
    with pytest.raises(IOError):
        Cityscapes(base_path='/nonexistent/cityscapes')


def test_augmentate_matches_reference_on_the_numpy_transforms():
    vec = np.load(os.path.join(GOLDEN, 'augmentation.npz'))
    blob0 = {m: vec['blob/' + m] for m in ('rgb', 'depth', 'labels')}
    cases = {'crop_flip_gamma': dict(crop=[1, 24], hflip=.95, vflip=.95, gamma=[1, 0.3, 1.2]),
             'labels': dict(label_flip=[3, 4], label_merge=[1, 2]),
             'nothing': dict(crop=[0, 24], gamma=[0, .3, 1.2])}
    for case_no, (name, kwargs) in enumerate(cases.items()):
        for seed in range(4):
            random.seed(10 * case_no + seed)
            np.random.seed(10 * case_no + seed)
            got = aug.augmentate({k: v.copy() for k, v in blob0.items()}, **kwargs)
            for m, v in got.items():
                assert np.array_equal(v, vec['{}/{}/{}'.format(name, seed, m)]), (name, seed, m)


def test_rotation_geometry_matches_reference():
    vec = np.load(os.path.join(GOLDEN, 'augmentation.npz'))
    for (w, h), row in zip(vec['rect_sizes'], vec['rect']):
        for d, want in zip(vec['rect_degs'], row):
            assert np.allclose(aug.inscribed_rect(w, h, np.deg2rad(d)), want, rtol=1e-12, atol=1e-9), (w, h, d)
    img = np.arange(30 * 44).reshape(30, 44)
    assert np.array_equal(aug._centre_crop(img, 17.6, 11.2), vec['centre_crop'])
    assert np.array_equal(aug._centre_crop(img, 100.0, 12.0), vec['centre_crop_big'])
    assert list(aug.crop_multiple(np.zeros((37, 50, 3))).shape) == vec['crop_multiple_shape'].tolist()
    assert aug.crop_multiple('not an image') == 'not an image'
    assert aug.crop_multiple(np.zeros((32, 48))).shape == (32, 48)


def test_resampling_conventions():
    """cv2 conventions restated in imageops: half-pixel-centre bilinear, floor-index nearest."""
    row = np.array([[10, 30]], dtype=np.uint8)
    assert imageops.resize_linear(row, 1, 4).tolist() == [[10, 15, 25, 30]]
    assert imageops.resize_nearest(row, 1, 4).tolist() == [[10, 10, 30, 30]]
    assert imageops.resize_nearest(np.arange(6).reshape(1, 6), 1, 3).tolist() == [[0, 2, 4]]
    img = np.random.default_rng(0).integers(0, 255, (12, 20, 3), dtype=np.uint8)
    assert np.array_equal(imageops.resize_linear(img, 12, 20), img)
    assert imageops.scale_image(img, 1.5, nearest=False).shape == (18, 30, 3)
    flat = np.full((9, 7), 1234, dtype=np.uint16)
    assert np.array_equal(imageops.resize_linear(flat, 20, 31), np.full((20, 31), 1234, dtype=np.uint16))
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(imageops.warp_affine(img, ident, 20, 12), img)
    shifted = imageops.warp_affine(img, np.array([[1.0, 0, 3], [0, 1.0, 2]]), 20, 12)
    assert np.array_equal(shifted[2:, 3:], img[:-2, :-3]) and not shifted[:2].any() and not shifted[:, :3].any()
    assert np.array_equal(aug._rotated_canvas(img, 0), img)
    quarter = aug._rotated_canvas(img[:, :, 0], 90)
    assert quarter.shape == (20, 12)
    # the centre is (w/2, h/2) in pixel-corner coordinates, as cv2.getRotationMatrix2D is called: one row off rot90
    assert np.array_equal(quarter[1:], np.rot90(img[:, :, 0])[:-1])


def _write_png(filename, array, filter_type=0):
    """Minimal PNG writer for [H,W,C] uint8 / uint16 arrays with one fixed row filter (0 none, 1 sub, 2 up)."""
    import struct
    import zlib
    h, w, c = array.shape
    depth = 16 if array.dtype == np.uint16 else 8
    rows = array.astype('>u2' if depth == 16 else np.uint8).reshape(h, -1).view(np.uint8).astype(np.int64)
    bpp = c * depth // 8
    prev = np.zeros_like(rows[0])
    lines = []
    for row in rows:
        if filter_type == 1:
            left = np.concatenate([np.zeros(bpp, dtype=np.int64), row[:-bpp]])
            enc = (row - left) & 255
        elif filter_type == 2:
            enc = (row - prev) & 255
        else:
            enc = row
        prev = row
        lines.append(bytes([filter_type]) + enc.astype(np.uint8).tobytes())

    def chunk(kind, body):
        return struct.pack('>I', len(body)) + kind + body + struct.pack('>I', zlib.crc32(kind + body) & 0xffffffff)

    colour = {1: 0, 3: 2, 4: 6}[c]
    with open(filename, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, colour, 0, 0, 0))
                + chunk(b'IDAT', zlib.compress(b''.join(lines))) + chunk(b'IEND', b''))


def test_png_decoder_and_synthia_preprocessing(tmp_path):
    """SYNTHIA label PNGs are 16-bit RGB with the class id in the first channel (synthia.py:215-228)."""
    from PIL import Image
    rng = np.random.default_rng(3)
    smooth = np.cumsum(rng.integers(-3, 4, (40, 50, 3)), axis=1).astype(np.uint8)
    Image.fromarray(smooth).save(str(tmp_path / 'adaptive.png'))          # PIL picks sub / up / average / paeth rows
    assert np.array_equal(imageops.png_channels(str(tmp_path / 'adaptive.png')), smooth)
    grey = rng.integers(0, 65536, (20, 30)).astype(np.uint16)
    Image.fromarray(grey).save(str(tmp_path / 'grey16.png'))
    assert np.array_equal(imageops.png_channels(str(tmp_path / 'grey16.png'))[:, :, 0], grey)
    assert np.array_equal(imageops.imread_anydepth(str(tmp_path / 'grey16.png')), grey)
    rgb16 = rng.integers(0, 65536, (12, 9, 3)).astype(np.uint16)
    for filter_type in (0, 1, 2):
        _write_png(str(tmp_path / 'rgb16.png'), rgb16, filter_type)
        assert np.array_equal(imageops.png_channels(str(tmp_path / 'rgb16.png')), rgb16), filter_type
    with pytest.raises(ValueError):
        (tmp_path / 'junk.png').write_bytes(b'not a png at all')
        imageops.png_channels(str(tmp_path / 'junk.png'))

    # a raw RAND_CITYSCAPES tree: label PNGs only, no LABELS_NPY, no split file
    root = str(tmp_path / 'synthia')
    fx.build_synthia_tree(root)
    base = os.path.join(root, 'RAND_CITYSCAPES')
    os.remove(os.path.join(base, 'train_test_split.json'))
    npy_dir = os.path.join(base, 'GT/LABELS_NPY/Stereo_Right/Omni_F')
    png_dir = os.path.join(base, 'GT/LABELS/Stereo_Right/Omni_F')
    os.makedirs(png_dir)
    want = {}
    for name in sorted(os.listdir(npy_dir)):
        labels = np.load(os.path.join(npy_dir, name))
        want[name] = labels
        planes = np.stack([labels.astype(np.uint16), rng.integers(0, 65536, labels.shape).astype(np.uint16),
                           rng.integers(0, 65536, labels.shape).astype(np.uint16)], axis=-1)
        _write_png(os.path.join(png_dir, name.replace('.npy', '.png')), planes, filter_type=1)
        os.remove(os.path.join(npy_dir, name))
    data = SynthiaCityscapes(base_path=root)
    for name, labels in want.items():
        assert np.array_equal(np.load(os.path.join(npy_dir, name)), labels)
    with open(os.path.join(base, 'train_test_split.json')) as f:
        split = json.load(f)
    assert len(split['testset']) == round(0.2 * len(want) + 0.499) and \
        sorted(split['trainset'] + split['testset']) == sorted(n[:-4] for n in want)
    assert len(data.trainset) + len(data.validation_set) == len(split['trainset'])
