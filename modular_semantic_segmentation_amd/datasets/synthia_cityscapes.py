"""SYNTHIA-RAND-CITYSCAPES RGB + depth + labels in the 12-class (optionally 13 with lane markings)
scheme of the experts.

Counterpart of the reference's `xview/datasets/synthia_cityscapes.py` (`__init__` :40-106,
`_load_data` :143-181, `_get_data` :183-222): `<base>/RAND_CITYSCAPES/{RGB,Depth,GT/LABELS_NPY}/
Stereo_Right/Omni_F/<name>.{png,png,npy}`, the train/test item lists of `train_test_split.json`, the
test list halved (seed 1) into measure and test sets.  `GT/LABELS_NPY` holds the first channel of the
16-bit label PNGs, made once by `preprocess()` (the reference's `_preprocessing`, :112-141, whose paths
do not agree with its own `_load_data`; the layout `_load_data` reads is the one written here).  In
training format the reference's reader emits one-hot labels, which its own
`BaseModel.fit` would one-hot a second time; labels stay integer maps here, the contract every model
(and the reference's Cityscapes reader) uses.
"""
import json
import tarfile
from copy import deepcopy
from os import environ, listdir, makedirs, path

import numpy as np
from sklearn.model_selection import train_test_split

from . import DATA_BASEPATH, imageops
from .augmentation import augmentate
from .data_baseclass import DataBaseclass

SYNTHIA_BASEPATH = path.join(DATA_BASEPATH, 'synthia')

LABELINFO = {i: {'name': name, 'color': colour} for i, (name, colour) in enumerate([
    ('void', [0, 0, 0]), ('sky', [128, 128, 128]), ('building', [128, 0, 0]), ('road', [128, 64, 128]),
    ('sidewalk', [0, 0, 192]), ('fence', [64, 64, 128]), ('vegetation', [128, 128, 0]),
    ('pole', [192, 192, 128]), ('car', [64, 0, 128]), ('traffic sign', [192, 128, 128]),
    ('pedestrian', [64, 64, 0]), ('bicycle', [0, 128, 192])])}
LANEMARKING = 12

# SYNTHIA ids 12..22 -> scheme above, as in the AdapNet paper (synthia_cityscapes.py:156-167):
# motorcycle, rider -> bicycle; truck, bus -> car; parking spot, lane marking -> lane marking;
# road work, traffic light, terrain, train, wall -> void
SYNTHIA_REMAP = {12: 11, 13: LANEMARKING, 14: 0, 15: 0, 16: 0, 17: 11, 18: 8, 19: 8, 20: 0, 21: 0,
                 22: LANEMARKING}

TRAIN_AUGMENTATION = {'crop': [1, 240], 'scale': [.4, 0.7, 1.5], 'vflip': .3, 'hflip': False,
                      'gamma': [.4, 0.3, 1.2], 'rotate': [.4, -13, 13], 'shear': [0, 0.01, 0.03],
                      'contrast': [.3, 0.5, 1.5], 'brightness': [.2, -40, 40]}


def remap_labels(labels, lanemarkings=False):
    """SYNTHIA class ids -> expert class ids through one lookup table."""
    table = np.arange(max(int(labels.max()) + 1, 23), dtype=labels.dtype)
    for src, dst in SYNTHIA_REMAP.items():
        table[src] = dst
    if not lanemarkings:
        table[table == LANEMARKING] = 0
    return table[labels]


class SynthiaCityscapes(DataBaseclass):
    """Driver for the SYNTHIA-RAND-CITYSCAPES set (http://synthia-dataset.net/)."""

    _data_shape_description = {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}
    _num_default_classes = 12

    def __init__(self, base_path=SYNTHIA_BASEPATH, force_preprocessing=False, batchsize=1, resize=False,
                 in_memory=False, **data_config):
        self.config = {'augmentation': dict(TRAIN_AUGMENTATION), 'labels': {'lanemarkings': False}}
        self.config.update(data_config)
        self.config['resize'] = resize
        if not path.exists(base_path):
            message = 'ERROR: Path to SYNTHIA dataset does not exist.'
            print(message)
            raise IOError(1, message, base_path)
        self.basepath = path.join(base_path, 'RAND_CITYSCAPES')

        if in_memory and 'TMPDIR' in environ and path.exists(path.join(base_path, 'RAND_CITYSCAPES.tar.gz')):
            with tarfile.open(path.join(base_path, 'RAND_CITYSCAPES.tar.gz')) as tar:
                tar.extractall(path=environ['TMPDIR'])
            self.basepath = environ['TMPDIR']
        if force_preprocessing or not path.exists(path.join(self.basepath, 'train_test_split.json')):
            self.preprocess(force=force_preprocessing)
        with open(path.join(self.basepath, 'train_test_split.json')) as f:
            split = json.load(f)
        if in_memory:
            trainset = [{'image': self._load_data(name)} for name in split['trainset']]
            testset = [{'image': self._load_data(name)} for name in split['testset']]
        else:
            trainset = [{'image_name': name} for name in split['trainset']]
            testset = [{'image_name': name} for name in split['testset']]
        measureset, testset = train_test_split(testset, test_size=0.5, random_state=1)

        labelinfo = deepcopy(LABELINFO)
        if self.config['labels']['lanemarkings']:
            labelinfo[LANEMARKING] = {'name': 'lanemarking', 'color': [0, 192, 0]}
        DataBaseclass.__init__(self, trainset, measureset, testset, labelinfo)

    def preprocess(self, force=False):
        """One-off: class-id channel of `GT/LABELS/.../*.png` -> `GT/LABELS_NPY/.../*.npy`, and an 80/20
        `train_test_split.json` if there is none yet."""
        source = path.join(self.basepath, 'GT/LABELS/Stereo_Right/Omni_F')
        target = path.join(self.basepath, 'GT/LABELS_NPY/Stereo_Right/Omni_F')
        makedirs(target, exist_ok=True)
        names = sorted(path.splitext(n)[0] for n in listdir(source))
        for name in names:
            if force or not path.exists(path.join(target, name + '.npy')):
                np.save(path.join(target, name), imageops.one_channel_image_reader(
                    path.join(source, name + '.png'), np.uint8))
        split_file = path.join(self.basepath, 'train_test_split.json')
        if not path.exists(split_file):
            trainset, testset = train_test_split(names, test_size=0.2)
            with open(split_file, 'w') as f:
                json.dump({'trainset': trainset, 'testset': testset}, f)

    def _load_data(self, image_name):
        def filename(folder, extension):
            return path.join(self.basepath, folder, 'Stereo_Right/Omni_F', '{}.{}'.format(image_name, extension))

        blob = {'rgb': imageops.imread_bgr(filename('RGB', 'png')),
                'depth': imageops.imread_anydepth(filename('Depth', 'png')),
                'labels': remap_labels(np.load(filename('GT/LABELS_NPY', 'npy')),
                                       self.config['labels']['lanemarkings'])}
        if self.config['resize']:
            blob['rgb'] = imageops.resize_linear(blob['rgb'], 384, 768)
            for m in ('depth', 'labels'):
                blob[m] = imageops.resize_nearest(blob[m], 384, 768)
        return blob

    def _get_data(self, image_name=False, image=False, training_format=True):
        assert image_name or image, 'an item names a file or carries a decoded image'
        blob = self._load_data(image_name) if image_name else {m: v.copy() for m, v in image.items()}
        if training_format:
            blob = augmentate(blob, **{k: self.config['augmentation'][k] for k in (
                'scale', 'crop', 'hflip', 'vflip', 'gamma', 'contrast', 'brightness', 'rotate', 'shear')})
        blob['depth'] = blob['depth'][:, :, None]
        return blob
