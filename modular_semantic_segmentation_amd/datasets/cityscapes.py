"""Cityscapes RGB + disparity + fine labels, remapped to the 12 classes the experts are trained on.

Counterpart of the reference's `xview/datasets/cityscapes.py` (label remap :65-118, file lists
:120-157, `_load_data` :159-184, `_get_data` :186-202): same directory layout, same splits
(`val` of munster/frankfurt/lindau = test set, 5 % of `train` with seed 4 = measure set, 15 of the
rest = validation set), rgb as raw BGR bytes, disparity as raw uint16 values with a channel axis.
Directory listings are sorted here, which makes the seeded splits reproducible across file systems
(the reference splits whatever order `listdir` returns).
"""
import tarfile
from os import environ, listdir, path

import numpy as np
from sklearn.model_selection import train_test_split

from . import DATA_BASEPATH, imageops
from .augmentation import augmentate
from .data_baseclass import DataBaseclass

CITYSCAPES_BASEPATH = path.join(DATA_BASEPATH, 'cityscapes')

CITIES = ['aachen', 'bremen', 'darmstadt', 'erfurt', 'hanover', 'krefeld', 'strasbourg', 'tubingen',
          'weimar', 'bochum', 'cologne', 'dusseldorf', 'hamburg', 'jena', 'monchengladbach', 'stuttgart',
          'ulm', 'zurich']
TEST_CITIES = ['munster', 'frankfurt', 'lindau']

# class index -> (name, colour, Cityscapes labelIds folded into it); every other id is void
CLASSES = [
    ('void', [0, 0, 0], []),
    ('sky', [128, 128, 128], [23]),
    ('building', [128, 0, 0], [11, 12]),
    ('road', [128, 64, 128], [7, 9]),
    ('sidewalk', [0, 0, 192], [8]),
    ('fence', [64, 64, 128], [13]),
    ('vegetation', [128, 128, 0], [21, 22]),
    ('pole', [192, 192, 128], [17]),
    ('vehicle', [64, 0, 128], [26, 27, 28, 29, 30, 31, 32]),
    ('traffic sign', [192, 128, 128], [20]),
    ('person', [64, 64, 0], [24, 25]),
    ('bicycle', [0, 128, 192], [33]),
]
NUM_CITYSCAPES_IDS = 34

TRAIN_AUGMENTATION = {'crop': [1, 240], 'scale': [.4, 1, 1.5], 'vflip': .3, 'hflip': False,
                      'gamma': [.4, 0.3, 1.2], 'rotate': False, 'shear': False,
                      'contrast': [.3, 0.5, 1.5], 'brightness': [.2, -40, 40]}

MODALITY_DIRS = {'rgb': ('leftImg8bit_trainvaltest/leftImg8bit', 'leftImg8bit'),
                 'labels': ('gtFine_trainvaltest/gtFine', 'gtFine_labelIds'),
                 'depth': ('disparity_trainvaltest/disparity', 'disparity')}


def label_lookup_table():
    """labelId -> class index, 34 entries (cityscapes.py:65-118)."""
    table = [0] * NUM_CITYSCAPES_IDS
    for index, (_, _, ids) in enumerate(CLASSES):
        for i in ids:
            table[i] = index
    return table


class Cityscapes(DataBaseclass):

    _data_shape_description = {'rgb': (None, None, 3), 'depth': (None, None, 1), 'labels': (None, None)}
    _num_default_classes = 12

    def __init__(self, base_path=CITYSCAPES_BASEPATH, batchsize=1, in_memory=False, cities=CITIES,
                 **data_config):
        self.config = {'augmentation': dict(TRAIN_AUGMENTATION), 'resize': False}
        self.config.update(data_config)
        if not path.exists(base_path):
            message = 'ERROR: Path to CITYSCAPES dataset does not exist.'
            print(message)
            raise IOError(1, message, base_path)
        self.base_path = base_path
        self.in_memory = in_memory
        self.images = {}
        self.label_lookup = label_lookup_table()

        if in_memory and 'TMPDIR' in environ and path.exists(path.join(base_path, 'cityscapes.tar.gz')):
            # unpack next to the process, then cache decoded images as they are first read
            with tarfile.open(path.join(base_path, 'cityscapes.tar.gz')) as tar:
                tar.extractall(path=environ['TMPDIR'])
            self.base_path = environ['TMPDIR']

        trainset = self._list_images('train', cities)
        testset = self._list_images('val', TEST_CITIES)
        trainset, measureset = train_test_split(trainset, test_size=0.05, random_state=4)
        labelinfo = {i: {'name': name, 'color': colour} for i, (name, colour, _) in enumerate(CLASSES)}
        DataBaseclass.__init__(self, trainset, measureset, testset, labelinfo)

    def _list_images(self, fileset, cities):
        root = path.join(self.base_path, MODALITY_DIRS['rgb'][0], fileset)
        items = []
        for city in sorted(listdir(root)):
            if cities and city not in cities:
                continue
            for name in sorted(listdir(path.join(root, city))):
                stem = '_'.join(path.splitext(name)[0].split('_')[:3])      # <city>_<seq>_<frame>
                items.append({'image_path': path.join(fileset, city, stem)})
        return items

    def _filename(self, image_path, modality):
        folder, suffix = MODALITY_DIRS[modality]
        return path.join(self.base_path, folder, '{}_{}.png'.format(image_path, suffix))

    def _load_data(self, image_path):
        blob = {'rgb': imageops.imread_bgr(self._filename(image_path, 'rgb')),
                'depth': imageops.imread_anydepth(self._filename(image_path, 'depth'))}
        ids = imageops.imread_anydepth(self._filename(image_path, 'labels'))
        blob['labels'] = np.asarray(self.label_lookup, dtype='int32')[ids]
        if self.config['resize']:
            blob['rgb'] = imageops.resize_linear(blob['rgb'], 384, 768)
            for m in ('depth', 'labels'):
                blob[m] = imageops.resize_nearest(blob[m], 384, 768)
        blob['depth'] = blob['depth'][:, :, None]
        return blob

    def _get_data(self, image_path, training_format=False):
        if self.in_memory:
            if image_path not in self.images:
                self.images[image_path] = self._load_data(image_path)
            blob = {m: v.copy() for m, v in self.images[image_path].items()}
        else:
            blob = self._load_data(image_path)
        if training_format:
            blob = augmentate(blob, **self.config['augmentation'])
        return blob

    def get_ego_vehicle_mask(self, image_path):
        """Blob whose labels are 1 on the ego vehicle (labelId 1) and 0 elsewhere (cityscapes.py:204-215)."""
        saved = self.label_lookup
        self.label_lookup = [int(i == 1) for i in range(NUM_CITYSCAPES_IDS)]
        try:
            return self._load_data(image_path)
        finally:
            self.label_lookup = saved
