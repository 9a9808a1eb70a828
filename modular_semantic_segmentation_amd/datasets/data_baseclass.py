"""Dataset contract the models consume: item lists split into train / validation / measure / test
sets, served either stacked as numpy batches or as re-iterable streams of per-image dicts.

Counterpart of the reference's `xview/datasets/data_baseclass.py:10-126`.  Where the reference wraps a
generator in `tf.data.Dataset.from_generator`, `SampleStream` is the plain-Python equivalent: the
models batch it themselves (`base_model.iterate_batches`), exactly as the reference's `BaseModel`
calls `.batch(batchsize)` on the dataset it is handed (`base_model.py:22-32,203-206`).
"""
from random import shuffle

import numpy as np
from sklearn.model_selection import train_test_split

from .augmentation import crop_multiple


class SampleStream:
    """Re-iterable stream of `{modality: array}` samples without a batch axis; every pass re-reads
    (and, in training format, re-augments) the items, like re-running the reference's generator."""

    def __init__(self, items, loader):
        self._items = list(items)
        self._loader = loader

    def __len__(self):
        return len(self._items)

    def __iter__(self):
        for item in self._items:
            yield self._loader(item)

    def take(self, count):
        return SampleStream(self._items[:count], self._loader)


class DataBaseclass:
    """Splits item lists and serves them; subclasses define `_data_shape_description`,
    `_num_default_classes` and `_get_data(training_format=..., **item)` (data_baseclass.py:14-61)."""

    def __init__(self, trainset, measureset, testset, labelinfo, validation_set=None, num_classes=None,
                 info=False):
        if validation_set is None:
            # 15 validation images with the reference's fixed seed (data_baseclass.py:17-18)
            self.trainset, self.validation_set = train_test_split(trainset, test_size=15,
                                                                  random_state=317243896)
        else:
            self.trainset, self.validation_set = trainset, validation_set
        self.measureset = measureset
        self.testset = testset
        self.num_classes = self._num_default_classes if num_classes is None else num_classes
        self.modalities = list(self._data_shape_description.keys())
        self.labelinfo = labelinfo
        self.print_info = info
        shuffle(self.trainset)

    @classmethod
    def get_data_description(cls, num_classes=None):
        """(dtypes, shapes, number of classes) per modality, what every model constructor takes
        (data_baseclass.py:33-55); dtypes are numpy names instead of tf dtypes."""
        shapes = cls._data_shape_description
        if num_classes is None:
            num_classes = cls._num_default_classes
        dtypes = {m: 'int32' if m == 'labels' else 'float32' for m in shapes}
        return dtypes, shapes, num_classes

    def _get_data(self, **kwargs):
        raise NotImplementedError

    def _load_sample(self, item, training_format):
        if self.print_info:
            print(item)
        data = self._get_data(training_format=training_format, **item)
        return {m: np.asarray(crop_multiple(data[m])).astype('int32' if m == 'labels' else 'float32')
                for m in self.modalities}

    def _get_batch(self, items, training_format=False):
        """All `items` stacked along a leading batch axis (data_baseclass.py:64-80)."""
        samples = [self._load_sample(item, training_format) for item in items]
        return {m: np.stack([s[m] for s in samples]) for m in self.modalities}

    def _get_stream(self, items, training_format=False):
        return SampleStream(items, lambda item: self._load_sample(item, training_format))

    def _serve(self, items, as_stream, training_format=False):
        if as_stream:
            return self._get_stream(items, training_format=training_format)
        return self._get_batch(items, training_format=training_format)

    # `tf_dataset` keeps the reference's keyword: True = lazily loaded stream, False = numpy batch.
    def get_trainset(self, tf_dataset=True, training_format=True):
        return self._serve(self.trainset, tf_dataset, training_format=training_format)

    def get_testset(self, num_items=None, tf_dataset=True):
        return self._serve(self.testset[:num_items], tf_dataset)

    def get_measureset(self, tf_dataset=True):
        return self._serve(self.measureset, tf_dataset)

    def get_validation_set(self, num_items=None, tf_dataset=True):
        return self._serve(self.validation_set[:num_items], tf_dataset)

    def coloured_labels(self, labels):
        """[...,3] uint8 picture of a label map with the dataset's colours (data_baseclass.py:120-126)."""
        table = np.array([self.labelinfo[i]['color'] for i in range(max(self.labelinfo) + 1)]).astype(int)
        return table[np.asarray(labels)].astype('uint8')
