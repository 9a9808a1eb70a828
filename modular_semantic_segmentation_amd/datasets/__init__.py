"""Dataset readers for the two corpora behind the reference's published numbers, and the registry
`get_dataset` (xview/datasets/__init__.py:9-23).  The data root is `$XVIEW_DATA_BASEPATH` (the
reference's `xview.settings.DATA_BASEPATH`)."""
import os

DATA_BASEPATH = os.environ.get('XVIEW_DATA_BASEPATH', '/tmp/xview_data')

from .data_baseclass import DataBaseclass, SampleStream  # noqa: E402
from .augmentation import augmentate, crop_multiple  # noqa: E402
from .cityscapes import Cityscapes  # noqa: E402
from .synthia_cityscapes import SynthiaCityscapes  # noqa: E402


def get_dataset(name):
    if name == 'synthia_cityscapes':
        return SynthiaCityscapes
    if name in ('cityscapes', 'cityscapes_c'):
        return Cityscapes
    raise UserWarning('ERROR: Dataset {} not found'.format(name))
