"""Average fusion (reference: xview/models/average_mix.py)."""
from . import ops
from .basic_fusion_model import FusionModel


class AverageFusion(FusionModel):
    expert_wants = ('prob',)

    def __init__(self, output_dir=None, **config):
        FusionModel.__init__(self, name='AverageFusion', output_dir=output_dir, **config)

    def _fusion(self, expert_outputs, output_attr=None):
        return ops.average_fuse([expert_outputs[m]['prob'] for m in self.modalities])
