"""SimpleFCN model (reference: xview/models/simple_fcn.py:173-224) on the MI355X engine."""
import numpy as np
import torch

from . import fcn as _fcn
from . import ops
from .base_model import BaseModel
from .fcn import FcnEngine, init_variables  # noqa: F401  (fcn-level API re-exported for callers)


class SimpleFCN(BaseModel):
    """FCN expert.  Args as the reference: prefix, data_description, modality, output_dir,
    **config with required `num_units`, `batch_normalization`; optional learning_rate, trainer,
    batchsize, train_encoder, dropout_rate (the last two are unused by the reference too,
    simple_fcn.py:192-211)."""

    def __init__(self, prefix, data_description, modality, output_dir=None, **config):
        self.prefix = prefix
        self.modality = modality
        standard_config = {'train_encoder': True, 'dropout_rate': 0}
        standard_config.update(config)
        BaseModel.__init__(self, data_description, output_dir=output_dir, **standard_config)

    def _build_graph(self):
        shape = self.testdata_description[1][self.modality]
        self.in_channels = int(shape[-1])
        self.variables = init_variables(self.prefix, self.in_channels, self.config['num_units'],
                                        self.config['num_classes'],
                                        batch_normalization=self.config['batch_normalization'],
                                        seed=self.config.get('seed'))
        self.engine = FcnEngine(self.prefix, self.in_channels, self.config['num_units'],
                                self.config['num_classes'], self.variables, device=self.device)
        self.loss = None            # scalar of the last training step (set by _train_batch)
        self.prediction = 'label'   # name of the engine output that is the model's prediction

    def _variables_changed(self):
        self.engine.load(self.variables)
        if getattr(self, 'trainer', None) is not None:
            self.trainer.load_from_variables(variables=self.variables)

    # ---- training (base_model.py:153-162,180-261) ------------------------------------------------------
    def _ensure_trainer(self):
        if getattr(self, 'trainer', None) is None:
            if self.config['batch_normalization']:
                raise NotImplementedError('training with batch normalization is not built on this path')
            from .trainer import FcnTrainer
            from .parallel import GradReducer, world
            self.trainer = FcnTrainer(self.engine, self.config.get('trainer', 'adam'),
                                      self.config.get('learning_rate', 0.0001))
            self.trainer.load_from_variables(variables=self.variables)
            self._reducer = GradReducer(self.device) if world()[1] > 1 else None
        return self.trainer

    def _train_batch(self, batch):
        tr = self._ensure_trainer()
        x = self._to_device(batch[self.modality], torch.float32)
        labels = self._to_device(batch['labels'], torch.int32)
        self.loss = tr.step(x, labels, reducer=self._reducer)
        self._dirty = True
        return self.loss.item() if self.config.get('sync_loss', True) else 0.0

    def _sync_variables(self):
        if getattr(self, 'trainer', None) is not None and getattr(self, '_dirty', False):
            self.trainer.to_variables(self.variables)
            self._dirty = False

    def export_weights(self, save_dir=None):
        self._sync_variables()
        return BaseModel.export_weights(self, save_dir)

    def _predict_batch(self, batch, output_attr=None):
        x = self._to_device(batch[self.modality], torch.float32)
        want = 'label'
        if output_attr in ('prob', 'score'):
            want = output_attr
        return self.engine.forward(x, want=(want,))[want]
