"""SimpleFCN model (reference: xview/models/simple_fcn.py:173-224) on the MI355X engine."""
import torch

from .base_model import BaseModel
from .fcn import FcnEngine, init_variables  # noqa: F401  (fcn-level API re-exported for callers)


_ENGINES = {}


def _engine_for(prefix, inputs, num_units, num_classes, variables):
    """Functional entry points cache one engine per (prefix, shape, weights object), like
    tf.variable_scope(prefix, reuse=tf.AUTO_REUSE) shares variables between calls (simple_fcn.py:11)."""
    key = (prefix, int(inputs.shape[-1]), int(num_units), int(num_classes), id(variables))
    ent = _ENGINES.get(key)
    # (the entry keeps the caller's dict alive and checks identity: an id in a key must not be handed to another dict while
    # the entry exists -- a freed dict's id is reused, and the cached engine would answer with the old weights)
    if ent is None or ent[0] is not variables:
        given = variables
        if variables is None:
            variables = init_variables(prefix, int(inputs.shape[-1]), num_units, num_classes)
        ent = (given, FcnEngine(prefix, int(inputs.shape[-1]), num_units, num_classes, variables, device=inputs.device))
        _ENGINES[key] = ent
    return ent[1]


def encoder(inputs, prefix, num_units, dropout_rate=0.0, variables=None, num_classes=2, dropout_layers=(),
            dropout_seed=0, **unused):
    """Functional form of simple_fcn.py:10-87: dict of layer outputs (ops.Act, bf16 padded NHWC), the
    encoding has key 'fused'.  inputs: float32 CUDA tensor [N,H,W,C_in]; variables: reference-schema dict
    (default: fresh [TF1] initialisers).  dropout_rate / dropout_layers ('pool3', 'conv4_3', 'conv5_3'): the
    reference's MC-dropout sites (always training=True; 'pool3' also drops after pool4, 'pool4' alone does nothing,
    simple_fcn.py:51-63), adding 'pool3_drop' / 'pool4_drop' to the dict; every call draws new masks.  trainable /
    batchnorm / is_training / reuse are accepted and ignored (inference graph, BN folded at load time)."""
    eng = _engine_for(prefix, inputs, num_units, num_classes, variables)
    eng.set_dropout([d for d in dropout_layers if d != 'features'], dropout_rate, dropout_seed)
    try:
        return eng.encoder(inputs, keep_all=True)
    finally:
        eng.set_dropout((), 0.0)


def decoder(features, prefix, num_units, num_classes, trainable=True, is_training=False, reuse=None, dropout_rate=None,
            batchnorm=True, variables=None, dropout_seed=0):
    """simple_fcn.py:90-134 under its own name: {'upscore': x8 bilinear deconv [+ batch norm] + relu of `features` (ops.Act,
    N x 8h x 8w x num_units), 'score': its 1x1 conv onto num_classes [+ batch norm], dense float32 [N,8h,8w,C]} -- the
    un-commuted form, layer by layer through custom_layers.deconv2d / conv2d (the engines' own decoder commutes the score
    conv in front of the deconv and never writes `upscore`; this is the entry point fusion_fcn.py:4-7 imports, for callers
    that want the layer dict).  features: ops.Act or dense float32 [N,h,w,num_units]; variables: reference-schema dict with
    `<prefix>/upscore/...` (kernel optional: the bilinear constant) and `<prefix>/score/...`; dropout_rate: dropout on the
    input features (always training=True, simple_fcn.py:124-126)."""
    from . import custom_layers as cl, ops
    x = cl._as_act(features)
    if dropout_rate:
        x = ops.dropout(x, float(dropout_rate), int(dropout_seed))
    up = cl.deconv2d(x, num_units, [16, 16], strides=[8, 8], name='upscore', activation='relu', padding='same',
                     batch_normalization=batchnorm, training=is_training, trainable=False, reuse=reuse,
                     variables=variables, scope=prefix)
    score = cl.conv2d(up, num_classes, [1, 1], name='score', activation=None, padding='same',
                      batch_normalization=batchnorm, training=is_training, trainable=trainable, reuse=reuse,
                      variables=variables, scope=prefix)
    return {'upscore': up, 'score': score}


def fcn(inputs, prefix, num_units, num_classes, variables=None, dropout_rate=0.0, dropout_layers=(), dropout_seed=0,
        **unused):
    """Functional form of simple_fcn.py:137-170 as experiments/timing.py uses it: all encoder layers plus
    'score' (float32 [N,H,W,C]); 'prob' and 'classification' of test_pipeline come for free from the same
    fused decoder kernel.  dropout_layers may name the encoder sites and 'features' (the decoder's input dropout,
    simple_fcn.py:124-126), as bayesian_fcn.py:74-89 passes them."""
    eng = _engine_for(prefix, inputs, num_units, num_classes, variables)
    eng.set_dropout(dropout_layers, dropout_rate, dropout_seed)
    try:
        out = eng.forward(inputs, want=('score', 'prob', 'label'), keep_all=True)
    finally:
        eng.set_dropout((), 0.0)
    layers = dict(out['layers'])
    layers.update(score=out['score'], prob=out['prob'], classification=out['label'])
    return layers


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
        engine_cls = FcnEngine
        if self.config.get('conv_dtype', 'bf16') == 'fp32':
            from .fcn_exact import FcnEngineF32 as engine_cls       # the plain-float32 parity mode (inference only)
        self.engine = engine_cls(self.prefix, self.in_channels, self.config['num_units'],
                                self.config['num_classes'], self.variables, device=self.device,
                                conv_dtype=self.config.get('conv_dtype', 'bf16'),
                                streamk=self.config.get('streamk', False), **({'fp8_deep': True} if self.config.get('fp8_deep') else {}),
                                **({'fp8_start': self.config['fp8_start']} if self.config.get('fp8_start') else {}))
        self.loss = None            # scalar of the last training step (set by _train_batch)
        self.prediction = 'label'   # name of the engine output that is the model's prediction

    def _variables_changed(self):
        BaseModel._variables_changed(self)
        self.engine.load(self.variables)
        if getattr(self, 'trainer', None) is not None:
            from .parallel import sync_trainer_from_rank0
            self.trainer.load_from_variables(variables=self.variables)
            self._after_rank0_sync(sync_trainer_from_rank0(self.trainer))

    # ---- training (base_model.py:153-162,180-261) ------------------------------------------------------
    def calibrate(self, data):
        """conv_dtype='fp8': fix the activation scales from the first batch of `data` (FcnEngine.calibrate) and -- unless the
        config names a plan (`fp8_start`, `fp8_deep`) -- choose the e4m3 plan by its label agreement with the bf16 graph on
        that batch (FcnEngine.calibrate_guarded; config `fp8_agreement`, default 0.995, 0 = off)."""
        from .base_model import iterate_batches
        from .basic_fusion_model import fp8_guard_bound
        batch = next(iterate_batches(data, self.config['batchsize']))
        self._graph = None
        x = self._to_device(batch[self.modality], torch.float32)
        bound = fp8_guard_bound(self.config)
        return self.engine.calibrate_guarded(x, bound) if bound is not None else self.engine.calibrate(x)

    def _ensure_trainer(self):
        if self.engine.conv_dtype != 'bf16':
            raise UserWarning("ERROR: conv_dtype='%s' is an inference configuration; train with conv_dtype='bf16'"
                              % self.engine.conv_dtype)
        if getattr(self, 'trainer', None) is None:
            from .trainer import FcnBnTrainer, FcnTrainer
            from .parallel import GradReducer, require_equal_batchsize, sync_trainer_from_rank0, world
            cls = FcnBnTrainer if self.config['batch_normalization'] else FcnTrainer
            self.trainer = cls(self.engine, self.config.get('trainer', 'adam'), self.config.get('learning_rate', 0.0001))
            self.trainer.load_from_variables(variables=self.variables)
            # every replica starts from rank 0's initialisers
            self._after_rank0_sync(sync_trainer_from_rank0(self.trainer))
            require_equal_batchsize(self.config['batchsize'], self.device)
            self._reducer = GradReducer(self.device) if world()[1] > 1 else None
        return self.trainer

    def _after_rank0_sync(self, synced):
        """Rank 0's parameters have just replaced this rank's own draw inside the trainer: mark the variable dict (and
        the batch-norm-folded inference engine) stale, so that predict / score / export_weights on every rank see them."""
        if synced:
            self._dirty = True

    def _train_batch(self, batch):
        tr = self._ensure_trainer()
        x = self._to_device(batch[self.modality], torch.float32)
        labels = self._to_device(batch['labels'], torch.int32)
        self._graph = None          # a captured inference graph holds the pre-update weight pointers
        if self._reducer is not None and len(labels) != self.config['batchsize']:
            raise ValueError('data-parallel step with %d images, batchsize is %d' % (len(labels), self.config['batchsize']))
        self.loss = tr.step(x, labels, reducer=self._reducer)
        self._dirty = True
        return self.loss.item() if self.config.get('sync_loss', True) else 0.0

    def _sync_variables(self):
        if getattr(self, 'trainer', None) is not None and getattr(self, '_dirty', False):
            self.trainer.to_variables(self.variables)
            self._dirty = False
            if self.config['batch_normalization']:
                # the inference engine folds the (moving) batch-norm statistics into its weights: rebuild them
                self.engine.load(self.variables)

    def export_weights(self, save_dir=None):
        self._sync_variables()
        return BaseModel.export_weights(self, save_dir)

    def _predict_batch_impl(self, batch, output_attr=None):
        if self.config['batch_normalization']:
            self._sync_variables()
        x = self._to_device(batch[self.modality], torch.float32)
        want = 'label'
        if output_attr in ('prob', 'score'):
            want = output_attr
        return self.engine.forward(x, want=(want,))[want]
