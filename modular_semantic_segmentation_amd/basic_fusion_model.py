"""FusionModel base + test_pipeline (reference: xview/models/basic_fusion_model.py)."""
import torch

from .base_model import BaseModel
from .fcn import FcnEngine, init_variables


def test_pipeline(engine, inputs, want=('prob', 'classification'), st=None):
    """Expert forward + softmax + argmax (basic_fusion_model.py:9-23) on an FcnEngine:
    returns {'prob': f32 [N,H,W,C], 'classification': i64 [N,H,W]} (only what `want` names).  st: an encoder state to
    finish (the shared launches of two experts, run_trunks) instead of `inputs`."""
    want = tuple('label' if w == 'classification' else w for w in want)
    return engine.forward(inputs, want=want, st=st) if st is not None else engine.forward(inputs, want=want)


def expert_factory(expert_model, conv_dtype='bf16'):
    """(engine class, initialiser) of an `expert_model` name, the choice test_pipeline makes
    (basic_fusion_model.py:13-20); conv_dtype='fp32': the plain-float32 FCN engine (fcn_exact, the parity mode)."""
    if expert_model == 'fcn':
        if conv_dtype == 'fp32':
            from .fcn_exact import FcnEngineF32
            return FcnEngineF32, init_variables
        return FcnEngine, init_variables
    if expert_model == 'adapnet':
        from .adapnet import AdapnetEngine, init_variables as init_adapnet
        return AdapnetEngine, init_adapnet
    raise UserWarning('ERROR: Expert Model %s not found' % expert_model)


def engine_options(config):
    """Extra constructor arguments of the expert engines named by the model config (conv_dtype: the fp8 conv path of
    the FCN expert, fp8_deep: e4m3 operands from conv1_2 on -- see fcn.fp8_plan; streamk: its split tail rounds, a batch-1
    latency option -- see FcnEngine; the AdapNet engine has none)."""
    opts = {}
    if config.get('expert_model', 'fcn') == 'fcn':
        if config.get('conv_dtype', 'bf16') != 'bf16':
            opts['conv_dtype'] = config['conv_dtype']
        if config.get('streamk', False):
            opts['streamk'] = True
        if config.get('fp8_deep', False):
            opts['fp8_deep'] = True
        if config.get('fp8_start'):
            opts['fp8_start'] = config['fp8_start']     # a layer name, or {prefix: layer name} per expert
    return opts


def fp8_guard_bound(config):
    """Label-agreement bound of the accuracy-guarded fp8 plan (FcnEngine.calibrate_guarded), or None where the plan is given:
    model config `fp8_agreement` (default fcn.FP8_GUARD_AGREEMENT = 0.995; 0 / None: off); an explicit `fp8_start` or
    `fp8_deep` is a plan of the caller's and is not second-guessed."""
    from .fcn import FP8_GUARD_AGREEMENT
    if config.get('fp8_start') or config.get('fp8_deep'):
        return None
    bound = config.get('fp8_agreement', FP8_GUARD_AGREEMENT)
    return float(bound) if bound else None


def calibrate_experts(model, data):
    """conv_dtype='fp8': fix every expert's activation scales from the first batch of `data` -- and, unless the config names
    a plan, choose every expert's e4m3 plan by its label agreement with the bf16 graph on that batch (fp8_guard_bound)."""
    from .base_model import iterate_batches
    batch = next(iterate_batches(data, model.config['batchsize']))
    model._graph = None
    bound = fp8_guard_bound(model.config)
    out = {}
    for m in model.modalities:
        x = model._to_device(batch[m], torch.float32)
        eng = model.experts[m]
        out[m] = eng.calibrate_guarded(x, bound) if bound is not None and hasattr(eng, 'calibrate_guarded') else eng.calibrate(x)
    return out


def fp8_plan_report(model):
    """{modality: the engine's calibrate_guarded report or its fixed plan} of an fp8 model (bench records, logs)."""
    rep = {}
    for m in model.modalities:
        eng = model.experts[m]
        rep[m] = eng.fp8_guard if getattr(eng, 'fp8_guard', None) else {
            'chosen': 'bf16' if getattr(eng, 'fp8_off', False) else (getattr(eng, 'fp8_start', None) or ('conv1_2' if getattr(eng, 'fp8_deep', False) else 'conv2_2')),
            'bound': None}
    return rep


def fills_the_chip(model, inputs, rounds=1):
    """Does ONE expert's conv4 map give at least `rounds` rounds of workgroups (16x32-pixel x 64-channel tiles against the CU
    count)?"""
    shapes = {tuple(v.shape[:3]) for v in inputs.values()}
    if len(shapes) != 1:
        return False
    n, h, w = next(iter(shapes))
    tiles = n * ((h // 8 + 15) // 16) * ((w // 8 + 31) // 32) * 8
    return tiles >= rounds * torch.cuda.get_device_properties(model.device).multi_processor_count


def expert_streams(model, inputs):
    """Run the experts side by side on two HIP streams?  model.concurrent_experts: True / False force it; None (default): two
    streams while the launches of one expert leave CUs idle or end in half-empty rounds (2 / 4 / 8 images of 768x384: 2 958 /
    3 406 / 3 552 images/s against 2 176 / 3 051 / 3 461 on one stream), ONE stream from three rounds of conv4 workgroups
    per expert on (12 images: 3 685 against 3 568; 16 images: 3 799-3 829 against 3 636-3 698 on one box, 3 735 against
    3 574-3 617 on another): there every launch fills the chip for several rounds, and two queues only make the persistent
    grids of two kernels share CUs.  Same kernels, same bits either way."""
    conc = getattr(model, 'concurrent_experts', None)
    if conc is None:
        conc = not fills_the_chip(model, inputs, rounds=3)
    return bool(conc)


def paired_from(model, inputs):
    """Index of the first encoder layer the two experts run as ONE launch each (fcn.encoder_layers_pair), or None: two FCN
    experts on the bf16 path without dropout sites or the stream-K option -- and a batch whose conv4 maps give one expert
    at least one round of workgroups (16x32-pixel x 64-channel tiles against the CU count).  Below that the launches are
    latency: both experts' kernels fit the chip side by side on their two streams, and the joins of a paired section only
    cost (one image: 0.46 -> 0.49 ms per step with it)."""
    from .fcn import group_from_index
    gi = group_from_index()
    if gi is None or len(model.modalities) != 2 or not model.config.get('paired_launches', True):
        return None
    if not all(type(e) is FcnEngine and e.pairable() for e in model.experts.values()):
        return None
    # (modalities of different sizes cannot share a launch: both stay on their own streams)
    if not fills_the_chip(model, inputs):
        return None
    return gi


def run_trunks(model, inputs, finish):
    """Both experts up to finish(modality, state-or-None) -> result: each on its own HIP stream (the experts are independent
    until the fusion kernel); from paired_from() on the layers both experts share are ONE launch each on the current stream
    -- whole rounds of workgroups where each expert alone leaves its last round half empty -- and the streams fork again for
    the heads.  The current stream waits for all of them before returning."""
    from .fcn import encoder_layers_pair
    mods = model.modalities
    conc = expert_streams(model, inputs)
    main = torch.cuda.current_stream(model.device)
    if conc and not hasattr(model, '_expert_streams'):
        model._expert_streams = {m: torch.cuda.Stream(device=model.device) for m in mods}

    def each(fn):
        out = {}
        for m in mods:
            if conc:
                side = model._expert_streams[m]
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    out[m] = fn(m)
            else:
                out[m] = fn(m)
        if conc:
            for m in mods:
                main.wait_stream(model._expert_streams[m])
        return out

    gi = paired_from(model, inputs)
    if gi is None:
        return each(lambda m: finish(m, None))
    st = each(lambda m: model.experts[m].encoder_begin(inputs[m], stop=gi))
    a, b = mods
    encoder_layers_pair(model.experts[a], st[a], model.experts[b], st[b])
    return each(lambda m: finish(m, st[m]))


def run_experts(model, batch, wants):
    """Forward every modality's expert (run_trunks), each to its test_pipeline outputs."""
    inputs = {m: model._to_device(batch[m], torch.float32) for m in model.modalities}
    main = torch.cuda.current_stream(model.device)

    def finish(m, st):
        out = test_pipeline(model.experts[m], inputs[m], want=wants, st=st)
        for v in out.values():
            if isinstance(v, torch.Tensor):
                v.record_stream(main)
        return out
    return run_trunks(model, inputs, finish)


def fused_head_applicable(model):
    """The fused two-expert head (ops.fused_head) serves the default prediction of a fusion model with two FCN experts
    whose decoder heads are in the commuted form; config fused_head=False keeps the per-expert outputs
    (`expert_outputs`, `probs`) materialised as the unfused path does."""
    if not model.config.get('fused_head', True) or len(model.modalities) != 2:
        return False
    return all(isinstance(e, FcnEngine) and e.commuted_head() for e in model.experts.values())


def run_fused_head(model, batch, tab, logprior, lognorm=None):
    """Both trunks (run_trunks) up to their low-resolution class scores, then ONE kernel: per-pixel logits, softmax / argmax
    per expert and the Bayes or Dirichlet fusion -> fused labels."""
    inputs = {m: model._to_device(batch[m], torch.float32) for m in model.modalities}
    res = run_trunks(model, inputs, lambda m, st: model.experts[m].lowres_scores(inputs[m], st=st))
    a, b = model.modalities
    S, geo = {m: r[0] for m, r in res.items()}, res[a][1]
    from . import ops
    return ops.fused_head(S[a], S[b], model.experts[a].b['score'], model.experts[b].b['score'], geo[0], geo[1], geo[2],
                          model.config['num_classes'], tab, logprior, lognorm=lognorm)


class FusionModel(BaseModel):
    """Mixture of per-modality FCN experts; subclasses implement `_fusion(expert_outputs)`.
    config: prefixes {modality: prefix}, num_units, num_channels {modality: C_in}, expert_model."""

    expert_wants = ('classification',)

    def __init__(self, name=None, output_dir=None, **config):
        self.modalities = list(config['prefixes'].keys())
        BaseModel.__init__(self, name=name, output_dir=output_dir, custom_training=True, **config)

    def _fusion(self, expert_outputs, output_attr=None):
        raise NotImplementedError

    def _build_graph(self):
        engine_cls, init = expert_factory(self.config['expert_model'], self.config.get('conv_dtype', 'bf16'))
        self.experts = {}
        for m in self.modalities:
            prefix = self.config['prefixes'][m]
            cin = self._modality_channels(m)
            # experts run with trainable=False, batchnorm=False (basic_fusion_model.py:17-18)
            self.variables.update(init(prefix, cin, self.config['num_units'], self.config['num_classes'],
                                       seed=self.config.get('seed')))
            self.experts[m] = engine_cls(prefix, cin, self.config['num_units'], self.config['num_classes'],
                                         self.variables, device=self.device, **engine_options(self.config))
        self.prediction = 'fused_label'

    def _modality_channels(self, m):
        if 'num_channels' in self.config:
            return int(self.config['num_channels'][m])
        return int(self.testdata_description[1][m][-1])

    def _variables_changed(self):
        BaseModel._variables_changed(self)
        for m in self.modalities:
            self.experts[m].load(self.variables)

    def calibrate(self, data):
        return calibrate_experts(self, data)

    def _expert_outputs(self, batch, wants):
        return run_experts(self, batch, wants)

    def _predict_batch_impl(self, batch, output_attr=None):
        wants = self.expert_wants
        if output_attr in ('probs', 'prob') and 'prob' not in wants:
            wants = tuple(wants) + ('prob',)
        self.expert_outputs = self._expert_outputs(batch, wants)
        return self._fusion(self.expert_outputs, output_attr=output_attr)
