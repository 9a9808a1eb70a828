"""Bayes fusion of the experts' label maps (reference: xview/models/bayes_mix.py)."""
from itertools import product

import numpy as np
import torch

from . import ops
from .basic_fusion_model import FusionModel, fused_head_applicable, run_fused_head

UNIFORM_PRIOR = 1.0 / 14     # the reference hard-codes 1/14 regardless of num_classes (bayes_mix.py:42,95)


def _conditional(confusion_T):
    """p(expert output | ground-truth class): nan_to_num(M / M.sum(0)) (bayes_mix.py:36,86)."""
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.nan_to_num(confusion_T / confusion_T.sum(0))


def _prior(confusion_T_last, class_prior):
    """bayes_mix.py:42-54: the data prior comes from the LAST expert's matrix."""
    data_prior = confusion_T_last.sum(0) / confusion_T_last.sum()
    if class_prior == 'uniform':
        return UNIFORM_PRIOR
    if class_prior == 'data':
        return data_prior
    weight = float(class_prior)
    prior = weight * UNIFORM_PRIOR + (1 - weight) * data_prior
    return prior / prior.sum()


def bayes_tables(confusion_matrices, class_prior='data'):
    """Host precompute for xv_bayes_fuse: float32 loglik [E,C,C] = log(1e-20 + cond_e) and
    logprior [C] = log(prior), the per-class constants of bayes_fusion (bayes_mix.py:33-58).
    confusion_matrices: list of float32 [C,C] already transposed (rows = predicted)."""
    C = confusion_matrices[0].shape[0]
    loglik = np.stack([np.log(np.float32(1e-20) + _conditional(m).astype(np.float32), dtype=np.float32)
                       for m in confusion_matrices])
    prior = np.broadcast_to(np.asarray(_prior(confusion_matrices[-1], class_prior), np.float32), (C,))
    with np.errstate(divide='ignore'):
        logprior = np.log(prior, dtype=np.float32)
    return np.ascontiguousarray(loglik), np.ascontiguousarray(logprior)


def bayes_fusion(classifications, confusion_matrices, class_prior='data'):
    """Functional entry point with the reference's signature (bayes_mix.py:12-58;
    experiments/timing.py:52-80).  classifications: list of int64 CUDA tensors [N,H,W].
    Returns (score f32 [N,H,W,C], log_likelihoods, conditionals); the per-expert lists hold the
    [C,C] tables the per-pixel gathers of the reference index into."""
    loglik, logprior = bayes_tables(confusion_matrices, class_prior)
    dev = classifications[0].device
    _, score = ops.bayes_fuse(list(classifications), torch.from_numpy(loglik).to(dev),
                              torch.from_numpy(logprior).to(dev), want_score=True)
    return score, [l for l in loglik], [_conditional(m) for m in confusion_matrices]


def bayes_decision_matrix(confusion_matrices, class_prior='data'):
    """Lookup table of the fused class for every combination of expert outputs
    (bayes_mix.py:61-112); float64 host math, int64 [C]*E."""
    num_classes = confusion_matrices[0].shape[0]
    num_experts = len(confusion_matrices)
    combos = np.array(list(product(*(range(num_classes) for _ in range(num_experts)))))
    log_likelihoods = np.zeros((combos.shape[0], num_experts, num_classes))
    for e, m in enumerate(confusion_matrices):
        with np.errstate(divide='ignore'):
            log_likelihoods[:, e, :] = np.log(1e-20 + _conditional(m)[combos[:, e]])
    with np.errstate(divide='ignore'):
        fused = np.argmax(log_likelihoods.sum(1) + np.log(_prior(confusion_matrices[-1], class_prior)), axis=1)
    return fused.reshape([num_classes for _ in range(num_experts)])


class BayesFusion(FusionModel):
    """config: num_units, num_classes (via data_description), prefixes, num_channels, expert_model,
    class_prior ('data' | 'uniform' | float), confusion_matrices {modality: [C,C] label x pred};
    decision_matrix=True fuses two experts through the bayes_decision_matrix lookup table instead of the
    per-pixel log-likelihood sum (the faster variant timed by experiments/timing.py:87-115); fused_head=False keeps
    the experts' label maps materialised (`expert_outputs`) instead of fusing inside the decoder-head kernel."""

    def __init__(self, output_dir=None, confusion_matrices=False, **config):
        standard_config = {'learning_rate': 0.0, 'class_prior': 'data'}
        standard_config.update(config)
        self.confusion_matrices = {}
        if not confusion_matrices:
            raise UserWarning('ERROR: BayesFusion needs confusion_matrices (the experiment database of '
                              'the reference, `eval_experiments`, is out of scope)')
        order = []
        for key, matrix in confusion_matrices.items():
            order.append(key)
            self.confusion_matrices[key] = np.asarray(matrix).astype('float32').T   # bayes_mix.py:141
        self._matrix_order = order
        FusionModel.__init__(self, 'BayesFusion', output_dir=output_dir, **standard_config)

    def _build_graph(self):
        FusionModel._build_graph(self)
        # modality order = order of the confusion_matrices dict (bayes_mix.py:137-141 overwrites
        # self.modalities before FusionModel.__init__ resets it from `prefixes`)
        mats = [self.confusion_matrices[m] for m in self.modalities]
        loglik, logprior = bayes_tables(mats, self.config['class_prior'])
        self.loglik = torch.from_numpy(loglik).to(self.device)
        self.logprior = torch.from_numpy(logprior).to(self.device)
        self.conditionals = [_conditional(m) for m in mats]
        self.decision_matrix = torch.from_numpy(
            bayes_decision_matrix(mats, self.config['class_prior']).astype(np.int64)).to(self.device)

    def _predict_batch_impl(self, batch, output_attr=None):
        if output_attr is None and not self.config.get('decision_matrix', False) and fused_head_applicable(self):
            # default prediction: nothing but the fused label map is wanted -> one fused head kernel after the trunks
            self.expert_outputs = None
            return run_fused_head(self, batch, self.loglik, self.logprior)
        return FusionModel._predict_batch_impl(self, batch, output_attr)

    def _fusion(self, expert_outputs, output_attr=None):
        labels = [expert_outputs[m]['classification'] for m in self.modalities]
        want_score = output_attr in ('fused_score', 'score')
        if self.config.get('decision_matrix', False) and len(labels) == 2 and not want_score:
            self.probs = {m: expert_outputs[m].get('prob') for m in self.modalities}
            return ops.bayes_fuse_lut(labels[0], labels[1], self.decision_matrix)
        fused, score = ops.bayes_fuse(labels, self.loglik, self.logprior, want_score=want_score)
        self.probs = {m: expert_outputs[m].get('prob') for m in self.modalities}
        if want_score:
            return score
        if output_attr in ('probs', 'prob'):
            return torch.stack([self.probs[m] for m in self.modalities], 1)
        return fused
