"""Dirichlet fusion of the experts' softmax outputs (reference: xview/models/dirichlet_mix.py)."""
from copy import deepcopy

import numpy as np
import torch
from scipy.special import gammaln

from . import ops
from .base_model import BaseModel, iterate_batches
from .basic_fusion_model import (calibrate_experts, engine_options, expert_factory, fused_head_applicable,  # noqa: F401
                                 run_experts, run_fused_head, test_pipeline)
from .dirichlet_fit import find_dirichlet_priors

UNIFORM_PRIOR = 1.0 / 14     # dirichlet_mix.py:116


def class_prior_vector(class_counts, class_prior, num_classes):
    """dirichlet_mix.py:115-129 (float32 like the graph constants)."""
    class_counts = np.asarray(class_counts, np.float32)
    data_prior = (class_counts / (1e-20 + class_counts.sum())).astype('float32')
    if class_prior == 'uniform':
        prior = UNIFORM_PRIOR
    elif class_prior == 'data':
        prior = data_prior
    else:
        weight = float(class_prior)
        prior = weight * UNIFORM_PRIOR + (1 - weight) * data_prior
        prior = prior / prior.sum()
    return np.broadcast_to(np.asarray(prior, np.float32), (num_classes,)).copy()


def dirichlet_tables(dirichlet_params, class_counts, class_prior, sigma):
    """Host precompute for xv_dirichlet_fuse from params[k, c] per expert:
    am1[e,c,k] = sigma*A_e[k,c] - 1, lognorm[e,c] = sum_k lgamma(sigma*A_e[k,c]) - lgamma(sum_k ..)
    ([TF1] Dirichlet.log_prob normaliser), logprior[c] = log(1e-20 + prior[c])  (dirichlet_mix.py:36)."""
    am1, lognorm = [], []
    for A in dirichlet_params:
        conc = (np.float32(sigma) * np.asarray(A, np.float32)).astype(np.float32)     # [k, c]
        am1.append((conc - np.float32(1)).T)
        c64 = conc.astype(np.float64)
        lognorm.append((gammaln(c64).sum(0) - gammaln(c64.sum(0))).astype(np.float32))
    C = am1[0].shape[0]
    prior = class_prior_vector(class_counts, class_prior, C)
    logprior = np.log(np.float32(1e-20) + prior, dtype=np.float32)
    return (np.ascontiguousarray(np.stack(am1), np.float32), np.ascontiguousarray(np.stack(lognorm), np.float32),
            np.ascontiguousarray(logprior, np.float32))


def dirichlet_fusion(probs, dirichlet_params, prior, sigma=1.0):
    """Functional entry point (dirichlet_mix.py:14-36; experiments/timing.py): probs = list of
    float32 CUDA tensors [N,H,W,C] (renormalised inside the kernel), dirichlet_params = list of
    [C,C] arrays (params[k, c]), prior = [C] probabilities.  Returns the fused score [N,H,W,C]."""
    am1, lognorm, _ = dirichlet_tables(dirichlet_params, np.ones(len(prior)), 'uniform', sigma)
    logprior = np.log(np.float32(1e-20) + np.asarray(prior, np.float32), dtype=np.float32)
    dev = probs[0].device
    _, score = ops.dirichlet_fuse(list(probs), torch.from_numpy(am1).to(dev), torch.from_numpy(lognorm).to(dev),
                                  torch.from_numpy(logprior).to(dev), want_score=True)
    return score


class DirichletFusion(BaseModel):
    """config: modalities, num_channels, num_units, expert_model, class_prior, sigma, delta, beta,
    optional dirichlet_params {modality: [C,C], 'class_counts': [C]}; the expert of modality m uses
    prefix m (dirichlet_mix.py:98)."""

    def __init__(self, output_dir=None, **config):
        standard_config = {'learning_rate': 0.0}
        standard_config.update(config)
        self.modalities = config['modalities']
        if 'dirichlet_params' in config:
            measurements = config['dirichlet_params']
            self.dirichlet_params = {m: np.asarray(measurements[m]).astype('float32') for m in self.modalities}
            self.class_counts = np.asarray(measurements['class_counts']).astype('float32')
        else:
            print('WARNING: Could not yet import measurements, you need to fit this model first.')
        BaseModel.__init__(self, name='DirichletFusion', output_dir=output_dir, custom_training=True,
                           **standard_config)

    def _build_graph(self):
        engine_cls, init = expert_factory(self.config['expert_model'], self.config.get('conv_dtype', 'bf16'))
        if not hasattr(self, 'experts'):
            self.experts = {}
            for m in self.modalities:
                cin = int(self.config['num_channels'][m])
                self.variables.update(init(m, cin, self.config['num_units'], self.config['num_classes'],
                                           seed=self.config.get('seed')))
                self.experts[m] = engine_cls(m, cin, self.config['num_units'], self.config['num_classes'],
                                             self.variables, device=self.device, **engine_options(self.config))
        if hasattr(self, 'dirichlet_params'):
            am1, lognorm, logprior = dirichlet_tables([self.dirichlet_params[m] for m in self.modalities],
                                                      self.class_counts, self.config['class_prior'],
                                                      self.config['sigma'])
            self.am1 = torch.from_numpy(am1).to(self.device)
            self.lognorm = torch.from_numpy(lognorm).to(self.device)
            self.logprior = torch.from_numpy(logprior).to(self.device)
            self.prediction = 'fused_label'
        else:
            self.prediction = 0      # dirichlet_mix.py:165-168: no fusion possible before fit()

    def _variables_changed(self):
        BaseModel._variables_changed(self)
        for m in self.modalities:
            self.experts[m].load(self.variables)

    def calibrate(self, data):
        return calibrate_experts(self, data)

    def _predict_batch_impl(self, batch, output_attr=None):
        if not hasattr(self, 'am1'):
            raise UserWarning('ERROR: DirichletFusion has no measurements yet, call fit() first')
        if output_attr is None and fused_head_applicable(self):
            # default prediction: the experts' probabilities never leave the registers of the fused head kernel
            self.probs = None
            return run_fused_head(self, batch, self.am1, self.logprior, lognorm=self.lognorm)
        outs = run_experts(self, batch, ('prob',))
        probs = [outs[m]['prob'] for m in self.modalities]
        self.probs = dict(zip(self.modalities, probs))
        want_score = output_attr in ('fused_score', 'score')
        fused, score = ops.dirichlet_fuse(probs, self.am1, self.lognorm, self.logprior, want_score=want_score)
        return score if want_score else fused

    # ---- fit = measure sufficient statistics on the GPU, Newton-fit on the host -------------------
    def _get_sufficient_statistic(self, data):
        """dirichlet_mix.py:175-205: per modality S[c,k] = sum_{label=c} log(1e-10 + p[k]) and the
        class counts, accumulated over all batches (float64 / int64 on the device)."""
        C = self.config['num_classes']
        S = {m: torch.zeros((C, C), dtype=torch.float64, device=self.device) for m in self.modalities}
        counts = torch.zeros(C, dtype=torch.int64, device=self.device)
        scratch = torch.zeros(C, dtype=torch.int64, device=self.device)
        for batch in self._device_batches(iterate_batches(data, self.config['batchsize']), labels=True):
            labels = self._to_device(batch['labels'], torch.int32)
            outs = run_experts(self, batch, ('prob',))
            for i, m in enumerate(self.modalities):
                ops.dirichlet_suffstats(outs[m]['prob'], labels, S[m], counts if i == 0 else scratch)
        return self._allreduce_statistics({m: S[m] for m in self.modalities}, counts)

    def _allreduce_statistics(self, S, counts):
        """Sum the per-rank statistics when running one process per GPU (each rank measured its
        shard of the data): one tiny all-reduce per tensor, RCCL over xGMI."""
        from .parallel import allreduce_sum_
        allreduce_sum_(counts, *[S[m] for m in S])
        return {m: S[m].cpu().numpy() for m in S}, counts.cpu().numpy()

    def _fit_sufficient_statistic(self, counts, class_counts):
        """dirichlet_mix.py:207-257."""
        C = self.config['num_classes']

        def dirichlet_em(measurements):
            params = np.ones((C, C)).astype('float64')
            for c in range(C):
                if class_counts[c] == 0:
                    params[:, c] = np.ones(C)
                    continue
                ss = (measurements[c, :] / class_counts[c]).astype('float64')
                neg_ss = (measurements.sum(0) - measurements[c, :]) / (class_counts.sum() - class_counts[c])
                params[:, c] = find_dirichlet_priors(ss, neg_ss, np.ones(C, 'float64'), max_iter=10000,
                                                     delta=self.config['delta'], beta=self.config['beta'])
            return params

        self.dirichlet_params = {m: dirichlet_em(counts[m]) for m in self.modalities}
        self.class_counts = class_counts
        self._initialize_graph()       # rebuild the tables with the new measurements

    def fit(self, data, *args, **kwargs):
        """Measure the experts against the ground truth of `data`, then fit the class-conditional
        Dirichlets (dirichlet_mix.py:259-273).  Returns {modality: [C,C], 'class_counts': [C]}."""
        modality_counts, class_counts = self._get_sufficient_statistic(data)
        self._fit_sufficient_statistic(modality_counts, class_counts)
        ret = deepcopy(self.dirichlet_params)
        ret['class_counts'] = self.class_counts
        return ret
