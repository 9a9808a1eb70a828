"""numpy restatement of the reference's per-pixel fusion, statistics and scoring
(TEST INFRASTRUCTURE, see oracle/__init__.py).  fp32 where the reference graph is fp32,
float64 where the reference runs numpy on the host."""
from itertools import product

import numpy as np
from scipy.special import gammaln

UNIFORM_PRIOR = 1.0 / 14   # hard-coded regardless of C: bayes_mix.py:42,95, dirichlet_mix.py:116


# ---- Bayes fusion -------------------------------------------------------------------

def bayes_conditional(confusion_T):
    """bayes_mix.py:36: nan_to_num(M / M.sum(0)) with M = confusion.T (rows = predicted,
    cols = true), so cond[p, t] = P(pred=p | true=t).  Dtype follows the input (float32 in
    BayesFusion, bayes_mix.py:141)."""
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.nan_to_num(confusion_T / confusion_T.sum(0))


def class_prior_from_matrix(confusion_T_last, class_prior):
    """bayes_mix.py:42-54; data prior from the LAST expert's matrix (bayes_mix.py:43)."""
    data_prior = confusion_T_last.sum(0) / confusion_T_last.sum()
    if class_prior == 'uniform':
        return UNIFORM_PRIOR
    if class_prior == 'data':
        return data_prior
    weight = float(class_prior)
    prior = weight * UNIFORM_PRIOR + (1 - weight) * data_prior
    return prior / prior.sum()


def bayes_fusion(classifications, confusion_matrices_T, class_prior='data'):
    """bayes_mix.py:12-58.  classifications: list of int [N,H,W]; confusion_matrices_T: list
    of float32 [C,C] already transposed as BayesFusion.__init__ does.  fp32 graph math.
    Returns (score [N,H,W,C] f32, [ll_e], [cond_e gathered])."""
    lls, conds = [], []
    for lab, m in zip(classifications, confusion_matrices_T):
        cond = bayes_conditional(m).astype(np.float32)
        g = cond[lab]                                   # tf.gather -> [N,H,W,C]
        conds.append(g)
        lls.append(np.log(np.float32(1e-20) + g, dtype=np.float32))
    prior = class_prior_from_matrix(confusion_matrices_T[-1], class_prior)
    logprior = np.log(np.asarray(prior, np.float32), dtype=np.float32)
    total = lls[0].copy()
    for ll in lls[1:]:
        total = total + ll                              # reduce_sum over the stack axis
    return (total + logprior).astype(np.float32), lls, conds


def bayes_decision_matrix(confusion_matrices_T, class_prior='data'):
    """bayes_mix.py:61-112 restated (float64 accumulation as numpy does there)."""
    C = confusion_matrices_T[0].shape[0]
    E = len(confusion_matrices_T)
    combos = np.array(list(product(*(range(C) for _ in range(E)))))
    ll = np.zeros((combos.shape[0], E, C))
    for e, m in enumerate(confusion_matrices_T):
        cond = bayes_conditional(m)
        with np.errstate(divide='ignore'):
            ll[:, e, :] = np.log(1e-20 + cond[combos[:, e]])
    prior = class_prior_from_matrix(confusion_matrices_T[-1], class_prior)
    with np.errstate(divide='ignore'):
        fused = np.argmax(ll.sum(1) + np.log(prior), axis=1)
    return fused.reshape([C] * E)


# ---- Dirichlet fusion ----------------------------------------------------------------

def dirichlet_log_prob(x, alpha):
    """[TF1] tf.contrib.distributions.Dirichlet(alpha).log_prob(x), no support validation:
    sum((alpha-1) log x) - (sum lgamma(alpha) - lgamma(sum alpha)).  fp32."""
    x = np.asarray(x, np.float32)
    alpha = np.asarray(alpha, np.float32)
    lognorm = np.float32(gammaln(alpha.astype(np.float64)).sum() - gammaln(alpha.astype(np.float64).sum()))
    return ((alpha - np.float32(1)) * np.log(x, dtype=np.float32)).sum(-1, dtype=np.float32) - lognorm


def dirichlet_prior(class_counts, class_prior):
    """dirichlet_mix.py:116-129."""
    class_counts = np.asarray(class_counts, np.float32)
    data_prior = (class_counts / (1e-20 + class_counts.sum())).astype(np.float32)
    if class_prior == 'uniform':
        return np.float32(UNIFORM_PRIOR)
    if class_prior == 'data':
        return data_prior
    weight = float(class_prior)
    prior = weight * UNIFORM_PRIOR + (1 - weight) * data_prior
    return prior / prior.sum()


def renormalise(prob):
    """dirichlet_mix.py:100-102: p / reduce_sum(p, axis=3, keepdims)."""
    prob = np.asarray(prob, np.float32)
    return (prob / prob.sum(-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def dirichlet_fusion(probs, dirichlet_params, prior, sigma=1.0):
    """dirichlet_mix.py:14-36 + 107-113: L_e[..., c] = Dir(sigma*A_e[:, c]).log_prob(1e-20+p_e);
    fused = sum_e L_e + log(1e-20 + prior).  probs: list of [N,H,W,C] f32 (already
    renormalised); dirichlet_params: list of [C,C] (params[k, c])."""
    C = probs[0].shape[-1]
    total = None
    for p, A in zip(probs, dirichlet_params):
        A = np.asarray(A, np.float32)
        x = np.float32(1e-20) + np.asarray(p, np.float32)
        L = np.stack([dirichlet_log_prob(x, np.float32(sigma) * A[:, c]) for c in range(C)], axis=-1)
        total = L if total is None else total + L
    return (total + np.log(np.float32(1e-20) + np.asarray(prior, np.float32), dtype=np.float32)).astype(np.float32)


def sufficient_statistics(prob, labels, num_classes):
    """dirichlet_mix.py:142-163: S[c,k] = sum_{pixels: label=c} log(1e-10 + p[k]);
    class_counts[c] = #pixels with label c.  Sums in float64 here (the reference sums each
    batch in fp32 on device, then accumulates batches in float64 on the host)."""
    prob = np.asarray(prob, np.float32)
    lab = np.asarray(labels)
    logp = np.log(np.float32(1e-10) + prob, dtype=np.float32).astype(np.float64)
    S = np.zeros((num_classes, num_classes))
    counts = np.zeros(num_classes, np.int64)
    for c in range(num_classes):
        m = lab == c
        counts[c] = int(m.sum())
        S[c] = logp[m].sum(0)
    return S, counts


# ---- confusion matrix and measures ------------------------------------------------------

def confusion_matrix(labels, prediction, num_classes):
    """base_model.py:136-151: labels<0 -> class C, tf.confusion_matrix(C+1) sliced to [C,C];
    rows = ground truth, cols = prediction."""
    lab = np.asarray(labels).reshape(-1).astype(np.int64)
    pred = np.asarray(prediction).reshape(-1).astype(np.int64)
    lab = np.where(lab < 0, num_classes, lab)
    cm = np.zeros((num_classes + 1, num_classes + 1), np.int64)
    np.add.at(cm, (lab, pred), 1)
    return cm[:num_classes, :num_classes]


def score_measures(cm):
    """base_model.py:315-329."""
    cm = np.asarray(cm, np.float64)
    with np.errstate(divide='ignore', invalid='ignore'):
        m = {'confusion_matrix': cm}
        m['recall'] = np.diag(cm) / cm.sum(1)
        m['precision'] = np.diag(cm) / cm.sum(0)
        m['F1'] = 2 * m['precision'] * m['recall'] / (m['precision'] + m['recall'])
        m['mean_F1'] = np.nanmean(m['F1'])
        m['total_accuracy'] = np.diag(cm)[1:].sum() / cm[1:, :].sum()
        m['IoU'] = np.diag(cm) / (cm.sum(1) + cm.sum(0) - np.diag(cm))
        m['mean_IoU'] = np.nanmean(m['IoU'][1:])
    return m


# ---- optimizers ([TF1] formulas) -----------------------------------

def adam_step(theta, g, m, v, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return theta - lr_t * m / (np.sqrt(v) + eps), m, v


def rmsprop_step(theta, g, ms, lr=1e-4, decay=0.9, eps=1e-10):
    """tf.train.RMSPropOptimizer (momentum 0): ms initialised to ONE by the caller."""
    ms = decay * ms + (1 - decay) * g * g
    return theta - lr * g / np.sqrt(ms + eps), ms


def adagrad_step(theta, g, acc, lr=1e-4):
    """tf.train.AdagradOptimizer: accumulator initialised to 0.1 by the caller."""
    acc = acc + g * g
    return theta - lr * g / np.sqrt(acc), acc
