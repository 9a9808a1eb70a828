"""Host-side regularised, discriminative Dirichlet maximum-likelihood fit.

Counterpart of the fitter the reference's DirichletFusion actually calls
(xview/models/dirichletDifferentiation.py:129-192 `findDirichletPriors`, used at
dirichlet_mix.py:243-245).  C*E tiny float64 solves per fit -> stays on the CPU (vectorised numpy).

Objective maximised over alpha > 0 (dirichletDifferentiation.py:38-45):
    (1-beta) * [lgamma(sum a) - sum lgamma(a)] + sum a*ss - delta * sum a^2 - beta * sum a*neg_ss
Iteration (same accept / fallback order as the reference so that results agree to rounding):
  1. stop when |grad|^2 < 2^-20
  2. quasi-Newton step with the Dirichlet Hessian  diag(h) + c*11^T,  h_k = (1-beta) trigamma(a_k),
     c = -(1-beta) trigamma(sum a)  (Minka, "Estimating a Dirichlet distribution", eq. 18; the
     delta term is NOT in the Hessian, as in the reference); accept if the loss decreases
  3. otherwise plain gradient steps a + lr*grad with lr = 0.9, 0.81, ... until the loss does not
     exceed the current one; stop when lr < 2^-10
"""
import numpy as np
from scipy.special import gammaln, polygamma, psi

GRAD_TOL_SQ = 2.0 ** -20
LEARN_RATE_TOL = 2.0 ** -10


def _loss(alpha, ss, neg_ss, beta, delta):
    if np.any(alpha <= 0):
        return float('inf')
    value = (1 - beta) * gammaln(alpha.sum())
    value -= (1 - beta) * gammaln(alpha).sum()
    value += (alpha * ss).sum()
    value -= delta * np.square(alpha).sum()
    value -= beta * (alpha * neg_ss).sum()
    return -value


def _gradient(alpha, ss, neg_ss, beta, delta):
    return (1 - beta) * psi(alpha.sum()) + ss - (1 - beta) * psi(alpha) - 2 * delta * alpha - beta * neg_ss


def _newton_step(alpha, grad, beta):
    h_const = -(1 - beta) * polygamma(1, alpha.sum())
    h_diag = (1 - beta) * polygamma(1, alpha)
    b = (grad / h_diag).sum() / (1.0 / h_const + (1.0 / h_diag).sum())
    return (b - grad) / h_diag


def find_dirichlet_priors(ss, neg_ss, init_alphas, max_iter=1000, delta=1e-2, beta=1e-2, verbose=False):
    """ss / neg_ss: mean log-probabilities of the class' own / all other pixels ([K] float64).
    Returns the fitted concentration parameters [K] (float64)."""
    ss = np.asarray(ss, np.float64)
    neg_ss = np.asarray(neg_ss, np.float64)
    alpha = np.asarray(init_alphas, np.float64).copy()
    current = _loss(alpha, ss, neg_ss, beta, delta)
    for _ in range(max_iter):
        grad = _gradient(alpha, ss, neg_ss, beta, delta)
        if (grad ** 2).sum() < GRAD_TOL_SQ:
            if verbose:
                print('Converged with small gradient')
            return alpha
        with np.errstate(all='ignore'):
            trial = alpha + _newton_step(alpha, grad, beta)
            loss = _loss(trial, ss, neg_ss, beta, delta)
        if loss < current:
            current, alpha = loss, trial
            continue
        loss, rate = float('inf'), 1.0
        while loss > current:
            rate *= 0.9
            trial = alpha + grad * rate
            loss = _loss(trial, ss, neg_ss, beta, delta)
        if rate < LEARN_RATE_TOL:
            if verbose:
                print('Converged with small learn rate')
            return alpha
        current, alpha = loss, trial
    if verbose:
        print('Reached max iterations')
    return alpha


# the reference's spelling, for callers that import it by that name
findDirichletPriors = find_dirichlet_priors
