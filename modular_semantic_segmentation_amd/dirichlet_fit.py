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
  3. the reference then evaluates a log-space trial a * exp(step) (:167-171) whose loss it never uses -- its effects
     are (i) `math.exp` overflowing (step > ~709.78) ends the fit with the current alpha (:172-174) and (ii) the trial
     is what the line search below starts from; both are kept
  4. plain gradient steps a + lr*grad with lr = 0.9, 0.81, ..., starting from the constant loss 10000000 (:176, so a
     current loss above that accepts the log-space trial unseen), until the loss does not exceed the current one; stop
     when lr < 2^-10
Pinned against the imported reference on well-conditioned AND degenerate statistics (tests/golden/dirichlet_fit.npz,
dirichlet_fit_degenerate.npz: single-pixel classes, certain experts, strong regularisers, far starts, run-away fits).
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


_EXP_OVERFLOW = 709.782712893384      # math.exp raises OverflowError above this (the reference's only source of one)


def _log_space_trial(alpha, grad, beta):
    """alpha * exp(step) with the diagonal-Hessian step on log alpha (getPredictedStepAlt, :80-99); None where the
    reference's math.exp would raise OverflowError."""
    h_const = -(1 - beta) * polygamma(1, alpha.sum())
    h_diag = (1 - beta) * polygamma(1, alpha)
    den = grad - alpha * h_diag
    z = h_const * (alpha / den).sum()
    big_s = (1.0 / den / (1 + z)).sum()
    step = grad / den * (1 - h_const * alpha * big_s)
    if np.any(step > _EXP_OVERFLOW):
        return None
    return alpha * np.exp(step)


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
        with np.errstate(all='ignore'):
            trial = _log_space_trial(alpha, grad, beta)
        if trial is None:
            if verbose:
                print('got overflow error, returning')
            return alpha
        loss, rate = 10000000, 1.0
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
