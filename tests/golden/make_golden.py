#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ from the reference.

TEST INFRASTRUCTURE.  Runs ONLY in the build container, where the read-only
reference checkout is mounted at /root/reference.  Nothing here is shipped or
imported by the product; only the *outputs* (small .npz / .json data files)
are committed.  No reference source is copied: the numpy/scipy-only reference
functions are imported in-place (TensorFlow stubbed, package __init__ files
bypassed) and called on inputs defined in this script.

What is produced (and what pins it):
  bilinear_kernels.npz   xview/models/custom_layers.py:8-25  bilinear_filter_initializer
  notebook_868.npz       'Experimental Details.ipynb' cell 12 output: experiment 868
                         confusion matrices (rgb, depth measure set; fused test set)
                         and the measures printed there (base_model.py:315-329)
  bayes_lut.npz          xview/models/bayes_mix.py:61-112 bayes_decision_matrix on the
                         notebook matrices, class_prior in {'data','uniform',0.5}
  dirichlet_fit.npz      xview/models/dirichletDifferentiation.py:129-192
                         findDirichletPriors on seeded synthetic sufficient statistics
  weight_names.json      'Synthia Rand Cityscapes Examples.ipynb' variable-name list of a
                         BN-free FCN expert (npz weight schema)

Usage:  python tests/golden/make_golden.py     (from the repo root)
"""
import contextlib
import importlib
import importlib.util
import io
import json
import os
import re
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


class _Any(types.ModuleType):
    """Attribute-absorbing stand-in for modules that are absent here (tensorflow...)."""

    def __getattr__(self, k):
        if k.startswith('__'):
            raise AttributeError(k)
        m = _Any(self.__name__ + '.' + k)
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return _Any('call')


def _install_stubs():
    for n in ['tensorflow', 'tensorflow.python', 'tensorflow.python.layers',
              'tensorflow.python.layers.layers', 'tensorflow.python.ops',
              'tensorflow.python.ops.init_ops', 'experiments', 'experiments.utils', 'tqdm']:
        sys.modules[n] = _Any(n)
    sys.modules['tensorflow.python.ops.init_ops'].Initializer = object
    sys.modules['tensorflow'].constant_initializer = lambda value=None, **k: value
    for pkg, path in [('xview', REF + '/xview'), ('xview.models', REF + '/xview/models')]:
        m = types.ModuleType(pkg)
        m.__path__ = [path]
        sys.modules[pkg] = m


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def notebook_868():
    nb = json.load(open(REF + '/Experimental Details.ipynb'))
    text = None
    for cell in nb['cells']:
        for o in cell.get('outputs', []):
            t = o.get('text') or o.get('data', {}).get('text/plain')
            if t and "'confusion_matrices'" in ''.join(t):
                text = ''.join(t)
    assert text is not None
    text = re.sub(r"\{'py/id': \d+\}", 'None', text)
    info = eval(text, {'array': np.array, 'nan': float('nan')})
    return info


def main():
    _install_stubs()
    cl = importlib.import_module('xview.models.custom_layers')
    bm = importlib.import_module('xview.models.bayes_mix')
    dd = _load_file('ref_dirichletDifferentiation',
                    REF + '/xview/models/dirichletDifferentiation.py')

    # --- bilinear kernels -------------------------------------------------
    k4 = np.asarray(cl.bilinear_filter_initializer([4, 4, 5, 5]), dtype=np.float64)
    k16 = np.asarray(cl.bilinear_filter_initializer([16, 16, 3, 3]), dtype=np.float64)
    # the [k,k,filters,in] kernels AdapNet asks for (adapnet.py:158,164): filters < in
    k4_rect = np.asarray(cl.bilinear_filter_initializer([4, 4, 3, 7]), dtype=np.float64)
    k16_rect = np.asarray(cl.bilinear_filter_initializer([16, 16, 2, 5]), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'bilinear_kernels.npz'), k4=k4, k16=k16, k4_rect=k4_rect, k16_rect=k16_rect)

    # --- notebook confusion matrices + measures ----------------------------
    info = notebook_868()
    cms = info['confusion_matrices']
    meas = info['measurements']
    # cm_*: measure-set matrices the Bayes fusion is built from (info['confusion_matrices']);
    # test_cm_*: the test-set matrices the printed measures were derived from.
    save = {'cm_rgb': np.asarray(cms['rgb'], np.float64),
            'cm_depth': np.asarray(cms['depth'], np.float64),
            'test_cm_fusion': np.asarray(info['confusion_matrix'], np.float64),
            'test_cm_rgb': np.asarray(meas['rgb']['confusion_matrix'], np.float64),
            'test_cm_depth': np.asarray(meas['depth']['confusion_matrix'], np.float64)}
    for who, m in meas.items():
        for key, val in m.items():
            if val is None or key == 'confusion_matrix':
                continue
            save['{}__{}'.format(who, key)] = np.asarray(val, np.float64)
    np.savez_compressed(os.path.join(OUT, 'notebook_868.npz'), **save)

    # --- bayes decision matrices (reference numpy function) ---------------
    # BayesFusion.__init__ (bayes_mix.py:141,145-147) hands bayes_fusion the TRANSPOSED
    # float32 matrices; the LUT function is fed the same way here.
    mats = [save['cm_rgb'].astype('float32').T, save['cm_depth'].astype('float32').T]
    luts = {}
    with np.errstate(divide='ignore', invalid='ignore'):
        for name, prior in [('data', 'data'), ('uniform', 'uniform'), ('w0p5', 0.5)]:
            luts['lut_' + name] = bm.bayes_decision_matrix(mats, prior).astype(np.int64)
        # a matrix with an empty ground-truth class (exercises nan_to_num)
        holes = [m.copy() for m in mats]
        for m in holes:
            m[:, 5] = 0
        luts['lut_holes_data'] = bm.bayes_decision_matrix(holes, 'data').astype(np.int64)
        luts['lut_holes_uniform'] = bm.bayes_decision_matrix(holes, 'uniform').astype(np.int64)
    np.savez_compressed(os.path.join(OUT, 'bayes_lut.npz'), **luts)

    # --- Dirichlet fitter ---------------------------------------------------
    rng = np.random.default_rng(20181001)
    cases = {}
    idx = 0
    for C in (4, 12):
        for (delta, beta) in [(0.0, 0.0), (1e-2, 1e-2), (0.05, 0.3)]:
            alpha_true = rng.uniform(0.3, 6.0, size=C)
            alpha_neg = rng.uniform(0.3, 3.0, size=C)
            x = rng.dirichlet(alpha_true, size=4000)
            y = rng.dirichlet(alpha_neg, size=4000)
            ss = np.log(1e-10 + x).mean(0)
            neg_ss = np.log(1e-10 + y).mean(0)
            with contextlib.redirect_stdout(io.StringIO()):
                fit = dd.findDirichletPriors(ss, neg_ss, np.ones(C), max_iter=10000,
                                             delta=delta, beta=beta)
            cases['case%d_ss' % idx] = ss
            cases['case%d_neg_ss' % idx] = neg_ss
            cases['case%d_delta_beta' % idx] = np.array([delta, beta])
            cases['case%d_alpha' % idx] = np.asarray(fit, np.float64)
            idx += 1
    cases['num_cases'] = np.array(idx)
    np.savez_compressed(os.path.join(OUT, 'dirichlet_fit.npz'), **cases)

    # --- Dirichlet fitter, degenerate inputs (round 3): classes seen on ONE pixel, near-certain experts (log-probabilities
    # near 0 / near log(1e-10)), strong regularisers, a far start with few iterations, statistics of opposite sign -- the
    # corners where the reference's fallbacks (log-space trial that only matters through its OverflowError, the line
    # search that starts from the constant 10000000: dirichletDifferentiation.py:167-187) decide the result
    rng = np.random.default_rng(20261003)
    deg, idx = {}, 0

    def add(ss, neg_ss, init, max_iter, delta, beta):
        nonlocal idx
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), np.errstate(all='ignore'):
            fit = dd.findDirichletPriors(np.asarray(ss, np.float64), np.asarray(neg_ss, np.float64),
                                         np.asarray(init, np.float64), max_iter=max_iter, delta=delta, beta=beta)
        deg['case%d_ss' % idx] = np.asarray(ss, np.float64)
        deg['case%d_neg_ss' % idx] = np.asarray(neg_ss, np.float64)
        deg['case%d_init' % idx] = np.asarray(init, np.float64)
        deg['case%d_params' % idx] = np.array([max_iter, delta, beta], np.float64)
        deg['case%d_alpha' % idx] = np.asarray(fit, np.float64)
        deg['case%d_message' % idx] = np.array(buf.getvalue().strip().splitlines()[-1] if buf.getvalue().strip() else '')
        idx += 1

    C = 12
    one = rng.dirichlet(np.full(C, 0.05))                      # a class seen on one pixel with a peaked posterior
    many = np.log(1e-10 + rng.dirichlet(np.full(C, 0.5), size=3000)).mean(0)
    add(np.log(1e-10 + one), many, np.ones(C), 10000, 1e-2, 1e-2)
    certain = np.full(C, np.log(1e-10))                        # experts certain of class 3: log p = 0 there, log(1e-10) elsewhere
    certain[3] = np.log(1e-10 + 1.0)
    add(certain, many, np.ones(C), 10000, 1e-2, 1e-2)
    add(np.full(C, -1e-3), np.full(C, -1e-3), np.ones(C), 10000, 1e-2, 1e-2)      # statistics near 0 (no such simplex)
    add(many, many, np.ones(C), 10000, 5.0, 1e-2)              # strong delta
    add(many, np.log(1e-10 + rng.dirichlet(np.full(C, 2.0), size=3000)).mean(0), np.ones(C), 10000, 1e-2, 0.9)  # strong beta
    add(many, many, np.full(C, 50.0), 20, 1e-2, 1e-2)          # far start, 20 iterations: "Reached max iterations"
    add(-many, many, np.ones(C), 200, 0.0, 0.0)                # positive statistics, no regulariser: runs away until max_iter
    add(np.log(1e-10 + one), np.log(1e-10 + one), np.ones(C), 10000, 0.0, 0.5)
    deg['num_cases'] = np.array(idx)
    np.savez_compressed(os.path.join(OUT, 'dirichlet_fit_degenerate.npz'), **deg)

    # --- variable-name list of a BN-free FCN expert -------------------------
    nb = json.load(open(REF + '/Synthia Rand Cityscapes Examples.ipynb'))
    names = None
    for cell in nb['cells']:
        for o in cell.get('outputs', []):
            t = o.get('text') or o.get('data', {}).get('text/plain')
            if not t:
                continue
            s = ''.join(t)
            # the list appears as import_weights' per-variable warnings (base_model.py:446-448)
            found = re.findall(r"WARNING: (rgb/[A-Za-z0-9_]+/(?:kernel|bias)) not found", s)
            if len(found) > 20:
                names = found
    assert names, 'variable list not found'
    # de-duplicate, keep order
    seen, uniq = set(), []
    for n in names:
        if n not in seen:
            seen.add(n)
            uniq.append(n)
    json.dump({'variables': uniq}, open(os.path.join(OUT, 'weight_names.json'), 'w'), indent=1)
    print('golden vectors written to', OUT)


if __name__ == '__main__':
    main()
