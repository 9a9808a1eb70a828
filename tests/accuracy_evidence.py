"""Accuracy evidence for BASELINE.json's north_star ("labels bit-exact / logits within a stated tolerance vs the
reference's fp32 path; mIoU within 0.1 pp"): train both SimpleFCN experts with the MI355X `fit()` on the procedural
RGB-D task (datasets/synthetic.py -- no real dataset is reachable), then run the SAME trained weights through

  * the HIP path (bf16 storage, fp32 accumulation) and
  * the fp32 oracle (oracle/fcn_oracle.py + fusion_oracle.py: the op-for-op restatement of the reference graph)

on held-out 768x384 images, for each expert and for Bayes and Dirichlet fusion, and report mean IoU
(base_model.py:315-329, the measure behind 'Synthia Rand Cityscapes Examples.ipynb':846-847) of both, their
difference, the label agreement and the logit error.

TEST INFRASTRUCTURE (imports oracle/): used by tests/test_accuracy_gpu.py and by bench.py's cpu_baseline leg, where
the oracle pass over the held-out images is also the timed CPU sample."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

C, U = 12, 64
MODS = (('rgb', 3), ('depth', 1))


DEPTH_UNIT = 16384.0     # the depth expert's first layer starts at 1 / DEPTH_UNIT of its Glorot draw (see train_experts)


def train_experts(h, w, steps, batch=8, n_train=32, seed=1, device='cuda', learning_rate=1e-4, log=None,
                  depth_unit=DEPTH_UNIT, min_miou=None, max_steps=5000):
    """Both experts from [TF1] initialisers with the reference's default optimizer (Adam, 1e-4; base_model.py:153-162)
    through `SimpleFCN.fit`.  Returns (variables of both experts, training set)."""
    from modular_semantic_segmentation_amd import get_model
    from modular_semantic_segmentation_amd.datasets.synthetic import augmented_stream, data_description, make_rgbd_shapes
    train = make_rgbd_shapes(n_train, h, w, seed=seed)                          # the measure set of the fusion fits
    clean = {k: torch.from_numpy(v).to(device)
             for k, v in make_rgbd_shapes(n_train, h, w, seed=seed + 50, rgb_noise=0, depth_noise=0).items()}
    val = make_rgbd_shapes(4, h, w, seed=seed + 77)
    desc = data_description()
    variables = {}
    for m, cin in MODS:
        net = get_model('fcn')(m, desc, m, num_units=U, batch_normalization=False, batchsize=batch,
                               learning_rate=learning_rate, trainer='adam', seed=seed + cin, device=str(device),
                               sync_loss=False)
        if m == 'depth':
            # Raw uint16 depth: the first layer of a trained expert absorbs the range; start it there.  The unit matters
            # for learnability, not for parity: a scalar input is split into classes by THRESHOLDS, i.e. by conv1_1's
            # biases, which start at zero and move ~1e-4 per Adam step -- they can only reach thresholds of order one,
            # so the kernel is scaled to make the 0..65535 range span a few units (with 1/256 the expert stalls at
            # 0.56 mIoU: measured).
            net.variables['depth/conv1_1/kernel'] = net.variables['depth/conv1_1/kernel'] / depth_unit
            net._variables_changed()
        t0 = time.perf_counter()
        # "train until useful": `steps` Adam steps, then -- training from scratch is run-to-run different (fp32 atomics) and
        # the depth expert sometimes lags -- further rounds of 500 until a validation set scores mean IoU > min_miou
        stream = augmented_stream(clean, m, seed=seed + cin)
        net.fit(stream, steps, output=False)
        done = steps
        # (RGB must get GOOD, not just useful: an expert that stops at 0.8 leaves ~0.1 % of the pixels on near-ties, and
        # bf16 noise then moves its mean IoU by up to 0.25 points either way -- measured once in ~15 runs; at 0.9+ the
        # difference stays within 0.08.  Depth alone cannot get there on this task: 0.65.)
        need = (min_miou or {}).get(m, 0.9 if m == 'rgb' else 0.65) if not isinstance(min_miou, float) else min_miou
        while done < max_steps and net.score(val)[0]['mean_IoU'] <= need:
            net.fit(stream, 500, output=False)
            done += 500
        net._sync_variables()
        torch.cuda.synchronize()
        if log is not None:
            log('trained %s expert: %d steps of %d images at %dx%d in %.1f s, last loss %.4f'
                % (m, done, batch, w, h, time.perf_counter() - t0, float(net.loss.item())))
        variables.update(net.variables)
    return variables, train


def _miou(labels, pred):
    from modular_semantic_segmentation_amd.base_model import score_measures
    lab = labels.reshape(-1).astype(np.int64)
    ok = lab >= 0
    cm = np.bincount(lab[ok] * C + pred.reshape(-1)[ok].astype(np.int64), minlength=C * C).reshape(C, C)
    return float(score_measures(cm)['mean_IoU']), cm


def hip_predictions_fp8(variables, calibration, heldout, cms, device='cuda', guarded=True):
    """The same experts with conv_dtype='fp8' (BASELINE config "fp8 MFMA conv path"): scales calibrated on a batch of the
    training set, then each expert's labels and the Bayes fusion of them on `heldout`.  guarded (the models' default): every
    expert's e4m3 plan is chosen by calibrate() from its label agreement with the bf16 graph on the calibration batch
    (FcnEngine.calibrate_guarded, bound 0.995); guarded=False: round 5's fixed plan (e4m3 operands from conv2_2 on) for both.
    out['plan'] = {model: {modality: report}}."""
    from modular_semantic_segmentation_amd import get_model
    from modular_semantic_segmentation_amd.basic_fusion_model import fp8_plan_report
    from modular_semantic_segmentation_amd.datasets.synthetic import data_description
    desc = data_description()
    extra = {} if guarded else {'fp8_agreement': 0}
    out = {'plan': {}}
    for m, cin in MODS:
        net = get_model('fcn')(m, desc, m, num_units=U, batch_normalization=False, batchsize=4, device=str(device),
                               conv_dtype='fp8', **extra)
        net.variables.update({k: v for k, v in variables.items() if k.startswith(m + '/')})
        net._variables_changed()
        net.calibrate(calibration)
        out[m] = net.predict(heldout)
        out['plan'][m] = net.engine.fp8_guard if net.engine.fp8_guard else {'chosen': 'conv2_2', 'bound': None}
    bayes = get_model('bayes_fusion')(confusion_matrices=cms, prefixes={'rgb': 'rgb', 'depth': 'depth'},
                                      data_description=desc, num_units=U, num_channels={'rgb': 3, 'depth': 1},
                                      expert_model='fcn', class_prior='data', batchsize=4, device=str(device),
                                      conv_dtype='fp8', **extra)
    bayes.variables.update(variables)
    bayes._variables_changed()
    bayes.calibrate(calibration)
    out['bayes'] = bayes.predict(heldout)
    out['plan']['bayes'] = fp8_plan_report(bayes)
    return out


def hip_predictions_exact(variables, heldout, cms, n_images, device='cuda'):
    """The same experts with conv_dtype='fp32' (csrc/exact_f32.hip: the graph in plain float32, no bf16 storage) on the
    first `n_images` held-out images: expert labels + logits and the Bayes fusion.  Against the fp32 oracle these differ
    only by fp32 summation order -- so what the bf16 path loses against the oracle, it loses to bf16."""
    from modular_semantic_segmentation_amd import get_model
    from modular_semantic_segmentation_amd.datasets.synthetic import data_description
    desc = data_description()
    sub = {k: v[:n_images] for k, v in heldout.items()}
    out = {}
    for m, cin in MODS:
        net = get_model('fcn')(m, desc, m, num_units=U, batch_normalization=False, batchsize=2, device=str(device),
                               conv_dtype='fp32')
        net.variables.update({k: v for k, v in variables.items() if k.startswith(m + '/')})
        net._variables_changed()
        out[m] = net.predict(sub)
        out[m + '_score'] = net.predict(sub, output_attr='score')
    bayes = get_model('bayes_fusion')(confusion_matrices=cms, prefixes={'rgb': 'rgb', 'depth': 'depth'},
                                      data_description=desc, num_units=U, num_channels={'rgb': 3, 'depth': 1},
                                      expert_model='fcn', class_prior='data', batchsize=2, device=str(device),
                                      conv_dtype='fp32')
    bayes.variables.update(variables)
    bayes._variables_changed()
    out['bayes'] = bayes.predict(sub)
    return out


def compare_exact(exact, ref, n_images):
    """Label maps of the float32 HIP path against the fp32 oracle's: pixels that differ (fp32 summation order at exact
    near-ties), and the worst logit difference relative to the logit scale."""
    res = {'images': n_images}
    for k in ('rgb', 'depth', 'bayes'):
        a, b = exact[k], ref[k][:n_images]
        res[k] = {'pixels': int(a.size), 'differing_pixels': int((a != b).sum())}
    for m, _ in MODS:
        s, r = exact[m + '_score'], ref[m + '_score'][:n_images]
        res[m]['logit_rel_err'] = float(np.abs(s - r).max() / np.abs(r).max())
    return res


def hip_predictions(variables, measure, heldout, device='cuda'):
    """HIP path: expert confusion matrices and the Dirichlet fit on `measure` (the flows of
    experiments/bayes_fusion.py:146-195 and dirichlet_fusion.py:58-81), then every model's labels (and the experts'
    logits) on `heldout`."""
    from modular_semantic_segmentation_amd import get_model
    from modular_semantic_segmentation_amd.datasets.synthetic import data_description
    desc = data_description()
    out, cms = {}, {}
    for m, cin in MODS:
        net = get_model('fcn')(m, desc, m, num_units=U, batch_normalization=False, batchsize=4, device=str(device))
        net.variables.update({k: v for k, v in variables.items() if k.startswith(m + '/')})
        net._variables_changed()
        cms[m] = net.score(measure)[1]
        out[m] = net.predict(heldout)
        out[m + '_score'] = net.predict(heldout, output_attr='score')
    common = dict(data_description=desc, num_units=U, num_channels={'rgb': 3, 'depth': 1}, expert_model='fcn',
                  class_prior='data', batchsize=4, device=str(device))
    bayes = get_model('bayes_fusion')(confusion_matrices=cms, prefixes={'rgb': 'rgb', 'depth': 'depth'}, **common)
    bayes.variables.update(variables)
    bayes._variables_changed()
    out['bayes'] = bayes.predict(heldout)
    diri = get_model('dirichlet_fusion')(modalities=['rgb', 'depth'], sigma=1.0, delta=1e-2, beta=1e-2, **common)
    diri.variables.update(variables)
    diri._variables_changed()
    dparams = diri.fit(measure)
    out['dirichlet'] = diri.predict(heldout)
    return out, cms, dparams


def oracle_predictions(variables, heldout, cms, dparams, policy='fp32'):
    """The reference pipeline restated on the CPU (fp32), image by image: expert logits -> softmax -> argmax
    (basic_fusion_model.py:9-23), Bayes fusion of the labels (bayes_mix.py:12-58), Dirichlet fusion of the
    probabilities (dirichlet_mix.py:96-138) with the same tables.  Returns (predictions, seconds, images)."""
    from oracle import fcn_oracle as fo
    from oracle import fusion_oracle as fu
    n = len(heldout['labels'])
    out = {k: [] for k in ('rgb', 'depth', 'rgb_score', 'depth_score', 'bayes', 'dirichlet')}
    mats = [np.asarray(cms[m], np.float32).T for m, _ in MODS]
    prior = fu.dirichlet_prior(dparams['class_counts'], 'data')
    t0 = time.perf_counter()
    for i in range(n):
        probs, labs = [], []
        for m, _ in MODS:
            score = fo.fcn_forward(heldout[m][i:i + 1], variables, m, policy, keep=['score'])['score']
            p = fo.softmax(score)
            out[m + '_score'].append(score)
            probs.append(p)
            labs.append(fo.argmax_last(p))
            out[m].append(labs[-1])
        out['bayes'].append(np.argmax(fu.bayes_fusion(labs, mats, 'data')[0], -1))
        fused = fu.dirichlet_fusion([fu.renormalise(p) for p in probs],
                                    [np.asarray(dparams[m], np.float32) for m, _ in MODS], prior, 1.0)
        out['dirichlet'].append(np.argmax(fused, -1))
    dt = time.perf_counter() - t0
    return {k: np.concatenate(v) for k, v in out.items()}, dt, n


LOGIT_TOL_REL = 0.04        # stated tolerance of the bf16 path: max |logit error| / max |logit|


def compare(hip, ref, labels):
    """Per model: mIoU of both paths, the difference in percentage points, label agreement; per expert the logit
    error relative to the largest |logit| and the agreement on pixels whose fp32 top-2 margin exceeds twice the STATED
    logit tolerance (LOGIT_TOL_REL of the logit scale)."""
    res = {}
    for k in ('rgb', 'depth', 'bayes', 'dirichlet'):
        a, _ = _miou(labels, hip[k])
        b, _ = _miou(labels, ref[k])
        res[k] = {'miou_hip_bf16': round(a, 5), 'miou_fp32_oracle': round(b, 5),
                  'delta_miou_pp': round(100 * (a - b), 4), 'label_agreement': round(float((hip[k] == ref[k]).mean()), 6)}
    for m, _ in MODS:
        s, r = hip[m + '_score'], ref[m + '_score']
        scale = float(np.abs(r).max())
        err = float(np.abs(s - r).max())
        res[m]['logit_mean_abs_err_rel'] = round(float(np.abs(s - r).mean()) / scale, 7)
        top2 = np.sort(r, -1)[..., -2:]
        # a FIXED mask from the stated bf16 tolerance (worst logit within 4 % of the logit scale), not from the measured
        # error: a mask of 2 x the measured error would make the agreement below 1.0 by construction
        clear = (top2[..., 1] - top2[..., 0]) > 2 * LOGIT_TOL_REL * scale
        res[m].update(logit_max_abs_err=round(err, 5), logit_scale=round(scale, 3),
                      logit_rel_err=round(err / scale, 6), clear_margin_fraction=round(float(clear.mean()), 5),
                      label_agreement_clear_margin=round(float((hip[m][clear] == ref[m][clear]).mean()), 6))
        # what the fixed mask leaves out (it covers 0.1 % of the depth expert's pixels): agreement on the
        # COMPLEMENT of the mask, and a second mask from the MEASURED worst error -- a label can only differ where the fp32 top-2
        # margin is below the sum of the two logits' errors, so agreement there is 1.0 for any correct kernel and its size
        # says how much of the map the logit bound pins
        near = ~clear
        tight = (top2[..., 1] - top2[..., 0]) > 2 * err
        res[m].update(label_agreement_inside_margin=round(float((hip[m][near] == ref[m][near]).mean()) if near.any() else 1.0, 6),
                      measured_margin_fraction=round(float(tight.mean()), 5),
                      label_agreement_measured_margin=round(float((hip[m][tight] == ref[m][tight]).mean()) if tight.any() else 1.0, 6))
    return res


def run(h=384, w=768, steps=1500, batch=8, n_heldout=8, n_measure=16, device='cuda', log=None, fp8=True,
        learning_rate=1e-4, depth_unit=DEPTH_UNIT, exact_images=0):
    """The whole protocol; returns (accuracy dict, oracle seconds, oracle images)."""
    from modular_semantic_segmentation_amd.datasets.synthetic import make_rgbd_shapes
    variables, train = train_experts(h, w, steps, batch=batch, device=device, log=log, learning_rate=learning_rate,
                                     depth_unit=depth_unit)
    measure = {k: v[:n_measure] for k, v in train.items()}
    heldout = make_rgbd_shapes(n_heldout, h, w, seed=1001)
    hip, cms, dparams = hip_predictions(variables, measure, heldout, device=device)
    ref, dt, n = oracle_predictions(variables, heldout, cms, dparams)
    acc = compare(hip, ref, heldout['labels'])
    if exact_images:
        acc['exact_fp32'] = compare_exact(hip_predictions_exact(variables, heldout, cms, exact_images, device=device), ref,
                                          exact_images)
    if fp8:
        # 'fp8': the product's default -- the accuracy-guarded plan per expert; 'fp8_fixed_plan': round 5's one global plan
        for key, guarded in (('fp8', True), ('fp8_fixed_plan', False)):
            hip8 = hip_predictions_fp8(variables, measure, heldout, cms, device=device, guarded=guarded)
            acc[key] = {'plan': hip8['plan']}
            for k in ('rgb', 'depth', 'bayes'):
                a, _ = _miou(heldout['labels'], hip8[k])
                acc[key][k] = {'miou_hip_fp8': round(a, 5), 'delta_miou_pp_vs_fp32': round(100 * (a - acc[k]['miou_fp32_oracle']), 4),
                               'label_agreement_vs_fp32': round(float((hip8[k] == ref[k]).mean()), 6)}
    acc['protocol'] = ('experts trained %d Adam steps x %d images on procedural RGB-D shapes at %dx%d through the HIP fit(); '
                       '%d held-out images; same trained weights through the HIP bf16 path and the fp32 CPU oracle; '
                       'mean IoU over classes 1..%d (base_model.py:329)' % (steps, batch, w, h, n_heldout, C - 1))
    return acc, dt, n


if __name__ == '__main__':
    import json
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
    unit = float(sys.argv[3]) if len(sys.argv) > 3 else DEPTH_UNIT
    torch.set_num_threads(min(32, torch.get_num_threads()))
    acc, dt, n = run(steps=steps, learning_rate=lr, depth_unit=unit, log=lambda s: print(s, file=sys.stderr, flush=True))
    acc['oracle_seconds'] = round(dt, 2)
    print(json.dumps(acc, indent=1))
