"""north_star's accuracy clause on TRAINED weights: the HIP bf16 path and the fp32 oracle (the restated reference
graph) segment held-out 768x384 RGB-D images with the same experts; mean IoU (base_model.py:315-329) of every model
-- RGB expert, depth expert, Bayes fusion, Dirichlet fusion -- must agree within 0.1 percentage points, labels must
be identical wherever the fp32 top-2 logit margin exceeds twice the STATED logit tolerance (a fixed mask: 8 % of the
logit scale), and the logits must be
within 4 % of the logit scale at the worst of ~2e8 values, 0.3 % on average (the stated fp tolerance of the bf16 path at 13
conv layers).

The mIoU difference is a noisy statistic with zero mean: 0.02-0.3 % of the pixels (near-ties at object borders) flip
either way, and a rare class moves by a point of IoU on a few hundred pixels.  Over 20 runs with 8-12 held-out images
the difference had a standard deviation of 0.045 points (40 measurements); 96 held-out images put 0.1 points at six sigma.

(The file sorts last on purpose: the driver runs the suite with -x, and this is its only statistical test.)"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_miou_of_hip_path_matches_fp32_oracle_on_trained_experts():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from accuracy_evidence import run
    from tolerances import LOGIT_MEAN_TOL_VS_FP32, LOGIT_TOL_VS_FP32
    torch.set_num_threads(min(32, torch.get_num_threads()))
    acc, _, _ = run(h=384, w=768, steps=1500, batch=8, n_heldout=96, exact_images=8)
    print(json.dumps(acc))                   # shown by pytest on failure; kept next to the other GPU-box outputs
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, 'accuracy_test_%d.json' % os.getpid()), 'w') as f:
            json.dump(acc, f, indent=1)
    for m in ('rgb', 'depth'):
        assert acc[m]['miou_fp32_oracle'] > 0.6, (m, acc[m])              # a trained, useful expert
        # stated fp tolerance of the bf16 path (13 conv layers, bf16 storage, fp32 accumulation): the WORST of the ~2e8
        # logits within 4 % of the logit scale (measured 0.3-2.5 %), the mean error within 0.3 % (measured 0.03-0.1 %)
        assert acc[m]['logit_rel_err'] < LOGIT_TOL_VS_FP32['bf16'], (m, acc[m])
        assert acc[m]['logit_mean_abs_err_rel'] < LOGIT_MEAN_TOL_VS_FP32['bf16'], (m, acc[m])
        assert acc[m]['label_agreement_clear_margin'] == 1.0, (m, acc[m])
        assert acc[m]['clear_margin_fraction'] > 0, (m, acc[m])         # (the mask is not empty; its size is a property of the net:
        #                                                                  0.3 % of the depth expert's pixels, 97 % of the RGB expert's)
        # ... so the fixed mask alone says little about the depth expert.  Beside it: labels identical on every pixel whose
        # fp32 margin exceeds twice the MEASURED worst logit error (true for any kernel whose logits are that close; the mask
        # must cover most of the map for the logit bound to mean something), and agreement on the complement of the fixed mask
        assert acc[m]['label_agreement_measured_margin'] == 1.0, (m, acc[m])
        assert acc[m]['measured_margin_fraction'] > 0.5, (m, acc[m])
        assert acc[m]['label_agreement_inside_margin'] > 0.97, (m, acc[m])      # (measured 0.992 RGB -- its mask leaves out only
        #                                                                           the 5 % nearest ties -- and 0.999 depth)
    for m in ('rgb', 'depth', 'bayes', 'dirichlet'):
        assert abs(acc[m]['delta_miou_pp']) <= 0.1, (m, acc[m])
        assert acc[m]['label_agreement'] > 0.995, (m, acc[m])
    # the fp8 conv path (config 5) on the same trained weights.  Round 6: the plan is chosen PER EXPERT by calibrate() from the
    # label agreement with the bf16 graph on the calibration batch (FcnEngine.calibrate_guarded: bound 0.995, deepest e4m3
    # plan that passes, bf16 operands for an expert none serves) -- north_star's bound (0.1 points of mIoU) then holds for the
    # fp8 configuration as for the bf16 one, and the agreement bound is the bf16 path's.  The weak depth expert (thresholds on
    # one raw uint16 channel: 1.1-1.3 % of its pixels change label under ANY e4m3 plan) is expected to fall back to bf16, the
    # RGB expert to keep the deepest plan; whatever is chosen must meet the bounds.
    plans = acc['fp8']['plan']
    for m in ('rgb', 'depth'):
        assert plans[m]['chosen'] in ('conv2_2', 'conv3_1', 'conv4_1', 'conv5_1', 'bf16'), plans
        assert plans['bayes'][m]['chosen'] == plans[m]['chosen'], plans          # the fusion model calibrates to the same plans
        if plans[m]['chosen'] != 'bf16':
            assert plans[m]['agreement'][plans[m]['chosen']] >= 0.995, plans
    for m in ('rgb', 'depth', 'bayes'):
        assert abs(acc['fp8'][m]['delta_miou_pp_vs_fp32']) <= 0.1, (m, acc['fp8'][m], plans)
        assert acc['fp8'][m]['label_agreement_vs_fp32'] > 0.995, (m, acc['fp8'][m], plans)
    # round 5's ONE global plan (e4m3 operands from conv2_2 on for both experts; still available as fp8_agreement=0) beside
    # it, at round 5's bounds: RGB +0.01 .. -0.07, Bayes +0.03 .. -0.04, depth 0.00 .. -0.53 over six trainings, agreement
    # 0.987-0.989 for depth -- the loss the guard exists to refuse.
    for m in ('rgb', 'depth', 'bayes'):
        fx = acc['fp8_fixed_plan'][m]
        assert abs(fx['delta_miou_pp_vs_fp32']) < (1.0 if m == 'depth' else 0.5), (m, fx)
        assert fx['label_agreement_vs_fp32'] > (0.97 if m == 'depth' else 0.99), (m, fx)
    # conv_dtype='fp32' (the graph in plain float32 through csrc/exact_f32.hip) on the same TRAINED weights: label maps equal to
    # the fp32 oracle's at 768x384 up to fp32 summation order (a pixel can differ only where two logits tie to ~1e-6 of
    # the logit scale) -- so the 0.02-0.3 % of pixels the bf16 path flips are lost to bf16 storage, not to a kernel
    ex = acc['exact_fp32']
    for m in ('rgb', 'depth', 'bayes'):
        assert ex[m]['differing_pixels'] <= 2e-6 * ex[m]['pixels'], (m, ex[m])
    for m in ('rgb', 'depth'):
        assert ex[m]['logit_rel_err'] < LOGIT_TOL_VS_FP32['fp32'], (m, ex[m])
    # the fusion has something to gain on this task (fusion above both experts)
    assert acc['bayes']['miou_fp32_oracle'] > min(acc['rgb']['miou_fp32_oracle'], acc['depth']['miou_fp32_oracle'])
