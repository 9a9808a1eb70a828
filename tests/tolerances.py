"""The logit tolerances of the test-suite, stated ONCE (DESIGN.md section 4 quotes this table).

north_star: "argmax label maps bit-exact, logits within a stated fp tolerance" against the reference's float32 path.  Two
different quantities are bounded in the tests, and they must not be confused:

LOGIT_TOL_VS_FP32 -- THE STATED TOLERANCE of an arithmetic type: the worst |logit difference| / max|logit| between the HIP path
    and the reference graph in float32 (the fp32 oracle), on trained experts at 768x384 (~2e8 logits per expert):
      'fp32'  1e-5   only fp32 summation order separates the two (measured 4e-7 .. 1.4e-6); labels identical up to exact ties
      'bf16'  4e-2   13 conv layers of bf16 storage, fp32 accumulation (measured 0.2 .. 2.5 %); the MEAN error is bounded by
                     LOGIT_MEAN_TOL_VS_FP32 (measured 0.03 .. 0.1 %)
    (conv_dtype='fp8' has no logit bound against fp32 -- 3-bit mantissas move individual logits by tens of per cent of the
    scale on near-degenerate pixels; its contract is the mIoU / label-agreement bound of tests/test_zz_accuracy_gpu.py.)

LOGIT_TOL_VS_POLICY -- an IMPLEMENTATION check, not a stated tolerance: a kernel path against the oracle that rounds at the same
    points (policy 'bf16' / 'fp8'); what remains is fp32 summation order moving values across rounding boundaries:
      'bf16'  1.8e-2 (measured 1.2e-2 at 768x384 on random-init weights; the bound the full-size test has always had)
      'fp8'   0.2    (the maximum over 3.5 M logits of a chaotic quantity; the layer-by-layer comparison is the strict one)
"""
LOGIT_TOL_VS_FP32 = {'fp32': 1e-5, 'bf16': 4e-2}
LOGIT_MEAN_TOL_VS_FP32 = {'bf16': 3e-3}
LOGIT_TOL_VS_POLICY = {'bf16': 1.8e-2, 'fp8': 0.2}
