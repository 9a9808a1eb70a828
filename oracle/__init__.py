"""CPU oracle for the two-stream FCN + probabilistic-fusion hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()`
and the `cpu_baseline` leg of `bench.py` may import this package, and only as the
checker.  The product (`modular_semantic_segmentation_amd`) never imports it and
fails loudly when its HIP library is missing.

The oracle restates, op for op, the reference graph of
ethz-asl/modular_semantic_segmentation (paths relative to the reference root):

  xview/models/simple_fcn.py:10-170        encoder / decoder / fcn
  xview/models/custom_layers.py:8-25,71-139 bilinear kernel, deconv2d, conv2d
  xview/models/utils.py:43-53              cross_entropy
  xview/models/basic_fusion_model.py:9-23  test_pipeline (softmax + argmax)
  xview/models/bayes_mix.py:12-112         bayes_fusion, bayes_decision_matrix
  xview/models/dirichlet_mix.py:14-36,96-168 dirichlet_fusion, priors, sufficient statistics
  xview/models/base_model.py:136-162,315-329 confusion matrix, optimizers, score measures

The arithmetic of those graphs lives in TensorFlow 1.x, which is neither vendored nor
pinned by the reference (requirements.txt has no tensorflow line; README.md:37) and is
not installable here.  Its published op semantics are restated in
PyTorch-CPU fp32 (`fcn_oracle`) with an independent pure-numpy loop twin for small cases.

PINNING STATUS
  * pinned by reference-generated golden vectors (tests/golden/, made by
    tests/golden/make_golden.py from the importable numpy parts of the reference):
    bilinear kernels, bayes_decision_matrix LUTs, score() measures (notebook data),
    Dirichlet Newton fitter, npz variable-name schema.
  * conv / pool / deconv / softmax / Dirichlet log_prob arithmetic: **parity unpinned**
    by the reference's own tests (it has none that assert numbers); cross-checked here
    against the naive numpy twin and scipy.stats.dirichlet.
"""
