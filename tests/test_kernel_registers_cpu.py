"""Register budgets of the kernels whose first compilations went wrong in round 5 (tools/occupancy_scan.py): the compiler kept a
loop-invariant LDS table in registers (one wave per SIMD) or requested every load of an unrolled body up front (412
registers).  hipcc cross-compiles for gfx950 without a GPU; a regression shows here before it shows as a slow kernel."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))

# (file, mangled-name fragment, fewest waves per SIMD, most scratch bytes)
BUDGETS = [
    ('backward.hip', 'head_loss_kernelILi12ELb1E', 2, 0),                  # 253 registers; 412 before the taps were loaded once
    ('backward.hip', 'count_valid_kernel', 4, 0),
    ('pointwise.hip', 'fused_dirichlet_head_pk_kernelILi12ELi4E', 3, 0),   # 162 registers
    ('pointwise.hip', 'fused_head_kernelILi12ELi0ELb1ELi4E', 4, 0),        # Bayes: 98 registers (168 with per-pixel class sums)
    ('pointwise.hip', 'decoder_head_label4_kernelILi12E', 5, 0),
    ('fusion.hip', 'dirichlet_fuse_pk_kernelILi12E', 6, 0),                # 66 registers; 372 inside a grid-stride loop
    # round 6: the fused first pair carried 32 bytes of scratch in its RGB forms -- the masked tap gather of its edge-tile path
    # selected between array elements, which the compiler turned into an indexed load from a stack copy of the array
    ('conv_first_fused.hip', 'conv_first_pair_kernelILi3ELb0E', 2, 0),
    ('conv_first_fused.hip', 'conv_first_pair_kernelILi3ELb1E', 2, 0),
    ('conv_first_fused.hip', 'conv_first_pair_kernelILi1ELb0E', 2, 0),
    ('conv_wgrad.hip', 'conv_wgrad_lw_kernel', 2, 0),                      # 248 registers: 144 accumulators + the fragment rings
    # (36 bytes of true spills at the 128 registers a 1 024-thread workgroup allows: nine 64-bit load addresses per column
    # batch; a 60 us kernel)
    ('backward.hip', 'head_bwd_lowres_kernelILi12E', 4, 36),
    # round 6: the flat 1x1 filter gradient (all three forms) and the wide flat GEMM with its gathering modes: two 8-wave
    # workgroups' worth of registers (<= 128), no scratch
    ('conv_wgrad.hip', 'conv_wgrad_1x1_gemm_kernel', 4, 0),
    ('conv1x1_gemm.hip', 'conv1x1_gemm_wide_kernel', 4, 0),
]


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
def test_register_budgets_of_the_head_kernels():
    import occupancy_scan
    csrc = occupancy_scan.CSRC
    files = sorted({f for f, _, _, _ in BUDGETS})
    table = occupancy_scan.scan([os.path.join(csrc, f) for f in files], workers=len(files))
    for fname, frag, min_waves, max_scratch in BUDGETS:
        rows = [r for r in table[fname] if frag in r[0]]
        assert rows, 'no kernel matching %s in %s' % (frag, fname)
        for kern, regs, scratch, waves in rows:
            assert waves >= min_waves and scratch <= max_scratch, \
                '%s: %d waves per SIMD (%d registers), %d B scratch; budget: >= %d waves, <= %d B' % (
                    kern, waves, regs, scratch, min_waves, max_scratch)
