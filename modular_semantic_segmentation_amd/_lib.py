"""ctypes binding of libxview_hip.so (the C ABI declared in include/xview_hip.h).

The library is the product's only compute path: if it is missing or a call fails this module
raises -- there is no CPU or PyTorch fallback.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libxview_hip.so')
CSRC = os.path.join(_HERE, 'csrc')


class XvError(RuntimeError):
    pass


class xv_act(ctypes.Structure):
    _fields_ = [('data', ctypes.c_void_p), ('n', ctypes.c_int32), ('h', ctypes.c_int32),
                ('w', ctypes.c_int32), ('c', ctypes.c_int32), ('dtype', ctypes.c_int32),
                ('scale_exp', ctypes.c_int32)]


class xv_pack_desc(ctypes.Structure):
    _fields_ = [('w_hwio', ctypes.c_void_p), ('packed', ctypes.c_void_p), ('packed_dgrad', ctypes.c_void_p),
                ('k', ctypes.c_int32), ('cin', ctypes.c_int32), ('cout', ctypes.c_int32), ('reserved', ctypes.c_int32)]


_vp, _i, _i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
_actp = ctypes.POINTER(xv_act)
_vpp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); every symbol include/xview_hip.h declares
SIGNATURES = {
    'xv_version': (_i, []),
    'xv_conv2d_choose_cfg': (_i, [_i] * 9),
    'xv_conv_first_pair_fwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _actp, _actp, _vp]),
    'xv_arch': (ctypes.c_char_p, []),
    'xv_source_hash': (ctypes.c_char_p, []),
    'xv_packed_weight_bytes': (ctypes.c_size_t, [_i, _i, _i]),
    'xv_pack_conv_weights': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'xv_pack_conv_weights_pair': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'xv_packed_weight_bytes_f8': (ctypes.c_size_t, [_i, _i, _i]),
    'xv_pack_conv_weights_f8': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'xv_conv2d_fwd': (_i, [_actp, _vp, _vp, _actp, _actp, _i, _i, _vp]),
    'xv_conv2d_fwd_cfg': (_i, [_actp, _vp, _vp, _actp, _actp, _i, _i, _i, _vp]),
    'xv_conv2d_fwd_pair': (_i, [_actp, _vp, _vp, _actp, _actp, _actp, _vp, _vp, _actp, _actp, _i, _vp]),
    'xv_conv2d_num_cfgs': (_i, []),
    'xv_conv2d_streamk_workspace_bytes': (ctypes.c_size_t, []),
    'xv_conv2d_fwd_ws': (_i, [_actp, _vp, _vp, _actp, _actp, _i, _i, _i, _vp, ctypes.c_size_t, _vp]),
    'xv_conv2d_split_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i, _i, _i]),
    'xv_conv2d_fwd_split': (_i, [_actp, _vp, _vp, _actp, _actp, _i, _i, _i, _vp, ctypes.c_size_t, _vp]),
    'xv_conv2d_bwd_data_ws': (_i, [_actp, _vp, _vp, _actp, _actp, _actp, _i, _vp, ctypes.c_size_t, _vp]),
    'xv_deconv_dense_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i, _i, _i]),
    'xv_deconv_dense_fwd': (_i, [_actp, _vp, _vp, _vp, _vp, _actp, _actp, _i, _i, _vp, ctypes.c_size_t, _vp]),
    'xv_conv2d_first_fwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _actp, _i, _vp]),
    'xv_conv2d_first_gather7s2_fwd': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _actp, _i, _vp]),
    'xv_maxpool2x2_fwd': (_i, [_actp, _actp, _vp]),
    'xv_upsample2x_relu_add': (_i, [_actp, _actp, _actp, _vp]),
    'xv_upsample2x_affine_relu_add': (_i, [_actp, _vp, _vp, _actp, _actp, _vp]),
    'xv_upsample2x_affine_act_add': (_i, [_actp, _vp, _vp, _actp, _actp, _i, _vp]),
    'xv_concat_channels': (_i, [_actp, _actp, _actp, _vp]),
    'xv_dropout': (_i, [_actp, _actp, ctypes.c_float, ctypes.c_uint64, _vp]),
    'xv_conv2d_fwd_residual': (_i, [_actp, _vp, _vp, _actp, _actp, _i, _vp]),
    'xv_subsample2': (_i, [_actp, _actp, _vp]),
    'xv_gather_conv7s2': (_i, [_actp, _actp, _vp]),
    'xv_im2col_dilated_pair': (_i, [_actp, _i, _i, _actp, _vp]),
    'xv_subsample2_bwd': (_i, [_actp, _actp, _vp]),
    'xv_gather_conv7s2_bwd': (_i, [_actp, _actp, _vp]),
    'xv_conv_dilated_pair_fwd': (_i, [_actp, _vp, _vp, _i, _i, _i, _actp, _vp]),
    'xv_im2col_dilated_pair_bwd': (_i, [_actp, _i, _i, _actp, _vp]),
    'xv_add': (_i, [_actp, _actp, _actp, _vp]),
    'xv_space_to_depth': (_i, [_actp, _i, _actp, _vp]),
    'xv_space_to_depth_dense': (_i, [_vp, _i, _i, _actp, _vp]),
    'xv_depth_to_space_dense': (_i, [_actp, _i, _i, _vp, _vp, _vp, _vp]),
    'xv_act_to_dense_f32': (_i, [_actp, _vp, _vp]),
    'xv_depth_to_space_dense_f32': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'xv_decoder_head_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i, _i]),
    'xv_decoder_head_fwd': (_i, [_actp, _vp, _vp, _i, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_decoder_head_affine_fwd': (_i, [_actp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'xv_softmax_argmax': (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    'xv_bayes_fuse': (_i, [_vpp, _i, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    'xv_bayes_fuse_lut': (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp]),
    'xv_dirichlet_fuse': (_i, [_vpp, _i, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    'xv_average_fuse': (_i, [_vpp, _i, _i, _i64, _vp, _vp]),
    'xv_pack_conv_weights_dgrad': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'xv_pack_conv_weights_multi': (_i, [_vp, _i, _vp]),
    'xv_memset_zero': (_i, [_vp, ctypes.c_size_t, _vp]),
    'xv_conv2d_bwd_data': (_i, [_actp, _vp, _vp, _actp, _actp, _actp, _i, _vp]),
    'xv_conv2d_bwd_filter': (_i, [_actp, _actp, _vp, _vp, _i, _vp]),
    'xv_set_wgrad_variant': (_i, [_i]),
    'xv_conv2d_bwd_filter_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i, _i, _i, _i]),
    'xv_conv2d_bwd_filter_ws': (_i, [_actp, _actp, _vp, _vp, _i, _vp, ctypes.c_size_t, _vp]),
    'xv_conv_dilated_pair_bwd_data': (_i, [_actp, _vp, _vp, _i, _i, _actp, _vp]),
    'xv_conv_dilated_pair_bwd_filter_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i, _i, _i]),
    'xv_conv_dilated_pair_bwd_filter_ws': (_i, [_actp, _actp, _i, _i, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_bias_grad': (_i, [_actp, _vp, _vp]),
    'xv_conv2d_first_bwd_filter': (_i, [_vp, _i, _i, _i, _i, _actp, _vp, _vp, _vp]),
    'xv_conv2d_first_bwd_filter_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i, _i]),
    'xv_conv2d_first_bwd_filter_ws': (_i, [_vp, _i, _i, _i, _i, _actp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_maxpool2x2_bwd': (_i, [_actp, _actp, _actp, _vp]),
    'xv_relu_bwd': (_i, [_actp, _actp, _actp, _vp]),
    'xv_upsample2x_bwd': (_i, [_actp, _actp, _actp, _vp]),
    'xv_count_valid_labels': (_i, [_vp, _i, _i64, _vp, _vp]),
    'xv_score_lowres': (_i, [_actp, _vp, _i, _vp, _vp]),
    'xv_fused_head_fwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'xv_decoder_head_bwd_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i, _i]),
    'xv_decoder_head_bwd': (_i, [_actp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _actp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_stats': (_i, [_actp, _vp, _vp]),
    'xv_bn_finalize': (_i, [_vp, _i, _i64, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'xv_bn_apply': (_i, [_actp, _vp, _vp, _i, _actp, _vp]),
    'xv_bn_apply_ups8': (_i, [_actp, _vp, _vp, _i, _actp, _vp]),
    'xv_score_dense_fwd_ups8': (_i, [_actp, _vp, _vp, _vp, _vp, _i, _actp, _vp, _vp]),
    'xv_bn_bwd': (_i, [_actp, _actp, _actp, _vp, _vp, _vp, _vp, _vp, _vp, _actp, _vp]),
    'xv_bn_bwd_reduce': (_i, [_actp, _actp, _actp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'xv_bn_bwd_apply': (_i, [_actp, _actp, _actp, _vp, _vp, _vp, _vp, _i64, _actp, _vp]),
    'xv_bn_bwd_reduce_zmask': (_i, [_actp, _actp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_bwd_reduce_zmask_ups8': (_i, [_actp, _actp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_workspace_bytes': (ctypes.c_size_t, [_i]),
    'xv_conv2d_stats_rows': (_i, []),
    'xv_conv2d_fwd_stats': (_i, [_actp, _vp, _vp, _actp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_sums_from_rows': (_i, [_vp, _i, _i, _vp, _vp]),
    'xv_conv2d_route_bytes': (ctypes.c_size_t, [_i, _i, _i, _i]),
    'xv_conv2d_fwd_route': (_i, [_actp, _vp, _vp, _actp, _vp, ctypes.c_size_t, _vp]),
    'xv_conv2d_bwd_data_route': (_i, [_actp, _vp, _vp, _vp, ctypes.c_size_t, _actp, _vp]),
    'xv_bn_finalize_from_rows': (_i, [_vp, _i, _i, _i64, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp]),
    'xv_bn_stats_finalize_ws': (_i, [_actp, _vp, _vp, ctypes.c_size_t, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp]),
    'xv_bn_stats_finalize_ups8_ws': (_i, [_actp, _vp, _vp, ctypes.c_size_t, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp]),
    'xv_bn_apply_pool': (_i, [_actp, _vp, _vp, _actp, _actp, _vp]),
    'xv_bn_pool_bwd_reduce': (_i, [_actp, _actp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_pool_bwd_apply': (_i, [_actp, _actp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _actp, _vp]),
    'xv_bn_stats_ws': (_i, [_actp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_stats_ups8_ws': (_i, [_actp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_bwd_reduce_ws': (_i, [_actp, _actp, _actp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_dense_stats_ws': (_i, [_vp, _i64, _i, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_dense_bwd_reduce_ws': (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_bn_bwd_apply_zmask': (_i, [_actp, _actp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _actp, _vp]),
    'xv_bn_bwd_apply_zmask_ups8': (_i, [_actp, _actp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _actp, _vp]),
    'xv_bn_dense_bwd_reduce': (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'xv_bn_dense_bwd_apply': (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'xv_bn_dense_stats': (_i, [_vp, _i64, _i, _vp, _vp]),
    'xv_bn_dense_apply': (_i, [_vp, _i64, _i, _vp, _vp, _vp, _vp]),
    'xv_bn_dense_bwd': (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'xv_upsample_raw_fwd': (_i, [_actp, _i, _actp, _vp]),
    'xv_upsample_raw_bwd': (_i, [_actp, _i, _actp, _vp]),
    'xv_upsample_raw_bwd_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i, _i]),
    'xv_upsample_raw_bwd_ws': (_i, [_actp, _i, _actp, _vp, ctypes.c_size_t, _vp]),
    'xv_score_dense_fwd': (_i, [_actp, _vp, _vp, _i, _vp, _vp]),
    'xv_softmax_ce_dense': (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    'xv_softmax_ce_dense_affine': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    'xv_softmax_ce_dense_workspace_bytes': (ctypes.c_size_t, [_i64]),
    'xv_softmax_ce_dense_ws': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    'xv_score_dense_bwd': (_i, [_actp, _vp, _vp, _i, _vp, _vp, _actp, _vp]),
    'xv_score_dense_bwd_workspace_bytes': (ctypes.c_size_t, [_i, _i, _i]),
    'xv_score_dense_bwd_ws': (_i, [_actp, _vp, _vp, _i, _vp, _vp, _actp, _vp, ctypes.c_size_t, _vp]),
    'xv_conv2d_f32': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'xv_conv2d_f32_pool': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'xv_conv2d_f32_scalar': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'xv_maxpool2x2_f32': (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    'xv_upsample2x_f32': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'xv_upsample2x_affine_f32': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'xv_decoder_head_affine_f32': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'xv_score_lowres_f32': (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    'xv_decoder_head_from_scores': (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'xv_adam_step': (_i, [_vp, _vp, _vp, _vp, _i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                          ctypes.c_float, _vp]),
    'xv_rmsprop_step': (_i, [_vp, _vp, _vp, _i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp]),
    'xv_adagrad_step': (_i, [_vp, _vp, _vp, _i64, ctypes.c_float, ctypes.c_float, _vp]),
    'xv_dirichlet_suffstats': (_i, [_vp, _vp, _i, _i64, _vp, _vp, _vp]),
    'xv_confusion_matrix': (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    'xv_narrow_labels': (_i, [_vp, _i64, _vp, _vp]),
}

_lib = None


def source_hash():
    """sha256 (first 16 hex digits) over the library's sources, as csrc/Makefile computes SRC_HASH."""
    import hashlib
    import re
    mk = open(os.path.join(CSRC, 'Makefile')).read()
    srcs = re.search(r'^SRCS := (.*)$', mk, re.M).group(1).split()
    h = hashlib.sha256()
    for rel in srcs + ['xv_common.h', '../../include/xview_hip.h', 'Makefile']:
        with open(os.path.join(CSRC, rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build(force=False):
    """Compile csrc/*.hip for gfx950 into libxview_hip.so (hipcc cross-compiles without a GPU).
    force=True (or XV_FORCE_REBUILD=1) recompiles every object from scratch; otherwise `make` rebuilds what
    changed.  Either way the result must carry the hash of the sources it sits next to."""
    if force or os.environ.get('XV_FORCE_REBUILD') == '1':
        subprocess.run(['make', '-C', CSRC, 'clean'], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(['make', '-C', CSRC, '-j4'], check=True, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise XvError('build did not produce ' + LIB_PATH)
    global _lib
    _lib = None
    got = lib().xv_source_hash().decode()
    if got != source_hash():
        raise XvError('built library is stamped %s but the sources hash to %s' % (got, source_hash()))
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises XvError if the library is absent or was built from
    other sources than the ones next to it (when those are present: they always are in this repository)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise XvError('%s not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                          '(hipcc --offload-arch=gfx950); there is no fallback path' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        if os.path.exists(os.path.join(CSRC, 'Makefile')) and os.environ.get('XV_ALLOW_STALE_LIB') != '1':
            got = handle.xv_source_hash().decode()
            if got != source_hash():
                raise XvError('%s was built from other sources (stamp %s, sources %s): rebuild with '
                              '`python -c "import __graft_entry__ as g; g.build()"`' % (LIB_PATH, got, source_hash()))
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        kind = {-1: 'XV_EINVAL', -2: 'XV_ESHAPE', -3: 'XV_EWORKSPACE'}.get(code, 'hipError_t %d' % code)
        raise XvError('%s failed: %s' % (what, kind))
