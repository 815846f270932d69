"""ctypes binding of libmliis_hip.so (the C ABI declared in include/mliis_hip.h).

The library is built in-tree (mliis_amd/libmliis_hip.so) by `__graft_entry__.build()` / `make -C mliis_amd/csrc`.
There is NO fallback: if the shared object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MLIIS_HIP_LIB: path of an alternative build of the library (kernel experiments); the default is the in-tree build
_DEFAULT_LIB = os.path.join(_HERE, "libmliis_hip.so")
LIB_PATH = os.environ.get("MLIIS_HIP_LIB", _DEFAULT_LIB)

_f = C.c_float
_i = C.c_int
_ll = C.c_longlong
_p = C.c_void_p
_sz = C.c_size_t

# name -> (restype, argtypes)
SIGNATURES = {
    "mliis_version": (_i, []),
    "mliis_last_error": (C.c_char_p, []),
    "mliis_stem_conv_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "mliis_stem_conv_fwd_stats_floats": (_sz, [_i, _i, _i, _i]),
    "mliis_stem_conv_fwd_stats": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _sz, _p, _p]),
    "mliis_stem_conv_bwd_filter_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "mliis_stem_conv_bwd_filter": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "mliis_dwconv_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _p]),
    "mliis_dwconv_bwd_data": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mliis_dwconv_bwd_data_bn": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _sz, _p, _p]),
    "mliis_dwconv_bwd_filter_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i]),
    "mliis_dwconv_bwd_filter": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "mliis_dwconv_bn_fwd_blocks": (_i, [_i, _i, _i, _i, _i, _i]),
    "mliis_dwconv_bn_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "mliis_dwconv_bn_bwd_blocks": (_i, [_i, _i, _i, _i, _i, _i]),
    "mliis_dwconv_bn_fwd": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _f, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _i, _i, _p]),
    "mliis_dwconv_bn_bwd": (_i, [_p] * 9 + [_i, _i, _i, _i, _i, _i, _p, _sz, _p, _sz, _p, _i, _i, _p]),
    "mliis_mbconv_dw_bwd_march": (_i, [_p] * 9 + [_i] + [_p] * 9 + [_i, _i, _i, _i, _i, _i, _p, _sz, _p, _sz, _p, _i, _i, _p]),
    "mliis_augment_stage": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mliis_rng_masks": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p]),
    "mliis_mbconv_dw_small_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "mliis_mbconv_dw_small_group_width": (_i, [_i, _i]),
    "mliis_mbconv_dw_fwd_small": (_i, [_p, _p, _i] + [_p] * 17 + [_i, _i, _i, _i, _i, _f, _f, _i, _i, _p, _i, _p]),
    "mliis_mbconv_dw_bwd_small": (_i, [_p] * 20 + [_i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "mliis_conv2d_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i]),
    "mliis_conv2d_plan": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "mliis_conv1x1_occupancy": (_i, [_i, _i, _i, _p]),
    "mliis_conv2d_kernel_name": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _p, _sz]),
    "mliis_conv2d_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _sz, _i, _f, _p, _i, _i, _p]),
    "mliis_head_ce_fused_supported": (_i, [_i, _i, _i, _i]),
    "mliis_head_ce_fused_workspace_floats": (_sz, [_i, _i, _i]),
    "mliis_head_ce_fused": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p, _p, _p, _sz, _p]),
    "mliis_conv2d_fwd_bnin_ok": (_i, [_i, _i, _i, _i, _i]),
    "mliis_conv2d_fwd_bnin": (_i, [_p, _i, _p, _i, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i,
                                   _f, _p, _i, _p]),
    "mliis_rsd_concat_pool_floats": (_sz, [_i, _i, _i, _i]),
    "mliis_rsd_concat_pool": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p, _sz, _p, _p]),
    "mliis_rsd_pool_fwd": (_i, [_p, _i, _f, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mliis_rsd_pool_bwd_workspace_floats": (_sz, [_i, _i]),
    "mliis_rsd_pool_bwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "mliis_transpose_weights": (_i, [_p, _p, _p, _i, _ll, _p, _p]),
    "mliis_weight_shadows": (_i, [_p, _p, _p, _i, _ll, _p, _p, _p, _i, _i, _p]),
    "mliis_weight_shadows_rng": (_i, [_p, _p, _p, _i, _ll, _p, _p, _p, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p]),
    "mliis_conv2d_bwd_data": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _i, _i, _i, _p]),
    "mliis_conv2d_bwd_data_bn": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _i, _p, _i, _p, _p, _p, _p, _sz, _p, _i, _i, _p]),
    "mliis_conv2d_bwd_data_gate": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _i, _p, _i, _p, _sz, _p, _i, _i, _p]),
    "mliis_x3_image_bytes": (_sz, [_i, _i, _i]),
    "mliis_x3_image_blocks": (_i, [_i, _i, _i]),
    "mliis_x3_pack_weights": (_i, [_p, _p, _p, _i, _i, _p]),
    "mliis_conv2d_x3_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i]),
    "mliis_conv2d_x3_plan": (_i, [_i, _i, _i, _i, _i, _i, _p]),
    "mliis_conv2d_fwd_x3": (_i, [_p, _i, _p, _sz, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _sz, _p]),
    "mliis_conv2d_bwd_data_x3": (_i, [_p, _i, _p, _sz, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p]),
    "mliis_conv2d_bwd_filter_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i]),
    "mliis_conv2d_bwd_filter": (_i, [_p, _i, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _i, _p]),
    "mliis_conv2d_bwd_filter_plan": (_i, [_i, _i, _i, _i, _i, _i, _p]),
    "mliis_conv2d_bwd_filter_batched": (_i, [_p, _i, _i, _i, _i, _i, _i, _p]),
    "mliis_colreduce_workspace_floats": (_sz, [_ll, _i, _i, _i]),
    "mliis_bn_stats": (_i, [_p, _i, _ll, _i, _i, _f, _f, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "mliis_bn_apply": (_i, [_p, _i, _p, _i, _ll, _i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _i, _p]),
    "mliis_bn_stats_partial": (_i, [_p, _i, _ll, _i, _i, _p, _sz, _p, _p]),
    "mliis_bn_apply_fused": (_i, [_p, _i, _p, _i, _ll, _i, _i, _p, _i, _f, _f, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _p, _sz, _p, _i, _p]),
    "mliis_bn_bwd": (_i, [_p, _i, _p, _i, _p, _i, _ll, _i, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p, _sz, _p, _sz, _p, _i, _i, _p]),
    "mliis_bn_apply_fused_pair": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _ll, _i, _f, _f, _i, _i, _i, _p]),
    "mliis_bn_bwd_pair": (_i, [_p] * 20 + [_i, _i, _i, _ll, _i, _i, _i, _sz, _p, _sz, _p]),
    "mliis_bn_bwd_dxsum_floats": (_sz, [_ll, _i]),
    "mliis_colsum": (_i, [_p, _i, _p, _i, _ll, _i, _i, _f, _p, _i, _p, _sz, _i, _p]),
    "mliis_se_mlp_fwd": (_i, [_p, _i, _f, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "mliis_se_mlp_bwd": (_i, [_p, _i] + [_p] * 13 + [_i, _i, _i, _i, _p]),
    "mliis_se_bn_bwd_sums_floats": (_sz, [_i, _i, _i]),
    "mliis_se_bn_bwd_sums": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _p, _i, _p]),
    "mliis_se_mlp_bwd_bn": (_i, [_p, _i] + [_p] * 9 + [_i, _i, _i, _i, _p]),
    "mliis_se_wgrad_batched": (_i, [_p, _i, _ll, _p]),
    "mliis_chan_affine": (_i, [_p, _i, _p, _p, _p, _i, _ll, _i, _i, _i, _p]),
    "mliis_chan_split": (_i, [_p, _i, _p, _p, _i, _i, _i, _p, _i, _i, _ll, _i, _i, _p]),
    "mliis_swish_mask_fwd": (_i, [_p, _i, _p, _i, _p, _i, _ll, _i, _i, _p]),
    "mliis_swish_mask_bwd": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _ll, _i, _i, _p]),
    "mliis_resize_bilinear_fwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mliis_resize_bilinear_bwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mliis_final_conv_fwd": (_i, [_p, _i, _p, _p, _p, _p, _ll, _i, _p]),
    "mliis_final_conv_bwd_data": (_i, [_p, _p, _p, _p, _i, _ll, _i, _p]),
    "mliis_final_conv_bwd_data_fin": (_i, [_p, _p, _p, _p, _i, _ll, _i, _p, _i, _i, _i, _i, _i, _f, _p, _p]),
    "mliis_final_conv_bwd_filter": (_i, [_p, _i, _p, _p, _ll, _i, _p, _p, _p, _sz, _p]),
    "mliis_softmax_ce_workspace_floats": (_sz, [_i, _i, _i]),
    "mliis_softmax_ce": (_i, [_p, _p, _p, _i, _i, _i, _f, _i, _f, _p, _p, _p, _p, _sz, _p]),
    "mliis_darc1": (_i, [_p, _i, _ll, _f, _p, _p, _p, _sz, _p]),
    "mliis_sgd_fused": (_i, [_p, _p, _p, _ll, _f, _p, _f, _f, _p]),
    "mliis_adam_b1zero_fused": (_i, [_p, _p, _p, _p, _ll, _f, _p, _f, _f, _f, _f, _p, _p, _p]),
    "mliis_axpby": (_i, [_f, _p, _f, _p, _ll, _p]),
    "mliis_lincomb": (_i, [_f, _p, _f, _p, _p, _ll, _p]),
    "mliis_copy_words": (_i, [_p, _p, _i, _p]),
    "mliis_fold_batched": (_i, [_p, _p, _p, _i, _ll, _p, _i, _ll, _p]),
    "mliis_fold_tile_outputs": (_i, []),
    "mliis_graph_begin_capture": (_i, [_p]),
    "mliis_graph_end_capture": (_i, [_p, C.POINTER(C.c_void_p)]),
    "mliis_graph_launch": (_i, [_p, _p]),
    "mliis_graph_destroy": (_i, [_p]),
}


class MliisError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        self._dll = None

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise MliisError(
                    "libmliis_hip.so not found at {} -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C mliis_amd/csrc`.  There is no CPU fallback.".format(LIB_PATH))
            # PyTorch-ROCm carries its own HIP runtime: import it FIRST so that this library resolves libamdhip64 to the copy torch
            # has already loaded.  Loaded the other way round, the process ends up with two HIP runtimes and the one this library is
            # bound to sees no device ("no ROCm-capable device is detected" at the first launch).
            import torch  # noqa: F401
            dll = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(dll, name)
                fn.restype = res
                fn.argtypes = args
            self._dll = dll
        return self._dll

    def raw(self, name):
        return getattr(self.load(), name)

    trace = None   # profiling aid (tools/site_times.py): a list that receives (entry point, args) of every call while it is set

    def call(self, name, *args):
        """Call an int-returning entry point; raise MliisError with the library's message on failure."""
        if self.trace is not None:
            self.trace.append((name, args))
        rc = getattr(self.load(), name)(*args)
        if rc != 0:
            msg = self._dll.mliis_last_error()
            raise MliisError("{} failed ({}): {}".format(name, rc, msg.decode() if msg else "?"))

    def size(self, name, *args) -> int:
        return int(getattr(self.load(), name)(*args))


lib = _Lib()
