"""ctypes binding of the gfx950 C-ABI library (include/lvae.h).

The product path is `load()`: it opens csrc/liblvae_hip.so (building it with hipcc when the file is
missing) and raises if that is impossible -- there is NO CPU fallback.  `bind()` only attaches argtypes
to an already opened CDLL; the GPU-less CI uses it on the emulator build under tests/emu/ to check the
same kernel sources (tests only; nothing in this package opens that library).
"""
import ctypes
import os
import threading

from . import build as _build

_vp = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_long
_f = ctypes.c_float
_u64 = ctypes.c_uint64

# name -> argtypes (all functions return int status: 0 ok, >0 hipError_t, <0 argument check)
SIGNATURES = {
    "lv_gemm_f32": [_i, _i, _i, _i, _i, _f, _vp, _l, _vp, _l, _vp, _l, _i, _vp, _l, _i, _vp, _l, _i, _vp, _l, _vp],
    "lv_gemm_bf16": [_i, _i, _i, _i, _i, _f, _vp, _l, _vp, _l, _vp, _l, _i, _vp, _l, _i, _vp, _l, _i, _vp, _l, _vp],
    "lv_gemm_b16": [_i, _i, _i, _i, _f, _vp, _l, _vp, _l, _vp, _l, _i, _vp, _l, _i, _vp, _l, _i, _vp, _l, _vp],
    "lv_gemm_h16": [_i, _i, _i, _f, _vp, _l, _vp, _l, _vp, _l, _i, _vp, _l, _i, _vp, _l, _i, _vp, _l, _vp],
    "lv_gemm_b16_dual_supported": [_i, _i, _i, _l],
    "lv_gemm_b16_dual": [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _i, _vp, _l, _vp, _l, _vp],
    "lv_gemm_b16_pair_supported": [_i, _i, _i, _i, _i, _i, _i, _i, _l],
    "lv_gemm_b16_pair_pending": [_vp, _vp],
    "lv_gemm_b16_pair": [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _i, _vp, _l,
                         _i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp],
    "lv_gemm_b16_keep": [_i, _i, _i, _vp, _l, _vp, _l, _vp, _vp, _f, _i, _vp, _l, _vp],
    "lv_gemm_b16_sumsq_parts": [_i, _i, _i, _l],
    "lv_gemm_b16_sumsq": [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _vp, _i, _vp],
    "lv_gemm_b16_nll_parts": [_i],
    "lv_gemm_b16_tile": [_i, _i, _i, _i, _i, _f, _vp, _l, _vp, _l, _vp, _l, _i, _vp, _l, _i, _vp, _l, _i, _vp, _l, _vp],
    "lv_gemm_b16_nll_tile": [_i, _i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _i, _i, _vp, _vp, _vp],
    "lv_gemm_b16_nll": [_i, _i, _i, _vp, _l, _vp, _l, _vp, _l, _vp, _l, _i, _i, _vp, _vp, _vp],
    "lv_softmax_nll_merge_f32": [_vp, _i, _vp, _vp, _vp, _i, _vp],
    "lv_softmax_nll_bwd_h16": [_vp, _l, _vp, _vp, _l, _i, _vp, _vp, _l, _i, _i, _i, _vp],
    "lv_cvt_bf16_f32": [_vp, _l, _i, _i, _vp, _l, _vp, _l, _vp],
    "lv_cvt_bf16_keep_f32": [_vp, _l, _i, _i, _i, _vp, _f, _vp, _l, _vp, _l, _vp],
    "lv_cvt_f32_bf16_scaled": [_vp, _l, _f, _vp, _vp],
    "lv_keep_scale_f32": [_vp, _vp, _f, _i, _i, _i, _vp],
    "lv_cvt_bf16_gates_f32": [_vp, _l, _i, _i, _vp, _l, _vp, _l, _vp],
    "lv_cvt_bf16_lo_f32": [_vp, _l, _i, _i, _i, _vp, _l, _i, _i, _vp, _l, _vp, _l, _vp],
    "lv_cvt_h16_f32": [_vp, _l, _i, _i, _i, _vp, _l, _i, _i, _vp, _l, _vp, _l, _vp],
    "lv_gate_interleave_f32": [_vp, _vp, _i, _i, _vp, _vp],
    "lv_lstm_fwd_bf16_ug": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp],
    "lv_lstm_fwd_f32_ug": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp],
    "lv_loss_assemble_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "lv_loss_assemble_rng_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _u64, _vp],
    "lv_enc_head_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lv_enc_head_bwd_f32": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lv_dec_tail_parts": [_i],
    "lv_dec_init_f32": [_vp, _vp, _vp, _l, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lv_dec_tail_bwd_f32": [_vp, _vp, _vp, _vp, _l, _i, _vp, _vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lv_clip_norm2_f32": [_vp, _l, _vp, _l, _vp, _f, _vp, _vp, _vp, _vp],
    "lv_clip_norm2_txn_f32": [_vp, _l, _vp, _l, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "lv_clip_norm2_fold_txn_f32": [_vp, _l, _vp, _l, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "lv_clip_coef_txn_f32": [_vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "lv_txn_guard_f32": [_vp, _vp, _vp, _vp],
    "lv_sgd_step_txn_f32": [_vp, _vp, _l, _vp, _vp, _i, _vp, _vp],
    "lv_scale_txn_f32": [_vp, _l, _vp, _vp, _vp],
    "lv_sgd_step_scale_txn_f32": [_vp, _vp, _l, _vp, _vp, _i, _vp, _l, _vp, _vp],
    "lv_rng_noise_step": [_vp, _l, _vp, _l, _f, _vp, _l, _f, _vp, _u64, _vp],
    "lv_lstm_persist16_wpk_floats": [],
    "lv_lstm_persist16_xch_floats": [],
    "lv_lstm_persist16_xch_clear": [_vp, _vp],
    "lv_lstm_persist16_saved_floats": [_i, _i],
    "lv_lstm_persist16_pack": [_vp, _vp, _i, _i, _vp],
    "lv_lstm_persist16_import_saved": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lv_lstm_persist16_pack2": [_vp, _vp, _vp, _i, _vp],
    "lv_lstm_persist16_pack2_h16": [_vp, _vp, _vp, _i, _vp],
    "lv_lstm_fwd_bf16_persist16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "lv_lstm_bwd_bf16_persist16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lv_transpose_f32": [_vp, _vp, _i, _i, _vp],
    "lv_transpose_ld_f32": [_vp, _l, _vp, _l, _i, _i, _vp],
    "lv_lstm_bwd_ksplit": [_i],
    "lv_lstm_ws_floats": [_i, _i],
    "lv_lstm_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp],
    "lv_lstm_bwd_f32": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lv_lstm_fwd_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _i, _vp],
    "lv_lstm_bwd_bf16": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lv_lstm_bwd_bf16_img": [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lv_embed_gather_f32": [_vp, _vp, _l, _vp, _f, _vp, _i, _i, _i, _i, _vp],
    "lv_embed_gather_b16": [_vp, _vp, _l, _vp, _f, _i, _i, _i, _i, _vp, _l, _vp, _l, _vp],
    "lv_token_sort": [_vp, _l, _i, _i, _i, _vp, _vp, _vp, _vp],
    "lv_embed_scatter_f32": [_vp, _vp, _f, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "lv_embed_scatter_full_f32": [_vp, _vp, _f, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "lv_embed_scatter_sumsq_parts": [_i, _i],
    "lv_embed_scatter_full_sumsq_f32": [_vp, _vp, _f, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp],
    "lv_rows_merge_f32": [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp],
    "lv_reparam_kl_fwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lv_reparam_kl_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lv_softmax_nll_fwd_f32": [_vp, _l, _vp, _l, _i, _vp, _vp, _i, _i, _i, _vp],
    "lv_softmax_nll_bwd_f32": [_vp, _l, _vp, _vp, _l, _i, _vp, _i, _i, _i, _vp],
    "lv_softmax_nll_bwd_b16": [_vp, _l, _vp, _vp, _l, _i, _vp, _vp, _l, _i, _i, _i, _vp],
    "lv_vae_loss_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "lv_loss_bwd_scales_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "lv_tanh_f32": [_vp, _vp, _l, _vp],
    "lv_colsum_f32": [_vp, _l, _i, _i, _vp, _vp, _vp],
    "lv_add_f32": [_vp, _vp, _vp, _l, _vp],
    "lv_sumsq_workspace_floats": [],
    "lv_sumsq_f32": [_vp, _l, _vp, _vp, _i, _vp],
    "lv_clip_coef_f32": [_vp, _f, _vp, _vp, _vp],
    "lv_sgd_step_f32": [_vp, _vp, _l, _vp, _vp, _i, _vp],
    "lv_scale_f32": [_vp, _l, _vp, _vp],
    "lv_adam_step_f32": [_vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _f, _f, _f, _i, _vp],
    "lv_add_scalar_f32": [_vp, _f, _vp],
    "lv_sum_accum_f32": [_vp, _l, _vp, _vp],
    "lv_gauss_logpdf_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lv_logsumexp_rows_f32": [_vp, _l, _i, _i, _f, _vp, _vp],
    "lv_calc_mi_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lv_au_accum_f32": [_vp, _vp, _vp, _i, _i, _vp],
    "lv_argmax_rows_f32": [_vp, _l, _i, _i, _vp, _vp],
    "lv_log_softmax_rows_f32": [_vp, _l, _i, _i, _vp, _vp, _l, _vp],
    "lv_sample_rows_f32": [_vp, _l, _i, _i, _vp, _vp, _vp],
    "lv_rng_normal_f32": [_vp, _l, _vp, _u64, _vp],
    "lv_rng_keepmask_u8": [_vp, _l, _f, _vp, _u64, _vp],
    "lv_rng_advance": [_vp, _u64, _vp],
    "lv_rng_bernoulli_f32": [_vp, _vp, _l, _vp, _u64, _vp],
    "lv_im2col_f32": [_vp, _vp, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "lv_col2im_f32": [_vp, _l, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "lv_conv_pack_w_f32": [_vp, _vp, _i, _i, _i, _vp],
    "lv_conv_unpack_dw_f32": [_vp, _vp, _i, _i, _i, _i, _vp],
    "lv_conv32_wpack_floats": [_i],
    "lv_conv32_pack_f32": [_vp, _vp, _i, _i, _i, _vp],
    "lv_conv32_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "lv_conv32_wgrad_slabs": [_i, _i],
    "lv_conv32_wgrad_ws_floats": [_i, _i],
    "lv_conv32_wgrad_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lv_conv1x1_f32": [_vp, _vp, _vp, _l, _i, _i, _i, _i, _vp],
    "lv_conv1x1_wgrad_ws_floats": [_i, _i],
    "lv_conv32_pack_b16": [_vp, _vp, _i, _i, _i, _vp],
    "lv_conv32_b16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "lv_conv32_bnbwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "lv_conv1x1_bnbwd_f32": [_vp, _vp, _vp, _vp, _l, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "lv_bn_bwd_apply_partials_f32": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _l, _i, _vp],
    "lv_conv32_wgrad_b16": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "lv_conv32_wgrad_parts": [_i, _i],
    "lv_conv1x1_wgrad_parts": [_l],
    "lv_wgrad_reduce_batched": [_vp, _i, _vp],
    "lv_conv32_blocks": [_i],
    "lv_conv32_bnstat_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "lv_conv1x1_blocks": [_l],
    "lv_conv1x1_bnstat_f32": [_vp, _vp, _vp, _vp, _l, _i, _i, _vp],
    "lv_bn_fwd_partials_f32": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _i, _l, _i, _vp],
    "lv_conv1x1_wgrad_f32": [_vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp],
    "lv_mul_inplace_f32": [_vp, _vp, _l, _vp],
    "lv_bn_workspace_floats": [_i],
    "lv_pixelcnn_net_words": [],
    "lv_pixelcnn_block_words": [],
    "lv_pixelcnn_pixel_step_f32": [_vp, _i, _i, _i, _vp],
    "lv_conv32_tap_split": [_i],
    "lv_bn_eval_f32": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp, _l, _i, _vp],
    "lv_bn_fwd_f32": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _l, _i, _vp],
    "lv_bn_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _l, _i, _vp],
    "lv_bn_bwd2_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _l, _i, _vp],
    "lv_bn_bwd4_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _l, _i, _vp],
    "lv_sigmoid_bce_fwd_f32": [_vp, _vp, _vp, _i, _i, _f, _vp],
    "lv_sigmoid_bce_bwd_f32": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp],
    "lv_dec_input_fwd_f32": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "lv_dec_input_bwd_f32": [_vp, _vp, _i, _i, _i, _vp],
}


_LONG_FNS = ("lv_lstm_ws_floats", "lv_conv1x1_wgrad_ws_floats", "lv_conv1x1_blocks", "lv_lstm_persist16_wpk_floats", "lv_lstm_persist16_xch_floats", "lv_lstm_persist16_saved_floats", "lv_conv32_wpack_floats",
             "lv_conv32_wgrad_ws_floats")


class LvaeError(RuntimeError):
    pass


class Lib(object):
    """Thin checked wrapper: lib.lv_xxx(*args) raises LvaeError on a non-zero status."""

    def __init__(self, cdll, path):
        self.cdll = cdll
        self.path = path
        missing = []
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(cdll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.argtypes = argtypes
            fn.restype = ctypes.c_long if name in _LONG_FNS else ctypes.c_int
            setattr(self, "_raw_" + name, fn)
        if missing:
            raise LvaeError("%s does not export: %s" % (path, ", ".join(missing)))
        # functions that return a value rather than a status
        self._value_fns = {"lv_gemm_b16_dual_supported", "lv_gemm_b16_pair_supported", "lv_lstm_bwd_ksplit", "lv_dec_tail_parts", "lv_gemm_b16_nll_parts", "lv_gemm_b16_sumsq_parts", "lv_embed_scatter_sumsq_parts", "lv_conv32_wpack_floats",
                           "lv_conv32_wgrad_slabs", "lv_conv32_wgrad_ws_floats", "lv_conv32_wgrad_parts", "lv_conv1x1_wgrad_parts", "lv_conv32_blocks", "lv_conv1x1_blocks", "lv_conv1x1_wgrad_ws_floats", "lv_sumsq_workspace_floats", "lv_lstm_ws_floats", "lv_bn_workspace_floats",
                           "lv_lstm_persist16_wpk_floats", "lv_lstm_persist16_xch_floats", "lv_lstm_persist16_saved_floats",
                           "lv_pixelcnn_net_words", "lv_pixelcnn_block_words", "lv_conv32_tap_split"}

    def __getattr__(self, name):
        if name.startswith("lv_"):
            raw = object.__getattribute__(self, "_raw_" + name)
            if name in object.__getattribute__(self, "_value_fns"):
                return raw

            def call(*args):
                rc = raw(*args)
                if rc != 0:
                    raise LvaeError("%s failed with status %d (%s)" % (
                        name, rc, "hipError_t" if rc > 0 else "argument check"))
            call.__name__ = name
            self.__dict__[name] = call
            return call
        raise AttributeError(name)


def bind(cdll, path="<cdll>"):
    return Lib(cdll, path)


_lock = threading.Lock()
_lib = None


def load():
    """Open the gfx950 library; build it first if the .so is absent.  Never falls back to a CPU path."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _build.LIB_PATH
        if not os.path.exists(path):
            try:
                _build.build_hip()
            except Exception as e:  # noqa
                raise LvaeError(
                    "HIP extension %s is missing and could not be built (%s). The MI355X path has no CPU "
                    "fallback: run `python -m vae_lagging_encoder_amd.build` where hipcc is available." % (path, e))
        try:
            cdll = ctypes.CDLL(path)
        except OSError as e:
            raise LvaeError("cannot load %s: %s" % (path, e))
        _lib = Lib(cdll, path)
        return _lib
