"""ctypes binding of librmcl_hip.so (C ABI: include/rmcl.h).  No fallback: a missing or stale
library raises immediately."""
import ctypes as C
import os

# torch bundles its own libamdhip64; it MUST be loaded first so that librmcl_hip.so (NEEDED
# libamdhip64.so.7) binds to that same runtime instance.  Two HIP runtimes in one process leave the
# second without a device ("no ROCm-capable device is detected").
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RMCL_LIB") or os.path.join(_HERE, "lib", "librmcl_hip.so")   # (RMCL_LIB: A/B runs of two builds in one call)

F32, BF16 = 0, 1
MODE_INFER, MODE_DATA, MODE_FULL = 0, 1, 2
MODE_CLS_TAIL = 16        # include/rmcl.h RMCL_MODE_CLS_TAIL
PGD_DELTA_ZERO, PGD_SUM_PREV = 1, 2   # include/rmcl.h RMCL_PGD_*
HEADS_NO_WGRAD = 1                    # include/rmcl.h RMCL_HEADS_NO_WGRAD
EPI_BIAS, EPI_GELU, EPI_SAVE_PREACT, EPI_RESIDUAL, EPI_DGELU, EPI_ATOMIC, EPI_ACCUM, EPI_TANH = 1, 2, 4, 8, 16, 32, 64, 128
EPI_LNFOLD, EPI_ROWSTAT = 2048, 4096


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("B", "L", "P", "D", "H", "layers", "mlp", "patch_k", "proj", "vocab", "dtype", "exact", "Pp")]


class Ragged(C.Structure):
    """include/rmcl.h rmcl_ragged: selection of a zero-padded batch of smaller images (device pointers)."""
    _fields_ = [("sel", C.c_void_p), ("counts", C.c_void_p), ("hw", C.c_void_p), ("sel_ld", C.c_int), ("gw", C.c_int),
                ("G0", C.c_int), ("pos_tok", C.c_void_p), ("dpos_tok", C.c_void_p)]


LAYOUT_FIELDS = ("word", "pos", "btype", "eln_w", "eln_b", "vtype", "cls", "pos_img", "patch_w", "patch_b",
                 "layer0", "layer_stride", "ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "ln2_w", "ln2_b",
                 "fc1_w", "fc1_b", "fc2_w", "fc2_b", "norm_w", "norm_b", "mh0_w", "mh0_b", "mh1_w", "mh1_b", "mh3_w",
                 "ema_end", "pool_w", "pool_b", "itm_w", "itm_b", "total")


class Layout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in LAYOUT_FIELDS]


class Fold(C.Structure):
    """include/rmcl.h rmcl_fold: LayerNorm-folded weights W' = W * gamma (bf16) and the s / c vectors of one arena."""
    _fields_ = [("wf", C.c_void_p), ("sc", C.c_void_p)]


class BtHead(C.Structure):
    """include/rmcl.h rmcl_bt_head: widths of the BarlowTwinsHead and the arena offsets of its tensors."""
    _fields_ = [("D", C.c_int32), ("H1", C.c_int32), ("H2", C.c_int32), ("H3", C.c_int32),
                ("w1", C.c_int64), ("g1", C.c_int64), ("b1", C.c_int64), ("w2", C.c_int64), ("g2", C.c_int64), ("b2", C.c_int64),
                ("w3", C.c_int64)]


class RmclError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise RmclError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  rmcl_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.rmcl_last_error.restype = C.c_char_p
    for name in ("rmcl_stash_bytes", "rmcl_workspace_bytes", "rmcl_heads_stash_bytes", "rmcl_infonce_ws_bytes",
                 "rmcl_attention_scratch_elems", "rmcl_ln_fold_elems", "rmcl_bt_stash_floats", "rmcl_bt_loss_ws_floats"):
        getattr(lib, name).restype = C.c_int64
    return lib


lib = _load()

# every symbol include/rmcl.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = (
    "rmcl_last_error", "rmcl_version", "rmcl_prof_begin", "rmcl_prof_end", "rmcl_tune_set", "rmcl_grad_ready_wait", "rmcl_set_side_stream", "rmcl_set_prefetch_stream", "rmcl_dropout_mask_apply", "rmcl_param_layout", "rmcl_ln_fold_elems", "rmcl_ln_fold", "rmcl_linear_rowstat", "rmcl_linear_lnfold", "rmcl_linear_rowstat_c", "rmcl_linear_lnfold_c", "rmcl_weight_transpose_bf16", "rmcl_stash_bytes", "rmcl_workspace_bytes",
    "rmcl_heads_stash_bytes", "rmcl_im2patch_f32", "rmcl_patch_select", "rmcl_im2patch_sel", "rmcl_image_u8_to_patches", "rmcl_image_resize_u8", "rmcl_add_cast_f32", "rmcl_shard_sum", "rmcl_encoder_forward", "rmcl_encoder_backward",
    "rmcl_heads_forward", "rmcl_heads_forward2", "rmcl_heads_backward", "rmcl_infonce_ws_bytes", "rmcl_infonce_f32", "rmcl_infonce_split_bf16", "rmcl_pgd_step", "rmcl_pgd_step_fused",
    "rmcl_delta_channel_norm", "rmcl_ema_f32", "rmcl_enqueue_f32", "rmcl_cast_f32", "rmcl_adamw_f32", "rmcl_ipot_f32", "rmcl_gemm_batched", "rmcl_l2norm_rows_fwd", "rmcl_l2norm_rows_bwd",
    "rmcl_wpa_cost_finish", "rmcl_wpa_distance", "rmcl_itm_fwd", "rmcl_itm_bwd",
    "rmcl_gemm", "rmcl_gemm_chain", "rmcl_l2_prefetch_experiment", "rmcl_gemm_route", "rmcl_gemm_kblk", "rmcl_layernorm_fwd", "rmcl_layernorm_bwd", "rmcl_attention_scratch_elems", "rmcl_attention_fwd",
    "rmcl_attention_bwd",
    "rmcl_bt_stash_floats", "rmcl_bt_head_forward", "rmcl_bt_head_backward", "rmcl_bt_corr", "rmcl_bt_loss_ws_floats", "rmcl_bt_loss",
    "rmcl_bt_dz", "rmcl_bt_pair_metrics",
)


def check(rc, what=""):
    if rc != 0:
        msg = lib.rmcl_last_error()
        raise RmclError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def P(t):
    """device pointer of a torch tensor (or NULL)"""
    return C.c_void_p(0 if t is None else t.data_ptr())


def I64(v):
    return C.c_int64(int(v))


def F(v):
    return C.c_float(float(v))
