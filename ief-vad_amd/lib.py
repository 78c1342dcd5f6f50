"""ctypes binding of libiefvad.so (the C ABI declared in include/iefvad.h).

Loading never falls back to anything: if the shared library is missing or does not export a
symbol the header declares, `load_library()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IEFVAD_LIB") or os.path.join(_HERE, "libiefvad.so")      # IEFVAD_LIB: A/B builds of the same ABI (tools)
CSRC_DIR = os.path.join(_HERE, "csrc")

ABI_VERSION = 8
MAX_LAYERS = 8
MAX_STEPS = 64
NOISE_GAUSSIAN, NOISE_STUDENT_T = 0, 1
IN_F32, IN_F16, IN_BF16 = 0, 1, 2
COMPUTE_F32, COMPUTE_BF16, COMPUTE_BF16X6, COMPUTE_FP16X3 = 0, 1, 2, 3
COMPUTE_CODES = {"f32": COMPUTE_F32, "bf16": COMPUTE_BF16, "bf16x6": COMPUTE_BF16X6, "fp16x3": COMPUTE_FP16X3}

# every symbol include/iefvad.h declares
SYMBOLS = ["iefvad_abi_version", "iefvad_create", "iefvad_set_weights", "iefvad_workspace_bytes",
           "iefvad_forward", "iefvad_forward_timed", "iefvad_gemm_bias", "iefvad_split_bf16x3", "iefvad_split_bf16x3_many", "iefvad_last_error",
           "iefvad_destroy", "iefvad_comm_unique_id", "iefvad_comm_create", "iefvad_comm_nranks", "iefvad_comm_destroy",
           "iefvad_gather_scores", "iefvad_gather_plan", "iefvad_rccl_version", "iefvad_forward_videos",
           "iefvad_videos_workspace_bytes", "iefvad_host_gather", "iefvad_loss_forward", "iefvad_loss_backward", "iefvad_adamw_step",
           "iefvad_loss_workspace_bytes", "iefvad_train_workspace_bytes", "iefvad_train_forward", "iefvad_train_backward", "iefvad_nan_rule",
           "iefvad_forward_videos_host", "iefvad_host_gather_bf16", "iefvad_auc_ap", "iefvad_auc_ap_workspace_bytes", "iefvad_forward_scaled", "iefvad_rowblock_unit",
           "iefvad_similarity_adj", "iefvad_similarity_adj_workspace_bytes", "iefvad_distance_adj", "iefvad_gcn_forward",
           "iefvad_gcn_workspace_bytes", "iefvad_gat_forward", "iefvad_gat_workspace_bytes", "iefvad_resblock_forward",
           "iefvad_resblock_workspace_bytes", "iefvad_adamw_step_multi"]
COMM_ID_BYTES = 128

_fp = C.c_void_p  # device pointers travel as integers


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("embed_dim", C.c_int32), ("seq_len", C.c_int32),
                ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("num_steps", C.c_int32),
                ("noise_model", C.c_int32), ("compute", C.c_int32), ("lambda_ref", C.c_float),
                ("nu", C.c_float), ("epsilon", C.c_float), ("micro_batch", C.c_int32), ("graph_chunks", C.c_int32)]


class Weights(C.Structure):
    _fields_ = [("in_proj_w", (_fp * MAX_LAYERS) * 2), ("in_proj_b", (_fp * MAX_LAYERS) * 2),
                ("out_proj_w", (_fp * MAX_LAYERS) * 2), ("out_proj_b", (_fp * MAX_LAYERS) * 2),
                ("norm_w", (_fp * MAX_LAYERS) * 2), ("norm_b", (_fp * MAX_LAYERS) * 2),
                ("whiten_w", _fp * 2), ("whiten_b", _fp * 2),
                ("mu_w", _fp * 2), ("mu_b", _fp * 2), ("logvar_w", _fp * 2), ("logvar_b", _fp * 2),
                ("ref_w1", _fp * MAX_STEPS), ("ref_b1", _fp * MAX_STEPS),
                ("ref_w2", _fp * MAX_STEPS), ("ref_b2", _fp * MAX_STEPS),
                ("cls_w", _fp), ("cls_b", _fp)]


class WeightGrads(C.Structure):
    """iefvad_weight_grads: gradient destinations, field for field the layout of `Weights`."""
    _fields_ = Weights._fields_


class TrainOptions(C.Structure):
    _fields_ = [("dropout_p", (C.c_float * MAX_LAYERS) * 2), ("seed", C.c_uint64), ("keep_mask", _fp)]


class OutputGrads(C.Structure):
    _fields_ = [("fused", _fp), ("logits", _fp), ("image_mu", _fp), ("event_mu", _fp),
                ("image_logvar", _fp), ("event_logvar", _fp), ("w_i", _fp), ("w_e", _fp)]


class Outputs(C.Structure):
    _fields_ = [("fused", _fp), ("logits", _fp), ("image_mu", _fp), ("event_mu", _fp),
                ("image_logvar", _fp), ("event_logvar", _fp), ("w_i", _fp), ("w_e", _fp),
                ("w_i_mean", _fp), ("w_e_mean", _fp)]


class StageTimes(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("qkv_gemm_ms", C.c_float), ("attention_ms", C.c_float),
                ("out_gemm_ms", C.c_float), ("layernorm_ms", C.c_float), ("head_gemm_ms", C.c_float),
                ("fusion_ms", C.c_float), ("refine_gemm_ms", C.c_float), ("scorer_ms", C.c_float),
                ("cast_ms", C.c_float), ("gemm_launches", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def build_library(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libiefvad.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


class UnitIO(C.Structure):
    """iefvad_unit_io (include/iefvad.h)."""
    _fields_ = [("x", C.c_void_p * 2), ("resid", C.c_void_p * 2), ("y", C.c_void_p * 2), ("yb", C.c_void_p * 2), ("mu", C.c_void_p * 2),
                ("logvar", C.c_void_p * 2), ("w", C.c_void_p * 2), ("z", C.c_void_p), ("logits", C.c_void_p)]


UNIT_INPROJ, UNIT_OUTPROJ_LN, UNIT_HEADS, UNIT_REFINE = 0, 1, 2, 3


class ResblockWeights(C.Structure):
    """iefvad_resblock_weights (include/iefvad.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "ln_1_w", "ln_1_b", "ln_2_w", "ln_2_b", "c_fc_w",
                                          "c_fc_b", "c_proj_w", "c_proj_b")]


def load_library() -> C.CDLL:
    """dlopen libiefvad.so and declare prototypes.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(hipcc --offload-arch=gfx950); there is no CPU fallback for this path")
    lib = C.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise RuntimeError(f"libiefvad.so does not export {s}")
    lib.iefvad_abi_version.restype = C.c_int
    lib.iefvad_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.iefvad_create.restype = C.c_int
    lib.iefvad_set_weights.argtypes = [C.c_void_p, C.POINTER(Weights), C.c_void_p]
    lib.iefvad_set_weights.restype = C.c_int
    lib.iefvad_workspace_bytes.argtypes = [C.c_void_p, C.c_int32]
    lib.iefvad_workspace_bytes.restype = C.c_size_t
    lib.iefvad_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_size_t, C.POINTER(Outputs), C.c_void_p]
    lib.iefvad_forward.restype = C.c_int
    lib.iefvad_forward_scaled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.POINTER(Outputs), C.c_void_p]
    lib.iefvad_forward_scaled.restype = C.c_int
    lib.iefvad_forward_timed.argtypes = lib.iefvad_forward.argtypes + [C.POINTER(StageTimes)]
    lib.iefvad_forward_timed.restype = C.c_int
    lib.iefvad_videos_workspace_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32]
    lib.iefvad_videos_workspace_bytes.restype = C.c_size_t
    lib.iefvad_forward_videos.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.iefvad_forward_videos.restype = C.c_int
    lib.iefvad_forward_videos_host.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                                               C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.iefvad_forward_videos_host.restype = C.c_int
    lib.iefvad_loss_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.iefvad_loss_workspace_bytes.restype = C.c_size_t
    lib.iefvad_loss_forward.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float,
                                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.iefvad_loss_forward.restype = C.c_int
    lib.iefvad_loss_backward.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float] + [C.c_void_p] * 7
    lib.iefvad_loss_backward.restype = C.c_int
    lib.iefvad_train_workspace_bytes.argtypes = [C.c_void_p, C.c_int32]
    lib.iefvad_train_workspace_bytes.restype = C.c_size_t
    lib.iefvad_train_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(TrainOptions), C.c_void_p,
                                         C.c_size_t, C.POINTER(Outputs), C.c_void_p]
    lib.iefvad_train_forward.restype = C.c_int
    lib.iefvad_train_backward.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(OutputGrads), C.POINTER(WeightGrads),
                                          C.c_void_p]
    lib.iefvad_train_backward.restype = C.c_int
    lib.iefvad_adamw_step.argtypes = [C.c_void_p] * 4 + [C.c_size_t] + [C.c_double] * 5 + [C.c_int32, C.c_void_p]
    lib.iefvad_adamw_step.restype = C.c_int
    lib.iefvad_adamw_step_multi.argtypes = [C.c_void_p, C.c_int32, C.c_uint64] + [C.c_double] * 5 + [C.c_int32, C.c_void_p]
    lib.iefvad_adamw_step_multi.restype = C.c_int
    lib.iefvad_host_gather.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int64, C.c_int32]
    lib.iefvad_host_gather.restype = C.c_int
    lib.iefvad_host_gather_bf16.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int64, C.c_int32]
    lib.iefvad_host_gather_bf16.restype = C.c_int
    lib.iefvad_auc_ap_workspace_bytes.argtypes = [C.c_int64]
    lib.iefvad_auc_ap_workspace_bytes.restype = C.c_size_t
    lib.iefvad_auc_ap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.iefvad_auc_ap.restype = C.c_int
    lib.iefvad_rowblock_unit.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(UnitIO), C.c_void_p]
    lib.iefvad_rowblock_unit.restype = C.c_int
    i32, vp = C.c_int32, C.c_void_p
    lib.iefvad_similarity_adj_workspace_bytes.argtypes = [i32, i32, i32]
    lib.iefvad_similarity_adj_workspace_bytes.restype = C.c_size_t
    lib.iefvad_similarity_adj.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, C.c_size_t, vp]
    lib.iefvad_similarity_adj.restype = C.c_int
    lib.iefvad_distance_adj.argtypes = [i32, i32, vp, vp]
    lib.iefvad_distance_adj.restype = C.c_int
    lib.iefvad_gcn_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.iefvad_gcn_workspace_bytes.restype = C.c_size_t
    lib.iefvad_gcn_forward.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, C.c_size_t, vp]
    lib.iefvad_gcn_forward.restype = C.c_int
    lib.iefvad_gat_workspace_bytes.argtypes = [i32, i32]
    lib.iefvad_gat_workspace_bytes.restype = C.c_size_t
    lib.iefvad_gat_forward.argtypes = [vp, vp, vp, vp, C.c_float, i32, i32, i32, i32, vp, vp, C.c_size_t, vp]
    lib.iefvad_gat_forward.restype = C.c_int
    lib.iefvad_resblock_workspace_bytes.argtypes = [i32, i32, i32]
    lib.iefvad_resblock_workspace_bytes.restype = C.c_size_t
    lib.iefvad_resblock_forward.argtypes = [vp, C.POINTER(ResblockWeights), vp, vp, i32, i32, i32, i32, vp, vp, C.c_size_t, vp]
    lib.iefvad_resblock_forward.restype = C.c_int
    lib.iefvad_gemm_bias.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_void_p]
    lib.iefvad_gemm_bias.restype = C.c_int
    lib.iefvad_split_bf16x3.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.iefvad_split_bf16x3.restype = C.c_int
    lib.iefvad_split_bf16x3_many.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.iefvad_split_bf16x3_many.restype = C.c_int
    lib.iefvad_nan_rule.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.iefvad_nan_rule.restype = C.c_int
    lib.iefvad_comm_unique_id.argtypes = [C.c_void_p]
    lib.iefvad_comm_unique_id.restype = C.c_int
    lib.iefvad_comm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.iefvad_comm_create.restype = C.c_int
    lib.iefvad_comm_nranks.argtypes = [C.c_void_p]
    lib.iefvad_comm_nranks.restype = C.c_int32
    lib.iefvad_comm_destroy.argtypes = [C.c_void_p]
    lib.iefvad_comm_destroy.restype = None
    lib.iefvad_gather_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.c_void_p, C.c_size_t,
                                         C.c_void_p]
    lib.iefvad_gather_scores.restype = C.c_int
    lib.iefvad_gather_plan.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int64)]
    lib.iefvad_gather_plan.restype = C.c_int
    lib.iefvad_rccl_version.argtypes = []
    lib.iefvad_rccl_version.restype = C.c_int32
    lib.iefvad_last_error.restype = C.c_char_p
    lib.iefvad_destroy.argtypes = [C.c_void_p]
    lib.iefvad_destroy.restype = None
    if lib.iefvad_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libiefvad.so ABI {lib.iefvad_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def last_error() -> str:
    return load_library().iefvad_last_error().decode("utf-8", "replace")
