"""ctypes binding of csrc/libgtgvae_hip.so (declared in include/gtgvae.h).

There is no CPU fallback: if the shared library is missing or a call fails the
caller gets an exception.  Build with ``make -C gtg-vae_amd/csrc`` (or
``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgtgvae_hip.so")

_lib = None


class GtgError(RuntimeError):
    pass


_P = c_void_p
_SIGS = {
    "gtg_version": (c_int, []),
    "gtg_last_error": (ctypes.c_char_p, []),
    "gtg_max_nodes": (c_int, []),
    "gtg_launch_count": (c_int64, []),
    "gtg_profile_enable": (c_int, [c_int]),
    "gtg_profile_report": (c_int64, [ctypes.c_char_p, c_size_t]),
    "gtg_stream_create": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "gtg_stream_destroy": (c_int, [_P]),
    "gtg_mst_batched_triu": (c_int, [_P, c_int64, c_int, _P, _P, _P]),
    "gtg_mst_batched_full": (c_int, [_P, c_int64, c_int, _P, _P, _P]),
    "gtg_edge_prf": (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, _P]),
    "gtg_adj_build": (c_int, [_P, c_int64, c_int64, c_int, _P, _P, _P, _P, _P]),
    "gtg_encoder_fwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    "gtg_encoder_fwd": (c_int, [_P] * 6 + [c_int64, c_int, c_int, c_int] + [_P] * 3 + [c_int, _P, c_size_t, _P]),
    "gtg_encoder_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int, c_int]),
    "gtg_encoder_bwd": (c_int, [_P] * 7 + [c_int64, c_int, c_int, c_int] + [_P] * 4 + [c_int, _P, c_size_t, _P]),
    "gtg_encoder_compact_fwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "gtg_encoder_compact_fwd": (c_int, [_P] * 6 + [c_int64, c_int, c_int] + [_P] * 5 + [c_int, _P, c_size_t, _P]),
    "gtg_encoder_compact_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "gtg_encoder_compact_bwd": (c_int, [_P] * 7 + [c_int64, c_int, c_int] + [_P] * 4 + [c_int, _P, c_size_t, _P]),
    "gtg_latent_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "gtg_latent_fwd": (c_int, [_P] * 6 + [c_int64, c_int] + [_P] * 4 + [_P, c_size_t, _P]),
    "gtg_latent_bwd": (c_int, [_P] * 7 + [c_float, _P, c_int64, c_int] + [_P] * 5 + [_P, c_size_t, _P]),
    "gtg_reparam_kl_fwd": (c_int, [_P, _P, _P, c_int64, c_int, _P, _P, _P, c_size_t, _P]),
    "gtg_reparam_kl_bwd": (c_int, [_P, _P, _P, _P, c_float, _P, c_int64, c_int, _P, _P, _P]),
    "gtg_decoder_fwd": (c_int, [_P] * 5 + [c_int64, c_int, c_int, c_int] + [_P, _P, _P]),
    "gtg_bce_workspace_bytes": (c_size_t, [c_int64]),
    "gtg_bce_loss_fwd": (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "gtg_bce_loss_fwd_noalloc_max_graphs": (c_int64, []),
    "gtg_bce_loss_fwd_noalloc": (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, _P]),
    "gtg_bce_loss_bwd": (c_int, [_P, _P, c_float, _P, c_int64, c_int, c_int, _P, _P]),
    "gtg_decoder_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "gtg_decoder_bwd": (c_int, [_P] * 7 + [c_int, c_float, _P, c_int64, c_int, c_int, c_int] + [_P] * 5 + [_P, c_size_t, _P]),
    "gtg_head_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "gtg_head_fwd_bwd": (c_int, [_P] * 11 + [c_int64, c_int, c_int, c_int] + [_P] * 20 + [_P, c_float, c_float, c_float]
                         + [_P, c_size_t, _P]),
    "gtg_stage_batch": (c_int, [_P, c_int64, _P, c_int64, _P, _P, c_int64, _P, _P, c_int64, _P]),
    "gtg_gather_batch": (c_int, [_P, c_int64, _P, _P, c_int64, c_int, _P, c_int64, _P, _P, c_int64, _P, _P, _P]),
    "gtg_proj_ce_fwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "gtg_proj_ce_fwd": (c_int, [_P] * 4 + [c_int64, c_int, c_int, c_int64] + [_P] * 5 + [c_int64, c_int, _P, c_size_t, _P]),
    "gtg_lstm_fwd_workspace_bytes": (c_size_t, [c_int] * 4),
    "gtg_lstm_fwd": (c_int, [_P] * 7 + [c_int] * 5 + [_P] * 3 + [c_int, _P, c_size_t, _P]),
    "gtg_lstm_bwd_workspace_bytes": (c_size_t, [c_int] * 4),
    "gtg_lstm_bwd": (c_int, [_P] * 11 + [c_int] * 5 + [_P] * 6 + [c_int, _P, c_size_t, _P]),
    "gtg_proj_ce_bwd_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "gtg_proj_ce_bwd": (c_int, [_P] * 5 + [c_int64, c_int64, c_int, c_int] + [_P] * 3 + [c_int, _P, c_size_t, _P]),
    "gtg_word_drop": (c_int, [_P, c_int, c_int, c_int64, c_float, ctypes.c_uint64, ctypes.c_uint64, _P, _P, _P]),
    "gtg_embed_dropout_fwd": (c_int, [_P, _P, c_int64, c_int64, c_int, c_float, ctypes.c_uint64, ctypes.c_uint64, _P, c_int,
                                      _P, _P, _P]),
    "gtg_embed_dropout_bwd_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "gtg_embed_dropout_bwd": (c_int, [_P, _P, c_int64, c_int64, c_int, c_float, ctypes.c_uint64, ctypes.c_uint64, _P, _P,
                                      c_int, _P, c_size_t, _P]),
    "gtg_dropout": (c_int, [_P, c_int64, c_float, ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    "gtg_adam_step": (c_int, [_P] * 4 + [c_int64, _P, c_float, c_float, c_float, c_float, c_float, _P]),
    "gtg_adam_update": (c_int, [_P] * 4 + [c_int64, _P, c_float, c_float, c_float, c_float, _P]),
    "gtg_gemm_f32": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P, c_int, _P, c_int, _P, c_size_t, c_int, _P]),
    "gtg_gemm_splitk_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gtg_gemm_rows_workspace_bytes": (c_size_t, []),
    "gtg_split_planes": (c_int, [_P, c_int64, c_int64, c_int64, c_int, _P, _P]),
    "gtg_gemm_planes_rows": (c_int, [_P, _P, _P] + [c_int] * 4 + [_P, c_int, _P, c_int, _P, _P, c_size_t, _P]),
    "gtg_gemm_f32_rows": (c_int, [_P] * 3 + [c_int] * 8 + [_P, c_int, _P, c_int, _P, c_int, _P, c_size_t, _P]),
}

EXPORTS = tuple(_SIGS)

# Matrix pipe of the large GEMMs (include/gtgvae.h, GTG_MATRIX_*).  "bf16x3" splits every fp32 operand exactly
# into three bf16 pieces and runs six products on the bf16 matrix cores with fp32 accumulation: the same error as
# the fp32 pipe at 1.5x its speed.  "f16x2" (the default) scales every operand by a power of two and splits it
# into two fp16 pieces, three products: the same error again with half the matrix instructions.
MATRIX_MODES = {"f32": 0, "bf16": 1, "f16x2": 2, "bf16x3": 3}
_default_matrix_mode = MATRIX_MODES["f16x2"]   # changed only through set_default_matrix_mode()


def matrix_mode(mode=None) -> int:
    """The GTG_MATRIX_* code of ``mode`` ("f32" | "bf16x3" | "f16x2" | "bf16" | code | None = the default)."""
    if mode is None:
        return _default_matrix_mode
    if isinstance(mode, str):
        if mode not in MATRIX_MODES:
            raise GtgError("matrix mode must be one of %s" % sorted(MATRIX_MODES))
        return MATRIX_MODES[mode]
    if int(mode) not in MATRIX_MODES.values():
        raise GtgError("matrix mode must be one of %s" % sorted(MATRIX_MODES))
    return int(mode)


def set_default_matrix_mode(mode) -> None:
    global _default_matrix_mode
    _default_matrix_mode = matrix_mode(mode)


def lib():
    """The loaded library; raises GtgError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GtgError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(make -C gtg-vae_amd/csrc); there is no fallback path"
            )
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def profile_enable(on: bool) -> None:
    lib().gtg_profile_enable(int(on))


def profile_report():
    """[(kernel, dim0, dim1, dim2, count, total_ms)] since profiling was enabled; clears the records."""
    buf = ctypes.create_string_buffer(1 << 16)
    lib().gtg_profile_report(buf, len(buf))
    rows = []
    for line in buf.value.decode().splitlines():
        k, a, b, c, n, ms = line.split()
        rows.append((k, int(a), int(b), int(c), int(n), float(ms)))
    return rows


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().gtg_last_error().decode(errors="replace")
        raise GtgError(f"{what or 'gtg call'} failed with code {rc}: {msg}")


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return t.data_ptr()


def stream_ptr():
    import torch

    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise GtgError(
                "gtg_vae_amd runs on the GPU only (tensor on %s); move the module and its inputs to 'cuda'" % t.device
            )
