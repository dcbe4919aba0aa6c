"""ctypes binding of libsynchformer_hip.so (C ABI declared in include/synchformer_hip.h).

The product path has NO fallback: if the library is missing or a symbol is absent this module raises, and
every op raises `RuntimeError` with `sf_last_error()` on a non-zero return (the reference's only error
convention is Python exceptions, e.g. scripts/train_sync.py:188-190).
"""
import ctypes as C
import os
from pathlib import Path

LIB_DIR = Path(__file__).resolve().parent / 'lib'
LIB_NAME = 'libsynchformer_hip.so'
ABI_VERSION = 10    # 10: sf_scale_rows_map (whole-token dropout of the sync transformer's inputs); 9: key-mask forms of the round-4 fused attention launches (sf_qkv_space_attention_masked, sf_qkv_time_attention2_masked), sf_qkv_time_attention2_mx, sf_side_rows, sf_gemm_bf16 config 12; 8: sf_qkv_time_attention2; 7: sf_qkv_space_attention (round 4); 6: sf_layernorm768_bwd_branch; 5: the CLS query inside the grouped attention backward kernels (sf_attention_{group,tiny}_bwd_clsq, sf_attention_cls(_combine)_stats); 4: MXFP8-output attention launches (sf_attention_cls_partial_mx, sf_attention_cls_combine_mx, sf_qkv_time_attention_mx_q); 3: round 3, second half (sf_gemm_mx_res_ln768, sf_qkv_time_attention_mx, sf_gemm_tn_pp, sf_branch_grad, ... added); 2: sf_gemm_res_ln_force_schedule
SF_NOT_APPLICABLE = -2   # include/synchformer_hip.h: "this launcher does not serve the shape, nothing was launched" (never a hipError_t)

_i64, _i32, _f32, _ptr = C.c_int64, C.c_int, C.c_float, C.c_void_p

# name -> argtypes; restype is int unless listed in _RESTYPES.  Mirrors include/synchformer_hip.h 1:1
# (tests/test_abi.py parses the header and checks every declared symbol is exported and listed here).
SIGNATURES = {
    'sf_abi_version': [],
    'sf_last_error': [],
    'sf_build_info': [],
    'sf_gemm_bf16': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i64, _ptr, _ptr, _i64, _ptr, _i32, _i64, _i64, _i64, _ptr],
    'sf_gemm_bf16_batched': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _ptr, _ptr, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _ptr],
    'sf_gemm_tn_splitk': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i32, _i64, _ptr],
    'sf_gemm_tn_pp': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _i64, _i64, _i32, _i64, _ptr],
    'sf_gemm_res_ln768': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _f32, _ptr, _i64, _i64, _i64, _ptr],
    'sf_quantize_mxfp8': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _ptr],
    'sf_gemm_mxfp8': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i32, _i64, _ptr, _i64, _ptr, _i64, _i32, _i64, _i64, _i64, _ptr],
    'sf_layernorm768_mxfp8': [_ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _f32, _ptr],
    'sf_qkv_time_attention_mx': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr],
    'sf_qkv_time_attention_mx_q': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr],
    'sf_gemm_mx_res_ln768': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _f32, _ptr, _i64, _ptr, _i64, _i64, _i64, _ptr],
    'sf_qkv_space_attention': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr],
    'sf_qkv_time_attention2': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr],
    'sf_qkv_time_attention2_mx': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr],
    'sf_side_rows': [_ptr, _i64, _ptr, _i64, _i32, _ptr, _i64, _ptr, _i64, _i32, _i64, _i32, _ptr],
    'sf_qkv_space_attention_masked': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr, _ptr],
    'sf_qkv_time_attention2_masked': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr, _ptr],
    'sf_qkv_space_attention_mx': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr],
    'sf_gemm_force_config': [_i32],
    'sf_gemm_bf16_auto_config': [_i64, _i64, _i64, _i32],
    'sf_gemm_bf16_gelu_dual': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _i64, _i64, _ptr],
    'sf_gemm_res_ln_force_schedule': [_i32],
    'sf_qkv_time_force_schedule': [_i32],
    'sf_gemm_mx_force_schedule': [_i32],
    'sf_layernorm768': [_ptr, _i64, _ptr, _ptr, _ptr, _ptr, _i32, _i64, _ptr, _i32, _i64, _f32, _ptr],
    'sf_broadcast_rows768': [_ptr, _i64, _i64, _ptr, _i64, _i64, _ptr],
    'sf_gather_rows768': [_ptr, _i64, _ptr, _ptr, _i32, _i64, _i64, _ptr],
    'sf_im2col_video': [_ptr, _i32, _ptr, _i64, _ptr],
    'sf_im2col_spec': [_ptr, _ptr, _i64, _i32, _i32, _ptr],
    'sf_attention': [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _ptr],
    'sf_mel_frontend': [_ptr, _i64, _i32, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _i32, _f32, _f32, _ptr],
    'sf_transpose_bf16': [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _ptr],
    'sf_cast_bf16': [_ptr, _i64, _ptr, _i64, _i64, _i32, _f32, _ptr],
    'sf_transpose_bf16_multi': [_ptr, _ptr, _i32, _i32, _ptr],
    'sf_softmax_rows': [_ptr, _i64, _ptr, _i64, _i64, _i32, _i32, _f32, _ptr],
    'sf_softmax_bwd_rows': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i32, _i32, _f32, _ptr],
    'sf_layernorm768_bwd': [_ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i32, _ptr, _ptr, _i32, _ptr, _i64, _f32, _ptr],
    'sf_layernorm768_bwd_bf16': [_ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i32, _ptr, _ptr, _i32, _ptr, _i64, _f32, _ptr],
    'sf_colsum': [_ptr, _i32, _i64, _i64, _i32, _ptr, _i32, _ptr, _ptr],
    'sf_seqsum': [_ptr, _i64, _i32, _i32, _i32, _ptr, _i32, _ptr],
    'sf_wgrad_sum': [_ptr, _i64, _i32, _ptr, _ptr, _i64, _ptr, _i32, _ptr],
    'sf_gelu_fwd': [_ptr, _ptr, _i64, _ptr],
    'sf_gelu_bwd': [_ptr, _ptr, _ptr, _i64, _ptr],
    'sf_gelu_bwd_bf16': [_ptr, _ptr, _ptr, _i64, _ptr],
    'sf_cross_entropy': [_ptr, _i64, _ptr, _i32, _i32, _ptr, _ptr, _i64, _f32, _ptr],
    'sf_scale_rows_map': [_ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _i32, _i32, _ptr],
    'sf_scale_seq_add': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i32, _ptr],
    'sf_add_scale_ln768': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _i64, _f32, _ptr],
    'sf_branch_grad': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i32, _ptr, _i32, _ptr, _ptr],
    'sf_dropout': [_ptr, _i32, _i64, _ptr, _i64, _ptr, _i64, _i64, _i32, _f32, C.c_uint32, _ptr],
    'sf_grad_norm': [_ptr, _i64, _ptr, _ptr, _ptr],
    'sf_adam_clip_step': [_ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _f32, _f32, _f32, _f32, _f32, _i32, _ptr],
    'sf_meanpool_l2norm768': [_ptr, _i64, _i32, _ptr, _i64, _i32, _i64, _ptr],
    'sf_similarity_f32': [_ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _i32, _i32, _f32, _ptr],
    'sf_copy_rows_bf16': [_ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _i32, _ptr],
    'sf_reduce_groups_bf16': [_ptr, _i64, _i64, _i32, _ptr, _i64, _i32, _i64, _i32, _ptr],
    'sf_attention_cls_bwd': [_ptr, _i64, _i32, _ptr, _ptr, _i64, _i64, _i32, _i32, _ptr, _i64, _i64, _i32, _ptr, _ptr, _ptr, _i64, _i64, _i32,
                             _i32, _f32, _i32, _ptr],
    'sf_meanpool_l2norm768_bwd': [_ptr, _i64, _i32, _ptr, _i64, _ptr, _i64, _i32, _i64, _ptr],
    'sf_im2col_video_clips': [_ptr, _i32, _i64, _i64, _i32, _i32, _i32, _ptr, _ptr],
    'sf_im2col_video_tokens': [_ptr, _i32, _i64, _i64, _i32, _i32, _i32, _ptr, _ptr],
    'sf_mel_frontend_clips': [_ptr, _i64, _i64, _i64, _i64, _i32, _i32, _i32, _ptr, _ptr, _ptr, _ptr, _ptr, _i32, _ptr, _ptr, _i32, _f32, _f32, _ptr],
    'sf_token_mask_video': [_ptr, _i64, _ptr, _ptr, _ptr],
    'sf_token_mask_spec': [_ptr, _i64, _i32, _i32, _ptr, _ptr, _ptr],
    'sf_attention_masked': [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _ptr, _ptr],
    'sf_attention_cls_masked': [_ptr, _i64, _i32, _ptr, _ptr, _i64, _i64, _i32, _i32, _ptr, _i64, _i64, _i32, _i64, _i32, _i32, _f32, _ptr, _ptr],
    'sf_shift_window_preds': [_ptr, _i64, _i32, _i32, _i32, _ptr, _ptr, _ptr],
    'sf_attention_tiny_bwd': [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                              _i32, _f32, _ptr],
    'sf_rowsum_bf16': [_ptr, _i64, _i32, _i64, _ptr, _i32, _ptr],
    'sf_attention_group_bwd': [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                               _i32, _f32, _ptr],
    'sf_attention_cls_partial': [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _ptr, _ptr],
    'sf_attention_cls_combine': [_ptr, _i32, _ptr, _i64, _i64, _i32, _i64, _i32, _ptr],
    'sf_layernorm768_bwd_branch': [_ptr, _i64, _ptr, _ptr, _i32, _i64, _ptr, _i64, _i32, _ptr, _ptr, _i32, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _f32, _ptr],
    'sf_attention_tiny_bwd_clsq': [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _ptr],
    'sf_attention_cls_stats': [_ptr, _i64, _i32, _ptr, _ptr, _i64, _i64, _i32, _i32, _ptr, _i64, _i64, _i32, _i64, _i32, _i32, _f32, _ptr, _ptr],
    'sf_attention_cls_combine_stats': [_ptr, _i32, _ptr, _i64, _i64, _i32, _i64, _i32, _ptr, _ptr],
    'sf_attention_group_bwd_clsq': [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _ptr],
    'sf_attention_cls_partial_mx': [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _ptr, _ptr],
    'sf_attention_cls_combine_mx': [_ptr, _i32, _ptr, _i64, _ptr, _i64, _i64, _i32, _i64, _i32, _ptr],
    'sf_qkv_time_attention': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr],
    'sf_qkv_time_attention_masked': [_ptr, _i64, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i32, _f32, _ptr, _ptr],
    'sf_attention_cls_partial_masked': [_ptr, _ptr, _ptr, _i64, _ptr, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _ptr, _ptr, _ptr],
    'sf_attention_cls': [_ptr, _i64, _i32, _ptr, _ptr, _i64, _i64, _i32, _i32, _ptr, _i64, _i64, _i32, _i64, _i32, _i32, _f32, _ptr],
}
_RESTYPES = {'sf_last_error': C.c_char_p, 'sf_build_info': C.c_char_p, 'sf_gemm_force_config': None, 'sf_gemm_res_ln_force_schedule': None, 'sf_qkv_time_force_schedule': None, 'sf_gemm_mx_force_schedule': None}

_lib = None


def lib_path() -> Path:
    return Path(os.environ.get('SYNCHFORMER_HIP_LIB', LIB_DIR / LIB_NAME))


def _typed(path: Path):
    lib = C.CDLL(str(path))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError -> loud failure on a stale build
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    got = lib.sf_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f'{path}: ABI version {got}, expected {ABI_VERSION} (stale build?)')
    return lib


def load():
    """Load (once) and type the shared library.  Raises if it is absent - build it with
    `python -c "import __graft_entry__ as g; g.build()"` or `make -C synchformer_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; import it FIRST so this library binds to the same HIP runtime instance
    # (loading ours first would pull /opt/rocm's copy and leave the process with two runtimes that cannot see
    # each other's allocations or streams).
    import torch  # noqa: F401
    path = lib_path()
    if not path.exists():
        raise RuntimeError(f'{path} not found: the HIP extension is not built. There is no CPU fallback; run '
                           f'`python -c "import __graft_entry__ as g; g.build()"` first.')
    _lib = _typed(path)
    return _lib


# ---- the ablation build (round 6) ------------------------------------------------------------------------------------------------------------------
# The measured-slower alternatives that earlier rounds kept behind force switches - sf_gemm_bf16 tile configs 1-3, 5, 6, 8-10 and 12 (sf_gemm_w4.hip), schedule 2 of
# sf_gemm_res_ln768 (sf_gemm_ln2.hip) - are NOT in libsynchformer_hip.so any more: the same sources compiled with -DSF_ABLATION give lib/ab/libsynchformer_hip_ablation.so
# (build.build_ablation()), which only the bit-identity tests and the tools/ benchmarks load, explicitly.  No product module calls load_ablation() / using().
ABLATION_PATH = LIB_DIR / 'ab' / 'libsynchformer_hip_ablation.so'
_ablation = None


def load_ablation():
    global _ablation
    if _ablation is None:
        import torch  # noqa: F401
        if not ABLATION_PATH.exists():
            raise RuntimeError(f'{ABLATION_PATH} not found: python -c "from synchformer_amd import build; build.build_ablation()"')
        _ablation = _typed(ABLATION_PATH)
    return _ablation


class using:
    """`with _lib.using(_lib.load_ablation()):` - every `ops.*` launch inside goes to that library instead of the product one (tests / tools only)."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        global _lib
        load()
        self.prev, _lib = _lib, self.lib
        return self.lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev


def check(rc: int, what: str):
    if rc != 0:
        msg = load().sf_last_error().decode(errors='replace')
        raise RuntimeError(f'{what} failed (rc={rc}): {msg}')
