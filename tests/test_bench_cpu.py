"""bench.py's live kernel timer wraps ops.* entry points: its wrappers must accept every keyword argument the engine passes (a mismatch kills the
driver's benchmark run, not a test) - checked here by signature, on CPU."""
import inspect


def test_gemm_timer_wrappers_accept_the_ops_keywords():
    import bench
    from synchformer_amd import ops
    gt = bench.GemmTimer()
    with gt:
        for name in ('gemm', 'gemm_res_ln', 'gemm_mxfp8', 'gemm_mx_res_ln', 'qkv_time_attention', 'qkv_time_attention_mx', 'qkv_space_attention', 'qkv_space_attention_mx'):
            wrapped, orig = getattr(ops, name), gt.orig[name]
            po, pw = inspect.signature(orig).parameters, inspect.signature(wrapped).parameters
            has_kw = any(p.kind == inspect.Parameter.VAR_KEYWORD for p in pw.values())
            missing = [k for k, p in po.items() if p.kind == inspect.Parameter.KEYWORD_ONLY and k not in pw]
            assert has_kw or not missing, f'bench.GemmTimer wrapper of ops.{name} does not accept {missing}'
    assert ops.gemm is gt.orig['gemm']
    # ... and the three GEMM entry points the train steps call straight on the C ABI are wrapped on the library object and put back
    from synchformer_amd import _lib
    lib = _lib.load()
    for name in ('sf_gemm_tn_pp', 'sf_gemm_tn_splitk', 'sf_gemm_bf16_gelu_dual'):
        assert getattr(lib, name) is gt.lib_orig[name]
        assert len(inspect.signature(gt.lib_orig[name].__call__).parameters) >= 0
