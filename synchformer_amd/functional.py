"""Functional dispatcher ops WITH autograd ("a PyTorch dispatcher library ... autograd via torch.library.register_autograd").

The model-level bridges (`train.SyncTrainFunction`, `stage1.AVCLIPTrainFunction`) are what the drop-in modules train through; these two ops expose the same
HIP kernels at operator granularity, so code outside this package can compose and differentiate them like any torch op:

    torch.ops.synchformer.linear(x, weight, bias) -> y          y = x @ weight.T + bias        (sf_gemm_bf16; nn.Linear of vit_helper.py:87-141 / modeling_ast.py)
    torch.ops.synchformer.layer_norm768(x, gamma, beta, eps)    LayerNorm over 768 columns     (sf_layernorm768; vit_helper.py:364-376)

Both are functional (fresh output tensor), carry FakeTensor implementations, an autograd formula on the backward kernels of the train steps (dgrad on
sf_gemm_bf16 against a bf16 W^T copy, wgrad + bias gradient on sf_gemm_tn_splitk straight from the row-major operands, sf_layernorm768_bwd) and an autocast
rule (inputs cast to bf16 / fp32 the way `torch.autocast('cuda')` treats linear / layer_norm, train_sync.py:178).  HIP device only - there is no CPU kernel,
a CPU tensor raises in the launcher.  Shapes the backward serves: weight (N, K) with N % 128 == 0 and K % 128 == 0, at least 512 rows - checked when the
FORWARD records its autograd node (setup_context), not first at backward time.
Output dtype contract: `linear` ALWAYS returns bf16 (the kernels' operand / output type), also for fp32 inputs outside autocast - unlike nn.Linear, which would return
fp32 there; cast the result if fp32 is needed.  Gradients come back in the dtype of the input they belong to, and only the ones autograd asks for are computed
(`ctx.needs_input_grad`); the bf16 W^T operand of the data gradient is cached per (leaf weight object, version) - never for autocast's temporaries."""
import weakref
from typing import Optional

import torch

from . import _lib, ops

_done = False


def _rows(x: torch.Tensor):
    return x.reshape(-1, x.shape[-1])


def register():
    """Idempotent; called on first use of `linear` / `layer_norm768` below (and by ops.register_torch_ops)."""
    global _done
    if _done:
        return
    from torch.library import custom_op, register_autocast
    from .train import _chk, _st, _wgrad_split, colsum, ln_bwd, transpose

    # ---- linear ----------------------------------------------------------------------------------------------------------------
    @custom_op('synchformer::linear', mutates_args=(), device_types='cuda')
    def _linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
        x2 = _rows(x).to(torch.bfloat16).contiguous()
        w = weight.to(torch.bfloat16).contiguous()
        b = bias.to(torch.float32).contiguous() if bias is not None else None
        out = torch.empty(x2.shape[0], w.shape[0], device=x.device, dtype=torch.bfloat16)
        ops.gemm(x2, w, b, out)
        return out.view(*x.shape[:-1], w.shape[0])

    @_linear.register_fake
    def _(x, weight, bias):
        torch._check(x.shape[-1] == weight.shape[1], lambda: 'synchformer::linear: x (..., K) against weight (N, K)')
        return x.new_empty((*x.shape[:-1], weight.shape[0]), dtype=torch.bfloat16)

    _wt_cache = {}                                                # id(weight) -> (weakref to the weight OBJECT, version, bf16 W^T): one transpose per optimizer step, not one per backward
                                                                  # (a WeakKeyDictionary cannot hold tensors: it compares keys with ==, which is elementwise)

    def _wT(weight, w, N, K, dev):
        """bf16 W^T for dX = dY W.  Cached only for a leaf weight (an nn.Parameter / a frozen tensor the caller keeps alive), keyed on the tensor OBJECT - a
        dead weight drops its entry (weakref callback), and a hit must be the same live object, so a recycled device address or a recycled id() can never serve
        another tensor's transpose - and on its version counter (in-place optimizer updates bump it; writes through `.data` do not - do not update weights that
        way).  Under autocast `weight` is the per-forward bf16 temporary of the cast (not a leaf, version 0, address reused by the allocator step after step):
        never cached, transposed per backward."""
        cacheable = weight.is_leaf and weight.grad_fn is None
        key = id(weight)
        if cacheable:
            hit = _wt_cache.get(key)
            if hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2].shape == (K, N):
                return hit[2]
        wT = torch.empty(K, N, device=dev, dtype=torch.bfloat16)
        transpose(w, K, 0, 0, wT, N, 0, 0, N, K, N)
        if cacheable:
            _wt_cache[key] = (weakref.ref(weight, lambda _r, k=key: _wt_cache.pop(k, None)), weight._version, wT)
        return wT

    def _bwd_shapes_ok(M, N, K):
        return not (N % 128 or K % 128 or M < 512)

    @custom_op('synchformer::linear_backward', mutates_args=(), device_types='cuda')
    def _linear_bwd(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, need_dx: bool, need_dw: bool, need_bias: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """-> (dx, dw, db); a gradient that was not asked for comes back as an EMPTY tensor and its launches are skipped."""
        dy2, x2 = _rows(dy).to(torch.bfloat16).contiguous(), _rows(x).to(torch.bfloat16).contiguous()
        w = weight.to(torch.bfloat16).contiguous()
        M, (N, K) = x2.shape[0], w.shape
        if not _bwd_shapes_ok(M, N, K):
            raise NotImplementedError(f'synchformer::linear backward serves N % 128 == 0, K % 128 == 0, >= 512 rows (got M {M}, N {N}, K {K})')
        dev = x.device
        dx = torch.empty(0, device=dev, dtype=torch.float32)       # (three distinct empties: a custom op may not return one tensor twice)
        if need_dx:                                                # dX = dY W: the forward kernel on a bf16 W^T copy
            dx = torch.empty(M, K, device=dev, dtype=torch.float32)
            ops.gemm(dy2, _wT(weight, w, N, K, dev), None, dx)
            dx = dx.view(*x.shape[:-1], K)
        dw, db = torch.empty(0, device=dev, dtype=torch.float32), torch.empty(0, device=dev, dtype=torch.float32)
        if need_dw or need_bias:                                   # dW = dY^T X (+ the bias gradient from the same launch): split-K partial planes summed by sf_seqsum
            m_pad = ((M + 63) // 64) * 64
            tiles = (N // 128) * (K // 128)
            split = _wgrad_split(tiles) if M >= 8192 else max(1, min(_wgrad_split(tiles), m_pad // 128))
            kc = ((m_pad // split + 63) // 64) * 64
            part = torch.empty(split * N, K, device=dev, dtype=torch.float32)
            bpart = torch.empty(split, N, device=dev, dtype=torch.float32)
            lib = _lib.load()
            _chk(lib.sf_gemm_tn_splitk(dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0), part.data_ptr(), bpart.data_ptr(), M, N, K, split, kc, _st()),
                 'sf_gemm_tn_splitk')
            if need_dw:
                dw = torch.empty(N, K, device=dev, dtype=torch.float32)
                _chk(lib.sf_seqsum(part.data_ptr(), K, split, N, K, dw.data_ptr(), 0, _st()), 'sf_seqsum')
            if need_bias:
                db = torch.empty(N, device=dev, dtype=torch.float32)
                _chk(lib.sf_seqsum(bpart.data_ptr(), N, split, 1, N, db.data_ptr(), 0, _st()), 'sf_seqsum')
        return dx, dw, db

    @_linear_bwd.register_fake
    def _(dy, x, weight, need_dx, need_dw, need_bias):
        e = lambda: x.new_empty((0,), dtype=torch.float32)         # noqa: E731
        return (x.new_empty(x.shape, dtype=torch.float32) if need_dx else e(), weight.new_empty(weight.shape, dtype=torch.float32) if need_dw else e(),
                weight.new_empty((weight.shape[0],), dtype=torch.float32) if need_bias else e())

    def _linear_setup(ctx, inputs, output):
        x, weight, bias = inputs
        M, (N, K) = x.numel() // max(x.shape[-1], 1), weight.shape
        if not _bwd_shapes_ok(M, N, K):                            # (setup_context only runs when some input requires grad: fail where the graph is built)
            raise NotImplementedError(f'synchformer::linear: the backward serves N % 128 == 0, K % 128 == 0, >= 512 rows (got M {M}, N {N}, K {K}); '
                                      'call it under torch.no_grad() or use torch.nn.functional.linear for this shape')
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None

    def _linear_backward(ctx, dy):
        x, weight = ctx.saved_tensors
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        dx, dw, db = torch.ops.synchformer.linear_backward(dy, x, weight, need_dx, need_dw, need_db)
        return (dx.to(x.dtype) if need_dx else None), (dw.to(weight.dtype) if need_dw else None), (db if need_db else None)

    _linear.register_autograd(_linear_backward, setup_context=_linear_setup)
    register_autocast('synchformer::linear', 'cuda', torch.bfloat16)          # like torch.nn.functional.linear under autocast

    # ---- LayerNorm over 768 columns ----------------------------------------------------------------------------------------------
    @custom_op('synchformer::layer_norm768', mutates_args=(), device_types='cuda')
    def _ln(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
        x2 = _rows(x).to(torch.float32).contiguous()
        out = torch.empty(x2.shape, device=x.device, dtype=torch.bfloat16)
        ops.layernorm(x2, gamma.to(torch.float32).contiguous(), beta.to(torch.float32).contiguous(), out, eps)
        return out.view(x.shape)

    @_ln.register_fake
    def _(x, gamma, beta, eps):
        torch._check(x.shape[-1] == 768, lambda: 'synchformer::layer_norm768: 768 columns')
        return x.new_empty(x.shape, dtype=torch.bfloat16)

    @custom_op('synchformer::layer_norm768_backward', mutates_args=(), device_types='cuda')
    def _ln_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, eps: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        x2 = _rows(x).to(torch.float32).contiguous()
        dy2 = _rows(dy).contiguous()
        if dy2.dtype not in (torch.bfloat16, torch.float32):
            dy2 = dy2.float()
        rows = x2.shape[0]
        dx = torch.empty_like(x2)
        dg, db = torch.empty(768, device=x.device, dtype=torch.float32), torch.empty(768, device=x.device, dtype=torch.float32)
        ws = torch.empty(3 * 768 * ((rows + 3) // 4), device=x.device, dtype=torch.float32)
        ln_bwd(x2, gamma.to(torch.float32).contiguous(), dy2, dx, dg, db, ws, rows, eps)
        return dx.view(x.shape), dg, db

    @_ln_bwd.register_fake
    def _(dy, x, gamma, eps):
        return x.new_empty(x.shape, dtype=torch.float32), gamma.new_empty((768,), dtype=torch.float32), gamma.new_empty((768,), dtype=torch.float32)

    def _ln_setup(ctx, inputs, output):
        x, gamma, beta, eps = inputs
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps

    def _ln_backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        dx, dg, db = torch.ops.synchformer.layer_norm768_backward(dy, x, gamma, ctx.eps)
        return dx.to(x.dtype), dg.to(gamma.dtype), db.to(gamma.dtype), None

    _ln.register_autograd(_ln_backward, setup_context=_ln_setup)
    register_autocast('synchformer::layer_norm768', 'cuda', torch.float32)    # like layer_norm under autocast: fp32 statistics
    _done = True


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    register()
    return torch.ops.synchformer.linear(x, weight, bias)


def layer_norm768(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    register()
    return torch.ops.synchformer.layer_norm768(x, gamma, beta, eps)
