"""Stage-2 train step on the HIP path (scripts/train_sync.py:153-237).

With `is_trainable: False` extractors (configs/sync.yaml:7,19; train_utils.py:199-204) only vproj, aproj and the sync
transformer train: 22,619,157 parameters.  `SyncTrainer` keeps them as ONE flat fp32 master buffer (+ flat grad, Adam m/v,
bf16 operand copies), runs forward with saved activations, the hand-scheduled backward, an optional RCCL gradient
all-reduce (one 90 MB bucket - the backward is 0.3 % of the step, there is nothing to overlap it with but the next step's
frozen extractor forward) and the fused clip + Adam kernel.  Dropout (the reference trains with embd/resid/attn_pdrop 0.1,
configs/sync.yaml:47-49) uses counter-based masks regenerated in the backward (sf_dropout); the mask stream is not torch's
Philox stream, so parity under dropout is exact only given the masks (SURVEY §7 hard part (g)).

All compute is in libsynchformer_hip (GEMMs via sf_gemm_bf16 / sf_gemm_bf16_batched on transposed copies); torch provides
memory, the stream and torch.distributed.
"""
import ctypes as C
import math
import os
from typing import Dict, List, Optional

import torch

from . import _lib, ops, synth
from .engine import D, FF, EPS_SYNC, SynchformerEngine

_F32, _BF16 = 0, 1


def _st():
    return torch.cuda.current_stream().cuda_stream


def _chk(rc, what):
    _lib.check(rc, what)


def transpose(inp: torch.Tensor, ld_in, sI0, sI1, out: torch.Tensor, ld_out, sO0, sO1, R, Cc, R_pad, b_outer=1, b_inner=1):
    _chk(_lib.load().sf_transpose_bf16(inp.data_ptr(), ld_in, sI0, sI1, out.data_ptr(), ld_out, sO0, sO1, R, Cc, R_pad, b_outer, b_inner, _st()),
         'sf_transpose_bf16')


def cast_bf16(x: torch.Tensor, y: torch.Tensor, rows, cols, scale=1.0):
    _chk(_lib.load().sf_cast_bf16(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), rows, cols, float(scale), _st()), 'sf_cast_bf16')


def bgemm(a, lda, sA0, sA1, w, ldw, sW0, sW1, c, ldc, sC0, sC1, M, N, K, b_outer, b_inner):
    _chk(_lib.load().sf_gemm_bf16_batched(a.data_ptr(), lda, sA0, sA1, w.data_ptr(), ldw, sW0, sW1, None, c.data_ptr(),
                                          _BF16 if c.dtype == torch.bfloat16 else _F32, ldc, sC0, sC1, M, N, K, b_outer, b_inner, _st()),
         'sf_gemm_bf16_batched')


def ln_bwd(x, gamma, dy, dx, dgamma, dbeta, ws, rows, eps, *, x_map=None, dy_map=None, dx_map=None, acc_dx=False, acc_dp=False):
    m = lambda t: (C.c_int64 * 6)(*t) if t is not None else None
    fn = _lib.load().sf_layernorm768_bwd_bf16 if dy.dtype == torch.bfloat16 else _lib.load().sf_layernorm768_bwd
    _chk(fn(x.data_ptr(), x.stride(0), m(x_map), gamma.data_ptr(), dy.data_ptr(), dy.stride(0), m(dy_map), dx.data_ptr(), dx.stride(0), m(dx_map),
            int(acc_dx), dgamma.data_ptr(), dbeta.data_ptr(), int(acc_dp), ws.data_ptr(), rows, float(eps), _st()), 'sf_layernorm768_bwd')


def dropout(x, y, rows, cols, p, seed, residual=None):
    """y = dropout_p(x) (+ residual) with the counter-based mask of (seed, rows x cols); x/y fp32 or bf16 2-D views."""
    _chk(_lib.load().sf_dropout(x.data_ptr(), _BF16 if x.dtype == torch.bfloat16 else _F32, x.stride(0),
                                residual.data_ptr() if residual is not None else None, residual.stride(0) if residual is not None else 0,
                                y.data_ptr(), y.stride(0), rows, cols, float(p), int(seed) & 0xFFFFFFFF, _st()), 'sf_dropout')


def colsum(x, rows, cols, out, ws, accumulate=False):
    _chk(_lib.load().sf_colsum(x.data_ptr(), _BF16 if x.dtype == torch.bfloat16 else _F32, x.stride(0), rows, cols, out.data_ptr(), int(accumulate),
                               ws.data_ptr(), _st()), 'sf_colsum')


def _wgrad_split(tiles: int, slots: int = 512) -> int:
    """Number of contraction chunks for a weight gradient with `tiles` 128x128 output tiles: the smallest split whose tiles * split
    workgroups fill whole rounds of the 512 resident workgroup slots (256 CUs x 2) to >= 94 % - e.g. 36 tiles -> 14 (504 workgroups),
    108 -> 9 (972), 144 -> 7 (1008).  Measured (tools/bench_wgrad.py, M = 43,932): 92 -> 77 us, 247 -> 211 us, 352 -> 261 us against the
    earlier min(32, 768 // tiles) rule."""
    best, best_eff = 1, 0.0
    for s_ in range(2, 33):
        n = tiles * s_
        eff = n / (((n + slots - 1) // slots) * slots)
        if eff >= 0.94:
            return s_
        if eff > best_eff:
            best, best_eff = s_, eff
    return best


def constant_with_warmup_lr(step: int, base_lr: float, warmup: int = 1000) -> float:
    """Stage-2 schedule `constant_with_warmup` (scripts/train_utils.py:236-246; configs/sync.yaml lr_scheduler): torch's
    SequentialLR([LinearLR(start_factor 1/100, total_iters warmup), ConstantLR(factor 1)], milestones [warmup]) as a function of the
    number of optimizer steps already taken: factor 0.01 + 0.99 * step / warmup, then 1."""
    if step >= warmup:
        return base_lr
    return base_lr * (0.01 + 0.99 * step / warmup)


# trainable keys, in the reference's state-dict order (vproj, aproj, transformer.*)
def trainable_keys(schema) -> List[str]:
    return [k for k in schema if k.startswith(('vproj.', 'aproj.', 'transformer.'))]


class FlatTrainer:
    """Shared machinery of the HIP train steps: ONE flat fp32 master buffer for the trainable tensors (+ flat grad, Adam m / v,
    bf16 operand copies and bf16 W^T copies for dgrad), linear / attention backward on the GEMM entry points, the flat-bucket
    gradient all-reduce and the fused clip + Adam step."""
    attn_pdrop = 0.0

    def _init_flat(self, state_dict: Dict[str, torch.Tensor], keys: List[str], device, lr, betas, eps, max_clip_norm):
        self.dev = torch.device(device)
        self.lr, self.betas, self.eps, self.max_clip_norm = lr, betas, eps, max_clip_norm
        self.keys = list(keys)
        sizes = [state_dict[k].numel() for k in self.keys]
        self.n = sum(sizes)
        self.flat_p = torch.empty(self.n, device=self.dev, dtype=torch.float32)
        self.flat_g = torch.zeros(self.n, device=self.dev, dtype=torch.float32)
        self.flat_m = torch.zeros_like(self.flat_g)
        self.flat_v = torch.zeros_like(self.flat_g)
        self.flat_b = torch.empty(self.n, device=self.dev, dtype=torch.bfloat16)
        self.p, self.g, self.b = {}, {}, {}
        o = 0
        for k, sz in zip(self.keys, sizes):
            shp = state_dict[k].shape
            self.p[k] = self.flat_p[o:o + sz].view(shp)
            self.g[k] = self.flat_g[o:o + sz].view(shp)
            self.b[k] = self.flat_b[o:o + sz].view(shp)
            self.p[k].copy_(state_dict[k])
            o += sz
        self.flat_b.copy_(self.flat_p)
        self.step_count = 0
        self._ws_prefix = ''
        self.tn_wgrad = os.environ.get('SF_TN_WGRAD', '1') != '0'      # weight gradients straight from row-major operands (sf_gemm_tn_splitk)
        self.tn_pp = os.environ.get('SF_TN_PP', '1') != '0'            # ... the big ones on the quadrant-phased 256 x 256 kernel (sf_gemm_tn_pp)
        self.n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count if torch.cuda.is_available() else 256
        # SF_WGRAD_SIDE=1: the big weight gradients next to their dgrad on a low-priority side stream (one per workspace prefix = per tower stream), _lin_bwd
        self.wgrad_side = os.environ.get('SF_WGRAD_SIDE', '0') != '0'
        self._wgrad_streams: Dict[str, tuple] = {}
        self.norm = torch.zeros(1, device=self.dev, dtype=torch.float32)
        self.loss = torch.zeros(1, device=self.dev, dtype=torch.float32)
        self._ws: Dict[str, torch.Tensor] = {}
        self._wT: Dict[str, torch.Tensor] = {}
        self._wT_table = None                     # device table of sf_transpose_bf16_multi, built on the first refresh
        self._wT_single: List[str] = []
        self._refresh_transposed()

    # ---- small helpers -----------------------------------------------------------------------------------
    def _buf(self, name, shape, dtype, zero=False):
        name = self._ws_prefix + name                                     # per-stream workspaces (Stage-1 runs its two towers concurrently)
        t = self._ws.get(name)
        n = int(math.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.zeros(n, device=self.dev, dtype=dtype)
            self._ws[name] = t
        v = t[:n].view(*shape)
        if zero:
            v.zero_()
        return v

    def _refresh_transposed(self):
        """bf16 W^T copies (dgrad operands) of every 2-D trainable weight; called after each optimizer step.  One launch for all of them
        (sf_transpose_bf16_multi over a device-resident table: the operand copies never move); odd shapes take sf_transpose_bf16 one by one."""
        if self._wT_table is None:
            rows, prefix, self._wT_single = [], [0], []
            for k in self.keys:
                w = self.b[k]
                if w.dim() != 2 or not k.endswith('weight'):
                    continue
                N, K = w.shape
                n_pad = ((N + 63) // 64) * 64
                t = self._wT[k] = torch.zeros(K, n_pad, device=self.dev, dtype=torch.bfloat16)
                if K % 8 == 0 and w.data_ptr() % 16 == 0 and w.stride(0) % 8 == 0:
                    tx, ty = (n_pad + 63) // 64, (K + 63) // 64
                    rows.append([w.data_ptr(), t.data_ptr(), w.stride(0), n_pad, N, K, n_pad, tx])
                    prefix.append(prefix[-1] + tx * ty)
                else:
                    self._wT_single.append(k)
            self._wT_table = (torch.tensor(rows, dtype=torch.int64, device=self.dev), torch.tensor(prefix, dtype=torch.int32, device=self.dev), len(rows), prefix[-1]) \
                if rows else ()
        if self._wT_table:
            tab, pre, n, total = self._wT_table
            _chk(_lib.load().sf_transpose_bf16_multi(tab.data_ptr(), pre.data_ptr(), n, total, _st()), 'sf_transpose_bf16_multi')
        for k in self._wT_single:
            w, t = self.b[k], self._wT[k]
            transpose(w, w.shape[1], 0, 0, t, t.shape[1], 0, 0, w.shape[0], w.shape[1], t.shape[1])

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach().clone() for k, v in self.p.items()}

    def load_params(self, tensors: Dict[str, torch.Tensor]):
        """Overwrite the fp32 master copy (e.g. from nn.Parameters an external optimizer updated) and refresh operand copies."""
        for k in self.keys:
            self.p[k].copy_(tensors[k])
        self.flat_b.copy_(self.flat_p)
        self._refresh_transposed()

    # ---- linear layer forward / backward -------------------------------------------------------------------
    def _wb(self, name):
        return self.b[name + '.weight'], self.p[name + '.bias']

    def _lin_bwd(self, name, dy_b, x_b, M, *, need_dx=True, dx_out=None, tag='', wkey=None, bkey=None, acc_bias=False, acc_dx=False, dy_f32=None,
                 dx_dtype=torch.float32, bias_done=False):
        """dy_b (M, N) bf16 (row stride may exceed N), x_b (M, K) bf16 saved input.  Fills g[W], g[b]; returns dx fp32 (M, K).
        `wkey` / `bkey` name tensors that do not follow the `<name>.weight` / `<name>.bias` convention (in_proj_weight, conv kernels).
        dx_dtype=torch.bfloat16 lets the dgrad GEMM write dx in bf16 when the only consumer is a bf16 kernel (no fp32 round trip + cast).
        bias_done: g[b] has been filled by the producer of dy_b (sf_branch_grad sums the fp32 gradient while it casts it)."""
        wkey, bkey = wkey or name + '.weight', bkey or name + '.bias'
        N = self.p[wkey].shape[0]
        K = self.p[wkey].numel() // N
        m_pad = ((M + 63) // 64) * 64
        if dy_f32 is not None and not bias_done:                                     # fp32 gradient at hand: sum that (cancellation-prone biases)
            ws = self._buf('colsum_ws', (N * ((M + 63) // 64),), torch.float32)
            colsum(dy_f32, M, N, self.g[bkey], ws, accumulate=acc_bias)
        # dW = dy^T x: an (N, K) output is only (N/128)*(K/128) tiles (36 for a 768x768 weight) however long the M contraction is,
        # so for long M the contraction is split into `split` chunks run as one batched GEMM (fills the 256 CUs) and summed after
        tiles = ((N + 127) // 128) * ((K + 127) // 128)
        tn = self.tn_wgrad and N % 128 == 0 and K % 128 == 0 and M >= 512 and dy_b.stride(0) % 8 == 0 and x_b.stride(0) % 8 == 0
        split = _wgrad_split(tiles) if M >= 8192 else (max(1, min(_wgrad_split(tiles), m_pad // 128)) if tn else 1)   # short M: >= 128 rows per chunk
        kc = ((m_pad // split + 63) // 64) * 64
        # (sf_gemm_tn_pp addresses its operands with 32-bit byte offsets: beyond 4 GiB of dY / X rows the split-K kernel below, which has no such limit, takes over)
        pp_rows = (M + 127) // 128 * 128 + 128 * self.n_cu
        if tn and self.tn_pp and N % 256 == 0 and K % 256 == 0 and M >= 8192 and pp_rows * max(dy_b.stride(0), x_b.stride(0)) * 2 < (1 << 32):
            # the big weight gradients: quadrant-phased 256 x 256 kernel (sf_gemm_tn_pp), as many chunks as fill the chip once with 256 x 256 tiles
            t256 = (N // 256) * (K // 256)
            sp = max(1, self.n_cu // t256)
            kc2 = ((M + sp - 1) // sp + 127) // 128 * 128
            sp = (M + kc2 - 1) // kc2
            want_b = dy_f32 is None and not bias_done
            if self.wgrad_side and need_dx:
                # dgrad and wgrad of one Linear are independent: the dgrad goes FIRST on the compute stream - with two clips its N = 768 grid is 516 tiles = 2.016
                # rounds of the 256 CUs, so its third round leaves the chip nearly empty - and the weight gradient (one round of chunk items + its reduction) runs
                # NEXT TO it on a low-priority side stream, filling the CUs the dgrad's workgroups leave.  The compute stream waits for the side stream before
                # anything else is launched, so every later kernel (and every gradient-bucket hook) sees g[W] / g[b] final and no workspace is overwritten early.
                main = torch.cuda.current_stream(self.dev)
                ws, ev_in, ev_out = self._wgrad_stream()
                ev_in.record(main)
                dx = self._lin_dgrad(dy_b, M, N, K, wkey, need_dx, dx_out, tag, acc_dx, dx_dtype)
                with torch.cuda.stream(ws):
                    ws.wait_event(ev_in)
                    part = self._buf('wgrad_part_s', (sp * N, K), torch.float32)        # (allocated under the side stream: its own blocks)
                    bpart = self._buf('bgrad_part_s', (sp, N), torch.float32) if want_b else None
                    _chk(_lib.load().sf_gemm_tn_pp(dy_b.data_ptr(), dy_b.stride(0), x_b.data_ptr(), x_b.stride(0), part.data_ptr(),
                                                   bpart.data_ptr() if bpart is not None else None, M, N, K, sp, kc2, _st()), 'sf_gemm_tn_pp')
                    self._wgrad_sum(part, bpart, sp, N, K, wkey, bkey, acc_bias)
                    ev_out.record(ws)
                main.wait_event(ev_out)
                return dx
            part = self._buf('wgrad_part', (sp * N, K), torch.float32)
            bpart = self._buf('bgrad_part', (sp, N), torch.float32) if want_b else None
            _chk(_lib.load().sf_gemm_tn_pp(dy_b.data_ptr(), dy_b.stride(0), x_b.data_ptr(), x_b.stride(0), part.data_ptr(),
                                           bpart.data_ptr() if bpart is not None else None, M, N, K, sp, kc2, _st()), 'sf_gemm_tn_pp')
            self._wgrad_sum(part, bpart, sp, N, K, wkey, bkey, acc_bias)
            return self._lin_dgrad(dy_b, M, N, K, wkey, need_dx, dx_out, tag, acc_dx, dx_dtype)
        if tn:
            # dW straight from the row-major gradient / saved input (ds_read_b64_tr_b16 operand reads): no transposed copies
            # the bias gradient (column sums of dY) rides in the same launch: per-chunk partials from an all-ones MFMA in the workgroups of column tile 0
            part = self._buf('wgrad_part', (split * N, K), torch.float32)
            bpart = self._buf('bgrad_part', (split, N), torch.float32) if dy_f32 is None and not bias_done else None
            _chk(_lib.load().sf_gemm_tn_splitk(dy_b.data_ptr(), dy_b.stride(0), x_b.data_ptr(), x_b.stride(0), part.data_ptr(),
                                               bpart.data_ptr() if bpart is not None else None, M, N, K, split, kc, _st()), 'sf_gemm_tn_splitk')
            self._wgrad_sum(part, bpart, split, N, K, wkey, bkey, acc_bias)
            return self._lin_dgrad(dy_b, M, N, K, wkey, need_dx, dx_out, tag, acc_dx, dx_dtype)
        m_pad = kc * split if split > 1 else m_pad
        dyT = self._buf('dyT', (N, m_pad), torch.bfloat16)
        xT = self._buf('xT', (K, m_pad), torch.bfloat16)
        transpose(dy_b, dy_b.stride(0), 0, 0, dyT, m_pad, 0, 0, M, N, m_pad)
        transpose(x_b, x_b.stride(0), 0, 0, xT, m_pad, 0, 0, M, K, m_pad)
        if dy_f32 is None and not bias_done:                                         # bias gradient = row sums of dy^T (zero-padded columns add 0)
            _chk(_lib.load().sf_rowsum_bf16(dyT.data_ptr(), m_pad, N, m_pad, self.g[bkey].data_ptr(), int(acc_bias), _st()), 'sf_rowsum_bf16')
        if split > 1:
            part = self._buf('wgrad_part', (split * N, K), torch.float32)
            bgemm(dyT, m_pad, kc, 0, xT, m_pad, kc, 0, part, K, N * K, 0, N, K, kc, split, 1)
            _chk(_lib.load().sf_seqsum(part.data_ptr(), K, split, N, K, self.g[wkey].data_ptr(), 0, _st()), 'sf_seqsum')
        else:
            ops.gemm(dyT, xT, None, self.g[wkey].view(N, K), M=N)
        return self._lin_dgrad(dy_b, M, N, K, wkey, need_dx, dx_out, tag, acc_dx, dx_dtype)

    def _wgrad_stream(self):
        t = self._wgrad_streams.get(self._ws_prefix)
        if t is None:
            prio = int(os.environ.get('SF_WGRAD_PRIO', '1'))              # lower priority than the compute stream (clamped to the device's range by torch)
            t = self._wgrad_streams[self._ws_prefix] = (torch.cuda.Stream(device=self.dev, priority=prio), torch.cuda.Event(), torch.cuda.Event())
        return t

    def _wgrad_sum(self, part, bpart, split, N, K, wkey, bkey, acc_bias):
        """g[W] = sum of the split-K chunk planes, g[b] (=|+=) sum of the bias partials - one launch (sf_wgrad_sum)."""
        gw = self.g[wkey]
        if N % 4 == 0 and gw.data_ptr() % 16 == 0 and (bpart is None or self.g[bkey].data_ptr() % 16 == 0):
            _chk(_lib.load().sf_wgrad_sum(part.data_ptr(), N * K, split, gw.data_ptr(), bpart.data_ptr() if bpart is not None else None, N,
                                          self.g[bkey].data_ptr() if bpart is not None else None, int(acc_bias), _st()), 'sf_wgrad_sum')
            return
        _chk(_lib.load().sf_seqsum(part.data_ptr(), K, split, N, K, gw.data_ptr(), 0, _st()), 'sf_seqsum')       # (a gradient slot that is not 16-byte aligned)
        if bpart is not None:
            _chk(_lib.load().sf_seqsum(bpart.data_ptr(), N, split, 1, N, self.g[bkey].data_ptr(), int(acc_bias), _st()), 'sf_seqsum')

    def _lin_dgrad(self, dy_b, M, N, K, wkey, need_dx, dx_out, tag, acc_dx, dx_dtype=torch.float32):
        if not need_dx:
            return None
        wT = self._wT[wkey]                                                           # (K, n_pad)
        assert dx_dtype == torch.float32 or (dx_out is None and not acc_dx)
        dx = dx_out if dx_out is not None else self._buf('dx_' + tag + ('_b' if dx_dtype == torch.bfloat16 else ''), (M, K), dx_dtype)
        if wT.shape[1] != N:                                                           # ragged N (heads): zero-padded contraction
            dyp = self._buf('dy_pad', (M, wT.shape[1]), torch.bfloat16, zero=True)
            dyp[:, :N].copy_(dy_b[:M, :N])
            ops.gemm(dyp, wT, None, dx, M=M, residual=dx if acc_dx else None)
        else:
            ops.gemm(dy_b, wT, None, dx, M=M, residual=dx if acc_dx else None)
        return dx

    # ---- full self-attention backward over B contiguous sequences of L rows: five strided-batched products ----------------
    def attn_bwd_seq(self, qkv, dO_b, dqkv, B, L, H, hd, P_saved=None, seed=None):
        """qkv (B*L, 3*H*hd) bf16 = q | k | v side by side, dO_b (B*L, H*hd) bf16 -> dqkv (B*L, 3*H*hd) bf16 (all rows written).
        softmax(q k^T / sqrt(hd)) is recomputed unless the (pre-dropout) probabilities were saved."""
        D = H * hd
        Lp = ((L + 31) // 32) * 32
        scale = 1.0 / math.sqrt(hd)
        ld3 = qkv.stride(0)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        S = self._buf('att_S', (B * H * L, Lp), torch.float32)
        if P_saved is None:                                                             # recompute the probabilities
            bgemm(q, ld3, L * ld3, hd, k, ld3, L * ld3, hd, S, Lp, H * L * Lp, L * Lp, L, L, hd, B, H)
            P = self._buf('att_P', (B * H * L, Lp), torch.bfloat16)
            _chk(_lib.load().sf_softmax_rows(S.data_ptr(), Lp, P.data_ptr(), Lp, B * H * L, L, Lp, scale, _st()), 'sf_softmax_rows')
            Pv = P                                                                      # the matrix that multiplied V in the forward
        else:                                                                           # attn dropout: P saved, drop(P) regenerated
            P = P_saved
            Pv = self._buf('att_Pd', (B * H * L, Lp), torch.bfloat16, zero=True)
            dropout(P, Pv, B * H * L, L, self.attn_pdrop, seed)
        dP = S                                                                          # reuse: S is dead once P exists
        bgemm(dO_b, D, L * D, hd, v, ld3, L * ld3, hd, dP, Lp, H * L * Lp, L * Lp, L, L, hd, B, H)
        if P_saved is not None:
            dropout(dP, dP, B * H * L, L, self.attn_pdrop, seed)                         # d(drop(P)) -> dP through the same mask
        dS = self._buf('att_dS', (B * H * L, Lp), torch.bfloat16)
        _chk(_lib.load().sf_softmax_bwd_rows(P.data_ptr(), Lp, dP.data_ptr(), Lp, dS.data_ptr(), Lp, B * H * L, L, Lp, scale, _st()),
             'sf_softmax_bwd_rows')
        # transposed per-(clip, head) operands, contraction dimension zero-padded to Lp
        kT = self._buf('att_kT', (B * H * hd, Lp), torch.bfloat16)
        qT = self._buf('att_qT', (B * H * hd, Lp), torch.bfloat16)
        dOT = self._buf('att_dOT', (B * H * hd, Lp), torch.bfloat16)
        transpose(k, ld3, L * ld3, hd, kT, Lp, H * hd * Lp, hd * Lp, L, hd, Lp, B, H)
        transpose(q, ld3, L * ld3, hd, qT, Lp, H * hd * Lp, hd * Lp, L, hd, Lp, B, H)
        transpose(dO_b, D, L * D, hd, dOT, Lp, H * hd * Lp, hd * Lp, L, hd, Lp, B, H)
        dST = self._buf('att_dST', (B * H * L, Lp), torch.bfloat16)
        PT = self._buf('att_PT', (B * H * L, Lp), torch.bfloat16)
        transpose(dS, Lp, H * L * Lp, L * Lp, dST, Lp, H * L * Lp, L * Lp, L, L, Lp, B, H)
        transpose(Pv, Lp, H * L * Lp, L * Lp, PT, Lp, H * L * Lp, L * Lp, L, L, Lp, B, H)
        dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
        ldg = dqkv.stride(0)
        bgemm(dS, Lp, H * L * Lp, L * Lp, kT, Lp, H * hd * Lp, hd * Lp, dq, ldg, L * ldg, hd, L, hd, Lp, B, H)     # dQ = dS K
        bgemm(dST, Lp, H * L * Lp, L * Lp, qT, Lp, H * hd * Lp, hd * Lp, dk, ldg, L * ldg, hd, L, hd, Lp, B, H)    # dK = dS^T Q
        bgemm(PT, Lp, H * L * Lp, L * Lp, dOT, Lp, H * hd * Lp, hd * Lp, dv, ldg, L * ldg, hd, L, hd, Lp, B, H)    # dV = P^T dO

    time_comm, _comm_ev = False, None

    def allreduce_grads(self):
        """DDP-equivalent gradient averaging: one flat 90 MB bucket over RCCL (C1 in SURVEY §2.2)."""
        from .dist import allreduce_mean_
        if self.time_comm and torch.distributed.is_available() and torch.distributed.is_initialized():
            self._comm_ev = self._comm_ev or (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._comm_ev[0].record()
            allreduce_mean_(self.flat_g)
            self._comm_ev[1].record()
        else:
            allreduce_mean_(self.flat_g)

    def exposed_comm_ms(self) -> float:
        """Milliseconds of the last step's gradient all-reduce on the compute stream (0 without a process group)."""
        if self._comm_ev is None:
            return 0.0
        self._comm_ev[1].synchronize()
        return self._comm_ev[0].elapsed_time(self._comm_ev[1])

    def optimizer_step(self, lr: Optional[float] = None):
        """clip_grad_norm_(max_clip_norm) + Adam on the flat buffers (train_utils.py:373-386), then refresh operand copies."""
        lib = _lib.load()
        ws = self._buf('norm_ws', (1024,), torch.float32)
        _chk(lib.sf_grad_norm(self.flat_g.data_ptr(), self.n, self.norm.data_ptr(), ws.data_ptr(), _st()), 'sf_grad_norm')
        self.step_count += 1
        _chk(lib.sf_adam_clip_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                                   self.flat_b.data_ptr(), self.n, self.norm.data_ptr(), float(self.max_clip_norm or 0.0),
                                   float(self.lr if lr is None else lr), self.betas[0], self.betas[1], self.eps, self.step_count, _st()),
             'sf_adam_clip_step')
        self._refresh_transposed()


class SyncTrainer(FlatTrainer):
    def __init__(self, state_dict: Dict[str, torch.Tensor], device='cuda:0', lr: float = 2e-6, betas=(0.9, 0.999), eps: float = 1e-7,
                 max_clip_norm: float = 1.0, embd_pdrop: float = 0.0, resid_pdrop: float = 0.0, attn_pdrop: float = 0.0, seed: int = 1337,
                 seg_chunk: int = 224, engine: Optional[SynchformerEngine] = None, fp8_towers: bool = False, tok_pdrop: float = 0.0):
        self.embd_pdrop, self.resid_pdrop, self.attn_pdrop, self.seed = float(embd_pdrop or 0), float(resid_pdrop or 0), float(attn_pdrop or 0), seed
        self.tok_pdrop = float(tok_pdrop or 0)            # whole-token dropout of the segment tokens (Dropout1d, sync_model.py:131-134, 160-161); every config uses 0.0
        self.fwd_count = 0
        self.engine = engine if engine is not None else SynchformerEngine(state_dict, torch.device(device), seg_chunk=seg_chunk,
                                                                          fp8_towers=fp8_towers)   # frozen extractors (MXFP8 GEMMs in the FT configuration)
        self._init_flat(state_dict, trainable_keys(state_dict), device, lr, betas, eps, max_clip_norm)
        self.n_blocks = len([k for k in self.keys if k.endswith('.ln1.weight')])
        self.heads = 8
        self.head_name = 'off_head' if 'transformer.off_head.weight' in self.p else 'sync_head'
        self.n_out = self.p[f'transformer.{self.head_name}.weight'].shape[0]

    def _site_seed(self, site: int) -> int:
        """uint32 seed of dropout site `site` for the current forward pass (embd 0; block i: attn 1+3i, proj 2+3i, mlp 3+3i; whole-token dropout: 1000 vis, 1001 aud)."""
        h = (self.seed * 0x9E3779B1 + self.fwd_count * 0x85EBCA6B + site * 0xC2B2AE35 + 0x165667B1) & 0xFFFFFFFF
        h ^= h >> 15
        return (h * 0x2C1B3C6D) & 0xFFFFFFFF

    def _token_scales(self, tag: str, n: int, seed: int) -> torch.Tensor:
        """(n,) fp32: 0 or 1 / (1 - tok_pdrop) per token - sf_dropout over a vector of ones, so the backward (and a recomputed forward) regenerates it from the seed."""
        w = ((n + 3) // 4) * 4
        ones = self._buf('tok_ones', (1, w), torch.float32)
        ones.fill_(1.0)
        sc = self._buf(f'{tag}_tok_scale', (1, w), torch.float32)
        dropout(ones, sc, 1, w, self.tok_pdrop, seed)
        return sc

    # ---- forward with saved activations ----------------------------------------------------------------------
    def _forward(self, vfeat, afeat):
        B = vfeat.shape[0]
        Sv, Sa = vfeat.shape[1] * vfeat.shape[2], afeat.shape[1] * afeat.shape[2]
        L = 2 + Sv + Sa
        M = B * L
        sv = self.sv = dict(B=B, Sv=Sv, Sa=Sa, L=L, M=M)
        # projections (vproj / aproj) on the frozen features
        for tag, feat, n_tok in (('v', vfeat, Sv), ('a', afeat, Sa)):
            fb = self._buf(f'{tag}_in', (B * n_tok, D), torch.bfloat16)
            ops.gather_rows(feat.reshape(B * n_tok, D), fb, B * n_tok)
            w, b = self._wb(f'{tag}proj')
            pr = self._buf(f'{tag}_proj', (B * n_tok, D), torch.float32)
            ops.gemm(fb, w, b, pr)
            sv[f'{tag}_in'], sv[f'{tag}_proj'] = fb, pr
        t = 'transformer'
        table = self.p[f'{t}.pos_emb_cfg.pos_emb'][0, :L].clone()
        table[0] += self.p[f'{t}.OFF_tok'][0, 0]
        table[1 + Sv] += self.p[f'{t}.MOD_tok'][0, 0]
        x = self._buf('x0', (M, D), torch.float32)
        ops.broadcast_rows(x, table.contiguous(), n_seq=B, dst_seq_rows=L)
        self.fwd_count += 1
        for tag, ln, n_tok, off in (('v', 'vis_in_lnorm', Sv, 1), ('a', 'aud_in_lnorm', Sa, 2 + Sv)):
            tokmap = ops.rowmap(n_tok, n_tok, L, 0, 1, off)
            if self.tok_pdrop > 0:
                # v, a = tok_drop_vis(v), tok_drop_aud(a) between the input norms and the concat (sync_model.py:158-163): Dropout1d on (B, S, D) zeroes whole tokens
                # and scales the kept ones by 1 / (1 - p).  The norm's rows go to a buffer of their own and enter the token matrix through sf_scale_rows_map.
                n = B * n_tok
                lnout = self._buf(f'{tag}_lnout', (n, D), torch.float32)
                ops.layernorm(sv[f'{tag}_proj'], self.p[f'{t}.{ln}.weight'], self.p[f'{t}.{ln}.bias'], lnout, EPS_SYNC)
                sc = sv[f'{tag}_tok_scale'] = self._token_scales(tag, n, self._site_seed(1000 + (tag == 'a')))
                _chk(_lib.load().sf_scale_rows_map(lnout.data_ptr(), D, None, sc.data_ptr(), x.data_ptr(), D, ops._map(tokmap), n, D, 1, _st()), 'sf_scale_rows_map')
            else:
                ops.layernorm(sv[f'{tag}_proj'], self.p[f'{t}.{ln}.weight'], self.p[f'{t}.{ln}.bias'], x, EPS_SYNC, out_map=tokmap, accumulate=True)
        if self.embd_pdrop > 0:                                                          # self.drop(x), sync_model.py:166
            sv['embd_seed'] = self._site_seed(0)
            dropout(x, x, M, D, self.embd_pdrop, sv['embd_seed'])
        hd = D // self.heads
        sv['blocks'] = []
        for i in range(self.n_blocks):
            p = f'{t}.blocks.{i}'
            s = dict(x=x)
            s['h1'] = self._buf(f'h1_{i}', (M, D), torch.bfloat16)
            ops.layernorm(x, self.p[p + '.ln1.weight'], self.p[p + '.ln1.bias'], s['h1'], EPS_SYNC)
            s['qkv'] = self._buf(f'qkv_{i}', (M, 3 * D), torch.bfloat16)
            for j, n in enumerate(('query', 'key', 'value')):                           # separate Linears, packed side by side
                w, b = self._wb(f'{p}.attn.{n}')
                ops.gemm(s['h1'], w, b, s['qkv'][:, j * D:(j + 1) * D])
            s['att'] = self._buf(f'att_{i}', (M, D), torch.bfloat16)
            q3 = s['qkv']
            if self.attn_pdrop > 0:                                                       # explicit softmax -> dropout -> P V
                s['attn_seed'] = self._site_seed(1 + 3 * i)
                s['P'] = self._attn_fwd_dropout(q3, s['att'], s['attn_seed'], i)
            else:
                ops.attention(q3[:, :D], q3[:, D:2 * D], q3[:, 2 * D:], s['att'], n_seq=B, seq_rows=L, n_groups=1, row0=0, group_stride=0,
                              tok_stride=1, n_tok=L, cls_row=-1, heads=self.heads, head_dim=hd, scale=1.0 / math.sqrt(hd))
            w, b = self._wb(p + '.attn.proj')
            s['x2'] = self._buf(f'x2_{i}', (M, D), torch.float32)
            if self.resid_pdrop > 0:                                                      # x + resid_drop(proj(y)), transformer.py:73,95
                s['proj_seed'] = self._site_seed(2 + 3 * i)
                tmp = self._buf('branch_tmp', (M, D), torch.float32)
                ops.gemm(s['att'], w, b, tmp)
                dropout(tmp, s['x2'], M, D, self.resid_pdrop, s['proj_seed'], residual=x)
            else:
                ops.gemm(s['att'], w, b, s['x2'], residual=x)
            s['h2'] = self._buf(f'h2_{i}', (M, D), torch.bfloat16)
            ops.layernorm(s['x2'], self.p[p + '.ln2.weight'], self.p[p + '.ln2.bias'], s['h2'], EPS_SYNC)
            w, b = self._wb(p + '.mlp.0')
            s['pre'] = self._buf(f'pre_{i}', (M, FF), torch.bfloat16)
            ops.gemm(s['h2'], w, b, s['pre'])
            s['act'] = self._buf(f'act_{i}', (M, FF), torch.bfloat16)
            _chk(_lib.load().sf_gelu_fwd(s['pre'].data_ptr(), s['act'].data_ptr(), M * FF, _st()), 'sf_gelu_fwd')
            w, b = self._wb(p + '.mlp.2')
            x = self._buf(f'xo_{i}', (M, D), torch.float32)
            if self.resid_pdrop > 0:                                                      # mlp[3] = Dropout(resid_pdrop), transformer.py:90
                s['mlp_seed'] = self._site_seed(3 + 3 * i)
                tmp = self._buf('branch_tmp', (M, D), torch.float32)
                ops.gemm(s['act'], w, b, tmp)
                dropout(tmp, x, M, D, self.resid_pdrop, s['mlp_seed'], residual=s['x2'])
            else:
                ops.gemm(s['act'], w, b, x, residual=s['x2'])
            sv['blocks'].append(s)
        sv['x_last'] = x
        cls = self._buf('cls_n', (B, D), torch.bfloat16)
        ops.layernorm(x, self.p[f'{t}.ln_f.weight'], self.p[f'{t}.ln_f.bias'], cls, EPS_SYNC, rows=B, in_map=ops.rowmap(1, 1, L, 0, 0, 0))
        sv['cls_n'] = cls
        w, b = self._wb(f'{t}.{self.head_name}')
        logits = self._buf('logits', (B, self.n_out), torch.float32)
        ops.gemm(cls, w, b, logits, M=B)
        return logits

    # ---- attention forward with probability dropout (train mode only) -------------------------------------------
    def _attn_fwd_dropout(self, qkv, att, seed, blk):
        sv = self.sv
        B, L, H = sv['B'], sv['L'], self.heads
        hd = D // H
        Lp = ((L + 31) // 32) * 32
        ld3 = qkv.stride(0)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        S = self._buf('att_S', (B * H * L, Lp), torch.float32)
        bgemm(q, ld3, L * ld3, hd, k, ld3, L * ld3, hd, S, Lp, H * L * Lp, L * Lp, L, L, hd, B, H)
        P = self._buf(f'att_Psave_{blk}', (B * H * L, Lp), torch.bfloat16)
        _chk(_lib.load().sf_softmax_rows(S.data_ptr(), Lp, P.data_ptr(), Lp, B * H * L, L, Lp, 1.0 / math.sqrt(hd), _st()), 'sf_softmax_rows')
        Pd = self._buf('att_Pd', (B * H * L, Lp), torch.bfloat16, zero=True)
        dropout(P, Pd, B * H * L, L, self.attn_pdrop, seed)
        vT = self._buf('att_vT', (B * H * hd, Lp), torch.bfloat16)
        transpose(v, ld3, L * ld3, hd, vT, Lp, H * hd * Lp, hd * Lp, L, hd, Lp, B, H)
        bgemm(Pd, Lp, H * L * Lp, L * Lp, vT, Lp, H * hd * Lp, hd * Lp, att, D, L * D, hd, L, hd, Lp, B, H)        # att = drop(P) V
        return P

    # ---- attention backward: five strided-batched products per block -------------------------------------------
    def _attn_bwd(self, qkv, dO_b, dqkv, P_saved=None, seed=None):
        sv = self.sv
        self.attn_bwd_seq(qkv, dO_b, dqkv, sv['B'], sv['L'], self.heads, D // self.heads, P_saved, seed)

    # ---- backward ------------------------------------------------------------------------------------------------
    def _backward(self, dlogits):
        sv = self.sv
        B, L, M, Sv, Sa = sv['B'], sv['L'], sv['M'], sv['Sv'], sv['Sa']
        t = 'transformer'
        lnws = self._buf('ln_ws', (2 * 768 * ((M + 3) // 4),), torch.float32)
        # head: logits = cls_n W^T + b
        dl_b = self._buf('dlogits_b', (B, 64), torch.bfloat16, zero=True)
        dl_b[:, :self.n_out].copy_(dlogits)
        head = f'{t}.{self.head_name}'
        dcls = self._lin_bwd(head, dl_b[:, :self.n_out], sv['cls_n'], B, tag='cls')      # (B, 768) fp32
        # ln_f on row 0 of each sequence; every other row of dX is zero
        dx = self._buf('dx_a', (M, D), torch.float32, zero=True)
        ln_bwd(sv['x_last'], self.p[f'{t}.ln_f.weight'], dcls, dx, self.g[f'{t}.ln_f.weight'], self.g[f'{t}.ln_f.bias'], lnws, B, EPS_SYNC,
               x_map=ops.rowmap(1, 1, L, 0, 0, 0), dx_map=ops.rowmap(1, 1, L, 0, 0, 0))
        dy_b = self._buf('dy_b', (M, FF), torch.bfloat16)
        for i in reversed(range(self.n_blocks)):
            p, s = f'{t}.blocks.{i}', sv['blocks'][i]
            # y = x2 + drop(act W2^T + b2)
            self._branch_grad(dx, dy_b, M, s.get('mlp_seed'))
            dact = self._lin_bwd(p + '.mlp.2', dy_b[:, :D], s['act'], M, tag='act')       # (M, 3072) fp32
            dpre = self._buf('dpre', (M, FF), torch.bfloat16)
            _chk(_lib.load().sf_gelu_bwd(s['pre'].data_ptr(), dact.data_ptr(), dpre.data_ptr(), M * FF, _st()), 'sf_gelu_bwd')
            dh2 = self._lin_bwd(p + '.mlp.0', dpre, s['h2'], M, tag='h')                   # (M, 768) fp32
            # dx2 = dy + LN2'(dh2)
            ln_bwd(s['x2'], self.p[p + '.ln2.weight'], dh2, dx, self.g[p + '.ln2.weight'], self.g[p + '.ln2.bias'], lnws, M, EPS_SYNC, acc_dx=True)
            # x2 = x + drop(att Wp^T + bp)
            self._branch_grad(dx, dy_b, M, s.get('proj_seed'))
            datt = self._lin_bwd(p + '.attn.proj', dy_b[:, :D], s['att'], M, tag='h')     # (M, 768) fp32
            dO_b = self._buf('dO_b', (M, D), torch.bfloat16)
            cast_bf16(datt, dO_b, M, D)
            dqkv = self._buf('dqkv', (M, 3 * D), torch.bfloat16)
            self._attn_bwd(s['qkv'], dO_b, dqkv, s.get('P'), s.get('attn_seed'))
            dh1 = self._buf('dh1', (M, D), torch.float32)
            for j, n in enumerate(('query', 'key', 'value')):
                part = self._lin_bwd(f'{p}.attn.{n}', dqkv[:, j * D:(j + 1) * D], s['h1'], M, tag='h')
                if j == 0:
                    dh1.copy_(part)
                else:
                    dh1.add_(part)                                                          # torch elementwise add: 3 small launches / block
            ln_bwd(s['x'], self.p[p + '.ln1.weight'], dh1, dx, self.g[p + '.ln1.weight'], self.g[p + '.ln1.bias'], lnws, M, EPS_SYNC, acc_dx=True)
        # x0 = drop(table + scatter(LN_v(pv)) + scatter(LN_a(pa)))
        if 'embd_seed' in sv:
            dropout(dx, dx, M, D, self.embd_pdrop, sv['embd_seed'])
        gtab = self._buf('gtab', (L, D), torch.float32)
        _chk(_lib.load().sf_seqsum(dx.data_ptr(), D, B, L, D, gtab.data_ptr(), 0, _st()), 'sf_seqsum')
        gpos = self.g[f'{t}.pos_emb_cfg.pos_emb']
        gpos.zero_()
        gpos[0, :L].copy_(gtab)
        self.g[f'{t}.OFF_tok'][0, 0].copy_(gtab[0])
        self.g[f'{t}.MOD_tok'][0, 0].copy_(gtab[1 + Sv])
        for tag, ln, n_tok, off in (('v', 'vis_in_lnorm', Sv, 1), ('a', 'aud_in_lnorm', Sa, 2 + Sv)):
            dpr = self._buf('dproj', (B * n_tok, D), torch.float32)
            if f'{tag}_tok_scale' in sv:                                               # whole-token dropout: d(norm rows) = scale[token] * d(token matrix rows)
                dln = self._buf(f'{tag}_dlnout', (B * n_tok, D), torch.float32)
                _chk(_lib.load().sf_scale_rows_map(dx.data_ptr(), D, ops._map(ops.rowmap(n_tok, n_tok, L, 0, 1, off)), sv[f'{tag}_tok_scale'].data_ptr(),
                                                   dln.data_ptr(), D, None, B * n_tok, D, 0, _st()), 'sf_scale_rows_map')
                ln_bwd(sv[f'{tag}_proj'], self.p[f'{t}.{ln}.weight'], dln, dpr, self.g[f'{t}.{ln}.weight'], self.g[f'{t}.{ln}.bias'], lnws, B * n_tok, EPS_SYNC)
            else:
                ln_bwd(sv[f'{tag}_proj'], self.p[f'{t}.{ln}.weight'], dx, dpr, self.g[f'{t}.{ln}.weight'], self.g[f'{t}.{ln}.bias'], lnws, B * n_tok,
                       EPS_SYNC, dy_map=ops.rowmap(n_tok, n_tok, L, 0, 1, off))
            cast_bf16(dpr, dy_b[:B * n_tok, :D], B * n_tok, D)
            self._lin_bwd(f'{tag}proj', dy_b[:B * n_tok, :D], sv[f'{tag}_in'], B * n_tok, need_dx=False)

    def _branch_grad(self, dx, dy_b, M, seed):
        """bf16 gradient of a residual branch output: dy_b[:, :768] = bf16(mask(dx)) (mask only when that branch had dropout)."""
        if seed is None:
            cast_bf16(dx, dy_b[:, :D], M, D)
        else:
            tmp = self._buf('branch_tmp', (M, D), torch.float32)
            dropout(dx, tmp, M, D, self.resid_pdrop, seed)
            cast_bf16(tmp, dy_b[:, :D], M, D)

    # ---- public API ----------------------------------------------------------------------------------------------
    def forward_backward(self, vfeat: torch.Tensor, afeat: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        """Segment features (B,S,tv,768)/(B,S,ta,768) fp32 + int64 targets -> loss (device scalar); fills the flat gradient."""
        logits = self._forward(vfeat, afeat)
        B = logits.shape[0]
        dlogits = self._buf('dlogits', (B, self.n_out), torch.float32)
        _chk(_lib.load().sf_cross_entropy(logits.data_ptr(), logits.stride(0), targets.data_ptr(), B, self.n_out, self.loss.data_ptr(),
                                          dlogits.data_ptr(), dlogits.stride(0), 1.0, _st()), 'sf_cross_entropy')
        self._backward(dlogits)
        self.logits = logits
        return self.loss

    def train_step(self, vis: torch.Tensor, aud: torch.Tensor, targets: torch.Tensor, lr: Optional[float] = None) -> torch.Tensor:
        """One Stage-2 iteration (train_sync.py:159-192): frozen extractors -> trainable forward/backward -> all-reduce -> clip+Adam."""
        vf, af = self.engine.both_towers(lambda: self.engine.extract_vfeats(vis), aud)
        loss = self.forward_backward(vf, af, targets)
        self.allreduce_grads()
        self.optimizer_step(lr)
        return loss


class SyncTrainFunction(torch.autograd.Function):
    """autograd bridge: logits = f(frozen features; trainable params).  backward runs the HIP backward and hands the
    per-parameter gradients to autograd, so `scaler.scale(loss).backward()`, DistributedDataParallel's reducer hooks and
    torch.optim.Adam of the reference loop (train_utils.py:373-386) work on the module's nn.Parameters unchanged."""

    @staticmethod
    def forward(ctx, trainer: SyncTrainer, vfeat, afeat, *params):
        ctx.trainer = trainer
        ctx.fwd_count0 = trainer.fwd_count                         # the dropout masks of this pass are a function of (seed, this counter, site)
        ctx.save_for_backward(vfeat, afeat)                        # (B, S, t, 768) frozen features: a few MB - kept for a recomputation, see backward
        logits = trainer._forward(vfeat, afeat).clone()
        trainer.generation = ctx.generation = getattr(trainer, 'generation', 0) + 1
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        tr = ctx.trainer
        if ctx.generation != tr.generation:
            # The saved activations live in the trainer's shared workspaces and a later grad-enabled forward of the same module has overwritten them
            # (forward, forward, backward, backward - e.g. two losses of one step).  nn.Module semantics ask for this to work, so the forward is re-run from
            # the kept inputs under the SAME dropout masks (counter restored) before the backward - activation checkpointing, paid only by this call pattern.
            vfeat, afeat = ctx.saved_tensors
            keep = tr.fwd_count
            tr.fwd_count = ctx.fwd_count0
            tr._forward(vfeat, afeat)
            tr.fwd_count = keep
            tr.generation += 1                                     # the workspaces now belong to THIS pass; any other pending pass recomputes in its turn
        tr._backward(dlogits.contiguous().float())
        return (None, None, None) + tuple(tr.g[k].clone() for k in tr.keys)
