#!/usr/bin/env python
"""Headline benchmark: clips/sec of Synchformer offset prediction (14 x 0.64 s segments, 21 offset classes).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
(a plain `python bench.py --gpus N` with N > 1 re-executes itself through torch.distributed.run, one rank per GPU.)

Workload = BASELINE.json configs[1]: batched offset inference, configs/sync.yaml model, random-init weights,
synthetic 224x224 @ 25 fps uint8 frames + 128x66 log-mel spectrograms already resident in HBM.  One "step" = one
full Synchformer.forward() (RGB front-end -> Motionformer -> AST -> sync transformer -> logits) over B clips per GPU.
Multi-GPU: inference shards by clip with no data-path collective ("replicas only"): every rank runs
its own B clips; value = all clips / max-over-ranks time (weak scaling).
Prints ONE JSON line on rank 0 with
  * `roofline`: every GEMM-family kernel of the step timed live with HIP events on the launch stream in a second pass of the same K steps
    (`roofline.kernels`, one entry per kernel symbol as rocprofv3 names it; `roofline.kernel` = the dominant one by time),
  * `cpu_baseline`: the CPU oracle timed on this box's host cores (N = 1 only),
  * `workloads`: BASELINE configs[2] / [3] / [4] (Stage-2 train step, Stage-1 AVCLIP train step at 2 clips per GPU, synchronizability fine-tune on
    MXFP8 extractor GEMMs), a few steps each AFTER the headline region - on N > 1 these run their RCCL collectives (gradient all-reduce, bucketed
    overlap); a watchdog prints the headline line without them if they do not finish.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

FLOP_PER_CLIP = 5.725e12          # algorithmic forward FLOPs per 14-segment clip
# FLOPs the engine actually executes per clip: both aggregator layers are computed for output row 0 only (motionformer.py:332 reads nothing
# else), which drops 14 x 8 x (out_proj 0.232 + MLP 1.859 + the 196 unused query rows of the attention 0.118) G = 0.247 T of the visual
# aggregator and 14 x 6 x 0.17 G of the audio one
FLOP_PER_CLIP_EXECUTED = 5.725e12 - 0.247e12 - 0.014e12
PEAK_BF16 = 2.5e15                # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_MXFP8 = 5.0e15               # dense MX-fp8 MFMA peak
HBM_ACHIEVABLE = 6.3e12           # achievable HBM3E stream rate (MI355X_MICROARCH.md: 6.29 TB/s measured of 8 TB/s)
TRAFFIC_FILE = 'profiles/r06_bench_roofline.json'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=None, help='clips per GPU per step (default 16 = configs/sync.yaml; stage1: 2 = segment_avclip.yaml)')
    ap.add_argument('--seg-chunk', type=int, default=224)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-workloads', action='store_true', help='headline workload only (no configs[2]/[3]/[4] section)')
    ap.add_argument('--workload-steps', type=int, default=5)
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL over xGMI; default) | gloo (plumbing tests)')
    ap.add_argument('--single-device', action='store_true', help='TEST ONLY: put every rank on cuda:0 (needs --dist-backend gloo)')
    ap.add_argument('--workload', choices=['infer', 'train', 'stage1', 'ft'], default='infer',
                    help="infer = BASELINE configs[1] (headline metric); train = configs[2]: Stage-2 step, frozen extractors, "
                         "backward + RCCL gradient all-reduce + fused clip/Adam (dropout 0.1 as in sync.yaml); stage1 = configs[3]; ft = configs[4]: the "
                         "synchronizability fine-tune step (13 segments, 2-way sync head) with the frozen extractors' Linears on MXFP8")
    ap.add_argument('--graph', action='store_true', help='replay the forward as one captured HIP graph (infer workload only)')
    ap.add_argument('--dropin', action='store_true', help="train workload through the drop-in nn.Module + torch.optim.Adam + GradScaler "
                    "(the reference's loop body, train_utils.py:373-386) instead of SyncTrainer's fused step: the wrapper overhead as a number")
    return ap.parse_args()


def respawn_distributed(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks through torch.distributed.run (one per GPU, rendezvous on 127.0.0.1)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    raise SystemExit(subprocess.run(cmd).returncode)


import torch  # noqa: E402


# ---- live kernel timing ---------------------------------------------------------------------------------------------------------------------
class GemmTimer:
    """Brackets every GEMM-family launch (sf_gemm_bf16, sf_gemm_res_ln768, sf_qkv_time_attention, sf_gemm_mxfp8, sf_gemm_mx_res_ln768) with HIP events on the launch
    stream (= torch's current stream) and files it under the kernel symbol rocprofv3 reports for it."""

    def __init__(self):
        from synchformer_amd import ops
        self.ops = ops
        self.records = []          # (e0, e1, flop, algorithmic bytes, kernel symbol, shape label, family)
        self.enabled = False

    @staticmethod
    def _gemm_symbol(m, n, k, out_bf16, gelu, res, mapped):
        from synchformer_amd import _lib
        cfg = 0 if mapped else _lib.load().sf_gemm_bf16_auto_config(m, n, k, int(res))     # the library's own choice (tile-round fill decides at small batches)
        b = lambda v: 'true' if v else 'false'
        if cfg == 11:
            return f'gemm_bf16_pp_kernel<{b(out_bf16)}, {b(gelu)}, {b(res)}, false>'
        if cfg == 7:
            return f'gemm_bf16_persistent_kernel<{b(out_bf16)}, {b(gelu)}, {b(res)}>'
        return 'gemm_bf16_kernel<GemmCfg<128, 128, 2, 2, 64, 2, 2, false>, ...> (small / mapped GEMMs: AST, aggregators, sync transformer, heads)'

    def _rec(self, fn, flop, nbytes, sym, shape, family='bf16'):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.records.append((e0, e1, flop, nbytes, sym, shape, family))
        return r

    def __enter__(self):
        ops = self.ops
        self.orig = {k: getattr(ops, k) for k in ('gemm', 'gemm_res_ln', 'gemm_mxfp8', 'gemm_mx_res_ln', 'qkv_time_attention', 'qkv_time_attention2', 'qkv_time_attention_mx', 'qkv_space_attention', 'qkv_space_attention_mx')}
        o = self.orig

        def timed(a, w, bias, out, *, M=None, **kw):
            if not self.enabled:
                return o['gemm'](a, w, bias, out, M=M, **kw)
            m = a.shape[0] if M is None else M
            n, k = (w.shape[1], w.shape[0] * 64) if w.dim() == 3 else w.shape
            res = kw.get('residual')
            nbytes = m * k * 2 + n * k * 2 + m * n * out.element_size() + (m * n * 4 if res is not None else 0)   # A + W + C (+ R), each once
            sym = self._gemm_symbol(m, n, k, out.dtype == torch.bfloat16, bool(kw.get('gelu')), res is not None,
                                    kw.get('c_map') is not None or kw.get('r_map') is not None)
            return self._rec(lambda: o['gemm'](a, w, bias, out, M=M, **kw), 2.0 * m * n * k, nbytes, sym, f'N={n} K={k}')

        def timed_ln(a, w, bias, x, gamma, beta, y, eps, *, M=None, residual=None, **kw):
            if not self.enabled:
                return o['gemm_res_ln'](a, w, bias, x, gamma, beta, y, eps, M=M, residual=residual, **kw)
            m = a.shape[0] if M is None else M
            n, k = 768, a.shape[1]                                                     # w is (768, K) or k-step-major (K/32, 768, 32)
            nbytes = m * k * 2 + n * k * 2 + m * n * (4 + 4 + 2)                       # A + W + R read, X + Y written, each once
            return self._rec(lambda: o['gemm_res_ln'](a, w, bias, x, gamma, beta, y, eps, M=M, residual=residual, **kw), 2.0 * m * n * k, nbytes,
                             'gemm_res_ln768_kernel<0, true>', f'K={k}')

        def timed_mx(a_q, a_s, w_q, w_s, bias, out, *, M=None, residual=None, gelu=False, out_scales=None):
            if not self.enabled:
                return o['gemm_mxfp8'](a_q, a_s, w_q, w_s, bias, out, M=M, residual=residual, gelu=gelu, out_scales=out_scales)
            m = a_q.shape[0] if M is None else M
            n, k = w_q.shape
            nbytes = m * k + n * k + (m + n) * k // 32 + m * n * out.element_size() + (m * n * 4 if residual is not None else 0)
            return self._rec(lambda: o['gemm_mxfp8'](a_q, a_s, w_q, w_s, bias, out, M=M, residual=residual, gelu=gelu, out_scales=out_scales),
                             2.0 * m * n * k, nbytes, 'gemm_mxfp8_pp_kernel' if k % 256 == 0 else 'gemm_mxfp8_persistent_kernel', f'N={n} K={k}', 'mxfp8')

        def timed_mxln(a_q, a_s, w_q, w_s, bias, x, gamma, beta, y_q, y_s, eps, *, M=None, residual=None, **kw):
            if not self.enabled:
                return o['gemm_mx_res_ln'](a_q, a_s, w_q, w_s, bias, x, gamma, beta, y_q, y_s, eps, M=M, residual=residual, **kw)
            m = a_q.shape[0] if M is None else M
            n, k = 768, a_q.shape[1]
            nbytes = m * k + n * k + (m + n) * k // 32 + m * n * (4 + 4 + 1) + m * n // 32     # A + W + scales + R read, X + Y + Y's scales written
            return self._rec(lambda: o['gemm_mx_res_ln'](a_q, a_s, w_q, w_s, bias, x, gamma, beta, y_q, y_s, eps, M=M, residual=residual, **kw),
                             2.0 * m * n * k, nbytes, 'gemm_mx_res_ln768_kernel', f'K={k}', 'mxfp8')

        def timed_qt(x, w, bias, qkv_cls, out, partials, *, n_seq, n_groups, scale, **kw):
            if not self.enabled:
                return o['qkv_time_attention'](x, w, bias, qkv_cls, out, partials, n_seq=n_seq, n_groups=n_groups, scale=scale, **kw)
            m, n, k = n_seq * 8 * n_groups, 2304, 768
            nbytes = m * k * 2 + n * k * 2 + m * 768 * 2                               # A + W read, the 768-wide attention output written
            return self._rec(lambda: o['qkv_time_attention'](x, w, bias, qkv_cls, out, partials, n_seq=n_seq, n_groups=n_groups, scale=scale, **kw),
                             2.0 * m * n * k, nbytes, 'qkv_time_attn_kernel<true, false>', 'N=2304 K=768')

        def timed_qt_mx(x_q, x_s, w_q, w_s, bias, qkv_cls, out, partials, *, n_seq, n_groups, scale, **kw):
            if not self.enabled:
                return o['qkv_time_attention_mx'](x_q, x_s, w_q, w_s, bias, qkv_cls, out, partials, n_seq=n_seq, n_groups=n_groups, scale=scale, **kw)
            m, n, k = n_seq * 8 * n_groups, 2304, 768
            nbytes = m * k + n * k + (m + n) * k // 32 + m * 768 * 2
            return self._rec(lambda: o['qkv_time_attention_mx'](x_q, x_s, w_q, w_s, bias, qkv_cls, out, partials, n_seq=n_seq, n_groups=n_groups, scale=scale, **kw),
                             2.0 * m * n * k, nbytes, 'qkv_time_attn_kernel<true, true>', 'N=2304 K=768', 'mxfp8')

        def timed_qs(x, w, bias, side, out, partials, *, n_seq, scale, n_tok=196, key_keep=None):
            if not self.enabled or key_keep is not None:
                return o['qkv_space_attention'](x, w, bias, side, out, partials, n_seq=n_seq, scale=scale, n_tok=n_tok, key_keep=key_keep)
            m, n, k = n_seq * 8 * 192, 2304, 768                                       # the rows the launch projects itself (the other 33 per sequence: the side GEMM)
            nbytes = n_seq * 8 * n_tok * k * 2 + n * k * 2 + n_seq * 8 * n_tok * 768 * 2   # A + W read, the 768-wide attention output written
            return self._rec(lambda: o['qkv_space_attention'](x, w, bias, side, out, partials, n_seq=n_seq, scale=scale, n_tok=n_tok),
                             2.0 * m * n * k, nbytes, 'qkv_space_attn_kernel', 'N=2304 K=768')

        def timed_qs_mx(x_q, x_s, w_q, w_s, bias, side, out, partials, *, n_seq, scale, out_scales=None, n_tok=196):
            if not self.enabled:
                return o['qkv_space_attention_mx'](x_q, x_s, w_q, w_s, bias, side, out, partials, n_seq=n_seq, scale=scale, out_scales=out_scales, n_tok=n_tok)
            m, n, k = n_seq * 8 * 192, 2304, 768
            nbytes = n_seq * 8 * n_tok * k + n * k + (n_seq * 8 * n_tok + n) * k // 32 + n_seq * 8 * n_tok * 768 * out.element_size()
            return self._rec(lambda: o['qkv_space_attention_mx'](x_q, x_s, w_q, w_s, bias, side, out, partials, n_seq=n_seq, scale=scale, out_scales=out_scales, n_tok=n_tok),
                             2.0 * m * n * k, nbytes, 'qkv_space_attn_mx_kernel', 'N=2304 K=768', 'mxfp8')

        def timed_qt2(x, w, bias, side, out, partials, *, n_seq, scale, n_tok=196, key_keep=None):
            if not self.enabled or key_keep is not None:
                return o['qkv_time_attention2'](x, w, bias, side, out, partials, n_seq=n_seq, scale=scale, n_tok=n_tok, key_keep=key_keep)
            m, n, k = n_seq * 8 * 192, 2304, 768                                       # the rows the launch projects itself (the other 33 per sequence: the side GEMM)
            nbytes = n_seq * 8 * n_tok * k * 2 + n * k * 2 + n_seq * 8 * n_tok * 768 * 2   # A + W read, the 768-wide attention output written
            return self._rec(lambda: o['qkv_time_attention2'](x, w, bias, side, out, partials, n_seq=n_seq, scale=scale, n_tok=n_tok),
                             2.0 * m * n * k, nbytes, 'qkv_time2_attn_kernel', 'N=2304 K=768')

        ops.qkv_time_attention2 = timed_qt2
        ops.qkv_space_attention = timed_qs
        ops.qkv_space_attention_mx = timed_qs_mx
        ops.gemm, ops.gemm_res_ln, ops.gemm_mxfp8, ops.gemm_mx_res_ln, ops.qkv_time_attention = timed, timed_ln, timed_mx, timed_mxln, timed_qt
        ops.qkv_time_attention_mx = timed_qt_mx
        # the train steps call three GEMM entry points straight on the C ABI (weight gradients, fc1 + GELU with two outputs): wrap those on the library object
        from synchformer_amd import _lib
        lib = self.lib = _lib.load()
        self.lib_orig = {n: getattr(lib, n) for n in ('sf_gemm_tn_pp', 'sf_gemm_tn_splitk', 'sf_gemm_bf16_gelu_dual')}
        lo = self.lib_orig

        def tn_pp(dy, ldy, x, ldx, part, bpart, M, N, K, split, kc, st):
            if not self.enabled:
                return lo['sf_gemm_tn_pp'](dy, ldy, x, ldx, part, bpart, M, N, K, split, kc, st)
            return self._rec(lambda: lo['sf_gemm_tn_pp'](dy, ldy, x, ldx, part, bpart, M, N, K, split, kc, st), 2.0 * M * N * K, M * (N + K) * 2 + split * N * K * 4,
                             'gemm_tn_pp_kernel', f'N={N} K={K}')

        def tn_splitk(dy, ldy, x, ldx, part, bpart, M, N, K, split, kc, st):
            if not self.enabled:
                return lo['sf_gemm_tn_splitk'](dy, ldy, x, ldx, part, bpart, M, N, K, split, kc, st)
            return self._rec(lambda: lo['sf_gemm_tn_splitk'](dy, ldy, x, ldx, part, bpart, M, N, K, split, kc, st), 2.0 * M * N * K,
                             M * (N + K) * 2 + split * N * K * 4, 'gemm_tn_splitk_kernel', f'N={N} K={K}')

        def gelu_dual(a, lda, w, ldw, bias, pre, act, ldc, M, N, K, st):
            if not self.enabled:
                return lo['sf_gemm_bf16_gelu_dual'](a, lda, w, ldw, bias, pre, act, ldc, M, N, K, st)
            box = {}

            def run():
                box['rc'] = lo['sf_gemm_bf16_gelu_dual'](a, lda, w, ldw, bias, pre, act, ldc, M, N, K, st)
                return box['rc']
            n0 = len(self.records)
            rc = self._rec(run, 2.0 * M * N * K, M * K * 2 + N * K * 2 + 2 * M * N * 2, 'gemm_bf16_pp_kernel<true, true, false, true>', f'N={N} K={K}')
            if rc == -2:                                               # SF_NOT_APPLICABLE: outside config 11's range: nothing was launched (the caller takes the two-launch path)
                del self.records[n0:]
            return rc

        lib.sf_gemm_tn_pp, lib.sf_gemm_tn_splitk, lib.sf_gemm_bf16_gelu_dual = tn_pp, tn_splitk, gelu_dual
        return self

    def __exit__(self, *a):
        for k, v in self.orig.items():
            setattr(self.ops, k, v)
        for k, v in self.lib_orig.items():
            setattr(self.lib, k, v)

    def kernels(self, steps):
        """Per kernel symbol (and per shape inside it): launches per step, average duration, TFLOP/s, fraction of the roofline that bounds it."""
        by = {}
        for e0, e1, fl, nb, sym, shape, fam in self.records:
            ms = e0.elapsed_time(e1)
            d = by.setdefault(sym, {'family': fam, 'n': 0, 'ms': 0.0, 'flop': 0.0, 'bytes': 0.0, 'shapes': {}})
            d['n'] += 1; d['ms'] += ms; d['flop'] += fl; d['bytes'] += nb
            s = d['shapes'].setdefault(shape, {'n': 0, 'ms': 0.0, 'flop': 0.0, 'bytes': 0.0})
            s['n'] += 1; s['ms'] += ms; s['flop'] += fl; s['bytes'] += nb

        def entry(name, d, fam):
            peak = PEAK_MXFP8 if fam == 'mxfp8' else PEAK_BF16
            tf, tbs = d['flop'] / (d['ms'] * 1e-3), d['bytes'] / (d['ms'] * 1e-3)
            hbm = tbs / HBM_ACHIEVABLE > tf / peak                    # the roofline it sits closer to
            e = {'name': name, 'launches': d['n'] // steps, 'avg_us': round(1e3 * d['ms'] / d['n'], 1), 'tflops': round(tf / 1e12, 1),
                 'frac': round(tf / peak, 4), 'bound': 'hbm' if hbm else 'mfma'}
            if hbm:
                e['algorithmic_TBps'] = round(tbs / 1e12, 2)
                e['frac_of_6.3TBps'] = round(tbs / HBM_ACHIEVABLE, 3)
            return e
        out = []
        for sym, d in sorted(by.items(), key=lambda kv: -kv[1]['ms']):
            e = entry(sym, d, d['family'])
            e['share_of_gemm_time'] = None
            if len(d['shapes']) > 1 and not sym.startswith('gemm_bf16_kernel'):
                e['shapes'] = [entry(sh, sd, d['family']) for sh, sd in sorted(d['shapes'].items(), key=lambda kv: -kv[1]['ms'])]
            out.append(e)
        tot = sum(d['ms'] for d in by.values())
        for e, (_, d) in zip(out, sorted(by.items(), key=lambda kv: -kv[1]['ms'])):
            e['share_of_gemm_time'] = round(d['ms'] / tot, 3)
        return out, by


def pmc_traffic(kernel=None):
    """HBM bytes per launch from the committed PMC passes of this same command (tools/profile_bench.sh -> profiles/*_bench_roofline.json; counters cannot
    be read from inside the benchmark process): the NAMED kernel's own FETCH_SIZE (x2, gfx950) + WRITE_SIZE when the file carries a per-kernel table
    (r04 on), else the GEMM-family average labelled as such.  (value, source) or (None, None)."""
    for f in (TRAFFIC_FILE, 'profiles/r05_bench_roofline.json', 'profiles/r04_bench_roofline.json', 'profiles/r03_bench_roofline.json', 'profiles/r02_bench_roofline.json', 'profiles/r01_bench_roofline.json'):
        try:
            with open(ROOT / f) as fh:
                d = json.load(fh)
            norm = lambda x: x.replace(' ', '')
            for sym, e in (d.get('per_kernel') or {}).items():
                if kernel and norm(sym).startswith(norm(kernel)):
                    return round(e['traffic_bytes_per_launch']), (f'{f}: counter bytes of {sym} itself, {e["launches"]} launches (static: rocprofv3 --pmc FETCH_SIZE x2 / '
                                                                  'WRITE_SIZE passes of this command, not measured in this run)')
            return round(d['traffic_bytes_per_launch']), (f'{f}: AVERAGE over all GEMM-family launches, not the named kernel alone (static: rocprofv3 --pmc FETCH_SIZE / '
                                                          'WRITE_SIZE passes of this command, not measured in this run)')
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def cpu_baseline(seconds_budget=25.0, max_threads=16):
    """The CPU oracle (restatement of the reference, proven equal to it in the build container) timed on the host
    cores of this box, fp32.  BOUNDED sample: the visual branch (97.6 % of the path's CPU time) is timed
    on `k` of the 14 segments of one clip and scaled by 14/k (segments are independent and identical in cost); the audio
    branch and the sync transformer are timed in full for the clip.  Threads are capped: beyond ~16 the 768-wide matmuls
    of this path slow down on a many-core host (measured), so `cores` reports the threads actually used.  A second, shorter sample
    times the visual branch at B = 4 clips (SURVEY §8d asks for B in {1, 4})."""
    from synchformer_amd import synth
    from oracle import synchformer_cpu as O
    sd = synth.make_state_dict(1337)
    vis = O.rgb_frontend(synth.make_video_u8(1, 14))
    aud = synth.make_spectrogram(1, 14)
    ncpu = os.cpu_count() or 1
    sweep = {}
    with torch.no_grad():
        for th in sorted({min(ncpu, c) for c in (max_threads, 32, 64)}):   # all 256 threads of the box: 78 s per segment (measured once in round 2, profiles/r02_batch_sweep.md)
            torch.set_num_threads(th)
            O.extract_vfeats(vis[:, :1], sd)                  # warm-up (thread pool, allocator)
            t0 = time.perf_counter()
            O.extract_vfeats(vis[:, :1], sd)
            sweep[th] = round(time.perf_counter() - t0, 3)
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    with torch.no_grad():
        t1 = sweep[threads]
        k = int(max(1, min(14, (seconds_budget - 2 * t1) // max(t1, 1e-3))))
        t0 = time.perf_counter()
        vf = O.extract_vfeats(vis[:, :k], sd, chunk=7)
        t_vis = (time.perf_counter() - t0) * 14.0 / k
        t0 = time.perf_counter()
        af = O.extract_afeats(aud, sd)
        v = O._lin(vf, sd, 'vproj')
        O.global_transformer(torch.cat([v] * (14 // k + 1), 1)[:, :14].reshape(1, -1, 768), O._lin(af, sd, 'aproj').reshape(1, -1, 768), sd)
        t_rest = time.perf_counter() - t0
        # B = 4: the visual branch of 4 clips on k4 segments each
        k4 = int(max(1, min(3, 8.0 // max(4 * t1, 1e-3))))
        vis4 = O.rgb_frontend(synth.make_video_u8(4, k4, seed=7))
        t0 = time.perf_counter()
        O.extract_vfeats(vis4, sd, chunk=4 * k4)
        t_vis4 = (time.perf_counter() - t0) * 14.0 / k4
    total = t_vis + t_rest
    total4 = t_vis4 + 4 * t_rest
    return {'value': round(1.0 / total, 5), 'unit': 'clips/s', 'cores': threads, 'kind': 'port',
            'sample': f'B = 1: visual branch on {k}/14 segments scaled x14/{k} ({t_vis:.1f} s/clip), audio branch + sync transformer in full '
                      f'({t_rest:.2f} s); fp32 torch CPU oracle; host has {os.cpu_count()} cpus',
            'b4': {'value': round(4.0 / total4, 5), 'unit': 'clips/s',
                   'sample': f'B = 4: visual branch of 4 clips on {k4}/14 segments each, scaled x14/{k4} ({t_vis4:.1f} s per 4 clips) + 4 x the B = 1 audio / sync time'},
            'thread_sweep_s_per_segment': sweep}


# ---- workloads ------------------------------------------------------------------------------------------------------------------------------
WORKLOAD_DOC = {
    'infer': ('clips/sec (14-seg offset pred)',
              'BASELINE configs[1]: batched offset inference, configs/sync.yaml model (237.5M params, random-init), uint8 224x224 frames + 16 kHz waveform '
              'segments resident in HBM; RGB + log-mel front-ends and the full forward to 21-way logits inside the timed step'),
    'train': ('clips/sec (Stage-2 train step, 14 segments)',
              'BASELINE configs[2]: Stage-2 sync-module train step (configs/sync.yaml, embd/resid/attn dropout 0.1): frozen extractors forward, '
              'backward of proj + sync transformer (22.6M params), flat 90 MB RCCL gradient all-reduce, fused clip+Adam'),
    'stage1': ('clips/sec (Stage-1 AVCLIP train step, 14 segments)',
               'BASELINE configs[3]: Stage-1 segment-level contrastive train step (configs/segment_avclip.yaml): forward with saved activations + backward of '
               'both towers (214.8M params), symmetric InfoNCE over B*14 segments, 7 gradient buckets all-reduced under the backward, fused clip+AdamW'),
    'ft': ('clips/sec (synchronizability fine-tune step, 13 segments, MXFP8 extractor GEMMs)',
           'BASELINE configs[4]: synchronizability fine-tune step (configs/ft_synchability.yaml): frozen extractors with their qkv / proj / fc1 / '
           'fc2 Linears on MXFP8 (OCP e4m3 + E8M0 per 32 k), 184-token sync transformer with the 2-way sync head trained in bf16 (22.6M params), '
           'RCCL gradient all-reduce, fused clip+Adam'),
}


def build_workload(name, args, dev, rank, world, local):
    """-> dict(step_fn, vis, aud, B, S, serial(on), trainer)"""
    from synchformer_amd import synth
    from synchformer_amd.dist import scaled_lr                            # base learning rate x number of GPUs (scripts/train_utils.py:218)
    from synchformer_amd.engine import SynchformerEngine
    B = args.batch if (args.batch is not None and name == args.workload) else (2 if name == 'stage1' else 16)
    trainer, eng, mel = None, None, None
    if name == 'train' and args.dropin:
        import synchformer_amd as sa
        model = sa.instantiate_from_config(sa.sync_yaml_model_config())
        model.load_state_dict(synth.make_state_dict(1337), strict=True)
        model.seg_chunk = args.seg_chunk
        model = model.to(dev)
        for p_ in list(model.vfeat_extractor.parameters()) + list(model.afeat_extractor.parameters()):
            p_.requires_grad = False
        model.train(); model.vfeat_extractor.eval(); model.afeat_extractor.eval()
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local]) if world > 1 else model
        opt = torch.optim.Adam([p_ for p_ in model.parameters() if p_.requires_grad], lr=scaled_lr(2e-6, world), betas=(0.9, 0.999), eps=1e-7)
        scaler = torch.amp.GradScaler('cuda')
        targets = synth.make_targets(B, 21, seed=1337 + rank).to(dev)

        def step_fn(v, a):
            opt.zero_grad(set_to_none=True)
            with torch.autocast('cuda'):
                loss, _ = ddp(v, a, targets)
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            scaler.step(opt)
            scaler.update()
            return loss.detach().reshape(1)
        eng = model._engine(need_sync=False)
    elif name == 'ft':
        # configs/ft_synchability.yaml - frozen extractors, GlobalTransformerWithSyncabilityHead over 13 segments (184 tokens), 2-way head, batch 16 per
        # GPU; the extractors' qkv / proj / fc1 / fc2 run on MXFP8 operands (v_mfma_scale_f32_32x32x64_f8f6f4)
        from synchformer_amd.train import SyncTrainer
        trainer = SyncTrainer(synth.make_state_dict(1337, n_pos=184, n_out=2, head='sync_head'), dev, lr=scaled_lr(2e-6, world), seg_chunk=args.seg_chunk,
                              embd_pdrop=0.1, resid_pdrop=0.1, attn_pdrop=0.1, seed=1337 + rank, fp8_towers=True)
        eng = trainer.engine
        targets = synth.make_targets(B, 2, seed=1337 + rank).to(dev)
        step_fn = lambda v, a: trainer.train_step(v, a, targets)
    elif name == 'train':
        from synchformer_amd.train import SyncTrainer
        trainer = SyncTrainer(synth.make_state_dict(1337), dev, lr=scaled_lr(2e-6, world), seg_chunk=args.seg_chunk, embd_pdrop=0.1, resid_pdrop=0.1,
                              attn_pdrop=0.1, seed=1337 + rank)
        eng = trainer.engine
        targets = synth.make_targets(B, 21, seed=1337 + rank).to(dev)
        step_fn = lambda v, a: trainer.train_step(v, a, targets)
    elif name == 'stage1':
        # Stage-1 AVCLIP train step (configs/segment_avclip.yaml: base_batch_size 2 clips x 14 segments per GPU, both towers trainable)
        from synchformer_amd.stage1 import AVCLIPTrainer
        sd = {k: v for k, v in synth.make_state_dict(1337).items() if k.startswith(('vfeat_extractor.', 'afeat_extractor.'))}
        trainer = AVCLIPTrainer(sd, dev, lr=1e-4, drop_path_rate=0.2, seed=1337 + rank)     # train mode: DropPath 0.2 like the reference's towers; Stage 1 does NOT scale its learning rate by the world size (train_clip.py:276,314 pass cfg.training.learning_rate as is; only the warm-up is divided)
        step_fn = lambda v, a: trainer.train_step(v, a).reshape(1)
    else:
        # headline: the step starts from what the north star names - 224 x 224 uint8 frames and 16 kHz WAVEFORM segments (B, 14, 10240), both resident in HBM; the
        # RGB front-end runs inside the patch gather and the log-mel front-end (sf_mel_frontend) inside the timed region, in front of the forward
        from synchformer_amd.frontend import MelFrontend
        eng = SynchformerEngine(synth.make_state_dict(1337), dev, seg_chunk=args.seg_chunk)
        mel = MelFrontend(dev)
        step_fn = lambda v, wv: eng.forward(v, mel(wv))
    S = 13 if name == 'ft' else 14                                         # ft_synchability.yaml: 13 segments (184-token sync transformer)
    vis = synth.make_video_u8(B, S, seed=1337 + rank).to(dev)             # (B,S,16,3,224,224) uint8, HBM-resident
    aud = synth.make_spectrogram(B, S, seed=1337 + rank).to(dev)          # (B,S,1,128,66) fp32
    if name == 'infer':
        aud = synth.make_wave(B, S, seed=1337 + rank).to(dev)             # (B,S,10240) fp32 waveform segments, 0.64 s @ 16 kHz
    serial = (lambda on: setattr(trainer, 'two_streams', on)) if name == 'stage1' else (lambda on: setattr(eng, 'audio_side_stream', on))
    return {'step_fn': step_fn, 'vis': vis, 'aud': aud, 'B': B, 'S': S, 'serial': serial, 'trainer': None if args.dropin else trainer, 'eng': eng,
            'mel': mel if name == 'infer' else None}


def power_probe(step_fn, vis, aud, seconds=2.0):
    """Socket power and shader clock while the forward runs (rank 0, single GPU, AFTER the timed region): `rocm-smi -P -c` sampled from a thread next to ~2 s of the
    same steps.  The step is power-limited on MI355X (profiles/r03_power.md): the clock this reports is what `roofline.frac` has to be read against.  None if
    rocm-smi is not there or prints something else."""
    import re, subprocess, threading
    samples, stop = [], threading.Event()
    gi = '0'                                                         # rocm-smi's index of the device this process calls cuda:0
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        first = os.environ.get(var, '').split(',')[0].strip()
        if first.isdigit():
            gi = first
            break

    def poll():
        while not stop.is_set():
            try:
                txt = subprocess.run(['rocm-smi', '-P', '-c'], capture_output=True, text=True, timeout=5).stdout
                pw = re.search(r'GPU\[' + gi + r'\].*Package Power \(W\): ([0-9.]+)', txt)
                ck = re.search(r'GPU\[' + gi + r'\].*sclk clock level: \S+ \((\d+)Mhz\)', txt)
                if pw and ck:
                    samples.append((time.perf_counter(), float(pw.group(1)), int(ck.group(1))))
            except Exception:                                          # noqa: BLE001 - an instrument, never an error of the benchmark
                return
            time.sleep(0.2)
    try:
        cap = None
        txt = subprocess.run(['rocm-smi', '--showmaxpower'], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r'GPU\[' + gi + r'\].*Max Graphics Package Power \(W\): ([0-9.]+)', txt)
        cap = float(m.group(1)) if m else None
        th = threading.Thread(target=poll, daemon=True)
        t0 = time.perf_counter()
        th.start()
        n = 0
        while time.perf_counter() - t0 < seconds:
            step_fn(vis, aud)
            torch.cuda.synchronize()
            n += 1
        t1 = time.perf_counter()
        stop.set()
        th.join(timeout=6)
        mid = [x for x in samples if t0 + 0.6 < x[0] < t1] or samples
        if not mid:
            return None
        med = lambda v: sorted(v)[len(v) // 2]
        w_med, cps = med([x[1] for x in mid]), n * vis.shape[0] / (t1 - t0)
        return {'socket_w': w_med, 'sclk_mhz': med([x[2] for x in mid]), 'cap_w': cap, 'samples': len(mid), 'steps': n,
                'clips_per_s': round(cps, 1), 'rocm_smi_gpu': int(gi),
                # the step is power-limited (time = joules / cap): J/clip is the figure a kernel change has to move (track it per item)
                'joules_per_clip': round(w_med / cps, 3),
                # (rocm-smi answering slowly puts every sample into the gaps between steps: an idle reading says nothing about the step)
                'suspect': bool(len(mid) < 3 or (cap is not None and w_med < 0.6 * cap)),
                'what': 'rocm-smi next to a further ~2 s of the same forward, after the timed region; dense MFMA peaks are quoted at 2400 MHz'}
    except Exception:                                                  # noqa: BLE001
        stop.set()
        return None


def roofline_of(kernels, by, name):
    """The `roofline` object for workload `name` from the per-kernel entries: the single dominant kernel (by time) + the family aggregate."""
    dom = kernels[0]
    fam = 'mxfp8' if name == 'ft' and any(d['family'] == 'mxfp8' for d in by.values()) else 'bf16'
    sel = [d for d in by.values() if d['family'] == fam]
    ms, fl, nb, n = sum(d['ms'] for d in sel), sum(d['flop'] for d in sel), sum(d['bytes'] for d in sel), sum(d['n'] for d in sel)
    peak = PEAK_MXFP8 if fam == 'mxfp8' else PEAK_BF16
    return dom, {'tflops': round(fl / (ms * 1e-3) / 1e12, 1), 'frac': round(fl / (ms * 1e-3) / peak, 4), 'launch_count': n, 'ms': ms,
                 'flop': fl, 'bytes': nb, 'peak': peak / 1e12}


def host_leg(eng, dev, B, steps, barrier, max_over_ranks, headline_clips_per_s, world):
    """`workloads.infer_from_host`: the same 16-clip forward fed from PINNED HOST memory - raw uint8 frames (B, 125, 3, 224, 224) and
    16 kHz waveforms (B, 80000) fp32, i.e. what the decoder hands over BEFORE GenerateMultipleSegments / RGB normalisation / mel (dataset/transforms.py:402-499,
    647-669, 815-889; the reference moves the already transformed batch in prepare_inputs, scripts/train_utils.py:359-369).  H2D of batch i+1 runs on a copy
    stream under the forward of batch i (synchformer_amd.frontend.HostClipPipeline); segmenting, RGB front-end and the mel front-end run on the device INSIDE
    the timed region.  Reported beside the HBM-resident headline, never as `value`."""
    from synchformer_amd.frontend import HostClipPipeline, MelFrontend
    T, n_samp = 125, 80000
    g = torch.Generator().manual_seed(1337)
    host = []
    for _ in range(2):                                                     # two distinct pinned batches, alternated
        f = torch.randint(0, 256, (B, T, 3, 224, 224), generator=g, dtype=torch.uint8).pin_memory()
        wv = (torch.rand(B, n_samp, generator=g) * 2.0 - 1.0).pin_memory()
        host.append((f, wv))
    mel = MelFrontend(dev)
    pipe = HostClipPipeline(eng, mel, B, T, n_samp)
    nbytes = host[0][0].numel() + host[0][1].numel() * 4
    # (a) the transfer alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(4):
        pipe.frames[0].copy_(host[i & 1][0], non_blocking=True)
        pipe.wave[0].copy_(host[i & 1][1], non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    h2d_ms = e0.elapsed_time(e1) / 4
    # (b) the pipelined step: warm-up, then exactly `steps` steps between barriers
    pipe.stage(*host[0])
    for i in range(2):
        logits = pipe.step(*host[(i + 1) & 1])
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        logits = pipe.step(*host[i & 1])
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    pipe.step()                                                            # drain the staged batch
    torch.cuda.synchronize()
    assert torch.isfinite(logits).all()
    # (c) no overlap: transfer, then forward, on one stream
    barrier()
    t1 = time.perf_counter()
    for i in range(steps):
        pipe.frames[0].copy_(host[i & 1][0], non_blocking=True)
        pipe.wave[0].copy_(host[i & 1][1], non_blocking=True)
        logits = eng.forward_clips(pipe.frames[0], pipe.wave[0], mel)
    barrier()
    dt_serial = max_over_ranks(time.perf_counter() - t1)
    cps = B * world * steps / dt
    return {'metric': 'clips/sec (14-seg offset pred), inputs in pinned host memory', 'clips_per_gpu': B, 'steps': steps,
            'clips_per_s': round(cps, 3), 'ms_per_step': round(1e3 * dt / steps, 3), 'ratio_to_hbm_resident_headline': round(cps / headline_clips_per_s, 4),
            'h2d_bytes_per_clip': nbytes // B, 'h2d_alone_ms_per_batch': round(h2d_ms, 3), 'h2d_GBps': round(nbytes / h2d_ms / 1e6, 1),
            'unpipelined_clips_per_s': round(B * world * steps / dt_serial, 3),
            'what': 'uint8 frames (B,125,3,224,224) + fp32 waveforms (B,80000) in pinned host buffers; H2D on a copy stream double-buffered against compute; '
                    'segmenting + RGB normalisation + log-mel on the device inside the timed region (HostClipPipeline -> engine.forward_clips)'}


def single_clip_and_dispatcher_legs(w, B, steps):
    """Two reported side numbers of the headline engine (rank 0, one GPU, after the timed region):
    * `latency_b1_ms` - BASELINE configs[0] is a ONE-clip workload (example.py:174-176): wall time of one 14-segment forward incl. the mel front-end, eager (ctypes
      launches) and replayed as one captured HIP graph (engine.capture), median of 15;
    * `dispatcher_route` - the same B-clip and 1-clip forwards with every launch of the default schedule going through the PyTorch dispatcher
      (`torch.ops.synchformer.*`, ops.via_dispatcher) instead of straight into the C ABI: what "registered as PyTorch-ROCm custom ops" costs per step."""
    from synchformer_amd import ops
    eng, mel = w['eng'], w['mel']
    v1, a1 = w['vis'][:1].contiguous(), w['aud'][:1].contiguous()

    def med_ms(fn, n=15):
        fn(); fn()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return round(1e3 * sorted(ts)[len(ts) // 2], 3)

    def batch_ms(fn, n):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return round(1e3 * (time.perf_counter() - t0) / n, 3)
    eager_b1 = med_ms(lambda: eng.forward(v1, mel(a1)))
    direct_bB = batch_ms(lambda: eng.forward(w['vis'], mel(w['aud'])), steps)
    with ops.via_dispatcher() as vd:
        disp_b1 = med_ms(lambda: eng.forward(v1, mel(a1)))
        disp_bB = batch_ms(lambda: eng.forward(w['vis'], mel(w['aud'])), steps)
        calls = vd.calls
    cap = eng.capture(v1, mel(a1))
    graph_b1 = med_ms(lambda: cap(v1, mel(a1)))
    # two clips = 28 segments: inside the window where a CAPTURED forward runs the visual tower as two halves on two HIP streams (engine._parts, round 6); beside it the
    # same graph with the single-stream schedule and the eager forward (single stream: issued eagerly, twice the launches are a host cost that depends on the box)
    b2 = None
    if w['vis'].shape[0] >= 2:
        v2, a2 = w['vis'][:2].contiguous(), w['aud'][:2].contiguous()
        eager_b2 = med_ms(lambda: eng.forward(v2, mel(a2)))
        cap2 = eng.capture(v2, mel(a2))
        graph_split = med_ms(lambda: cap2(v2, mel(a2)))
        keep, eng.vis_split_mode = eng.vis_split_mode, 'never'
        try:
            cap2s = eng.capture(v2, mel(a2))
            graph_single = med_ms(lambda: cap2s(v2, mel(a2)))
        finally:
            eng.vis_split_mode = keep
        b2 = {'eager': eager_b2, 'hip_graph': graph_split, 'hip_graph_single_stream': graph_single, 'clips': 2, 'segments': 2 * w['S']}
    return {'latency_b2_ms': b2,
            'latency_b1_ms': {'eager': eager_b1, 'hip_graph': graph_b1, 'clips': 1, 'segments': w['S'],
                              'what': 'one 14-segment clip, uint8 frames + waveform resident in HBM -> logits (mel front-end included), median of 15 synchronised forwards'},
            'dispatcher_route': {'ms_per_step_direct': direct_bB, 'ms_per_step_dispatcher': disp_bB, 'overhead_frac': round(disp_bB / direct_bB - 1.0, 4),
                                 'latency_b1_ms_dispatcher': disp_b1, 'latency_b1_ms_direct': eager_b1, 'clips_per_gpu': B, 'dispatcher_calls_counted': calls,
                                 'what': 'every launch of the default schedule through torch.ops.synchformer.* (ops.via_dispatcher) against the direct ctypes path, same engine'}}


def run_steps(w, steps, barrier, world, dist):
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = w['step_fn'](w['vis'], w['aud'])
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    return dt


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world == 1 and args.gpus > 1 and 'RANK' not in os.environ:
        respawn_distributed(args)
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    rank = int(os.environ.get('RANK', 0))
    local = 0 if args.single_device else int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    # one launcher process per GPU: keep its threads on the CPUs of the NUMA node that GPU hangs off (8 Python launchers on a 256-cpu host otherwise migrate across
    # sockets and every hipLaunchKernel crosses the fabric).  Advisory: silently skipped where sysfs does not say or affinity cannot be set.
    numa_node = -1
    if world > 1 and not args.single_device:
        from synchformer_amd.dist import numa_cpus_of_gpu
        nc = numa_cpus_of_gpu(local)
        if nc:
            try:
                os.sched_setaffinity(0, nc[1])
                numa_node = nc[0]
            except OSError:
                pass
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()

    def gather_ranks(x):
        if world == 1:
            return [x]
        mine = torch.tensor([x], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        return [float(t.item()) for t in allr]

    name = args.workload
    w = build_workload(name, args, dev, rank, world, local)
    B, S = w['B'], w['S']
    step_fn = w['step_fn']
    if args.graph:
        assert name == 'infer', '--graph serves the inference workload'
        cap = w['eng'].capture(w['vis'], w['mel'](w['aud']))              # (the mel launches stay in front of the replayed graph)
        w['step_fn'] = step_fn = lambda v, wv: cap(v, w['mel'](wv))
    for _ in range(args.warmup):
        logits = step_fn(w['vis'], w['aud'])
    timed_trainer = w['trainer'] if name in ('train', 'stage1', 'ft') else None
    if timed_trainer is not None and world > 1:
        timed_trainer.time_comm = True
    # ---- the timed region: EXACTLY K steps of the product configuration, no instrumentation -------------------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = step_fn(w['vis'], w['aud'])
    barrier()
    dt_local = time.perf_counter() - t0
    assert torch.isfinite(logits).all()
    comm_ms = []
    if timed_trainer is not None and world > 1:
        timed_trainer.time_comm = False
        comm_ms = [round(x, 3) for x in gather_ranks(timed_trainer.exposed_comm_ms())]     # last step's stall on the gradient buckets
    per_rank = [round(B * args.steps / x, 3) for x in gather_ranks(dt_local)]
    numa_nodes = [int(x) for x in gather_ranks(float(numa_node))] if world > 1 else []
    dt = max_over_ranks(dt_local)

    # ---- roofline pass: the same K steps again with every GEMM-family launch bracketed by HIP events on its launch stream.  The
    # product runs the audio tower on a second stream next to the visual one; an event pair on that stream would also time the queueing
    # behind the other tower's kernels, so for THIS pass the two towers are serialised on one stream (same kernels, same shapes, same
    # launch count; `value` above is not affected).  rocprofv3 --kernel-trace of this command sees both passes (profiles/).
    def kernel_pass(wl, steps):
        wl['serial'](False)
        with GemmTimer() as gt:
            gt.enabled = rank == 0
            barrier()
            t1 = time.perf_counter()
            for _ in range(steps):
                wl['step_fn'](wl['vis'], wl['aud'])
            barrier()
            dt_r = time.perf_counter() - t1
            res = gt.kernels(steps) if gt.enabled and gt.records else (None, None)
        wl['serial'](True)
        return res[0], res[1], dt_r

    kernels = by = None
    dt_roof = 0.0
    roofline_error = None
    if not args.no_kernel_timing and not args.graph:
        try:
            kernels, by, dt_roof = kernel_pass(w, args.steps)
        except Exception as ex:                                    # noqa: BLE001 - instrumentation must not cost the headline number
            if world > 1:
                raise                                              # (ranks would diverge at the next barrier)
            roofline_error = f'{type(ex).__name__}: {ex}'[:300]

    clips = B * world * args.steps
    value = clips / dt
    out = None
    if rank == 0:
        train_x = 3 if name == 'stage1' else 1
        out = {
            'metric': WORKLOAD_DOC[name][0],
            'value': round(value, 3), 'unit': 'clips/s', 'n_gpus': world, 'rccl_ranks': world if (world > 1 and args.dist_backend == 'nccl') else 0,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'mxfp8 (extractor Linears) + bf16' if name == 'ft' else 'bf16',
            'data': 'synthetic',
            'config': {'workload': WORKLOAD_DOC[name][1], 'clips_per_gpu': B, 'segments': S, 'seg_chunk': args.seg_chunk, 'hip_graph': bool(args.graph),
                       'parallelism': f'replicas x{world}' if name == 'infer' else f'dp{world}'},
            'clips_per_s_by_rank': per_rank,
            # stage1: forward + dgrad + wgrad of every linear ~ 3x the forward FLOPs (attention backward ~2.5x; approximate)
            'path_flop_per_clip': FLOP_PER_CLIP * train_x * S / 14,
            'path_mfma_frac': round(value * FLOP_PER_CLIP * train_x * S / 14 / (world * PEAK_BF16), 4),
            # the same with the FLOPs the engine executes (aggregators computed for their one consumed output row only)
            'path_mfma_frac_executed': round(value * FLOP_PER_CLIP_EXECUTED * train_x * S / 14 / (world * PEAK_BF16), 4),
        }
        if args.dropin:
            out['config']['dropin'] = 'nn.Module + autocast + GradScaler + clip_grad_norm_ + torch.optim.Adam' + (' + DistributedDataParallel' if world > 1 else '')
        if world > 1:
            out['launcher_numa_node_by_rank'] = numa_nodes
        if comm_ms:
            out['comm'] = {'exposed_ms_last_step_by_rank': comm_ms,
                           'what': 'time the compute stream waited for the gradient all-reduce (Stage-2: one flat 90 MB bucket after the backward; '
                                   'Stage-1: 7 buckets launched under the backward, the wait is for what did not overlap)'}
        if roofline_error:
            out['roofline_error'] = roofline_error
        if name == 'infer' and world == 1 and not args.no_kernel_timing and not args.graph:
            pw = power_probe(step_fn, w['vis'], w['aud'])
            if pw:
                out['power'] = pw
        if kernels:
            dom, agg = roofline_of(kernels, by, name)
            traffic, tsrc = pmc_traffic(dom['name']) if name == 'infer' else (None, None)
            out['roofline'] = {
                'bound': dom['bound'], 'kernel': dom['name'], 'achieved': dom['tflops'] if dom['bound'] == 'mfma' else dom['algorithmic_TBps'],
                'peak': (agg['peak'] if dom['bound'] == 'mfma' else 8.0), 'unit': 'TFLOP/s' if dom['bound'] == 'mfma' else 'TB/s',
                'frac': dom['frac'] if dom['bound'] == 'mfma' else round(dom['algorithmic_TBps'] / 8.0, 4),
                # the PATH-level figure - clips/s x 5.725 TFLOP per clip / (N x 2.5 PFLOP/s bf16) - beside the dominant kernel's own fraction above
                'path_frac': out['path_mfma_frac'], 'path_peak_TFLOPs': PEAK_BF16 * world / 1e12,
                'avg_launch_us': dom['avg_us'], 'launches': dom['launches'],
                'traffic': traffic, 'traffic_source': tsrc,
                'kernels': kernels,
                'gemm_family': {'what': 'all GEMM-family launches of a step together (sf_gemm_bf16 + sf_gemm_res_ln768 + sf_qkv_time_attention'
                                        + (' ; MXFP8 launches only' if name == 'ft' else '') + '), GEMM FLOPs only - the fused LayerNorm / attention / GELU work is not counted',
                                'tflops': agg['tflops'], 'frac': agg['frac'], 'launches': agg['launch_count'] // args.steps,
                                'avg_launch_ms': round(agg['ms'] / agg['launch_count'], 4), 'flop_per_launch': agg['flop'] / agg['launch_count'],
                                'algorithmic_bytes_per_launch': round(agg['bytes'] / agg['launch_count']),
                                'share_of_step_time': round(agg['ms'] * 1e-3 / dt_roof, 3)},
                'measured': f'HIP events on the launch stream over a second pass of the same {args.steps} steps with the two towers '
                            f'serialised on one stream ({1e3 * dt_roof / args.steps:.1f} ms/step); value is the un-instrumented two-stream pass'}

    # ---- the other BASELINE workloads (configs[2], [3], [4]): a few steps each, after the headline region, into the same line ----------------
    def emit():
        if rank == 0 and out is not None:
            print(json.dumps(out), flush=True)

    if name == 'infer' and not args.no_workloads and not args.graph:
        done = threading.Event()

        def watchdog():                                            # a hung collective in an extra workload must not cost the headline number
            if not done.wait(240.0):
                if rank == 0:
                    out['workloads_error'] = 'watchdog: the extra workloads did not finish within 240 s'
                    emit()
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        wl_out = {}
        try:
            hl = host_leg(w['eng'], dev, B, max(args.workload_steps, 3), barrier, max_over_ranks, value, world)
            if rank == 0:
                wl_out['infer_from_host'] = hl
        except Exception as ex:                                    # noqa: BLE001 - reported in the line, the headline number stands
            if world > 1:
                raise
            wl_out['infer_from_host'] = {'error': f'{type(ex).__name__}: {ex}'[:300]}
        if world == 1:
            try:
                extra = single_clip_and_dispatcher_legs(w, B, max(args.workload_steps, 3))
                out['latency_b1_ms'] = extra['latency_b1_ms']
                out['latency_b2_ms'] = extra['latency_b2_ms']
                wl_out['dispatcher_route'] = extra['dispatcher_route']
            except Exception as ex:                                # noqa: BLE001
                out['latency_b1_ms'] = {'error': f'{type(ex).__name__}: {ex}'[:300]}
        del w, step_fn, logits
        torch.cuda.empty_cache()
        for wn in ('train', 'stage1', 'ft'):
            try:
                ww = build_workload(wn, args, dev, rank, world, local)
                for _ in range(2):
                    ww['step_fn'](ww['vis'], ww['aud'])
                if ww['trainer'] is not None and world > 1:
                    ww['trainer'].time_comm = True
                dtw = max_over_ranks(run_steps(ww, args.workload_steps, barrier, world, dist))
                comm = None
                if ww['trainer'] is not None and world > 1:
                    ww['trainer'].time_comm = False
                    comm = [round(x, 3) for x in gather_ranks(ww['trainer'].exposed_comm_ms())]
                kk, bb, _ = kernel_pass(ww, 2) if not args.no_kernel_timing else (None, None, 0.0)
                if rank == 0:
                    e = {'metric': WORKLOAD_DOC[wn][0], 'config': WORKLOAD_DOC[wn][1], 'clips_per_gpu': ww['B'], 'segments': ww['S'], 'steps': args.workload_steps,
                         'ms_per_step': round(1e3 * dtw / args.workload_steps, 3), 'clips_per_s': round(ww['B'] * world * args.workload_steps / dtw, 3)}
                    if comm is not None:
                        e['comm_exposed_ms_last_step_by_rank'] = comm
                    if kk:
                        dom, agg = roofline_of(kk, bb, wn)
                        e['roofline'] = {'kernel': dom['name'], 'bound': dom['bound'], 'frac': dom['frac'], 'tflops': dom['tflops'], 'avg_us': dom['avg_us'],
                                         'gemm_family_tflops': agg['tflops'], 'gemm_family_frac': agg['frac'],
                                         'kernels': [{k_: v_ for k_, v_ in x.items() if k_ != 'shapes'} for x in kk[:6]]}
                    wl_out[wn] = e
                del ww
                torch.cuda.empty_cache()
            except Exception as ex:                                # noqa: BLE001 - reported in the line, the headline number stands
                wl_out[wn] = {'error': f'{type(ex).__name__}: {ex}'[:300]}
        if rank == 0:
            out['workloads'] = wl_out
        done.set()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    emit()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
