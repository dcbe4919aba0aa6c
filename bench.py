#!/usr/bin/env python
"""Headline benchmark: clips/sec of Synchformer offset prediction (14 x 0.64 s segments, 21 offset classes).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: batched offset inference, configs/sync.yaml model, random-init weights,
synthetic 224x224 @ 25 fps uint8 frames + 128x66 log-mel spectrograms already resident in HBM.  One "step" = one
full Synchformer.forward() (RGB front-end -> Motionformer -> AST -> sync transformer -> logits) over B clips per GPU.
Multi-GPU: inference shards by clip with no data-path collective ("replicas only"): every rank runs
its own B clips; value = all clips / max-over-ranks time (weak scaling).
Prints ONE JSON line on rank 0 with `roofline` (dominant kernel = the bf16 GEMM, timed live with HIP events on the
launch stream in a second pass of the same K steps - see main()) and `cpu_baseline` (the CPU oracle timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402

FLOP_PER_CLIP = 5.725e12          # algorithmic forward FLOPs per 14-segment clip
# FLOPs the engine actually executes per clip: both aggregator layers are computed for output row 0 only (motionformer.py:332 reads nothing
# else), which drops 14 x 8 x (out_proj 0.232 + MLP 1.859 + the 196 unused query rows of the attention 0.118) G = 0.247 T of the visual
# aggregator and 14 x 6 x 0.17 G of the audio one
FLOP_PER_CLIP_EXECUTED = 5.725e12 - 0.247e12 - 0.014e12
PEAK_BF16 = 2.5e15                # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16, help='clips per GPU per step (configs/sync.yaml batch = 16)')
    ap.add_argument('--seg-chunk', type=int, default=224)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL over xGMI; default) | gloo (plumbing tests)')
    ap.add_argument('--single-device', action='store_true', help='TEST ONLY: put every rank on cuda:0 (needs --dist-backend gloo)')
    ap.add_argument('--workload', choices=['infer', 'train', 'stage1', 'ft'], default='infer',
                    help="infer = BASELINE configs[1] (headline metric); train = configs[2]: Stage-2 step, frozen extractors, "
                         "backward + RCCL gradient all-reduce + fused clip/Adam (dropout 0.1 as in sync.yaml); ft = configs[4]: the "
                         "synchronizability fine-tune step (13 segments, 2-way sync head) with the frozen extractors' Linears on MXFP8")
    ap.add_argument('--graph', action='store_true', help='replay the forward as one captured HIP graph (infer workload only)')
    ap.add_argument('--dropin', action='store_true', help="train workload through the drop-in nn.Module + torch.optim.Adam + GradScaler "
                    "(the reference's loop body, train_utils.py:373-386) instead of SyncTrainer's fused step: the wrapper overhead as a number")
    return ap.parse_args()


class GemmTimer:
    """Brackets every sf_gemm_bf16 launch with HIP events on the launch stream (= torch's current stream)."""

    def __init__(self):
        from synchformer_amd import ops
        self.ops = ops
        self.orig = ops.gemm
        self.records = []
        self.enabled = False

    def __enter__(self):
        def timed(a, w, bias, out, *, M=None, **kw):
            if not self.enabled:
                return self.orig(a, w, bias, out, M=M, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig(a, w, bias, out, M=M, **kw)
            e1.record()
            m = a.shape[0] if M is None else M
            n, k = w.shape
            res = kw.get('residual')
            nbytes = m * k * 2 + n * k * 2 + m * n * out.element_size() + (m * n * 4 if res is not None else 0)   # A + W + C (+ R), each once
            self.records.append((e0, e1, 2.0 * m * n * k, nbytes))
            return r
        def timed_ln(a, w, bias, x, gamma, beta, y, eps, *, M=None, residual=None):
            # the full-row GEMM + residual + LayerNorm launches are GEMM launches of the same family (sf_gemm_res_ln768)
            if not self.enabled:
                return self.orig_ln(a, w, bias, x, gamma, beta, y, eps, M=M, residual=residual)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig_ln(a, w, bias, x, gamma, beta, y, eps, M=M, residual=residual)
            e1.record()
            m = a.shape[0] if M is None else M
            n, k = 768, a.shape[1]                                                     # w is (768, K) or k-step-major (K/32, 768, 32)
            nbytes = m * k * 2 + n * k * 2 + m * n * (4 + 4 + 2)                       # A + W + R read, X + Y written, each once
            self.records.append((e0, e1, 2.0 * m * n * k, nbytes))
            return r
        def timed_mx(a_q, a_s, w_q, w_s, bias, out, *, M=None, residual=None, gelu=False, out_scales=None):
            # the MXFP8 launches are recorded apart: their roofline is the MX-fp8 matrix peak, not the bf16 one
            if not self.enabled:
                return self.orig_mx(a_q, a_s, w_q, w_s, bias, out, M=M, residual=residual, gelu=gelu, out_scales=out_scales)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig_mx(a_q, a_s, w_q, w_s, bias, out, M=M, residual=residual, gelu=gelu, out_scales=out_scales)
            e1.record()
            m = a_q.shape[0] if M is None else M
            n, k = w_q.shape
            nbytes = m * k + n * k + (m + n) * k // 32 + m * n * out.element_size() + (m * n * 4 if residual is not None else 0)
            self.mx_records.append((e0, e1, 2.0 * m * n * k, nbytes))
            return r
        def timed_qt(x, w, bias, qkv_cls, out, partials, *, n_seq, n_groups, scale):
            # the fused temporal qkv + time-attention launch is a GEMM launch too (sf_qkv_time_attention): 2304 x 768 over the patch rows
            if not self.enabled:
                return self.orig_qt(x, w, bias, qkv_cls, out, partials, n_seq=n_seq, n_groups=n_groups, scale=scale)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig_qt(x, w, bias, qkv_cls, out, partials, n_seq=n_seq, n_groups=n_groups, scale=scale)
            e1.record()
            m, n, k = n_seq * 8 * n_groups, 2304, 768
            nbytes = m * k * 2 + n * k * 2 + m * 768 * 2                               # A + W read, the 768-wide attention output written
            self.records.append((e0, e1, 2.0 * m * n * k, nbytes))
            return r
        self.mx_records = []
        self.orig_qt = self.ops.qkv_time_attention
        self.ops.qkv_time_attention = timed_qt
        self.ops.gemm = timed
        self.orig_ln = self.ops.gemm_res_ln
        self.ops.gemm_res_ln = timed_ln
        self.orig_mx = self.ops.gemm_mxfp8
        self.ops.gemm_mxfp8 = timed_mx
        return self

    def __exit__(self, *a):
        self.ops.gemm = self.orig
        self.ops.gemm_res_ln = self.orig_ln
        self.ops.gemm_mxfp8 = self.orig_mx
        self.ops.qkv_time_attention = self.orig_qt

    def mx_summary(self):
        ms = sum(r[0].elapsed_time(r[1]) for r in self.mx_records)
        return len(self.mx_records), ms, sum(r[2] for r in self.mx_records), sum(r[3] for r in self.mx_records)

    def summary(self):
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        fl = sum(r[2] for r in self.records)
        self.bytes = sum(r[3] for r in self.records)
        return len(self.records), ms, fl


def pmc_traffic():
    """HBM bytes per sf_gemm_bf16 launch from the committed PMC passes of this same command (tools/profile_bench.sh ->
    profiles/r02_bench_roofline.json; counters cannot be read from inside the benchmark process).  None if absent."""
    for tag in ('r02', 'r01'):
        f = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', f'{tag}_bench_roofline.json')
        try:
            with open(f) as fh:
                return round(json.load(fh)['traffic_bytes_per_launch'])
        except (OSError, KeyError, ValueError):
            continue
    return None


def cpu_baseline(seconds_budget=25.0, max_threads=16):
    """The CPU oracle (restatement of the reference, proven equal to it in the build container) timed on the host
    cores of this box, fp32.  BOUNDED sample: the visual branch (97.6 % of the path's CPU time) is timed
    on `k` of the 14 segments of one clip and scaled by 14/k (segments are independent and identical in cost); the audio
    branch and the sync transformer are timed in full for the clip.  Threads are capped: beyond ~16 the 768-wide matmuls
    of this path slow down on a many-core host (measured), so `cores` reports the threads actually used."""
    from synchformer_amd import synth
    from oracle import synchformer_cpu as O
    sd = synth.make_state_dict(1337)
    vis = O.rgb_frontend(synth.make_video_u8(1, 14))
    aud = synth.make_spectrogram(1, 14)
    # thread count: measured on this host, one segment per candidate (the 768-wide matmuls of this path stop scaling long before a
    # 256-thread host is full); the sweep is kept in the output so that `cores` is evidence, not an assertion
    ncpu = os.cpu_count() or 1
    sweep = {}
    with torch.no_grad():
        for th in sorted({min(ncpu, c) for c in (max_threads, 32, 64)}):   # all 256 threads of the box: 78 s per segment (oversubscribed; measured once in round 2, profiles/r02_batch_sweep.md) - not re-timed on every run
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
    total = t_vis + t_rest
    return {'value': 1.0 / total, 'unit': 'clips/s', 'cores': threads, 'kind': 'port',
            'sample': f'1 clip (clips are independent: the B = 4 rate is the same): visual branch on {k}/14 segments scaled x14/{k} '
                      f'({t_vis:.1f} s/clip), audio branch + sync transformer in full ({t_rest:.2f} s); fp32 torch CPU oracle; host has '
                      f'{os.cpu_count()} cpus',
            'thread_sweep_s_per_segment': sweep}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node N for --gpus N > 1')
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    B = args.batch
    if args.workload == 'train' and args.dropin:
        import synchformer_amd as sa
        model = sa.instantiate_from_config(sa.sync_yaml_model_config())
        model.load_state_dict(synth.make_state_dict(1337), strict=True)
        model.seg_chunk = args.seg_chunk
        model = model.to(dev)
        for p_ in list(model.vfeat_extractor.parameters()) + list(model.afeat_extractor.parameters()):
            p_.requires_grad = False
        model.train(); model.vfeat_extractor.eval(); model.afeat_extractor.eval()
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local]) if world > 1 else model
        opt = torch.optim.Adam([p_ for p_ in model.parameters() if p_.requires_grad], lr=2e-6 * world, betas=(0.9, 0.999), eps=1e-7)
        scaler = torch.amp.GradScaler('cuda')
        targets = synth.make_targets(B, 21, seed=1337 + rank).to(dev)
        trainer = None

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
    elif args.workload == 'ft':
        # BASELINE configs[4]: configs/ft_synchability.yaml - frozen extractors, GlobalTransformerWithSyncabilityHead over 13 segments (184 tokens),
        # 2-way head, batch 16 per GPU; the extractors' qkv / proj / fc1 / fc2 run on MXFP8 operands (v_mfma_scale_f32_32x32x64_f8f6f4)
        from synchformer_amd.train import SyncTrainer
        trainer = SyncTrainer(synth.make_state_dict(1337, n_pos=184, n_out=2, head='sync_head'), dev, lr=2e-6 * world, seg_chunk=args.seg_chunk,
                              embd_pdrop=0.1, resid_pdrop=0.1, attn_pdrop=0.1, seed=1337 + rank, fp8_towers=True)
        eng = trainer.engine
        targets = synth.make_targets(B, 2, seed=1337 + rank).to(dev)
        step_fn = lambda v, a: trainer.train_step(v, a, targets)
    elif args.workload == 'train':
        from synchformer_amd.train import SyncTrainer
        trainer = SyncTrainer(synth.make_state_dict(1337), dev, lr=2e-6 * world, seg_chunk=args.seg_chunk, embd_pdrop=0.1, resid_pdrop=0.1,
                              attn_pdrop=0.1, seed=1337 + rank)
        eng = trainer.engine
        targets = synth.make_targets(B, 21, seed=1337 + rank).to(dev)
        step_fn = lambda v, a: trainer.train_step(v, a, targets)
    elif args.workload == 'stage1':
        # Stage-1 AVCLIP train step (configs/segment_avclip.yaml: base_batch_size 2 clips x 14 segments per GPU, both towers trainable)
        from synchformer_amd.stage1 import AVCLIPTrainer
        sd = {k: v for k, v in synth.make_state_dict(1337).items() if k.startswith(('vfeat_extractor.', 'afeat_extractor.'))}
        trainer = AVCLIPTrainer(sd, dev, lr=1e-4, drop_path_rate=0.2, seed=1337 + rank)     # train mode: DropPath 0.2 like the reference's towers
        step_fn = lambda v, a: trainer.train_step(v, a).reshape(1)
    else:
        eng = SynchformerEngine(synth.make_state_dict(1337), dev, seg_chunk=args.seg_chunk)
        step_fn = eng.forward
    S = 13 if args.workload == 'ft' else 14                                # ft_synchability.yaml: 13 segments (184-token sync transformer)
    vis = synth.make_video_u8(B, S, seed=1337 + rank).to(dev)             # (B,S,16,3,224,224) uint8, HBM-resident
    aud = synth.make_spectrogram(B, S, seed=1337 + rank).to(dev)          # (B,S,1,128,66) fp32

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.graph:
        assert args.workload == 'infer', '--graph serves the inference workload'
        step_fn = eng.capture(vis, aud)
    for _ in range(args.warmup):
        logits = step_fn(vis, aud)
    comm_ms = []
    timed_trainer = trainer if args.workload in ('train', 'stage1', 'ft') and not args.dropin else None
    if timed_trainer is not None and world > 1:
        timed_trainer.time_comm = True
    # ---- the timed region: EXACTLY K steps of the product configuration, no instrumentation -------------------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        logits = step_fn(vis, aud)
        if timed_trainer is not None and world > 1:
            comm_ms.append(timed_trainer._comm_ev)                 # events are read after the timed region
    barrier()
    dt = time.perf_counter() - t0
    assert torch.isfinite(logits).all()
    if timed_trainer is not None and world > 1:
        timed_trainer.time_comm = False
        mine = torch.tensor([timed_trainer.exposed_comm_ms()], device=dev, dtype=torch.float64)     # last step's stall on the gradient buckets
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        comm_ms = [round(float(t.item()), 3) for t in allr]
    # ---- roofline pass: the same K steps again with every sf_gemm_bf16 launch bracketed by HIP events on its launch stream.  The
    # product runs the audio tower on a second stream next to the visual one; an event pair on that stream would also time the queueing
    # behind the other tower's kernels, so for THIS pass the two towers are serialised on one stream (same kernels, same shapes, same
    # launch count; `value` above is not affected).  rocprofv3 --kernel-trace of this command sees both passes (profiles/).
    n_gemm, gemm_ms, gemm_flop, dt_roof = 0, 0.0, 0.0, 0.0
    n_mx, mx_ms, mx_flop, mx_bytes = 0, 0.0, 0.0, 0
    if not args.no_kernel_timing and not args.graph:
        serial = {'infer': lambda on: setattr(eng, 'audio_side_stream', on),
                  'train': lambda on: setattr(eng, 'audio_side_stream', on),
                  'ft': lambda on: setattr(eng, 'audio_side_stream', on),
                  'stage1': lambda on: setattr(trainer, 'two_streams', on)}[args.workload]
        serial(False)
        with GemmTimer() as gt:
            gt.enabled = rank == 0
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step_fn(vis, aud)
            barrier()
            dt_roof = time.perf_counter() - t1
            if gt.enabled:
                n_gemm, gemm_ms, gemm_flop = gt.summary()
                n_mx, mx_ms, mx_flop, mx_bytes = gt.mx_summary()
        serial(True)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    clips = B * world * args.steps
    value = clips / dt
    if rank == 0:
        out = {
            'metric': {'infer': 'clips/sec (14-seg offset pred)', 'train': 'clips/sec (Stage-2 train step, 14 segments)',
                       'stage1': 'clips/sec (Stage-1 AVCLIP train step, 14 segments)',
                       'ft': 'clips/sec (synchronizability fine-tune step, 13 segments, MXFP8 extractor GEMMs)'}[args.workload],
            'value': round(value, 3), 'unit': 'clips/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'mxfp8 (extractor Linears) + bf16' if args.workload == 'ft' else 'bf16',
            'data': 'synthetic',
            'config': {'workload': ('BASELINE configs[1]: batched offset inference, configs/sync.yaml model (237.5M params, '
                                    'random-init), uint8 224x224 frames + 128x66 log-mel resident in HBM, full forward to 21-way logits')
                       if args.workload == 'infer' else
                       ('Stage-1 segment-level contrastive train step (configs/segment_avclip.yaml): forward with saved activations + backward of '
                        'both towers (214.8M params), symmetric InfoNCE over B*14 segments, flat gradient all-reduce, fused clip+AdamW')
                       if args.workload == 'stage1' else
                       ('BASELINE configs[4]: synchronizability fine-tune step (configs/ft_synchability.yaml): frozen extractors with their qkv / proj / fc1 / '
                        'fc2 Linears on MXFP8 (OCP e4m3 + E8M0 per 32 k), 184-token sync transformer with the 2-way sync head trained in bf16 (22.6M params), '
                        'RCCL gradient all-reduce, fused clip+Adam')
                       if args.workload == 'ft' else
                       ('BASELINE configs[2]: Stage-2 sync-module train step (configs/sync.yaml, embd/resid/attn dropout 0.1): frozen extractors forward, '
                        'backward of proj + sync transformer (22.6M params), flat 90 MB RCCL gradient all-reduce, fused clip+Adam'),
                       'clips_per_gpu': B, 'segments': S, 'seg_chunk': args.seg_chunk, 'hip_graph': bool(args.graph),
                       'parallelism': f'replicas x{world}' if args.workload == 'infer' else f'dp{world}'},
            # stage1: forward + dgrad + wgrad of every linear ~ 3x the forward FLOPs (attention backward ~2.5x; approximate)
            'path_flop_per_clip': FLOP_PER_CLIP * (3 if args.workload == 'stage1' else 1) * S / 14,
            'path_mfma_frac': round(value * FLOP_PER_CLIP * (3 if args.workload == 'stage1' else 1) * S / 14 / (world * PEAK_BF16), 4),
            # the same with the FLOPs the engine executes (aggregators computed for their one consumed output row only)
            'path_mfma_frac_executed': round(value * FLOP_PER_CLIP_EXECUTED * (3 if args.workload == 'stage1' else 1) * S / 14 / (world * PEAK_BF16), 4),
        }
        if args.dropin:
            out['config']['dropin'] = 'nn.Module + autocast + GradScaler + clip_grad_norm_ + torch.optim.Adam' + (' + DistributedDataParallel' if world > 1 else '')
        if comm_ms:
            out['comm'] = {'exposed_ms_last_step_by_rank': comm_ms,
                           'what': 'time the compute stream waited for the gradient all-reduce (Stage-2: one flat 90 MB bucket after the backward; '
                                   'Stage-1: 7 buckets launched under the backward, the wait is for what did not overlap)'}
        if n_gemm:
            ach = gemm_flop / (gemm_ms * 1e-3) / 1e12
            out['roofline'] = {'bound': 'mfma', 'kernel': 'sf_gemm_bf16 + sf_gemm_res_ln768 + sf_qkv_time_attention (gemm_bf16_persistent_kernel, gemm_bf16_kernel, gemm_res_ln768_kernel, qkv_time_attn_kernel; GEMM FLOPs only - the fused LayerNorm / attention work of the last two is not counted)', 'achieved': round(ach, 1),
                               'peak': PEAK_BF16 / 1e12, 'unit': 'TFLOP/s', 'frac': round(ach / (PEAK_BF16 / 1e12), 4),
                               'traffic': pmc_traffic(), 'algorithmic_bytes_per_launch': round(gt.bytes / n_gemm),
                               'launches': n_gemm // args.steps,
                               'avg_launch_ms': round(gemm_ms / n_gemm, 4),
                               'flop_per_launch': gemm_flop / n_gemm,
                               'share_of_step_time': round(gemm_ms * 1e-3 / dt_roof, 3),
                               'measured': f'HIP events on the launch stream over a second pass of the same {args.steps} steps with the two towers '
                                           f'serialised on one stream ({1e3 * dt_roof / args.steps:.1f} ms/step); value is the un-instrumented two-stream pass'}
        if n_mx:
            ach = mx_flop / (mx_ms * 1e-3) / 1e12
            # FT: the dominant kernel is the MXFP8 GEMM; its roofline is the dense MX-fp8 matrix peak (MI355X_MICROARCH.md: ~5 PFLOP/s)
            out['roofline'] = {'bound': 'mfma', 'kernel': 'sf_gemm_mxfp8 (gemm_mxfp8_persistent_kernel, v_mfma_scale_f32_32x32x64_f8f6f4)',
                               'achieved': round(ach, 1), 'peak': 5000.0, 'unit': 'TFLOP/s', 'frac': round(ach / 5000.0, 4), 'traffic': None,
                               'algorithmic_bytes_per_launch': round(mx_bytes / n_mx), 'launches': n_mx // args.steps,
                               'avg_launch_ms': round(mx_ms / n_mx, 4), 'flop_per_launch': mx_flop / n_mx,
                               'share_of_step_time': round(mx_ms * 1e-3 / dt_roof, 3),
                               'bf16_gemm_launches': {'launches': n_gemm // args.steps, 'tflops': round(gemm_flop / max(gemm_ms, 1e-9) / 1e9, 1)},
                               'measured': f'HIP events on the launch stream over a second pass of the same {args.steps} steps, towers serialised on one stream'}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
