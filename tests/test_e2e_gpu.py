"""End-to-end parity of the HIP forward path (through the C ABI) against
  (a) golden outputs of the REAL reference (tests/golden/*.npz, made by tests/golden/make_golden.py), and
  (b) the CPU oracle on the same seeded inputs, at sizes the oracle finishes in seconds.
Tolerance (north_star: "within a stated fp tolerance"): the HIP path computes GEMMs/attention in bf16 with fp32
accumulation and keeps the residual stream, LayerNorm and softmax statistics in fp32; the reference's own fp32 vs
bf16-autocast logit gap at this init is 4.3e-3.  Bars used below:
  logits           |delta| <= 1.5e-2  (gain 1)   / 4e-2 (gain 2, where clips differ by ~0.2)
  segment features relative RMS error <= 1.5 %, max |delta| <= 4e-2 (features have std ~0.44)"""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / 'golden'


def _rel_rms(a, b):
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


def _engine(gpu, **kw):
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    sd = synth.make_state_dict(1337, **kw)
    return SynchformerEngine(sd, gpu), sd


def _inputs(B, S):
    from synchformer_amd import synth
    return synth.make_video_u8(B, S, 1337), synth.make_spectrogram(B, S, 1337)


@pytest.mark.parametrize('tag,gain,tol', [('e2e_sync_B2', 1.0, 1.5e-2), ('e2e_sync_gain2_B2', 2.0, 4e-2)])
def test_golden_sync_logits_and_features(gpu, tag, gain, tol):
    f = GOLD / f'{tag}.npz'
    if not f.exists():
        pytest.skip(f'{f.name} missing')
    g = np.load(f)
    B = int(g['B'])
    eng, _ = _engine(gpu, gain=gain)
    u8, aud = _inputs(B, 14)
    vf = eng.extract_vfeats(u8.to(gpu))          # uint8 in: the fused RGB front-end is part of the path
    af = eng.extract_afeats(aud.to(gpu))
    logits = eng.sync_transformer(vf, af).cpu()
    gv = torch.from_numpy(g['vfeat_extractor__spatial_attn_agg']).reshape(B, 14, 8, 768)
    ga = torch.from_numpy(g['afeat_extractor__freq_attn_agg']).reshape(B, 14, 6, 768)
    ev, ea = _rel_rms(vf.cpu(), gv), _rel_rms(af.cpu(), ga)
    dv, da = (vf.cpu() - gv).abs().max().item(), (af.cpu() - ga).abs().max().item()
    dl = (logits - torch.from_numpy(g['logits'])).abs().max().item()
    print(f'{tag}: vfeat relrms {ev:.4f} max {dv:.4f} | afeat relrms {ea:.4f} max {da:.4f} | logits max {dl:.5f}')
    assert ev < 1.5e-2 and ea < 1.5e-2, (ev, ea)
    assert dv < 4e-2 * gain and da < 4e-2 * gain, (dv, da)
    assert dl < tol, dl
    assert (logits.argmax(-1) == torch.from_numpy(g['logits']).argmax(-1)).all()


@pytest.mark.parametrize('fuse', [True, False])
def test_golden_per_block_outputs(gpu, fuse):
    """The residual stream right after visual blocks 0, 5 and 11 against the REAL reference's hooked block outputs (e2e_sync_B2.npz keeps
    token rows 0, 1, 2, 197, 1000, 1568 of segments 0, 13 and 27): an error in one DividedSpaceTimeBlock is caught at that block, not 12
    blocks later.  Both schedules (fused GEMM + LayerNorm kernel on 28 segments = 43,932 rows, and the un-fused pair).  Bars: relative RMS
    1 %; max |delta| 2e-2 after block 0 growing to 7.5e-2 after block 11 (bf16 GEMM operands, fp32 stream of magnitude up to ~10; measured 3.6e-2)."""
    g = np.load(GOLD / 'e2e_sync_B2.npz')
    eng, _ = _engine(gpu)
    eng.fuse_ln = fuse
    eng.capture_blocks = {}
    u8, _ = _inputs(2, 14)
    eng.extract_vfeats(u8.to(gpu))
    rows = [0, 1, 2, 197, 1000, 1568]
    for bi in (0, 5, 11):
        x = eng.capture_blocks[bi].view(28, 1569, 768)[[0, 13, 27]][:, rows].cpu()
        ref = torch.from_numpy(g[f'vfeat_extractor__blocks__{bi}'])
        assert ref.shape == x.shape, (ref.shape, x.shape)
        rel, mx = _rel_rms(x, ref), (x - ref).abs().max().item()
        print(f'block {bi} (fuse_ln {fuse}): rel-RMS {rel:.5f} max {mx:.5f}')
        assert rel < 1e-2 and mx < 2e-2 * (1 + bi / 4), (bi, rel, mx)
    eng.capture_blocks = None


def test_golden_syncability(gpu):
    """S=13 segments, 184-token sync transformer, 2-way sync_head (configs/ft_synchability.yaml)."""
    g = np.load(GOLD / 'e2e_syncability_B1.npz')
    eng, _ = _engine(gpu, n_pos=184, n_out=2, head='sync_head')
    u8, aud = _inputs(1, 13)
    logits = eng.forward(u8.to(gpu), aud.to(gpu)).cpu()
    dl = (logits - torch.from_numpy(g['logits'])).abs().max().item()
    print('syncability logits max delta', dl)
    assert dl < 1.5e-2, dl


def test_mxfp8_towers_bound(gpu):
    """BASELINE configs[4]: the synchronizability model (13 segments, 184 tokens, 2-way sync head) with the frozen extractors' Linears on MXFP8
    operands (engine fp8_towers=True) against the bf16 engine on the same inputs and against the REAL reference's golden logits.  Stated bound of
    the fp8 path at this (reference-like, gain 1) initialisation: segment features within 12 % relative RMS of the bf16 ones (measured 7.7 %: two
    e4m3 roundings per product, 72 quantised GEMMs deep), logits within 3e-2 of the bf16 logits (measured 1e-2) and within 4e-2 of the fp32 reference."""
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    g = np.load(GOLD / 'e2e_syncability_B1.npz')
    sd = synth.make_state_dict(1337, n_pos=184, n_out=2, head='sync_head')
    e16, e8 = SynchformerEngine(sd, gpu), SynchformerEngine(sd, gpu, fp8_towers=True)
    u8, aud = synth.make_video_u8(1, 13, 1337).to(gpu), synth.make_spectrogram(1, 13, 1337).to(gpu)
    v16, v8 = e16.extract_vfeats(u8), e8.extract_vfeats(u8)
    l16, l8 = e16.forward(u8, aud).cpu(), e8.forward(u8, aud).cpu()
    rel = _rel_rms(v8.cpu(), v16.cpu())
    d16, dref = (l8 - l16).abs().max().item(), (l8 - torch.from_numpy(g['logits'])).abs().max().item()
    print(f'mxfp8 towers: vfeat rel-RMS vs bf16 {rel:.4f} | logits max |d| vs bf16 {d16:.4f}, vs fp32 reference {dref:.4f}')
    assert 1e-3 < rel < 0.12, rel                   # > 1e-3: the fp8 kernels really ran
    assert d16 < 3e-2 and dref < 4e-2, (d16, dref)
    assert (l8.argmax(-1) == torch.from_numpy(g['logits']).argmax(-1)).all()
    # the fused proj / fc2 + residual + LayerNorm + quantisation launches (sf_gemm_mx_res_ln768, the default) against the un-fused pairs: the same
    # products summed in another order - features agree far inside the fp8 path's own noise
    # the attention kernels writing the projections' MXFP8 operands themselves (sf_attention_cls_partial_mx, sf_qkv_time_attention_mx_q - the default) against
    # their bf16 outputs + sf_quantize_mxfp8: the same bytes, so the same features bit for bit
    # (compared on round 3's launches: sf_qkv_space_attention_mx - the default since round 4 - and sf_qkv_time_attention2_mx - round 5 - sum the projection in another tile order)
    e8.fuse_space = e8.fuse_time2 = False
    v8s = e8.extract_vfeats(u8)
    rels = _rel_rms(v8.cpu(), v8s.cpu())
    print(f'mxfp8 towers, round-5 fused launches vs round 3\'s (spatial qkv + space attention un-fused, sf_qkv_time_attention_mx): vfeat rel-RMS {rels:.5f}')
    assert 0 < rels < 3e-2
    e8.fuse_mx_attn = False
    assert torch.equal(e8.extract_vfeats(u8), v8s)
    e8.fuse_mx_attn = True
    e8.fuse_space = True                                                  # the temporal half alone on the round-5 launch
    relt = _rel_rms(e8.extract_vfeats(u8).cpu(), v8.cpu())
    e8.fuse_time2 = True
    print(f'mxfp8 towers, sf_qkv_time_attention2_mx vs sf_qkv_time_attention_mx (space fused in both): vfeat rel-RMS {relt:.5f}')
    assert 0 < relt < 3e-2
    e8.fuse_mx_ln = e8.fuse_mx_time = False
    v8u, l8u = e8.extract_vfeats(u8), e8.forward(u8, aud).cpu()
    relu = _rel_rms(v8.cpu(), v8u.cpu())
    print(f'mxfp8 towers, fused vs un-fused res+LN: vfeat rel-RMS {relu:.5f} | logits max |d| {(l8 - l8u).abs().max().item():.5f}')
    assert 0 < relu < 3e-2 and (l8 - l8u).abs().max().item() < 5e-3      # (two fp8 runs whose roundings differ in a few places decorrelate by ~1-2 %: a fifth of the path's own 7.8 % noise)


def test_forward_through_the_dispatcher(gpu):
    """north_star: "registered as PyTorch-ROCm custom ops".  The fused launches of the default schedule are `torch.ops.synchformer.*` operators of the compiled
    dispatcher library (TORCH_LIBRARY in csrc/sf_torch_library.cpp, round 5); with ops.via_dispatcher() the engine's forward - bf16 and MXFP8 towers, masked and
    unmasked - runs through the PyTorch dispatcher and gives the same bits as the direct ctypes path."""
    from synchformer_amd import ops, synth
    from synchformer_amd.engine import SynchformerEngine
    for name in ('gemm_bf16', 'layernorm768', 'gemm_res_ln768', 'qkv_time_attention', 'attention_cls_partial', 'attention_cls_combine', 'gemm_mxfp8',
                 'gemm_mx_res_ln768', 'quantize_mxfp8', 'layernorm768_mxfp8', 'qkv_time_attention_mx_q', 'attention_cls_partial_mx', 'attention_cls_combine_mx',
                 'qkv_time_attention2', 'qkv_space_attention', 'qkv_space_attention_mx_q', 'space_side_rows'):
        assert hasattr(torch.ops.synchformer, name), name
    for fp8 in (False, True):
        sd = synth.make_state_dict(1337, n_pos=184, n_out=2, head='sync_head') if fp8 else synth.make_state_dict(1337)
        S = 13 if fp8 else 14
        eng = SynchformerEngine(sd, gpu, fp8_towers=fp8)
        u8, aud = synth.make_video_u8(6, S, 7).to(gpu), synth.make_spectrogram(6, S, 7).to(gpu)      # 6 clips: the fused (large-batch) schedule
        direct = eng.forward(u8, aud).clone()
        with ops.via_dispatcher() as d:
            disp = eng.forward(u8, aud).clone()
        assert d.calls > 100, d.calls          # (launches with row maps or an explicit M keep the direct path)
        assert torch.equal(direct, disp)
        # round 3's launches (SF_FUSE_SPACE / SF_FUSE_TIME2 off) are operators of the same library
        eng.fuse_space = eng.fuse_time2 = False
        direct3 = eng.forward(u8, aud).clone()
        with ops.via_dispatcher():
            assert torch.equal(eng.forward(u8, aud), direct3)
        eng.fuse_space = eng.fuse_time2 = True
        if not fp8:                            # token masks: the *_masked operators (round 5)
            vm, am = synth.make_masks(6, S, 7)
            dm = eng.forward(u8, aud, vm.to(gpu), am.to(gpu)).clone()
            with ops.via_dispatcher() as d2:
                assert torch.equal(eng.forward(u8, aud, vm.to(gpu), am.to(gpu)), dm)
            assert d2.calls > 100 and not torch.equal(dm, direct)


def test_two_stream_split_of_small_batches(gpu):
    """Round 6: between 14 and 112 segments (one to eight clips) a CAPTURED forward runs the visual tower as two halves of the segments on two HIP streams with workspaces of their
    own (engine._parts; profiles/r06_small_m.md; eager forwards keep one stream unless vis_split_mode = 'always': issued eagerly the doubled launch count is a host
    cost).  The halves are independent until vproj, so the split changes launch geometry only: logits within the bar of test_benchmarked_geometry_parity's geometry case
    (tile configurations follow M), bit-identical on repetition, the graph replays them bit for bit; fewer than 14 segments keep the single-stream schedule."""
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    sd = synth.make_state_dict(1337)
    eng = SynchformerEngine(sd, gpu)
    assert (eng.vis_split_min, eng.vis_split_max, eng.vis_split_mode) == (14, 112, 'graph')
    u8, aud = synth.make_video_u8(2, 14, 5).to(gpu), synth.make_spectrogram(2, 14, 5).to(gpu)
    single = eng.forward(u8, aud).clone()
    assert eng._v_side is None                                      # eager: one stream
    eng.vis_split_mode = 'always'
    split = eng.forward(u8, aud).clone()
    assert eng._v_side is not None                                  # the second visual stream was used
    assert torch.equal(eng.forward(u8, aud), split)
    eng.vis_split_mode = 'graph'
    d = (split - single).abs().max().item()
    print(f'two-stream split vs single stream, 2 clips: max |dlogit| {d:.5f}')
    assert d < 8e-3
    run = eng.capture(u8, aud)                                      # the captured forward takes the split
    assert torch.equal(run(u8, aud), split)
    assert torch.equal(eng.forward(u8, aud), single)                # ... and leaves the eager schedule alone
    u1, a1 = u8[:1, :6].contiguous(), aud[:1, :6].contiguous()
    eng2 = SynchformerEngine(synth.make_state_dict(1337, n_pos=2 + 6 * 14), gpu)
    run1 = eng2.capture(u1, a1)
    assert eng2._v_side is None and torch.isfinite(run1(u1, a1)).all()      # 6 segments: below the window, also under capture


def test_dispatcher_operators_check_their_buffers(gpu):
    """the C ABI sees raw pointers, so the public `torch.ops.synchformer.*` operators must refuse what would become a silent out-of-bounds device access or a
    reinterpreted buffer - wrong dtype, too few rows, partials too small, out aliasing x - exactly as the ctypes wrappers of ops.py do; and they launch on the tensor's device."""
    from synchformer_amd import ops
    ops.register_torch_ops()
    T = torch.ops.synchformer
    n = 2
    rows = n * 1569
    x = torch.zeros(rows, 768, device=gpu, dtype=torch.bfloat16)
    w = torch.zeros(2304, 768, device=gpu, dtype=torch.bfloat16)
    b = torch.zeros(2304, device=gpu)
    side = torch.zeros(n * 33, 2304, device=gpu, dtype=torch.bfloat16)
    out = torch.empty_like(x)
    p_t, p_s = torch.empty(n * 12 * 33 * 66, device=gpu), torch.empty(n * 12 * 8 * 66, device=gpu)
    T.qkv_time_attention2(x, w, b, side, out, p_t, n, 0.125)                      # the well-formed calls pass
    T.qkv_space_attention(x, w, b, side, out, p_s, n, 0.125)
    torch.cuda.synchronize()
    bad = [
        (lambda: T.qkv_time_attention2(x.float(), w, b, side, out, p_t, n, 0.125), 'bf16'),                      # fp32 x would be read as bf16
        (lambda: T.qkv_time_attention2(x, w, b.bfloat16(), side, out, p_t, n, 0.125), 'bias'),                    # bf16 bias would be read as fp32
        (lambda: T.qkv_time_attention2(x[:rows - 1], w, b, side, out, p_t, n, 0.125), 'rows'),                    # too few rows
        (lambda: T.qkv_time_attention2(x, w, b, side, out[:100], p_t, n, 0.125), 'rows'),
        (lambda: T.qkv_time_attention2(x, w, b, side[:n * 33 - 1], out, p_t, n, 0.125), 'side'),
        (lambda: T.qkv_time_attention2(x, w, b, side, out, p_s, n, 0.125), 'partials'),                           # the spatial launch's (smaller) partials
        (lambda: T.qkv_time_attention2(x, w, b, side, x, p_t, n, 0.125), 'alias'),
        (lambda: T.qkv_space_attention(x, w[:, :512], b, side, out, p_s, n, 0.125), r'2304, 768'),
        (lambda: T.qkv_space_attention(x, w, b, side, out, p_s.double(), n, 0.125), 'fp32'),
        (lambda: T.qkv_space_attention_masked(x, w, b, side, out, p_s, n, 0.125, torch.ones(rows - 1, device=gpu, dtype=torch.uint8)), 'key_keep'),
        (lambda: T.space_side_rows(x, side[:, :768].contiguous()[:10], n), 'rows'),
        (lambda: T.attention_cls_combine(p_s[:100], out, 8, n, 1569, 0, 12), 'partials'),
        (lambda: T.gemm_bf16(x, w, b.bfloat16(), torch.empty(rows, 2304, device=gpu, dtype=torch.bfloat16), None, False), 'bias'),
        (lambda: T.gemm_bf16(x, w, b, torch.empty(rows - 1, 2304, device=gpu, dtype=torch.bfloat16), None, False), 'out'),
    ]
    for fn, pat in bad:
        with pytest.raises(RuntimeError, match=pat):
            fn()
    torch.cuda.synchronize()


@pytest.mark.parametrize('fp8', [False, True])
def test_forward_soak_bitwise_repeatable(gpu, fp8):
    """Race screen of the whole default schedule: every kernel of the hot path waits for its LDS-DMA operands with hand-counted `vmcnt` waits that hipcc
    does not see, and reuses LDS regions one or two phases after their last read.  A mistake there shows up as rare wrong tiles, not as a crash: run the
    fused (large-batch) forward 25 times on the same inputs - bf16 towers and MXFP8 towers - and require the same bits every time."""
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    sd = synth.make_state_dict(1337, n_pos=184, n_out=2, head='sync_head') if fp8 else synth.make_state_dict(1337)
    S = 13 if fp8 else 14
    eng = SynchformerEngine(sd, gpu, fp8_towers=fp8)
    u8, aud = synth.make_video_u8(6, S, 11).to(gpu), synth.make_spectrogram(6, S, 11).to(gpu)
    first_v, first_l = eng.extract_vfeats(u8).clone(), eng.forward(u8, aud).clone()
    assert torch.isfinite(first_l).all()
    for rep in range(25):
        assert torch.equal(eng.forward(u8, aud), first_l), f'logits differ in repetition {rep}'
    assert torch.equal(eng.extract_vfeats(u8), first_v)


def test_oracle_small(gpu):
    """2 segments through both extractors + the sync transformer on random features, vs the CPU oracle."""
    from oracle import synchformer_cpu as O
    eng, sd = _engine(gpu)
    u8, aud = _inputs(1, 2)
    with torch.no_grad():
        ov = O.extract_vfeats(O.rgb_frontend(u8), sd)
        oa = O.extract_afeats(aud, sd)
    vf, af = eng.extract_vfeats(u8.to(gpu)).cpu(), eng.extract_afeats(aud.to(gpu)).cpu()
    assert _rel_rms(vf, ov) < 1.5e-2 and _rel_rms(af, oa) < 1.5e-2, (_rel_rms(vf, ov), _rel_rms(af, oa))
    # input dtypes: fp16 / bf16 / fp32 frames (what a reference caller passes) must agree with the u8 path
    x = O.rgb_frontend(u8)
    for dt in (torch.float16, torch.float32):
        v2 = eng.extract_vfeats(x.to(dt).to(gpu)).cpu()
        assert _rel_rms(v2, vf) < 2e-3, dt
    g = torch.Generator().manual_seed(3)
    v, a = torch.randn(3, 14, 8, 768, generator=g), torch.randn(3, 14, 6, 768, generator=g)
    with torch.no_grad():
        ol = O.global_transformer(O._lin(v, sd, 'vproj').reshape(3, -1, 768), O._lin(a, sd, 'aproj').reshape(3, -1, 768), sd)
    gl = eng.sync_transformer(v.to(gpu), a.to(gpu)).cpu()
    assert (gl - ol).abs().max().item() < 1.5e-2, (gl - ol).abs().max().item()
    # attempt_to_apply_heads=False (sync_model.py:170-172, what GlobalTransformerWithSyncabilityHead asks its parent for, :187): ln_f of EVERY token
    pv, pa = O._lin(v, sd, 'vproj').reshape(3, -1, 768), O._lin(a, sd, 'aproj').reshape(3, -1, 768)
    with torch.no_grad():
        ot = O.global_transformer(pv, pa, sd, apply_head=False)
    gt = eng.global_transformer(pv.to(gpu), pa.to(gpu), apply_head=False).cpu()
    assert gt.shape == ot.shape == (3, 198, 768) and _rel_rms(gt, ot) < 1.5e-2, _rel_rms(gt, ot)


def test_chunking_invariance(gpu):
    """seg_chunk only changes scheduling: results must be bit-identical (segments are independent)."""
    eng, _ = _engine(gpu)
    u8, _ = _inputs(1, 5)
    eng.seg_chunk = 5
    a = eng.extract_vfeats(u8.to(gpu)).clone()
    eng.seg_chunk = 2
    b = eng.extract_vfeats(u8.to(gpu)).clone()
    assert torch.equal(a, b)


def test_benchmarked_geometry_parity(gpu):
    """The launch geometry bench.py times - 16 clips x 14 segments = 224 segments in ONE chunk (351,456 token rows: X 1.08 GB, qkv / hidden
    buffers 1.6 / 2.2 GB, 32-bit buffer offsets close to their 4 GiB limit; configs/sync.yaml:63 batch 16) - against the same inputs run
    14 segments at a time, for both the fused (sf_gemm_res_ln768) and the un-fused schedule, and clips 0-1 against the REAL reference's
    golden outputs (e2e_sync_B2.npz).  Bitwise equality between chunkings is not a property of the path: sf_gemm_bf16 picks its tile
    configuration (256x256 on 32x32x16 MFMA / 128x128 on 16x16x32) by M, and the fused kernel rotates its k-loop per workgroup, so fp32
    summation orders differ, and a last-bit fp32 difference that flips one bf16 rounding (ulp 4e-3 at 1.0) propagates through the 12 blocks.
    Bars: features (std 0.44) within 2e-2 and logits within 8e-3 absolute between any two schedules (measured 8e-3 / 3e-3; an addressing
    fault in the big geometry is an O(0.5) error on at least one segment), and the golden bars of test_golden_sync_logits_and_features."""
    from synchformer_amd import synth
    g = np.load(GOLD / 'e2e_sync_B2.npz')
    eng, _ = _engine(gpu)
    u8 = torch.cat([synth.make_video_u8(2, 14, 1337), synth.make_video_u8(14, 14, 4242)]).to(gpu)     # clips 0-1 = the golden's inputs
    aud = torch.cat([synth.make_spectrogram(2, 14, 1337), synth.make_spectrogram(14, 14, 4242)]).to(gpu)
    res = {}
    for fuse in (False, True):
        eng.fuse_ln = fuse
        for chunk in (224, 14):
            eng.seg_chunk = chunk
            vf = eng.extract_vfeats(u8)
            af = eng.extract_afeats(aud)
            res[fuse, chunk] = (vf.clone(), af.clone(), eng.sync_transformer(vf, af).clone())
    ref = res[False, 14]
    for key, out in res.items():
        d = [(a - b).abs().max().item() for a, b in zip(out, ref)]
        print(f'fuse_ln {key[0]} seg_chunk {key[1]}: max |delta| vs un-fused/14: vfeat {d[0]:.2e} afeat {d[1]:.2e} logits {d[2]:.2e}')
        assert d[0] < 2e-2 and d[1] < 2e-2 and d[2] < 8e-3, (key, d)
        assert all(torch.isfinite(t).all() for t in out)
    vf, af, logits = (t.cpu() for t in res[True, 224])
    gv = torch.from_numpy(g['vfeat_extractor__spatial_attn_agg']).reshape(2, 14, 8, 768)
    ga = torch.from_numpy(g['afeat_extractor__freq_attn_agg']).reshape(2, 14, 6, 768)
    assert _rel_rms(vf[:2], gv) < 1.5e-2 and _rel_rms(af[:2], ga) < 1.5e-2
    assert (logits[:2] - torch.from_numpy(g['logits'])).abs().max().item() < 1.5e-2


def test_full_size_clip_permutation_equivariance(gpu):
    """A size-independent property at the benchmarked geometry (16 clips x 14 segments, one 224-segment chunk): offset prediction is per clip (sync_model.py:38-70: no
    operation mixes clips), so permuting the clips of a batch permutes the logits - a clip's logits do not depend on its batch position or neighbours, although every
    launch of the path tiles ACROSS clips (a 256-row GEMM tile, a 128-row fused tile and a 24-patch attention block all straddle segment boundaries).  Not bitwise: the
    fused GEMM + LayerNorm kernel rotates its k-loop per workgroup, so a row's fp32 summation order depends on where its tile falls; bar (gain-2 weights, where clips
    differ by ~0.2): 1.6e-2, twice test_benchmarked_geometry_parity's gain-1 bar.  The same forward twice IS bitwise (fixed schedule, no atomics)."""
    from synchformer_amd import synth
    eng, _ = _engine(gpu, gain=2.0)
    u8, aud = synth.make_video_u8(16, 14, 2024).to(gpu), synth.make_spectrogram(16, 14, 2024).to(gpu)
    base = eng.forward(u8, aud).clone()
    assert torch.equal(eng.forward(u8, aud), base)
    perm = torch.tensor([11, 3, 7, 0, 15, 9, 1, 13, 5, 2, 14, 8, 6, 12, 4, 10], device=gpu)
    got = eng.forward(u8[perm].contiguous(), aud[perm].contiguous())
    d = (got - base[perm]).abs().max().item()
    spread = (base - base.mean(0, keepdim=True)).abs().max().item()
    print(f'clip permutation at 16 x 14: max |logits(perm) - perm(logits)| {d:.2e}; clips differ by up to {spread:.3f}')
    assert d < 1.6e-2 and spread > 5 * d, (d, spread)


def test_dropin_module_matches_golden(gpu):
    """The reference-shaped plugin path: instantiate_from_config(sync.yaml model) -> load_state_dict -> model(vis, aud, targets)
    returns (loss, logits) like Synchformer.forward (sync_model.py:38-70); checked against the real reference's outputs."""
    import synchformer_amd as sa
    from synchformer_amd import synth
    g = np.load(GOLD / 'e2e_sync_B2.npz')
    model = sa.instantiate_from_config(sa.sync_yaml_model_config())
    status = model.load_state_dict(synth.make_state_dict(1337), strict=True)
    assert not status.missing_keys and not status.unexpected_keys
    model = model.to(gpu).eval()
    u8, aud = _inputs(2, 14)
    tgt = torch.from_numpy(g['targets']).to(gpu)
    from synchformer_amd import ops
    assert model.dispatcher_route                                              # the drop-in module's launches are dispatcher calls (torch.ops.synchformer.*) by default ...
    n0 = ops.via_dispatcher.calls_total
    with torch.no_grad():
        loss, logits = model(u8.to(gpu), aud.to(gpu), tgt)
    assert ops.via_dispatcher.calls_total - n0 > 100
    assert (logits.cpu() - torch.from_numpy(g['logits'])).abs().max().item() < 1.5e-2
    assert abs(loss.item() - float(g['loss'])) < 1e-2
    model.dispatcher_route = False                                             # ... and the direct ctypes route gives the same logits bit for bit
    n1 = ops.via_dispatcher.calls_total
    with torch.no_grad():
        _, direct = model(u8.to(gpu), aud.to(gpu), tgt)
    assert ops.via_dispatcher.calls_total == n1 and torch.equal(direct, logits)
    model.dispatcher_route = True
    # fp16 frames, as the reference's RGBToHalfToZeroOne pipeline delivers them (dataset/transforms.py:653)
    from oracle import synchformer_cpu as O
    with torch.no_grad():
        _, l16 = model(O.rgb_frontend(u8).half().to(gpu), aud.to(gpu))
    assert (l16 - logits).abs().max().item() < 2e-3
    # a weight update must invalidate the cached engine
    with torch.no_grad():
        model.transformer.off_head.bias.add_(1.0)
        _, l2 = model(u8.to(gpu), aud.to(gpu))
    assert (l2 - logits - 1.0).abs().max().item() < 1e-4


def test_checkpoint_adapter_path_reproduces_golden(gpu, tmp_path):
    """f4 end to end on the GPU: the towers come out of a Stage-1 `epoch_best.pt` through `MotionFormer(ckpt_path=)` / `AST(ckpt_path=)` (the
    adapters of synchformer_amd.checkpoint, pinned to the reference's loaders by test_checkpoint_adapters_match_reference_loaders), the sync module
    out of a released-format `ckpt['model']` saved from a DDP wrapper - and the forward reproduces the real reference's golden logits."""
    import sys
    import synchformer_amd as sa
    from synchformer_amd import checkpoint as ck, synth
    sys.path.insert(0, str(GOLD))
    import ckpt_fixtures as cf
    g = np.load(GOLD / 'e2e_sync_B2.npz')
    s1 = cf.write_stage1_ckpt(tmp_path / 'epoch_best.pt', seed=1337, gain=1.0)
    cfg = sa.sync_yaml_model_config()
    cfg['params']['vfeat_extractor']['params']['ckpt_path'] = str(s1)
    cfg['params']['afeat_extractor']['params']['ckpt_path'] = str(s1)
    model = sa.instantiate_from_config(cfg)
    sd = synth.make_state_dict(1337)
    torch.save({'model': {'module.' + k: v for k, v in sd.items() if not k.startswith(('vfeat_extractor.', 'afeat_extractor.'))}, 'epoch': 3}, tmp_path / 'sync.pt')
    status = model.load_state_dict(ck.synchformer_state(ck.load_file(tmp_path / 'sync.pt')), strict=False)
    assert not status.unexpected_keys and all(k.startswith(('vfeat_extractor.', 'afeat_extractor.')) for k in status.missing_keys)
    model = model.to(gpu).eval()
    u8, aud = _inputs(2, 14)
    with torch.no_grad():
        loss, logits = model(u8.to(gpu), aud.to(gpu), torch.from_numpy(g['targets']).to(gpu))
    assert (logits.cpu() - torch.from_numpy(g['logits'])).abs().max().item() < 1.5e-2
    assert abs(loss.item() - float(g['loss'])) < 1e-2


def test_graph_replay_matches_eager(gpu):
    """engine.capture(): the whole forward as one HIP graph; replay on new inputs must reproduce the eager launches bit for bit."""
    from synchformer_amd import synth
    eng, _ = _engine(gpu, gain=2.0)
    u8a, auda = synth.make_video_u8(1, 3, 11), synth.make_spectrogram(1, 3, 11)
    u8b, audb = synth.make_video_u8(1, 3, 12), synth.make_spectrogram(1, 3, 12)
    run = eng.capture(u8a.to(gpu), auda.to(gpu))
    for u8, aud in ((u8b, audb), (u8a, auda)):
        got = run(u8.to(gpu), aud.to(gpu)).clone()
        ref = eng.forward(u8.to(gpu), aud.to(gpu))
        assert torch.equal(got, ref)
    with pytest.raises(ValueError):
        run(synth.make_video_u8(2, 3, 1).to(gpu), synth.make_spectrogram(2, 3, 1).to(gpu))


def test_token_masks_match_reference_golden(gpu):
    """Synchformer.forward(vis_mask=, aud_mask=): NaN-trick token masks + key masking in every tower attention, against the REAL
    reference (tests/golden/e2e_masked_B1S2.npz) and the oracle's token masks bit for bit."""
    from synchformer_amd import ops, synth
    from synchformer_amd.engine import SynchformerEngine
    from oracle import synchformer_cpu as O
    g = np.load(GOLD / 'e2e_masked_B1S2.npz')
    B, S, gain = int(g['B']), int(g['S']), float(g['gain'])
    sd = synth.make_state_dict(1337, gain=gain, n_pos=2 + S * 14)
    eng = SynchformerEngine(sd, gpu)
    u8, aud = _inputs(B, S)
    vm, am = synth.make_masks(B, S, 1337)
    # token masks: exact
    n = B * S
    tk = ops.token_mask_video(vm.reshape(n, 16, 3, 224, 224).to(gpu), eng.v_w0_sign, torch.empty(n * 1569, device=gpu, dtype=torch.uint8)).cpu()
    w = sd['vfeat_extractor.patch_embed_3d.proj.weight']
    keep = vm.reshape(n, 16, 3, 224, 224).permute(0, 2, 1, 3, 4)                       # (n, C, T, H, W)
    pk = keep.reshape(n, 3, 8, 2, 14, 16, 14, 16).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(n, 1568, 1536)
    ref_tk = torch.cat([torch.ones(n, 1, dtype=torch.bool), O.token_mask_from_content(pk, w[0].reshape(-1))], 1)
    assert torch.equal(tk.view(n, 1569).bool(), ref_tk)
    assert 0 < int((~ref_tk).sum()) < ref_tk.numel() // 2
    vf = eng.extract_vfeats(u8.to(gpu), vm.to(gpu)).cpu()
    af = eng.extract_afeats(aud.to(gpu), am.to(gpu)).cpu()
    logits = eng.forward(u8.to(gpu), aud.to(gpu), vm.to(gpu), am.to(gpu)).cpu()
    ev = _rel_rms(vf.reshape(-1, 768), torch.from_numpy(g['vfeat']).reshape(-1, 768))
    ea = _rel_rms(af.reshape(-1, 768), torch.from_numpy(g['afeat']).reshape(-1, 768))
    dl = (logits - torch.from_numpy(g['logits'])).abs().max().item()
    dn = (logits - torch.from_numpy(g['logits_nomask'])).abs().max().item()
    print(f'masked: vfeat relrms {ev:.4f} afeat relrms {ea:.4f} logits max {dl:.5f} (distance to the unmasked logits {dn:.3f})')
    assert ev < 1.5e-2 and ea < 1.5e-2 and dl < 4e-2 and dn > 0.1
    # an all-ones mask must be a no-op, bit for bit: since round 3 the masked forward runs the same fused schedule (space attention with CLS partials,
    # at >= 6 segments also the fused temporal qkv + time attention) with key flags
    ones_v, ones_a = torch.ones_like(vm), torch.ones_like(am)
    got = eng.forward(u8.to(gpu), aud.to(gpu), ones_v.to(gpu), ones_a.to(gpu))
    ref = eng.forward(u8.to(gpu), aud.to(gpu))
    assert torch.equal(got, ref)


def test_masked_forward_on_the_fused_schedule(gpu):
    """Masks at a batch large enough for every fused launch (6 segments = 9414 token rows: sf_qkv_time_attention2_masked, sf_qkv_space_attention_masked,
    sf_gemm_res_ln768).  Round 6: against the REAL reference's masked forward on the same inputs (tests/golden/e2e_masked_B1S6.npz,
    make_golden.py::e2e_masked_fused - sync_model.py:72-89, vit_helper.py:107-141) - logits, segment features, and token rows of blocks 0 and 11 -, then against
    the un-fused masked schedule of round 2, and an all-ones mask bit-equal to no mask.  One whole frame, one whole 4-patch x 8-frame wave, one whole patch column
    and the left-over patches 192-195 are masked on top of the random boxes (synth.make_masks_fused_case), so that CLS partial records and whole time groups with
    every key masked (m = -inf, l = 0) go through the kernels."""
    import os
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    g = np.load(GOLD / 'e2e_masked_B1S6.npz')
    B, S, seed = int(g['B']), int(g['S']), int(g['seed'])
    assert (B, S) == (1, 6) and B * S * 1569 >= 128 * 64          # the fused launches' threshold (engine._visual_chunk)
    sd = synth.make_state_dict(1337, gain=float(g['gain']), n_pos=2 + S * 14)
    eng = SynchformerEngine(sd, gpu)
    u8, aud = synth.make_video_u8(B, S, seed), synth.make_spectrogram(B, S, seed)
    vm, am = synth.make_masks_fused_case(B, S, seed)
    eng.capture_blocks = {}
    vf = eng.extract_vfeats(u8.to(gpu), vm.to(gpu)).cpu()
    blocks, eng.capture_blocks = eng.capture_blocks, None
    af = eng.extract_afeats(aud.to(gpu), am.to(gpu)).cpu()
    TOK_V = [0, 1, 2, 197, 1000, 1568]
    ev = _rel_rms(vf.reshape(-1, 768), torch.from_numpy(g['vfeat']).reshape(-1, 768))
    ea = _rel_rms(af.reshape(-1, 768), torch.from_numpy(g['afeat']).reshape(-1, 768))
    e0 = _rel_rms(blocks[0].cpu().view(B * S, 1569, 768)[:, TOK_V], torch.from_numpy(g['vblock0_rows']))
    e11 = _rel_rms(blocks[11].cpu().view(B * S, 1569, 768)[:, TOK_V], torch.from_numpy(g['vblock11_rows']))
    fused = eng.forward(u8.to(gpu), aud.to(gpu), vm.to(gpu), am.to(gpu))
    assert torch.isfinite(fused).all()
    dl = (fused.cpu() - torch.from_numpy(g['logits'])).abs().max().item()
    dref = float(np.abs(g['logits'] - g['logits_nomask']).max())
    print(f'masked, fused launches vs the REAL reference: vfeat relrms {ev:.4f} afeat {ea:.4f} block0 rows {e0:.4f} block11 rows {e11:.4f} logits max {dl:.5f} '
          f'(the masks move the reference logits by {dref:.3f})')
    assert ev < 1.5e-2 and ea < 1.5e-2 and e0 < 1e-2 and e11 < 1.5e-2 and dl < 4e-2 and dref > 0.1       # bars of test_token_masks_match_reference_golden (gain-2 init)
    old = os.environ.get('SF_CLS_FUSION')
    os.environ['SF_CLS_FUSION'] = 'none'
    eng.fuse_time = False
    try:
        unfused = eng.forward(u8.to(gpu), aud.to(gpu), vm.to(gpu), am.to(gpu))
    finally:
        eng.fuse_time = True
        if old is None:
            del os.environ['SF_CLS_FUSION']
        else:
            os.environ['SF_CLS_FUSION'] = old
    nomask = eng.forward(u8.to(gpu), aud.to(gpu))
    d, dn = (fused - unfused).abs().max().item(), (fused - nomask).abs().max().item()
    print(f'masked fused vs un-fused: {d:.5f}; distance to the unmasked logits {dn:.3f}')
    assert d < 2e-2 and dn > 0.05            # (two bf16 schedules of one masked forward; the anchor is the real reference above: fused 0.013, bar 4e-2)
    ones = eng.forward(u8.to(gpu), aud.to(gpu), torch.ones_like(vm).to(gpu), torch.ones_like(am).to(gpu))
    # round 5: a masked forward runs the SAME launches as an unmasked one (sf_qkv_space_attention_masked / sf_qkv_time_attention2_masked: the key flags are the starting
    # values of the score accumulators) - an all-ones mask is bit-equal to no mask on the default schedule ...
    assert torch.equal(ones, nomask)
    # ... and the masked forward agrees with round 4's masked schedule (sf_gemm_bf16 + sf_attention_cls_partial_masked, sf_qkv_time_attention_masked)
    eng.fuse_space = eng.fuse_time2 = False
    try:
        r4 = eng.forward(u8.to(gpu), aud.to(gpu), vm.to(gpu), am.to(gpu))
        assert torch.equal(eng.forward(u8.to(gpu), aud.to(gpu), torch.ones_like(vm).to(gpu), torch.ones_like(am).to(gpu)), eng.forward(u8.to(gpu), aud.to(gpu)))
    finally:
        eng.fuse_space = eng.fuse_time2 = True
    d4 = (fused - r4).abs().max().item()
    print(f'masked forward, round-5 fused launches vs round-4 masked schedule: {d4:.5f}')
    assert d4 < 2e-2            # (gain-2 weights: two bf16 schedules of the same masked forward; the golden bar for this init is 4e-2, test_golden_sync_logits_and_features)


def test_dropin_module_forward_with_masks(gpu):
    """Synchformer.forward(vis, aud, targets, vis_mask=, aud_mask=) on the drop-in module == the engine path, loss included."""
    import synchformer_amd as sa
    from synchformer_amd import synth
    B, S = 1, 2
    m = sa.instantiate_from_config(sa.sync_yaml_model_config(n_pos=2 + S * 14))
    sd = synth.make_state_dict(1337, gain=2.0, n_pos=2 + S * 14)
    m.load_state_dict(sd, strict=True)
    m = m.to(gpu).eval()
    u8, aud = _inputs(B, S)
    vm, am = synth.make_masks(B, S, 1337)
    tgt = synth.make_targets(B, 21, 1337).to(gpu)
    with torch.no_grad():
        loss, logits = m(u8.to(gpu), aud.to(gpu), tgt, vis_mask=vm.to(gpu), aud_mask=am.to(gpu))
        _, plain = m(u8.to(gpu), aud.to(gpu))
    ref = m._engine().forward(u8.to(gpu), aud.to(gpu), vm.to(gpu), am.to(gpu))
    assert torch.equal(logits, ref) and (logits - plain).abs().max() > 0.05
    assert abs(float(loss) - float(torch.nn.functional.cross_entropy(ref, tgt))) < 1e-6
    with pytest.raises(AssertionError, match='for_loop'):
        m(u8.to(gpu), aud.to(gpu), for_loop=True, vis_mask=vm.to(gpu))


def test_logits_only_32_clips_reference_parity(gpu):
    """Acc@1-parity proxy at scale: 32 STRUCTURED clips (synth.make_structured_clips: clips differ in content) at the benchmarked launch
    geometry - two batches of 16 clips, each ONE 224-segment chunk - against the REAL reference's logits (tests/golden/logits_only_32.npz, made by
    tests/golden/make_golden.py logits_only), at the reference-like init ('gain1') and at a trained-scale init ('trained': gain-2 weights, offset head x5, top
    logit ~ 10 as in README.md:79).  Bars: max |dlogit| <= 1e-2 ('gain1', logit std 0.54) / <= 0.75 % of the logit range ('trained'); argmax agreement on >= 90 % of
    the clips with every disagreement a TIE inside the numerical error (the picked class within 2 x max |dlogit| of the reference's top logit: these synthetic clips put
    two or three classes within 0.002-0.02 of each other), and the metric the reference reports - accuracy_1 / accuracy_1_tol1 / accuracy_5 of calc_cls_metrics
    (scripts/train_utils.py:632) - with the reference's argmax as the target.  Measured: 30 / 32 and 31 / 32 agree, accuracy_5 = 1."""
    import zlib
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    from synchformer_amd.postprocess import offset_accuracy
    g = np.load(GOLD / 'logits_only_32.npz')
    n = int(g['n_clips'])
    for c in (0, n - 1):
        u8, aud = synth.make_structured_clip(c, 14, int(g['seed']))
        assert zlib.crc32(u8.numpy().tobytes()) == int(g[f'crc_vis_{c}']) and zlib.crc32(aud.numpy().tobytes()) == int(g[f'crc_aud_{c}']), c
    report = {}
    for variant in ('gain1', 'trained'):
        if variant == 'gain1':
            sd = synth.make_state_dict(int(g['seed']))
        else:
            sd = synth.make_state_dict(int(g['seed']), gain=2.0)
            sd['transformer.off_head.weight'] = sd['transformer.off_head.weight'] * float(g['head_scale'])
        eng = SynchformerEngine(sd, gpu, seg_chunk=224)
        got = []
        for c0 in range(0, n, 16):
            u8, aud = synth.make_structured_clips(c0, min(16, n - c0), 14, int(g['seed']))
            got.append(eng.forward(u8.to(gpu), aud.to(gpu)).cpu())
        got = torch.cat(got)
        ref = torch.from_numpy(g['logits_' + variant])
        rng = float(ref.max() - ref.min())
        bar = 1e-2 if variant == 'gain1' else 7.5e-3 * rng                     # measured: 6.3e-3 (logit std 0.54) / 0.095 = 0.5 % of the 19-wide logit range
        err = float((got - ref).abs().max())
        agree = got.argmax(1) == ref.argmax(1)
        # a disagreement is only acceptable as a TIE inside the numerical error: the class the HIP path picked must be within 2 x err of the reference's top logit
        picked = ref.gather(1, got.argmax(1, keepdim=True)).squeeze(1)
        gap = ref.max(1).values - picked
        acc = offset_accuracy(ref.argmax(1), got, topk=(1, 5))
        report[variant] = dict(err=err, bar=bar, range=rng, agree=int(agree.sum()), worst_gap_of_a_flip=float(gap.max()), **acc)
        assert err <= bar, report
        assert float(gap.max()) <= 2 * err + 1e-6, report
        assert int(agree.sum()) >= int(0.9 * n) and acc['accuracy_5'] == 1.0 and acc['accuracy_1_tol1'] >= acc['accuracy_1'], report
        assert len(set(ref.argmax(1).tolist())) >= (3 if variant == 'trained' else 1), 'the clips must not all land on one class'
        del eng
        torch.cuda.empty_cache()
    print('logits_only parity:', report)


def test_syncability_head_32_clips_reference_parity(gpu):
    """BASELINE configs[4]'s Acc@1-parity number: 32 structured 13-segment clips through the 2-way synchronizability head
    (GlobalTransformerWithSyncabilityHead, sync_model.py:176-190) against the REAL reference's logits (tests/golden/syncability_logits_32.npz, make_golden.py
    syncability_logits) at the reference-like init and at a trained scale (gain-2 weights, sync_head x 5: l1 - l0 spreads over 3.6), for the bf16 engine, the MXFP8 towers
    configs[4] runs on, and the MXFP8 towers with fc2's operands kept in bf16 (engine.mx_bf16 = {'fc2'}).  d = l1 - l0 is what the decision reads.  On these synthetic clips
    every d has one sign, so the plain argmax agrees trivially (asserted: 32 / 32); the informative numbers are max / rms |dd|, the rank correlation of d, and the agreement
    with the decision boundary moved to the MEDIAN of the reference's d (a bias shift of the head), where a flip must be a tie inside 2 x max |dd|.
    Measured (tools/syncability_parity.py, profiles/r06_mxfp8_policy.md):      max |dd|   rms     centred   rank corr
        trained  bf16                                                          0.047     0.025   30 / 32    0.996
        trained  MXFP8 towers                                                  0.59      0.47    23 / 32    0.994   <- a near-constant offset of ~0.45: the order of the clips is kept
        trained  MXFP8, fc2 in bf16                                            0.30      0.20    27 / 32    0.993
    i.e. the MXFP8 towers move the 2-way margin by up to 16 % of its spread - fine for the frozen extractor of a fine-tune whose head is trained on these features (the offset
    is absorbed by the head's bias), NOT interchangeable with the bf16 path under a head trained elsewhere; fc2 (the GELU output operand, K = 3072) carries half of it."""
    import zlib
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    g = np.load(GOLD / 'syncability_logits_32.npz')
    n, seed = int(g['n_clips']), int(g['seed'])
    for c in (0, n - 1):
        u8, aud = synth.make_structured_clip(c, 13, seed)
        assert zlib.crc32(u8.numpy().tobytes()) == int(g[f'crc_vis_{c}']) and zlib.crc32(aud.numpy().tobytes()) == int(g[f'crc_aud_{c}']), c
    bars = {('gain1', 'bf16'): (1e-2, 6e-3), ('gain1', 'mxfp8'): (2.5e-2, 1.6e-2), ('gain1', 'mxfp8+fc2'): (1.5e-2, 8e-3),
            ('trained', 'bf16'): (0.1, 0.05), ('trained', 'mxfp8'): (0.9, 0.7), ('trained', 'mxfp8+fc2'): (0.45, 0.3)}          # (max, rms) of |dd|: ~1.5 x measured
    report = {}
    for variant in ('gain1', 'trained'):
        if variant == 'gain1':
            sd = synth.make_state_dict(seed, n_pos=184, n_out=2, head='sync_head')
        else:
            sd = synth.make_state_dict(seed, gain=2.0, n_pos=184, n_out=2, head='sync_head')
            sd['transformer.sync_head.weight'] = sd['transformer.sync_head.weight'] * float(g['head_scale'])
        ref = torch.from_numpy(g['logits_' + variant])
        r = (ref[:, 1] - ref[:, 0]).double()
        med = r.median()
        for mode in ('bf16', 'mxfp8', 'mxfp8+fc2'):
            eng = SynchformerEngine(sd, gpu, fp8_towers=mode != 'bf16')
            eng.mx_bf16 = frozenset({'fc2'}) if mode == 'mxfp8+fc2' else frozenset()
            got = torch.cat([eng.forward(*(t.to(gpu) for t in synth.make_structured_clips(c0, min(16, n - c0), 13, seed))).cpu() for c0 in range(0, n, 16)])
            d = (got[:, 1] - got[:, 0]).double()
            err = (d - r).abs()
            same = (d > med) == (r > med)
            ties = (~same) & ((r - med).abs() <= 2 * err.max())
            rank = float(np.corrcoef(np.argsort(np.argsort(d.numpy())), np.argsort(np.argsort(r.numpy())))[0, 1])
            rep = report[(variant, mode)] = dict(max_dd=float(err.max()), rms_dd=float(err.pow(2).mean().sqrt()), argmax=int((got.argmax(1) == ref.argmax(1)).sum()),
                                                 centred=int(same.sum()), flips_not_ties=int((~same).sum() - ties.sum()), rank_corr=rank, spread=float(r.max() - r.min()))
            mx, rms = bars[(variant, mode)]
            assert rep['max_dd'] <= mx and rep['rms_dd'] <= rms, (variant, mode, rep)
            assert rep['argmax'] == n and rep['flips_not_ties'] == 0 and rep['rank_corr'] >= 0.97, (variant, mode, rep)
            del eng
            torch.cuda.empty_cache()
    print('syncability parity:', {f'{k[0]}/{k[1]}': {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()} for k, v in report.items()})
    assert report[('trained', 'bf16')]['centred'] >= 28 and report[('trained', 'mxfp8+fc2')]['max_dd'] < report[('trained', 'mxfp8')]['max_dd']


def test_logits_only_32_clips_mxfp8_towers(gpu):
    """BASELINE configs[4] at the scale a user of it cares about: the 32 structured clips of `logits_only_32.npz` through `fp8_towers=True` (the
    extractors' Linears on MXFP8 operands) against the REAL reference's fp32 logits, at the reference-like init AND at the trained logit scale (top logit ~ 10,
    logit range 19).  What MXFP8 costs there, measured (tools/fp8_trained_scale.py; the bf16 engine beside it: 0.3 % / 0.56 % of the range, 31 / 32):
        gain1:   max |dlogit| 0.017 = 0.80 % of the 2.17-wide range, rms 0.0078, argmax 23 / 32 (these clips put two or three classes within 0.01 of each other)
        trained: max |dlogit| 0.70  = 3.7 % of the 19-wide range,   rms 0.27,   argmax 26 / 32, accuracy_1_tol1 0.906, accuracy_5 1.0 - every flip lands on a class
                 the reference itself scores within 0.30 of its top logit (the reference's median top-1 / top-2 margin on these clips is 0.29).
    Stated bars (about 1.4 x the measured values): max <= 5 % of the range, rms <= 2 %, accuracy_5 == 1, accuracy_1_tol1 >= 0.85 at trained scale, argmax agreement >= 65 %
    / 75 %, and no flip further from the reference's top logit than the path's own max error.  A 3.7 % logit error is NOT inside an argmax bar for near-tied classes:
    MXFP8 towers are a throughput mode for a frozen extractor under fine-tuning (the 2-way synchronizability head), not a drop-in for 21-way offset read-out."""
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    from synchformer_amd.postprocess import offset_accuracy
    g = np.load(GOLD / 'logits_only_32.npz')
    n = int(g['n_clips'])
    report = {}
    for variant, agree_bar in (('gain1', 0.65), ('trained', 0.75)):
        sd = synth.make_state_dict(int(g['seed'])) if variant == 'gain1' else synth.make_state_dict(int(g['seed']), gain=2.0)
        if variant == 'trained':
            sd['transformer.off_head.weight'] = sd['transformer.off_head.weight'] * float(g['head_scale'])
        eng = SynchformerEngine(sd, gpu, seg_chunk=224, fp8_towers=True)
        got = []
        for c0 in range(0, n, 16):
            u8, aud = synth.make_structured_clips(c0, min(16, n - c0), 14, int(g['seed']))
            got.append(eng.forward(u8.to(gpu), aud.to(gpu)).cpu())
        got = torch.cat(got)
        ref = torch.from_numpy(g['logits_' + variant])
        rng = float(ref.max() - ref.min())
        err, rms = float((got - ref).abs().max()), float((got - ref).pow(2).mean().sqrt())
        agree = int((got.argmax(1) == ref.argmax(1)).sum())
        gap = float((ref.max(1).values - ref.gather(1, got.argmax(1, keepdim=True)).squeeze(1)).max())
        acc = offset_accuracy(ref.argmax(1), got, topk=(1, 5))
        report[variant] = dict(err=err, err_frac=err / rng, rms_frac=rms / rng, agree=agree, worst_gap_of_a_flip=gap, **acc)
        assert 1e-3 * rng < err <= 0.05 * rng and rms <= 0.02 * rng, report        # (> 0.1 %: the fp8 kernels really ran)
        assert acc['accuracy_5'] == 1.0 and agree >= agree_bar * n and gap <= err + 1e-6, report
        if variant == 'trained':
            assert acc['accuracy_1_tol1'] >= 0.85, report
        del eng
        torch.cuda.empty_cache()
    print('logits_only parity, MXFP8 towers:', report)
