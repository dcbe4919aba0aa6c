"""Per-kernel parity on the GPU: each `sf_*` entry point (called through the C ABI via synchformer_amd.ops)
against a plain fp32 torch evaluation of the same op on the same (bf16-rounded) operands.
Tolerances: outputs are bf16 (rel 2^-8) or fp32 sums of bf16 products; stated per test."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.bfloat16()


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_gemm_identity_asymmetric(gpu, gemm_cfg):
    """A = I against an asymmetric W catches swapped fragment rows/cols (guide rule 16)."""
    from synchformer_amd import ops
    K = 320 if gemm_cfg != 11 else 384      # config 11 walks pairs of 64-deep k-tiles
    a = torch.eye(K)
    w = torch.arange(320 * K, dtype=torch.float32).reshape(320, K) % 251 - 125.0   # exactly representable in bf16
    out = torch.empty(K, 320, device=gpu)
    ops.gemm(_bf(a).to(gpu), _bf(w).to(gpu), None, out)
    torch.testing.assert_close(out.cpu(), w.t().contiguous(), rtol=0, atol=0)


PRODUCT_CFGS = (-1, 0, 4, 7, 11)     # tile configurations of the product library; every other one is a measured-slower alternative of the ablation build (-DSF_ABLATION)


@pytest.fixture(params=[-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], ids=['auto'] + [f'cfg{i}' for i in range(12)])
def gemm_cfg(request, gpu):
    from synchformer_amd import _lib
    lib = _lib.load() if request.param in PRODUCT_CFGS else _lib.load_ablation()
    with _lib.using(lib):
        lib.sf_gemm_force_config(request.param)
        yield request.param
        lib.sf_gemm_force_config(-1)


@pytest.fixture
def ablation(gpu):
    """ops.* launch into lib/ab/libsynchformer_hip_ablation.so for the duration of the test (schedule 2 of sf_gemm_res_ln768, config 10 / 12 of sf_gemm_bf16)."""
    from synchformer_amd import _lib
    with _lib.using(_lib.load_ablation()) as lib:
        yield lib


def test_product_library_refuses_ablation_only_alternatives(gpu):
    """configs 1-3, 5, 6, 8-10, 12 of sf_gemm_bf16 and schedule 2 of sf_gemm_res_ln768 are not in libsynchformer_hip.so."""
    from synchformer_amd import ops, _lib
    lib = _lib.load()
    a, w = torch.zeros(512, 768, device=gpu, dtype=torch.bfloat16), torch.zeros(768, 768, device=gpu, dtype=torch.bfloat16)
    out = torch.empty(512, 768, device=gpu, dtype=torch.bfloat16)
    for cfg in (1, 2, 3, 5, 6, 8, 9, 10, 12):
        lib.sf_gemm_force_config(cfg)
        try:
            with pytest.raises(RuntimeError, match='ablation build'):
                ops.gemm(a, w, None, out)
        finally:
            lib.sf_gemm_force_config(-1)
    lib.sf_gemm_res_ln_force_schedule(2)
    try:
        with pytest.raises(RuntimeError, match='ablation build'):
            ops.gemm_res_ln(a, w, None, torch.zeros(512, 768, device=gpu), torch.ones(768, device=gpu), torch.zeros(768, device=gpu), out, 1e-6)
    finally:
        lib.sf_gemm_res_ln_force_schedule(-1)


@pytest.mark.parametrize('M,N,K', [(300, 768, 768), (9000, 768, 768), (8500, 2304, 768), (128, 2304, 768), (1000, 768, 3072), (257, 3072, 768), (5, 21, 768),
                                   (3, 2, 768), (1568 * 2, 768, 1536), (144, 768, 256)])
def test_gemm_bias(gpu, gemm_cfg, M, N, K):
    from synchformer_amd import ops
    if gemm_cfg in (7, 10, 11) and N % 64:
        pytest.skip('persistent configs serve N % 64 == 0 only')
    a, w, b = _bf(_rand(M, K, seed=1)), _bf(_rand(N, K, seed=2, scale=0.05)), _rand(N, seed=3)
    ref = a.float() @ w.float().t() + b
    out = torch.full((M + 3, N), 7.0, device=gpu)
    ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), out, M=M)
    torch.testing.assert_close(out[:M].cpu(), ref, rtol=1e-4, atol=2e-4)
    assert (out[M:] == 7.0).all(), 'rows beyond M were written'
    outb = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
    ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), outb)
    torch.testing.assert_close(outb.float().cpu(), ref, rtol=1e-2, atol=1e-2)


def test_gemm_gelu_residual_maps(gpu, gemm_cfg):
    from synchformer_amd import ops
    if gemm_cfg in (7, 10, 11):
        pytest.skip('persistent configs serve identity row maps only')
    M, N, K = 400, 768, 768
    a, w, b = _bf(_rand(M, K, seed=4)), _bf(_rand(N, K, seed=5, scale=0.05)), _rand(N, seed=6)
    lin = a.float() @ w.float().t() + b
    # GELU -> bf16
    out = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
    ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), out, gelu=True)
    torch.testing.assert_close(out.float().cpu(), torch.nn.functional.gelu(lin), rtol=1e-2, atol=1e-2)
    # in-place fp32 residual
    x = _rand(M, N, seed=7)
    xd = x.to(gpu)
    ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), xd, residual=xd)
    torch.testing.assert_close(xd.cpu(), lin + x, rtol=1e-4, atol=3e-4)
    # mapped output rows: 4 "sequences" of 100 rows dropped at offset 1 of 101-row sequences, residual from there too
    cmap = ops.rowmap(100, 100, 101, 0, 1, 1)
    y = _rand(4 * 101, N, seed=8)
    yd = y.to(gpu)
    ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), yd, residual=yd, c_map=cmap, r_map=cmap)
    ref = y.clone().reshape(4, 101, N)
    ref[:, 1:] += lin.reshape(4, 100, N)
    torch.testing.assert_close(yd.cpu(), ref.reshape(-1, N), rtol=1e-4, atol=3e-4)


@pytest.mark.parametrize('eps', [1e-12, 1e-6, 1e-5])
def test_layernorm(gpu, eps):
    from synchformer_amd import ops
    rows = 1003
    x = _rand(rows, 768, seed=9, scale=3.0) + 0.5
    g, b = 1 + 0.1 * _rand(768, seed=10), 0.1 * _rand(768, seed=11)
    ref = torch.nn.functional.layer_norm(x, (768,), g, b, eps)
    out = torch.empty(rows, 768, device=gpu)
    ops.layernorm(x.to(gpu), g.to(gpu), b.to(gpu), out, eps)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)
    outb = torch.empty(rows, 768, device=gpu, dtype=torch.bfloat16)
    ops.layernorm(x.to(gpu), g.to(gpu), b.to(gpu), outb, eps)
    torch.testing.assert_close(outb.float().cpu(), ref, rtol=8e-3, atol=8e-3)


def test_layernorm_maps_accumulate(gpu):
    from synchformer_amd import ops
    # drop-CLS + per-frame regroup exactly as the engine uses it: (2 seq x 1569) -> (2*8 x 197) rows 1..196
    x = _rand(2 * 1569, 768, seed=12)
    g, b = 1 + 0.1 * _rand(768, seed=13), 0.1 * _rand(768, seed=14)
    z = torch.zeros(2 * 8 * 197, 768)
    zd = z.to(gpu)
    ops.layernorm(x.to(gpu), g.to(gpu), b.to(gpu), zd, 1e-6, rows=2 * 1568,
                  in_map=ops.rowmap(1568, 1568, 1569, 0, 1, 1), out_map=ops.rowmap(1568, 196, 8 * 197, 197, 1, 1))
    ref = torch.nn.functional.layer_norm(x.reshape(2, 1569, 768)[:, 1:], (768,), g, b, 1e-6).reshape(16, 196, 768)
    got = zd.cpu().reshape(16, 197, 768)
    torch.testing.assert_close(got[:, 1:], ref, rtol=1e-5, atol=1e-5)
    assert (got[:, 0] == 0).all()
    # transposing map (audio aggregator): (bs, fi, ti) -> (bs*6 + ti)*13 + 1 + fi, accumulate onto a table
    xa = _rand(3 * 74, 768, seed=15)
    base = _rand(3 * 6 * 13, 768, seed=16)
    bd = base.to(gpu)
    ops.layernorm(xa.to(gpu), g.to(gpu), b.to(gpu), bd, 1e-12, rows=3 * 72, in_map=ops.rowmap(72, 72, 74, 0, 1, 2),
                  out_map=ops.rowmap(72, 6, 78, 1, 13, 1), accumulate=True)
    ln = torch.nn.functional.layer_norm(xa.reshape(3, 74, 768)[:, 2:], (768,), g, b, 1e-12).reshape(3, 12, 6, 768)
    ref = base.clone().reshape(3, 6, 13, 768)
    ref[:, :, 1:] += ln.permute(0, 2, 1, 3)
    torch.testing.assert_close(bd.cpu().reshape(3, 6, 13, 768), ref, rtol=1e-5, atol=1e-5)


def test_broadcast_gather(gpu):
    from synchformer_amd import ops
    table = _rand(5, 768, seed=17)
    dst = torch.zeros(3 * 9, 768, device=gpu)
    ops.broadcast_rows(dst, table.to(gpu), n_seq=3, dst_seq_rows=9)
    got = dst.cpu().reshape(3, 9, 768)
    assert (got[:, :5] == table).all() and (got[:, 5:] == 0).all()
    x = _rand(4 * 197, 768, seed=18)
    out = torch.empty(4, 768, device=gpu, dtype=torch.bfloat16)
    ops.gather_rows(x.to(gpu), out, 4, in_map=ops.rowmap(1, 1, 197, 0, 0, 0))
    torch.testing.assert_close(out.float().cpu(), x.reshape(4, 197, 768)[:, 0].bfloat16().float(), rtol=0, atol=0)


@pytest.mark.parametrize('dtype', [torch.uint8, torch.float16, torch.bfloat16, torch.float32])
def test_im2col_video(gpu, dtype):
    from synchformer_amd import ops, synth
    from oracle import synchformer_cpu as O
    u8 = synth.make_video_u8(1, 2, seed=5)[0]                 # (2, 16, 3, 224, 224)
    if dtype == torch.uint8:
        vid, ref_in = u8, O.rgb_frontend(u8)
    else:
        vid = O.rgb_frontend(u8).to(dtype)
        ref_in = vid.float()
    out = torch.empty(2 * 1568, 1536, device=gpu, dtype=torch.bfloat16)
    ops.im2col_video(vid.contiguous().to(gpu), out)
    x = ref_in.permute(0, 2, 1, 3, 4)                          # (N, C, T, H, W) as patch_embed_3d expects
    ref = x.reshape(2, 3, 8, 2, 14, 16, 14, 16).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(2 * 1568, 1536)
    torch.testing.assert_close(out.float().cpu(), ref.bfloat16().float(), rtol=0, atol=0)


def test_im2col_spec(gpu):
    from synchformer_amd import ops
    spec = _rand(3, 128, 66, seed=19)
    out = torch.empty(3 * 72, 256, device=gpu, dtype=torch.bfloat16)
    ops.im2col_spec(spec.to(gpu), out)
    ref = spec.unfold(1, 16, 10).unfold(2, 16, 10).reshape(3 * 72, 256)
    torch.testing.assert_close(out.float().cpu(), ref.bfloat16().float(), rtol=0, atol=0)


def _attn_ref(q, k, v, scale):
    s = (q @ k.transpose(-1, -2)) * scale
    return torch.softmax(s, -1) @ v


def _divided_ref(qkv, mode, heads=12, frames=8, n=196):
    """fp32 divided attention on packed (N, L, 3*768) projections (vit_helper.py:100-158 semantics), patches only."""
    N, L, _ = qkv.shape
    d = 64
    t = qkv.reshape(N, L, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = t[0], t[1], t[2]
    def grp(x):
        x = x[:, :, 1:].reshape(N, heads, frames, n, d)
        return x.transpose(2, 3) if mode == 'time' else x
    qg, kg, vg = grp(q), grp(k), grp(v)
    G = qg.shape[2]
    kc = k[:, :, :1].unsqueeze(2).expand(N, heads, G, 1, d)
    vc = v[:, :, :1].unsqueeze(2).expand(N, heads, G, 1, d)
    og = _attn_ref(qg, torch.cat([kc, kg], 3), torch.cat([vc, vg], 3), d ** -0.5)
    if mode == 'time':
        og = og.transpose(2, 3)
    cls = _attn_ref(q[:, :, :1], k, v, d ** -0.5)
    out = torch.cat([cls, og.reshape(N, heads, frames * n, d)], 2)
    return out.transpose(1, 2).reshape(N, L, heads * d)


@pytest.mark.parametrize('mode', ['time', 'space'])
def test_divided_attention(gpu, mode):
    from synchformer_amd import ops
    N, L = 2, 1569
    qkv = _bf(_rand(N * L, 2304, seed=20, scale=1.5))
    ref = _divided_ref(qkv.float().reshape(N, L, 2304), mode)
    qd = qkv.to(gpu)
    out = torch.zeros(N * L, 768, device=gpu, dtype=torch.bfloat16)
    q, k, v = qd[:, :768], qd[:, 768:1536], qd[:, 1536:]
    if mode == 'time':
        ops.attention(q, k, v, out, n_seq=N, seq_rows=L, n_groups=196, row0=1, group_stride=1, tok_stride=196, n_tok=8,
                      cls_row=0, heads=12, head_dim=64, scale=0.125)
    else:
        ops.attention(q, k, v, out, n_seq=N, seq_rows=L, n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196,
                      cls_row=0, heads=12, head_dim=64, scale=0.125)
    ops.attention_cls(q, k, v, out, n_seq=N, q_seq_rows=L, q_row=0, kv_seq_rows=L, kv_row0=0, n_keys=L, out_seq_rows=L,
                      out_row=0, heads=12, head_dim=64, scale=0.125)
    torch.testing.assert_close(out.float().cpu().reshape(N, L, 768), ref, rtol=2e-2, atol=2e-2)
    # fused variant: CLS-query partials from the same kernel + combine must reproduce row 0 and leave the patches unchanged
    out2 = torch.zeros_like(out)
    groups = 196 if mode == 'time' else 8
    part = torch.empty(N * 12 * groups * 66, device=gpu)
    kw = dict(n_groups=196, row0=1, group_stride=1, tok_stride=196, n_tok=8) if mode == 'time' else \
        dict(n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196)
    ops.attention_cls_partial(q, k, v, out2, part, n_seq=N, seq_rows=L, cls_row=0, heads=12, head_dim=64, scale=0.125, **kw)
    ops.attention_cls_combine(part, out2, n_part=groups, n_seq=N, out_seq_rows=L, out_row=0, heads=12)
    o1, o2 = out.float().cpu().reshape(N, L, 768), out2.float().cpu().reshape(N, L, 768)
    assert torch.equal(o1[:, 1:], o2[:, 1:])
    torch.testing.assert_close(o2[:, 0], ref[:, 0], rtol=2e-2, atol=2e-2)


def test_space_attention_mxfp8_output(gpu):
    """sf_attention_cls_partial_mx + sf_attention_cls_combine_mx == the bf16 kernels followed by sf_quantize_mxfp8, byte for byte (bytes and scale planes; rows the
    kernels do not own keep their fill)."""
    from synchformer_amd import ops
    N, L = 3, 1569
    qkv = _bf(_rand(N * L, 2304, seed=23, scale=1.5)).to(gpu)
    qkv[5:9] *= 40.0                                                        # a few large rows: other scale exponents
    qkv[:, 1536 + 64:1536 + 96] = 0                                         # one all-zero value block per row: amax = 0 -> scale byte 1
    q, k, v = qkv[:, :768], qkv[:, 768:1536], qkv[:, 1536:]
    kw = dict(n_seq=N, seq_rows=L, n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196, cls_row=0, heads=12)
    out = torch.zeros(N * L, 768, device=gpu, dtype=torch.bfloat16)
    part = torch.empty(N * 12 * 8 * 66, device=gpu)
    ops.attention_cls_partial(q, k, v, out, part, head_dim=64, scale=0.125, **kw)
    ops.attention_cls_combine(part, out, n_part=8, n_seq=N, out_seq_rows=L, out_row=0, heads=12)
    q0, s0 = torch.empty(N * L, 768, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(N * L, 768, gpu)
    ops.quantize_mxfp8(out, q0, s0)
    q1, s1 = torch.full((N * L + 3, 768), 7, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(N * L, 768, gpu)
    part1 = torch.empty_like(part)
    ops.attention_cls_partial_mx(q, k, v, q1, s1, part1, scale=0.125, **kw)
    assert (q1.view(-1, 768)[:N * L].view(N, L, 768)[:, 0] == 7).all()       # the CLS rows belong to the combine kernel
    ops.attention_cls_combine_mx(part1, q1, s1, n_part=8, n_seq=N, out_seq_rows=L, out_row=0, heads=12)
    assert torch.equal(part, part1)
    assert torch.equal(q0, q1[:N * L]) and (q1[N * L:] == 7).all()
    assert torch.equal(s0[:, :N * L], s1[:, :N * L]) and (s1[:, N * L:] == 0).all()


@pytest.mark.parametrize('L,heads,d', [(74, 12, 64), (198, 8, 96), (184, 8, 96), (13, 12, 64), (197, 12, 64), (17, 12, 64)])
def test_full_attention(gpu, L, heads, d):
    from synchformer_amd import ops
    N, Dm = 3, heads * d
    qkv = _bf(_rand(N * L, 3 * Dm, seed=21, scale=1.2))
    t = qkv.float().reshape(N, L, 3, heads, d).permute(2, 0, 3, 1, 4)
    ref = _attn_ref(t[0], t[1], t[2], 1 / math.sqrt(d)).transpose(1, 2).reshape(N, L, Dm)
    qd = qkv.to(gpu)
    out = torch.zeros(N * L, Dm, device=gpu, dtype=torch.bfloat16)
    ops.attention(qd[:, :Dm], qd[:, Dm:2 * Dm], qd[:, 2 * Dm:], out, n_seq=N, seq_rows=L, n_groups=1, row0=0,
                  group_stride=0, tok_stride=1, n_tok=L, cls_row=-1, heads=heads, head_dim=d, scale=1 / math.sqrt(d))
    torch.testing.assert_close(out.float().cpu().reshape(N, L, Dm), ref, rtol=2e-2, atol=2e-2)
    if d == 64:   # the CLS-only kernel must agree with row 0 of the full attention
        oc = torch.zeros(N, Dm, device=gpu, dtype=torch.bfloat16)
        ops.attention_cls(qd[:, :Dm], qd[:, Dm:2 * Dm], qd[:, 2 * Dm:], oc, n_seq=N, q_seq_rows=L, q_row=0, kv_seq_rows=L,
                          kv_row0=0, n_keys=L, out_seq_rows=1, out_row=0, heads=heads, head_dim=d, scale=1 / math.sqrt(d))
        torch.testing.assert_close(oc.float().cpu(), ref[:, 0], rtol=2e-2, atol=2e-2)


def test_attention_sharp_softmax(gpu):
    """Large-magnitude scores (trained-model regime): exercises max-subtraction and the online-softmax merges."""
    from synchformer_amd import ops
    N, L, heads, d = 2, 197, 12, 64
    qkv = _bf(_rand(N * L, 2304, seed=22, scale=6.0))
    t = qkv.float().reshape(N, L, 3, heads, d).permute(2, 0, 3, 1, 4)
    ref = _attn_ref(t[0], t[1], t[2], 0.125).transpose(1, 2).reshape(N, L, 768)
    qd = qkv.to(gpu)
    out = torch.zeros(N * L, 768, device=gpu, dtype=torch.bfloat16)
    ops.attention(qd[:, :768], qd[:, 768:1536], qd[:, 1536:], out, n_seq=N, seq_rows=L, n_groups=1, row0=0,
                  group_stride=0, tok_stride=1, n_tok=L, cls_row=-1, heads=heads, head_dim=d, scale=0.125)
    torch.testing.assert_close(out.float().cpu().reshape(N, L, 768), ref, rtol=3e-2, atol=6e-2)
    oc = torch.zeros(N, 768, device=gpu, dtype=torch.bfloat16)
    ops.attention_cls(qd[:, :768], qd[:, 768:1536], qd[:, 1536:], oc, n_seq=N, q_seq_rows=L, q_row=0, kv_seq_rows=L,
                      kv_row0=0, n_keys=L, out_seq_rows=1, out_row=0, heads=heads, head_dim=d, scale=0.125)
    torch.testing.assert_close(oc.float().cpu(), ref[:, 0], rtol=3e-2, atol=6e-2)


def test_torch_ops_registered(gpu):
    """The launchers are also dispatcher-visible custom ops (torch.ops.synchformer.*), as a PyTorch-ROCm extension would offer."""
    import synchformer_amd  # noqa: F401  (registers the ops)
    a, w, b = _bf(_rand(130, 768, seed=31)), _bf(_rand(768, 768, seed=32, scale=0.05)), _rand(768, seed=33)
    out = torch.empty(130, 768, device=gpu)
    torch.ops.synchformer.gemm_bf16(a.to(gpu), w.to(gpu), b.to(gpu), out, None, False)
    torch.testing.assert_close(out.cpu(), a.float() @ w.float().t() + b, rtol=1e-4, atol=2e-4)
    x = _rand(10, 768, seed=34)
    y = torch.empty(10, 768, device=gpu)
    torch.ops.synchformer.layernorm768(x.to(gpu), torch.ones(768, device=gpu), torch.zeros(768, device=gpu), y, 1e-5)
    torch.testing.assert_close(y.cpu(), torch.nn.functional.layer_norm(x, (768,)), rtol=1e-5, atol=1e-5)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.synchformer.layernorm768(x, torch.ones(768), torch.zeros(768), torch.empty(10, 768), 1e-5)   # CPU: no kernel


@pytest.mark.parametrize('M,K', [(1, 768), (127, 768), (128, 768), (1000, 768), (4500, 3072), (40000, 768), (33000, 3072)])
def test_gemm_res_ln(gpu, M, K):
    """sf_gemm_res_ln768 (full-row GEMM + bias + fp32 residual + the next LayerNorm) against fp32 torch on the same bf16 operands and
    against the un-fused HIP pair (sf_gemm_bf16 + sf_layernorm768).  X: fp32 sums of bf16 products (rtol 1e-4); Y: bf16 (2^-8 relative)
    of a LayerNorm output of magnitude <= ~4.  M = 40000 / 33000 give 313 / 258 row tiles: more than the 256 persistent workgroups, so the
    tile loop with its next-tile prefetch, and the ragged last tile, are exercised; A rows beyond M are poisoned."""
    from synchformer_amd import ops
    a, w = _bf(_rand(M + 5, K, seed=1)), _bf(_rand(768, K, seed=2, scale=0.05))
    a[M:] = float('nan')
    b, r = _rand(768, seed=3), _rand(M, 768, seed=4, scale=2.0) + 0.5
    gam, bet = 1.0 + 0.1 * _rand(768, seed=5), 0.1 * _rand(768, seed=6)
    eps = 1e-6
    x_ref = a[:M].float() @ w.float().t() + b + r
    y_ref = torch.nn.functional.layer_norm(x_ref, (768,), gam, bet, eps)
    x = torch.full((M + 3, 768), 7.0, device=gpu)
    x[:M] = r.to(gpu)
    y = torch.full((M + 3, 768), 3.0, device=gpu, dtype=torch.bfloat16)
    ops.gemm_res_ln(a.to(gpu), w.to(gpu), b.to(gpu), x, gam.to(gpu), bet.to(gpu), y, eps, M=M)
    torch.testing.assert_close(x[:M].cpu(), x_ref, rtol=1e-4, atol=3e-4)
    torch.testing.assert_close(y[:M].float().cpu(), y_ref, rtol=1e-2, atol=1e-2)
    assert (x[M:] == 7.0).all() and (y[M:] == 3.0).all(), 'rows beyond M were written'
    # the k-step-major weight layout (what the engine passes): same products in the same order -> bit-identical
    xk = torch.full((M + 3, 768), 7.0, device=gpu)
    xk[:M] = r.to(gpu)
    yk = torch.full((M + 3, 768), 3.0, device=gpu, dtype=torch.bfloat16)
    ops.gemm_res_ln(a.to(gpu), ops.kmajor_weight(w.to(gpu)), b.to(gpu), xk, gam.to(gpu), bet.to(gpu), yk, eps, M=M)
    assert torch.equal(xk, x) and torch.equal(yk, y)
    # the un-fused pair on the same operands: same fp32 X up to summation order, same bf16 Y up to one rounding
    x2 = torch.empty(M, 768, device=gpu)
    x2.copy_(r)
    y2 = torch.empty(M, 768, device=gpu, dtype=torch.bfloat16)
    ops.gemm(a.to(gpu), w.to(gpu), b.to(gpu), x2, M=M, residual=x2)
    ops.layernorm(x2, gam.to(gpu), bet.to(gpu), y2, eps)
    torch.testing.assert_close(x[:M], x2, rtol=1e-5, atol=2e-5)
    # (|dx| <= 2e-5 in the fp32 stream moves the normalised value by about as much; near a bf16 rounding boundary that flips one ulp = 2^-8 |y|)
    assert ((y[:M].float() - y2.float()).abs() <= 2.0 ** -7 * y2.float().abs() + 1e-4).all()


@pytest.mark.parametrize('M,K', [(128 * 300 + 37, 768), (128 * 290 + 1, 3072), (200, 768)])
def test_gemm_res_ln_schedules_bitwise(gpu, M, K):
    """The quadrant-phased main loop of sf_gemm_res_ln768 (round 3) sums every accumulator in the same k order as round 2's loop: X and Y must be
    bit-identical, on every repetition (the repetitions screen for LDS-DMA / ds_read ordering races), for both weight layouts."""
    from synchformer_amd import ops, _lib
    lib = _lib.load()
    a, w = _bf(_rand(M, K, seed=61)).to(gpu), _bf(_rand(768, K, seed=62, scale=0.05)).to(gpu)
    b, r = _rand(768, seed=63).to(gpu), (_rand(M, 768, seed=64, scale=2.0) + 0.5).to(gpu)
    gam, bet = (1.0 + 0.1 * _rand(768, seed=65)).to(gpu), (0.1 * _rand(768, seed=66)).to(gpu)

    def run(sched, wt):
        lib.sf_gemm_res_ln_force_schedule(sched)
        try:
            x, y = r.clone(), torch.empty(M, 768, device=gpu, dtype=torch.bfloat16)
            ops.gemm_res_ln(a, wt, b, x, gam, bet, y, 1e-6)
        finally:
            lib.sf_gemm_res_ln_force_schedule(-1)
        return x, y
    x0, y0 = run(0, w)
    wk = ops.kmajor_weight(w)
    for rep in range(5):
        for wt in (w, wk):
            x1, y1 = run(1, wt)
            assert torch.equal(x1, x0) and torch.equal(y1, y0), f'repetition {rep}'


@pytest.mark.parametrize('M,K', [(192 * 300 + 37, 768), (192 * 270 + 191, 3072), (192 * 256 + 1, 768), (100, 768), (192, 3072)])
def test_gemm_res_ln_schedule2(gpu, ablation, M, K):
    """Schedule 2 of sf_gemm_res_ln768 (round 4: 192-row tiles, two 384-column passes, the residual and the chunk exchange through LDS) against
    fp32 torch and against schedule 1 on the same operands: X to fp32 summation order, Y to one bf16 rounding; rows beyond M untouched; A rows beyond
    M poisoned (the ragged last tile clamps its reads); five repetitions bit-identical (a screen for LDS-DMA / ds_read ordering races)."""
    from synchformer_amd import ops, _lib
    lib = _lib.load()
    a, w = _bf(_rand(M + 5, K, seed=71)), _bf(_rand(768, K, seed=72, scale=0.05))
    a[M:] = float('nan')
    b, r = _rand(768, seed=73), _rand(M, 768, seed=74, scale=2.0) + 0.5
    gam, bet = 1.0 + 0.1 * _rand(768, seed=75), 0.1 * _rand(768, seed=76)
    x_ref = a[:M].float() @ w.float().t() + b + r
    y_ref = torch.nn.functional.layer_norm(x_ref, (768,), gam, bet, 1e-6)
    ag, wg, bg, gg, btg = a.to(gpu), w.to(gpu), b.to(gpu), gam.to(gpu), bet.to(gpu)

    def run(sched):
        lib.sf_gemm_res_ln_force_schedule(sched)
        try:
            x = torch.full((M + 3, 768), 7.0, device=gpu)
            x[:M] = r.to(gpu)
            y = torch.full((M + 3, 768), 3.0, device=gpu, dtype=torch.bfloat16)
            ops.gemm_res_ln(ag, wg, bg, x, gg, btg, y, 1e-6, M=M)
        finally:
            lib.sf_gemm_res_ln_force_schedule(-1)
        return x, y
    x, y = run(2)
    torch.testing.assert_close(x[:M].cpu(), x_ref, rtol=1e-4, atol=3e-4)
    torch.testing.assert_close(y[:M].float().cpu(), y_ref, rtol=1e-2, atol=1e-2)
    assert (x[M:] == 7.0).all() and (y[M:] == 3.0).all(), 'rows beyond M were written'
    x1, y1 = run(1)
    torch.testing.assert_close(x[:M], x1[:M], rtol=1e-5, atol=2e-5)
    assert ((y[:M].float() - y1[:M].float()).abs() <= 2.0 ** -7 * y1[:M].float().abs() + 1e-4).all()
    for rep in range(5):
        xr, yr = run(2)
        assert torch.equal(xr, x) and torch.equal(yr, y), f'repetition {rep}'
    # in place: Y aliasing A (K = 768 only: the engine's XN buffer) -- each tile reads all of its A rows before its first Y store
    if K == 768:
        lib.sf_gemm_res_ln_force_schedule(2)
        try:
            buf, x2 = ag[:M].clone(), r.to(gpu)
            ops.gemm_res_ln(buf, wg, bg, x2, gg, btg, buf, 1e-6)
        finally:
            lib.sf_gemm_res_ln_force_schedule(-1)
        assert torch.equal(x2, x[:M]) and torch.equal(buf, y[:M])


def test_gemm_res_ln_in_place_operand(gpu):
    """Y aliasing A (the engine's XN buffer holds the attention output going in and the normalised rows coming out) and X aliasing R."""
    from synchformer_amd import ops
    M, K = 36000, 768
    a, w = _bf(_rand(M, K, seed=11)).to(gpu), _bf(_rand(768, K, seed=12, scale=0.05)).to(gpu)
    b, r = _rand(768, seed=13).to(gpu), _rand(M, 768, seed=14).to(gpu)
    gam, bet = (1.0 + 0.1 * _rand(768, seed=15)).to(gpu), (0.1 * _rand(768, seed=16)).to(gpu)
    x0, y0 = r.clone(), torch.empty(M, 768, device=gpu, dtype=torch.bfloat16)
    ops.gemm_res_ln(a, w, b, x0, gam, bet, y0, 1e-6)
    x1, buf = r.clone(), a.clone()
    ops.gemm_res_ln(buf, w, b, x1, gam, bet, buf, 1e-6)
    assert torch.equal(x0, x1) and torch.equal(y0, buf)


def _mx_dequant(q, s):
    """OCP MXFP8 -> fp32: e4m3 bytes times 2^(scale - 127), one scale per 32 consecutive k."""
    x = q.view(torch.float8_e4m3fn).float()
    rows = q.shape[0]
    sc = s[:, :rows].permute(1, 0, 2).reshape(rows, -1)                                 # stage-major (K/128, rows, 4) -> (rows, K/32)
    return x * torch.exp2(sc.float() - 127.0).repeat_interleave(32, dim=1)


def _mx_quant_ref(x):
    """OCP Microscaling MXFP8 quantisation of a bf16 matrix, restated in torch: per 32-block scale 2^(floor(log2 amax) - 8), elements rounded to
    nearest-even e4m3 after saturation to +-448."""
    xf = x.float()
    R, K = xf.shape
    blk = xf.view(R, K // 32, 32)
    amax = blk.abs().amax(-1)
    e = torch.floor(torch.log2(torch.clamp(amax, min=2.0 ** -126))) - 8
    byte = torch.clamp(e + 127, 1, 254)
    q = torch.clamp(blk * torch.exp2(127 - byte).unsqueeze(-1), -448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(R, K).view(torch.uint8), byte.to(torch.uint8)


@pytest.mark.parametrize('rows,K', [(5, 768), (300, 3072), (1000, 128)])
def test_quantize_mxfp8(gpu, rows, K):
    """sf_quantize_mxfp8 against the torch restatement of the OCP MX rule: scale bytes and element bytes are integers - exact."""
    from synchformer_amd import ops
    x = _bf(_rand(rows, K, seed=rows) * torch.exp(_rand(rows, 1, seed=K) * 2))      # rows of very different magnitude
    x[0, :32] = 0.0                                                                    # an all-zero block
    q = torch.empty(rows, K, device=gpu, dtype=torch.uint8)
    s = torch.full((K // 128, rows + 2, 4), 255, device=gpu, dtype=torch.uint8)
    ops.quantize_mxfp8(x.to(gpu), q, s)
    q_ref, s_ref = _mx_quant_ref(x)
    s_rows = s.cpu()[:, :rows].permute(1, 0, 2).reshape(rows, -1)
    assert torch.equal(s_rows[1:], s_ref[1:]) and torch.equal(q.cpu(), q_ref) and (s.cpu()[:, rows:] == 255).all()
    # elements within a factor 16 of their block maximum: one e4m3 rounding (2^-4 relative), or up to 2^-3 where the block maximum saturates at 448
    err = (_mx_dequant(q.cpu(), s.cpu()) - x.float()).abs() / x.float().abs().clamp(min=1e-20)
    assert err[x.float().abs() > x.float().abs().view(rows, -1, 32).amax(-1).repeat_interleave(32, 1) / 16].max() < 2.0 ** -3 + 1e-3


@pytest.mark.parametrize('M,N,K', [(256, 256, 128), (300, 768, 768), (9000, 2304, 768), (4100, 768, 3072), (70000, 3072, 768)])
def test_gemm_mxfp8(gpu, M, N, K):
    """sf_gemm_mxfp8 against an fp32 matmul of the DEQUANTISED operands (the kernel's exact contract: products of e4m3 values, scaled by the two
    block scales, accumulated in fp32), with operands whose block scales differ widely along K and across rows so that a wrong scale byte, a
    swapped k half or a transposed fragment cannot hide.  Then bias + exact-erf GELU (bf16 out) and bias + fp32 residual epilogues."""
    from synchformer_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-6, 6, (M, K // 32), generator=g).float()).repeat_interleave(32, 1))
    w = _bf(torch.randn(N, K, generator=g) * 0.05 * torch.exp2(torch.randint(-4, 4, (N, K // 32), generator=g).float()).repeat_interleave(32, 1))
    b = _rand(N, seed=3)
    aq, asc = torch.empty(M, K, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(M, K, gpu)
    wq, wsc = torch.empty(N, K, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(N, K, gpu)
    ops.quantize_mxfp8(a.to(gpu), aq, asc)
    ops.quantize_mxfp8(w.to(gpu), wq, wsc)
    ref = (_mx_dequant(aq, asc).double() @ _mx_dequant(wq, wsc).double().t()).float().cpu() + b
    out = torch.full((M + 3, N), 7.0, device=gpu)
    ops.gemm_mxfp8(aq, asc, wq, wsc, b.to(gpu), out, M=M)
    scale = ref.abs().max().item()
    assert (out[:M].cpu() - ref).abs().max().item() < 1e-4 * scale, ((out[:M].cpu() - ref).abs().max().item(), scale)   # fp32 accumulation inside the MFMA: measured 3e-5
    assert (out[M:] == 7.0).all(), 'rows beyond M were written'
    # quantisation error against the un-quantised bf16 product: the property the FT path relies on (two e4m3 roundings per product, relative
    # Frobenius error ~ 2^-4.5 = 4.3 % measured on Gaussian operands)
    exact = a.float() @ w.float().t() + b
    rel = ((out[:M].cpu() - exact).norm() / exact.norm()).item()
    assert rel < 6e-2, rel
    r = _rand(M, N, seed=5)
    res = r.to(gpu).clone()
    ops.gemm_mxfp8(aq, asc, wq, wsc, b.to(gpu), res, residual=res)
    assert (res.cpu() - (ref + r)).abs().max().item() < 1e-4 * scale + 1e-5
    hb = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
    ops.gemm_mxfp8(aq, asc, wq, wsc, b.to(gpu), hb, gelu=True)
    torch.testing.assert_close(hb.float().cpu(), torch.nn.functional.gelu(ref), rtol=1e-2, atol=1e-2 * max(1.0, scale / 8))


@pytest.mark.parametrize('M,N,K', [(256 * 70 + 37, 2304, 768), (256 * 67 + 255, 768, 3072), (300, 768, 256)])
def test_gemm_mxfp8_schedules_bitwise(gpu, M, N, K):
    """The quadrant-phased MXFP8 kernel (round 3) against round 2's loop: same products in the same order, so fp32 / bf16 + GELU / MXFP8 outputs (bytes and
    scale planes) and the fp32-residual epilogue must be bit-identical on every repetition (the race screen for the LDS-DMA stream incl. the scale pieces)."""
    from synchformer_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K + 1)
    a = _bf(torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-6, 6, (M, K // 32), generator=g).float()).repeat_interleave(32, 1)).to(gpu)
    w = _bf(torch.randn(N, K, generator=g) * 0.05 * torch.exp2(torch.randint(-4, 4, (N, K // 32), generator=g).float()).repeat_interleave(32, 1)).to(gpu)
    b, r = _rand(N, seed=3).to(gpu), _rand(M, N, seed=5).to(gpu)
    aq, asc = torch.empty(M, K, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(M, K, gpu)
    wq, wsc = torch.empty(N, K, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(N, K, gpu)
    ops.quantize_mxfp8(a, aq, asc)
    ops.quantize_mxfp8(w, wq, wsc)

    def run(sched):
        lib.sf_gemm_mx_force_schedule(sched)
        try:
            o32 = torch.empty(M, N, device=gpu)
            ops.gemm_mxfp8(aq, asc, wq, wsc, b, o32)
            ores = r.clone()
            ops.gemm_mxfp8(aq, asc, wq, wsc, b, ores, residual=ores)
            ob = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
            ops.gemm_mxfp8(aq, asc, wq, wsc, b, ob, gelu=True)
            outs = [o32, ores, ob]
            if N % 128 == 0:
                oq, osc = torch.zeros(M, N, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(M, N, gpu)
                osc.zero_()
                ops.gemm_mxfp8(aq, asc, wq, wsc, b, oq, gelu=True, out_scales=osc)
                outs += [oq, osc]
        finally:
            lib.sf_gemm_mx_force_schedule(-1)
        return outs
    ref = run(0)
    for rep in range(4):
        got = run(1)
        for i, (x, y) in enumerate(zip(got, ref)):
            assert torch.equal(x, y), f'repetition {rep}, output {i}'


def test_layernorm_mxfp8_and_fp8_epilogue(gpu):
    """sf_layernorm768_mxfp8 == sf_layernorm768 (bf16) followed by sf_quantize_mxfp8, byte for byte; and the MX GEMM's MXFP8-output epilogue
    (fc1 + GELU -> the operand of fc2) == its bf16-output epilogue followed by sf_quantize_mxfp8."""
    from synchformer_amd import ops
    rows = 1000
    x = (_rand(rows, 768, seed=1) * 2 + 0.3).to(gpu)
    gam, bet = (1 + 0.1 * _rand(768, seed=2)).to(gpu), (0.1 * _rand(768, seed=3)).to(gpu)
    yb = torch.empty(rows, 768, device=gpu, dtype=torch.bfloat16)
    ops.layernorm(x, gam, bet, yb, 1e-6)
    q0, s0 = torch.empty(rows, 768, device=gpu, dtype=torch.uint8), torch.zeros(6, rows, 4, device=gpu, dtype=torch.uint8)
    ops.quantize_mxfp8(yb, q0, s0)
    q1, s1 = torch.empty_like(q0), torch.zeros_like(s0)
    ops.layernorm_mxfp8(x, gam, bet, q1, s1, 1e-6)
    assert torch.equal(q0, q1) and torch.equal(s0, s1)
    M, N, K = 2600, 3072, 768
    w = _bf(_rand(N, K, seed=4, scale=0.05)).to(gpu)
    wq, ws = torch.empty(N, K, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(N, K, gpu)
    ops.quantize_mxfp8(w, wq, ws)
    a = _bf(_rand(M, K, seed=5)).to(gpu)
    aq, asc = torch.empty(M, K, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(M, K, gpu)
    ops.quantize_mxfp8(a, aq, asc)
    b = _rand(N, seed=6).to(gpu)
    hb = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
    ops.gemm_mxfp8(aq, asc, wq, ws, b, hb, gelu=True)
    h0, hs0 = torch.empty(M, N, device=gpu, dtype=torch.uint8), torch.zeros(N // 128, M, 4, device=gpu, dtype=torch.uint8)
    ops.quantize_mxfp8(hb, h0, hs0)
    h1, hs1 = torch.full((M + 2, N), 9, device=gpu, dtype=torch.uint8), torch.zeros(N // 128, M + 2, 4, device=gpu, dtype=torch.uint8)
    ops.gemm_mxfp8(aq, asc, wq, ws, b, h1, gelu=True, out_scales=hs1, M=M)
    assert torch.equal(h0, h1[:M]) and torch.equal(hs0, hs1[:, :M]) and (h1[M:] == 9).all() and (hs1[:, M:] == 0).all()


@pytest.mark.parametrize('M,K', [(100, 128), (128 * 5 + 17, 768), (128 * 300 + 77, 768), (128 * 270 + 1, 3072)])
def test_gemm_mx_res_ln(gpu, M, K):
    """sf_gemm_mx_res_ln768 against the pair it replaces - sf_gemm_mxfp8 with the residual epilogue, then sf_layernorm768_mxfp8 - and against an fp64
    reference on the DEQUANTISED operands.  The fused kernel rotates its k-loop per workgroup, so X agrees to fp32 summation-order noise, and the MXFP8
    bytes of Y agree except where that noise crosses a bf16 / e4m3 rounding boundary (a bounded fraction; the dequantised values then differ by one step).
    Repetitions must be bit-identical (race screen); rows beyond M stay untouched; (Y, sY) may be the (A, sA) buffers when K == 768."""
    from synchformer_amd import ops
    g = torch.Generator().manual_seed(M + K)
    a = _bf(torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-3, 3, (M, K // 32), generator=g).float()).repeat_interleave(32, 1))
    w = _bf(torch.randn(768, K, generator=g) * 0.03 * torch.exp2(torch.randint(-2, 2, (768, K // 32), generator=g).float()).repeat_interleave(32, 1))
    b, r = _rand(768, seed=3).to(gpu), (_rand(M, 768, seed=5) * 2 + 0.2).to(gpu)
    gam, bet = (1 + 0.1 * _rand(768, seed=6)).to(gpu), (0.1 * _rand(768, seed=7)).to(gpu)
    aq, asc = torch.empty(M, K, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(M, K, gpu)
    wq, wsc = torch.empty(768, K, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(768, K, gpu)
    ops.quantize_mxfp8(a.to(gpu), aq, asc)
    ops.quantize_mxfp8(w.to(gpu), wq, wsc)
    # the pair
    x0 = r.clone()
    ops.gemm_mxfp8(aq, asc, wq, wsc, b, x0, residual=x0)
    q0, s0 = torch.empty(M, 768, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(M, 768, gpu)
    ops.layernorm_mxfp8(x0, gam, bet, q0, s0, 1e-6)
    # fused
    def fused():
        x = torch.full((M + 3, 768), 7.0, device=gpu)
        x[:M] = r
        q, sc = torch.full((M + 3, 768), 9, device=gpu, dtype=torch.uint8), torch.full((6, ((M + 255) // 256) * 256 + 256, 4), 3, device=gpu, dtype=torch.uint8)
        ops.gemm_mx_res_ln(aq, asc, wq, wsc, b, x, gam, bet, q, sc, 1e-6, M=M)
        return x, q, sc
    x1, q1, s1 = fused()
    assert (x1[M:] == 7.0).all() and (q1[M:] == 9).all() and (s1[:, M:] == 3).all(), 'rows beyond M were written'
    for rep in range(3):
        x2, q2, s2 = fused()
        assert torch.equal(x1, x2) and torch.equal(q1, q2) and torch.equal(s1, s2), f'repetition {rep}'
    ref = (_mx_dequant(aq, asc).double() @ _mx_dequant(wq, wsc).double().t()).float() + b + r
    scale = ref.abs().max().item()
    assert (x1[:M] - ref).abs().max().item() < 1e-4 * scale, ((x1[:M] - ref).abs().max().item(), scale)
    assert (x1[:M] - x0).abs().max().item() < 2e-5 * scale
    d0, d1 = _mx_dequant(q0, s0), _mx_dequant(q1[:M], s1)
    diff_bytes = (q0 != q1[:M]).float().mean().item()
    diff_scales = (s0[:, :M] != s1[:, :M]).float().mean().item()
    assert diff_bytes < 2e-3 and diff_scales < 2e-3, (diff_bytes, diff_scales)
    blockmax = d0.abs().view(M, 24, 32).amax(-1).repeat_interleave(32, 1)
    assert ((d1 - d0).abs() <= blockmax * 2.0 ** -2 + 1e-6).all()      # one e4m3 step of the block (two where the block's scale byte itself moved)
    ln = torch.nn.functional.layer_norm(ref.double(), (768,), gam.double(), bet.double(), 1e-6).float()
    rel = ((d1 - ln).norm() / ln.norm()).item()
    assert rel < 4e-2, rel
    if K == 768:                                                     # in place: the output overwrites the A operand and its scale planes
        x3 = r.clone()
        aq3, as3 = aq.clone(), asc.clone()
        ops.gemm_mx_res_ln(aq3, as3, wq, wsc, b, x3, gam, bet, aq3, as3, 1e-6)
        assert torch.equal(x3, x1[:M]) and torch.equal(aq3, q1[:M]) and torch.equal(as3[:, :M], s1[:, :M])


@pytest.mark.parametrize('n_seq', [3, 70])
def test_qkv_time_attention(gpu, n_seq):
    """sf_qkv_time_attention (temporal qkv projection + time attention + CLS-query partials in one launch) against the un-fused sequence it
    replaces: sf_gemm_bf16 -> sf_attention (time groups, CLS key first) + sf_attention_cls.  Both round the projection to bf16 before the
    attention, but sum over k in different tile orders: a few q / k / v elements land one bf16 ulp apart, so the outputs agree to one ulp of
    their magnitude (|o| < 2: 2^-7), not bit for bit.  n_seq = 3: a ragged last tile
    (588 patches = 18.4 tiles); 70: several tile rounds, tiles straddling sequences."""
    from synchformer_amd import ops
    L, D = 1569, 768
    rows = n_seq * L
    x = _bf(_rand(rows, D, seed=50)).to(gpu)
    w, b = _bf(_rand(3 * D, D, seed=51, scale=0.05)).to(gpu), (0.1 * _rand(3 * D, seed=52)).to(gpu)
    # un-fused
    qkv = torch.empty(rows, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x, w, b, qkv)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ref = torch.zeros(rows, D, device=gpu, dtype=torch.bfloat16)
    ops.attention(q, k, v, ref, n_seq=n_seq, seq_rows=L, cls_row=0, heads=12, head_dim=64, scale=0.125, n_groups=196, row0=1, group_stride=1, tok_stride=196, n_tok=8)
    ops.attention_cls(q, k, v, ref, n_seq=n_seq, q_seq_rows=L, q_row=0, kv_seq_rows=L, kv_row0=0, n_keys=L, out_seq_rows=L, out_row=0, heads=12, head_dim=64,
                      scale=0.125)
    # fused
    qkv_cls = torch.empty(n_seq, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x.view(n_seq, L, D)[:, 0], w, b, qkv_cls)
    assert torch.equal(qkv_cls, qkv.view(n_seq, L, 3 * D)[:, 0])
    out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
    part = torch.empty(n_seq * 12 * 196 * 66, device=gpu)
    ops.qkv_time_attention(x, w, b, qkv_cls, out, part, n_seq=n_seq, n_groups=196, scale=0.125)
    assert (out.view(n_seq, L, D)[:, 0] == 7.0).all(), 'the fused kernel must not touch the CLS rows'
    ops.attention_cls_combine(part, out, n_part=49, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    o, r = out.float().view(n_seq, L, D), ref.float().view(n_seq, L, D)
    torch.testing.assert_close(o[:, 1:], r[:, 1:], rtol=2 ** -7, atol=2 ** -7)
    assert (o[:, 1:] - r[:, 1:]).abs().gt(1e-3).float().mean() < 2e-3                    # ... and those are rare
    torch.testing.assert_close(o[:, 0], r[:, 0], rtol=2 ** -6, atol=2 ** -7)
    # and against fp32 torch on the bf16 projection (independent of the un-fused kernels)
    qf = qkv.float().view(n_seq, L, 3, 12, 64)
    qq, kk, vv = qf[:, :, 0], qf[:, :, 1], qf[:, :, 2]                                   # (n, L, 12, 64)
    pk = torch.cat([kk[:, :1].unsqueeze(2).expand(-1, -1, 196, -1, -1), kk[:, 1:].reshape(n_seq, 8, 196, 12, 64)], 1)   # (n, 9, 196, 12, 64)
    pv = torch.cat([vv[:, :1].unsqueeze(2).expand(-1, -1, 196, -1, -1), vv[:, 1:].reshape(n_seq, 8, 196, 12, 64)], 1)
    pq = qq[:, 1:].reshape(n_seq, 8, 196, 12, 64)
    att = torch.einsum('nfphd,ngphd->nphfg', pq, pk) * 0.125
    po = torch.einsum('nphfg,ngphd->nfphd', att.softmax(-1), pv).reshape(n_seq, 8 * 196, D)
    torch.testing.assert_close(o[:, 1:], po, rtol=2 ** -7, atol=2 ** -7)


def test_im2col_video_tokens(gpu):
    """sf_im2col_video_tokens = sf_im2col_video / _clips with every segment's 1568 patch rows at rows 1 .. 1568 of a 1569-row block whose row 0 is zeroed
    (whatever the buffer held before): bit-identical patch rows, for the per-segment input and for segments read in place from un-segmented clips."""
    from synchformer_amd import ops
    g = torch.Generator().manual_seed(7)
    vid = torch.randint(0, 256, (3, 16, 3, 224, 224), generator=g, dtype=torch.uint8).to(gpu)
    ref = torch.empty(3 * 1568, 1536, device=gpu, dtype=torch.bfloat16)
    ops.im2col_video(vid, ref)
    tok = torch.full((3 * 1569 + 2, 1536), 5.0, device=gpu, dtype=torch.bfloat16)
    ops.im2col_video_tokens(vid, tok)
    t3 = tok[:3 * 1569].view(3, 1569, 1536)
    assert torch.equal(t3[:, 1:].reshape(-1, 1536), ref) and (t3[:, 0] == 0).all() and (tok[3 * 1569:] == 5.0).all()
    clips = torch.randint(0, 256, (2, 40, 3, 224, 224), generator=g, dtype=torch.uint8).to(gpu)
    ref2 = torch.empty(2 * 4 * 1568, 1536, device=gpu, dtype=torch.bfloat16)
    ops.im2col_video_clips(clips, ref2, 2, 7, 4)
    tok2 = torch.full((2 * 4 * 1569, 1536), 5.0, device=gpu, dtype=torch.bfloat16)
    ops.im2col_video_tokens(clips, tok2, 2, 7, 4)
    t8 = tok2.view(8, 1569, 1536)
    assert torch.equal(t8[:, 1:].reshape(-1, 1536), ref2) and (t8[:, 0] == 0).all()


@pytest.mark.parametrize('n_seq', [3, 40])
def test_qkv_time_attention_mx(gpu, n_seq):
    """sf_qkv_time_attention_mx against the un-fused MX sequence it replaces: sf_gemm_mxfp8 (bf16 output) -> sf_attention (time groups, CLS key first) +
    sf_attention_cls, on operands whose block scales differ widely along K and across rows (a wrong scale byte, a swapped k half or a wrong row of the gathered
    scale dwords cannot hide).  Both round the projection to bf16 before the attention; the MX products are exact in fp32 up to summation order, so the outputs
    agree to one bf16 ulp of their magnitude.  Repetitions are bit-identical (race screen: the single scale area is refilled one phase after its last read)."""
    from synchformer_amd import ops
    L, D = 1569, 768
    rows = n_seq * L
    g = torch.Generator().manual_seed(90 + n_seq)
    x = _bf(torch.randn(rows, D, generator=g) * torch.exp2(torch.randint(-3, 3, (rows, D // 32), generator=g).float()).repeat_interleave(32, 1)).to(gpu)
    w = _bf(torch.randn(3 * D, D, generator=g) * 0.03 * torch.exp2(torch.randint(-2, 2, (3 * D, D // 32), generator=g).float()).repeat_interleave(32, 1)).to(gpu)
    b = (0.1 * _rand(3 * D, seed=92)).to(gpu)
    xq, xs = torch.empty(rows, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(rows, D, gpu)
    wq, ws = torch.empty(3 * D, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(3 * D, D, gpu)
    ops.quantize_mxfp8(x, xq, xs)
    ops.quantize_mxfp8(w, wq, ws)
    # un-fused
    qkv = torch.empty(rows, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm_mxfp8(xq, xs, wq, ws, b, qkv)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ref = torch.zeros(rows, D, device=gpu, dtype=torch.bfloat16)
    ops.attention(q, k, v, ref, n_seq=n_seq, seq_rows=L, cls_row=0, heads=12, head_dim=64, scale=0.125, n_groups=196, row0=1, group_stride=1, tok_stride=196, n_tok=8)
    ops.attention_cls(q, k, v, ref, n_seq=n_seq, q_seq_rows=L, q_row=0, kv_seq_rows=L, kv_row0=0, n_keys=L, out_seq_rows=L, out_row=0, heads=12, head_dim=64,
                      scale=0.125)
    # fused
    qkv_cls = qkv.view(n_seq, L, 3 * D)[:, 0].contiguous()

    def fused():
        out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
        part = torch.zeros(n_seq * 12 * 49 * 66, device=gpu)
        ops.qkv_time_attention_mx(xq, xs, wq, ws, b, qkv_cls, out, part, n_seq=n_seq, n_groups=196, scale=0.125)
        return out, part
    out, part = fused()
    for rep in range(3):
        o2, p2 = fused()
        assert torch.equal(o2, out), f'repetition {rep}: {(o2 != out).sum().item()} output elements differ'
        assert torch.equal(p2, part), f'repetition {rep}: {(p2 != part).sum().item()} partial elements differ'
    assert (out.view(n_seq, L, D)[:, 0] == 7.0).all(), 'the fused kernel must not touch the CLS rows'
    ops.attention_cls_combine(part, out, n_part=49, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    o, r = out.float().view(n_seq, L, D), ref.float().view(n_seq, L, D)
    scale = r.abs().max().item()
    torch.testing.assert_close(o[:, 1:], r[:, 1:], rtol=2 ** -6, atol=2 ** -7 * max(1.0, scale))
    assert (o[:, 1:] - r[:, 1:]).abs().gt(1e-3 * max(1.0, scale)).float().mean() < 5e-3
    torch.testing.assert_close(o[:, 0], r[:, 0], rtol=2 ** -6, atol=2 ** -7 * max(1.0, scale))
    # the MXFP8-output variant (+ the MX combine for the CLS rows) == sf_quantize_mxfp8 of the bf16 output above, byte for byte
    q0, s0 = torch.empty(rows, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(rows, D, gpu)
    ops.quantize_mxfp8(out, q0, s0)
    q1, s1 = torch.full((rows + 1, D), 7, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(rows, D, gpu)
    part1 = torch.zeros_like(part)
    ops.qkv_time_attention_mx(xq, xs, wq, ws, b, qkv_cls, q1, part1, n_seq=n_seq, n_groups=196, scale=0.125, out_scales=s1)
    assert torch.equal(part1, part) and (q1[:rows].view(n_seq, L, D)[:, 0] == 7).all()
    ops.attention_cls_combine_mx(part1, q1, s1, n_part=49, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    assert torch.equal(q0, q1[:rows]) and (q1[rows:] == 7).all()
    assert torch.equal(s0[:, :rows], s1[:, :rows]) and (s1[:, rows:] == 0).all()


@pytest.mark.parametrize('sched', [0, 1])
def test_qkv_time_attention_masked(gpu, sched):
    """sf_qkv_time_attention_masked against the un-fused masked kernels (sf_gemm_bf16 -> sf_attention_masked + sf_attention_cls_masked, pinned to the real
    reference through tests/golden/e2e_masked_B1S2.npz): random token masks plus a fully masked patch (all 8 frames), a fully masked 4-patch wave (an
    empty CLS partial record) and a masked CLS key in one sequence."""
    from synchformer_amd import ops, _lib
    n_seq, D, L = 3, 768, 1 + 8 * 196
    rows = n_seq * L
    x = _bf(_rand(rows, D, seed=80)).to(gpu)
    w, b = _bf(_rand(3 * D, D, seed=81, scale=0.05)).to(gpu), _rand(3 * D, seed=82).to(gpu)
    g = torch.Generator().manual_seed(83)
    keep = (torch.rand(n_seq, L, generator=g) > 0.2).to(torch.uint8)
    keep[:, 0] = 1
    kp = keep[:, 1:].view(n_seq, 8, 196)
    kp[0, :, 17] = 0                                                          # one patch with all 8 frames masked
    kp[1, :, 40:44] = 0                                                       # one whole wave (patches 40-43 x 8 frames)
    kp[2, :, 0:4] = 0                                                         # the wave that also holds the CLS key's share ...
    keep[2, 0] = 0                                                            # ... with the CLS key masked as well
    keep = keep.reshape(-1).to(gpu)
    qkv = torch.empty(rows, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x, w, b, qkv)
    q, kk, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ref = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
    ops.attention(q, kk, v, ref, n_seq=n_seq, seq_rows=L, cls_row=0, heads=12, head_dim=64, scale=0.125, n_groups=196, row0=1, group_stride=1, tok_stride=196,
                  n_tok=8, key_keep=keep)
    ops.attention_cls(q, kk, v, ref, n_seq=n_seq, q_seq_rows=L, q_row=0, kv_seq_rows=L, kv_row0=0, n_keys=L, out_seq_rows=L, out_row=0, heads=12, head_dim=64,
                      scale=0.125, key_keep=keep)
    qkv_cls = torch.empty(n_seq, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x.view(n_seq, L, D)[:, 0], w, b, qkv_cls)
    out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
    part = torch.empty(n_seq * 12 * 49 * 66, device=gpu)
    lib = _lib.load()
    lib.sf_qkv_time_force_schedule(sched)
    try:
        ops.qkv_time_attention(x, w, b, qkv_cls, out, part, n_seq=n_seq, n_groups=196, scale=0.125, key_keep=keep)
    finally:
        lib.sf_qkv_time_force_schedule(-1)
    ops.attention_cls_combine(part, out, n_part=49, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    o, r = out.float().view(n_seq, L, D), ref.float().view(n_seq, L, D)
    assert torch.isfinite(o[:2]).all()
    torch.testing.assert_close(o[:2, 1:], r[:2, 1:], rtol=2 ** -7, atol=2 ** -7)
    torch.testing.assert_close(o[:2, 0], r[:2, 0], rtol=2 ** -6, atol=2 ** -7)
    # sequence 2 (CLS key masked): the patch rows agree wherever a patch keeps at least one key
    # (the two paths round q | k | v to bf16 out of different GEMM kernels - 32x32x16 vs 16x16x32 MFMA summation order - so a score can move by a bf16 ulp
    # of its operands; with few kept keys and cancelling values that shows as a rare 1e-2 outlier: bounded here, and counted)
    some = kp[2].bool().any(0)                                                # (196,)
    o2, r2 = o[2, 1:].view(8, 196, D)[:, some], r[2, 1:].view(8, 196, D)[:, some]
    torch.testing.assert_close(o2, r2, rtol=2 ** -6, atol=2 ** -6)
    assert (o2 - r2).abs().gt(2 ** -7 * (1 + r2.abs())).float().mean() < 1e-4


@pytest.mark.parametrize('n_seq', [2, 29])
def test_qkv_time_attention_schedules_bitwise(gpu, n_seq):
    """The quadrant-phased main loop of sf_qkv_time_attention (round 3; CLS slices by LDS-DMA) against round 2's loop: same products in the same order,
    same epilogue - attention output and CLS partials bit-identical on every repetition (29 sequences = 178 row tiles x 12 heads: several tiles per
    workgroup, ragged last tile)."""
    from synchformer_amd import ops, _lib
    lib = _lib.load()
    D, L = 768, 1 + 8 * 196
    rows = n_seq * L
    x = _bf(_rand(rows, D, seed=70)).to(gpu)
    w, b = _bf(_rand(3 * D, D, seed=71, scale=0.05)).to(gpu), _rand(3 * D, seed=72).to(gpu)
    qkv_cls = torch.empty(n_seq, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x.view(n_seq, L, D)[:, 0], w, b, qkv_cls)

    def run(sched):
        lib.sf_qkv_time_force_schedule(sched)
        try:
            out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
            part = torch.zeros(n_seq * 12 * 49 * 66, device=gpu)
            ops.qkv_time_attention(x, w, b, qkv_cls, out, part, n_seq=n_seq, n_groups=196, scale=0.125)
        finally:
            lib.sf_qkv_time_force_schedule(-1)
        return out, part
    o0, p0 = run(0)
    for rep in range(5):
        o1, p1 = run(1)
        assert torch.equal(o1, o0) and torch.equal(p1, p0), f'repetition {rep}'


@pytest.mark.parametrize('cfg', [7, 10, 11])
def test_gemm_persistent_epilogues(gpu, request, cfg):
    """The persistent 256x256 kernels (8 waves, 4 waves) over several tile rounds with a ragged last row
    panel: GELU -> bf16 and in-place fp32 residual against fp32 torch on the same bf16 operands."""
    from synchformer_amd import ops, _lib
    if cfg not in PRODUCT_CFGS:
        request.getfixturevalue('ablation')           # config 10 (4 waves) lives in the ablation build
    M, N, K = 256 * 70 + 37, 768, 768
    a, w, b = _bf(_rand(M, K, seed=40)).to(gpu), _bf(_rand(N, K, seed=41, scale=0.05)).to(gpu), _rand(N, seed=42).to(gpu)
    lin = a.float() @ w.float().t() + b
    _lib.load().sf_gemm_force_config(cfg)
    try:
        out = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
        ops.gemm(a, w, b, out, gelu=True)
        x = _rand(M, N, seed=43).to(gpu)
        x0 = x.clone()
        ops.gemm(a, w, b, x, residual=x)
    finally:
        _lib.load().sf_gemm_force_config(-1)
    torch.testing.assert_close(out.float(), torch.nn.functional.gelu(lin), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(x, lin + x0, rtol=1e-4, atol=3e-4)


@pytest.mark.parametrize('M,N,K,gelu,res', [(256 * 70 + 37, 2304, 768, False, False), (256 * 131 + 1, 3072, 768, True, False), (256 * 67 + 255, 768, 3072, False, True),
                                            (256 * 9, 768, 1536, False, False), (300, 768, 256, True, False)])
def test_gemm_pp_bitwise_equals_config7(gpu, M, N, K, gelu, res):
    """The quadrant-phased persistent kernel (config 11) sums every accumulator over k in the same order with the same MFMA as config 7: the
    outputs must be bit-identical, on every repetition (the repetitions screen for LDS-DMA / ds_read ordering races, which would show up as
    rare wrong tiles)."""
    from synchformer_amd import ops, _lib
    a, w, b = _bf(_rand(M, K, seed=50)).to(gpu), _bf(_rand(N, K, seed=51, scale=0.05)).to(gpu), _rand(N, seed=52).to(gpu)
    x0 = _rand(M, N, seed=53).to(gpu)
    lib = _lib.load()

    def run(cfg, k_major):
        lib.sf_gemm_force_config(cfg)
        try:
            if res:
                out = x0.clone()
                ops.gemm(a, ops.ktile_major_weight(w) if k_major else w, b, out, residual=out)
            else:
                out = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
                ops.gemm(a, ops.ktile_major_weight(w) if k_major else w, b, out, gelu=gelu)
        finally:
            lib.sf_gemm_force_config(-1)
        return out
    ref = run(7, False)
    for rep in range(6):
        assert torch.equal(run(11, False), ref), f'repetition {rep}'
    if M >= 8192:
        assert torch.equal(run(11, True), ref), 'k-tile-major weight'


@pytest.mark.parametrize('M,N,K,gelu,bias', [
    (256 * 9 + 77, 768, 768, False, True),          # ragged last row tile, several tiles per workgroup on a small grid
    (256 * 40 + 1, 3072, 768, True, True),          # fc1 + GELU
    (256 * 3, 2304, 256, False, False),             # the shortest K (two k-tile pairs), no bias
    (8192 + 130, 768, 3072, True, True),            # long K
    (300, 128, 384, False, True),                   # one tile, N < 256: the wn = 1 waves of the tile's second column half have nothing to store
])
def test_gemm_r4_bitwise_equals_config11(gpu, ablation, M, N, K, gelu, bias):
    """Config 12 (sf_gemm_w4.hip: four waves, register-resident fragments, one LDS-DMA piece behind every fourth MFMA) multiplies the same 32x32x16 blocks in the
    same k order as config 11 and runs the same bias / exact-erf GELU / bf16 rounding on them: bit-identical outputs on every repetition (the repetitions screen
    for LDS-DMA / ds_read ordering races of the landing ring); rows beyond M stay untouched."""
    from synchformer_amd import ops, _lib
    a, w = _bf(_rand(M, K, seed=150)).to(gpu), _bf(_rand(N, K, seed=151, scale=0.05)).to(gpu)
    b = _rand(N, seed=152).to(gpu) if bias else None
    lib = _lib.load()

    def run(cfg):
        lib.sf_gemm_force_config(cfg)
        try:
            out = torch.full((M + 3, N), 7.0, device=gpu, dtype=torch.bfloat16)
            ops.gemm(a, w, b, out, M=M, gelu=gelu)
        finally:
            lib.sf_gemm_force_config(-1)
        return out
    ref = run(11)
    assert torch.isfinite(ref.float()).all()
    for rep in range(6):
        got = run(12)
        assert torch.equal(got[M:], torch.full((3, N), 7.0, device=gpu, dtype=torch.bfloat16)), 'rows beyond M were written'
        assert torch.equal(got, ref), f'repetition {rep}: {(got.float() - ref.float()).abs().max().item()}'


def test_gemm_gelu_dual(gpu):
    """sf_gemm_bf16_gelu_dual (fc1 of a trained MLP: pre-activation AND gelu(pre) from one launch of config 11): the pre-activation is bit-identical to
    sf_gemm_bf16's bf16 output on config 11, the activation to its GELU epilogue; shapes outside config 11's range return 1 without launching."""
    from synchformer_amd import ops, _lib
    lib = _lib.load()
    M, N, K = 256 * 37 + 19, 3072, 768
    a, w, b = _bf(_rand(M, K, seed=70)).to(gpu), _bf(_rand(N, K, seed=71, scale=0.05)).to(gpu), _rand(N, seed=72).to(gpu)
    st = torch.cuda.current_stream().cuda_stream
    pre, act = torch.full((M + 2, N), 3.0, device=gpu, dtype=torch.bfloat16), torch.full((M + 2, N), 5.0, device=gpu, dtype=torch.bfloat16)
    rc = lib.sf_gemm_bf16_gelu_dual(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), pre.data_ptr(), act.data_ptr(), N, M, N, K, st)
    assert rc == 0
    lib.sf_gemm_force_config(11)
    try:
        p0, a0 = torch.empty(M, N, device=gpu, dtype=torch.bfloat16), torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
        ops.gemm(a, w, b, p0)
        ops.gemm(a, w, b, a0, gelu=True)
    finally:
        lib.sf_gemm_force_config(-1)
    assert torch.equal(pre[:M], p0) and torch.equal(act[:M], a0)
    assert (pre[M:] == 3.0).all() and (act[M:] == 5.0).all(), 'rows beyond M were written'
    ref = a.float() @ w.float().t() + b
    torch.testing.assert_close(act[:M].float(), torch.nn.functional.gelu(ref), rtol=1e-2, atol=2e-2)
    assert lib.sf_gemm_bf16_gelu_dual(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), pre.data_ptr(), act.data_ptr(), N, M, N, 192, st) == -2    # K % 128 != 0


def test_gemm_ktile_major_weight(gpu):
    """sf_gemm_bf16 with the weight given k-tile-major ((K/64, N, 64), ldw == 64): the same products in the same order as the row-major weight."""
    from synchformer_amd import ops
    M, N, K = 9000, 2304, 768
    a, w, b = _bf(_rand(M, K, seed=1)).to(gpu), _bf(_rand(N, K, seed=2, scale=0.05)).to(gpu), _rand(N, seed=3).to(gpu)
    o0, o1 = torch.empty(M, N, device=gpu, dtype=torch.bfloat16), torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
    ops.gemm(a, w, b, o0, gelu=True)
    ops.gemm(a, ops.ktile_major_weight(w), b, o1, gelu=True)
    assert torch.equal(o0, o1)
    with pytest.raises(RuntimeError, match='k-tile-major'):
        ops.gemm(a[:100], ops.ktile_major_weight(w), b, o1[:100])


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def test_functional_ops_autograd(gpu):
    """torch.ops.synchformer.linear / layer_norm768 (synchformer_amd/functional.py): functional dispatcher ops with an autograd formula on the train steps'
    backward kernels, an autocast rule and FakeTensor implementations - forward and every gradient against fp32 torch on the same (bf16-rounded) operands."""
    from synchformer_amd import functional as SF
    g = torch.Generator().manual_seed(7)
    B, L, K, N = 2, 520, 768, 2304
    x = torch.randn(B, L, K, generator=g).to(gpu).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(K, generator=g)).to(gpu).requires_grad_(True)
    beta = (0.1 * torch.randn(K, generator=g)).to(gpu).requires_grad_(True)
    w = (0.02 * torch.randn(N, K, generator=g)).to(gpu).requires_grad_(True)
    b = (0.02 * torch.randn(N, generator=g)).to(gpu).requires_grad_(True)
    dy = torch.randn(B, L, N, generator=g).to(gpu)
    with torch.autocast('cuda', dtype=torch.bfloat16):                          # the rules: layer_norm768 sees fp32, linear sees bf16
        h = SF.layer_norm768(x, gamma, beta, 1e-6)
        y = SF.linear(h, w, b)
    assert h.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and y.shape == (B, L, N)
    (y.float() * dy).sum().backward()
    got = [t.grad.clone() for t in (x, gamma, beta, w, b)]
    # fp32 torch on the operands the kernels saw: bf16(h), bf16(w), bf16(dy)
    xr, gr, br, wr, bb = (t.detach().clone().requires_grad_(True) for t in (x, gamma, beta, w, b))
    hr = torch.nn.functional.layer_norm(xr, (K,), gr, br, 1e-6)
    h_b = hr.detach().bfloat16().float().requires_grad_(True)
    yr = torch.nn.functional.linear(h_b, wr.detach().bfloat16().float().requires_grad_(True), bb)
    assert (y.float() - yr).abs().max().item() < 3e-2 and _rel(h.float(), hr) < 4e-3
    dyb = dy.bfloat16().float()
    wq = wr.detach().bfloat16().float()
    dh = dyb.reshape(-1, N) @ wq                                               # dX of the linear
    dw_ref = dyb.reshape(-1, N).t() @ h_b.detach().reshape(-1, K)
    db_ref = dyb.reshape(-1, N).sum(0)
    hr.backward(dh.reshape(B, L, K).bfloat16().float())                         # the LayerNorm backward receives dX as autograd hands it over (bf16 here)
    assert _rel(got[3], dw_ref) < 5e-3 and _rel(got[4], db_ref) < 5e-3, (_rel(got[3], dw_ref), _rel(got[4], db_ref))
    assert _rel(got[0], xr.grad) < 1e-2 and _rel(got[1], gr.grad) < 1e-2 and _rel(got[2], br.grad) < 1e-2, [_rel(a, b_) for a, b_ in zip(got[:3], (xr.grad, gr.grad, br.grad))]
    torch.library.opcheck(torch.ops.synchformer.linear.default, (x.detach().bfloat16(), w.detach().bfloat16(), b.detach()), test_utils=('test_schema', 'test_faketensor'))
    # a shape the backward does not serve raises where the graph is BUILT (setup_context), not first in backward(); without grad the same call is fine
    with pytest.raises(NotImplementedError, match='N % 128'):
        SF.linear(x.detach()[:, :, :64].contiguous().requires_grad_(True), torch.zeros(100, 64, device=gpu, requires_grad=True))
    with torch.no_grad():
        assert SF.linear(x.detach()[:, :, :64].contiguous(), torch.zeros(100, 64, device=gpu)).dtype == torch.bfloat16
    # only the gradients autograd asks for are computed: a frozen weight / bias gets none, dX is unchanged
    x2 = x.detach().clone().requires_grad_(True)
    y2 = SF.linear(SF.layer_norm768(x2, gamma.detach(), beta.detach(), 1e-6), w.detach(), b.detach())
    (y2.float() * dy).sum().backward()
    assert _rel(x2.grad, got[0]) < 1e-6
    dxe, dwe, dbe = torch.ops.synchformer.linear_backward(dy.bfloat16(), h.detach(), w.detach(), False, True, False)
    assert dxe.numel() == 0 and dbe.numel() == 0 and _rel(dwe, dw_ref) < 5e-3


def test_functional_linear_autocast_training_steps_fresh_dx(gpu):
    """under autocast `linear_backward` receives the per-forward bf16 cast of the weight - a temporary whose device address the caching allocator
    reuses step after step and whose version is always 0.  A W^T cache keyed on (data_ptr, version) served the FIRST step's transpose for ever after.  Three SGD
    steps under autocast with a large learning rate: dX of every step must follow the UPDATED weight (fp32 torch on the same bf16 operands), and the same for a
    leaf weight updated in place outside autocast (the cached case: the version counter moves)."""
    from synchformer_amd import functional as SF
    g = torch.Generator().manual_seed(11)
    M, K, N = 1024, 768, 768
    for autocast in (True, False):
        w = (0.05 * torch.randn(N, K, generator=g)).to(gpu)
        w = (w if autocast else w.bfloat16()).requires_grad_(True)
        for step in range(3):
            x = torch.randn(M, K, generator=g).to(gpu).bfloat16().requires_grad_(True)
            dy = torch.randn(M, N, generator=g).to(gpu).bfloat16()
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=autocast):
                y = SF.linear(x, w, None)
            y.backward(dy)
            ref = dy.float() @ w.detach().bfloat16().float()
            assert _rel(x.grad, ref) < 5e-3, (autocast, step, _rel(x.grad, ref))
            with torch.no_grad():
                w.add_(torch.randn(N, K, generator=g).to(gpu).to(w.dtype), alpha=0.05)     # a step that changes W by as much as W itself
            w.grad = None


@pytest.mark.parametrize('n_seq', [3, 40])
def test_qkv_space_attention(gpu, n_seq):
    """sf_qkv_space_attention (spatial qkv projection + space attention + CLS-query partials in one launch; the side rows from a 33-rows-per-segment GEMM) against
    the un-fused sequence it replaces: sf_gemm_bf16 -> sf_attention_cls_partial (space groups, CLS key first) + sf_attention_cls_combine.  Both round the projection
    to bf16 before the attention but sum over k in different tile orders: a few q / k / v elements land one bf16 ulp apart, so the outputs agree to one ulp of their
    magnitude, not bit for bit.  Repetitions are bit-identical (race screen of the LDS-DMA schedule).  n_seq = 3: 144 work items, fewer than CUs; 40: several rounds."""
    from synchformer_amd import ops
    L, D = 1569, 768
    rows = n_seq * L
    x = _bf(_rand(rows, D, seed=150)).to(gpu)
    w, b = _bf(_rand(3 * D, D, seed=151, scale=0.05)).to(gpu), (0.1 * _rand(3 * D, seed=152)).to(gpu)
    # un-fused
    qkv = torch.empty(rows, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x, w, b, qkv)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ref = torch.zeros(rows, D, device=gpu, dtype=torch.bfloat16)
    part_ref = torch.zeros(n_seq * 12 * 8 * 66, device=gpu)
    ops.attention_cls_partial(q, k, v, ref, part_ref, n_seq=n_seq, seq_rows=L, n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196, cls_row=0, heads=12,
                              head_dim=64, scale=0.125)
    ops.attention_cls_combine(part_ref, ref, n_part=8, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    # fused
    side_in = torch.empty(n_seq * 33, D, device=gpu, dtype=torch.bfloat16)
    ops.space_side_rows(x, side_in, n_seq)
    xv = x.view(n_seq, L, D)
    assert torch.equal(side_in.view(n_seq, 33, D)[:, 0], xv[:, 0])
    assert torch.equal(side_in.view(n_seq, 33, D)[:, 1:].reshape(n_seq, 8, 4, D), xv[:, 1:].reshape(n_seq, 8, 196, D)[:, :, 192:])
    side = torch.empty(n_seq * 33, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(side_in, w, b, side)

    def fused():
        out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
        part = torch.zeros(n_seq * 12 * 8 * 66, device=gpu)
        ops.qkv_space_attention(x, w, b, side, out, part, n_seq=n_seq, scale=0.125)
        return out, part
    out, part = fused()
    for rep in range(3):
        o2, p2 = fused()
        assert torch.equal(o2, out), f'repetition {rep}: {(o2 != out).sum().item()} output elements differ'
        assert torch.equal(p2, part), f'repetition {rep}: {(p2 != part).sum().item()} partial elements differ'
    assert (out.view(n_seq, L, D)[:, 0] == 7.0).all(), 'the fused kernel must not touch the CLS rows'
    ops.attention_cls_combine(part, out, n_part=8, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    o, r = out.float().view(n_seq, L, D), ref.float().view(n_seq, L, D)
    torch.testing.assert_close(o[:, 1:], r[:, 1:], rtol=2 ** -7, atol=2 ** -7)
    assert (o[:, 1:] - r[:, 1:]).abs().gt(1e-3).float().mean() < 2e-3                    # ... and those are rare
    torch.testing.assert_close(o[:, 0], r[:, 0], rtol=2 ** -6, atol=2 ** -7)
    # and against fp32 torch on the bf16 projection (independent of the un-fused attention kernel): per frame, keys [CLS; the frame's 196 tokens]
    qf = qkv.float().view(n_seq, L, 3, 12, 64)[:2]
    qq, kk, vv = qf[:, :, 0], qf[:, :, 1], qf[:, :, 2]                                   # (n, L, 12, 64)
    m = qq.shape[0]
    pk = torch.cat([kk[:, :1].unsqueeze(1).expand(-1, 8, -1, -1, -1), kk[:, 1:].reshape(m, 8, 196, 12, 64)], 2)   # (n, 8, 197, 12, 64)
    pv = torch.cat([vv[:, :1].unsqueeze(1).expand(-1, 8, -1, -1, -1), vv[:, 1:].reshape(m, 8, 196, 12, 64)], 2)
    pq = qq[:, 1:].reshape(m, 8, 196, 12, 64)
    att = torch.einsum('nfqhd,nfkhd->nfhqk', pq, pk) * 0.125
    po = torch.einsum('nfhqk,nfkhd->nfqhd', att.softmax(-1), pv).reshape(m, 8 * 196, D)
    torch.testing.assert_close(o[:2, 1:], po, rtol=2 ** -7, atol=2 ** -7)


@pytest.mark.parametrize('n_seq', [3, 40])
def test_qkv_fused_attention_key_masks(gpu, n_seq):
    """Round 5: the token-mask forms of the two fused attention launches (sf_qkv_space_attention_masked, sf_qkv_time_attention2_masked; vit_helper.py:107-141 with the
    masks of sync_model.py:72-80) against the un-fused masked launches they replace on masked forwards - sf_gemm_bf16 + sf_attention_cls_partial_masked (space groups)
    and sf_qkv_time_attention_masked - and the all-ones mask bit-identical to the unmasked launch.  The mask drops random tokens, one whole frame (a CLS partial record
    with every key masked: m = -inf, l = 0 through the combine) and one patch in all its 8 frames (a time group whose only key is the CLS key)."""
    from synchformer_amd import ops
    L, D = 1569, 768
    rows = n_seq * L
    x = _bf(_rand(rows, D, seed=160)).to(gpu)
    w, b = _bf(_rand(3 * D, D, seed=161, scale=0.05)).to(gpu), (0.1 * _rand(3 * D, seed=162)).to(gpu)
    g = torch.Generator().manual_seed(163)
    keep = (torch.rand(n_seq, L, generator=g) > 0.3)
    keep[:, 0] = True                                                   # the CLS token is never masked (sf_token_mask_video leaves it 1)
    keep[1, 1 + 3 * 196:1 + 4 * 196] = False                            # sequence 1: frame 3 completely masked
    keep[2, 1 + 50::196] = False                                        # sequence 2: patch 50 masked in every frame
    keep[0, 1 + 194::196] = False                                       # ... and a left-over patch (tokens 192..195 go through the side rows)
    keep = keep.reshape(rows).to(torch.uint8).to(gpu)
    ones = torch.ones(rows, device=gpu, dtype=torch.uint8)
    side_in = torch.empty(n_seq * 33, D, device=gpu, dtype=torch.bfloat16)
    ops.space_side_rows(x, side_in, n_seq)
    side = torch.empty(n_seq * 33, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(side_in, w, b, side)
    qkv = torch.empty(rows, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x, w, b, qkv)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]

    # ---- space
    def space(kk):
        out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
        part = torch.zeros(n_seq * 12 * 8 * 66, device=gpu)
        ops.qkv_space_attention(x, w, b, side, out, part, n_seq=n_seq, scale=0.125, key_keep=kk)
        return out, part
    o_none, p_none = space(None)
    o_ones, p_ones = space(ones)
    assert torch.equal(o_ones, o_none) and torch.equal(p_ones, p_none), 'an all-ones mask must be bit-identical to the unmasked launch'
    o_m, p_m = space(keep)
    o_m2, p_m2 = space(keep)
    assert torch.equal(o_m, o_m2) and torch.equal(p_m, p_m2)
    assert (o_m.float() - o_none.float()).abs().max() > 0.05
    ref = torch.zeros(rows, D, device=gpu, dtype=torch.bfloat16)
    part_ref = torch.zeros(n_seq * 12 * 8 * 66, device=gpu)
    ops.attention_cls_partial(q, k, v, ref, part_ref, n_seq=n_seq, seq_rows=L, n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196, cls_row=0, heads=12,
                              head_dim=64, scale=0.125, key_keep=keep)
    ops.attention_cls_combine(part_ref, ref, n_part=8, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    ops.attention_cls_combine(p_m, o_m, n_part=8, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    o, r = o_m.float().view(n_seq, L, D), ref.float().view(n_seq, L, D)
    assert torch.isfinite(o).all()
    # (both round the projection to bf16 but sum over k in different tile orders: a few q / k / v elements land one bf16 ulp apart; with a third of the keys gone a
    #  softmax averages over fewer of them than in test_qkv_space_attention - one more ulp of head room, and the large deviations stay rare)
    torch.testing.assert_close(o[:, 1:], r[:, 1:], rtol=2 ** -6, atol=2 ** -6)
    assert (o[:, 1:] - r[:, 1:]).abs().gt(2e-3).float().mean() < 2e-3
    torch.testing.assert_close(o[:, 0], r[:, 0], rtol=2 ** -6, atol=2 ** -7)

    # ---- time
    def time2(kk):
        out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
        part = torch.zeros(n_seq * 12 * 33 * 66, device=gpu)
        ops.qkv_time_attention2(x, w, b, side, out, part, n_seq=n_seq, scale=0.125, key_keep=kk)
        return out, part
    t_none, tp_none = time2(None)
    t_ones, tp_ones = time2(ones)
    assert torch.equal(t_ones, t_none) and torch.equal(tp_ones, tp_none)
    t_m, tp_m = time2(keep)
    t_m2, tp_m2 = time2(keep)
    assert torch.equal(t_m, t_m2) and torch.equal(tp_m, tp_m2)
    assert (t_m.float() - t_none.float()).abs().max() > 0.05
    qkv_cls = torch.empty(n_seq, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x.view(n_seq, L, D)[:, 0], w, b, qkv_cls)
    ref_t = torch.zeros(rows, D, device=gpu, dtype=torch.bfloat16)
    part_t = torch.zeros(n_seq * 12 * 49 * 66, device=gpu)
    ops.qkv_time_attention(x, w, b, qkv_cls, ref_t, part_t, n_seq=n_seq, n_groups=196, scale=0.125, key_keep=keep)
    ops.attention_cls_combine(part_t, ref_t, n_part=49, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    ops.attention_cls_combine(tp_m, t_m, n_part=33, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    o, r = t_m.float().view(n_seq, L, D), ref_t.float().view(n_seq, L, D)
    assert torch.isfinite(o).all()
    torch.testing.assert_close(o[:, 1:], r[:, 1:], rtol=2 ** -6, atol=2 ** -6)     # (P rounded to bf16 for the P V MFMA here, 9 keys per query: as test_qkv_time_attention2)
    torch.testing.assert_close(o[:, 0], r[:, 0], rtol=2 ** -6, atol=2 ** -7)


@pytest.mark.parametrize('n_seq', [3, 40])
def test_qkv_time_attention2(gpu, n_seq):
    """sf_qkv_time_attention2 (temporal qkv projection + time attention + CLS-query partials on the 192 x 384 main loop; 24-patch blocks, the 4 left-over patches
    and the CLS row from the 33-rows-per-segment side GEMM) against the un-fused sequence: sf_gemm_bf16 -> sf_attention_cls_partial (time groups, CLS key first) +
    sf_attention_cls_combine, against sf_qkv_time_attention, and against fp32 torch on the bf16 projection.  All round the projection to bf16 before the attention; this
    kernel rounds the probabilities to bf16 for the P V MFMA (as the space attention does): outputs agree to one bf16 ulp of their magnitude.  Repetitions bit-identical."""
    from synchformer_amd import ops
    L, D = 1569, 768
    rows = n_seq * L
    x = _bf(_rand(rows, D, seed=160)).to(gpu)
    w, b = _bf(_rand(3 * D, D, seed=161, scale=0.05)).to(gpu), (0.1 * _rand(3 * D, seed=162)).to(gpu)
    qkv = torch.empty(rows, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x, w, b, qkv)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ref = torch.zeros(rows, D, device=gpu, dtype=torch.bfloat16)
    part_ref = torch.zeros(n_seq * 12 * 196 * 66, device=gpu)
    ops.attention_cls_partial(q, k, v, ref, part_ref, n_seq=n_seq, seq_rows=L, n_groups=196, row0=1, group_stride=1, tok_stride=196, n_tok=8, cls_row=0, heads=12,
                              head_dim=64, scale=0.125)
    ops.attention_cls_combine(part_ref, ref, n_part=196, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    side_in = torch.empty(n_seq * 33, D, device=gpu, dtype=torch.bfloat16)
    ops.space_side_rows(x, side_in, n_seq)
    side = torch.empty(n_seq * 33, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(side_in, w, b, side)

    def fused():
        out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
        part = torch.zeros(n_seq * 12 * 33 * 66, device=gpu)
        ops.qkv_time_attention2(x, w, b, side, out, part, n_seq=n_seq, scale=0.125)
        return out, part
    out, part = fused()
    for rep in range(3):
        o2, p2 = fused()
        assert torch.equal(o2, out), f'repetition {rep}: {(o2 != out).sum().item()} output elements differ'
        assert torch.equal(p2, part), f'repetition {rep}: {(p2 != part).sum().item()} partial elements differ'
    assert (out.view(n_seq, L, D)[:, 0] == 7.0).all(), 'the fused kernel must not touch the CLS rows'
    ops.attention_cls_combine(part, out, n_part=33, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    o, r = out.float().view(n_seq, L, D), ref.float().view(n_seq, L, D)
    # (9 keys per query: P rounded to bf16 - 2^-9 relative on up to 9 terms of |v| <= ~4 - does not average out as over 197 keys: a handful of 48 M elements reach 0.012)
    torch.testing.assert_close(o[:, 1:], r[:, 1:], rtol=2 ** -7, atol=2 ** -6)
    assert (o[:, 1:] - r[:, 1:]).abs().mean() < 1e-3                                      # (9 keys per query: the bf16 rounding of P does not average out as over 197)
    torch.testing.assert_close(o[:, 0], r[:, 0], rtol=2 ** -6, atol=2 ** -7)
    # the round-3 fused launch on the same operands
    qkv_cls = torch.empty(n_seq, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm(x.view(n_seq, L, D)[:, 0], w, b, qkv_cls)
    o3 = torch.zeros(rows, D, device=gpu, dtype=torch.bfloat16)
    p3 = torch.zeros(n_seq * 12 * 49 * 66, device=gpu)
    ops.qkv_time_attention(x, w, b, qkv_cls, o3, p3, n_seq=n_seq, n_groups=196, scale=0.125)
    ops.attention_cls_combine(p3, o3, n_part=49, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    torch.testing.assert_close(o, o3.float().view(n_seq, L, D), rtol=2 ** -6, atol=2 ** -6)
    # fp32 torch on the bf16 projection: per patch, keys [CLS; the patch's 8 frames]; the CLS query over all 1569 keys
    qf = qkv.float().view(n_seq, L, 3, 12, 64)[:2]
    qq, kk, vv = qf[:, :, 0], qf[:, :, 1], qf[:, :, 2]                                   # (n, L, 12, 64)
    m = qq.shape[0]
    tk = torch.cat([kk[:, :1].unsqueeze(1).expand(-1, 196, -1, -1, -1), kk[:, 1:].reshape(m, 8, 196, 12, 64).transpose(1, 2)], 2)   # (n, 196, 9, 12, 64)
    tv = torch.cat([vv[:, :1].unsqueeze(1).expand(-1, 196, -1, -1, -1), vv[:, 1:].reshape(m, 8, 196, 12, 64).transpose(1, 2)], 2)
    tq = qq[:, 1:].reshape(m, 8, 196, 12, 64).transpose(1, 2)                            # (n, 196, 8, 12, 64)
    att = torch.einsum('npqhd,npkhd->nphqk', tq, tk) * 0.125
    po = torch.einsum('nphqk,npkhd->npqhd', att.softmax(-1), tv).transpose(1, 2).reshape(m, 8 * 196, D)
    torch.testing.assert_close(o[:2, 1:], po, rtol=2 ** -7, atol=2 ** -6)
    ca = torch.einsum('nhd,nkhd->nhk', qq[:, 0], kk) * 0.125
    co = torch.einsum('nhk,nkhd->nhd', ca.softmax(-1), vv).reshape(m, D)
    torch.testing.assert_close(o[:2, 0], co, rtol=2 ** -6, atol=2 ** -7)


def _space_side_index(n_seq, dev):
    """row ids of [CLS; tokens 192..195 of each of the 8 frames] per sequence (33 per sequence), as synchformer_amd.ops.space_side_rows gathers them"""
    seq = torch.arange(n_seq, device=dev).view(n_seq, 1) * 1569
    left = 1 + torch.arange(8, device=dev).view(8, 1) * 196 + 192 + torch.arange(4, device=dev).view(1, 4)
    return torch.cat([seq, seq + left.reshape(1, 32)], 1).reshape(-1)


@pytest.mark.parametrize('n_seq', [3, 40])
def test_qkv_time_attention2_mx(gpu, n_seq):
    """sf_qkv_time_attention2_mx (round 5: the temporal half of the fp8 towers on the 192 x 384 main loop) against the launch it replaces, sf_qkv_time_attention_mx with the
    CLS rows' projection from a small MX GEMM, on operands whose block scales differ widely along K and across rows.  Both round the projection to bf16 before the
    attention; this kernel rounds the probabilities to bf16 for the P V MFMA (9 keys per query): one bf16 ulp of head room, as test_qkv_time_attention2.  Repetitions are
    bit-identical; the MXFP8 output form is byte for byte sf_quantize_mxfp8 of the bf16 output; the side rows (e4m3 bytes AND scale dwords) come from ONE sf_side_rows launch."""
    from synchformer_amd import ops
    L, D = 1569, 768
    rows = n_seq * L
    g = torch.Generator().manual_seed(290 + n_seq)
    x = _bf(torch.randn(rows, D, generator=g) * torch.exp2(torch.randint(-3, 3, (rows, D // 32), generator=g).float()).repeat_interleave(32, 1)).to(gpu)
    w = _bf(torch.randn(3 * D, D, generator=g) * 0.03 * torch.exp2(torch.randint(-2, 2, (3 * D, D // 32), generator=g).float()).repeat_interleave(32, 1)).to(gpu)
    b = (0.1 * _rand(3 * D, seed=292)).to(gpu)
    xq, xs = torch.empty(rows, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(rows, D, gpu)
    wq, ws = torch.empty(3 * D, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(3 * D, D, gpu)
    ops.quantize_mxfp8(x, xq, xs)
    ops.quantize_mxfp8(w, wq, ws)
    # the side rows: one gather of the e4m3 rows and of their scale dwords == index_select of both
    idx = _space_side_index(n_seq, gpu)
    sq, ss = torch.zeros(n_seq * 33, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(n_seq * 33, D, gpu)
    ops.space_side_rows_mx(xq, xs, sq, ss, n_seq)
    assert torch.equal(sq, xq.index_select(0, idx)) and torch.equal(ss[:, :n_seq * 33], xs.index_select(1, idx))
    side = torch.empty(n_seq * 33, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm_mxfp8(sq, ss, wq, ws, b, side)
    # reference: round 3's fused MX launch
    cq = xq.view(n_seq, L, D)[:, 0].contiguous()
    cs = ops.mx_scale_planes(n_seq, D, gpu)
    cs[:, :n_seq] = xs[:, :rows].view(6, n_seq, L, 4)[:, :, 0]
    qkv_cls = torch.empty(n_seq, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm_mxfp8(cq, cs, wq, ws, b, qkv_cls)
    assert torch.equal(qkv_cls, side.view(n_seq, 33, 3 * D)[:, 0])
    ref = torch.zeros(rows, D, device=gpu, dtype=torch.bfloat16)
    part_ref = torch.zeros(n_seq * 12 * 49 * 66, device=gpu)
    ops.qkv_time_attention_mx(xq, xs, wq, ws, b, qkv_cls, ref, part_ref, n_seq=n_seq, n_groups=196, scale=0.125)
    ops.attention_cls_combine(part_ref, ref, n_part=49, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)

    def fused():
        out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
        part = torch.zeros(n_seq * 12 * 33 * 66, device=gpu)
        ops.qkv_time_attention2_mx(xq, xs, wq, ws, b, side, out, part, n_seq=n_seq, scale=0.125)
        return out, part
    out, part = fused()
    for rep in range(3):
        o2, p2 = fused()
        assert torch.equal(o2, out), f'repetition {rep}: {(o2 != out).sum().item()} output elements differ'
        assert torch.equal(p2, part), f'repetition {rep}: {(p2 != part).sum().item()} partial elements differ'
    assert (out.view(n_seq, L, D)[:, 0] == 7.0).all(), 'the fused kernel must not touch the CLS rows'
    # the MXFP8 output form: byte for byte the quantisation of the bf16 output (patch rows)
    oq, os_ = torch.zeros(rows, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(rows, D, gpu)
    part2 = torch.zeros_like(part)
    ops.qkv_time_attention2_mx(xq, xs, wq, ws, b, side, oq, part2, n_seq=n_seq, scale=0.125, out_scales=os_)
    wq_, ws_ = torch.zeros(rows, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(rows, D, gpu)
    ops.quantize_mxfp8(out, wq_, ws_)
    patch = torch.ones(n_seq, L, dtype=torch.bool, device=gpu); patch[:, 0] = False
    patch = patch.reshape(-1)
    assert torch.equal(oq[patch], wq_[patch]) and torch.equal(os_[:, :rows][:, patch], ws_[:, :rows][:, patch]) and torch.equal(part2, part)
    ops.attention_cls_combine(part, out, n_part=33, n_seq=n_seq, out_seq_rows=L, out_row=0, heads=12)
    o, r = out.float().view(n_seq, L, D), ref.float().view(n_seq, L, D)
    scale = r.abs().max().item()
    torch.testing.assert_close(o[:, 1:], r[:, 1:], rtol=2 ** -5, atol=2 ** -6 * max(1.0, scale))
    assert (o[:, 1:] - r[:, 1:]).abs().gt(2e-3 * max(1.0, scale)).float().mean() < 1e-2
    torch.testing.assert_close(o[:, 0], r[:, 0], rtol=2 ** -5, atol=2 ** -6 * max(1.0, scale))


@pytest.mark.parametrize('n_seq', [3, 40])
def test_qkv_space_attention_mx(gpu, n_seq):
    """sf_qkv_space_attention_mx against the un-fused MX sequence it replaces: sf_gemm_mxfp8 (bf16 output) -> sf_attention_cls_partial (space groups) + combine, on operands
    whose block scales differ widely along K and across rows (a wrong scale byte, a swapped k half or a wrong row of the scale dwords cannot hide).  Both round the
    projection to bf16 before the attention; the MX products are exact in fp32 up to summation order, so the outputs agree to one bf16 ulp of their magnitude.  Repetitions
    are bit-identical; the MXFP8 output form is byte for byte sf_quantize_mxfp8 of the bf16 output."""
    from synchformer_amd import ops
    L, D = 1569, 768
    rows = n_seq * L
    g = torch.Generator().manual_seed(190 + n_seq)
    x = _bf(torch.randn(rows, D, generator=g) * torch.exp2(torch.randint(-3, 3, (rows, D // 32), generator=g).float()).repeat_interleave(32, 1)).to(gpu)
    w = _bf(torch.randn(3 * D, D, generator=g) * 0.03 * torch.exp2(torch.randint(-2, 2, (3 * D, D // 32), generator=g).float()).repeat_interleave(32, 1)).to(gpu)
    b = (0.1 * _rand(3 * D, seed=192)).to(gpu)
    xq, xs = torch.empty(rows, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(rows, D, gpu)
    wq, ws = torch.empty(3 * D, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(3 * D, D, gpu)
    ops.quantize_mxfp8(x, xq, xs)
    ops.quantize_mxfp8(w, wq, ws)
    # un-fused
    qkv = torch.empty(rows, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm_mxfp8(xq, xs, wq, ws, b, qkv)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    ref = torch.zeros(rows, D, device=gpu, dtype=torch.bfloat16)
    part_ref = torch.zeros(n_seq * 12 * 8 * 66, device=gpu)
    ops.attention_cls_partial(q, k, v, ref, part_ref, n_seq=n_seq, seq_rows=L, n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196, cls_row=0, heads=12,
                              head_dim=64, scale=0.125)
    # fused: side rows = gathered copies of the rows and of their scale dwords through the MX GEMM
    idx = _space_side_index(n_seq, gpu)
    sq = xq.index_select(0, idx).contiguous()
    ss = ops.mx_scale_planes(n_seq * 33, D, gpu)
    ss[:, :n_seq * 33] = xs.index_select(1, idx)
    side = torch.empty(n_seq * 33, 3 * D, device=gpu, dtype=torch.bfloat16)
    ops.gemm_mxfp8(sq, ss, wq, ws, b, side)
    assert torch.equal(side, qkv.index_select(0, idx))

    def fused():
        out = torch.full((rows, D), 7.0, device=gpu, dtype=torch.bfloat16)
        part = torch.zeros(n_seq * 12 * 8 * 66, device=gpu)
        ops.qkv_space_attention_mx(xq, xs, wq, ws, b, side, out, part, n_seq=n_seq, scale=0.125)
        return out, part
    out, part = fused()
    for rep in range(3):
        o2, p2 = fused()
        assert torch.equal(o2, out), f'repetition {rep}: {(o2 != out).sum().item()} output elements differ'
        assert torch.equal(p2, part), f'repetition {rep}: {(p2 != part).sum().item()} partial elements differ'
    assert (out.view(n_seq, L, D)[:, 0] == 7.0).all(), 'the fused kernel must not touch the CLS rows'
    o, r = out.float().view(n_seq, L, D), ref.float().view(n_seq, L, D)
    scale = r.abs().max().item()
    torch.testing.assert_close(o[:, 1:], r[:, 1:], rtol=2 ** -6, atol=2 ** -7 * max(1.0, scale))
    assert (o[:, 1:] - r[:, 1:]).abs().gt(1e-3 * max(1.0, scale)).float().mean() < 5e-3
    torch.testing.assert_close(part.view(-1, 66)[:, 2:], part_ref.view(-1, 66)[:, 2:], rtol=2e-2, atol=2e-2 * max(1.0, scale))
    # the MXFP8 output form
    oq, os_ = torch.zeros(rows, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(rows, D, gpu)
    part2 = torch.zeros_like(part)
    ops.qkv_space_attention_mx(xq, xs, wq, ws, b, side, oq, part2, n_seq=n_seq, scale=0.125, out_scales=os_)
    wq_, ws_ = torch.zeros(rows, D, device=gpu, dtype=torch.uint8), ops.mx_scale_planes(rows, D, gpu)
    ops.quantize_mxfp8(out, wq_, ws_)
    patch = torch.ones(n_seq, L, dtype=torch.bool, device=gpu); patch[:, 0] = False
    patch = patch.reshape(-1)
    assert torch.equal(oq[patch], wq_[patch]) and torch.equal(os_[:, :rows][:, patch], ws_[:, :rows][:, patch]) and torch.equal(part2, part)
