"""Deterministic synthetic weights and inputs for the Synchformer hot path.

There is no network for checkpoints or datasets, so parity fixtures, `bench.py` and `smoke()` all run on
random-init weights of the reference's architecture and on synthetic clips.  Values
come from numpy's Philox bit generator keyed by `(seed, crc32(name))`, which is bit-reproducible on any
machine with this numpy version - so the GPU box regenerates exactly the tensors the golden fixtures
were produced from (tests/golden/make_golden.py) without shipping 237 M parameters.

State-dict schema = SURVEY.md §8(b) (513 tensors for configs/sync.yaml); key names, shapes and order are
the reference's (`model/sync_model.py`, `motionformer.py`, `modeling_ast.py`, `modules/transformer.py`).
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

D = 768
FF = 3072


def _enc_layer(prefix):  # nn.TransformerEncoderLayer + cls_token (motionformer.py:275-347)
    return [
        (f'{prefix}.cls_token', (1, 1, D)),
        (f'{prefix}.self_attn.in_proj_weight', (3 * D, D)), (f'{prefix}.self_attn.in_proj_bias', (3 * D,)),
        (f'{prefix}.self_attn.out_proj.weight', (D, D)), (f'{prefix}.self_attn.out_proj.bias', (D,)),
        (f'{prefix}.linear1.weight', (FF, D)), (f'{prefix}.linear1.bias', (FF,)),
        (f'{prefix}.linear2.weight', (D, FF)), (f'{prefix}.linear2.bias', (D,)),
        (f'{prefix}.norm1.weight', (D,)), (f'{prefix}.norm1.bias', (D,)),
        (f'{prefix}.norm2.weight', (D,)), (f'{prefix}.norm2.bias', (D,)),
    ]


def state_dict_schema(n_pos: int = 198, n_out: int = 21, vis_depth: int = 12, aud_depth: int = 12,
                      sync_depth: int = 3, head: str = 'off_head') -> 'OrderedDict[str, tuple]':
    """Ordered name -> shape for `Synchformer.state_dict()` built from configs/sync.yaml.
    `head='sync_head'` gives the GlobalTransformerWithSyncabilityHead variant (sync_model.py:176-190)."""
    s = []
    v = 'vfeat_extractor'
    s += [(f'{v}.cls_token', (1, 1, D)), (f'{v}.pos_embed', (1, 197, D)), (f'{v}.temp_embed', (1, 8, D)),
          (f'{v}.patch_embed.proj.weight', (D, 3, 16, 16)), (f'{v}.patch_embed.proj.bias', (D,)),
          (f'{v}.patch_embed_3d.proj.weight', (D, 3, 2, 16, 16)), (f'{v}.patch_embed_3d.proj.bias', (D,))]
    for i in range(vis_depth):
        b = f'{v}.blocks.{i}'
        s += [(f'{b}.norm1.weight', (D,)), (f'{b}.norm1.bias', (D,))]
        for a in ('attn', 'timeattn'):
            s += [(f'{b}.{a}.qkv.weight', (3 * D, D)), (f'{b}.{a}.qkv.bias', (3 * D,)),
                  (f'{b}.{a}.proj.weight', (D, D)), (f'{b}.{a}.proj.bias', (D,))]
        s += [(f'{b}.norm2.weight', (D,)), (f'{b}.norm2.bias', (D,)),
              (f'{b}.mlp.fc1.weight', (FF, D)), (f'{b}.mlp.fc1.bias', (FF,)),
              (f'{b}.mlp.fc2.weight', (D, FF)), (f'{b}.mlp.fc2.bias', (D,)),
              (f'{b}.norm3.weight', (D,)), (f'{b}.norm3.bias', (D,))]
    s += [(f'{v}.norm.weight', (D,)), (f'{v}.norm.bias', (D,))]
    s += _enc_layer(f'{v}.spatial_attn_agg')
    a = 'afeat_extractor'
    e = f'{a}.ast.embeddings'
    s += [(f'{e}.cls_token', (1, 1, D)), (f'{e}.distillation_token', (1, 1, D)),
          (f'{e}.position_embeddings', (1, 74, D)),
          (f'{e}.patch_embeddings.projection.weight', (D, 1, 16, 16)),
          (f'{e}.patch_embeddings.projection.bias', (D,))]
    for i in range(aud_depth):
        L = f'{a}.ast.encoder.layer.{i}'
        for n in ('query', 'key', 'value'):
            s += [(f'{L}.attention.attention.{n}.weight', (D, D)), (f'{L}.attention.attention.{n}.bias', (D,))]
        s += [(f'{L}.attention.output.dense.weight', (D, D)), (f'{L}.attention.output.dense.bias', (D,)),
              (f'{L}.intermediate.dense.weight', (FF, D)), (f'{L}.intermediate.dense.bias', (FF,)),
              (f'{L}.output.dense.weight', (D, FF)), (f'{L}.output.dense.bias', (D,)),
              (f'{L}.layernorm_before.weight', (D,)), (f'{L}.layernorm_before.bias', (D,)),
              (f'{L}.layernorm_after.weight', (D,)), (f'{L}.layernorm_after.bias', (D,))]
    s += [(f'{a}.ast.layernorm.weight', (D,)), (f'{a}.ast.layernorm.bias', (D,))]
    s += _enc_layer(f'{a}.freq_attn_agg')
    s += [('vproj.weight', (D, D)), ('vproj.bias', (D,)), ('aproj.weight', (D, D)), ('aproj.bias', (D,))]
    t = 'transformer'
    s += [(f'{t}.OFF_tok', (1, 1, D)), (f'{t}.MOD_tok', (1, 1, D)),
          (f'{t}.vis_in_lnorm.weight', (D,)), (f'{t}.vis_in_lnorm.bias', (D,)),
          (f'{t}.aud_in_lnorm.weight', (D,)), (f'{t}.aud_in_lnorm.bias', (D,)),
          (f'{t}.pos_emb_cfg.pos_emb', (1, n_pos, D))]
    for i in range(sync_depth):
        b = f'{t}.blocks.{i}'
        s += [(f'{b}.ln1.weight', (D,)), (f'{b}.ln1.bias', (D,)), (f'{b}.ln2.weight', (D,)), (f'{b}.ln2.bias', (D,))]
        for n in ('key', 'query', 'value', 'proj'):
            s += [(f'{b}.attn.{n}.weight', (D, D)), (f'{b}.attn.{n}.bias', (D,))]
        s += [(f'{b}.mlp.0.weight', (FF, D)), (f'{b}.mlp.0.bias', (FF,)),
              (f'{b}.mlp.2.weight', (D, FF)), (f'{b}.mlp.2.bias', (D,))]
    s += [(f'{t}.ln_f.weight', (D,)), (f'{t}.ln_f.bias', (D,)),
          (f'{t}.{head}.weight', (n_out, D)), (f'{t}.{head}.bias', (n_out,))]
    return OrderedDict(s)


_NORM_KEYS = ('norm', 'lnorm', '.ln1.', '.ln2.', '.ln_f.', 'layernorm')


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def fill_tensor(name: str, shape, seed: int, gain: float = 1.0) -> torch.Tensor:
    """One tensor of the synthetic checkpoint (fp32).  Everything is non-degenerate on purpose: biases and
    `patch_embed_3d.proj.weight` are non-zero (the reference zero-inits the latter, vmb:61, which would
    make the visual branch input-independent - BASELINE.md §3), LayerNorm gains are 1 +- 0.1."""
    n = _rng(seed, name).standard_normal(size=tuple(shape), dtype=np.float32)
    is_norm = any(k in name for k in _NORM_KEYS)
    if is_norm and name.endswith('weight'):
        arr = 1.0 + 0.1 * n
    elif name.endswith(('OFF_tok', 'MOD_tok', 'pos_emb_cfg.pos_emb')):
        arr = n  # torch.randn init in the reference (sync_model.py:129-130, transformer.py:127)
    elif name.endswith('weight') and len(shape) >= 2:
        arr = (0.02 * gain) * n
    else:
        arr = 0.02 * n  # biases, tokens, position tables
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))


def make_state_dict(seed: int = 1337, gain: float = 1.0, **schema_kw) -> 'OrderedDict[str, torch.Tensor]':
    return OrderedDict((k, fill_tensor(k, shp, seed, gain)) for k, shp in state_dict_schema(**schema_kw).items())


def make_video_u8(B: int, S: int = 14, seed: int = 1337, T: int = 16, H: int = 224, W: int = 224) -> torch.Tensor:
    """(B, S, T, 3, H, W) uint8 frames, U{0..255} (SURVEY §8d synthetic input)."""
    g = _rng(seed, f'video{B}x{S}')
    return torch.from_numpy(g.integers(0, 256, size=(B, S, T, 3, H, W), dtype=np.uint8))


def make_wave(B: int, S: int = 14, seed: int = 1337, n: int = 10240) -> torch.Tensor:
    """(B, S, n) fp32 waveform segments U(-1, 1): 0.64 s @ 16 kHz (transforms.py:431)."""
    g = _rng(seed, f'wave{B}x{S}')
    return torch.from_numpy((g.random(size=(B, S, n), dtype=np.float32) * 2.0 - 1.0).astype(np.float32))


def make_spectrogram(B: int, S: int = 14, seed: int = 1337, F: int = 128, Ta: int = 66) -> torch.Tensor:
    """(B, S, 1, F, Ta) fp32 'already normalised' log-mel: N(0, 0.5^2) (SURVEY §8d alternative input)."""
    g = _rng(seed, f'spec{B}x{S}')
    return torch.from_numpy(0.5 * g.standard_normal(size=(B, S, 1, F, Ta), dtype=np.float32))


def make_structured_clip(c: int, S: int = 14, seed: int = 1337, T: int = 16, H: int = 224, W: int = 224, F: int = 128, Ta: int = 66):
    """Clip `c` of a synthetic evaluation set whose clips DIFFER in structure (the U{0..255} noise clips of make_video_u8 are statistically identical, so
    every clip lands on nearly the same logits): per clip a patch-level colour field (14 x 14 blocks of 16 x 16 pixels) that drifts over the frames at
    clip-specific rates, under pixel noise of a clip-specific amplitude; the spectrogram carries clip-specific ridges (a band of mel bins pulsing with a
    clip-specific period and phase) over noise.  Video values come from integer arithmetic on Philox draws only, the spectrogram from float32 adds /
    multiplies of Philox normals: bit-reproducible wherever numpy is (the fixtures hold the reference's outputs for exactly these tensors).
    -> (vis (S, T, 3, H, W) uint8, aud (S, 1, F, Ta) fp32)."""
    g = _rng(seed, f'sclip{c}')
    gh, gw = H // 16, W // 16
    amp = int(g.integers(8, 97))
    base = g.integers(0, 256, size=(3, gh, gw), dtype=np.int64)
    drift = g.integers(-6, 7, size=(3, gh, gw), dtype=np.int64)
    seg_step = int(g.integers(1, 9))
    t_idx = (np.arange(S, dtype=np.int64)[:, None] * 8 + np.arange(T, dtype=np.int64)[None, :])                 # absolute frame index (50 % overlapping segments)
    field = (base[None, None] + drift[None, None] * (t_idx[:, :, None, None, None] * seg_step // 4)) % 256      # (S, T, 3, gh, gw)
    field = np.repeat(np.repeat(field.astype(np.int16), 16, axis=3), 16, axis=4)
    noise = g.integers(-amp, amp + 1, size=(S, T, 3, H, W), dtype=np.int16)
    noise += field
    vis = np.clip(noise, 0, 255, out=noise).astype(np.uint8)
    spec = (0.35 * g.standard_normal(size=(S, 1, F, Ta), dtype=np.float32)).astype(np.float32)
    f0, bw = int(g.integers(0, F - 24)), int(g.integers(6, 24))
    period, phase = int(g.integers(3, 17)), int(g.integers(0, 16))
    level = np.float32(0.5 + 0.125 * int(g.integers(0, 9)))
    cols = ((np.arange(S, dtype=np.int64)[:, None] * 32 + np.arange(Ta, dtype=np.int64)[None, :] + phase) % period) < max(1, period // 3)   # (S, Ta)
    spec[:, 0, f0:f0 + bw, :] += level * cols[:, None, :].astype(np.float32)
    return torch.from_numpy(vis), torch.from_numpy(spec)


def make_structured_clips(c0: int, n: int, S: int = 14, seed: int = 1337):
    """Clips c0 .. c0 + n - 1 of the structured evaluation set: (vis (n, S, 16, 3, 224, 224) uint8, aud (n, S, 1, 128, 66) fp32)."""
    clips = [make_structured_clip(c, S, seed) for c in range(c0, c0 + n)]
    return torch.stack([v for v, _ in clips]), torch.stack([a for _, a in clips])


def make_masks(B: int, S: int = 14, seed: int = 1337, T: int = 16, H: int = 224, W: int = 224, F: int = 128, Ta: int = 66):
    """Deterministic content masks for Synchformer.forward(vis_mask=, aud_mask=) (True = kept): per segment a few boxes that are NOT
    aligned to the 16-pixel patch grid and span some frames (all channels), plus isolated single elements - those exercise the
    reference's NaN-trick quirk (one masked element meets one filter weight: +-inf, not NaN, so the token stays).
    -> vis_mask (B, S, T, 3, H, W) bool, aud_mask (B, S, 1, F, Ta) bool."""
    g = _rng(seed, f'masks{B}x{S}')
    vm = np.ones((B, S, T, 3, H, W), dtype=bool)
    am = np.ones((B, S, 1, F, Ta), dtype=bool)
    for b in range(B):
        for s in range(S):
            for _ in range(3):
                t0, t1 = sorted(g.integers(0, T + 1, size=2))
                y0, x0 = g.integers(0, H - 40), g.integers(0, W - 40)
                h, w = g.integers(8, 90), g.integers(8, 90)
                vm[b, s, t0:max(t1, t0 + 1), :, y0:y0 + h, x0:x0 + w] = False
            for _ in range(4):                                            # isolated pixels (single channel, single frame)
                vm[b, s, g.integers(0, T), g.integers(0, 3), g.integers(0, H), g.integers(0, W)] = False
            for _ in range(2):
                f0, a0 = g.integers(0, F - 20), g.integers(0, Ta - 10)
                am[b, s, 0, f0:f0 + g.integers(4, 40), a0:a0 + g.integers(3, 25)] = False
            for _ in range(3):
                am[b, s, 0, g.integers(0, F), g.integers(0, Ta)] = False
    return torch.from_numpy(vm), torch.from_numpy(am)


def make_masks_fused_case(B: int = 1, S: int = 6, seed: int = 77):
    """The mask set of the fused-schedule token-mask parity case (tests/golden/e2e_masked_B1S6.npz, tests/test_e2e_gpu.py): `make_masks` plus, in clip 0,
    segment 1: video frames 4-5 = one WHOLE token frame (all 196 keys of a space group masked for nobody, every time group loses one key);
    segment 2: patches 0-3 in all 8 token frames = one whole wave of sf_qkv_time_attention2 (a CLS partial record with every key masked: m = -inf, l = 0);
    segment 3: pixel columns 32-47 in every frame = one whole PATCH COLUMN (14 patches x 8 frames: 14 complete time groups masked, one key in 14 of every space group);
    segment 4: patches 192-195 in all frames = the left-over rows both fused launches take from the side GEMM."""
    vm, am = make_masks(B, S, seed)
    vm[0, 1, 4:6] = False
    vm[0, 2, :, :, 0:16, 0:64] = False
    if S > 3:
        vm[0, 3, :, :, :, 32:48] = False
    if S > 4:
        vm[0, 4, :, :, 208:224, 160:224] = False
    return vm, am


def make_targets(B: int, n_cls: int = 21, seed: int = 1337) -> torch.Tensor:
    g = _rng(seed, f'targets{B}')
    return torch.from_numpy(g.integers(0, n_cls, size=(B,), dtype=np.int64))
