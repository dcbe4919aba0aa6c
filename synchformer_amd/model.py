"""Drop-in host mirror of the reference's plugin interface for the hot path.

The reference builds its model with `instantiate_from_config({'target': 'pkg.mod.Class', 'params': {...}})`
(utils/utils.py:78-88) from configs/sync.yaml.  The classes below keep the reference's constructor signatures,
attribute names (`vfeat_extractor`, `afeat_extractor`, `vproj`, `aproj`, `transformer.pos_emb_cfg.pos_emb`, ...) and
the exact `state_dict()` schema (513 tensors for configs/sync.yaml), so `get_model`, `toggle_mode`,
`load_state_dict(ckpt['model'])` and `model(vis, aud, targets)` in scripts/train_sync.py / example.py work unchanged once
`target:` points here - or, without editing any yaml, after `synchformer_amd.install_reference_aliases()` has
registered these classes under the reference's own dotted paths.

All compute runs through libsynchformer_hip (synchformer_amd.engine); there is no CPU / eager fallback: calling
forward on CPU tensors raises.  Out-of-scope options of the reference constructors (SparseSync-era bridges, S3D/ResNet
extractors, joint/trajectory attention - SURVEY §2 rows 4b/5/6) raise NotImplementedError
naming the option.
"""
import contextlib
import importlib
import logging
import os
import sys
import types
from typing import Any, Mapping, Optional

import torch

from . import ops, synth
from .engine import SynchformerEngine

# reference dotted path -> class name in this module (nested `target:` strings of configs/*.yaml)
_TARGET_ALIASES = {
    'model.sync_model.Synchformer': 'Synchformer',
    'model.sync_model.GlobalTransformer': 'GlobalTransformer',
    'model.sync_model.GlobalTransformerWithSyncabilityHead': 'GlobalTransformerWithSyncabilityHead',
    'model.modules.feat_extractors.visual.motionformer.MotionFormer': 'MotionFormer',
    'model.modules.feat_extractors.audio.ast.AST': 'AST',
    'model.modules.transformer.RandInitPositionalEncoding': 'RandInitPositionalEncoding',
    'model.modules.bridges.DoNothingBridge': 'DoNothingBridge',
    'model.modules.feat_extractors.train_clip_src.open_clip.model.AVCLIP': 'AVCLIP',
}


def get_obj_from_str(string: str):
    """utils/utils.py:78-83, with the reference's hot-path classes resolved to this package."""
    if string in _TARGET_ALIASES:
        return globals()[_TARGET_ALIASES[string]]
    module, cls = string.rsplit('.', 1)
    return getattr(importlib.import_module(module), cls)


def instantiate_from_config(config: Mapping[str, Any]):
    """utils/utils.py:85-88 (same KeyError on a missing `target`)."""
    if 'target' not in config:
        raise KeyError('Expected key `target` to instantiate.')
    return get_obj_from_str(config['target'])(**config.get('params', dict()))


def install_reference_aliases():
    """Register this module under the reference's dotted module paths (`model.sync_model`, ...), so an UNMODIFIED
    configs/sync.yaml instantiates the HIP-backed classes (SURVEY §8b 'registering itself under the same dotted path')."""
    me = sys.modules[__name__]
    for path in {p.rsplit('.', 1)[0] for p in _TARGET_ALIASES}:
        parts = path.split('.')
        for i in range(1, len(parts) + 1):
            name = '.'.join(parts[:i])
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
            if i > 1:
                setattr(sys.modules['.'.join(parts[:i - 1])], parts[i - 1], sys.modules[name])
        for ref_path, cls in _TARGET_ALIASES.items():
            if ref_path.rsplit('.', 1)[0] == path:
                setattr(sys.modules[path], ref_path.rsplit('.', 1)[1], getattr(me, cls))


def uninstall_reference_aliases():
    """Remove the alias modules again (they have no __file__), e.g. before importing the real reference in tests."""
    for name in [n for n, m in sys.modules.items()
                 if (n == 'model' or n.startswith('model.')) and getattr(m, '__file__', None) is None
                 and not hasattr(m, '__path__')]:
        del sys.modules[name]


_NORM_PARTS = ('norm', 'ln1', 'ln2', 'ln_f', 'lnorm', 'layernorm')


def _reference_init(full: str, shape) -> torch.Tensor:
    """One freshly initialised tensor by the reference's rules, drawn from torch's global RNG (so torch.manual_seed governs it):
    sync transformer - Linear N(0, 0.02), zero biases, LayerNorm 1 / 0 (sync_model.py:13-20), OFF/MOD tokens and the position table
    torch.randn (sync_model.py:129-130, modules/transformer.py:127); feature extractors - Linear / conv trunc_normal(std 0.02), zero
    biases, LayerNorm 1 / 0, trunc_normal(0.02) cls / position tables (video_model_builder.py:71-83,151-158, motionformer.py:288-299,
    336-343, modeling_ast.py:397-409), zeros for `temp_embed` and the AST tokens / position table (vmb:83, modeling_ast.py:65-71).
    Differences kept on purpose: nn.MultiheadAttention's xavier in_proj of the aggregators is drawn as trunc_normal(0.02) like every other
    Linear there (motionformer.py:299 re-initialises the nn.Linear members only), and `patch_embed_3d.proj` is not zero-initialised
    (vmb:61 zeroes it, which makes a from-scratch visual tower input-independent; every shipped config loads it from a checkpoint)."""
    leaf = full.rsplit('.', 1)[-1]
    t = torch.empty(tuple(shape), dtype=torch.float32)
    in_sync = full.startswith('transformer.')
    is_norm = any(p in part for part in full.split('.')[:-1] for p in _NORM_PARTS) and leaf in ('weight', 'bias')
    if is_norm:
        return t.fill_(1.0 if leaf == 'weight' else 0.0)
    if leaf in ('OFF_tok', 'MOD_tok') or full.endswith('pos_emb_cfg.pos_emb'):
        return t.normal_()
    if leaf in ('temp_embed', 'distillation_token', 'position_embeddings') or (leaf == 'cls_token' and '.ast.' in full):
        return t.zero_()
    if leaf in ('cls_token', 'pos_embed', 'pos_emb'):
        return torch.nn.init.trunc_normal_(t, std=0.02)
    if leaf.endswith('bias'):
        return t.zero_()
    if len(shape) >= 2:
        return t.normal_(0.0, 0.02) if in_sync else torch.nn.init.trunc_normal_(t, std=0.02)
    return t.zero_()


def _register_tree(root: torch.nn.Module, schema: Mapping[str, tuple], prefix: str, seed: Optional[int]):
    """Create nested containers + nn.Parameters so that root.state_dict() has exactly the reference's keys.  seed None (default):
    the reference's initialisation rules on torch's RNG; an int: the deterministic synthetic fill of synchformer_amd.synth (tests, benches)."""
    for full, shape in schema.items():
        if not full.startswith(prefix):
            continue
        parts = full[len(prefix):].split('.')
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, torch.nn.Module())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], torch.nn.Parameter(_reference_init(full, shape) if seed is None else synth.fill_tensor(full, shape, seed)))


class DoNothingBridge(torch.nn.Identity):
    """model/modules/bridges.py:64-68."""

    def __init__(self, in_features=None, out_features=None, **_):
        super().__init__()


class RandInitPositionalEncoding(torch.nn.Module):
    """model/modules/transformer.py:120-130 (parameter holder; the add is folded into the engine's token table)."""

    def __init__(self, block_shape: list, n_embd: int):
        super().__init__()
        self.block_shape, self.n_embd = block_shape, n_embd
        self.pos_emb = torch.nn.Parameter(torch.randn(1, *block_shape, n_embd))

    def forward(self, token_embeddings):
        return token_embeddings + self.pos_emb


def _parse_agg_time(agg_time_module) -> bool:
    """True for 'AveragePooling' (Stage-1, configs/segment_avclip.yaml:21,33), False for torch.nn.Identity (Stage-2)."""
    if agg_time_module == 'AveragePooling':
        return True
    if agg_time_module is not None and 'Identity' in agg_time_module:
        return False
    raise NotImplementedError(f"agg_time_module={agg_time_module!r}: 'AveragePooling' and 'torch.nn.Identity' are built natively; "
                              "the temporal TransformerEncoderLayer aggregator is not used by the shipped configs")


def _no_extractor_backward(module: torch.nn.Module):
    """The extractors have a forward only (Stage-1 fine-tuning of the towers needs their backward: SURVEY §8 a22/a24, next)."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(f'{type(module).__name__}: no backward through the HIP feature extractors yet - call under '
                                  'torch.no_grad() or requires_grad_(False) the module (scripts/train_utils.py:199-204)')


class MotionFormer(torch.nn.Module):
    """motionformer.py:24-272 (divided space-time, factorised spatial aggregation).  Parameter holder + standalone
    feature extractor; inside `Synchformer` the engine reads these parameters directly."""

    def __init__(self, extract_features: bool = False, ckpt_path: str = None, factorize_space_time: bool = None,
                 agg_space_module: str = None, agg_time_module: str = None, add_global_repr: bool = True,
                 agg_segments_module: str = None, max_segments: int = None, _seed: Optional[int] = None):
        super().__init__()
        if not extract_features or not factorize_space_time or agg_space_module != 'TransformerEncoderLayer':
            raise NotImplementedError('only extract_features=True, factorize_space_time=True, '
                                      "agg_space_module='TransformerEncoderLayer' (configs/sync.yaml) is built natively")
        self.pool_time = _parse_agg_time(agg_time_module)
        if add_global_repr:
            raise NotImplementedError('add_global_repr=True (global segment aggregation) is outside the hot path')
        self.ckpt_path = ckpt_path
        self.extract_features, self.factorize_space_time, self.add_global_repr = True, True, False
        self.embed_dim, self.num_heads, self.temporal_resolution = 768, 12, 8
        self.drop_path_rate = 0.2                       # VIT.DROP_PATH of divided_224_16x4.yaml:59 (train mode only; video_model_builder.py:33,86-87)
        schema = synth.state_dict_schema()
        _register_tree(self, schema, 'vfeat_extractor.', _seed)
        if ckpt_path is not None:                       # ssv2_divided_224_16x4.pyth or a Stage-1 *.pt (motionformer.py:52-80, 109-116, 156-173)
            from .checkpoint import init_motionformer
            init_motionformer(self, ckpt_path)
        self.patch_embed.requires_grad_(False)          # motionformer.py:177
        logging.info(f'vfeat_extractor: {sum(p.numel() for p in self.parameters() if p.requires_grad):,}')

    def forward(self, x, for_loop: bool = False, cont_mask: torch.Tensor = None):
        """x (B, S, C, T, H, W) -> ((B, S, 8, 768), None), or ((B, S, 768), None) with agg_time_module='AveragePooling'
        (motionformer.py:182-223)."""
        if cont_mask is not None and for_loop:
            raise AssertionError('cont_mask is not supported with for_loop=True')       # motionformer.py:201
        _no_extractor_backward(self)
        eng = _engine_for(self, 'vfeat_extractor.')
        feat = eng.extract_vfeats(x.permute(0, 1, 3, 2, 4, 5).contiguous(),
                                  None if cont_mask is None else cont_mask.permute(0, 1, 3, 2, 4, 5).contiguous())
        if self.pool_time:
            feat = eng.pool_segments(feat).view(feat.shape[0], feat.shape[1], -1)
        return feat, None


class AST(torch.nn.Module):
    """ast.py:13-250 (factorised frequency aggregation)."""

    def __init__(self, extract_features: bool = False, ckpt_path: str = None, feat_type: str = None,
                 max_spec_t: int = None, factorize_freq_time: bool = None, agg_freq_module: str = None,
                 agg_time_module: str = None, add_global_repr: bool = True, agg_segments_module: str = None,
                 max_segments: int = None, _seed: Optional[int] = None):
        super().__init__()
        if not extract_features or not factorize_freq_time or agg_freq_module != 'TransformerEncoderLayer':
            raise NotImplementedError('only extract_features=True, factorize_freq_time=True, '
                                      "agg_freq_module='TransformerEncoderLayer' (configs/sync.yaml) is built natively")
        self.pool_time = _parse_agg_time(agg_time_module)
        if add_global_repr:
            raise NotImplementedError('add_global_repr=True (global segment aggregation) is outside the hot path')
        self.ckpt_path = ckpt_path
        if max_spec_t != 66:
            raise NotImplementedError('max_spec_t must be 66 (74-token position table, configs/sync.yaml:14)')
        self.extract_features, self.max_spec_t, self.factorize_freq_time, self.add_global_repr = True, 66, True, False
        self.feat_type = 'last_hidden_state'
        _register_tree(self, synth.state_dict_schema(), 'afeat_extractor.', _seed)
        if ckpt_path is not None:                       # HF AST weights (local) or a Stage-1 *.pt (ast.py:49-53, 113-131, 240-245)
            from .checkpoint import init_ast
            init_ast(self, ckpt_path)

    def forward(self, x, for_loop: bool = False, cont_mask: torch.Tensor = None, **ast_kwargs):
        """x (B, S, T, F) -> ((B, S, 6, 768), None), or ((B, S, 768), None) with agg_time_module='AveragePooling'
        (ast.py:137-176)."""
        if cont_mask is not None and for_loop:
            raise AssertionError('cont_mask is not supported with for_loop=True')       # ast.py:153
        _no_extractor_backward(self)
        eng = _engine_for(self, 'afeat_extractor.')
        B, S, T, Fq = x.shape
        feat = eng.extract_afeats(x.permute(0, 1, 3, 2).reshape(B, S, 1, Fq, T),
                                  None if cont_mask is None else cont_mask.permute(0, 1, 3, 2).reshape(B, S, 1, Fq, T))
        if self.pool_time:
            feat = eng.pool_segments(feat).view(B, S, -1)
        return feat, None


class GlobalTransformer(torch.nn.Module):
    """sync_model.py:117-173.  Dropouts are inference-only no-ops here (the reference runs eval() for inference)."""
    _head_name = 'off_head'

    def __init__(self, tok_pdrop, embd_pdrop, resid_pdrop, attn_pdrop, n_layer, n_head, n_embd, pos_emb_cfg=None,
                 off_head_cfg=None, _seed: Optional[int] = None):
        super().__init__()
        if n_embd != 768 or n_head != 8:
            raise NotImplementedError('native sync transformer is built for n_embd=768, n_head=8 (configs/sync.yaml:44-46)')
        # whole-token dropout (sync_model.py:131-134, 160-161) acts in train() mode only: the value is kept, evaluation ignores it like the reference's
        # Dropout1d in eval(); the TRAINING forward of Synchformer applies it (train.SyncTrainer.tok_pdrop: sf_scale_rows_map on the input norms' rows and on their gradient)
        self.tok_pdrop = float(tok_pdrop or 0.0)
        self.n_layer, self.n_head, self.n_embd = n_layer, n_head, n_embd
        self.pdrops = dict(embd_pdrop=embd_pdrop, resid_pdrop=resid_pdrop, attn_pdrop=attn_pdrop)
        n_pos = pos_emb_cfg['params']['block_shape'][0] if pos_emb_cfg is not None else 198
        n_out = off_head_cfg['params']['out_features'] if off_head_cfg is not None else 21
        schema = synth.state_dict_schema(n_pos=n_pos, n_out=n_out, sync_depth=n_layer, head='off_head')
        skip = ('transformer.pos_emb_cfg.', 'transformer.off_head.')
        _register_tree(self, {k: v for k, v in schema.items() if not k.startswith(skip)}, 'transformer.', _seed)
        if pos_emb_cfg is not None:
            self.pos_emb_cfg = instantiate_from_config(pos_emb_cfg)      # torch.randn, modules/transformer.py:127
            if _seed is not None:
                with torch.no_grad():
                    self.pos_emb_cfg.pos_emb.copy_(synth.fill_tensor('transformer.pos_emb_cfg.pos_emb', (1, n_pos, n_embd), _seed))
        if off_head_cfg is not None:
            self.off_head = instantiate_from_config(off_head_cfg)
            with torch.no_grad():                                         # self.apply(init_weights), sync_model.py:147
                for nm, p_ in self.off_head.named_parameters():
                    key = 'transformer.off_head.' + nm
                    p_.copy_(_reference_init(key, p_.shape) if _seed is None else synth.fill_tensor(key, p_.shape, _seed))
        _reorder_like(self, [k[len('transformer.'):] for k in schema])

    def forward(self, v: torch.Tensor, a: torch.Tensor, targets=None, attempt_to_apply_heads=True):
        """sync_model.py:150-173: logits (B, n_out) - or, with attempt_to_apply_heads=False, ln_f of ALL tokens (B, 2 + Sv + Sa, 768), which is what the
        reference's subclass asks its parent for (sync_model.py:187)."""
        if self.training and self.tok_pdrop > 0:                          # (the reference's Dropout1d drops tokens in train() mode whatever the grad mode, sync_model.py:131-134,160-161)
            raise NotImplementedError('a STANDALONE GlobalTransformer.forward in train() mode runs the inference schedule (no dropout of any kind); whole-token dropout is '
                                      'applied where training happens - Synchformer.forward with grad enabled (train.SyncTrainer, tok_pdrop) - or call .eval() here')
        eng = _engine_for(self, 'transformer.')
        B = v.shape[0]
        return eng.global_transformer(v.reshape(B, -1, self.n_embd).float(), a.reshape(B, -1, self.n_embd).float(), apply_head=bool(attempt_to_apply_heads))


class GlobalTransformerWithSyncabilityHead(GlobalTransformer):
    """sync_model.py:176-190: off_head -> Identity, 2-way sync_head on token 0."""

    def __init__(self, tok_pdrop, embd_pdrop, resid_pdrop, attn_pdrop, n_layer, n_head, n_embd, pos_emb_cfg=None,
                 off_head_cfg=None, _seed: Optional[int] = None):
        super().__init__(tok_pdrop, embd_pdrop, resid_pdrop, attn_pdrop, n_layer, n_head, n_embd, pos_emb_cfg, off_head_cfg,
                         _seed)
        self.off_head = torch.nn.Identity()
        self.sync_head = torch.nn.Linear(n_embd, 2)
        with torch.no_grad():                                             # self.apply(init_weights), sync_model.py:185
            for nm, p_ in self.sync_head.named_parameters():
                key = 'transformer.sync_head.' + nm
                p_.copy_(_reference_init(key, p_.shape) if _seed is None else synth.fill_tensor(key, p_.shape, _seed))


class _CrossEntropyFunction(torch.autograd.Function):
    """F.cross_entropy(logits, targets) (mean reduction) on sf_cross_entropy: loss and dlogits come out of one launch."""

    @staticmethod
    def forward(ctx, logits, targets):
        from . import ops
        z = logits.detach().float().contiguous()
        loss = torch.empty(1, device=z.device, dtype=torch.float32)
        dz = torch.empty_like(z)
        ops.cross_entropy(z, targets.to(torch.int64).contiguous(), loss, dz)
        ctx.save_for_backward(dz)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None


def _cross_entropy(logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    if targets.dtype not in (torch.int64, torch.int32) or targets.dim() != 1:
        raise NotImplementedError('cross_entropy: class-index targets of shape (B,) only (scripts/train_sync.py:171)')
    return _CrossEntropyFunction.apply(logits, targets)


def _reorder_like(mod: torch.nn.Module, order):
    """state_dict() order follows registration order; make it the reference's (cosmetic: strict loading is by key)."""
    want = [k.split('.')[0] for k in order]
    seen, first = set(), []
    for k in want:
        if k not in seen:
            seen.add(k)
            first.append(k)
    params = dict(mod._parameters)
    mods = dict(mod._modules)
    mod._parameters.clear()
    mod._modules.clear()
    for k in first:
        if k in params:
            mod._parameters[k] = params.pop(k)
        elif k in mods:
            mod._modules[k] = mods.pop(k)
    mod._parameters.update(params)
    mod._modules.update(mods)


def _param_key(module: torch.nn.Module):
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


def _engine_for(module: torch.nn.Module, prefix: str) -> SynchformerEngine:
    """Stand-alone use of a sub-module: build (and cache) an engine that only has this sub-module's weights real."""
    key = _param_key(module)
    cached = getattr(module, '_sf_engine', None)
    if cached is not None and cached[0] == key:
        return cached[1]
    dev = next(module.parameters()).device
    if dev.type != 'cuda':
        raise RuntimeError('synchformer_amd modules compute on a HIP device only (no CPU fallback); call .to("cuda") first')
    n_pos, n_out, head = 198, 21, 'off_head'
    own = {prefix + k: v for k, v in module.state_dict().items()}
    if prefix == 'transformer.':
        n_pos = own['transformer.pos_emb_cfg.pos_emb'].shape[1]
        head = 'sync_head' if 'transformer.sync_head.weight' in own else 'off_head'
        n_out = own[f'transformer.{head}.weight'].shape[0]
    sd = synth.make_state_dict(0, n_pos=n_pos, n_out=n_out, head=head, sync_depth=len([k for k in own if k.endswith('ln1.weight')]) or 3)
    sd.update(own)
    eng = SynchformerEngine(sd, dev)
    object.__setattr__(module, '_sf_engine', (key, eng))
    return eng


class Synchformer(torch.nn.Module):
    """model/sync_model.py:23-114 - same constructor, forward contract and state-dict schema."""
    dispatcher_route = os.environ.get('SF_DISPATCHER', '1') != '0'      # forward() launches through torch.ops.synchformer.* (see forward)

    def __init__(self, afeat_extractor, vfeat_extractor, aproj, vproj, transformer):
        super().__init__()
        self.vfeat_extractor = instantiate_from_config(vfeat_extractor)
        self.afeat_extractor = instantiate_from_config(afeat_extractor)
        self.vproj = instantiate_from_config(vproj)
        self.aproj = instantiate_from_config(aproj)
        self.transformer = instantiate_from_config(transformer)
        for name in ('vproj', 'aproj'):
            m = getattr(self, name)
            if not isinstance(m, torch.nn.Linear) or m.in_features != 768 or m.out_features != 768:
                raise NotImplementedError(f'{name}: only torch.nn.Linear(768, 768) (configs/sync.yaml:28-39) is built natively')
            # nn.Linear's own default initialisation stays, as in the reference (sync_model.py:33-34 never re-initialises the bridges)
        self._sf_engine = None
        self.seg_chunk = 224

    # -- engine cache ---------------------------------------------------------------------------------------
    def _split_keys(self):
        # (storage, version) of every parameter, extractor side and sync side.  The parameter LISTS are cached (walking named_parameters() of ~700 tensors
        # three to four times per forward cost a few ms of host time per training step) as (owner module, name, Parameter) and VERIFIED on every call with one
        # flat identity loop: a Parameter OBJECT that was replaced - load_state_dict(assign=True), `m.weight = nn.Parameter(..)`, parametrize / prune,
        # set_overwrite_module_params_on_conversion - is no longer what its owner holds, so the lists are rebuilt and the engine sees the new storage.
        lists = self.__dict__.get('_sf_param_lists')
        if lists is not None:
            for group in lists:
                for owner, pname, p in group:
                    if owner._parameters.get(pname) is not p:
                        lists = None
                        break
                if lists is None:
                    break
        if lists is None:
            owners = dict(self.named_modules())
            frozen, sync = [], []
            for n, p in self.named_parameters():
                mod, _, pname = n.rpartition('.')
                (sync if n.startswith(('vproj.', 'aproj.', 'transformer.')) else frozen).append((owners[mod], pname, p))
            lists = self.__dict__['_sf_param_lists'] = (frozen, sync)
        return (tuple((p.data_ptr(), p._version) for _, _, p in lists[0]), tuple((p.data_ptr(), p._version) for _, _, p in lists[1]))

    def _engine(self, need_sync: bool = True) -> SynchformerEngine:
        """The engine is keyed on the EXTRACTOR parameters only (214.8M weights, the multi-GB workspaces): an optimizer step on
        vproj / aproj / the sync transformer (Stage-2 training, train_utils.py:199-204) refreshes just those 22.6M operand copies in
        place, and only when the inference path (`need_sync`) is about to read them - the train step has its own copies."""
        k_frozen, k_sync = self._split_keys()
        if self._sf_engine is not None and self._sf_engine[0] == k_frozen:
            eng = self._sf_engine[1]
            if need_sync and self._sf_engine[2] != k_sync:
                eng.load_sync_weights({k: v for k, v in self.state_dict().items() if k.startswith(('vproj.', 'aproj.', 'transformer.'))})
                self._sf_engine = (k_frozen, eng, k_sync)
            return eng
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('Synchformer computes on a HIP device only (no CPU fallback); call .to("cuda") first')
        eng = SynchformerEngine(self.state_dict(), dev, seg_chunk=self.seg_chunk)
        self._sf_engine = (k_frozen, eng, k_sync)
        return eng

    # -- reference API --------------------------------------------------------------------------------------
    def forward(self, vis: torch.Tensor, aud: torch.Tensor, targets: torch.Tensor = None, for_loop=False,
                vis_mask: torch.Tensor = None, aud_mask: torch.Tensor = None, loss_fn=None):
        """vis (B, S, Tv, C, H, W) u8|f16|bf16|f32, aud (B, S, 1, F, Ta) -> (loss | None, logits)  (sync_model.py:38-70).
        `for_loop` only trades memory for speed in the reference (bit-equal results); here segments are always
        processed in `self.seg_chunk`-sized chunks."""
        if aud_mask is not None and for_loop:
            raise AssertionError('cont_mask is not supported with for_loop=True')       # ast.py:153
        vis_in = vis
        # "registered as PyTorch-ROCm custom ops" (SURVEY 8b(1)): the drop-in module's launches go through the PyTorch dispatcher - torch.ops.synchformer.*, visible
        # to profilers / dispatch modes / torch.library consumers - unless `dispatcher_route` is switched off.  Measured cost (bench.py `workloads.dispatcher_route`):
        # -0.1 % at 16 clips (noise), +0.5 % on a single clip; the engine's own loops (bench.py, SyncTrainer) keep the direct ctypes path.
        route = ops.via_dispatcher() if self.dispatcher_route else contextlib.nullcontext()
        with route:
            # the two extractors are independent (sync_model.py:45-52): the audio tower runs on a second stream next to the visual one
            trainable = self._trainable_params()
            training = torch.is_grad_enabled() and any(p.requires_grad for p in trainable.values())
            vis, aud = self._engine(need_sync=not training).both_towers(lambda: self.extract_vfeats(vis_in, for_loop, vis_mask=vis_mask), aud, aud_mask)
            if training:
                logits = self._train_forward(vis, aud, trainable)
            else:
                logits = self._engine().sync_transformer(vis, aud)
        return self.compute_loss(logits, targets, loss_fn), logits

    # -- Stage-2 training (extractors frozen, train_utils.py:199-204) ---------------------------------------
    def _trainable_params(self):
        return {n: p for n, p in self.named_parameters() if n.startswith(('vproj.', 'aproj.', 'transformer.'))}

    def _train_forward(self, vfeat, afeat, trainable):
        from .train import SyncTrainer, SyncTrainFunction
        if any(p.requires_grad for n, p in self.named_parameters() if n.startswith(('vfeat_extractor.', 'afeat_extractor.'))
               and not n.startswith('vfeat_extractor.patch_embed.')):
            raise NotImplementedError('only Stage-2 training with frozen extractors (is_trainable: False, configs/sync.yaml:7,19) has a '
                                      'backward; requires_grad_(False) the extractors as scripts/train_utils.py:199-204 does')
        pd = getattr(self.transformer, 'pdrops', {}) if self.transformer.training else {}
        if any(not p.requires_grad for p in trainable.values()):
            raise NotImplementedError('partially frozen sync transformer is not supported')
        eng = self._engine(need_sync=False)
        tr = getattr(self, '_sf_trainer', None)
        if tr is None:                                       # ONE trainer per module: its forward counter seeds the dropout masks of each step
            tr = SyncTrainer(self.state_dict(), next(self.parameters()).device, engine=eng)
            object.__setattr__(self, '_sf_trainer', tr)
            object.__setattr__(self, '_sf_trainer_key', None)
        tr.engine = eng
        tr.embd_pdrop, tr.resid_pdrop, tr.attn_pdrop = (float(pd.get(k) or 0.0) for k in ('embd_pdrop', 'resid_pdrop', 'attn_pdrop'))
        tr.tok_pdrop = float(getattr(self.transformer, 'tok_pdrop', 0.0) or 0.0) if self.transformer.training else 0.0    # whole-token dropout (sync_model.py:131-134, 160-161)
        key = tuple((p.data_ptr(), p._version) for p in trainable.values())
        if key != self._sf_trainer_key:                      # an external optimizer moved the nn.Parameters
            tr.load_params({k: p.detach() for k, p in trainable.items()})
            object.__setattr__(self, '_sf_trainer_key', key)
        return SyncTrainFunction.apply(tr, vfeat.detach(), afeat.detach(), *[trainable[k] for k in tr.keys])

    def extract_vfeats(self, vis, for_loop, vis_mask=None):
        """vis_mask: bool, shaped like vis, 0 = masked content (sync_model.py:72-80); like the reference, not with for_loop=True."""
        if vis_mask is not None and for_loop:
            raise AssertionError('cont_mask is not supported with for_loop=True')       # motionformer.py:201
        return self._engine(need_sync=False).extract_vfeats(vis, vis_mask)

    def extract_afeats(self, aud, for_loop, aud_mask=None):
        if aud_mask is not None and for_loop:
            raise AssertionError('cont_mask is not supported with for_loop=True')       # ast.py:153
        return self._engine(need_sync=False).extract_afeats(aud, aud_mask)

    def compute_loss(self, logits, targets, loss_fn: str = None):
        loss = None
        if targets is not None:
            if loss_fn is None or loss_fn == 'cross_entropy':
                loss = _cross_entropy(logits, targets)
            else:
                raise NotImplementedError(f'Loss {loss_fn} not implemented')
        return loss

    def load_state_dict(self, sd: Mapping[str, Any], strict: bool = True, assign: bool = False):
        """sync_model.py:101-114: a longer checkpoint pos_emb is trimmed, a shorter one is an error (`assign` = the newer nn.Module keyword, passed through)."""
        if 'transformer.pos_emb_cfg.pos_emb' in sd:
            weight_len = sd['transformer.pos_emb_cfg.pos_emb'].shape[1]
            self_len = self.transformer.pos_emb_cfg.pos_emb.shape[1]
            if weight_len > self_len:
                sd = dict(sd)
                sd['transformer.pos_emb_cfg.pos_emb'] = sd['transformer.pos_emb_cfg.pos_emb'][:, :self_len, :]
                logging.warning(f'Trimming the state dict for pos emb from {weight_len} to {self_len}')
            elif weight_len < self_len:
                raise ValueError(f'Cant load state dict with shorter seq len ({weight_len} vs {self_len})')
        self._sf_engine = None
        object.__setattr__(self, '_sf_trainer_key', None)      # the train step re-reads its parameter copies (the trainer and its dropout counter stay)
        self.__dict__.pop('_sf_param_lists', None)
        return super().load_state_dict(sd, strict, assign=assign) if assign else super().load_state_dict(sd, strict)


class AVCLIP(torch.nn.Module):
    """Stage-1 segment-level audio-visual contrastive model (train_clip_src/open_clip/model.py:449-585,
    configs/segment_avclip.yaml:4-46): same constructor, attribute names (`v_encoder`, `a_encoder`, `vproj`, `aproj`,
    `logit_scale`) and output dict.  Under torch.no_grad() it is the evaluation / zero-shot path; with autograd enabled the
    step runs on synchformer_amd.stage1.AVCLIPTrainer (HIP forward with saved activations + HIP backward) and the returned loss
    carries the parameter gradients through an autograd bridge, so `backward(total_loss, scaler)`, clip_grad_norm_, AdamW and
    DistributedDataParallel of train_clip_src/training/train.py:103-154 run unchanged."""

    def __init__(self, n_embd: int, afeat_extractor, vfeat_extractor, aproj, vproj, init_scale: float = 0.07,
                 clamp_scale_min: float = 0.001, clamp_scale_max: float = 0.5, gather_for_loss: bool = False):
        super().__init__()
        self.output_dict = True
        self.n_embd = n_embd
        self.v_encoder = instantiate_from_config(vfeat_extractor)
        self.a_encoder = instantiate_from_config(afeat_extractor)
        self.aproj = instantiate_from_config(aproj)
        self.vproj = instantiate_from_config(vproj)
        for name in ('vproj', 'aproj'):
            if not isinstance(getattr(self, name), torch.nn.Identity):
                raise NotImplementedError(f'{name}: only DoNothingBridge (configs/segment_avclip.yaml:37-46) is built natively')
        if not (self.v_encoder.pool_time and self.a_encoder.pool_time):
            raise NotImplementedError("AVCLIP needs agg_time_module='AveragePooling' towers (configs/segment_avclip.yaml:21,33)")
        self.clamp_scale_min, self.clamp_scale_max = clamp_scale_min, clamp_scale_max
        self.init_scale = init_scale
        self.logit_scale = torch.nn.Parameter(torch.ones([]) * self.init_scale)
        self.gather_for_loss = gather_for_loss
        self._sf_engine = None
        self.seg_chunk = 224

    def _engine(self) -> SynchformerEngine:
        key = _param_key(self)
        if self._sf_engine is not None and self._sf_engine[0] == key:
            return self._sf_engine[1]
        dev = self.logit_scale.device
        if dev.type != 'cuda':
            raise RuntimeError('AVCLIP computes on a HIP device only (no CPU fallback); call .to("cuda") first')
        sd = synth.make_state_dict(0)                       # vproj / aproj / transformer slots are unused by this model
        sd.update({'vfeat_extractor.' + k: v for k, v in self.v_encoder.state_dict().items()})
        sd.update({'afeat_extractor.' + k: v for k, v in self.a_encoder.state_dict().items()})
        eng = SynchformerEngine(sd, dev, seg_chunk=self.seg_chunk)
        self._sf_engine = (key, eng)
        return eng

    @torch.no_grad()
    def clamp_logit_scales(self):
        self.logit_scale.clamp_(self.clamp_scale_min, self.clamp_scale_max)
        return (self.logit_scale, None)

    def encode_streams(self, vis, aud, for_loop=False, do_norm=True):
        """vis (B, S, C, Tv, H, W), aud (B, S, Ta, F) -> (B*S, D) visual, None, (B*S, D) audio, None (open_clip/model.py:515-520)."""
        for tower in (self.v_encoder, self.a_encoder):
            _no_extractor_backward(tower)
        eng = self._engine()
        B, S, Ta, Fq = aud.shape
        vfeat = eng.pool_segments(eng.extract_vfeats(vis.permute(0, 1, 3, 2, 4, 5).contiguous()), normalize=do_norm)
        afeat = eng.pool_segments(eng.extract_afeats(aud.permute(0, 1, 3, 2).reshape(B, S, 1, Fq, Ta)), normalize=do_norm)
        return vfeat, None, afeat, None

    def compute_loss(self, vfeat, afeat, vfeat_all, afeat_all, scale, alpha=0.0, vfeat_m=None, afeat_m=None):
        """open_clip/model.py:506-512; `*_all` are passed TRANSPOSED (D, N) like the reference's call sites (`.mT`)."""
        assert alpha == 0.0, f'alpha={alpha} not supported yet'
        losses, sim_v2a, sim_a2v = self._engine().contrastive_loss(vfeat, afeat, vfeat_all.mT.contiguous(), afeat_all.mT.contiguous(),
                                                                   float(scale))
        return losses.mean(), (sim_v2a, sim_a2v)

    def forward(self, vis: torch.Tensor, aud: torch.Tensor, alpha: float = 0.0, for_loop: bool = False, world_size=1):
        """open_clip/model.py:475-504."""
        assert alpha == 0.0, f'alpha={alpha} not supported yet'
        logit_scales = self.clamp_logit_scales()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._train_forward(vis, aud, logit_scales)
        vfeat, _, afeat, _ = self.encode_streams(vis, aud, for_loop, do_norm=True)
        if world_size > 1 and self.gather_for_loss:
            from .dist import all_gather_pair
            vfeat_all, afeat_all = all_gather_pair(vfeat, afeat)
        else:
            vfeat_all, afeat_all = vfeat, afeat
        loss_avc, _ = self.compute_loss(vfeat, afeat, vfeat_all.mT, afeat_all.mT, self.logit_scale, alpha=0)
        return {'rgb_features': (vfeat, None), 'audio_features': (afeat, None), 'logit_scales': logit_scales,
                'losses': {'segment_contrastive_loss': loss_avc}}

    # -- Stage-1 training (both towers trainable, configs/segment_avclip.yaml:13,25) ----------------------------
    def _named_trainables(self):
        named = [('vfeat_extractor.' + n, p) for n, p in self.v_encoder.named_parameters()]
        named += [('afeat_extractor.' + n, p) for n, p in self.a_encoder.named_parameters()] + [('logit_scale', self.logit_scale)]
        return {n: p for n, p in named if not n.startswith('vfeat_extractor.patch_embed.')}

    def _train_forward(self, vis, aud, logit_scales):
        from .stage1 import AVCLIPTrainer, AVCLIPTrainFunction
        named = self._named_trainables()
        if any(not p.requires_grad for p in named.values()):
            raise NotImplementedError('partially frozen AVCLIP (lock_rgb / lock_audio) is not supported: train all of it or none')
        tr = getattr(self, '_sf_trainer', None)
        if tr is None:
            tr = AVCLIPTrainer({k: p.detach() for k, p in named.items()}, self.logit_scale.device,
                               clamp_scale=(self.clamp_scale_min, self.clamp_scale_max), gather_for_loss=self.gather_for_loss,
                               seed=torch.initial_seed() & 0x7FFFFFFF)
            assert tr.keys == list(named), 'parameter order mismatch'
            object.__setattr__(self, '_sf_trainer', tr)
            object.__setattr__(self, '_sf_trainer_key', None)
        # stochastic depth follows the module's mode like the reference's DropPath (train(): on, eval(): identity)
        tr.drop_path_rate = float(self.v_encoder.drop_path_rate) if self.v_encoder.training else 0.0
        key = tuple((p.data_ptr(), p._version) for p in named.values())
        if key != self._sf_trainer_key:                                      # an external optimizer moved the nn.Parameters
            tr.load_params({k: p.detach() for k, p in named.items()})
            object.__setattr__(self, '_sf_trainer_key', key)
        B, S, Ta, Fq = aud.shape
        loss = AVCLIPTrainFunction.apply(tr, vis.permute(0, 1, 3, 2, 4, 5), aud.permute(0, 1, 3, 2).reshape(B, S, 1, Fq, Ta), *named.values())
        return {'rgb_features': (tr.vfeat.clone(), None), 'audio_features': (tr.afeat.clone(), None), 'logit_scales': logit_scales,
                'losses': {'segment_contrastive_loss': loss}}

    def forward_for_logging(self, vis, aud, for_momentum=False, for_loop=False, do_norm=True):
        """open_clip/model.py:535-567: features + the four similarity matrices + loss (zero-shot evaluation feeds on these)."""
        eng = self._engine()
        vfeat, _, afeat, _ = self.encode_streams(vis, aud, for_loop, do_norm)
        out = {'segment_vfeat': vfeat.clone(), 'segment_afeat': afeat.clone()}
        inv = 1.0 / float(self.logit_scale)
        n = vfeat.shape[0]
        for name, (x, y) in {'segment_sim_v2a': (vfeat, afeat), 'segment_sim_a2v': (afeat, vfeat), 'segment_sim_v2v': (vfeat, vfeat),
                             'segment_sim_a2a': (afeat, afeat)}.items():
            from . import ops
            out[name] = ops.similarity(x, y, torch.empty(n, n, device=x.device, dtype=torch.float32), inv)
        losses, _, _ = eng.contrastive_loss(vfeat, afeat, vfeat, afeat, float(self.logit_scale))
        out['segment_contrastive_loss'] = losses.mean()
        return out


def avclip_yaml_model_config(gather_for_loss: bool = False) -> dict:
    """`configs/segment_avclip.yaml: model` with its `${...}` interpolations resolved and ckpt_path: null."""
    tower = dict(ckpt_path=None, extract_features=True, agg_time_module='AveragePooling', add_global_repr=False,
                 agg_segments_module='AveragePooling', max_segments=14)
    bridge = dict(target='model.modules.bridges.DoNothingBridge', params=dict(in_features=768, out_features=768))
    return dict(target='model.modules.feat_extractors.train_clip_src.open_clip.model.AVCLIP', params=dict(
        init_scale=0.07, clamp_scale_min=0.001, clamp_scale_max=0.5, n_embd=768, gather_for_loss=gather_for_loss,
        afeat_extractor=dict(target='model.modules.feat_extractors.audio.ast.AST', params=dict(
            max_spec_t=66, factorize_freq_time=True, agg_freq_module='TransformerEncoderLayer', **tower)),
        vfeat_extractor=dict(target='model.modules.feat_extractors.visual.motionformer.MotionFormer', params=dict(
            factorize_space_time=True, agg_space_module='TransformerEncoderLayer', **tower)),
        aproj=bridge, vproj=bridge))


def sync_yaml_model_config(n_pos: int = 198, num_off_cls: int = 21,
                           transformer_target: str = 'model.sync_model.GlobalTransformer') -> dict:
    """`configs/sync.yaml: model` with its `${...}` interpolations resolved (what OmegaConf hands to the reference)."""
    n_embd = 768
    return dict(target='model.sync_model.Synchformer', params=dict(
        afeat_extractor=dict(target='model.modules.feat_extractors.audio.ast.AST', params=dict(
            ckpt_path=None, extract_features=True, max_spec_t=66, factorize_freq_time=True,
            agg_freq_module='TransformerEncoderLayer', agg_time_module='torch.nn.Identity', add_global_repr=False)),
        vfeat_extractor=dict(target='model.modules.feat_extractors.visual.motionformer.MotionFormer', params=dict(
            ckpt_path=None, extract_features=True, factorize_space_time=True, agg_space_module='TransformerEncoderLayer',
            agg_time_module='torch.nn.Identity', add_global_repr=False)),
        aproj=dict(target='torch.nn.Linear', params=dict(in_features=768, out_features=n_embd)),
        vproj=dict(target='torch.nn.Linear', params=dict(in_features=768, out_features=n_embd)),
        transformer=dict(target=transformer_target, params=dict(
            n_layer=3, n_head=8, n_embd=n_embd, tok_pdrop=0.0, embd_pdrop=0.1, resid_pdrop=0.1, attn_pdrop=0.1,
            pos_emb_cfg=dict(target='model.modules.transformer.RandInitPositionalEncoding',
                             params=dict(block_shape=[n_pos], n_embd=n_embd)),
            off_head_cfg=dict(target='torch.nn.Linear', params=dict(in_features=n_embd, out_features=num_off_cls))))))
