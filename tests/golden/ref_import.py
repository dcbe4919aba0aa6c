"""Import the REAL reference (v-iashin/Synchformer at /root/reference) in the build container.

Test-time helper only: used by tests/golden/make_golden.py (fixture generation) and by
tests/test_oracle_cpu.py::test_oracle_matches_real_reference (skipped when /root/reference is absent, i.e. on the GPU box).
Nothing in the shipped package, bench.py or smoke() imports this file.

Recipe = SURVEY.md Appendix A: shims for omegaconf/timm on sys.path, two transformers-5.x patches,
cwd=/root/reference.  Nothing under /root/reference is modified or copied.
"""
import contextlib
import os
import sys
from pathlib import Path

REF = Path(os.environ.get('SYNCHFORMER_REFERENCE', '/root/reference'))
SHIMS = Path(__file__).resolve().parent / 'ref_shims'


def reference_available() -> bool:
    return (REF / 'model' / 'sync_model.py').exists()


@contextlib.contextmanager
def _cwd(path):
    old = os.getcwd()
    os.chdir(path)
    try:
        yield
    finally:
        os.chdir(old)


_imported = {}


def import_reference():
    """Returns a dict of the reference's classes (imported once per process)."""
    if _imported:
        return _imported
    if not reference_available():
        raise RuntimeError(f'reference not found at {REF}')
    # alias modules installed by synchformer_amd.install_reference_aliases() would shadow the real `model` package
    for name in [n for n, m in sys.modules.items() if (n == 'model' or n.startswith('model.')) and getattr(m, '__file__', None) is None and not hasattr(m, '__path__')]:
        del sys.modules[name]
    for p in (str(REF), str(SHIMS)):
        if p not in sys.path:
            sys.path.insert(0, p)
    # transformers 5.x lacks two names the reference (pinned to 4.27) imports / calls
    import transformers.pytorch_utils as tpu
    if not hasattr(tpu, 'find_pruneable_heads_and_indices'):
        tpu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    from transformers import PreTrainedModel
    if not hasattr(PreTrainedModel, 'get_head_mask'):
        PreTrainedModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    with _cwd(REF):
        import model.modules.feat_extractors.visual  # noqa: F401  (appends visual/ to sys.path)
        from model.sync_model import Synchformer, GlobalTransformer, GlobalTransformerWithSyncabilityHead
        from model.modules.feat_extractors.visual.motionformer import MotionFormer
        from model.modules.feat_extractors.audio.ast import AST
        from model.modules import transformer as ref_transformer
    _imported.update(Synchformer=Synchformer, GlobalTransformer=GlobalTransformer,
                     GlobalTransformerWithSyncabilityHead=GlobalTransformerWithSyncabilityHead,
                     MotionFormer=MotionFormer, AST=AST, transformer=ref_transformer)
    return _imported


def import_reference_avclip():
    """The REAL `AVCLIP` class (train_clip_src/open_clip/model.py:449-585) and `shift_and_get_preds` machinery of Stage 1.  Its package
    imports torchvision and ftfy at module scope (image transforms / tokenizer clean-up, neither on the audio-visual path); both are absent
    from this image, so import-only stand-ins (tests/golden/ref_shims_late/: every callable in them raises) go on sys.path AFTER
    `transformers` has been imported - had transformers seen a `torchvision` it would try to use it."""
    ref = import_reference()
    if 'AVCLIP' in ref:
        return ref
    late = Path(__file__).resolve().parent / 'ref_shims_late'
    for p in (str(late), str(REF / 'model' / 'modules' / 'feat_extractors' / 'train_clip_src')):     # `import open_clip` (top level) is used inside
        if p not in sys.path:
            sys.path.insert(0, p)
    with _cwd(REF):
        from model.modules.feat_extractors.train_clip_src.open_clip.model import AVCLIP
    ref['AVCLIP'] = AVCLIP
    return ref


def sync_yaml_model_params(n_segments_tokens: int = 198, num_off_cls: int = 21,
                           transformer_target: str = 'model.sync_model.GlobalTransformer') -> dict:
    """`configs/sync.yaml: model.params` with the four `${...}` interpolations resolved by hand and the
    `is_trainable` keys (siblings of `params`, sync.yaml:7,19) left in place (instantiate_from_config
    only forwards `params`, utils/utils.py:88)."""
    n_embd = 768
    return dict(
        afeat_extractor=dict(
            target='model.modules.feat_extractors.audio.ast.AST',
            params=dict(ckpt_path=None, extract_features=True, max_spec_t=66, factorize_freq_time=True,
                        agg_freq_module='TransformerEncoderLayer', agg_time_module='torch.nn.Identity',
                        add_global_repr=False)),
        vfeat_extractor=dict(
            target='model.modules.feat_extractors.visual.motionformer.MotionFormer',
            params=dict(ckpt_path=None, extract_features=True, factorize_space_time=True,
                        agg_space_module='TransformerEncoderLayer', agg_time_module='torch.nn.Identity',
                        add_global_repr=False)),
        aproj=dict(target='torch.nn.Linear', params=dict(in_features=768, out_features=n_embd)),
        vproj=dict(target='torch.nn.Linear', params=dict(in_features=768, out_features=n_embd)),
        transformer=dict(
            target=transformer_target,
            params=dict(n_layer=3, n_head=8, n_embd=n_embd, tok_pdrop=0.0, embd_pdrop=0.1, resid_pdrop=0.1,
                        attn_pdrop=0.1,
                        pos_emb_cfg=dict(target='model.modules.transformer.RandInitPositionalEncoding',
                                         params=dict(block_shape=[n_segments_tokens], n_embd=n_embd)),
                        off_head_cfg=dict(target='torch.nn.Linear',
                                          params=dict(in_features=n_embd, out_features=num_off_cls)))),
    )


def build_reference_synchformer(**kw):
    ref = import_reference()
    with _cwd(REF):
        model = ref['Synchformer'](**sync_yaml_model_params(**kw))
    return model.eval()
