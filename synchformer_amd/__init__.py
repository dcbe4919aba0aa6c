"""synchformer_amd - MI355X-native hot path of v-iashin/Synchformer.

    from synchformer_amd import Synchformer, instantiate_from_config, install_reference_aliases
"""
from .model import (AST, AVCLIP, DoNothingBridge, GlobalTransformer, GlobalTransformerWithSyncabilityHead, MotionFormer,  # noqa: F401
                    RandInitPositionalEncoding, Synchformer, get_obj_from_str, install_reference_aliases,
                    instantiate_from_config, avclip_yaml_model_config, sync_yaml_model_config, uninstall_reference_aliases)

__all__ = ['Synchformer', 'AVCLIP', 'MotionFormer', 'AST', 'GlobalTransformer', 'GlobalTransformerWithSyncabilityHead',
           'RandInitPositionalEncoding', 'DoNothingBridge', 'instantiate_from_config', 'get_obj_from_str',
           'install_reference_aliases', 'uninstall_reference_aliases', 'sync_yaml_model_config', 'avclip_yaml_model_config']

from . import ops as _ops  # noqa: E402
_ops.register_torch_ops()      # torch.ops.synchformer.* (dispatcher-visible leaf ops)
