"""Import-only stand-in for torchvision (absent from this image, no network).  The reference's open_clip package imports it at module
scope for image transforms and a frozen BatchNorm; nothing on the Synchformer / AVCLIP hot path calls into it.  Every name below raises
if it is actually used, so a golden vector can never depend on this shim."""
from . import ops, transforms  # noqa: F401

__version__ = '0.0.0+shim'
