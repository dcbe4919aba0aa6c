"""Throw-away stand-in for `timm` (absent here); only the names the reference imports at
vit_helper.py:18-22, motionformer.py:7, video_model_builder.py:12."""
