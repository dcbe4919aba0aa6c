"""Import-only stand-in for ftfy (text clean-up of the CLIP tokenizer; not on the audio-visual path)."""


def fix_text(s):
    raise NotImplementedError('ftfy shim: import-only stand-in')
