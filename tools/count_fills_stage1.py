"""Where do the torch fill / zero launches of a Stage-1 step come from?  (python-level call sites; run on the GPU box)"""
import sys, traceback, collections
sys.path.insert(0, '.')
import torch
from synchformer_amd import synth
from synchformer_amd.stage1 import AVCLIPTrainer
dev = torch.device('cuda:0')
sd = {k: v for k, v in synth.make_state_dict(1337).items() if k.startswith(('vfeat_extractor.', 'afeat_extractor.'))}
tr = AVCLIPTrainer(sd, dev, lr=1e-4, drop_path_rate=0.2, seed=1337)
vis, aud = synth.make_video_u8(2, 14, seed=1337).to(dev), synth.make_spectrogram(2, 14, seed=1337).to(dev)
for _ in range(2):
    tr.train_step(vis, aud)
torch.cuda.synchronize()
c = collections.Counter()
b = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'synchformer_amd' in fr.filename:
            return f'{fr.filename.split("/")[-1]}:{fr.lineno} {fr.line[:90]}'
    return '?'


for name in ('zero_', 'fill_', 'copy_'):
    orig = getattr(torch.Tensor, name)

    def mk(orig, name):
        def f(self, *a, **k):
            s = site()
            c[(name, s)] += 1
            b[(name, s)] += self.numel() * self.element_size()
            return orig(self, *a, **k)
        return f
    setattr(torch.Tensor, name, mk(orig, name))
for fn in ('zeros', 'zeros_like', 'full', 'ones', 'empty'):
    orig = getattr(torch, fn)

    def mk2(orig, fn):
        def f(*a, **k):
            c[(fn, site())] += 1
            return orig(*a, **k)
        return f
    setattr(torch, fn, mk2(orig, fn))
tr.train_step(vis, aud)
torch.cuda.synchronize()
for k, v in c.most_common(40):
    print(v, f'{b[k] / 1e6:9.2f} MB', k)
