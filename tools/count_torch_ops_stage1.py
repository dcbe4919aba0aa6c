"""Which torch-side (non-HIP-extension) kernels does one Stage-1 train step launch, and from where?  (run on the GPU box)
Uses torch.profiler with stacks; prints kernel name x python call site counts for one step."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import collections, traceback
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
counts = collections.Counter()
names = ['zero_', 'fill_', 'copy_', 'clone', 'contiguous', 'float', 'to']
orig = {}
def wrap(n):
    f = getattr(torch.Tensor, n)
    orig[n] = f
    def g(self, *a, **k):
        if self.is_cuda:
            fr = [x for x in traceback.extract_stack(limit=6)[:-1] if 'synchformer_amd' in x.filename]
            site = f'{Path(fr[-1].filename).name}:{fr[-1].lineno}' if fr else '?'
            counts[(n, site, tuple(self.shape) if self.dim() <= 3 else self.numel())] += 1
        return f(self, *a, **k)
    setattr(torch.Tensor, n, g)
for n in names:
    wrap(n)
for fn in ('zeros', 'ones', 'tensor', 'empty', 'cat', 'arange', 'full'):
    f = getattr(torch, fn)
    def g(*a, _f=f, _n=fn, **k):
        fr = [x for x in traceback.extract_stack(limit=6)[:-1] if 'synchformer_amd' in x.filename]
        site = f'{Path(fr[-1].filename).name}:{fr[-1].lineno}' if fr else '?'
        counts[('torch.' + _n, site, str(k.get('device', '')))] += 1
        return _f(*a, **k)
    setattr(torch, fn, g)
tr.train_step(vis, aud)
torch.cuda.synchronize()
for (n, site, shp), c in sorted(counts.items(), key=lambda kv: -kv[1])[:60]:
    print(f'{c:5d}  {n:12s} {site:24s} {shp}')
