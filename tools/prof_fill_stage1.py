import sys; sys.path.insert(0, '.')
import torch, collections
from synchformer_amd import synth
from synchformer_amd.stage1 import AVCLIPTrainer
dev = torch.device('cuda:0')
sd = {k: v for k, v in synth.make_state_dict(1337).items() if k.startswith(('vfeat_extractor.', 'afeat_extractor.'))}
tr = AVCLIPTrainer(sd, dev, lr=1e-4, drop_path_rate=0.2, seed=1337)
vis, aud = synth.make_video_u8(2, 14, seed=1337).to(dev), synth.make_spectrogram(2, 14, seed=1337).to(dev)
for _ in range(2): tr.train_step(vis, aud)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(vis, aud); torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::zeros', 'aten::empty'):
        st = [s for s in (e.stack or []) if 'synchformer_amd' in s]
        c[(e.name, st[0] if st else '?')] += 1
for k, v in c.most_common(25): print(v, k)
