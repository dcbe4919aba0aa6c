"""Which parameter gradients does the Stage-1 backward fully overwrite?  Poison the flat gradient buffer with NaN instead of zeroing it and list what is still NaN."""
import os, sys
sys.path.insert(0, '.')
os.environ['SF_S1_POISON'] = '1'
import torch
from synchformer_amd import synth
from synchformer_amd.stage1 import AVCLIPTrainer
dev = torch.device('cuda:0')
sd = {k: v for k, v in synth.make_state_dict(1337).items() if k.startswith(('vfeat_extractor.', 'afeat_extractor.'))}
tr = AVCLIPTrainer(sd, dev, lr=1e-4, drop_path_rate=0.2, seed=1337)
vis, aud = synth.make_video_u8(2, 14, seed=1337).to(dev), synth.make_spectrogram(2, 14, seed=1337).to(dev)
tr.forward_backward(vis, aud)
torch.cuda.synchronize()
bad = [(k, int(torch.isnan(tr.g[k]).sum()), tr.g[k].numel()) for k in tr.keys if torch.isnan(tr.g[k]).any()]
print(len(bad), 'of', len(tr.keys), 'gradients hold NaN after the backward')
for b in bad[:40]:
    print(b)
