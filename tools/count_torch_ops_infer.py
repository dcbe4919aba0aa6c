"""Which torch-level ops run inside one inference forward (16 clips)?  torch profiler, grouped by op name and GPU kernel.  (run on the GPU box)"""
import sys, collections
sys.path.insert(0, '.')
import torch
from synchformer_amd import synth
from synchformer_amd.engine import SynchformerEngine
from synchformer_amd.frontend import MelFrontend
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
eng = SynchformerEngine(synth.make_state_dict(1337), dev)
mel = MelFrontend(dev)
vis, wav = synth.make_video_u8(B, 14, seed=1337).to(dev), synth.make_wave(B, 14, seed=1337).to(dev)
for _ in range(2):
    eng.forward(vis, mel(wav))
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    eng.forward(vis, mel(wav)); torch.cuda.synchronize()
c = collections.Counter(); t = collections.Counter()
for e in prof.key_averages():
    if e.key.startswith('aten::') or 'copyBuffer' in e.key or 'elementwise' in e.key or 'Memcpy' in e.key or 'Memset' in e.key:
        print(f'{e.count:5d} x {e.key[:90]:90s} cpu {e.cpu_time_total / 1e3:8.2f} ms  gpu {e.device_time_total / 1e3:8.3f} ms')
