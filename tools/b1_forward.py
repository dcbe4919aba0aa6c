"""BASELINE configs[0]'s one-clip workload (example.py:174-176) as a profiling target: N synchronised 14-segment forwards incl. the mel front-end (run on the GPU box).
    python tools/b1_forward.py [n_forwards] [clips]      |  rocprofv3 --kernel-trace --stats -- python tools/b1_forward.py 20
Prints the median wall time per forward; SF_AUDIO_SIDE_STREAM=0 serialises the two towers (un-stretched kernel durations for a profile)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import synth
from synchformer_amd.engine import SynchformerEngine
from synchformer_amd.frontend import MelFrontend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device('cuda:0')
eng = SynchformerEngine(synth.make_state_dict(1337), dev)
mel = MelFrontend(dev)
vis = synth.make_video_u8(B, 14, 1337).to(dev)
wav = synth.make_wave(B, 14, 1337).to(dev)
aud = lambda: mel(wav)   # noqa: E731
for _ in range(3):
    eng.forward(vis, aud())
ts = []
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.forward(vis, aud())
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print(f'{B} clip(s), SF_VIS_SPLIT_MAX={eng.vis_split_max}: eager median {1e3 * sorted(ts)[len(ts) // 2]:.3f} ms per forward over {n} (min {1e3 * min(ts):.3f})', end='')
import os
if os.environ.get('B1_GRAPH', '1') == '1':
    a0 = aud()
    run = eng.capture(vis, a0)
    tg = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(vis, mel(wav))
        torch.cuda.synchronize()
        tg.append(time.perf_counter() - t0)
    print(f' | hip graph median {1e3 * sorted(tg)[len(tg) // 2]:.3f} ms (min {1e3 * min(tg):.3f})')
else:
    print()
