"""Where does the two-stream split of the visual tower pay under a captured HIP graph?  Clips 1..8 (14 segments each): graph replay with the split forced on / off.
Run on the GPU box:  python tools/r06_split_window.py"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import synth
from synchformer_amd.engine import SynchformerEngine

dev = torch.device('cuda:0')
eng = SynchformerEngine(synth.make_state_dict(1337), dev)
eng.vis_split_min, eng.vis_split_max = 2, 10 ** 6


def med(fn, n=15):
    fn(); fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return 1e3 * sorted(ts)[n // 2]


for B in ((1, 1, 2) if len(sys.argv) > 1 and sys.argv[1] == "b1" else (1, 2, 3, 4, 6, 8, 16)):
    vis, aud = synth.make_video_u8(B, 14, 3).to(dev), synth.make_spectrogram(B, 14, 3).to(dev)
    out = {}
    for mode, parts in (('never', 1), ('always', 2), ('always', 3), ('always', 4)):
        eng.vis_split_mode, eng.vis_split_parts = mode, max(parts, 2)
        run = eng.capture(vis, aud)
        out[parts] = med(lambda: run(vis, aud))
        del run
    print(f'{B:2d} clips ({B * 14:3d} segments): graph single stream {out[1]:8.3f} ms | 2 parts {out[2]:8.3f} ({100 * (out[2] / out[1] - 1):+5.1f} %) | 3 parts {out[3]:8.3f} ({100 * (out[3] / out[1] - 1):+5.1f} %) | 4 parts {out[4]:8.3f} ({100 * (out[4] / out[1] - 1):+5.1f} %)', flush=True)
