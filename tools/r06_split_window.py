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


for B in ((1, 1, 1, 2) if len(sys.argv) > 1 and sys.argv[1] == "b1" else (1, 2, 3, 4, 6, 8, 16)):
    vis, aud = synth.make_video_u8(B, 14, 3).to(dev), synth.make_spectrogram(B, 14, 3).to(dev)
    out = {}
    for mode in ('never', 'always'):
        eng.vis_split_mode = mode
        run = eng.capture(vis, aud)
        out[mode] = med(lambda: run(vis, aud))
        del run
    print(f'{B:2d} clips ({B * 14:3d} segments): graph single stream {out["never"]:8.3f} ms | two halves {out["always"]:8.3f} ms | {100 * (out["always"] / out["never"] - 1):+5.1f} %', flush=True)
