"""MXFP8 towers vs bf16 towers on the same inputs (run on the GPU box): feature / logit deviations and step times."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import synth
from synchformer_amd.engine import SynchformerEngine
dev = torch.device('cuda:0')
for gain in (1.0, 2.0):
    sd = synth.make_state_dict(1337, gain=gain, n_pos=184, n_out=2, head='sync_head')
    e16, e8 = SynchformerEngine(sd, dev), SynchformerEngine(sd, dev, fp8_towers=True)
    u8, aud = synth.make_video_u8(4, 13, 1337).to(dev), synth.make_spectrogram(4, 13, 1337).to(dev)
    v16, v8 = e16.extract_vfeats(u8), e8.extract_vfeats(u8)
    l16, l8 = e16.forward(u8, aud), e8.forward(u8, aud)
    rel = ((v8 - v16).pow(2).mean().sqrt() / v16.pow(2).mean().sqrt()).item()
    print(f'gain {gain}: vfeat rel-RMS {rel:.4f} max {float((v8 - v16).abs().max()):.4f} (std {float(v16.std()):.3f}) | logits bf16 {l16.flatten().tolist()[:4]} fp8 {l8.flatten().tolist()[:4]} max |d| {float((l8 - l16).abs().max()):.4f}')
sd = synth.make_state_dict(1337, n_pos=184, n_out=2, head='sync_head')
for fp8 in (False, True):
    e = SynchformerEngine(sd, dev, fp8_towers=fp8)
    u8, aud = synth.make_video_u8(16, 13, 1337).to(dev), synth.make_spectrogram(16, 13, 1337).to(dev)
    for _ in range(2):
        e.forward(u8, aud)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        e.forward(u8, aud)
    torch.cuda.synchronize()
    print('fp8' if fp8 else 'bf16', f'forward 16 clips x 13 segments: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms')
