"""Is the Stage-1 train step (BASELINE configs[3]) bound by the GPU or by the Python launcher?

Times, per step, (a) how long `AVCLIPTrainer.train_step` takes to RETURN (all launches issued, nothing awaited: the step holds no host synchronisation) and (b) the
step with a device synchronise behind it, plus the launch count of one step (every C-ABI call that takes a stream).  If (a) ~ (b) the GPU waits for the launcher.

    python tools/s1_host_issue.py [steps]
"""
import sys
import time

import torch

sys.path.insert(0, '.')
from synchformer_amd import _lib, synth                                   # noqa: E402
from synchformer_amd.stage1 import AVCLIPTrainer                          # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device('cuda:0')
    sd = {k: v for k, v in synth.make_state_dict(1337).items() if k.startswith(('vfeat_extractor.', 'afeat_extractor.'))}
    tr = AVCLIPTrainer(sd, dev, lr=1e-4, drop_path_rate=0.2, seed=1337)
    vis = synth.make_video_u8(2, 14, seed=1337).to(dev)
    aud = synth.make_spectrogram(2, 14, seed=1337).to(dev)
    for _ in range(3):
        tr.train_step(vis, aud)
    torch.cuda.synchronize()
    issue, total = [], []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_step(vis, aud)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    issue.sort(), total.sort()
    # back-to-back steps, the way bench.py times them
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.train_step(vis, aud)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'issue (host returns)  median {issue[len(issue) // 2]:.2f} ms   min {issue[0]:.2f}')
    print(f'issue + synchronise   median {total[len(total) // 2]:.2f} ms   min {total[0]:.2f}')
    print(f'{steps} steps back to back: host done after {(t1 - t0) * 1e3 / steps:.2f} ms/step, device after {(t2 - t0) * 1e3 / steps:.2f} ms/step')
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    tr.train_step(vis, aud)
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
    print(s.getvalue()[:5000])


if __name__ == '__main__':
    main()
