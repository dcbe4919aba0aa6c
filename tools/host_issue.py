"""Launcher time against device time of one step of any bench.py workload (infer | train | stage1 | ft): is the GPU ever waiting for Python?

    python tools/host_issue.py <workload> [steps]

Prints how long the step function takes to RETURN on an idle GPU (launches issued, nothing awaited), the synchronised step, and both for steps issued back to back.
(tools/s1_host_issue.py is the Stage-1 version with a cProfile of the launcher.)"""
import argparse
import sys
import time

import torch

sys.path.insert(0, '.')
import bench                                                             # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'infer'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    args = argparse.Namespace(batch=None, workload=name, dropin=False, seg_chunk=224)
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    w = bench.build_workload(name, args, dev, 0, 1, 0)
    fn, vis, aud = w['step_fn'], w['vis'], w['aud']
    for _ in range(3):
        fn(vis, aud)
    torch.cuda.synchronize()
    issue, total = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        fn(vis, aud)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    issue.sort(), total.sort()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn(vis, aud)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name}: launcher returns after {issue[len(issue) // 2]:.2f} ms (idle GPU), synchronised step {total[len(total) // 2]:.2f} ms; back to back: launcher '
          f'{(t1 - t0) * 1e3 / steps:.2f} ms/step, device {(t2 - t0) * 1e3 / steps:.2f} ms/step')


if __name__ == '__main__':
    main()
