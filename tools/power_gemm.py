"""Energy of one GEMM launch: loops sf_gemm_bf16 on one shape for a few seconds while rocm-smi is sampled, prints time, power, clock and joules per launch.
Run under tools/ab_pp_run.sh (BENCH=tools/power_gemm.py) to compare the ablation builds of config 11: which part of the launch's energy is not MFMA?
    SHAPE=qkv|fc1|fc2|proj  DATA=rand|zero  SECS=3  python tools/power_gemm.py [n_seg]
"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synchformer_amd import ops
from synchformer_amd._lib import load

lib = load()
dev = torch.device('cuda:0')
SHAPES = {'qkv': (2304, 768, torch.bfloat16, False, False), 'proj': (768, 768, torch.float32, False, True),
          'fc1': (3072, 768, torch.bfloat16, True, False), 'fc2': (768, 3072, torch.float32, False, True)}


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(['rocm-smi', '-P', '-c'], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r'Package Power \(W\): ([0-9.]+)', txt)
            c = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', txt)
            if p and c:
                out.append((time.time(), float(p.group(1)), int(c.group(1))))
        except Exception:
            pass
        time.sleep(0.25)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 224
    M = n * 1569
    N, K, out_dt, gelu, res = SHAPES[os.environ.get('SHAPE', 'qkv')]
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    if os.environ.get('DATA') == 'zero':
        a.zero_(); w.zero_()
    b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=out_dt)
    lib.sf_gemm_force_config(int(os.environ.get('CFG', '11')))
    run = lambda: ops.gemm(a, w, b, out, gelu=gelu, residual=out if res else None)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples)); th.start()
    secs = float(os.environ.get('SECS', '3'))
    t0 = time.time(); launches = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            run()
        launches += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    t_end = time.time()
    stop.set(); th.join()
    us = e0.elapsed_time(e1) * 1e3 / launches
    mid = [s for s in samples if t0 + 1.0 < s[0] < t_end - 0.2] or samples
    pw = sorted(s[1] for s in mid)[len(mid) // 2] if mid else float('nan')
    ck = sorted(s[2] for s in mid)[len(mid) // 2] if mid else 0
    print(f"{os.environ.get('SHAPE', 'qkv')} M {M} N {N} K {K} data {os.environ.get('DATA', 'rand')}: {us:7.1f} us/launch  {2.0 * M * N * K / us / 1e6:5.0f} TF(nominal)  "
          f"{pw:6.0f} W  {ck} MHz  {pw * us * 1e-6:6.3f} J/launch  ({len(mid)} samples)", flush=True)
    lib.sf_gemm_force_config(-1)


if __name__ == '__main__':
    main()
