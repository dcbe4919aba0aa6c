"""Probe of the MX-scaled MFMA operand / scale conventions through sf_gemm_mxfp8 (run on the GPU box)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import ops
dev = torch.device('cuda:0')
M = N = 256
K = 128
ONE = 0x38          # e4m3 1.0


def run(aq, asc, wq, wsc):
    out = torch.zeros(M, N, device=dev)
    ops.gemm_mxfp8(aq.to(dev), asc.to(dev), wq.to(dev), wsc.to(dev), None, out)
    return out.cpu()


s1 = torch.full((1, M, 4), 127, dtype=torch.uint8)
print('pairing of 32-blocks (rows: A block, cols: W block) -> out[0,0]')
for a in range(4):
    row = []
    for b in range(4):
        aq = torch.zeros(M, K, dtype=torch.uint8); wq = torch.zeros(N, K, dtype=torch.uint8)
        aq[:, a * 32:(a + 1) * 32] = ONE
        wq[:, b * 32:(b + 1) * 32] = ONE
        row.append(float(run(aq, s1, wq, s1)[0, 0]))
    print(a, row)
print('byte-position pairing inside block 0: A nonzero at byte i only, W nonzero at byte j only')
for i in (0, 1, 4, 15, 16, 31):
    row = []
    for j in (0, 1, 4, 15, 16, 31):
        aq = torch.zeros(M, K, dtype=torch.uint8); wq = torch.zeros(N, K, dtype=torch.uint8)
        aq[:, i] = ONE; wq[:, j] = ONE
        row.append(float(run(aq, s1, wq, s1)[0, 0]))
    print(i, row)
print('row/col identity: A[m, 0] = 1 for m == 5 only; W[n, 0] = 1 for n == 9 only -> nonzero outputs at:')
aq = torch.zeros(M, K, dtype=torch.uint8); wq = torch.zeros(N, K, dtype=torch.uint8)
aq[5, 0] = ONE; wq[9, 0] = ONE
o = run(aq, s1, wq, s1)
print(torch.nonzero(o).tolist(), o[5, 9].item())
aq = torch.zeros(M, K, dtype=torch.uint8); wq = torch.zeros(N, K, dtype=torch.uint8)
aq[133, 70] = ONE; wq[200, 70] = ONE
o = run(aq, s1, wq, s1)
print(torch.nonzero(o).tolist())
print('scales: all ones in block 1 (k 32..63); A scale byte of (row 7, block b) = 128 -> which rows double?')
for b in range(4):
    aq = torch.zeros(M, K, dtype=torch.uint8); wq = torch.zeros(N, K, dtype=torch.uint8)
    aq[:, 32:64] = ONE; wq[:, 32:64] = ONE
    sa = s1.clone(); sa[0, 7, b] = 128
    o = run(aq, sa, wq, s1)
    print('A scale byte', b, '-> rows != 32:', torch.nonzero(o[:, 0] != 32).flatten().tolist(), o[7, 0].item())
for b in range(4):
    aq = torch.zeros(M, K, dtype=torch.uint8); wq = torch.zeros(N, K, dtype=torch.uint8)
    aq[:, 32:64] = ONE; wq[:, 32:64] = ONE
    sw = s1.clone(); sw[0, 11, b] = 129
    o = run(aq, s1, wq, sw)
    print('W scale byte', b, '-> cols != 32:', torch.nonzero(o[0, :] != 32).flatten().tolist(), o[0, 11].item())
