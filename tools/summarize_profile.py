"""Condense a tools/profile_bench.sh output directory into the small files kept under profiles/.
    python tools/summarize_profile.py gpurun_out/prof_r01 profiles/r01_bench"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
dst.parent.mkdir(parents=True, exist_ok=True)
shutil.copy(src / 'stats' / 'stats_kernel_stats.csv', f'{dst}_kernel_stats.csv')


def short(name):
    name = name.replace('void ', '')
    return name.split('(')[0][:110]


pmc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sorted(src.glob('pmc_*')):
    if not d.is_dir():
        continue
    f = d / f'{d.name[4:]}_counter_collection.csv'
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        pmc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if d.name == 'pmc_sq' and (r['Dispatch_Id'], k) not in seen:
            seen.add((r['Dispatch_Id'], k))
            calls[k] += 1
stats = list(csv.DictReader(open(src / 'stats' / 'stats_kernel_stats.csv')))
bench = None
for line in open(src / 'stats.log'):
    if line.startswith('{"metric"'):
        bench = json.loads(line)
lines = [f'# rocprofv3 summary - {dst.name}', '',
         'Command: `SF_AUDIO_SIDE_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-workloads`  (the configuration of bench.py\'s roofline pass: both towers on one stream, so kernel durations are not stretched by co-running kernels; the product runs the audio tower on a second stream)',
         '(PMC counters from separate `--pmc` passes of the same command; 3 forward passes x 16 clips in every pass).', '']
if bench:
    lines += [f'bench line under the profiler: {bench["value"]} clips/s, {bench["ms_per_step"]} ms/step', '']
lines += ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
for r in stats[:16]:
    lines.append(f'| `{short(r["Name"])}` | {r["Calls"]} | {float(r["TotalDurationNs"]) / 1e6:.1f} | {float(r["AverageNs"]) / 1e3:.1f} | {float(r["Percentage"]):.1f} |')
gem = [r for r in stats if 'gemm_bf16' in r['Name'] or 'gemm_res_ln768' in r['Name'] or 'qkv_time_attn' in r['Name'] or 'qkv_time2_attn' in r['Name'] or 'qkv_space_attn' in r['Name']]
if gem:
    calls_g = sum(int(r['Calls']) for r in gem)
    tot_g = sum(float(r['TotalDurationNs']) for r in gem) / 1e6
    tot_all = sum(float(r['TotalDurationNs']) for r in stats) / 1e6
    lines += ['', f'All `sf_gemm_bf16` + `sf_gemm_res_ln768` + `sf_qkv_time_attention` kernels together (the kernel `bench.py` reports as `roofline`): {calls_g} launches over the 3 passes '
              f'= {calls_g // 3} per step, {tot_g:.1f} ms, **{1e3 * tot_g / calls_g:.1f} us average**, {100 * tot_g / tot_all:.1f} % of GPU time '
              '(compare `roofline.launches`, `roofline.avg_launch_ms`, `roofline.share_of_step_time` of the bench line).']
lines += ['', 'PMC per kernel (summed over all dispatches of the run, then per launch).  FETCH_SIZE/WRITE_SIZE are in KiB as rocprofv3',
          'reports them; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md §HBM) - the `fetch x2` column',
          'applies that correction.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x GRBM_GUI_ACTIVE/8).', '',
          '| kernel | launches | fetch MiB/launch (x2) | write MiB/launch | MFMA busy % | wait_any % | L2 hit % | LDS bank-conflict % |', '|---|---|---|---|---|---|---|---|']
for k, c in sorted(pmc.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0))[:11]:
    n = max(calls[k], 1)
    fetch = c.get('FETCH_SIZE', 0) * 2 / 1024 / n
    write = c.get('WRITE_SIZE', 0) / 1024 / n
    gui = c.get('GRBM_GUI_ACTIVE', 0) / 8
    mfma = 100 * c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui) if gui else 0
    wait = 100 * c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES'] if c.get('SQ_WAVE_CYCLES') else 0
    hit = 100 * c.get('TCC_HIT_sum', 0) / max(c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 0), 1)
    bank = 100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_LDS_IDX_ACTIVE', 0), 1)
    lines.append(f'| `{k[:70]}` | {n} | {fetch:.0f} | {write:.0f} | {mfma:.1f} | {wait:.1f} | {hit:.1f} | {bank:.1f} |')
# HBM traffic of the roofline kernel (all sf_gemm_bf16 launches), per launch: FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, KiB -> bytes
gk = [(k, c) for k, c in pmc.items() if ('gemm_bf16' in k or 'gemm_res_ln768' in k or 'qkv_time_attn' in k or 'qkv_time2_attn' in k or 'qkv_space_attn' in k) and calls[k]]
if gk:
    n_l = sum(calls[k] for k, _ in gk)
    fetch_b = sum(c.get('FETCH_SIZE', 0) for _, c in gk) * 2 * 1024
    write_b = sum(c.get('WRITE_SIZE', 0) for _, c in gk) * 1024
    # per kernel symbol (the name rocprofv3 prints = the name bench.py files its live timings under): bench.py attaches the NAMED kernel's own counter bytes
    # to `roofline.traffic`, not the family average
    per = {k: {'launches': calls[k], 'hbm_read_bytes_per_launch': c.get('FETCH_SIZE', 0) * 2 * 1024 / calls[k], 'hbm_write_bytes_per_launch': c.get('WRITE_SIZE', 0) * 1024 / calls[k],
               'traffic_bytes_per_launch': (c.get('FETCH_SIZE', 0) * 2 + c.get('WRITE_SIZE', 0)) * 1024 / calls[k]} for k, c in gk}
    roof = {'kernel': 'sf_gemm_bf16 + sf_gemm_res_ln768 + sf_qkv_time_attention (all launches)', 'launches_profiled': n_l, 'hbm_read_bytes_per_launch': fetch_b / n_l,
            'hbm_write_bytes_per_launch': write_b / n_l, 'traffic_bytes_per_launch': (fetch_b + write_b) / n_l, 'per_kernel': per,
            'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh (FETCH_SIZE x2 per MI355X_MICROARCH.md)'}
    Path(f'{dst}_roofline.json').write_text(json.dumps(roof, indent=1) + '\n')
    lines += ['', f'HBM traffic of the `sf_gemm_bf16` launches from the PMC passes: {fetch_b / n_l / 1e6:.0f} MB read (x2-corrected) + '
              f'{write_b / n_l / 1e6:.0f} MB written per launch (`{dst.name}_roofline.json`, read by `bench.py` for `roofline.traffic`).']
Path(f'{dst}_summary.md').write_text('\n'.join(lines) + '\n')
print('\n'.join(lines))
