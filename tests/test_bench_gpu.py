"""The multi-rank benchmark path as a TESTED path: `bench.py --gpus 2` re-executes itself through torch.distributed.run, runs the
headline region with its barrier / MAX-over-ranks timing, then the three train workloads with their collectives (Stage-2: one flat gradient all-reduce,
`scripts/train_utils.py:208-210` DDP in the reference; Stage-1: 7 buckets launched under the backward + the embedding all-gather / reduce-scatter,
`train_clip_src/open_clip/model.py:489-491`) and the watchdog.  On a single-GPU box the two ranks share cuda:0 over gloo (`--single-device`); on a box with
>= 2 GPUs the same command also runs on RCCL, one GPU per rank."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def _run_bench(*extra, timeout=900, gpus=2, steps=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / 'bench.py'), '--gpus', str(gpus), '--steps', str(steps), '--warmup', '1', '--workload-steps', '1', '--no-cpu-baseline', *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    assert r.returncode == 0, f'bench.py failed (rc {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}'
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, f'expected ONE JSON line from rank 0, got {len(lines)}:\n{r.stdout[-2000:]}'
    return json.loads(lines[0])


def _check_headline(d, backend):
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['unit'] == 'clips/s'
    assert d['rccl_ranks'] == (2 if backend == 'nccl' else 0)
    assert len(d['clips_per_s_by_rank']) == 2 and all(x > 0 for x in d['clips_per_s_by_rank'])
    # whole-job aggregate over the MAX-over-ranks time: never above the sum of the per-rank rates
    assert 0 < d['value'] <= sum(d['clips_per_s_by_rank']) * 1.001
    assert d['config']['parallelism'] == 'replicas x2'


@pytest.mark.parametrize('backend', ['gloo', 'nccl'])
def test_bench_two_ranks_all_workloads(gpu, backend):
    if backend == 'nccl' and torch.cuda.device_count() < 2:
        pytest.skip('RCCL needs one GPU per rank; this box has one (the gloo variant shares cuda:0)')
    extra = ['--dist-backend', 'gloo', '--single-device'] if backend == 'gloo' else ['--dist-backend', 'nccl']
    d = _run_bench(*extra)
    _check_headline(d, backend)
    assert 'workloads_error' not in d, d.get('workloads_error')
    assert set(d['workloads']) == {'infer_from_host', 'train', 'stage1', 'ft'}
    for name, w in d['workloads'].items():
        assert 'error' not in w, f'{name}: {w.get("error")}'
        assert w['clips_per_s'] > 0 and w['ms_per_step'] > 0
        if name == 'infer_from_host':                                         # pinned host inputs, H2D under compute: a ratio to the HBM-resident headline
            assert 0 < w['ratio_to_hbm_resident_headline'] < 1.5 and w['h2d_bytes_per_clip'] == 125 * 3 * 224 * 224 + 80000 * 4
            continue
        comm = w['comm_exposed_ms_last_step_by_rank']                         # HIP events around the gradient-bucket waits, one entry per rank
        assert len(comm) == 2 and all(c >= 0 for c in comm), (name, comm)


@pytest.mark.parametrize('workload', ['stage1', 'train'])
def test_bench_two_ranks_train_workload_as_headline(gpu, workload):
    d = _run_bench('--dist-backend', 'gloo', '--single-device', '--workload', workload, '--no-kernel-timing')
    assert d['n_gpus'] == 2 and d['config']['parallelism'] == 'dp2' and len(d['clips_per_s_by_rank']) == 2
    assert d['config']['clips_per_gpu'] == (2 if workload == 'stage1' else 16)
    comm = d['comm']['exposed_ms_last_step_by_rank']
    assert len(comm) == 2 and all(c >= 0 for c in comm)


@pytest.mark.parametrize('workload', ['infer', 'stage1', 'train'])
def test_bench_eight_ranks_single_device(gpu, workload):
    """The size the driver's scaling run uses - EIGHT ranks - as a tested path before the first 8-GPU run: `bench.py --gpus 8` through torch.distributed.run with all
    ranks on cuda:0 over gloo, one clip per rank; the inference headline (replicas, barrier + MAX-over-ranks timing, per-rank rates) and the two train steps whose
    collectives span the world (Stage-1: 7 gradient buckets under the backward; Stage-2: the flat 90 MB bucket), with the learning rate scaled by the world size."""
    d = _run_bench('--dist-backend', 'gloo', '--single-device', '--workload', workload, '--batch', '1', '--no-kernel-timing', '--no-workloads', gpus=8, steps=1, timeout=1500)
    assert d['n_gpus'] == 8 and d['steps'] == 1 and d['scaling'] == 'weak' and d['rccl_ranks'] == 0
    assert len(d['clips_per_s_by_rank']) == 8 and all(x > 0 for x in d['clips_per_s_by_rank'])
    assert 0 < d['value'] <= sum(d['clips_per_s_by_rank']) * 1.001
    assert d['config']['clips_per_gpu'] == 1 and d['config']['parallelism'] == ('replicas x8' if workload == 'infer' else 'dp8')
    assert d['launcher_numa_node_by_rank'] == [-1] * 8                     # (--single-device: no pinning)
    if workload != 'infer':
        comm = d['comm']['exposed_ms_last_step_by_rank']
        assert len(comm) == 8 and all(c >= 0 for c in comm)
