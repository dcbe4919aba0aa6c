"""Build libsynchformer_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / 'csrc'
OUT = PKG / 'lib' / 'libsynchformer_hip.so'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(CSRC.glob('*.hip'))


def needs_build() -> bool:
    if not OUT.exists():
        return True
    deps = sources() + sorted(CSRC.glob('*.h')) + [PKG.parent / 'include' / 'synchformer_hip.h']
    return any(d.stat().st_mtime > OUT.stat().st_mtime for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not Path(hipcc).exists():
        raise RuntimeError('hipcc not found; cannot build libsynchformer_hip.so')
    OUT.parent.mkdir(parents=True, exist_ok=True)
    tmp = OUT.with_suffix('.so.tmp')
    cmd = [hipcc, *FLAGS, *map(str, sources()), '-o', str(tmp)]
    if verbose:
        print('[build]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, OUT)
    return OUT


if __name__ == '__main__':
    build(force=True)
