"""Build libsynchformer_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).
One object per .hip source (compiled in parallel, rebuilt only when the source or a header changed), then one link.
Then libsynchformer_torch.so: the host-only TORCH_LIBRARY registration (csrc/sf_torch_library.cpp, g++ against the installed torch's headers) that
`torch.ops.load_library` loads - it links against libsynchformer_hip.so next to it (RPATH $ORIGIN)."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / 'csrc'
OUT = PKG / 'lib' / 'libsynchformer_hip.so'
TORCH_OUT = PKG / 'lib' / 'libsynchformer_torch.so'
TORCH_SRC = CSRC / 'sf_torch_library.cpp'
OBJ = PKG / 'lib' / 'obj'
CFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# extra -D flags for throw-away ablation builds (tools/ab_*.sh); the product build has none
EXTRA = os.environ.get('SF_EXTRA_FLAGS', '').split()


def sources():
    return sorted(CSRC.glob('*.hip'))


def _headers():
    return sorted(CSRC.glob('*.h')) + [PKG.parent / 'include' / 'synchformer_hip.h']


def _stale(target: Path, deps) -> bool:
    return (not target.exists()) or any(d.stat().st_mtime > target.stat().st_mtime for d in deps)


def _flags_txt() -> str:
    return ' '.join(CFLAGS + EXTRA)


def needs_build() -> bool:
    """Stale if a source / header is newer than the library OR the library was built under other flags (an SF_EXTRA_FLAGS ablation build must never be
    served as the product library by a later plain build() / _lib.load())."""
    stamp = OBJ / '.flags'
    return _stale(OUT, sources() + _headers()) or not stamp.exists() or stamp.read_text() != _flags_txt() or _stale(TORCH_OUT, [TORCH_SRC, OUT] + _headers())


def build_torch_library(verbose: bool = True) -> Path:
    """The dispatcher library (TORCH_LIBRARY(synchformer) + TORCH_LIBRARY_IMPL(..., CUDA)): host C++ only, ~10 s with g++."""
    import torch                                                       # headers, libraries and the C++ ABI flag of the interpreter that will load it
    from torch.utils import cpp_extension as ce
    cxx = shutil.which('g++') or shutil.which('c++')
    if cxx is None:
        raise RuntimeError('g++ not found; cannot build libsynchformer_torch.so')
    tlib = ce.library_paths()[0]
    cmd = [cxx, '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-D__HIP_PLATFORM_AMD__', '-DUSE_ROCM', f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}',
           *[f'-I{p}' for p in ce.include_paths()], '-I/opt/rocm/include', str(TORCH_SRC), '-o', str(TORCH_OUT.with_suffix('.so.tmp')),
           f'-L{tlib}', '-ltorch', '-ltorch_cpu', '-lc10', '-lc10_hip', '-ltorch_hip', f'-L{OUT.parent}', '-lsynchformer_hip', '-Wl,-rpath,$ORIGIN']
    if verbose:
        print('[build]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(TORCH_OUT.with_suffix('.so.tmp'), TORCH_OUT)
    return TORCH_OUT


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return OUT
    if not force and not _stale(OUT, sources() + _headers()) and (OBJ / '.flags').exists() and (OBJ / '.flags').read_text() == _flags_txt():
        build_torch_library(verbose)                                    # only the dispatcher library is stale
        return OUT
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not Path(hipcc).exists():
        raise RuntimeError('hipcc not found; cannot build libsynchformer_hip.so')
    OBJ.mkdir(parents=True, exist_ok=True)
    hdrs = _headers()
    stamp = OBJ / '.flags'
    flags_txt = _flags_txt()
    if not stamp.exists() or stamp.read_text() != flags_txt:
        force = True
    jobs = []
    for src in sources():
        obj = OBJ / (src.stem + '.o')
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *CFLAGS, *EXTRA, '-c', str(src), '-o', str(obj)]
        if verbose:
            print('[build]', ' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    stamp.write_text(flags_txt)
    tmp = OUT.with_suffix('.so.tmp')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *[str(OBJ / (s.stem + '.o')) for s in sources()], '-o', str(tmp)]
    if verbose:
        print('[build]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, OUT)
    build_torch_library(verbose)
    return OUT


AB_DIR = PKG / 'lib' / 'ab'
AB_OUT = AB_DIR / 'libsynchformer_hip_ablation.so'


def build_ablation(force: bool = False, verbose: bool = True) -> Path:
    """lib/ab/libsynchformer_hip_ablation.so: the same sources with -DSF_ABLATION, i.e. INCLUDING the measured-slower alternatives the product library no longer carries
    (sf_gemm_bf16 tile configs 1-3, 5, 6, 8-10, 12; schedule 2 of sf_gemm_res_ln768).  Loaded explicitly by the bit-identity tests and the tools/ benchmarks
    (`_lib.load_ablation()`), never by the package."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    obj = AB_DIR / 'obj'
    obj.mkdir(parents=True, exist_ok=True)
    hdrs = _headers()
    flags = CFLAGS + ['-DSF_ABLATION']
    stamp = obj / '.flags'
    if not stamp.exists() or stamp.read_text() != ' '.join(flags):
        force = True
    jobs = [(src, obj / (src.stem + '.o')) for src in sources() if force or _stale(obj / (src.stem + '.o'), [src] + hdrs)]
    if not jobs and AB_OUT.exists():
        return AB_OUT

    def compile_one(job):
        cmd = [hipcc, *flags, '-c', str(job[0]), '-o', str(job[1])]
        if verbose:
            print('[build]', ' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    stamp.write_text(' '.join(flags))
    tmp = AB_OUT.with_suffix('.so.tmp')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', *[str(obj / (s.stem + '.o')) for s in sources()], '-o', str(tmp)]
    if verbose:
        print('[build]', ' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, AB_OUT)
    return AB_OUT


if __name__ == '__main__':
    import sys
    build(force='--force' in sys.argv)
    if '--ablation' in sys.argv:
        build_ablation(force='--force' in sys.argv)
