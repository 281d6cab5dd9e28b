"""Builds libfwdtaco_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m forwardtacotron_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to forwardtacotron_amd/csrc/build/, the shared
library to forwardtacotron_amd/libfwdtaco_hip.so (git-ignored; it travels with gpurun snapshots).
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libfwdtaco_hip.so')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-I', INCLUDE, '-I', CSRC]
# per-file additions (each explained in the file's own header comments)
EXTRA = {'ft_gemm_b3.hip': ['-fno-slp-vectorize']}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers_mtime():
    m = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith('.h'):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, force, hdr_m):
    obj = os.path.join(OBJ, src[:-4] + '.o')
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) > max(os.path.getmtime(sp), hdr_m)):
        return obj, False, ''
    r = subprocess.run([HIPCC] + FLAGS + EXTRA.get(src, []) + ['-c', sp, '-o', obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
    return obj, True, r.stderr


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _headers_mtime()
    srcs = _sources()
    rebuilt = False
    objs = []
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        for obj, did, warn in ex.map(lambda s: _compile(s, force, hdr_m), srcs):
            objs.append(obj)
            rebuilt |= did
            if warn and verbose:
                sys.stderr.write(warn)
    if rebuilt or not os.path.exists(LIB):
        r = subprocess.run([HIPCC, '-shared', '--offload-arch=gfx950', '-Wl,-z,defs', '-o', LIB] + objs,   # undefined symbols fail the build
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'built {LIB}')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
