"""Build libpsvae_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libpsvae_hip.so')
SOURCES = ['gemm.hip', 'fused_layer.hip', 'graph_kernels.hip', 'head_kernels.hip', 'gru.hip', 'step.hip', 'inference.hip', 'collate_device.hip', 'capi.hip', 'profiler.hip', 'rng.hip', 'collate.cpp']
HEADERS = ['psvae_internal.h', 'fused_device.h', os.path.join('..', '..', 'include', 'psvae.h')]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src.rsplit('.', 1)[0] + '.o')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f'--- {src} failed ---\n{out.decode()}\n')
        elif verbose and out:
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError('hipcc failed')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
