"""In-tree build of libd2p_hip.so with hipcc for gfx950 (no JIT cache, no cmake)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
SOURCES = ['api.hip', 'gemm.hip', 'conv.hip', 'conv_direct.hip', 'conv_frames.hip', 'conv_rows.hip', 'bn.hip', 'lstm.hip', 'lstm_step.hip', 'greedy.hip', 'xent.hip', 'misc.hip',
           'adam.hip']
HEADERS = ['common.h', 'conv_geom.h', 'gemm_core.h', 'prof.h', 'lstm_math.h', os.path.join('..', '..', 'include', 'd2p.h')]
OUT = os.path.join(CSRC, 'libd2p_hip.so')


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build_library(force=False, verbose=False):
    """Compile every HIP source into csrc/libd2p_hip.so.  hipcc cross-compiles for gfx950
    without a GPU present."""
    if not force and not _stale():
        return OUT
    # one builder at a time (several ranks of a multi-GPU launch may get here together): the others
    # wait on the lock and then find the library fresh
    import fcntl
    with open(os.path.join(CSRC, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return OUT
            hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
            if not os.path.exists(hipcc):
                hipcc = 'hipcc'
            tmp = '%s.tmp.%d' % (OUT, os.getpid())
            cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
                   '-Wno-unused-result'] + SOURCES + ['-o', tmp]
            if verbose:
                print(' '.join(cmd))
            subprocess.run(cmd, cwd=CSRC, check=True)
            os.replace(tmp, OUT)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUT


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
