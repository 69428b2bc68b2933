"""In-tree build of libd2p_hip.so with hipcc for gfx950 (no JIT cache, no cmake).

Every source is compiled to its own object (csrc/build/*.o, in parallel, rebuilt only when the
source or a header is newer) and the objects are linked into csrc/libd2p_hip.so."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
SOURCES = ['api.hip', 'gemm.hip', 'conv.hip', 'conv_direct.hip', 'conv_frames.hip', 'conv_rows.hip', 'conv_wide.hip', 'bn.hip',
           'lstm.hip', 'lstm_step.hip', 'lstm_persist.hip', 'greedy.hip', 'xent.hip', 'misc.hip', 'adam.hip', 'rn.hip', 'small_products.hip']
HEADERS = ['common.h', 'conv_geom.h', 'gemm_core.h', 'prof.h', 'lstm_math.h', 'lstm_internal.h',
           os.path.join('..', '..', 'include', 'd2p.h')]
OUT = os.path.join(CSRC, 'libd2p_hip.so')
OBJDIR = os.path.join(CSRC, 'build')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _newest_header():
    return max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)


def _hipcc():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    return hipcc if os.path.exists(hipcc) else 'hipcc'


def _stamp():
    """Compiler + flags: objects built with other flags (or another hipcc) are stale whatever their mtime."""
    import hashlib
    return hashlib.sha1((' '.join([_hipcc()] + FLAGS)).encode()).hexdigest()[:12]


def _obj(src):
    return os.path.join(OBJDIR, '%s.%s.o' % (os.path.splitext(src)[0], _stamp()))


def _stale_objects():
    hdr = _newest_header()
    return [s for s in SOURCES if _mtime(_obj(s)) < max(_mtime(os.path.join(CSRC, s)), hdr)]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(_mtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


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
            hipcc = _hipcc()
            os.makedirs(OBJDIR, exist_ok=True)
            todo = list(SOURCES) if force else _stale_objects()

            def compile_one(src):
                # to a temporary name, renamed on success: a compile that is killed part-way must not leave a
                # truncated object that looks newer than its source
                tmp = '%s.tmp.%d' % (_obj(src), os.getpid())
                cmd = [hipcc] + FLAGS + ['-c', src, '-o', tmp]
                if verbose:
                    print(' '.join(cmd), flush=True)
                try:
                    subprocess.run(cmd, cwd=CSRC, check=True)
                    os.replace(tmp, _obj(src))
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)

            jobs = int(os.environ.get('D2P_BUILD_JOBS', str(min(8, os.cpu_count() or 1))))
            with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
                list(pool.map(compile_one, todo))
            tmp = '%s.tmp.%d' % (OUT, os.getpid())
            cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + [_obj(s) for s in SOURCES] + ['-o', tmp]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, cwd=CSRC, check=True)
            os.replace(tmp, OUT)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return OUT


def build_variant_library(tag, defines, source='lstm_persist.hip', verbose=False):
    """An A/B copy of the library: `source` compiled with extra -D `defines`, every other object shared with the product
    library -> csrc/libd2p_hip_<tag>.so (git-ignored; loaded through D2P_LIB_PATH by tools/ab_env.sh)."""
    build_library()
    out = os.path.join(CSRC, 'libd2p_hip_%s.so' % tag)
    obj = os.path.join(OBJDIR, '%s.%s.%s.o' % (os.path.splitext(source)[0], tag, _stamp()))
    # (an item that starts with '-' is a compiler flag, e.g. -mllvm -amdgpu-igrouplp-exact-solver; anything else a define)
    cmd = [_hipcc()] + FLAGS + [d if d.startswith('-') else '-D' + d for d in defines] + ['-c', source, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    objs = [obj if s == source else _obj(s) for s in SOURCES]
    subprocess.run([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out], cwd=CSRC, check=True)
    return out


def build_stamps_library(verbose=False):
    """A DIAGNOSTIC copy of the library for tools/lstm_launch_stamps.py: lstm_persist.hip compiled with -DD2P_PS_STAMPS
    (launch-boundary time stamps in the persistent recurrences), every other object shared with the product library.
    Loaded through D2P_LIB_PATH by that tool only; the product library's code is unchanged by the macro's absence."""
    build_library()
    out = os.path.join(CSRC, 'libd2p_hip_stamps.so')
    src = os.path.join(CSRC, 'lstm_persist.hip')
    obj = os.path.join(OBJDIR, 'lstm_persist.stamps.%s.o' % _stamp())
    if _mtime(obj) < max(_mtime(src), _newest_header()):
        cmd = [_hipcc()] + FLAGS + ['-DD2P_PS_STAMPS', '-c', 'lstm_persist.hip', '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, cwd=CSRC, check=True)
    if _mtime(out) < max(_mtime(obj), _mtime(OUT)):
        objs = [obj if s == 'lstm_persist.hip' else _obj(s) for s in SOURCES]
        subprocess.run([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out], cwd=CSRC, check=True)
    return out


if __name__ == '__main__':
    import sys
    if '--stamps' in sys.argv:
        print(build_stamps_library(verbose=True))
    elif '--variant' in sys.argv:            # [--source gemm.hip] --variant TAG DEFINE|-flag [...]
        src = 'lstm_persist.hip'
        if '--source' in sys.argv:
            j = sys.argv.index('--source')
            src = sys.argv[j + 1]
            del sys.argv[j:j + 2]
        i = sys.argv.index('--variant')
        print(build_variant_library(sys.argv[i + 1], sys.argv[i + 2:], source=src, verbose=True))
    else:
        print(build_library(force='--force' in sys.argv, verbose=True))
