#!/opt/conda/bin/python3.9
"""Builds the dataset fixtures by RUNNING THE REFERENCE's dataset pipeline (build container
only: needs /root/reference and /opt/conda/bin/python3.9 with h5py, ply, numpy 1.26):

  karel_env/generator.py  ->  append_demonstration.py  ->  add_per.py      (a 12-program data.hdf5)
  tools/convert_karel_hdf5.py                                               (h5py-free copy of it)
  karel_env/dataset_karel.py: all_ids() + Dataset.get_data(id)              (expected outputs)

Shims, nothing edited or copied: sys.path for the Python-2 implicit-relative imports, stub
modules for `progressbar` / `colorlog` (absent here), `np.bool`, and h5py's removed
`Dataset.value` accessor.  Outputs (committed):
  tests/golden/karel_dataset/            converted dataset (tools/convert_karel_hdf5.py format)
  tests/golden/karel_dataset_expected.npz  get_data() results + split order of the reference
"""
import os
import runpy
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

np.bool = bool
import h5py  # noqa: E402

h5py.Dataset.value = property(
    lambda self: (lambda v: v.decode() if isinstance(v, bytes) else v)(self[()]))   # py2 str semantics
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [REF, os.path.join(REF, 'karel_env'), os.path.join(REF, 'karel_env', 'dsl')]


class _Bar(object):
    def __init__(self, *a, **k):
        pass

    def start(self):
        pass

    def update(self, *_):
        pass

    def finish(self):
        pass


pb = types.ModuleType('progressbar')
pb.ProgressBar, pb.Bar, pb.Percentage = _Bar, (lambda *a, **k: None), (lambda *a, **k: None)
sys.modules['progressbar'] = pb
cl = types.ModuleType('colorlog')


class _Fmt(object):
    def __init__(self, *a, **k):
        import logging
        self._f = logging.Formatter('%(message)s')

    def format(self, record):
        return self._f.format(record)

    def __getattr__(self, name):
        return getattr(self._f, name)


cl.ColoredFormatter = _Fmt
sys.modules['colorlog'] = cl


def run_script(path, argv, cwd):
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [path] + argv
    os.chdir(cwd)
    try:
        runpy.run_path(path, run_name='__main__')
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)


def main():
    work = tempfile.mkdtemp(prefix='karel_ds_')
    ke = os.path.join(REF, 'karel_env')
    common = ['--dir_name', 'tiny']
    run_script(os.path.join(ke, 'generator.py'),
               common + ['--num_train', '8', '--num_test', '2', '--num_val', '2', '--seed', '123',
                         '--num_demo_per_program', '4', '--min_demo_length', '3'], work)
    run_script(os.path.join(ke, 'append_demonstration.py'),
               ['--dir_name', os.path.join('datasets', 'tiny'), '--num_test_demo_per_program', '2',
                '--min_demo_length', '3'], work)
    ds_dir = os.path.join(work, 'datasets', 'tiny')
    run_script(os.path.join(ke, 'add_per.py'), ['--dir_name', ds_dir], work)

    out_dir = os.path.join(HERE, 'karel_dataset')
    shutil.rmtree(out_dir, ignore_errors=True)
    subprocess.check_call([sys.executable, os.path.join(REPO, 'tools', 'convert_karel_hdf5.py'), ds_dir, out_dir])

    # expected results from the reference's own reader
    sys.modules.pop('karel_env', None)
    from karel_env import dataset_karel as ref_ds        # noqa: E402  (reference module)
    ref_ds.rs = np.random.RandomState(123)               # the module-level stream, fresh
    tr, te, va = ref_ds.create_default_splits(ds_dir, num_k=3)
    blob = {'ids_train': np.array(tr.ids), 'ids_test': np.array(te.ids), 'ids_val': np.array(va.ids)}
    names = ['program', 'program_tokens', 's_h', 'test_s_h', 'a_h', 'a_h_tokens', 'test_a_h', 'test_a_h_tokens',
             'program_len', 'demo_len', 'test_demo_len', 'per', 'test_per']
    for ds in (tr, te, va):
        for id_ in ds.ids:
            for n, v in zip(names, ds.get_data(id_)):
                blob['%s/%s' % (id_, n)] = np.asarray(v)
    blob['info'] = np.array([tr.num_demo, tr.max_demo_len, tr.max_program_len, tr.num_program_tokens,
                             tr.num_action_tokens])
    np.savez_compressed(os.path.join(HERE, 'karel_dataset_expected.npz'), **blob)
    shutil.rmtree(work, ignore_errors=True)
    print('ok: %d train / %d test / %d val programs' % (len(tr.ids), len(te.ids), len(va.ids)))


if __name__ == '__main__':
    main()
