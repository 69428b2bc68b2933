"""Karel dataset reader with the reference's surface (karel_env/dataset_karel.py):
`Dataset(ids, dataset_path, name, num_k, is_train)`, `.get_data(id)` -> the same 13 arrays,
`create_default_splits`, `all_ids` (same seeded shuffles, so the same split order).

On disk it reads the h5py-free layout written by tools/convert_karel_hdf5.py from a reference
`data.hdf5` (this build's training interpreter has no h5py): whole-dataset .npy arrays, memory-
mapped, bit-packed frames.  Padding / one-hot rules restated from dataset_karel.py:38-115:
programs one-hot over [num_program_tokens, max_program_len]; demonstrations zero-padded to
max_demo_len; action sequences closed by the extra <e> token and zero after it; only the first
num_k seen demonstrations are returned, all held-out ones.
"""
import json
import os.path as osp

import numpy as np

rs = np.random.RandomState(123)      # module-level stream, as the reference's (dataset_karel.py:11)

_ARRAYS = ('program', 'program_len', 's_h', 's_h_len', 'a_h', 'a_h_len', 'per',
           'test_s_h', 'test_s_h_len', 'test_a_h', 'test_a_h_len', 'test_per')


class _Store(object):
    """Memory-mapped arrays of one converted dataset directory (shared by the three splits)."""
    _cache = {}

    def __init__(self, path):
        if not osp.exists(osp.join(path, 'data_info.json')):
            hint = ''
            if osp.exists(osp.join(path, 'data.hdf5')):
                hint = (' (found data.hdf5: convert it first with `python3.9 tools/convert_karel_hdf5.py '
                        '%s <out_dir>` -- needs h5py)' % path)
            raise IOError('no converted Karel dataset under %s%s' % (path, hint))
        with open(osp.join(path, 'data_info.json')) as f:
            self.info = json.load(f)
        with open(osp.join(path, 'ids.txt')) as f:
            self.ids = [s.strip() for s in f.readlines() if s.strip()]
        self.index = {id_: i for i, id_ in enumerate(self.ids)}
        self.arr = {n: np.load(osp.join(path, n + '.npy'), mmap_mode='r') for n in _ARRAYS
                    if osp.exists(osp.join(path, n + '.npy'))}

    @classmethod
    def open(cls, path):
        path = osp.abspath(path)
        if path not in cls._cache:
            cls._cache[path] = cls(path)
        return cls._cache[path]


class Dataset(object):

    def __init__(self, ids, dataset_path, name='default', num_k=10, is_train=True):
        self._ids = list(ids)
        self.name = name
        self.num_k = num_k
        self.is_train = is_train
        self.data = _Store.open(dataset_path)
        info = self.data.info
        self.dsl_type = info['dsl_type']
        self.num_demo = int(info['num_demo_per_program'])
        self.max_demo_len = int(info['max_demo_length'])
        self.max_program_len = int(info['max_program_length'])
        self.num_program_tokens = int(info['num_program_tokens'])
        self.num_action_tokens = int(info['num_action_tokens'])
        self.env_type = info.get('env_type')
        self.h, self.w, self.depth = int(info['height']), int(info['width']), int(info['depth'])

    def _frames(self, packed, lens):
        """packed [D, Tdisk, bytes] -> bool [D, max_demo_len, h, w, depth] (zero past each length)."""
        D = packed.shape[0]
        bits = np.unpackbits(np.asarray(packed), axis=-1)[..., :self.h * self.w * self.depth]
        out = np.zeros((D, self.max_demo_len, self.h, self.w, self.depth), bool)
        t = min(bits.shape[1], self.max_demo_len)
        out[:, :t] = bits[:, :t].reshape(D, t, self.h, self.w, self.depth)
        return out

    def _actions(self, tokens, lens):
        """-> (one-hot [D, max_demo_len, A+1] bool, argmax tokens).

        Reference behaviour kept on purpose (dataset_karel.py:69-79): the HDF5 stores a program's
        action sequences as ONE zero-padded array whose width is the longest of its demos
        (generator.py:113-115), and the reader one-hots each padded row whole -- `len(a_h_tokens)`
        is that width, not the demo's own action count.  So a shorter demo carries action 0
        ('move') in its padding and the <e> token sits after the padded width for every demo of
        the program."""
        D = tokens.shape[0]
        A = self.num_action_tokens
        a_h = np.zeros((D, self.max_demo_len, A + 1), bool)
        n = int(np.max(lens)) if D else 0               # stored width of this program's array
        for d in range(D):
            a_h[d, np.arange(n), np.asarray(tokens[d, :n], dtype=np.int64)] = True
            a_h[d, n, A] = True
        return a_h, np.argmax(a_h, axis=2)

    def _per(self, per):
        D = per.shape[0]
        out = np.zeros((D, self.max_demo_len, per.shape[2]), np.float64)
        t = min(per.shape[1], self.max_demo_len)
        out[:, :t] = per[:, :t]
        return out

    def get_data(self, id, order=None):
        """-> program [V, L] bool, program_tokens [L], s_h [num_k, T, h, w, 16], test_s_h [Dt, ...],
        a_h [num_k, T, A+1], a_h_tokens [num_k, T], test_a_h, test_a_h_tokens, program_len [1] f32,
        demo_len [num_k], test_demo_len [Dt], per [num_k, T, 5], test_per [Dt, T, 5]."""
        i = self.data.index[id]
        A = self.data.arr
        n = int(A['program_len'][i])
        tokens = np.asarray(A['program'][i, :n])
        program = np.zeros((self.num_program_tokens, self.max_program_len), bool)
        program[np.asarray(tokens, dtype=np.int64), np.arange(n)] = True
        padded_tokens = np.zeros(self.max_program_len, tokens.dtype)
        padded_tokens[:n] = tokens
        k = self.num_k
        s_h = self._frames(A['s_h'][i], A['s_h_len'][i])
        test_s_h = self._frames(A['test_s_h'][i], A['test_s_h_len'][i])
        a_h, a_tok = self._actions(A['a_h'][i], A['a_h_len'][i])
        ta_h, ta_tok = self._actions(A['test_a_h'][i], A['test_a_h_len'][i])
        return (program, padded_tokens, s_h[:k], test_s_h, a_h[:k], a_tok[:k], ta_h, ta_tok,
                np.array([n], dtype=np.float32), np.asarray(A['s_h_len'][i])[:k].copy(),
                np.asarray(A['test_s_h_len'][i]).copy(), self._per(A['per'][i])[:k], self._per(A['test_per'][i]))

    @property
    def ids(self):
        return self._ids

    def __len__(self):
        return len(self.ids)

    def __repr__(self):
        return 'Dataset (%s, %d examples)' % (self.name, len(self))


def all_ids(dataset_path):
    st = _Store.open(dataset_path)
    num_train, num_test, num_val = (int(st.info[n]) for n in ('num_train', 'num_test', 'num_val'))
    ids_total = list(st.ids)
    ids_train = ids_total[:num_train]
    ids_test = ids_total[num_train: num_train + num_test]
    ids_val = ids_total[num_train + num_test: num_train + num_test + num_val]
    rs.shuffle(ids_train)
    rs.shuffle(ids_test)
    rs.shuffle(ids_val)
    return ids_train, ids_test, ids_val


def create_default_splits(dataset_path, num_k=10, is_train=True):
    ids_train, ids_test, ids_val = all_ids(dataset_path)
    return (Dataset(ids_train, dataset_path, name='train', num_k=num_k, is_train=is_train),
            Dataset(ids_test, dataset_path, name='test', num_k=num_k, is_train=is_train),
            Dataset(ids_val, dataset_path, name='val', num_k=num_k, is_train=is_train))
