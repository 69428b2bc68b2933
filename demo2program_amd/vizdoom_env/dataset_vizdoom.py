"""ViZDoom dataset reader with the reference's surface (vizdoom_env/dataset_vizdoom.py):
`Dataset(ids, dataset_path, name, num_k, is_train)`, `.get_data(id)` -> the same 17 arrays,
`create_default_splits`, `all_ids` (same seeded shuffles, so the same split order).

On disk it reads the h5py-free layout written by tools/convert_vizdoom_hdf5.py from a reference
`data.hdf5`: memory-mapped whole-dataset arrays with the frames stored ragged as uint8 (the
published datasets are 250-500 GB of int16 frames).  Padding / one-hot rules restated from
dataset_vizdoom.py:49-139: programs one-hot over [num_program_tokens, max_program_len];
demonstrations, perception vectors and initial positions zero-padded to the dataset maxima;
action sequences closed by the extra <e> token; only the first num_k seen demonstrations are
returned, all held-out ones.
"""
import json
import os.path as osp

import numpy as np

rs = np.random.RandomState(123)      # module-level stream, as the reference's (dataset_vizdoom.py:12)

_ARRAYS = ('program', 'program_len', 'frames', 'frame_offset', 's_h_len', 'test_s_h_len', 'a_h', 'test_a_h',
           'a_h_len', 'test_a_h_len', 'a_h_width', 'test_a_h_width', 's_h_width', 'test_s_h_width',
           'p_v_h', 'test_p_v_h', 'init_pos', 'init_pos_len', 'test_init_pos', 'test_init_pos_len')


class _Store(object):
    """Memory-mapped arrays of one converted dataset directory (shared by the three splits)."""
    _cache = {}

    def __init__(self, path):
        if not osp.exists(osp.join(path, 'data_info.json')):
            hint = ''
            if osp.exists(osp.join(path, 'data.hdf5')):
                hint = (' (found data.hdf5: convert it first with `python3.9 tools/convert_vizdoom_hdf5.py '
                        '%s <out_dir>` -- needs h5py)' % path)
            raise IOError('no converted ViZDoom dataset under %s%s' % (path, hint))
        with open(osp.join(path, 'data_info.json')) as f:
            self.info = json.load(f)
        if 'vizdoom_pos_keys' not in self.info:
            raise IOError('%s is not a ViZDoom dataset (no vizdoom_pos_keys in data_info)' % path)
        with open(osp.join(path, 'ids.txt')) as f:
            self.ids = [s.strip() for s in f.readlines() if s.strip()]
        self.index = {id_: i for i, id_ in enumerate(self.ids)}
        self.arr = {n: np.load(osp.join(path, n + '.npy'), mmap_mode='r') for n in _ARRAYS}

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
        self.is_train = is_train
        self.num_k = num_k
        self.data = _Store.open(dataset_path)
        info = self.data.info
        self.num_demo = int(info['num_demo_per_program'])
        self.max_demo_len = int(info['max_demo_length'])
        self.max_program_len = int(info['max_program_length'])
        self.num_program_tokens = int(info['num_program_tokens'])
        self.num_action_tokens = int(info['num_action_tokens'])
        self.vizdoom_pos_keys = list(info['vizdoom_pos_keys'])
        self.vizdoom_max_init_pos_len = int(info['vizdoom_max_init_pos_len'])
        self.perception_type = info['perception_type']
        self.level = info.get('level', 'not_simple')
        self.k = int(info['num_demo_per_program'])
        self.test_k = int(info['num_test_demo_per_program'])
        self.s_h_h, self.s_h_w, self.s_h_c = int(info['s_h_h']), int(info['s_h_w']), int(info['s_h_c'])

    def _frames(self, i, col0, count, dtype):
        """Demonstrations col0 .. col0+count of program i -> [count, max_demo_len, h, w, c], zero
        past each demonstration's length."""
        A = self.data.arr
        lens = np.concatenate([A['s_h_len'][i], A['test_s_h_len'][i]])
        out = np.zeros((count, self.max_demo_len, self.s_h_h, self.s_h_w, self.s_h_c), dtype)
        for d in range(count):
            o, m = int(A['frame_offset'][i, col0 + d]), int(lens[col0 + d])
            out[d, :m] = A['frames'][o:o + m]
        return out

    def _actions(self, tokens, width):
        """-> (one-hot [D, max_demo_len, A+1] bool, argmax tokens).  Reference behaviour kept on
        purpose (dataset_vizdoom.py:73-85): a program's action sequences are stored as ONE
        zero-padded array as wide as the longest of all its demonstrations
        (generator.py:193-197), and the reader one-hots each padded row whole -- action 0 in the
        padding, <e> after the stored width for every demonstration of the program."""
        D = tokens.shape[0]
        A = self.num_action_tokens
        a_h = np.zeros((D, self.max_demo_len, A + 1), bool)
        for d in range(D):
            a_h[d, np.arange(width), np.asarray(tokens[d, :width], dtype=np.int64)] = True
            a_h[d, width, A] = True
        return a_h, np.argmax(a_h, axis=2)

    def _padded(self, x, stored, axis_len):
        out = np.zeros((x.shape[0], axis_len) + x.shape[2:], x.dtype)
        t = min(int(stored), axis_len)
        out[:, :t] = x[:, :t]
        return out

    def get_data(self, id, order=None, frames_dtype=np.int16):
        """-> program [V, L] bool, program_tokens [L], s_h [num_k, T, h, w, c] int16 (the HDF5's
        dtype; `frames_dtype` is this build's shortcut to uint8), test_s_h, a_h [num_k, T, A+1],
        a_h_tokens [num_k, T], test_a_h, test_a_h_tokens, program_len [1] f32, demo_len [num_k],
        test_demo_len, per [num_k, T, P] bool, test_per, init_pos [num_k, K, Pmax, 2] int32,
        init_pos_len [num_k, K], test_init_pos, test_init_pos_len."""
        i = self.data.index[id]
        A = self.data.arr
        n = int(A['program_len'][i])
        tokens = np.asarray(A['program'][i, :n])
        program = np.zeros((self.num_program_tokens, self.max_program_len), bool)
        program[np.asarray(tokens, dtype=np.int64), np.arange(n)] = True
        padded_tokens = np.zeros(self.max_program_len, tokens.dtype)
        padded_tokens[:n] = tokens
        k = min(self.num_k, self.k)
        demo = self._frames(i, 0, k, frames_dtype)
        test_demo = self._frames(i, self.k, self.test_k, frames_dtype)
        a_h, a_tok = self._actions(np.asarray(A['a_h'][i][:k]), int(A['a_h_width'][i]))
        ta_h, ta_tok = self._actions(np.asarray(A['test_a_h'][i]), int(A['test_a_h_width'][i]))
        per = self._padded(np.asarray(A['p_v_h'][i][:k]), A['s_h_width'][i], self.max_demo_len)
        test_per = self._padded(np.asarray(A['test_p_v_h'][i]), A['test_s_h_width'][i], self.max_demo_len)
        P = self.vizdoom_max_init_pos_len
        init_pos = np.array(A['init_pos'][i][:k, :, :P])
        test_init_pos = np.array(A['test_init_pos'][i][:, :, :P])
        return (program, padded_tokens, demo, test_demo, a_h, a_tok, ta_h, ta_tok,
                np.array([n], dtype=np.float32), np.array(A['s_h_len'][i][:k]), np.array(A['test_s_h_len'][i]),
                per, test_per, init_pos, np.array(A['init_pos_len'][i][:k]),
                test_init_pos, np.array(A['test_init_pos_len'][i]))

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
