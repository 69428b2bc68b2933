"""Batching for the ViZDoom dataset: the role of vizdoom_env/input_ops_vizdoom.py (a TF queue of
py_func-loaded examples behind tf.train.shuffle_batch / tf.train.batch).

Same machinery as the Karel one (karel_env/input_ops_karel.py here): `create_input_ops(dataset,
batch_size, is_training, shuffle)` -> `(input_ops, batch)`; `batch.next()` yields the batch_chunk
dictionary with the 18 keys and dtypes of input_ops_vizdoom.py:71-78 -- the Karel ones plus the
int32 `init_pos`, `init_pos_len`, `test_init_pos`, `test_init_pos_len` the execution metric
hands to the game engine.  With `frames_dtype=np.uint8` the frames go from the memory map to the
batch without the float32 detour (the model widens them on the GPU).
"""
import numpy as np

from ..karel_env.input_ops_karel import KEYS as _KAREL_KEYS, _DTYPES as _KAREL_DTYPES, BatchIterator, check_data_id

KEYS = _KAREL_KEYS + ('init_pos', 'init_pos_len', 'test_init_pos', 'test_init_pos_len')
_DTYPES = dict(_KAREL_DTYPES, init_pos=np.int32, init_pos_len=np.int32, test_init_pos=np.int32,
               test_init_pos_len=np.int32)


def load_example(dataset, id_, frames_dtype=np.float32):
    native = np.uint8 if np.dtype(frames_dtype) == np.uint8 else np.int16
    out = {'id': id_}
    for n, v in zip(KEYS, dataset.get_data(id_, frames_dtype=native)):
        dt = frames_dtype if n in ('s_h', 'test_s_h') else _DTYPES[n]
        out[n] = np.asarray(v).astype(dt, copy=False)
    return out


def create_input_ops(dataset, batch_size, num_threads=16, is_training=False, data_id=None, scope='inputs',
                     shuffle=True, seed=123, frames_dtype=np.float32):
    """-> (input_ops, batch): `input_ops` maps each key to the (shape, dtype) of one example (what
    the reference's placeholders carry); `batch.next()` yields batch_chunk dictionaries."""
    if data_id is None:
        data_id = dataset.ids
    else:
        check_data_id(dataset, data_id)
    one = load_example(dataset, data_id[0], frames_dtype)
    input_ops = {n: (tuple(one[n].shape), one[n].dtype) for n in KEYS}
    input_ops['id'] = ((), np.dtype('O'))
    return input_ops, BatchIterator(dataset, batch_size, data_id, shuffle, num_threads=num_threads, seed=seed,
                                    frames_dtype=frames_dtype, load=load_example, keys=KEYS)
