"""Batching for the Karel dataset: the role of karel_env/input_ops_karel.py (a TF queue of
py_func-loaded examples behind tf.train.shuffle_batch / tf.train.batch with 16 loader threads).

`create_input_ops(dataset, batch_size, is_training, shuffle)` returns `(input_ops, batch)` like
the reference; `batch.next()` yields the batch_chunk dictionary the model's get_feed_dict takes:
the 14 keys with the dtypes of input_ops_karel.py:69-75 (float32 frames / one-hots / lengths,
int32 tokens).  Examples are assembled by a pool of loader threads ahead of the consumer (the
reference's queue runners); shuffling draws from a seeded reservoir of `min_after_dequeue`
examples like tf.train.shuffle_batch (the reference's order is thread-timing dependent and not
reproducible; this one is, given the seed).
"""
import queue
import threading

import numpy as np

KEYS = ('program', 'program_tokens', 's_h', 'test_s_h', 'a_h', 'a_h_tokens', 'test_a_h', 'test_a_h_tokens',
        'program_len', 'demo_len', 'test_demo_len', 'per', 'test_per')
_DTYPES = dict(program=np.float32, program_tokens=np.int32, s_h=np.float32, test_s_h=np.float32,
               a_h=np.float32, a_h_tokens=np.int32, test_a_h=np.float32, test_a_h_tokens=np.int32,
               program_len=np.float32, demo_len=np.float32, test_demo_len=np.float32,
               per=np.float32, test_per=np.float32)


def check_data_id(dataset, data_id):
    if not data_id:
        return
    wrong = [i for i in data_id if i not in dataset.data.index]
    if wrong:
        raise RuntimeError('There are %d invalid ids, including %s' % (len(wrong), wrong[:5]))


def load_example(dataset, id_, frames_dtype=np.float32, keys=KEYS, dtypes=_DTYPES):
    out = {'id': id_}
    for n, v in zip(keys, dataset.get_data(id_)):
        dt = frames_dtype if n in ('s_h', 'test_s_h') else dtypes[n]
        out[n] = np.asarray(v).astype(dt)
    return out


class BatchIterator(object):
    """Endless stream of batches over `data_id` (cycled, like string_input_producer)."""

    def __init__(self, dataset, batch_size, data_id, shuffle, num_threads=4, seed=123, prefetch=4,
                 frames_dtype=np.float32, load=load_example, keys=KEYS):
        self.dataset, self.batch_size, self.ids = dataset, batch_size, list(data_id)
        self._load, self.keys = load, keys
        self.shuffle = shuffle
        self.rng = np.random.RandomState(seed)
        self.frames_dtype = frames_dtype
        capacity = 2 * batch_size * 16                                   # input_ops_karel.py:108-109
        self.min_after_dequeue = min(int(capacity * 0.75), 1024) if shuffle else 0
        self._pool = []
        self._cursor = 0
        self._q = queue.Queue(maxsize=max(prefetch, 1))
        self._stop = False
        self._lock = threading.Lock()
        self._thread = None
        self.num_threads = num_threads

    # -- example order (deterministic, single producer) ---------------------------------------
    def _next_id(self):
        id_ = self.ids[self._cursor % len(self.ids)]
        self._cursor += 1
        return id_

    def _draw_ids(self):
        if not self.shuffle:
            return [self._next_id() for _ in range(self.batch_size)]
        while len(self._pool) < self.min_after_dequeue + self.batch_size:
            self._pool.append(self._next_id())
        picked = []
        for _ in range(self.batch_size):
            j = self.rng.randint(len(self._pool))
            self._pool[j], self._pool[-1] = self._pool[-1], self._pool[j]
            picked.append(self._pool.pop())
        return picked

    def _assemble(self, ids):
        ex = [self._load(self.dataset, i, self.frames_dtype) for i in ids]
        batch = {n: np.stack([e[n] for e in ex]) for n in self.keys}
        batch['id'] = np.array(ids)
        return batch

    def _run(self):
        while not self._stop:
            batch = self._assemble(self._draw_ids())
            while not self._stop:
                try:
                    self._q.put(batch, timeout=0.2)
                    break
                except queue.Full:
                    continue

    def next(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self._q.get()

    def next_sync(self):
        """One batch assembled on the calling thread (no prefetch thread; for tests)."""
        return self._assemble(self._draw_ids())

    def close(self):
        self._stop = True


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
                                    frames_dtype=frames_dtype)
