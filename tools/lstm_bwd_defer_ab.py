import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import lstm_persist_pair as T
from demo2program_amd.lib import load
from demo2program_amd import kernels as K
lib = load()
for v in (1 << 20, 5, 1 << 20, 5):
    lib.d2p_lstm_persist_set_bwd_defer(v if v < 100 else 0)
    print('defer_from', v if v < 100 else 'never', end=': ')
    T.triple()
# single 640-row sequence: 8 domains x 5 phases
f, b = T.mk(640, 20, seed=5)
K.lstm_seq_fwd_multi([f])
for v in (0, 5, 0, 5):
    lib.d2p_lstm_persist_set_bwd_defer(v)
    print('640 rows x 20, defer_from', v, '%.0f us' % T.timed(lambda: K.lstm_seq_bwd_multi([b])), 'err', K.lstm_persist_error(True))
