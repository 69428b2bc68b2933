#!/usr/bin/env python
"""Backward persistent kernel, per-phase clock breakdown of one workgroup (tools/trace_lstm_persist.py) with and
without the deferred form, at 640 rows (8 domains x 5 phases) and 320 rows (2-3 phases)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import trace_lstm_persist as TR  # noqa: E402
from demo2program_amd import build  # noqa: E402
from demo2program_amd.lib import call  # noqa: E402

build.build_library()
for M in (640, 320):
    for v in (0, 5):
        call.d2p_lstm_persist_set_bwd_defer(v)
        print('---- defer_from', v)
        TR.trace('bwd', M, 512, 20, 0, 0)
call.d2p_lstm_persist_set_bwd_defer(5)
