"""The C-ABI shared library loads (no GPU needed) and exports every symbol that
include/d2p.h declares, with the argument counts the ctypes table uses."""
import os
import re

from demo2program_amd import lib as d2plib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_prototypes():
    src = open(os.path.join(ROOT, 'include', 'd2p.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(?:int|size_t|const char\*)\s+(d2p_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',') if a.strip()])
        protos[name] = n
    return protos


def test_header_matches_ctypes_table():
    protos = _header_prototypes()
    assert len(protos) >= 40
    assert set(protos) == set(d2plib.SIGNATURES), (
        set(protos) ^ set(d2plib.SIGNATURES))
    for name, n in protos.items():
        assert len(d2plib.SIGNATURES[name][1]) == n, name


def test_library_loads_and_exports_everything(d2p_lib):
    for name in _header_prototypes():
        assert hasattr(d2p_lib, name), name
    assert d2p_lib.d2p_version() == 2        # (2: d2p_gemm_set_option bit layout, include/d2p.h)
    assert d2p_lib.d2p_last_error() is not None


def test_product_library_carries_no_diagnostic_stamps(d2p_lib):
    """tools/lstm_launch_stamps.py runs against a DIAGNOSTIC build of lstm_persist.hip (-DD2P_PS_STAMPS,
    build.build_stamps_library): its two entry points exist there only -- the product library exports neither, and the
    header declares neither."""
    for name in ('d2p_lstm_persist_set_stamps', 'd2p_lstm_persist_stamp_launches'):
        assert not hasattr(d2p_lib, name), name
        assert name not in _header_prototypes()
    src = open(os.path.join(ROOT, 'demo2program_amd', 'csrc', 'lstm_persist.hip')).read()
    body = src[src.index('#ifdef D2P_PS_STAMPS'):src.index('#else   // (every macro expands to NOTHING')]
    assert 'd2p_lstm_persist_set_stamps' in body and src.count('d2p_lstm_persist_set_stamps') == 1


def test_workspace_queries_need_no_gpu(d2p_lib):
    assert d2p_lib.d2p_lstm_ws_bytes(320, 512) >= 3 * 320 * 512 * 4
    assert d2p_lib.d2p_gemm_ws_bytes(144, 16, 102400) > 0      # conv1 wgrad needs split-K
    assert d2p_lib.d2p_gemm_ws_bytes(6400, 2048, 512) == 0
    assert d2p_lib.d2p_bn_ws_bytes(6400 * 16, 16, 10) > 0
    assert d2p_lib.d2p_xent_ws_bytes(10) == 10 * 64 * 2 * 4


def test_argument_errors_do_not_touch_the_gpu(d2p_lib):
    # negative size -> D2P_EINVAL before any launch
    rc = d2p_lib.d2p_gemm_f32_nn(-1, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, None, 0, None)
    assert rc == -1
    assert b'negative' in d2p_lib.d2p_last_error()
    rc = d2p_lib.d2p_bn_group_fwd(10, 4, 3, 1, None, None, None, None, None, None, None, None, None, 0.9, None, 0, None)
    assert rc == -1
