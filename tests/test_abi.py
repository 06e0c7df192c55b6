"""CPU: the C-ABI library builds for gfx950 without a GPU, loads, and exports every symbol that
include/t2v_hip.h declares; the ctypes table binds exactly that set. No kernel is launched here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 't2v_hip.h')
SO = os.path.join(ROOT, 'txt2vid_amd', 'csrc', 'libt2v_hip.so')


def declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(t2v_[a-z0-9_]+)\s*\(', txt)))


@pytest.fixture(scope='module')
def built():
    if not os.path.exists(SO):
        subprocess.check_call(['make', '-C', os.path.dirname(SO), '-j4'])
    return SO


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ('t2v_conv_fwd', 't2v_conv_wgrad', 't2v_pack_weight', 't2v_bn_stats', 't2v_softmax', 't2v_bmm',
                 't2v_lstm_gates', 't2v_rsgan', 't2v_row_sqnorm', 't2v_adam', 't2v_pyramid_gather'):
        assert must in syms


def test_library_exports_every_declared_symbol(built):
    out = subprocess.check_output(['nm', '-D', '--defined-only', built]).decode()
    exported = set(re.findall(r'\bT\s+(t2v_[a-z0-9_]+)', out))
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, missing


def test_ctypes_table_matches_header(built):
    from txt2vid_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == declared_symbols()
    l = _lib.lib()                       # binds argtypes for every symbol; raises if one is absent
    assert l.t2v_version().startswith(b't2v_hip')


def test_geometry_struct_layout_matches_header():
    """ctypes mirror of t2v_conv_geom: 7 int32 + 3*27 int8 + 3 pad = 112 bytes."""
    import ctypes
    from txt2vid_amd._lib import ConvGeom
    assert ctypes.sizeof(ConvGeom) == 7 * 4 + 3 * 27 + 3
    from txt2vid_amd._lib import ConvGroup
    assert ctypes.sizeof(ConvGroup) == 168                             # 3 pointers + 9 int32 (incl. dstride, ydstride, yoff, Dy) + 4 x 27 int8


def test_argument_validation_without_gpu(built):
    """Bad arguments are rejected on the host side (negative status) before any launch."""
    import ctypes as C
    from txt2vid_amd._lib import lib, ConvGeom
    g = ConvGeom()
    assert lib().t2v_conv_fwd_ws_floats(C.byref(g)) < 0          # all-zero geometry
    assert lib().t2v_conv_fwd(None, None, None, None, None, C.byref(g), 0, None) < 0
    assert lib().t2v_relu(None, None, -1, None) < 0
    assert lib().t2v_bmm(None, None, None, 1, 1, 1, 1, 0, 0, 0, None) < 0


def test_product_ops_refuse_cpu_tensors(built):
    """There is no CPU / PyTorch fallback behind the product ops: host tensors raise."""
    import torch
    from txt2vid_amd import functional as TF
    with pytest.raises(RuntimeError):
        TF.conv(torch.zeros(1, 4, 2, 2, 2), torch.zeros(4, 4, 3, 3, 3), None)
    with pytest.raises(RuntimeError):
        TF.relu(torch.zeros(4))
