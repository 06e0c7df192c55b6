"""Shared plumbing of the autograd surface (txt2vid_amd.functional and its sub-modules): the current HIP stream, raw device
pointers and the dense-fp32-device-tensor check every kernel launch goes through."""
import ctypes as C

import torch

from ._lib import lib, check            # noqa: F401  (re-exported)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _c(t):
    """contiguous fp32 device tensor (kernels assume dense NCDHW)."""
    if t.dtype != torch.float32:
        raise TypeError('t2v kernels are fp32, got %s' % t.dtype)
    if not t.is_cuda:
        raise RuntimeError('t2v kernels need device tensors: the HIP path has no CPU fallback')
    return t if t.is_contiguous() else t.contiguous()
