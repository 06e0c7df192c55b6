"""North-star target line: "MFMA-roofline fraction on the TGANv2 3D-conv discriminator forward+backward at 64x64x16".

`d_fwdbwd_roofline` runs the Resnet3D discriminator (resnet3d.py:6-57) forward + full backward (data and weight
gradients) on one un-subsampled clip batch [B,1,16,64,64], fp32, and reports
  * conv kernels only: executed MAC flops / summed kernel time (HIP events around each launch), vs 157.3 TFLOP/s;
  * all-in: the same flops / wall time of forward+backward (pointwise, pooling, attention and launch gaps included).
"""
import ctypes as C
import time

import torch

PEAK_FP32_MFMA_TFLOPS = 157.3


def d_fwdbwd_roofline(batch=32, iters=5, frames=16, size=64, attn=True, device=None):
    from .. import functional as TF
    from .._lib import lib
    from ..models.resnet3d import Resnet3D
    from .torch.init import init
    dev = device or torch.device('cuda', torch.cuda.current_device())
    rng = torch.random.get_rng_state()
    torch.manual_seed(100)
    D = Resnet3D(num_channels=1, with_attn=attn)
    init(D, 'xavier')
    D.to(dev)
    x = (torch.rand(batch, 1, frames, size, size) * 2 - 1).to(dev)
    torch.random.set_rng_state(rng)

    def fwd_bwd():
        for p in D.parameters():
            p.grad = None
        u, _, _ = D(x=x)
        TF.vec_sum(u.reshape(-1)).backward()

    for _ in range(2):
        fwd_bwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fwd_bwd()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    lib().t2v_prof_begin(1 << 14)
    for _ in range(iters):
        fwd_bwd()
    torch.cuda.synchronize()
    out = (C.c_double * 18)()
    lib().t2v_prof_end(out, 6)          # kinds: 0 igemm, 1 wgrad, 2 wgrad reduce, 3 thin convs, 4 split-K reduce, 5 bf16 igemm
    out[0] += out[15]                   # (bf16-compute mode: its GEMM launches count as forward / data-gradient work)
    out[1] += out[16]
    out[2] += out[17]
    ig_ms, ig_fl, ig_n, wg_ms, wg_fl, wg_n, red_ms = [out[i] / iters for i in range(7)]
    ig_ms += (out[9] + out[12]) / iters          # thin convs and split-K passes belong to forward / data-gradient
    ig_fl += out[10] / iters
    flops, conv_ms = ig_fl + wg_fl, ig_ms + wg_ms + red_ms
    P = PEAK_FP32_MFMA_TFLOPS
    return {'workload': 'Resnet3D D forward+backward, x=[%d,1,%d,%d,%d] fp32, attention %s' % (batch, frames, size, size,
                                                                                          'on' if attn else 'off'),
            'conv_tflop_per_pass': flops / 1e12,
            'conv_kernels': {'ms': conv_ms, 'tflops': flops / conv_ms / 1e9, 'frac_of_fp32_mfma_peak': flops / conv_ms / 1e9 / P,
                             'igemm_fwd_dgrad': {'ms': ig_ms, 'tflops': ig_fl / ig_ms / 1e9, 'launches': ig_n},
                             'wgrad': {'ms': wg_ms + red_ms, 'tflops': wg_fl / (wg_ms + red_ms) / 1e9, 'launches': wg_n}},
            'all_in': {'wall_ms': wall * 1e3, 'tflops': flops / wall / 1e12, 'frac_of_fp32_mfma_peak': flops / wall / 1e12 / P,
                       'videos_per_s': batch / wall},
            'peak_tflops': P}
