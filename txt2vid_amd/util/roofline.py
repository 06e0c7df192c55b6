"""North-star target line: "MFMA-roofline fraction on the TGANv2 3D-conv discriminator forward+backward at 64x64x16".

`d_fwdbwd_roofline` runs the Resnet3D discriminator (resnet3d.py:6-57) forward + full backward (data and weight
gradients, written the way the training step writes them: into the model's flat arena through `functional.GradSink`) on one
un-subsampled clip batch [B,1,16,64,64], fp32, and reports
  * conv kernels only: executed MAC flops / summed kernel time (HIP events around each launch), vs 157.3 TFLOP/s;
  * all-in: the same flops / wall time of forward+backward (pointwise, pooling, attention and launch gaps included).
"""
import ctypes as C
import os
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

    # the backward pass as the training step runs it (gan/trainer.py TrainStep): weight / bias gradients land in the model's flat
    # arena through the gradient sink — bias sums ride in the weight-gradient kernels, ONE reduce launch per pass
    from ..dist import model_arena
    sink = TF.GradSink([model_arena(D, TF.copy_into)])
    old_sink = TF.set_grad_sink(sink)

    def fwd_bwd():
        for p in D.parameters():         # (what the optimiser step does between two passes: a kept p.grad would make autograd ADD
            p.grad = None                # the arena view it is handed to the arena view it already holds)
        TF.grad_sink_reset()
        u, _, _ = D(x=x)
        TF.vec_sum(u.reshape(-1)).backward()
        TF.grad_sink_flush()

    # warm-up, eager timing and capture all run on ONE side stream (AccumulateGrad nodes remember their stream), as GraphedTrainStep does
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fwd_bwd()
        torch.cuda.synchronize()
        wall_eager = (time.perf_counter() - t0) / iters
    # the pass as the training loop runs it: captured once, replayed (no host time between launches)
    wall, mode = wall_eager, 'eager'
    if os.environ.get('T2V_D_ROOFLINE_EAGER') is None:
        for p in D.parameters():
            p.grad = None
        graph = torch.cuda.CUDAGraph()
        sink.frozen += 1
        try:
            with torch.cuda.graph(graph, stream=side, capture_error_mode='thread_local'):
                fwd_bwd()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            reps = max(iters, 10)
            t0 = time.perf_counter()
            for _ in range(reps):
                graph.replay()
            torch.cuda.synchronize()
            wall, mode = (time.perf_counter() - t0) / reps, 'hip-graph replay'
        except RuntimeError as e:            # (a failed capture must not take the benchmark line down: report the eager wall time)
            mode = 'eager (graph capture failed: %s)' % str(e).splitlines()[0][:120]
            torch.cuda.synchronize()
        finally:
            del graph
            sink.frozen -= 1
    lib().t2v_prof_begin(1 << 14)
    with torch.cuda.stream(side):
        for _ in range(iters):
            fwd_bwd()
    torch.cuda.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    out = (C.c_double * 18)()
    lib().t2v_prof_end(out, 6)          # kinds: 0 igemm, 1 wgrad, 2 wgrad reduce, 3 thin convs, 4 split-K reduce, 5 bf16 igemm
    TF.set_grad_sink(old_sink)
    out[0] += out[15]                   # (bf16-compute mode: its GEMM launches count as forward / data-gradient work)
    out[1] += out[16]
    out[2] += out[17]
    ig_ms, ig_fl, ig_n, wg_ms, wg_fl, wg_n, red_ms = [out[i] / iters for i in range(7)]
    ig_ms += (out[9] + out[12]) / iters          # thin convs and split-K passes belong to forward / data-gradient
    ig_fl += out[10] / iters
    flops, conv_ms = ig_fl + wg_fl, ig_ms + wg_ms + red_ms
    P = PEAK_FP32_MFMA_TFLOPS
    return {'workload': 'Resnet3D D forward+backward (parameter gradients through the gradient sink, as in the training step), x=[%d,1,%d,%d,%d] fp32, attention %s' % (batch, frames, size, size,
                                                                                          'on' if attn else 'off'),
            'conv_tflop_per_pass': flops / 1e12,
            'conv_kernels': {'ms': conv_ms, 'tflops': flops / conv_ms / 1e9, 'frac_of_fp32_mfma_peak': flops / conv_ms / 1e9 / P,
                             'igemm_fwd_dgrad': {'ms': ig_ms, 'tflops': ig_fl / ig_ms / 1e9, 'launches': ig_n},
                             'wgrad': {'ms': wg_ms + red_ms, 'tflops': wg_fl / (wg_ms + red_ms) / 1e9, 'launches': wg_n}},
            'all_in': {'wall_ms': wall * 1e3, 'tflops': flops / wall / 1e12, 'frac_of_fp32_mfma_peak': flops / wall / 1e12 / P,
                       'videos_per_s': batch / wall, 'launch_mode': mode, 'eager_wall_ms': wall_eager * 1e3},
            # SURVEY §8(d)'s algorithmic figure for this unit of work: 75.75 GFLOP per [1,1,16,64,64] sample forward+backward (what the
            # reference's convolutions execute; the pooled second convolutions here execute a quarter / an eighth of their share). Over
            # the wall time: the rate a reference-equivalent implementation would have to sustain to match — NOT matrix-pipe utilisation
            'survey_algorithmic': ({'gflop_per_sample': 75.75, 'tflops': 75.75e9 * batch / wall / 1e12,
                                    'frac_of_fp32_mfma_peak': 75.75e9 * batch / wall / 1e12 / P}
                                   if (frames, size) == (16, 64) else None),
            'peak_tflops': P}


PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def hbm_bound_lines(device=None, batch=32, iters=20):
    """The HBM-bound sub-operations of the iteration, each timed in isolation at the benchmark's shapes with
    events on the launch stream: achieved GB/s on the ALGORITHMIC bytes (what a single streaming pass must move) against the
    ~8 TB/s of HBM3E. {name: {bytes, us, achieved GB/s, frac}}. Timed under HIP-graph replay (no host time between launches)."""
    from .. import functional as TF
    from ..functional_pool import boxsum_raw, unbox_raw, pool_dgrad_raw
    dev = device or torch.device('cuda', torch.cuda.current_device())
    out = {}

    side = torch.cuda.Stream(device=dev)

    def timed(fn):
        """us per call with the host out of the picture: `iters` calls are captured into ONE HIP graph (the Python side of a call —
        group tables, allocations — costs more than these kernels run) and the replay is bracketed by events."""
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
        torch.cuda.synchronize()
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (3 * iters) * 1e3          # us per call

    def line(name, nbytes, us, what):
        out[name] = {'algorithmic_bytes': int(nbytes), 'us': us, 'achieved': nbytes / us / 1e3, 'unit': 'GB/s', 'peak': PEAK_HBM_GBS,
                     'frac': nbytes / us / 1e3 / PEAK_HBM_GBS, 'what': what}

    g = torch.Generator(device='cpu')
    g.manual_seed(5)
    # ---- Adam: the discriminator's 29 M parameters as 64 tensors, one multi-tensor launch (read p, g, m, v; write p, m, v)
    n_each = 29_040_000 // 64
    items = [tuple(torch.randn(n_each, generator=g).to(dev) for _ in range(4)) for _ in range(64)]
    for it in items:
        it[3].abs_()
    us = timed(lambda: TF.adam_step_multi(items, 2e-4, 0.5, 0.999, 1e-8, 3))
    line('adam_multi_k', 64 * n_each * 28, us, 'Adam on 29.0 M parameters (64 tensors, one launch): 28 B per parameter')
    del items
    # ---- weight repack: [512,256,3,3,3] -> wp[27][256][512]
    w = torch.randn(512, 256, 3, 3, 3, generator=g).to(dev)
    geom = TF.conv_geom(1, 256, 4, 4, 4, 512, 3, 3, 3)
    wp = torch.empty((27, 256 * 512), device=dev)
    us = timed(lambda: TF.check(TF.lib().t2v_pack_weight(TF._p(w), TF._p(wp), 512, 256, 27, geom.taps_c, 27, 0, TF._stream()), 'pack'))
    line('pack_weight_kernel', w.numel() * 8, us, 'repack of a [512,256,3,3,3] weight (read 4 B + write 4 B per weight)')
    del w, wp
    # ---- BatchNorm2d forward (training statistics + apply + ReLU) on the generator's [512,128,8,8] map
    x = torch.randn(batch * 16, 128, 8, 8, generator=g).to(dev)
    gam, bet = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    rm, rv = torch.zeros(128, device=dev), torch.ones(128, device=dev)
    us = timed(lambda: TF.batch_norm_act(x, gam, bet, rm, rv, True, 0.1, 1e-5, True))
    line('bn_train_fwd', x.numel() * 12, us, 'BatchNorm2d + ReLU, training mode, [%d,128,8,8]: statistics pass (4 B read) + apply pass (4 B read + 4 B write)' % (batch * 16))
    del x
    # ---- the Cin = 1 stem convolution over the 8 discriminator-step members (reads 4 B, writes 64 x 4 B per voxel)
    members = [(max(1, (2 * batch) >> l), 16 >> l, 8 << l) for l in range(4)] + [(max(1, batch >> l), 16 >> l, 8 << l) for l in range(4)]
    xs = [torch.randn(n, 1, d, s, s, generator=g).to(dev) for n, d, s in members]
    w1 = torch.nn.Parameter(torch.randn(64, 1, 3, 3, 3, generator=g).to(dev) * 0.1)       # (a Parameter: its packed form is cached)
    b1 = torch.zeros(64, device=dev)
    M = sum(t.numel() for t in xs)
    us = timed(lambda: TF.conv_group_raw(xs, w1, b1, False, 0))
    line('stem_conv1_cin1', M * (4 + 256), us, 'stem conv 1->64, 3x3x3, the 8 D-step members (M = %d voxels): 4 B read + 256 B written per voxel' % M)
    # ---- RenderBlock convolution ch -> 1 (thin kernel): [16,32,64,64]
    xr = torch.randn(max(1, batch // 8) * 2 * 2, 32, 1, 64, 64, generator=g).to(dev)
    wr = torch.nn.Parameter(torch.randn(1, 32, 1, 3, 3, generator=g).to(dev) * 0.1)
    us = timed(lambda: TF.conv_group_raw([xr], wr, None, False, 0))
    line('render_conv_cout1', xr.numel() * 4 + xr.numel() // 32 * 4, us, 'RenderBlock conv 32->1, 3x3 on [%d,32,64,64]: reads the map once, writes one channel' % xr.shape[0])
    del xr
    # ---- the pooled convolution's streaming passes on the stem's 8 members ([.,64,.,.,.]): box-sum and its adjoint
    hs = [torch.randn(n, 64, d, s, s, generator=g).to(dev) for n, d, s in members]
    tm = [2] * 8
    shapes = [tuple(h.shape) for h in hs]
    nb_in = sum(h.numel() for h in hs) * 4
    rts = boxsum_raw(hs, tm, True)
    us = timed(lambda: boxsum_raw(hs, tm, True))
    line('pool_boxsum_k', nb_in + sum(r.numel() for r in rts) * 4, us, 'box-sum (+ReLU) of the stem activation, 8 members: reads r, writes the padded r~')
    gz = [torch.randn(sh[0], 64, sh[2] // 2, sh[3] // 2, sh[4] // 2, generator=g).to(dev) for sh in shapes]
    planes = pool_dgrad_raw(gz, shapes, tm, torch.randn(64, 64, 3, 3, 3, generator=g).to(dev) * 0.05)
    us = timed(lambda: unbox_raw(planes, shapes, tm, masks=hs))

    def plane_entries_read(sh, tmode):
        """Class-plane entries pool_unbox_k reads for one member: an odd-parity class only lives on the first D' / H' / W' grid
        positions of its axis, and a tmode-2 member reads D' time positions of either time class (the padded planes'
        full extent over-counted the algorithmic bytes)."""
        n, c, d, h, w = sh
        dn, hn, wn = (d // 2 if tmode else 1), h // 2, w // 2
        tot = 0
        for ct in ((0, 1) if tmode else (0,)):
            dt = 1 if tmode == 0 else (dn if (tmode == 2 or ct) else dn + 1)
            for cy in (0, 1):
                for cx in (0, 1):
                    tot += dt * (hn if cy else hn + 1) * (wn if cx else wn + 1)
        return n * c * tot
    nb_planes = sum(plane_entries_read(sh, t) for sh, t in zip(shapes, tm)) * 4
    line('pool_unbox_k', nb_planes + 2 * nb_in, us, 'adjoint of the box-sum: reads the class-plane entries it needs and the ReLU mask, writes dL/dr')
    return out
