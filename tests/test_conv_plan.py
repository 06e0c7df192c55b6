"""CPU: the launch-plan queries of the C ABI (host arithmetic, no GPU) say which kernel instantiation a convolution runs
on. This file pins (1) what the BENCHMARK shapes select — the 256x64 strip GEMM and the 171-way split weight gradient that
the round-1 review found in no parity test — and (2) that EVERY instantiation compiled into libt2v_hip.so is reached by at
least one case of tests/conv_cases.py, which tests/test_ops_gpu.py checks against torch on the GPU box."""
import itertools

import conv_cases as cc


def test_benchmark_shapes_select_the_big_tile_paths():
    """BASELINE configs[1] at per-GPU batch 32 (what bench.py times): stem conv2 over the 8 discriminator-step members."""
    name, cin, cout, k, members, _ = cc.GROUPED_CASES[0]
    assert name == 'stem_conv2_B32_8members'
    assert sum(n * d * h * w for n, d, h, w in members) == 393216
    assert cc.fwd_plan(members, cin, cout, k) == ('strip3', 256, 64, 16, 1, 1, 2, 1)         # conv_igemm_strip3_kernel<256,16,1,true,true> (double-buffered)
    assert cc.fwd_plan(members, cout, cin, k) == ('strip3', 256, 64, 16, 1, 1, 2, 1)         # its data gradient
    w = cc.wgrad_plan(members, cin, cout, k)
    assert w[0] == 'rows3' and w[4] == 'reduce_small' and w[1] >= 16                          # conv_wgrad3_kernel, many-splits reduce
    # the gradient-penalty members alone (M = 131072) still fill 512 tiles of 256 voxels
    _, cin, cout, k, members, _ = cc.GROUPED_CASES[1]
    assert cc.fwd_plan(members, cin, cout, k)[:4] == ('strip3', 256, 64, 16)
    # 64 -> 128 at M = 49152: 128x64 tiles forward, 64x64 tiles for the data gradient
    _, cin, cout, k, members, _ = cc.GROUPED_CASES[3]
    assert sum(n * d * h * w for n, d, h, w in members) == 49152
    assert cc.fwd_plan(members, cin, cout, k)[:4] == ('strip3', 128, 64, 32)
    assert cc.fwd_plan(members, cout, cin, k)[:7] == ('strip3', 64, 64, 16, 1, 1, 2)      # the double-buffered form: 16-channel rounds


# every instantiation the launchers in conv.hip can select (launch_conv_t + the thin kernels)
_TILES = [(128, 32, 32), (128, 32, 16), (256, 64, 16), (128, 64, 32), (128, 64, 16), (64, 64, 32), (64, 64, 16)]
ALL_FWD = set()
for bm, bn, bk in _TILES:
    for vecb in (1, 0):
        ALL_FWD.add(('igemm', bm, bn, bk, 1, vecb, 1))                       # conv_igemm_kernel<BM,BN,*,BK,true,VECB>
        if (bm, bn, bk) in ((64, 64, 32), (128, 64, 32), (256, 64, 16)):
            ALL_FWD.add(('strip3', bm, bn, bk, 1, vecb, 1))                  # conv_igemm_strip3_kernel<BM,VECB>: 3 dx taps per round
        elif bk == 32 or bm == 256:
            ALL_FWD.add(('strip', bm, bn, bk, 1, vecb, 1))                   # conv_igemm_strip_kernel<...,VECB,1>
ALL_FWD.add(('strip3', 64, 64, 16, 1, 1, 2))                                 # strip3<64>'s double-buffered form (every member three taps wide)
ALL_FWD.add(('strip3', 256, 64, 16, 1, 1, 2))                                # ... and strip3<256>'s
for bm, bn in ((128, 32), (128, 64), (64, 64)):
    ALL_FWD.add(('igemm', bm, bn, 16, 0, 0, 1))                              # generic-K: conv_igemm_kernel<BM,BN,*,16,false,false>
ALL_FWD |= {('stem', 256, 1, 0, 0, 0, 0), ('stem', 256, 3, 0, 0, 0, 0),          # conv_stem_kernel<1 / 3>: the clips' first convolution
            ('thin', 256, 1, 0, 0, 0, 0), ('thin', 256, 4, 0, 0, 0, 0), ('thin2', 256, 1, 0, 0, 0, 0),
            ('linear', 0, 1, 0, 0, 0, 0), ('linear', 0, 4, 0, 0, 0, 0)}
ALL_WGRAD = set(itertools.product(('taps', 'cols', 'rows3', 'gemm', 'thin'), ('reduce', 'reduce_small')))


def test_every_forward_instantiation_is_reached_by_a_parity_case():
    seen = cc.all_checked_fwd_variants()
    assert not (set(seen) - ALL_FWD), 'plan query reports an instantiation this list does not know: %s' % (set(seen) - ALL_FWD)
    missing = ALL_FWD - set(seen)
    assert not missing, 'no parity case runs on %s' % sorted(missing)


def test_every_weight_gradient_instantiation_is_reached_by_a_parity_case():
    seen = cc.all_checked_wgrad_variants()
    assert set(seen) == ALL_WGRAD, sorted(ALL_WGRAD - set(seen))


def test_every_bf16_instantiation_is_reached_by_a_parity_case():
    """bf16-compute mode (BASELINE configs 2-4): the four compiled `conv_igemm_bf16*` instantiations and the two bf16 weight-gradient
    kernels are each reached by a case of conv_cases.BF16_CASES — the frame-strided forms (stem conv2 on the even frames:
    `dstride` forward, the two `ydstride` data-gradient launches, the even-frame 3-tap weight gradient) at the BENCHMARK size."""
    fwd, wg = cc.all_checked_bf16_variants()
    assert {v[:2] for v in fwd} == cc.ALL_BF16_FWD, sorted(cc.ALL_BF16_FWD - {v[:2] for v in fwd})
    assert {v[:2] for v in wg} >= cc.ALL_BF16_WGRAD
    name, cin, cout, k, members, relu_in, even = cc.BF16_CASES[0]
    assert name == 'bf16_stem_conv2_B32_even' and even and sum(n * d * h * w for n, d, h, w in members) == 393216
    f, w = cc.bf16_case_plans(cc.BF16_CASES[0])
    assert f == {('strip3_bf16', 128, 1)} and w == ('rows3', 1, 1)          # forward AND both strided-output data-gradient launches
    assert ('strip3_bf16', 64, 1) in fwd                                   # the 64-voxel tile knows the frame stride too
    # what the bf16 entry point refuses runs in fp32 (and says so): Cin not a multiple of 32, thin outputs
    assert cc.bf16_fwd_plan_of(cc._group_array([(2, 4, 8, 8)], 48, 64, (3, 3, 3)), 1, 48, 64) is None
    assert cc.bf16_fwd_plan_of(cc._group_array([(2, 4, 8, 8)], 64, 1, (3, 3, 3)), 1, 64, 1) is None


def test_pooled_convolution_plans():
    """The pooled convolution (functional_pool.py) at the benchmark size: the stem's 393 216 voxels become 49 152 pooled GEMM rows
    (768 tiles: one round of resident workgroups, no split-K), down0's launch splits K to fill the chip; every pooled kernel and
    both reduce forms are reached by a POOL_CASES entry; members the pooled form cannot take are refused (the caller then runs
    the un-pooled path)."""
    import ctypes as C
    from txt2vid_amd._lib import lib, ConvGroup
    name, cin, cout, members, stem = cc.POOL_CASES[0]
    assert sum(n * d * h * w for n, d, h, w in members) == 393216
    assert cc.pool_plan(0, members, cin, cout, stem) == ('pool_fwd', 64, 64, 16, 1, 1, 2, 1)          # (the double-buffered form: 16-channel rounds)
    assert cc.pool_plan(1, members, cout, cin, stem)[0] == 'pool_dgrad'
    w = cc.pool_plan(2, members, cin, cout, stem)
    assert w[0] == 'pool_rows3' and 900 <= w[5] <= 1024                    # one round of weight-gradient workgroups
    assert cc.pool_plan(0, cc.POOL_CASES[1][3], 64, 128, False)[7] > 1     # down0: 112 tiles x 2 -> split-K
    fwd, wg = cc.all_checked_pool_variants()
    assert set(fwd) == {('pool_fwd', 64, 64, 16, 1, 1, 2), ('pool_dgrad', 64, 64, 32, 1, 1, 1)}
    assert set(wg) == {('pool_rows3', 'reduce'), ('pool_rows3', 'reduce_small')}
    out = (C.c_int32 * 8)()
    arr = (ConvGroup * 1)()
    arr[0].N, arr[0].D, arr[0].H, arr[0].W, arr[0].dstride, arr[0].ntaps = 2, 3, 8, 8, 1, 27        # odd frame count
    assert lib().t2v_pool_conv_plan(2, arr, 1, 64, 64, out) < 0
    arr[0].D, arr[0].W = 4, 1                                                                     # one voxel wide
    assert lib().t2v_pool_conv_plan(2, arr, 1, 64, 64, out) < 0
    arr[0].W = 8
    assert lib().t2v_pool_conv_plan(2, arr, 1, 64, 64, out) == 0 and lib().t2v_pool_conv_plan(1, arr, 1, 48, 64, out) < 0   # K % 32


def test_plan_queries_reject_bad_geometry():
    import ctypes as C
    from txt2vid_amd._lib import lib, ConvGroup
    arr = (ConvGroup * 1)()
    out = (C.c_int32 * 8)()
    assert lib().t2v_conv_fwd_plan(arr, 1, 64, 64, 0, out) < 0             # all-zero member
    assert lib().t2v_conv_wgrad_plan(arr, 1, 64, 64, 3, 3, 3, out) < 0
    assert lib().t2v_conv_fwd_plan(arr, 0, 64, 64, 0, out) < 0


def _plans_in_child(env):
    """Forward and weight-gradient plans of three probe shapes in a FRESH process (the tunables are read once per process)."""
    import json
    import os
    import subprocess
    import sys
    code = (
        "import json, sys; sys.path.insert(0, %r); import conv_cases as cc\n"
        "small = [(2, 4, 16, 16)]                     # M = 2048: 8 tiles of 256\n"
        "mid = [(12, 8, 16, 16)]                      # M = 24576: 192 tiles of 128\n"
        "big = cc.d_step_members(32, 0)               # M = 393216\n"
        "print(json.dumps({'small': cc.fwd_plan(small, 64, 64, (3, 3, 3)), 'mid': cc.fwd_plan(mid, 64, 64, (3, 3, 3)),\n"
        "                  'one': cc.fwd_plan([(2, 4, 8, 8)], 256, 64, (1, 1, 1)), 'deep': cc.fwd_plan([(2, 1, 4, 4)], 256, 256, (3, 3, 3)),\n"
        "                  'wbig': cc.wgrad_plan(big, 64, 64, (3, 3, 3)), 'wgp': cc.wgrad_plan(cc.gp_members(32, 0), 64, 64, (3, 3, 3))}))\n"
    ) % os.path.dirname(os.path.abspath(__file__))
    full = dict(os.environ)
    for k in list(full):
        if k.startswith('T2V_'):
            del full[k]
    full.update(env)
    out = subprocess.run([sys.executable, '-c', code], env=full, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return json.loads(out.stdout.decode().strip().splitlines()[-1])


def test_environment_tunables_move_the_plan():
    """Every plan-visible override of `struct Tunables` (conv.hip) changes the plan the way its comment says, and the defaults
    are what the benchmark runs on (T2V_NO_OCC_PAD acts at launch time only: the resident-workgroup padding of the 256-voxel
    tile, covered on the GPU by the benchmark-size parity cases)."""
    base = _plans_in_child({})
    assert base['small'][:4] == ['strip3', 64, 64, 16] and base['small'][6:] == [2, base['small'][7]] and base['small'][7] > 1   # 32 tiles: 64-voxel tiles (double-buffered form), split K
    assert base['mid'][:4] == ['strip3', 64, 64, 16] and base['mid'][7] == 1             # 384 tiles of 64: no split
    assert base['one'][7] == 1                                                          # 1x1x1, K = 8 chunks: never split
    assert base['deep'][7] > 1
    assert base['wbig'][0] == 'rows3' and base['wbig'][1] == 113                         # one round: 113 x 9 = 1017 workgroups
    assert 900 <= base['wgp'][5] <= 1024                                                # 114 x 9 = 1026 would start a second round: quantised down
    # tile thresholds
    assert _plans_in_child({'T2V_TILE256_MIN': '1'})['small'][:4] == ['strip3', 256, 64, 16]
    assert _plans_in_child({'T2V_TILE128_MIN': '1', 'T2V_TILE256_MIN': '100000'})['small'][:4] == ['strip3', 128, 64, 32]
    # split-K knobs
    assert _plans_in_child({'T2V_NOSPLIT_CHUNKS': '0'})['one'][7] > 1
    assert _plans_in_child({'T2V_NOSPLIT_CHUNKS': '1000'})['deep'][7] == 1
    forced = _plans_in_child({'T2V_FORCE_S': '3'})
    assert forced['small'][7] == 3 and forced['mid'][7] == 3
    # strip kernels off: the plain implicit-GEMM instantiation of the same tile
    assert _plans_in_child({'T2V_NO_STRIP': '1'})['mid'][:4] == ['igemm', 64, 64, 32]
    # the single-stage form of the 64-voxel three-tap tile
    assert _plans_in_child({'T2V_STRIP3_DB': '0'})['mid'][:7] == ['strip3', 64, 64, 32, 1, 1, 1]
    assert _plans_in_child({'T2V_STRIP3_DB': '32'})['mid'][:7] == ['strip3', 64, 64, 32, 1, 1, 2]
    # weight-gradient split count
    assert _plans_in_child({'T2V_WGRAD_SCAP': '64'})['wbig'][1] == 64
    assert _plans_in_child({'T2V_WGRAD_TARGET': '512'})['wbig'][5] <= 600
    assert _plans_in_child({'T2V_WGRAD_NOQ': '1'})['wgp'][5] == 1026
    assert _plans_in_child({'T2V_WGRAD_TARGET': '3072'})['wbig'][1] == 256              # capped by T2V_WGRAD_SCAP


def test_frame_strided_members_plan_on_the_host():
    """`t2v_conv_group.dstride / ydstride` (the stem conv2 on the even frames): the plan queries count the GEMM rows over
    the frames that are computed, never split K for such launches, and the weight gradient takes dL/dy on the even frames on the
    3-tap-row kernel only (host arithmetic, no GPU)."""
    import ctypes as C
    from txt2vid_amd._lib import lib
    members = cc.d_step_members(32, 0)                       # the benchmark's 8 stem members, M = 393 216
    k = (3, 3, 3)
    arr = cc._group_array(members, 64, 64, k)
    out = (C.c_int32 * 8)()
    assert lib().t2v_conv_fwd_plan(arr, len(members), 64, 64, 0, out) == 0 and out[0] == 5 and out[1] == 256
    for a in arr:
        a.dstride = 2
    assert lib().t2v_conv_fwd_plan(arr, len(members), 64, 64, 0, out) == 0
    assert (out[0], out[1], out[7]) == (5, 256, 1)           # strip3, 196 608 rows = 768 tiles of 256, no split-K
    w = (C.c_int32 * 6)()
    assert lib().t2v_conv_wgrad_plan(arr, len(members), 64, 64, 3, 3, 3, w) == 0 and w[0] == 2      # rows3 kernel
    full = (C.c_int32 * 6)()
    for a in arr:
        a.dstride = 0
    assert lib().t2v_conv_wgrad_plan(arr, len(members), 64, 64, 3, 3, 3, full) == 0
    assert w[1] * w[2] * 2 <= full[1] * full[2] * 1.1 + 64   # about half the 32-voxel chunks (splits x chunks per split)
    # narrow inputs run on the (tap, ci) column kernel, which does not know the frame stride: rejected
    arr1 = cc._group_array(members, 1, 64, k)
    for a in arr1:
        a.dstride = 2
    assert lib().t2v_conv_wgrad_plan(arr1, len(members), 1, 64, 3, 3, 3, w) < 0
    # strided OUTPUT: one parity of the frames of y, x (dL/dy on the even frames) has ceil(Dy / 2) frames
    half = [(n, (d + 1) // 2, h, wd) for n, d, h, wd in members]
    arr2 = cc._group_array(half, 64, 64, k)
    for a, (n, d, h, wd) in zip(arr2, members):
        a.ydstride, a.yoff, a.Dy = 2, 1, d
    assert lib().t2v_conv_fwd_plan(arr2, len(members), 64, 64, 0, out) == 0 and out[0] == 5 and out[7] == 1
    arr2[0].yoff = 2
    assert lib().t2v_conv_fwd_plan(arr2, len(members), 64, 64, 0, out) < 0
