"""Convolution parity cases shared by the GPU op tests (tests/test_ops_gpu.py) and the CPU launch-plan coverage test
(tests/test_conv_plan.py), plus host-side access to the launch-plan queries of the C ABI (`t2v_conv_fwd_plan`,
`t2v_conv_wgrad_plan`: pure host arithmetic, no GPU needed).

The kernel instantiation a convolution runs on is size-dependent (conv.hip `build_table` / `conv_variant` /
`build_wtable`), so "the op is tested" does not imply "the instantiation the benchmark runs is tested". The lists here
make that explicit: SINGLE_CASES are small single-tensor shapes, GROUPED_CASES include the discriminator's grouped
launches at the BENCHMARK size (BASELINE configs[1]: per-GPU batch 32, 16x64x64x1 clips)."""
import ctypes as C

# (x shape, Cout, kernel) — single-tensor cases, checked forward / data gradient / weight + bias gradient vs torch on the CPU
SINGLE_CASES = [
    ((2, 16, 4, 6, 6), 32, (3, 3, 3)),
    ((3, 1, 4, 8, 8), 64, (3, 3, 3)),        # Cin=1: generic-K path
    ((2, 1, 9, 32, 32), 64, (3, 3, 3)),      # Cin=1, M=18432: its data gradient takes the two-pass thin path
    ((5, 1, 1, 64, 64), 32, (3, 3, 3)),      # same with D=1 (9 live taps)
    ((2, 3, 2, 5, 7), 8, (3, 3, 3)),         # Cin=3, odd sizes
    ((2, 32, 1, 8, 8), 16, (3, 3, 3)),       # D=1: centre plane of taps only
    ((5, 64, 1, 1, 1), 48, (3, 3, 3)),       # 1x1x1 map: centre tap only
    ((2, 16, 3, 4, 4), 24, (1, 1, 1)),       # 1x1x1 kernel
    ((4, 32, 6, 6), 1, (3, 3)),              # 2-D, Cout=1 (render block)
    ((3, 64, 2, 2), 128, (3, 3)),            # 2-D tiny map
    ((7, 80), 33, ()),                       # Linear
    ((2, 64, 4, 32, 32), 64, (3, 3, 3)),     # M=8192: 64x64 strip tiles, split-K
    ((8, 32, 8, 64, 64), 128, (1, 1, 1)),    # M=262144, Cout=128, 1x1x1: non-strip 256x64 tile
    ((6, 1024), 1, ()),                      # the discriminator heads: thin linear kernel, one output
    ((5, 96), 3, ()),                        # thin linear kernel, up to 4 outputs
    # instantiations the small / model shapes above never select (odd channel counts, mid-size and large launches)
    ((2, 32, 4, 16, 16), 6, (3, 3, 3)),      # Cout=6: 128x32 strip tile without vector weight loads
    ((2, 16, 4, 6, 6), 6, (1, 1, 1)),        # 128x32x16 tile, scalar weight loads
    ((2, 32, 4, 6, 6), 6, (1, 1, 1)),        # 128x32x32 tile, scalar weight loads
    ((2, 64, 2, 8, 8), 70, (3, 3, 3)),       # Cout=70: 64x64 strip tiles (K-split waves) without vector loads, ragged channel tile
    ((2, 32, 2, 8, 8), 70, (1, 1, 1)),       # 64x64x32 non-strip tile, scalar weight loads
    ((4, 32, 8, 32, 32), 66, (3, 3, 3)),     # M=32768 x 2 channel tiles: 64x64 strip tiles, one accumulator chain per wave, scalar loads
    ((12, 48, 8, 32, 32), 64, (3, 3, 3)),    # Cin=48 (chunks of 16, no strip), M=98304: 128x64x16 tile with 3-D taps
    ((6, 48, 8, 32, 32), 66, (1, 1, 1)),     # 128x64x16 tile, scalar weight loads
    ((6, 32, 16, 32, 32), 64, (1, 1, 1)),    # M=98304: non-strip 128x64x32 tile
    ((6, 32, 8, 32, 32), 66, (1, 1, 1)),     # 128x64x32 tile, scalar weight loads
    ((6, 64, 8, 32, 32), 128, (3, 3, 3)),    # M=49152, Cout=128: 128x64x32 strip tile
    ((6, 32, 8, 32, 32), 66, (3, 3, 3)),     # 128x64x32 strip tile, scalar weight loads
    ((8, 32, 8, 32, 32), 66, (1, 1, 1)),     # M=65536 x 2 channel tiles: non-strip 256x64 tile, scalar weight loads
    ((8, 16, 8, 32, 32), 66, (3, 3, 3)),     # 256x64 strip tile, scalar weight loads
    ((6, 3, 16, 32, 32), 64, (3, 3, 3)),     # Cin=3 stem at M=98304 (the configs[4] RGB stem): generic-K 128x64 tile
    ((2, 40, 4, 8, 8), 64, (3, 3, 3)),       # Cin=40: generic-K path on 64x64 tiles with Cout=64
    ((16, 128, 4, 4), 256, (3, 3)),          # 2-D, M=256, K=1152: deep split-K
    ((8, 64, 8, 32, 32), 64, (1, 1, 1)),     # 1x1x1 64->64 at M=65536: per-tap weight gradient with the many-splits reduce
    # big 1- / 3-channel 3-D stems run on the narrow-input kernel (conv_stem_kernel): the generic-K tiles they used to reach need other cases
    ((4, 1, 16, 32, 32), 64, (3, 3, 3)),     # the grey-clip stem at M=65536: conv_stem_mfma_kernel (RGB: the Cin=3 case above, conv_stem_kernel<3>)
    ((4, 1, 16, 32, 32), 40, (3, 3, 3)),     # Cout % 32 != 0: the lane-per-voxel conv_stem_kernel<1>
    ((6, 20, 16, 32, 32), 64, (3, 3, 3)),    # Cin=20 at M=98304: generic-K 128x64 tile
    ((2, 5, 2, 5, 7), 8, (3, 3, 3)),         # Cin=5, Cout=8: generic-K 128x32 tile
    # 1x1x1 maps with Cin >= 64: the weight gradient is the TN-product kernel (conv_wgrad_gemm_kernel)
    ((37, 192), 136, ()),                    # ragged 128x128 tiles in both directions, one ragged 32-row chunk pair
    ((1500, 256, 1, 1, 1), 320, (3, 3, 3)),  # k-split (6 tiles -> S = 12), centre tap of a 3x3x3 kernel
    ((2100, 128), 128, ()),                  # one tile, 16 splits: the many-splits reduce
    # narrow inputs, <= 31 (tap, ci) columns: the streaming weight-gradient kernel (conv_wgrad_thin_kernel)
    ((3, 3, 10, 12), 40, (3, 3)),            # Cin=3 x 9 taps = 27 columns (ci > 0), ragged channel tile, extents not powers of two
    ((5, 2, 3, 6, 6), 64, (1, 1, 1)),        # 2 columns, voxels per sample (108) a multiple of 4 but no power of two
    ((3, 1, 3, 5, 7), 33, (3, 3, 3)),        # voxels per sample (105) NOT a multiple of 4: the four-byte gather path of dL/dy
]


def _stage(B, s):
    """[(N, D, H, W)] of the 4 pyramid levels of a per-GPU batch B after s DownSample / stem poolings
    (models/resnet3d.py:12-32: every pooling halves each dim of extent > 1, odd extents are padded)."""
    out = []
    for lvl in range(4):
        b, t, sz = -(-B // (1 << lvl)), 16 >> lvl, 8 << lvl
        for _ in range(s):
            t, sz = (t + 1) // 2 if t > 1 else 1, max(1, sz // 2)
        out.append((b, t, sz, sz))
    return out


def d_step_members(B, s):
    """The 8 members of a discriminator-step forward launch: real||fake per level (batch 2*b) and the gradient-penalty
    interpolates x-hat per level (batch b) — gan/cond_gan.py all_discrim_forward + losses.gradient_penalty."""
    lv = _stage(B, s)
    return [(2 * n, d, h, w) for n, d, h, w in lv] + lv


def gp_members(B, s):
    """The 4 x-hat members alone (the gradient penalty's first-order sweep and its double backward)."""
    return _stage(B, s)


# name, Cin, Cout, kernel, members [(N, D, H, W)], relu_in — grouped launches of the discriminator at the benchmark size
GROUPED_CASES = [
    ('stem_conv2_B32_8members', 64, 64, (3, 3, 3), d_step_members(32, 0), True),     # M=393216: 256x64 strip tile, wgrad S=171
    ('stem_conv2_B32_gp', 64, 64, (3, 3, 3), gp_members(32, 0), True),               # M=131072
    ('down0_conv1_B32_8members', 64, 64, (3, 3, 3), d_step_members(32, 1), True),    # M=49152
    ('down0_conv2_B32_8members', 64, 128, (3, 3, 3), d_step_members(32, 1), True),   # M=49152, 64->128
    ('down1_conv2_B32_8members', 128, 256, (3, 3, 3), d_step_members(32, 2), True),  # M=7680 (ragged T), 128->256
    ('small_ragged_group', 64, 64, (3, 3, 3), [(4, 2, 16, 16), (2, 4, 8, 8), (1, 1, 5, 3)], False),
    ('one_voxel_wide_member', 64, 64, (3, 3, 3), [(2, 4, 8, 8), (3, 4, 6, 1)], False),    # a W = 1 member (no dx taps): the single-stage strip3 form
    ('one_voxel_wide_member_256', 64, 64, (3, 3, 3), [(8, 8, 64, 32), (2, 8, 8, 1)], False),   # ... of the 256-voxel tile (513 tiles)
    ('stem_conv1_B32_8members', 1, 64, (3, 3, 3), d_step_members(32, 0), False),      # M=393216, Cin=1: stem kernels, streaming weight gradient
]


def _group_array(members, cin, cout, k):
    from txt2vid_amd.functional import conv_geom
    from txt2vid_amd._lib import ConvGroup
    geoms = [conv_geom(n, cin, d, h, w, cout, k[0], k[1], k[2]) for n, d, h, w in members]
    mask = 0
    for g in geoms:
        mask |= g.mask
    taps = [t for t in range(geoms[0].T) if (mask >> t) & 1]
    slot_of = {t: j for j, t in enumerate(taps)}
    arr = (ConvGroup * len(members))()
    for a, g in zip(arr, geoms):
        a.x = a.y = a.mask = None
        a.N, a.D, a.H, a.W, a.ntaps = g.cg.N, g.cg.D, g.cg.H, g.cg.W, g.cg.ntaps
        for j, t in enumerate(g.taps):
            a.dz[j], a.dy[j], a.dx[j] = g.cg.dz[j], g.cg.dy[j], g.cg.dx[j]
            a.widx[j] = slot_of[t]
    return arr


def k3(k):
    k = tuple(k)
    return (1,) * (3 - len(k)) + k


def fwd_plan(members, cin, cout, k, flags=0):
    """('igemm'|'strip'|'thin'|'linear'|'thin2'|'strip3', BM, BN, BK, fast, vecb, KS, S) for a forward (or, with cin/cout swapped,
    data-gradient) launch over `members`."""
    from txt2vid_amd._lib import lib
    out = (C.c_int32 * 8)()
    rc = lib().t2v_conv_fwd_plan(_group_array(members, cin, cout, k3(k)), len(members), cin, cout, flags, out)
    assert rc == 0, rc
    v = list(out)
    return ({0: 'igemm', 1: 'strip', 2: 'thin', 3: 'linear', 4: 'thin2', 5: 'strip3', 12: 'stem'}[v[0]],) + tuple(v[1:])


def wgrad_plan(members, cin, cout, k):
    """('taps'|'cols'|'rows3'|'gemm'|'thin', S, chunks per split, slab slots, 'reduce'|'reduce_small', workgroups)."""
    from txt2vid_amd._lib import lib
    out = (C.c_int32 * 6)()
    kk = k3(k)
    rc = lib().t2v_conv_wgrad_plan(_group_array(members, cin, cout, kk), len(members), cin, cout, kk[0], kk[1], kk[2], out)
    assert rc == 0, rc
    v = list(out)
    return (('taps', 'cols', 'rows3', 'gemm', 'thin')[v[0]], v[1], v[2], v[3], ('reduce', 'reduce_small')[v[4]], v[5])


def members_of_single(xs):
    """(N, D, H, W) of a single-tensor case given as [N,C] / [N,C,H,W] / [N,C,D,H,W]."""
    if len(xs) == 2:
        return (xs[0], 1, 1, 1)
    if len(xs) == 4:
        return (xs[0], 1, xs[2], xs[3])
    return (xs[0], xs[2], xs[3], xs[4])


def variant_key(plan):
    """The part of a forward plan that names a kernel instantiation (the split count is a launch parameter)."""
    return plan[:7]


def all_checked_fwd_variants():
    """Every forward / data-gradient instantiation reached by SINGLE_CASES and GROUPED_CASES."""
    seen = {}
    for xs, cout, k in SINGLE_CASES:
        m = [members_of_single(xs)]
        seen.setdefault(variant_key(fwd_plan(m, xs[1], cout, k)), 'single %s->%d fwd' % (xs, cout))
        seen.setdefault(variant_key(fwd_plan(m, cout, xs[1], k)), 'single %s->%d dgrad' % (xs, cout))
    for name, cin, cout, k, members, _ in GROUPED_CASES:
        seen.setdefault(variant_key(fwd_plan(members, cin, cout, k)), name + ' fwd')
        seen.setdefault(variant_key(fwd_plan(members, cout, cin, k)), name + ' dgrad')
    return seen


def all_checked_wgrad_variants():
    seen = {}
    for xs, cout, k in SINGLE_CASES:
        p = wgrad_plan([members_of_single(xs)], xs[1], cout, k)
        seen.setdefault((p[0], p[4]), 'single %s->%d' % (xs, cout))
    for name, cin, cout, k, members, _ in GROUPED_CASES:
        p = wgrad_plan(members, cin, cout, k)
        seen.setdefault((p[0], p[4]), name)
    return seen


# ------------------------------------------------------------------------------------------------
# bf16-compute mode (BASELINE configs 2-4): `t2v_conv_fwd_grouped_bf16` / the T2V_CONV_BF16 weight-gradient kernels.
# name, Cin, Cout, kernel, members [(N, D, H, W)], relu_in, even — checked by tests/test_ops_gpu.py::test_bf16_cases against the
# EXACT convolution / data gradient / weight gradient of the bf16-rounded operands. `even`: the stem conv2's frame-strided
# form (forward on the even output frames `dstride = 2`, data gradient as two `ydstride = 2` launches from even-frame dL/dy,
# weight gradient from even-frame dL/dy).
# ------------------------------------------------------------------------------------------------
BF16_CASES = [
    ('bf16_stem_conv2_B32_even', 64, 64, (3, 3, 3), d_step_members(32, 0), True, True),       # benchmark size, frame-strided: strip3<128>
    ('bf16_down0_conv2_B32', 64, 128, (3, 3, 3), d_step_members(32, 1), True, False),         # M=49152: strip3<128> fwd, strip3<64> dgrad
    ('bf16_down1_conv2_B32', 128, 256, (3, 3, 3), d_step_members(32, 2), True, False),        # M=7680 ragged: strip3<64>, split-K
    ('bf16_small_even', 64, 64, (3, 3, 3), [(2, 4, 16, 16), (1, 5, 8, 8), (3, 2, 4, 4)], True, True),    # strip3<64>, odd frame counts
    ('bf16_skip_1x1_B32', 64, 128, (1, 1, 1), d_step_members(32, 1), False, False),           # 1x1x1 at M=49152: igemm<128> / per-tap bf16 wgrad
    ('bf16_1x1_small', 128, 64, (1, 1, 1), [(2, 4, 8, 8), (5, 1, 1, 1)], False, False),       # igemm<64>
    ('bf16_cout32', 64, 32, (3, 3, 3), [(2, 4, 16, 16)], True, False),                        # Cout <= 32: re-tiled to 128 x 64
]


def _even_dgrad_arrays(members, cin, cout):
    """The two `ydstride = 2` launches of functional._dgrad_even_frames_raw as host-side group tables (no pointers): output frame 2e
    sees the dz = 0 taps, frame 2e + 1 the dz = -1 / +1 taps. Returns [(array, n)]."""
    from txt2vid_amd._lib import ConvGroup
    slot_of = {t: t for t in range(27)}
    out = []
    for yoff, planes in ((0, ((1, 0),)), (1, ((0, 0), (2, 1)))):
        mem = [m for m in members if (m[1] + 1 - yoff) // 2 >= 1]
        if not mem:
            continue
        arr = (ConvGroup * len(mem))()
        for a_, (n, d, h, w) in zip(arr, mem):
            a_.x = a_.y = a_.mask = None
            a_.N, a_.D, a_.H, a_.W = n, (d + 1) // 2, h, w
            a_.dstride, a_.ydstride, a_.yoff, a_.Dy = 0, 2, yoff, d
            j = 0
            for pa, dzc in planes:
                for b in range(3):
                    for c in range(3):
                        a_.dz[j], a_.dy[j], a_.dx[j] = dzc, b - 1, c - 1
                        a_.widx[j] = slot_of[(pa * 3 + b) * 3 + c]
                        j += 1
            a_.ntaps = j
        out.append((arr, len(mem)))
    return out


def bf16_fwd_plan_of(arr, n, cin, cout, flags=0):
    """('igemm_bf16'|'strip3_bf16', BM, frame-strided) of a bf16 launch, or None where the bf16 entry point refuses it (fp32 runs)."""
    from txt2vid_amd._lib import lib
    out = (C.c_int32 * 8)()
    if lib().t2v_conv_fwd_bf16_plan(arr, n, cin, cout, flags, out) != 0:
        return None
    return ({6: 'igemm_bf16', 8: 'strip3_bf16'}[out[0]], out[1], out[4])


def bf16_case_plans(case):
    """Forward / data-gradient bf16 instantiations and the (kernel, bf16, frame-strided) weight-gradient variant of one BF16_CASES entry."""
    name, cin, cout, k, members, relu_in, even = case
    kk = k3(k)
    fwd = set()
    arr = _group_array(members, cin, cout, kk)
    if even:
        for a in arr:
            a.dstride = 2
    fwd.add(bf16_fwd_plan_of(arr, len(members), cin, cout))
    if even:
        for darr, n in _even_dgrad_arrays(members, cout, cin):
            fwd.add(bf16_fwd_plan_of(darr, n, cout, cin, 8))          # T2V_CONV_MASK_OUT
    else:
        fwd.add(bf16_fwd_plan_of(_group_array(members, cout, cin, kk), len(members), cout, cin))
    wp = wgrad_plan(members, cin, cout, k)
    wg = (wp[0], 1 if wp[0] in ('rows3', 'taps') else 0, 1 if even else 0)     # (the Cin < 64 column kernel stays fp32)
    return fwd - {None}, wg


ALL_BF16_FWD = {('igemm_bf16', 128), ('igemm_bf16', 64), ('strip3_bf16', 128), ('strip3_bf16', 64)}       # the compiled instantiations
ALL_BF16_WGRAD = {('rows3', 1), ('taps', 1)}                                                                # conv_wgrad3_kernel<true>, conv_wgrad_bf16_kernel


def all_checked_bf16_variants():
    """(forward / data-gradient variants incl. the frame-strided flag, weight-gradient variants) reached by BF16_CASES."""
    fwd, wg = {}, {}
    for case in BF16_CASES:
        f, w = bf16_case_plans(case)
        for v in f:
            fwd.setdefault(v, case[0])
        wg.setdefault(w, case[0])
    return fwd, wg


# ------------------------------------------------------------------------------------------------
# pooled convolution (functional_pool.py: box-sum + stride-2 GEMMs for the conv2 -> pooling pair of every discriminator block):
# name, Cin, Cout, members, stem — the first two at the benchmark size (tests/test_ops_gpu.py::test_pool_conv_group_at_benchmark_size)
# ------------------------------------------------------------------------------------------------
POOL_CASES = [
    ('pool_stem_conv2_B32', 64, 64, d_step_members(32, 0), True),       # 393 216 voxels -> 49 152 pooled rows
    ('pool_down0_conv2_B32', 64, 128, d_step_members(32, 1), False),    # 49 152 -> 7 168 rows: split-K forward
    ('pool_down1_conv2_B32', 128, 256, d_step_members(32, 2), False),
    ('pool_small', 64, 64, [(2, 4, 16, 16), (1, 2, 8, 8), (3, 1, 4, 4)], True),
]


def pool_plan(what, members, cin, cout, stem):
    """('pool_fwd'|'pool_dgrad', 64, 64, 32, 1, 1, 1, S) or ('pool_rows3', S, chunks per split, slots, reduce kind, workgroups): the
    launch plan of the pooled forward (what 0), data gradient (1: cin = channels of dL/dy) or weight gradient (2)."""
    from txt2vid_amd._lib import lib, ConvGroup
    from txt2vid_amd.functional_pool import pool_tmode
    arr = (ConvGroup * len(members))()
    for a, (n, d, h, w) in zip(arr, members):
        tm = pool_tmode((n, 0, d, h, w), stem)
        assert tm is not None
        a.x = a.y = a.mask = None
        a.N, a.D, a.H, a.W, a.dstride = n, d, h, w, tm
        j = 0
        for dz in ((-1, 0, 1) if tm else (0,)):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    a.dz[j], a.dy[j], a.dx[j], a.widx[j] = dz, dy, dx, j
                    j += 1
        a.ntaps = j
        if what == 1:
            for f in range(27):
                a.widx[f] = f
    out = (C.c_int32 * 8)()
    rc = lib().t2v_pool_conv_plan(what, arr, len(members), cin, cout, out)
    assert rc == 0, rc
    v = list(out)
    if what == 2:
        return ('pool_rows3', v[1], v[2], v[3], ('reduce', 'reduce_small')[v[4]], v[5])
    return (('pool_fwd', 'pool_dgrad')[what],) + tuple(v[1:])


def all_checked_pool_variants():
    """(forward / data-gradient keys, weight-gradient keys) the POOL_CASES reach — same key shapes as the un-pooled kernels'."""
    fwd, wg = {}, {}
    for name, cin, cout, members, stem in POOL_CASES:
        fwd.setdefault(variant_key(pool_plan(0, members, cin, cout, stem)), name + ' fwd')
        fwd.setdefault(variant_key(pool_plan(1, members, cout, cin, stem)), name + ' dgrad')
        p = pool_plan(2, members, cin, cout, stem)
        wg.setdefault((p[0], p[4]), name)
    return fwd, wg
