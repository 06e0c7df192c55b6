/* t2v_hip.h — C ABI of libt2v_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the TGANv2
 * training hot path of miguelmartin75/txt2vid.
 *
 * The reference has NO native layer (100 % Python on PyTorch); every entry point below
 * replaces a *PyTorch operator call site* of the reference, cited as  <file>:<line>  relative to the
 * reference root.  Conventions:
 *   - plain pointers to DEVICE memory + sizes; no torch types; the caller owns every buffer
 *     (kernels never allocate; workspaces are passed in);
 *   - all tensors fp32, contiguous, NCDHW (2-D tensors are the D=1 case, Linear is D=H=W=1);
 *   - work is enqueued on `stream` (a hipStream_t passed as void*), nothing synchronises;
 *   - return 0 on success, a negative T2V_E* code on a bad argument, the (negated) hipError_t
 *     if the launch itself failed.  Nothing throws across the boundary.  No global state.
 */
#ifndef T2V_HIP_H
#define T2V_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2V_OK 0
#define T2V_EINVAL (-1)
#define T2V_ELAUNCH (-2)

#define T2V_MAX_TAPS 27

/* Geometry of a stride-1, "same"-padded convolution (k in {1,3} per dim) — the only kind on the hot
 * path: nn.Conv3d / nn.Conv2d call sites txt2vid/models/resnet3d.py:13-18, layers.py:174-183,231-238,
 * 251, conv_lstm.py:19-26; nn.Linear (resnet3d.py:33-35, tganv2_cond/gen.py:39) is D=H=W=1,k=1.
 * Only the `ntaps` kernel taps that can touch a non-padding voxel are listed (a dim of extent 1
 * keeps its centre tap only): tap j has spatial offset (dz,dy,dx)[j] and sits at position j of the
 * packed weight (see t2v_pack_weight). */
typedef struct {
    int32_t N, Cin, D, H, W, Cout;
    int32_t ntaps;
    int8_t dz[T2V_MAX_TAPS], dy[T2V_MAX_TAPS], dx[T2V_MAX_TAPS];
    int8_t pad_[3];
} t2v_conv_geom;

/* One member of a GROUPED convolution: tensors that share the weights but not the geometry (the pyramid
 * levels of the multi-scale discriminator, models/tganv2/discrim.py:23-31 forward loop) are convolved by ONE launch.
 * Tap j of this member has offset (dz,dy,dx)[j] and reads slot widx[j] of the packed weight. */
#define T2V_MAX_GROUPS 8
typedef struct {
    const float* x;      /* input  [N,Cin,D,H,W]                      (wgrad: the layer input)      */
    float* y;            /* output [N,Cout,D,H,W]                     (wgrad: dL/dy, read only)     */
    const float* mask;   /* T2V_CONV_MASK_OUT: same shape as y; y = mask > 0 ? result : 0  (else NULL) */
    int32_t N, D, H, W;
    int32_t ntaps;
    int32_t dstride;     /* 0 / 1: every frame; 2 (forward launches on the strip3 kernels only): y holds the EVEN frames only,
                            [N,Cout,ceil(D/2),H,W] — the stem's conv2 -> AvgPool3d((1,2,2), stride 2) pair (resnet3d.py:12-19) drops the
                            odd frames of conv2, so they are not computed. The weight-gradient 3-tap-row kernel takes 2 as well (dL/dy
                            on the even frames only). Other entry points reject it. */
    int32_t ydstride;    /* 0 / 1: y is dense. 2 (strip3 forward launches, i.e. the data gradient of the pair above): GEMM row
                            (n, e, h, w) is written to frame 2e + yoff of y [N,Cout,Dy,H,W]; x (dL/dy on the even frames) has D
                            frames, rows run over e < (Dy + 1 - yoff) / 2. Two launches (yoff = 0 with the dz = 0 taps, yoff = 1 with
                            the dz = -1 / +1 taps re-based to x frames e, e + 1) write every frame of y exactly once. */
    int32_t yoff, Dy;
    int8_t dz[T2V_MAX_TAPS], dy[T2V_MAX_TAPS], dx[T2V_MAX_TAPS], widx[T2V_MAX_TAPS];
} t2v_conv_group;

#define T2V_CONV_BIAS 1      /* add bias[Cout] in the epilogue                                   */
#define T2V_CONV_RELU_IN 2   /* apply max(.,0) to the input while gathering (ReLU->conv fusion)  */
#define T2V_CONV_ACCUM 4     /* y += result instead of y = result                                */
#define T2V_CONV_ACCUM_BIAS 16 /* t2v_conv_wgrad_grouped_bias: dbias += result (T2V_CONV_ACCUM covers dw only)      */
#define T2V_CONV_BF16 32      /* t2v_conv_wgrad_grouped[_bias]: bf16-compute mode for the MFMA kernels (fp32 tensors)        */
#define T2V_CONV_MASK_OUT 8  /* zero the result where groups[i].mask <= 0: the ReLU adjoint fused into the data
                                gradient of a ReLU->conv pair (layers.py:230-233). Not combined with ACCUM. */

/* w[Cout][Cin][T] (PyTorch layout, T = kD*kH*kW) -> wp[ntaps][Cin][Cout] for the forward GEMM
 * (mode 0: wp[j][ci][co] = w[co][ci][taps[j]]) or wp[ntaps][Cout][Cin] for the data-gradient GEMM
 * (mode 1: wp[j][co][ci] = w[co][ci][T-1-taps[j]], i.e. the spatially mirrored kernel).
 * mode | 8: the source is stored tap-major, w[T][Cout][Cin] (the ConvLSTM's master weights, whose live taps are then contiguous).
 * `taps` is a HOST array of the ntaps original tap indices. */
int t2v_pack_weight(const float* w, float* wp, int Cout, int Cin, int T, const int32_t* taps, int ntaps,
                    int mode, void* stream);

/* Same, into a wider packed matrix: several weights that are applied to the same input (mode 0: side by side
 * along the output channels, per-tap matrix [dst_rows=Cin][dst_cols=sum Cout], col_off = first output
 * channel) or summed into the same output (mode 1: stacked, [dst_rows=sum Cout][dst_cols=Cin], row_off)
 * become ONE GEMM — the four gates of the ConvLSTM cell (conv_lstm.py:19-26,33-37). */
int t2v_pack_weight_into(const float* w, float* wp, int Cout, int Cin, int T, const int32_t* taps, int ntaps,
                         int mode, int dst_rows, int dst_cols, int row_off, int col_off, void* stream);

/* Multi-tensor form: ONE launch refreshes every packed variant listed in a device-resident job table (what an
 * optimiser step invalidates: ~170 small packs per step otherwise). Record layout (t2v_pack_job_bytes() bytes): */
typedef struct {
    const float* src;     /* w[Cout][Cin][T] */
    float* dst;           /* packed matrix base */
    int32_t Cout, Cin, T, ntaps, mode, dst_rows, dst_cols, row_off, col_off;
    int32_t block_begin;  /* first workgroup of this job */
    int32_t bx, by;       /* ceil(Cin/32), ceil(Cout/32); the job owns bx*by*ntaps workgroups */
    int8_t taps[T2V_MAX_TAPS];
    int8_t pad_[5];
} t2v_pack_job;
int t2v_pack_job_bytes(void);
int t2v_pack_multi(const void* table /* device t2v_pack_job[njobs] */, int njobs, int total_blocks, void* stream);

/* y[N,Cout,D,H,W] = conv(x[N,Cin,D,H,W], wp) (+bias).  Implicit GEMM on v_mfma_f32_32x32x2_f32:
 * M = N*D*H*W voxels, N = Cout, K = ntaps*Cin.  The same entry point computes the data gradient
 * when given the mode-1 packed weight (x := dL/dy, Cin := Cout_fwd, Cout := Cin_fwd).
 * Replaces F.conv3d / F.conv2d / F.linear forward and their input-gradient
 * (autograd of the call sites above; double backward for losses.py:178). */
int t2v_conv_fwd(const float* x, const float* wp, const float* bias, float* y, float* ws, const t2v_conv_geom* g,
                 int flags, void* stream);
/* floats of workspace `ws` the call above needs for this geometry (0: none). Layers with few output
 * voxels and a long reduction are run split-K: partial tiles go to ws and are summed in a fixed order. */
int64_t t2v_conv_fwd_ws_floats(const t2v_conv_geom* g);

/* dw[Cout][Cin][T] = sum_m gy[m][co] * x[m + off(tap)][ci]   (weight gradient; PyTorch layout out).
 * `slab` is a workspace of t2v_conv_wgrad_slab_floats(g, T) floats (split-K partial sums, reduced
 * deterministically by a second kernel).  `taps`: HOST array, original tap index of geom tap j.
 * flags: T2V_CONV_RELU_IN applies max(.,0) to x; T2V_CONV_ACCUM adds into dw. */
int64_t t2v_conv_wgrad_slab_floats(const t2v_conv_geom* g, int T);
int t2v_conv_wgrad(const float* x, const float* gy, float* dw, float* slab, const t2v_conv_geom* g,
                   const int32_t* taps, int T, int flags, void* stream);

/* Grouped forms: one launch for all members (same Cin/Cout/kernel). wp must hold every slot the members
 * reference. The weight gradient is the SUM over all members (what autograd would otherwise accumulate
 * level by level); its kernel extent is given explicitly and every live tap is computed. */
int64_t t2v_conv_fwd_grouped_ws_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout);
int t2v_conv_fwd_grouped(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, const float* wp,
                         const float* bias, float* ws, int flags, void* stream);
int64_t t2v_conv_wgrad_grouped_slab_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD,
                                           int kH, int kW);
int t2v_conv_wgrad_grouped(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW,
                           float* dw, float* slab, int flags, void* stream);
/* Deferred, batched reduction of the weight-gradient partial sums (one reduce launch per backward pass instead of one per
 * layer). t2v_conv_wgrad_grouped_partial = t2v_conv_wgrad_grouped[_bias] without its second (reduce) kernel: the k-split
 * partial sums stay in `slab` (same size query as the full call; `want_bias` != 0 also keeps the dL/dy side-sums behind
 * it) and *out_src describes them. t2v_wgrad_reduce_multi then sums, for every
 * destination of a DEVICE table, its sources in order: dw (+)= sum_src sum_splits slab, dbias likewise.
 * Replaces the per-parameter accumulation autograd does for conv weights (AccumulateGrad of the call sites above). */
#define T2V_WGRAD_MAX_SRC 6
typedef struct {
    const float* slab;        /* [S][ntaps][Cout*Cin] partial sums                                   */
    const float* bias_slab;   /* [S][Cout] partial dL/dy sums, or NULL                                */
    int32_t ntaps, S;         /* slab slots, k-splits                                                */
    int8_t map[T2V_MAX_TAPS]; /* original tap t -> slab slot, -1: tap never touched (contributes 0)  */
    int8_t pad_[5];
    int64_t tap_stride;       /* floats between the planes of two slots (Cout*Cin of the LAUNCH; a destination that is a
                                 row block of a fused weight — one gate of the ConvLSTM's 4-gate GEMM — points `slab` at its
                                 first row and keeps the launch's strides)                            */
    int64_t split_stride;     /* floats between two k-splits (ntaps * tap_stride)                    */
} t2v_wgrad_src;
typedef struct {
    float* dw;                /* [Cout][Cin][T] (PyTorch layout)                                      */
    float* dbias;             /* [Cout] or NULL                                                       */
    int64_t CoCi;             /* Cout * Cin                                                           */
    int32_t T, Cout, nsrc;
    int32_t accum, accum_bias; /* add to what dw / dbias hold instead of overwriting                  */
    int32_t kind;             /* 0: one workgroup per 64 (co,ci) pairs; 1: per (64 pairs, tap), for small weights with many splits;
                                 2 / 3: the same with 256 pairs per workgroup and 16-byte loads (CoCi % 4 == 0, slabs 16-byte aligned) */
    int32_t block_begin, nblocks;
    int32_t tap_major;        /* dw is stored [T][Cout][Cin] (tap-major master weights); taps no source touches stay untouched */
    int32_t pad_;
    t2v_wgrad_src src[T2V_WGRAD_MAX_SRC];
} t2v_wgrad_dest;
int t2v_conv_wgrad_grouped_partial(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW,
                                   float* slab, int want_bias, int flags, t2v_wgrad_src* out_src, void* stream);
int t2v_wgrad_dest_bytes(void);
int t2v_wgrad_reduce_multi(const void* table /* device t2v_wgrad_dest[ndest] */, int ndest, int total_blocks, void* stream);

/* Launch-plan queries (host-side arithmetic only, nothing is launched): which kernel instantiation the two grouped entry
 * points above select for these members. The tile choice is size-dependent (256x64 strips only for launches of >= 512 such
 * tiles, K-split wave layouts for small grids, the many-splits reduce for small weights), so the parity tests use these to
 * assert that every instantiation — the benchmark shapes' in particular — is reached by a checked case.
 * fwd  out[8]: kind (0 implicit GEMM, 1 strip implicit GEMM, 2 thin conv, 3 thin linear, 4 thin two-pass, 5 strip GEMM with
 *              all three dx taps per barrier round), BM, BN, K chunk,
 *              FAST, VECB, KS (2: the double-buffered form of the 64-voxel three-tap tile — one barrier per round, the
 *              next round's loads and LDS writes between the MFMAs; K chunk then = its channels per round), split-K S.
 *              (groups[].x / .y may be NULL)
 * wgrad out[6]: kernel (0 per-tap tiles, 1 (tap,ci) column tiles, 2 three-tap rows, 3 TN product on 1x1x1 maps — fp32 in bf16-compute mode too, 4 the streaming kernel for <= 31 (tap, ci) columns), S, chunks per split, slab slots,
 *              reduce kernel (0 / 1 = many-splits small-weight form), workgroups of the main launch. */
int t2v_conv_fwd_plan(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int flags, int32_t* out);
int t2v_conv_wgrad_plan(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW, int32_t* out);
/* Same, plus the bias gradient dbias[Cout] = sum over all members and voxels of dL/dy (what t2v_channel_sum_grouped
 * computes, layers.py / resnet3d.py conv biases): every weight-gradient kernel adds up the dL/dy tiles it stages anyway
 * (k-split partial sums behind the slab, summed by the reduce pass). The slab must hold
 * t2v_conv_wgrad_grouped_bias_slab_floats() floats. */
int64_t t2v_conv_wgrad_grouped_bias_slab_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD,
                                                int kH, int kW);
int t2v_conv_wgrad_grouped_bias(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW,
                                float* dw, float* dbias, float* slab, int flags, void* stream);

/* ---- bf16 compute / fp32 storage (BASELINE configs 2-4: "bf16 compute / fp32 master") -------------------------------
 * Forward / data gradient with the tiles rounded to bf16 on their way into LDS and multiplied on v_mfma_f32_32x32x16_bf16
 * (fp32 accumulation, fp32 tensors in HBM). wpb: bf16 [ntaps][rows][K] from t2v_pack_weight_bf16 (mode 0: rows = Cout,
 * K = Cin; mode 1: rows = Cin, K = Cout, mirrored taps). t2v_conv_fwd_grouped_bf16_ok() says whether the bf16 kernel takes a
 * shape (Cin % 32 == 0, Cout > 4); otherwise use the fp32 entry point. Workspace: t2v_conv_fwd_grouped_ws_floats(). */
int t2v_pack_weight_bf16(const float* w, void* wpb, int Cout, int Cin, int T, const int32_t* taps, int ntaps, int mode, void* stream);
int t2v_conv_fwd_grouped_bf16_ok(const t2v_conv_group* groups, int ngroups, int Cin, int Cout);
/* Launch plan of t2v_conv_fwd_grouped_bf16 (host arithmetic, no launch; t2v_conv_fwd_plan layout): out[0] kind (6
 * conv_igemm_bf16_kernel, 8 conv_igemm_bf16_strip3_kernel), out[1] BM (128 | 64), out[2] 64, out[3] 32, out[4] 1 for
 * frame-strided members, out[7] split-K S. T2V_EINVAL where the bf16 entry point refuses the launch. Used by the parity tests to assert
 * that every compiled bf16 instantiation — and every one the benchmark's bf16 iteration launches — is covered by an op case. */
int t2v_conv_fwd_bf16_plan(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int flags, int32_t* out);
int t2v_conv_fwd_grouped_bf16(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, const void* wpb, const float* bias,
                              float* ws, int flags, void* stream);

/* out[c] = sum_{n,s} x[n,c,s]  (bias gradient; also BatchNorm reductions).  accum: out += */
int64_t t2v_channel_sum_ws_floats(int N, int C, int64_t S);   /* floats of `ws` needed (0: none) */
int t2v_channel_sum(const float* x, float* out, float* ws, int N, int C, int64_t S, int accum, void* stream);
/* grouped: the sum runs over every member (groups[i].x = [N_i, C, D_i, H_i, W_i]) — bias gradient of a grouped conv */
int64_t t2v_channel_sum_grouped_ws_floats(const t2v_conv_group* groups, int ngroups, int C);
int t2v_channel_sum_grouped(const t2v_conv_group* groups, int ngroups, int C, float* out, float* ws, int accum, void* stream);

/* ---- pointwise / pooling (txt2vid/models/layers.py, resnet3d.py) ------------------------------ */
int t2v_relu(const float* x, float* y, int64_t n, void* stream);                 /* layers.py:172,230 */
int t2v_relu_mask(const float* g, const float* x, float* gx, int64_t n, void* stream); /* g*(x>0)   */
int t2v_add(const float* a, const float* b, float* y, int64_t n, void* stream);  /* layers.py:96     */
int t2v_axpby(float alpha, const float* a, float beta, const float* b, float* y, int64_t n, void* stream);
int t2v_scale_dev(const float* s, float mul, const float* a, float* y, int64_t n, void* stream); /* y = (*s*mul)*a */
int t2v_dot(const float* a, const float* b, float* out, float* ws /*256 floats*/, int64_t n, int accum, void* stream);
int t2v_fill(float* y, float v, int64_t n, void* stream);
int t2v_tanh(const float* x, float* y, int64_t n, void* stream);                 /* layers.py:258    */
int t2v_tanh_bwd(const float* g, const float* y, float* gx, int64_t n, void* stream);

/* avg_pool3d with per-dim kernel/stride/pad, count_include_pad=True (layers.py:217, resnet3d.py:16).
 * bwd scatters g/(kd*kh*kw) back (windows never overlap on this path: stride >= kernel). */
/* x2 != NULL: y = pool(x + x2) — the residual add fused into the block's DownSample (layers.py:93-96,217). */
int t2v_avgpool3d(const float* x, const float* x2, float* y, int NC, int D, int H, int W, int Do, int Ho, int Wo,
                  const int32_t k[3], const int32_t s[3], const int32_t p[3], void* stream);
int t2v_avgpool3d_bwd(const float* gy, float* gx, int NC, int D, int H, int W, int Do, int Ho, int Wo,
                      const int32_t k[3], const int32_t s[3], const int32_t p[3], void* stream);
/* Multi-tensor forms: `jobs` is a HOST array; 8 jobs per launch, descriptors passed in the kernel arguments (the pyramid
   levels of one DownBlock are pooled together). fwd: y = pool(x [+ x2]) [+ add]. bwd: x = dL/dy [NC,Do,Ho,Wo], y = dL/dx
   [NC,D,H,W]; only un-padded windows with kernel <= stride (what DownSample produces, layers.py:197-217). */
typedef struct t2v_pool_job {
    const float* x; const float* x2; float* y;
    const float* add;    /* optional. fwd: y = pool(x [+ x2]) + add  (add has y's shape: the stem's skip, resnet3d.py:16-19);
                            bwd: dL/dx = pool^T(dL/dy) + add  (add has dL/dx's shape: the gradient the pooled tensor's OTHER consumer sent) */
    int32_t NC, D, H, W, Do, Ho, Wo;
    int32_t k[3], s[3], p[3];
} t2v_pool_job;
int t2v_avgpool3d_multi(const t2v_pool_job* jobs, int njobs, void* stream);
int t2v_avgpool3d_bwd_multi(const t2v_pool_job* jobs, int njobs, void* stream);

/* max_pool [1,2,2] / [2,2] (layers.py:26-27,57-58): y + flat argmax index inside the (H,W) plane.
 * `_scatter`: gx = 0 except gx[idx] = g;  `_gather`: y = x[idx]. */
int t2v_maxpool2x2(const float* x, float* y, int32_t* idx, int64_t planes, int H, int W, void* stream);
int t2v_maxpool2x2_scatter(const float* g, const int32_t* idx, float* gx, int64_t planes, int H, int W, void* stream);
int t2v_maxpool2x2_gather(const float* x, const int32_t* idx, float* y, int64_t planes, int H, int W, void* stream);

/* sum over the S = T*H*W voxels of each (n,c) row (resnet3d.py:48) and its broadcast adjoint. */
int t2v_rowsum(const float* x, float* y, int64_t rows, int64_t S, void* stream);
int t2v_rowbcast(const float* g, float* gx, int64_t rows, int64_t S, void* stream);

/* nearest x2 up-sampling of (H,W) planes (nn.Upsample, layers.py:168,180) and its adjoint. */
int t2v_upsample2x(const float* x, float* y, int64_t planes, int H, int W, void* stream);
int t2v_upsample2x_bwd(const float* gy, float* gx, int64_t planes, int H, int W, void* stream);
/* y[planes,2H,2W] = upsample2x(x[planes,H,W]) + h[planes,2H,2W]: UpBlock's closing residual add (layers.py:152-195) with the
 * identity path's nearest up-sampling folded in (its 1x1 convolution runs BEFORE the up-sampling: same values, a quarter of the work). */
int t2v_upsample2x_add(const float* x, const float* h, float* y, int64_t planes, int H, int W, void* stream);

/* BatchNorm2d, training mode (layers.py:171,175,249): per-channel batch statistics over (N,H,W),
 * running stats updated with `momentum` (unbiased variance), y = relu?((x-mean)*invstd*gamma+beta).
 * stats[0:C] = mean, stats[C:2C] = invstd (saved for backward). */
int64_t t2v_bn_ws_floats(int N, int C, int64_t S);   /* floats of `ws` for bn_stats / bn_bwd */
int t2v_bn_stats(const float* x, float* stats, float* running_mean, float* running_var, float* ws, int N, int C,
                 int64_t S, float momentum, float eps, void* stream);
int t2v_bn_apply(const float* x, const float* stats, const float* gamma, const float* beta, float* y,
                 int N, int C, int64_t S, int relu, void* stream);
/* backward of bn_apply(+relu): needs y (for the relu mask) ; writes gx, ggamma[C], gbeta[C] */
int t2v_bn_bwd(const float* gy, const float* x, const float* y, const float* stats, const float* gamma,
               float* gx, float* ggamma, float* gbeta, float* ws /*t2v_bn_ws_floats*/, int N, int C, int64_t S,
               int relu, void* stream);
/* eval-mode affine: y = relu?((x-rm)*rsqrt(rv+eps)*gamma+beta) */
/* Two-launch forms (partial statistics, then a pass whose workgroups merge the channel's partials themselves: no `final`
   launch): training forward = t2v_bn_stats + t2v_bn_apply, backward = t2v_bn_bwd. ws: t2v_bn_ws_floats floats. */
int t2v_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* running_mean,
                     float* running_var, float* ws, int N, int C, int64_t S, float momentum, float eps, int relu,
                     int64_t* num_batches_tracked /* nn.BatchNorm's step counter, += 1 on the device; may be NULL */, void* stream);
int t2v_bn_train_bwd(const float* gy, const float* x, const float* y, const float* stats, const float* gamma, float* gx,
                     float* ggamma, float* gbeta, float* ws, int N, int C, int64_t S, int relu, void* stream);
/* BatchNorm2d -> [ReLU] -> Upsample(2), the head of UpBlock's main path (layers.py:152-195), in the same two launches:
   x / gx are [N,C,H,W], y / gy the up-sampled [N,C,2H,2W] tensors (the adjoint sums the 2x2 copies while it reads gy). */
int t2v_bn_train_fwd_up(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* running_mean,
                        float* running_var, float* ws, int N, int C, int H, int W, float momentum, float eps, int relu,
                        int64_t* num_batches_tracked, void* stream);
int t2v_bn_train_bwd_up(const float* gy, const float* x, const float* y, const float* stats, const float* gamma, float* gx,
                        float* ggamma, float* gbeta, float* ws, int N, int C, int H, int W, int relu, void* stream);
/* the two adjoints above (`up` = 0 / 1; S = H * W) with gx = BatchNorm's input gradient + gx_add (may be NULL): the gradient
   the same input received from its other consumer — the UpBlock's skip path, the next level behind a RenderBlock's map
   (layers.py:152-195, 245-259) — is summed by this pass. */
int t2v_bn_train_bwd_add(const float* gy, const float* x, const float* y, const float* stats, const float* gamma, const float* gx_add,
                         float* gx, float* ggamma, float* gbeta, float* ws, int N, int C, int H, int W, int up, int relu, void* stream);
int t2v_bn_eval(const float* x, const float* rm, const float* rv, const float* gamma, const float* beta,
                float* y, int N, int C, int64_t S, float eps, int relu, void* stream);

/* ConvLSTM gate math (conv_lstm.py:32-38, peepholes are constant zeros):
 * pre[B,4,C,S] (order i,f,c,o: the NCHW output of one convolution over the 4 side-by-side packed gate
 * weights) ; c_prev[B,C,S] -> h, c_new; act saved [B,4,C,S] (i,f,g,o after the non-linearities) for the
 * backward; gpre likewise [B,4,C,S]. */
int t2v_lstm_gates(const float* pre, const float* c_prev, float* h, float* c_new, float* act,
                   int B, int64_t CS, void* stream);
int t2v_lstm_gates_bwd(const float* gh, const float* gc_in, const float* act, const float* c_prev,
                       const float* c_new, float* gpre, float* gc_prev, int B, int64_t CS, void* stream);
/* The same recurrence on 1x1 feature maps (the generator's ConvLSTM state is [B,1024,1,1]): a step is a 32-row GEMM.
   t2v_skinny_gemm_slab: slab[s][m][n] = sum over the s-th K slice of x[m][k] * w[k][n] (x [M][K], w [K][N] row-major =
   the packed 1-tap weight); S = t2v_skinny_gemm_splits(M,K,N) slices (K % 128 == 0, else < 0), slab holds S*M*N floats.
   t2v_lstm_gates_slab / _bwd_slab: the gate kernels with  pre = bias + sum_s slab[s]  resp.  dL/dh = gh + sum_s slab[s]
   (S = 0: no incoming term), so the split-K pass and the add are fused into them. */
int t2v_skinny_gemm_splits(int M, int K, int N);
int t2v_skinny_gemm_slab(const float* x, const float* w, float* slab, int M, int K, int N, void* stream);
int t2v_lstm_gates_slab(const float* slab, int S, const float* bias, const float* c_prev, float* h, float* c_new,
                        float* act, int B, int C, void* stream);
int t2v_lstm_gates_bwd_slab(const float* gh, const float* slab, int S, const float* gc_in, const float* act,
                            const float* c_prev, const float* c_new, float* gpre, float* gc_prev, int B, int C, void* stream);
/* One launch per recurrence step on 1x1 maps (conv_lstm.py:75-97 at [B,1024,1,1]): pre = x [B][K] . w [K][4C] + bias, then the gate
   math of t2v_lstm_gates, fused. `wr` is the unit-major copy of the packed weight, wr[u][k/4][g*4 + j][k%4] = w[k][g*C + 4u + j]
   (t2v_lstm_pack_major, K*4C floats), so that a workgroup's four hidden units are one contiguous stream. act is [B][4C] (i,f,c,o),
   h / c_prev / c_new are [B][C]. t2v_lstm_step_fused_ok: B <= 32, K % 128 == 0, C % 4 == 0 (else the slab pair above). */
/* out[g*n + i] = src_g[i] for four same-length vectors (the ConvLSTM's gate biases, conv_lstm.py:19-26, packed side by side). */
int t2v_concat4(const float* a, const float* b, const float* c, const float* d, float* out, int n, void* stream);
int t2v_lstm_step_fused_ok(int B, int K, int C);
int t2v_lstm_pack_major(const float* w, float* wr, int K, int C, void* stream);
int t2v_lstm_step_fused(const float* x, const float* wr, const float* bias, const float* c_prev, float* h, float* c_new,
                        float* act, int B, int K, int C, void* stream);
/* Its adjoint, also one launch per step: dL/dh_t = gh + gnext [B][4C] . w1 [4C][C] (gnext = dL/dpre of step t+1, NULL at the last
   step), then the gate adjoints of t2v_lstm_gates_bwd. w1r is the column-block-major copy of the data-gradient form of the packed
   Wh: w1r[ub][n/4][j][n%4] = w1[n][16 ub + j] (t2v_lstm_pack_cols with R = 4C rows, Cn = C columns). Additionally C % 64 == 0. */
int t2v_lstm_pack_cols(const float* w, float* wr, int R, int Cn, void* stream);
int t2v_lstm_step_bwd_fused(const float* gh, const float* gnext, const float* w1r, const float* gc_in, const float* act,
                            const float* c_prev, const float* c_new, float* gpre, float* gc_prev, int B, int C, void* stream);

/* ---- non-local block (layers.py:23-36, 52-68) -------------------------------------------------- */
/* C[b] = alpha * op(A[b]) x op(B[b]); row-major A[b]: (ta? K x M : M x K), B[b]: (tb? N x K : K x N). */
int t2v_bmm(const float* A, const float* B, float* C, int batch, int M, int N, int K, int ta, int tb,
            int accum, void* stream);
int t2v_softmax(const float* x, float* y, int64_t rows, int n, void* stream);            /* layers.py:33 */
int t2v_softmax_bwd(const float* y, const float* gy, float* gx, int64_t rows, int n, void* stream);
/* adjoint of softmax_bwd w.r.t. y: gyv = gg*(gy - s) - gy*sum(gg*y), s = sum(gy*y) */
int t2v_softmax_bwd_bwd_y(const float* y, const float* gy, const float* gg, float* out, int64_t rows, int n, void* stream);

/* ---- sentence encoder (txt2vid/models/txt/basic.py:49-70: packed-sequence nn.LSTM, forward only) -----------------
 * One time step of one (layer, direction): pre = xproj_t[b] (row stride xstride; = x_t W_ih^T + b_ih + b_hh from one GEMM
 * over all steps) + h_prev[b] W_hh^T; gates i,f,g,o; samples with t >= lengths[b] keep (h, c) and emit 0 into out_t[b]
 * (row stride ostride). h/c are ping-pong buffers [B,H]. w_hh_t is nn.LSTM's weight_hh TRANSPOSED: [H][4H] (one workgroup per
 * sample reads it with consecutive threads on consecutive addresses; t2v_permute01 makes it once per sequence). 4H <= 1024. */
int t2v_lstm_seq_step(const float* xproj_t, int64_t xstride, const float* w_hh_t, const float* h_prev, const float* c_prev,
                      float* h_next, float* c_next, float* out_t, int64_t ostride, const int32_t* lengths, int t, int B, int H,
                      void* stream);
/* the forward AND the reverse direction's step of one loop iteration in one launch (they are independent). ptrs: 2 x 7 pointers
   {xproj_t, w_hh_t, h_prev, c_prev, h_next, c_next, out_t} (forward, then reverse), ts: their two time steps. */
int t2v_lstm_seq_step2(const void* const* ptrs, const int32_t* ts, int64_t xstride, int64_t ostride, const int32_t* lengths,
                       int B, int H, void* stream);

/* ---- text-encoder pre-training (txt2vid/train/txt.py:160-178: encode -> greedy / teacher-forced decode -> cross entropy;
 *      models/txt/basic.py:49-101). The nn.LSTM / nn.Embedding / nn.Linear / nn.CrossEntropyLoss calls of the reference map to:
 * t2v_lstm_train_step      the step of t2v_lstm_seq_step, keeping what the adjoint needs: (pointer, row stride) pairs address the
 *                          state ENTERING (h_prev, c_prev) and LEAVING (h_next, c_next) the step inside [B,L,H] buffers (or h_n / c_n
 *                          for the last step); gates_t receives the post-activation i,f,g,o of the step.
 * t2v_lstm_train_step_bwd  adjoint of one step, called for the steps in reverse order. DH / DC [B,H] hold dL/dh, dL/dc of the state
 *                          leaving the step and are updated in place. Unless `first`, DH is first rebuilt from the later step:
 *                          samples that were active there get dG_later[b] . W_hh, carried samples keep DH. Then the gate adjoints
 *                          (pre-activation, order i,f,g,o) go to dG_t and DC <- dc * f; samples with t >= lengths[b] get dG_t = 0.
 *                          epilogue_only = 1 performs just the DH rebuild (gradient w.r.t. the initial state).
 * t2v_embedding_bwd        dW[tokens[n]] += g[n], n in order, without atomics (fixed summation order); dW zero-filled by the caller.
 * t2v_xent_fwd / _bwd      rows of nn.CrossEntropyLoss: loss[n] = lse[n] - x[n][target[n]];  dx = gloss[n] * (softmax(x[n]) - onehot).
 * t2v_argmax_rows          greedy decoding (basic.py:86): first index of each row's maximum. */
int t2v_lstm_train_step(const float* xproj_t, int64_t xstride, const float* w_hh_t /* [H][4H], as above */, const float* h_prev, int64_t hp_stride,
                        const float* c_prev, int64_t cp_stride, float* h_next, int64_t hn_stride, float* c_next, int64_t cn_stride,
                        float* out_t, int64_t ostride, float* gates_t, int64_t gstride, const int32_t* lengths, int t, int B, int H,
                        void* stream);
int t2v_lstm_train_step_bwd(const float* dout_t, int64_t ostride, const float* dG_later, int64_t gl_stride, int t_later,
                            const float* w_hh, float* DH, float* DC, const float* gates_t, int64_t gstride, const float* c_in,
                            int64_t ci_stride, const float* c_out, int64_t co_stride, float* dG_t, int64_t dg_stride,
                            const int32_t* lengths, int t, int B, int H, int first, int epilogue_only, void* stream);
int t2v_embedding_bwd(const float* g, const int32_t* tokens, float* dW, int64_t N, int E, int V, void* stream);
int t2v_xent_fwd(const float* logits, const int32_t* target, float* loss_rows, float* lse, int64_t rows, int V, void* stream);
int t2v_xent_bwd(const float* logits, const int32_t* target, const float* lse, const float* gloss_rows, float* dlogits, int64_t rows,
                 int V, void* stream);
int t2v_argmax_rows(const float* x, int32_t* idx, int64_t rows, int V, void* stream);

/* ---- multi-job launches: the non-local block's small ops over all pyramid levels at once -------------------------
 * Up to 8 differently shaped jobs per call (HOST array; the descriptors travel in the kernel arguments). Field roles:
 *   T2V_MJ_SCALE            out = scalar[0] * a                         n = elements
 *   T2V_MJ_SCALE_ADD        out = scalar[0] * a + b                     (layers.py:36,68: gamma * o + x)
 *   T2V_MJ_DOT              jobs[0].out[0] (+= if jobs[0].f0) sum over ALL jobs of <a, b>; ws: t2v_multi_ws_floats floats
 *   T2V_MJ_MAXPOOL          out = maxpool2x2(a), out2 = int32 argmax    n = planes, d0 = H, d1 = W  (layers.py:26-27,57-58)
 *   T2V_MJ_MAXSCATTER       out[planes,H,W] = scatter(a = g, b = idx)   T2V_MJ_MAXGATHER: out = a[b = idx]
 *   T2V_MJ_SOFTMAX          out = softmax over rows of a                n = rows, d0 = row length
 *   T2V_MJ_SOFTMAX_BWD      out = a*(b - sum(a*b))  (a = y, b = gy);  _BWD_BWD_Y: a = y, b = gy, c = gg (see t2v_softmax_bwd_bwd_y)
 *   T2V_MJ_BMM              out[b] = op(a[b]) @ op(b[b])                n = batch, d0 = M, d1 = N, d2 = K, f0 = ta, f1 = tb
 *   T2V_MJ_RELU_MASK        out = b > 0 ? a : 0                         n = elements (ReLU adjoint over several tensors)
 *   T2V_MJ_ROWSUM           out[row] = sum_s a[row][s]                  n = rows, d0 = S (resnet3d.py:48 over all levels)
 *   T2V_MJ_ROWBCAST         out[row][s] = a[row]                        its adjoint
 *   T2V_MJ_ADD              out = a + b (+ c when c != NULL)            n = elements (gradient sums where an activation feeds
 *                                                                       several consumers: one launch for all pyramid levels)
 *   T2V_MJ_CATLERP          out = [a; b] (cat along the batch), out2 (optional) = c[row] * a + (1 - c[row]) * b
 *                                                                       n = elements of a, d0 = elements per row (cond_gan.py:121-154 +
 *                                                                       losses.py:146: the D step's real||fake batch and x-hat per level)
 *   T2V_MJ_CATCOLS          out[r] = [a[r, 0:d0], b[r, 0:d1]]           n = rows (resnet3d.py:53: torch.cat((features, cond), 1), all levels)
 *   T2V_MJ_SLICECOLS        out[r, 0:d2] = a[r, d1:d1+d2], a is [rows, d0]   its adjoint pieces; T2V_MJ_EMBEDCOLS: out [rows, d0] = 0 except
 *                                                                       out[r, d1:d1+d2] = a[r, 0:d2] (the adjoint of the slice) */
enum { T2V_MJ_SCALE = 1, T2V_MJ_SCALE_ADD, T2V_MJ_DOT, T2V_MJ_MAXPOOL, T2V_MJ_MAXSCATTER, T2V_MJ_MAXGATHER, T2V_MJ_SOFTMAX,
       T2V_MJ_SOFTMAX_BWD, T2V_MJ_SOFTMAX_BWD_BWD_Y, T2V_MJ_BMM, T2V_MJ_RELU_MASK, T2V_MJ_ROWSUM, T2V_MJ_ROWBCAST, T2V_MJ_ADD, T2V_MJ_CATLERP, T2V_MJ_CATCOLS,
       T2V_MJ_SLICECOLS, T2V_MJ_EMBEDCOLS };
typedef struct t2v_multi_job {
    const void* a; const void* b; const void* c; void* out; void* out2;
    int64_t n;
    int32_t d0, d1, d2, f0, f1, reserved;
} t2v_multi_job;
int64_t t2v_multi_ws_floats(int op, const t2v_multi_job* jobs, int njobs);
int t2v_multi(int op, const t2v_multi_job* jobs, int njobs, const float* scalar_dev, float* ws, void* stream);

/* ---- losses (txt2vid/gan/losses.py) ------------------------------------------------------------ */
/* RSGAN (losses.py:79-85): loss = mean softplus(-(a-b)) ; ga = -sigmoid(-(a-b))/n * gscale, gb = -ga. */
int t2v_rsgan(const float* a, const float* b, float* loss, int n, void* stream);
int t2v_rsgan_bwd(const float* a, const float* b, const float* gloss, float* ga, float* gb, int n, void* stream);
/* mean over up to 8 pyramid levels of the relativistic loss (cond_gan.py:121-154: RSGANLoss per level, then the mean) in ONE
   launch, and one for its gradient. jobs[i]: a, b = the two logit vectors of level i (n elements); backward: out / out2 = the
   gradients w.r.t. a / b (either may be NULL), gl = dL/dloss (device scalar). Same arithmetic order as t2v_rsgan per level +
   t2v_scalar_combine with weights 1/L. */
int t2v_rsgan_mean_multi(const t2v_multi_job* jobs, int njobs, float* loss, void* stream);
int t2v_rsgan_mean_multi_bwd(const t2v_multi_job* jobs, int njobs, const float* gl, void* stream);
/* The rest of the loss zoo on D's logits (losses.py:19-68,87-133). kind: 1 VanillaGanLoss (BCE, labels wired as
   losses.py:27-28: fake->1, real->0), 2 HingeGanLoss(margin), 3 WassersteinGanLoss, 4 RaSGANLoss, 5 RaLSGANLoss;
   side 0 = discrim_loss, 1 = gen_loss. `real` may be NULL where the loss ignores it (gen side of kinds 1-3).
   Forward writes the scalar loss; backward writes d loss/d real, d loss/d fake (either may be NULL) times gloss[0]. */
enum { T2V_LOSS_VANILLA = 1, T2V_LOSS_HINGE = 2, T2V_LOSS_WASSERSTEIN = 3, T2V_LOSS_RASGAN = 4, T2V_LOSS_RALSGAN = 5 };
int t2v_gan_loss(const float* real, const float* fake, float* loss, int n_real, int n_fake, int kind, int side,
                 float margin, void* stream);
int t2v_gan_loss_bwd(const float* real, const float* fake, const float* gloss, float* g_real, float* g_fake, int n_real,
                     int n_fake, int kind, int side, float margin, void* stream);
/* GP (losses.py:135-186): xhat = alpha[b]*xr + (1-alpha[b])*xf ; sq[b] = sum g^2 ; scale rows. */
int t2v_lerp_rows(const float* alpha, const float* xr, const float* xf, float* y, int rows, int64_t S, void* stream);
int t2v_row_sqnorm(const float* g, float* out, int rows, int64_t S, void* stream);
int t2v_row_scale(const float* s, float mul, const float* g, float* y, int rows, int64_t S, void* stream);

/* ---- optimiser (torch.optim.Adam call site txt2vid/train/gan.py:93-94) ------------------------- */
int t2v_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
             float eps, float bc1, float bc2, float gscale, const float* step_dev, void* stream);
/* step_dev != NULL: a device-resident {step, 1-b1^step, 1-b2^step} triple supplies the bias corrections
 * (HIP-graph replay cannot bake a host-side step number); t2v_adam_tick advances it by one step. */
/* Multi-tensor form of t2v_adam: `jobs` is a HOST array (device pointers inside); tensors are processed 64 per launch
   with their pointers in the kernel arguments. Same arithmetic, element for element, as t2v_adam. */
typedef struct t2v_adam_job { void* p; const void* g; void* m; void* v; int64_t n; } t2v_adam_job;
int t2v_adam_multi(const t2v_adam_job* jobs, int njobs, float lr, float b1, float b2, float eps, float bc1, float bc2,
                   float gscale, const float* step_dev, void* stream);
/* torch.optim.SGD(lr, momentum) semantics, multi-tensor (the reference's --sgd branch, train/gan.py:86-89): jobs[i].m is the
   momentum buffer (may be NULL when momentum == 0), .v unused; first_step: buf = g (torch's first-step rule). */
int t2v_sgd_multi(const t2v_adam_job* jobs, int njobs, float lr, float momentum, float gscale, int first_step, void* stream);
int t2v_adam_tick(float* state, float b1, float b2, void* stream);

/* ---- pyramid (trainer.py:131-165, layers.py:106-111) ------------------------------------------- */
/* y[b,c,t,h,w] = x[b*sb, c, t*st+bt, (h*H)/Ho, (w*W)/Wo]  — Subsample + nearest F.interpolate. */
int t2v_pyramid_gather(const float* x, float* y, int B, int C, int T, int H, int W, int Bo, int To, int Ho, int Wo,
                       int sb, int st, int bt, const int32_t* bt_dev /* NULL or device-resident phase */, void* stream);


/* ---- layout glue: the permutes / slices between the reference's tensor layouts --------------------
 * (torch.stack/permute/view of models/tganv2/gen.py:75-119, torch.cat of resnet3d.py:53) */
int t2v_copy2d(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int64_t rows, int64_t cols, void* stream);
/* fp32 <-> bf16 stream (round to nearest even) for the opt-in bf16 gradient exchange of the data-parallel path (SURVEY §8(e);
 * the exchange the reference's call sites txt2vid/gan/trainer.py:240-241,262-263 imply under one process per GPU).
 * dir 0: dst (bf16) <- src (fp32); dir 1: dst (fp32) <- src (bf16); n elements, 16-byte aligned pointers. */
int t2v_cast_bf16(const void* src, void* dst, int64_t n, int dir, void* stream);
int t2v_permute01(const float* x, float* y, int64_t A, int64_t B, int64_t inner, void* stream);   /* [A,B,i]->[B,A,i]   */
int t2v_permute12(const float* x, float* y, int64_t A, int64_t B, int64_t C, int64_t inner, void* stream); /* [A,B,C,i]->[A,C,B,i] */
/* merged-frames [b*T, inner]: keep samples ::2 and frames bt::2 (gen.py:98-109); adjoint=1 scatters y into x */
int t2v_subsample_frames(float* x, float* y, int64_t b, int64_t T, int64_t inner, int64_t bo, int64_t To, int bt,
                         int adjoint, const int32_t* bt_dev /* NULL or device-resident phase */, void* stream);
/* adjoint of t2v_pyramid_gather for Ho=H, Wo=W: gx[b*sb, c, t*st+bt, :] = g[b, c, t, :] (gx pre-zeroed) */
int t2v_pyramid_scatter(const float* g, float* gx, int B, int C, int T, int64_t HW, int Bo, int To, int sb, int st,
                        int bt, void* stream);


/* out = sum_i w[i] * *ptrs[i] over n <= 16 device scalars (torch.stack(...).mean()/.sum() of
 * cond_gan.py:51-61,106-118, losses.py:207). ptrs / weights are HOST arrays. */
int t2v_scalar_combine(const void* const* ptrs, const float* weights, int n, float* out, void* stream);
/* out[r,:] = x[perm[r],:] (inverse=1: out[perm[r],:] = x[r,:]) — cond[gen_perm(B)] of cond_gan.py:133 */
int t2v_gather_rows(const float* x, const int32_t* perm, float* out, int64_t rows, int64_t cols, int inverse, void* stream);


/* ---- pooled convolution: the second 3^3 convolution of every discriminator block together with the average pooling behind it
 * (txt2vid/models/resnet3d.py:12-19 stem: conv -> AvgPool3d((1,2,2), 2); txt2vid/models/layers.py:219-243 DownBlock: conv ->
 * DownSample). Box filter and convolution commute: pool(conv3(r)) = stride-2 conv3 of the box-summed activation r~ — 27 taps
 * over the POOLED voxels (4x / 8x fewer MACs), same values up to summation order. Four pieces, closed under differentiation like
 * conv / dgrad / wgrad:
 *   t2v_pool_boxsum      r (+ fused ReLU, or a cotangent masked by [mask > 0]) -> r~ on the padded grid [N*C, Dp, H+1, W+2]
 *   t2v_pool_conv_fwd    y[N,Cout,D',H',W'] = sum_taps wp[tap] . r~[2*pos + tap + 1]   (+ bias; split-K like t2v_conv_fwd_grouped)
 *   t2v_pool_conv_dgrad  the 8 parity-class planes of the gradient on the padded grid from dL/dy (no zero-stuffed intermediate)
 *   t2v_pool_unbox       planes -> dL/dr at full resolution, ReLU mask fused (the adjoint of t2v_pool_boxsum)
 *   t2v_pool_conv_wgrad[_partial]  dW[tap] = sum_pos dL/dy[pos] (x) r~[2*pos + tap + 1] in the k-split slab format of
 *                        t2v_conv_wgrad_grouped[_partial] (same reduce kernels, same t2v_wgrad_src)
 * Members are t2v_conv_group with (D, H, W) = the FULL-RESOLUTION extents of r (H, W even >= 2; D == 1 or even) and
 * dstride = tmode: 0 no time axis (D == 1), 1 time box-summed and strided like H / W (DownSample), 2 time strided without a
 * box (the stem keeps the even frames). fwd / wgrad: x = r~, y = pooled output / dL/dy, taps in (dz,dy,dx) product order (9 or
 * 27) with widx = packed slot (mode 0). dgrad: x = dL/dy, y = planes [8][N, C, Dq, H/2+1, W/2+1], widx[f] for the 27 FORWARD
 * taps f = ((dz+1)*3 + dy+1)*3 + dx+1: the slot of the mode-1 packed weight holding that tap's matrix (ntaps ignored).
 *
 * The SAME kernels serve the generator's `Upsample(2) -> conv3x3` pairs (txt2vid/models/layers.py:152-195, UpBlock) in transposed
 * roles — nearest up-sampling is the box filter of the zero-stuffed map, so conv3(up2(x)) = box-sum of the class planes of x:
 *   forward  = t2v_pool_conv_dgrad (x = the SMALL map, mode-0 packed weight, K = Cin, C = Cout) + t2v_pool_unbox (scale 1, + bias)
 *   d/dx     = t2v_pool_boxsum(dL/dy, scale 1) + t2v_pool_conv_fwd with the mode-1 packed weight (Cin := Cout, Cout := Cin)
 *   d/dw     = t2v_pool_conv_wgrad(x = box-summed dL/dy, y = the small map) + t2v_wgrad_swap
 * — 9 taps per INPUT pixel instead of 9 per output pixel: a quarter of the MACs. */
typedef struct t2v_poolbox_job {
    const float* in;      /* boxsum: r (or a cotangent); unbox: the 8 class planes                                  */
    const float* mask;    /* optional: boxsum multiplies `in` by [mask > 0]; unbox zeroes the result where mask <= 0 */
    float* out;           /* boxsum: r~ [NC, Dp, H+1, W+2]; unbox: [NC, D, H, W]                                     */
    const float* bias;    /* unbox only, optional: out += bias[channel] (channel = (row of NC) % C)                  */
    int32_t NC, D, H, W;  /* full-resolution extents                                                                */
    int32_t tmode, relu;  /* relu: boxsum clamps its input at 0 first; unbox: number of plane sets to sum (0 / 1: one)      */
    float scale;          /* 1 / window volume (0.25 or 0.125; 1 for the up-sampling form): folded into both passes */
    int32_t C;            /* channels (only read with `bias`)                                                       */
} t2v_poolbox_job;
int t2v_pool_boxsum(const t2v_poolbox_job* jobs, int njobs, void* stream);
int t2v_pool_unbox(const t2v_poolbox_job* jobs, int njobs, void* stream);
int64_t t2v_pool_conv_fwd_ws_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout);
int t2v_pool_conv_fwd(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, const float* wp, const float* bias, float* ws,
                      int flags, void* stream);
/* the data gradient may split K: it then writes t2v_pool_conv_dgrad_splits() SETS of planes ([S][8][...] per member) and
 * t2v_pool_unbox adds them up when its job's `relu` field carries S */
int t2v_pool_conv_dgrad_splits(const t2v_conv_group* groups, int ngroups, int K, int C);
int t2v_pool_conv_dgrad(const t2v_conv_group* groups, int ngroups, int K, int C, const float* wp, void* stream);
/* kD: 3 for [Cout,Cin,3,3,3] weights, 1 for [Cout,Cin,(1,)3,3] (then no member may have a time axis) */
int64_t t2v_pool_conv_wgrad_slab_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int want_bias);
int t2v_pool_conv_wgrad(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, float* dw, float* dbias, float* slab,
                        int flags, void* stream);
int t2v_pool_conv_wgrad_partial(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, float* slab, int want_bias,
                                int flags, t2v_wgrad_src* out_src, void* stream);
/* dst[co][ci][t] (+= if accum) src[ci][co][T-1-t]: the weight gradient of the up-sampling form (below) comes out of the pooled
 * weight-gradient kernel with its operand roles — hence (co, ci) — swapped and its taps mirrored. */
int t2v_wgrad_swap(const float* src, float* dst, int Cout, int Cin, int T, int accum, void* stream);
/* Launch plans (host arithmetic): out[0] kind (9 pool forward, 10 pool data gradient, 11 pool weight gradient), out[1] tile
 * voxels, out[2] 64, out[3] 32, out[5] VECB, out[7] split-K / k-split count S. */
int t2v_pool_conv_plan(int what, const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int32_t* out);

/* ---- input row: synthetic Moving-MNIST-shaped clips + captions generated in HBM -----------------------------
 * Replaces, for machines without a dataset, the host pipeline in front of the loop (txt2vid/data/__init__.py:201-255 contract;
 * clip statistics of txt2vid/data/synthetic/generate.py:18-47,136-170): clip b is EXACTLY what the host-side
 * txt2vid_amd.data.SyntheticMovingDigits(seed=seed, size=S, channels=C, num_frames=T)[index[b]] yields (numpy RandomState
 * reproduced on the device: MT19937, 53-bit random_sample, masked-rejection randint). index_dev: int64 [B] on the device;
 * vocab_ids: HOST array of T2V_SYNTH_VOCAB ids in the order <start> digit 0..9 is left and right top bottom <end>;
 * vids: float [B,T,C,S,S] in [-1,1] (S % 4 == 0, S >= 28, T >= 2); tokens: int64 [B,8]; err_dev: optional int32 the kernel sets
 * to 1 if a draw sequence ran out (never in practice; the caller zeroes it). */
#define T2V_SYNTH_VOCAB 19
int t2v_synth_clips(const int64_t* index_dev, int B, int64_t seed, int T, int C, int S, const int32_t* vocab_ids, float* vids,
                    int64_t* tokens, int32_t* err_dev, void* stream);

/* ---- optional launch instrumentation for bench.py's roofline line -----------------------------------
 * Between begin and end every conv-GEMM launch is bracketed by a hipEvent pair recorded on the launch
 * stream. end() synchronises and fills out[kind*3+{0,1,2}] = {total ms, executed flops, launches};
 * kind 0 = forward/data-gradient implicit GEMM, 1 = weight-gradient GEMM, 2 = its split-K reduce.
 * Returns 1 if the record pool overflowed (totals then cover the recorded launches only). */
int t2v_prof_begin(int max_records);
int t2v_prof_end(double* out, int nkinds);

const char* t2v_version(void);

/* Fused non-local attention of the generator's 2-D block (txt2vid/models/layers.py:23-36: theta^T phi -> softmax over the
 * pooled positions -> weighted sum of g) without materialising beta [b, N, Nk]: o[b][c][i] = sum_j softmax_j(theta[:, i] .
 * phi[:, j]) g[c][j]; theta [b,C8,N], phi [b,C8,Nk], g [b,C2,Nk], o [b,C2,N], lse [b,N] (row log-sum-exp, kept for the
 * adjoint). t2v_nonlocal_bwd: dtheta, dphi, dg from dL/do with beta recomputed from lse; ws: b*N floats.
 * Built for the head sizes of the generator's block (C8 = 4, C2 = 16: t2v_nonlocal_ok); other sizes and the
 * discriminator's 3-D block (which needs a second-order adjoint) stay on t2v_bmm / t2v_softmax. */
int t2v_nonlocal_ok(int C8, int C2);
int t2v_nonlocal_fwd(const float* theta, const float* phi, const float* g, float* o, float* lse, int b, int C8, int C2, int N,
                     int Nk, void* stream);
int t2v_nonlocal_bwd(const float* theta, const float* phi, const float* g, const float* o, const float* lse, const float* go,
                     float* dtheta, float* dphi, float* dg, float* ws, int b, int C8, int C2, int N, int Nk, void* stream);

#ifdef __cplusplus
}
#endif
#endif
