// conv.hip — stride-1 "same" convolution family (3-D / 2-D / Linear) for gfx950 as implicit GEMM on
// the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 157 TFLOP/s peak).
//
//   forward / data-gradient : Y^T[co][m] = sum_{tap,ci} Wp[tap][ci][co] * X[m + off(tap)][ci]
//   weight-gradient         : dW[co][ci][tap] = sum_m  gY[m][co] * X[m + off(tap)][ci]
//
// m runs over the N*D*H*W output voxels (NCDHW: consecutive m = consecutive w = consecutive addresses),
// so the MFMA "B" operand (columns = m) is gathered with lane-contiguous loads and the accumulator
// tile (rows = co in registers, columns = m on lanes) stores 128-byte segments straight into NCDHW.
// Replaces the cuDNN conv fwd/dgrad/wgrad behind txt2vid/models/resnet3d.py:13-18,
// layers.py:174-183,231-238,251, conv_lstm.py:19-26 (reference root: miguelmartin75/txt2vid).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/t2v_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// In-kernel timeline stamps: a DIAGNOSTIC build only (-DT2V_STAMPS, tools/stamps.py); the product library has none of it.
// One record of 16 x u64 per wave: [0] entry, [1] loop start, [2] loop end, [3] exit (s_memtime ticks), [4..8] summed ticks of
// the stage / first barrier / load issue / MFMA loop / second barrier phases, [9] rounds, [10] HW_ID, [11] XCC_ID, [12] s_memrealtime at entry.
#ifdef T2V_STAMPS
__device__ unsigned long long* g_stamps = nullptr;
__device__ unsigned int g_stamps_cap = 0;
extern "C" int t2v_stamps_set(void* buf, unsigned int cap_records) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_cap), &cap_records, sizeof(cap_records)) != hipSuccess) return -1;
    return 0;
}
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define STAMP(v) const unsigned long long v = stamp_now()
#define STAMP_ACC(acc, a, b) acc += (b) - (a)
#else
#define STAMP(v)
#define STAMP_ACC(acc, a, b)
#endif

// XCD-aware workgroup remap (bijective for any count): the dispatcher deals workgroups round-robin over the 8
// XCDs, each with a private L2; id -> slot such that the workgroups of ONE XCD own a contiguous range of slots,
// so that neighbours in the slot order (adjacent voxel tiles sharing halos, the taps of one k-split) share an L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// q = m / d for 0 <= m < 2^24 via the float reciprocal (+-1 fix-up): ~8 VALU instead of the ~40 of an
// integer division. The per-chunk voxel decode of the weight-gradient kernels runs three of these per lane.
__device__ __forceinline__ int fast_div(int m, int d, float rcp) {
    int q = (int)((float)m * rcp);
    int r = m - q * d;
    q += (r >= d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

#define BK 16

// hipGetLastError() is sticky per thread and also reports errors of calls the HOST framework made and
// handled earlier; clear it before every launch so that launch_status() reflects this launch only.
#define T2V_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

static inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? T2V_OK : -(int)e - 1000;
}


// ------------------------------------------------------------------------------------------------
// optional launch instrumentation (bench.py roofline): hipEvent pairs recorded ON THE LAUNCH STREAM
// around every conv-GEMM launch while enabled. Off by default; the product path never enables it.
// ------------------------------------------------------------------------------------------------
#include <hip/hip_ext.h>
#include <vector>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
struct ProfRec { hipEvent_t e0, e1; double flops; int kind; long M; int Cin, Cout, taps, groups, S; int32_t plan[8]; };
static struct {
    bool on = false;
    std::vector<ProfRec> recs;
    size_t used = 0;
} g_prof;

// A ProfScope arms ONE record; the next T2V_LAUNCH_PROF in the thread consumes it and launches through
// hipExtLaunchKernelGGL, which stamps the two events with the dispatch's own begin / end times (the same clock the
// rocprofv3 kernel trace reports), so neither the launch gap nor the event packets are counted.
static thread_local ProfRec* g_prof_cur = nullptr;
struct ProfScope {
    ProfScope(int kind, double flops, hipStream_t, long M = 0, int Cin = 0, int Cout = 0, int taps = 0, int groups = 0, int S = 0) {
        g_prof_cur = nullptr;
        if (g_prof.on && g_prof.used < g_prof.recs.size()) {
            ProfRec* r = &g_prof.recs[g_prof.used++];
            r->kind = kind;
            r->flops = flops;
            r->M = M; r->Cin = Cin; r->Cout = Cout; r->taps = taps; r->groups = groups; r->S = S;
            for (int i = 0; i < 8; ++i) r->plan[i] = -1;
            g_prof_cur = r;
        }
    }
    ~ProfScope() { g_prof_cur = nullptr; }
    // the launch plan (t2v_conv_fwd_plan / t2v_conv_wgrad_plan layout) of the launch this scope brackets, for the dump
    static void set_plan(const int32_t* plan, int n) {
        if (g_prof_cur) for (int i = 0; i < n && i < 8; ++i) g_prof_cur->plan[i] = plan[i];
    }
};
#define T2V_LAUNCH_PROF(kernel, grid, block, shm, stream, ...) do {                                                      \
        (void)hipGetLastError();                                                                                          \
        if (g_prof_cur) {                                                                                                 \
            ProfRec* r_ = g_prof_cur; g_prof_cur = nullptr;                                                               \
            hipExtLaunchKernelGGL(kernel, grid, block, shm, stream, r_->e0, r_->e1, 0, __VA_ARGS__);                      \
        } else hipLaunchKernelGGL(kernel, grid, block, shm, stream, __VA_ARGS__);                                         \
    } while (0)

extern "C" int t2v_prof_begin(int max_records) {
    if (max_records < 1 || max_records > (1 << 20)) return T2V_EINVAL;
    if ((int)g_prof.recs.size() < max_records) {
        size_t old = g_prof.recs.size();
        g_prof.recs.resize(max_records);
        for (size_t i = old; i < g_prof.recs.size(); ++i) {
            if (hipEventCreate(&g_prof.recs[i].e0) != hipSuccess || hipEventCreate(&g_prof.recs[i].e1) != hipSuccess) return T2V_ELAUNCH;
        }
    }
    g_prof.used = 0;
    g_prof.on = true;
    return T2V_OK;
}
// out[kind*3 + {0,1,2}] = {total ms, total flops, launches} for kind in 0..nkinds-1. Synchronises.
extern "C" int t2v_prof_end(double* out, int nkinds) {
    g_prof.on = false;
    if (!out || nkinds < 1) return T2V_EINVAL;
    for (int i = 0; i < nkinds * 3; ++i) out[i] = 0.0;
    FILE* dump = nullptr;
    if (const char* path = getenv("T2V_PROF_DUMP")) dump = fopen(path, "w");   // developer aid: one line per launch
    if (dump) fprintf(dump, "kind,flops,ms,M,Cin,Cout,taps,groups,S,plan\n");
    for (size_t i = 0; i < g_prof.used; ++i) {
        ProfRec& r = g_prof.recs[i];
        if (hipEventSynchronize(r.e1) != hipSuccess) return T2V_ELAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return T2V_ELAUNCH;
        if (r.kind >= 0 && r.kind < nkinds) { out[r.kind * 3] += ms; out[r.kind * 3 + 1] += r.flops; out[r.kind * 3 + 2] += 1.0; }
        if (dump) fprintf(dump, "%d,%.0f,%.6f,%ld,%d,%d,%d,%d,%d,%d:%d:%d:%d:%d:%d:%d:%d\n", r.kind, r.flops, ms, r.M, r.Cin, r.Cout, r.taps,
                          r.groups, r.S, r.plan[0], r.plan[1], r.plan[2], r.plan[3], r.plan[4], r.plan[5], r.plan[6], r.plan[7]);
    }
    if (dump) fclose(dump);
    int dropped = (g_prof.used >= g_prof.recs.size()) ? 1 : 0;
    g_prof.used = 0;
    return dropped;    // 1: the record pool filled up (totals cover the recorded launches only)
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
struct TapList { int32_t n; int32_t t[T2V_MAX_TAPS]; };

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                          int Cout, int Cin, int T, TapList taps, int mode, int dst_rows,
                                                          int dst_cols, int row_off, int col_off) {
    // mode 0: wp[j][row_off + ci][col_off + co] = w[co][ci][t_j]      (per-tap matrix [dst_rows][dst_cols], inner = co)
    // mode 1: wp[j][row_off + co][col_off + ci] = w[co][ci][T-1-t_j]  (inner = ci)
    // dst_rows / dst_cols > the weight's own extents let several weights share one packed matrix (the four
    // ConvLSTM gates become one GEMM). 32x32 tile transpose of the (co,ci) plane through LDS for mode 0.
    __shared__ float tile[32][33];
    const int j = blockIdx.z;
    const bool src_tm = mode & 8;                   // source stored tap-major: w[t][co][ci]
    mode &= 7;
    const int t = mode ? (T - 1 - taps.t[j]) : taps.t[j];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const size_t sco = src_tm ? (size_t)Cin : (size_t)Cin * T, sci = src_tm ? 1 : (size_t)T;
    const size_t st = src_tm ? (size_t)t * Cout * Cin : (size_t)t;
    if (mode == 0) {
        for (int r = ty; r < 32; r += 8) {          // read rows co, lanes ci (stride T)
            int co = co0 + r, ci = ci0 + tx;
            tile[r][tx] = (co < Cout && ci < Cin) ? w[co * sco + ci * sci + st] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {          // write rows ci, lanes co
            int ci = ci0 + r, co = co0 + tx;
            if (ci < Cin && co < Cout) wp[((size_t)j * dst_rows + row_off + ci) * dst_cols + col_off + co] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            int co = co0 + r, ci = ci0 + tx;
            if (co < Cout && ci < Cin)
                wp[((size_t)j * dst_rows + row_off + co) * dst_cols + col_off + ci] = w[co * sco + ci * sci + st];
        }
    }
}

extern "C" int t2v_pack_weight_into(const float* w, float* wp, int Cout, int Cin, int T, const int32_t* taps, int ntaps,
                                    int mode, int dst_rows, int dst_cols, int row_off, int col_off, void* stream) {
    if (!w || !wp || ntaps < 1 || ntaps > T2V_MAX_TAPS || T > T2V_MAX_TAPS || row_off < 0 || col_off < 0) return T2V_EINVAL;
    if ((mode & ~8) != 0 && (mode & ~8) != 1) return T2V_EINVAL;
    const int rows = (mode & 7) ? Cout : Cin, cols = (mode & 7) ? Cin : Cout;
    if (row_off + rows > dst_rows || col_off + cols > dst_cols) return T2V_EINVAL;
    TapList tl;
    tl.n = ntaps;
    for (int i = 0; i < ntaps; ++i) {
        if (taps[i] < 0 || taps[i] >= T) return T2V_EINVAL;
        tl.t[i] = taps[i];
    }
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, ntaps);
    T2V_LAUNCH(pack_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, T, tl, mode, dst_rows, dst_cols,
               row_off, col_off);
    return launch_status();
}

extern "C" int t2v_pack_weight(const float* w, float* wp, int Cout, int Cin, int T, const int32_t* taps, int ntaps,
                               int mode, void* stream) {
    return t2v_pack_weight_into(w, wp, Cout, Cin, T, taps, ntaps, mode, (mode & 7) ? Cout : Cin, (mode & 7) ? Cin : Cout, 0, 0, stream);
}


// ------------------------------------------------------------------------------------------------
// multi-tensor pack: every packed variant of every weight an optimiser just updated, in ONE launch.
// `table` is a device array of PackJob; workgroup b serves the job whose [block_begin, next) range holds b.
// ------------------------------------------------------------------------------------------------
struct PackJob {
    const float* src;
    float* dst;
    int32_t Cout, Cin, T, ntaps, mode, dst_rows, dst_cols, row_off, col_off, block_begin, bx, by;
    int8_t taps[T2V_MAX_TAPS];
    int8_t pad_[5];
};

// (round 3: a form where one workgroup reads its tile's rows as contiguous runs of (tile ci) x T floats and writes all live taps from
// LDS — no T-fold line re-reads — was measured and dropped: 141 vs 125 us per launch; the re-reads hit L2 and the 23 000 small
// workgroups of this form keep more bytes in flight.)
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackJob* __restrict__ table, int njobs) {
    __shared__ float tile[32][33];
    __shared__ PackJob job;
    // binary search: last job with block_begin <= blockIdx.x
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        job = table[lo];
    }
    __syncthreads();
    const int b = (int)blockIdx.x - job.block_begin;
    const int bxi = b % job.bx, byi = (b / job.bx) % job.by, j = b / (job.bx * job.by);
    const int Cout = job.Cout, Cin = job.Cin, T = job.T;
    // mode 0 / 1: fp32 forward / mirrored data-gradient layouts (see t2v_pack_weight); mode 2 / 3: the bf16-compute layouts
    // (t2v_pack_weight_bf16 mode 0 / 1): 2 = dst[j][co][ci], 3 = dst[j][ci][co] mirrored, both bf16
    const int jm = job.mode & 7;
    const bool src_tm = job.mode & 8;               // source stored tap-major: w[t][co][ci]
    const bool mirror = jm == 1 || jm == 3, transpose = jm == 0 || jm == 3, b16 = jm >= 2;
    const int t = mirror ? (T - 1 - job.taps[j]) : job.taps[j];
    const size_t sco = src_tm ? (size_t)Cin : (size_t)Cin * T, sci = src_tm ? 1 : (size_t)T;
    const size_t st = src_tm ? (size_t)t * Cout * Cin : (size_t)t;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int co0 = byi * 32, ci0 = bxi * 32;
    const float* __restrict__ w = job.src;
    float* __restrict__ wp = job.dst;
    __bf16* __restrict__ wpb = reinterpret_cast<__bf16*>(job.dst);
    if (transpose) {
        for (int r = ty; r < 32; r += 8) {
            int co = co0 + r, ci = ci0 + tx;
            tile[r][tx] = (co < Cout && ci < Cin) ? w[co * sco + ci * sci + st] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {
            int ci = ci0 + r, co = co0 + tx;
            if (ci < Cin && co < Cout) {
                const size_t o = ((size_t)j * job.dst_rows + job.row_off + ci) * job.dst_cols + job.col_off + co;
                if (b16) wpb[o] = (__bf16)tile[tx][r]; else wp[o] = tile[tx][r];
            }
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            int co = co0 + r, ci = ci0 + tx;
            if (co < Cout && ci < Cin) {
                const size_t o = ((size_t)j * job.dst_rows + job.row_off + co) * job.dst_cols + job.col_off + ci;
                const float v = w[co * sco + ci * sci + st];
                if (b16) wpb[o] = (__bf16)v; else wp[o] = v;
            }
        }
    }
}

extern "C" int t2v_pack_job_bytes(void) { return (int)sizeof(PackJob); }

// `table`: DEVICE array of njobs PackJob records (layout: t2v_pack_job in the header); total_blocks = sum of bx*by*ntaps
extern "C" int t2v_pack_multi(const void* table, int njobs, int total_blocks, void* stream) {
    if (!table || njobs < 1 || total_blocks < 1) return T2V_EINVAL;
    T2V_LAUNCH(pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackJob*)table, njobs);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad implicit GEMM — GROUPED: one launch convolves up to T2V_MAX_GROUPS tensors that share
// the weights but not the geometry (the 4 pyramid levels of the multi-scale discriminator). A workgroup
// owns one BM x BN tile of ONE group (looked up from the tile prefix table); everything per-group
// (pointers, extents, valid taps, packed-weight slot of each tap) comes from the group descriptor.
//   tile BM (voxels) x BN (output channels), K consumed in chunks of BKT (one tap x BKT channels on the
//   FAST path), 4 waves, each owning (BN/WAVES_CO) x (BM/WAVES_M) as 32x32 MFMA tiles.
//   Global -> registers (issued one chunk ahead, right after the barrier) -> LDS -> MFMA.
//   Split-K (gridDim.z > 1): every split writes its partial tile into slab[z] and a second kernel sums
//   the splits in a fixed order (deterministic) and adds the bias: the layers with M of a few dozen
//   voxels and K of several thousand (the deep discriminator blocks, the ConvLSTM) are otherwise a
//   handful of workgroups each walking K serially at memory latency.
// ------------------------------------------------------------------------------------------------
struct GroupTable {
    t2v_conv_group g[T2V_MAX_GROUPS];
    int32_t tile_start[T2V_MAX_GROUPS + 1];   // first m-tile of each group (prefix sum)
    int64_t out_start[T2V_MAX_GROUPS + 1];    // first element of each group in the virtual concatenated output
    int32_t n;
};

// ---- epilogue shared by the implicit-GEMM kernels: rows (registers) = co, columns (lanes) = m
template <int NCO, int NM, int WCO, int WM>
__device__ __forceinline__ void igemm_epilogue(f32x16 (&acc)[NCO][NM], const GroupTable& tab, const int gi, const t2v_conv_group& gd,
                                               const float* __restrict__ bias, float* __restrict__ slab, const int Cout,
                                               const int flags, const int nsplit, const int m0, const int co0, const int M,
                                               const int DHW, const int wm, const int wco, const int l31, const int hi,
                                               const int yds = 1, const int yoff = 0, const int DHWy = 0, const int HW = 1) {
    // yds == 2 (ydstride): row (n, e, r) of the GEMM goes to frame 2e + yoff of a y with DHWy voxels per sample and channel
    const bool split = nsplit > 1;
    const bool has_bias = !split && (flags & T2V_CONV_BIAS) && bias != nullptr;
    const bool accum = !split && (flags & T2V_CONV_ACCUM);
    const bool mask_out = !split && (flags & T2V_CONV_MASK_OUT) && gd.mask != nullptr;
    float* out = split ? slab + (size_t)blockIdx.z * (size_t)tab.out_start[tab.n] + (size_t)tab.out_start[gi] : gd.y;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int m = m0 + wm * WM + j * 32 + l31;
        if (m >= M) continue;
        const int n = m / DHW;
        int sp = m - n * DHW;
        int cs = DHW;                      // channel stride of y
        if (yds == 2) {
            const int e = sp / HW;
            sp += (e + yoff) * HW;         // (2e + yoff) * HW + r
            cs = DHWy;
        }
        float* py = out + (size_t)n * Cout * cs + sp;
        if (mask_out) {                    // ReLU adjoint fused: y = mask > 0 ? result : 0
            const float* pm = gd.mask + (size_t)n * Cout * cs + sp;
#pragma unroll
            for (int i = 0; i < NCO; ++i) {
                float mv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco * WCO + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    mv[r] = pm[(size_t)(co < Cout ? co : Cout - 1) * cs];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco * WCO + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float v = acc[i][j][r] + (has_bias ? bias[co < Cout ? co : 0] : 0.f);
                    if (co < Cout) py[(size_t)co * cs] = mv[r] > 0.f ? v : 0.f;
                }
            }
        } else if (!accum) {               // (block-uniform) plain stores: no load, no wait in the store tail
#pragma unroll
            for (int i = 0; i < NCO; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco * WCO + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (co < Cout) py[(size_t)co * cs] = acc[i][j][r] + (has_bias ? bias[co] : 0.f);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCO; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco * WCO + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (co < Cout) py[(size_t)co * cs] += acc[i][j][r] + (has_bias ? bias[co] : 0.f);
                }
            }
        }
    }
}


template <int BM, int BN, int WAVES_CO, int BKT, bool FAST, bool VECB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const GroupTable tab, const float* __restrict__ wp,
                                                         const float* __restrict__ bias, float* __restrict__ slab,
                                                         const int Cin, const int Cout, const int flags, const int nsplit) {
    constexpr int WAVES_M = 4 / WAVES_CO;
    constexpr int WCO = BN / WAVES_CO;      // co extent per wave
    constexpr int WM = BM / WAVES_M;        // m extent per wave
    constexpr int NCO = WCO / 32, NM = WM / 32;
    constexpr int LA = BKT * BM / 256;
    constexpr int KSA = 256 / BM;           // k stride between a thread's successive A loads
    constexpr int LB = BKT * BN / 256;      // scalar B loads per thread
    constexpr int KSB = 256 / BN;
    constexpr int NV = BKT * BN / 4;        // float4s in the B tile
    constexpr int LBV = NV >= 256 ? NV / 256 : 1;   // float4 B loads per thread (threads >= NV idle when NV < 256)
    constexpr int KSBV = 1024 / BN;         // k rows covered by one float4 pass of the workgroup
    static_assert(NCO >= 1 && NM >= 1 && LA >= 1 && LB >= 1, "tile");

    __shared__ __attribute__((aligned(16))) float As[2 * BKT * BM];     // two stages
    __shared__ __attribute__((aligned(16))) float Bs[2 * BKT * BN];
    __shared__ int s_off[T2V_MAX_TAPS];
    __shared__ int s_widx[T2V_MAX_TAPS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave % WAVES_CO, wm = wave / WAVES_CO;

    // ---- which group does this m-tile belong to (block-uniform scan of <= 8 entries)
    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && tile >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const float* __restrict__ x = gd.x;
    const int D = gd.D, H = gd.H, W = gd.W;
    const int HW = H * W, DHW = D * HW;
    const int M = gd.N * DHW;
    const int m0 = (tile - tab.tile_start[gi]) * BM, co0 = blockIdx.y * BN;
    const int ntaps = gd.ntaps;

    const int lane_t = lane < ntaps ? lane : 0;
    const int tab_tdz = gd.dz[lane_t], tab_tdy = gd.dy[lane_t], tab_tdx = gd.dx[lane_t];
    if (tid < ntaps) {
        s_off[tid] = tab_tdz * HW + tab_tdy * W + tab_tdx;
        s_widx[tid] = gd.widx[tid];
    }

    // ---- per-thread gather coordinates (fixed m for the whole K loop)
    const int ma_l = tid % BM, ka_l = tid / BM;
    const int m_a = m0 + ma_l;
    uint32_t tapmask = 0;
    size_t xbase = 0;
    if (m_a < M) {
        int n = m_a / DHW, sp = m_a - n * DHW;
        int d = sp / HW, r = sp - d * HW;
        int h = r / W, w_ = r - h * W;
        xbase = (size_t)n * Cin * DHW + sp;
        for (int t = 0; t < ntaps; ++t) {            // (per-lane table copies + v_readlane: see conv_igemm_strip3_kernel)
            int dd = d + __builtin_amdgcn_readlane(tab_tdz, t), hh = h + __builtin_amdgcn_readlane(tab_tdy, t),
                ww = w_ + __builtin_amdgcn_readlane(tab_tdx, t);
            if ((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) tapmask |= 1u << t;
        }
    }
    const int cob_l = tid % BN, kb_l = tid / BN;                 // scalar B mapping
    const int cv_l = (tid % (BN / 4)) * 4, kv_l = tid / (BN / 4); // float4 B mapping
    const bool vact = (NV >= 256) || (tid < NV);
    const bool co_ok = VECB ? (vact && (co0 + cv_l) < Cout) : (co0 + cob_l) < Cout;
    const bool relu_in = flags & T2V_CONV_RELU_IN;

    f32x16 acc[NCO][NM];
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[LA];
    float rb[VECB ? 1 : LB];
    float4 rbv[VECB ? LBV : 1];
    const int Ktot = ntaps * Cin;
    const int nchunks = FAST ? ntaps * (Cin / BKT) : (Ktot + BKT - 1) / BKT;
    const int cps = (nchunks + nsplit - 1) / nsplit;          // this group's chunks per split
    const int q0 = blockIdx.z * cps;
    int q1 = q0 + cps;
    if (q1 > nchunks) q1 = nchunks;
    __syncthreads();   // s_off / s_widx visible

    // Loads are UNCONDITIONAL from clamped (always valid) addresses and masked afterwards: a
    // `valid ? *p : 0` select makes hipcc branch around every load and wait for it (s_waitcnt per element),
    // which serialises the 16 gathers of a chunk at full memory latency.
    // The masking (and the fused ReLU) is applied when the registers are written to LDS, one chunk later, so
    // that nothing consumes the loads before the MFMAs of the current chunk have been issued.
    const float* wp_safe = wp;
    uint32_t pend_a = 0;           // validity bits of the pending chunk's A elements
    const bool dbg_noload = flags & 64, dbg_nostage = flags & 128;
    auto load_chunk = [&](int q, int t, int c0) {
        if (dbg_noload && q != q0) return;
        if (FAST) {
            const bool v = (tapmask >> t) & 1u;
            pend_a = v ? 0xFFFFFFFFu : 0u;
            const ptrdiff_t off = v ? (ptrdiff_t)s_off[t] : 0;
            const float* px = x + xbase + off + (size_t)(c0 + ka_l) * DHW;
#pragma unroll
            for (int j = 0; j < LA; ++j) ra[j] = px[(size_t)j * KSA * DHW];
            if (VECB) {
                const float* pw = co_ok ? wp + ((size_t)s_widx[t] * Cin + c0 + kv_l) * Cout + co0 + cv_l : wp_safe;
#pragma unroll
                for (int j = 0; j < LBV; ++j) rbv[j] = *reinterpret_cast<const float4*>(pw + (size_t)j * KSBV * Cout);
            } else {
                const float* pw = co_ok ? wp + ((size_t)s_widx[t] * Cin + c0 + kb_l) * Cout + co0 + cob_l : wp_safe;
#pragma unroll
                for (int j = 0; j < LB; ++j) rb[j] = pw[(size_t)j * KSB * Cout];
            }
        } else {
            pend_a = 0;
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const int kk = q * BKT + ka_l + j * KSA;
                const int kc = kk < Ktot ? kk : Ktot - 1;
                const int tt = kc / Cin, ci = kc - tt * Cin;
                const bool v = kk < Ktot && ((tapmask >> tt) & 1u);
                pend_a |= (v ? 1u : 0u) << j;
                const ptrdiff_t off = v ? (ptrdiff_t)s_off[tt] : 0;
                ra[j] = x[xbase + off + (size_t)ci * DHW];
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                const int kk = q * BKT + kb_l + j * KSB;
                const int kc = kk < Ktot ? kk : Ktot - 1;
                const int tt = kc / Cin, ci = kc - tt * Cin;
                // rows of the K padding (kk >= Ktot) multiply A elements that are masked to zero
                rb[j] = wp[((size_t)s_widx[tt] * Cin + ci) * Cout + (co_ok ? co0 + cob_l : 0)];
            }
        }
    };

    int t_cur = 0, c_cur = 0;
    if (FAST) {
        const int cpt = Cin / BKT;
        t_cur = q0 / cpt;
        c_cur = (q0 - t_cur * cpt) * BKT;
    }
    auto advance = [&]() {
        c_cur += BKT;
        if (FAST && c_cur >= Cin) { c_cur = 0; ++t_cur; }
    };
    // registers -> LDS buffer `b` (masking + fused ReLU happen here)
    auto stage = [&](int b) {
        float* as = As + b * (BKT * BM);
        float* bs = Bs + b * (BKT * BN);
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            float val = ((pend_a >> j) & 1u) ? ra[j] : 0.f;
            as[(ka_l + j * KSA) * BM + ma_l] = relu_in ? fmaxf(val, 0.f) : val;
        }
        if (VECB) {
#pragma unroll
            for (int j = 0; j < LBV; ++j)
                if (vact) *reinterpret_cast<float4*>(&bs[(kv_l + j * KSBV) * BN + cv_l]) = co_ok ? rbv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int j = 0; j < LB; ++j) bs[(kb_l + j * KSB) * BN + cob_l] = co_ok ? rb[j] : 0.f;
        }
    };
    // Two LDS stages, ONE barrier per chunk: while the waves of this workgroup multiply chunk q out of
    // stage `cur`, chunk q+1 sits in registers (its loads were issued a whole MFMA phase ago) and is written
    // to the other stage right after this wave's MFMAs; chunk q+2's loads are issued after the barrier.
    int cur = 0;
    if (q0 < q1) {
        load_chunk(q0, t_cur, c_cur);
        stage(0);
        __syncthreads();
        if (q0 + 1 < q1) { advance(); load_chunk(q0 + 1, t_cur, c_cur); }
    }
    for (int q = q0; q < q1; ++q) {
        const float* as = As + cur * (BKT * BM);
        const float* bs = Bs + cur * (BKT * BN);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k2 = 0; k2 < BKT / 2; ++k2) {
            float a[NCO], b[NM];
            const int krow = k2 * 2 + hi;
#pragma unroll
            for (int i = 0; i < NCO; ++i) a[i] = bs[krow * BN + wco * WCO + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < NM; ++j) b[j] = as[krow * BM + wm * WM + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < NCO; ++i)
#pragma unroll
                for (int j = 0; j < NM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (q + 1 < q1 && !dbg_nostage) stage(cur ^ 1);
        __syncthreads();
        if (q + 2 < q1) { advance(); load_chunk(q + 2, t_cur, c_cur); }
        cur ^= 1;
    }

    igemm_epilogue<NCO, NM, WCO, WM>(acc, tab, gi, gd, bias, slab, Cout, flags, nsplit, m0, co0, M, DHW, wm, wco, l31, hi);
}

// ------------------------------------------------------------------------------------------------
// STRIP variant of the implicit GEMM for kernels that are 3 taps wide along W (all the 3x3x3 / 3x3 layers on maps
// wider than one voxel). The three dx taps of one (dz,dy) row read the same input row shifted by one voxel, so the
// A tile is gathered and staged ONCE per (row, channel block) as a strip of BM + 2 voxels and the three chunks read
// it at offsets -1 / 0 / +1 (row ends masked at read time, one v_cndmask per operand): a third of the scattered
// global gathers and LDS writes per MFMA. Chunk order: (row tap, channel block, dx) with dx fastest; weights are
// staged per chunk as before. Requires taps in product order with dx fastest (what conv_geom emits) and Cin % BKT == 0.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_CO, int BKT, bool VECB>
__global__ __launch_bounds__(256) void conv_igemm_strip_kernel(const GroupTable tab, const float* __restrict__ wp,
                                                               const float* __restrict__ bias, float* __restrict__ slab,
                                                               const int Cin, const int Cout, const int flags, const int nsplit) {
    constexpr int WAVES_M = 4 / WAVES_CO;
    constexpr int WCO = BN / WAVES_CO;
    constexpr int WM = BM / WAVES_M;
    constexpr int NCO = WCO / 32, NM = WM / 32;
    constexpr int AP = BM + 4;              // [left halo][BM voxels][right halo][pad]
    constexpr int LA = BKT * BM / 256;
    constexpr int KSA = 256 / BM;
    constexpr int LB = BKT * BN / 256;
    constexpr int KSB = 256 / BN;
    constexpr int NV = BKT * BN / 4;
    constexpr int LBV = NV >= 256 ? NV / 256 : 1;
    constexpr int KSBV = 1024 / BN;
    static_assert(NCO >= 1 && NM >= 1 && LA >= 1 && LB >= 1 && 2 * BKT <= 256, "tile");

    __shared__ __attribute__((aligned(16))) float As[2 * BKT * AP];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BKT * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave % WAVES_CO, wm = (wave / WAVES_CO) % WAVES_M;

    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && tile >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int D = gd.D, H = gd.H, W = gd.W;
    const int HW = H * W, DHW = D * HW;
    const int M = gd.N * DHW;
    const int m0 = (tile - tab.tile_start[gi]) * BM, co0 = blockIdx.y * BN;
    const int ntaps = gd.ntaps;
    const int ndx = gd.dx[0] < 0 ? 3 : 1;           // taps r*ndx + {0,1,2} = dx -1, 0, +1 of row tap r
    const int nrow = ntaps / ndx;

    // per-tap tables held one entry per LANE (read back with v_readlane at the wave-uniform tap index: no LDS round trip
    // and no wait in the chunk loop): byte offset of row tap r inside x, packed-weight slot of tap t
    const int lane_r = lane < nrow ? lane : 0, lane_t = lane < ntaps ? lane : 0;
    // (the mask loops below read these per-lane copies with v_readlane: indexing the kernel-argument tables with the loop counter
    // made hipcc issue a global_load_sbyte + s_waitcnt vmcnt(0) PER ITERATION — 9 x 2 dependent loads = most of a ~4 us prologue)
    const int tab_rdz = gd.dz[lane_r * ndx], tab_rdy = gd.dy[lane_r * ndx];
    const int tab_roff = (tab_rdz * HW + tab_rdy * W) * 4;
    const int tab_widx = gd.widx[lane_t];
    // buffer descriptors (wave-uniform: kernel arguments only). All gathers use 32-bit byte offsets — the host only selects
    // this kernel when every member's M * Cin * 4 fits 32 bits — with the channel stride in the scalar offset.
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)M * (uint32_t)Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, -1, 0x00020000);

    // ---- staging coordinates: this thread's voxel of the strip body, and (threads < 2 BKT) one halo element
    const int ma_l = tid % BM, ka_l = tid / BM;
    const int he = tid / BKT, hk = tid % BKT;        // halo: he = 0 left (voxel m0 - 1), 1 right (voxel m0 + BM)
    const bool halo_thread = tid < 2 * BKT;
    uint32_t rowmask = 0, rowmask_h = 0;
    uint32_t xbase = 0, xbase_h = 0;         // element offsets of (n, channel 0, voxel)
    {
        const int m_a = m0 + ma_l;
        if (m_a < M) {
            const int n = m_a / DHW, sp = m_a - n * DHW;
            const int d = sp / HW, r = sp - d * HW;
            const int h = r / W;
            xbase = (uint32_t)n * (uint32_t)Cin * (uint32_t)DHW + (uint32_t)sp;
            for (int t = 0; t < nrow; ++t) {
                const int dd = d + __builtin_amdgcn_readlane(tab_rdz, t), hh = h + __builtin_amdgcn_readlane(tab_rdy, t);
                if ((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H) rowmask |= 1u << t;
            }
        }
        const int m_h = he ? m0 + BM : m0 - 1;
        if (halo_thread && m_h >= 0 && m_h < M) {
            const int n = m_h / DHW, sp = m_h - n * DHW;
            const int d = sp / HW, r = sp - d * HW;
            const int h = r / W;
            xbase_h = (uint32_t)n * (uint32_t)Cin * (uint32_t)DHW + (uint32_t)sp;
            for (int t = 0; t < nrow; ++t) {
                const int dd = d + __builtin_amdgcn_readlane(tab_rdz, t), hh = h + __builtin_amdgcn_readlane(tab_rdy, t);
                if ((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H) rowmask_h |= 1u << t;
            }
        }
    }
    // ---- reader coordinates: may this lane's voxel look one step left / right inside its row?
    bool can_l[NM], can_r[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int m = m0 + wm * WM + j * 32 + l31;
        const int w_ = m % W;
        can_l[j] = w_ > 0;
        can_r[j] = w_ < W - 1;
    }
    const int cob_l = tid % BN, kb_l = tid / BN;
    const int cv_l = (tid % (BN / 4)) * 4, kv_l = tid / (BN / 4);
    const bool vact = (NV >= 256) || (tid < NV);
    const bool co_ok = VECB ? (vact && (co0 + cv_l) < Cout) : (co0 + cob_l) < Cout;
    const float relu_floor = (flags & T2V_CONV_RELU_IN) ? 0.f : -__builtin_inff();
    // this thread's fixed byte offsets: strip body (channel ka_l), halo element (channel hk), weight element
    const uint32_t xoff = (xbase + (uint32_t)ka_l * (uint32_t)DHW) * 4u, xoff_h = (xbase_h + (uint32_t)hk * (uint32_t)DHW) * 4u;
    const uint32_t woff = co_ok ? (uint32_t)((VECB ? kv_l : kb_l) * Cout + co0 + (VECB ? cv_l : cob_l)) * 4u : 0u;

    f32x16 acc[NCO][NM];
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[LA], rah = 0.f;
    float rb[VECB ? 1 : LB];
    float4 rbv[VECB ? LBV : 1];
    const int ncb = Cin / BKT;
    const int nchunks = ntaps * ncb;
    const int cps = (nchunks + nsplit - 1) / nsplit;
    const int q0 = blockIdx.z * cps;
    int q1 = q0 + cps;
    if (q1 > nchunks) q1 = nchunks;

    // chunk q = (r * ncb + cb) * ndx + d
    int r_cur = q0 / (ncb * ndx);
    int cb_cur = (q0 - r_cur * ncb * ndx) / ndx;
    int d_cur = q0 - (r_cur * ncb + cb_cur) * ndx;
    auto advance = [&]() {
        if (++d_cur == ndx) { d_cur = 0; if (++cb_cur == ncb) { cb_cur = 0; ++r_cur; } }
    };
    bool pend_has_a = false, pend_av = false, pend_hv = false;
    int pend_dx = 0;
    // loads are unconditional from clamped addresses; masking happens when the registers go to LDS (see conv_igemm_kernel)
    auto load_chunk = [&](bool force_a) {
        const int c0 = cb_cur * BKT;
        pend_has_a = force_a || d_cur == 0;
        pend_dx = ndx == 3 ? d_cur - 1 : 0;
        if (pend_has_a) {
            const int roff = __builtin_amdgcn_readlane(tab_roff, r_cur);
            const int sx = c0 * DHW * 4;                       // (scalar) first channel of the block
            pend_av = (rowmask >> r_cur) & 1u;
            const uint32_t vo = xoff + (pend_av ? (uint32_t)roff : 0u);
#pragma unroll
            for (int j = 0; j < LA; ++j)
                ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo, sx + j * (KSA * 4) * DHW, 0));
            if (halo_thread) {
                pend_hv = (rowmask_h >> r_cur) & 1u;
                rah = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff_h + (pend_hv ? (uint32_t)roff : 0u), sx, 0));
            }
        }
        const int t = r_cur * ndx + d_cur;
        const int sw = (__builtin_amdgcn_readlane(tab_widx, t) * Cin + c0) * Cout * 4;     // (scalar) first row of the chunk's weights
        if (VECB) {
#pragma unroll
            for (int j = 0; j < LBV; ++j)
                rbv[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, sw + j * (KSBV * 4) * Cout, 0));
        } else {
#pragma unroll
            for (int j = 0; j < LB; ++j)
                rb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woff, sw + j * (KSB * 4) * Cout, 0));
        }
    };
    auto stage = [&](int ab, int bb) {
        if (pend_has_a) {
            float* as = As + ab * (BKT * AP);
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                as[(ka_l + j * KSA) * AP + 1 + ma_l] = pend_av ? fmaxf(ra[j], relu_floor) : 0.f;
            }
            if (halo_thread) {
                as[hk * AP + (he ? BM + 1 : 0)] = pend_hv ? fmaxf(rah, relu_floor) : 0.f;
            }
        }
        float* bs = Bs + bb * (BKT * BN);
        if (VECB) {
#pragma unroll
            for (int j = 0; j < LBV; ++j)
                if (vact) *reinterpret_cast<float4*>(&bs[(kv_l + j * KSBV) * BN + cv_l]) = co_ok ? rbv[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
#pragma unroll
            for (int j = 0; j < LB; ++j) bs[(kb_l + j * KSB) * BN + cob_l] = co_ok ? rb[j] : 0.f;
        }
    };

    int acur = 0, bcur = 0, dx_now = 0;
    if (q0 < q1) {
        load_chunk(true);
        stage(0, 0);
        dx_now = pend_dx;
        __syncthreads();
        if (q0 + 1 < q1) { advance(); load_chunk(false); }
    }
    for (int q = q0; q < q1; ++q) {
        const float* as = As + acur * (BKT * AP) + 1 + dx_now + wm * WM + l31;
        const float* bs = Bs + bcur * (BKT * BN) + wco * WCO + l31;
        bool keep[NM];
#pragma unroll
        for (int j = 0; j < NM; ++j) keep[j] = dx_now < 0 ? can_l[j] : (dx_now > 0 ? can_r[j] : true);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int k2 = 0; k2 < BKT / 2; ++k2) {
            float a[NCO], b[NM];
            const int krow = k2 * 2 + hi;
#pragma unroll
            for (int i = 0; i < NCO; ++i) a[i] = bs[krow * BN + i * 32];
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                const float v = as[krow * AP + j * 32];
                b[j] = keep[j] ? v : 0.f;
            }
#pragma unroll
            for (int i = 0; i < NCO; ++i)
#pragma unroll
                for (int j = 0; j < NM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        const bool more = q + 1 < q1;
        const bool next_a = more && pend_has_a;
        int dx_next = dx_now;
        if (more) { stage(acur ^ 1, bcur ^ 1); dx_next = pend_dx; }
        __syncthreads();
        if (q + 2 < q1) { advance(); load_chunk(false); }
        bcur ^= 1;
        if (next_a) acur ^= 1;
        dx_now = dx_next;
    }
    igemm_epilogue<NCO, NM, WCO, WM>(acc, tab, gi, gd, bias, slab, Cout, flags, nsplit, m0, co0, M, DHW, wm, wco, l31, hi);
}

// ------------------------------------------------------------------------------------------------
// STRIP3: the strip GEMM with ALL THREE dx taps of a (row tap, channel block) in one barrier round, for the 64-voxel and
// 128-voxel tiles. The per-dx strip kernel above runs only 16 MFMAs per wave between two barriers on those tiles, and the
// vector instructions around them (staging, operand reads, masks, waits: ~250 per round) set its pace: 45-80 TFLOP/s
// where the 256-voxel tile (64 MFMAs per round) reaches 100. Here a round = one staged strip + the weights of its three
// taps = 48 (96) MFMAs per wave against about the same instruction overhead. Single LDS stage, two barriers per round
// (the next round's gathers are issued right after the first and land during the MFMA loop), 4 waves as 2 (co) x 2 (m).
// Rounds = (row tap, channel block of 32); split-K runs over rounds. Members one voxel wide (ndx = 1) run their single tap.
// ------------------------------------------------------------------------------------------------
#ifndef DB_STAGE_LATE
#define DB_STAGE_LATE 1      // double-buffered form: the LDS writes of round q + 1 sit in the LAST slots of round q (0: right after the loads)
#endif
template <int BM, int BKT, int WAVES_CO, bool VECB, bool DB = false>
__global__ __launch_bounds__(256, ((BM == 256 || (DB && BKT == 32)) ? 2 : (DB ? 4 : 3))) void conv_igemm_strip3_kernel(const GroupTable tab, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, float* __restrict__ slab,
                                                                const int Cin, const int Cout, const int flags, const int nsplit) {
#ifdef T2V_STAMPS
    const unsigned long long t_real = __builtin_amdgcn_s_memrealtime();
#endif
    STAMP(t_entry);
    constexpr int BN = 64, WAVES_M = 4 / WAVES_CO;
    constexpr int WCO = BN / WAVES_CO, WM = BM / WAVES_M;
    constexpr int NCO = WCO / 32, NM = WM / 32;
    constexpr int AP = BM + 4;              // [left halo][BM voxels][right halo][pad]
    constexpr int LA = BKT * BM / 256;      // strip values per thread
    constexpr int KSA = 256 / BM;
    constexpr int NV = BKT * BN / 4;        // float4 per tap = 512 -> 2 per thread
    constexpr int LBV = NV / 256;
    constexpr int KSBV = 1024 / BN;         // k rows covered by one pass of 256 float4 loads
    constexpr int LB = BKT * BN / 256;      // scalar path: 8 per thread per tap
    constexpr int KSB = 256 / BN;
    static_assert(NCO >= 1 && NM >= 1 && LA >= 1 && LBV >= 1 && 2 * BKT <= 256, "tile");

    __shared__ __attribute__((aligned(16))) float As[(DB ? 2 : 1) * BKT * AP];
    __shared__ __attribute__((aligned(16))) float Bs[(DB ? 2 : 1) * 3 * BKT * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave % WAVES_CO, wm = wave / WAVES_CO;

    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && tile >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int D = gd.D, H = gd.H, W = gd.W;
    const int HW = H * W, DHW = D * HW;
    // frame-strided output (dstride = 2: the stem's conv2 feeds a pooling that keeps the even frames only): GEMM rows run over
    // the output voxels [N, Do, H, W]; row (n, do, r) gathers around input frame d = 2 do
    const int ds = gd.dstride == 2 ? 2 : 1;
    // strided OUTPUT (ydstride = 2: the data gradient of that pair): rows (n, e, r) read x frame e (+ dz) and are written to
    // frame 2e + yoff of y, rows run over the frames of that parity
    const int yds = gd.ydstride == 2 ? 2 : 1;
    const int Do = yds == 2 ? (gd.Dy + 1 - gd.yoff) / 2 : (ds == 2 ? (D + 1) / 2 : D);
    const int DHWo = Do * HW;
    const int M = gd.N * DHWo;
    const int m0 = (tile - tab.tile_start[gi]) * BM, co0 = blockIdx.y * BN;
    const int ntaps = gd.ntaps;
    const int ndx = gd.dx[0] < 0 ? 3 : 1;           // taps r*ndx + {0,1,2} = dx -1, 0, +1 of row tap r
    const int nrow = ntaps / ndx;

    const int lane_r = lane < nrow ? lane : 0, lane_t = lane < ntaps ? lane : 0;
    // (the mask loops below read these per-lane copies with v_readlane: indexing the kernel-argument tables with the loop counter
    // made hipcc issue a global_load_sbyte + s_waitcnt vmcnt(0) PER ITERATION — 9 x 2 dependent loads = most of a ~4 us prologue)
    const int tab_rdz = gd.dz[lane_r * ndx], tab_rdy = gd.dy[lane_r * ndx];
    const int tab_roff = (tab_rdz * HW + tab_rdy * W) * 4;
    const int tab_widx = gd.widx[lane_t];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)(gd.N * DHW) * (uint32_t)Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, -1, 0x00020000);

    // ---- staging coordinates (as in the per-dx strip kernel)
    const int ma_l = tid % BM, ka_l = tid / BM;
    const int he = tid / BKT, hk = tid % BKT;
    const bool halo_thread = tid < 2 * BKT;
    uint32_t rowmask = 0, rowmask_h = 0;
    uint32_t xbase = 0, xbase_h = 0;
    {
        const int m_a = m0 + ma_l;
        if (m_a < M) {
            const int n = m_a / DHWo, spo = m_a - n * DHWo;
            const int d_o = spo / HW, r = spo - d_o * HW;
            const int d = d_o * ds, sp = d * HW + r;
            const int h = r / W;
            xbase = (uint32_t)n * (uint32_t)Cin * (uint32_t)DHW + (uint32_t)sp;
            for (int t = 0; t < nrow; ++t) {
                const int dd = d + __builtin_amdgcn_readlane(tab_rdz, t), hh = h + __builtin_amdgcn_readlane(tab_rdy, t);
                if ((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H) rowmask |= 1u << t;
            }
        }
        const int m_h = he ? m0 + BM : m0 - 1;
        if (halo_thread && m_h >= 0 && m_h < M) {
            const int n = m_h / DHWo, spo = m_h - n * DHWo;
            const int d_o = spo / HW, r = spo - d_o * HW;
            const int d = d_o * ds, sp = d * HW + r;
            const int h = r / W;
            xbase_h = (uint32_t)n * (uint32_t)Cin * (uint32_t)DHW + (uint32_t)sp;
            for (int t = 0; t < nrow; ++t) {
                const int dd = d + __builtin_amdgcn_readlane(tab_rdz, t), hh = h + __builtin_amdgcn_readlane(tab_rdy, t);
                if ((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H) rowmask_h |= 1u << t;
            }
        }
    }
    bool can_l[NM], can_r[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int m = m0 + wm * WM + j * 32 + l31;
        const int w_ = m % W;
        can_l[j] = w_ > 0;
        can_r[j] = w_ < W - 1;
    }
    const int cob_l = tid % BN, kb_l = tid / BN;
    const int cv_l = (tid % (BN / 4)) * 4, kv_l = tid / (BN / 4);
    const bool co_ok = VECB ? (co0 + cv_l) < Cout : (co0 + cob_l) < Cout;
    const float relu_floor = (flags & T2V_CONV_RELU_IN) ? 0.f : -__builtin_inff();
    const uint32_t xoff = (xbase + (uint32_t)ka_l * (uint32_t)DHW) * 4u, xoff_h = (xbase_h + (uint32_t)hk * (uint32_t)DHW) * 4u;
    const uint32_t woff = co_ok ? (uint32_t)((VECB ? kv_l : kb_l) * Cout + co0 + (VECB ? cv_l : cob_l)) * 4u : 0u;

    f32x16 acc[NCO][NM];
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if constexpr (DB) {
        // ---- DOUBLE-BUFFERED form (64-voxel tiles, every member three taps wide): two LDS stages and two register sets, ONE barrier
        // per round, and the round's global loads (data of round q + 2) and LDS writes (data of round q + 1) sit BETWEEN the 48 MFMAs
        // of round q in program order — one item per MFMA — so that a wave's own matrix chain keeps issuing while its staging runs
        // (the single-stage form above runs stage -> barrier -> load issue -> MFMAs -> barrier and relies on the other workgroups
        // of the CU to cover a wave's staging; profiles/r04_strip3_stamps.txt: 3.1 k of a round's 11.6 k cycles are its MFMAs).
        // The body is branch-free: past the last round it re-loads / re-stages the last round's data into the stage nobody reads.
        static_assert(((BM == 64 && (BKT == 32 || BKT == 16)) || (BM == 256 && BKT == 16)) && VECB, "double-buffered form: 64 x 64 x {32, 16} and 256 x 64 x 16 tiles");
        constexpr int ASZ = BKT * AP, BSZ = 3 * BKT * BN;
        constexpr int NI = LA + 1 + 3 * LBV;         // loads (and LDS writes) per thread and round: 15 / 8 / 20
        constexpr int NSLOT = 3 * (BKT / 2);         // k-pair slots per round (NCO x NM MFMAs each): 48 / 24 / 24
        constexpr int IPS = (2 * NI + NSLOT - 1) / NSLOT;      // items per slot: 1 / 1 / 2
        constexpr int HS = (NI + IPS - 1) / IPS;     // slots that carry loads (and, at the other end of the round, LDS writes)
        const int ncb = Cin / BKT;
        const int nrounds = nrow * ncb;
        const int rps = (nrounds + nsplit - 1) / nsplit;
        const int q0 = blockIdx.z * rps;
        int q1 = q0 + rps;
        if (q1 > nrounds) q1 = nrounds;
        float ra2[2][LA], rah2[2] = {0.f, 0.f};
        float4 rb2[2][3 * LBV];
        bool pav[2] = {false, false}, phv[2] = {false, false};
        uint32_t l_vo[2] = {0u, 0u}, l_vh[2] = {0u, 0u};
        int l_sx[2] = {0, 0}, l_sw[2][3] = {{0, 0, 0}, {0, 0, 0}};
        const uint32_t xoff_hd = halo_thread ? xoff_h : 0u;          // (threads without a halo element load element 0 and store to the pad column)
        const int halo_col = halo_thread ? (he ? BM + 1 : 0) : BM + 2;
        const int halo_row = halo_thread ? hk : (tid % BKT);
        // addresses of round q's loads into set `st` (scalars + two vector offsets)
        auto prep = [&](int st, int q) {
            const int qq = q < q1 ? q : q1 - 1;
            const int r_cur = qq / ncb, cb = qq - r_cur * ncb;
            const int c0 = cb * BKT;
            const int roff = __builtin_amdgcn_readlane(tab_roff, r_cur);
            l_sx[st] = c0 * DHW * 4;
            pav[st] = (rowmask >> r_cur) & 1u;
            phv[st] = halo_thread && ((rowmask_h >> r_cur) & 1u);
            l_vo[st] = xoff + (pav[st] ? (uint32_t)roff : 0u);
            l_vh[st] = xoff_hd + (phv[st] ? (uint32_t)roff : 0u);
#pragma unroll
            for (int d = 0; d < 3; ++d) l_sw[st][d] = (__builtin_amdgcn_readlane(tab_widx, r_cur * 3 + d) * Cin + c0) * Cout * 4;
        };
        // item i of a round's 15 loads / 15 LDS writes (i is a compile-time constant at every call site)
        auto load_item = [&](int st, int i) {
            if (i < LA) ra2[st][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, l_vo[st], l_sx[st] + i * (KSA * 4) * DHW, 0));
            else if (i == LA) rah2[st] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, l_vh[st], l_sx[st], 0));
            else if (i < LA + 1 + 3 * LBV) {
                const int e = i - LA - 1, d = e / LBV, j = e - d * LBV;
                rb2[st][e] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, l_sw[st][d] + j * (KSBV * 4) * Cout, 0));
            }
        };
        auto stage_item = [&](int st, float* A, float* B, int i) {
            if (i < LA) A[(ka_l + i * KSA) * AP + 1 + ma_l] = pav[st] ? fmaxf(ra2[st][i], relu_floor) : 0.f;
            else if (i == LA) A[halo_row * AP + halo_col] = phv[st] ? fmaxf(rah2[st], relu_floor) : 0.f;
            else if (i < LA + 1 + 3 * LBV) {
                const int e = i - LA - 1, d = e / LBV, j = e - d * LBV;
                *reinterpret_cast<float4*>(&B[d * (BKT * BN) + (kv_l + j * KSBV) * BN + cv_l]) = co_ok ? rb2[st][e] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        const float* a_tap[3][NM];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                const bool keep = d == 0 ? can_l[j] : (d == 2 ? can_r[j] : true);
                a_tap[d][j] = keep ? As + d + wm * WM + j * 32 + l31 : As + (BM + 3);          // (1 + dx, dx = d - 1)
            }
        // one round: MFMAs on stage `cur`; loads of round q + 2 into register set `cur`, LDS writes of set `cur ^ 1` (round q + 1)
        // into stage `cur ^ 1`
        auto round = [&](auto CUR, int q) {
            constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
            prep(cur, q + 2);
            const float* Ab = As + cur * ASZ;
            const float* Bb = Bs + cur * BSZ + wco * WCO + l31;
            float* An = As + nxt * ASZ;
            float* Bn = Bs + nxt * BSZ;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
#pragma unroll
                for (int k2 = 0; k2 < BKT / 2; ++k2) {
                    const int krow = k2 * 2 + hi, slot = d * (BKT / 2) + k2;
                    float a[NCO], b[NM];
#pragma unroll
                    for (int i = 0; i < NCO; ++i) a[i] = Bb[d * (BKT * BN) + krow * BN + i * 32];
#pragma unroll
                    for (int j = 0; j < NM; ++j) b[j] = (a_tap[d][j] + cur * ASZ)[krow * AP];
#pragma unroll
                    for (int i = 0; i < NCO; ++i)
#pragma unroll
                        for (int j = 0; j < NM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                    constexpr int SS = DB_STAGE_LATE ? NSLOT - HS : HS;        // first slot of the LDS writes
#pragma unroll
                    for (int e = 0; e < IPS; ++e) {
                        if (slot < HS) load_item(cur, slot * IPS + e);
                        else if (slot >= SS && slot < SS + HS) stage_item(nxt, An, Bn, (slot - SS) * IPS + e);
                    }
                    // (hipcc otherwise sinks the loads to the end of the round and hoists the LDS writes to its start: the data
                    //  would be waited for right after it was requested)
                    if (slot == HS - 1 || slot == SS - 1 || slot == SS + HS - 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            (void)Ab;
            __syncthreads();
        };
        if (tid < BKT) { As[tid * AP + BM + 3] = 0.f; As[ASZ + tid * AP + BM + 3] = 0.f; }
        if (q0 < q1) {
            prep(0, q0);
#pragma unroll
            for (int i = 0; i < NI; ++i) load_item(0, i);
            prep(1, q0 + 1);
#pragma unroll
            for (int i = 0; i < NI; ++i) load_item(1, i);
#pragma unroll
            for (int i = 0; i < NI; ++i) stage_item(0, As, Bs, i);
            // (set 0 is free again: round q0 loads round q0 + 2 into it; set 1 = round q0 + 1 is staged during round q0)
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(1);
        for (int q = q0; q < q1; q += 2) {
            round(std::integral_constant<int, 0>{}, q);
            if (q + 1 < q1) round(std::integral_constant<int, 1>{}, q + 1);
        }
        __builtin_amdgcn_s_setprio(0);
        igemm_epilogue<NCO, NM, WCO, WM>(acc, tab, gi, gd, bias, slab, Cout, flags, nsplit, m0, co0, M, DHWo, wm, wco, l31, hi, yds, gd.yoff,
                                         gd.Dy * HW, HW);
        return;
    }
    float ra[LA], rah = 0.f;
    float rb[VECB ? 1 : 3 * LB];
    float4 rbv[VECB ? 3 * LBV : 1];
    const int ncb = Cin / BKT;
    const int nrounds = nrow * ncb;
    const int rps = (nrounds + nsplit - 1) / nsplit;
    const int q0 = blockIdx.z * rps;
    int q1 = q0 + rps;
    if (q1 > nrounds) q1 = nrounds;
    bool pend_av = false, pend_hv = false;

    auto load_round = [&](int q) {
        const int r_cur = q / ncb, cb = q - r_cur * ncb;
        const int c0 = cb * BKT;
        const int roff = __builtin_amdgcn_readlane(tab_roff, r_cur);
        const int sx = c0 * DHW * 4;
        pend_av = (rowmask >> r_cur) & 1u;
        const uint32_t vo = xoff + (pend_av ? (uint32_t)roff : 0u);
#pragma unroll
        for (int j = 0; j < LA; ++j)
            ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo, sx + j * (KSA * 4) * DHW, 0));
        if (halo_thread) {
            pend_hv = (rowmask_h >> r_cur) & 1u;
            rah = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xoff_h + (pend_hv ? (uint32_t)roff : 0u), sx, 0));
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d < ndx) {                                        // (uniform)
                const int sw = (__builtin_amdgcn_readlane(tab_widx, r_cur * ndx + d) * Cin + c0) * Cout * 4;
                if (VECB) {
#pragma unroll
                    for (int j = 0; j < LBV; ++j)
                        rbv[d * LBV + j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, sw + j * (KSBV * 4) * Cout, 0));
                } else {
#pragma unroll
                    for (int j = 0; j < LB; ++j)
                        rb[d * LB + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woff, sw + j * (KSB * 4) * Cout, 0));
                }
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < LA; ++j) As[(ka_l + j * KSA) * AP + 1 + ma_l] = pend_av ? fmaxf(ra[j], relu_floor) : 0.f;
        if (halo_thread) As[hk * AP + (he ? BM + 1 : 0)] = pend_hv ? fmaxf(rah, relu_floor) : 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d < ndx) {
                float* bs = Bs + d * (BKT * BN);
                if (VECB) {
#pragma unroll
                    for (int j = 0; j < LBV; ++j)
                        *reinterpret_cast<float4*>(&bs[(kv_l + j * KSBV) * BN + cv_l]) = co_ok ? rbv[d * LBV + j] : make_float4(0.f, 0.f, 0.f, 0.f);
                } else {
#pragma unroll
                    for (int j = 0; j < LB; ++j) bs[(kb_l + j * KSB) * BN + cob_l] = co_ok ? rb[d * LB + j] : 0.f;
                }
            }
        }
    };

    if (tid < BKT) As[tid * AP + BM + 3] = 0.f;          // the zero column (staging never writes past column BM + 1)
    if (q0 < q1) load_round(q0);
#ifdef T2V_STAMPS
    unsigned long long st_stage = 0, st_bar1 = 0, st_load = 0, st_mfma = 0, st_bar2 = 0;
#endif
#ifdef T2V_ABLATION      // developer ablations (wrong results): which phase of a round costs what (T2V_DEBUG_FLAGS)
    const bool ab_noload = flags & 64, ab_nostage = flags & 128, ab_nobar = flags & 256, ab_nomfma = flags & 512, ab_nostore = flags & 1024;
#else
    constexpr bool ab_noload = false, ab_nostage = false, ab_nobar = false, ab_nomfma = false, ab_nostore = false;
#endif
    STAMP(t_loop0);
    for (int q = q0; q < q1; ++q) {
        STAMP(ta);
        if (!ab_nostage || q == q0) stage();
        STAMP(tb);
        if (!ab_nobar || q == q0) __syncthreads();
        STAMP(tc);
        if (q + 1 < q1 && !ab_noload) load_round(q + 1);
        STAMP(td);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d < ndx && !ab_nomfma) {
                const int dx = ndx == 3 ? d - 1 : 0;
                const float* bs = Bs + d * (BKT * BN) + wco * WCO + l31;
                // a lane whose voxel sits at a row end reads this tap from the all-zero column instead of masking every
                // operand (one address select per tap instead of one v_cndmask per MFMA)
                const float* as[NM];
#pragma unroll
                for (int j = 0; j < NM; ++j) {
                    const bool keep = dx < 0 ? can_l[j] : (dx > 0 ? can_r[j] : true);
                    as[j] = keep ? As + 1 + dx + wm * WM + j * 32 + l31 : As + (BM + 3);
                }
#pragma unroll
                for (int k2 = 0; k2 < BKT / 2; ++k2) {
                    const int krow = k2 * 2 + hi;
                    float a[NCO], b[NM];
#pragma unroll
                    for (int i = 0; i < NCO; ++i) a[i] = bs[krow * BN + i * 32];
#pragma unroll
                    for (int j = 0; j < NM; ++j) b[j] = as[j][krow * AP];
#pragma unroll
                    for (int i = 0; i < NCO; ++i)
#pragma unroll
                        for (int j = 0; j < NM; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        STAMP(te);
        if (!ab_nobar) __syncthreads();
        STAMP(tf);
        STAMP_ACC(st_stage, ta, tb); STAMP_ACC(st_bar1, tb, tc); STAMP_ACC(st_load, tc, td); STAMP_ACC(st_mfma, td, te); STAMP_ACC(st_bar2, te, tf);
    }
    STAMP(t_loop1);
    if (ab_nostore && acc[0][0][0] != 12345.678f) return;
    igemm_epilogue<NCO, NM, WCO, WM>(acc, tab, gi, gd, bias, slab, Cout, flags, nsplit, m0, co0, M, DHWo, wm, wco, l31, hi, yds, gd.yoff,
                                     gd.Dy * HW, HW);
#ifdef T2V_STAMPS
    {
        STAMP(t_exit);
        const unsigned int rec = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave;
        if (g_stamps && rec < g_stamps_cap && lane == 0) {
            unsigned long long* o = g_stamps + (size_t)rec * 16;
            o[0] = t_entry; o[1] = t_loop0; o[2] = t_loop1; o[3] = t_exit;
            o[4] = st_stage; o[5] = st_bar1; o[6] = st_load; o[7] = st_mfma; o[8] = st_bar2;
            o[9] = (unsigned long long)(q1 - q0);
            o[10] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
            o[11] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
            o[12] = t_real;
            o[13] = (unsigned long long)gi;
            o[14] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// POOLED CONVOLUTION (forward): y = stride-2 conv3 of the box-summed activation r~ (pointwise.hip: pool_boxsum_k) — what
// `AvgPool(conv3(relu(r)))` equals, since box filter and convolution commute: 27 taps over the POOLED voxels instead of 27 taps
// over all of them. r~ sits on a padded grid [N, Cin, Dp, H+1, W+2] (padded index = position + 1; every tap of every pooled
// voxel reads real data: no bounds checks, no masks), so this is the strip3 GEMM with a mask-free gather: row m' = (n, e, i, j)
// of the GEMM reads, for the kernel row (dz, dy), the three values r~[2e+dz+1, 2i+dy+1, 2j + {0,1,2}] — one 8-byte and one 4-byte
// buffer load per (voxel, channel) — and stages them as three dx planes. One barrier round = 3 taps x 32 channels = 48 MFMAs per
// wave, exactly as in strip3. Members: t2v_conv_group with x = r~, y = the pooled output, (D, H, W) = the FULL-RESOLUTION extents
// and dstride = tmode (0: D == 1; 1 / 2: time strided with / without the box sum); taps in (dz, dy, dx) product order.
// ------------------------------------------------------------------------------------------------
template <int BM, bool DB = false>
__global__ __launch_bounds__(256, 3) void conv_pool_fwd_kernel(const GroupTable tab, const float* __restrict__ wp,
                                                               const float* __restrict__ bias, float* __restrict__ slab,
                                                               const int Cin, const int Cout, const int flags, const int nsplit) {
    constexpr bool VECB = true;             // (host: Cout % 4 == 0 — the pooled path takes channel counts that are multiples of 32)
    constexpr int BN = 64, BKT = DB ? 16 : 32, WAVES_CO = 2, WAVES_M = 2;
    constexpr int WCO = BN / WAVES_CO, WM = BM / WAVES_M;
    constexpr int NCO = WCO / 32, NM = WM / 32;
    constexpr int AP = BM + 4;
    constexpr int LA = BKT * BM / 256;
    constexpr int KSA = 256 / BM;
    constexpr int NV = BKT * BN / 4;
    constexpr int LBV = NV / 256;
    constexpr int KSBV = 1024 / BN;
    constexpr int LB = BKT * BN / 256;
    constexpr int KSB = 256 / BN;
    static_assert(NCO == 1 && NM >= 1 && LA >= 1 && LBV >= 1, "tile");

    __shared__ __attribute__((aligned(16))) float As[(DB ? 2 : 1) * 3 * BKT * AP];      // [dx][k][voxel]
    __shared__ __attribute__((aligned(16))) float Bs[(DB ? 2 : 1) * 3 * BKT * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave % WAVES_CO, wm = wave / WAVES_CO;

    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && tile >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int tm = gd.dstride;
    const int Dn = tm ? gd.D / 2 : 1, Hn = gd.H / 2, Wn = gd.W / 2;          // pooled extents
    const int Hp = gd.H + 1, Wp = gd.W + 2, Dp = tm ? gd.D + 1 : 1;
    const int Vp = Dp * Hp * Wp;
    const int HWn = Hn * Wn, DHWn = Dn * HWn;
    const int M = gd.N * DHWn;
    const int m0 = (tile - tab.tile_start[gi]) * BM, co0 = blockIdx.y * BN;
    const int ntaps = gd.ntaps;
    const int nrow = ntaps / 3;

    const int lane_r = lane < nrow ? lane : 0, lane_t = lane < ntaps ? lane : 0;
    const int tab_roff = ((tm ? gd.dz[lane_r * 3] + 1 : 0) * Hp + gd.dy[lane_r * 3] + 1) * Wp * 4;
    const int tab_widx = gd.widx[lane_t];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)(gd.N * Vp) * (uint32_t)Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, -1, 0x00020000);

    const int ma_l = tid % BM, ka_l = tid / BM;
    uint32_t xbase = 0;
    {
        const int m_a = m0 + ma_l;
        if (m_a < M) {                              // (rows past the end gather voxel 0: their columns are never stored)
            const int n = m_a / DHWn, sp = m_a - n * DHWn;
            const int e = sp / HWn, r = sp - e * HWn;
            const int i = r / Wn, j = r - i * Wn;
            xbase = (uint32_t)n * (uint32_t)Cin * (uint32_t)Vp + (uint32_t)(((tm ? 2 * e : 0) * Hp + 2 * i) * Wp + 2 * j);
        }
    }
    const int cob_l = tid % BN, kb_l = tid / BN;
    const int cv_l = (tid % (BN / 4)) * 4, kv_l = tid / (BN / 4);
    const bool co_ok = VECB ? (co0 + cv_l) < Cout : (co0 + cob_l) < Cout;
    const uint32_t xoff = (xbase + (uint32_t)ka_l * (uint32_t)Vp) * 4u;
    const uint32_t woff = co_ok ? (uint32_t)((VECB ? kv_l : kb_l) * Cout + co0 + (VECB ? cv_l : cob_l)) * 4u : 0u;

    f32x16 acc[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    if constexpr (DB) {
        // ---- double-buffered form (see conv_igemm_strip3_kernel): 16-channel rounds, two LDS stages and register sets, one barrier
        // per round; the loads of round q + 2 sit between the first MFMAs of round q, the LDS writes of round q + 1 between its last
        static_assert(BM == 64 && NM == 1 && LA == 4 && LBV == 1, "double-buffered form: 64 x 64 x 16 tiles");
        constexpr int ASZ = 3 * BKT * AP, BSZ = 3 * BKT * BN;
        constexpr int NSLOT = 3 * (BKT / 2);          // 24 MFMAs per round
        constexpr int NLD = 2 * LA + 3 * LBV;         // 11 loads: 4 x (8 + 4 bytes) of r~, 3 x 16 bytes of weights
        constexpr int NST = 3 * LA + 3 * LBV;         // 15 LDS writes
        const int ncb = Cin / BKT;
        const int nrounds = nrow * ncb;
        const int rps = (nrounds + nsplit - 1) / nsplit;
        const int q0 = blockIdx.z * rps;
        int q1 = q0 + rps;
        if (q1 > nrounds) q1 = nrounds;
        float2 ra2[2][LA];
        float ra1[2][LA];
        float4 rb2[2][3 * LBV];
        uint32_t l_vo[2] = {0u, 0u};
        int l_sx[2] = {0, 0}, l_sw[2][3] = {{0, 0, 0}, {0, 0, 0}};
        auto prep = [&](int st, int q) {
            const int qq = q < q1 ? q : q1 - 1;
            const int r_cur = qq / ncb, cb = qq - r_cur * ncb;
            const int c0 = cb * BKT;
            l_vo[st] = xoff + (uint32_t)__builtin_amdgcn_readlane(tab_roff, r_cur);
            l_sx[st] = c0 * Vp * 4;
#pragma unroll
            for (int d = 0; d < 3; ++d) l_sw[st][d] = (__builtin_amdgcn_readlane(tab_widx, r_cur * 3 + d) * Cin + c0) * Cout * 4;
        };
        auto load_item = [&](int st, int i) {
            if (i < LA) ra2[st][i] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rx, l_vo[st], l_sx[st] + i * (KSA * 4) * Vp, 0));
            else if (i < 2 * LA) ra1[st][i - LA] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, l_vo[st] + 8u, l_sx[st] + (i - LA) * (KSA * 4) * Vp, 0));
            else if (i < NLD) {
                const int d = i - 2 * LA;
                rb2[st][d] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, l_sw[st][d], 0));
            }
        };
        auto stage_item = [&](int st, float* A, float* B, int i) {
            if (i < 3 * LA) {
                const int j = i / 3, d = i - j * 3;
                A[d * (BKT * AP) + (ka_l + j * KSA) * AP + ma_l] = d == 0 ? ra2[st][j].x : (d == 1 ? ra2[st][j].y : ra1[st][j]);
            } else if (i < NST) {
                const int d = i - 3 * LA;
                *reinterpret_cast<float4*>(&B[d * (BKT * BN) + kv_l * BN + cv_l]) = co_ok ? rb2[st][d] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto round = [&](auto CUR, int q) {
            constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
            prep(cur, q + 2);
            const float* Ab = As + cur * ASZ + wm * WM + l31;
            const float* Bb = Bs + cur * BSZ + wco * WCO + l31;
            float* An = As + nxt * ASZ;
            float* Bn = Bs + nxt * BSZ;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
#pragma unroll
                for (int k2 = 0; k2 < BKT / 2; ++k2) {
                    const int krow = k2 * 2 + hi, slot = d * (BKT / 2) + k2;
                    const float a = Bb[d * (BKT * BN) + krow * BN];
                    const float b = Ab[d * (BKT * AP) + krow * AP];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                    if (2 * slot < NLD) load_item(cur, 2 * slot);
                    if (2 * slot + 1 < NLD) load_item(cur, 2 * slot + 1);
                    if (slot >= NSLOT - NST) stage_item(nxt, An, Bn, slot - (NSLOT - NST));
                    if (slot == (NLD + 1) / 2 - 1 || slot == NSLOT - NST - 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
        };
        if (q0 < q1) {
            prep(0, q0);
#pragma unroll
            for (int i = 0; i < NLD; ++i) load_item(0, i);
            prep(1, q0 + 1);
#pragma unroll
            for (int i = 0; i < NLD; ++i) load_item(1, i);
#pragma unroll
            for (int i = 0; i < NST; ++i) stage_item(0, As, Bs, i);
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(1);
        for (int q = q0; q < q1; q += 2) {
            round(std::integral_constant<int, 0>{}, q);
            if (q + 1 < q1) round(std::integral_constant<int, 1>{}, q + 1);
        }
        __builtin_amdgcn_s_setprio(0);
        f32x16 (&accd)[1][NM] = reinterpret_cast<f32x16 (&)[1][NM]>(acc);
        igemm_epilogue<1, NM, WCO, WM>(accd, tab, gi, gd, bias, slab, Cout, flags, nsplit, m0, co0, M, DHWn, wm, wco, l31, hi);
        return;
    }
    float2 ra2[LA];
    float ra1[LA];
    float rb[VECB ? 1 : 3 * LB];
    float4 rbv[VECB ? 3 * LBV : 1];
    const int ncb = Cin / BKT;
    const int nrounds = nrow * ncb;
    const int rps = (nrounds + nsplit - 1) / nsplit;
    const int q0 = blockIdx.z * rps;
    int q1 = q0 + rps;
    if (q1 > nrounds) q1 = nrounds;

    auto load_round = [&](int q) {
        const int r_cur = q / ncb, cb = q - r_cur * ncb;
        const int c0 = cb * BKT;
        const uint32_t vo = xoff + (uint32_t)__builtin_amdgcn_readlane(tab_roff, r_cur);
        const int sx = c0 * Vp * 4;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            ra2[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rx, vo, sx + j * (KSA * 4) * Vp, 0));
            ra1[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo + 8u, sx + j * (KSA * 4) * Vp, 0));
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int sw = (__builtin_amdgcn_readlane(tab_widx, r_cur * 3 + d) * Cin + c0) * Cout * 4;
            if (VECB) {
#pragma unroll
                for (int j = 0; j < LBV; ++j)
                    rbv[d * LBV + j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, sw + j * (KSBV * 4) * Cout, 0));
            } else {
#pragma unroll
                for (int j = 0; j < LB; ++j)
                    rb[d * LB + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woff, sw + j * (KSB * 4) * Cout, 0));
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            float* a = As + (ka_l + j * KSA) * AP + ma_l;
            a[0] = ra2[j].x;
            a[BKT * AP] = ra2[j].y;
            a[2 * BKT * AP] = ra1[j];
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float* bs = Bs + d * (BKT * BN);
            if (VECB) {
#pragma unroll
                for (int j = 0; j < LBV; ++j)
                    *reinterpret_cast<float4*>(&bs[(kv_l + j * KSBV) * BN + cv_l]) = co_ok ? rbv[d * LBV + j] : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < LB; ++j) bs[(kb_l + j * KSB) * BN + cob_l] = co_ok ? rb[d * LB + j] : 0.f;
            }
        }
    };

    if (q0 < q1) load_round(q0);
    for (int q = q0; q < q1; ++q) {
        stage();
        __syncthreads();
        if (q + 1 < q1) load_round(q + 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float* bs = Bs + d * (BKT * BN) + wco * WCO + l31;
            const float* as = As + d * (BKT * AP) + wm * WM + l31;
#pragma unroll
            for (int k2 = 0; k2 < BKT / 2; ++k2) {
                const int krow = k2 * 2 + hi;
                const float a = bs[krow * BN];
                float b[NM];
#pragma unroll
                for (int j = 0; j < NM; ++j) b[j] = as[krow * AP + j * 32];
#pragma unroll
                for (int j = 0; j < NM; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc[j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
    f32x16 (&acc2)[1][NM] = reinterpret_cast<f32x16 (&)[1][NM]>(acc);
    igemm_epilogue<1, NM, WCO, WM>(acc2, tab, gi, gd, bias, slab, Cout, flags, nsplit, m0, co0, M, DHWn, wm, wco, l31, hi);
}

// ------------------------------------------------------------------------------------------------
// POOLED CONVOLUTION (data gradient): g~ = conv^T(subsample^T(dL/dy)) on the padded r~ grid, WITHOUT the zero-stuffed
// intermediate. A padded index p = 2a + c receives, per axis, from the taps d = -1 (source a) and d = +1 (source a - 1) when
// its parity c is 0, and from the tap d = 0 (source a) when c is 1: every forward tap feeds exactly one of the 8 parity classes,
// so the 27 taps are spent once over the POOLED grid. Each class is written as a dense plane over the grid (a_t, a, b) of extents
// (D/2+1, H/2+1, W/2+1) (pointwise.hip: pool_unbox_k sums the planes back to full resolution and applies the ReLU mask).
// A workgroup owns a tile of 64 grid voxels x 64 channels of ONE (ct, cy) class pair (blockIdx.y & 3): its kernel rows are the
// (dz, dy) of that pair, and the three dx taps of a row share one staged strip of dL/dy exactly as in strip3 — dx = -1 and +1
// accumulate into the cx = 0 plane (reading the strip at b and b - 1), dx = 0 into the cx = 1 plane.
// Members: x = dL/dy [N, K, D', H', W'], y = the 8 planes, (D, H, W) full-resolution extents, dstride = tmode;
// widx[f] = packed slot (mode 1: [K][C]) holding forward tap f's matrix, f = ((dz+1)*3 + dy+1)*3 + dx+1.
// ------------------------------------------------------------------------------------------------
template <bool DB>
__global__ __launch_bounds__(256, 3) void conv_pool_dgrad_kernel(const GroupTable tab, const float* __restrict__ wp,
                                                                 const int K, const int C) {
    constexpr bool VECB = true;             // (host: C % 4 == 0)
    constexpr int BM = 64, BN = 64, BKT = DB ? 16 : 32, WAVES_CO = 2;
    constexpr int WCO = 32, WM = 32;
    constexpr int AP = BM + 4;              // [left halo][BM voxels][pad][zero column]
    constexpr int LA = BKT * BM / 256;
    constexpr int KSA = 256 / BM;
    constexpr int LBV = BKT * BN / 4 / 256;
    constexpr int KSBV = 1024 / BN;
    constexpr int LB = BKT * BN / 256;
    constexpr int KSB = 256 / BN;

    __shared__ __attribute__((aligned(16))) float As[(DB ? 2 : 1) * BKT * AP];
    __shared__ __attribute__((aligned(16))) float Bs[(DB ? 2 : 1) * 3 * BKT * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave % WAVES_CO, wm = wave / WAVES_CO;

    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && tile >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int tm = gd.dstride;
    // blockIdx.y = pair * (channel tiles) + channel tile: the class pair (0, 0) runs 4 kernel rows, (0, 1) and (1, 0) two, (1, 1) one —
    // the dispatcher hands out workgroups in blockIdx order, so the long ones start first and the short ones fill the tail
    const int nct = (int)gridDim.y >> 2;
    const int pair = (int)blockIdx.y / nct, ct = pair >> 1, cy = pair & 1;
    if (tm == 0 && ct) return;                                   // (uniform) members without a time axis have one time class
    const int co0 = ((int)blockIdx.y - pair * nct) * BN;
    const int Dn = tm ? gd.D / 2 : 1, Hn = gd.H / 2, Wn = gd.W / 2;           // extents of dL/dy
    const int Dq = tm ? Dn + 1 : 1, Hq = Hn + 1, Wq = Wn + 1;                 // the padded-grid planes
    const int HWn = Hn * Wn, Vn = Dn * HWn;
    const int HWq = Hq * Wq, Vq = Dq * HWq;
    // rows run over THIS class pair's part of the padded grid: an odd-parity class only lives on the first D' / H' positions of its
    // axis (its padded index 2a + 1 must stay inside the box-summed tensor), so its rows are (De, He, Wq) with De = D' / He = H' —
    // the unbox pass never reads the rest. The tile table counts tiles of the full padded grid: the surplus workgroups leave at once.
    const int De = (tm && ct) ? Dn : Dq, He = cy ? Hn : Hq;
    const int HWe = He * Wq, Ve = De * HWe;
    const int M = gd.N * Ve;
    const int m0 = (tile - tab.tile_start[gi]) * BM;
    if (m0 >= M) return;                                         // (uniform)
    // kernel rows of this class pair: dz in {-1,+1} (ct = 0) or {0} (ct = 1; also tmode 0), dy likewise
    const int ndz = (tm && ct == 0) ? 2 : 1, ndy = cy == 0 ? 2 : 1;
    const int nrow = ndz * ndy;
    const int lane_t = lane < 27 ? lane : 0;
    const int tab_widx = gd.widx[lane_t];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)(gd.N * Vn) * (uint32_t)K * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, -1, 0x00020000);

    // ---- my staged voxel (tid % BM) and, for the first threads, the tile's left halo
    const int ma_l = tid % BM, ka_l = tid / BM;
    const int hk = tid % BKT;
    const bool halo_thread = tid < BKT;
    int s_at = 0, s_a = 0, s_b = Wn;          // (b = Wn: the padded column, never a valid source)
    uint32_t s_n = 0;
    int h_at = 0, h_a = 0, h_b = Wn;
    uint32_t h_n = 0;
    {
        const int m_a = m0 + ma_l;
        if (m_a < M) {
            const int n = m_a / Ve, sp = m_a - n * Ve;
            s_at = sp / HWe;
            const int r = sp - s_at * HWe;
            s_a = r / Wq;
            s_b = r - s_a * Wq;
            s_n = (uint32_t)n;
        }
        if (m0 < M) {                            // left neighbour of the tile's first voxel: same row, b - 1 (only read when b >= 1)
            const int n = m0 / Ve, sp = m0 - n * Ve;
            h_at = sp / HWe;
            const int r = sp - h_at * HWe;
            h_a = r / Wq;
            h_b = r - h_a * Wq - 1;
            h_n = (uint32_t)n;
            if (h_b < 0) h_b = Wn;               // (row start: no left neighbour)
        }
    }
    bool can_l[1];
    {
        const int m = m0 + wm * WM + l31;
        can_l[0] = (m % Wq) >= 1;
    }
    const int cob_l = tid % BN, kb_l = tid / BN;
    const int cv_l = (tid % (BN / 4)) * 4, kv_l = tid / (BN / 4);
    const bool co_ok = VECB ? (co0 + cv_l) < C : (co0 + cob_l) < C;
    const uint32_t woff = co_ok ? (uint32_t)((VECB ? kv_l : kb_l) * C + co0 + (VECB ? cv_l : cob_l)) * 4u : 0u;

    f32x16 acc0, acc1;                       // cx = 0 (dx = -1, +1) and cx = 1 (dx = 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    float ra[LA], rah = 0.f;
    float rb[VECB ? 1 : 3 * LB];
    float4 rbv[VECB ? 3 * LBV : 1];
    const int ncb = K / BKT;
    // k-split (gridDim.z): split z takes the channel blocks [cb0, cb1) of EVERY kernel row and writes its own set of planes;
    // pool_unbox_k adds the sets up (launches of a few dozen tiles with K of several hundred are otherwise one workgroup per CU
    // walking its rounds at memory latency)
    const int cps = (ncb + (int)gridDim.z - 1) / (int)gridDim.z;
    const int cb0 = (int)blockIdx.z * cps;
    const int cbn = (cb0 + cps <= ncb ? cps : ncb - cb0);
    // kernel rows whose source voxels are out of range for EVERY voxel of this tile (a tile inside the first / last plane of the
    // padded grid, the rows of a member with a single pooled frame, ...) are skipped whole: 4 workgroup-wide votes up front
    uint32_t rows_live = 0;
    for (int r = 0; r < nrow; ++r) {
        const int rz = r / ndy, ry = r - rz * ndy;
        const int dz = (tm && ct == 0) ? 2 * rz - 1 : 0, dy = cy == 0 ? 2 * ry - 1 : 0;
        const int at = s_at + (dz > 0 ? -1 : 0), a = s_a + (dy > 0 ? -1 : 0);
        const int hat = h_at + (dz > 0 ? -1 : 0), ha = h_a + (dy > 0 ? -1 : 0);
        const bool v = ((unsigned)at < (unsigned)Dn && (unsigned)a < (unsigned)Hn && s_b < Wn) ||
                       (halo_thread && (unsigned)hat < (unsigned)Dn && (unsigned)ha < (unsigned)Hn && h_b < Wn);
        if (__syncthreads_or(v ? 1 : 0)) rows_live |= 1u << r;
    }
    uint32_t row_of = 0;                     // live kernel rows, one per byte
    int nlive_rows = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r < nrow && ((rows_live >> r) & 1u)) { row_of |= (uint32_t)r << (8 * nlive_rows); ++nlive_rows; }
    const int nrounds = cbn > 0 ? nlive_rows * cbn : 0;
    if constexpr (DB) {
        // ---- double-buffered form (see conv_igemm_strip3_kernel): 16-channel rounds, one barrier per round; a round = 8 k-pairs of
        // three MFMAs; the loads of round q + 2 sit behind the first four, the LDS writes of round q + 1 behind the last four
        static_assert(LA == 4 && LBV == 1, "double-buffered form: 64 x 64 x 16 tiles");
        constexpr int ASZ = BKT * AP, BSZ = 3 * BKT * BN;
        constexpr int NI = LA + 1 + 3;                // loads / LDS writes per thread and round
        float ra2[2][LA], rah2[2] = {0.f, 0.f};
        float4 rb2[2][3];
        bool pav[2] = {false, false}, phv[2] = {false, false};
        uint32_t l_vo[2] = {0u, 0u}, l_vh[2] = {0u, 0u};
        int l_sx[2] = {0, 0}, l_sw[2][3] = {{0, 0, 0}, {0, 0, 0}};
        const int halo_col = halo_thread ? 0 : BM + 2;         // (threads without a halo element store a zero into the pad column)
        auto prep = [&](int st, int q) {
            const int qq = q < nrounds ? q : nrounds - 1;
            const int r_i = qq / cbn, cb = cb0 + (qq - r_i * cbn);
            const int r_cur = (int)((row_of >> (8 * r_i)) & 0xffu);
            const int c0 = cb * BKT;
            const int rz = r_cur / ndy, ry = r_cur - rz * ndy;
            const int dz = (tm && ct == 0) ? 2 * rz - 1 : 0, dy = cy == 0 ? 2 * ry - 1 : 0;
            const int st_ = dz > 0 ? -1 : 0, sy = dy > 0 ? -1 : 0;
            l_sx[st] = c0 * Vn * 4;
            {
                const int at = s_at + st_, a = s_a + sy;
                pav[st] = (unsigned)at < (unsigned)Dn && (unsigned)a < (unsigned)Hn && s_b < Wn;
                l_vo[st] = pav[st] ? ((s_n * (uint32_t)K + (uint32_t)ka_l) * (uint32_t)Vn + (uint32_t)((at * Hn + a) * Wn + s_b)) * 4u : 0u;
            }
            {
                const int at = h_at + st_, a = h_a + sy;
                phv[st] = halo_thread && (unsigned)at < (unsigned)Dn && (unsigned)a < (unsigned)Hn && h_b < Wn;
                l_vh[st] = phv[st] ? ((h_n * (uint32_t)K + (uint32_t)hk) * (uint32_t)Vn + (uint32_t)((at * Hn + a) * Wn + h_b)) * 4u : 0u;
            }
            const int f0 = ((dz + 1) * 3 + dy + 1) * 3;
#pragma unroll
            for (int d = 0; d < 3; ++d) l_sw[st][d] = (__builtin_amdgcn_readlane(tab_widx, f0 + d) * K + c0) * C * 4;
        };
        auto load_item = [&](int st, int i) {
            if (i < LA) ra2[st][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, l_vo[st], l_sx[st] + i * (KSA * 4) * Vn, 0));
            else if (i == LA) rah2[st] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, l_vh[st], l_sx[st], 0));
            else if (i < NI) rb2[st][i - LA - 1] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, l_sw[st][i - LA - 1], 0));
        };
        auto stage_item = [&](int st, float* A, float* B, int i) {
            if (i < LA) A[(ka_l + i * KSA) * AP + 1 + ma_l] = pav[st] ? ra2[st][i] : 0.f;
            else if (i == LA) A[(tid % BKT) * AP + halo_col] = phv[st] ? rah2[st] : 0.f;
            else if (i < NI) *reinterpret_cast<float4*>(&B[(i - LA - 1) * (BKT * BN) + kv_l * BN + cv_l]) = co_ok ? rb2[st][i - LA - 1] : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto round = [&](auto CUR, int q) {
            constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
            prep(cur, q + 2);
            const float* a_own = As + cur * ASZ + 1 + wm * WM + l31;
            const float* a_left = can_l[0] ? As + cur * ASZ + wm * WM + l31 : As + cur * ASZ + (BM + 3);
            const float* bb = Bs + cur * BSZ + wco * WCO + l31;
            float* An = As + nxt * ASZ;
            float* Bn = Bs + nxt * BSZ;
#pragma unroll
            for (int k2 = 0; k2 < BKT / 2; ++k2) {
                const int krow = k2 * 2 + hi;
                const float xo = a_own[krow * AP], xl = a_left[krow * AP];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[krow * BN], xo, acc0, 0, 0, 0);                      // dx = -1: source b
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[BKT * BN + krow * BN], xo, acc1, 0, 0, 0);           // dx =  0: source b
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[2 * BKT * BN + krow * BN], xl, acc0, 0, 0, 0);       // dx = +1: source b - 1
                if (k2 < 4) { load_item(cur, 2 * k2); load_item(cur, 2 * k2 + 1); }
                else { stage_item(nxt, An, Bn, 2 * (k2 - 4)); stage_item(nxt, An, Bn, 2 * (k2 - 4) + 1); }
                if (k2 == 3) __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        if (tid < BKT) { As[tid * AP + BM + 3] = 0.f; As[ASZ + tid * AP + BM + 3] = 0.f; }
        if (nrounds > 0) {
            prep(0, 0);
#pragma unroll
            for (int i = 0; i < NI; ++i) load_item(0, i);
            prep(1, 1);
#pragma unroll
            for (int i = 0; i < NI; ++i) load_item(1, i);
#pragma unroll
            for (int i = 0; i < NI; ++i) stage_item(0, As, Bs, i);
        }
        __syncthreads();
        __builtin_amdgcn_s_setprio(1);
        for (int q = 0; q < nrounds; q += 2) {
            round(std::integral_constant<int, 0>{}, q);
            if (q + 1 < nrounds) round(std::integral_constant<int, 1>{}, q + 1);
        }
        __builtin_amdgcn_s_setprio(0);
    } else {
    bool pend_av = false, pend_hv = false;

    auto load_round = [&](int q) {
        const int r_i = q / cbn, cb = cb0 + (q - r_i * cbn);
        const int r_cur = (int)((row_of >> (8 * r_i)) & 0xffu);
        const int c0 = cb * BKT;
        const int rz = r_cur / ndy, ry = r_cur - rz * ndy;
        const int dz = (tm && ct == 0) ? 2 * rz - 1 : 0, dy = cy == 0 ? 2 * ry - 1 : 0;
        const int st = dz > 0 ? -1 : 0, sy = dy > 0 ? -1 : 0;          // source offsets on the dL/dy grid
        const int sx = c0 * Vn * 4;
        {
            const int at = s_at + st, a = s_a + sy;
            pend_av = (unsigned)at < (unsigned)Dn && (unsigned)a < (unsigned)Hn && s_b < Wn;
            const uint32_t vo = pend_av ? ((s_n * (uint32_t)K + (uint32_t)ka_l) * (uint32_t)Vn + (uint32_t)((at * Hn + a) * Wn + s_b)) * 4u : 0u;
#pragma unroll
            for (int j = 0; j < LA; ++j)
                ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo, sx + j * (KSA * 4) * Vn, 0));
        }
        if (halo_thread) {
            const int at = h_at + st, a = h_a + sy;
            pend_hv = (unsigned)at < (unsigned)Dn && (unsigned)a < (unsigned)Hn && h_b < Wn;
            const uint32_t vo = pend_hv ? ((h_n * (uint32_t)K + (uint32_t)hk) * (uint32_t)Vn + (uint32_t)((at * Hn + a) * Wn + h_b)) * 4u : 0u;
            rah = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo, sx, 0));
        }
        const int f0 = ((dz + 1) * 3 + dy + 1) * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int sw = (__builtin_amdgcn_readlane(tab_widx, f0 + d) * K + c0) * C * 4;
            if (VECB) {
#pragma unroll
                for (int j = 0; j < LBV; ++j)
                    rbv[d * LBV + j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, sw + j * (KSBV * 4) * C, 0));
            } else {
#pragma unroll
                for (int j = 0; j < LB; ++j)
                    rb[d * LB + j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, woff, sw + j * (KSB * 4) * C, 0));
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < LA; ++j) As[(ka_l + j * KSA) * AP + 1 + ma_l] = pend_av ? ra[j] : 0.f;
        if (halo_thread) As[hk * AP] = pend_hv ? rah : 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float* bs = Bs + d * (BKT * BN);
            if (VECB) {
#pragma unroll
                for (int j = 0; j < LBV; ++j)
                    *reinterpret_cast<float4*>(&bs[(kv_l + j * KSBV) * BN + cv_l]) = co_ok ? rbv[d * LBV + j] : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int j = 0; j < LB; ++j) bs[(kb_l + j * KSB) * BN + cob_l] = co_ok ? rb[d * LB + j] : 0.f;
            }
        }
    };

    if (tid < BKT) As[tid * AP + BM + 3] = 0.f;          // the zero column
    if (nrounds > 0) load_round(0);
    for (int q = 0; q < nrounds; ++q) {
        stage();
        __syncthreads();
        if (q + 1 < nrounds) load_round(q + 1);
        __builtin_amdgcn_s_setprio(1);
        {
            const float* a_own = As + 1 + wm * WM + l31;
            const float* a_left = can_l[0] ? As + wm * WM + l31 : As + (BM + 3);
            const float* b0 = Bs + 0 * (BKT * BN) + wco * WCO + l31;
            const float* b1 = Bs + 1 * (BKT * BN) + wco * WCO + l31;
            const float* b2 = Bs + 2 * (BKT * BN) + wco * WCO + l31;
#pragma unroll
            for (int k2 = 0; k2 < BKT / 2; ++k2) {
                const int krow = k2 * 2 + hi;
                const float xo = a_own[krow * AP], xl = a_left[krow * AP];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[krow * BN], xo, acc0, 0, 0, 0);      // dx = -1: source b
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[krow * BN], xo, acc1, 0, 0, 0);      // dx =  0: source b
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b2[krow * BN], xl, acc0, 0, 0, 0);      // dx = +1: source b - 1
            }
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
    }
    // ---- epilogue: rows (registers) = channel, columns (lanes) = grid voxel; two class planes
    // Entries nobody reads are not written: an odd-parity class only lives on the first D' / H' / W' grid positions of its axis (its
    // padded index 2a + 1 must stay inside the box-summed tensor), the unbox pass never touches the rest — about a quarter of the
    // plane bytes, in a kernel whose 32 KB of output per 2-8 barrier rounds make it store-bound.
    const int m = m0 + wm * WM + l31;
    if (m < M) {
        const int n = m / Ve, spe = m - n * Ve;
        const int at = spe / HWe, r2 = spe - at * HWe;
        const int a = r2 / Wq, b = r2 - a * Wq;
        const int sp = (at * Hq + a) * Wq + b;                   // position in the PADDED plane (what pool_unbox_k indexes)
        constexpr bool row_ok = true;                            // (every row of the class grid is read)
        const bool ok1 = b < Wn;
        const size_t plane = (size_t)gd.N * C * Vq;
        float* p0 = gd.y + ((size_t)blockIdx.z * 8 + (size_t)((ct * 2 + cy) * 2 + 0)) * plane + (size_t)n * C * Vq + sp;
        float* p1 = p0 + plane;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * WCO + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < C) {
                if (row_ok) p0[(size_t)co * Vq] = acc0[r];
                if (ok1) p1[(size_t)co * Vq] = acc1[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16-compute variant of the implicit GEMM ("bf16 compute / fp32 master", BASELINE configs 2-4): activations and
// results stay fp32 in HBM; the tiles are rounded to bf16 on their way into LDS and multiplied by
// v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate) with fp32 accumulation. Weights come pre-packed as bf16
// wpb[slot][Cout][Cin] (K contiguous). LDS tiles are [row][k] with k contiguous (one ds_read_b128 = the 8 k of a lane's
// fragment); same workgroup tile, grouping, split-K and epilogue as conv_igemm_kernel. Cin % 32 == 0 only.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define B16_KP 40      // LDS row pitch in bf16: 32 k + 8 pad (80 B: 16-byte aligned rows, 2-way at worst on b128 reads)

template <int BM>
__global__ __launch_bounds__(256) void conv_igemm_bf16_kernel(const GroupTable tab, const __bf16* __restrict__ wpb,
                                                              const float* __restrict__ bias, float* __restrict__ slab,
                                                              const int Cin, const int Cout, const int flags, const int nsplit) {
    constexpr int BN = 64, BKT = 32, WAVES_CO = 2;
    constexpr int WAVES_M = 2;
    constexpr int WCO = BN / WAVES_CO, WM = BM / WAVES_M;
    constexpr int NCO = WCO / 32, NM = WM / 32;
    constexpr int KPT = BKT * BM / 256;          // k values per thread of the activation tile (16 for BM = 128, 8 for 64)
    static_assert(NCO == 1 && NM >= 1 && (KPT == 16 || KPT == 8), "tile");       // BKT / KPT threads share a voxel
    __shared__ __attribute__((aligned(16))) __bf16 Xs[2 * BM * B16_KP];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[2 * BN * B16_KP];
    __shared__ int s_off[T2V_MAX_TAPS];
    __shared__ int s_widx[T2V_MAX_TAPS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave % WAVES_CO, wm = wave / WAVES_CO;
    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && tile >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const float* __restrict__ x = gd.x;
    const int D = gd.D, H = gd.H, W = gd.W;
    const int HW = H * W, DHW = D * HW;
    const int M = gd.N * DHW;
    const int m0 = (tile - tab.tile_start[gi]) * BM, co0 = blockIdx.y * BN;
    const int ntaps = gd.ntaps;
    const int lane_t = lane < ntaps ? lane : 0;      // (per-lane table copies + v_readlane in the mask loop: see conv_igemm_strip3_kernel)
    const int tab_tdz = gd.dz[lane_t], tab_tdy = gd.dy[lane_t], tab_tdx = gd.dx[lane_t];
    if (tid < ntaps) {
        s_off[tid] = tab_tdz * HW + tab_tdy * W + tab_tdx;
        s_widx[tid] = gd.widx[tid];
    }
    // activation staging: thread -> (voxel, group of KPT consecutive channels)
    const int ma_l = tid % BM, kq = tid / BM;
    const int m_a = m0 + ma_l;
    uint32_t tapmask = 0;
    size_t xbase = 0;
    if (m_a < M) {
        const int n = m_a / DHW, sp = m_a - n * DHW;
        const int d = sp / HW, r = sp - d * HW;
        const int h = r / W, w_ = r - h * W;
        xbase = (size_t)n * Cin * DHW + sp;
        for (int t = 0; t < ntaps; ++t) {
            const int dd = d + __builtin_amdgcn_readlane(tab_tdz, t), hh = h + __builtin_amdgcn_readlane(tab_tdy, t),
                      ww = w_ + __builtin_amdgcn_readlane(tab_tdx, t);
            if ((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) tapmask |= 1u << t;
        }
    }
    // weight staging: thread -> (output channel, 8 consecutive k)
    const int wc_l = tid >> 2, wk_l = (tid & 3) * 8;
    const bool w_ok = co0 + wc_l < Cout;
    const bool relu_in = flags & T2V_CONV_RELU_IN;

    f32x16 acc[NCO][NM];
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    float ra[KPT];
    bf16x8 rw;
    bool pend_v = false;
    const int cpt = Cin / BKT;
    const int nchunks = ntaps * cpt;
    const int cps = (nchunks + nsplit - 1) / nsplit;
    const int q0 = blockIdx.z * cps;
    int q1 = q0 + cps;
    if (q1 > nchunks) q1 = nchunks;
    __syncthreads();

    int t_cur = q0 / cpt, c_cur = (q0 - t_cur * cpt) * BKT;
    auto advance = [&]() {
        c_cur += BKT;
        if (c_cur >= Cin) { c_cur = 0; ++t_cur; }
    };
    auto load_chunk = [&]() {
        pend_v = (tapmask >> t_cur) & 1u;
        const float* px = x + xbase + (pend_v ? (ptrdiff_t)s_off[t_cur] : 0) + (size_t)(c_cur + kq * KPT) * DHW;
#pragma unroll
        for (int j = 0; j < KPT; ++j) ra[j] = px[(size_t)j * DHW];
        const __bf16* pw = wpb + ((size_t)s_widx[t_cur] * Cout + (w_ok ? co0 + wc_l : 0)) * Cin + c_cur + wk_l;
        rw = *reinterpret_cast<const bf16x8*>(pw);
    };
    auto stage = [&](int b) {
        __bf16* xs = Xs + b * (BM * B16_KP) + ma_l * B16_KP + kq * KPT;
#pragma unroll
        for (int g = 0; g < KPT / 8; ++g) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float f = pend_v ? ra[g * 8 + j] : 0.f;
                if (relu_in) f = fmaxf(f, 0.f);
                v[j] = (__bf16)f;
            }
            *reinterpret_cast<bf16x8*>(xs + g * 8) = v;
        }
        bf16x8 wv = rw;
        if (!w_ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = (__bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(Ws + b * (BN * B16_KP) + wc_l * B16_KP + wk_l) = wv;
    };

    int cur = 0;
    if (q0 < q1) {
        load_chunk();
        stage(0);
        __syncthreads();
        if (q0 + 1 < q1) { advance(); load_chunk(); }
    }
    for (int q = q0; q < q1; ++q) {
        const __bf16* xs = Xs + cur * (BM * B16_KP) + (wm * WM + l31) * B16_KP + 8 * hi;
        const __bf16* ws = Ws + cur * (BN * B16_KP) + (wco * WCO + l31) * B16_KP + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ws + ks * 16);
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(xs + j * 32 * B16_KP + ks * 16);
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0][j], 0, 0, 0);
            }
        }
        if (q + 1 < q1) stage(cur ^ 1);
        __syncthreads();
        if (q + 2 < q1) { advance(); load_chunk(); }
        cur ^= 1;
    }
    igemm_epilogue<NCO, NM, WCO, WM>(acc, tab, gi, gd, bias, slab, Cout, flags, nsplit, m0, co0, M, DHW, wm, wco, l31, hi);
}

// STRIP3 form of the bf16-compute GEMM (see conv_igemm_strip3_kernel): one barrier round = the staged strip of a (kernel row,
// 32-channel block) + the weights of its THREE dx taps = 6 (64-voxel tile) / 12 (128) bf16 MFMAs per wave, where a per-dx
// round would run 2 / 4 between two barriers. Member state and addressing as in the fp32 strip3 kernel: 32-bit byte offsets
// through buffer loads (host: strip_fits32), lane tables read back with v_readlane, a zero ROW for the row-end lanes.
// Single LDS stage, two barriers per round; the next round's gathers are issued right after the first.
template <int BM>
__global__ __launch_bounds__(256) void conv_igemm_bf16_strip3_kernel(const GroupTable tab, const __bf16* __restrict__ wpb,
                                                                     const float* __restrict__ bias, float* __restrict__ slab,
                                                                     const int Cin, const int Cout, const int flags, const int nsplit) {
    constexpr int BN = 64, BKT = 32, WAVES_CO = 2, WAVES_M = 2;
    constexpr int WCO = BN / WAVES_CO, WM = BM / WAVES_M;
    constexpr int NCO = WCO / 32, NM = WM / 32;
    constexpr int KPT = BKT * BM / 256;          // channels per thread of the strip (16 for BM = 128, 8 for 64)
    constexpr int HT = BKT / KPT;                // threads that cover the 32 channels of ONE halo voxel
    constexpr int ZROW = BM + 2;                 // the all-zero row
    static_assert(NCO == 1 && NM >= 1 && (KPT == 16 || KPT == 8), "tile");
    __shared__ __attribute__((aligned(16))) __bf16 Xs[(BM + 3) * B16_KP];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[3 * BN * B16_KP];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave % WAVES_CO, wm = wave / WAVES_CO;
    const int tile = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && tile >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int D = gd.D, H = gd.H, W = gd.W;
    const int HW = H * W, DHW = D * HW;
    // frame-strided input rows (dstride = 2) / strided output (ydstride = 2): as in conv_igemm_strip3_kernel
    const int ds = gd.dstride == 2 ? 2 : 1;
    const int yds = gd.ydstride == 2 ? 2 : 1;
    const int Do = yds == 2 ? (gd.Dy + 1 - gd.yoff) / 2 : (ds == 2 ? (D + 1) / 2 : D);
    const int DHWo = Do * HW;
    const int M = gd.N * DHWo;
    const int m0 = (tile - tab.tile_start[gi]) * BM, co0 = blockIdx.y * BN;
    const int ntaps = gd.ntaps;
    const int ndx = gd.dx[0] < 0 ? 3 : 1;
    const int nrow = ntaps / ndx;

    const int lane_r = lane < nrow ? lane : 0, lane_t = lane < ntaps ? lane : 0;
    // (the mask loops below read these per-lane copies with v_readlane: indexing the kernel-argument tables with the loop counter
    // made hipcc issue a global_load_sbyte + s_waitcnt vmcnt(0) PER ITERATION — 9 x 2 dependent loads = most of a ~4 us prologue)
    const int tab_rdz = gd.dz[lane_r * ndx], tab_rdy = gd.dy[lane_r * ndx];
    const int tab_roff = (tab_rdz * HW + tab_rdy * W) * 4;
    const int tab_widx = gd.widx[lane_t];
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)(gd.N * DHW) * (uint32_t)Cin * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wpb, 0, -1, 0x00020000);

    const int ma_l = tid % BM, kq = tid / BM;
    const bool halo_thread = tid < 2 * HT;
    const int he = tid / HT, hq = tid % HT;          // halo: he = 0 left (voxel m0 - 1), 1 right (voxel m0 + BM)
    uint32_t rowmask = 0, rowmask_h = 0;
    uint32_t xbase = 0, xbase_h = 0;
    {
        const int m_a = m0 + ma_l;
        if (m_a < M) {
            const int n = m_a / DHWo, spo = m_a - n * DHWo;
            const int d_o = spo / HW, r = spo - d_o * HW;
            const int d = d_o * ds, sp = d * HW + r;
            const int h = r / W;
            xbase = (uint32_t)n * (uint32_t)Cin * (uint32_t)DHW + (uint32_t)sp;
            for (int t = 0; t < nrow; ++t) {
                const int dd = d + __builtin_amdgcn_readlane(tab_rdz, t), hh = h + __builtin_amdgcn_readlane(tab_rdy, t);
                if ((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H) rowmask |= 1u << t;
            }
        }
        const int m_h = he ? m0 + BM : m0 - 1;
        if (halo_thread && m_h >= 0 && m_h < M) {
            const int n = m_h / DHWo, spo = m_h - n * DHWo;
            const int d_o = spo / HW, r = spo - d_o * HW;
            const int d = d_o * ds, sp = d * HW + r;
            const int h = r / W;
            xbase_h = (uint32_t)n * (uint32_t)Cin * (uint32_t)DHW + (uint32_t)sp;
            for (int t = 0; t < nrow; ++t) {
                const int dd = d + __builtin_amdgcn_readlane(tab_rdz, t), hh = h + __builtin_amdgcn_readlane(tab_rdy, t);
                if ((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H) rowmask_h |= 1u << t;
            }
        }
    }
    bool can_l[NM], can_r[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int m = m0 + wm * WM + j * 32 + l31;
        const int w_ = m % W;
        can_l[j] = w_ > 0;
        can_r[j] = w_ < W - 1;
    }
    const int wc_l = tid >> 2, wk_l = (tid & 3) * 8;
    const bool w_ok = co0 + wc_l < Cout;
    const float relu_floor = (flags & T2V_CONV_RELU_IN) ? 0.f : -__builtin_inff();
    const uint32_t xoff = (xbase + (uint32_t)(kq * KPT) * (uint32_t)DHW) * 4u, xoff_h = (xbase_h + (uint32_t)(hq * KPT) * (uint32_t)DHW) * 4u;
    const uint32_t woff = w_ok ? (uint32_t)((co0 + wc_l) * Cin + wk_l) * 2u : 0u;

    f32x16 acc[NCO][NM];
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    float ra[KPT], rh[KPT];
    bf16x8 rw8[3];
    const int ncb = Cin / BKT;
    const int nrounds = nrow * ncb;
    const int rps = (nrounds + nsplit - 1) / nsplit;
    const int q0 = blockIdx.z * rps;
    int q1 = q0 + rps;
    if (q1 > nrounds) q1 = nrounds;
    bool pend_av = false, pend_hv = false;

    auto load_round = [&](int q) {
        const int r_cur = q / ncb, cb = q - r_cur * ncb;
        const int c0 = cb * BKT;
        const int roff = __builtin_amdgcn_readlane(tab_roff, r_cur);
        const int sx = c0 * DHW * 4;
        pend_av = (rowmask >> r_cur) & 1u;
        const uint32_t vo = xoff + (pend_av ? (uint32_t)roff : 0u);
#pragma unroll
        for (int j = 0; j < KPT; ++j) ra[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo, sx + j * 4 * DHW, 0));
        if (halo_thread) {
            pend_hv = (rowmask_h >> r_cur) & 1u;
            const uint32_t vh = xoff_h + (pend_hv ? (uint32_t)roff : 0u);
#pragma unroll
            for (int j = 0; j < KPT; ++j) rh[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vh, sx + j * 4 * DHW, 0));
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d < ndx) {                                        // (uniform)
                const int sw = (__builtin_amdgcn_readlane(tab_widx, r_cur * ndx + d) * Cout * Cin + c0) * 2;
                rw8[d] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, sw, 0));
            }
        }
    };
    auto put_row = [&](__bf16* dst, const float* v, bool ok) {
#pragma unroll
        for (int g = 0; g < KPT / 8; ++g) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)(ok ? fmaxf(v[g * 8 + j], relu_floor) : 0.f);
            *reinterpret_cast<bf16x8*>(dst + g * 8) = o;
        }
    };
    bf16x8 zero8;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero8[j] = (__bf16)0.f;
    auto stage = [&]() {
        put_row(Xs + (1 + ma_l) * B16_KP + kq * KPT, ra, pend_av);
        if (halo_thread) put_row(Xs + (he ? BM + 1 : 0) * B16_KP + hq * KPT, rh, pend_hv);
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d < ndx) *reinterpret_cast<bf16x8*>(Ws + d * (BN * B16_KP) + wc_l * B16_KP + wk_l) = w_ok ? rw8[d] : zero8;
    };

    if (tid < 4) *reinterpret_cast<bf16x8*>(Xs + ZROW * B16_KP + tid * 8) = zero8;      // the zero row (staging never writes it)
    if (q0 < q1) load_round(q0);
    for (int q = q0; q < q1; ++q) {
        stage();
        __syncthreads();
        if (q + 1 < q1) load_round(q + 1);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d < ndx) {
                const int dx = ndx == 3 ? d - 1 : 0;
                const __bf16* ws = Ws + d * (BN * B16_KP) + (wco * WCO + l31) * B16_KP + 8 * hi;
                const __bf16* xs[NM];
#pragma unroll
                for (int j = 0; j < NM; ++j) {
                    const bool keep = dx < 0 ? can_l[j] : (dx > 0 ? can_r[j] : true);
                    xs[j] = Xs + (keep ? 1 + dx + wm * WM + j * 32 + l31 : ZROW) * B16_KP + 8 * hi;
                }
#pragma unroll
                for (int ks = 0; ks < BKT / 16; ++ks) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(ws + ks * 16);
#pragma unroll
                    for (int j = 0; j < NM; ++j) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(xs[j] + ks * 16);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0][j], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    igemm_epilogue<NCO, NM, WCO, WM>(acc, tab, gi, gd, bias, slab, Cout, flags, nsplit, m0, co0, M, DHWo, wm, wco, l31, hi, yds, gd.yoff,
                                     gd.Dy * HW, HW);
}

// w[Cout][Cin][T] fp32 -> bf16 wpb[j][rows][K] with K contiguous: mode 0 rows = co, K = ci (forward);
// mode 1 rows = ci, K = co, mirrored taps (data gradient)
__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wpb, int Cout, int Cin, int T,
                                        TapList taps, int mode) {
    const int j = blockIdx.y;
    const int t = mode ? (T - 1 - taps.t[j]) : taps.t[j];
    const long n = (long)Cout * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int row, k;
        if (mode == 0) { row = (int)(i / Cin); k = (int)(i - (long)row * Cin); wpb[(size_t)j * n + i] = (__bf16)w[((size_t)row * Cin + k) * T + t]; }
        else { row = (int)(i / Cout); k = (int)(i - (long)row * Cout); wpb[(size_t)j * n + i] = (__bf16)w[((size_t)k * Cin + row) * T + t]; }
    }
}
extern "C" int t2v_pack_weight_bf16(const float* w, void* wpb, int Cout, int Cin, int T, const int32_t* taps, int ntaps, int mode,
                                    void* stream) {
    if (!w || !wpb || !taps || Cout < 1 || Cin < 1 || T < 1 || ntaps < 1 || ntaps > T2V_MAX_TAPS) return T2V_EINVAL;
    TapList tl;
    tl.n = ntaps;
    for (int i = 0; i < T2V_MAX_TAPS; ++i) tl.t[i] = i < ntaps ? taps[i] : 0;
    long nb = ((long)Cout * Cin + 255) / 256;
    if (nb > 1024) nb = 1024;
    T2V_LAUNCH(pack_weight_bf16_kernel, dim3((unsigned)nb, (unsigned)ntaps), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wpb, Cout, Cin,
               T, tl, mode);
    return launch_status();
}

// y_g = (accum ? y_g : 0) + bias[co] + sum_s slab[s]   (fixed summation order), for every group
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GroupTable tab, const float* __restrict__ slab,
                                                            const float* __restrict__ bias, int S, int Cout, int flags) {
    const bool has_bias = (flags & T2V_CONV_BIAS) && bias != nullptr;
    const bool accum = flags & T2V_CONV_ACCUM;
    const long total = tab.out_start[tab.n];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int gi = 0;
#pragma unroll
        for (int k = 1; k < T2V_MAX_GROUPS; ++k)
            if (k < tab.n && i >= tab.out_start[k]) gi = k;
        const long li = i - tab.out_start[gi];
        float v = 0.f;
#pragma unroll 8
        for (int s = 0; s < S; ++s) v += slab[(size_t)s * total + i];      // (unrolled: 8 loads in flight, summed in order)
        if (has_bias) {
            const int DHW = tab.g[gi].D * tab.g[gi].H * tab.g[gi].W;
            v += bias[(li / DHW) % Cout];
        }
        float* y = tab.g[gi].y;
        if ((flags & T2V_CONV_MASK_OUT) && tab.g[gi].mask) v = tab.g[gi].mask[li] > 0.f ? v : 0.f;
        y[li] = accum ? y[li] + v : v;
    }
}


// ------------------------------------------------------------------------------------------------
// thin-N convolution: Cout <= 4 (the render blocks' ch -> 1|3 convs, the data gradient of the C -> 64 stem
// convs, the 1024 -> 1 heads). One MFMA tile would waste >= 28/32 of its columns; here a lane owns one
// output voxel, the whole packed weight sits in LDS and the input is streamed with lane-contiguous loads:
// bandwidth-bound on the (cache-served) 27x / 9x re-read of the input.
// ------------------------------------------------------------------------------------------------
#define THIN_MAX_W 16384      // floats of LDS for the packed weight (64 KiB)

template <int NC>
__global__ __launch_bounds__(256) void conv_thin_kernel(const GroupTable tab, const float* __restrict__ wp,
                                                        const float* __restrict__ bias, const int Cin, const int Cout,
                                                        const int nslots, const int flags) {
    __shared__ float sw[THIN_MAX_W];
    __shared__ int s_off[T2V_MAX_TAPS];
    __shared__ int s_widx[T2V_MAX_TAPS];
    const int tid = threadIdx.x;
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && (int)blockIdx.x >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int D = gd.D, H = gd.H, W = gd.W, HW = H * W, DHW = D * HW;
    const int M = gd.N * DHW;
    const int ntaps = gd.ntaps;
    for (int i = tid; i < nslots * Cin * Cout; i += 256) sw[i] = wp[i];
    __shared__ int s_d3[T2V_MAX_TAPS];     // (dz+1) | (dy+1) << 2 | (dx+1) << 4: the tap loop reads LDS, not the kernel arguments
    if (tid < ntaps) {
        const int dz = gd.dz[tid], dy = gd.dy[tid], dx = gd.dx[tid];
        s_off[tid] = dz * HW + dy * W + dx;
        s_d3[tid] = (dz + 1) | ((dy + 1) << 2) | ((dx + 1) << 4);
        s_widx[tid] = gd.widx[tid];
    }
    __syncthreads();
    const int m = ((int)blockIdx.x - tab.tile_start[gi]) * 256 + tid;
    if (m >= M) return;
    const int n = m / DHW, sp = m - n * DHW;
    const int d = sp / HW, r = sp - d * HW;
    const int h = r / W, w_ = r - h * W;
    const float* __restrict__ px = gd.x + (size_t)n * Cin * DHW + sp;
    const bool relu_in = flags & T2V_CONV_RELU_IN;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    for (int t = 0; t < ntaps; ++t) {
        const int d3 = s_d3[t];
        const int dd = d + (d3 & 3) - 1, hh = h + ((d3 >> 2) & 3) - 1, ww = w_ + ((d3 >> 4) & 3) - 1;
        if (!((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W)) continue;
        const float* p = px + (ptrdiff_t)s_off[t];
        const float* wt = sw + (size_t)s_widx[t] * Cin * Cout;
        int ci = 0;
        for (; ci + 4 <= Cin; ci += 4) {
            float v0 = p[(size_t)ci * DHW], v1 = p[(size_t)(ci + 1) * DHW], v2 = p[(size_t)(ci + 2) * DHW], v3 = p[(size_t)(ci + 3) * DHW];
            if (relu_in) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
#pragma unroll
            for (int c = 0; c < NC; ++c)
                acc[c] += v0 * wt[ci * Cout + c] + v1 * wt[(ci + 1) * Cout + c] + v2 * wt[(ci + 2) * Cout + c] + v3 * wt[(ci + 3) * Cout + c];
        }
        for (; ci < Cin; ++ci) {
            float v = p[(size_t)ci * DHW];
            if (relu_in) v = fmaxf(v, 0.f);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] += v * wt[ci * Cout + c];
        }
    }
    const bool has_bias = (flags & T2V_CONV_BIAS) && bias != nullptr;
    const bool accum = flags & T2V_CONV_ACCUM;
    const bool mask_out = (flags & T2V_CONV_MASK_OUT) && gd.mask != nullptr;
    float* py = gd.y + (size_t)n * Cout * DHW + sp;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c < Cout) {
            float v = acc[c] + (has_bias ? bias[c] : 0.f);
            if (mask_out) v = gd.mask[(size_t)n * Cout * DHW + sp + (size_t)c * DHW] > 0.f ? v : 0.f;
            py[(size_t)c * DHW] = accum ? py[(size_t)c * DHW] + v : v;
        }
    }
}


// Linear head with <= 4 outputs (resnet3d.py:33-35: 1024(+cond) -> 1): one WAVE per input row, lanes stride over the
// features (coalesced), wavefront reduction. (The voxel-per-lane kernel above would walk a 4 KB-strided row per lane.)
template <int NC>
__global__ __launch_bounds__(256) void linear_thin_kernel(const GroupTable tab, const float* __restrict__ wp,
                                                          const float* __restrict__ bias, const int Cin, const int Cout,
                                                          const int flags) {
    const int lane = threadIdx.x & 63;
    const int row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);     // rows of all members, concatenated
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && row >= tab.tile_start[k]) gi = k;
    if (row >= tab.tile_start[tab.n]) return;
    const t2v_conv_group& gd = tab.g[gi];
    const int r = row - tab.tile_start[gi];
    const float* __restrict__ px = gd.x + (size_t)r * Cin;
    const bool relu_in = flags & T2V_CONV_RELU_IN;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    for (int ci = lane; ci < Cin; ci += 64) {
        float v = px[ci];
        if (relu_in) v = fmaxf(v, 0.f);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < Cout) acc[c] += v * wp[(size_t)ci * Cout + c];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
    if (lane == 0) {
        const bool has_bias = (flags & T2V_CONV_BIAS) && bias != nullptr;
        const bool accum = flags & T2V_CONV_ACCUM;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < Cout) {
                float v = acc[c] + (has_bias ? bias[c] : 0.f);
                float* p = gd.y + (size_t)r * Cout + c;
                if ((flags & T2V_CONV_MASK_OUT) && gd.mask) v = gd.mask[(size_t)r * Cout + c] > 0.f ? v : 0.f;
                *p = accum ? *p + v : v;
            }
    }
}

// Thin convolution with ONE output channel and many taps (the data gradient of the 1 -> 64 stem convolution,
// 64 channels x 27 taps -> 1), in two passes that read the input once instead of 27 times:
//   taps pass:   P[slot][m] = sum_ci x[m][ci] * w[slot][ci]      for every voxel m and packed tap slot
//   shift pass:  y[m]       = sum_taps valid(m, tap) * P[slot(tap)][m + offset(tap)]
// P (nslots x M floats per member, members back to back) lives in the caller's workspace.
__global__ __launch_bounds__(256) void thin_taps_kernel(const GroupTable tab, const float* __restrict__ wp, float* __restrict__ P,
                                                        const int Cin, const int nslots, const int flags) {
    // (round 3: reading the weights at wave-uniform addresses through scalar loads instead of this LDS copy was measured and dropped:
    // 441 dependent s_load / s_waitcnt pairs per wave, 24.6 -> 41.7 us per launch)
    __shared__ float sw[THIN_MAX_W];                     // w[slot][ci]
    const int tid = threadIdx.x;
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && (int)blockIdx.x >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int DHW = gd.D * gd.H * gd.W, M = gd.N * DHW;
    for (int i = tid; i < nslots * Cin; i += 256) sw[i] = wp[i];          // Cout == 1: wp[slot][ci][0]
    __syncthreads();
    const int m = ((int)blockIdx.x - tab.tile_start[gi]) * 256 + tid;
    if (m >= M) return;
    const int n = m / DHW, sp = m - n * DHW;
    const float* __restrict__ px = gd.x + (size_t)n * Cin * DHW + sp;
    const bool relu_in = flags & T2V_CONV_RELU_IN;
    float acc[T2V_MAX_TAPS];
#pragma unroll
    for (int t = 0; t < T2V_MAX_TAPS; ++t) acc[t] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += 16) {               // 16 channel loads in flight per lane (latency, not bytes, bounds this)
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int ci = c0 + u < Cin ? c0 + u : Cin - 1;
            v[u] = px[(size_t)ci * DHW];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (relu_in) v[u] = fmaxf(v[u], 0.f);
            if (c0 + u >= Cin) v[u] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < T2V_MAX_TAPS; ++t) {
            if (t < nslots) {
                const float* wt = sw + t * Cin + c0;     // rows past Cin multiply zeros; THIN_MAX_W leaves the slack
#pragma unroll
                for (int u = 0; u < 16; ++u) acc[t] += v[u] * wt[c0 + u < Cin ? u : 0];
            }
        }
    }
    // P of this member starts at nslots * (voxels of the members before it) = nslots * out_start (Cout == 1)
    float* pp = P + (size_t)nslots * (size_t)tab.out_start[gi] + m;
#pragma unroll
    for (int t = 0; t < T2V_MAX_TAPS; ++t)
        if (t < nslots) pp[(size_t)t * M] = acc[t];
}
// The same first pass on the matrix pipes (round 3; Cin in {16, 32, 64, 128}): P[t][v] = sum_c w[t][c] x[c][v] is a [taps x Cin] . [Cin x
// voxels] product with at most 27 rows. A = the weights (row = tap, held in K2 registers per lane for the whole kernel), B = the
// activations straight from global memory (column = voxel: 32 consecutive voxels per half-wave, one k per half), no LDS at all. A
// wave owns 32 voxels = one 32x32x2 column tile (512-thread workgroups); rows (registers) = taps, columns (lanes) = voxels, so the stores are the same
// 128-byte segments as the VALU form's. Measured on the stem's 64 -> 1 data gradient (M = 131 072): 40.9 us for the VALU form, 38 with 64
// voxels per wave here, 31 with 32 — the pass is bound by the latency of its channel-strided loads (waves in flight per SIMD), not by
// the arithmetic.
template <int K2>
__global__ __launch_bounds__(512) void thin_taps_mfma_kernel(const GroupTable tab, const float* __restrict__ wp, float* __restrict__ P,
                                                             const int nslots, const int flags) {
    constexpr int Cin = 2 * K2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && (int)blockIdx.x >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int DHW = gd.D * gd.H * gd.W, M = gd.N * DHW;
    const bool relu_in = flags & T2V_CONV_RELU_IN;
    float a[K2];
#pragma unroll
    for (int k2 = 0; k2 < K2; ++k2) a[k2] = l31 < nslots ? wp[l31 * Cin + 2 * k2 + hi] : 0.f;      // Cout == 1: wp[slot][ci]
    float* pbase = P + (size_t)nslots * (size_t)tab.out_start[gi];
    const int m_w = ((int)blockIdx.x - tab.tile_start[gi]) * 256 + wave * 32;        // eight waves x 32 voxels: more loads in flight per SIMD
    {
        const int m = m_w + l31;
        if (m_w >= M) return;                                           // (wave-uniform)
        const int mc = m < M ? m : M - 1;
        const int n = mc / DHW, sp = mc - n * DHW;
        const float* __restrict__ px = gd.x + ((size_t)n * Cin + hi) * DHW + sp;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float b[K2];
#pragma unroll
        for (int k2 = 0; k2 < K2; ++k2) b[k2] = px[(size_t)(2 * k2) * DHW];
#pragma unroll
        for (int k2 = 0; k2 < K2; ++k2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2], relu_in ? fmaxf(b[k2], 0.f) : b[k2], acc, 0, 0, 0);
        if (m < M) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (t < nslots) pbase[(size_t)t * M + m] = acc[r];
            }
        }
    }
}
__global__ __launch_bounds__(256) void thin_shift_sum_kernel(const GroupTable tab, const float* __restrict__ P,
                                                             const float* __restrict__ bias, const int nslots, const int flags) {
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && (int)blockIdx.x >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int D = gd.D, H = gd.H, W = gd.W, HW = H * W, DHW = D * HW, M = gd.N * DHW;
    const int m = ((int)blockIdx.x - tab.tile_start[gi]) * 256 + (int)threadIdx.x;
    const int ntaps = gd.ntaps;
    // per-lane copies of the tap tables, read back with v_readlane (indexing the kernel arguments per tap cost one dependent
    // byte load + wait per table access); loaded before any lane leaves
    const int lane_t = ((int)threadIdx.x & 63) < ntaps ? ((int)threadIdx.x & 63) : 0;
    const int tab_tdz = gd.dz[lane_t], tab_tdy = gd.dy[lane_t], tab_tdx = gd.dx[lane_t], tab_tw = gd.widx[lane_t];
    if (m >= M) return;
    const int n = m / DHW, sp = m - n * DHW;
    const int d = sp / HW, r = sp - d * HW;
    const int h = r / W, w_ = r - h * W;
    const float* __restrict__ pp = P + (size_t)nslots * (size_t)tab.out_start[gi] + m;
    float acc = 0.f;
    float pv[T2V_MAX_TAPS];
#pragma unroll
    for (int t = 0; t < T2V_MAX_TAPS; ++t) {             // all (unconditional, clamped) loads first ...
        const int tt = t < ntaps ? t : 0;
        const int tdz = __builtin_amdgcn_readlane(tab_tdz, tt), tdy = __builtin_amdgcn_readlane(tab_tdy, tt),
                  tdx = __builtin_amdgcn_readlane(tab_tdx, tt);
        const int dd = d + tdz, hh = h + tdy, ww = w_ + tdx;
        const bool ok = t < ntaps && (unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const int off = ok ? tdz * HW + tdy * W + tdx : 0;
        const float v = pp[(size_t)__builtin_amdgcn_readlane(tab_tw, tt) * M + off];
        pv[t] = ok ? v : 0.f;
    }
#pragma unroll
    for (int t = 0; t < T2V_MAX_TAPS; ++t) acc += pv[t];   // ... then the sum in tap order
    float v = acc + (((flags & T2V_CONV_BIAS) && bias) ? bias[0] : 0.f);
    if ((flags & T2V_CONV_MASK_OUT) && gd.mask) v = gd.mask[m] > 0.f ? v : 0.f;
    gd.y[m] = (flags & T2V_CONV_ACCUM) ? gd.y[m] + v : v;
}
// ------------------------------------------------------------------------------------------------
// NARROW-INPUT convolution (Cin <= 3: the stem's first convolution on 1- or 3-channel clips, resnet3d.py:13). K = 27 * Cin is far
// too short for a matrix tile (the generic implicit GEMM ran it at 1.4 TB/s of output bandwidth); the layer is one read of the clip
// and Cout writes per voxel, i.e. HBM-write bound. One lane per voxel: its <= 27 * Cin input values sit in registers, the weights
// arrive through scalar loads (uniform addresses: wp[tap][ci][co .. co+3] as one s_load_dwordx4), four output channels at a time,
// every store a 256-byte row of consecutive voxels. Algorithmic traffic: 4 * Cin B read + 4 * Cout B written per voxel.
// ------------------------------------------------------------------------------------------------
// The grey-clip stem (Cin = 1, Cout a multiple of 32) on the matrix pipes: y[co][v] = sum_t w[t][co] x[v + off_t] is a [Cout x taps] .
// [taps x voxels] product with K = 27 (padded to 28). A = the weights (rows = channels), B = the shifted input values straight from
// global memory (columns = 32 consecutive voxels per half-wave, one tap per half), so rows (registers) = channels and columns
// (lanes) = voxels: every store is a 128-byte run of one channel. Eight waves x 32 voxels per workgroup, no LDS. The lane-per-voxel
// form below spends 1 728 FMAs + scalar weight loads per lane on what is 28 MFMAs per 32 voxels here; the kernel is then bound by
// its 256 B of output per voxel.
__global__ __launch_bounds__(512) void conv_stem_mfma_kernel(const GroupTable tab, const float* __restrict__ wp,
                                                             const float* __restrict__ bias, const int Cout, const int flags) {
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && (int)blockIdx.x >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int D = gd.D, H = gd.H, W = gd.W, HW = H * W, DHW = D * HW;
    const int M = gd.N * DHW;
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, l31 = lane & 31, hi = lane >> 5;
    const int m_w = ((int)blockIdx.x - tab.tile_start[gi]) * 256 + wave * 32;
    if (m_w >= M) return;                                               // (wave-uniform)
    const int m = m_w + l31;
    const bool mv = m < M;
    const int mm = mv ? m : M - 1;
    const int n = mm / DHW, sp = mm - n * DHW;
    const int d = sp / HW, r_ = sp - d * HW;
    const int h = r_ / W, w_ = r_ - h * W;
    const float* __restrict__ px = gd.x + (size_t)n * DHW;
    const bool relu = flags & T2V_CONV_RELU_IN;
    const bool has_bias = (flags & T2V_CONV_BIAS) && bias != nullptr;
    const int ntaps = gd.ntaps;
    const int lt = lane < ntaps ? lane : 0;
    const int tab_dz = gd.dz[lt], tab_dy = gd.dy[lt], tab_dx = gd.dx[lt];
    const int tab_w = gd.widx[lt] * Cout;
    // B operand: lane (voxel l31, k-slot hi) holds x at tap 2 k2 + hi
    float b[14];
    int wrow[14];                                                       // float offset of this lane's tap row in wp, -1: padding tap
#pragma unroll
    for (int k2 = 0; k2 < 14; ++k2) {
        const int t0 = 2 * k2, t1 = 2 * k2 + 1;                         // (t1 = 27 is the K padding)
        const int dz = hi ? __builtin_amdgcn_readlane(tab_dz, t1 < 27 ? t1 : 0) : __builtin_amdgcn_readlane(tab_dz, t0);
        const int dy = hi ? __builtin_amdgcn_readlane(tab_dy, t1 < 27 ? t1 : 0) : __builtin_amdgcn_readlane(tab_dy, t0);
        const int dx = hi ? __builtin_amdgcn_readlane(tab_dx, t1 < 27 ? t1 : 0) : __builtin_amdgcn_readlane(tab_dx, t0);
        const int wr = hi ? __builtin_amdgcn_readlane(tab_w, t1 < 27 ? t1 : 0) : __builtin_amdgcn_readlane(tab_w, t0);
        const int t = t0 + hi;
        const bool live = t < ntaps;
        const int dd = d + dz, hh = h + dy, ww = w_ + dx;
        const bool ok = live && mv && (unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const float v = px[ok ? (dd * H + hh) * W + ww : 0];
        b[k2] = ok ? (relu ? fmaxf(v, 0.f) : v) : 0.f;
        wrow[k2] = live ? wr : -1;
    }
    float* __restrict__ py = gd.y + (size_t)n * Cout * DHW + sp;
    for (int co0 = 0; co0 < Cout; co0 += 32) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = has_bias ? bias[co0 + (r & 3) + 8 * (r >> 2) + 4 * hi] : 0.f;
        float a[14];
#pragma unroll
        for (int k2 = 0; k2 < 14; ++k2) a[k2] = wrow[k2] >= 0 ? wp[wrow[k2] + co0 + l31] : 0.f;     // A: lane (channel l31, k-slot hi)
#pragma unroll
        for (int k2 = 0; k2 < 14; ++k2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k2], b[k2], acc, 0, 0, 0);
        if (mv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) py[(size_t)(co0 + (r & 3) + 8 * (r >> 2) + 4 * hi) * DHW] = acc[r];
        }
    }
}
template <int CIN>
__global__ __launch_bounds__(256) void conv_stem_kernel(const GroupTable tab, const float* __restrict__ wp,
                                                            const float* __restrict__ bias, const int Cout, const int flags) {
    // one lane per voxel, all Cout channels: the <= 27 * Cin input values sit in registers, the weights arrive through scalar loads
    // (per-tap tables live one entry per lane and are read back with v_readlane: no dependent scalar loads, so the 27 weight loads of
    // a channel group are in flight together). Measured against the alternatives (graph-timed, 8 D-step members, 393 216 voxels): this
    // form 59 us, the generic implicit GEMM 74 us, a channel-group-major form with contiguous 16 KB output runs 171 us (its 16x
    // repeated address / bounds arithmetic costs more than the stores it straightens).
    int gi = 0;
#pragma unroll
    for (int k = 1; k < T2V_MAX_GROUPS; ++k)
        if (k < tab.n && (int)blockIdx.x >= tab.tile_start[k]) gi = k;
    const t2v_conv_group& gd = tab.g[gi];
    const int D = gd.D, H = gd.H, W = gd.W, HW = H * W, DHW = D * HW;
    const int M = gd.N * DHW;
    const int m = ((int)blockIdx.x - tab.tile_start[gi]) * 256 + (int)threadIdx.x;
    const bool mv = m < M;
    const int mm = mv ? m : 0;
    const int n = mm / DHW, sp = mm - n * DHW;
    const int d = sp / HW, r = sp - d * HW;
    const int h = r / W, w_ = r - h * W;
    const float* __restrict__ px = gd.x + (size_t)n * CIN * DHW;
    const bool relu = flags & T2V_CONV_RELU_IN;
    const bool has_bias = (flags & T2V_CONV_BIAS) && bias != nullptr;
    const int ntaps = gd.ntaps;
    const int lane = (int)threadIdx.x & 63;
    const int lt = lane < ntaps ? lane : 0;
    const int tab_dz = gd.dz[lt], tab_dy = gd.dy[lt], tab_dx = gd.dx[lt];
    const int tab_w = lane < ntaps ? gd.widx[lt] * CIN * Cout : 0;
    float xv[T2V_MAX_TAPS * CIN];
#pragma unroll
    for (int t = 0; t < T2V_MAX_TAPS; ++t) {
        const bool live = t < ntaps;
        const int dd = d + __builtin_amdgcn_readlane(tab_dz, t), hh = h + __builtin_amdgcn_readlane(tab_dy, t);
        const int ww = w_ + __builtin_amdgcn_readlane(tab_dx, t);
        const bool ok = live && mv && (unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const int off = ok ? (dd * H + hh) * W + ww : 0;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const float v = px[(size_t)ci * DHW + off];
            xv[t * CIN + ci] = ok ? (relu ? fmaxf(v, 0.f) : v) : 0.f;
        }
    }
    float* __restrict__ py = gd.y + (size_t)n * Cout * DHW + sp;
    for (int co = 0; co < Cout; co += 4) {
        float a0 = has_bias ? bias[co] : 0.f, a1 = has_bias ? bias[co + 1] : 0.f;
        float a2 = has_bias ? bias[co + 2] : 0.f, a3 = has_bias ? bias[co + 3] : 0.f;
#pragma unroll
        for (int t = 0; t < T2V_MAX_TAPS; ++t) {
            const float* __restrict__ wt = wp + __builtin_amdgcn_readlane(tab_w, t) + co;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                const float4 w4 = *reinterpret_cast<const float4*>(wt + ci * Cout);
                const float v = xv[t * CIN + ci];
                a0 = fmaf(w4.x, v, a0); a1 = fmaf(w4.y, v, a1); a2 = fmaf(w4.z, v, a2); a3 = fmaf(w4.w, v, a3);
            }
        }
        if (mv) {
            py[(size_t)co * DHW] = a0;
            py[(size_t)(co + 1) * DHW] = a1;
            py[(size_t)(co + 2) * DHW] = a2;
            py[(size_t)(co + 3) * DHW] = a3;
        }
    }
}
// the narrow-input kernel takes: 1 or 3 input channels, Cout a multiple of 4 (>= 8), plain stores (no accumulate / masked epilogue /
// frame stride)
static bool stem_ok(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int flags) {
    if ((Cin != 1 && Cin != 3) || Cout < 8 || Cout > 256 || (Cout % 4) != 0) return false;      // (grey or RGB clips)
    if (flags & (T2V_CONV_ACCUM | T2V_CONV_MASK_OUT)) return false;
    long M = 0;
    int taps = 0;
    for (int i = 0; i < ngroups; ++i) {
        if (groups[i].dstride == 2 || groups[i].ydstride == 2) return false;
        M += (long)groups[i].N * groups[i].D * groups[i].H * groups[i].W;
        if (groups[i].ntaps > taps) taps = groups[i].ntaps;
    }
    // a lane per voxel walks ALL output channels and all 27 tap slots: only worth it for real 3-D stems with enough voxels to fill
    // the chip. (Other "1-channel convolutions" exist: the data gradient of a 1-output head is 32 voxels x 1024 channels and took
    // 220 us on this kernel; the data gradients of the render convolutions and the stem's 1x1x1 skip have 9 / 1 live taps and ran
    // 2-4x slower here than on the generic tiles.)
    return M >= 65536 && taps > 18;
}

// worth it when the input would otherwise be re-read many times: several taps, enough channels and voxels
static bool thin_two_pass(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int nslots) {
    if (Cout != 1 || nslots < 9 || Cin < 16 || nslots * Cin > THIN_MAX_W) return false;
    long M = 0;
    for (int i = 0; i < ngroups; ++i) M += (long)groups[i].N * groups[i].D * groups[i].H * groups[i].W;
    return M >= 16384;
}

static bool thin_ok(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int& nslots) {
    if (Cout > 4) return false;
    nslots = 0;
    for (int i = 0; i < ngroups; ++i)
        for (int t = 0; t < groups[i].ntaps; ++t)
            if (groups[i].widx[t] + 1 > nslots) nslots = groups[i].widx[t] + 1;
    return (long)nslots * Cin * Cout <= THIN_MAX_W;
}

// ------------------------------------------------------------------------------------------------
// Launch-heuristic tunables, in ONE place. The defaults are the measured optima on MI355X. The environment
// overrides exist for developer sweeps (tools/conv_suite.py) and for tests that force an instantiation onto a small input;
// tests/test_conv_plan.py::test_environment_tunables_move_the_plan checks every one of them through the plan queries.
// Read once, at the first launch or plan query of the process.
// ------------------------------------------------------------------------------------------------
struct Tunables {
    long tile128_min;      // T2V_TILE128_MIN   (768)  128-voxel tiles from this many tiles on, else 64
    long tile256_min;      // T2V_TILE256_MIN   (512)  256-voxel tiles from this many tiles on
    long nosplit_chunks;   // T2V_NOSPLIT_CHUNKS  (8)  no split-K for reductions of at most this many K chunks
    long force_splits;     // T2V_FORCE_S         (0)  > 0: this split-K count for every forward / data-gradient launch
    bool strip;            // T2V_NO_STRIP unset       strip (three-dx-taps-per-row) kernels enabled
    bool occ_pad;          // T2V_NO_OCC_PAD unset     resident-workgroup choice of the 256-voxel tile (occupancy_pad)
    long wgrad_target;     // T2V_WGRAD_TARGET    (0)  weight-gradient workgroups to aim at; 0: 1024 = one round of resident workgroups
    long wgrad_scap;       // T2V_WGRAD_SCAP    (256)  upper bound of the weight-gradient k-split count
    bool wgrad_quantise;   // T2V_WGRAD_NOQ unset      drop a nearly empty last round of weight-gradient workgroups
    long wgrad_min_cps;    // T2V_WGRAD_MINCPS    (4)  fewest 32-voxel chunks a weight-gradient k-split may own
    bool wgrad_thin;       // T2V_NO_WGRAD_THIN unset  the streaming kernel for narrow inputs (<= 31 (tap, ci) columns)
};
static long env_long(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
static const Tunables& tun() {
    static const Tunables t = {env_long("T2V_TILE128_MIN", 768), env_long("T2V_TILE256_MIN", 512), env_long("T2V_NOSPLIT_CHUNKS", 8),
                               env_long("T2V_FORCE_S", 0), getenv("T2V_NO_STRIP") == nullptr, getenv("T2V_NO_OCC_PAD") == nullptr,
                               env_long("T2V_WGRAD_TARGET", 0), env_long("T2V_WGRAD_SCAP", 256), getenv("T2V_WGRAD_NOQ") == nullptr,
                               env_long("T2V_WGRAD_MINCPS", 4), getenv("T2V_NO_WGRAD_THIN") == nullptr};
    return t;
}

struct ConvPlan { int bm, bn, bk; bool fast, vecb; int S; long tiles; bool dstride2; };

// frames the GEMM rows of a member run over: every frame; the even ones (dstride = 2); one parity of y's frames (ydstride = 2)
static inline int out_frames(const t2v_conv_group& g) {
    if (g.ydstride == 2) return (g.Dy + 1 - g.yoff) / 2;
    return g.dstride == 2 ? (g.D + 1) / 2 : g.D;
}
static bool group_ok(const t2v_conv_group& g, bool need_ptrs) {
    if (need_ptrs && (!g.x || !g.y)) return false;
    if (g.dstride < 0 || g.dstride > 2 || g.ydstride < 0 || g.ydstride > 2) return false;
    if (g.ydstride == 2 && (g.dstride == 2 || g.yoff < 0 || g.yoff > 1 || g.Dy < 1 || (g.Dy + 1 - g.yoff) / 2 < 1 || g.D < (g.Dy + 1) / 2)) return false;
    if (g.N < 1 || g.D < 1 || g.H < 1 || g.W < 1 || g.ntaps < 1 || g.ntaps > T2V_MAX_TAPS) return false;
    for (int t = 0; t < g.ntaps; ++t) {
        if (g.dz[t] < -1 || g.dz[t] > 1 || g.dy[t] < -1 || g.dy[t] > 1 || g.dx[t] < -1 || g.dx[t] > 1) return false;
        if (g.widx[t] < 0 || g.widx[t] >= T2V_MAX_TAPS) return false;
    }
    return true;
}

static bool build_table(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, bool need_ptrs, GroupTable& tab,
                        ConvPlan& p) {
    if (!groups || ngroups < 1 || ngroups > T2V_MAX_GROUPS || Cin < 1 || Cout < 1) return false;
    long Mtot = 0, Mmax = 0;
    int max_chunk_taps = 1;
    p.dstride2 = false;
    for (int i = 0; i < ngroups; ++i) {
        if (!group_ok(groups[i], need_ptrs)) return false;
        const long Min = (long)groups[i].N * groups[i].D * groups[i].H * groups[i].W;
        if (Min * (long)(Cin > Cout ? Cin : Cout) >= (1L << 31)) return false;     // 32-bit voxel indices
        const long M = (long)groups[i].N * out_frames(groups[i]) * groups[i].H * groups[i].W;      // GEMM rows = output voxels
        if (groups[i].dstride == 2 || groups[i].ydstride == 2) p.dstride2 = true;
        Mtot += M;
        if (M > Mmax) Mmax = M;
        if (groups[i].ntaps > max_chunk_taps) max_chunk_taps = groups[i].ntaps;
    }
    p.bk = (Cin % 64 == 0) ? 64 : (Cin % 32 == 0) ? 32 : 16;
    p.fast = (Cin % 16 == 0);
    p.vecb = (Cout % 4 == 0);
    if (Cout <= 32) { p.bm = 128; p.bn = 32; if (p.bk > 32) p.bk = 32; }
    else {
        p.bn = 64;
        const long t128 = ((Mtot + 127) / 128) * ((Cout + 63) / 64);
        p.bm = (t128 >= tun().tile128_min) ? 128 : 64;
        // 256 x 64 tiles, K chunks of 16, one wave = 64 co x 64 m (four accumulator chains, one LDS read per MFMA) for the
        // launches big enough to fill the chip with them (+3-6 % over the 128 x 64 tile there)
        if (p.fast && (Cin % 16) == 0 && ((Mtot + 255) / 256) * ((Cout + 63) / 64) >= tun().tile256_min) { p.bm = 256; p.bk = 16; }
        if (p.bk > 32 && p.bm != 256) p.bk = 32;   // two LDS stages: 48 KB (128x64) / 32 KB (64x64) per workgroup
    }
    if (!p.fast) p.bk = 16;
    long mt = 0, ot = 0;
    tab.n = ngroups;
    for (int i = 0; i < ngroups; ++i) {
        tab.g[i] = groups[i];
        const long M = (long)groups[i].N * out_frames(groups[i]) * groups[i].H * groups[i].W;
        tab.tile_start[i] = (int32_t)mt;
        tab.out_start[i] = ot;
        mt += (M + p.bm - 1) / p.bm;
        ot += M * Cout;
    }
    for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) { tab.tile_start[i] = (int32_t)mt; tab.out_start[i] = ot; }
    tab.tile_start[ngroups] = (int32_t)mt;
    tab.out_start[ngroups] = ot;
    p.tiles = mt * ((Cout + p.bn - 1) / p.bn);
    // split-K: the member with the longest reduction keeps >= 2 chunks per split; members with fewer chunks
    // than splits just leave their surplus splits empty (they write zero partial tiles)
    const long min_chunks = p.fast ? (long)max_chunk_taps * (Cin / p.bk) : ((long)max_chunk_taps * Cin + p.bk - 1) / p.bk;
    long S = 1;
    // a reduction of <= 8 chunks (the 1x1 convolutions up to 256 input channels) is not worth a split: the second launch costs
    // more than the idle CUs (measured: -26 launches, -0.07 ms per iteration)
    static const long split_target = env_long("T2V_SPLIT_TARGET", 768);      // workgroups a split launch aims at (developer sweeps)
    if (p.tiles < 384 && min_chunks > tun().nosplit_chunks) {
        S = (split_target + p.tiles - 1) / p.tiles;
        long maxS = min_chunks / 2;
        if (S > maxS) S = maxS;
        if (S > 64) S = 64;
        while (S > 1 && (double)S * ot * 4.0 > 256e6) --S;   // keep the slab small (L2 / MALL resident)
        if (S < 1) S = 1;
    }
    if (p.dstride2) S = 1;                        // (frame-strided outputs: the split-K reduce pass does not know them)
    const long force_S = tun().force_splits;                                                  // developer knob (tools/conv_suite.py sweeps)
    if (force_S > 0 && !p.dstride2) S = force_S > min_chunks / 2 ? (min_chunks / 2 > 0 ? min_chunks / 2 : 1) : force_S;
    p.S = (int)S;
    return true;
}

// strip variant: every member's taps come in product order with dx fastest, and members that are wider than one voxel
// carry all three dx taps per row
static bool strip_ok(const GroupTable& tab) {
    bool any3 = false;
    for (int i = 0; i < tab.n; ++i) {
        const t2v_conv_group& g = tab.g[i];
        const int ndx = g.dx[0] < 0 ? 3 : 1;
        if (g.ntaps % ndx) return false;
        for (int r = 0; r < g.ntaps / ndx; ++r)
            for (int d = 0; d < ndx; ++d) {
                const int t = r * ndx + d;
                if (g.dx[t] != (ndx == 3 ? d - 1 : 0) || g.dz[t] != g.dz[r * ndx] || g.dy[t] != g.dy[r * ndx]) return false;
            }
        if (ndx == 3 && g.W < 2) return false;
        any3 = any3 || ndx == 3;
    }
    return any3;
}
// the fp32 strip kernel gathers x through 32-bit byte offsets (buffer loads)
static bool strip_fits32(const GroupTable& tab, int Cin) {
    for (int i = 0; i < tab.n; ++i) {
        const t2v_conv_group& g = tab.g[i];
        if ((long)g.N * g.D * g.H * g.W * (long)Cin >= (1L << 30)) return false;
    }
    return true;
}

// Which instantiation a (tile, chunk) choice ends up in: the strip variant (three dx taps from one staged strip) when the
// members carry their taps in (row, dx) order. Shared by the launcher and by t2v_conv_fwd_plan. (`ks`: K-split wave layout of
// the plan query's out[6]; always 1 — the two-chains-per-wave form was dropped for the three-taps-per-round kernel.)
struct ConvVariant { bool strip; int ks; bool s3; };
static ConvVariant conv_variant(const GroupTable& tab, const ConvPlan& p, int BM, int BN, int BKT, int Cin, int Cout, int flags) {
    ConvVariant v{false, 1, false};
    if (p.fast && (BKT == 32 || BM == 256) && (Cin % BKT) == 0 && tun().strip && strip_ok(tab) && strip_fits32(tab, Cin)) {
        v.strip = true;
        // tiles with 64 output channels (64 / 128 voxels x 32 channels, 256 voxels x 16 channels): all three dx taps per barrier
        // round (conv_igemm_strip3_kernel; +12-15 % over one dx per round on every one of them); 128 x 32 keeps the per-dx form
        v.s3 = BN == 64;
    }
    return v;
}

// The double-buffered form of the 64- and 256-voxel strip3 tiles (conv_igemm_strip3_kernel<64, BKT, 2, true, true>, <256, 16, 1, true,
// true>): 0 = not taken, else its channels per round (64 voxels: 16 = four workgroups per CU, 32 = two; 256 voxels: 16). Every member
// must carry all three dx taps and Cout % 4 == 0. T2V_STRIP3_DB (0 / 16 / 32) and T2V_STRIP3_DB256 (0 / 1) override. Shared by the launcher and the plan query (out[3] = channels per round, out[6] = 2).
static int strip3_db(const GroupTable& tab, const ConvPlan& p) {
    static const long db = env_long("T2V_STRIP3_DB", 16), db256 = env_long("T2V_STRIP3_DB256", 1);
    if (!p.vecb || p.bn != 64 || (p.bm != 64 && p.bm != 256)) return 0;
    if (p.bm == 64 ? (db != 16 && db != 32) : db256 == 0) return 0;
    for (int i = 0; i < tab.n; ++i)
        if (tab.g[i].dx[0] >= 0) return 0;
    return p.bm == 64 ? (int)db : 16;              // (the 256-voxel tile: 16-channel rounds, two workgroups per CU as before)
}

// Wave quantisation of the 256-voxel tile: the kernel fits 3 workgroups on a CU but runs no faster per CU with 3 than with 2
// (2 x 255 us = 3 x 382 us per 256 tiles), so a launch of e.g. 1024 tiles (the generator step's D forward, M = 262144) took
// one round of 768 and a tail of 256 = 692 us where two rounds of 512 take 510. The launcher therefore picks the resident
// workgroups per CU (2 or 3) that minimises rounds x residents and pads the dynamic LDS request to enforce it.
struct OccInfo { int occ = 0, lds = 0, cus = 0; };
template <class K>
static int occupancy_pad(K kernel, OccInfo& info, long nwg) {
    if (!info.occ) {
        hipFuncAttributes a;
        int dev = 0, n = 0, cus = 0;
        if (hipFuncGetAttributes(&a, (const void*)kernel) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0) != hipSuccess || n < 1 || cus < 1) {
            (void)hipGetLastError();
            info.occ = -1;
        } else {
            info.occ = n; info.lds = (int)a.sharedSizeBytes; info.cus = cus;
        }
    }
    if (info.occ < 3 || !tun().occ_pad) return 0;
    long best = -1;
    int pick = info.occ;
    for (int o = info.occ; o >= 2; --o) {                   // ties go to the higher occupancy
        const long cost = ((nwg + (long)info.cus * o - 1) / ((long)info.cus * o)) * o;
        if (best < 0 || cost < best) { best = cost; pick = o; }
    }
    if (pick == info.occ) return 0;
    const int lds_total = 160 * 1024;
    const int need = lds_total / (pick + 1) + 512;          // more than a (pick+1)-th of the CU's LDS per workgroup
    return need > info.lds ? need - info.lds : 0;
}

template <int BM, int BN, int WAVES_CO, int BKT>
static void launch_conv_t(const GroupTable& tab, const float* wp, const float* bias, float* slab, int Cin, int Cout, int flags,
                          const ConvPlan& p, hipStream_t s) {
    dim3 grid((unsigned)tab.tile_start[tab.n], (unsigned)((Cout + BN - 1) / BN), (unsigned)p.S);
    const ConvVariant v = conv_variant(tab, p, BM, BN, BKT, Cin, Cout, flags);
    if (v.s3) {
        if constexpr (BN == 64 && ((BM != 256 && BKT == 32) || (BM == 256 && BKT == 16))) {
            constexpr int WCO3 = BM == 256 ? 1 : 2;
            const long nwg = (long)grid.x * grid.y * grid.z;
            int pad = 0;
            if constexpr (BM == 256) {
                if (strip3_db(tab, p)) {
                    T2V_LAUNCH_PROF((conv_igemm_strip3_kernel<256, 16, 1, true, true>), grid, dim3(256), 0, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
                    return;
                }
            }
            if constexpr (BM == 64) {
                // the double-buffered form (one barrier per round, staging between the MFMAs): every member three taps wide
                const int db = strip3_db(tab, p);
                const bool all3 = true;
                if (db == 32 && all3) {
                    T2V_LAUNCH_PROF((conv_igemm_strip3_kernel<64, 32, 2, true, true>), grid, dim3(256), 0, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
                    return;
                }
                if (db == 16 && all3) {
                    T2V_LAUNCH_PROF((conv_igemm_strip3_kernel<64, 16, 2, true, true>), grid, dim3(256), 0, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
                    return;
                }
            }
            if (p.vecb) {
                static OccInfo oi;
                if (BM == 256) pad = occupancy_pad(conv_igemm_strip3_kernel<BM, BKT, WCO3, true>, oi, nwg);
                T2V_LAUNCH_PROF((conv_igemm_strip3_kernel<BM, BKT, WCO3, true>), grid, dim3(256), pad, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
            } else {
                static OccInfo oi;
                if (BM == 256) pad = occupancy_pad(conv_igemm_strip3_kernel<BM, BKT, WCO3, false>, oi, nwg);
                T2V_LAUNCH_PROF((conv_igemm_strip3_kernel<BM, BKT, WCO3, false>), grid, dim3(256), pad, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
            }
        }
        return;
    }
    if (v.strip) {
        if constexpr (BN == 32) {
            if (p.vecb) T2V_LAUNCH_PROF((conv_igemm_strip_kernel<BM, BN, WAVES_CO, BKT, true>), grid, dim3(256), 0, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
            else T2V_LAUNCH_PROF((conv_igemm_strip_kernel<BM, BN, WAVES_CO, BKT, false>), grid, dim3(256), 0, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
        }
        return;
    }
    if (p.fast) {
        if (p.vecb) T2V_LAUNCH_PROF((conv_igemm_kernel<BM, BN, WAVES_CO, BKT, true, true>), grid, dim3(256), 0, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
        else T2V_LAUNCH_PROF((conv_igemm_kernel<BM, BN, WAVES_CO, BKT, true, false>), grid, dim3(256), 0, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
    } else {
        T2V_LAUNCH_PROF((conv_igemm_kernel<BM, BN, WAVES_CO, 16, false, false>), grid, dim3(256), 0, s, tab, wp, bias, slab, Cin, Cout, flags, p.S);
    }
}

// Launch-plan query (no launch): the kernel instantiation t2v_conv_fwd_grouped would run for these members.
// out[0] kind (0 implicit GEMM, 1 strip implicit GEMM, 2 thin conv (Cout <= 4), 3 thin linear, 4 thin two-pass),
// out[1..3] tile BM, BN and K chunk, out[4] FAST (Cin % 16 == 0), out[5] VECB (Cout % 4 == 0), out[6] KS, out[7] split-K S.
// Used by the parity tests to assert that every instantiation — in particular the ones the benchmark shapes select —
// is reached by at least one checked case.
static void fill_fwd_plan(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int flags, const GroupTable& tab,
                          const ConvPlan& p, int32_t* out) {
    for (int i = 0; i < 8; ++i) out[i] = 0;
    int nslots;
    bool pure_linear = Cout <= 4;
    for (int i = 0; i < ngroups; ++i)
        pure_linear = pure_linear && groups[i].D == 1 && groups[i].H == 1 && groups[i].W == 1 && groups[i].ntaps == 1 && groups[i].widx[0] == 0;
    if (pure_linear) { out[0] = 3; out[2] = Cout == 1 ? 1 : 4; out[7] = 1; return; }
    if (thin_ok(groups, ngroups, Cin, Cout, nslots)) {
        out[0] = thin_two_pass(groups, ngroups, Cin, Cout, nslots) ? 4 : 2;
        out[1] = 256; out[2] = Cout == 1 ? 1 : 4; out[7] = 1;
        return;
    }
    if (stem_ok(groups, ngroups, Cin, Cout, flags)) { out[0] = 12; out[1] = 256; out[2] = Cin; out[7] = 1; return; }      // conv_stem_kernel<Cin>
    const int bk = (p.bm == 256 && p.bn == 64) ? 16 : (p.bk == 32 ? 32 : 16);
    const ConvVariant v = conv_variant(tab, p, p.bm, p.bn, bk, Cin, Cout, flags);
    out[0] = v.s3 ? 5 : (v.strip ? 1 : 0);
    out[1] = p.bm; out[2] = p.bn; out[3] = p.fast ? bk : 16;
    out[4] = p.fast ? 1 : 0; out[5] = (p.fast && p.vecb) ? 1 : 0; out[6] = v.ks; out[7] = p.S;
    if (v.s3 && (bk == 32 || p.bm == 256)) {
        const int db = strip3_db(tab, p);
        if (db) { out[3] = db; out[6] = 2; }
    }
}
extern "C" int t2v_conv_fwd_plan(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int flags, int32_t* out) {
    GroupTable tab;
    ConvPlan p;
    if (!out || !build_table(groups, ngroups, Cin, Cout, false, tab, p)) return T2V_EINVAL;
    fill_fwd_plan(groups, ngroups, Cin, Cout, flags, tab, p, out);
    return T2V_OK;
}

extern "C" int64_t t2v_conv_fwd_grouped_ws_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout) {
    GroupTable tab;
    ConvPlan p;
    if (!build_table(groups, ngroups, Cin, Cout, false, tab, p)) return T2V_EINVAL;
    int nslots;
    if (thin_ok(groups, ngroups, Cin, Cout, nslots))
        return thin_two_pass(groups, ngroups, Cin, Cout, nslots) ? (int64_t)nslots * tab.out_start[ngroups] : 0;
    if (Cout <= 4) {
        bool pl = true;
        for (int i = 0; i < ngroups; ++i) pl = pl && groups[i].D == 1 && groups[i].H == 1 && groups[i].W == 1 && groups[i].ntaps == 1;
        if (pl) return 0;
    }
    return p.S > 1 ? (int64_t)p.S * tab.out_start[ngroups] : 0;
}

extern "C" int t2v_conv_fwd_grouped(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, const float* wp,
                                    const float* bias, float* ws, int flags, void* stream) {
    GroupTable tab;
    ConvPlan p;
    if (!wp || !build_table(groups, ngroups, Cin, Cout, true, tab, p)) return T2V_EINVAL;
    if (flags & T2V_CONV_MASK_OUT) {
        if (flags & T2V_CONV_ACCUM) return T2V_EINVAL;
        for (int i = 0; i < ngroups; ++i)
            if (!groups[i].mask) return T2V_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
#ifdef T2V_ABLATION
    if (const char* e = getenv("T2V_DEBUG_FLAGS")) flags |= atoi(e);     // developer ablations (wrong results)
#endif
    int nslots;
    const bool thin = thin_ok(groups, ngroups, Cin, Cout, nslots);
    if (!thin && p.S > 1 && !ws && !stem_ok(groups, ngroups, Cin, Cout, flags)) return T2V_EINVAL;
    if (p.dstride2) {                      // frame-strided outputs: the three-taps-per-round strip kernels only
        if (thin || (flags & T2V_CONV_ACCUM)) return T2V_EINVAL;
        bool s3 = false;
        if (p.bn == 64 && p.bm == 256) s3 = conv_variant(tab, p, 256, 64, 16, Cin, Cout, flags).s3;
        else if (p.bn == 64 && p.bk == 32) s3 = conv_variant(tab, p, p.bm, 64, 32, Cin, Cout, flags).s3;
        if (!s3) return T2V_EINVAL;
    }
    double flops = 0;
    for (int i = 0; i < ngroups; ++i)
        flops += 2.0 * (double)groups[i].N * out_frames(groups[i]) * groups[i].H * groups[i].W * Cout * Cin * groups[i].ntaps;
    long Mtot_ = 0;
    int taps_ = 0;
    for (int i = 0; i < ngroups; ++i) {
        Mtot_ += (long)groups[i].N * out_frames(groups[i]) * groups[i].H * groups[i].W;
        if (groups[i].ntaps > taps_) taps_ = groups[i].ntaps;
    }
    int32_t plan_[8];
    fill_fwd_plan(groups, ngroups, Cin, Cout, flags, tab, p, plan_);
    bool pure_linear = Cout <= 4;
    for (int i = 0; i < ngroups; ++i)
        pure_linear = pure_linear && groups[i].D == 1 && groups[i].H == 1 && groups[i].W == 1 && groups[i].ntaps == 1 && groups[i].widx[0] == 0;
    if (pure_linear) {
        long rows = 0;
        for (int i = 0; i < ngroups; ++i) { tab.tile_start[i] = (int32_t)rows; rows += groups[i].N; }
        for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) tab.tile_start[i] = (int32_t)rows;
        ProfScope prof(3, flops, s, Mtot_, Cin, Cout, 1, ngroups, 1);
        ProfScope::set_plan(plan_, 8);
        if (Cout == 1) T2V_LAUNCH_PROF(linear_thin_kernel<1>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, tab, wp, bias, Cin, Cout, flags);
        else T2V_LAUNCH_PROF(linear_thin_kernel<4>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, tab, wp, bias, Cin, Cout, flags);
        return launch_status();
    }
    if (thin) {
        // re-tile for 256 voxels per workgroup
        long mt = 0;
        for (int i = 0; i < ngroups; ++i) {
            tab.tile_start[i] = (int32_t)mt;
            mt += ((long)groups[i].N * groups[i].D * groups[i].H * groups[i].W + 255) / 256;
        }
        for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) tab.tile_start[i] = (int32_t)mt;
        if (ws && thin_two_pass(groups, ngroups, Cin, Cout, nslots)) {
            {
                ProfScope prof(3, flops, s, Mtot_, Cin, Cout, taps_, ngroups, 1);
                ProfScope::set_plan(plan_, 8);
                if (Cin == 64) T2V_LAUNCH_PROF(thin_taps_mfma_kernel<32>, dim3((unsigned)mt), dim3(512), 0, s, tab, wp, ws, nslots, flags);
                else if (Cin == 32) T2V_LAUNCH_PROF(thin_taps_mfma_kernel<16>, dim3((unsigned)mt), dim3(512), 0, s, tab, wp, ws, nslots, flags);
                else if (Cin == 16) T2V_LAUNCH_PROF(thin_taps_mfma_kernel<8>, dim3((unsigned)mt), dim3(512), 0, s, tab, wp, ws, nslots, flags);
                else if (Cin == 128) T2V_LAUNCH_PROF(thin_taps_mfma_kernel<64>, dim3((unsigned)mt), dim3(512), 0, s, tab, wp, ws, nslots, flags);
                else T2V_LAUNCH_PROF(thin_taps_kernel, dim3((unsigned)mt), dim3(256), 0, s, tab, wp, ws, Cin, nslots, flags);
            }
            ProfScope prof2(3, 0.0, s, Mtot_, Cin, Cout, taps_, ngroups, 2);
            T2V_LAUNCH_PROF(thin_shift_sum_kernel, dim3((unsigned)mt), dim3(256), 0, s, tab, ws, bias, nslots, flags);
            return launch_status();
        }
        ProfScope prof(3, flops, s, Mtot_, Cin, Cout, taps_, ngroups, 1);
        ProfScope::set_plan(plan_, 8);
        if (Cout == 1) T2V_LAUNCH_PROF(conv_thin_kernel<1>, dim3((unsigned)mt), dim3(256), 0, s, tab, wp, bias, Cin, Cout, nslots, flags);
        else T2V_LAUNCH_PROF(conv_thin_kernel<4>, dim3((unsigned)mt), dim3(256), 0, s, tab, wp, bias, Cin, Cout, nslots, flags);
        return launch_status();
    }
    if (stem_ok(groups, ngroups, Cin, Cout, flags)) {
        long mt = 0;
        for (int i = 0; i < ngroups; ++i) {
            tab.tile_start[i] = (int32_t)mt;
            mt += ((long)groups[i].N * groups[i].D * groups[i].H * groups[i].W + 255) / 256;
        }
        for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) tab.tile_start[i] = (int32_t)mt;
        ProfScope prof(3, flops, s, Mtot_, Cin, Cout, taps_, ngroups, 1);
        ProfScope::set_plan(plan_, 8);
        if (Cin == 1 && (Cout % 32) == 0) T2V_LAUNCH_PROF(conv_stem_mfma_kernel, dim3((unsigned)mt), dim3(512), 0, s, tab, wp, bias, Cout, flags);
        else if (Cin == 1) T2V_LAUNCH_PROF(conv_stem_kernel<1>, dim3((unsigned)mt), dim3(256), 0, s, tab, wp, bias, Cout, flags);
        else T2V_LAUNCH_PROF(conv_stem_kernel<3>, dim3((unsigned)mt), dim3(256), 0, s, tab, wp, bias, Cout, flags);
        return launch_status();
    }
    ProfScope prof(0, flops, s, Mtot_, Cin, Cout, taps_, ngroups, p.S);      // executed (non-padding-tap) MACs x 2
    ProfScope::set_plan(plan_, 8);
    const int bk = p.bk;
    if (p.bn == 32) {
        if (bk == 32) launch_conv_t<128, 32, 1, 32>(tab, wp, bias, ws, Cin, Cout, flags, p, s);
        else launch_conv_t<128, 32, 1, 16>(tab, wp, bias, ws, Cin, Cout, flags, p, s);
    } else if (p.bm == 256) {
        launch_conv_t<256, 64, 1, 16>(tab, wp, bias, ws, Cin, Cout, flags, p, s);
    } else if (p.bm == 128) {
        if (bk == 32) launch_conv_t<128, 64, 2, 32>(tab, wp, bias, ws, Cin, Cout, flags, p, s);
        else launch_conv_t<128, 64, 2, 16>(tab, wp, bias, ws, Cin, Cout, flags, p, s);
    } else {
        if (bk == 32) launch_conv_t<64, 64, 2, 32>(tab, wp, bias, ws, Cin, Cout, flags, p, s);
        else launch_conv_t<64, 64, 2, 16>(tab, wp, bias, ws, Cin, Cout, flags, p, s);
    }
    int st = launch_status();
    if (st) return st;
    if (p.S > 1) {
        long blocks = (tab.out_start[ngroups] + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        ProfScope prof_r(4, 0.0, s, Mtot_, Cin, Cout, taps_, ngroups, p.S);     // the split-K pass, timed on its own
        T2V_LAUNCH_PROF(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tab, ws, bias, p.S, Cout, flags);
    }
    return launch_status();
}

// bf16-compute forward / data gradient: same contract as t2v_conv_fwd_grouped with wpb from t2v_pack_weight_bf16.
// Returns T2V_EINVAL for shapes the bf16 kernel does not take (Cin % 32, thin outputs): the caller uses the fp32 entry point.
extern "C" int t2v_conv_fwd_grouped_bf16_ok(const t2v_conv_group* groups, int ngroups, int Cin, int Cout) {
    GroupTable tab;
    ConvPlan p;
    int nslots;
    if (!build_table(groups, ngroups, Cin, Cout, false, tab, p)) return 0;
    if ((Cin % 32) != 0 || Cout <= 4 || thin_ok(groups, ngroups, Cin, Cout, nslots)) return 0;
    return 1;
}
// Which bf16 instantiation a launch lands on (shared by the launcher and by t2v_conv_fwd_bf16_plan): same tiling decisions as
// the fp32 path, restricted to the tiles the bf16 kernels have (128 x 64, 64 x 64) — re-tiled when build_table picked 128 x 32
// (Cout <= 32) or 256 x 64. kind: 6 plain implicit GEMM (1-wide kernels, W = 1 members, tensors beyond 32-bit byte offsets),
// 8 strip3 (three dx taps per barrier round).
struct Bf16Variant { int kind; int bm; };
static bool bf16_variant(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int flags, GroupTable& tab, ConvPlan& p, Bf16Variant& v) {
    if (p.bn != 64 || p.bm == 256) {
        long mt = 0;
        p.bn = 64;
        p.bm = 128;
        for (int i = 0; i < ngroups; ++i) {
            const long M = (long)groups[i].N * out_frames(groups[i]) * groups[i].H * groups[i].W;
            tab.tile_start[i] = (int32_t)mt;
            mt += (M + p.bm - 1) / p.bm;
        }
        for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) tab.tile_start[i] = (int32_t)mt;
    }
    const long wbytes = 2L * T2V_MAX_TAPS * Cout * Cin;                            // (32-bit byte offsets into the packed weight)
    const bool s3_b16 = tun().strip && strip_ok(tab) && strip_fits32(tab, Cin) && wbytes < (1L << 31);
    if (p.dstride2 && (!s3_b16 || (flags & T2V_CONV_ACCUM))) return false;        // frame-strided members: strip3 form only
    v.bm = p.bm == 128 ? 128 : 64;
    v.kind = s3_b16 ? 8 : 6;
    return true;
}
// Launch-plan query of the bf16-compute entry point (no launch; host arithmetic): out[0] kind (6 conv_igemm_bf16_kernel,
// 8 conv_igemm_bf16_strip3_kernel), out[1] BM (128 / 64), out[2] BN = 64, out[3] K chunk = 32, out[4] 1 for frame-strided members
// (dstride / ydstride), out[7] split-K S; the rest 0. T2V_EINVAL where t2v_conv_fwd_grouped_bf16 would refuse the launch (the caller then uses fp32).
extern "C" int t2v_conv_fwd_bf16_plan(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int flags, int32_t* out) {
    GroupTable tab;
    ConvPlan p;
    Bf16Variant v;
    if (!out || !t2v_conv_fwd_grouped_bf16_ok(groups, ngroups, Cin, Cout) || !build_table(groups, ngroups, Cin, Cout, false, tab, p) ||
        !bf16_variant(groups, ngroups, Cin, Cout, flags, tab, p, v))
        return T2V_EINVAL;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    out[0] = v.kind; out[1] = v.bm; out[2] = 64; out[3] = 32; out[4] = p.dstride2 ? 1 : 0; out[7] = p.S;
    return T2V_OK;
}
extern "C" int t2v_conv_fwd_grouped_bf16(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, const void* wpb,
                                         const float* bias, float* ws, int flags, void* stream) {
    GroupTable tab;
    ConvPlan p;
    if (!wpb || !t2v_conv_fwd_grouped_bf16_ok(groups, ngroups, Cin, Cout) || !build_table(groups, ngroups, Cin, Cout, true, tab, p))
        return T2V_EINVAL;
    if (flags & T2V_CONV_MASK_OUT) {
        if (flags & T2V_CONV_ACCUM) return T2V_EINVAL;
        for (int i = 0; i < ngroups; ++i)
            if (!groups[i].mask) return T2V_EINVAL;
    }
    Bf16Variant v;
    if (!bf16_variant(groups, ngroups, Cin, Cout, flags, tab, p, v)) return T2V_EINVAL;
    if (p.S > 1 && !ws) return T2V_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    double flops = 0;
    long Mtot_ = 0;
    int taps_ = 0;
    for (int i = 0; i < ngroups; ++i) {
        const long M = (long)groups[i].N * out_frames(groups[i]) * groups[i].H * groups[i].W;
        flops += 2.0 * (double)M * Cout * Cin * groups[i].ntaps;
        Mtot_ += M;
        if (groups[i].ntaps > taps_) taps_ = groups[i].ntaps;
    }
    {
        ProfScope prof(5, flops, s, Mtot_, Cin, Cout, taps_, ngroups, p.S);
        int32_t plan_[8] = {v.kind, v.bm, 64, 32, p.dstride2 ? 1 : 0, 0, 0, p.S};
        ProfScope::set_plan(plan_, 8);
        dim3 grid((unsigned)tab.tile_start[tab.n], (unsigned)((Cout + 63) / 64), (unsigned)p.S);
        if (v.kind == 8) {
            if (v.bm == 128) T2V_LAUNCH_PROF(conv_igemm_bf16_strip3_kernel<128>, grid, dim3(256), 0, s, tab, (const __bf16*)wpb, bias, ws, Cin, Cout, flags, p.S);
            else T2V_LAUNCH_PROF(conv_igemm_bf16_strip3_kernel<64>, grid, dim3(256), 0, s, tab, (const __bf16*)wpb, bias, ws, Cin, Cout, flags, p.S);
        } else if (v.bm == 128) T2V_LAUNCH_PROF(conv_igemm_bf16_kernel<128>, grid, dim3(256), 0, s, tab, (const __bf16*)wpb, bias, ws, Cin, Cout, flags, p.S);
        else T2V_LAUNCH_PROF(conv_igemm_bf16_kernel<64>, grid, dim3(256), 0, s, tab, (const __bf16*)wpb, bias, ws, Cin, Cout, flags, p.S);
    }
    int st = launch_status();
    if (st) return st;
    if (p.S > 1) {
        long blocks = (tab.out_start[ngroups] + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        ProfScope prof_r(4, 0.0, s, Mtot_, Cin, Cout, taps_, ngroups, p.S);
        T2V_LAUNCH_PROF(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tab, ws, bias, p.S, Cout, flags);
    }
    return launch_status();
}

// single-tensor entry points: a group of one whose tap j sits in packed slot j
static bool geom_ok(const t2v_conv_geom* g) {
    if (!g) return false;
    if (g->N < 1 || g->Cin < 1 || g->Cout < 1 || g->D < 1 || g->H < 1 || g->W < 1) return false;
    if (g->ntaps < 1 || g->ntaps > T2V_MAX_TAPS) return false;
    return true;
}
static t2v_conv_group group_of(const t2v_conv_geom* g, const float* x, float* y) {
    t2v_conv_group q;
    q.x = x; q.y = y; q.mask = nullptr;
    q.N = g->N; q.D = g->D; q.H = g->H; q.W = g->W; q.ntaps = g->ntaps; q.dstride = 0; q.ydstride = 0; q.yoff = 0; q.Dy = 0;
    for (int t = 0; t < T2V_MAX_TAPS; ++t) {
        q.dz[t] = t < g->ntaps ? g->dz[t] : 0; q.dy[t] = t < g->ntaps ? g->dy[t] : 0; q.dx[t] = t < g->ntaps ? g->dx[t] : 0;
        q.widx[t] = (int8_t)(t < g->ntaps ? t : 0);
    }
    return q;
}
extern "C" int64_t t2v_conv_fwd_ws_floats(const t2v_conv_geom* g) {
    if (!geom_ok(g)) return T2V_EINVAL;
    t2v_conv_group q = group_of(g, nullptr, nullptr);
    return t2v_conv_fwd_grouped_ws_floats(&q, 1, g->Cin, g->Cout);
}
extern "C" int t2v_conv_fwd(const float* x, const float* wp, const float* bias, float* y, float* ws, const t2v_conv_geom* g,
                            int flags, void* stream) {
    if (!x || !wp || !y || !geom_ok(g)) return T2V_EINVAL;
    t2v_conv_group q = group_of(g, x, y);
    return t2v_conv_fwd_grouped(&q, 1, g->Cin, g->Cout, wp, bias, ws, flags, stream);
}

// ------------------------------------------------------------------------------------------------
// weight gradient (GROUPED): dW[co][ci][t] = sum over all groups and their voxels of gy * x_shift.
// Per (original tap t, co-tile, ci-tile, k-split) a 64x64 tile of dW over a range of 32-voxel chunks of
// the concatenated chunk list of all groups; groups for which tap t only touches padding are skipped.
// ------------------------------------------------------------------------------------------------
#define WG_BK 32
#define WG_PITCH (WG_BK + 1)   // odd pitch: conflict-free column reads (ds_read_b32); b128 reads measured 5 % slower

struct WGroupTable {
    t2v_conv_group g[T2V_MAX_GROUPS];          // x = layer input, y = dL/dy (read only here)
    int32_t chunk_start[T2V_MAX_GROUPS + 1];   // prefix sum of ceil(M_g / 32)
    int32_t n;
};

struct LiveTaps { int8_t t[T2V_MAX_TAPS]; int32_t n; };   // slab slot j -> original tap index

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WGroupTable tab, float* __restrict__ slab, const int Cin,
                                                         const int Cout, const int T, const int kH, const int kW,
                                                         const int flags, const int chunks_per_split, const LiveTaps live,
                                                         float* __restrict__ bias_slab) {
    __shared__ __attribute__((aligned(16))) float As[64 * WG_PITCH];   // gy^T tile  [co][m]
    __shared__ __attribute__((aligned(16))) float Bs[64 * WG_PITCH];   // x   tile   [ci][m]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave & 1, wci = wave >> 1;
    const int nco_t = (Cout + 63) / 64;
    const int co0 = (blockIdx.x % nco_t) * 64, ci0 = (blockIdx.x / nco_t) * 64;
    // (slot, split) from an XCD-aware linear id: the taps of one k-split (same x / gy voxels) share an L2
    const int lin = xcd_remap((int)(blockIdx.y + blockIdx.z * gridDim.y), (int)(gridDim.y * gridDim.z));
    const int slot = lin % (int)gridDim.y;
    const int split = lin / (int)gridDim.y;
    const int t = live.t[slot];         // ORIGINAL tap index of slab slot
    const bool relu_in = flags & T2V_CONV_RELU_IN;
    // offset of original tap t (k in {1,3} per dim, centred)
    const int kD = T / (kH * kW);
    const int ta = t / (kH * kW), tb = (t / kW) % kH, tc = t % kW;
    const int dz = ta - kD / 2, dy = tb - kH / 2, dx = tc - kW / 2;

    const int ml = tid & 31, rl = tid >> 5;     // 32 m x 8 rows per pass, 8 passes -> 64 rows
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nchunks = tab.chunk_start[tab.n];
    const int q0 = split * chunks_per_split;
    int q1 = q0 + chunks_per_split;
    if (q1 > nchunks) q1 = nchunks;

    float ra[8], rb[8];
    bool pend_v = false;
    // bias gradient on the side (see conv_wgrad3_kernel): the centre tap's workgroups of the first channel tile see every
    // voxel of every member exactly once
    const bool do_bias = bias_slab != nullptr && ci0 == 0 && t == T / 2;
    bool pend_mv = false;
    float bsum[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) bsum[p] = 0.f;
    // member state cached in scalars + shift decode + buffer loads with scalar row strides: see conv_wgrad3_kernel (this kernel
    // runs only 16 MFMAs per wave and chunk, so the ~400 vector instructions of the old per-chunk decode set its pace)
    const bool full = co0 + 64 <= Cout && ci0 + 64 <= Cin;
    int g_i = -1, g_begin = 0, g_end = 0;
    int gD = 1, gH = 1, gW = 1, gHW = 1, gDHW = 1, gM = 0, g_lw = 0, g_lhw = 0, g_shift = 0;
    bool g_pow2 = false, g_live = false;
    __amdgpu_buffer_rsrc_t g_x = __builtin_amdgcn_make_buffer_rsrc((void*)tab.g[0].x, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t g_y = g_x;
    auto enter_group = [&](int q) {
        int gi = 0;
#pragma unroll
        for (int k = 1; k < T2V_MAX_GROUPS; ++k)
            if (k < tab.n && q >= tab.chunk_start[k]) gi = k;
        const t2v_conv_group& gd = tab.g[gi];
        g_i = gi;
        g_begin = tab.chunk_start[gi];
        g_end = tab.chunk_start[gi + 1];
        gD = gd.D; gH = gd.H; gW = gd.W; gHW = gH * gW; gDHW = gD * gHW; gM = gd.N * gDHW;
        g_pow2 = ((gD & (gD - 1)) | (gH & (gH - 1)) | (gW & (gW - 1))) == 0;
        g_lw = __builtin_ctz(gW);
        g_lhw = g_lw + __builtin_ctz(gH);
        g_shift = dz * gHW + dy * gW + dx;
        // a dim of extent 1 keeps its centre tap only (same rule as the forward)
        g_live = !((gD == 1 && dz) || (gH == 1 && dy) || (gW == 1 && dx));
        g_x = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)gM * (uint32_t)Cin * 4u), 0x00020000);      // (host: M * C < 2^30)
        g_y = __builtin_amdgcn_make_buffer_rsrc((void*)gd.y, 0, (int)((uint32_t)gM * (uint32_t)Cout * 4u), 0x00020000);
    };
    auto load_chunk = [&](int q) {
        if (q >= g_end || g_i < 0) enter_group(q);
        const int D = gD, H = gH, W = gW, HW = gHW, DHW = gDHW;
        const int m = (q - g_begin) * WG_BK + ml;
        const bool mv = g_live && m < gM;
        int sp, d, h, w_;
        if (g_pow2) {
            sp = m & (DHW - 1);
            d = sp >> g_lhw;
            h = (sp >> g_lw) & (H - 1);
            w_ = sp & (W - 1);
        } else {
            const bool small = gM < (1 << 24);
            const int n = small ? fast_div(m, DHW, 1.0f / (float)DHW) : m / DHW;
            sp = m - n * DHW;
            d = small ? fast_div(sp, HW, 1.0f / (float)HW) : sp / HW;
            const int r = sp - d * HW;
            h = small ? fast_div(r, W, 1.0f / (float)W) : r / W;
            w_ = r - h * W;
        }
        const bool xv = mv && (unsigned)(d + dz) < (unsigned)D && (unsigned)(h + dy) < (unsigned)H && (unsigned)(w_ + dx) < (unsigned)W;
        const uint32_t gbase = mv ? (uint32_t)(m - sp) * (uint32_t)Cout + (uint32_t)sp : 0u;       // clamped: voxel 0 of sample 0
        const uint32_t xb = mv ? (uint32_t)(m - sp) * (uint32_t)Cin + (uint32_t)(sp + (xv ? g_shift : 0)) : 0u;
        const uint32_t uDHW = (uint32_t)DHW;
        if (full) {
            const uint32_t oa = (gbase + (uint32_t)(co0 + rl) * uDHW) * 4u, ob = (xb + (uint32_t)(ci0 + rl) * uDHW) * 4u;
            const int st = 32 * DHW;
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                ra[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_y, oa, p * st, 0));
                rb[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_x, ob, p * st, 0));
            }
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int co = co0 + rl + p * 8, ci = ci0 + rl + p * 8;
                ra[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    g_y, (gbase + (uint32_t)(co < Cout ? co : Cout - 1) * uDHW) * 4u, 0, 0));
                rb[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    g_x, (xb + (uint32_t)(ci < Cin ? ci : Cin - 1) * uDHW) * 4u, 0, 0));
            }
        }
        pend_v = xv;
        pend_mv = mv;
    };

    if (q0 < q1) load_chunk(q0);
    for (int q = q0; q < q1; ++q) {
        if (do_bias) {
#pragma unroll
            for (int p = 0; p < 8; ++p) bsum[p] += pend_mv ? ra[p] : 0.f;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {          // masking + fused ReLU at the LDS write, one chunk after the loads
            const int co = co0 + rl + p * 8, ci = ci0 + rl + p * 8;
            As[(rl + p * 8) * WG_PITCH + ml] = (pend_v && co < Cout) ? ra[p] : 0.f;
            const float v = (pend_v && ci < Cin) ? rb[p] : 0.f;
            Bs[(rl + p * 8) * WG_PITCH + ml] = relu_in ? fmaxf(v, 0.f) : v;
        }
        __syncthreads();
        if (q + 1 < q1) load_chunk(q + 1);
#pragma unroll
        for (int k2 = 0; k2 < WG_BK / 2; ++k2) {
            const int kc = k2 * 2 + hi;
            float a = As[(wco * 32 + l31) * WG_PITCH + kc];
            float b = Bs[(wci * 32 + l31) * WG_PITCH + kc];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (do_bias) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float v = bsum[p];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int co = co0 + rl + p * 8;
            if (ml == 0 && co < Cout) bias_slab[(size_t)split * Cout + co] = v;
        }
    }
    // slab[((split*nlive + slot)*Cout + co)*Cin + ci]; rows = co (registers), cols = ci (lanes)
    const int ci = ci0 + wci * 32 + l31;
    if (ci < Cin) {
        float* ps = slab + ((size_t)split * live.n + slot) * Cout * Cin + ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < Cout) ps[(size_t)co * Cin] = acc[r];
        }
    }
}


// bf16-compute variant of conv_wgrad_kernel (opt-in mode): tiles rounded to bf16 at the LDS write, v_mfma_f32_32x32x16_bf16.
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(const WGroupTable tab, float* __restrict__ slab, const int Cin,
                                                         const int Cout, const int T, const int kH, const int kW,
                                                         const int flags, const int chunks_per_split, const LiveTaps live,
                                                         float* __restrict__ bias_slab) {
    __shared__ __attribute__((aligned(16))) __bf16 As[64 * B16_KP];    // gy^T tile  [co][m], m contiguous
    __shared__ __attribute__((aligned(16))) __bf16 Bs[64 * B16_KP];    // x   tile   [ci][m]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave & 1, wci = wave >> 1;
    const int nco_t = (Cout + 63) / 64;
    const int co0 = (blockIdx.x % nco_t) * 64, ci0 = (blockIdx.x / nco_t) * 64;
    // (slot, split) from an XCD-aware linear id: the taps of one k-split (same x / gy voxels) share an L2
    const int lin = xcd_remap((int)(blockIdx.y + blockIdx.z * gridDim.y), (int)(gridDim.y * gridDim.z));
    const int slot = lin % (int)gridDim.y;
    const int split = lin / (int)gridDim.y;
    const int t = live.t[slot];         // ORIGINAL tap index of slab slot
    const bool relu_in = flags & T2V_CONV_RELU_IN;
    // offset of original tap t (k in {1,3} per dim, centred)
    const int kD = T / (kH * kW);
    const int ta = t / (kH * kW), tb = (t / kW) % kH, tc = t % kW;
    const int dz = ta - kD / 2, dy = tb - kH / 2, dx = tc - kW / 2;

    const int ml = tid & 31, rl = tid >> 5;     // 32 m x 8 rows per pass, 8 passes -> 64 rows
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nchunks = tab.chunk_start[tab.n];
    const int q0 = split * chunks_per_split;
    int q1 = q0 + chunks_per_split;
    if (q1 > nchunks) q1 = nchunks;

    float ra[8], rb[8];
    bool pend_v = false;
    // bias gradient on the side (see conv_wgrad3_kernel): the centre tap's workgroups of the first channel tile see every
    // voxel of every member exactly once
    const bool do_bias = bias_slab != nullptr && ci0 == 0 && t == T / 2;
    bool pend_mv = false;
    float bsum[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) bsum[p] = 0.f;
    const bool dbg_noload = flags & 64, dbg_nostage = flags & 128;
    auto load_chunk = [&](int q) {
        if (dbg_noload && q != q0) return;
        int gi = 0;
#pragma unroll
        for (int k = 1; k < T2V_MAX_GROUPS; ++k)
            if (k < tab.n && q >= tab.chunk_start[k]) gi = k;
        const t2v_conv_group& gd = tab.g[gi];
        const int D = gd.D, H = gd.H, W = gd.W, HW = H * W, DHW = D * HW, M = gd.N * DHW;
        // a dim of extent 1 keeps its centre tap only (same rule as the forward)
        const bool tap_live = !((D == 1 && dz) || (H == 1 && dy) || (W == 1 && dx));
        const int m = (q - tab.chunk_start[gi]) * WG_BK + ml;
        const bool mv = tap_live && m < M;
        bool xv = false;
        size_t gbase = 0, xb = 0;      // clamped: voxel 0 of sample 0 when this lane has nothing to load
        if (mv) {
            const bool small = M < (1 << 24);
            int n = small ? fast_div(m, DHW, 1.0f / (float)DHW) : m / DHW;
            int sp = m - n * DHW;
            int d = small ? fast_div(sp, HW, 1.0f / (float)HW) : sp / HW;
            int r = sp - d * HW;
            int h = small ? fast_div(r, W, 1.0f / (float)W) : r / W;
            int w_ = r - h * W;
            gbase = (size_t)n * Cout * DHW + sp;
            int dd = d + dz, hh = h + dy, ww = w_ + dx;
            xv = (unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            xb = (size_t)n * Cin * DHW + sp + (xv ? (ptrdiff_t)(dz * HW + dy * W + dx) : 0);
        }
        const float* __restrict__ gy = gd.y;
        const float* __restrict__ x = gd.x;
        // unconditional loads from clamped addresses, masked afterwards (no branch + wait per element)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int co = co0 + rl + p * 8, ci = ci0 + rl + p * 8;
            ra[p] = gy[gbase + (size_t)(co < Cout ? co : Cout - 1) * DHW];
            rb[p] = x[xb + (size_t)(ci < Cin ? ci : Cin - 1) * DHW];
        }
        pend_v = xv;
        pend_mv = mv;
    };

    if (q0 < q1) load_chunk(q0);
    for (int q = q0; q < q1; ++q) {
        if (do_bias) {
#pragma unroll
            for (int p = 0; p < 8; ++p) bsum[p] += pend_mv ? ra[p] : 0.f;
        }
        if (!(dbg_nostage && q != q0))
#pragma unroll
        for (int p = 0; p < 8; ++p) {          // masking + fused ReLU at the LDS write, one chunk after the loads
            const int co = co0 + rl + p * 8, ci = ci0 + rl + p * 8;
            As[(rl + p * 8) * B16_KP + ml] = (__bf16)((pend_v && co < Cout) ? ra[p] : 0.f);
            const float v = (pend_v && ci < Cin) ? rb[p] : 0.f;
            Bs[(rl + p * 8) * B16_KP + ml] = (__bf16)(relu_in ? fmaxf(v, 0.f) : v);
        }
        __syncthreads();
        if (q + 1 < q1) load_chunk(q + 1);
#pragma unroll
        for (int ks = 0; ks < WG_BK / 16; ++ks) {
            const int kc = ks * 16 + 8 * hi;
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(As + (wco * 32 + l31) * B16_KP + kc);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + (wci * 32 + l31) * B16_KP + kc);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (do_bias) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float v = bsum[p];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int co = co0 + rl + p * 8;
            if (ml == 0 && co < Cout) bias_slab[(size_t)split * Cout + co] = v;
        }
    }
    // slab[((split*nlive + slot)*Cout + co)*Cin + ci]; rows = co (registers), cols = ci (lanes)
    const int ci = ci0 + wci * 32 + l31;
    if (ci < Cin) {
        float* ps = slab + ((size_t)split * live.n + slot) * Cout * Cin + ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < Cout) ps[(size_t)co * Cin] = acc[r];
        }
    }
}



// Weight gradient for Cin < 64: the 64 tile columns run over (tap, ci) pairs instead of one tap's channels
// (C -> 64 stem convs: 27 columns instead of 27 tiles that are 1/64 full).
__global__ __launch_bounds__(256) void conv_wgrad_cols_kernel(const WGroupTable tab, float* __restrict__ slab, const int Cin,
                                                              const int Cout, const int T, const int kH, const int kW,
                                                              const int flags, const int chunks_per_split, const LiveTaps live,
                                                              float* __restrict__ bias_slab) {
    __shared__ __attribute__((aligned(16))) float As[64 * WG_PITCH];   // gy^T tile  [co][m]
    __shared__ __attribute__((aligned(16))) float Bs[64 * WG_PITCH];   // x   tile   [col][m]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave & 1, wcl = wave >> 1;
    const int nco_t = (Cout + 63) / 64;
    const int co0 = (blockIdx.x % nco_t) * 64, col0 = (blockIdx.x / nco_t) * 64;
    const int ncols = live.n * Cin;
    const int split = blockIdx.z;
    const bool relu_in = flags & T2V_CONV_RELU_IN;
    const int kD = T / (kH * kW);
    const int ml = tid & 31, rl = tid >> 5;
    // this thread's 8 columns: (slot, ci, dz, dy, dx)
    int c_ci[8], c_dz[8], c_dy[8], c_dx[8];
    bool c_ok[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int col = col0 + rl + p * 8;
        c_ok[p] = col < ncols;
        const int slot = c_ok[p] ? col / Cin : 0;
        c_ci[p] = c_ok[p] ? col - slot * Cin : 0;
        const int t = live.t[slot];
        c_dz[p] = t / (kH * kW) - kD / 2;
        c_dy[p] = (t / kW) % kH - kH / 2;
        c_dx[p] = t % kW - kW / 2;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunks = tab.chunk_start[tab.n];
    const int q0 = split * chunks_per_split;
    int q1 = q0 + chunks_per_split;
    if (q1 > nchunks) q1 = nchunks;
    float ra[8], rb[8];
    bool pend_m = false;
    // bias gradient on the side (see conv_wgrad3_kernel): the first column tile's workgroups see every voxel exactly once
    const bool do_bias = bias_slab != nullptr && col0 == 0;
    float bsum[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) bsum[p] = 0.f;
    uint32_t pend_x = 0;
    // member state cached in scalars + shift decode + buffer loads (see conv_wgrad3_kernel)
    int g_i = -1, g_begin = 0, g_end = 0;
    int gD = 1, gH = 1, gW = 1, gHW = 1, gDHW = 1, gM = 0, g_lw = 0, g_lhw = 0;
    bool g_pow2 = false;
    int c_off[8];                      // per column: x offset of its tap inside the member, -1 << 30 when the tap is dead there
    __amdgpu_buffer_rsrc_t g_x = __builtin_amdgcn_make_buffer_rsrc((void*)tab.g[0].x, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t g_y = g_x;
    bool c_live[8];
    auto enter_group = [&](int q) {
        int gi = 0;
#pragma unroll
        for (int k = 1; k < T2V_MAX_GROUPS; ++k)
            if (k < tab.n && q >= tab.chunk_start[k]) gi = k;
        const t2v_conv_group& gd = tab.g[gi];
        g_i = gi;
        g_begin = tab.chunk_start[gi];
        g_end = tab.chunk_start[gi + 1];
        gD = gd.D; gH = gd.H; gW = gd.W; gHW = gH * gW; gDHW = gD * gHW; gM = gd.N * gDHW;
        g_pow2 = ((gD & (gD - 1)) | (gH & (gH - 1)) | (gW & (gW - 1))) == 0;
        g_lw = __builtin_ctz(gW);
        g_lhw = g_lw + __builtin_ctz(gH);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            c_live[p] = c_ok[p] && !((gD == 1 && c_dz[p]) || (gH == 1 && c_dy[p]) || (gW == 1 && c_dx[p]));
            c_off[p] = c_dz[p] * gHW + c_dy[p] * gW + c_dx[p];
        }
        g_x = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)gM * (uint32_t)Cin * 4u), 0x00020000);      // (host: M * C < 2^30)
        g_y = __builtin_amdgcn_make_buffer_rsrc((void*)gd.y, 0, (int)((uint32_t)gM * (uint32_t)Cout * 4u), 0x00020000);
    };
    auto load_chunk = [&](int q) {
        if (q >= g_end || g_i < 0) enter_group(q);
        const int D = gD, H = gH, W = gW, HW = gHW, DHW = gDHW;
        const int m = (q - g_begin) * WG_BK + ml;
        const bool mv = m < gM;
        int sp = 0, d = 0, h = 0, w_ = 0;
        if (g_pow2) {
            sp = m & (DHW - 1);
            d = sp >> g_lhw;
            h = (sp >> g_lw) & (H - 1);
            w_ = sp & (W - 1);
        } else {
            const bool small = gM < (1 << 24);
            const int n = small ? fast_div(m, DHW, 1.0f / (float)DHW) : m / DHW;
            sp = m - n * DHW;
            d = small ? fast_div(sp, HW, 1.0f / (float)HW) : sp / HW;
            const int r = sp - d * HW;
            h = small ? fast_div(r, W, 1.0f / (float)W) : r / W;
            w_ = r - h * W;
        }
        const uint32_t uDHW = (uint32_t)DHW;
        const uint32_t gbase = mv ? (uint32_t)(m - sp) * (uint32_t)Cout + (uint32_t)sp : 0u;
        const uint32_t xb = mv ? (uint32_t)(m - sp) * (uint32_t)Cin + (uint32_t)sp : 0u;
        bool xv[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {          // unconditional loads from clamped addresses
            const int co = co0 + rl + p * 8;
            ra[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                g_y, (gbase + (uint32_t)(co < Cout ? co : Cout - 1) * uDHW) * 4u, 0, 0));
            const int dd = d + c_dz[p], hh = h + c_dy[p], ww = w_ + c_dx[p];
            xv[p] = mv && c_live[p] && (unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            rb[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                g_x, (xb + (uint32_t)c_ci[p] * uDHW + (uint32_t)(xv[p] ? c_off[p] : 0)) * 4u, 0, 0));
        }
        pend_m = mv;
        pend_x = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) pend_x |= (xv[p] ? 1u : 0u) << p;
    };
    if (q0 < q1) load_chunk(q0);
    for (int q = q0; q < q1; ++q) {
        if (do_bias) {
#pragma unroll
            for (int p = 0; p < 8; ++p) bsum[p] += pend_m ? ra[p] : 0.f;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int co = co0 + rl + p * 8;
            As[(rl + p * 8) * WG_PITCH + ml] = (pend_m && co < Cout) ? ra[p] : 0.f;
            const float v = ((pend_x >> p) & 1u) ? rb[p] : 0.f;
            Bs[(rl + p * 8) * WG_PITCH + ml] = relu_in ? fmaxf(v, 0.f) : v;
        }
        __syncthreads();
        if (q + 1 < q1) load_chunk(q + 1);
#pragma unroll
        for (int k2 = 0; k2 < WG_BK / 2; ++k2) {
            const int kc = k2 * 2 + hi;
            float a = As[(wco * 32 + l31) * WG_PITCH + kc];
            float b = Bs[(wcl * 32 + l31) * WG_PITCH + kc];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (do_bias) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float v = bsum[p];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int co = co0 + rl + p * 8;
            if (ml == 0 && co < Cout) bias_slab[(size_t)split * Cout + co] = v;
        }
    }
    // column -> (slot, ci): slab[((split*nlive + slot)*Cout + co)*Cin + ci]
    const int col = col0 + wcl * 32 + l31;
    if (col < ncols) {
        const int slot = col / Cin, ci = col - slot * Cin;
        float* ps = slab + ((size_t)split * live.n + slot) * Cout * Cin + ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < Cout) ps[(size_t)co * Cin] = acc[r];
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Weight gradient for NARROW inputs (round 4): (live taps) x Cin <= 31 columns in all — the clips' first convolution (Cin = 1:
// 27 columns) and the stem's 1x1x1 skip convolution (1 column). The launch is a STREAM over dL/dy (256 B per voxel for 27 x 64
// outputs): the column-tile kernel above moves it in 32-voxel chunks with 16 four-byte gathers per lane and two barriers per
// 16 MFMAs (85 us for the 8 discriminator-step members = 1.2 TB/s; 0.42 ms per pass on full clips). Here a round is 128 voxels:
// dL/dy arrives as 16-byte loads along the voxels (8 per lane) and goes to LDS as [co][voxel] rows of pitch 132, the shifted /
// masked input values of the (tap, ci) columns are computed once per voxel and column, and the MFMA k slots are assigned so that
// a lane's 32 operand values are CONTIGUOUS in its row (slot (k2, hi) <-> voxel hi*32 + k2 of the wave's half: 8 ds_read_b128
// per operand instead of 32 ds_read_b32; any slot <-> voxel bijection is a valid contraction order as long as both operands
// use it). Column 31 holds 1 on valid voxels: its accumulator column IS the bias gradient. Waves: 2 (co halves) x 2 (voxel
// halves, summed through LDS at the end). Same chunk table, k-split slab and bias slab as the other weight-gradient kernels.
// ------------------------------------------------------------------------------------------------
#define WT_P 132
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(const WGroupTable tab, float* __restrict__ slab, const int Cin,
                                                              const int Cout, const int T, const int kH, const int kW,
                                                              const int flags, const int chunks_per_split, const LiveTaps live,
                                                              float* __restrict__ bias_slab) {
    __shared__ __attribute__((aligned(16))) float As[64 * WT_P];   // dL/dy tile [co][voxel]
    __shared__ __attribute__((aligned(16))) float Bs[32 * WT_P];   // column tile [col][voxel]
    __shared__ int s_col[32];                                      // (dz+1) | (dy+1) << 2 | (dx+1) << 4 | ci << 8, or -1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave & 1, wk = wave >> 1;
    const int co0 = (int)blockIdx.x * 64;
    const int split = blockIdx.z;
    const int ncols = live.n * Cin;
    const float relu_floor = (flags & T2V_CONV_RELU_IN) ? 0.f : -__builtin_inff();
    const int kD = T / (kH * kW);
#ifdef T2V_ABLATION
    // developer ablations (tools/ablate_thin.py; wrong results): 64 no dL/dy loads, 128 no input gathers, 256 no staging, 512 no MFMAs
    const bool ab_noy = flags & 64, ab_nox = flags & 128, ab_nostage = flags & 256, ab_nomfma = flags & 512;
#else
    constexpr bool ab_noy = false, ab_nox = false, ab_nostage = false, ab_nomfma = false;
#endif
    if (tid < 32) {
        int info = -1;
        if (tid < ncols) {
            const int slot = tid / Cin, ci = tid - slot * Cin;
            const int t = live.t[slot];
            const int dz = t / (kH * kW) - kD / 2, dy = (t / kW) % kH - kH / 2, dx = t % kW - kW / 2;
            info = (dz + 1) | ((dy + 1) << 2) | ((dx + 1) << 4) | (ci << 8);
        }
        s_col[tid] = info;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunks = tab.chunk_start[tab.n];
    const int q0 = split * chunks_per_split;
    int q1 = q0 + chunks_per_split;
    if (q1 > nchunks) q1 = nchunks;

    // ---- member state (scalars), as in the other weight-gradient kernels
    int g_i = -1, g_begin = 0, g_end = 0;
    int gD = 1, gH = 1, gW = 1, gHW = 1, gDHW = 1, gM = 0, g_lw = 0, g_lhw = 0, g_ldhw = 0;
    bool g_pow2 = false;
    __amdgpu_buffer_rsrc_t g_x = __builtin_amdgcn_make_buffer_rsrc((void*)tab.g[0].x, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t g_y = g_x;
    auto enter_group = [&](int q) {
        int gi = 0;
#pragma unroll
        for (int k = 1; k < T2V_MAX_GROUPS; ++k)
            if (k < tab.n && q >= tab.chunk_start[k]) gi = k;
        const t2v_conv_group& gd = tab.g[gi];
        g_i = gi;
        g_begin = tab.chunk_start[gi];
        g_end = tab.chunk_start[gi + 1];
        gD = gd.D; gH = gd.H; gW = gd.W; gHW = gH * gW; gDHW = gD * gHW; gM = gd.N * gDHW;
        g_pow2 = ((gD & (gD - 1)) | (gH & (gH - 1)) | (gW & (gW - 1))) == 0;
        g_lw = __builtin_ctz(gW);
        g_lhw = g_lw + __builtin_ctz(gH);
        g_ldhw = g_lhw + __builtin_ctz(gD);
        g_x = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)gM * (uint32_t)Cin * 4u), 0x00020000);      // (host: M * C < 2^30)
        g_y = __builtin_amdgcn_make_buffer_rsrc((void*)gd.y, 0, (int)((uint32_t)gM * (uint32_t)Cout * 4u), 0x00020000);
    };
    // staging coordinates: dL/dy as 8 x (4 voxels of one channel) per thread; columns as 16 x (one voxel, one column) per thread
    const int a_co = tid >> 5, a_vq = (tid & 31) * 4;
    const int b_v = tid & 127, b_cg = tid >> 7;
    float4 ra[8];
    float rb[16];
    int rnd_chunks = 0;                        // 32-voxel chunks of the pending round
    auto load_round = [&](int q) {
        if (q >= g_end || g_i < 0) enter_group(q);                  // (uniform)
        int R = g_end - q;
        if (R > q1 - q) R = q1 - q;
        if (R > 4) R = 4;
        rnd_chunks = R;
        const int DHW = gDHW, HW = gHW, W = gW, H = gH, D = gD;
        const uint32_t uDHW = (uint32_t)DHW;
        const int mbase = (q - g_begin) * WG_BK;
        auto split_m = [&](int m, int& n, int& sp) {
            if (g_pow2) { n = m >> g_ldhw; sp = m & (DHW - 1); }
            else { n = gM < (1 << 24) ? fast_div(m, DHW, 1.0f / (float)DHW) : m / DHW; sp = m - n * DHW; }
        };
        // ---- dL/dy: voxels a_vq .. a_vq + 3 of channels co0 + a_co + 8 p
        {
            const int m = mbase + a_vq;
            const bool in_round = a_vq < R * WG_BK;
            if (ab_noy && q != q0) {
            } else if ((DHW & 3) == 0) {                            // (uniform) the four voxels share a sample and a 16-byte line
                int n, sp;
                split_m(m, n, sp);
                const bool mv = in_round && m < gM;
                const uint32_t base = mv ? (uint32_t)(n * DHW) * (uint32_t)Cout + (uint32_t)sp : 0u;
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int co = co0 + a_co + 8 * p;
                    const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                        g_y, (base + (uint32_t)(co < Cout ? co : Cout - 1) * uDHW) * 4u, 0, 0));
                    ra[p] = (mv && co < Cout) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            } else {
                uint32_t base[4];
                bool mv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int n, sp;
                    split_m(m + e, n, sp);
                    mv[e] = in_round && m + e < gM;
                    base[e] = mv[e] ? (uint32_t)(n * DHW) * (uint32_t)Cout + (uint32_t)sp : 0u;
                }
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int co = co0 + a_co + 8 * p;
                    const uint32_t cc = (uint32_t)(co < Cout ? co : Cout - 1) * uDHW;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_y, (base[e] + cc) * 4u, 0, 0));
                        v[e] = (mv[e] && co < Cout) ? v[e] : 0.f;
                    }
                    ra[p] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
        // ---- columns cg + 2 j of voxel b_v
        {
            const int m = mbase + b_v;
            const bool mv = b_v < R * WG_BK && m < gM;
            int n, sp;
            split_m(mv ? m : 0, n, sp);
            int d, h, w_;
            if (g_pow2) { d = sp >> g_lhw; h = (sp >> g_lw) & (H - 1); w_ = sp & (W - 1); }
            else {
                const bool small = gM < (1 << 24);
                d = small ? fast_div(sp, HW, 1.0f / (float)HW) : sp / HW;
                const int r = sp - d * HW;
                h = small ? fast_div(r, W, 1.0f / (float)W) : r / W;
                w_ = r - h * W;
            }
            const uint32_t xb = (uint32_t)(n * DHW) * (uint32_t)Cin + (uint32_t)sp;
            // validity of the voxel's neighbours, one bit per offset -1 / 0 / +1 and axis: a column's mask is three shifts
            // (the ablations put these gathers and their index arithmetic at 45 % of the launch on full clips)
            const uint32_t vz = ((unsigned)(d - 1) < (unsigned)D ? 1u : 0u) | 2u | ((unsigned)(d + 1) < (unsigned)D ? 4u : 0u);
            const uint32_t vy = ((unsigned)(h - 1) < (unsigned)H ? 1u : 0u) | 2u | ((unsigned)(h + 1) < (unsigned)H ? 4u : 0u);
            const uint32_t vx = ((unsigned)(w_ - 1) < (unsigned)W ? 1u : 0u) | 2u | ((unsigned)(w_ + 1) < (unsigned)W ? 4u : 0u);
            if (!(ab_nox && q != q0))
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int c = b_cg + 2 * j;
                const int info = s_col[c];
                if (c == 31) { rb[j] = mv ? 1.f : 0.f; continue; }  // the ones column: sum of dL/dy = the bias gradient
                if (info < 0) { rb[j] = 0.f; continue; }            // (wave-uniform) a dead column: no load
                const int iz = info & 3, iy = (info >> 2) & 3, ix = (info >> 4) & 3, ci = (info >> 8) & 0xffff;
                const bool ok = mv && ((vz >> iz) & (vy >> iy) & (vx >> ix) & 1u);
                const uint32_t off = ok ? xb + (uint32_t)ci * uDHW + (uint32_t)((iz - 1) * HW + (iy - 1) * W + (ix - 1)) : 0u;
                const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_x, off * 4u, 0, 0));
                rb[j] = ok ? fmaxf(v, relu_floor) : 0.f;
            }
        }
    };

    __syncthreads();                       // s_col
    int q = q0;
    if (q < q1) load_round(q);
    while (q < q1) {
        const int R = rnd_chunks;
        if (!(ab_nostage && q != q0)) {
#pragma unroll
            for (int p = 0; p < 8; ++p) *reinterpret_cast<float4*>(&As[(a_co + 8 * p) * WT_P + a_vq]) = ra[p];
#pragma unroll
            for (int j = 0; j < 16; ++j) Bs[(b_cg + 2 * j) * WT_P + b_v] = rb[j];
        }
        __syncthreads();
        q += R;
        if (q < q1) load_round(q);
        if (!ab_nomfma) {
            const float* pa = As + (wco * 32 + l31) * WT_P + wk * 64 + hi * 32;
            const float* pb = Bs + l31 * WT_P + wk * 64 + hi * 32;
            float4 a4[8], b4[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                a4[i] = *reinterpret_cast<const float4*>(pa + 4 * i);
                b4[i] = *reinterpret_cast<const float4*>(pb + 4 * i);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[i].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[i].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[i].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[i].w, acc, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- the two voxel halves meet in LDS (fixed order: half 0 + half 1)
    float* red = As;                           // 2 waves x 16 x 64 floats
    if (wk == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wco * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[(wco * 16 + r) * 64 + lane];
        const int col = l31;
        if (col < ncols) {
            const int slot = col / Cin, ci = col - slot * Cin;
            float* ps = slab + ((size_t)split * live.n + slot) * Cout * Cin + ci;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (co < Cout) ps[(size_t)co * Cin] = acc[r];
            }
        } else if (col == 31 && bias_slab != nullptr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (co < Cout) bias_slab[(size_t)split * Cout + co] = acc[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient, kW = 3: one workgroup computes the three taps dx = -1, 0, +1 of one (dz, dy) kernel row.
// They multiply the SAME dL/dy tile and x rows that differ by a one-voxel shift, so dL/dy is staged once and x
// once as a 34-column strip (the 32 voxels of the chunk plus one on either side, by linear index); the three taps
// read the strip at column offsets 0/1/2 and out-of-row neighbours are zeroed with per-chunk ballot masks.
// 17 gathers per lane feed 48 MFMAs per wave (the per-tap kernel: 16 gathers per 16 MFMAs).
// ------------------------------------------------------------------------------------------------
#define W3_AP (WG_BK + 1)      // dL/dy tile pitch
#define W3_BP (WG_BK + 3)      // x strip pitch: 34 columns + 1 pad (odd)
struct LiveRows { int8_t r[9]; int32_t n; };    // slab row slot -> original kernel row (dz,dy) index a*kH + b

#define W3_XP (WG_BK + 3)      // x copies: columns -1 .. 32 (two dummy columns take the out-of-tile writes), odd pitch

// Per-chunk work outside the MFMA loop is what bounds this kernel (a chunk is only 32 voxels = 48 MFMAs per wave): the
// voxel decode and the 17 gather addresses used to cost ~400 vector instructions per chunk (three reciprocal divisions,
// 64-bit multiply-adds per load, the member's descriptor re-read from the kernel arguments), the staging another ~230
// (exec-masked edge writes). Now: the member's descriptor is cached in scalar registers and re-read only when the chunk
// range moves on to the next member; power-of-two extents (every shape of the model) decode with shifts; all offsets are
// 32-bit element offsets from the member's base pointers (build_wtable guarantees M * C < 2^31) and the 8 rows a thread
// gathers are a constant stride apart; the three shifted copies of x are written unconditionally (the two writes that
// fall outside the tile land in dummy columns).
template <bool BF16>
__global__ __launch_bounds__(256, BF16 ? 4 : 0) void conv_wgrad3_kernel(const WGroupTable tab, float* __restrict__ slab, const int Cin,
                                                          const int Cout, const int kD, const int kH, const int flags,
                                                          const int chunks_per_split, const LiveRows live,
                                                          float* __restrict__ bias_slab) {
    // fp32: As[co][m] (pitch 33), Bs[dx][ci][1 + m] (pitch 35, columns -1 and 32 are dummies)
    // bf16-compute mode (BF16): the same tiles rounded to bf16 when written, [row][voxel] with voxel contiguous (= the K of
    //   v_mfma_f32_32x32x16_bf16), pitch 40: columns 32 / 33 of the padding take the two out-of-tile writes
    constexpr int LDS_A = BF16 ? 64 * B16_KP * 2 : 64 * W3_AP * 4;
    constexpr int LDS_B = BF16 ? 3 * 64 * B16_KP * 2 : 3 * 64 * W3_XP * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_A + LDS_B];
    float* const As = reinterpret_cast<float*>(lds_raw);
    float* const Bs = reinterpret_cast<float*>(lds_raw + LDS_A);
    __bf16* const Ah = reinterpret_cast<__bf16*>(lds_raw);
    __bf16* const Bh = reinterpret_cast<__bf16*>(lds_raw + LDS_A);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave & 1, wci = wave >> 1;
    const int nco_t = (Cout + 63) / 64;
    const int co0 = (blockIdx.x % nco_t) * 64, ci0 = (blockIdx.x / nco_t) * 64;
    const int lin = xcd_remap((int)(blockIdx.y + blockIdx.z * gridDim.y), (int)(gridDim.y * gridDim.z));
    const int rslot = lin % (int)gridDim.y;
    const int split = lin / (int)gridDim.y;
    const int krow = live.r[rslot];
    const int dz = krow / kH - kD / 2, dy = krow % kH - kH / 2;
    const float relu_floor = (flags & T2V_CONV_RELU_IN) ? 0.f : -__builtin_inff();
    const int ml = tid & 31, rl = tid >> 5;
    const bool full = co0 + 64 <= Cout && ci0 + 64 <= Cin;       // (block-uniform) no channel clamping needed

    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }

    const int nchunks = tab.chunk_start[tab.n];
    const int q0 = split * chunks_per_split;
    int q1 = q0 + chunks_per_split;
    if (q1 > nchunks) q1 = nchunks;

    float ra[8], rb[8], rh = 0.f;
    uint32_t pm0 = 0, pm1 = 0, pm2 = 0;       // pending chunk: validity of (voxel, dx) as 32-bit masks
    bool pend_mv = false;
    // bias gradient on the side: the workgroups of the first channel tile and first kernel row also add up the dL/dy
    // values they stage anyway (bias_slab[split][co], summed over the splits by one workgroup of the reduce kernel)
    const bool do_bias = bias_slab != nullptr && ci0 == 0 && rslot == 0;
    float bsum[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) bsum[p] = 0.f;

    // ---- the member the current chunk range lies in (scalar state, reloaded when q passes g_end)
    int g_i = -1, g_begin = 0, g_end = 0;
    int gD = 1, gH = 1, gW = 1, gHW = 1, gDHW = 1, gM = 0, g_lw = 0, g_lhw = 0, g_shift = 0;
    int gDHWo = 1, g_ds = 1, g_lo = 0;        // frame-strided dL/dy (dstride = 2): its own frame count / voxel count, log2 of the latter
    bool g_pow2 = false;
    // buffer descriptors of the member's x and dL/dy: 32-bit byte offsets in the gathers (one add, no 64-bit address
    // arithmetic), out-of-range offsets read 0. Built from kernel-argument scalars only (provably wave-uniform).
    __amdgpu_buffer_rsrc_t g_x = __builtin_amdgcn_make_buffer_rsrc((void*)tab.g[0].x, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t g_y = g_x;
    auto enter_group = [&](int q) {
        int gi = 0;
#pragma unroll
        for (int k = 1; k < T2V_MAX_GROUPS; ++k)
            if (k < tab.n && q >= tab.chunk_start[k]) gi = k;
        const t2v_conv_group& gd = tab.g[gi];
        g_i = gi;
        g_begin = tab.chunk_start[gi];
        g_end = tab.chunk_start[gi + 1];
        gD = gd.D; gH = gd.H; gW = gd.W; gHW = gH * gW; gDHW = gD * gHW;
        g_ds = gd.dstride == 2 ? 2 : 1;
        const int Do = g_ds == 2 ? (gD + 1) / 2 : gD;
        gDHWo = Do * gHW;
        gM = gd.N * gDHWo;                                          // chunks run over the voxels of dL/dy
        g_pow2 = ((Do & (Do - 1)) | (gH & (gH - 1)) | (gW & (gW - 1))) == 0;
        g_lw = __builtin_ctz(gW);
        g_lhw = g_lw + __builtin_ctz(gH);
        g_lo = g_lhw + __builtin_ctz(Do);
        g_shift = dz * gHW + dy * gW;
        g_x = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)(gd.N * gDHW) * (uint32_t)Cin * 4u), 0x00020000);      // (host: M * C < 2^30)
        g_y = __builtin_amdgcn_make_buffer_rsrc((void*)gd.y, 0, (int)((uint32_t)gM * (uint32_t)Cout * 4u), 0x00020000);
    };

    auto load_chunk = [&](int q) {
        if (q >= g_end || g_i < 0) enter_group(q);                  // (uniform)
        const int D = gD, H = gH, W = gW, HW = gHW, DHW = gDHW, DHWo = gDHWo;
        const int m = (q - g_begin) * WG_BK + ml;                   // voxel of dL/dy: (n, d_o, h, w)
        const bool mv = m < gM;
        int n, spo, d_o, h, w_;
        if (g_pow2) {                                               // (uniform) every extent a power of two: shifts
            n = m >> g_lo;
            spo = m & (DHWo - 1);
            d_o = spo >> g_lhw;
            h = (spo >> g_lw) & (H - 1);
            w_ = spo & (W - 1);
        } else {
            const bool small = gM < (1 << 24);
            n = small ? fast_div(m, DHWo, 1.0f / (float)DHWo) : m / DHWo;
            spo = m - n * DHWo;
            d_o = small ? fast_div(spo, HW, 1.0f / (float)HW) : spo / HW;
            const int r = spo - d_o * HW;
            h = small ? fast_div(r, W, 1.0f / (float)W) : r / W;
            w_ = r - h * W;
        }
        const int d = d_o * g_ds;                                   // the input frame this dL/dy voxel sits on
        const int sp = spo + d_o * (g_ds - 1) * HW;                 // ... and its voxel index inside the sample of x
        const bool vc = mv && (unsigned)(d + dz) < (unsigned)D && (unsigned)(h + dy) < (unsigned)H;
        // element offsets of (n, channel 0, voxel) in dL/dy and in x
        const uint32_t gbase = mv ? (uint32_t)(n * DHWo) * (uint32_t)Cout + (uint32_t)spo : 0u;
        const uint32_t xb = vc ? (uint32_t)(n * DHW) * (uint32_t)Cin + (uint32_t)(sp + g_shift) : 0u;
        // wave-uniform validity masks (every wave sees the same 32 voxels in lanes 0-31)
        pm0 = (uint32_t)__ballot(vc && w_ >= 1);
        pm1 = (uint32_t)__ballot(vc);
        pm2 = (uint32_t)__ballot(vc && w_ + 1 < W);
        pend_mv = mv;
        const uint32_t uDHW = (uint32_t)DHW, uDHWo = (uint32_t)DHWo;
        if (full) {                                                 // rows rl, rl+8, ...: a constant (scalar) stride apart
            const uint32_t oa = (gbase + (uint32_t)(co0 + rl) * uDHWo) * 4u, ob = (xb + (uint32_t)(ci0 + rl) * uDHW) * 4u;
            const int st = 32 * DHW, sty = 32 * DHWo;               // 8 rows, in bytes
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                ra[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_y, oa, p * sty, 0));
                rb[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_x, ob, p * st, 0));
            }
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p) {                           // clamped rows (zeroed when staged)
                const int co = co0 + rl + p * 8, ci = ci0 + rl + p * 8;
                ra[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    g_y, (gbase + (uint32_t)(co < Cout ? co : Cout - 1) * uDHWo) * 4u, 0, 0));
                rb[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    g_x, (xb + (uint32_t)(ci < Cin ? ci : Cin - 1) * uDHW) * 4u, 0, 0));
            }
        }
        // the two extra strip columns: linear neighbours of voxel 0 (left) and voxel 31 (right) of the chunk
        const uint32_t xb_l = __shfl(xb, 0, 64), xb_r = __shfl(xb, 31, 64);
        if (tid < 128) {
            const int row = tid >> 1, side = tid & 1;
            const int ci = ci0 + row;
            const uint32_t base = side ? (((pm2 >> 31) & 1u) ? xb_r + 1u : 0u) : ((pm0 & 1u) ? xb_l - 1u : 0u);
            rh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                g_x, (base + (uint32_t)(ci < Cin ? ci : Cin - 1) * uDHW) * 4u, 0, 0));
        }
    };

    if (q0 < q1) load_chunk(q0);
    for (int q = q0; q < q1; ++q) {
        {
            // my voxel ml is the LEFT neighbour of voxel ml+1 (copy dx=-1, column ml+1), itself (column ml), and the RIGHT
            // neighbour of voxel ml-1 (copy dx=+1, column ml-1); columns -1 and 32 are dummies
            const bool v0 = (((uint64_t)pm0 >> (ml + 1)) & 1u) != 0;
            const bool v1 = ((pm1 >> ml) & 1u) != 0;
            const bool v2 = ((((uint64_t)pm2 << 1) >> ml) & 1u) != 0;
            if (do_bias) {
#pragma unroll
                for (int p = 0; p < 8; ++p) bsum[p] += pend_mv ? ra[p] : 0.f;
            }
            if constexpr (BF16) {
                const int c_m1 = ml + 1, c_p1 = ml >= 1 ? ml - 1 : 33;       // (columns 32 / 33: padding)
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    const int row = rl + p * 8;
                    float a = ra[p], v = fmaxf(rb[p], relu_floor);
                    if (!full) {
                        a = (co0 + row < Cout) ? a : 0.f;
                        v = (ci0 + row < Cin) ? v : 0.f;
                    }
                    Ah[row * B16_KP + ml] = (__bf16)a;
                    Bh[(0 * 64 + row) * B16_KP + c_m1] = (__bf16)(v0 ? v : 0.f);
                    Bh[(1 * 64 + row) * B16_KP + ml] = (__bf16)(v1 ? v : 0.f);
                    Bh[(2 * 64 + row) * B16_KP + c_p1] = (__bf16)(v2 ? v : 0.f);
                }
                if (tid < 128) {
                    const int row = tid >> 1, side = tid & 1;
                    float v = (ci0 + row < Cin) ? fmaxf(rh, relu_floor) : 0.f;
                    if (side) Bh[(2 * 64 + row) * B16_KP + WG_BK - 1] = (__bf16)(((pm2 >> 31) & 1u) ? v : 0.f);
                    else Bh[(0 * 64 + row) * B16_KP + 0] = (__bf16)((pm0 & 1u) ? v : 0.f);
                }
            } else {
            float* pa = &As[rl * W3_AP + ml];
            float* pb = &Bs[rl * W3_XP + 1 + ml];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float a = ra[p], v = fmaxf(rb[p], relu_floor);
                if (!full) {
                    const int row = rl + p * 8;
                    a = (co0 + row < Cout) ? a : 0.f;
                    v = (ci0 + row < Cin) ? v : 0.f;
                }
                pa[p * 8 * W3_AP] = a;
                pb[(0 * 64 + p * 8) * W3_XP + 1] = v0 ? v : 0.f;
                pb[(1 * 64 + p * 8) * W3_XP] = v1 ? v : 0.f;
                pb[(2 * 64 + p * 8) * W3_XP - 1] = v2 ? v : 0.f;
            }
            if (tid < 128) {
                const int row = tid >> 1, side = tid & 1;
                float v = (ci0 + row < Cin) ? fmaxf(rh, relu_floor) : 0.f;
                if (side) Bs[(2 * 64 + row) * W3_XP + 1 + WG_BK - 1] = ((pm2 >> 31) & 1u) ? v : 0.f;
                else Bs[(0 * 64 + row) * W3_XP + 1 + 0] = (pm0 & 1u) ? v : 0.f;
            }
            }
        }
        __syncthreads();
        if (q + 1 < q1) load_chunk(q + 1);
        if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < WG_BK / 16; ++ks) {      // K = 16 voxels per bf16 MFMA: lane (row l31, half hi) reads 8 of them
                const int kc = ks * 16 + 8 * hi;
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ah + (wco * 32 + l31) * B16_KP + kc);
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bh + (0 * 64 + wci * 32 + l31) * B16_KP + kc);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bh + (1 * 64 + wci * 32 + l31) * B16_KP + kc);
                const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(Bh + (2 * 64 + wci * 32 + l31) * B16_KP + kc);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, acc2, 0, 0, 0);
            }
        } else {
        const float* qa = &As[(wco * 32 + l31) * W3_AP + hi];
        const float* qb = &Bs[(wci * 32 + l31) * W3_XP + 1 + hi];
#pragma unroll
        for (int k2 = 0; k2 < WG_BK / 2; ++k2) {
            const float a = qa[k2 * 2];
            const float b0 = qb[0 * 64 * W3_XP + k2 * 2];
            const float b1 = qb[1 * 64 * W3_XP + k2 * 2];
            const float b2 = qb[2 * 64 * W3_XP + k2 * 2];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc2, 0, 0, 0);
        }
        }
        __syncthreads();
    }
    if (do_bias) {                            // lanes ml = 0..31 of a half-wave hold the same 8 output channels
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float v = bsum[p];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int co = co0 + rl + p * 8;
            if (ml == 0 && co < Cout) bias_slab[(size_t)split * Cout + co] = v;
        }
    }
    // slab[((split*nslots + rslot*3 + dx+1)*Cout + co)*Cin + ci]
    const int ci = ci0 + wci * 32 + l31;
    if (ci < Cin) {
        const size_t CoCi = (size_t)Cout * Cin;
        float* ps = slab + ((size_t)split * (live.n * 3) + (size_t)rslot * 3) * CoCi + ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < Cout) {
                ps[(size_t)co * Cin] = acc0[r];
                ps[CoCi + (size_t)co * Cin] = acc1[r];
                ps[2 * CoCi + (size_t)co * Cin] = acc2[r];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// POOLED CONVOLUTION (weight gradient): dW[dz,dy,dx] = sum over the POOLED voxels m' of dL/dy[m'] (x) r~[2 m' + tap + 1] — the
// 3-tap-row kernel above with a mask-free stride-2 gather from the padded box-summed activation (see conv_pool_fwd_kernel): one
// 8-byte and one 4-byte load per (voxel, channel) give the three dx operands. Same slab layout as conv_wgrad3_kernel (the
// reduce kernels do not know the difference), same bias side-sum. Chunks of 32 pooled voxels; members without a time axis
// (tmode 0) contribute to the dz = 0 rows only.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 4) void conv_pool_wgrad_kernel(const WGroupTable tab, float* __restrict__ slab, const int Cin,
                                                              const int Cout, const int chunks_per_split, const LiveRows live,
                                                              float* __restrict__ bias_slab) {
    constexpr int PA = WG_BK + 1;
    __shared__ __attribute__((aligned(16))) float As[64 * PA];          // dL/dy tile [co][m]
    __shared__ __attribute__((aligned(16))) float Bs[3 * 64 * PA];      // r~ [dx][ci][m]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave & 1, wci = wave >> 1;
    const int nco_t = (Cout + 63) / 64;
    const int co0 = (blockIdx.x % nco_t) * 64, ci0 = (blockIdx.x / nco_t) * 64;
    const int lin = xcd_remap((int)(blockIdx.y + blockIdx.z * gridDim.y), (int)(gridDim.y * gridDim.z));
    const int rslot = lin % (int)gridDim.y;
    const int split = lin / (int)gridDim.y;
    const int krow = live.r[rslot];
    const int dz = krow / 3 - 1, dy = krow % 3 - 1;
    const int ml = tid & 31, rl = tid >> 5;
    const bool full = co0 + 64 <= Cout && ci0 + 64 <= Cin;

    f32x16 acc0, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }

    const int nchunks = tab.chunk_start[tab.n];
    const int q0 = split * chunks_per_split;
    int q1 = q0 + chunks_per_split;
    if (q1 > nchunks) q1 = nchunks;

    float ra[8], rb1[8];
    float2 rb2[8];
    bool pend_mv = false, pend_vc = false;
    const bool do_bias = bias_slab != nullptr && ci0 == 0 && rslot == 0;
    float bsum[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) bsum[p] = 0.f;

    int g_i = -1, g_begin = 0, g_end = 0;
    int g_tm = 0, gHn = 1, gWn = 1, gHWn = 1, gVn = 1, gM = 0, gHp = 1, gWp = 2, gVp = 2;
    int g_lw = 0, g_lhw = 0, g_lv = 0;       // log2 of the pooled extents (every shape of the model is a power of two: shift decode)
    bool g_pow2 = false;
    __amdgpu_buffer_rsrc_t g_x = __builtin_amdgcn_make_buffer_rsrc((void*)tab.g[0].x, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t g_y = g_x;
    auto enter_group = [&](int q) {
        int gi = 0;
#pragma unroll
        for (int k = 1; k < T2V_MAX_GROUPS; ++k)
            if (k < tab.n && q >= tab.chunk_start[k]) gi = k;
        const t2v_conv_group& gd = tab.g[gi];
        g_i = gi;
        g_begin = tab.chunk_start[gi];
        g_end = tab.chunk_start[gi + 1];
        g_tm = gd.dstride;
        const int Dn = g_tm ? gd.D / 2 : 1;
        gHn = gd.H / 2; gWn = gd.W / 2; gHWn = gHn * gWn; gVn = Dn * gHWn;
        gM = gd.N * gVn;
        g_pow2 = ((Dn & (Dn - 1)) | (gHn & (gHn - 1)) | (gWn & (gWn - 1))) == 0;
        g_lw = __builtin_ctz(gWn);
        g_lhw = g_lw + __builtin_ctz(gHn);
        g_lv = g_lhw + __builtin_ctz(Dn);
        gHp = gd.H + 1; gWp = gd.W + 2;
        gVp = (g_tm ? gd.D + 1 : 1) * gHp * gWp;
        g_x = __builtin_amdgcn_make_buffer_rsrc((void*)gd.x, 0, (int)((uint32_t)(gd.N * gVp) * (uint32_t)Cin * 4u), 0x00020000);
        g_y = __builtin_amdgcn_make_buffer_rsrc((void*)gd.y, 0, (int)((uint32_t)gM * (uint32_t)Cout * 4u), 0x00020000);
    };

    auto load_chunk = [&](int q) {
        if (q >= g_end || g_i < 0) enter_group(q);                  // (uniform)
        const int m = (q - g_begin) * WG_BK + ml;                   // pooled voxel (n, e, i, j)
        const bool mv = m < gM;
        int n, sp, e, i, j;
        if (g_pow2) {                                               // (uniform)
            n = m >> g_lv;
            sp = m & (gVn - 1);
            e = sp >> g_lhw;
            i = (sp >> g_lw) & (gHn - 1);
            j = sp & (gWn - 1);
        } else {
            const bool small = gM < (1 << 24);
            n = small ? fast_div(m, gVn, 1.0f / (float)gVn) : m / gVn;
            sp = m - n * gVn;
            e = small ? fast_div(sp, gHWn, 1.0f / (float)gHWn) : sp / gHWn;
            const int r = sp - e * gHWn;
            i = small ? fast_div(r, gWn, 1.0f / (float)gWn) : r / gWn;
            j = r - i * gWn;
        }
        const bool vc = mv && (g_tm != 0 || dz == 0);               // members without a time axis only have the dz = 0 rows
        const uint32_t gbase = mv ? (uint32_t)(n * gVn) * (uint32_t)Cout + (uint32_t)sp : 0u;
        const uint32_t xb = vc ? (uint32_t)(n * gVp) * (uint32_t)Cin +
                                 (uint32_t)(((g_tm ? 2 * e + dz + 1 : 0) * gHp + 2 * i + dy + 1) * gWp + 2 * j) : 0u;
        pend_mv = mv;
        pend_vc = vc;
        const uint32_t uVp = (uint32_t)gVp, uVn = (uint32_t)gVn;
        if (full) {
            const uint32_t oa = (gbase + (uint32_t)(co0 + rl) * uVn) * 4u, ob = (xb + (uint32_t)(ci0 + rl) * uVp) * 4u;
            const int st = 32 * gVp, sty = 32 * gVn;                // 8 rows, in bytes
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                ra[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_y, oa, p * sty, 0));
                rb2[p] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(g_x, ob, p * st, 0));
                rb1[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_x, ob + 8u, p * st, 0));
            }
        } else {
#pragma unroll
            for (int p = 0; p < 8; ++p) {                           // clamped rows (zeroed when staged)
                const int co = co0 + rl + p * 8, ci = ci0 + rl + p * 8;
                ra[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    g_y, (gbase + (uint32_t)(co < Cout ? co : Cout - 1) * uVn) * 4u, 0, 0));
                const uint32_t ob = (xb + (uint32_t)(ci < Cin ? ci : Cin - 1) * uVp) * 4u;
                rb2[p] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(g_x, ob, 0, 0));
                rb1[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_x, ob + 8u, 0, 0));
            }
        }
    };

    if (q0 < q1) load_chunk(q0);
    for (int q = q0; q < q1; ++q) {
        {
            if (do_bias) {
#pragma unroll
                for (int p = 0; p < 8; ++p) bsum[p] += pend_mv ? ra[p] : 0.f;
            }
            float* pa = &As[rl * PA + ml];
            float* pb = &Bs[rl * PA + ml];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float a = pend_mv ? ra[p] : 0.f;
                float v0 = pend_vc ? rb2[p].x : 0.f, v1 = pend_vc ? rb2[p].y : 0.f, v2 = pend_vc ? rb1[p] : 0.f;
                if (!full) {
                    const int row = rl + p * 8;
                    a = (co0 + row < Cout) ? a : 0.f;
                    const bool ok = ci0 + row < Cin;
                    v0 = ok ? v0 : 0.f; v1 = ok ? v1 : 0.f; v2 = ok ? v2 : 0.f;
                }
                pa[p * 8 * PA] = a;
                pb[(0 * 64 + p * 8) * PA] = v0;
                pb[(1 * 64 + p * 8) * PA] = v1;
                pb[(2 * 64 + p * 8) * PA] = v2;
            }
        }
        __syncthreads();
        if (q + 1 < q1) load_chunk(q + 1);
        const float* qa = &As[(wco * 32 + l31) * PA + hi];
        const float* qb = &Bs[(wci * 32 + l31) * PA + hi];
#pragma unroll
        for (int k2 = 0; k2 < WG_BK / 2; ++k2) {
            const float a = qa[k2 * 2];
            const float b0 = qb[0 * 64 * PA + k2 * 2];
            const float b1 = qb[1 * 64 * PA + k2 * 2];
            const float b2 = qb[2 * 64 * PA + k2 * 2];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc2, 0, 0, 0);
        }
        __syncthreads();
    }
    if (do_bias) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float v = bsum[p];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int co = co0 + rl + p * 8;
            if (ml == 0 && co < Cout) bias_slab[(size_t)split * Cout + co] = v;
        }
    }
    const int ci = ci0 + wci * 32 + l31;
    if (ci < Cin) {
        const size_t CoCi = (size_t)Cout * Cin;
        float* ps = slab + ((size_t)split * (live.n * 3) + (size_t)rslot * 3) * CoCi + ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < Cout) {
                ps[(size_t)co * Cin] = acc0[r];
                ps[CoCi + (size_t)co * Cin] = acc1[r];
                ps[2 * CoCi + (size_t)co * Cin] = acc2[r];
            }
        }
    }
}


// one workgroup's share of  dbias[co] = (accum ? dbias[co] : 0) + sum_s bias_slab[s][co]  (fixed order); called by ONE
// workgroup of the weight-gradient reduce kernels so that the bias needs no launch of its own
__device__ __forceinline__ void bias_slab_reduce(const float* __restrict__ bias_slab, float* __restrict__ dbias, int S, int Cout,
                                                 int accum) {
    for (int co = threadIdx.x; co < Cout; co += 256) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        int s = 0;
        for (; s + 4 <= S; s += 4) {
            v0 += bias_slab[(size_t)s * Cout + co]; v1 += bias_slab[(size_t)(s + 1) * Cout + co];
            v2 += bias_slab[(size_t)(s + 2) * Cout + co]; v3 += bias_slab[(size_t)(s + 3) * Cout + co];
        }
        for (; s < S; ++s) v0 += bias_slab[(size_t)s * Cout + co];
        const float v = (v0 + v1) + (v2 + v3);
        dbias[co] = accum ? dbias[co] + v : v;
    }
}
// sum over the k-splits of one (tap, pair): eight independent partial sums keep eight loads in flight per lane (the reduce
// is a pure stream of `S` planes; with four it ran at 2.7 TB/s); fixed order -> deterministic
__device__ __forceinline__ float slab_sum4(const float* __restrict__ p, size_t st, int S) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] += p[(size_t)(s + u) * st];
    }
    for (; s < S; ++s) v[0] += p[(size_t)s * st];
    return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}
// the same (and in the same order) over four neighbouring pairs: 16-byte loads, 1 KB per wave and plane
__device__ __forceinline__ float4 slab_sum_v4(const float* __restrict__ p, size_t st, int S) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= S; s += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 a = *reinterpret_cast<const float4*>(p + (size_t)(s + u) * st);
            v[u].x += a.x; v[u].y += a.y; v[u].z += a.z; v[u].w += a.w;
        }
    }
    for (; s < S; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(p + (size_t)s * st);
        v[0].x += a.x; v[0].y += a.y; v[0].z += a.z; v[0].w += a.w;
    }
    float4 r;
    r.x = ((v[0].x + v[1].x) + (v[2].x + v[3].x)) + ((v[4].x + v[5].x) + (v[6].x + v[7].x));
    r.y = ((v[0].y + v[1].y) + (v[2].y + v[3].y)) + ((v[4].y + v[5].y) + (v[6].y + v[7].y));
    r.z = ((v[0].z + v[1].z) + (v[2].z + v[3].z)) + ((v[4].z + v[5].z) + (v[6].z + v[7].z));
    r.w = ((v[0].w + v[1].w) + (v[2].w + v[3].w)) + ((v[4].w + v[5].w) + (v[6].w + v[7].w));
    return r;
}
struct TapMap { int32_t j[T2V_MAX_TAPS]; };   // original tap t -> slab slot or -1 (never touched: write 0)

// dw[co][ci][t] = sum_s slab[s][j(t)][co][ci] (0 for taps that only ever multiply padding). Reads are
// lane-contiguous along (co,ci); the [i][t] transposition goes through LDS so that the PyTorch-layout
// gradient is written as one contiguous run per workgroup.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           long CoCi, int T, int ntaps, int S, TapMap map, int accum,
                                                           const float* __restrict__ bias_slab, float* __restrict__ dbias,
                                                           int Cout, int accum_bias) {
    if (bias_slab && blockIdx.x == 0) bias_slab_reduce(bias_slab, dbias, S, Cout, accum_bias);
    // workgroup = 64 (co,ci) pairs x T taps; the 4 waves take taps t = wave, wave+4, ...
    __shared__ float tile[64 * T2V_MAX_TAPS];
    const long i0 = (long)blockIdx.x * 64;
    const int il = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const long i = i0 + il;
    for (int t = tg; t < T; t += 4) {
        const int j = map.j[t];
        float v = 0.f;
        if (j >= 0 && i < CoCi) v = slab_sum4(slab + (size_t)j * CoCi + i, (size_t)ntaps * CoCi, S);
        tile[il * T + t] = v;
    }
    __syncthreads();
    long cnt = CoCi - i0;
    if (cnt > 64) cnt = 64;
    const long nval = cnt * T;
    float* p = dw + (size_t)i0 * T;
    for (long k = threadIdx.x; k < nval; k += 256) p[k] = accum ? p[k] + tile[k] : tile[k];
}

// Small weights with many k-splits (the 64x64 and 64x1 stem convs: 171-256 partial slabs): one workgroup per
// (64 pairs, tap); the 4 waves each sum a quarter of the splits and combine through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_small_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                 long CoCi, int T, int ntaps, int S, TapMap map, int accum,
                                                                 const float* __restrict__ bias_slab, float* __restrict__ dbias,
                                                                 int Cout, int accum_bias) {
    if (bias_slab && blockIdx.x == 0 && blockIdx.y == 0) bias_slab_reduce(bias_slab, dbias, S, Cout, accum_bias);
    __shared__ float part[4][64];
    const int t = blockIdx.y;
    const long i = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int wv = threadIdx.x >> 6;
    const int j = map.j[t];
    float v0 = 0.f, v1 = 0.f;
    if (j >= 0 && i < CoCi) {
        const float* p = slab + (size_t)j * CoCi + i;
        const size_t st = (size_t)ntaps * CoCi;
        int s = wv;
        for (; s + 4 < S; s += 8) { v0 += p[(size_t)s * st]; v1 += p[(size_t)(s + 4) * st]; }
        if (s < S) v0 += p[(size_t)s * st];
    }
    part[wv][threadIdx.x & 63] = v0 + v1;
    __syncthreads();
    if (wv == 0 && i < CoCi) {
        const float v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        float* q = dw + (size_t)i * T + t;
        *q = accum ? *q + v : v;
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient on 1x1x1 maps (the ConvLSTM's state [B,1024,1,1] with time folded into the batch, nn.Linear, any 3^k kernel whose
// only live tap is the centre): x [M][Cin] and dL/dy [M][Cout] are row-major with the channels contiguous, so
// dw[co][ci] = sum_m gy[m][co] * x[m][ci] is a plain TN product. The voxel-major gathers of the per-tap kernel read such operands
// with a stride of C floats per lane (35 TFLOP/s on the [480 x 4096]^T . [480 x 1024] launch); here a workgroup owns a 128 x 128
// tile of dw, streams 32-row chunks of both operands with 16-byte loads along the channels, and each wave runs a 2 x 2 block of
// 32x32x2 MFMA tiles (four LDS reads per four MFMAs). Same chunk table, k-split slab format (one slot: the centre tap) and bias
// side-sums as the other weight-gradient kernels. Cin % 4 == 0, Cout % 4 == 0.
// ------------------------------------------------------------------------------------------------
#define WGM_AP 132
__global__ __launch_bounds__(256) void conv_wgrad_gemm_kernel(const WGroupTable tab, float* __restrict__ slab, const int Cin,
                                                              const int Cout, const int flags, const int chunks_per_split,
                                                              float* __restrict__ bias_slab) {
    __shared__ __attribute__((aligned(16))) float As[WG_BK * WGM_AP];   // gy chunk [m][co]
    __shared__ __attribute__((aligned(16))) float Bs[WG_BK * WGM_AP];   // x  chunk [m][ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave & 1, wci = wave >> 1;
    const int nci_t = (Cin + 127) / 128;
    const int co0 = ((int)blockIdx.x / nci_t) * 128, ci0 = ((int)blockIdx.x % nci_t) * 128;
    const int split = blockIdx.z;
    const bool relu_in = flags & T2V_CONV_RELU_IN;
    const int nchunks = tab.chunk_start[tab.n];
    const int q0 = split * chunks_per_split;
    int q1 = q0 + chunks_per_split;
    if (q1 > nchunks) q1 = nchunks;
    const int c4 = (tid & 31) * 4, r8 = tid >> 5;                      // my float4 column, my first row (rows r8 + 8 j)
    const bool a_ok = co0 + c4 < Cout, b_ok = ci0 + c4 < Cin;
    const bool do_bias = bias_slab != nullptr && ci0 == 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[4], rb[4];
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_chunk = [&](int q) {
        int gi = 0;
#pragma unroll
        for (int k = 1; k < T2V_MAX_GROUPS; ++k)
            if (k < tab.n && q >= tab.chunk_start[k]) gi = k;
        const t2v_conv_group& gd = tab.g[gi];
        const int row0 = (q - tab.chunk_start[gi]) * WG_BK + r8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = row0 + 8 * j;
            const bool v = m < gd.N;
            ra[j] = (v && a_ok) ? *reinterpret_cast<const float4*>(gd.y + (size_t)m * Cout + co0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 t = (v && b_ok) ? *reinterpret_cast<const float4*>(gd.x + (size_t)m * Cin + ci0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (relu_in) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
            rb[j] = t;
        }
    };
    if (q0 < q1) load_chunk(q0);
    for (int q = q0; q < q1; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4*>(&As[(r8 + 8 * j) * WGM_AP + c4]) = ra[j];
            *reinterpret_cast<float4*>(&Bs[(r8 + 8 * j) * WGM_AP + c4]) = rb[j];
            if (do_bias) { bsum.x += ra[j].x; bsum.y += ra[j].y; bsum.z += ra[j].z; bsum.w += ra[j].w; }
        }
        __syncthreads();
        if (q + 1 < q1) load_chunk(q + 1);
        const float* as = As + wco * 64 + l31;
        const float* bs = Bs + wci * 64 + l31;
#pragma unroll
        for (int k2 = 0; k2 < WG_BK / 2; ++k2) {
            const int krow = (k2 * 2 + hi) * WGM_AP;
            const float a0 = as[krow], a1 = as[krow + 32], b0 = bs[krow], b1 = bs[krow + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    // rows (registers) = co, columns (lanes) = ci: 128-byte row segments
    float* out = slab + (size_t)split * Cout * Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wci * 64 + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (co < Cout && ci < Cin) out[(size_t)co * Cin + ci] = acc[i][j][r];
            }
        }
    if (do_bias) {                       // column sums of every gy chunk this split staged: 8 row groups -> one value per channel
        float* red = As;                 // (the last chunk's reads are behind the loop's closing barrier)
        *reinterpret_cast<float4*>(&red[r8 * WGM_AP + c4]) = bsum;
        __syncthreads();
        if (tid < 128 && co0 + tid < Cout) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) v += red[k * WGM_AP + tid];
            bias_slab[(size_t)split * Cout + co0 + tid] = v;
        }
    }
}

struct WgradPlan { int S, cps; uint32_t live; int nlive; long nchunks; bool rows3; uint32_t liverows; int nrows; bool strided; bool gemm; bool thin; };

static bool build_wtable(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW, bool need_ptrs,
                         WGroupTable& tab, WgradPlan& p) {
    if (!groups || ngroups < 1 || ngroups > T2V_MAX_GROUPS || Cin < 1 || Cout < 1) return false;
    if ((kD != 1 && kD != 3) || (kH != 1 && kH != 3) || (kW != 1 && kW != 3)) return false;
    long nch = 0, maxMC = 0;
    bool strided = false;
    p.live = 0;
    tab.n = ngroups;
    for (int i = 0; i < ngroups; ++i) {
        const t2v_conv_group& g = groups[i];
        if (need_ptrs && (!g.x || !g.y)) return false;
        if (g.N < 1 || g.D < 1 || g.H < 1 || g.W < 1 || g.dstride < 0 || g.dstride > 2 || g.ydstride > 1 || g.ydstride < 0) return false;
        if (g.dstride == 2) strided = true;          // dL/dy lives on the even frames only ([N,Cout,ceil(D/2),H,W]): 3-tap-row kernel only
        const long M = (long)g.N * g.D * g.H * g.W;
        if (M * (long)(Cin > Cout ? Cin : Cout) >= (1L << 30)) return false;     // the gathers use 32-bit BYTE offsets (buffer loads)
        if (M * (long)(Cin > Cout ? Cin : Cout) > maxMC) maxMC = M * (long)(Cin > Cout ? Cin : Cout);
        tab.g[i] = g;
        tab.chunk_start[i] = (int32_t)nch;
        nch += ((long)g.N * out_frames(g) * g.H * g.W + WG_BK - 1) / WG_BK;       // chunks run over the voxels of dL/dy
        for (int a = 0; a < kD; ++a) for (int b = 0; b < kH; ++b) for (int c = 0; c < kW; ++c) {
            const int dz = a - kD / 2, dy = b - kH / 2, dx = c - kW / 2;
            if (!((g.D == 1 && dz) || (g.H == 1 && dy) || (g.W == 1 && dx))) p.live |= 1u << ((a * kH + b) * kW + c);
        }
    }
    for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) tab.chunk_start[i] = (int32_t)nch;
    p.nchunks = nch;
    p.nlive = __builtin_popcount(p.live);
    // kW == 3 with at least one member wider than one voxel: the three dx taps of a kernel row share a workgroup
    bool anyw = false;
    for (int i = 0; i < ngroups; ++i) anyw = anyw || groups[i].W > 1;
    p.rows3 = (kW == 3) && anyw && Cin >= 64 && maxMC < (1L << 30);   // the 3-tap kernel gathers through 32-bit byte offsets
    // every member on a 1x1x1 map (only the centre tap lives): the TN-product kernel
    bool all1 = true;
    for (int i = 0; i < ngroups; ++i) all1 = all1 && groups[i].D == 1 && groups[i].H == 1 && groups[i].W == 1;
    p.gemm = all1 && !strided && Cin >= 64 && (Cin % 4) == 0 && (Cout % 4) == 0;
    if (strided && !p.rows3) return false;
    p.strided = strided;
    p.liverows = 0;
    for (int r = 0; r < kD * kH; ++r)
        if ((p.live >> (r * kW)) & 7u) p.liverows |= 1u << r;
    p.nrows = __builtin_popcount(p.liverows);
    if (p.rows3) p.nlive = p.nrows * 3;            // slab slots (dead dx taps of a live row are written as zeros)
    // narrow inputs: every (live tap, ci) column fits ONE 32-column tile with a spare column for the bias sums (conv_wgrad_thin_kernel)
    p.thin = !p.rows3 && !p.gemm && !strided && Cin < 64 && (long)p.nlive * Cin <= 31 && tun().wgrad_thin;
    if (p.thin) {
        // a stream over dL/dy: three workgroups per CU (LDS) in ONE round, rounds of 128 voxels (4 chunks), at least 2 rounds per split
        const long tiles = (Cout + 63) / 64;
        long S = (768 + tiles - 1) / tiles;
        const long maxS = (nch + 7) / 8;
        if (S > maxS) S = maxS;
        if (S < 1) S = 1;
        if (S > 1024) S = 1024;
        long cps = (nch + S - 1) / S;
        cps = (cps + 3) / 4 * 4;                  // whole rounds
        p.cps = (int)cps;
        p.S = (int)((nch + cps - 1) / cps);
        return true;
    }
    if (p.gemm) {                                 // 128 x 128 tiles, one per CU: k-splits only until the chip is covered once
        const long tiles = (long)((Cout + 127) / 128) * ((Cin + 127) / 128);
        long S = tiles >= 192 ? 1 : (256 + tiles - 1) / tiles;
        const long maxS = (nch + 3) / 4;          // at least 4 chunks (128 rows) per split
        if (S > maxS) S = maxS;
        if (S < 1) S = 1;
        if (S > tun().wgrad_scap) S = tun().wgrad_scap;
        p.cps = (int)((nch + S - 1) / S);
        p.S = (int)((nch + p.cps - 1) / p.cps);
        return true;
    }
    const long base = p.rows3 ? (long)((Cout + 63) / 64) * ((Cin + 63) / 64) * p.nrows
                    : (Cin < 64) ? (long)((Cout + 63) / 64) * (((long)p.nlive * Cin + 63) / 64)
                                 : (long)((Cout + 63) / 64) * ((Cin + 63) / 64) * p.nlive;
    // workgroups to aim at: the kernels run 4 workgroups per CU, 1024 resident at a time. WHOLE rounds of them are what counts
    // (measured on the stem layer, 8 ragged members: 1024 -> 732 us, 1536 -> 837, 2048 -> 743, 3072 -> 735; the same picture on the
    // mid-size and the generator's layers), so ONE round: the fewest k-splits, i.e. the smallest slab (50 instead of 113 MB on the
    // stem) and half the reduce time, at the same or a better kernel time.
    const long wg_env = tun().wgrad_target, s_cap = tun().wgrad_scap;
    // (768 instead of 1024 — a quarter less slab — was worth 0.02 ms on the benchmark's pyramid launches and cost the full-clip D pass
    // 0.2 ms: its weight-gradient launches want whole rounds, 3.26 vs 3.05 ms; the default stays)
    const long wg_target = wg_env ? wg_env : 1024;
    long S = base >= 1024 ? 1 : (wg_target + base - 1) / base;      // (the tiles of a big weight fill the chip on their own)
    const long mincps = tun().wgrad_min_cps > 0 ? tun().wgrad_min_cps : 8;
    long maxS = (nch + mincps - 1) / mincps;      // at least 4 chunks (128 voxels) per split (8 -> 4: -0.10 ms per iteration, the ~25 small launches are latency chains)
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    if (S > s_cap) S = s_cap;
    // wave quantisation: 1024 workgroups are resident at a time; a last round that is under a fifth full (2052 = 2 x 1024 + 4)
    // costs most of a round for little work: give those workgroups' chunks to the full rounds instead
    const bool no_q = !tun().wgrad_quantise;
    const long rounds = (base * S) / 1024, tail = (base * S) % 1024;
    if (!no_q && rounds >= 1 && tail > 0 && tail * 5 <= 1024 && (rounds * 1024) / base >= 1) S = (rounds * 1024) / base;
    p.cps = (int)((nch + S - 1) / S);
    p.S = (int)((nch + p.cps - 1) / p.cps);
    return true;
}

extern "C" int64_t t2v_conv_wgrad_grouped_slab_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD,
                                                      int kH, int kW) {
    WGroupTable tab;
    WgradPlan p;
    if (!build_wtable(groups, ngroups, Cin, Cout, kD, kH, kW, false, tab, p)) return T2V_EINVAL;
    return (int64_t)p.S * p.nlive * Cout * Cin;
}

// Launch-plan query (no launch) for t2v_conv_wgrad_grouped[_bias]: out[0] kernel (0 per-tap 64x64 tiles, 1 (tap, ci) column
// tiles for Cin < 64, 2 three-tap kernel rows, 3 the TN product on 1x1x1 maps, 4 the narrow-input streaming kernel), out[1] k-splits S, out[2] 32-voxel chunks per split, out[3] slab slots,
// out[4] reduce kernel (0 per-64-pairs, 1 the many-splits small-weight form), out[5] workgroups of the main launch.
static void fill_wgrad_plan(int Cin, int Cout, const WgradPlan& p, int32_t* out) {
    const long tiles = (long)((Cout + 63) / 64) * ((Cin + 63) / 64);
    out[0] = p.gemm ? 3 : p.rows3 ? 2 : p.thin ? 4 : (Cin < 64 ? 1 : 0);
    out[1] = p.S;
    out[2] = p.cps;
    out[3] = p.nlive;
    out[4] = ((long)Cout * Cin <= 16384 && p.S >= 16) ? 1 : 0;
    if (p.gemm) { out[5] = (int32_t)((long)((Cout + 127) / 128) * ((Cin + 127) / 128) * p.S); return; }
    if (p.thin) { out[5] = (int32_t)((long)((Cout + 63) / 64) * p.S); return; }
    out[5] = (int32_t)((p.rows3 ? tiles * p.nrows : Cin < 64 ? (long)((Cout + 63) / 64) * (((long)p.nlive * Cin + 63) / 64)
                                                              : tiles * p.nlive) * p.S);
}
extern "C" int t2v_conv_wgrad_plan(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW,
                                   int32_t* out) {
    WGroupTable tab;
    WgradPlan p;
    if (!out || !build_wtable(groups, ngroups, Cin, Cout, kD, kH, kW, false, tab, p)) return T2V_EINVAL;
    fill_wgrad_plan(Cin, Cout, p, out);
    return T2V_OK;
}

extern "C" int64_t t2v_channel_sum_grouped_ws_floats(const t2v_conv_group* groups, int ngroups, int C);
extern "C" int t2v_channel_sum_grouped(const t2v_conv_group* groups, int ngroups, int C, float* out, float* ws, int accum, void* stream);

// floats the bias part needs behind the weight-gradient slab: S x Cout partial sums (every weight-gradient kernel sums the
// dL/dy tiles it stages on the side)
static int64_t wgrad_bias_extra(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, const WgradPlan& p) {
    return (int64_t)p.S * Cout;
}
extern "C" int64_t t2v_conv_wgrad_grouped_bias_slab_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD,
                                                           int kH, int kW) {
    WGroupTable tab;
    WgradPlan p;
    if (!build_wtable(groups, ngroups, Cin, Cout, kD, kH, kW, false, tab, p)) return T2V_EINVAL;
    return (int64_t)p.S * p.nlive * Cout * Cin + wgrad_bias_extra(groups, ngroups, Cin, Cout, p);
}

static int wgrad_impl(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW, float* dw,
                      float* dbias, float* slab, int flags, void* stream, t2v_wgrad_src* out_src = nullptr);
extern "C" int t2v_conv_wgrad_grouped(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW,
                                      float* dw, float* slab, int flags, void* stream) {
    return wgrad_impl(groups, ngroups, Cin, Cout, kD, kH, kW, dw, nullptr, slab, flags, stream);
}
extern "C" int t2v_conv_wgrad_grouped_bias(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW,
                                           float* dw, float* dbias, float* slab, int flags, void* stream) {
    if (!dbias) return T2V_EINVAL;
    return wgrad_impl(groups, ngroups, Cin, Cout, kD, kH, kW, dw, dbias, slab, flags, stream);
}
static int wgrad_impl(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH, int kW, float* dw,
                      float* dbias, float* slab, int flags, void* stream, t2v_wgrad_src* out_src) {
    WGroupTable tab;
    WgradPlan p;
    if ((!dw && !out_src) || !slab || !build_wtable(groups, ngroups, Cin, Cout, kD, kH, kW, true, tab, p)) return T2V_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* bias_part = slab + (size_t)p.S * p.nlive * Cout * Cin;         // behind the weight-gradient slab
    const bool bias_fused = true;                                         // all three weight-gradient kernels sum dL/dy on the side
#ifdef T2V_ABLATION
    if (const char* e = getenv("T2V_DEBUG_FLAGS")) flags |= atoi(e);     // developer ablations (wrong results)
#endif
    const int T = kD * kH * kW;
    TapMap map;
    LiveTaps live;
    live.n = 0;
    for (int t = 0; t < T2V_MAX_TAPS; ++t) {
        live.t[t] = 0;
        map.j[t] = -1;
    }
    LiveRows lrows;
    lrows.n = 0;
    for (int r = 0; r < 9; ++r) lrows.r[r] = 0;
    if (p.rows3) {
        for (int r = 0; r < kD * kH; ++r)
            if ((p.liverows >> r) & 1u) {
                for (int c = 0; c < 3; ++c) map.j[r * 3 + c] = lrows.n * 3 + c;
                lrows.r[lrows.n++] = (int8_t)r;
            }
        live.n = lrows.n * 3;
    } else {
        for (int t = 0; t < T; ++t)
            if ((p.live >> t) & 1u) { map.j[t] = live.n; live.t[live.n++] = (int8_t)t; }
    }
    double flops = 0;
    for (int i = 0; i < ngroups; ++i) {
        int live_g = 0;
        for (int a = 0; a < kD; ++a) for (int b = 0; b < kH; ++b) for (int c = 0; c < kW; ++c) {
            const int dz = a - kD / 2, dy = b - kH / 2, dx = c - kW / 2;
            if (!((groups[i].D == 1 && dz) || (groups[i].H == 1 && dy) || (groups[i].W == 1 && dx))) ++live_g;
        }
        flops += 2.0 * (double)groups[i].N * out_frames(groups[i]) * groups[i].H * groups[i].W * Cout * Cin * live_g;
    }
    // grid.y runs over the taps at least one member can touch; the others are written as zeros by the reduce
    {
        ProfScope prof(1, flops, s, p.nchunks * WG_BK, Cin, Cout, live.n, ngroups, p.S);
        int32_t plan_[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
        fill_wgrad_plan(Cin, Cout, p, plan_);
        plan_[6] = ((flags & T2V_CONV_BF16) && !p.gemm && (p.rows3 || Cin >= 64)) ? 1 : 0;   // conv_wgrad3_kernel<true> / conv_wgrad_bf16_kernel
        plan_[7] = p.strided ? 1 : 0;                                                  // dL/dy on the even frames (dstride = 2)
        ProfScope::set_plan(plan_, 8);
        if (p.gemm) {                          // (bf16-compute mode too: the fp32 TN product beats the voxel-major bf16 per-tap kernel here)
            dim3 grid((unsigned)(((Cout + 127) / 128) * ((Cin + 127) / 128)), 1u, (unsigned)p.S);
            T2V_LAUNCH_PROF(conv_wgrad_gemm_kernel, grid, dim3(256), 0, s, tab, slab, Cin, Cout, flags, p.cps,
                            dbias ? bias_part : (float*)nullptr);
        } else if (p.rows3) {
            dim3 grid((unsigned)(((Cout + 63) / 64) * ((Cin + 63) / 64)), (unsigned)lrows.n, (unsigned)p.S);
            if (flags & T2V_CONV_BF16)
                T2V_LAUNCH_PROF(conv_wgrad3_kernel<true>, grid, dim3(256), 0, s, tab, slab, Cin, Cout, kD, kH, flags, p.cps, lrows,
                                dbias ? bias_part : (float*)nullptr);
            else
                T2V_LAUNCH_PROF(conv_wgrad3_kernel<false>, grid, dim3(256), 0, s, tab, slab, Cin, Cout, kD, kH, flags, p.cps, lrows,
                                dbias ? bias_part : (float*)nullptr);
        } else if (Cin < 64 && p.thin) {
            dim3 grid((unsigned)((Cout + 63) / 64), 1u, (unsigned)p.S);
            T2V_LAUNCH_PROF(conv_wgrad_thin_kernel, grid, dim3(256), 0, s, tab, slab, Cin, Cout, T, kH, kW, flags, p.cps, live,
                            dbias ? bias_part : (float*)nullptr);
        } else if (Cin < 64) {
            dim3 grid((unsigned)(((Cout + 63) / 64) * ((live.n * Cin + 63) / 64)), 1u, (unsigned)p.S);
            T2V_LAUNCH_PROF(conv_wgrad_cols_kernel, grid, dim3(256), 0, s, tab, slab, Cin, Cout, T, kH, kW, flags, p.cps, live,
                            dbias ? bias_part : (float*)nullptr);
        } else {
            dim3 grid((unsigned)(((Cout + 63) / 64) * ((Cin + 63) / 64)), (unsigned)live.n, (unsigned)p.S);
            if (flags & T2V_CONV_BF16)
                T2V_LAUNCH_PROF(conv_wgrad_bf16_kernel, grid, dim3(256), 0, s, tab, slab, Cin, Cout, T, kH, kW, flags, p.cps, live,
                                dbias ? bias_part : (float*)nullptr);
            else
                T2V_LAUNCH_PROF(conv_wgrad_kernel, grid, dim3(256), 0, s, tab, slab, Cin, Cout, T, kH, kW, flags, p.cps, live,
                                dbias ? bias_part : (float*)nullptr);
        }
    }
    int st = launch_status();
    if (st) return st;
    if (out_src) {                  // deferred reduce (t2v_wgrad_reduce_multi): describe the partial sums instead of summing them
        out_src->slab = slab;
        out_src->bias_slab = (dbias && bias_fused) ? bias_part : (const float*)nullptr;
        out_src->ntaps = live.n;
        out_src->S = p.S;
        out_src->tap_stride = (int64_t)Cout * Cin;
        out_src->split_stride = (int64_t)live.n * Cout * Cin;
        for (int t = 0; t < T2V_MAX_TAPS; ++t) out_src->map[t] = (int8_t)(t < T ? map.j[t] : -1);
        return T2V_OK;
    }
    const float* bias_in = (dbias && bias_fused) ? bias_part : (const float*)nullptr;   // summed by one workgroup of the reduce below
    const int accum_bias = (flags & T2V_CONV_ACCUM_BIAS) ? 1 : 0;
    const long CoCi = (long)Cout * Cin;
    ProfScope prof2(2, 0.0, s, CoCi, Cin, Cout, T, live.n, p.S);
    if (CoCi <= 16384 && p.S >= 16)
        T2V_LAUNCH_PROF(wgrad_reduce_small_kernel, dim3((unsigned)((CoCi + 63) / 64), (unsigned)T), dim3(256), 0, s, slab, dw, CoCi, T,
                   live.n, p.S, map, (flags & T2V_CONV_ACCUM) ? 1 : 0, bias_in, dbias, Cout, accum_bias);
    else
        T2V_LAUNCH_PROF(wgrad_reduce_kernel, dim3((unsigned)((CoCi + 63) / 64)), dim3(256), 0, s, slab, dw, CoCi, T, live.n, p.S, map,
                   (flags & T2V_CONV_ACCUM) ? 1 : 0, bias_in, dbias, Cout, accum_bias);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// Deferred, batched reduction of weight-gradient partial sums. A backward pass issues one weight-gradient launch per
// layer (and more for the second-order terms of the gradient penalty); summing each one's k-split slab right away costs a
// launch of ~10 us of pure latency per layer (60 per iteration). t2v_conv_wgrad_grouped_partial runs the main kernel only and
// describes its slab; t2v_wgrad_reduce_multi sums every pending slab of every parameter in ONE launch driven by a
// device-resident table of destinations, each listing its sources in the order they were produced (fixed summation order:
// source by source, splits in the same 4-way interleave as the single reduce kernels).
// ------------------------------------------------------------------------------------------------
extern "C" int t2v_conv_wgrad_grouped_partial(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int kH,
                                              int kW, float* slab, int want_bias, int flags, t2v_wgrad_src* out_src,
                                              void* stream) {
    if (!out_src) return T2V_EINVAL;
    WGroupTable tab;
    WgradPlan p;
    if (!build_wtable(groups, ngroups, Cin, Cout, kD, kH, kW, false, tab, p)) return T2V_EINVAL;
    float dummy;                                                    // (non-null marker: the bias side-sums are wanted)
    return wgrad_impl(groups, ngroups, Cin, Cout, kD, kH, kW, nullptr, want_bias ? &dummy : nullptr, slab, flags, stream, out_src);
}
// ------------------------------------------------------------------------------------------------
// Pooled convolution: host side (kernels: conv_pool_fwd_kernel, conv_pool_dgrad_kernel, conv_pool_wgrad_kernel; the
// pointwise halves t2v_pool_boxsum / t2v_pool_unbox live in pointwise.hip).
// ------------------------------------------------------------------------------------------------
struct PoolPlan { int S; long tiles; bool vecb; };
static bool pool_member_ok(const t2v_conv_group& g, bool need_ptrs) {
    if (need_ptrs && (!g.x || !g.y)) return false;
    if (g.N < 1 || g.H < 2 || g.W < 2 || (g.H & 1) || (g.W & 1)) return false;
    if (g.dstride == 0) return g.D == 1;
    if (g.dstride != 1 && g.dstride != 2) return false;
    return g.D >= 2 && !(g.D & 1);
}
static inline long pool_rows(const t2v_conv_group& g) { return (long)g.N * (g.dstride ? g.D / 2 : 1) * (g.H / 2) * (g.W / 2); }
static inline long pool_padded(const t2v_conv_group& g) { return (long)(g.dstride ? g.D + 1 : 1) * (g.H + 1) * (g.W + 2); }
static inline long pool_grid(const t2v_conv_group& g) { return (long)(g.dstride ? g.D / 2 + 1 : 1) * (g.H / 2 + 1) * (g.W / 2 + 1); }

// forward: rows = pooled voxels, 64-voxel x 64-channel tiles, rounds of (kernel row, 32 channels)
static bool build_pool_fwd(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, bool need_ptrs, GroupTable& tab, PoolPlan& p) {
    if (!groups || ngroups < 1 || ngroups > T2V_MAX_GROUPS || Cin < 32 || (Cin % 32) || Cout < 4 || (Cout % 4)) return false;
    long mt = 0, ot = 0;
    int max_rows = 1;
    tab.n = ngroups;
    for (int i = 0; i < ngroups; ++i) {
        const t2v_conv_group& g = groups[i];
        if (!pool_member_ok(g, need_ptrs)) return false;
        if ((long)g.N * pool_padded(g) * Cin >= (1L << 29) || pool_rows(g) * Cout >= (1L << 31)) return false;     // 32-bit byte offsets
        if (g.ntaps != 9 && g.ntaps != 27) return false;
        if (g.dstride == 0 && g.ntaps != 9) return false;
        for (int r = 0; r < g.ntaps / 3; ++r)
            for (int d = 0; d < 3; ++d) {
                const int t = r * 3 + d;
                if (g.dx[t] != d - 1 || g.dz[t] != g.dz[r * 3] || g.dy[t] != g.dy[r * 3] || g.dz[t] < -1 || g.dz[t] > 1 || g.dy[t] < -1 ||
                    g.dy[t] > 1 || g.widx[t] < 0 || g.widx[t] >= T2V_MAX_TAPS)
                    return false;
                if (g.dstride == 0 && g.dz[t] != 0) return false;
            }
        if (g.ntaps / 3 > max_rows) max_rows = g.ntaps / 3;
        tab.g[i] = g;
        tab.tile_start[i] = (int32_t)mt;
        tab.out_start[i] = ot;
        mt += (pool_rows(g) + 63) / 64;
        ot += pool_rows(g) * Cout;
    }
    for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) { tab.tile_start[i] = (int32_t)mt; tab.out_start[i] = ot; }
    p.tiles = mt * ((Cout + 63) / 64);
    p.vecb = (Cout % 4) == 0;
    const long rounds = (long)max_rows * (Cin / 32);
    long S = 1;
    if (p.tiles < 384 && rounds > tun().nosplit_chunks) {          // same rule as build_table: fill ~768 workgroup slots
        S = (768 + p.tiles - 1) / p.tiles;
        if (S > rounds / 2) S = rounds / 2;
        if (S > 64) S = 64;
        while (S > 1 && (double)S * ot * 4.0 > 256e6) --S;
        if (S < 1) S = 1;
    }
    p.S = (int)S;
    return true;
}
// the double-buffered forms of the pooled GEMMs (16-channel rounds, one barrier per round; see conv_igemm_strip3_kernel): the forward
// takes it by default (-8 %), the data gradient does not (+3 % on its 2-8 round loops). Plan queries report K chunk 16, KS 2 for them.
static bool pool_fwd_db() { static const bool v = env_long("T2V_POOL_FWD_DB", 1) != 0; return v; }
static bool pool_dgrad_db() { static const bool v = env_long("T2V_POOL_DGRAD_DB", 0) != 0; return v; }
extern "C" int64_t t2v_pool_conv_fwd_ws_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout) {
    GroupTable tab;
    PoolPlan p;
    if (!build_pool_fwd(groups, ngroups, Cin, Cout, false, tab, p)) return T2V_EINVAL;
    return p.S > 1 ? (int64_t)p.S * tab.out_start[ngroups] : 0;
}
extern "C" int t2v_pool_conv_fwd(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, const float* wp, const float* bias,
                                 float* ws, int flags, void* stream) {
    GroupTable tab;
    PoolPlan p;
    if (!wp || !build_pool_fwd(groups, ngroups, Cin, Cout, true, tab, p)) return T2V_EINVAL;
    if (flags & ~(T2V_CONV_BIAS | T2V_CONV_ACCUM)) return T2V_EINVAL;
    if (p.S > 1 && !ws) return T2V_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    double flops = 0;
    long Mtot = 0;
    int taps = 0;
    for (int i = 0; i < ngroups; ++i) {
        flops += 2.0 * (double)pool_rows(groups[i]) * Cout * Cin * groups[i].ntaps;
        Mtot += pool_rows(groups[i]);
        if (groups[i].ntaps > taps) taps = groups[i].ntaps;
    }
    {
        ProfScope prof(0, flops, s, Mtot, Cin, Cout, taps, ngroups, p.S);
        int32_t plan_[8] = {9, 64, 64, pool_fwd_db() ? 16 : 32, 1, p.vecb ? 1 : 0, pool_fwd_db() ? 2 : 1, p.S};
        ProfScope::set_plan(plan_, 8);
        dim3 grid((unsigned)tab.tile_start[tab.n], (unsigned)((Cout + 63) / 64), (unsigned)p.S);
        if (pool_fwd_db()) T2V_LAUNCH_PROF((conv_pool_fwd_kernel<64, true>), grid, dim3(256), 0, s, tab, wp, bias, ws, Cin, Cout, flags, p.S);
        else
        T2V_LAUNCH_PROF(conv_pool_fwd_kernel<64>, grid, dim3(256), 0, s, tab, wp, bias, ws, Cin, Cout, flags, p.S);
    }
    int st = launch_status();
    if (st) return st;
    if (p.S > 1) {
        for (int i = 0; i < ngroups; ++i) {                          // the reduce decodes the bias channel from the OUTPUT extents
            tab.g[i].D = groups[i].dstride ? groups[i].D / 2 : 1;
            tab.g[i].H = groups[i].H / 2;
            tab.g[i].W = groups[i].W / 2;
        }
        long blocks = (tab.out_start[ngroups] + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        ProfScope prof_r(4, 0.0, s, Mtot, Cin, Cout, taps, ngroups, p.S);
        T2V_LAUNCH_PROF(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tab, ws, bias, p.S, Cout, flags);
    }
    return launch_status();
}

// data gradient: rows = voxels of the padded grid, one (ct, cy) class pair per workgroup (blockIdx.y & 3)
static bool build_pool_dgrad(const t2v_conv_group* groups, int ngroups, int K, int C, bool need_ptrs, GroupTable& tab, PoolPlan& p) {
    if (!groups || ngroups < 1 || ngroups > T2V_MAX_GROUPS || K < 32 || (K % 32) || C < 4 || (C % 4)) return false;
    long mt = 0;
    tab.n = ngroups;
    for (int i = 0; i < ngroups; ++i) {
        const t2v_conv_group& g = groups[i];
        if (!pool_member_ok(g, need_ptrs)) return false;
        if (pool_rows(g) * K >= (1L << 29) || (long)g.N * pool_grid(g) * C * 8 >= (1L << 31)) return false;
        for (int f = 0; f < 27; ++f) {
            const bool live = g.dstride != 0 || (f / 9) == 1;
            if (live && (g.widx[f] < 0 || g.widx[f] >= T2V_MAX_TAPS)) return false;
        }
        tab.g[i] = g;
        tab.tile_start[i] = (int32_t)mt;
        tab.out_start[i] = 0;
        mt += ((long)g.N * pool_grid(g) + 63) / 64;
    }
    for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) { tab.tile_start[i] = (int32_t)mt; tab.out_start[i] = 0; }
    p.tiles = mt * 4 * ((C + 63) / 64);
    p.vecb = (C % 4) == 0;
    // k-split over the channel blocks when the launch cannot fill the chip: every split writes its own set of 8 planes
    long S = 1;
    const long ncb = K / 32;
    // (tiles counts all four class pairs although the pairs of an odd time class leave at once on members without a time axis and
    //  the class (1, 1) runs a quarter of the rounds of (0, 0): up to 576 tiles two sets still pay — 400 tiles x 64 rounds, the
    //  full-clip discriminator's 512 -> 256 block, 192 -> us)
    static const long split_below = env_long("T2V_POOL_DGRAD_SPLIT_BELOW", 576);
    if (p.tiles < split_below && ncb >= 4) {
        S = (768 + p.tiles - 1) / p.tiles;
        if (S > ncb / 2) S = ncb / 2;
        if (S > 8) S = 8;
        if (S < 1) S = 1;
    }
    p.S = (int)S;
    return true;
}
// number of plane SETS t2v_pool_conv_dgrad writes for these members (its k-split count): the caller allocates
// [S][8][N, C, Dq, Hq, Wq] per member and hands S to t2v_pool_unbox (job.relu), which sums the sets
extern "C" int t2v_pool_conv_dgrad_splits(const t2v_conv_group* groups, int ngroups, int K, int C) {
    GroupTable tab;
    PoolPlan p;
    if (!build_pool_dgrad(groups, ngroups, K, C, false, tab, p)) return T2V_EINVAL;
    return p.S;
}
extern "C" int t2v_pool_conv_dgrad(const t2v_conv_group* groups, int ngroups, int K, int C, const float* wp, void* stream) {
    GroupTable tab;
    PoolPlan p;
    if (!wp || !build_pool_dgrad(groups, ngroups, K, C, true, tab, p)) return T2V_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    double flops = 0;
    long Mtot = 0;
    for (int i = 0; i < ngroups; ++i) {
        flops += 2.0 * (double)pool_rows(groups[i]) * K * C * (groups[i].dstride ? 27 : 9);
        Mtot += (long)groups[i].N * pool_grid(groups[i]);
    }
    ProfScope prof(0, flops, s, Mtot, K, C, 27, ngroups, p.S);
    int32_t plan_[8] = {10, 64, 64, pool_dgrad_db() ? 16 : 32, 1, p.vecb ? 1 : 0, pool_dgrad_db() ? 2 : 1, p.S};
    ProfScope::set_plan(plan_, 8);
    dim3 grid((unsigned)tab.tile_start[tab.n], (unsigned)(4 * ((C + 63) / 64)), (unsigned)p.S);
    if (pool_dgrad_db()) T2V_LAUNCH_PROF(conv_pool_dgrad_kernel<true>, grid, dim3(256), 0, s, tab, wp, K, C);
    else T2V_LAUNCH_PROF(conv_pool_dgrad_kernel<false>, grid, dim3(256), 0, s, tab, wp, K, C);
    return launch_status();
}

// weight gradient: chunks of 32 pooled voxels, one kernel row (dz, dy) per workgroup, k-split like build_wtable
static bool build_pool_wtable(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, bool need_ptrs, WGroupTable& tab, WgradPlan& p) {
    if (!groups || ngroups < 1 || ngroups > T2V_MAX_GROUPS || Cin < 1 || Cout < 1 || (kD != 1 && kD != 3)) return false;
    long nch = 0;
    bool any_t = false;
    tab.n = ngroups;
    for (int i = 0; i < ngroups; ++i) {
        const t2v_conv_group& g = groups[i];
        if (!pool_member_ok(g, need_ptrs)) return false;
        if ((long)g.N * pool_padded(g) * Cin >= (1L << 29) || pool_rows(g) * Cout >= (1L << 29)) return false;
        any_t = any_t || g.dstride != 0;
        if (kD == 1 && g.dstride != 0) return false;             // a [.,.,3,3] weight has no time taps
        tab.g[i] = g;
        tab.chunk_start[i] = (int32_t)nch;
        nch += (pool_rows(g) + WG_BK - 1) / WG_BK;
    }
    for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) tab.chunk_start[i] = (int32_t)nch;
    p.nchunks = nch;
    p.rows3 = true;
    p.strided = false;
    p.liverows = any_t ? 0x1ffu : 0x038u;            // kernel rows (dz, dy): all nine, or the dz = 0 ones
    p.nrows = any_t ? 9 : 3;
    p.live = any_t ? 0x7ffffffu : (0x1ffu << 9);
    p.nlive = p.nrows * 3;
    const long base = (long)((Cout + 63) / 64) * ((Cin + 63) / 64) * p.nrows;
    const long wg_env = tun().wgrad_target, s_cap = tun().wgrad_scap;
    const long wg_target = wg_env ? wg_env : 1024;
    long S = base >= 1024 ? 1 : (wg_target + base - 1) / base;
    // at least 4 chunks (128 pooled voxels) per split, as for the un-pooled kernels (8 -> 4: -0.04 ms per iteration once the kernel
    // ran four workgroups per CU; no better before)
    static const long pool_mincps = env_long("T2V_POOL_WGRAD_MINCPS", 4);
    long maxS = (nch + pool_mincps - 1) / pool_mincps;
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    if (S > s_cap) S = s_cap;
    const long rounds = (base * S) / 1024, tail = (base * S) % 1024;
    if (tun().wgrad_quantise && rounds >= 1 && tail > 0 && tail * 5 <= 1024 && (rounds * 1024) / base >= 1) S = (rounds * 1024) / base;
    p.cps = (int)((nch + S - 1) / S);
    p.S = (int)((nch + p.cps - 1) / p.cps);
    return true;
}
extern "C" int64_t t2v_pool_conv_wgrad_slab_floats(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, int want_bias) {
    WGroupTable tab;
    WgradPlan p;
    if (!build_pool_wtable(groups, ngroups, Cin, Cout, kD, false, tab, p)) return T2V_EINVAL;
    return (int64_t)p.S * p.nlive * Cout * Cin + (want_bias ? (int64_t)p.S * Cout : 0);
}
static int pool_wgrad_impl(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, float* dw, float* dbias, bool want_bias,
                           float* slab, int flags, void* stream, t2v_wgrad_src* out_src) {
    WGroupTable tab;
    WgradPlan p;
    if ((!dw && !out_src) || !slab || !build_pool_wtable(groups, ngroups, Cin, Cout, kD, true, tab, p)) return T2V_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* bias_part = slab + (size_t)p.S * p.nlive * Cout * Cin;
    const int T = kD * 9;
    TapMap map;
    for (int t = 0; t < T2V_MAX_TAPS; ++t) map.j[t] = -1;
    LiveRows lrows;
    lrows.n = 0;
    for (int r = 0; r < 9; ++r) lrows.r[r] = 0;
    for (int r = 0; r < 9; ++r)                              // kernel rows (dz, dy) in the 3x3x3 numbering the kernel decodes
        if ((p.liverows >> r) & 1u) {
            const int rw = kD == 3 ? r : r - 3;              // ... and in the weight's own (a [.,.,3,3] weight only has the dz = 0 rows)
            for (int c = 0; c < 3; ++c) map.j[rw * 3 + c] = lrows.n * 3 + c;
            lrows.r[lrows.n++] = (int8_t)r;
        }
    double flops = 0;
    for (int i = 0; i < ngroups; ++i) flops += 2.0 * (double)pool_rows(groups[i]) * Cout * Cin * (groups[i].dstride ? 27 : 9);
    {
        ProfScope prof(1, flops, s, p.nchunks * WG_BK, Cin, Cout, p.nlive, ngroups, p.S);
        int32_t plan_[8] = {11, p.S, p.cps, p.nlive, ((long)Cout * Cin <= 16384 && p.S >= 16) ? 1 : 0,
                            (int32_t)((long)((Cout + 63) / 64) * ((Cin + 63) / 64) * p.nrows * p.S), 0, 0};
        ProfScope::set_plan(plan_, 8);
        dim3 grid((unsigned)(((Cout + 63) / 64) * ((Cin + 63) / 64)), (unsigned)lrows.n, (unsigned)p.S);
        T2V_LAUNCH_PROF(conv_pool_wgrad_kernel, grid, dim3(256), 0, s, tab, slab, Cin, Cout, p.cps, lrows, want_bias ? bias_part : (float*)nullptr);
    }
    int st = launch_status();
    if (st) return st;
    if (out_src) {
        out_src->slab = slab;
        out_src->bias_slab = want_bias ? bias_part : (const float*)nullptr;
        out_src->ntaps = p.nlive;
        out_src->S = p.S;
        out_src->tap_stride = (int64_t)Cout * Cin;
        out_src->split_stride = (int64_t)p.nlive * Cout * Cin;
        for (int t = 0; t < T2V_MAX_TAPS; ++t) out_src->map[t] = (int8_t)(t < T ? map.j[t] : -1);
        return T2V_OK;
    }
    const float* bias_in = want_bias ? bias_part : (const float*)nullptr;
    const int accum_bias = (flags & T2V_CONV_ACCUM_BIAS) ? 1 : 0;
    const long CoCi = (long)Cout * Cin;
    ProfScope prof2(2, 0.0, s, CoCi, Cin, Cout, T, p.nlive, p.S);
    if (CoCi <= 16384 && p.S >= 16)
        T2V_LAUNCH_PROF(wgrad_reduce_small_kernel, dim3((unsigned)((CoCi + 63) / 64), (unsigned)T), dim3(256), 0, s, slab, dw, CoCi, T,
                   p.nlive, p.S, map, (flags & T2V_CONV_ACCUM) ? 1 : 0, bias_in, dbias, Cout, accum_bias);
    else
        T2V_LAUNCH_PROF(wgrad_reduce_kernel, dim3((unsigned)((CoCi + 63) / 64)), dim3(256), 0, s, slab, dw, CoCi, T, p.nlive, p.S, map,
                   (flags & T2V_CONV_ACCUM) ? 1 : 0, bias_in, dbias, Cout, accum_bias);
    return launch_status();
}
extern "C" int t2v_pool_conv_wgrad(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, float* dw, float* dbias,
                                   float* slab, int flags, void* stream) {
    if (!dw) return T2V_EINVAL;
    return pool_wgrad_impl(groups, ngroups, Cin, Cout, kD, dw, dbias, dbias != nullptr, slab, flags, stream, nullptr);
}
extern "C" int t2v_pool_conv_wgrad_partial(const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int kD, float* slab,
                                           int want_bias, int flags, t2v_wgrad_src* out_src, void* stream) {
    if (!out_src) return T2V_EINVAL;
    return pool_wgrad_impl(groups, ngroups, Cin, Cout, kD, nullptr, nullptr, want_bias != 0, slab, flags, stream, out_src);
}
// what: 0 forward, 1 data gradient (Cin = K of dL/dy, Cout = channels of the planes), 2 weight gradient
extern "C" int t2v_pool_conv_plan(int what, const t2v_conv_group* groups, int ngroups, int Cin, int Cout, int32_t* out) {
    if (!out) return T2V_EINVAL;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (what == 0 || what == 1) {
        GroupTable tab;
        PoolPlan p;
        if (!(what == 0 ? build_pool_fwd(groups, ngroups, Cin, Cout, false, tab, p) : build_pool_dgrad(groups, ngroups, Cin, Cout, false, tab, p)))
            return T2V_EINVAL;
        const bool db = what == 0 ? pool_fwd_db() : pool_dgrad_db();
        out[0] = what == 0 ? 9 : 10; out[1] = 64; out[2] = 64; out[3] = db ? 16 : 32; out[4] = 1; out[5] = p.vecb ? 1 : 0; out[6] = db ? 2 : 1; out[7] = p.S;
        return T2V_OK;
    }
    if (what == 2) {
        WGroupTable tab;
        WgradPlan p;
        if (!build_pool_wtable(groups, ngroups, Cin, Cout, 3, false, tab, p)) return T2V_EINVAL;
        out[0] = 11; out[1] = p.S; out[2] = p.cps; out[3] = p.nlive; out[4] = ((long)Cout * Cin <= 16384 && p.S >= 16) ? 1 : 0;
        out[5] = (int32_t)((long)((Cout + 63) / 64) * ((Cin + 63) / 64) * p.nrows * p.S);
        return T2V_OK;
    }
    return T2V_EINVAL;
}

extern "C" int t2v_wgrad_dest_bytes(void) { return (int)sizeof(t2v_wgrad_dest); }


__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const t2v_wgrad_dest* __restrict__ table, int ndest) {
    __shared__ __attribute__((aligned(16))) float tile[256 * T2V_MAX_TAPS];
    __shared__ __attribute__((aligned(16))) float part[4][256];
    __shared__ int s_dest;
    if (threadIdx.x == 0) {                  // last destination with block_begin <= blockIdx.x
        int lo = 0, hi = ndest - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        s_dest = lo;
    }
    __syncthreads();
    const t2v_wgrad_dest& d = table[s_dest];
    const int b = (int)blockIdx.x - d.block_begin;
    const long CoCi = d.CoCi;
    const int T = d.T, nsrc = d.nsrc;
    if (d.dbias && b == 0) {                 // bias: one workgroup, sources in order
        for (int co = threadIdx.x; co < d.Cout; co += 256) {
            float v = 0.f;
            bool first = true;
            for (int k = 0; k < nsrc; ++k) {
                const float* bs = d.src[k].bias_slab;
                if (!bs) continue;
                const float vk = slab_sum4(bs + co, (size_t)d.Cout, d.src[k].S);
                v = first ? vk : v + vk;
                first = false;
            }
            if (!first) d.dbias[co] = d.accum_bias ? d.dbias[co] + v : v;
        }
    }
    if (d.kind == 2) {                       // kind 0 with 16-byte loads: 256 pairs x T taps per workgroup, a lane owns 4 pairs
        const long i0 = (long)b * 256;
        const int l4 = (threadIdx.x & 63) * 4, tg = threadIdx.x >> 6;
        const long i = i0 + l4;
        for (int t = tg; t < T; t += 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            bool any = false;
            for (int k = 0; k < nsrc; ++k) {
                const int j = d.src[k].map[t];
                float4 vk = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j >= 0 && i < CoCi) vk = slab_sum_v4(d.src[k].slab + (size_t)j * d.src[k].tap_stride + i, (size_t)d.src[k].split_stride, d.src[k].S);
                any = any || j >= 0;
                if (k == 0) v = vk; else { v.x += vk.x; v.y += vk.y; v.z += vk.z; v.w += vk.w; }
            }
            if (d.tap_major) {
                if (any && i < CoCi) {
                    float* q = d.dw + (size_t)t * CoCi + i;
                    if (d.accum) { v.x += q[0]; v.y += q[1]; v.z += q[2]; v.w += q[3]; }
                    q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w;
                }
            } else {
                tile[(l4 + 0) * T + t] = v.x; tile[(l4 + 1) * T + t] = v.y; tile[(l4 + 2) * T + t] = v.z; tile[(l4 + 3) * T + t] = v.w;
            }
        }
        if (d.tap_major) return;
        __syncthreads();
        long cnt = CoCi - i0;
        if (cnt > 256) cnt = 256;
        const long nval = cnt * T;
        float* p = d.dw + (size_t)i0 * T;
        for (long k = threadIdx.x; k < nval; k += 256) p[k] = d.accum ? p[k] + tile[k] : tile[k];
    } else if (d.kind == 3) {                // kind 1 with 16-byte loads: one workgroup per (256 pairs, tap), the waves split S
        const int t = b % T;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const long i = (long)(b / T) * 256 + lane * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool any = false;
        for (int k = 0; k < nsrc; ++k) {
            const int j = d.src[k].map[t];
            float4 vp = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j >= 0 && i < CoCi) {            // (same summation order as kind 1 and wgrad_reduce_small_kernel)
                const float* p = d.src[k].slab + (size_t)j * d.src[k].tap_stride + i;
                const size_t st = (size_t)d.src[k].split_stride;
                const int S = d.src[k].S;
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                int s = wv;
#pragma unroll 4
                for (; s + 4 < S; s += 8) {
                    const float4 a = *reinterpret_cast<const float4*>(p + (size_t)s * st), c = *reinterpret_cast<const float4*>(p + (size_t)(s + 4) * st);
                    v0.x += a.x; v0.y += a.y; v0.z += a.z; v0.w += a.w;
                    v1.x += c.x; v1.y += c.y; v1.z += c.z; v1.w += c.w;
                }
                if (s < S) {
                    const float4 a = *reinterpret_cast<const float4*>(p + (size_t)s * st);
                    v0.x += a.x; v0.y += a.y; v0.z += a.z; v0.w += a.w;
                }
                vp.x = v0.x + v1.x; vp.y = v0.y + v1.y; vp.z = v0.z + v1.z; vp.w = v0.w + v1.w;
            }
            any = any || j >= 0;
            *reinterpret_cast<float4*>(&part[wv][lane * 4]) = vp;
            __syncthreads();
            if (wv == 0) {
                const float4 p0 = *reinterpret_cast<const float4*>(&part[0][lane * 4]), p1 = *reinterpret_cast<const float4*>(&part[1][lane * 4]);
                const float4 p2 = *reinterpret_cast<const float4*>(&part[2][lane * 4]), p3 = *reinterpret_cast<const float4*>(&part[3][lane * 4]);
                float4 vk;
                vk.x = (p0.x + p1.x) + (p2.x + p3.x); vk.y = (p0.y + p1.y) + (p2.y + p3.y);
                vk.z = (p0.z + p1.z) + (p2.z + p3.z); vk.w = (p0.w + p1.w) + (p2.w + p3.w);
                if (k == 0) v = vk; else { v.x += vk.x; v.y += vk.y; v.z += vk.z; v.w += vk.w; }
            }
            __syncthreads();
        }
        if (wv == 0 && i < CoCi && (any || !d.tap_major)) {
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* q = d.tap_major ? d.dw + (size_t)t * CoCi + i + e : d.dw + (size_t)(i + e) * T + t;
                *q = d.accum ? *q + vv[e] : vv[e];
            }
        }
    } else if (d.kind == 0) {                // 64 (co,ci) pairs x T taps; the 4 waves take taps t = wave, wave+4, ...
        const long i0 = (long)b * 64;
        const int il = threadIdx.x & 63, tg = threadIdx.x >> 6;
        const long i = i0 + il;
        for (int t = tg; t < T; t += 4) {
            float v = 0.f;
            bool any = false;
            for (int k = 0; k < nsrc; ++k) {
                const int j = d.src[k].map[t];
                float vk = 0.f;
                if (j >= 0 && i < CoCi) vk = slab_sum4(d.src[k].slab + (size_t)j * d.src[k].tap_stride + i, (size_t)d.src[k].split_stride, d.src[k].S);
                any = any || j >= 0;
                v = k == 0 ? vk : v + vk;
            }
            if (d.tap_major) {               // dw[t][co][ci]: lane-contiguous, taps no source touches are left as they are
                if (any && i < CoCi) {
                    float* q = d.dw + (size_t)t * CoCi + i;
                    *q = d.accum ? *q + v : v;
                }
            } else tile[il * T + t] = v;
        }
        if (d.tap_major) return;
        __syncthreads();
        long cnt = CoCi - i0;
        if (cnt > 64) cnt = 64;
        const long nval = cnt * T;
        float* p = d.dw + (size_t)i0 * T;
        for (long k = threadIdx.x; k < nval; k += 256) p[k] = d.accum ? p[k] + tile[k] : tile[k];
    } else {                                 // small weights with many splits: one workgroup per (64 pairs, tap), waves split S
        const int t = b % T;
        const long i = (long)(b / T) * 64 + (threadIdx.x & 63);
        const int wv = threadIdx.x >> 6;
        float v = 0.f;
        bool any = false;
        for (int k = 0; k < nsrc; ++k) {
            const int j = d.src[k].map[t];
            float v0 = 0.f, v1 = 0.f;
            if (j >= 0 && i < CoCi) {
                const float* p = d.src[k].slab + (size_t)j * d.src[k].tap_stride + i;
                const size_t st = (size_t)d.src[k].split_stride;
                const int S = d.src[k].S;
                int s = wv;
                for (; s + 4 < S; s += 8) { v0 += p[(size_t)s * st]; v1 += p[(size_t)(s + 4) * st]; }
                if (s < S) v0 += p[(size_t)s * st];
            }
            any = any || j >= 0;
            part[wv][threadIdx.x & 63] = v0 + v1;
            __syncthreads();
            if (wv == 0) {
                const float vk = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
                v = k == 0 ? vk : v + vk;
            }
            __syncthreads();
        }
        if (wv == 0 && i < CoCi && (any || !d.tap_major)) {
            float* q = d.tap_major ? d.dw + (size_t)t * CoCi + i : d.dw + (size_t)i * T + t;
            *q = d.accum ? *q + v : v;
        }
    }
}

// `table`: DEVICE array of ndest t2v_wgrad_dest records; total_blocks = sum of nblocks
// (kind 0: ceil(CoCi / 64) workgroups, kind 1: ceil(CoCi / 64) * T; kinds 2 / 3: the same with 256 pairs per workgroup and
// 16-byte slab loads — they need CoCi % 4 == 0 and 16-byte aligned slabs)
extern "C" int t2v_wgrad_reduce_multi(const void* table, int ndest, int total_blocks, void* stream) {
    if (!table || ndest < 1 || total_blocks < 1) return T2V_EINVAL;
    T2V_LAUNCH(wgrad_reduce_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
               (const t2v_wgrad_dest*)table, ndest);
    return launch_status();
}

// single-tensor entry points
static void kernel_extent(const t2v_conv_geom* g, int T, int& kD, int& kH, int& kW) {
    // T in {1, 3, 9, 27} with the singleton dims leading (Linear: 1, 2-D 3x3: 9 = 1x3x3, 3-D: 27)
    kD = (T == 27) ? 3 : 1;
    kH = (T >= 9) ? 3 : 1;
    kW = (T >= 3) ? 3 : 1;
    (void)g;
}
extern "C" int64_t t2v_conv_wgrad_slab_floats(const t2v_conv_geom* g, int T) {
    if (!geom_ok(g) || (T != 1 && T != 3 && T != 9 && T != 27)) return T2V_EINVAL;
    t2v_conv_group q = group_of(g, nullptr, nullptr);
    int kD, kH, kW;
    kernel_extent(g, T, kD, kH, kW);
    return t2v_conv_wgrad_grouped_slab_floats(&q, 1, g->Cin, g->Cout, kD, kH, kW);
}
extern "C" int t2v_conv_wgrad(const float* x, const float* gy, float* dw, float* slab, const t2v_conv_geom* g,
                              const int32_t* taps, int T, int flags, void* stream) {
    if (!x || !gy || !dw || !slab || !geom_ok(g) || (T != 1 && T != 3 && T != 9 && T != 27)) return T2V_EINVAL;
    (void)taps;
    t2v_conv_group q = group_of(g, x, const_cast<float*>(gy));
    int kD, kH, kW;
    kernel_extent(g, T, kD, kH, kW);
    return t2v_conv_wgrad_grouped(&q, 1, g->Cin, g->Cout, kD, kH, kW, dw, slab, flags, stream);
}

// ------------------------------------------------------------------------------------------------
// per-channel sum over (N, S): bias gradient. grid (C, SPLIT): each workgroup sums a contiguous range of
// the N*S elements of its channel; SPLIT > 1 leaves partials in `ws` for the finalize kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int N,
                                                          int C, long S, int accum, int split, float* __restrict__ ws) {
    const int c = blockIdx.x;
    const long total = (long)N * S;
    const long per = (total + split - 1) / split;
    const long e0 = (long)blockIdx.y * per;
    long e1 = e0 + per;
    if (e1 > total) e1 = total;
    float acc = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const long n = e / S, sp = e - n * S;
        acc += x[((size_t)n * C + c) * S + sp];
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (split > 1) ws[(size_t)c * split + blockIdx.y] = red[0];
        else out[c] = accum ? out[c] + red[0] : red[0];
    }
}
__global__ void channel_sum_final_kernel(const float* __restrict__ ws, float* __restrict__ out, int C, int split, int accum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float v = 0.f;
    for (int k = 0; k < split; ++k) v += ws[(size_t)c * split + k];
    out[c] = accum ? out[c] + v : v;
}

static int channel_split(int N, int C, int64_t S) {
    const long total = (long)N * S;
    long sp = (1024 + C - 1) / C;
    if (sp > total / 2048) sp = total / 2048;      // >= 2048 elements per workgroup
    if (sp > 64) sp = 64;
    if (sp < 1) sp = 1;
    return (int)sp;
}
extern "C" int64_t t2v_channel_sum_ws_floats(int N, int C, int64_t S) {
    if (N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    const int sp = channel_split(N, C, S);
    return sp > 1 ? (int64_t)C * sp : 0;
}
extern "C" int t2v_channel_sum(const float* x, float* out, float* ws, int N, int C, int64_t S, int accum, void* stream) {
    if (!x || !out || N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    const int sp = channel_split(N, C, S);
    if (sp > 1 && !ws) return T2V_EINVAL;
    T2V_LAUNCH(channel_sum_kernel, dim3(C, sp), dim3(256), 0, (hipStream_t)stream, x, out, N, C, (long)S, accum, sp, ws);
    if (sp > 1) T2V_LAUNCH(channel_sum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, out, C, sp, accum);
    return launch_status();
}

// bias gradient of a grouped convolution: out[c] = sum over all members, samples and voxels — one launch
struct CSumTable { const float* p[T2V_MAX_GROUPS]; int32_t N[T2V_MAX_GROUPS]; int64_t S[T2V_MAX_GROUPS]; int64_t start[T2V_MAX_GROUPS + 1]; int32_t n; };

__global__ __launch_bounds__(256) void channel_sum_grouped_kernel(const CSumTable tab, float* __restrict__ out, int C, int accum,
                                                                  int split, float* __restrict__ ws) {
    const int c = blockIdx.x;
    const long total = tab.start[tab.n];
    const long per = (total + split - 1) / split;
    const long e0 = (long)blockIdx.y * per;
    long e1 = e0 + per;
    if (e1 > total) e1 = total;
    float acc = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        int gi = 0;
#pragma unroll
        for (int k = 1; k < T2V_MAX_GROUPS; ++k)
            if (k < tab.n && e >= tab.start[k]) gi = k;
        const long le = e - tab.start[gi];
        const long S = tab.S[gi];
        const long n = le / S, sp = le - n * S;
        acc += tab.p[gi][((size_t)n * C + c) * S + sp];
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (split > 1) ws[(size_t)c * split + blockIdx.y] = red[0];
        else out[c] = accum ? out[c] + red[0] : red[0];
    }
}

extern "C" int64_t t2v_channel_sum_grouped_ws_floats(const t2v_conv_group* groups, int ngroups, int C) {
    if (!groups || ngroups < 1 || ngroups > T2V_MAX_GROUPS || C < 1) return T2V_EINVAL;
    long total = 0;
    for (int i = 0; i < ngroups; ++i) total += (long)groups[i].N * groups[i].D * groups[i].H * groups[i].W;
    const int sp = channel_split(1, C, total);
    return sp > 1 ? (int64_t)C * sp : 0;
}
// members: groups[i].x = tensor [N, C, D, H, W]
extern "C" int t2v_channel_sum_grouped(const t2v_conv_group* groups, int ngroups, int C, float* out, float* ws, int accum,
                                       void* stream) {
    if (!groups || !out || ngroups < 1 || ngroups > T2V_MAX_GROUPS || C < 1) return T2V_EINVAL;
    CSumTable tab;
    long total = 0;
    tab.n = ngroups;
    for (int i = 0; i < ngroups; ++i) {
        if (!groups[i].x || groups[i].N < 1) return T2V_EINVAL;
        tab.p[i] = groups[i].x;
        tab.N[i] = groups[i].N;
        tab.S[i] = (int64_t)groups[i].D * groups[i].H * groups[i].W;
        tab.start[i] = total;
        total += (long)groups[i].N * tab.S[i];
    }
    for (int i = ngroups; i <= T2V_MAX_GROUPS; ++i) tab.start[i] = total;
    const int sp = channel_split(1, C, total);
    if (sp > 1 && !ws) return T2V_EINVAL;
    T2V_LAUNCH(channel_sum_grouped_kernel, dim3(C, sp), dim3(256), 0, (hipStream_t)stream, tab, out, C, accum, sp, ws);
    if (sp > 1) T2V_LAUNCH(channel_sum_final_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, out, C, sp, accum);
    return launch_status();
}
