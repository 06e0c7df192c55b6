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
#include <vector>
struct ProfRec { hipEvent_t e0, e1; double flops; int kind; };
static struct {
    bool on = false;
    std::vector<ProfRec> recs;
    size_t used = 0;
} g_prof;

struct ProfScope {
    ProfRec* r = nullptr;
    hipStream_t s;
    ProfScope(int kind, double flops, hipStream_t stream) : s(stream) {
        if (g_prof.on && g_prof.used < g_prof.recs.size()) {
            r = &g_prof.recs[g_prof.used++];
            r->kind = kind;
            r->flops = flops;
            (void)hipEventRecord(r->e0, s);
        }
    }
    ~ProfScope() { if (r) (void)hipEventRecord(r->e1, s); }
};

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
    for (size_t i = 0; i < g_prof.used; ++i) {
        ProfRec& r = g_prof.recs[i];
        if (hipEventSynchronize(r.e1) != hipSuccess) return T2V_ELAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return T2V_ELAUNCH;
        if (r.kind >= 0 && r.kind < nkinds) { out[r.kind * 3] += ms; out[r.kind * 3 + 1] += r.flops; out[r.kind * 3 + 2] += 1.0; }
    }
    int dropped = (g_prof.used >= g_prof.recs.size()) ? 1 : 0;
    g_prof.used = 0;
    return dropped;    // 1: the record pool filled up (totals cover the recorded launches only)
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
struct TapList { int32_t n; int32_t t[T2V_MAX_TAPS]; };

__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                                          int Cout, int Cin, int T, TapList taps, int mode) {
    // mode 0: wp[j][ci][co] = w[co][ci][t_j]      (inner = co)
    // mode 1: wp[j][co][ci] = w[co][ci][T-1-t_j]  (inner = ci)
    // 32x32 tile transpose of the (co,ci) plane through LDS for mode 0; mode 1 is a strided copy.
    __shared__ float tile[32][33];
    const int j = blockIdx.z;
    const int t = mode ? (T - 1 - taps.t[j]) : taps.t[j];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    if (mode == 0) {
        for (int r = ty; r < 32; r += 8) {          // read rows co, lanes ci (stride T)
            int co = co0 + r, ci = ci0 + tx;
            tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * T + t] : 0.f;
        }
        __syncthreads();
        for (int r = ty; r < 32; r += 8) {          // write rows ci, lanes co
            int ci = ci0 + r, co = co0 + tx;
            if (ci < Cin && co < Cout) wp[((size_t)j * Cin + ci) * Cout + co] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < 32; r += 8) {
            int co = co0 + r, ci = ci0 + tx;
            if (co < Cout && ci < Cin) wp[((size_t)j * Cout + co) * Cin + ci] = w[((size_t)co * Cin + ci) * T + t];
        }
    }
}

extern "C" int t2v_pack_weight(const float* w, float* wp, int Cout, int Cin, int T, const int32_t* taps, int ntaps,
                               int mode, void* stream) {
    if (!w || !wp || ntaps < 1 || ntaps > T2V_MAX_TAPS || T > T2V_MAX_TAPS) return T2V_EINVAL;
    TapList tl;
    tl.n = ntaps;
    for (int i = 0; i < ntaps; ++i) {
        if (taps[i] < 0 || taps[i] >= T) return T2V_EINVAL;
        tl.t[i] = taps[i];
    }
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, ntaps);
    T2V_LAUNCH(pack_weight_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wp, Cout, Cin, T, tl, mode);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad implicit GEMM
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_CO, bool FAST>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         const t2v_conv_geom g, const int flags) {
    constexpr int WAVES_M = 4 / WAVES_CO;
    constexpr int WCO = BN / WAVES_CO;      // co extent per wave
    constexpr int WM = BM / WAVES_M;        // m extent per wave
    constexpr int NCO = WCO / 32, NM = WM / 32;
    constexpr int LA = BK * BM / 256, LB = BK * BN / 256;
    constexpr int KSA = 256 / BM, KSB = 256 / BN;   // k stride between a thread's successive loads
    static_assert(NCO >= 1 && NM >= 1, "tile");

    __shared__ float As[BK * BM];
    __shared__ float Bs[BK * BN];
    __shared__ int s_off[T2V_MAX_TAPS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave % WAVES_CO, wm = wave / WAVES_CO;

    const int D = g.D, H = g.H, W = g.W, Cin = g.Cin, Cout = g.Cout;
    const int HW = H * W, DHW = D * HW;
    const int M = g.N * DHW;
    const int m0 = blockIdx.x * BM, co0 = blockIdx.y * BN;
    const int ntaps = g.ntaps;

    if (tid < ntaps) s_off[tid] = g.dz[tid] * HW + g.dy[tid] * W + g.dx[tid];

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
        for (int t = 0; t < ntaps; ++t) {
            int dd = d + g.dz[t], hh = h + g.dy[t], ww = w_ + g.dx[t];
            if ((unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) tapmask |= 1u << t;
        }
    }
    const int cob_l = tid % BN, kb_l = tid / BN;
    const bool co_ok = (co0 + cob_l) < Cout;
    const bool relu_in = flags & T2V_CONV_RELU_IN;

    f32x16 acc[NCO][NM];
#pragma unroll
    for (int i = 0; i < NCO; ++i)
#pragma unroll
        for (int j = 0; j < NM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[LA], rb[LB];
    const int Ktot = ntaps * Cin;
    const int nchunks = (Ktot + BK - 1) / BK;
    __syncthreads();   // s_off visible

    auto load_chunk = [&](int q, int t, int c0) {
        if (FAST) {
            const bool v = (tapmask >> t) & 1u;
            const float* px = x + xbase + (ptrdiff_t)s_off[t] + (size_t)(c0 + ka_l) * DHW;
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                float val = v ? px[(size_t)j * KSA * DHW] : 0.f;
                ra[j] = relu_in ? fmaxf(val, 0.f) : val;
            }
            const float* pw = wp + ((size_t)t * Cin + c0 + kb_l) * Cout + co0 + cob_l;
#pragma unroll
            for (int j = 0; j < LB; ++j) rb[j] = co_ok ? pw[(size_t)j * KSB * Cout] : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                int kk = q * BK + ka_l + j * KSA;
                float val = 0.f;
                if (kk < Ktot) {
                    int tt = kk / Cin, ci = kk - tt * Cin;
                    if ((tapmask >> tt) & 1u) val = x[xbase + (ptrdiff_t)s_off[tt] + (size_t)ci * DHW];
                }
                ra[j] = relu_in ? fmaxf(val, 0.f) : val;
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                int kk = q * BK + kb_l + j * KSB;
                rb[j] = (co_ok && kk < Ktot) ? wp[(size_t)kk * Cout + co0 + cob_l] : 0.f;
            }
        }
    };

    int t_cur = 0, c_cur = 0;
    load_chunk(0, 0, 0);
    for (int q = 0; q < nchunks; ++q) {
        // registers -> LDS
#pragma unroll
        for (int j = 0; j < LA; ++j) As[(ka_l + j * KSA) * BM + ma_l] = ra[j];
#pragma unroll
        for (int j = 0; j < LB; ++j) Bs[(kb_l + j * KSB) * BN + cob_l] = rb[j];
        __syncthreads();
        // issue the next chunk's global loads before the MFMAs (latency hides under compute)
        if (q + 1 < nchunks) {
            c_cur += BK;
            if (FAST && c_cur >= Cin) { c_cur = 0; ++t_cur; }
            load_chunk(q + 1, t_cur, c_cur);
        }
#pragma unroll
        for (int k2 = 0; k2 < BK / 2; ++k2) {
            float a[NCO], b[NM];
            const int krow = k2 * 2 + hi;
#pragma unroll
            for (int i = 0; i < NCO; ++i) a[i] = Bs[krow * BN + wco * WCO + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < NM; ++j) b[j] = As[krow * BM + wm * WM + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < NCO; ++i)
#pragma unroll
                for (int j = 0; j < NM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: rows (registers) = co, columns (lanes) = m
    const bool has_bias = (flags & T2V_CONV_BIAS) && bias != nullptr;
    const bool accum = flags & T2V_CONV_ACCUM;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int m = m0 + wm * WM + j * 32 + l31;
        if (m >= M) continue;
        const int n = m / DHW, sp = m - n * DHW;
        float* py = y + (size_t)n * Cout * DHW + sp;
#pragma unroll
        for (int i = 0; i < NCO; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * WCO + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (co < Cout) {
                    float v = acc[i][j][r];
                    if (has_bias) v += bias[co];
                    float* p = py + (size_t)co * DHW;
                    *p = accum ? (*p + v) : v;
                }
            }
        }
    }
}

template <int BM, int BN, int WAVES_CO>
static int launch_conv(const float* x, const float* wp, const float* bias, float* y, const t2v_conv_geom& g, int flags,
                       hipStream_t s) {
    const long M = (long)g.N * g.D * g.H * g.W;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((g.Cout + BN - 1) / BN));
    ProfScope prof(0, 2.0 * (double)M * g.Cout * g.Cin * g.ntaps, s);      // executed (non-padding-tap) MACs x 2
    if (g.Cin % BK == 0)
        T2V_LAUNCH((conv_igemm_kernel<BM, BN, WAVES_CO, true>), grid, dim3(256), 0, s, x, wp, bias, y, g, flags);
    else
        T2V_LAUNCH((conv_igemm_kernel<BM, BN, WAVES_CO, false>), grid, dim3(256), 0, s, x, wp, bias, y, g, flags);
    return launch_status();
}

static bool geom_ok(const t2v_conv_geom* g) {
    if (!g) return false;
    if (g->N < 1 || g->Cin < 1 || g->Cout < 1 || g->D < 1 || g->H < 1 || g->W < 1) return false;
    if (g->ntaps < 1 || g->ntaps > T2V_MAX_TAPS) return false;
    long M = (long)g->N * g->D * g->H * g->W;
    if (M * (long)(g->Cin > g->Cout ? g->Cin : g->Cout) >= (1L << 31)) return false;   // 32-bit voxel indices
    for (int t = 0; t < g->ntaps; ++t) {
        if (g->dz[t] < -1 || g->dz[t] > 1 || g->dy[t] < -1 || g->dy[t] > 1 || g->dx[t] < -1 || g->dx[t] > 1) return false;
    }
    return true;
}

extern "C" int t2v_conv_fwd(const float* x, const float* wp, const float* bias, float* y, const t2v_conv_geom* g,
                            int flags, void* stream) {
    if (!x || !wp || !y || !geom_ok(g)) return T2V_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long M = (long)g->N * g->D * g->H * g->W;
    const int Cout = g->Cout;
    // tile choice: fill >= ~2 workgroups per CU where the problem allows it
    if (Cout <= 32) return launch_conv<128, 32, 1>(x, wp, bias, y, *g, flags, s);
    const long t128 = ((M + 127) / 128) * ((Cout + 127) / 128);
    const long t12864 = ((M + 127) / 128) * ((Cout + 63) / 64);
    if (Cout >= 128 && t128 >= 1024) return launch_conv<128, 128, 2>(x, wp, bias, y, *g, flags, s);
    if (t12864 >= 1024) return launch_conv<128, 64, 2>(x, wp, bias, y, *g, flags, s);
    return launch_conv<64, 64, 2>(x, wp, bias, y, *g, flags, s);
}

// ------------------------------------------------------------------------------------------------
// weight gradient: per (tap, co-tile, ci-tile, k-split) a 64x64 tile of dW over a range of voxels
// ------------------------------------------------------------------------------------------------
#define WG_BK 32
#define WG_PITCH (WG_BK + 1)

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                         float* __restrict__ slab, const t2v_conv_geom g,
                                                         const int flags, const int chunks_per_split) {
    __shared__ float As[64 * WG_PITCH];   // gy^T tile  [co][m]
    __shared__ float Bs[64 * WG_PITCH];   // x   tile   [ci][m]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wco = wave & 1, wci = wave >> 1;
    const int D = g.D, H = g.H, W = g.W, Cin = g.Cin, Cout = g.Cout;
    const int HW = H * W, DHW = D * HW, M = g.N * DHW;
    const int nco_t = (Cout + 63) / 64;
    const int co0 = (blockIdx.x % nco_t) * 64, ci0 = (blockIdx.x / nco_t) * 64;
    const int j = blockIdx.y;           // geometry tap
    const int split = blockIdx.z;
    const int dz = g.dz[j], dy = g.dy[j], dx = g.dx[j];
    const int off = dz * HW + dy * W + dx;
    const bool relu_in = flags & T2V_CONV_RELU_IN;

    const int ml = tid & 31, rl = tid >> 5;     // 32 m x 8 rows per pass, 8 passes -> 64 rows
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nchunks = (M + WG_BK - 1) / WG_BK;
    const int q0 = split * chunks_per_split;
    int q1 = q0 + chunks_per_split;
    if (q1 > nchunks) q1 = nchunks;

    float ra[8], rb[8];
    auto load_chunk = [&](int q) {
        const int m = q * WG_BK + ml;
        bool mv = m < M, xv = false;
        size_t gbase = 0, xb = 0;
        if (mv) {
            int n = m / DHW, sp = m - n * DHW;
            int d = sp / HW, r = sp - d * HW;
            int h = r / W, w_ = r - h * W;
            gbase = (size_t)n * Cout * DHW + sp;
            int dd = d + dz, hh = h + dy, ww = w_ + dx;
            xv = (unsigned)dd < (unsigned)D && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            xb = (size_t)n * Cin * DHW + sp + (ptrdiff_t)off;
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            int co = co0 + rl + p * 8, ci = ci0 + rl + p * 8;
            ra[p] = (mv && co < Cout) ? gy[gbase + (size_t)co * DHW] : 0.f;
            float v = (xv && ci < Cin) ? x[xb + (size_t)ci * DHW] : 0.f;
            rb[p] = relu_in ? fmaxf(v, 0.f) : v;
        }
    };

    if (q0 < q1) load_chunk(q0);
    for (int q = q0; q < q1; ++q) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            As[(rl + p * 8) * WG_PITCH + ml] = ra[p];
            Bs[(rl + p * 8) * WG_PITCH + ml] = rb[p];
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
    // slab[((split*ntaps + j)*Cout + co)*Cin + ci]; rows = co (registers), cols = ci (lanes)
    const int ci = ci0 + wci * 32 + l31;
    if (ci < Cin) {
        float* ps = slab + ((size_t)split * g.ntaps + j) * Cout * Cin + ci;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (co < Cout) ps[(size_t)co * Cin] = acc[r];
        }
    }
}

struct TapMap { int32_t j[T2V_MAX_TAPS]; };   // original tap t -> geometry tap j or -1

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           long CoCi, int T, int ntaps, int S, TapMap map, int accum) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= CoCi) return;
    for (int t = 0; t < T; ++t) {
        const int j = map.j[t];
        float v = 0.f;
        if (j >= 0)
            for (int s = 0; s < S; ++s) v += slab[((size_t)s * ntaps + j) * CoCi + i];
        float* p = dw + (size_t)i * T + t;
        *p = accum ? (*p + v) : v;
    }
}

static int wgrad_splits(const t2v_conv_geom* g) {
    const long M = (long)g->N * g->D * g->H * g->W;
    const long nchunks = (M + WG_BK - 1) / WG_BK;
    const long base = (long)((g->Cout + 63) / 64) * ((g->Cin + 63) / 64) * g->ntaps;
    long S = (1536 + base - 1) / base;            // aim at ~6 workgroups per CU
    long maxS = (nchunks + 7) / 8;                // at least 8 chunks (256 voxels) per split
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    if (S > 256) S = 256;
    return (int)S;
}

extern "C" int64_t t2v_conv_wgrad_slab_floats(const t2v_conv_geom* g) {
    if (!geom_ok(g)) return T2V_EINVAL;
    return (int64_t)wgrad_splits(g) * g->ntaps * g->Cout * g->Cin;
}

extern "C" int t2v_conv_wgrad(const float* x, const float* gy, float* dw, float* slab, const t2v_conv_geom* g,
                              const int32_t* taps, int T, int flags, void* stream) {
    if (!x || !gy || !dw || !slab || !geom_ok(g) || !taps || T < g->ntaps || T > T2V_MAX_TAPS) return T2V_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    TapMap map;
    for (int t = 0; t < T2V_MAX_TAPS; ++t) map.j[t] = -1;
    for (int j = 0; j < g->ntaps; ++j) {
        if (taps[j] < 0 || taps[j] >= T) return T2V_EINVAL;
        map.j[taps[j]] = j;
    }
    const long M = (long)g->N * g->D * g->H * g->W;
    const int S = wgrad_splits(g);
    const long nchunks = (M + WG_BK - 1) / WG_BK;
    const int cps = (int)((nchunks + S - 1) / S);
    dim3 grid((unsigned)(((g->Cout + 63) / 64) * ((g->Cin + 63) / 64)), (unsigned)g->ntaps, (unsigned)S);
    {
        ProfScope prof(1, 2.0 * (double)M * g->Cout * g->Cin * g->ntaps, s);
        T2V_LAUNCH(conv_wgrad_kernel, grid, dim3(256), 0, s, x, gy, slab, *g, flags, cps);
    }
    int st = launch_status();
    if (st) return st;
    const long CoCi = (long)g->Cout * g->Cin;
    ProfScope prof2(2, 0.0, s);
    T2V_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((CoCi + 255) / 256)), dim3(256), 0, s, slab, dw, CoCi, T,
                       g->ntaps, S, map, (flags & T2V_CONV_ACCUM) ? 1 : 0);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// per-channel sum over (N, S): bias gradient / BatchNorm reductions
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int N,
                                                          int C, long S, int accum) {
    const int c = blockIdx.x;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* p = x + ((size_t)n * C + c) * S;
        for (long i = threadIdx.x; i < S; i += 256) acc += p[i];
    }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = accum ? out[c] + red[0] : red[0];
}

extern "C" int t2v_channel_sum(const float* x, float* out, int N, int C, int64_t S, int accum, void* stream) {
    if (!x || !out || N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    T2V_LAUNCH(channel_sum_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, out, N, C, (long)S, accum);
    return launch_status();
}
