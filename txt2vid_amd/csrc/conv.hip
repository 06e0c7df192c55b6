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
//   tile BM (voxels) x BN (output channels), K consumed in chunks of BKT (one tap x BKT channels on the
//   FAST path), 4 waves, each owning (BN/WAVES_CO) x (BM/WAVES_M) as 32x32 MFMA tiles.
//   Global -> registers (issued one chunk ahead, right after the barrier) -> LDS -> MFMA.
//   Split-K (gridDim.z > 1): every split writes its partial tile into slab[z] and a second kernel sums
//   the splits in a fixed order (deterministic) and adds the bias: the layers with M of a few dozen
//   voxels and K of several thousand (the deep discriminator blocks, the ConvLSTM) are otherwise a
//   handful of workgroups each walking K serially at memory latency.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_CO, int BKT, bool FAST, bool VECB>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         float* __restrict__ slab, const t2v_conv_geom g, const int flags,
                                                         const int chunks_per_split) {
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

    __shared__ __attribute__((aligned(16))) float As[BKT * BM];
    __shared__ __attribute__((aligned(16))) float Bs[BKT * BN];
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
    const int q0 = blockIdx.z * chunks_per_split;
    int q1 = q0 + chunks_per_split;
    if (q1 > nchunks) q1 = nchunks;
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
            if (VECB) {
                const float* pw = wp + ((size_t)t * Cin + c0 + kv_l) * Cout + co0 + cv_l;
#pragma unroll
                for (int j = 0; j < LBV; ++j)
                    rbv[j] = co_ok ? *reinterpret_cast<const float4*>(pw + (size_t)j * KSBV * Cout) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const float* pw = wp + ((size_t)t * Cin + c0 + kb_l) * Cout + co0 + cob_l;
#pragma unroll
                for (int j = 0; j < LB; ++j) rb[j] = co_ok ? pw[(size_t)j * KSB * Cout] : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                int kk = q * BKT + ka_l + j * KSA;
                float val = 0.f;
                if (kk < Ktot) {
                    int tt = kk / Cin, ci = kk - tt * Cin;
                    if ((tapmask >> tt) & 1u) val = x[xbase + (ptrdiff_t)s_off[tt] + (size_t)ci * DHW];
                }
                ra[j] = relu_in ? fmaxf(val, 0.f) : val;
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                int kk = q * BKT + kb_l + j * KSB;
                rb[j] = (co_ok && kk < Ktot) ? wp[(size_t)kk * Cout + co0 + cob_l] : 0.f;
            }
        }
    };

    int t_cur = 0, c_cur = 0;
    if (FAST) {
        const int cpt = Cin / BKT;
        t_cur = q0 / cpt;
        c_cur = (q0 - t_cur * cpt) * BKT;
    }
    if (q0 < q1) load_chunk(q0, t_cur, c_cur);
    for (int q = q0; q < q1; ++q) {
        // registers -> LDS
#pragma unroll
        for (int j = 0; j < LA; ++j) As[(ka_l + j * KSA) * BM + ma_l] = ra[j];
        if (VECB) {
#pragma unroll
            for (int j = 0; j < LBV; ++j)
                if (vact) *reinterpret_cast<float4*>(&Bs[(kv_l + j * KSBV) * BN + cv_l]) = rbv[j];
        } else {
#pragma unroll
            for (int j = 0; j < LB; ++j) Bs[(kb_l + j * KSB) * BN + cob_l] = rb[j];
        }
        __syncthreads();
        // issue the next chunk's global loads before the MFMAs (latency hides under compute)
        if (q + 1 < q1) {
            c_cur += BKT;
            if (FAST && c_cur >= Cin) { c_cur = 0; ++t_cur; }
            load_chunk(q + 1, t_cur, c_cur);
        }
#pragma unroll 8
        for (int k2 = 0; k2 < BKT / 2; ++k2) {
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
    const bool split = gridDim.z > 1;
    const bool has_bias = !split && (flags & T2V_CONV_BIAS) && bias != nullptr;
    const bool accum = !split && (flags & T2V_CONV_ACCUM);
    float* out = split ? slab + (size_t)blockIdx.z * ((size_t)M * Cout) : y;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        const int m = m0 + wm * WM + j * 32 + l31;
        if (m >= M) continue;
        const int n = m / DHW, sp = m - n * DHW;
        float* py = out + (size_t)n * Cout * DHW + sp;
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

// y = (accum ? y : 0) + bias[co] + sum_s slab[s]   (fixed summation order)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                            float* __restrict__ y, long total, int S, int Cout, int DHW, int flags) {
    const bool has_bias = (flags & T2V_CONV_BIAS) && bias != nullptr;
    const bool accum = flags & T2V_CONV_ACCUM;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += slab[(size_t)s * total + i];
        if (has_bias) v += bias[(i / DHW) % Cout];
        y[i] = accum ? y[i] + v : v;
    }
}

struct ConvPlan { int bm, bn, bk; bool fast, vecb; int S, cps; long tiles; };

static ConvPlan conv_plan(const t2v_conv_geom& g) {
    ConvPlan p;
    const long M = (long)g.N * g.D * g.H * g.W;
    p.bk = (g.Cin % 64 == 0) ? 64 : (g.Cin % 32 == 0) ? 32 : 16;
    p.fast = (g.Cin % 16 == 0);
    p.vecb = (g.Cout % 4 == 0);
    if (g.Cout <= 32) { p.bm = 128; p.bn = 32; if (p.bk > 32) p.bk = 32; }
    else {
        p.bn = 64;
        const long t128 = ((M + 127) / 128) * ((g.Cout + 63) / 64);
        p.bm = (t128 >= 768) ? 128 : 64;
        if (p.bm == 128 && p.bk > 32) p.bk = 32;
    }
    p.tiles = ((M + p.bm - 1) / p.bm) * ((g.Cout + p.bn - 1) / p.bn);
    const long K = (long)g.ntaps * g.Cin;
    const long nchunks = p.fast ? (long)g.ntaps * (g.Cin / p.bk) : (K + p.bk - 1) / p.bk;
    long S = 1;
    if (p.tiles < 384) {
        S = (768 + p.tiles - 1) / p.tiles;
        long maxS = nchunks / 2;                       // >= 2 chunks per split
        if (S > maxS) S = maxS;
        if (S > 64) S = 64;
        while (S > 1 && (double)S * M * g.Cout * 4.0 > 256e6) --S;   // keep the slab small (L2/MALL resident)
        if (S < 1) S = 1;
    }
    p.cps = (int)((nchunks + S - 1) / S);
    p.S = (int)((nchunks + p.cps - 1) / p.cps);
    return p;
}

template <int BM, int BN, int WAVES_CO, int BKT>
static void launch_conv_t(const float* x, const float* wp, const float* bias, float* y, float* slab, const t2v_conv_geom& g,
                          int flags, const ConvPlan& p, hipStream_t s) {
    const long M = (long)g.N * g.D * g.H * g.W;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((g.Cout + BN - 1) / BN), (unsigned)p.S);
    if (p.fast) {
        if (p.vecb) T2V_LAUNCH((conv_igemm_kernel<BM, BN, WAVES_CO, BKT, true, true>), grid, dim3(256), 0, s, x, wp, bias, y, slab, g, flags, p.cps);
        else T2V_LAUNCH((conv_igemm_kernel<BM, BN, WAVES_CO, BKT, true, false>), grid, dim3(256), 0, s, x, wp, bias, y, slab, g, flags, p.cps);
    } else {
        T2V_LAUNCH((conv_igemm_kernel<BM, BN, WAVES_CO, 16, false, false>), grid, dim3(256), 0, s, x, wp, bias, y, slab, g, flags, p.cps);
    }
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

extern "C" int64_t t2v_conv_fwd_ws_floats(const t2v_conv_geom* g) {
    if (!geom_ok(g)) return T2V_EINVAL;
    ConvPlan p = conv_plan(*g);
    if (p.S <= 1) return 0;
    return (int64_t)p.S * g->N * g->D * g->H * g->W * g->Cout;
}

extern "C" int t2v_conv_fwd(const float* x, const float* wp, const float* bias, float* y, float* ws, const t2v_conv_geom* g,
                            int flags, void* stream) {
    if (!x || !wp || !y || !geom_ok(g)) return T2V_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const ConvPlan p = conv_plan(*g);
    if (p.S > 1 && !ws) return T2V_EINVAL;
    const long M = (long)g->N * g->D * g->H * g->W;
    {
        ProfScope prof(0, 2.0 * (double)M * g->Cout * g->Cin * g->ntaps, s);      // executed (non-padding-tap) MACs x 2
        const int bk = p.fast ? p.bk : 16;
        if (p.bn == 32) {
            if (bk == 32) launch_conv_t<128, 32, 1, 32>(x, wp, bias, y, ws, *g, flags, p, s);
            else launch_conv_t<128, 32, 1, 16>(x, wp, bias, y, ws, *g, flags, p, s);
        } else if (p.bm == 128) {
            if (bk == 32) launch_conv_t<128, 64, 2, 32>(x, wp, bias, y, ws, *g, flags, p, s);
            else launch_conv_t<128, 64, 2, 16>(x, wp, bias, y, ws, *g, flags, p, s);
        } else {
            if (bk == 64) launch_conv_t<64, 64, 2, 64>(x, wp, bias, y, ws, *g, flags, p, s);
            else if (bk == 32) launch_conv_t<64, 64, 2, 32>(x, wp, bias, y, ws, *g, flags, p, s);
            else launch_conv_t<64, 64, 2, 16>(x, wp, bias, y, ws, *g, flags, p, s);
        }
        int st = launch_status();
        if (st) return st;
        if (p.S > 1) {
            const long total = M * g->Cout;
            long blocks = (total + 255) / 256;
            if (blocks > 2048) blocks = 2048;
            T2V_LAUNCH(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ws, bias, y, total, p.S, g->Cout,
                       g->D * g->H * g->W, flags);
        }
    }
    return launch_status();
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

// dw[co][ci][t] = sum_s slab[s][j(t)][co][ci] (0 for taps that only ever multiply padding). Reads are
// lane-contiguous along (co,ci); the [i][t] transposition goes through LDS so that the PyTorch-layout
// gradient is written as one contiguous run per workgroup.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                           long CoCi, int T, int ntaps, int S, TapMap map, int accum) {
    // workgroup = 64 (co,ci) pairs x T taps; the 4 waves take taps t = wave, wave+4, ...
    __shared__ float tile[64 * T2V_MAX_TAPS];
    const long i0 = (long)blockIdx.x * 64;
    const int il = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const long i = i0 + il;
    for (int t = tg; t < T; t += 4) {
        const int j = map.j[t];
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (j >= 0 && i < CoCi) {
            const float* p = slab + (size_t)j * CoCi + i;
            const size_t st = (size_t)ntaps * CoCi;
            int s = 0;
            for (; s + 4 <= S; s += 4) {
                v0 += p[(size_t)s * st]; v1 += p[(size_t)(s + 1) * st]; v2 += p[(size_t)(s + 2) * st]; v3 += p[(size_t)(s + 3) * st];
            }
            for (; s < S; ++s) v0 += p[(size_t)s * st];
        }
        tile[il * T + t] = (v0 + v1) + (v2 + v3);
    }
    __syncthreads();
    long cnt = CoCi - i0;
    if (cnt > 64) cnt = 64;
    const long nval = cnt * T;
    float* p = dw + (size_t)i0 * T;
    for (long k = threadIdx.x; k < nval; k += 256) p[k] = accum ? p[k] + tile[k] : tile[k];
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
    T2V_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((CoCi + 63) / 64)), dim3(256), 0, s, slab, dw, CoCi, T,
                       g->ntaps, S, map, (flags & T2V_CONV_ACCUM) ? 1 : 0);
    return launch_status();
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
