// pointwise.hip — HBM-bound element-wise, pooling, normalisation, reduction, loss and optimiser
// kernels of the TGANv2 hot path (gfx950). Lane-contiguous 4-/16-byte accesses, grid-stride loops
// capped at 2048 workgroups (256 CUs x 8), wave64 shuffles for the row reductions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/t2v_hip.h"

// hipGetLastError() is sticky per thread and also reports errors of calls the HOST framework made and
// handled earlier; clear it before every launch so that launch_status() reflects this launch only.
#define T2V_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

static inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? T2V_OK : -(int)e - 1000;
}
static inline unsigned nblocks(long n, int per = 256) {
    long b = (n + per - 1) / per;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
#define S_(stream) ((hipStream_t)(stream))

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide sum for 256 threads; every thread gets the result
__device__ __forceinline__ float block_sum(float v, float* red /*4 floats*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------- element-wise
__global__ void relu_k(const float* x, float* y, long n) { GRID_STRIDE(i, n) y[i] = fmaxf(x[i], 0.f); }
__global__ void relu_mask_k(const float* g, const float* x, float* gx, long n) { GRID_STRIDE(i, n) gx[i] = x[i] > 0.f ? g[i] : 0.f; }
__global__ void add_k(const float* a, const float* b, float* y, long n) { GRID_STRIDE(i, n) y[i] = a[i] + b[i]; }
__global__ void axpby_k(float al, const float* a, float be, const float* b, float* y, long n) {
    GRID_STRIDE(i, n) y[i] = al * a[i] + (b ? be * b[i] : 0.f);
}
__global__ void scale_dev_k(const float* s, float mul, const float* a, float* y, long n) {
    const float f = s[0] * mul;
    GRID_STRIDE(i, n) y[i] = f * a[i];
}
__global__ void fill_k(float* y, float v, long n) { GRID_STRIDE(i, n) y[i] = v; }
__global__ void tanh_k(const float* x, float* y, long n) { GRID_STRIDE(i, n) y[i] = tanhf(x[i]); }
__global__ void tanh_bwd_k(const float* g, const float* y, float* gx, long n) { GRID_STRIDE(i, n) gx[i] = g[i] * (1.f - y[i] * y[i]); }

extern "C" int t2v_relu(const float* x, float* y, int64_t n, void* st) {
    if (n <= 0) return n == 0 ? T2V_OK : T2V_EINVAL;
    T2V_LAUNCH(relu_k, dim3(nblocks(n)), dim3(256), 0, S_(st), x, y, (long)n);
    return launch_status();
}
extern "C" int t2v_relu_mask(const float* g, const float* x, float* gx, int64_t n, void* st) {
    if (n <= 0) return n == 0 ? T2V_OK : T2V_EINVAL;
    T2V_LAUNCH(relu_mask_k, dim3(nblocks(n)), dim3(256), 0, S_(st), g, x, gx, (long)n);
    return launch_status();
}
extern "C" int t2v_add(const float* a, const float* b, float* y, int64_t n, void* st) {
    if (n <= 0) return n == 0 ? T2V_OK : T2V_EINVAL;
    T2V_LAUNCH(add_k, dim3(nblocks(n)), dim3(256), 0, S_(st), a, b, y, (long)n);
    return launch_status();
}
extern "C" int t2v_axpby(float al, const float* a, float be, const float* b, float* y, int64_t n, void* st) {
    if (n <= 0) return n == 0 ? T2V_OK : T2V_EINVAL;
    T2V_LAUNCH(axpby_k, dim3(nblocks(n)), dim3(256), 0, S_(st), al, a, be, b, y, (long)n);
    return launch_status();
}
extern "C" int t2v_scale_dev(const float* s, float mul, const float* a, float* y, int64_t n, void* st) {
    if (n <= 0) return n == 0 ? T2V_OK : T2V_EINVAL;
    T2V_LAUNCH(scale_dev_k, dim3(nblocks(n)), dim3(256), 0, S_(st), s, mul, a, y, (long)n);
    return launch_status();
}
extern "C" int t2v_fill(float* y, float v, int64_t n, void* st) {
    if (n <= 0) return n == 0 ? T2V_OK : T2V_EINVAL;
    T2V_LAUNCH(fill_k, dim3(nblocks(n)), dim3(256), 0, S_(st), y, v, (long)n);
    return launch_status();
}
extern "C" int t2v_tanh(const float* x, float* y, int64_t n, void* st) {
    if (n <= 0) return n == 0 ? T2V_OK : T2V_EINVAL;
    T2V_LAUNCH(tanh_k, dim3(nblocks(n)), dim3(256), 0, S_(st), x, y, (long)n);
    return launch_status();
}
extern "C" int t2v_tanh_bwd(const float* g, const float* y, float* gx, int64_t n, void* st) {
    if (n <= 0) return n == 0 ? T2V_OK : T2V_EINVAL;
    T2V_LAUNCH(tanh_bwd_k, dim3(nblocks(n)), dim3(256), 0, S_(st), g, y, gx, (long)n);
    return launch_status();
}

// dot product, two deterministic stages: up to 256 workgroups leave partial sums in ws, one wave-pass sums them.
__global__ __launch_bounds__(256) void dot_partial_k(const float* a, const float* b, float* ws, long n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += a[i] * b[i];
    float s = block_sum(acc, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void dot_final_k(const float* ws, float* out, int nb, int accum) {
    __shared__ float red[4];
    float acc = threadIdx.x < nb ? ws[threadIdx.x] : 0.f;
    float s = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = accum ? out[0] + s : s;
}
extern "C" int t2v_dot(const float* a, const float* b, float* out, float* ws, int64_t n, int accum, void* st) {
    if (n <= 0 || !a || !b || !out || !ws) return T2V_EINVAL;
    long nb = (n + 4095) / 4096;
    if (nb > 256) nb = 256;
    T2V_LAUNCH(dot_partial_k, dim3((unsigned)nb), dim3(256), 0, S_(st), a, b, ws, (long)n);
    T2V_LAUNCH(dot_final_k, dim3(1), dim3(256), 0, S_(st), ws, out, (int)nb, accum);
    return launch_status();
}

// ---------------------------------------------------------------- avg-pool (count_include_pad)
struct Pool3 { int k[3], s[3], p[3]; };

// x2 != nullptr: y = pool(x + x2) (the residual add of DownBlock fused into its DownSample: pooling is linear)
__global__ void avgpool3d_k(const float* x, const float* x2, float* y, int NC, int D, int H, int W, int Do, int Ho, int Wo, Pool3 q) {
    const long n = (long)NC * Do * Ho * Wo;
    const float inv = 1.f / (float)(q.k[0] * q.k[1] * q.k[2]);
    GRID_STRIDE(i, n) {
        int wo = i % Wo; long r = i / Wo;
        int ho = r % Ho; r /= Ho;
        int d_o = r % Do; long nc = r / Do;
        const float* px = x + nc * (long)D * H * W;
        const float* px2 = x2 ? x2 + nc * (long)D * H * W : nullptr;
        float acc = 0.f;
        for (int a = 0; a < q.k[0]; ++a) {
            int d = d_o * q.s[0] - q.p[0] + a;
            if ((unsigned)d >= (unsigned)D) continue;
            for (int b = 0; b < q.k[1]; ++b) {
                int h = ho * q.s[1] - q.p[1] + b;
                if ((unsigned)h >= (unsigned)H) continue;
                for (int c = 0; c < q.k[2]; ++c) {
                    int w = wo * q.s[2] - q.p[2] + c;
                    if ((unsigned)w >= (unsigned)W) continue;
                    const long o = ((long)d * H + h) * W + w;
                    acc += px2 ? px[o] + px2[o] : px[o];
                }
            }
        }
        y[i] = acc * inv;
    }
}
// adjoint (gather form): gx[d,h,w] = sum over windows containing it of gy/vol. stride >= kernel on this
// path, but the general overlap case is handled as well.
__global__ void avgpool3d_bwd_k(const float* gy, float* gx, int NC, int D, int H, int W, int Do, int Ho, int Wo, Pool3 q) {
    const long n = (long)NC * D * H * W;
    const float inv = 1.f / (float)(q.k[0] * q.k[1] * q.k[2]);
    GRID_STRIDE(i, n) {
        int w = i % W; long r = i / W;
        int h = r % H; r /= H;
        int d = r % D; long nc = r / D;
        const float* pg = gy + nc * (long)Do * Ho * Wo;
        float acc = 0.f;
        for (int a = 0; a < q.k[0]; ++a) {
            int td = d + q.p[0] - a;
            if (td < 0 || td % q.s[0]) continue;
            int d_o = td / q.s[0];
            if (d_o >= Do) continue;
            for (int b = 0; b < q.k[1]; ++b) {
                int th = h + q.p[1] - b;
                if (th < 0 || th % q.s[1]) continue;
                int ho = th / q.s[1];
                if (ho >= Ho) continue;
                for (int c = 0; c < q.k[2]; ++c) {
                    int tw = w + q.p[2] - c;
                    if (tw < 0 || tw % q.s[2]) continue;
                    int wo = tw / q.s[2];
                    if (wo >= Wo) continue;
                    acc += pg[((long)d_o * Ho + ho) * Wo + wo];
                }
            }
        }
        gx[i] = acc * inv;
    }
}
// non-overlapping, unpadded windows (kernel <= stride, the only kind on the hot path): every input voxel belongs
// to at most one window -> one read, no divisibility tests
__global__ void avgpool3d_bwd_simple_k(const float* gy, float* gx, int NC, int D, int H, int W, int Do, int Ho, int Wo, Pool3 q) {
    const long n = (long)NC * D * H * W;
    const float inv = 1.f / (float)(q.k[0] * q.k[1] * q.k[2]);
    GRID_STRIDE(i, n) {
        int w = i % W; long r = i / W;
        int h = r % H; r /= H;
        int d = r % D; long nc = r / D;
        const int d_o = d / q.s[0], ho = h / q.s[1], wo = w / q.s[2];
        const bool in = (d - d_o * q.s[0] < q.k[0]) && (h - ho * q.s[1] < q.k[1]) && (w - wo * q.s[2] < q.k[2]) &&
                        d_o < Do && ho < Ho && wo < Wo;
        gx[i] = in ? gy[((nc * Do + d_o) * Ho + ho) * (long)Wo + wo] * inv : 0.f;
    }
}
// Multi-tensor forms (the pyramid levels of one DownBlock pooled by ONE launch): up to 8 jobs by value in the kernel
// arguments; a workgroup handles 1024 consecutive output elements of one job.
#define POOL_MT 8
#ifndef POOL_CHUNK
#define POOL_CHUNK 512           // (512 vs 1024: -0.12 ms per iteration: the small members of a launch get more workgroups in flight; 256: the same)
#endif
struct PoolBatch { t2v_pool_job j[POOL_MT]; int begin[POOL_MT + 1]; int n; };
__global__ __launch_bounds__(256) void avgpool3d_multi_k(const PoolBatch tb) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < POOL_MT; ++k)
        if (k < tb.n && (int)blockIdx.x >= tb.begin[k]) ji = k;
    const t2v_pool_job& q = tb.j[ji];
    const int D = q.D, H = q.H, W = q.W, Do = q.Do, Ho = q.Ho, Wo = q.Wo;
    const long n = (long)q.NC * Do * Ho * Wo;
    const float inv = 1.f / (float)(q.k[0] * q.k[1] * q.k[2]);
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * POOL_CHUNK;
    const float* __restrict__ x = q.x;
    const float* __restrict__ x2 = q.x2;
    for (long i = base + threadIdx.x; i < base + POOL_CHUNK && i < n; i += 256) {
        int wo = i % Wo; long r = i / Wo;
        int ho = r % Ho; r /= Ho;
        int d_o = r % Do; long nc = r / Do;
        const long off = nc * (long)D * H * W;
        float acc = 0.f;
        for (int a = 0; a < q.k[0]; ++a) {
            int d = d_o * q.s[0] - q.p[0] + a;
            if ((unsigned)d >= (unsigned)D) continue;
            for (int b = 0; b < q.k[1]; ++b) {
                int h = ho * q.s[1] - q.p[1] + b;
                if ((unsigned)h >= (unsigned)H) continue;
                for (int c = 0; c < q.k[2]; ++c) {
                    int w = wo * q.s[2] - q.p[2] + c;
                    if ((unsigned)w >= (unsigned)W) continue;
                    const long o = off + ((long)d * H + h) * W + w;
                    acc += x2 ? x[o] + x2[o] : x[o];
                }
            }
        }
        q.y[i] = acc * inv + (q.add ? q.add[i] : 0.f);
    }
}
// adjoint for non-overlapping unpadded windows (what DownSample produces): job.x = dL/dy [NC,Do,Ho,Wo], job.y = dL/dx
__global__ __launch_bounds__(256) void avgpool3d_bwd_multi_k(const PoolBatch tb) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < POOL_MT; ++k)
        if (k < tb.n && (int)blockIdx.x >= tb.begin[k]) ji = k;
    const t2v_pool_job& q = tb.j[ji];
    const int D = q.D, H = q.H, W = q.W, Do = q.Do, Ho = q.Ho, Wo = q.Wo;
    const long n = (long)q.NC * D * H * W;
    const float inv = 1.f / (float)(q.k[0] * q.k[1] * q.k[2]);
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * POOL_CHUNK;
    const float* __restrict__ gy = q.x;
    for (long i = base + threadIdx.x; i < base + POOL_CHUNK && i < n; i += 256) {
        int w = i % W; long r = i / W;
        int h = r % H; r /= H;
        int d = r % D; long nc = r / D;
        const int d_o = d / q.s[0], ho = h / q.s[1], wo = w / q.s[2];
        const bool in = (d - d_o * q.s[0] < q.k[0]) && (h - ho * q.s[1] < q.k[1]) && (w - wo * q.s[2] < q.k[2]) &&
                        d_o < Do && ho < Ho && wo < Wo;
        const float v = in ? gy[((nc * Do + d_o) * Ho + ho) * (long)Wo + wo] * inv : 0.f;
        q.y[i] = q.add ? v + q.add[i] : v;          // (add: dL/dx's other contribution, same shape as dL/dx — a forked block input)
    }
}
static int pool_multi(const t2v_pool_job* jobs, int njobs, bool bwd, void* st) {
    if (!jobs || njobs < 1) return T2V_EINVAL;
    for (int at = 0; at < njobs; at += POOL_MT) {
        PoolBatch tb;
        const int cnt = njobs - at < POOL_MT ? njobs - at : POOL_MT;
        long blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const t2v_pool_job& q = jobs[at + i];
            if (!q.x || !q.y || q.NC < 1 || q.D < 1 || q.H < 1 || q.W < 1 || q.Do < 1 || q.Ho < 1 || q.Wo < 1) return T2V_EINVAL;
            for (int a = 0; a < 3; ++a) {
                if (q.k[a] < 1 || q.k[a] > 4 || q.s[a] < 1 || q.p[a] < 0) return T2V_EINVAL;
                if (bwd && (q.p[a] != 0 || q.k[a] > q.s[a])) return T2V_EINVAL;      // the simple adjoint only
            }
            tb.j[i] = q;
            tb.begin[i] = (int)blocks;
            const long n = bwd ? (long)q.NC * q.D * q.H * q.W : (long)q.NC * q.Do * q.Ho * q.Wo;
            blocks += (n + POOL_CHUNK - 1) / POOL_CHUNK;
        }
        for (int i = cnt; i <= POOL_MT; ++i) tb.begin[i] = (int)blocks;
        for (int i = cnt; i < POOL_MT; ++i) tb.j[i] = tb.j[0];
        tb.n = cnt;
        if (blocks > 0x7fffffffL) return T2V_EINVAL;
        if (bwd) T2V_LAUNCH(avgpool3d_bwd_multi_k, dim3((unsigned)blocks), dim3(256), 0, S_(st), tb);
        else T2V_LAUNCH(avgpool3d_multi_k, dim3((unsigned)blocks), dim3(256), 0, S_(st), tb);
    }
    return launch_status();
}
extern "C" int t2v_avgpool3d_multi(const t2v_pool_job* jobs, int njobs, void* st) { return pool_multi(jobs, njobs, false, st); }
extern "C" int t2v_avgpool3d_bwd_multi(const t2v_pool_job* jobs, int njobs, void* st) { return pool_multi(jobs, njobs, true, st); }

static bool pool_ok(const int32_t* k, const int32_t* s, const int32_t* p) {
    for (int i = 0; i < 3; ++i)
        if (k[i] < 1 || k[i] > 4 || s[i] < 1 || p[i] < 0) return false;
    return true;
}
extern "C" int t2v_avgpool3d(const float* x, const float* x2, float* y, int NC, int D, int H, int W, int Do, int Ho, int Wo,
                             const int32_t k[3], const int32_t s[3], const int32_t p[3], void* st) {
    if (!x || !y || NC < 1 || !pool_ok(k, s, p)) return T2V_EINVAL;
    Pool3 q;
    for (int i = 0; i < 3; ++i) { q.k[i] = k[i]; q.s[i] = s[i]; q.p[i] = p[i]; }
    T2V_LAUNCH(avgpool3d_k, dim3(nblocks((long)NC * Do * Ho * Wo)), dim3(256), 0, S_(st), x, x2, y, NC, D, H, W, Do, Ho, Wo, q);
    return launch_status();
}
extern "C" int t2v_avgpool3d_bwd(const float* gy, float* gx, int NC, int D, int H, int W, int Do, int Ho, int Wo,
                                 const int32_t k[3], const int32_t s[3], const int32_t p[3], void* st) {
    if (!gy || !gx || NC < 1 || !pool_ok(k, s, p)) return T2V_EINVAL;
    Pool3 q;
    for (int i = 0; i < 3; ++i) { q.k[i] = k[i]; q.s[i] = s[i]; q.p[i] = p[i]; }
    const bool simple = p[0] == 0 && p[1] == 0 && p[2] == 0 && k[0] <= s[0] && k[1] <= s[1] && k[2] <= s[2];
    if (simple) T2V_LAUNCH(avgpool3d_bwd_simple_k, dim3(nblocks((long)NC * D * H * W)), dim3(256), 0, S_(st), gy, gx, NC, D, H, W, Do, Ho, Wo, q);
    else T2V_LAUNCH(avgpool3d_bwd_k, dim3(nblocks((long)NC * D * H * W)), dim3(256), 0, S_(st), gy, gx, NC, D, H, W, Do, Ho, Wo, q);
    return launch_status();
}

// ---------------------------------------------------------------- 2x2 max-pool over (H,W) planes (floor)
__global__ void maxpool2x2_k(const float* x, float* y, int32_t* idx, long planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long n = planes * Ho * Wo;
    GRID_STRIDE(i, n) {
        int wo = i % Wo; long r = i / Wo;
        int ho = r % Ho; long pl = r / Ho;
        const float* px = x + pl * (long)H * W;
        int best = (2 * ho) * W + 2 * wo;
        float bv = px[best];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                int id = (2 * ho + a) * W + 2 * wo + b;
                float v = px[id];
                if (v > bv || (v != v && bv == bv)) { bv = v; best = id; }   // first max wins; NaN propagates
            }
        y[i] = bv;
        idx[i] = best;
    }
}
__global__ void maxpool2x2_scatter_k(const float* g, const int32_t* idx, float* gx, long planes, int H, int W) {
    // every input element belongs to at most one window -> gather form, no atomics
    const int Ho = H / 2, Wo = W / 2;
    const long n = planes * H * W;
    GRID_STRIDE(i, n) {
        int w = i % W; long r = i / W;
        int h = r % H; long pl = r / H;
        int ho = h >> 1, wo = w >> 1;
        float v = 0.f;
        if (ho < Ho && wo < Wo) {
            long o = (pl * Ho + ho) * Wo + wo;
            if (idx[o] == h * W + w) v = g[o];
        }
        gx[i] = v;
    }
}
__global__ void maxpool2x2_gather_k(const float* x, const int32_t* idx, float* y, long planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long n = planes * Ho * Wo;
    GRID_STRIDE(i, n) {
        long pl = i / ((long)Ho * Wo);
        y[i] = x[pl * (long)H * W + idx[i]];
    }
}
extern "C" int t2v_maxpool2x2(const float* x, float* y, int32_t* idx, int64_t planes, int H, int W, void* st) {
    if (!x || !y || !idx || planes < 1 || H < 2 || W < 2) return T2V_EINVAL;
    T2V_LAUNCH(maxpool2x2_k, dim3(nblocks(planes * (H / 2) * (W / 2))), dim3(256), 0, S_(st), x, y, idx, (long)planes, H, W);
    return launch_status();
}
extern "C" int t2v_maxpool2x2_scatter(const float* g, const int32_t* idx, float* gx, int64_t planes, int H, int W, void* st) {
    if (!g || !gx || !idx || planes < 1 || H < 2 || W < 2) return T2V_EINVAL;
    T2V_LAUNCH(maxpool2x2_scatter_k, dim3(nblocks(planes * H * W)), dim3(256), 0, S_(st), g, idx, gx, (long)planes, H, W);
    return launch_status();
}
extern "C" int t2v_maxpool2x2_gather(const float* x, const int32_t* idx, float* y, int64_t planes, int H, int W, void* st) {
    if (!x || !y || !idx || planes < 1 || H < 2 || W < 2) return T2V_EINVAL;
    T2V_LAUNCH(maxpool2x2_gather_k, dim3(nblocks(planes * (H / 2) * (W / 2))), dim3(256), 0, S_(st), x, idx, y, (long)planes, H, W);
    return launch_status();
}

// ---------------------------------------------------------------- row reductions (one wave per row)
__global__ __launch_bounds__(256) void rowsum_k(const float* x, float* y, long rows, long S) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + row * S;
    float acc = 0.f;
    for (long i = threadIdx.x & 63; i < S; i += 64) acc += p[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) y[row] = acc;
}
__global__ void rowbcast_k(const float* g, float* gx, long rows, long S) {
    const long n = rows * S;
    GRID_STRIDE(i, n) gx[i] = g[i / S];
}
extern "C" int t2v_rowsum(const float* x, float* y, int64_t rows, int64_t S, void* st) {
    if (!x || !y || rows < 1 || S < 1) return T2V_EINVAL;
    T2V_LAUNCH(rowsum_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_(st), x, y, (long)rows, (long)S);
    return launch_status();
}
extern "C" int t2v_rowbcast(const float* g, float* gx, int64_t rows, int64_t S, void* st) {
    if (!g || !gx || rows < 1 || S < 1) return T2V_EINVAL;
    T2V_LAUNCH(rowbcast_k, dim3(nblocks(rows * S)), dim3(256), 0, S_(st), g, gx, (long)rows, (long)S);
    return launch_status();
}

// ---------------------------------------------------------------- nearest x2 up-sampling
__global__ void upsample2x_k(const float* x, const float* add, float* y, long planes, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const long n = planes * Ho * Wo;
    GRID_STRIDE(i, n) {
        int wo = i % Wo; long r = i / Wo;
        int ho = r % Ho; long pl = r / Ho;
        const float v = x[(pl * H + (ho >> 1)) * W + (wo >> 1)];
        y[i] = add ? v + add[i] : v;
    }
}
__global__ void upsample2x_bwd_k(const float* gy, float* gx, long planes, int H, int W) {
    const int Wo = 2 * W;
    const long n = planes * H * W;
    GRID_STRIDE(i, n) {
        int w = i % W; long r = i / W;
        int h = r % H; long pl = r / H;
        const float* p = gy + ((pl * 2 * H + 2 * h) * Wo + 2 * w);
        gx[i] = (p[0] + p[1]) + (p[Wo] + p[Wo + 1]);
    }
}
extern "C" int t2v_upsample2x(const float* x, float* y, int64_t planes, int H, int W, void* st) {
    if (!x || !y || planes < 1 || H < 1 || W < 1) return T2V_EINVAL;
    T2V_LAUNCH(upsample2x_k, dim3(nblocks(planes * 4 * H * W)), dim3(256), 0, S_(st), x, (const float*)nullptr, y, (long)planes, H, W);
    return launch_status();
}
// y = upsample2x(x) + h: the residual add of an UpBlock whose identity path is Upsample [-> 1x1 conv] (layers.py:152-195; the
// 1x1 convolution commutes with nearest up-sampling, so it runs on the small map and this kernel closes the block)
extern "C" int t2v_upsample2x_add(const float* x, const float* h, float* y, int64_t planes, int H, int W, void* st) {
    if (!x || !h || !y || planes < 1 || H < 1 || W < 1) return T2V_EINVAL;
    T2V_LAUNCH(upsample2x_k, dim3(nblocks(planes * 4 * H * W)), dim3(256), 0, S_(st), x, h, y, (long)planes, H, W);
    return launch_status();
}
extern "C" int t2v_upsample2x_bwd(const float* gy, float* gx, int64_t planes, int H, int W, void* st) {
    if (!gy || !gx || planes < 1 || H < 1 || W < 1) return T2V_EINVAL;
    T2V_LAUNCH(upsample2x_bwd_k, dim3(nblocks(planes * H * W)), dim3(256), 0, S_(st), gy, gx, (long)planes, H, W);
    return launch_status();
}

// ---------------------------------------------------------------- BatchNorm2d (training)
// Statistics in two deterministic stages: grid (C, SPLIT) workgroups compute (count, mean, M2) of a
// contiguous slice of their channel (two passes over the slice, which stays in L2), the finalize kernel
// merges the slices with Chan's parallel-variance formula and updates the running statistics.
#define BN_MAXSPLIT 64
__device__ __forceinline__ float bn_elem(const float* x, long e, int C, int c, long S) {
    const long n = e / S, sp = e - n * S;
    return x[((size_t)n * C + c) * S + sp];
}
__global__ __launch_bounds__(256) void bn_stats_part_k(const float* x, float* part, int N, int C, long S, int split) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const long total = (long)N * S;
    const long per = (total + split - 1) / split;
    const long e0 = (long)blockIdx.y * per;
    long e1 = e0 + per;
    if (e1 > total) e1 = total;
    const long cnt = e1 > e0 ? e1 - e0 : 0;
    float acc = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) acc += bn_elem(x, e, C, c, S);
    const float mean = cnt > 0 ? block_sum(acc, red) / (float)cnt : 0.f;
    acc = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) { const float d = bn_elem(x, e, C, c, S) - mean; acc += d * d; }
    const float m2 = block_sum(acc, red);
    if (threadIdx.x == 0) {
        float* p = part + ((size_t)c * split + blockIdx.y) * 3;
        p[0] = (float)cnt; p[1] = mean; p[2] = m2;
    }
}
__global__ void bn_stats_final_k(const float* part, float* stats, float* rmean, float* rvar, int C, int split, float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int k = 0; k < split; ++k) {
        const float* p = part + ((size_t)c * split + k) * 3;
        const float nb = p[0];
        if (nb <= 0.f) continue;
        const float delta = p[1] - mean, nt = n + nb;
        mean += delta * (nb / nt);
        m2 += p[2] + delta * delta * (n * nb / nt);
        n = nt;
    }
    const float var = m2 / n;
    stats[c] = mean;
    stats[C + c] = 1.f / sqrtf(var + eps);
    if (rmean) {
        const float unb = n > 1.f ? m2 / (n - 1.f) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
}
#ifndef BN_STATS_WGS
#define BN_STATS_WGS 1024      // workgroups the statistics passes aim at
#define BN_APPLY_WGS 2048      // ... and the apply passes
#define BN_MIN_ELEMS 1024      // fewest elements of a channel one workgroup takes
#endif
static int bn_split(int N, int C, long S) {
    const long total = (long)N * S;
    long sp = (BN_STATS_WGS + C - 1) / C;
    if (sp > total / BN_MIN_ELEMS) sp = total / BN_MIN_ELEMS;
    if (sp > BN_MAXSPLIT) sp = BN_MAXSPLIT;
    if (sp < 1) sp = 1;
    return (int)sp;
}
__global__ void bn_apply_k(const float* x, const float* stats, const float* gamma, const float* beta, float* y, int N, int C,
                           long S, int relu) {
    const long n = (long)N * C * S;
    GRID_STRIDE(i, n) {
        const int c = (i / S) % C;
        float v = (x[i] - stats[c]) * stats[C + c] * gamma[c] + beta[c];
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
// pass 1: per channel sums  s1 = sum g', s2 = sum g' * xhat   (g' = gy masked by relu); grid (C, SPLIT)
// UP: the forward wrote its output through a nearest x2 up-sampling (y and gy are [N,C,2H,2W], W = width of x): the
// gradient of the small map is the sum of its 2x2 copies, the ReLU mask is read from the top-left copy
template <bool UP>
__device__ __forceinline__ float bn_gy(const float* gy, const float* y, size_t idx, long sp, int W, int relu) {
    if (!UP) {
        float g = gy[idx];
        if (relu && !(y[idx] > 0.f)) g = 0.f;
        return g;
    }
    const long h = sp / W, w = sp - h * W;
    const size_t plane = idx - sp;                       // (n*C + c) * S
    const size_t o = plane * 4 + (size_t)(2 * h) * (2 * W) + 2 * w;
    float g = (gy[o] + gy[o + 1]) + (gy[o + 2 * W] + gy[o + 2 * W + 1]);
    if (relu && !(y[o] > 0.f)) g = 0.f;
    return g;
}
template <bool UP>
__global__ __launch_bounds__(256) void bn_bwd_part_k(const float* gy, const float* x, const float* y, const float* stats,
                                                     float* part, int N, int C, long S, int relu, int split, int W) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const float mean = stats[c], istd = stats[C + c];
    const long total = (long)N * S;
    const long per = (total + split - 1) / split;
    const long e0 = (long)blockIdx.y * per;
    long e1 = e0 + per;
    if (e1 > total) e1 = total;
    float s1 = 0.f, s2 = 0.f;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const long n = e / S, sp = e - n * S;
        const size_t idx = ((size_t)n * C + c) * S + sp;
        const float g = bn_gy<UP>(gy, y, idx, sp, W, relu);
        s1 += g;
        s2 += g * (x[idx] - mean) * istd;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        float* p = part + ((size_t)c * split + blockIdx.y) * 2;
        p[0] = s1; p[1] = s2;
    }
}
__global__ void bn_bwd_final_k(const float* part, float* ws, float* ggamma, float* gbeta, int C, int split) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < split; ++k) { s1 += part[((size_t)c * split + k) * 2]; s2 += part[((size_t)c * split + k) * 2 + 1]; }
    ws[c] = s1; ws[C + c] = s2;
    gbeta[c] = s1; ggamma[c] = s2;
}
__global__ void bn_bwd_apply_k(const float* gy, const float* x, const float* y, const float* stats, const float* gamma,
                               const float* ws, float* gx, int N, int C, long S, int relu) {
    const long n = (long)N * C * S;
    const float invcnt = 1.f / (float)((long)N * S);
    GRID_STRIDE(i, n) {
        const int c = (i / S) % C;
        float g = gy[i];
        if (relu && !(y[i] > 0.f)) g = 0.f;
        const float istd = stats[C + c];
        const float xh = (x[i] - stats[c]) * istd;
        gx[i] = gamma[c] * istd * (g - ws[c] * invcnt - xh * ws[C + c] * invcnt);
    }
}
// Fused second passes: every workgroup (channel c, slice y) first merges the channel's partial statistics itself — a
// short serial Chan merge / sum, identical in every workgroup of the channel — and then normalises its slice; the
// y == 0 workgroup also publishes the merged values (stats, running stats / dgamma, dbeta). No `final` launch.
template <bool UP>
__global__ __launch_bounds__(256) void bn_apply_merge_k(const float* x, const float* part, int split, const float* gamma,
                                                        const float* beta, float* y, float* stats, float* rmean, float* rvar,
                                                        int N, int C, long S, float momentum, float eps, int relu,
                                                        long long* batches_tracked, int W) {
    const int c = blockIdx.x;
    if (batches_tracked && c == 0 && blockIdx.y == 0 && threadIdx.x == 0) *batches_tracked += 1;     // nn.BatchNorm's step counter
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int k = 0; k < split; ++k) {
        const float* p = part + ((size_t)c * split + k) * 3;
        const float nb = p[0];
        if (nb <= 0.f) continue;
        const float delta = p[1] - mean, nt = n + nb;
        mean += delta * (nb / nt);
        m2 += p[2] + delta * delta * (n * nb / nt);
        n = nt;
    }
    const float var = m2 / n;
    const float istd = 1.f / sqrtf(var + eps);
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        stats[c] = mean;
        stats[C + c] = istd;
        if (rmean) {
            const float unb = n > 1.f ? m2 / (n - 1.f) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
        }
    }
    const float sc = istd * gamma[c], sh = beta[c];
    const long total = (long)N * S;
    const long per = (total + gridDim.y - 1) / gridDim.y;
    const long e0 = (long)blockIdx.y * per;
    long e1 = e0 + per;
    if (e1 > total) e1 = total;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const long nn = e / S, sp = e - nn * S;
        const size_t idx = ((size_t)nn * C + c) * S + sp;
        float v = (x[idx] - mean) * sc + sh;
        v = relu ? fmaxf(v, 0.f) : v;
        if (UP) {                                        // BN -> ReLU -> Upsample(2) of an UpBlock in one pass
            const long h = sp / W, w = sp - h * W;
            float* q = y + (idx - sp) * 4 + (size_t)(2 * h) * (2 * W) + 2 * w;
            q[0] = v; q[1] = v; q[2 * W] = v; q[2 * W + 1] = v;
        } else y[idx] = v;
    }
}
template <bool UP>
__global__ __launch_bounds__(256) void bn_bwd_apply_merge_k(const float* gy, const float* x, const float* y, const float* stats,
                                                            const float* gamma, const float* part, int split, float* gx,
                                                            float* ggamma, float* gbeta, int N, int C, long S, int relu, int W,
                                                            const float* __restrict__ add = nullptr) {
    // add: the gradient the input received from its OTHER consumers (t2v_bn_train_bwd_add): summed in here, no add launch
    const int c = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < split; ++k) { s1 += part[((size_t)c * split + k) * 2]; s2 += part[((size_t)c * split + k) * 2 + 1]; }
    if (blockIdx.y == 0 && threadIdx.x == 0) { gbeta[c] = s1; ggamma[c] = s2; }
    const float mean = stats[c], istd = stats[C + c];
    const float invcnt = 1.f / (float)((long)N * S);
    const float k0 = gamma[c] * istd, a1 = s1 * invcnt, a2 = s2 * invcnt;
    const long total = (long)N * S;
    const long per = (total + gridDim.y - 1) / gridDim.y;
    const long e0 = (long)blockIdx.y * per;
    long e1 = e0 + per;
    if (e1 > total) e1 = total;
    for (long e = e0 + threadIdx.x; e < e1; e += 256) {
        const long nn = e / S, sp = e - nn * S;
        const size_t idx = ((size_t)nn * C + c) * S + sp;
        const float g = bn_gy<UP>(gy, y, idx, sp, W, relu);
        const float xh = (x[idx] - mean) * istd;
        const float v = k0 * (g - a1 - xh * a2);
        gx[idx] = add ? v + add[idx] : v;
    }
}
static int bn_slices(int N, int C, long S) {           // workgroups per channel of the fused second passes
    const long total = (long)N * S;
    long sl = (BN_APPLY_WGS + C - 1) / C;
    if (sl > (total + BN_MIN_ELEMS - 1) / BN_MIN_ELEMS) sl = (total + BN_MIN_ELEMS - 1) / BN_MIN_ELEMS;
    if (sl < 1) sl = 1;
    if (sl > 1024) sl = 1024;
    return (int)sl;
}
__global__ void bn_eval_k(const float* x, const float* rm, const float* rv, const float* gamma, const float* beta, float* y,
                          int N, int C, long S, float eps, int relu) {
    const long n = (long)N * C * S;
    GRID_STRIDE(i, n) {
        const int c = (i / S) % C;
        float v = (x[i] - rm[c]) * (1.f / sqrtf(rv[c] + eps)) * gamma[c] + beta[c];
        y[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
extern "C" int64_t t2v_bn_ws_floats(int N, int C, int64_t S) {
    if (N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    return (int64_t)C * bn_split(N, C, (long)S) * 3 + 2 * C;
}
extern "C" int t2v_bn_stats(const float* x, float* stats, float* rm, float* rv, float* ws, int N, int C, int64_t S, float momentum,
                            float eps, void* st) {
    if (!x || !stats || !ws || N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    const int sp = bn_split(N, C, (long)S);
    T2V_LAUNCH(bn_stats_part_k, dim3(C, sp), dim3(256), 0, S_(st), x, ws, N, C, (long)S, sp);
    T2V_LAUNCH(bn_stats_final_k, dim3((C + 255) / 256), dim3(256), 0, S_(st), ws, stats, rm, rv, C, sp, momentum, eps);
    return launch_status();
}
extern "C" int t2v_bn_apply(const float* x, const float* stats, const float* gamma, const float* beta, float* y, int N, int C,
                            int64_t S, int relu, void* st) {
    if (!x || !stats || !gamma || !beta || !y || N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    T2V_LAUNCH(bn_apply_k, dim3(nblocks((long)N * C * S)), dim3(256), 0, S_(st), x, stats, gamma, beta, y, N, C, (long)S, relu);
    return launch_status();
}
extern "C" int t2v_bn_bwd(const float* gy, const float* x, const float* y, const float* stats, const float* gamma, float* gx,
                          float* ggamma, float* gbeta, float* ws, int N, int C, int64_t S, int relu, void* st) {
    if (!gy || !x || !stats || !gamma || !gx || !ggamma || !gbeta || !ws || N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    if (relu && !y) return T2V_EINVAL;
    const int sp = bn_split(N, C, (long)S);
    float* part = ws + 2 * C;
    T2V_LAUNCH(bn_bwd_part_k<false>, dim3(C, sp), dim3(256), 0, S_(st), gy, x, y, stats, part, N, C, (long)S, relu, sp, 1);
    T2V_LAUNCH(bn_bwd_final_k, dim3((C + 255) / 256), dim3(256), 0, S_(st), part, ws, ggamma, gbeta, C, sp);
    T2V_LAUNCH(bn_bwd_apply_k, dim3(nblocks((long)N * C * S)), dim3(256), 0, S_(st), gy, x, y, stats, gamma, ws, gx, N, C, (long)S, relu);
    return launch_status();
}
extern "C" int t2v_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* rm, float* rv,
                                float* ws, int N, int C, int64_t S, float momentum, float eps, int relu, int64_t* batches_tracked,
                                void* st) {
    if (!x || !gamma || !beta || !y || !stats || !ws || N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    const int sp = bn_split(N, C, (long)S);
    T2V_LAUNCH(bn_stats_part_k, dim3(C, sp), dim3(256), 0, S_(st), x, ws, N, C, (long)S, sp);
    T2V_LAUNCH(bn_apply_merge_k<false>, dim3(C, bn_slices(N, C, (long)S)), dim3(256), 0, S_(st), x, ws, sp, gamma, beta, y, stats, rm, rv, N,
               C, (long)S, momentum, eps, relu, (long long*)batches_tracked, 1);
    return launch_status();
}
// BatchNorm2d -> [ReLU] -> Upsample(2) (the head of UpBlock's main path, layers.py:152-195) in the same two launches: x is
// [N,C,H,W], y is [N,C,2H,2W]
extern "C" int t2v_bn_train_fwd_up(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* rm,
                                   float* rv, float* ws, int N, int C, int H, int W, float momentum, float eps, int relu,
                                   int64_t* batches_tracked, void* st) {
    if (!x || !gamma || !beta || !y || !stats || !ws || N < 1 || C < 1 || H < 1 || W < 1) return T2V_EINVAL;
    const long S = (long)H * W;
    const int sp = bn_split(N, C, S);
    T2V_LAUNCH(bn_stats_part_k, dim3(C, sp), dim3(256), 0, S_(st), x, ws, N, C, S, sp);
    T2V_LAUNCH(bn_apply_merge_k<true>, dim3(C, bn_slices(N, C, S)), dim3(256), 0, S_(st), x, ws, sp, gamma, beta, y, stats, rm, rv, N,
               C, S, momentum, eps, relu, (long long*)batches_tracked, W);
    return launch_status();
}
// its adjoint: gy and y are the up-sampled [N,C,2H,2W] tensors, gx is [N,C,H,W]
extern "C" int t2v_bn_train_bwd_up(const float* gy, const float* x, const float* y, const float* stats, const float* gamma, float* gx,
                                   float* ggamma, float* gbeta, float* ws, int N, int C, int H, int W, int relu, void* st) {
    if (!gy || !x || !stats || !gamma || !gx || !ggamma || !gbeta || !ws || N < 1 || C < 1 || H < 1 || W < 1) return T2V_EINVAL;
    if (relu && !y) return T2V_EINVAL;
    const long S = (long)H * W;
    const int sp = bn_split(N, C, S);
    T2V_LAUNCH(bn_bwd_part_k<true>, dim3(C, sp), dim3(256), 0, S_(st), gy, x, y, stats, ws, N, C, S, relu, sp, W);
    T2V_LAUNCH(bn_bwd_apply_merge_k<true>, dim3(C, bn_slices(N, C, S)), dim3(256), 0, S_(st), gy, x, y, stats, gamma, ws, sp, gx,
               ggamma, gbeta, N, C, S, relu, W);
    return launch_status();
}
extern "C" int t2v_bn_train_bwd(const float* gy, const float* x, const float* y, const float* stats, const float* gamma, float* gx,
                                float* ggamma, float* gbeta, float* ws, int N, int C, int64_t S, int relu, void* st) {
    if (!gy || !x || !stats || !gamma || !gx || !ggamma || !gbeta || !ws || N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    if (relu && !y) return T2V_EINVAL;
    const int sp = bn_split(N, C, (long)S);
    T2V_LAUNCH(bn_bwd_part_k<false>, dim3(C, sp), dim3(256), 0, S_(st), gy, x, y, stats, ws, N, C, (long)S, relu, sp, 1);
    T2V_LAUNCH(bn_bwd_apply_merge_k<false>, dim3(C, bn_slices(N, C, (long)S)), dim3(256), 0, S_(st), gy, x, y, stats, gamma, ws, sp, gx,
               ggamma, gbeta, N, C, (long)S, relu, 1);
    return launch_status();
}
// t2v_bn_train_bwd / t2v_bn_train_bwd_up (`up`) with gx = (BatchNorm input gradient) + gx_add: the block input of an UpBlock /
// the abstract map in front of a RenderBlock feeds the BatchNorm AND another consumer (layers.py:152-195, tganv2/gen.py:
// 107-116); the other consumer's gradient is summed into this pass instead of by a separate add launch of the autograd engine.
extern "C" int t2v_bn_train_bwd_add(const float* gy, const float* x, const float* y, const float* stats, const float* gamma,
                                    const float* gx_add, float* gx, float* ggamma, float* gbeta, float* ws, int N, int C, int H,
                                    int W, int up, int relu, void* st) {
    if (!gy || !x || !stats || !gamma || !gx || !ggamma || !gbeta || !ws || N < 1 || C < 1 || H < 1 || W < 1) return T2V_EINVAL;
    if (relu && !y) return T2V_EINVAL;
    const long S = (long)H * W;
    const int sp = bn_split(N, C, S);
    if (up) {
        T2V_LAUNCH(bn_bwd_part_k<true>, dim3(C, sp), dim3(256), 0, S_(st), gy, x, y, stats, ws, N, C, S, relu, sp, W);
        T2V_LAUNCH(bn_bwd_apply_merge_k<true>, dim3(C, bn_slices(N, C, S)), dim3(256), 0, S_(st), gy, x, y, stats, gamma, ws, sp, gx,
                   ggamma, gbeta, N, C, S, relu, W, gx_add);
    } else {
        T2V_LAUNCH(bn_bwd_part_k<false>, dim3(C, sp), dim3(256), 0, S_(st), gy, x, y, stats, ws, N, C, S, relu, sp, 1);
        T2V_LAUNCH(bn_bwd_apply_merge_k<false>, dim3(C, bn_slices(N, C, S)), dim3(256), 0, S_(st), gy, x, y, stats, gamma, ws, sp, gx,
                   ggamma, gbeta, N, C, S, relu, 1, gx_add);
    }
    return launch_status();
}
extern "C" int t2v_bn_eval(const float* x, const float* rm, const float* rv, const float* gamma, const float* beta, float* y,
                           int N, int C, int64_t S, float eps, int relu, void* st) {
    if (!x || !rm || !rv || !gamma || !beta || !y || N < 1 || C < 1 || S < 1) return T2V_EINVAL;
    T2V_LAUNCH(bn_eval_k, dim3(nblocks((long)N * C * S)), dim3(256), 0, S_(st), x, rm, rv, gamma, beta, y, N, C, (long)S, eps, relu);
    return launch_status();
}

// ---------------------------------------------------------------- ConvLSTM gates
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

// pre / act / gpre are [B][4][C*S] (gate order i,f,c,o): exactly the NCHW output of ONE convolution whose 4C output
// channels are the four gates (their weights are packed side by side).
__global__ void lstm_gates_k(const float* pre, const float* c_prev, float* h, float* c_new, float* act, int B, long CS) {
    const long n = (long)B * CS;
    GRID_STRIDE(i, n) {
        const long b = i / CS, r = i - b * CS;
        const float* pp = pre + b * 4 * CS + r;
        const float gi = sigm(pp[0]), gf = sigm(pp[CS]), gg = tanhf(pp[2 * CS]), go = sigm(pp[3 * CS]);
        const float cc = gf * c_prev[i] + gi * gg;
        c_new[i] = cc;
        h[i] = go * tanhf(cc);
        float* pa = act + b * 4 * CS + r;
        pa[0] = gi; pa[CS] = gf; pa[2 * CS] = gg; pa[3 * CS] = go;
    }
}
__global__ void lstm_gates_bwd_k(const float* gh, const float* gc_in, const float* act, const float* c_prev, const float* c_new,
                                 float* gpre, float* gc_prev, int B, long CS) {
    const long n = (long)B * CS;
    GRID_STRIDE(i, n) {
        const long b = i / CS, r = i - b * CS;
        const float* pa = act + b * 4 * CS + r;
        const float gi = pa[0], gf = pa[CS], gg = pa[2 * CS], go = pa[3 * CS];
        const float tc = tanhf(c_new[i]);
        const float dh = gh[i];
        const float dc = dh * go * (1.f - tc * tc) + (gc_in ? gc_in[i] : 0.f);
        float* pg = gpre + b * 4 * CS + r;
        pg[0] = dc * gg * gi * (1.f - gi);
        pg[CS] = dc * c_prev[i] * gf * (1.f - gf);
        pg[2 * CS] = dc * gi * (1.f - gg * gg);
        pg[3 * CS] = dh * tc * go * (1.f - go);
        gc_prev[i] = dc * gf;
    }
}
extern "C" int t2v_lstm_gates(const float* pre, const float* c_prev, float* h, float* c_new, float* act, int B, int64_t CS, void* st) {
    if (!pre || !c_prev || !h || !c_new || !act || B < 1 || CS < 1) return T2V_EINVAL;
    T2V_LAUNCH(lstm_gates_k, dim3(nblocks((long)B * CS)), dim3(256), 0, S_(st), pre, c_prev, h, c_new, act, B, (long)CS);
    return launch_status();
}
extern "C" int t2v_lstm_gates_bwd(const float* gh, const float* gc_in, const float* act, const float* c_prev, const float* c_new,
                                  float* gpre, float* gc_prev, int B, int64_t CS, void* st) {
    if (!gh || !act || !c_prev || !c_new || !gpre || !gc_prev || B < 1 || CS < 1) return T2V_EINVAL;
    T2V_LAUNCH(lstm_gates_bwd_k, dim3(nblocks((long)B * CS)), dim3(256), 0, S_(st), gh, gc_in, act, c_prev, c_new, gpre, gc_prev, B, (long)CS);
    return launch_status();
}

// ---- the recurrence on 1x1 feature maps (the TGANv2 frame-seed generator's ConvLSTM, conv_lstm.py:75-97, runs on
// [B,1024,1,1]): each step is a [B x K] . [K x N] product with B = 32 rows — one MFMA row tile. A tiled GEMM spends its
// time in prologue / epilogue / a separate split-K pass (25-37 us per step); here one WAVE owns a 32-column strip and a
// K slice, streams its weights straight from memory into the MFMA B operand (no LDS, no barrier: nothing is shared between
// waves) and writes its partial sums to a slab; the gate kernels below add the slab's slices in a fixed order.
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// slab[s][m][n] = sum_{k in [s KW, (s+1) KW)} x[m][k] * w[k][n]      x: [M][K] row-major, w: [K][N] row-major
template <int KW>
__global__ __launch_bounds__(64) void skinny_gemm_slab_k(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ slab, const int M, const int K, const int N) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    const int n = blockIdx.x * 32 + l31, s = blockIdx.y, m = blockIdx.z * 32 + l31;
    const int k0 = s * KW + 4 * hi;
    const float* __restrict__ px = x + (size_t)(m < M ? m : M - 1) * K + k0;        // clamped row, masked below
    const float* __restrict__ pw = w + (size_t)k0 * N + (n < N ? n : N - 1);
    const float am = m < M ? 1.f : 0.f;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // MFMA 32x32x2: lane (l31, hi) supplies A[m = l31][k] and B[k][n = l31] for ONE k per issue; the lane's four k of an
    // 8-wide group are k0 + kb + {0,1,2,3} (+4 for hi = 1) — any assignment works as long as A and B agree.
#pragma unroll
    for (int kb = 0; kb < KW; kb += 8) {
        const float4 a = *(const float4*)(px + kb);
        const float b0 = pw[(size_t)(kb + 0) * N], b1 = pw[(size_t)(kb + 1) * N], b2 = pw[(size_t)(kb + 2) * N],
                    b3 = pw[(size_t)(kb + 3) * N];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x * am, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y * am, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z * am, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w * am, b3, acc, 0, 0, 0);
    }
    if (n < N) {
        float* out = slab + (size_t)s * M * N + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mr = blockIdx.z * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (mr < M) out[(size_t)mr * N] = acc[r];
        }
    }
}
// K slice per wave: 128 keeps >= 1 wave per SIMD busy for the [32 x 1024 x 4096] steps; must divide K
extern "C" int t2v_skinny_gemm_splits(int M, int K, int N) {
    if (M < 1 || K < 1 || N < 1 || (K % 128) != 0) return T2V_EINVAL;
    const long waves256 = (long)(K / 256) * ((N + 31) / 32) * ((M + 31) / 32);
    return ((K % 256) == 0 && waves256 >= 1024) ? K / 256 : K / 128;
}
extern "C" int t2v_skinny_gemm_slab(const float* x, const float* w, float* slab, int M, int K, int N, void* st) {
    const int S = t2v_skinny_gemm_splits(M, K, N);
    if (!x || !w || !slab || S < 1 || S > 65535) return T2V_EINVAL;
    dim3 grid((N + 31) / 32, S, (M + 31) / 32);
    if (K / S == 256) T2V_LAUNCH(skinny_gemm_slab_k<256>, grid, dim3(64), 0, S_(st), x, w, slab, M, K, N);
    else T2V_LAUNCH(skinny_gemm_slab_k<128>, grid, dim3(64), 0, S_(st), x, w, slab, M, K, N);
    return launch_status();
}

// gate kernels that take their pre-activation / incoming dL/dh as  sum_s slab[s]  (+ bias, + a second addend)
__global__ void lstm_gates_slab_k(const float* slab, int S, const float* bias, const float* c_prev, float* h, float* c_new,
                                  float* act, int B, int Cc) {
    const long n = (long)B * Cc, stride = n * 4;
    GRID_STRIDE(i, n) {
        const long b = i / Cc, r = i - b * Cc;
        float p[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float* ps = slab + b * 4 * Cc + (long)g * Cc + r;
            float v = 0.f;
#pragma unroll 8
            for (int s = 0; s < S; ++s) v += ps[(long)s * stride];      // unrolled: 8 loads in flight, summed in order
            p[g] = v + (bias ? bias[g * Cc + r] : 0.f);
        }
        const float gi = sigm(p[0]), gf = sigm(p[1]), gg = tanhf(p[2]), go = sigm(p[3]);
        const float cc = gf * c_prev[i] + gi * gg;
        c_new[i] = cc;
        h[i] = go * tanhf(cc);
        float* pa = act + b * 4 * Cc + r;
        pa[0] = gi; pa[Cc] = gf; pa[2 * Cc] = gg; pa[3 * Cc] = go;
    }
}
__global__ void lstm_gates_bwd_slab_k(const float* gh, const float* slab, int S, const float* gc_in, const float* act,
                                      const float* c_prev, const float* c_new, float* gpre, float* gc_prev, int B, int Cc) {
    const long n = (long)B * Cc;
    GRID_STRIDE(i, n) {
        const long b = i / Cc, r = i - b * Cc;
        const float* pa = act + b * 4 * Cc + r;
        const float gi = pa[0], gf = pa[Cc], gg = pa[2 * Cc], go = pa[3 * Cc];
        const float tc = tanhf(c_new[i]);
        float back = 0.f;                                  // dL/dh_t arriving from step t+1 = sum of the slab's slices
#pragma unroll 8
        for (int s = 0; s < S; ++s) back += slab[(long)s * n + i];
        const float dh = S > 0 ? gh[i] + back : gh[i];
        const float dc = dh * go * (1.f - tc * tc) + (gc_in ? gc_in[i] : 0.f);
        float* pg = gpre + b * 4 * Cc + r;
        pg[0] = dc * gg * gi * (1.f - gi);
        pg[Cc] = dc * c_prev[i] * gf * (1.f - gf);
        pg[2 * Cc] = dc * gi * (1.f - gg * gg);
        pg[3 * Cc] = dh * tc * go * (1.f - go);
        gc_prev[i] = dc * gf;
    }
}
extern "C" int t2v_lstm_gates_slab(const float* slab, int S, const float* bias, const float* c_prev, float* h, float* c_new,
                                   float* act, int B, int C, void* st) {
    if (!slab || S < 1 || !c_prev || !h || !c_new || !act || B < 1 || C < 1) return T2V_EINVAL;
    T2V_LAUNCH(lstm_gates_slab_k, dim3((unsigned)(((long)B * C + 63) / 64)), dim3(64), 0, S_(st), slab, S, bias, c_prev, h, c_new, act, B, C);
    return launch_status();
}
extern "C" int t2v_lstm_gates_bwd_slab(const float* gh, const float* slab, int S, const float* gc_in, const float* act,
                                       const float* c_prev, const float* c_new, float* gpre, float* gc_prev, int B, int C, void* st) {
    if (!gh || S < 0 || (S > 0 && !slab) || !act || !c_prev || !c_new || !gpre || !gc_prev || B < 1 || C < 1) return T2V_EINVAL;
    T2V_LAUNCH(lstm_gates_bwd_slab_k, dim3((unsigned)(((long)B * C + 63) / 64)), dim3(64), 0, S_(st), gh, slab, S, gc_in, act, c_prev, c_new,
               gpre, gc_prev, B, C);
    return launch_status();
}

// out[g*n + i] = src_g[i], g = 0..3: the four gate biases side by side in one launch (was four 4-us copies per forward pass)
__global__ void concat4_k(const float* a, const float* b, const float* c, const float* d, float* out, int n) {
    GRID_STRIDE(i, 4L * n) {
        const int g = (int)(i / n), r = (int)(i - (long)g * n);
        const float* src = g == 0 ? a : g == 1 ? b : g == 2 ? c : d;
        out[i] = src[r];
    }
}
extern "C" int t2v_concat4(const float* a, const float* b, const float* c, const float* d, float* out, int n, void* st) {
    if (!a || !b || !c || !d || !out || n < 1) return T2V_EINVAL;
    T2V_LAUNCH(concat4_k, dim3(nblocks(4L * n)), dim3(256), 0, S_(st), a, b, c, d, out, n);
    return launch_status();
}
// ---- one launch per recurrence step: GEMM + gate math fused (replaces skinny_gemm_slab + lstm_gates_slab, 17.5 -> ~7 us per step).
// A workgroup owns FOUR hidden units = 16 of the 4C gate columns (i,f,c,o of each unit), so the gate math needs nothing from any other
// workgroup. Its 16 weight columns live contiguously in the "unit-major" copy  wr[u][k/4][g*4 + j][k%4] = wp[k][g*C + 4u + j]  (lstm_pack_major_k,
// one streaming pass per weight update); the four waves split K, each runs two 16x16x4 fp32 MFMA row tiles (32 batch rows) and the partial
// sums meet in LDS. Workgroups are numbered so that the 32 of one XCD take consecutive units (their weight lines share that XCD's L2).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// tile = 8 k x 64 units: reads run along the units (1 KB per (k, gate)); a thread then gathers the four k of one (unit, column) into
// the float4 the MFMA lane will load, 1 KB of output per unit
__global__ __launch_bounds__(256) void lstm_pack_major_k(const float* __restrict__ wp, float* __restrict__ wr, int K, int Cc) {
    __shared__ float tile[8 * 4][260];
    const int U = Cc / 4, u0 = blockIdx.x * 64, k0 = blockIdx.y * 8;
    for (int i = threadIdx.x; i < 8 * 4 * 64; i += 256) {
        const int uu = i & 63, kg = i >> 6, g = kg & 3, kk = kg >> 2;
        if (u0 + uu < U && k0 + kk < K) *(float4*)&tile[kg][4 * uu] = *(const float4*)(wp + ((size_t)(k0 + kk) * 4 + g) * Cc + 4 * (u0 + uu));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 2 * 16; i += 256) {
        const int c16 = i & 15, kgrp = (i >> 4) & 1, uu = i >> 5, g = c16 >> 2, j = c16 & 3;
        if (u0 + uu < U && k0 + 4 * kgrp < K) {
            const float4 v = {tile[(4 * kgrp + 0) * 4 + g][4 * uu + j], tile[(4 * kgrp + 1) * 4 + g][4 * uu + j],
                              tile[(4 * kgrp + 2) * 4 + g][4 * uu + j], tile[(4 * kgrp + 3) * 4 + g][4 * uu + j]};
            *(float4*)(wr + ((((size_t)(u0 + uu) * K + k0) >> 2) + kgrp) * 64 + 4 * c16) = v;
        }
    }
}
template <int KWT, int NW>                                               // K slice per wave when known at compile time (0: any); waves
__global__ __launch_bounds__(64 * NW) void lstm_step_fused_k(const float* __restrict__ xin, const float* __restrict__ wr,
                                                         const float* __restrict__ bias, const float* __restrict__ c_prev,
                                                         float* __restrict__ h, float* __restrict__ c_new, float* __restrict__ act,
                                                         const int B, const int K, const int Cc) {
    __shared__ float part[NW][32][17];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int u = (nwg % 8 == 0) ? (bid % 8) * (nwg / 8) + bid / 8 : bid;
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, c16 = l & 15, kq = l >> 4;
    const int KW = KWT ? KWT : K / NW;                                   // K slice of this wave (multiple of 16)
    const float* __restrict__ pw = wr + ((((size_t)u * K + (size_t)wv * KW) >> 2) + kq) * 64 + 4 * c16;   // [u][k/4][16][4]
    const int r0 = c16 < B ? c16 : B - 1, r1 = 16 + c16 < B ? 16 + c16 : B - 1;      // clamped rows: computed, never stored
    const float* __restrict__ pa0 = xin + (size_t)r0 * K + (size_t)wv * KW + 4 * kq;
    const float* __restrict__ pa1 = xin + (size_t)r1 * K + (size_t)wv * KW + 4 * kq;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // 16x16x4: lane (c16, kq) supplies A[row c16][k-slot kq] and B[k-slot kq][col c16]; the four k of a lane's float4 go to four issues
#define LSTM_KSTEP(kb) {                                                                              \
        const float4 a0 = *(const float4*)(pa0 + (kb)), a1 = *(const float4*)(pa1 + (kb));                \
        const float4 b = *(const float4*)(pw + (size_t)(kb) * 16);                                        \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc0, 0, 0, 0);                            \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc1, 0, 0, 0);                            \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc0, 0, 0, 0);                            \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc1, 0, 0, 0);                            \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc0, 0, 0, 0);                            \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc1, 0, 0, 0);                            \
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc0, 0, 0, 0);                            \
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc1, 0, 0, 0); }
    if (KWT) {
#pragma unroll
        for (int kb = 0; kb < KWT; kb += 16) LSTM_KSTEP(kb)
    } else {
        for (int kb = 0; kb < KW; kb += 16) LSTM_KSTEP(kb)
    }
#undef LSTM_KSTEP
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                        // D: col = lane & 15, row = 4 (lane >> 4) + r
        part[wv][4 * kq + r][c16] = acc0[r];
        part[wv][16 + 4 * kq + r][c16] = acc1[r];
    }
    __syncthreads();
    if (tid < 128) {
        const int b = tid >> 2, j = tid & 3;
        if (b < B) {
            float p[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = 0.f;
#pragma unroll
                for (int q = 0; q < NW; ++q) v += part[q][b][4 * g + j];
                p[g] = v + (bias ? bias[g * Cc + 4 * u + j] : 0.f);
            }
            const float gi = sigm(p[0]), gf = sigm(p[1]), gg = tanhf(p[2]), go = sigm(p[3]);
            const size_t i = (size_t)b * Cc + 4 * u + j;
            const float cc = gf * c_prev[i] + gi * gg;
            c_new[i] = cc;
            h[i] = go * tanhf(cc);
            float* pa = act + (size_t)b * 4 * Cc + 4 * u + j;
            pa[0] = gi; pa[Cc] = gf; pa[2 * (size_t)Cc] = gg; pa[3 * (size_t)Cc] = go;
        }
    }
}
// B <= 32 rows (one pair of MFMA row tiles), K a multiple of 128 (up to eight waves x 16-wide k groups), C a multiple of 4
extern "C" int t2v_lstm_step_fused_ok(int B, int K, int C) {
    return (B >= 1 && B <= 32 && K >= 128 && (K % 128) == 0 && C >= 4 && (C % 4) == 0) ? 1 : 0;
}
static int lstm_waves() { static const int w = getenv("T2V_LSTM_WAVES") ? atoi(getenv("T2V_LSTM_WAVES")) : 8; return w == 4 ? 4 : 8; }
extern "C" int t2v_lstm_pack_major(const float* wp, float* wr, int K, int C, void* st) {
    if (!wp || !wr || K < 1 || C < 4 || (C % 4) != 0) return T2V_EINVAL;
    T2V_LAUNCH(lstm_pack_major_k, dim3((unsigned)((C / 4 + 63) / 64), (unsigned)((K + 7) / 8)), dim3(256), 0, S_(st), wp, wr, K, C);
    return launch_status();
}
extern "C" int t2v_lstm_step_fused(const float* x, const float* wr, const float* bias, const float* c_prev, float* h, float* c_new,
                                   float* act, int B, int K, int C, void* st) {
    if (!x || !wr || !c_prev || !h || !c_new || !act || !t2v_lstm_step_fused_ok(B, K, C)) return T2V_EINVAL;
    const dim3 grid((unsigned)(C / 4));
    if (lstm_waves() == 4) {
        if (K == 1024) T2V_LAUNCH((lstm_step_fused_k<256, 4>), grid, dim3(256), 0, S_(st), x, wr, bias, c_prev, h, c_new, act, B, K, C);
        else T2V_LAUNCH((lstm_step_fused_k<0, 4>), grid, dim3(256), 0, S_(st), x, wr, bias, c_prev, h, c_new, act, B, K, C);
    } else {
        if (K == 1024) T2V_LAUNCH((lstm_step_fused_k<128, 8>), grid, dim3(512), 0, S_(st), x, wr, bias, c_prev, h, c_new, act, B, K, C);
        else T2V_LAUNCH((lstm_step_fused_k<0, 8>), grid, dim3(512), 0, S_(st), x, wr, bias, c_prev, h, c_new, act, B, K, C);
    }
    return launch_status();
}

// column-block-major, k-interleaved copy of a row-major matrix: wr[cb][r / 4][j][r % 4] = w[r][16 cb + j] (the adjoint step's 16-unit
// weight stream). Tile = 16 rows x 256 columns: reads 1 KB per row, writes 1 KB per column block.
__global__ __launch_bounds__(256) void lstm_pack_cols_k(const float* __restrict__ w, float* __restrict__ wr, int R, int Cn) {
    __shared__ float tile[16][260];
    const int c0 = blockIdx.x * 256, r0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int q = i & 63, rr = i >> 6;
        if (c0 + 4 * q < Cn && r0 + rr < R) *(float4*)&tile[rr][4 * q] = *(const float4*)(w + (size_t)(r0 + rr) * Cn + c0 + 4 * q);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * 4 * 16; i += 256) {
        const int j = i & 15, rg = (i >> 4) & 3, cbl = i >> 6;           // column j of block cbl, row group rg (4 rows)
        const int cb = (c0 >> 4) + cbl;
        if (16 * cb < Cn && r0 + 4 * rg < R) {
            const float4 v = {tile[4 * rg][16 * cbl + j], tile[4 * rg + 1][16 * cbl + j], tile[4 * rg + 2][16 * cbl + j],
                              tile[4 * rg + 3][16 * cbl + j]};
            *(float4*)(wr + ((((size_t)cb * R + r0) >> 2) + rg) * 64 + 4 * j) = v;
        }
    }
}
// The adjoint step in one launch: dL/dh_t = gh_t + gpre_{t+1} [B][4C] . w1 [4C][C], then the gate adjoints of t2v_lstm_gates_bwd. w1
// arrives as its column-block-major copy w1r[ub][n][16] (lstm_pack_cols_k; reading the 64-byte pieces of the row-major form cost 21 us
// per step against 17 for the two-launch form). A workgroup owns 16 hidden
// units x 16 batch rows = one 16x16x4 MFMA tile; four waves split K = 4C. Workgroup pairs (the two batch halves read the same weights)
// and neighbouring unit blocks share an XCD.
template <int CT, int NW>                                                // C when known at compile time (0: any); waves (4 | 8)
__global__ __launch_bounds__(64 * NW) void lstm_step_bwd_fused_k(const float* __restrict__ gh, const float* __restrict__ gnext,
                                                             const float* __restrict__ w1, const float* __restrict__ gc_in,
                                                             const float* __restrict__ act, const float* __restrict__ c_prev,
                                                             const float* __restrict__ c_new, float* __restrict__ gpre,
                                                             float* __restrict__ gc_prev, const int B, const int Cc, const int halves) {
    __shared__ float part[NW][16][17];
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int v = (nwg % 8 == 0) ? (bid % 8) * (nwg / 8) + bid / 8 : bid;
    const int ub = v / halves, bh = v - ub * halves;
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, c16 = l & 15, kq = l >> 4;
    if (gnext) {
        const int K = 4 * Cc, KW = CT ? CT * 4 / NW : Cc * 4 / NW;       // K slice of this wave
        const int row = 16 * bh + c16 < B ? 16 * bh + c16 : B - 1;
        const float* __restrict__ pa = gnext + (size_t)row * K + (size_t)wv * KW + 4 * kq;
        const float* __restrict__ pw = w1 + ((((size_t)ub * K + (size_t)wv * KW) >> 2) + kq) * 64 + 4 * c16;   // [ub][k/4][16][4]
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#define LSTM_BSTEP(kb) {                                                                                  \
            const float4 a = *(const float4*)(pa + (kb)), a2 = *(const float4*)(pa + (kb) + 16);         \
            const float4 b = *(const float4*)(pw + (size_t)(kb) * 16), d = *(const float4*)(pw + (size_t)(kb) * 16 + 256);     \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);                           \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, d.x, acc2, 0, 0, 0);                        \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);                           \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, d.y, acc2, 0, 0, 0);                        \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);                           \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.z, d.z, acc2, 0, 0, 0);                        \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);                           \
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.w, d.w, acc2, 0, 0, 0); }
        if (CT) {
#pragma unroll 8
            for (int kb = 0; kb < CT * 4 / NW; kb += 32) LSTM_BSTEP(kb)
        } else {
            for (int kb = 0; kb < KW; kb += 32) LSTM_BSTEP(kb)
        }
#undef LSTM_BSTEP
#pragma unroll
        for (int r = 0; r < 4; ++r) part[wv][4 * kq + r][c16] = acc[r] + acc2[r];
        __syncthreads();
    }
    const int bl = (tid >> 4) & 15, c = tid & 15, b = 16 * bh + bl;
    if (tid < 256 && b < B) {
        const size_t i = (size_t)b * Cc + 16 * ub + c;
        float back = 0.f;
        if (gnext) {
#pragma unroll
            for (int q = 0; q < NW; ++q) back += part[q][bl][c];
        }
        const float* pa = act + (size_t)b * 4 * Cc + 16 * ub + c;
        const float gi = pa[0], gf = pa[Cc], gg = pa[2 * (size_t)Cc], go = pa[3 * (size_t)Cc];
        const float tc = tanhf(c_new[i]);
        const float dh = gh[i] + back;
        const float dc = dh * go * (1.f - tc * tc) + (gc_in ? gc_in[i] : 0.f);
        float* pg = gpre + (size_t)b * 4 * Cc + 16 * ub + c;
        pg[0] = dc * gg * gi * (1.f - gi);
        pg[Cc] = dc * c_prev[i] * gf * (1.f - gf);
        pg[2 * (size_t)Cc] = dc * gi * (1.f - gg * gg);
        pg[3 * (size_t)Cc] = dh * tc * go * (1.f - go);
        gc_prev[i] = dc * gf;
    }
}
// gnext = NULL: the last step (nothing arrives from a later one). w1r = t2v_lstm_pack_cols of the [4C][C] data-gradient weight. Same
// shape rules as the forward step plus C % 32.
extern "C" int t2v_lstm_pack_cols(const float* w, float* wr, int R, int Cn, void* st) {
    if (!w || !wr || R < 1 || Cn < 16 || (Cn % 16) != 0) return T2V_EINVAL;
    T2V_LAUNCH(lstm_pack_cols_k, dim3((unsigned)((Cn + 255) / 256), (unsigned)((R + 15) / 16)), dim3(256), 0, S_(st), w, wr, R, Cn);
    return launch_status();
}
extern "C" int t2v_lstm_step_bwd_fused(const float* gh, const float* gnext, const float* w1, const float* gc_in, const float* act,
                                       const float* c_prev, const float* c_new, float* gpre, float* gc_prev, int B, int C, void* st) {
    if (!gh || (gnext && !w1) || !act || !c_prev || !c_new || !gpre || !gc_prev || !t2v_lstm_step_fused_ok(B, C, C) || (C % 64) != 0)
        return T2V_EINVAL;
    const int halves = (B + 15) / 16;
    const dim3 grid((unsigned)(C / 16 * halves));
    if (lstm_waves() == 4) {
        if (C == 1024) T2V_LAUNCH((lstm_step_bwd_fused_k<1024, 4>), grid, dim3(256), 0, S_(st), gh, gnext, w1, gc_in, act, c_prev, c_new, gpre,
                                  gc_prev, B, C, halves);
        else T2V_LAUNCH((lstm_step_bwd_fused_k<0, 4>), grid, dim3(256), 0, S_(st), gh, gnext, w1, gc_in, act, c_prev, c_new, gpre, gc_prev, B, C, halves);
    } else {
        if (C == 1024) T2V_LAUNCH((lstm_step_bwd_fused_k<1024, 8>), grid, dim3(512), 0, S_(st), gh, gnext, w1, gc_in, act, c_prev, c_new, gpre,
                                  gc_prev, B, C, halves);
        else T2V_LAUNCH((lstm_step_bwd_fused_k<0, 8>), grid, dim3(512), 0, S_(st), gh, gnext, w1, gc_in, act, c_prev, c_new, gpre, gc_prev, B, C, halves);
    }
    return launch_status();
}

// ---------------------------------------------------------------- non-local block
// Batched thin GEMM: head dims are 4..64, so no MFMA tile fits; 16x16 output tile per workgroup,
// K staged through LDS in 16-wide slabs.
__global__ __launch_bounds__(256) void bmm_k(const float* A, const float* B, float* C, int M, int N, int K, int ta, int tb, int accum) {
    __shared__ float As[16][17], Bs[16][17];
    const int b = blockIdx.z;
    const float* a = A + (long)b * M * K;
    const float* bb = B + (long)b * K * N;
    float* c = C + (long)b * M * N;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        {   // A tile: As[ty][tx] = opA[row0+ty][k0+tx]
            int r = blockIdx.y * 16 + ty, k = k0 + tx;
            As[ty][tx] = (r < M && k < K) ? (ta ? a[(long)k * M + r] : a[(long)r * K + k]) : 0.f;
            int kk = k0 + ty, cc = blockIdx.x * 16 + tx;
            Bs[ty][tx] = (kk < K && cc < N) ? (tb ? bb[(long)cc * K + kk] : bb[(long)kk * N + cc]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += As[ty][k] * Bs[k][tx];
        __syncthreads();
    }
    if (row < M && col < N) c[(long)row * N + col] = accum ? c[(long)row * N + col] + acc : acc;
}
// The same product with a 16 x 64 output tile (four consecutive columns per thread, 16-byte stores) and tile loads that run
// along whichever dimension is contiguous in memory for the given transposition flags: the non-local block's products move
// the N x N/4 attention matrix (hundreds of MB at 64 x 64 maps) and are bound by exactly these accesses. N % 4 == 0.
__global__ __launch_bounds__(256) void bmm_wide_k(const float* A, const float* B, float* C, int M, int N, int K, int ta, int tb, int accum) {
    __shared__ float As[16][17];                 // [row][k]
    __shared__ __attribute__((aligned(16))) float Bs[16][68];     // [k][col]
    const int b = blockIdx.z;
    const float* a = A + (long)b * M * K;
    const float* bb = B + (long)b * K * N;
    float* c = C + (long)b * M * N;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 64;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
        if (ta) {                                // a[k][r]: consecutive threads along r
            const int r = row0 + tx, k = k0 + ty;
            As[tx][ty] = (r < M && k < K) ? a[(long)k * M + r] : 0.f;
        } else {                                 // a[r][k]: consecutive threads along k
            const int r = row0 + ty, k = k0 + tx;
            As[ty][tx] = (r < M && k < K) ? a[(long)r * K + k] : 0.f;
        }
        if (tb) {                                // bb[col][k]: consecutive threads along k
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cl = ty + 16 * j, cc = col0 + cl, k = k0 + tx;
                Bs[tx][cl] = (cc < N && k < K) ? bb[(long)cc * K + k] : 0.f;
            }
        } else {                                 // bb[k][col]: consecutive threads along col
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cl = tx + 16 * j, cc = col0 + cl, k = k0 + ty;
                Bs[ty][cl] = (cc < N && k < K) ? bb[(long)k * N + cc] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float av = As[ty][k];
            const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
            acc[0] += av * bv.x; acc[1] += av * bv.y; acc[2] += av * bv.z; acc[3] += av * bv.w;
        }
        __syncthreads();
    }
    const int row = row0 + ty, col = col0 + tx * 4;
    if (row < M && col < N) {                    // N % 4 == 0: the four columns are in range together
        float4* dst = reinterpret_cast<float4*>(c + (long)row * N + col);
        float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (accum) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dst = v;
    }
}
extern "C" int t2v_bmm(const float* A, const float* B, float* C, int batch, int M, int N, int K, int ta, int tb, int accum, void* st) {
    if (!A || !B || !C || batch < 1 || M < 1 || N < 1 || K < 1 || batch > 65535) return T2V_EINVAL;
    if (N >= 64 && (N & 3) == 0 && ((uintptr_t)C & 15) == 0) {
        dim3 grid((N + 63) / 64, (M + 15) / 16, batch);
        T2V_LAUNCH(bmm_wide_k, grid, dim3(256), 0, S_(st), A, B, C, M, N, K, ta, tb, accum);
        return launch_status();
    }
    dim3 grid((N + 15) / 16, (M + 15) / 16, batch);
    T2V_LAUNCH(bmm_k, grid, dim3(256), 0, S_(st), A, B, C, M, N, K, ta, tb, accum);
    return launch_status();
}

// softmax over the last dim: one wave per row, wavefront reductions
__global__ __launch_bounds__(256) void softmax_k(const float* x, float* y, long rows, int n) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = x + row * n;
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, p[i]);
    mx = wave_max(mx);
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(p[i] - mx);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int i = lane; i < n; i += 64) y[row * n + i] = expf(p[i] - mx) * inv;
}
__global__ __launch_bounds__(256) void softmax_bwd_k(const float* y, const float* gy, float* gx, long rows, int n) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* py = y + row * n;
    const float* pg = gy + row * n;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += pg[i] * py[i];
    s = wave_sum(s);
    for (int i = lane; i < n; i += 64) gx[row * n + i] = py[i] * (pg[i] - s);
}
__global__ __launch_bounds__(256) void softmax_bwd_bwd_y_k(const float* y, const float* gy, const float* gg, float* out, long rows, int n) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* py = y + row * n;
    const float* pg = gy + row * n;
    const float* pq = gg + row * n;
    float s = 0.f, u = 0.f;
    for (int i = lane; i < n; i += 64) { s += pg[i] * py[i]; u += pq[i] * py[i]; }
    s = wave_sum(s);
    u = wave_sum(u);
    for (int i = lane; i < n; i += 64) out[row * n + i] = pq[i] * (pg[i] - s) - pg[i] * u;
}
extern "C" int t2v_softmax(const float* x, float* y, int64_t rows, int n, void* st) {
    if (!x || !y || rows < 1 || n < 1) return T2V_EINVAL;
    T2V_LAUNCH(softmax_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_(st), x, y, (long)rows, n);
    return launch_status();
}
extern "C" int t2v_softmax_bwd(const float* y, const float* gy, float* gx, int64_t rows, int n, void* st) {
    if (!y || !gy || !gx || rows < 1 || n < 1) return T2V_EINVAL;
    T2V_LAUNCH(softmax_bwd_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_(st), y, gy, gx, (long)rows, n);
    return launch_status();
}
extern "C" int t2v_softmax_bwd_bwd_y(const float* y, const float* gy, const float* gg, float* out, int64_t rows, int n, void* st) {
    if (!y || !gy || !gg || !out || rows < 1 || n < 1) return T2V_EINVAL;
    T2V_LAUNCH(softmax_bwd_bwd_y_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, S_(st), y, gy, gg, out, (long)rows, n);
    return launch_status();
}

// ---------------------------------------------------------------- fused non-local attention (layers.py:23-36)
// o[b][c][i] = sum_j softmax_j(sum_k theta[b][k][i] * phi[b][k][j]) * g[b][c][j]   for i < N queries, j < Nk keys.
// The generator's 2-D block has thin heads (C8 = ch/8 = 4, C2 = ch/2 = 16: no MFMA tile fits), so this is vector-ALU
// work; what the unfused form pays for is beta [b, N, Nk] in HBM (268 MB per materialisation at 64x64 maps, written
// and re-read by bmm / softmax / bmm and again by their adjoints). Here beta never leaves registers:
//   forward  : one lane per QUERY; phi / g tiles of 256 keys staged in LDS as [key][channel] (every lane reads the same
//              key: LDS broadcast); two passes over the keys (row max, then exp + weighted sum) = the exact
//              max-subtracted softmax of the unfused kernels; saves lse[b][i] = max + log(sum) for the adjoint;
//   backward : D_i = sum_c do[c][i] * o[c][i];  ds_ij = beta_ij * (sum_c do[c][i] g[c][j] - D_i),  beta_ij recomputed
//              from lse. One lane per query accumulates dtheta[:, i] = sum_j ds_ij phi[:, j]; a second kernel with one
//              lane per KEY (query tiles in LDS) accumulates dphi[:, j] = sum_i ds_ij theta[:, i] and
//              dg[:, j] = sum_i beta_ij do[:, i]. No atomics: deterministic.
#define NL_TILE 256
template <int C8, int C2>
__global__ __launch_bounds__(256) void nonlocal_fwd_k(const float* __restrict__ theta, const float* __restrict__ phi,
                                                      const float* __restrict__ g, float* __restrict__ o,
                                                      float* __restrict__ lse, int N, int Nk) {
    __shared__ float sk[NL_TILE * (C8 + C2)];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const bool qv = i < N;
    const float* th = theta + (size_t)b * C8 * N;
    const float* ph = phi + (size_t)b * C8 * Nk;
    const float* gg = g + (size_t)b * C2 * Nk;
    float t[C8];
#pragma unroll
    for (int k = 0; k < C8; ++k) t[k] = qv ? th[(size_t)k * N + i] : 0.f;
    // pass 1: row maximum (scores only need phi)
    float mx = -INFINITY;
    for (int j0 = 0; j0 < Nk; j0 += NL_TILE) {
        const int nj = min(NL_TILE, Nk - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < nj * C8; e += 256) { const int k = e / nj, j = e - k * nj; sk[j * C8 + k] = ph[(size_t)k * Nk + j0 + j]; }
        __syncthreads();
        for (int j = 0; j < nj; ++j) {
            float sc = 0.f;
#pragma unroll
            for (int k = 0; k < C8; ++k) sc += t[k] * sk[j * C8 + k];
            mx = fmaxf(mx, sc);
        }
    }
    // pass 2: exp, sum, weighted sum of g
    float l = 0.f, acc[C2];
#pragma unroll
    for (int c = 0; c < C2; ++c) acc[c] = 0.f;
    for (int j0 = 0; j0 < Nk; j0 += NL_TILE) {
        const int nj = min(NL_TILE, Nk - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < nj * C8; e += 256) { const int k = e / nj, j = e - k * nj; sk[j * (C8 + C2) + k] = ph[(size_t)k * Nk + j0 + j]; }
        for (int e = threadIdx.x; e < nj * C2; e += 256) { const int c = e / nj, j = e - c * nj; sk[j * (C8 + C2) + C8 + c] = gg[(size_t)c * Nk + j0 + j]; }
        __syncthreads();
        for (int j = 0; j < nj; ++j) {
            const float* r = &sk[j * (C8 + C2)];
            float sc = 0.f;
#pragma unroll
            for (int k = 0; k < C8; ++k) sc += t[k] * r[k];
            const float p = expf(sc - mx);
            l += p;
#pragma unroll
            for (int c = 0; c < C2; ++c) acc[c] += p * r[C8 + c];
        }
    }
    if (qv) {
        const float inv = 1.f / l;
        float* po = o + (size_t)b * C2 * N + i;
#pragma unroll
        for (int c = 0; c < C2; ++c) po[(size_t)c * N] = acc[c] * inv;
        lse[(size_t)b * N + i] = mx + logf(l);
    }
}

template <int C8, int C2>
__global__ __launch_bounds__(256) void nonlocal_bwd_q_k(const float* __restrict__ theta, const float* __restrict__ phi,
                                                        const float* __restrict__ g, const float* __restrict__ o,
                                                        const float* __restrict__ lse, const float* __restrict__ go,
                                                        float* __restrict__ dtheta, float* __restrict__ dsum, int N, int Nk) {
    __shared__ float sk[NL_TILE * (C8 + C2)];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const bool qv = i < N;
    const float* ph = phi + (size_t)b * C8 * Nk;
    const float* gg = g + (size_t)b * C2 * Nk;
    float t[C8], d_o[C2], dt[C8], Dsum = 0.f;
#pragma unroll
    for (int k = 0; k < C8; ++k) { t[k] = qv ? theta[((size_t)b * C8 + k) * N + i] : 0.f; dt[k] = 0.f; }
#pragma unroll
    for (int c = 0; c < C2; ++c) {
        d_o[c] = qv ? go[((size_t)b * C2 + c) * N + i] : 0.f;
        Dsum += d_o[c] * (qv ? o[((size_t)b * C2 + c) * N + i] : 0.f);
    }
    const float ls = qv ? lse[(size_t)b * N + i] : 0.f;
    for (int j0 = 0; j0 < Nk; j0 += NL_TILE) {
        const int nj = min(NL_TILE, Nk - j0);
        __syncthreads();
        for (int e = threadIdx.x; e < nj * C8; e += 256) { const int k = e / nj, j = e - k * nj; sk[j * (C8 + C2) + k] = ph[(size_t)k * Nk + j0 + j]; }
        for (int e = threadIdx.x; e < nj * C2; e += 256) { const int c = e / nj, j = e - c * nj; sk[j * (C8 + C2) + C8 + c] = gg[(size_t)c * Nk + j0 + j]; }
        __syncthreads();
        for (int j = 0; j < nj; ++j) {
            const float* r = &sk[j * (C8 + C2)];
            float sc = 0.f, db = 0.f;
#pragma unroll
            for (int k = 0; k < C8; ++k) sc += t[k] * r[k];
#pragma unroll
            for (int c = 0; c < C2; ++c) db += d_o[c] * r[C8 + c];
            const float ds = expf(sc - ls) * (db - Dsum);
#pragma unroll
            for (int k = 0; k < C8; ++k) dt[k] += ds * r[k];
        }
    }
    if (qv) {
#pragma unroll
        for (int k = 0; k < C8; ++k) dtheta[((size_t)b * C8 + k) * N + i] = dt[k];
        dsum[(size_t)b * N + i] = Dsum;
    }
}

template <int C8, int C2>
__global__ __launch_bounds__(256) void nonlocal_bwd_k_k(const float* __restrict__ theta, const float* __restrict__ phi,
                                                        const float* __restrict__ g, const float* __restrict__ lse,
                                                        const float* __restrict__ go, const float* __restrict__ dsum,
                                                        float* __restrict__ dphi, float* __restrict__ dg, int N, int Nk) {
    // A workgroup owns NL_KPW = 64 keys of one sample; its four waves split the queries of every staged tile (query i of the
    // tile goes to wave i % 4) and their partial sums are added in wave order at the end (fixed order). One key per THREAD with
    // 256 keys per workgroup left a 32-sample, 256-key block (the generator's 32x32 maps) on 32 workgroups: 240 us.
    constexpr int QW = C8 + C2 + 2;                 // per query: theta, do, lse, D
    constexpr int KPW = 64, CW = C8 + C2;
    static_assert(3 * CW * KPW <= NL_TILE * QW, "reduction buffer aliases the query tile");
    __shared__ float sq[NL_TILE * QW];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int b = blockIdx.y, j = blockIdx.x * KPW + lane;
    const bool kv = j < Nk;
    float p_[C8], g_[C2], dp[C8], dgv[C2];
#pragma unroll
    for (int k = 0; k < C8; ++k) { p_[k] = kv ? phi[((size_t)b * C8 + k) * Nk + j] : 0.f; dp[k] = 0.f; }
#pragma unroll
    for (int c = 0; c < C2; ++c) { g_[c] = kv ? g[((size_t)b * C2 + c) * Nk + j] : 0.f; dgv[c] = 0.f; }
    const float* th = theta + (size_t)b * C8 * N;
    const float* pg = go + (size_t)b * C2 * N;
    for (int i0 = 0; i0 < N; i0 += NL_TILE) {
        const int ni = min(NL_TILE, N - i0);
        __syncthreads();
        for (int e = threadIdx.x; e < ni * C8; e += 256) { const int k = e / ni, i = e - k * ni; sq[i * QW + k] = th[(size_t)k * N + i0 + i]; }
        for (int e = threadIdx.x; e < ni * C2; e += 256) { const int c = e / ni, i = e - c * ni; sq[i * QW + C8 + c] = pg[(size_t)c * N + i0 + i]; }
        for (int i = threadIdx.x; i < ni; i += 256) {
            sq[i * QW + C8 + C2] = lse[(size_t)b * N + i0 + i];
            sq[i * QW + C8 + C2 + 1] = dsum[(size_t)b * N + i0 + i];
        }
        __syncthreads();
        for (int i = sl; i < ni; i += 4) {
            const float* r = &sq[i * QW];
            float sc = 0.f, db = 0.f;
#pragma unroll
            for (int k = 0; k < C8; ++k) sc += r[k] * p_[k];
#pragma unroll
            for (int c = 0; c < C2; ++c) db += r[C8 + c] * g_[c];
            const float p = expf(sc - r[C8 + C2]);
            const float ds = p * (db - r[C8 + C2 + 1]);
#pragma unroll
            for (int k = 0; k < C8; ++k) dp[k] += ds * r[k];
#pragma unroll
            for (int c = 0; c < C2; ++c) dgv[c] += p * r[C8 + c];
        }
    }
    __syncthreads();
    if (sl > 0) {
        float* red = sq + (size_t)(sl - 1) * CW * KPW;
#pragma unroll
        for (int k = 0; k < C8; ++k) red[k * KPW + lane] = dp[k];
#pragma unroll
        for (int c = 0; c < C2; ++c) red[(C8 + c) * KPW + lane] = dgv[c];
    }
    __syncthreads();
    if (sl == 0 && kv) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float* red = sq + (size_t)w * CW * KPW;
#pragma unroll
            for (int k = 0; k < C8; ++k) dp[k] += red[k * KPW + lane];
#pragma unroll
            for (int c = 0; c < C2; ++c) dgv[c] += red[(C8 + c) * KPW + lane];
        }
#pragma unroll
        for (int k = 0; k < C8; ++k) dphi[((size_t)b * C8 + k) * Nk + j] = dp[k];
#pragma unroll
        for (int c = 0; c < C2; ++c) dg[((size_t)b * C2 + c) * Nk + j] = dgv[c];
    }
}

extern "C" int t2v_nonlocal_ok(int C8, int C2) { return (C8 == 4 && C2 == 16) ? 1 : 0; }
extern "C" int t2v_nonlocal_fwd(const float* theta, const float* phi, const float* g, float* o, float* lse, int b, int C8,
                                int C2, int N, int Nk, void* st) {
    if (!theta || !phi || !g || !o || !lse || b < 1 || N < 1 || Nk < 1 || !t2v_nonlocal_ok(C8, C2)) return T2V_EINVAL;
    T2V_LAUNCH((nonlocal_fwd_k<4, 16>), dim3((unsigned)((N + 255) / 256), (unsigned)b), dim3(256), 0, S_(st), theta, phi, g, o, lse, N, Nk);
    return launch_status();
}
extern "C" int t2v_nonlocal_bwd(const float* theta, const float* phi, const float* g, const float* o, const float* lse,
                                const float* go, float* dtheta, float* dphi, float* dg, float* ws, int b, int C8, int C2,
                                int N, int Nk, void* st) {
    if (!theta || !phi || !g || !o || !lse || !go || !dtheta || !dphi || !dg || !ws || b < 1 || N < 1 || Nk < 1 ||
        !t2v_nonlocal_ok(C8, C2)) return T2V_EINVAL;
    T2V_LAUNCH((nonlocal_bwd_q_k<4, 16>), dim3((unsigned)((N + 255) / 256), (unsigned)b), dim3(256), 0, S_(st), theta, phi, g, o, lse, go,
               dtheta, ws, N, Nk);
    int rc = launch_status();
    if (rc) return rc;
    T2V_LAUNCH((nonlocal_bwd_k_k<4, 16>), dim3((unsigned)((Nk + 63) / 64), (unsigned)b), dim3(256), 0, S_(st), theta, phi, g, lse, go, ws,
               dphi, dg, N, Nk);
    return launch_status();
}

// ---------------------------------------------------------------- sentence encoder (models/txt/basic.py:49-70)
// One time step of one (layer, direction) of the packed-sequence LSTM: the recurrent product h_prev[b] . W_hh^T is added to the
// input projection of step t (computed for all steps by one GEMM), the gates (order i, f, g, o) are applied and — pack_padded_
// sequence semantics — only samples with t < length[b] advance; the others keep their state and emit zeros. H <= 256.
// One workgroup per sample, one thread per gate output j = g*H + u: the recurrent product reads the TRANSPOSED weight
// whh_t[k][j] (consecutive threads -> consecutive addresses; the [4H,H] layout made every thread walk its own row) with h_prev[b]
// in LDS; the four pre-activations of a unit meet in LDS for the gate math. 4H <= 1024.
__device__ __forceinline__ float lstm_recurrent_pre(const float* __restrict__ xrow, const float* __restrict__ whh_t,
                                                    const float* __restrict__ hrow, float* sh, int H) {
    const int j = threadIdx.x, H4 = 4 * H;
    for (int k = j; k < H; k += H4) sh[k] = hrow[k];
    __syncthreads();
    float acc = xrow[j];
#pragma unroll 8
    for (int k = 0; k < H; ++k) acc += sh[k] * whh_t[(size_t)k * H4 + j];
    sh[H + j] = acc;
    __syncthreads();
    return acc;
}
__global__ __launch_bounds__(1024) void lstm_seq_step_k(const float* __restrict__ xproj, long xstride, const float* __restrict__ whh_t,
                                                        const float* __restrict__ h_prev, const float* __restrict__ c_prev,
                                                        float* __restrict__ h_next, float* __restrict__ c_next,
                                                        float* __restrict__ out, long ostride, const int32_t* __restrict__ lengths,
                                                        int t, int B, int H) {
    extern __shared__ float sh[];                 // H (h_prev[b]) + 4H (pre-activations)
    const int b = blockIdx.x, u = threadIdx.x;
    lstm_recurrent_pre(xproj + (size_t)b * xstride, whh_t, h_prev + (size_t)b * H, sh, H);
    if (u >= H) return;
    const int i = b * H + u;
    const bool active = t < lengths[b];
    const float gi = sigm(sh[H + u]), gf = sigm(sh[2 * H + u]), gg = tanhf(sh[3 * H + u]), go = sigm(sh[4 * H + u]);
    const float cc = gf * c_prev[i] + gi * gg;
    const float hh = go * tanhf(cc);
    c_next[i] = active ? cc : c_prev[i];
    h_next[i] = active ? hh : sh[u];
    out[(size_t)b * ostride + u] = active ? hh : 0.f;
}
// Both directions of one layer in ONE launch (blockIdx.y = direction): the forward direction's step and the reverse direction's
// step of the same loop iteration are independent (models/txt/basic.py:49-70, nn.LSTM(bidirectional=True)).
struct LstmDirArgs { const float* xproj; const float* whh_t; const float* h_prev; const float* c_prev; float* h_next; float* c_next;
                     float* out; int t; };
__global__ __launch_bounds__(1024) void lstm_seq_step2_k(const LstmDirArgs a0, const LstmDirArgs a1, long xstride, long ostride,
                                                         const int32_t* __restrict__ lengths, int B, int H) {
    extern __shared__ float sh[];
    const LstmDirArgs& a = blockIdx.y ? a1 : a0;
    const int b = blockIdx.x, u = threadIdx.x;
    lstm_recurrent_pre(a.xproj + (size_t)b * xstride, a.whh_t, a.h_prev + (size_t)b * H, sh, H);
    if (u >= H) return;
    const int i = b * H + u;
    const bool active = a.t < lengths[b];
    const float gi = sigm(sh[H + u]), gf = sigm(sh[2 * H + u]), gg = tanhf(sh[3 * H + u]), go = sigm(sh[4 * H + u]);
    const float cc = gf * a.c_prev[i] + gi * gg;
    const float hh = go * tanhf(cc);
    a.c_next[i] = active ? cc : a.c_prev[i];
    a.h_next[i] = active ? hh : sh[u];
    a.out[(size_t)b * ostride + u] = active ? hh : 0.f;
}
// ptrs: 2 x 7 pointers (xproj_t, w_hh_t, h_prev, c_prev, h_next, c_next, out_t) of the forward / reverse direction; ts: their steps
extern "C" int t2v_lstm_seq_step2(const void* const* ptrs, const int32_t* ts, int64_t xstride, int64_t ostride, const int32_t* lengths,
                                  int B, int H, void* st) {
    if (!ptrs || !ts || !lengths || B < 1 || H < 1 || 4 * H > 1024) return T2V_EINVAL;
    LstmDirArgs a[2];
    for (int d = 0; d < 2; ++d) {
        const void* const* q = ptrs + 7 * d;
        for (int k = 0; k < 7; ++k) if (!q[k]) return T2V_EINVAL;
        if (ts[d] < 0 || q[2] == q[4] || q[3] == q[5]) return T2V_EINVAL;
        a[d] = {(const float*)q[0], (const float*)q[1], (const float*)q[2], (const float*)q[3], (float*)q[4], (float*)q[5], (float*)q[6], ts[d]};
    }
    T2V_LAUNCH(lstm_seq_step2_k, dim3((unsigned)B, 2u), dim3((unsigned)(4 * H)), (size_t)5 * H * sizeof(float), S_(st), a[0], a[1],
               (long)xstride, (long)ostride, lengths, B, H);
    return launch_status();
}
extern "C" int t2v_lstm_seq_step(const float* xproj_t, int64_t xstride, const float* w_hh_t, const float* h_prev, const float* c_prev,
                                 float* h_next, float* c_next, float* out_t, int64_t ostride, const int32_t* lengths, int t, int B,
                                 int H, void* st) {
    if (!xproj_t || !w_hh_t || !h_prev || !c_prev || !h_next || !c_next || !out_t || !lengths || B < 1 || H < 1 || 4 * H > 1024 ||
        t < 0 || h_prev == h_next || c_prev == c_next) return T2V_EINVAL;
    T2V_LAUNCH(lstm_seq_step_k, dim3((unsigned)B), dim3((unsigned)(4 * H)), (size_t)5 * H * sizeof(float), S_(st), xproj_t,
               (long)xstride, w_hh_t, h_prev, c_prev, h_next, c_next, out_t, (long)ostride, lengths, t, B, H);
    return launch_status();
}

// ---------------------------------------------------------------- text-encoder PRE-training (txt2vid/train/txt.py:160-178)
// Same step as lstm_seq_step_k with everything the backward needs kept: states live in [B,L,H] buffers indexed by the time
// step they ENTER (h_prev / c_prev / h_next / c_next are (pointer, row stride) pairs, so the last step can write h_n / c_n
// directly) and the post-activation gates of the step are stored (row stride gstride, order i,f,g,o).
__global__ __launch_bounds__(1024) void lstm_train_step_k(const float* __restrict__ xproj, long xs, const float* __restrict__ whh_t,
                                                          const float* __restrict__ hp, long hps, const float* __restrict__ cp, long cps,
                                                          float* __restrict__ hn, long hns, float* __restrict__ cn, long cns,
                                                          float* __restrict__ out, long os, float* __restrict__ gates, long gs,
                                                          const int32_t* __restrict__ lengths, int t, int B, int H) {
    extern __shared__ float sh[];
    const int b = blockIdx.x, u = threadIdx.x;
    lstm_recurrent_pre(xproj + (size_t)b * xs, whh_t, hp + (size_t)b * hps, sh, H);
    if (u >= H) return;
    const bool active = t < lengths[b];
    const float gi = sigm(sh[H + u]), gf = sigm(sh[2 * H + u]), gg = tanhf(sh[3 * H + u]), go = sigm(sh[4 * H + u]);
    const float c0 = cp[(size_t)b * cps + u], h0 = sh[u];
    const float cc = gf * c0 + gi * gg;
    const float hh = go * tanhf(cc);
    float* gr = gates + (size_t)b * gs + u;
    gr[0] = gi; gr[H] = gf; gr[2 * H] = gg; gr[3 * H] = go;
    cn[(size_t)b * cns + u] = active ? cc : c0;
    hn[(size_t)b * hns + u] = active ? hh : h0;
    out[(size_t)b * os + u] = active ? hh : 0.f;
}
extern "C" int t2v_lstm_train_step(const float* xproj_t, int64_t xstride, const float* w_hh_t, const float* h_prev, int64_t hp_stride,
                                   const float* c_prev, int64_t cp_stride, float* h_next, int64_t hn_stride, float* c_next,
                                   int64_t cn_stride, float* out_t, int64_t ostride, float* gates_t, int64_t gstride,
                                   const int32_t* lengths, int t, int B, int H, void* st) {
    if (!xproj_t || !w_hh_t || !h_prev || !c_prev || !h_next || !c_next || !out_t || !gates_t || !lengths || B < 1 || H < 1 ||
        4 * H > 1024 || t < 0 || h_prev == h_next || c_prev == c_next) return T2V_EINVAL;
    T2V_LAUNCH(lstm_train_step_k, dim3((unsigned)B), dim3((unsigned)(4 * H)), (size_t)5 * H * sizeof(float), S_(st), xproj_t,
               (long)xstride, w_hh_t, h_prev, (long)hp_stride, c_prev, (long)cp_stride, h_next, (long)hn_stride, c_next,
               (long)cn_stride, out_t, (long)ostride, gates_t, (long)gstride, lengths, t, B, H);
    return launch_status();
}

// Adjoint of one step, walking the steps backwards. DH / DC [B,H] carry dL/dh and dL/dc of the state LEAVING the step.
//   prologue: DH <- dL/d(state leaving this step) = (first ? DH as given (dL/dh_n)
//                    : the later step was active for b ? dG_later[b] . W_hh (column u) : DH unchanged (state was carried))
//   then (unless epilogue_only, which just finishes DH for the initial state):
//   active: dh = DH + dout_t[b]; gate adjoints -> dG_t[b] (pre-activation, i,f,g,o); DC <- dc * f
//   carried sample: dG_t[b] = 0, DC unchanged.
// DH is read and written by the same thread only (in place).
__global__ __launch_bounds__(256) void lstm_train_step_bwd_k(const float* __restrict__ dout, long os, const float* __restrict__ dG_later,
                                                             long gls, int t_later, const float* __restrict__ whh, float* __restrict__ DH,
                                                             float* __restrict__ DC, const float* __restrict__ gates, long gs,
                                                             const float* __restrict__ c_in, long cis, const float* __restrict__ c_out,
                                                             long cos_, float* __restrict__ dG, long dgs,
                                                             const int32_t* __restrict__ lengths, int t, int B, int H, int first,
                                                             int epilogue_only) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, u = i - b * H;
    const int len = lengths[b];
    float dh_state = DH[i];
    if (!first && t_later < len) {
        const float* gl = dG_later + (size_t)b * gls;
        float a = 0.f;
        for (int j = 0; j < 4 * H; ++j) a += gl[j] * whh[(size_t)j * H + u];
        dh_state = a;
        DH[i] = a;
    }
    if (epilogue_only) return;
    float* dg = dG + (size_t)b * dgs + u;
    if (t >= len) { dg[0] = 0.f; dg[H] = 0.f; dg[2 * H] = 0.f; dg[3 * H] = 0.f; return; }
    const float* gr = gates + (size_t)b * gs + u;
    const float gi = gr[0], gf = gr[H], gg = gr[2 * H], go = gr[3 * H];
    const float dh = dh_state + (dout ? dout[(size_t)b * os + u] : 0.f);
    const float tc = tanhf(c_out[(size_t)b * cos_ + u]);
    const float dc = DC[i] + dh * go * (1.f - tc * tc);
    dg[0] = dc * gg * gi * (1.f - gi);
    dg[H] = dc * c_in[(size_t)b * cis + u] * gf * (1.f - gf);
    dg[2 * H] = dc * gi * (1.f - gg * gg);
    dg[3 * H] = dh * tc * go * (1.f - go);
    DC[i] = dc * gf;
}
extern "C" int t2v_lstm_train_step_bwd(const float* dout_t, int64_t ostride, const float* dG_later, int64_t gl_stride, int t_later,
                                       const float* w_hh, float* DH, float* DC, const float* gates_t, int64_t gstride,
                                       const float* c_in, int64_t ci_stride, const float* c_out, int64_t co_stride, float* dG_t,
                                       int64_t dg_stride, const int32_t* lengths, int t, int B, int H, int first, int epilogue_only,
                                       void* st) {
    if (!w_hh || !DH || !DC || !lengths || B < 1 || H < 1 || (!first && !dG_later)) return T2V_EINVAL;
    if (!epilogue_only && (!gates_t || !c_in || !c_out || !dG_t || t < 0)) return T2V_EINVAL;
    T2V_LAUNCH(lstm_train_step_bwd_k, dim3((unsigned)((B * H + 255) / 256)), dim3(256), 0, S_(st), dout_t, (long)ostride, dG_later,
               (long)gl_stride, t_later, w_hh, DH, DC, gates_t, (long)gstride, c_in, (long)ci_stride, c_out, (long)co_stride, dG_t,
               (long)dg_stride, lengths, t, B, H, first, epilogue_only);
    return launch_status();
}

// dW[tok[n]] += g[n] for n = 0..N-1 in order (no atomics: a thread owns column e of the vocabulary rows v with v % P == p and
// walks the tokens sequentially, so repeated tokens add up in a fixed order). dW must be zero-filled (or hold the value to add to).
__global__ __launch_bounds__(256) void embedding_bwd_k(const float* __restrict__ g, const int32_t* __restrict__ tok, float* __restrict__ dW,
                                                       long N, int E, int V, int P) {
    const int p = blockIdx.y;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < E; e += gridDim.x * 256)
        for (long n = 0; n < N; ++n) {
            const int v = tok[n];
            if (v % P == p && (unsigned)v < (unsigned)V) dW[(size_t)v * E + e] += g[n * E + e];
        }
}
extern "C" int t2v_embedding_bwd(const float* g, const int32_t* tokens, float* dW, int64_t N, int E, int V, void* st) {
    if (!g || !tokens || !dW || N < 1 || E < 1 || V < 1) return T2V_EINVAL;
    const int P = V < 16 ? V : 16;
    T2V_LAUNCH(embedding_bwd_k, dim3((unsigned)((E + 255) / 256), (unsigned)P), dim3(256), 0, S_(st), g, tokens, dW, (long)N, E, V, P);
    return launch_status();
}

// Cross entropy rows (nn.CrossEntropyLoss, train/txt.py:158,172): one workgroup per row.
//   forward: lse[n] = log sum_v exp(x[n][v]); loss[n] = lse[n] - x[n][target[n]]
//   backward: dx[n][v] = gl[n] * (exp(x[n][v] - lse[n]) - [v == target[n]])
__global__ __launch_bounds__(256) void xent_fwd_k(const float* __restrict__ x, const int32_t* __restrict__ target, float* __restrict__ loss,
                                                  float* __restrict__ lse, int V) {
    __shared__ float red[4];
    __shared__ float bc;
    const float* row = x + (size_t)blockIdx.x * V;
    float m = -3.4e38f;
    for (int v = threadIdx.x; v < V; v += 256) m = fmaxf(m, row[v]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) bc = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    m = bc;
    float acc = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) acc += expf(row[v] - m);
    __syncthreads();
    const float ssum = block_sum(acc, red);
    if (threadIdx.x == 0) {
        const float l = m + logf(ssum);
        lse[blockIdx.x] = l;
        loss[blockIdx.x] = l - row[target[blockIdx.x]];
    }
}
__global__ __launch_bounds__(256) void xent_bwd_k(const float* __restrict__ x, const int32_t* __restrict__ target, const float* __restrict__ lse,
                                                  const float* __restrict__ gl, float* __restrict__ dx, int V) {
    const size_t base = (size_t)blockIdx.x * V;
    const float l = lse[blockIdx.x], g = gl[blockIdx.x];
    const int tg = target[blockIdx.x];
    for (int v = threadIdx.x; v < V; v += 256) dx[base + v] = g * (expf(x[base + v] - l) - (v == tg ? 1.f : 0.f));
}
extern "C" int t2v_xent_fwd(const float* logits, const int32_t* target, float* loss_rows, float* lse, int64_t rows, int V, void* st) {
    if (!logits || !target || !loss_rows || !lse || rows < 1 || V < 1) return T2V_EINVAL;
    T2V_LAUNCH(xent_fwd_k, dim3((unsigned)rows), dim3(256), 0, S_(st), logits, target, loss_rows, lse, V);
    return launch_status();
}
extern "C" int t2v_xent_bwd(const float* logits, const int32_t* target, const float* lse, const float* gloss_rows, float* dlogits,
                            int64_t rows, int V, void* st) {
    if (!logits || !target || !lse || !gloss_rows || !dlogits || rows < 1 || V < 1) return T2V_EINVAL;
    T2V_LAUNCH(xent_bwd_k, dim3((unsigned)rows), dim3(256), 0, S_(st), logits, target, lse, gloss_rows, dlogits, V);
    return launch_status();
}
// greedy decoding (txt/basic.py:86: outputs.max(1)): first index of the row maximum
__global__ __launch_bounds__(256) void argmax_rows_k(const float* __restrict__ x, int32_t* __restrict__ idx, int V) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const float* row = x + (size_t)blockIdx.x * V;
    float m = -3.4e38f;
    int mi = 0x7fffffff;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float f = row[v];
        if (f > m) { m = f; mi = v; }
    }
    sv[threadIdx.x] = m; si[threadIdx.x] = mi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const float a = sv[threadIdx.x + o];
            const int ai = si[threadIdx.x + o];
            if (a > sv[threadIdx.x] || (a == sv[threadIdx.x] && ai < si[threadIdx.x])) { sv[threadIdx.x] = a; si[threadIdx.x] = ai; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) idx[blockIdx.x] = si[0] == 0x7fffffff ? 0 : si[0];
}
extern "C" int t2v_argmax_rows(const float* x, int32_t* idx, int64_t rows, int V, void* st) {
    if (!x || !idx || rows < 1 || V < 1) return T2V_EINVAL;
    T2V_LAUNCH(argmax_rows_k, dim3((unsigned)rows), dim3(256), 0, S_(st), x, idx, V);
    return launch_status();
}

// ---------------------------------------------------------------- multi-job launches (non-local block over pyramid levels)
// The non-local block runs the same tiny op on every pyramid level (and on the real||fake and x-hat members of a level):
// up to 8 differently shaped jobs share ONE launch; the job descriptors travel in the kernel arguments and a workgroup
// finds its job in a prefix table of per-job workgroup counts.
#define MJ_MAX 8
struct MultiBatch { t2v_multi_job j[MJ_MAX]; int begin[MJ_MAX + 1]; int n; };
__device__ __forceinline__ int mj_find(const MultiBatch& tb) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < MJ_MAX; ++k)
        if (k < tb.n && (int)blockIdx.x >= tb.begin[k]) ji = k;
    return ji;
}
#ifndef MJ_CHUNK
#define MJ_CHUNK 256       // elements per workgroup of the element-wise ops (256 vs 1024: -0.06 ms per iteration)
#endif

// out = s * a (+ b)          a, b, out: n floats; s = scalar[0]
__global__ __launch_bounds__(256) void mj_scale_k(const MultiBatch tb, const float* __restrict__ scalar) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const float f = scalar[0];
    const float* a = (const float*)q.a; const float* b = (const float*)q.b; float* o = (float*)q.out;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < q.n; i += 256) o[i] = b ? f * a[i] + b[i] : f * a[i];
}
// partial[block] = sum over this workgroup's elements of a*b; mj_dot_final sums the partials in block order
__global__ __launch_bounds__(256) void mj_dot_partial_k(const MultiBatch tb, float* __restrict__ partial) {
    __shared__ float red[4];
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const float* a = (const float*)q.a; const float* b = (const float*)q.b;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * (MJ_CHUNK * 4);
    float acc = 0.f;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK * 4 && i < q.n; i += 256) acc += a[i] * b[i];
    const float v = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void mj_dot_final_k(const float* __restrict__ partial, float* __restrict__ out, int nb, int accum) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
    const float v = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = accum ? out[0] + v : v;
}
// out = (b > 0) ? a : 0      (ReLU adjoint; a = g, b = x)
__global__ __launch_bounds__(256) void mj_relu_mask_k(const MultiBatch tb) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const float* g = (const float*)q.a; const float* x = (const float*)q.b; float* o = (float*)q.out;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < q.n; i += 256) o[i] = x[i] > 0.f ? g[i] : 0.f;
}
// the discriminator step's inputs of one pyramid level in one pass over the real and the generated clips:
// out = [a; b] (torch.cat along the batch) and, when out2 != NULL, out2 = alpha * a + (1 - alpha) * b with alpha = c[row]
// (the gradient penalty's interpolates, losses.py:146; same expression as lerp_rows_k). n = elements of a, d0 = elements per row
__global__ __launch_bounds__(256) void mj_catlerp_k(const MultiBatch tb) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const float* a = (const float*)q.a; const float* b = (const float*)q.b; const float* al = (const float*)q.c;
    float* o = (float*)q.out; float* o2 = (float*)q.out2;
    const long S = q.d0 > 0 ? q.d0 : 1;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < q.n; i += 256) {
        const float rv = a[i], fv = b[i];
        o[i] = rv;
        o[q.n + i] = fv;
        if (o2) { const float w = al[i / S]; o2[i] = w * rv + (1.f - w) * fv; }
    }
}
// column glue of the conditional heads over several members at once (resnet3d.py:53 torch.cat((features, cond), 1) per level):
// mode 0  out[r] = [a[r, 0:d0], b[r, 0:d1]]                       (n = rows)
// mode 1  out[r, 0:d2] = a[r, d1 : d1 + d2]        of a [rows, d0] (slice;  adjoint of mode 0 / of mode 2)
// mode 2  out[r, :] = 0 except out[r, d1 : d1 + d2] = a[r, 0:d2]   out is [rows, d0] (embed; adjoint of mode 1)
__global__ __launch_bounds__(256) void mj_cols_k(const MultiBatch tb, const int mode) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const float* a = (const float*)q.a; const float* b = (const float*)q.b; float* o = (float*)q.out;
    const long wo = mode == 0 ? (long)q.d0 + q.d1 : (mode == 1 ? (long)q.d2 : (long)q.d0);
    const long total = q.n * wo;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < total; i += 256) {
        const long r = i / wo, c = i - r * wo;
        float v;
        if (mode == 0) v = c < q.d0 ? a[r * q.d0 + c] : b[r * q.d1 + (c - q.d0)];
        else if (mode == 1) v = a[r * q.d0 + q.d1 + c];
        else v = (c >= q.d1 && c < q.d1 + q.d2) ? a[r * q.d2 + (c - q.d1)] : 0.f;
        o[i] = v;
    }
}
// out = a + b (+ c): the gradient sums of a grouped fork (one launch for all members instead of one ATen add per member)
__global__ __launch_bounds__(256) void mj_add_k(const MultiBatch tb) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const float* a = (const float*)q.a; const float* b = (const float*)q.b; const float* c = (const float*)q.c;
    float* o = (float*)q.out;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < q.n; i += 256) o[i] = c ? (a[i] + b[i]) + c[i] : a[i] + b[i];
}
// row sums (one wave per row): out[row] = sum_s a[row][s], n = rows, d0 = S;  mode 1: broadcast out[row][s] = a[row]
__global__ __launch_bounds__(256) void mj_rowsum_k(const MultiBatch tb, const int bcast) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const long S = q.d0;
    if (!bcast) {
        const long row = (long)((int)blockIdx.x - tb.begin[ji]) * 4 + (threadIdx.x >> 6);
        if (row >= q.n) return;
        const float* p = (const float*)q.a + row * S;
        float acc = 0.f;
        for (long i = threadIdx.x & 63; i < S; i += 64) acc += p[i];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) ((float*)q.out)[row] = acc;
    } else {
        const long n = q.n * S;
        const float* g = (const float*)q.a; float* o = (float*)q.out;
        const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
        for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < n; i += 256) o[i] = g[i / S];
    }
}
// 2x2 max-pool over the trailing (d0 = H, d1 = W) plane: a = x, out = y, out2 = idx (int32); n = planes
__global__ __launch_bounds__(256) void mj_maxpool_k(const MultiBatch tb) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const int H = q.d0, W = q.d1, Ho = H / 2, Wo = W / 2;
    const long n = q.n * Ho * Wo;
    const float* x = (const float*)q.a; float* y = (float*)q.out; int32_t* idx = (int32_t*)q.out2;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < n; i += 256) {
        int wo = i % Wo; long r = i / Wo;
        int ho = r % Ho; long pl = r / Ho;
        const float* px = x + pl * (long)H * W;
        int best = (2 * ho) * W + 2 * wo;
        float bv = px[best];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                int id = (2 * ho + a) * W + 2 * wo + b;
                float v = px[id];
                if (v > bv || (v != v && bv == bv)) { bv = v; best = id; }
            }
        y[i] = bv;
        idx[i] = best;
    }
}
// scatter: a = g [planes,Ho,Wo], b = idx, out = gx [planes,H,W]
__global__ __launch_bounds__(256) void mj_maxscatter_k(const MultiBatch tb) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const int H = q.d0, W = q.d1, Ho = H / 2, Wo = W / 2;
    const long n = q.n * H * W;
    const float* g = (const float*)q.a; const int32_t* idx = (const int32_t*)q.b; float* gx = (float*)q.out;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < n; i += 256) {
        int w = i % W; long r = i / W;
        int h = r % H; long pl = r / H;
        int ho = h >> 1, wo = w >> 1;
        float v = 0.f;
        if (ho < Ho && wo < Wo) {
            long o = (pl * Ho + ho) * Wo + wo;
            if (idx[o] == h * W + w) v = g[o];
        }
        gx[i] = v;
    }
}
// gather: a = x [planes,H,W], b = idx, out = y [planes,Ho,Wo]
__global__ __launch_bounds__(256) void mj_maxgather_k(const MultiBatch tb) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const int H = q.d0, W = q.d1, Ho = H / 2, Wo = W / 2;
    const long n = q.n * Ho * Wo;
    const float* x = (const float*)q.a; const int32_t* idx = (const int32_t*)q.b; float* y = (float*)q.out;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < n; i += 256) {
        long pl = i / ((long)Ho * Wo);
        y[i] = x[pl * (long)H * W + idx[i]];
    }
}
// row softmax family (one wave per row, d0 = row length, n = rows). mode 0: out = softmax(a); 1: out = y*(gy - sum gy*y)
// with a = y, b = gy; 2: out = gg*(gy - s) - gy*sum(gg*y) with a = y, b = gy, c = gg
__global__ __launch_bounds__(256) void mj_softmax_k(const MultiBatch tb, const int mode) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const long row = (long)((int)blockIdx.x - tb.begin[ji]) * 4 + (threadIdx.x >> 6);
    if (row >= q.n) return;
    const int n = q.d0, lane = threadIdx.x & 63;
    const float* pa = (const float*)q.a + row * n;
    float* po = (float*)q.out + row * n;
    if (mode == 0) {
        float mx = -INFINITY;
        for (int i = lane; i < n; i += 64) mx = fmaxf(mx, pa[i]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int i = lane; i < n; i += 64) s += expf(pa[i] - mx);
        s = wave_sum(s);
        const float inv = 1.f / s;
        for (int i = lane; i < n; i += 64) po[i] = expf(pa[i] - mx) * inv;
    } else if (mode == 1) {
        const float* pg = (const float*)q.b + row * n;
        float s = 0.f;
        for (int i = lane; i < n; i += 64) s += pg[i] * pa[i];
        s = wave_sum(s);
        for (int i = lane; i < n; i += 64) po[i] = pa[i] * (pg[i] - s);
    } else {
        const float* pg = (const float*)q.b + row * n;
        const float* pq = (const float*)q.c + row * n;
        float s = 0.f, u = 0.f;
        for (int i = lane; i < n; i += 64) { s += pg[i] * pa[i]; u += pq[i] * pa[i]; }
        s = wave_sum(s);
        u = wave_sum(u);
        for (int i = lane; i < n; i += 64) po[i] = pq[i] * (pg[i] - s) - pg[i] * u;
    }
}
// batched GEMM of the non-local block (layers.py:28-33,60-65: theta^T phi, g beta^T and their adjoints): n = batch, d0 = M, d1 = N,
// d2 = K, f0 = ta, f1 = tb; a = A, b = B, out = C (row-major [M][N]). Round 4: on the fp32 matrix cores. A workgroup owns a 64 x 64
// tile of one batch entry (four waves, one 32x32 v_mfma_f32_32x32x2_f32 accumulator each: rows = M in registers, columns = N on
// lanes, so the stores are 128-byte runs of C); K goes through LDS in chunks of 16 as As[k][m] / Bs[k][n], each operand read from
// global memory along ITS contiguous axis (m or n when the transposition flag makes that the inner one, k otherwise) and zero
// filled past the edges — any M, N, K. (The 16 x 16 one-output-per-thread tiles this replaces ran the discriminator's full-
// resolution block at ~10 TFLOP/s: 0.39 ms per forward + backward pass at B = 32.)
#define BMM_P 68      // LDS row pitch (64 + 4: the k-major writes of a k-contiguous operand are 2-way conflicted at worst)
#define BMM_KS_MIN_K 128
// KS: the K-SPLIT form for products with few output tiles and a long reduction (the adjoints: dg = do . beta is 64 x 256 per
// sample over K = 1024): a workgroup owns a 32 x 32 tile, its four waves each reduce a QUARTER of K into their own accumulator
// (staging their own chunks in a wave-private part of the LDS tiles) and the four partial tiles are summed through LDS in a
// fixed order — 4x the waves in flight where the 64 x 64 form would leave half the chip idle walking K at memory latency.
__host__ __device__ static inline bool bmm_use_ks(long M, long N, long K, long batch) {
    return K >= BMM_KS_MIN_K && ((M + 63) / 64) * ((N + 63) / 64) * batch < 1024;
}
template <bool KS>
__device__ __forceinline__ void mj_bmm_body(const t2v_multi_job& q, int lb, float* As, float* Bs) {
    constexpr int TM = KS ? 32 : 64;                  // tile edge
    const int M = q.d0, N = q.d1, K = q.d2, ta = q.f0, tb_ = q.f1;
    const int tn = (N + TM - 1) / TM, tm = (M + TM - 1) / TM;
    const int bx = lb % tn; lb /= tn;
    const int by = lb % tm;
    const int b = lb / tm;
    const float* __restrict__ a = (const float*)q.a + (long)b * M * K;
    const float* __restrict__ bb = (const float*)q.b + (long)b * K * N;
    float* __restrict__ c = (float*)q.out + (long)b * M * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = KS ? 0 : (wave & 1), wn = KS ? 0 : (wave >> 1);
    const int m0 = by * TM, n0 = bx * TM;
    // staging coordinates. 64 x 64 form: the workgroup stages one 16-deep chunk, 4 elements per thread and operand — inner-
    // contiguous operand (A stored [K][M], B stored [K][N]): lane = inner index, k = tid / 64 + 4 j; k-contiguous operand
    // (A [M][K], B [N][K]): k = tid % 16, row = tid / 16 + 16 j. KS form: every WAVE stages its own chunk of its own K range
    // (32 rows x 16 k = 8 elements per lane and operand): inner-contiguous: lane & 31 = inner index, k = lane / 32 + 2 j;
    // k-contiguous: k = lane % 16, row = lane / 16 + 4 j.
    const int t_ = KS ? lane : tid;
    const int ik = KS ? (t_ >> 5) : (t_ >> 6), ii = KS ? (t_ & 31) : (t_ & 63);
    const int kk = t_ & 15, kr = t_ >> 4;
    constexpr int NJ = KS ? 8 : 4, KSTEP = KS ? 2 : 4, RSTEP = KS ? 4 : 16;
    // K range of this wave (KS) / of the workgroup
    const int kq = KS ? (((K + 3) / 4 + 15) / 16) * 16 : K;
    const int kbeg = KS ? wave * kq : 0;
    const int kend = KS ? (kbeg + kq < K ? kbeg + kq : K) : K;
    float* as = As + (KS ? wave * 16 * 36 : 0);
    float* bs = Bs + (KS ? wave * 16 * 36 : 0);
    constexpr int P = KS ? 36 : BMM_P;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float ra[NJ], rb[NJ];
    auto load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (ta) {
                const int k = k0 + ik + KSTEP * j, m = m0 + ii;
                ra[j] = (k < kend && m < M) ? a[(long)k * M + m] : 0.f;
            } else {
                const int k = k0 + kk, m = m0 + kr + RSTEP * j;
                ra[j] = (k < kend && m < M) ? a[(long)m * K + k] : 0.f;
            }
            if (!tb_) {
                const int k = k0 + ik + KSTEP * j, n = n0 + ii;
                rb[j] = (k < kend && n < N) ? bb[(long)k * N + n] : 0.f;
            } else {
                const int k = k0 + kk, n = n0 + kr + RSTEP * j;
                rb[j] = (k < kend && n < N) ? bb[(long)n * K + k] : 0.f;
            }
        }
    };
    const int nch = KS ? kq / 16 : (K + 15) / 16;     // (KS: the same trip count for every wave — the loop holds workgroup barriers)
    load(kbeg);
    for (int ch = 0; ch < nch; ++ch) {
        const int k0 = kbeg + ch * 16;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (ta) as[(ik + KSTEP * j) * P + ii] = ra[j]; else as[kk * P + kr + RSTEP * j] = ra[j];
            if (!tb_) bs[(ik + KSTEP * j) * P + ii] = rb[j]; else bs[kk * P + kr + RSTEP * j] = rb[j];
        }
        __syncthreads();
        if (ch + 1 < nch) load(k0 + 16);                         // (in flight during the MFMAs)
        const float* pa = as + hi * P + wm * 32 + l31;
        const float* pb = bs + hi * P + wn * 32 + l31;
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k2 * 2 * P], pb[k2 * 2 * P], acc, 0, 0, 0);
        __syncthreads();
    }
    if constexpr (KS) {
        // the four partial 32 x 32 tiles -> LDS [wave][reg][lane]; thread (wave w, lane l) then owns registers 4w .. 4w+3 of lane l
        float* red = As;                                         // 4 x 16 x 64 floats = 16 KB (As + Bs hold 2 x 4 x 16 x 36 = 18 KB)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
        __syncthreads();
        const int n = n0 + l31;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = wave * 4 + rr;
            float v = red[(0 * 16 + r) * 64 + lane];
            v += red[(1 * 16 + r) * 64 + lane];
            v += red[(2 * 16 + r) * 64 + lane];
            v += red[(3 * 16 + r) * 64 + lane];
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (m < M && n < N) c[(long)m * N + n] = v;
        }
    } else {
        const int n = n0 + wn * 32 + l31;
        if (n < N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < M) c[(long)m * N + n] = acc[r];
            }
        }
    }
}
__global__ __launch_bounds__(256) void mj_bmm_k(const MultiBatch tb) {
    constexpr int HALF = 4 * 16 * 36 > 16 * BMM_P ? 4 * 16 * 36 : 16 * BMM_P;
    __shared__ float lds[2 * HALF];                              // ONE object: the K-split reduce reuses both halves as one buffer
    float* As = lds;
    float* Bs = lds + HALF;
    static_assert(2 * HALF >= 4 * 16 * 64, "the partial tiles fit the staging tiles");
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const int lb = (int)blockIdx.x - tb.begin[ji];
    if (bmm_use_ks(q.d0, q.d1, q.d2, q.n)) mj_bmm_body<true>(q, lb, As, Bs);        // (uniform per job)
    else mj_bmm_body<false>(q, lb, As, Bs);
}
extern "C" int64_t t2v_multi_ws_floats(int op, const t2v_multi_job* jobs, int njobs) {
    if (!jobs || njobs < 1 || njobs > MJ_MAX) return T2V_EINVAL;
    if (op != T2V_MJ_DOT) return 0;
    long blocks = 0;
    for (int i = 0; i < njobs; ++i) blocks += (jobs[i].n + MJ_CHUNK * 4 - 1) / (MJ_CHUNK * 4);
    return blocks;
}
extern "C" int t2v_multi(int op, const t2v_multi_job* jobs, int njobs, const float* scalar, float* ws, void* st) {
    if (!jobs || njobs < 1 || njobs > MJ_MAX) return T2V_EINVAL;
    MultiBatch tb;
    long blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const t2v_multi_job& q = jobs[i];
        if (q.n < 1 || !q.a) return T2V_EINVAL;
        long nb = 0;
        switch (op) {
            case T2V_MJ_SCALE: case T2V_MJ_SCALE_ADD:
                if (!q.out || !scalar || (op == T2V_MJ_SCALE_ADD && !q.b)) return T2V_EINVAL;
                nb = (q.n + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_DOT:
                if (!q.b || !q.out || !ws) return T2V_EINVAL;
                nb = (q.n + MJ_CHUNK * 4 - 1) / (MJ_CHUNK * 4); break;
            case T2V_MJ_MAXPOOL:
                if (!q.out || !q.out2 || q.d0 < 2 || q.d1 < 2) return T2V_EINVAL;
                nb = (q.n * (q.d0 / 2) * (q.d1 / 2) + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_MAXSCATTER:
                if (!q.b || !q.out || q.d0 < 2 || q.d1 < 2) return T2V_EINVAL;
                nb = (q.n * q.d0 * q.d1 + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_MAXGATHER:
                if (!q.b || !q.out || q.d0 < 2 || q.d1 < 2) return T2V_EINVAL;
                nb = (q.n * (q.d0 / 2) * (q.d1 / 2) + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_SOFTMAX: case T2V_MJ_SOFTMAX_BWD: case T2V_MJ_SOFTMAX_BWD_BWD_Y:
                if (!q.out || q.d0 < 1 || (op != T2V_MJ_SOFTMAX && !q.b) || (op == T2V_MJ_SOFTMAX_BWD_BWD_Y && !q.c)) return T2V_EINVAL;
                nb = (q.n + 3) / 4; break;
            case T2V_MJ_RELU_MASK:
                if (!q.b || !q.out) return T2V_EINVAL;
                nb = (q.n + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_ADD:
                if (!q.b || !q.out) return T2V_EINVAL;
                nb = (q.n + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_CATLERP:
                if (!q.b || !q.out || (q.out2 && (!q.c || q.d0 < 1))) return T2V_EINVAL;
                nb = (q.n + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_CATCOLS:
                if (!q.b || !q.out || q.d0 < 1 || q.d1 < 1) return T2V_EINVAL;
                nb = (q.n * ((long)q.d0 + q.d1) + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_SLICECOLS:
                if (!q.out || q.d0 < 1 || q.d1 < 0 || q.d2 < 1 || q.d1 + q.d2 > q.d0) return T2V_EINVAL;
                nb = (q.n * (long)q.d2 + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_EMBEDCOLS:
                if (!q.out || q.d0 < 1 || q.d1 < 0 || q.d2 < 1 || q.d1 + q.d2 > q.d0) return T2V_EINVAL;
                nb = (q.n * (long)q.d0 + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_ROWSUM:
                if (!q.out || q.d0 < 1) return T2V_EINVAL;
                nb = (q.n + 3) / 4; break;
            case T2V_MJ_ROWBCAST:
                if (!q.out || q.d0 < 1) return T2V_EINVAL;
                nb = (q.n * q.d0 + MJ_CHUNK - 1) / MJ_CHUNK; break;
            case T2V_MJ_BMM:
                if (!q.b || !q.out || q.d0 < 1 || q.d1 < 1 || q.d2 < 1) return T2V_EINVAL;
                nb = bmm_use_ks(q.d0, q.d1, q.d2, q.n) ? (long)((q.d1 + 31) / 32) * ((q.d0 + 31) / 32) * q.n
                                                        : (long)((q.d1 + 63) / 64) * ((q.d0 + 63) / 64) * q.n;
                break;
            default: return T2V_EINVAL;
        }
        tb.j[i] = q;
        tb.begin[i] = (int)blocks;
        blocks += nb;
        if (blocks > 0x7fffffffL) return T2V_EINVAL;
    }
    for (int i = njobs; i <= MJ_MAX; ++i) tb.begin[i] = (int)blocks;
    for (int i = njobs; i < MJ_MAX; ++i) tb.j[i] = tb.j[0];
    tb.n = njobs;
    const dim3 grid((unsigned)blocks), blk(256);
    switch (op) {
        case T2V_MJ_SCALE: case T2V_MJ_SCALE_ADD:
            if (op == T2V_MJ_SCALE) for (int i = 0; i < MJ_MAX; ++i) tb.j[i].b = nullptr;
            T2V_LAUNCH(mj_scale_k, grid, blk, 0, S_(st), tb, scalar); break;
        case T2V_MJ_DOT:
            T2V_LAUNCH(mj_dot_partial_k, grid, blk, 0, S_(st), tb, ws);
            T2V_LAUNCH(mj_dot_final_k, dim3(1), blk, 0, S_(st), ws, (float*)jobs[0].out, (int)blocks, jobs[0].f0); break;
        case T2V_MJ_MAXPOOL: T2V_LAUNCH(mj_maxpool_k, grid, blk, 0, S_(st), tb); break;
        case T2V_MJ_MAXSCATTER: T2V_LAUNCH(mj_maxscatter_k, grid, blk, 0, S_(st), tb); break;
        case T2V_MJ_MAXGATHER: T2V_LAUNCH(mj_maxgather_k, grid, blk, 0, S_(st), tb); break;
        case T2V_MJ_SOFTMAX: T2V_LAUNCH(mj_softmax_k, grid, blk, 0, S_(st), tb, 0); break;
        case T2V_MJ_SOFTMAX_BWD: T2V_LAUNCH(mj_softmax_k, grid, blk, 0, S_(st), tb, 1); break;
        case T2V_MJ_SOFTMAX_BWD_BWD_Y: T2V_LAUNCH(mj_softmax_k, grid, blk, 0, S_(st), tb, 2); break;
        case T2V_MJ_BMM: T2V_LAUNCH(mj_bmm_k, grid, blk, 0, S_(st), tb); break;
        case T2V_MJ_RELU_MASK: T2V_LAUNCH(mj_relu_mask_k, grid, blk, 0, S_(st), tb); break;
        case T2V_MJ_ADD: T2V_LAUNCH(mj_add_k, grid, blk, 0, S_(st), tb); break;
        case T2V_MJ_CATLERP: T2V_LAUNCH(mj_catlerp_k, grid, blk, 0, S_(st), tb); break;
        case T2V_MJ_CATCOLS: T2V_LAUNCH(mj_cols_k, grid, blk, 0, S_(st), tb, 0); break;
        case T2V_MJ_SLICECOLS: T2V_LAUNCH(mj_cols_k, grid, blk, 0, S_(st), tb, 1); break;
        case T2V_MJ_EMBEDCOLS: T2V_LAUNCH(mj_cols_k, grid, blk, 0, S_(st), tb, 2); break;
        case T2V_MJ_ROWSUM: T2V_LAUNCH(mj_rowsum_k, grid, blk, 0, S_(st), tb, 0); break;
        case T2V_MJ_ROWBCAST: T2V_LAUNCH(mj_rowsum_k, grid, blk, 0, S_(st), tb, 1); break;
    }
    return launch_status();
}

// ---------------------------------------------------------------- losses
__device__ __forceinline__ float softplus(float v) { return fmaxf(v, 0.f) + log1pf(expf(-fabsf(v))); }

__global__ __launch_bounds__(256) void rsgan_k(const float* a, const float* b, float* loss, int n) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += softplus(-(a[i] - b[i]));   // BCEWithLogits(a-b, 1)
    float s = block_sum(acc, red);
    if (threadIdx.x == 0) loss[0] = s / (float)n;
}
__global__ void rsgan_bwd_k(const float* a, const float* b, const float* gl, float* ga, float* gb, int n) {
    const float sc = gl[0] / (float)n;
    GRID_STRIDE(i, n) {
        const float d = a[i] - b[i];
        const float g = -sc / (1.f + expf(d));          // d/dd softplus(-d) = -sigmoid(-d)
        if (ga) ga[i] = g;
        if (gb) gb[i] = -g;
    }
}
extern "C" int t2v_rsgan(const float* a, const float* b, float* loss, int n, void* st) {
    if (!a || !b || !loss || n < 1) return T2V_EINVAL;
    T2V_LAUNCH(rsgan_k, dim3(1), dim3(256), 0, S_(st), a, b, loss, n);
    return launch_status();
}
extern "C" int t2v_rsgan_bwd(const float* a, const float* b, const float* gl, float* ga, float* gb, int n, void* st) {
    if (!a || !b || !gl || n < 1) return T2V_EINVAL;
    T2V_LAUNCH(rsgan_bwd_k, dim3(nblocks(n)), dim3(256), 0, S_(st), a, b, gl, ga, gb, n);
    return launch_status();
}

// The mean over the pyramid levels of the relativistic loss (cond_gan.py:121-154: loss per level, then the mean) in one launch
// and one for its gradient: same arithmetic, in the same order, as rsgan_k per level + scalar_combine_k with weights 1/L.
__global__ __launch_bounds__(256) void rsgan_mean_multi_k(const MultiBatch tb, float* loss) {
    __shared__ float red[4];
    const float w = 1.0f / (float)tb.n;
    float total = 0.f;
    for (int l = 0; l < tb.n; ++l) {
        const float* a = (const float*)tb.j[l].a; const float* b = (const float*)tb.j[l].b;
        const int n = (int)tb.j[l].n;
        float acc = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) acc += softplus(-(a[i] - b[i]));
        const float s = block_sum(acc, red);
        total += w * (s / (float)n);
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = total;
}
__global__ __launch_bounds__(256) void rsgan_mean_multi_bwd_k(const MultiBatch tb, const float* gl) {
    const int ji = mj_find(tb);
    const t2v_multi_job& q = tb.j[ji];
    const float* a = (const float*)q.a; const float* b = (const float*)q.b;
    float* ga = (float*)q.out; float* gb = (float*)q.out2;
    const float gw = gl[0] * (1.0f / (float)tb.n);          // the mean's share of dL (ScalarCombine's adjoint)
    const float sc = gw / (float)q.n;
    const long base = (long)((int)blockIdx.x - tb.begin[ji]) * MJ_CHUNK;
    for (long i = base + threadIdx.x; i < base + MJ_CHUNK && i < q.n; i += 256) {
        const float d = a[i] - b[i];
        const float g = -sc / (1.f + expf(d));
        if (ga) ga[i] = g;
        if (gb) gb[i] = -g;
    }
}
static int rsgan_multi_table(const t2v_multi_job* jobs, int njobs, MultiBatch& tb, long& blocks) {
    if (!jobs || njobs < 1 || njobs > MJ_MAX) return T2V_EINVAL;
    blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        if (!jobs[i].a || !jobs[i].b || jobs[i].n < 1 || jobs[i].n > 0x7fffffffL) return T2V_EINVAL;
        tb.j[i] = jobs[i];
        tb.begin[i] = (int)blocks;
        blocks += (jobs[i].n + MJ_CHUNK - 1) / MJ_CHUNK;
    }
    for (int i = njobs; i <= MJ_MAX; ++i) tb.begin[i] = (int)blocks;
    for (int i = njobs; i < MJ_MAX; ++i) tb.j[i] = tb.j[0];
    tb.n = njobs;
    return T2V_OK;
}
extern "C" int t2v_rsgan_mean_multi(const t2v_multi_job* jobs, int njobs, float* loss, void* st) {
    MultiBatch tb;
    long blocks;
    if (!loss || rsgan_multi_table(jobs, njobs, tb, blocks)) return T2V_EINVAL;
    T2V_LAUNCH(rsgan_mean_multi_k, dim3(1), dim3(256), 0, S_(st), tb, loss);
    return launch_status();
}
// jobs[i].out / out2: gradients w.r.t. a / b (either may be NULL)
extern "C" int t2v_rsgan_mean_multi_bwd(const t2v_multi_job* jobs, int njobs, const float* gl, void* st) {
    MultiBatch tb;
    long blocks;
    if (!gl || rsgan_multi_table(jobs, njobs, tb, blocks)) return T2V_EINVAL;
    T2V_LAUNCH(rsgan_mean_multi_bwd_k, dim3((unsigned)blocks), dim3(256), 0, S_(st), tb, gl);
    return launch_status();
}

// ---- the loss zoo (gan/losses.py:19-133) on D's logits. One block: n is batch x heads (<= a few thousand).
// Every loss has the form  w * [ mean_i p1(u_i) + mean_i p2(v_i) ]  with  u = r - cr*mean(f),  v = f - cf*mean(r)
// (cr = cf = 0 for the non-averaged kinds).  kind: 0 RSGAN (handled by rsgan_k), 1 vanilla BCE, 2 hinge, 3 Wasserstein,
// 4 RaSGAN, 5 RaLSGAN;  side 0 = discriminator loss, 1 = generator loss.
struct LossTerm { int fn; float sign, shift; };      // fn: 0 none, 1 softplus(sign*t), 2 sign*t, 3 relu(shift - t), 4 (t + shift)^2
struct LossSpec { LossTerm r, f; float rel, w; };
__host__ __device__ inline LossSpec loss_spec(int kind, int side, float margin) {
    LossSpec s = {{0, 0.f, 0.f}, {0, 0.f, 0.f}, 0.f, 1.f};
    if (kind == 1) {            // labels as the reference wires them (losses.py:27-28): fake -> 1, real -> 0
        if (side == 0) { s.r = {1, 1.f, 0.f}; s.f = {1, -1.f, 0.f}; } else { s.f = {1, 1.f, 0.f}; }
    } else if (kind == 2) {     // HingeEmbedding(margin): label 1 -> t, label -1 -> relu(margin - t); fake -> 1, real -> -1
        if (side == 0) { s.r = {3, 0.f, margin}; s.f = {2, 1.f, 0.f}; } else { s.f = {3, 0.f, margin}; }
    } else if (kind == 3) {
        if (side == 0) { s.r = {2, -1.f, 0.f}; s.f = {2, 1.f, 0.f}; } else { s.f = {2, -1.f, 0.f}; }
    } else if (kind == 4) {
        s.rel = 1.f; s.w = 0.5f;
        if (side == 0) { s.r = {1, -1.f, 0.f}; s.f = {1, 1.f, 0.f}; } else { s.r = {1, 1.f, 0.f}; s.f = {1, -1.f, 0.f}; }
    } else if (kind == 5) {
        s.rel = 1.f; s.w = 0.5f;
        if (side == 0) { s.r = {4, 0.f, -1.f}; s.f = {4, 0.f, 1.f}; } else { s.r = {4, 0.f, 1.f}; s.f = {4, 0.f, -1.f}; }
    }
    return s;
}
__device__ __forceinline__ float loss_val(const LossTerm& t, float x) {
    switch (t.fn) {
        case 1: return softplus(t.sign * x);
        case 2: return t.sign * x;
        case 3: return fmaxf(t.shift - x, 0.f);
        case 4: return (x + t.shift) * (x + t.shift);
        default: return 0.f;
    }
}
__device__ __forceinline__ float loss_der(const LossTerm& t, float x) {
    switch (t.fn) {
        case 1: return t.sign / (1.f + expf(-t.sign * x));
        case 2: return t.sign;
        case 3: return (t.shift - x) > 0.f ? -1.f : 0.f;
        case 4: return 2.f * (x + t.shift);
        default: return 0.f;
    }
}
__global__ __launch_bounds__(256) void gan_loss_k(const float* r, const float* f, float* loss, int nr, int nf, LossSpec sp) {
    __shared__ float red[4];
    float mr = 0.f, mf = 0.f;
    if (sp.rel != 0.f) {
        float a = 0.f, b = 0.f;
        for (int i = threadIdx.x; i < nr; i += 256) a += r[i];
        for (int i = threadIdx.x; i < nf; i += 256) b += f[i];
        mr = block_sum(a, red) / (float)nr;
        mf = block_sum(b, red) / (float)nf;
    }
    float a = 0.f, b = 0.f;
    if (sp.r.fn) for (int i = threadIdx.x; i < nr; i += 256) a += loss_val(sp.r, r[i] - sp.rel * mf);
    if (sp.f.fn) for (int i = threadIdx.x; i < nf; i += 256) b += loss_val(sp.f, f[i] - sp.rel * mr);
    const float sa = block_sum(a, red), sb = block_sum(b, red);
    if (threadIdx.x == 0) loss[0] = sp.w * ((sp.r.fn ? sa / (float)nr : 0.f) + (sp.f.fn ? sb / (float)nf : 0.f));
}
__global__ __launch_bounds__(256) void gan_loss_bwd_k(const float* r, const float* f, const float* gl, float* gr, float* gf,
                                                      int nr, int nf, LossSpec sp) {
    __shared__ float red[4];
    const float g = gl[0] * sp.w;
    float mr = 0.f, mf = 0.f, dr = 0.f, df = 0.f;        // means of the logits and of the terms' derivatives
    if (sp.rel != 0.f) {
        float a = 0.f, b = 0.f;
        for (int i = threadIdx.x; i < nr; i += 256) a += r[i];
        for (int i = threadIdx.x; i < nf; i += 256) b += f[i];
        mr = block_sum(a, red) / (float)nr;
        mf = block_sum(b, red) / (float)nf;
        a = 0.f; b = 0.f;
        if (sp.r.fn) for (int i = threadIdx.x; i < nr; i += 256) a += loss_der(sp.r, r[i] - mf);
        if (sp.f.fn) for (int i = threadIdx.x; i < nf; i += 256) b += loss_der(sp.f, f[i] - mr);
        dr = block_sum(a, red) / (float)nr;
        df = block_sum(b, red) / (float)nf;
    }
    // d/dr_i = p1'(u_i)/nr - rel * mean_j p2'(v_j) / nr ;   d/df_i symmetric
    if (gr) for (int i = threadIdx.x; i < nr; i += 256)
        gr[i] = g * ((sp.r.fn ? loss_der(sp.r, r[i] - sp.rel * mf) : 0.f) - sp.rel * df) / (float)nr;
    if (gf) for (int i = threadIdx.x; i < nf; i += 256)
        gf[i] = g * ((sp.f.fn ? loss_der(sp.f, f[i] - sp.rel * mr) : 0.f) - sp.rel * dr) / (float)nf;
}
static bool gan_loss_args_ok(const void* r, const void* f, int nr, int nf, int kind, int side) {
    if (kind < 1 || kind > 5 || (side != 0 && side != 1) || nf < 1 || !f) return false;
    const LossSpec sp = loss_spec(kind, side, 0.f);
    if ((sp.r.fn || sp.rel != 0.f) && (!r || nr < 1)) return false;
    return true;
}
extern "C" int t2v_gan_loss(const float* real, const float* fake, float* loss, int n_real, int n_fake, int kind, int side,
                            float margin, void* st) {
    if (!loss || !gan_loss_args_ok(real, fake, n_real, n_fake, kind, side)) return T2V_EINVAL;
    T2V_LAUNCH(gan_loss_k, dim3(1), dim3(256), 0, S_(st), real, fake, loss, n_real, n_fake, loss_spec(kind, side, margin));
    return launch_status();
}
extern "C" int t2v_gan_loss_bwd(const float* real, const float* fake, const float* gl, float* g_real, float* g_fake, int n_real,
                                int n_fake, int kind, int side, float margin, void* st) {
    if (!gl || !gan_loss_args_ok(real, fake, n_real, n_fake, kind, side)) return T2V_EINVAL;
    T2V_LAUNCH(gan_loss_bwd_k, dim3(1), dim3(256), 0, S_(st), real, fake, gl, g_real, g_fake, n_real, n_fake,
               loss_spec(kind, side, margin));
    return launch_status();
}

__global__ void lerp_rows_k(const float* alpha, const float* xr, const float* xf, float* y, int rows, long S) {
    const long n = (long)rows * S;
    GRID_STRIDE(i, n) {
        const float a = alpha[i / S];
        y[i] = a * xr[i] + (1.f - a) * xf[i];
    }
}
__global__ __launch_bounds__(256) void row_sqnorm_k(const float* g, float* out, int rows, long S) {
    __shared__ float red[4];
    const float* p = g + (long)blockIdx.x * S;
    float acc = 0.f;
    for (long i = threadIdx.x; i < S; i += 256) acc += p[i] * p[i];
    float s = block_sum(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}
__global__ void row_scale_k(const float* s, float mul, const float* g, float* y, int rows, long S) {
    const long n = (long)rows * S;
    GRID_STRIDE(i, n) y[i] = s[i / S] * mul * g[i];
}
extern "C" int t2v_lerp_rows(const float* alpha, const float* xr, const float* xf, float* y, int rows, int64_t S, void* st) {
    if (!alpha || !xr || !xf || !y || rows < 1 || S < 1) return T2V_EINVAL;
    T2V_LAUNCH(lerp_rows_k, dim3(nblocks((long)rows * S)), dim3(256), 0, S_(st), alpha, xr, xf, y, rows, (long)S);
    return launch_status();
}
extern "C" int t2v_row_sqnorm(const float* g, float* out, int rows, int64_t S, void* st) {
    if (!g || !out || rows < 1 || S < 1) return T2V_EINVAL;
    T2V_LAUNCH(row_sqnorm_k, dim3(rows), dim3(256), 0, S_(st), g, out, rows, (long)S);
    return launch_status();
}
extern "C" int t2v_row_scale(const float* s, float mul, const float* g, float* y, int rows, int64_t S, void* st) {
    if (!s || !g || !y || rows < 1 || S < 1) return T2V_EINVAL;
    T2V_LAUNCH(row_scale_k, dim3(nblocks((long)rows * S)), dim3(256), 0, S_(st), s, mul, g, y, rows, (long)S);
    return launch_status();
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam semantics)
__global__ void adam_k(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                       float bc1, float bc2, float gscale, const float* step_dev) {
    if (step_dev) {                      // device-resident step state (HIP-graph replay): {step, bc1, bc2}
        bc1 = step_dev[1];
        bc2 = step_dev[2];
    }
    const float step = lr / bc1;
    const float isq = 1.f / sqrtf(bc2);
    GRID_STRIDE(i, n) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;          // exp_avg.lerp_(grad, 1-b1)
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;     // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * isq + eps;             // (sqrt(v)/sqrt(bc2)).add_(eps)
        p[i] = p[i] - step * (mi / denom);                     // addcdiv_(m, denom, -lr/bc1)
    }
}
extern "C" int t2v_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                        float bc1, float bc2, float gscale, const float* step_dev, void* st) {
    if (!p || !g || !m || !v || n < 1) return T2V_EINVAL;
    T2V_LAUNCH(adam_k, dim3(nblocks(n)), dim3(256), 0, S_(st), p, g, m, v, (long)n, lr, b1, b2, eps, bc1, bc2, gscale, step_dev);
    return launch_status();
}
// Multi-tensor form: up to 64 parameter tensors per launch, their pointers passed BY VALUE in the kernel arguments (no
// device-side table to keep in sync, and a captured HIP graph keeps the pointers in its kernel node). Block b works on
// elements [4096 c, 4096 (c+1)) of tensor j, where begin[j] <= b < begin[j+1] and c = b - begin[j].
#define ADAM_MT 64
#ifndef ADAM_CHUNK
#define ADAM_CHUNK 4096
#endif
struct AdamBatch {
    float* p[ADAM_MT]; const float* g[ADAM_MT]; float* m[ADAM_MT]; float* v[ADAM_MT];
    int n[ADAM_MT]; int begin[ADAM_MT + 1]; int njobs;
};
__global__ __launch_bounds__(256) void adam_multi_k(AdamBatch tb, float lr, float b1, float b2, float eps, float bc1, float bc2,
                                                     float gscale, const float* step_dev) {
    if (step_dev) { bc1 = step_dev[1]; bc2 = step_dev[2]; }
    const float step = lr / bc1;
    const float isq = 1.f / sqrtf(bc2);
    int lo = 0, hi = tb.njobs;                         // largest j with begin[j] <= blockIdx.x
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tb.begin[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
    const int j = lo;
    const long base = (long)((int)blockIdx.x - tb.begin[j]) * ADAM_CHUNK;
    const int len = (int)min((long)ADAM_CHUNK, (long)tb.n[j] - base);
    float* __restrict__ p = tb.p[j] + base;
    const float* __restrict__ g = tb.g[j] + base;
    float* __restrict__ m = tb.m[j] + base;
    float* __restrict__ v = tb.v[j] + base;
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    const int nv = vec ? (len >> 2) : 0;
    for (int q = threadIdx.x; q < nv; q += 256) {
        float4 P = ((float4*)p)[q], M = ((float4*)m)[q], V = ((float4*)v)[q];
        const float4 G = ((const float4*)g)[q];
        float* pp = (float*)&P; float* mm = (float*)&M; float* vv = (float*)&V; const float* gg = (const float*)&G;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gi = gg[e] * gscale;
            const float mi = b1 * mm[e] + (1.f - b1) * gi;
            const float vi = b2 * vv[e] + (1.f - b2) * gi * gi;
            mm[e] = mi; vv[e] = vi;
            pp[e] = pp[e] - step * (mi / (sqrtf(vi) * isq + eps));
        }
        ((float4*)p)[q] = P; ((float4*)m)[q] = M; ((float4*)v)[q] = V;
    }
    for (int i = nv * 4 + threadIdx.x; i < len; i += 256) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step * (mi / (sqrtf(vi) * isq + eps));
    }
}
extern "C" int t2v_adam_multi(const t2v_adam_job* jobs, int njobs, float lr, float b1, float b2, float eps, float bc1, float bc2,
                              float gscale, const float* step_dev, void* st) {
    if (!jobs || njobs < 1) return T2V_EINVAL;
    for (int i = 0; i < njobs; ++i)
        if (!jobs[i].p || !jobs[i].g || !jobs[i].m || !jobs[i].v || jobs[i].n < 1 || jobs[i].n > 0x7fffffffL) return T2V_EINVAL;
    for (int at = 0; at < njobs; at += ADAM_MT) {
        AdamBatch tb;
        const int cnt = njobs - at < ADAM_MT ? njobs - at : ADAM_MT;
        long blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const t2v_adam_job& jb = jobs[at + i];
            tb.p[i] = (float*)jb.p; tb.g[i] = (const float*)jb.g; tb.m[i] = (float*)jb.m; tb.v[i] = (float*)jb.v;
            tb.n[i] = (int)jb.n; tb.begin[i] = (int)blocks;
            blocks += (jb.n + ADAM_CHUNK - 1) / ADAM_CHUNK;
        }
        for (int i = cnt; i < ADAM_MT; ++i) { tb.p[i] = nullptr; tb.g[i] = nullptr; tb.m[i] = nullptr; tb.v[i] = nullptr; tb.n[i] = 0; tb.begin[i] = (int)blocks; }
        tb.begin[ADAM_MT] = (int)blocks;
        tb.njobs = cnt;
        if (blocks > 0x7fffffffL) return T2V_EINVAL;
        T2V_LAUNCH(adam_multi_k, dim3((unsigned)blocks), dim3(256), 0, S_(st), tb, lr, b1, b2, eps, bc1, bc2, gscale, step_dev);
    }
    return launch_status();
}
// torch.optim.SGD(lr, momentum) semantics (the reference's --sgd branch, train/gan.py:86-89), multi-tensor like adam_multi_k:
// buf = first ? g : momentum * buf + g ;  p -= lr * (momentum != 0 ? buf : g).   jobs[i].m = momentum buffer, .v unused
__global__ __launch_bounds__(256) void sgd_multi_k(AdamBatch tb, float lr, float momentum, float gscale, int first) {
    int lo = 0, hi = tb.njobs;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tb.begin[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
    const int j = lo;
    const long base = (long)((int)blockIdx.x - tb.begin[j]) * ADAM_CHUNK;
    const int len = (int)min((long)ADAM_CHUNK, (long)tb.n[j] - base);
    float* __restrict__ p = tb.p[j] + base;
    const float* __restrict__ g = tb.g[j] + base;
    float* __restrict__ m = tb.m[j] ? tb.m[j] + base : nullptr;
    for (int i = threadIdx.x; i < len; i += 256) {
        const float gi = g[i] * gscale;
        float d = gi;
        if (m) { d = first ? gi : momentum * m[i] + gi; m[i] = d; }
        p[i] = p[i] - lr * d;
    }
}
extern "C" int t2v_sgd_multi(const t2v_adam_job* jobs, int njobs, float lr, float momentum, float gscale, int first_step, void* st) {
    if (!jobs || njobs < 1) return T2V_EINVAL;
    for (int i = 0; i < njobs; ++i)
        if (!jobs[i].p || !jobs[i].g || (momentum != 0.f && !jobs[i].m) || jobs[i].n < 1 || jobs[i].n > 0x7fffffffL) return T2V_EINVAL;
    for (int at = 0; at < njobs; at += ADAM_MT) {
        AdamBatch tb;
        const int cnt = njobs - at < ADAM_MT ? njobs - at : ADAM_MT;
        long blocks = 0;
        for (int i = 0; i < ADAM_MT; ++i) {
            if (i < cnt) {
                const t2v_adam_job& jb = jobs[at + i];
                tb.p[i] = (float*)jb.p; tb.g[i] = (const float*)jb.g; tb.m[i] = momentum != 0.f ? (float*)jb.m : nullptr; tb.v[i] = nullptr;
                tb.n[i] = (int)jb.n; tb.begin[i] = (int)blocks;
                blocks += (jb.n + ADAM_CHUNK - 1) / ADAM_CHUNK;
            } else { tb.p[i] = nullptr; tb.g[i] = nullptr; tb.m[i] = nullptr; tb.v[i] = nullptr; tb.n[i] = 0; tb.begin[i] = (int)blocks; }
        }
        tb.begin[ADAM_MT] = (int)blocks;
        tb.njobs = cnt;
        if (blocks > 0x7fffffffL) return T2V_EINVAL;
        T2V_LAUNCH(sgd_multi_k, dim3((unsigned)blocks), dim3(256), 0, S_(st), tb, lr, momentum, gscale, first_step);
    }
    return launch_status();
}
// step state {step, bc1, bc2}: step += 1 and the bias corrections 1 - beta^step, evaluated in double and
// rounded to float exactly like the host path does (python float -> c_float).
__global__ void adam_tick_k(float* state, float b1, float b2) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float t = state[0] + 1.f;
        state[0] = t;
        state[1] = (float)(1.0 - pow((double)b1, (double)t));
        state[2] = (float)(1.0 - pow((double)b2, (double)t));
    }
}
extern "C" int t2v_adam_tick(float* state, float b1, float b2, void* st) {
    if (!state) return T2V_EINVAL;
    T2V_LAUNCH(adam_tick_k, dim3(1), dim3(64), 0, S_(st), state, b1, b2);
    return launch_status();
}

// ---------------------------------------------------------------- pyramid gather
__global__ void pyramid_gather_k(const float* x, float* y, int B, int C, int T, int H, int W, int Bo, int To, int Ho, int Wo,
                                 int sb, int stt, int bt, const int32_t* bt_dev) {
    if (bt_dev) bt = bt_dev[0];
    const long n = (long)Bo * C * To * Ho * Wo;
    GRID_STRIDE(i, n) {
        int wo = i % Wo; long r = i / Wo;
        int ho = r % Ho; r /= Ho;
        int to = r % To; r /= To;
        int c = r % C; int bo = r / C;
        // nearest: src = floor(dst * in / out)  (F.interpolate default mode)
        const int h = (int)(((long)ho * H) / Ho), w = (int)(((long)wo * W) / Wo);
        y[i] = x[((((long)bo * sb) * C + c) * T + (to * stt + bt)) * (long)H * W + (long)h * W + w];
    }
}
extern "C" int t2v_pyramid_gather(const float* x, float* y, int B, int C, int T, int H, int W, int Bo, int To, int Ho, int Wo,
                                  int sb, int stt, int bt, const int32_t* bt_dev, void* st) {
    if (!x || !y || B < 1 || C < 1 || T < 1 || Bo < 1 || To < 1 || Ho < 1 || Wo < 1) return T2V_EINVAL;
    // with a device-resident phase the caller guarantees 0 <= *bt_dev < stt and (To-1)*stt + stt-1 < T
    const int btmax = bt_dev ? stt - 1 : bt;
    if ((long)(Bo - 1) * sb >= B || (long)(To - 1) * stt + btmax >= T) return T2V_EINVAL;
    T2V_LAUNCH(pyramid_gather_k, dim3(nblocks((long)Bo * C * To * Ho * Wo)), dim3(256), 0, S_(st), x, y, B, C, T, H, W, Bo, To, Ho, Wo, sb, stt, bt, bt_dev);
    return launch_status();
}


// ---------------------------------------------------------------- layout glue (dense strided copies)
__global__ void copy2d_k(const float* src, long src_ld, float* dst, long dst_ld, long rows, long cols) {
    const long n = rows * cols;
    GRID_STRIDE(i, n) {
        const long r = i / cols, c = i - r * cols;
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}
extern "C" int t2v_copy2d(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int64_t rows, int64_t cols, void* st) {
    if (!src || !dst || rows < 1 || cols < 1 || src_ld < cols || dst_ld < cols) return T2V_EINVAL;
    T2V_LAUNCH(copy2d_k, dim3(nblocks(rows * cols)), dim3(256), 0, S_(st), src, (long)src_ld, dst, (long)dst_ld, (long)rows, (long)cols);
    return launch_status();
}
// fp32 <-> bf16 streams for the opt-in bf16 gradient exchange (txt2vid_amd.dist, SURVEY §8(e) "optionally reduce in bf16"):
// round-to-nearest-even through the hardware convert (a NaN stays a NaN); 8 elements per thread where the count allows.
__global__ void cast_f32_bf16_k(const float* __restrict__ src, __bf16* __restrict__ dst, long n) {
    const long n8 = n >> 3;
    GRID_STRIDE(i, n8) {
        const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
        typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
        bf8 o;
        o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
        o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
        reinterpret_cast<bf8*>(dst)[i] = o;
    }
    GRID_STRIDE(j, n - (n8 << 3)) dst[(n8 << 3) + j] = (__bf16)src[(n8 << 3) + j];
}
__global__ void cast_bf16_f32_k(const __bf16* __restrict__ src, float* __restrict__ dst, long n) {
    const long n8 = n >> 3;
    GRID_STRIDE(i, n8) {
        typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
        const bf8 v = reinterpret_cast<const bf8*>(src)[i];
        reinterpret_cast<float4*>(dst)[2 * i] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
        reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
    }
    GRID_STRIDE(j, n - (n8 << 3)) dst[(n8 << 3) + j] = (float)src[(n8 << 3) + j];
}
// dir 0: bf16 dst <- fp32 src; dir 1: fp32 dst <- bf16 src. Both pointers 16-byte aligned (32 for the fp32 side).
extern "C" int t2v_cast_bf16(const void* src, void* dst, int64_t n, int dir, void* st) {
    if (!src || !dst || n < 1 || (dir != 0 && dir != 1)) return T2V_EINVAL;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return T2V_EINVAL;
    long nb = nblocks((n + 7) / 8);
    if (dir == 0) T2V_LAUNCH(cast_f32_bf16_k, dim3(nb), dim3(256), 0, S_(st), (const float*)src, (__bf16*)dst, (long)n);
    else T2V_LAUNCH(cast_bf16_f32_k, dim3(nb), dim3(256), 0, S_(st), (const __bf16*)src, (float*)dst, (long)n);
    return launch_status();
}
// x[A][B][inner] -> y[B][A][inner]
__global__ void permute01_k(const float* x, float* y, long A, long B, long inner) {
    const long n = A * B * inner;
    GRID_STRIDE(i, n) {
        const long in = i % inner; long r = i / inner;
        const long a = r % A, b = r / A;
        y[i] = x[(a * B + b) * inner + in];
    }
}
extern "C" int t2v_permute01(const float* x, float* y, int64_t A, int64_t B, int64_t inner, void* st) {
    if (!x || !y || A < 1 || B < 1 || inner < 1) return T2V_EINVAL;
    T2V_LAUNCH(permute01_k, dim3(nblocks(A * B * inner)), dim3(256), 0, S_(st), x, y, (long)A, (long)B, (long)inner);
    return launch_status();
}
// x[A][B][C][inner] -> y[A][C][B][inner]
__global__ void permute12_k(const float* x, float* y, long A, long B, long Cc, long inner) {
    const long n = A * B * Cc * inner;
    GRID_STRIDE(i, n) {
        const long in = i % inner; long r = i / inner;
        const long b = r % B; r /= B;
        const long c = r % Cc, a = r / Cc;
        y[i] = x[((a * B + b) * Cc + c) * inner + in];
    }
}
extern "C" int t2v_permute12(const float* x, float* y, int64_t A, int64_t B, int64_t Cc, int64_t inner, void* st) {
    if (!x || !y || A < 1 || B < 1 || Cc < 1 || inner < 1) return T2V_EINVAL;
    T2V_LAUNCH(permute12_k, dim3(nblocks(A * B * Cc * inner)), dim3(256), 0, S_(st), x, y, (long)A, (long)B, (long)Cc, (long)inner);
    return launch_status();
}
// merged-frames layout [b*T][inner]: keep samples ::2, frames bt::2. adjoint=1 writes y back into x.
__global__ void subsample_frames_k(float* x, float* y, long T, long inner, long bo, long To, int bt, int adjoint,
                                   const int32_t* bt_dev) {
    if (bt_dev) bt = bt_dev[0];
    const long n = bo * To * inner;
    GRID_STRIDE(i, n) {
        const long in = i % inner; long r = i / inner;
        const long to = r % To, b2 = r / To;
        const long src = ((2 * b2) * T + (2 * to + bt)) * inner + in;
        if (adjoint) x[src] = y[i]; else y[i] = x[src];
    }
}
extern "C" int t2v_subsample_frames(float* x, float* y, int64_t b, int64_t T, int64_t inner, int64_t bo, int64_t To, int bt,
                                    int adjoint, const int32_t* bt_dev, void* st) {
    if (!x || !y || b < 1 || T < 1 || inner < 1 || bo < 1 || To < 1 || bt < 0 || bt > 1) return T2V_EINVAL;
    const int btmax = bt_dev ? 1 : bt;
    if (2 * (bo - 1) >= b || 2 * (To - 1) + btmax >= T) return T2V_EINVAL;
    T2V_LAUNCH(subsample_frames_k, dim3(nblocks(bo * To * inner)), dim3(256), 0, S_(st), x, y, (long)T, (long)inner, (long)bo, (long)To, bt, adjoint, bt_dev);
    return launch_status();
}
// adjoint of pyramid_gather without spatial resampling: gx[b*sb, c, t*st+bt, :] = g[b, c, t, :]
__global__ void pyramid_scatter_k(const float* g, float* gx, long Cc, long T, long HW, long Bo, long To, int sb, int stt, int bt) {
    const long n = Bo * Cc * To * HW;
    GRID_STRIDE(i, n) {
        const long in = i % HW; long r = i / HW;
        const long to = r % To; r /= To;
        const long c = r % Cc, bo = r / Cc;
        gx[(((bo * sb) * Cc + c) * T + (to * stt + bt)) * HW + in] = g[i];
    }
}
extern "C" int t2v_pyramid_scatter(const float* g, float* gx, int B, int Cc, int T, int64_t HW, int Bo, int To, int sb, int stt,
                                   int bt, void* st) {
    if (!g || !gx || B < 1 || Cc < 1 || T < 1 || HW < 1 || Bo < 1 || To < 1) return T2V_EINVAL;
    if ((long)(Bo - 1) * sb >= B || (long)(To - 1) * stt + bt >= T) return T2V_EINVAL;
    T2V_LAUNCH(pyramid_scatter_k, dim3(nblocks((long)Bo * Cc * To * HW)), dim3(256), 0, S_(st), g, gx, (long)Cc, (long)T, (long)HW, (long)Bo, (long)To, sb, stt, bt);
    return launch_status();
}


// ---------------------------------------------------------------- scalar / row glue
struct ScalarList { const float* p[16]; float w[16]; int n; };
__global__ void scalar_combine_k(ScalarList l, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float acc = 0.f;
        for (int i = 0; i < l.n; ++i) acc += l.w[i] * l.p[i][0];
        out[0] = acc;
    }
}
extern "C" int t2v_scalar_combine(const void* const* ptrs, const float* weights, int n, float* out, void* st) {
    if (!ptrs || !weights || !out || n < 1 || n > 16) return T2V_EINVAL;
    ScalarList l;
    l.n = n;
    for (int i = 0; i < n; ++i) { l.p[i] = (const float*)ptrs[i]; l.w[i] = weights[i]; if (!l.p[i]) return T2V_EINVAL; }
    T2V_LAUNCH(scalar_combine_k, dim3(1), dim3(64), 0, S_(st), l, out);
    return launch_status();
}
// out[r] = x[perm[r]]  (inverse: out[perm[r]] = x[r])
__global__ void gather_rows_k(const float* x, const int32_t* perm, float* out, long rows, long cols, int inverse) {
    const long n = rows * cols;
    GRID_STRIDE(i, n) {
        const long r = i / cols, c = i - r * cols;
        if (inverse) out[(long)perm[r] * cols + c] = x[i]; else out[i] = x[(long)perm[r] * cols + c];
    }
}
extern "C" int t2v_gather_rows(const float* x, const int32_t* perm, float* out, int64_t rows, int64_t cols, int inverse, void* st) {
    if (!x || !perm || !out || rows < 1 || cols < 1) return T2V_EINVAL;
    T2V_LAUNCH(gather_rows_k, dim3(nblocks(rows * cols)), dim3(256), 0, S_(st), x, perm, out, (long)rows, (long)cols, inverse);
    return launch_status();
}

extern "C" const char* t2v_version(void) { return "t2v_hip 0.1 (gfx950, fp32 MFMA)"; }

// ------------------------------------------------------------------------------------------------
// Pooled convolution, pointwise halves (DESIGN §5 "pooled second convolution"). In every discriminator block the second 3^3
// convolution is consumed only by the average pooling behind it. Box filter and convolution commute, so
//     pool(conv3(r)) = stride-2 conv3 of the BOX-SUMMED activation  r~[p] = scale * sum_{delta in {0,1}^k} r[p - 1 + delta]
// — 27 taps over the POOLED voxels instead of 27 taps over all of them (4x fewer MACs for the stem's (1,2,2) pooling, 8x for
// the DownBlocks' (2,2,2)), at the price of one streaming pass that writes r~. r~ lives on a PADDED grid (index p = position + 1,
// one extra row / column / frame of real border sums), so the strided GEMM kernels in conv.hip gather it without a single bounds
// check. t2v_pool_boxsum writes r~; t2v_pool_unbox is its adjoint applied to the data gradient the GEMM leaves on the padded grid
// (as 8 parity-class planes), with the ReLU mask of the activation fused in.
//   tmode 0: no time axis (D == 1); 1: time is box-summed and strided like H and W; 2: time is strided WITHOUT a box (the stem:
//   AvgPool3d((1,2,2), stride 2) keeps the even frames)
// Layouts: r~ [NC, Dp, H+1, W+2] (Dp = D+1, or 1 for tmode 0; the last column is an alignment pad, written as 0);
//          planes [8][NC, Dq, H/2+1, W/2+1] (Dq = D/2+1, or 1), class = (ct*2 + cy)*2 + cx = parity of the padded index per axis.
// ------------------------------------------------------------------------------------------------
struct PoolBoxBatch { t2v_poolbox_job j[POOL_MT]; int begin[POOL_MT + 1]; int n; };

// q = m / d for small m (< 2^22) via the float reciprocal with a +-1 fix-up
__device__ __forceinline__ uint32_t small_div(uint32_t m, uint32_t d, float rcp) {
    int q = (int)((float)m * rcp);
    const int r = (int)m - q * (int)d;
    q += (r >= (int)d) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return (uint32_t)q;
}
// Both passes work on PAIRS of neighbouring x positions (W and W + 2 are even): one index decode, three column loads per row
// instead of four, and 8-byte stores. A workgroup owns POOL_CHUNK consecutive pairs; its first pair is decoded with integer
// divisions once, every pair inside the chunk with small float-reciprocal divisions of small numbers.
__global__ __launch_bounds__(256) void pool_boxsum_k(const PoolBoxBatch tb) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < POOL_MT; ++k)
        if (k < tb.n && (int)blockIdx.x >= tb.begin[k]) ji = k;
    const t2v_poolbox_job& q = tb.j[ji];
    const int D = q.D, H = q.H, W = q.W, tm = q.tmode;
    const uint32_t Dp = tm ? D + 1 : 1, Hp = H + 1, Wp = W + 2, Wp2 = Wp / 2;
    const uint32_t n2 = (uint32_t)q.NC * Dp * Hp * Wp2;           // pairs (host: elements < 2^31)
    const uint32_t base = (uint32_t)((int)blockIdx.x - tb.begin[ji]) * POOL_CHUNK;
    const uint32_t b_r0 = base / Wp2, b_x = base - b_r0 * Wp2;
    const uint32_t b_r1 = b_r0 / Hp, b_y = b_r0 - b_r1 * Hp;
    const uint32_t b_nc = b_r1 / Dp, b_f = b_r1 - b_nc * Dp;
    const float rW = 1.0f / (float)Wp2, rH = 1.0f / (float)Hp, rD = 1.0f / (float)Dp;
    const float* __restrict__ x = q.in;
    const float* __restrict__ mk = q.mask;
    const float floor_ = q.relu ? 0.f : -__builtin_inff();
    const float scale = q.scale;
    const int nf = tm == 1 ? 2 : 1;
    float2* __restrict__ out2 = reinterpret_cast<float2*>(q.out);
#pragma unroll
    for (int it = 0; it < POOL_CHUNK / 256; ++it) {
        const uint32_t t = it * 256 + threadIdx.x;
        if (base + t >= n2) break;
        const uint32_t ix = b_x + t, q1 = small_div(ix, Wp2, rW), xp = (ix - q1 * Wp2) * 2;
        const uint32_t iy = b_y + q1, q2 = small_div(iy, Hp, rH), yp = iy - q2 * Hp;
        const uint32_t if_ = b_f + q2, q3 = small_div(if_, Dp, rD), fp = if_ - q3 * Dp;
        const uint32_t nc = b_nc + q3;
        float a0 = 0.f, a1 = 0.f;                                  // outputs xp and xp + 1
        const int f0 = tm == 0 ? 0 : (int)fp - 1;
        const bool y0 = yp >= 1, y1 = yp < (uint32_t)H;
        const bool c0 = xp >= 1 && xp <= (uint32_t)W, c1 = xp < (uint32_t)W, c2 = xp + 1 < (uint32_t)W;       // input columns xp-1, xp, xp+1
        for (int a = 0; a < nf; ++a) {
            const int f = f0 + a;
            if ((unsigned)f >= (unsigned)D) continue;
            const uint32_t o = ((nc * D + f) * H + (yp - 1)) * W + (xp - 1);                      // (row yp-1, column xp-1)
            float v00 = 0.f, v01 = 0.f, v02 = 0.f, v10 = 0.f, v11 = 0.f, v12 = 0.f;
            if (y0 && c0) v00 = x[o];
            if (y0 && c1) v01 = x[o + 1];
            if (y0 && c2) v02 = x[o + 2];
            if (y1 && c0) v10 = x[o + W];
            if (y1 && c1) v11 = x[o + W + 1];
            if (y1 && c2) v12 = x[o + W + 2];
            if (mk) {
                if (y0 && c0) v00 = mk[o] > 0.f ? v00 : 0.f;
                if (y0 && c1) v01 = mk[o + 1] > 0.f ? v01 : 0.f;
                if (y0 && c2) v02 = mk[o + 2] > 0.f ? v02 : 0.f;
                if (y1 && c0) v10 = mk[o + W] > 0.f ? v10 : 0.f;
                if (y1 && c1) v11 = mk[o + W + 1] > 0.f ? v11 : 0.f;
                if (y1 && c2) v12 = mk[o + W + 2] > 0.f ? v12 : 0.f;
            }
            v00 = fmaxf(v00, floor_); v01 = fmaxf(v01, floor_); v02 = fmaxf(v02, floor_);
            v10 = fmaxf(v10, floor_); v11 = fmaxf(v11, floor_); v12 = fmaxf(v12, floor_);
            a0 += (v00 + v01) + (v10 + v11);
            a1 += (v01 + v02) + (v11 + v12);
        }
        out2[base + t] = make_float2(a0 * scale, xp + 1 <= (uint32_t)W ? a1 * scale : 0.f);
    }
}

__global__ __launch_bounds__(256) void pool_unbox_k(const PoolBoxBatch tb) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < POOL_MT; ++k)
        if (k < tb.n && (int)blockIdx.x >= tb.begin[k]) ji = k;
    const t2v_poolbox_job& q = tb.j[ji];
    const uint32_t D = q.D, H = q.H, W = q.W, W2 = W / 2;
    const int tm = q.tmode;
    const uint32_t Dq = tm ? D / 2 + 1 : 1, Hq = H / 2 + 1, Wq = W / 2 + 1;
    const uint32_t n2 = (uint32_t)q.NC * D * H * W2;
    const uint32_t plane = (uint32_t)q.NC * Dq * Hq * Wq;          // (host: 8 * sets * plane < 2^31)
    const int sets = q.relu > 1 ? q.relu : 1;                     // plane sets to add up (the data-gradient GEMM's k-split)
    const uint32_t base = (uint32_t)((int)blockIdx.x - tb.begin[ji]) * POOL_CHUNK;
    const uint32_t b_r0 = base / W2, b_x = base - b_r0 * W2;
    const uint32_t b_r1 = b_r0 / H, b_y = b_r0 - b_r1 * H;
    const uint32_t b_nc = b_r1 / D, b_f = b_r1 - b_nc * D;
    const float rW = 1.0f / (float)W2, rH = 1.0f / (float)H, rD = 1.0f / (float)D;
    const float* __restrict__ pl = q.in;
    const float2* __restrict__ mk2 = reinterpret_cast<const float2*>(q.mask);
    float2* __restrict__ out2 = reinterpret_cast<float2*>(q.out);
    const int nf = tm == 1 ? 2 : 1;
#pragma unroll
    for (int it = 0; it < POOL_CHUNK / 256; ++it) {
        const uint32_t t = it * 256 + threadIdx.x;
        if (base + t >= n2) break;
        const uint32_t ix = b_x + t, q1 = small_div(ix, W2, rW), bx = ix - q1 * W2;          // x = 2 bx, 2 bx + 1
        const uint32_t iy = b_y + q1, q2 = small_div(iy, H, rH), y = iy - q2 * H;
        const uint32_t if_ = b_f + q2, q3 = small_div(if_, D, rD), f = if_ - q3 * D;
        const uint32_t nc = b_nc + q3;
        float2 m2 = make_float2(1.f, 1.f);
        if (mk2) m2 = mk2[base + t];
        float a0 = 0.f, a1 = 0.f;
        if (m2.x > 0.f || m2.y > 0.f) {
            for (int a = 0; a < nf; ++a) {
                const uint32_t pt = tm == 0 ? 0u : f + 1 - a;         // padded time index of r~ these voxels contributed to
                const uint32_t ct = pt & 1u, at = pt >> 1;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const uint32_t py = y + 1 - b, cy = py & 1u, ay = py >> 1;
                    // x = 2 bx reads padded columns 2 bx + 1 (class 1, index bx) and 2 bx (class 0, index bx);
                    // x = 2 bx + 1 reads 2 bx + 2 (class 0, index bx + 1) and 2 bx + 1 (class 1, index bx)
                    const uint32_t o0 = ((ct * 2 + cy) * 2 + 0) * plane + ((nc * Dq + at) * Hq + ay) * Wq + bx;
                    const uint32_t o1 = o0 + plane;
                    for (int s_ = 0; s_ < sets; ++s_) {
                        const uint32_t so = (uint32_t)s_ * 8u * plane;
                        const float e0 = pl[o0 + so], e1 = pl[o0 + 1 + so], d1 = pl[o1 + so];
                        a0 += e0 + d1;
                        a1 += e1 + d1;
                    }
                }
            }
        }
        const float bv = q.bias ? q.bias[nc % (uint32_t)q.C] : 0.f;
        out2[base + t] = make_float2(m2.x > 0.f ? a0 * q.scale + bv : 0.f, m2.y > 0.f ? a1 * q.scale + bv : 0.f);
    }
}

static int poolbox_multi(const t2v_poolbox_job* jobs, int njobs, bool unbox, void* st) {
    if (!jobs || njobs < 1) return T2V_EINVAL;
    for (int at = 0; at < njobs; at += POOL_MT) {
        PoolBoxBatch tb;
        const int cnt = njobs - at < POOL_MT ? njobs - at : POOL_MT;
        long blocks = 0;
        for (int i = 0; i < cnt; ++i) {
            const t2v_poolbox_job& q = jobs[at + i];
            if (!q.in || !q.out || q.NC < 1 || q.D < 1 || q.H < 2 || q.W < 2 || (q.H & 1) || (q.W & 1) || q.tmode < 0 || q.tmode > 2) return T2V_EINVAL;
            if ((q.tmode == 0 && q.D != 1) || (q.tmode != 0 && (q.D < 2 || (q.D & 1)))) return T2V_EINVAL;
            if (q.bias && (!unbox || q.C < 1 || (q.NC % q.C) != 0)) return T2V_EINVAL;
            tb.j[i] = q;
            tb.begin[i] = (int)blocks;
            const long n = unbox ? (long)q.NC * q.D * q.H * q.W
                                 : (long)q.NC * (q.tmode ? q.D + 1 : 1) * (q.H + 1) * (q.W + 2);
            const long sets = unbox && q.relu > 1 ? q.relu : 1;
            if (n >= (1L << 31) || (unbox && 8 * sets * (long)q.NC * (q.tmode ? q.D / 2 + 1 : 1) * (q.H / 2 + 1) * (q.W / 2 + 1) >= (1L << 31)))
                return T2V_EINVAL;                                   // 32-bit element indices inside the kernels
            if (((uintptr_t)q.out & 7u) || (unbox && q.mask && ((uintptr_t)q.mask & 7u))) return T2V_EINVAL;      // 8-byte pair accesses
            blocks += (n / 2 + POOL_CHUNK - 1) / POOL_CHUNK;          // a workgroup owns POOL_CHUNK PAIRS of neighbouring x positions
        }
        for (int i = cnt; i <= POOL_MT; ++i) tb.begin[i] = (int)blocks;
        for (int i = cnt; i < POOL_MT; ++i) tb.j[i] = tb.j[0];
        tb.n = cnt;
        if (blocks > 0x7fffffffL) return T2V_EINVAL;
        if (unbox) T2V_LAUNCH(pool_unbox_k, dim3((unsigned)blocks), dim3(256), 0, S_(st), tb);
        else T2V_LAUNCH(pool_boxsum_k, dim3((unsigned)blocks), dim3(256), 0, S_(st), tb);
    }
    return launch_status();
}
extern "C" int t2v_pool_boxsum(const t2v_poolbox_job* jobs, int njobs, void* st) { return poolbox_multi(jobs, njobs, false, st); }
extern "C" int t2v_pool_unbox(const t2v_poolbox_job* jobs, int njobs, void* st) { return poolbox_multi(jobs, njobs, true, st); }

__global__ __launch_bounds__(256) void wgrad_swap_k(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int T, int accum) {
    const long n = (long)Cout * Cin * T;
    GRID_STRIDE(i, n) {
        const int t = (int)(i % T); const long r = i / T;
        const int ci = (int)(r % Cin), co = (int)(r / Cin);
        const float v = src[((long)ci * Cout + co) * T + (T - 1 - t)];
        dst[i] = accum ? dst[i] + v : v;
    }
}
extern "C" int t2v_wgrad_swap(const float* src, float* dst, int Cout, int Cin, int T, int accum, void* st) {
    if (!src || !dst || Cout < 1 || Cin < 1 || T < 1) return T2V_EINVAL;
    T2V_LAUNCH(wgrad_swap_k, dim3(nblocks((long)Cout * Cin * T)), dim3(256), 0, S_(st), src, dst, Cout, Cin, T, accum);
    return launch_status();
}
