// Input row of the hot path: synthetic Moving-MNIST-shaped clips + captions generated straight in HBM.
//
// txt2vid/data/synthetic/generate.py:18-47,136-170 draws one 28x28 blob, a digit label, one of four bounce motions and a fixed
// coordinate per clip, then translates the blob linearly between two edge points and back. The host-side dataset of this build
// (`txt2vid_amd.data.SyntheticMovingDigits`) takes those draws from numpy's legacy RandomState seeded per (seed, index); this
// kernel reproduces that generator bit for bit — MT19937 seeding / twist / tempering, numpy's 53-bit `random_sample`, its
// masked-rejection `randint` — so that a clip generated on the GPU equals the host clip exactly (integer / geometry work; the one
// floating-point step, -1 + 2u in double rounded to float, is the same IEEE operation on both sides).
//
// One workgroup per clip: thread 0 seeds the 624-word state (a serial recurrence), the twist runs in parallel in its four
// dependency phases, every thread tempers and converts, then all 256 lanes write the T x C x S x S frames with 16-byte stores
// (HBM-write bound: 4 bytes per output element, nothing is read).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/t2v_hip.h"

namespace {

constexpr int MT_N = 624, MT_M = 397, MT_BLOCKS = 4, BLOB = 28;

struct SynthVocab { int32_t id[T2V_SYNTH_VOCAB]; };

__device__ inline uint32_t mt_mix(uint32_t cur, uint32_t next, uint32_t far_) {
    const uint32_t y = (cur & 0x80000000u) | (next & 0x7fffffffu);
    return far_ ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__global__ __launch_bounds__(256) void synth_clips_k(const int64_t* __restrict__ index, int64_t seed, int T, int C, int S, SynthVocab voc,
                                                     float* __restrict__ vids, int64_t* __restrict__ tokens, int32_t* __restrict__ err) {
    __shared__ uint32_t mt[MT_N];
    __shared__ uint32_t draws[MT_BLOCKS * MT_N];
    __shared__ float blob[BLOB * BLOB];
    __shared__ int32_t meta[4];                       // digit, motion, fixed coordinate
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {                                   // numpy `_legacy_seeding(int)` == init_genrand
        uint32_t s = (uint32_t)(((uint64_t)seed * 1000003ull + (uint64_t)index[b]) & 0x7FFFFFFFull);
        for (int k = 0; k < MT_N; ++k) {
            mt[k] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)k + 1u;
        }
    }
    __syncthreads();
    for (int blk = 0; blk < MT_BLOCKS; ++blk) {
        // the twist's four dependency phases; inside a phase every element reads values no lane of the phase writes except its
        // own slot and its right neighbour's OLD value: read into registers, barrier, write
        uint32_t v = 0;
        if (tid < MT_N - MT_M) v = mt_mix(mt[tid], mt[tid + 1], mt[tid + MT_M]);                           // [0, 227)
        __syncthreads();
        if (tid < MT_N - MT_M) mt[tid] = v;
        __syncthreads();
        const int i2 = (MT_N - MT_M) + tid;                                                              // [227, 454)
        if (tid < MT_N - MT_M) v = mt_mix(mt[i2], mt[i2 + 1], mt[i2 - (MT_N - MT_M)]);
        __syncthreads();
        if (tid < MT_N - MT_M) mt[i2] = v;
        __syncthreads();
        const int i3 = 2 * (MT_N - MT_M) + tid;                                                          // [454, 623)
        if (i3 < MT_N - 1) v = mt_mix(mt[i3], mt[i3 + 1], mt[i3 - (MT_N - MT_M)]);
        __syncthreads();
        if (i3 < MT_N - 1) mt[i3] = v;
        __syncthreads();
        if (tid == 0) mt[MT_N - 1] = mt_mix(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
        __syncthreads();
        for (int k = tid; k < MT_N; k += 256) {
            uint32_t y = mt[k];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            draws[blk * MT_N + k] = y;
        }
        __syncthreads();
    }
    // rs.uniform(-1, 1, (28, 28)).astype(float32): lower + range * random_sample(), random_sample = (a >> 5, b >> 6) / 2^53
    for (int e = tid; e < BLOB * BLOB; e += 256) {
        const double a = (double)(draws[2 * e] >> 5), c = (double)(draws[2 * e + 1] >> 6);
        const double u = (a * 67108864.0 + c) / 9007199254740992.0;
        blob[e] = (float)(-1.0 + 2.0 * u);
    }
    if (tid == 0) {                                   // rs.randint(10), rs.randint(4), rs.randint(S - 28 + 1): masked rejection on 32-bit draws
        int pos = 2 * BLOB * BLOB;
        const uint32_t rngs[3] = {9u, 3u, (uint32_t)(S - BLOB)};
        bool bad = false;
        for (int r = 0; r < 3; ++r) {
            uint32_t val = 0;
            if (rngs[r] != 0) {
                uint32_t mask = rngs[r];
                mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
                for (;;) {
                    if (pos >= MT_BLOCKS * MT_N) { bad = true; val = 0; break; }     // (> 900 rejections in a row: never; flagged, not hidden)
                    val = draws[pos++] & mask;
                    if (val <= rngs[r]) break;
                }
            }
            meta[r] = (int32_t)val;
        }
        if (bad && err) *err = 1;
        // caption: <start> digit N is A and B <end>   (MOTIONS: left-right, right-left, top-bottom, bottom-top)
        const int mot = meta[1];
        const int a_ = mot == 0 ? 13 : mot == 1 ? 15 : mot == 2 ? 16 : 17;
        const int b_ = mot == 0 ? 15 : mot == 1 ? 13 : mot == 2 ? 17 : 16;
        int64_t* tk = tokens + (int64_t)b * 8;
        tk[0] = voc.id[0]; tk[1] = voc.id[1]; tk[2] = voc.id[2 + meta[0]]; tk[3] = voc.id[12];
        tk[4] = voc.id[a_]; tk[5] = voc.id[14]; tk[6] = voc.id[b_]; tk[7] = voc.id[18];
    }
    __syncthreads();
    const int mot = meta[1], fixed = meta[2], lim = S - BLOB;
    const int64_t frame = (int64_t)S * S;
    float* out = vids + (int64_t)b * T * C * frame;
    const int rowq = S / 4;                            // S % 4 == 0 (host-checked): a float4 never straddles a row
    const int64_t nq = (int64_t)T * C * frame / 4;
    for (int64_t q = tid; q < nq; q += 256) {
        const int x0 = (int)(q % rowq) * 4;
        const int64_t r = q / rowq;
        const int y = (int)(r % S);
        const int t = (int)(r / ((int64_t)S * C));
        // pos = int(round((ph if ph <= 1 else 2 - ph) * lim)), ph = t / float(T - 1) * 2.0 — Python doubles, round half to even
        const double ph = (double)t / (double)(T - 1) * 2.0;
        int p = (int)rint((ph <= 1.0 ? ph : 2.0 - ph) * (double)lim);
        if (mot == 1 || mot == 3) p = lim - p;
        const int by = mot < 2 ? fixed : p, bx = mot < 2 ? p : fixed;
        float4 v4;
        float* pv = reinterpret_cast<float*>(&v4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = x0 + j;
            const bool in = y >= by && y < by + BLOB && x >= bx && x < bx + BLOB;
            pv[j] = in ? blob[(y - by) * BLOB + (x - bx)] : -1.0f;
        }
        reinterpret_cast<float4*>(out)[q] = v4;
    }
}

}  // namespace

extern "C" int t2v_synth_clips(const int64_t* index_dev, int B, int64_t seed, int T, int C, int S, const int32_t* vocab_ids, float* vids,
                               int64_t* tokens, int32_t* err_dev, void* stream) {
    if (!index_dev || !vocab_ids || !vids || !tokens || B < 1 || T < 2 || C < 1 || S < BLOB || (S % 4) != 0) return T2V_EINVAL;
    if ((int64_t)B * T * C * S * S >= (1LL << 40)) return T2V_EINVAL;
    SynthVocab voc;
    for (int i = 0; i < T2V_SYNTH_VOCAB; ++i) voc.id[i] = vocab_ids[i];
    (void)hipGetLastError();
    hipLaunchKernelGGL(synth_clips_k, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, index_dev, seed, T, C, S, voc, vids, tokens, err_dev);
    return hipGetLastError() == hipSuccess ? T2V_OK : T2V_ELAUNCH;
}
