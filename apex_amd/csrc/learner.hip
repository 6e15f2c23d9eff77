// Learner half of the Cassie-v0 PPO hot path as HIP kernels for gfx950 (MI355X).
//
//   returns scan        <- PPOBuffer.finish_path             rl/algos/ppo.py:73-89
//   advantage normalise <- PPO.train                         rl/algos/ppo.py:395-396
//   MLP forward/backward (fp32 MFMA 32x32x2, LDS-tiled)      rl/policies/actor.py:142-215, critic.py:37-77
//   PPO clipped-ratio / value / mirror losses                rl/algos/ppo.py:276-345
//   global-norm clip + Adam                                  rl/algos/ppo.py:322-336, :355-356
//
// Wave = 64 lanes.  GEMMs use v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain) so that the parity mode
// meets the 1e-5 bar; everything elementwise is fused into GEMM epilogues or the single loss kernel.
#include "apx_common.h"
#include <gfx950/dynamic_lds.h>
#include <cmath>

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

template <int NV>
__device__ __forceinline__ void block_atomic_add(double (&v)[NV], double* out) {
    __shared__ double sm[NV][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = wave_sum(v[i]);
        if (lane == 0) sm[i][w] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0;
        for (int k = 0; k < nw; ++k) s += sm[threadIdx.x][k];
        atomicAdd(out + threadIdx.x, s);
    }
}

// ------------------------------------------------------------------------------------------------ returns scan
// One lane per env column; loads of rew/end/boot at a fixed t are contiguous across lanes (coalesced).
// Algorithmic HBM bytes per (t, env): 4 (rew) + 1 (end) + 4 (boot) + 4 (ret) = 13 B.
__global__ void returns_scan_kernel(const float* __restrict__ rew, const uint8_t* __restrict__ end,
                                    const float* __restrict__ boot, const float* __restrict__ last_val, double gamma,
                                    int T, int N, float* __restrict__ ret) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float g32 = (float)gamma;
    double R = (double)last_val[n];
    bool fresh = true;  // R holds an fp32 bootstrap: the reference's first gamma*R is an fp32 product (ppo.py:80-82)
    for (int t = T - 1; t >= 0; --t) {
        const size_t i = (size_t)t * N + n;
        if (end[i]) { R = (double)boot[i]; fresh = true; }
        const double prod = fresh ? (double)(g32 * (float)R) : gamma * R;
        R = prod + (double)rew[i];
        fresh = false;
        ret[i] = (float)R;
    }
}

extern "C" int apx_returns_scan(const float* rew, const uint8_t* end, const float* boot, const float* last_val,
                                double gamma, int T, int N, float* ret, void* stream) {
    APX_REQUIRE(T >= 0 && N >= 0, "T,N");
    if (T == 0 || N == 0) return APX_OK;
    APX_REQUIRE(rew && end && boot && last_val && ret, "null pointer");
    hipLaunchKernelGGL(returns_scan_kernel, dim3(apx_cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, rew, end, boot,
                       last_val, gamma, T, N, ret);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

// ------------------------------------------------------------------------------------------------ advantages
__global__ void adv_moments_kernel(const float* __restrict__ ret, const float* __restrict__ val, int64_t n,
                                   double* __restrict__ mom) {
    double acc[2] = {0, 0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double a = (double)(ret[i] - val[i]);
        acc[0] += a; acc[1] += a * a;
    }
    block_atomic_add<2>(acc, mom);
    if (blockIdx.x == 0 && threadIdx.x == 0) mom[2] = (double)n;
}

__global__ void adv_apply_kernel(const float* __restrict__ ret, const float* __restrict__ val, int64_t n, float mean,
                                 float denom, float* __restrict__ adv) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        adv[i] = ((ret[i] - val[i]) - mean) / denom;
}

// same as adv_apply_kernel with (mean, unbiased std) taken from the moments on the device: no host round trip between the
// moments (possibly all-reduced over ranks) and the normalisation
__global__ void adv_apply_moments_kernel(const float* __restrict__ ret, const float* __restrict__ val, int64_t n,
                                         const double* __restrict__ mom, float eps, float* __restrict__ adv) {
    const double cnt = mom[2], mean_d = mom[0] / cnt;
    double var = (mom[1] - cnt * mean_d * mean_d) / (cnt - 1.0 > 1.0 ? cnt - 1.0 : 1.0);
    var = var > 0.0 ? var : 0.0;
    const float mean = (float)mean_d, denom = (float)sqrt(var) + eps;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        adv[i] = ((ret[i] - val[i]) - mean) / denom;
}

extern "C" int apx_adv_apply_moments(const float* ret, const float* val, int64_t n, const double* moments, double eps, float* adv, void* stream) {
    APX_REQUIRE(ret && val && adv && moments && n > 0, "args");
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(adv_apply_moments_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ret, val, n, moments, (float)eps, adv);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

extern "C" int apx_adv_moments(const float* ret, const float* val, int64_t n, double* moments, void* stream) {
    APX_REQUIRE(ret && val && moments && n > 0, "args");
    APX_HIP(hipMemsetAsync(moments, 0, 3 * sizeof(double), (hipStream_t)stream));
    const int grid = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(adv_moments_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ret, val, n, moments);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

extern "C" int apx_adv_apply(const float* ret, const float* val, int64_t n, double mean, double std_unbiased,
                             double eps, float* adv, void* stream) {
    APX_REQUIRE(ret && val && adv && n > 0, "args");
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(adv_apply_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, ret, val, n, (float)mean,
                       (float)std_unbiased + (float)eps, adv);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

// ------------------------------------------------------------------------------------------------ input prep
// xn[b, c] = normalise( mirror( x[idx[b], :] ) )[c]; one thread per element, rows contiguous.
__global__ void prep_obs_kernel(const float* __restrict__ x, int64_t B, int D, const int64_t* __restrict__ idx,
                                const int32_t* __restrict__ sign_perm, uint64_t clock_mask,
                                const float* __restrict__ mean, const float* __restrict__ stdv,
                                float* __restrict__ out) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= B * D) return;
    const int64_t b = e / D;
    const int c = (int)(e - b * D);
    const int64_t row = idx ? idx[b] : b;
    float v;
    if (sign_perm) {
        const int32_t s = sign_perm[c];
        v = s >= 0 ? x[row * D + s] : -x[row * D + (-s - 1)];
        if (c < 64 && ((clock_mask >> c) & 1ull)) v = sinf(asinf(v) + 3.14159265358979323846f);  // wrappers.py:65-66
    } else {
        v = x[row * D + c];
    }
    if (mean) v = (v - mean[c]) / stdv[c];
    out[e] = v;
}

// ------------------------------------------------------------------------------------------------ fp32 MFMA GEMM
// C[M,N] = epi( sum_k A(m,k) B(k,n) ),  A(m,k) = A[m*a_rs + k*a_cs],  B(k,n) = B[k*b_rs + n*b_cs].
// Block tile 64x64x16, 4 waves, each wave one 32x32 v_mfma_f32_32x32x2_f32 accumulator.
// LDS tiles are k-major ([k][m], [k][n]) so the MFMA operand fetch (lane l: k = l>>5, m|n = l&31) is a
// conflict-free ds_read_b32 (two 32-lane halves, 32 consecutive dwords each).
enum { EPI_STORE = 0, EPI_BIAS = 1, EPI_BIAS_RELU = 2, EPI_MASK = 3, EPI_ATOMIC = 4, EPI_ACC = 5 /* C += AB, no split-K */,
       EPI_PARTIAL = 6 /* split-K, every K chunk STORES its product to its own slab C + z * part_stride (summed by grad_reduce_kernel); the bias gradient stays atomic */ };

struct GemmArgs {
    const float* A; long a_rs, a_cs;
    const float* B; long b_rs, b_cs;
    float* C; long ldc;
    const float* aux; long ld_aux;   // bias[N] or mask[M, ld_aux]
    int M, N, K, kchunk;
    long part_stride;                // EPI_PARTIAL: floats between the slabs of consecutive K chunks
};

#define GBM 64
#define GBN 64
#define GBK 16

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ float As[GBK][GBM + 1];
    __shared__ float Bs[GBK][GBN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    floatx16 acc = {0};
    const bool a_kmajor = (g.a_cs == 1);   // k contiguous in memory
    const bool b_kmajor = (g.b_rs == 1);
    const bool bias_grad = (EPI == EPI_ATOMIC || EPI == EPI_PARTIAL) && g.aux != nullptr && blockIdx.y == 0;
    float bsum = 0.f;
    // software pipeline: the global loads of k-tile t+1 are in flight while the MFMAs of tile t run out of LDS
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m, k;
            if (a_kmajor) { k = tid & 15; m = (tid >> 4) + 16 * i; }
            else          { m = tid & 63; k = (tid >> 6) + 4 * i; }
            const int gm = m0 + m, gk = k0 + k;
            ra[i] = (gm < g.M && gk < kend) ? g.A[gm * g.a_rs + gk * g.a_cs] : 0.f;
            int n, kk;
            if (b_kmajor) { kk = tid & 15; n = (tid >> 4) + 16 * i; }
            else          { n = tid & 63; kk = (tid >> 6) + 4 * i; }
            const int gn = n0 + n, gkk = k0 + kk;
            rb[i] = (gn < g.N && gkk < kend) ? g.B[gkk * g.b_rs + gn * g.b_cs] : 0.f;
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m, k;
            if (a_kmajor) { k = tid & 15; m = (tid >> 4) + 16 * i; }
            else          { m = tid & 63; k = (tid >> 6) + 4 * i; }
            As[k][m] = ra[i];
            int n, kk;
            if (b_kmajor) { kk = tid & 15; n = (tid >> 4) + 16 * i; }
            else          { n = tid & 63; kk = (tid >> 6) + 4 * i; }
            Bs[kk][n] = rb[i];
        }
        __syncthreads();
        if (k0 + GBK < kend) fetch(k0 + GBK);
        if ((EPI == EPI_ATOMIC || EPI == EPI_PARTIAL) && bias_grad && tid < GBM) {      // fused bias gradient: column sums of dY ride on the A tile
#pragma unroll
            for (int kk = 0; kk < GBK; ++kk) bsum += As[kk][tid];
        }
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            const float a = As[kk + (lane >> 5)][wm + (lane & 31)];
            const float b = Bs[kk + (lane >> 5)][wn + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if ((EPI == EPI_ATOMIC || EPI == EPI_PARTIAL) && bias_grad && tid < GBM && m0 + tid < g.M) atomicAdd(const_cast<float*>(g.aux) + m0 + tid, bsum);
    const int col = n0 + wn + (lane & 31);
    if (col >= g.N) return;
    float bias = 0.f;
    if (EPI == EPI_BIAS || EPI == EPI_BIAS_RELU) bias = g.aux[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        float v = acc[r];
        float* c = g.C + (EPI == EPI_PARTIAL ? (long)blockIdx.z * g.part_stride : 0l) + (long)row * g.ldc + col;
        if (EPI == EPI_BIAS) v += bias;
        if (EPI == EPI_BIAS_RELU) v = fmaxf(v + bias, 0.f);
        if (EPI == EPI_MASK) v = g.aux[(long)row * g.ld_aux + col] > 0.f ? v : 0.f;
        if (EPI == EPI_ATOMIC) atomicAdd(c, v);
        else if (EPI == EPI_ACC) *c += v;
        else *c = v;
    }
}

// ------------------------------------------------------------------------------------------------ fp32 MFMA GEMM, 128 x 128 tiles
// The two big shapes of the backward pass - dX = dY W (M = batch, N = K = 256) and dW = dY^T X (M = N = 256, K = batch) - run at half the operand traffic per flop of the
// 64 x 64 kernel: block tile 128 x 128 x 16, 4 waves, each wave a 64 x 64 sub-tile = 2 x 2 accumulators of v_mfma_f32_32x32x2_f32 (four MFMAs per four ds_read_b32), operands
// fetched from HBM / L2 as float4 one k-tile ahead of the MFMAs.  Shapes: M, N multiples of 128, K chunks multiples of 16, B(k, n) contiguous in n (both call sites), A(m, k)
// contiguous in k (dX) or in m (dW); B(k, n) contiguous in n, or in k (Y = X W^T + b: the LSTM's input projection of the upper layers, EPI_BIAS); everything else stays on
// gemm_f32_kernel.  Epilogues: EPI_STORE, EPI_BIAS, EPI_MASK, EPI_PARTIAL (+ the fused bias gradient).
#define G2M 128
#define G2N 128
#define G2P 132      /* LDS row pitch in floats: 16-byte aligned rows for the float4 stores of the m- / n-contiguous operands */
typedef float f4v __attribute__((ext_vector_type(4)));
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_128_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[GBK][G2P];
    __shared__ __attribute__((aligned(16))) float Bs[GBK][G2P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * G2M, n0 = blockIdx.y * G2N;
    const int kbeg = blockIdx.z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = floatx16{0};
    const bool a_kmajor = g.a_cs == 1;
    const bool b_kmajor = g.b_cs != 1;      // B(k, n) contiguous in k (Y = X W^T with W in torch layout: the forward pass); gemm128_ok admits b_cs == 1 or b_rs == 1
    const bool bias_grad = EPI == EPI_PARTIAL && g.aux != nullptr && blockIdx.y == 0;
    float bsum = 0.f;
    f4v ra[2], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (a_kmajor) { const int row = (tid >> 2) + 64 * i, kq = (tid & 3) * 4; ra[i] = *(const f4v*)(g.A + (long)(m0 + row) * g.a_rs + k0 + kq); }
            else          { const int k = (tid >> 5) + 8 * i, mq = (tid & 31) * 4;  ra[i] = *(const f4v*)(g.A + (long)(k0 + k) * g.a_cs + m0 + mq); }
            if (b_kmajor) { const int row = (tid >> 2) + 64 * i, kq = (tid & 3) * 4; rb[i] = *(const f4v*)(g.B + (long)(n0 + row) * g.b_cs + k0 + kq); }
            else          { const int k = (tid >> 5) + 8 * i, nq = (tid & 31) * 4; rb[i] = *(const f4v*)(g.B + (long)(k0 + k) * g.b_rs + n0 + nq); }
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GBK) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (a_kmajor) {
                const int row = (tid >> 2) + 64 * i, kq = (tid & 3) * 4;
                As[kq][row] = ra[i].x; As[kq + 1][row] = ra[i].y; As[kq + 2][row] = ra[i].z; As[kq + 3][row] = ra[i].w;
            } else { const int k = (tid >> 5) + 8 * i, mq = (tid & 31) * 4; *(f4v*)&As[k][mq] = ra[i]; }
            if (b_kmajor) {
                const int row = (tid >> 2) + 64 * i, kq = (tid & 3) * 4;
                Bs[kq][row] = rb[i].x; Bs[kq + 1][row] = rb[i].y; Bs[kq + 2][row] = rb[i].z; Bs[kq + 3][row] = rb[i].w;
            } else { const int k = (tid >> 5) + 8 * i, nq = (tid & 31) * 4; *(f4v*)&Bs[k][nq] = rb[i]; }
        }
        __syncthreads();
#if !defined(APX_GEMM_ABL) || APX_GEMM_ABL < 2
        if (k0 + GBK < kend) fetch(k0 + GBK);
#endif
        if (EPI == EPI_PARTIAL && bias_grad && tid < G2M) {      // fused bias gradient: column sums of dY ride on the A tile
#pragma unroll
            for (int kk = 0; kk < GBK; ++kk) bsum += As[kk][tid];
        }
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2) {
            const int kr = kk + (lane >> 5), c = lane & 31;
            const float a0 = As[kr][wm + c], a1 = As[kr][wm + 32 + c], b0 = Bs[kr][wn + c], b1 = Bs[kr][wn + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    if (EPI == EPI_PARTIAL && bias_grad && tid < G2M) atomicAdd(const_cast<float*>(g.aux) + m0 + tid, bsum);
    float* Cb = g.C + (EPI == EPI_PARTIAL ? (long)blockIdx.z * g.part_stride : 0l);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r];
                if (EPI == EPI_BIAS) v += g.aux[col];
#if defined(APX_GEMM_ABL) && APX_GEMM_ABL >= 1
                if (v == 123.456f) Cb[(long)row * g.ldc + col] = v;
#else
                if (EPI == EPI_MASK) v = g.aux[(long)row * g.ld_aux + col] > 0.f ? v : 0.f;
                Cb[(long)row * g.ldc + col] = v;
#endif
            }
        }
}
static bool gemm128_ok(int epi, const GemmArgs& g, int kchunk) {
    static const int knob = getenv("APX_GEMM128") ? atoi(getenv("APX_GEMM128")) : 1;      // 0: never; 2: whenever the shape allows it (tests: the 128-tile kernel on small problems)
    const bool off = knob == 0;
    if (off || !(epi == EPI_STORE || epi == EPI_MASK || epi == EPI_PARTIAL || epi == EPI_BIAS)) return false;
    if (g.M % G2M || g.N % G2N || g.K % GBK || kchunk % GBK || g.K < 64) return false;
    if (knob != 2 && epi != EPI_PARTIAL && (long)(g.M / G2M) * (g.N / G2N) < 384) return false;      // too few 128 x 128 tiles to fill the chip: the 64 x 64 kernel's 4x workgroups hide the latency better
    if (((uintptr_t)g.B & 15) || ((uintptr_t)g.A & 15)) return false;
    if (!((g.b_cs == 1 && g.b_rs % 4 == 0) || (g.b_rs == 1 && g.b_cs % 4 == 0 && g.b_cs != 1))) return false;
    if (g.a_cs == 1) return g.a_rs % 4 == 0;
    return g.a_rs == 1 && g.a_cs % 4 == 0;
}

static int launch_gemm(int epi, const GemmArgs& g0, int ksplit, hipStream_t s) {
    GemmArgs g = g0;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return APX_OK;
    if (epi != EPI_ATOMIC && epi != EPI_PARTIAL) ksplit = 1;
    const int kt = GBK;
    int kchunk = (g.K + ksplit - 1) / ksplit;
    kchunk = ((kchunk + kt - 1) / kt) * kt;
    g.kchunk = kchunk;
    const int nz = (g.K + kchunk - 1) / kchunk;
    dim3 grid(apx_cdiv(g.M, GBM), apx_cdiv(g.N, GBN), nz), block(256);
    if (gemm128_ok(epi, g, kchunk)) {
        dim3 grid2(g.M / G2M, g.N / G2N, nz);
        switch (epi) {
            case EPI_STORE: hipLaunchKernelGGL(gemm_f32_128_kernel<EPI_STORE>, grid2, block, 0, s, g); break;
            case EPI_MASK: hipLaunchKernelGGL(gemm_f32_128_kernel<EPI_MASK>, grid2, block, 0, s, g); break;
            case EPI_BIAS: hipLaunchKernelGGL(gemm_f32_128_kernel<EPI_BIAS>, grid2, block, 0, s, g); break;
            default: hipLaunchKernelGGL(gemm_f32_128_kernel<EPI_PARTIAL>, grid2, block, 0, s, g); break;
        }
        APX_LAUNCH_CHECK();
        return APX_OK;
    }
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL(gemm_f32_kernel<EPI_STORE>, grid, block, 0, s, g); break;
        case EPI_BIAS: hipLaunchKernelGGL(gemm_f32_kernel<EPI_BIAS>, grid, block, 0, s, g); break;
        case EPI_BIAS_RELU: hipLaunchKernelGGL(gemm_f32_kernel<EPI_BIAS_RELU>, grid, block, 0, s, g); break;
        case EPI_MASK: hipLaunchKernelGGL(gemm_f32_kernel<EPI_MASK>, grid, block, 0, s, g); break;
        case EPI_ATOMIC: hipLaunchKernelGGL(gemm_f32_kernel<EPI_ATOMIC>, grid, block, 0, s, g); break;
        case EPI_ACC: hipLaunchKernelGGL(gemm_f32_kernel<EPI_ACC>, grid, block, 0, s, g); break;
        case EPI_PARTIAL: hipLaunchKernelGGL(gemm_f32_kernel<EPI_PARTIAL>, grid, block, 0, s, g); break;
        default: return APX_E_ARG;
    }
    APX_LAUNCH_CHECK();
    return APX_OK;
}

// Y[B,Dout] = act(X[B,Din] W^T + b), W torch layout [Dout, Din]
static int linear_fwd(const float* X, const float* W, const float* b, float* Y, long B, int Din, int Dout, bool relu,
                      hipStream_t s) {
    GemmArgs g{X, Din, 1, W, 1, Din, Y, Dout, b, 0, (int)B, Dout, Din, 0, 0};
    return launch_gemm(relu ? EPI_BIAS_RELU : EPI_BIAS, g, 1, s);
}
// dX[B,Din] = (dY[B,Dout] W) * (mask > 0)   (mask = saved post-ReLU activation of the layer below, or NULL)
static int linear_bwd_input(const float* dY, const float* W, const float* mask, float* dX, long B, int Din, int Dout,
                            hipStream_t s) {
    GemmArgs g{dY, Dout, 1, W, Din, 1, dX, Din, mask, Din, (int)B, Din, Dout, 0, 0};
    return launch_gemm(mask ? EPI_MASK : EPI_STORE, g, 1, s);
}
// dW[Dout,Din] += dY^T X  (split-K over the batch)
// db[Dout] += column sums of dY, fused into the same launch (they ride on the dY tile already in LDS)
// Two ways to add the K chunks up.  fp32 atomics straight into dW: simple, but the L2 atomic units bound the launch (the 256 x 256 gradient at 64 chunks is 4.2 M atomic adds,
// the 256 x 50 one at 256 chunks 3.3 M: 20 - 30 us each, more than their MFMA time).  Or, when the caller lends scratch (GradParts): every chunk STORES its product to its own
// slab and ONE grad_reduce_kernel per minibatch adds the slabs of all six weight gradients into the flat gradient - coalesced streams instead of atomics.
struct GradSeg { float* dst; const float* src; int count, nz; };
struct GradParts {
    float* scratch; size_t cap, used;      // floats
    GradSeg seg[8]; int nseg;
};
constexpr size_t GRAD_PART_FLOATS = (size_t)1024 * GBM * GBN;      // upper bound of one weight gradient's slabs: about 1024 workgroups x one 64 x 64 tile each
static int linear_bwd_weight(const float* dY, const float* X, float* dW, float* db, long B, int Din, int Dout, hipStream_t s, GradParts* parts = nullptr) {
    GemmArgs g{dY, 1, Dout, X, Din, 1, dW, Din, db, 0, Dout, Din, (int)B, 0, 0};
    // split-K sized to the chip, not to the batch: about 1024 workgroups whatever the shape of dW, at least 64 batch rows per workgroup (fewer leave CUs idle on the
    // 10 x 256 and 256 x 50 gradients; measured with APX_KSPLIT_WGS = 256 / 512 / 1024 / 2048 on the bench minibatch, tools/t_ksplit_sweep.sh)
    const long tiles = (long)apx_cdiv(Dout, GBM) * apx_cdiv(Din, GBN);
    static const long target_wgs = getenv("APX_KSPLIT_WGS") ? atol(getenv("APX_KSPLIT_WGS")) : 1024;
    long ksplit = target_wgs / tiles;
    static const long ks128 = getenv("APX_KSPLIT128") ? atol(getenv("APX_KSPLIT128")) : 64;
    if (parts && Dout % G2M == 0 && Din % G2N == 0) ksplit = ks128;      // 128 x 128 tiles (gemm_f32_128_kernel): 4 tiles x 64 chunks = one workgroup per CU
    if (ksplit > B / 64) ksplit = B / 64;
    if (ksplit < 1) ksplit = 1;
    if (parts && ksplit > 1 && parts->nseg < 8) {
        int kchunk = (int)((B + ksplit - 1) / ksplit); kchunk = ((kchunk + GBK - 1) / GBK) * GBK;
        const int nz = (int)((B + kchunk - 1) / kchunk);
        const size_t need = (size_t)nz * Dout * Din;
        if (parts->used + need <= parts->cap) {
            float* slab = parts->scratch + parts->used; parts->used += need;
            g.C = slab; g.part_stride = (long)Dout * Din;
            parts->seg[parts->nseg++] = GradSeg{dW, slab, Dout * Din, nz};
            return launch_gemm(EPI_PARTIAL, g, (int)ksplit, s);
        }
    }
    return launch_gemm(EPI_ATOMIC, g, (int)ksplit, s);
}
// Output layer of the backward pass, one launch instead of two GEMMs: dW2 = dY^T A2 (K chunk slab), db2 += column sums of dY, and dH2 = (dY W2) * (A2 > 0).  With O = 10 (or 1)
// outputs both products are 10 multiply-adds per element of A2 and the launch is the stream A2 in, dH2 out; as 64 x 64 MFMA tiles they were two launches that each read A2
// (and wasted 54 of the 64 tile rows).  One thread per hidden column: W2's column and the O running sums stay in registers, the chunk's rows of dY are broadcast from LDS.
#define BH_ROWS 64
template <int O>
__global__ __launch_bounds__(256) void bwd_head_kernel(const float* __restrict__ dY, const float* __restrict__ W2, const float* __restrict__ A2, float* __restrict__ dH2,
                                                       float* __restrict__ slab, float* __restrict__ db, long B, int H) {
    __shared__ float dys[BH_ROWS][O];
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * BH_ROWS;
    const int nr = (int)min((long)BH_ROWS, B - r0);
    for (int e = tid; e < BH_ROWS * O; e += 256) { const int r = e / O, o = e - r * O; dys[r][o] = r < nr ? dY[(r0 + r) * O + o] : 0.f; }
    __syncthreads();
    for (int c = tid; c < H; c += 256) {
        float w[O], acc[O];
#pragma unroll
        for (int o = 0; o < O; ++o) { w[o] = W2[(long)o * H + c]; acc[o] = 0.f; }
        auto row = [&](int r, float a) {
            float d = 0.f;
#pragma unroll
            for (int o = 0; o < O; ++o) { d += dys[r][o] * w[o]; acc[o] += dys[r][o] * a; }
            dH2[(r0 + r) * H + c] = a > 0.f ? d : 0.f;
        };
        if (nr == BH_ROWS) {      // whole chunk: 16 rows of A2 in flight per thread (the launch is a stream, its speed is the number of loads in flight)
#pragma unroll 1
            for (int rb = 0; rb < BH_ROWS; rb += 16) {
                float av[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) av[k] = A2[(r0 + rb + k) * H + c];
#pragma unroll
                for (int k = 0; k < 16; ++k) row(rb + k, av[k]);
            }
        } else for (int r = 0; r < nr; ++r) row(r, A2[(r0 + r) * H + c]);
#pragma unroll
        for (int o = 0; o < O; ++o) slab[(long)blockIdx.x * O * H + (long)o * H + c] = acc[o];
    }
    if (tid < O) { float b = 0.f; for (int r = 0; r < BH_ROWS; ++r) b += dys[r][tid]; atomicAdd(db + tid, b); }
}
// returns false when the shape / scratch does not fit (the caller falls back to the two GEMMs)
static bool bwd_head(const float* dY, const float* W2, const float* A2, float* dH2, float* dW2, float* db2, long B, int H, int O, GradParts* parts, hipStream_t s) {
    if (!parts || parts->nseg >= 8 || (O != 1 && O != 10)) return false;
    const int nz = (int)apx_cdiv(B, BH_ROWS);
    const size_t need = (size_t)nz * O * H;
    if (parts->used + need > parts->cap) return false;
    float* slab = parts->scratch + parts->used; parts->used += need;
    parts->seg[parts->nseg++] = GradSeg{dW2, slab, O * H, nz};
    if (O == 10) hipLaunchKernelGGL(bwd_head_kernel<10>, dim3(nz), dim3(256), 0, s, dY, W2, A2, dH2, slab, db2, B, H);
    else hipLaunchKernelGGL(bwd_head_kernel<1>, dim3(nz), dim3(256), 0, s, dY, W2, A2, dH2, slab, db2, B, H);
    return hipGetLastError() == hipSuccess;
}
struct GradSegs { GradSeg seg[8]; int nseg; };
// 64 elements x 4 groups of slabs per workgroup: the slab loop is a chain of dependent-latency loads, so the parallelism comes from threads, not from the loop
__global__ __launch_bounds__(256) void grad_reduce_kernel(GradSegs G) {
    __shared__ float part[4][64];
    const int el = threadIdx.x & 63, zg = threadIdx.x >> 6;
    long e = blockIdx.x * 64l + el;
    float sum = 0.f; float* dst = nullptr;
#pragma unroll 1
    for (int k = 0; k < G.nseg; ++k) {
        const GradSeg& q = G.seg[k];
        const long padded = ((long)q.count + 63) / 64 * 64;      // a workgroup never straddles two segments
        if (e < padded) {
            if (e < q.count) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                const int z0 = (int)((long)q.nz * zg / 4), z1 = (int)((long)q.nz * (zg + 1) / 4);
                int z = z0;
                for (; z + 3 < z1; z += 4) {
                    a0 += q.src[(long)z * q.count + e]; a1 += q.src[(long)(z + 1) * q.count + e]; a2 += q.src[(long)(z + 2) * q.count + e]; a3 += q.src[(long)(z + 3) * q.count + e];
                }
                for (; z < z1; ++z) a0 += q.src[(long)z * q.count + e];
                sum = (a0 + a1) + (a2 + a3); dst = q.dst + e;
            }
            break;
        }
        e -= padded;
    }
    part[zg][el] = sum;
    __syncthreads();
    if (zg == 0 && dst) *dst += (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
}
static int grad_reduce(GradParts& parts, hipStream_t s) {
    if (parts.nseg == 0) return APX_OK;
    GradSegs G; long total = 0;
    for (int k = 0; k < parts.nseg; ++k) { G.seg[k] = parts.seg[k]; total += ((long)parts.seg[k].count + 63) / 64; }
    G.nseg = parts.nseg;
    hipLaunchKernelGGL(grad_reduce_kernel, dim3(total), dim3(256), 0, s, G);
    APX_LAUNCH_CHECK();
    parts.nseg = 0; parts.used = 0;
    return APX_OK;
}

extern "C" size_t apx_mlp_param_count(int D, int H, int O) {
    return (size_t)H * D + H + (size_t)H * H + H + (size_t)O * H + O;
}

struct MlpView {
    const float *W0, *b0, *W1, *b1, *W2, *b2;
    MlpView(const float* p, int D, int H, int O) {
        W0 = p; b0 = W0 + (size_t)H * D; W1 = b0 + H; b1 = W1 + (size_t)H * H; W2 = b1 + H; b2 = W2 + (size_t)O * H;
    }
};
struct MlpGrad {
    float *W0, *b0, *W1, *b1, *W2, *b2;
    MlpGrad(float* p, int D, int H, int O) {
        W0 = p; b0 = W0 + (size_t)H * D; W1 = b0 + H; b1 = W1 + (size_t)H * H; W2 = b1 + H; b2 = W2 + (size_t)O * H;
    }
};

#define APX_TRY(x) do { int rc__ = (x); if (rc__ != APX_OK) return rc__; } while (0)

// ------------------------------------------------------------------------------------------------ fused 3-layer forward
// One workgroup carries 32 rows through Linear-ReLU-Linear-ReLU-Linear (H = 256, D <= 64, O <= 128) with the activations resident
// in LDS (k-major, so the MFMA operand fetch stays a conflict-free ds_read_b32); only the weight tiles stream in, through a
// register pipeline FPF tiles deep (an L2 hit costs ~0.7 us, one 16-deep k-tile of MFMAs only ~0.25 us).  Each of the 4 waves
// owns a 32-column slice of a 128-column pass, i.e. one 32x32 MFMA accumulator.  a1 / a2 are still written to HBM: the backward
// pass needs them.  LDS: A1[256][33] + A2[256][33] (X aliases the head of A2) + W tile [16][129] = 75 840 B: two workgroups per CU.
#define FH 256
#define FBM 32
#define FBN 128
#define FPF 4
struct FusedLds { float (*A1s)[FBM + 1]; float (*A2s)[FBM + 1]; float (*Ws)[FBN + 1]; };
// one layer: out[32, N] = act(In[32, K] W^T + b), In = k-major LDS, W torch layout [N, K]; NKT = ceil(K / 16) k-tiles
template <int NKT>
__device__ __forceinline__ void fused_layer(const FusedLds& L, float (*In)[FBM + 1], int K, const float* __restrict__ W, const float* __restrict__ bias,
                                            int N, bool relu, float (*OutS)[FBM + 1], float* __restrict__ outG, int ldo, long m0, long B,
                                            float* __restrict__ actG = nullptr, const float* __restrict__ noise = nullptr, float sigma = 0.f) {
    // Round 4: the weight operand no longer goes through LDS.  A lane of v_mfma_f32_32x32x2_f32 holds B[k][n] for n = lane & 31 and ONE k per instruction; with the k of a
    // 16-deep tile dealt as k = 8 (lane >> 5) + j over the 8 instructions j of the tile (instead of 2 j + (lane >> 5)), a lane's eight operands are 32 contiguous bytes of
    // its weight row W[n][.]: four dwordx2 loads straight from L2 into the registers the MFMAs read, FPF tiles ahead.  No weight tile in LDS, no barrier inside a layer
    // (the old form: write tile, barrier, 8 MFMAs, barrier - 0.68 us per tile step against 0.24 us of MFMA work).  Rows k >= K of `In` are zero, so what a lane reads
    // past the end of its weight row (the next row / the next parameter block: finite numbers) is multiplied by zero.
    const int tid = threadIdx.x, lane = tid & 63, wn = (tid >> 6) * 32, h = lane >> 5;
    typedef float f2w __attribute__((ext_vector_type(2)));
    for (int nb = 0; nb * FBN < N; ++nb) {
        floatx16 acc = {0};
        const int col = nb * FBN + wn + (lane & 31);
        const bool wave_on = nb * FBN + wn < N;      // a wave whose 32-column slice lies beyond N (output layer) leaves the MFMA pipe to the co-resident workgroup
        const float* wrow = W + (long)(col < N ? col : N - 1) * K + 8 * h;
        f2w rw[FPF][4];
        const bool k_even = (K & 1) == 0 && (reinterpret_cast<uintptr_t>(W) & 7) == 0;      // an odd row length (D = 25 / 55: min / phase profiles) or a block at an odd float offset (TD3's second critic inside critic_flat) leaves the rows 4-byte aligned only: scalar loads there
        auto fetch = [&](int kt, f2w (&r)[4]) {
            if (k_even) {
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const f2w*>(wrow + kt * GBK + 2 * q);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = f2w{wrow[kt * GBK + 2 * q], wrow[kt * GBK + 2 * q + 1]};
            }
        };
        if (wave_on) {
#pragma unroll
            for (int p = 0; p < FPF; ++p) if (p < NKT) fetch(p, rw[p]);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                f2w cur[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[q] = rw[kt % FPF][q];
                if (kt + FPF < NKT) fetch(kt + FPF, rw[kt % FPF]);
                // (without the fences the scheduler sinks every fetch to just above its MFMAs to save registers - one load, s_waitcnt vmcnt(0), four MFMAs, twice per
                // tile: an exposed L2 round trip per half tile and an MFMA pipe that is busy half of the time)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float av = In[kt * GBK + 8 * h + j][lane & 31];
                    const float bv = (j & 1) ? cur[j >> 1].y : cur[j >> 1].x;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const float bs = col < N ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            float v = acc[r] + bs;
            if (relu) v = fmaxf(v, 0.f);
            if (OutS && col < N) OutS[col][row] = v;
            if (outG && col < N && m0 + row < B) outG[(m0 + row) * ldo + col] = v;
            if (actG && col < N && m0 + row < B) actG[(m0 + row) * ldo + col] = v + (noise ? sigma * noise[(m0 + row) * ldo + col] : 0.f);
        }
    }
    __syncthreads();
}
// The OUTPUT layer (N <= 32: one 32-column slice): as a column-sliced layer it is 16 k-tiles on wave 0 alone while the other three waves - and, with two workgroups per
// CU, three of the four MFMA pipes - idle: SIMD 0 then carries 28.7 k MFMA cycles per workgroup against 20.5 k on the others and bounds the launch.  Here the K range is
// dealt over the four waves (4 k-tiles each, all prefetched at once), the four partial accumulators meet in the (free) A1 tile and wave 0 runs the epilogue.
template <int NKT>
__device__ __forceinline__ void fused_out_layer(const FusedLds& L, float (*In)[FBM + 1], int K, const float* __restrict__ W, const float* __restrict__ bias, int N,
                                                float* __restrict__ outG, int ldo, long m0, long B, float* __restrict__ actG, const float* __restrict__ noise, float sigma) {
    static_assert(NKT % 4 == 0, "k-tiles dealt over four waves");
    constexpr int KW = NKT / 4;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, col = lane & 31;
    typedef float f2w __attribute__((ext_vector_type(2)));
    const float* wrow = W + (long)(col < N ? col : N - 1) * K + 8 * h;
    const bool w_al8 = (reinterpret_cast<uintptr_t>(W) & 7) == 0 && (K & 1) == 0;
    f2w rw[KW][4];
#pragma unroll
    for (int p = 0; p < KW; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // (K = 256: even; 8-byte loads when the block itself is 8-byte aligned - TD3's second critic sits at an odd float offset of critic_flat)
            const float* wp = wrow + (w * KW + p) * GBK + 2 * q;
            rw[p][q] = w_al8 ? *reinterpret_cast<const f2w*>(wp) : f2w{wp[0], wp[1]};
        }
    floatx16 acc = {0};
#pragma unroll
    for (int p = 0; p < KW; ++p)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float av = In[(w * KW + p) * GBK + 8 * h + j][col];
            const float bv = (j & 1) ? rw[p][j >> 1].y : rw[p][j >> 1].x;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    float* red = reinterpret_cast<float*>(L.A1s);      // [4 waves][16][64] floats = 16 KB of the 33 KB tile
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (w == 0) {
        const float bs = col < N ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float v = ((red[r * 64 + lane] + red[(16 + r) * 64 + lane]) + (red[(32 + r) * 64 + lane] + red[(48 + r) * 64 + lane])) + bs;
            if (outG && col < N && m0 + row < B) outG[(m0 + row) * ldo + col] = v;
            if (actG && col < N && m0 + row < B) actG[(m0 + row) * ldo + col] = v + (noise ? sigma * noise[(m0 + row) * ldo + col] : 0.f);
        }
    }
    __syncthreads();
}
struct FusedIn {           // where the 32 x D input tile comes from: prepared rows (everything else NULL) or raw observations + the prep of prep_obs_kernel
    const float* x; const int64_t* idx; const int32_t* sign_perm; uint64_t clock_mask; const float *mean, *stdv; float* xn_out;
    // optional epilogue of the output layer (the rollout, apx_rollout): act = y + sigma * noise (noise may be NULL: act = y), saving the act_noise launch of every env step
    float* act_out; const float* noise; float sigma;
};
__global__ __launch_bounds__(256) void mlp_fused_fwd_kernel(const float* __restrict__ W0, const float* __restrict__ b0, const float* __restrict__ W1,
                                                            const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                                                            FusedIn I, long B, int D, int O, float* __restrict__ a1,
                                                            float* __restrict__ a2, float* __restrict__ y) {
    APX_DYNAMIC_LDS(float, fls, 4);
    FusedLds L;
    L.A1s = reinterpret_cast<float (*)[FBM + 1]>(fls + 0);
    L.A2s = reinterpret_cast<float (*)[FBM + 1]>(fls + FH * (FBM + 1));
    L.Ws = reinterpret_cast<float (*)[FBN + 1]>(fls + 2 * FH * (FBM + 1));
    const int tid = threadIdx.x;
    const long m0 = (long)blockIdx.x * FBM;
    for (int e = tid; e < FBM * 64; e += 256) {        // X tile -> A2s[k][m], zero padded to 64 columns
        const int m = e >> 6, k = e & 63;
        float v = 0.f;
        if (m0 + m < B && k < D) {
            const long row = I.idx ? I.idx[m0 + m] : m0 + m;
            if (I.sign_perm) {
                const int32_t sp = I.sign_perm[k];
                v = sp >= 0 ? I.x[row * D + sp] : -I.x[row * D + (-sp - 1)];
                if ((I.clock_mask >> k) & 1ull) v = sinf(asinf(v) + 3.14159265358979323846f);      // wrappers.py:65-66
            } else {
                v = I.x[row * D + k];
            }
            if (I.mean) v = (v - I.mean[k]) / I.stdv[k];
            if (I.xn_out) I.xn_out[(m0 + m) * D + k] = v;
        }
        L.A2s[k][m] = v;
    }
    __syncthreads();
    fused_layer<4>(L, L.A2s, D, W0, b0, FH, true, L.A1s, a1, FH, m0, B);      // k-tiles beyond D: zero-padded X rows, zero-guarded weights
    fused_layer<FH / GBK>(L, L.A1s, FH, W1, b1, FH, true, L.A2s, a2, FH, m0, B);
    if (O <= 32) fused_out_layer<FH / GBK>(L, L.A2s, FH, W2, b2, O, y, O, m0, B, I.act_out, I.noise, I.sigma);
    else fused_layer<FH / GBK>(L, L.A2s, FH, W2, b2, O, false, nullptr, y, O, m0, B, I.act_out, I.noise, I.sigma);
}

static bool g_fused_attr_set = false;
static bool fused_ok(int D, int H, int O) {
    static const bool use_fused = getenv("APX_MLP_UNFUSED") == nullptr;
    return use_fused && H == FH && D <= 64 && O <= FBN;     // the reference's 2 x 256 nets
}
static int mlp_fused_launch(const float* params, int D, int H, int O, const FusedIn& in, long B, float* a1, float* a2, float* y, hipStream_t s) {
    MlpView p(params, D, H, O);
    const size_t lds = sizeof(float) * (2 * FH * (FBM + 1));      // activations only: the weights go from L2 straight into the MFMA operand registers
    if (!g_fused_attr_set) {
        APX_HIP(hipFuncSetAttribute((const void*)mlp_fused_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        g_fused_attr_set = true;
    }
    hipLaunchKernelGGL(mlp_fused_fwd_kernel, dim3(apx_cdiv(B, FBM)), dim3(256), lds, s, p.W0, p.b0, p.W1, p.b1, p.W2, p.b2, in, B, D, O, a1, a2, y);
    APX_LAUNCH_CHECK();
    return APX_OK;
}
static int mlp_forward_impl(const float* params, int D, int H, int O, const float* xn, long B, float* a1, float* a2,
                            float* y, hipStream_t s) {
    MlpView p(params, D, H, O);
    if (fused_ok(D, H, O))       // one launch, activations stay in LDS (fp32 MFMA)
        return mlp_fused_launch(params, D, H, O, FusedIn{xn, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f}, B, a1, a2, y, s);
    APX_TRY(linear_fwd(xn, p.W0, p.b0, a1, B, D, H, true, s));
    APX_TRY(linear_fwd(a1, p.W1, p.b1, a2, B, H, H, true, s));
    APX_TRY(linear_fwd(a2, p.W2, p.b2, y, B, H, O, false, s));
    return APX_OK;
}

// grads += d(loss)/d(params) given dy = d(loss)/d(y); dh1/dh2 are [B,H] scratch
static int mlp_backward_impl(const float* params, float* grads, int D, int H, int O, const float* xn, const float* a1,
                             const float* a2, const float* dy, long B, float* dh2, float* dh1, hipStream_t s, GradParts* parts = nullptr) {
    MlpView p(params, D, H, O);
    MlpGrad g(grads, D, H, O);
    if (!(bwd_head(dy, p.W2, a2, dh2, g.W2, g.b2, B, H, O, parts, s))) {
        APX_TRY(linear_bwd_weight(dy, a2, g.W2, g.b2, B, H, O, s, parts));
        APX_TRY(linear_bwd_input(dy, p.W2, a2, dh2, B, H, O, s));
    }
    APX_TRY(linear_bwd_weight(dh2, a1, g.W1, g.b1, B, H, H, s, parts));
    APX_TRY(linear_bwd_input(dh2, p.W1, a1, dh1, B, H, H, s));
    APX_TRY(linear_bwd_weight(dh1, xn, g.W0, g.b0, B, D, H, s, parts));
    return APX_OK;
}

static int prep_obs(const float* x, long B, int D, const int64_t* idx, const int32_t* sp, uint64_t cm, const float* mean,
                    const float* stdv, float* out, hipStream_t s) {
    hipLaunchKernelGGL(prep_obs_kernel, dim3(apx_cdiv(B * D, 256)), dim3(256), 0, s, x, (int64_t)B, D, idx, sp, cm, mean,
                       stdv, out);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

extern "C" int apx_mlp_forward(const float* params, int D, int H, int O, const float* x, int64_t B, const int64_t* idx,
                               const int32_t* sign_perm, uint64_t clock_mask, const float* obs_mean,
                               const float* obs_std, float* xn_out, float* act1, float* act2, float* y,
                               void* stream) {
    APX_REQUIRE(D > 0 && H > 0 && O > 0 && B >= 0, "dims");
    if (B == 0) return APX_OK;   // empty batch: nothing to do (empty tensors have NULL data pointers)
    APX_REQUIRE(params && x && y, "null pointer");
    APX_REQUIRE((obs_mean == nullptr) == (obs_std == nullptr), "obs_mean/obs_std");
    hipStream_t s = (hipStream_t)stream;
    if (fused_ok(D, H, O))      // input prep inside the fused kernel; NULL outputs are simply not written
        return mlp_fused_launch(params, D, H, O, FusedIn{x, idx, sign_perm, clock_mask, obs_mean, obs_std, xn_out, nullptr, nullptr, 0.f}, B, act1, act2, y, s);
    APX_REQUIRE(xn_out && act1 && act2, "xn_out / act1 / act2 may only be NULL for the fused fp32 shapes (H = 256, D <= 64, O <= 128)");
    APX_TRY(prep_obs(x, B, D, idx, sign_perm, clock_mask, obs_mean, obs_std, xn_out, s));
    return mlp_forward_impl(params, D, H, O, xn_out, B, act1, act2, y, s);
}

// apx_rollout's policy step: mu = pi(normalise(obs)) and act = mu + sigma * noise in ONE launch when the fused shape applies (returns 1 if it did, 0 if the caller has to
// run apx_mlp_forward + the noise kernel, < 0 on error); internal (apx_common.h), not part of the C ABI
int apx_mlp_forward_act(const float* params, int D, int H, int O, const float* x, int64_t B, const float* obs_mean, const float* obs_std, float* y, float* act,
                        const float* noise, float sigma, void* stream) {
    if (!fused_ok(D, H, O) || B <= 0) return 0;
    const int rc = mlp_fused_launch(params, D, H, O, FusedIn{x, nullptr, nullptr, 0, obs_mean, obs_std, nullptr, act, noise, sigma}, B, nullptr, nullptr, y, (hipStream_t)stream);
    return rc == APX_OK ? 1 : -1;
}

// ------------------------------------------------------------------------------------------------ TD3 primitives (next row f2)
// grads += d(loss)/d(params) of a 3-layer ReLU MLP for dy[B,O] (xn, a1, a2 from apx_mlp_forward), optionally dx[B,D] = d(loss)/d(input)
extern "C" int apx_mlp_backward(const float* params, float* grads, int D, int H, int O, const float* xn, const float* a1, const float* a2,
                                const float* dy, int64_t B, float* dx, float* scratch /* 2*B*H floats */, void* stream) {
    APX_REQUIRE(params && xn && a1 && a2 && dy && scratch && B > 0 && (grads || dx), "mlp backward arguments");
    hipStream_t s = (hipStream_t)stream;
    float* dh2 = scratch; float* dh1 = scratch + (size_t)B * H;
    if (grads) APX_TRY(mlp_backward_impl(params, grads, D, H, O, xn, a1, a2, dy, B, dh2, dh1, s));
    else {      // input gradient only (the actor loss -Q1(s, pi(s)) must not touch the critic's parameter gradients)
        MlpView p(params, D, H, O);
        APX_TRY(linear_bwd_input(dy, p.W2, a2, dh2, B, H, O, s));
        APX_TRY(linear_bwd_input(dh2, p.W1, a1, dh1, B, H, H, s));
    }
    if (dx) { MlpView p(params, D, H, O); APX_TRY(linear_bwd_input(dh1, p.W0, nullptr, dx, B, D, H, s)); }
    return APX_OK;
}
__global__ void polyak_kernel(float* __restrict__ target, const float* __restrict__ param, long n, float tau) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e < n) target[e] = tau * param[e] + (1.f - tau) * target[e];
}
// target <- tau * param + (1 - tau) * target (sync_td3.py:196-202)
extern "C" int apx_polyak(float* target, const float* param, int64_t n, float tau, void* stream) {
    APX_REQUIRE(target && param && n > 0, "polyak arguments");
    hipLaunchKernelGGL(polyak_kernel, dim3(apx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, target, param, (long)n, tau);
    APX_LAUNCH_CHECK();
    return APX_OK;
}
// out[b, :D] = s[b], out[b, D + j] = max_action * tanh(pre[b, j]) (+ clamp(noise)) clamped to +-max_action: the critic input cat(state, action)
__global__ void td3_cat_action_kernel(const float* __restrict__ s, const float* __restrict__ pre, const float* __restrict__ noise, float noise_clip,
                                      float max_action, long B, int D, int A, float* __restrict__ out) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= B * (D + A)) return;
    const long b = e / (D + A); const int c = (int)(e - b * (D + A));
    if (c < D) { out[e] = s[b * D + c]; return; }
    const int j = c - D;
    float a = max_action * tanhf(pre[b * A + j]);
    if (noise) a = fminf(fmaxf(a + fminf(fmaxf(noise[b * A + j], -noise_clip), noise_clip), -max_action), max_action);
    out[e] = a;
}
extern "C" int apx_td3_cat_action(const float* state, const float* pre_tanh, const float* noise, float noise_clip, float max_action, int64_t B,
                                  int D, int A, float* out, void* stream) {
    APX_REQUIRE(state && pre_tanh && out && B > 0, "td3 cat arguments");
    hipLaunchKernelGGL(td3_cat_action_kernel, dim3(apx_cdiv(B * (D + A), 256)), dim3(256), 0, (hipStream_t)stream, state, pre_tanh, noise, noise_clip,
                       max_action, (long)B, D, A, out);
    APX_LAUNCH_CHECK();
    return APX_OK;
}
// target = r + notdone * discount * min(tq1, tq2); dq1 = 2 (q1 - target) / B, dq2 likewise; acc += (critic loss, sum q1, sum q2)
__global__ void td3_critic_loss_kernel(const float* __restrict__ q1, const float* __restrict__ q2, const float* __restrict__ tq1, const float* __restrict__ tq2,
                                       const float* __restrict__ r, const float* __restrict__ notdone, float discount, long B, float* __restrict__ dq1,
                                       float* __restrict__ dq2, double* __restrict__ acc) {
    const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
    double a[3] = {0, 0, 0};
    if (b < B) {
        const float t = r[b] + notdone[b] * discount * fminf(tq1[b], tq2[b]);
        const float e1 = q1[b] - t, e2 = q2[b] - t;
        dq1[b] = 2.f * e1 / (float)B; dq2[b] = 2.f * e2 / (float)B;
        a[0] = ((double)e1 * e1 + (double)e2 * e2) / (double)B; a[1] = q1[b]; a[2] = q2[b];
    }
    block_atomic_add<3>(a, acc);
}
extern "C" int apx_td3_critic_loss(const float* q1, const float* q2, const float* tq1, const float* tq2, const float* reward, const float* notdone,
                                   float discount, int64_t B, float* dq1, float* dq2, double* acc3, void* stream) {
    APX_REQUIRE(q1 && q2 && tq1 && tq2 && reward && notdone && dq1 && dq2 && acc3 && B > 0, "td3 critic loss arguments");
    hipStream_t s = (hipStream_t)stream;
    APX_HIP(hipMemsetAsync(acc3, 0, 3 * sizeof(double), s));
    hipLaunchKernelGGL(td3_critic_loss_kernel, dim3(apx_cdiv(B, 256)), dim3(256), 0, s, q1, q2, tq1, tq2, reward, notdone, discount, (long)B, dq1, dq2, acc3);
    APX_LAUNCH_CHECK();
    return APX_OK;
}
// actor loss -mean Q1(s, pi(s)): d(loss)/d(pre_tanh)[b, j] = dx[b, D + j] * max_action * (1 - tanh(pre)^2), with dx = d(-mean q1)/d(critic input)
__global__ void td3_actor_grad_kernel(const float* __restrict__ dx, const float* __restrict__ pre, float max_action, long B, int D, int A,
                                      float* __restrict__ dpre) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= B * A) return;
    const long b = e / A; const int j = (int)(e - b * A);
    const float t = tanhf(pre[e]);
    dpre[e] = dx[b * (D + A) + D + j] * max_action * (1.f - t * t);
}
extern "C" int apx_td3_actor_grad(const float* dx, const float* pre_tanh, float max_action, int64_t B, int D, int A, float* dpre, void* stream) {
    APX_REQUIRE(dx && pre_tanh && dpre && B > 0, "td3 actor grad arguments");
    hipLaunchKernelGGL(td3_actor_grad_kernel, dim3(apx_cdiv(B * A, 256)), dim3(256), 0, (hipStream_t)stream, dx, pre_tanh, max_action, (long)B, D, A, dpre);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

// ------------------------------------------------------------------------------------------------ LSTM (recurrent actor / critic)
// Gaussian_LSTM_Actor / LSTM_V (rl/policies/actor.py:218-311, critic.py:236-296): L stacked nn.LSTMCell(H) + Linear(H, O), run over a
// padded batch of trajectories x[T, B, D] from zero state (or one step from a carried state).  Parameters in state_dict order: per
// cell weight_ih[4H, in] weight_hh[4H, H] bias_ih[4H] bias_hh[4H] (gate order i, f, g, o), then network_out weight[O, H] bias[O].
// The input projections of ALL time steps are one GEMM per layer; only h_{t-1} W_hh^T + the gate non-linearities are sequential
// (one accumulate-GEMM + one pointwise launch per step and layer).  Saved for the backward pass per layer: activated gates
// [T, B, 4H], cell state [T, B, H], hidden state [T, B, H].
struct LstmView {
    const float *Wih[4], *Whh[4], *bih[4], *bhh[4], *Wo, *bo; int in[4];
    LstmView(const float* p, int D, int H, int L, int O) {
        for (int l = 0; l < L; ++l) {
            in[l] = l ? H : D;
            Wih[l] = p; p += (size_t)4 * H * in[l]; Whh[l] = p; p += (size_t)4 * H * H; bih[l] = p; p += 4 * H; bhh[l] = p; p += 4 * H;
        }
        Wo = p; bo = p + (size_t)O * H;
    }
};
extern "C" size_t apx_lstm_param_count(int D, int H, int L, int O) {
    size_t n = 0;
    for (int l = 0; l < L; ++l) n += (size_t)4 * H * (l ? H : D) + (size_t)4 * H * H + 8 * H;
    return n + (size_t)O * H + O;
}
extern "C" size_t apx_lstm_workspace_floats(int T, int64_t B, int H, int L) { return (size_t)L * T * B * 6 * H; }

__device__ __forceinline__ float sigmf(float x) { return 1.f / (1.f + expf(-x)); }
// gates[B, 4H]: pre-activations (x W_ih^T + b_ih + h W_hh^T) in, activated (i, f, g, o) out
__global__ void lstm_cell_fwd_kernel(float* __restrict__ gates, const float* __restrict__ bhh, const float* c_prev /* may alias c_out: the one-step call updates c in place */,
                                     float* c_out, float* __restrict__ h_out, long B, int H) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= B * H) return;
    const long b = e / H; const int j = (int)(e - b * H);
    float* g = gates + b * 4 * H;
    const float i = sigmf(g[j] + bhh[j]), f = sigmf(g[H + j] + bhh[H + j]), gg = tanhf(g[2 * H + j] + bhh[2 * H + j]),
                o = sigmf(g[3 * H + j] + bhh[3 * H + j]);
    const float c = f * (c_prev ? c_prev[e] : 0.f) + i * gg;
    g[j] = i; g[H + j] = f; g[2 * H + j] = gg; g[3 * H + j] = o;
    c_out[e] = c; h_out[e] = o * tanhf(c);
}
// dgates[B, 4H] <- d(loss)/d(pre-activations); dc_io[B, H]: d(loss)/d(c_t) from step t+1 in, d(loss)/d(c_{t-1}) out
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c, const float* __restrict__ c_prev,
                                     const float* __restrict__ dh_a, const float* __restrict__ dh_b, float* __restrict__ dc_io,
                                     float* __restrict__ dgates, long B, int H) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e >= B * H) return;
    const long b = e / H; const int j = (int)(e - b * H);
    const float* g = gates + b * 4 * H;
    const float i = g[j], f = g[H + j], gg = g[2 * H + j], o = g[3 * H + j];
    const float dh = (dh_a ? dh_a[e] : 0.f) + (dh_b ? dh_b[e] : 0.f);
    const float tc = tanhf(c[e]);
    const float dc = dh * o * (1.f - tc * tc) + dc_io[e];
    float* d = dgates + b * 4 * H;
    d[j] = dc * gg * i * (1.f - i);
    d[H + j] = dc * (c_prev ? c_prev[e] : 0.f) * f * (1.f - f);
    d[2 * H + j] = dc * i * (1.f - gg * gg);
    d[3 * H + j] = dh * tc * o * (1.f - o);
    dc_io[e] = dc * f;
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long n) {
    const long e = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (e < n) y[e] += x[e];
}

// ------------------------------------------------------------------------------------------------ persistent LSTM layer (sequence)
// One launch runs ALL time steps of one LSTM layer (rl/policies/actor.py:253-289: nn.LSTMCell stepped over the padded sequence).  A
// workgroup of four waves owns 16 batch rows for the whole sequence; W_hh (4H x H fp32 = 256 KB at H = 128) never leaves the REGISTER
// FILE: wave w keeps the recurrent weights of units [w H/4, (w + 1) H/4) for all four gates as v_mfma_f32_16x16x4_f32 B operands
// (H/4 k-quads x 4 gates x H/64 unit tiles = 256 VGPRs at H = 128).  Per step: A operand = h_{t-1} from LDS, 256 MFMAs per wave on
// top of the precomputed input projection, and the gate maths is lane-local because a lane's accumulators hold i, f, g, o of the same
// (row, unit) pairs.  Replaces 2 launches per step (accumulate-GEMM + gate kernel); the save format (activated gates, c, h) is unchanged.
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int H, int NW>      // NW waves per workgroup: wave w owns units [w H/NW, (w + 1) H/NW) of all four gates
__global__ __launch_bounds__(64 * NW, 1) void lstm_seq_fwd_kernel(float* __restrict__ G, const float* __restrict__ Whh, const float* __restrict__ bhh,
                                                               float* __restrict__ hc_h, float* __restrict__ hc_c, float* __restrict__ Cc,
                                                               float* __restrict__ Hh, int T, long B) {
    constexpr int UW = H / NW, NT = UW / 16, KQ = H / 4, HP = H + 4;
    static_assert(UW % 16 == 0 && NT >= 1, "a wave owns whole 16-unit tiles");
    __shared__ float hs[16][HP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 15, kg = lane >> 4;
    const long r0 = (long)blockIdx.x * 16;
    float W[KQ][4][NT];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) W[kq][g][nt] = Whh[(size_t)(g * H + wave * UW + nt * 16 + col) * H + 4 * kq + kg];
    float bh[4][NT], c[NT][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bh[g][nt] = bhh[g * H + wave * UW + nt * 16 + col];
    // initial state
    for (int e = tid; e < 16 * H; e += 64 * NW) { const int r = e / H, u = e - r * H; hs[r][u] = (hc_h && r0 + r < B) ? hc_h[(r0 + r) * H + u] : 0.f; }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const long row = r0 + 4 * kg + r; c[nt][r] = (hc_c && row < B) ? hc_c[row * H + wave * UW + nt * 16 + col] : 0.f; }
    __syncthreads();
    // the input projection of step t + 1 is fetched while the MFMAs of step t run (as the first thing of its own step the 16 loads were an exposed HBM round trip per
    // step: 6.9 us per step against 3.4 us of MFMA issue).  A workgroup whose 16 rows all exist runs the loop WITHOUT row guards: a guarded load / store is an exec-mask
    // region behind a branch, the compiler cannot count the stores in flight behind it and ends every step on s_waitcnt vmcnt(0) - the write latency of the step's 24
    // stores, exposed 400 times.  The h_{t-1} operands of MFMA group kq + 2 are read from LDS before the MFMAs of group kq (fenced: the scheduler had put every
    // ds_read + s_waitcnt lgkmcnt(0) directly in front of its eight MFMAs).
    float gn[4][NT][4];
    auto run = [&](auto FullTag) {
        constexpr bool FULL = decltype(FullTag)::value;
        auto fetch_g = [&](int t) {
            const float* Gt = G + (size_t)t * B * 4 * H;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const long row = r0 + 4 * kg + r;
                        gn[g][nt][r] = (FULL || row < B) ? Gt[row * 4 * H + g * H + wave * UW + nt * 16 + col] : 0.f;
                    }
        };
        fetch_g(0);
        for (int t = 0; t < T; ++t) {
            float* Gt = G + (size_t)t * B * 4 * H;
            floatx4 acc[4][NT];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[g][nt][r] = gn[g][nt][r] + bh[g][nt];
            if (t + 1 < T) fetch_g(t + 1);
            float a0 = hs[col][kg], a1 = hs[col][4 + kg];
#pragma unroll
            for (int kq = 0; kq < KQ; kq += 2) {
                float n0 = 0.f, n1 = 0.f;
                if (kq + 2 < KQ) { n0 = hs[col][4 * (kq + 2) + kg]; n1 = hs[col][4 * (kq + 3) + kg]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, W[kq][g][nt], acc[g][nt], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[g][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, W[kq + 1][g][nt], acc[g][nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a0 = n0; a1 = n1;
            }
            __syncthreads();                                   // every wave has read h_{t-1}
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long row = r0 + 4 * kg + r;
                    const int u = wave * UW + nt * 16 + col;
                    const float i = sigmf(acc[0][nt][r]), f = sigmf(acc[1][nt][r]), gg = tanhf(acc[2][nt][r]), o = sigmf(acc[3][nt][r]);
                    const float cn = f * c[nt][r] + i * gg, hn = o * tanhf(cn);
                    c[nt][r] = cn;
                    hs[4 * kg + r][u] = hn;
                    if (FULL || row < B) {
                        float* g4 = Gt + row * 4 * H + u;
                        g4[0] = i; g4[H] = f; g4[2 * H] = gg; g4[3 * H] = o;
                        Cc[(size_t)t * B * H + row * H + u] = cn; Hh[(size_t)t * B * H + row * H + u] = hn;
                    }
                }
            __syncthreads();
        }
    };
    static_assert(KQ % 2 == 0, "MFMA groups are taken in pairs");
    if (r0 + 16 <= B) run(std::true_type{}); else run(std::false_type{});
    if (hc_h)      // carried state out (rollout: hidden state of the next call)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = r0 + 4 * kg + r; const int u = wave * UW + nt * 16 + col;
                if (row < B) { hc_h[row * H + u] = hs[4 * kg + r][u]; hc_c[row * H + u] = c[nt][r]; }
            }
}

// BPTT of one layer in one launch: per step (descending) the gate backward of the wave's own units, then dh_{t-1} = dG_t W_hh with W_hh again
// register-resident (now as the [4H x H/4] slice that produces the wave's units); dG_t goes to global memory for the weight-gradient GEMMs.
template <int H, int NW>
__global__ __launch_bounds__(64 * NW, 1) void lstm_seq_bwd_kernel(const float* __restrict__ G, const float* __restrict__ Cc, const float* __restrict__ Whh,
                                                               const float* __restrict__ dHa, float* __restrict__ dG, int T, long B) {
    constexpr int UW = H / NW, NT = UW / 16, KQ = H, GP = 4 * H + 4;
    __shared__ float dgs[16][GP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 15, kg = lane >> 4;
    const long r0 = (long)blockIdx.x * 16;
    float W[KQ][NT];                                       // B operand of dh = dG W_hh: B[k = gate column][n = unit] = Whh[k][n]
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) W[kq][nt] = Whh[(size_t)(4 * kq + kg) * H + wave * UW + nt * 16 + col];
    float dc[NT][4], dhr[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dc[nt][r] = dhr[nt][r] = 0.f;
    // what step t - 1 reads (activated gates, c_{t-2}, dh from above) is fetched while the MFMAs of step t run; c_{t-1} is carried over from step t.  Like the forward
    // kernel: no row guards for a workgroup whose 16 rows all exist (exact vmcnt waits instead of vmcnt(0) behind exec-mask regions), LDS operands one block of four
    // MFMAs ahead.
    float pg[NT][4][4], pct[NT][4], pcp[NT][4], pdh[NT][4];
    auto run = [&](auto FullTag) {
        constexpr bool FULL = decltype(FullTag)::value;
        auto fetch = [&](int t, bool first) {
            const float* Gt = G + (size_t)t * B * 4 * H;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long row = r0 + 4 * kg + r; const int u = wave * UW + nt * 16 + col;
                    if (FULL || row < B) {
                        const float* g4 = Gt + row * 4 * H + u;
                        pg[nt][r][0] = g4[0]; pg[nt][r][1] = g4[H]; pg[nt][r][2] = g4[2 * H]; pg[nt][r][3] = g4[3 * H];
                        pct[nt][r] = first ? Cc[(size_t)t * B * H + row * H + u] : pcp[nt][r];
                        pcp[nt][r] = t ? Cc[(size_t)(t - 1) * B * H + row * H + u] : 0.f;
                        pdh[nt][r] = dHa[(size_t)t * B * H + row * H + u];
                    }
                }
        };
        fetch(T - 1, true);
        for (int t = T - 1; t >= 0; --t) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long row = r0 + 4 * kg + r; const int u = wave * UW + nt * 16 + col;
                    float di = 0.f, df = 0.f, dgg = 0.f, dob = 0.f;
                    if (FULL || row < B) {
                        const float i = pg[nt][r][0], f = pg[nt][r][1], gg = pg[nt][r][2], o = pg[nt][r][3];
                        const float ct = pct[nt][r], cp = pcp[nt][r];
                        const float dh = pdh[nt][r] + dhr[nt][r];
                        const float tc = tanhf(ct);
                        const float dcv = dh * o * (1.f - tc * tc) + dc[nt][r];
                        di = dcv * gg * i * (1.f - i); df = dcv * cp * f * (1.f - f); dgg = dcv * i * (1.f - gg * gg); dob = dh * tc * o * (1.f - o);
                        dc[nt][r] = dcv * f;
                        float* d4 = dG + (size_t)t * B * 4 * H + row * 4 * H + u;
                        d4[0] = di; d4[H] = df; d4[2 * H] = dgg; d4[3 * H] = dob;
                    }
                    float* ds = &dgs[4 * kg + r][u];
                    ds[0] = di; ds[H] = df; ds[2 * H] = dgg; ds[3 * H] = dob;
                }
            __syncthreads();
            if (t) {
                fetch(t - 1, false);
                floatx4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = floatx4{0.f, 0.f, 0.f, 0.f};
                float av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = dgs[col][4 * j + kg];
#pragma unroll
                for (int kq = 0; kq < KQ; kq += 4) {
                    float nv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (kq + 4 < KQ) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) nv[j] = dgs[col][4 * (kq + 4 + j) + kg];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], W[kq + j][nt], acc[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[j] = nv[j];
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dhr[nt][r] = acc[nt][r];
            }
            __syncthreads();
        }
    };
    static_assert(KQ % 4 == 0, "MFMA groups are taken in blocks of four");
    if (r0 + 16 <= B) run(std::true_type{}); else run(std::false_type{});
}
#ifndef LSTM_NW128
#define LSTM_NW128 8      /* waves per workgroup at H = 128: 8 = one 16-unit tile per wave, two waves per SIMD (4: 0.56 ms per 400-step layer pass, 8: see DESIGN.md section 4.2) */
#endif
static bool lstm_persistent_ok(int H, int T) {
    static const bool on = getenv("APX_LSTM_STEPWISE") == nullptr;      // A/B: the per-step launches of round 1
    // a launch re-loads W_hh into every workgroup's registers (256 KB at H = 128): worth it from a few time steps on; the one-step calls
    // of the rollout (169 us per layer call against ~50 us for accumulate-GEMM + gate kernel) stay on the per-step path
    return on && T >= 4 && (H == 128 || H == 64);
}

// x[T, B, D] prepared input; hc = [L][2][B][H] carried (h, c) in / out, or NULL (zero start, final state dropped);
// save = apx_lstm_workspace_floats(T, B, H, L) floats; y[T, B, O]
extern "C" int apx_lstm_forward(const float* params, int D, int H, int L, int O, const float* x, int T, int64_t B, float* hc,
                                float* save, float* y, void* stream) {
    APX_REQUIRE(params && x && save && y && T > 0 && B > 0 && L >= 1 && L <= 4 && D > 0 && H > 0 && O > 0, "lstm forward arguments");
    hipStream_t s = (hipStream_t)stream;
    const LstmView P(params, D, H, L, O);
    const long TB = (long)T * B;
    const float* in = x;
    for (int l = 0; l < L; ++l) {
        float* G = save + (size_t)l * TB * 6 * H; float* Cc = G + TB * 4 * H; float* Hh = Cc + TB * H;
        APX_TRY(linear_fwd(in, P.Wih[l], P.bih[l], G, TB, P.in[l], 4 * H, false, s));            // all time steps at once
        if (lstm_persistent_ok(H, T)) {      // the whole sequence of this layer in one launch, W_hh resident in registers
            float* hh = hc ? hc + (size_t)(2 * l) * B * H : nullptr; float* hcc = hc ? hc + (size_t)(2 * l + 1) * B * H : nullptr;
            if (H == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(lstm_seq_fwd_kernel<128, LSTM_NW128>), dim3(apx_cdiv(B, 16)), dim3(64 * LSTM_NW128), 0, s, G, P.Whh[l], P.bhh[l], hh, hcc, Cc, Hh, T, (long)B);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(lstm_seq_fwd_kernel<64, 4>), dim3(apx_cdiv(B, 16)), dim3(256), 0, s, G, P.Whh[l], P.bhh[l], hh, hcc, Cc, Hh, T, (long)B);
            APX_LAUNCH_CHECK();
            in = Hh;
            continue;
        }
        if (T == 1 && hc) {      // the rollout's one-step call: the gate kernel updates the carried (h, c) in place (element-wise: a thread reads c[i] and
            // writes c[i], and the recurrent GEMM that read h has finished on this stream), the next layer reads h from there: no copies
            float* hslot = hc + (size_t)(2 * l) * B * H; float* cslot = hc + (size_t)(2 * l + 1) * B * H;
            GemmArgs g{hslot, H, 1, P.Whh[l], 1, H, G, 4 * H, nullptr, 0, (int)B, 4 * H, H, 0, 0};
            APX_TRY(launch_gemm(EPI_ACC, g, 1, s));
            hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(apx_cdiv(B * H, 256)), dim3(256), 0, s, G, P.bhh[l], cslot, cslot, hslot, (long)B, H);
            APX_LAUNCH_CHECK();
            in = hslot;
            continue;
        }
        for (int t = 0; t < T; ++t) {
            const float* hp = t ? Hh + (size_t)(t - 1) * B * H : (hc ? hc + (size_t)(2 * l) * B * H : nullptr);
            const float* cp = t ? Cc + (size_t)(t - 1) * B * H : (hc ? hc + (size_t)(2 * l + 1) * B * H : nullptr);
            float* Gt = G + (size_t)t * B * 4 * H;
            if (hp) {       // G_t += h_{t-1} W_hh^T
                GemmArgs g{hp, H, 1, P.Whh[l], 1, H, Gt, 4 * H, nullptr, 0, (int)B, 4 * H, H, 0, 0};
                APX_TRY(launch_gemm(EPI_ACC, g, 1, s));
            }
            hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(apx_cdiv(B * H, 256)), dim3(256), 0, s, Gt, P.bhh[l], cp, Cc + (size_t)t * B * H,
                               Hh + (size_t)t * B * H, (long)B, H);
            APX_LAUNCH_CHECK();
        }
        if (hc) {
            APX_HIP(hipMemcpyAsync(hc + (size_t)(2 * l) * B * H, Hh + (size_t)(T - 1) * B * H, sizeof(float) * B * H, hipMemcpyDeviceToDevice, s));
            APX_HIP(hipMemcpyAsync(hc + (size_t)(2 * l + 1) * B * H, Cc + (size_t)(T - 1) * B * H, sizeof(float) * B * H, hipMemcpyDeviceToDevice, s));
        }
        in = Hh;
    }
    return linear_fwd(in, P.Wo, P.bo, y, TB, H, O, false, s);
}

// ------------------------------------------------------------------------------------------------ fused one-step recurrent pass (the rollout)
// PPO.sample's policy step for a recurrent actor / critic (rl/policies/actor.py:253-289 stepped once, rl/algos/ppo.py:160-184): input normalisation, two stacked
// LSTMCell(128) and the linear head in ONE launch.  As separate launches a step was 7 kernels (input GEMM, accumulate GEMM, gate kernel per layer, head GEMM) behind 4
// element-wise torch kernels (normalise, hidden-state reset, noise, add): ~0.2 ms of the 2.6 ms env step at 2048 envs.  A workgroup of eight waves carries 16 rows:
// the layer input [x | h_prev] (then [h1 | h2_prev]) sits in LDS, wave w owns units [16 w, 16 w + 16) of all four gates as v_mfma_f32_16x16x4_f32 accumulators (a lane
// holds i, f, g, o of the same (row, unit) pairs, so the gate arithmetic is lane-local), and the weights stream from L2 as float4: the k of a 16-deep tile is dealt
// k = 4 (lane >> 4) + q over the tile's four MFMAs q, so a lane's four B operands are 16 contiguous bytes of its row of the PACKED weight [W_ih | W_hh].
// apx_lstm_step_pack builds that packed block (W_ih padded to 64 columns, the two bias vectors summed) once per rollout.  Restrictions: L = 2, H = 128, D <= 64, O <= 16.
#define LS_H 128
#define LS_DP 64
struct LstmStepView {      // the packed block
    const float *W1, *b1, *W2, *b2, *Wo, *bo;
    __host__ __device__ LstmStepView(const float* p, int O) {
        W1 = p; p += (size_t)4 * LS_H * (LS_DP + LS_H); b1 = p; p += 4 * LS_H; W2 = p; p += (size_t)4 * LS_H * 2 * LS_H; b2 = p; p += 4 * LS_H; Wo = p; p += (size_t)O * LS_H; bo = p;
    }
};
extern "C" size_t apx_lstm_step_pack_floats(int D, int H, int L, int O) {
    if (H != LS_H || L != 2 || D > LS_DP || O > 16) return 0;
    return (size_t)4 * LS_H * (LS_DP + LS_H) + 4 * LS_H + (size_t)4 * LS_H * 2 * LS_H + 4 * LS_H + (size_t)O * LS_H + O;
}
__global__ void lstm_step_pack_kernel(const float* __restrict__ Wih0, const float* __restrict__ Whh0, const float* __restrict__ bih0, const float* __restrict__ bhh0,
                                      const float* __restrict__ Wih1, const float* __restrict__ Whh1, const float* __restrict__ bih1, const float* __restrict__ bhh1,
                                      const float* __restrict__ Wo, const float* __restrict__ bo, int D, int O, float* __restrict__ out) {
    constexpr int H = LS_H, K1 = LS_DP + LS_H, K2 = 2 * LS_H;
    const long n1 = (long)4 * H * K1, n2 = (long)4 * H * K2, total = n1 + 4 * H + n2 + 4 * H + (long)O * H + O;
    for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        float v;
        if (e < n1) { const int n = (int)(e / K1), k = (int)(e - (long)n * K1); v = k < LS_DP ? (k < D ? Wih0[(long)n * D + k] : 0.f) : Whh0[(long)n * H + (k - LS_DP)]; }
        else if (e < n1 + 4 * H) { const int n = (int)(e - n1); v = bih0[n] + bhh0[n]; }
        else if (e < n1 + 4 * H + n2) { const long f = e - n1 - 4 * H; const int n = (int)(f / K2), k = (int)(f - (long)n * K2); v = k < H ? Wih1[(long)n * H + k] : Whh1[(long)n * H + (k - H)]; }
        else if (e < n1 + 4 * H + n2 + 4 * H) { const int n = (int)(e - n1 - 4 * H - n2); v = bih1[n] + bhh1[n]; }
        else if (e < n1 + 4 * H + n2 + 4 * H + (long)O * H) v = Wo[e - (n1 + 4 * H + n2 + 4 * H)];
        else v = bo[e - (n1 + 4 * H + n2 + 4 * H + (long)O * H)];
        out[e] = v;
    }
}
extern "C" int apx_lstm_step_pack(const float* params, int D, int H, int L, int O, float* packed, void* stream) {
    APX_REQUIRE(params && packed && apx_lstm_step_pack_floats(D, H, L, O) > 0, "lstm step pack: L = 2, H = 128, D <= 64, O <= 16");
    const LstmView P(params, D, H, L, O);
    hipLaunchKernelGGL(lstm_step_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, P.Wih[0], P.Whh[0], P.bih[0], P.bhh[0], P.Wih[1], P.Whh[1], P.bih[1], P.bhh[1], P.Wo, P.bo, D, O, packed);
    APX_LAUNCH_CHECK();
    return APX_OK;
}
// one LSTM layer of the step for the workgroup's 16 rows: pre-activations over the K-wide LDS tile, gates, state update; h_new -> hn (LDS, row-major) and the carried state.
// The launch is a latency chain (128 workgroups, every one streams the whole packed weight out of L2): eight waves per workgroup (wave w owns units [16 w, 16 w + 16) of
// all four gates, two waves per SIMD) and the weight tiles fetched LS_PF k-tiles ahead of their MFMAs (four waves, one tile ahead: 67 us per 2048-row call).
#define LS_NW 8
#define LS_PF 3
template <int K>
__device__ __forceinline__ void lstm_step_layer(const float (*tile)[K + 4], const float* __restrict__ Wc, const float* __restrict__ bs, float* __restrict__ hslot, float* __restrict__ cslot,
                                                const bool (&live)[4], long r0, long B, float* hn, int hn_pitch) {
    constexpr int H = LS_H, NTILE = K / 16, UW = LS_H / LS_NW;
    static_assert(UW == 16, "a wave owns one 16-unit tile of every gate");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, col = lane & 15, kg = lane >> 4;
    typedef float f4w __attribute__((ext_vector_type(4)));
    floatx4 acc[4];
    const float* wrow[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = g * H + wave * UW + col;
        const float b = bs[n];
        acc[g] = floatx4{b, b, b, b};
        wrow[g] = Wc + (long)n * K + 4 * kg;
    }
    float cprev[4];                                     // the carried cell state of the lane's (row, unit) pairs: in flight across the MFMAs
#pragma unroll
    for (int r = 0; r < 4; ++r) { const long grow = r0 + 4 * kg + r; cprev[r] = (grow < B && live[r]) ? cslot[grow * H + wave * UW + col] : 0.f; }
    f4w wq[LS_PF + 1][4];
#pragma unroll
    for (int p = 0; p < LS_PF; ++p)
#pragma unroll
        for (int g = 0; g < 4; ++g) wq[p][g] = *reinterpret_cast<const f4w*>(wrow[g] + 16 * p);
#pragma unroll
    for (int t = 0; t < NTILE; ++t) {
        if (t + LS_PF < NTILE) {
#pragma unroll
            for (int g = 0; g < 4; ++g) wq[(t + LS_PF) % (LS_PF + 1)][g] = *reinterpret_cast<const f4w*>(wrow[g] + 16 * (t + LS_PF));
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the fetch LS_PF tiles ahead: left alone, the scheduler sinks every load to just above its MFMAs (vmcnt(1) waits) to save registers
        const f4w a = *reinterpret_cast<const f4w*>(&tile[col][16 * t + 4 * kg]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], wq[t % (LS_PF + 1)][g][q], acc[g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * kg + r, unit = wave * UW + col;
        const long grow = r0 + row;
        const float i = sigmf(acc[0][r]), f = sigmf(acc[1][r]), gg = tanhf(acc[2][r]), o = sigmf(acc[3][r]);
        const float c = f * cprev[r] + i * gg, h = o * tanhf(c);
        if (grow < B) { cslot[grow * H + unit] = c; hslot[grow * H + unit] = h; }
        hn[row * hn_pitch + unit] = h;
    }
}
__global__ __launch_bounds__(64 * LS_NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_step_fused_kernel(const float* __restrict__ packed, int D, int O, const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ stdv, const uint8_t* __restrict__ reset, float* __restrict__ hc, long B,
                                                              float* __restrict__ y, float* __restrict__ act, const float* __restrict__ noise, float sigma) {
    constexpr int H = LS_H, K1 = LS_DP + LS_H, K2 = 2 * LS_H, NTH = 64 * LS_NW;
    __shared__ __attribute__((aligned(16))) float t1[16][K1 + 4];      // [x (64, zero padded) | h1_prev]; later rows 0..15 x [0, 128) = h2 for the head
    __shared__ __attribute__((aligned(16))) float t2[16][K2 + 4];      // [h1 | h2_prev]
    const LstmStepView P(packed, O);
    const int tid = threadIdx.x, kg = (tid & 63) >> 4;
    const long r0 = (long)blockIdx.x * 16;
    float* h0 = hc; float* c0 = hc + (size_t)B * H; float* h1 = hc + (size_t)2 * B * H; float* c1 = hc + (size_t)3 * B * H;
    for (int e = tid; e < 16 * LS_DP; e += NTH) {
        const int r = e / LS_DP, k = e - r * LS_DP; const long row = r0 + r;
        float v = 0.f;
        if (row < B && k < D) { v = x[row * D + k]; if (mean) v = (v - mean[k]) / stdv[k]; }
        t1[r][k] = v;
    }
    for (int e = tid; e < 16 * H; e += NTH) {
        const int r = e / H, u = e - r * H; const long row = r0 + r;
        const bool z = row < B && !(reset && reset[row] != 0);      // init_hidden_state at an episode start (ppo.py:164-168): the carried state reads as zero
        t1[r][LS_DP + u] = z ? h0[row * H + u] : 0.f;
        t2[r][H + u] = z ? h1[row * H + u] : 0.f;
    }
    bool live[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const long row = r0 + 4 * kg + r; live[r] = row < B && !(reset && reset[row] != 0); }
    __syncthreads();
    lstm_step_layer<K1>(t1, P.W1, P.b1, h0, c0, live, r0, B, &t2[0][0], K2 + 4);
    __syncthreads();
    lstm_step_layer<K2>(t2, P.W2, P.b2, h1, c1, live, r0, B, &t1[0][0], K1 + 4);
    __syncthreads();
    // head: two lanes per (row, output), 64 k each as 16 float4 of W_o (all in flight at once) against the row of h2 in LDS, pair sum over DPP.  (One thread per output with
    // a 128-long loop was a chain of sixteen L2 round trips at the end of the launch.)
    {
        typedef float f4w __attribute__((ext_vector_type(4)));
        const int pair = tid >> 1, half = tid & 1;
        const bool on = pair < 16 * O;
        const int r = on ? pair / O : 0, o = on ? pair - r * O : 0; const long row = r0 + r;
        f4w w[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = *reinterpret_cast<const f4w*>(P.Wo + (size_t)o * H + 64 * half + 4 * q);
        float sacc = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f4w h = *reinterpret_cast<const f4w*>(&t1[r][64 * half + 4 * q]);
            sacc += h[0] * w[q][0] + h[1] * w[q][1] + h[2] * w[q][2] + h[3] * w[q][3];
        }
        sacc += __shfl_xor(sacc, 1);
        if (on && half == 0 && row < B) {
            sacc += P.bo[o];
            y[row * O + o] = sacc;
            if (act) act[row * O + o] = sacc + (noise ? sigma * noise[row * O + o] : 0.f);
        }
    }
}
extern "C" int apx_lstm_step(const float* packed, int D, int H, int L, int O, const float* x, const float* obs_mean, const float* obs_std, const uint8_t* reset, float* hc,
                             int64_t B, float* y, float* act, const float* noise, float sigma, void* stream) {
    APX_REQUIRE(packed && x && hc && y && B > 0 && apx_lstm_step_pack_floats(D, H, L, O) > 0, "lstm step: L = 2, H = 128, D <= 64, O <= 16, carried state required");
    APX_REQUIRE((obs_mean == nullptr) == (obs_std == nullptr), "obs_mean/obs_std");
    hipLaunchKernelGGL(lstm_step_fused_kernel, dim3(apx_cdiv(B, 16)), dim3(64 * LS_NW), 0, (hipStream_t)stream, packed, D, O, x, obs_mean, obs_std, reset, hc, (long)B, y, act, noise, sigma);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

// grads += d(loss)/d(params) for dy[T, B, O] (zero start state).  scratch: (8H + max(D, H)) * T * B ... see apx_lstm_bwd_scratch_floats
// (+ the K-chunk slabs of four weight gradients: two-stage split-K like the MLP path, one grad_reduce_kernel at the end instead of fp32 atomics into dW)
constexpr int LSTM_PART_GRADS = 4;
extern "C" size_t apx_lstm_bwd_scratch_floats(int T, int64_t B, int D, int H) { return (size_t)T * B * (4 * H + 2 * H) + (size_t)B * 2 * H + 4 * H + LSTM_PART_GRADS * GRAD_PART_FLOATS; }
extern "C" int apx_lstm_backward(const float* params, float* grads, int D, int H, int L, int O, const float* x, int T, int64_t B,
                                 const float* save, const float* dy, float* scratch, void* stream) {
    APX_REQUIRE(params && grads && x && save && dy && scratch && T > 0 && B > 0 && L >= 1 && L <= 4, "lstm backward arguments");
    hipStream_t s = (hipStream_t)stream;
    const LstmView P(params, D, H, L, O);
    const LstmView Gd(grads, D, H, L, O);
    const long TB = (long)T * B;
    float* dG = scratch;                         // [T, B, 4H]
    float* dHa = dG + TB * 4 * H;                // [T, B, H]  d(loss)/d(h_t) from the layer above (or the output layer)
    float* dHx = dHa + TB * H;                   // [T, B, H]  input gradient of the current layer = dHa of the layer below
    float* dhr = dHx + TB * H;                   // [B, H]     recurrent d(loss)/d(h_{t-1})
    float* dc = dhr + B * H;                     // [B, H]
    float* dbt = dc + B * H;                     // [4H]
    // W_ih / W_hh gradients: K = T B reductions.  As fp32 atomics into dW they were the longest launches of the recurrent minibatch (220 - 270 us each on the 2 B-column
    // actor pass, bound by the L2 atomic units); every K chunk stores its product to its own slab instead and one grad_reduce_kernel adds them up (128 x 128 tiles where
    // the shape allows: W_hh and the upper layers' W_ih are 512 x 128)
    static const bool use_parts = !(getenv("APX_LSTM_PARTS") && atoi(getenv("APX_LSTM_PARTS")) == 0);
    GradParts parts{dbt + 4 * H, LSTM_PART_GRADS * GRAD_PART_FLOATS, 0, {}, 0};
    GradParts* pp = use_parts ? &parts : nullptr;
    const float* Htop = save + (size_t)(L - 1) * TB * 6 * H + TB * 5 * H;
    APX_TRY(linear_bwd_weight(dy, Htop, const_cast<float*>(Gd.Wo), const_cast<float*>(Gd.bo), TB, H, O, s));
    APX_TRY(linear_bwd_input(dy, P.Wo, nullptr, dHa, TB, H, O, s));
    for (int l = L - 1; l >= 0; --l) {
        const float* G = save + (size_t)l * TB * 6 * H; const float* Cc = G + TB * 4 * H; const float* Hh = Cc + TB * H;
        const float* in = l ? save + (size_t)(l - 1) * TB * 6 * H + TB * 5 * H : x;
        APX_HIP(hipMemsetAsync(dc, 0, sizeof(float) * B * H, s));
        if (lstm_persistent_ok(H, T)) {
            if (H == 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(lstm_seq_bwd_kernel<128, LSTM_NW128>), dim3(apx_cdiv(B, 16)), dim3(64 * LSTM_NW128), 0, s, G, Cc, P.Whh[l], dHa, dG, T, (long)B);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(lstm_seq_bwd_kernel<64, 4>), dim3(apx_cdiv(B, 16)), dim3(256), 0, s, G, Cc, P.Whh[l], dHa, dG, T, (long)B);
            APX_LAUNCH_CHECK();
        } else
        for (int t = T - 1; t >= 0; --t) {
            hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(apx_cdiv(B * H, 256)), dim3(256), 0, s, G + (size_t)t * B * 4 * H, Cc + (size_t)t * B * H,
                               t ? Cc + (size_t)(t - 1) * B * H : nullptr, dHa + (size_t)t * B * H, t < T - 1 ? dhr : nullptr, dc,
                               dG + (size_t)t * B * 4 * H, (long)B, H);
            APX_LAUNCH_CHECK();
            if (t) APX_TRY(linear_bwd_input(dG + (size_t)t * B * 4 * H, P.Whh[l], nullptr, dhr, B, H, 4 * H, s));      // dh_{t-1} = dG_t W_hh
        }
        if (l) {      // the input gradient first: the layer below waits for it, the weight gradients of this layer do not
            APX_TRY(linear_bwd_input(dG, P.Wih[l], nullptr, dHx, TB, P.in[l], 4 * H, s));
            float* tmp = dHa; dHa = dHx; dHx = tmp;
        }
        APX_HIP(hipMemsetAsync(dbt, 0, sizeof(float) * 4 * H, s));
        APX_TRY(linear_bwd_weight(dG, in, const_cast<float*>(Gd.Wih[l]), dbt, TB, P.in[l], 4 * H, s, pp));
        hipLaunchKernelGGL(axpy_kernel, dim3(apx_cdiv(4 * H, 256)), dim3(256), 0, s, const_cast<float*>(Gd.bih[l]), dbt, (long)4 * H);
        hipLaunchKernelGGL(axpy_kernel, dim3(apx_cdiv(4 * H, 256)), dim3(256), 0, s, const_cast<float*>(Gd.bhh[l]), dbt, (long)4 * H);
        APX_LAUNCH_CHECK();
        if (T > 1)      // dW_hh += sum_{t >= 1} dG_t^T h_{t-1}
            APX_TRY(linear_bwd_weight(dG + (size_t)B * 4 * H, Hh, const_cast<float*>(Gd.Whh[l]), nullptr, (long)(T - 1) * B, H, 4 * H, s, pp));
    }
    return grad_reduce(parts, s);
}

// ------------------------------------------------------------------------------------------------ recurrent minibatch assembly
// The padded [T_max, B] batch of rl/algos/ppo.py:411-430 (pad_sequence over the sampled trajectories: zero rows behind a trajectory's end) gathered out of the rollout
// grid in ONE launch: raw observations (LSTM_V's input), normalised observations (old policy), [normalised | mirrored + normalised] side by side along the batch axis
// (pi(s) and pi(M s) as one 2 B-column pass), actions, returns, advantages and the 0 / 1 mask.  As torch ops this was 26 launches per minibatch (4 gathers, the masks,
// normalise, mirror with its asin / sin columns, cat): 0.45 ms of a 5.9 ms minibatch.  One wave per padded row, lane = column.
__global__ __launch_bounds__(256) void rec_gather_kernel(const int64_t* __restrict__ idx, const int64_t* __restrict__ traj, const int64_t* __restrict__ sel, long N, long rows, long B, int D, int A, const float* __restrict__ obs, const float* __restrict__ act,
                                                         const float* __restrict__ ret, const float* __restrict__ adv, const int32_t* __restrict__ sign_perm, uint64_t clock_mask,
                                                         const float* __restrict__ mean, const float* __restrict__ stdv, float* __restrict__ obs_raw, float* __restrict__ xn,
                                                         float* __restrict__ xa, float* __restrict__ act_p, float* __restrict__ ret_p, float* __restrict__ adv_p, float* __restrict__ mask) {
    const long r = blockIdx.x * 4l + (threadIdx.x >> 6);
    const int c = threadIdx.x & 63;
    if (r >= rows) return;
    const long t = r / B, b = r - t * B;
    long src;
    if (idx) src = idx[r];
    else {      // column b = trajectory sel[b] = (grid column n, t0, t1): its step t sits in grid row (t0 + t) N + n
        const int64_t* tr = traj + 3 * (sel ? sel[b] : b);
        src = t < tr[2] - tr[1] ? (tr[1] + t) * N + tr[0] : -1;
    }
    const bool valid = src >= 0;
    for (int k = c; k < D; k += 64) {
        const float v = valid ? obs[src * D + k] : 0.f;
        obs_raw[r * D + k] = v;
        const float n = (v - mean[k]) / stdv[k];
        xn[r * D + k] = n;
        if (xa) {
            xa[(t * 2 * B + b) * D + k] = n;
            const int32_t sp = sign_perm[k];
            float m = valid ? (sp >= 0 ? obs[src * D + sp] : -obs[src * D + (-sp - 1)]) : (sp >= 0 ? 0.f : -0.f);
            if (k < 64 && ((clock_mask >> k) & 1ull)) m = sinf(asinf(m) + 3.14159265358979323846f);      // wrappers.py:65-66 (a padded row: sin(pi), like the reference's mirrored zeros)
            xa[(t * 2 * B + B + b) * D + k] = (m - mean[k]) / stdv[k];
        }
    }
    for (int k = c; k < A; k += 64) act_p[r * A + k] = valid ? act[src * A + k] : 0.f;
    if (c == 0) { ret_p[r] = valid ? ret[src] : 0.f; adv_p[r] = valid ? adv[src] : 0.f; mask[r] = valid ? 1.f : 0.f; }
}
extern "C" int apx_rec_gather(const int64_t* idx, const int64_t* traj, const int64_t* sel, int64_t N, int T, int64_t B, int D, int A, const float* obs, const float* act, const float* ret, const float* adv,
                              const int32_t* obs_sign_perm, uint64_t clock_mask, const float* obs_mean, const float* obs_std, float* obs_raw, float* xn, float* xa,
                              float* act_p, float* ret_p, float* adv_p, float* mask, void* stream) {
    APX_REQUIRE((idx || (traj && N > 0)) && obs && act && ret && adv && obs_mean && obs_std && obs_raw && xn && act_p && ret_p && adv_p && mask && T > 0 && B > 0 && D > 0 && A > 0, "rec gather arguments");
    APX_REQUIRE((xa == nullptr) == (obs_sign_perm == nullptr), "xa (the [x | mirror(x)] batch) goes with obs_sign_perm");
    const long rows = (long)T * B;
    hipLaunchKernelGGL(rec_gather_kernel, dim3(apx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, idx, traj, sel, (long)N, rows, (long)B, D, A, obs, act, ret, adv, obs_sign_perm, clock_mask,
                       obs_mean, obs_std, obs_raw, xn, xa, act_p, ret_p, adv_p, mask);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

// ------------------------------------------------------------------------------------------------ PPO losses
// One thread per sample.  Produces d(loss)/d(mu) for the policy branch and the mirrored branch, d(loss)/d(v), and
// the six scalars of update_policy (sums; divided on the host side of the ABI by nothing: already means).
struct LossArgs {
    const float *mu, *mum, *v;            // [mb,A], [mb,A] or NULL, [mb]
    const float *act, *ret, *adv, *old_mu; const int64_t* idx;
    const float* mask;                    // [rows] 0/1 weights of the padded recurrent batch (ppo.py:293-299: actor and critic terms only), or NULL
    const int32_t* act_sp;                // mirror_action as signed permutation, or NULL
    float *dmu, *dmum, *dv;
    double* acc;                          // [8]: actor_loss, ratio, kl, mirror, critic_loss
    long mb; int A;
    float sd, clip, mirror_coeff;
};

__global__ __launch_bounds__(256) void ppo_loss_kernel(LossArgs L) {
    const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    if (b < L.mb) {
        const long row = L.idx ? L.idx[b] : b;
        const int A = L.A;
        const float inv_var = 1.f / (L.sd * L.sd);
        const float inv_mb = 1.f / (float)L.mb;
        float dlp = 0.f, klsum = 0.f;
        for (int j = 0; j < A; ++j) {
            const float a = L.act[row * A + j], m = L.mu[b * A + j], mo = L.old_mu[row * A + j];
            // log N(a; m, sd) - log N(a; mo, sd): the -log(sd) - log(sqrt(2 pi)) terms cancel exactly
            dlp += (-(a - m) * (a - m) + (a - mo) * (a - mo)) * (0.5f * inv_var);
            const float z = (m - mo) / L.sd;
            klsum += 0.5f * z * z;
        }
        const float ratio = expf(dlp);
        const float mk = L.mask ? L.mask[row] : 1.f;
        const float adv = L.adv[row] * mk;
        const float cpi = ratio * adv;
        const float lo = 1.f - L.clip, hi = 1.f + L.clip;
        const float rc = fminf(fmaxf(ratio, lo), hi);
        const float clp = rc * adv;
        acc[0] = -(double)fminf(cpi, clp) * inv_mb;
        acc[1] = (double)ratio * inv_mb;
        acc[2] = (double)klsum * inv_mb / A;
        // torch.min backward: ties split 1/2 + 1/2 (both branches carry the same derivative inside the clip range)
        const float w_cpi = cpi < clp ? 1.f : (cpi == clp ? 0.5f : 0.f);
        const float inside = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
        const float dsur = w_cpi * adv + (1.f - w_cpi) * adv * inside;
        const float dlogp = -(dsur * ratio) * inv_mb;
        float msum = 0.f;
        const float dscale = 2.f * L.mirror_coeff * inv_mb / A;
        for (int j = 0; j < A; ++j) {
            const float a = L.act[row * A + j], m = L.mu[b * A + j];
            float d = dlogp * (a - m) * inv_var;
            if (L.mum) {
                const int32_t sp = L.act_sp[j];
                const int src = sp >= 0 ? sp : -sp - 1;
                const float sg = sp >= 0 ? 1.f : -1.f;
                const float diff = m - sg * L.mum[b * A + src];
                msum += diff * diff;
                const float dd = dscale * diff;
                d += dd;
                L.dmum[b * A + src] = -sg * dd;
            }
            L.dmu[b * A + j] = d;
        }
        acc[3] = (double)L.mirror_coeff * msum * inv_mb / A;
        const float v = L.v[b], r = L.ret[row];
        acc[4] = 0.5 * (double)(r - v) * (double)(r - v) * inv_mb * mk * mk;
        L.dv[b] = -(r - v) * inv_mb * mk * mk;
    }
    block_atomic_add<5>(acc, L.acc);
}

__global__ void finish_scalars_kernel(const double* acc, float sd, double* out) {
    out[0] = acc[0];
    out[1] = 0.5 + 0.5 * log(2.0 * 3.14159265358979323846) + log((double)sd);
    out[2] = acc[4];
    out[3] = acc[1];
    out[4] = acc[2];
    out[5] = acc[3];
}

// The loss stage alone, for callers that run their own forward / backward (the recurrent path): rows = T * B entries of a padded batch
extern "C" int apx_ppo_loss(const float* mu, const float* mum, const float* v, const float* act, const float* ret, const float* adv,
                            const float* old_mu, const float* mask, const int32_t* act_sign_perm, int64_t rows, int A, float fixed_std,
                            float clip, float mirror_coeff, float* dmu, float* dmum, float* dv, double* scalars_out, double* acc_ws,
                            void* stream) {
    APX_REQUIRE(mu && v && act && ret && adv && old_mu && dmu && dv && scalars_out && acc_ws && rows > 0 && A > 0, "ppo loss arguments");
    APX_REQUIRE((mum == nullptr) == (act_sign_perm == nullptr) && (mum == nullptr) == (dmum == nullptr), "mirror branch: mum, dmum and act_sign_perm together");
    hipStream_t s = (hipStream_t)stream;
    APX_HIP(hipMemsetAsync(acc_ws, 0, 8 * sizeof(double), s));
    LossArgs L{mu, mum, v, act, ret, adv, old_mu, nullptr, mask, act_sign_perm, dmu, dmum, dv, acc_ws, (long)rows, A, fixed_std, clip, mirror_coeff};
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(apx_cdiv(rows, 256)), dim3(256), 0, s, L);
    APX_LAUNCH_CHECK();
    hipLaunchKernelGGL(finish_scalars_kernel, dim3(1), dim3(1), 0, s, acc_ws, fixed_std, scalars_out);
    APX_LAUNCH_CHECK();
    return APX_OK;
}

// ------------------------------------------------------------------------------------------------ clip + Adam
__global__ void sumsq_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
    double acc[1] = {0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc[0] += (double)g[i] * (double)g[i];
    block_atomic_add<1>(acc, out);
}

__global__ void clip_adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                 const float* __restrict__ g, int64_t n, const double* __restrict__ sumsq,
                                 float grad_scale, float grad_clip, float step_size, float inv_bc2_sqrt, float eps) {
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), applied when < 1
    const float total = (float)sqrt(*sumsq) * fabsf(grad_scale);
    float coef = grad_clip / (total + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
    const float sc = coef * grad_scale;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * sc;
        const float mi = 0.9f * m[i] + 0.1f * gi;
        const float vi = 0.999f * v[i] + 0.001f * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        p[i] -= step_size * (mi / denom);
    }
}

static int clip_adam_impl(float* param, float* m, float* v, float* grad, int64_t n, float grad_scale, float grad_clip, float lr, float adam_eps, int adam_t,
                          double* sumsq_scratch, hipStream_t s, bool clear) {
    if (clear) APX_HIP(hipMemsetAsync(sumsq_scratch, 0, sizeof(double), s));
    const int grid = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, grad, n, sumsq_scratch);
    APX_LAUNCH_CHECK();
    const double bc1 = 1.0 - pow(0.9, adam_t), bc2 = 1.0 - pow(0.999, adam_t);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(grid), dim3(256), 0, s, param, m, v, grad, n, sumsq_scratch, grad_scale,
                       grad_clip, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), adam_eps);
    APX_LAUNCH_CHECK();
    return APX_OK;
}
extern "C" int apx_clip_adam(float* param, float* m, float* v, float* grad, int64_t n, float grad_scale,
                             float grad_clip, float lr, float adam_eps, int adam_t, double* sumsq_scratch,
                             void* stream) {
    APX_REQUIRE(param && m && v && grad && sumsq_scratch && n > 0 && adam_t >= 1, "args");
    return clip_adam_impl(param, m, v, grad, n, grad_scale, grad_clip, lr, adam_eps, adam_t, sumsq_scratch, (hipStream_t)stream, true);
}

// ---- the small launches of a minibatch, merged (each launch is ~5 us of stream time whatever it does: 13 of them were a tenth of the minibatch)
// head: clears the loss accumulators and the flat gradient, gathers + normalises the minibatch rows (xn) and their mirrored copy (xm = xn + rows * D)
__global__ __launch_bounds__(256) void ppo_head_kernel(const float* __restrict__ x, int64_t rows, int D, const int64_t* __restrict__ idx, const int32_t* __restrict__ sign_perm,
                                                       uint64_t clock_mask, const float* __restrict__ mean, const float* __restrict__ stdv, float* __restrict__ xn,
                                                       float* __restrict__ grad, int64_t ngrad, float* __restrict__ acc32) {
    const int64_t e0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = e0; e < ngrad; e += stride) grad[e] = 0.f;
    if (e0 < 32) acc32[e0] = 0.f;
    const int64_t total = (sign_perm ? 2 : 1) * rows * D;
    for (int64_t e = e0; e < total; e += stride) {
        const bool mir = e >= rows * D;
        const int64_t f = mir ? e - rows * D : e, b = f / D;
        const int c = (int)(f - b * D);
        const int64_t row = idx ? idx[b] : b;
        float v;
        if (mir) {
            const int32_t sp = sign_perm[c];
            v = sp >= 0 ? x[row * D + sp] : -x[row * D + (-sp - 1)];
            if (c < 64 && ((clock_mask >> c) & 1ull)) v = sinf(asinf(v) + 3.14159265358979323846f);  // wrappers.py:65-66
        } else v = x[row * D + c];
        xn[e] = (v - mean[c]) / stdv[c];
    }
}
// tail: squared gradient norms of both networks in one launch (+ the loss scalars), then both clipped Adam steps in one launch
struct AdamSeg { float *p, *m, *v; const float* g; int64_t n; const double* sumsq; };
__global__ __launch_bounds__(256) void sumsq2_kernel(const float* __restrict__ ga, int64_t na, double* __restrict__ outa, const float* __restrict__ gc, int64_t nc,
                                                     double* __restrict__ outc, int blocks_a, const double* acc, float sd, double* scal) {
    const bool first = (int)blockIdx.x < blocks_a;
    const float* g = first ? ga : gc; const int64_t n = first ? na : nc;
    const int b = first ? blockIdx.x : blockIdx.x - blocks_a, nb = first ? blocks_a : gridDim.x - blocks_a;
    double a[1] = {0};
    for (int64_t i = b * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)nb * blockDim.x) a[0] += (double)g[i] * (double)g[i];
    block_atomic_add<1>(a, first ? outa : outc);
    if (scal && blockIdx.x == 0 && threadIdx.x == 0) {      // (finish_scalars_kernel)
        scal[0] = acc[0]; scal[1] = 0.5 + 0.5 * log(2.0 * 3.14159265358979323846) + log((double)sd); scal[2] = acc[4]; scal[3] = acc[1]; scal[4] = acc[2]; scal[5] = acc[3];
    }
}
__global__ __launch_bounds__(256) void clip_adam2_kernel(AdamSeg A, AdamSeg C, int blocks_a, float grad_clip, float step_size, float inv_bc2_sqrt, float eps) {
    const bool first = (int)blockIdx.x < blocks_a;
    const AdamSeg& S = first ? A : C;
    const int b = first ? blockIdx.x : blockIdx.x - blocks_a, nb = first ? blocks_a : gridDim.x - blocks_a;
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), applied when < 1
    const float total = (float)sqrt(*S.sumsq);
    float coef = grad_clip / (total + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
    for (int64_t i = b * (int64_t)blockDim.x + threadIdx.x; i < S.n; i += (int64_t)nb * blockDim.x) {
        const float gi = S.g[i] * coef;
        const float mi = 0.9f * S.m[i] + 0.1f * gi;
        const float vi = 0.999f * S.v[i] + 0.001f * gi * gi;
        S.m[i] = mi; S.v[i] = vi;
        const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
        S.p[i] -= step_size * (mi / denom);
    }
}

// ------------------------------------------------------------------------------------------------ PPO minibatch
static size_t align_up(size_t x) { return (x + 63) & ~(size_t)63; }

struct PpoWs {
    float *xn, *xm, *xr, *a1, *a2, *m1, *m2, *c1, *c2, *mu, *mum, *v, *dmu, *dmum, *dv, *dh2, *dh1, *parts;
    double* acc;
    size_t bytes;
    PpoWs(void* base, long mb, int D, int H, int A) {
        char* p = (char*)base;
        size_t off = 0;
        auto take = [&](size_t nfloat) { float* r = (float*)((uintptr_t)p + off);      /* (integer arithmetic: the size query runs this with a NULL base) */ off += align_up(nfloat * sizeof(float)); return r; };
        // the actor's two grad-carrying instances pi(s) and pi(M_s s) share the weights: their rows are ADJACENT ([2 mb, .] blocks: first s, then M_s s), so that
        // the forward is one 2 mb-row launch and the backward one chain of five GEMMs over 2 mb rows instead of two chains
        auto second = [](float* q, size_t nfloat) { return (float*)((uintptr_t)q + nfloat * sizeof(float)); };      // the M_s s half of a [2 mb, .] block
        xn = take(2 * mb * D); xm = second(xn, mb * D); xr = take(mb * D);
        a1 = take(2 * mb * H); m1 = second(a1, mb * H); a2 = take(2 * mb * H); m2 = second(a2, mb * H); c1 = take(mb * H); c2 = take(mb * H);
        mu = take(2 * mb * A); mum = second(mu, mb * A); v = take(mb);
        dmu = take(2 * mb * A); dmum = second(dmu, mb * A); dv = take(mb);
        dh2 = take(2 * mb * H); dh1 = take(2 * mb * H);
        parts = take(6 * GRAD_PART_FLOATS);      // K-chunk slabs of the six weight gradients of a minibatch (linear_bwd_weight)
        acc = (double*)((uintptr_t)p + off); off += align_up(16 * sizeof(double));
        bytes = off;
    }
};

extern "C" size_t apx_ppo_workspace_bytes(int64_t mb, int D, int H, int A) {
    PpoWs w(nullptr, mb, D, H, A);
    return w.bytes;
}

extern "C" int apx_ppo_minibatch(const apx_ppo_args* a, void* stream) {
    APX_REQUIRE(a, "args");
    APX_REQUIRE(a->actor && a->actor_grad && a->critic && a->critic_grad, "network pointers");
    APX_REQUIRE(a->grad_only || (a->actor_m && a->actor_v && a->critic_m && a->critic_v), "Adam state");
    APX_REQUIRE(a->obs && a->act && a->ret && a->adv && a->old_mu && a->obs_mean && a->obs_std, "batch pointers");
    APX_REQUIRE(a->mb > 0 && a->D > 0 && a->H > 0 && a->A > 0 && a->A <= 64, "dims");
    APX_REQUIRE(a->workspace && a->workspace_bytes >= apx_ppo_workspace_bytes(a->mb, a->D, a->H, a->A), "workspace");
    APX_REQUIRE(a->scalars_out, "scalars_out");
    APX_REQUIRE((a->obs_sign_perm == nullptr) == (a->act_sign_perm == nullptr), "mirror tables");
    APX_REQUIRE(a->grad_only >= 0 && a->grad_only <= 3, "grad_only: 0 full step, 1 gradients only, 2 / 3 gradients in two calls (actor half first)");
    hipStream_t s = (hipStream_t)stream;
    const long mb = a->mb;
    const int D = a->D, H = a->H, A = a->A;
    const bool mirror = a->obs_sign_perm != nullptr;
    PpoWs w(a->workspace, mb, D, H, A);
    const size_t na = apx_mlp_param_count(D, H, A), nc = apx_mlp_param_count(D, H, 1);
    if (a->grad_only == 3) {      // second call of the two-call gradient form: the critic's backward on the activations the first call left in the workspace
        GradParts parts{w.parts, 6 * GRAD_PART_FLOATS, 0, {}, 0};
        APX_TRY(mlp_backward_impl(a->critic, a->critic_grad, D, H, 1, w.xr, w.c1, w.c2, w.dv, mb, w.dh2, w.dh1, s, &parts));
        return grad_reduce(parts, s);
    }
    // forwards: pi(s) and pi(M_s s) as ONE pass over 2 mb rows
    const long ma = mirror ? 2 * mb : mb;
    if (a->critic_grad == a->actor_grad + na) {      // one flat gradient buffer (engine.PPOLearner): clears, gather and mirror in one launch
        hipLaunchKernelGGL(ppo_head_kernel, dim3(apx_cdiv(ma * D, 256)), dim3(256), 0, s, a->obs, (int64_t)mb, D, a->idx, a->obs_sign_perm, a->clock_mask, a->obs_mean, a->obs_std,
                           w.xn, a->actor_grad, (int64_t)(na + nc), (float*)w.acc);
        APX_LAUNCH_CHECK();
    } else {
        APX_HIP(hipMemsetAsync(w.acc, 0, 16 * sizeof(double), s));
        APX_HIP(hipMemsetAsync(a->actor_grad, 0, na * sizeof(float), s));
        APX_HIP(hipMemsetAsync(a->critic_grad, 0, nc * sizeof(float), s));
        APX_TRY(prep_obs(a->obs, mb, D, a->idx, nullptr, 0, a->obs_mean, a->obs_std, w.xn, s));
        if (mirror) APX_TRY(prep_obs(a->obs, mb, D, a->idx, a->obs_sign_perm, a->clock_mask, a->obs_mean, a->obs_std, w.xm, s));
    }
    APX_TRY(mlp_forward_impl(a->actor, D, H, A, w.xn, ma, w.a1, w.a2, w.mu, s));
    if (fused_ok(D, H, 1))      // critic: raw obs (critic.py:66); the row gather rides in the fused forward, which also leaves the gathered rows in w.xr for the backward
        APX_TRY(mlp_fused_launch(a->critic, D, H, 1, FusedIn{a->obs, a->idx, nullptr, 0, nullptr, nullptr, w.xr, nullptr, nullptr, 0.f}, mb, w.c1, w.c2, w.v, s));
    else {
        APX_TRY(prep_obs(a->obs, mb, D, a->idx, nullptr, 0, nullptr, nullptr, w.xr, s));
        APX_TRY(mlp_forward_impl(a->critic, D, H, 1, w.xr, mb, w.c1, w.c2, w.v, s));
    }
    // losses
    LossArgs L{w.mu, mirror ? w.mum : nullptr, w.v, a->act, a->ret, a->adv, a->old_mu, a->idx, nullptr, a->act_sign_perm,
               w.dmu, w.dmum, w.dv, w.acc, mb, A, a->fixed_std, a->clip, a->mirror_coeff};
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(apx_cdiv(mb, 64)), dim3(64), 0, s, L);
    APX_LAUNCH_CHECK();
    if (a->grad_only) { hipLaunchKernelGGL(finish_scalars_kernel, dim3(1), dim3(1), 0, s, w.acc, a->fixed_std, a->scalars_out); APX_LAUNCH_CHECK(); }      // (otherwise: in the tail)
    // backwards
    GradParts parts{w.parts, 6 * GRAD_PART_FLOATS, 0, {}, 0};
    APX_TRY(mlp_backward_impl(a->actor, a->actor_grad, D, H, A, w.xn, w.a1, w.a2, w.dmu, ma, w.dh2, w.dh1, s, &parts));      // both instances: 2 mb rows
    if (a->grad_only == 2) return grad_reduce(parts, s);      // the actor's gradient is final: the caller starts its all-reduce and comes back with grad_only = 3 for the critic's half
    APX_TRY(mlp_backward_impl(a->critic, a->critic_grad, D, H, 1, w.xr, w.c1, w.c2, w.dv, mb, w.dh2, w.dh1, s, &parts));
    APX_TRY(grad_reduce(parts, s));                     // the K-chunk slabs of the six weight gradients -> the flat gradient, one launch
    if (a->grad_only) return APX_OK;
    {   // (w.acc was cleared by the head: no second fill)
        const int ba = (int)((na + 255) / 256 < 384 ? (na + 255) / 256 : 384), bc = (int)((nc + 255) / 256 < 384 ? (nc + 255) / 256 : 384);
        hipLaunchKernelGGL(sumsq2_kernel, dim3(ba + bc), dim3(256), 0, s, a->actor_grad, (int64_t)na, w.acc + 8, a->critic_grad, (int64_t)nc, w.acc + 9, ba, w.acc, a->fixed_std, a->scalars_out);
        APX_LAUNCH_CHECK();
        const double bc1 = 1.0 - pow(0.9, a->adam_t), bc2 = 1.0 - pow(0.999, a->adam_t);
        hipLaunchKernelGGL(clip_adam2_kernel, dim3(ba + bc), dim3(256), 0, s, AdamSeg{a->actor, a->actor_m, a->actor_v, a->actor_grad, (int64_t)na, w.acc + 8},
                           AdamSeg{a->critic, a->critic_m, a->critic_v, a->critic_grad, (int64_t)nc, w.acc + 9}, ba, a->grad_clip, (float)(a->lr / bc1), (float)(1.0 / sqrt(bc2)), a->adam_eps);
        APX_LAUNCH_CHECK();
    }
    return APX_OK;
}
