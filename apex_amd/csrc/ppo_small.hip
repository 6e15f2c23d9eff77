// One EPOCH of small-minibatch PPO optimiser steps as ONE launch (gfx950).
//
//   for indices in BatchSampler(SubsetRandomSampler(range(B)), minibatch_size, drop_last=True):      rl/algos/ppo.py:414-417
//       scalars = self.update_policy(obs[indices], ...)                                              rl/algos/ppo.py:276-345, :355-356
//
// The reference's CLI default is minibatch_size = 64 (apex.py:242): one iteration of 131 072 samples is 3 x 2048 optimiser steps of 134 MFLOP each.  As launches
// (apx_ppo_minibatch: 16 per step) that regime is bound by launch latency: 156 us per step, 0.96 s per iteration (profiles/r05_bench_line_mb64.json).  Here the
// whole epoch is one persistent grid of G workgroups that walks through the steps with a grid-wide barrier between the phases of a step:
//
//   A  layer 0 of both networks (actor rows = [s ; M s], critic rows = s)                16 x 16 output tiles, one wave each
//   B  layer 1
//   C  output layer, PPO losses, d(loss)/d(pre-activation 2)                             one workgroup per (16 samples, 64 hidden columns); the K = 256 output
//                                                                                        layer is recomputed per column block (4 waves split K) instead of a barrier
//   D  d(loss)/d(pre-activation 1), output-layer weight / bias gradients
//   E  weight / bias gradients of layers 1 and 0, squared-norm partials
//   F  global-norm clip + Adam on both networks, the step's six scalars, input gather of the NEXT step
//
// Everything between two barriers is a list of independent 16 x 16 tiles on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: the parity mode of learner.hip), dealt
// round-robin over the grid's waves; operands go from L2 straight into the MFMA registers (all loads of a tile are issued before its first MFMA: one L2 round trip per
// tile).  Activations and gradients of a step are ~1 MB and never leave L2 / MALL.  No atomics in the data path: per-workgroup partial sums are stored and added in a
// fixed order (reruns are bit-identical).  Cross-XCD visibility between phases comes from the barrier's agent-scope release / acquire fences (buffer_wbl2 / buffer_inv).
#include "mlp_tiles.h"
#include <cmath>

namespace {
using namespace tiles;

constexpr int SH = 256;       // hidden width (the reference's 2 x 256 networks, apex.py:252 / actor.py:142)
constexpr int SXP = 64;       // padded input width: D <= 64, columns D.. are zero
constexpr int SYP = 16;       // padded output width: A <= 16, columns A.. are zero
constexpr int SMAXG = 128;    // upper bound of the grid

struct SmallNet {
    float *p, *m, *v, *g;     // parameters, Adam moments, flat gradient (layout of MlpView in learner.hip: W0 b0 W1 b1 W2 b2)
    float *X, *H1, *H2, *dZ2, *dZ1, *dY;      // [R, 64] padded input, [R, 256] activations / pre-activation gradients, [R, 16] padded output gradient
    int O, R, Rp;             // outputs, rows of a step, rows rounded up to a multiple of 64 (the rows R..Rp of every row buffer stay zero)
    long n;                   // parameter count
};

struct SmallArgs {
    SmallNet net[2];          // 0 actor, 1 critic
    int D, A, mb, nb, mirror;
    const float *obs, *act, *ret, *adv, *old_mu, *mean, *stdv;
    const int64_t* perm;
    const int32_t *osp, *asp; uint64_t cmask;
    float sd, clip, grad_clip, lr, eps, mirror_coeff;
    int adam_t0;
    float* tab;               // [nb, 2] step size lr / (1 - 0.9^t), 1 / sqrt(1 - 0.999^t)
    double* part;             // [G, 2] squared gradient norm partials of a step
    double* lossp;            // [mb / 16, 5] loss partials of a step
    unsigned* bar;            // barrier counter (zero at launch)
    double* scal;             // [nb, 6] out
};

// the normalised (actor) / raw (critic) input rows of step k, zero-padded to 64 columns: ppo_head_kernel's arithmetic (learner.hip), wrappers.py:59-67 for the mirror
__device__ __forceinline__ void gather_inputs(const SmallArgs& S, int k, int gtid, int gthreads) {
    const int Ra = S.net[0].R, Rc = S.net[1].R, D = S.D, mb = S.mb;
    const int64_t* idx = S.perm + (int64_t)k * mb;
    for (int e = gtid; e < (Ra + Rc) * SXP; e += gthreads) {
        int r = e / SXP; const int c = e - r * SXP;
        if (r < Ra) {
            const bool mir = r >= mb;
            const int64_t row = idx[mir ? r - mb : r];
            float o = 0.f;
            if (c < D) {
                float v;
                if (mir) {
                    const int32_t sp = S.osp[c];
                    v = sp >= 0 ? S.obs[row * D + sp] : -S.obs[row * D + (-sp - 1)];
                    if ((S.cmask >> c) & 1ull) v = sinf(asinf(v) + 3.14159265358979323846f);
                } else v = S.obs[row * D + c];
                o = (v - S.mean[c]) / S.stdv[c];
            }
            S.net[0].X[e] = o;
        } else {
            r -= Ra;
            const int64_t row = idx[r];
            S.net[1].X[r * SXP + c] = c < D ? S.obs[row * D + c] : 0.f;
        }
    }
}

}  // namespace

__global__ __launch_bounds__(256) void ppo_small_epoch_kernel(SmallArgs S) {
    __shared__ float Yp[4][3][16][17];      // phase C: per-wave K-quarter partials of the three output tiles
    __shared__ float Ys[3][16][17];         //          the outputs mu(s), v(s), mu(M s) of the block's 16 samples
    __shared__ float dYs[3][16][17];        //          d(loss)/d(output), zero-padded to 16 columns
    __shared__ double red[4][2];
    __shared__ float cst[4];

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int G = gridDim.x, gw = blockIdx.x * 4 + w, NW = G * 4;
    const int gtid = blockIdx.x * 256 + tid, gthreads = G * 256;
    const int D = S.D, A = S.A, mb = S.mb;
    const int Ra = S.net[0].R, Rc = S.net[1].R;
    const int nta = (Ra >> 4) * (SH >> 4), ntc = (Rc >> 4) * (SH >> 4);      // 16 x 16 tiles of an [R, 256] matrix
    unsigned bar_n = 0;
    auto barrier = [&]() { bar_n += 1; grid_barrier(S.bar, bar_n * (unsigned)G); };
    auto W0 = [&](const SmallNet& n) { return n.p; };
    auto B0 = [&](const SmallNet& n) { return n.p + (size_t)SH * D; };
    auto W1 = [&](const SmallNet& n) { return n.p + (size_t)SH * D + SH; };
    auto B1 = [&](const SmallNet& n) { return n.p + (size_t)SH * D + SH + (size_t)SH * SH; };
    auto W2 = [&](const SmallNet& n) { return n.p + (size_t)SH * D + SH + (size_t)SH * SH + SH; };
    auto B2 = [&](const SmallNet& n) { return n.p + (size_t)SH * D + SH + (size_t)SH * SH + SH + (size_t)n.O * SH; };
    const size_t oW0 = 0, oB0 = (size_t)SH * D, oW1 = oB0 + SH, oB1 = oW1 + (size_t)SH * SH, oW2 = oB1 + SH;      // the same offsets into the flat gradient

    // prologue: Adam's bias-correction table (torch.optim.Adam: step_size = lr / (1 - beta1^t), denom = sqrt(v) / sqrt(1 - beta2^t) + eps) and the inputs of step 0
    for (int i = gtid; i < S.nb; i += gthreads) {
        const double t = (double)(S.adam_t0 + i);
        const double bc1 = 1.0 - pow(0.9, t), bc2 = 1.0 - pow(0.999, t);
        S.tab[2 * i] = (float)((double)S.lr / bc1);
        S.tab[2 * i + 1] = (float)(1.0 / sqrt(bc2));
    }
    gather_inputs(S, 0, gtid, gthreads);
    barrier();

    for (int k = 0; k < S.nb; ++k) {
        const int64_t* idx = S.perm + (int64_t)k * mb;
        // ---------------------------------------------------------------- A: H1 = relu(X W0^T + b0)
        for (int t = gw; t < nta + ntc; t += NW) {
            const int ni = t >= nta; const SmallNet& n = S.net[ni];
            const int tt = ni ? t - nta : t, r0 = (tt >> 4) << 4, n0 = (tt & 15) << 4;
            const float* Xr = n.X + (size_t)(r0 + c) * SXP + 4 * g;
            const float* Wr = W0(n) + (size_t)(n0 + c) * D;
            const int lim = D - 4 * g;
            const floatx4 acc = wave_tile<4, false>(0, 4,
                [&](int kc, float (&a)[4]) { ld4(Xr + 16 * kc, a); },
                // columns D..63 of the padded contraction: the load stays inside the parameter block (what lies behind a W0 row is the next row, b0, W1) and the operand
                // is an exact zero by a select (no exec-mask region; 0 x a non-finite b0 / W1 entry would be NaN)
                [&](int kc, float (&b)[4]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float w = Wr[16 * kc + 4 * g + j]; b[j] = 16 * kc + j < lim ? w : 0.f; }
                }, nullptr);
            const float bias = B0(n)[n0 + c];
#pragma unroll
            for (int v = 0; v < 4; ++v) n.H1[(size_t)(r0 + 4 * g + v) * SH + n0 + c] = fmaxf(acc[v] + bias, 0.f);
        }
        barrier();
        // ---------------------------------------------------------------- B: H2 = relu(H1 W1^T + b1)
        for (int t = gw; t < nta + ntc; t += NW) {
            const int ni = t >= nta; const SmallNet& n = S.net[ni];
            const int tt = ni ? t - nta : t, r0 = (tt >> 4) << 4, n0 = (tt & 15) << 4;
            const float* Ar = n.H1 + (size_t)(r0 + c) * SH + 4 * g;
            const float* Wr = W1(n) + (size_t)(n0 + c) * SH + 4 * g;
            const floatx4 acc = wave_tile<16, false>(0, 16,
                [&](int kc, float (&a)[4]) { ld4(Ar + 16 * kc, a); },
                [&](int kc, float (&b)[4]) { ld4(Wr + 16 * kc, b); }, nullptr);
            const float bias = B1(n)[n0 + c];
#pragma unroll
            for (int v = 0; v < 4; ++v) n.H2[(size_t)(r0 + 4 * g + v) * SH + n0 + c] = fmaxf(acc[v] + bias, 0.f);
        }
        barrier();
        // ---------------------------------------------------------------- C: outputs, losses, dZ2 = (dY W2) * (H2 > 0)
        {
            const int ncb = SH / 64, nitem = (mb >> 4) * ncb, ntile = S.mirror ? 3 : 2;      // tile 0: actor rows of s, 1: critic rows, 2: actor rows of M s
            for (int item = blockIdx.x; item < nitem; item += G) {
                const int sb = item / ncb, cb = item - sb * ncb;
                for (int tt = 0; tt < ntile; ++tt) {
                    const SmallNet& n = S.net[tt == 1];
                    const int rbase = (tt == 2 ? mb : 0) + sb * 16;
                    const float* Ar = n.H2 + (size_t)(rbase + c) * SH + 4 * g;
                    const float* Wr = W2(n) + (size_t)c * SH + 4 * g;
                    const bool bok = c < n.O;
                    const floatx4 acc = wave_tile<4, false>(4 * w, 4 * w + 4,
                        [&](int kc, float (&a)[4]) { ld4(Ar + 16 * kc, a); },
                        [&](int kc, float (&b)[4]) { if (bok) ld4(Wr + 16 * kc, b); else zero4(b); }, nullptr);
#pragma unroll
                    for (int v = 0; v < 4; ++v) Yp[w][tt][4 * g + v][c] = acc[v];
                }
                __syncthreads();
                {
                    const int r = tid >> 4, o = tid & 15;
                    for (int tt = 0; tt < ntile; ++tt) {
                        const SmallNet& n = S.net[tt == 1];
                        float y = o < n.O ? B2(n)[o] : 0.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) y += Yp[q][tt][r][o];
                        Ys[tt][r][o] = y;
                        dYs[tt][r][o] = 0.f;
                    }
                }
                __syncthreads();
                double la[5] = {0, 0, 0, 0, 0};
                if (tid < 16) {      // ppo_loss_kernel's arithmetic (learner.hip), one sample per lane
                    const int r = tid;
                    const int64_t row = idx[sb * 16 + r];
                    const float inv_var = 1.f / (S.sd * S.sd), inv_mb = 1.f / (float)mb;
                    float dlp = 0.f, klsum = 0.f;
                    for (int j = 0; j < A; ++j) {
                        const float a = S.act[row * A + j], m = Ys[0][r][j], mo = S.old_mu[row * A + j];
                        dlp += (-(a - m) * (a - m) + (a - mo) * (a - mo)) * (0.5f * inv_var);
                        const float z = (m - mo) / S.sd;
                        klsum += 0.5f * z * z;
                    }
                    const float ratio = expf(dlp);
                    const float adv = S.adv[row];
                    const float cpi = ratio * adv;
                    const float lo = 1.f - S.clip, hi = 1.f + S.clip;
                    const float rc = fminf(fmaxf(ratio, lo), hi);
                    const float clp = rc * adv;
                    la[0] = -(double)fminf(cpi, clp) * inv_mb;
                    la[1] = (double)ratio * inv_mb;
                    la[2] = (double)klsum * inv_mb / A;
                    const float w_cpi = cpi < clp ? 1.f : (cpi == clp ? 0.5f : 0.f);
                    const float inside = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
                    const float dsur = w_cpi * adv + (1.f - w_cpi) * adv * inside;
                    const float dlogp = -(dsur * ratio) * inv_mb;
                    float msum = 0.f;
                    const float dscale = 2.f * S.mirror_coeff * inv_mb / A;
                    for (int j = 0; j < A; ++j) {
                        const float a = S.act[row * A + j], m = Ys[0][r][j];
                        float d = dlogp * (a - m) * inv_var;
                        if (S.mirror) {
                            const int32_t sp = S.asp[j];
                            const int src = sp >= 0 ? sp : -sp - 1;
                            const float sg = sp >= 0 ? 1.f : -1.f;
                            const float diff = m - sg * Ys[2][r][src];
                            msum += diff * diff;
                            const float dd = dscale * diff;
                            d += dd;
                            dYs[2][r][src] = -sg * dd;
                        }
                        dYs[0][r][j] = d;
                    }
                    la[3] = (double)S.mirror_coeff * msum * inv_mb / A;
                    const float vv = Ys[1][r][0], rr = S.ret[row];
                    la[4] = 0.5 * (double)(rr - vv) * (double)(rr - vv) * inv_mb;
                    dYs[1][r][0] = -(rr - vv) * inv_mb;
                }
                if (cb == 0 && w == 0) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) { const double s = wsum(la[q]); if (lane == 0) S.lossp[sb * 5 + q] = s; }
                }
                __syncthreads();
                if (cb == 0) {      // the output gradients of the block's rows, for the output-layer weight gradients of phase D
                    const int r = tid >> 4, o = tid & 15;
                    S.net[0].dY[(size_t)(sb * 16 + r) * SYP + o] = dYs[0][r][o];
                    S.net[1].dY[(size_t)(sb * 16 + r) * SYP + o] = dYs[1][r][o];
                    if (S.mirror) S.net[0].dY[(size_t)(mb + sb * 16 + r) * SYP + o] = dYs[2][r][o];
                }
                for (int ti = w; ti < ntile * 4; ti += 4) {
                    const int tt = ti >> 2, n0 = cb * 64 + (ti & 3) * 16;
                    const SmallNet& n = S.net[tt == 1];
                    const int rbase = (tt == 2 ? mb : 0) + sb * 16;
                    const float* Wc = W2(n) + n0 + c;
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kk = 4 * g + j;
                        const float a = dYs[tt][c][kk];
                        const float b = kk < n.O ? Wc[(size_t)kk * SH] : 0.f;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const size_t e = (size_t)(rbase + 4 * g + v) * SH + n0 + c;
                        n.dZ2[e] = n.H2[e] > 0.f ? acc[v] : 0.f;
                    }
                }
                __syncthreads();
            }
        }
        barrier();
        double ssq[2] = {0.0, 0.0};
        // ---------------------------------------------------------------- D: dZ1 = (dZ2 W1) * (H1 > 0);  dW2 = dY^T H2, db2 = column sums of dY
        {
            const int nd = nta + ntc, n5 = 2 * (SH >> 4);
            for (int t = gw; t < nd + n5; t += NW) {
                if (t < nd) {
                    const int ni = t >= nta; const SmallNet& n = S.net[ni];
                    const int tt = ni ? t - nta : t, r0 = (tt >> 4) << 4, k0 = (tt & 15) << 4;
                    const float* Ar = n.dZ2 + (size_t)(r0 + c) * SH + 4 * g;
                    const float* Wc = W1(n) + (size_t)(4 * g) * SH + k0 + c;
                    const floatx4 acc = wave_tile<16, false>(0, 16,
                        [&](int kc, float (&a)[4]) { ld4(Ar + 16 * kc, a); },
                        [&](int kc, float (&b)[4]) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) b[j] = Wc[(size_t)(16 * kc + j) * SH];
                        }, nullptr);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const size_t e = (size_t)(r0 + 4 * g + v) * SH + k0 + c;
                        n.dZ1[e] = n.H1[e] > 0.f ? acc[v] : 0.f;
                    }
                } else {
                    const int u = t - nd, ni = u >= (SH >> 4); const SmallNet& n = S.net[ni];
                    const int c0 = (ni ? u - (SH >> 4) : u) << 4;
                    float as;
                    const floatx4 acc = rows_tile(n.dY + (size_t)(4 * g) * SYP + c, SYP, n.H2 + (size_t)(4 * g) * SH + c0 + c, SH, n.Rp >> 4, &as);
                    float* gW2 = n.g + oW2;
                    double s2 = 0.0;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int o = 4 * g + v;
                        if (o < n.O) { gW2[(size_t)o * SH + c0 + c] = acc[v]; s2 += (double)acc[v] * (double)acc[v]; }
                    }
                    if (c0 == 0) {
                        as += __shfl_xor(as, 16, 64); as += __shfl_xor(as, 32, 64);
                        if (g == 0 && c < n.O) { n.g[oW2 + (size_t)n.O * SH + c] = as; s2 += (double)as * (double)as; }
                    }
                    ssq[ni] += s2;
                }
            }
        }
        barrier();
        // ---------------------------------------------------------------- E: dW1 = dZ2^T H1, db1;  dW0 = dZ1^T X, db0
        {
            const int n6 = (SH >> 4) * (SH >> 4), n8 = (SH >> 4) * (SXP >> 4);
            for (int t = gw; t < 2 * n6 + 2 * n8; t += NW) {
                const bool six = t < 2 * n6;
                const int u = six ? t : t - 2 * n6, per = six ? n6 : n8;
                const int ni = u >= per; const SmallNet& n = S.net[ni];
                const int tt = ni ? u - per : u;
                const int n0 = (six ? tt >> 4 : tt >> 2) << 4, k0 = (six ? tt & 15 : tt & 3) << 4;
                const int ldb = six ? SH : SXP;
                float as;
                const floatx4 acc = rows_tile((six ? n.dZ2 : n.dZ1) + (size_t)(4 * g) * SH + n0 + c, SH, (six ? n.H1 : n.X) + (size_t)(4 * g) * ldb + k0 + c, ldb, n.Rp >> 4, &as);
                double s2 = 0.0;
                if (six) {
                    float* gW = n.g + oW1;
#pragma unroll
                    for (int v = 0; v < 4; ++v) { gW[(size_t)(n0 + 4 * g + v) * SH + k0 + c] = acc[v]; s2 += (double)acc[v] * (double)acc[v]; }
                } else if (k0 + c < D) {
                    float* gW = n.g + oW0;
#pragma unroll
                    for (int v = 0; v < 4; ++v) { gW[(size_t)(n0 + 4 * g + v) * D + k0 + c] = acc[v]; s2 += (double)acc[v] * (double)acc[v]; }
                }
                if (k0 == 0) {      // the bias gradient of the tile's 16 units: column sums of dZ over the rows
                    as += __shfl_xor(as, 16, 64); as += __shfl_xor(as, 32, 64);
                    if (g == 0) { n.g[(six ? oB1 : oB0) + n0 + c] = as; s2 += (double)as * (double)as; }
                }
                ssq[ni] += s2;
            }
        }
        {   // the workgroup's share of both squared gradient norms (phases D and E), stored: added in workgroup order in phase F
            const double sa = wsum(ssq[0]), sc = wsum(ssq[1]);
            if (lane == 0) { red[w][0] = sa; red[w][1] = sc; }
            __syncthreads();
            if (tid < 2) S.part[blockIdx.x * 2 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        }
        barrier();
        // ---------------------------------------------------------------- F: clip_grad_norm_ + Adam (ppo.py:322-336, :355-356), scalars, next step's inputs
        {
            if (w == 0) {
                double sa = 0.0, sc = 0.0;
                for (int i = lane; i < G; i += 64) { sa += ld_agent(S.part + 2 * i); sc += ld_agent(S.part + 2 * i + 1); }
                sa = wsum(sa); sc = wsum(sc);
                if (lane == 0) {
                    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), applied when < 1
                    float ca = S.grad_clip / ((float)sqrt(sa) + 1e-6f), cc = S.grad_clip / ((float)sqrt(sc) + 1e-6f);
                    cst[0] = ca < 1.f ? ca : 1.f; cst[1] = cc < 1.f ? cc : 1.f;
                    cst[2] = ld_agent(S.tab + 2 * k); cst[3] = ld_agent(S.tab + 2 * k + 1);
                }
            }
            __syncthreads();
            const float step_size = cst[2], inv_bc2_sqrt = cst[3];
            for (int ni = 0; ni < 2; ++ni) {
                const SmallNet& n = S.net[ni];
                const float coef = cst[ni];
                for (long i = gtid; i < n.n; i += gthreads) {
                    const float gi = n.g[i] * coef;
                    const float mi = 0.9f * n.m[i] + 0.1f * gi;
                    const float vi = 0.999f * n.v[i] + 0.001f * gi * gi;
                    n.m[i] = mi; n.v[i] = vi;
                    const float denom = sqrtf(vi) * inv_bc2_sqrt + S.eps;
                    n.p[i] -= step_size * (mi / denom);
                }
            }
            if (blockIdx.x == 0 && tid == 0) {      // update_policy's return tuple (ppo.py:338-345): sums of the 16-sample blocks in block order
                double a[5] = {0, 0, 0, 0, 0};
                for (int sb = 0; sb < (mb >> 4); ++sb)
                    for (int q = 0; q < 5; ++q) a[q] += ld_agent(S.lossp + sb * 5 + q);
                double* out = S.scal + (size_t)k * 6;
                out[0] = a[0]; out[1] = 0.5 + 0.5 * log(2.0 * 3.14159265358979323846) + log((double)S.sd); out[2] = a[4]; out[3] = a[1]; out[4] = a[2]; out[5] = a[3];
            }
            if (k + 1 < S.nb) gather_inputs(S, k + 1, gtid, gthreads);
        }
        barrier();
    }
    if (blockIdx.x == 0 && tid == 0 && ld_agent(S.bar + 1) != 0u)      // the watchdog fired: nothing of this epoch is valid (PPO.update raises on non-finite losses)
        for (int i = 0; i < S.nb * 6; ++i) S.scal[i] = __builtin_nan("");
}

// ---------------------------------------------------------------------------------------------------------------------- host side
namespace {
size_t up64(size_t x) { return (x + 63) & ~(size_t)63; }
long rpad(long r) { return (r + 63) / 64 * 64; }
struct EpochWs {
    float *X[2], *H1[2], *H2[2], *dZ2[2], *dZ1[2], *dY[2], *tab;
    double *part, *lossp; unsigned* bar;
    size_t bytes;
    EpochWs(void* base, long mb, long nb, bool mirror) {
        char* p = (char*)base; size_t off = 0;
        auto take = [&](size_t nbytes) { char* r = (char*)((uintptr_t)p + off); off += up64(nbytes); return r;      /* (integer arithmetic: the size query runs this with a NULL base) */ };
        const long R[2] = {rpad(mirror ? 2 * mb : mb), rpad(mb)};
        for (int i = 0; i < 2; ++i) {
            X[i] = (float*)take(R[i] * SXP * 4);
            H1[i] = (float*)take(R[i] * SH * 4); H2[i] = (float*)take(R[i] * SH * 4);
            dZ2[i] = (float*)take(R[i] * SH * 4); dZ1[i] = (float*)take(R[i] * SH * 4);
            dY[i] = (float*)take(R[i] * SYP * 4);
        }
        tab = (float*)take(nb * 2 * 4);
        part = (double*)take(SMAXG * 2 * 8);
        lossp = (double*)take((mb / 16) * 5 * 8);
        bar = (unsigned*)take(64);
        bytes = off;
    }
};
bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
}  // namespace

extern "C" int apx_ppo_epoch_supported(int64_t mb, int D, int H, int A) {
    return H == SH && D > 0 && D <= SXP && A > 0 && A <= SYP && mb >= 16 && mb <= 1024 && mb % 16 == 0;
}

extern "C" size_t apx_ppo_epoch_workspace_bytes(int64_t mb, int64_t nb, int D, int H, int A) {
    if (!apx_ppo_epoch_supported(mb, D, H, A) || nb <= 0) return 0;
    EpochWs w(nullptr, mb, nb, true);
    return w.bytes;
}

extern "C" int apx_ppo_epoch(const apx_ppo_args* a, const int64_t* perm, int64_t nb, void* stream) {
    APX_REQUIRE(a && perm && nb > 0 && nb < (1 << 22), "args");
    APX_REQUIRE(a->actor && a->actor_grad && a->critic && a->critic_grad && a->actor_m && a->actor_v && a->critic_m && a->critic_v, "network pointers");
    APX_REQUIRE(a->obs && a->act && a->ret && a->adv && a->old_mu && a->obs_mean && a->obs_std, "batch pointers");
    APX_REQUIRE(apx_ppo_epoch_supported(a->mb, a->D, a->H, a->A), "shape: H = 256, D <= 64, A <= 16, mb a multiple of 16 in 16..1024 (otherwise: apx_ppo_minibatch per step)");
    APX_REQUIRE(a->idx == nullptr && a->grad_only == 0, "idx / grad_only belong to apx_ppo_minibatch");
    APX_REQUIRE((a->obs_sign_perm == nullptr) == (a->act_sign_perm == nullptr), "mirror tables");
    APX_REQUIRE(a->adam_t >= 1 && a->scalars_out, "adam_t / scalars_out");
    APX_REQUIRE(aligned16(a->actor) && aligned16(a->critic) && aligned16(a->workspace), "16-byte aligned parameter blocks and workspace");
    const bool mirror = a->obs_sign_perm != nullptr;
    APX_REQUIRE(a->workspace && a->workspace_bytes >= apx_ppo_epoch_workspace_bytes(a->mb, nb, a->D, a->H, a->A), "workspace (apx_ppo_epoch_workspace_bytes)");
    hipStream_t s = (hipStream_t)stream;
    EpochWs w(a->workspace, a->mb, nb, mirror);
    SmallArgs S;
    float* P[2] = {a->actor, a->critic}; float* M[2] = {a->actor_m, a->critic_m}; float* V[2] = {a->actor_v, a->critic_v}; float* Gr[2] = {a->actor_grad, a->critic_grad};
    for (int i = 0; i < 2; ++i) {
        SmallNet& n = S.net[i];
        n.p = P[i]; n.m = M[i]; n.v = V[i]; n.g = Gr[i];
        n.X = w.X[i]; n.H1 = w.H1[i]; n.H2 = w.H2[i]; n.dZ2 = w.dZ2[i]; n.dZ1 = w.dZ1[i]; n.dY = w.dY[i];
        n.O = i ? 1 : a->A; n.R = (int)(i == 0 && mirror ? 2 * a->mb : a->mb); n.Rp = (int)rpad(n.R);
        n.n = (long)SH * a->D + SH + (long)SH * SH + SH + (long)n.O * SH + n.O;
    }
    S.D = a->D; S.A = a->A; S.mb = (int)a->mb; S.nb = (int)nb; S.mirror = mirror ? 1 : 0;
    S.obs = a->obs; S.act = a->act; S.ret = a->ret; S.adv = a->adv; S.old_mu = a->old_mu; S.mean = a->obs_mean; S.stdv = a->obs_std;
    S.perm = perm; S.osp = a->obs_sign_perm; S.asp = a->act_sign_perm; S.cmask = a->clock_mask;
    S.sd = a->fixed_std; S.clip = a->clip; S.grad_clip = a->grad_clip; S.lr = a->lr; S.eps = a->adam_eps; S.mirror_coeff = a->mirror_coeff;
    S.adam_t0 = a->adam_t;
    S.tab = w.tab; S.part = w.part; S.lossp = w.lossp; S.bar = w.bar; S.scal = a->scalars_out;
    // grid: every workgroup must be resident at once (the barrier spins): far below the 256 CUs.  One wave per 16 x 16 tile of the layer-1 phases when that fits
    const long tiles = (long)(S.net[0].R + S.net[1].R) / 16 * (SH / 16);
    static const int forced = getenv("APX_PPO_EPOCH_WGS") ? atoi(getenv("APX_PPO_EPOCH_WGS")) : 0;
    int G = tiles <= 256 ? 64 : SMAXG;
    if (forced >= 1 && forced <= SMAXG) G = forced;
    APX_HIP(hipMemsetAsync(a->workspace, 0, w.bytes, s));      // the barrier counter and the padding rows of the row buffers
    return tiles::launch_resident(ppo_small_epoch_kernel, G, 256, s, S, "apx_ppo_epoch");
}
