// A block of TD3 updates as ONE launch (gfx950): `for it in range(iterations): TD3.train's loop body` of rl/algos/sync_td3.py:133-209 on batches drawn from the replay
// buffer in HBM - target-policy smoothing, clipped double-Q target, both critic regressions + Adam, and every policy_freq-th iteration the actor step on
// -Q1(s, pi(s)), its Adam and the Polyak averaging of both target networks.
//
// As launches (apex_amd/engine.py::TD3Learner.train_step) an update is ~60 launches = 383 us at batch 1024 (profiles/r05_td3_kernel_stats.txt: the update block is 49 ms of
// a 135 ms iteration).  Here one persistent grid walks through the U updates with a grid-wide barrier between the phases of an update (10 barriers for a critic-only
// update, 18 with the actor step); everything between two barriers is a list of independent 16 x 16 fp32 MFMA tiles dealt over the grid's waves, or of 16-row items on
// one workgroup each (output layers: 4 waves split K = 256; losses; the rank-one d(loss)/d(pre-activation 2) of the scalar-output critics).  Same building blocks
// as ppo_small.hip (mlp_tiles.h).  Adam has no gradient clipping in TD3, so it runs in the epilogue of the weight-gradient tile that produced the gradient - and the
// Polyak average of that weight right behind it - with no pass over the parameters and no gradient buffer at all.
#include "mlp_tiles.h"
#include <cmath>

namespace {
using namespace tiles;

constexpr int TH = 256;        // hidden width (the reference's 2 x 256 networks)
constexpr int TXP = 64;        // padded input width: D + A <= 64, columns beyond the input are zero
constexpr int TYP = 16;        // padded output width
constexpr int TMAXG = 128;
enum { AT = 0, Q1T, Q2T, Q1, Q2, PA, QA, NPASS };      // actor_target(s'), Q1/Q2_target(s', a'), Q1/Q2(s, a), actor(s), Q1(s, pi(s))

struct Pass {                  // one pass of a 3-layer ReLU MLP over the rows of an update
    float* p;                  // parameter block W0 [H, D] b0 W1 [H, H] b1 W2 [O, H] b2
    float *m, *v, *tp;         // Adam moments and the Polyak target of the block (NULL: not optimised in this pass)
    float *X, *H1, *H2, *dZ2, *dZ1, *dY, *Y;      // [Rp, 64] input, [Rp, 256] activations / pre-activation gradients, [Rp, 16] output gradient / output
    int D, O, al;              // input width, outputs, parameter block 16-byte aligned (Q2 sits at an odd float offset of the flat critic block)
};

struct TArgs {
    Pass P[NPASS];
    int D, A, B, Rp, U, it0, policy_freq;
    const float *rs, *rs2, *ra, *rr, *rnd;      // replay: state, next state [cap, D], action [cap, A], reward, not-done [cap]
    const int64_t* ind;        // [U, B] replay rows of every update
    const float* noise;        // [U, B, A] ~ N(0, policy_noise), clamped here (sync_td3.py:151)
    float max_action, noise_clip, discount, tau, a_lr, c_lr, eps;
    int t_a0, t_c0;            // Adam steps already taken by the two optimisers
    float *rb, *ndb;           // [Rp] reward / not-done of the update's rows
    float* tab;                // [U, 4] step size and 1 / sqrt(bias correction 2) of the critic's and the actor's Adam step of update u
    double* part;              // [B / 16, 4] per 16-row block: critic loss, sum q1, sum q2, -sum Q1(s, pi(s)) / B
    unsigned* bar;
    double* stats;             // [U, 4] out: critic loss, sum q1, sum q2, actor loss (0 on updates without the actor step)
};

}  // namespace

__global__ __launch_bounds__(256) void td3_small_kernel(TArgs S) {
    __shared__ float Yp[4][4][16][17];      // per-wave K-quarter partials of up to four output tiles of a 16-row item
    __shared__ float Ys[4][16][17];
    __shared__ float rowv[2][16];

    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int G = gridDim.x, gw = blockIdx.x * 4 + w, NW = G * 4;
    const int gtid = blockIdx.x * 256 + tid, gthreads = G * 256;
    const int D = S.D, A = S.A, B = S.B, RB = B >> 4;
    unsigned bar_n = 0;
    auto barrier = [&]() { bar_n += 1; grid_barrier(S.bar, bar_n * (unsigned)G); };
    auto oB0 = [&](const Pass& n) { return (size_t)TH * n.D; };
    auto oW1 = [&](const Pass& n) { return (size_t)TH * n.D + TH; };
    auto oB1 = [&](const Pass& n) { return (size_t)TH * n.D + TH + (size_t)TH * TH; };
    auto oW2 = [&](const Pass& n) { return (size_t)TH * n.D + TH + (size_t)TH * TH + TH; };
    auto oB2 = [&](const Pass& n) { return (size_t)TH * n.D + TH + (size_t)TH * TH + TH + (size_t)n.O * TH; };
    auto row4 = [&](const float* q, bool al, float (&o)[4]) {      // four consecutive floats of a parameter row
        if (al) ld4(q, o);
        else { o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3]; }
    };

    // ---- phases ------------------------------------------------------------------------------------------------------------------------------------------------
    auto gather = [&](int u) {      // the rows of update u out of the replay buffer (remote_replay.py:78-90), zero-padded to 64 columns
        const int64_t* idx = S.ind + (int64_t)u * B;
        for (int e = gtid; e < B * TXP; e += gthreads) {
            const int r = e / TXP, col = e - r * TXP;
            const int64_t row = idx[r];
            const float s = col < D ? S.rs[row * D + col] : 0.f, s2 = col < D ? S.rs2[row * D + col] : 0.f;
            S.P[AT].X[e] = s2;
            if (col < D) { S.P[Q1T].X[e] = s2; S.P[QA].X[e] = s; }      // (the action columns of these two are written by the output phase, the rest stays zero)
            S.P[Q1].X[e] = col < D ? s : (col < D + A ? S.ra[row * A + col - D] : 0.f);
            S.P[PA].X[e] = s;
            if (col == 0) { S.rb[r] = S.rr[row]; S.ndb[r] = S.rnd[row]; }
        }
    };
    auto fwd_tiles = [&](int layer, const int* ps, int np) {      // H1 = relu(X W0^T + b0) / H2 = relu(H1 W1^T + b1) of the listed passes
        const int per = RB * (TH >> 4);
        for (int t = gw; t < np * per; t += NW) {
            const int pi = t / per, tt = t - pi * per;
            const Pass& n = S.P[ps[pi]];
            const int r0 = (tt >> 4) << 4, n0 = (tt & 15) << 4;
            floatx4 acc; float bias;
            if (layer == 0) {
                const float* Xr = n.X + (size_t)(r0 + c) * TXP + 4 * g;
                const int lim = n.D - 4 * g;
                const float* Wr = n.p + (size_t)(n0 + c) * n.D;
                acc = wave_tile<4, false>(0, 4,
                    [&](int kc, float (&a)[4]) { ld4(Xr + 16 * kc, a); },
                    [&](int kc, float (&b)[4]) {      // columns D.. of the padded contraction: the load stays inside the parameter block (next row, b0, W1), the operand is an exact zero (see ppo_small.hip)
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const float w = Wr[16 * kc + 4 * g + j]; b[j] = 16 * kc + j < lim ? w : 0.f; }
                    }, nullptr);
                bias = n.p[oB0(n) + n0 + c];
            } else {
                const float* Ar = n.H1 + (size_t)(r0 + c) * TH + 4 * g;
                const float* Wr = n.p + oW1(n) + (size_t)(n0 + c) * TH + 4 * g;
                const bool al = n.al != 0;
                acc = wave_tile<16, false>(0, 16,
                    [&](int kc, float (&a)[4]) { ld4(Ar + 16 * kc, a); },
                    [&](int kc, float (&b)[4]) { row4(Wr + 16 * kc, al, b); }, nullptr);
                bias = n.p[oB1(n) + n0 + c];
            }
            float* Hout = layer == 0 ? n.H1 : n.H2;
#pragma unroll
            for (int v = 0; v < 4; ++v) Hout[(size_t)(r0 + 4 * g + v) * TH + n0 + c] = fmaxf(acc[v] + bias, 0.f);
        }
    };
    // the output layers of the listed passes for one 16-row block: Ys[i][r][o] = (H2 W2^T + b2)[rb 16 + r][o]; 4 waves split K = 256
    auto out_block = [&](int rb, const int* ps, int np) {
        for (int i = 0; i < np; ++i) {
            const Pass& n = S.P[ps[i]];
            const float* Ar = n.H2 + (size_t)(rb * 16 + c) * TH + 4 * g;
            const float* Wr = n.p + oW2(n) + (size_t)c * TH + 4 * g;
            const bool bok = c < n.O, al = n.al != 0;
            const floatx4 acc = wave_tile<4, false>(4 * w, 4 * w + 4,
                [&](int kc, float (&a)[4]) { ld4(Ar + 16 * kc, a); },
                [&](int kc, float (&b)[4]) { if (bok) row4(Wr + 16 * kc, al, b); else zero4(b); }, nullptr);
#pragma unroll
            for (int v = 0; v < 4; ++v) Yp[w][i][4 * g + v][c] = acc[v];
        }
        __syncthreads();
        const int r = tid >> 4, o = tid & 15;
        for (int i = 0; i < np; ++i) {
            const Pass& n = S.P[ps[i]];
            float y = o < n.O ? n.p[oB2(n) + o] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) y += Yp[q][i][r][o];
            Ys[i][r][o] = y;
        }
        __syncthreads();
    };
    // dZ2 = dy W2 * (H2 > 0) of a scalar-output pass for one 16-row block, dy[r] in rowv[slot]
    auto dz2_rank_one = [&](const Pass& n, int rb, int slot) {
        const float* W2 = n.p + oW2(n);
        for (int e = tid; e < 16 * TH; e += 256) {
            const int r = e >> 8, col = e & (TH - 1);
            const size_t at = (size_t)(rb * 16 + r) * TH + col;
            n.dZ2[at] = n.H2[at] > 0.f ? rowv[slot][r] * W2[col] : 0.f;
        }
    };
    auto dz1_tiles = [&](const int* ps, int np) {      // dZ1 = (dZ2 W1) * (H1 > 0)
        const int per = RB * (TH >> 4);
        for (int t = gw; t < np * per; t += NW) {
            const int pi = t / per, tt = t - pi * per;
            const Pass& n = S.P[ps[pi]];
            const int r0 = (tt >> 4) << 4, k0 = (tt & 15) << 4;
            const float* Ar = n.dZ2 + (size_t)(r0 + c) * TH + 4 * g;
            const float* Wc = n.p + oW1(n) + (size_t)(4 * g) * TH + k0 + c;
            const floatx4 acc = wave_tile<16, false>(0, 16,
                [&](int kc, float (&a)[4]) { ld4(Ar + 16 * kc, a); },
                [&](int kc, float (&b)[4]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j] = Wc[(size_t)(16 * kc + j) * TH];
                }, nullptr);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const size_t e = (size_t)(r0 + 4 * g + v) * TH + k0 + c;
                n.dZ1[e] = n.H1[e] > 0.f ? acc[v] : 0.f;
            }
        }
    };
    // one Adam step of one parameter (torch.optim.Adam, no weight decay), then the Polyak average of its target (sync_td3.py:196-202) when asked for
    auto adam = [&](const Pass& n, size_t off, float gr, float step_size, float inv_bc2_sqrt, bool polyak) {
        const float mi = 0.9f * n.m[off] + 0.1f * gr;
        const float vi = 0.999f * n.v[off] + 0.001f * gr * gr;
        n.m[off] = mi; n.v[off] = vi;
        const float pn = n.p[off] - step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + S.eps));
        n.p[off] = pn;
        if (polyak) n.tp[off] = S.tau * pn + (1.f - S.tau) * n.tp[off];
    };
    // weight / bias gradients of the listed passes as row-contraction tiles with the optimiser step in the epilogue.  which: 2 = output layer, 1 = layer 1, 0 = layer 0
    auto grad_tiles = [&](int which, const int* ps, int np, float step_size, float inv_bc2_sqrt, bool polyak) {
        const int per = which == 2 ? (TH >> 4) : which == 1 ? (TH >> 4) * (TH >> 4) : (TH >> 4) * (TXP >> 4);
        for (int t = gw; t < np * per; t += NW) {
            const int pi = t / per, tt = t - pi * per;
            const Pass& n = S.P[ps[pi]];
            float as;
            if (which == 2) {
                const int c0 = tt << 4;
                const floatx4 acc = rows_tile(n.dY + (size_t)(4 * g) * TYP + c, TYP, n.H2 + (size_t)(4 * g) * TH + c0 + c, TH, S.Rp >> 4, &as);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int o = 4 * g + v;
                    if (o < n.O) adam(n, oW2(n) + (size_t)o * TH + c0 + c, acc[v], step_size, inv_bc2_sqrt, polyak);
                }
                if (c0 == 0) {
                    as += __shfl_xor(as, 16, 64); as += __shfl_xor(as, 32, 64);
                    if (g == 0 && c < n.O) adam(n, oB2(n) + c, as, step_size, inv_bc2_sqrt, polyak);
                }
            } else {
                const bool one = which == 1;
                const int n0 = (one ? tt >> 4 : tt >> 2) << 4, k0 = (one ? tt & 15 : tt & 3) << 4;
                const int ldb = one ? TH : TXP;
                const floatx4 acc = rows_tile((one ? n.dZ2 : n.dZ1) + (size_t)(4 * g) * TH + n0 + c, TH, (one ? n.H1 : n.X) + (size_t)(4 * g) * ldb + k0 + c, ldb, S.Rp >> 4, &as);
                if (one) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) adam(n, oW1(n) + (size_t)(n0 + 4 * g + v) * TH + k0 + c, acc[v], step_size, inv_bc2_sqrt, polyak);
                } else if (k0 + c < n.D) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) adam(n, (size_t)(n0 + 4 * g + v) * n.D + k0 + c, acc[v], step_size, inv_bc2_sqrt, polyak);
                }
                if (k0 == 0) {
                    as += __shfl_xor(as, 16, 64); as += __shfl_xor(as, 32, 64);
                    if (g == 0) adam(n, (one ? oB1(n) : oB0(n)) + n0 + c, as, step_size, inv_bc2_sqrt, polyak);
                }
            }
        }
    };

    // ---- prologue: the Adam constants of every update, the rows of update 0 ---------------------------------------------------------------------------------------
    for (int u = gtid; u < S.U; u += gthreads) {
        int npol = 0;
        for (int q = 0; q <= u; ++q) npol += ((S.it0 + q) % S.policy_freq) == 0;
        const double tc = (double)(S.t_c0 + u + 1), ta = (double)(S.t_a0 + (npol > 0 ? npol : 1));
        S.tab[4 * u + 0] = (float)((double)S.c_lr / (1.0 - pow(0.9, tc))); S.tab[4 * u + 1] = (float)(1.0 / sqrt(1.0 - pow(0.999, tc)));
        S.tab[4 * u + 2] = (float)((double)S.a_lr / (1.0 - pow(0.9, ta))); S.tab[4 * u + 3] = (float)(1.0 / sqrt(1.0 - pow(0.999, ta)));
    }
    gather(0);
    barrier();

    for (int u = 0; u < S.U; ++u) {
        const bool pol = ((S.it0 + u) % S.policy_freq) == 0;
        const float c_step = ld_agent(S.tab + 4 * u), c_ibc = ld_agent(S.tab + 4 * u + 1), a_step = ld_agent(S.tab + 4 * u + 2), a_ibc = ld_agent(S.tab + 4 * u + 3);
        const float inv_b = 1.f / (float)B;
        // 1-2: hidden layers of actor_target(s'), Q1(s, a), Q2(s, a) and, with the actor step, actor(s)
        { const int ps[4] = {AT, Q1, Q2, PA}; fwd_tiles(0, ps, pol ? 4 : 3); }
        barrier();
        { const int ps[4] = {AT, Q1, Q2, PA}; fwd_tiles(1, ps, pol ? 4 : 3); }
        barrier();
        // 3: output layers: a' = clamp(max_action tanh(actor_target(s')) + clamp(noise)) -> action columns of the target critics' input; q1, q2; pi(s) -> Q1's input
        for (int rb = blockIdx.x; rb < RB; rb += G) {
            const int ps[4] = {AT, Q1, Q2, PA};
            out_block(rb, ps, pol ? 4 : 3);
            const int r = tid >> 4, j = tid & 15, row = rb * 16 + r;
            if (j < A) {
                float nz = S.noise[((size_t)u * B + row) * A + j];
                nz = fminf(fmaxf(nz, -S.noise_clip), S.noise_clip);
                const float a2 = fminf(fmaxf(S.max_action * tanhf(Ys[0][r][j]) + nz, -S.max_action), S.max_action);
                S.P[Q1T].X[(size_t)row * TXP + D + j] = a2;
                if (pol) {
                    const float pre = Ys[3][r][j];
                    S.P[PA].Y[(size_t)row * TYP + j] = pre;
                    S.P[QA].X[(size_t)row * TXP + D + j] = S.max_action * tanhf(pre);
                }
            }
            if (j == 0) { S.P[Q1].Y[(size_t)row * TYP] = Ys[1][r][0]; S.P[Q2].Y[(size_t)row * TYP] = Ys[2][r][0]; }
            __syncthreads();
        }
        barrier();
        // 4-5: hidden layers of the target critics on (s', a')
        { const int ps[2] = {Q1T, Q2T}; fwd_tiles(0, ps, 2); }
        barrier();
        { const int ps[2] = {Q1T, Q2T}; fwd_tiles(1, ps, 2); }
        barrier();
        // 6: target = r + notdone discount min(Q1', Q2') (sync_td3.py:157-159), both regression losses, d(loss)/d(q), d(loss)/d(pre-activation 2) of both critics
        for (int rb = blockIdx.x; rb < RB; rb += G) {
            const int ps[2] = {Q1T, Q2T};
            out_block(rb, ps, 2);
            double la[3] = {0, 0, 0};
            if (tid < 16) {
                const int row = rb * 16 + tid;
                const float t = S.rb[row] + S.ndb[row] * S.discount * fminf(Ys[0][tid][0], Ys[1][tid][0]);
                const float q1 = S.P[Q1].Y[(size_t)row * TYP], q2 = S.P[Q2].Y[(size_t)row * TYP];
                const float e1 = q1 - t, e2 = q2 - t;
                const float d1 = 2.f * e1 * inv_b, d2 = 2.f * e2 * inv_b;
                S.P[Q1].dY[(size_t)row * TYP] = d1; S.P[Q2].dY[(size_t)row * TYP] = d2;
                rowv[0][tid] = d1; rowv[1][tid] = d2;
                la[0] = ((double)e1 * e1 + (double)e2 * e2) / (double)B; la[1] = q1; la[2] = q2;
            }
            if (w == 0) {
#pragma unroll
                for (int q = 0; q < 3; ++q) { const double s = wsum(la[q]); if (lane == 0) S.part[rb * 4 + q] = s; }
            }
            __syncthreads();
            dz2_rank_one(S.P[Q1], rb, 0); dz2_rank_one(S.P[Q2], rb, 1);
            __syncthreads();
        }
        barrier();
        // 7: d(loss)/d(pre-activation 1) of both critics; their output layers' gradient + Adam (+ Polyak with the actor step: the targets were last read in phase 6)
        { const int ps[2] = {Q1, Q2}; dz1_tiles(ps, 2); grad_tiles(2, ps, 2, c_step, c_ibc, pol); }
        barrier();
        // 8: layers 1 and 0 of both critics: gradient + Adam (+ Polyak)
        { const int ps[2] = {Q1, Q2}; grad_tiles(1, ps, 2, c_step, c_ibc, pol); grad_tiles(0, ps, 2, c_step, c_ibc, pol); }
        if (pol) {
            barrier();
            // 9-10: Q1(s, pi(s)) with the UPDATED critic (sync_td3.py:178)
            { const int ps[1] = {QA}; fwd_tiles(0, ps, 1); }
            barrier();
            { const int ps[1] = {QA}; fwd_tiles(1, ps, 1); }
            barrier();
            // 11: actor loss -mean Q1(s, pi(s)); d(loss)/d(pre-activation 2) of that pass
            for (int rb = blockIdx.x; rb < RB; rb += G) {
                const int ps[1] = {QA};
                out_block(rb, ps, 1);
                double la = 0.0;
                if (tid < 16) { la = -(double)Ys[0][tid][0] / (double)B; rowv[0][tid] = -inv_b; }
                if (w == 0) { const double s = wsum(la); if (lane == 0) S.part[rb * 4 + 3] = s; }
                __syncthreads();
                dz2_rank_one(S.P[QA], rb, 0);
                __syncthreads();
            }
            barrier();
            { const int ps[1] = {QA}; dz1_tiles(ps, 1); }
            barrier();
            // 13: d(loss)/d(critic input), action columns only (input columns 48..63 as one tile) -> through max_action tanh -> d(loss)/d(actor output)
            {
                const Pass& n = S.P[QA]; const Pass& pa = S.P[PA];
                for (int t = gw; t < RB; t += NW) {
                    const int r0 = t << 4, d0 = 48;
                    const float* Ar = n.dZ1 + (size_t)(r0 + c) * TH + 4 * g;
                    const float* Wc = n.p + (size_t)(4 * g) * n.D + d0 + c;
                    const bool cok = d0 + c < n.D;
                    const floatx4 acc = wave_tile<16, false>(0, 16,
                        [&](int kc, float (&a)[4]) { ld4(Ar + 16 * kc, a); },
                        [&](int kc, float (&b)[4]) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) b[j] = cok ? Wc[(size_t)(16 * kc + j) * n.D] : 0.f;
                        }, nullptr);
                    const int j = d0 + c - D;
                    if (j >= 0 && j < A) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const size_t at = (size_t)(r0 + 4 * g + v) * TYP + j;
                            const float th = tanhf(pa.Y[at]);
                            pa.dY[at] = acc[v] * S.max_action * (1.f - th * th);
                        }
                    }
                }
            }
            barrier();
            // 14: d(loss)/d(pre-activation 2) of the actor = (dY W2) * (H2 > 0), K = 16 (outputs padded)
            {
                const Pass& n = S.P[PA];
                for (int t = gw; t < RB * (TH >> 4); t += NW) {
                    const int r0 = (t >> 4) << 4, n0 = (t & 15) << 4;
                    const float* Wc = n.p + oW2(n) + n0 + c;
                    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
                    float a[4], b[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int kk = 4 * g + j; a[j] = n.dY[(size_t)(r0 + c) * TYP + kk]; b[j] = kk < n.O ? Wc[(size_t)kk * TH] : 0.f; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const size_t e = (size_t)(r0 + 4 * g + v) * TH + n0 + c;
                        n.dZ2[e] = n.H2[e] > 0.f ? acc[v] : 0.f;
                    }
                }
            }
            barrier();
            // 15-16: the actor's backward, Adam and the Polyak average of the target actor (last read in phase 3)
            { const int ps[1] = {PA}; dz1_tiles(ps, 1); grad_tiles(2, ps, 1, a_step, a_ibc, true); }
            barrier();
            { const int ps[1] = {PA}; grad_tiles(1, ps, 1, a_step, a_ibc, true); grad_tiles(0, ps, 1, a_step, a_ibc, true); }
        }
        barrier();
        // the update's statistics (TD3.train's running sums, sync_td3.py:204-207) and the rows of the next update (behind a barrier of their own: the last gradient tiles
        // read the inputs the gather overwrites)
        if (blockIdx.x == 0 && tid == 0) {
            double a[4] = {0, 0, 0, 0};
            for (int rb = 0; rb < RB; ++rb)
                for (int q = 0; q < 4; ++q) a[q] += ld_agent(S.part + rb * 4 + q);      // (written in phases 6 and 11, at least one barrier ago)
            double* out = S.stats + (size_t)u * 4;
            out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = pol ? a[3] : 0.0;
        }
        if (u + 1 < S.U) gather(u + 1);
        barrier();
    }
    if (blockIdx.x == 0 && tid == 0 && ld_agent(S.bar + 1) != 0u)      // the barrier's watchdog fired: nothing of this block of updates is valid
        for (int i = 0; i < S.U * 4; ++i) S.stats[i] = __builtin_nan("");
}

// ---------------------------------------------------------------------------------------------------------------------- host side
namespace {
size_t tup64(size_t x) { return (x + 63) & ~(size_t)63; }
long trpad(long r) { return (r + 63) / 64 * 64; }
struct Td3Ws {
    float *X[5], *H1[NPASS], *H2[NPASS], *dZ2[NPASS], *dZ1[NPASS], *dY[NPASS], *Y[NPASS], *rb, *ndb, *tab;
    double* part; unsigned* bar;
    size_t bytes;
    Td3Ws(void* base, long B, long U) {
        char* p = (char*)base; size_t off = 0;
        auto take = [&](size_t nbytes) { char* r = (char*)((uintptr_t)p + off); off += tup64(nbytes); return (float*)r;      /* (integer arithmetic: the size query runs this with a NULL base) */ };
        const long R = trpad(B);
        for (int i = 0; i < 5; ++i) X[i] = take(R * TXP * 4);      // inputs of AT, (Q1T, Q2T), (Q1, Q2), PA, QA
        for (int i = 0; i < NPASS; ++i) {
            H1[i] = take(R * TH * 4); H2[i] = take(R * TH * 4);
            const bool bw = i == Q1 || i == Q2 || i == PA || i == QA;
            dZ2[i] = bw ? take(R * TH * 4) : nullptr; dZ1[i] = bw ? take(R * TH * 4) : nullptr;
            const bool oy = i == Q1 || i == Q2 || i == PA;
            dY[i] = oy ? take(R * TYP * 4) : nullptr; Y[i] = oy ? take(R * TYP * 4) : nullptr;
        }
        rb = take(R * 4); ndb = take(R * 4); tab = take(U * 4 * 4);
        part = (double*)take((B / 16) * 4 * 8);
        bar = (unsigned*)take(64);
        bytes = off;
    }
};
}  // namespace

extern "C" int apx_td3_updates_supported(int64_t B, int D, int H, int A) {
    return H == TH && D > 0 && A > 0 && A <= 12 && D >= 48 && D + A <= TXP && B >= 16 && B <= 4096 && B % 16 == 0;      // (the action columns of the critic input sit in columns 48..63)
}
extern "C" size_t apx_td3_updates_workspace_bytes(int64_t B, int64_t U, int D, int H, int A) {
    if (!apx_td3_updates_supported(B, D, H, A) || U <= 0) return 0;
    Td3Ws w(nullptr, B, U);
    return w.bytes;
}

extern "C" int apx_td3_updates(const apx_td3_args* a, void* stream) {
    APX_REQUIRE(a, "args");
    APX_REQUIRE(a->actor && a->actor_t && a->actor_m && a->actor_v && a->critic && a->critic_t && a->critic_m && a->critic_v, "network / optimiser pointers");
    APX_REQUIRE(a->state && a->next_state && a->action && a->reward && a->notdone && a->ind && a->noise && a->stats_out, "replay / index / noise / output pointers");
    APX_REQUIRE(apx_td3_updates_supported(a->B, a->D, a->H, a->A), "shape: H = 256, 48 <= D, D + A <= 64, A <= 12, batch a multiple of 16 in 16..4096");
    APX_REQUIRE(a->U > 0 && a->U < (1 << 20) && a->policy_freq >= 1 && a->it0 >= 0 && a->t_a >= 0 && a->t_c >= 0, "update count / policy_freq / step counters");
    APX_REQUIRE(a->workspace && a->workspace_bytes >= apx_td3_updates_workspace_bytes(a->B, a->U, a->D, a->H, a->A), "workspace (apx_td3_updates_workspace_bytes)");
    APX_REQUIRE((((uintptr_t)a->actor | (uintptr_t)a->actor_t | (uintptr_t)a->critic | (uintptr_t)a->critic_t | (uintptr_t)a->workspace) & 15) == 0, "16-byte aligned parameter blocks and workspace");
    hipStream_t s = (hipStream_t)stream;
    Td3Ws w(a->workspace, a->B, a->U);
    const long n1 = (long)TH * (a->D + a->A) + TH + (long)TH * TH + TH + TH + 1;      // one critic of the flat twin block (apx_mlp_param_count(D + A, 256, 1))
    TArgs S;
    const int xi[NPASS] = {0, 1, 1, 2, 2, 3, 4};
    for (int i = 0; i < NPASS; ++i) {
        Pass& n = S.P[i];
        n.m = n.v = n.tp = nullptr;
        n.X = w.X[xi[i]]; n.H1 = w.H1[i]; n.H2 = w.H2[i]; n.dZ2 = w.dZ2[i]; n.dZ1 = w.dZ1[i]; n.dY = w.dY[i]; n.Y = w.Y[i];
        n.D = (i == AT || i == PA) ? a->D : a->D + a->A; n.O = (i == AT || i == PA) ? a->A : 1;
    }
    S.P[AT].p = a->actor_t; S.P[PA].p = a->actor; S.P[PA].m = a->actor_m; S.P[PA].v = a->actor_v; S.P[PA].tp = a->actor_t;
    S.P[Q1T].p = a->critic_t; S.P[Q2T].p = a->critic_t + n1;
    S.P[Q1].p = a->critic; S.P[Q1].m = a->critic_m; S.P[Q1].v = a->critic_v; S.P[Q1].tp = a->critic_t;
    S.P[Q2].p = a->critic + n1; S.P[Q2].m = a->critic_m + n1; S.P[Q2].v = a->critic_v + n1; S.P[Q2].tp = a->critic_t + n1;
    S.P[QA].p = a->critic;
    for (int i = 0; i < NPASS; ++i) S.P[i].al = ((uintptr_t)S.P[i].p & 15) == 0;
    S.D = a->D; S.A = a->A; S.B = (int)a->B; S.Rp = (int)trpad(a->B); S.U = (int)a->U; S.it0 = a->it0; S.policy_freq = a->policy_freq;
    S.rs = a->state; S.rs2 = a->next_state; S.ra = a->action; S.rr = a->reward; S.rnd = a->notdone; S.ind = a->ind; S.noise = a->noise;
    S.max_action = a->max_action; S.noise_clip = a->noise_clip; S.discount = a->discount; S.tau = a->tau; S.a_lr = a->a_lr; S.c_lr = a->c_lr; S.eps = a->adam_eps;
    S.t_a0 = a->t_a; S.t_c0 = a->t_c;
    S.rb = w.rb; S.ndb = w.ndb; S.tab = w.tab; S.part = w.part; S.bar = w.bar; S.stats = a->stats_out;
    static const int forced = getenv("APX_TD3_UPDATES_WGS") ? atoi(getenv("APX_TD3_UPDATES_WGS")) : 0;
    int G = a->B <= 64 ? 64 : TMAXG;      // every workgroup resident at once (the barrier spins): far below the 256 CUs
    if (forced >= 1 && forced <= TMAXG) G = forced;
    APX_HIP(hipMemsetAsync(a->workspace, 0, w.bytes, s));      // the barrier counter, the padding rows / columns of the row buffers
    return tiles::launch_resident(td3_small_kernel, G, 256, s, S, "apx_td3_updates");
}
