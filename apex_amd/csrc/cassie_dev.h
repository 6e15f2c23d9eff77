// Device-side Cassie rigid-body step for gfx950: one environment per lane (wave = 64 envs in lock-step).
//
// What it replaces: cassie_sim_step_pd -> mj_step inside libcassiemujoco.so / MuJoCo 2.00
// (cassie/cassiemujoco/cassiemujoco.py:46-49, include/cassiemujoco.h:80).
//
// Formulation — fp32, tree-sparse, no dense 32x32 objects:
//   * FK / CRBA / RNE about a common reference point (the pelvis origin) so spatial quantities add without transforms
//   * mass matrix in MuJoCo's sparse ancestor-chain layout (CM_NM = 307 entries), L^T D L factorisation in place
//   * every constraint row is built sparse over [6 pelvis dofs | 13 dofs of ONE leg], immediately half-solved and
//     scaled:  y~_r = D^-1/2 L^-T J_r^T.  Then A = Y~ Y~^T + R is never formed: projected Gauss-Seidel runs in the
//     32-dim "whitened" space  z~ = sum_r y~_r f_r  (row update = two 19-term dot/axpy), rows live in LDS.
//   * Euler with implicit joint damping: second sparse factorisation of M + h D.
#pragma once
#include <hip/hip_runtime.h>
#include "cassie_model_gen.h"

namespace cas {

constexpr int NB = CM_NBODY, NV = CM_NV, NQ = CM_NQ, NJ = CM_NJNT, NG = CM_NGEOM, NEQ = CM_NEQ, NU = CM_NU, NM = CM_NM;
constexpr int MAXCON = 8, MAXLIM = 4, MAXEFC = 3 * NEQ + MAXLIM + 4 * MAXCON;   // 48, same caps as the oracle
constexpr int YW = 19;                 // row width: 6 pelvis dofs + 13 dofs of the row's leg
constexpr float DT = 0.0005f, GRAV = 9.81f, MINVAL = 1e-15f;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
    const float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    if (!(n2 > 1e-30f)) return {1.f, 0.f, 0.f, 0.f};
    const float s = rsqrtf(n2);
    return {q.w * s, q.x * s, q.y * s, q.z * s};
}
struct M3 { float m[9]; };
__device__ __forceinline__ M3 q2m(Q4 q) {
    const float w = q.w, x = q.x, y = q.y, z = q.z;
    return {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
             1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
             1 - 2 * (x * x + y * y)}};
}
__device__ __forceinline__ V3 mul(const M3& R, V3 v) {
    return {R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
            R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z};
}
__device__ __forceinline__ V3 col(const M3& R, int k) { return {R.m[k], R.m[3 + k], R.m[6 + k]}; }

struct SV { V3 a, l; };   // spatial motion / force about the reference point
__device__ __forceinline__ SV operator+(SV p, SV q) { return {p.a + q.a, p.l + q.l}; }
__device__ __forceinline__ SV operator*(SV p, float s) { return {p.a * s, p.l * s}; }
__device__ __forceinline__ SV crossMotion(SV v, SV s) { return {cross(v.a, s.a), cross(v.a, s.l) + cross(v.l, s.a)}; }
__device__ __forceinline__ SV crossForce(SV v, SV f) { return {cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }
__device__ __forceinline__ float sdot(SV m, SV f) { return dot(m.a, f.a) + dot(m.l, f.l); }
struct SI { float m; V3 h; float I[6]; };
__device__ __forceinline__ V3 symmul(const float* I, V3 v) {
    return {I[0] * v.x + I[3] * v.y + I[4] * v.z, I[3] * v.x + I[1] * v.y + I[5] * v.z, I[4] * v.x + I[5] * v.y + I[2] * v.z};
}
__device__ __forceinline__ SV imul(const SI& s, SV v) { return {symmul(s.I, v.a) + cross(s.h, v.l), v.l * s.m - cross(s.h, v.a)}; }

// per-env model parameters and solver products that outlive one forward pass
struct Dyn {
    float mass[NB];
    float damping[NV];
    float friction;
    V3 fn, ft1, ft2;          // floor normal and tangents (world)
    float biw[NB];            // body_invweight0 (translational part)
    float diw[NV];            // dof_invweight0
};

// working set of one forward pass (thread-private; the compiler keeps what it can in VGPRs, the rest is scratch)
struct Work {
    V3 xpos[NB]; M3 xmat[NB]; Q4 xquat[NB];
    SV cdof[NV];
    SV cvel[NB];
    SI crb[NB];
    float M[NM], LD[NM];
    float dsqrt[NV], disqrt[NV];   // sqrt(D), 1/sqrt(D)
    float ut[NV];                  // u~ = D^-1/2 L^-T qfrc_smooth
    float zt[NV];                  // z~ = sum_r y~_r f_r
    float smooth[NV];
    float qacc[NV];
    V3 o;
    // constraint rows (meta per row; the y~ vectors live in LDS)
    int nefc, ncon;
    float f[MAXEFC], Rr[MAXEFC], br[MAXEFC], diag[MAXEFC];
    unsigned char leg[MAXEFC], typ[MAXEFC];
    int con_row[MAXCON]; unsigned char con_geom[MAXCON];
    float foot_force[2][3];
    float acc[3];
};

__device__ __forceinline__ int body_lastdof(int b) {
    while (b > 0 && cm_body_dofnum[b] == 0) b = cm_body_parent[b];
    return b > 0 ? cm_body_dofadr[b] + cm_body_dofnum[b] - 1 : -1;
}
// local column k (0..18) of a row of leg `leg` -> global dof
__device__ __forceinline__ int col2dof(int k, int leg) { return k < 6 ? k : k + 13 * leg; }
__device__ __forceinline__ int dof2col(int d) { return d < 6 ? d : (d < 19 ? d : d - 13); }

// ------------------------------------------------------------------------------------------------ kinematics + CRBA
template <class QP>
__device__ void kin_crba(const QP& qpos, const Dyn& dy, Work& w) {
    w.xpos[0] = {0, 0, 0}; w.xquat[0] = {1, 0, 0, 0}; w.xmat[0] = q2m(w.xquat[0]);
    V3 axis_w[NJ], anchor[NJ];
    int j = 0;   // joints are stored in body order
    for (int b = 1; b < NB; ++b) {
        const int p = cm_body_parent[b];
        V3 pos = w.xpos[p] + mul(w.xmat[p], ld3(cm_body_pos + 3 * b));
        Q4 quat = qmul(w.xquat[p], Q4{cm_body_quat[4 * b], cm_body_quat[4 * b + 1], cm_body_quat[4 * b + 2], cm_body_quat[4 * b + 3]});
        while (j < NJ && cm_jnt_body[j] == b) {
            const int adr = cm_jnt_qposadr[j];
            const M3 R = q2m(quat);
            axis_w[j] = mul(R, ld3(cm_jnt_axis + 3 * j));
            anchor[j] = pos;
            const int t = cm_jnt_type[j];
            if (t == 0) pos = pos + axis_w[j] * (qpos(adr) - cm_jnt_ref[j]);
            else if (t == 1) {
                const float ang = 0.5f * (qpos(adr) - cm_jnt_ref[j]);
                float sn, cs;
                sincosf(ang, &sn, &cs);
                quat = qmul(quat, Q4{cs, cm_jnt_axis[3 * j] * sn, cm_jnt_axis[3 * j + 1] * sn, cm_jnt_axis[3 * j + 2] * sn});
            } else quat = qmul(quat, qnormalize(Q4{qpos(adr), qpos(adr + 1), qpos(adr + 2), qpos(adr + 3)}));
            ++j;
        }
        w.xquat[b] = qnormalize(quat); w.xpos[b] = pos; w.xmat[b] = q2m(w.xquat[b]);
    }
    w.o = w.xpos[1];
    for (int d = 0; d < NV; ++d) {
        const int jj = cm_dof_jnt[d], b = cm_dof_body[d], t = cm_jnt_type[jj];
        if (t == 0) w.cdof[d] = {{0, 0, 0}, axis_w[jj]};
        else if (t == 1) w.cdof[d] = {axis_w[jj], cross(axis_w[jj], w.o - anchor[jj])};
        else {
            const V3 ax = col(w.xmat[b], d - cm_jnt_dofadr[jj]);
            w.cdof[d] = {ax, cross(ax, w.o - anchor[jj])};
        }
    }
    // body inertias about o in world axes, composite (subtree) sums
    for (int b = 1; b < NB; ++b) {
        const M3& R = w.xmat[b];
        const float* Ib = cm_body_inertia + 9 * b;
        float RI[9], Iw[9];
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) RI[3 * i + k] = R.m[3 * i] * Ib[k] + R.m[3 * i + 1] * Ib[3 + k] + R.m[3 * i + 2] * Ib[6 + k];
        for (int i = 0; i < 3; ++i)
            for (int k = i; k < 3; ++k) Iw[3 * i + k] = RI[3 * i] * R.m[3 * k] + RI[3 * i + 1] * R.m[3 * k + 1] + RI[3 * i + 2] * R.m[3 * k + 2];
        const float m = dy.mass[b];
        const V3 r = w.xpos[b] + mul(R, ld3(cm_body_ipos + 3 * b)) - w.o;
        const float rr = dot(r, r);
        SI c;
        c.m = m; c.h = r * m;
        c.I[0] = Iw[0] + m * (rr - r.x * r.x); c.I[1] = Iw[4] + m * (rr - r.y * r.y); c.I[2] = Iw[8] + m * (rr - r.z * r.z);
        c.I[3] = Iw[1] - m * r.x * r.y; c.I[4] = Iw[2] - m * r.x * r.z; c.I[5] = Iw[5] - m * r.y * r.z;
        w.crb[b] = c;
    }
}

// composite accumulation is separated so that the RNE pass can read the per-body inertias first
__device__ void crba(Work& w) {
    for (int b = NB - 1; b >= 2; --b) {
        const int p = cm_body_parent[b];
        SI& P = w.crb[p]; const SI& C = w.crb[b];
        P.m += C.m; P.h = P.h + C.h;
        for (int i = 0; i < 6; ++i) P.I[i] += C.I[i];
    }
    for (int i = 0; i < NV; ++i) {
        const SV f = imul(w.crb[cm_dof_body[i]], w.cdof[i]);
        int a = cm_dof_madr[i];
        w.M[a++] = sdot(w.cdof[i], f) + cm_dof_armature[i];
        for (int jd = cm_dof_parent[i]; jd >= 0; jd = cm_dof_parent[jd]) w.M[a++] = sdot(w.cdof[jd], f);
    }
}

// in-place sparse L^T D L (MuJoCo mj_factorM layout): LD[madr[k]] = D_k, LD[madr[k]+n] = L[k][n-th ancestor]
__device__ void factor(float* LD, float* dsqrt, float* disqrt) {
    for (int k = NV - 1; k >= 0; --k) {
        const int kk = cm_dof_madr[k];
        const float dinv = 1.f / LD[kk];
        int ki = kk + 1;
        for (int i = cm_dof_parent[k]; i >= 0; i = cm_dof_parent[i], ++ki) {
            const float tmp = LD[ki] * dinv;
            int ij = cm_dof_madr[i], kj = ki;
            for (int jd = i; jd >= 0; jd = cm_dof_parent[jd]) LD[ij++] -= tmp * LD[kj++];
            LD[ki] = tmp;
        }
        if (dsqrt) { dsqrt[k] = sqrtf(LD[kk]); disqrt[k] = rsqrtf(LD[kk]); }
    }
}
// x <- L^-T x   (leaves -> root)
__device__ __forceinline__ void solve_LT(const float* LD, float* x) {
    for (int i = NV - 1; i >= 0; --i) {
        int a = cm_dof_madr[i] + 1;
        const float xi = x[i];
        for (int jd = cm_dof_parent[i]; jd >= 0; jd = cm_dof_parent[jd]) x[jd] -= LD[a++] * xi;
    }
}
// x <- L^-1 x   (root -> leaves)
__device__ __forceinline__ void solve_L(const float* LD, float* x) {
    for (int i = 0; i < NV; ++i) {
        int a = cm_dof_madr[i] + 1;
        float xi = x[i];
        for (int jd = cm_dof_parent[i]; jd >= 0; jd = cm_dof_parent[jd]) xi -= LD[a++] * x[jd];
        x[i] = xi;
    }
}
// y <- L x
__device__ __forceinline__ void mul_L(const float* LD, const float* x, float* y) {
    for (int i = 0; i < NV; ++i) {
        int a = cm_dof_madr[i] + 1;
        float s = x[i];
        for (int jd = cm_dof_parent[i]; jd >= 0; jd = cm_dof_parent[jd]) s += LD[a++] * x[jd];
        y[i] = s;
    }
}
// y <- L^T x
__device__ __forceinline__ void mul_LT(const float* LD, const float* x, float* y) {
    for (int i = 0; i < NV; ++i) y[i] = x[i];
    for (int i = 0; i < NV; ++i) {
        int a = cm_dof_madr[i] + 1;
        for (int jd = cm_dof_parent[i]; jd >= 0; jd = cm_dof_parent[jd]) y[jd] += LD[a++] * x[i];
    }
}

// Row storage in LDS: element k of row r of this lane at  rows[(r * YW + k) * 64 + lane]  (lane-contiguous: no bank
// conflicts, each ds_read_b32 serves the whole wave).
// The first NLDS rows fit the CU's 160 KiB of LDS (33 * 19 * 256 B = 160,512 B, one wave per CU); the rare rows beyond
// that (more than ~5 simultaneous contacts) spill to a thread-private array.
constexpr int NLDS = 33;
struct Rows {
    float* base; int lane; float* ext;
    __device__ __forceinline__ float& at(int r, int k) const {
        return r < NLDS ? base[(r * YW + k) * 64 + lane] : ext[(r - NLDS) * YW + k];
    }
};

// Half-solve + scale one row given as sparse J over the dofs of (pelvis, leg):  y~ = D^-1/2 L^-T J^T.
// `Jl` is indexed by local column (0..18).
__device__ __forceinline__ void whiten_row(const Work& w, float* Jl, int leg) {
    // leaves -> root inside the leg, then the pelvis chain; ancestors of a leg dof stay inside (pelvis, leg)
    for (int k = YW - 1; k >= 0; --k) {
        const int i = col2dof(k, leg);
        int a = cm_dof_madr[i] + 1;
        const float xi = Jl[k];
        for (int jd = cm_dof_parent[i]; jd >= 0; jd = cm_dof_parent[jd]) Jl[dof2col(jd)] -= w.LD[a++] * xi;
    }
    for (int k = 0; k < YW; ++k) Jl[k] *= w.disqrt[col2dof(k, leg)];
}

__device__ __forceinline__ float impedance(float pos) {   // MuJoCo solimp defaults 0.9 0.95 0.001 0.5 2
    const float x = fabsf(pos) * 1000.f;
    if (x >= 1.f) return 0.95f;
    if (x <= 0.f) return 0.9f;
    const float y = x <= 0.5f ? 2.f * x * x : 1.f - 2.f * (1.f - x) * (1.f - x);
    return 0.9f + y * 0.05f;
}

// translational Jacobian of point p of body b, accumulated (with sign) into three local-column rows
__device__ __forceinline__ void jac_point(const Work& w, int b, V3 p, float sign, float* Jx, float* Jy, float* Jz) {
    const V3 r = p - w.o;
    for (int d = body_lastdof(b); d >= 0; d = cm_dof_parent[d]) {
        const V3 v = w.cdof[d].l + cross(w.cdof[d].a, r);
        const int k = dof2col(d);
        Jx[k] += sign * v.x; Jy[k] += sign * v.y; Jz[k] += sign * v.z;
    }
}

// finishes row r: y~ (already whitened, in Jl) -> LDS; velocity, regulariser, reference acceleration, b
__device__ __forceinline__ void commit_row(Work& w, const Rows& Y, int r, const float* Jl, int leg, int typ, float pos,
                                           float imp_pos, float diag, float timeconst, const float* vt, const float* wt,
                                           float* jar_out) {
    float vel = 0.f, ju = 0.f, jw = 0.f, nn = 0.f;
    for (int k = 0; k < YW; ++k) {
        const int d = col2dof(k, leg);
        const float y = Jl[k];
        Y.at(r, k) = y;
        vel += y * vt[d]; ju += y * w.ut[d]; jw += y * wt[d]; nn += y * y;
    }
    const float dmax = 0.95f;
    const float K = 1.f / (dmax * dmax * timeconst * timeconst), B = 2.f / (dmax * timeconst);
    const float imp = impedance(imp_pos);
    const float R = fmaxf(MINVAL, (1.f - imp) / imp * diag);
    const float aref = -B * vel - K * imp * pos;
    w.leg[r] = (unsigned char)leg; w.typ[r] = (unsigned char)typ;
    w.Rr[r] = R; w.br[r] = ju - aref; w.diag[r] = nn;
    *jar_out = jw - aref;
}

}  // namespace cas
