// CPU ORACLE (test infrastructure only).  See cassie_phys.h for scope and provenance.
#include "cassie_phys.h"

namespace orc {

thread_local unsigned long long g_flops = 0;
#define CNT(n) (g_flops += (unsigned long long)(n))
#define NZ(a, b) (((a) != 0.0 && (b) != 0.0) ? 2 : 0)

static int body_lastdof(int b) {
    while (b > 0 && cm_body_dofnum[b] == 0) b = cm_body_parent[b];
    return b > 0 ? cm_body_dofadr[b] + cm_body_dofnum[b] - 1 : -1;
}

static bool cholesky(const double A[NV][NV], double L[NV][NV]) {
    for (int i = 0; i < NV; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i][j];
            for (int k = 0; k < j; ++k) { CNT(NZ(L[i][k], L[j][k])); s -= L[i][k] * L[j][k]; }
            if (i == j) {
                if (s <= 0) return false;
                L[i][i] = std::sqrt(s); CNT(1);
            } else {
                L[i][j] = s / L[j][j]; CNT(s != 0.0 ? 1 : 0);
            }
        }
    return true;
}
static void chol_solve(const double L[NV][NV], const double* b, double* x) {
    double y[NV];
    for (int i = 0; i < NV; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) { CNT(NZ(L[i][k], y[k])); s -= L[i][k] * y[k]; }
        y[i] = s / L[i][i]; CNT(s != 0.0 ? 1 : 0);
    }
    for (int i = NV - 1; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < NV; ++k) { CNT(NZ(L[k][i], x[k])); s -= L[k][i] * x[k]; }
        x[i] = s / L[i][i]; CNT(s != 0.0 ? 1 : 0);
    }
}

// ---------------------------------------------------------------------------------------------- kinematics
// MuJoCo mj_kinematics + mj_comPos for the joint types cassie.xml uses (slide / hinge / ball; joint pos = 0).
static void kinematics(const double* qpos, State& s, Work& w) {
    s.xpos[0] = {0, 0, 0}; s.xquat[0] = {1, 0, 0, 0}; s.xmat[0] = q2m(s.xquat[0]);
    V3 axis_w[NJ];
    for (int b = 1; b < NB; ++b) {
        const int p = cm_body_parent[b];
        V3 pos = s.xpos[p] + mul(s.xmat[p], v3(cm_body_pos + 3 * b));
        Q4 quat = qmul(s.xquat[p], Q4{cm_body_quat[4 * b], cm_body_quat[4 * b + 1], cm_body_quat[4 * b + 2], cm_body_quat[4 * b + 3]});
        for (int j = 0; j < NJ; ++j) {
            if (cm_jnt_body[j] != b) continue;
            const int adr = cm_jnt_qposadr[j];
            const M3 R = q2m(quat);
            axis_w[j] = mul(R, v3(cm_jnt_axis + 3 * j));
            w.anchor[j] = pos;
            if (cm_jnt_type[j] == 0) pos = pos + axis_w[j] * (qpos[adr] - cm_jnt_ref[j]);
            else if (cm_jnt_type[j] == 1) quat = qmul(quat, qaxisangle(v3(cm_jnt_axis + 3 * j), qpos[adr] - cm_jnt_ref[j]));
            else quat = qmul(quat, qnormalize(Q4{qpos[adr], qpos[adr + 1], qpos[adr + 2], qpos[adr + 3]}));
        }
        s.xquat[b] = qnormalize(quat); s.xpos[b] = pos; s.xmat[b] = q2m(s.xquat[b]);
        CNT(15 + 3 + 28 + 13 + 30 + (cm_body_jntnum[b] ? 15 + 40 : 0));      // frame composition, joint rotation, normalisation, matrix
    }
    w.o = s.xpos[1];
    for (int d = 0; d < NV; ++d) {
        const int j = cm_dof_jnt[d], b = cm_dof_body[d];
        if (cm_jnt_type[j] == 0) w.cdof[d] = {{0, 0, 0}, axis_w[j]};
        else if (cm_jnt_type[j] == 1) { w.cdof[d] = {axis_w[j], cross(axis_w[j], w.o - w.anchor[j])}; CNT(12); }
        else {
            const V3 ax = col(s.xmat[b], d - cm_jnt_dofadr[j]);
            w.cdof[d] = {ax, cross(ax, w.o - w.anchor[j])}; CNT(12);
        }
    }
}

static void inertias(const Params& p, const State& s, Work& w) {
    for (int b = 0; b < NB; ++b) {
        const M3& R = s.xmat[b];
        const double* Ib = cm_body_inertia + 9 * b;
        double RI[9], Iw[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double a = 0;
                for (int k = 0; k < 3; ++k) a += R.m[3 * i + k] * Ib[3 * k + j];
                RI[3 * i + j] = a;
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double a = 0;
                for (int k = 0; k < 3; ++k) a += RI[3 * i + k] * R.m[3 * j + k];
                Iw[3 * i + j] = a;
            }
        const double m = p.mass[b];       // mass randomisation changes mass only; body_inertia stays (cassie.py:640)
        const V3 r = s.xpos[b] + mul(R, v3(cm_body_ipos + 3 * b)) - w.o;
        const double rr = dot(r, r);
        SI c;
        c.m = m; c.h = r * m;
        c.I[0] = Iw[0] + m * (rr - r.x * r.x); c.I[1] = Iw[4] + m * (rr - r.y * r.y); c.I[2] = Iw[8] + m * (rr - r.z * r.z);
        c.I[3] = Iw[1] - m * r.x * r.y; c.I[4] = Iw[2] - m * r.x * r.z; c.I[5] = Iw[5] - m * r.y * r.z;
        w.cinert[b] = c; w.crb[b] = c; CNT(54 + 54 + 18 + 3 + 24);
    }
    for (int b = NB - 1; b >= 1; --b) { w.crb[cm_body_parent[b]] = w.crb[cm_body_parent[b]] + w.crb[b]; CNT(10); }
    // CRBA
    for (int i = 0; i < NV; ++i)
        for (int j = 0; j < NV; ++j) w.M[i][j] = 0;
    for (int i = 0; i < NV; ++i) {
        const SV f = imul(w.crb[cm_dof_body[i]], w.cdof[i]);
        w.M[i][i] = sdot(w.cdof[i], f) + cm_dof_armature[i]; CNT(45 + 12);
        for (int j = cm_dof_parent[i]; j >= 0; j = cm_dof_parent[j]) { w.M[i][j] = w.M[j][i] = sdot(w.cdof[j], f); CNT(11); }
    }
}

// point Jacobian (translational) of a point fixed to body b
static void jac_point(const Work& w, int b, V3 p, double Jx[NV], double Jy[NV], double Jz[NV], double sign) {
    const V3 r = p - w.o;
    for (int d = body_lastdof(b); d >= 0; d = cm_dof_parent[d]) {
        const V3 v = w.cdof[d].l + cross(w.cdof[d].a, r);
        Jx[d] += sign * v.x; Jy[d] += sign * v.y; Jz[d] += sign * v.z; CNT(12 + 6);
    }
}

// closest points of two segments p1 + s d1, p2 + t d2, s, t in [0, 1] (mjc_CapsuleCapsule reduces to this for non-parallel axes; for
// parallel axes MuJoCo may emit two contacts, here the clamped solution gives one)
static void closest_seg_seg(V3 p1, V3 q1, V3 p2, V3 q2, V3& c1, V3& c2) {
    const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const double a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    double sp, tp;
    if (a <= MINVAL && e <= MINVAL) { c1 = p1; c2 = p2; return; }
    if (a <= MINVAL) { sp = 0; tp = std::min(std::max(f / e, 0.0), 1.0); }
    else {
        const double c = dot(d1, r);
        if (e <= MINVAL) { tp = 0; sp = std::min(std::max(-c / a, 0.0), 1.0); }
        else {
            const double b = dot(d1, d2), den = a * e - b * b;
            sp = den > MINVAL ? std::min(std::max((b * f - c * e) / den, 0.0), 1.0) : 0.0;
            tp = (b * sp + f) / e;
            if (tp < 0) { tp = 0; sp = std::min(std::max(-c / a, 0.0), 1.0); }
            else if (tp > 1) { tp = 1; sp = std::min(std::max((b - c) / a, 0.0), 1.0); }
        }
    }
    c1 = p1 + d1 * sp; c2 = p2 + d2 * tp;
}

static void impedance(double pos, double& imp) {
    // MuJoCo solimp = (d0, dwidth, width, midpoint, power) default 0.9 0.95 0.001 0.5 2
    const double d0 = 0.9, d1 = 0.95, width = 0.001, mid = 0.5, power = 2;
    double x = std::fabs(pos) / width;
    if (x >= 1) { imp = d1; return; }
    if (x <= 0) { imp = d0; return; }
    double y;
    if (x <= mid) y = std::pow(x, power) / std::pow(mid, power - 1);
    else y = 1 - std::pow(1 - x, power) / std::pow(1 - mid, power - 1);
    imp = d0 + y * (d1 - d0);
}

static void finish_row(Row& r, const double* qvel, double imp_pos, double timeconst, double dampratio) {
    const double dmax = 0.95;
    const double K = 1.0 / std::max(MINVAL, dmax * dmax * timeconst * timeconst * dampratio * dampratio);
    const double B = 2.0 / std::max(MINVAL, dmax * timeconst);
    double imp;
    impedance(imp_pos, imp);
    r.vel = 0;
    for (int d = 0; d < NV; ++d) { CNT(NZ(r.J[d], qvel[d])); r.vel += r.J[d] * qvel[d]; }
    CNT(20);
    r.R = std::max(MINVAL, (1 - imp) / imp * r.diag);
    r.aref = -B * r.vel - K * imp * r.pos;
}

// Height field of cassie_hfield.xml as MuJoCo lays it out (hfield_data row-major, rows along y, columns along x, elevation =
// data * size[2]; the geom sits at the origin).  Every grid cell is split into the two triangles (00, 10, 01) and (11, 01, 10); a point
// collides with the plane of the triangle under it.  MuJoCo itself collides the capsule with triangular PRISMS through its general
// convex routine (mjc_ConvexHField, closed source): same surface, contact point / normal can differ near cell edges.
void floor_query(const Params& p, double x, double y, double& hh, V3& n) {
    if (!p.hf_data) {
        const M3 Rf = q2m(p.floor_quat);
        n = col(Rf, 2);
        const V3 p0 = v3(cm_floor_pos);
        hh = p0.z - (n.x * (x - p0.x) + n.y * (y - p0.y)) / n.z;
        return;
    }
    const int nc = p.hf_ncol, nr = p.hf_nrow;
    const double dx = 2 * p.hf_size[0] / (nc - 1), dy = 2 * p.hf_size[1] / (nr - 1);
    double u = std::min(std::max((x + p.hf_size[0]) / dx, 0.0), nc - 1 - 1e-9), v = std::min(std::max((y + p.hf_size[1]) / dy, 0.0), nr - 1 - 1e-9);
    const int c = (int)u, r = (int)v;
    const double fu = u - c, fv = v - r, sz = p.hf_size[2];
    const double h00 = sz * p.hf_data[r * nc + c], h10 = sz * p.hf_data[r * nc + c + 1], h01 = sz * p.hf_data[(r + 1) * nc + c], h11 = sz * p.hf_data[(r + 1) * nc + c + 1];
    double gx, gy;
    if (fu + fv <= 1) { gx = (h10 - h00) / dx; gy = (h01 - h00) / dy; hh = h00 + fu * (h10 - h00) + fv * (h01 - h00); }
    else { gx = (h11 - h01) / dx; gy = (h11 - h10) / dy; hh = h11 + (1 - fu) * (h01 - h11) + (1 - fv) * (h10 - h11); }
    const double inv = 1.0 / std::sqrt(gx * gx + gy * gy + 1);
    n = {-gx * inv, -gy * inv, inv};
}

void default_params(Params& p) {
    for (int b = 0; b < NB; ++b) p.mass[b] = cm_body_mass[b];
    for (int d = 0; d < NV; ++d) p.damping[d] = cm_dof_damping[d];
    p.friction = 1.0;
    p.floor_quat = {1, 0, 0, 0};
    p.pgs_iters = 50; p.tolerance = 0;      // see Params::tolerance
    set_const(p);
}

// mj_setConst subset: body_invweight0 / dof_invweight0 from M^-1 at qpos0 (used by the constraint regulariser R)
void set_const(Params& p) {
    static thread_local State s; static thread_local Work w;      // (thread_local: resets run concurrently in orc_rollout_bench and in the threaded tests; plain statics raced until round 4)
    kinematics(cm_qpos0, s, w);
    inertias(p, s, w);
    {   // set0: stat.meaninertia = mean of the diagonal of M at qpos0 (scale of the solver's termination test)
        double tr = 0;
        for (int d = 0; d < NV; ++d) tr += w.M[d][d];
        p.meaninertia = tr / NV;
    }
    if (!cholesky(w.M, w.L)) return;
    static thread_local double Minv[NV][NV];
    for (int c = 0; c < NV; ++c) {
        double e[NV] = {0}, x[NV];
        e[c] = 1;
        chol_solve(w.L, e, x);
        for (int r = 0; r < NV; ++r) Minv[r][c] = x[r];
    }
    p.body_invweight0[0][0] = p.body_invweight0[0][1] = 0;
    for (int b = 1; b < NB; ++b) {
        double J[6][NV];
        std::memset(J, 0, sizeof(J));
        const V3 c = s.xpos[b] + mul(s.xmat[b], v3(cm_body_ipos + 3 * b));
        jac_point(w, b, c, J[0], J[1], J[2], 1.0);
        for (int d = body_lastdof(b); d >= 0; d = cm_dof_parent[d]) { J[3][d] = w.cdof[d].a.x; J[4][d] = w.cdof[d].a.y; J[5][d] = w.cdof[d].a.z; }
        double tr[2] = {0, 0};
        for (int k = 0; k < 6; ++k) {
            double acc = 0;
            for (int i = 0; i < NV; ++i)
                for (int j = 0; j < NV; ++j) acc += J[k][i] * Minv[i][j] * J[k][j];
            tr[k / 3] += acc;
        }
        p.body_invweight0[b][0] = tr[0] / 3; p.body_invweight0[b][1] = tr[1] / 3;
    }
    for (int j = 0; j < NJ; ++j) {
        const int a = cm_jnt_dofadr[j];
        if (cm_jnt_type[j] == 2) {
            const double v = (Minv[a][a] + Minv[a + 1][a + 1] + Minv[a + 2][a + 2]) / 3;
            p.dof_invweight0[a] = p.dof_invweight0[a + 1] = p.dof_invweight0[a + 2] = v;
        } else
            p.dof_invweight0[a] = Minv[a][a];
    }
}

void reset_state(State& s) {
    std::memset(&s, 0, sizeof(s));
    for (int i = 0; i < NQ; ++i) s.qpos[i] = cm_init_qpos[i];
}

// ---------------------------------------------------------------------------------------------- forward dynamics
void forward(const Params& p, State& s, Work& w, const double* ctrl) {
    kinematics(s.qpos, s, w);
    inertias(p, s, w);
    // velocities (mj_comVel)
    SV cvel_dofs[NB];
    w.cvel[0] = {{0, 0, 0}, {0, 0, 0}};
    for (int b = 1; b < NB; ++b) {
        SV v = w.cvel[cm_body_parent[b]];
        for (int j = 0; j < NJ; ++j) {
            if (cm_jnt_body[j] != b) continue;
            const int a = cm_jnt_dofadr[j], nd = cm_jnt_type[j] == 2 ? 3 : 1;
            for (int k = 0; k < nd; ++k) w.cdofdot[a + k] = crossMotion(v, w.cdof[a + k]);   // ball: all 3 from the same v
            for (int k = 0; k < nd; ++k) v = v + w.cdof[a + k] * s.qvel[a + k];
            CNT(nd * (27 + 12));
        }
        w.cvel[b] = v;
    }
    (void)cvel_dofs;
    // bias forces: RNE with qacc = 0 and base acceleration = -gravity (mj_rne)
    SV cacc[NB], cfrc[NB];
    cacc[0] = {{0, 0, 0}, {0, 0, GRAV}};
    for (int b = 1; b < NB; ++b) {
        SV a = cacc[cm_body_parent[b]];
        for (int d = cm_body_dofadr[b]; d < cm_body_dofadr[b] + cm_body_dofnum[b]; ++d) a = a + w.cdofdot[d] * s.qvel[d];
        cacc[b] = a;
        cfrc[b] = imul(w.cinert[b], a) + crossForce(w.cvel[b], imul(w.cinert[b], w.cvel[b]));
        CNT(cm_body_dofnum[b] * 12 + 45 + 45 + 27 + 6);
    }
    for (int b = NB - 1; b >= 1; --b)
        if (cm_body_parent[b] > 0) { cfrc[cm_body_parent[b]] = cfrc[cm_body_parent[b]] + cfrc[b]; CNT(6); }
    for (int d = 0; d < NV; ++d) { w.bias[d] = sdot(w.cdof[d], cfrc[cm_dof_body[d]]); CNT(11); }
    CNT(NV * 4 + NU * 3 + 40);      // passive, actuation, external wrench
    // passive: joint springs (springref = 0) and dampers (mj_passive)
    for (int d = 0; d < NV; ++d) {
        const int j = cm_dof_jnt[d];
        double f = -p.damping[d] * s.qvel[d];
        if (cm_jnt_type[j] != 2) f -= cm_jnt_stiffness[j] * s.qpos[cm_jnt_qposadr[j]];
        w.passive[d] = f;
    }
    // actuation: motors, ctrl clamped to ctrlrange, force = gear * ctrl (mj_fwdActuation)
    for (int d = 0; d < NV; ++d) w.smooth[d] = w.passive[d] - w.bias[d];
    for (int u = 0; u < NU; ++u) {
        const double c = std::min(std::max(ctrl ? ctrl[u] : 0.0, -cm_act_ctrlmax[u]), cm_act_ctrlmax[u]);
        w.smooth[cm_act_dof[u]] += cm_act_gear[u] * c;
    }
    {   // mj_xfrcAccumulate for the pushed body: J^T (f, tau) with the wrench acting at that body's xipos, over the dofs of its ancestor chain
        const int b = s.xfrc_body;
        const V3 f = {s.xfrc[0], s.xfrc[1], s.xfrc[2]}, t = {s.xfrc[3], s.xfrc[4], s.xfrc[5]};
        const V3 p = s.xpos[b] + mul(s.xmat[b], v3(cm_body_ipos + 3 * b));
        const V3 to = t + cross(p - w.o, f);
        int last = -1;
        for (int a = b; a >= 1 && last < 0; a = cm_body_parent[a]) if (cm_body_dofnum[a] > 0) last = cm_body_dofadr[a] + cm_body_dofnum[a] - 1;
        for (int d = last; d >= 0; d = cm_dof_parent[d]) w.smooth[d] += dot(w.cdof[d].a, to) + dot(w.cdof[d].l, f);
    }
    cholesky(w.M, w.L);
    double qacc_smooth[NV];
    chol_solve(w.L, w.smooth, qacc_smooth);

    // ------------------------------------------------------------------ constraint rows (mj_makeConstraint)
    // Row order is LEG-MAJOR: [left: 2 connects (6 rows), <=1 limit row, <=3 contacts (4 rows each)] then the same for the
    // right leg.  MuJoCo orders rows type-major (equality, limit, contact); projected Gauss-Seidel converges to the same
    // solution of the strictly convex dual for any sweep order, only the iterates differ.
    int n = 0;
    Row* rows = w.rows;
    s.ncon = 0; s.sat = 0; s.rowsig[0] = s.rowsig[1] = 0;
    int con_row[MAXCON];
    // signed distance of a sphere (centre ctr, radius rad) to the floor under it: the plane (cassie.xml:73, tilted by dynamics randomisation)
    // or the height-field triangle (cassie_hfield.xml:74); nrm = that surface's unit normal
    auto floor_dist = [&](V3 ctr, double rad, V3& nrm) {
        double hh;
        floor_query(p, ctr.x, ctr.y, hh, nrm);
        return (ctr.z - hh) * nrm.z - rad;       // (ctr - (x, y, h)) . n
    };
    // pyramidal floor contact (condim 3): 4 rows n +- mu t1, n +- mu t2 in the contact's own frame (mju_makeFrame)
    auto add_floor_contact = [&](int g, int b, V3 ctr, double dist, V3 nrm) {
        V3 t1 = std::fabs(nrm.y) < 0.5 ? V3{0, 1, 0} : V3{0, 0, 1};
        t1 = t1 - nrm * dot(nrm, t1); t1 = t1 * (1.0 / norm(t1));
        const V3 t2 = cross(nrm, t1);
        const V3 cp = ctr - nrm * (cm_geom_radius[g] + 0.5 * dist);
        double Jx[NV] = {0}, Jy[NV] = {0}, Jz[NV] = {0};
        jac_point(w, b, cp, Jx, Jy, Jz, 1.0);
        const double mu = p.friction;
        const double tran = p.body_invweight0[b][0];     // + world (0)
        const V3 dirs[4] = {nrm + t1 * mu, nrm - t1 * mu, nrm + t2 * mu, nrm - t2 * mu};
        con_row[s.ncon] = n; s.con_dist[s.ncon] = dist; s.con_geom[s.ncon] = g;
        s.con_frame[s.ncon][0] = nrm; s.con_frame[s.ncon][1] = t1; s.con_frame[s.ncon][2] = t2;
        for (int k = 0; k < 4; ++k) {
            Row& r = rows[n + k];
            for (int d = 0; d < NV; ++d) { CNT(Jx[d] != 0.0 || Jy[d] != 0.0 || Jz[d] != 0.0 ? 5 : 0); r.J[d] = dirs[k].x * Jx[d] + dirs[k].y * Jy[d] + dirs[k].z * Jz[d]; }
            r.pos = dist; r.type = 2; r.diag = tran + mu * mu * tran;
            finish_row(r, s.qvel, dist, 0.005, 1.0);
        }
        // pyramidal regulariser: all rows of the contact share Rpy = 2 mu^2 R(first row), impratio = 1
        const double Rpy = std::max(MINVAL, 2 * mu * mu * rows[n].R);
        for (int k = 0; k < 4; ++k) rows[n + k].R = Rpy;
        n += 4; ++s.ncon;
    };
    for (int leg = 0; leg < 2; ++leg) {
        const int body_lo = leg == 0 ? 2 : 14, body_hi = leg == 0 ? 13 : 25;
        for (int e = 2 * leg; e < 2 * leg + 2; ++e) {   // connect equalities, cassie.xml:225-230
            const int b1 = cm_eq_body1[e], b2 = cm_eq_body2[e];
            const V3 p1 = s.xpos[b1] + mul(s.xmat[b1], v3(cm_eq_anchor1 + 3 * e));
            const V3 p2 = s.xpos[b2] + mul(s.xmat[b2], v3(cm_eq_anchor2 + 3 * e));
            const V3 c = p1 - p2;
            for (int k = 0; k < 3; ++k) std::memset(rows[n + k].J, 0, sizeof(rows[n + k].J));
            jac_point(w, b1, p1, rows[n].J, rows[n + 1].J, rows[n + 2].J, 1.0);
            jac_point(w, b2, p2, rows[n].J, rows[n + 1].J, rows[n + 2].J, -1.0);
            const double cp[3] = {c.x, c.y, c.z};
            const double tran = p.body_invweight0[b1][0] + p.body_invweight0[b2][0];
            for (int k = 0; k < 3; ++k) {
                rows[n + k].pos = cp[k]; rows[n + k].type = 0; rows[n + k].diag = tran;
                finish_row(rows[n + k], s.qvel, norm(c), 0.005, 1.0);
            }
            n += 3;
        }
        int nlim = 0;   // joint limits (mj_instantiateLimit), solreflimit default 0.02 1: every active one
        int lbit = -1;
        for (int j = 0; j < NJ && nlim < MAXLIM_LEG; ++j) {
            if (!cm_jnt_limited[j] || cm_jnt_body[j] < body_lo || cm_jnt_body[j] > body_hi) continue;
            ++lbit;
            const double q = s.qpos[cm_jnt_qposadr[j]];
            for (int side = 0; side < 2 && nlim < MAXLIM_LEG; ++side) {
                const double dist = side == 0 ? q - cm_jnt_range[2 * j] : cm_jnt_range[2 * j + 1] - q;
                if (dist >= 0) continue;
                s.rowsig[0] |= 1u << (lbit + 8 * leg);
                if (nlim >= KERNEL_MAXLIM_LEG) { s.sat |= SAT_LIMITS; if (p.kernel_caps) continue; }
                Row& r = rows[n];
                std::memset(r.J, 0, sizeof(r.J));
                r.J[cm_jnt_dofadr[j]] = side == 0 ? 1.0 : -1.0;
                r.pos = dist; r.type = 1; r.diag = p.dof_invweight0[cm_jnt_dofadr[j]];
                finish_row(r, s.qvel, dist, 0.02, 1.0);
                ++n; ++nlim;
            }
        }
        // contacts: foot / tarsus / shin / hip-pitch capsules of this leg vs the floor plane (mjc_PlaneCapsule: one contact per
        // penetrating end), pyramidal cone, condim 3 (the floor's condim / friction win through its priority, cassie.xml:73)
        int ncl = 0;
        for (int g = leg; g < 8 && ncl < MAXCON_LEG; g += 2) {
            const int b = cm_geom_body[g];
            const V3 c = s.xpos[b] + mul(s.xmat[b], v3(cm_geom_pos + 3 * g));
            const V3 ax = mul(s.xmat[b], v3(cm_geom_axis + 3 * g));
            for (int e = 0; e < 2 && ncl < MAXCON_LEG; ++e) {
                const V3 ctr = c + ax * (e == 0 ? cm_geom_half[g] : -cm_geom_half[g]);
                V3 nrm;
                const double dist = floor_dist(ctr, cm_geom_radius[g], nrm);
                if (dist >= 0) continue;
                s.rowsig[0] |= g >= 6 ? (2u << 28) : (1u << (16 + 6 * leg + (g / 2) * 2 + e));
                if (g >= 6) { s.sat |= SAT_BODY_FLOOR; if (p.kernel_caps) continue; }
                else if (ncl >= KERNEL_MAXCON_LEG) { s.sat |= SAT_CONTACTS; if (p.kernel_caps) continue; }
                add_floor_contact(g, b, ctr, dist, nrm);
                ++ncl;
            }
        }
    }
    {   // pelvis sphere vs the floor (cassie.xml:87; mjc_PlaneSphere), pyramidal like the other floor contacts
        const int g = 8, b = cm_geom_body[g];
        const V3 ctr = s.xpos[b] + mul(s.xmat[b], v3(cm_geom_pos + 3 * g));
        V3 nrm;
        const double dist = floor_dist(ctr, cm_geom_radius[g], nrm);
        if (dist < 0) { s.sat |= SAT_BODY_FLOOR; s.rowsig[0] |= 1u << 28; }
        if (dist < 0 && !p.kernel_caps) add_floor_contact(g, b, ctr, dist, nrm);
    }
    // left-leg vs right-leg capsules (contype 2 / conaffinity 4 against contype 4 / conaffinity 2, cassie.xml:23-35): foot, tarsus, shin
    // of one leg against foot, tarsus, shin of the other, condim 1 (frictionless): one unilateral row along the contact normal
    s.ncon1 = 0;
    for (int gl = 0; gl < 6; gl += 2)
        for (int gr = 1; gr < 6; gr += 2) {
            const int bl = cm_geom_body[gl], br = cm_geom_body[gr];
            const V3 cl = s.xpos[bl] + mul(s.xmat[bl], v3(cm_geom_pos + 3 * gl)), al = mul(s.xmat[bl], v3(cm_geom_axis + 3 * gl)) * cm_geom_half[gl];
            const V3 cr = s.xpos[br] + mul(s.xmat[br], v3(cm_geom_pos + 3 * gr)), ar = mul(s.xmat[br], v3(cm_geom_axis + 3 * gr)) * cm_geom_half[gr];
            V3 c1, c2;
            closest_seg_seg(cl - al, cl + al, cr - ar, cr + ar, c1, c2);
            const V3 dv = c2 - c1;
            const double len = norm(dv), dist = len - cm_geom_radius[gl] - cm_geom_radius[gr];
            if (dist >= 0 || len < MINVAL) continue;
            s.rowsig[1] |= 1u << (3 * (gl / 2) + gr / 2);
            if (s.ncon1 >= KERNEL_MAXLEGLEG) { s.sat |= SAT_LEG_LEG; if (p.kernel_caps) continue; }
            const V3 nn = dv * (1.0 / len);                                     // from the left geom to the right geom
            const V3 cp = c1 + nn * (cm_geom_radius[gl] + 0.5 * dist);
            double Jx[NV] = {0}, Jy[NV] = {0}, Jz[NV] = {0};
            jac_point(w, br, cp, Jx, Jy, Jz, 1.0);
            jac_point(w, bl, cp, Jx, Jy, Jz, -1.0);
            Row& r = rows[n];
            for (int d = 0; d < NV; ++d) r.J[d] = nn.x * Jx[d] + nn.y * Jy[d] + nn.z * Jz[d];
            r.pos = dist; r.type = 2; r.diag = p.body_invweight0[bl][0] + p.body_invweight0[br][0];
            finish_row(r, s.qvel, dist, 0.005, 1.0);
            ++n; ++s.ncon1;
        }
    s.nefc = n;

    // ------------------------------------------------------------------ dual problem + PGS (mj_projectConstraint, mj_solPGS)
    static thread_local double MiJ[MAXEFC][NV], AR[MAXEFC][MAXEFC];
    double b[MAXEFC], f[MAXEFC];
    for (int i = 0; i < n; ++i) chol_solve(w.L, rows[i].J, MiJ[i]);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            double a = 0;
            for (int d = 0; d < NV; ++d) { CNT(NZ(rows[i].J[d], MiJ[j][d])); a += rows[i].J[d] * MiJ[j][d]; }
            AR[i][j] = a;
        }
        AR[i][i] += rows[i].R;
        double a = 0;
        for (int d = 0; d < NV; ++d) { CNT(NZ(rows[i].J[d], qacc_smooth[d])); a += rows[i].J[d] * qacc_smooth[d]; }
        b[i] = a - rows[i].aref;
    }
    // warm start from the previous qacc (mj_fwdConstraint + mj_constraintUpdate), kept only if it beats f = 0
    for (int i = 0; i < n; ++i) {
        double jar = -rows[i].aref;
        for (int d = 0; d < NV; ++d) { CNT(NZ(rows[i].J[d], s.qacc_warm[d])); jar += rows[i].J[d] * s.qacc_warm[d]; }
        double fi = -jar / rows[i].R;
        if (rows[i].type != 0 && fi < 0) fi = 0;
        f[i] = fi;
    }
    double cost = 0;
    for (int i = 0; i < n; ++i) {
        double a = 0;
        for (int j = 0; j < n; ++j) { CNT(NZ(AR[i][j], f[j])); a += AR[i][j] * f[j]; }
        cost += f[i] * (0.5 * a + b[i]);
    }
    if (cost > 0) for (int i = 0; i < n; ++i) f[i] = 0;
    // mj_solPGS: scalar Gauss-Seidel sweeps; a sweep's cost decrease, made dimensionless by 1 / (meaninertia * max(1, nv)), is compared with
    // opt.tolerance after the sweep and the solver stops early once it falls below (at most opt.iterations = 50 sweeps, cassie.xml:5)
    const double scale = 1.0 / (p.meaninertia * NV);
    s.solver_iter = 0;
    for (int it = 0; it < p.pgs_iters; ++it) {
        double improvement = 0;
        for (int i = 0; i < n; ++i) {
            double res = b[i];
            for (int j = 0; j < n; ++j) res += AR[i][j] * f[j];
            CNT(2 * n + 3);                                  // the residual of a row touches every multiplier (A is dense: every row reaches the pelvis)
            double fi = f[i] - res / AR[i][i];
            if (rows[i].type != 0 && fi < 0) fi = 0;
            const double df = fi - f[i];
            improvement -= df * (0.5 * df * AR[i][i] + res); // costChange: the dual cost 1/2 f'(A + R) f + f'b moved by df along row i
            f[i] = fi;
        }
        ++s.solver_iter;
        if (p.tolerance > 0 && improvement * scale < p.tolerance) break;
    }
    for (int d = 0; d < NV; ++d) {
        double a = qacc_smooth[d];
        for (int i = 0; i < n; ++i) { CNT(NZ(MiJ[i][d], f[i])); a += MiJ[i][d] * f[i]; }
        s.qacc[d] = a;
    }
    for (int i = 0; i < n; ++i) { s.efc_force[i] = f[i]; s.efc_type[i] = rows[i].type; }
    // contact force on the foot bodies in world axes (cassie_sim_foot_forces: mj_contactForce summed per foot body)
    std::memset(s.foot_force, 0, sizeof(s.foot_force));
    for (int c = 0; c < s.ncon; ++c) {
        const int b = cm_geom_body[s.con_geom[c]];
        const int foot = b == 13 ? 0 : (b == 25 ? 1 : -1);   // foot bodies (cassie.xml:140,203)
        if (foot < 0) continue;
        const double* ff = f + con_row[c];
        const double fn = ff[0] + ff[1] + ff[2] + ff[3], f1 = p.friction * (ff[0] - ff[1]), f2 = p.friction * (ff[2] - ff[3]);
        const V3 F = s.con_frame[c][0] * fn + s.con_frame[c][1] * f1 + s.con_frame[c][2] * f2;
        s.foot_force[foot][0] += F.x; s.foot_force[foot][1] += F.y; s.foot_force[foot][2] += F.z;
    }
    // IMU (cassie.xml:265-268): gyro = pelvis angular velocity in the site (= pelvis) frame; accelerometer = classical
    // acceleration of the site point minus gravity, site frame
    {
        SV A = cacc[0];
        for (int d = 0; d < 6; ++d) A = A + w.cdofdot[d] * s.qvel[d] + w.cdof[d] * s.qacc[d];
        const V3 r = mul(s.xmat[1], v3(cm_imu_pos));
        const V3 om = w.cvel[1].a;
        const V3 vp = w.cvel[1].l + cross(om, r);
        const V3 a = A.l + cross(A.a, r) + cross(om, vp);
        const M3& R = s.xmat[1];
        s.sens_acc[0] = dot(col(R, 0), a); s.sens_acc[1] = dot(col(R, 1), a); s.sens_acc[2] = dot(col(R, 2), a);
        s.sens_gyro[0] = s.qvel[3]; s.sens_gyro[1] = s.qvel[4]; s.sens_gyro[2] = s.qvel[5];
    }
}

// mj_Euler: damping treated implicitly, (M + h D) a = M qacc; velocity first, then positions with the new velocity
void euler(const Params& p, State& s, Work& w) {
    double rhs[NV], a[NV];
    for (int i = 0; i < NV; ++i) {
        double acc = 0;
        for (int j = 0; j < NV; ++j) { CNT(NZ(w.M[i][j], s.qacc[j])); acc += w.M[i][j] * s.qacc[j]; }
        rhs[i] = acc;
    }
    static thread_local double MM[NV][NV], LL[NV][NV];
    for (int i = 0; i < NV; ++i) {
        for (int j = 0; j < NV; ++j) MM[i][j] = w.M[i][j];
        MM[i][i] += DT * p.damping[i];
    }
    cholesky(MM, LL);
    chol_solve(LL, rhs, a);
    for (int d = 0; d < NV; ++d) { s.qacc_warm[d] = s.qacc[d]; s.qvel[d] += DT * a[d]; }
    CNT(NV * 2 + NQ * 2 + 3 * 60);      // integration (three ball joints)
    for (int j = 0; j < NJ; ++j) {
        const int qa = cm_jnt_qposadr[j], da = cm_jnt_dofadr[j];
        if (cm_jnt_type[j] != 2) { s.qpos[qa] += DT * s.qvel[da]; continue; }
        const V3 wv = {s.qvel[da], s.qvel[da + 1], s.qvel[da + 2]};      // local-frame angular velocity (mju_quatIntegrate)
        const double ang = norm(wv) * DT;
        Q4 q = {s.qpos[qa], s.qpos[qa + 1], s.qpos[qa + 2], s.qpos[qa + 3]};
        if (ang > 0) q = qmul(q, qaxisangle(wv * (1.0 / norm(wv)), ang));
        q = qnormalize(q);
        s.qpos[qa] = q.w; s.qpos[qa + 1] = q.x; s.qpos[qa + 2] = q.y; s.qpos[qa + 3] = q.z;
    }
}

double constraint_violation(const State& s) {
    double m = 0;
    for (int e = 0; e < NEQ; ++e) {
        const int b1 = cm_eq_body1[e], b2 = cm_eq_body2[e];
        const V3 p1 = s.xpos[b1] + mul(s.xmat[b1], v3(cm_eq_anchor1 + 3 * e));
        const V3 p2 = s.xpos[b2] + mul(s.xmat[b2], v3(cm_eq_anchor2 + 3 * e));
        m = std::max(m, norm(p1 - p2));
    }
    return m;
}

// total linear momentum / total mass: the translational rows of the free joint are world-aligned, so p = M[0:3, :] qvel
void com_velocity(const Params& p, const State& s0, Work& w, double out[3]) {
    State s = s0;
    kinematics(s.qpos, s, w);
    inertias(p, s, w);
    double m = 0;
    for (int b = 1; b < NB; ++b) m += p.mass[b];
    for (int k = 0; k < 3; ++k) {
        double a = 0;
        for (int j = 0; j < NV; ++j) a += w.M[k][j] * s.qvel[j];
        out[k] = a / m;
    }
}

void momentum(const Params& p, const State& s0, Work& w, double out[25]) {
    State s = s0;
    kinematics(s.qpos, s, w);
    inertias(p, s, w);
    double row[6];
    for (int k = 0; k < 6; ++k) {
        double a = 0;
        for (int j = 0; j < NV; ++j) a += w.M[k][j] * s.qvel[j];
        row[k] = a;
    }
    V3 L = {0, 0, 0};
    for (int k = 0; k < 3; ++k) L = L + col(s.xmat[1], k) * row[3 + k];      // the ball joint's dofs are rotations about the pelvis body axes, taken about the pelvis origin (w.o)
    out[0] = row[0]; out[1] = row[1]; out[2] = row[2]; out[3] = L.x; out[4] = L.y; out[5] = L.z;
    out[6] = w.o.x; out[7] = w.o.y; out[8] = w.o.z;
    const SI& c = w.crb[1];
    out[9] = c.m; out[10] = w.o.x + c.h.x / c.m; out[11] = w.o.y + c.h.y / c.m; out[12] = w.o.z + c.h.z / c.m;
    for (int leg = 0; leg < 2; ++leg) {
        const int g = leg, b = cm_geom_body[g];
        const V3 ctr = s.xpos[b] + mul(s.xmat[b], v3(cm_geom_pos + 3 * g)), ax = mul(s.xmat[b], v3(cm_geom_axis + 3 * g));
        double* o = out + 13 + 6 * leg;
        o[0] = ctr.x; o[1] = ctr.y; o[2] = ctr.z; o[3] = ax.x; o[4] = ax.y; o[5] = ax.z;
    }
}

void inverse_dynamics(const Params& p, const State& s0, Work& w, const double* qacc, double* out) {
    State s = s0;
    s.xfrc_body = 0;
    forward(p, s, w, nullptr);      // kinematics, M, bias, passive (the constraint part of the pass is not used)
    for (int d = 0; d < NV; ++d) {
        double a = w.bias[d] - w.passive[d];
        for (int j = 0; j < NV; ++j) a += w.M[d][j] * qacc[j];
        out[d] = a;
    }
}

double total_energy(const Params& p, const State& s0, Work& w) {
    State s = s0;
    kinematics(s.qpos, s, w);
    inertias(p, s, w);
    double ke = 0;
    for (int i = 0; i < NV; ++i)
        for (int j = 0; j < NV; ++j) ke += 0.5 * s.qvel[i] * w.M[i][j] * s.qvel[j];
    double pe = 0;
    for (int b = 1; b < NB; ++b) pe += p.mass[b] * GRAV * (s.xpos[b] + mul(s.xmat[b], v3(cm_body_ipos + 3 * b))).z;
    for (int j = 0; j < NJ; ++j)
        if (cm_jnt_type[j] != 2) pe += 0.5 * cm_jnt_stiffness[j] * s.qpos[cm_jnt_qposadr[j]] * s.qpos[cm_jnt_qposadr[j]];
    return ke + pe;
}

}  // namespace orc
