// CPU ORACLE (test infrastructure only): fp64 restatement of the reference's state estimator.  See cassie_estimator.h.
#include "cassie_estimator.h"

namespace orc {

// instrumented operation count, same convention as cassie_phys.cpp (multiplies + adds where they happen, structural zeros skipped, one per transcendental)
#define CNT(n) (g_flops += (unsigned long long)(n))
#define NZ(a, b) (((a) != 0.0 && (b) != 0.0) ? 2 : 0)

namespace {
constexpr double EST_DT = 0.0005, EST_G = 9.806, EST_H = 1.0, EST_M = 31.0;      // filter constants of state_output_setup (object dump: dt, g, pendulum height, mass)
constexpr double K_SHIN = 1500.0, K_HEEL = 1250.0;                               // leg-spring stiffnesses the routine uses (cassie.xml:117,127)
constexpr double FOOT_OFF[3] = {0.017620176, 0.052189981, 0.0};                  // origin of the routine's foot frame inside the foot body (probed: 80 random poses, spread 2e-8)
constexpr double IMU_R[3] = {0.03155, 0.0, -0.079996};                           // IMU offset the routine uses (least squares on its own output, residual 5e-15; cassie.xml:265 has -0.07996)
constexpr double LM_TAU = 1e-3, LM_KMAX = 5.0, LM_EPS = 1.4901161193847656e-08;  // object dump: tau, iteration limit, sqrt(eps) tolerances
constexpr double HEEL_LB = -0.78539816339744828 + 1e-6, HEEL_UB = 0.78539816339744828 - 1e-6;
// achilles-rod closure as the routine evaluates it (0x18e00): r = sum_k c_k cos(nK knee + nS shin + nT tarsus + nX heel + phi_k) + c0, constants from .rodata 0x2f7c0-0x2f950
const double kHeelTerm[14][6] = {   // nK nS nT nX, coefficient, phase
    {0, 0, 1, -1, -0.00015856770032083081, 0.60103412848472038},
    {1, 1, 1, 0, -0.0079482557784711465, 1.1775614272403667},
    {0, 0, 0, 1, -0.0085406453671835261, 1.4203984658003772},
    {1, 1, 1, -1, -4.3720724282115655e-05, 0.64700411135490599},
    {1, 0, 0, 0, 0.018481200740211659, 0.66324136802307054},
    {0, 0, 1, 0, -0.02882698448043499, 1.1315914443701811},
    {1, 1, 1, 1, -0.02856193646166626, -0.50615334252706334},
    {0, 0, 0, 2, -7.543150810775465e-19, 0.34667816485989922},
    {0, 1, 1, 1, -0.018328433377681426, -1.1693947105501337},
    {0, 1, 0, 0, 0.067028201125701736, -0.61727138515288482},
    {1, 1, 0, 0, 0.1044527473921103, 0.045969982870185652},
    {0, 1, 1, 0, -0.0051004621727941879, 0.51432005921729607},
    {0, 1, 1, -1, -2.8055954234903667e-05, -0.016237256668164488},
    {0, 0, 1, 1, -0.10358933105069197, -0.55212332539724895},
};
const double kHeelConst = -0.024585738400000001;
const int kLegBody[2][7] = {{2, 3, 4, 6, 8, 9, 13}, {14, 15, 16, 18, 20, 21, 25}};   // hip roll, yaw, pitch, knee, shin, tarsus, foot (cassie.xml:89-140,152-203)

inline M3 matmul(const M3& a, const M3& b) {
    M3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return r;
}
inline V3 mulT(const M3& R, V3 v) { return {R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z, R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z}; }

// foot-frame origin relative to the pelvis (pelvis axes) and its partial derivatives with respect to the shin and tarsus angles
void foot_kinematics(int leg, const double q[7], V3& p, V3& dshin, V3& dtarsus, M3& Rfoot) {
    M3 R = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    V3 o = {0, 0, 0}, jo[7]; M3 jR[7];
    for (int k = 0; k < 7; ++k) {
        const int b = kLegBody[leg][k];
        o = o + mul(R, v3(cm_body_pos + 3 * b));
        const double ang = q[k] - cm_jnt_ref[cm_body_jntadr[b]];
        const double c = std::cos(ang), s = std::sin(ang);
        const M3 Rz = {{c, -s, 0, s, c, 0, 0, 0, 1}};
        R = matmul(matmul(R, q2m(Q4{cm_body_quat[4 * b], cm_body_quat[4 * b + 1], cm_body_quat[4 * b + 2], cm_body_quat[4 * b + 3]})), Rz);
        jo[k] = o; jR[k] = R; CNT(15 + 3 + 2 + 45 + 36);      // origin, angle, sin / cos, two 3 x 3 products (the joint rotation has 4 non-trivial entries)
    }
    p = o + mul(R, v3(FOOT_OFF)); Rfoot = R; CNT(15 + 2 * (9 + 3));
    dshin = cross(col(jR[4], 2), p - jo[4]);
    dtarsus = cross(col(jR[5], 2), p - jo[5]);
}

// one scalar measurement z = x[i] - x[j] (j < 0: z = x[i]) with variance r; sequential processing of a diagonal-R update is the batch update
void kf_scalar_update(int n, double* x, double* P, int i, int j, double z, double r) {
    double Ph[6], hP[6];
    for (int a = 0; a < n; ++a) { Ph[a] = P[a * n + i] - (j >= 0 ? P[a * n + j] : 0.0); hP[a] = P[i * n + a] - (j >= 0 ? P[j * n + a] : 0.0); }
    const double s = Ph[i] - (j >= 0 ? Ph[j] : 0.0) + r, innov = z - (x[i] - (j >= 0 ? x[j] : 0.0));
    CNT((j >= 0 ? 2 * n : 0) + 4);
    for (int a = 0; a < n; ++a) {
        const double K = Ph[a] / s;
        x[a] += K * innov; CNT(Ph[a] != 0.0 ? 3 : 0);
        for (int b = 0; b < n; ++b) { CNT(NZ(K, hP[b])); P[a * n + b] -= K * hP[b]; }
    }
}
void apa(int n, const double* A, double* P) {      // P <- A P A^T
    double T[36], R[36];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) { CNT(i == k ? 0 : NZ(A[i * n + k], P[k * n + j])); s += A[i * n + k] * P[k * n + j]; } T[i * n + j] = s; }      // (the unit diagonal of A costs nothing)
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) { CNT(j == k ? 0 : NZ(T[i * n + k], A[j * n + k])); s += T[i * n + k] * A[j * n + k]; } R[i * n + j] = s; }
    for (int i = 0; i < n * n; ++i) P[i] = R[i];
}
}  // namespace

// horizontal filter step (0x1cd10): extended Kalman filter on the linear inverted pendulum
void hfilter_step(double* x, double* P, double zL, double zR, double fl, double fr, double acc) {
    const double tot = fl + fr;
    const bool contact = !(1.0 > tot);
    const double alpha_m = contact ? fl / tot : 0.5;
    const double p = x[0], v = x[1], pL = x[2], pR = x[3], al = x[4], fd = x[5], w2 = EST_G / EST_H;
    double A[36] = {0};
    for (int i = 0; i < 6; ++i) A[7 * i] = 1;
    A[1] = EST_DT;
    x[0] = p + EST_DT * v;
    if (contact) {
        x[1] = v + EST_DT * (w2 * (p - al * pL - (1 - al) * pR) + fd / EST_M); CNT(11 + 12);
        A[6] = EST_DT * w2; A[8] = -EST_DT * w2 * al; A[9] = -EST_DT * w2 * (1 - al); A[10] = -EST_DT * w2 * (pL - pR); A[11] = EST_DT / EST_M;
    }
    apa(6, A, P);
    const double Q[6] = {1e-8, 1e-8, 50.0 > fl ? 1e-6 : 1e-10, 50.0 > fr ? 1e-6 : 1e-10, 1e-5, 1e-2};
    for (int i = 0; i < 6; ++i) P[7 * i] += Q[i];
    // the batch update linearises at the predicted state; the velocity pseudo-measurement is the PRIOR velocity advanced by the IMU acceleration
    double xs[6], Ps[36];
    for (int i = 0; i < 6; ++i) xs[i] = x[i];
    for (int i = 0; i < 36; ++i) Ps[i] = P[i];
    // batch update with diagonal R == sequential scalar updates (exact in exact arithmetic)
    kf_scalar_update(6, xs, Ps, 0, 2, zL, 1e-6);
    kf_scalar_update(6, xs, Ps, 0, 3, zR, 1e-6);
    kf_scalar_update(6, xs, Ps, 4, -1, alpha_m, 1e-6);
    kf_scalar_update(6, xs, Ps, 1, -1, v + EST_DT * acc, 1.0);
    for (int i = 0; i < 6; ++i) x[i] = xs[i];
    for (int i = 0; i < 36; ++i) P[i] = Ps[i];
}
// vertical filter step (inline in state_output_step, 0x2a858-0x2c607)
void zfilter_step(double* x, double* P, double zL, double zR, double fl, double fr) {
    double A[25] = {0};
    for (int i = 0; i < 5; ++i) A[6 * i] = 1;
    A[1] = EST_DT; A[9] = EST_DT / EST_M;
    const double u = (fl + fr) / EST_M - EST_G;
    const double x0 = x[0] + EST_DT * x[1], x1 = x[1] + EST_DT / EST_M * x[4] + EST_DT * u;
    x[0] = x0; x[1] = x1;
    apa(5, A, P);
    const double Q[5] = {1e-8, 1e-8, 50.0 > fl ? 1e-6 : 1e-10, 50.0 > fr ? 1e-6 : 1e-10, 0.01};
    for (int i = 0; i < 5; ++i) P[6 * i] += Q[i];
    kf_scalar_update(5, x, P, 0, 2, zL, 1e-6);
    kf_scalar_update(5, x, P, 0, 3, zR, 1e-6);
}

double heel_residual(double knee, double shin, double tarsus, double heel, double* grad4) {
    double r = kHeelConst;
    if (grad4) grad4[0] = grad4[1] = grad4[2] = grad4[3] = 0;
    for (const auto& t : kHeelTerm) {
        const double u = t[0] * knee + t[1] * shin + t[2] * tarsus + t[3] * heel + t[5];
        r += t[4] * std::cos(u); CNT(2 * ((t[0] != 0) + (t[1] != 0) + (t[2] != 0) + (t[3] != 0)) + 1 + 3);
        if (grad4) { const double ds = -t[4] * std::sin(u); for (int k = 0; k < 4; ++k) { CNT(t[k] != 0 ? 2 : 0); grad4[k] += ds * t[k]; } CNT(2); }
    }
    return r;
}

// Levenberg-Marquardt with Nielsen's damping update (Madsen, Nielsen, Tingleff: Methods for non-linear least squares problems, alg. 3.16), as
// the routine runs it on the two decoupled closure residuals with ONE shared damping parameter, from the previous solution (0x22a93-0x2316d)
void heel_solve(double heel[2], const double legL[3], const double legR[3], int* iters) {
    const double* leg[2] = {legL, legR};
    double x[2], r[2], J[2], g[2], A[2];
    auto eval = [&](const double* xx, double* rr, double* JJ) {
        for (int i = 0; i < 2; ++i) { double gr[4]; rr[i] = heel_residual(leg[i][0], leg[i][1], leg[i][2], xx[i], gr); JJ[i] = gr[3]; }
    };
    for (int i = 0; i < 2; ++i) x[i] = std::min(std::max(heel[i], HEEL_LB), HEEL_UB);
    eval(x, r, J);
    for (int i = 0; i < 2; ++i) { g[i] = J[i] * r[i]; A[i] = J[i] * J[i]; }
    int k = 0;
    if (!(std::max(std::fabs(g[0]), std::fabs(g[1])) <= LM_EPS)) {
        double mu = LM_TAU * std::max(A[0], A[1]), nu = 2.0, F = 0.5 * (r[0] * r[0] + r[1] * r[1]);
        bool stop = false;
        while (!stop) {
            ++k;
            if ((double)k > LM_KMAX) stop = true;                     // the body still runs once more after the limit is passed (0x22cd0-0x22ce9)
            const double h[2] = {-g[0] / (A[0] + mu), -g[1] / (A[1] + mu)};
            if (std::hypot(h[0], h[1]) <= LM_EPS * (std::hypot(x[0], x[1]) + LM_EPS)) break;
            double xn[2] = {std::min(std::max(x[0] + h[0], HEEL_LB), HEEL_UB), std::min(std::max(x[1] + h[1], HEEL_LB), HEEL_UB)}, rn[2], Jn[2];
            eval(xn, rn, Jn);
            const double Fn = 0.5 * (rn[0] * rn[0] + rn[1] * rn[1]);
            const double rho = (F - Fn) / (0.5 * (h[0] * (mu * h[0] - g[0]) + h[1] * (mu * h[1] - g[1]))); CNT(4 + 6 + 3 + 12 + 6);
            if (rho > 0) {
                for (int i = 0; i < 2; ++i) { x[i] = xn[i]; r[i] = rn[i]; J[i] = Jn[i]; g[i] = J[i] * r[i]; A[i] = J[i] * J[i]; }
                F = Fn;
                if (std::max(std::fabs(g[0]), std::fabs(g[1])) <= LM_EPS) break;
                const double c = 2 * rho - 1;
                mu *= std::max(1.0 / 3.0, 1.0 - c * c * c); nu = 2.0;
            } else { mu *= nu; nu *= 2.0; }
        }
    }
    heel[0] = x[0]; heel[1] = x[1];
    if (iters) *iters = k;
}

// x = M \ tau for a full-rank 2 x 3 M the way MATLAB's mldivide does it (0x21000): Householder QR with column pivoting, then the basic solution
// on the two pivot columns (the third component is exactly 0)
void mldivide23(const double M[2][3], const double tau[2], double x[3]) {
    int j1 = 0; double best = -1;
    for (int j = 0; j < 3; ++j) { const double n = M[0][j] * M[0][j] + M[1][j] * M[1][j]; if (n > best) { best = n; j1 = j; } }
    const double n1 = std::sqrt(best), e0 = M[0][j1] / n1, e1 = M[1][j1] / n1;
    int j2 = -1; best = -1;
    for (int j = 0; j < 3; ++j) {
        if (j == j1) continue;
        const double pr = e0 * M[0][j] + e1 * M[1][j], a = M[0][j] - e0 * pr, b = M[1][j] - e1 * pr, n = a * a + b * b;
        if (n > best) { best = n; j2 = j; }
    }
    CNT(9 + 2 * 14 + 3 + 8);      // column norms, projections, 2 x 2 solve
    const double det = M[0][j1] * M[1][j2] - M[0][j2] * M[1][j1];
    x[0] = x[1] = x[2] = 0;
    x[j1] = (tau[0] * M[1][j2] - M[0][j2] * tau[1]) / det;
    x[j2] = (M[0][j1] * tau[1] - tau[0] * M[1][j1]) / det;
}

void state_output_setup(StateOutput& s) { std::memset(&s, 0, sizeof(s)); }

void state_output_step(StateOutput& s, const EstSensors& in) {
    // --- heel springs (both legs in one damped Gauss-Newton solve)
    const double legL[3] = {in.mpos[3], in.jpos[0], in.jpos[1]}, legR[3] = {in.mpos[8], in.jpos[3], in.jpos[4]};
    heel_solve(s.heel, legL, legR, &s.lm_iters);
    const M3 R = q2m(Q4{in.quat[0], in.quat[1], in.quat[2], in.quat[3]});
    // --- per leg: foot position, spring Jacobian with the tarsus on the rod closure, foot force
    V3 fw[2]; double fz[2];
    for (int leg = 0; leg < 2; ++leg) {
        const double q[7] = {in.mpos[5 * leg], in.mpos[5 * leg + 1], in.mpos[5 * leg + 2], in.mpos[5 * leg + 3], in.jpos[3 * leg], in.jpos[3 * leg + 1], in.mpos[5 * leg + 4]};
        V3 p, dS, dT; double gr[4]; M3 Rf;
        foot_kinematics(leg, q, p, dS, dT, Rf);
        {   // foot orientation output: foot body frame times the routine's constant frame offset C = [[-c, 0, -s], [s, 0, -c], [0, -1, 0]], c = cos 40 deg, s = sin 40 deg
            const double c40 = 0.76604444311897803, s40 = 0.64278760968653933;
            const M3 C = {{-c40, 0, -s40, s40, 0, -c40, 0, -1, 0}};
            const M3 Re = matmul(Rf, C);
            const double w = 0.5 * std::sqrt(std::max(0.0, 1 + Re.m[0] + Re.m[4] + Re.m[8]));      // the foot frame stays within a few degrees of the pelvis frame: w ~ 1
            s.foot_quat[leg][0] = w; s.foot_quat[leg][1] = (Re.m[7] - Re.m[5]) / (4 * w); s.foot_quat[leg][2] = (Re.m[2] - Re.m[6]) / (4 * w); s.foot_quat[leg][3] = (Re.m[3] - Re.m[1]) / (4 * w);
        }
        heel_residual(q[3], q[4], q[5], s.heel[leg], gr);
        CNT(2 * 9 + 8 + 9 + 15);           // two cross products (foot_kinematics), closure Jacobian, rotation of the force
        const V3 a = dS - dT * (gr[1] / gr[2]), b = dT * (-gr[3] / gr[2]);       // d foot / d shin, d foot / d heel spring: the tarsus angle follows the closure
        const double M[2][3] = {{-a.x, -a.y, -a.z}, {-b.x, -b.y, -b.z}}, tau[2] = {K_SHIN * in.jpos[3 * leg], K_HEEL * s.heel[leg]};
        double f[3];
        mldivide23(M, tau, f);
        const V3 fwv = mul(R, V3{f[0], f[1], f[2]});
        s.foot_force[leg][0] = fwv.x; s.foot_force[leg][1] = fwv.y; s.foot_force[leg][2] = fwv.z;
        fz[leg] = std::fmax(0.0, -fwv.z);
        s.foot_rel[leg][0] = p.x; s.foot_rel[leg][1] = p.y; s.foot_rel[leg][2] = p.z;
        fw[leg] = mul(R, p);
    }
    // --- translationalAcceleration (pelvis frame) and its world-aligned copy for the velocity pseudo-measurement
    const V3 w = v3(in.gyro), cen = cross(w, cross(w, v3(IMU_R))), gb = mulT(R, V3{0, 0, EST_G});
    const V3 ab = {in.acc[0] - gb.x - cen.x, in.acc[1] - gb.y - cen.y, in.acc[2] - gb.z - cen.z};
    s.tacc[0] = ab.x; s.tacc[1] = ab.y; s.tacc[2] = ab.z;
    const V3 aw = mul(R, ab); CNT(12 + 15 + 18 + 15 + 2 * 15);      // quaternion to matrix, gravity, centripetal term, world acceleration, world foot offsets
    const double lf[3] = {fw[0].x, fw[0].y, fw[0].z}, rf[3] = {fw[1].x, fw[1].y, fw[1].z}, awv[3] = {aw.x, aw.y, aw.z};
    // --- first call after setup (0x2d9c0-0x2e26f): zero pelvis state, foot states at MINUS the kinematic foot offset, P = 1e-6 I
    if (!s.inited) {
        for (int ax = 0; ax < 2; ++ax) {
            double* x = s.hx[ax];
            x[0] = x[1] = 0; x[2] = -lf[ax]; x[3] = -rf[ax]; x[4] = 0.5; x[5] = 0;
            for (int i = 0; i < 36; ++i) s.hP[ax][i] = (i % 7 == 0) ? 1e-6 : 0.0;
        }
        s.zx[0] = s.zx[1] = 0; s.zx[2] = -lf[2]; s.zx[3] = -rf[2]; s.zx[4] = EST_M * EST_G;
        for (int i = 0; i < 25; ++i) s.zP[i] = (i % 6 == 0) ? 1e-6 : 0.0;
        s.inited = 1;
    }
    s.sw = (50.0 > fz[0] ? 1 : 0) | (50.0 > fz[1] ? 2 : 0) | (fz[0] + fz[1] > 1.0 ? 4 : 0);
    for (int ax = 0; ax < 2; ++ax) hfilter_step(s.hx[ax], s.hP[ax], -lf[ax], -rf[ax], fz[0], fz[1], awv[ax]);
    zfilter_step(s.zx, s.zP, -lf[2], -rf[2], fz[0], fz[1]);
    // --- terrain height (0x2c70d-0x2cad7): only while the legs carry more than 1 N
    if (fz[0] + fz[1] > 1.0) {
        const double a = fz[0] / (fz[0] + fz[1]), u = a * (s.zx[0] + lf[2]) + (1 - a) * (s.zx[0] + rf[2]);
        s.terrain = 0.0004997501249375313 * u + 0.9995002498750625 * s.terrain;
    }
    s.pos[0] = s.hx[0][0]; s.pos[1] = s.hx[1][0]; s.pos[2] = s.zx[0];
    s.vel[0] = s.hx[0][1]; s.vel[1] = s.hx[1][1]; s.vel[2] = s.zx[1];
}

}  // namespace orc
