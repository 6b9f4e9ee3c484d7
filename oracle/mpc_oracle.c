/*
 * mpc_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See mpc_oracle.h for the scope statement and the "PARITY UNPINNED" note.
 *
 * Everything here is deliberately plain: dense 18x18 stage blocks, textbook
 * Riccati, scalar loops.  It shares no source with the HIP product path
 * (robotic_mpc_amd/csrc), which exploits the block structure instead.
 *
 * Citations "file:line" are into the reference checkout (/root/reference).
 */
#define _POSIX_C_SOURCE 200809L
#include "mpc_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NQ ORC_NQ
#define NX ORC_NX
#define NU ORC_NU
#define NW ORC_NW
#define NR ORC_NR
#define NB ORC_NB
#define BIG 1e29

/* ------------------------------------------------------------------ */
/* small vector helpers                                                */
/* ------------------------------------------------------------------ */
static void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void matmul3(const double *A, const double *B, double *C)
{
    double T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += A[3 * i + k] * B[3 * k + j];
            T[3 * i + j] = s;
        }
    memcpy(C, T, sizeof T);
}
static void matvec3(const double *A, const double *x, double *y)
{
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
    y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
/* Rodrigues rotation about a unit axis (revolute joint motion, as Pinocchio's
 * JointModelRevolute* used by loader.py:24-28). */
static void axis_rot(const double *a, double th, double *R)
{
    double c = cos(th), s = sin(th), v = 1.0 - c;
    R[0] = c + v * a[0] * a[0];        R[1] = v * a[0] * a[1] - s * a[2]; R[2] = v * a[0] * a[2] + s * a[1];
    R[3] = v * a[1] * a[0] + s * a[2]; R[4] = c + v * a[1] * a[1];        R[5] = v * a[1] * a[2] - s * a[0];
    R[6] = v * a[2] * a[0] - s * a[1]; R[7] = v * a[2] * a[1] + s * a[0]; R[8] = c + v * a[2] * a[2];
}

/* ------------------------------------------------------------------ */
/* kinematics                                                          */
/* ------------------------------------------------------------------ */
typedef struct {
    double o[6][3]; /* joint origins in WORLD */
    double z[6][3]; /* joint axes in WORLD */
    double p[3];    /* end-effector origin */
    double R[9];    /* end-effector rotation, row-major */
} kin_t;

/* forwardKinematics + updateFramePlacements -> oMf[ee] (prediction_model.py:126-132,
 * simulation_model.py:61-64); world_joint is identity (ur10.urdf:286-290). */
static void kin_eval(const orc_robot *rb, const double *q, kin_t *k)
{
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0}, t[3], Rj[9];
    for (int i = 0; i < 6; i++) {
        matvec3(R, rb->place[i] + 9, t);
        p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
        matmul3(R, rb->place[i], R);
        memcpy(k->o[i], p, sizeof p);
        matvec3(R, rb->axis[i], k->z[i]);
        axis_rot(rb->axis[i], q[i], Rj);
        matmul3(R, Rj, R);
    }
    matvec3(R, rb->place[6] + 9, t);
    k->p[0] = p[0] + t[0]; k->p[1] = p[1] + t[1]; k->p[2] = p[2] + t[2];
    matmul3(R, rb->place[6], k->R);
}

/* pose = [p; R flattened row-major] (prediction_model.py:134-142, simulation_model.py:65,69) */
void orc_fk(const orc_robot *rb, const double *q, double *pose12)
{
    kin_t k;
    kin_eval(rb, q, &k);
    memcpy(pose12, k.p, 3 * sizeof(double));
    memcpy(pose12 + 3, k.R, 9 * sizeof(double));
}

/* computeFrameJacobian(..., WORLD): spatial Jacobian at the world origin, column i =
 * [o_i x z_i ; z_i] (prediction_model.py:163-164, simulation_model.py:76). */
void orc_jacobian_world(const orc_robot *rb, const double *q, double *J36)
{
    kin_t k;
    kin_eval(rb, q, &k);
    for (int i = 0; i < 6; i++) {
        double c[3];
        cross3(k.o[i], k.z[i], c);
        for (int r = 0; r < 3; r++) {
            J36[r * 6 + i] = c[r];
            J36[(r + 3) * 6 + i] = k.z[i][r];
        }
    }
}

/* simulation_model.py:66-68 */
void orc_rpy(const double *pose12, double *rpy3)
{
    const double *R = pose12 + 3;
    rpy3[0] = atan2(R[7], R[8]);
    rpy3[1] = atan2(-R[6], sqrt(R[0] * R[0] + R[3] * R[3]));
    rpy3[2] = atan2(R[3], R[0]);
}

/* y = [p_task; R_task columns; v_task] (prediction_model.py:256-281, 299-314).
 * R_ee_t is the identity in the code (prediction_model.py:265-269). */
void orc_task_output(const orc_robot *rb, const double *q, const double *qd, double *y15)
{
    kin_t k;
    kin_eval(rb, q, &k);
    double tw[3], vl[3] = {0, 0, 0}, om[3] = {0, 0, 0}, c[3], s[3];
    matvec3(k.R, rb->t_ee, tw);
    for (int i = 0; i < 3; i++) y15[i] = k.p[i] + tw[i];
    for (int col = 0; col < 3; col++)
        for (int r = 0; r < 3; r++) y15[3 + 3 * col + r] = k.R[3 * r + col];
    for (int j = 0; j < 6; j++) {
        cross3(k.o[j], k.z[j], c);
        for (int r = 0; r < 3; r++) {
            vl[r] += c[r] * qd[j];
            om[r] += k.z[j][r] * qd[j];
        }
    }
    cross3(om, tw, c);
    for (int r = 0; r < 3; r++) s[r] = vl[r] + c[r];
    for (int col = 0; col < 3; col++)
        y15[12 + col] = k.R[col] * s[0] + k.R[3 + col] * s[1] + k.R[6 + col] * s[2]; /* R^T s */
}

/* Task functions g1..g5 (trajectory_optimizer.py:104-126) with the surface and its unit
 * normal (surface.py:21,243-262) and the analytic Jacobian wrt [q; qdot] that CasADi
 * obtains by AD in the reference's generated cost_y_fun_jac_ut_xt. */
void orc_task_g(const orc_robot *rb, const double *cf, const double *q, const double *qd, double *g5,
                double *G)
{
    kin_t k;
    kin_eval(rb, q, &k);
    double xh[3], yh[3], zh[3], tw[3], pt[3];
    for (int r = 0; r < 3; r++) { xh[r] = k.R[3 * r]; yh[r] = k.R[3 * r + 1]; zh[r] = k.R[3 * r + 2]; }
    (void)xh;
    matvec3(k.R, rb->t_ee, tw);
    for (int r = 0; r < 3; r++) pt[r] = k.p[r] + tw[r];
    const double a = cf[0], b = cf[1], c = cf[2], d = cf[3], e = cf[4], f = cf[5];
    const double X = pt[0], Y = pt[1];
    const double S = a * X * X + b * Y * Y + c * X * Y + d * X + e * Y + f;
    const double Sx = 2 * a * X + c * Y + d, Sy = 2 * b * Y + c * X + e;
    const double nn = sqrt(Sx * Sx + Sy * Sy + 1.0);
    const double n[3] = {Sx / nn, Sy / nn, -1.0 / nn};
    /* d n / dX, d n / dY : n = m/|m|, m = (Sx,Sy,-1) */
    const double mX[3] = {2 * a, c, 0}, mY[3] = {c, 2 * b, 0};
    double nX[3], nY[3];
    {
        double pX = dot3(n, mX), pY = dot3(n, mY);
        for (int r = 0; r < 3; r++) {
            nX[r] = (mX[r] - n[r] * pX) / nn;
            nY[r] = (mY[r] - n[r] * pY) / nn;
        }
    }
    /* spatial velocity at the world origin */
    double vl[3] = {0, 0, 0}, om[3] = {0, 0, 0}, cjl[6][3];
    for (int j = 0; j < 6; j++) {
        cross3(k.o[j], k.z[j], cjl[j]);
        for (int r = 0; r < 3; r++) {
            vl[r] += cjl[j][r] * qd[j];
            om[r] += k.z[j][r] * qd[j];
        }
    }
    double owt[3], s[3];
    cross3(om, tw, owt);
    for (int r = 0; r < 3; r++) s[r] = vl[r] + owt[r];

    g5[0] = S - pt[2];
    g5[1] = dot3(n, zh);
    g5[2] = yh[0];
    g5[3] = pt[0];
    g5[4] = dot3(yh, s);
    if (!G) return;
    memset(G, 0, 60 * sizeof(double));
    for (int i = 0; i < 6; i++) {
        const double *zi = k.z[i];
        double rel[3], dpt[3], dzh[3], dyh[3], dtw[3];
        for (int r = 0; r < 3; r++) rel[r] = pt[r] - k.o[i][r];
        cross3(zi, rel, dpt);
        cross3(zi, zh, dzh);
        cross3(zi, yh, dyh);
        cross3(zi, tw, dtw);
        G[0 * 12 + i] = Sx * dpt[0] + Sy * dpt[1] - dpt[2];
        double dn[3];
        for (int r = 0; r < 3; r++) dn[r] = nX[r] * dpt[0] + nY[r] * dpt[1];
        G[1 * 12 + i] = dot3(dn, zh) + dot3(n, dzh);
        G[2 * 12 + i] = dyh[0];
        G[3 * 12 + i] = dpt[0];
        /* d vl / dq_i, d om / dq_i : only joints after i move */
        double dvl[3] = {0, 0, 0}, otail[3] = {0, 0, 0};
        for (int j = i + 1; j < 6; j++) {
            double roj[3], doj[3], dzj[3], t1[3], t2[3];
            for (int r = 0; r < 3; r++) roj[r] = k.o[j][r] - k.o[i][r];
            cross3(zi, roj, doj);
            cross3(zi, k.z[j], dzj);
            cross3(doj, k.z[j], t1);
            cross3(k.o[j], dzj, t2);
            for (int r = 0; r < 3; r++) {
                dvl[r] += (t1[r] + t2[r]) * qd[j];
                otail[r] += k.z[j][r] * qd[j];
            }
        }
        double dom[3], t3[3], t4[3];
        cross3(zi, otail, dom);
        cross3(dom, tw, t3);
        cross3(om, dtw, t4);
        double ds[3];
        for (int r = 0; r < 3; r++) ds[r] = dvl[r] + t3[r] + t4[r];
        G[4 * 12 + i] = dot3(dyh, s) + dot3(yh, ds);
        /* d g5 / d qdot_i */
        double zt[3];
        cross3(zi, tw, zt);
        G[4 * 12 + 6 + i] = yh[0] * (cjl[i][0] + zt[0]) + yh[1] * (cjl[i][1] + zt[1]) + yh[2] * (cjl[i][2] + zt[2]);
    }
}

/* ------------------------------------------------------------------ */
/* models                                                              */
/* ------------------------------------------------------------------ */
/* prediction_model.py:87-115: exact ZOH of the decoupled velocity loop */
void orc_lti(const double *wcv, double Ts, double *a12, double *a22, double *b1, double *b2)
{
    for (int j = 0; j < 6; j++) {
        a22[j] = exp(-wcv[j] * Ts);
        a12[j] = (1.0 - a22[j]) / wcv[j];
        b2[j] = 1.0 - a22[j];
        b1[j] = Ts - a12[j];
    }
}

/* simulation_model.py:79-83 */
static void plant_f(const double *wcv, const double *z, const double *u, double *zd)
{
    for (int j = 0; j < 6; j++) {
        zd[j] = z[6 + j];
        zd[6 + j] = -wcv[j] * z[6 + j] + wcv[j] * u[j];
    }
}
/* simulation_model.py:111-117 */
void orc_rk4(const double *wcv, double dt, const double *z, const double *u, double *zn)
{
    double k1[12], k2[12], k3[12], k4[12], t[12];
    plant_f(wcv, z, u, k1);
    for (int i = 0; i < 12; i++) t[i] = z[i] + 0.5 * dt * k1[i];
    plant_f(wcv, t, u, k2);
    for (int i = 0; i < 12; i++) t[i] = z[i] + 0.5 * dt * k2[i];
    plant_f(wcv, t, u, k3);
    for (int i = 0; i < 12; i++) t[i] = z[i] + dt * k3[i];
    plant_f(wcv, t, u, k4);
    for (int i = 0; i < 12; i++)
        zn[i] = z[i] + (dt / 6) * k1[i] + (dt / 3) * k2[i] + (dt / 3) * k3[i] + (dt / 6) * k4[i];
}

/* simulation_model.py:93-109: Euler, RK2 (midpoint), RK3; anything else: RK4 */
void orc_plant_step(int integrator, const double *wcv, double dt, const double *z, const double *u, double *zn)
{
    double k1[12], k2[12], k3[12], t[12];
    if (integrator < 1 || integrator > 3) { orc_rk4(wcv, dt, z, u, zn); return; }
    plant_f(wcv, z, u, k1);
    if (integrator == 1) {
        for (int i = 0; i < 12; i++) zn[i] = z[i] + dt * k1[i];
        return;
    }
    for (int i = 0; i < 12; i++) t[i] = z[i] + 0.5 * dt * k1[i];
    plant_f(wcv, t, u, k2);
    if (integrator == 2) {
        for (int i = 0; i < 12; i++) zn[i] = z[i] + dt * k2[i];
        return;
    }
    for (int i = 0; i < 12; i++) t[i] = z[i] - dt * k1[i] + 2.0 * dt * k2[i];
    plant_f(wcv, t, u, k3);
    for (int i = 0; i < 12; i++) zn[i] = z[i] + (dt / 6) * (k1[i] + 4.0 * k2[i] + k3[i]);
}

/* r = [g - g_ref (5); u (6); qddot_k (6)], Jr = d r / d [u; q; qdot]
 * (trajectory_optimizer.py:107-126,154-155; prediction_model.py:322-326). */
void orc_stage_residual(const orc_robot *rb, const orc_params *p, const double *x, const double *u,
                        double *r, double *Jr)
{
    double a12[6], a22[6], b1[6], b2[6], g[5], G[60];
    orc_lti(p->wcv, p->dt, a12, a22, b1, b2);
    orc_task_g(rb, p->coeffs, x, x + 6, g, Jr ? G : NULL);
    const double gref[5] = {0.0, 1.0, 0.0, p->px_ref, p->vy_ref};
    for (int i = 0; i < 5; i++) r[i] = g[i] - gref[i];
    for (int j = 0; j < 6; j++) {
        r[5 + j] = u[j];
        /* (qdot_next - qdot)/Ts with qdot_next = a22 qdot + b2 u */
        r[11 + j] = ((a22[j] * x[6 + j] + b2[j] * u[j]) - x[6 + j]) / p->dt;
    }
    if (!Jr) return;
    memset(Jr, 0, NR * NW * sizeof(double));
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 12; j++) Jr[i * NW + 6 + j] = G[i * 12 + j];
    for (int j = 0; j < 6; j++) {
        Jr[(5 + j) * NW + j] = 1.0;
        Jr[(11 + j) * NW + j] = b2[j] / p->dt;
        Jr[(11 + j) * NW + 12 + j] = (a22[j] - 1.0) / p->dt;
    }
}

/* ------------------------------------------------------------------ */
/* dense helpers for the QP                                            */
/* ------------------------------------------------------------------ */
static int chol_lower(double *A, int n) /* in place, row-major, lower */
{
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}

typedef struct {
    int N;
    /* Newton-system data */
    double *Ht;  /* [N+1][324] H + Gamma */
    double *gt;  /* [N+1][18] */
    double *rg, *rb, *rd, *rm; /* residuals */
    double *dw, *dpi, *dlam, *dt;
    double *dlam_a, *dt_a;
    double *L;   /* [N][36]  chol(G_uu) */
    double *Kt;  /* [N][72]  L^-1 G_ux */
    double *l;   /* [N][6] */
    double *P;   /* [N+1][144] */
    double *pv;  /* [N+1][12] */
    unsigned char *mask; /* [N+1][24] */
} ipm_ws;

static size_t ipm_ws_doubles(int N)
{
    size_t n1 = (size_t)N + 1;
    return n1 * 324 + n1 * 18 + n1 * 18 + n1 * 12 + n1 * 24 * 2 + n1 * 18 + n1 * 12 + n1 * 24 * 4 +
           n1 * 36 + n1 * 72 + n1 * 6 + n1 * 144 + n1 * 12;
}
static void ipm_ws_carve(ipm_ws *ws, int N, double *mem, unsigned char *mask)
{
    size_t n1 = (size_t)N + 1;
    ws->N = N;
    ws->Ht = mem; mem += n1 * 324;
    ws->gt = mem; mem += n1 * 18;
    ws->rg = mem; mem += n1 * 18;
    ws->rb = mem; mem += n1 * 12;
    ws->rd = mem; mem += n1 * 24;
    ws->rm = mem; mem += n1 * 24;
    ws->dw = mem; mem += n1 * 18;
    ws->dpi = mem; mem += n1 * 12;
    ws->dlam = mem; mem += n1 * 24;
    ws->dt = mem; mem += n1 * 24;
    ws->dlam_a = mem; mem += n1 * 24;
    ws->dt_a = mem; mem += n1 * 24;
    ws->L = mem; mem += n1 * 36;
    ws->Kt = mem; mem += n1 * 72;
    ws->l = mem; mem += n1 * 6;
    ws->P = mem; mem += n1 * 144;
    ws->pv = mem; mem += n1 * 12;
    ws->mask = mask;
}

/* OCP-QP residuals (HPIPM d_ocp_qp_res_compute restated for this block structure).
 * Lagrangian: sum 1/2 w'Hw + g'w + pi_k'(A x_k + B u_k + b_k - x_{k+1})
 *             - lam_lb'(v - lb - t_lb) - lam_ub'(ub - v - t_ub).               */
static void ipm_residuals(int N, const double *H, const double *g, const double *b, const double *A,
                          const double *B, const double *lb, const double *ub, const double *w,
                          const double *pi, const double *lam, const double *t, ipm_ws *ws, double *nrm4,
                          double *mu_out)
{
    double ng = 0, nb = 0, nd = 0, nm = 0, mu = 0;
    int nc = 0;
    for (int k = 0; k <= N; k++) {
        const double *Hk = H + (size_t)k * 324, *wk = w + (size_t)k * NW;
        double *rg = ws->rg + (size_t)k * NW;
        for (int i = 0; i < NW; i++) {
            double s = g[(size_t)k * NW + i];
            for (int j = 0; j < NW; j++) s += Hk[i * NW + j] * wk[j];
            rg[i] = s;
        }
        if (k < N) {
            const double *pk = pi + (size_t)k * NX;
            for (int j = 0; j < NU; j++) {
                double s = 0;
                for (int i = 0; i < NX; i++) s += B[i * NU + j] * pk[i];
                rg[j] += s;
            }
            for (int j = 0; j < NX; j++) {
                double s = 0;
                for (int i = 0; i < NX; i++) s += A[i * NX + j] * pk[i];
                rg[NU + j] += s;
            }
        }
        if (k >= 1) {
            const double *pm = pi + (size_t)(k - 1) * NX;
            for (int j = 0; j < NX; j++) rg[NU + j] -= pm[j];
        }
        for (int j = 0; j < NB; j++) {
            const unsigned char *mk = ws->mask + (size_t)k * 24;
            double *rd = ws->rd + (size_t)k * 24, *rm = ws->rm + (size_t)k * 24;
            const double *lk = lam + (size_t)k * 24, *tk = t + (size_t)k * 24;
            if (mk[j]) {
                rg[j] -= lk[j];
                rd[j] = wk[j] - lb[(size_t)k * NB + j] - tk[j];
                rm[j] = lk[j] * tk[j];
                mu += rm[j]; nc++;
                if (fabs(rd[j]) > nd) nd = fabs(rd[j]);
                if (fabs(rm[j]) > nm) nm = fabs(rm[j]);
            } else { rd[j] = 0; rm[j] = 0; }
            if (mk[12 + j]) {
                rg[j] += lk[12 + j];
                rd[12 + j] = ub[(size_t)k * NB + j] - wk[j] - tk[12 + j];
                rm[12 + j] = lk[12 + j] * tk[12 + j];
                mu += rm[12 + j]; nc++;
                if (fabs(rd[12 + j]) > nd) nd = fabs(rd[12 + j]);
                if (fabs(rm[12 + j]) > nm) nm = fabs(rm[12 + j]);
            } else { rd[12 + j] = 0; rm[12 + j] = 0; }
        }
        /* x_0 is eliminated (lbx_0 = ubx_0, simulator.py:210-211); stage N has no input */
        if (k == 0) for (int j = 0; j < NX; j++) rg[NU + j] = 0;
        if (k == N) for (int j = 0; j < NU; j++) rg[j] = 0;
        for (int i = 0; i < NW; i++) if (fabs(rg[i]) > ng) ng = fabs(rg[i]);
        if (k < N) {
            double *rbk = ws->rb + (size_t)k * NX;
            const double *xk = wk + NU, *uk = wk, *xn = w + (size_t)(k + 1) * NW + NU;
            for (int i = 0; i < NX; i++) {
                double s = b[(size_t)k * NX + i] - xn[i];
                for (int j = 0; j < NX; j++) s += A[i * NX + j] * xk[j];
                for (int j = 0; j < NU; j++) s += B[i * NU + j] * uk[j];
                rbk[i] = s;
                if (fabs(s) > nb) nb = fabs(s);
            }
        }
    }
    nrm4[0] = ng; nrm4[1] = nb; nrm4[2] = nd; nrm4[3] = nm;
    *mu_out = nc ? mu / nc : 0.0;
}

/* Backward Riccati factorisation + solve of the Newton system
 *   (H+Gamma) dw + E' dpi = -gt ,  E dw = -rb   (dx_0 = 0)
 * classical recursion on dense 18x18 blocks; factor != 0 refactorises. */
static int ipm_riccati_fx(int N, const double *A, const double *B, ipm_ws *ws, int factor, const unsigned char *fixed /* [N][6] or NULL */)
{
    double BA[NX * NW];
    for (int i = 0; i < NX; i++) {
        for (int j = 0; j < NU; j++) BA[i * NW + j] = B[i * NU + j];
        for (int j = 0; j < NX; j++) BA[i * NW + NU + j] = A[i * NX + j];
    }
    /* terminal */
    {
        double *PN = ws->P + (size_t)N * 144, *pN = ws->pv + (size_t)N * NX;
        const double *HN = ws->Ht + (size_t)N * 324, *gN = ws->gt + (size_t)N * NW;
        if (factor)
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NX; j++) PN[i * NX + j] = HN[(NU + i) * NW + NU + j];
        for (int i = 0; i < NX; i++) pN[i] = gN[NU + i];
    }
    for (int k = N - 1; k >= 0; k--) {
        const double *Pn = ws->P + (size_t)(k + 1) * 144, *pn = ws->pv + (size_t)(k + 1) * NX;
        const double *Hk = ws->Ht + (size_t)k * 324, *gk = ws->gt + (size_t)k * NW;
        const double *rbk = ws->rb + (size_t)k * NX;
        double *L = ws->L + (size_t)k * 36, *Kt = ws->Kt + (size_t)k * 72, *l = ws->l + (size_t)k * 6;
        double *Pk = ws->P + (size_t)k * 144, *pk = ws->pv + (size_t)k * NX;
        double h[NW], m[NX];
        /* m = p_{k+1} + P_{k+1} rb_k */
        for (int i = 0; i < NX; i++) {
            double s = pn[i];
            for (int j = 0; j < NX; j++) s += Pn[i * NX + j] * rbk[j];
            m[i] = s;
        }
        for (int i = 0; i < NW; i++) {
            double s = gk[i];
            for (int j = 0; j < NX; j++) s += BA[j * NW + i] * m[j];
            h[i] = s;
        }
        /* active-set fast path: a FIXED input (du_j = 0 in this solve) leaves the stage problem -- identity row / column in
         * R~ = G_uu, zero row in S~ = G_ux, zero entry in h_u -- AFTER the B'PB terms have been added */
        if (fixed) for (int j = 0; j < NU; j++) if (fixed[(size_t)k * NU + j]) h[j] = 0.0;
        if (factor) {
            double PBA[NX * NW], G[NW * NW];
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NW; j++) {
                    double s = 0;
                    for (int r = 0; r < NX; r++) s += Pn[i * NX + r] * BA[r * NW + j];
                    PBA[i * NW + j] = s;
                }
            for (int i = 0; i < NW; i++)
                for (int j = 0; j < NW; j++) {
                    double s = Hk[i * NW + j];
                    for (int r = 0; r < NX; r++) s += BA[r * NW + i] * PBA[r * NW + j];
                    G[i * NW + j] = s;
                }
            if (fixed)
                for (int j = 0; j < NU; j++)
                    if (fixed[(size_t)k * NU + j]) {
                        for (int c = 0; c < NW; c++) { G[j * NW + c] = 0.0; G[c * NW + j] = 0.0; }
                        G[j * NW + j] = 1.0;
                    }
            for (int i = 0; i < NU; i++)
                for (int j = 0; j < NU; j++) L[i * NU + j] = G[i * NW + j];
            if (chol_lower(L, NU)) return -1;
            /* Kt = L^-1 G_ux */
            for (int c = 0; c < NX; c++)
                for (int i = 0; i < NU; i++) {
                    double s = G[i * NW + NU + c];
                    for (int r = 0; r < i; r++) s -= L[i * NU + r] * Kt[r * NX + c];
                    Kt[i * NX + c] = s / L[i * NU + i];
                }
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < NX; j++) {
                    double s = G[(NU + i) * NW + NU + j];
                    for (int r = 0; r < NU; r++) s -= Kt[r * NX + i] * Kt[r * NX + j];
                    Pk[i * NX + j] = s;
                }
            /* symmetrise against rounding drift */
            for (int i = 0; i < NX; i++)
                for (int j = 0; j < i; j++) {
                    double s = 0.5 * (Pk[i * NX + j] + Pk[j * NX + i]);
                    Pk[i * NX + j] = s; Pk[j * NX + i] = s;
                }
        }
        for (int i = 0; i < NU; i++) {
            double s = h[i];
            for (int r = 0; r < i; r++) s -= L[i * NU + r] * l[r];
            l[i] = s / L[i * NU + i];
        }
        for (int i = 0; i < NX; i++) {
            double s = h[NU + i];
            for (int r = 0; r < NU; r++) s -= Kt[r * NX + i] * l[r];
            pk[i] = s;
        }
    }
    /* forward */
    double dx[NX];
    memset(dx, 0, sizeof dx);
    for (int k = 0; k < N; k++) {
        const double *L = ws->L + (size_t)k * 36, *Kt = ws->Kt + (size_t)k * 72, *l = ws->l + (size_t)k * 6;
        double *dwk = ws->dw + (size_t)k * NW, y[NU], du[NU], dxn[NX];
        for (int i = 0; i < NU; i++) {
            double s = l[i];
            for (int j = 0; j < NX; j++) s += Kt[i * NX + j] * dx[j];
            y[i] = -s;
        }
        for (int i = NU - 1; i >= 0; i--) { /* L' du = y */
            double s = y[i];
            for (int r = i + 1; r < NU; r++) s -= L[r * NU + i] * du[r];
            du[i] = s / L[i * NU + i];
        }
        for (int i = 0; i < NU; i++) dwk[i] = du[i];
        for (int i = 0; i < NX; i++) dwk[NU + i] = dx[i];
        const double *rbk = ws->rb + (size_t)k * NX;
        for (int i = 0; i < NX; i++) {
            double s = rbk[i];
            for (int j = 0; j < NX; j++) s += A[i * NX + j] * dx[j];
            for (int j = 0; j < NU; j++) s += B[i * NU + j] * du[j];
            dxn[i] = s;
        }
        const double *Pn = ws->P + (size_t)(k + 1) * 144, *pn = ws->pv + (size_t)(k + 1) * NX;
        double *dpk = ws->dpi + (size_t)k * NX;
        for (int i = 0; i < NX; i++) {
            double s = pn[i];
            for (int j = 0; j < NX; j++) s += Pn[i * NX + j] * dxn[j];
            dpk[i] = s;
        }
        memcpy(dx, dxn, sizeof dx);
    }
    {
        double *dwN = ws->dw + (size_t)N * NW;
        for (int i = 0; i < NU; i++) dwN[i] = 0;
        for (int i = 0; i < NX; i++) dwN[NU + i] = dx[i];
    }
    return 0;
}

static int ipm_riccati(int N, const double *A, const double *B, ipm_ws *ws, int factor)
{
    return ipm_riccati_fx(N, A, B, ws, factor, NULL);
}

/* Build (H+Gamma, gt) from the current residuals (HPIPM compute_Gamma_gamma restated),
 * solve, then recover dt, dlam (compute_lam_t). */
static int ipm_newton(int N, const double *H, const double *A, const double *B, const double *lam,
                      const double *t, ipm_ws *ws, int factor)
{
    for (int k = 0; k <= N; k++) {
        double *Ht = ws->Ht + (size_t)k * 324, *gt = ws->gt + (size_t)k * NW;
        const double *rg = ws->rg + (size_t)k * NW, *rd = ws->rd + (size_t)k * 24, *rm = ws->rm + (size_t)k * 24;
        const double *lk = lam + (size_t)k * 24, *tk = t + (size_t)k * 24;
        const unsigned char *mk = ws->mask + (size_t)k * 24;
        if (factor) memcpy(Ht, H + (size_t)k * 324, 324 * sizeof(double));
        memcpy(gt, rg, NW * sizeof(double));
        for (int j = 0; j < NB; j++) {
            if (mk[j]) {
                if (factor) Ht[j * NW + j] += lk[j] / tk[j];
                gt[j] += (rm[j] + lk[j] * rd[j]) / tk[j];
            }
            if (mk[12 + j]) {
                if (factor) Ht[j * NW + j] += lk[12 + j] / tk[12 + j];
                gt[j] -= (rm[12 + j] + lk[12 + j] * rd[12 + j]) / tk[12 + j];
            }
        }
    }
    if (ipm_riccati(N, A, B, ws, factor)) return -1;
    for (int k = 0; k <= N; k++) {
        const double *dw = ws->dw + (size_t)k * NW, *rd = ws->rd + (size_t)k * 24, *rm = ws->rm + (size_t)k * 24;
        const double *lk = lam + (size_t)k * 24, *tk = t + (size_t)k * 24;
        const unsigned char *mk = ws->mask + (size_t)k * 24;
        double *dl = ws->dlam + (size_t)k * 24, *dtt = ws->dt + (size_t)k * 24;
        for (int j = 0; j < NB; j++) {
            if (mk[j]) {
                dtt[j] = dw[j] + rd[j];
                dl[j] = -(rm[j] + lk[j] * dtt[j]) / tk[j];
            } else { dtt[j] = 0; dl[j] = 0; }
            if (mk[12 + j]) {
                dtt[12 + j] = -dw[j] + rd[12 + j];
                dl[12 + j] = -(rm[12 + j] + lk[12 + j] * dtt[12 + j]) / tk[12 + j];
            } else { dtt[12 + j] = 0; dl[12 + j] = 0; }
        }
    }
    return 0;
}

static double ipm_alpha(int N, const double *lam, const double *t, const ipm_ws *ws)
{
    double alpha = 1.0;
    size_t n = ((size_t)N + 1) * 24;
    for (size_t i = 0; i < n; i++) {
        if (!ws->mask[i]) continue;
        if (ws->dlam[i] < 0 && lam[i] + alpha * ws->dlam[i] < 0) alpha = -lam[i] / ws->dlam[i];
        if (ws->dt[i] < 0 && t[i] + alpha * ws->dt[i] < 0) alpha = -t[i] / ws->dt[i];
    }
    return alpha;
}

/* which bound sides exist: inputs on stages 0..N-1, joint positions wherever lb/ub is finite (the caller
 * passes +-1e30 on the stages without position bounds, trajectory_optimizer.py:164-171) */
static int ipm_build_mask(int N, const double *lb, const double *ub, unsigned char *mask)
{
    int nc = 0;
    for (int k = 0; k <= N; k++)
        for (int j = 0; j < NB; j++) {
            int has_u = (j < NU) ? (k < N) : 1;
            mask[(size_t)k * 24 + j] = has_u && lb[(size_t)k * NB + j] > -BIG;
            mask[(size_t)k * 24 + 12 + j] = has_u && ub[(size_t)k * NB + j] < BIG;
            nc += mask[(size_t)k * 24 + j] + mask[(size_t)k * 24 + 12 + j];
        }
    return nc;
}

/* BOUND-INACTIVE FAST PATH (not HPIPM: an addition of this build, orc_params.fast_path).  The QP is strictly convex
 * (SURVEY A.6), so when its EQUALITY-constrained minimiser -- one Riccati factorisation with Gamma = 0, started from
 * w = 0 with x_0 embedded -- keeps every bounded component at least ORC_FAST_MARGIN inside its bounds, that point is
 * the QP's solution with all bound multipliers 0, and the interior-point loop (which would need two factorisations
 * to bring lam*t from the 0.1 warm-start clamp down to qp_tol) is skipped.  Returns 1 and writes (w, pi, lam = 0,
 * t = slack) on acceptance; returns 0 and leaves the warm start (w, pi, lam, t) UNTOUCHED otherwise.
 * Replaces nothing in the reference's results beyond qp_tol: simulator.py:212 sees the same u to ~1e-9. */
#define ORC_FAST_MARGIN 1e-3
#define ORC_AS_MAXTRY 3
/* ACTIVE-SET form of the fast path (fast_path >= 3).  Some inputs ride their bounds (the start-up transient, tight limits): guess the
 * active set -- the input components whose multiplier exceeded their slack in the previous QP's solution, one stage later (the
 * horizon recedes by one stage per MPC step) -- FIX them at their bounds (du = bound - u), solve the reduced equality-constrained QP
 * by the same Riccati sweep (identity rows in R~, zero rows in S~), and accept when every FREE bounded component clears its bounds by
 * the margin and every FIXED one has a multiplier >= 0: those are the KKT conditions of the inequality-constrained QP, so the point
 * is its (unique) solution.  Otherwise fix what is violated, release what has a negative multiplier, and solve again, at most
 * ORC_AS_MAXTRY times; position (q) bounds are never fixed -- one of them violated rejects.  With an empty guess this is the plain
 * fast path.  `shift`: this is the first QP of an MPC step (later QPs of a full-SQP step keep the stage index: same horizon).  `nsolve` returns the number of Riccati factorisations spent. */
static int ipm_fast_path_as(int N, const double *H, const double *g, const double *b, const double *A, const double *B,
                            const double *lb, const double *ub, const double *dx0, double *w, double *pi, double *lam,
                            double *t, ipm_ws *ws, int use_as, int shift, int *nsolve)
{
    unsigned char *code = (unsigned char *)calloc((size_t)(N + 1) * NU, 1), *fixed = (unsigned char *)calloc((size_t)(N + 1) * NU, 1);
    double *w0 = (double *)calloc((size_t)(N + 1) * NW, sizeof(double)), *mult = (double *)calloc((size_t)(N + 1) * NU, sizeof(double));
    int accepted = 0;
    *nsolve = 0;
    if (!code || !fixed || !w0 || !mult) goto done;
    ipm_build_mask(N, lb, ub, ws->mask);
    if (use_as)
        for (int k = 0; k < N; k++) {
            const int ks = (shift && k + 1 < N) ? k + 1 : k;     /* the previous solution, one stage later at the first QP of an MPC step */
            for (int j = 0; j < NU; j++) {
                const unsigned char *mk = ws->mask + (size_t)k * 24;
                const double *lk = lam + (size_t)ks * 24, *tk = t + (size_t)ks * 24;
                code[(size_t)k * NU + j] = (mk[j] && lk[j] > tk[j]) ? 1 : ((mk[12 + j] && lk[12 + j] > tk[12 + j]) ? 2 : 0);
            }
        }
    for (int tr = 0; tr < (use_as ? ORC_AS_MAXTRY : 1); tr++) {
        /* point the solve starts from: x_0 embedded, fixed inputs on their bounds, everything else 0 */
        memset(w0, 0, (size_t)(N + 1) * NW * sizeof(double));
        for (int i = 0; i < NX; i++) w0[NU + i] = dx0[i];
        for (int k = 0; k < N; k++)
            for (int j = 0; j < NU; j++) {
                const unsigned char c = code[(size_t)k * NU + j];
                fixed[(size_t)k * NU + j] = c != 0;
                if (c) w0[(size_t)k * NW + j] = c == 1 ? lb[(size_t)k * NB + j] : ub[(size_t)k * NB + j];
            }
        for (int k = 0; k <= N; k++) {
            double *Ht = ws->Ht + (size_t)k * 324, *gt = ws->gt + (size_t)k * NW;
            memcpy(Ht, H + (size_t)k * 324, 324 * sizeof(double));
            for (int i = 0; i < NW; i++) {
                double s = g[(size_t)k * NW + i];
                for (int c = 0; c < NW; c++) s += Ht[i * NW + c] * w0[(size_t)k * NW + c];
                gt[i] = s;
            }
            if (k == 0) for (int j = 0; j < NX; j++) gt[NU + j] = 0;   /* x_0 is eliminated */
            if (k == N) for (int j = 0; j < NU; j++) gt[j] = 0;        /* no input at stage N */
            if (k < N) {
                double *rbk = ws->rb + (size_t)k * NX;
                for (int i = 0; i < NX; i++) {
                    double s = b[(size_t)k * NX + i] - w0[(size_t)(k + 1) * NW + NU + i];
                    for (int j = 0; j < NX; j++) s += A[i * NX + j] * w0[(size_t)k * NW + NU + j];
                    for (int j = 0; j < NU; j++) s += B[i * NU + j] * w0[(size_t)k * NW + j];
                    rbk[i] = s;
                }
            }
        }
        (*nsolve)++;
        if (ipm_riccati_fx(N, A, B, ws, 1, fixed)) break;
        /* candidate = w0 + dw ; verdict per component */
        int ok = 1, changes = 0, hopeless = 0;
        for (int k = 0; k <= N && !hopeless; k++) {
            const unsigned char *mk = ws->mask + (size_t)k * 24;
            for (int i = 0; i < NW; i++) {
                const double v = w0[(size_t)k * NW + i] + ws->dw[(size_t)k * NW + i];
                if (v != v) { hopeless = 1; break; }
                if (i >= NB) continue;
                if (i < NU && k < N && code[(size_t)k * NU + i]) {
                    /* fixed input: its multiplier is the stationarity residual of the candidate in that row */
                    double r = g[(size_t)k * NW + i];
                    for (int c = 0; c < NW; c++) r += H[(size_t)k * 324 + i * NW + c] * (w0[(size_t)k * NW + c] + ws->dw[(size_t)k * NW + c]);
                    for (int c = 0; c < NX; c++) r += B[c * NU + i] * ws->dpi[(size_t)k * NX + c];
                    const double m_ = code[(size_t)k * NU + i] == 1 ? r : -r;
                    mult[(size_t)k * NU + i] = m_;
                    if (!(m_ >= 0.0)) { ok = 0; code[(size_t)k * NU + i] = 0; changes++; }
                    continue;
                }
                const int lo_ok = !mk[i] || v - lb[(size_t)k * NB + i] >= ORC_FAST_MARGIN;
                const int hi_ok = !mk[12 + i] || ub[(size_t)k * NB + i] - v >= ORC_FAST_MARGIN;
                if (lo_ok && hi_ok) continue;
                ok = 0;
                const int below = mk[i] && v - lb[(size_t)k * NB + i] < 0.0, above = mk[12 + i] && ub[(size_t)k * NB + i] - v < 0.0;
                if (below || above) {
                    if (i >= NU || !use_as) { hopeless = 1; break; }       /* a position bound (or no active-set mode): not ours */
                    code[(size_t)k * NU + i] = below ? 1 : 2; changes++;
                }   /* inside the margin but feasible: neither accepted nor a reason to fix it */
            }
        }
        if (hopeless) break;
        if (ok) {
            for (int k = 0; k <= N; k++) {
                const unsigned char *mk = ws->mask + (size_t)k * 24;
                for (int i = 0; i < NW; i++) w[(size_t)k * NW + i] = w0[(size_t)k * NW + i] + ws->dw[(size_t)k * NW + i];
                if (k < N) memcpy(pi + (size_t)k * NX, ws->dpi + (size_t)k * NX, NX * sizeof(double));
                for (int j = 0; j < NB; j++) {
                    const double v = w[(size_t)k * NW + j];
                    const unsigned char c = (j < NU && k < N) ? code[(size_t)k * NU + j] : 0;
                    lam[(size_t)k * 24 + j] = c == 1 ? mult[(size_t)k * NU + j] : 0.0;
                    lam[(size_t)k * 24 + 12 + j] = c == 2 ? mult[(size_t)k * NU + j] : 0.0;
                    t[(size_t)k * 24 + j] = mk[j] ? (c == 1 ? 0.0 : v - lb[(size_t)k * NB + j]) : 1.0;
                    t[(size_t)k * 24 + 12 + j] = mk[12 + j] ? (c == 2 ? 0.0 : ub[(size_t)k * NB + j] - v) : 1.0;
                }
            }
            accepted = 1;
            break;
        }
        if (!changes) break;
    }
done:
    free(code); free(fixed); free(w0); free(mult);
    return accepted;
}

static int ipm_fast_path(int N, const double *H, const double *g, const double *b, const double *A, const double *B,
                         const double *lb, const double *ub, const double *dx0, double *w, double *pi, double *lam,
                         double *t, ipm_ws *ws)
{
    int n;
    return ipm_fast_path_as(N, H, g, b, A, B, lb, ub, dx0, w, pi, lam, t, ws, 0, 0, &n);
}

/* Mehrotra predictor-corrector IPM; restates HPIPM's d_ocp_qp_ipm_solve main loop
 * (init with warm_start=2: keep everything, clamp lam,t >= 0.1; predictor; sigma =
 * (mu_aff/mu)^3; centering-corrector on the same factorisation; step
 * alpha*((1-alpha)*0.99+alpha*0.9999999); exit on the four inf-norms <= tol). */
static int ipm_solve(int N, const double *H, const double *g, const double *b, const double *A, const double *B,
                     const double *lb, const double *ub, const double *dx0, double *w, double *pi, double *lam,
                     double *t, double tol, int iter_max, int *iters, double *res4, ipm_ws *ws)
{
    const double thr0 = 0.1, alpha_min = 1e-12, lam_min = 1e-16, t_min = 1e-16;
    size_t nlt = ((size_t)N + 1) * 24;
    const int nc = ipm_build_mask(N, lb, ub, ws->mask);
    for (size_t i = 0; i < nlt; i++) {
        if (ws->mask[i]) {
            if (lam[i] < thr0) lam[i] = thr0;
            if (t[i] < thr0) t[i] = thr0;
        } else { lam[i] = 0; t[i] = 1; }
    }
    for (int i = 0; i < NX; i++) w[NU + i] = dx0[i];
    for (int i = 0; i < NU; i++) w[(size_t)N * NW + i] = 0;

    double nrm[4], mu;
    ipm_residuals(N, H, g, b, A, B, lb, ub, w, pi, lam, t, ws, nrm, &mu);
    int it = 0, status = 1;
    double alpha = 1.0;
    for (;; it++) {
        if (nrm[0] != nrm[0] || nrm[1] != nrm[1] || nrm[2] != nrm[2] || nrm[3] != nrm[3]) { status = 3; break; }
        if (!(nrm[0] > tol || nrm[1] > tol || nrm[2] > tol || nrm[3] > tol)) { status = 0; break; }
        if (it >= iter_max) { status = 1; break; }
        if (!(alpha > alpha_min)) { status = 2; break; }
        /* predictor (affine) */
        if (ipm_newton(N, H, A, B, lam, t, ws, 1)) { status = 3; break; }
        double a_aff = ipm_alpha(N, lam, t, ws);
        if (nc > 0) {
            double mu_aff = 0;
            for (size_t i = 0; i < nlt; i++)
                if (ws->mask[i]) mu_aff += (lam[i] + a_aff * ws->dlam[i]) * (t[i] + a_aff * ws->dt[i]);
            mu_aff /= nc;
            double tmp = mu_aff / mu, sigma = tmp * tmp * tmp;
            memcpy(ws->dlam_a, ws->dlam, nlt * sizeof(double));
            memcpy(ws->dt_a, ws->dt, nlt * sizeof(double));
            /* centering-corrector: rm <- lam t + dlam_a dt_a - sigma mu */
            for (size_t i = 0; i < nlt; i++)
                if (ws->mask[i]) ws->rm[i] = lam[i] * t[i] + ws->dlam_a[i] * ws->dt_a[i] - sigma * mu;
            if (ipm_newton(N, H, A, B, lam, t, ws, 0)) { status = 3; break; }
            alpha = ipm_alpha(N, lam, t, ws);
        } else alpha = a_aff;
        if (getenv("ORC_DEBUG"))
            fprintf(stderr, "ipm it %d: res %.3e %.3e %.3e %.3e mu %.3e a_aff %.4f alpha %.4f\n", it, nrm[0], nrm[1],
                    nrm[2], nrm[3], mu, a_aff, alpha);
        double a = alpha * ((1.0 - alpha) * 0.99 + alpha * 0.9999999);
        size_t nw = ((size_t)N + 1) * NW, np = (size_t)N * NX;
        for (size_t i = 0; i < nw; i++) w[i] += a * ws->dw[i];
        for (size_t i = 0; i < np; i++) pi[i] += a * ws->dpi[i];
        for (size_t i = 0; i < nlt; i++)
            if (ws->mask[i]) {
                lam[i] += a * ws->dlam[i];
                t[i] += a * ws->dt[i];
                if (lam[i] < lam_min) lam[i] = lam_min;
                if (t[i] < t_min) t[i] = t_min;
            }
        ipm_residuals(N, H, g, b, A, B, lb, ub, w, pi, lam, t, ws, nrm, &mu);
    }
    if (iters) *iters = it;
    if (res4) memcpy(res4, nrm, sizeof nrm);
    return status;
}

int orc_qp_ipm(int N, const double *H, const double *g, const double *b, const double *A, const double *B,
               const double *lb, const double *ub, const double *dx0, double *w, double *pi, double *lam,
               double *t, double tol, int iter_max, int *iters, double *res4)
{
    ipm_ws ws;
    double *mem = (double *)calloc(ipm_ws_doubles(N), sizeof(double));
    unsigned char *mask = (unsigned char *)calloc(((size_t)N + 1) * 24, 1);
    if (!mem || !mask) { free(mem); free(mask); return -1; }
    ipm_ws_carve(&ws, N, mem, mask);
    int st = ipm_solve(N, H, g, b, A, B, lb, ub, dx0, w, pi, lam, t, tol, iter_max, iters, res4, &ws);
    free(mem); free(mask);
    return st;
}

/* the fast path on a raw QP (tests): 1 = accepted, (w, pi, lam, t) hold the solution; 0 = rejected, they are untouched */
int orc_qp_fast(int N, const double *H, const double *g, const double *b, const double *A, const double *B,
                const double *lb, const double *ub, const double *dx0, double *w, double *pi, double *lam, double *t)
{
    ipm_ws ws;
    double *mem = (double *)calloc(ipm_ws_doubles(N), sizeof(double));
    unsigned char *mask = (unsigned char *)calloc(((size_t)N + 1) * 24, 1);
    if (!mem || !mask) { free(mem); free(mask); return -1; }
    ipm_ws_carve(&ws, N, mem, mask);
    int ok = ipm_fast_path(N, H, g, b, A, B, lb, ub, dx0, w, pi, lam, t, &ws);
    free(mem); free(mask);
    return ok;
}

/* ------------------------------------------------------------------ */
/* NLP layer: acados SQP / SQP_RTI restated                            */
/* ------------------------------------------------------------------ */
struct orc_solver {
    orc_robot rb;
    orc_params p;
    int N;
    double A[144], B[72];
    double a12[6], a22[6], b1[6], b2[6];
    /* NLP iterate + multipliers (acados nlp_out) */
    double *x, *u, *pi, *lam, *t;
    /* QP memory (acados qp_out; HPIPM warm_start = 2 reuses it) */
    double *qw, *qpi, *qlam, *qt;
    /* linearisation */
    double *r, *H, *g, *b, *lb, *ub;
    double cost;
    /* merit weights (MERIT_BACKTRACKING) */
    double *mw_dyn, *mw_ineq, mw_x0[12];
    double *tx, *tu; /* trial iterate */
    ipm_ws ws;
    double *wsmem;
    unsigned char *mask;
    int fast_skip, fast_back; /* fast path: QPs left before the next attempt; length of the current suspension */
    int n_fast, n_reject;  /* diagnostics: accepted / rejected attempts */
};

orc_solver *orc_solver_create(const orc_robot *rb, const orc_params *p)
{
    orc_solver *s = (orc_solver *)calloc(1, sizeof *s);
    if (!s) return NULL;
    s->rb = *rb; s->p = *p;
    int N = s->N = p->N;
    size_t n1 = (size_t)N + 1;
    orc_lti(p->wcv, p->dt, s->a12, s->a22, s->b1, s->b2);
    /* Ad, Bd: prediction_model.py:104-112 */
    for (int j = 0; j < 6; j++) {
        s->A[j * 12 + j] = 1.0;
        s->A[j * 12 + 6 + j] = s->a12[j];
        s->A[(6 + j) * 12 + 6 + j] = s->a22[j];
        s->B[j * 6 + j] = s->b1[j];
        s->B[(6 + j) * 6 + j] = s->b2[j];
    }
#define ALLOC(n) (double *)calloc((n), sizeof(double))
    s->x = ALLOC(n1 * 12); s->u = ALLOC(n1 * 6); s->pi = ALLOC(n1 * 12); s->lam = ALLOC(n1 * 24); s->t = ALLOC(n1 * 24);
    s->qw = ALLOC(n1 * 18); s->qpi = ALLOC(n1 * 12); s->qlam = ALLOC(n1 * 24); s->qt = ALLOC(n1 * 24);
    s->r = ALLOC(n1 * NR); s->H = ALLOC(n1 * 324); s->g = ALLOC(n1 * 18); s->b = ALLOC(n1 * 12);
    s->lb = ALLOC(n1 * 12); s->ub = ALLOC(n1 * 12);
    s->mw_dyn = ALLOC(n1 * 12); s->mw_ineq = ALLOC(n1 * 24);
    s->tx = ALLOC(n1 * 12); s->tu = ALLOC(n1 * 6);
    s->wsmem = ALLOC(ipm_ws_doubles(N));
    s->mask = (unsigned char *)calloc(n1 * 24, 1);
#undef ALLOC
    ipm_ws_carve(&s->ws, N, s->wsmem, s->mask);
    /* acados initial guess: x_k = x0 for all k, u_k = 0, multipliers 0 (SURVEY A.7 iv) */
    for (int k = 0; k <= N; k++) {
        memcpy(s->x + (size_t)k * 12, p->q0, 6 * sizeof(double));
        memcpy(s->x + (size_t)k * 12 + 6, p->qdot0, 6 * sizeof(double));
    }
    return s;
}

void orc_solver_destroy(orc_solver *s)
{
    if (!s) return;
    free(s->x); free(s->u); free(s->pi); free(s->lam); free(s->t);
    free(s->qw); free(s->qpi); free(s->qlam); free(s->qt);
    free(s->r); free(s->H); free(s->g); free(s->b); free(s->lb); free(s->ub);
    free(s->mw_dyn); free(s->mw_ineq); free(s->tx); free(s->tu);
    free(s->wsmem); free(s->mask);
    free(s);
}

void orc_solver_get_iterate(const orc_solver *s, double *x, double *u, double *pi)
{
    if (x) memcpy(x, s->x, ((size_t)s->N + 1) * 12 * sizeof(double));
    if (u) memcpy(u, s->u, (size_t)s->N * 6 * sizeof(double));
    if (pi) memcpy(pi, s->pi, (size_t)s->N * 12 * sizeof(double));
}

static void weights17(const orc_params *p, double *W)
{
    for (int i = 0; i < 5; i++) W[i] = p->w_task[i];
    for (int i = 0; i < 6; i++) W[5 + i] = 2.0 * p->w_u;   /* trajectory_optimizer.py:148 */
    for (int i = 0; i < 6; i++) W[11 + i] = p->w_qddot;    /* trajectory_optimizer.py:150 */
}

/* total cost sum_k dt * 1/2 r'Wr at (x,u) -- acados get_cost() (simulator.py:221) */
static double eval_cost(const orc_solver *s, const double *x, const double *u)
{
    double W[NR], r[NR], c = 0;
    weights17(&s->p, W);
    for (int k = 0; k < s->N; k++) {
        orc_stage_residual(&s->rb, &s->p, x + (size_t)k * 12, u + (size_t)k * 6, r, NULL);
        double sk = 0;
        for (int i = 0; i < NR; i++) sk += W[i] * r[i] * r[i];
        c += 0.5 * s->p.dt * sk;
    }
    return c;
}

/* Linearise at the current iterate: GN Hessian dt*Jr'WJr, gradient dt*Jr'Wr
 * (hessian_approx GAUSS_NEWTON, trajectory_optimizer.py:61; stage cost scaled by the
 * time step, SURVEY A.7 ii), dynamics defect, bounds relative to the iterate
 * (trajectory_optimizer.py:164-171; lbx on stages 1..N-1, lbu on 0..N-1). */
static void linearize(orc_solver *s)
{
    const int N = s->N;
    double W[NR], Jr[NR * NW];
    weights17(&s->p, W);
    s->cost = 0;
    for (int k = 0; k <= N; k++) {
        double *Hk = s->H + (size_t)k * 324, *gk = s->g + (size_t)k * 18, *rk = s->r + (size_t)k * NR;
        const double *xk = s->x + (size_t)k * 12, *uk = s->u + (size_t)k * 6;
        memset(Hk, 0, 324 * sizeof(double));
        memset(gk, 0, 18 * sizeof(double));
        if (k < N) {
            orc_stage_residual(&s->rb, &s->p, xk, uk, rk, Jr);
            double sk = 0;
            for (int i = 0; i < NR; i++) sk += W[i] * rk[i] * rk[i];
            s->cost += 0.5 * s->p.dt * sk;
            for (int a = 0; a < NW; a++) {
                double ga = 0;
                for (int i = 0; i < NR; i++) ga += Jr[i * NW + a] * W[i] * rk[i];
                gk[a] = s->p.dt * ga;
                for (int c = 0; c < NW; c++) {
                    double h = 0;
                    for (int i = 0; i < NR; i++) h += Jr[i * NW + a] * W[i] * Jr[i * NW + c];
                    Hk[a * NW + c] = s->p.dt * h;
                }
            }
            /* acados ocp_nlp_approximate_qp_matrices: "Levenberg Marquardt term: Ts[i] * levenberg_marquardt * eye()" */
            for (int a = 0; a < NW; a++) Hk[a * NW + a] += s->p.dt * s->p.lm;
            const double *xn = s->x + (size_t)(k + 1) * 12;
            double *bk = s->b + (size_t)k * 12;
            for (int i = 0; i < 12; i++) {
                double v = -xn[i];
                for (int j = 0; j < 12; j++) v += s->A[i * 12 + j] * xk[j];
                for (int j = 0; j < 6; j++) v += s->B[i * 6 + j] * uk[j];
                bk[i] = v;
            }
        }
        else {
            /* terminal stage: "1.0 * levenberg_marquardt * eye()" on the nx x nx block (nu_N = 0) */
            for (int a = 6; a < NW; a++) Hk[a * NW + a] += s->p.lm;
        }
        double *lbk = s->lb + (size_t)k * 12, *ubk = s->ub + (size_t)k * 12;
        for (int j = 0; j < 6; j++) {
            if (k < N) { lbk[j] = s->p.umin[j] - uk[j]; ubk[j] = s->p.umax[j] - uk[j]; }
            else { lbk[j] = -1e30; ubk[j] = 1e30; }
            if (k >= 1 && k < N) { lbk[6 + j] = s->p.qmin[j] - xk[j]; ubk[6 + j] = s->p.qmax[j] - xk[j]; }
            else { lbk[6 + j] = -1e30; ubk[6 + j] = 1e30; }
        }
    }
}

/* acados ocp_nlp_res_compute restated: inf-norms of stationarity, dynamics defect,
 * inequality residual (fun + t) and complementarity (lam*t) at the NLP iterate. */
static void nlp_residuals(const orc_solver *s, const double *xhat, double *res4)
{
    const int N = s->N;
    double rs = 0, re = 0, ri = 0, rc = 0;
    for (int k = 0; k <= N; k++) {
        double v[NW];
        const double *gk = s->g + (size_t)k * 18;
        for (int i = 0; i < NW; i++) v[i] = gk[i];
        if (k < N) {
            const double *pk = s->pi + (size_t)k * 12;
            for (int j = 0; j < 6; j++) {
                double a = 0;
                for (int i = 0; i < 12; i++) a += s->B[i * 6 + j] * pk[i];
                v[j] += a;
            }
            for (int j = 0; j < 12; j++) {
                double a = 0;
                for (int i = 0; i < 12; i++) a += s->A[i * 12 + j] * pk[i];
                v[6 + j] += a;
            }
        }
        if (k >= 1) for (int j = 0; j < 12; j++) v[6 + j] -= s->pi[(size_t)(k - 1) * 12 + j];
        const double *lbk = s->lb + (size_t)k * 12, *ubk = s->ub + (size_t)k * 12;
        const double *lk = s->lam + (size_t)k * 24, *tk = s->t + (size_t)k * 24;
        for (int j = 0; j < NB; j++) {
            if (lbk[j] > -BIG) {
                v[j] -= lk[j];
                double d = fabs(lbk[j] + tk[j]); /* (lb - v) + t, lbk is already lb - v */
                if (d > ri) ri = d;
                if (fabs(lk[j] * tk[j]) > rc) rc = fabs(lk[j] * tk[j]);
            }
            if (ubk[j] < BIG) {
                v[j] += lk[12 + j];
                double d = fabs(-ubk[j] + tk[12 + j]);
                if (d > ri) ri = d;
                if (fabs(lk[12 + j] * tk[12 + j]) > rc) rc = fabs(lk[12 + j] * tk[12 + j]);
            }
        }
        if (k == 0) for (int j = 0; j < 12; j++) v[6 + j] = 0;
        if (k == N) for (int j = 0; j < 6; j++) v[j] = 0;
        for (int i = 0; i < NW; i++) if (fabs(v[i]) > rs) rs = fabs(v[i]);
        if (k < N)
            for (int i = 0; i < 12; i++) if (fabs(s->b[(size_t)k * 12 + i]) > re) re = fabs(s->b[(size_t)k * 12 + i]);
    }
    for (int i = 0; i < 12; i++) if (fabs(xhat[i] - s->x[i]) > ri) ri = fabs(xhat[i] - s->x[i]);
    res4[0] = rs; res4[1] = re; res4[2] = ri; res4[3] = rc;
}

/* One QP.  `iters` counts Riccati FACTORISATIONS: one per interior-point iteration, plus one for a fast-path attempt.
 * The fast path is attempted at every QP, except that a REJECTED attempt (some bound within the margin: the factorisation
 * was wasted) suspends the attempts for the next 1, 2, 4, 8, 8, ... QPs of this solver (doubling while the rejections go on,
 * back to none after an acceptance), so that a simulation riding its bounds pays at most one wasted factorisation in nine. */
static int solve_qp(orc_solver *s, const double *xhat, int *iters, int first_qp)
{
    double dx0[12];
    int tried = 0, it = 0;
    for (int i = 0; i < 12; i++) dx0[i] = xhat[i] - s->x[i];
    if (s->p.fast_path && s->fast_skip > 0 && s->p.fast_path != 2) s->fast_skip--;
    else if (s->p.fast_path) {
        if (ipm_fast_path_as(s->N, s->H, s->g, s->b, s->A, s->B, s->lb, s->ub, dx0, s->qw, s->qpi, s->qlam, s->qt, &s->ws,
                             s->p.fast_path >= 3, first_qp, &tried)) {
            s->fast_back = 0; s->n_fast++;
            *iters = tried;
            return 0;
        }
        s->fast_back = s->fast_back ? (2 * s->fast_back < 8 ? 2 * s->fast_back : 8) : 1;
        s->fast_skip = s->fast_back;
        s->n_reject++;
    }
    int st = ipm_solve(s->N, s->H, s->g, s->b, s->A, s->B, s->lb, s->ub, dx0, s->qw, s->qpi, s->qlam, s->qt,
                       s->p.qp_tol, s->p.qp_iter_max, &it, NULL, &s->ws);
    *iters = it + tried;
    return st;
}

/* acados ocp_nlp_update_variables_sqp restated */
static void update_iterate(orc_solver *s, double alpha)
{
    const int N = s->N;
    for (int k = 0; k <= N; k++) {
        for (int i = 0; i < 12; i++) s->x[(size_t)k * 12 + i] += alpha * s->qw[(size_t)k * 18 + 6 + i];
        if (k < N) {
            for (int i = 0; i < 6; i++) s->u[(size_t)k * 6 + i] += alpha * s->qw[(size_t)k * 18 + i];
            for (int i = 0; i < 12; i++) {
                double *v = &s->pi[(size_t)k * 12 + i];
                *v += alpha * (s->qpi[(size_t)k * 12 + i] - *v);
            }
        }
        for (int i = 0; i < 24; i++) {
            double *l = &s->lam[(size_t)k * 24 + i], *tt = &s->t[(size_t)k * 24 + i];
            *l += alpha * (s->qlam[(size_t)k * 24 + i] - *l);
            *tt += alpha * (s->qt[(size_t)k * 24 + i] - *tt);
        }
    }
}

/* L1 merit function (acados ocp_nlp_evaluate_merit_fun restated) */
static double merit_fun(const orc_solver *s, const double *x, const double *u, const double *xhat)
{
    const int N = s->N;
    double m = eval_cost(s, x, u);
    for (int k = 0; k < N; k++) {
        const double *xk = x + (size_t)k * 12, *uk = u + (size_t)k * 6, *xn = x + (size_t)(k + 1) * 12;
        for (int i = 0; i < 12; i++) {
            double v = -xn[i];
            for (int j = 0; j < 12; j++) v += s->A[i * 12 + j] * xk[j];
            for (int j = 0; j < 6; j++) v += s->B[i * 6 + j] * uk[j];
            m += s->mw_dyn[(size_t)k * 12 + i] * fabs(v);
        }
        for (int j = 0; j < 6; j++) {
            double vl = s->p.umin[j] - uk[j], vu = uk[j] - s->p.umax[j];
            if (vl > 0) m += s->mw_ineq[(size_t)k * 24 + j] * vl;
            if (vu > 0) m += s->mw_ineq[(size_t)k * 24 + 12 + j] * vu;
            if (k >= 1) {
                vl = s->p.qmin[j] - xk[j]; vu = xk[j] - s->p.qmax[j];
                if (vl > 0) m += s->mw_ineq[(size_t)k * 24 + 6 + j] * vl;
                if (vu > 0) m += s->mw_ineq[(size_t)k * 24 + 18 + j] * vu;
            }
        }
    }
    for (int i = 0; i < 12; i++) m += s->mw_x0[i] * fabs(xhat[i] - x[i]);
    return m;
}

/* MERIT_BACKTRACKING (trajectory_optimizer.py:68): weights per Leineweber's rule, alpha
 * reduced by 0.7 down to alpha_min = 0.05, plain decrease test (acados defaults
 * alpha_reduction 0.7, alpha_min 0.05, line_search_use_sufficient_descent 0). */
static double line_search(orc_solver *s, const double *xhat, int sqp_iter)
{
    const int N = s->N;
    /* multiplier of the eliminated x_0 constraint from stage-0 stationarity of the QP */
    double nu0[12];
    for (int j = 0; j < 12; j++) {
        double v = s->g[6 + j];
        for (int c = 0; c < NW; c++) v += s->H[(6 + j) * NW + c] * s->qw[c];
        for (int i = 0; i < 12; i++) v += s->A[i * 12 + j] * s->qpi[i];
        nu0[j] = fabs(v);
    }
    for (int k = 0; k <= N; k++) {
        for (int i = 0; i < 12 && k < N; i++) {
            double a = fabs(s->qpi[(size_t)k * 12 + i]), *wv = &s->mw_dyn[(size_t)k * 12 + i];
            if (sqp_iter == 0) *wv = a; else { double h = 0.5 * (*wv + a); *wv = a > h ? a : h; }
        }
        for (int i = 0; i < 24; i++) {
            double a = fabs(s->qlam[(size_t)k * 24 + i]), *wv = &s->mw_ineq[(size_t)k * 24 + i];
            if (sqp_iter == 0) *wv = a; else { double h = 0.5 * (*wv + a); *wv = a > h ? a : h; }
        }
    }
    for (int i = 0; i < 12; i++) {
        if (sqp_iter == 0) s->mw_x0[i] = nu0[i];
        else { double h = 0.5 * (s->mw_x0[i] + nu0[i]); s->mw_x0[i] = nu0[i] > h ? nu0[i] : h; }
    }
    const double m0 = merit_fun(s, s->x, s->u, xhat);
    double alpha = 1.0;
    while (alpha >= 0.05) {
        for (int k = 0; k <= N; k++) {
            for (int i = 0; i < 12; i++) s->tx[(size_t)k * 12 + i] = s->x[(size_t)k * 12 + i] + alpha * s->qw[(size_t)k * 18 + 6 + i];
            if (k < N)
                for (int i = 0; i < 6; i++) s->tu[(size_t)k * 6 + i] = s->u[(size_t)k * 6 + i] + alpha * s->qw[(size_t)k * 18 + i];
        }
        const double m1 = merit_fun(s, s->tx, s->tu, xhat);
        if (getenv("ORC_DEBUG")) fprintf(stderr, "line search sqp %d alpha %.4f: merit %.17g vs %.17g (%+.3e)\n", sqp_iter, alpha, m1, m0, m1 - m0);
        if (m1 < m0) break;
        alpha *= 0.7;
    }
    return alpha;
}

int orc_solver_step(orc_solver *s, const double *xhat, double *u0, int *sqp_iter_out, int *qp_iter_out,
                    double *res4, double *cost)
{
    int status = 0, sqp_iter = 0, qp_iter = 0, it;
    double res[4] = {0, 0, 0, 0};
    if (s->p.solver_type == 1) {
        /* SQP_RTI: one linearisation, one QP, full step (acados ocp_nlp_sqp_rti) */
        linearize(s);
        int qs = solve_qp(s, xhat, &it, 1);
        qp_iter += it;
        sqp_iter = 1;
        if (qs != 0 && qs != 1) status = 4; /* ACADOS_QP_FAILURE; iterate left untouched */
        else update_iterate(s, 1.0);
        /* get_residuals() re-evaluates for RTI (simulator.py:219); get_cost() evaluates
         * at the current iterate (simulator.py:221) */
        linearize(s);
        nlp_residuals(s, xhat, res);
    } else {
        status = 2; /* ACADOS_MAXITER unless the loop says otherwise */
        for (sqp_iter = 0; sqp_iter < s->p.max_iter; sqp_iter++) {
            linearize(s);
            nlp_residuals(s, xhat, res);
            {   /* acados ocp_nlp_sqp: res_stat < tol_stat && res_eq < tol_eq && res_ineq < tol_ineq && res_comp < tol_comp */
                const double te = s->p.tol_eq > 0 ? s->p.tol_eq : s->p.tol, ti = s->p.tol_ineq > 0 ? s->p.tol_ineq : s->p.tol,
                             tc = s->p.tol_comp > 0 ? s->p.tol_comp : s->p.tol;
                if (res[0] < s->p.tol && res[1] < te && res[2] < ti && res[3] < tc) { status = 0; break; }
            }
            if (res[0] != res[0] || s->cost != s->cost) { status = 1; break; }
            int qs = solve_qp(s, xhat, &it, sqp_iter == 0);
            qp_iter += it;
            if (qs != 0 && qs != 1) { status = 4; break; }
            double alpha = line_search(s, xhat, sqp_iter);
            update_iterate(s, alpha);
        }
        if (status == 2) linearize(s); /* cost at the final iterate */
    }
    if (u0) memcpy(u0, s->u, 6 * sizeof(double));
    if (sqp_iter_out) *sqp_iter_out = sqp_iter;
    if (qp_iter_out) *qp_iter_out = qp_iter;
    if (res4) memcpy(res4, res, sizeof res);
    if (cost) *cost = s->cost;
    return status;
}

/* ------------------------------------------------------------------ */
/* closed loop: Simulator.run (simulator.py:199-241)                   */
/* ------------------------------------------------------------------ */
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void log_state(const orc_robot *rb, const orc_output *o, int T1, int col, const double *z, const double *u)
{
    double pose[12], rpy[3], J[36];
    if (o->z) for (int i = 0; i < 12; i++) o->z[(size_t)i * T1 + col] = z[i];
    if (o->u) for (int i = 0; i < 6; i++) o->u[(size_t)i * T1 + col] = u[i];
    orc_fk(rb, z, pose);
    orc_rpy(pose, rpy);
    if (o->ee_pose) for (int i = 0; i < 12; i++) o->ee_pose[(size_t)i * T1 + col] = pose[i];
    if (o->ee_rpy) for (int i = 0; i < 3; i++) o->ee_rpy[(size_t)i * T1 + col] = rpy[i];
    if (o->ee_vel) {
        orc_jacobian_world(rb, z, J);
        for (int i = 0; i < 6; i++) {
            double s = 0;
            for (int j = 0; j < 6; j++) s += J[i * 6 + j] * z[6 + j];
            o->ee_vel[(size_t)i * T1 + col] = s;
        }
    }
}

int orc_run(const orc_robot *rb, const orc_params *p, orc_output *o)
{
    orc_solver *s = orc_solver_create(rb, p);
    if (!s) return -1;
    const int T1 = p->Nsim + 1;
    double z[12], zn[12], u[6], res[4], cost;
    memcpy(z, p->q0, 6 * sizeof(double));
    memcpy(z + 6, p->qdot0, 6 * sizeof(double));
    /* simulation_model.py:32-35: z[:,0]=z0, u[:,0]=u0=qdot_0 (simulator.py:81) */
    log_state(rb, o, T1, 0, z, p->qdot0);
    for (int i = 0; i < p->Nsim; i++) {
        int sqp_iter, qp_iter;
        double t0 = now_s();
        int st = orc_solver_step(s, z, u, &sqp_iter, &qp_iter, res, &cost);
        double t1 = now_s();
        if (o->status) o->status[i] = st;
        if (o->sqp_iter) o->sqp_iter[i] = sqp_iter;
        if (o->qp_iter) o->qp_iter[i] = qp_iter;
        if (o->residuals) memcpy(o->residuals + (size_t)i * 4, res, sizeof res);
        if (o->cost) o->cost[i] = cost;
        if (o->solver_time) o->solver_time[i] = t1 - t0;
        orc_plant_step(p->integrator, p->wcv, p->dt, z, u, zn); /* simulation_model.py:85-91 */
        memcpy(z, zn, sizeof z);
        log_state(rb, o, T1, i + 1, z, u);
    }
    if (getenv("ORC_FAST_STATS")) fprintf(stderr, "fast path: accepted %d rejected %d of %d steps\n", s->n_fast, s->n_reject, p->Nsim);
    orc_solver_destroy(s);
    return 0;
}
