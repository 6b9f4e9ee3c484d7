// mpc_kin.h -- analytic kinematics of the 6-DoF chain, per lane (one lane = one stage).
//
// Replaces the CasADi-generated cost_y_fun / cost_y_fun_jac_ut_xt of the reference
// (built from prediction_model.py:126-173, 256-314 and trajectory_optimizer.py:104-126)
// and Pinocchio's FK / frame Jacobian used by the plant log (simulation_model.py:60-77).
// All loops have compile-time bounds and are fully unrolled: everything stays in VGPRs.
#pragma once
#include <math.h>

#include "mpc_layout.h"

namespace mpcb {

struct V3 {
    double x, y, z;
};
MPC_HD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
MPC_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
MPC_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
MPC_HD V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
MPC_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MPC_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

struct M3 {
    V3 r0, r1, r2;  // rows
};
MPC_HD V3 mul(const M3 &A, V3 x) { return v3(dot(A.r0, x), dot(A.r1, x), dot(A.r2, x)); }
MPC_HD V3 col(const M3 &A, int c)
{
    return c == 0 ? v3(A.r0.x, A.r1.x, A.r2.x) : (c == 1 ? v3(A.r0.y, A.r1.y, A.r2.y) : v3(A.r0.z, A.r1.z, A.r2.z));
}
MPC_HD M3 mul(const M3 &A, const M3 &B)
{
    const V3 c0 = col(B, 0), c1 = col(B, 1), c2 = col(B, 2);
    M3 C;
    C.r0 = v3(dot(A.r0, c0), dot(A.r0, c1), dot(A.r0, c2));
    C.r1 = v3(dot(A.r1, c0), dot(A.r1, c1), dot(A.r1, c2));
    C.r2 = v3(dot(A.r2, c0), dot(A.r2, c1), dot(A.r2, c2));
    return C;
}
MPC_HD M3 load_m3(const double *p)
{
    M3 A;
    A.r0 = v3(p[0], p[1], p[2]); A.r1 = v3(p[3], p[4], p[5]); A.r2 = v3(p[6], p[7], p[8]);
    return A;
}
// sin and cos of a joint angle.  The library sincos carries a large-argument path (Payne-Hanek, v_trig_preop_f64) that a joint angle never
// takes but that is compiled into every one of the six calls of a linearisation (~90 instructions each, and its temporaries are where the
// 256-register builds spill).  Here: Cody-Waite reduction by pi/2 in two fused steps (exact products, k = nearest integer) and the two
// minimax kernels of fdlibm (k_sin.c / k_cos.c, public domain; < 1 ulp on [-pi/4, pi/4]) -- ~35 instructions, branch-free, absolute error
// <= 2e-16 for |th| <= 1e6 rad (tests/test_oracle.py checks it against libm on a grid; beyond that the reduction loses bits gradually).
MPC_HD void sincos_joint(double th, double *sn, double *cs)
{
    const double kf = rint(th * 6.36619772367581382433e-01);          // 2/pi
    double r = fma(-kf, 1.57079632679489655800e+00, th);                // pi/2, rounded
    r = fma(-kf, 6.12323399573676603587e-17, r);                        // pi/2 - the above
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06); ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03); ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s0 = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07); pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03); pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c0 = fma(z * z, pc, fma(z, -0.5, 1.0));
    const int q = (int)kf & 3;                                          // quadrant
    const double sv = (q & 1) ? c0 : s0, cv = (q & 1) ? s0 : c0;
    *sn = (q & 2) ? -sv : sv;
    *cs = ((q + 1) & 2) ? -cv : cv;
}

// Rodrigues rotation about a unit axis
MPC_HD M3 axis_rot(V3 a, double th)
{
    double s, c;
#ifdef MPCB_LIBM_SINCOS      // (A/B builds)
    sincos(th, &s, &c);
#else
    sincos_joint(th, &s, &c);
#endif
    const double v = 1.0 - c;
    M3 R;
    R.r0 = v3(c + v * a.x * a.x, v * a.x * a.y - s * a.z, v * a.x * a.z + s * a.y);
    R.r1 = v3(v * a.y * a.x + s * a.z, c + v * a.y * a.y, v * a.y * a.z - s * a.x);
    R.r2 = v3(v * a.z * a.x - s * a.y, v * a.z * a.y + s * a.x, c + v * a.z * a.z);
    return R;
}

struct Kin {
    V3 o[6];  // joint origins (WORLD)
    V3 z[6];  // joint axes (WORLD)
    V3 p;     // EE origin
    M3 R;     // EE rotation
};

// forwardKinematics + updateFramePlacements (prediction_model.py:126-132)
MPC_HD void kin_eval(const Robot &rb, const double *q, Kin &k)
{
    M3 R;
    R.r0 = v3(1, 0, 0); R.r1 = v3(0, 1, 0); R.r2 = v3(0, 0, 1);
    V3 p = v3(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        p = p + mul(R, v3(rb.place[i][9], rb.place[i][10], rb.place[i][11]));
        R = mul(R, load_m3(rb.place[i]));
        const V3 a = v3(rb.axis[i][0], rb.axis[i][1], rb.axis[i][2]);
        k.o[i] = p;
        k.z[i] = mul(R, a);
        R = mul(R, axis_rot(a, q[i]));
    }
    k.p = p + mul(R, v3(rb.place[6][9], rb.place[6][10], rb.place[6][11]));
    k.R = mul(R, load_m3(rb.place[6]));
}

// Task functions g1..g5 minus their references (trajectory_optimizer.py:109-126), and, when
// JAC, the Jacobian rows wrt q (5x6) and d g5 / d qdot (6).  Output goes straight into a G2
// (LINR) stage record: rec[O_R..] = r, rec[O_GQ..] = Gq, rec[O_GV..] = gv5.
// rec_r receives r, rec_g the Jacobian entries (the same record for both, or r kept in registers while the
// Jacobian goes straight to memory).
// (PG: the Jacobian's destination keeps its address space -- a `double *` parameter would turn the throughput engine's stores to its HBM
// record into flat_store, which occupy the LDS queue as well)
template <bool JAC, class PG>
MPC_HD void task_lin(const Robot &rb, const InstParams &P, const double *q, const double *qd, double *rec_r, PG rec_g)
{
    Kin k;
    kin_eval(rb, q, k);
    const V3 yh = col(k.R, 1), zh = col(k.R, 2);
    const V3 tw = mul(k.R, v3(rb.t_ee[0], rb.t_ee[1], rb.t_ee[2]));
    const V3 pt = k.p + tw;
    const double a = P.coeffs[0], b = P.coeffs[1], c = P.coeffs[2], d = P.coeffs[3], e = P.coeffs[4], f = P.coeffs[5];
    const double X = pt.x, Y = pt.y;
    const double S = a * X * X + b * Y * Y + c * X * Y + d * X + e * Y + f;          // surface.py:21
    const double Sx = 2 * a * X + c * Y + d, Sy = 2 * b * Y + c * X + e;             // surface.py:246-247
    const double nn = sqrt(Sx * Sx + Sy * Sy + 1.0);                                // surface.py:255
    const double inn = 1.0 / nn;
    const V3 n = v3(Sx * inn, Sy * inn, -inn);
    V3 vl = v3(0, 0, 0), om = v3(0, 0, 0);
    V3 cj[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        cj[j] = cross(k.o[j], k.z[j]);  // WORLD spatial Jacobian, linear rows (prediction_model.py:163-164)
        vl = vl + qd[j] * cj[j];
        om = om + qd[j] * k.z[j];
    }
    const V3 s = vl + cross(om, tw);
    rec_r[O_R + 0] = (S - pt.z) - 0.0;
    rec_r[O_R + 1] = dot(n, zh) - 1.0;
    rec_r[O_R + 2] = yh.x - 0.0;
    rec_r[O_R + 3] = pt.x - P.px_ref;
    rec_r[O_R + 4] = dot(yh, s) - P.vy_ref;  // v_task,y = (R^T (v + w x t_w))_y, prediction_model.py:313
    if (!JAC) return;
    const V3 mX = v3(2 * a, c, 0), mY = v3(c, 2 * b, 0);
    const double pX = dot(n, mX), pY = dot(n, mY);
    const V3 nX = inn * (mX - pX * n), nY = inn * (mY - pY * n);
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const V3 zi = k.z[i];
        const V3 dpt = cross(zi, pt - k.o[i]);
        const V3 dzh = cross(zi, zh), dyh = cross(zi, yh), dtw = cross(zi, tw);
        rec_g[O_GQ + 0 * 6 + i] = Sx * dpt.x + Sy * dpt.y - dpt.z;
        const V3 dn = dpt.x * nX + dpt.y * nY;
        rec_g[O_GQ + 1 * 6 + i] = dot(dn, zh) + dot(n, dzh);
        rec_g[O_GQ + 2 * 6 + i] = dyh.x;
        rec_g[O_GQ + 3 * 6 + i] = dpt.x;
        V3 dvl = v3(0, 0, 0), otail = v3(0, 0, 0);
#pragma unroll
        for (int j = i + 1; j < 6; j++) {
            const V3 doj = cross(zi, k.o[j] - k.o[i]);
            const V3 dzj = cross(zi, k.z[j]);
            dvl = dvl + qd[j] * (cross(doj, k.z[j]) + cross(k.o[j], dzj));
            otail = otail + qd[j] * k.z[j];
        }
        const V3 dom = cross(zi, otail);
        const V3 ds = dvl + cross(dom, tw) + cross(om, dtw);
        rec_g[O_GQ + 4 * 6 + i] = dot(dyh, s) + dot(yh, ds);
        rec_g[O_GV + i] = dot(yh, cj[i] + cross(zi, tw));
#if defined(__HIP_DEVICE_COMPILE__)
        // one Jacobian column at a time: left alone, the scheduler interleaves all six (each ~100 independent operations) and the
        // 256-register builds spill ~60 values around this loop
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

template <bool JAC>
MPC_HD void task_lin(const Robot &rb, const InstParams &P, const double *q, const double *qd, double *rec)
{
    task_lin<JAC>(rb, P, q, qd, rec, rec);
}

// Plant log (simulation_model.py:60-77): pose = [p; R row-major], rpy, J_world * qdot.
MPC_HD void plant_log(const Robot &rb, const double *z, double *out33)
{
    Kin k;
    kin_eval(rb, z, k);
    out33[0] = k.p.x; out33[1] = k.p.y; out33[2] = k.p.z;
    out33[3] = k.R.r0.x; out33[4] = k.R.r0.y; out33[5] = k.R.r0.z;
    out33[6] = k.R.r1.x; out33[7] = k.R.r1.y; out33[8] = k.R.r1.z;
    out33[9] = k.R.r2.x; out33[10] = k.R.r2.y; out33[11] = k.R.r2.z;
    out33[12] = atan2(k.R.r2.y, k.R.r2.z);                                          // roll,  :66
    out33[13] = atan2(-k.R.r2.x, sqrt(k.R.r0.x * k.R.r0.x + k.R.r1.x * k.R.r1.x));  // pitch, :67
    out33[14] = atan2(k.R.r1.x, k.R.r0.x);                                          // yaw,   :68
    V3 vl = v3(0, 0, 0), om = v3(0, 0, 0);
#pragma unroll
    for (int j = 0; j < 6; j++) {
        vl = vl + z[6 + j] * cross(k.o[j], k.z[j]);
        om = om + z[6 + j] * k.z[j];
    }
    out33[15] = vl.x; out33[16] = vl.y; out33[17] = vl.z;
    out33[18] = om.x; out33[19] = om.y; out33[20] = om.z;
}

// Task errors of Simulator.errors (simulator.py:265-344) from one logged pose / velocity column:
// out7 = [e1, e2, e3, e4, e5, p_task_z, p_ee_y].  Reproduces the reference's e5 literally:
// v_task = R (v + (w . t_w)) -- a dot product broadcast onto the linear velocity, R not transposed
// (simulator.py:317) -- unlike the OCP's R^T (v + w x t_w) (prediction_model.py:313).
MPC_HD void task_errors(const InstParams &P, const Robot &rb, const double *pose12, const double *vel6, double *out7)
{
    const double r00 = pose12[3], r01 = pose12[4], r02 = pose12[5], r10 = pose12[6], r11 = pose12[7], r12 = pose12[8],
                 r20 = pose12[9], r21 = pose12[10], r22 = pose12[11];
    const double tx = rb.t_ee[0], ty = rb.t_ee[1], tz = rb.t_ee[2];
    const double twx = r00 * tx + r01 * ty + r02 * tz, twy = r10 * tx + r11 * ty + r12 * tz, twz = r20 * tx + r21 * ty + r22 * tz;  // :309
    const double X = pose12[0] + twx, Y = pose12[1] + twy, Z = pose12[2] + twz;                                                    // :314
    const double dotw = vel6[3] * twx + vel6[4] * twy + vel6[5] * twz;
    const double vty = r10 * (vel6[0] + dotw) + r11 * (vel6[1] + dotw) + r12 * (vel6[2] + dotw);                                   // :317, row 1
    const double a = P.coeffs[0], b = P.coeffs[1], c = P.coeffs[2], d = P.coeffs[3], e = P.coeffs[4], f = P.coeffs[5];
    const double nx = 2 * a * X + c * Y + d, ny = 2 * b * Y + c * X + e;
    const double nn = sqrt(nx * nx + ny * ny + 1.0);
    const double S = a * X * X + b * Y * Y + c * X * Y + d * X + e * Y + f;                                                        // :330
    out7[0] = S - Z;                                                   // e1 = g1            :331,337
    out7[1] = 1.0 - (nx / nn * r02 + ny / nn * r12 + (-1.0 / nn) * r22);   // e2 = 1 - n . z_task :332,338
    out7[2] = r01;                                                     // e3 = y_task[0]     :333,339
    out7[3] = P.px_ref - X;                                            // e4                 :334,340
    out7[4] = P.vy_ref - vty;                                          // e5                 :335,341
    out7[5] = Z;                                                       // p_task_z           :342
    out7[6] = pose12[1];                                               // p_ee_y             :344
}

}  // namespace mpcb
