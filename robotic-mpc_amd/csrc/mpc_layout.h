// mpc_layout.h -- data layout of the batched MPC rollout engine (gfx950).
//
// One wavefront (64 lanes) owns one simulation instance.  All per-instance solver
// state lives in an HBM workspace laid out STAGE-MAJOR: for every array the record of
// stage k is contiguous, so (a) the sequential Riccati sweeps read/write one stage's
// K/P/Rinv blocks with fully coalesced wave accesses and (b) the element-wise IPM
// phases walk the flat arrays with lane-contiguous addresses.
//
// Reference being replaced: simulator.py:199-241 (Simulator.run) and the acados /
// HPIPM solver behind trajectory_optimizer.py:183-186.
#pragma once

#define MPC_HD __host__ __device__ __forceinline__
#define MPC_HDN __host__ __device__ __noinline__

namespace mpcb {

constexpr int WAVE = 64;
constexpr int NQ = 6;   // joints
constexpr int NX = 12;  // state  x = [q; qdot]              (prediction_model.py:46)
constexpr int NU = 6;   // input  u = qdot_ref                (prediction_model.py:47)
constexpr int NW = 18;  // stage variable w = [u; q; qdot]
constexpr int NB = 12;  // bounded components per stage: u (6) then q (6)
constexpr int NL = 24;  // multipliers / slacks per stage: lower (12) then upper (12)
constexpr int NTASK = 5;

// ---- per-stage record widths (doubles) ------------------------------------------
constexpr int W_X = 12;
constexpr int W_U = 6;
constexpr int W_QW = 18;    // QP primal [du; dq; dqdot]
constexpr int W_PI = 12;
constexpr int W_LAM = 24;
constexpr int W_T = 24;
constexpr int W_LIN = 16;   // [0..4] r = g - gref, [5..9] y = W (r + G delta), pad
constexpr int W_RIC = 80;   // everything the factorisation sweep reads for one stage, contiguous:
                            // [0..29] Gq (5x6), [30..35] gv5, [36..47] Gamma, [48..65] gt, [66..77] rb
constexpr int W_BD = 12;    // dynamics defect of the NLP iterate
constexpr int W_RG = 18;
constexpr int W_RD = 24;
constexpr int W_RM = 24;
constexpr int W_DW = 18;
constexpr int W_DPI = 12;
constexpr int W_DLAM = 24;
constexpr int W_DT = 24;
constexpr int W_FAC = 128;  // what the solve sweeps read for one stage, contiguous:
                            // [0..71] Kfb = R^-1 S (6x12), [72..107] R^-1 (6x6), [108..113] h_u, [114..125] p_k
constexpr int W_PM = 144;   // cost-to-go matrix P_k (12x12, full)
constexpr int W_MW = 36;    // merit weights: dyn (12) + ineq (24)    (SQP only)

constexpr int LIN_R = 0, LIN_Y = 5;
constexpr int RIC_GQ = 0, RIC_GV = 30, RIC_GAM = 36, RIC_GT = 48, RIC_RB = 66;
constexpr int FAC_K = 0, FAC_RI = 72, FAC_HU = 108, FAC_PV = 114;

constexpr int STAGE_DOUBLES = W_X + W_U + W_QW + W_PI + W_LAM + W_T + W_LIN + W_RIC + W_BD + W_RG + W_RD + W_RM +
                              W_DW + W_DPI + W_DLAM + W_DT + W_FAC + W_PM +
                              /* NLP multipliers + trial iterate + merit weights (SQP) */
                              W_PI + W_LAM + W_T + W_X + W_U + W_MW;

// Kinematic constants (robots.py KinematicChain.packed): 105 doubles
struct Robot {
    double place[7][12];  // [R row-major (9); p (3)] of joint i in its parent, [6] = EE frame
    double axis[6][3];
    double t_ee[3];       // prediction_model.py:9
};

// Per-instance parameters, one record per instance in HBM (packed by mpcb_pack_params).
struct InstParams {
    double dt, tol, qp_tol, w_u, w_qddot, px_ref, vy_ref, pad0;
    double wcv[6], q0[6], qdot0[6], qmin[6], qmax[6], umin[6], umax[6];
    double coeffs[6];   // a b c d e f   (surface.py:14-17)
    double w_task[5];   // trajectory_optimizer.py:44-48
    double pad1;
    // derived on the host in C (prediction_model.py:87-115, 322-326)
    double a12[6], a22[6], b1[6], b2[6];
    double cq[6];       // qddot gain (1-a22)/Ts
};

// Batch-uniform problem description.
struct Problem {
    int batch;
    int N;            // prediction_horizon
    int Nsim;         // closed-loop steps
    int solver_type;  // 0 = SQP, 1 = SQP_RTI
    int max_iter;     // nlp_solver_max_iter
    int qp_iter_max;  // HPIPM iter_max
    int fixed_step;   // globalization FIXED_STEP instead of MERIT_BACKTRACKING
    int pad;
};

// Device pointers to the result logs, batch-major, same per-instance shapes as the
// reference's logs (simulation_model.py:25-29, simulator.py:59-65).
struct Outputs {
    double *z;         // [batch][12][Nsim+1]
    double *u;         // [batch][6][Nsim+1]
    double *ee_pose;   // [batch][12][Nsim+1]
    double *ee_rpy;    // [batch][3][Nsim+1]
    double *ee_vel;    // [batch][6][Nsim+1]
    int *status;       // [batch][Nsim]
    int *sqp_iter;     // [batch][Nsim]
    int *qp_iter;      // [batch][Nsim]
    double *residuals; // [batch][Nsim][4]
    double *cost;      // [batch][Nsim]
    double *solver_time; // [batch][Nsim] seconds (device realtime counter)
};

// Views into one instance's workspace.
struct Ws {
    double *X, *U, *QW, *QPI, *QLAM, *QT, *LIN, *RIC, *BD, *RG, *RD, *RM, *DW, *DPI, *DLAM, *DT, *FAC, *PM;
    double *NPI, *NLAM, *NT, *TX, *TU, *MW;  // SQP extras
    double *state;                           // persistent scalars between launches
};

// [0..11] plant state z, [12] cost of the held linearisation, [13..24] merit weights of the
// x0 constraint, [25] linearisation-valid flag, [32..47] profile counters (diagnostic build)
constexpr int STATE_DOUBLES = 64;
constexpr int NPROF = 16;

MPC_HD size_t ws_doubles_per_instance(int N)
{
    return (size_t)(N + 1) * STAGE_DOUBLES + STATE_DOUBLES;
}

MPC_HD Ws ws_carve(double *base, int N)
{
    const size_t n1 = (size_t)N + 1;
    Ws w;
    double *p = base;
    w.X = p; p += n1 * W_X;
    w.U = p; p += n1 * W_U;
    w.QW = p; p += n1 * W_QW;
    w.QPI = p; p += n1 * W_PI;
    w.QLAM = p; p += n1 * W_LAM;
    w.QT = p; p += n1 * W_T;
    w.LIN = p; p += n1 * W_LIN;
    w.RIC = p; p += n1 * W_RIC;
    w.BD = p; p += n1 * W_BD;
    w.RG = p; p += n1 * W_RG;
    w.RD = p; p += n1 * W_RD;
    w.RM = p; p += n1 * W_RM;
    w.DW = p; p += n1 * W_DW;
    w.DPI = p; p += n1 * W_DPI;
    w.DLAM = p; p += n1 * W_DLAM;
    w.DT = p; p += n1 * W_DT;
    w.FAC = p; p += n1 * W_FAC;
    w.PM = p; p += n1 * W_PM;
    w.NPI = p; p += n1 * W_PI;
    w.NLAM = p; p += n1 * W_LAM;
    w.NT = p; p += n1 * W_T;
    w.TX = p; p += n1 * W_X;
    w.TU = p; p += n1 * W_U;
    w.MW = p; p += n1 * W_MW;
    w.state = p;
    return w;
}

// LDS working set of one wavefront (one instance).
constexpr int STG_DOUBLES = 2 * 288;  // two staging buffers
struct Smem {
    InstParams P;       // this instance's parameters (lane-indexed reads stay on chip)
    Robot rb;
    double stg[STG_DOUBLES];  // stage record prefetched from HBM one stage ahead of the sweeps
    double M[2][144];   // P_{k+1} / P_k double buffer during the factorisation sweep
    double pv[2][12];
    double Rt[36];      // R~ = H_uu + Gamma_u + B'MB
    double St[72];      // S~ = H_ux + B'MA           (6x12)
    double Kf[72];      // R~^-1 S~
    double mt[12];      // p_{k+1} + P_{k+1} rb_k
    double dx[2][12];
    double du[6];
    double red[5][WAVE];
    double xhat[12];    // current plant state (feedback, simulator.py:206)
    double u0[6];
    double logv[40];
};

}  // namespace mpcb
