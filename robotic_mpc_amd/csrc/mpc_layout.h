// mpc_layout.h -- data layout of the batched MPC rollout engine (gfx950).
//
// One workgroup (1-8 wavefronts) owns one simulation instance.  All per-instance solver state
// lives in an HBM workspace, STAGE-MAJOR, split into five record groups so that every pass of
// the solver streams whole contiguous stage records:
//     G1 ITER  iterate + QP iterate      G2 LINR  linearisation + Newton right-hand sides
//     G3 STEP  residuals + Newton step   G4 FACT  Riccati factor (K, R^-1 h_u, e, p, P, w, R^-1)
//     G5 SQPX  SQP-only extras (NLP multipliers, trial point, merit weights)
// A pass copies a CHUNK of consecutive stages HBM -> LDS in one coalesced burst (16 B per
// lane, all loads in flight together), works on the chunk entirely in LDS -- the sequential
// Riccati recursions never touch HBM inside the stage loop -- and writes its outputs back in
// one burst.  This matters on gfx950 because a `s_waitcnt vmcnt` that covers a store costs
// ~1 us (stores are acknowledged from the memory side), so per-stage loads/stores would
// serialise the whole recursion on memory latency.
//
// Reference being replaced: simulator.py:199-241 (Simulator.run) and the acados / HPIPM
// solver behind trajectory_optimizer.py:183-186.
#pragma once

#include <stddef.h>

#define MPC_HD __host__ __device__ __forceinline__
// a solver pass: its own function (own register allocation) so the kernel does not become one
// giant inlined body that the register allocator spills
#define MPC_PASS __host__ __device__ __noinline__

namespace mpcb {

constexpr int WAVE = 64;
constexpr int NWV_MAX = 8;    // at most eight wavefronts cooperate on one simulation
constexpr int NT_MAX = WAVE * NWV_MAX;
constexpr int NQ = 6;   // joints
constexpr int NX = 12;  // state  x = [q; qdot]              (prediction_model.py:46)
constexpr int NU = 6;   // input  u = qdot_ref                (prediction_model.py:47)
constexpr int NW = 18;  // stage variable w = [u; q; qdot]
constexpr int NB = 12;  // bounded components per stage: u (6) then q (6)
constexpr int NL = 24;  // multipliers / slacks per stage: lower (12) then upper (12)
constexpr int NTASK = 5;

// ---- G1 ITER ----------------------------------------------------------------------
constexpr int W1 = 96;
constexpr int O_X = 0, O_U = 12, O_QW = 18, O_QPI = 36, O_QLAM = 48, O_QT = 72;
// ---- G2 LINR ----------------------------------------------------------------------
constexpr int W2 = 112;
constexpr int O_R = 0;     // r = g - g_ref (5)
constexpr int O_Y = 5;     // y = W (r + G delta) (5)
constexpr int O_BD = 12;   // dynamics defect of the NLP iterate (12)
constexpr int O_GQ = 24;   // d g / d q (5x6)
constexpr int O_GV = 54;   // d g5 / d qdot (6)
constexpr int O_GAM = 60;  // Gamma = lam/t (12)
constexpr int O_GT = 72;   // condensed gradient of the Newton system (18)
constexpr int O_RB = 90;   // dynamics residual of the QP iterate (12)
constexpr int W2_LIN = 60;   // [R..GV]: what a linearisation produces
// ---- G3 STEP ----------------------------------------------------------------------
constexpr int W3 = 144;
constexpr int O_RG = 0, O_RD = 18, O_RM = 42;  // residuals g (18), d (24), m (24)
constexpr int O_DW = 66;    // Newton step [du; dq; dqdot] (18)
constexpr int O_DPI = 84;   // step of the multiplier of dynamics k-1 -> k (12)   (note the shift)
constexpr int O_DLAM = 96, O_DT = 120;
// ---- G4 FACT ----------------------------------------------------------------------
// Ordered by consumer: the forward sweeps read a prefix ([K..E] predictor, [K..P] final sweep), the
// corrector reads K and the tail [w, R~^-1].
constexpr int W4 = 228;
constexpr int O_K = 0;      // Kfb = R~^-1 S~ (6x12)
constexpr int O_VH = 72;    // R~^-1 h_u (6)
constexpr int O_E = 78;     // e = rb - B R~^-1 h_u (12): the part of dx_{k+1} that does not depend on dx_k
constexpr int O_PV = 90;    // p_k (12)
constexpr int O_PM = 102;   // P_k, symmetric 12x12 PACKED: upper triangle by rows (78), see tri()
constexpr int NPM = 78;
constexpr int O_WV = 180;   // w_k = P_{k+1} rb_k (12): reused by the corrector's backward solve
constexpr int O_RI = 192;   // R~^-1 (6x6)
constexpr int W4_AFF = O_PV;   // what the predictor sweep loads
constexpr int W4_FWD = O_WV;   // what the final forward sweep loads
// ---- G5 SQPX ----------------------------------------------------------------------
constexpr int W5 = 116;
constexpr int O_NPI = 0, O_NLAM = 12, O_NT = 36, O_TX = 60, O_TU = 72, O_MW = 78;  // MW: dyn 12 + ineq 24

constexpr int STAGE_DOUBLES = W1 + W2 + W3 + W4 + W5;
// [0..11] plant state z, [12] cost of the held linearisation, [13..24] merit weights of the x0
// constraint, [25] linearisation-valid flag, [26] fast path: QPs left before the next attempt, [27] length of the
// current suspension, [28] throughput engine, SQP_RTI: the slot of the stage records that holds the QP iterate (0: G1, 1: G3),
// [32..47] profile counters (diagnostic build)
constexpr int STATE_DOUBLES = 64;
constexpr int NPROF = 16;

// Kinematic constants (robots.py KinematicChain.packed): 105 doubles
struct Robot {
    double place[7][12];  // [R row-major (9); p (3)] of joint i in its parent, [6] = EE frame
    double axis[6][3];
    double t_ee[3];       // prediction_model.py:9
};

// Per-instance parameters, one record per instance in HBM (packed by pack_inst_params).
struct InstParams {
    double dt, tol, qp_tol, w_u, w_qddot, px_ref, vy_ref, integ;   // integ: plant integrator code
    double wcv[6], q0[6], qdot0[6], qmin[6], qmax[6], umin[6], umax[6];
    double coeffs[6];   // a b c d e f   (surface.py:14-17)
    double w_task[5];   // trajectory_optimizer.py:44-48
    double lm;          // acados levenberg_marquardt: dt*lm*I on the stage Hessians, lm*I on the terminal one
    // derived on the host in C (prediction_model.py:87-115, 322-326)
    double a12[6], a22[6], b1[6], b2[6];
    double cq[6];       // qddot gain (1-a22)/Ts
    // acados nlp_solver_tol_eq / _ineq / _comp (`tol` above is nlp_solver_tol_stat)
    double tol_eq, tol_ineq, tol_comp;
    double n_hor;       // this simulation's prediction horizon when it differs from the launch's (throughput engine: ragged batches); 0 = Problem::N
    double fast_off;    // 0: the bound-inactive fast path of the QP solve is on (mpc_ipm.h, ipm::FAST_MARGIN); 1: every QP through the interior-point loop
};

// Batch-uniform problem description (== mpcb_problem).
struct Problem {
    int batch;
    int N;            // prediction_horizon
    int Nsim;         // closed-loop steps
    int solver_type;  // 0 = SQP, 1 = SQP_RTI
    int max_iter;     // nlp_solver_max_iter
    int qp_iter_max;  // HPIPM iter_max
    int fixed_step;   // globalization FIXED_STEP instead of MERIT_BACKTRACKING
    int precision;    // 0: fp64 everywhere; 1: Riccati factor and solve sweeps in fp32 (throughput engine only)
};

// Device pointers to the result logs (== mpcb_result), batch-major, per-instance shapes of
// the reference's logs (simulation_model.py:25-29, simulator.py:59-65).
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
    double *errors;    // [batch][7][Nsim+1]  e1..e5, p_task_z, p_ee_y (simulator.py:265-344)
    double *plant_time; // [batch][Nsim] seconds: plant step + FK / J qdot / error logging (simulator.py:224-226)
};

// One instance's workspace: five stage-major group arrays + persistent scalars.
struct Ws {
    double *G1, *G2, *G3, *G4, *G5, *PH, *state;
};
// Segmented residency of the latency engine (horizons whose Riccati factor does not fit a CU's LDS): the solve sweeps hold the
// factor of one SEGMENT of SEG_T transitions at a time (16 chunks of SEG_L); PH keeps the chunk transition matrices (12x12 each).
// Segment geometry by pool size: a whole CU's pool holds 16 chunks of 7 transitions, half a pool (two simulations per CU) 8 chunks of 5.
constexpr int SEG_L_FULL = 7, SEG_J_FULL = 16, SEG_L_HALF = 5, SEG_J_HALF = 8, SEG_POOL_FULL = 17000;
MPC_HD size_t ws_phi_doubles(int N) { return (size_t)((N + SEG_L_HALF - 1) / SEG_L_HALF + 1) * 144; }
// Residency predicates of the latency engine -- ONE formula for the device (Engine::resident_ok / segment_ok) and for mpcb_setup's choice
// of the launch geometry: per stage the solve sweeps keep K (72) | R~^-1 h_u (6) | e (12) | p (12) in LDS, plus one 12x12
// chunk transition matrix per lane group (`groups` = min(lanes / 16, 16)); the rest of the pool is scratch.
constexpr int RS_PER_STAGE_L = 72 + 6 + 12 + 12;
MPC_HD bool lay_resident_ok(int N, int pool_n, int groups)
{
    const int NS = N + 1, scr = pool_n - (NS * RS_PER_STAGE_L + groups * 144);
    // scratch: the factorisation's double-buffered chunks (>= 6 stages) ; the corrector's gt (18) + w (12) + slots
    return scr >= 4096 && scr >= NS * 30 + 2 * groups * 12 + 16;
}
MPC_HD bool lay_segment_ok(int N, int pool_n, int groups)
{
    if (groups < 8 || pool_n < SEG_POOL_FULL || lay_resident_ok(N, pool_n, groups)) return false;
    const int Jc = groups < SEG_J_FULL ? groups : SEG_J_FULL, T = SEG_L_FULL * Jc;
    const int scr = pool_n - ((T + 1) * RS_PER_STAGE_L + Jc * 144);
    return scr >= (T + 1) * 30 + 2 * groups * 12 + 64;
}

// offset of entry (r, c), r <= c, of a symmetric 12x12 matrix stored as its upper triangle by rows
MPC_HD int tri(int r, int c) { return r * 12 - (r * (r - 1)) / 2 + (c - r); }
MPC_HD int tri_sym(int i, int j) { return i <= j ? tri(i, j) : tri(j, i); }

MPC_HD size_t ws_doubles_per_instance(int N) { return (size_t)(N + 1) * STAGE_DOUBLES + ws_phi_doubles(N) + STATE_DOUBLES; }

MPC_HD Ws ws_carve(double *base, int N)
{
    const size_t n1 = (size_t)N + 1;
    Ws w;
    double *p = base;
    w.G1 = p; p += n1 * W1;
    w.G2 = p; p += n1 * W2;
    w.G3 = p; p += n1 * W3;
    w.G4 = p; p += n1 * W4;
    w.G5 = p; p += n1 * W5;
    w.PH = p; p += ws_phi_doubles(N);
    w.state = p;
    return w;
}

// rows of the per-step trajectory log: z 12 | u 6 | ee_pose 12 | ee_rpy 3 | ee_vel 6 | errors 7
constexpr int LOG_ROWS = 46, LOGB = 8;

// Static LDS working set of one simulation (workgroup); the chunk pool follows it (dynamic LDS).
// Every array is 16-byte aligned: the compiler merges neighbouring doubles into ds_read/write_b128,
// and a b128 DS access off its 16-byte alignment is replayed at ~64 cycles (MI355X_MICROARCH.md, LDS).
struct Smem {
    InstParams P;       // this instance's parameters (lane-indexed reads stay on chip)
    alignas(16) Robot rb;
    Ws w;                  // this instance's workspace pointers: kept here because the engine object lives in
    int n_hor, pool_n;     // scratch memory inside a non-inlined pass (a flat load + full wait per use); LDS is ~10x closer
    int prog;              // progress of the state recursion (last finished stage), Ex::post / await
    int prog1;             // progress of the follower recursion (vector half of the factorisation sweep)
    int flg[2];            // factorisation pipeline: input chunks landed, factor buffers released (counted per background wavefront)
    int flg_pad;
    alignas(16) double mt2[2][16];     // p_{k+1} + P_{k+1} rb_k hand-over slots (host executor only)
    alignas(16) double pv[2][12];
    alignas(16) double Rt[36];      // R~ = H_uu + Gamma_u + B'MB
    alignas(16) double St2[2][72];  // S~ double buffer of the factorisation sweep (stage k read, k-1 written)
    alignas(16) double Kf[72];      // R~^-1 S~
    alignas(16) double dx[2][12];
    alignas(16) double ret[8];      // results a pass hands back to the solver driver (every wavefront writes the same values)
    alignas(16) double cen[4];      // resident sweeps: step length and the three centering sums S0, S1, S2
    alignas(16) double bon[24];     // 1.0 where a bound exists: lower 12 (u 6 | q 6), upper 12 (stage rules: has_comp)
    alignas(16) double red[8][NWV_MAX];   // one partial per wavefront (Ex::put_* / get_*)
    alignas(16) double xhat[12];    // current plant state (feedback, simulator.py:206)
    alignas(16) double u0[6];
    alignas(16) double logv[48];    // [0..20] pose, rpy, J qdot | [24..35] next plant state | [36..42] task errors
    // log columns wait here until LOGB of them leave as contiguous runs (one 64-byte run per row and flush
    // instead of one 8-byte store per row and step)
    alignas(16) double logbuf[LOG_ROWS][LOGB];
};

// Chunk pool sizes (doubles).  The widest pass needs ~500 doubles per stage (+1 halo stage).
constexpr int POOL_MIN_DOUBLES = 2048;
constexpr int POOL_DEFAULT_DOUBLES = 19456;  // 152 KiB cap: one simulation per CU (batch <= 256 per GPU) gets all the LDS left beside Smem

}  // namespace mpcb
