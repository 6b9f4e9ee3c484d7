// mpc_core.h -- the batched MPC rollout engine, single source for the gfx950 kernel.
//
// One wavefront = one closed-loop simulation (Simulator.run, simulator.py:199-241).
// The code is written as bulk-synchronous phases `ex.par([&](int lane){...})`: inside a
// phase a lane only reads data produced by earlier phases (LDS or the HBM workspace) and
// writes entries no other lane reads in that phase; `par` ends with a wavefront-scope fence.
// On the GPU `Ex` is DevExec (mpc_kernel.hip: lane = threadIdx.x, fence = compiler-only
// because a single wave executes its LDS/VMEM instructions in order).  tests/emu
// instantiates the same template with a host executor that loops over the 64 lanes -- a
// debugging aid for a container without a GPU, never part of the product library.
//
// Algorithm (what acados + HPIPM do behind trajectory_optimizer.py:183-186):
//   SQP_RTI / SQP with Gauss-Newton Hessian  ->  OCP-QP in delta form  ->  Mehrotra
//   predictor-corrector interior point  ->  Riccati recursion for every Newton system.
// Structure exploited here (the oracle does none of this): A = [[I,D1],[0,D2]],
// B = [[E1],[E2]] with diagonal blocks (prediction_model.py:104-112), so B'MB, B'MA, A'MA
// are row/column scalings of the 12x12 cost-to-go M; H_uu, H_uv are diagonal; the only dense
// coupling is the rank-5 task term 50*dt*G'G.  The 6x6 R~ is factorised redundantly by all
// lanes (LDL'), the 18 right-hand sides (12 columns of S~, 6 of I) are solved one per lane.
//
// Memory: the sequential sweeps never wait on HBM inside the recursion.  Each stage's inputs
// live in ONE contiguous record (RIC / FAC / PM, mpc_layout.h); a sweep loads the record of
// stage k-1 into registers while stage k+1 computes (coalesced 8 B/lane loads) and drops it
// into an LDS staging buffer one stage before it is consumed.
#pragma once
#include "mpc_kin.h"

namespace mpcb {

constexpr double BOUND_INF = 1e29;

#ifdef MPCB_PROFILE
#define PROF_T0(v) const double v = ex.clock()
#define PROF_ADD(i, v) prof[i] += ex.clock() - v
#else
#define PROF_T0(v)
#define PROF_ADD(i, v)
#endif
enum { PF_LIN = 0, PF_NRES, PF_INIT, PF_RES, PF_FACT, PF_BWD, PF_FWD, PF_STEP, PF_MUAFF, PF_CORR, PF_UPD, PF_NUPD,
       PF_PLANT, PF_TOTAL, PF_COUNT_IPM };

struct Ctx {
    const Problem *pb;
    Ws w;
    Smem *sm;
    int N;
};

struct PfRegs {
    double d[5];
};

// ---- bound bookkeeping (trajectory_optimizer.py:164-171: lbu on stages 0..N-1, lbx on
// q of stages 1..N-1; x_0 is fixed by lbx_0 = ubx_0, simulator.py:210-211) -------------
MPC_HD bool has_comp(int N, int k, int j) { return j < 6 ? (k < N) : (k >= 1 && k < N); }
MPC_HD double bnd_lo(const InstParams &P, int j) { return j < 6 ? P.umin[j] : P.qmin[j - 6]; }
MPC_HD double bnd_hi(const InstParams &P, int j) { return j < 6 ? P.umax[j] : P.qmax[j - 6]; }
MPC_HD double cur_val(const Ws &w, int k, int j) { return j < 6 ? w.U[k * W_U + j] : w.X[k * W_X + (j - 6)]; }

// element e of the concatenation of three HBM segments (0 beyond the end)
MPC_HD double seg_load(int e, const double *p0, int n0, const double *p1, int n1, const double *p2, int n2)
{
    const double *p = e < n0 ? p0 + e : (e < n0 + n1 ? p1 + (e - n0) : p2 + (e - n0 - n1));
    return e < n0 + n1 + n2 ? *p : 0.0;
}

// lower-triangle index e -> (i, j), i >= j
MPC_HD void tri_index(int e, int &i, int &j)
{
    i = 0;
    while (e > i) { e -= i + 1; i++; }
    j = e;
}

// Copy the instance parameters and the kinematic constants into LDS (once per launch).
template <class Ex>
MPC_HD void load_constants(Ex &ex, Smem &sm, const InstParams *P, const Robot *rb)
{
    ex.par([&](int lane) {
        const double *ps = reinterpret_cast<const double *>(P);
        double *pd = reinterpret_cast<double *>(&sm.P);
        for (int e = lane; e < (int)(sizeof(InstParams) / sizeof(double)); e += WAVE) pd[e] = ps[e];
        const double *rs = reinterpret_cast<const double *>(rb);
        double *rd = reinterpret_cast<double *>(&sm.rb);
        for (int e = lane; e < (int)(sizeof(Robot) / sizeof(double)); e += WAVE) rd[e] = rs[e];
    });
}

template <class Ex>
struct Engine {
    Ex &ex;
    Ctx c;
    int N;
    double lin_cost;  // cost of the linearisation currently held in LIN/RIC
    typename Ex::template PerLane<PfRegs> pf;
#ifdef MPCB_PROFILE
    double prof[NPROF];
#endif

    MPC_HD Engine(Ex &e, const Ctx &cc) : ex(e), c(cc), N(cc.N), lin_cost(0.0)
    {
#ifdef MPCB_PROFILE
        for (int i = 0; i < NPROF; i++) prof[i] = 0.0;
#endif
    }

    // =========================================================================== NLP level
    // Linearise at the iterate (X,U): task residual + Jacobian per stage, dynamics defect,
    // cost = sum_k dt/2 r'Wr (acados get_cost(), simulator.py:221).  Lane <-> stage.
    MPC_HD double linearize(const double *X, const double *U, bool jac)
    {
        PROF_T0(t0);
        Smem &sm = *c.sm;
        const InstParams &P = sm.P;
        const Robot &rb = sm.rb;
        Ws &w = c.w;
        const int Nl = N;
        ex.par([&](int lane) {
            double csum = 0.0;
            for (int k = lane; k <= Nl; k += WAVE) {
                double *lin = w.LIN + (size_t)k * W_LIN;
                double *ric = w.RIC + (size_t)k * W_RIC;
                if (k < Nl) {
                    const double *x = X + (size_t)k * W_X, *u = U + (size_t)k * W_U, *xn = X + (size_t)(k + 1) * W_X;
                    double xx[12], uu[6];
#pragma unroll
                    for (int i = 0; i < 12; i++) xx[i] = x[i];
#pragma unroll
                    for (int i = 0; i < 6; i++) uu[i] = u[i];
                    if (jac) task_lin<true>(rb, P, xx, xx + 6, lin, ric);
                    else task_lin<false>(rb, P, xx, xx + 6, lin, ric);
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < NTASK; i++) s += P.w_task[i] * lin[LIN_R + i] * lin[LIN_R + i];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        const double uj = uu[j], vj = xx[6 + j];
                        const double qdd = P.cq[j] * (uj - vj);  // prediction_model.py:326
                        s += 2.0 * P.w_u * uj * uj + P.w_qddot * qdd * qdd;
                        if (jac) {
                            w.BD[(size_t)k * W_BD + j] = (xx[j] + P.a12[j] * vj + P.b1[j] * uj) - xn[j];
                            w.BD[(size_t)k * W_BD + 6 + j] = (P.a22[j] * vj + P.b2[j] * uj) - xn[6 + j];
                        }
                    }
                    csum += 0.5 * P.dt * s;
                } else if (jac) {
#pragma unroll
                    for (int i = 0; i < W_LIN; i++) lin[i] = 0.0;
#pragma unroll
                    for (int i = 0; i < RIC_GAM; i++) ric[i] = 0.0;
                }
            }
            sm.red[0][lane] = csum;
        });
        const double r = ex.reduce_sum(sm.red[0]);
        PROF_ADD(PF_LIN, t0);
        return r;
    }

    // y_ki = w_i (r_ki + G_ki . delta_k): weighted (linearised) task residual.  Flat (k,i).
    MPC_HD void phase_y(const double *QW)
    {
        const InstParams &P = c.sm->P;
        Ws &w = c.w;
        const int total = N * NTASK;
        ex.par([&](int lane) {
            for (int e = lane; e < total; e += WAVE) {
                const int k = e / NTASK, i = e - k * NTASK;
                double *lin = w.LIN + (size_t)k * W_LIN;
                const double *ric = w.RIC + (size_t)k * W_RIC;
                double v = lin[LIN_R + i];
                if (QW) {
                    const double *dw = QW + (size_t)k * W_QW;
#pragma unroll
                    for (int j = 0; j < 6; j++) v += ric[RIC_GQ + i * 6 + j] * dw[6 + j];
                    if (i == 4) {
#pragma unroll
                        for (int j = 0; j < 6; j++) v += ric[RIC_GV + j] * dw[12 + j];
                    }
                }
                lin[LIN_Y + i] = P.w_task[i] * v;
            }
        });
    }

    // Stationarity element (k,c) of the Lagrangian: cost gradient (+ GN Hessian * delta),
    // dynamics adjoints; bound multipliers are added by the caller.
    MPC_HD double stat_elem(int k, int cidx, const double *QW, const double *PI) const
    {
        const InstParams &P = c.sm->P;
        const Ws &w = c.w;
        double val = 0.0;
        const double *pk = PI + (size_t)k * W_PI;
        const double *pm = PI + (size_t)(k > 0 ? k - 1 : 0) * W_PI;
        if (cidx < 6) {
            if (k >= N) return 0.0;
            const int j = cidx;
            double uj = w.U[k * W_U + j], vj = w.X[k * W_X + 6 + j];
            if (QW) { uj += QW[k * W_QW + j]; vj += QW[k * W_QW + 12 + j]; }
            const double c2 = P.w_qddot * P.cq[j] * P.cq[j];
            val = P.dt * (2.0 * P.w_u * uj + c2 * (uj - vj));
            val += P.b1[j] * pk[j] + P.b2[j] * pk[6 + j];
        } else if (cidx < 12) {
            if (k == 0) return 0.0;
            const int j = cidx - 6;
            if (k < N) {
                const double *lin = w.LIN + (size_t)k * W_LIN;
                const double *ric = w.RIC + (size_t)k * W_RIC;
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < NTASK; i++) s += ric[RIC_GQ + i * 6 + j] * lin[LIN_Y + i];
                val = P.dt * s + pk[j];
            }
            val -= pm[j];
        } else {
            if (k == 0) return 0.0;
            const int j = cidx - 12;
            if (k < N) {
                const double *lin = w.LIN + (size_t)k * W_LIN;
                const double *ric = w.RIC + (size_t)k * W_RIC;
                double uj = w.U[k * W_U + j], vj = w.X[k * W_X + 6 + j];
                if (QW) { uj += QW[k * W_QW + j]; vj += QW[k * W_QW + 12 + j]; }
                const double c2 = P.w_qddot * P.cq[j] * P.cq[j];
                val = P.dt * (ric[RIC_GV + j] * lin[LIN_Y + 4] + c2 * (vj - uj));
                val += P.a12[j] * pk[j] + P.a22[j] * pk[6 + j];
            }
            val -= pm[6 + j];
        }
        return val;
    }

    // acados ocp_nlp_res_compute: inf-norms [stat, eq, ineq, comp] at the NLP iterate with
    // multipliers (PI, LAM, T).  Needs a fresh linearisation (LIN, RIC, BD).
    MPC_HD void nlp_residuals(const double *PI, const double *LAM, const double *T, double *res4)
    {
        PROF_T0(t0);
        Smem &sm = *c.sm;
        const InstParams &P = sm.P;
        Ws &w = c.w;
        phase_y(nullptr);
        const int tot_g = (N + 1) * NW, tot_b = N * NX, tot_c = (N + 1) * NB;
        ex.par([&](int lane) {
            double rs = 0, re = 0, ri = 0, rc = 0;
            for (int e = lane; e < tot_g; e += WAVE) {
                const int k = e / NW, ci = e - k * NW;
                double v = stat_elem(k, ci, nullptr, PI);
                if (ci < NB && has_comp(N, k, ci)) {
                    if (bnd_lo(P, ci) > -BOUND_INF) v -= LAM[k * W_LAM + ci];
                    if (bnd_hi(P, ci) < BOUND_INF) v += LAM[k * W_LAM + 12 + ci];
                }
                if (ci >= 6 && k == 0) v = 0.0;
                rs = fmax(rs, fabs(v));
            }
            for (int e = lane; e < tot_b; e += WAVE) re = fmax(re, fabs(w.BD[e]));
            for (int e = lane; e < tot_c; e += WAVE) {
                const int k = e / NB, j = e - k * NB;
                if (!has_comp(N, k, j)) continue;
                const double v = cur_val(w, k, j);
                if (bnd_lo(P, j) > -BOUND_INF) {
                    const double l = LAM[k * W_LAM + j], t = T[k * W_T + j];
                    ri = fmax(ri, fabs((bnd_lo(P, j) - v) + t));
                    rc = fmax(rc, fabs(l * t));
                }
                if (bnd_hi(P, j) < BOUND_INF) {
                    const double l = LAM[k * W_LAM + 12 + j], t = T[k * W_T + 12 + j];
                    ri = fmax(ri, fabs((v - bnd_hi(P, j)) + t));
                    rc = fmax(rc, fabs(l * t));
                }
            }
            if (lane < NX) ri = fmax(ri, fabs(sm.xhat[lane] - w.X[lane]));  // lbx_0 = ubx_0 = x_hat
            sm.red[0][lane] = rs; sm.red[1][lane] = re; sm.red[2][lane] = ri; sm.red[3][lane] = rc;
        });
        res4[0] = ex.reduce_max(sm.red[0]);
        res4[1] = ex.reduce_max(sm.red[1]);
        res4[2] = ex.reduce_max(sm.red[2]);
        res4[3] = ex.reduce_max(sm.red[3]);
        PROF_ADD(PF_NRES, t0);
    }

    // =========================================================================== IPM pieces
    // HPIPM init with warm_start = 2: keep (w, pi, lam, t) of the previous QP, clamp
    // lam, t >= 0.1; embed x0.  Returns the number of active (finite) bound sides.
    MPC_HD double ipm_init()
    {
        PROF_T0(t0);
        Smem &sm = *c.sm;
        const InstParams &P = sm.P;
        Ws &w = c.w;
        const int tot = (N + 1) * NB;
        ex.par([&](int lane) {
            double nc = 0.0;
            for (int e = lane; e < tot; e += WAVE) {
                const int k = e / NB, j = e - k * NB;
                const bool hc = has_comp(N, k, j);
                const bool lo = hc && bnd_lo(P, j) > -BOUND_INF, hi = hc && bnd_hi(P, j) < BOUND_INF;
                double *lam = w.QLAM + (size_t)k * W_LAM, *t = w.QT + (size_t)k * W_T;
                if (lo) { lam[j] = fmax(lam[j], 0.1); t[j] = fmax(t[j], 0.1); nc += 1.0; }
                else { lam[j] = 0.0; t[j] = 1.0; }
                if (hi) { lam[12 + j] = fmax(lam[12 + j], 0.1); t[12 + j] = fmax(t[12 + j], 0.1); nc += 1.0; }
                else { lam[12 + j] = 0.0; t[12 + j] = 1.0; }
            }
            if (lane < NX) w.QW[6 + lane] = sm.xhat[lane] - w.X[lane];
            if (lane < NU) w.QW[(size_t)N * W_QW + lane] = 0.0;
            sm.red[0][lane] = nc;
        });
        const double r = ex.reduce_sum(sm.red[0]);
        PROF_ADD(PF_INIT, t0);
        return r;
    }

    // QP residuals at (QW, QPI, QLAM, QT); also Gamma and the condensed gradient gt of the
    // Newton system (HPIPM compute_Gamma_gamma).  nrm = [g, b, d, m], returns sum(lam*t).
    MPC_HD double ipm_residuals(double *nrm)
    {
        PROF_T0(t0);
        Smem &sm = *c.sm;
        const InstParams &P = sm.P;
        Ws &w = c.w;
        phase_y(w.QW);
        const int tot_g = (N + 1) * NW, tot_b = N * NX;
        ex.par([&](int lane) {
            double ng = 0, nb = 0, nd = 0, nm = 0, smu = 0;
            for (int e = lane; e < tot_g; e += WAVE) {
                const int k = e / NW, ci = e - k * NW;
                double *ric = w.RIC + (size_t)k * W_RIC;
                double rg = stat_elem(k, ci, w.QW, w.QPI);
                double gt = rg;
                if (ci < NB) {
                    const bool hc = has_comp(N, k, ci);
                    const bool lo = hc && bnd_lo(P, ci) > -BOUND_INF, hi = hc && bnd_hi(P, ci) < BOUND_INF;
                    const double v = hc ? cur_val(w, k, ci) : 0.0, dv = w.QW[(size_t)k * W_QW + ci];
                    double gam = 0.0, rdl = 0, rml = 0, rdu = 0, rmu = 0;
                    if (lo) {
                        const double l = w.QLAM[k * W_LAM + ci], t = w.QT[k * W_T + ci];
                        rdl = dv - (bnd_lo(P, ci) - v) - t;
                        rml = l * t;
                        rg -= l;
                        gam += l / t;
                        smu += rml;
                        nd = fmax(nd, fabs(rdl)); nm = fmax(nm, fabs(rml));
                    }
                    if (hi) {
                        const double l = w.QLAM[k * W_LAM + 12 + ci], t = w.QT[k * W_T + 12 + ci];
                        rdu = (bnd_hi(P, ci) - v) - dv - t;
                        rmu = l * t;
                        rg += l;
                        gam += l / t;
                        smu += rmu;
                        nd = fmax(nd, fabs(rdu)); nm = fmax(nm, fabs(rmu));
                    }
                    gt = rg;
                    if (lo) gt += (rml + w.QLAM[k * W_LAM + ci] * rdl) / w.QT[k * W_T + ci];
                    if (hi) gt -= (rmu + w.QLAM[k * W_LAM + 12 + ci] * rdu) / w.QT[k * W_T + 12 + ci];
                    w.RD[k * W_RD + ci] = rdl; w.RD[k * W_RD + 12 + ci] = rdu;
                    w.RM[k * W_RM + ci] = rml; w.RM[k * W_RM + 12 + ci] = rmu;
                    ric[RIC_GAM + ci] = gam;
                }
                w.RG[e] = rg;
                ric[RIC_GT + ci] = gt;
                ng = fmax(ng, fabs(rg));
            }
            for (int e = lane; e < tot_b; e += WAVE) {
                const int k = e / NX, i = e - k * NX;
                const double *dw = w.QW + (size_t)k * W_QW, *dn = w.QW + (size_t)(k + 1) * W_QW;
                double v;
                if (i < 6) v = dw[6 + i] + P.a12[i] * dw[12 + i] + P.b1[i] * dw[i];
                else v = P.a22[i - 6] * dw[6 + i] + P.b2[i - 6] * dw[i - 6];
                v += w.BD[e] - dn[6 + i];
                w.RIC[(size_t)k * W_RIC + RIC_RB + i] = v;
                nb = fmax(nb, fabs(v));
            }
            sm.red[0][lane] = ng; sm.red[1][lane] = nb; sm.red[2][lane] = nd; sm.red[3][lane] = nm;
            sm.red[4][lane] = smu;
        });
        nrm[0] = ex.reduce_max(sm.red[0]);
        nrm[1] = ex.reduce_max(sm.red[1]);
        nrm[2] = ex.reduce_max(sm.red[2]);
        nrm[3] = ex.reduce_max(sm.red[3]);
        const double r = ex.reduce_sum(sm.red[4]);
        PROF_ADD(PF_RES, t0);
        return r;
    }

    // Centering-corrector right-hand side (HPIPM compute_centering_correction):
    // rm <- lam*t + dlam_aff*dt_aff - sigma*mu ; rebuild gt.
    MPC_HD void ipm_corrector_rhs(double sigma_mu)
    {
        PROF_T0(t0);
        const InstParams &P = c.sm->P;
        Ws &w = c.w;
        const int tot = (N + 1) * NB;
        ex.par([&](int lane) {
            for (int e = lane; e < tot; e += WAVE) {
                const int k = e / NB, j = e - k * NB;
                if (!has_comp(N, k, j)) continue;
                double gt = w.RG[k * W_RG + j];
                if (bnd_lo(P, j) > -BOUND_INF) {
                    const double l = w.QLAM[k * W_LAM + j], t = w.QT[k * W_T + j];
                    const double rm = l * t + w.DLAM[k * W_DLAM + j] * w.DT[k * W_DT + j] - sigma_mu;
                    w.RM[k * W_RM + j] = rm;
                    gt += (rm + l * w.RD[k * W_RD + j]) / t;
                }
                if (bnd_hi(P, j) < BOUND_INF) {
                    const double l = w.QLAM[k * W_LAM + 12 + j], t = w.QT[k * W_T + 12 + j];
                    const double rm = l * t + w.DLAM[k * W_DLAM + 12 + j] * w.DT[k * W_DT + 12 + j] - sigma_mu;
                    w.RM[k * W_RM + 12 + j] = rm;
                    gt -= (rm + l * w.RD[k * W_RD + 12 + j]) / t;
                }
                w.RIC[(size_t)k * W_RIC + RIC_GT + j] = gt;
            }
        });
        PROF_ADD(PF_CORR, t0);
    }

    // dt, dlam from the primal step (HPIPM compute_lam_t) and the largest feasible step.
    MPC_HD double ipm_step_lam_t()
    {
        PROF_T0(t0);
        Smem &sm = *c.sm;
        const InstParams &P = sm.P;
        Ws &w = c.w;
        const int tot = (N + 1) * NB;
        ex.par([&](int lane) {
            double alpha = 1.0;
            for (int e = lane; e < tot; e += WAVE) {
                const int k = e / NB, j = e - k * NB;
                const bool hc = has_comp(N, k, j);
                const double dv = w.DW[(size_t)k * W_DW + j];
                double dtl = 0, dll = 0, dtu = 0, dlu = 0;
                if (hc && bnd_lo(P, j) > -BOUND_INF) {
                    const double l = w.QLAM[k * W_LAM + j], t = w.QT[k * W_T + j];
                    dtl = dv + w.RD[k * W_RD + j];
                    dll = -(w.RM[k * W_RM + j] + l * dtl) / t;
                    if (dll < 0 && l + alpha * dll < 0) alpha = -l / dll;
                    if (dtl < 0 && t + alpha * dtl < 0) alpha = -t / dtl;
                }
                if (hc && bnd_hi(P, j) < BOUND_INF) {
                    const double l = w.QLAM[k * W_LAM + 12 + j], t = w.QT[k * W_T + 12 + j];
                    dtu = -dv + w.RD[k * W_RD + 12 + j];
                    dlu = -(w.RM[k * W_RM + 12 + j] + l * dtu) / t;
                    if (dlu < 0 && l + alpha * dlu < 0) alpha = -l / dlu;
                    if (dtu < 0 && t + alpha * dtu < 0) alpha = -t / dtu;
                }
                w.DT[k * W_DT + j] = dtl; w.DLAM[k * W_DLAM + j] = dll;
                w.DT[k * W_DT + 12 + j] = dtu; w.DLAM[k * W_DLAM + 12 + j] = dlu;
            }
            sm.red[0][lane] = alpha;
        });
        const double r = ex.reduce_min(sm.red[0]);
        PROF_ADD(PF_STEP, t0);
        return r;
    }

    MPC_HD double ipm_mu_aff(double alpha)
    {
        PROF_T0(t0);
        Ws &w = c.w;
        Smem &sm = *c.sm;
        const int tot = (N + 1) * NL;
        ex.par([&](int lane) {
            double s = 0.0;
            for (int e = lane; e < tot; e += WAVE) {
                // masked-out sides carry lam = 0, dlam = 0 -> contribute 0
                s += (w.QLAM[e] + alpha * w.DLAM[e]) * (w.QT[e] + alpha * w.DT[e]);
            }
            sm.red[0][lane] = s;
        });
        const double r = ex.reduce_sum(sm.red[0]);
        PROF_ADD(PF_MUAFF, t0);
        return r;
    }

    MPC_HD void ipm_update(double a)
    {
        PROF_T0(t0);
        const InstParams &P = c.sm->P;
        Ws &w = c.w;
        const int tw = (N + 1) * NW, tp = N * NX, tc = (N + 1) * NB;
        ex.par([&](int lane) {
            for (int e = lane; e < tw; e += WAVE) w.QW[e] += a * w.DW[e];
            for (int e = lane; e < tp; e += WAVE) w.QPI[e] += a * w.DPI[e];
            for (int e = lane; e < tc; e += WAVE) {
                const int k = e / NB, j = e - k * NB;
                if (!has_comp(N, k, j)) continue;
                if (bnd_lo(P, j) > -BOUND_INF) {
                    w.QLAM[k * W_LAM + j] = fmax(w.QLAM[k * W_LAM + j] + a * w.DLAM[k * W_DLAM + j], 1e-16);
                    w.QT[k * W_T + j] = fmax(w.QT[k * W_T + j] + a * w.DT[k * W_DT + j], 1e-16);
                }
                if (bnd_hi(P, j) < BOUND_INF) {
                    w.QLAM[k * W_LAM + 12 + j] = fmax(w.QLAM[k * W_LAM + 12 + j] + a * w.DLAM[k * W_DLAM + 12 + j], 1e-16);
                    w.QT[k * W_T + 12 + j] = fmax(w.QT[k * W_T + 12 + j] + a * w.DT[k * W_DT + 12 + j], 1e-16);
                }
            }
        });
        PROF_ADD(PF_UPD, t0);
    }

    // =========================================================================== Riccati
    static constexpr int STG_HALF = STG_DOUBLES / 2;
    // staging offsets of the solve sweeps
    static constexpr int SB_PM = 0, SB_K = 144, SB_GT = 216, SB_RB = 234, SB_TOT = 246;      // backward solve
    static constexpr int SF_FAC = 0, SF_PM = 126, SF_RB = 270, SF_TOT = 282;                  // forward

    // Backward sweep.  FACT: rebuild (R~, S~, P) per stage from Gamma and the Jacobians and
    // write Kfb = R~^-1 S~, R~^-1, P_k; always propagates the vector part (gt, rb) -> p_k, h_u.
    template <bool FACT>
    MPC_HD void riccati_backward()
    {
        PROF_T0(t0);
        Smem &sm = *c.sm;
        const InstParams &P = sm.P;
        Ws &w = c.w;
        const int Nl = N;
        // fetch of the stage-k record into registers (coalesced), commit into LDS staging
        auto issue = [&](int lane, int k) {
            PfRegs &r = pf.at(lane);
            if (FACT) {
                const double *ric = w.RIC + (size_t)k * W_RIC;
                r.d[0] = ric[lane];
                r.d[1] = lane < W_RIC - WAVE ? ric[WAVE + lane] : 0.0;
            } else {
                const double *p0 = w.PM + (size_t)(k + 1) * W_PM, *p1 = w.FAC + (size_t)k * W_FAC + FAC_K,
                             *p2 = w.RIC + (size_t)k * W_RIC + RIC_GT;
#pragma unroll
                for (int i = 0; i < 4; i++) r.d[i] = seg_load(lane + WAVE * i, p0, 144, p1, 72, p2, 30);
            }
        };
        auto commit = [&](int lane, int buf) {
            const PfRegs &r = pf.at(lane);
            double *s = sm.stg + buf * STG_HALF;
            if (FACT) {
                s[lane] = r.d[0];
                if (lane < W_RIC - WAVE) s[WAVE + lane] = r.d[1];
            } else {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (lane + WAVE * i < SB_TOT) s[lane + WAVE * i] = r.d[i];
            }
        };
        // terminal stage: no cost, no bounds -> P_N = 0, p_N = gt_N,x ; first stage record
        ex.par([&](int lane) {
            if (FACT) {
                for (int e = lane; e < 144; e += WAVE) { sm.M[0][e] = 0.0; w.PM[(size_t)Nl * W_PM + e] = 0.0; }
            }
            if (lane < NX) {
                const double v = w.RIC[(size_t)Nl * W_RIC + RIC_GT + 6 + lane];
                sm.pv[0][lane] = v;
                w.FAC[(size_t)Nl * W_FAC + FAC_PV + lane] = v;
            }
            issue(lane, Nl - 1);
        });
        ex.par([&](int lane) {
            commit(lane, 0);
            if (Nl >= 2) issue(lane, Nl - 2);
        });
        int cur = 0, sb = 0;
        for (int k = Nl - 1; k >= 0; k--) {
            const double *stg = sm.stg + sb * STG_HALF;
            double *fac = w.FAC + (size_t)k * W_FAC;
            const int nxt = cur ^ 1;
            const double *gam = stg + RIC_GAM;
            const double *gt = FACT ? stg + RIC_GT : stg + SB_GT;
            const double *rb = FACT ? stg + RIC_RB : stg + SB_RB;
            // ---- F0: R~ (21 lower entries), S~ (72), m~ = p_{k+1} + P_{k+1} rb_k (12)
            ex.par([&](int lane) {
                const double *M = FACT ? sm.M[cur] : stg + SB_PM;
                if (FACT) {
                    if (lane < 21) {
                        int i, j;
                        tri_index(lane, i, j);
                        double r = P.b1[i] * P.b1[j] * M[i * 12 + j] + P.b2[i] * P.b1[j] * M[(6 + i) * 12 + j] +
                                   P.b1[i] * P.b2[j] * M[i * 12 + 6 + j] + P.b2[i] * P.b2[j] * M[(6 + i) * 12 + 6 + j];
                        if (i == j) {
                            const double c2 = P.w_qddot * P.cq[i] * P.cq[i];
                            r += P.dt * (2.0 * P.w_u + c2) + gam[i];
                        }
                        sm.Rt[i * 6 + j] = r;
                        sm.Rt[j * 6 + i] = r;
                    }
                    for (int e = lane; e < 72; e += WAVE) {
                        const int m = e / 12, cc = e - m * 12;
                        double s;
                        if (cc < 6) {
                            s = P.b1[m] * M[m * 12 + cc] + P.b2[m] * M[(6 + m) * 12 + cc];
                        } else {
                            const int j = cc - 6;
                            const double fq = P.b1[m] * M[m * 12 + j] + P.b2[m] * M[(6 + m) * 12 + j];
                            const double fv = P.b1[m] * M[m * 12 + 6 + j] + P.b2[m] * M[(6 + m) * 12 + 6 + j];
                            s = fq * P.a12[j] + fv * P.a22[j];
                            if (m == j) s -= P.dt * P.w_qddot * P.cq[j] * P.cq[j];
                        }
                        sm.St[e] = s;
                    }
                }
                if (lane < NX) {
                    double s = sm.pv[cur][lane];
#pragma unroll
                    for (int j = 0; j < NX; j++) s += M[lane * 12 + j] * rb[j];
                    sm.mt[lane] = s;
                }
            });
            // ---- F1: LDL' of R~ (redundant), one right-hand side per lane; vector part;
            //          stage k-1's record moves from registers to the other staging buffer
            ex.par([&](int lane) {
                double hu[6];
#pragma unroll
                for (int j = 0; j < 6; j++) hu[j] = gt[j] + P.b1[j] * sm.mt[j] + P.b2[j] * sm.mt[6 + j];
                if (FACT) {
                    double L[6][6], dd[6], dinv[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        double d = sm.Rt[j * 6 + j];
#pragma unroll
                        for (int r = 0; r < j; r++) d -= L[j][r] * L[j][r] * dd[r];
                        dd[j] = d;
                        dinv[j] = 1.0 / d;
#pragma unroll
                        for (int i = j + 1; i < 6; i++) {
                            double s = sm.Rt[i * 6 + j];
#pragma unroll
                            for (int r = 0; r < j; r++) s -= L[i][r] * L[j][r] * dd[r];
                            L[i][j] = s * dinv[j];
                        }
                    }
                    if (lane < 18) {
                        double x[6];
#pragma unroll
                        for (int i = 0; i < 6; i++)
                            x[i] = lane < 12 ? sm.St[i * 12 + (lane < 12 ? lane : 0)] : (i == lane - 12 ? 1.0 : 0.0);
#pragma unroll
                        for (int i = 0; i < 6; i++) {  // L y = rhs
#pragma unroll
                            for (int r = 0; r < i; r++) x[i] -= L[i][r] * x[r];
                        }
#pragma unroll
                        for (int i = 0; i < 6; i++) x[i] *= dinv[i];
#pragma unroll
                        for (int i = 5; i >= 0; i--) {  // L' x = y
#pragma unroll
                            for (int r = i + 1; r < 6; r++) x[i] -= L[r][i] * x[r];
                        }
                        if (lane < 12) {
#pragma unroll
                            for (int i = 0; i < 6; i++) { fac[FAC_K + i * 12 + lane] = x[i]; sm.Kf[i * 12 + lane] = x[i]; }
                        } else {
#pragma unroll
                            for (int i = 0; i < 6; i++) fac[FAC_RI + i * 6 + (lane - 12)] = x[i];
                        }
                    }
                }
                if (lane < NX) {
                    double hx = gt[6 + lane];
                    if (lane < 6) hx += sm.mt[lane];
                    else hx += P.a12[lane - 6] * sm.mt[lane - 6] + P.a22[lane - 6] * sm.mt[lane];
                    double pj = hx;
                    const double *Kc = FACT ? sm.Kf : stg + SB_K;
#pragma unroll
                    for (int m = 0; m < 6; m++) pj -= Kc[m * 12 + lane] * hu[m];
                    sm.pv[nxt][lane] = pj;
                    fac[FAC_PV + lane] = pj;
                }
                if (lane < 6) {
                    double v = hu[0];
#pragma unroll
                    for (int j = 1; j < 6; j++) v = lane == j ? hu[j] : v;
                    fac[FAC_HU + lane] = v;
                }
                if (k >= 1) {
                    commit(lane, sb ^ 1);
                    if (k >= 2) issue(lane, k - 2);
                }
            });
            // ---- F2: P_k = H_xx + Gamma_q + A'MA - S~' Kfb   (78 unique entries)
            if (FACT && k > 0) {
                ex.par([&](int lane) {
                    const double *M = sm.M[cur];
                    const double *gq = stg + RIC_GQ, *gv = stg + RIC_GV;
                    for (int e = lane; e < 78; e += WAVE) {
                        int i, j;
                        tri_index(e, i, j);
                        double v;
                        if (i < 6) {  // qq
                            v = M[i * 12 + j];
                            double s = 0.0;
#pragma unroll
                            for (int r = 0; r < NTASK; r++) s += P.w_task[r] * gq[r * 6 + i] * gq[r * 6 + j];
                            v += P.dt * s;
                            if (i == j) v += gam[6 + i];
                        } else if (j < 6) {  // vq: row 6+a, col b
                            const int a = i - 6, b = j;
                            v = P.a12[a] * M[a * 12 + b] + P.a22[a] * M[(6 + a) * 12 + b];
                            v += P.dt * P.w_task[4] * gv[a] * gq[4 * 6 + b];
                        } else {  // vv
                            const int a = i - 6, b = j - 6;
                            const double cq = P.a12[a] * M[a * 12 + b] + P.a22[a] * M[(6 + a) * 12 + b];
                            const double cv = P.a12[a] * M[a * 12 + 6 + b] + P.a22[a] * M[(6 + a) * 12 + 6 + b];
                            v = cq * P.a12[b] + cv * P.a22[b];
                            v += P.dt * P.w_task[4] * gv[a] * gv[b];
                            if (a == b) v += P.dt * P.w_qddot * P.cq[a] * P.cq[a];
                        }
#pragma unroll
                        for (int m = 0; m < 6; m++) v -= sm.St[m * 12 + i] * sm.Kf[m * 12 + j];
                        sm.M[nxt][i * 12 + j] = v;
                        sm.M[nxt][j * 12 + i] = v;
                        w.PM[(size_t)k * W_PM + i * 12 + j] = v;
                        w.PM[(size_t)k * W_PM + j * 12 + i] = v;
                    }
                });
            }
            cur = nxt;
            sb ^= 1;
        }
        PROF_ADD(FACT ? PF_FACT : PF_BWD, t0);
    }

    // Forward sweep: du = -Kfb dx - Rinv h_u ; dx+ = A dx + B du + rb ; dpi = P dx+ + p.
    MPC_HD void riccati_forward()
    {
        PROF_T0(t0);
        Smem &sm = *c.sm;
        const InstParams &P = sm.P;
        Ws &w = c.w;
        const int Nl = N;
        auto issue = [&](int lane, int k) {
            PfRegs &r = pf.at(lane);
            const double *p0 = w.FAC + (size_t)k * W_FAC, *p1 = w.PM + (size_t)k * W_PM,
                         *p2 = w.RIC + (size_t)k * W_RIC + RIC_RB;
#pragma unroll
            for (int i = 0; i < 5; i++) r.d[i] = seg_load(lane + WAVE * i, p0, 126, p1, 144, p2, 12);
        };
        auto commit = [&](int lane, int buf) {
            const PfRegs &r = pf.at(lane);
            double *s = sm.stg + buf * STG_HALF;
#pragma unroll
            for (int i = 0; i < 5; i++)
                if (lane + WAVE * i < SF_TOT) s[lane + WAVE * i] = r.d[i];
        };
        ex.par([&](int lane) {
            if (lane < NX) sm.dx[0][lane] = 0.0;  // dx_0 = 0: x_0 is pinned by ipm_init
            issue(lane, 0);
        });
        ex.par([&](int lane) {
            commit(lane, 0);
            issue(lane, 1);  // N >= 1
        });
        int cur = 0, sb = 0;
        for (int k = 0; k <= Nl; k++) {
            const int nxt = cur ^ 1;
            const double *stg = sm.stg + sb * STG_HALF;
            // W0: lanes 0..5 -> du_k ; lanes 6..17 -> dpi_{k-1} = P_k dx_k + p_k ; lanes 18..29 log dx_k
            ex.par([&](int lane) {
                if (lane < 18) {
                    const bool isu = lane < 6;
                    if ((isu && k < Nl) || (!isu && k >= 1)) {
                        const double *row = isu ? stg + SF_FAC + FAC_K + lane * 12 : stg + SF_PM + (lane - 6) * 12;
                        double s = 0.0;
#pragma unroll
                        for (int j = 0; j < NX; j++) s += row[j] * sm.dx[cur][j];
                        if (isu) {
#pragma unroll
                            for (int m = 0; m < 6; m++) s += stg[SF_FAC + FAC_RI + lane * 6 + m] * stg[SF_FAC + FAC_HU + m];
                            s = -s;
                            sm.du[lane] = s;
                            w.DW[(size_t)k * W_DW + lane] = s;
                        } else {
                            w.DPI[(size_t)(k - 1) * W_DPI + (lane - 6)] = s + stg[SF_FAC + FAC_PV + (lane - 6)];
                        }
                    } else if (isu) {
                        w.DW[(size_t)k * W_DW + lane] = 0.0;  // stage N has no input
                    }
                } else if (lane < 30) {
                    w.DW[(size_t)k * W_DW + 6 + (lane - 18)] = sm.dx[cur][lane - 18];
                }
            });
            if (k == Nl) break;
            // W1: dx_{k+1}; next stage's record -> other staging buffer
            ex.par([&](int lane) {
                if (lane < NX) {
                    double v;
                    if (lane < 6) v = sm.dx[cur][lane] + P.a12[lane] * sm.dx[cur][6 + lane] + P.b1[lane] * sm.du[lane];
                    else v = P.a22[lane - 6] * sm.dx[cur][lane] + P.b2[lane - 6] * sm.du[lane - 6];
                    sm.dx[nxt][lane] = v + stg[SF_RB + lane];
                }
                commit(lane, sb ^ 1);
                if (k + 2 <= Nl) issue(lane, k + 2);
            });
            cur = nxt;
            sb ^= 1;
        }
        PROF_ADD(PF_FWD, t0);
    }

    // =========================================================================== IPM driver
    // Restates HPIPM's d_ocp_qp_ipm_solve main loop (see oracle/mpc_oracle.c ipm_solve).
    // Returns HPIPM status 0 ok / 1 max-iter / 2 min-step / 3 NaN.
    MPC_HD int ipm_solve(int *iters_out)
    {
        const double tol = c.sm->P.qp_tol;
        const double nc = ipm_init();
        double nrm[4];
        double mu = ipm_residuals(nrm);
        if (nc > 0) mu /= nc;
        int it = 0, status = 1;
        double alpha = 1.0;
        for (;; it++) {
            if (nrm[0] != nrm[0] || nrm[1] != nrm[1] || nrm[2] != nrm[2] || nrm[3] != nrm[3]) { status = 3; break; }
            if (!(nrm[0] > tol || nrm[1] > tol || nrm[2] > tol || nrm[3] > tol)) { status = 0; break; }
            if (it >= c.pb->qp_iter_max) { status = 1; break; }
            if (!(alpha > 1e-12)) { status = 2; break; }
            riccati_backward<true>();
            riccati_forward();
            const double a_aff = ipm_step_lam_t();
            if (nc > 0) {
                const double mu_aff = ipm_mu_aff(a_aff) / nc;
                const double tmp = mu_aff / mu;
                const double sigma = tmp * tmp * tmp;
                ipm_corrector_rhs(sigma * mu);
                riccati_backward<false>();
                riccati_forward();
                alpha = ipm_step_lam_t();
            } else {
                alpha = a_aff;
            }
            const double a = alpha * ((1.0 - alpha) * 0.99 + alpha * 0.9999999);
            ipm_update(a);
            mu = ipm_residuals(nrm);
            if (nc > 0) mu /= nc;
        }
#ifdef MPCB_PROFILE
        prof[PF_COUNT_IPM] += it;
#endif
        *iters_out = it;
        return status;
    }

    // =========================================================================== SQP / RTI
    // x += alpha dx etc. (acados ocp_nlp_update_variables_sqp); multipliers blend for SQP.
    MPC_HD void nlp_update(double alpha, bool blend_mult)
    {
        PROF_T0(t0);
        Ws &w = c.w;
        const int tx = (N + 1) * NX, tu = N * NU, tp = N * NX, tl = (N + 1) * NL;
        ex.par([&](int lane) {
            for (int e = lane; e < tx; e += WAVE) {
                const int k = e / NX, i = e - k * NX;
                w.X[e] += alpha * w.QW[(size_t)k * W_QW + 6 + i];
            }
            for (int e = lane; e < tu; e += WAVE) {
                const int k = e / NU, i = e - k * NU;
                w.U[e] += alpha * w.QW[(size_t)k * W_QW + i];
            }
            if (blend_mult) {
                for (int e = lane; e < tp; e += WAVE) w.NPI[e] += alpha * (w.QPI[e] - w.NPI[e]);
                for (int e = lane; e < tl; e += WAVE) {
                    w.NLAM[e] += alpha * (w.QLAM[e] - w.NLAM[e]);
                    w.NT[e] += alpha * (w.QT[e] - w.NT[e]);
                }
            }
        });
        PROF_ADD(PF_NUPD, t0);
    }

    // L1 merit function at (X,U) (acados ocp_nlp_evaluate_merit_fun restated).
    MPC_HD double merit_fun(const double *X, const double *U)
    {
        Smem &sm = *c.sm;
        const InstParams &P = sm.P;
        Ws &w = c.w;
        double m = linearize(X, U, false);
        const int tb = N * NX, tc = (N + 1) * NB;
        ex.par([&](int lane) {
            double s = 0.0;
            for (int e = lane; e < tb; e += WAVE) {
                const int k = e / NX, i = e - k * NX;
                const double *x = X + (size_t)k * W_X, *u = U + (size_t)k * W_U, *xn = X + (size_t)(k + 1) * W_X;
                double v;
                if (i < 6) v = (x[i] + P.a12[i] * x[6 + i] + P.b1[i] * u[i]) - xn[i];
                else v = (P.a22[i - 6] * x[i] + P.b2[i - 6] * u[i - 6]) - xn[i];
                s += w.MW[(size_t)k * W_MW + i] * fabs(v);
            }
            for (int e = lane; e < tc; e += WAVE) {
                const int k = e / NB, j = e - k * NB;
                if (!has_comp(N, k, j)) continue;
                const double v = j < 6 ? U[k * W_U + j] : X[k * W_X + (j - 6)];
                const double vl = bnd_lo(P, j) - v, vu = v - bnd_hi(P, j);
                if (vl > 0) s += w.MW[(size_t)k * W_MW + 12 + j] * vl;
                if (vu > 0) s += w.MW[(size_t)k * W_MW + 24 + j] * vu;
            }
            if (lane < NX) s += w.state[13 + lane] * fabs(sm.xhat[lane] - X[lane]);
            sm.red[1][lane] = s;
        });
        return m + ex.reduce_sum(sm.red[1]);
    }

    // MERIT_BACKTRACKING (trajectory_optimizer.py:68; acados alpha_reduction 0.7, alpha_min 0.05)
    MPC_HD double line_search(int sqp_iter)
    {
        const InstParams &P = c.sm->P;
        Ws &w = c.w;
        const int tb = N * NX, tl = (N + 1) * NL;
        phase_y(w.QW);
        ex.par([&](int lane) {
            for (int e = lane; e < tb; e += WAVE) {
                const int k = e / NX, i = e - k * NX;
                const double a = fabs(w.QPI[e]);
                double *mw = &w.MW[(size_t)k * W_MW + i];
                *mw = sqp_iter == 0 ? a : fmax(a, 0.5 * (*mw + a));
            }
            for (int e = lane; e < tl; e += WAVE) {
                const int k = e / NL, i = e - k * NL;
                const double a = fabs(w.QLAM[e]);
                double *mw = &w.MW[(size_t)k * W_MW + 12 + i];
                *mw = sqp_iter == 0 ? a : fmax(a, 0.5 * (*mw + a));
            }
            if (lane < NX) {
                // multiplier of the eliminated x_0 constraint: stage-0 stationarity of the QP
                const int j = lane;
                double v;
                if (j < 6) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < NTASK; i++) s += w.RIC[RIC_GQ + i * 6 + j] * w.LIN[LIN_Y + i];
                    v = P.dt * s + w.QPI[j];
                } else {
                    const int jj = j - 6;
                    const double uj = w.U[jj] + w.QW[jj], vj = w.X[6 + jj] + w.QW[12 + jj];
                    const double c2 = P.w_qddot * P.cq[jj] * P.cq[jj];
                    v = P.dt * (w.RIC[RIC_GV + jj] * w.LIN[LIN_Y + 4] + c2 * (vj - uj)) + P.a12[jj] * w.QPI[jj] +
                        P.a22[jj] * w.QPI[6 + jj];
                }
                const double a = fabs(v);
                double *mw = &w.state[13 + lane];
                *mw = sqp_iter == 0 ? a : fmax(a, 0.5 * (*mw + a));
            }
        });
        const double m0 = merit_fun(w.X, w.U);
        double alpha = 1.0;
        const int tx = (N + 1) * NX, tu = N * NU;
        while (alpha >= 0.05) {
            ex.par([&](int lane) {
                for (int e = lane; e < tx; e += WAVE) {
                    const int k = e / NX, i = e - k * NX;
                    w.TX[e] = w.X[e] + alpha * w.QW[(size_t)k * W_QW + 6 + i];
                }
                for (int e = lane; e < tu; e += WAVE) {
                    const int k = e / NU, i = e - k * NU;
                    w.TU[e] = w.U[e] + alpha * w.QW[(size_t)k * W_QW + i];
                }
            });
            if (merit_fun(w.TX, w.TU) < m0) break;
            alpha *= 0.7;
        }
        return alpha;
    }

    // One solver.solve() call (simulator.py:210-221).  On entry sm.xhat holds the feedback
    // state and LIN/RIC/BD hold the linearisation at the current iterate when `lin_valid`.
    // On exit they are valid for the (new) iterate again.
    MPC_HD int nlp_step(bool &lin_valid, int *sqp_iter_out, int *qp_iter_out, double *res4, double *cost_out)
    {
        PROF_T0(t0);
        Ws &w = c.w;
        int status = 0, sqp_iter = 0, qp_iter = 0, it = 0;
        double cost = lin_cost;
        if (c.pb->solver_type == 1) {
            // SQP_RTI: one linearisation, one QP, full step
            if (!lin_valid) cost = linearize(w.X, w.U, true);
            const int qs = ipm_solve(&it);
            qp_iter += it;
            sqp_iter = 1;
            if (qs != 0 && qs != 1) status = 4;  // ACADOS_QP_FAILURE, iterate untouched
            else nlp_update(1.0, false);
            // residuals / cost are evaluated at the new iterate (acados get_residuals() for RTI,
            // get_cost()); this linearisation is reused by the next solve() call
            cost = linearize(w.X, w.U, true);
            lin_valid = true;
            nlp_residuals(w.QPI, w.QLAM, w.QT, res4);
        } else {
            const double tol = c.sm->P.tol;
            status = 2;  // ACADOS_MAXITER unless decided otherwise
            for (sqp_iter = 0; sqp_iter < c.pb->max_iter; sqp_iter++) {
                if (!lin_valid) cost = linearize(w.X, w.U, true);
                lin_valid = false;
                nlp_residuals(w.NPI, w.NLAM, w.NT, res4);
                if (res4[0] < tol && res4[1] < tol && res4[2] < tol && res4[3] < tol) { status = 0; lin_valid = true; break; }
                if (res4[0] != res4[0] || cost != cost) { status = 1; break; }
                const int qs = ipm_solve(&it);
                qp_iter += it;
                if (qs != 0 && qs != 1) { status = 4; break; }
                const double alpha = c.pb->fixed_step ? 1.0 : line_search(sqp_iter);
                nlp_update(alpha, true);
            }
            if (!lin_valid) { cost = linearize(w.X, w.U, true); lin_valid = true; }
        }
        lin_cost = cost;
        *sqp_iter_out = sqp_iter;
        *qp_iter_out = qp_iter;
        *cost_out = cost;
        PROF_ADD(PF_TOTAL, t0);
        return status;
    }

    // ======================================================================= closed loop
    // Simulator.run (simulator.py:199-241) for steps [step0, step1) of one instance.
    MPC_HD void rollout(const Outputs &out, int inst, int step0, int step1)
    {
        Smem &sm = *c.sm;
        const InstParams &P = sm.P;
        Ws &w = c.w;
        const int Nsim = c.pb->Nsim;
        const size_t sb = (size_t)inst * Nsim;
        bool lin_valid = false;
        if (step0 == 0) {
            // acados initial guess: x_k = x0, u_k = 0, all multipliers / QP memory 0 (SURVEY A.7 iv)
            const size_t tot = (size_t)(N + 1) * STAGE_DOUBLES + STATE_DOUBLES;
            ex.par([&](int lane) {
                for (size_t e = lane; e < tot; e += WAVE) w.X[e] = 0.0;  // X is the workspace base
            });
            ex.par([&](int lane) {
                for (int e = lane; e < (N + 1) * NX; e += WAVE) {
                    const int i = e % NX;
                    w.X[e] = i < 6 ? P.q0[i] : P.qdot0[i - 6];
                }
                if (lane < NX) sm.xhat[lane] = lane < 6 ? P.q0[lane] : P.qdot0[lane - 6];
                if (lane < NU) sm.u0[lane] = P.qdot0[lane];  // u[:,0] = qdot_0 (simulator.py:81)
            });
            log_state(out, inst, 0);
        } else {
            ex.par([&](int lane) {
                if (lane < NX) sm.xhat[lane] = w.state[lane];
            });
            lin_cost = w.state[12];
            lin_valid = w.state[25] != 0.0;
        }
        for (int i = step0; i < step1; i++) {
            int sqp_iter = 0, qp_iter = 0;
            double res4[4] = {0, 0, 0, 0}, cost = 0.0;
            const double t0 = ex.clock();
            const int status = nlp_step(lin_valid, &sqp_iter, &qp_iter, res4, &cost);
            const double t1 = ex.clock();
            PROF_T0(tp);
            // u = solver.get(0,'u'); RK4 plant step (simulation_model.py:111-117)
            ex.par([&](int lane) {
                if (lane < 6) {
                    const int j = lane;
                    const double u = w.U[j], wc = P.wcv[j], dt = P.dt;
                    const double q = sm.xhat[j], v = sm.xhat[6 + j];
                    const double k1q = v, k1v = -wc * v + wc * u;
                    const double v2 = v + 0.5 * dt * k1v;
                    const double k2q = v2, k2v = -wc * v2 + wc * u;
                    const double v3 = v + 0.5 * dt * k2v;
                    const double k3q = v3, k3v = -wc * v3 + wc * u;
                    const double v4 = v + dt * k3v;
                    const double k4q = v4, k4v = -wc * v4 + wc * u;
                    sm.logv[24 + j] = q + (dt / 6) * k1q + (dt / 3) * k2q + (dt / 3) * k3q + (dt / 6) * k4q;
                    sm.logv[30 + j] = v + (dt / 6) * k1v + (dt / 3) * k2v + (dt / 3) * k3v + (dt / 6) * k4v;
                    sm.u0[j] = u;
                }
                if (lane == 8) {
                    out.status[sb + i] = status;
                    out.sqp_iter[sb + i] = sqp_iter;
                    out.qp_iter[sb + i] = qp_iter;
                    out.cost[sb + i] = cost;
                    out.solver_time[sb + i] = t1 - t0;
                }
                if (lane >= 12 && lane < 16) out.residuals[(sb + i) * 4 + (lane - 12)] =
                    lane == 12 ? res4[0] : (lane == 13 ? res4[1] : (lane == 14 ? res4[2] : res4[3]));
            });
            ex.par([&](int lane) {
                if (lane < NX) sm.xhat[lane] = sm.logv[24 + lane];
            });
            log_state(out, inst, i + 1);
            PROF_ADD(PF_PLANT, tp);
        }
        ex.par([&](int lane) {
            if (lane < NX) w.state[lane] = sm.xhat[lane];
            if (lane == 12) { w.state[12] = lin_cost; w.state[25] = lin_valid ? 1.0 : 0.0; }
#ifdef MPCB_PROFILE
            if (lane < NPROF) {
                double v = prof[0];
#pragma unroll
                for (int j = 1; j < NPROF; j++) v = lane == j ? prof[j] : v;
                w.state[32 + lane] += v;
            }
#endif
        });
    }

    // simulation_model.py:87-90: log state, input, FK pose, rpy, J*qdot at column `col`
    MPC_HD void log_state(const Outputs &out, int inst, int col)
    {
        Smem &sm = *c.sm;
        const Robot &rb = sm.rb;
        const int T1 = c.pb->Nsim + 1;
        ex.par([&](int lane) {
            if (lane == 0) {
                double z[12];
#pragma unroll
                for (int i = 0; i < 12; i++) z[i] = sm.xhat[i];
                plant_log(rb, z, sm.logv);
            }
        });
        ex.par([&](int lane) {
            if (lane < 12) {
                out.z[((size_t)inst * 12 + lane) * T1 + col] = sm.xhat[lane];
                out.ee_pose[((size_t)inst * 12 + lane) * T1 + col] = sm.logv[lane];
            } else if (lane < 18) {
                out.u[((size_t)inst * 6 + (lane - 12)) * T1 + col] = sm.u0[lane - 12];
            } else if (lane < 21) {
                out.ee_rpy[((size_t)inst * 3 + (lane - 18)) * T1 + col] = sm.logv[12 + (lane - 18)];
            } else if (lane < 27) {
                out.ee_vel[((size_t)inst * 6 + (lane - 21)) * T1 + col] = sm.logv[15 + (lane - 21)];
            }
        });
    }
};

}  // namespace mpcb
