// mpc_ipm.h -- the interior-point formulas BOTH engines use (latency engine mpc_core.h, throughput engine mpc_stream.h), in one
// place: a change to HPIPM's step rules (oracle/mpc_oracle.c ipm_solve is the restatement they follow) lands here once.
// Everything is a small inline function of scalars; the callers own the data movement (LDS rows, registers, HBM items).
// Reference being restated: acados/HPIPM behind trajectory_optimizer.py:183-186 (d_ocp_qp_ipm_solve: warm start 2, update_var,
// compute_lam_t, compute_alpha, compute_centering_correction), semantics as listed in SURVEY.md A.7.
#pragma once
#include "mpc_layout.h"

namespace mpcb {

// 1/d for the LDL' pivots and the slack divisions: hardware reciprocal seed (measured 4.6e-8 relative on gfx950,
// scripts/microbench/rcptest.hip) + one third-order correction x (1 + e + e^2), e = 1 - d x: three dependent FMAs to full fp64
// accuracy instead of the ~40-instruction IEEE division on the sequential critical path.
MPC_HD double fast_rcp(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double x = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, x, 1.0);
    return fma(x, fma(e, e, e), x);
#else
    return 1.0 / d;
#endif
}

namespace ipm {

// which (stage, bounded component) pairs carry constraints (trajectory_optimizer.py:164-171): inputs j < 6 on stages 0..N-1,
// joint positions j >= 6 on stages 1..N-1
MPC_HD bool has_comp(int N, int k, int j) { return j < 6 ? (k < N) : (k >= 1 && k < N); }

// HPIPM warm_start = 2: (lam, t) of the previous QP, clamped to >= 0.1; an absent bound side holds lam = 0, t = 1
MPC_HD double warm_lam(bool on, double lam) { return on ? fmax(lam, 0.1) : 0.0; }
MPC_HD double warm_t(bool on, double t) { return on ? fmax(t, 0.1) : 1.0; }

// HPIPM update_var for lam and t: x + a dx, floored at 1e-16 (absent sides stay as they are)
MPC_HD double step_floor(bool on, double x, double a, double dx) { return on ? fmax(x + a * dx, 1e-16) : x; }

// step actually taken for a feasible length alpha (HPIPM: alpha * ((1 - alpha) * 0.99 + alpha * 0.9999999))
MPC_HD double step_scale(double alpha) { return alpha * ((1.0 - alpha) * 0.99 + alpha * 0.9999999); }

// Mehrotra centering parameter sigma = (mu_aff / mu)^3
MPC_HD double sigma(double mu_aff, double mu) { const double tmp = mu_aff / mu; return tmp * tmp * tmp; }

// One bound side of compute_lam_t + compute_alpha + the centering sums, branch-free: `sdv` is +dv for a lower, -dv for an upper
// bound.  An absent side (on = false) holds lam = 0, t = 1, rd = rm = 0: its dt is forced to 0 and everything else vanishes by
// itself.  al: running largest feasible step; a0, a1, a2: sums with mu(alpha) * nc = a0 + alpha a1 + alpha^2 a2.
MPC_HD void lam_t_side(bool on, double sdv, double l, double t, double rd, double rm, double &al, double &a0, double &a1, double &a2,
                       double &dt_o, double &dl_o)
{
    const double dt = on ? sdv + rd : 0.0;
    const double dl = on ? -(rm + l * dt) * fast_rcp(t) : 0.0;
    const double c1 = -l * fast_rcp(dl);
    al = (dl < 0 && l + al * dl < 0) ? c1 : al;
    const double c2 = -t * fast_rcp(dt);
    al = (dt < 0 && t + al * dt < 0) ? c2 : al;
    a0 += l * t; a1 += l * dt + t * dl; a2 += dl * dt;
    dt_o = dt; dl_o = dl;
}

// Centering corrector of one bound side (compute_centering_correction): rm <- lam t + dlam_aff dt_aff - sigma mu; returns the
// side's contribution (rm + lam rd) / t to the condensed gradient (added for a lower, subtracted for an upper bound)
MPC_HD double corrector_side(bool on, double l, double t, double dl_aff, double dt_aff, double rd, double sigma_mu, double &rm_o)
{
    const double rmv = on ? l * t + dl_aff * dt_aff - sigma_mu : 0.0;
    rm_o = rmv;
    return on ? (rmv + l * rd) * fast_rcp(t) : 0.0;
}

// ---- bound-inactive fast path (both engines; oracle/mpc_oracle.c ipm_fast_path is the restatement) -----------------------------
// The GN QP is strictly convex, so when its EQUALITY-constrained minimiser (one Riccati factorisation with Gamma = 0, from w = 0
// with x_0 embedded) keeps every bounded component at least FAST_MARGIN inside its bounds, it IS the QP's solution with all bound
// multipliers zero, and the interior-point loop -- two factorisations to bring lam t from the 0.1 warm-start clamp below qp_tol --
// is skipped.  Accepted: (w, pi) = the solve's, lam = 0, t = slack.  Rejected: the warm start is untouched and the loop runs as
// before.  A rejected attempt suspends further attempts for 1, 2, 4, 8, 8, ... QPs (back to none after an acceptance).
constexpr double FAST_MARGIN = 1e-3;
// slack of one bound side of a candidate step dv (lo / hi: the bound relative to the NLP iterate) and the acceptance test;
// an absent side holds t = 1.  NaN fails the comparison, i.e. rejects.
MPC_HD bool fast_side(bool on_lo, bool on_hi, double dv, double lo, double hi, double &t_lo, double &t_hi)
{
    t_lo = on_lo ? dv - lo : 1.0;
    t_hi = on_hi ? hi - dv : 1.0;
    return (!on_lo || t_lo >= FAST_MARGIN) && (!on_hi || t_hi >= FAST_MARGIN);
}
MPC_HD int fast_backoff(int back) { return back ? (2 * back < 8 ? 2 * back : 8) : 1; }

}  // namespace ipm
}  // namespace mpcb
