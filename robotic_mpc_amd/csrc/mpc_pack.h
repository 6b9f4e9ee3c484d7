// mpc_pack.h -- host-side packing of one Simulator(**config) (simulator.py:18-35) into the
// device parameter record.  The C-ABI takes, per instance, MPCB_NPARAM doubles:
//   [0] dt [1] tol [2] qp_tol [3] w_u [4] w_qddot [5] px_ref [6] vy_ref [7] plant integrator (0 RK4, 1 Euler, 2 RK2, 3 RK3)
//   [8..13] wcv  [14..19] q_0  [20..25] qdot_0  [26..31] q_min  [32..37] q_max
//   [38..43] qdot_min  [44..49] qdot_max  [50..55] surface coeffs a..f  [56..60] task weights
//   [61] nlp_solver_tol_eq [62] nlp_solver_tol_ineq [63] nlp_solver_tol_comp  (0: same as [1] = tol_stat)
//   [64] levenberg_marquardt   [65] this simulation's prediction horizon (0: the launch's N)
//   [66] 1: bound-inactive fast path of the QP solve OFF (0, the default: on)   [67..71] reserved
#pragma once
#include <math.h>

#include "mpc_layout.h"

#define MPCB_NPARAM 72

namespace mpcb {

inline void pack_inst_params(const double *p, InstParams *P)
{
    P->dt = p[0]; P->tol = p[1]; P->qp_tol = p[2]; P->w_u = p[3]; P->w_qddot = p[4];
    P->px_ref = p[5]; P->vy_ref = p[6]; P->integ = p[7]; P->lm = p[64]; P->n_hor = p[65]; P->fast_off = p[66];
    P->tol_eq = p[61] > 0.0 ? p[61] : p[1]; P->tol_ineq = p[62] > 0.0 ? p[62] : p[1]; P->tol_comp = p[63] > 0.0 ? p[63] : p[1];
    for (int j = 0; j < 6; j++) {
        P->wcv[j] = p[8 + j]; P->q0[j] = p[14 + j]; P->qdot0[j] = p[20 + j];
        P->qmin[j] = p[26 + j]; P->qmax[j] = p[32 + j]; P->umin[j] = p[38 + j]; P->umax[j] = p[44 + j];
        P->coeffs[j] = p[50 + j];
        // exact zero-order hold of the decoupled velocity loop (prediction_model.py:93-102)
        P->a22[j] = exp(-P->wcv[j] * P->dt);
        P->a12[j] = (1.0 - P->a22[j]) / P->wcv[j];
        P->b2[j] = 1.0 - P->a22[j];
        P->b1[j] = P->dt - P->a12[j];
        // qddot_k = (qdot_{k+1} - qdot_k)/Ts = cq (u - qdot)   (prediction_model.py:322-326)
        P->cq[j] = P->b2[j] / P->dt;
    }
    for (int i = 0; i < 5; i++) P->w_task[i] = p[56 + i];
}

}  // namespace mpcb
