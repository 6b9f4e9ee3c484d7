/*
 * mpc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C fp64 restatement of the lynet55/robotic-mpc hot path
 * (SimulationManager.run_all -> Simulator.run: per-step kinematics, SQP /
 * SQP_RTI OCP-QP solve, RK4 plant step, logging).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference's arithmetic lives in un-vendored third-party
 * code (acados + HPIPM + BLASFEO, CasADi, Pinocchio -- none present in
 * /root/reference, none installable offline) and the reference holds no tests
 * or golden vectors.  This restatement follows the
 * reference's own problem definition line by line (citations at each
 * function) and the published acados SQP/SQP_RTI + HPIPM Mehrotra/Riccati
 * algorithms; it is pinned only by independent checks in tests/ (sympy
 * kinematics, scipy dense QP/NLP solves, closed forms).
 */
#ifndef MPC_ORACLE_H
#define MPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NQ 6
#define ORC_NX 12
#define ORC_NU 6
#define ORC_NW 18 /* stage variable [u; q; qdot] */
#define ORC_NR 17 /* effective residuals (18th has weight 0, trajectory_optimizer.py:133,152) */
#define ORC_NB 12 /* bounded comps per stage: u(6) then q(6) */

/* Kinematic chain constants (ur_description/urdf/ur10.urdf joints; loader.py:33-36
 * picks the end-effector frame).  place[i] = [R row-major (9); p (3)] of joint i in
 * its parent frame at q=0; place[6] = fixed end-effector frame in the last link. */
typedef struct {
    double place[7][12];
    double axis[6][3];
    double t_ee[3]; /* prediction_model.py:9 translation_ee_t */
} orc_robot;

/* One Simulator(**config) instance (simulator.py:18-35). */
typedef struct {
    int N;           /* prediction_horizon */
    int Nsim;        /* int(simulation_time/dt), simulator.py:41 */
    int solver_type; /* 0 = SQP (trajectory_optimizer.py:60), 1 = SQP_RTI */
    int max_iter;    /* nlp_solver_max_iter, trajectory_optimizer.py:67 */
    int qp_iter_max; /* HPIPM iter_max (acados qp_solver_iter_max default 50) */
    double dt;
    double tol;    /* acados nlp tol (default 1e-6) */
    double qp_tol; /* trajectory_optimizer.py:63 */
    double wcv[6], q0[6], qdot0[6];
    double qmin[6], qmax[6], umin[6], umax[6];
    double w_u, w_qddot, px_ref, vy_ref;
    double coeffs[6]; /* a b c d e f, surface.py:14-17 */
    double w_task[5]; /* trajectory_optimizer.py:44-48 (all 50.0) */
    int integrator;   /* plant integrator, simulation_model.py:39-49: 0 RK4 (simulator.py:85), 1 Euler, 2 RK2, 3 RK3 */
    /* acados nlp_solver_tol_eq / _ineq / _comp (`tol` above is nlp_solver_tol_stat); a value <= 0 means
     * "same as tol" (the acados `tol` setter writes all four; simulator.py:129-135 forwards any of them) */
    double tol_eq, tol_ineq, tol_comp;
    /* acados levenberg_marquardt (default 0): dt*lm*I is added to every stage Hessian, lm*I to the terminal one */
    double lm;
    /* bound-inactive fast path of the QP solve (an addition of this build, see ipm_fast_path in mpc_oracle.c):
     * 0 off (every QP through the interior-point loop, as HPIPM), 1 on (attempted by the rule in solve_qp),
     * 2 attempted at every QP (diagnostic), 3 the active-set form (diagnostic: exists in the oracle only -- measured on the
     * engines and not adopted) */
    int fast_path;
} orc_params;

/* Per-instance outputs, C-contiguous [row][time] like the reference's logs
 * (simulation_model.py:25-29, simulator.py:59-65). Any pointer may be NULL. */
typedef struct {
    double *z;        /* [12][Nsim+1] */
    double *u;        /* [6][Nsim+1]  */
    double *ee_pose;  /* [12][Nsim+1] */
    double *ee_rpy;   /* [3][Nsim+1]  */
    double *ee_vel;   /* [6][Nsim+1]  */
    int *status;      /* [Nsim] */
    int *sqp_iter;    /* [Nsim] */
    int *qp_iter;     /* [Nsim] total IPM iterations in the step */
    double *residuals; /* [Nsim][4] */
    double *cost;      /* [Nsim] */
    double *solver_time; /* [Nsim] seconds */
} orc_output;

/* ---- kinematics (prediction_model.py:126-173, simulation_model.py:60-77) ---- */
void orc_fk(const orc_robot *rb, const double *q, double *pose12);
void orc_jacobian_world(const orc_robot *rb, const double *q, double *J36 /* row-major 6x6 */);
void orc_rpy(const double *pose12, double *rpy3);
/* y in R^15, prediction_model.py:299-314 */
void orc_task_output(const orc_robot *rb, const double *q, const double *qd, double *y15);
/* g(5) and its Jacobian wrt [q;qdot] (5x12 row-major), trajectory_optimizer.py:104-126 */
void orc_task_g(const orc_robot *rb, const double *coeffs, const double *q, const double *qd,
                double *g5, double *G60);

/* ---- model (prediction_model.py:87-115, 317-326) ---- */
void orc_lti(const double *wcv, double Ts, double *a12, double *a22, double *b1, double *b2);
void orc_rk4(const double *wcv, double dt, const double *z, const double *u, double *znext);
void orc_plant_step(int integrator, const double *wcv, double dt, const double *z, const double *u, double *znext);

/* ---- stage residual / Jacobian (trajectory_optimizer.py:131-160) ---- */
void orc_stage_residual(const orc_robot *rb, const orc_params *p, const double *x, const double *u,
                        double *r17, double *Jr /* 17x18 row-major, cols [u;q;qdot] */);

/* ---- OCP-QP interior point on this block structure (HPIPM restatement) ----
 * Dense stage data: H[N+1][18*18], g[N+1][18], b[N][12], A[144], B[72] (LTI),
 * bounds lb/ub [N+1][12] on [u;q] (use +-1e30 for "none": they are skipped),
 * dx0[12] fixed initial step.  Iterate (in/out, warm start = 2 semantics):
 * w[N+1][18], pi[N][12], lam[N+1][24], t[N+1][24] (lb then ub).
 * Returns HPIPM-style status 0 ok / 1 max-iter / 2 min-step / 3 NaN. */
int orc_qp_ipm(int N, const double *H, const double *g, const double *b, const double *A,
               const double *B, const double *lb, const double *ub, const double *dx0,
               double *w, double *pi, double *lam, double *t, double tol, int iter_max,
               int *iters, double *res4);

/* The bound-inactive fast path on the same data (an addition of this build, not HPIPM; see ipm_fast_path): one Riccati
 * solve of the equality-constrained QP from w = 0; returns 1 and writes (w, pi, lam = 0, t = slack) when every bounded
 * component clears its bounds by 1e-3, else returns 0 and leaves (w, pi, lam, t) untouched. */
int orc_qp_fast(int N, const double *H, const double *g, const double *b, const double *A,
                const double *B, const double *lb, const double *ub, const double *dx0,
                double *w, double *pi, double *lam, double *t);

/* ---- whole closed loop (simulator.py:199-241) ---- */
int orc_run(const orc_robot *rb, const orc_params *p, orc_output *out);

/* ---- one solver handle for step-level tests ---- */
typedef struct orc_solver orc_solver;
orc_solver *orc_solver_create(const orc_robot *rb, const orc_params *p);
void orc_solver_destroy(orc_solver *s);
/* set x0, solve, return status; mirrors simulator.py:210-221 */
int orc_solver_step(orc_solver *s, const double *xhat, double *u0, int *sqp_iter, int *qp_iter,
                    double *res4, double *cost);
/* copy out the iterate: x[N+1][12], u[N][6], pi[N][12] */
void orc_solver_get_iterate(const orc_solver *s, double *x, double *u, double *pi);

#ifdef __cplusplus
}
#endif
#endif
