/*
 * mpcbatch.h -- C ABI of libmpcbatch.so, the MI355X-native batched MPC rollout engine.
 *
 * Drop-in boundary for the hot path of lynet55/robotic-mpc:
 *     SimulationManager.run_all -> Simulator.__init__ / Simulator.run
 *     (simulator.py:641-676, 18-127, 199-241)
 * i.e. everything the reference does through acados_template's ctypes binding to its
 * per-instance generated libacados_ocp_solver_<model>.so:
 *     AcadosOcpSolver(ocp, json_file)            trajectory_optimizer.py:183-186
 *     solver.set(0,'lbx'|'ubx',x)                simulator.py:210-211
 *     solver.solve()                             simulator.py:212
 *     solver.get(0,'u')                          simulator.py:213
 *     solver.get_stats('sqp_iter'|'time_tot')    simulator.py:218,220
 *     solver.get_residuals(), get_cost()         simulator.py:219,221
 * plus the plant step and logging of simulation_model.Robot.update
 * (simulation_model.py:85-91).  Where the reference builds and drives ONE solver per
 * simulation, this ABI takes a BATCH of simulations (one 72-double parameter record each)
 * and runs all closed loops on the GPU, one wavefront per simulation.
 *
 * Conventions: plain C, no exceptions; every function returns 0 on success or a negative
 * MPCB_E* code, with text in mpcb_last_error(); the caller owns every buffer it passes;
 * a handle is bound to one device and is not thread-safe; per-simulation solver failures
 * are DATA (status arrays, acados codes 0/1/2/3/4), never call failures.
 * There is no CPU fallback: without a HIP device mpcb_create fails.
 */
#ifndef MPCBATCH_H
#define MPCBATCH_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCB_VERSION 200      /* 0.2.0 */
#define MPCB_NPARAM 72        /* doubles per simulation, layout below */
#define MPCB_NROBOT 105       /* doubles of kinematic constants, layout below */

#define MPCB_OK 0
#define MPCB_EINVAL (-1)      /* bad argument / inconsistent sizes */
#define MPCB_ENODEV (-2)      /* no usable HIP device */
#define MPCB_ENOMEM (-3)      /* device allocation failed */
#define MPCB_EHIP (-4)        /* HIP runtime error, see mpcb_last_error */
#define MPCB_ESTATE (-5)      /* call order violated (e.g. rollout before setup) */

#define MPCB_SOLVER_SQP 0     /* trajectory_optimizer.py:60 (default) */
#define MPCB_SOLVER_SQP_RTI 1 /* solver_options {'nlp_solver_type': 'SQP_RTI'} */

#define MPCB_PRECISION_FP64 0          /* everything in fp64 (the reference's arithmetic)                      */
#define MPCB_PRECISION_FP32_RICCATI 1  /* Riccati factor K, P, R~^-1, p and the three solve sweeps in fp32; iterate,
                                          residuals, right-hand sides, steps and all outputs stay fp64
                                          (BASELINE.json configs[4]; SQP_RTI on the throughput engine only).
                                          An OPT-IN study leg: slower than fp64 at every size measured (-6..-12 %),
                                          never chosen by mpcb_setup's routing -- only by this field.             */

/* Two kernel families sit behind this ABI: the LATENCY engine (one workgroup of 4-8
 * wavefronts and half or all of a CU's LDS per simulation; batches up to a few simulations per CU, and most SQP runs) and
 * the THROUGHPUT engine (one wavefront per simulation, records streamed; SQP_RTI batches of >= MPCB_STREAM_MIN_BATCH
 * simulations, full-SQP batches of >= MPCB_STREAM_MIN_BATCH_SQP simulations running >= MPCB_STREAM_MIN_STEPS_SQP steps,
 * every fp32-Riccati run, every ragged batch).  mpcb_setup picks; the environment variable MPCB_ENGINE=latency|stream
 * overrides the choice where both apply.  An SQP_RTI bucket of at least as many simulations as the GPU holds wavefronts of
 * the throughput engine (8 per CU) is launched as a work queue over (simulation, MPCB_STREAM_CHUNK = 10 closed-loop steps)
 * items: same results bit for bit, balanced launch (MPCB_STREAM_CHUNK=0 turns it off; a hand-off that does not complete
 * within a bound derived from the work limit of one chunk -- 4 x chunk steps x SQP iterations x QP iterations x (N+1) x 20 us
 * + 30 s; MPCB_QUEUE_TIMEOUT_S overrides it -- is reported by mpcb_sync as MPCB_EHIP instead of hanging). */
/* Measured crossovers, N=100, 600 steps, one MI355X, fast path of the QP solve on, the throughput engine's first pass of a step
 * item-parallel (profiles/r04_engine_sweep2.txt; latency engine two simulations per CU vs throughput engine, steps/s).  SQP_RTI: 1024
 * simulations 2.40 M vs 2.10 M, 1280: 2.34 M vs 2.55 M, 2048: 2.55 M vs 3.69 M, 4096: 2.67 M vs 4.08 M.  Full SQP: 2048 simulations
 * 474 k vs 435 k, 2560: 447 k vs 489 k, 4096: 503 k vs 601 k -- and once more at the round's last kernels (joint-angle sincos, SQP merit pass without
 * spills: the latency engine gained more; profiles/r04_engine_sweep3.txt): 2560: 620 k vs 600 k, 3072: 665 k vs 670 k, 3584: 686 k vs 715 k, 4096: 723 k
 * vs 758 k; 200 steps: 3072: 283 k vs 244 k, 4096: 308 k vs 307 k (after the throughput engine's last SQP fixes, r04_engine_sweep4.txt: 2560: 618 k
 * vs 615-647 k, 3072: 661-671 k vs 685-713 k, 4096: 845-870 k on the throughput engine; 3072 x 200 steps: 282 k vs 261 k -- thresholds unchanged).  SQP_RTI there: 1280: 2.49 M vs 2.60 M, 4096: 4.23 M on the throughput engine. */
#define MPCB_STREAM_MIN_BATCH_SQP 3072   /* full SQP: from this many simulations on ... */
#define MPCB_STREAM_MIN_STEPS_SQP 300    /* ... for runs of at least this many closed-loop steps */
#define MPCB_STREAM_MIN_BATCH 1280

typedef struct mpcb_handle mpcb_handle;

/* Batch-uniform part of the configuration (one launch = one bucket of simulations that
 * share these). Mirrors Simulator.__init__ / MPC.__init__ arguments as noted. */
typedef struct {
    int batch;        /* number of simulations in this call                               */
    int N;            /* prediction_horizon (simulator.py:42)                             */
    int Nsim;         /* int(simulation_time/dt) (simulator.py:41)                        */
    int solver_type;  /* MPCB_SOLVER_*                                                    */
    int max_iter;     /* nlp_solver_max_iter (trajectory_optimizer.py:67)                 */
    int qp_iter_max;  /* acados qp_solver_iter_max (default 50)                           */
    int fixed_step;   /* 1: globalization FIXED_STEP, 0: MERIT_BACKTRACKING (:68)         */
    int precision;    /* MPCB_PRECISION_*: arithmetic of the Riccati factor and solve sweeps */
} mpcb_problem;

/* Per-simulation parameter record, MPCB_NPARAM doubles (simulator.py:18-35):
 *   [0] dt  [1] tol (acados nlp tol, 1e-6)  [2] qp_tol (trajectory_optimizer.py:63)
 *   [3] w_u [4] w_qddot [5] px_ref [6] vy_ref [7] plant integrator (0 RK4, 1 Euler, 2 RK2, 3 RK3; simulation_model.py:39-49)
 *   [8..13] wcv   [14..19] q_0   [20..25] qdot_0   [26..31] q_min   [32..37] q_max
 *   [38..43] qdot_min (lbu)   [44..49] qdot_max (ubu)
 *   [50..55] surface_coeffs a,b,c,d,e,f (surface.py:14-17)
 *   [56..60] task weights (trajectory_optimizer.py:44-48, all 50.0)
 *   [61] nlp_solver_tol_eq [62] nlp_solver_tol_ineq [63] nlp_solver_tol_comp -- 0 means "same as [1]", which is
 *        nlp_solver_tol_stat (any acados option reaches the solver through simulator.py:129-135)
 *   [64] levenberg_marquardt (acados: dt*lm*I added to every stage Hessian, lm*I to the terminal one)
 *   [65] this simulation's prediction horizon, when the simulations of one call have DIFFERENT horizons ("ragged"
 *        batch, e.g. a grid search over prediction_horizon run as one launch): 1 <= [65] <= mpcb_problem.N, and
 *        mpcb_problem.N is the largest of them; 0 means N.  Ragged batches run on the throughput engine.
 *   [66] bound-inactive fast path of the QP solve (csrc/mpc_ipm.h): 0 = on (the default), 1 = off.  On: a QP whose
 *        equality-constrained minimiser -- ONE Riccati factorisation -- keeps every bounded component >= 1e-3 inside its bounds is
 *        solved by that factorisation alone (the solution of the strictly convex QP, lam = 0); every other QP, and every QP when
 *        off, goes through the HPIPM-style interior-point loop.  `qp_iter` then counts Riccati factorisations.  Not used by the
 *        fp32-Riccati leg (one fp32 solve is not a solution to qp_tol).
 *   [67..71] reserved (0)
 * Bounds with |value| >= 1e29 are treated as absent.
 *
 * Kinematic constants, MPCB_NROBOT doubles (what loader.py:24-36 extracts from the URDF):
 *   [0..83]  7 placements [R row-major (9); p (3)]: joint i in its parent at q=0, i=0..5,
 *            then the end-effector frame in the last link
 *   [84..101] 6 unit joint axes   [102..104] translation_ee_t (prediction_model.py:9)
 */

/* Result logs, batch-major; per simulation the shapes of the reference's own logs
 * (simulation_model.py:25-29, simulator.py:59-65).  T1 = Nsim+1. */
typedef struct {
    double *z;           /* [batch][12][T1]  q;qdot                      */
    double *u;           /* [batch][6][T1]   u[:,0]=qdot_0, u[:,i+1]=u_i */
    double *ee_pose;     /* [batch][12][T1]  p; R row-major              */
    double *ee_rpy;      /* [batch][3][T1]                               */
    double *ee_vel;      /* [batch][6][T1]   J_world * qdot              */
    int *status;         /* [batch][Nsim]    acados status 0/1/2/3/4     */
    int *sqp_iter;       /* [batch][Nsim]                                */
    int *qp_iter;        /* [batch][Nsim]    Riccati factorisations: interior-point iterations (+1 per fast-path attempt) */
    double *residuals;   /* [batch][Nsim][4] stat, eq, ineq, comp        */
    double *cost;        /* [batch][Nsim]                                */
    double *solver_time; /* [batch][Nsim]    device seconds of solver.solve() (simulator.py:209-214,220)  */
    double *errors;      /* [batch][7][T1]   e1..e5, p_task_z, p_ee_y of Simulator.errors (simulator.py:265-344),
                                             computed with the log column on the device                  */
    double *plant_time;  /* [batch][Nsim]    device seconds of the plant step + FK / J qdot / error logging
                                             (integration_time, simulator.py:224-226)                    */
} mpcb_result;

#define MPCB_NSUMMARY 24      /* doubles per simulation written by mpcb_summary, layout below */

int mpcb_version(void);
/* Number of HIP devices visible; 0 when none (never an error by itself). */
int mpcb_device_count(void);

/* Create a handle on `device` (>= 0).  Fails with MPCB_ENODEV when no GPU is usable. */
int mpcb_create(mpcb_handle **h, int device);
void mpcb_destroy(mpcb_handle *h);
const char *mpcb_last_error(const mpcb_handle *h);

/* Bytes of device workspace mpcb_setup will hold for `p`. */
size_t mpcb_workspace_bytes(const mpcb_problem *p);
/* Bytes of one simulation's result logs (sum over the mpcb_result arrays). */
size_t mpcb_result_bytes_per_sim(const mpcb_problem *p);

/* Replaces Simulator.__init__ + MPC.finalize_solver for the whole batch: packs the
 * parameter records (host pointers), uploads them, sizes the workspace.  No code generation. */
int mpcb_setup(mpcb_handle *h, const mpcb_problem *p, const double *params_host, const double *robot_host);

/* Replaces the Simulator.run loop for closed-loop steps [step0, step1) of every simulation.
 * `out_dev` holds DEVICE pointers (caller-allocated, e.g. torch tensors); `stream` is a
 * hipStream_t (NULL = default stream).  Asynchronous; step0 must continue where the previous
 * call stopped, or be 0, which restarts every simulation from its initial state. */
int mpcb_rollout(mpcb_handle *h, int step0, int step1, const mpcb_result *out_dev, void *stream);

/* Wait for the stream of the last rollout. */
int mpcb_sync(mpcb_handle *h);

/* Device time of the last mpcb_rollout launch, measured with HIP events on its stream (ms). */
int mpcb_last_kernel_ms(mpcb_handle *h, float *ms);

/* Static resources of the rollout kernel: VGPRs, SGPRs(0 if unknown), LDS bytes, scratch bytes. */
int mpcb_kernel_info(mpcb_handle *h, int *vgprs, int *sgprs, int *lds_bytes, int *scratch_bytes);

/* Kernel family mpcb_setup chose for the current problem: 0 latency engine, 1 throughput engine (< 0: error). */
int mpcb_engine(mpcb_handle *h);
/* The family mpcb_setup WOULD choose for a uniform (not ragged) problem of this shape: host logic only, no device touched. */
int mpcb_engine_for(const mpcb_problem *p);

/* Launch geometry chosen by mpcb_setup for the current batch: wavefronts cooperating on one
 * simulation (one workgroup per simulation), and the dynamic-LDS chunk pool per workgroup. */
int mpcb_launch_info(mpcb_handle *h, int *waves_per_sim, int *pool_bytes);

/* Replaces Simulator.metrics / solver_stats / timings / get_summary (simulator.py:347-448, 509-547) for the whole
 * batch: one pass over the device logs of a finished rollout.  `out_dev` as for mpcb_rollout; `summary_dev` is a
 * DEVICE array [batch][MPCB_NSUMMARY]:
 *   [0..4] rmse_e1..e5  [5..9] itse_e1..e5  [10] weighted_rmse  [11] total_sqp_iterations  [12] avg_sqp_iterations
 *   [13] num_failures  [14] max_kkt_residual  [15] total_solver_time  [16] avg_mpc_time  [17] avg_solver_time
 *   [18] avg_integration_time  [19] total_computation_time  [20] total_qp_iterations  [21..23] reserved.
 * mpc_time and solver_time are both the device time of the solve (no Python call overhead exists here),
 * integration_time is `plant_time`.  Asynchronous on `stream`. */
int mpcb_summary(mpcb_handle *h, const mpcb_result *out_dev, double *summary_dev, void *stream);

/* Convenience for callers without device buffers of their own: setup + rollout(0,Nsim) +
 * copy-back into HOST arrays `out_host`. */
int mpcb_run(mpcb_handle *h, const mpcb_problem *p, const double *params_host, const double *robot_host,
             const mpcb_result *out_host);

#ifdef __cplusplus
}
#endif
#endif /* MPCBATCH_H */
