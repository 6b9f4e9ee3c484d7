// mpc_kernel.hip -- gfx950 kernel + C ABI (include/mpcbatch.h) of the batched MPC engine.
//
// Launch geometry: one workgroup of NWV wavefronts (1, 2, 4 or 8; mpcb_setup picks it from the
// batch) per simulation instance, grid = batch.  DevExec<NWV> below is the device executor of the
// engine template (mpc_core.h): workgroup barriers between bulk-synchronous phases, wave-local
// fences inside a recursion, role-split windows (overlap3), DPP/readlane cross-lane helpers.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mpcbatch.h"
#include "mpc_core.h"
#include "mpc_pack.h"
#include "mpc_stream.h"

using namespace mpcb;

static_assert(sizeof(mpcb_problem) == sizeof(Problem), "ABI struct mismatch");
static_assert(sizeof(mpcb_result) == sizeof(Outputs), "ABI struct mismatch");
static_assert(sizeof(Robot) == MPCB_NROBOT * sizeof(double), "robot layout");

// LDS of the workgroup (= one simulation): fixed working set + chunk pool.
__shared__ __attribute__((aligned(16))) Smem g_sm;
extern __shared__ __attribute__((aligned(16))) double g_pool[];

// Stateless on purpose: inside a non-inlined pass the executor is reached through `this`, and a
// data member (e.g. a cached lane id) would be re-loaded from the stack at every phase.
// NWV wavefronts (one workgroup) cooperate on one simulation.
#ifndef MPCB_POLL_SLEEP
#define MPCB_POLL_SLEEP 2
#endif

// WPE (wavefronts per SIMD the kernel is compiled for) only makes the executor -- and with it every pass of the
// engine template -- a distinct type per kernel variant, so each variant gets its own register allocation.
template <int NWV, int WPE = 1>
struct DevExec {
    static constexpr int NT = WAVE * NWV;
    static constexpr int VGPR_BUDGET = (WPE >= 2 || NWV > 4) ? 256 : 512;   // registers per lane this variant is compiled for (two wavefronts per SIMD: 256)
    __device__ __forceinline__ static int lane_id() { return (int)threadIdx.x; }
    __device__ __forceinline__ Smem &smem() const { return g_sm; }
    __device__ __forceinline__ double *pool() const { return g_pool; }
    // value known to be identical in every lane -> scalar register (and scalar control flow)
    __device__ __forceinline__ static int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
    __device__ __forceinline__ static bool uni(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
    __device__ __forceinline__ static double uni(double v)
    {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
    }
    template <class T>
    __device__ __forceinline__ static T *uni(T *p)
    {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return (T *)(((unsigned long long)hi << 32) | lo);
    }
    // per-lane registers that live across phases
    template <class T>
    struct PerLane {
        T v;
        __device__ __forceinline__ T &at(int) { return v; }
    };
    __device__ __forceinline__ static void wave_fence()
    {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // phase on all NT lanes, then a workgroup barrier (a wave-local fence when one wave owns the sim)
    template <class F>
    __device__ __forceinline__ void par(F &&f)
    {
        f(lane_id());
        if (NWV == 1) wave_fence();
        else __syncthreads();
    }
    // wave-local phase on EVERY wavefront, no workgroup barrier: consecutive wpar phases of one wavefront see each
    // other's LDS writes (in-order LDS); data of another wavefront needs a barrier() first
    template <class F>
    __device__ __forceinline__ void wpar(F &&f)
    {
        f(lane_id());
        wave_fence();
    }
    __device__ __forceinline__ static void barrier()
    {
        if (NWV == 1) wave_fence();
        else __syncthreads();
    }
    // phase on wavefront 0 only; consecutive seq phases need no s_barrier (one wave, in-order LDS)
    template <class F>
    __device__ __forceinline__ void seq(F &&f)
    {
        if (NWV == 1 || threadIdx.x < WAVE) {
            f(lane_id());
            wave_fence();
        }
    }
    // A stage-by-stage recursion (`fg`, made of seq phases, wavefront 0) with the other wavefronts
    // doing barrier-free background work `bg(lane, lanes)` (chunk copies for the neighbouring
    // chunks) in its shadow; ends with the workgroup barrier.  With one wavefront per simulation
    // the two simply run one after the other.
    template <class FG, class BG>
    __device__ __forceinline__ void overlap(FG &&fg, BG &&bg)
    {
        if (NWV == 1) {
            fg();
            wave_fence();
            bg(lane_id(), std::integral_constant<int, WAVE>{});
            wave_fence();
        } else {
            if (threadIdx.x < WAVE) fg();
            else bg(lane_id() - WAVE, std::integral_constant<int, WAVE *(NWV > 1 ? NWV - 1 : 1)>{});
            __syncthreads();
        }
    }
    // Three concurrent roles: wavefront 0 runs `fg` (seq phases), wavefront 1 runs `mid` (sub
    // phases, wave-local), the remaining wavefronts run the barrier-free `bg(lane, lanes)`.
    // With fewer wavefronts the roles run one after the other on the last wavefront.
    template <class FG, class MID, class BG>
    __device__ __forceinline__ void overlap3(FG &&fg, MID &&mid, BG &&bg)
    {
        if (NWV == 1) {
            fg(); wave_fence();
            mid(); wave_fence();
            bg(lane_id(), std::integral_constant<int, WAVE>{});
            wave_fence();
        } else if (NWV == 2) {
            if (threadIdx.x < WAVE) fg();
            else { mid(); wave_fence(); bg(lane_id() - WAVE, std::integral_constant<int, WAVE>{}); }
            __syncthreads();
        } else {
            if (threadIdx.x < WAVE) fg();
            else if (threadIdx.x < 2 * WAVE) mid();
            else bg(lane_id() - 2 * WAVE, std::integral_constant<int, WAVE *(NWV > 2 ? NWV - 2 : 1)>{});
            __syncthreads();
        }
    }
    // A loop of `nwin` windows of three roles WITHOUT workgroup barriers between the windows (four wavefronts and more):
    // every role runs its own loop and the roles meet through LDS counters (post / post_add / await) only; one barrier at
    // the end.  With fewer wavefronts (roles share a wavefront) each window is an overlap3 with its barrier.
    static constexpr int BG_WAVES = NWV > 2 ? NWV - 2 : 1;
    template <class FG, class MID, class BG>
    __device__ __forceinline__ void pipeline3(int nwin, FG &&fg, MID &&mid, BG &&bg)
    {
        if (NWV >= 4) {
            if (threadIdx.x < WAVE) { for (int ci = 0; ci < nwin; ci++) fg(ci); }
            else if (threadIdx.x < 2 * WAVE) { for (int ci = 0; ci < nwin; ci++) mid(ci); }
            else { for (int ci = 0; ci < nwin; ci++) { bg(ci, lane_id() - 2 * WAVE, std::integral_constant<int, WAVE * BG_WAVES>{}); wave_fence(); } }
            __syncthreads();
        } else {
            for (int ci = 0; ci < nwin; ci++)
                overlap3([&]() { fg(ci); }, [&]() { mid(ci); }, [&](int lane, auto nl) { bg(ci, lane, nl); });
        }
    }
    __device__ __forceinline__ static void post_add(int *flag, int v)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __hip_atomic_fetch_add(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // progress counter between the recursion wavefront and the one following it (LDS, same CU):
    // a wavefront's LDS operations complete in issue order, so data written before post() is
    // visible to whoever has seen the posted value.
    __device__ __forceinline__ static void post(int *flag, int v)
    {
        // compiler-only ordering: the LDS unit executes one wavefront's DS instructions in issue order,
        // so no s_waitcnt is needed between the data writes and the flag write
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ static void await(int *flag, int v)
    {
        if (NWV > 1) {   // with a single wavefront the recursion has finished before the follower starts
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(MPCB_POLL_SLEEP);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // one step of a recursion that runs on a single wavefront inside overlap3's `mid`
    template <class F>
    __device__ __forceinline__ void sub(F &&f)
    {
        f(lane_id() & (WAVE - 1));
        wave_fence();
    }
    // ---- values handed from lane to lane between consecutive seq phases (registers, no LDS) ----
    // share(): publish this lane's value for the next phase (a register stays a register here);
    // gather(j): the value lane j published; shl6 / shr6: the value of lane + 6 / lane - 6
    // (DPP row shifts, rows of 16 lanes; used by lanes < 12 only).
    __device__ __forceinline__ static void share(double *, int, double) {}
    __device__ __forceinline__ static double gather(const double *, int j, double mine) { return row_lane(mine, j); }
    // entry `idx` of a small LDS array whose 16-byte item l lane l of this wavefront has just read (`mine` = the half
    // holding the entry, `src` = idx / 2): a scalar here; the host executor reads the array
    __device__ __forceinline__ static double lane_value(const double *, int, double mine, int src) { return row_lane(mine, src); }
    __device__ __forceinline__ static double shl6(const double *, int, double mine) { return dpp<0x106>(mine); }
    __device__ __forceinline__ static double shr6(const double *, int, double mine) { return dpp<0x116>(mine); }
    // ---- reductions over the NT lanes of a simulation ------------------------------------
    // put_*: called by every lane at the end of a par phase; the wavefront reduces its 64 values
    // with DPP row operations (no LDS round trips) and leaves one partial per wavefront in r[].
    // get_*: after the phase barrier, combines the NWV partials (same order in every lane).
    template <int CTRL>
    __device__ __forceinline__ static double dpp(double v)
    {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    __device__ __forceinline__ static double row_lane(double v, int l)
    {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    }
    template <class Op>
    __device__ __forceinline__ static double wave_reduce(double v, Op op)
    {
        v = op(v, dpp<0xB1>(v));    // quad_perm [1,0,3,2]
        v = op(v, dpp<0x4E>(v));    // quad_perm [2,3,0,1]
        v = op(v, dpp<0x141>(v));   // row_half_mirror
        v = op(v, dpp<0x140>(v));   // row_mirror: every lane of a row of 16 holds the row result
        return op(op(row_lane(v, 0), row_lane(v, 16)), op(row_lane(v, 32), row_lane(v, 48)));
    }
    struct OpSum { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
    struct OpMax { __device__ __forceinline__ double operator()(double a, double b) const { return fmax(a, b); } };
    struct OpMin { __device__ __forceinline__ double operator()(double a, double b) const { return fmin(a, b); } };
    template <class Op>
    __device__ __forceinline__ static void put(double *r, int lane, double v, Op op)
    {
        const double t = wave_reduce(v, op);
        if ((lane & (WAVE - 1)) == 0) r[lane >> 6] = t;
    }
    __device__ __forceinline__ static void put_sum(double *r, int lane, double v) { put(r, lane, v, OpSum()); }
    __device__ __forceinline__ static void put_max(double *r, int lane, double v) { put(r, lane, v, OpMax()); }
    __device__ __forceinline__ static void put_min(double *r, int lane, double v) { put(r, lane, v, OpMin()); }
    template <class Op>
    __device__ __forceinline__ static double get(const double *r, Op op)
    {
        double tot = r[0];
#pragma unroll
        for (int w = 1; w < NWV; w++) tot = op(tot, r[w]);
        return uni(tot);
    }
    // single-wavefront variants (inside overlap3's `mid`): one result in r[0]
    __device__ __forceinline__ static void put1_sum(double *r, int lane, double v) { const double t = wave_reduce(v, OpSum()); if (lane == 0) r[0] = t; }
    __device__ __forceinline__ static void put1_min(double *r, int lane, double v) { const double t = wave_reduce(v, OpMin()); if (lane == 0) r[0] = t; }
    __device__ __forceinline__ static double get1(const double *r) { return uni(r[0]); }
    __device__ __forceinline__ static double get_sum(const double *r) { return get(r, OpSum()); }
    // per-wavefront partial sums (put_sum leaves exactly those) and the sum over wavefronts [w0, w0 + n)
    __device__ __forceinline__ static void put_wsum(double *r, int lane, double v) { put(r, lane, v, OpSum()); }
    __device__ __forceinline__ static double get_sum_range(const double *r, int w0, int n)
    {
        double tot = r[w0];
        for (int w = 1; w < n; w++) tot += r[w0 + w];
        return uni(tot);
    }
    __device__ __forceinline__ static double get_max(const double *r) { return get(r, OpMax()); }
    __device__ __forceinline__ static double get_min(const double *r) { return get(r, OpMin()); }
    // constant 100 MHz counter (s_memrealtime)
    __device__ __forceinline__ double clock() { return (double)wall_clock64() * 1e-8; }
};

// WPE = 1: one wavefront per SIMD owns the whole 512-entry register file (one simulation per CU: batch <= #CUs, and
// the 1 / 2 / 8-wavefront geometries).  WPE = 2: 256 registers, two 4-wavefront simulations resident per CU -- the
// geometry of batches beyond one simulation per CU (measured at batch 4096, N = 100: 540 k steps/s against 468 k for
// two 2-wavefront simulations per CU at 512 registers).
template <int NWV, int WPE = 1>
__global__ __launch_bounds__(WAVE *NWV, WPE) void mpc_rollout_kernel(Problem pb, Robot rb, const InstParams *__restrict__ params,
                                                           double *ws_base, size_t ws_stride, Outputs out, int step0,
                                                           int step1, int pool_doubles)
{
    const int inst = blockIdx.x;
    if (inst >= pb.batch) return;
    DevExec<NWV, WPE> ex;
    load_constants(ex, params + inst, &rb);
    Ctx c{&pb, ws_carve(ws_base + (size_t)inst * ws_stride, pb.N), pool_doubles, pb.N};
    Engine<DevExec<NWV, WPE>> eng(ex, c);
    eng.rollout(out, inst, step0, step1);
}

// THROUGHPUT engine (mpc_stream.h): one wavefront = one workgroup = one simulation, MPCB_STREAM_WPE wavefronts per SIMD.
#ifndef MPCB_STREAM_WPE
#define MPCB_STREAM_WPE 2
#endif
// Two launch shapes:
//   * one workgroup per simulation, all of [step0, step1) (queue == nullptr);
//   * WORK QUEUE (batch larger than the wavefronts resident at once): the grid is the resident set, every wavefront
//     pops items (simulation, chunk of `chunk_steps` closed-loop steps) off an atomic counter, chunk-major -- all
//     simulations advance together, fast wavefronts take more items, and the launch ends within one chunk of the
//     balanced time instead of with the slowest pair of whole simulations.  Chunk c of a simulation follows chunk c-1,
//     which may have run on any CU: the finishing wavefront releases (stores drained, agent-scope release, progress
//     counter), the next one polls the counter and acquires (MI355X_MICROARCH.md, hand-off recipe).  An item's
//     predecessor was popped earlier, so it is running or done: no wavefront ever waits for work nobody has taken.  A
//     bounded poll (`timeout_ticks` of the 100 MHz clock, queue_timeout_ticks below) turns a broken hand-off into an error flag, never a hang.
//   queue[0] item counter, queue[1] error flag, queue[2 + i] chunks of simulation i done in this launch.
// The bound is a bug guard, not a schedule: a waiting item's predecessor is always running, so it is sized from the WORK LIMIT of
// one chunk (queue_timeout_ticks below), never from typical times -- a legitimately slow chunk (full SQP at a long horizon, every
// QP running to qp_solver_iter_max) must not trip it and take the whole bucket down.
template <class FT>
__global__ __launch_bounds__(WAVE, MPCB_STREAM_WPE) void mpc_stream_kernel(Problem pb, const Robot *__restrict__ rbd, const InstParams *__restrict__ params,
                                                                           double *ws_base, size_t ws_stride, Outputs out, int step0, int step1,
                                                                           int *queue, int chunk_steps, long long timeout_ticks)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const bool queued = queue != nullptr;
    const int n_chunks = queued ? (step1 - step0 + chunk_steps - 1) / chunk_steps : 1;
    const int n_items = n_chunks * pb.batch;
    for (;;) {
        int inst = blockIdx.x, s0 = step0, s1 = step1, c = 0;
        if (queued) {
            int q = 0;
            if (threadIdx.x == 0) q = __hip_atomic_fetch_add(&queue[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q = __builtin_amdgcn_readfirstlane(q);
            if (q >= n_items) break;
            c = q / pb.batch;
            inst = q - c * pb.batch;
            s0 = step0 + c * chunk_steps;
            s1 = s0 + chunk_steps < step1 ? s0 + chunk_steps : step1;
            if (c > 0) {
                int ok = 1;
                if (threadIdx.x == 0) {
                    const long long t_end = (long long)wall_clock64() + timeout_ticks;
                    while (__hip_atomic_load(&queue[2 + inst], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c) {
                        if ((long long)wall_clock64() > t_end || __hip_atomic_load(&queue[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            __hip_atomic_store(&queue[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(32);
                    }
                }
                ok = __builtin_amdgcn_readfirstlane(ok);
                if (!ok) break;                                           // error flag set: every wavefront drains out
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // this CU's L1 may hold lines of an earlier chunk
                __builtin_amdgcn_s_waitcnt(0);
            }
        } else if (inst >= pb.batch) {
            break;
        }
        se::rollout<FT>(pb, params, rbd, ws_base, ws_stride, out, inst, s0, s1);
        if (!queued) break;
        __builtin_amdgcn_s_waitcnt(0);                                    // every store of this chunk has been acknowledged
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_s_waitcnt(0);
        if (threadIdx.x == 0) __hip_atomic_store(&queue[2 + inst], c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
}

// Per-simulation summary of Simulator.get_summary (simulator.py:509-547) from the device logs: one wavefront per
// simulation streams its error rows and per-step statistics once (HBM-bound, coalesced).  out[MPCB_NSUMMARY]:
//   [0..4] rmse e1..e5  [5..9] itse e1..e5  [10] weighted_rmse  [11] total_sqp_iterations  [12] avg_sqp_iterations
//   [13] num_failures  [14] max_kkt_residual  [15] total_solver_time  [16] avg_mpc_time  [17] avg_solver_time
//   [18] avg_integration_time  [19] total_computation_time  [20] total_qp_iterations  [21..23] reserved
__global__ __launch_bounds__(WAVE) void mpc_summary_kernel(int batch, int Nsim, const InstParams *__restrict__ params, Outputs o,
                                                           double *__restrict__ summary)
{
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= batch) return;
    const int T1 = Nsim + 1;
    const double dt = params[inst].dt;
    double se[5] = {0, 0, 0, 0, 0}, st[5] = {0, 0, 0, 0, 0};
    for (int c = lane; c < T1; c += WAVE) {
        const double tk = c * dt;                                    // simulator.py:367
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double e = o.errors[((size_t)inst * 7 + j) * T1 + c];
            se[j] += e * e;
            st[j] += tk * e * e;                                     // :368
        }
    }
    double sq = 0, qp = 0, fail = 0, kkt = 0, tsol = 0, tpl = 0, anynan = 0;   // (np.max propagates NaN, fmax drops it: flag)
    for (int i = lane; i < Nsim; i += WAVE) {
        const size_t k = (size_t)inst * Nsim + i;
        sq += o.sqp_iter[k]; qp += o.qp_iter[k]; fail += o.status[k] != 0 ? 1.0 : 0.0;
        tsol += o.solver_time[k]; tpl += o.plant_time[k];
        const double *r = o.residuals + k * 4;
        kkt = fmax(kkt, fmax(fmax(r[0], r[1]), fmax(r[2], r[3])));    // :402
        anynan += (r[0] != r[0] || r[1] != r[1] || r[2] != r[2] || r[3] != r[3]) ? 1.0 : 0.0;
    }
    using X = DevExec<1>;
    double wsum = 0.0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        se[j] = X::wave_reduce(se[j], X::OpSum());
        st[j] = X::wave_reduce(st[j], X::OpSum());
        wsum += params[inst].w_task[j] * se[j];                      // :383-384 (weights 50 each)
    }
    sq = X::wave_reduce(sq, X::OpSum()); qp = X::wave_reduce(qp, X::OpSum()); fail = X::wave_reduce(fail, X::OpSum());
    tsol = X::wave_reduce(tsol, X::OpSum()); tpl = X::wave_reduce(tpl, X::OpSum()); kkt = X::wave_reduce(kkt, X::OpMax());
    anynan = X::wave_reduce(anynan, X::OpSum());
    if (anynan > 0) kkt = __builtin_nan("");     // a diverged simulation (NaN residuals) must show in max_kkt_residual, as in the reference
    if (lane == 0) {
        double *s = summary + (size_t)inst * MPCB_NSUMMARY;
        for (int j = 0; j < 5; j++) { s[j] = sqrt(se[j] / T1); s[5 + j] = st[j] * dt; }
        s[10] = sqrt(wsum / T1);
        s[11] = sq; s[12] = sq / Nsim; s[13] = fail; s[14] = kkt; s[15] = tsol;
        // mpc_time (simulator.py:209-214) = the device time of the solve, integration_time (:224-226) = plant step + logging
        s[16] = tsol / Nsim; s[17] = tsol / Nsim; s[18] = tpl / Nsim; s[19] = tsol + tpl;
        s[20] = qp; s[21] = s[22] = s[23] = 0.0;
    }
}

// Diagnostic entry (tests): the device linearisation task_lin on n points, one lane each.
__global__ void mpc_debug_task_lin_kernel(int n, Robot rb, const InstParams *__restrict__ params, const double *__restrict__ x,
                                          double *__restrict__ rec)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xx[12], out[W2_LIN];
    for (int j = 0; j < 12; j++) xx[j] = x[(size_t)i * 12 + j];
    for (int j = 0; j < W2_LIN; j++) out[j] = 0.0;
    task_lin<true>(rb, params[i], xx, xx + 6, out);
    for (int j = 0; j < W2_LIN; j++) rec[(size_t)i * W2_LIN + j] = out[j];
}

// ------------------------------------------------------------------------------------ host
struct mpcb_handle {
    int device = -1;
    std::string err;
    Problem pb{};
    Robot rb{};
    bool ready = false;
    int next_step = 0;
    InstParams *d_params = nullptr;
    Robot *d_rb = nullptr;     // robot record in device memory (behind the parameter records)
    size_t params_cap = 0;
    double *d_ws = nullptr;
    size_t ws_cap = 0;
    size_t ws_stride = 0;
    int pool_doubles = POOL_DEFAULT_DOUBLES;
    int num_cus = 256;
    int waves_per_sim = 4;
    int wpe = 1;               // latency engine: kernel variant compiled for this many wavefronts per SIMD
    int engine = 0;            // 0: latency engine (mpc_core.h), 1: throughput engine (mpc_stream.h)
    int *d_queue = nullptr;    // throughput engine, work-queue launches: counter, error flag, per-simulation progress
    size_t queue_cap = 0;
    bool queue_used = false;   // the last launch went through the work queue (mpcb_sync checks its error flag)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t last_stream = nullptr;
    bool timed = false;
};

static int fail(mpcb_handle *h, int code, const char *what, hipError_t e = hipSuccess)
{
    if (h) {
        h->err = what;
        if (e != hipSuccess) { h->err += ": "; h->err += hipGetErrorString(e); }
    }
    return code;
}
#define HIPCHK(h, call)                                                   \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return fail((h), MPCB_EHIP, #call, e_);      \
    } while (0)

static int check_problem(mpcb_handle *h, const mpcb_problem *p)
{
    if (!p) return fail(h, MPCB_EINVAL, "problem is NULL");
    if (p->batch < 1) return fail(h, MPCB_EINVAL, "batch must be >= 1");
    if (p->N < 1 || p->N > 4096) return fail(h, MPCB_EINVAL, "prediction horizon N out of range [1,4096]");
    if (p->Nsim < 1) return fail(h, MPCB_EINVAL, "Nsim must be >= 1");
    if (p->solver_type != MPCB_SOLVER_SQP && p->solver_type != MPCB_SOLVER_SQP_RTI)
        return fail(h, MPCB_EINVAL, "solver_type must be MPCB_SOLVER_SQP or MPCB_SOLVER_SQP_RTI");
    if (p->max_iter < 1 || p->qp_iter_max < 1) return fail(h, MPCB_EINVAL, "iteration limits must be >= 1");
    if (p->precision != MPCB_PRECISION_FP64 && p->precision != MPCB_PRECISION_FP32_RICCATI)
        return fail(h, MPCB_EINVAL, "precision must be MPCB_PRECISION_FP64 or MPCB_PRECISION_FP32_RICCATI");
    if (p->precision == MPCB_PRECISION_FP32_RICCATI && p->solver_type != MPCB_SOLVER_SQP_RTI)
        return fail(h, MPCB_EINVAL, "the fp32 Riccati leg is validated for SQP_RTI only");
    return MPCB_OK;
}

extern "C" {

int mpcb_version(void) { return MPCB_VERSION; }

int mpcb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mpcb_create(mpcb_handle **out, int device)
{
    if (!out) return MPCB_EINVAL;
    *out = nullptr;
    int n = mpcb_device_count();
    if (n <= 0 || device < 0 || device >= n) return MPCB_ENODEV;
    mpcb_handle *h = new (std::nothrow) mpcb_handle;
    if (!h) return MPCB_ENOMEM;
    h->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            h->num_cus = prop.multiProcessorCount;
    }
    if (hipSetDevice(device) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess ||
        hipEventCreate(&h->ev1) != hipSuccess) {
        delete h;
        return MPCB_EHIP;
    }
    *out = h;
    return MPCB_OK;
}

void mpcb_destroy(mpcb_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_params) (void)hipFree(h->d_params);
    if (h->d_ws) (void)hipFree(h->d_ws);
    if (h->d_queue) (void)hipFree(h->d_queue);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
}

const char *mpcb_last_error(const mpcb_handle *h) { return h ? h->err.c_str() : "invalid handle"; }

// which kernel family runs `p` (see include/mpcbatch.h); `ragged`: the simulations have different horizons
static int pick_engine(const mpcb_problem *p, bool ragged = false)
{
    if (p->precision == MPCB_PRECISION_FP32_RICCATI || ragged) return 1;
    // Full SQP work per step is heavy-tailed while the closed loop settles (a few simulations run into
    // nlp_solver_max_iter): short or small runs end with those simulations' sequential chains, where the latency engine is
    // faster per simulation; long runs of large batches are throughput work again, and the work-queue launch balances them.
    // Measured at N = 100 (profiles/r03_engine_sweep.txt), throughput engine queued / latency engine (two simulations per CU,
    // register-resident sweeps), steps/s: batch 4096 x 600 steps 321 k / 289 k, x 100 steps 72 k / 80 k; batch 3072 x 600:
    // 265 k / 265 k; 2560 x 600: 229 k / 256 k.  (Round 2, before those sweeps: 299 k / 214 k, 67 k / 59 k, 245 k / 195 k.)
    int e = 0;
    if (p->solver_type == MPCB_SOLVER_SQP_RTI) e = p->batch >= MPCB_STREAM_MIN_BATCH ? 1 : 0;
    else e = p->batch >= MPCB_STREAM_MIN_BATCH_SQP && p->Nsim >= MPCB_STREAM_MIN_STEPS_SQP ? 1 : 0;
    if (const char *env = getenv("MPCB_ENGINE")) {
        if (!strcmp(env, "stream")) e = 1;
        else if (!strcmp(env, "latency")) e = 0;
    }
    return e;
}
static size_t ws_doubles_for(const mpcb_problem *p, bool ragged = false)
{
    if (pick_engine(p, ragged) == 0) return ws_doubles_per_instance(p->N);
    const bool sqp = p->solver_type == MPCB_SOLVER_SQP;
    return p->precision == MPCB_PRECISION_FP32_RICCATI ? se::sws_doubles_per_instance<float>(p->N, sqp) : se::sws_doubles_per_instance<double>(p->N, sqp);
}

size_t mpcb_workspace_bytes(const mpcb_problem *p)
{
    if (!p || p->N < 1 || p->batch < 1) return 0;
    return (size_t)p->batch * ws_doubles_for(p) * sizeof(double);
}

size_t mpcb_result_bytes_per_sim(const mpcb_problem *p)
{
    if (!p) return 0;
    const size_t T1 = (size_t)p->Nsim + 1, S = (size_t)p->Nsim;
    return (12 + 6 + 12 + 3 + 6 + 7) * T1 * sizeof(double) + 3 * S * sizeof(int) + (4 + 1 + 1 + 1) * S * sizeof(double);
}

int mpcb_setup(mpcb_handle *h, const mpcb_problem *p, const double *params_host, const double *robot_host)
{
    if (!h) return MPCB_EINVAL;
    int rc = check_problem(h, p);
    if (rc) return rc;
    if (!params_host || !robot_host) return fail(h, MPCB_EINVAL, "params/robot pointer is NULL");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<InstParams> packed((size_t)p->batch);
    bool ragged = false;
    for (int i = 0; i < p->batch; i++) {
        const double *pp = params_host + (size_t)i * MPCB_NPARAM;
        if (pp[65] != 0.0) {
            if (!(pp[65] >= 1.0 && pp[65] <= (double)p->N) || pp[65] != std::floor(pp[65]))
                return fail(h, MPCB_EINVAL, "per-simulation horizon (parameter [65]) must be an integer in [1, N]");
            if ((int)pp[65] != p->N) ragged = true;
        }
        if (!(pp[0] > 0.0)) return fail(h, MPCB_EINVAL, "dt must be positive");
        for (int j = 0; j < 6; j++)
            if (!(pp[8 + j] > 0.0)) return fail(h, MPCB_EINVAL, "wcv must be positive");
        for (int j = 0; j < MPCB_NPARAM; j++)
            if (std::isnan(pp[j])) return fail(h, MPCB_EINVAL, "NaN in parameter record");
        pack_inst_params(pp, &packed[(size_t)i]);
    }
    std::memcpy(&h->rb, robot_host, sizeof(Robot));
    std::memcpy(&h->pb, p, sizeof(Problem));
    const size_t pbytes = packed.size() * sizeof(InstParams);
    if (pbytes + sizeof(Robot) > h->params_cap) {   // the robot record rides behind the parameter records (throughput engine reads it from memory)
        if (h->d_params) (void)hipFree(h->d_params);
        h->d_params = nullptr; h->params_cap = 0;
        if (hipMalloc((void **)&h->d_params, pbytes + sizeof(Robot)) != hipSuccess) return fail(h, MPCB_ENOMEM, "hipMalloc(params)");
        h->params_cap = pbytes + sizeof(Robot);
    }
    HIPCHK(h, hipMemcpy(h->d_params, packed.data(), pbytes, hipMemcpyHostToDevice));
    h->d_rb = reinterpret_cast<Robot *>(reinterpret_cast<char *>(h->d_params) + pbytes);
    HIPCHK(h, hipMemcpy(h->d_rb, &h->rb, sizeof(Robot), hipMemcpyHostToDevice));
    h->engine = pick_engine(p, ragged);
    h->ws_stride = ws_doubles_for(p, ragged);
    const size_t wbytes = (size_t)p->batch * h->ws_stride * sizeof(double);
    if (wbytes > h->ws_cap) {
        if (h->d_ws) (void)hipFree(h->d_ws);
        h->d_ws = nullptr; h->ws_cap = 0;
        if (hipMalloc((void **)&h->d_ws, wbytes) != hipSuccess) return fail(h, MPCB_ENOMEM, "hipMalloc(workspace)");
        h->ws_cap = wbytes;
    }
    // LDS chunk pool: the whole 160 KiB of a CU is shared by the waves resident on it, so size the
    // pool for the number of simulations per CU this batch implies (1 for batch <= #CUs).
    {
        // More than a few resident simulations per CU only shrinks the chunks (more, smaller bursts);
        // beyond that the grid simply queues.  MPCB_SIMS_PER_CU overrides the residency target.
        int wpc = (p->batch + h->num_cus - 1) / h->num_cus;
        if (wpc > 2) wpc = 2;
        // (Measured, N = 100, 600 steps, profiles/r03_engine_sweep.txt: batch 512 as two simulations per CU at half the pool on the
        // streaming path 730 k steps/s; as one per CU with the LDS-resident factor, the grid running in two rounds, 640 k: two
        // factorisation chains side by side on a CU beat one faster one, so the residency target stays 2 beyond #CUs.)
        if (const char *e2 = getenv("MPCB_SIMS_PER_CU")) { const int v = atoi(e2); if (v >= 1 && v <= 8) wpc = v; }
        const int lds_total = 160 * 1024, fixed = (int)sizeof(Smem) + 64;
        int bytes = lds_total / (wpc < 1 ? 1 : wpc) - fixed;
        if (bytes > POOL_DEFAULT_DOUBLES * 8) bytes = POOL_DEFAULT_DOUBLES * 8;
        if (bytes < POOL_MIN_DOUBLES * 8) bytes = POOL_MIN_DOUBLES * 8;
        h->pool_doubles = (bytes / 16) * 2;
        // wavefronts per simulation: the four SIMDs of a CU are otherwise idle at one simulation per CU
        const char *env = getenv("MPCB_WAVES_PER_SIM");
        // one simulation per CU: 4 wavefronts, 512 registers; beyond: two 4-wavefront simulations per CU at 256 registers
        int nw = wpc <= 2 ? 4 : 1;
        int wpe = wpc == 2 ? 2 : 1;
        // one simulation per CU AND the factor LDS-resident (Engine::resident_ok): every pass but the factorisation sweep is item- or
        // chunk-parallel then, and a second wavefront per SIMD buys issue slots (665 k vs 651 k steps/s on configs[1]); the streaming
        // path's role layout prefers 4 (round 1).
        {
            // (the device's own predicates, mpc_layout.h; 16 lane groups at four and at eight wavefronts)
            // (measured: N = 100 665 k vs 651 k; N = 50 1.045 M vs 1.054 M; N = 20 1.59 M vs 1.68 M -- short horizons have too few items)
            const bool resident = lay_resident_ok(p->N, h->pool_doubles, 16);
            // longer horizons at one simulation per CU run the same sweeps one LDS segment at a time (Engine::segment_ok): eight
            // wavefronts there as well (batch 256, N = 200: 244.8 k vs 234.5 k steps/s; N = 300: 144.5 k vs 137.9 k)
            const bool segments = lay_segment_ok(p->N, h->pool_doubles, 16);
            // Round 4, with one factorisation per step (the fast path) the item phases are a fifth of a step instead of two fifths, and
            // the recursion wavefront alone on its SIMD matters more (profiles/r04_geometry.txt, batch 256, 4 / 8 wavefronts, ms per
            // launch): N = 80 83.8 / 85.9, N = 100 99.9-100.6 / 101.6-102.4, N = 125 90.4 / 88.0 (400 steps), N = 200 116.4 / 115.4,
            // N = 300 139.0 / 135.1 -- eight from N = 112 on (80 in round 3).  End of round 4: those 256-register builds had been paying for
            // scratch spills around the library sincos in the NLP pass; with the joint-angle sincos of mpc_kin.h (no spills) eight win
            // from N = 80 again (profiles/r04_geometry2.txt: N = 50 58.7 / 59.7 ms, N = 80 84.4 / 83.7, N = 100 97.5 / 95.3,
            // N = 112 111.3 / 106.6)
            if (wpc == 1 && p->N >= 80 && (resident || segments)) nw = 8;
        }
        if (env && (atoi(env) == 1 || atoi(env) == 2 || atoi(env) == 4 || atoi(env) == 8)) { nw = atoi(env); wpe = 1; }
        if (const char *e3 = getenv("MPCB_WPE")) { if (atoi(e3) == 2 && nw == 4) wpe = 2; else if (atoi(e3) == 1) wpe = 1; }
        h->waves_per_sim = nw;
        h->wpe = wpe;
        if (h->engine == 1) { h->waves_per_sim = 1; h->pool_doubles = 0; }   // throughput engine: one wavefront, static LDS only
    }
    h->ready = true;
    h->next_step = 0;
    h->timed = false;
    return MPCB_OK;
}

// Hand-off timeout of the work-queue launch in ticks of the 100 MHz clock: the most work one chunk of one simulation can be
// (chunk_steps x SQP iterations x (interior-point iterations + 1) x (N + 1) stage-iterations), priced at 20 us per stage-iteration --
// more than 10x what a wavefront sharing its SIMD takes, profiling and eight buckets in flight included -- times 4, plus 30 s.
// MPCB_QUEUE_TIMEOUT_S overrides it (seconds).
static long long queue_timeout_ticks(const Problem &pb, int chunk_steps)
{
    if (const char *e = getenv("MPCB_QUEUE_TIMEOUT_S")) { const double v = atof(e); if (v > 0) return (long long)(v * 1e8); }
    const double sqp = pb.solver_type == MPCB_SOLVER_SQP ? (double)(pb.max_iter > 1 ? pb.max_iter : 1) : 1.0;
    const double work_s = (double)(chunk_steps > 1 ? chunk_steps : 1) * sqp * (double)(pb.qp_iter_max + 1) * (double)(pb.N + 1) * 20e-6;
    return (long long)((4.0 * work_s + 30.0) * 1e8);
}

int mpcb_rollout(mpcb_handle *h, int step0, int step1, const mpcb_result *o, void *stream)
{
    if (!h) return MPCB_EINVAL;
    if (!h->ready) return fail(h, MPCB_ESTATE, "mpcb_rollout before mpcb_setup");
    if (!o) return fail(h, MPCB_EINVAL, "result pointer is NULL");
    if (step0 != h->next_step && step0 != 0)
        return fail(h, MPCB_ESTATE, "step0 must continue the previous rollout (or be 0 to restart)");
    if (step1 <= step0 || step1 > h->pb.Nsim) return fail(h, MPCB_EINVAL, "step range out of bounds");
    if (!o->z || !o->u || !o->ee_pose || !o->ee_rpy || !o->ee_vel || !o->status || !o->sqp_iter || !o->qp_iter ||
        !o->residuals || !o->cost || !o->solver_time || !o->errors || !o->plant_time)
        return fail(h, MPCB_EINVAL, "every result array must be provided");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    Outputs out;
    std::memcpy(&out, o, sizeof out);
    if (h->engine == 1) {
        // at least as many simulations as resident wavefronts: work queue over (simulation, chunk of steps) items.
        // Measured at N=100, 600 steps (profiles/r02_work_queue.txt): batch 4096 890 -> 968 k steps/s, 2560 668 -> 939 k,
        // 2048 876 -> 909 k; chunks of 8..20 steps are equivalent, 1 step costs 2.5 %, 50 steps 1.4 %.  Full SQP: only for
        // runs of >= 100 steps (see pick_engine: the settling phase is heavy-tailed, hand-off waits then add to the chains).
        const int slots = h->num_cus * 4 * MPCB_STREAM_WPE;
        int chunk = 10;
        bool chunk_forced = false;                                   // (an explicit MPCB_STREAM_CHUNK also applies to full SQP)
        if (const char *e = getenv("MPCB_STREAM_CHUNK")) { chunk = atoi(e); chunk_forced = true; }
        int nslots = slots;
        if (const char *e = getenv("MPCB_STREAM_SLOTS")) { if (atoi(e) > 0) nslots = atoi(e); }   // (tests: force hand-offs on small batches)
        const bool queued = chunk > 0 && h->pb.batch >= nslots && (h->pb.solver_type == 1 || chunk_forced || h->pb.Nsim >= 100);
        int *queue = nullptr;
        if (queued) {
            const size_t need = (size_t)(2 + h->pb.batch) * sizeof(int);
            if (need > h->queue_cap) {
                if (h->d_queue) (void)hipFree(h->d_queue);
                h->d_queue = nullptr; h->queue_cap = 0;
                if (hipMalloc((void **)&h->d_queue, need) != hipSuccess) return fail(h, MPCB_ENOMEM, "work queue allocation failed");
                h->queue_cap = need;
            }
            HIPCHK(h, hipMemsetAsync(h->d_queue, 0, need, s));
            queue = h->d_queue;
        }
        h->queue_used = queued;
        const long long qticks = queue_timeout_ticks(h->pb, chunk);
        HIPCHK(h, hipEventRecord(h->ev0, s));
        const dim3 sgrid((unsigned)(queued ? nslots : h->pb.batch));
        if (h->pb.precision == MPCB_PRECISION_FP32_RICCATI)
            hipLaunchKernelGGL(mpc_stream_kernel<float>, sgrid, dim3(WAVE), 0, s, h->pb, h->d_rb, h->d_params, h->d_ws, h->ws_stride, out,
                               step0, step1, queue, chunk, qticks);
        else
            hipLaunchKernelGGL(mpc_stream_kernel<double>, sgrid, dim3(WAVE), 0, s, h->pb, h->d_rb, h->d_params, h->d_ws, h->ws_stride, out,
                               step0, step1, queue, chunk, qticks);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipEventRecord(h->ev1, s));
        h->last_stream = s;
        h->timed = true;
        h->next_step = step1;
        return MPCB_OK;
    }
    const size_t lds = (size_t)h->pool_doubles * sizeof(double);
    {
        // the dynamic-LDS ceiling is a process-wide attribute of the kernel, not of this handle: always raise it to
        // the largest pool any handle can ask for, right before the launch
        static const int max_lds = (160 * 1024 - (int)sizeof(Smem) - 64) / 16 * 16;
        const void *fn = h->waves_per_sim == 8 ? (const void *)mpc_rollout_kernel<8>
                       : h->waves_per_sim == 4 ? (h->wpe == 2 ? (const void *)mpc_rollout_kernel<4, 2> : (const void *)mpc_rollout_kernel<4>)
                       : h->waves_per_sim == 2 ? (const void *)mpc_rollout_kernel<2> : (const void *)mpc_rollout_kernel<1>;
        HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds));
    }
    HIPCHK(h, hipEventRecord(h->ev0, s));
    const dim3 grid((unsigned)h->pb.batch);
    if (h->waves_per_sim == 8)
        hipLaunchKernelGGL(mpc_rollout_kernel<8>, grid, dim3(WAVE * 8), lds, s, h->pb, h->rb, h->d_params, h->d_ws,
                           h->ws_stride, out, step0, step1, h->pool_doubles);
    else if (h->waves_per_sim == 4 && h->wpe == 2)
        hipLaunchKernelGGL((mpc_rollout_kernel<4, 2>), grid, dim3(WAVE * 4), lds, s, h->pb, h->rb, h->d_params, h->d_ws,
                           h->ws_stride, out, step0, step1, h->pool_doubles);
    else if (h->waves_per_sim == 4)
        hipLaunchKernelGGL(mpc_rollout_kernel<4>, grid, dim3(WAVE * 4), lds, s, h->pb, h->rb, h->d_params, h->d_ws,
                           h->ws_stride, out, step0, step1, h->pool_doubles);
    else if (h->waves_per_sim == 2)
        hipLaunchKernelGGL(mpc_rollout_kernel<2>, grid, dim3(WAVE * 2), lds, s, h->pb, h->rb, h->d_params, h->d_ws,
                           h->ws_stride, out, step0, step1, h->pool_doubles);
    else
        hipLaunchKernelGGL(mpc_rollout_kernel<1>, grid, dim3(WAVE), lds, s, h->pb, h->rb, h->d_params, h->d_ws,
                           h->ws_stride, out, step0, step1, h->pool_doubles);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, s));
    h->last_stream = s;
    h->timed = true;
    h->next_step = step1;
    return MPCB_OK;
}

int mpcb_sync(mpcb_handle *h)
{
    if (!h) return MPCB_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->last_stream));
    if (h->engine == 1 && h->queue_used && h->d_queue) {
        int flag = 0;
        HIPCHK(h, hipMemcpy(&flag, h->d_queue + 1, sizeof flag, hipMemcpyDeviceToHost));
        if (flag) return fail(h, MPCB_EHIP, "work-queue hand-off timed out (results of this launch are incomplete)");
    }
    return MPCB_OK;
}

int mpcb_last_kernel_ms(mpcb_handle *h, float *ms)
{
    if (!h || !ms) return MPCB_EINVAL;
    if (!h->timed) return fail(h, MPCB_ESTATE, "no rollout has been launched");
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return MPCB_OK;
}

int mpcb_kernel_info(mpcb_handle *h, int *vgprs, int *sgprs, int *lds_bytes, int *scratch_bytes)
{
    if (!h) return MPCB_EINVAL;
    hipFuncAttributes a;
    if (h->engine == 1) {
        HIPCHK(h, hipFuncGetAttributes(&a, h->pb.precision == MPCB_PRECISION_FP32_RICCATI ? (const void *)mpc_stream_kernel<float>
                                                                                          : (const void *)mpc_stream_kernel<double>));
    } else
    HIPCHK(h, hipFuncGetAttributes(&a, h->waves_per_sim == 8 ? (const void *)mpc_rollout_kernel<8>
                                       : h->waves_per_sim == 4 ? (h->wpe == 2 ? (const void *)mpc_rollout_kernel<4, 2> : (const void *)mpc_rollout_kernel<4>)
                                       : h->waves_per_sim == 2 ? (const void *)mpc_rollout_kernel<2>
                                                               : (const void *)mpc_rollout_kernel<1>));
    if (vgprs) *vgprs = a.numRegs;
    if (sgprs) *sgprs = 0;
    if (lds_bytes) *lds_bytes = (int)a.sharedSizeBytes;
    if (scratch_bytes) *scratch_bytes = (int)a.localSizeBytes;
    return MPCB_OK;
}

int mpcb_engine(mpcb_handle *h)
{
    if (!h || !h->ready) return MPCB_EINVAL;
    return h->engine;
}

int mpcb_engine_for(const mpcb_problem *p)
{
    if (!p || p->N < 1 || p->batch < 1) return MPCB_EINVAL;
    return pick_engine(p);
}

int mpcb_launch_info(mpcb_handle *h, int *waves_per_sim, int *pool_bytes)
{
    if (!h || !h->ready) return MPCB_EINVAL;
    if (waves_per_sim) *waves_per_sim = h->waves_per_sim;
    if (pool_bytes) *pool_bytes = h->pool_doubles * (int)sizeof(double);
    return MPCB_OK;
}

// Diagnostic builds only (-DMPCB_PROFILE): per-section device seconds accumulated by instance
// `inst`; all zeros in the product build.  Not part of include/mpcbatch.h.
int mpcb_debug_profile(mpcb_handle *h, int inst, double *out16)
{
    if (!h || !out16 || !h->ready || inst < 0 || inst >= h->pb.batch) return MPCB_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    const double *src = h->d_ws + (size_t)inst * h->ws_stride + (h->ws_stride - STATE_DOUBLES) + 32;
    HIPCHK(h, hipMemcpy(out16, src, NPROF * sizeof(double), hipMemcpyDeviceToHost));
    return MPCB_OK;
}

// Diagnostic: copy one simulation's HBM workspace to the host (layout: mpc_layout.h ws_carve).
int mpcb_debug_workspace(mpcb_handle *h, int inst, double *out, size_t n_doubles)
{
    if (!h || !out || !h->ready || inst < 0 || inst >= h->pb.batch) return MPCB_EINVAL;
    if (n_doubles > h->ws_stride) n_doubles = h->ws_stride;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpy(out, h->d_ws + (size_t)inst * h->ws_stride, n_doubles * sizeof(double), hipMemcpyDeviceToHost));
    return MPCB_OK;
}

int mpcb_summary(mpcb_handle *h, const mpcb_result *o, double *summary_dev, void *stream)
{
    if (!h) return MPCB_EINVAL;
    if (!h->ready) return fail(h, MPCB_ESTATE, "mpcb_summary before mpcb_setup");
    if (!o || !summary_dev || !o->errors || !o->status || !o->sqp_iter || !o->qp_iter || !o->residuals || !o->solver_time ||
        !o->plant_time)
        return fail(h, MPCB_EINVAL, "mpcb_summary needs the errors, status, iteration, residual and time arrays");
    HIPCHK(h, hipSetDevice(h->device));
    Outputs out;
    std::memcpy(&out, o, sizeof out);
    hipLaunchKernelGGL(mpc_summary_kernel, dim3((unsigned)h->pb.batch), dim3(WAVE), 0, (hipStream_t)stream, h->pb.batch, h->pb.Nsim,
                       h->d_params, out, summary_dev);
    HIPCHK(h, hipGetLastError());
    return MPCB_OK;
}

// Diagnostic (tests/test_gpu_parity.py): evaluate the device linearisation (task residual r, dg/dq, dg5/dqdot) at
// n points; params_host = n parameter records, x_host = n x 12 [q; qdot], rec_host = n x 60 (layout of a G2 record,
// mpc_layout.h O_R / O_GQ / O_GV).  Not part of include/mpcbatch.h.
int mpcb_debug_task_lin(mpcb_handle *h, int n, const double *params_host, const double *robot_host, const double *x_host,
                        double *rec_host)
{
    if (!h || n < 1 || !params_host || !robot_host || !x_host || !rec_host) return MPCB_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<InstParams> packed((size_t)n);
    for (int i = 0; i < n; i++) pack_inst_params(params_host + (size_t)i * MPCB_NPARAM, &packed[(size_t)i]);
    Robot rb;
    std::memcpy(&rb, robot_host, sizeof rb);
    InstParams *dp = nullptr;
    double *dx = nullptr, *dr = nullptr;
    int rc = MPCB_OK;
    if (hipMalloc((void **)&dp, n * sizeof(InstParams)) != hipSuccess || hipMalloc((void **)&dx, (size_t)n * 12 * 8) != hipSuccess ||
        hipMalloc((void **)&dr, (size_t)n * W2_LIN * 8) != hipSuccess)
        rc = fail(h, MPCB_ENOMEM, "hipMalloc(debug)");
    if (rc == MPCB_OK && (hipMemcpy(dp, packed.data(), n * sizeof(InstParams), hipMemcpyHostToDevice) != hipSuccess ||
                          hipMemcpy(dx, x_host, (size_t)n * 12 * 8, hipMemcpyHostToDevice) != hipSuccess))
        rc = fail(h, MPCB_EHIP, "hipMemcpy(debug in)");
    if (rc == MPCB_OK) {
        hipLaunchKernelGGL(mpc_debug_task_lin_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, n, rb, dp, dx, dr);
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(rec_host, dr, (size_t)n * W2_LIN * 8, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(h, MPCB_EHIP, "debug task_lin kernel");
    }
    if (dp) (void)hipFree(dp);
    if (dx) (void)hipFree(dx);
    if (dr) (void)hipFree(dr);
    return rc;
}

int mpcb_run(mpcb_handle *h, const mpcb_problem *p, const double *params_host, const double *robot_host,
             const mpcb_result *oh)
{
    if (!h) return MPCB_EINVAL;
    if (!oh) return fail(h, MPCB_EINVAL, "result pointer is NULL");
    int rc = mpcb_setup(h, p, params_host, robot_host);
    if (rc) return rc;
    const size_t B = (size_t)p->batch, T1 = (size_t)p->Nsim + 1, S = (size_t)p->Nsim;
    const size_t nd[6] = {12 * T1, 6 * T1, 12 * T1, 3 * T1, 6 * T1, 7 * T1};
    const size_t dbl_total = B * (nd[0] + nd[1] + nd[2] + nd[3] + nd[4] + nd[5] + 7 * S);
    const size_t int_total = B * 3 * S;
    double *dd = nullptr;
    int *di = nullptr;
    if (hipMalloc((void **)&dd, dbl_total * sizeof(double)) != hipSuccess) return fail(h, MPCB_ENOMEM, "hipMalloc(results)");
    if (hipMalloc((void **)&di, int_total * sizeof(int)) != hipSuccess) { (void)hipFree(dd); return fail(h, MPCB_ENOMEM, "hipMalloc(results)"); }
    mpcb_result od;
    double *pd = dd;
    od.z = pd; pd += B * nd[0];
    od.u = pd; pd += B * nd[1];
    od.ee_pose = pd; pd += B * nd[2];
    od.ee_rpy = pd; pd += B * nd[3];
    od.ee_vel = pd; pd += B * nd[4];
    od.errors = pd; pd += B * nd[5];
    od.residuals = pd; pd += B * 4 * S;
    od.cost = pd; pd += B * S;
    od.solver_time = pd; pd += B * S;
    od.plant_time = pd; pd += B * S;
    od.status = di; od.sqp_iter = di + B * S; od.qp_iter = di + 2 * B * S;
    rc = mpcb_rollout(h, 0, p->Nsim, &od, nullptr);
    if (rc == MPCB_OK) rc = mpcb_sync(h);
    if (rc == MPCB_OK) {
        struct { void *dst; const void *src; size_t n; } cp[] = {
            {oh->z, od.z, B * nd[0] * 8}, {oh->u, od.u, B * nd[1] * 8}, {oh->ee_pose, od.ee_pose, B * nd[2] * 8},
            {oh->ee_rpy, od.ee_rpy, B * nd[3] * 8}, {oh->ee_vel, od.ee_vel, B * nd[4] * 8}, {oh->errors, od.errors, B * nd[5] * 8},
            {oh->residuals, od.residuals, B * 4 * S * 8}, {oh->cost, od.cost, B * S * 8},
            {oh->solver_time, od.solver_time, B * S * 8}, {oh->plant_time, od.plant_time, B * S * 8}, {oh->status, od.status, B * S * 4},
            {oh->sqp_iter, od.sqp_iter, B * S * 4}, {oh->qp_iter, od.qp_iter, B * S * 4}};
        for (auto &c : cp) {
            if (!c.dst) continue;
            hipError_t e = hipMemcpy(c.dst, c.src, c.n, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { rc = fail(h, MPCB_EHIP, "hipMemcpy(results)", e); break; }
        }
    }
    (void)hipFree(dd);
    (void)hipFree(di);
    return rc;
}

}  // extern "C"
