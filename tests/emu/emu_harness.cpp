// emu_harness.cpp -- TEST-ONLY host emulation of the wave-cooperative engine.
//
// Instantiates robotic_mpc_amd/csrc/mpc_core.h with an executor that runs the 64 lanes of
// every bulk-synchronous phase one after the other on the CPU.  Purpose: debug the product's
// device code (lane mappings, phase hazards, algebra) against the oracle in a container
// that has no GPU.  It is compiled host-only (hipcc --offload-host-only), is loaded only by
// tests/test_emulation.py, and is NOT linked into libmpcbatch.so: the product has no CPU path.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <vector>
#include <type_traits>

#include "../../robotic_mpc_amd/csrc/mpc_core.h"
#include "../../robotic_mpc_amd/csrc/mpc_pack.h"

using namespace mpcb;

template <int NWV>
struct HostExec {
    static constexpr int NT = WAVE * NWV;
    static constexpr int VGPR_BUDGET = 512;
    Smem *sm_;
    double *pool_;
    Smem &smem() const { return *sm_; }
    double *pool() const { return pool_; }
    static int uni(int v) { return v; }
    static bool uni(bool v) { return v; }
    static double uni(double v) { return v; }
    template <class T>
    static T *uni(T *p) { return p; }
    template <class T>
    struct PerLane {
        T v[WAVE * NWV];
        T &at(int l) { return v[l]; }
    };
    template <class F>
    void par(F &&f)
    {
        for (int l = 0; l < NT; l++) f(l);
    }
    template <class F>
    void wpar(F &&f)     // wave-local phase on every wavefront: the wavefronts are independent, any order is valid
    {
        for (int l = 0; l < NT; l++) f(l);
    }
    static void barrier() {}
    template <class F>
    void seq(F &&f)
    {
        for (int l = 0; l < WAVE; l++) f(l);
    }
    template <class FG, class BG>
    void overlap(FG &&fg, BG &&bg)
    {
        // no concurrency here: the background work must not depend on the recursion (or the
        // other way round), so any order is valid; run it first to catch a forward dependency
        for (int l = 0; l < NT; l++) bg(l, std::integral_constant<int, NT>{});
        fg();
    }
    template <class FG, class MID, class BG>
    void overlap3(FG &&fg, MID &&mid, BG &&bg)
    {
        fg();
        mid();   // mid and bg may follow fg's progress (post / await): fg has finished here
        for (int l = 0; l < NT; l++) bg(l, std::integral_constant<int, NT>{});
    }
    static constexpr int BG_WAVES = NWV;    // overlap3 hands every lane to the background role here
    template <class FG, class MID, class BG>
    void pipeline3(int nwin, FG &&fg, MID &&mid, BG &&bg)
    {
        for (int ci = 0; ci < nwin; ci++)
            overlap3([&]() { fg(ci); }, [&]() { mid(ci); }, [&](int lane, auto nl) { bg(ci, lane, nl); });
    }
    static void post_add(int *flag, int v) { *flag += v; }
    static void post(int *flag, int v) { *flag = v; }
    static void await(int *flag, int v) { if (*flag < v) std::abort(); }
    template <class F>
    void sub(F &&f)
    {
        for (int l = 0; l < WAVE; l++) f(l);
    }
    // lane-to-lane hand-over between seq phases: through the (double-buffered) slot array here
    static void share(double *slot, int lane, double v) { slot[lane] = v; }
    static double gather(const double *slot, int j, double) { return slot[j]; }
    static double lane_value(const double *arr, int idx, double, int) { return arr[idx]; }
    static double shl6(const double *slot, int lane, double) { return lane + 6 < 12 ? slot[lane + 6] : 0.0; }
    static double shr6(const double *slot, int lane, double) { return lane >= 6 && lane < 12 ? slot[lane - 6] : 0.0; }
    // lanes run one after the other here: the slot accumulates in lane order
    static void put_sum(double *r, int lane, double v) { r[0] = lane == 0 ? v : r[0] + v; }
    static void put_max(double *r, int lane, double v) { r[0] = lane == 0 ? v : fmax(r[0], v); }
    static void put_min(double *r, int lane, double v) { r[0] = lane == 0 ? v : fmin(r[0], v); }
    static void put1_sum(double *r, int lane, double v) { put_sum(r, lane, v); }
    static void put1_min(double *r, int lane, double v) { put_min(r, lane, v); }
    static double get1(const double *r) { return r[0]; }
    static double get_sum(const double *r) { return r[0]; }
    static void put_wsum(double *r, int lane, double v) { r[lane >> 6] = (lane & 63) == 0 ? v : r[lane >> 6] + v; }
    static double get_sum_range(const double *r, int w0, int n)
    {
        double tot = r[w0];
        for (int w = 1; w < n; w++) tot += r[w0 + w];
        return tot;
    }
    static double get_max(const double *r) { return r[0]; }
    static double get_min(const double *r) { return r[0]; }
    double clock() { return 0.0; }
};

template <int NWV>
static int emu_run_t(const Problem *pb, const double *robot105, const double *params, Outputs out, int step_chunk,
                     int pool_doubles)
{
    Robot rb;
    std::memcpy(&rb, robot105, sizeof(Robot));
    const size_t wsd = ws_doubles_per_instance(pb->N);
    std::vector<double> ws(wsd);
    for (int inst = 0; inst < pb->batch; inst++) {
        InstParams P;
        pack_inst_params(params + (size_t)inst * MPCB_NPARAM, &P);
        Smem sm;
        std::memset(&sm, 0, sizeof sm);
        if (pool_doubles <= 0) pool_doubles = POOL_DEFAULT_DOUBLES;
        std::vector<double> pool((size_t)pool_doubles + 64, 0.0);
        HostExec<NWV> ex{&sm, pool.data()};
        if (step_chunk <= 0) step_chunk = pb->Nsim;
        for (int s0 = 0; s0 < pb->Nsim; s0 += step_chunk) {
            load_constants(ex, &P, &rb);
            Ctx c{pb, ws_carve(ws.data(), pb->N), pool_doubles, pb->N};
            Engine<HostExec<NWV>> eng(ex, c);
            const int s1 = s0 + step_chunk < pb->Nsim ? s0 + step_chunk : pb->Nsim;
            eng.rollout(out, inst, s0, s1);
        }
    }
    return 0;
}

extern "C" int emu_run(const Problem *pb, const double *robot105, const double *params /* [batch][MPCB_NPARAM] */,
                       double *z, double *u, double *ee_pose, double *ee_rpy, double *ee_vel, int *status,
                       int *sqp_iter, int *qp_iter, double *residuals, double *cost, double *solver_time, double *errors, double *plant_time,
                       int step_chunk, int pool_doubles, int waves)
{
    Outputs out{z, u, ee_pose, ee_rpy, ee_vel, status, sqp_iter, qp_iter, residuals, cost, solver_time, errors, plant_time};
    if (waves == 8) return emu_run_t<8>(pb, robot105, params, out, step_chunk, pool_doubles);
    if (waves == 4) return emu_run_t<4>(pb, robot105, params, out, step_chunk, pool_doubles);
    if (waves == 2) return emu_run_t<2>(pb, robot105, params, out, step_chunk, pool_doubles);
    return emu_run_t<1>(pb, robot105, params, out, step_chunk, pool_doubles);
}

// the product's joint-angle sincos (mpc_kin.h), for tests/test_emulation.py::test_joint_sincos_against_libm
extern "C" void emu_sincos(int n, const double *th, double *sn, double *cs)
{
    for (int i = 0; i < n; i++) sincos_joint(th[i], sn + i, cs + i);
}
