// mpc_core.h -- the batched MPC rollout engine, single source for the gfx950 kernel.
//
// One workgroup of NW wavefronts (NT = 64*NW lanes) = one closed-loop simulation (Simulator.run,
// simulator.py:199-241).  The code is written against a small executor `Ex`:
//   ex.par(f)             bulk-synchronous phase on all NT lanes; a lane only reads what earlier phases
//                         produced and writes entries nobody reads in that phase; ends with the barrier
//   ex.seq(f)             one step of a stage-by-stage recursion: wavefront 0 only, wave-local fence
//   ex.overlap3(fg,mid,bg) one WINDOW: the recursion `fg` on wavefront 0, a follower `mid` on wavefront 1
//                         (its own wave-local `ex.sub` phases), barrier-free background work
//                         `bg(lane, lanes)` -- chunk copies -- on the rest; one barrier at the end
//   ex.post / ex.await    progress counter (LDS) from the recursion to its followers
//   ex.share/gather/shl6/shr6  hand a value from lane to lane between seq phases (registers)
//   ex.put_* / get_*      reductions over the simulation's lanes;  ex.uni(v): wave-uniform value
// On the GPU `Ex` is DevExec<NW> (mpc_kernel.hip).  tests/emu instantiates the same template with
// a host executor that loops over the lanes and runs the roles one after the other -- a
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
// Memory: every pass streams CHUNKS of consecutive stage records HBM -> LDS -> HBM
// (mpc_layout.h); the three sweeps keep two chunks in flight so that the copies run in the shadow
// of the recursion.  One IPM iteration is five passes: factorisation sweep, predictor sweep +
// step lengths, corrector + backward solve, forward sweep + step lengths, update + residuals.
#pragma once
#include "mpc_kin.h"
#include "mpc_ipm.h"

// 16-way manual unrolling with individually named registers (see Engine::copy_lanes)
#define MPC_REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define MPC_REP4(X) X(0) X(1) X(2) X(3)

namespace mpcb {

constexpr double BOUND_INF = 1e29;

// LDS row stride of a G1 (ITER) record in the chunk-parallel passes: 96 doubles = 768 B would put
// the same column of EVERY stage on the same LDS banks (256-B bank period) -- lanes that work on
// different stages of one column then serialise; 98 shifts consecutive stages by 16 B.
constexpr int L1 = 98;
// same reason for the part [R..GV] (60 doubles) of the G2 record that the NLP pass builds with one
// lane per stage (every lane touches the same column of a different stage): stride 62
constexpr int L2N = 62;
// the whole G2 record (112 doubles: the linearisation AND the fast path's right-hand side, written by the same NLP pass): stride 114
// (912 B = 3 x 256 + 9 x 16: sixteen consecutive stages fall on sixteen different 16-byte bank groups)
constexpr int L2W = 114;

#define NLP_PASS nlp_direct
// 1 (default): SQP_RTI folds the fast path's commit and right-hand-side item passes into the NLP pass; 0: separate passes (A/B builds)
#ifndef MPCB_FUSE
#define MPCB_FUSE 1
#endif
#ifdef MPCB_PROFILE
#define PROF_T0(v) const double v = ex.clock()
#define PROF_ADD(i, v) prof[i] += ex.clock() - v
#else
#define PROF_T0(v)
#define PROF_ADD(i, v)
#endif
enum { PF_NLP = 0, PF_RES, PF_FACT, PF_BWD, PF_FWD, PF_MERIT, PF_PLANT, PF_TOTAL, PF_COUNT_IPM, PF_IO, PF_SEQ_FACT, PF_SEQ_BWD, PF_SEQ_FWD, PF_X1, PF_X2, PF_X3 };

// The LDS working set is reached through the executor (ex.smem(), ex.pool()) and never through a
// stored pointer: inside a non-inlined pass a pointer loaded from `this` is a generic (flat)
// pointer, and every LDS access through it becomes a flat_load/flat_store with vmcnt+lgkmcnt
// waits.  The accessors return the __shared__ objects themselves, so the compiler keeps ds_* ops.
struct Ctx {
    const Problem *pb;
    Ws w;
    int pool_n;     // doubles in the LDS chunk pool
    int N;
};

// HBM pointers inside the bulk copies carry the global address space explicitly (same reason).
#if defined(__HIP_DEVICE_COMPILE__)
#define MPC_GLOBAL __attribute__((address_space(1)))
#define MPC_LOCAL __attribute__((address_space(3)))
#else
#define MPC_GLOBAL
#define MPC_LOCAL
#endif

// 16-byte register type of the bulk copies: a native vector (a struct here turns every load into
// a memcpy to a stack slot that SROA does not always remove)
typedef double D2 __attribute__((ext_vector_type(2)));

// ---- bound bookkeeping (trajectory_optimizer.py:164-171: lbu on stages 0..N-1, lbx on
// q of stages 1..N-1; x_0 is fixed by lbx_0 = ubx_0, simulator.py:210-211) -------------
using ipm::has_comp;
MPC_HD double bnd_lo(const InstParams &P, int j) { return j < 6 ? P.umin[j] : P.qmin[j - 6]; }
MPC_HD double bnd_hi(const InstParams &P, int j) { return j < 6 ? P.umax[j] : P.qmax[j - 6]; }
MPC_HD int imin(int a, int b) { return a < b ? a : b; }
MPC_HD int imax(int a, int b) { return a > b ? a : b; }

// 24-bit multiply (full rate; a 32-bit integer multiply is a quarter-rate v_mul_lo / v_mad_u64_u32 on gfx950):
// row * stride products of the chunk copies stay far below 2^24
MPC_HD int mul24(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
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
MPC_HD void load_constants(Ex &ex, const InstParams *P, const Robot *rb)
{
    Smem &sm = ex.smem();
    ex.par([&](int lane) {
        const double *ps = reinterpret_cast<const double *>(P);
        double *pd = reinterpret_cast<double *>(&sm.P);
        for (int e = lane; e < (int)(sizeof(InstParams) / sizeof(double)); e += Ex::NT) pd[e] = ps[e];
        const double *rs = reinterpret_cast<const double *>(rb);
        double *rd = reinterpret_cast<double *>(&sm.rb);
        for (int e = lane; e < (int)(sizeof(Robot) / sizeof(double)); e += Ex::NT) rd[e] = rs[e];
        if (lane < 24) {   // which bound sides exist (|bound| >= 1e29 means absent, include/mpcbatch.h)
            const int j = lane < 12 ? lane : lane - 12;
            const double b = lane < 12 ? (j < 6 ? ps[offsetof(InstParams, umin) / 8 + j] : ps[offsetof(InstParams, qmin) / 8 + j - 6])
                                       : (j < 6 ? ps[offsetof(InstParams, umax) / 8 + j] : ps[offsetof(InstParams, qmax) / 8 + j - 6]);
            sm.bon[lane] = (lane < 12 ? b > -BOUND_INF : b < BOUND_INF) ? 1.0 : 0.0;
        }
    });
}

template <class Ex>
struct Engine {
    static constexpr int NT = Ex::NT;  // lanes per simulation
    Ex &ex;
    Ctx c;
    int N;
    double lin_cost;  // cost of the linearisation currently held in G2
    int fast_skip, fast_back;   // fast path: QPs left before the next attempt; length of the current suspension (ipm::fast_backoff)
    // SQP_RTI, fast path: the two item passes around the sweeps are folded into the NLP pass that follows / precedes them
    //   commit_pending  the accepted candidate has not been written to the QP iterate yet: the next NLP pass does it with its update
    //   rhs_valid       G2 already holds the fast path's right-hand side (gt, Gamma = 0, rb) for the current linearisation and x_hat
    bool commit_pending, rhs_valid;
#ifdef MPCB_PROFILE
    double prof[NPROF];
#endif

    MPC_HD Engine(Ex &e, const Ctx &cc) : ex(e), c(cc), N(cc.N), lin_cost(0.0), fast_skip(0), fast_back(0), commit_pending(false), rhs_valid(false)
    {
        ex.par([&](int lane) {
            if (lane == 0) { Smem &sm = ex.smem(); sm.w = cc.w; sm.n_hor = cc.N; sm.pool_n = cc.pool_n; }
        });
#ifdef MPCB_PROFILE
        for (int i = 0; i < NPROF; i++) prof[i] = 0.0;
#endif
    }

    // =========================================================================== chunk I/O
    // Rectangle = columns [C0, C0+W) of stages [k_lo, k_hi] of a group with row stride LDG (HBM);
    // the LDS copy has row stride LDL.  W, C0, LDG, LDL even: everything moves as 16-byte items,
    // sixteen per lane in flight (counted vmcnt).  Lanes map to (row, column) with a power-of-two
    // column pitch, so addressing is shifts and masks -- at one or four waves per simulation the
    // copies are bound by instruction issue and latency, not by bytes.  Out-of-range lanes are
    // clamped onto the last valid item (duplicate copies of the same value are harmless).
    // The lanes' share of one rectangle copy: NL lanes (numbered 0..NL-1) take part, no barrier.
    template <int W, int C0, int LDG, int LDL, bool LOAD, int NL>
    MPC_HD void copy_lanes(double *l, double *g, int k_lo, int k_hi, int lane)
    {
        static_assert(W % 2 == 0 && C0 % 2 == 0 && LDG % 2 == 0 && LDL % 2 == 0, "16-byte granularity");
        if constexpr (!(W == LDG && W == LDL) && W > 2 * WAVE) {
            // partial rows wider than one wavefront of 16-byte items: column pieces of 128 doubles
            copy_lanes<2 * WAVE, C0, LDG, LDL, LOAD, NL>(l, g, k_lo, k_hi, lane);
            copy_lanes<W - 2 * WAVE, C0 + 2 * WAVE, LDG, LDL, LOAD, NL>(l + 2 * WAVE, g, k_lo, k_hi, lane);
            return;
        }
        constexpr int W2h = W / 2;
        constexpr int SH = W2h <= 1 ? 0 : (W2h <= 2 ? 1 : (W2h <= 4 ? 2 : (W2h <= 8 ? 3 : (W2h <= 16 ? 4 : (W2h <= 32 ? 5 : (W2h <= 64 ? 6 : (W2h <= 128 ? 7 : 8)))))));
        constexpr int PITCH = 1 << SH;             // >= W2h
        constexpr bool FLAT = (W == LDG && W == LDL);   // whole rows: one contiguous span on both sides
        static_assert(FLAT || W > 2 * WAVE || (PITCH >= W2h && PITCH <= WAVE), "one partial row fits a wavefront of 16-byte items");
        static_assert(NL % WAVE == 0, "whole wavefronts");
        constexpr int RPI = FLAT ? 1 : NL / PITCH;  // rows covered by one instruction group
        // the bounds are the same in every lane; telling the compiler so keeps the copy loops scalar
        // (a lane-divergent loop here also trips an AGPR-reload-under-empty-exec miscompile in hipcc 7.2)
        k_lo = ex.uni(k_lo);
        const int rows = ex.uni(k_hi - k_lo + 1);
        if (rows <= 0) return;
        MPC_GLOBAL D2 *gb = (MPC_GLOBAL D2 *)(ex.uni(g) + (size_t)k_lo * LDG + C0);
        MPC_LOCAL D2 *lb = (MPC_LOCAL D2 *)ex.uni(l);
        if (FLAT) {
            const int tot = rows * W2h;
            for (int base = 0; base < tot; base += NL * 16) {
#define MPC_AD(u) const int e##u = imin(base + u * NL + lane, tot - 1); \
    MPC_GLOBAL D2 *gq##u = (MPC_GLOBAL D2 *)((MPC_GLOBAL char *)gb + (unsigned)(e##u * 16));
#define MPC_LD(u) const D2 v##u = LOAD ? *gq##u : lb[e##u];
#define MPC_ST(u)                  \
    if (LOAD) lb[e##u] = v##u;     \
    else *gq##u = v##u;
                MPC_REP16(MPC_AD)
                MPC_REP16(MPC_LD)
                MPC_REP16(MPC_ST)
#undef MPC_AD
#undef MPC_LD
#undef MPC_ST
            }
            return;
        }
        // lanes beyond the row's last 16-byte item sit out (a clamped duplicate would be a same-address
        // LDS write conflict); rows beyond the last one are clamped, in groups of 4 or 16 instructions
        const int col = lane & (PITCH - 1);
        const int r0 = lane >> SH;
        if (col >= W2h) return;
#define MPC_AD(u)                                               \
    const int row##u = imin(rb0 + u * RPI + r0, rows - 1);      \
    MPC_GLOBAL D2 *gp##u = (MPC_GLOBAL D2 *)((MPC_GLOBAL char *)gb + (unsigned)(mul24(row##u, LDG * 8) + col * 16)); /* scalar base + 32-bit byte offset */ \
    MPC_LOCAL D2 *lp##u = (MPC_LOCAL D2 *)((MPC_LOCAL char *)lb + (mul24(row##u, LDL * 8) + col * 16));
#define MPC_LD(u) const D2 v##u = LOAD ? *gp##u : *lp##u;
#define MPC_ST(u)                  \
    if (LOAD) *lp##u = v##u;       \
    else *gp##u = v##u;
        int rb0 = 0;
        for (; rb0 + RPI * 4 < rows; rb0 += RPI * 16) {     // more than 4 instruction groups left
            MPC_REP16(MPC_AD)
            MPC_REP16(MPC_LD)
            MPC_REP16(MPC_ST)
        }
        if (rb0 < rows) {                                    // at most 4 groups left
            MPC_REP4(MPC_AD)
            MPC_REP4(MPC_LD)
            MPC_REP4(MPC_ST)
        }
#undef MPC_AD
#undef MPC_LD
#undef MPC_ST
    }
    template <int W, int C0, int LDG, int LDL, bool LOAD>
    MPC_HD void copy_rect(double *l, double *g, int k_lo, int k_hi)
    {
        PROF_T0(t0);
        ex.par([&](int lane) { copy_lanes<W, C0, LDG, LDL, LOAD, NT>(l, g, k_lo, k_hi, lane); });
        PROF_ADD(PF_IO, t0);
    }
    // several rectangle copies in ONE phase (one barrier, the loads of all rectangles in flight together)
    template <class F>
    MPC_HD void copies(F &&f)
    {
        PROF_T0(t0);
        ex.par([&](int lane) { f(lane, std::integral_constant<int, NT>{}); });
        PROF_ADD(PF_IO, t0);
    }
    template <int W, int C0, int LDG>
    MPC_HD void load_rect(double *l, const double *g, int k_lo, int k_hi)
    {
        copy_rect<W, C0, LDG, W, true>(l, const_cast<double *>(g), k_lo, k_hi);
    }
    template <int W, int C0, int LDG>
    MPC_HD void store_rect(const double *l, double *g, int k_lo, int k_hi)
    {
        copy_rect<W, C0, LDG, W, false>(const_cast<double *>(l), g, k_lo, k_hi);
    }

    MPC_HD int chunk_len(int per_stage, int halo_doubles) const
    {
        const int ch = (ex.smem().pool_n - halo_doubles) / per_stage;
        return ex.uni(imax(1, imin(ch, ex.smem().n_hor + 1)));
    }
    MPC_HD int chunk_len_in(int budget, int per_stage, int halo_doubles) const
    {
        const int ch = (budget - halo_doubles) / per_stage;
        return ex.uni(imax(1, imin(ch, ex.smem().n_hor + 1)));
    }

    // =========================================================================== LDS-resident factor
    // When the horizon is short enough (N <= ~125 with the whole pool of a CU), the part of the Riccati factor the
    // three solve sweeps read -- K (72), R~^-1 h_u (6), e (12), p (12) per stage -- stays in LDS from the factorisation
    // sweep to the final forward sweep of the same interior-point iteration and never travels through HBM.  With the
    // whole horizon on chip the two vector recursions (dx forward, p backward) are no longer stage-by-stage chains on
    // one wavefront: the horizon is cut into J = RS_GROUPS chunks of L transitions, one 16-lane group each;
    //   pass 1: every group runs its chunk from a zero boundary value,
    //   pass 2: one group walks the J chunk boundaries with the chunk transition matrices Phi_c = Acl_{t-1} ... Acl_s
    //           (12x12, accumulated for free by an idle wavefront while the factorisation sweep runs),
    //   pass 3: every group runs its chunk again from its true boundary value.
    // 2 L + J steps instead of N (30 instead of 100 at N = 100, four wavefronts).  Everything else of the sweeps is
    // item-parallel (lane <-> (stage, component)) with its operands loaded straight from HBM into registers, issued
    // before the recursion so that their latency runs under it.
    struct ResMap {
        double *K, *VH, *E, *P, *PHI;   // persistent: [NS][72], [NS][6], [NS][12], [NS][12], [J][144]
        double *scr;                    // scratch below them (fact: chunk buffers; sweeps: dx / gt / w / hand-over slots)
        int scr_n, L, J, T;             // chunk length, lane groups in use, transitions the maps hold (J * L; the horizon when resident)
    };
    static constexpr int RS_GROUPS = (NT / 16) < 16 ? (NT / 16) : 16;
    // items a lane holds in flight per batch, sized so that ONE batch covers N ~ 105 at this lane count (more lanes: fewer items each,
    // and the 8-wavefront geometry has half the registers)
    static constexpr int rounds_for(int items_per_stage)
    {
        return NT >= 256 ? (items_per_stage * 106 + NT - 1) / NT : (items_per_stage * 106 + 255) / 256;   // (1-2 wavefronts: more batches instead)
    }
    static constexpr int RS_ROUNDS = rounds_for(12);
    static constexpr int RS_PER_STAGE = RS_PER_STAGE_L;
    MPC_HD ResMap res_map() const
    {
        const int Nl = ex.uni(ex.smem().n_hor), NS = Nl + 1, pool_n = ex.uni(ex.smem().pool_n);
        const int persist = NS * RS_PER_STAGE + RS_GROUPS * 144;
        ResMap m;
        m.scr = ex.pool();
        m.scr_n = pool_n - persist;
        m.K = ex.pool() + m.scr_n;
        m.VH = m.K + (size_t)NS * 72;
        m.E = m.VH + (size_t)NS * 6;
        m.P = m.E + (size_t)NS * 12;
        m.PHI = m.P + (size_t)NS * 12;
        m.L = (Nl + RS_GROUPS - 1) / RS_GROUPS;
        m.J = RS_GROUPS;
        m.T = Nl;
        return m;
    }
    // SEGMENT map: the same arrays sized for one segment of SEG_T transitions (+ its end state), chunk length SEG_L
    MPC_HD ResMap seg_map() const
    {
        const int pool_n = ex.uni(ex.smem().pool_n);
        const bool full = pool_n >= SEG_POOL_FULL;
        const int Lc = full ? SEG_L_FULL : SEG_L_HALF, Jc = imin(full ? SEG_J_FULL : SEG_J_HALF, RS_GROUPS), NSS = Lc * Jc + 1;
        const int persist = NSS * RS_PER_STAGE + Jc * 144;
        ResMap m;
        m.scr = ex.pool();
        m.scr_n = pool_n - persist;
        m.K = ex.pool() + m.scr_n;
        m.VH = m.K + (size_t)NSS * 72;
        m.E = m.VH + (size_t)NSS * 6;
        m.P = m.E + (size_t)NSS * 12;
        m.PHI = m.P + (size_t)NSS * 12;
        m.L = Lc; m.J = Jc; m.T = Lc * Jc;
        return m;
    }
    // REGISTER map (rs_recursion_reg): nothing of the factor in LDS -- only p (c -> p), the chunk transition matrices and scratch
    MPC_HD ResMap reg_map() const
    {
        const int pool_n = ex.uni(ex.smem().pool_n);
        constexpr int NSS = 16 * RL + 1;
        const int persist = NSS * 12 + 16 * 144;
        ResMap m;
        m.scr = ex.pool();
        m.scr_n = pool_n - persist;
        m.K = nullptr; m.VH = nullptr; m.E = nullptr;
        m.P = ex.pool() + m.scr_n;
        m.PHI = m.P + (size_t)NSS * 12;
        m.L = RL; m.J = 16; m.T = 16 * RL;
        return m;
    }
    // MPCB_REG_MODE: 0 never, 1 (default) with half a CU's pool (two simulations per CU: batch 512, N = 100: 876 k steps/s against
    // 760 k with the streaming sweeps, N = 200: 325 k against 277 k), 2 also with the whole pool instead of the LDS segments (level
    // with them: N = 200 / 300, batch 256: 238 k / 138 k either way)
#ifndef MPCB_REG_MODE
#define MPCB_REG_MODE 1
#endif
    MPC_HD bool reg_ok() const
    {
#if defined(MPCB_NO_RESIDENT) || MPCB_REG_MODE == 0
        return false;
#else
        const ResMap m = reg_map();
        const bool full = ex.smem().pool_n >= SEG_POOL_FULL;
        return ex.uni(RS_GROUPS == 16 && !resident_ok() && (MPCB_REG_MODE >= 2 || !full) && m.scr_n >= (m.T + 1) * 30 + 3 * 16 * 12 + 64);
#endif
    }
    // horizons beyond the resident limit, with a whole CU's pool: segment-wise residency (the factor goes through HBM, the sweeps
    // load it back one segment at a time and run chunk-parallel inside the segment)
    MPC_HD bool segment_ok() const
    {
#ifdef MPCB_NO_RESIDENT
        return false;
#else
        // (with half a pool -- two simulations per CU -- segments of 8 x 5 transitions fit, but lose to the streaming sweeps there:
        // batch 512, N = 100: 592 k vs 752 k steps/s; N = 200: 238 k vs 282 k: the segment loads are exposed and short)
        return ex.uni(lay_segment_ok(ex.uni(ex.smem().n_hor), ex.uni(ex.smem().pool_n), RS_GROUPS));
#endif
    }
    MPC_HD bool resident_ok() const
    {
#ifdef MPCB_NO_RESIDENT
        return false;
#else
        return ex.uni(lay_resident_ok(ex.uni(ex.smem().n_hor), ex.uni(ex.smem().pool_n), RS_GROUPS));
#endif
    }
    MPC_HD static double gld(const double *p) { return *(MPC_GLOBAL const double *)p; }
    MPC_HD static void gst(double *p, double v) { *(MPC_GLOBAL double *)p = v; }

    // The fast path's right-hand side of joint j at one stage (Gamma = 0): condensed gradient gt = g at (dw, pi, lam) = 0 with the
    // feedback step dx_0 = x_hat - x_0 embedded, and rb = b + A dx_0.  ONE function with explicit fused multiply-adds for its two
    // callers -- fast_rhs (a pass of its own: first attempt of a launch, SQP) and the residual-norm items of nlp_direct (SQP_RTI) -- so
    // that a run cut into several launches reproduces the single launch bit for bit whichever of them formed the right-hand side.
    struct RhsItem { double gtu, gtq, gtv, rbq, rbv; };
    MPC_HD static RhsItem rhs_item(const InstParams &P, int j, bool st, bool inner, double uj, double vj, double dxq, double dxv,
                                   double g0, double g1, double g2, double g3, double g4, double y0, double y1, double y2, double y3,
                                   double y4, double gv, double bdq, double bdv)
    {
        const double c2 = P.w_qddot * P.cq[j] * P.cq[j];
        const double vjn = vj + dxv;
        double s_ = g0 * y0;
        s_ = fma(g1, y1, s_); s_ = fma(g2, y2, s_); s_ = fma(g3, y3, s_); s_ = fma(g4, y4, s_);
        const double tu = fma(c2, uj - vjn, 2.0 * P.w_u * uj), tv = fma(c2, vjn - uj, gv * y4);
        RhsItem o;
        o.gtu = st ? P.dt * tu : 0.0;
        o.gtq = st && inner ? P.dt * s_ : 0.0;
        o.gtv = st && inner ? P.dt * tv : 0.0;
        o.rbq = st ? fma(P.a12[j], dxv, dxq) + bdq : 0.0;
        o.rbv = st ? fma(P.a22[j], dxv, bdv) : 0.0;
        return o;
    }

    // =========================================================================== NLP pass
    // One pass over the horizon that (optionally) applies the SQP/RTI step to the iterate
    // (acados ocp_nlp_update_variables_sqp), linearises at the new iterate -- task residual and
    // Jacobian per stage (lane <-> stage), dynamics defect, cost = sum_k dt/2 r'Wr (acados
    // get_cost(), simulator.py:221) -- and evaluates acados' ocp_nlp_res_compute inf-norms
    // [stat, eq, ineq, comp] with the NLP multipliers (RTI: the QP's; SQP: the blended ones).
    // One joint of the plant step (simulation_model.py:93-117): Euler / RK2 (midpoint) / RK3 / RK4 of z' = [qdot; -W (qdot - u)]
    MPC_HD static void plant_rk(const InstParams &P, int j, double q, double v, double u, double &qn, double &vn)
    {
        const double wc = P.wcv[j], dt = P.dt;
        const int integ = (int)P.integ;
        const double k1q = v, k1v = -wc * v + wc * u;
        const double v2 = v + 0.5 * dt * k1v;
        const double k2q = v2, k2v = -wc * v2 + wc * u;
        if (integ == 1) {
            qn = q + dt * k1q; vn = v + dt * k1v;
        } else if (integ == 2) {
            qn = q + dt * k2q; vn = v + dt * k2v;
        } else if (integ == 3) {
            const double v3 = v - dt * k1v + 2.0 * dt * k2v;
            const double k3q = v3, k3v = -wc * v3 + wc * u;
            qn = q + (dt / 6) * (k1q + 4.0 * k2q + k3q); vn = v + (dt / 6) * (k1v + 4.0 * k2v + k3v);
        } else {
            const double v3 = v + 0.5 * dt * k2v;
            const double k3q = v3, k3v = -wc * v3 + wc * u;
            const double v4 = v + dt * k3v;
            const double k4q = v4, k4v = -wc * v4 + wc * u;
            qn = q + (dt / 6) * k1q + (dt / 3) * k2q + (dt / 3) * k3q + (dt / 6) * k4q;
            vn = v + (dt / 6) * k1v + (dt / 3) * k2v + (dt / 3) * k3v + (dt / 6) * k4v;
        }
    }

    // Item-parallel (any horizon): the iterate is updated IN PLACE in HBM by 16-byte items, every
    // lane linearises one stage from operands it loads itself, and the residual norms are joint items (u_j, q_j, v_j together) --
    // no staging of the 96-column iterate record through LDS (81 KB per pass at N = 100), no copy-back.  The linearisation records
    // (60 doubles per stage) collect in LDS and leave with one coalesced store.
    // `do_plant` (SQP_RTI: this is the last pass of solve()): the plant step with u = solver.get(0,'u') and the FK / J qdot / task-error
    // log of the NEW plant state (simulation_model.py:85-91) are one lane's work; they run here on a lane that has no stage to
    // linearise, in the shadow of the linearisation, instead of alone after the solve (6 us per MPC step).  Results in sm.logv / sm.u0.
    // `fuse_commit` (SQP_RTI, the fast path accepted its candidate and left it in the Newton-step slots): fast_commit's work is done by
    // the update items here -- the candidate becomes the QP iterate AND is added to (X | U) by the same item, one pass and one read of the
    // step less.  `want_rhs` (SQP_RTI, the next QP will try the fast path): the joint items of the residual norms also form the fast
    // path's right-hand side for the NEW plant state (gt = g, Gamma = 0, rb = b + A dx_0: the same products the stationarity rows are
    // made of), so fast_rhs needs no pass of its own; with room for whole G2 records in LDS (stride L2W) it leaves with the
    // linearisation in one coalesced store, otherwise by 8-byte stores from the items.
    MPC_PASS double nlp_direct(double alpha, bool do_update, bool sqp_mult, double *res4, bool do_plant = false, bool fuse_commit = false,
                               bool want_rhs = false)
    {
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const Robot &rb = sm.rb;
        const int Nl = ex.uni(ex.smem().n_hor), NS = Nl + 1;
        double *const G1 = ex.smem().w.G1, *const G5 = ex.smem().w.G5;
        want_rhs = ex.uni(want_rhs && res4 != nullptr && do_plant);
        const bool wide = ex.uni(want_rhs && ex.smem().pool_n >= NS * L2W);   // the whole horizon's G2 records fit: one chunk
        const int LS = wide ? L2W : L2N;
        // LDS: the linearisation records of as many stages as fit (row stride L2N: lane <-> stage accesses without bank aliasing)
        const int CH = ex.uni(imax(1, imin(ex.smem().pool_n / LS, NS)));
        double *const v2 = ex.pool();
        PROF_T0(tu);
        if (ex.uni(do_update && fuse_commit)) {
            // fast_commit + update in one: item (k, c) of the 15 pairs of (QW | QPI) <- (DW | DPI) [x_0 embedded, the multiplier shifted by
            // one stage]; the 9 pairs of QW are also the step of (X | U).  Then QLAM <- 0, QT <- the candidate's slacks.
            double *const G3 = ex.smem().w.G3;
            {
                constexpr int IPS = 15, R = rounds_for(IPS);
                const int items = NS * IPS;
                for (int base = 0; base < items; base += R * NT) {
                    ex.wpar([&](int lane) {
                        D2 cur[R], stp[R];
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const int e = imin(base + r * NT + lane, items - 1), k = e / IPS, c = 2 * (e - k * IPS);
                            stp[r] = *(MPC_GLOBAL const D2 *)(G3 + (size_t)(c >= 18 ? imin(k + 1, Nl) : k) * W3 + O_DW + c);
                            cur[r] = *(MPC_GLOBAL const D2 *)(G1 + (size_t)k * W1 + (c < 6 ? 12 + c : (c < 18 ? c - 6 : 0)));
                        }
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const int e = base + r * NT + lane;
                            if (e < items) {
                                const int k = e / IPS, c = 2 * (e - k * IPS);
                                D2 v = stp[r];
                                if (c >= 18 && k >= Nl) { v.x = 0.0; v.y = 0.0; }              // no multiplier beyond the last dynamics
                                if (c < 18) {
                                    D2 x = cur[r];
                                    if (k == 0 && c >= 6) { v.x = sm.xhat[c - 6] - x.x; v.y = sm.xhat[c - 5] - x.y; }   // x_0 = x_hat (lbx_0 = ubx_0)
                                    const double aa = (c < 6 && k >= Nl) ? 0.0 : alpha;   // no input at stage N
                                    x.x += aa * v.x; x.y += aa * v.y;
                                    *(MPC_GLOBAL D2 *)(G1 + (size_t)k * W1 + (c < 6 ? 12 + c : c - 6)) = x;
                                }
                                *(MPC_GLOBAL D2 *)(G1 + (size_t)k * W1 + O_QW + c) = v;
                            }
                        }
                    });
                }
            }
            {
                constexpr int IPS = 24, R = rounds_for(IPS);
                const int items = NS * IPS;
                for (int base = 0; base < items; base += R * NT) {
                    ex.wpar([&](int lane) {
                        D2 stp[R];
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const int e = imin(base + r * NT + lane, items - 1), k = e / IPS, q = 2 * (e - k * IPS);
                            stp[r] = *(MPC_GLOBAL const D2 *)(G3 + (size_t)k * W3 + O_DT + (q >= 24 ? q - 24 : 0));
                        }
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const int e = base + r * NT + lane;
                            if (e < items) {
                                const int k = e / IPS, q = 2 * (e - k * IPS);
                                D2 v = stp[r];
                                if (q < 24) { v.x = 0.0; v.y = 0.0; }
                                *(MPC_GLOBAL D2 *)(G1 + (size_t)k * W1 + O_QLAM + q) = v;
                            }
                        }
                    });
                }
            }
            ex.barrier();
        } else if (do_update) {
            // (X | U) += alpha * (dx | du): G1 columns [0, 18) <- columns [24, 36) | [18, 24); SQP: multipliers blend towards the QP's
            constexpr int IPS = 9, R = rounds_for(IPS);
            const int items = NS * IPS;
            for (int base = 0; base < items; base += R * NT) {
                ex.wpar([&](int lane) {
                    D2 cur[R], stp[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), k = e / IPS, c = 2 * (e - k * IPS);
                        const double *g1 = G1 + (size_t)k * W1;
                        cur[r] = *(MPC_GLOBAL const D2 *)(g1 + c);
                        stp[r] = *(MPC_GLOBAL const D2 *)(g1 + O_QW + (c < 12 ? 6 + c : c - 12));
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int k = e / IPS, c = 2 * (e - k * IPS);
                            const double aa = (c >= 12 && k >= Nl) ? 0.0 : alpha;   // no input at stage N
                            D2 v = cur[r];
                            v.x += aa * stp[r].x; v.y += aa * stp[r].y;
                            *(MPC_GLOBAL D2 *)(G1 + (size_t)k * W1 + c) = v;
                        }
                    }
                });
            }
            if (sqp_mult) {
                // NPI | NLAM | NT (G5 columns [0, 60))  <-  blend towards QPI | QLAM | QT (G1 columns [36, 96))
                constexpr int IPS5 = 30, R5 = rounds_for(IPS5);
                const int items5 = NS * IPS5;
                for (int base = 0; base < items5; base += R5 * NT) {
                    ex.wpar([&](int lane) {
                        D2 cur[R5], qp[R5];
#pragma unroll
                        for (int r = 0; r < R5; r++) {
                            const int e = imin(base + r * NT + lane, items5 - 1), k = e / IPS5, c = 2 * (e - k * IPS5);
                            cur[r] = *(MPC_GLOBAL const D2 *)(G5 + (size_t)k * W5 + c);
                            qp[r] = *(MPC_GLOBAL const D2 *)(G1 + (size_t)k * W1 + O_QPI + c);
                        }
#pragma unroll
                        for (int r = 0; r < R5; r++) {
                            const int e = base + r * NT + lane;
                            if (e < items5) {
                                const int k = e / IPS5, c = 2 * (e - k * IPS5);
                                D2 v = cur[r];
                                v.x += alpha * (qp[r].x - v.x); v.y += alpha * (qp[r].y - v.y);
                                *(MPC_GLOBAL D2 *)(G5 + (size_t)k * W5 + c) = v;
                            }
                        }
                    });
                }
            }
            ex.barrier();
        }
        PROF_ADD(PF_X1, tu);
        double cost = 0.0, rs = 0.0, re = 0.0, ri = 0.0, rc = 0.0;
        for (int k0 = 0; k0 <= Nl; k0 += CH) {
            const int k1 = imin(k0 + CH - 1, Nl);
            // ---- linearise: lane <-> stage
            PROF_T0(tl);
            ex.par([&](int lane) {
                double csum = 0.0;
                for (int k = k0 + lane; k <= k1; k += NT) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MPCB_NO_LICM_BLOCK)
                    asm volatile("" ::: "memory");     // (parameters are read where they are used, not hoisted out of the loop and spilled: see merit_pass)
#endif
                    double *rec = v2 + (size_t)(k - k0) * LS;
                    if (k < Nl) {
                        const double *g1 = G1 + (size_t)k * W1;
                        double xx[12], uu[6], xn[12];
#pragma unroll
                        for (int i = 0; i < 12; i++) { xx[i] = gld(g1 + O_X + i); xn[i] = gld(g1 + W1 + O_X + i); }
#pragma unroll
                        for (int i = 0; i < 6; i++) uu[i] = gld(g1 + O_U + i);
                        // everything that needs u and x_{k+1} first: the kinematics below want every register (Kin alone is 48
                        // doubles; the 256-register builds spilled what stayed live across it)
                        double cu[6];
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            const double uj = uu[j], vj = xx[6 + j];
                            const double qdd = P.cq[j] * (uj - vj);  // prediction_model.py:326
                            cu[j] = 2.0 * P.w_u * uj * uj + P.w_qddot * qdd * qdd;
                            rec[O_BD + j] = (xx[j] + P.a12[j] * vj + P.b1[j] * uj) - xn[j];
                            rec[O_BD + 6 + j] = (P.a22[j] * vj + P.b2[j] * uj) - xn[6 + j];
                        }
                        task_lin<true>(rb, P, xx, xx + 6, rec);
                        double s_ = 0.0;
#pragma unroll
                        for (int i = 0; i < NTASK; i++) {
                            const double r = rec[O_R + i];
                            s_ += P.w_task[i] * r * r;
                            rec[O_Y + i] = P.w_task[i] * r;
                        }
#pragma unroll
                        for (int j = 0; j < 6; j++) s_ += cu[j];
                        csum += 0.5 * P.dt * s_;
                    } else {
#pragma unroll
                        for (int i = 0; i < W2_LIN; i++) rec[i] = 0.0;
                    }
                }
                if (do_plant && k0 == 0 && lane == NT - WAVE) {
                    const double tp0 = ex.clock();
                    double zn[12];
                    for (int j = 0; j < 6; j++) {
                        const double u = gld(G1 + O_U + j);                 // (stage 0, after the update above)
                        plant_rk(P, j, sm.xhat[j], sm.xhat[6 + j], u, zn[j], zn[6 + j]);
                        sm.u0[j] = u;
                        sm.logv[24 + j] = zn[j]; sm.logv[30 + j] = zn[6 + j];
                    }
                    plant_log(rb, zn, sm.logv);
                    task_errors(sm.P, rb, sm.logv, sm.logv + 15, sm.logv + 36);
                    sm.ret[6] = ex.clock() - tp0;
                }
                ex.put_sum(sm.red[4], lane, csum);
            });
            cost += ex.get_sum(sm.red[4]);
            PROF_ADD(PF_X2, tl);
            PROF_T0(tn);
            // ---- residual norms: joint items (k, j < 6): rows u_j, q_j, v_j of the stationarity residual, their bounds, the defect
            if (res4) {
                constexpr int R = rounds_for(6);
                const int items = (k1 - k0 + 1) * 6;
                typename Ex::template PerLane<double> a_s, a_e, a_i, a_c;
                ex.wpar([&](int lane) { a_s.at(lane) = 0; a_e.at(lane) = 0; a_i.at(lane) = 0; a_c.at(lane) = 0; });
                for (int base = 0; base < items; base += R * NT) {
                    ex.wpar([&](int lane) {
                        double v[R][15];
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const int e = imin(base + r * NT + lane, items - 1), s = e / 6, j = e - s * 6, k = k0 + s, km = imax(k - 1, 0);
                            const double *g1 = G1 + (size_t)k * W1;
                            // multipliers: the QP's (RTI) or the blended NLP ones (SQP); both laid out pi 12 | lam 24 | t 24
                            const double *mk = sqp_mult ? G5 + (size_t)k * W5 : g1 + O_QPI;
                            const double *mm = sqp_mult ? G5 + (size_t)km * W5 : G1 + (size_t)km * W1 + O_QPI;
                            v[r][0] = gld(g1 + O_U + j); v[r][1] = gld(g1 + O_X + j); v[r][2] = gld(g1 + O_X + 6 + j);
                            v[r][3] = gld(mk + j); v[r][4] = gld(mk + 6 + j); v[r][5] = gld(mm + j); v[r][6] = gld(mm + 6 + j);
                            v[r][7] = gld(mk + 12 + j);  v[r][8] = gld(mk + 24 + j);  v[r][9] = gld(mk + 36 + j);   v[r][10] = gld(mk + 48 + j);   // u_j: lam lo, hi, t lo, hi
                            v[r][11] = gld(mk + 18 + j); v[r][12] = gld(mk + 30 + j); v[r][13] = gld(mk + 42 + j);  v[r][14] = gld(mk + 54 + j);   // q_j
                        }
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const int e = base + r * NT + lane;
                            if (e < items) {
                                const int s = e / 6, j = e - s * 6, k = k0 + s;
                                double *rec = v2 + (size_t)s * LS;
                                const double uj = v[r][0], qj = v[r][1], vj = v[r][2];
                                const double c2 = P.w_qddot * P.cq[j] * P.cq[j];
                                double as_ = a_s.at(lane), ai_ = a_i.at(lane), ac_ = a_c.at(lane);
                                auto bound = [&](int ci, double cur, double l_lo, double l_hi, double t_lo, double t_hi, double &val) {
                                    if (has_comp(Nl, k, ci)) {
                                        if (bnd_lo(P, ci) > -BOUND_INF) {
                                            val -= l_lo;
                                            ai_ = fmax(ai_, fabs((bnd_lo(P, ci) - cur) + t_lo));
                                            ac_ = fmax(ac_, fabs(l_lo * t_lo));
                                        }
                                        if (bnd_hi(P, ci) < BOUND_INF) {
                                            val += l_hi;
                                            ai_ = fmax(ai_, fabs((cur - bnd_hi(P, ci)) + t_hi));
                                            ac_ = fmax(ac_, fabs(l_hi * t_hi));
                                        }
                                    }
                                };
                                // u_j (stat_cls<0>, no step)
                                double ru = 0.0;
                                if (k < Nl) {
                                    ru = P.dt * (2.0 * P.w_u * uj + c2 * (uj - vj));
                                    ru += P.b1[j] * v[r][3] + P.b2[j] * v[r][4];
                                }
                                bound(j, uj, v[r][7], v[r][8], v[r][9], v[r][10], ru);
                                // q_j (stat_cls<1>)
                                double rq = 0.0;
                                if (k >= 1) {
                                    if (k < Nl) {
                                        double s_ = 0.0;
#pragma unroll
                                        for (int i = 0; i < NTASK; i++) s_ += rec[O_GQ + i * 6 + j] * rec[O_Y + i];
                                        rq = P.dt * s_ + v[r][3];
                                    }
                                    rq -= v[r][5];
                                }
                                bound(6 + j, qj, v[r][11], v[r][12], v[r][13], v[r][14], rq);
                                // v_j (stat_cls<2>)
                                double rv = 0.0;
                                if (k >= 1) {
                                    if (k < Nl) {
                                        rv = P.dt * (rec[O_GV + j] * rec[O_Y + 4] + c2 * (vj - uj));
                                        rv += P.a12[j] * v[r][3] + P.a22[j] * v[r][4];
                                    }
                                    rv -= v[r][6];
                                }
                                if (k == 0) { rq = 0.0; rv = 0.0; }    // x_0 is eliminated (lbx_0 = ubx_0)
                                as_ = fmax(as_, fmax(fabs(ru), fmax(fabs(rq), fabs(rv))));
                                if (k < Nl) a_e.at(lane) = fmax(a_e.at(lane), fmax(fabs(rec[O_BD + j]), fabs(rec[O_BD + 6 + j])));
                                if (k == 0) ai_ = fmax(ai_, fmax(fabs(sm.xhat[j] - qj), fabs(sm.xhat[6 + j] - vj)));   // lbx_0 = ubx_0 = x_hat
                                a_s.at(lane) = as_; a_i.at(lane) = ai_; a_c.at(lane) = ac_;
                                if (want_rhs) {
                                    // the fast path's right-hand side of the NEXT QP (fast_rhs): the new plant state is in sm.logv[24..35]
                                    // (the plant lane of the linearisation phase above)
                                    const double dxq = k == 0 ? sm.logv[24 + j] - qj : 0.0, dxv = k == 0 ? sm.logv[30 + j] - vj : 0.0;
                                    const RhsItem o = rhs_item(P, j, k < Nl, k >= 1, uj, vj, dxq, dxv, rec[O_GQ + j], rec[O_GQ + 6 + j],
                                                               rec[O_GQ + 12 + j], rec[O_GQ + 18 + j], rec[O_GQ + 24 + j], rec[O_Y], rec[O_Y + 1],
                                                               rec[O_Y + 2], rec[O_Y + 3], rec[O_Y + 4], rec[O_GV + j], rec[O_BD + j],
                                                               rec[O_BD + 6 + j]);
                                    if (wide) {
                                        rec[O_GT + j] = o.gtu; rec[O_GT + 6 + j] = o.gtq; rec[O_GT + 12 + j] = o.gtv;
                                        rec[O_GAM + j] = 0.0; rec[O_GAM + 6 + j] = 0.0;
                                        rec[O_RB + j] = o.rbq; rec[O_RB + 6 + j] = o.rbv;
                                        if (j < 2) rec[10 + j] = 0.0;     // (the record's two unused columns leave with it)
                                    } else {
                                        double *g2 = ex.smem().w.G2 + (size_t)k * W2;
                                        gst(g2 + O_GT + j, o.gtu); gst(g2 + O_GT + 6 + j, o.gtq); gst(g2 + O_GT + 12 + j, o.gtv);
                                        gst(g2 + O_GAM + j, 0.0); gst(g2 + O_GAM + 6 + j, 0.0);
                                        gst(g2 + O_RB + j, o.rbq); gst(g2 + O_RB + 6 + j, o.rbv);
                                    }
                                }
                            }
                        }
                    });
                }
                ex.par([&](int lane) {
                    ex.put_max(sm.red[0], lane, a_s.at(lane)); ex.put_max(sm.red[1], lane, a_e.at(lane));
                    ex.put_max(sm.red[2], lane, a_i.at(lane)); ex.put_max(sm.red[3], lane, a_c.at(lane));
                });
                rs = fmax(rs, ex.get_max(sm.red[0]));
                re = fmax(re, ex.get_max(sm.red[1]));
                ri = fmax(ri, ex.get_max(sm.red[2]));
                rc = fmax(rc, ex.get_max(sm.red[3]));
            }
            copies([&](int lane, auto nl) {
                constexpr int NL = decltype(nl)::value;
                if (wide) copy_lanes<W2, 0, W2, L2W, false, NL>(v2, ex.smem().w.G2, k0, k1, lane);
                else copy_lanes<W2_LIN, 0, W2, L2N, false, NL>(v2, ex.smem().w.G2, k0, k1, lane);
            });
            PROF_ADD(PF_X3, tn);
        }
        if (res4) { res4[0] = rs; res4[1] = re; res4[2] = ri; res4[3] = rc; }
        PROF_ADD(PF_NLP, t0);
        return cost;
    }

    // =========================================================================== IPM: residual pass, item-parallel
    // MODE 0 (init, HPIPM warm_start = 2): keep (w, pi, lam, t) of the previous QP, clamp lam, t >= 0.1, embed x0.  MODE 1: apply the
    // Newton step with length `a` (HPIPM update_var).  Then the QP residuals, Gamma and the condensed gradient gt of the Newton system
    // (HPIPM compute_Gamma_gamma).  Results -- inf-norms of the four residuals [g, b, d, m], sum of the complementarity products,
    // number of bound sides in mode 0 -- are left in sm.ret[0..5] (handing them back through pointers costs scratch round trips).
    // No LDS staging and no chunks: every lane works on ITEMS whose
    // operands it loads straight from HBM (8- and 16-byte loads, coalesced in runs of 6..12 doubles), all loads of a
    // batch of items in flight before the first is used.
    //   U  elementwise update of (dw, pi, lam, t) -- every element is read and written by exactly one item;
    //   Y  y_ki = w_i (r_ki + G_ki . delta_k)  -> LDS (and HBM: the SQP merit weights read stage 0's);
    //   S  stationarity / bound items (u_j and v_j of a joint together; q_j), D  dynamics residual items:
    //      read the updated iterate only, write the residual records (G2: Gamma | gt | rb, G3: rg | rd | rm).
    // Workgroup barriers separate U | Y | S,D (vector-memory operations of one CU are performed in order by its L1).
    MPC_PASS void residual_direct(int mode, double a)
    {
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(ex.smem().n_hor), NS = Nl + 1;
        double *const G1 = ex.smem().w.G1, *const G2 = ex.smem().w.G2, *const G3 = ex.smem().w.G3;
        double *const Y = ex.pool();   // [NS][6]: y of every stage (5 used)
        a = ex.uni(a);                 // the step length is the same in every lane: a scalar register pair, not a (spilled) vector one
        typename Ex::template PerLane<double> n_g, n_b, n_d, n_m, n_mu, n_c;
        PROF_T0(tx);
        // ---------------------------------------------------------------- U: 16-byte items (pairs never straddle a field)
        // (a CU's HBM rate is bytes in flight / latency, ~1.5 us here: all items of a lane in ONE batch; and everything in
        // this kernel is bound by one wavefront's instruction issue, ~6 cycles each: the items are as lean as they can be)
        ex.wpar([&](int lane) { n_g.at(lane) = 0; n_b.at(lane) = 0; n_d.at(lane) = 0; n_m.at(lane) = 0; n_mu.at(lane) = 0; n_c.at(lane) = 0; });
        if (mode == 1) {
            // dw (18) | pi (12) += a * step: G1 columns [18, 48) <- G3 columns [66, 96); dpi of multiplier k -> k+1 sits with stage k+1
            constexpr int IPS = 15, R = rounds_for(IPS);
            const int items = NS * IPS;
            for (int base = 0; base < items; base += R * NT) {
                ex.wpar([&](int lane) {
                    D2 cur[R], stp[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), k = e / IPS, c = 2 * (e - k * IPS);
                        cur[r] = *(MPC_GLOBAL const D2 *)(G1 + (size_t)k * W1 + O_QW + c);
                        stp[r] = *(MPC_GLOBAL const D2 *)(G3 + (size_t)(c >= 18 ? imin(k + 1, Nl) : k) * W3 + O_DW + c);
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int k = e / IPS, c = 2 * (e - k * IPS);
                            const double aa = (c >= 18 && k >= Nl) ? 0.0 : a;   // no multiplier beyond the last dynamics
                            D2 v = cur[r];
                            v.x += aa * stp[r].x; v.y += aa * stp[r].y;
                            *(MPC_GLOBAL D2 *)(G1 + (size_t)k * W1 + O_QW + c) = v;
                        }
                    }
                });
            }
        } else {
            // HPIPM warm start: the previous (w, pi) stay; embed x_0 (lbx_0 = ubx_0 = x_hat); no input at stage N
            ex.wpar([&](int lane) {
                if (lane < NX) gst(G1 + O_QW + 6 + lane, sm.xhat[lane] - gld(G1 + O_X + lane));
                if (lane >= 16 && lane < 16 + NU) gst(G1 + (size_t)Nl * W1 + O_QW + lane - 16, 0.0);
            });
        }
        {
            // lam (24) | t (24): G1 columns [48, 96) <- G3 columns [96, 144); q = column - 48: lam lower 12 | upper 12 | t lower | upper
            constexpr int IPS = 24, R = rounds_for(IPS);
            const int items = NS * IPS;
            for (int base = 0; base < items; base += R * NT) {
                ex.wpar([&](int lane) {
                    D2 cur[R], stp[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), k = e / IPS, q = 2 * (e - k * IPS);
                        cur[r] = *(MPC_GLOBAL const D2 *)(G1 + (size_t)k * W1 + O_QLAM + q);
                        if (mode == 1) stp[r] = *(MPC_GLOBAL const D2 *)(G3 + (size_t)k * W3 + O_DLAM + q);
                    }
                    double ncl = n_c.at(lane);
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int k = e / IPS, q = 2 * (e - k * IPS);
                            const bool is_t = q >= 24;
                            const int sc = is_t ? q - 24 : q, j = sc < 12 ? sc : sc - 12;        // side/component, component
                            const bool ok = j < 6 ? k < Nl : (k >= 1 && k < Nl);                 // has_comp (same for j and j + 1)
                            const D2 m = *reinterpret_cast<const D2 *>(&sm.bon[sc]);
                            const bool on0 = ok && m.x != 0.0, on1 = ok && m.y != 0.0;
                            D2 v = cur[r];
                            if (mode == 1) {
                                v.x = ipm::step_floor(on0, v.x, a, stp[r].x); v.y = ipm::step_floor(on1, v.y, a, stp[r].y);
                            } else {
                                v.x = is_t ? ipm::warm_t(on0, v.x) : ipm::warm_lam(on0, v.x);
                                v.y = is_t ? ipm::warm_t(on1, v.y) : ipm::warm_lam(on1, v.y);
                                ncl += (on0 && !is_t ? 1.0 : 0.0) + (on1 && !is_t ? 1.0 : 0.0);
                            }
                            *(MPC_GLOBAL D2 *)(G1 + (size_t)k * W1 + O_QLAM + q) = v;
                        }
                    }
                    n_c.at(lane) = ncl;
                });
            }
        }
        ex.barrier();
        PROF_ADD(PF_X1, tx);
        PROF_T0(ty);
        // y of a stage waits in LDS between its Y and its S items: [stages][6] of the pool.  Horizons whose y does not fit the pool
        // (N + 1 > pool / 6: beyond ~3 200 stages with a whole CU's pool, ~340 with the smallest) take the phases Y | S,D in stage
        // ranges of CY stages; everything else is one range, i.e. the code as it was.
        const int CY = ex.uni(imax(1, imin(ex.smem().pool_n / 6, NS)));
        for (int y0 = 0; y0 < NS; y0 += CY) {
        const int y1 = imin(y0 + CY, NS);           // stages [y0, y1)
        if (y0 > 0) ex.barrier();                   // the S items of the range before are done with Y
        // ---------------------------------------------------------------- Y: items (k < N, i < 5)
        {
            constexpr int R = rounds_for(NTASK);
            const int items = (imin(y1, Nl) - y0) * NTASK;
            for (int base = 0; base < items; base += R * NT) {
                ex.wpar([&](int lane) {
                    double g[R][12], d[R][12], rr[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imax(imin(base + r * NT + lane, items - 1), 0), k = y0 + e / NTASK, i = e - (k - y0) * NTASK;
                        const double *g1 = G1 + (size_t)k * W1 + O_QW, *g2 = G2 + (size_t)k * W2;
                        rr[r] = gld(g2 + O_R + i);
#pragma unroll
                        for (int j = 0; j < 6; j++) { g[r][j] = gld(g2 + O_GQ + i * 6 + j); d[r][j] = gld(g1 + 6 + j); }
                        if (i == 4) {
#pragma unroll
                            for (int j = 0; j < 6; j++) { g[r][6 + j] = gld(g2 + O_GV + j); d[r][6 + j] = gld(g1 + 12 + j); }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int k = y0 + e / NTASK, i = e - (k - y0) * NTASK;
                            double v = rr[r];
#pragma unroll
                            for (int j = 0; j < 6; j++) v += g[r][j] * d[r][j];
                            if (i == 4) {
#pragma unroll
                                for (int j = 0; j < 6; j++) v += g[r][6 + j] * d[r][6 + j];
                            }
                            v *= P.w_task[i];
                            Y[(k - y0) * 6 + i] = v;
                            gst(G2 + (size_t)k * W2 + O_Y + i, v);
                        }
                    }
                });
            }
        }
        ex.barrier();
        PROF_ADD(PF_X2, ty);
        PROF_T0(tz);
        // ---------------------------------------------------------------- S: joint items (k, j < 6), two kinds
        {
            // (three joint items per lane and batch -- four wavefronts per simulation -- are 105 operands in flight: with the 256
            // registers of the two-simulations-per-CU build that spilled ten of them; two there, the other simulation covers the latency)
            constexpr int R = Ex::VGPR_BUDGET <= 256 && rounds_for(6) > 2 ? 2 : rounds_for(6);
            const int items = (y1 - y0) * 6;
            const double dt = P.dt, lm = P.lm;
            // bound part of a bounded component ci (value `val`, step `dv`): returns gt, updates rg, writes rd | rm | Gamma
            auto bounds = [&](int lane, int k, int ci, double val_, double dv, double l_lo, double l_hi, double t_lo, double t_hi,
                              double &rg) {
                // branch-free: an absent bound side holds lam = 0, t = 1, so its terms vanish by themselves; only rd needs a select
                const bool hc = ci < 6 ? k < Nl : (k >= 1 && k < Nl);
                const bool blo = hc && sm.bon[ci] != 0.0, bhi = hc && sm.bon[12 + ci] != 0.0;
                const double val = hc ? val_ : 0.0;
                const double ll = blo ? l_lo : 0.0, lu = bhi ? l_hi : 0.0;
                const double itl = fast_rcp(t_lo), itu = fast_rcp(t_hi);
                const double rdl = blo ? dv - (bnd_lo(P, ci) - val) - t_lo : 0.0;
                const double rdu = bhi ? (bnd_hi(P, ci) - val) - dv - t_hi : 0.0;
                const double rml = ll * t_lo, rmu = lu * t_hi;
                double gt = rg;
                rg -= ll; gt -= ll;
                double gam = ll * itl;
                gt += (rml + ll * rdl) * itl;
                rg += lu; gt += lu;
                gam += lu * itu;
                gt -= (rmu + lu * rdu) * itu;
                n_mu.at(lane) += rml + rmu;
                n_d.at(lane) = fmax(n_d.at(lane), fmax(fabs(rdl), fabs(rdu)));
                n_m.at(lane) = fmax(n_m.at(lane), fmax(fabs(rml), fabs(rmu)));
                double *g3 = G3 + (size_t)k * W3, *g2 = G2 + (size_t)k * W2;
                gst(g3 + O_RD + ci, rdl); gst(g3 + O_RD + 12 + ci, rdu);
                gst(g3 + O_RM + ci, rml); gst(g3 + O_RM + 12 + ci, rmu);
                gst(g2 + O_GAM + ci, gam);
                return gt;
            };
            // ---- one phase: the operands of the u/v items, the q items and the dynamics items of a batch all in flight first
            constexpr int RD = 2 * R;                 // twice as many dynamics items (12 per stage) as joint items (6)
            const int items_d = (y1 - y0) * NX;
            for (int base = 0; base < items; base += R * NT) {
                ex.wpar([&](int lane) {
                    double v[R][12], q[R][13], d[RD][5];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), k = y0 + e / 6, j = e - (k - y0) * 6, km = imax(k - 1, 0);
                        const double *g1 = G1 + (size_t)k * W1, *g2 = G2 + (size_t)k * W2, *gm = G1 + (size_t)km * W1;
                        v[r][0] = gld(g1 + O_U + j);      v[r][1] = gld(g1 + O_X + 6 + j);
                        v[r][2] = gld(g1 + O_QW + j);     v[r][3] = gld(g1 + O_QW + 12 + j);
                        v[r][4] = gld(g1 + O_QPI + j);    v[r][5] = gld(g1 + O_QPI + 6 + j);
                        v[r][6] = gld(gm + O_QPI + 6 + j);
                        v[r][7] = gld(g2 + O_GV + j);
                        v[r][8] = gld(g1 + O_QLAM + j);   v[r][9] = gld(g1 + O_QLAM + 12 + j);
                        v[r][10] = gld(g1 + O_QT + j);    v[r][11] = gld(g1 + O_QT + 12 + j);
                        q[r][0] = gld(g1 + O_X + j);  q[r][1] = gld(g1 + O_QW + 6 + j);
#pragma unroll
                        for (int i = 0; i < NTASK; i++) q[r][2 + i] = gld(g2 + O_GQ + i * 6 + j);
                        q[r][7] = v[r][4]; q[r][8] = gld(gm + O_QPI + j);
                        q[r][9] = gld(g1 + O_QLAM + 6 + j);  q[r][10] = gld(g1 + O_QLAM + 18 + j);
                        q[r][11] = gld(g1 + O_QT + 6 + j);   q[r][12] = gld(g1 + O_QT + 18 + j);
                    }
#pragma unroll
                    for (int r = 0; r < RD; r++) {
                        const int e = imin(2 * base + r * NT + lane, items_d - 1), k = y0 + e / NX, i = e - (k - y0) * NX, kn = imin(k + 1, Nl);
                        const double *dw = G1 + (size_t)k * W1 + O_QW;
                        d[r][0] = gld(dw + 6 + i);
                        d[r][1] = gld(dw + (i < 6 ? 12 + i : i - 6));
                        d[r][2] = gld(dw + (i < 6 ? i : i - 6));
                        d[r][3] = gld(G2 + (size_t)k * W2 + O_BD + i);
                        d[r][4] = gld(G1 + (size_t)kn * W1 + O_QW + 6 + i);
                    }
                    // u_j and v_j of joint j (they share u_j + du_j and v_j + dv_j)
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int k = y0 + e / 6, j = e - (k - y0) * 6;
                            const double du = v[r][2], dvv = v[r][3];
                            const double uj = v[r][0] + du, vj = v[r][1] + dvv;
                            const double c2 = P.w_qddot * P.cq[j] * P.cq[j];
                            double *g3 = G3 + (size_t)k * W3, *g2 = G2 + (size_t)k * W2;
                            // u_j (stat_cls<0>)
                            double rgu = 0.0;
                            if (k < Nl) {
                                rgu = dt * (2.0 * P.w_u * uj + c2 * (uj - vj));
                                rgu += P.b1[j] * v[r][4] + P.b2[j] * v[r][5];
                                rgu += dt * lm * du;
                            }
                            const double gtu = bounds(lane, k, j, v[r][0], du, v[r][8], v[r][9], v[r][10], v[r][11], rgu);
                            gst(g3 + O_RG + j, rgu); gst(g2 + O_GT + j, gtu);
                            // v_j (stat_cls<2>): no bounds
                            double rgv = 0.0;
                            if (k >= 1) {
                                if (k < Nl) {
                                    rgv = dt * (v[r][7] * Y[(k - y0) * 6 + 4] + c2 * (vj - uj));
                                    rgv += P.a12[j] * v[r][4] + P.a22[j] * v[r][5];
                                }
                                rgv += (k < Nl ? dt : 1.0) * lm * dvv;
                                rgv -= v[r][6];
                            }
                            gst(g3 + O_RG + 12 + j, rgv); gst(g2 + O_GT + 12 + j, rgv);
                            // q_j (stat_cls<1>); pi_k[j] is v[r][4]
                            const double dq = q[r][1];
                            double rg = 0.0;
                            if (k >= 1) {
                                if (k < Nl) {
                                    double s_ = 0.0;
#pragma unroll
                                    for (int i = 0; i < NTASK; i++) s_ += q[r][2 + i] * Y[(k - y0) * 6 + i];
                                    rg = dt * s_ + q[r][7];
                                }
                                rg += (k < Nl ? dt : 1.0) * lm * dq;
                                rg -= q[r][8];
                            }
                            const double gt = bounds(lane, k, 6 + j, q[r][0], dq, q[r][9], q[r][10], q[r][11], q[r][12], rg);
                            gst(g3 + O_RG + 6 + j, rg); gst(g2 + O_GT + 6 + j, gt);
                            n_g.at(lane) = fmax(n_g.at(lane), fmax(fmax(fabs(rgu), fabs(rgv)), fabs(rg)));
                        }
                    }
                    // dynamics residual of (k, i)
#pragma unroll
                    for (int r = 0; r < RD; r++) {
                        const int e = 2 * base + r * NT + lane;
                        if (e < items_d && e < 2 * (base + R * NT)) {
                            const int k = y0 + e / NX, i = e - (k - y0) * NX;
                            double vv = 0.0;
                            if (k < Nl) {
                                if (i < 6) vv = d[r][0] + P.a12[i] * d[r][1] + P.b1[i] * d[r][2];
                                else vv = P.a22[i - 6] * d[r][0] + P.b2[i - 6] * d[r][1];
                                vv += d[r][3] - d[r][4];
                                n_b.at(lane) = fmax(n_b.at(lane), fabs(vv));
                            }
                            gst(G2 + (size_t)k * W2 + O_RB + i, vv);
                        }
                    }
                });
            }
        }
        PROF_ADD(PF_X3, tz);
        }   // stage ranges
        ex.par([&](int lane) {
            ex.put_max(sm.red[0], lane, n_g.at(lane)); ex.put_max(sm.red[1], lane, n_b.at(lane));
            ex.put_max(sm.red[2], lane, n_d.at(lane)); ex.put_max(sm.red[3], lane, n_m.at(lane));
            ex.put_sum(sm.red[4], lane, n_mu.at(lane)); ex.put_sum(sm.red[5], lane, n_c.at(lane));
        });
        const double ng = ex.get_max(sm.red[0]), nb = ex.get_max(sm.red[1]), nd = ex.get_max(sm.red[2]), nm = ex.get_max(sm.red[3]);
        const double smu = ex.get_sum(sm.red[4]), nc = ex.get_sum(sm.red[5]);
        ex.par([&](int lane) {
            if ((lane & (WAVE - 1)) == 0) {
                sm.ret[0] = ng; sm.ret[1] = nb; sm.ret[2] = nd; sm.ret[3] = nm; sm.ret[4] = smu;
                if (mode == 0) sm.ret[5] = nc;
            }
        });
        PROF_ADD(PF_RES, t0);
    }

    // =========================================================================== Riccati passes
    // Factorisation sweep.  Lane l < 36 owns block position (a,b) = (l/6, l%6) of the 12x12
    // cost-to-go M = [[Mqq Mqv],[Mvq Mvv]] -- its four entries live in the lane's registers for
    // the whole sweep.  Because A and B have diagonal blocks, everything stage k-1 needs from
    // M = P_k at position (a,b) -- R~(a,b), S~(a,b), S~(a,6+b) and the A'MA block -- is a
    // combination of those four numbers, so M never travels through LDS.  Two phases per stage:
    //   B : LDL' of R~ (redundant in all lanes, right-looking), Kfb columns / R~^-1 columns one per
    //       lane; lanes 40..51: m~ = p_{k+1} + P_{k+1} rb_k
    //   CA: P_k block = A'MA + H_xx + Gamma_q - S~' Kfb  (-> registers, LDS record), then at once
    //       R~, S~ of stage k-1; lanes 40..51: h_u, p_k
    struct FactLane {
        double b1a, b2a, b1b, b2b, a12a, a22a, a12b, a22b;
        double hu_c;     // dt (2 w_u + w_qddot c_a^2 + lm) on the diagonal of R~, 0 off it
        double huv_c;    // dt w_qddot c_a^2 on the diagonals, 0 off them
        double lm_c;     // dt lm on the diagonals of H_qq, H_vv (levenberg_marquardt), 0 off them
        double mqq, mqv, mvq, mvv;   // P_{k+1} block (a,b)
        double qqq, qqv, qvq, qvv;   // A' P_{k+1} A block (a,b)
        int a, b;
        int oqq, oqv, ovv;           // packed offsets (tri) of the block's qq, qv, vv entries
    };

    // RES: the factor stays in LDS (see above): K, R~^-1 h_u, e, p go to the resident arrays, the chunk record and the
    // HBM store shrink to [P | w | R~^-1] (126 of 228 columns), and the first background wavefront accumulates Phi_c.
    // MODE 0: plain (everything through HBM, streaming sweeps follow); 1: RES; 2: SEG -- the plain factor record in HBM plus the
    // chunk transition matrices (chunks of SEG_L transitions) for the segment-resident sweeps.
    template <int MODE>
    MPC_PASS void fact_pass_t()
    {
        constexpr bool RES = MODE == 1, SEG = MODE == 2;
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(ex.smem().n_hor);
        constexpr int WR = 78;  // G2 columns [O_GQ, O_RB+12): GQ 0, GV 30, GAM 36, GT 48, RB 66
        constexpr int WF = RES ? W4 - O_PM : W4;   // LDS factor record of a chunk
        constexpr int FO = RES ? O_PM : 0;         // G4 column it starts at (`fac` below is biased by -FO: fac[O_*] works in both)
        const ResMap rm = res_map();
        const int CH = RES ? chunk_len_in(rm.scr_n, 2 * (WR + WF), 2 * WR) : chunk_len(2 * (WR + W4), 2 * WR);
        typename Ex::template PerLane<FactLane> fl;
        ex.seq([&](int lane) {
            FactLane &f = fl.at(lane);
            const int a = lane < 36 ? lane / 6 : 0, b = lane < 36 ? lane % 6 : 0;
            f.a = a; f.b = b;
            f.oqq = tri(imin(a, b), imax(a, b)); f.oqv = tri(a, 6 + b); f.ovv = tri(6 + imin(a, b), 6 + imax(a, b));
            f.b1a = P.b1[a]; f.b2a = P.b2[a]; f.b1b = P.b1[b]; f.b2b = P.b2[b];
            f.a12a = P.a12[a]; f.a22a = P.a22[a]; f.a12b = P.a12[b]; f.a22b = P.a22[b];
            const double c2 = P.dt * P.w_qddot * P.cq[a] * P.cq[a];
            f.hu_c = a == b ? P.dt * 2.0 * P.w_u + c2 + P.dt * P.lm : 0.0;
            f.huv_c = a == b ? c2 : 0.0;
            f.lm_c = a == b ? P.dt * P.lm : 0.0;
            f.mqq = f.mqv = f.mvq = f.mvv = 0.0;
            f.qqq = f.qqv = f.qvq = f.qvv = 0.0;
        });
        const double dw0 = P.dt * P.w_task[0], dw1 = P.dt * P.w_task[1], dw2 = P.dt * P.w_task[2], dw3 = P.dt * P.w_task[3],
                     dw4 = P.dt * P.w_task[4];
        // R~, S~ of stage `kd` from the lane's block of P_{kd+1}; also the A'MA block for stage kd
        // (gam_u = Gamma_u[a] of stage kd, loaded by the caller together with its other LDS reads: a load
        // inside the a == b branch would put a whole LDS round trip on the critical path)
        auto next_stage = [&](int lane, FactLane &f, double gam_u, int sb) {
            const double fq = f.b1a * f.mqq + f.b2a * f.mvq, fv = f.b1a * f.mqv + f.b2a * f.mvv;
            double r = fq * f.b1b + fv * f.b2b + f.hu_c;
            r += f.a == f.b ? gam_u : 0.0;
            sm.Rt[f.a * 6 + f.b] = r;
            double *St = sm.St2[sb];
            St[f.a * 12 + f.b] = fq;
            St[f.a * 12 + 6 + f.b] = fq * f.a12b + fv * f.a22b - f.huv_c;
            const double cq = f.a12a * f.mqq + f.a22a * f.mvq, cv = f.a12a * f.mqv + f.a22a * f.mvv;
            f.qqq = f.mqq;
            f.qqv = f.mqq * f.a12b + f.mqv * f.a22b;
            f.qvq = cq;
            f.qvv = cq * f.a12b + cv * f.a22b;
            (void)lane;
        };
        // Three roles per window (Ex::overlap3), chunk index ci counts down the horizon:
        //   wavefront 0   matrix recursion (P, K, R~^-1) of chunk ci; posts its progress every stage
        //   wavefront 1   follows it stage by stage with the vector recursion (w, h_u, p, R~^-1 h_u, e)
        //   wavefronts 2+ fetch the inputs of chunk ci+1, write the finished factor of chunk ci-1 back
        // Two chunks in flight (inputs and factor double-buffered).
        int sb = 0;
        const size_t buf = (size_t)(CH + 1) * WR + (size_t)CH * WF;
        auto vr_of = [&](int ci) { return ex.pool() + (size_t)(ci & 1) * buf; };
        auto vf_of = [&](int ci) { return vr_of(ci) + (size_t)(CH + 1) * WR; };
        auto k1_of = [&](int ci) { return Nl - ci * CH; };
        typename Ex::template PerLane<double> pr;   // wavefront 1, lanes < 12: p_{k+1}[lane]
        typename Ex::template PerLane<double> tr, wr;
        typename Ex::template PerLane<D2> ab;       // wavefront 1, lanes < 12: (a12, a22) of the lane's joint
        double b1r[6], b2r[6], a12r[6], a22r[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { b1r[i] = P.b1[i]; b2r[i] = P.b2[i]; a12r[i] = P.a12[i]; a22r[i] = P.a22[i]; }
        typename Ex::template PerLane<double> phi[12];   // RES, first background wavefront: row `lane` of Phi_c
        int vcur = 0;
        // vector recursion of chunk ci (stages k1c..k0c) -- runs on ONE wavefront (Ex::sub phases)
        auto vec_sweep = [&](int ci) {
            const int k1c = k1_of(ci), k0c = imax(k1c - CH + 1, 0), klc = imax(k0c - 1, 0);
            const double *vr = vr_of(ci);
            double *vf = vf_of(ci);
            for (int k = k1c; k >= k0c; k--) {
                const double *ric = vr + (size_t)(k - klc) * WR;
                const double *gt = ric + 48, *rbv = ric + 66;
                double *fac = vf + (size_t)(k - k0c) * WF - FO;
                double *kk = RES ? rm.K + (size_t)k * 72 : fac + O_K, *vhp = RES ? rm.VH + (size_t)k * 6 : fac + O_VH;
                double *ep = RES ? rm.E + (size_t)k * 12 : fac + O_E, *pvp = RES ? rm.P + (size_t)k * 12 : fac + O_PV;
                // P_{k+1}: next row of this chunk, or the lowest row of the chunk above (other buffer, intact)
                const double *Pn = (k < k1c ? fac + WF : vf_of(ci + 1) - FO) + O_PM;
                const int vnxt = vcur ^ 1;
                ex.await(&sm.prog, Nl - k);   // the matrices of stage k are in LDS
                if (k == Nl) {
                    ex.sub([&](int lane) {
                        if (lane < NX) {
                            const double v = gt[6 + lane];
                            pr.at(lane) = v; ex.share(sm.pv[vcur], lane, v);
                            pvp[lane] = v; fac[O_WV + lane] = 0.0; ep[lane] = 0.0;
                            const int j = lane % 6;
                            D2 c2; c2.x = P.a12[j]; c2.y = P.a22[j];
                            ab.at(lane) = c2;
                        }
                        if (lane < NU) vhp[lane] = 0.0;
                        if (lane == 0) ex.post(&sm.prog1, 0);
                    });
                    continue;
                }
                ex.sub([&](int lane) {
                    // t = p_{k+1} + P_{k+1} rb_k (lane i < 12 owns component i)
                    double t = 0.0, w = 0.0;
                    if (lane < NX) {
                        double w0 = 0.0, w1 = 0.0;   // row `lane` of the packed symmetric P_{k+1}
#pragma unroll
                        for (int j = 0; j < NX; j += 2) { w0 += Pn[tri_sym(lane, j)] * rbv[j]; w1 += Pn[tri_sym(lane, j + 1)] * rbv[j + 1]; }
                        w = w0 + w1;
                        t = pr.at(lane) + w;
                    }
                    tr.at(lane) = t; wr.at(lane) = w;
                    ex.share(sm.mt2[vcur], lane < NX ? lane : NX, t);
                });
                ex.sub([&](int lane) {
                    const double t = tr.at(lane), w = wr.at(lane);
                    double tv[12];
#pragma unroll
                    for (int i = 0; i < NX; i++) tv[i] = ex.gather(sm.mt2[vcur], i, t);
                    const double oq = ex.shr6(sm.mt2[vcur], lane, t);      // lanes 6..11: t[lane - 6]
                    if (lane < NX) {
                        const int j = lane;
                        double hu[6];
#pragma unroll
                        for (int m = 0; m < 6; m++) hu[m] = gt[m] + b1r[m] * tv[m] + b2r[m] * tv[6 + m];
                        const D2 c2 = ab.at(lane);
                        double pj = gt[6 + j] + (j < 6 ? t : c2.x * oq + c2.y * t);
                        double s0 = 0.0, s1 = 0.0;
#pragma unroll
                        for (int m = 0; m < 6; m += 2) { s0 += kk[m * 12 + j] * hu[m]; s1 += kk[(m + 1) * 12 + j] * hu[m + 1]; }
                        pj -= s0 + s1;
                        pr.at(lane) = pj; ex.share(sm.pv[vnxt], lane, pj);
                        pvp[j] = pj;
                        fac[O_WV + j] = w;
                        // what the forward sweep needs of h_u: R~^-1 h_u and e = rb - B R~^-1 h_u
                        const int i6 = j < 6 ? j : j - 6;
                        double v0 = 0.0, v1 = 0.0;
#pragma unroll
                        for (int m = 0; m < 6; m += 2) { v0 += fac[O_RI + i6 * 6 + m] * hu[m]; v1 += fac[O_RI + i6 * 6 + m + 1] * hu[m + 1]; }
                        const double vh = v0 + v1;
                        if (j < 6) vhp[j] = vh;
                        ep[j] = rbv[j] - (j < 6 ? P.b1[i6] : P.b2[i6]) * vh;
                    }
                    if (lane == 0) ex.post(&sm.prog1, Nl - k);
                });
                vcur = vnxt;
            }
        };
        // The windows are a PIPELINE without workgroup barriers (Ex::pipeline3): each role loops over the windows itself and
        // the hand-overs are LDS counters -- the recursion wavefront never waits for its followers to catch up at a window end
        // (that lag, ~1.3 us per window, was what a window cost):
        //   sm.prog / sm.prog1   stages finished by the matrix / the vector recursion (Nl - k)
        //   sm.flg[0]            input chunks landed in LDS, counted per background wavefront
        //   sm.flg[1]            factor buffers released (stored, and no longer read by the vector recursion), per background wavefront
        constexpr int NBW = Ex::BG_WAVES;
        ex.par([&](int lane) {
            if (lane == 0) { ex.post(&sm.prog, -1); ex.post(&sm.prog1, -1); ex.post(&sm.flg[0], NBW); ex.post(&sm.flg[1], 0); }
        });
        load_rect<WR, O_GQ, W2>(vr_of(0), ex.smem().w.G2, imax(imax(Nl - CH + 1, 0) - 1, 0), Nl);
        const int nwin = (Nl + CH) / CH;
        PROF_T0(ts);
        ex.pipeline3(nwin, [&](int ci) {
            const int k1 = k1_of(ci), k0 = imax(k1 - CH + 1, 0), kl = imax(k0 - 1, 0);
            double *vr = vr_of(ci), *vf = vf_of(ci);
            ex.await(&sm.flg[0], NBW * (ci + 1));          // this chunk's inputs have landed
            if (ci >= 2) ex.await(&sm.flg[1], NBW * ci);   // the factor buffer (that of chunk ci-2) is free
            for (int k = k1; k >= k0; k--) {
                const double *ric = vr + (size_t)(k - kl) * WR;
                const double *ricd = ric - WR;           // stage k-1 (valid for k >= 1)
                double *fac = vf + (size_t)(k - k0) * WF - FO;
                double *kk = RES ? rm.K + (size_t)k * 72 : fac + O_K;   // K_k: the resident array, or the chunk record
                const double *gam = ric + 36;
                if (k == Nl) {
                    // terminal stage: no cost, no bounds -> P_N = lm I (0 without levenberg_marquardt) ; R~, S~ of stage N-1
                    ex.seq([&](int lane) {
                        const double lmN = ex.smem().P.lm;
                        for (int e = lane; e < NPM; e += WAVE) fac[O_PM + e] = 0.0;
                        if (lane < NX) fac[O_PM + tri(lane, lane)] = lmN;   // (same wavefront, program order: after the zero fill)
                        if (lane < 36) {
                            FactLane &f = fl.at(lane);
                            f.mqv = f.mvq = 0.0;
                            f.mqq = f.mvv = f.a == f.b ? lmN : 0.0;
                            next_stage(lane, f, ricd[36 + f.a], sb);
                        }
                        if (lane == 0) ex.post(&sm.prog, 0);
                    });
                    continue;
                }
                // ---- B: LDL' (right-looking) + one right-hand side per lane
#ifndef MPCB_DIAG_NO_B
                ex.seq([&](int lane) {
                    const double *St = sm.St2[sb];
                    double A_[6][6];
#pragma unroll
                    for (int i = 0; i < 6; i++)
#pragma unroll
                        for (int j = 0; j <= i; j++) A_[i][j] = sm.Rt[i * 6 + j];
                    double dinv[6];
#pragma unroll
                    for (int j = 0; j < 6; j++) {
#ifdef MPCB_DIAG_NO_LDL
                        dinv[j] = A_[j][j]; continue;
#endif
                        dinv[j] = fast_rcp(A_[j][j]);
                        double lj[6];
#pragma unroll
                        for (int i = j + 1; i < 6; i++) lj[i] = A_[i][j] * dinv[j];
#pragma unroll
                        for (int i = j + 1; i < 6; i++)
#pragma unroll
                            for (int r = j + 1; r <= i; r++) A_[i][r] -= lj[i] * A_[r][j];
#pragma unroll
                        for (int i = j + 1; i < 6; i++) A_[i][j] = lj[i];   // L(i,j)
                    }
                    if (lane < 18) {
                        double x[6];
                        const int col = lane < 12 ? lane : 0;
#pragma unroll
                        for (int i = 0; i < 6; i++) x[i] = St[i * 12 + col];   // unconditional loads, then select
#pragma unroll
                        for (int i = 0; i < 6; i++) x[i] = lane < 12 ? x[i] : (i == lane - 12 ? 1.0 : 0.0);
#pragma unroll
                        for (int j = 0; j < 6; j++) {          // L y = rhs, column oriented
#pragma unroll
                            for (int i = j + 1; i < 6; i++) x[i] -= A_[i][j] * x[j];
                        }
#pragma unroll
                        for (int i = 0; i < 6; i++) x[i] *= dinv[i];
#pragma unroll
                        for (int j = 5; j >= 0; j--) {         // L' x = y, column oriented
#pragma unroll
                            for (int i = 0; i < j; i++) x[i] -= A_[j][i] * x[j];
                        }
                        if (lane < 12) {
#pragma unroll
                            for (int i = 0; i < 6; i++) kk[i * 12 + lane] = x[i];
                        } else {
#pragma unroll
                            for (int i = 0; i < 6; i++) fac[O_RI + i * 6 + (lane - 12)] = x[i];
                        }
                    }
                });
#endif
                // ---- CA: P_k block, then R~/S~ of stage k-1
                ex.seq([&](int lane) {
#ifdef MPCB_DIAG_NO_CA
                    if (lane == 0) ex.post(&sm.prog, Nl - k);
                    return;
#endif
                    if (lane < 36) {
                        FactLane &f = fl.at(lane);
                        if (k > 0) {
                            const double *St = sm.St2[sb];
                            const double *gq = ric, *gv = ric + 30;
                            // Operands in the order they can be used (LDS returns in issue order): the linearisation and S~ do not
                            // depend on phase B, K does -- its write -> read round trip runs under the K-independent part of P_k.
                            const double ga0 = gq[f.a], ga1 = gq[6 + f.a], ga2 = gq[12 + f.a], ga3 = gq[18 + f.a], ga4 = gq[24 + f.a];
                            const double gb0 = gq[f.b], gb1 = gq[6 + f.b], gb2 = gq[12 + f.b], gb3 = gq[18 + f.b], gb4 = gq[24 + f.b];
                            const double gva = gv[f.a], gvb = gv[f.b];
                            const double gam_q = gam[6 + f.a], gam_u = ricd[36 + f.a];   // unconditional (see next_stage)
                            double sa[6], sva[6], kb[6], kvb[6];
#pragma unroll
                            for (int m = 0; m < 6; m++) { sa[m] = St[m * 12 + f.a]; sva[m] = St[m * 12 + 6 + f.a]; }
#pragma unroll
                            for (int m = 0; m < 6; m++) { kb[m] = kk[m * 12 + f.b]; kvb[m] = kk[m * 12 + 6 + f.b]; }
#if defined(__HIP_DEVICE_COMPILE__)
                            asm volatile("" ::: "memory");    // keep that issue order
#endif
                            double pqq = f.qqq + (dw0 * ga0 * gb0 + dw1 * ga1 * gb1 + dw2 * ga2 * gb2 + dw3 * ga3 * gb3 + dw4 * ga4 * gb4) + f.lm_c;
                            double pqv = f.qqv + dw4 * ga4 * gvb;
                            double pvq = f.qvq + dw4 * gva * gb4;
                            double pvv = f.qvv + dw4 * gva * gvb + (f.huv_c + f.lm_c);
                            pqq += f.a == f.b ? gam_q : 0.0;
#pragma unroll
                            for (int m = 0; m < 6; m++) {
                                pqq -= sa[m] * kb[m]; pqv -= sa[m] * kvb[m];
                                pvq -= sva[m] * kb[m]; pvv -= sva[m] * kvb[m];
                            }
                            f.mqq = pqq; f.mqv = pqv; f.mvq = pvq; f.mvv = pvv;
                            // R~, S~ of stage k-1 first (the next phase B waits for them), then the record of P_k
                            next_stage(lane, f, gam_u, sb ^ 1);
                            // packed upper triangle: (a,b) of the qq and vv blocks only for a <= b, the qv block in full
                            if (f.a <= f.b) { fac[O_PM + f.oqq] = pqq; fac[O_PM + f.ovv] = pvv; }
                            fac[O_PM + f.oqv] = pqv;
                        }
                    }
                    if (lane == 0) ex.post(&sm.prog, Nl - k);   // K_k, R~^-1_k, P_k are in LDS
                });
                sb ^= 1;
            }
            }, [&](int ci) {
#ifndef MPCB_DIAG_NO_VEC
                vec_sweep(ci);   // wavefront 1: vector recursion of this chunk, one stage behind the matrices
#endif
            }, [&](int ci, int lane, auto nl) {
                constexpr int NL = decltype(nl)::value;
                const int k1 = k1_of(ci), k0 = imax(k1 - CH + 1, 0);
                const int nk1 = k0 - 1, nk0 = imax(nk1 - CH + 1, 0), nkl = imax(nk0 - 1, 0);   // chunk ci+1
                const int pk1 = k1 + CH, pk0 = k1 + 1;                                          // chunk ci-1
                // chunk ci+1's inputs go where chunk ci-1's were: both recursions must be done with chunk ci-1
                if (ci > 0) ex.await(&sm.prog1, Nl - pk0);
                if (nk1 >= 0) copy_lanes<WR, O_GQ, W2, WR, true, NL>(vr_of(ci + 1), ex.smem().w.G2, nkl, nk1, lane);
                if ((lane & (WAVE - 1)) == 0) ex.post_add(&sm.flg[0], 1);
                if (ci > 0) copy_lanes<WF, FO, W4, WF, false, NL>(vf_of(ci - 1), ex.smem().w.G4, pk0, pk1, lane);
                // chunk ci-1's factor buffer: stored now; the vector recursion reads its lowest row (P) once more, at
                // the first stage of chunk ci
                ex.await(&sm.prog1, Nl - k1);
                if ((lane & (WAVE - 1)) == 0) ex.post_add(&sm.flg[1], 1);
                if constexpr (RES || SEG) {
                    // copies issued: the first background wavefront follows the recursion and accumulates the chunk
                    // transition matrix Phi_c <- Phi_c Acl_k (lane i < 12 holds row i in registers), k descending
#ifndef MPCB_DIAG_NO_PHI
                    if (lane < WAVE) {
                        const int L = SEG ? (reg_ok() ? RL : seg_map().L) : rm.L;
                        for (int k = imin(k1, Nl - 1); k >= k0; k--) {
                            ex.await(&sm.prog, Nl - k);
                            const double *kk = RES ? rm.K + (size_t)k * 72 : vf_of(ci) + (size_t)(k - k0) * WF - FO + O_K;
                            const int c = k / L;
                            const bool first = ex.uni(k == imin((c + 1) * L, Nl) - 1), last = ex.uni(k == c * L);
                            // K_k is the same for every lane: ONE 16-byte LDS read per lane (lanes 0..35 cover the 72 entries),
                            // then each entry reaches the arithmetic as a scalar (v_readlane) -- 36 broadcast reads per stage
                            // here held up the recursion wavefront's own LDS reads (measured: +8 us per sweep)
                            const D2 kv2 = reinterpret_cast<const D2 *>(kk)[(lane & (WAVE - 1)) < 36 ? (lane & (WAVE - 1)) : 0];
                            auto Kat = [&](int idx) { return ex.lane_value(kk, idx, (idx & 1) ? kv2.y : kv2.x, idx >> 1); };
                            {
                                double ph[12], g[6], nw[12];
#pragma unroll
                                for (int j = 0; j < 12; j++) ph[j] = first ? (j == lane ? 1.0 : 0.0) : phi[j].at(lane);
#pragma unroll
                                for (int m = 0; m < 6; m++) g[m] = ph[m] * b1r[m] + ph[6 + m] * b2r[m];
#pragma unroll
                                for (int j = 0; j < 6; j++) {
                                    double s0 = ph[j], s1 = a12r[j] * ph[j] + a22r[j] * ph[6 + j];
#pragma unroll
                                    for (int m = 0; m < 6; m++) { s0 -= g[m] * Kat(m * 12 + j); s1 -= g[m] * Kat(m * 12 + 6 + j); }
                                    nw[j] = s0; nw[6 + j] = s1;
                                }
#pragma unroll
                                for (int j = 0; j < 12; j++) phi[j].at(lane) = nw[j];
                                if (last && lane < NX) {
                                    if (SEG) {
#pragma unroll
                                        for (int j = 0; j < 12; j++) gst(ex.smem().w.PH + (size_t)c * 144 + lane * 12 + j, nw[j]);
                                    } else {
#pragma unroll
                                        for (int j = 0; j < 12; j++) rm.PHI[(size_t)c * 144 + lane * 12 + j] = nw[j];
                                    }
                                }
                            }
                        }
                    }
#endif
                }
            });
        PROF_ADD(PF_SEQ_FACT, ts);
        {   // last chunk: nothing left to hide the store behind
            const int k1 = k1_of(nwin - 1);
            copy_rect<WF, FO, W4, WF, false>(vf_of(nwin - 1), ex.smem().w.G4, 0, k1);
        }
        PROF_ADD(PF_FACT, t0);
    }

    // Centering-corrector right-hand side (HPIPM compute_centering_correction: rm <- lam*t +
    // dlam_aff*dt_aff - sigma*mu, rebuild gt) followed by the backward SOLVE sweep on the
    // existing factorisation.  With Acl = A - B Kfb the vector recursion is
    //     p_k = (gt_x - Kfb' gt_u) + Acl' (p_{k+1} + P_{k+1} rb_k),
    // so everything except one 12-vector update per stage is computed for the whole chunk in
    // parallel (before: P_{k+1} rb_k and gt_x - Kfb' gt_u; after: h_u = gt_u + B'(p_{k+1} + P_{k+1} rb_k)).
    MPC_PASS void corrector_bwd_pass(double sigma_mu)
    {
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(ex.smem().n_hor);
        // LDS record of a stage: inputs | scratch | outputs.  Two chunks in flight (see fact_pass).
        constexpr int WLT = 50;                  // QLAM | QT (48 used; LDS strides of 48 would alias every 2nd stage on the banks)
        constexpr int L3 = 90;                   // RG 0 | RD 18 | DLAM 42 | DT 66  (G3 without RM and the step)
        constexpr int C_DLAM = 42, C_DT = 66;
        constexpr int WGR = 30;                  // GT 0 (rebuilt in place) | RB 18
        constexpr int WK = 72, WW = 50;          // Kfb ; w = P_{k+1} rb_k (12) | R~^-1 (36)  (adjacent in G4)
        constexpr int WC = 12;                   // c_k: the part of p_k that does not depend on p_{k+1}
        constexpr int WRM = 24, WHP = 30;        // outputs: RM ; R~^-1 h_u (6) | e (12) | p (12)  (adjacent in G4)
        constexpr int C_PV = 18;
        constexpr int PER = WLT + L3 + WGR + WK + WW + WC + WRM + WHP;
        const int CH = chunk_len(2 * PER, 0);
        typename Ex::template PerLane<D2> ab;       // lanes < 12: (a12, a22) of the lane's joint
        typename Ex::template PerLane<double> pr;   // lanes < 12: p_{k+1}[lane], carried from stage to stage
        double b1r[6], b2r[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { b1r[i] = P.b1[i]; b2r[i] = P.b2[i]; }
        ex.seq([&](int lane) {
            const int j = lane % 6;
            D2 v; v.x = P.a12[j]; v.y = P.a22[j];
            ab.at(lane) = v;
            pr.at(lane) = 0.0;
        });
        auto carve = [&](int sel, double *&vlt, double *&v3, double *&vgr, double *&vk, double *&vw, double *&vc, double *&orm,
                         double *&ohp) {
            vlt = ex.pool() + (size_t)sel * CH * PER;
            v3 = vlt + (size_t)CH * WLT; vgr = v3 + (size_t)CH * L3; vk = vgr + (size_t)CH * WGR;
            vw = vk + (size_t)CH * WK; vc = vw + (size_t)CH * WW; orm = vc + (size_t)CH * WC; ohp = orm + (size_t)CH * WRM;
        };
        int cur = 0, bsel = 0;
        {
            double *vlt, *v3, *vgr, *vk, *vw, *vc, *orm, *ohp;
            carve(0, vlt, v3, vgr, vk, vw, vc, orm, ohp);
            const int f0 = imax(Nl - CH + 1, 0);
            copies([&](int lane, auto nl) {
                constexpr int NL = decltype(nl)::value;
                copy_lanes<48, O_QLAM, W1, WLT, true, NL>(vlt, ex.smem().w.G1, f0, Nl, lane);
                copy_lanes<42, 0, W3, L3, true, NL>(v3, ex.smem().w.G3, f0, Nl, lane);
                copy_lanes<48, O_DLAM, W3, L3, true, NL>(v3 + C_DLAM, ex.smem().w.G3, f0, Nl, lane);
                copy_lanes<WGR, O_GT, W2, WGR, true, NL>(vgr, ex.smem().w.G2, f0, Nl, lane);
                copy_lanes<WK, O_K, W4, WK, true, NL>(vk, ex.smem().w.G4, f0, Nl, lane);
                copy_lanes<48, O_WV, W4, WW, true, NL>(vw, ex.smem().w.G4, f0, Nl, lane);
            });
        }
        for (int k1 = Nl; k1 >= 0; k1 -= CH, bsel ^= 1) {
            const int k0 = imax(k1 - CH + 1, 0);
            double *vlt, *v3, *vgr, *vk, *vw, *vc, *orm, *ohp;
            carve(bsel, vlt, v3, vgr, vk, vw, vc, orm, ohp);
            double *nlt, *n3, *ngr, *nk, *nw, *nc_, *prm, *php;   // other buffer: next inputs, previous outputs
            carve(bsel ^ 1, nlt, n3, ngr, nk, nw, nc_, prm, php);
            const int nk1 = k0 - 1, nk0 = imax(nk1 - CH + 1, 0);
            // centering corrector: rm and the condensed gradient gt of every bounded component
            ex.par([&](int lane) {
                const int rows = k1 - k0 + 1;
                for (int e = lane; e < rows * NB; e += NT) {
                    const int s = e / NB, j = e - s * NB, k = k0 + s;
                    const double *lt = vlt + (size_t)s * WLT;
                    const double *r3 = v3 + (size_t)s * L3;
                    double *rmo = orm + (size_t)s * WRM;
                    // all loads first, decisions on registers only: an LDS load inside a branch costs a round trip
                    // each (one wave per SIMD: nothing hides it).  Unbounded components hold lam = 0, t = 1.
                    const bool hc = has_comp(Nl, k, j);
                    const bool blo = hc && bnd_lo(P, j) > -BOUND_INF, bhi = hc && bnd_hi(P, j) < BOUND_INF;
                    const double ll = lt[j], tl = lt[24 + j], lu = lt[12 + j], tu = lt[36 + j];
                    const double dll = r3[C_DLAM + j], dtl = r3[C_DT + j], dlu = r3[C_DLAM + 12 + j], dtu = r3[C_DT + 12 + j];
                    const double rdl = r3[18 + j], rdu = r3[18 + 12 + j];
                    double gt = r3[j];
                    double rml, rmu;
                    gt += ipm::corrector_side(blo, ll, tl, dll, dtl, rdl, sigma_mu, rml);
                    gt -= ipm::corrector_side(bhi, lu, tu, dlu, dtu, rdu, sigma_mu, rmu);
                    if (hc) vgr[(size_t)s * WGR + j] = gt;
                    rmo[j] = rml; rmo[12 + j] = rmu;
                }
            });
            // c_k = gt_x + A' w - Kfb' (gt_u + B' w): all of p_k = c_k + Acl' p_{k+1} that is known up front
            ex.par([&](int lane) {
                const int rows = k1 - k0 + 1;
                for (int e = lane; e < rows * NX; e += NT) {
                    const int s = e / NX, j = e - s * NX, k = k0 + s;
                    const double *gt = vgr + (size_t)s * WGR, *w = vw + (size_t)s * WW, *kf = vk + (size_t)s * WK;
                    double v = 0.0;
                    if (k < Nl) {
                        v = gt[6 + j] + (j < 6 ? w[j] : P.a12[j - 6] * w[j - 6] + P.a22[j - 6] * w[j]);
#pragma unroll
                        for (int m = 0; m < 6; m++) v -= kf[m * 12 + j] * (gt[m] + P.b1[m] * w[m] + P.b2[m] * w[6 + m]);
                    }
                    vc[(size_t)s * WC + j] = v;
                }
            });
            PROF_T0(ts);
            ex.overlap([&]() {
            for (int k = k1; k >= k0; k--) {
                const double *gt = vgr + (size_t)(k - k0) * WGR;
                const double *cv = vc + (size_t)(k - k0) * WC;
                const double *kf = vk + (size_t)(k - k0) * WK;
                double *hp = ohp + (size_t)(k - k0) * WHP;
                const int nxt = cur ^ 1;
                if (k == Nl) {
                    ex.seq([&](int lane) {
                        if (lane < NX) {
                            const double v = gt[6 + lane];
                            pr.at(lane) = v; ex.share(sm.pv[cur], lane, v);
                            hp[C_PV + lane] = v;
                            hp[6 + lane] = 0.0;
                        }
                        if (lane < NU) hp[lane] = 0.0;
                    });
                    continue;
                }
                // p_k = c_k + Acl' p_{k+1}: p_{k+1} travels lane to lane in registers (no LDS round trip)
                ex.seq([&](int lane) {
                    // this stage's column of K and c entry first (their LDS latency runs under the gather)
                    const int jc = lane < NX ? lane : 0;
                    double kc[6];
#pragma unroll
                    for (int i = 0; i < 6; i++) kc[i] = kf[i * 12 + jc];
                    const double ck = cv[jc];
                    const double mine = pr.at(lane);
                    double pn[12];
#pragma unroll
                    for (int i = 0; i < NX; i++) pn[i] = ex.gather(sm.pv[cur], i, mine);
                    const double oq = ex.shr6(sm.pv[cur], lane, mine);   // lanes 6..11: p_{k+1}[lane - 6]
                    double acc0 = 0.0, acc1 = 0.0;   // (in every lane, see forward_step_pass)
#pragma unroll
                    for (int i = 0; i < 6; i += 2) {
                        acc0 += kc[i] * (b1r[i] * pn[i] + b2r[i] * pn[6 + i]);
                        acc1 += kc[i + 1] * (b1r[i + 1] * pn[i + 1] + b2r[i + 1] * pn[7 + i]);
                    }
                    const D2 c2 = ab.at(lane);
                    const double at = lane < 6 ? mine : c2.x * oq + c2.y * mine;
                    const double pj = ck + (at - (acc0 + acc1));
                    pr.at(lane) = pj;                       // unconditional (see forward_step_pass)
                    if (lane < NX) {
                        ex.share(sm.pv[nxt], lane, pj);
                        hp[C_PV + lane] = pj;
                    }
                });
                cur = nxt;
            }
            }, [&](int lane, auto nl) {
                constexpr int NL = decltype(nl)::value;
                if (nk1 >= 0) {
                    copy_lanes<48, O_QLAM, W1, WLT, true, NL>(nlt, ex.smem().w.G1, nk0, nk1, lane);
                    copy_lanes<42, 0, W3, L3, true, NL>(n3, ex.smem().w.G3, nk0, nk1, lane);
                    copy_lanes<48, O_DLAM, W3, L3, true, NL>(n3 + C_DLAM, ex.smem().w.G3, nk0, nk1, lane);
                    copy_lanes<WGR, O_GT, W2, WGR, true, NL>(ngr, ex.smem().w.G2, nk0, nk1, lane);
                    copy_lanes<WK, O_K, W4, WK, true, NL>(nk, ex.smem().w.G4, nk0, nk1, lane);
                    copy_lanes<48, O_WV, W4, WW, true, NL>(nw, ex.smem().w.G4, nk0, nk1, lane);
                }
                if (k1 < Nl) {
                    copy_lanes<WRM, O_RM, W3, WRM, false, NL>(prm, ex.smem().w.G3, k1 + 1, k1 + CH, lane);
                    copy_lanes<WHP, O_VH, W4, WHP, false, NL>(php, ex.smem().w.G4, k1 + 1, k1 + CH, lane);
                }
            });
            PROF_ADD(PF_SEQ_BWD, ts);
            // chunk-parallel: h_u,k = gt_u + B'(p_{k+1} + w_k); the forward sweep wants it as R~^-1 h_u and
            // e = rb - B R~^-1 h_u (lane <-> (stage, state component j); h_u is formed redundantly)
            ex.par([&](int lane) {
                const int rows = k1 - k0 + 1;
                for (int e = lane; e < rows * NX; e += NT) {
                    const int s = e / NX, j = e - s * NX, k = k0 + s;
                    if (k >= Nl) continue;
                    const double *gt = vgr + (size_t)s * WGR, *w = vw + (size_t)s * WW, *ri = w + 12;
                    // p_{k+1}: next row of this chunk, or the lowest row of the previous chunk (other buffer)
                    const double *pn = (s + 1 < rows ? ohp + (size_t)(s + 1) * WHP : php) + C_PV;
                    const int i6 = j < 6 ? j : j - 6;
                    double v0 = 0.0, v1 = 0.0;
#pragma unroll
                    for (int m = 0; m < 6; m += 2) {
                        const double h0 = gt[m] + P.b1[m] * (pn[m] + w[m]) + P.b2[m] * (pn[6 + m] + w[6 + m]);
                        const double h1 = gt[m + 1] + P.b1[m + 1] * (pn[m + 1] + w[m + 1]) + P.b2[m + 1] * (pn[7 + m] + w[7 + m]);
                        v0 += ri[i6 * 6 + m] * h0; v1 += ri[i6 * 6 + m + 1] * h1;
                    }
                    const double vh = v0 + v1;
                    double *hp = ohp + (size_t)s * WHP;
                    if (j < 6) hp[j] = vh;
                    hp[6 + j] = gt[18 + j] - (j < 6 ? P.b1[i6] : P.b2[i6]) * vh;
                }
            });
            if (k0 == 0) {
                copies([&](int lane, auto nl) {
                    constexpr int NL = decltype(nl)::value;
                    copy_lanes<WRM, O_RM, W3, WRM, false, NL>(orm, ex.smem().w.G3, k0, k1, lane);
                    copy_lanes<WHP, O_VH, W4, WHP, false, NL>(ohp, ex.smem().w.G4, k0, k1, lane);
                });
            }
        }
        PROF_ADD(PF_BWD, t0);
    }

    // Forward sweep: du = -Kfb dx - Rinv h_u ; dx+ = A dx + B du + rb ; dpi = P dx+ + p, then
    // dt, dlam from the primal step (HPIPM compute_lam_t), the largest feasible step and the
    // three sums S_i with mu(alpha) * nc = S0 + alpha S1 + alpha^2 S2.  Only the state recursion
    // dx_{k+1} = A dx_k - B (Kfb dx_k + Rinv h_u) + rb_k is sequential (one phase per stage);
    // Rinv h_u and e = rb - B Rinv h_u come with the factor (written by whoever produced h_u).
    // FAST: see fwd_resident (the follower forms the candidate's slacks instead of dlam / dt; returns 1.0 / 0.0).
    template <bool AFFINE, bool FAST = false>
    // (the centering sums S0, S1, S2 stay in sm.red[1..3][0]: read them with ex.get1)
    MPC_PASS double forward_step_pass()
    {
        static_assert(!FAST || !AFFINE, "the fast path takes the full step");
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(ex.smem().n_hor);
        constexpr int WLT = 50, WR = 50, WO = 78;   // LDS row strides; 48 columns each are used (48 would alias every 2nd stage on the LDS banks)
        // the affine (predictor) sweep only feeds the step length and the centering sums: it needs
        // K, R~^-1 h_u, e (no p, no P) and leaves only dlam, dt behind for the corrector
        constexpr int LF = AFFINE ? W4_AFF : W4_FWD;
        // Three roles per window ci (Ex::overlap3), two chunks in flight:
        //   wavefront 0   state recursion dx of chunk ci; publishes its progress (sm.prog)
        //   wavefront 1   follows it in blocks of BLK stages: du, dpi, dlam, dt, step length, centering sums
        //   wavefronts 2+ fetch the inputs of chunk ci+1, write the step of chunk ci-1 back
        constexpr int PER = LF + WLT + WR + WO;
#ifndef MPCB_FWD_BLK
#define MPCB_FWD_BLK 4
#endif
        constexpr int BLK = MPCB_FWD_BLK;          // stages per follower block: BLK x 12 bounded components <= 64 lanes
        const int CH = chunk_len(2 * PER, 0);
        const int NCH = (Nl + CH) / CH;            // number of chunks
        double *const pool = ex.pool();
        typename Ex::template PerLane<D2> ab, bb;   // lanes < 12: (a12, a22), (b1, b2) of the lane's joint
        typename Ex::template PerLane<double> dxr;  // lanes < 12: dx_k[lane], carried from stage to stage
        typename Ex::template PerLane<double> r_al, r_a0, r_a1, r_a2;   // wavefront 1: running min / sums
        ex.par([&](int lane) {
            const int j = (lane & (WAVE - 1)) % 6;
            D2 v; v.x = P.a12[j]; v.y = P.a22[j];
            ab.at(lane) = v;
            D2 u; u.x = P.b1[j]; u.y = P.b2[j];
            bb.at(lane) = u;
            dxr.at(lane) = 0.0;                    // dx_0 = 0: x_0 is pinned by the init pass
            if (lane < NX) ex.share(sm.dx[0], lane, 0.0);
            r_al.at(lane) = 1.0; r_a0.at(lane) = 0.0; r_a1.at(lane) = 0.0; r_a2.at(lane) = 0.0;
            if (lane == 0) ex.post(&sm.prog, -1);
        });
        int cur = 0;
        {
            double *q4 = pool, *qlt = q4 + (size_t)CH * LF, *qr = qlt + (size_t)CH * WLT;
            const int e1 = imin(CH - 1, Nl);
            copies([&](int lane, auto nl) {
                constexpr int NL = decltype(nl)::value;
                copy_lanes<LF, 0, W4, LF, true, NL>(q4, ex.smem().w.G4, 0, e1, lane);
                if (FAST) copy_lanes<18, 0, W1, WLT, true, NL>(qlt, ex.smem().w.G1, 0, e1, lane);   // X | U of the NLP iterate
                else {
                copy_lanes<48, O_QLAM, W1, WLT, true, NL>(qlt, ex.smem().w.G1, 0, e1, lane);
                copy_lanes<48, O_RD, W3, WR, true, NL>(qr, ex.smem().w.G3, 0, e1, lane);
                }
            });
        }
        for (int ci = 0; ci < NCH; ci++) {
            const int k0 = ci * CH, k1 = imin(k0 + CH - 1, Nl);
            double *v4 = pool + (size_t)(ci & 1) * CH * PER;   // rows k0..k1, G4
            double *vlt = v4 + (size_t)CH * LF;       // QLAM | QT
            double *vr = vlt + (size_t)CH * WLT;      // RD | RM
            double *vo = vr + (size_t)CH * WR;        // DW | DPI | DLAM | DT  (out)
            // the other buffer: inputs of the next chunk, output of the previous one
            double *n4 = pool + (size_t)((ci + 1) & 1) * CH * PER, *nlt = n4 + (size_t)CH * LF,
                   *nr = nlt + (size_t)CH * WLT, *po = nr + (size_t)CH * WR;
            const int nk0 = k1 + 1, nk1 = imin(nk0 + CH - 1, Nl);
            PROF_T0(ts);
            ex.overlap3([&]() {
            for (int k = k0; k <= k1; k++) {
                const int nxt = cur ^ 1;
                const double *fac = v4 + (size_t)(k - k0) * LF;
                double *o = vo + (size_t)(k - k0) * WO;
                // dx_{k+1} = e_k + A dx_k - B K dx_k: dx_k travels lane to lane in registers
                ex.seq([&](int lane) {
                    // this stage's row of K and e entry first: their LDS latency runs under the 24 v_readlane
                    // of the gather instead of after it (all lanes load, from clamped in-range addresses)
                    const int lc = lane < NX ? lane : 0, i = lc < 6 ? lc : lc - 6;
                    double kr[12];
#pragma unroll
                    for (int j = 0; j < NX; j++) kr[j] = fac[O_K + i * 12 + j];
                    const double ek = fac[O_E + lc];
                    const double own = dxr.at(lane);
                    double dxv[12];
#pragma unroll
                    for (int j = 0; j < NX; j++) dxv[j] = ex.gather(sm.dx[cur], j, own);
                    const double ov = ex.shl6(sm.dx[cur], lane, own);   // lanes 0..5: dx_k[6 + lane]
                    // (the dot product runs in every lane: inside the lane < 12 branch the compiler sinks the loads
                    // into the branch, behind the gather)
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int j = 0; j < NX; j += 2) {
                        s0 += kr[j] * dxv[j];
                        s1 += kr[j + 1] * dxv[j + 1];
                    }
                    const double kd = s0 + s1;
                    // the register update is unconditional too (lanes >= 12 carry a harmless value): anything the
                    // compiler can prove is only needed by lanes < 12 it sinks into that branch, loads included
                    const D2 a = ab.at(lane), b = bb.at(lane);
                    const double v = ek + (lane < 6 ? own + a.x * ov - b.x * kd : a.y * own - b.y * kd);
                    dxr.at(lane) = v;   // (also at k = N, where it is never used again)
                    if (lane < NX) {
                        o[6 + lane] = own;   // dx_k
                        if (k < Nl) ex.share(sm.dx[nxt], lane, v);
                    }
                    if (lane == 0) ex.post(&sm.prog, k);   // dx_k is in LDS (an unconditional post is cheaper than a modulo)
                });
                if (k < Nl) cur = nxt;
            }
            }, [&]() {
                // wavefront 1 follows the recursion block by block: du_k = -(R~^-1 h_u + K dx_k), then
                // dlam, dt (HPIPM compute_lam_t), largest feasible step, centering sums -- one phase,
                // lane <-> (stage of the block, bounded component j): j < 6 input, j >= 6 joint position
                for (int kb = k0; kb <= k1; kb += BLK) {
                    const int ke = imin(kb + BLK - 1, k1), rows = ke - kb + 1;
                    ex.await(&sm.prog, ke);
                    ex.sub([&](int lane) {
                        double al = r_al.at(lane), a0 = r_a0.at(lane), a1 = r_a1.at(lane), a2 = r_a2.at(lane);
                        if (lane < rows * NB) {
                            const int s = lane / NB, j = lane - s * NB, k = kb + s;
                            const double *fac = v4 + (size_t)(k - k0) * LF;
                            const double *lt = vlt + (size_t)(k - k0) * WLT, *r = vr + (size_t)(k - k0) * WR;
                            double *o = vo + (size_t)(k - k0) * WO;
                            const double *dxk = o + 6;
                            double dv;
                            if (j < 6) {
                                dv = 0.0;
                                if (k < Nl) {
                                    double s0 = fac[O_VH + j], s1 = 0.0;
#pragma unroll
                                    for (int i = 0; i < NX; i += 2) { s0 += fac[O_K + j * 12 + i] * dxk[i]; s1 += fac[O_K + j * 12 + i + 1] * dxk[i + 1]; }
                                    dv = -(s0 + s1);
                                }
                                o[j] = dv;                      // du_k (stage N has no input: 0)
                            } else {
                                dv = dxk[j - 6];
                            }
                            // all loads first, decisions on registers only (see corrector_bwd_pass)
                            const bool hc = has_comp(Nl, k, j);
                            const bool blo = hc && bnd_lo(P, j) > -BOUND_INF, bhi = hc && bnd_hi(P, j) < BOUND_INF;
                            double dtl, dll = 0.0, dtu, dlu = 0.0;
                            if (FAST) {
                                const double val = lt[j < 6 ? O_U + j : O_X + j - 6];   // the NLP iterate's u_j / q_{j-6}
                                const double dvel = dxk[j < 6 ? 6 + j : j];              // (the velocity step of the same joint: NaN check)
                                const bool good = ipm::fast_side(blo, bhi, dv, bnd_lo(P, j) - val, bnd_hi(P, j) - val, dtl, dtu);
                                al = good && dv == dv && dvel == dvel ? al : 0.0;
                            } else {
                            const double ll = lt[j], tl = lt[24 + j], lu = lt[12 + j], tu = lt[36 + j];
                            const double rdl = r[j], rdu = r[12 + j], rml = r[24 + j], rmu = r[36 + j];
                            ipm::lam_t_side(blo, dv, ll, tl, rdl, rml, al, a0, a1, a2, dtl, dll);
                            ipm::lam_t_side(bhi, -dv, lu, tu, rdu, rmu, al, a0, a1, a2, dtu, dlu);
                            }
                            o[30 + j] = dll; o[42 + j] = dlu;   // DLAM lower | upper
                            o[54 + j] = dtl; o[66 + j] = dtu;   // DT lower | upper
                        }
                        r_al.at(lane) = al; r_a0.at(lane) = a0; r_a1.at(lane) = a1; r_a2.at(lane) = a2;
                    });
                }
                if (ci == NCH - 1) {   // last window: publish the reductions over the whole horizon
                    ex.sub([&](int lane) {
                        ex.put1_min(sm.red[0], lane, r_al.at(lane)); ex.put1_sum(sm.red[1], lane, r_a0.at(lane));
                        ex.put1_sum(sm.red[2], lane, r_a1.at(lane)); ex.put1_sum(sm.red[3], lane, r_a2.at(lane));
                    });
                }
            }, [&](int lane, auto nl) {
                constexpr int NL = decltype(nl)::value;
                if (nk0 <= Nl) {
                    copy_lanes<LF, 0, W4, LF, true, NL>(n4, ex.smem().w.G4, nk0, nk1, lane);
                    if (FAST) copy_lanes<18, 0, W1, WLT, true, NL>(nlt, ex.smem().w.G1, nk0, nk1, lane);
                    else {
                    copy_lanes<48, O_QLAM, W1, WLT, true, NL>(nlt, ex.smem().w.G1, nk0, nk1, lane);
                    copy_lanes<48, O_RD, W3, WR, true, NL>(nr, ex.smem().w.G3, nk0, nk1, lane);
                    }
                }
                if (k0 > 0) {
                    if (AFFINE) copy_lanes<48, O_DLAM, W3, WO, false, NL>(po + 30, ex.smem().w.G3, k0 - CH, k0 - 1, lane);
                    else copy_lanes<WO, O_DW, W3, WO, false, NL>(po, ex.smem().w.G3, k0 - CH, k0 - 1, lane);
                }
                if (!AFFINE) {
                    // copies issued: the copy wavefronts now follow the recursion too and form
                    // dpi_{k-1} = p_k + P_k dx_k, block b on wavefront b mod (NL / 64)
                    const int wv = ex.uni(lane >> 6), l6 = lane & (WAVE - 1);
                    for (int kb = k0 + wv * BLK; kb <= k1; kb += BLK * (NL / WAVE)) {
                        const int ke = imin(kb + BLK - 1, k1), rows = ke - kb + 1;
                        ex.await(&sm.prog, ke);
                        if (l6 < rows * NX) {
                            const int s = l6 / NX, j = l6 - s * NX, k = kb + s;
                            const double *fac = v4 + (size_t)(k - k0) * LF;
                            double *o = vo + (size_t)(k - k0) * WO;
                            const double *dxk = o + 6;
                            double v = 0.0;
                            if (k >= 1) {
                                double s0 = fac[(AFFINE ? 0 : O_PV) + j], s1 = 0.0;
#pragma unroll
                                for (int i = 0; i < NX; i += 2) { s0 += fac[(AFFINE ? 0 : O_PM) + tri_sym(j, i)] * dxk[i]; s1 += fac[(AFFINE ? 0 : O_PM) + tri_sym(j, i + 1)] * dxk[i + 1]; }
                                v = s0 + s1;
                            }
                            o[18 + j] = v;                  // DPI slot of stage k holds dpi_{k-1}
                        }
                    }
                }
            });
            PROF_ADD(PF_SEQ_FWD, ts);
            if (k1 == Nl) {   // last chunk: nothing left to hide the store behind
                if (AFFINE) copy_rect<48, O_DLAM, W3, WO, false>(const_cast<double *>(vo + 30), ex.smem().w.G3, k0, k1);
                else store_rect<WO, O_DW, W3>(vo, ex.smem().w.G3, k0, k1);
            }
        }
        const double alpha = ex.get1(sm.red[0]);
        PROF_ADD(PF_FWD, t0);
        return alpha;
    }

    // =========================================================================== resident sweeps
    // Chunk-parallel affine recursion over the resident factor (see "LDS-resident factor" above).
    //   FWD:  dx_{k+1} = Acl_k dx_k + e_k,  dx_0 = 0                       -> vec[k] = dx_k, k = 0..N
    //   BWD:  p_k = c_k + Acl_k' p_{k+1},   p_N = c_N (vec holds c on entry) -> vec[k] = p_k, in place
    // with Acl_k = A - B K_k.  `xch`, `xs`: [RS_GROUPS][12] hand-over slots / chunk boundary values (scratch).
    // Ends with every lane's LDS writes issued, NOT with a barrier.
    // All indices are LOCAL to the range the maps describe (the whole horizon, or one segment): `nt` transitions, states 0..nt;
    // `x0`: FWD boundary value at state 0 (nullptr: zero).
    template <bool FWD>
    MPC_HD void rs_recursion(const ResMap &rm, double *vec, double *xch, double *xs, int nt, const double *x0 = nullptr)
    {
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(nt), L = rm.L;
        const int Jused = (Nl + L - 1) / L;
        typename Ex::template PerLane<double> z;
        typename Ex::template PerLane<D2> ab, bb;
        double b1r[6], b2r[6];
#pragma unroll
        for (int m = 0; m < 6; m++) { b1r[m] = P.b1[m]; b2r[m] = P.b2[m]; }
        // one step of chunk c = lane / 16 on transition k: reads the group's vector from its slot, leaves the new one in z
        auto step_compute = [&](int lane, int k) {
            const int c = lane >> 4, i = lane & 15, i6 = i < 6 ? i : i - 6;
            const double *zz = xch + c * 12;
            const D2 a = ab.at(lane), b = bb.at(lane);
            // the group's vector as six 16-byte reads; the lane's own entries by their own (lane-addressed) reads: a
            // register array indexed by a lane-dependent value becomes a chain of 2 x 11 v_cndmask per access
            const D2 *z2 = reinterpret_cast<const D2 *>(zz);
            double zv[12];
#pragma unroll
            for (int j = 0; j < NX; j += 2) { const D2 t = z2[j >> 1]; zv[j] = t.x; zv[j + 1] = t.y; }
            if (FWD) {
                const D2 *kr = reinterpret_cast<const D2 *>(rm.K + (size_t)k * 72 + i6 * 12);
                const double own = zz[i], ov = zz[i6 + 6], ek = rm.E[(size_t)k * 12 + i];
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int j = 0; j < NX; j += 2) { const D2 t = kr[j >> 1]; s0 += t.x * zv[j]; s1 += t.y * zv[j + 1]; }
                const double kd = s0 + s1;
                z.at(lane) = ek + (i < 6 ? own + a.x * ov - b.x * kd : a.y * own - b.y * kd);
            } else {
                const double *kc = rm.K + (size_t)k * 72 + i;   // column i of K_k
                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m += 2) {
                    acc0 += kc[m * 12] * (b1r[m] * zv[m] + b2r[m] * zv[6 + m]);
                    acc1 += kc[(m + 1) * 12] * (b1r[m + 1] * zv[m + 1] + b2r[m + 1] * zv[7 + m]);
                }
                const double mine = zz[i], oq = zz[i6];
                const double at = i < 6 ? mine : a.x * oq + a.y * mine;
                z.at(lane) = vec[(size_t)k * 12 + i] + (at - (acc0 + acc1));
            }
        };
        // local sweeps of all chunks; STORE: pass 3 (writes the results), else pass 1 (zero boundary value).
        // Per-lane chunk geometry (first / one-past-last transition, activity) is worked out once, not per step.
        typename Ex::template PerLane<int> g_ks, g_n;    // first transition of the lane's chunk, number of steps (0: lane idle)
        ex.wpar([&](int lane) {
            const int c = lane >> 4, i = lane & 15;
            const int ks = c * L, ke = imin(ks + L, Nl);
            g_ks.at(lane) = ks;
            g_n.at(lane) = (lane < rm.J * 16 && i < NX && ke > ks) ? ke - ks : 0;
        });
        auto sweep = [&](auto store_tag) {
            constexpr bool STORE = decltype(store_tag)::value;
            ex.wpar([&](int lane) {
                const int c = lane >> 4, i = lane & 15;
                if (lane < rm.J * 16 && i < NX) {
                    const double v = STORE ? xs[c * 12 + i] : 0.0;
                    z.at(lane) = v;
                    xch[c * 12 + i] = v;
                    if (STORE && FWD && c < Jused) vec[(size_t)c * L * 12 + i] = v;   // dx at the chunk's first stage
                }
            });
            for (int t = 0; t < L; t++) {
                ex.wpar([&](int lane) {
                    const int n = g_n.at(lane), ks = g_ks.at(lane);
                    if (t < n) step_compute(lane, FWD ? ks + t : ks + n - 1 - t);
                });
                ex.wpar([&](int lane) {
                    const int n = g_n.at(lane), ks = g_ks.at(lane);
                    if (t < n) {
                        const int c = lane >> 4, i = lane & 15, k = FWD ? ks + t : ks + n - 1 - t;
                        const double v = z.at(lane);
                        xch[c * 12 + i] = v;
                        if (STORE) {
                            // FWD: v = dx_{k+1}; the first stage of the next chunk belongs to that chunk (boundary value)
                            if (FWD) { if (t + 1 < n || k + 1 == Nl) vec[(size_t)(k + 1) * 12 + i] = v; }
                            else vec[(size_t)k * 12 + i] = v;
                        }
                    }
                });
            }
        };
        ex.wpar([&](int lane) {
            const int j6 = (lane & 15) % 6;
            D2 a; a.x = P.a12[j6]; a.y = P.a22[j6]; ab.at(lane) = a;
            D2 b; b.x = P.b1[j6]; b.y = P.b2[j6]; bb.at(lane) = b;
        });
        sweep(std::false_type{});
        ex.barrier();
        // pass 2: chunk boundary values, one group (wavefront 0, lanes < 12); y_c = the slot pass 1 left
        {
            typename Ex::template PerLane<double> xr;
            int cur = 0;
            ex.seq([&](int lane) {
                const double v = lane < NX ? (FWD ? (x0 ? x0[lane] : 0.0) : vec[(size_t)Nl * 12 + lane]) : 0.0;
                xr.at(lane) = v;
                if (lane < NX) ex.share(sm.pv[cur], lane, v);
            });
            for (int cc = 0; cc < Jused; cc++) {
                const int c = FWD ? cc : Jused - 1 - cc, nxt = cur ^ 1;
                ex.seq([&](int lane) {
                    // this chunk's row (column) of Phi and y first: their latency runs under the gather
                    const int lc = lane < NX ? lane : 0;
                    const double *ph = rm.PHI + (size_t)c * 144;
                    double pr[12];
#pragma unroll
                    for (int j = 0; j < NX; j++) pr[j] = FWD ? ph[lc * 12 + j] : ph[j * 12 + lc];
                    const double y = xch[c * 12 + lc];
                    const double own = xr.at(lane);
                    double xv[12];
#pragma unroll
                    for (int j = 0; j < NX; j++) xv[j] = ex.gather(sm.pv[cur], j, own);
                    double s0 = y, s1 = 0.0;
#pragma unroll
                    for (int j = 0; j < NX; j += 2) { s0 += pr[j] * xv[j]; s1 += pr[j + 1] * xv[j + 1]; }
                    const double v = s0 + s1;
                    xr.at(lane) = v;
                    if (lane < NX) { xs[c * 12 + lane] = own; ex.share(sm.pv[nxt], lane, v); }
                });
                cur = nxt;
            }
        }
        ex.barrier();
        sweep(std::true_type{});
    }

    // The same recursion with the factor in REGISTERS instead of LDS (chunks of exactly RL = 7 transitions): every lane of a group
    // loads, once per sweep and straight from the HBM factor record, what it needs of K for ALL steps of its chunk -- forward: HALF a
    // row of K (lanes i and i + 6 share row i and swap their partial dot products by a DPP row shift), 6 doubles per step, plus e_k;
    // backward: column i, 6 doubles per step -- 49 doubles per lane.  The sweeps then need no LDS for the factor at all, so they run
    // with half a pool (two simulations per CU) and at any horizon (segment by segment).  `kb`: first stage of the segment (HBM index).
    static constexpr int RL = 7;
    template <int T0, class F>
    MPC_HD static void unroll_rl(F &&f)
    {
        if constexpr (T0 < RL) { f(std::integral_constant<int, T0>{}); unroll_rl<T0 + 1>(f); }
    }
    template <bool FWD>
    MPC_HD void rs_recursion_reg(const ResMap &rm, double *vec, double *xch, double *xs, double *psl, int nt, int kb, const double *x0 = nullptr)
    {
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(nt);
        constexpr int L = RL;
        const int Jused = (Nl + L - 1) / L;
        const double *const G4 = ex.smem().w.G4;
        typename Ex::template PerLane<double> z, pp;
        typename Ex::template PerLane<D2> ab, bb;
        typename Ex::template PerLane<double> kreg[RL][6], ereg[RL];
        typename Ex::template PerLane<int> g_ks, g_n;
        double b1r[6], b2r[6];
#pragma unroll
        for (int m = 0; m < 6; m++) { b1r[m] = P.b1[m]; b2r[m] = P.b2[m]; }
        // chunk geometry, model constants, and this lane's share of K (and e) for every step of its chunk
        ex.wpar([&](int lane) {
            const int c = lane >> 4, i = lane & 15, i6 = i < 6 ? i : i - 6, j6 = i % 6;
            const int ks = c * L, ke = imin(ks + L, Nl);
            const int n = (lane < rm.J * 16 && i < NX && ke > ks) ? ke - ks : 0;
            g_ks.at(lane) = ks; g_n.at(lane) = n;
            D2 a; a.x = P.a12[j6]; a.y = P.a22[j6]; ab.at(lane) = a;
            D2 b; b.x = P.b1[j6]; b.y = P.b2[j6]; bb.at(lane) = b;
            unroll_rl<0>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                const int k = kb + (FWD ? ks + t : ks + n - 1 - t);
                if (t < n) {
                    const double *g4 = G4 + (size_t)k * W4;
                    if (FWD) {
                        const double *kr = g4 + O_K + i6 * 12 + (i < 6 ? 0 : 6);
#pragma unroll
                        for (int j = 0; j < 6; j++) kreg[t][j].at(lane) = gld(kr + j);
                        ereg[t].at(lane) = gld(g4 + O_E + i);
                    } else {
#pragma unroll
                        for (int m = 0; m < 6; m++) kreg[t][m].at(lane) = gld(g4 + O_K + m * 12 + i);
                    }
                }
            });
        });
        auto sweep = [&](auto store_tag) {
            constexpr bool STORE = decltype(store_tag)::value;
            ex.wpar([&](int lane) {
                const int c = lane >> 4, i = lane & 15;
                if (lane < rm.J * 16 && i < NX) {
                    const double v = STORE ? xs[c * 12 + i] : 0.0;
                    z.at(lane) = v;
                    xch[c * 12 + i] = v;
                    if (STORE && FWD && c < Jused) vec[(size_t)c * L * 12 + i] = v;   // dx at the chunk's first stage
                }
            });
            unroll_rl<0>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if (t < L && ex.uni(t < Nl)) {
                    ex.wpar([&](int lane) {
                        const int n = g_n.at(lane), ks = g_ks.at(lane);
                        if (t < n) {
                            const int c = lane >> 4, i = lane & 15, i6 = i < 6 ? i : i - 6;
                            const double *zz = xch + c * 12;
                            if (FWD) {
                                // partial dot product of row i6 of K with this lane's half of the vector
                                const D2 *z2 = reinterpret_cast<const D2 *>(zz + (i < 6 ? 0 : 6));
                                const D2 v0 = z2[0], v1 = z2[1], v2 = z2[2];
                                const double s0 = kreg[t][0].at(lane) * v0.x + kreg[t][2].at(lane) * v1.x + kreg[t][4].at(lane) * v2.x;
                                const double s1 = kreg[t][1].at(lane) * v0.y + kreg[t][3].at(lane) * v1.y + kreg[t][5].at(lane) * v2.y;
                                const double ph = s0 + s1;
                                pp.at(lane) = ph;
                                ex.share(psl + c * 12, i, ph);
                            } else {
                                const int k = ks + n - 1 - t;
                                const D2 *z2 = reinterpret_cast<const D2 *>(zz);
                                double zv[12];
#pragma unroll
                                for (int j = 0; j < NX; j += 2) { const D2 tt = z2[j >> 1]; zv[j] = tt.x; zv[j + 1] = tt.y; }
                                double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                                for (int m = 0; m < 6; m += 2) {
                                    acc0 += kreg[t][m].at(lane) * (b1r[m] * zv[m] + b2r[m] * zv[6 + m]);
                                    acc1 += kreg[t][m + 1].at(lane) * (b1r[m + 1] * zv[m + 1] + b2r[m + 1] * zv[7 + m]);
                                }
                                const D2 a = ab.at(lane);
                                const double mine = zz[i], oq = zz[i6];
                                const double at = i < 6 ? mine : a.x * oq + a.y * mine;
                                z.at(lane) = vec[(size_t)k * 12 + i] + (at - (acc0 + acc1));
                            }
                        }
                    });
                    if (FWD) {
                        ex.wpar([&](int lane) {
                            const int n = g_n.at(lane);
                            // (the DPP shift is executed by every lane of the wavefront: no lane-dependent branch around it)
                            const int c = lane >> 4, i = lane & 15;
                            const double mine = pp.at(lane);
                            const double up = ex.shl6(psl + c * 12, i, mine), dn = ex.shr6(psl + c * 12, i, mine);
                            if (t < n) {
                                const double *zz = xch + c * 12;
                                const int i6 = i < 6 ? i : i - 6;
                                const double kd = i < 6 ? mine + up : dn + mine;        // lower half + upper half, in both lanes
                                const D2 a = ab.at(lane), b = bb.at(lane);
                                const double own = zz[i], ov = zz[i6 + 6];
                                z.at(lane) = ereg[t].at(lane) + (i < 6 ? own + a.x * ov - b.x * kd : a.y * own - b.y * kd);
                            }
                        });
                    }
                    ex.wpar([&](int lane) {
                        const int n = g_n.at(lane), ks = g_ks.at(lane);
                        if (t < n) {
                            const int c = lane >> 4, i = lane & 15, k = FWD ? ks + t : ks + n - 1 - t;
                            const double v = z.at(lane);
                            xch[c * 12 + i] = v;
                            if (STORE) {
                                if (FWD) { if (t + 1 < n || k + 1 == Nl) vec[(size_t)(k + 1) * 12 + i] = v; }
                                else vec[(size_t)k * 12 + i] = v;
                            }
                        }
                    });
                }
            });
        };
        sweep(std::false_type{});
        ex.barrier();
        // pass 2: chunk boundary values, one group (wavefront 0, lanes < 12); y_c = the slot pass 1 left
        {
            typename Ex::template PerLane<double> xr;
            int cur = 0;
            ex.seq([&](int lane) {
                const double v = lane < NX ? (FWD ? (x0 ? x0[lane] : 0.0) : vec[(size_t)Nl * 12 + lane]) : 0.0;
                xr.at(lane) = v;
                if (lane < NX) ex.share(sm.pv[cur], lane, v);
            });
            for (int cc = 0; cc < Jused; cc++) {
                const int c = FWD ? cc : Jused - 1 - cc, nxt = cur ^ 1;
                ex.seq([&](int lane) {
                    const int lc = lane < NX ? lane : 0;
                    const double *ph = rm.PHI + (size_t)c * 144;
                    double pr[12];
#pragma unroll
                    for (int j = 0; j < NX; j++) pr[j] = FWD ? ph[lc * 12 + j] : ph[j * 12 + lc];
                    const double y = xch[c * 12 + lc];
                    const double own = xr.at(lane);
                    double xv[12];
#pragma unroll
                    for (int j = 0; j < NX; j++) xv[j] = ex.gather(sm.pv[cur], j, own);
                    double s0 = y, s1 = 0.0;
#pragma unroll
                    for (int j = 0; j < NX; j += 2) { s0 += pr[j] * xv[j]; s1 += pr[j + 1] * xv[j + 1]; }
                    const double v = s0 + s1;
                    xr.at(lane) = v;
                    if (lane < NX) { xs[c * 12 + lane] = own; ex.share(sm.pv[nxt], lane, v); }
                });
                cur = nxt;
            }
        }
        ex.barrier();
        sweep(std::true_type{});
    }

    // Forward sweep on the resident factor: dx by rs_recursion, then item-parallel (lane <-> (stage, bounded component)):
    // du = -(R~^-1 h_u + K dx), dt, dlam (HPIPM compute_lam_t), largest feasible step, centering sums; the final sweep
    // also dpi_{k-1} = p_k + P_k dx_k and the whole Newton step to HBM.  Leaves alpha, S0, S1, S2 in sm.cen[0..3].
    // SEG: the factor of ONE SEGMENT of SEG_T transitions at a time (loaded from the HBM record the plain factorisation wrote),
    // segments in ascending order, dx handed from segment to segment; all LDS indices are local to the segment (kb = its first stage).
    // KREG (with SEG): nothing of the factor in LDS -- the recursion holds its share of K in registers (rs_recursion_reg), the items
    // load their row of K and R~^-1 h_u from the HBM record themselves, after the recursion (one batch of one item per lane at a
    // time: the register budget of two simulations per CU is 256).
    // FAST (bound-inactive fast path, mpc_ipm.h): the Newton system is the equality-constrained QP's at w = 0, so the step IS the
    // candidate solution; instead of dlam / dt / step length the items form the candidate's slacks (-> G3 DT) and the pass returns
    // 1.0 when every bounded component clears its bounds by ipm::FAST_MARGIN (and nothing is NaN), else 0.0.
    template <bool AFFINE, bool SEG, bool KREG = false, bool FAST = false>
    MPC_PASS double fwd_resident()
    {
        static_assert(!KREG || SEG, "register mode runs segment by segment");
        static_assert(!FAST || !AFFINE, "the fast path takes the full step");
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(ex.smem().n_hor);
        const ResMap rm = KREG ? reg_map() : (SEG ? seg_map() : res_map());
        const int NSL = rm.T + 1;       // states the maps hold
        double *X = rm.scr, *xch = X + (size_t)NSL * 12, *xs = xch + RS_GROUPS * 12, *xin = xs + RS_GROUPS * 12, *psl = xin + 16;
        constexpr int R = KREG ? 1 : rounds_for(6);
        constexpr int NLD = KREG ? 29 : 16;    // KREG: + row j of K (12) and R~^-1 h_u [j]
        double *const G1 = ex.smem().w.G1, *const G3 = ex.smem().w.G3, *const G4 = ex.smem().w.G4;
        typename Ex::template PerLane<double> ld[R][NLD];
        typename Ex::template PerLane<double> r_al, r_a0, r_a1, r_a2;
        constexpr int RP = RS_ROUNDS;
        ex.wpar([&](int lane) { r_al.at(lane) = 1.0; r_a0.at(lane) = 0.0; r_a1.at(lane) = 0.0; r_a2.at(lane) = 0.0; });
        for (int kb = 0; kb == 0 || kb < Nl; kb += SEG ? rm.T : Nl + 1) {
            const int nt = ex.uni(SEG ? imin(rm.T, Nl - kb) : Nl);          // transitions of this segment
            const int nsi = nt + (kb + nt == Nl ? 1 : 0);                    // stages with items: the end state only in the last segment
            const int items = nsi * 6;        // joint items (k, j): the bounded components u_j and q_j together
            const int items_pi = nsi * NB;    // dpi items (k, state component)
            // operands of a batch of items: lam, t (G1), rd, rm (G3); lower | upper of u_j, then of q_j -- unconditional, clamped
            auto issue = [&](int base) {
                ex.wpar([&](int lane) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), kl = e / 6, j = e - kl * 6, k = kb + kl;
                        const double *g1 = G1 + (size_t)k * W1, *g3 = G3 + (size_t)k * W3;
                        if (FAST) {
                            ld[r][0].at(lane) = gld(g1 + O_U + j); ld[r][8].at(lane) = gld(g1 + O_X + j);   // the NLP iterate's u_j, q_j
                        } else {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int c = j + 6 * h;
                            ld[r][8 * h + 0].at(lane) = gld(g1 + O_QLAM + c); ld[r][8 * h + 1].at(lane) = gld(g1 + O_QLAM + 12 + c);
                            ld[r][8 * h + 2].at(lane) = gld(g1 + O_QT + c);   ld[r][8 * h + 3].at(lane) = gld(g1 + O_QT + 12 + c);
                            ld[r][8 * h + 4].at(lane) = gld(g3 + O_RD + c);   ld[r][8 * h + 5].at(lane) = gld(g3 + O_RD + 12 + c);
                            ld[r][8 * h + 6].at(lane) = gld(g3 + O_RM + c);   ld[r][8 * h + 7].at(lane) = gld(g3 + O_RM + 12 + c);
                        }
                        }
                        if (KREG) {
                            const double *g4 = G4 + (size_t)(kb + imin(kl, nt - 1)) * W4;
#pragma unroll
                            for (int i = 0; i < NX; i++) ld[r][16 + i].at(lane) = gld(g4 + O_K + j * 12 + i);
                            ld[r][28].at(lane) = gld(g4 + O_VH + j);
                        }
                    }
                });
            };
            if (SEG) {
                // this segment's factor and chunk transition matrices: HBM -> the resident arrays (coalesced bursts)
                const int c0 = kb / rm.L, nc = (nt + rm.L - 1) / rm.L;
                copies([&](int lane, auto nl) {
                    constexpr int NL = decltype(nl)::value;
                    if (!KREG) {
                        copy_lanes<72, O_K, W4, 72, true, NL>(rm.K, G4, kb, kb + nt - 1, lane);
                        copy_lanes<6, O_VH, W4, 6, true, NL>(rm.VH, G4, kb, kb + nt - 1, lane);
                        copy_lanes<12, O_E, W4, 12, true, NL>(rm.E, G4, kb, kb + nt - 1, lane);
                    }
                    if (!AFFINE) copy_lanes<12, O_PV, W4, 12, true, NL>(rm.P, G4, kb, kb + nsi - 1, lane);
                    copy_lanes<144, 0, 144, 144, true, NL>(rm.PHI, ex.smem().w.PH, c0, c0 + nc - 1, lane);
                });
            }
            if (!KREG) issue(0);
            PROF_T0(ts);
            if (KREG) rs_recursion_reg<true>(rm, X, xch, xs, psl, nt, kb, kb > 0 ? xin : nullptr);
            else rs_recursion<true>(rm, X, xch, xs, nt, kb > 0 ? xin : nullptr);
            ex.barrier();
            PROF_ADD(PF_SEQ_FWD, ts);
            for (int base = 0; base < items; base += R * NT) {
                if (KREG || base > 0) issue(base);
                ex.wpar([&](int lane) {
                    double al = r_al.at(lane), a0 = r_a0.at(lane), a1 = r_a1.at(lane), a2 = r_a2.at(lane);
                    auto side = [&](bool on, double sdv, double l, double t, double rd, double rmv, double &dt_o, double &dl_o) {
                        ipm::lam_t_side(on, sdv, l, t, rd, rmv, al, a0, a1, a2, dt_o, dl_o);   // (mpc_ipm.h: compute_lam_t, step length, centering sums)
                    };
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int kl = e / 6, j = e - kl * 6, k = kb + kl;
                            const double *dxk = X + (size_t)kl * 12;
                            double *g3 = G3 + (size_t)k * W3;
                            // du_k[j] = -(R~^-1 h_u + K dx_k)[j]  (stage N has no input: 0)
                            const int kc = imin(kl, nt - 1);
                            const D2 *x2 = reinterpret_cast<const D2 *>(dxk);
                            double s0, s1 = 0.0;
                            if (KREG) {
                                s0 = ld[r][NLD - 1].at(lane);
#pragma unroll
                                for (int i = 0; i < NX; i += 2) { const D2 xv = x2[i >> 1]; s0 += ld[r][NLD - 13 + i].at(lane) * xv.x; s1 += ld[r][NLD - 12 + i].at(lane) * xv.y; }
                            } else {
                                const D2 *kr = reinterpret_cast<const D2 *>(rm.K + (size_t)kc * 72 + j * 12);
                                s0 = rm.VH[(size_t)kc * 6 + j];
#pragma unroll
                                for (int i = 0; i < NX; i += 2) { const D2 kv = kr[i >> 1], xv = x2[i >> 1]; s0 += kv.x * xv.x; s1 += kv.y * xv.y; }
                            }
                            const double du = k < Nl ? -(s0 + s1) : 0.0, dq = dxk[j];
                            if (!AFFINE) {
                                gst(g3 + O_DW + j, du); gst(g3 + O_DW + 6 + j, dq); gst(g3 + O_DW + 12 + j, dxk[6 + j]);   // du_k, dx_k
                            }
                            const bool oku = k < Nl, okq = k >= 1 && k < Nl;
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                const int c = j + 6 * h;
                                const bool ok = h == 0 ? oku : okq;
                                const bool blo = ok && sm.bon[c] != 0.0, bhi = ok && sm.bon[12 + c] != 0.0;
                                const double dv = h == 0 ? du : dq;
                                double dtl, dll, dtu, dlu;
                                if (FAST) {
                                    const double val = ld[r][8 * h].at(lane);
                                    const bool good = ipm::fast_side(blo, bhi, dv, bnd_lo(P, c) - val, bnd_hi(P, c) - val, dtl, dtu);
                                    al = good && dv == dv && dxk[6 + j] == dxk[6 + j] ? al : 0.0;
                                } else {
                                side(blo, dv, ld[r][8 * h + 0].at(lane), ld[r][8 * h + 2].at(lane), ld[r][8 * h + 4].at(lane), ld[r][8 * h + 6].at(lane), dtl, dll);
                                side(bhi, -dv, ld[r][8 * h + 1].at(lane), ld[r][8 * h + 3].at(lane), ld[r][8 * h + 5].at(lane), ld[r][8 * h + 7].at(lane), dtu, dlu);
                                gst(g3 + O_DLAM + c, dll); gst(g3 + O_DLAM + 12 + c, dlu);
                                }
                                gst(g3 + O_DT + c, dtl);   gst(g3 + O_DT + 12 + c, dtu);
                            }
                        }
                    }
                    r_al.at(lane) = al; r_a0.at(lane) = a0; r_a1.at(lane) = a1; r_a2.at(lane) = a2;
                });
            }
            if (!AFFINE) {
                for (int base = 0; base < items_pi; base += RP * NT) {
                    // dpi_{k-1} = p_k + P_k dx_k (the DPI slot of stage k holds dpi_{k-1}): row j of the packed P_k straight from HBM
                    // (loaded here, not with the other operands: held across the recursion they cost the 8-wavefront build its registers)
                    ex.wpar([&](int lane) {
                        double pm[RP][12];
#pragma unroll
                        for (int r = 0; r < RP; r++) {
                            const int e = imin(base + r * NT + lane, items_pi - 1), kl = e / NB, j = e - kl * NB;
                            const double *g4 = G4 + (size_t)(kb + kl) * W4 + O_PM;
#pragma unroll
                            for (int i = 0; i < NX; i++) pm[r][i] = gld(g4 + tri_sym(j, i));
                        }
#pragma unroll
                        for (int r = 0; r < RP; r++) {
                            const int e = base + r * NT + lane;
                            if (e < items_pi) {
                                const int kl = e / NB, j = e - kl * NB, k = kb + kl;
                                const double *dxk = X + (size_t)kl * 12;
                                double v = 0.0;
                                if (k >= 1) {
                                    double s0 = rm.P[(size_t)kl * 12 + j], s1 = 0.0;
#pragma unroll
                                    for (int i = 0; i < NX; i += 2) { s0 += pm[r][i] * dxk[i]; s1 += pm[r][i + 1] * dxk[i + 1]; }
                                    v = s0 + s1;
                                }
                                gst(G3 + (size_t)k * W3 + O_DPI + j, v);
                            }
                        }
                    });
                }
            }
            if (SEG) {
                // dx at the segment's end state is the next segment's boundary value (the arrays are about to be reloaded)
                ex.wpar([&](int lane) { if (lane < NX) xin[lane] = X[(size_t)nt * 12 + lane]; });
                ex.barrier();
            }
        }
        ex.par([&](int lane) {
            ex.put_min(sm.red[0], lane, r_al.at(lane)); ex.put_sum(sm.red[1], lane, r_a0.at(lane));
            ex.put_sum(sm.red[2], lane, r_a1.at(lane)); ex.put_sum(sm.red[3], lane, r_a2.at(lane));
        });
        const double alpha = ex.get_min(sm.red[0]);
        const double S0 = ex.get_sum(sm.red[1]), S1 = ex.get_sum(sm.red[2]), S2 = ex.get_sum(sm.red[3]);
        ex.par([&](int lane) {
            if ((lane & (WAVE - 1)) == 0) { sm.cen[0] = alpha; sm.cen[1] = S0; sm.cen[2] = S1; sm.cen[3] = S2; }
        });
        PROF_ADD(PF_FWD, t0);
        return alpha;
    }

    // Centering corrector + backward solve on the resident factor (see corrector_bwd_pass for the algebra):
    //   items (stage, component): rm, rebuilt gt -> LDS ; c_k -> resident p array ; rs_recursion (p in place) ;
    //   items: h_u, R~^-1 h_u, e -> resident.  Only rm goes back to HBM.
    // SEG: segment by segment from the top of the horizon, p handed down; K comes from HBM, R~^-1 h_u | e | p go back there.
    // KREG: K columns for c_k and for the recursion come from the HBM record (items / registers); R~^-1 h_u and e go straight back to it.
    template <bool SEG, bool KREG = false>
    MPC_PASS void corr_resident(double sigma_mu)
    {
        static_assert(!KREG || SEG, "register mode runs segment by segment");
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(ex.smem().n_hor);
        const ResMap rm = KREG ? reg_map() : (SEG ? seg_map() : res_map());
        const int NSL = rm.T + 1;
        double *GT = rm.scr, *RW = GT + (size_t)NSL * 18, *xch = RW + (size_t)NSL * 12, *xs = xch + RS_GROUPS * 12, *pin = xs + RS_GROUPS * 12,
               *psl = pin + 16;
        constexpr int R = RS_ROUNDS;
        double *const G1 = ex.smem().w.G1, *const G2 = ex.smem().w.G2, *const G3 = ex.smem().w.G3, *const G4 = ex.smem().w.G4;
        typename Ex::template PerLane<double> ld[R][7];
        const int nseg = SEG ? (Nl + rm.T - 1) / rm.T : 1;
        for (int sg = nseg - 1; sg >= 0; sg--) {
            const int kb = SEG ? sg * rm.T : 0;
            const int nt = ex.uni(SEG ? imin(rm.T, Nl - kb) : Nl);
            const bool top = kb + nt == Nl;
            const int nsi = nt + (top ? 1 : 0);
            const int items = nsi * NB;
            if (SEG) {
                const int c0 = kb / rm.L, nc = (nt + rm.L - 1) / rm.L;
                copies([&](int lane, auto nl) {
                    constexpr int NL = decltype(nl)::value;
                    if (!KREG) copy_lanes<72, O_K, W4, 72, true, NL>(rm.K, G4, kb, kb + nt - 1, lane);
                    copy_lanes<144, 0, 144, 144, true, NL>(rm.PHI, ex.smem().w.PH, c0, c0 + nc - 1, lane);
                });
            }
            // ---- rm, gt of every bounded component (+ w_k into LDS); gt of the velocity components is rg
            for (int base = 0; base < items; base += R * NT) {
                ex.wpar([&](int lane) {
                    double v[R][12];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), kl = e / NB, j = e - kl * NB, k = kb + kl;
                        const double *g1 = G1 + (size_t)k * W1, *g3 = G3 + (size_t)k * W3;
                        v[r][0] = gld(g1 + O_QLAM + j); v[r][1] = gld(g1 + O_QLAM + 12 + j);
                        v[r][2] = gld(g1 + O_QT + j);   v[r][3] = gld(g1 + O_QT + 12 + j);
                        v[r][4] = gld(g3 + O_DLAM + j); v[r][5] = gld(g3 + O_DLAM + 12 + j);
                        v[r][6] = gld(g3 + O_DT + j);   v[r][7] = gld(g3 + O_DT + 12 + j);
                        v[r][8] = gld(g3 + O_RD + j);   v[r][9] = gld(g3 + O_RD + 12 + j);
                        v[r][10] = gld(g3 + O_RG + j);
                        v[r][11] = gld(G4 + (size_t)k * W4 + O_WV + j);
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int kl = e / NB, j = e - kl * NB, k = kb + kl;
                            const bool hc = j < 6 ? k < Nl : (k >= 1 && k < Nl);
                            const bool blo = hc && sm.bon[j] != 0.0, bhi = hc && sm.bon[12 + j] != 0.0;
                            const double ll = v[r][0], lu = v[r][1], tl = v[r][2], tu = v[r][3];
                            const double dll = v[r][4], dlu = v[r][5], dtl = v[r][6], dtu = v[r][7], rdl = v[r][8], rdu = v[r][9];
                            double gt = v[r][10];
                            double rml, rmu;
                            gt += ipm::corrector_side(blo, ll, tl, dll, dtl, rdl, sigma_mu, rml);
                            gt -= ipm::corrector_side(bhi, lu, tu, dlu, dtu, rdu, sigma_mu, rmu);
                            GT[(size_t)kl * 18 + j] = gt;
                            RW[(size_t)kl * 12 + j] = v[r][11];
                            gst(G3 + (size_t)k * W3 + O_RM + j, rml); gst(G3 + (size_t)k * W3 + O_RM + 12 + j, rmu);
                        }
                    }
                });
            }
            ex.wpar([&](int lane) {
                for (int e = lane; e < nsi * 6; e += NT) {
                    const int kl = e / 6, j = e - kl * 6;
                    GT[(size_t)kl * 18 + 12 + j] = gld(G3 + (size_t)(kb + kl) * W3 + O_RG + 12 + j);
                }
            });
            ex.barrier();
            // ---- c_k = gt_x + A' w - Kfb' (gt_u + B' w) -> resident p array (stage N: p_N = gt_x; a lower segment's end state: p
            //      handed down from the segment above)
            for (int base = 0; base < items; base += R * NT) {
                ex.wpar([&](int lane) {
                    double kc[R][6];     // column j of K_k: LDS (resident / segment) or, KREG, the HBM record
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), kl = e / NX, j = e - kl * NX, kcl = imin(kl, nt - 1);
#pragma unroll
                        for (int m = 0; m < 6; m++)
                            kc[r][m] = KREG ? gld(G4 + (size_t)(kb + kcl) * W4 + O_K + m * 12 + j) : rm.K[(size_t)kcl * 72 + m * 12 + j];
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int kl = e / NX, j = e - kl * NX, k = kb + kl;
                            const double *gt = GT + (size_t)kl * 18, *w = RW + (size_t)kl * 12;
                            double vv = gt[6 + j];
                            if (k < Nl) {
                                vv += (j < 6 ? w[j] : P.a12[j - 6] * w[j - 6] + P.a22[j - 6] * w[j]);
#pragma unroll
                                for (int m = 0; m < 6; m++) vv -= kc[r][m] * (gt[m] + P.b1[m] * w[m] + P.b2[m] * w[6 + m]);
                            }
                            rm.P[(size_t)kl * 12 + j] = vv;
                        }
                    }
                    if (SEG && !top && base == 0 && lane < NX) rm.P[(size_t)nt * 12 + lane] = pin[lane];
                });
            }
            ex.barrier();
            // operands of the last phase (R~^-1 row, rb), issued before the recursion
            auto issue = [&](int base) {
                ex.wpar([&](int lane) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), kl = e / NX, j = e - kl * NX, i6 = j < 6 ? j : j - 6, k = kb + kl;
                        const double *ri = G4 + (size_t)k * W4 + O_RI + i6 * 6;
#pragma unroll
                        for (int m = 0; m < 6; m++) ld[r][m].at(lane) = gld(ri + m);
                        ld[r][6].at(lane) = gld(G2 + (size_t)k * W2 + O_RB + j);
                    }
                });
            };
            if (!KREG) issue(0);
            PROF_T0(ts);
            if (KREG) rs_recursion_reg<false>(rm, rm.P, xch, xs, psl, nt, kb);
            else rs_recursion<false>(rm, rm.P, xch, xs, nt);
            ex.barrier();
            PROF_ADD(PF_SEQ_BWD, ts);
            // ---- h_u,k = gt_u + B'(p_{k+1} + w_k) ; R~^-1 h_u and e = rb - B R~^-1 h_u for the forward sweep
            for (int base = 0; base < items; base += R * NT) {
                if (KREG || base > 0) issue(base);
                ex.wpar([&](int lane) {
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int kl = e / NX, j = e - kl * NX, i6 = j < 6 ? j : j - 6, k = kb + kl;
                            double vh = 0.0, ee = 0.0;
                            if (k < Nl) {
                                const double *gt = GT + (size_t)kl * 18, *w = RW + (size_t)kl * 12, *pn = rm.P + (size_t)(kl + 1) * 12;
                                double v0 = 0.0, v1 = 0.0;
#pragma unroll
                                for (int m = 0; m < 6; m += 2) {
                                    const double h0 = gt[m] + P.b1[m] * (pn[m] + w[m]) + P.b2[m] * (pn[6 + m] + w[6 + m]);
                                    const double h1 = gt[m + 1] + P.b1[m + 1] * (pn[m + 1] + w[m + 1]) + P.b2[m + 1] * (pn[7 + m] + w[7 + m]);
                                    v0 += ld[r][m].at(lane) * h0; v1 += ld[r][m + 1].at(lane) * h1;
                                }
                                vh = v0 + v1;
                                ee = ld[r][6].at(lane) - (j < 6 ? P.b1[i6] : P.b2[i6]) * vh;
                            }
                            if (KREG) {
                                if (j < 6) gst(G4 + (size_t)k * W4 + O_VH + j, vh);
                                gst(G4 + (size_t)k * W4 + O_E + j, ee);
                            } else {
                                if (j < 6) rm.VH[(size_t)kl * 6 + j] = vh;
                                rm.E[(size_t)kl * 12 + j] = ee;
                            }
                        }
                    }
                });
            }
            ex.barrier();
            if (SEG) {
                // what the final forward sweep reads of this segment goes back to the HBM record; p at its first stage goes down
                copies([&](int lane, auto nl) {
                    constexpr int NL = decltype(nl)::value;
                    if (!KREG) {
                        copy_lanes<6, O_VH, W4, 6, false, NL>(rm.VH, G4, kb, kb + nsi - 1, lane);
                        copy_lanes<12, O_E, W4, 12, false, NL>(rm.E, G4, kb, kb + nsi - 1, lane);
                    }
                    copy_lanes<12, O_PV, W4, 12, false, NL>(rm.P, G4, kb, kb + nsi - 1, lane);
                    if (lane < NX) pin[lane] = rm.P[lane];
                });
            }
        }
        PROF_ADD(PF_BWD, t0);
    }

    // =========================================================================== bound-inactive fast path (mpc_ipm.h)
    // Right-hand side of the equality-constrained QP's Newton system at w = 0 with x_0 embedded (dx0 = x_hat - x_0): no bound terms,
    //   Gamma = 0 ; gt = g (stage 0: + H_0 [0; dx0], x rows eliminated) ; rb = b (stage 0: + A dx0)
    // from the linearisation the NLP pass left in G2 (r -> y = W r, G, the dynamics defect).  Joint items (k, j < 6), operands straight
    // from HBM like the S phase of residual_direct, whose formulas these are at (dw, pi, lam) = 0.
    MPC_PASS void fast_rhs()
    {
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(ex.smem().n_hor), NS = Nl + 1;
        double *const G1 = ex.smem().w.G1, *const G2 = ex.smem().w.G2;
        constexpr int R = rounds_for(6);
        const int items = NS * 6;
        for (int base = 0; base < items; base += R * NT) {
            ex.wpar([&](int lane) {
                double v[R][16];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int e = imin(base + r * NT + lane, items - 1), k = e / 6, j = e - k * 6;
                    const double *g1 = G1 + (size_t)k * W1, *g2 = G2 + (size_t)k * W2;
                    v[r][0] = gld(g1 + O_U + j); v[r][1] = gld(g1 + O_X + 6 + j); v[r][2] = gld(g1 + O_X + j);
                    v[r][3] = gld(g2 + O_GV + j);
#pragma unroll
                    for (int i = 0; i < NTASK; i++) { v[r][4 + i] = gld(g2 + O_GQ + i * 6 + j); v[r][9 + i] = gld(g2 + O_Y + i); }
                    v[r][14] = gld(g2 + O_BD + j); v[r][15] = gld(g2 + O_BD + 6 + j);
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int e = base + r * NT + lane;
                    if (e < items) {
                        const int k = e / 6, j = e - k * 6;
                        const double dxq = k == 0 ? sm.xhat[j] - v[r][2] : 0.0, dxv = k == 0 ? sm.xhat[6 + j] - v[r][1] : 0.0;
                        const RhsItem o = rhs_item(P, j, k < Nl, k >= 1, v[r][0], v[r][1], dxq, dxv, v[r][4], v[r][5], v[r][6], v[r][7], v[r][8],
                                                   v[r][9], v[r][10], v[r][11], v[r][12], v[r][13], v[r][3], v[r][14], v[r][15]);
                        double *g2 = G2 + (size_t)k * W2;
                        gst(g2 + O_GT + j, o.gtu); gst(g2 + O_GT + 6 + j, o.gtq); gst(g2 + O_GT + 12 + j, o.gtv);
                        gst(g2 + O_GAM + j, 0.0); gst(g2 + O_GAM + 6 + j, 0.0);
                        gst(g2 + O_RB + j, o.rbq); gst(g2 + O_RB + 6 + j, o.rbv);
                    }
                }
            });
        }
        ex.barrier();
        PROF_ADD(PF_RES, t0);
    }

    // The accepted candidate becomes the QP iterate: (QW | QPI) <- (DW | DPI), x_0 embedded; QLAM <- 0; QT <- the slacks the forward
    // sweep left in DT (1 on absent sides).  Stage 0's y = W (r + G [dx0]) is refreshed for the SQP merit weights (update_x0_weights).
    MPC_PASS void fast_commit()
    {
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const int Nl = ex.uni(ex.smem().n_hor), NS = Nl + 1;
        double *const G1 = ex.smem().w.G1, *const G2 = ex.smem().w.G2, *const G3 = ex.smem().w.G3;
        {
            constexpr int IPS = 15, R = rounds_for(IPS);
            const int items = NS * IPS;
            for (int base = 0; base < items; base += R * NT) {
                ex.wpar([&](int lane) {
                    D2 stp[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), k = e / IPS, c = 2 * (e - k * IPS);
                        stp[r] = *(MPC_GLOBAL const D2 *)(G3 + (size_t)(c >= 18 ? imin(k + 1, Nl) : k) * W3 + O_DW + c);
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int k = e / IPS, c = 2 * (e - k * IPS);
                            D2 v = stp[r];
                            if (c >= 18 && k >= Nl) { v.x = 0.0; v.y = 0.0; }              // no multiplier beyond the last dynamics
                            if (k == 0 && c >= 6 && c < 18) {                                // x_0 = x_hat (lbx_0 = ubx_0)
                                v.x = sm.xhat[c - 6] - gld(G1 + O_X + c - 6); v.y = sm.xhat[c - 5] - gld(G1 + O_X + c - 5);
                            }
                            *(MPC_GLOBAL D2 *)(G1 + (size_t)k * W1 + O_QW + c) = v;
                        }
                    }
                });
            }
        }
        {
            constexpr int IPS = 24, R = rounds_for(IPS);
            const int items = NS * IPS;
            for (int base = 0; base < items; base += R * NT) {
                ex.wpar([&](int lane) {
                    D2 stp[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = imin(base + r * NT + lane, items - 1), k = e / IPS, q = 2 * (e - k * IPS);
                        stp[r] = *(MPC_GLOBAL const D2 *)(G3 + (size_t)k * W3 + O_DT + (q >= 24 ? q - 24 : 0));
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int e = base + r * NT + lane;
                        if (e < items) {
                            const int k = e / IPS, q = 2 * (e - k * IPS);
                            D2 v = stp[r];
                            if (q < 24) { v.x = 0.0; v.y = 0.0; }
                            *(MPC_GLOBAL D2 *)(G1 + (size_t)k * W1 + O_QLAM + q) = v;
                        }
                    }
                });
            }
        }
        ex.wpar([&](int lane) {
            if (lane < NTASK) {
                const int i = lane;
                double v = gld(G2 + O_R + i);
#pragma unroll
                for (int j = 0; j < 6; j++) v += gld(G2 + O_GQ + i * 6 + j) * (sm.xhat[j] - gld(G1 + O_X + j));
                if (i == 4) {
#pragma unroll
                    for (int j = 0; j < 6; j++) v += gld(G2 + O_GV + j) * (sm.xhat[6 + j] - gld(G1 + O_X + 6 + j));
                }
                gst(G2 + O_Y + i, P.w_task[i] * v);
            }
        });
        ex.barrier();
        PROF_ADD(PF_RES, t0);
    }

    // =========================================================================== IPM driver
    // Restates HPIPM's d_ocp_qp_ipm_solve main loop (see oracle/mpc_oracle.c ipm_solve).
    // Returns HPIPM status 0 ok / 1 max-iter / 2 min-step / 3 NaN.
    // `defer_commit` (SQP_RTI): an accepted fast-path candidate is left in the Newton-step slots with commit_pending set; the NLP pass
    // that follows writes it to the QP iterate together with its own update (nlp_direct fuse_commit).
    MPC_HD int ipm_solve(int *iters_out, bool defer_commit = false)
    {
        const double tol = ex.smem().P.qp_tol;
        Smem &sm = ex.smem();
        const bool res = resident_ok();   // the horizon's factor fits the LDS pool: resident sweeps
        const bool reg = reg_ok();        // no room in LDS (half a pool, or a long horizon): the factor in registers / straight from HBM
        const bool seg = !reg && segment_ok();    // a segment of the factor fits (whole pool of a CU): segment-resident sweeps
#ifdef MPC_EMU_TRACE
        { static int once = 0; if (!once) { once = 1; fprintf(stderr, "[emu] res %d reg %d seg %d pool %d\n", (int)res, (int)reg, (int)seg, (int)ex.smem().pool_n); } }
#endif
        // bound-inactive fast path first (mpc_ipm.h; oracle solve_qp): `iters_out` counts Riccati factorisations
        int tried = 0;
        if (ex.uni(sm.P.fast_off == 0.0)) {
            if (fast_skip > 0) fast_skip--;
            else {
                tried = 1;
                if (!rhs_valid) fast_rhs();   // (SQP_RTI: normally formed by the previous step's NLP pass, nlp_direct want_rhs)
                rhs_valid = false;
                double ok;
                if (res) { fact_pass_t<1>(); ok = fwd_resident<false, false, false, true>(); }
                else if (reg) { fact_pass_t<2>(); ok = fwd_resident<false, true, true, true>(); }
                else if (seg) { fact_pass_t<2>(); ok = fwd_resident<false, true, false, true>(); }
                else { fact_pass_t<0>(); ok = forward_step_pass<false, true>(); }
                if (ex.uni(ok > 0.5)) {
                    if (defer_commit) commit_pending = true;
                    else fast_commit();
                    fast_back = 0;
#ifdef MPCB_PROFILE
                    prof[PF_COUNT_IPM] += 1;
#endif
                    *iters_out = 1;
                    return 0;
                }
                fast_back = ipm::fast_backoff(fast_back);
                fast_skip = fast_back;
            }
        }
        rhs_valid = false;            // (the interior-point passes rebuild Gamma, gt, rb)
        residual_direct(0, 0.0);
        const double nc = ex.uni(sm.ret[5]);
        double mu = nc > 0 ? ex.uni(sm.ret[4]) / nc : 0.0;
        int it = 0, status = 1;
        double alpha = 1.0;
        for (;; it++) {
            // every lane holds the same scalars; the decision is made uniform explicitly (scalar branch)
            const double n0 = sm.ret[0], n1 = sm.ret[1], n2 = sm.ret[2], n3 = sm.ret[3];
            int stop = -1;
            if (n0 != n0 || n1 != n1 || n2 != n2 || n3 != n3) stop = 3;
            else if (!(n0 > tol || n1 > tol || n2 > tol || n3 > tol)) stop = 0;
            else if (it >= c.pb->qp_iter_max) stop = 1;
            else if (!(alpha > 1e-12)) stop = 2;
            stop = ex.uni(stop);
#ifdef MPCB_DIAG_FIXED_IT   // timing-only diagnostic builds: the same work whatever the (possibly knocked-out) arithmetic
            stop = it >= MPCB_DIAG_FIXED_IT ? 0 : -1;
#endif
            if (stop >= 0) { status = stop; break; }
            const bool has_bounds = ex.uni(nc > 0);
            double a_aff;
            if (res) {
                fact_pass_t<1>();
                a_aff = has_bounds ? fwd_resident<true, false>() : fwd_resident<false, false>();
            } else if (reg) {
                fact_pass_t<2>();
                a_aff = has_bounds ? fwd_resident<true, true, true>() : fwd_resident<false, true, true>();
            } else if (seg) {
                fact_pass_t<2>();
                a_aff = has_bounds ? fwd_resident<true, true>() : fwd_resident<false, true>();
            } else {
                fact_pass_t<0>();
                a_aff = has_bounds ? forward_step_pass<true>() : forward_step_pass<false>();
            }
            if (has_bounds) {
                const bool rs = res || seg || reg;
                const double S0 = rs ? ex.uni(sm.cen[1]) : ex.get1(sm.red[1]), S1 = rs ? ex.uni(sm.cen[2]) : ex.get1(sm.red[2]),
                             S2 = rs ? ex.uni(sm.cen[3]) : ex.get1(sm.red[3]);
                const double mu_aff = (S0 + a_aff * (S1 + a_aff * S2)) / nc;
                const double sigma = ipm::sigma(mu_aff, mu);
                if (res) { corr_resident<false>(sigma * mu); alpha = fwd_resident<false, false>(); }
                else if (reg) { corr_resident<true, true>(sigma * mu); alpha = fwd_resident<false, true, true>(); }
                else if (seg) { corr_resident<true>(sigma * mu); alpha = fwd_resident<false, true>(); }
                else { corrector_bwd_pass(sigma * mu); alpha = forward_step_pass<false>(); }
            } else {
                alpha = a_aff;
            }
            const double a = ipm::step_scale(alpha);
            residual_direct(1, a);
            mu = nc > 0 ? ex.uni(sm.ret[4]) / nc : 0.0;
        }
#ifdef MPCB_PROFILE
        prof[PF_COUNT_IPM] += it + tried;
#endif
        *iters_out = it + tried;
        return status;
    }

    // =========================================================================== SQP line search
    // The L1 merit function (acados ocp_nlp_evaluate_merit_fun restated) at the trial points (X,U) + alpha_g (dX,dU), g < na, in ONE
    // pass: the stages of a trial point are one lane each, so a horizon of up to 127 stages keeps only two wavefronts busy -- the
    // other wavefronts of the simulation evaluate the NEXT step length of the backtracking sequence at the same time (two trial
    // points per pass); the records are staged through LDS once for both.  Each value is the
    // same sum, in the same order, as a pass of its own would form.  With `update_weights` the merit weights are first refreshed
    // from the QP multipliers by Leineweber's rule (acados merit_backtracking_*_weights).
    static constexpr int MERIT_MAX = 2;     // (four at eight wavefronts measured slower: the speculative trial points share the SIMDs of the two that count)
    MPC_HD int merit_lanes() const                       // lanes per trial point: whole wavefronts, as many as stages if possible
    {
        const int NS = ex.uni(ex.smem().n_hor) + 1;
        int lpg = NT < 2 * WAVE ? NT : 2 * WAVE;
        while (lpg < NS && lpg < NT) lpg *= 2;
        return lpg;
    }
    MPC_HD int merit_groups() const { return imin(MERIT_MAX, imax(1, NT / merit_lanes())); }
    MPC_PASS void merit_pass(const double *alphas, int na, bool update_weights, int sqp_iter, double *out)
    {
        PROF_T0(t0);
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        const Robot &rb = sm.rb;
        const int Nl = ex.uni(ex.smem().n_hor);
        constexpr int WMW = 36, LMW = 38;   // 36 merit weights per stage; LDS row stride 38 (lane <-> stage accesses: no bank aliasing)
        constexpr int LVT = 22;   // per-stage and trial-point scratch in LDS: trial state (12) | task residual record (8); a local array would live in scratch memory
        const int LPG = merit_lanes();                   // lanes per trial point (trial point g: lanes [g LPG, (g + 1) LPG))
        const int CH = chunk_len(L1 + LMW + LVT * na, L1);
        double al[MERIT_MAX];
#pragma unroll
        for (int g = 0; g < MERIT_MAX; g++) { al[g] = g < na ? alphas[g] : 0.0; out[g] = 0.0; }
        for (int k0 = 0; k0 <= Nl; k0 += CH) {
            const int k1 = imin(k0 + CH - 1, Nl), hi = imin(k1 + 1, Nl);
            double *v1 = ex.pool();                        // rows k0..hi, L1
            double *vm = v1 + (size_t)(CH + 1) * L1;    // rows k0..k1, MW
            double *vt = vm + (size_t)CH * LMW;         // [trial point][rows k0..k1]: trial state and its task residual
            copies([&](int lane, auto nl) {
                constexpr int NL = decltype(nl)::value;
                copy_lanes<W1, 0, W1, L1, true, NL>(v1, ex.smem().w.G1, k0, hi, lane);
                copy_lanes<WMW, O_MW, W5, LMW, true, NL>(vm, ex.smem().w.G5, k0, k1, lane);
            });
            if (update_weights) {
                ex.par([&](int lane) {
                    const int rows = k1 - k0 + 1;
                    for (int e = lane; e < rows * WMW; e += NT) {
                        const int s = e / WMW, i = e - s * WMW;
                        const double *r1 = v1 + (size_t)s * L1;
                        const double a = i < 12 ? fabs(r1[O_QPI + i]) : fabs(r1[O_QLAM + i - 12]);
                        double *mw = vm + (size_t)s * LMW + i;
                        *mw = sqp_iter == 0 ? a : fmax(a, 0.5 * (*mw + a));
                    }
                });
                copy_rect<WMW, O_MW, W5, LMW, false>(vm, ex.smem().w.G5, k0, k1);
            }
            ex.par([&](int lane_) {
                const int g = lane_ / LPG, lane = lane_ - g * LPG;
                double alpha = al[0];
#pragma unroll
                for (int j = 1; j < MERIT_MAX; j++) alpha = g == j ? al[j] : alpha;
                double acc = 0.0;
                if (g < na) {
                for (int k = k0 + lane; k <= k1; k += LPG) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MPCB_NO_LICM_BLOCK)
                    // (the parameter block is read from LDS where it is used: hoisted out of this loop -- ~60 values -- it is spilled to
                    // scratch memory as a whole by the 256-register builds)
                    asm volatile("" ::: "memory");
#endif
                    const double *r1 = v1 + (size_t)(k - k0) * L1, *rn = r1 + L1;
                    const double *mw = vm + (size_t)(k - k0) * LMW;
                    double *xx = vt + ((size_t)g * CH + (k - k0)) * LVT, *rec = xx + 12;   // task_lin<false> only writes rec[O_R..O_R+4]
                    double uu[6];
#pragma unroll
                    for (int i = 0; i < 12; i++) xx[i] = r1[O_X + i] + alpha * r1[O_QW + 6 + i];
                    if (k < Nl) {
#pragma unroll
                        for (int i = 0; i < 6; i++) uu[i] = r1[O_U + i] + alpha * r1[O_QW + i];
                        task_lin<false>(rb, P, xx, xx + 6, rec);
                        double s = 0.0;
#pragma unroll
                        for (int i = 0; i < NTASK; i++) s += P.w_task[i] * rec[O_R + i] * rec[O_R + i];
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            const double qdd = P.cq[j] * (uu[j] - xx[6 + j]);
                            s += 2.0 * P.w_u * uu[j] * uu[j] + P.w_qddot * qdd * qdd;
                            const double xnq = rn[O_X + j] + alpha * rn[O_QW + 6 + j];
                            const double xnv = rn[O_X + 6 + j] + alpha * rn[O_QW + 12 + j];
                            acc += mw[j] * fabs((xx[j] + P.a12[j] * xx[6 + j] + P.b1[j] * uu[j]) - xnq);
                            acc += mw[6 + j] * fabs((P.a22[j] * xx[6 + j] + P.b2[j] * uu[j]) - xnv);
                            // branch-free (every lane-dependent branch costs a saved exec mask: this phase ran out of SGPRs)
                            const double vl = P.umin[j] - uu[j], vu = uu[j] - P.umax[j];
                            acc += mw[12 + j] * fmax(vl, 0.0) + mw[24 + j] * fmax(vu, 0.0);
                            const double ql = P.qmin[j] - xx[j], qu = xx[j] - P.qmax[j], on = k >= 1 ? 1.0 : 0.0;
                            acc += on * (mw[18 + j] * fmax(ql, 0.0) + mw[30 + j] * fmax(qu, 0.0));
                        }
                        acc += 0.5 * P.dt * s;
                    }
                    if (k == 0) {
#pragma unroll
                        for (int i = 0; i < 12; i++) acc += sm.w.state[13 + i] * fabs(sm.xhat[i] - xx[i]);
                    }
                }
                }
                ex.put_wsum(sm.red[0], lane_, acc);     // one partial per wavefront: a trial point's are consecutive
            });
            // combine the wavefronts of each trial point (same order in every lane)
            {
                const int wpg = LPG / WAVE;
#pragma unroll
                for (int g = 0; g < MERIT_MAX; g++)
                    if (g < na) out[g] += ex.get_sum_range(sm.red[0], g * wpg, wpg);
            }
        }
        PROF_ADD(PF_MERIT, t0);
    }

    // Merit weight of the eliminated x_0 constraint: |stage-0 stationarity of the QP wrt x_0|
    MPC_PASS void update_x0_weights(int sqp_iter)
    {
        Smem &sm = ex.smem();
        ex.par([&](int lane) {
            if (lane < NX) {
                const InstParams &P = sm.P;
                const double *r1 = ex.smem().w.G1, *r2 = ex.smem().w.G2;  // stage 0 records in HBM (y holds W(r + G delta))
                double v;
                if (lane < 6) {
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < NTASK; i++) s += r2[O_GQ + i * 6 + lane] * r2[O_Y + i];
                    v = P.dt * s + r1[O_QPI + lane] + P.dt * P.lm * r1[O_QW + 6 + lane];
                } else {
                    const int jj = lane - 6;
                    const double uj = r1[O_U + jj] + r1[O_QW + jj], vj = r1[O_X + 6 + jj] + r1[O_QW + 12 + jj];
                    const double c2 = P.w_qddot * P.cq[jj] * P.cq[jj];
                    v = P.dt * (r2[O_GV + jj] * r2[O_Y + 4] + c2 * (vj - uj)) + P.a12[jj] * r1[O_QPI + jj] +
                        P.a22[jj] * r1[O_QPI + 6 + jj] + P.dt * P.lm * r1[O_QW + 12 + jj];
                }
                const double a = fabs(v);
                double *mw = &ex.smem().w.state[13 + lane];
                *mw = sqp_iter == 0 ? a : fmax(a, 0.5 * (*mw + a));
            }
        });
    }

    // MERIT_BACKTRACKING (trajectory_optimizer.py:68; acados alpha_reduction 0.7, alpha_min 0.05)
    MPC_HD double line_search(int sqp_iter)
    {
        update_x0_weights(sqp_iter);
        const int G = merit_groups();
        // the backtracking sequence is fixed (1, 0.7, 0.49, ...): G trial points per pass, the reference point 0 in the first
        double al[MERIT_MAX], m[MERIT_MAX], m0 = 0.0, alpha = 1.0;
        bool first = true;
        while (alpha >= 0.05) {
            int na = 0;
            if (first) al[na++] = 0.0;
            double a = alpha;
            while (na < G && a >= 0.05) { al[na++] = a; a *= 0.7; }
            merit_pass(al, na, first, sqp_iter, m);
            int j = 0;
            if (first) { m0 = m[0]; j = 1; first = false; }
            for (; j < na; j++) {
                if (ex.uni(m[j] < m0)) return al[j];
                alpha = al[j] * 0.7;
            }
        }
        return alpha;
    }

    // One solver.solve() call (simulator.py:210-221).  On entry sm.xhat holds the feedback
    // state and G2 holds the linearisation at the current iterate when `lin_valid`; on exit it
    // is valid for the (new) iterate again, together with its cost and NLP residuals.
    // `plant_done` (out): the plant step and the log of the new state were done inside the last NLP pass (SQP_RTI)
    MPC_HD int nlp_step(bool &lin_valid, int *sqp_iter_out, int *qp_iter_out, double *res4, double *cost_out, bool *plant_done)
    {
        *plant_done = false;
        PROF_T0(t0);
        int status = 0, sqp_iter = 0, qp_iter = 0, it = 0;
        double cost = lin_cost;
        if (c.pb->solver_type == 1) {
            // SQP_RTI: one linearisation, one QP, full step
            if (!lin_valid) { cost = NLP_PASS(0.0, false, false, nullptr); rhs_valid = false; }
            commit_pending = false;
            const int qs = ipm_solve(&it, MPCB_FUSE != 0);
            qp_iter += it;
            sqp_iter = 1;
            const bool ok = qs == 0 || qs == 1;
            if (!ok) status = 4;  // ACADOS_QP_FAILURE, iterate untouched
            // residuals / cost are evaluated at the new iterate (acados get_residuals() for RTI,
            // get_cost()); this linearisation is reused by the next solve() call -- and so is the fast path's right-hand side,
            // formed by the same pass when the next QP will try the fast path (ipm_solve: fast_off == 0 and no suspension left)
            const bool next_fast = ex.uni(MPCB_FUSE != 0 && ex.smem().P.fast_off == 0.0 && fast_skip == 0);
            cost = NLP_PASS(1.0, ok, false, res4, true, commit_pending, next_fast);
            rhs_valid = next_fast;
            commit_pending = false;
            *plant_done = true;
            lin_valid = true;
        } else {
            const double tol = ex.smem().P.tol, tol_eq = ex.smem().P.tol_eq, tol_in = ex.smem().P.tol_ineq, tol_co = ex.smem().P.tol_comp;
            status = 2;  // ACADOS_MAXITER unless decided otherwise
            double alpha = 0.0;
            bool pending = false;  // a step (alpha) waits to be applied by the next nlp_pass
            for (sqp_iter = 0; sqp_iter < c.pb->max_iter; sqp_iter++) {
                if (pending || !lin_valid || sqp_iter == 0) {
                    cost = NLP_PASS(alpha, pending, true, res4);
                    pending = false;
                    lin_valid = true;
                }
                if (ex.uni(res4[0] < tol && res4[1] < tol_eq && res4[2] < tol_in && res4[3] < tol_co)) { status = 0; break; }
                if (ex.uni(res4[0] != res4[0] || cost != cost)) { status = 1; break; }
                const int qs = ipm_solve(&it);
                qp_iter += it;
                if (qs != 0 && qs != 1) { status = 4; break; }
                alpha = c.pb->fixed_step ? 1.0 : line_search(sqp_iter);
                pending = true;
            }
            if (pending) { cost = NLP_PASS(alpha, true, true, nullptr); lin_valid = true; }  // max-iter exit: residuals of the last check stay
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
        Smem &sm = ex.smem();
        const InstParams &P = sm.P;
        Ws &w = c.w;
        const int Nsim = c.pb->Nsim;
        const size_t sb = (size_t)inst * Nsim;
        bool lin_valid = false;
        int log_lo = step0 == 0 ? 0 : step0 + 1;   // first log column this launch produces
        if (step0 == 0) {
            // acados initial guess: x_k = x0, u_k = 0, all multipliers / QP memory 0 (SURVEY A.7 iv)
            const size_t tot = ws_doubles_per_instance(N);
            ex.par([&](int lane) {
                for (size_t e = lane; e < tot; e += NT) w.G1[e] = 0.0;  // G1 is the workspace base
            });
            ex.par([&](int lane) {
                for (int e = lane; e < (N + 1) * NX; e += NT) {
                    const int k = e / NX, i = e - k * NX;
                    w.G1[(size_t)k * W1 + O_X + i] = i < 6 ? P.q0[i] : P.qdot0[i - 6];
                }
                if (lane < NX) sm.xhat[lane] = lane < 6 ? P.q0[lane] : P.qdot0[lane - 6];
                if (lane < NU) sm.u0[lane] = P.qdot0[lane];  // u[:,0] = qdot_0 (simulator.py:81)
            });
            log_lo = ex.uni(log_state(out, inst, 0, log_lo, false));
        } else {
            ex.par([&](int lane) {
                if (lane < NX) sm.xhat[lane] = w.state[lane];
            });
            lin_cost = w.state[12];
            lin_valid = ex.uni(w.state[25] != 0.0);
            rhs_valid = false;     // (not carried across launches: the first fast attempt of a launch forms its right-hand side itself)
            fast_skip = ex.uni((int)w.state[26]); fast_back = ex.uni((int)w.state[27]);
        }
        for (int i = step0; i < step1; i++) {
            int sqp_iter = 0, qp_iter = 0;
            double res4[4] = {0, 0, 0, 0}, cost = 0.0;
            const double t0 = ex.clock();
            bool plant_done = false;
            const int status = nlp_step(lin_valid, &sqp_iter, &qp_iter, res4, &cost, &plant_done);
            const double t1 = ex.clock();
            PROF_T0(tp);
            // u = solver.get(0,'u'); RK4 plant step (simulation_model.py:111-117) -- unless the last NLP pass has done it
            ex.par([&](int lane) {
                if (lane < 6 && !plant_done) {
                    const int j = lane;
                    const double u = w.G1[O_U + j];
                    double qn, vn;
                    plant_rk(P, j, sm.xhat[j], sm.xhat[6 + j], u, qn, vn);
                    sm.logv[24 + j] = qn;
                    sm.logv[30 + j] = vn;
                    sm.u0[j] = u;
                }
                if (lane == 8) {
                    out.status[sb + i] = status;
                    out.sqp_iter[sb + i] = sqp_iter;
                    out.qp_iter[sb + i] = qp_iter;
                    out.cost[sb + i] = cost;
                    // (SQP_RTI: the plant step and its log ran on a spare lane INSIDE the last NLP pass, i.e. inside [t0, t1]: its time is
                    // reported as plant_time and taken out here, so that solver_time is the solve alone -- acados time_tot,
                    // simulator.py:220 -- and the two columns do not count the same microseconds twice)
                    out.solver_time[sb + i] = (t1 - t0) - (plant_done ? sm.ret[6] : 0.0);
                }
                if (lane >= 12 && lane < 16) out.residuals[(sb + i) * 4 + (lane - 12)] =
                    lane == 12 ? res4[0] : (lane == 13 ? res4[1] : (lane == 14 ? res4[2] : res4[3]));
            });
            ex.par([&](int lane) {
                if (lane < NX) sm.xhat[lane] = sm.logv[24 + lane];
            });
            log_lo = ex.uni(log_state(out, inst, i + 1, log_lo, plant_done));
            const double t2 = ex.clock();
            // plant_time: what is left of it here + the plant lane's own time inside the NLP pass
            ex.par([&](int lane) { if (lane == 0) out.plant_time[sb + i] = (t2 - t1) + (plant_done ? sm.ret[6] : 0.0); });
            PROF_ADD(PF_PLANT, tp);
        }
        if (log_lo <= step1) log_flush(out, inst, log_lo, step1);   // the columns of a partly filled block
        ex.par([&](int lane) {
            if (lane < NX) w.state[lane] = sm.xhat[lane];
            if (lane == 12) { w.state[12] = lin_cost; w.state[25] = lin_valid ? 1.0 : 0.0; w.state[26] = fast_skip; w.state[27] = fast_back; }
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

    // simulation_model.py:87-90: log state, input, FK pose, rpy, J*qdot at column `col`, plus the task errors of
    // Simulator.errors (simulator.py:265-344) of that column.  Columns collect in LDS (sm.logbuf) and leave as
    // contiguous runs of up to LOGB columns per row: `log_lo` = first column of the current block still in LDS.
    // `computed`: sm.logv already holds the pose / rpy / J qdot / task errors of sm.xhat (done inside the last NLP pass)
    MPC_PASS int log_state(const Outputs &out, int inst, int col, int log_lo, bool computed)
    {
        Smem &sm = ex.smem();
        const Robot &rb = sm.rb;
        if (!computed) ex.par([&](int lane) {
            if (lane == 0) {
                double z[12];
#pragma unroll
                for (int i = 0; i < 12; i++) z[i] = sm.xhat[i];
                plant_log(rb, z, sm.logv);
                task_errors(sm.P, rb, sm.logv, sm.logv + 15, sm.logv + 36);
            }
        });
        ex.par([&](int lane) {
            if (lane < LOG_ROWS) {
                const double v = lane < 12 ? sm.xhat[lane]
                               : lane < 18 ? sm.u0[lane - 12]
                               : lane < 30 ? sm.logv[lane - 18]
                               : lane < 33 ? sm.logv[12 + (lane - 30)]
                               : lane < 39 ? sm.logv[15 + (lane - 33)]
                                           : sm.logv[36 + (lane - 39)];
                sm.logbuf[lane][col & (LOGB - 1)] = v;
            }
        });
        col = ex.uni(col); log_lo = ex.uni(log_lo);   // wave-uniform: scalar branch
        if ((col & (LOGB - 1)) == LOGB - 1) { log_flush(out, inst, log_lo, col); log_lo = col + 1; }
        return log_lo;
    }

    // write columns [c_lo, c_hi] (all inside one block of LOGB columns) of every log row to HBM
    MPC_PASS void log_flush(const Outputs &out, int inst, int c_lo, int c_hi)
    {
        Smem &sm = ex.smem();
        const size_t T1 = (size_t)c.pb->Nsim + 1;
        ex.par([&](int lane) {
            for (int e = lane; e < LOG_ROWS * LOGB; e += NT) {
                const int row = e / LOGB, cc = (c_hi & ~(LOGB - 1)) + (e & (LOGB - 1));
                if (cc < c_lo || cc > c_hi) continue;
                double *dst = row < 12 ? out.z + ((size_t)inst * 12 + row) * T1
                            : row < 18 ? out.u + ((size_t)inst * 6 + (row - 12)) * T1
                            : row < 30 ? out.ee_pose + ((size_t)inst * 12 + (row - 18)) * T1
                            : row < 33 ? out.ee_rpy + ((size_t)inst * 3 + (row - 30)) * T1
                            : row < 39 ? out.ee_vel + ((size_t)inst * 6 + (row - 33)) * T1
                                       : out.errors + ((size_t)inst * 7 + (row - 39)) * T1;
                dst[cc] = sm.logbuf[row][e & (LOGB - 1)];
            }
        });
    }
};

}  // namespace mpcb
