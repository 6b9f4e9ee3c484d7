// mpc_stream.h -- the THROUGHPUT engine of the batched MPC rollout (gfx950): one wavefront = one simulation,
// several wavefronts per SIMD, stage records STREAMED through a few KB of LDS.
//
// mpc_core.h is the latency engine: one workgroup of 2-4 wavefronts and (almost) a whole CU's LDS per simulation,
// big chunks of the horizon staged in LDS, the sequential Riccati recursion of ONE wavefront as the critical path.
// That design idles a CU's issue slots once there are many more simulations than CUs (batch >= ~1000 per GPU:
// BASELINE configs[2], [3]).  Here the roles are collapsed into one wavefront per simulation:
//   * every pass of the interior-point iteration is a loop over the stages; the records of the next stage of the
//     sweep are fetched -- 16 B per lane, one to three instructions per record bundle -- while the current stage is
//     computed, and the results of a stage leave as soon as they exist.  No chunk pool, no halo rows, no
//     `s_barrier` (a wavefront's LDS operations complete in issue order);
//   * what hides the memory and LDS latency is not a second wavefront of the same simulation but the OTHER
//     simulations resident on the same SIMD (2 at 256 VGPRs), each with < 20 KB of LDS;
//   * in every stage the lanes of the wavefront take different roles at once (matrix recursion | vector recursion;
//     state recursion | input step | multiplier step | step-length terms of the stage before).  Roles side by side in
//     one wavefront execute one after the other (SIMT), so what matters is the length of that chain: every phase of a
//     stage first fetches the operands of ALL its roles -- per-lane index tables, idle lanes read offset 0, values held
//     by pin() -- and only then runs the exec-masked role blocks, on registers (one LDS round trip per phase, not one
//     per block); roles that multiply a factor row with the same vector share one dot product.
// Record groups G1..G3 in HBM are those of mpc_layout.h; the factor record G4 has its fields padded to multiples
// of four scalars (S* offsets below) so that every bundle is a whole number of 16-byte items in fp64 AND fp32.
//
// The Riccati factor and the three solve sweeps are templated on a scalar type FT: FT = double is the reference
// arithmetic; FT = float is the "fp32 Riccati" leg of BASELINE configs[4] (factor K, P, R~^-1, p and the sweep
// recursions in fp32 -- half the bytes of the largest record; iterate, residuals, right-hand sides, step lengths,
// multiplier steps and all logged outputs stay fp64).
//
// Algorithm and formulas: those of mpc_core.h (acados SQP / SQP_RTI + HPIPM Mehrotra IPM + Riccati, restated from
// trajectory_optimizer.py:57-176 and simulator.py:199-241); see there for derivations.  Both solver types are
// implemented; which batches are sent here: mpc_kernel.hip pick_engine (SQP_RTI from 1280 simulations, full SQP from 2560
// simulations x 100 steps, every fp32-Riccati and every ragged batch).
#pragma once
#include "mpc_core.h"

namespace mpcb {
namespace se {

typedef double D2 __attribute__((ext_vector_type(2)));

// factor record of one stage, in scalars of type FT (fields padded to multiples of 4 scalars)
constexpr int SK = 0;      // K = R~^-1 S~ (6x12)
constexpr int SVH = 72;    // R~^-1 h_u (6) + 2 pad
constexpr int SEo = 80;    // e = rb - B R~^-1 h_u (12)
constexpr int SPV = 92;    // p_k (12)
constexpr int SPM = 104;   // P_k packed upper triangle (78) + 2 pad
constexpr int SWV = 184;   // w_k = P_{k+1} rb_k (12)
constexpr int SRI = 196;   // R~^-1 (36)
constexpr int SW4 = 232;
constexpr int SW4_AFF = SPV, SW4_FWD = SWV;

// One simulation's workspace: ONE record per stage, [G1 | G2 | G3 | G4] side by side (all the records a pass touches
// of a stage sit in the same few KB: fewer DRAM pages open per pass than with one array per group), then the scalars.
struct SWs {
    double *G1, *G2, *G3;   // row 0 of each group inside the stage record
    char *G4;               // SW4 scalars of the factor type
    double *G5;             // SQP extras (NLP multipliers, merit weights): present in SQP launches only
    double *state;
    int ld;                 // stage record stride (bytes)
};
template <class FT>
MPC_HD size_t sws_doubles_per_instance(int N, bool sqp = false)
{
    return (size_t)(N + 1) * (W1 + W2 + W3 + SW4 * sizeof(FT) / 8 + (sqp ? W5 : 0)) + STATE_DOUBLES;
}
template <class FT>
MPC_HD SWs sws_carve(double *base, int N, bool sqp = false)
{
    const size_t n1 = (size_t)N + 1;
    SWs w;
    w.ld = (W1 + W2 + W3 + (sqp ? W5 : 0)) * 8 + SW4 * (int)sizeof(FT);
    w.G1 = base;
    w.G2 = base + W1;
    w.G3 = base + W1 + W2;
    w.G4 = (char *)(base + W1 + W2 + W3);
    w.G5 = (double *)(w.G4 + SW4 * sizeof(FT));
    w.state = base + n1 * (size_t)(w.ld / 8);
    return w;
}

// --------------------------------------------------------------------------------------------- LDS of one simulation
constexpr int OUT_MAX = 232;   // doubles of the largest output bundle (factor record)
// Input ring: a flat LDS area cut into as many slots as fit the pass's bundle (rounded up to whole 1-KiB copy pieces:
// one global_load_lds_dwordx4 writes 64 x 16 B contiguously).  Slot = row % slots; rows k-1, k, k+1 of the sweep are
// valid while stage k is computed, the slots beyond hold rows in flight.
#ifndef MPCB_RING_DOUBLES
#define MPCB_RING_DOUBLES 1476
#endif
constexpr int RING_DOUBLES = MPCB_RING_DOUBLES;
constexpr int SLOGB = 2;       // log columns collected before they leave (LOGB of the latency engine is 8: here every KB of LDS is
                               // prefetch depth of the input ring, and the sweeps wait on the memory latency that depth hides)

struct SSmem {
    InstParams P;
    const Robot *rbp;             // kinematic constants: copied into the (idle) ring by the passes that need them, robot_in_lds()
    SWs w;
    int n_hor, pad0;
    alignas(16) double ring[RING_DOUBLES];
    alignas(16) double out[2][OUT_MAX];
    alignas(16) double Rt[36];        // R~ of the stage in work
    alignas(16) double St2[2][72];    // S~ (stage k read, k-1 written)
    alignas(16) double Kf[72];        // K of the stage in work
    alignas(16) double vec[4][16];    // 12-vectors handed from phase to phase / stage to stage
    alignas(16) double gtc[2][20];    // corrected condensed gradient of a stage (corrector sweep, one stage ahead)
    alignas(16) double xhat[12];
    alignas(16) double u0[6];
    alignas(16) double logv[48];
    alignas(16) double logbuf[LOG_ROWS][SLOGB];
};
#ifndef MPCB_STREAM_WPE
#define MPCB_STREAM_WPE 2
#endif
static_assert(sizeof(SSmem) <= 163840 / (4 * MPCB_STREAM_WPE), "MPCB_STREAM_WPE wavefronts per SIMD share the 160 KiB of a CU");

#ifndef MPCB_RTI_R
#define MPCB_RTI_R 2
#endif
#define SE_DEV __device__ __forceinline__
#define SE_PASS __device__ __noinline__

#if defined(__HIP_DEVICE_COMPILE__)
__shared__ __attribute__((aligned(16))) SSmem g_ssm;

// ---- wave-level primitives (one wavefront = one simulation: no workgroup barrier anywhere) ----
SE_DEV void fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int CTRL>
SE_DEV double dpp(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
SE_DEV double lane_val(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <class Op>
SE_DEV double wave_reduce(double v, Op op)
{
    v = op(v, dpp<0xB1>(v));
    v = op(v, dpp<0x4E>(v));
    v = op(v, dpp<0x141>(v));
    v = op(v, dpp<0x140>(v));
    return op(op(lane_val(v, 0), lane_val(v, 16)), op(lane_val(v, 32), lane_val(v, 48)));
}
struct OpSum { SE_DEV double operator()(double a, double b) const { return a + b; } };
struct OpMax { SE_DEV double operator()(double a, double b) const { return fmax(a, b); } };
struct OpMin { SE_DEV double operator()(double a, double b) const { return fmin(a, b); } };
SE_DEV double wsum(double v) { return wave_reduce(v, OpSum()); }
SE_DEV double wmax(double v) { return wave_reduce(v, OpMax()); }
SE_DEV double wmin(double v) { return wave_reduce(v, OpMin()); }
SE_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
SE_DEV double unid(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
SE_DEV double wclock() { return (double)wall_clock64() * 1e-8; }
// Keeps a value that was loaded ahead of its (exec-masked) use where it is: without it the compiler sinks the load into
// the role's block, and every block then waits out its own LDS round trip (one s_waitcnt lgkmcnt(0) per block).
SE_DEV void pin(double &v) { asm volatile("" : "+v"(v)); }
SE_DEV void pin(D2 &v) { asm volatile("" : "+v"(v)); }

// --------------------------------------------------------------------------------------------- record bundles
// A BUNDLE is what one stage of a pass reads (or writes): a few segments (byte ranges, multiples of 16 B) of the
// stage's rows in the record groups, concatenated in LDS.  Lane l of copy instruction j moves 16-byte item
// j*64+l; which segment the item belongs to, its address in row 0 and the row stride are worked out once per pass.
struct Seg {
    char *base;     // group array, row 0
    int ld;         // row stride (bytes)
    int off, w;     // byte range inside the row (multiples of 16)
};
SE_DEV Seg segd(double *base, int ld_bytes, int c0, int w) { Seg s; s.base = (char *)base; s.ld = ld_bytes; s.off = c0 * 8; s.w = w * 8; return s; }
template <class FT>
SE_DEV Seg segf(char *base, int ld_bytes, int c0, int w) { Seg s; s.base = base; s.ld = ld_bytes; s.off = c0 * (int)sizeof(FT); s.w = w * (int)sizeof(FT); return s; }

// ITEMS (when given): the bundle's number of 16-byte items -- copy instructions that are full (every lane busy) then
// need no exec mask (a v_cmp / s_and_saveexec / s_cbranch / s_or sequence per instruction and stage otherwise).
template <int NI, int ITEMS = 0>
struct Bundle {
    static_assert(ITEMS == 0 || (ITEMS + WAVE - 1) / WAVE == NI, "bundle size and copy instruction count disagree");
    static constexpr bool full(int j) { return ITEMS > 0 && (j + 1) * WAVE <= ITEMS; }
    MPC_GLOBAL char *g[NI];   // this lane's item in row 0 (nullptr: lane idle in this instruction)
    int stride[NI];           // bytes per row
    MPC_GLOBAL char *p[NI];   // running pointer: this lane's item in the next row to be copied (a sweep visits consecutive
    int step[NI];             // rows: one 64-bit add per copy instead of a 64-bit multiply-add, which runs at quarter rate)
    SE_DEV void seek(int k, int dir)
    {
#pragma unroll
        for (int j = 0; j < NI; j++) {
            p[j] = g[j] ? g[j] + (long long)k * stride[j] : nullptr;
            step[j] = dir * stride[j];
        }
    }
    template <int NS>
    SE_DEV void setup(const Seg (&s)[NS], int lane)
    {
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const int e = j * WAVE + lane;
            g[j] = nullptr;
            stride[j] = 0;
            int cum = 0;
#pragma unroll
            for (int i = 0; i < NS; i++) {
                const int n = s[i].w / 16;
                if (e >= cum && e < cum + n) {
                    g[j] = (MPC_GLOBAL char *)(s[i].base + s[i].off) + (size_t)(e - cum) * 16;
                    stride[j] = s[i].ld;
                }
                cum += n;
            }
        }
    }
};
// Asynchronous fetch of row k of a bundle straight into LDS (global_load_lds_dwordx4: no VGPR destination, the data
// land at slot + j KiB + lane * 16 B; completion is visible only through vmcnt).
template <int NI, int BI>
SE_DEV void dma_issue(Bundle<NI, BI> &b, double *slot)
{
#pragma unroll
    for (int j = 0; j < NI; j++) {
#ifndef MPCB_NODMA
        if (Bundle<NI, BI>::full(j) || b.g[j])
            __builtin_amdgcn_global_load_lds((const MPC_GLOBAL void *)b.p[j], (MPC_LOCAL void *)(slot + j * 2 * WAVE), 16, 0, 0);
#else
        (void)slot;
#endif
        b.p[j] += b.step[j];
    }
}
// stores the NEXT row of the bundle's sweep (seek() names the first one and the direction)
template <int NI, int BI>
SE_DEV void store_out(Bundle<NI, BI> &b, const double *lds, int lane)
{
#pragma unroll
    for (int j = 0; j < NI; j++)
    {
#ifndef MPCB_NOSTORE
        if (Bundle<NI, BI>::full(j) || b.g[j]) *(MPC_GLOBAL D2 *)b.p[j] = ((const MPC_LOCAL D2 *)lds)[j * WAVE + lane];
#endif
        b.p[j] += b.step[j];
    }
}
constexpr int ni_of(int bytes) { return (bytes / 16 + WAVE - 1) / WAVE; }

// s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14)
template <int n>
SE_DEV void wait_vm()
{
    static_assert(n >= 0 && n < 64, "vmcnt is six bits wide");
    __builtin_amdgcn_s_waitcnt((n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14));
}

// Sweep driver: calls body(k, ring) for the rows k of a forward (0..N) or backward (N..0) sweep with row k and the
// NEXT row of the sweep landed in the ring (the row before is still valid as well), while D further rows are in
// flight.  Vector-memory operations complete in issue order (MI355X_MICROARCH.md, s_waitcnt), so "row i+1 has landed"
// is a counted wait on what was issued after it: D-1 fetches of NI instructions and D-1 stores of NO instructions in the
// steady state -- fewer stores in the first D stages, fewer fetches in the last D, where the smaller (safe) count is used.
// The fetch of row i+1+D is issued after stage i has been computed: it overwrites the slot of row i-1.
struct Ring {
    int slot_doubles, slots;
    SE_DEV double *row(int k) const { return g_ssm.ring + (k % slots) * slot_doubles; }
};
// ITEMS = 16-byte items of the bundle: a slot is exactly that long (the last copy instruction of a row is
// exec-masked beyond its items, so nothing spills into the next slot)
template <int ITEMS>
SE_DEV Ring make_ring()
{
    Ring rg;
    rg.slot_doubles = ITEMS * 2;
    rg.slots = RING_DOUBLES / rg.slot_doubles > 12 ? 12 : RING_DOUBLES / rg.slot_doubles;
    return rg;
}
// The kinematic constants (105 doubles) are needed by the lane-per-stage passes and the plant log only -- never while a
// sweep streams through the ring: they borrow its first bytes instead of holding LDS of their own.
SE_DEV const Robot &robot_in_lds()
{
    SSmem &sm = g_ssm;
    wait_vm<0>();                                            // no fetch of an earlier sweep is still landing in the ring
    fence();
    const double *rs = reinterpret_cast<const double *>(sm.rbp);
    for (int e = threadIdx.x; e < (int)(sizeof(Robot) / sizeof(double)); e += WAVE) sm.ring[e] = rs[e];
    fence();
    return *reinterpret_cast<const Robot *>(sm.ring);
}
template <int ITEMS, int NO, bool BACK, int NI, int BI, class F>
SE_DEV void sweep(Bundle<NI, BI> &bin, int N, int lane, F &&body)
{
    static_assert((ITEMS + WAVE - 1) / WAVE == NI && (BI == 0 || BI == ITEMS), "bundle size and copy instruction count disagree");
    constexpr int SLOT = ITEMS * 2;                          // doubles
    constexpr int SLOTS = RING_DOUBLES / SLOT > 12 ? 12 : RING_DOUBLES / SLOT;
    constexpr int D = SLOTS - 2;
    static_assert(D >= 1, "ring too small for this bundle");
    // issue order per stage: [wait] compute, stores of row i, fetch of row i+1+D
    constexpr int W_STEADY = (D - 1) * (NI + NO), W_EARLY = (D - 1) * NI, W_LATE = (D - 1) * NO;
    static_assert(W_STEADY < 64, "vmcnt range");
    const Ring rg = make_ring<ITEMS>();
    auto row = [&](int i) { return BACK ? N - i : i; };
    (void)lane;
    wait_vm<0>();                                            // nothing of an earlier pass is in flight
    fence();
    bin.seek(row(0), BACK ? -1 : 1);
    for (int j = 0; j <= D; j++)
        if (j <= N) dma_issue(bin, rg.row(row(j)));
    if (N + 1 <= D) wait_vm<0>();                            // short horizon: everything was issued above
    // ring slots of the row in work, the next and the previous row of the sweep, and the row to fetch: stepped, not
    // recomputed (k % slots is a multiply-high sequence on the scalar unit, several times per stage)
    constexpr int STEP = BACK ? SLOTS - 1 : 1;               // +1 forward, -1 backward (mod SLOTS)
    int s_cur = row(0) % SLOTS;
    int s_nxt = (s_cur + STEP) % SLOTS, s_prv = (s_cur + SLOTS - STEP) % SLOTS;
    int s_dma = row(N + 1 > D ? D + 1 : 0) % SLOTS;          // slot of row(i + 1 + D) at i = 0
    for (int i = 0; i <= N; i++) {
        if (N + 1 > D) {
            if (i + 1 > N) wait_vm<0>();
            else if (i + D > N) { if (i < D) wait_vm<0>(); else wait_vm<W_LATE>(); }
            else if (i < D) wait_vm<(W_EARLY < W_STEADY ? W_EARLY : W_STEADY)>();
            else wait_vm<W_STEADY>();
        }
        fence();
        body(row(i), g_ssm.ring + s_cur * SLOT, g_ssm.ring + s_nxt * SLOT, g_ssm.ring + s_prv * SLOT);
        if (i + 1 + D <= N) dma_issue(bin, g_ssm.ring + s_dma * SLOT);
        s_prv = s_cur; s_cur = s_nxt;
        s_nxt = s_nxt + STEP; s_nxt = s_nxt >= SLOTS ? s_nxt - SLOTS : s_nxt;
        s_dma = s_dma + STEP; s_dma = s_dma >= SLOTS ? s_dma - SLOTS : s_dma;
    }
}

// Stationarity element of class CLS (0: u_j, 1: q_j, 2: v_j) -- mpc_core.h Engine::stat_cls with the records in LDS.
// r1: G1 row of stage k, r2: [R..GV] of stage k, pk / pm: pi_k / pi_{k-1}.
template <int CLS>
SE_DEV double stat_cls(int N, int k, int j, const double *r1, const double *r2, bool with_delta, const double *pk, const double *pm)
{
    const InstParams &P = g_ssm.P;
    double val = 0.0;
    if (CLS == 0) {
        if (k >= N) return 0.0;
        double uj = r1[O_U + j], vj = r1[O_X + 6 + j];
        if (with_delta) { uj += r1[O_QW + j]; vj += r1[O_QW + 12 + j]; }
        const double c2 = P.w_qddot * P.cq[j] * P.cq[j];
        val = P.dt * (2.0 * P.w_u * uj + c2 * (uj - vj));
        val += P.b1[j] * pk[j] + P.b2[j] * pk[6 + j];
        if (with_delta) val += P.dt * P.lm * r1[O_QW + j];
    } else if (CLS == 1) {
        if (k == 0) return 0.0;
        if (k < N) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < NTASK; i++) s += r2[O_GQ + i * 6 + j] * r2[O_Y + i];
            val = P.dt * s + pk[j];
        }
        if (with_delta) val += (k < N ? P.dt : 1.0) * P.lm * r1[O_QW + 6 + j];
        val -= pm[j];
    } else {
        if (k == 0) return 0.0;
        if (k < N) {
            double uj = r1[O_U + j], vj = r1[O_X + 6 + j];
            if (with_delta) { uj += r1[O_QW + j]; vj += r1[O_QW + 12 + j]; }
            const double c2 = P.w_qddot * P.cq[j] * P.cq[j];
            val = P.dt * (r2[O_GV + j] * r2[O_Y + 4] + c2 * (vj - uj));
            val += P.a12[j] * pk[j] + P.a22[j] * pk[6 + j];
        }
        if (with_delta) val += (k < N ? P.dt : 1.0) * P.lm * r1[O_QW + 12 + j];
        val -= pm[6 + j];
    }
    return val;
}

struct IpmNorms {
    double ng, nb, nd, nm, smu, nc;
};

// =============================================================================================== residual pass
// MODE 0 (HPIPM warm start 2): keep (w, pi, lam, t) of the previous QP, clamp lam, t >= 0.1, embed x0.
// MODE 1: apply the Newton step of length `a`.  Then QP residuals, Gamma, the condensed gradient gt and rb.
// Forward sweep with one row of lookahead: stage k needs the UPDATED dw_{k+1} (dynamics residual) and the
// multiplier step stored with stage k+1; pi_{k-1} travels in LDS from the previous iteration.
//   in  : G1 row | G3 [DW..DT] | G2 [R..GV]            (234 doubles)
//   out : G1 [QW..QT] | G2 [R,Y] | G2 [GAM|GT|RB] | G3 [RG|RD|RM]   (196 doubles)
// MODE 2 = MODE 0 fused with nlp_res_pass<false> of the step BEFORE (SQP_RTI): the first pass of a step streams every
// record the NLP residual pass needs, so the dynamics defect BD (an input of this very pass), the cost and acados'
// residual norms of the previous step's iterate are formed here -- from the rows as they arrive, i.e. before the
// warm-start clamp touches lam, t -- and the separate sweep disappears.  nlp_out = [cost, stat, eq, ineq, comp];
// sm.vec[3] holds the x_hat the previous QP was solved for.  The output bundle then carries G2 [R..BD] (210 doubles).
// MODE 3 / 4 = right-hand side of the bound-inactive FAST PATH (mpc_ipm.h; mpc_core.h fast_rhs), plain / fused like MODE 2: the same
// sweep evaluated at (dw, pi, lam) = 0 with x_0 embedded and no bound terms -- Gamma = 0, gt = g, rb = b (+ A dx0 at stage 0) -- and
// WITHOUT the warm-start clamp: the previous QP's (w, pi, lam, t) in G1 stay as they are (only the x_0 embedding, which MODE 0 would
// write identically, and y go out), so a rejected attempt is followed by residual_pass<0> as if nothing had happened.
template <int MODE>
SE_PASS IpmNorms residual_pass(double a, double *nlp_out = nullptr)
{
    constexpr bool FUSE = MODE == 2 || MODE == 4;
    constexpr bool FASTM = MODE >= 3;
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor);
    const SWs w = sm.w;
    // FAST modes stream less (round 4): no Newton step comes in (dw = 0: G1 row | lin, 156 doubles) and neither the QP iterate
    // (untouched) nor the residual records rg | rd | rm (nobody reads them on the fast path) go out: [R, Y(, BD)] | [GAM | GT | RB].
    constexpr int I_D = 96, I_L = FASTM ? 96 : 174;          // input image: G1 row | D | lin   (FAST: G1 row | lin)
    constexpr int IN_ITEMS = FASTM ? 78 : 117;
    constexpr int RYW = FUSE ? O_BD + 12 : 10;               // G2 columns written from 0: [R, Y] or [R, Y, pad, BD]
    constexpr int O_W = 0, O_RY = FASTM ? 0 : 78, O_G = O_RY + RYW, O_3 = O_G + 42;   // output image
    constexpr int OUT_ITEMS = FASTM ? (RYW + 42) / 2 : (78 + RYW + 42 + 66) / 2;
    constexpr int OUT_NI = (OUT_ITEMS + WAVE - 1) / WAVE;
    Bundle<2, IN_ITEMS> bin;
    Bundle<OUT_NI, OUT_ITEMS> bout;
    {
        if (FASTM) { const Seg si[2] = {segd(w.G1, w.ld, 0, W1), segd(w.G2, w.ld, 0, W2_LIN)}; bin.setup(si, lane); }
        else { const Seg si[3] = {segd(w.G1, w.ld, 0, W1), segd(w.G3, w.ld, O_DW, 78), segd(w.G2, w.ld, 0, W2_LIN)}; bin.setup(si, lane); }
        if (FASTM) { const Seg so[2] = {segd(w.G2, w.ld, 0, RYW), segd(w.G2, w.ld, O_GAM, 42)}; bout.setup(so, lane); }
        else { const Seg so[4] = {segd(w.G1, w.ld, O_QW, 78), segd(w.G2, w.ld, 0, RYW), segd(w.G2, w.ld, O_GAM, 42), segd(w.G3, w.ld, 0, 66)}; bout.setup(so, lane); }
        bout.seek(0, 1);
    }
    double a_g = 0, a_b = 0, a_d = 0, a_m = 0, a_mu = 0, ncl = 0;
    // Everything a lane needs of the parameters sits in its registers for the whole sweep (LDS reads of the parameter
    // block inside the stage loop also make the compiler drain the fetches in flight: same LDS object as the ring).
    const int lj = lane >= 18 && lane < 30 ? lane - 18 : 0;                 // lanes 18..29: bounded component (update) / row of rb
    const bool lj_lo = bnd_lo(P, lj) > -BOUND_INF, lj_hi = bnd_hi(P, lj) < BOUND_INF;
    // lanes 0..17: stationarity row of the QP; MODE 2: lanes 30..47 hold the same rows for the NLP residual of the row ahead
    const int ci = lane < NW ? lane : (FUSE && lane >= 30 && lane < 30 + NW ? lane - 30 : 0), cls = ci / 6, cj = ci - cls * 6;
    const bool c_lo = cls < 2 && bnd_lo(P, ci < NB ? ci : 0) > -BOUND_INF, c_hi = cls < 2 && bnd_hi(P, ci < NB ? ci : 0) < BOUND_INF;
    const double cb_lo = bnd_lo(P, ci < NB ? ci : 0), cb_hi = bnd_hi(P, ci < NB ? ci : 0);
    const double k_dt = P.dt, k_2wu = 2.0 * P.w_u, k_c2 = P.w_qddot * P.cq[cj] * P.cq[cj], k_lm = P.lm;
    const double k_p1 = cls == 0 ? P.b1[cj] : P.a12[cj], k_p2 = cls == 0 ? P.b2[cj] : P.a22[cj];
    const double k_wy = P.w_task[lane >= 48 && lane < 48 + NTASK ? lane - 48 : 0];
    const double k_ra = lj < 6 ? P.a12[lj] : P.a22[lj - 6], k_rb = lj < 6 ? P.b1[lj] : P.b2[lj - 6];
    const double xh = lane < NX ? sm.xhat[lane] : 0.0;
    // MODE 2, lanes 0..11: dynamics row / cost share of the NLP residual (nlp_res_pass), x_hat of the previous QP
    const int jd = lane < 12 ? lane % 6 : 0;
    const double d_a = lane < 6 ? P.a12[jd] : P.a22[jd], d_b = lane < 6 ? P.b1[jd] : P.b2[jd], d_cq = P.cq[jd];
    const double d_wt = P.w_task[lane >= 6 && lane < 6 + NTASK ? lane - 6 : 0], k_wq = P.w_qddot;
    const double xprev = FUSE && lane < NX ? sm.vec[3][lane] : 0.0;
    double csum = 0.0, n_s = 0, n_e = 0, n_i = 0, n_c = 0;
    // NLP stationarity / bound violation / complementarity of stage kk from its landed row (multipliers as the QP left
    // them) -- nlp_res_pass lanes 16..33, here lanes 30..47; ppi = pi_{kk-1}
    const int ix_v = ci < 6 ? O_U + ci : O_X + ci - 6;      // the bounded variable of stationarity row ci
    const int ix_h = ci >= 6 ? ci - 6 : 0;                    // its entry of pi_{k-1}
    auto nlp_stat = [&](const double *row, const double *ppi, int kk) {
        // operands first, by every lane (idle lanes: row offsets of ci = 0), then the rows on registers
        const double *r1 = row, *r2 = row + I_L, *pk = row + O_QPI;
        const int j = cj;
        double uj = r1[O_U + j], vj = r1[O_X + 6 + j], pkq = pk[j], pkv = pk[6 + j], pp = ppi[ix_h];
        double h0 = r2[O_GQ + j], h1 = r2[O_GQ + 6 + j], h2 = r2[O_GQ + 12 + j], h3 = r2[O_GQ + 18 + j], h4 = r2[O_GQ + 24 + j], hv = r2[O_GV + j];
        double z0 = r2[O_Y], z1 = r2[O_Y + 1], z2 = r2[O_Y + 2], z3 = r2[O_Y + 3], z4 = r2[O_Y + 4];
        double curv = r1[ix_v], la = row[O_QLAM + ci], ta = row[O_QT + ci], lb = row[O_QLAM + 12 + ci], tb = row[O_QT + 12 + ci];
        pin(uj); pin(vj); pin(pkq); pin(pkv); pin(pp); pin(h0); pin(h1); pin(h2); pin(h3); pin(h4); pin(hv);
        pin(z0); pin(z1); pin(z2); pin(z3); pin(z4); pin(curv); pin(la); pin(ta); pin(lb); pin(tb);
        if (lane < 30 || lane >= 30 + NW) return;
        double v = 0.0;
        if (cls == 0) {
            if (kk < N) {
                v = k_dt * (k_2wu * uj + k_c2 * (uj - vj));
                v += k_p1 * pkq + k_p2 * pkv;
            }
        } else if (cls == 1) {
            if (kk > 0) {
                if (kk < N) {
                    double s_ = 0.0;
                    s_ += h0 * z0; s_ += h1 * z1; s_ += h2 * z2; s_ += h3 * z3; s_ += h4 * z4;
                    v = k_dt * s_ + pkq;
                }
                v -= pp;
            }
        } else {
            if (kk > 0) {
                if (kk < N) {
                    v = k_dt * (hv * z4 + k_c2 * (vj - uj));
                    v += k_p1 * pkq + k_p2 * pkv;
                }
                v -= pp;
            }
        }
        const bool hc = cls == 0 ? kk < N : (cls == 1 && kk >= 1 && kk < N);
        if (hc) {
            if (c_lo) {
                v -= la;
                n_i = fmax(n_i, fabs((cb_lo - curv) + ta));
                n_c = fmax(n_c, fabs(la * ta));
            }
            if (c_hi) {
                v += lb;
                n_i = fmax(n_i, fabs((curv - cb_hi) + tb));
                n_c = fmax(n_c, fabs(lb * tb));
            }
        }
        if (ci >= 6 && kk == 0) v = 0.0;
        n_s = fmax(n_s, fabs(v));
    };
    // update of one landed row: dw += a ddw ; (lam, t) += a (dlam, dt) -- or the warm-start clamp in mode 0
    // phase R operand slots: every lane fetches its role's operands up front (idle lanes: offset 0)
    //   lanes 0..17 stationarity row | 18..29 dynamics residual row | 32.. record copies, pi hand-over
    const int lc = lane >= 32 ? lane - 32 : 0;
    const int ix_a = lane < NW ? O_U + cj : (lane < 30 ? O_QW + 6 + lj : (lane >= 48 && lane < 60 ? O_QPI + (lane - 48) : 0));
    const int ix_b = lane < NW ? O_QW + cj : (lane < 30 ? O_QW + (lj < 6 ? 12 + lj : lj - 6) : 0);
    const int ix_c = lane < NW ? O_X + 6 + cj : (lane < 30 ? O_QW + (lj < 6 ? lj : 0) : 0);
    const int ix_d = lane < NW ? O_QW + 12 + cj : (lane < 30 ? I_L + O_BD + lj : 0);
    const int ix_c2 = lane >= 32 ? (lc < 7 ? O_QW / 2 + 32 + lc : (lc < 12 ? I_L / 2 + lc - 7 : 0)) : 0;   // 16-byte items
    static_assert(O_QW % 2 == 0 && I_L % 2 == 0, "copy role layout");
    constexpr bool STEP = MODE == 1;
    auto upd_row = [&](double *row, int kr) {
        if (STEP) {
            if (lane < NW) row[O_QW + lane] += a * row[I_D + lane];
        } else {
            if (kr == 0 && lane < NX) row[O_QW + 6 + lane] = xh - row[O_X + lane];
            if (!FASTM && kr == N && lane >= 12 && lane < 18) row[O_QW + lane - 12] = 0.0;
        }
        if (!FASTM && lane >= 18 && lane < 30) {
            const int j = lj;
            const bool hc = j < 6 ? kr < N : (kr >= 1 && kr < N);
            const bool blo = hc && lj_lo, bhi = hc && lj_hi;
            double *lam = row + O_QLAM, *t = row + O_QT;
            if (!STEP) {
                lam[j] = ipm::warm_lam(blo, lam[j]); t[j] = ipm::warm_t(blo, t[j]);
                lam[12 + j] = ipm::warm_lam(bhi, lam[12 + j]); t[12 + j] = ipm::warm_t(bhi, t[12 + j]);
                ncl += (blo ? 1.0 : 0.0) + (bhi ? 1.0 : 0.0);
            } else {
                const double *dl = row + I_D + 30, *dt = row + I_D + 54;
                const double l0 = lam[j], t0 = t[j], l1 = lam[12 + j], t1 = t[12 + j];
                const double d0 = dl[j], e0 = dt[j], d1 = dl[12 + j], e1 = dt[12 + j];
                // (selects, not branches: every exec-mask branch costs scalar instructions and a bubble)
                lam[j] = ipm::step_floor(blo, l0, a, d0); t[j] = ipm::step_floor(blo, t0, a, e0);
                lam[12 + j] = ipm::step_floor(bhi, l1, a, d1); t[12 + j] = ipm::step_floor(bhi, t1, a, e1);
            }
        }
    };
    if (lane < NX) sm.vec[0][lane] = 0.0;      // pi_{-1}: never read (k = 0 rows return early)
    sweep<IN_ITEMS, OUT_NI, false>(bin, N, lane, [&](int k, double *cur, double *nxt, double *) {
        double *o = sm.out[k & 1];
#ifdef MPCB_NOCOMPUTE
        (void)cur; (void)nxt; store_out(bout, o, lane); return;
#endif
        if (FUSE) {
            // NLP residual of the previous step's iterate, from the rows as they landed (program order: these LDS reads
            // are issued before the warm-start writes below)
            if (k == 0) nlp_stat(cur, cur, 0);
            // (the defect lanes' operands go out with the stationarity lanes': one round trip for both)
            const int l12 = lane < 12 ? lane : 0;
            double b_xq = cur[O_X + jd], b_xv = cur[O_X + 6 + jd], b_u = cur[O_U + jd], b_xn = nxt[O_X + l12], b_x0 = cur[O_X + l12];
            double b_r = cur[I_L + O_R + (lane >= 6 && lane < 6 + NTASK ? lane - 6 : 0)];
            if (k + 1 <= N) nlp_stat(nxt, cur + O_QPI, k + 1);
            pin(b_xq); pin(b_xv); pin(b_u); pin(b_xn); pin(b_x0); pin(b_r);
            if (lane < 12) {
                // dynamics defect (prediction_model.py:317-320) and this stage's share of the cost
                double v = 0.0;
                if (k < N) {
                    v = lane < 6 ? (b_xq + d_a * b_xv + d_b * b_u) - b_xn : (d_a * b_xv + d_b * b_u) - b_xn;
                    n_e = fmax(n_e, fabs(v));
                    if (lane < 6) {
                        const double qdd = d_cq * (b_u - b_xv);
                        csum += 0.5 * k_dt * (k_2wu * b_u * b_u + k_wq * qdd * qdd);
                    } else if (lane < 6 + NTASK) {
                        csum += 0.5 * k_dt * d_wt * b_r * b_r;
                    }
                }
                if (k == 0) n_i = fmax(n_i, fabs(xprev - b_x0));   // lbx_0 = ubx_0 = x_hat of that QP
                cur[I_L + O_BD + lane] = v;          // the QP's dynamics residual rb of this stage reads it below
                o[O_RY + O_BD + lane] = v;
            } else if (lane < 14) {
                o[O_RY + 10 + (lane - 12)] = 0.0;    // padding scalars of the record
            }
            fence();
        }
        if (k == 0) { upd_row(cur, 0); fence(); }
        // ---- U: update the lookahead row; pi_k += a dpi (stored with stage k+1)
        //      Y (lanes 48..52): y_i = w_i (r_i + G_i . delta_k) -- dw_k was updated when row k was the lookahead row
        // operands of every role first (see phase R), then the role blocks on registers
        const int iy = lane >= 48 && lane < 48 + NTASK ? lane - 48 : 0;
        double yr = cur[I_L + O_R + iy];
        double gq0 = cur[I_L + O_GQ + iy * 6], gq1 = cur[I_L + O_GQ + iy * 6 + 1], gq2 = cur[I_L + O_GQ + iy * 6 + 2],
               gq3 = cur[I_L + O_GQ + iy * 6 + 3], gq4 = cur[I_L + O_GQ + iy * 6 + 4], gq5 = cur[I_L + O_GQ + iy * 6 + 5];
        double wq0 = cur[O_QW + 6], wq1 = cur[O_QW + 7], wq2 = cur[O_QW + 8], wq3 = cur[O_QW + 9], wq4 = cur[O_QW + 10], wq5 = cur[O_QW + 11];
        double wv0 = cur[O_QW + 12], wv1 = cur[O_QW + 13], wv2 = cur[O_QW + 14], wv3 = cur[O_QW + 15], wv4 = cur[O_QW + 16], wv5 = cur[O_QW + 17];
        double gv0 = cur[I_L + O_GV], gv1 = cur[I_L + O_GV + 1], gv2 = cur[I_L + O_GV + 2], gv3 = cur[I_L + O_GV + 3],
               gv4 = cur[I_L + O_GV + 4], gv5 = cur[I_L + O_GV + 5];
        if (STEP) {
            const int lw = lane < NW ? lane : 0, lp = lane >= 32 && lane < 44 ? lane - 32 : 0;
            double u1 = nxt[O_QW + lw], u2 = nxt[lane < NW ? I_D + lane : I_D + 18 + lp], p1 = cur[O_QPI + lp];
            double l0 = nxt[O_QLAM + lj], t0 = nxt[O_QT + lj], l1 = nxt[O_QLAM + 12 + lj], t1 = nxt[O_QT + 12 + lj];
            double d0 = nxt[I_D + 30 + lj], e0 = nxt[I_D + 54 + lj], d1 = nxt[I_D + 42 + lj], e1 = nxt[I_D + 66 + lj];
            pin(u1); pin(u2); pin(p1); pin(l0); pin(t0); pin(l1); pin(t1); pin(d0); pin(e0); pin(d1); pin(e1);
            pin(yr); pin(gq0); pin(gq1); pin(gq2); pin(gq3); pin(gq4); pin(gq5);
            pin(wq0); pin(wq1); pin(wq2); pin(wq3); pin(wq4); pin(wq5); pin(wv0); pin(wv1); pin(wv2); pin(wv3); pin(wv4); pin(wv5);
            pin(gv0); pin(gv1); pin(gv2); pin(gv3); pin(gv4); pin(gv5);
            if (k + 1 <= N) {
                // (upd_row on registers)
                if (lane < NW) {
                    nxt[O_QW + lane] = u1 + a * u2;
                } else if (lane < 30) {
                    const int j = lj, kr = k + 1;
                    const bool hc = j < 6 ? kr < N : (kr >= 1 && kr < N);
                    const bool blo = hc && lj_lo, bhi = hc && lj_hi;
                    double *lam = nxt + O_QLAM, *t = nxt + O_QT;
                    lam[j] = ipm::step_floor(blo, l0, a, d0); t[j] = ipm::step_floor(blo, t0, a, e0);
                    lam[12 + j] = ipm::step_floor(bhi, l1, a, d1); t[12 + j] = ipm::step_floor(bhi, t1, a, e1);
                } else if (lane >= 32 && lane < 44) {
                    cur[O_QPI + lane - 32] = p1 + a * u2;
                }
            }
        } else {
            pin(yr); pin(gq0); pin(gq1); pin(gq2); pin(gq3); pin(gq4); pin(gq5);
            pin(wq0); pin(wq1); pin(wq2); pin(wq3); pin(wq4); pin(wq5); pin(wv0); pin(wv1); pin(wv2); pin(wv3); pin(wv4); pin(wv5);
            pin(gv0); pin(gv1); pin(gv2); pin(gv3); pin(gv4); pin(gv5);
            double l0 = nxt[O_QLAM + lj], t0 = nxt[O_QT + lj], l1 = nxt[O_QLAM + 12 + lj], t1 = nxt[O_QT + 12 + lj];
            pin(l0); pin(t0); pin(l1); pin(t1);
            if (FASTM && k > 0) {   // dw = 0 beyond the embedded x_0
                wq0 = wq1 = wq2 = wq3 = wq4 = wq5 = 0.0; wv0 = wv1 = wv2 = wv3 = wv4 = wv5 = 0.0;
            }
            if (!FASTM && k + 1 <= N) {
                // (upd_row of the lookahead row on registers: warm-start clamp; the row is never row 0)
                const int kr = k + 1;
                if (kr == N && lane >= 12 && lane < 18) nxt[O_QW + lane - 12] = 0.0;
                if (lane >= 18 && lane < 30) {
                    const int j = lj;
                    const bool hc = j < 6 ? kr < N : (kr >= 1 && kr < N);
                    const bool blo = hc && lj_lo, bhi = hc && lj_hi;
                    double *lam = nxt + O_QLAM, *t = nxt + O_QT;
                    lam[j] = ipm::warm_lam(blo, l0); t[j] = ipm::warm_t(blo, t0);
                    lam[12 + j] = ipm::warm_lam(bhi, l1); t[12 + j] = ipm::warm_t(bhi, t1);
                    ncl += (blo ? 1.0 : 0.0) + (bhi ? 1.0 : 0.0);
                }
            }
        }
        if (lane >= 48 && lane < 48 + NTASK && k < N) {
            const int i = lane - 48;
            double v = yr;
            v += gq0 * wq0; v += gq1 * wq1; v += gq2 * wq2; v += gq3 * wq3; v += gq4 * wq4; v += gq5 * wq5;
            if (i == 4) { v += gv0 * wv0; v += gv1 * wv1; v += gv2 * wv2; v += gv3 * wv3; v += gv4 * wv4; v += gv5 * wv5; }
            cur[I_L + O_Y + i] = k_wy * v;
        }
        fence();
        // ---- R: residuals, Gamma, gt (lanes 0..17), dynamics residual (lanes 18..29), copies (lanes 32..)
        {
            const double *pm = sm.vec[k & 1];
            // operands first (see ix_*), one LDS round trip for the whole phase
            double q_a = cur[ix_a], q_b = cur[ix_b], q_c = cur[ix_c], q_d = cur[ix_d];
            double q_e = cur[O_QPI + cj], q_f = cur[O_QPI + 6 + cj], q_g = cur[O_QW + ci], q_v = cur[ix_v];
            double l_lo = cur[O_QLAM + ci], t_lo = cur[O_QT + ci], l_hi = cur[O_QLAM + 12 + ci], t_hi = cur[O_QT + 12 + ci];
            double g0 = cur[I_L + O_GQ + cj], g1 = cur[I_L + O_GQ + 6 + cj], g2 = cur[I_L + O_GQ + 12 + cj], g3 = cur[I_L + O_GQ + 18 + cj],
                   g4 = cur[I_L + O_GQ + 24 + cj], gv = cur[I_L + O_GV + cj];
            double y0 = cur[I_L + O_Y], y1 = cur[I_L + O_Y + 1], y2 = cur[I_L + O_Y + 2], y3 = cur[I_L + O_Y + 3], y4 = cur[I_L + O_Y + 4];
            double q_h = pm[ix_h], q_n = nxt[O_QW + 6 + lj];
            D2 c1 = ((MPC_LOCAL D2 *)(cur + O_QW))[lc], c2 = ((MPC_LOCAL D2 *)cur)[ix_c2];
            pin(q_a); pin(q_b); pin(q_c); pin(q_d); pin(q_e); pin(q_f); pin(q_g); pin(q_v);
            pin(l_lo); pin(t_lo); pin(l_hi); pin(t_hi);
            pin(g0); pin(g1); pin(g2); pin(g3); pin(g4); pin(gv); pin(y0); pin(y1); pin(y2); pin(y3); pin(y4);
            pin(q_h); pin(q_n); pin(c1); pin(c2);
            if (FASTM) {
                // (dw, pi, lam) = 0: only the embedded x_0 part of stage 0 survives
                if (lane < NW) {
                    q_b = 0.0; q_e = 0.0; q_f = 0.0; q_h = 0.0;
                    q_d = k == 0 ? q_d : 0.0; q_g = 0.0;
                } else if (lane < 30) {
                    // dynamics rows: q_a = dx_k[lj], q_b = dv_lj (lj < 6) | du_(lj-6), q_c = du_lj, q_n = dx_{k+1}[lj]
                    q_a = k == 0 ? q_a : 0.0;
                    q_b = k == 0 && lj < 6 ? q_b : 0.0;
                    q_c = 0.0; q_n = 0.0;
                }
            }
            if (lane < NW) {
                // stationarity of the QP at w + dw (mpc_core.h stat_cls with `with_delta`), same operation order
                double rg = 0.0;
                if (cls == 0) {
                    if (k < N) {
                        const double uj = q_a + q_b, vj = q_c + q_d;
                        rg = k_dt * (k_2wu * uj + k_c2 * (uj - vj));
                        rg += k_p1 * q_e + k_p2 * q_f;
                        rg += k_dt * k_lm * q_b;
                    }
                } else if (cls == 1) {
                    if (k > 0) {
                        if (k < N) {
                            double s = 0.0;
                            s += g0 * y0; s += g1 * y1; s += g2 * y2; s += g3 * y3; s += g4 * y4;
                            rg = k_dt * s + q_e;
                        }
                        rg += (k < N ? k_dt : 1.0) * k_lm * q_g;
                        rg -= q_h;
                    }
                } else {
                    if (k > 0) {
                        if (k < N) {
                            const double uj = q_a + q_b, vj = q_c + q_d;
                            rg = k_dt * (gv * y4 + k_c2 * (vj - uj));
                            rg += k_p1 * q_e + k_p2 * q_f;
                        }
                        rg += (k < N ? k_dt : 1.0) * k_lm * q_d;
                        rg -= q_h;
                    }
                }
                double gt = rg;
                if (cls < 2) {
                    const bool hc = cls == 0 ? k < N : (k >= 1 && k < N);
                    const bool blo = !FASTM && hc && c_lo, bhi = !FASTM && hc && c_hi;
                    const double v = hc ? q_v : 0.0, dv = q_g;
                    // both sides are evaluated in every lane and masked by selects (an absent bound holds lam = 0, t = 1:
                    // its terms are exact zeros or are deselected), in the operation order of the branchy form
                    const double it_lo = fast_rcp(t_lo), it_hi = fast_rcp(t_hi);
                    const double rdl = blo ? dv - (cb_lo - v) - t_lo : 0.0, rml = blo ? l_lo * t_lo : 0.0;
                    const double rdu = bhi ? (cb_hi - v) - dv - t_hi : 0.0, rmu = bhi ? l_hi * t_hi : 0.0;
                    double gam = 0.0;
                    rg = blo ? rg - l_lo : rg;
                    gt = blo ? gt - l_lo : gt;
                    gam = blo ? gam + l_lo * it_lo : gam;
                    gt = blo ? gt + (rml + l_lo * rdl) * it_lo : gt;
                    a_mu = blo ? a_mu + rml : a_mu;
                    a_d = fmax(a_d, fabs(rdl)); a_m = fmax(a_m, fabs(rml));
                    rg = bhi ? rg + l_hi : rg;
                    gt = bhi ? gt + l_hi : gt;
                    gam = bhi ? gam + l_hi * it_hi : gam;
                    gt = bhi ? gt - (rmu + l_hi * rdu) * it_hi : gt;
                    a_mu = bhi ? a_mu + rmu : a_mu;
                    a_d = fmax(a_d, fabs(rdu)); a_m = fmax(a_m, fabs(rmu));
                    if (!FASTM) {
                        o[O_3 + O_RD + ci] = rdl; o[O_3 + O_RD + 12 + ci] = rdu;
                        o[O_3 + O_RM + ci] = rml; o[O_3 + O_RM + 12 + ci] = rmu;
                    }
                    o[O_G + ci] = gam;
                }
                if (!FASTM) o[O_3 + O_RG + ci] = rg;
                o[O_G + 12 + ci] = gt;
                a_g = fmax(a_g, fabs(rg));
            } else if (lane < 30) {
                const int i = lj;
                double v = 0.0;
                if (k < N) {
                    if (i < 6) v = q_a + k_ra * q_b + k_rb * q_c;
                    else v = k_ra * q_a + k_rb * q_b;
                    v += q_d - q_n;
                    a_b = fmax(a_b, fabs(v));
                }
                o[O_G + 30 + i] = v;
            } else if (lane >= 32) {
                // the updated QW..QT of this stage and r, y go out with the residual records (39 + 5 items, two per lane
                // where needed; MODE 2: BD was put there by its lanes); lanes 48..59: pi_k for the next stage
                if (!FASTM) {
                    ((MPC_LOCAL D2 *)(o + O_W))[lc] = c1;
                    if (lc < 7) ((MPC_LOCAL D2 *)(o + O_W))[32 + lc] = c2;
                }
                if (lc >= 7 && lc < 12) ((MPC_LOCAL D2 *)(o + O_RY))[lc - 7] = c2;
                if (lane >= 48 && lane < 60) sm.vec[(k + 1) & 1][lane - 48] = q_a;
            }
        }
        fence();
        store_out(bout, o, lane);
    });
    IpmNorms r;
    r.ng = wmax(a_g); r.nb = wmax(a_b); r.nd = wmax(a_d); r.nm = wmax(a_m); r.smu = wsum(a_mu); r.nc = wsum(ncl);
    if (FUSE) {
        nlp_out[0] = wsum(csum);
        nlp_out[1] = wmax(n_s); nlp_out[2] = wmax(n_e); nlp_out[3] = wmax(n_i); nlp_out[4] = wmax(n_c);
    }
    return r;
}

// MODE 1 of the residual pass (the pass of every interior-point iteration but the first), ITEM-parallel: the residual pass has
// no recursion in it -- every element is a function of its own stage's records and one row of each neighbour -- so instead of
// walking the stages with a handful of lanes busy per role (residual_pass<1>: ~300 instructions per stage), the 64 lanes take
// 64 ITEMS at a time, operands straight from the stage records in HBM / L2 into registers (mpc_core.h residual_direct, the same
// three phases U | Y | S,D and the same arithmetic per element).  ~17 batches of loads per pass instead of 101 stages of chain;
// while a batch is in flight the SIMD's other simulation runs.  y of every stage waits in the (idle) ring between Y and S.
SE_DEV bool residual_items_ok(int N) { return (N + 1) * 6 <= RING_DOUBLES; }
SE_PASS IpmNorms residual_items(double a)
{
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor), NS = N + 1;
    const SWs w = sm.w;
    const int LD = uni(w.ld >> 3);
    // ONE record per stage, [G1 | G2 | G3 | G4]: every field of a stage is a constant offset from the stage's base, so an item
    // forms one address (base + stage * LD + its lane-dependent column) and all its loads and stores use immediate offsets
    // (uniform 64-bit base + 32-bit byte offset in a register + immediate: the addressing mode of global_load / global_store)
    MPC_GLOBAL char *const gb = (MPC_GLOBAL char *)(((unsigned long long)(unsigned)uni((int)((unsigned long long)w.G1 >> 32)) << 32) |
                                                    (unsigned)uni((int)(unsigned long long)w.G1));
    auto rec = [&](int k, int col) { return (MPC_GLOBAL double *)(gb + (unsigned)((k * LD + col) << 3)); };
    constexpr int C2 = W1, C3 = W1 + W2;                      // G2, G3 columns inside the record
    double *const Y = sm.ring;                               // [NS][6], 5 used
    auto gld = [](const MPC_GLOBAL double *p) { return *p; };
    auto gst = [](MPC_GLOBAL double *p, double v) { *p = v; };
    double a_g = 0, a_b = 0, a_d = 0, a_m = 0, a_mu = 0;
    wait_vm<0>();                                            // no fetch of the sweep before is still landing in the ring
    fence();
    // ---------------------------------------------------------------- U: 16-byte items
    {
        // dw (18) | pi (12) += a * step: G1 [QW, QW + 30) <- G3 [DW, DW + 30); the step of multiplier k sits with stage k + 1
        constexpr int IPS = 15, R = 8;
        const int items = NS * IPS;
        for (int base = 0; base < items; base += R * WAVE) {
            D2 cur[R], stp[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = imin(base + r * WAVE + lane, items - 1), k = e / IPS, c = 2 * (e - k * IPS);
                cur[r] = *(MPC_GLOBAL const D2 *)(rec(k, c) + O_QW);
                stp[r] = *(MPC_GLOBAL const D2 *)(rec(c >= 18 ? imin(k + 1, N) : k, c) + C3 + O_DW);
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = base + r * WAVE + lane;
                if (e < items) {
                    const int k = e / IPS, c = 2 * (e - k * IPS);
                    const double aa = (c >= 18 && k >= N) ? 0.0 : a;   // no multiplier beyond the last dynamics
                    D2 v = cur[r];
                    v.x += aa * stp[r].x; v.y += aa * stp[r].y;
                    *(MPC_GLOBAL D2 *)(rec(k, c) + O_QW) = v;
                }
            }
        }
    }
    {
        // lam (24) | t (24): G1 [QLAM, QLAM + 48) <- G3 [DLAM, DLAM + 48)
        constexpr int IPS = 24, R = 8;
        const int items = NS * IPS;
        for (int base = 0; base < items; base += R * WAVE) {
            D2 cur[R], stp[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = imin(base + r * WAVE + lane, items - 1), k = e / IPS, q = 2 * (e - k * IPS);
                cur[r] = *(MPC_GLOBAL const D2 *)(rec(k, q) + O_QLAM);
                stp[r] = *(MPC_GLOBAL const D2 *)(rec(k, q) + C3 + O_DLAM);
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = base + r * WAVE + lane;
                if (e < items) {
                    const int k = e / IPS, q = 2 * (e - k * IPS);
                    const int sc = q >= 24 ? q - 24 : q, j = sc < 12 ? sc : sc - 12;        // side/component, component
                    const bool ok = j < 6 ? k < N : (k >= 1 && k < N);                      // has_comp (same for j and j + 1)
                    // which bound sides exist (|bound| >= 1e29 means absent, include/mpcbatch.h); sc < 12: lower, else upper
                    const bool on0 = ok && (sc < 12 ? bnd_lo(P, j) > -BOUND_INF : bnd_hi(P, j) < BOUND_INF);
                    const bool on1 = ok && (sc < 12 ? bnd_lo(P, j + 1) > -BOUND_INF : bnd_hi(P, j + 1) < BOUND_INF);
                    D2 v = cur[r];
                    v.x = ipm::step_floor(on0, v.x, a, stp[r].x); v.y = ipm::step_floor(on1, v.y, a, stp[r].y);
                    *(MPC_GLOBAL D2 *)(rec(k, q) + O_QLAM) = v;
                }
            }
        }
    }
    wait_vm<0>();
    fence();
    // ---------------------------------------------------------------- Y: items (k < N, i < 5)
    {
        constexpr int R = 3;
        const int items = N * NTASK;
        for (int base = 0; base < items; base += R * WAVE) {
            double g[R][12], d[R][12], rr[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = imin(base + r * WAVE + lane, items - 1), k = e / NTASK, i = e - k * NTASK;
                const MPC_GLOBAL double *g1 = rec(k, 0) + O_QW, *g2 = rec(k, i) + C2, *gq = rec(k, i * 6) + C2;
                rr[r] = gld(g2 + O_R);
#pragma unroll
                for (int j = 0; j < 6; j++) { g[r][j] = gld(gq + O_GQ + j); d[r][j] = gld(g1 + 6 + j); }
                // (row 4 alone has a velocity part; the other rows fetch it too -- all loads of the batch in flight before any branch)
#pragma unroll
                for (int j = 0; j < 6; j++) { g[r][6 + j] = gld(g1 - O_QW + C2 + O_GV + j); d[r][6 + j] = gld(g1 + 12 + j); }
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = base + r * WAVE + lane;
                if (e < items) {
                    const int k = e / NTASK, i = e - k * NTASK;
                    double v = rr[r];
#pragma unroll
                    for (int j = 0; j < 6; j++) v += g[r][j] * d[r][j];
                    if (i == 4) {
#pragma unroll
                        for (int j = 0; j < 6; j++) v += g[r][6 + j] * d[r][6 + j];
                    }
                    v *= P.w_task[i];
                    Y[k * 6 + i] = v;
                    gst(rec(k, i) + C2 + O_Y, v);
                }
            }
        }
    }
    wait_vm<0>();
    fence();
    // ---------------------------------------------------------------- S: joint items (k, j < 6): u_j, v_j, q_j ; D: dynamics items
    {
        constexpr int R = 2, RD = 2 * R;
        const int items = NS * 6, items_d = NS * NX;
        const double dt = P.dt, lm = P.lm;
        // bound part of a bounded component ci: returns gt, updates rg, writes rd | rm | Gamma (an absent side holds lam = 0, t = 1)
        auto bounds = [&](int k, int ci, double val_, double dv, double l_lo, double l_hi, double t_lo, double t_hi, double &rg) {
            const bool hc = ci < 6 ? k < N : (k >= 1 && k < N);
            const double b_lo = bnd_lo(P, ci), b_hi = bnd_hi(P, ci);
            const bool blo = hc && b_lo > -BOUND_INF, bhi = hc && b_hi < BOUND_INF;
            const double val = hc ? val_ : 0.0;
            const double ll = blo ? l_lo : 0.0, lu = bhi ? l_hi : 0.0;
            const double itl = fast_rcp(t_lo), itu = fast_rcp(t_hi);
            const double rdl = blo ? dv - (b_lo - val) - t_lo : 0.0;
            const double rdu = bhi ? (b_hi - val) - dv - t_hi : 0.0;
            const double rml = ll * t_lo, rmu = lu * t_hi;
            double gt = rg;
            rg -= ll; gt -= ll;
            double gam = ll * itl;
            gt += (rml + ll * rdl) * itl;
            rg += lu; gt += lu;
            gam += lu * itu;
            gt -= (rmu + lu * rdu) * itu;
            a_mu += rml + rmu;
            a_d = fmax(a_d, fmax(fabs(rdl), fabs(rdu)));
            a_m = fmax(a_m, fmax(fabs(rml), fabs(rmu)));
            MPC_GLOBAL double *rc = rec(k, ci);
            gst(rc + C3 + O_RD, rdl); gst(rc + C3 + O_RD + 12, rdu);
            gst(rc + C3 + O_RM, rml); gst(rc + C3 + O_RM + 12, rmu);
            gst(rc + C2 + O_GAM, gam);
            return gt;
        };
        for (int base = 0; base < items; base += R * WAVE) {
            double v[R][12], q[R][13], d[RD][5];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = imin(base + r * WAVE + lane, items - 1), k = e / 6, j = e - k * 6, km = imax(k - 1, 0);
                const MPC_GLOBAL double *g1 = rec(k, j), *g2 = g1 + C2, *gm = rec(km, j);
                v[r][0] = gld(g1 + O_U);      v[r][1] = gld(g1 + O_X + 6);
                v[r][2] = gld(g1 + O_QW);     v[r][3] = gld(g1 + O_QW + 12);
                v[r][4] = gld(g1 + O_QPI);    v[r][5] = gld(g1 + O_QPI + 6);
                v[r][6] = gld(gm + O_QPI + 6);
                v[r][7] = gld(g2 + O_GV);
                v[r][8] = gld(g1 + O_QLAM);   v[r][9] = gld(g1 + O_QLAM + 12);
                v[r][10] = gld(g1 + O_QT);    v[r][11] = gld(g1 + O_QT + 12);
                q[r][0] = gld(g1 + O_X);  q[r][1] = gld(g1 + O_QW + 6);
#pragma unroll
                for (int i = 0; i < NTASK; i++) q[r][2 + i] = gld(g2 + O_GQ + i * 6);
                q[r][7] = v[r][4]; q[r][8] = gld(gm + O_QPI);
                q[r][9] = gld(g1 + O_QLAM + 6);  q[r][10] = gld(g1 + O_QLAM + 18);
                q[r][11] = gld(g1 + O_QT + 6);   q[r][12] = gld(g1 + O_QT + 18);
            }
#pragma unroll
            for (int r = 0; r < RD; r++) {
                const int e = imin(2 * base + r * WAVE + lane, items_d - 1), k = e / NX, i = e - k * NX, kn = imin(k + 1, N);
                const MPC_GLOBAL double *ri = rec(k, i), *r6 = rec(k, i < 6 ? i : i - 6);
                d[r][0] = gld(ri + O_QW + 6);
                d[r][1] = gld(i < 6 ? r6 + O_QW + 12 : r6 + O_QW);     // dv_i | du_(i-6)
                d[r][2] = gld(r6 + O_QW);
                d[r][3] = gld(ri + C2 + O_BD);
                d[r][4] = gld(rec(kn, i) + O_QW + 6);
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = base + r * WAVE + lane;
                if (e < items) {
                    const int k = e / 6, j = e - k * 6;
                    const double du = v[r][2], dvv = v[r][3];
                    const double uj = v[r][0] + du, vj = v[r][1] + dvv;
                    const double c2 = P.w_qddot * P.cq[j] * P.cq[j];
                    MPC_GLOBAL double *g3 = rec(k, j) + C3, *g2 = rec(k, j) + C2;
                    // u_j
                    double rgu = 0.0;
                    if (k < N) {
                        rgu = dt * (2.0 * P.w_u * uj + c2 * (uj - vj));
                        rgu += P.b1[j] * v[r][4] + P.b2[j] * v[r][5];
                        rgu += dt * lm * du;
                    }
                    const double gtu = bounds(k, j, v[r][0], du, v[r][8], v[r][9], v[r][10], v[r][11], rgu);
                    gst(g3 + O_RG, rgu); gst(g2 + O_GT, gtu);
                    // v_j: no bounds
                    double rgv = 0.0;
                    if (k >= 1) {
                        if (k < N) {
                            rgv = dt * (v[r][7] * Y[k * 6 + 4] + c2 * (vj - uj));
                            rgv += P.a12[j] * v[r][4] + P.a22[j] * v[r][5];
                        }
                        rgv += (k < N ? dt : 1.0) * lm * dvv;
                        rgv -= v[r][6];
                    }
                    gst(g3 + O_RG + 12, rgv); gst(g2 + O_GT + 12, rgv);
                    // q_j; pi_k[j] is v[r][4]
                    const double dq = q[r][1];
                    double rg = 0.0;
                    if (k >= 1) {
                        if (k < N) {
                            double s_ = 0.0;
#pragma unroll
                            for (int i = 0; i < NTASK; i++) s_ += q[r][2 + i] * Y[k * 6 + i];
                            rg = dt * s_ + q[r][7];
                        }
                        rg += (k < N ? dt : 1.0) * lm * dq;
                        rg -= q[r][8];
                    }
                    const double gt = bounds(k, 6 + j, q[r][0], dq, q[r][9], q[r][10], q[r][11], q[r][12], rg);
                    gst(g3 + O_RG + 6, rg); gst(g2 + O_GT + 6, gt);
                    a_g = fmax(a_g, fmax(fmax(fabs(rgu), fabs(rgv)), fabs(rg)));
                }
            }
#pragma unroll
            for (int r = 0; r < RD; r++) {
                const int e = 2 * base + r * WAVE + lane;
                if (e < items_d && e < 2 * (base + R * WAVE)) {
                    const int k = e / NX, i = e - k * NX;
                    double vv = 0.0;
                    if (k < N) {
                        if (i < 6) vv = d[r][0] + P.a12[i] * d[r][1] + P.b1[i] * d[r][2];
                        else vv = P.a22[i - 6] * d[r][0] + P.b2[i - 6] * d[r][1];
                        vv += d[r][3] - d[r][4];
                        a_b = fmax(a_b, fabs(vv));
                    }
                    gst(rec(k, i) + C2 + O_RB, vv);
                }
            }
        }
    }
    IpmNorms r;
    r.ng = wmax(a_g); r.nb = wmax(a_b); r.nd = wmax(a_d); r.nm = wmax(a_m); r.smu = wsum(a_mu); r.nc = 0.0;
    return r;
}

// SQP_RTI, the first pass of a step, ITEM-parallel (round 4): what residual_pass<3 | 4> and nlp_res_pass<false> did stage by stage with
// 10-20 lanes busy per role (~400 instructions per stage: 154 us of a 586 us fast-path step at batch 4096) has no recursion in it -- every
// element is a function of its own stage's records and one row of each neighbour -- so the 64 lanes take 64 JOINT ITEMS (k, j < 6) at a
// time, operands straight from the stage records into registers (residual_items above; mpc_core.h nlp_direct's joint items):
//   NORMS  acados get_residuals / get_cost of the iterate lin_pass has just linearised (nlp_res_pass<false>): dynamics defect BD -> G2,
//          cost = sum_k dt/2 r'Wr, inf-norms [stat, eq, ineq, comp] with the QP's multipliers as the last solve left them;
//          out5 = [cost, stat, eq, ineq, comp]; `xsel`: x_hat that solve was made for (0: sm.xhat, 1: sm.vec[3])
//   RHS    right-hand side of the bound-inactive fast path for the NEW x_hat (residual_pass<3>; mpc_core.h fast_rhs): Gamma = 0,
//          gt = g at (dw, pi, lam) = 0 with dx_0 = x_hat - x_0 embedded, rb = b + A dx_0; stage 0's y = W (r + G dx_0)
// ONE function for every place the norms are formed (fused with the right-hand side, alone at the end of a launch / work item, before
// an interior-point solve), so a run cut into work items reproduces the plain launch bit for bit.
// SQPM (full SQP): the norms use the blended NLP multipliers of G5 (NPI | NLAM | NT: the same relative layout as QPI | QLAM | QT of G1).
// `slot` (SQP_RTI): where the QP iterate, hence its multipliers, sits in the stage record -- 0: G1 [QPI | QLAM | QT], 1: G3 [DPI | DLAM | DT]
// (an accepted fast-path candidate that has become the iterate by a flip of the index, fast_commit / ipm_solve).
template <bool NORMS, bool RHS, bool SQPM = false>
SE_PASS void rti_items(int xsel, double *out5, int slot = 0)
{
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor), NS = N + 1;
    const SWs w = sm.w;
    const int LD = uni(w.ld >> 3);
    MPC_GLOBAL char *const gb = (MPC_GLOBAL char *)(((unsigned long long)(unsigned)uni((int)((unsigned long long)w.G1 >> 32)) << 32) |
                                                    (unsigned)uni((int)(unsigned long long)w.G1));
    auto rec = [&](int k, int col) { return (MPC_GLOBAL double *)(gb + (unsigned)((k * LD + col) << 3)); };
    constexpr int C2 = W1;
    const int CM = SQPM ? uni((int)(w.G5 - w.G1)) + O_NPI : (uni(slot) ? W1 + W2 + O_DPI : O_QPI);     // column of the multipliers [pi 12 | lam 24 | t 24] in the stage record
    auto gld = [](const MPC_GLOBAL double *p) { return *p; };
    auto gst = [](MPC_GLOBAL double *p, double v) { *p = v; };
    // stage 0's joint items are lanes 0..5 of the first batch: their x_hat entries wait in registers (an LDS read inside the item loop
    // would make the compiler drain the loads in flight)
    const int l6 = lane < 6 ? lane : 0;
    const double xh_q = sm.xhat[l6], xh_v = sm.xhat[6 + l6];
    const double xp_q = xsel ? sm.vec[3][l6] : xh_q, xp_v = xsel ? sm.vec[3][6 + l6] : xh_v;
    wait_vm<0>();
    fence();
    double csum = 0.0, n_s = 0, n_e = 0, n_i = 0, n_c = 0;
    constexpr int R = MPCB_RTI_R;
    const int items = NS * 6;
    const double dt = P.dt;
    for (int base = 0; base < items; base += R * WAVE) {
        double x[R][3], m[R][14], g[R][12], bd[R][2];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int e = imin(base + r * WAVE + lane, items - 1), k = e / 6, j = e - k * 6, km = imax(k - 1, 0), kn = imin(k + 1, N);
            const MPC_GLOBAL double *g1 = rec(k, j), *g2 = g1 + C2, *y = rec(k, 0) + C2 + O_Y;
            x[r][0] = gld(g1 + O_U); x[r][1] = gld(g1 + O_X); x[r][2] = gld(g1 + O_X + 6);
            g[r][0] = gld(g2 + O_GV);
#pragma unroll
            for (int i = 0; i < NTASK; i++) { g[r][1 + i] = gld(g2 + O_GQ + i * 6); g[r][6 + i] = gld(y + i); }
            if (NORMS) {
                const MPC_GLOBAL double *gn = rec(kn, j), *mk = rec(k, j + CM), *mm = rec(km, j + CM);
                m[r][0] = gld(mk); m[r][1] = gld(mk + 6); m[r][2] = gld(mm); m[r][3] = gld(mm + 6);
                m[r][4] = gld(mk + 12); m[r][5] = gld(mk + 24); m[r][6] = gld(mk + 36);  m[r][7] = gld(mk + 48);    // u_j: lam lo, hi, t lo, hi
                m[r][8] = gld(mk + 18); m[r][9] = gld(mk + 30); m[r][10] = gld(mk + 42); m[r][11] = gld(mk + 54);   // q_j
                m[r][12] = gld(gn + O_X); m[r][13] = gld(gn + O_X + 6);
                g[r][11] = gld(rec(k, j < NTASK ? j : 0) + C2 + O_R);
            } else {
                bd[r][0] = gld(g2 + O_BD); bd[r][1] = gld(g2 + O_BD + 6);
            }
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int e = base + r * WAVE + lane;
            if (e < items) {
                const int k = e / 6, j = e - k * 6;
                const double uj = x[r][0], qj = x[r][1], vj = x[r][2];
                const double c2 = P.w_qddot * P.cq[j] * P.cq[j];
                const bool st = k < N;
                double s_ = 0.0;
#pragma unroll
                for (int i = 0; i < NTASK; i++) s_ += g[r][1 + i] * g[r][6 + i];
                double bdq, bdv;
                MPC_GLOBAL double *g2 = rec(k, j) + C2;
                if (NORMS) {
                    // dynamics defect (prediction_model.py:317-320) and this joint's share of the cost
                    bdq = 0.0; bdv = 0.0;
                    if (st) {
                        bdq = (qj + P.a12[j] * vj + P.b1[j] * uj) - m[r][12];
                        bdv = (P.a22[j] * vj + P.b2[j] * uj) - m[r][13];
                        n_e = fmax(n_e, fmax(fabs(bdq), fabs(bdv)));
                        const double qdd = P.cq[j] * (uj - vj);
                        csum += 0.5 * dt * (2.0 * P.w_u * uj * uj + P.w_qddot * qdd * qdd);
                        if (j < NTASK) csum += 0.5 * dt * P.w_task[j] * g[r][11] * g[r][11];
                    }
                    gst(g2 + O_BD, bdq); gst(g2 + O_BD + 6, bdv);
                    // stationarity of the NLP at the iterate, rows u_j, q_j, v_j (mpc_core.h stat_cls without delta), bounds, complementarity
                    auto bound = [&](int ci, bool hc, double cur, double l_lo, double l_hi, double t_lo, double t_hi, double &val) {
                        if (hc) {
                            if (bnd_lo(P, ci) > -BOUND_INF) {
                                val -= l_lo;
                                n_i = fmax(n_i, fabs((bnd_lo(P, ci) - cur) + t_lo));
                                n_c = fmax(n_c, fabs(l_lo * t_lo));
                            }
                            if (bnd_hi(P, ci) < BOUND_INF) {
                                val += l_hi;
                                n_i = fmax(n_i, fabs((cur - bnd_hi(P, ci)) + t_hi));
                                n_c = fmax(n_c, fabs(l_hi * t_hi));
                            }
                        }
                    };
                    double ru = 0.0;
                    if (st) {
                        ru = dt * (2.0 * P.w_u * uj + c2 * (uj - vj));
                        ru += P.b1[j] * m[r][0] + P.b2[j] * m[r][1];
                    }
                    bound(j, st, uj, m[r][4], m[r][5], m[r][6], m[r][7], ru);
                    double rq = 0.0;
                    if (k > 0) {
                        if (st) rq = dt * s_ + m[r][0];
                        rq -= m[r][2];
                    }
                    bound(6 + j, k >= 1 && st, qj, m[r][8], m[r][9], m[r][10], m[r][11], rq);
                    double rv = 0.0;
                    if (k > 0) {
                        if (st) {
                            rv = dt * (g[r][0] * g[r][10] + c2 * (vj - uj));
                            rv += P.a12[j] * m[r][0] + P.a22[j] * m[r][1];
                        }
                        rv -= m[r][3];
                    }
                    if (k == 0) {
                        rq = 0.0; rv = 0.0;                                               // x_0 is eliminated (lbx_0 = ubx_0)
                        n_i = fmax(n_i, fmax(fabs(xp_q - qj), fabs(xp_v - vj)));          // ... = x_hat of that QP
                    }
                    n_s = fmax(n_s, fmax(fabs(ru), fmax(fabs(rq), fabs(rv))));
                } else {
                    bdq = bd[r][0]; bdv = bd[r][1];
                }
                if (RHS) {
                    const double dxq = k == 0 ? xh_q - qj : 0.0, dxv = k == 0 ? xh_v - vj : 0.0;
                    const double vjn = vj + dxv;
                    const double gtu = st ? dt * (2.0 * P.w_u * uj + c2 * (uj - vjn)) : 0.0;
                    const double gtq = st && k >= 1 ? dt * s_ : 0.0;
                    const double gtv = st && k >= 1 ? dt * (g[r][0] * g[r][10] + c2 * (vjn - uj)) : 0.0;
                    const double rbq = st ? (dxq + P.a12[j] * dxv) + bdq : 0.0;
                    const double rbv = st ? P.a22[j] * dxv + bdv : 0.0;
                    gst(g2 + O_GT, gtu); gst(g2 + O_GT + 6, gtq); gst(g2 + O_GT + 12, gtv);
                    gst(g2 + O_GAM, 0.0); gst(g2 + O_GAM + 6, 0.0);
                    gst(g2 + O_RB, rbq); gst(g2 + O_RB + 6, rbv);
                }
            }
        }
    }
    if (RHS) {
        // stage 0's y with the feedback step embedded (residual_pass lanes 48..52; read by the SQP merit weights): lane i < 5
        const int i = lane < NTASK ? lane : 0;
        const MPC_GLOBAL double *r0 = rec(0, 0);
        double v = gld(r0 + C2 + O_R + i), gq[6], gv[6], xq[6], xv[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { gq[j] = gld(r0 + C2 + O_GQ + i * 6 + j); gv[j] = gld(r0 + C2 + O_GV + j); xq[j] = gld(r0 + O_X + j); xv[j] = gld(r0 + O_X + 6 + j); }
        double dq[6], dv[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { dq[j] = lane_val(xh_q, j) - xq[j]; dv[j] = lane_val(xh_v, j) - xv[j]; }
        if (N > 0) {
#pragma unroll
            for (int j = 0; j < 6; j++) v += gq[j] * dq[j];
            if (i == 4) {
#pragma unroll
                for (int j = 0; j < 6; j++) v += gv[j] * dv[j];
            }
            if (lane < NTASK) gst(rec(0, i) + C2 + O_Y, P.w_task[i] * v);
        }
    }
    if (NORMS && out5) {
        out5[0] = wsum(csum);
        out5[1] = wmax(n_s); out5[2] = wmax(n_e); out5[3] = wmax(n_i); out5[4] = wmax(n_c);
    }
    wait_vm<0>();
    fence();
}

// =============================================================================================== factorisation sweep
// Backward Riccati sweep, matrix AND vector recursion of a stage in the same two phases (mpc_core.h fact_pass):
//   lanes 0..35  block (a,b) of the 12x12 cost-to-go in registers for the whole sweep
//   lanes 0..17  one right-hand side of the 6x6 LDL' each (12 columns of S~ -> K, 6 of I -> R~^-1)
//   lanes 40..51 vector recursion: t = p_{k+1} + P_{k+1} rb_k ; h_u ; p_k ; R~^-1 h_u ; e
//   in  : G2 [GQ..RB] (78 doubles; row k-1 one stage ahead for Gamma_u)     out : factor row (SW4 x FT)
template <class FT>
struct FactLane {
    FT b1a, b2a, b1b, b2b, a12a, a22a, a12b, a22b;
    FT hu_c, huv_c, lm_c;
    FT mqq, mqv, mvq, mvv;
    FT qqq, qqv, qvq, qvv;
    int a, b, oqq, oqv, ovv;
};

// FASTF (fast path): the record goes out without its tail [w | R~^-1] -- only the corrector reads that (48 of 232 scalars).
template <class FT, bool FASTF = false>
SE_PASS void fact_pass()
{
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor);
    const SWs w = sm.w;
    constexpr int SOUT = FASTF ? SW4_FWD : SW4;
    constexpr int NIO = ni_of(SOUT * (int)sizeof(FT));
    Bundle<1, 39> bin;
    Bundle<NIO, SOUT * (int)sizeof(FT) / 16> bout;
    {
        const Seg si[1] = {segd(w.G2, w.ld, O_GQ, 78)};
        bin.setup(si, lane);
        const Seg so[1] = {segf<FT>(w.G4, w.ld, 0, SOUT)};
        bout.setup(so, lane);
        bout.seek(N, -1);
    }
    FactLane<FT> f;
    {
        const int a = lane < 36 ? lane / 6 : 0, b = lane < 36 ? lane % 6 : 0;
        f.a = a; f.b = b;
        f.oqq = tri(imin(a, b), imax(a, b)); f.oqv = tri(a, 6 + b); f.ovv = tri(6 + imin(a, b), 6 + imax(a, b));
        f.b1a = (FT)P.b1[a]; f.b2a = (FT)P.b2[a]; f.b1b = (FT)P.b1[b]; f.b2b = (FT)P.b2[b];
        f.a12a = (FT)P.a12[a]; f.a22a = (FT)P.a22[a]; f.a12b = (FT)P.a12[b]; f.a22b = (FT)P.a22[b];
        const double c2 = P.dt * P.w_qddot * P.cq[a] * P.cq[a];
        f.hu_c = (FT)(a == b ? P.dt * 2.0 * P.w_u + c2 + P.dt * P.lm : 0.0);
        f.huv_c = (FT)(a == b ? c2 : 0.0);
        f.lm_c = (FT)(a == b ? P.dt * P.lm : 0.0);
        f.mqq = f.mqv = f.mvq = f.mvv = (FT)0;
        f.qqq = f.qqv = f.qvq = f.qvv = (FT)0;
    }
    const FT dw0 = (FT)(P.dt * P.w_task[0]), dw1 = (FT)(P.dt * P.w_task[1]), dw2 = (FT)(P.dt * P.w_task[2]),
             dw3 = (FT)(P.dt * P.w_task[3]), dw4 = (FT)(P.dt * P.w_task[4]);
    const int jv = lane >= 40 && lane < 52 ? lane - 40 : 0;      // vector lanes: component
    const FT va12 = (FT)P.a12[jv % 6], va22 = (FT)P.a22[jv % 6];
    const FT vb = (FT)(jv < 6 ? P.b1[jv] : P.b2[jv - 6]);         // the lane's row of B
    FT b1r[6], b2r[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { b1r[i] = (FT)P.b1[i]; b2r[i] = (FT)P.b2[i]; }
    FT pr = (FT)0;                                                // vector lanes: p_{k+1}[jv]
    int tro[12];                                                  // packed offsets of row jv of the symmetric P (constant indices only)
#pragma unroll
    for (int j = 0; j < NX; j++) tro[j] = tri_sym(jv, j);
    const FT lmN = (FT)P.lm;
    int sb = 0;
    auto next_stage = [&](FT gam_u, int sbw) {
        const FT fq = f.b1a * f.mqq + f.b2a * f.mvq, fv = f.b1a * f.mqv + f.b2a * f.mvv;
        FT r = fq * f.b1b + fv * f.b2b + f.hu_c;
        r += f.a == f.b ? gam_u : (FT)0;
        sm.Rt[f.a * 6 + f.b] = (double)r;
        double *St = sm.St2[sbw];
        St[f.a * 12 + f.b] = (double)fq;
        St[f.a * 12 + 6 + f.b] = (double)(fq * f.a12b + fv * f.a22b - f.huv_c);
        const FT cq = f.a12a * f.mqq + f.a22a * f.mvq, cv = f.a12a * f.mqv + f.a22a * f.mvv;
        f.qqq = f.mqq;
        f.qqv = f.mqq * f.a12b + f.mqv * f.a22b;
        f.qvq = cq;
        f.qvv = cq * f.a12b + cv * f.a22b;
    };
    sweep<39, NIO, true>(bin, N, lane, [&](int k, double *ric_, double *ricd_, double *) {
        const double *ric = ric_;
        const double *ricd = ricd_;                               // row k-1: the next row of this backward sweep (valid for k >= 1)
        MPC_LOCAL FT *fac = (MPC_LOCAL FT *)sm.out[k & 1];
        const MPC_LOCAL FT *facn = (const MPC_LOCAL FT *)sm.out[(k + 1) & 1];   // row k+1 (valid for k < N)
#ifdef MPCB_NOCOMPUTE
        (void)ric; (void)ricd; (void)fac; (void)facn; store_out(bout, sm.out[k & 1], lane); return;
#endif
        const double *gam = ric + 36, *gt = ric + 48, *rbv = ric + 66;
        if (k == N) {
            // terminal stage: no cost, no bounds -> P_N = lm I ; p_N = gt_x ; R~, S~ of stage N-1
            for (int e = lane; e < SW4; e += WAVE) fac[e] = (FT)0;
            fence();
            if (lane < NX) fac[SPM + tri(lane, lane)] = lmN;
            if (lane < 36) {
                f.mqv = f.mvq = (FT)0;
                f.mqq = f.mvv = f.a == f.b ? lmN : (FT)0;
                if (k >= 1) next_stage((FT)ricd[36 + f.a], sb);
            } else if (lane >= 40 && lane < 52) {
                pr = (FT)gt[6 + jv];
                fac[SPV + jv] = pr;
                sm.vec[0][jv] = (double)pr;
            }
            fence();
            store_out(bout, sm.out[k & 1], lane);
            return;
        }
        // ---- B: LDL' (right-looking, redundant in every lane) + one right-hand side per lane ; vector lanes: t
        {
            const double *St = sm.St2[sb];
            FT A_[6][6];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) A_[i][j] = (FT)sm.Rt[i * 6 + j];
            FT dinv[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                dinv[j] = sizeof(FT) == 8 ? (FT)fast_rcp((double)A_[j][j]) : (FT)1 / A_[j][j];
                FT lj[6];
#pragma unroll
                for (int i = j + 1; i < 6; i++) lj[i] = A_[i][j] * dinv[j];
#pragma unroll
                for (int i = j + 1; i < 6; i++)
#pragma unroll
                    for (int r = j + 1; r <= i; r++) A_[i][r] -= lj[i] * A_[r][j];
#pragma unroll
                for (int i = j + 1; i < 6; i++) A_[i][j] = lj[i];
            }
            if (lane < 18) {
                FT x[6];
                const int col = lane < 12 ? lane : 0;
#pragma unroll
                for (int i = 0; i < 6; i++) x[i] = (FT)St[i * 12 + col];
#pragma unroll
                for (int i = 0; i < 6; i++) x[i] = lane < 12 ? x[i] : (i == lane - 12 ? (FT)1 : (FT)0);
#pragma unroll
                for (int j = 0; j < 6; j++) {
#pragma unroll
                    for (int i = j + 1; i < 6; i++) x[i] -= A_[i][j] * x[j];
                }
#pragma unroll
                for (int i = 0; i < 6; i++) x[i] *= dinv[i];
#pragma unroll
                for (int j = 5; j >= 0; j--) {
#pragma unroll
                    for (int i = 0; i < j; i++) x[i] -= A_[j][i] * x[j];
                }
                if (lane < 12) {
#pragma unroll
                    for (int i = 0; i < 6; i++) { fac[SK + i * 12 + lane] = x[i]; sm.Kf[i * 12 + lane] = (double)x[i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < 6; i++) fac[SRI + i * 6 + (lane - 12)] = x[i];
                }
            } else if (lane >= 40 && lane < 52) {
                // t = p_{k+1} + P_{k+1} rb_k (all 24 operands first: the compiler otherwise fetches them in four batches)
                FT pm_[12];
                double rb_[12];
#pragma unroll
                for (int j = 0; j < NX; j++) { pm_[j] = facn[SPM + tro[j]]; rb_[j] = rbv[j]; }
#pragma unroll
                for (int j = 0; j < NX; j++) { asm volatile("" : "+v"(pm_[j])); pin(rb_[j]); }
                FT w0 = (FT)0, w1 = (FT)0;
#pragma unroll
                for (int j = 0; j < NX; j += 2) { w0 += pm_[j] * (FT)rb_[j]; w1 += pm_[j + 1] * (FT)rb_[j + 1]; }
                const FT wv = w0 + w1;
                fac[SWV + jv] = wv;
                sm.vec[1][jv] = (double)(pr + wv);
            }
        }
        fence();
        // ---- CA: P_k block, then R~ / S~ of stage k-1 ; vector lanes: h_u, p_k, R~^-1 h_u, e
        if (lane < 36) {
            if (k > 0) {
                const double *St = sm.St2[sb];
                const double *gq = ric, *gv = ric + 30;
                FT sa[6], sva[6], kb[6], kvb[6];
#pragma unroll
                for (int m = 0; m < 6; m++) {
                    sa[m] = (FT)St[m * 12 + f.a]; sva[m] = (FT)St[m * 12 + 6 + f.a];
                    kb[m] = (FT)sm.Kf[m * 12 + f.b]; kvb[m] = (FT)sm.Kf[m * 12 + 6 + f.b];
                }
                const FT ga0 = (FT)gq[f.a], ga1 = (FT)gq[6 + f.a], ga2 = (FT)gq[12 + f.a], ga3 = (FT)gq[18 + f.a], ga4 = (FT)gq[24 + f.a];
                const FT gb0 = (FT)gq[f.b], gb1 = (FT)gq[6 + f.b], gb2 = (FT)gq[12 + f.b], gb3 = (FT)gq[18 + f.b], gb4 = (FT)gq[24 + f.b];
                const FT gva = (FT)gv[f.a], gvb = (FT)gv[f.b];
                const FT gam_q = (FT)gam[6 + f.a], gam_u = (FT)ricd[36 + f.a];
                FT pqq = f.qqq + (dw0 * ga0 * gb0 + dw1 * ga1 * gb1 + dw2 * ga2 * gb2 + dw3 * ga3 * gb3 + dw4 * ga4 * gb4) + f.lm_c;
                FT pqv = f.qqv + dw4 * ga4 * gvb;
                FT pvq = f.qvq + dw4 * gva * gb4;
                FT pvv = f.qvv + dw4 * gva * gvb + (f.huv_c + f.lm_c);
                pqq += f.a == f.b ? gam_q : (FT)0;
#pragma unroll
                for (int m = 0; m < 6; m++) {
                    pqq -= sa[m] * kb[m]; pqv -= sa[m] * kvb[m];
                    pvq -= sva[m] * kb[m]; pvv -= sva[m] * kvb[m];
                }
                f.mqq = pqq; f.mqv = pqv; f.mvq = pvq; f.mvv = pvv;
                if (f.a <= f.b) { fac[SPM + f.oqq] = pqq; fac[SPM + f.ovv] = pvv; }
                fac[SPM + f.oqv] = pqv;
                next_stage(gam_u, sb ^ 1);
            }
        } else if (lane >= 40 && lane < 52) {
            // (register arrays are only ever indexed by constants: a lane-dependent index would put them in scratch memory,
            // and a scratch load in the stage loop drains every fetch in flight)
            // operands of the whole block first (none of them is computed here): one LDS round trip instead of six
            const int i6 = jv < 6 ? jv : jv - 6;
            double g_[6], v1_[12];
            FT kf_[6], ri_[6];
#pragma unroll
            for (int m = 0; m < 6; m++) { g_[m] = gt[m]; kf_[m] = (FT)sm.Kf[m * 12 + jv]; ri_[m] = fac[SRI + i6 * 6 + m]; }
#pragma unroll
            for (int m = 0; m < NX; m++) v1_[m] = sm.vec[1][m];
            double t_d = sm.vec[1][jv], oq_d = sm.vec[1][jv >= 6 ? jv - 6 : jv], gj_d = gt[6 + jv], rb_d = rbv[jv];
#pragma unroll
            for (int m = 0; m < 6; m++) { pin(g_[m]); asm volatile("" : "+v"(kf_[m])); asm volatile("" : "+v"(ri_[m])); }
#pragma unroll
            for (int m = 0; m < NX; m++) pin(v1_[m]);
            pin(t_d); pin(oq_d); pin(gj_d); pin(rb_d);
            FT hu[6];
#pragma unroll
            for (int m = 0; m < 6; m++) hu[m] = (FT)g_[m] + b1r[m] * (FT)v1_[m] + b2r[m] * (FT)v1_[6 + m];
            const FT t = (FT)t_d;
            const FT oq = (FT)oq_d;
            FT pj = (FT)gj_d + (jv < 6 ? t : va12 * oq + va22 * t);
            FT s0 = (FT)0, s1 = (FT)0;
#pragma unroll
            for (int m = 0; m < 6; m += 2) { s0 += kf_[m] * hu[m]; s1 += kf_[m + 1] * hu[m + 1]; }
            pj -= s0 + s1;
            pr = pj;
            fac[SPV + jv] = pj;
            FT v0 = (FT)0, v1 = (FT)0;
#pragma unroll
            for (int m = 0; m < 6; m += 2) { v0 += ri_[m] * hu[m]; v1 += ri_[m + 1] * hu[m + 1]; }
            const FT vh = v0 + v1;
            if (jv < 6) fac[SVH + jv] = vh;
            fac[SEo + jv] = (FT)rb_d - vb * vh;
        } else if (lane >= 52 && lane < 56) {
            // padding scalars of the record: defined values
            if (lane < 54) fac[SVH + 6 + (lane - 52)] = (FT)0;
            else fac[SPM + 78 + (lane - 54)] = (FT)0;
        }
        sb ^= 1;
        fence();
        store_out(bout, sm.out[k & 1], lane);
    });
}

// =============================================================================================== forward sweeps
// du = -K dx - R~^-1 h_u ; dx+ = e + A dx - B K dx ; dpi = p + P dx ; dlam, dt (HPIPM compute_lam_t), largest
// feasible step, the three centering sums.  One phase per stage, four roles at once:
//   lanes 0..11  dx_{k+1}           lanes 16..21  du_k           lanes 32..43  dpi_{k-1} (final sweep only)
//   lanes 48..59 dlam, dt, step length and sums of stage k-1 (its du / dx were written one phase earlier)
//   in  : factor [K..E] or [K..P] | G1 [QLAM,QT] | G3 [RD,RM]      out : G3 [DLAM,DT] (affine) or [DW..DT]
struct StepInfo {
    double alpha, S0, S1, S2;
};
// FAST (bound-inactive fast path, mpc_ipm.h): the step is the candidate solution of the equality-constrained QP; the lagging role
// forms the candidate's slacks (-> DT; DLAM = 0) from the NLP iterate's u, q instead of dlam / dt, and `alpha` comes back 1.0 when
// every bounded component clears its bounds by ipm::FAST_MARGIN (and nothing is NaN), else 0.0.
// FAST: `cand` = where the candidate goes: 1 -- the Newton-step slots of G3 (DW | DPI | DLAM | DT), 0 -- the QP-iterate slots of G1
// (QW | QPI | QLAM | QT), whichever does NOT hold the QP iterate at the moment (SQP_RTI keeps the iterate in either and flips on
// acceptance instead of copying, see ipm_solve); the multiplier step is stored UNSHIFTED there (dpi_k with stage k, as QPI is), so that the
// 78 doubles of a stage are the QP iterate as they stand.
template <class FT, bool AFFINE, bool FAST = false>
SE_PASS StepInfo forward_pass(int cand = 1)
{
    static_assert(!FAST || !AFFINE, "the fast path takes the full step");
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor);
    const SWs w = sm.w;
    constexpr int LF = AFFINE ? SW4_AFF : SW4_FWD;
    constexpr int FB = LF * (int)sizeof(FT);                       // bytes of the factor part
    constexpr int REST = FAST ? 18 : 96;                           // doubles behind the factor: [QLAM, QT | RD, RM], or (FAST) X | U of the NLP iterate
    constexpr int ITEMS = (FB + REST * 8) / 16;
    constexpr int NII = ni_of(FB + REST * 8);
    constexpr int I_LT = FB / 8, I_R = FAST ? I_LT : I_LT + 48;    // doubles (FAST: no residual part -- the operand fetches below stay inside the slot)
    Bundle<NII, ITEMS> bin;
    Bundle<1, AFFINE ? 24 : 39> bout;
    {
        if (FAST) { const Seg si[2] = {segf<FT>(w.G4, w.ld, 0, LF), segd(w.G1, w.ld, 0, 18)}; bin.setup(si, lane); }
        else { const Seg si[3] = {segf<FT>(w.G4, w.ld, 0, LF), segd(w.G1, w.ld, O_QLAM, 48), segd(w.G3, w.ld, O_RD, 48)}; bin.setup(si, lane); }
        if (AFFINE) { const Seg so[1] = {segd(w.G3, w.ld, O_DLAM, 48)}; bout.setup(so, lane); }
        else if (FAST && uni(cand) == 0) { const Seg so[1] = {segd(w.G1, w.ld, O_QW, 78)}; bout.setup(so, lane); }
        else { const Seg so[1] = {segd(w.G3, w.ld, O_DW, 78)}; bout.setup(so, lane); }
        bout.seek(0, 1);                                           // (stage k stores row k-1, starting with row 0)
    }
    const int j12 = lane & 15;                                     // component of the 12-lane roles
    const int j6 = j12 % 6;
    const FT a12 = (FT)P.a12[j6], a22 = (FT)P.a22[j6], b1 = (FT)P.b1[j6], b2 = (FT)P.b2[j6];
    double al = 1.0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
    // the 12 factor entries a lane multiplies with dx_k, by role: row i of K (state recursion, lanes 0..11, and input
    // step, lanes 16..21) or the packed offsets of a row of P (dpi, lanes 32..43); `ex`: the role's additive entry
    int tix[12];
    const int krow = lane < 12 ? (lane < 6 ? lane : lane - 6) : (lane >= 16 && lane < 22 ? lane - 16 : 0);
#pragma unroll
    for (int i = 0; i < NX; i++) tix[i] = lane >= 32 && lane < 44 ? SPM + tri_sym(lane - 32, i) : SK + krow * 12 + i;
    const int ex = lane < 12 ? SEo + lane : (lane >= 16 && lane < 22 ? SVH + (lane - 16) : (lane >= 32 && lane < 44 ? SPV + (lane - 32) : 0));
    const int lo12 = lane < 12 ? lane : 0, lov = lane < 6 ? lane + 6 : (lane < 12 ? lane : 0);
    const int jl = lane >= 48 && lane < 60 ? lane - 48 : 0;        // lagging role: bounded component
    const bool jl_lo = bnd_lo(P, jl) > -BOUND_INF, jl_hi = bnd_hi(P, jl) < BOUND_INF;
    const double jb_lo = bnd_lo(P, jl), jb_hi = bnd_hi(P, jl);
    const int jl_val = jl < 6 ? O_U + jl : O_X + jl - 6;           // FAST: the bounded variable itself in the G1 row
    if (lane < NX) sm.vec[0][lane] = 0.0;                          // dx_0 = 0: x_0 is pinned by the init pass
    auto stage = [&](int k, double *row_, double *, double *rowp_) {
        const double *row = row_, *rowp = rowp_;                   // rowp: row k-1, the previous row of this forward sweep
        const MPC_LOCAL FT *fac = (const MPC_LOCAL FT *)row;
        double *o = sm.out[k & 1], *op = sm.out[(k + 1) & 1];
#ifdef MPCB_NOCOMPUTE
        (void)row; (void)rowp; (void)fac; (void)o; if (k >= 1) store_out(bout, AFFINE ? op + 30 : op, lane); return;
#endif
        const double *dxk = sm.vec[k & 1];
        // operands of all four roles first (one LDS round trip per stage), then the role blocks on registers
        FT m[12];
        double x[12];
#pragma unroll
        for (int j = 0; j < NX; j++) { m[j] = fac[tix[j]]; x[j] = dxk[j]; }
        FT e1 = fac[ex];
        double own_d = dxk[lo12], ov_d = dxk[lov];
        const double *lt = rowp + I_LT, *r = rowp + I_R;
        double dvl = op[jl], ll = lt[FAST ? jl_val : jl], tl = lt[FAST ? 0 : 24 + jl], lu = lt[FAST ? 0 : 12 + jl], tu = lt[FAST ? 0 : 36 + jl];
        double rdl = r[FAST ? 0 : jl], rdu = r[FAST ? 0 : 12 + jl], rml = r[FAST ? 0 : 24 + jl], rmu = r[FAST ? 0 : 36 + jl];
        if (FAST) rdl = op[12 + (jl < 6 ? jl : jl - 6)];           // (the velocity step of the same joint: NaN check)
#pragma unroll
        for (int j = 0; j < NX; j++) { asm volatile("" : "+v"(m[j])); pin(x[j]); }
        asm volatile("" : "+v"(e1));
        pin(own_d); pin(ov_d); pin(dvl); pin(ll); pin(tl); pin(lu); pin(tu); pin(rdl); pin(rdu); pin(rml); pin(rmu);
        // the three roles that multiply a factor row with dx_k share ONE dot product (same instruction sequence, per-lane
        // operands; the input-step and dpi roles start their even-term sum from their additive entry, as before)
        FT s0 = lane < 12 ? (FT)0 : e1, s1 = (FT)0;
#pragma unroll
        for (int j = 0; j < NX; j += 2) { s0 += m[j] * (FT)x[j]; s1 += m[j + 1] * (FT)x[j + 1]; }
        const FT kd = s0 + s1;
        if (lane < 12 && k <= N) {
            const FT own = (FT)own_d, ov = (FT)ov_d;
            const FT v = e1 + (lane < 6 ? own + a12 * ov - b1 * kd : a22 * own - b2 * kd);
            o[6 + lane] = (double)own;                              // dx_k
            sm.vec[(k + 1) & 1][lane] = (double)v;                  // dx_{k+1}
        } else if (lane >= 16 && lane < 22 && k <= N) {
            o[lane - 16] = k < N ? -(double)kd : 0.0;               // du_k (stage N has no input: 0)
        } else if (!AFFINE && lane >= 32 && lane < 44 && k <= N) {
            if (FAST) {
                if (k >= 1) op[18 + (lane - 32)] = (double)kd;      // dpi_{k-1} goes with stage k-1, whose row is still here (stored below)
                if (k == N) o[18 + (lane - 32)] = 0.0;              // no multiplier beyond the last dynamics
            } else {
                o[18 + (lane - 32)] = k >= 1 ? (double)kd : 0.0;    // DPI slot of stage k holds dpi_{k-1}
            }
        } else if (lane >= 48 && lane < 60 && k >= 1) {
            const int j = jl, kp = k - 1;
            const double dv = dvl;                                  // du_{k-1}[j] or dq_{k-1}[j-6]
            const bool hc = j < 6 ? kp < N : (kp >= 1 && kp < N);
            const bool blo = hc && jl_lo, bhi = hc && jl_hi;
            double dtl = 0, dll = 0, dtu = 0, dlu = 0;
            if (FAST) {
                const double val = ll;                              // the NLP iterate's u_j / q_(j-6) of stage k-1
                const bool good = ipm::fast_side(blo, bhi, dv, jb_lo - val, jb_hi - val, dtl, dtu);
                al = good && dv == dv && rdl == rdl ? al : 0.0;
            } else {
            if (blo) {
                dtl = dv + rdl;
                dll = -(rml + ll * dtl) * fast_rcp(tl);
                if (dll < 0 && ll + al * dll < 0) al = -ll * fast_rcp(dll);
                if (dtl < 0 && tl + al * dtl < 0) al = -tl * fast_rcp(dtl);
                a0 += ll * tl; a1 += ll * dtl + tl * dll; a2 += dll * dtl;
            }
            if (bhi) {
                dtu = -dv + rdu;
                dlu = -(rmu + lu * dtu) * fast_rcp(tu);
                if (dlu < 0 && lu + al * dlu < 0) al = -lu * fast_rcp(dlu);
                if (dtu < 0 && tu + al * dtu < 0) al = -tu * fast_rcp(dtu);
                a0 += lu * tu; a1 += lu * dtu + tu * dlu; a2 += dlu * dtu;
            }
            }
            op[30 + j] = dll; op[42 + j] = dlu;
            op[54 + j] = dtl; op[66 + j] = dtu;
        }
        fence();
        if (k >= 1) store_out(bout, AFFINE ? op + 30 : op, lane);
    };
    sweep<ITEMS, 1, false>(bin, N, lane, stage);
    fence();
    {   // the lagging role's last stage: "row N+1" does not exist, its predecessor is row N
        const Ring rg = make_ring<ITEMS>();
        stage(N + 1, rg.row(N + 1), nullptr, rg.row(N));
    }
    StepInfo s;
    s.alpha = wmin(al); s.S0 = wsum(a0); s.S1 = wsum(a1); s.S2 = wsum(a2);
    return s;
}

// =============================================================================================== corrector + backward solve
// Centering-corrector right-hand side (HPIPM compute_centering_correction) and the backward SOLVE sweep on the
// existing factor:  t = p_{k+1} + w_k ; h_u = gt_u + B' t ; p_k = gt_x + A' t - K' h_u ; R~^-1 h_u ; e.
// Two phases per stage; the corrected gradient of stage k-1 is formed (lanes 32..43) while stage k is solved.
//   in  : G1 [QLAM,QT] | G3 [RG,RD] | G3 [DLAM,DT] | G2 [GT,RB] | factor K | factor [w, R~^-1]
//   out : G3 RM | factor [R~^-1 h_u, e, p]
template <class FT>
SE_PASS void corrector_pass(double sigma_mu)
{
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor);
    const SWs w = sm.w;
    constexpr int I_3 = 48, I_DL = 90, I_GB = 138, I_F = 168;      // doubles: lam,t | RG,RD | DLAM,DT | GT,RB | factor parts
    constexpr int FBYTES = (72 + 48) * (int)sizeof(FT);
    constexpr int ITEMS = (168 * 8 + FBYTES) / 16;
    constexpr int NII = ni_of(168 * 8 + FBYTES);
    constexpr int OB = 24 * 8 + 32 * (int)sizeof(FT);
    Bundle<NII, ITEMS> bin;
    Bundle<1, OB / 16> bout;
    {
        const Seg si[6] = {segd(w.G1, w.ld, O_QLAM, 48), segd(w.G3, w.ld, 0, 42), segd(w.G3, w.ld, O_DLAM, 48), segd(w.G2, w.ld, O_GT, 30),
                           segf<FT>(w.G4, w.ld, SK, 72), segf<FT>(w.G4, w.ld, SWV, 48)};
        bin.setup(si, lane);
        const Seg so[2] = {segd(w.G3, w.ld, O_RM, 24), segf<FT>(w.G4, w.ld, SVH, 32)};
        bout.setup(so, lane);
        bout.seek(N, -1);
        (void)OB;
    }
    const int j12 = lane & 15, j6 = j12 % 6;
    const FT va12 = (FT)P.a12[j6], va22 = (FT)P.a22[j6];
    const FT hb1 = (FT)P.b1[lane < 6 ? lane : 0], hb2 = (FT)P.b2[lane < 6 ? lane : 0];       // h_u lanes
    const int je = lane >= 16 && lane < 28 ? lane - 16 : 0;
    const FT eb = (FT)(je < 6 ? P.b1[je] : P.b2[je - 6]);                                   // e lanes: row of B
    const int jc = lane >= 32 && lane < 44 ? lane - 32 : 0;
    const bool jc_lo = bnd_lo(P, jc) > -BOUND_INF, jc_hi = bnd_hi(P, jc) < BOUND_INF;      // corrector lanes
    // phase Y: the six factor entries a lane multiplies with h_u: column `lane` of K (p_k lanes) or row i6 of R~^-1 (84 = offset of R~^-1 behind K and w)
    int fx[6];
#pragma unroll
    for (int m = 0; m < 6; m++) fx[m] = lane < 12 ? m * 12 + lane : (lane >= 16 && lane < 28 ? 84 + (je < 6 ? je : je - 6) * 6 + m : 0);
    // corrector of one landed row -> sm.gtc[kr & 1] (18 entries) and the RM slot of that row's output image
    auto corr_row = [&](const double *row, int kr, int j) {
        const double *lt = row, *r3 = row + I_3, *dl = row + I_DL, *gtb = row + I_GB;
        double *og = sm.gtc[kr & 1], *rmo = sm.out[kr & 1];
        const bool hc = j < 6 ? kr < N : (kr >= 1 && kr < N);
        const bool blo = hc && jc_lo, bhi = hc && jc_hi;
        const double ll = lt[j], tl = lt[24 + j], lu = lt[12 + j], tu = lt[36 + j];
        const double dll = dl[j], dtl = dl[24 + j], dlu = dl[12 + j], dtu = dl[36 + j];
        const double rdl = r3[18 + j], rdu = r3[18 + 12 + j];
        double gt = r3[j];
        double rml, rmu;
        gt += ipm::corrector_side(blo, ll, tl, dll, dtl, rdl, sigma_mu, rml);
        gt -= ipm::corrector_side(bhi, lu, tu, dlu, dtu, rdu, sigma_mu, rmu);
        og[j] = hc ? gt : gtb[j];
        if (j < 6) og[12 + j] = gtb[12 + j];
        rmo[j] = rml; rmo[12 + j] = rmu;
    };
    sweep<ITEMS, 1, true>(bin, N, lane, [&](int k, double *row_, double *rowd_, double *) {
        const double *row = row_, *rowd = rowd_;                   // rowd: row k-1, the next row of this backward sweep
        if (k == N) {
            if (lane >= 32 && lane < 44) corr_row(row, N, lane - 32);
            fence();
        }
#ifdef MPCB_NOCOMPUTE
        (void)rowd; store_out(bout, sm.out[k & 1], lane); return;
#endif
        const double *gtc = sm.gtc[k & 1], *gtb = row + I_GB;
        const MPC_LOCAL FT *kf = (const MPC_LOCAL FT *)(row + I_F), *wv = kf + 72, *ri = kf + 84;
        double *o = sm.out[k & 1];
        MPC_LOCAL FT *ofac = (MPC_LOCAL FT *)(o + 24);             // [VH 6 +2 | E 12 | PV 12]
        const double *pn = sm.vec[(k + 1) & 1];                     // p_{k+1}
        // ---- X: h_u (lanes 0..5) ; corrector of row k-1 (lanes 32..43).  Operands of both roles first (one LDS round
        //      trip per phase), then the role blocks on registers.
        {
            const int l5 = lane < 6 ? lane : 0;
            double pn0 = pn[l5], pn1 = pn[6 + l5], g0 = gtc[l5];
            FT wv0 = wv[l5], wv1 = wv[6 + l5];
            const double *lt = rowd, *r3 = rowd + I_3, *dl = rowd + I_DL, *gtd = rowd + I_GB;
            double ll = lt[jc], tl = lt[24 + jc], lu = lt[12 + jc], tu = lt[36 + jc];
            double dll = dl[jc], dtl = dl[24 + jc], dlu = dl[12 + jc], dtu = dl[36 + jc];
            double rdl = r3[18 + jc], rdu = r3[18 + 12 + jc], gt = r3[jc], gb0 = gtd[jc], gb1 = gtd[12 + (jc < 6 ? jc : 0)];
            pin(pn0); pin(pn1); pin(g0); asm volatile("" : "+v"(wv0)); asm volatile("" : "+v"(wv1));
            pin(ll); pin(tl); pin(lu); pin(tu); pin(dll); pin(dtl); pin(dlu); pin(dtu); pin(rdl); pin(rdu); pin(gt); pin(gb0); pin(gb1);
            if (lane < 6 && k < N) {
                const FT t0 = (FT)pn0 + wv0, t1 = (FT)pn1 + wv1;
                sm.vec[2][lane] = (double)((FT)g0 + hb1 * t0 + hb2 * t1);
            } else if (lane >= 32 && lane < 44 && k >= 1) {
                // (corr_row of row k-1 on registers)
                const int kr = k - 1, j = jc;
                double *og = sm.gtc[kr & 1], *rmo = sm.out[kr & 1];
                const bool hc = j < 6 ? kr < N : (kr >= 1 && kr < N);
                const bool blo = hc && jc_lo, bhi = hc && jc_hi;
                double rml, rmu;
                gt += ipm::corrector_side(blo, ll, tl, dll, dtl, rdl, sigma_mu, rml);
                gt -= ipm::corrector_side(bhi, lu, tu, dlu, dtu, rdu, sigma_mu, rmu);
                og[j] = hc ? gt : gb0;
                if (j < 6) og[12 + j] = gb1;
                rmo[j] = rml; rmo[12 + j] = rmu;
            }
        }
        fence();
        // ---- Y: p_k (lanes 0..11), R~^-1 h_u and e (lanes 16..27)
        {
            const int l12 = lane < 12 ? lane : 0, l6 = l12 >= 6 ? l12 - 6 : 0;
            double gc = gtc[6 + l12], pa = pn[l12], pb = pn[l6], gb = gtb[18 + je];
            FT wa = wv[l12], wb = wv[l6];
            FT fm[6];
            double hm[6];
#pragma unroll
            for (int m = 0; m < 6; m++) { fm[m] = kf[fx[m]]; hm[m] = sm.vec[2][m]; }
            pin(gc); pin(pa); pin(pb); pin(gb); asm volatile("" : "+v"(wa)); asm volatile("" : "+v"(wb));
#pragma unroll
            for (int m = 0; m < 6; m++) { asm volatile("" : "+v"(fm[m])); pin(hm[m]); }
            // both roles multiply six factor entries with h_u: one dot product for the two
            FT d0 = (FT)0, d1 = (FT)0;
#pragma unroll
            for (int m = 0; m < 6; m += 2) { d0 += fm[m] * (FT)hm[m]; d1 += fm[m + 1] * (FT)hm[m + 1]; }
            const FT dd = d0 + d1;
            if (lane < 12) {
                FT pj;
                if (k == N) {
                    pj = (FT)gc;
                } else {
                    const FT t = (FT)pa + wa;
                    const FT oq = lane >= 6 ? (FT)pb + wb : (FT)0;
                    pj = (FT)gc + (lane < 6 ? t : va12 * oq + va22 * t) - dd;
                }
                sm.vec[k & 1][lane] = (double)pj;
                ofac[20 + lane] = pj;
            } else if (lane >= 16 && lane < 28) {
                const int j = lane - 16;
                FT vh = (FT)0, e = (FT)0;
                if (k < N) {
                    vh = dd;
                    e = (FT)gb - eb * vh;
                }
                if (j < 6) ofac[j] = vh;
                ofac[8 + j] = e;
            } else if (lane >= 28 && lane < 30) {
                ofac[6 + (lane - 28)] = (FT)0;
            }
        }
        fence();
        store_out(bout, o, lane);
    });
}

// =============================================================================================== fast path: commit
// The accepted candidate becomes the QP iterate.  SQP_RTI does not copy anything for that: the forward sweep left the candidate in
// whichever of the two 78-double slots of a stage (G1 [QW | QPI | QLAM | QT], G3 [DW | DPI | DLAM | DT]: same layout) does not hold the
// iterate, unshifted, and ipm_solve flips the slot index (`cur`, kept in the workspace, state[28]); lin_pass and rti_items read the iterate
// through it.  This pass is what remains: the COPY of the G3 slot into the G1 slot -- 39 16-byte items per stage -- that puts the iterate
// where the interior-point passes and the SQP line search expect it: before an interior-point solve when the iterate sits in G3
// (rare), and at every acceptance in full SQP (`embed_x0`: stage 0's x part is x_hat - x_0 there, lbx_0 = ubx_0; SQP_RTI's lin_pass
// has written it already).
SE_PASS void fast_commit(bool embed_x0)
{
    SSmem &sm = g_ssm;
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor), NS = N + 1;
    const SWs w = sm.w;
    const int LD = uni(w.ld >> 3);
    MPC_GLOBAL char *const gb = (MPC_GLOBAL char *)(((unsigned long long)(unsigned)uni((int)((unsigned long long)w.G1 >> 32)) << 32) |
                                                    (unsigned)uni((int)(unsigned long long)w.G1));
    auto rec = [&](int k, int col) { return (MPC_GLOBAL double *)(gb + (unsigned)((k * LD + col) << 3)); };
    constexpr int C3 = W1 + W2;
    wait_vm<0>();
    fence();
    {
        constexpr int IPS = 39, R = 8;
        const int items = NS * IPS;
        for (int base = 0; base < items; base += R * WAVE) {
            D2 stp[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = imin(base + r * WAVE + lane, items - 1), k = e / IPS, c = 2 * (e - k * IPS);
                stp[r] = *(MPC_GLOBAL const D2 *)(rec(k, c) + C3 + O_DW);
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int e = base + r * WAVE + lane;
                if (e < items) {
                    const int k = e / IPS, c = 2 * (e - k * IPS);
                    D2 v = stp[r];
                    if (embed_x0 && k == 0 && c >= 6 && c < 18) {                // x_0 = x_hat (lbx_0 = ubx_0): the sweep's dx_0 is 0
                        v.x = sm.xhat[c - 6] - *(rec(0, c - 6) + O_X); v.y = sm.xhat[c - 5] - *(rec(0, c - 5) + O_X);
                    }
                    *(MPC_GLOBAL D2 *)(rec(k, c) + O_QW) = v;
                }
            }
        }
    }
    wait_vm<0>();
    fence();
}

// =============================================================================================== IPM driver
// HPIPM d_ocp_qp_ipm_solve main loop (mpc_core.h ipm_solve).  Returns HPIPM status 0 ok / 1 max-iter / 2 min-step / 3 NaN.
#ifdef MPCB_SPROF
#define SPROF_T0(v) const double v = wclock()
#define SPROF_ADD(i, v) if (threadIdx.x == 0) g_ssm.w.state[32 + (i)] += wclock() - v
#else
#define SPROF_T0(v)
#define SPROF_ADD(i, v)
#endif
// `fast`: [0] QPs left before the next fast-path attempt, [1] length of the current suspension (ipm::fast_backoff); the fast path
// runs in fp64 only -- ONE fp32 Riccati solve is not a solution to qp_tol (the fp32 leg refines through the fp64 residuals of the loop).
// `cur` (SQP_RTI; null = full SQP): which slot of the stage records holds the QP iterate -- 0: G1 [QW..QT], 1: G3 [DW..DT] (fast_commit
// comment).  An accepted fast-path candidate becomes the iterate by flipping it; the interior-point loop wants the iterate in G1.
template <class FT>
SE_DEV int ipm_solve(int qp_iter_max, int *iters_out, int *fast, double *nlp_prev = nullptr, int *cur = nullptr)
{
    SSmem &sm = g_ssm;
    const double tol = sm.P.qp_tol;
    int tried = 0;
#ifdef MPCB_STREAM_SEQ_RES      // (A/B builds: the sequential passes read the iterate in G1 only)
    cur = nullptr;
#endif
    const int slot = cur ? *cur : 0;
    if (sizeof(FT) == 8 && uni(sm.P.fast_off == 0.0 ? 1 : 0)) {
        if (fast[0] > 0) fast[0]--;
        else {
            tried = 1;
            SPROF_T0(t_r);
#ifdef MPCB_STREAM_SEQ_RES
            if (nlp_prev) residual_pass<4>(0.0, nlp_prev); else residual_pass<3>(0.0);
#else
            if (nlp_prev) rti_items<true, true>(1, nlp_prev, slot); else rti_items<false, true>(0, nullptr);
#endif
            SPROF_ADD(10, t_r);
            nlp_prev = nullptr;                                    // the previous step's residuals are done, whatever happens next
            SPROF_T0(t_f);
            fact_pass<FT, true>();
            SPROF_ADD(11, t_f);
            SPROF_T0(t_w);
            const double ok = unid(forward_pass<FT, false, true>(slot ^ 1).alpha);     // the candidate goes to the slot that is NOT the iterate
            SPROF_ADD(12, t_w);
            if (uni(ok > 0.5 ? 1 : 0)) {
                SPROF_T0(t_c);
                if (cur) *cur = slot ^ 1;        // SQP_RTI: it IS the iterate now
                else fast_commit(true);          // full SQP: copied to G1, x_0 embedded
                SPROF_ADD(13, t_c);
#ifdef MPCB_SPROF
                if (threadIdx.x == 0) g_ssm.w.state[32 + 15] += 1.0;
#endif
                fast[1] = 0;
                *iters_out = 1;
                return 0;
            }
            fast[1] = ipm::fast_backoff(fast[1]);
            fast[0] = fast[1];
        }
    }
    // nlp_prev: the NLP residual / cost of the previous step's iterate is still to be evaluated -- by this QP's first pass
#ifdef MPCB_STREAM_SEQ_RES
    IpmNorms r = nlp_prev ? residual_pass<2>(0.0, nlp_prev) : residual_pass<0>(0.0);
#else
    if (nlp_prev) rti_items<true, false>(1, nlp_prev, slot);     // (the same items as everywhere else: see rti_items)
    if (cur && slot == 1) { fast_commit(false); *cur = 0; }      // the interior-point passes expect the iterate in G1 (this also overwrites a rejected candidate there)
    IpmNorms r = residual_pass<0>(0.0);
#endif
    const double nc = unid(r.nc);
    double mu = nc > 0 ? unid(r.smu) / nc : 0.0;
    int it = 0, status = 1;
    double alpha = 1.0;
    for (;; it++) {
        const double n0 = unid(r.ng), n1 = unid(r.nb), n2 = unid(r.nd), n3 = unid(r.nm);
        int stop = -1;
        if (n0 != n0 || n1 != n1 || n2 != n2 || n3 != n3) stop = 3;
        else if (!(n0 > tol || n1 > tol || n2 > tol || n3 > tol)) stop = 0;
        else if (it >= qp_iter_max) stop = 1;
        else if (!(alpha > 1e-12)) stop = 2;
        stop = uni(stop);
#if defined(MPCB_NOCOMPUTE) || defined(MPCB_FIXED_IT)
        stop = it >= 3 ? 0 : -1;     // timing builds: three iterations per QP, whatever the (meaningless) norms say
#endif
        if (stop >= 0) { status = stop; break; }
        SPROF_T0(tf);
        fact_pass<FT>();
        SPROF_ADD(0, tf);
        const bool has_bounds = nc > 0;
        if (has_bounds) {
            SPROF_T0(ta);
            const StepInfo sa = forward_pass<FT, true>();
            SPROF_ADD(1, ta);
            const double a_aff = unid(sa.alpha);
            const double mu_aff = (unid(sa.S0) + a_aff * (unid(sa.S1) + a_aff * unid(sa.S2))) / nc;
            const double sigma = ipm::sigma(mu_aff, mu);
            SPROF_T0(tc);
            corrector_pass<FT>(sigma * mu);
            SPROF_ADD(2, tc);
            SPROF_T0(tw);
            alpha = unid(forward_pass<FT, false>().alpha);
            SPROF_ADD(3, tw);
        } else {
            alpha = unid(forward_pass<FT, false>().alpha);
        }
        const double a = ipm::step_scale(alpha);
        SPROF_T0(tr);
#ifdef MPCB_STREAM_SEQ_RES
        r = residual_pass<1>(a);
#else
        r = residual_items_ok(uni(sm.n_hor)) ? residual_items(a) : residual_pass<1>(a);
#endif
        SPROF_ADD(4, tr);
        mu = nc > 0 ? unid(r.smu) / nc : 0.0;
    }
#ifdef MPCB_SPROF
    if (threadIdx.x == 0) g_ssm.w.state[32 + 7] += it;
#endif
    *iters_out = it + tried;
    return status;
}

// =============================================================================================== NLP passes (SQP_RTI)
// Linearisation: lane <-> stage, straight from / to HBM (once per MPC step; each lane's 41 outputs land in its own
// stage record, the L2 merges them into full lines).  Optionally applies the QP step first.
// With `sqp_mult` the NLP multipliers are blended towards the QP's with the same step (acados
// ocp_nlp_update_variables_sqp): N* += alpha (Q* - N*) on [pi | lam | t] (60 entries of G5).
// `slot` (SQP_RTI: 0 / 1, full SQP: -1): where the QP iterate -- the step to apply -- sits (rti_items); in SQP_RTI stage 0's x part of it,
// x_hat - x_0 (lbx_0 = ubx_0: the sweeps' dx_0 is 0), is formed AND written here, the only pass that has the x_0 it refers to.
SE_PASS void lin_pass(double alpha, bool do_update, bool sqp_mult = false, int slot = -1)
{
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const Robot &rb = robot_in_lds();
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor);
    const SWs w = sm.w;
    for (int k0 = 0; k0 <= N; k0 += WAVE) {
        const int k = k0 + lane;
#ifdef MPCB_SPROF_LIN
        const double tl0 = wclock();
#endif
        if (k > N) continue;
        MPC_GLOBAL double *r1 = (MPC_GLOBAL double *)((char *)w.G1 + (size_t)k * w.ld);
        MPC_GLOBAL double *r2 = (MPC_GLOBAL double *)((char *)w.G2 + (size_t)k * w.ld);
        double xx[12], uu[6];
#pragma unroll
        for (int i = 0; i < 12; i++) xx[i] = r1[O_X + i];
#pragma unroll
        for (int i = 0; i < 6; i++) uu[i] = r1[O_U + i];
        if (do_update && uni(slot) >= 0) {
            MPC_GLOBAL double *qw = uni(slot) ? (MPC_GLOBAL double *)((char *)w.G3 + (size_t)k * w.ld) + O_DW : r1 + O_QW;
            // (every load before the first store: a load issued behind a store waits for the store's acknowledgement, ~1 us each)
            double st[18];
#pragma unroll
            for (int i = 0; i < 18; i++) st[i] = qw[i];
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < 12; i++) st[6 + i] = sm.xhat[i] - xx[i];
            }
#pragma unroll
            for (int i = 0; i < 12; i++) xx[i] += alpha * st[6 + i];
#pragma unroll
            for (int i = 0; i < 6; i++) uu[i] += alpha * st[i];
#pragma unroll
            for (int i = 0; i < 12; i++) r1[O_X + i] = xx[i];
            if (k < N) {
#pragma unroll
                for (int i = 0; i < 6; i++) r1[O_U + i] = uu[i];
            }
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < 12; i++) qw[6 + i] = st[6 + i];
            }
        } else if (do_update) {
#pragma unroll
            for (int i = 0; i < 12; i++) { xx[i] += alpha * r1[O_QW + 6 + i]; r1[O_X + i] = xx[i]; }
            if (k < N) {
#pragma unroll
                for (int i = 0; i < 6; i++) { uu[i] += alpha * r1[O_QW + i]; r1[O_U + i] = uu[i]; }
            }
            if (sqp_mult) {
                MPC_GLOBAL double *r5 = (MPC_GLOBAL double *)((char *)w.G5 + (size_t)k * w.ld);
                // NPI | NLAM | NT <- QPI | QLAM | QT, in three blocks of 20 with every load of a block before its first store (load / load /
                // store per entry made each of the 60 loads wait for the store before it, ~1 us each)
#pragma unroll
                for (int b0 = 0; b0 < 60; b0 += 20) {
                    double q_[20], n_[20];
#pragma unroll
                    for (int i = 0; i < 20; i++) { q_[i] = r1[O_QPI + b0 + i]; n_[i] = r5[b0 + i]; }
#pragma unroll
                    for (int i = 0; i < 20; i++) n_[i] += alpha * (q_[i] - n_[i]);
#pragma unroll
                    for (int i = 0; i < 20; i++) r5[b0 + i] = n_[i];
                }
            }
        }
#ifdef MPCB_SPROF_LIN
        __builtin_amdgcn_s_waitcnt(0);
        const double tl1 = wclock();
        if (lane == 0) { g_ssm.w.state[32 + 0] += tl1 - tl0; }
#endif
        if (k < N) {
            double rl[10];                                             // r | Y; the Jacobian goes straight to the record
            task_lin<true>(rb, P, xx, xx + 6, rl, r2);
#pragma unroll
            for (int i = 0; i < NTASK; i++) rl[O_Y + i] = P.w_task[i] * rl[O_R + i];
#pragma unroll
            for (int i = 0; i < 10; i++) r2[i] = rl[i];
        } else {
#pragma unroll
            for (int i = 0; i < 10; i++) r2[i] = 0.0;
#pragma unroll
            for (int i = O_GQ; i < W2_LIN; i++) r2[i] = 0.0;
        }
#ifdef MPCB_SPROF_LIN
        const double tl2 = wclock();
        __builtin_amdgcn_s_waitcnt(0);
        const double tl3 = wclock();
        if (lane == 0) { g_ssm.w.state[32 + 1] += tl2 - tl1; g_ssm.w.state[32 + 2] += tl3 - tl2; }
#endif
    }
}

// Dynamics defect of the NLP iterate, cost = sum_k dt/2 r'Wr (acados get_cost()) and acados' ocp_nlp_res_compute
// inf-norms [stat, eq, ineq, comp] with the QP multipliers (SQP_RTI).  Forward sweep, one row of lookahead (x_{k+1}).
//   in : G1 row | G2 [R..GV]      out : G2 BD
// SQPM: the residuals use the NLP multipliers of G5 (full SQP) instead of the QP's.
template <bool SQPM>
SE_PASS double nlp_res_pass(double *res4)
{
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor);
    const SWs w = sm.w;
    constexpr int I_L = 96, I_5 = 156;
    constexpr int ITEMS = SQPM ? 108 : 78;
    constexpr int MO_PI = SQPM ? I_5 + O_NPI : O_QPI, MO_LAM = SQPM ? I_5 + O_NLAM : O_QLAM, MO_T = SQPM ? I_5 + O_NT : O_QT;
    Bundle<2, ITEMS> bin;
    Bundle<1, 6> bout;
    {
        if (SQPM) { const Seg si[3] = {segd(w.G1, w.ld, 0, W1), segd(w.G2, w.ld, 0, W2_LIN), segd(w.G5, w.ld, 0, 60)}; bin.setup(si, lane); }
        else { const Seg si[2] = {segd(w.G1, w.ld, 0, W1), segd(w.G2, w.ld, 0, W2_LIN)}; bin.setup(si, lane); }
        const Seg so[1] = {segd(w.G2, w.ld, O_BD, 12)};
        bout.setup(so, lane);
        bout.seek(0, 1);
    }
    double csum = 0.0, a_s = 0, a_e = 0, a_i = 0, a_c = 0;
    // per-lane parameters in registers for the whole sweep (see residual_pass)
    const int jd = lane < 12 ? lane % 6 : 0;                                  // lanes 0..11: dynamics row / cost share
    const double d_a = lane < 6 ? P.a12[jd] : P.a22[jd], d_b = lane < 6 ? P.b1[jd] : P.b2[jd], d_cq = P.cq[jd];
    const double d_wt = P.w_task[lane >= 6 && lane < 6 + NTASK ? lane - 6 : 0];
    const int ci = lane >= 16 && lane < 34 ? lane - 16 : 0, cls = ci / 6, cj = ci - cls * 6;   // lanes 16..33: stationarity row
    const bool c_lo = cls < 2 && bnd_lo(P, ci < NB ? ci : 0) > -BOUND_INF, c_hi = cls < 2 && bnd_hi(P, ci < NB ? ci : 0) < BOUND_INF;
    const double cb_lo = bnd_lo(P, ci < NB ? ci : 0), cb_hi = bnd_hi(P, ci < NB ? ci : 0);
    const double k_dt = P.dt, k_2wu = 2.0 * P.w_u, k_wq = P.w_qddot, k_c2 = P.w_qddot * P.cq[cj] * P.cq[cj];
    const double k_p1 = cls == 0 ? P.b1[cj] : P.a12[cj], k_p2 = cls == 0 ? P.b2[cj] : P.a22[cj];
    const double xh = lane >= 48 && lane < 60 ? sm.xhat[lane - 48] : 0.0;
    if (lane < NX) sm.vec[0][lane] = 0.0;
    sweep<ITEMS, 1, false>(bin, N, lane, [&](int k, double *cur, double *nxt, double *) {
        double *o = sm.out[k & 1];
        const double *r1 = cur, *r2 = cur + I_L;
        if (lane < 12) {
            // dynamics defect (prediction_model.py:317-320) and this stage's share of the cost
            double v = 0.0;
            if (k < N) {
                const int j = jd;
                const double xq = r1[O_X + j], xv = r1[O_X + 6 + j], uj = r1[O_U + j];
                v = lane < 6 ? (xq + d_a * xv + d_b * uj) - nxt[O_X + j] : (d_a * xv + d_b * uj) - nxt[O_X + 6 + j];
                a_e = fmax(a_e, fabs(v));
                if (lane < 6) {
                    const double qdd = d_cq * (uj - xv);
                    csum += 0.5 * k_dt * (k_2wu * uj * uj + k_wq * qdd * qdd);
                } else if (lane < 6 + NTASK) {
                    const double r = r2[O_R + (lane - 6)];
                    csum += 0.5 * k_dt * d_wt * r * r;
                }
            }
            o[lane] = v;
        } else if (lane >= 16 && lane < 34 && res4) {
            const int j = cj;
            const double *pk = cur + MO_PI, *pm = sm.vec[k & 1];
            // stationarity of the NLP at the iterate (mpc_core.h stat_cls without delta), same operation order
            double v = 0.0;
            if (cls == 0) {
                if (k < N) {
                    const double uj = r1[O_U + j], vj = r1[O_X + 6 + j];
                    v = k_dt * (k_2wu * uj + k_c2 * (uj - vj));
                    v += k_p1 * pk[j] + k_p2 * pk[6 + j];
                }
            } else if (cls == 1) {
                if (k > 0) {
                    if (k < N) {
                        double s_ = 0.0;
#pragma unroll
                        for (int i = 0; i < NTASK; i++) s_ += r2[O_GQ + i * 6 + j] * r2[O_Y + i];
                        v = k_dt * s_ + pk[j];
                    }
                    v -= pm[j];
                }
            } else {
                if (k > 0) {
                    if (k < N) {
                        const double uj = r1[O_U + j], vj = r1[O_X + 6 + j];
                        v = k_dt * (r2[O_GV + j] * r2[O_Y + 4] + k_c2 * (vj - uj));
                        v += k_p1 * pk[j] + k_p2 * pk[6 + j];
                    }
                    v -= pm[6 + j];
                }
            }
            const bool hc = cls == 0 ? k < N : (cls == 1 && k >= 1 && k < N);
            if (hc) {
                const double curv = r1[ci < 6 ? O_U + ci : O_X + ci - 6];
                const double *lam = cur + MO_LAM, *tt = cur + MO_T;
                if (c_lo) {
                    v -= lam[ci];
                    a_i = fmax(a_i, fabs((cb_lo - curv) + tt[ci]));
                    a_c = fmax(a_c, fabs(lam[ci] * tt[ci]));
                }
                if (c_hi) {
                    v += lam[12 + ci];
                    a_i = fmax(a_i, fabs((curv - cb_hi) + tt[12 + ci]));
                    a_c = fmax(a_c, fabs(lam[12 + ci] * tt[12 + ci]));
                }
            }
            if (ci >= 6 && k == 0) v = 0.0;
            a_s = fmax(a_s, fabs(v));
        } else if (lane >= 48 && lane < 60) {
            if (k == 0 && res4) a_i = fmax(a_i, fabs(xh - r1[O_X + lane - 48]));   // lbx_0 = ubx_0 = x_hat
            sm.vec[(k + 1) & 1][lane - 48] = cur[MO_PI + lane - 48];
        }
        fence();
        store_out(bout, o, lane);
    });
    const double cost = wsum(csum);
    if (res4) { res4[0] = wmax(a_s); res4[1] = wmax(a_e); res4[2] = wmax(a_i); res4[3] = wmax(a_c); }
    return cost;
}

// =============================================================================================== SQP line search
// L1 merit function at the trial point (X,U) + alpha (dX,dU) (acados ocp_nlp_evaluate_merit_fun restated;
// mpc_core.h merit_pass).  lane <-> stage, straight from HBM.  With `update_weights` the merit weights (G5 MW) are
// first refreshed from the QP multipliers by Leineweber's rule.
SE_PASS double merit_pass(double alpha, bool update_weights, int sqp_iter)
{
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const Robot &rb = robot_in_lds();
    const int lane = threadIdx.x;
    const int N = uni(sm.n_hor);
    const SWs w = sm.w;
    double acc = 0.0;
    for (int k0 = 0; k0 <= N; k0 += WAVE) {
#ifndef MPCB_NO_LICM_BLOCK
        asm volatile("" ::: "memory");     // (parameters are read where they are used, not hoisted out of the loop and spilled: mpc_core.h merit_pass)
#endif
        const int k = k0 + lane;
        if (k > N) continue;
        MPC_GLOBAL double *r1 = (MPC_GLOBAL double *)((char *)w.G1 + (size_t)k * w.ld);
        MPC_GLOBAL double *rn = (MPC_GLOBAL double *)((char *)w.G1 + (size_t)(k < N ? k + 1 : k) * w.ld);
        MPC_GLOBAL double *mw = (MPC_GLOBAL double *)((char *)w.G5 + (size_t)k * w.ld) + O_MW;
        // the 36 merit weights of the stage in registers; refreshed first when asked (Leineweber's rule) -- every load BEFORE the first
        // store: load / load / store per weight made each of the 36 loads wait for the store before it (~1 us each, see lin_pass)
        double mwv[36];
        if (update_weights) {
            double a_[36];
#pragma unroll
            for (int i = 0; i < 36; i++) a_[i] = i < 12 ? fabs(r1[O_QPI + i]) : fabs(r1[O_QLAM + i - 12]);
            if (sqp_iter == 0) {
#pragma unroll
                for (int i = 0; i < 36; i++) mwv[i] = a_[i];
            } else {
#pragma unroll
                for (int i = 0; i < 36; i++) mwv[i] = mw[i];
#pragma unroll
                for (int i = 0; i < 36; i++) mwv[i] = fmax(a_[i], 0.5 * (mwv[i] + a_[i]));
            }
#pragma unroll
            for (int i = 0; i < 36; i++) mw[i] = mwv[i];
        } else {
#pragma unroll
            for (int i = 0; i < 36; i++) mwv[i] = mw[i];
        }
        double xx[12], uu[6], rec[8];
#pragma unroll
        for (int i = 0; i < 12; i++) xx[i] = r1[O_X + i] + alpha * r1[O_QW + 6 + i];
        if (k < N) {
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
                acc += mwv[j] * fabs((xx[j] + P.a12[j] * xx[6 + j] + P.b1[j] * uu[j]) - xnq);
                acc += mwv[6 + j] * fabs((P.a22[j] * xx[6 + j] + P.b2[j] * uu[j]) - xnv);
                const double vl = P.umin[j] - uu[j], vu = uu[j] - P.umax[j];
                acc += mwv[12 + j] * fmax(vl, 0.0) + mwv[24 + j] * fmax(vu, 0.0);
                const double ql = P.qmin[j] - xx[j], qu = xx[j] - P.qmax[j], on = k >= 1 ? 1.0 : 0.0;
                acc += on * (mwv[18 + j] * fmax(ql, 0.0) + mwv[30 + j] * fmax(qu, 0.0));
            }
            acc += 0.5 * P.dt * s;
        }
        if (k == 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) acc += w.state[13 + i] * fabs(sm.xhat[i] - xx[i]);
        }
    }
    return wsum(acc);
}

// Merit weight of the eliminated x_0 constraint: |stage-0 stationarity of the QP wrt x_0| (mpc_core.h update_x0_weights)
SE_PASS void update_x0_weights(int sqp_iter)
{
    SSmem &sm = g_ssm;
    const InstParams &P = sm.P;
    const int lane = threadIdx.x;
    const SWs w = sm.w;
    if (lane < NX) {
        const double *r1 = w.G1, *r2 = w.G2;      // stage 0 records (y holds W(r + G delta) of the last residual pass)
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
        double *mwp = &w.state[13 + lane];
        *mwp = sqp_iter == 0 ? a : fmax(a, 0.5 * (*mwp + a));
    }
    __builtin_amdgcn_s_waitcnt(0);
    fence();
}

// MERIT_BACKTRACKING (trajectory_optimizer.py:68; acados alpha_reduction 0.7, alpha_min 0.05)
SE_DEV double line_search(int sqp_iter)
{
    update_x0_weights(sqp_iter);
    const double m0 = unid(merit_pass(0.0, true, sqp_iter));
    __builtin_amdgcn_s_waitcnt(0);
    fence();
    double alpha = 1.0;
    while (alpha >= 0.05) {
        if (uni(unid(merit_pass(alpha, false, sqp_iter)) < m0 ? 1 : 0)) break;
        alpha *= 0.7;
    }
    return alpha;
}

// =============================================================================================== closed loop
SE_DEV void log_flush(const Outputs &out, int inst, int T1, int c_lo, int c_hi)
{
    SSmem &sm = g_ssm;
    const int lane = threadIdx.x;
    for (int e = lane; e < LOG_ROWS * SLOGB; e += WAVE) {
        const int row = e / SLOGB, cc = (c_hi & ~(SLOGB - 1)) + (e & (SLOGB - 1));
        if (cc < c_lo || cc > c_hi) continue;
        double *dst = row < 12 ? out.z + ((size_t)inst * 12 + row) * T1
                    : row < 18 ? out.u + ((size_t)inst * 6 + (row - 12)) * T1
                    : row < 30 ? out.ee_pose + ((size_t)inst * 12 + (row - 18)) * T1
                    : row < 33 ? out.ee_rpy + ((size_t)inst * 3 + (row - 30)) * T1
                    : row < 39 ? out.ee_vel + ((size_t)inst * 6 + (row - 33)) * T1
                               : out.errors + ((size_t)inst * 7 + (row - 39)) * T1;
        dst[cc] = sm.logbuf[row][e & (SLOGB - 1)];
    }
    fence();
}

SE_PASS int log_state(const Outputs &out, int inst, int T1, int col, int log_lo)
{
    SSmem &sm = g_ssm;
    const int lane = threadIdx.x;
    const Robot &rb = robot_in_lds();
    if (lane == 0) {
        double z[12];
#pragma unroll
        for (int i = 0; i < 12; i++) z[i] = sm.xhat[i];
        plant_log(rb, z, sm.logv);
        task_errors(sm.P, rb, sm.logv, sm.logv + 15, sm.logv + 36);
    }
    fence();
    if (lane < LOG_ROWS) {
        const double v = lane < 12 ? sm.xhat[lane]
                       : lane < 18 ? sm.u0[lane - 12]
                       : lane < 30 ? sm.logv[lane - 18]
                       : lane < 33 ? sm.logv[12 + (lane - 30)]
                       : lane < 39 ? sm.logv[15 + (lane - 33)]
                                   : sm.logv[36 + (lane - 39)];
        sm.logbuf[lane][col & (SLOGB - 1)] = v;
    }
    fence();
    col = uni(col); log_lo = uni(log_lo);
    if ((col & (SLOGB - 1)) == SLOGB - 1) { log_flush(out, inst, T1, log_lo, col); log_lo = col + 1; }
    return log_lo;
}

// Simulator.run (simulator.py:199-241) for steps [step0, step1) of one simulation, SQP_RTI.
template <class FT>
SE_DEV void rollout(const Problem &pb, const InstParams *params, const Robot *rbp, double *ws_base, size_t ws_stride,
                    const Outputs &out, int inst, int step0, int step1)
{
    SSmem &sm = g_ssm;
    const int lane = threadIdx.x;
    // ragged batch: this simulation's own horizon (the workspace stride is sized for the longest)
    const double nh = params[inst].n_hor;
    const int N = uni(nh > 0.0 ? (int)nh : pb.N), Nsim = pb.Nsim, T1 = Nsim + 1;
    {
        const double *ps = reinterpret_cast<const double *>(params + inst);
        double *pd = reinterpret_cast<double *>(&sm.P);
        for (int e = lane; e < (int)(sizeof(InstParams) / sizeof(double)); e += WAVE) pd[e] = ps[e];
        if (lane == 0) {
            sm.rbp = rbp;
            SWs ws = sws_carve<FT>(ws_base + (size_t)inst * ws_stride, N, pb.solver_type == 0);
            ws.state = ws_base + (size_t)inst * ws_stride + (ws_stride - STATE_DOUBLES);   // at the end of the stride whatever this simulation's horizon
            sm.w = ws;
            sm.n_hor = N;
        }
    }
    fence();
    const InstParams &P = sm.P;
    const SWs w = sm.w;
    const size_t sbase = (size_t)inst * Nsim;
    bool lin_valid = false;
    bool res_pending = false;          // SQP_RTI: cost / residual norms of the previous step are formed by this step's first pass
    double lin_cost = 0.0;
    int fast[2] = {0, 0};              // fast path: QPs left before the next attempt, length of the current suspension
    int cur = 0;                       // SQP_RTI: slot of the stage records that holds the QP iterate (fast_commit comment)
    int log_lo = step0 == 0 ? 0 : step0 + 1;
    if (step0 == 0) {
        // acados initial guess: x_k = x0, u_k = 0, all multipliers / QP memory 0
        const size_t tot = sws_doubles_per_instance<FT>(N, pb.solver_type == 0) - STATE_DOUBLES;
        for (size_t e = lane; e < tot; e += WAVE) w.G1[e] = 0.0;      // G1 is the workspace base
        if (lane < STATE_DOUBLES) w.state[lane] = 0.0;
        fence();
        for (int e = lane; e < (N + 1) * NX; e += WAVE) {
            const int k = e / NX, i = e - k * NX;
            w.G1[(size_t)k * (w.ld / 8) + O_X + i] = i < 6 ? P.q0[i] : P.qdot0[i - 6];
        }
        if (lane < NX) sm.xhat[lane] = lane < 6 ? P.q0[lane] : P.qdot0[lane - 6];
        if (lane < NU) sm.u0[lane] = P.qdot0[lane];                   // u[:,0] = qdot_0 (simulator.py:81)
        __builtin_amdgcn_s_waitcnt(0);                                // the initial iterate is in memory before the first pass reads it
        fence();
        log_lo = uni(log_state(out, inst, T1, 0, log_lo));
    } else {
        if (lane < NX) sm.xhat[lane] = w.state[lane];
        lin_cost = unid(w.state[12]);
        lin_valid = uni(w.state[25] != 0.0 ? 1 : 0) != 0;
        fast[0] = uni((int)w.state[26]); fast[1] = uni((int)w.state[27]); cur = uni((int)w.state[28]);
        fence();
    }
    for (int i = step0; i < step1; i++) {
        int qp_iter = 0, status = 0, sqp_iter = 1;
        double res4[4] = {0, 0, 0, 0};
        double cost = lin_cost;
        const double t0 = wclock();
        if (pb.solver_type == 1) {
            // SQP_RTI: one linearisation, one QP, full step (mpc_core.h nlp_step)
#ifdef MPCB_STREAM_SEQ_RES
            if (!lin_valid) { lin_pass(0.0, false); __builtin_amdgcn_s_waitcnt(0); fence(); lin_cost = unid(nlp_res_pass<false>(nullptr)); }
#else
            if (!lin_valid) {
                lin_pass(0.0, false); __builtin_amdgcn_s_waitcnt(0); fence();
                double o5[5];
                rti_items<true, false>(0, o5, cur);
                lin_cost = unid(o5[0]);
            }
#endif
            double nlp_prev[5];
            const int qs = ipm_solve<FT>(pb.qp_iter_max, &qp_iter, fast, res_pending ? nlp_prev : nullptr, &cur);
            if (res_pending) {
                // cost and residual norms of step i-1, evaluated by this step's first pass
                if (lane == 8) out.cost[sbase + i - 1] = nlp_prev[0];
                if (lane >= 12 && lane < 16) out.residuals[(sbase + i - 1) * 4 + (lane - 12)] =
                    lane == 12 ? nlp_prev[1] : (lane == 13 ? nlp_prev[2] : (lane == 14 ? nlp_prev[3] : nlp_prev[4]));
                res_pending = false;
            }
#ifdef MPCB_SPROF
            if (lane == 0) { w.state[32 + 5] += wclock() - t0; }
#endif
            const bool ok = qs == 0 || qs == 1;
            if (!ok) status = 4;                                       // ACADOS_QP_FAILURE, iterate untouched
            __builtin_amdgcn_s_waitcnt(0);
            SPROF_T0(tl);
#ifdef MPCB_STREAM_SEQ_RES
            lin_pass(1.0, ok);
#else
            lin_pass(1.0, ok, false, cur);
#endif
            __builtin_amdgcn_s_waitcnt(0);                             // the records written lane by lane are complete before they are streamed
            fence();
            SPROF_ADD(8, tl);
            SPROF_T0(tn);
            if (i + 1 < step1) {
                // the next step's first pass streams the same records: it evaluates the defect, cost and residuals there
                res_pending = true;
                if (lane < NX) sm.vec[3][lane] = sm.xhat[lane];        // the x_hat this QP was solved for
            } else {
#ifdef MPCB_STREAM_SEQ_RES
                cost = unid(nlp_res_pass<false>(res4));                // last step of this launch / work item
#else
                double o5[5];
                rti_items<true, false>(0, o5, cur);                    // last step of this launch / work item
                cost = unid(o5[0]); res4[0] = o5[1]; res4[1] = o5[2]; res4[2] = o5[3]; res4[3] = o5[4];
#endif
            }
            SPROF_ADD(9, tn);
            lin_valid = true;
        } else {
            // full SQP (acados ocp_nlp_sqp restated, mpc_core.h nlp_step): linearise -> residuals / convergence test -> QP ->
            // merit backtracking -> update; sqp_iter counts QPs
            const double tol = P.tol, tol_eq = P.tol_eq, tol_in = P.tol_ineq, tol_co = P.tol_comp;
            status = 2;                                                // ACADOS_MAXITER unless decided otherwise
            double alpha = 0.0;
            bool pending = false;                                      // a step (alpha) waits to be applied by the next linearisation
            for (sqp_iter = 0; sqp_iter < pb.max_iter; sqp_iter++) {
                if (pending || !lin_valid || sqp_iter == 0) {
                    __builtin_amdgcn_s_waitcnt(0);
                    lin_pass(alpha, pending, true);
                    __builtin_amdgcn_s_waitcnt(0);
                    fence();
#ifdef MPCB_STREAM_SEQ_RES
                    cost = unid(nlp_res_pass<true>(res4));
                    res4[0] = unid(res4[0]); res4[1] = unid(res4[1]); res4[2] = unid(res4[2]); res4[3] = unid(res4[3]);
#else
                    double o5[5];
                    rti_items<true, false, true>(0, o5);
                    cost = unid(o5[0]); res4[0] = unid(o5[1]); res4[1] = unid(o5[2]); res4[2] = unid(o5[3]); res4[3] = unid(o5[4]);
#endif
                    pending = false;
                    lin_valid = true;
                }
                if (res4[0] < tol && res4[1] < tol_eq && res4[2] < tol_in && res4[3] < tol_co) { status = 0; break; }
                if (res4[0] != res4[0] || cost != cost) { status = 1; break; }
                int it = 0;
                const int qs = ipm_solve<FT>(pb.qp_iter_max, &it, fast);
                qp_iter += it;
                if (qs != 0 && qs != 1) { status = 4; break; }
                __builtin_amdgcn_s_waitcnt(0);
                fence();
                alpha = pb.fixed_step ? 1.0 : line_search(sqp_iter);
                pending = true;
            }
            if (pending) {   // max-iter exit: apply the last step; the residuals of the last check stay
                __builtin_amdgcn_s_waitcnt(0);
                lin_pass(alpha, true, true);
                __builtin_amdgcn_s_waitcnt(0);
                fence();
#ifdef MPCB_STREAM_SEQ_RES
                cost = unid(nlp_res_pass<true>(nullptr));
#else
                double o5[5];
                rti_items<true, false, true>(0, o5);
                cost = unid(o5[0]);
#endif
                lin_valid = true;
            }
        }
        lin_cost = cost;
        __builtin_amdgcn_s_waitcnt(0);
        fence();
        const double t1 = wclock();
#ifdef MPCB_SPROF
        if (lane == 0) { w.state[32 + 6] += t1 - t0; }
#endif
        // u = solver.get(0,'u'); plant step (simulation_model.py:93-117)
        if (lane < 6) {
            const int j = lane;
            const double u = w.G1[O_U + j], wc = P.wcv[j], dt = P.dt;
            const double q = sm.xhat[j], v = sm.xhat[6 + j];
            const int integ = (int)P.integ;
            const double k1q = v, k1v = -wc * v + wc * u;
            const double v2 = v + 0.5 * dt * k1v;
            const double k2q = v2, k2v = -wc * v2 + wc * u;
            double qn, vn;
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
            sm.logv[24 + j] = qn;
            sm.logv[30 + j] = vn;
            sm.u0[j] = u;
        }
        if (lane == 8) {
            out.status[sbase + i] = status;
            out.sqp_iter[sbase + i] = sqp_iter;
            out.qp_iter[sbase + i] = qp_iter;
            if (!res_pending) out.cost[sbase + i] = cost;
            out.solver_time[sbase + i] = t1 - t0;
        }
        if (!res_pending && lane >= 12 && lane < 16) out.residuals[(sbase + i) * 4 + (lane - 12)] =
            lane == 12 ? res4[0] : (lane == 13 ? res4[1] : (lane == 14 ? res4[2] : res4[3]));
        fence();
        if (lane < NX) sm.xhat[lane] = sm.logv[24 + lane];
        fence();
        log_lo = uni(log_state(out, inst, T1, i + 1, log_lo));
        const double t2 = wclock();
        if (lane == 0) out.plant_time[sbase + i] = t2 - t1;
#ifdef MPCB_SPROF
        if (lane == 0) { w.state[32 + 14] += t2 - t1; }
#endif
    }
    if (log_lo <= step1) log_flush(out, inst, T1, log_lo, step1);
    if (lane < NX) w.state[lane] = sm.xhat[lane];
    if (lane == 12) { w.state[12] = lin_cost; w.state[25] = lin_valid ? 1.0 : 0.0; w.state[26] = fast[0]; w.state[27] = fast[1]; w.state[28] = cur; }
}
#endif  // __HIP_DEVICE_COMPILE__

}  // namespace se
}  // namespace mpcb
