// papteam_opticalflow_amd/csrc/sor.hip -- the SOR sweeps of OpticalFlow::SmoothFlowSOR
// (/root/reference/Code/Serial/src/OpticalFlow.cpp:451-505), the hot loop of the whole path.
//
// Per cell and sweep the reference computes, in place and in sweep -> row -> column order,
//     s1 = wL*duL + wR*duR + wU*duU + wD*duD          (terms only for existing neighbours, that order)
//     s2 = same with dv
//     s1 *= -alpha ; s2 *= -alpha
//     s1 += imdxy*dv          ; du = (1-omega)*du + a1*(b1 - s1)      a1 = omega/(imdx2+alpha*.05+coeff)
//     s2 += imdxy*du(new)     ; dv = (1-omega)*dv + a2*(b2 - s2)
// with wL = phi(left cell), wR = wD = phi(this cell), wU = phi(upper cell).  a1/a2 are constant over the
// sweeps and precomputed by the assembly kernel (kernels.hip), so a cell-update reads 8 and writes 2
// doubles = 80 B of algorithmic traffic and ~30 fp64 operations: HBM-bound, no MFMA.
//
// Three orderings are provided (papof.h):
//
// PAPOF_SOR_EXACT -- bit-compatible with the reference.  In-place Gauss-Seidel order is kept EXACTLY
//   by a wavefront-parallel hyperplane schedule: cell (i, j, sweep k) may run once
//   (i, j-1, k), (i-1, j, k), (i, j+1, k-1), (i+1, j, k-1) are done.
//     * TASK = (band b, sweep k) = one wavefront.  Bands are TIME-SKEWED: at sweep k band b owns the 62 rows
//       62b-k .. 62b-k+61 (lanes 1..62); lanes 0 / 63 are ghost lanes for the row above / below.  Because the bands
//       climb one row per sweep, the row below a band -- needed with its previous-sweep value -- belonged to the SAME
//       band one sweep earlier: a task depends only on (b, k-1) and (b-1, k), never on the band below, which removes
//       one of the two hand-offs from the sweep-to-sweep critical path;
//     * at STEP s lane l works on column j = s - l (the lane above runs one column ahead), NS = W + 63 steps;
//     * the six coefficient operands live in three paired, globally skewed planes of 16-byte cells,
//       (i, j) -> (i+j+qt)*hp + i+rt (common.h): at step s the 64 lanes of a task read 64 consecutive cells -- one
//       contiguous 1-KiB access per plane, ghost lanes included (a1 = a2 = 0 and omega-1 -> 1 make a ghost's update
//       a pass-through).  Non-cells are 0.0, so image borders and padding need no predicates: the reference's
//       conditional terms become +-0 added in the same order;
//     * the unknowns live in BANDED PING-PONG planes D[k & 1][position][band][64 cells]: at step s task (b, k) stores
//       all 64 lanes (ghosts store their pass-through) to the aligned 1-KiB block D[k&1][s+1][b], which no other task
//       of the sweep touches (full-line write-through stores, no sharing of lines between writers), and reads its
//       old values from D[(k-1)&1][.][b] shifted by one cell (lane l <- cell l-1: the band climbed one row, and
//       cell 0 is the previous sweep's pass-through of the row above); ghost lane 0 reads the NEW value of the row
//       above from D[k&1][s+64][b-1][62].  Re-using the parity buffer two sweeps later is a write-after-read on the
//       block's cell 62 w.r.t. the band below, hence a third, practically never binding dependency (below);
//     * left-new is the lane's own previous result, up-new the previous result of lane l-1, down-old the pending
//       centre of lane l+1 (DPP wave shifts); right-old is loaded and becomes the next centre;
//     * every load is issued R = 8 steps before its use (register software pipeline; 5 memory operations per step
//       keep R <= 12 within gfx9's 6-bit vmcnt);
//     * all nb * n_sor tasks are launched at once (one 64-thread workgroup each; a few hundred waves, all
//       co-resident) and pipeline through per-task progress counters (one 128-byte line each).  Before issuing the
//       loads of steps < e a task waits for
//           prog[k-1][b]   >= min(NS, e)          own band, previous sweep (its step s wrote what our step s reads)
//           prog[k][b-1]   >= min(NS, e + 63)     band above, this sweep (ghost lane 0)
//           prog[k-2][b+1] >= min(NS, e - 62)     band below, two sweeps ago (write-after-read of our block)
//       A waiter only ever waits on lower block indices.  tests/sim_sor_wave.py executes this exact dataflow under
//       a random scheduler and is checked bit-for-bit against the oracle on the CPU;
//     * cross-workgroup visibility follows MI355X guide G16/R1: du/dv are stored write-through (`sc1`) and read with
//       `sc1` loads; a counter only ever advertises steps whose stores are PROVEN complete -- by marker loads that
//       retire (vmcnt is in order) after them, twice per R steps without draining the pipeline, and by a final
//       `s_waitcnt vmcnt(0)`; consumers poll relaxed, one iteration ahead.  Counters are zeroed by a memset node
//       before every launch, every spin is bounded, and a timeout raises an abort word (PAPOF_ETIMEOUT).
//
// PAPOF_SOR_REDBLACK / PAPOF_SOR_JACOBI -- throughput and correctness-gate modes whose results differ from the
//   reference's order.  One LDS-tiled, temporally blocked kernel (k_sor_blocked, documented where it is
//   defined): a workgroup keeps a region of the row-major planes on chip for ~10 half-sweeps per launch.  The first
//   implementation -- one launch per half-sweep / sweep straight on the planes -- is kept as a cross-check (PAPOF_RB_NAIVE).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"

namespace papof {

namespace {

// Tuning constants of the hand-off protocol, overridable at compile time for A/B builds (tools/ab + PAPOF_LIB; measured on
// one box, profiles/r01_s3_ab_variants.txt): steps after which a marker is consumed at pipeline depth 6 / 8 (4 / 5: the
// hand-off-bound levels gain 2-3 % over 3 / 5 and 4 / 4), the same for the fused kernel (3; 4 costs level 0 4 %), and
// the depth from which the progress poll is issued in mid-iteration (8; at depth 6 it costs 12 %).
#ifndef PAPOF_V_DM6
#define PAPOF_V_DM6 4
#endif
#ifndef PAPOF_V_DM8
#define PAPOF_V_DM8 5
#endif
#ifndef PAPOF_V_FDM6
#define PAPOF_V_FDM6 3
#endif
#ifndef PAPOF_V_FDM8
#define PAPOF_V_FDM8 4  // ... at depth 8, the default since round 2 (in-pair A/B: 4 is 0.05 ms per 1080p pair faster than 5, 6 slower)
#endif
#ifndef PAPOF_V_FMIDPOLL
#define PAPOF_V_FMIDPOLL 10  // fused kernel: depth from which the progress poll is issued in mid-iteration
#endif
#ifndef PAPOF_V_FSELECTIVE
#define PAPOF_V_FSELECTIVE false  // fused kernel: poll only the counter that is still missing while waiting
#endif
#ifndef PAPOF_V_MIDPOLL
#define PAPOF_V_MIDPOLL 8
#endif
#ifndef PAPOF_V_SLEEP
#define PAPOF_V_SLEEP 1  // s_sleep argument (64 clocks each) between two polls of a waiting task
#endif
constexpr int kProgStride = 32;  // unsigneds between progress counters = one 128-byte cache line each
constexpr unsigned kSpinLimit = 4u << 20;  // bounded wait: ~4M polls (seconds), then abort

struct ExactArgs {
    const double *phi, *xy, *a1, *a2, *b1, *b2;
    double *du, *dv;
    unsigned* prog;   // [n_sor][nb]
    unsigned* abort;  // one word
    int H, W, nb, ns, hp, npos, qt, rt, npos_d, n_sor;
    int xcd_affine;
    int k0;  // first sweep (k_sor_exact) / pair (k_sor_fused) of THIS launch: a solve whose tasks exceed what the chip
             // keeps resident is issued as consecutive launches over ranges of sweeps (sor_solve)
    int b0, nbl;  // bands b0 .. b0 + nbl - 1 are THIS launch's (a strip of the plane, sor_solve_bands: the strips of a solve
                  // are launched on different streams and meet through the progress counters); a whole solve: 0, nb
    double nalpha, om1;
    unsigned long long* dbg;  // diagnostics (PAPOF_SOR_DBG): per task 8 time stamps (s_memrealtime, 100 MHz), else null
    unsigned long long* stamp;  // phase stamp (flow_internal.h: PhaseClock): block 0 writes the 100 MHz clock on entry
    // SPLIT launches (one solve cut into ranges of bands that live in DIFFERENT handles / address spaces, tiles.hip:
    // bands_flow).  Every rank keeps the full-size banded (du, dv) planes and counters of the global layout and runs the bands
    // b0 .. b0 + nbl - 1.  The one edge that crosses a cut -- ghost lane 0 of band b0 reads, per step, the new value of the
    // row above, which is lane 62 of band b0 - 1 on the rank above -- goes through an INBOX inside the reader's own planes:
    // the slot of block b0 - 1 (never computed locally), cell k >> 1 of parity k & 1 -- one cell per sweep, so nothing is
    // ever reused within a solve and no write-after-read edge crosses the cut (n_sor <= 128).  The rank above stores that
    // cell into the reader's planes (peer_du) beside its own store, with the same position arithmetic, and publishes its
    // progress into the reader's counters (peer_prog) beside its own.
    double* peer_du;      // (du, dv) planes of the rank that runs band b0 + nbl (same layout), or null
    unsigned* peer_prog;  // its progress counters of this solve
    int top_cut;          // band b0 has its upper neighbour on another rank: read the inbox cell
    int bot_cut;          // band b0 + nbl - 1 has its lower neighbour on another rank: also write that rank's inbox / counter
    // BATCHED solves (api.hip: flow_batch): blockIdx.y is the frame pair; its operands are the first pair's advanced by these
    // strides -- doubles between the pairs' coefficient planes / (du, dv) planes, unsigneds between their counters.  0: one solve.
    size_t bs_coef, bs_d, bs_prog;
    int skip_dead;  // lanes whose image row lies outside [0, H) for the whole task neither load nor store (k_sor_exact, below)
};

__device__ __forceinline__ void stamp_now(unsigned long long* s) {
    if (s && (blockIdx.x | threadIdx.x) == 0u) *s = __builtin_amdgcn_s_memrealtime();
}

__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Cross-lane move of one fp64 value by one lane (the wave-edge lane receives an unspecified value that the
// ghost-lane scheme never uses).  DPP wave_shr:1 / wave_shl:1 (GFX9 full-wavefront shifts, dpp_ctrl 0x138 / 0x130)
// when the start-up probe verified their direction on this device, else ds_bpermute.
template <bool DPP>
__device__ __forceinline__ double from_above(double x) {  // lane l <- lane l-1
    if (DPP) {
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    return __shfl_up(x, 1);
}
template <bool DPP>
__device__ __forceinline__ double from_below(double x) {  // lane l <- lane l+1
    if (DPP) {
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true);
        int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    return __shfl_down(x, 1);
}

__global__ void k_xlane_probe(int* out) {  // out[0..63] = from_above, out[64..127] = from_below (DPP forms)
    const int lane = threadIdx.x;
    const double v = (double)(lane + 1);
    out[lane] = (int)from_above<true>(v);
    out[64 + lane] = (int)from_below<true>(v);
}

// Diagnostic (PAPOF_PROBE=1 at handle creation): cost of the per-step arithmetic alone, one wave, no memory.
// out[0] = s_memtime ticks, out[1] = s_memrealtime ticks (100 MHz) for `n` steps.
__global__ __launch_bounds__(1024) void k_alu_probe(unsigned long long* out, double* sink, int n, double seed) {
    double duL = seed, dvL = seed * 0.5, phiL = 0.7, duC = 0.1, dvC = 0.2;
    const double phiC = 0.9, xy = 0.01, a1 = 0.3, a2 = 0.4, b1 = 0.001, b2 = 0.002, nalpha = -0.012, om1 = -0.8;
    double duR = 0.05 * seed, dvR = 0.06 * seed;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i++) {
        const double duU = from_above<true>(duL), dvU = from_above<true>(dvL), phiU = from_above<true>(phiL);
        const double duD = from_below<true>(duR), dvD = from_below<true>(dvR);
        double s1 = phiL * duL, s2 = phiL * dvL;
        s1 += phiC * duR;
        s2 += phiC * dvR;
        s1 += phiU * duU;
        s2 += phiU * dvU;
        s1 += phiC * duD;
        s2 += phiC * dvD;
        s1 *= nalpha;
        s2 *= nalpha;
        s1 += xy * dvC;
        const double duN = om1 * duC + a1 * (b1 - s1);
        s2 += xy * duN;
        const double dvN = om1 * dvC + a2 * (b2 - s2);
        duL = duN;
        dvL = dvN;
        duC = duR;
        dvC = dvR;
        duR = duR * 0.999;
        dvR = dvR * 0.999;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = r1 - r0;
    }
    sink[threadIdx.x] = duL + dvL;
}

// ---- 16-byte accesses through buffer descriptors (one SGPR quad per paired plane) -------------------------
// aux = 16 sets `sc1` (agent-coherent: write-through stores, L1-bypassing loads) on the du/dv plane.  Offsets
// beyond num_records read 0 / DROP the store: the kernel uses that to switch lanes off without touching EXEC.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct D2 {
    double x, y;
};
__device__ __forceinline__ D2 as_d2(u32x4 v) {
    D2 r;
    __builtin_memcpy(&r, &v, sizeof r);
    return r;
}
__device__ __forceinline__ u32x4 as_u4(double x, double y) {
    D2 d{x, y};
    u32x4 r;
    __builtin_memcpy(&r, &d, sizeof r);
    return r;
}
constexpr int kAuxPlain = 0, kAuxSc1 = 16;
constexpr unsigned kOob = 0x80000000u;  // byte offset beyond every plane (sor_solve enforces planes < 1 GiB) that
                                        // cannot wrap while it advances by 1 KiB per step

// Operands of R consecutive steps held in registers: slot t of the running iteration is consumed by step
// i*R + t and immediately refilled with the operands of step (i+1)*R + t, i.e. every global load is issued
// R steps before its use, so the latency of the write-through du/dv traffic and of the coefficient streams is
// hidden behind R steps of arithmetic.  A step costs 4 loads + 2 stores; R <= 9 keeps the operations in
// flight within the 6-bit vmcnt range (63) of gfx9.
template <int R>
struct Slots {
    u32x4 pa[R], pb[R], pc[R];  // (phi, xy) (a1, a2) (b1, b2) of this lane's cell at skew position s
    u32x4 pd[R];                // (du, dv) right-old of step s (the next centre)
};

struct Task {  // wave-uniform task constants (SGPRs)
    __amdgpu_buffer_rsrc_t ra, rb, rc, rd;  // descriptors of the four paired planes
    __amdgpu_buffer_rsrc_t rp;              // the progress counters (k_sor_exact / k_sor_fused)
    __amdgpu_buffer_rsrc_t rd2, rp2;        // SPLIT launches: (du, dv) planes and counters of the rank below the cut
};

// Publication of a task's progress WITHOUT a branch: every lane executes the store, the per-lane offset is out of
// range for all lanes but lane 0 (dropped).  An `if (lane == 0)` is a divergent branch, and one divergent branch inside
// the sweep loop makes the compiler structurise the whole loop, the scalar early exits included (see uni() below).
__device__ __forceinline__ void publish(const Task& T, unsigned lane_off, unsigned steps) {
    __builtin_amdgcn_raw_buffer_store_b32(steps, T.rp, lane_off, 0, 16 /* sc1 */);
}

// Per-lane ABSOLUTE byte offsets (constant VGPRs); the step index enters only through uniform soffsets
// (s * pos_c for the coefficient planes, s * pos_d for the unknowns).  kOob switches an access off (reads 0.0)
// without touching EXEC.
struct LaneOffs {
    unsigned pa;     // (phi, xy) of this lane's cell at step s (64 consecutive rows of one skew position)
    unsigned pbc;    // (a1, a2) and (b1, b2)                     [ghosts: off -> 0 -> pass-through]
    unsigned pd;     // (du, dv) right-old of step s: own block, previous parity, cell lane-1 [lane 0: block above]
    unsigned st;     // (du, dv) store of step s: own block, this parity, cell lane (ghosts store their pass-through)
    unsigned pos_c;  // bytes per coefficient skew position = hp * 16
    unsigned pos_d;  // bytes per (du, dv) position = nb * 1 KiB
    unsigned hs;     // SPLIT: store of step s into the inbox of the rank below (lane 62 of the band above a cut; else off)
};

template <int R, int t>
__device__ __forceinline__ void load_coef(const Task& T, const LaneOffs& L, int s, Slots<R>& c) {
    const unsigned off = (unsigned)s * L.pos_c;  // wave-uniform byte offsets -> soffset
    c.pa[t] = __builtin_amdgcn_raw_buffer_load_b128(T.ra, L.pa, off, kAuxPlain);
    c.pb[t] = __builtin_amdgcn_raw_buffer_load_b128(T.rb, L.pbc, off, kAuxPlain);
    c.pc[t] = __builtin_amdgcn_raw_buffer_load_b128(T.rc, L.pbc, off, kAuxPlain);
}
template <int R, int t>
__device__ __forceinline__ void load_unknowns(const Task& T, const LaneOffs& L, int s, Slots<R>& c) {
    c.pd[t] = __builtin_amdgcn_raw_buffer_load_b128(T.rd, L.pd, (unsigned)s * L.pos_d, kAuxSc1);
}
template <int R, int t>
__device__ __forceinline__ void load_slot(const Task& T, const LaneOffs& L, int s, Slots<R>& c) {
    load_coef<R, t>(T, L, s, c);
    load_unknowns<R, t>(T, L, s, c);
}

struct State {
    double duL, dvL, phiL, duC, dvC;
};

// A value that outlives the operand slot it came from is MOVED out of the slot's registers (an opaque v_mov the
// compiler cannot fold away).  Otherwise the slot's registers stay live across the refill, the refill lands in other
// registers, and the loop-carried slots need register copies at the top of every iteration -- which the compiler
// guards with `s_waitcnt vmcnt(0)`: a full drain of the software pipeline once per iteration.
// A 16-byte store reads its data registers over more than one cycle: a VALU write of those registers needs two wait
// states behind the store (gfx940+).  Nothing provides them here: the compiler's hazard recognizer does not look inside
// inline-asm statements (the moves of moved() below, which it may schedule and register-allocate straight behind a
// store), and it skips this hazard ALTOGETHER for buffer stores whose soffset is a register -- the form of every solver
// store -- so compiler-generated VALU writes are unguarded at these stores as well.  Found in round 2: with a second wave on
// the SIMD the store can slip a cycle, and the write then overwrites data not yet read (wrong quads of lanes, only under
// concurrency -- DESIGN.md 5.1).  Every store of the solver steps is therefore followed by store_data_guard(): two wait
// states, whatever comes next; tools/scan_store_hazard.py checks the generated code of every build (csrc/Makefile).
__device__ __forceinline__ void store_data_guard() { asm volatile("s_nop 1" ::: "memory"); }

__device__ __forceinline__ double moved(double x) {
    double y;
    asm volatile("v_mov_b64 %0, %1" : "=v"(y) : "v"(x));
    return y;
}

template <int R, int t, bool DPP, bool SPLIT = false>
__device__ __forceinline__ void step(const ExactArgs& A, const Task& T, const LaneOffs& L, double om1, int s,
                                     Slots<R>& c, State& S) {
    const double nalpha = A.nalpha;
    const D2 pa = as_d2(c.pa[t]), pb = as_d2(c.pb[t]), pc = as_d2(c.pc[t]), pd = as_d2(c.pd[t]);
    const double phiC = pa.x, xy = pa.y, duR = pd.x, dvR = pd.y;
    const double duU = from_above<DPP>(S.duL);
    const double dvU = from_above<DPP>(S.dvL);
    const double phiU = from_above<DPP>(S.phiL);
    const double duD = from_below<DPP>(duR);
    const double dvD = from_below<DPP>(dvR);
    // Every operand that does not exist (image border, padding) is an exact 0.0 here, so the reference's
    // conditional terms (src/OpticalFlow.cpp:468-495) reduce to adding +-0 in the same order.  Ghost lanes
    // (om1 == 1, a1 == a2 == 0) pass their centre value through: 1*c + 0*(..) == c.
    double s1 = S.phiL * S.duL;
    double s2 = S.phiL * S.dvL;
    s1 += phiC * duR;
    s2 += phiC * dvR;
    s1 += phiU * duU;
    s2 += phiU * dvU;
    s1 += phiC * duD;
    s2 += phiC * dvD;
    s1 *= nalpha;
    s2 *= nalpha;
    s1 += xy * S.dvC;
    const double duN = om1 * S.duC + pb.x * (pc.x - s1);
    s2 += xy * duN;
    const double dvN = om1 * S.dvC + pb.y * (pc.y - s2);
    __builtin_amdgcn_raw_buffer_store_b128(as_u4(duN, dvN), T.rd, L.st, (unsigned)s * L.pos_d, kAuxSc1);
    if (SPLIT)  // lane 62 of the band above a cut: the same cell into the inbox of the rank below (every other lane: off)
        __builtin_amdgcn_raw_buffer_store_b128(as_u4(duN, dvN), T.rd2, L.hs, (unsigned)s * L.pos_d, kAuxSc1);
    store_data_guard();
    S.duL = duN;
    S.dvL = dvN;
    S.phiL = moved(phiC);
    S.duC = moved(duR);
    S.dvC = moved(dvR);
    asm volatile("" ::: "memory");  // keep this step's store ahead of its refill loads in the instruction stream
    load_slot<R, t>(T, L, s + R, c);  // refill this slot for the step R ahead
}

template <int R, int t, bool DPP>
struct Unroll {
    static __device__ __forceinline__ void run(const ExactArgs& A, const Task& T, const LaneOffs& L, double om1,
                                               int s0, Slots<R>& c, State& S) {
        Unroll<R, t - 1, DPP>::run(A, T, L, om1, s0, c, S);
        step<R, t, DPP, false>(A, T, L, om1, s0 + t, c, S);
    }
    static __device__ __forceinline__ void fill_coef(const Task& T, const LaneOffs& L, Slots<R>& c) {
        Unroll<R, t - 1, DPP>::fill_coef(T, L, c);
        load_coef<R, t>(T, L, t, c);
    }
    static __device__ __forceinline__ void fill_unknowns(const Task& T, const LaneOffs& L, Slots<R>& c) {
        Unroll<R, t - 1, DPP>::fill_unknowns(T, L, c);
        load_unknowns<R, t>(T, L, t, c);
    }
};
template <int R, bool DPP>
struct Unroll<R, -1, DPP> {
    static __device__ __forceinline__ void run(const ExactArgs&, const Task&, const LaneOffs&, double, int, Slots<R>&,
                                               State&) {}
    static __device__ __forceinline__ void fill_coef(const Task&, const LaneOffs&, Slots<R>&) {}
    static __device__ __forceinline__ void fill_unknowns(const Task&, const LaneOffs&, Slots<R>&) {}
};

// steps t0 .. t1-1 of the unrolled iteration
template <int R, int t0, int t1, bool DPP, bool SPLIT = false>
struct Seg {
    static __device__ __forceinline__ void run(const ExactArgs& A, const Task& T, const LaneOffs& L, double om1,
                                               int s0, Slots<R>& c, State& S) {
        step<R, t0, DPP, SPLIT>(A, T, L, om1, s0 + t0, c, S);
        Seg<R, t0 + 1, t1, DPP, SPLIT>::run(A, T, L, om1, s0, c, S);
    }
};
template <int R, int t1, bool DPP, bool SPLIT>
struct Seg<R, t1, t1, DPP, SPLIT> {
    static __device__ __forceinline__ void run(const ExactArgs&, const Task&, const LaneOffs&, double, int, Slots<R>&,
                                               State&) {}
};

// Progress of the two producers of a task, polled together (a missing producer reads as "finished").
struct Polls {
    unsigned own, up, dn2;
};
struct Deps {  // wave-uniform
    unsigned *own, *up, *dn2;  // counters of (b, k-1), (b-1, k) and (b+1, k-2)
    bool has_own, has_up, has_dn2;
};
__device__ __forceinline__ Polls poll(const Deps& d) {
    Polls p;
    p.own = d.has_own ? __hip_atomic_load(d.own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffffu;
    p.up = d.has_up ? __hip_atomic_load(d.up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffffu;
    p.dn2 = d.has_dn2 ? __hip_atomic_load(d.dn2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffffu;
    return p;
}
// The progress values and the markers are wave-uniform by construction (every lane loads the same word).  Reading them
// through v_readfirstlane makes the branches on them SCALAR: the compiler then keeps real early exits instead of
// structurising them into exec-masked paths that rejoin the sweep loop -- along those phantom paths a slot refilled a
// few instructions earlier would reach its next use, and the waitcnt insertion pass answered with `s_waitcnt vmcnt(0)`
// at the top of every iteration (a full drain of the operand pipeline, ~0.06 us per step).
__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ Polls uni(const Polls& p) { return Polls{uni(p.own), uni(p.up), uni(p.dn2)}; }

// May every load that touches steps < s_end be issued?  (see the dependency tables in the file header; UP = how far the
// ghost lanes read ahead in the band above: 63 steps in k_sor_exact, 64 in k_sor_fused; OWN = 0 / 1 likewise for the
// own band's previous sweep / pair)
template <int OWN, int UP>
__device__ __forceinline__ bool covered(const Polls& p, int ns, int s_end) {
    return p.own >= (unsigned)min(ns, s_end + OWN) && p.up >= (unsigned)min(ns, s_end + UP) &&
           p.dn2 >= (unsigned)min(ns, max(0, s_end - 62));
}

// Bounded wave-uniform wait until the producers cover steps < s_end.  false = abort / timeout.  The fast path consumes
// the poll that was issued an iteration earlier (an in-order wait on that poll only); the slow path polls afresh.
// A task that sees the abort word (or gives up) simply ENDS (s_endpgm): an early `return` out of the sweep loop would
// again be merged into the loop's latch block by the compiler (same phantom paths as above).
__device__ __forceinline__ void end_task() { __builtin_amdgcn_endpgm(); }

template <int OWN, int UP, bool SELECTIVE = false>
__device__ __forceinline__ bool wait_covered(const ExactArgs& A, const Polls& pl, const Deps& d, int s_end) {
    Polls p = uni(pl);
    if (!covered<OWN, UP>(p, A.ns, s_end)) {
        unsigned spins = 0;
        do {
            __builtin_amdgcn_s_sleep(PAPOF_V_SLEEP);
            if ((++spins & 255u) == 0u) {
                if (uni(__hip_atomic_load(A.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) return false;
                if (spins > kSpinLimit) {
                    __hip_atomic_store(A.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
            if (!SELECTIVE) {
                p = uni(poll(d));
                continue;
            }
            // k_sor_exact polls only what is still missing (usually one counter): a shorter round, the hand-off-bound
            // levels gain 1 %; the bandwidth-bound fused level-0 solve LOSES 4 % with it (A/B on one box), hence the flag
            const unsigned need_own = (unsigned)min(A.ns, s_end + OWN), need_up = (unsigned)min(A.ns, s_end + UP);
            const unsigned need_dn2 = (unsigned)min(A.ns, max(0, s_end - 62));
            if (p.own < need_own) p.own = uni(__hip_atomic_load(d.own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (p.up < need_up) p.up = uni(__hip_atomic_load(d.up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (p.dn2 < need_dn2) p.dn2 = uni(__hip_atomic_load(d.dn2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        } while (!covered<OWN, UP>(p, A.ns, s_end));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler-only: keep the loads below the polls
    return true;
}

template <int R, bool DPP, bool SPLIT = false>
__global__ __launch_bounds__(64) void k_sor_exact(ExactArgs A_in) {
    static_assert(R >= 4 && R % 2 == 0, "two markers per iteration");
    ExactArgs A = A_in;
    {   // the frame pair of a batched launch (wave-uniform: scalar arithmetic on the kernel arguments)
        const size_t p = blockIdx.y;
        A.phi += p * A.bs_coef;
        A.a1 += p * A.bs_coef;
        A.b1 += p * A.bs_coef;
        A.du += p * A.bs_d;
        A.prog += p * A.bs_prog;
    }
    stamp_now(A.stamp);
    const unsigned lane = threadIdx.x;
    // Workgroups are dealt round-robin over the 8 XCDs (blocks x and x + 8 share one; observed, speed only), so with
    // xcd_affine the sweeps of one band all run on one XCD: sweep k+1 re-reads the coefficient cells sweep k read a
    // few dozen steps earlier out of that XCD's L2 instead of the Infinity Cache.  Dependencies still point to lower
    // block indices only.
    int k, b;
    if (A.xcd_affine) {
        // rows of S blocks per sweep, S a multiple of 8: the first 8 * (nb / 8) bands sit at their own index (band b on
        // XCD b % 8 in every sweep); the nb % 8 remaining bands occupy consecutive places of one more group of 8 whose
        // start moves with the sweep, so that they load every XCD equally instead of always the same ones
        const int nf = A.nb >> 3, rem = A.nb & 7, S = 8 * nf + (rem ? 8 : 0);
        const int kl = blockIdx.x / S;
        k = A.k0 + kl;
        const int r = blockIdx.x - kl * S;
        if (r < 8 * nf) {
            b = r;
        } else {
            const int j = r - 8 * nf - (nf ? k % (9 - rem) : 0);
            if (j < 0 || j >= rem) return;
            b = 8 * nf + j;
        }
    } else {
        const int kl = blockIdx.x / A.nbl;
        k = A.k0 + kl;
        b = A.b0 + (blockIdx.x - kl * A.nbl);
    }
    const int ns = A.ns;
    const bool ghost = lane == 0 || lane == kLanes - 1;
    unsigned long long* const dbg = A.dbg ? A.dbg + (size_t)(k * A.nb + b) * 8 : nullptr;  // every lane stores the same
    if (dbg) dbg[0] = __builtin_amdgcn_s_memrealtime();
    Task T;
    const unsigned plane_bytes = (unsigned)(((size_t)A.npos * A.hp + kLanes) * 16u);
    constexpr unsigned kBlock = kLanes * 16u;                                  // one task's cells of one position
    const unsigned par_bytes = (unsigned)A.npos_d * (unsigned)A.nb * kBlock;  // one parity of the (du, dv) planes
    T.ra = __builtin_amdgcn_make_buffer_rsrc((void*)A.phi, 0, plane_bytes, 0x00020000);
    T.rb = __builtin_amdgcn_make_buffer_rsrc((void*)A.a1, 0, plane_bytes, 0x00020000);
    T.rc = __builtin_amdgcn_make_buffer_rsrc((void*)A.b1, 0, plane_bytes, 0x00020000);
    T.rd = __builtin_amdgcn_make_buffer_rsrc((void*)A.du, 0, 2u * par_bytes, 0x00020000);
    // lane 0 of task (b, k) stands for image row r0 = 62b - k - 1; at step 0 the task sits at position r0 + qt
    const int r0 = kBandRows * b - k - 1;
    LaneOffs L;
    L.pos_c = (unsigned)A.hp * 16u;
    L.pos_d = (unsigned)A.nb * kBlock;
    const unsigned base = ((unsigned)(r0 + A.qt) * (unsigned)A.hp + (unsigned)(r0 + A.rt) + lane) * 16u;
    // DEAD lanes (round 4): a lane stands for ONE image row during its whole task; where that row lies outside the image --
    // the top band's first k + 1 lanes at sweep k, the last band's lanes beyond row H - 1 -- every operand it would load is a
    // non-cell (0.0 by the layout's contract) or a zero it stored itself, and nobody but another dead lane reads what it stores.
    // Its accesses are switched off like a ghost lane's coefficients (kOob: the load returns 0.0, the store is dropped, no line is
    // touched): 9 % (1440x810) to 13 % (1080x607) of a level's lanes.  The one thing that can differ is the SIGN of a zero a
    // dead lane used to store ((1 - omega) * 0 + 0 * x) and a live neighbour multiplies by a weight and adds to a sum -- which
    // changes nothing unless that sum is itself an exact zero of the opposite sign; the full-array SHA-256 comparisons with the
    // reference's goldens and the bit-for-bit tests against the oracle hold on every shape tested.  PAPOF_SOR_DEAD=0 restores
    // the loads and stores (A/B).  Not in SPLIT launches (the inbox cells of a cut are always exchanged).
    const int row = r0 + (int)lane;
    const bool dead = !SPLIT && A.skip_dead != 0 && (row < 0 || row >= A.H);
    L.pa = dead ? kOob : base;
    L.pbc = (ghost || dead) ? kOob : base;
    // (du, dv): the writer of step s uses position s + 1 (position 0 is never written: the centre before step 0)
    const unsigned mine = (unsigned)(k & 1) * par_bytes, prev = (unsigned)((k + 1) & 1) * par_bytes;
    L.st = dead ? kOob : mine + L.pos_d + (unsigned)b * kBlock + lane * 16u;
    // lanes >= 1: own block, cell lane - 1; lane 0: the row above, NEW value: block b-1, cell 62, written by (b-1, k) 63
    // steps ahead of ours.  (Selects, not branches: a divergent branch anywhere makes the compiler structurise the kernel.)
    // (SPLIT, first band below a cut: the inbox cell of this sweep -- see ExactArgs -- instead of cell 62)
    const bool cut_top = SPLIT && A.top_cut && b == A.b0, cut_bot = SPLIT && A.bot_cut && b == A.b0 + A.nbl - 1;
    const unsigned up_cell = cut_top ? (unsigned)(k >> 1) : 62u;
    const unsigned pd_above = b > 0 ? mine + 64u * L.pos_d + (unsigned)(b - 1) * kBlock + up_cell * 16u : kOob;
    L.hs = (cut_bot && lane == kLanes - 2) ? mine + L.pos_d + (unsigned)b * kBlock + (unsigned)(k >> 1) * 16u : kOob;
    T.rd2 = __builtin_amdgcn_make_buffer_rsrc((void*)((SPLIT && A.peer_du) ? A.peer_du : A.du), 0, 2u * par_bytes, 0x00020000);
    // Sweep 0 reads du = dv = 0 (src/OpticalFlow.cpp:452-453) as out-of-range offsets, not from memory: the planes need
    // no clearing between solves except for the tail positions no task writes (sor_solve clears them anyway: cache warming).
    L.pd = dead ? kOob : (lane == 0 ? pd_above : (k > 0 ? prev + L.pos_d + (unsigned)b * kBlock + (lane - 1u) * 16u : kOob));
    const double om1 = ghost ? 1.0 : A.om1;  // ghost lanes pass their centre value through unchanged

    // one 128-byte line per counter: hundreds of waves publish and poll concurrently, and counters sharing a
    // line would serialise at the memory side
    const unsigned prog_bytes = (unsigned)A.n_sor * (unsigned)A.nb * kProgStride * 4u;
    T.rp = __builtin_amdgcn_make_buffer_rsrc((void*)A.prog, 0, prog_bytes, 0x00020000);
    T.rp2 = __builtin_amdgcn_make_buffer_rsrc((void*)((SPLIT && A.peer_prog) ? A.peer_prog : A.prog), 0, prog_bytes, 0x00020000);
    const unsigned my_prog = lane == 0 ? (unsigned)(k * A.nb + b) * kProgStride * 4u : kOob;
    const unsigned my_prog2 = cut_bot ? my_prog : kOob;  // the same counter in the memory of the rank below the cut
    Deps D;
    D.has_own = k > 0;
    D.has_up = b > 0;
    // (above a cut the band below reads its own per-sweep inbox cells, never this block: no write-after-read edge)
    D.has_dn2 = k > 1 && b + 1 < A.nb && !cut_bot;
    D.own = A.prog + ((size_t)(k - 1) * A.nb + b) * kProgStride;
    D.up = A.prog + ((size_t)k * A.nb + (b - 1)) * kProgStride;
    D.dn2 = A.prog + ((size_t)(k - 2) * A.nb + (b + 1)) * kProgStride;
    const auto publish_all = [&](unsigned steps) {
        publish(T, my_prog, steps);
        if (SPLIT) __builtin_amdgcn_raw_buffer_store_b32(steps, T.rp2, my_prog2, 0, 16 /* sc1 */);
    };

    const int n_iter = (ns + R - 1) / R;  // steps beyond ns only touch padding (npos leaves room for them)
    State S;
    S.duL = S.dvL = S.phiL = 0.0;
    Slots<R> c;

    // Every load must be covered by the producers' published progress before it is issued.  Iteration i issues the
    // loads of steps < (i + 2) R; its coverage check uses a poll that was itself issued one iteration earlier
    // (consuming a poll forces an in-order vmcnt wait on every older load, so finer-grained polling stalls).
    // The coefficient operands depend on nobody: their first R steps are requested before the task waits.
    // Start-up in two stages: the first R steps' unknowns are requested as soon as the producers cover THEM; the check
    // for the next R steps (the refills of iteration 0) is made with a fresh poll at the start of iteration 0, while those
    // first loads are in flight -- a hand-off then costs R steps and one load latency less than waiting for 2R up front.
    Unroll<R, R - 1, DPP>::fill_coef(T, L, c);
    Polls pl = poll(D);
    if (!wait_covered<0, 63, true>(A, pl, D, R)) end_task();
    {  // centre of the first cells = the right-old of "step -1" (lanes >= 1: left of column 0, zero; lane 0: position 63
       // of the block above)
        const unsigned first = (lane != 0 || L.pd == kOob) ? kOob : L.pd - L.pos_d;
        const D2 c0 = as_d2(__builtin_amdgcn_raw_buffer_load_b128(T.rd, first, 0, kAuxSc1));
        S.duC = c0.x;
        S.dvC = c0.y;
    }
    if (dbg) dbg[1] = __builtin_amdgcn_s_memrealtime();
    Unroll<R, R - 1, DPP>::fill_unknowns(T, L, c);
    pl = poll(D);

    // MARKER loads: a 4-byte load issued right after the store of some step retires (vmcnt is in order) only after
    // that store has completed, so consuming it a few steps later proves the step complete and lets the task
    // publish it without draining the pipeline.  Two markers per iteration (after steps H-1 and R-1), each consumed
    // DM steps later (4 / 5 / R - 3 at depth 6 / 8 / >= 10, see PAPOF_V_DM*), in the next iteration: published progress lags
    // the real one by DM .. DM + R/2 steps.  The marker reads the abort word, so a raised abort also ends every running task within two iterations.
    constexpr int H = R / 2, DM = R >= 10 ? R - 3 : (R == 8 ? PAPOF_V_DM8 : (R == 6 ? PAPOF_V_DM6 : H));
    constexpr int CA = H + DM - R, CB = DM;  // steps of the NEXT iteration after which markers A / B are consumed
    static_assert(CA >= 0 && CA < H && CB >= H && CB < R, "marker consumption points");
    unsigned ma = 0u, mb = 0u;
    // one iteration = R steps; `i` is only used for step numbers.  The first iteration is peeled off the loop so that the
    // loop header joins two states with the same pipeline contents (after an iteration / after an iteration).
    const auto iteration = [&](int i, bool first) {
        if (!wait_covered<0, 63, true>(A, pl, D, (i + 2) * R)) end_task();
        // The poll for the next iteration's check is consumed at the start of the next iteration, where it waits (in
        // order) for every load issued before it.  Issued in the middle of this iteration it is half an iteration fresher
        // (a shorter hand-off) but then waits for the refills of the first half: at R >= 8 those are >= 4 steps old and long
        // there (1080x607: 0.519 -> 0.505 ms, 810x455: 0.380 -> 0.370), at R = 6 only 3 steps (607x341: 0.336 -> 0.378).
        constexpr bool kMidPoll = R >= PAPOF_V_MIDPOLL;
        Polls pn{0u, 0u, 0u};
        if (!kMidPoll) pn = poll(D);
        // Markers are consumed DM = R - 3 steps after they were issued -- i.e. in the NEXT iteration: consuming a marker
        // waits (in-order vmcnt) for every load issued before it, and with DM = 3 that exposed ~0.35 us of latency twice
        // per iteration (the refills issued in those 3 steps).  R - 3 steps later all of them have long arrived.
        Seg<R, 0, CA, DPP, SPLIT>::run(A, T, L, om1, i * R, c, S);
        if (!first) {  // marker A of the previous iteration: its steps < (i-1)R + H are complete
            asm volatile("" ::"v"(ma), "v"(S.duL), "v"(S.dvL) : "memory");
            if (uni(ma) != 0u) end_task();
            publish_all((unsigned)min(ns, (i - 1) * R + H));
        }
        Seg<R, CA, H, DPP, SPLIT>::run(A, T, L, om1, i * R, c, S);
        ma = __hip_atomic_load(A.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // marker A: after step H - 1
        if (kMidPoll) pn = poll(D);
        Seg<R, H, CB, DPP, SPLIT>::run(A, T, L, om1, i * R, c, S);
        if (!first) {  // marker B of the previous iteration: steps < i R are complete
            asm volatile("" ::"v"(mb), "v"(S.duL), "v"(S.dvL) : "memory");
            if (uni(mb) != 0u) end_task();
            publish_all((unsigned)min(ns, i * R));
        }
        Seg<R, CB, R, DPP, SPLIT>::run(A, T, L, om1, i * R, c, S);
        mb = __hip_atomic_load(A.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // marker B: after step R - 1
        pl = pn;
    };
    iteration(0, true);
    if (dbg) dbg[2] = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i < n_iter; ++i) {
        iteration(i, false);
        if (dbg && i < 4) dbg[2 + i] = __builtin_amdgcn_s_memrealtime();
    }
    // final publication: every store of this wave has left the CU before the counter moves (guide G16/R1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish_all((unsigned)ns);
    if (dbg) dbg[6] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------------------------------------
// FUSED PAIRS: two consecutive sweeps of a band in ONE wavefront (temporal blocking in registers).
//
// Task (band b, pair q) runs sweeps k = 2q and k + 1.  Lane l stands for image row r0 + l, r0 = 61b - 2q - 2, in BOTH
// sweeps; at step s the first sweep updates column s - l (exactly as k_sor_exact), the second sweep column s - l - 2:
// everything the second sweep needs from the first -- centre (r, j), right (r, j+1), down (r+1, j) at sweep k -- was
// produced one / two steps earlier by this lane or by lane l + 1, so it is taken from registers; the coefficient cells
// of step s - 2 are simply kept two steps longer (their slot is refilled after the second sweep has used it).  The two
// updates of a step are independent chains, so the wave issues them interleaved: a step costs little more than one
// sweep's, but moves the memory operations of ONE sweep (4 loads + 1 store) for TWO sweeps of work, and the number of
// sweep-to-sweep hand-offs through memory halves.
//   lanes, first sweep:   0 carrier (below)   1 ghost: row above, sweep k      2..62 real (61 rows)   63 ghost: row below
//   lanes, second sweep:  0 ghost: row above, sweep k+1      1..61 real (the same 61 rows + 1 = bands climb two rows
//                         per pair)            62 its first-sweep value IS the row below at sweep k    63 unused
//   lane 0 is a pass-through in both sweeps: it loads the sweep-(k+1) values of its row (written by the band above) as
//   its first-sweep "right" operand, hands them unchanged to its second sweep, and lane 1 reads them there as up-new.
// One store per step: lanes 0..61 their second-sweep result, lane 62 its FIRST-sweep result (the band below needs
// that row at sweep k: its lane 1), lane 63 nothing.  Planes alternate per pair.  With an odd sweep count the last
// pair's second sweep is the identity (compile-time variant), so the result layout does not depend on parity.
// Dependencies of (b, q), in completed steps, before the loads of steps < e are issued:
//     prog[q-1][b]   >= min(NS, e + 1)     own band, previous pair (lane l reads its cell l - 2 at position s + 2)
//     prog[q][b-1]   >= min(NS, e + 64)    band above, this pair (lane 1: cell 62 at position s + 63, lane 0: cell 61
//                                          at position s + 65)
//     prog[q-2][b+1] >= min(NS, e - 62)    band below, two pairs ago (write-after-read of cells 61 / 62 of our block)
// ------------------------------------------------------------------------------------------------
struct FLane {       // per-lane constants of the fused kernel (VGPRs)
    unsigned m1;     // first sweep: all ones where the lane is real (2..62), else 0 -> a1 = a2 = 0 -> pass-through
    double om1a;     // 1 - omega of the first sweep (1.0 on ghost lanes)
    double om1b;     // ... of the second sweep
    bool first_out;  // lane 62: stores its first-sweep result
    bool own_block;  // lanes >= 2: the first sweep's operands come from the own block (lanes 0 / 1: from the band above)
};

template <int R, int t, bool DPP, bool SKIP2, bool ID2>
__device__ __forceinline__ void f_step(const ExactArgs& A, const Task& T, const LaneOffs& L, const FLane& F, int s,
                                       Slots<R>& c, State& S1, State& S2) {
    const double nalpha = A.nalpha;
    constexpr int t2 = (t + R - 2) % R;  // slot holding the coefficient cells of step s - 2
    // ---- second sweep, column s - l - 2: operands are the first sweep's results of steps s - 2 (centre) and s - 1
    const double duR2 = S1.duL, dvR2 = S1.dvL;
    double duN2, dvN2;
    if (SKIP2) {  // steps 0 and 1: every lane is still left of column 0
        duN2 = 0.0;
        dvN2 = 0.0;
    } else {
        // ID2 -- the identity second sweep of the LAST pair of an odd sweep count -- is the same arithmetic with (a1, a2)
        // masked to zero and 1 - omega -> 1 (F.om1b, set by the kernel): the exact pass-through the ghost lanes use.  (A
        // shortcut that simply forwarded the first sweep's results two steps later returned wrong quads of lanes whenever
        // OTHER kernels ran on the chip at the same time: in its loop the compiler had placed the inline-asm moves of
        // moved() straight behind the 16-byte stores, into the stores' data registers -- the hazard store_data_guard() now
        // closes for every variant.  The shortcut saved nothing that matters and stays out.)
        const D2 qa = as_d2(c.pa[t2]), qc = as_d2(c.pc[t2]);
        const u32x4 qbm = ID2 ? (c.pb[t2] & u32x4{0u, 0u, 0u, 0u}) : c.pb[t2];
        const D2 qb = as_d2(qbm);
        const double phiC = qa.x, xy = qa.y;
        const double duU = from_above<DPP>(S2.duL);
        const double dvU = from_above<DPP>(S2.dvL);
        const double phiU = from_above<DPP>(S2.phiL);
        const double duD = from_below<DPP>(duR2);
        const double dvD = from_below<DPP>(dvR2);
        double s1 = S2.phiL * S2.duL;
        double s2 = S2.phiL * S2.dvL;
        s1 += phiC * duR2;
        s2 += phiC * dvR2;
        s1 += phiU * duU;
        s2 += phiU * dvU;
        s1 += phiC * duD;
        s2 += phiC * dvD;
        s1 *= nalpha;
        s2 *= nalpha;
        s1 += xy * S2.dvC;
        duN2 = F.om1b * S2.duC + qb.x * (qc.x - s1);
        s2 += xy * duN2;
        dvN2 = F.om1b * S2.dvC + qb.y * (qc.y - s2);
        S2.phiL = moved(phiC);
    }
    // ---- first sweep, column s - l (as k_sor_exact::step, with the per-lane ghost mask on (a1, a2))
    const D2 pa = as_d2(c.pa[t]), pc = as_d2(c.pc[t]), pd = as_d2(c.pd[t]);
    const u32x4 pbm = c.pb[t] & u32x4{F.m1, F.m1, F.m1, F.m1};
    const D2 pb = as_d2(pbm);
    const double phiC = pa.x, xy = pa.y, duR = pd.x, dvR = pd.y;
    const double duU = from_above<DPP>(S1.duL);
    const double dvU = from_above<DPP>(S1.dvL);
    const double phiU = from_above<DPP>(S1.phiL);
    const double duD = from_below<DPP>(duR);
    const double dvD = from_below<DPP>(dvR);
    double s1 = S1.phiL * S1.duL;
    double s2 = S1.phiL * S1.dvL;
    s1 += phiC * duR;
    s2 += phiC * dvR;
    s1 += phiU * duU;
    s2 += phiU * dvU;
    s1 += phiC * duD;
    s2 += phiC * dvD;
    s1 *= nalpha;
    s2 *= nalpha;
    s1 += xy * S1.dvC;
    const double duN = F.om1a * S1.duC + pb.x * (pc.x - s1);
    s2 += xy * duN;
    const double dvN = F.om1a * S1.dvC + pb.y * (pc.y - s2);
    const double o1 = F.first_out ? duN : duN2, o2 = F.first_out ? dvN : dvN2;
    __builtin_amdgcn_raw_buffer_store_b128(as_u4(o1, o2), T.rd, L.st, (unsigned)s * L.pos_d, kAuxSc1);
    store_data_guard();
    S2.duL = duN2;
    S2.dvL = dvN2;
    S2.duC = duR2;
    S2.dvC = dvR2;
    S1.duL = duN;
    S1.dvL = dvN;
    S1.phiL = phiC;  // its slot lives two more steps
    S1.duC = moved(duR);
    S1.dvC = moved(dvR);
    asm volatile("" ::: "memory");  // keep this step's store ahead of its refill loads in the instruction stream
    load_unknowns<R, t>(T, L, s + R, c);
    if (!SKIP2) load_coef<R, t2>(T, L, s - 2 + R, c);  // the slot of step s - 2 is free now
}

template <int R, int t0, int t1, bool DPP, bool FIRST, bool ID2>
struct FSeg {
    static __device__ __forceinline__ void run(const ExactArgs& A, const Task& T, const LaneOffs& L, const FLane& F,
                                               int s0, Slots<R>& c, State& S1, State& S2) {
        f_step<R, t0, DPP, FIRST && t0 < 2, ID2>(A, T, L, F, s0 + t0, c, S1, S2);
        FSeg<R, t0 + 1, t1, DPP, FIRST, ID2>::run(A, T, L, F, s0, c, S1, S2);
    }
};
template <int R, int t1, bool DPP, bool FIRST, bool ID2>
struct FSeg<R, t1, t1, DPP, FIRST, ID2> {
    static __device__ __forceinline__ void run(const ExactArgs&, const Task&, const LaneOffs&, const FLane&, int,
                                               Slots<R>&, State&, State&) {}
};

template <int R, bool DPP, bool ID2>
__device__ __forceinline__ void f_run(const ExactArgs& A, const Task& T, const LaneOffs& L, const FLane& F,
                                      const Deps& D, unsigned my_prog) {
    const int ns = A.ns;
    const int n_iter = (ns + R - 1) / R;
    State S1, S2;
    S1.duL = S1.dvL = S1.phiL = 0.0;
    S2.duL = S2.dvL = S2.phiL = S2.duC = S2.dvC = 0.0;
    Slots<R> c;
    Unroll<R, R - 1, DPP>::fill_coef(T, L, c);
    Polls pl = poll(D);
    if (!wait_covered<1, 64, PAPOF_V_FSELECTIVE>(A, pl, D, R)) end_task();  // staged start-up, see k_sor_exact
    {  // centre of the first cells = the right operand of "step -1" (lanes >= 2: left of column 0, zero)
        const unsigned first = (F.own_block || L.pd == kOob) ? kOob : L.pd - L.pos_d;
        const D2 c0 = as_d2(__builtin_amdgcn_raw_buffer_load_b128(T.rd, first, 0, kAuxSc1));
        S1.duC = c0.x;
        S1.dvC = c0.y;
    }
    Unroll<R, R - 1, DPP>::fill_unknowns(T, L, c);
    pl = poll(D);
    // markers, polls and the peeled first iteration: exactly as in k_sor_exact (see there)
    constexpr int H = R / 2, DM = R == 6 ? PAPOF_V_FDM6 : (R == 8 ? PAPOF_V_FDM8 : (R >= 6 ? R - 3 : H));
    constexpr int CA = H + DM - R, CB = DM;
    static_assert(CA >= 0 && CA < H && CB >= H && CB < R, "marker consumption points");
    unsigned ma = 0u, mb = 0u;
    const auto iteration = [&](int i, auto first_c) {
        constexpr bool first = decltype(first_c)::value;
        if (!wait_covered<1, 64, PAPOF_V_FSELECTIVE>(A, pl, D, (i + 2) * R)) end_task();
        constexpr bool kMidPoll = R >= PAPOF_V_FMIDPOLL;
        Polls pn{0u, 0u, 0u};
        if (!kMidPoll) pn = poll(D);
        FSeg<R, 0, CA, DPP, first, ID2>::run(A, T, L, F, i * R, c, S1, S2);
        if (!first) {
            asm volatile("" ::"v"(ma), "v"(S1.duL), "v"(S1.dvL) : "memory");
            if (uni(ma) != 0u) end_task();
            publish(T, my_prog, (unsigned)min(ns, (i - 1) * R + H));
        }
        FSeg<R, CA, H, DPP, first, ID2>::run(A, T, L, F, i * R, c, S1, S2);
        ma = __hip_atomic_load(A.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // marker A: after step H - 1
        if (kMidPoll) pn = poll(D);
        FSeg<R, H, CB, DPP, first, ID2>::run(A, T, L, F, i * R, c, S1, S2);
        if (!first) {
            asm volatile("" ::"v"(mb), "v"(S1.duL), "v"(S1.dvL) : "memory");
            if (uni(mb) != 0u) end_task();
            publish(T, my_prog, (unsigned)min(ns, i * R));
        }
        FSeg<R, CB, R, DPP, first, ID2>::run(A, T, L, F, i * R, c, S1, S2);
        mb = __hip_atomic_load(A.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // marker B: after step R - 1
        pl = pn;
    };
    iteration(0, std::true_type{});
    for (int i = 1; i < n_iter; ++i) iteration(i, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish(T, my_prog, (unsigned)ns);
}

template <int R, bool DPP>
__global__ __launch_bounds__(64) void k_sor_fused(ExactArgs A) {
    static_assert(R >= 6 && R % 2 == 0, "two markers per iteration; coefficient slots live two extra steps");
    stamp_now(A.stamp);
    const unsigned lane = threadIdx.x;
    const int pairs = (A.n_sor + 1) >> 1;
    int q, b;
    if (A.xcd_affine) {  // as k_sor_exact: all pairs of a band on one XCD (<= 8 bands)
        const int nf = A.nb >> 3, rem = A.nb & 7, S = 8 * nf + (rem ? 8 : 0);
        const int ql = blockIdx.x / S;
        q = A.k0 + ql;
        const int r = blockIdx.x - ql * S;
        if (r < 8 * nf) {
            b = r;
        } else {
            const int j = r - 8 * nf - (nf ? q % (9 - rem) : 0);
            if (j < 0 || j >= rem) return;
            b = 8 * nf + j;
        }
    } else {
        const int ql = blockIdx.x / A.nbl;
        q = A.k0 + ql;
        b = A.b0 + (blockIdx.x - ql * A.nbl);
    }
    Task T;
    const unsigned plane_bytes = (unsigned)(((size_t)A.npos * A.hp + kLanes) * 16u);
    constexpr unsigned kBlock = kLanes * 16u;
    const unsigned par_bytes = (unsigned)A.npos_d * (unsigned)A.nb * kBlock;
    T.ra = __builtin_amdgcn_make_buffer_rsrc((void*)A.phi, 0, plane_bytes, 0x00020000);
    T.rb = __builtin_amdgcn_make_buffer_rsrc((void*)A.a1, 0, plane_bytes, 0x00020000);
    T.rc = __builtin_amdgcn_make_buffer_rsrc((void*)A.b1, 0, plane_bytes, 0x00020000);
    T.rd = __builtin_amdgcn_make_buffer_rsrc((void*)A.du, 0, 2u * par_bytes, 0x00020000);
    const int r0 = kFusedRows * b - 2 * q - 2;  // image row of lane 0
    LaneOffs L;
    L.pos_c = (unsigned)A.hp * 16u;
    L.pos_d = (unsigned)A.nb * kBlock;
    const unsigned base = ((unsigned)(r0 + A.qt) * (unsigned)A.hp + (unsigned)(r0 + A.rt) + lane) * 16u;
    L.pa = base;
    L.pbc = (lane == 0 || lane == kLanes - 1) ? kOob : base;
    const unsigned mine = (unsigned)(q & 1) * par_bytes, prev = (unsigned)((q + 1) & 1) * par_bytes;
    L.st = lane == kLanes - 1 ? kOob : mine + L.pos_d + (unsigned)b * kBlock + lane * 16u;  // step s -> position s + 1
    // first-sweep right operand of step s: the previous pair's cell lane - 2 at position s + 2 (the band climbed two rows)
    // lane 1: row above at sweep k = first-sweep result of lane 62 of (b-1, q), stored at its step s + 62;
    // lane 0: row above that at sweep k+1 = second-sweep result of lane 61 of (b-1, q), stored at its step s + 64
    const unsigned pd1 = b > 0 ? mine + 63u * L.pos_d + (unsigned)(b - 1) * kBlock + 62u * 16u : kOob;
    const unsigned pd0 = b > 0 ? mine + 65u * L.pos_d + (unsigned)(b - 1) * kBlock + 61u * 16u : kOob;
    const unsigned pd_own = q > 0 ? prev + 2u * L.pos_d + (unsigned)b * kBlock + (lane - 2u) * 16u : kOob;  // pair 0: zeros
    // [dead lanes as in k_sor_exact were tried here and taken out: the level-0 solve is bound by its chain, not by bytes, and the
    // three extra selects of the set-up changed the kernel's register allocation -- 964 instead of 917 us per solve, switch on or off]
    L.pd = lane == 0 ? pd0 : (lane == 1 ? pd1 : pd_own);
    FLane F;
    const bool real1 = lane >= 2 && lane <= kLanes - 2;
    F.m1 = real1 ? 0xffffffffu : 0u;
    F.om1a = real1 ? A.om1 : 1.0;
    F.om1b = (lane == 0 || lane == kLanes - 1) ? 1.0 : A.om1;
    if ((A.n_sor & 1) && q == pairs - 1) F.om1b = 1.0;  // identity second sweep (f_step: ID2)
    F.first_out = lane == kLanes - 2;
    F.own_block = lane >= 2;

    const unsigned prog_bytes = (unsigned)A.n_sor * (unsigned)A.nb * kProgStride * 4u;
    T.rp = __builtin_amdgcn_make_buffer_rsrc((void*)A.prog, 0, prog_bytes, 0x00020000);
    const unsigned my_prog = lane == 0 ? (unsigned)(q * A.nb + b) * kProgStride * 4u : kOob;
    Deps D;
    D.has_own = q > 0;
    D.has_up = b > 0;
    D.has_dn2 = q > 1 && b + 1 < A.nb;
    D.own = A.prog + ((size_t)(q - 1) * A.nb + b) * kProgStride;
    D.up = A.prog + ((size_t)q * A.nb + (b - 1)) * kProgStride;
    D.dn2 = A.prog + ((size_t)(q - 2) * A.nb + (b + 1)) * kProgStride;
    if ((A.n_sor & 1) && q == pairs - 1)
        f_run<R, DPP, true>(A, T, L, F, D, my_prog);
    else
        f_run<R, DPP, false>(A, T, L, F, D, my_prog);
}

// ------------------------------------------------------------------------------------------------
// GROUPED exact-order solver: M consecutive sweeps of a band in ONE workgroup (temporal blocking).
//
// Same tasks, same lanes, same arithmetic and the same cross-workgroup protocol as k_sor_exact above -- but wave m of
// workgroup (band b, group g) runs sweep k = g*M + m, and the sweep-to-sweep hand-off INSIDE a group goes through LDS
// instead of HBM:
//   * wave m writes each step's 64 (du, dv) cells into a 16-slot LDS ring and, once per half-iteration (H = R/2
//     steps), advances an LDS progress word; wave m+1 waits for that word, reads the H blocks it needs (lane l <- cell
//     l-1: the band climbed one row) and acknowledges them in a second word (write-after-read guard of the ring).
//     A hand-off costs ~H steps instead of ~40 steps + three memory round trips, and it moves no HBM bytes;
//   * only wave 0 of a group loads (du, dv) from the ping-pong planes (written by the LAST wave of the previous group,
//     so the planes now alternate per group) and only the last wave stores them there; the coefficient operands of waves
//     1..M-1 are the cache lines wave 0 pulled a few steps earlier (same CU: L1 / L2 hits);
//   * what another BAND needs is one cell per step: the new value of the row above (ghost lane 0).  Every wave writes
//     that cell (its lane 62) to a small halo row HALO[sweep][band][position] behind the two planes; being per sweep,
//     halo rows are never reused, so the write-after-read dependency of k_sor_exact disappears.  The (du, dv) planes and
//     the halo rows sit in one allocation and are addressed through one buffer descriptor with per-lane strides.
// HBM traffic per cell-update drops from 80 B towards 80/M + 32/M B; the critical path loses (M-1)/M of its sweep hops.
// fp64 issue is not the limit: 16 waves of this arithmetic on one CU each still take 244 cycles per step (measured).
// ------------------------------------------------------------------------------------------------
struct GroupArgs {
    const double *phi, *a1, *b1;
    double* du;
    unsigned* prog;   // [n_sor][nb]
    unsigned* abort;  // one word
    int nb, ns, hp, npos, qt, rt, npos_d, n_sor;
    int xcd_affine;
    int g0;  // first group of sweeps of THIS launch (see ExactArgs::k0)
    unsigned long long* dbg;  // diagnostics (PAPOF_SOR_DBG): per task {total, wait_covered, lds in, lds out} shader clocks
    unsigned halo_off;  // byte offset of the halo rows from du
    double nalpha, om1;
    unsigned long long* stamp;  // as ExactArgs::stamp
};

constexpr int kRing = 16;  // LDS ring slots of (du, dv) per producing wave (>= 2 half-iterations)

struct GLane {  // per-lane constant byte offsets (VGPRs)
    unsigned pa, pbc;      // coefficient cells (as LaneOffs)
    unsigned pd, pd_step;  // (du, dv) right-old of step s at pd + s * pd_step   [kOob: LDS-fed lane]
    unsigned st;           // store of step s at st + s * pos_d                   [kOob unless last wave of the group]
    unsigned hs;           // halo store of step s at hs + s * 16                 [lane 62 only]
    unsigned pos_c, pos_d;
};

template <int R>
struct GSlots {
    u32x4 pa[R], pb[R], pc[R], pd[R];
};

template <int R, int t>
__device__ __forceinline__ void g_load_coef(const Task& T, const GLane& L, int s, GSlots<R>& c) {
    const unsigned off = (unsigned)s * L.pos_c;
    c.pa[t] = __builtin_amdgcn_raw_buffer_load_b128(T.ra, L.pa, off, kAuxPlain);
    c.pb[t] = __builtin_amdgcn_raw_buffer_load_b128(T.rb, L.pbc, off, kAuxPlain);
    c.pc[t] = __builtin_amdgcn_raw_buffer_load_b128(T.rc, L.pbc, off, kAuxPlain);
}
template <int R, int t>
__device__ __forceinline__ void g_load_unknowns(const Task& T, const GLane& L, int s, GSlots<R>& c) {
    c.pd[t] = __builtin_amdgcn_raw_buffer_load_b128(T.rd, L.pd + (unsigned)s * L.pd_step, 0, kAuxSc1);
}

// Role of a wave inside its group (compile-time: four straight-line variants of the sweep loop, chosen once per
// wave by a scalar branch, so that the memory-counter bookkeeping of the software pipeline stays exact).
//   FROM_LDS: (du, dv) of the previous sweep come from the ring of the wave before (else: from the planes)
//   TO_LDS:   this sweep's (du, dv) go to the own ring (else: to the planes)
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) volatile unsigned lds_word;

template <int R, int t, bool DPP, bool FROM_LDS, bool TO_LDS>
__device__ __forceinline__ void g_step(const GroupArgs& A, const Task& T, const GLane& L, lds_u32x4* ring_out,
                                       unsigned lane, double om1, int s, GSlots<R>& c, const u32x4 (&lpd)[R / 2],
                                       State& S) {
    const double nalpha = A.nalpha;
    const D2 pa = as_d2(c.pa[t]), pb = as_d2(c.pb[t]), pc = as_d2(c.pc[t]);
    u32x4 raw = c.pd[t];
    if (FROM_LDS) {
        const u32x4 l = lpd[t % (R / 2)];
        raw = lane != 0 ? l : raw;
    }
    const D2 pd = as_d2(raw);
    const double phiC = pa.x, xy = pa.y, duR = pd.x, dvR = pd.y;
    const double duU = from_above<DPP>(S.duL);
    const double dvU = from_above<DPP>(S.dvL);
    const double phiU = from_above<DPP>(S.phiL);
    const double duD = from_below<DPP>(duR);
    const double dvD = from_below<DPP>(dvR);
    double s1 = S.phiL * S.duL;
    double s2 = S.phiL * S.dvL;
    s1 += phiC * duR;
    s2 += phiC * dvR;
    s1 += phiU * duU;
    s2 += phiU * dvU;
    s1 += phiC * duD;
    s2 += phiC * dvD;
    s1 *= nalpha;
    s2 *= nalpha;
    s1 += xy * S.dvC;
    const double duN = om1 * S.duC + pb.x * (pc.x - s1);
    s2 += xy * duN;
    const double dvN = om1 * S.dvC + pb.y * (pc.y - s2);
    const u32x4 out = as_u4(duN, dvN);
    if (TO_LDS)
        ring_out[(s & (kRing - 1)) * kLanes + lane] = out;
    else
        __builtin_amdgcn_raw_buffer_store_b128(out, T.rd, L.st, (unsigned)s * L.pos_d, kAuxSc1);
    __builtin_amdgcn_raw_buffer_store_b128(out, T.rd, L.hs + (unsigned)s * 16u, 0, kAuxSc1);  // lane 62 -> halo row
    store_data_guard();
    S.duL = duN;
    S.dvL = dvN;
    S.phiL = moved(phiC);  // see moved(): no slot register may stay live across its refill
    S.duC = moved(duR);
    S.dvC = moved(dvR);
    asm volatile("" ::: "memory");
    g_load_coef<R, t>(T, L, s + R, c);
    g_load_unknowns<R, t>(T, L, s + R, c);
}

template <int R, int t0, int t1, bool DPP, bool FROM_LDS, bool TO_LDS>
struct GSeg {
    static __device__ __forceinline__ void run(const GroupArgs& A, const Task& T, const GLane& L, lds_u32x4* ring_out,
                                               unsigned lane, double om1, int s0, GSlots<R>& c,
                                               const u32x4 (&lpd)[R / 2], State& S) {
        g_step<R, t0, DPP, FROM_LDS, TO_LDS>(A, T, L, ring_out, lane, om1, s0 + t0, c, lpd, S);
        GSeg<R, t0 + 1, t1, DPP, FROM_LDS, TO_LDS>::run(A, T, L, ring_out, lane, om1, s0, c, lpd, S);
    }
};
template <int R, int t1, bool DPP, bool FROM_LDS, bool TO_LDS>
struct GSeg<R, t1, t1, DPP, FROM_LDS, TO_LDS> {
    static __device__ __forceinline__ void run(const GroupArgs&, const Task&, const GLane&, lds_u32x4*, unsigned,
                                               double, int, GSlots<R>&, const u32x4 (&)[R / 2], State&) {}
};
template <int R, int t>
struct GFill {
    static __device__ __forceinline__ void coef(const Task& T, const GLane& L, GSlots<R>& c) {
        GFill<R, t - 1>::coef(T, L, c);
        g_load_coef<R, t>(T, L, t, c);
    }
    static __device__ __forceinline__ void unknowns(const Task& T, const GLane& L, GSlots<R>& c) {
        GFill<R, t - 1>::unknowns(T, L, c);
        g_load_unknowns<R, t>(T, L, t, c);
    }
};
template <int R>
struct GFill<R, -1> {
    static __device__ __forceinline__ void coef(const Task&, const GLane&, GSlots<R>&) {}
    static __device__ __forceinline__ void unknowns(const Task&, const GLane&, GSlots<R>&) {}
};

// LDS progress words: the LDS unit executes a wave's operations in order, so cells written before a word are visible to
// whoever has seen the word; an explicit lgkmcnt wait keeps the issue order.  Bounded like every other wait.
// All control flow stays wave-uniform (see uni()): words are read through v_readfirstlane, and a word is written by
// lane 0 WITHOUT a branch -- EXEC is narrowed to lane 0 around the ds_write inside one asm statement.
__device__ __forceinline__ void lds_store_lane0(lds_word* p, unsigned v) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_write_b32 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"((unsigned)(unsigned long long)(const volatile void __attribute__((address_space(3)))*)p), "v"(v)
                 : "memory");
}
__device__ __forceinline__ bool lds_wait_ge(lds_word* p, unsigned need, lds_word* group_abort) {
    // LDS accesses only: a vector-memory instruction inside this loop would make the loop exit a join of different
    // vmcnt states, and the compiler would drain the operand pipeline (`s_waitcnt vmcnt(0)`) after every wait
    unsigned spins = 0;
    while (uni(*p) < need) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0u && (uni(*group_abort) != 0u || spins > (kSpinLimit << 3))) return false;
    }
    asm volatile("" ::: "memory");
    return true;
}

struct GWaveCtx {  // wave-uniform
    lds_u32x4 *ring_in, *ring_out;
    lds_word *done_in, *taken_in, *done_out, *taken_out;
    lds_word* group_abort;  // set by a wave of the group that gives up, so that its siblings stop waiting for it
    int task;               // k * nb + b (diagnostics)
    unsigned my_prog;  // lane 0: byte offset of the task's counter, other lanes: out of range
    Deps D;
    int ns, n_iter;
};

template <int R, bool DPP, bool FROM_LDS, bool TO_LDS>
__device__ __forceinline__ void g_run_wave(const GroupArgs& A, const Task& T, const GLane& L, const GWaveCtx& W,
                                           unsigned lane, double om1) {
    constexpr int H = R / 2;
    const int ns = W.ns;
    ExactArgs X;  // the waiting helpers only look at these
    X.abort = A.abort;
    X.ns = ns;
    // giving up (abort seen, or a bounded wait expired): tell the siblings, and the host through the abort word
    const auto give_up = [&]() {
        lds_store_lane0(W.group_abort, 1u);
        __hip_atomic_store(A.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    State S;
    S.duL = S.dvL = S.phiL = 0.0;
    GSlots<R> c;
    u32x4 lpd[H];
#pragma unroll
    for (int q = 0; q < H; q++) lpd[q] = u32x4{0u, 0u, 0u, 0u};
    const unsigned lcell = lane > 0 ? lane - 1u : 0u;  // lane l reads cell l - 1: the band climbed one row

    unsigned long long t_cov = 0, t_in = 0, t_out = 0, t_start = 0;
    const bool dbg = A.dbg != nullptr;
    GFill<R, R - 1>::coef(T, L, c);
    Polls pl = poll(W.D);
    if (!wait_covered<1, 63>(X, pl, W.D, 2 * R)) return give_up();
    if (dbg) t_start = __builtin_amdgcn_s_memtime();
    {  // centre of the first cells: position 0 (zero; LDS-fed lanes: the same zero), lane 0: halo position 63
        const unsigned first = L.pd == kOob ? kOob : L.pd - L.pd_step;
        const D2 c0 = as_d2(__builtin_amdgcn_raw_buffer_load_b128(T.rd, first, 0, kAuxSc1));
        S.duC = c0.x;
        S.dvC = c0.y;
    }
    GFill<R, R - 1>::unknowns(T, L, c);

    // one half-iteration = H steps: LDS operands in, steps, progress words out
    const auto half_begin = [&](int sa) -> bool {
        if (FROM_LDS) {
            const unsigned long long ta = dbg ? __builtin_amdgcn_s_memtime() : 0;
            if (!lds_wait_ge(W.done_in, (unsigned)(sa + H), W.group_abort)) return false;
            if (dbg) t_in += __builtin_amdgcn_s_memtime() - ta;
#pragma unroll
            for (int q = 0; q < H; q++) lpd[q] = W.ring_in[((sa + q) & (kRing - 1)) * kLanes + lcell];
        }
        const unsigned long long tb = dbg ? __builtin_amdgcn_s_memtime() : 0;
        if (TO_LDS && sa + H > kRing && !lds_wait_ge(W.taken_out, (unsigned)(sa + H - kRing), W.group_abort))
            return false;
        if (dbg) t_out += __builtin_amdgcn_s_memtime() - tb;
        return true;
    };
    const auto half_end = [&](int sa) {
        if (FROM_LDS) {
            asm volatile("" ::: "memory");
            lds_store_lane0(W.taken_in, (unsigned)(sa + H));  // the H blocks are in registers and consumed
        }
        if (TO_LDS) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            lds_store_lane0(W.done_out, (unsigned)(sa + H));
        }
    };

    // markers consumed R - 3 steps after issue, in the next iteration, and the first iteration peeled -- as in
    // k_sor_exact (see there)
    constexpr int DL = R - 3, CA = H + DL - R, CB = DL;
    static_assert(CA >= 0 && CA < H && CB >= H && CB < R, "marker consumption points");
    unsigned ma = 0u, mb = 0u;
    const auto iteration = [&](int i, bool first) -> bool {
        const unsigned long long tc = dbg ? __builtin_amdgcn_s_memtime() : 0;
        if (!first && !wait_covered<1, 63>(X, pl, W.D, (i + 2) * R)) return false;
        if (dbg) t_cov += __builtin_amdgcn_s_memtime() - tc;
        const Polls pn = poll(W.D);
        const int s0 = i * R;
        if (!half_begin(s0)) return false;
        GSeg<R, 0, CA, DPP, FROM_LDS, TO_LDS>::run(A, T, L, W.ring_out, lane, om1, s0, c, lpd, S);
        if (!first) {  // marker A of the previous iteration
            asm volatile("" ::"v"(ma), "v"(S.duL), "v"(S.dvL) : "memory");
            if (uni(ma) != 0u) return false;
            publish(T, W.my_prog, (unsigned)min(ns, s0 - R + H));
        }
        GSeg<R, CA, H, DPP, FROM_LDS, TO_LDS>::run(A, T, L, W.ring_out, lane, om1, s0, c, lpd, S);
        ma = __hip_atomic_load(A.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // marker A: after step H - 1
        half_end(s0);
        if (!half_begin(s0 + H)) return false;
        GSeg<R, H, CB, DPP, FROM_LDS, TO_LDS>::run(A, T, L, W.ring_out, lane, om1, s0, c, lpd, S);
        if (!first) {  // marker B of the previous iteration
            asm volatile("" ::"v"(mb), "v"(S.duL), "v"(S.dvL) : "memory");
            if (uni(mb) != 0u) return false;
            publish(T, W.my_prog, (unsigned)min(ns, s0));
        }
        GSeg<R, CB, R, DPP, FROM_LDS, TO_LDS>::run(A, T, L, W.ring_out, lane, om1, s0, c, lpd, S);
        mb = __hip_atomic_load(A.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // marker B: after step R - 1
        half_end(s0 + H);
        pl = pn;
        return true;
    };
    if (!iteration(0, true)) return give_up();
    for (int i = 1; i < W.n_iter; ++i)
        if (!iteration(i, false)) return give_up();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish(T, W.my_prog, (unsigned)ns);
    if (dbg) {  // every lane stores the same
        unsigned long long* o = A.dbg + (size_t)W.task * 4;
        o[0] = __builtin_amdgcn_s_memtime() - t_start;
        o[1] = t_cov;
        o[2] = t_in;
        o[3] = t_out;
    }
}

template <int R, int M, bool DPP>
__global__ __launch_bounds__(64 * M) void k_sor_group(GroupArgs A) {
    static_assert(R >= 6 && R % 2 == 0 && R <= kRing, "half-iterations of R/2 steps; the ring holds two of them");
    __shared__ u32x4 ring[(M > 1 ? M - 1 : 1) * kRing * kLanes];
    __shared__ unsigned lds_done[M];   // [m]: steps of wave m whose cells are in its ring
    __shared__ unsigned lds_taken[M];  // [m]: steps of ring m the wave after has read
    __shared__ unsigned lds_abort;
    stamp_now(A.stamp);
    const unsigned lane = threadIdx.x & 63u;
    const int m = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform by construction
    lds_store_lane0((lds_word*)&lds_done[m], 0u);
    lds_store_lane0((lds_word*)&lds_taken[m], 0u);
    lds_store_lane0((lds_word*)&lds_abort, 0u);  // every wave writes the same 0 before the barrier
    __syncthreads();  // the only barrier: every wave is still here
    int g, b;
    if (A.xcd_affine) {
        const int x = blockIdx.x & 7, y = blockIdx.x >> 3, nb8 = (A.nb + 7) >> 3;
        const int gl = y / nb8;
        g = A.g0 + gl;
        b = (y - gl * nb8) * 8 + x;
        if (b >= A.nb) return;
    } else {
        const int gl = blockIdx.x / A.nb;
        g = A.g0 + gl;
        b = blockIdx.x - gl * A.nb;
    }
    const int k = g * M + m;
    if (k >= A.n_sor) return;
    const bool ghost = lane == 0 || lane == kLanes - 1;
    const bool from_lds = m > 0, to_planes = m == M - 1 || k == A.n_sor - 1;
    GWaveCtx W;
    lds_u32x4* const ring3 = (lds_u32x4*)ring;
    W.ring_in = ring3 + (m > 0 ? m - 1 : 0) * kRing * kLanes;
    W.ring_out = ring3 + (m < M - 1 ? m : 0) * kRing * kLanes;
    W.done_in = (lds_word*)&lds_done[m > 0 ? m - 1 : 0];
    W.taken_in = (lds_word*)&lds_taken[m > 0 ? m - 1 : 0];
    W.done_out = (lds_word*)&lds_done[m];
    W.taken_out = (lds_word*)&lds_taken[m];
    W.group_abort = (lds_word*)&lds_abort;
    W.task = k * A.nb + b;
    W.ns = A.ns;
    W.n_iter = (A.ns + R - 1) / R;

    Task T;
    const unsigned plane_bytes = (unsigned)(((size_t)A.npos * A.hp + kLanes) * 16u);
    constexpr unsigned kBlock = kLanes * 16u;
    const unsigned par_bytes = (unsigned)A.npos_d * (unsigned)A.nb * kBlock;
    const unsigned all_bytes = A.halo_off + (unsigned)A.n_sor * (unsigned)A.nb * (unsigned)A.npos_d * 16u;
    T.ra = __builtin_amdgcn_make_buffer_rsrc((void*)A.phi, 0, plane_bytes, 0x00020000);
    T.rb = __builtin_amdgcn_make_buffer_rsrc((void*)A.a1, 0, plane_bytes, 0x00020000);
    T.rc = __builtin_amdgcn_make_buffer_rsrc((void*)A.b1, 0, plane_bytes, 0x00020000);
    T.rd = __builtin_amdgcn_make_buffer_rsrc((void*)A.du, 0, all_bytes, 0x00020000);
    const int r0 = kBandRows * b - k - 1;
    GLane L;
    L.pos_c = (unsigned)A.hp * 16u;
    L.pos_d = (unsigned)A.nb * kBlock;
    const unsigned base = ((unsigned)(r0 + A.qt) * (unsigned)A.hp + (unsigned)(r0 + A.rt) + lane) * 16u;
    L.pa = base;
    L.pbc = ghost ? kOob : base;
    // planes alternate per group: group g writes plane g & 1 and reads plane (g - 1) & 1
    const unsigned mine = (unsigned)(g & 1) * par_bytes, prev = (unsigned)((g + 1) & 1) * par_bytes;
    L.st = to_planes ? mine + L.pos_d + (unsigned)b * kBlock + lane * 16u : kOob;
    const auto halo_row = [&](int kk, int bb) {  // byte offset of HALO[kk][bb][0]
        return A.halo_off + (unsigned)((kk * A.nb + bb) * A.npos_d) * 16u;
    };
    // lane 0: the row above, NEW value: lane 62 of (b-1, k), 63 steps ahead of ours -> halo position s + 64;
    // LDS-fed lanes: off; wave 0: own block of the previous group's plane, cell lane - 1   (selects, no branches)
    const unsigned pd_halo = b > 0 ? halo_row(k, b - 1) + 64u * 16u : kOob;
    const unsigned pd_plane = prev + L.pos_d + (unsigned)b * kBlock + (lane - 1u) * 16u;
    L.pd = lane == 0 ? pd_halo : (from_lds ? kOob : pd_plane);
    L.pd_step = lane == 0 ? 16u : (from_lds ? 0u : L.pos_d);
    L.hs = (lane == kLanes - 2 && b + 1 < A.nb) ? halo_row(k, b) + 16u : kOob;  // step s -> position s + 1
    const double om1 = ghost ? 1.0 : A.om1;

    T.rp = __builtin_amdgcn_make_buffer_rsrc((void*)A.prog, 0, (unsigned)A.n_sor * (unsigned)A.nb * kProgStride * 4u, 0x00020000);
    W.my_prog = lane == 0 ? (unsigned)(k * A.nb + b) * kProgStride * 4u : kOob;
    W.D.has_own = m == 0 && k > 0;  // inside a group the previous sweep arrives through LDS
    W.D.has_up = b > 0;
    W.D.has_dn2 = false;            // planes are read by the own band only, halo rows are never reused
    W.D.own = A.prog + ((size_t)(k - 1) * A.nb + b) * kProgStride;
    W.D.up = A.prog + ((size_t)k * A.nb + (b - 1)) * kProgStride;
    W.D.dn2 = W.D.own;
    if (from_lds) {
        if (to_planes)
            g_run_wave<R, DPP, true, false>(A, T, L, W, lane, om1);
        else
            g_run_wave<R, DPP, true, true>(A, T, L, W, lane, om1);
    } else {
        if (to_planes)
            g_run_wave<R, DPP, false, false>(A, T, L, W, lane, om1);
        else
            g_run_wave<R, DPP, false, true>(A, T, L, W, lane, om1);
    }
}

// ------------------------------------------------------------------------------------------------
// row-major modes
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cell_update(const double* __restrict__ phi, const double* __restrict__ xy,
                                            const double* __restrict__ a1, const double* __restrict__ a2,
                                            const double* __restrict__ b1, const double* __restrict__ b2,
                                            const double* ru, const double* rv, double* wu, double* wv, int i, int j,
                                            int H, int W, double nalpha, double om1) {
    const size_t o = (size_t)i * W + j;
    const double pc = phi[o];
    double s1 = 0.0, s2 = 0.0;
    if (j > 0) {
        const double w = phi[o - 1];
        s1 += w * ru[o - 1];
        s2 += w * rv[o - 1];
    }
    if (j < W - 1) {
        s1 += pc * ru[o + 1];
        s2 += pc * rv[o + 1];
    }
    if (i > 0) {
        const double w = phi[o - W];
        s1 += w * ru[o - W];
        s2 += w * rv[o - W];
    }
    if (i < H - 1) {
        s1 += pc * ru[o + W];
        s2 += pc * rv[o + W];
    }
    s1 *= nalpha;
    s2 *= nalpha;
    s1 += xy[o] * rv[o];
    const double nu = om1 * ru[o] + a1[o] * (b1[o] - s1);
    s2 += xy[o] * nu;
    const double nv = om1 * rv[o] + a2[o] * (b2[o] - s2);
    wu[o] = nu;
    wv[o] = nv;
}

// one colour of a red-black sweep: thread t of a row handles column 2t + ((i + colour) & 1)
__global__ void k_sor_redblack(const double* __restrict__ phi, const double* __restrict__ xy,
                               const double* __restrict__ a1, const double* __restrict__ a2,
                               const double* __restrict__ b1, const double* __restrict__ b2, double* du, double* dv,
                               int H, int W, double nalpha, double om1, int colour, Rect rc) {
    const int i = rc.y0 + blockIdx.y * 4 + threadIdx.y;
    if (i >= rc.y1) return;
    // cells with (i + j) % 2 == colour; the class of a cell does not depend on the region it is visited in
    const int j = rc.x0 + 2 * (blockIdx.x * 64 + threadIdx.x) + ((i + rc.x0 + colour) & 1);
    if (j >= rc.x1) return;
    cell_update(phi, xy, a1, a2, b1, b2, du, dv, du, dv, i, j, H, W, nalpha, om1);
}

__global__ void k_sor_jacobi(const double* __restrict__ phi, const double* __restrict__ xy,
                             const double* __restrict__ a1, const double* __restrict__ a2,
                             const double* __restrict__ b1, const double* __restrict__ b2,
                             const double* __restrict__ ru, const double* __restrict__ rv, double* __restrict__ wu,
                             double* __restrict__ wv, int H, int W, double nalpha, double om1) {
    const int i = blockIdx.y * 4 + threadIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    if (i >= H || j >= W) return;
    cell_update(phi, xy, a1, a2, b1, b2, ru, rv, wu, wv, i, j, H, W, nalpha, om1);
}


// ------------------------------------------------------------------------------------------------
// LDS-TILED, TEMPORALLY BLOCKED red-black / Jacobi solver (the kernel behind PAPOF_SOR_REDBLACK / _JACOBI and behind
// the multi-GPU tiles).  Same per-cell arithmetic as cell_update() above (src/OpticalFlow.cpp:458-505), other sweep order.
//
// A workgroup owns a REGION of 128 x (NW * RPT) cells = its core tile grown by a ghost ring `g` deep, keeps it on chip
// and runs g half-sweeps (Jacobi: g sweeps) on it in ONE launch: half-sweep m is exact on the region shrunk by m + 1
// cells from every edge that is not an image border (a cell one ring further out misses a neighbour one step earlier --
// the same ghost-zone algebra tiles.hip uses between GPUs), so after g half-sweeps the core tile is exact and is the
// only part written back.  The six coefficient planes are read ONCE per g half-sweeps (the naive kernels read them once
// per half-sweep) with full-line coalesced row loads (16 bytes per lane where rows are 16-byte aligned), never with
// stride 2; a 30-sweep solve is ceil(60 / g) launches instead of 60.
//   * ownership: lane t of wave w owns columns 2t, 2t+1 of the RPT rows w*RPT .. w*RPT+RPT-1 of the region: one cell of
//     each colour per row, so every lane has the same work in every half-sweep.  Its coefficients stay in REGISTERS for
//     the whole launch: per row three weights phi(j0-1), phi(j0), phi(j0+1) (the left weight of the right cell is the
//     centre weight of the left one; the upper weights are the row above's centre weights) and five doubles per cell
//     (imdxy, omega/diag_u, omega/diag_v, rhs_u, rhs_v): 13 * RPT + 2 doubles per lane.
//   * the unknowns live in LDS as 16-byte (du, dv) cells in two COLOUR-SPLIT arrays [colour][row][t]: a half-sweep of
//     colour c reads own / left / right / up / down = lds[c][r][t], lds[1-c][r][t-1+p], [r][t+p], [r-1][t], [r+1][t]
//     (p = parity of the cell's column in the lane's pair) -- consecutive lanes touch consecutive 16-byte cells, so every
//     ds_read_b128 / ds_write_b128 is bank-conflict free; rows are padded by one zero cell and the array by a zero row
//     above and below, cells outside the image hold (0, 0): border terms become +-0 added in the reference's order, no
//     predicates.  A colour-c half-sweep writes only colour-c cells, each by its owner, and reads colour 1-c cells: one
//     workgroup barrier per half-sweep.  (Jacobi: all new values of a sweep in registers, barrier, write, barrier.)
//   * results go to the OTHER pair of (du, dv) planes (neighbouring workgroups still read the old ghost cells): the host
//     alternates the pairs and arranges that the last launch lands in sp.du / sp.dv.  The first launch of a solve reads
//     no unknowns at all (du = dv = 0, src/OpticalFlow.cpp:452-453).
// ------------------------------------------------------------------------------------------------
typedef double f64x2 __attribute__((ext_vector_type(2)));

struct BlockedArgs {
    const double *phi, *xy, *a1, *a2, *b1, *b2;
    const double *su, *sv;  // unknowns before this launch (null: all zero)
    double *du, *dv;        // unknowns after it (another pair of planes)
    int H, W;
    Rect out;               // cells this launch must deliver (the whole plane on one GPU)
    int g;                  // half-sweeps (Jacobi: sweeps) of this launch = ghost depth
    int hs0;                // index of the first half-sweep of this launch within the solve (its parity = its colour)
    int cw, ch, ntx, nblk;  // core tile size, tiles per row, tiles
    int shift;              // 1: regions start one column further left, so that every region starts on an even column
    unsigned long long* stamp;  // as ExactArgs::stamp
    double nalpha, om1;
};

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <bool VEC>
__device__ __forceinline__ f64x2 load_pair(const double* __restrict__ plane, size_t o, bool ok0, bool ok1) {
    f64x2 r = {0.0, 0.0};
    if (VEC) {  // 16-byte aligned by construction (even row pitch, even first column)
        if (ok0) r = *reinterpret_cast<const f64x2*>(plane + o);
    } else {
        if (ok0) r.x = plane[o];
        if (ok1) r.y = plane[o + 1];
    }
    return r;
}

template <int MODE, int NW, int RPT, bool VEC>
__global__ __launch_bounds__(NW * 64) void k_sor_blocked(BlockedArgs A) {
    constexpr int RH = NW * RPT, S = kLanes + 1, CELLS = (RH + 2) * S + 2;
    __shared__ f64x2 lds[2][CELLS];
    stamp_now(A.stamp);
    const int t = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Workgroups are dealt round-robin over the 8 XCDs (observed; speed only): give each XCD a contiguous run of tiles so
    // that the ghost cells two neighbouring tiles both read are found in that XCD's L2.
    const int per = (A.nblk + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || tile >= A.nblk) return;  // whole workgroup, before any barrier
    const int by = tile / A.ntx, bx = tile - by * A.ntx;
    const int cx0 = A.out.x0 + bx * A.cw, cx1 = min(cx0 + A.cw, A.out.x1);
    const int cy0 = A.out.y0 + by * A.ch, cy1 = min(cy0 + A.ch, A.out.y1);
    const int rx0 = max(0, cx0 - A.g - A.shift), rx1 = min(A.W, cx1 + A.g);
    const int ry0 = max(0, cy0 - A.g), ry1 = min(A.H, cy1 + A.g);
    const int W = A.W;
    const int j0 = rx0 + 2 * t;      // this lane's columns: j0, j0 + 1
    const int r0 = w * RPT;          // first region row of this wave
    const int q0 = (ry0 + rx0 + r0) & 1;  // colour of cell (row r0, column j0)

    // ---- coefficients -> registers
    double ph[RPT][3], up[2], cxy[RPT][2], ca1[RPT][2], ca2[RPT][2], cb1[RPT][2], cb2[RPT][2];
    f64x2 u0[RPT], v0[RPT];
    {
        const int iu = ry0 + r0 - 1;  // the row above this wave's rows: only its weights are needed
        const bool rok = iu >= 0 && iu < ry1;
        const f64x2 pu = load_pair<VEC>(A.phi, (size_t)max(iu, 0) * W + j0, rok && j0 < rx1, rok && j0 + 1 < rx1);
        up[0] = pu.x;
        up[1] = pu.y;
    }
    static_for<RPT>([&](auto RR) __attribute__((always_inline)) {
        constexpr int rr = decltype(RR)::value;
        const int i = ry0 + r0 + rr;
        const bool rok = i < ry1, ok0 = rok && j0 < rx1, ok1 = rok && j0 + 1 < rx1;
        const size_t o = (size_t)min(i, A.H - 1) * W + j0;
        const f64x2 p = load_pair<VEC>(A.phi, o, ok0, ok1);
        const f64x2 x = load_pair<VEC>(A.xy, o, ok0, ok1);
        const f64x2 q1 = load_pair<VEC>(A.a1, o, ok0, ok1);
        const f64x2 q2 = load_pair<VEC>(A.a2, o, ok0, ok1);
        const f64x2 e1 = load_pair<VEC>(A.b1, o, ok0, ok1);
        const f64x2 e2 = load_pair<VEC>(A.b2, o, ok0, ok1);
        u0[rr] = f64x2{0.0, 0.0};
        v0[rr] = f64x2{0.0, 0.0};
        if (A.su) {  // wave-uniform
            u0[rr] = load_pair<VEC>(A.su, o, ok0, ok1);
            v0[rr] = load_pair<VEC>(A.sv, o, ok0, ok1);
        }
        ph[rr][1] = p.x;
        ph[rr][2] = p.y;
        cxy[rr][0] = x.x;
        cxy[rr][1] = x.y;
        ca1[rr][0] = q1.x;
        ca1[rr][1] = q1.y;
        ca2[rr][0] = q2.x;
        ca2[rr][1] = q2.y;
        cb1[rr][0] = e1.x;
        cb1[rr][1] = e1.y;
        cb2[rr][0] = e2.x;
        cb2[rr][1] = e2.y;
    });
    // the weight left of the lane's left cell is the centre weight of the previous lane's right cell (lane 0: column
    // rx0 - 1, needed only where column rx0 is updated, i.e. at the image border, where that term is +-0 anyway)
    static_for<RPT>([&](auto RR) __attribute__((always_inline)) {
        constexpr int rr = decltype(RR)::value;
        const double left = __shfl_up(ph[rr][2], 1);
        ph[rr][0] = t == 0 ? 0.0 : left;
    });

    // ---- unknowns -> LDS (colour-split), pads -> 0
    const int idx0 = (r0 + 1) * S + t + 1;  // cell index of (row r0, this lane) in either colour array
    static_for<RPT>([&](auto RR) __attribute__((always_inline)) {
        constexpr int rr = decltype(RR)::value;
        const int c0 = (q0 + rr) & 1;  // colour of the lane's LEFT cell in this row
        lds[c0][idx0 + rr * S] = f64x2{u0[rr].x, v0[rr].x};
        lds[c0 ^ 1][idx0 + rr * S] = f64x2{u0[rr].y, v0[rr].y};
        if (t == 0) {  // the pad cell in front of this row (= behind the previous one)
            lds[0][idx0 + rr * S - 1] = f64x2{0.0, 0.0};
            lds[1][idx0 + rr * S - 1] = f64x2{0.0, 0.0};
        }
    });
    if (w == 0) {  // pad row above the region: indices 0 .. S-1
        lds[0][t] = f64x2{0.0, 0.0};
        lds[1][t] = f64x2{0.0, 0.0};
        if (t == 0) {
            lds[0][S - 1] = f64x2{0.0, 0.0};
            lds[1][S - 1] = f64x2{0.0, 0.0};
        }
    }
    if (w == NW - 1) {  // pad row below: indices (RH+1)*S .. (RH+1)*S + S (its own pad cell and the trailing one included)
        lds[0][(RH + 1) * S + t] = f64x2{0.0, 0.0};
        lds[1][(RH + 1) * S + t] = f64x2{0.0, 0.0};
        if (t < 3) {
            lds[0][(RH + 1) * S + 64 + t - 1 + 1] = f64x2{0.0, 0.0};
            lds[1][(RH + 1) * S + 64 + t - 1 + 1] = f64x2{0.0, 0.0};
        }
    }
    __syncthreads();

    // ---- the sweeps
    const int ox0 = rx0 > 0, ox1 = rx1 < A.W, oy0 = ry0 > 0, oy1 = ry1 < A.H;  // 1 where the region edge is open
    const int rw = rx1 - rx0, rh = ry1 - ry0;
    const double nalpha = A.nalpha, om1 = A.om1;
    // one cell: row r0 + rr, column pair member p, colour array `cc`; returns the new (du, dv)
    const auto cell = [&](auto RR, auto PP, int cc) __attribute__((always_inline)) -> f64x2 {
        constexpr int rr = decltype(RR)::value, p = decltype(PP)::value;
        const int idx = idx0 + rr * S;
        const f64x2* __restrict__ Lo = lds[cc ^ 1];
        const f64x2 own = lds[cc][idx];
        const f64x2 L = Lo[idx - 1 + p], R = Lo[idx + p], U = Lo[idx - S], D = Lo[idx + S];
        const double pc = ph[rr][1 + p], wl = ph[rr][p];
        double wu;
        if constexpr (rr > 0)
            wu = ph[rr - 1][1 + p];
        else
            wu = up[p];
        double s1 = wl * L.x, s2 = wl * L.y;
        s1 += pc * R.x;
        s2 += pc * R.y;
        s1 += wu * U.x;
        s2 += wu * U.y;
        s1 += pc * D.x;
        s2 += pc * D.y;
        s1 *= nalpha;
        s2 *= nalpha;
        s1 += cxy[rr][p] * own.y;
        const double nu = om1 * own.x + ca1[rr][p] * (cb1[rr][p] - s1);
        s2 += cxy[rr][p] * nu;
        const double nv = om1 * own.y + ca2[rr][p] * (cb2[rr][p] - s2);
        return f64x2{nu, nv};
    };
    if (MODE == PAPOF_SOR_REDBLACK) {
        // PAR = parity of (q0 + colour): the member of the lane's pair that has the half-sweep's colour in row rr is
        // p = (PAR + rr) & 1 -- a compile-time constant inside each of the two instantiations
        const auto half = [&](auto PARC, int m, int c) __attribute__((always_inline)) {
            constexpr int PAR = decltype(PARC)::value;
            const int fx0 = ox0 * (m + 1), fx1 = rw - ox1 * (m + 1), fy0 = oy0 * (m + 1), fy1 = rh - oy1 * (m + 1);
            if (r0 + RPT <= fy0 || r0 >= fy1) {  // wave-uniform: none of this wave's rows is inside the frame any more
                __syncthreads();
                return;
            }
            if constexpr (NW < 16) {
                // IL rows are computed in ONE basic block (the frame test only predicates the stores): the cell updates are
                // independent 14-deep fp64 chains, and interleaved they fill the issue slots a single chain leaves empty
                // -- a per-row branch kept them apart (SQ counters: 31 % of a wave's cycles were issue stalls).  Measured:
                // -10 % per solve for the 8-wave shapes (256-register budget: all rows at once); 12 waves (168 registers)
                // interleave two rows at a time.
                constexpr int IL = NW == 12 ? 2 : RPT;
                static_assert(RPT % IL == 0, "row groups");
                static_for<RPT / IL>([&](auto GG) __attribute__((always_inline)) {
                    constexpr int g0 = decltype(GG)::value * IL;
                    f64x2 nw[IL];
                    static_for<IL>([&](auto QQ) __attribute__((always_inline)) {
                        constexpr int rr = g0 + decltype(QQ)::value;
                        nw[decltype(QQ)::value] = cell(std::integral_constant<int, rr>{}, std::integral_constant<int, (PAR + rr) & 1>{}, c);
                    });
                    static_for<IL>([&](auto QQ) __attribute__((always_inline)) {
                        constexpr int rr = g0 + decltype(QQ)::value;
                        constexpr int p = (PAR + rr) & 1;
                        const int r = r0 + rr, lc = 2 * t + p;
                        if (r >= fy0 && r < fy1 && lc >= fx0 && lc < fx1) lds[c][idx0 + rr * S] = nw[decltype(QQ)::value];
                    });
                });
            } else {
                // 16 waves have 128 registers each, 82 of them coefficients: interleaving three rows spills (measured
                // +40 %); one row at a time, and four waves per SIMD fill each other's stalls instead
                static_for<RPT>([&](auto RR) __attribute__((always_inline)) {
                    constexpr int rr = decltype(RR)::value;
                    constexpr int p = (PAR + rr) & 1;
                    const int r = r0 + rr;
                    if (r >= fy0 && r < fy1) {  // wave-uniform
                        const f64x2 nw = cell(RR, std::integral_constant<int, p>{}, c);
                        const int lc = 2 * t + p;
                        if (lc >= fx0 && lc < fx1) lds[c][idx0 + rr * S] = nw;
                    }
                });
            }
            __syncthreads();
        };
        for (int m = 0; m < A.g; m++) {
            const int c = (A.hs0 + m) & 1;
            if ((q0 + c) & 1)
                half(std::integral_constant<int, 1>{}, m, c);
            else
                half(std::integral_constant<int, 0>{}, m, c);
        }
    } else {  // Jacobi: every cell from the previous sweep's values
        for (int m = 0; m < A.g; m++) {
            const int fx0 = ox0 * (m + 1), fx1 = rw - ox1 * (m + 1), fy0 = oy0 * (m + 1), fy1 = rh - oy1 * (m + 1);
            f64x2 nw[RPT][2];
            static_for<RPT>([&](auto RR) __attribute__((always_inline)) {
                constexpr int rr = decltype(RR)::value;
                const int c0 = (q0 + rr) & 1;  // colour array of the left cell
                nw[rr][0] = cell(RR, std::integral_constant<int, 0>{}, c0);
                nw[rr][1] = cell(RR, std::integral_constant<int, 1>{}, c0 ^ 1);
            });
            __syncthreads();
            static_for<RPT>([&](auto RR) __attribute__((always_inline)) {
                constexpr int rr = decltype(RR)::value;
                const int c0 = (q0 + rr) & 1, r = r0 + rr;
                if (r >= fy0 && r < fy1) {
                    if (2 * t >= fx0 && 2 * t < fx1) lds[c0][idx0 + rr * S] = nw[rr][0];
                    if (2 * t + 1 >= fx0 && 2 * t + 1 < fx1) lds[c0 ^ 1][idx0 + rr * S] = nw[rr][1];
                }
            });
            __syncthreads();
        }
    }

    // ---- core tile -> the other pair of planes (own cells only: no barrier needed after the last write)
    static_for<RPT>([&](auto RR) __attribute__((always_inline)) {
        constexpr int rr = decltype(RR)::value;
        const int i = ry0 + r0 + rr;
        if (i >= cy0 && i < cy1) {  // wave-uniform
            const int c0 = (q0 + rr) & 1;
            const f64x2 a = lds[c0][idx0 + rr * S], b = lds[c0 ^ 1][idx0 + rr * S];
            const size_t o = (size_t)i * W + j0;
            const bool in0 = j0 >= cx0 && j0 < cx1, in1 = j0 + 1 >= cx0 && j0 + 1 < cx1;
            if (VEC && in0 && in1) {
                *reinterpret_cast<f64x2*>(A.du + o) = f64x2{a.x, b.x};
                *reinterpret_cast<f64x2*>(A.dv + o) = f64x2{a.y, b.y};
            } else {
                if (in0) {
                    A.du[o] = a.x;
                    A.dv[o] = a.y;
                }
                if (in1) {
                    A.du[o + 1] = b.x;
                    A.dv[o + 1] = b.y;
                }
            }
        }
    });
}


// ------------------------------------------------------------------------------------------------
// TINY PLANES, exact order: the whole solve inside ONE workgroup (k_sor_tiny).
//
// Deep pyramids end in levels of a few thousand cells (the reference's own test matrix runs 15 levels: 1920x1080 goes
// down to 34x19, with 60-72 sweeps and 17-21 outer iterations there: src/OpticalFlow.cpp:823, Code/Serial/TestSuite.py:91).
// On such a plane the task pipeline above is nothing but its chain of hand-offs -- every solve costs ~(sweeps x 3.4 us),
// 170-290 us whatever the size -- so those levels dominated the call.  Here the plane never leaves the CU:
//   * a TILE is C consecutive cells of one row; tile (row r, segment q) of sweep k runs at tile-time T = r + q + 2k: its
//     left and upper neighbours (sweep k) ran at T - 1, its right and lower neighbours (sweep k - 1) at T - 1 as well, its
//     own previous sweep at T - 2 -- the hyperplane schedule of the file header on tiles, so the results are those of the
//     in-place lexicographic sweeps (src/OpticalFlow.cpp:458-505) exactly; tiles that share a T never touch each other's
//     cells (r + q has one parity per T);
//   * a lane owns TWO neighbouring tiles of a row, one of each parity -- it works in every tile-time -- and keeps their
//     coefficients in REGISTERS for the whole solve (per cell phi, imdxy, the hoisted diagonals, the right-hand sides, and
//     the weight of the cell above; 14 C + 2 doubles per lane), loaded once with plain row loads;
//   * the unknowns live in LDS, one 16-byte (du, dv) cell each, in place, framed by a ring of (0, 0) cells: a neighbour that
//     does not exist contributes w * 0 -- the reference's conditional terms as +-0 added in the same order, as in the
//     kernels above; a tile reads its 3 C + 2 neighbour cells, updates its C cells left to right (left-new in registers)
//     and writes them back; ONE workgroup barrier per tile-time orders everything -- no counters, no polling, no
//     inter-workgroup visibility to argue about;
//   * R + Q + 2 K - 2 tile-times, each bound by the vector issue of the one CU (~40 instructions per cell and wave).  Measured
//     (MI355X, inside a 15-level call): 34x19 x 72 sweeps 60 us instead of 287, 60x33 x 66 sweeps 98 instead of 263, 108x60 x
//     60 sweeps 219 instead of 249; a 240x135 pair on 15 levels 51.5 -> 18.9 ms.
// Capacity: registers (7 doubles per cell) -- ~5 k cells (tiles of at most 10 cells; wider ones are no faster than the task
// pipeline).  Operands are the
// ROW-MAJOR planes the red-black / Jacobi modes use (k_assemble writes them, k_update_warp_phi<false> reads the result).
// ------------------------------------------------------------------------------------------------
struct TinyArgs {
    const double *phi, *xy, *a1, *a2, *b1, *b2;
    double *du, *dv;
    int H, W, K;
    int Q, QP;  // segments per row, lanes per row = ceil(Q / 2)
    double nalpha, om1;
    size_t bstride;  // batched solves: doubles between the planes of consecutive frame pairs (blockIdx.x = the pair); else 0
};

template <int C, int NW>  // NW: the most waves a launch may have (register budget); a launch has as many as its lanes need
__global__ __launch_bounds__(NW * 64) void k_sor_tiny(TinyArgs A_in) {
    TinyArgs A = A_in;
    {
        const size_t o = (size_t)blockIdx.x * A.bstride;
        A.phi += o;
        A.xy += o;
        A.a1 += o;
        A.a2 += o;
        A.b1 += o;
        A.b2 += o;
        A.du += o;
        A.dv += o;
    }
    // LDS layout: cell (row r, column j) -> ((r + 1) * 2C + j % 2C) * (QP + 2) + j / 2C + 1: within a row the cells that the
    // lanes of a row address with the SAME register index (same position in a lane's pair of tiles) are contiguous, so a
    // wave's 16-byte accesses fall on consecutive cells -- no bank conflicts for any C (row-major cells would be 2C cells
    // apart: 4-way conflicts at C = 2).  Row 0 / row H + 1 and the lane slots 0 / QP + 1 are the frame of (0, 0) cells.
    extern __shared__ f64x2 tiny_lds[];
    const int H = A.H, W = A.W, QP = A.QP, LS = QP + 2, RS = 2 * C * LS;  // cells per position-row, per image row
    const int g = threadIdx.x, nthreads = blockDim.x;
    for (int c = g; c < (H + 2) * RS; c += nthreads) tiny_lds[c] = f64x2{0.0, 0.0};  // du = dv = 0 (:452-453)
    const int r = g / QP, m = g - r * QP;
    const bool lane_on = r < H;
    // tile 0 of the lane has r + q even, tile 1 odd: at even tile-times every lane works on its tile 0, at odd ones on tile 1
    const int q0 = 2 * m + (r & 1), q1 = 2 * m + 1 - (r & 1);
    // ---- coefficients -> registers
    double cphi[2][C], cxy[2][C], ca1[2][C], ca2[2][C], cb1[2][C], cb2[2][C], cup[2][C], cleft[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int q = t ? q1 : q0, j0 = q * C;
        const bool tile_on = lane_on && q < A.Q;
        cleft[t] = (tile_on && j0 > 0 && j0 - 1 < W) ? A.phi[(size_t)r * W + j0 - 1] : 0.0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int j = j0 + c;
            const bool on = tile_on && j < W;
            const size_t o = on ? (size_t)r * W + j : 0;
            cphi[t][c] = on ? A.phi[o] : 0.0;
            cxy[t][c] = on ? A.xy[o] : 0.0;
            ca1[t][c] = on ? A.a1[o] : 0.0;
            ca2[t][c] = on ? A.a2[o] : 0.0;
            cb1[t][c] = on ? A.b1[o] : 0.0;
            cb2[t][c] = on ? A.b2[o] : 0.0;
            cup[t][c] = (on && r > 0) ? A.phi[o - W] : 0.0;
        }
    }
    __syncthreads();
    const double nalpha = A.nalpha, om1 = A.om1;
    const int n_t = H + A.Q + 2 * A.K - 2;
    // one tile: cells j0 .. j0 + C - 1 of row r, left to right
    const auto tile = [&](auto TT, int T) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value;
        const int q = t ? q1 : q0;
        const int d = T - r - q;  // = 2 k
        const bool on = lane_on && q < A.Q && d >= 0 && (d >> 1) < A.K;
        if (on) {
            const int j0 = q * C, half = q - 2 * m;  // which half of the lane's pair of tiles: positions half * C ..
            f64x2* const own0 = tiny_lds + (size_t)((r + 1) * 2 * C + half * C) * LS + (m + 1);  // cell (r, j0)
            // the cell left of the tile / right of it: the neighbouring position of this lane's slot, or -- at the ends of the
            // pair -- the last / first position of the neighbouring lane's slot (slots 0 and QP + 1: the zero frame)
            const f64x2* const lcell = half ? own0 - LS : own0 + (2 * C - 1) * LS - 1;
            const f64x2* const rcell = half ? own0 - C * LS + 1 : own0 + C * LS;
            f64x2 left = *lcell;
            double wl = cleft[t];
            f64x2 own = own0[0];
            // the neighbour cells of cell c + 1 are requested BEFORE cell c is computed and written (distinct cells), so their
            // LDS latency hides behind the arithmetic
            f64x2 right = C > 1 ? own0[LS] : *rcell, up = own0[-RS], down = own0[RS];
#pragma unroll
            for (int c = 0; c < C; c++) {
                f64x2 nright = right, nup = up, ndown = down;
                if (c + 1 < C) {
                    nright = c + 2 < C ? own0[(c + 2) * LS] : *rcell;
                    nup = own0[(c + 1) * LS - RS];
                    ndown = own0[(c + 1) * LS + RS];
                }
                const double pc = cphi[t][c];
                // the terms in the reference's order: left, right, up, down (src/OpticalFlow.cpp:468-495)
                double s1 = wl * left.x, s2 = wl * left.y;
                s1 += pc * right.x;
                s2 += pc * right.y;
                s1 += cup[t][c] * up.x;
                s2 += cup[t][c] * up.y;
                s1 += pc * down.x;
                s2 += pc * down.y;
                s1 *= nalpha;
                s2 *= nalpha;
                s1 += cxy[t][c] * own.y;
                const double nu = om1 * own.x + ca1[t][c] * (cb1[t][c] - s1);
                s2 += cxy[t][c] * nu;
                const double nv = om1 * own.y + ca2[t][c] * (cb2[t][c] - s2);
                left = f64x2{nu, nv};
                wl = pc;
                own = right;
                // written at once: nobody reads this cell in this tile-time (tiles that share a T are not neighbours, and the
                // next cell takes it from `left`).  Columns beyond the plane stay (0, 0): they are somebody's right neighbour
                if (j0 + c < W) own0[c * LS] = left;
                right = nright;
                up = nup;
                down = ndown;
            }
        }
    };
    for (int T = 0; T < n_t; T++) {
        if (T & 1)
            tile(std::integral_constant<int, 1>{}, T);
        else
            tile(std::integral_constant<int, 0>{}, T);
        __syncthreads();
    }
    for (int c = g; c < H * W; c += nthreads) {
        const int i = c / W, j = c - i * W;
        const f64x2 v = tiny_lds[(size_t)((i + 1) * 2 * C + j % (2 * C)) * LS + j / (2 * C) + 1];
        A.du[c] = v.x;
        A.dv[c] = v.y;
    }
}

}  // namespace

// One colour of a red-black sweep on a region of the (row-major) operand planes.  The whole plane on one GPU; a tile
// grown by its remaining ghost depth when a frame is sharded (tiles.hip).
int sor_redblack_halfsweep(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int colour,
                           const Rect& r) {
    if (sp.skew) return PAPOF_EINVAL;
    if (r.empty()) return PAPOF_OK;
    const dim3 grid(((r.w() + 1) / 2 + 63) / 64, (r.h() + 3) / 4), block(64, 4);
    hipLaunchKernelGGL(k_sor_redblack, grid, block, 0, h->stream, sp.phi, sp.xy, sp.a1, sp.a2, sp.b1, sp.b2, sp.du,
                       sp.dv, H, W, -alpha, 1 - omega, colour, r);
    PAPOF_HIP(hipGetLastError());
    return PAPOF_OK;
}

// Tasks (waves) of the exact-order kernels that one launch may hold.  A task spins on progress counters written by tasks
// of lower block index, which is deadlock-free only while every task of the launch is resident or the hardware
// dispatches blocks in index order -- observed behaviour, not a HIP guarantee.  So a launch never exceeds what the chip
// keeps resident: 8 one-wave workgroups per CU (the kernels need 140-162 VGPRs = 3 waves per SIMD = 12 per CU, the
// margin is for whatever else runs beside the solve); larger solves become consecutive launches over ranges of sweeps --
// the ping-pong planes and the counters carry the state across the launch boundary.  PAPOF_SOR_RESIDENT (read when the
// handle is created) overrides the bound: the tests use it to force many launches per solve.
static int fill_pairs(papof_handle* h, void* base, size_t stride_bytes, size_t bytes, int batch);  // (defined with the strips' helpers below)
static int resident_tasks(const papof_handle* h) {
    if (h->sor_resident > 0) return h->sor_resident;
    return std::max(64, (h->cu_count > 0 ? h->cu_count : 256) * 8);
}

// ---- the blocked solver's host side -------------------------------------------------------------------
// Region shapes (waves per workgroup x rows per lane; the region is 128 columns x NW * RPT rows).  Registers hold
// 13 * RPT + 2 doubles of coefficients per lane and LDS 2 * 16 B per cell, so 128 x 48 cells is what one CU can keep
// (8 waves x 6 rows: 160 + temporaries of 256 VGPRs, 104 KiB LDS); small planes get smaller regions = more workgroups.
struct BlockedShape {
    int nw, rpt;
};
static BlockedShape blocked_shape_any(const papof_handle* h, int H, int W) {
    if (h->rb_shape == 1) return {8, 6};
    if (h->rb_shape == 2) return {8, 4};
    if (h->rb_shape == 3) return {4, 6};
    if (h->rb_shape == 4) return {16, 3};
    if (h->rb_shape == 5) return {12, 4};
    // measured (tools/rb_sweep.py, profiles/r02_rb_shape_depth_sweep*.txt): 12 waves x 4 rows (two rows interleaved per
    // wave, three waves per SIMD) and 16 x 3 beat 8 x 6 on every plane, the smaller 8 x 4 region wins below ~300 k cells
    // (more workgroups for the same plane)
    if (W <= 128 && H <= 24) return {4, 6};
    if (W <= 128 && H <= 32) return {8, 4};
    if (W <= 128 && H <= 48) return {8, 6};  // a plane that fits ONE region: one workgroup, one launch, no ghost cells
    if ((size_t)H * W >= (size_t)300 * 1000) return {12, 4};
    return {8, 4};
}
static BlockedShape blocked_shape(const papof_handle* h, int mode, int H, int W) {
    const BlockedShape bs = blocked_shape_any(h, H, W);
    // Jacobi keeps a sweep's new values of all 2 * RPT cells in registers beside the coefficients: no 8 x 6 variant
    if (mode == PAPOF_SOR_JACOBI && ((bs.nw == 8 && bs.rpt == 6) || bs.nw >= 12)) return {8, 4};
    return bs;
}

int sor_blocked_depth(const papof_handle* h, int mode, int H, int W) {
    if (h->rb_depth > 0) return h->rb_depth;
    const BlockedShape bs = blocked_shape(h, mode, H, W);
    const int rh = bs.nw * bs.rpt;
    if (mode == PAPOF_SOR_JACOBI) return rh >= 48 ? 8 : (rh >= 32 ? 6 : 4);
    return rh >= 32 ? 10 : 6;  // half-sweeps: 60 = 6 x 10
}

// One launch: `g` half-sweeps (Jacobi: sweeps) starting with half-sweep `hs0` of the solve, delivering the cells of `out`;
// (su, sv) -> (du, dv) must be different pairs of planes; su == null: the unknowns are zero before the launch.
int sor_blocked_launch(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int mode, int g,
                       int hs0, const Rect& out, const double* su, const double* sv, double* du, double* dv) {
    if (sp.skew || g < 1 || (mode != PAPOF_SOR_REDBLACK && mode != PAPOF_SOR_JACOBI)) return PAPOF_EINVAL;
    if (out.empty()) return PAPOF_OK;
    if (su == du || sv == dv || !du || !dv) return PAPOF_EINVAL;
    const BlockedShape bs = blocked_shape(h, mode, out.h(), out.w());  // by the size of what is solved (a tile: tiles.hip)
    const int RW = 2 * kLanes, RH = bs.nw * bs.rpt;
    BlockedArgs A;
    A.phi = sp.phi;
    A.xy = sp.xy;
    A.a1 = sp.a1;
    A.a2 = sp.a2;
    A.b1 = sp.b1;
    A.b2 = sp.b2;
    A.su = su;
    A.sv = sv;
    A.du = du;
    A.dv = dv;
    A.H = H;
    A.W = W;
    A.out = out;
    A.g = g;
    A.hs0 = hs0;
    A.nalpha = -alpha;
    A.om1 = 1 - omega;
    A.stamp = nullptr;
    // Core tile: the region minus the ghost ring; where `out` reaches an image border no ring is needed on that side, so a
    // plane that fits one region is ONE workgroup with no ghost cells at all, whatever g.
    const bool span_x = out.x0 == 0 && out.x1 == W && W <= RW, span_y = out.y0 == 0 && out.y1 == H && H <= RH;
    // 16-byte row loads / stores need every region to start on an even column (of an even-pitched plane): tiles are an
    // even number of columns wide, and where out.x0 - g is odd the regions start one column further left (shift).
    A.shift = span_x ? 0 : ((out.x0 - g) & 1);
    A.cw = span_x ? W : RW - 2 * g - 2 * A.shift;
    A.ch = span_y ? H : RH - 2 * g;
    if (A.cw < (span_x ? 1 : 2) || A.ch < 1) return PAPOF_EINVAL;  // g too deep for this region shape
    A.ntx = (out.w() + A.cw - 1) / A.cw;
    const int nty = (out.h() + A.ch - 1) / A.ch;
    A.nblk = A.ntx * nty;
    const bool vec = (W & 1) == 0;
    const dim3 grid(8 * ((A.nblk + 7) / 8)), block(bs.nw * kLanes);
#define PAPOF_BLOCKED(MODE, NW, RPT)                                                                       \
    do {                                                                                                   \
        if (vec)                                                                                           \
            hipLaunchKernelGGL((k_sor_blocked<MODE, NW, RPT, true>), grid, block, 0, h->stream, A);       \
        else                                                                                               \
            hipLaunchKernelGGL((k_sor_blocked<MODE, NW, RPT, false>), grid, block, 0, h->stream, A);      \
    } while (0)
    if (mode == PAPOF_SOR_REDBLACK) {
        if (bs.nw == 8 && bs.rpt == 6)
            PAPOF_BLOCKED(PAPOF_SOR_REDBLACK, 8, 6);
        else if (bs.nw == 8 && bs.rpt == 4)
            PAPOF_BLOCKED(PAPOF_SOR_REDBLACK, 8, 4);
        else if (bs.nw == 4 && bs.rpt == 6)
            PAPOF_BLOCKED(PAPOF_SOR_REDBLACK, 4, 6);
        else if (bs.nw == 12 && bs.rpt == 4)
            PAPOF_BLOCKED(PAPOF_SOR_REDBLACK, 12, 4);
        else
            PAPOF_BLOCKED(PAPOF_SOR_REDBLACK, 16, 3);
    } else {
        if (bs.nw == 4 && bs.rpt == 6)
            PAPOF_BLOCKED(PAPOF_SOR_JACOBI, 4, 6);
        else
            PAPOF_BLOCKED(PAPOF_SOR_JACOBI, 8, 4);
    }
#undef PAPOF_BLOCKED
    PAPOF_HIP(hipGetLastError());
    return PAPOF_OK;
}


// ---- the tiny-plane solver's host side ------------------------------------------------------------------
// Shapes: (cells per tile, waves).  Registers bound the tile: 28 C + 4 for the coefficients of a lane's two tiles, beside
// ~40 of working set, within 128 / 256 / 512 registers at 16 / 8 / 4 waves per workgroup.
struct TinyShape {
    int c, nw;  // cells per tile; most waves the instantiation may be launched with
    int waves;  // waves this plane needs: ceil(rows x lanes per row / 64)
};
static bool tiny_shape(int H, int W, int K, TinyShape& best) {
    // Narrow tiles on many waves win while the lanes fit (tools/tiny_sweep.py on MI355X, profiles/r03_tiny_sweep.txt: e.g.
    // 60x33 x 66 sweeps 91 / 97 / 104 / 113 / 124 / 177 us at C = 1 / 2 / 3 / 4 / 6 / 8): the narrowest shape that holds the
    // plane is taken.  Tiles of 14 cells (the only shape that holds ~6.5 k cells) run as long as the task pipeline does
    // (108x60 x 60 sweeps: 249 us either way) and are left to it.
    static const TinyShape cand[] = {{1, 16, 0}, {2, 16, 0}, {3, 12, 0}, {4, 8, 0}, {5, 8, 0}, {6, 8, 0}, {8, 4, 0}, {10, 4, 0}};
    static const int force_c = std::getenv("PAPOF_TINY_C") ? std::atoi(std::getenv("PAPOF_TINY_C")) : 0;  // A/B: force C where it fits
    (void)K;
    bool found = false;
    for (TinyShape t : cand) {
        const int Q = (W + t.c - 1) / t.c, QP = (Q + 1) / 2;
        if ((long long)H * QP > t.nw * 64) continue;
        if ((size_t)(H + 2) * 2 * t.c * (QP + 2) * 16 > (size_t)150 * 1024) continue;  // LDS: the framed plane of 16-byte cells
        t.waves = (H * QP + 63) / 64;
        if (!found || force_c == t.c) best = t;
        found = true;
        if (force_c <= 0) break;
    }
    return found;
}

// May a height x width plane be solved in exact order by k_sor_tiny?  (PAPOF_SOR_TINY=0 switches the path off: A/B)
bool sor_tiny_fits(const papof_handle* h, int H, int W, int n_sor) {
    static const bool off = std::getenv("PAPOF_SOR_TINY") && std::atoi(std::getenv("PAPOF_SOR_TINY")) == 0;
    (void)h;
    TinyShape t;
    return !off && H >= 1 && W >= 1 && n_sor >= 1 && tiny_shape(H, W, n_sor, t);
}

static int sor_tiny_solve(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int n_sor,
                          int batch = 1, size_t bstride = 0) {
    TinyShape t;
    if (sp.skew || !tiny_shape(H, W, n_sor, t)) return PAPOF_EINVAL;
    TinyArgs A;
    A.phi = sp.phi;
    A.xy = sp.xy;
    A.a1 = sp.a1;
    A.a2 = sp.a2;
    A.b1 = sp.b1;
    A.b2 = sp.b2;
    A.du = sp.du;
    A.dv = sp.dv;
    A.H = H;
    A.W = W;
    A.K = n_sor;
    A.Q = (W + t.c - 1) / t.c;
    A.QP = (A.Q + 1) / 2;
    A.nalpha = -alpha;
    A.om1 = 1 - omega;
    A.bstride = batch > 1 ? bstride : 0;
    const size_t lds = (size_t)(H + 2) * 2 * t.c * (A.QP + 2) * 16;
#define PAPOF_TINY(CC, NWW)                                                                                           \
    do {                                                                                                              \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            PAPOF_HIP(hipFuncSetAttribute((const void*)k_sor_tiny<CC, NWW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                          150 * 1024));                                                               \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL((k_sor_tiny<CC, NWW>), dim3(batch), dim3(t.waves * 64), lds, h->stream, A);                    \
    } while (0)
    if (t.c == 1)
        PAPOF_TINY(1, 16);
    else if (t.c == 2)
        PAPOF_TINY(2, 16);
    else if (t.c == 3)
        PAPOF_TINY(3, 12);
    else if (t.c == 4)
        PAPOF_TINY(4, 8);
    else if (t.c == 5)
        PAPOF_TINY(5, 8);
    else if (t.c == 6)
        PAPOF_TINY(6, 8);
    else if (t.c == 8)
        PAPOF_TINY(8, 4);
    else
        PAPOF_TINY(10, 4);
#undef PAPOF_TINY
    PAPOF_HIP(hipGetLastError());
    return PAPOF_OK;
}

// `units` half-sweeps (Jacobi: sweeps) from zero on the whole plane: ceil(units / depth) launches of nearly equal depth,
// alternating between the two pairs of planes so that the LAST one writes (sp.du, sp.dv).
int sor_group_size(const papof_handle* h, int H, int W, int n_sor);
static int sor_fuse_size(const papof_handle* h, int H, int W, int n_sor, int group);

struct BlockedPlan {
    int q, n_launch, base, rem;  // launch l runs q * (base + (l < rem)) half-sweeps (Jacobi: sweeps)
};
static BlockedPlan blocked_plan(const papof_handle* h, int mode, int H, int W, int units) {
    const BlockedShape bs = blocked_shape(h, mode, H, W);
    const bool one_block = W <= 2 * kLanes && H <= bs.nw * bs.rpt;
    BlockedPlan p;
    // Red-black: launches hold whole sweeps (an even number of half-sweeps) whenever the depth allows
    p.q = (mode == PAPOF_SOR_REDBLACK && units % 2 == 0) ? 2 : 1;  // half-sweeps per planning unit
    const int n_units = units / p.q;
    const int depth = one_block ? n_units : std::max(1, sor_blocked_depth(h, mode, H, W) / p.q);
    p.n_launch = (n_units + depth - 1) / depth;
    p.base = n_units / p.n_launch;
    p.rem = n_units % p.n_launch;
    return p;
}

static int sor_blocked_solve(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int mode,
                             int units) {
    if (!sp.du2 || !sp.dv2) return PAPOF_EINVAL;
    const BlockedPlan bp = blocked_plan(h, mode, H, W, units);
    const Rect all{0, 0, W, H};
    const double *su = nullptr, *sv = nullptr;
    bool to_main = (bp.n_launch & 1) != 0;  // odd count: start in the main pair, so that the last launch ends there too
    int done = 0;
    for (int l = 0; l < bp.n_launch; l++) {
        const int g = bp.q * (bp.base + (l < bp.rem ? 1 : 0));
        double *du = to_main ? sp.du : sp.du2, *dv = to_main ? sp.dv : sp.dv2;
        PAPOF_TRY(sor_blocked_launch(h, sp, H, W, alpha, omega, mode, g, done, all, su, sv, du, dv));
        su = du;
        sv = dv;
        done += g;
        to_main = !to_main;
    }
    return PAPOF_OK;
}

// How sor_solve() runs a solve (measurement aid, papof_sor_plan): solver kernel launches per solve and, for the blocked
// solver, the largest number of half-sweeps (Jacobi: sweeps) one launch runs.
int sor_plan(const papof_handle* h, int H, int W, int n_sor, int mode, int* launches, int* depth) {
    if (!h || H < 1 || W < 1 || n_sor < 1) return PAPOF_EINVAL;
    int nl = 0, d = 0;
    if (mode == PAPOF_SOR_EXACT) {
        const int group = sor_group_size(h, H, W, n_sor);
        const SkewDims sd = skew_dims(H, W, n_sor, group, sor_fuse_size(h, H, W, n_sor, group));
        const bool aff = h->sor_xcd_affine && (sd.nb <= 8 || (h->sor_xcd_affine > 1 && sd.group == 1));
        const int per = aff ? 8 * ((sd.nb + 7) / 8) : sd.nb;
        const int units = sd.group > 1 ? (n_sor + sd.group - 1) / sd.group : (sd.fuse == 2 ? (n_sor + 1) / 2 : n_sor);
        const int chunk = std::max(1, resident_tasks(h) / (per * std::max(1, sd.group)));
        nl = (units + chunk - 1) / chunk;
        d = sd.fuse == 2 ? 2 : sd.group;
    } else if (mode == PAPOF_SOR_REDBLACK || mode == PAPOF_SOR_JACOBI) {
        if (h->rb_naive) {
            nl = mode == PAPOF_SOR_REDBLACK ? 2 * n_sor : n_sor;
            d = 1;
        } else {
            const BlockedPlan bp = blocked_plan(h, mode, H, W, mode == PAPOF_SOR_REDBLACK ? 2 * n_sor : n_sor);
            nl = bp.n_launch;
            d = bp.q * (bp.base + (bp.rem ? 1 : 0));
        }
    } else {
        return PAPOF_EINVAL;
    }
    if (launches) *launches = nl;
    if (depth) *depth = d;
    return PAPOF_OK;
}

int sor_solve(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int n_sor, int mode,
              const SorBatch* bt) {
    const double nalpha = -alpha, om1 = 1 - omega;
    if (n_sor <= 0) return PAPOF_EINVAL;
    const int batch = bt ? bt->n : 1;
    if (batch < 1 || (bt && mode != PAPOF_SOR_EXACT)) return PAPOF_EINVAL;
    const auto mark = [&](int on) {
        if (h->sor_mark) h->sor_mark(h->sor_mark_ctx, on);
    };
    const auto log_solve = [&](int kind, int depth, int launches) {
        if (h->sor_log.size() >= 4096) h->sor_log.clear();  // (stage / micro-benchmark entry points never reset it)
        h->sor_log.push_back(papof_handle::SorSolveLog{H, W, n_sor, kind, depth, launches, 0.0});
    };
    if (mode == PAPOF_SOR_EXACT && !sp.skew) {  // row-major operands: the plane is solved inside one workgroup (k_sor_tiny)
        mark(1);
        PAPOF_TRY(sor_tiny_solve(h, sp, H, W, alpha, omega, n_sor, batch, bt ? bt->tiny : 0));
        mark(0);
        log_solve(6, 0, 1);
        return PAPOF_OK;
    }
    if (mode == PAPOF_SOR_EXACT) {
        const SkewDims sd = skew_dims(H, W, n_sor, sp.sd.group, sp.sd.fuse);
        if (sd.fuse != sp.sd.fuse || sd.hp != sp.sd.hp || sd.npos != sp.sd.npos || sd.qt != sp.sd.qt || sd.nb != sp.sd.nb ||
            sd.n_sor != sp.sd.n_sor || sd.n > sp.cap_cells || sd.nd + sd.nh > sp.cap_cells_d)
            return PAPOF_EINVAL;  // sor_bind() must have chosen this layout (the operands were assembled in it)
        if ((sd.n + kLanes) * 16 >= (size_t(1) << 30) || (sd.nd + sd.nh) * 16 >= (size_t(1) << 30))
            return PAPOF_EINVAL;  // 32-bit byte offsets, see kOob
        if (bt && (sd.fuse != 1 || sd.group != 1 || !h->sor_prog_next)) return PAPOF_EINVAL;  // batches: the plain kernel, counters cleared ahead
        const size_t words = (size_t)sd.nb * n_sor * kProgStride + kProgStride;
        if (!bt && words > h->sync_cap) {
            PAPOF_HIP(hipStreamSynchronize(h->stream));
            if (h->sync_words) PAPOF_HIP(hipFree(h->sync_words));
            h->sync_words = nullptr;
            h->sync_cap = 0;
            const size_t cap = (words + 1023) & ~size_t(1023);
            PAPOF_HIP(hipMalloc((void**)&h->sync_words, cap * sizeof(unsigned)));
            h->sync_cap = cap;
            PAPOF_HIP(hipMemsetAsync(h->sync_words, 0, cap * sizeof(unsigned), h->stream));
        }
        // first line: abort word (kept across solves; checked by sor_check); counters start at the second line.  The
        // orchestrator may hand over counters it has cleared already (flow_device clears those of every solve of a call
        // on the preparation stream: sor_counters_*), else this solve's are cleared here
        unsigned* prog = h->sync_words + kProgStride;
        if (h->sor_prog_next) {
            prog = h->sor_prog_next;
            h->sor_prog_next = nullptr;
        } else {
            const size_t nbytes = (size_t)sd.nb * n_sor * kProgStride * sizeof(unsigned);
            PAPOF_HIP(hipMemsetAsync(prog, 0, nbytes, h->stream));
        }
        ExactArgs A;
        A.phi = sp.phi;
        A.xy = sp.xy;  // paired planes: only phi / a1 / b1 / du are used as the four plane bases
        A.a1 = sp.a1;
        A.a2 = sp.a2;
        A.b1 = sp.b1;
        A.b2 = sp.b2;
        A.du = sp.du;
        A.dv = sp.dv;
        A.prog = prog;
        A.abort = h->sync_words;
        A.H = H;
        A.W = W;
        A.nb = sd.nb;
        A.ns = sd.ns;
        A.hp = sd.hp;
        A.npos = sd.npos;
        A.qt = sd.qt;
        A.rt = sd.rt;
        A.npos_d = sd.npos_d;
        A.n_sor = n_sor;
        A.nalpha = nalpha;
        A.om1 = om1;
        A.k0 = 0;
        A.b0 = 0;
        A.nbl = sd.nb;
        A.peer_du = nullptr;
        A.peer_prog = nullptr;
        A.top_cut = A.bot_cut = 0;
        A.skip_dead = h->sor_skip_dead;
        A.bs_coef = bt ? bt->coef : 0;
        A.bs_d = bt ? bt->d : 0;
        A.bs_prog = bt ? bt->prog : 0;
        A.dbg = bt ? nullptr : h->sor_dbg;
        // Fault injection for the tests (tests/test_gpu_parity.py): raise the abort word before the launch, as a task whose
        // bounded wait expired would -- every task must then END (s_endpgm on the fast path, the polling loops' abort
        // check elsewhere) and the call must report PAPOF_ETIMEOUT instead of hanging or returning numbers.
        if (std::getenv("PAPOF_SOR_INJECT_ABORT")) PAPOF_HIP(hipMemsetAsync(h->sync_words, 1, sizeof(unsigned), h->stream));
        // The (du, dv) planes.  The grouped kernel reads du = dv = 0 before the first sweep (src/OpticalFlow.cpp:452-453) and
        // its halo rows from memory.  k_sor_exact / k_sor_fused read those zeros as out-of-range offsets and write every
        // position 1 .. n_iter * R of a plane before anybody reads it, so all they NEED cleared are the tail positions
        // that ghost lanes read up to 65 steps ahead and nobody writes (PAPOF_SOR_CLEAR=tail: one strided memset).  Yet
        // clearing both planes entirely is what is done by default, because it is FASTER, its own cost included: the
        // memset leaves the planes resident in the Infinity Cache, and every level's solve then runs 4-5 % faster
        // (1920x1080: 0.941 -> 0.900 ms, 607x341: 0.354 -> 0.338 ms, a whole 1080p pair 11.98 -> 11.56 ms).
        static const char* const clear_env = std::getenv("PAPOF_SOR_CLEAR");
        static const bool clear_tail_only = clear_env && std::strcmp(clear_env, "tail") == 0;
        if (bt) {  // every pair's two planes in one fill node (the pairs' planes lie bt->d doubles apart)
            PAPOF_TRY(fill_pairs(h, sp.du, bt->d * sizeof(double), sd.nd * 16, batch));
        } else if (sd.group > 1 || !clear_tail_only) {
            PAPOF_HIP(hipMemsetAsync(sp.du, 0, (sd.nd + sd.nh) * 16, h->stream));  // both planes (+ the halo rows)
        } else {
            const size_t block = (size_t)sd.nb * kLanes * 16, par = (size_t)sd.npos_d * block;
            PAPOF_HIP(hipMemset2DAsync((char*)sp.du + (size_t)sd.ns * block, par, 0, (size_t)(sd.npos_d - sd.ns) * block, 2,
                                       h->stream));
        }
        mark(1);  // everything below is solver kernels
        if (sd.group > 1) {
            GroupArgs Ga;
            Ga.phi = sp.phi;
            Ga.a1 = sp.a1;
            Ga.b1 = sp.b1;
            Ga.du = sp.du;
            Ga.prog = prog;
            Ga.abort = h->sync_words;
            Ga.nb = sd.nb;
            Ga.ns = sd.ns;
            Ga.hp = sd.hp;
            Ga.npos = sd.npos;
            Ga.qt = sd.qt;
            Ga.rt = sd.rt;
            Ga.npos_d = sd.npos_d;
            Ga.n_sor = n_sor;
            Ga.halo_off = (unsigned)(sd.nd * 16);
            Ga.dbg = h->sor_dbg;
            Ga.nalpha = nalpha;
            Ga.om1 = om1;
            Ga.g0 = 0;
            const int groups = (n_sor + sd.group - 1) / sd.group;
            Ga.xcd_affine = (h->sor_xcd_affine && sd.nb <= 8) ? 1 : 0;
            const int per_g = Ga.xcd_affine ? 8 * ((sd.nb + 7) / 8) : sd.nb;  // workgroups per group of sweeps
            const int Rg = h->sor_depth > 0 ? h->sor_depth : (sd.nb >= 9 ? 10 : 8);
            const int chunk = std::max(1, resident_tasks(h) / (per_g * sd.group));
            for (int g0 = 0; g0 < groups; g0 += chunk) {
                Ga.g0 = g0;
                h->sor_launches++;
                Ga.stamp = nullptr;  // no stamp at the head of the critical task (flow_internal.h: PhaseClock)
                const dim3 ggrid(per_g * std::min(chunk, groups - g0));
                if (sd.group == 4 && Rg >= 12)
                    hipLaunchKernelGGL((k_sor_group<12, 4, true>), ggrid, dim3(kLanes * 4), 0, h->stream, Ga);
                else if (sd.group == 4 && Rg >= 10)
                    hipLaunchKernelGGL((k_sor_group<10, 4, true>), ggrid, dim3(kLanes * 4), 0, h->stream, Ga);
                else if (sd.group == 4 && Rg >= 8)
                    hipLaunchKernelGGL((k_sor_group<8, 4, true>), ggrid, dim3(kLanes * 4), 0, h->stream, Ga);
                else if (sd.group == 4)
                    hipLaunchKernelGGL((k_sor_group<6, 4, true>), ggrid, dim3(kLanes * 4), 0, h->stream, Ga);
                else if (sd.group == 2)
                    hipLaunchKernelGGL((k_sor_group<8, 2, true>), ggrid, dim3(kLanes * 2), 0, h->stream, Ga);
                else
                    return PAPOF_EINVAL;
            }
            PAPOF_HIP(hipGetLastError());
            mark(0);
            log_solve(2, Rg, (groups + chunk - 1) / chunk);
            return PAPOF_OK;
        }
        if (sd.fuse == 2) {  // two sweeps per wave (k_sor_fused); effective pipeline depth R - 2
            const int pairs = (n_sor + 1) / 2;
            A.xcd_affine = (h->sor_xcd_affine && (sd.nb <= 8 || h->sor_xcd_affine > 1)) ? 1 : 0;
            const int per_q = A.xcd_affine ? 8 * ((sd.nb + 7) / 8) : sd.nb;  // workgroups per pair of sweeps
            // depth 8: isolated solves are equal at 6 / 8 / 10 (profiles/r01_s3_sor_sweeps.txt), but INSIDE a pair -- other
            // cache contents, the preparation stream beside it -- 8 is 0.18 ms per 1080p pair faster than 6 and 0.07 faster
            // than 10 (same-box A/B, profiles/r02_ab_variants.txt)
            const int Rf = h->sor_depth > 0 ? h->sor_depth : 8;
            const int chunk = std::max(1, resident_tasks(h) / per_q);
            for (int q0 = 0; q0 < pairs; q0 += chunk) {
                A.k0 = q0;
                h->sor_launches++;
                A.stamp = nullptr;
                const dim3 fgrid(per_q * std::min(chunk, pairs - q0));
                if (Rf <= 6)
                    hipLaunchKernelGGL((k_sor_fused<6, true>), fgrid, dim3(kLanes), 0, h->stream, A);
                else if (Rf <= 8)
                    hipLaunchKernelGGL((k_sor_fused<8, true>), fgrid, dim3(kLanes), 0, h->stream, A);
                else if (Rf <= 10)
                    hipLaunchKernelGGL((k_sor_fused<10, true>), fgrid, dim3(kLanes), 0, h->stream, A);
                else
                    hipLaunchKernelGGL((k_sor_fused<12, true>), fgrid, dim3(kLanes), 0, h->stream, A);
            }
            PAPOF_HIP(hipGetLastError());
            mark(0);
            log_solve(1, Rf, (pairs + chunk - 1) / chunk);
            return PAPOF_OK;
        }
        // measured: with at most one band per XCD the affinity is worth 3-4 % (small pyramid levels); beyond that the
        // uneven band count per XCD costs more than the L2 hits give (1920x1080: 18 bands over 8 XCDs, -5 %)
        A.xcd_affine = (h->sor_xcd_affine && (sd.nb <= 8 || h->sor_xcd_affine > 1)) ? 1 : 0;
        // Batched launches: NO affinity.  It would put band b of EVERY pair on XCD b % 8 -- the three bands of sixteen 240x135
        // pairs on XCDs 0..2 -- and pad a sweep's row of tasks to 8 blocks, which cuts a solve into more launches: measured
        // 1.49 vs 0.61 ms of solver time per pair in a batch of 16 (profiles/r04_batch_knobs.txt; rotating the bands per pair so
        // that the XCDs are loaded equally: 0.82).
        if (bt) A.xcd_affine = 0;
        const int per_k = A.xcd_affine ? 8 * ((sd.nb + 7) / 8) : sd.nb;  // workgroups per sweep
        const dim3 block(kLanes);
        // pipeline depth: every load is issued R steps ahead and a task looks 2R steps ahead of its producers, so R is
        // also what a hand-off costs.  Measured (profiles/r01_s2_sor_depth_sweep.txt): the big levels want the deeper
        // pipeline (throughput; with the mid-iteration poll 1440x810: 0.85 -> 0.81 ms, 1080x607: 0.68 -> 0.65 ms), the
        // small, hand-off-bound levels the shorter one (607x341: 0.53 -> 0.47 ms).
        const int R = h->sor_depth > 0 ? h->sor_depth : (sd.nb >= 8 ? 8 : 6);  // (in-pair A/B of round 2: 6 / 10 on the big
                                                                                // levels, 4 / 8 on the small ones: all worse)
        const int chunk = std::max(1, resident_tasks(h) / (per_k * batch));  // (every task of a launch resident: all pairs' count)
        for (int k0 = 0; k0 < n_sor; k0 += chunk) {
            A.k0 = k0;
            h->sor_launches++;
            A.stamp = nullptr;
            const dim3 grid(per_k * std::min(chunk, n_sor - k0), batch);
            if (!h->use_dpp)
                hipLaunchKernelGGL((k_sor_exact<8, false>), grid, block, 0, h->stream, A);
            else if (R <= 4)
                hipLaunchKernelGGL((k_sor_exact<4, true>), grid, block, 0, h->stream, A);
            else if (R <= 6)
                hipLaunchKernelGGL((k_sor_exact<6, true>), grid, block, 0, h->stream, A);
            else if (R <= 8)
                hipLaunchKernelGGL((k_sor_exact<8, true>), grid, block, 0, h->stream, A);
            else if (R <= 10)
                hipLaunchKernelGGL((k_sor_exact<10, true>), grid, block, 0, h->stream, A);
            else
                hipLaunchKernelGGL((k_sor_exact<12, true>), grid, block, 0, h->stream, A);
        }
        PAPOF_HIP(hipGetLastError());
        mark(0);
        log_solve(0, R, (n_sor + chunk - 1) / chunk);
        return PAPOF_OK;
    }
    if (sp.skew) return PAPOF_EINVAL;
    if (!h->rb_naive) {  // the LDS-tiled, temporally blocked kernel: no plane clears (the first launch reads no unknowns)
        mark(1);
        PAPOF_TRY(sor_blocked_solve(h, sp, H, W, alpha, omega, mode, mode == PAPOF_SOR_REDBLACK ? 2 * n_sor : n_sor));
        mark(0);
        {
            const BlockedPlan bp = blocked_plan(h, mode, H, W, mode == PAPOF_SOR_REDBLACK ? 2 * n_sor : n_sor);
            log_solve(mode == PAPOF_SOR_REDBLACK ? 3 : 4, bp.q * (bp.base + (bp.rem ? 1 : 0)), bp.n_launch);
        }
        return PAPOF_OK;
    }
    // PAPOF_RB_NAIVE=1: one launch per half-sweep straight on the planes (the first implementation; kept as a cross-check)
    const size_t np = (size_t)H * W;
    PAPOF_HIP(hipMemsetAsync(sp.du, 0, np * sizeof(double), h->stream));  // src/OpticalFlow.cpp:452-453
    PAPOF_HIP(hipMemsetAsync(sp.dv, 0, np * sizeof(double), h->stream));
    mark(1);
    if (mode == PAPOF_SOR_REDBLACK) {
        for (int k = 0; k < n_sor; k++)
            for (int colour = 0; colour < 2; colour++)
                PAPOF_TRY(sor_redblack_halfsweep(h, sp, H, W, alpha, omega, colour, Rect{0, 0, W, H}));
        mark(0);
        log_solve(5, 1, 2 * n_sor);
        return PAPOF_OK;
    }
    if (mode == PAPOF_SOR_JACOBI) {
        if (!sp.du2 || !sp.dv2) return PAPOF_EINVAL;
        const dim3 grid((W + 63) / 64, (H + 3) / 4), block(64, 4);
        double *ru = sp.du, *rv = sp.dv, *wu = sp.du2, *wv = sp.dv2;
        for (int k = 0; k < n_sor; k++) {
            hipLaunchKernelGGL(k_sor_jacobi, grid, block, 0, h->stream, sp.phi, sp.xy, sp.a1, sp.a2, sp.b1, sp.b2, ru,
                               rv, wu, wv, H, W, nalpha, om1);
            double* t = ru;
            ru = wu;
            wu = t;
            t = rv;
            rv = wv;
            wv = t;
        }
        PAPOF_HIP(hipGetLastError());
        if (ru != sp.du) {  // odd sweep count: latest values are in the ping-pong buffers
            PAPOF_HIP(hipMemcpyAsync(sp.du, ru, np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            PAPOF_HIP(hipMemcpyAsync(sp.dv, rv, np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
        mark(0);
        log_solve(5, 1, n_sor);
        return PAPOF_OK;
    }
    return PAPOF_EINVAL;
}

// ---- one solve as STRIPS of bands on several streams (api.hip: smooth_flow_strips) -------------------------
// The bands b0 .. b1-1 of a solve are one launch; the launches of a solve run on different streams and meet through the
// progress counters exactly as the tasks of one launch do (a task waits on (b, k-1), (b-1, k) and -- write-after-read --
// (b+1, k-2): the first two point into the same or the upper strip, the third into the same or the LOWER one, so an upper
// strip can run at most two sweeps per band ahead of the strip below it; every wait is bounded).  Because a strip starts
// before the strips below it have even been launched, the counters of a solve must be zero before its FIRST strip
// starts: every solve of a level has its own counter array, all cleared at once by sor_strips_begin() before the
// streams fork.  The (du, dv) blocks of the strip's bands are cleared by the strip itself (cache warming, sor_solve).
// Zero n16 16-byte cells of every pair of a batch (the pairs' blocks lie stride16 cells apart): blockIdx.y = pair.  The runtime's
// 2-D fill reaches 0.75 TB/s on these shapes (43 MB in 58 us: profiles/r04_batch16_kernel_avgs_by_grid_xcd_affine.txt); this one
// is a plain streaming store.
__global__ __launch_bounds__(256) void k_fill_pairs(uint4* __restrict__ d, size_t stride16, size_t n16) {
    uint4* const p = d + (size_t)blockIdx.y * stride16;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < n16; c += (size_t)gridDim.x * 256) p[c] = uint4{0u, 0u, 0u, 0u};
}
static int fill_pairs(papof_handle* h, void* base, size_t stride_bytes, size_t bytes, int batch) {
    if ((stride_bytes | bytes | (size_t)(uintptr_t)base) & 15) return PAPOF_EINVAL;
    const size_t n16 = bytes / 16;
    const unsigned gx = (unsigned)std::min<size_t>((n16 + 1023) / 1024, 2048);  // four cells per thread and more on big planes
    hipLaunchKernelGGL(k_fill_pairs, dim3(std::max(1u, gx), batch), dim3(256), 0, h->stream, (uint4*)base, stride_bytes / 16, n16);
    PAPOF_HIP(hipGetLastError());
    return PAPOF_OK;
}

__global__ void k_sor_clear_bands(uint4* __restrict__ d, int nb, int b0, int nbl, unsigned n16) {
    // the 16-byte cells [position][band b0 .. b0+nbl-1][64] of both parities; n16 = positions * nbl * 64
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n16) return;
    const unsigned per = (unsigned)nbl * kLanes, pos = t / per, r = t - pos * per;
    d[((size_t)pos * nb + b0) * kLanes + r] = uint4{0u, 0u, 0u, 0u};
}

bool sor_strips_supported(const papof_handle* h, const SorPlanes& sp, int n_sor) {
    if (!sp.skew || sp.sd.group > 1 || !h->use_dpp || n_sor < 3) return false;
    if (h->sor_xcd_affine && (sp.sd.nb <= 8 || h->sor_xcd_affine > 1)) return false;  // the XCD-affine task mapping
    const int per = sp.sd.nb, tasks = per * (sp.sd.fuse == 2 ? (n_sor + 1) / 2 : n_sor);
    return tasks <= resident_tasks(h);  // one launch per strip: every task of the solve resident
}

// Counters of one solve of a height x width plane (whichever layout sor_bind() picks), in unsigneds
size_t sor_counters_words(int H, int W, int n_sor) {
    const int nb = std::max(skew_dims(H, W, n_sor, 1, 1).nb, skew_dims(H, W, n_sor, 1, 2).nb);
    return (size_t)nb * n_sor * kProgStride;
}

// Room for `words` unsigneds of counters behind the abort line (may reallocate: call it before anything is enqueued)
int sor_counters_ensure(papof_handle* h, size_t words) {
    const size_t need = words + kProgStride;
    if (need <= h->sync_cap) return PAPOF_OK;
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    if (h->sync_words) PAPOF_HIP(hipFree(h->sync_words));
    h->sync_words = nullptr;
    h->sync_cap = 0;
    const size_t cap = (need + 1023) & ~size_t(1023);
    PAPOF_HIP(hipMalloc((void**)&h->sync_words, cap * sizeof(unsigned)));
    h->sync_cap = cap;
    PAPOF_HIP(hipMemsetAsync(h->sync_words, 0, cap * sizeof(unsigned), h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    return PAPOF_OK;
}

// Clear `words` unsigneds of counters (on h->stream); returns their base: solve i of a level uses base + i * per_solve
unsigned* sor_counters_clear(papof_handle* h, size_t offset_words, size_t words) {
    if (offset_words + words + kProgStride > h->sync_cap) return nullptr;
    unsigned* base = h->sync_words + kProgStride + offset_words;
    if (hipMemsetAsync(base, 0, words * sizeof(unsigned), h->stream) != hipSuccess) return nullptr;
    return base;
}

int sor_strips_begin(papof_handle* h, const SorPlanes& sp, int n_sor, int n_solves) {  // the test aid's (api.hip)
    const SkewDims& sd = sp.sd;
    const size_t per = (size_t)sd.nb * n_sor * kProgStride;
    PAPOF_TRY(sor_counters_ensure(h, per * n_solves));
    if (!sor_counters_clear(h, 0, per * n_solves)) return PAPOF_EDEVICE;
    return PAPOF_OK;
}

int sor_solve_bands(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int n_sor,
                    unsigned* prog, int b0, int b1, const SorSplit* split, int k0, int k1) {
    // k0 .. k1-1: the sweeps of THIS launch (split solves only: tiles.hip issues a rank's solve as launches over ranges of
    // sweeps so that the ranks pipeline; the ping-pong planes and the counters carry the state from launch to launch)
    if (k1 < 0) k1 = n_sor;
    if (!sp.skew || n_sor <= 0 || k0 < 0 || k1 > n_sor || k1 <= k0 || ((k0 != 0 || k1 != n_sor) && !split)) return PAPOF_EINVAL;
    // split over handles (tiles.hip: bands_flow): the plain kernel only, one inbox cell per sweep (ExactArgs)
    if (split && (sp.sd.fuse != 1 || sp.sd.group != 1 || n_sor > 128 || !h->use_dpp)) return PAPOF_EINVAL;
    const SkewDims sd = skew_dims(H, W, n_sor, sp.sd.group, sp.sd.fuse);
    if (sd.fuse != sp.sd.fuse || sd.hp != sp.sd.hp || sd.npos != sp.sd.npos || sd.qt != sp.sd.qt || sd.nb != sp.sd.nb ||
        sd.n_sor != sp.sd.n_sor || sd.n > sp.cap_cells || sd.nd + sd.nh > sp.cap_cells_d || sd.group > 1)
        return PAPOF_EINVAL;
    if ((sd.n + kLanes) * 16 >= (size_t(1) << 30) || (sd.nd + sd.nh) * 16 >= (size_t(1) << 30)) return PAPOF_EINVAL;
    if (b0 < 0 || b1 > sd.nb || b1 <= b0 || !prog) return PAPOF_EINVAL;
    const size_t per = (size_t)sd.nb * n_sor * kProgStride;
    if (prog < h->sync_words + kProgStride || prog + per > h->sync_words + h->sync_cap) return PAPOF_EINVAL;
    const int nbl = b1 - b0;
    if (std::getenv("PAPOF_SOR_INJECT_ABORT"))  // fault injection for the tests, as in sor_solve
        PAPOF_HIP(hipMemsetAsync(h->sync_words, 1, sizeof(unsigned), h->stream));
    if (k0 == 0) {  // (once per solve: later launches of the solve read what the earlier ones wrote)
        const unsigned n16 = (unsigned)(2 * sd.npos_d) * (unsigned)nbl * kLanes;
        hipLaunchKernelGGL(k_sor_clear_bands, dim3((n16 + 255) / 256), dim3(256), 0, h->stream, (uint4*)sp.du, sd.nb, b0,
                           nbl, n16);
    }
    ExactArgs A;
    A.phi = sp.phi;
    A.xy = sp.xy;
    A.a1 = sp.a1;
    A.a2 = sp.a2;
    A.b1 = sp.b1;
    A.b2 = sp.b2;
    A.du = sp.du;
    A.dv = sp.dv;
    A.prog = prog;
    A.abort = h->sync_words;
    A.H = H;
    A.W = W;
    A.nb = sd.nb;
    A.ns = sd.ns;
    A.hp = sd.hp;
    A.npos = sd.npos;
    A.qt = sd.qt;
    A.rt = sd.rt;
    A.npos_d = sd.npos_d;
    A.n_sor = n_sor;
    A.nalpha = -alpha;
    A.om1 = 1 - omega;
    A.k0 = k0;
    A.b0 = b0;
    A.nbl = nbl;
    A.xcd_affine = 0;
    A.dbg = nullptr;
    A.stamp = nullptr;
    A.bs_coef = A.bs_d = A.bs_prog = 0;
    A.skip_dead = h->sor_skip_dead;
    A.peer_du = split ? split->peer_du : nullptr;
    A.peer_prog = split ? split->peer_prog : nullptr;
    A.top_cut = split && split->top_cut ? 1 : 0;
    A.bot_cut = split && split->bot_cut && split->peer_du && split->peer_prog ? 1 : 0;
    if (split && split->bot_cut && !A.bot_cut) return PAPOF_EINVAL;
    if (h->sor_mark) h->sor_mark(h->sor_mark_ctx, 1);
    if (split) {
        const int R = h->sor_depth > 0 ? h->sor_depth : (sd.nb >= 8 ? 8 : 6);
        const dim3 grid(nbl * (k1 - k0));
        if (R <= 6)
            hipLaunchKernelGGL((k_sor_exact<6, true, true>), grid, dim3(kLanes), 0, h->stream, A);
        else
            hipLaunchKernelGGL((k_sor_exact<8, true, true>), grid, dim3(kLanes), 0, h->stream, A);
    } else if (sd.fuse == 2) {
        const int pairs = (n_sor + 1) / 2, Rf = h->sor_depth > 0 ? h->sor_depth : 8;
        const dim3 grid(nbl * pairs);
        if (Rf <= 6)
            hipLaunchKernelGGL((k_sor_fused<6, true>), grid, dim3(kLanes), 0, h->stream, A);
        else if (Rf <= 8)
            hipLaunchKernelGGL((k_sor_fused<8, true>), grid, dim3(kLanes), 0, h->stream, A);
        else if (Rf <= 10)
            hipLaunchKernelGGL((k_sor_fused<10, true>), grid, dim3(kLanes), 0, h->stream, A);
        else
            hipLaunchKernelGGL((k_sor_fused<12, true>), grid, dim3(kLanes), 0, h->stream, A);
    } else {
        const int R = h->sor_depth > 0 ? h->sor_depth : (sd.nb >= 8 ? 8 : 6);
        const dim3 grid(nbl * n_sor);
        if (R <= 4)
            hipLaunchKernelGGL((k_sor_exact<4, true>), grid, dim3(kLanes), 0, h->stream, A);
        else if (R <= 6)
            hipLaunchKernelGGL((k_sor_exact<6, true>), grid, dim3(kLanes), 0, h->stream, A);
        else if (R <= 8)
            hipLaunchKernelGGL((k_sor_exact<8, true>), grid, dim3(kLanes), 0, h->stream, A);
        else if (R <= 10)
            hipLaunchKernelGGL((k_sor_exact<10, true>), grid, dim3(kLanes), 0, h->stream, A);
        else
            hipLaunchKernelGGL((k_sor_exact<12, true>), grid, dim3(kLanes), 0, h->stream, A);
    }
    PAPOF_HIP(hipGetLastError());
    h->sor_launches++;
    if (h->sor_mark) h->sor_mark(h->sor_mark_ctx, 0);
    return PAPOF_OK;
}

// All skew positions that are not real cells must read as 0.0 (the kernel relies on it instead of predicates).
// Real cells are rewritten by the assembly kernel each outer iteration and padding is only ever written with
// zeros, so one memset per (level, plane) suffices.
// doubles one paired coefficient plane of the bound layout occupies (whole 1-KiB blocks): sor_bind() packs the three planes
// at that pitch, so that sor_reset_planes() is ONE fill node instead of three (ten launches fewer on a 5-level call's chain)
static size_t plane_pitch(const SkewDims& sd) { return 2 * (((size_t)sd.n + kLanes + 63) / 64 * 64); }
static void pack_planes(SorPlanes& sp) {
    const size_t pitch = plane_pitch(sp.sd);
    sp.xy = sp.phi + 1;
    sp.a1 = sp.phi + pitch;
    sp.a2 = sp.a1 + 1;
    sp.b1 = sp.a1 + pitch;
    sp.b2 = sp.b1 + 1;
}
int sor_reset_planes(papof_handle* h, const SorPlanes& sp) {
    if (!sp.skew) return PAPOF_OK;
    PAPOF_HIP(hipMemsetAsync(sp.phi, 0, 3 * plane_pitch(sp.sd) * sizeof(double), h->stream));
    return PAPOF_OK;  // the (du, dv) planes are cleared by every solve
}

// ... of every pair of a batch at once: the pairs' operand blocks lie `stride` doubles apart (api.hip / batch.hip: flow_batch)
int sor_reset_planes_batch(papof_handle* h, const SorPlanes& sp, int batch, size_t stride) {
    if (!sp.skew) return PAPOF_OK;
    return fill_pairs(h, sp.phi, stride * sizeof(double), 3 * plane_pitch(sp.sd) * sizeof(double), batch);
}

// Sweeps per workgroup of the exact-order solver for this problem: the grouped kernel needs the verified DPP lane
// shifts; PAPOF_SOR_GROUP = 1 selects the one-wave-per-workgroup kernel.
int sor_group_size(const papof_handle* h, int H, int W, int n_sor) {
    if (!h || !h->use_dpp || n_sor < 2) return 1;
    int g = h->sor_group;
    if (g == 0) {
        // measured: an isolated 240x135 solve gains 17 % (0.25 -> 0.21 ms), a 240x135 pair nothing.  Round 3, inside
        // 15-level calls (profiles/r03_mid_level_kernels.txt): four sweeps per workgroup through LDS win where a solve is
        // nothing but its chain of sweep hand-offs -- at most 2 bands from 36 sweeps on (135x75 x 36: 154 vs 167 us, 108x60 x 60:
        // 206 vs 250), 3 bands from 50 sweeps on (144x81 x 57: 237 vs 258, 192x107 x 54: 243 vs 254) -- and lose elsewhere
        const int nb = skew_dims(H, W, n_sor).nb;
        g = ((nb <= 2 && n_sor >= 36) || (nb == 3 && n_sor >= 50)) ? 4 : 1;
    }
    if (g >= 4) return 4;  // LDS (160 KB) holds the rings of up to 4 waves... and 4 waves = one per SIMD
    return g >= 2 ? 2 : 1;
}

// Sweeps per wave: the fused-pair kernel needs the verified DPP lane shifts and excludes the grouped kernel.  Measured
// (profiles/r01_s3_sor_sweeps.txt): its step costs 1.7x a plain step (the arithmetic of two sweeps; a wave cannot issue
// fp64 faster), but it moves half the bytes, so it wins where the solve is bandwidth-bound -- 1920x1080: 0.98 -> 0.91 ms
// -- and loses on the hand-off-bound levels below ~16 bands.
static int sor_fuse_size(const papof_handle* h, int H, int W, int n_sor, int group) {
    if (!h || !h->use_dpp || n_sor < 2 || group > 1) return 1;
    if (h->sor_fuse > 0) return h->sor_fuse == 2 ? 2 : 1;
    return skew_dims(H, W, n_sor).nb >= 16 ? 2 : 1;  // (inside a pair, round 2: from 14 / 10 bands on: 0.15 / 0.47 ms slower)
}

int sor_bind(papof_handle* h, SorPlanes& sp, int H, int W, int n_sor) {
    if (!sp.skew) return PAPOF_OK;
    const int group = sor_group_size(h, H, W, n_sor);
    const SkewDims sd = skew_dims(H, W, n_sor, group, sor_fuse_size(h, H, W, n_sor, group));
    if (sd.n > sp.cap_cells || sd.nd + sd.nh > sp.cap_cells_d) return PAPOF_ENOMEM;
    sp.sd = sd;
    pack_planes(sp);
    return PAPOF_OK;
}

int sor_bind_plain(papof_handle* h, SorPlanes& sp, int H, int W, int n_sor) {
    (void)h;
    if (!sp.skew) return PAPOF_EINVAL;
    const SkewDims sd = skew_dims(H, W, n_sor, 1, 1);
    if (sd.n > sp.cap_cells || sd.nd + sd.nh > sp.cap_cells_d) return PAPOF_ENOMEM;
    sp.sd = sd;
    pack_planes(sp);
    return PAPOF_OK;
}

int sor_alloc_planes(Arena& A, int H, int W, int mode, int n_sor_cap, SorPlanes& sp) {
    sp.skew = mode == PAPOF_SOR_EXACT;
    sp.du2 = sp.dv2 = nullptr;
    sp.cap_cells = sp.cap_cells_d = 0;
    sp.sd = SkewDims{};
    if (sp.skew) {
        sp.sd = skew_dims(H, W, n_sor_cap, 2);
        skew_capacity(H, W, n_sor_cap, sp.cap_cells, sp.cap_cells_d);  // whichever layout sor_bind() chooses later
        const size_t n = 2 * ((sp.cap_cells + kLanes + 63) / 64 * 64);  // doubles per paired plane, whole 1-KiB blocks
        sp.phi = A.f64(3 * n);  // the three paired planes, one behind the other at the pitch of the layout in use (pack_planes)
        if (sp.phi) pack_planes(sp);
        sp.du = A.f64(2 * (sp.cap_cells_d + kLanes));
        sp.dv = sp.du ? sp.du + 1 : nullptr;
    } else {
        const size_t n = (size_t)H * W;
        sp.phi = A.f64(n);
        sp.xy = A.f64(n);
        sp.a1 = A.f64(n);
        sp.a2 = A.f64(n);
        sp.b1 = A.f64(n);
        sp.b2 = A.f64(n);
        sp.du = A.f64(n);
        sp.dv = A.f64(n);
        sp.du2 = A.f64(n);  // second pair of unknown planes: the blocked solver / Jacobi alternate between the two
        sp.dv2 = A.f64(n);
    }
    return A.overflow ? PAPOF_ENOMEM : PAPOF_OK;
}

// Row-major operand planes for the tiny-plane exact-order solver (k_sor_tiny), `cells` doubles each
int sor_alloc_tiny_planes(Arena& A, size_t cells, SorPlanes& sp) {
    sp = SorPlanes{};
    sp.skew = false;
    sp.phi = A.f64(cells);
    sp.xy = A.f64(cells);
    sp.a1 = A.f64(cells);
    sp.a2 = A.f64(cells);
    sp.b1 = A.f64(cells);
    sp.b2 = A.f64(cells);
    sp.du = A.f64(cells);
    sp.dv = A.f64(cells);
    return A.overflow ? PAPOF_ENOMEM : PAPOF_OK;
}

int sor_probe_dpp(papof_handle* h) {
    h->use_dpp = false;
    int* d = nullptr;
    PAPOF_HIP(hipMalloc((void**)&d, 128 * sizeof(int)));
    PAPOF_HIP(hipMemsetAsync(d, 0xff, 128 * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_xlane_probe, dim3(1), dim3(kLanes), 0, h->stream, d);
    int host[128];
    hipError_t e = hipMemcpyAsync(host, d, sizeof host, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d);
    if (e != hipSuccess) {
        set_last_error("dpp probe", e, __FILE__, __LINE__);
        return PAPOF_EDEVICE;
    }
    bool ok = true;
    for (int l = 1; l < 64; l++) ok = ok && host[l] == l;             // lane l sees lane l-1 (value l-1+1)
    for (int l = 0; l < 63; l++) ok = ok && host[64 + l] == l + 2;    // lane l sees lane l+1 (value l+1+1)
    h->use_dpp = ok;
    if (std::getenv("PAPOF_PROBE")) {
        unsigned long long* dt = nullptr;
        double* sink = nullptr;
        if (hipMalloc((void**)&dt, 16) == hipSuccess && hipMalloc((void**)&sink, 1024 * 8) == hipSuccess) {
            for (int rep = 0; rep < 5; rep++) {  // 1, 2, 4, 8, 16 waves of ONE workgroup = one CU: does its fp64 issue scale?
                const int n = 200000;
                const int lanes = 64 << rep;
                std::fprintf(stderr, "[papof probe] waves on one CU %d: ", lanes / 64);
                hipLaunchKernelGGL(k_alu_probe, dim3(1), dim3(lanes), 0, h->stream, dt, sink, n, 1.0 + rep);
                unsigned long long t[2] = {0, 0};
                hipMemcpyAsync(t, dt, 16, hipMemcpyDeviceToHost, h->stream);
                hipStreamSynchronize(h->stream);
                std::fprintf(stderr, "[papof probe] dpp=%d  %d steps: %.1f shader cycles/step, %.4f us/step, clock %.0f MHz\n",
                             (int)ok, n, (double)t[0] / n, (double)t[1] * 0.01 / n, (double)t[0] / ((double)t[1] * 0.01));
            }
        }
        hipFree(dt);
        hipFree(sink);
    }
    if (const char* s = std::getenv("PAPOF_SOR_XLANE")) {
        if (std::strcmp(s, "shfl") == 0) h->use_dpp = false;
    }
    return PAPOF_OK;
}

int sor_check(papof_handle* h) {
    if (!h->sync_words) return PAPOF_OK;
    unsigned flag = 0;
    // on the handle's own stream: the legacy stream would synchronise with every other stream of the process, and is
    // off limits while another thread captures a graph
    PAPOF_HIP(hipMemcpyAsync(&flag, h->sync_words, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    PAPOF_HIP(hipStreamSynchronize(h->stream));
    if (flag != 0) {
        PAPOF_HIP(hipMemsetAsync(h->sync_words, 0, sizeof(unsigned), h->stream));
        PAPOF_HIP(hipStreamSynchronize(h->stream));
        return PAPOF_ETIMEOUT;
    }
    return PAPOF_OK;
}

}  // namespace papof
