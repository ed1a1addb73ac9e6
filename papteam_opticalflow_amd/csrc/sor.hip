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
//     * BAND  = 64 consecutive rows; TASK = (band b, sweep k) = one wavefront, lane r <-> row 64b + r;
//     * at STEP s lane r updates column j = s - r (the lane above runs one column ahead), NS = W + 63 steps;
//     * operands live in the per-band SKEWED layout ((b*NS + j + r)*64 + r): at step s the 64 lanes touch
//       64 consecutive doubles = one fully coalesced 512-byte access per operand;
//     * left-new is the lane's own previous result, up-new the previous result of lane r-1, down-old the
//       pending centre of lane r+1 (cross-lane moves); right-old is loaded and becomes the next centre;
//       lane 0 / lane 63 fetch their up / down neighbours from the adjacent bands' planes;
//     * all nb * n_sor tasks are launched at once (one 64-thread workgroup each; at most a few hundred
//       waves, far below the chip's 8192 wave slots, so all are co-resident) and pipeline through
//       per-task progress counters:  before steps [s0, s1) task (b, k) waits for
//           prog[k-1][b]   >= min(NS, s1 + 1)      own band, previous sweep (centre / right-old)
//           prog[k][b-1]   >= min(NS, s1 + 63)     band above, this sweep   (row 63 = up-new of lane 0)
//           prog[k-1][b+1] >= min(NS, s1 - 63)     band below, previous sweep (row 0 = down-old of lane 63)
//         and publishes prog[k][b] = s1 afterwards.  A waiter only ever waits on lower block indices.
//       The model in tests/sim_sor_wave.py executes this exact dataflow under a random scheduler and is
//       checked bit-for-bit against the oracle on the CPU.
//     * cross-workgroup visibility follows MI355X guide G16/R1: du/dv are stored write-through
//       (agent-scope relaxed atomics => `sc1`), every publishing wave drains `s_waitcnt vmcnt(0)` before
//       its one-lane relaxed agent-scope counter store; consumers poll the counter relaxed, then read
//       du/dv only with agent-scope (`sc1`, L1-bypassing) loads.  Counters are zeroed by a memset node
//       before every launch, every spin is bounded, and a timeout raises an abort word (PAPOF_ETIMEOUT).
//
// PAPOF_SOR_REDBLACK / PAPOF_SOR_JACOBI -- one launch per half-sweep / sweep on row-major planes;
//   throughput and correctness-gate modes whose results differ from the reference's order.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace papof {

namespace {

constexpr unsigned kSpinLimit = 4u << 20;  // bounded wait: ~4M polls (seconds), then abort

struct ExactArgs {
    const double *phi, *xy, *a1, *a2, *b1, *b2;
    double *du, *dv;
    unsigned* prog;   // [n_sor][nb]
    unsigned* abort;  // one word
    int H, W, nb, ns, nsp, n_sor, chunk;
    double nalpha, om1;
};

__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave-uniform bounded wait for *p >= need.  Returns false on abort/timeout.
__device__ __forceinline__ bool wait_ge(unsigned* p, unsigned need, unsigned* abort_word) {
    unsigned spins = 0;
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(2);
        ++spins;
        if ((spins & 255u) == 0u) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > kSpinLimit) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

// Cross-lane move of one fp64 value by one lane, with the wave-edge lane taking `edge` instead:
//   from_above(edge, x): lane r <- x of lane r-1 ; lane 0  <- its own `edge`
//   from_below(edge, x): lane r <- x of lane r+1 ; lane 63 <- its own `edge`
// DPP wave_shr:1 / wave_shl:1 (GFX9 full-wavefront shifts, dpp_ctrl 0x138 / 0x130; a lane without a source keeps
// the `old` operand when bound_ctrl = 0) when the start-up probe verified that behaviour on this device,
// else ds_bpermute + select.
template <bool DPP>
__device__ __forceinline__ double from_above(double edge, double x, bool is0) {
    if (DPP) {
        int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(x), 0x138, 0xf, 0xf, false);
        int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(x), 0x138, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    const double y = __shfl_up(x, 1);
    return is0 ? edge : y;
}
template <bool DPP>
__device__ __forceinline__ double from_below(double edge, double x, bool is63) {
    if (DPP) {
        int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(x), 0x130, 0xf, 0xf, false);
        int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(x), 0x130, 0xf, 0xf, false);
        return __hiloint2double(hi, lo);
    }
    const double y = __shfl_down(x, 1);
    return is63 ? edge : y;
}

__global__ void k_xlane_probe(int* out) {  // out[0..63] = from_above, out[64..127] = from_below (DPP forms)
    const int lane = threadIdx.x;
    const double v = (double)(lane + 1), edge = (double)(1000 + lane);
    out[lane] = (int)from_above<true>(edge, v, lane == 0);
    out[64 + lane] = (int)from_below<true>(edge, v, lane == 63);
}

constexpr int U = kSorGroup;

// Operands of U consecutive steps, prefetched into registers one group ahead of their use.
struct Group {
    double phi[U], xy[U], a1[U], a2[U], b1[U], b2[U];  // this lane's cell (r, s - r)
    double duR[U], dvR[U];                             // right-old = skew position s + 1, previous sweep
    double hu[U], hv[U], hp[U];                        // halo: lane 0 <- row above (du, dv, phi); lane 63 <- row below
};

struct Lane {  // per-task constants
    unsigned lane;       // 0..63
    unsigned band;       // element offset of (this band, position 0, lane 0) in a plane -- wave-uniform
    unsigned up_src;     // element offset of (band above, position 63, lane 63)         -- wave-uniform
    unsigned dn_src;     // element offset of (band below, position -63, lane 0)         -- wave-uniform (may wrap)
    unsigned zero_src;   // element offset of a padding cell of this band (always 0.0)    -- wave-uniform
    bool is0, is63, prev, up_any, dn_any;
    int W;
};

// All addresses are (uniform plane pointer advanced by a uniform element offset) + lane: scalar address
// arithmetic and one constant VGPR offset, so the vector ALU is left to the fp64 work.
__device__ __forceinline__ void load_group(const ExactArgs& A, const Lane& L, int sg, Group& g) {
    const unsigned base = L.band + (unsigned)sg * kLanes;  // wave-uniform
    const double* const pphi = A.phi + base;
    const double* const pxy = A.xy + base;
    const double* const pa1 = A.a1 + base;
    const double* const pa2 = A.a2 + base;
    const double* const pb1 = A.b1 + base;
    const double* const pb2 = A.b2 + base;
    const double* const pdu = A.du + base + kLanes;  // right-old: next skew position
    const double* const pdv = A.dv + base + kLanes;
#pragma unroll
    for (int t = 0; t < U; t++) {
        const unsigned o = (unsigned)t * kLanes + L.lane;
        g.phi[t] = pphi[o];
        g.xy[t] = pxy[o];
        g.a1[t] = pa1[o];
        g.a2[t] = pa2[o];
        g.b1[t] = pb1[o];
        g.b2[t] = pb2[o];
        double r0 = 0.0, r1 = 0.0;
        if (L.prev) {  // wave-uniform
            r0 = ld_agent(pdu + o);
            r1 = ld_agent(pdv + o);
        }
        g.duR[t] = r0;
        g.dvR[t] = r1;
        // halo: row 63 of the band above holds column s at its skew position s + 63 (lane 63);
        //       row 0 of the band below holds column s - 63 at its skew position s - 63 (lane 0).
        // A source that does not exist is redirected to a padding cell, which always reads 0.0.
        const int s = sg + t;
        const bool up_ok = L.up_any && s < L.W;                                    // wave-uniform
        const bool dn_ok = L.dn_any && s >= kLanes - 1 && s - (kLanes - 1) < L.W;  // wave-uniform
        const unsigned qu = up_ok ? L.up_src + (unsigned)s * kLanes : L.zero_src;  // scalar selects
        const unsigned qd = dn_ok ? L.dn_src + (unsigned)s * kLanes : L.zero_src;
        const unsigned q = L.is63 ? qd : qu;
        g.hu[t] = ld_agent(A.du + q);
        g.hv[t] = ld_agent(A.dv + q);
        g.hp[t] = A.phi[q];
    }
}

struct State {
    double duL, dvL, phiL, duC, dvC;
};

template <bool DPP>
__device__ __forceinline__ void compute_group(const ExactArgs& A, const Lane& L, int sg, const Group& g, State& S) {
    const double nalpha = A.nalpha, om1 = A.om1;
    const unsigned base = L.band + (unsigned)sg * kLanes;  // wave-uniform
    double* const pdu = A.du + base;
    double* const pdv = A.dv + base;
#pragma unroll
    for (int t = 0; t < U; t++) {
        const unsigned o = (unsigned)t * kLanes + L.lane;
        const double phiC = g.phi[t], duR = g.duR[t], dvR = g.dvR[t];
        const double duU = from_above<DPP>(g.hu[t], S.duL, L.is0);
        const double dvU = from_above<DPP>(g.hv[t], S.dvL, L.is0);
        const double phiU = from_above<DPP>(g.hp[t], S.phiL, L.is0);
        const double duD = from_below<DPP>(g.hu[t], duR, L.is63);
        const double dvD = from_below<DPP>(g.hv[t], dvR, L.is63);
        // Every operand that does not exist (image border, padding, first sweep) is an exact 0.0 here, so the
        // reference's conditional terms (src/OpticalFlow.cpp:468-495) reduce to adding +-0 in the same order.
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
        s1 += g.xy[t] * S.dvC;
        const double duN = om1 * S.duC + g.a1[t] * (g.b1[t] - s1);
        s2 += g.xy[t] * duN;
        const double dvN = om1 * S.dvC + g.a2[t] * (g.b2[t] - s2);
        st_agent(pdu + o, duN);  // padding cells compute and store an exact (+-)0
        st_agent(pdv + o, dvN);
        S.duL = duN;
        S.dvL = dvN;
        S.phiL = phiC;
        S.duC = duR;
        S.dvC = dvR;
    }
}

// Wait until the three producers of this task have published enough progress for steps [.., s1).
__device__ __forceinline__ bool wait_chunk(const ExactArgs& A, unsigned* p_own, unsigned* p_up, unsigned* p_dn,
                                           bool prev, bool has_up, bool has_dn, int s1) {
    const int ns = A.ns;
    const unsigned need_own = (unsigned)min(ns, s1 + 1);
    const unsigned need_up = (unsigned)min(ns, s1 + 63);
    const unsigned need_dn = (unsigned)min(ns, max(0, s1 - 63));
    unsigned spins = 0;
    for (;;) {
        // the three polls are issued together; a missing producer is polled on a satisfied dummy
        const unsigned a = prev ? __hip_atomic_load(p_own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need_own;
        const unsigned b = has_up ? __hip_atomic_load(p_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need_up;
        const unsigned c =
            (prev && has_dn) ? __hip_atomic_load(p_dn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need_dn;
        if (a >= need_own && b >= need_up && c >= need_dn) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0u) {
            if (__hip_atomic_load(A.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            if (spins > kSpinLimit) {
                __hip_atomic_store(A.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

template <bool DPP>
__global__ __launch_bounds__(64) void k_sor_exact(ExactArgs A) {
    const int lane = threadIdx.x;
    const int task = blockIdx.x;
    const int k = task / A.nb, b = task - k * A.nb;
    const int ns = A.ns, nsp = A.nsp;
    const bool has_up = b > 0, has_dn = (b + 1 < A.nb), prev = k > 0;
    Lane L;
    L.lane = (unsigned)lane;
    L.band = (unsigned)b * (unsigned)nsp * kLanes;
    L.is0 = lane == 0;
    L.is63 = lane == kLanes - 1;
    L.prev = prev;
    L.up_any = has_up;
    L.dn_any = prev && has_dn;
    L.W = A.W;
    // (band above, position s + 63, lane 63) and (band below, position s - 63, lane 0), relative to step 0;
    // the unsigned wrap of the unused / early ones is harmless: they are only dereferenced when *_ok holds.
    L.up_src = ((unsigned)(b - 1) * (unsigned)nsp + (kLanes - 1)) * kLanes + (kLanes - 1);
    L.dn_src = ((unsigned)(b + 1) * (unsigned)nsp - (kLanes - 1)) * kLanes;
    L.zero_src = L.band + (unsigned)(nsp - 1) * kLanes;  // last spare position of this band: never a real cell
    unsigned* const my_prog = A.prog + (size_t)k * A.nb + b;
    unsigned* const p_own = A.prog + (size_t)(k - 1) * A.nb + b;
    unsigned* const p_up = A.prog + (size_t)k * A.nb + (b - 1);
    unsigned* const p_dn = A.prog + (size_t)(k - 1) * A.nb + (b + 1);

    const int chunk = A.chunk;                        // multiple of 2U
    const int nsteps = (ns + 2 * U - 1) / (2 * U) * (2 * U);  // whole pairs of groups; <= nsp - 2U
    State S{0.0, 0.0, 0.0, 0.0, 0.0};
    Group ga, gb;

    if (!wait_chunk(A, p_own, p_up, p_dn, prev, has_up, has_dn, min(ns, chunk))) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler-only: keep the loads below the polls
    if (prev) {  // centre of the first cell (skew position 0: a real cell for lane 0, padding = 0 for the others)
        S.duC = ld_agent(A.du + L.band + L.lane);
        S.dvC = ld_agent(A.dv + L.band + L.lane);
    }
    load_group(A, L, 0, ga);
    for (int sg = 0; sg < nsteps; sg += 2 * U) {
        // ---- group A computes while group B's operands are in flight ----
        load_group(A, L, sg + U, gb);
        compute_group<DPP>(A, L, sg, ga, S);
        // ---- group B computes while the next pair's first group is in flight ----
        const int nxt = sg + 2 * U;
        const bool boundary = (nxt % chunk) == 0 || nxt >= nsteps;
        if (nxt < nsteps) {
            if (boundary) {
                if (!wait_chunk(A, p_own, p_up, p_dn, prev, has_up, has_dn, min(ns, nxt + chunk))) return;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            load_group(A, L, nxt, ga);
        }
        compute_group<DPP>(A, L, sg + U, gb, S);
        if (boundary) {
            // publish: every store of this wave has left the CU before the counter moves (guide G16/R1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(my_prog, (unsigned)min(ns, nxt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
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
                               int H, int W, double nalpha, double om1, int colour) {
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (i >= H) return;
    const int j = 2 * (blockIdx.x * 64 + threadIdx.x) + ((i + colour) & 1);
    if (j >= W) return;
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

}  // namespace

int sor_solve(papof_handle* h, const SorPlanes& sp, int H, int W, double alpha, double omega, int n_sor, int mode) {
    const double nalpha = -alpha, om1 = 1 - omega;
    if (n_sor <= 0) return PAPOF_EINVAL;
    if (mode == PAPOF_SOR_EXACT) {
        if (!sp.skew) return PAPOF_EINVAL;
        const SkewDims sd = skew_dims(H, W);
        if ((sd.n + kLanes) * sizeof(double) >= (size_t(1) << 32)) return PAPOF_EINVAL;  // 32-bit element offsets
        const size_t words = (size_t)sd.nb * n_sor + 4;
        if (words > h->sync_cap) {
            PAPOF_HIP(hipStreamSynchronize(h->stream));
            if (h->sync_words) PAPOF_HIP(hipFree(h->sync_words));
            h->sync_words = nullptr;
            h->sync_cap = 0;
            const size_t cap = (words + 1023) & ~size_t(1023);
            PAPOF_HIP(hipMalloc((void**)&h->sync_words, cap * sizeof(unsigned)));
            h->sync_cap = cap;
            PAPOF_HIP(hipMemsetAsync(h->sync_words, 0, cap * sizeof(unsigned), h->stream));
        }
        // word 0..3: abort block (kept across solves; checked by sor_check), counters start at word 4
        unsigned* prog = h->sync_words + 4;
        const size_t nbytes = (((size_t)sd.nb * n_sor * sizeof(unsigned)) + 15) & ~size_t(15);
        PAPOF_HIP(hipMemsetAsync(prog, 0, nbytes, h->stream));
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
        A.nsp = sd.nsp;
        A.n_sor = n_sor;
        A.chunk = std::max(2 * U, h->sor_chunk / (2 * U) * (2 * U));
        A.nalpha = nalpha;
        A.om1 = om1;
        if (h->use_dpp)
            hipLaunchKernelGGL(k_sor_exact<true>, dim3(sd.nb * n_sor), dim3(kLanes), 0, h->stream, A);
        else
            hipLaunchKernelGGL(k_sor_exact<false>, dim3(sd.nb * n_sor), dim3(kLanes), 0, h->stream, A);
        PAPOF_HIP(hipGetLastError());
        return PAPOF_OK;
    }
    if (sp.skew) return PAPOF_EINVAL;
    const size_t np = (size_t)H * W;
    PAPOF_HIP(hipMemsetAsync(sp.du, 0, np * sizeof(double), h->stream));  // src/OpticalFlow.cpp:452-453
    PAPOF_HIP(hipMemsetAsync(sp.dv, 0, np * sizeof(double), h->stream));
    if (mode == PAPOF_SOR_REDBLACK) {
        const dim3 grid(((W + 1) / 2 + 63) / 64, (H + 3) / 4), block(64, 4);
        for (int k = 0; k < n_sor; k++)
            for (int colour = 0; colour < 2; colour++)
                hipLaunchKernelGGL(k_sor_redblack, grid, block, 0, h->stream, sp.phi, sp.xy, sp.a1, sp.a2, sp.b1,
                                   sp.b2, sp.du, sp.dv, H, W, nalpha, om1, colour);
        PAPOF_HIP(hipGetLastError());
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
        return PAPOF_OK;
    }
    return PAPOF_EINVAL;
}

// All skew positions that are not real cells must read as 0.0 (the kernel relies on it instead of predicates).
// Real cells are rewritten by the assembly kernel each outer iteration and padding is only ever written with
// zeros, so one memset per (level, plane) suffices.
int sor_reset_planes(papof_handle* h, const SorPlanes& sp, int H, int W) {
    if (!sp.skew) return PAPOF_OK;
    const size_t bytes = (skew_dims(H, W).n + kLanes) * sizeof(double);
    double* planes[8] = {sp.phi, sp.xy, sp.a1, sp.a2, sp.b1, sp.b2, sp.du, sp.dv};
    for (double* p : planes) PAPOF_HIP(hipMemsetAsync(p, 0, bytes, h->stream));
    return PAPOF_OK;
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
    ok = ok && host[0] == 1000 && host[127] == 1063;                  // edge lanes keep their own `edge` operand
    h->use_dpp = ok;
    if (const char* s = std::getenv("PAPOF_SOR_XLANE")) {
        if (std::strcmp(s, "shfl") == 0) h->use_dpp = false;
    }
    return PAPOF_OK;
}

int sor_check(papof_handle* h) {
    if (!h->sync_words) return PAPOF_OK;
    unsigned flag = 0;
    PAPOF_HIP(hipMemcpy(&flag, h->sync_words, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (flag != 0) {
        PAPOF_HIP(hipMemset(h->sync_words, 0, 4 * sizeof(unsigned)));
        return PAPOF_ETIMEOUT;
    }
    return PAPOF_OK;
}

}  // namespace papof
