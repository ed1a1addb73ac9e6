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
#include "common.h"

namespace papof {

namespace {

constexpr unsigned kSpinLimit = 4u << 20;  // bounded wait: ~4M polls (seconds), then abort

struct ExactArgs {
    const double *phi, *xy, *a1, *a2, *b1, *b2;
    double *du, *dv;
    unsigned* prog;   // [n_sor][nb]
    unsigned* abort;  // one word
    int H, W, nb, ns, n_sor, chunk;
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

__global__ __launch_bounds__(64) void k_sor_exact(ExactArgs A) {
    const int lane = threadIdx.x;
    const int task = blockIdx.x;
    const int k = task / A.nb, b = task - k * A.nb;
    const int row = b * kLanes + lane;
    const bool rowok = row < A.H;
    const int W = A.W, ns = A.ns;
    const size_t band = (size_t)b * ns * kLanes + lane;  // + s*64 -> this lane's element at step s
    const bool has_up = b > 0, has_dn = (b + 1 < A.nb), prev = k > 0;
    // adjacent-band halo sources: row 63 of band b-1 at column j sits at skew position j+63, lane 63;
    // row 0 of band b+1 at column j at skew position j, lane 0.
    const size_t up_base = ((size_t)(b - 1) * ns + 63) * kLanes + 63;
    const size_t dn_base = (size_t)(b + 1) * ns * kLanes;
    unsigned* const my_prog = A.prog + (size_t)k * A.nb + b;
    unsigned* const p_own = A.prog + (size_t)(k - 1) * A.nb + b;
    unsigned* const p_up = A.prog + (size_t)k * A.nb + (b - 1);
    unsigned* const p_dn = A.prog + (size_t)(k - 1) * A.nb + (b + 1);
    const double nalpha = A.nalpha, om1 = A.om1;
    const bool top_row = row == 0, last_row = row >= A.H - 1;

    double duL = 0.0, dvL = 0.0, phiL = 0.0, duC = 0.0, dvC = 0.0;

    for (int s0 = 0; s0 < ns; s0 += A.chunk) {
        const int s1 = min(ns, s0 + A.chunk);
        bool ok = true;
        if (prev) ok = ok && wait_ge(p_own, (unsigned)min(ns, s1 + 1), A.abort);
        if (ok && has_up) ok = wait_ge(p_up, (unsigned)min(ns, s1 + 63), A.abort);
        if (ok && prev && has_dn) ok = wait_ge(p_dn, (unsigned)min(ns, max(0, s1 - 63)), A.abort);
        if (!ok) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // compiler-only: keep the loads below the polls

        if (s0 == 0 && prev && lane == 0) {  // centre of the first cell of row 64b (skew position 0)
            duC = ld_agent(A.du + band);
            dvC = ld_agent(A.dv + band);
        }
        for (int s = s0; s < s1; ++s) {
            const int j = s - lane;
            const bool valid = rowok && j >= 0 && j < W;
            const bool rvalid = rowok && j + 1 >= 0 && j + 1 < W;
            const size_t e = band + (size_t)s * kLanes;
            double phiC = 0.0, xy = 0.0, a1 = 0.0, a2 = 0.0, b1 = 0.0, b2 = 0.0;
            if (valid) {
                phiC = A.phi[e];
                xy = A.xy[e];
                a1 = A.a1[e];
                a2 = A.a2[e];
                b1 = A.b1[e];
                b2 = A.b2[e];
            }
            double duR = 0.0, dvR = 0.0;
            if (prev && rvalid) {
                duR = ld_agent(A.du + e + kLanes);
                dvR = ld_agent(A.dv + e + kLanes);
            }
            // cross-lane neighbours
            double duU = __shfl_up(duL, 1), dvU = __shfl_up(dvL, 1), phiU = __shfl_up(phiL, 1);
            double duD = __shfl_down(duR, 1), dvD = __shfl_down(dvR, 1);
            if (lane == 0) {
                duU = dvU = phiU = 0.0;
                if (has_up && s < W) {  // column j = s of the row above
                    const size_t q = up_base + (size_t)s * kLanes;
                    duU = ld_agent(A.du + q);
                    dvU = ld_agent(A.dv + q);
                    phiU = A.phi[q];
                }
            }
            if (lane == kLanes - 1) {
                duD = dvD = 0.0;
                const int j63 = s - (kLanes - 1);
                if (prev && has_dn && j63 >= 0 && j63 < W) {
                    const size_t q = dn_base + (size_t)j63 * kLanes;
                    duD = ld_agent(A.du + q);
                    dvD = ld_agent(A.dv + q);
                }
            }
            const double wL = phiL;
            const double wR = (j < W - 1) ? phiC : 0.0;
            const double wU = top_row ? 0.0 : phiU;
            const double wD = last_row ? 0.0 : phiC;
            double s1v = wL * duL;
            double s2v = wL * dvL;
            s1v += wR * duR;
            s2v += wR * dvR;
            s1v += wU * duU;
            s2v += wU * dvU;
            s1v += wD * duD;
            s2v += wD * dvD;
            s1v *= nalpha;
            s2v *= nalpha;
            s1v += xy * dvC;
            double duN = om1 * duC + a1 * (b1 - s1v);
            s2v += xy * duN;
            double dvN = om1 * dvC + a2 * (b2 - s2v);
            if (!valid) {
                duN = 0.0;
                dvN = 0.0;
            } else {
                st_agent(A.du + e, duN);
                st_agent(A.dv + e, dvN);
            }
            duL = duN;
            dvL = dvN;
            phiL = phiC;
            duC = duR;
            dvC = dvR;
        }
        // publish: every store of this wave has left the CU before the counter moves (guide G16/R1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(my_prog, (unsigned)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        A.n_sor = n_sor;
        A.chunk = 16;
        A.nalpha = nalpha;
        A.om1 = om1;
        hipLaunchKernelGGL(k_sor_exact, dim3(sd.nb * n_sor), dim3(kLanes), 0, h->stream, A);
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
