// papteam_opticalflow_amd/csrc/kernels.hip -- every stage of the hot path except the SOR sweeps.
//
// One thread per output pixel, 64 x 4 thread blocks with threadIdx.x along the row so that each
// wavefront reads/writes 512 contiguous bytes of a plane (planar fp64 layout, see common.h).
// All kernels are HBM-bound stencils / gathers (no dense contraction => no MFMA).  Floating-point
// operation ORDER follows the reference line by line (cited per kernel; paths relative to
// /root/reference/Code/Serial/) because the parity bar is bit-compatibility, and the file is compiled
// with -ffp-contract=off for the same reason.
#include "common.h"

#include <cmath>

namespace papof {

namespace {

constexpr int BX = 64, BY = 4;

inline dim3 grid2d(int W, int H, int planes = 1) { return dim3((W + BX - 1) / BX, (H + BY - 1) / BY, planes); }
inline Rect region(const Rect* rc, int W, int H) { return rc ? *rc : Rect{0, 0, W, H}; }
inline dim3 grid2d(const Rect& r, int planes = 1) { return grid2d(r.x1 - r.x0, r.y1 - r.y0, planes); }

// Phase stamp: the first thread of a kernel's first block writes the constant 100 MHz clock where the host asked for
// it (flow_internal.h: PhaseClock in stamp mode) -- the start of the first kernel of a phase IS the start of the phase,
// at no cost to the stream (a HIP event between two kernels costs ~2.6 us of stream time; ~100 per call = 2 %).
__device__ __forceinline__ void stamp_now(unsigned long long* s) {
    if (s && (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x | threadIdx.y) == 0u) *s = __builtin_amdgcn_s_memrealtime();
}
__global__ void k_stamp(unsigned long long* s) { stamp_now(s); }

__device__ __forceinline__ int clampi(int x, int n) {  // EnforceRange, src/ImageProcessing.h:34
    x = x < 0 ? 0 : x;
    return x > n - 1 ? n - 1 : x;
}

// Tiled kernels launched on a 1-D grid of xcd_grid(tiles) blocks (x; channels, if any, in z): workgroups are dealt round-robin
// over the 8 XCDs (observed; speed only), so block b works on tile (b % 8) * per + b / 8 -- a contiguous run of tiles (row-major)
// per XCD: the halo rows and columns neighbouring tiles both read are found in that XCD's L2.  false: no tile for this block.
__device__ __forceinline__ bool xcd_tile(int ntx, int nty, int& tx, int& ty) {
    const int ntiles = ntx * nty, per = (ntiles + 7) >> 3;
    const int vt = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || vt >= ntiles) return false;
    ty = vt / ntx;
    tx = vt - ty * ntx;
    return true;
}
static inline unsigned xcd_grid(int ntx, int nty) { return 8u * (unsigned)(((size_t)ntx * nty + 7) / 8); }

// ------------------------------------------------------------------------------------------------
// layout conversion (entry / exit of the call): HWC interleaved <-> planar
// ------------------------------------------------------------------------------------------------
__global__ void k_hwc_to_planar(const double* __restrict__ hwc, double* __restrict__ planar, int H, int W, int C) {
    const int j = blockIdx.x * BX + threadIdx.x, i = blockIdx.y * BY + threadIdx.y;
    if (j >= W || i >= H) return;
    const size_t o = (size_t)i * W + j, np = (size_t)H * W;
    hwc += blockIdx.z * np * C;  // blockIdx.z: the frame of a batch (frames are contiguous on both sides)
    planar += blockIdx.z * np * C;
    for (int k = 0; k < C; k++) planar[k * np + o] = hwc[o * C + k];
}

// uint8 frames (what the caller decodes from JPEG): the caller's `im.astype(float) / 255.` (OpticalFlowCalculation.py:69-70)
// is done on load -- one IEEE fp64 division per sample, hence the same bits -- so 1 byte per sample crosses PCIe, not 8.
__global__ void k_hwc_u8_to_planar(const unsigned char* __restrict__ hwc, double* __restrict__ planar, int H, int W,
                                   int C) {
    const int j = blockIdx.x * BX + threadIdx.x, i = blockIdx.y * BY + threadIdx.y;
    if (j >= W || i >= H) return;
    const size_t o = (size_t)i * W + j, np = (size_t)H * W;
    hwc += blockIdx.z * np * C;  // blockIdx.z: the frame of a batch
    planar += blockIdx.z * np * C;
    for (int k = 0; k < C; k++) planar[k * np + o] = (double)hwc[o * C + k] / 255.0;
}

__global__ void k_planar_to_hwc(const double* __restrict__ planar, double* __restrict__ hwc, int H, int W, int C) {
    const int j = blockIdx.x * BX + threadIdx.x, i = blockIdx.y * BY + threadIdx.y;
    if (j >= W || i >= H) return;
    const size_t o = (size_t)i * W + j, np = (size_t)H * W;
    for (int k = 0; k < C; k++) hwc[o * C + k] = planar[k * np + o];
}

// ------------------------------------------------------------------------------------------------
// separable correlation with clamped borders: src/ImageProcessing.h:259-279 (h), :350-369 (v).
// Accumulation into a zeroed destination, taps in order l = -fsize..fsize.
// ------------------------------------------------------------------------------------------------
__global__ void k_filter_h(const double* __restrict__ src, double* __restrict__ dst, int H, int W, Taps f, Rect rc) {
    const int j = rc.x0 + blockIdx.x * BX + threadIdx.x, i = rc.y0 + blockIdx.y * BY + threadIdx.y;
    if (j >= rc.x1 || i >= rc.y1) return;
    const size_t np = (size_t)H * W;
    const double* row = src + blockIdx.z * np + (size_t)i * W;
    double acc = 0.0;
    for (int l = -f.fsize; l <= f.fsize; l++) acc += row[clampi(j + l, W)] * f.t[l + f.fsize];
    dst[blockIdx.z * np + (size_t)i * W + j] = acc;
}

__global__ void k_filter_v(const double* __restrict__ src, double* __restrict__ dst, int H, int W, Taps f, Rect rc) {
    const int j = rc.x0 + blockIdx.x * BX + threadIdx.x, i = rc.y0 + blockIdx.y * BY + threadIdx.y;
    if (j >= rc.x1 || i >= rc.y1) return;
    const size_t np = (size_t)H * W;
    const double* p = src + blockIdx.z * np + j;
    double acc = 0.0;
    for (int l = -f.fsize; l <= f.fsize; l++) acc += p[(size_t)clampi(i + l, H) * W] * f.t[l + f.fsize];
    dst[blockIdx.z * np + (size_t)i * W + j] = acc;
}

// Both passes of a separable filter in ONE kernel (h pass into LDS, v pass out of it): the operations and their order
// are those of k_filter_h followed by k_filter_v -- the v pass reads the h-filtered values of the rows clamp(i + l), here
// rows of the block's LDS image -- so the bits are the same, but the h-filtered plane never goes to HBM (2.4 instead of
// 4 plane transfers at half-width 3).  A block owns kHvRows rows x 64 columns of one plane.
#ifndef PAPOF_V_HVROWS
#define PAPOF_V_HVROWS 64  // in-pair A/B (round 2): 16 / 32 / 64 / 96 / 128 rows: 10.99 / 10.98 / 10.92 / 10.91 / 10.94 ms per 1080p pair
#endif
constexpr int kHvRows = PAPOF_V_HVROWS, kHvMaxF = 4;
template <int F>  // F = the half-width of both passes as a compile-time constant (fully unrolled tap loops), or -1: run time
__global__ __launch_bounds__(256) void k_filter_hv(const double* __restrict__ src, double* __restrict__ dst, int H, int W,
                                                   Taps fh, Taps fv) {
    __shared__ double hs[kHvRows + 2 * kHvMaxF][BX];
    const int j = blockIdx.x * BX + threadIdx.x, i0 = blockIdx.y * kHvRows;
    const int f = F >= 0 ? F : fv.fsize, fx = F >= 0 ? F : fh.fsize;
    const size_t np = (size_t)H * W;
    const double* plane = src + blockIdx.z * np;
    if (j < W) {
        for (int r = threadIdx.y; r < kHvRows + 2 * f; r += BY) {
            const double* row = plane + (size_t)clampi(i0 + r - f, H) * W;
            double acc = 0.0;
#pragma unroll
            for (int l = -fx; l <= fx; l++) acc += row[clampi(j + l, W)] * fh.t[l + fx];
            hs[r][threadIdx.x] = acc;
        }
    }
    __syncthreads();
    if (j >= W) return;
    for (int r = threadIdx.y; r < kHvRows; r += BY) {
        const int i = i0 + r;
        if (i >= H) break;
        double acc = 0.0;
#pragma unroll
        for (int l = -f; l <= f; l++) acc += hs[r + l + f][threadIdx.x] * fv.t[l + f];
        dst[blockIdx.z * np + (size_t)i * W + j] = acc;
    }
}

// The same two passes with the source tile (halo included) staged in LDS first: every input value is loaded once, by a
// coalesced row read, instead of 2F+1 times through clamped indices (the kernel above is bound by its L1 traffic and index
// arithmetic).  Same operations in the same order: same bits.
#ifndef PAPOF_V_HVSTAGED
#define PAPOF_V_HVSTAGED 1  // same-box A/B (round 3, ms per 1080p pair): 10.536 without, 10.507 with 16-row tiles, 10.61 with 32
#endif
#ifndef PAPOF_V_HVSROWS
#define PAPOF_V_HVSROWS 16
#endif
constexpr int kHvsRows = PAPOF_V_HVSROWS;
template <int F>
__global__ __launch_bounds__(256) void k_filter_hv_staged(const double* __restrict__ src, double* __restrict__ dst, int H,
                                                          int W, Taps fh, Taps fv) {
    __shared__ double raw[kHvsRows + 2 * F][BX + 2 * F];
    __shared__ double hs[kHvsRows + 2 * F][BX];
    int tx, ty;
    if (!xcd_tile((W + BX - 1) / BX, (H + kHvsRows - 1) / kHvsRows, tx, ty)) return;  // whole workgroup, before any barrier
    const int j0 = tx * BX, i0 = ty * kHvsRows;
    const size_t np = (size_t)H * W;
    const double* plane = src + blockIdx.z * np;
    constexpr int kCells = (kHvsRows + 2 * F) * (BX + 2 * F);
    for (int c = threadIdx.y * BX + threadIdx.x; c < kCells; c += BX * BY) {
        const int r = c / (BX + 2 * F), cc = c - r * (BX + 2 * F);
        raw[r][cc] = plane[(size_t)clampi(i0 + r - F, H) * W + clampi(j0 + cc - F, W)];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < kHvsRows + 2 * F; r += BY) {
        double acc = 0.0;
#pragma unroll
        for (int l = -F; l <= F; l++) acc += raw[r][threadIdx.x + F + l] * fh.t[l + F];
        hs[r][threadIdx.x] = acc;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= W) return;
    for (int r = threadIdx.y; r < kHvsRows; r += BY) {
        const int i = i0 + r;
        if (i >= H) break;
        double acc = 0.0;
#pragma unroll
        for (int l = -F; l <= F; l++) acc += hs[r + l + F][threadIdx.x] * fv.t[l + F];
        dst[blockIdx.z * np + (size_t)i * W + j] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// bilinear sampling: src/ImageProcessing.h:138-157.  Integer part by truncation toward zero,
// fraction clamped to [0,1], taps visited x-offset outer / y-offset inner and ACCUMULATED from 0.
// The four weights are shared by all planes of a pixel.
// ------------------------------------------------------------------------------------------------
struct BilinearTaps {
    int o[4];     // plane offsets of the 4 taps in visiting order
    double s[4];  // their weights
};

__device__ __forceinline__ BilinearTaps bilinear_taps(int W, int H, double x, double y) {
    BilinearTaps b;
    const int xx = (int)x, yy = (int)y;
    double dx = x - xx, dy = y - yy;
    dx = dx > 1 ? 1.0 : dx;
    dx = dx < 0 ? 0.0 : dx;
    dy = dy > 1 ? 1.0 : dy;
    dy = dy < 0 ? 0.0 : dy;
#pragma unroll
    for (int m = 0; m <= 1; m++)
#pragma unroll
        for (int n = 0; n <= 1; n++) {
            const int u = clampi(xx + m, W), v = clampi(yy + n, H);
            b.o[m * 2 + n] = v * W + u;
            b.s[m * 2 + n] = fabs((double)(1 - m) - dx) * fabs((double)(1 - n) - dy);
        }
    return b;
}

__device__ __forceinline__ double bilinear_apply(const double* __restrict__ p, const BilinearTaps& b) {
    double acc = 0.0;
#pragma unroll
    for (int t = 0; t < 4; t++) acc += p[b.o[t]] * b.s[t];
    return acc;
}

// ImageProcessing::ResizeImage, src/ImageProcessing.h:214-253 (both overloads: xr == yr for the ratio form).
// Optional post-scale = Image::Multiplywith (src/Image.h:1841-1850) for the flow up-sampling of
// src/OpticalFlow.cpp:809-812.
__global__ void k_resize(const double* __restrict__ src, double* __restrict__ dst, int sh, int sw, int dh, int dw,
                         double xr, double yr, int use_post, double post, Rect rc, unsigned long long* stamp) {
    stamp_now(stamp);
    int tbx, tby;  // a contiguous run of blocks per XCD (xcd_tile): the four taps of neighbouring rows meet in one L2
    if (!xcd_tile((rc.x1 - rc.x0 + BX - 1) / BX, (rc.y1 - rc.y0 + BY - 1) / BY, tbx, tby)) return;
    const int j = rc.x0 + tbx * BX + threadIdx.x, i = rc.y0 + tby * BY + threadIdx.y;
    if (j >= rc.x1 || i >= rc.y1) return;
    const double x = (double)(j + 1) / xr - 1;
    const double y = (double)(i + 1) / yr - 1;
    const BilinearTaps b = bilinear_taps(sw, sh, x, y);
    double r = bilinear_apply(src + blockIdx.z * (size_t)sh * sw, b);
    if (use_post) r *= post;
    dst[blockIdx.z * (size_t)dh * dw + (size_t)i * dw + j] = r;
}

// ------------------------------------------------------------------------------------------------
// OpticalFlow::im2feature, src/OpticalFlow.cpp:911-961 (3-channel branch :934-955, 1-channel :918-933);
// desaturate src/Image.h:1461-1480 with colorType RGB; 5-point derivative taps {1,-8,0,8,-1}/12
// (src/Image.h:987-992) applied to the gray image with clamped borders.  Gray values of the stencil
// neighbours are recomputed (same three-term expression => same bits) instead of staged.
// ------------------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ double gray_at(const double* __restrict__ im, size_t np, size_t o) {
    if (C == 3) return im[o] * .299 + im[np + o] * .587 + im[2 * np + o] * .114;
    return im[o];
}

template <int C>
__global__ void k_im2feature(const double* __restrict__ im, double* __restrict__ feat, int H, int W, Taps d) {
    const int j = blockIdx.x * BX + threadIdx.x, i = blockIdx.y * BY + threadIdx.y;
    if (j >= W || i >= H) return;
    const size_t np = (size_t)H * W, o = (size_t)i * W + j;
    double gx = 0.0, gy = 0.0;
#pragma unroll
    for (int l = -2; l <= 2; l++) gx += gray_at<C>(im, np, (size_t)i * W + clampi(j + l, W)) * d.t[l + 2];
#pragma unroll
    for (int l = -2; l <= 2; l++) gy += gray_at<C>(im, np, (size_t)clampi(i + l, H) * W + j) * d.t[l + 2];
    feat[o] = gray_at<C>(im, np, o);
    feat[np + o] = gx;
    feat[2 * np + o] = gy;
    if (C == 3) {
        feat[3 * np + o] = im[np + o] - im[o];
        feat[4 * np + o] = im[np + o] - im[2 * np + o];
    }
}

// The same through an LDS tile of gray values (one-GPU path): a block owns kFeatRows rows x 64 columns, evaluates gray
// ONCE per cell of the tile grown by the 5-tap filters' reach (same expression: same bits) and takes the two derivative
// sums out of LDS in the same order -- 4 instead of 27 cached loads per pixel.
#ifndef PAPOF_V_FEATROWS
#define PAPOF_V_FEATROWS 16
#endif
constexpr int kFeatRows = PAPOF_V_FEATROWS;
template <int C, bool BATCH = false>
__global__ __launch_bounds__(256) void k_im2feature_tiled(const double* __restrict__ im, double* __restrict__ feat, int H,
                                                          int W, Taps d, unsigned* __restrict__ nz, unsigned mark,
                                                          size_t b_im, size_t b_feat, size_t b_nz) {
    // nz (may be null): nz[k] is set when feature channel k has a non-zero value anywhere -- a channel that is all zero in BOTH
    // frames has no valid sample for estLaplacianNoise (every |Im1 - warpIm2| is exactly 0), so its LapPara is 0.001 whatever
    // the flow (api.hip: LapGuard).  One ballot per wave, at most one store per wave and channel.
    __shared__ double g[kFeatRows + 4][BX + 4];
    int tx, ty;
    if (!xcd_tile((W + BX - 1) / BX, (H + kFeatRows - 1) / kFeatRows, tx, ty)) return;  // whole workgroup, before any barrier
    const int j0 = tx * BX, i0 = ty * kFeatRows;
    const size_t np = (size_t)H * W;
    if (BATCH) {  // blockIdx.y: the frame of a batch (common.h: BatchK)
        im += blockIdx.y * b_im;
        feat += blockIdx.y * b_feat;
        if (nz != nullptr) nz += blockIdx.y * b_nz;
    }
    for (int c = threadIdx.y * BX + threadIdx.x; c < (kFeatRows + 4) * (BX + 4); c += BX * BY) {
        const int r = c / (BX + 4), cc = c - r * (BX + 4);
        g[r][cc] = gray_at<C>(im, np, (size_t)clampi(i0 + r - 2, H) * W + clampi(j0 + cc - 2, W));
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    bool any[5] = {false, false, false, false, false};
    if (j < W) {
        for (int r = threadIdx.y; r < kFeatRows; r += BY) {
            const int i = i0 + r;
            if (i >= H) break;
            const size_t o = (size_t)i * W + j;
            double gx = 0.0, gy = 0.0;
#pragma unroll
            for (int l = -2; l <= 2; l++) gx += g[r + 2][threadIdx.x + 2 + l] * d.t[l + 2];
#pragma unroll
            for (int l = -2; l <= 2; l++) gy += g[r + 2 + l][threadIdx.x + 2] * d.t[l + 2];
            const double f0 = g[r + 2][threadIdx.x + 2];
            feat[o] = f0;
            feat[np + o] = gx;
            feat[2 * np + o] = gy;
            any[0] |= f0 != 0.0;
            any[1] |= gx != 0.0;
            any[2] |= gy != 0.0;
            if (C == 3) {
                const double gg = im[np + o];
                const double f3 = gg - im[o], f4 = gg - im[2 * np + o];
                feat[3 * np + o] = f3;
                feat[4 * np + o] = f4;
                any[3] |= f3 != 0.0;
                any[4] |= f4 != 0.0;
            }
        }
    }
    if (nz != nullptr) {
#pragma unroll
        for (int k = 0; k < (C == 3 ? 5 : 3); k++)
            if (__ballot(any[k]) != 0ull && threadIdx.x == 0) nz[k] = mark;
    }
}

// The three derivative planes of the final bicubic warp in one launch (Image::warpImageBicubicRef, src/Image.h:2590-2594:
// imfilter_h, imfilter_v, imfilter_v of the first with {-.5, 0, .5}): gx, gy as k_filter_h / k_filter_v compute them,
// gxy = the v pass over gx, whose three rows are recomputed here with the h pass's own operations (same bits).
__global__ void k_central3_all(const double* __restrict__ src, double* __restrict__ gx, double* __restrict__ gy,
                               double* __restrict__ gxy, int H, int W, Taps c3) {
    int tbx, tby;
    if (!xcd_tile((W + BX - 1) / BX, (H + BY - 1) / BY, tbx, tby)) return;
    const int j = tbx * BX + threadIdx.x, i = tby * BY + threadIdx.y;
    if (j >= W || i >= H) return;
    const size_t np = (size_t)H * W;
    const double* p = src + blockIdx.z * np;
    const auto hrow = [&](int ii) {
        const double* row = p + (size_t)ii * W;
        double acc = 0.0;
#pragma unroll
        for (int l = -1; l <= 1; l++) acc += row[clampi(j + l, W)] * c3.t[l + 1];
        return acc;
    };
    const double hx[3] = {hrow(clampi(i - 1, H)), hrow(i), hrow(clampi(i + 1, H))};
    double vy = 0.0, vxy = 0.0;
#pragma unroll
    for (int l = -1; l <= 1; l++) vy += p[(size_t)clampi(i + l, H) * W + j] * c3.t[l + 1];
#pragma unroll
    for (int l = -1; l <= 1; l++) vxy += hx[l + 1] * c3.t[l + 1];
    const size_t o = blockIdx.z * np + (size_t)i * W + j;
    gx[o] = hx[1];
    gy[o] = vy;
    gxy[o] = vxy;
}

// ------------------------------------------------------------------------------------------------
// OpticalFlow::warpFL -> ImageProcessing::warpImage, src/ImageProcessing.h:483-503.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void warp_pixel(const double* __restrict__ im1, const double* __restrict__ im2,
                                           double* __restrict__ out, double fx, double fy, int i, int j, int H,
                                           int W, int planes) {
    const size_t np = (size_t)H * W, o = (size_t)i * W + j;
    const double y = i + fy;
    const double x = j + fx;
    if (x < 0 || x > W - 1 || y < 0 || y > H - 1) {
        for (int k = 0; k < planes; k++) out[k * np + o] = im1[k * np + o];
        return;
    }
    const BilinearTaps b = bilinear_taps(W, H, x, y);
    for (int k = 0; k < planes; k++) out[k * np + o] = bilinear_apply(im2 + k * np, b);
}

__global__ void k_warp(const double* __restrict__ im1, const double* __restrict__ im2, const double* __restrict__ vx,
                       const double* __restrict__ vy, double* __restrict__ out, int H, int W, int planes, Rect rc,
                       unsigned long long* stamp) {
    stamp_now(stamp);
    const int j = rc.x0 + blockIdx.x * BX + threadIdx.x, i = rc.y0 + blockIdx.y * BY + threadIdx.y;
    if (j >= rc.x1 || i >= rc.y1) return;
    const size_t o = (size_t)i * W + j;
    warp_pixel(im1, im2, out, vx[o], vy[o], i, j, H, W, planes);
}

// ------------------------------------------------------------------------------------------------
// Second half of OpticalFlow::getDxs (src/OpticalFlow.cpp:89-97) for the warped frame: vertical pass of
// the 5-tap smoothing, then Im = Im1s*0.4 + Im2s*0.6 (Multiplywith :92, Add(...,0.6) :93) and
// imdt = Im2s - Im1s (:97).  Im1s (smoothed frame 1) is constant within a level and computed once.
// ------------------------------------------------------------------------------------------------
__global__ void k_smooth_v_blend(const double* __restrict__ tmp, const double* __restrict__ im1s,
                                 double* __restrict__ blend, double* __restrict__ imdt, int H, int W, Taps g, Rect rc) {
    const int j = rc.x0 + blockIdx.x * BX + threadIdx.x, i = rc.y0 + blockIdx.y * BY + threadIdx.y;
    if (j >= rc.x1 || i >= rc.y1) return;
    const size_t np = (size_t)H * W, o = blockIdx.z * np + (size_t)i * W + j;
    const double* p = tmp + blockIdx.z * np + j;
    double s2 = 0.0;
#pragma unroll
    for (int l = -2; l <= 2; l++) s2 += p[(size_t)clampi(i + l, H) * W] * g.t[l + 2];
    const double s1 = im1s[o];
    double t = s1;
    t *= 0.4;
    t += s2 * 0.6;
    blend[o] = t;
    imdt[o] = s2 - s1;
}

// Both passes of the 5-tap smoothing of the warped frame + blend + imdt in ONE kernel (one-GPU path): a block owns
// kFuseRows rows x 64 columns of one plane, filters kFuseRows + 4 rows horizontally into LDS (same taps, same order as
// k_filter_h), then runs the vertical pass out of LDS (same order as k_smooth_v_blend).  The horizontally filtered
// plane never goes to HBM: 4 instead of 6 plane transfers per channel, same bits.
constexpr int kFuseRows = 16;
__global__ __launch_bounds__(256) void k_smooth_hv_blend(const double* __restrict__ warp,
                                                         const double* __restrict__ im1s,
                                                         double* __restrict__ blend, double* __restrict__ imdt, int H,
                                                         int W, Taps g, unsigned long long* stamp, int row0, int row1) {
    __shared__ double hs[kFuseRows + 4][BX];
    stamp_now(stamp);
    // rows row0 .. row1-1 are written (a strip of the plane, api.hip: smooth_flow_strips; the whole plane otherwise)
    const int j = blockIdx.x * BX + threadIdx.x, i0 = row0 + blockIdx.y * kFuseRows;
    const size_t np = (size_t)H * W;
    const double* src = warp + blockIdx.z * np;
    if (j < W) {
        for (int r = threadIdx.y; r < kFuseRows + 4; r += BY) {
            const double* row = src + (size_t)clampi(i0 + r - 2, H) * W;
            double acc = 0.0;
#pragma unroll
            for (int l = -2; l <= 2; l++) acc += row[clampi(j + l, W)] * g.t[l + 2];
            hs[r][threadIdx.x] = acc;
        }
    }
    __syncthreads();
    if (j >= W) return;
    for (int r = threadIdx.y; r < kFuseRows; r += BY) {
        const int i = i0 + r;
        if (i >= row1) break;
        const size_t o = blockIdx.z * np + (size_t)i * W + j;
        double s2 = 0.0;
#pragma unroll
        for (int l = -2; l <= 2; l++) s2 += hs[r + l + 2][threadIdx.x] * g.t[l + 2];
        const double s1 = im1s[o];
        double t = s1;
        t *= 0.4;
        t += s2 * 0.6;
        blend[o] = t;
        imdt[o] = s2 - s1;
    }
}

// The same with the WARP folded in (one-GPU default path): the warped frame 2 (OpticalFlow::warpFL, src/OpticalFlow.cpp:
// 154-159, 516) is read by nobody but this smoothing, so it is never written to HBM -- a block evaluates warp_value() for
// every cell of its tile grown by the filters' reach (clamped coordinates, i.e. exactly the cells k_smooth_hv_blend would
// have loaded from the warped plane: same bits), keeps them in LDS, and runs both passes out of LDS.  Saves the plane's
// round trip (16 B x channels per pixel and iteration) and the warp kernel at the start of every level.
__device__ __forceinline__ double warp_value(const double* __restrict__ im1, const double* __restrict__ im2, double fx,
                                             double fy, int i, int j, int H, int W) {  // one plane of warp_pixel()
    const double y = i + fy;
    const double x = j + fx;
    if (x < 0 || x > W - 1 || y < 0 || y > H - 1) return im1[(size_t)i * W + j];
    return bilinear_apply(im2, bilinear_taps(W, H, x, y));
}
#ifndef PAPOF_V_WSROWS
#define PAPOF_V_WSROWS 16
#endif
#ifndef PAPOF_V_WSBATCH
#define PAPOF_V_WSBATCH 3
#endif
constexpr int kWsBatch = PAPOF_V_WSBATCH;  // cells per thread whose gathers are in flight together (k_warp_smooth_blend)
constexpr int kWsRows = PAPOF_V_WSROWS;  // rows per block of the warp-folded smoothing (32 rows: 0.13 ms per 1080p pair slower)
__global__ __launch_bounds__(256) void k_warp_smooth_blend(const double* __restrict__ im1, const double* __restrict__ im2,
                                                           const double* __restrict__ u, const double* __restrict__ v,
                                                           const double* __restrict__ im1s, double* __restrict__ blend,
                                                           double* __restrict__ imdt, int H, int W, Taps g,
                                                           unsigned long long* stamp, unsigned* __restrict__ wit,
                                                           double wit_thr, unsigned mark, double* __restrict__ phi_out) {
    // one channel per block (blockIdx.z).  [Also measured: one block looping over the channels with the sampling taps of its
    // cells kept in registers -- 0.2 ms per 1080p pair SLOWER (register pressure, ten barriers per block).]
    __shared__ double raw[kWsRows + 4][BX + 4];
    __shared__ double hs[kWsRows + 4][BX];
    stamp_now(stamp);
    const int j0 = blockIdx.x * BX, i0 = blockIdx.y * kWsRows;
    const size_t np = (size_t)H * W;
    const double *p1 = im1 + blockIdx.z * np, *p2 = im2 + blockIdx.z * np;
    // WITNESS for the Laplacian-noise guard (api.hip: LapGuard): estLaplacianNoise (src/OpticalFlow.cpp:594-639) averages
    // |Im1 - warpIm2| over its samples in (0, 1e6), and ONE sample of at least wit_thr = 2e-20 x pixels proves that mean is not
    // below 1e-20 (a sum of positives is at least its largest term), i.e. that the guard of :399-400 did not trip.  This kernel
    // evaluates exactly that warp -- frame 2 at the flow the previous outer iteration left -- so the first wave of every
    // block compares 64 cells spread over the tile (their frame-1 values are fetched here, beside the gathers: no latency of
    // their own; halo cells are clamped copies of image pixels, warped as such: as good a sample as any).
    // Where the grid is large, one block in sixteen samples: thousands of stores to one word are not free.
    // A level that is ONE block per channel is checked EXHAUSTIVELY instead (every cell of the tile): on the few-pixel levels of a
    // deep pyramid the flow can leave the image altogether -- every warped value is then frame 1's, no sample is valid and the
    // estimate is the constant 0.001; the block that has seen every pixel says so (kLapNone), which is a proof as well.
    const bool exhaustive = wit != nullptr && gridDim.x == 1u && gridDim.y == 1u;
    bool ex_hit = false, ex_valid = false;
    const bool sampler = wit != nullptr && !exhaustive && threadIdx.y == 0u &&
                         (gridDim.x * gridDim.y <= 256u || ((blockIdx.x + 5u * blockIdx.y) & 15u) == 0u);
    constexpr int kWitStride = (kWsRows + 4) * (BX + 4) / 64;
    const int wr = (int)(threadIdx.x * kWitStride) / (BX + 4), wc = (int)(threadIdx.x * kWitStride) - wr * (BX + 4);
    double own1 = 0.0;
    if (sampler) own1 = p1[(size_t)clampi(i0 + wr - 2, H) * W + clampi(j0 + wc - 2, W)];
    // The tile's cells (halo of 2 included) are warped in BATCHES of kWsBatch cells per thread: first the flow of every cell of
    // the batch, then all their taps and the 4 x kWsBatch gathers, then the blends -- two dependent memory round trips per
    // batch instead of two per cell (the kernel was bound by those: 130 us at level 0 against ~45 us of HBM traffic).  Same
    // expressions as warp_value(): same bits.
    constexpr int kCells = (kWsRows + 4) * (BX + 4), kPer = (kCells + BX * BY - 1) / (BX * BY);
    const int tid = threadIdx.y * BX + threadIdx.x;
#pragma unroll
    for (int q0 = 0; q0 < kPer; q0 += kWsBatch) {
        double fx[kWsBatch], fy[kWsBatch], val[kWsBatch][4];
        int off[kWsBatch];
#pragma unroll
        for (int q = 0; q < kWsBatch; q++) {
            const int c = min(tid + (q0 + q) * BX * BY, kCells - 1);  // (a thread beyond the tile repeats the last cell: discarded)
            const int r = c / (BX + 4), cc = c - r * (BX + 4);
            const int i = clampi(i0 + r - 2, H), j = clampi(j0 + cc - 2, W);
            off[q] = i * W + j;
            fx[q] = u[off[q]];
            fy[q] = v[off[q]];
        }
        bool outside[kWsBatch];
#pragma unroll
        for (int q = 0; q < kWsBatch; q++) {
            const int c = min(tid + (q0 + q) * BX * BY, kCells - 1);
            const int r = c / (BX + 4), cc = c - r * (BX + 4);
            const int i = clampi(i0 + r - 2, H), j = clampi(j0 + cc - 2, W);
            double y = i + fy[q], x = j + fx[q];
            outside[q] = x < 0 || x > W - 1 || y < 0 || y > H - 1;  // warp_value(): such a pixel takes frame 1's value
            if (outside[q]) {  // (any valid position: its gathers are issued and not used)
                x = j;
                y = i;
            }
            // bilinear_taps() for a position INSIDE the image: 0 <= (int)x <= W-1 needs no clamp from below, dx = x - (int)x is in
            // [0, 1) as it is, |1 - dx| = 1 - dx and |0 - dx| = dx -- the same values with a third of the instructions
            const int x0 = (int)x, y0 = (int)y;
            const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
            val[q][0] = p2[y0 * W + x0];  // visiting order of bilinear_apply(): (m, n) = (0,0) (0,1) (1,0) (1,1), tap = (x_m, y_n)
            val[q][1] = p2[y1 * W + x0];
            val[q][2] = p2[y0 * W + x1];
            val[q][3] = p2[y1 * W + x1];
            fx[q] = x - x0;
            fy[q] = y - y0;
        }
#pragma unroll
        for (int q = 0; q < kWsBatch; q++) {
            const int c = tid + (q0 + q) * BX * BY;
            if (c >= kCells) continue;
            const int r = c / (BX + 4), cc = c - r * (BX + 4);
            double res;
            if (outside[q]) {
                res = p1[off[q]];
            } else {
                const double dx = fx[q], dy = fy[q], ex = 1.0 - dx, ey = 1.0 - dy;
                res = 0.0;
                res += val[q][0] * (ex * ey);
                res += val[q][1] * (ex * dy);
                res += val[q][2] * (dx * ey);
                res += val[q][3] * (dx * dy);
            }
            raw[r][cc] = res;
            if (exhaustive) {
                const double d = fabs(p1[off[q]] - res);
                ex_valid = ex_valid || (d > 0 && d < 1000000);
                ex_hit = ex_hit || (d >= wit_thr && d < 1000000);
            }
        }
    }
    __syncthreads();
    if (exhaustive) {  // (block-uniform)
        const int any_hit = __syncthreads_or(ex_hit ? 1 : 0), any_valid = __syncthreads_or(ex_valid ? 1 : 0);
        if (tid == 0 && (any_hit || !any_valid)) wit[blockIdx.z] = any_hit ? mark : (mark ^ kLapNone);
    }
    if (sampler) {
        const double d = fabs(own1 - raw[wr][wc]);
        const unsigned long long hits = __ballot(d >= wit_thr && d < 1000000);
        if (hits != 0ull && threadIdx.x == 0u) wit[blockIdx.z] = mark;
    }
    for (int r = threadIdx.y; r < kWsRows + 4; r += BY) {
        double acc = 0.0;
#pragma unroll
        for (int l = -2; l <= 2; l++) acc += raw[r][threadIdx.x + 2 + l] * g.t[l + 2];
        hs[r][threadIdx.x] = acc;
    }
    __syncthreads();
    const int j = j0 + threadIdx.x;
    if (j >= W) return;
    for (int r = threadIdx.y; r < kWsRows; r += BY) {
        const int i = i0 + r;
        if (i >= H) break;
        const size_t o = blockIdx.z * np + (size_t)i * W + j;
        double s2 = 0.0;
#pragma unroll
        for (int l = -2; l <= 2; l++) s2 += hs[r + l + 2][threadIdx.x] * g.t[l + 2];
        const double s1 = im1s[o];
        double t = s1;
        t *= 0.4;
        t += s2 * 0.6;
        blend[o] = t;
        imdt[o] = s2 - s1;
        // phi of the flow this outer iteration starts from: k_phi's expressions without an increment, by the blocks of channel
        // 0 -- no launch of its own, and the update kernel is spared the scattered reads of its neighbours' increments
        if (phi_out != nullptr && blockIdx.z == 0) {
            const size_t q = (size_t)i * W + j;
            const double uc = u[q], vc = v[q];
            double ur = 0.0, vr = 0.0, ud = 0.0, vd = 0.0;
            if (j < W - 1) {
                ur = u[q + 1];
                vr = v[q + 1];
            }
            if (i < H - 1) {
                ud = u[q + W];
                vd = v[q + W];
            }
            const double ux = j < W - 1 ? ur - uc : 0.0;
            const double uy = i < H - 1 ? ud - uc : 0.0;
            const double vx = j < W - 1 ? vr - vc : 0.0;
            const double vy = i < H - 1 ? vd - vc : 0.0;
            const double tt = ux * ux + uy * uy + vx * vx + vy * vy;
            phi_out[q] = 0.5 / sqrt(tt + 0.001 * 0.001);
        }
    }
}

// index (in doubles) of cell (i, j) in an SOR operand plane.  Skew mode: paired planes, (phi,xy) (a1,a2) (b1,b2)
// (du,dv) interleaved, cell (i, j) at (i + j + qt) * hp + (i + rt) -- see common.h.
struct SkewIdx {
    int hp, qt, rt;        // coefficient planes
    int nb, npos_d;        // (du, dv) planes: bands, positions
    int dpar;              // which of the two (du, dv) planes holds the last sweep's values
    int band_rows, koff, poff;  // rows per band, climb of the bands up to the last sweep, position offset (common.h)
};
__device__ __forceinline__ size_t skew_cell(int i, int j, const SkewIdx& k) {
    return (size_t)(i + j + k.qt) * k.hp + (size_t)(i + k.rt);
}
template <bool SKEW>
__device__ __forceinline__ size_t sor_index(int i, int j, int W, const SkewIdx& k) {
    if (SKEW) return 2 * skew_cell(i, j, k);
    return (size_t)i * W + j;
}
// cell of (du, dv)(i, j) after the last sweep in the banded ping-pong planes (common.h)
__device__ __forceinline__ size_t dudv_cell(int i, int j, const SkewIdx& k) {
    const int t = i + k.koff, b = t / k.band_rows, c = 1 + (t - b * k.band_rows);
    const size_t parity = (size_t)k.dpar * k.npos_d * k.nb * kLanes;
    return parity + ((size_t)(j + c + k.poff) * k.nb + b) * kLanes + c;
}

// ------------------------------------------------------------------------------------------------
// The flow increment (du, dv) of the previous inner fixed-point iteration, read where the solver left it
// (paired skewed plane or two row-major planes).  `du == nullptr` means the first inner iteration: du = dv = 0.
// ------------------------------------------------------------------------------------------------
struct Increment {
    const double *du, *dv;
    int skew;
    SkewIdx sk;
    const double* gm;  // Gaussian-mixture noise model (src/OpticalFlow.cpp:359-367): alpha[C], sigma[C], beta[C],
                       // sigma_square[C], beta_square[C] of GaussianMixture (src/NoiseModel.h); null = Laplacian (default)
    const double* lap; // Laplacian noise level per channel (LapPara, estLaplacianNoise src/OpticalFlow.cpp:594-639) for the guard of
                       // :399-400 -- psi of a channel stays 0 (Psi_1st.reset(), :333) while LapPara[k] < 1E-20; null = not consulted
                       // (the optimistic pass of flow_device: api.hip, LapGuard)
};
__device__ __forceinline__ void increment_at(const Increment& I, int i, int j, int W, double& du, double& dv) {
    if (I.du == nullptr) {
        du = 0.0;
        dv = 0.0;
    } else if (I.skew) {
        const size_t q = 2 * dudv_cell(i, j, I.sk);
        du = I.du[q];
        dv = I.du[q + 1];
    } else {
        du = I.du[(size_t)i * W + j];
        dv = I.dv[(size_t)i * W + j];
    }
}

// ------------------------------------------------------------------------------------------------
// phi, src/OpticalFlow.cpp:295-331: uu = u (+ du after the first inner iteration, :297-303), forward differences
// (zero in the last column / row, src/Image.h:979-986, :1022-1029), phi = 0.5/sqrt(ux^2+uy^2+vx^2+vy^2+eps).
// ------------------------------------------------------------------------------------------------
__global__ void k_phi(const double* __restrict__ u, const double* __restrict__ v, Increment I,
                      double* __restrict__ phi, int H, int W, Rect rc, unsigned long long* stamp) {
    stamp_now(stamp);
    const int j = rc.x0 + blockIdx.x * BX + threadIdx.x, i = rc.y0 + blockIdx.y * BY + threadIdx.y;
    if (j >= rc.x1 || i >= rc.y1) return;
    const size_t o = (size_t)i * W + j;
    double uc = u[o], vc = v[o], ur = 0.0, vr = 0.0, ud = 0.0, vd = 0.0;
    if (j < W - 1) {
        ur = u[o + 1];
        vr = v[o + 1];
    }
    if (i < H - 1) {
        ud = u[o + W];
        vd = v[o + W];
    }
    if (I.du != nullptr) {  // uu = u + du, vv = v + dv (Image::Add, src/Image.h:1857-1877)
        double a, b;
        increment_at(I, i, j, W, a, b);
        uc = uc + a;
        vc = vc + b;
        if (j < W - 1) {
            increment_at(I, i, j + 1, W, a, b);
            ur = ur + a;
            vr = vr + b;
        }
        if (i < H - 1) {
            increment_at(I, i + 1, j, W, a, b);
            ud = ud + a;
            vd = vd + b;
        }
    }
    const double ux = j < W - 1 ? ur - uc : 0.0;
    const double uy = i < H - 1 ? ud - uc : 0.0;
    const double vx = j < W - 1 ? vr - vc : 0.0;
    const double vy = i < H - 1 ? vd - vc : 0.0;
    const double t = ux * ux + uy * uy + vx * vx + vy * vy;
    phi[o] = 0.5 / sqrt(t + 0.001 * 0.001);
}

// OpticalFlow::Laplacian at one cell, src/OpticalFlow.cpp:641-690: column W-1 receives no horizontal
// term and row H-1 no vertical term (the loops stop at W-2 / H-2).
__device__ __forceinline__ double laplacian_at(const double* __restrict__ in, const double* __restrict__ wt, int i,
                                               int j, int H, int W) {
    const size_t o = (size_t)i * W + j;
    const double c = in[o];
    double out = 0.0;
    if (j < W - 1) {
        out -= (in[o + 1] - c) * wt[o];
        if (j > 0) out += (c - in[o - 1]) * wt[o - 1];
    }
    if (i < H - 1) {
        out -= (in[o + W] - c) * wt[o];
        if (i > 0) out += (c - in[o - W]) * wt[o - W];
    }
    return out;
}

__global__ void k_laplacian(const double* __restrict__ in, const double* __restrict__ wt, double* __restrict__ out,
                            int H, int W) {
    const int j = blockIdx.x * BX + threadIdx.x, i = blockIdx.y * BY + threadIdx.y;
    if (j >= W || i >= H) return;
    out[(size_t)i * W + j] = laplacian_at(in, wt, i, j, H, W);
}

// diagonal terms of the SOR update, src/OpticalFlow.cpp:468-501: coeff accumulated left, right, up, down,
// scaled by alpha; a = omega / (imd?2 + alpha*0.05 + coeff).  They do not change during the sweeps, so
// the two divisions per cell-update of the reference are hoisted out of the sweep loop (same bits:
// the reference evaluates (omega/denominator) * (rhs - sigma) left to right).
__device__ __forceinline__ void sor_diagonals(const double* __restrict__ phi, int i, int j, int H, int W,
                                              double imdx2, double imdy2, double alpha, double omega, double& a1,
                                              double& a2) {
    const size_t o = (size_t)i * W + j;
    double coeff = 0.0;
    if (j > 0) coeff += phi[o - 1];
    if (j < W - 1) coeff += phi[o];
    if (i > 0) coeff += phi[o - W];
    if (i < H - 1) coeff += phi[o];
    coeff *= alpha;
    a1 = omega / (imdx2 + alpha * 0.05 + coeff);
    a2 = omega / (imdy2 + alpha * 0.05 + coeff);
}

// ------------------------------------------------------------------------------------------------
// Fused assembly of the linear system of one fixed-point iteration (du = dv = 0, i.e. nInner == 1):
//   imdx/imdy  = 5-point derivatives of the blended image, src/OpticalFlow.cpp:95-96 (computed inline)
//   psi        = 1/(2 sqrt(imdt^2 + eps)) per channel, :377-406
//   imdxy,...  = channel means of (psi*a)*b, :414-427 with src/Image.h:1747-1763 and :1537-1545
//   rhs        = -imdtd? - alpha*Laplacian(flow, phi), :437-448
//   a1, a2     = hoisted SOR diagonals (above)
// Reads 2*planes + 3 streams, writes 6 SOR operand planes in the layout the solver wants.
// ------------------------------------------------------------------------------------------------
struct SystemCell {
    double phi, xy, a1, a2, b1, b2, x2, y2;
};

// EDGE = false: the cell is at least 2 pixels away from every image border, so no index needs clamping and every
// neighbour exists -- the same operations in the same order with constant address offsets (the kernel is bound by
// instruction issue, a third of it index arithmetic for the clamps).
// PLANES > 0: the channel count as a compile-time constant -- the channel loop is unrolled, so the stencil loads of all
// channels are in flight together instead of one channel's per memory round trip (same operations, same order).
// FAST: the default branches, known at compile time -- Laplacian noise model without a consulted estimate (I.gm == I.lap ==
// nullptr) and no increment of an earlier inner iteration (I.du == nullptr): no branch is left inside the channel loop, so
// the stencil loads of all channels really are in flight together (the uniform branches cut the loop into basic blocks, each
// with its own wait: five memory round trips per cell).
template <bool EDGE = true, int PLANES = 0, bool FAST = false>
__device__ __forceinline__ SystemCell assemble_cell(const double* __restrict__ blend, const double* __restrict__ imdt,
                                                    const double* __restrict__ phi, const double* __restrict__ u,
                                                    const double* __restrict__ v, int i, int j, int H, int W,
                                                    int planes_rt, double alpha, double omega, const Taps& d,
                                                    const Increment& I) {
    const int planes = PLANES > 0 ? PLANES : planes_rt;
    const size_t np = (size_t)H * W, o = (size_t)i * W + j;
    double sxy = 0.0, sx2 = 0.0, sy2 = 0.0, stx = 0.0, sty = 0.0;
    double du = 0.0, dv = 0.0;
    if (!FAST) increment_at(I, i, j, W, du, dv);
    const auto channel = [&](int k) {
        const double* im = blend + k * np;
        double gx = 0.0, gy = 0.0;
        if (EDGE) {
#pragma unroll
            for (int l = -2; l <= 2; l++) gx += im[(size_t)i * W + clampi(j + l, W)] * d.t[l + 2];
#pragma unroll
            for (int l = -2; l <= 2; l++) gy += im[(size_t)clampi(i + l, H) * W + j] * d.t[l + 2];
        } else {
            const double* p = im + o;
#pragma unroll
            for (int l = -2; l <= 2; l++) gx += p[l] * d.t[l + 2];
#pragma unroll
            for (int l = -2; l <= 2; l++) gy += p[(ptrdiff_t)l * W] * d.t[l + 2];
        }
        const double gt = imdt[k * np + o];
        double t = gt;  // imdt + imdx*du + imdy*dv (src/OpticalFlow.cpp:384); du = dv = 0 in the first inner iteration
        if (!FAST && I.du != nullptr) t = gt + gx * du + gy * dv;
        t *= t;
        double psi;
        if (FAST) {
            psi = 1 / (2 * sqrt(t + 0.001 * 0.001));
        } else if (I.gm != nullptr) {  // :392-396 with GaussianMixture::Gaussian (src/NoiseModel.h:120-126).  The reference's PI there is
                                // 3.1415927: Stochastic.h:19 defines it before NoiseModel.h's #ifndef (oracle/papof_oracle.c).
                                // exp() is the device library's (<= 1 ulp, not glibc's bits): tolerance-checked branch.
            const double* g = I.gm;
            const double alpha_k = g[k], sigma = g[planes + k], beta = g[2 * planes + k];
            const double s2 = g[3 * planes + k], b2 = g[4 * planes + k];
            const double prob1 = exp(-t / (2 * s2)) / (2 * 3.1415927 * sigma) * alpha_k;
            const double prob2 = exp(-t / (2 * b2)) / (2 * 3.1415927 * beta) * (1 - alpha_k);
            const double prob11 = prob1 / (2 * s2);
            const double prob22 = prob2 / (2 * b2);
            psi = (prob11 + prob22) / (prob1 + prob2);
        } else if (I.lap != nullptr && I.lap[k] < 1E-20) {
            psi = 0.0;  // :399-400 `continue` on the freshly reset Psi_1st (:333)
        } else {
            psi = 1 / (2 * sqrt(t + 0.001 * 0.001));
        }
        const double pgx = psi * gx, pgy = psi * gy;
        if (planes == 1) {
            sxy = pgx * gy;
            sx2 = pgx * gx;
            sy2 = pgy * gy;
            stx = pgx * gt;
            sty = pgy * gt;
        } else {
            sxy += pgx * gy;
            sx2 += pgx * gx;
            sy2 += pgy * gy;
            stx += pgx * gt;
            sty += pgy * gt;
        }
    };
    if (PLANES > 0) {
#pragma unroll
        for (int k = 0; k < PLANES; k++) channel(k);
    } else {
#pragma unroll 1
        for (int k = 0; k < planes; k++) channel(k);
    }
    if (planes > 1) {
        sxy = sxy / planes;
        sx2 = sx2 / planes;
        sy2 = sy2 / planes;
        stx = stx / planes;
        sty = sty / planes;
    }
    SystemCell c;
    if (EDGE) {
        c.b1 = -stx - alpha * laplacian_at(u, phi, i, j, H, W);
        c.b2 = -sty - alpha * laplacian_at(v, phi, i, j, H, W);
        sor_diagonals(phi, i, j, H, W, sx2, sy2, alpha, omega, c.a1, c.a2);
    } else {  // laplacian_at / sor_diagonals with every neighbour present
        const double pc = phi[o], pl = phi[o - 1], pu = phi[o - W];
        const double uc = u[o], vc = v[o];
        double lu = 0.0, lv = 0.0;
        lu -= (u[o + 1] - uc) * pc;
        lu += (uc - u[o - 1]) * pl;
        lu -= (u[o + W] - uc) * pc;
        lu += (uc - u[o - W]) * pu;
        lv -= (v[o + 1] - vc) * pc;
        lv += (vc - v[o - 1]) * pl;
        lv -= (v[o + W] - vc) * pc;
        lv += (vc - v[o - W]) * pu;
        c.b1 = -stx - alpha * lu;
        c.b2 = -sty - alpha * lv;
        double coeff = 0.0;
        coeff += pl;
        coeff += pc;
        coeff += pu;
        coeff += pc;
        coeff *= alpha;
        c.a1 = omega / (sx2 + alpha * 0.05 + coeff);
        c.a2 = omega / (sy2 + alpha * 0.05 + coeff);
    }
    c.phi = phi[o];
    c.xy = sxy;
    c.x2 = sx2;
    c.y2 = sy2;
    return c;
}

// row-major operands (red-black / Jacobi modes): one thread per cell
__global__ void k_assemble(const double* __restrict__ blend, const double* __restrict__ imdt,
                           const double* __restrict__ phi, const double* __restrict__ u,
                           const double* __restrict__ v, int H, int W, int planes, double alpha, double omega,
                           double* __restrict__ o_phi, double* __restrict__ o_xy, double* __restrict__ o_a1,
                           double* __restrict__ o_a2, double* __restrict__ o_b1, double* __restrict__ o_b2,
                           double* __restrict__ o_x2, double* __restrict__ o_y2, Taps d, Increment I, Rect rc,
                           unsigned long long* stamp) {
    stamp_now(stamp);
    const int j = rc.x0 + blockIdx.x * BX + threadIdx.x, i = rc.y0 + blockIdx.y * BY + threadIdx.y;
    if (j >= rc.x1 || i >= rc.y1) return;
    const size_t o = (size_t)i * W + j;
    const SystemCell c = assemble_cell(blend, imdt, phi, u, v, i, j, H, W, planes, alpha, omega, d, I);
    o_phi[o] = c.phi;
    o_xy[o] = c.xy;
    o_a1[o] = c.a1;
    o_a2[o] = c.a2;
    o_b1[o] = c.b1;
    o_b2[o] = c.b2;
    if (o_x2) o_x2[o] = c.x2;
    if (o_y2) o_y2[o] = c.y2;
}

// Skewed, paired operands of the exact-order solver.  A block owns a tile of 62 rows x kTileJ columns: cells are
// COMPUTED in row-major order (coalesced plane reads), staged in LDS, and WRITTEN in skew order -- for one skew
// position (anti-diagonal) the 16 threads of a group store 16 neighbouring rows = 256 contiguous bytes per paired
// plane -- instead of scattered 16-byte cells.
#ifndef PAPOF_V_ASMCOLS
#define PAPOF_V_ASMCOLS 16
#endif
constexpr int kTileJ = PAPOF_V_ASMCOLS;  // columns of the assembly kernel's tile = cells of one contiguous chunk it writes
#ifndef PAPOF_V_ASMROWS
#define PAPOF_V_ASMROWS 16  // same-box A/B (round 3, ms per 1080p pair): 62 rows 10.93-10.96, 31: 10.80, 24: 10.91, 16: 10.73-10.83, 12: 10.76-10.78, 8: 10.85-10.87
#endif
constexpr int kAsmRows = PAPOF_V_ASMROWS;  // rows of the assembly kernel's tile (its LDS stage bounds the workgroups per CU)
struct double2s {
    double x, y;
};
template <int PLANES, bool FAST>
__global__ __launch_bounds__(256) void k_assemble_skew(const double* __restrict__ blend,
                                                       const double* __restrict__ imdt,
                                                       const double* __restrict__ phi, const double* __restrict__ u,
                                                       const double* __restrict__ v, int H, int W, int planes,
                                                       double alpha, double omega, SkewIdx sk,
                                                       double2s* __restrict__ pa, double2s* __restrict__ pb,
                                                       double2s* __restrict__ pc, double* __restrict__ o_x2,
                                                       double* __restrict__ o_y2, Taps d, Increment I,
                                                       unsigned long long* stamp, int row0, int row1) {
    __shared__ double stage[6][kAsmRows][kTileJ + 1];
    stamp_now(stamp);
    // tiles start at row0 and rows row0 .. row1-1 are written (a strip of the plane; the whole plane otherwise)
    const int ib = row0 + blockIdx.y * kAsmRows, j0 = blockIdx.x * kTileJ, tid = threadIdx.x;
    // block-uniform: no cell of this tile is closer than 2 pixels to an image border
    const bool interior = ib >= 2 && ib + kAsmRows + 2 <= H && j0 >= 2 && j0 + kTileJ + 2 <= W;
    for (int c = tid; c < kAsmRows * kTileJ; c += 256) {
        const int r = c / kTileJ, jj = c - r * kTileJ;
        const int i = ib + r, j = j0 + jj;
        if (i < row1 && j < W) {
            const SystemCell s =
                interior ? assemble_cell<false, PLANES, FAST>(blend, imdt, phi, u, v, i, j, H, W, planes, alpha, omega, d, I)
                         : assemble_cell<true, PLANES, FAST>(blend, imdt, phi, u, v, i, j, H, W, planes, alpha, omega, d, I);
            stage[0][r][jj] = s.phi;
            stage[1][r][jj] = s.xy;
            stage[2][r][jj] = s.a1;
            stage[3][r][jj] = s.a2;
            stage[4][r][jj] = s.b1;
            stage[5][r][jj] = s.b2;
            if (o_x2) o_x2[(size_t)i * W + j] = s.x2;
            if (o_y2) o_y2[(size_t)i * W + j] = s.y2;
        }
    }
    __syncthreads();
    const int g = tid / kTileJ, jj = tid - g * kTileJ;  // 16 groups of 16 threads; a group walks anti-diagonals
    const int j = j0 + jj;
    if (j >= W) return;
    for (int pp = g; pp <= kAsmRows + kTileJ - 2; pp += 256 / kTileJ) {  // pp = jj + (row in tile)
        const int r = pp - jj;
        if (r < 0 || r >= kAsmRows) continue;
        const int i = ib + r;
        if (i >= row1) continue;
        const size_t q = skew_cell(i, j, sk);
        pa[q] = double2s{stage[0][r][jj], stage[1][r][jj]};
        pb[q] = double2s{stage[2][r][jj], stage[3][r][jj]};
        pc[q] = double2s{stage[4][r][jj], stage[5][r][jj]};
    }
}

// ------------------------------------------------------------------------------------------------
// k_flow_system: k_warp_smooth_blend + k_assemble_skew<PLANES, FAST> in ONE kernel (default branches, exact-order layout) -- the
// warped, smoothed and blended tile goes to the derivative / psi / coefficient stage through LDS, so `blend` and `imdt` (166 MB
// written and read again per level-0 iteration) never go to HBM.  A block owns kFT x kFT cells and works, channel by channel,
// on tiles indexed by PIXEL (position - origin; only pixels inside the image are computed, a consumer looks its neighbour
// clamp(p + l) up): the warp on the tile + 4, its smoothing on + 2 (which is where the blend and imdt live), the derivatives on
// the tile.  phi of the tile (and its upper / left neighbours) is computed from (u, v) into LDS.  Every expression and its
// order are those of the two kernels it replaces: same bits.
// ------------------------------------------------------------------------------------------------
constexpr int kFT = 16, kFH = 4, kFW = kFT + 2 * kFH;  // tile rows (and columns at TX = 16), halo, tile rows with halo
#ifndef PAPOF_V_FS_TX
#define PAPOF_V_FS_TX 16
#endif
constexpr int kFlowSystemTX = PAPOF_V_FS_TX;  // default tile width of k_flow_system (PAPOF_FS_TX overrides per process)
// [163 registers: three workgroups per CU; forcing four or five waves per SIMD spills: 11.2 / 12.8 ms per 1080p pair against 10.35]
// SKEW: the operands go to the exact-order solver's paired, skewed planes (pa, pb, pc); otherwise to six row-major planes
// (q[0..5] = phi, xy, a1, a2, b1, b2: the one-workgroup solver of the small levels, k_sor_tiny).
struct SixPlanes {
    double* q[6];
};
template <int PLANES, bool SKEW, bool BATCH = false, int TX = 16>  // TX: columns of the tile (16 rows): 16 or 32; TX * 16 threads
__global__ __launch_bounds__(TX * 16, TX == 32 ? 4 : 1) void k_flow_system(const double* __restrict__ im1, const double* __restrict__ im2,
                                                     const double* __restrict__ u, const double* __restrict__ v,
                                                     const double* __restrict__ im1s, int H, int W, double alpha,
                                                     double omega, SkewIdx sk, SixPlanes out, Taps g, Taps d,
                                                     unsigned long long* stamp, unsigned* __restrict__ wit, double wit_thr,
                                                     unsigned mark, int row0, int row1, BatchK bk) {
    if (BATCH) {  // blockIdx.y: the pair of a batch (common.h: BatchK)
        const size_t p = blockIdx.y;
        im1 += p * bk.im;
        im2 += p * bk.im;
        im1s += p * bk.im;
        u += p * bk.uv;
        v += p * bk.uv;
        if (wit != nullptr) wit += p * bk.wit;
#pragma unroll
        for (int q = 0; q < 6; q++) out.q[q] += p * bk.sp;
    }
    // rows row0 .. row1-1 are assembled (a rank's range of rows, tiles.hip: bands_flow; the whole plane otherwise): (u, v) are
    // read on rows row0 - 4 .. row1 + 3 only, the smoothed frame 1 on row0 - 2 .. row1 + 1
    constexpr int NT = TX * kFT, HWc = TX + 2 * kFH;  // threads = cells of the tile; columns of the tile with its halo
    __shared__ double raw[kFW][HWc + 1];            // warped frame 2, pixels (ib - 4 .. ib + 19) x (j0 - 4 .. j0 + 19)
    __shared__ double hs[kFW][TX + 4 + 1];         // h-smoothed, columns j0 - 2 .. j0 + 17
    __shared__ double bl[kFT + 4][TX + 4 + 1];     // blend, pixels (ib - 2 .. ib + 17) x (j0 - 2 .. j0 + 17)
    __shared__ double it[kFT][TX + 1];             // imdt of the tile
    __shared__ double ph[kFT + 1][TX + 1 + 1];     // phi, pixels (ib - 1 .. ib + 15) x (j0 - 1 .. j0 + 15)
    __shared__ double stage[6][kFT][TX + 1];
    stamp_now(stamp);
    // Workgroups are dealt round-robin over the 8 XCDs (observed; speed only): give each XCD a contiguous run of tiles, so
    // that the halo pixels neighbouring tiles both gather are found in that XCD's L2 (as k_sor_blocked does)
    const int ntx = (W + TX - 1) / TX, ntiles = ntx * ((row1 - row0 + kFT - 1) / kFT), per = (ntiles + 7) >> 3;
    const int vt = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || vt >= ntiles) return;  // whole workgroup, before any barrier
    const int tby = vt / ntx, tbx = vt - tby * ntx;
    const int ib = row0 + tby * kFT, j0 = tbx * TX, tid = threadIdx.x;
    const size_t np = (size_t)H * W;
    const int vy0 = max(0, row0 - kFH), vy1 = min(H, row1 + kFH);  // rows on which the operands are valid
    const bool interior = ib >= vy0 + kFH && ib + kFT <= row1 && ib + kFT + kFH <= vy1 && j0 >= kFH && j0 + TX + kFH <= W;  // block-uniform
    // ---- bilinear taps of the tile's pixels, once for all channels: (x0, y0), (dx, dy), or "outside" (frame 1's value)
    constexpr int kPer = (kFW * HWc + NT - 1) / NT;  // 3 (TX = 16: 576 pixels, NT = 16 x 16), 2 (TX = 32: 960, NT = 32 x 16)
    int t_o[kPer], t_x0[kPer], t_y0[kPer];
    double t_dx[kPer], t_dy[kPer];
    bool t_in[kPer], t_out[kPer];
#pragma unroll
    for (int q = 0; q < kPer; q++) {
        const int c = tid + q * NT, r = c / HWc, cc = c - r * HWc;
        const int i = ib - kFH + r, j = j0 - kFH + cc;
        t_in[q] = c < kFW * HWc && i >= vy0 && i < vy1 && j >= 0 && j < W;
        t_o[q] = t_in[q] ? i * W + j : 0;
        double x = 0.0, y = 0.0;  // (a cell outside the image: any valid position, its value is not used)
        t_out[q] = false;
        if (t_in[q]) {
            y = i + v[t_o[q]];
            x = j + u[t_o[q]];
            t_out[q] = x < 0 || x > W - 1 || y < 0 || y > H - 1;  // warp_value()
            if (t_out[q]) {
                x = j;
                y = i;
            }
        }
        t_x0[q] = (int)x;
        t_y0[q] = (int)y;
        t_dx[q] = x - t_x0[q];
        t_dy[q] = y - t_y0[q];
    }
    // ---- phi of the tile and its upper / left neighbours (k_phi without an increment), and the flow of the own cell
    for (int c = tid; c < (kFT + 1) * (TX + 1); c += NT) {
        const int r = c / (TX + 1), cc = c - r * (TX + 1);
        const int i = ib - 1 + r, j = j0 - 1 + cc;
        if (i < 0 || i >= min(H, row1) || j < 0 || j >= W) continue;
        const size_t o = (size_t)i * W + j;
        const double uc = u[o], vc = v[o];
        double ur = 0.0, vr = 0.0, ud = 0.0, vd = 0.0;
        if (j < W - 1) {
            ur = u[o + 1];
            vr = v[o + 1];
        }
        if (i < H - 1) {
            ud = u[o + W];
            vd = v[o + W];
        }
        const double ux = j < W - 1 ? ur - uc : 0.0;
        const double uy = i < H - 1 ? ud - uc : 0.0;
        const double vx = j < W - 1 ? vr - vc : 0.0;
        const double vy = i < H - 1 ? vd - vc : 0.0;
        const double tt = ux * ux + uy * uy + vx * vx + vy * vy;
        ph[r][cc] = 0.5 / sqrt(tt + 0.001 * 0.001);
    }
    const int orow = tid / TX, ocol = tid - orow * TX;  // this thread's cell of the tile
    const int oi = ib + orow, oj = j0 + ocol;
    const bool own = oi < row1 && oj < W;
    double sxy = 0.0, sx2 = 0.0, sy2 = 0.0, stx = 0.0, sty = 0.0;
    // witnesses of the Laplacian-noise guard (k_warp_smooth_blend): one block in sixteen on large grids, its first wave
    const bool sampler = wit != nullptr && tid < 64 &&
                         (ntiles <= 256 || ((tbx + 5 * tby) & 15) == 0);
    // The global operands of a channel are PREFETCHED while the previous channel is worked on (a workgroup barrier does not
    // wait for loads in flight): the gathers (or frame 1's value for a pixel that leaves the image) right behind P1, the smoothed
    // frame 1 of the blend's pixels right behind P3 -- the phases between are LDS only.
    constexpr int kP3 = ((kFT + 4) * (TX + 4) + NT - 1) / NT;  // 2
    double gv[kPer][4], s1v[kP3], wv = 0.0;
    const auto load_gathers = [&](int k) {
        const double *p1 = im1 + k * np, *p2 = im2 + k * np;
#pragma unroll
        for (int q = 0; q < kPer; q++) {  // branch-free: a pixel that leaves the image gathers around ITSELF in frame 1 (its taps were
            // set to its own position: tap 0 is frame 1's value), a cell outside the image reads pixel 0 and is not used
            const double* src = t_out[q] ? p1 : p2;
            const int x0 = t_x0[q], y0 = t_y0[q], x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
            gv[q][0] = src[y0 * W + x0];
            gv[q][1] = src[y1 * W + x0];
            gv[q][2] = src[y0 * W + x1];
            gv[q][3] = src[y1 * W + x1];
        }
        if (sampler) wv = p1[t_o[0]];
    };
    const auto load_s1 = [&](int k) {
        const double* ps = im1s + k * np;
#pragma unroll
        for (int q = 0; q < kP3; q++) {  // (branch-free: a position outside the image reads its clamped pixel and is not used)
            const int c = min(tid + q * NT, (kFT + 4) * (TX + 4) - 1), r = c / (TX + 4), cc = c - r * (TX + 4);
            s1v[q] = ps[(size_t)clampi(ib - 2 + r, H) * W + clampi(j0 - 2 + cc, W)];
        }
    };
    load_gathers(0);
    load_s1(0);
#pragma unroll 1
    for (int k = 0; k < PLANES; k++) {
        // -- P1: the warp
        bool hit = false;
#pragma unroll
        for (int q = 0; q < kPer; q++) {
            const int c = tid + q * NT, r = c / HWc, cc = c - r * HWc;
            const double dx = t_dx[q], dy = t_dy[q], ex = 1.0 - dx, ey = 1.0 - dy;
            double res = 0.0;
            res += gv[q][0] * (ex * ey);
            res += gv[q][1] * (ex * dy);
            res += gv[q][2] * (dx * ey);
            res += gv[q][3] * (dx * dy);
            res = t_out[q] ? gv[q][0] : res;  // warp_value(): a pixel that leaves the image takes frame 1's value
            if (t_in[q]) raw[r][cc] = res;
            if (sampler && q == 0) {
                const double dd = fabs(wv - res);
                hit = t_in[q] && dd >= wit_thr && dd < 1000000;
            }
        }
        if (sampler) {
            const unsigned long long hits = __ballot(hit);
            if (hits != 0ull && tid == 0) wit[k] = mark;
        }
        if (k + 1 < PLANES) load_gathers(k + 1);
        __syncthreads();
        // -- P2: horizontal pass on every row of the tile, columns j0 - 2 .. j0 + 17
        if (interior) {  // (block-uniform: every pixel of the tile + 4 is inside the image -- no clamps, no tests)
            for (int c = tid; c < kFW * (TX + 4); c += NT) {
                const int r = c / (TX + 4), cc = c - r * (TX + 4);
                double acc = 0.0;
#pragma unroll
                for (int l = -2; l <= 2; l++) acc += raw[r][cc + 2 + l] * g.t[l + 2];
                hs[r][cc] = acc;
            }
        } else {
            for (int c = tid; c < kFW * (TX + 4); c += NT) {
                const int r = c / (TX + 4), cc = c - r * (TX + 4);
                const int i = ib - kFH + r, j = j0 - 2 + cc;
                if (i < 0 || i >= H || j < 0 || j >= W) continue;
                double acc = 0.0;
#pragma unroll
                for (int l = -2; l <= 2; l++) acc += raw[r][clampi(j + l, W) - (j0 - kFH)] * g.t[l + 2];
                hs[r][cc] = acc;
            }
        }
        __syncthreads();
        // -- P3: vertical pass, blend and imdt on (ib - 2 .. ib + 17) x (j0 - 2 .. j0 + 17)
#pragma unroll
        for (int q = 0; q < kP3; q++) {
            const int c = tid + q * NT, r = c / (TX + 4), cc = c - r * (TX + 4);
            const int i = ib - 2 + r, j = j0 - 2 + cc;
            if (c >= (kFT + 4) * (TX + 4) || i < 0 || i >= H || j < 0 || j >= W) continue;
            double s2 = 0.0;
            if (interior) {
#pragma unroll
                for (int l = -2; l <= 2; l++) s2 += hs[r + 2 + l][cc] * g.t[l + 2];
            } else {
#pragma unroll
                for (int l = -2; l <= 2; l++) s2 += hs[clampi(i + l, H) - (ib - kFH)][cc] * g.t[l + 2];
            }
            const double s1 = s1v[q];
            double t = s1;
            t *= 0.4;
            t += s2 * 0.6;
            bl[r][cc] = t;
            if (r >= 2 && r < kFT + 2 && cc >= 2 && cc < TX + 2) it[r - 2][cc - 2] = s2 - s1;
        }
        if (k + 1 < PLANES) load_s1(k + 1);
        __syncthreads();
        // -- P4: derivatives, psi and the channel's terms of the own cell (assemble_cell, FAST)
        if (own) {
            double gx = 0.0, gy = 0.0;
            if (interior) {
#pragma unroll
                for (int l = -2; l <= 2; l++) gx += bl[orow + 2][ocol + 2 + l] * d.t[l + 2];
#pragma unroll
                for (int l = -2; l <= 2; l++) gy += bl[orow + 2 + l][ocol + 2] * d.t[l + 2];
            } else {
#pragma unroll
                for (int l = -2; l <= 2; l++) gx += bl[orow + 2][clampi(oj + l, W) - (j0 - 2)] * d.t[l + 2];
#pragma unroll
                for (int l = -2; l <= 2; l++) gy += bl[clampi(oi + l, H) - (ib - 2)][ocol + 2] * d.t[l + 2];
            }
            const double gt = it[orow][ocol];
            double t = gt;
            t *= t;
            const double psi = 1 / (2 * sqrt(t + 0.001 * 0.001));
            const double pgx = psi * gx, pgy = psi * gy;
            if (PLANES == 1) {
                sxy = pgx * gy;
                sx2 = pgx * gx;
                sy2 = pgy * gy;
                stx = pgx * gt;
                sty = pgy * gt;
            } else {
                sxy += pgx * gy;
                sx2 += pgx * gx;
                sy2 += pgy * gy;
                stx += pgx * gt;
                sty += pgy * gt;
            }
        }
        // (the next channel's P1 writes raw, last read in P2; its P3 writes bl / it, read above: two barriers lie between)
    }
    if (own) {
        if (PLANES > 1) {
            sxy = sxy / PLANES;
            sx2 = sx2 / PLANES;
            sy2 = sy2 / PLANES;
            stx = stx / PLANES;
            sty = sty / PLANES;
        }
        const int i = oi, j = oj;
        const size_t o = (size_t)i * W + j;
        const double pc_ = ph[orow + 1][ocol + 1];
        const double pl = j > 0 ? ph[orow + 1][ocol] : 0.0, pu = i > 0 ? ph[orow][ocol + 1] : 0.0;
        // laplacian_at(u, phi) / laplacian_at(v, phi) and sor_diagonals(phi) with phi from LDS
        const double uc = u[o], vc = v[o];
        double lu = 0.0, lv = 0.0;
        if (j < W - 1) {
            lu -= (u[o + 1] - uc) * pc_;
            if (j > 0) lu += (uc - u[o - 1]) * pl;
        }
        if (i < H - 1) {
            lu -= (u[o + W] - uc) * pc_;
            if (i > 0) lu += (uc - u[o - W]) * pu;
        }
        if (j < W - 1) {
            lv -= (v[o + 1] - vc) * pc_;
            if (j > 0) lv += (vc - v[o - 1]) * pl;
        }
        if (i < H - 1) {
            lv -= (v[o + W] - vc) * pc_;
            if (i > 0) lv += (vc - v[o - W]) * pu;
        }
        double coeff = 0.0;
        if (j > 0) coeff += pl;
        if (j < W - 1) coeff += pc_;
        if (i > 0) coeff += pu;
        if (i < H - 1) coeff += pc_;
        coeff *= alpha;
        const double a1 = omega / (sx2 + alpha * 0.05 + coeff), a2 = omega / (sy2 + alpha * 0.05 + coeff);
        const double b1 = -stx - alpha * lu, b2 = -sty - alpha * lv;
        if (SKEW) {
            stage[0][orow][ocol] = pc_;
            stage[1][orow][ocol] = sxy;
            stage[2][orow][ocol] = a1;
            stage[3][orow][ocol] = a2;
            stage[4][orow][ocol] = b1;
            stage[5][orow][ocol] = b2;
        } else {
            out.q[0][o] = pc_;
            out.q[1][o] = sxy;
            out.q[2][o] = a1;
            out.q[3][o] = a2;
            out.q[4][o] = b1;
            out.q[5][o] = b2;
        }
    }
    if (!SKEW) return;
    double2s* const pa = reinterpret_cast<double2s*>(out.q[0]);
    double2s* const pb = reinterpret_cast<double2s*>(out.q[2]);
    double2s* const pc = reinterpret_cast<double2s*>(out.q[4]);
    __syncthreads();
    // ---- written in skew order, as k_assemble_skew does: a group of 16 threads stores 16 neighbouring rows of one position
    // (a tile of TX columns is TX / 16 sub-tiles of 16 x 16 cells: 16 groups of 16 threads each, as for TX = 16)
    const int g16 = tid / kFT, jj = tid - g16 * kFT, sub = g16 % (TX / kFT), grp = g16 / (TX / kFT);
    const int col = sub * kFT + jj, j = j0 + col;
    if (j >= W) return;
    for (int pp = grp; pp <= 2 * kFT - 2; pp += kFT) {
        const int r = pp - jj;
        if (r < 0 || r >= kFT) continue;
        const int i = ib + r;
        if (i >= row1) continue;
        const size_t q = skew_cell(i, j, sk);
        pa[q] = double2s{stage[0][r][col], stage[1][r][col]};
        pb[q] = double2s{stage[2][r][col], stage[3][r][col]};
        pc[q] = double2s{stage[4][r][col], stage[5][r][col]};
    }
}

// stage helper: SOR operands from already assembled row-major planes (tests / micro-benchmark)
template <bool SKEW>
__global__ void k_sor_prep(const double* __restrict__ phi, const double* __restrict__ imdxy,
                           const double* __restrict__ imdx2, const double* __restrict__ imdy2,
                           const double* __restrict__ rhs1, const double* __restrict__ rhs2, int H, int W,
                           double alpha, double omega, SkewIdx sk, double* __restrict__ o_phi,
                           double* __restrict__ o_xy, double* __restrict__ o_a1, double* __restrict__ o_a2,
                           double* __restrict__ o_b1, double* __restrict__ o_b2) {
    const int j = blockIdx.x * BX + threadIdx.x, i = blockIdx.y * BY + threadIdx.y;
    if (j >= W || i >= H) return;
    const size_t o = (size_t)i * W + j;
    double a1, a2;
    sor_diagonals(phi, i, j, H, W, imdx2[o], imdy2[o], alpha, omega, a1, a2);
    const size_t q = sor_index<SKEW>(i, j, W, sk);
    o_phi[q] = phi[o];
    o_xy[q] = imdxy[o];
    o_a1[q] = a1;
    o_a2[q] = a2;
    o_b1[q] = rhs1[o];
    o_b2[q] = rhs2[o];
}

template <bool SKEW>
__global__ void k_sor_unpack(const double* __restrict__ sdu, const double* __restrict__ sdv,
                             double* __restrict__ du, double* __restrict__ dv, int H, int W, SkewIdx sk) {
    const int j = blockIdx.x * BX + threadIdx.x, i = blockIdx.y * BY + threadIdx.y;
    if (j >= W || i >= H) return;
    const size_t o = (size_t)i * W + j;
    if (SKEW) {
        const size_t q = 2 * dudv_cell(i, j, sk);
        du[o] = sdu[q];
        dv[o] = sdu[q + 1];
    } else {
        du[o] = sdu[o];
        dv[o] = sdv[o];
    }
}

// ------------------------------------------------------------------------------------------------
// Phase 6, src/OpticalFlow.cpp:513-516: u += du, v += dv (Image::Add, src/Image.h:1925-1941) and re-warp
// frame 2 with the updated flow -- fused, the update of a pixel only feeds its own warp.
// ------------------------------------------------------------------------------------------------
__global__ void k_update_warp(const double* __restrict__ sdu, const double* __restrict__ sdv, double* __restrict__ u,
                              double* __restrict__ v, const double* __restrict__ im1,
                              const double* __restrict__ im2, double* __restrict__ warp, int H, int W, int planes, Rect rc,
                              int do_warp, unsigned long long* stamp) {
    stamp_now(stamp);
    const int j = rc.x0 + blockIdx.x * BX + threadIdx.x, i = rc.y0 + blockIdx.y * BY + threadIdx.y;
    if (j >= rc.x1 || i >= rc.y1) return;
    const size_t o = (size_t)i * W + j;
    double fu = u[o], fv = v[o];
    fu += sdu[o];
    fv += sdv[o];
    u[o] = fu;
    v[o] = fv;
    if (do_warp) warp_pixel(im1, im2, warp, fu, fv, i, j, H, W, planes);
}

// One-GPU path of Phase 6 with the NEXT outer iteration's Phase 2 folded in: u_out = u + du, v_out = v + dv go to
// another pair of planes (the neighbours' old values are still read below), the re-warp uses the new flow, and -- when
// phi_out is given -- phi = 0.5 / sqrt(ux^2 + uy^2 + vx^2 + vy^2 + eps) of the NEW flow (src/OpticalFlow.cpp:295-331, the
// expressions of k_phi) is written as well: u_new(i, j+1) = u(i, j+1) + du(i, j+1) is the same addition its own thread
// performs, hence the same bits.  Saves one launch and one pass over (u, v) per outer iteration.
template <bool SKEW, bool BATCH = false>
__global__ void k_update_warp_phi(const double* __restrict__ sdu, const double* __restrict__ sdv, SkewIdx sk,
                                  const double* __restrict__ u, const double* __restrict__ v,
                                  double* __restrict__ u_out, double* __restrict__ v_out,
                                  const double* __restrict__ im1, const double* __restrict__ im2,
                                  double* __restrict__ warp, double* __restrict__ phi_out, int H, int W, int planes,
                                  int do_warp, unsigned long long* stamp, int row0, int row1, unsigned* __restrict__ wit,
                                  double wit_thr, unsigned mark, BatchK bk) {
    stamp_now(stamp);
    if (BATCH) {  // blockIdx.y: the pair of a batch (common.h: BatchK; warp and phi_out are not used there)
        const size_t p = blockIdx.y;
        sdu += p * bk.d;
        sdv += p * bk.d;
        u += p * bk.uv;
        v += p * bk.uv;
        u_out += p * bk.uv;
        v_out += p * bk.uv;
        im1 += p * bk.im;
        im2 += p * bk.im;
        if (wit != nullptr) wit += p * bk.wit;
    }
    int tbx, tby;  // a contiguous run of blocks per XCD (xcd_tile): blocks above each other share the lines of the banded (du, dv)
    if (!xcd_tile((W + BX - 1) / BX, (row1 - row0 + BY - 1) / BY, tbx, tby)) return;
    const int j = tbx * BX + threadIdx.x, i = row0 + tby * BY + threadIdx.y;  // rows row0 .. row1-1
    if (j >= W || i >= row1) return;
    const size_t o = (size_t)i * W + j;
    const auto inc = [&](int ii, int jj, double& a, double& b) {
        if (SKEW) {
            const double2s c = reinterpret_cast<const double2s*>(sdu)[dudv_cell(ii, jj, sk)];
            a = c.x;
            b = c.y;
        } else {
            a = sdu[(size_t)ii * W + jj];
            b = sdv[(size_t)ii * W + jj];
        }
    };
    double a, b;
    inc(i, j, a, b);
    double fu = u[o], fv = v[o];
    fu += a;
    fv += b;
    u_out[o] = fu;
    v_out[o] = fv;
    // WITNESS for the Laplacian-noise guard (see k_warp_smooth_blend, which takes the witnesses wherever another outer
    // iteration follows on the level): behind the LAST update of a level nobody evaluates the warp at the new flow, so here the
    // first `planes` threads of every block evaluate it at their own pixel, one channel each.
    if (wit != nullptr) {
        const int bw = min(W - tbx * BX, BX), bh = min(row1 - (row0 + tby * BY), BY);  // the block's pixels
        const int lin = threadIdx.y * bw + threadIdx.x;
        const size_t np = (size_t)H * W;
        if (np <= 65536) {  // a coarse level: few blocks, so every thread samples (channel = its number within the block mod planes)
            for (int k = bw * bh >= planes ? lin % planes : lin; k < planes; k += bw * bh) {
                const double w = warp_value(im1 + k * np, im2 + k * np, fu, fv, i, j, H, W);
                const double d = fabs(im1[k * np + o] - w);
                if (d >= wit_thr && d < 1000000) wit[k] = mark;
            }
        } else if (((tbx + 5 * tby) & 15) == 0)  // a large grid: one block in sixteen samples
        for (int k = lin; k < planes; k += bw * bh) {  // (one trip of one thread per channel, unless the block has fewer pixels)
            const double w = warp_value(im1 + k * np, im2 + k * np, fu, fv, i, j, H, W);
            const double d = fabs(im1[k * np + o] - w);
            if (d >= wit_thr && d < 1000000) wit[k] = mark;
        }
    }
    if (phi_out != nullptr) {
        double ur = 0.0, vr = 0.0, ud = 0.0, vd = 0.0;
        if (j < W - 1) {
            inc(i, j + 1, a, b);
            ur = u[o + 1];
            vr = v[o + 1];
            ur += a;
            vr += b;
        }
        if (i < H - 1) {
            inc(i + 1, j, a, b);
            ud = u[o + W];
            vd = v[o + W];
            ud += a;
            vd += b;
        }
        const double ux = j < W - 1 ? ur - fu : 0.0;
        const double uy = i < H - 1 ? ud - fu : 0.0;
        const double vx = j < W - 1 ? vr - fv : 0.0;
        const double vy = i < H - 1 ? vd - fv : 0.0;
        const double t = ux * ux + uy * uy + vx * vx + vy * vy;
        phi_out[o] = 0.5 / sqrt(t + 0.001 * 0.001);
    }
    if (do_warp) warp_pixel(im1, im2, warp, fu, fv, i, j, H, W, planes);
}

// Exact-order solver layout: (du, dv) read straight from the paired skewed plane (16 bytes per thread; an LDS-staged
// transposition of the tile was measured slower: the kernel is bound by the 4-tap x C-plane gather of the warp).
__global__ void k_update_warp_skew(const double2s* __restrict__ pd, double* __restrict__ u, double* __restrict__ v,
                                   const double* __restrict__ im1, const double* __restrict__ im2,
                                   double* __restrict__ warp, int H, int W, int planes, SkewIdx sk,
                                   unsigned long long* stamp, int do_warp) {
    stamp_now(stamp);
    const int j = blockIdx.x * BX + threadIdx.x, i = blockIdx.y * BY + threadIdx.y;
    if (j >= W || i >= H) return;
    const size_t o = (size_t)i * W + j;
    const double2s c = pd[dudv_cell(i, j, sk)];
    double fu = u[o], fv = v[o];
    fu += c.x;
    fv += c.y;
    u[o] = fu;
    v[o] = fv;
    if (do_warp) warp_pixel(im1, im2, warp, fu, fv, i, j, H, W, planes);
}

// ------------------------------------------------------------------------------------------------
// Final warp of the ORIGINAL frames: Image::warpImageBicubicRef src/Image.h:2624-2701 with the Hermite
// coefficients of BicubicCoeff :2497-2530, then threshold() :2031-2045.  gx, gy, gxy are the central
// differences {-0.5,0,0.5} of frame 2 (imfilter_h / imfilter_v / imfilter_v(gx), :2590-2594).
// Corners A=(x0,y0) B=(x1,y0) C=(x0,y1) D=(x1,y1); cXY multiplies dx^X dy^Y; term order as in the
// reference expressions.  Output goes straight to the interleaved HWC buffer handed back to the caller.
// ------------------------------------------------------------------------------------------------
template <bool BATCH>  // BATCH: blockIdx.y = the pair of a batch (a separate instantiation: the single call's code is untouched --
                       // offsetting the pointers in the one kernel changed its register allocation and cost 47 us at 1080p)
__global__ void k_bicubic(const double* __restrict__ im1, const double* __restrict__ im2,
                          const double* __restrict__ gx, const double* __restrict__ gy,
                          const double* __restrict__ gxy, const double* __restrict__ vx,
                          const double* __restrict__ vy, double* __restrict__ out, int H, int W, int C, Rect rc,
                          unsigned long long* stamp, int planar_out, int clamp, BatchK bk) {
    if (BATCH) {  // blockIdx.y: the pair of a batch (common.h: BatchK)
        const size_t p = blockIdx.y;
        im1 += p * bk.im;
        im2 += p * bk.im;
        gx += p * bk.im;
        gy += p * bk.im;
        gxy += p * bk.im;
        vx += p * bk.uv;
        vy += p * bk.uv;
        out += p * bk.out;
    }
    // planar_out / clamp: the in-loop use on the feature planes (src/OpticalFlow.cpp:517-521 with threshold(), :816
    // without); the final warp of the originals writes interleaved HWC and always clamps.
    stamp_now(stamp);
    int tbx, tby;  // a contiguous run of blocks per XCD (xcd_tile)
    if (!xcd_tile((rc.x1 - rc.x0 + BX - 1) / BX, (rc.y1 - rc.y0 + BY - 1) / BY, tbx, tby)) return;
    const int j = rc.x0 + tbx * BX + threadIdx.x, i = rc.y0 + tby * BY + threadIdx.y;
    if (j >= rc.x1 || i >= rc.y1) return;
    const size_t np = (size_t)H * W, o = (size_t)i * W + j;
    const size_t ostride = planar_out ? np : 1, obase = planar_out ? o : o * C;
    const double x = j + vx[o];
    const double y = i + vy[o];
    if (x < 0 || x > W - 1 || y < 0 || y > H - 1) {
        for (int k = 0; k < C; k++) {
            double r = im1[k * np + o];
            if (clamp) {
                r = r < 0 ? 0.0 : r;
                r = r > 1 ? 1.0 : r;
            }
            out[obase + k * ostride] = r;
        }
        return;
    }
    int x0 = (int)x, y0 = (int)y;
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = clampi(x0, W);
    x1 = clampi(x1, W);
    y0 = clampi(y0, H);
    y1 = clampi(y1, H);
    const double dx = x - x0, dy = y - y0;
    const double dx2 = dx * dx, dy2 = dy * dy;
    const double dx3 = dx * dx2, dy3 = dy * dy2;
    const size_t a = (size_t)y0 * W + x0, b = (size_t)y0 * W + x1, c = (size_t)y1 * W + x0, d = (size_t)y1 * W + x1;
    for (int k = 0; k < C; k++) {
        const double *f = im2 + k * np, *px = gx + k * np, *py = gy + k * np, *pz = gxy + k * np;
        const double fA = f[a], fB = f[b], fC = f[c], fD = f[d];
        const double xA = px[a], xB = px[b], xC = px[c], xD = px[d];
        const double yA = py[a], yB = py[b], yC = py[c], yD = py[d];
        const double zA = pz[a], zB = pz[b], zC = pz[c], zD = pz[d];
        const double c00 = fA;
        const double c10 = xA;
        const double c20 = -3 * fA + 3 * fB - 2 * xA - xB;
        const double c30 = 2 * fA - 2 * fB + xA + xB;
        const double c01 = yA;
        const double c11 = zA;
        const double c21 = -3 * yA + 3 * yB - 2 * zA - zB;
        const double c31 = 2 * yA - 2 * yB + zA + zB;
        const double c02 = -3 * fA + 3 * fC - 2 * yA - yC;
        const double c12 = -3 * xA + 3 * xC - 2 * zA - zC;
        const double c22 = 9 * fA - 9 * fB - 9 * fC + 9 * fD + 6 * xA + 3 * xB - 6 * xC - 3 * xD + 6 * yA - 6 * yB +
                           3 * yC - 3 * yD + 4 * zA + 2 * zB + 2 * zC + zD;
        const double c32 = -6 * fA + 6 * fB + 6 * fC - 6 * fD - 3 * xA - 3 * xB + 3 * xC + 3 * xD - 4 * yA + 4 * yB -
                           2 * yC + 2 * yD - 2 * zA - 2 * zB - zC - zD;
        const double c03 = 2 * fA - 2 * fC + yA + yC;
        const double c13 = 2 * xA - 2 * xC + zA + zC;
        const double c23 = -6 * fA + 6 * fB + 6 * fC - 6 * fD - 4 * xA - 2 * xB + 4 * xC + 2 * xD - 3 * yA + 3 * yB -
                           3 * yC + 3 * yD - 2 * zA - zB - 2 * zC - zD;
        const double c33 = 4 * fA - 4 * fB - 4 * fC + 4 * fD + 2 * xA + 2 * xB - 2 * xC - 2 * xD + 2 * yA - 2 * yB +
                           2 * yC - 2 * yD + zA + zB + zC + zD;
        double r = c00 + c01 * dy + c02 * dy2 + c03 * dy3 + c10 * dx + c11 * dx * dy + c12 * dx * dy2 +
                   c13 * dx * dy3 + c20 * dx2 + c21 * dx2 * dy + c22 * dx2 * dy2 + c23 * dx2 * dy3 + c30 * dx3 +
                   c31 * dx3 * dy + c32 * dx3 * dy2 + c33 * dx3 * dy3;
        if (clamp) {
            r = r < 0 ? 0.0 : r;
            r = r > 1 ? 1.0 : r;
        }
        out[obase + k * ostride] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// The reference's 16-bit flow encoding, OpticalFlow::SaveOpticalFlow / LoadOpticalFlow (src/OpticalFlow.cpp:963-1015):
// q = (unsigned short)((min(max(f, -200), 200) + 200) * 160), (vx, vy) interleaved per pixel (AssembleFlow,
// src/OpticalFlow.h:70-79); back: f = (double)q / 160 - 200.
// ------------------------------------------------------------------------------------------------
__global__ void k_flow_quantize16(const double* __restrict__ vx, const double* __restrict__ vy,
                                  unsigned short* __restrict__ q, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = vx[i], b = vy[i];
    a = (a < -200) ? -200.0 : a;
    a = (a > 200) ? 200.0 : a;
    b = (b < -200) ? -200.0 : b;
    b = (b > 200) ? 200.0 : b;
    q[i * 2] = (unsigned short)((a + 200) * 160);
    q[i * 2 + 1] = (unsigned short)((b + 200) * 160);
}

__global__ void k_flow_dequantize16(const unsigned short* __restrict__ q, double* __restrict__ vx,
                                    double* __restrict__ vy, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vx[i] = (double)q[i * 2] / 160 - 200;
    vy[i] = (double)q[i * 2 + 1] / 160 - 200;
}

// ------------------------------------------------------------------------------------------------
// Flow visualisation of the reference's caller, generateOutputFlowImageFile (Code/Serial/OpticalFlowCalculation.py:
// 143-162): hue = angle / 2 in degrees, saturation 255, value = magnitude min-max normalised to 0..255, converted
// HSV -> BGR as OpenCV's 8-bit path does (float32: h * 6/180, sector table, round to nearest).  PARITY UNPINNED: cv2 is
// not installed here, and cv2.cartToPolar uses a polynomial atan2 (~0.3 degrees), so a hue may differ by one step
// from an OpenCV run; this kernel uses the exact atan2.
// ------------------------------------------------------------------------------------------------
__global__ void k_flow_mag_minmax(const double* __restrict__ vx, const double* __restrict__ vy, size_t n,
                                  double* __restrict__ partial) {  // partial[2*block] = min, [2*block+1] = max
    __shared__ double smin[256], smax[256];
    double lo = INFINITY, hi = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double m = sqrt(vx[i] * vx[i] + vy[i] * vy[i]);
        lo = m < lo ? m : lo;
        hi = m > hi ? m : hi;
    }
    smin[threadIdx.x] = lo;
    smax[threadIdx.x] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + s]);
            smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = smin[0];
        partial[2 * blockIdx.x + 1] = smax[0];
    }
}

__global__ void k_flow_bgr(const double* __restrict__ vx, const double* __restrict__ vy, size_t n,
                           const double* __restrict__ partial, int nblocks, unsigned char* __restrict__ bgr) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double lo = INFINITY, hi = -INFINITY;
    for (int b = 0; b < nblocks; b++) {
        lo = fmin(lo, partial[2 * b]);
        hi = fmax(hi, partial[2 * b + 1]);
    }
    const double x = vx[i], y = vy[i];
    const double mag = sqrt(x * x + y * y);
    double ang = atan2(y, x);  // cv2.cartToPolar: radians in [0, 2 pi)
    if (ang < 0) ang += 2 * M_PI;
    const unsigned char H = (unsigned char)(int)(ang * 180 / M_PI / 2);
    const double scale = hi > lo ? 255.0 / (hi - lo) : 0.0;  // cv2.normalize(..., 0, 255, NORM_MINMAX)
    const unsigned char V = (unsigned char)(int)((mag - lo) * scale);
    // OpenCV HSV2BGR, 8-bit, S = 255
    const float v = V * (1.f / 255.f);
    float h = H * (6.f / 180.f);
    int sector = (int)floorf(h);
    h -= sector;
    sector = ((sector % 6) + 6) % 6;
    const float tab[4] = {v, 0.f, v * (1.f - h), v * h};
    const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    bgr[i * 3] = (unsigned char)(int)rintf(tab[sd[sector][0]] * 255.f);
    bgr[i * 3 + 1] = (unsigned char)(int)rintf(tab[sd[sector][1]] * 255.f);
    bgr[i * 3 + 2] = (unsigned char)(int)rintf(tab[sd[sector][2]] * 255.f);
}

}  // namespace

// ================================================================================================
// host-side tap tables and launch wrappers
// ================================================================================================
Taps gaussian_taps(double sigma, int fsize) {  // src/Image.h:1203-1219
    Taps f{};
    f.fsize = fsize;
    double sum = 0;
    const double s2 = sigma * sigma * 2;
    for (int i = -fsize; i <= fsize; i++) {
        f.t[i + fsize] = std::exp(-(double)(i * i) / s2);
        sum += f.t[i + fsize];
    }
    for (int i = 0; i < 2 * fsize + 1; i++) f.t[i] /= sum;
    return f;
}

Taps smooth5_taps() {  // src/OpticalFlow.cpp:84
    Taps f{};
    f.fsize = 2;
    const double g[5] = {0.02, 0.11, 0.74, 0.11, 0.02};
    for (int i = 0; i < 5; i++) f.t[i] = g[i];
    return f;
}

Taps deriv5_taps() {  // src/Image.h:987-992
    Taps f{};
    f.fsize = 2;
    const double d[5] = {1, -8, 0, 8, -1};
    for (int i = 0; i < 5; i++) {
        f.t[i] = d[i];
        f.t[i] /= 12;
    }
    return f;
}

Taps central3_taps() {  // src/Image.h:2589
    Taps f{};
    f.fsize = 1;
    f.t[0] = -0.5;
    f.t[1] = 0;
    f.t[2] = 0.5;
    return f;
}

#define LAUNCH_CHECK() PAPOF_HIP(hipGetLastError())

int hwc_to_planar(papof_handle* h, const double* hwc, double* planar, int H, int W, int C, int frames) {
    hipLaunchKernelGGL(k_hwc_to_planar, grid2d(W, H, frames), dim3(BX, BY), 0, h->stream, hwc, planar, H, W, C);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int hwc_u8_to_planar(papof_handle* h, const unsigned char* hwc, double* planar, int H, int W, int C, int frames) {
    hipLaunchKernelGGL(k_hwc_u8_to_planar, grid2d(W, H, frames), dim3(BX, BY), 0, h->stream, hwc, planar, H, W, C);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int planar_to_hwc(papof_handle* h, const double* planar, double* hwc, int H, int W, int C) {
    hipLaunchKernelGGL(k_planar_to_hwc, grid2d(W, H), dim3(BX, BY), 0, h->stream, planar, hwc, H, W, C);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int filter_h(papof_handle* h, const double* src, double* dst, int H, int W, int planes, const Taps& f, const Rect* rc) {
    const Rect r = region(rc, W, H);
    if (r.empty()) return PAPOF_OK;
    hipLaunchKernelGGL(k_filter_h, grid2d(r, planes), dim3(BX, BY), 0, h->stream, src, dst, H, W, f, r);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int filter_v(papof_handle* h, const double* src, double* dst, int H, int W, int planes, const Taps& f, const Rect* rc) {
    const Rect r = region(rc, W, H);
    if (r.empty()) return PAPOF_OK;
    hipLaunchKernelGGL(k_filter_v, grid2d(r, planes), dim3(BX, BY), 0, h->stream, src, dst, H, W, f, r);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int central3_planes(papof_handle* h, const double* src, double* gx, double* gy, double* gxy, int H, int W, int planes) {
    hipLaunchKernelGGL(k_central3_all, dim3(xcd_grid((W + BX - 1) / BX, (H + BY - 1) / BY), 1, planes), dim3(BX, BY), 0, h->stream, src, gx, gy, gxy, H, W,
                       central3_taps());
    LAUNCH_CHECK();
    return PAPOF_OK;
}

// dst = v-filter(h-filter(src)) in one launch when both half-widths fit the fused kernel, else through `tmp` (two passes)
int filter_hv(papof_handle* h, const double* src, double* dst, double* tmp, int H, int W, int planes, const Taps& fh,
              const Taps& fv) {
    if (fh.fsize > kHvMaxF || fv.fsize > kHvMaxF) {
        if (!tmp) return PAPOF_EINVAL;
        PAPOF_TRY(filter_h(h, src, tmp, H, W, planes, fh));
        return filter_v(h, tmp, dst, H, W, planes, fv);
    }
    const dim3 grid((W + BX - 1) / BX, (H + kHvRows - 1) / kHvRows, planes), block(BX, BY);
    const int f = fh.fsize == fv.fsize ? fh.fsize : -1;
    if (PAPOF_V_HVSTAGED && f >= 1 && f <= 3) {
        const dim3 sgrid(xcd_grid((W + BX - 1) / BX, (H + kHvsRows - 1) / kHvsRows), 1, planes);
        if (f == 1)
            hipLaunchKernelGGL(k_filter_hv_staged<1>, sgrid, block, 0, h->stream, src, dst, H, W, fh, fv);
        else if (f == 2)
            hipLaunchKernelGGL(k_filter_hv_staged<2>, sgrid, block, 0, h->stream, src, dst, H, W, fh, fv);
        else
            hipLaunchKernelGGL(k_filter_hv_staged<3>, sgrid, block, 0, h->stream, src, dst, H, W, fh, fv);
        LAUNCH_CHECK();
        return PAPOF_OK;
    }
    if (f == 1)
        hipLaunchKernelGGL(k_filter_hv<1>, grid, block, 0, h->stream, src, dst, H, W, fh, fv);
    else if (f == 2)
        hipLaunchKernelGGL(k_filter_hv<2>, grid, block, 0, h->stream, src, dst, H, W, fh, fv);
    else if (f == 3)
        hipLaunchKernelGGL(k_filter_hv<3>, grid, block, 0, h->stream, src, dst, H, W, fh, fv);
    else
        hipLaunchKernelGGL(k_filter_hv<-1>, grid, block, 0, h->stream, src, dst, H, W, fh, fv);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int resize(papof_handle* h, const double* src, double* dst, int sh, int sw, int planes, int dh, int dw, double xr,
           double yr, bool use_post, double post, const Rect* rc) {
    const Rect r = region(rc, dw, dh);
    if (r.empty()) return PAPOF_OK;
    hipLaunchKernelGGL(k_resize, dim3(xcd_grid((r.x1 - r.x0 + BX - 1) / BX, (r.y1 - r.y0 + BY - 1) / BY), 1, planes), dim3(BX, BY), 0, h->stream, src, dst, sh, sw, dh, dw, xr, yr,
                       use_post ? 1 : 0, post, r, take_stamp(h));
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int im2feature(papof_handle* h, const double* im, double* feat, int H, int W, int C, unsigned* nz, int frames, size_t nz_stride) {
    // frames > 1 (a batch): frame f reads im + f * (C planes), writes feat + f * (5 or 3 planes), flags nz + f * nz_stride
    const size_t np = (size_t)H * W;
    if (C == 3) {
        hipLaunchKernelGGL((frames > 1 ? k_im2feature_tiled<3, true> : k_im2feature_tiled<3, false>), dim3(xcd_grid((W + BX - 1) / BX, (H + kFeatRows - 1) / kFeatRows), frames), dim3(BX, BY), 0,
                           h->stream, im, feat, H, W, deriv5_taps(), nz, h->lap_epoch, 3 * np, 5 * np, nz_stride);
    } else if (C == 1) {
        hipLaunchKernelGGL((frames > 1 ? k_im2feature_tiled<1, true> : k_im2feature_tiled<1, false>), dim3(xcd_grid((W + BX - 1) / BX, (H + kFeatRows - 1) / kFeatRows), frames), dim3(BX, BY), 0,
                           h->stream, im, feat, H, W, deriv5_taps(), nz, h->lap_epoch, np, 3 * np, nz_stride);
    } else {  // src/OpticalFlow.cpp:956-957: any other channel count is passed through
        if (frames != 1) return PAPOF_EINVAL;
        PAPOF_HIP(hipMemcpyAsync(feat, im, sizeof(double) * (size_t)H * W * C, hipMemcpyDeviceToDevice, h->stream));
        return PAPOF_OK;
    }
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int warp_bilinear(papof_handle* h, const double* im1, const double* im2, const double* vx, const double* vy,
                  double* out, int H, int W, int planes, const Rect* rc) {
    const Rect r = region(rc, W, H);
    if (r.empty()) return PAPOF_OK;
    hipLaunchKernelGGL(k_warp, grid2d(r), dim3(BX, BY), 0, h->stream, im1, im2, vx, vy, out, H, W, planes, r,
                       take_stamp(h));
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int smooth_v_blend(papof_handle* h, const double* tmp, const double* im1s, double* blend, double* imdt, int H,
                   int W, int planes, const Rect* rc) {
    const Rect r = region(rc, W, H);
    if (r.empty()) return PAPOF_OK;
    hipLaunchKernelGGL(k_smooth_v_blend, grid2d(r, planes), dim3(BX, BY), 0, h->stream, tmp, im1s, blend, imdt, H, W,
                       smooth5_taps(), r);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

static SkewIdx skew_idx(const SorPlanes& sp) {
    return SkewIdx{sp.sd.hp, sp.sd.qt, sp.sd.rt, sp.sd.nb, sp.sd.npos_d, sp.sd.dpar, sp.sd.band_rows, sp.sd.koff, sp.sd.poff};
}

// `prev` = operands of the previous inner iteration's solve (nullptr in the first one: du = dv = 0)
static Increment increment_of(const SorPlanes* prev, const double* gm = nullptr, const double* lap = nullptr) {
    if (!prev) return Increment{nullptr, nullptr, 0, SkewIdx{0, 0, 0, 0, 0, 0, 0}, gm, lap};
    return Increment{prev->du, prev->dv, prev->skew ? 1 : 0, prev->skew ? skew_idx(*prev) : SkewIdx{0, 0, 0, 0, 0, 0, 0},
                     gm, lap};
}

int smooth_hv_blend(papof_handle* h, const double* warp, const double* im1s, double* blend, double* imdt, int H,
                    int W, int planes, int row0, int row1) {
    if (row1 < 0) row1 = H;
    if (row0 < 0 || row1 > H) return PAPOF_EINVAL;
    if (row1 <= row0) return PAPOF_OK;
    hipLaunchKernelGGL(k_smooth_hv_blend, dim3((W + BX - 1) / BX, (row1 - row0 + kFuseRows - 1) / kFuseRows, planes),
                       dim3(BX, BY), 0, h->stream, warp, im1s, blend, imdt, H, W, smooth5_taps(), take_stamp(h), row0,
                       row1);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int warp_smooth_blend(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v,
                      const double* im1s, double* blend, double* imdt, int H, int W, int planes, unsigned* wit,
                      double* phi_out) {
    hipLaunchKernelGGL(k_warp_smooth_blend, dim3((W + BX - 1) / BX, (H + kWsRows - 1) / kWsRows, planes), dim3(BX, BY), 0,
                       h->stream, im1, im2, u, v, im1s, blend, imdt, H, W, smooth5_taps(), take_stamp(h), wit,
                       2e-20 * (double)H * (double)W, h->lap_epoch, phi_out);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int compute_phi(papof_handle* h, const double* u, const double* v, const SorPlanes* prev, double* phi, int H, int W,
                const Rect* rc) {
    const Rect r = region(rc, W, H);
    if (r.empty()) return PAPOF_OK;
    hipLaunchKernelGGL(k_phi, grid2d(r), dim3(BX, BY), 0, h->stream, u, v, increment_of(prev), phi, H, W, r,
                       take_stamp(h));
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int assemble_system(papof_handle* h, const double* blend, const double* imdt, const double* phi, const double* u,
                    const double* v, int H, int W, int planes, double alpha, double omega, const SorPlanes& out,
                    double* opt_imdx2, double* opt_imdy2, const SorPlanes* prev, const Rect* rc, const double* gm,
                    const double* lap) {
    const Increment I = increment_of(prev, gm, lap);
    const Rect r = region(rc, W, H);
    if (r.empty()) return PAPOF_OK;
    if (out.skew) {
        if (rc && (r.x0 != 0 || r.x1 != W)) return PAPOF_EINVAL;  // skew layout: whole rows only (strips of a plane)
        // 5 feature channels (colour frames) and 3 (gray frames) have their own instantiation with the channel loop unrolled
        const bool fast = I.du == nullptr && I.gm == nullptr && I.lap == nullptr;  // the default branches (assemble_cell: FAST; 0.04 ms)
        // Where its 109 us at level 0 go (variants of the kernel under rocprofv3, round 3): without the three plane stores 79 us
        // (100 MB at the ~3.3 TB/s a plain fill reaches too), without the stencil loads 82, without sqrt / division 106 -- loads and
        // stores, not arithmetic.  Not kept: tiles of 16x32 / 24x32 / 32x32 / 32x16 (512-byte chunks: 0 ... +0.12 ms per 1080p pair),
        // non-temporal plane stores (+0.26 ms: the solver wants the planes where these stores leave them), the blended tile
        // staged in LDS (+0.05 ms).
        const auto kern = planes == 5   ? (fast ? k_assemble_skew<5, true> : k_assemble_skew<5, false>)
                          : planes == 3 ? (fast ? k_assemble_skew<3, true> : k_assemble_skew<3, false>)
                                        : k_assemble_skew<0, false>;
        hipLaunchKernelGGL(kern, dim3((W + kTileJ - 1) / kTileJ, (r.y1 - r.y0 + kAsmRows - 1) / kAsmRows),
                           dim3(256), 0, h->stream, blend, imdt, phi, u, v, H, W, planes, alpha, omega, skew_idx(out),
                           (double2s*)out.phi, (double2s*)out.a1, (double2s*)out.b1, opt_imdx2, opt_imdy2,
                           deriv5_taps(), I, take_stamp(h), r.y0, r.y1);
    } else {
        hipLaunchKernelGGL(k_assemble, grid2d(r), dim3(BX, BY), 0, h->stream, blend, imdt, phi, u, v, H, W, planes,
                           alpha, omega, out.phi, out.xy, out.a1, out.a2, out.b1, out.b2, opt_imdx2, opt_imdy2,
                           deriv5_taps(), I, r, take_stamp(h));
    }
    LAUNCH_CHECK();
    return PAPOF_OK;
}

// k_flow_system: the default branches of warp_smooth_blend() + assemble_system() in one launch (exact-order layout, 5 or 3
// feature channels); returns PAPOF_EINVAL where it does not apply
int flow_system(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v, const double* im1s,
                int H, int W, int planes, double alpha, double omega, const SorPlanes& out, unsigned* wit, int row0, int row1,
                int batch, const BatchK* bk) {
    if (planes != 5 && planes != 3) return PAPOF_EINVAL;
    if (row1 < 0) row1 = H;
    if (row0 < 0 || row1 > H) return PAPOF_EINVAL;
    if (row1 <= row0) return PAPOF_OK;
    // tile width: 16 columns (256 threads) or 32 (512 threads: 1.875 instead of 2.25 warped pixels per cell, two slots of taps
    // and gathers per thread instead of three) -- PAPOF_FS_TX, read once per process (an A/B switch; both forms give the same bits)
    static const int tx_env = std::getenv("PAPOF_FS_TX") ? std::atoi(std::getenv("PAPOF_FS_TX")) : 0;
    const int TXr = tx_env == 32 ? 32 : (tx_env == 16 ? 16 : kFlowSystemTX);
    const int ntiles = ((W + TXr - 1) / TXr) * ((row1 - row0 + kFT - 1) / kFT);
    const dim3 grid(8 * ((ntiles + 7) / 8), batch);  // (the kernel maps block -> tile: a contiguous run of tiles per XCD)
    const SixPlanes six{{out.phi, out.xy, out.a1, out.a2, out.b1, out.b2}};
    const BatchK bk0{0, 0, 0, 0, 0, 0};
    const auto kern16 = batch > 1 ? (out.skew ? (planes == 5 ? k_flow_system<5, true, true> : k_flow_system<3, true, true>)
                                              : (planes == 5 ? k_flow_system<5, false, true> : k_flow_system<3, false, true>))
                                  : (out.skew ? (planes == 5 ? k_flow_system<5, true> : k_flow_system<3, true>)
                                              : (planes == 5 ? k_flow_system<5, false> : k_flow_system<3, false>));
    const auto kern32 = batch > 1 ? (out.skew ? (planes == 5 ? k_flow_system<5, true, true, 32> : k_flow_system<3, true, true, 32>)
                                              : (planes == 5 ? k_flow_system<5, false, true, 32> : k_flow_system<3, false, true, 32>))
                                  : (out.skew ? (planes == 5 ? k_flow_system<5, true, false, 32> : k_flow_system<3, true, false, 32>)
                                              : (planes == 5 ? k_flow_system<5, false, false, 32> : k_flow_system<3, false, false, 32>));
    const auto kern = TXr == 32 ? kern32 : kern16;
    hipLaunchKernelGGL(kern, grid, dim3(TXr * kFT), 0, h->stream, im1, im2, u, v, im1s, H, W, alpha, omega,
                       out.skew ? skew_idx(out) : SkewIdx{0, 0, 0, 0, 0, 0, 0, 0, 0}, six, smooth5_taps(), deriv5_taps(),
                       take_stamp(h), wit, 2e-20 * (double)H * (double)W, h->lap_epoch, row0, row1, bk ? *bk : bk0);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int laplacian(papof_handle* h, const double* in, const double* weight, double* out, int H, int W) {
    hipLaunchKernelGGL(k_laplacian, grid2d(W, H), dim3(BX, BY), 0, h->stream, in, weight, out, H, W);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int update_and_warp(papof_handle* h, const SorPlanes& sp, double* u, double* v, const double* im1,
                    const double* im2, double* warp, int H, int W, int planes, bool do_warp) {
    if (sp.skew) {
        hipLaunchKernelGGL(k_update_warp_skew, grid2d(W, H), dim3(BX, BY), 0, h->stream, (const double2s*)sp.du, u, v,
                           im1, im2, warp, H, W, planes, skew_idx(sp), take_stamp(h), do_warp ? 1 : 0);
    } else {
        hipLaunchKernelGGL(k_update_warp, grid2d(W, H), dim3(BX, BY), 0, h->stream, sp.du, sp.dv, u, v, im1, im2, warp,
                           H, W, planes, Rect{0, 0, W, H}, do_warp ? 1 : 0, take_stamp(h));
    }
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int update_warp_phi(papof_handle* h, const SorPlanes& sp, const double* u, const double* v, double* u_out, double* v_out,
                    const double* im1, const double* im2, double* warp, double* phi_out, int H, int W, int planes,
                    bool do_warp, int row0, int row1, unsigned* wit, int batch, const BatchK* bk) {
    const double wit_thr = 2e-20 * (double)H * (double)W;
    const BatchK bk0{0, 0, 0, 0, 0, 0};
    if (batch > 1 && (warp || phi_out || do_warp)) return PAPOF_EINVAL;
    if (u == u_out || v == v_out) return PAPOF_EINVAL;
    if (row1 < 0) row1 = H;
    if (row0 < 0 || row1 > H) return PAPOF_EINVAL;
    if (row1 <= row0) return PAPOF_OK;
    const dim3 grid(xcd_grid((W + BX - 1) / BX, (row1 - row0 + BY - 1) / BY), batch);
    if (sp.skew)
        hipLaunchKernelGGL((batch > 1 ? k_update_warp_phi<true, true> : k_update_warp_phi<true, false>), grid, dim3(BX, BY), 0,
                           h->stream, sp.du, sp.dv, skew_idx(sp),
                           u, v, u_out, v_out, im1, im2, warp, phi_out, H, W, planes, do_warp ? 1 : 0, take_stamp(h),
                           row0, row1, wit, wit_thr, h->lap_epoch, bk ? *bk : bk0);
    else
        hipLaunchKernelGGL((batch > 1 ? k_update_warp_phi<false, true> : k_update_warp_phi<false, false>), grid, dim3(BX, BY), 0, h->stream, sp.du, sp.dv,
                           SkewIdx{0, 0, 0, 0, 0, 0, 0, 0, 0}, u, v, u_out, v_out, im1, im2, warp, phi_out, H, W, planes,
                           do_warp ? 1 : 0, take_stamp(h), row0, row1, wit, wit_thr, h->lap_epoch, bk ? *bk : bk0);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

// tile path: u += du, v += dv on a region, no warp (the warp follows the halo exchange of u, v)
int update_flow(papof_handle* h, const SorPlanes& sp, double* u, double* v, int H, int W, const Rect& r) {
    if (sp.skew) return PAPOF_EINVAL;
    if (r.empty()) return PAPOF_OK;
    hipLaunchKernelGGL(k_update_warp, grid2d(r), dim3(BX, BY), 0, h->stream, sp.du, sp.dv, u, v, nullptr, nullptr, nullptr,
                       H, W, 0, r, 0, take_stamp(h));
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int flow_quantize16(papof_handle* h, const double* vx, const double* vy, unsigned short* q, size_t n) {
    if (!n) return PAPOF_OK;
    hipLaunchKernelGGL(k_flow_quantize16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, vx, vy, q, n);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int flow_dequantize16(papof_handle* h, const unsigned short* q, double* vx, double* vy, size_t n) {
    if (!n) return PAPOF_OK;
    hipLaunchKernelGGL(k_flow_dequantize16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, q, vx, vy, n);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int flow_to_bgr(papof_handle* h, const double* vx, const double* vy, size_t n, double* partial /* 2 * 256 */,
                unsigned char* bgr) {
    if (!n) return PAPOF_OK;
    const int nblocks = 256;
    hipLaunchKernelGGL(k_flow_mag_minmax, dim3(nblocks), dim3(256), 0, h->stream, vx, vy, n, partial);
    hipLaunchKernelGGL(k_flow_bgr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, vx, vy, n, partial,
                       nblocks, bgr);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int bicubic_warp(papof_handle* h, const double* im1, const double* im2, const double* gx, const double* gy,
                 const double* gxy, const double* vx, const double* vy, double* out_hwc, int H, int W, int C,
                 const Rect* rc, bool planar_out, bool clamp, int batch, const BatchK* bk) {
    const Rect r = region(rc, W, H);
    if (r.empty()) return PAPOF_OK;
    const BatchK bk0{0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(batch > 1 ? k_bicubic<true> : k_bicubic<false>, dim3(xcd_grid((r.x1 - r.x0 + BX - 1) / BX, (r.y1 - r.y0 + BY - 1) / BY), batch), dim3(BX, BY), 0, h->stream, im1, im2, gx, gy, gxy, vx, vy, out_hwc, H, W,
                       C, r, take_stamp(h), planar_out ? 1 : 0, clamp ? 1 : 0, bk ? *bk : bk0);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

// ------------------------------------------------------------------------------------------------
// OpticalFlow::estGaussianMixture, src/OpticalFlow.cpp:539-591 (prior 0.9): three EM iterations over all pixels and
// channels.  One iteration = one reduction kernel (per channel: sum w1, sum w2, sum w1*d^2, sum w2*d^2 with the weights
// of the CURRENT parameters, :551-575; the reference stores the weights, here they are recomputed -- same expressions)
// + one single-block kernel that sums the per-block partials in a fixed order and applies the M step (:576-583, with
// the reference's quirk of accumulating onto the reset values 0.05 / 0.5).  The reference sums sequentially over the
// pixels; a parallel sum rounds differently in the last bits: this branch is checked with a tolerance.
// Deterministic: fixed grid, fixed in-block tree, fixed order over the blocks.
// ------------------------------------------------------------------------------------------------
constexpr int kGmBlocks = 256, kGmThreads = 256, kGmMaxC = 8;
static __global__ __launch_bounds__(kGmThreads) void k_gm_partial(const double* __restrict__ im1,
                                                           const double* __restrict__ im2, size_t np, int C,
                                                           const double* __restrict__ gm, double* __restrict__ partial) {
    __shared__ double red[kGmThreads];
    double acc[kGmMaxC][4];
    for (int k = 0; k < kGmMaxC; k++) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0;
    for (size_t o = (size_t)blockIdx.x * kGmThreads + threadIdx.x; o < np; o += (size_t)kGmBlocks * kGmThreads) {
#pragma unroll
        for (int k = 0; k < kGmMaxC; k++) {
            if (k >= C) break;
            double t = im1[k * np + o] - im2[k * np + o];
            t *= t;
            const double alpha_k = gm[k], sigma = gm[C + k], beta = gm[2 * C + k], s2 = gm[3 * C + k], b2 = gm[4 * C + k];
            double w1 = exp(-t / (2 * s2)) / (2 * 3.1415927 * sigma) * alpha_k;
            double w2 = exp(-t / (2 * b2)) / (2 * 3.1415927 * beta) * (1 - alpha_k);
            const double s = w1 + w2;
            w1 /= s;
            w2 /= s;
            acc[k][0] += w1;
            acc[k][1] += w2;
            acc[k][2] += w1 * t;
            acc[k][3] += w2 * t;
        }
    }
    for (int k = 0; k < C; k++)
        for (int q = 0; q < 4; q++) {
            red[threadIdx.x] = acc[k][q];
            __syncthreads();
            for (int s = kGmThreads / 2; s > 0; s >>= 1) {
                if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
                __syncthreads();
            }
            if (threadIdx.x == 0) partial[((size_t)blockIdx.x * kGmMaxC + k) * 4 + q] = red[0];
            __syncthreads();
        }
}

static __global__ void k_gm_mstep(const double* __restrict__ partial, int C, double prior, double* __restrict__ gm) {
    const int k = threadIdx.x;
    if (k >= C) return;
    double t1 = 0.0, t2 = 0.0, s1 = 0.05, s2 = 0.5;  // para.reset() before the accumulation (:564, src/NoiseModel.h:97-107)
    for (int b = 0; b < kGmBlocks; b++) {
        const double* p = partial + ((size_t)b * kGmMaxC + k) * 4;
        t1 += p[0];
        t2 += p[1];
        s1 += p[2];
        s2 += p[3];
    }
    const double alpha = t1 / (t1 + t2) * (1 - prior) + 0.95 * prior;
    const double sigma = sqrt(s1 / t1);
    const double beta = sqrt(s2 / t2) * (1 - prior) + 0.3 * prior;
    gm[k] = alpha;
    gm[C + k] = sigma;
    gm[2 * C + k] = beta;
    gm[3 * C + k] = sigma * sigma;
    gm[4 * C + k] = beta * beta;
}

int gm_scratch_doubles() { return kGmBlocks * kGmMaxC * 4; }

// gm (device, 5 * C doubles) in/out; scratch: gm_scratch_doubles() doubles
int est_gaussian_mixture(papof_handle* h, const double* im1, const double* im2, int H, int W, int C, double* gm,
                         double* scratch) {
    if (C > kGmMaxC) return PAPOF_EINVAL;
    for (int it = 0; it < 3; it++) {
        hipLaunchKernelGGL(k_gm_partial, dim3(kGmBlocks), dim3(kGmThreads), 0, h->stream, im1, im2, (size_t)H * W, C, gm,
                           scratch);
        hipLaunchKernelGGL(k_gm_mstep, dim3(1), dim3(64), 0, h->stream, scratch, C, 0.9, gm);
    }
    LAUNCH_CHECK();
    return PAPOF_OK;
}

// ------------------------------------------------------------------------------------------------
// OpticalFlow::estLaplacianNoise, src/OpticalFlow.cpp:594-639, for the EXACT pass of the Laplacian-noise guard (api.hip:
// LapGuard; only calls whose optimistic pass found a channel without a witness get here): per channel the mean of
// |Im1 - warpIm2| over the samples in (0, 1e6), 0.001 when there is none.  The warped frame is evaluated on the fly (it is
// never materialised on this path; u == nullptr: im2 is the warped frame, the bicubic branch).  One reduction kernel (per block: sum and count in a fixed tree) + one single-block
// kernel that adds the blocks' partials in block order: deterministic; the reference adds sequentially over the pixels, so
// the mean can differ in its last bits -- it only feeds the `< 1E-20` test.
// ------------------------------------------------------------------------------------------------
constexpr int kLapBlocks = 256, kLapThreads = 256, kLapMaxC = 8;
static __global__ __launch_bounds__(kLapThreads) void k_lap_partial(const double* __restrict__ im1,
                                                             const double* __restrict__ im2,
                                                             const double* __restrict__ u, const double* __restrict__ v,
                                                             int H, int W, int C, double* __restrict__ partial) {
    __shared__ double red[kLapThreads];
    const size_t np = (size_t)H * W;
    double acc[kLapMaxC][2];
    for (int k = 0; k < kLapMaxC; k++) acc[k][0] = acc[k][1] = 0.0;
    for (size_t o = (size_t)blockIdx.x * kLapThreads + threadIdx.x; o < np; o += (size_t)kLapBlocks * kLapThreads) {
        const int i = (int)(o / W), j = (int)(o - (size_t)i * W);
        const double fu = u ? u[o] : 0.0, fv = u ? v[o] : 0.0;  // u == nullptr: im2 IS the warped frame (bicubic branch)
#pragma unroll
        for (int k = 0; k < kLapMaxC; k++) {
            if (k >= C) break;
            const double w = u ? warp_value(im1 + k * np, im2 + k * np, fu, fv, i, j, H, W) : im2[k * np + o];
            const double d = fabs(im1[k * np + o] - w);
            if (d > 0 && d < 1000000) {
                acc[k][0] += d;
                acc[k][1] += 1.0;
            }
        }
    }
    for (int k = 0; k < C; k++)
        for (int q = 0; q < 2; q++) {
            red[threadIdx.x] = acc[k][q];
            __syncthreads();
            for (int s = kLapThreads / 2; s > 0; s >>= 1) {
                if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
                __syncthreads();
            }
            if (threadIdx.x == 0) partial[((size_t)blockIdx.x * kLapMaxC + k) * 2 + q] = red[0];
            __syncthreads();
        }
}
static __global__ void k_lap_finish(const double* __restrict__ partial, int C, double* __restrict__ lap) {
    const int k = threadIdx.x;
    if (k >= C) return;
    double s = 0.0, n = 0.0;
    for (int b = 0; b < kLapBlocks; b++) {
        s += partial[((size_t)b * kLapMaxC + k) * 2];
        n += partial[((size_t)b * kLapMaxC + k) * 2 + 1];
    }
    lap[k] = n == 0 ? 0.001 : s / n;
}
// The witness / no-valid-sample check of k_warp_smooth_blend's exhaustive mode for the LAST outer iteration of a one-block
// level (nobody evaluates the warp at that flow): one block per channel over every pixel.
static __global__ __launch_bounds__(256) void k_lap_small(const double* __restrict__ im1, const double* __restrict__ im2,
                                                       const double* __restrict__ u, const double* __restrict__ v, int H,
                                                       int W, unsigned* __restrict__ wit, double wit_thr, unsigned mark,
                                                       BatchK bk) {
    const int np = H * W;
    {   // blockIdx.y: the pair of a batch (common.h: BatchK)
        const size_t p = blockIdx.y;
        im1 += p * bk.im;
        im2 += p * bk.im;
        u += p * bk.uv;
        v += p * bk.uv;
        wit += p * bk.wit;
    }
    const double *p1 = im1 + (size_t)blockIdx.x * np, *p2 = im2 + (size_t)blockIdx.x * np;
    bool hit = false, valid = false;
    for (int o = threadIdx.x; o < np; o += 256) {
        const int i = o / W, j = o - i * W;
        const double d = fabs(p1[o] - warp_value(p1, p2, u[o], v[o], i, j, H, W));
        valid = valid || (d > 0 && d < 1000000);
        hit = hit || (d >= wit_thr && d < 1000000);
    }
    const int any_hit = __syncthreads_or(hit ? 1 : 0), any_valid = __syncthreads_or(valid ? 1 : 0);
    if (threadIdx.x == 0 && (any_hit || !any_valid)) wit[blockIdx.x] = any_hit ? mark : (mark ^ kLapNone);
}
// The same two flags over a RANGE OF ROWS (every pixel of rows r0 .. r1-1), for a rank of the exact-order band split
// (tiles.hip: bands_flow): wit[k] = mark when a sample witnesses, val[k] = mark when the rows hold a valid sample at all.
// Proofs combine over the ranks: a witness anywhere, or no valid sample anywhere.
static __global__ __launch_bounds__(256) void k_lap_rows(const double* __restrict__ im1, const double* __restrict__ im2,
                                                      const double* __restrict__ u, const double* __restrict__ v, int H,
                                                      int W, int r0, int r1, unsigned* __restrict__ wit,
                                                      unsigned* __restrict__ val, double wit_thr, unsigned mark) {
    const size_t np = (size_t)H * W;
    const double *p1 = im1 + blockIdx.y * np, *p2 = im2 + blockIdx.y * np;
    bool hit = false, valid = false;
    const int n = (r1 - r0) * W;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
        const int i = r0 + c / W, j = c - (c / W) * W;
        const size_t o = (size_t)i * W + j;
        const double d = fabs(p1[o] - warp_value(p1, p2, u[o], v[o], i, j, H, W));
        valid = valid || (d > 0 && d < 1000000);
        hit = hit || (d >= wit_thr && d < 1000000);
    }
    const int any_hit = __syncthreads_or(hit ? 1 : 0), any_valid = __syncthreads_or(valid ? 1 : 0);
    if (threadIdx.x == 0) {
        if (any_hit) wit[blockIdx.y] = mark;
        if (any_valid) val[blockIdx.y] = mark;
    }
}
int lap_rows_check(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v, int H, int W,
                   int C, int r0, int r1, unsigned* wit, unsigned* val, unsigned mark) {
    if (r1 <= r0) return PAPOF_OK;
    const int blocks = std::min(256, ((r1 - r0) * W + 255) / 256);
    hipLaunchKernelGGL(k_lap_rows, dim3(blocks, C), dim3(256), 0, h->stream, im1, im2, u, v, H, W, r0, r1, wit, val,
                       2e-20 * (double)H * (double)W, mark);
    LAUNCH_CHECK();
    return PAPOF_OK;
}
bool lap_one_block_level(int H, int W) { return H <= kWsRows && W <= BX; }
int lap_small_check(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v, int H, int W,
                    int C, unsigned* wit, int batch, const BatchK* bk) {
    const BatchK bk0{0, 0, 0, 0, 0, 0};
    hipLaunchKernelGGL(k_lap_small, dim3(C, batch), dim3(256), 0, h->stream, im1, im2, u, v, H, W, wit,
                       2e-20 * (double)H * (double)W, h->lap_epoch, bk ? *bk : bk0);
    LAUNCH_CHECK();
    return PAPOF_OK;
}
int lap_scratch_doubles() { return kLapBlocks * kLapMaxC * 2; }
int est_laplacian_noise(papof_handle* h, const double* im1, const double* im2, const double* u, const double* v, int H,
                        int W, int C, double* lap, double* scratch) {
    if (C > kLapMaxC) return PAPOF_EINVAL;
    hipLaunchKernelGGL(k_lap_partial, dim3(kLapBlocks), dim3(kLapThreads), 0, h->stream, im1, im2, u, v, H, W, C, scratch);
    hipLaunchKernelGGL(k_lap_finish, dim3(1), dim3(64), 0, h->stream, scratch, C, lap);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

// closes the last phase of a call: a one-thread kernel that only writes the pending stamp (if any)
int stamp_only(papof_handle* h) {
    unsigned long long* s = take_stamp(h);
    if (!s) return PAPOF_OK;
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, h->stream, s);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int sor_prep(papof_handle* h, const double* phi, const double* imdxy, const double* imdx2, const double* imdy2,
             const double* rhs1, const double* rhs2, int H, int W, double alpha, double omega, const SorPlanes& out) {
    if (out.skew)
        hipLaunchKernelGGL(k_sor_prep<true>, grid2d(W, H), dim3(BX, BY), 0, h->stream, phi, imdxy, imdx2, imdy2,
                           rhs1, rhs2, H, W, alpha, omega, skew_idx(out), out.phi, out.xy, out.a1, out.a2, out.b1,
                           out.b2);
    else
        hipLaunchKernelGGL(k_sor_prep<false>, grid2d(W, H), dim3(BX, BY), 0, h->stream, phi, imdxy, imdx2, imdy2,
                           rhs1, rhs2, H, W, alpha, omega, SkewIdx{0, 0, 0, 0, 0, 0, 0}, out.phi, out.xy, out.a1, out.a2, out.b1,
                           out.b2);
    LAUNCH_CHECK();
    return PAPOF_OK;
}

int sor_unpack(papof_handle* h, const SorPlanes& sp, double* du, double* dv, int H, int W) {
    if (sp.skew)
        hipLaunchKernelGGL(k_sor_unpack<true>, grid2d(W, H), dim3(BX, BY), 0, h->stream, sp.du, sp.dv, du, dv, H, W,
                           skew_idx(sp));
    else
        hipLaunchKernelGGL(k_sor_unpack<false>, grid2d(W, H), dim3(BX, BY), 0, h->stream, sp.du, sp.dv, du, dv, H,
                           W, SkewIdx{0, 0, 0, 0, 0, 0, 0});
    LAUNCH_CHECK();
    return PAPOF_OK;
}

}  // namespace papof
