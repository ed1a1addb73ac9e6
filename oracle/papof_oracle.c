/* oracle/papof_oracle.c -- TEST INFRASTRUCTURE ONLY (see papof_oracle.h for scope and pin status).
 *
 * CPU restatement of the reference hot path.  Written from the behavioural spec in SURVEY.md §8(a);
 * each function cites the reference lines whose floating-point operation ORDER it reproduces
 * (the gate is bit-for-bit equality with the untouched reference, so evaluation order is part of
 * the spec).  Paths are relative to /root/reference/Code/Serial/.
 *
 * Build: gcc -O2 -ffp-contract=off -std=c99 (oracle/Makefile).  No FMA, no fast-math.
 */
#define _POSIX_C_SOURCE 199309L
#include "papof_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static double* zalloc(size_t n) {
    double* p = (double*)calloc(n ? n : 1, sizeof(double));
    if (!p) abort();
    return p;
}

/* EnforceRange: src/ImageProcessing.h:34 -- clamp an index to [0, n-1]. */
static inline int clampi(int x, int n) {
    if (x < 0) x = 0;
    if (x > n - 1) x = n - 1;
    return x;
}

void orc_default_params(orc_params* p) {
    p->alpha = 0.012; /* src/OpticalFlow.cpp:747 */
    p->ratio = 0.75;  /* :748 */
    p->n_outer = 7;   /* :749 */
    p->n_outer_per_level = 1;
    p->n_inner = 1; /* :750 */
    p->n_sor = 30;  /* :751 */
    p->n_sor_per_level = 3;
    p->omega = 1.8; /* :451 */
    p->sor_mode = ORC_SOR_EXACT;
    p->interpolation = ORC_INTERP_BILINEAR; /* :33 */
    p->noise_model = ORC_NOISE_LAP;         /* :34 */
}

/* ---------------------------------------------------------------------------------------------
 * Separable correlation with clamped borders.
 * src/ImageProcessing.h:259-279 (hfiltering), :350-369 (vfiltering): destination zeroed, taps
 * accumulated in order l = -fsize..fsize.
 * ------------------------------------------------------------------------------------------- */
void orc_hfilter(const double* src, double* dst, int w, int h, int c, const double* f, int fsize) {
    memset(dst, 0, sizeof(double) * (size_t)w * h * c);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            double* out = dst + ((size_t)i * w + j) * c;
            for (int l = -fsize; l <= fsize; l++) {
                const double tap = f[l + fsize];
                const double* in = src + ((size_t)i * w + clampi(j + l, w)) * c;
                for (int k = 0; k < c; k++) out[k] += in[k] * tap;
            }
        }
}

void orc_vfilter(const double* src, double* dst, int w, int h, int c, const double* f, int fsize) {
    memset(dst, 0, sizeof(double) * (size_t)w * h * c);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            double* out = dst + ((size_t)i * w + j) * c;
            for (int l = -fsize; l <= fsize; l++) {
                const double tap = f[l + fsize];
                const double* in = src + ((size_t)clampi(i + l, h) * w + j) * c;
                for (int k = 0; k < c; k++) out[k] += in[k] * tap;
            }
        }
}

/* src/Image.h:1347-1356 (imfilter_hv): horizontal pass into a temporary, then vertical pass. */
void orc_filter_hv(const double* src, double* dst, int w, int h, int c, const double* hf, int hfs,
                   const double* vf, int vfs) {
    double* tmp = zalloc((size_t)w * h * c);
    orc_hfilter(src, tmp, w, h, c, hf, hfs);
    orc_vfilter(tmp, dst, w, h, c, vf, vfs);
    free(tmp);
}

/* src/Image.h:1203-1225 (GaussianSmoothing): taps exp(-i^2/(2 sigma^2)) normalised by their sum. */
void orc_gaussian_smoothing(const double* src, double* dst, int w, int h, int c, double sigma, int fsize) {
    double* g = zalloc((size_t)(2 * fsize + 1));
    double sum = 0;
    const double s2 = sigma * sigma * 2;
    for (int i = -fsize; i <= fsize; i++) {
        g[i + fsize] = exp(-(double)(i * i) / s2);
        sum += g[i + fsize];
    }
    for (int i = 0; i < 2 * fsize + 1; i++) g[i] /= sum;
    orc_filter_hv(src, dst, w, h, c, g, fsize, g, fsize);
    free(g);
}

/* ---------------------------------------------------------------------------------------------
 * Bilinear sample, ACCUMULATED into result (src/ImageProcessing.h:138-157): truncation toward
 * zero for the integer part, fractional part clamped to [0,1], taps visited x-offset outer,
 * y-offset inner, indices clamped to the image.
 * ------------------------------------------------------------------------------------------- */
static inline void bilinear_acc(const double* im, int w, int h, int c, double x, double y, double* result) {
    const int xx = (int)x, yy = (int)y;
    double dx = x - xx, dy = y - yy;
    dx = dx > 1 ? 1.0 : dx; /* __min(double,int) then __max(double,int): src/project.h:41-50 */
    dx = dx < 0 ? 0.0 : dx;
    dy = dy > 1 ? 1.0 : dy;
    dy = dy < 0 ? 0.0 : dy;
    for (int m = 0; m <= 1; m++)
        for (int n = 0; n <= 1; n++) {
            const int u = clampi(xx + m, w), v = clampi(yy + n, h);
            const double s = fabs((double)(1 - m) - dx) * fabs((double)(1 - n) - dy);
            const double* p = im + ((size_t)v * w + u) * c;
            for (int l = 0; l < c; l++) result[l] += p[l] * s;
        }
}

/* src/ImageProcessing.h:214-232: destination dims are int((double)S*ratio) (src/Image.h:755-756). */
void orc_resize_ratio(const double* src, double* dst, int sw, int sh, int c, double ratio) {
    const int dw = (int)((double)sw * ratio), dh = (int)((double)sh * ratio);
    memset(dst, 0, sizeof(double) * (size_t)dw * dh * c);
    for (int i = 0; i < dh; i++)
        for (int j = 0; j < dw; j++) {
            const double x = (double)(j + 1) / ratio - 1;
            const double y = (double)(i + 1) / ratio - 1;
            bilinear_acc(src, sw, sh, c, x, y, dst + ((size_t)i * dw + j) * c);
        }
}

/* src/ImageProcessing.h:235-253: separate x / y ratios from the integer target size. */
void orc_resize_wh(const double* src, double* dst, int sw, int sh, int c, int dw, int dh) {
    const double xr = (double)dw / sw, yr = (double)dh / sh;
    memset(dst, 0, sizeof(double) * (size_t)dw * dh * c);
    for (int i = 0; i < dh; i++)
        for (int j = 0; j < dw; j++) {
            const double x = (double)(j + 1) / xr - 1;
            const double y = (double)(i + 1) / yr - 1;
            bilinear_acc(src, sw, sh, c, x, y, dst + ((size_t)i * dw + j) * c);
        }
}

/* ---------------------------------------------------------------------------------------------
 * GaussianPyramid::ConstructPyramidLevels, src/GaussianPyramid.cpp:79-108.
 * ------------------------------------------------------------------------------------------- */
/* GaussianPyramid::ConstructPyramid (src/GaussianPyramid.cpp:47-77) differs from ConstructPyramidLevels (:79-108) in
 * ONE line: the level count comes from a minimum width, nLevels = log((double)minWidth / width) / log(ratio) (:53,
 * double -> int truncation), after the same clamp of the ratio (:50-51).  Everything else is orc_pyramid(). */
int orc_pyramid_levels_for_min_width(int width, double ratio, int min_width) {
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75;
    const int n = log((double)min_width / width) / log(ratio);
    return n;
}

long orc_pyramid(const double* im, int h, int w, int c, double ratio, int levels, int* dims, double* data) {
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75; /* :82-83 */
    const double base_sigma = 1 / ratio - 1;       /* :89 */
    const int n = (int)(log(0.25) / log(ratio));   /* :90 */
    const double n_sigma = base_sigma * n;         /* :91 */
    long* offs = (long*)calloc((size_t)levels + 1, sizeof(long));
    dims[0] = w;
    dims[1] = h;
    offs[0] = 0;
    offs[1] = (long)w * h * c;
    if (data) memcpy(data, im, sizeof(double) * (size_t)w * h * c);
    for (int i = 1; i < levels; i++) {
        int sw, sh;
        const double* src;
        double sigma, rate;
        int fsize;
        if (i <= n) { /* :95-100: smooth the ORIGINAL, resize by ratio^i */
            sw = w;
            sh = h;
            src = im;
            sigma = base_sigma * i;
            fsize = (int)(sigma * 3); /* double -> int parameter conversion, :98 */
            rate = pow(ratio, i);
        } else { /* :101-106: smooth level i-n, resize by ratio^i * W / width(level i-n) */
            sw = dims[2 * (i - n)];
            sh = dims[2 * (i - n) + 1];
            src = data ? data + offs[i - n] : NULL;
            sigma = n_sigma;
            fsize = (int)(n_sigma * 3);
            rate = (double)pow(ratio, i) * w / sw;
        }
        const int dw = (int)((double)sw * rate), dh = (int)((double)sh * rate);
        dims[2 * i] = dw;
        dims[2 * i + 1] = dh;
        offs[i + 1] = offs[i] + (long)dw * dh * c;
        if (data) {
            double* foo = zalloc((size_t)sw * sh * c);
            orc_gaussian_smoothing(src, foo, sw, sh, c, sigma, fsize);
            orc_resize_ratio(foo, data + offs[i], sw, sh, c, rate);
            free(foo);
        }
    }
    long total = offs[levels];
    free(offs);
    return total;
}

/* 5-point derivative taps {1,-8,0,8,-1}, each divided by 12: src/Image.h:987-992, :1030-1035. */
static void deriv_taps(double* d) {
    d[0] = 1;
    d[1] = -8;
    d[2] = 0;
    d[3] = 8;
    d[4] = -1;
    for (int i = 0; i < 5; i++) d[i] /= 12;
}

/* ---------------------------------------------------------------------------------------------
 * OpticalFlow::im2feature, src/OpticalFlow.cpp:911-961; desaturate src/Image.h:1461-1480 (RGB).
 * ------------------------------------------------------------------------------------------- */
int orc_im2feature(const double* im, int h, int w, int c, double* out) {
    const size_t np = (size_t)w * h;
    double d[5];
    deriv_taps(d);
    if (c == 1) {
        if (!out) return 3;
        double* gx = zalloc(np);
        double* gy = zalloc(np);
        orc_hfilter(im, gx, w, h, 1, d, 2);
        orc_vfilter(im, gy, w, h, 1, d, 2);
        for (size_t o = 0; o < np; o++) {
            out[o * 3] = im[o];
            out[o * 3 + 1] = gx[o];
            out[o * 3 + 2] = gy[o];
        }
        free(gx);
        free(gy);
        return 3;
    }
    if (c == 3) {
        if (!out) return 5;
        double* gray = zalloc(np);
        double* gx = zalloc(np);
        double* gy = zalloc(np);
        for (size_t o = 0; o < np; o++)
            gray[o] = im[o * 3] * .299 + im[o * 3 + 1] * .587 + im[o * 3 + 2] * .114;
        orc_hfilter(gray, gx, w, h, 1, d, 2);
        orc_vfilter(gray, gy, w, h, 1, d, 2);
        for (size_t o = 0; o < np; o++) {
            out[o * 5] = gray[o];
            out[o * 5 + 1] = gx[o];
            out[o * 5 + 2] = gy[o];
            out[o * 5 + 3] = im[o * 3 + 1] - im[o * 3];
            out[o * 5 + 4] = im[o * 3 + 1] - im[o * 3 + 2];
        }
        free(gray);
        free(gx);
        free(gy);
        return 5;
    }
    if (out) memcpy(out, im, sizeof(double) * np * c);
    return c;
}

/* ---------------------------------------------------------------------------------------------
 * OpticalFlow::warpFL -> ImageProcessing::warpImage, src/ImageProcessing.h:483-503.
 * ------------------------------------------------------------------------------------------- */
void orc_warpFL(const double* im1, const double* im2, const double* vx, const double* vy, int h, int w, int c,
                double* out) {
    memset(out, 0, sizeof(double) * (size_t)w * h * c);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const size_t o = (size_t)i * w + j;
            const double y = i + vy[o];
            const double x = j + vx[o];
            if (x < 0 || x > w - 1 || y < 0 || y > h - 1) {
                for (int k = 0; k < c; k++) out[o * c + k] = im1[o * c + k];
                continue;
            }
            bilinear_acc(im2, w, h, c, x, y, out + o * c);
        }
}

/* ---------------------------------------------------------------------------------------------
 * OpticalFlow::getDxs, src/OpticalFlow.cpp:80-122 (the `if(1)` branch).
 * ------------------------------------------------------------------------------------------- */
void orc_getDxs(const double* im1, const double* im2, int h, int w, int c, double* imdx, double* imdy,
                double* imdt) {
    const double g[5] = {0.02, 0.11, 0.74, 0.11, 0.02};
    const size_t n = (size_t)w * h * c;
    double d[5];
    deriv_taps(d);
    double* s1 = zalloc(n);
    double* s2 = zalloc(n);
    double* bl = zalloc(n);
    orc_filter_hv(im1, s1, w, h, c, g, 2, g, 2);
    orc_filter_hv(im2, s2, w, h, c, g, 2, g, 2);
    for (size_t i = 0; i < n; i++) {
        double t = s1[i];
        t *= 0.4;         /* Multiplywith(0.4), :92 */
        t += s2[i] * 0.6; /* Add(Im2,0.6), :93 ; src/Image.h:1905-1921 */
        bl[i] = t;
    }
    orc_hfilter(bl, imdx, w, h, c, d, 2);
    orc_vfilter(bl, imdy, w, h, c, d, 2);
    for (size_t i = 0; i < n; i++) imdt[i] = s2[i] - s1[i]; /* :97 */
    free(s1);
    free(s2);
    free(bl);
}

/* ---------------------------------------------------------------------------------------------
 * OpticalFlow::Laplacian, src/OpticalFlow.cpp:641-690.  The horizontal pass visits columns
 * 0..W-2 only and the vertical pass rows 0..H-2 only, so the last column gets no horizontal term
 * and the last row no vertical term (known-answer matrix in SURVEY.md §4).
 * ------------------------------------------------------------------------------------------- */
void orc_laplacian(const double* in, const double* weight, int h, int w, double* out) {
    const size_t np = (size_t)w * h;
    double* foo = zalloc(np);
    memset(out, 0, sizeof(double) * np);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w - 1; j++) {
            const size_t o = (size_t)i * w + j;
            foo[o] = (in[o + 1] - in[o]) * weight[o];
            out[o] -= foo[o];
            if (j > 0) out[o] += foo[o - 1];
        }
    memset(foo, 0, sizeof(double) * np);
    for (int i = 0; i < h - 1; i++)
        for (int j = 0; j < w; j++) {
            const size_t o = (size_t)i * w + j;
            foo[o] = (in[o + w] - in[o]) * weight[o];
            out[o] -= foo[o];
            if (i > 0) out[o] += foo[o - w];
        }
    free(foo);
}

/* ---------------------------------------------------------------------------------------------
 * Linear system of one inner fixed-point iteration, src/OpticalFlow.cpp:295-448:
 *   flow forward differences (src/Image.h:979-986, :1022-1029), phi (:325-331), psi (:377-406),
 *   per-channel products (psi*a)*b (src/Image.h:1747-1763) averaged over channels in order
 *   (src/Image.h:1537-1545), weighted Laplacian of u and v, right-hand sides (:444-448).
 * NOTE the reference differentiates uu=u+du for phi but applies Laplacian to u (:437-438).
 * ------------------------------------------------------------------------------------------- */
/* GaussianMixture::Gaussian, src/NoiseModel.h:120-126.  gm = alpha[c], sigma[c], beta[c], sigma_square[c],
 * beta_square[c] concatenated.  QUIRK: NoiseModel.h defines PI only `#ifndef PI` (:10-12), and in OpticalFlow.cpp it is
 * reached through OpticalFlow.h:6-7 AFTER Image.h -> Stochastic.h:19 `#define PI 3.1415927`: the mixture densities are
 * normalised with that 8-digit value (it cancels in the weights up to rounding -- which is what the golden vectors pin). */
#define ORC_PI 3.1415927
static inline double gm_gaussian(const double* gm, int c, double x, int i, int k) {
    const double *sigma = gm + c, *beta = gm + 2 * c, *sigma2 = gm + 3 * c, *beta2 = gm + 4 * c;
    if (i == 0) return exp(-x / (2 * sigma2[k])) / (2 * ORC_PI * sigma[k]);
    return exp(-x / (2 * beta2[k])) / (2 * ORC_PI * beta[k]);
}
/* GaussianMixture::reset, src/NoiseModel.h:97-107 (+ square(), :131-138) */
void orc_gm_reset(double* gm, int c) {
    for (int k = 0; k < c; k++) {
        gm[k] = 0.95;
        gm[c + k] = 0.05;
        gm[2 * c + k] = 0.5;
        gm[3 * c + k] = gm[c + k] * gm[c + k];
        gm[4 * c + k] = gm[2 * c + k] * gm[2 * c + k];
    }
}

static void linear_system_impl(const double* imdx, const double* imdy, const double* imdt, const double* u,
                               const double* v, const double* du, const double* dv, int h, int w, int c,
                               double alpha, const double* lappara, const double* gm, double* phi, double* imdxy,
                               double* imdx2, double* imdy2, double* imdtdx, double* imdtdy);

void orc_linear_system(const double* imdx, const double* imdy, const double* imdt, const double* u,
                       const double* v, const double* du, const double* dv, int h, int w, int c, double alpha,
                       const double* lappara, double* phi, double* imdxy, double* imdx2, double* imdy2,
                       double* imdtdx, double* imdtdy) {
    linear_system_impl(imdx, imdy, imdt, u, v, du, dv, h, w, c, alpha, lappara, NULL, phi, imdxy, imdx2, imdy2, imdtdx,
                       imdtdy);
}

/* gm != NULL: the Gaussian-mixture noise model (noiseModel == GMixture, src/OpticalFlow.cpp:359-367 / :389-397) */
void orc_linear_system_gm(const double* imdx, const double* imdy, const double* imdt, const double* u,
                          const double* v, const double* du, const double* dv, int h, int w, int c, double alpha,
                          const double* gm, double* phi, double* imdxy, double* imdx2, double* imdy2,
                          double* imdtdx, double* imdtdy) {
    linear_system_impl(imdx, imdy, imdt, u, v, du, dv, h, w, c, alpha, NULL, gm, phi, imdxy, imdx2, imdy2, imdtdx,
                       imdtdy);
}

static void linear_system_impl(const double* imdx, const double* imdy, const double* imdt, const double* u,
                               const double* v, const double* du, const double* dv, int h, int w, int c,
                               double alpha, const double* lappara, const double* gm, double* phi, double* imdxy,
                               double* imdx2, double* imdy2, double* imdtdx, double* imdtdy) {
    const size_t np = (size_t)w * h;
    const double eps_phi = 0.001 * 0.001, eps_psi = 0.001 * 0.001; /* pow(0.001,2), :261-262 */
    double* uu = zalloc(np);
    double* vv = zalloc(np);
    for (size_t o = 0; o < np; o++) {
        uu[o] = du ? u[o] + du[o] : u[o]; /* :297-303 (hh==0: copy; else Add(u,du)) */
        vv[o] = dv ? v[o] + dv[o] : v[o];
    }
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const size_t o = (size_t)i * w + j;
            const double ux = j < w - 1 ? uu[o + 1] - uu[o] : 0.0;
            const double uy = i < h - 1 ? uu[o + w] - uu[o] : 0.0;
            const double vx = j < w - 1 ? vv[o + 1] - vv[o] : 0.0;
            const double vy = i < h - 1 ? vv[o + w] - vv[o] : 0.0;
            const double t = ux * ux + uy * uy + vx * vx + vy * vy;
            phi[o] = 0.5 / sqrt(t + eps_phi);
        }
    free(uu);
    free(vv);
    for (size_t o = 0; o < np; o++) {
        double sxy = 0, sx2 = 0, sy2 = 0, stx = 0, sty = 0;
        const double duo = du ? du[o] : 0.0, dvo = dv ? dv[o] : 0.0;
        for (int k = 0; k < c; k++) {
            const size_t e = o * c + k;
            double t = imdt[e] + imdx[e] * duo + imdy[e] * dvo;
            t *= t;
            double psi = 0.0;
            if (gm) { /* :392-396 */
                const double prob1 = gm_gaussian(gm, c, t, 0, k) * gm[k];
                const double prob2 = gm_gaussian(gm, c, t, 1, k) * (1 - gm[k]);
                const double prob11 = prob1 / (2 * gm[3 * c + k]);
                const double prob22 = prob2 / (2 * gm[4 * c + k]);
                psi = (prob11 + prob22) / (prob1 + prob2);
            } else if (!(lappara[k] < 1E-20))
                psi = 1 / (2 * sqrt(t + eps_psi));
            if (c == 1) { /* :428-435: no collapse, plain copies */
                sxy = psi * imdx[e] * imdy[e];
                sx2 = psi * imdx[e] * imdx[e];
                sy2 = psi * imdy[e] * imdy[e];
                stx = psi * imdx[e] * imdt[e];
                sty = psi * imdy[e] * imdt[e];
            } else {
                sxy += psi * imdx[e] * imdy[e];
                sx2 += psi * imdx[e] * imdx[e];
                sy2 += psi * imdy[e] * imdy[e];
                stx += psi * imdx[e] * imdt[e];
                sty += psi * imdy[e] * imdt[e];
            }
        }
        if (c == 1) {
            imdxy[o] = sxy;
            imdx2[o] = sx2;
            imdy2[o] = sy2;
            imdtdx[o] = stx;
            imdtdy[o] = sty;
        } else {
            imdxy[o] = sxy / c;
            imdx2[o] = sx2 / c;
            imdy2[o] = sy2 / c;
            imdtdx[o] = stx / c;
            imdtdy[o] = sty / c;
        }
    }
    double* f1 = zalloc(np);
    double* f2 = zalloc(np);
    orc_laplacian(u, phi, h, w, f1);
    orc_laplacian(v, phi, h, w, f2);
    for (size_t o = 0; o < np; o++) {
        imdtdx[o] = -imdtdx[o] - alpha * f1[o];
        imdtdy[o] = -imdtdy[o] - alpha * f2[o];
    }
    free(f1);
    free(f2);
}

/* ---------------------------------------------------------------------------------------------
 * The coupled 5-point update of one cell, src/OpticalFlow.cpp:463-504.  `dur/dvr` are the arrays
 * the neighbour values are READ from, `duw/dvw` the ones written (same arrays for the in-place
 * modes, previous-sweep copies for Jacobi).
 * ------------------------------------------------------------------------------------------- */
static inline void sor_cell(int i, int j, int h, int w, const double* phi, const double* imdxy,
                            const double* imdx2, const double* imdy2, const double* imdtdx,
                            const double* imdtdy, const double* dur, const double* dvr, double* duw,
                            double* dvw, double alpha, double omega) {
    const size_t o = (size_t)i * w + j;
    double s1 = 0, s2 = 0, coeff = 0, wt;
    if (j > 0) {
        wt = phi[o - 1];
        s1 += wt * dur[o - 1];
        s2 += wt * dvr[o - 1];
        coeff += wt;
    }
    if (j < w - 1) {
        wt = phi[o];
        s1 += wt * dur[o + 1];
        s2 += wt * dvr[o + 1];
        coeff += wt;
    }
    if (i > 0) {
        wt = phi[o - w];
        s1 += wt * dur[o - w];
        s2 += wt * dvr[o - w];
        coeff += wt;
    }
    if (i < h - 1) {
        wt = phi[o];
        s1 += wt * dur[o + w];
        s2 += wt * dvr[o + w];
        coeff += wt;
    }
    s1 *= -alpha;
    s2 *= -alpha;
    coeff *= alpha;
    s1 += imdxy[o] * dvr[o];
    duw[o] = (1 - omega) * dur[o] + omega / (imdx2[o] + alpha * 0.05 + coeff) * (imdtdx[o] - s1);
    s2 += imdxy[o] * duw[o];
    dvw[o] = (1 - omega) * dvr[o] + omega / (imdy2[o] + alpha * 0.05 + coeff) * (imdtdy[o] - s2);
}

/* ORC_SOR_EXACT   : in place, sweep -> row -> column (the reference, :458-460).
 * ORC_SOR_REDBLACK: in place, per sweep all cells with (i+j) even, then all with (i+j) odd.
 * ORC_SOR_JACOBI  : per sweep every cell from the previous sweep's values (the in-cell du->dv
 *                   coupling of :503 is kept).  Throughput modes; NOT reference parity. */
void orc_sor(const double* phi, const double* imdxy, const double* imdx2, const double* imdy2,
             const double* imdtdx, const double* imdtdy, double* du, double* dv, int h, int w, double alpha,
             double omega, int n_sor, int mode) {
    if (mode == ORC_SOR_EXACT) {
        for (int k = 0; k < n_sor; k++)
            for (int i = 0; i < h; i++)
                for (int j = 0; j < w; j++)
                    sor_cell(i, j, h, w, phi, imdxy, imdx2, imdy2, imdtdx, imdtdy, du, dv, du, dv, alpha, omega);
    } else if (mode == ORC_SOR_REDBLACK) {
        for (int k = 0; k < n_sor; k++)
            for (int colour = 0; colour < 2; colour++)
                for (int i = 0; i < h; i++)
                    for (int j = (i + colour) & 1; j < w; j += 2)
                        sor_cell(i, j, h, w, phi, imdxy, imdx2, imdy2, imdtdx, imdtdy, du, dv, du, dv, alpha,
                                 omega);
    } else {
        const size_t np = (size_t)w * h;
        double* pu = zalloc(np);
        double* pv = zalloc(np);
        for (int k = 0; k < n_sor; k++) {
            memcpy(pu, du, sizeof(double) * np);
            memcpy(pv, dv, sizeof(double) * np);
            for (int i = 0; i < h; i++)
                for (int j = 0; j < w; j++)
                    sor_cell(i, j, h, w, phi, imdxy, imdx2, imdy2, imdtdx, imdtdy, pu, pv, du, dv, alpha, omega);
        }
        free(pu);
        free(pv);
    }
}

/* OpticalFlow::estLaplacianNoise, src/OpticalFlow.cpp:594-639. */
static void est_laplacian_noise(const double* im1, const double* im2, size_t np, int c, double* para) {
    double* total = zalloc((size_t)c);
    for (int k = 0; k < c; k++) para[k] = 0;
    for (size_t i = 0; i < np; i++)
        for (int k = 0; k < c; k++) {
            const double t = fabs(im1[i * c + k] - im2[i * c + k]);
            if (t > 0 && t < 1000000) {
                para[k] += t;
                total[k]++;
            }
        }
    for (int k = 0; k < c; k++) {
        if (total[k] == 0)
            para[k] = 0.001;
        else
            para[k] /= total[k];
    }
    free(total);
}

/* OpticalFlow::estGaussianMixture, src/OpticalFlow.cpp:539-591 (prior = 0.9, src/OpticalFlow.h:43): three EM iterations.
 * Quirk kept: the M step calls para.reset() (:564) and then ACCUMULATES onto the reset values sigma = 0.05, beta = 0.5
 * (:573-574) before dividing by the totals. */
void orc_est_gaussian_mixture(const double* im1, const double* im2, long np, int c, double* gm, double prior) {
    const size_t n = (size_t)np * c;
    double* w1 = zalloc(n);
    double* w2 = zalloc(n);
    double* total1 = zalloc((size_t)c);
    double* total2 = zalloc((size_t)c);
    for (int count = 0; count < 3; count++) {
        for (int k = 0; k < c; k++) total1[k] = total2[k] = 0;
        for (long i = 0; i < np; i++) /* E step */
            for (int k = 0; k < c; k++) {
                const size_t o = (size_t)i * c + k;
                double t = im1[o] - im2[o];
                t *= t;
                w1[o] = gm_gaussian(gm, c, t, 0, k) * gm[k];
                w2[o] = gm_gaussian(gm, c, t, 1, k) * (1 - gm[k]);
                t = w1[o] + w2[o];
                w1[o] /= t;
                w2[o] /= t;
                total1[k] += w1[o];
                total2[k] += w2[o];
            }
        orc_gm_reset(gm, c); /* M step */
        for (long i = 0; i < np; i++)
            for (int k = 0; k < c; k++) {
                const size_t o = (size_t)i * c + k;
                double t = im1[o] - im2[o];
                t *= t;
                gm[c + k] += w1[o] * t;
                gm[2 * c + k] += w2[o] * t;
            }
        for (int k = 0; k < c; k++) {
            gm[k] = total1[k] / (total1[k] + total2[k]) * (1 - prior) + 0.95 * prior;
            gm[c + k] = sqrt(gm[c + k] / total1[k]);
            gm[2 * c + k] = sqrt(gm[2 * c + k] / total2[k]) * (1 - prior) + 0.3 * prior;
        }
        for (int k = 0; k < c; k++) {
            gm[3 * c + k] = gm[c + k] * gm[c + k];
            gm[4 * c + k] = gm[2 * c + k] * gm[2 * c + k];
        }
    }
    free(w1);
    free(w2);
    free(total1);
    free(total2);
}

static void bicubic_warp_impl(const double* im1, const double* im2, const double* vx, const double* vy, int h, int w,
                              int c, double* out, int clamp);

/* ---------------------------------------------------------------------------------------------
 * OpticalFlow::SmoothFlowSOR, src/OpticalFlow.cpp:238-536.
 * phase_sec (may be NULL): [0]=Phase1 getDxs, [1]=Phase2..4 linear system, [2]=Phase5 SOR, [3]=Phase6.
 * genInImageMask (:278) is computed by the reference but never read: omitted (results-neutral).
 * ------------------------------------------------------------------------------------------- */
void orc_smoothflow_sor(const double* im1, const double* im2, double* warp, double* u, double* v, int h, int w,
                        int c, double alpha, int n_outer, int n_inner, int n_sor, double omega, int mode,
                        double* lappara, double* phase_sec) {
    orc_smoothflow_sor_ex(im1, im2, warp, u, v, h, w, c, alpha, n_outer, n_inner, n_sor, omega, mode, lappara, phase_sec,
                          ORC_INTERP_BILINEAR, NULL);
}

/* interpolation: how frame 2 is re-warped after each outer iteration (:515-521; Bicubic = warpImageBicubicRef +
 * threshold on the FEATURE images).  gm != NULL: Gaussian-mixture noise model (psi :359-367, estGaussianMixture :524-528);
 * gm is in/out (5 * c doubles, see orc_gm_reset). */
void orc_smoothflow_sor_ex(const double* im1, const double* im2, double* warp, double* u, double* v, int h, int w,
                           int c, double alpha, int n_outer, int n_inner, int n_sor, double omega, int mode,
                           double* lappara, double* phase_sec, int interpolation, double* gm) {
    const size_t np = (size_t)w * h, n = np * c;
    double* imdx = zalloc(n);
    double* imdy = zalloc(n);
    double* imdt = zalloc(n);
    double* du = zalloc(np);
    double* dv = zalloc(np);
    double* phi = zalloc(np);
    double* imdxy = zalloc(np);
    double* imdx2 = zalloc(np);
    double* imdy2 = zalloc(np);
    double* imdtdx = zalloc(np);
    double* imdtdy = zalloc(np);
    for (int count = 0; count < n_outer; count++) {
        double t0 = now_sec();
        orc_getDxs(im1, warp, h, w, c, imdx, imdy, imdt);
        double t1 = now_sec();
        if (phase_sec) phase_sec[0] += t1 - t0;
        memset(du, 0, sizeof(double) * np);
        memset(dv, 0, sizeof(double) * np);
        for (int hh = 0; hh < n_inner; hh++) {
            t0 = now_sec();
            linear_system_impl(imdx, imdy, imdt, u, v, hh == 0 ? NULL : du, hh == 0 ? NULL : dv, h, w, c, alpha,
                               lappara, gm, phi, imdxy, imdx2, imdy2, imdtdx, imdtdy);
            memset(du, 0, sizeof(double) * np); /* :452-453 */
            memset(dv, 0, sizeof(double) * np);
            t1 = now_sec();
            if (phase_sec) phase_sec[1] += t1 - t0;
            orc_sor(phi, imdxy, imdx2, imdy2, imdtdx, imdtdy, du, dv, h, w, alpha, omega, n_sor, mode);
            if (phase_sec) phase_sec[2] += now_sec() - t1;
        }
        t0 = now_sec();
        for (size_t o = 0; o < np; o++) { /* :513-514 */
            u[o] += du[o];
            v[o] += dv[o];
        }
        if (interpolation == ORC_INTERP_BILINEAR)
            orc_warpFL(im1, im2, u, v, h, w, c, warp); /* :516 */
        else
            bicubic_warp_impl(im1, im2, u, v, h, w, c, warp, 1); /* :519-520: warpImageBicubicRef + threshold */
        if (gm)
            orc_est_gaussian_mixture(im1, warp, (long)np, c, gm, 0.9); /* :527 */
        else
            est_laplacian_noise(im1, warp, np, c, lappara); /* :530 */
        if (phase_sec) phase_sec[3] += now_sec() - t0;
    }
    free(imdx);
    free(imdy);
    free(imdt);
    free(du);
    free(dv);
    free(phi);
    free(imdxy);
    free(imdx2);
    free(imdy2);
    free(imdtdx);
    free(imdtdy);
}

/* ---------------------------------------------------------------------------------------------
 * Final warp of the originals: Image::warpImageBicubicRef src/Image.h:2587-2595 + :2624-2701,
 * Hermite coefficients src/Image.h:2497-2530, clamp to [0,1] src/Image.h:2031-2045.
 * Corner naming: A=(x0,y0) B=(x1,y0) C=(x0,y1) D=(x1,y1); f value, fx/fy/fxy central differences.
 * cXY multiplies dx^X * dy^Y.  Term ORDER inside each sum follows the reference expressions.
 * ------------------------------------------------------------------------------------------- */
void orc_bicubic_warp(const double* im1, const double* im2, const double* vx, const double* vy, int h, int w,
                      int c, double* out) {
    bicubic_warp_impl(im1, im2, vx, vy, h, w, c, out, 1);
}
/* warpImageBicubicRef alone, without threshold(): the first warp of a level with interpolation == Bicubic
 * (src/OpticalFlow.cpp:816) */
void orc_bicubic_warp_noclamp(const double* im1, const double* im2, const double* vx, const double* vy, int h, int w,
                              int c, double* out) {
    bicubic_warp_impl(im1, im2, vx, vy, h, w, c, out, 0);
}
static void bicubic_warp_impl(const double* im1, const double* im2, const double* vx, const double* vy, int h, int w,
                              int c, double* out, int clamp) {
    const size_t n = (size_t)w * h * c;
    const double df[3] = {-0.5, 0, 0.5};
    double* gx = zalloc(n);
    double* gy = zalloc(n);
    double* gxy = zalloc(n);
    orc_hfilter(im2, gx, w, h, c, df, 1);
    orc_vfilter(im2, gy, w, h, c, df, 1);
    orc_vfilter(gx, gxy, w, h, c, df, 1);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const size_t o = (size_t)i * w + j;
            const double x = j + vx[o];
            const double y = i + vy[o];
            if (x < 0 || x > w - 1 || y < 0 || y > h - 1) {
                for (int k = 0; k < c; k++) out[o * c + k] = im1[o * c + k];
                continue;
            }
            int x0 = (int)x, y0 = (int)y;
            int x1 = x0 + 1, y1 = y0 + 1;
            x0 = clampi(x0, w);
            x1 = clampi(x1, w);
            y0 = clampi(y0, h);
            y1 = clampi(y1, h);
            const double dx = x - x0, dy = y - y0;
            const double dx2 = dx * dx, dy2 = dy * dy;
            const double dx3 = dx * dx2, dy3 = dy * dy2;
            for (int k = 0; k < c; k++) {
                const size_t a = ((size_t)y0 * w + x0) * c + k, b = ((size_t)y0 * w + x1) * c + k;
                const size_t cc = ((size_t)y1 * w + x0) * c + k, d = ((size_t)y1 * w + x1) * c + k;
                const double fA = im2[a], fB = im2[b], fC = im2[cc], fD = im2[d];
                const double xA = gx[a], xB = gx[b], xC = gx[cc], xD = gx[d];
                const double yA = gy[a], yB = gy[b], yC = gy[cc], yD = gy[d];
                const double zA = gxy[a], zB = gxy[b], zC = gxy[cc], zD = gxy[d];
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
                const double c22 = 9 * fA - 9 * fB - 9 * fC + 9 * fD + 6 * xA + 3 * xB - 6 * xC - 3 * xD +
                                   6 * yA - 6 * yB + 3 * yC - 3 * yD + 4 * zA + 2 * zB + 2 * zC + zD;
                const double c32 = -6 * fA + 6 * fB + 6 * fC - 6 * fD - 3 * xA - 3 * xB + 3 * xC + 3 * xD -
                                   4 * yA + 4 * yB - 2 * yC + 2 * yD - 2 * zA - 2 * zB - zC - zD;
                const double c03 = 2 * fA - 2 * fC + yA + yC;
                const double c13 = 2 * xA - 2 * xC + zA + zC;
                const double c23 = -6 * fA + 6 * fB + 6 * fC - 6 * fD - 4 * xA - 2 * xB + 4 * xC + 2 * xD -
                                   3 * yA + 3 * yB - 3 * yC + 3 * yD - 2 * zA - zB - 2 * zC - zD;
                const double c33 = 4 * fA - 4 * fB - 4 * fC + 4 * fD + 2 * xA + 2 * xB - 2 * xC - 2 * xD +
                                   2 * yA - 2 * yB + 2 * yC - 2 * yD + zA + zB + zC + zD;
                double r = c00 + c01 * dy + c02 * dy2 + c03 * dy3 + c10 * dx + c11 * dx * dy + c12 * dx * dy2 +
                           c13 * dx * dy3 + c20 * dx2 + c21 * dx2 * dy + c22 * dx2 * dy2 + c23 * dx2 * dy3 +
                           c30 * dx3 + c31 * dx3 * dy + c32 * dx3 * dy2 + c33 * dx3 * dy3;
                out[o * c + k] = r;
            }
        }
    for (size_t e = 0; clamp && e < n; e++) { /* threshold(): __min(__max(x,0),1) */
        double r = out[e];
        r = r < 0 ? 0.0 : r;
        r = r > 1 ? 1.0 : r;
        out[e] = r;
    }
    free(gx);
    free(gy);
    free(gxy);
}

/* OpticalFlow::SaveOpticalFlow (src/OpticalFlow.cpp:993-1003): clamp with the __max/__min templates of src/project.h:41-50
 * (they return the first argument's type: double), shift, scale, convert to unsigned short (truncation). */
void orc_flow_quantize16(const double* vx, const double* vy, int h, int w, unsigned short* q) {
    const long n = (long)h * w;
    for (long i = 0; i < n; i++) {
        double a = vx[i], b = vy[i];
        a = (a < -200) ? -200.0 : a; /* __max(f, -200) */
        a = (a > 200) ? 200.0 : a;   /* __min(.., 200)  */
        b = (b < -200) ? -200.0 : b;
        b = (b > 200) ? 200.0 : b;
        q[i * 2] = (unsigned short)((a + 200) * 160);
        q[i * 2 + 1] = (unsigned short)((b + 200) * 160);
    }
}

/* OpticalFlow::LoadOpticalFlow (src/OpticalFlow.cpp:963-976) + DissembleFlow (src/OpticalFlow.h:80-91). */
void orc_flow_dequantize16(const unsigned short* q, int h, int w, double* vx, double* vy) {
    const long n = (long)h * w;
    for (long i = 0; i < n; i++) {
        vx[i] = (double)q[i * 2] / 160 - 200;
        vy[i] = (double)q[i * 2 + 1] / 160 - 200;
    }
}


/* ---------------------------------------------------------------------------------------------
 * OpticalFlow::Coarse2FineFlow, src/OpticalFlow.cpp:735-903.
 * ------------------------------------------------------------------------------------------- */
int orc_coarse2fine_flow(const double* im1, const double* im2, int h, int w, int c, int levels,
                         const orc_params* pp, double* vx, double* vy, double* warpI2, double* timing) {
    orc_params P;
    if (pp)
        P = *pp;
    else
        orc_default_params(&P);
    if (!im1 || !im2 || !vx || !vy || !warpI2 || h < 1 || w < 1 || c < 1 || levels < 1) return -1;
    double tm[10];
    memset(tm, 0, sizeof tm);
    const double t_total = now_sec();
    double ratio = P.ratio;
    if (ratio > 0.98 || ratio < 0.4) ratio = 0.75; /* applied inside the pyramid, src/GaussianPyramid.cpp:82 */

    /* Construction: two pyramids + LapPara (:743-778) */
    double t0 = now_sec();
    int* dims = (int*)calloc((size_t)2 * levels, sizeof(int));
    long total = orc_pyramid(im1, h, w, c, P.ratio, levels, dims, NULL);
    double* p1 = zalloc((size_t)total);
    double* p2 = zalloc((size_t)total);
    orc_pyramid(im1, h, w, c, P.ratio, levels, dims, p1);
    orc_pyramid(im2, h, w, c, P.ratio, levels, dims, p2);
    long* offs = (long*)calloc((size_t)levels + 1, sizeof(long));
    for (int i = 0; i < levels; i++) offs[i + 1] = offs[i] + (long)dims[2 * i] * dims[2 * i + 1] * c;
    const int fc = orc_im2feature(NULL, 1, 1, c, NULL);
    double* lappara = zalloc((size_t)(c + 2 > fc ? c + 2 : fc));
    for (int i = 0; i < c + 2; i++) lappara[i] = 0.02; /* :773-775 */
    double* gm = NULL;
    if (P.noise_model == ORC_NOISE_GMIXTURE) { /* :769-771: GMPara.reset(Im1.nchannels() + 2) */
        gm = zalloc((size_t)5 * (c + 2 > fc ? c + 2 : fc));
        orc_gm_reset(gm, fc);
    }
    tm[1] = now_sec() - t0;

    double *u = NULL, *v = NULL;
    int pw = 0, ph = 0;
    double phase[4] = {0, 0, 0, 0};
    for (int k = levels - 1; k >= 0; k--) {
        t0 = now_sec();
        const int lw = dims[2 * k], lh = dims[2 * k + 1];
        const size_t np = (size_t)lw * lh;
        double* f1 = zalloc(np * fc);
        double* f2 = zalloc(np * fc);
        double* warp = zalloc(np * fc);
        orc_im2feature(p1 + offs[k], lh, lw, c, f1); /* :797-798 */
        orc_im2feature(p2 + offs[k], lh, lw, c, f2);
        if (k == levels - 1) { /* :801-806 */
            u = zalloc(np);
            v = zalloc(np);
            memcpy(warp, f2, sizeof(double) * np * fc);
        } else { /* :809-814 */
            double* nu = zalloc(np);
            double* nv = zalloc(np);
            orc_resize_wh(u, nu, pw, ph, 1, lw, lh);
            orc_resize_wh(v, nv, pw, ph, 1, lw, lh);
            const double inv = 1 / ratio;
            for (size_t o = 0; o < np; o++) {
                nu[o] *= inv;
                nv[o] *= inv;
            }
            free(u);
            free(v);
            u = nu;
            v = nv;
            if (P.interpolation == ORC_INTERP_BILINEAR)
                orc_warpFL(f1, f2, u, v, lh, lw, fc, warp);
            else
                bicubic_warp_impl(f1, f2, u, v, lh, lw, fc, warp, 0); /* :816: no threshold here */
        }
        tm[0] += now_sec() - t0;
        orc_smoothflow_sor_ex(f1, f2, warp, u, v, lh, lw, fc, P.alpha, P.n_outer + k * P.n_outer_per_level,
                              P.n_inner, P.n_sor + k * P.n_sor_per_level, P.omega, P.sor_mode, lappara, phase,
                              P.interpolation, gm);
        pw = lw;
        ph = lh;
        free(f1);
        free(f2);
        free(warp);
    }
    t0 = now_sec();
    memcpy(vx, u, sizeof(double) * (size_t)w * h);
    memcpy(vy, v, sizeof(double) * (size_t)w * h);
    orc_bicubic_warp(im1, im2, u, v, h, w, c, warpI2); /* :841-842 */
    tm[8] = now_sec() - t0;
    tm[2] = phase[0];
    tm[5] = phase[1]; /* Phase2+3+4 are reported together under Phase4_LinearSystem */
    tm[6] = phase[2];
    tm[7] = phase[3];
    tm[9] = now_sec() - t_total;
    if (timing) memcpy(timing, tm, sizeof tm);
    free(u);
    free(v);
    free(p1);
    free(p2);
    free(dims);
    free(offs);
    free(lappara);
    free(gm);
    return 0;
}
