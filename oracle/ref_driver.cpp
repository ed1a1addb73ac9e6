// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// A thin extern "C" driver over the *untouched* reference Serial C++ sources.  It is compiled
// by oracle/Makefile against the sources where they lie under /root/reference/Code/Serial/src
// (nothing from the reference is copied into this repository) into oracle/_ref/libpapof_ref.so.
// It exists only in the build container: it pins the CPU restatement (papof_oracle.c) and
// generates the golden vectors in tests/golden/ (see tests/golden/make_golden.py).
//
// All images cross this boundary in the reference's own layout: row-major, channel-interleaved
// (HWC) IEEE fp64, exactly what DImage::pData holds (src/Image.h:36-56).
//
// Every entry point below calls a public static of the reference:
//   OpticalFlow::Coarse2FineFlow   src/OpticalFlow.cpp:735-903
//   OpticalFlow::SmoothFlowSOR     src/OpticalFlow.cpp:238-536
//   OpticalFlow::getDxs            src/OpticalFlow.cpp:80-122
//   OpticalFlow::warpFL            src/OpticalFlow.cpp:154-159
//   OpticalFlow::Laplacian         src/OpticalFlow.cpp:641-690
//   OpticalFlow::im2feature        src/OpticalFlow.cpp:911-961
//   GaussianPyramid::ConstructPyramidLevels  src/GaussianPyramid.cpp:79-108
//   Image<T>::imresize / GaussianSmoothing / warpImageBicubicRef / threshold  src/Image.h
#include "OpticalFlow.h"
#include "GaussianPyramid.h"
#include <cstring>
#include <map>
#include <string>
#include <cstdio>
#include <unistd.h>
#include <fcntl.h>

namespace {

void load(DImage& d, const double* src, int h, int w, int c, bool rgb = false) {
    d.allocate(w, h, c);
    std::memcpy(d.pData, src, sizeof(double) * (size_t)h * w * c);
    if (rgb) d.setColorType(0);  // RGB, as src/Coarse2FineFlowWrapper.cpp:21-28 does
}
void store(double* dst, const DImage& d) {
    std::memcpy(dst, d.data(), sizeof(double) * (size_t)d.width() * d.height() * d.nchannels());
}
void set_lappara(int n) {
    // src/OpticalFlow.cpp:773-775
    OpticalFlow::LapPara.allocate(n);
    for (int i = 0; i < OpticalFlow::LapPara.dim(); i++) OpticalFlow::LapPara[i] = 0.02;
}

// The reference prints progress on stdout (src/OpticalFlow.cpp:787-788,832-834); silence it.
struct Quiet {
    int saved;
    Quiet() {
        fflush(stdout);
        std::cout.flush();
        saved = dup(1);
        int nul = open("/dev/null", O_WRONLY);
        dup2(nul, 1);
        close(nul);
    }
    ~Quiet() {
        fflush(stdout);
        std::cout.flush();
        dup2(saved, 1);
        close(saved);
    }
};

const char* kKeys[10] = {"Allocation",         "Construction",  "Phase1_Generate", "Phase2_Derivatives",
                         "Phase3_PsiData",     "Phase4_LinearSystem", "Phase5_SOR", "Phase6_Update",
                         "PostProcessing",     "Total C++ Execution"};
}  // namespace

extern "C" {

// Full reference call.  timing[10] follows the std::map key order of src/OpticalFlow.cpp:850-860.
int ref_coarse2fine_flow(const double* im1, const double* im2, int h, int w, int c, int levels, double* vx,
                         double* vy, double* warpI2, double* timing) {
    Quiet q;
    DImage I1, I2, VX, VY, W2;
    load(I1, im1, h, w, c, true);
    load(I2, im2, h, w, c, true);
    std::map<std::string, std::string> t;
    OpticalFlow::Coarse2FineFlow(&t, VX, VY, W2, I1, I2, levels);
    store(vx, VX);
    store(vy, VY);
    store(warpI2, W2);
    if (timing)
        for (int i = 0; i < 10; i++) timing[i] = t.count(kKeys[i]) ? atof(t[kKeys[i]].c_str()) : -1.0;
    return 0;
}

// Same level loop as src/OpticalFlow.cpp:784-842 but with the iteration schedule as arguments
// (n_outer + k*outer_step, n_sor + k*sor_step at level k), composed from the reference's own
// public statics.  With (7,1,1,30,3) it must reproduce ref_coarse2fine_flow bit for bit
// (asserted by tests/test_oracle_golden.py::test_recomposed_level_loop_equals_the_reference_entry_point,
// levels 1, 3, 5); config-4's "3 outer / 30 SOR" is (3,0,1,30,0).
int ref_coarse2fine_flow_sched(const double* im1, const double* im2, int h, int w, int c, int levels,
                               double alpha, double ratio, int n_outer, int outer_step, int n_inner,
                               int n_sor, int sor_step, double* vx, double* vy, double* warpI2) {
    Quiet q;
    DImage Im1, Im2;
    load(Im1, im1, h, w, c, true);
    load(Im2, im2, h, w, c, true);
    GaussianPyramid P1, P2;
    P1.ConstructPyramidLevels(Im1, ratio, levels);
    P2.ConstructPyramidLevels(Im2, ratio, levels);
    set_lappara(c + 2);
    DImage Image1, Image2, WarpImage2, VX, VY, W2;
    for (int k = P1.nlevels() - 1; k >= 0; k--) {
        int width = P1.Image(k).width(), height = P1.Image(k).height();
        OpticalFlow::im2feature(Image1, P1.Image(k));
        OpticalFlow::im2feature(Image2, P2.Image(k));
        if (k == P1.nlevels() - 1) {
            VX.allocate(width, height);
            VY.allocate(width, height);
            WarpImage2.copyData(Image2);
        } else {
            VX.imresize(width, height);
            VX.Multiplywith(1 / ratio);
            VY.imresize(width, height);
            VY.Multiplywith(1 / ratio);
            OpticalFlow::warpFL(WarpImage2, Image1, Image2, VX, VY);
        }
        OpticalFlow::SmoothFlowSOR(Image1, Image2, WarpImage2, VX, VY, alpha, n_outer + k * outer_step, n_inner,
                                   n_sor + k * sor_step);
    }
    Im2.warpImageBicubicRef(Im1, W2, VX, VY);
    W2.threshold();
    store(vx, VX);
    store(vy, VY);
    store(warpI2, W2);
    return 0;
}

// Pyramid: dims[2*i] = width, dims[2*i+1] = height; data = levels concatenated (HWC).
// Call with data == NULL first to get dims.
int ref_pyramid(const double* im, int h, int w, int c, double ratio, int levels, int* dims, double* data) {
    DImage I;
    load(I, im, h, w, c, true);
    GaussianPyramid P;
    P.ConstructPyramidLevels(I, ratio, levels);
    size_t off = 0;
    for (int i = 0; i < P.nlevels(); i++) {
        dims[2 * i] = P.Image(i).width();
        dims[2 * i + 1] = P.Image(i).height();
        size_t n = (size_t)P.Image(i).width() * P.Image(i).height() * c;
        if (data) std::memcpy(data + off, P.Image(i).data(), n * sizeof(double));
        off += n;
    }
    return P.nlevels();
}

int ref_gaussian_smoothing(const double* im, int h, int w, int c, double sigma, int fsize, double* out) {
    DImage I, O;
    load(I, im, h, w, c);
    I.GaussianSmoothing(O, sigma, fsize);
    store(out, O);
    return 0;
}

int ref_resize_ratio(const double* im, int h, int w, int c, double ratio, int* dw, int* dh, double* out) {
    DImage I, O;
    load(I, im, h, w, c);
    I.imresize(O, ratio);
    *dw = O.width();
    *dh = O.height();
    if (out) store(out, O);
    return 0;
}

int ref_resize_wh(const double* im, int h, int w, int c, int dw, int dh, double* out) {
    DImage I;
    load(I, im, h, w, c);
    I.imresize(dw, dh);
    store(out, I);
    return 0;
}

// returns number of feature channels
int ref_im2feature(const double* im, int h, int w, int c, double* out) {
    DImage I, F;
    load(I, im, h, w, c, true);
    OpticalFlow::im2feature(F, I);
    if (out) store(out, F);
    return F.nchannels();
}

int ref_warpFL(const double* im1, const double* im2, const double* vx, const double* vy, int h, int w, int c,
               double* out) {
    DImage I1, I2, VX, VY, O;
    load(I1, im1, h, w, c);
    load(I2, im2, h, w, c);
    load(VX, vx, h, w, 1);
    load(VY, vy, h, w, 1);
    OpticalFlow::warpFL(O, I1, I2, VX, VY);
    store(out, O);
    return 0;
}

int ref_getDxs(const double* im1, const double* im2, int h, int w, int c, double* imdx, double* imdy,
               double* imdt) {
    DImage I1, I2, DX, DY, DT;
    load(I1, im1, h, w, c);
    load(I2, im2, h, w, c);
    OpticalFlow::getDxs(DX, DY, DT, I1, I2);
    store(imdx, DX);
    store(imdy, DY);
    store(imdt, DT);
    return 0;
}

int ref_laplacian(const double* in, const double* weight, int h, int w, double* out) {
    DImage I, Wt, O;
    load(I, in, h, w, 1);
    load(Wt, weight, h, w, 1);
    OpticalFlow::Laplacian(O, I, Wt);
    store(out, O);
    return 0;
}

// One SmoothFlowSOR call (the whole IRLS loop of one pyramid level).  warp, u, v are in/out.
int ref_smoothflow_sor(const double* im1, const double* im2, double* warp, double* u, double* v, int h, int w,
                       int c, double alpha, int n_outer, int n_inner, int n_sor) {
    Quiet q;
    DImage I1, I2, W2, U, V;
    load(I1, im1, h, w, c);
    load(I2, im2, h, w, c);
    load(W2, warp, h, w, c);
    load(U, u, h, w, 1);
    load(V, v, h, w, 1);
    set_lappara(c);
    OpticalFlow::SmoothFlowSOR(I1, I2, W2, U, V, alpha, n_outer, n_inner, n_sor);
    store(warp, W2);
    store(u, U);
    store(v, V);
    return 0;
}

// Final warp of the RGB originals: src/OpticalFlow.cpp:841-842.
int ref_bicubic_warp(const double* im1, const double* im2, const double* vx, const double* vy, int h, int w, int c,
                     double* out) {
    DImage I1, I2, VX, VY, O;
    load(I1, im1, h, w, c, true);
    load(I2, im2, h, w, c, true);
    load(VX, vx, h, w, 1);
    load(VY, vy, h, w, 1);
    I2.warpImageBicubicRef(I1, O, VX, VY);
    O.threshold();
    store(out, O);
    return 0;
}

// The reference's own 16-bit flow file: AssembleFlow (src/OpticalFlow.h:70-79) + SaveOpticalFlow (src/OpticalFlow.cpp:993-1003),
// and back: LoadOpticalFlow (:963-976) + DissembleFlow (src/OpticalFlow.h:80-91).
int ref_flow_save16(const double* vx, const double* vy, int h, int w, const char* path) {
    DImage x(w, h), y(w, h), flow;
    std::memcpy(x.data(), vx, sizeof(double) * (size_t)h * w);
    std::memcpy(y.data(), vy, sizeof(double) * (size_t)h * w);
    OpticalFlow::AssembleFlow(x, y, flow);
    return OpticalFlow::SaveOpticalFlow(flow, path) ? 0 : -1;
}

int ref_flow_load16(const char* path, int h, int w, double* vx, double* vy) {
    DImage flow, x, y;
    if (!OpticalFlow::LoadOpticalFlow(path, flow)) return -1;
    if (flow.width() != w || flow.height() != h || flow.nchannels() != 2) return -2;
    OpticalFlow::DissembleFlow(flow, x, y);
    std::memcpy(vx, x.data(), sizeof(double) * (size_t)h * w);
    std::memcpy(vy, y.data(), sizeof(double) * (size_t)h * w);
    return 0;
}



// ---- non-default branches of the reference: selected through its public statics
// OpticalFlow::interpolation / noiseModel / GMPara (src/OpticalFlow.h:20-26), always restored afterwards ----
namespace {
struct Branches {  // interpolation: 0 bilinear, 1 bicubic;  noise_model: 0 Laplacian (the default), 1 Gaussian mixture
    OpticalFlow::InterpolationMethod i0;
    OpticalFlow::NoiseModel n0;
    Branches(int interpolation, int noise_model) : i0(OpticalFlow::interpolation), n0(OpticalFlow::noiseModel) {
        OpticalFlow::interpolation = interpolation ? OpticalFlow::Bicubic : OpticalFlow::Bilinear;
        OpticalFlow::noiseModel = noise_model ? OpticalFlow::GMixture : OpticalFlow::Lap;
    }
    ~Branches() {
        OpticalFlow::interpolation = i0;
        OpticalFlow::noiseModel = n0;
    }
};
void gm_out(double* gm, int c) {  // alpha[c], sigma[c], beta[c], sigma_square[c], beta_square[c]
    const GaussianMixture& G = OpticalFlow::GMPara;
    for (int k = 0; k < c; k++) {
        gm[k] = G.alpha[k];
        gm[c + k] = G.sigma[k];
        gm[2 * c + k] = G.beta[k];
        gm[3 * c + k] = G.sigma_square[k];
        gm[4 * c + k] = G.beta_square[k];
    }
}
void gm_in(const double* gm, int c) {
    OpticalFlow::GMPara.reset(c);
    GaussianMixture& G = OpticalFlow::GMPara;
    for (int k = 0; k < c; k++) {
        G.alpha[k] = gm[k];
        G.sigma[k] = gm[c + k];
        G.beta[k] = gm[2 * c + k];
    }
    G.square();
}
}  // namespace

// OpticalFlow::Coarse2FineFlow itself (src/OpticalFlow.cpp:735-903) with the two statics set.
int ref_coarse2fine_flow_opts(const double* im1, const double* im2, int h, int w, int c, int levels,
                              int interpolation, int noise_model, double* vx, double* vy, double* warpI2) {
    Quiet q;
    Branches b(interpolation, noise_model);
    DImage I1, I2, VX, VY, W2;
    load(I1, im1, h, w, c, true);
    load(I2, im2, h, w, c, true);
    std::map<std::string, std::string> t;
    OpticalFlow::Coarse2FineFlow(&t, VX, VY, W2, I1, I2, levels);
    store(vx, VX);
    store(vy, VY);
    store(warpI2, W2);
    return 0;
}

// One SmoothFlowSOR call with the branches selected; gm (5 * c doubles) is in/out when noise_model == 1.
int ref_smoothflow_sor_opts(const double* im1, const double* im2, double* warp, double* u, double* v, int h, int w,
                            int c, double alpha, int n_outer, int n_inner, int n_sor, int interpolation,
                            int noise_model, double* gm) {
    Quiet q;
    Branches b(interpolation, noise_model);
    DImage I1, I2, W2, U, V;
    load(I1, im1, h, w, c);
    load(I2, im2, h, w, c);
    load(W2, warp, h, w, c);
    load(U, u, h, w, 1);
    load(V, v, h, w, 1);
    set_lappara(c);
    if (noise_model && gm) gm_in(gm, c);
    OpticalFlow::SmoothFlowSOR(I1, I2, W2, U, V, alpha, n_outer, n_inner, n_sor);
    if (noise_model && gm) gm_out(gm, c);
    store(warp, W2);
    store(u, U);
    store(v, V);
    return 0;
}

// OpticalFlow::estGaussianMixture (src/OpticalFlow.cpp:539-591), prior = its default 0.9; gm in/out.
int ref_est_gaussian_mixture(const double* im1, const double* im2, int h, int w, int c, double* gm) {
    DImage I1, I2;
    load(I1, im1, h, w, c);
    load(I2, im2, h, w, c);
    gm_in(gm, c);
    OpticalFlow::estGaussianMixture(I1, I2, OpticalFlow::GMPara);
    gm_out(gm, c);
    return 0;
}

// GaussianPyramid::ConstructPyramid (the min-width variant, src/GaussianPyramid.cpp:47-77); returns the level count.
// dims must hold 2 * 64 ints; call with data == NULL first.
int ref_pyramid_minwidth(const double* im, int h, int w, int c, double ratio, int min_width, int* dims, double* data) {
    DImage I;
    load(I, im, h, w, c, true);
    GaussianPyramid P;
    P.ConstructPyramid(I, ratio, min_width);
    size_t off = 0;
    for (int i = 0; i < P.nlevels() && i < 64; i++) {
        dims[2 * i] = P.Image(i).width();
        dims[2 * i + 1] = P.Image(i).height();
        size_t n = (size_t)P.Image(i).width() * P.Image(i).height() * c;
        if (data) std::memcpy(data + off, P.Image(i).data(), n * sizeof(double));
        off += n;
    }
    return P.nlevels();
}

// Image::warpImageBicubicRef WITHOUT threshold(): the first warp of a level when interpolation == Bicubic (:816).
int ref_bicubic_warp_noclamp(const double* im1, const double* im2, const double* vx, const double* vy, int h, int w,
                             int c, double* out) {
    DImage I1, I2, VX, VY, O;
    load(I1, im1, h, w, c);
    load(I2, im2, h, w, c);
    load(VX, vx, h, w, 1);
    load(VY, vy, h, w, 1);
    I2.warpImageBicubicRef(I1, O, VX, VY);
    store(out, O);
    return 0;
}

}  // extern "C"
