/* include/papof.h -- C ABI of the MI355X-native coarse-to-fine optical-flow hot path.
 *
 * Drop-in boundary.  This library replaces everything beneath the reference's only FFI seam:
 *
 *   Code/Serial/coarse2Fine.pxd:8-12 and Code/Serial/src/Coarse2FineFlowWrapper.h:12-15
 *       map<string,string> Coarse2FineFlowWrapper(double* vx, double* vy, double* warpI2,
 *                                                 const double* Im1, const double* Im2,
 *                                                 int pyramidLevels, int h, int w, int c);
 *   (Code/Parallel/coarse2Fine.pxd:11 inserts `int nCores` after pyramidLevels)
 *
 * which Code/Serial/pyflow.pyx:61-66 calls with caller-allocated, C-contiguous float64 buffers
 * (Im1, Im2, warpI2: h*w*c, channel-interleaved; vx, vy: h*w).  The C++ std::map return value cannot
 * cross a C ABI, so the ten timers the reference publishes (src/OpticalFlow.cpp:850-860) come back in
 * `timing_sec[10]`, in the map's sorted key order (PAPOF_TIMING_KEYS); the Python/Cython side formats
 * them into the dict of strings pyflow.coarse2fine_flow returns.
 *
 * Conventions: plain pointers and sizes only; no exceptions cross the boundary; every function returns
 * PAPOF_OK (0) or a negative PAPOF_E* code; the caller owns every buffer; the library retains nothing
 * between calls except the device arena inside a papof_handle.  The compute path is HIP on gfx950 only:
 * there is NO CPU fallback -- without a usable GPU every compute entry point returns PAPOF_ENODEVICE.
 */
#ifndef PAPOF_H
#define PAPOF_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAPOF_VERSION 105 /* 0.1.5: papof_flow_batch / papof_flow_batch_u8, papof_last_host_times, PAPOF_RCCL_LIB / PAPOF_TILES_TIMEOUT_S; 0.1.4: the Laplacian-noise guard (papof_lap_guard_stats); 0.1.3: papof_last_sor_solves, exact-order band split over ranks (papof_tiles_*, PAPOF_SOR_EXACT); 0.1.2: measurement / test aids (papof_last_sor_stats, papof_strip_plan, papof_test_sor_strips); 0.1.1: papof_params gained interpolation / noise_model */

enum {
    PAPOF_OK = 0,
    PAPOF_EINVAL = -1,     /* NULL pointer, non-positive size, pyramid_levels < 1 (UB in the reference:
                              src/GaussianPyramid.cpp:87-88), unsupported parameter */
    PAPOF_ENODEVICE = -2,  /* no gfx950 device / HIP runtime error at start-up */
    PAPOF_ENOMEM = -3,     /* device or host allocation failed */
    PAPOF_EDEVICE = -4,    /* a HIP call failed mid-flight (papof_last_error() has the text) */
    PAPOF_ETIMEOUT = -5    /* a bounded device-side wait expired (exact-order SOR progress counters) */
};

/* Order of the SOR sweep (src/OpticalFlow.cpp:458-505). */
enum {
    PAPOF_SOR_EXACT = 0,    /* the reference's in-place sweep->row->column order, bit-compatible results */
    PAPOF_SOR_REDBLACK = 1, /* in-place two-colour sweeps: throughput mode, NOT reference parity      */
    PAPOF_SOR_JACOBI = 2    /* every cell from the previous sweep: correctness-gate mode (config 2)   */
};

/* The reference's two non-default branches, selected there by file-scope statics that its Python entry point cannot
 * reach (src/OpticalFlow.cpp:32-34). */
enum {
    PAPOF_INTERP_BILINEAR = 0, /* warpFL (the default, src/OpticalFlow.cpp:33)                                     */
    PAPOF_INTERP_BICUBIC = 1   /* in-loop Image::warpImageBicubicRef (+ threshold) on the feature images, :517-521, :816 */
};
enum {
    PAPOF_NOISE_LAPLACIAN = 0, /* psi = 1 / (2 sqrt(t^2 + eps)) (the default, :34, :399-402)                        */
    PAPOF_NOISE_GMIXTURE = 1   /* two-component Gaussian mixture per channel (:359-367, :389-397), re-estimated by EM
                                  after every outer iteration (estGaussianMixture, :539-591).  Uses exp() and global
                                  sums: NOT bit-compatible with the reference (device exp <= 1 ulp, parallel sums), and
                                  with this model the reference's own iteration amplifies a 1e-13 perturbation ~7x per
                                  outer iteration (measured on its arithmetic): one SmoothFlowSOR call of 3 outer
                                  iterations agrees to 1e-9, whole calls only within that conditioning  */
};

/* Solver parameters.  The reference hard-codes all of them (src/OpticalFlow.cpp:747-751, :451, :823);
 * papof_default_params() reproduces those values, so passing NULL == the reference. */
typedef struct papof_params {
    double alpha;          /* 0.012  regularisation weight                      src/OpticalFlow.cpp:747 */
    double ratio;          /* 0.75   pyramid down-sampling ratio                :748                    */
    int n_outer;           /* 7      outer fixed-point iterations at level 0    :749                    */
    int n_outer_per_level; /* 1      ... plus this many per pyramid level k     :823 (nOuter+k)         */
    int n_inner;           /* 1      inner fixed-point iterations               :750                    */
    int n_sor;             /* 30     SOR sweeps at level 0                      :751                    */
    int n_sor_per_level;   /* 3      ... plus this many per pyramid level k     :823 (nCG+k*3)          */
    double omega;          /* 1.8    over-relaxation factor                     :451                    */
    int sor_mode;          /* PAPOF_SOR_*                                                                */
    int phase_timing;      /* All ten reference timers are always measured (HIP events on the streams, no synchronisation).
                              0: flow-independent work (pyramids = Construction, features = Allocation, derivative planes of
                                 the final warp = PostProcessing) runs on a second stream BESIDE the solver phases, so the
                                 ten values overlap in time and add up to more than the total;
                              1: everything on one stream: phases do not overlap (slower by the lost overlap);
                              2: as 0, but only "Total C++ Execution" and "Phase5_SOR" are measured (the other eight read 0):
                                 no phase stamps at all -- what bench.py's device-resident headline runs with, as in round 1
                                 (the stamps cost <= 0.1 ms per 1080p call).
                              Phase5_SOR is the solver kernels' own duration in both cases.  Phase3_PsiData and
                              Phase4_LinearSystem are ONE fused kernel here: its time is apportioned 30 : 70; on the default
                              branches phi (Phase2_Derivatives) is written by the warp-and-smooth kernel of Phase1_Generate
                              and reported as a fixed 4 % of it; where ONE kernel does Phase1 ... Phase4 (k_flow_system: default
                              branches, one GPU) its time is apportioned 51 : 2 : 14 : 33.                              */
    int interpolation;     /* PAPOF_INTERP_*  (0 = the reference's default)                                          */
    int noise_model;       /* PAPOF_NOISE_*   (0 = the reference's default)                                          */
} papof_params;

/* Index of each reference timer in timing_sec[] == sorted std::map key order, src/OpticalFlow.cpp:850-860 */
enum {
    PAPOF_T_ALLOCATION = 0,
    PAPOF_T_CONSTRUCTION = 1,
    PAPOF_T_PHASE1_GENERATE = 2,
    PAPOF_T_PHASE2_DERIVATIVES = 3,
    PAPOF_T_PHASE3_PSIDATA = 4,
    PAPOF_T_PHASE4_LINEARSYSTEM = 5,
    PAPOF_T_PHASE5_SOR = 6,
    PAPOF_T_PHASE6_UPDATE = 7,
    PAPOF_T_POSTPROCESSING = 8,
    PAPOF_T_TOTAL = 9,
    PAPOF_N_TIMERS = 10
};

typedef struct papof_handle papof_handle; /* device arena + stream; one per GPU, not thread-safe */

int papof_version(void);
void papof_default_params(papof_params* p);
const char* papof_strerror(int code);
const char* papof_last_error(void); /* text of the last HIP failure on this thread ("" if none) */
const char* papof_timing_key(int index); /* "Allocation" ... "Total C++ Execution" */
int papof_device_count(void);            /* number of visible gfx950 devices (0 if none / no driver) */

int papof_create(int device, papof_handle** out);
void papof_destroy(papof_handle* h);

/* ---- THE drop-in entry point: replaces Coarse2FineFlowWrapper (src/Coarse2FineFlowWrapper.cpp:14-51).
 * Host buffers in, host buffers out, uses a process-wide lazily created handle on device 0
 * (or $PAPOF_DEVICE).  Internally serialised by a mutex (the reference is not re-entrant either:
 * file-scope timers, src/OpticalFlow.cpp:39-64). */
int papof_coarse2fine_flow(const double* im1, const double* im2, int h, int w, int c, int pyramid_levels,
                           const papof_params* params /* NULL = reference defaults */, double* vx, double* vy,
                           double* warpI2, double timing_sec[PAPOF_N_TIMERS] /* may be NULL */);

/* Same, on an explicit handle (arena reuse across the 101 pairs of a collection, TestSuite.py:69-81). */
int papof_flow(papof_handle* h, const double* im1, const double* im2, int height, int width, int c,
               int pyramid_levels, const papof_params* params, double* vx, double* vy, double* warpI2,
               double timing_sec[PAPOF_N_TIMERS]);

/* Same, with every buffer already resident in this handle's device memory (HWC in, planar vx/vy and HWC
 * warpI2 out; all fp64).  Enqueues on the handle's stream and returns after the stream has drained. */
int papof_flow_device(papof_handle* h, const double* d_im1, const double* d_im2, int height, int width, int c,
                      int pyramid_levels, const papof_params* params, double* d_vx, double* d_vy,
                      double* d_warpI2, double timing_sec[PAPOF_N_TIMERS]);

/* ---- uint8 frames.  The reference's caller decodes a JPEG to uint8 and hands
 * `im.astype(float) / 255.` to pyflow (Code/Serial/OpticalFlowCalculation.py:65-70); these entry points take the
 * uint8 samples (HWC interleaved) and do that one IEEE division per sample on the device while the frame is
 * planarised -- the same bits, 1 byte instead of 8 per sample over PCIe.  Outputs as papof_flow. */
int papof_flow_u8(papof_handle* h, const unsigned char* im1, const unsigned char* im2, int height, int width,
                  int c, int pyramid_levels, const papof_params* params, double* vx, double* vy, double* warpI2,
                  double timing_sec[PAPOF_N_TIMERS]);
int papof_flow_device_u8(papof_handle* h, const unsigned char* d_im1, const unsigned char* d_im2, int height,
                         int width, int c, int pyramid_levels, const papof_params* params, double* d_vx,
                         double* d_vy, double* d_warpI2, double timing_sec[PAPOF_N_TIMERS]);

/* ---- sequence mode.  The reference's TestSuite walks a 102-frame collection as 101
 * overlapping pairs (Code/Serial/TestSuite.py:69-81: frame n -> n+1, then n+1 -> n+2, ...), rebuilding the
 * pyramid of every frame twice.  Here the handle keeps the pyramid of the last pushed frame in its arena: each
 * push uploads ONE frame, builds ONE pyramid and returns the flow from the previous frame to this one --
 * bit-identical to papof_flow(previous, frame).  *have_flow = 0 for the push that primes a sequence (first push,
 * after papof_seq_reset, or when shape / levels / ratio differ from the kept frame: a new sequence starts),
 * 1 otherwise; outputs are written only when *have_flow == 1.  Any other call on the handle ends the sequence. */
int papof_seq_reset(papof_handle* h);
int papof_seq_push(papof_handle* h, const double* frame, int height, int width, int c, int pyramid_levels,
                   const papof_params* params, double* vx, double* vy, double* warpI2,
                   double timing_sec[PAPOF_N_TIMERS], int* have_flow);
int papof_seq_push_u8(papof_handle* h, const unsigned char* frame, int height, int width, int c,
                      int pyramid_levels, const papof_params* params, double* vx, double* vy, double* warpI2,
                      double timing_sec[PAPOF_N_TIMERS], int* have_flow);
/* frame already resident in device memory (fp64 HWC, or uint8 HWC when is_u8 != 0) */
int papof_seq_push_device(papof_handle* h, const void* d_frame, int is_u8, int height, int width, int c,
                          int pyramid_levels, const papof_params* params, double* d_vx, double* d_vy,
                          double* d_warpI2, double timing_sec[PAPOF_N_TIMERS], int* have_flow);

/* ---- ONE frame pair sharded as 2-D tiles over several GPUs (BASELINE.json configs[4]: 1920x1080
 * tiled 2x4 across 8 GPUs with RCCL halo exchange over xGMI).  One rank per GPU / process; every rank holds both
 * frames; rank 0 receives the assembled (vx, vy, warpI2).  Red-black SOR order (PAPOF_SOR_REDBLACK, n_inner = 1): a
 * red-black half-sweep reads only the other colour's previous values, so the tiled result is bit-identical to
 * papof_flow_device() in that mode on one GPU.  (PAPOF_SOR_EXACT: see papof_bands_plan below.)  `halo` = ghost-zone depth in half-sweeps (one (du, dv) exchange per
 * `halo` half-sweeps; 0 = default 10).  All calls below except _grid/_rect/_halo_message are collective. */
typedef struct papof_tiles papof_tiles;
#define PAPOF_TILES_ID_BYTES 128
#define PAPOF_BAND_ROWS 62 /* rows per solver band of the exact-order kernel (sor.hip) */
/* rows x cols grid for n ranks (8 -> 2 x 4: tiles of 480 x 540 at 1080p, 4 -> 2 x 2, 2 -> 1 x 2) */
int papof_tiles_grid(int nranks, int* rows, int* cols);
/* rect = {x0, y0, x1, y1} (half-open) of rank's tile of a width x height plane */
int papof_tiles_rect(int width, int height, int rows, int cols, int rank, int rect[4]);
/* the rectangle rank `src` sends to rank `dst` when every rank needs its tile grown by `halo` pixels (empty: x1 <= x0);
 * pure function of its arguments -- both ends of a message compute it, nothing is negotiated */
int papof_tiles_halo_message(int width, int height, int rows, int cols, int halo, int src, int dst, int rect[4]);
/* PAPOF_SOR_EXACT on a tile group: the reference's own sweep order, split into `nranks` horizontal ranges of SOLVER BANDS
 * (csrc/tiles.hip: bands_flow; the rows x cols grid is ignored) -- bit-identical to papof_flow_device(), i.e. the sharded
 * configuration that meets the 1e-4 parity bar; default branches, n_inner = 1, at most 128 sweeps.  Every rank runs the
 * bands [B0, B1) of every solve; the one 16-byte cell per step that crosses a cut goes, with the producer's progress, straight
 * into the planes of the rank below (LOCAL transport: one process, one device), or -- transports that cannot address peer
 * memory from a kernel: RCCL -- as one message per solve and cut after the producer's kernel has finished (the ranks then take
 * turns within a solve).  papof_bands_plan: for a height x width level with n_sor sweeps, out = {B0, B1, first / one-past-last
 * coefficient row the rank's tasks touch, first / one-past-last row whose final increments it holds (a partition)}. */
int papof_bands_plan(int height, int width, int n_sor, int nranks, int rank, int out[6]);
/* RCCL transport: rank 0 obtains an id (ncclGetUniqueId) and hands it to every rank by whatever means the launcher
 * has (bench.py: torch.distributed broadcast); then every rank creates its member of the group on its own handle. */
int papof_tiles_unique_id(unsigned char id[PAPOF_TILES_ID_BYTES]);
int papof_tiles_create(papof_handle* h, const unsigned char id[PAPOF_TILES_ID_BYTES], int rank, int nranks, int rows,
                       int cols, int halo, papof_tiles** out);
/* LOCAL transport: all `nranks` ranks live in this process (one handle each, e.g. all on one device); messages are
 * device-to-device copies.  Each out[r] must then be driven by its own host thread.  Same orchestration code as the
 * RCCL transport; exists so that the tiled path can be parity-tested on a one-GPU box. */
int papof_tiles_create_local(papof_handle* const* handles, int nranks, int rows, int cols, int halo,
                             papof_tiles** out /* nranks entries */);
int papof_tiles_flow_device(papof_tiles* t, const double* d_im1, const double* d_im2, int height, int width, int c,
                            int pyramid_levels, const papof_params* params /* NULL = reference schedule, red-black */,
                            double* d_vx, double* d_vy, double* d_warpI2 /* rank 0 only; may be NULL elsewhere */,
                            double timing_sec[PAPOF_N_TIMERS]);
int papof_tiles_stats(const papof_tiles* t, long* exchanges, size_t* bytes); /* of the last call, this rank */
/* what the TRANSPORT reports about the group (RCCL: ncclCommCount / ncclCommUserRank of this member's communicator), and
 * the tile grid / ghost depth in use: lets a multi-GPU bench line show that RCCL saw N ranks and the rows x cols split */
int papof_tiles_comm_info(const papof_tiles* t, int* nranks_seen, int* rank_seen, int* rows, int* cols, int* halo);
void papof_tiles_destroy(papof_tiles* t);

/* hipGraph replay of whole calls (also PAPOF_GRAPH=1 when the handle is created).  A call with given arguments runs
 * eagerly the first time, is captured (both streams, every kernel, memset and copy) the second time and is one
 * hipGraphLaunch from then on; only "Total C++ Execution" is timed in that mode.  For small frames, and for several
 * calls in flight, the host-side launch path is the limit: 240x135 pairs on the reference schedule are ~700 launches
 * each.  Results are the same bits. */
int papof_set_graph_mode(papof_handle* h, int on);

/* Whether a call spreads over several streams of its own (the preparation beside the coarse levels' solves, PCIe copies beside
 * kernels: the default, fastest for ONE call at a time) or stays on the handle's one stream (on = 0; also PAPOF_OVERLAP=0 when
 * the handle is created).  With several handles in flight -- a collection of pairs, flow_collection() -- the other handles'
 * calls fill the idle time and one stream per handle is faster (fewer hardware queues shared): 240x135 pairs on the
 * reference schedule 3.2 -> 2.2 ms per pair with 16 in flight, 480x270 4.9 -> 3.4.  Results are the same bits. */
int papof_set_stream_overlap(papof_handle* h, int on);

/* Device memory helpers for callers without a HIP binding (bench.py, ctypes users). */
int papof_dev_alloc(papof_handle* h, size_t bytes, void** out);
int papof_dev_free(papof_handle* h, void* p);
int papof_dev_upload(papof_handle* h, void* dst, const void* src, size_t bytes);
int papof_dev_download(papof_handle* h, void* dst, const void* src, size_t bytes);
void* papof_stream(papof_handle* h); /* the hipStream_t every kernel of this handle is launched on */
/* Page-locked host memory for RESULT arrays (hipHostMalloc).  The reference's pyflow.pyx allocates vx, vy, warpI2 itself
 * for every call (np.zeros, Code/Serial/pyflow.pyx:44-52): 83 MB of fresh pages per 1080p pair, i.e. ~20 k first-touch
 * page faults, a staged device-to-host copy, and an munmap when the caller drops them.  A binding that takes its result
 * arrays from here (and recycles them when they are garbage-collected: papteam_opticalflow_amd/dropin/pyflow.pyx,
 * capi.py) gets the results by direct DMA into memory that is already resident.  Any host pointer works as an output
 * of papof_flow / papof_coarse2fine_flow; these are merely the fastest ones. */
int papof_host_alloc(size_t bytes, void** out);
int papof_host_free(void* p);

/* ---- stage entry points (host buffers, reference HWC layout): one per reference function on the path,
 * used by the parity tests to check each kernel in isolation against the oracle. ---- */

/* GaussianPyramid::ConstructPyramidLevels, src/GaussianPyramid.cpp:79-108.  dims[2i]=width, dims[2i+1]=
 * height of level i; data==NULL returns only dims.  *n_elems = total doubles of all levels. */
int papof_stage_pyramid(papof_handle* h, const double* im, int height, int width, int c, double ratio,
                        int levels, int* dims, double* data, long* n_elems);
/* Image::GaussianSmoothing, src/Image.h:1203-1225 (fsize <= 8). */
int papof_stage_gaussian(papof_handle* h, const double* im, int height, int width, int c, double sigma,
                         int fsize, double* out);
/* Image::imresize(result, ratio) src/Image.h:751-763 ; out is int(h*ratio) x int(w*ratio). */
int papof_stage_resize_ratio(papof_handle* h, const double* im, int height, int width, int c, double ratio,
                             double* out);
/* Image::imresize(w,h) src/Image.h:778-783. */
int papof_stage_resize_wh(papof_handle* h, const double* im, int height, int width, int c, int dst_w,
                          int dst_h, double* out);
/* OpticalFlow::im2feature, src/OpticalFlow.cpp:911-961; returns (in *fc) 5 for c==3, 3 for c==1, else c. */
int papof_stage_im2feature(papof_handle* h, const double* im, int height, int width, int c, double* out,
                           int* fc);
/* OpticalFlow::warpFL, src/OpticalFlow.cpp:154-159. */
int papof_stage_warpFL(papof_handle* h, const double* im1, const double* im2, const double* vx,
                       const double* vy, int height, int width, int c, double* out);
/* OpticalFlow::getDxs, src/OpticalFlow.cpp:80-122. */
int papof_stage_getDxs(papof_handle* h, const double* im1, const double* im2, int height, int width, int c,
                       double* imdx, double* imdy, double* imdt);
/* Linear system of one inner iteration with du=dv=0 (src/OpticalFlow.cpp:295-448): outputs phi, imdxy,
 * imdx2, imdy2 and the two right-hand sides, each height*width. */
int papof_stage_linear_system(papof_handle* h, const double* im1, const double* warp, const double* u,
                              const double* v, int height, int width, int c, double alpha, double* phi,
                              double* imdxy, double* imdx2, double* imdy2, double* rhs1, double* rhs2);
/* OpticalFlow::Laplacian, src/OpticalFlow.cpp:641-690. */
int papof_stage_laplacian(papof_handle* h, const double* in, const double* weight, int height, int width,
                          double* out);
/* The SOR sweeps, src/OpticalFlow.cpp:451-505, starting from du=dv=0. */
int papof_stage_sor(papof_handle* h, const double* phi, const double* imdxy, const double* imdx2,
                    const double* imdy2, const double* rhs1, const double* rhs2, int height, int width,
                    double alpha, double omega, int n_sor, int sor_mode, double* du, double* dv);
/* One pyramid level: OpticalFlow::SmoothFlowSOR, src/OpticalFlow.cpp:238-536.
 * warp, u, v are in/out. */
int papof_stage_smoothflow(papof_handle* h, const double* im1, const double* im2, double* warp, double* u,
                           double* v, int height, int width, int c, double alpha, int n_outer, int n_inner,
                           int n_sor, double omega, int sor_mode);
/* GaussianPyramid::ConstructPyramid (the min-width variant the reference does not call, src/GaussianPyramid.cpp:47-77)
 * differs from ConstructPyramidLevels in its level count only: *levels = (int)(log((double)min_width / width) /
 * log(ratio)) after the same clamp of the ratio (:50-53); pass it as `levels` / `pyramid_levels` anywhere. */
int papof_pyramid_levels_for_min_width(int width, double ratio, int min_width, int* levels);
/* SmoothFlowSOR with the non-default branches: interpolation / noise_model as in papof_params; gm (5 * c doubles: alpha,
 * sigma, beta, sigma^2, beta^2 per channel; NULL = GaussianMixture::reset() values) is in/out when noise_model = GMIXTURE. */
int papof_stage_smoothflow_ex(papof_handle* h, const double* im1, const double* im2, double* warp, double* u,
                              double* v, int height, int width, int c, double alpha, int n_outer, int n_inner,
                              int n_sor, double omega, int sor_mode, int interpolation, int noise_model, double* gm);
/* OpticalFlow::estGaussianMixture, src/OpticalFlow.cpp:539-591 (prior 0.9); gm in/out as above. */
int papof_stage_est_gaussian_mixture(papof_handle* h, const double* im1, const double* im2, int height, int width,
                                     int c, double* gm);
/* Image::warpImageBicubicRef alone (clamp = 0: the first warp of a level, src/OpticalFlow.cpp:816) or with threshold(). */
int papof_stage_bicubic_warp_ex(papof_handle* h, const double* im1, const double* im2, const double* vx,
                                const double* vy, int height, int width, int c, int clamp, double* out);
/* Image::warpImageBicubicRef + threshold, src/Image.h:2587-2701, :2031-2045 (final warp of the originals). */
int papof_stage_bicubic_warp(papof_handle* h, const double* im1, const double* im2, const double* vx,
                             const double* vy, int height, int width, int c, double* out);

/* ---- the step after the path: the reference's 16-bit flow encoding,
 * OpticalFlow::SaveOpticalFlow / LoadOpticalFlow (src/OpticalFlow.cpp:963-1015) with AssembleFlow / DissembleFlow
 * (src/OpticalFlow.h:70-91): out[(i*width + j)*2 + {0,1}] = (unsigned short)((clamp(v{x,y}, -200, 200) + 200) * 160);
 * back: (double)q / 160 - 200.  The file Image<unsigned short>::saveImage writes around it (src/Image.h:825-837:
 * 16-byte type name, width, height, channels as int, one bool) is plain host I/O: papteam_opticalflow_amd.save_flow16. */
int papof_flow_quantize16(papof_handle* h, const double* vx, const double* vy, int height, int width,
                          unsigned short* out);
int papof_flow_dequantize16(papof_handle* h, const unsigned short* q, int height, int width, double* vx,
                            double* vy);

/* Flow visualisation of the reference's caller, generateOutputFlowImageFile (Code/Serial/OpticalFlowCalculation.py:
 * 143-162, disabled there at :137 and dependent on cv2): hue = direction, value = magnitude (min-max normalised),
 * saturation 255, HSV -> BGR the way OpenCV's 8-bit conversion does; bgr is height*width*3 bytes, B first.
 * PARITY UNPINNED: OpenCV is not installed in this environment and its cartToPolar uses an approximate atan2. */
int papof_flow_to_bgr(papof_handle* h, const double* vx, const double* vy, int height, int width,
                      unsigned char* bgr);

/* ---- measurement hook for bench.py: time `reps` back-to-back SOR solves of `n_sor` sweeps on synthetic
 * coefficient planes already resident in HBM (SURVEY.md §8d micro-benchmark), with HIP events recorded
 * on the handle's stream.  Returns average milliseconds per solve in *ms_per_solve. */
int papof_bench_sor(papof_handle* h, int height, int width, int n_sor, int sor_mode, int reps, unsigned seed,
                    double* ms_per_solve);

/* Measurement aid for bench.py: how a solve of `n_sor` sweeps on a height x width plane is issued in `sor_mode` on this
 * handle -- solver-kernel launches per solve, and the sweeps one launch runs (exact order: 1, or 2 with two sweeps per
 * wave; red-black: HALF-sweeps per launch of the LDS-tiled, temporally blocked kernel; Jacobi: sweeps per launch). */
int papof_sor_plan(papof_handle* h, int height, int width, int n_sor, int sor_mode, int* launches, int* depth);

/* Measurement aid for bench.py: what the LAST papof_flow* / papof_seq_push* call on this handle launched -- exact-order
 * solver kernels in all (a level solved as strips of bands issues one launch per strip and solve), and how many
 * seconds of the reported Phase5_SOR are launches on the strip streams, i.e. ran beside the main stream's time line. */
int papof_last_sor_stats(papof_handle* h, int* launches, double* strip_streams_sec);

/* Measurement aid for bench.py (roofline.by_level / by_kernel): the solves of the LAST papof_flow* / papof_seq_push* call on
 * this handle, in stream order.  *n = their number; for i < min(*n, cap): info[6 i ..] = {height, width, sweeps, kind,
 * depth, launches} with kind 0 = k_sor_exact, 1 = k_sor_fused, 2 = k_sor_group, 3 / 4 = k_sor_blocked red-black / Jacobi,
 * 5 = one launch per (half-)sweep, 6 = k_sor_tiny (exact order, whole plane in one workgroup); depth = software-pipeline depth R (exact order) or (half-)sweeps per launch (blocked);
 * sec[i] = the solver kernels' own HIP-event seconds of that solve (0 when the call collected no timers). */
int papof_last_sor_solves(papof_handle* h, int cap, int* n, int* info, double* sec);

/* The Laplacian-noise guard of the reference (src/OpticalFlow.cpp:399-400: psi of a feature channel stays 0 while the
 * channel's noise estimate, estLaplacianNoise :594-639, is below 1E-20 -- duplicate frames, flat synthetic images).  A call
 * runs WITHOUT the estimate while it collects proofs that the guard could not have tripped (csrc/api.hip: LapGuard); a call
 * that ends without them is run again inside the same papof_flow* call with the estimate after every outer iteration and
 * the guard in the assembly, and the handle then stays in that exact pass until a call proves it unnecessary again.  Results
 * are the reference's either way.  out[0] = calls that were run twice, out[1] = calls run in the exact pass from the start,
 * out[2] = 1 when the next call will start in the exact pass, out[3] = 0 when PAPOF_LAP_GUARD=0 switched the guard off (an
 * A/B switch for its cost: results then differ from the reference's on tripping inputs).  hipGraph replay, the bicubic
 * branch and the papof_stage_smoothflow* entry points always take the exact pass; the Gaussian-mixture branch has no such
 * guard (:381-397).  One pair over several ranks (papof_tiles_*): the exact-order band split has no exact pass, so it PROVES
 * per call that the guard cannot have tripped (every rank checks every pixel of its rows behind every update; the flags of
 * all ranks are gathered) or returns PAPOF_EINVAL on every rank with a message that names the one-GPU call; the red-black
 * tiles -- not the reference's sweep order anyway -- run without the guard. */
int papof_lap_guard_stats(papof_handle* h, int out[4]);

/* B = n_pairs frame pairs of ONE shape in ONE launch chain (csrc/batch.hip) -- the reference's benchmark walks collections of
 * small frames (Code/Serial/TestSuite.py:69-81: 101 pairs per collection; :91: 240x135 ... 1920x1080), and a 240x135 pair
 * alone is ~210 launches of a few microseconds each: here every launch serves all pairs of the batch.  Host buffers in and out,
 * as papof_flow / papof_flow_u8.  sequence != 0: pair i = (frames[i], frames[i + 1]), n_pairs + 1 frames -- a video; every
 * frame's pyramid and features are built once.  sequence == 0: pair i = (frames[2 i], frames[2 i + 1]), 2 n_pairs frames.
 * vx[i], vy[i] (height x width) and warpI2[i] (height x width x c) receive pair i's results: bit-identical to
 * papof_flow(frames of pair i).  timing_sec: Total = the caller's wall time of the batch, Phase5_SOR = the solver kernels' own
 * time; the other phases are not separated in a batch.  What the batched chain covers: the default branches in the reference's
 * own sweep order (PAPOF_SOR_EXACT, n_inner = 1, bilinear, Laplacian noise model), 1 or 3 channels, fewer than 16 solver
 * bands per level (frames up to ~900 rows); anything else -- and n_pairs = 1 -- runs as consecutive single calls through this
 * same entry point.  A pair whose Laplacian-noise guard (papof_lap_guard_stats) cannot be proven open in the batch is run again
 * through the single call.  Any sequence kept by papof_seq_push* on the handle ends. */
int papof_flow_batch(papof_handle* h, int n_pairs, int sequence, const double* const* frames, int height, int width, int c,
                     int pyramid_levels, const papof_params* params, double* const* vx, double* const* vy,
                     double* const* warpI2, double timing_sec[PAPOF_N_TIMERS]);
int papof_flow_batch_u8(papof_handle* h, int n_pairs, int sequence, const unsigned char* const* frames, int height,
                        int width, int c, int pyramid_levels, const papof_params* params, double* const* vx,
                        double* const* vy, double* const* warpI2, double timing_sec[PAPOF_N_TIMERS]);

/* Measurement aid (tools/collection_trace.py): host-side wall seconds of the LAST papof_flow* / papof_seq_push* call on this
 * handle -- out[0] from the call's entry until everything was enqueued (the runtime's launch path: ~200 launches for a
 * 240x135 pair on the reference schedule), out[1] the wait for the streams that followed, out[2] reserved (0). */
int papof_last_host_times(papof_handle* h, double out[3]);

/* Test aid: the strip schedule (api.hip: smooth_flow_strips) a level of height x width with `n_sor` sweeps and
 * `n_outer` outer iterations gets on this handle.  *strips = S (1: the level is not cut).  out, if not NULL, receives
 * for n = 0 .. n_outer and s = 0 .. S five ints each -- first band, first row of the update / phi / smoothing / assembly
 * stage of strip s before solve n -- (n_outer + 1) * (S + 1) * 5 ints (cap = ints available); *band_rows, *koff: rows
 * per band and climb of the bands up to the last sweep of the bound solver layout. */
int papof_strip_plan(papof_handle* h, int height, int width, int n_sor, int n_outer, int want_strips, int* strips,
                     int* out, int cap, int* band_rows, int* koff, int* bands);

/* Test aid: one exact-order solve of `n_sor` sweeps on synthetic height x width planes (the micro-benchmark's), once
 * whole and `reps` times as two strips of solver bands -- bands < split_band on a second stream, the rest `delay_us`
 * microseconds later on the handle's stream (sor.hip: sor_solve_bands).  *mismatches = 16-byte cells of the solver's
 * (du, dv) planes, intermediate sweeps included, that differ from the whole solve's (0 expected); *bands = bands of the
 * layout.  PAPOF_EINVAL when this layout cannot be solved in strips (<= 8 bands, < 3 sweeps, split out of range). */
int papof_test_sor_strips(papof_handle* h, int height, int width, int n_sor, int split_band, int reps, int delay_us,
                          long long* mismatches, int* bands);

#ifdef __cplusplus
}
#endif
#endif /* PAPOF_H */
