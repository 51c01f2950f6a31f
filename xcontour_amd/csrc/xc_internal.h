// Internal declarations shared by the translation units of libxcontour_hip.so.
// Not part of the C ABI (that is include/xcontour_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <type_traits>
#include <vector>
#include "../../include/xcontour_hip.h"

// Experiment knobs of the K3 geometry, read from the environment ONCE, in xc_create (contexts may be driven from several
// threads; nothing reads the environment after that).  0 / -1 = the built-in choice.
struct HistKnobs {
    int copy_out_kb = 256; // XC_COPY_OUT_KB  results up to this size leave through the copy kernel / are written straight into the pinned buffer (1 .. 1024; measured at the demo size: the 217 KB block of the fused keff 93-99 -> 86-88 us per call against a DMA copy)
    int copy_kernel = 1; // XC_COPY_KERNEL   small transfers of the host-form entry points (inputs <= 64 KB, results <= copy_out_kb): 1 one copy KERNEL per direction between the pinned bounce buffers and device memory, 0 one DMA copy per array
    int xcd_map = 1;     // XC_HIST_XCDMAP   XCD-aware block order when blocks per slab is a multiple of 8
    int tile_map = 1;    // XC_HIST_TILEMAP  strip-fastest wave order
    int vec4 = -1;       // XC_HIST_VEC4     four cells per lane: -1 float32 tracers only, 0 never, 1 always
    int e32 = 1;         // XC_HIST_E32      float32 tracer + float32 levels: raw float32 rows, float32 bin search, three rows in flight
    int threads = 0;     // XC_HIST_THREADS  threads per block
    int ncopy = 0;       // XC_HIST_NCOPY    LDS histogram copies
    int rows = 0;        // XC_HIST_ROWS     (strip, row) pairs per wave
    int bps = 0;         // XC_HIST_BPS      blocks per slab
    int cross_ncopy = 0, cross_blocks = 0;   // XC_CROSS_NCOPY, XC_CROSS_BLOCKS (K9)
    int k1_nt = 0;       // XC_K1_NT         K1: 1 forces the non-temporal loads also for launches that fit the Infinity Cache
    int lwa_fast = 1;    // XC_LWA_FAST      K7: the O(ny log ny) interval kernel for planes of more than 512 rows (0: never, 2: for every plane)
    int lwa_strip = 1;   // XC_LWA_STRIP     K7: the one-launch kernel with the strip in LDS where it fits (0: always prep + streaming kernel)
    int sort_range = 1;  // XC_SORT_RANGE    K8: three range-key passes + short-run repair for float64 tracers (0: always eight passes)
    int single = 1;      // XC_KEFF_SINGLE   xc_keff_dev calls of at most kSingleMaxSlabs slabs: 1 the single-read kernel where the slab fits the chip, 0 never
    int single_timeout_us = 50000;   // XC_KEFF_SINGLE_TIMEOUT_US   bound of every wait on another workgroup inside that kernel (status 2 when it expires)
    int single_map = 0;  // XC_KEFF_SINGLE_MAP  wave -> tile order of that kernel: 0 chunks of one strip per workgroup, 1 adjacent strips of the same rows
};

struct xc_ctx {
    int device = 0;
    HistKnobs knobs;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;            // uploads that overlap compute (xc_memcpy_h2d_async)
    hipStream_t comm_stream = nullptr;            // the gather's blocks leave on it while the next launch set computes (xc_comm_*; created on first use)
    hipEvent_t ev_comm_in = nullptr, ev_comm_out = nullptr;
    hipEvent_t ev_copy = nullptr, ev_compute = nullptr;
    unsigned* pinned_flag = nullptr;   // 64 bytes of pinned host memory: the sort's one read-back
    // Small transfers of the host-form entry points go through pinned bounce buffers (round 5): a copy between PAGEABLE memory and the
    // device is staged and waited for inside hipMemcpyAsync itself -- every small input and every result vector of a call cost a blocking
    // round trip of its own (three D2H copies in xc_levels, five in the Keff epilogue).  Inputs are memcpy'd into `pin_in` and leave
    // with a truly asynchronous copy; results land in `pin_out` and are handed to the caller's arrays after the call's ONE stream
    // synchronisation (xc_sync).  Both are bump-allocated per call and reset by xc_sync; what does not fit takes the direct path.
    char* pin_in = nullptr;  size_t pin_in_off = 0;
    char* pin_out = nullptr; size_t pin_out_off = 0;
    struct PendingOut { void* host; const void* pinned; size_t bytes; const void* dev; };   // dev != nullptr: still to be fetched by the copy kernel (flush_out)
    std::vector<PendingOut> pending_out;
    // the last few SMALL inputs a host-form call uploaded (bin edges, gradient metrics: <= 64 KB each), kept on the device with a host copy
    // of their bytes: a call that hands over the same bytes again -- the reference's sequence bins the tracer twice against the same
    // contours, cal_integral_within_contours(ctr) and (ctr, integrand) -- reads the device copy after one memcmp, with no transfer at all.
    // Recognised by CONTENT, never by address.  An entry written during the current call (same `epoch`: xc_sync starts a new one) is
    // not evicted: kernels already enqueued may still read it.
    struct SmallIn { std::vector<char> host; void* dev = nullptr; uint64_t used = 0; uint64_t epoch = ~0ull; };
    SmallIn small_in[4];
    uint64_t small_clock = 0, small_epoch = 0, small_hits = 0, small_misses = 0;
    struct PendingIn { void* dev; const void* pinned; size_t bytes; };                      // staged in pin_in, not yet on the device (flush_in)
    std::vector<PendingIn> pending_in;
    // where a host-form call spends its time, accumulated between two xc_trace calls (seconds): input staging (memcpy + enqueue),
    // result hand-over (enqueue + memcpy), waiting for the stream
    double tr_h2d = 0.0, tr_d2h = 0.0, tr_sync = 0.0;
    struct Resident { const char* host; size_t bytes; void* dev; };
    std::vector<Resident> resident;    // host arrays with a device mirror (xc_keep_resident): the host-form entry points copy from the mirror
    int cus = 0;
    char name[256] = {0};
    std::string err;
    // grow-only device scratch owned by the context
    void*  scratch = nullptr;   size_t scratch_bytes = 0;   // kernel workspaces (partials, minmax, edges)
    void*  arena = nullptr;     size_t arena_bytes = 0;     // staging for the host-pointer entry points
    double* ones = nullptr;     size_t ones_n = 0;          // per-row weight 1.0 (dA_rank == XC_DA_NONE)
    void*  big = nullptr;       size_t big_bytes = 0;       // k_finalize work arrays when they do not fit the LDS (many contours)
    // timing of the dominant kernel
    int timing = 0;
    hipEvent_t ev_hist0 = nullptr, ev_hist1 = nullptr;
    int ev_valid = 0;
    void* comm = nullptr; int comm_nranks = 0, comm_rank = 0;   // RCCL communicator (xc_comm_*)
    hipEvent_t user_ev0 = nullptr, user_ev1 = nullptr;     // one-shot caller events around the next K3 launch
    // min/max partials of the NEXT batch, produced inside the K3 pass (xc_keff_desc.q_next)
    double* mmnext[2] = {nullptr, nullptr};  size_t mmnext_bytes[2] = {0, 0};
    int mm_cur = 0, mm_valid = 0, mm_P = 0, mm_dtype = 0;
    const void* mm_q = nullptr;  int64_t mm_nslab = 0, mm_ny = 0, mm_nx = 0;  int mm_gen = 0;
    int lwa_exact = 0;          // xc_set_lwa_exact: keep the bit-exact band walk for every plane
    int last_lwa_path = 0;      // K7, last call: 0 band walk, 1 interval kernel, 2 its premises failed the check (band walk), -1 decided on the device (read lwa_flag)
    unsigned lwa_epoch = 0;
    unsigned* lwa_flag = nullptr;   // device word written by k_lwa_check: the interval kernel and the band walk gate themselves on it
    int last_sort_path = 0;     // K8, last call: 0 eight / four key passes, 1 three range-key passes sufficed, 2 they did not (re-sorted)
    // the single-read Keff kernel (xc_keff1.hip): two sets of synchronisation records + per-slab accumulators; launch n works in set n % 2
    // and clears what launch n - 1 left in the other one (the kernel boundary orders the two), so no memset launch sits in the chain
    void* single_ws = nullptr;  unsigned single_launches = 0;  int single_dirty_bins[2] = {0, 0};
    int last_keff_path = 0;     // last xc_keff_dev call: 0 min/max pass + histogram pass (two reads of the tracer), 1 the single-read kernel
    unsigned long long* single_stamps = nullptr;   // diagnostics (xc_dbg_single_stamps): wall-clock stamps of every workgroup at the phase boundaries
};

namespace xc {

int fail(xc_ctx* ctx, int code, const std::string& msg);
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device)
int ensure_big_lds(xc_ctx* ctx, const void* kernel, int bytes);
int hipfail(xc_ctx* ctx, hipError_t e, const char* what);

#define XC_HIP(ctx, call)                                                     \
    do {                                                                      \
        hipError_t _e = (call);                                               \
        if (_e != hipSuccess) return xc::hipfail((ctx), _e, #call);           \
    } while (0)

// grow-only scratch (never shrinks; pointer may change between calls, never within one)
int ensure_scratch(xc_ctx* ctx, size_t bytes);
int ensure_arena(xc_ctx* ctx, size_t bytes);
int ensure_ones(xc_ctx* ctx, size_t n);
int ensure_big(xc_ctx* ctx, size_t bytes);

// ---------------------------------------------------------------- launch geometry
constexpr int kMinmaxBlocks = 2048;   // partial min/max pairs per slab (upper bound, see minmax_blocks)
// blocks per slab of the K1 pass: at least ~8192 cells per block, so that a stack of many small slabs is not
// shredded into half a million 16-cell blocks (512 slabs of 90x180: 70 -> ~2 ps per cell).  A launch of FEW large slabs
// (one cfg2 slab = 52 MB) instead wants every byte requested at once: up to 8 blocks per CU on 256 CUs, one batch of eight
// 16-byte loads per thread (4096 float64 cells per block) -- one slab: 792 blocks, two dependent rounds -> 1583, one round
inline int minmax_blocks(int64_t ncell, int64_t nslab)
{
    int64_t p = (ncell + 8191) / 8192;
    if (p > 1024) p = 1024;
    if (nslab < 1) nslab = 1;
    if (p * nslab < 2048) {
        int64_t f = (ncell + 4095) / 4096;
        if (f * nslab > 2048) f = 2048 / nslab;
        if (f > p) p = f;
    }
    return (int)(p < 1 ? 1 : (p > kMinmaxBlocks ? kMinmaxBlocks : p));
}
constexpr int kHistThreads  = 1024;   // 16 waves: one block per CU (LDS-bound)
#ifndef XC_MAX_COPIES
#define XC_MAX_COPIES 16
#endif
#ifndef XC_LDS_BUDGET_KB
#define XC_LDS_BUDGET_KB 150
#endif
constexpr int kMaxCopies    = XC_MAX_COPIES;     // lane-privatised LDS histogram copies
constexpr size_t kLdsBudget = (size_t)XC_LDS_BUDGET_KB * 1024;
constexpr int kE32MaxBins  = 2048;   // the float32-edges variant of K3 (E32) keeps a float32 copy of the edges in LDS: ordinary contour counts only

struct HistGeom {
    int threads;    // threads per block (multiple of 64, <= kHistThreads)
    int vec;        // cells per lane per row load (1, 2 or 4)
    int nstrip;     // column strips of 64*vec cells
    int bps;        // blocks per slab
    int ncopy;      // LDS histogram copies (power of two)
    int nch;        // weight channels
    size_t lds;     // dynamic LDS bytes
    size_t part_h_doubles;   // per-(slab,block) partial doubles = nch*nbin
};

// ---------------------------------------------------------------- kernel argument blocks
struct HistArgs {
    const void*   q;
    const void*   q_next;       // optional next batch (min/max partials by-product)
    double*       mm_next;      // [nslab][bps][2]
    const double* dA;
    const void*   integ[XC_MAX_INTEGRANDS];
    int           integ_f32[XC_MAX_INTEGRANDS];
    const double* edges;        // explicit edges (levels_mode == 0)
    const double* mmpart;       // [nslab][P][2] partial min/max (levels_mode == 1)
    int           P;
    int           levels_mode;
    int           nbin;
    int           edges_per_slab;
    int           last_closed;
    int           negate;
    int           increase, q_f32, ctr_f32, right_edge;
    double        inv_nm1;      // 1.0/(N-1) (levels mode)
    int           dA_rank, prod_f32;
    int           dA_pos_finite;   // host verified: every dA value is finite and >= 0 (skips fillna selects)
    const double* rdx;
    const double* rdy;
    int           periodic_x;
    int64_t       ny, nx;
    int           nstrip, ncopy;
    int           bps, nslab_grid, xcd_map;   // launch geometry (set by launch_hist): blocks per slab, slabs, XCD-aware block order
    int           nchunk;       // > 0: wave -> (row chunk, strip) with the strip fastest (see k_hist); 0: even split of the strip-major pairs
    double*       part_h;       // [nslab][bps][nch][nbin]
    unsigned*     part_c;       // [nslab][bps][nbin]
    double*       acc_h;        // non-null: a launch of FEW slabs (hundreds of blocks per slab): the block ADDS its sums to [nslab][nch][nbin] here (zeroed
    unsigned long long* acc_c;  //           by the caller; global float64 / 64-bit atomics) instead of storing a partial, and k_reduce_partials is not launched
    double*       ctr_out;      // levels of slab s at ctr_out + s * ctr_stride (levels mode, may be null)
    int           ctr_stride;   // doubles between consecutive slabs in ctr_out (nbin: dense)
    double*       edges_out;    // [nslab][nbin+1] (levels mode, may be null)
    int32_t*      status;       // [nslab]         (levels mode, may be null)
    // DET == 3 (one-pass superaccumulator): the bounds that fix every channel's window BEFORE the pass
    double        det_dA_max;   // max finite |dA| (host value), or < 0: read det_dA_max_dev
    const double* det_dA_max_dev;   // [nslab or 1][2] (min, max) pairs of the dA array, as K1 leaves them (stride det_dA_stride pairs per slab)
    int           det_dA_stride;
    const double* det_q_mm;     // [nslab][2] min / max of the tracer (explicit-edges mode with the in-kernel gradient), else null (levels mode: the prologue's own)
    const double* det_int_mm[XC_MAX_INTEGRANDS];   // [nslab][2] min / max of every supplied integrand
    int*          det_c0_out;   // [nslab][nch]: the window constants the blocks derived (the reduction needs them)
};

struct FinalArgs {
    const double*   part_h;
    const unsigned* part_c;
    double*         red_h;    // [nslab][nch][nbin]  stage-1 output (scratch)
    unsigned long long* red_c;// [nslab][nbin]
    int             bps, nch, nbin;
    int             lt, reverse;
    int             skip_reduce;   // red_h / red_c are already filled (deterministic path): run stage 2 only
    int             fuse_reduce;   // set by launch_finalize: stage 1 folded into stage 2 (few partials per slab)
    double*         pdf;      // [nslab][nch][nbin] or null
    uint64_t*       counts;   // [nslab][nbin] or null
    double*         cdf;      // [nslab][nch][nbin] or null
    // Keff epilogue (enabled when keff != 0); channel 0 = area, channel 1 = intgrdS
    int             keff;
    int             ctr_f32;
    const double*   ctr;      // level order, slab s at ctr + s * vstride
    int             vstride;  // doubles between consecutive slabs in ctr and in the o_* vector outputs (0: nbin)
    const double*   tbl;      const double* tbl_coord;   int ntbl;   int tbl_in_lds;
    double*         big;      size_t big_stride;         // work arrays in global memory instead of LDS (doubles per slab), or null
    const double*   preY;     int npre;
    double          nkeff_mask, lmin_scale;
    double *o_area, *o_intS, *o_latEq, *o_dqdA, *o_dSdA, *o_Leq2, *o_Lmin, *o_nkeff, *o_interp;
    const unsigned* abort_flag;   // non-null (behind the single-read Keff kernel): a non-zero word means that launch gave up -- the slab gets
    int32_t*        status_out;   //   status_out[slab] = 2 and nothing else is written
    unsigned long long* dbg;      // diagnostics (xc_dbg_single_stamps): eight wall-clock stamps of the finalize stage of slab 0, or null
};

// ---------------------------------------------------------------- the single-read Keff kernel (xc_keff1.hip)
constexpr int kSingleThreads  = 512;    // 8 waves per workgroup = 2 per SIMD at up to 256 VGPRs: one workgroup per CU, the whole grid co-resident.  (Round 6
                                        // measured 768 threads x 18-row tiles too: the 3072 waves take 4.3 us to dispatch against 2.3 for 2048, carry 11 % of
                                        // halo rows against 7 %, and the tile lands 3.7 us later; their faster binning -- 4.9 against 6.6 us -- gives back 1.7)
constexpr int kSingleRows     = 27;     // rows of a wave's chunk of the slab (its register tile: kSingleRows + 2 rows of 2 cells per lane = 116 VGPRs)
constexpr int kSingleCols     = 124;    // computed columns of a strip (lanes 1..62, two cells each; lanes 0 and 63 hold the halo columns)
constexpr int kSingleMaxSlabs = 2;      // calls of more slabs stream (the two-read path overlaps slabs, this kernel cannot)
constexpr int kSingleMaxBins  = 1024;
constexpr int kSingleStampSlots = 16;
constexpr int kSingleMaxGrid = 256;     // workgroups (= CUs used) at most
struct SingleSlot { unsigned long long kmn, kmx; };     // a workgroup's extrema of the slab as order-preserving keys (~key(min), key(max)); zero = not there yet
struct SingleSet {                       // everything a launch dirties: cleared by the NEXT launch (the accumulators: up to the bins it used)
    SingleSlot slot[kSingleMaxSlabs][kSingleMaxGrid];
    unsigned abort; unsigned pad[15];
    double acc_h[kSingleMaxSlabs * 2 * kSingleMaxBins];           // dense for the launch's own N: [slab][channel][N] (what the finalize stage reads)
    unsigned long long acc_c[kSingleMaxSlabs * kSingleMaxBins];   // [slab][N]
};
struct SingleGeom { int G, nstrip, cps, rpc, ncopy; size_t lds; };
struct SingleArgs {
    const void*   q;  const double* dA;  int dA_rank, dA_pos_finite;
    const double* rdx; const double* rdy; int periodic_x;
    int64_t       ny, nx;  int nslab;
    int           nbin, ncopy, increase, q_f32, ctr_f32, right_edge, last_closed, want_counts;
    double        inv_nm1, inv_n;
    int           G, nstrip, cps, rpc, strip_fast;
    SingleSet*    cur;       SingleSet* other;  int other_dirty_bins;      // the set this launch works in; the one it clears for its successor
    double*       ctr_out;   int ctr_stride;    int32_t* status;
    unsigned long long timeout_ticks;          // of the 100 MHz wall clock
    unsigned long long* stamps;                // diagnostics or null
};
bool single_geometry(const xc_ctx* ctx, int q_dtype, int64_t nslab, int64_t ny, int64_t nx, int N, const void* q, const double* dA,
                     int dA_rank, SingleGeom* g);
int launch_keff_single(xc_ctx* ctx, int q_dtype, const SingleArgs& a, const SingleGeom& g);

// ---------------------------------------------------------------- launchers (defined in the .hip files)
int launch_minmax_partial(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ncell, double* part,
                          double* zero = nullptr, int64_t nzero = 0, bool finite_only = false);      // finite_only: +-inf skipped like NaN; zero: `nzero` 8-byte words cleared by the same launch (the accumulators of HistArgs::acc_h)
struct SmallCopies { const void* src[8]; void* dst[8]; unsigned bytes[8]; };
int launch_copy_small(xc_ctx* ctx, const SmallCopies& c, int count);
int launch_minmax_final(xc_ctx* ctx, const double* part, int64_t nslab, int P, double* out);
int launch_levels(xc_ctx* ctx, const double* minmax, int q_dtype, int64_t nslab, int N, int increase,
                  int ctr_dtype, int right_edge, double* ctr, double* edges, int32_t* status, int P = 0, double* minmax_out = nullptr);
int hist_geometry(xc_ctx* ctx, int q_dtype, int64_t nslab, int64_t ny, int64_t nx, int nbin, int nch,
                  const void* q, HistGeom* g, int keff_fast_layout = 0, int det = 0);
int launch_hist(xc_ctx* ctx, int q_dtype, int nint, int grad, const HistGeom& g, int64_t nslab, const HistArgs& a);
// the one-pass deterministic sums (DET == 3, xc_binning.h): the histogram pass itself and the exact reduction of its limbs
int launch_hist_det3(xc_ctx* ctx, int q_dtype, int nint, int grad, const HistGeom& g, int64_t nslab, const HistArgs& a);
int launch_det3_reduce(xc_ctx* ctx, int64_t nslab, int bps, int nch, int nbin, const double* part_l, const unsigned* part_c,
                       const int* c0, double* red_h, unsigned long long* red_c);
int det_limbs_total(int nch);
int launch_finalize(xc_ctx* ctx, int64_t nslab, const FinalArgs& a);
int launch_rowsum(xc_ctx* ctx, const void* mask, int mask_dtype, const double* dA, int dA_rank,
                  int64_t ny, int64_t nx, int multiply, double* out_rows);
int launch_grad2(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                 const double* rdx, const double* rdy, int periodic_x, double* out);
int launch_lwa(xc_ctx* ctx, const void* q, int q_dtype, const double* Q, const double* coord,
               const double* dA, int dA_rank, double dA_max, const double* M, int M_rank,
               int64_t nslab, int64_t ny, int64_t nx, int increase, int part, int variant,
               const int32_t* mask_idx, int nmask, double* out_lwa, int8_t* out_masks);
size_t sort_workspace_bytes(int64_t n, int64_t nslab);
int launch_sort_profile(xc_ctx* ctx, const void* q, int q_dtype, const void* mask, int mask_dtype, int mask_per_slab,
                        const double* dA, int dA_rank, int64_t nslab, int64_t ny, int64_t nx, int negate,
                        const double* targets, int J, const double* tbl, const double* coord, int ntbl,
                        void* workspace, double* out_Q, double* out_qsorted, double* out_acum,
                        unsigned* out_nvalid, double* out_bpe);
int launch_crossing(xc_ctx* ctx, const void* q, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                    int pad_x, int pad_mode, const double* contours, int N, int contours_per_slab,
                    const void* area, int area_dtype, int area_per_slab, int stride, int full_width,
                    double* out_len, uint64_t* out_cnt);
int launch_synth(xc_ctx* ctx, void* out, int q_dtype, int64_t nslab, int64_t ny, int64_t nx,
                 const double* lat_deg, const double* lon_deg, uint64_t seed, int variant);

}  // namespace xc
