// Host side of libcg_hip.so: the C ABI declared in include/cg_hip.h.
//
// One cg_ctx owns one row block of the four CG vectors in the HBM of one GPU,
// a private HIP stream, the device-resident scalar state and (optionally) a
// captured hipGraph of the iteration body.  The host never reads a scalar inside
// the iteration loop; tau, beta and the stop flag are produced and consumed on
// the device (cg_kernels.hpp, k_finish).
#include "cg_kernels.hpp"
#include "../../include/cg_hip.h"
#include "cg_comm.hpp"
#include "cg_exact.hpp"

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(CG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                \
    } while (0)

#define CGCHK(expr)                       \
    do {                                  \
        int rc_ = (expr);                 \
        if (rc_ != CG_OK) return rc_;     \
    } while (0)

using cg::kBlock;
using cg::Scalars;
using cg::StencilArgs;

// cg.c:33's coefficients {0,-1,0,-1,4,-1,0,-1,0}.
constexpr cg::Coef9 kPoisson9 = {0.0, -1.0, 0.0, -1.0, 4.0, -1.0, 0.0, -1.0, 0.0, 0};

// A 3x3 integer coefficient array (row-wise, as the reference's stencil_struct holds it).  The
// reference's kernels read entries 1, 3, 4, 5, 7 only (utils.c:27-31); the corner entries 0, 2, 6, 8
// are honoured here too, and switch the kernels to the nine-term sum when any of them is non-zero.
cg::Coef9 coef_from9(const int *s9)
{
    cg::Coef9 c{(double)s9[0], (double)s9[1], (double)s9[2], (double)s9[3], (double)s9[4],
                (double)s9[5], (double)s9[6], (double)s9[7], (double)s9[8], 0};
    c.nine = (s9[0] != 0 || s9[2] != 0 || s9[6] != 0 || s9[8] != 0) ? 1 : 0;
    return c;
}

constexpr int kMaxFlatBlocks = 512;       // two workgroups per CU: best measured HBM efficiency for mixed streams
constexpr int kMaxDevices = 64;

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int select_device(int device)
{

    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(CG_ERR_NOGPU, "no HIP device visible: the CG hot path has no CPU fallback");
    if (device < 0 || device >= count)
        return fail(CG_ERR_ARG, "device %d out of range (%d visible)", device, count);
    HIPCHK(hipSetDevice(device));
    return CG_OK;
}

// Per-device resources of the stand-alone operator entry points.
struct DeviceScratch {
    hipStream_t stream = nullptr;
    double *partials = nullptr;            // kMaxPartials doubles
    Scalars *sc = nullptr;
    double *slot = nullptr;                // strict-dot result
    cg::exact::Workspace xs;               // workspace of the parallel strict-order dot product
    bool ready = false;
};

// Strict-order dot product: below this many elements the literal one-wavefront replay is
// as fast as the five-kernel parallel evaluation.
constexpr long kExactMin = 1L << 15;
// `strict` argument of cg_dot / value of CG_OPT_STRICT_IMPL: how the reference's order is evaluated.
enum { kStrictAuto = 1, kStrictReplay = 2, kStrictParallel = 3 };

bool use_parallel_strict(int impl, long count)
{
    return impl == kStrictParallel || (impl != kStrictReplay && count >= kExactMin);
}
constexpr long kMaxPartials = 1 << 16;
DeviceScratch g_scratch[kMaxDevices];
std::mutex g_scratch_mu;

int resolve_static_scratch(int device);              // the static array of cg_stencil_noalloc: address looked up while no capture can be running

int get_scratch(int device, DeviceScratch **out)
{
    CGCHK(select_device(device));
    if (device >= kMaxDevices) return fail(CG_ERR_ARG, "device index too large");
    (void)resolve_static_scratch(device);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    DeviceScratch &s = g_scratch[device];
    if (!s.ready) {
        HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        HIPCHK(hipMalloc(&s.partials, kMaxPartials * sizeof(double)));
        HIPCHK(hipMalloc(&s.sc, sizeof(Scalars)));
        HIPCHK(hipMalloc(&s.slot, sizeof(double)));
        HIPCHK(hipMemset(s.sc, 0, sizeof(Scalars)));
        HIPCHK(hipStreamSynchronize(nullptr));
        s.ready = true;
    }
    *out = &s;
    return CG_OK;
}

int flat_blocks(long nvec, long want)
{
    long b = (nvec + kBlock - 1) / kBlock;
    const long cap = want > 0 ? want : kMaxFlatBlocks;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// Geometry of the tiled (stencil-shaped) kernels for one row block.
struct TileGeom {
    int contig = 0;   // 1: halo rows are contiguous with the block (solver context layout)
    int pf = 4;       // rows of loads in flight per thread (stand-alone entry point)
    int pf_mode[6] = {4, 4, 4, 4, 4, 4};   // ... per kernel mode (solver context)
    int pf_spec_read = 4;                  // ... of the read-only pass when it also forms the strict-order span sums
    int pf_spec_u = 1;                     // ... of pass 2 with the u update when it does (one row: measured, 1-2 % over two)
    int wg_mult = 2;                       // workgroups per CU of the read-only pass
    bool plain_order = false;              // the read-only pass takes its tiles in plain order (no XCD-aware placement)
    int vec;          // 2 when rows are 16-byte aligned (even G), else 1
    int nstrips;
    int tile_h;
    long ntiles;
    long max_wgs = 0; // > 0: at most this many workgroups per launch, each walking several tiles
};

// The tiled kernel addresses a tile's rows (and its two halo rows) through one buffer descriptor with 32-bit byte
// offsets, and masks lanes off with an offset of 0x7fffff00: a tile must stay below that many bytes.
long max_tile_rows(long G)
{
    const long fit = 0x7ff00000L / (G * 8) - 2;
    return fit < 1 ? 1 : fit;
}

TileGeom tile_geom(long G, long rows, bool all_aligned, long tile_h_opt)
{
    TileGeom t;
    t.vec = (G % 2 == 0 && all_aligned) ? 2 : 1;
    const long W = (long)kBlock * t.vec;
    t.nstrips = (int)((G + W - 1) / W);
    long h = tile_h_opt;
    if (h <= 0) {
        // One workgroup per CU (256 of them): measured on MI355X, mixed read/write
        // streams reach their best HBM efficiency with few, long-lived workgroups that
        // each keep several rows of loads in flight (5.3-5.6 TB/s at 256 workgroups
        // against 5.0-5.2 at 1024+).  Each workgroup walks tiles of about 512 rows and
        // then jumps ahead to its next tile, which keeps the 256 of them within a narrower
        // window of memory than one tall tile each would (1-2 % faster at 16384^2); the
        // two halo rows a tile re-reads stay negligible.  Heights that leave a short last
        // tile lose to imbalance, so round the tile count per strip, not the height.
        long per_strip = 256 / t.nstrips;
        if (per_strip < 1) per_strip = 1;
        const long tall = (rows + per_strip - 1) / per_strip;      // one tile per workgroup
        const long rounds = tall >= 768 ? (tall + 256) / 512 : 1;  // tiles each workgroup walks
        h = (rows + per_strip * rounds - 1) / (per_strip * rounds);
        if (h < 16) h = 16;
        t.max_wgs = (long)t.nstrips * per_strip;
    }
    if (h > max_tile_rows(G)) h = max_tile_rows(G);
    if (h > rows) h = rows;
    if (h < 1) h = 1;
    t.tile_h = (int)h;
    t.ntiles = (long)t.nstrips * ((rows + h - 1) / h);
    return t;
}

// Launches the tiled kernel over rows [row_lo, row_hi) of the block (default: all of it) in tiles of
// t.tile_h rows counted from row_lo; the tiles' partial sums go to a0.partials[part_base ...].
// Returns the number of tiles launched.
template <int MODE, bool WRITE_Q, int NH = 0>
long launch_stencil(const TileGeom &t, const StencilArgs &a0, hipStream_t s, long row_lo = 0, long row_hi = -1,
                    long part_base = 0)
{
    if (row_hi < 0) row_hi = a0.rows;
    if (row_hi <= row_lo) return 0;
    const long nrt = (row_hi - row_lo + t.tile_h - 1) / t.tile_h;
    StencilArgs a = a0;
    a.tile_h = t.tile_h;
    a.row_lo = row_lo; a.row_hi = row_hi; a.part_base = part_base;
    a.ntiles_launch = nrt * t.nstrips;
    // The read-only pass gets two workgroups per CU: measured 6.6 TB/s against 5.6 with one.
    const long cap = (MODE == 3) ? t.wg_mult * t.max_wgs : t.max_wgs;
    if (MODE == 3 && t.plain_order) a.remap = 0;
    const long nwg = (cap > 0 && cap < a.ntiles_launch) ? cap : a.ntiles_launch;
    const dim3 grid((unsigned)nwg), block(kBlock);
#define CG_LAUNCH(V, P, C) hipLaunchKernelGGL((cg::k_stencil<V, MODE, WRITE_Q, P, C, NH>), grid, block, 0, s, a)
    if constexpr (NH > 0) {           // speculative strict summaries: 16-byte rows in the context's layout only
        switch (MODE == 3 ? t.pf_spec_read : (MODE == 5 ? t.pf_spec_u : t.pf_mode[MODE])) {
        case 1: CG_LAUNCH(2, 1, true); break;
        case 4: case 8: CG_LAUNCH(2, 4, true); break;      // at most four rows in flight with the span sums
        default: CG_LAUNCH(2, 2, true); break;
        }
    } else if (!t.contig) {                  // stand-alone entry point: caller-supplied halo pointers
        if (t.vec == 2) CG_LAUNCH(2, 4, false);
        else CG_LAUNCH(1, 4, false);
    } else if (t.vec == 2) {
        switch (t.pf_mode[MODE]) {
        case 1: CG_LAUNCH(2, 1, true); break;
        case 4: CG_LAUNCH(2, 4, true); break;
        case 8: CG_LAUNCH(2, 8, true); break;
        default: CG_LAUNCH(2, 2, true); break;
        }
    } else {
        CG_LAUNCH(1, 2, true);
    }
#undef CG_LAUNCH
    return a.ntiles_launch;
}

} // namespace

// =========================================================================
// Context
// =========================================================================

struct GraphSlot {
    hipGraphExec_t exec = nullptr;
    long len = 0;                         // iterations in the captured graph
    int sched = -1, reduce = -1, dcur = -1, gcur = -1, pend = -1;   // state the capture began in
    unsigned long stamp = 0;              // last use (least recently used slot is replaced)
};

struct cg_ctx {
    int device = 0;
    long n = 0, G = 0, row0 = 0, rows = 0, count = 0;
    int rank = 0, nranks = 1;

    // options
    int reduce = CG_REDUCE_FAST;
    int schedule = -1;                    // -1 = pick from `reduce`
    long tile_h = 0;
    long graph_iters = 10;
    int remap = 1;
    long flat_blocks_opt = 0;
    int prefetch = 0;                     // 0 = per-kernel defaults (see ctx_geom)
    int prefetch_read = 0;                // read-only pass of the lean schedule; 0 = default
    int pf_update_u = 0;                  // pass 2 with the u update: what choose_placement measured (0 = default)
    long tile_wgs = 0;
    bool graph_explicit = false;          // the caller set CG_OPT_GRAPH_ITERS
    int overlap = 0;                      // multi-rank lean schedule: 1 = halo exchange beside the interior rows.  Off by default:
                                          // the one measurement there is (a single-rank communicator, the wire being the GPU's own
                                          // memory) has the fork and join cost more than they hide; bench.py times both and keeps the faster
    int rehearse = 0;                     // single rank with an RCCL communicator: issue the multi-rank traffic anyway
    int resum_q0 = 0;                     // a solve that continues the previous one sums q0 = g.g again (cg.c:55) instead of taking it over
    int strict_chain = 1;                 // several ranks, strict order, lean schedule: 1 = the ranks publish their plans' briefs and every
                                          // rank finishes the walk (two collectives per dot product); 0 = the ranks walk in turn
    int time_dir = -1;                    // cg_time_kernel, lean passes: -1 = the buffers swap roles from launch to launch (as in a solve),
                                          // 0 / 1 = every launch reads buffer pair 0 / 1 (choose_placement times the two directions apart)
    int time_warm = 2;                    // ... untimed launches first
    double pass_ms[3][2] = {{0, 0}, {0, 0}, {0, 0}};   // measured by choose_placement: lean pass (d.q, g', g' + u) x direction, 0 = not measured
    int band_left = 0;                    // strict order, lean schedule: so many of the next iterations predict the growth of d.q from a band
                                          // of rows (the band's last sum is in strict_slot[4]; cg_exact.hpp, k_band_sum)
    long band_lo = 0, band_hi = 0;
    int u_dir = -1;                       // lean schedule: the buffer pair the pass with the u update read in the last solve (-1: no solve yet);
                                          // cg_time_kernel times the two kinds of pass 2 accordingly
    bool hyp_valid = false;               // strict order: the hypotheses in the workspace stem from an iteration of the present problem
    bool lead_defers_u = false;           // lean schedule: the leading iteration of the present solve leaves its u update to the first pair
    bool strict_safe = false;             // the iteration being enqueued completes a faulted dot product: ranks walk in turn
    long fault_every = 0;                 // test hook: the published walk declares every n-th dot product failed
    long strict_faults = 0;               // dot products completed by the ranks walking in turn after the published walk gave up

    // device state
    hipStream_t stream = nullptr;
    double *u = nullptr, *g = nullptr, *q = nullptr;
    static constexpr int kMaxSlots = 2;
    double *g_alloc[2] = {nullptr, nullptr};                       // g_buf[] and d[] point one row into these
    double *d_alloc[kMaxSlots] = {nullptr};
    double *g_buf[2] = {nullptr, nullptr};                         // the lean schedule alternates between two gradients
    int gcur = 0;                                                  // which of g_buf[] is `g`
    size_t front_elems = 0;                                        // doubles between an allocation's start and row 0
    double *u_alloc = nullptr, *q_alloc = nullptr;
    size_t padded_bytes = 0;              // size of each of those allocations
    double *d[kMaxSlots] = {nullptr};     // two direction buffers: the fused and lean schedules alternate
    int dcur = 0;                         // which of d[] holds the current direction
    double *partials = nullptr;           // 2 * npart doubles (d.q | g.g)
    long npart = 0;
    Scalars *sc = nullptr;
    double *strict_slot = nullptr;
    cg::exact::Workspace xs;              // strict order, parallel evaluation
    cg::Coef9 coef = kPoisson9;           // stencil coefficients (cg_set_stencil)
    int strict_impl = kStrictAuto;
    int strict_spec = 1;                  // lean schedule, strict order: summarise the products inside the passes
    int strict_audit = 0;                 // ... and check every summary and walk against a recomputation (self-test, slow)
    long strict_blocks = 0;               // s > 1: sum in the layout of the reference on s*s ranks
    double *pack[2] = {nullptr, nullptr}; // one rank's block of v and w, contiguous
    double *block_sums = nullptr;         // its s*s sequential sums
    long pack_cap = 0;
    double *hist = nullptr;
    long hist_cap = 0;
    long hist_len = 0;
    bool q_valid = true;

    // State a solve leaves behind for the next one.  The fused and lean schedules end a solve with the last
    // d = beta*d - g (and, lean, at most one u += tau*d) still owed: a following cg_solve simply carries on
    // (those updates ride on its first iteration, as they do between any two iterations); anything that
    // looks at the vectors applies them first (materialise()).
    bool pending_dupdate = false;         // the last d update has not been applied to d[dcur]
    int lean_pending = 0;                 // lean schedule: iterations whose tau*d is not yet in u (0 or 1;
                                          // the newest direction is d[dcur])
    bool live = false;                    // q0 / beta on the device belong to the vectors' present state: a
                                          // solve may continue from them without summing g.g again
    bool needs_init = false;              // cg_time_kernel ran: the vectors are scratch until cg_init_problem
    bool d_halo_stale = false;            // several ranks: d[dcur] changed after its boundary rows last travelled (the classic
                                          // schedule exchanges them at the start of an iteration and updates d at its end)

    // halo rows (multi-GPU).  *_top = the row above the block, *_bot = below.
    double *halo_d[kMaxSlots][2] = {{nullptr, nullptr}};               // [buffer][top/bot]
    double *halo_g[2] = {nullptr, nullptr};
    cg::Comm *comm = nullptr;
    hipStream_t comm_stream = nullptr;    // halo exchange runs here, beside the interior rows
    hipEvent_t ev_fork = nullptr, ev_halo = nullptr;

    // captured iteration bodies, one per starting state that has occurred
    GraphSlot graphs[4];
    unsigned long graph_clock = 0;
    Scalars *host_sc = nullptr;           // pinned read-back buffer

    bool lean_unavailable = false;        // not enough free HBM for the lean schedule's second gradient buffer
    int effective_schedule() const
    {
        if (schedule >= 0) return schedule;
        return lean_unavailable ? CG_SCHED_FUSED_NOQ : CG_SCHED_LEAN;
    }
    void set_gcur(int k)
    {
        gcur = k; g = g_buf[k];
        halo_g[0] = g - G; halo_g[1] = g + count;
    }
    bool has_top() const { return row0 > 0; }
    bool has_bot() const { return row0 + rows < G; }
    bool multi() const { return nranks > 1 || rehearse != 0; }      // dot products and halo rows go through the communicator
};

namespace {

TileGeom ctx_geom(const cg_ctx *c)
{
    TileGeom t = tile_geom(c->G, c->rows, true, c->tile_h);
    // Rows of loads in flight per thread.  Measured at 16384^2: the mixed read/write passes are fastest with
    // LITTLE in flight: two rows for 2R2W (5.85 TB/s against 5.6 at 4 and 4.6 at 1); for 3R3W two rows on most
    // grids, blocks and placements (2-5 % over one row), one row on some placements of the 16384^2 grid (1.5 %
    // the other way), so a context that times its placement times both (choose_placement).  The pure read pass
    // with two workgroups per CU: four rows (1.5-4 % over eight from 2048-row blocks to 65536^2; with one
    // workgroup per CU it wanted eight).
    static const int auto_pf[6] = {4, 4, 4, 4, 2, 2};
    for (int m = 0; m < 6; ++m) t.pf_mode[m] = c->prefetch > 0 ? c->prefetch : auto_pf[m];
    if (c->prefetch == 0 && c->pf_update_u > 0) t.pf_mode[5] = c->pf_update_u;
    if (c->prefetch_read > 0) t.pf_mode[3] = c->prefetch_read;
    // With the span sums of the strict order the read-only pass is bound by its instruction stream and its
    // registers, not by memory latency: two rows in flight leave it 147 registers, enough for three workgroups
    // per CU (dq_geom), and the four-row batched reduction through LDS does the rest.
    t.pf_spec_read = c->prefetch_read > 0 ? c->prefetch_read : (c->prefetch > 0 ? c->prefetch : 2);
    t.pf_spec_u = c->prefetch > 0 ? c->prefetch : 1;
    t.pf = t.pf_mode[0];
    if (c->tile_wgs > 0) t.max_wgs = c->tile_wgs;
    t.contig = 1;
    return t;
}

// Geometry of the kernels that produce d.q (MODE 0, 1, 3).  The lean schedule's first pass only
// reads and is fastest with two workgroups per CU; a row block too short to give it that many
// tiles (one rank's share of a multi-GPU run) gets tiles of half the height there: 83 instead of
// 92 us at 2048 x 16384.  The g.g kernels keep `t`; each dot product has its own partials.
bool strict_speculative(const cg_ctx *c, const TileGeom &t);

TileGeom dq_geom(const cg_ctx *c, TileGeom t)
{
    // Strict order with span sums formed in the pass: three workgroups per CU.  That pass is bound by its
    // instruction stream, not by memory, and wants another geometry than the others -- measured at 16384^2:
    // tiles of 256 rows taken in plain order 0.90 ms, the same with the XCD-aware placement 1.00 ms (tiles of 512
    // rows: 1.20 ms), two tiles of 342 rows per workgroup 0.93 ms either way.  Blocks too small for two rounds of
    // 256-row tiles get one or two tiles per workgroup.
    if (c->tile_h == 0 && t.max_wgs > 0 && c->reduce == CG_REDUCE_STRICT && t.pf_spec_read <= 2 && strict_speculative(c, t)) {
        long per_strip = 3 * t.max_wgs / t.nstrips;
        if (per_strip < 1) per_strip = 1;
        long h = 256;
        if ((long)t.nstrips * ((c->rows + h - 1) / h) < 6 * t.max_wgs) {
            h = (c->rows + 2 * per_strip - 1) / (2 * per_strip);
            if (h < 32) h = (c->rows + per_strip - 1) / per_strip;
        }
        t.plain_order = true;
        if (h < 16) h = 16;
        if (h > max_tile_rows(c->G)) h = max_tile_rows(c->G);
        if (h > c->rows) h = c->rows;
        t.tile_h = (int)h;
        t.ntiles = (long)t.nstrips * ((c->rows + h - 1) / h);
        t.wg_mult = 3;
        return t;
    }
    if (c->tile_h == 0 && t.max_wgs > 0 && t.ntiles < 2 * t.max_wgs && t.tile_h >= 64) {
        const long h = (t.tile_h + 1) / 2;
        t.tile_h = (int)h;
        t.ntiles = (long)t.nstrips * ((c->rows + h - 1) / h);
    }
    return t;
}

int comm_rc(const cg_ctx *c, int rc)
{
    return rc == 0 ? CG_OK : fail(CG_ERR_COMM, "rank %d: %s", c->rank, c->comm->error());
}

void drop_graph(cg_ctx *c)
{
    for (GraphSlot &g : c->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        g = GraphSlot();
    }
}

// ---- one CG iteration, enqueued on the context's stream -------------------
//
// Every launch is stream-ordered; nothing here synchronises with the host (the host-staged test
// transport excepted), so the same code runs eagerly and under stream capture.
//
// Where a kernel puts the `npart` block partials of a dot product: the context's own array, or, in a
// multi-rank context, this rank's slice of the all-gather buffer (see enqueue_dot_finish).
int rank_partials(cg_ctx *c, double *own, long npart, double **out)
{
    if (npart > c->npart)
        return fail(CG_ERR_ARG, "%ld tiles exceed the %ld partial sums the context holds: use a larger CG_OPT_TILE_H", npart, c->npart);
    if (!c->multi() || c->reduce == CG_REDUCE_STRICT) { *out = own; return CG_OK; }
    if (npart > cg::Comm::kBlockCap)
        return fail(CG_ERR_ARG, "%ld tiles per rank exceed the all-gather buffer (%ld): use a larger CG_OPT_TILE_H",
                    npart, cg::Comm::kBlockCap);
    *out = c->comm->block_buf() + (long)c->rank * npart;
    return CG_OK;
}

// Strict order, parallel evaluation: may the kernel that produces the operands hand over
// approximate span sums (cg::exact::kSub products each)?  Needs 16-byte rows whose length is a
// multiple of the span, so that spans never straddle rows; not used with the 2-D block layout.
double *strict_sub(const cg_ctx *c, const TileGeom &t)
{
    if (c->reduce != CG_REDUCE_STRICT || c->strict_blocks > 1) return nullptr;
    if (!use_parallel_strict(c->strict_impl, c->count)) return nullptr;
    if (t.vec != 2 || c->G % cg::exact::kSub != 0) return nullptr;
    return c->xs.sub;
}

// Strict order in the lean schedule: the two passes summarise their products for the sequential sum
// while they form them (two hypothesised exponents per span, cg_exact.hpp), instead of storing the products
// d'*q and reading them and g' once more.  Needs rows that are whole chunks.
bool strict_speculative(const cg_ctx *c, const TileGeom &t)
{
    return c->strict_spec && strict_sub(c, t) != nullptr && c->G % cg::exact::kChunk == 0 &&
           c->xs.cap_spec >= cg::exact::Workspace::chunks_for(c->count);
}

// A strict-order dot product over the ranks' row blocks, which follow one another in the reference's index order:
// every rank forms its approximate sums and, seeded with the approximate totals of the ranks before it (one
// all-gather), its summaries and its plan -- all ranks at once; only the walks go in turn, the exact accumulator
// handed from rank to rank by one all-gather of a double per round.  approx(total_out), summaries(seed, nseed) and
// walk(start_dev, result) enqueue the three phases; *result_at is where the final value is on this rank.
// The last phase alone: the ranks' walks in turn, rank k continuing from rank k-1's exact value.
template <class FW>
int strict_chain_walks(cg_ctx *c, FW walk, const double **result_at)
{
    hipStream_t s = c->stream;
    double *gb = c->comm->gather_buf();
    for (int k = 0; k < c->nranks; ++k) {
        if (k == c->rank) {
            walk(k > 0 ? (const double *)(gb + (k - 1)) : (const double *)nullptr, gb + c->rank);
            HIPCHK(hipGetLastError());
        }
        CGCHK(comm_rc(c, c->comm->allgather_scalar(s)));
    }
    // a single rank rehearsing the traffic of `rehearse` ranks: the all-gathers the other ranks' turns would bring
    for (int k = c->nranks; k < c->rehearse; ++k) CGCHK(comm_rc(c, c->comm->allgather_scalar(s)));
    *result_at = gb + (c->nranks - 1);
    return CG_OK;
}

template <class FA, class FS, class FW>
int strict_rank_chain(cg_ctx *c, FA approx, FS summaries, FW walk, const double **result_at)
{
    hipStream_t s = c->stream;
    if (c->nranks == 1 && c->rehearse <= 1) {
        approx((double *)nullptr);
        summaries((const double *)nullptr, 0);
        walk((const double *)nullptr, c->strict_slot);
        HIPCHK(hipGetLastError());
        *result_at = c->strict_slot;
        return CG_OK;
    }
    double *gb = c->comm->gather_buf();
    approx(gb + c->rank);
    HIPCHK(hipGetLastError());
    CGCHK(comm_rc(c, c->comm->allgather_scalar(s)));
    summaries((const double *)gb, c->rank);
    HIPCHK(hipGetLastError());
    return strict_chain_walks(c, walk, result_at);
}

// The same dot product with O(1) collectives (lean schedule with span sums only; cg_exact.hpp, k_xs_walk_remote): after
// the approximate totals (one all-gather) every rank builds its plan; rank 0 -- whose block holds nearly all binade
// crossings -- walks its own block; one all-gather publishes rank 0's exact accumulator and every rank's briefs; every
// rank finishes the walk through the other ranks' briefs.  phase: 0 = d.q, 1 = g.g (what a fault reports).  partials2 /
// count2: this rank's q.q partial sums (prediction input, summed over the ranks on the way), or null.  The result lands
// in strict_slot[0], the grid's q.q in strict_slot[1].  If a brief does not hold for the true accumulator the kernel
// raises Scalars::done = 2 instead, and cg_solve completes the dot product with strict_chain_walks.
template <class FA, class FS, class FW>
int strict_chain_published(cg_ctx *c, FA approx, FS summaries, FW walk, int phase, const double *partials2, long count2,
                           const cg::exact::FinishArgs &fin, const double **result_at)
{
    hipStream_t s = c->stream;
    const bool self = c->comm->self_only();             // timing rehearsal: this rank stands in for all
    double *gb = c->comm->gather_buf();
    approx(gb + c->rank);
    HIPCHK(hipGetLastError());
    CGCHK(comm_rc(c, c->comm->allgather_scalar(s)));
    summaries((const double *)gb, c->rank);
    HIPCHK(hipGetLastError());
    const bool walker = self || c->rank == 0;
    if (walker) walk((const double *)nullptr, c->strict_slot);
    double *slots = c->comm->block_buf();
    static_assert(cg::exact::kPayload <= cg::Comm::kBlockCap, "a rank's published briefs must fit its slice of the all-gather buffer");
    hipLaunchKernelGGL(cg::exact::k_xs_pack, dim3(1), dim3(64), 0, s, c->xs.pn, c->xs.dump, walker ? (const double *)c->strict_slot : (const double *)nullptr,
                       partials2, count2, slots + (long)c->rank * cg::exact::kPayload, c->xs.skip);
    HIPCHK(hipGetLastError());
    CGCHK(comm_rc(c, c->comm->allgather_block(s, cg::exact::kPayload)));
    hipLaunchKernelGGL(cg::exact::k_xs_walk_remote, dim3(1), dim3(64), 0, s, self ? slots + (long)c->rank * cg::exact::kPayload : slots,
                       self ? 1 : c->nranks, self ? 1 : 0, c->sc, phase, c->fault_every, c->strict_slot, c->strict_slot + 1, fin);
    HIPCHK(hipGetLastError());
    *result_at = c->strict_slot;
    return CG_OK;
}

int enqueue_dot_finish(cg_ctx *c, const double *v, const double *w, double *partials, long npart,
                       int what, double rtol, int have_sub = 0)   // 1: xs.sub holds span sums, linear; 2: by rows
{
    // Completes a dot product whose fast-order block partials are already in
    // `partials` (npart of them), or, in strict mode, recomputes it with the
    // sequential kernel from v and w (w == nullptr: v already holds the products); then runs
    // the scalar recurrence `what`.
    hipStream_t s = c->stream;
    const double *src = nullptr;
    int nsrc = 1;
    if (c->reduce == CG_REDUCE_STRICT) {
        // One accumulator over the whole grid in index order.  Row blocks are contiguous
        // in that order, so with several ranks the accumulator is handed from rank to rank:
        // in round k rank k continues from rank k-1's value, and an all-gather of one double
        // per rank publishes it.  Every rank ends with the same bits a single rank would compute.
        // Only the ordered walks take turns; what they walk over is prepared on all ranks at once
        // (strict_rank_chain).
        if (c->strict_blocks > 1) {
            // The reference's own multi-rank layout on this one GPU: rank r = ci*s + cj of its s x s
            // decomposition sums the products of its (G/s)^2 block sequentially in local row-wise
            // order (utils.c:372-379); MPI_Allreduce combines the sums (utils.c:380).
            const long sdim = c->strict_blocks, cl = c->G / sdim, n2 = cl * cl;
            const bool par = use_parallel_strict(c->strict_impl, n2);
            const int pb = flat_blocks(n2, 0);
            for (long r = 0; r < sdim * sdim; ++r) {
                const long ci = r / sdim, cj = r % sdim;
                hipLaunchKernelGGL(cg::k_pack_block, dim3(pb), dim3(kBlock), 0, s, v, c->G, ci * cl, cj * cl, cl, c->pack[0]);
                if (w != v && w != nullptr)
                    hipLaunchKernelGGL(cg::k_pack_block, dim3(pb), dim3(kBlock), 0, s, w, c->G, ci * cl, cj * cl, cl, c->pack[1]);
                const double *pv = c->pack[0], *pw = (w == nullptr) ? nullptr : ((w != v) ? c->pack[1] : c->pack[0]);   // null: v holds products
                if (par) cg::exact::enqueue(s, c->xs, pv, pw, n2, 0.0, c->block_sums + r);
                else hipLaunchKernelGGL(cg::k_dot_strict, dim3(1), dim3(64), 0, s, pv, pw, n2, 0.0, nullptr, c->block_sums + r);
            }
            hipLaunchKernelGGL(cg::k_tree_sum, dim3(1), dim3(1), 0, s, c->block_sums, (int)(sdim * sdim), c->strict_slot);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(cg::k_finish, dim3(1), dim3(kBlock), 0, s, partials, npart, c->strict_slot, 1, c->sc, what, rtol,
                               c->hist, c->hist_cap);
            return CG_OK;
        }
        const bool par = use_parallel_strict(c->strict_impl, c->count);
        if (par) {                                             // workspace reserved by cg_solve
            const cg::exact::Arrays ar{v, w, c->count, have_sub != 0, have_sub == 2 ? c->G : 0, c->rows};
            CGCHK(strict_rank_chain(
                c, [&](double *total) { cg::exact::enqueue_approx(s, c->xs, ar, total); },
                [&](const double *seed, int nseed) { cg::exact::enqueue_summaries(s, c->xs, ar, 0.0, seed, nseed); },
                [&](const double *start_dev, double *result) {
                    cg::exact::enqueue_walk(s, c->xs, ar.src(), ar.count, 0.0, start_dev, result, c->xs.stats);
                },
                &src));
        } else {
            // the literal replay: the ranks take turns from beginning to end
            for (int k = 0; k < c->nranks; ++k) {
                double *slot = c->nranks > 1 ? c->comm->gather_buf() + c->rank : c->strict_slot;
                const double *seed = k > 0 ? c->comm->gather_buf() + (k - 1) : nullptr;
                if (k == c->rank)
                    hipLaunchKernelGGL(cg::k_dot_strict, dim3(1), dim3(64), 0, s, v, w, c->count, 0.0, seed, slot);
                if (c->nranks > 1) CGCHK(comm_rc(c, c->comm->allgather_scalar(s)));
            }
            src = c->nranks > 1 ? c->comm->gather_buf() + (c->nranks - 1) : c->strict_slot;
        }
    } else if (c->multi()) {
        // The kernels of a multi-rank context write their partials straight into this rank's slice of
        // the all-gather buffer (rank_partials); one RCCL all-gather concatenates the ranks' slices in
        // rank order, and k_finish adds the whole concatenation in its fixed order on every rank.
        if (partials != c->comm->block_buf() + (long)c->rank * npart)
            return fail(CG_ERR_STATE, "multi-rank dot product: partials are not in the all-gather buffer");
        CGCHK(comm_rc(c, c->comm->allgather_block(s, npart)));
        partials = c->comm->block_buf();
        npart *= c->nranks;
    }
    hipLaunchKernelGGL(cg::k_finish, dim3(1), dim3(kBlock), 0, s, partials, npart, src, nsrc, c->sc, what, rtol,
                       c->hist, c->hist_cap);
    return CG_OK;
}

StencilArgs base_args(cg_ctx *c, const TileGeom &t)
{
    StencilArgs a;
    memset(&a, 0, sizeof a);
    a.sc = c->sc;
    a.coef = c->coef;
    a.G = c->G; a.rows = c->rows; a.row0 = c->row0;
    a.nstrips = t.nstrips; a.tile_h = t.tile_h; a.remap = c->remap;
    return a;
}

// The neighbours' boundary rows of g into g's halo rows (the fused and lean schedules keep d's halo rows
// current locally, so g's rows are all that travels).  On a single rank that rehearses the multi-rank
// traffic the same two rows are sent to the rank itself, into a scratch row.
int exchange_g_rows(cg_ctx *c, hipStream_t on, double *gbuf)
{
    if (c->nranks > 1)
        return comm_rc(c, c->comm->exchange_rows(on, gbuf, c->rows, c->G, gbuf - c->G, gbuf + c->count));
    if (c->rehearse) return comm_rc(c, c->comm->rehearse_rows(on, gbuf, c->rows, c->G));
    return CG_OK;
}

// u updates the lean schedule still owes when a solve ends (at most two): u += tau*d, oldest first.
int enqueue_uflush(cg_ctx *c, int pending, int newest_slot)
{
    if (pending <= 0) return CG_OK;
    if (pending > 2) return fail(CG_ERR_STATE, "lean schedule: %d deferred u updates", pending);
    const TileGeom t = ctx_geom(c);
    const int fb = flat_blocks(c->count / t.vec, c->flat_blocks_opt);
    cg::UFlushArgs f;
    memset(&f, 0, sizeof f);
    f.n = pending;
    for (int e = 0; e < pending; ++e) f.d[e] = c->d[(newest_slot + (pending - 1 - e)) & 1];
    if (t.vec == 2) hipLaunchKernelGGL(cg::k_uflush<2>, dim3(fb), dim3(kBlock), 0, c->stream, c->u, c->count, f, c->sc);
    else hipLaunchKernelGGL(cg::k_uflush<1>, dim3(fb), dim3(kBlock), 0, c->stream, c->u, c->count, f, c->sc);
    hipLaunchKernelGGL(cg::k_umark, dim3(1), dim3(1), 0, c->stream, c->sc, pending);
    HIPCHK(hipGetLastError());
    return CG_OK;
}

// Strict order: the scalar recurrences predict the growth of d.q from one iteration to the next well enough for the
// hypotheses of the span sums (cg_exact.hpp), but not behind the very first iteration of a problem (the first step length
// is orders of magnitude off the rest): a wrong prediction has every chunk of the pass summarised twice, two milliseconds
// at 16384^2.  So the iteration behind the first one measures the growth instead, on a band of 64 rows ahead of its pass 1
// (two small launches, ~13 us), and runs eagerly.  Only that one: measured over the first 32 iterations the band's ratio
// did worse than the recurrences once the step lengths alternate (the shape of the prefix sums alternates with them),
// and a few iterations around the 20th, where that pattern sets in, miss either way.
constexpr int kBandIters = 1;

// Rows at each end of a rank's block that the lean schedule's second pass computes in launches of their own,
// ahead of the rest, so that the new gradient's boundary rows can travel while the other rows are computed.
constexpr long kEdgeRows = 8;

// Steady-state iteration of the lean schedule.  State on entry: d[dcur] = d_{i-1}, g = g_i (halo rows
// included), beta_i on the device; d_i = beta_i*d_{i-1} - g_i is never loaded back before it is used:
//   pass 1 (reads d_{i-1}, g_i):            d_i on the fly, partials of d_i . A d_i       -> tau_i
//   pass 2 (reads d_{i-1}, g_i once more):  d_i again -> the other direction buffer,
//                                           g_{i+1} = g_i + tau_i*A d_i -> other gradient buffer,
//                                           partials of g_{i+1} . g_{i+1}                 -> beta_{i+1}
//   every second pass 2 also:               u = (u + tau_{i-1}*d_{i-1}) + tau_i*d_i  (both operands are at hand)
// 16 + 32 + 16/2 = 56 B per point against 64 for the schedule that stores d' between the passes, and
// pass 1 is a pure read stream.  Same arithmetic in the same order: bit-identical vectors.
// Several ranks: pass 2 computes the block's first and last kEdgeRows rows in two small launches on the
// communication stream, which then sends their g_{i+1} rows to the neighbours, all of it beside the large
// launch for the rows in between on the main stream (the reference's post - interior - wait - edges of
// utils.c:45-104, with the edges first because it is their OUTPUT that travels here); the streams join
// before the dot product is completed, so every iteration is self-contained (capturable as it stands).
// lead: the iteration that starts a solve, in this form (single rank, strict order with hypotheses at hand): d[dcur] holds
// the direction itself, so pass 1 reads it alone (plain_d), and pass 2 is the in-place update of the schedule without q
// (MODE 2: recompute q, g += tau*q, u += tau*d unless cg_solve defers that) with the span sums formed inside; nothing
// changes roles.
int enqueue_lean_iteration(cg_ctx *c, bool lead = false)
{
    hipStream_t s = c->stream;
    const TileGeom t = ctx_geom(c), tq = dq_geom(c, t);
    const bool overlap = !lead && c->multi() && c->overlap && c->rows >= 4 * kEdgeRows;
    TileGeom ti = t;                                       // pass 2, rows between the edge strips
    long n_gg = t.ntiles;
    if (overlap) {
        ti = tile_geom(c->G, c->rows - 2 * kEdgeRows, true, c->tile_h);
        ti.contig = 1; ti.pf = t.pf; ti.max_wgs = t.max_wgs;
        for (int m = 0; m < 6; ++m) ti.pf_mode[m] = t.pf_mode[m];
        ti.pf_spec_read = t.pf_spec_read; ti.pf_spec_u = t.pf_spec_u;
        n_gg = 2 * (long)t.nstrips + ti.ntiles;
    }
    double *p_dq, *p_gg;
    CGCHK(rank_partials(c, c->partials, tq.ntiles, &p_dq));
    CGCHK(rank_partials(c, c->partials + c->npart, n_gg, &p_gg));
    const int in = c->dcur, out = lead ? c->dcur : c->dcur ^ 1;
    StencilArgs a = base_args(c, tq);
    a.partials = p_dq;
    a.d_in = c->d[in]; a.g = c->g;
    a.plain_d = lead ? 1 : 0;
    a.d_top = c->has_top() ? c->halo_d[in][0] : nullptr;
    a.d_bot = c->has_bot() ? c->halo_d[in][1] : nullptr;
    a.g_top = c->has_top() ? c->halo_g[0] : nullptr;
    a.g_bot = c->has_bot() ? c->halo_g[1] : nullptr;
    // Strict order: the pass also stores the rounded products d'*q (in the q array) and span sums of
    // them; the sequential sum is then evaluated from the products alone.
    const bool strict = c->reduce == CG_REDUCE_STRICT;
    double *const sub = strict_sub(c, t);
    const bool spec = strict && strict_speculative(c, t);
    a.sub = sub;
    a.rec = c->xs.rec;
    if (!spec) c->band_left = 0;
    if (spec) {
        if (c->band_left > 0) {
            // the first iterations behind a leading one: the growth of d.q from a band of rows, evaluated ahead of the pass
            --c->band_left;
            StencilArgs ab = a;
            ab.sub = nullptr; ab.partials = c->partials + 2 * c->npart;
            const long nb = launch_stencil<3, false>(tq, ab, s, c->band_lo, c->band_hi, 0);
            hipLaunchKernelGGL(cg::exact::k_band_rho, dim3(1), dim3(64), 0, s, (const double *)ab.partials, nb,
                               c->strict_slot + 4, c->sc, c->xs.skip);
            HIPCHK(hipGetLastError());
        }
        a.hyp = c->xs.hyp[0]; a.hyp_which = 0;
        a.partials2 = c->partials + 2 * c->npart;
        launch_stencil<3, false, 2>(tq, a, s);
        HIPCHK(hipGetLastError());
        cg::exact::SrcStencil src;
        src.d = c->d[in]; src.g = c->g; src.sc = c->sc; src.coef = c->coef;
        src.G = c->G; src.rows = c->rows; src.row0 = c->row0; src.count = c->count; src.beta = 0.0;
        src.plain = lead ? 1 : 0;
        const double *at = nullptr;
        auto f_approx = [&](double *total) { cg::exact::spec_approx(s, c->xs, c->G, c->rows, total); };
        auto f_summ = [&](const double *seed, int nseed) {
            cg::exact::spec_summaries(s, c->xs, src, c->G, c->rows, 0, false, &c->sc->rho[0], 0.0, seed, nseed);
        };
        // the scalar step (tau, the growth prediction for g.g) rides in the kernel that completes the dot product: the walk of
        // a single rank, the published walk of several; the ranks walking in turn (and the self-check) keep k_finish
        const bool published = c->multi() && c->strict_chain && !c->strict_safe && !c->strict_audit;
        const bool fused_finish = !c->strict_audit && (published || !c->multi());
        cg::exact::FinishArgs fin;
        fin.sc = c->sc; fin.what = 0; fin.hist = c->hist; fin.hist_cap = c->hist_cap;
        fin.partials2 = a.partials2; fin.count2 = tq.ntiles;
        auto f_walk = [&](const double *start_dev, double *result) {
            cg::exact::enqueue_walk(s, c->xs, src, c->count, 0.0, start_dev, result, c->xs.stats,
                                    (fused_finish && !published) ? fin : cg::exact::FinishArgs());
            if (c->strict_audit) cg::exact::enqueue_audit(s, c->xs, src, c->count, 0.0, start_dev, result);
        };
        if (published) CGCHK(strict_chain_published(c, f_approx, f_summ, f_walk, 0, a.partials2, tq.ntiles, fin, &at));
        else CGCHK(strict_rank_chain(c, f_approx, f_summ, f_walk, &at));
        // (the growth prediction for g.g needs q.q of the whole grid: this rank's partial sums, or, published walk, the
        //  ranks' sums added up on the way; the ranks walking in turn fall back on the last beta)
        if (!fused_finish)
            hipLaunchKernelGGL(cg::k_finish, dim3(1), dim3(kBlock), 0, s, p_dq, tq.ntiles, at, 1, c->sc, 0, 0.0,
                               c->hist, c->hist_cap, c->nranks == 1 ? (const double *)a.partials2 : (const double *)nullptr,
                               c->nranks == 1 ? tq.ntiles : -1L);
    } else {
        if (strict) {
            a.q = c->q; a.store_products = 1;
            launch_stencil<3, true>(tq, a, s);
        } else {
            launch_stencil<3, false>(tq, a, s);
        }
        HIPCHK(hipGetLastError());
        CGCHK(enqueue_dot_finish(c, strict ? c->q : nullptr, nullptr, p_dq, tq.ntiles, 0, 0.0, sub ? 2 : 0));
    }
    StencilArgs b = a;
    b.store_products = 0;
    b.partials = p_gg;
    b.d_out = c->d[out];
    b.d_top_out = c->has_top() ? c->halo_d[out][0] : nullptr;
    b.d_bot_out = c->has_bot() ? c->halo_d[out][1] : nullptr;
    b.g_out = lead ? c->g : c->g_buf[c->gcur ^ 1];
    b.u = c->u;
    // the previous iteration's u update is outstanding: do both (a leading iteration: its own, unless it is deferred)
    const bool with_u = lead ? !c->lead_defers_u : c->lean_pending == 1;
    if (spec) { b.hyp = c->xs.hyp[1]; b.hyp_which = 1; }
    if (lead) { b.plain_d = 0; b.gw = c->g; b.d_out = nullptr; if (!with_u) b.u = nullptr; }
    auto pass2 = [&](hipStream_t on, const TileGeom &tg, long lo, long hi, long base) -> long {
        if (lead) return launch_stencil<2, false, 2>(tg, b, on, lo, hi, base);
        if (spec) return with_u ? launch_stencil<5, false, 2>(tg, b, on, lo, hi, base) : launch_stencil<4, false, 2>(tg, b, on, lo, hi, base);
        return with_u ? launch_stencil<5, false>(tg, b, on, lo, hi, base) : launch_stencil<4, false>(tg, b, on, lo, hi, base);
    };
    if (overlap) {
        // Fork: the communication stream computes the edge rows (two small launches that run beside the large
        // one) and sends their gradient rows off; the main stream computes the rows in between.  Join before
        // the dot product is completed: it needs the edge tiles' partial sums, and by then the rows have landed.
        TileGeom te = t;                                   // one tile per strip at each end
        te.tile_h = (int)kEdgeRows;
        HIPCHK(hipEventRecord(c->ev_fork, s));
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_fork, 0));
        pass2(c->comm_stream, te, 0, kEdgeRows, 0);
        pass2(c->comm_stream, te, c->rows - kEdgeRows, c->rows, t.nstrips);
        HIPCHK(hipGetLastError());
        CGCHK(exchange_g_rows(c, c->comm_stream, b.g_out));
        HIPCHK(hipEventRecord(c->ev_halo, c->comm_stream));
        pass2(s, ti, kEdgeRows, c->rows - kEdgeRows, 2 * (long)t.nstrips);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamWaitEvent(s, c->ev_halo, 0));
    } else {
        pass2(s, t, 0, c->rows, 0);
        HIPCHK(hipGetLastError());
        if (c->multi()) CGCHK(exchange_g_rows(c, s, b.g_out));
    }
    if (spec) {
        const int vec = ((reinterpret_cast<uintptr_t>(b.g_out)) & 15) == 0 ? 1 : 0;
        const cg::exact::SrcArrays src{b.g_out, b.g_out, c->count, vec};
        const double *at = nullptr;
        auto f_approx = [&](double *total) { cg::exact::spec_approx(s, c->xs, c->G, c->rows, total); };
        auto f_summ = [&](const double *seed, int nseed) {
            cg::exact::spec_summaries(s, c->xs, src, c->G, c->rows, 1, true, &c->sc->rho[1], 0.0, seed, nseed);
        };
        const bool published = c->multi() && c->strict_chain && !c->strict_safe && !c->strict_audit;
        const bool fused_finish = !c->strict_audit && (published || !c->multi());
        cg::exact::FinishArgs fin;
        fin.sc = c->sc; fin.what = with_u ? 1 : 4; fin.hist = c->hist; fin.hist_cap = c->hist_cap;
        auto f_walk = [&](const double *start_dev, double *result) {
            cg::exact::enqueue_walk(s, c->xs, src, c->count, 0.0, start_dev, result, c->xs.stats + 8,
                                    (fused_finish && !published) ? fin : cg::exact::FinishArgs());
            if (c->strict_audit) cg::exact::enqueue_audit(s, c->xs, src, c->count, 0.0, start_dev, result);
        };
        if (published) CGCHK(strict_chain_published(c, f_approx, f_summ, f_walk, 1, (const double *)nullptr, 0L, fin, &at));
        else CGCHK(strict_rank_chain(c, f_approx, f_summ, f_walk, &at));
        if (!fused_finish)
            hipLaunchKernelGGL(cg::k_finish, dim3(1), dim3(kBlock), 0, s, p_gg, n_gg, at, 1, c->sc, with_u ? 1 : 4, 0.0,
                               c->hist, c->hist_cap);
    } else {
        CGCHK(enqueue_dot_finish(c, b.g_out, b.g_out, p_gg, n_gg, with_u ? 1 : 4, 0.0, sub ? 2 : 0));
    }
    if (spec) c->hyp_valid = true;
    if (lead) {
        c->lean_pending = with_u ? 0 : 1;
        return CG_OK;
    }
    c->dcur = out;
    c->set_gcur(c->gcur ^ 1);
    c->lean_pending = with_u ? 0 : 1;
    return CG_OK;
}

// A lean iteration whose published walk (strict_chain_published) gave up on dot product `phase` (0 = d.q, 1 = g.g): every
// kernel launched after it returned at once, so the vectors, the device scalars and the strict-order workspace are as that
// dot product left them, and the context's buffer roles have been set back to those before the iteration.  Complete the
// iteration with the ranks walking in turn.  Collective: every rank saw the same fault (the walk that raised it ran on
// identical data everywhere).
int recover_faulted_iteration(cg_ctx *c, int phase)
{
    hipStream_t s = c->stream;
    HIPCHK(hipMemsetAsync(&c->sc->done, 0, 2 * sizeof(int), s));       // done and fault
    ++c->strict_faults;
    if (phase == 0) {
        // pass 1 only reads: the whole iteration once more, its dot products handed from rank to rank
        c->strict_safe = true;
        const int rc = enqueue_lean_iteration(c);
        c->strict_safe = false;
        return rc;
    }
    // pass 2 has run (the new direction and gradient are in the other buffers, u is updated where it was due); the
    // summaries and the plan of g'.g' are in the workspace: only the ordered walks and the scalar step are missing
    const bool with_u = c->lean_pending == 1;
    double *g_out = c->g_buf[c->gcur ^ 1];
    const int vec = ((reinterpret_cast<uintptr_t>(g_out)) & 15) == 0 ? 1 : 0;
    const cg::exact::SrcArrays src{g_out, g_out, c->count, vec};
    const double *at = nullptr;
    CGCHK(strict_chain_walks(
        c, [&](const double *start_dev, double *result) { cg::exact::enqueue_walk(s, c->xs, src, c->count, 0.0, start_dev, result, c->xs.stats + 8); },
        &at));
    hipLaunchKernelGGL(cg::k_finish, dim3(1), dim3(kBlock), 0, s, c->partials + c->npart, 0L, at, 1, c->sc, with_u ? 1 : 4, 0.0,
                       c->hist, c->hist_cap);
    HIPCHK(hipGetLastError());
    c->dcur ^= 1;
    c->set_gcur(c->gcur ^ 1);
    c->lean_pending = with_u ? 0 : 1;
    return CG_OK;
}

// An iteration of the classic / fused schedules, or (`first`) the iteration that starts from a materialised
// d -- no d update owed -- in any schedule; the lean schedule runs it in the fused_noq form.
int enqueue_iteration(cg_ctx *c, bool first)
{
    hipStream_t s = c->stream;
    int sched = c->effective_schedule();
    const bool lean = sched == CG_SCHED_LEAN;
    if (lean) {
        if (!first) return enqueue_lean_iteration(c);
        // d is materialised.  With hypotheses of this problem at hand (a solve that continues another one) the iteration
        // runs in the lean form all the same; else as stencil pass + in-place update pass with stored products.
        if (c->hyp_valid && !c->multi() && c->reduce == CG_REDUCE_STRICT && c->strict_blocks <= 1 &&
            strict_speculative(c, ctx_geom(c)))
            return enqueue_lean_iteration(c, true);
        sched = CG_SCHED_FUSED_NOQ;
    }
    // The lean schedule applies u's updates in pairs; cg_solve may ask the leading iteration to leave its own to the
    // first pair (then a solve of an even number of iterations ends with none left over).
    const bool defer_u = lean && first && c->lead_defers_u;
    const TileGeom t = ctx_geom(c), tq = dq_geom(c, t);
    const long nvec = c->count / t.vec;
    const int fb = flat_blocks(nvec, c->flat_blocks_opt);
    const long npart_gg = (sched == CG_SCHED_FUSED_NOQ) ? t.ntiles : (long)fb;   // tiled or flat update kernel
    double *p_dq, *p_gg;
    CGCHK(rank_partials(c, c->partials, tq.ntiles, &p_dq));
    CGCHK(rank_partials(c, c->partials + c->npart, npart_gg, &p_gg));

    // ---- q = A d (+ d.q), with the pending d update folded in where scheduled
    StencilArgs a = base_args(c, tq);
    a.partials = p_dq;
    a.q = c->q;
    double *const sub = strict_sub(c, t);                  // both dot products use it, one after the other
    a.sub = sub;
    // FUSED_NOQ keeps no q.  The strict order needs the products d*q in index order, not q itself: there
    // the stencil kernel stores them (rounded once, as the reference's loop rounds them) where q would go.
    const bool products = (sched == CG_SCHED_FUSED_NOQ) && c->reduce == CG_REDUCE_STRICT;
    a.store_products = products ? 1 : 0;
    const bool fuse = (sched != CG_SCHED_CLASSIC) && !first;
    bool split = false;                   // interior tiles run beside the halo exchange
    const long nrt = tq.ntiles / tq.nstrips;               // row tiles of the d.q geometry
    if (c->multi()) {
        // Row halos: the fused kernel needs the neighbours' boundary rows of g (d's
        // halo rows are kept current locally); the plain kernel needs those of d.
        if (fuse) {
            // beta (from the all-gather that has just completed on s) gates every tile, so
            // there is nothing to overlap with: the exchange simply precedes the kernel.
            CGCHK(exchange_g_rows(c, s, c->g));
        } else if (sched == CG_SCHED_CLASSIC && c->nranks > 1) {
            // The reference's structure (utils.c:45-62): post the exchange, run the
            // stencil on the rows that need no halo, wait, then do the edge rows.  The
            // exchange goes to its own stream; events express the fork and the join.
            split = nrt >= 3;
            if (split) {
                HIPCHK(hipEventRecord(c->ev_fork, s));
                HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_fork, 0));
                CGCHK(comm_rc(c, c->comm->exchange_rows(c->comm_stream, c->d[c->dcur], c->rows, c->G,
                                                        c->halo_d[c->dcur][0], c->halo_d[c->dcur][1])));
                HIPCHK(hipEventRecord(c->ev_halo, c->comm_stream));
            } else {
                CGCHK(comm_rc(c, c->comm->exchange_rows(s, c->d[c->dcur], c->rows, c->G, c->halo_d[c->dcur][0],
                                                        c->halo_d[c->dcur][1])));
            }
        }
        // First iteration of a fused schedule: d's halo rows are current -- a closed form at set-up, exchanged
        // when a pending d update was applied -- unless a classic solve ran last, which leaves them one d update
        // behind: then they travel once more.
        if (first && sched != CG_SCHED_CLASSIC && c->nranks > 1 && c->d_halo_stale) {
            CGCHK(comm_rc(c, c->comm->exchange_rows(s, c->d[c->dcur], c->rows, c->G, c->halo_d[c->dcur][0], c->halo_d[c->dcur][1])));
            c->d_halo_stale = false;
        }
    }
    if (fuse) {
        const int in = c->dcur, out = c->dcur ^ 1;
        a.d_in = c->d[in]; a.g = c->g; a.d_out = c->d[out];
        a.d_top = c->has_top() ? c->halo_d[in][0] : nullptr;
        a.d_bot = c->has_bot() ? c->halo_d[in][1] : nullptr;
        a.g_top = c->has_top() ? c->halo_g[0] : nullptr;
        a.g_bot = c->has_bot() ? c->halo_g[1] : nullptr;
        a.d_top_out = c->has_top() ? c->halo_d[out][0] : nullptr;
        a.d_bot_out = c->has_bot() ? c->halo_d[out][1] : nullptr;
        if (sched == CG_SCHED_FUSED || products) launch_stencil<1, true>(tq, a, s);
        else launch_stencil<1, false>(tq, a, s);
        c->dcur = out;
    } else {
        a.d_in = c->d[c->dcur];
        a.d_top = c->has_top() ? c->halo_d[c->dcur][0] : nullptr;
        a.d_bot = c->has_bot() ? c->halo_d[c->dcur][1] : nullptr;
        if (sched == CG_SCHED_FUSED_NOQ && !products) {
            launch_stencil<0, false>(tq, a, s);
        } else if (!split) {
            launch_stencil<0, true>(tq, a, s);
        } else {
            // row tiles 1 .. nrt-2 touch no halo row; the first and the last one wait for the halos
            const long h = tq.tile_h;
            launch_stencil<0, true>(tq, a, s, h, (nrt - 1) * h, tq.nstrips);
            HIPCHK(hipStreamWaitEvent(s, c->ev_halo, 0));
            launch_stencil<0, true>(tq, a, s, 0, h, 0);
            launch_stencil<0, true>(tq, a, s, (nrt - 1) * h, c->rows, (nrt - 1) * tq.nstrips);
        }
    }
    HIPCHK(hipGetLastError());
    if (products) CGCHK(enqueue_dot_finish(c, c->q, nullptr, p_dq, tq.ntiles, 0, 0.0, a.sub ? 2 : 0));
    else CGCHK(enqueue_dot_finish(c, c->d[c->dcur], c->q, p_dq, tq.ntiles, 0, 0.0, a.sub ? 2 : 0));
    // The lean iterations that follow summarise their products under hypotheses taken from the iteration before;
    // this one hands them its approximate prefix sums (without them the first of them would summarise every chunk twice).
    const bool seed = lean && sub != nullptr && strict_speculative(c, t) && c->strict_blocks <= 1 &&
                      use_parallel_strict(c->strict_impl, c->count);
    if (seed) {
        cg::exact::spec_seed(s, c->xs, c->G, c->rows, 0);
        c->band_lo = c->rows > 64 ? c->rows / 2 - 32 : 0;
        c->band_hi = c->rows > 64 ? c->band_lo + 64 : c->rows;
        hipLaunchKernelGGL(cg::exact::k_band_sum, dim3(1), dim3(256), 0, s, (const double *)sub, c->G, c->rows, c->band_lo, c->band_hi,
                           c->strict_slot + 4, c->xs.skip);
        HIPCHK(hipGetLastError());
        c->band_left = c->live ? 0 : kBandIters;      // a solve that continues another has its scalars' history to predict from
    }

    // ---- u += tau d ; g += tau q (+ g.g)
    double *gg_sub = nullptr;
    if (sched == CG_SCHED_FUSED_NOQ) {
        StencilArgs b = base_args(c, t);
        b.d_in = c->d[c->dcur];
        b.d_top = c->has_top() ? c->halo_d[c->dcur][0] : nullptr;
        b.d_bot = c->has_bot() ? c->halo_d[c->dcur][1] : nullptr;
        b.u = defer_u ? nullptr : c->u; b.gw = c->g; b.partials = p_gg;
        b.sub = gg_sub = sub;
        launch_stencil<2, false>(t, b, s);
    } else {
        gg_sub = sub;
        if (t.vec == 2)
            hipLaunchKernelGGL(cg::k_update<2>, dim3(fb), dim3(kBlock), 0, s, c->u, c->g, c->d[c->dcur], c->q, c->count, c->sc, p_gg, gg_sub);
        else
            hipLaunchKernelGGL(cg::k_update<1>, dim3(fb), dim3(kBlock), 0, s, c->u, c->g, c->d[c->dcur], c->q, c->count, c->sc, p_gg, gg_sub);
    }
    HIPCHK(hipGetLastError());
    // The lean schedule's iterations expect g's halo rows in place when they begin (each one sends its
    // own new rows off before it ends): the iteration that leads into them does the same.
    if (lean && c->multi()) CGCHK(exchange_g_rows(c, s, c->g));
    CGCHK(enqueue_dot_finish(c, c->g, c->g, p_gg, npart_gg, defer_u ? 4 : 1, 0.0,
                             gg_sub == nullptr ? 0 : (sched == CG_SCHED_FUSED_NOQ ? 2 : 1)));   // tiled kernel: by rows; flat: linear
    if (lean) c->lean_pending = defer_u ? 1 : 0;
    if (seed) { cg::exact::spec_seed(s, c->xs, c->G, c->rows, 1); HIPCHK(hipGetLastError()); }

    // ---- d = beta d - g: now (classic) or inside the next stencil kernel (fused)
    if (sched == CG_SCHED_CLASSIC) {
        if (t.vec == 2)
            hipLaunchKernelGGL(cg::k_dupdate<2>, dim3(fb), dim3(kBlock), 0, s, c->d[c->dcur], c->g, c->count, c->sc, 1);
        else
            hipLaunchKernelGGL(cg::k_dupdate<1>, dim3(fb), dim3(kBlock), 0, s, c->d[c->dcur], c->g, c->count, c->sc, 1);
        HIPCHK(hipGetLastError());
    }
    return CG_OK;
}

// The d update a fused schedule still owes after its last iteration: applied
// once, so that d ends in the state the reference leaves it in (cg.c:81-87 runs
// before the loop exits).  beta is the one computed by the last k_finish.
int flush_pending_dupdate(cg_ctx *c)
{
    if (!c->pending_dupdate) return CG_OK;
    const TileGeom t = ctx_geom(c);
    const long nvec = c->count / t.vec;
    const int fb = flat_blocks(nvec, c->flat_blocks_opt);
    if (t.vec == 2)
        hipLaunchKernelGGL(cg::k_dupdate<2>, dim3(fb), dim3(kBlock), 0, c->stream, c->d[c->dcur], c->g, c->count, c->sc, 0);
    else
        hipLaunchKernelGGL(cg::k_dupdate<1>, dim3(fb), dim3(kBlock), 0, c->stream, c->d[c->dcur], c->g, c->count, c->sc, 0);
    HIPCHK(hipGetLastError());
    if (c->nranks > 1) {
        // keep the locally maintained halo rows of d consistent with the flushed state
        CGCHK(comm_rc(c, c->comm->exchange_rows(c->stream, c->d[c->dcur], c->rows, c->G, c->halo_d[c->dcur][0], c->halo_d[c->dcur][1])));
    }
    c->pending_dupdate = false;
    return CG_OK;
}

// Bring u and d to the state the reference's loop leaves them in after the iterations run so far (cg.c:63-69,
// :81-87): apply the u and d updates the last solve left owed.  With several ranks the d update ends with a
// halo exchange, so this is a collective step there: every rank reaches it through the same call.
int materialise(cg_ctx *c)
{
    if (!c->pending_dupdate && c->lean_pending == 0) return CG_OK;
    CGCHK(select_device(c->device));
    CGCHK(enqueue_uflush(c, c->lean_pending, c->dcur));
    c->lean_pending = 0;
    CGCHK(flush_pending_dupdate(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    return CG_OK;
}

// Lean schedule set-up: the second gradient buffer (with its guard rows, zero-filled) is
// allocated on first use.  Without enough free HBM an automatically chosen schedule falls
// back to FUSED_NOQ; an explicitly requested LEAN fails.
int lean_setup(cg_ctx *c)
{
    if (c->g_alloc[1]) return CG_OK;
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    hipError_t e = hipErrorOutOfMemory;
    if (c->padded_bytes + ((size_t)1 << 30) <= free_b) e = hipMalloc(&c->g_alloc[1], c->padded_bytes);
    if (e != hipSuccess) {
        c->g_alloc[1] = nullptr;
        (void)hipGetLastError();
        if (c->schedule < 0) { c->lean_unavailable = true; drop_graph(c); return CG_OK; }
        return fail(CG_ERR_HIP, "lean schedule: no room for a second gradient buffer (%zu bytes, %zu free)", c->padded_bytes, free_b);
    }
    HIPCHK(hipMemset(c->g_alloc[1], 0, c->padded_bytes));
    HIPCHK(hipStreamSynchronize(nullptr));               // the fill ran on the null stream
    c->g_buf[1] = c->g_alloc[1] + c->front_elems;
    return CG_OK;
}

// The captured graph of `len` steady-state iterations that starts from the context's present buffer
// roles, or nullptr.  Fused schedules swap d's buffers every iteration (lean: also g's, and the paired u
// update has period 2), so len is forced even and the roles are the same at every replay.
GraphSlot *find_graph(cg_ctx *c, int sched)
{
    for (GraphSlot &g : c->graphs)
        if (g.exec && g.sched == sched && g.reduce == c->reduce && g.dcur == c->dcur && g.gcur == c->gcur &&
            g.pend == c->lean_pending) {
            g.stamp = ++c->graph_clock;
            return &g;
        }
    return nullptr;
}

int build_graph(cg_ctx *c, long len, GraphSlot **out)
{
    *out = nullptr;
    const int sched = c->effective_schedule();
    if (sched != CG_SCHED_CLASSIC && (len & 1)) ++len;
    GraphSlot *slot = &c->graphs[0];
    for (GraphSlot &g : c->graphs) {
        if (!g.exec) { slot = &g; break; }
        if (g.stamp < slot->stamp) slot = &g;
    }
    if (slot->exec) (void)hipGraphExecDestroy(slot->exec);
    *slot = GraphSlot();
    hipGraph_t graph = nullptr;
    const int dcur0 = c->dcur, gcur0 = c->gcur, pend0 = c->lean_pending;
    const bool hyp0 = c->hyp_valid;
    const int band0 = c->band_left;
    c->band_left = 0;                      // (the step of an iteration behind a leading one is never part of a graph)
    HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = CG_OK;
    for (long i = 0; i < len && rc == CG_OK; ++i) rc = enqueue_iteration(c, false);
    hipError_t e = hipStreamEndCapture(c->stream, &graph);
    c->dcur = dcur0;                       // capture executed nothing
    c->set_gcur(gcur0);
    c->lean_pending = pend0;
    c->hyp_valid = hyp0;
    c->band_left = band0;
    if (rc != CG_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(CG_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    e = hipGraphInstantiate(&slot->exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) { slot->exec = nullptr; return fail(CG_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    slot->len = len;
    slot->sched = sched;
    slot->reduce = c->reduce;
    slot->dcur = dcur0;
    slot->gcur = gcur0;
    slot->pend = pend0;
    slot->stamp = ++c->graph_clock;
    *out = slot;
    return CG_OK;
}

} // namespace

// =========================================================================
// extern "C"
// =========================================================================

extern "C" {

const char *cg_last_error(void) { return g_err; }

int cg_device_count(int *count)
{
    if (!count) return fail(CG_ERR_ARG, "count is NULL");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *count = c;
    return CG_OK;
}

int cg_malloc(int device, long count, double **out)
{
    if (!out || count <= 0) return fail(CG_ERR_ARG, "cg_malloc: bad arguments");
    CGCHK(select_device(device));
    HIPCHK(hipMalloc(out, (size_t)count * sizeof(double)));
    HIPCHK(hipMemset(*out, 0, (size_t)count * sizeof(double)));
    // The fill runs on the null stream; the library's own streams are non-blocking and
    // would not wait for it, so finish it before the pointer is handed out.
    HIPCHK(hipStreamSynchronize(nullptr));
    return CG_OK;
}

int cg_free(int device, double *p)
{
    CGCHK(select_device(device));
    HIPCHK(hipFree(p));
    return CG_OK;
}

int cg_copy_to_host(int device, double *host_dst, const double *dev_src, long count)
{
    if (!host_dst || !dev_src || count < 0) return fail(CG_ERR_ARG, "cg_copy_to_host: bad arguments");
    CGCHK(select_device(device));
    HIPCHK(hipMemcpy(host_dst, dev_src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    return CG_OK;
}

int cg_copy_to_device(int device, double *dev_dst, const double *host_src, long count)
{
    if (!dev_dst || !host_src || count < 0) return fail(CG_ERR_ARG, "cg_copy_to_device: bad arguments");
    CGCHK(select_device(device));
    HIPCHK(hipMemcpy(dev_dst, host_src, (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    return CG_OK;
}

int cg_init_d(int device, long n, long row0, long rows, double *d)
{
    if (!d || n < 2 || rows < 1 || row0 < 0 || row0 + rows > n + 1)
        return fail(CG_ERR_ARG, "cg_init_d: need n >= 2 and 0 <= row0 < row0+rows <= n+1");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    const long G = n + 1;
    const double h = (double)1 / (double)n;             // utils.c:323
    hipLaunchKernelGGL(cg::k_init_d, dim3(flat_blocks(rows * G, 0)), dim3(kBlock), 0, sc->stream, d, G, row0, rows, h);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(sc->stream));
    return CG_OK;
}

int cg_init_g(int device, long count, const double *d, double *g)
{
    if (!d || !g || count < 1) return fail(CG_ERR_ARG, "cg_init_g: bad arguments");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    hipLaunchKernelGGL(cg::k_init_g, dim3(flat_blocks(count, 0)), dim3(kBlock), 0, sc->stream, g, d, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(sc->stream));
    return CG_OK;
}

int cg_stencil(int device, long G, long row0, long rows, const double *d, const double *top,
               const double *bot, double *q, double *dq_host)
{
    static const int poisson9[9] = {0, -1, 0, -1, 4, -1, 0, -1, 0};      // cg.c:33
    return cg_stencil9(device, poisson9, G, row0, rows, d, top, bot, q, dq_host);
}

int cg_stencil9(int device, const int *stencil9, long G, long row0, long rows, const double *d, const double *top,
                const double *bot, double *q, double *dq_host)
{
    if (!stencil9) return fail(CG_ERR_ARG, "cg_stencil9: coefficient array is NULL");
    if (!d || !q || G < 3 || rows < 1 || row0 < 0 || row0 + rows > G)
        return fail(CG_ERR_ARG, "cg_stencil: need G >= 3 and 0 <= row0 < row0+rows <= G");
    if ((row0 > 0 && !top) || (row0 + rows < G && !bot))
        return fail(CG_ERR_ARG, "cg_stencil: a block that does not touch the global boundary needs its halo row");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    const bool al = aligned16(d) && aligned16(q) && (!top || aligned16(top)) && (!bot || aligned16(bot));
    const TileGeom t = tile_geom(G, rows, al, 0);
    if (t.ntiles > kMaxPartials) return fail(CG_ERR_ARG, "cg_stencil: block too large for the stand-alone entry point");
    StencilArgs a;
    memset(&a, 0, sizeof a);
    a.d_in = d; a.d_top = row0 > 0 ? top : nullptr; a.d_bot = row0 + rows < G ? bot : nullptr;
    a.q = q; a.partials = sc->partials; a.sc = nullptr;
    a.coef = coef_from9(stencil9);
    a.G = G; a.rows = rows; a.row0 = row0; a.nstrips = t.nstrips; a.tile_h = t.tile_h; a.remap = 1;
    launch_stencil<0, true>(t, a, sc->stream);
    HIPCHK(hipGetLastError());
    if (dq_host) {
        hipLaunchKernelGGL(cg::k_finish, dim3(1), dim3(kBlock), 0, sc->stream, sc->partials, t.ntiles,
                           (const double *)nullptr, 1, sc->sc, 3, 0.0, (double *)nullptr, 0L);
        HIPCHK(hipGetLastError());
        Scalars h;
        HIPCHK(hipMemcpyAsync(&h, sc->sc, sizeof h, hipMemcpyDeviceToHost, sc->stream));
        HIPCHK(hipStreamSynchronize(sc->stream));
        *dq_host = h.dot;
    } else {
        HIPCHK(hipStreamSynchronize(sc->stream));
    }
    return CG_OK;
}

int cg_dot(int device, long count, const double *v, const double *w, int strict, double start,
           double *result_host)
{
    if (!v || !w || !result_host || count < 0) return fail(CG_ERR_ARG, "cg_dot: bad arguments");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    if (strict) {
        if (use_parallel_strict(strict, count)) {
            HIPCHK(sc->xs.reserve(count));
            cg::exact::enqueue(sc->stream, sc->xs, v, w, count, start, sc->slot);
        } else {
            hipLaunchKernelGGL(cg::k_dot_strict, dim3(1), dim3(64), 0, sc->stream, v, w, count, start, nullptr, sc->slot);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(result_host, sc->slot, sizeof(double), hipMemcpyDeviceToHost, sc->stream));
        HIPCHK(hipStreamSynchronize(sc->stream));
        if (getenv("CG_DEBUG_WALK") && sc->xs.stats && use_parallel_strict(strict, count)) {      // diagnostic: the walk's tally
            long st[8];
            HIPCHK(hipMemcpy(st, sc->xs.stats, sizeof st, hipMemcpyDeviceToHost));
            fprintf(stderr, "cg_dot walk: stretches walked %ld chunks %ld (planned %ld, brief failed %ld) rounds %ld blind %ld\n",
                    st[0], st[1], st[4], st[5], st[2], st[3]);
        }
        return CG_OK;
    }
    const int vec = (count % 2 == 0 && aligned16(v) && aligned16(w)) ? 2 : 1;
    const int fb = flat_blocks(count / vec, 0);
    if (vec == 2)
        hipLaunchKernelGGL(cg::k_dot_partial<2>, dim3(fb), dim3(kBlock), 0, sc->stream, v, w, count, sc->partials);
    else
        hipLaunchKernelGGL(cg::k_dot_partial<1>, dim3(fb), dim3(kBlock), 0, sc->stream, v, w, count, sc->partials);
    hipLaunchKernelGGL(cg::k_finish, dim3(1), dim3(kBlock), 0, sc->stream, sc->partials, (long)fb,
                       (const double *)nullptr, 1, sc->sc, 3, 0.0, (double *)nullptr, 0L);
    HIPCHK(hipGetLastError());
    Scalars h;
    HIPCHK(hipMemcpyAsync(&h, sc->sc, sizeof h, hipMemcpyDeviceToHost, sc->stream));
    HIPCHK(hipStreamSynchronize(sc->stream));
    *result_host = h.dot + start;
    return CG_OK;
}

int cg_axpy(int device, long count, double a, const double *x, double *y)
{
    if (!x || !y || count < 0) return fail(CG_ERR_ARG, "cg_axpy: bad arguments");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    const int vec = (count % 2 == 0 && aligned16(x) && aligned16(y)) ? 2 : 1;
    const int fb = flat_blocks(count / vec, 0);
    if (vec == 2) hipLaunchKernelGGL(cg::k_axpy<2>, dim3(fb), dim3(kBlock), 0, sc->stream, y, x, a, count);
    else hipLaunchKernelGGL(cg::k_axpy<1>, dim3(fb), dim3(kBlock), 0, sc->stream, y, x, a, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(sc->stream));
    return CG_OK;
}

int cg_xpay(int device, long count, double b, double *d, const double *g)
{
    if (!d || !g || count < 0) return fail(CG_ERR_ARG, "cg_xpay: bad arguments");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    const int vec = (count % 2 == 0 && aligned16(d) && aligned16(g)) ? 2 : 1;
    const int fb = flat_blocks(count / vec, 0);
    if (vec == 2) hipLaunchKernelGGL(cg::k_xpay<2>, dim3(fb), dim3(kBlock), 0, sc->stream, d, g, b, count);
    else hipLaunchKernelGGL(cg::k_xpay<1>, dim3(fb), dim3(kBlock), 0, sc->stream, d, g, b, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(sc->stream));
    return CG_OK;
}

int cg_init_d_block(int device, long n, long row0, long col0, long rows, long cols, double *d)
{
    if (!d || n < 2 || rows < 1 || cols < 1 || row0 < 0 || col0 < 0 || row0 + rows > n + 1 || col0 + cols > n + 1)
        return fail(CG_ERR_ARG, "cg_init_d_block: the block must lie inside the (n+1)^2 grid, n >= 2");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    const double h = (double)1 / (double)n;             // utils.c:323
    hipLaunchKernelGGL(cg::k_init_d_block, dim3(flat_blocks(rows * cols, 0)), dim3(kBlock), 0, sc->stream, d, n + 1, row0, col0,
                       rows, cols, h);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(sc->stream));
    return CG_OK;
}

int cg_stencil_block(int device, const int *stencil9, long rows, long cols, const double *d, const double *top,
                     const double *bot, const double *left, const double *right, double *q)
{
    return cg_stencil_block9(device, stencil9, rows, cols, d, top, bot, left, right, nullptr, q);
}

int cg_stencil_block9(int device, const int *stencil9, long rows, long cols, const double *d, const double *top,
                      const double *bot, const double *left, const double *right, const double *corners, double *q)
{
    if (!stencil9 || !d || !q || rows < 1 || cols < 1) return fail(CG_ERR_ARG, "cg_stencil_block: bad arguments");
    const bool nine = stencil9[0] != 0 || stencil9[2] != 0 || stencil9[6] != 0 || stencil9[8] != 0;
    const bool diagonal = (top && left) || (top && right) || (bot && left) || (bot && right);
    if (nine && diagonal && !corners)
        return fail(CG_ERR_ARG, "cg_stencil_block: corner coefficients on a chunk with diagonal neighbours need the four corner "
                    "values (cg_stencil_block9; the reference's corner_pad, utils.h:15)");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    cg::BlockArgs a;
    a.d = d; a.top = top; a.bot = bot; a.left = left; a.right = right; a.corner = corners; a.q = q; a.rows = rows; a.cols = cols;
    a.coef = coef_from9(stencil9);
    hipLaunchKernelGGL(cg::k_stencil_block, dim3(flat_blocks(rows * cols, 0)), dim3(kBlock), 0, sc->stream, a);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(sc->stream));
    return CG_OK;
}

int cg_pack_column(int device, long rows, long cols, const double *src, long col, double *dst)
{
    if (!src || !dst || rows < 1 || cols < 1 || col < 0 || col >= cols) return fail(CG_ERR_ARG, "cg_pack_column: bad arguments");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    hipLaunchKernelGGL(cg::k_pack_column, dim3(flat_blocks(rows, 0)), dim3(kBlock), 0, sc->stream, src, rows, cols, col, dst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(sc->stream));
    return CG_OK;
}

int cg_verify_u(int device, long n, long row0, long rows, const double *u, double sum_start, double *sum_u_host,
                double *max_err_host)
{
    if (!u || n < 2 || rows < 1 || row0 < 0 || row0 + rows > n + 1)
        return fail(CG_ERR_ARG, "cg_verify_u: need n >= 2 and 0 <= row0 < row0+rows <= n+1");
    DeviceScratch *sc;
    CGCHK(get_scratch(device, &sc));
    const long G = n + 1, count = rows * G;
    if (sum_u_host) {
        // the sequential sum of utils.c:372-379 with the products replaced by the elements themselves
        if (use_parallel_strict(kStrictAuto, count)) {
            HIPCHK(sc->xs.reserve(count));
            cg::exact::enqueue(sc->stream, sc->xs, u, nullptr, count, sum_start, sc->slot);
        } else {
            hipLaunchKernelGGL(cg::k_dot_strict, dim3(1), dim3(64), 0, sc->stream, u, (const double *)nullptr, count, sum_start,
                               (const double *)nullptr, sc->slot);
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(sum_u_host, sc->slot, sizeof(double), hipMemcpyDeviceToHost, sc->stream));
    }
    if (max_err_host) {
        const int fb = flat_blocks(count, 0);
        const double h = (double)1 / (double)n;         // utils.c:323
        hipLaunchKernelGGL(cg::k_max_err, dim3(fb), dim3(kBlock), 0, sc->stream, u, G, row0, rows, h, sc->partials);
        HIPCHK(hipGetLastError());
        std::vector<double> bm((size_t)fb);
        HIPCHK(hipMemcpyAsync(bm.data(), sc->partials, (size_t)fb * sizeof(double), hipMemcpyDeviceToHost, sc->stream));
        HIPCHK(hipStreamSynchronize(sc->stream));
        double m = 0.0;
        for (double v : bm) m = (v > m || v != v) ? v : m;
        *max_err_host = m;
    }
    HIPCHK(hipStreamSynchronize(sc->stream));
    return CG_OK;
}

// ---- the entry point that allocates nothing --------------------------------------------------------
// cg_stencil's kernels with their scratch (one partial sum per tile) in a file-scope device array instead
// of memory the library allocates: nothing here calls hipMalloc, hipFree, a stream or event constructor
// or any other allocating routine, and nothing synchronises, so the call may sit inside a stream capture.
// Each GPU holds its own copy of the array until the process ends; its address on a device is looked up
// once, with that device current, and only that address is ever handed to kernels.  The look-up itself
// (which may make the runtime load the code object) is kept out of captures: every other entry point that
// touches a device does it on the way (static_scratch_ptr from get_scratch / cg_create), and a first call on
// a device that arrives under capture is refused.
constexpr long kStaticPartials = 1L << 16;               // tiles of one call: 512 KiB of partial sums
static __device__ __align__(256) double scratch[kStaticPartials];
static double *g_static_ptr[kMaxDevices] = {nullptr};
static std::mutex g_static_mu;

// Address of `scratch` on `device` (which must be current), resolved on first use.
static int static_scratch_ptr(int device, double **out)
{
    if (device < 0 || device >= kMaxDevices) return fail(CG_ERR_ARG, "device index too large");
    std::lock_guard<std::mutex> lock(g_static_mu);
    if (!g_static_ptr[device]) {
        void *p = nullptr;
        HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_static_ptr[device] = static_cast<double *>(p);
    }
    if (out) *out = g_static_ptr[device];
    return CG_OK;
}

int cg_stencil_noalloc(int device, long G, long row0, long rows, const double *d, const double *top,
                       const double *bot, double *q, double *dq_dev, void *stream)
{
    static const int poisson9[9] = {0, -1, 0, -1, 4, -1, 0, -1, 0};      // cg.c:33
    if (!d || !q || G < 3 || rows < 1 || row0 < 0 || row0 + rows > G)
        return fail(CG_ERR_ARG, "cg_stencil_noalloc: need G >= 3 and 0 <= row0 < row0+rows <= G");
    if ((row0 > 0 && !top) || (row0 + rows < G && !bot))
        return fail(CG_ERR_ARG, "cg_stencil_noalloc: a block that does not touch the global boundary needs its halo row");
    if (device < 0 || device >= kMaxDevices) return fail(CG_ERR_ARG, "device index too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int caller_device = -1;
    if (hipGetDevice(&caller_device) != hipSuccess) caller_device = -1;
    if (caller_device != device) CGCHK(select_device(device));
    struct Restore {                                    // the caller's current device is the caller's business
        int dev, mine;
        ~Restore() { if (dev >= 0 && dev != mine) (void)hipSetDevice(dev); }
    } restore{caller_device, device};
    double *ptr = nullptr;
    {
        bool known;
        { std::lock_guard<std::mutex> lock(g_static_mu); known = g_static_ptr[device] != nullptr; }
        if (!known) {
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); st = hipStreamCaptureStatusNone; }
            if (st != hipStreamCaptureStatusNone)
                return fail(CG_ERR_STATE, "cg_stencil_noalloc: the first call on a device must be made outside stream capture "
                            "(it looks up the address of the static scratch array; any earlier library call on the device does too)");
        }
        CGCHK(static_scratch_ptr(device, &ptr));
    }
    const bool al = aligned16(d) && aligned16(q) && (!top || aligned16(top)) && (!bot || aligned16(bot));
    const TileGeom t = tile_geom(G, rows, al, 0);
    if (t.ntiles > kStaticPartials)
        return fail(CG_ERR_ARG, "cg_stencil_noalloc: %ld tiles (%d column strips x %ld row tiles of %d rows) exceed the %ld partial "
                    "sums the static scratch array holds; use cg_stencil", t.ntiles, t.nstrips, t.ntiles / t.nstrips, t.tile_h,
                    kStaticPartials);
    StencilArgs a;
    memset(&a, 0, sizeof a);
    a.d_in = d; a.d_top = row0 > 0 ? top : nullptr; a.d_bot = row0 + rows < G ? bot : nullptr;
    a.q = q; a.partials = ptr; a.sc = nullptr;
    a.coef = coef_from9(poisson9);
    a.G = G; a.rows = rows; a.row0 = row0; a.nstrips = t.nstrips; a.tile_h = t.tile_h; a.remap = 1;
    launch_stencil<0, true>(t, a, s);                   // every tile of this call writes its slot: nothing to clear
    HIPCHK(hipGetLastError());
    if (dq_dev) {
        hipLaunchKernelGGL(cg::k_sum_to, dim3(1), dim3(kBlock), 0, s, ptr, t.ntiles, dq_dev);
        HIPCHK(hipGetLastError());
    }
    return CG_OK;
}

} // extern "C"
namespace {
int resolve_static_scratch(int device) { return static_scratch_ptr(device, nullptr); }
}
extern "C" {

// ---------------------------------------------------------------- context ---

// Where the driver places a 2 GiB vector in HBM is not ours to choose, and it matters: the same four- and
// six-stream passes run 5 % (once in a while 30 %) slower on some sets of allocations than on others, in a way
// that is stable for a given set (measured at 16384^2: 1.49 / 2.38 ms on a good set, 1.56-1.59 / 2.41-2.53 ms on
// others; a fill of each single buffer takes the same time everywhere, it is how the streams of a pass meet in
// the memory channels).  Offsets inside one large allocation were never better than separate allocations.  So a
// context for a large grid draws up to eight sets of vectors, times the three lean passes on each (zero data:
// the kernels leave zeros behind) and keeps the fastest; the others are released afterwards, not in between,
// or the driver would hand the same memory out again.  Costs ~0.35 s once (and 12 GiB per set at 16384^2 while it
// lasts); CG_PLACEMENT_DRAWS=1 in the environment turns it off.
struct VectorSet {
    double *u = nullptr, *q = nullptr, *g[2] = {nullptr, nullptr}, *d[2] = {nullptr, nullptr};
    void release()
    {
        (void)hipFree(u); (void)hipFree(q); (void)hipFree(g[0]); (void)hipFree(g[1]); (void)hipFree(d[0]); (void)hipFree(d[1]);
        *this = VectorSet();
    }
};

static void adopt_vectors(cg_ctx *c, const VectorSet &v)
{
    const size_t front = c->front_elems;
    c->u_alloc = v.u; c->q_alloc = v.q; c->g_alloc[0] = v.g[0]; c->g_alloc[1] = v.g[1]; c->d_alloc[0] = v.d[0]; c->d_alloc[1] = v.d[1];
    c->u = c->u_alloc + front; c->q = c->q_alloc + front;
    c->g_buf[0] = c->g_alloc[0] + front;
    c->g_buf[1] = c->g_alloc[1] ? c->g_alloc[1] + front : nullptr;
    c->set_gcur(0);
    c->dcur = 0;
    for (int b = 0; b < 2; ++b) {
        c->d[b] = c->d_alloc[b] + front;
        c->halo_d[b][0] = c->d[b] - c->G; c->halo_d[b][1] = c->d[b] + c->count;
    }
}

static bool alloc_vectors(VectorSet &v, size_t bytes)
{
    double **all[6] = {&v.u, &v.q, &v.g[0], &v.d[0], &v.d[1], &v.g[1]};
    for (double **p : all)
        if (hipMalloc(p, bytes) != hipSuccess || hipMemset(*p, 0, bytes) != hipSuccess) {
            (void)hipGetLastError();
            v.release();
            return false;
        }
    return true;
}

static int choose_placement(cg_ctx *c)
{
    int draws = 8;                                      // a good set turns up in one draw out of four to nine, depending on the box
    if (const char *e = getenv("CG_PLACEMENT_DRAWS")) draws = atoi(e);
    long min_points = 1L << 27;
    if (const char *e = getenv("CG_PLACEMENT_MIN_POINTS")) min_points = atol(e);     // (tests run the search on small grids)
    if (draws < 1 || c->count < min_points || c->effective_schedule() != CG_SCHED_LEAN) return CG_OK;
    CGCHK(lean_setup(c));                               // the second gradient buffer belongs to the set
    if (c->effective_schedule() != CG_SCHED_LEAN) return CG_OK;
    size_t free_start = 0;
    auto room = [&]() {                                 // a further set, headroom for everything allocated later, and never
        size_t free_b = 0, total_b = 0;                 // more than half of what was free (other ranks may share the GPU)
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
        if (free_start == 0) free_start = free_b;
        return free_b >= 8 * c->padded_bytes && free_b - 6 * c->padded_bytes >= free_start / 2;
    };
    if (!room()) return CG_OK;
    // Two iterations of the lean schedule.  Every pass is timed in both directions (reading buffer pair 0 and writing
    // pair 1, and the other way round: the same kernel can differ by 10 % between them), the six-stream pass with one
    // and with two rows in flight (which is better depends on the set).  A solve runs the four-stream pass in one
    // direction and the six-stream pass in the other throughout, and cg_solve picks the cheaper assignment.
    // Where the grid allows it the passes are timed in the form the reference's summation order gives them (span sums
    // formed inside: more instructions, two small extra streams), the form the bit-exact solves run; the six-stream pass
    // of the plain form is timed as well, both directions in turn, to settle its rows in flight.
    struct Cost { double ms; int pf; double pass[3][2]; };
    const int reduce0 = c->reduce;
    auto cost = [&](Cost *out) -> int {
        double t[3][2], six[2] = {0.0, 0.0};
        c->time_warm = 1;
        c->reduce = CG_REDUCE_STRICT;                   // (cg_time_kernel falls back on the plain form by itself where the grid asks for it)
        int rc = CG_OK;
        for (int which = 0; which < 3 && rc == CG_OK; ++which)
            for (int dir = 0; dir < 2 && rc == CG_OK; ++dir) {
                c->pf_update_u = 2;
                c->time_dir = dir;
                rc = cg_time_kernel(c, which, 2, &t[which][dir], nullptr);
            }
        c->reduce = CG_REDUCE_FAST;
        c->time_dir = -1;
        for (int pf = 1; pf <= 2 && rc == CG_OK; ++pf) {
            c->pf_update_u = pf;
            rc = cg_time_kernel(c, 2, 2, &six[pf - 1], nullptr);
        }
        c->reduce = reduce0;
        c->time_warm = 2;
        if (rc != CG_OK) return rc;
        const double best6 = (t[1][0] + t[2][1] < t[1][1] + t[2][0]) ? t[1][0] + t[2][1] : t[1][1] + t[2][0];
        out->ms = t[0][0] + t[0][1] + best6;
        out->pf = six[0] < six[1] ? 1 : 2;
        memcpy(out->pass, t, sizeof t);
        return CG_OK;
    };
    // diagnostic: CG_PLACEMENT_KEEP=worst keeps the slowest set instead (to put a slow placement under the profiler)
    const char *keep = getenv("CG_PLACEMENT_KEEP");
    const bool keep_worst = keep && strcmp(keep, "worst") == 0;
    VectorSet best;
    best.u = c->u_alloc; best.q = c->q_alloc; best.g[0] = c->g_alloc[0]; best.g[1] = c->g_alloc[1];
    best.d[0] = c->d_alloc[0]; best.d[1] = c->d_alloc[1];
    Cost best_cost;
    int rc = cost(&best_cost);
    VectorSet rejected[8];
    int nrej = 0;
    for (int k = 1; rc == CG_OK && k < draws && k < 8; ++k) {
        if (!room()) break;
        VectorSet v;
        if (!alloc_vectors(v, c->padded_bytes)) break;
        adopt_vectors(c, v);
        Cost now;
        rc = cost(&now);
        if (getenv("CG_DEBUG_PLACEMENT")) fprintf(stderr, "placement draw %d: %.3f ms per two iterations (kept so far %.3f)\n", k, now.ms, best_cost.ms);
        if (rc == CG_OK && (keep_worst ? now.ms > best_cost.ms : now.ms < 0.995 * best_cost.ms)) { rejected[nrej++] = best; best = v; best_cost = now; }
        else rejected[nrej++] = v;
    }
    // Second stage: what makes a set slow is how the streams of a pass meet in the L2's sets, a matter of the vectors'
    // physical frames relative to one another -- so single vectors of the kept set are traded against the same vector
    // of every rejected set, one role after the other, and a trade stays where the passes get faster.
    int refine = 1;
    if (const char *e = getenv("CG_PLACEMENT_REFINE")) refine = atoi(e);
    auto role_of = [](VectorSet &v, int role) -> double *& {
        return role == 0 ? v.d[0] : role == 1 ? v.d[1] : role == 2 ? v.g[0] : role == 3 ? v.g[1] : v.u;
    };
    for (int round = 0; rc == CG_OK && !keep_worst && round < refine && nrej > 0; ++round) {
        bool traded = false;
        for (int role = 0; rc == CG_OK && role < 5; ++role)
            for (int k = 0; rc == CG_OK && k < nrej; ++k) {
                std::swap(role_of(best, role), role_of(rejected[k], role));
                adopt_vectors(c, best);
                Cost now;
                rc = cost(&now);
                if (getenv("CG_DEBUG_PLACEMENT")) fprintf(stderr, "placement trade: role %d with set %d: %.3f ms (kept so far %.3f)\n", role, k, now.ms, best_cost.ms);
                if (rc == CG_OK && now.ms < 0.995 * best_cost.ms) { best_cost = now; traded = true; }
                else std::swap(role_of(best, role), role_of(rejected[k], role));
            }
        if (!traded) break;
    }
    adopt_vectors(c, best);
    c->pf_update_u = (rc == CG_OK) ? best_cost.pf : 0;
    if (rc == CG_OK && !keep_worst) memcpy(c->pass_ms, best_cost.pass, sizeof c->pass_ms);
    if (getenv("CG_DEBUG_PLACEMENT") && rc == CG_OK)
        fprintf(stderr, "placement kept: d.q %.3f / %.3f  g' %.3f / %.3f  g'+u %.3f / %.3f ms (direction 0 / 1), %d row(s) in flight\n", best_cost.pass[0][0],
                best_cost.pass[0][1], best_cost.pass[1][0], best_cost.pass[1][1], best_cost.pass[2][0], best_cost.pass[2][1], best_cost.pf);
    for (int k = 0; k < nrej; ++k) rejected[k].release();
    // back to the state of a fresh context
    double *all[6] = {best.u, best.q, best.g[0], best.g[1], best.d[0], best.d[1]};
    for (double *p : all) HIPCHK(hipMemset(p, 0, c->padded_bytes));
    HIPCHK(hipMemset(c->sc, 0, sizeof(Scalars)));
    HIPCHK(hipDeviceSynchronize());
    c->needs_init = false; c->pending_dupdate = false; c->lean_pending = 0; c->live = false; c->q_valid = true;
    drop_graph(c);
    return CG_OK;                                       // a timing that could not run only ends the draws: the context is as good as without them
}

int cg_create(cg_ctx **out, int device, long n, long row0, long rows, int rank, int nranks)
{
    if (!out) return fail(CG_ERR_ARG, "cg_create: out is NULL");
    *out = nullptr;
    if (n < 2) return fail(CG_ERR_ARG, "cg_create: n must be >= 2 (got %ld)", n);
    const long G = n + 1;
    if (rows < 1 || row0 < 0 || row0 + rows > G)
        return fail(CG_ERR_ARG, "cg_create: rows [%ld,%ld) outside the %ld-row grid", row0, row0 + rows, G);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(CG_ERR_ARG, "cg_create: bad rank %d of %d", rank, nranks);
    if (nranks == 1 && (row0 != 0 || rows != G))
        return fail(CG_ERR_ARG, "cg_create: a single rank must own the whole grid");
    // Equal row blocks in rank order: the all-gathers of the dot products place rank r's partial sums at
    // r * (its own count) and RCCL needs the same count on every rank.
    if (nranks > 1 && (G % nranks != 0 || rows != G / nranks || row0 != (long)rank * rows))
        return fail(CG_ERR_ARG, "cg_create: rank %d of %d must own rows [%ld, %ld) of the %ld-row grid "
                    "(equal blocks in rank order; the grid side must be divisible by the rank count, cf. cg.c:23-26)",
                    rank, nranks, (long)rank * (G / nranks), (long)(rank + 1) * (G / nranks), G);
    CGCHK(select_device(device));
    (void)resolve_static_scratch(device);

    cg_ctx *c = new cg_ctx;
    c->device = device; c->n = n; c->G = G; c->row0 = row0; c->rows = rows; c->count = rows * G;
    c->rank = rank; c->nranks = nranks;
    const size_t bytes = (size_t)c->count * sizeof(double);
    // The most tiles any tile_h option can produce: one row per tile.
    c->npart = (long)((G + kBlock - 1) / kBlock) * rows;
    if (c->npart < kMaxFlatBlocks) c->npart = kMaxFlatBlocks;
    if (c->npart > (1L << 22)) c->npart = 1L << 22;
    int rc = CG_OK;
    auto alloc = [&](double **p, size_t b) {
        if (rc != CG_OK) return;
        hipError_t e = hipMalloc(p, b);
        if (e != hipSuccess) rc = fail(CG_ERR_HIP, "hipMalloc(%zu bytes): %s", b, hipGetErrorString(e));
    };
    // d (both buffers) and g carry their halo rows in the same allocation: row -1 directly
    // above row 0 and row `rows` directly below the last one, so that the kernels address
    // every row as base + r*G.  Where there is no neighbour the halo row stays zero.
    // Row 0 is placed a whole number of 2 MiB pages into the allocation (front >= one row), so that the
    // vectors keep the page-aligned placement a plain hipMalloc would give them.
    const long page = (2L << 20) / (long)sizeof(double);
    const long front = ((G + page - 1) / page) * page;
    const size_t padded = bytes + (size_t)(front + G) * sizeof(double);
    // (u and q get the same one-row offset: equal placement of all vectors relative to their
    // allocations measured ~10 % faster in the update kernel than mixed offsets)
    c->padded_bytes = padded;
    alloc(&c->u_alloc, padded); alloc(&c->q_alloc, padded);
    c->front_elems = (size_t)front;
    alloc(&c->g_alloc[0], padded); alloc(&c->d_alloc[0], padded); alloc(&c->d_alloc[1], padded);
    alloc(&c->partials, 3 * (size_t)c->npart * sizeof(double));     // d.q | g.g | q.q (strict order, prediction input)
    alloc(&c->strict_slot, 256);
    alloc(reinterpret_cast<double **>(&c->sc), sizeof(Scalars));
    if (rc == CG_OK) {
        c->u = c->u_alloc + front; c->q = c->q_alloc + front;
        c->g_buf[0] = c->g_alloc[0] + front;
        c->set_gcur(0);
        for (int b = 0; b < 2; ++b) {
            c->d[b] = c->d_alloc[b] + front;
            c->halo_d[b][0] = c->d[b] - G; c->halo_d[b][1] = c->d[b] + c->count;
        }
        if (hipMemset(c->g_alloc[0], 0, padded) != hipSuccess || hipMemset(c->d_alloc[0], 0, padded) != hipSuccess ||
            hipMemset(c->d_alloc[1], 0, padded) != hipSuccess)
            rc = fail(CG_ERR_HIP, "hipMemset of the padded vectors failed");
    }
    if (rc == CG_OK && (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
                        hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) != hipSuccess ||
                        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming) != hipSuccess))
        rc = fail(CG_ERR_HIP, "creating the context's streams and events failed");
    if (rc == CG_OK && hipHostMalloc(reinterpret_cast<void **>(&c->host_sc), sizeof(Scalars), hipHostMallocDefault) != hipSuccess)
        rc = fail(CG_ERR_HIP, "hipHostMalloc failed");
    if (rc != CG_OK) { cg_destroy(c); return rc; }
    (void)hipMemset(c->sc, 0, sizeof(Scalars));
    (void)hipStreamSynchronize(nullptr);
    c->xs.skip = &c->sc->done;                           // the strict-order kernels stand still behind a fired stop rule or a fault
    rc = choose_placement(c);
    if (rc != CG_OK) { cg_destroy(c); return rc; }
    *out = c;
    return CG_OK;
}

int cg_destroy(cg_ctx *c)
{
    if (!c) return CG_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graph(c);
    delete c->comm;
    c->xs.release();
    (void)hipFree(c->u_alloc); (void)hipFree(c->g_alloc[0]); (void)hipFree(c->g_alloc[1]); (void)hipFree(c->q_alloc);
    for (double *p : c->d_alloc) (void)hipFree(p);
    (void)hipFree(c->pack[0]); (void)hipFree(c->pack[1]); (void)hipFree(c->block_sums);
    (void)hipFree(c->partials); (void)hipFree(c->strict_slot); (void)hipFree(c->sc); (void)hipFree(c->hist);
    if (c->host_sc) (void)hipHostFree(c->host_sc);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return CG_OK;
}

int cg_set_option(cg_ctx *c, int key, long value)
{
    if (!c) return fail(CG_ERR_ARG, "cg_set_option: ctx is NULL");
    CGCHK(materialise(c));                              // the next solve may run another schedule: start it from plain vectors
    switch (key) {
    case CG_OPT_REDUCE:
        if (value != CG_REDUCE_FAST && value != CG_REDUCE_STRICT) return fail(CG_ERR_ARG, "unknown reduction order %ld", value);
        if (c->reduce != (int)value) c->live = false;   // q0 on the device was summed in the other order
        c->reduce = (int)value; break;
    case CG_OPT_SCHEDULE:
        if (value < -1 || value > CG_SCHED_LEAN) return fail(CG_ERR_ARG, "unknown schedule %ld", value);
        c->schedule = (int)value; break;
    case CG_OPT_TILE_H:
        if (value < 0) return fail(CG_ERR_ARG, "tile height must be >= 0");
        c->tile_h = value; break;
    case CG_OPT_GRAPH_ITERS:
        if (value < 0) return fail(CG_ERR_ARG, "graph length must be >= 0");
        c->graph_iters = value; c->graph_explicit = true; break;
    case CG_OPT_XCD_REMAP: c->remap = value ? 1 : 0; break;
    case CG_OPT_FLAT_BLOCKS:
        if (value < 0 || value > (1 << 20)) return fail(CG_ERR_ARG, "flat block count out of range");
        c->flat_blocks_opt = value; break;
    case CG_OPT_PREFETCH:
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(CG_ERR_ARG, "prefetch depth must be 0, 1, 2, 4 or 8");
        c->prefetch = (int)value; break;
    case CG_OPT_PREFETCH_READ:
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(CG_ERR_ARG, "prefetch depth must be 0, 1, 2, 4 or 8");
        c->prefetch_read = (int)value; break;
    case CG_OPT_TILE_WGS:
        if (value < 0) return fail(CG_ERR_ARG, "workgroup cap must be >= 0");
        c->tile_wgs = value; break;
    case CG_OPT_STRICT_BLOCKS:
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8)
            return fail(CG_ERR_ARG, "rank-grid side must be 0/1 (off), 2, 4 or 8");
        if (c->strict_blocks != value) c->live = false;
        c->strict_blocks = value; break;
    case CG_OPT_OVERLAP: c->overlap = value ? 1 : 0; break;
    case CG_OPT_RESUM_Q0: c->resum_q0 = value ? 1 : 0; break;
    case CG_OPT_STRICT_CHAIN: c->strict_chain = value ? 1 : 0; break;
    case CG_OPT_STRICT_FAULT_EVERY:
        if (value < 0) return fail(CG_ERR_ARG, "fault period must be >= 0");
        c->fault_every = value; break;
    case CG_OPT_STRICT_SPEC: c->strict_spec = value ? 1 : 0; break;
    case CG_OPT_STRICT_AUDIT: c->strict_audit = value ? 1 : 0; drop_graph(c); break;
    case CG_OPT_COMM_REHEARSAL:
        if (value && c->nranks != 1) return fail(CG_ERR_ARG, "the communication rehearsal is for single-rank contexts");
        if (value < 0 || value > 64) return fail(CG_ERR_ARG, "rehearsal rank count out of range");
        c->rehearse = (int)value; break;
    case CG_OPT_STRICT_IMPL:
        if (value < kStrictAuto || value > kStrictParallel) return fail(CG_ERR_ARG, "strict implementation must be 1, 2 or 3");
        c->strict_impl = (int)value; break;
    default: return fail(CG_ERR_ARG, "unknown option key %d", key);
    }
    drop_graph(c);
    return CG_OK;
}

int cg_set_stencil(cg_ctx *c, const int *stencil9)
{
    if (!c || !stencil9) return fail(CG_ERR_ARG, "cg_set_stencil: NULL argument");
    CGCHK(materialise(c));
    c->coef = coef_from9(stencil9);
    drop_graph(c);                                      // captured launches hold the old coefficients
    return CG_OK;
}

int cg_get_option(cg_ctx *c, int key, long *value)
{
    if (!c || !value) return fail(CG_ERR_ARG, "cg_get_option: NULL argument");
    switch (key) {
    case CG_OPT_REDUCE: *value = c->reduce; break;
    case CG_OPT_SCHEDULE: *value = c->effective_schedule(); break;      // what the next solve will run
    case CG_OPT_TILE_H: *value = ctx_geom(c).tile_h; break;
    case CG_OPT_GRAPH_ITERS: *value = c->graph_iters; break;
    case CG_OPT_XCD_REMAP: *value = c->remap; break;
    case CG_OPT_FLAT_BLOCKS: *value = c->flat_blocks_opt; break;
    case CG_OPT_PREFETCH: *value = c->prefetch; break;
    case CG_OPT_PREFETCH_READ: *value = ctx_geom(c).pf_mode[3]; break;
    case CG_OPT_STRICT_IMPL: *value = c->strict_impl; break;
    case CG_OPT_STRICT_BLOCKS: *value = c->strict_blocks; break;
    case CG_OPT_TILE_WGS: *value = ctx_geom(c).max_wgs; break;
    case CG_OPT_OVERLAP: *value = c->overlap; break;
    case CG_OPT_RESUM_Q0: *value = c->resum_q0; break;
    case CG_OPT_STRICT_CHAIN: *value = c->strict_chain; break;
    case CG_OPT_STRICT_FAULT_EVERY: *value = c->fault_every; break;
    case CG_OPT_STRICT_FAULTS: *value = c->strict_faults; break;
    case CG_OPT_STRICT_SPEC: *value = c->strict_spec; break;
    case CG_OPT_STRICT_AUDIT: {                         // failures the self-check has counted so far (0 when it is off)
        unsigned long long n = 0;
        if (c->xs.audit) {
            CGCHK(select_device(c->device));
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpy(&n, c->xs.audit, sizeof n, hipMemcpyDeviceToHost));
        }
        *value = (long)n;
        break;
    }
    case CG_OPT_STRICT_MISSES: {
        int n = 0;
        if (c->xs.nfix) {
            CGCHK(select_device(c->device));
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpy(&n, c->xs.nfix, sizeof n, hipMemcpyDeviceToHost));
        }
        *value = n;
        break;
    }
    case CG_OPT_STRICT_EVENTS: {
        long st[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (c->xs.stats) {
            CGCHK(select_device(c->device));
            HIPCHK(hipStreamSynchronize(c->stream));
            HIPCHK(hipMemcpy(st, c->xs.stats, sizeof st, hipMemcpyDeviceToHost));
        }
        *value = st[1] * 1000000L + st[9];              // chunks taken apart: d.q walk * 10^6 + g.g walk
        if (const char *dbg = getenv("CG_DEBUG_WALK"))   // diagnostic: the whole tally on stderr
            if (*dbg) fprintf(stderr, "walk d.q: groups %ld chunks %ld (planned %ld, brief failed %ld) rounds %ld blind %ld clocks %ld | "
                              "g.g: groups %ld chunks %ld (planned %ld, brief failed %ld) rounds %ld blind %ld clocks %ld\n",
                              st[0], st[1], st[4], st[5], st[2], st[3], st[6], st[8], st[9], st[12], st[13], st[10], st[11], st[14]);
        break;
    }
    case CG_OPT_COMM_REHEARSAL: *value = c->rehearse; break;
    default: return fail(CG_ERR_ARG, "unknown option key %d", key);
    }
    return CG_OK;
}

int cg_init_problem(cg_ctx *c)
{
    if (!c) return fail(CG_ERR_ARG, "cg_init_problem: ctx is NULL");
    CGCHK(select_device(c->device));
    hipStream_t s = c->stream;
    const double h = (double)1 / (double)c->n;           // utils.c:323
    const size_t bytes = (size_t)c->count * sizeof(double);
    c->dcur = 0;
    c->set_gcur(0);
    c->lean_pending = 0;
    c->pending_dupdate = false;
    c->live = false;
    c->needs_init = false;
    c->d_halo_stale = false;
    c->q_valid = true;
    c->xs.forget_hypotheses();
    c->hyp_valid = false;
    c->u_dir = -1;
    HIPCHK(hipMemsetAsync(c->u, 0, bytes, s));           // calloc, cg.c:44
    HIPCHK(hipMemsetAsync(c->q, 0, bytes, s));           // calloc, cg.c:45
    const int fb = flat_blocks(c->count, 0);
    hipLaunchKernelGGL(cg::k_init_d, dim3(fb), dim3(kBlock), 0, s, c->d[0], c->G, c->row0, c->rows, h);
    hipLaunchKernelGGL(cg::k_init_g, dim3(fb), dim3(kBlock), 0, s, c->g, c->d[0], c->count);
    // Halo rows of d0 are a closed form of the global row index: no exchange needed.
    const int hb = flat_blocks(c->G, 0);
    if (c->has_top()) hipLaunchKernelGGL(cg::k_init_d, dim3(hb), dim3(kBlock), 0, s, c->halo_d[0][0], c->G, c->row0 - 1, 1L, h);
    if (c->has_bot()) hipLaunchKernelGGL(cg::k_init_d, dim3(hb), dim3(kBlock), 0, s, c->halo_d[0][1], c->G, c->row0 + c->rows, 1L, h);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(c->sc, 0, sizeof(Scalars), s));
    HIPCHK(hipStreamSynchronize(s));
    return CG_OK;
}

int cg_sync_vectors(cg_ctx *c)
{
    if (!c) return fail(CG_ERR_ARG, "cg_sync_vectors: ctx is NULL");
    return materialise(c);
}

int cg_solve(cg_ctx *c, long max_iters, double rtol, cg_result *out)
{
    if (!c || !out) return fail(CG_ERR_ARG, "cg_solve: NULL argument");
    if (max_iters < 0 || rtol < 0) return fail(CG_ERR_ARG, "cg_solve: max_iters and rtol must be >= 0");
    if (c->needs_init) return fail(CG_ERR_STATE, "cg_time_kernel overwrote the vectors: call cg_init_problem before solving");
    CGCHK(select_device(c->device));
    if (c->nranks > 1 && !c->comm) return fail(CG_ERR_STATE, "multi-rank context without cg_comm_init");
    if (c->rehearse && (!c->comm || !c->comm->stream_ordered()))
        return fail(CG_ERR_STATE, "CG_OPT_COMM_REHEARSAL needs cg_comm_init (RCCL) on the single-rank context");
    if (c->rehearse) CGCHK(comm_rc(c, c->comm->prepare_rehearsal(c->G)));
    if (c->effective_schedule() == CG_SCHED_LEAN) CGCHK(lean_setup(c));   // may fall back to FUSED_NOQ (auto schedule only)
    const int sched = c->effective_schedule();
    hipStream_t s = c->stream;

    if (c->reduce == CG_REDUCE_STRICT && c->strict_blocks > 1) {
        const long sdim = c->strict_blocks;
        if (c->nranks != 1) return fail(CG_ERR_STATE, "the reference's multi-rank summation layout is emulated on one GPU only");
        if (c->G % sdim != 0)
            return fail(CG_ERR_ARG, "grid side %ld not divisible by %ld (the reference has the same rule, cg.c:23-26)", c->G, sdim);
        const long n2 = (c->G / sdim) * (c->G / sdim);
        if (c->pack_cap < n2) {
            drop_graph(c);
            (void)hipFree(c->pack[0]); (void)hipFree(c->pack[1]); (void)hipFree(c->block_sums);
            c->pack[0] = c->pack[1] = c->block_sums = nullptr; c->pack_cap = 0;
            HIPCHK(hipMalloc(&c->pack[0], (size_t)n2 * sizeof(double)));
            HIPCHK(hipMalloc(&c->pack[1], (size_t)n2 * sizeof(double)));
            HIPCHK(hipMalloc(&c->block_sums, 64 * sizeof(double)));
            c->pack_cap = n2;
        }
    }
    if (c->reduce == CG_REDUCE_STRICT && use_parallel_strict(c->strict_impl, c->count)) {
        const long before = c->xs.cap_chunks;
        HIPCHK(c->xs.reserve(c->count));
        if (c->xs.cap_chunks != before) drop_graph(c);      // captured nodes hold the old workspace
        if (c->strict_spec && c->effective_schedule() == CG_SCHED_LEAN && c->G % cg::exact::kChunk == 0 &&
            c->xs.cap_spec < cg::exact::Workspace::chunks_for(c->count)) {
            HIPCHK(c->xs.reserve_spec(c->count));
            drop_graph(c);
        }
    }
    // History buffer (set-up, outside the timed region): room for at least 65536 iterations, so that
    // solves of ordinary length never reallocate it (captured graphs hold its address); at most 2^24.
    long hist_want = max_iters > (1L << 16) ? max_iters : (1L << 16);
    if (hist_want > (1L << 24)) hist_want = 1L << 24;
    if (c->hist_cap < hist_want) {
        drop_graph(c);
        (void)hipFree(c->hist); c->hist = nullptr; c->hist_cap = 0;
        HIPCHK(hipMalloc(&c->hist, (size_t)hist_want * sizeof(double)));
        c->hist_cap = hist_want;
    }

    // A solve that follows another one on the same state continues it: q0 is the q1 already on the device,
    // and the d / u updates the last iteration left owed ride on this solve's first iteration, as between
    // any two iterations.  Otherwise q0 = g.g is summed and the first iteration starts from a materialised d.
    const bool cont = c->live;
    // Lean schedule: a solve that continues another one inherits which of the two kinds of pass 2 reads which buffer
    // pair.  Where the context measured the passes per direction (one rank, large grid) and the other assignment would
    // pay for applying the owed updates now and starting over with a leading iteration, do that: the bits are the same,
    // the iterations 4 % faster where the directions differ as they do on some placements.
    if (sched == CG_SCHED_LEAN && c->pending_dupdate && c->nranks == 1 && c->pass_ms[1][0] > 0.0 && !getenv("CG_LEAD_DEFER_U")) {
        const int next_u = (c->lean_pending == 1) ? c->dcur : c->dcur ^ 1;      // the pair the next pass with the u update reads
        const double keep = c->pass_ms[2][next_u] + c->pass_ms[1][next_u ^ 1];
        const double other = c->pass_ms[2][next_u ^ 1] + c->pass_ms[1][next_u];
        const double per_byte = 0.5 * (c->pass_ms[1][0] + c->pass_ms[1][1]) / 32.0;
        const double restart = (24.0 + (c->lean_pending ? 24.0 : 0.0) + 8.0) * per_byte;   // the owed updates on their own, a flush at the end
        if (0.5 * (double)max_iters * (keep - other) > 1.25 * restart) {
            if (getenv("CG_DEBUG_PLACEMENT"))
                fprintf(stderr, "pairing: %ld iterations would keep %.3f ms per pair where %.3f is to be had: starting over\n", max_iters, keep, other);
            CGCHK(materialise(c));
        }
    }
    const bool first_form = (sched == CG_SCHED_CLASSIC) ? false : !c->pending_dupdate;   // iteration 0 has no d update to fold in
    const long lead = (first_form && max_iters > 0) ? 1 : 0;                             // iterations enqueued before the graph
    // Lean schedule: every second iteration applies two u updates at once.  With an odd number of iterations behind
    // the leading one, the leading one leaves its update to the first of them, so that the last iteration completes
    // a pair and the solve ends with no update of u owed (a stop rule that fires early may still leave one).
    c->lead_defers_u = false;
    if (lead && sched == CG_SCHED_LEAN) {
        // Cost of the n iterations behind the leading one either way, in units that only need to be consistent: the
        // passes' measured times per direction where the context measured them (one rank, large grid: the four- and the
        // six-stream pass keep one direction each for the whole solve), else their bytes per point.
        const long n = max_iters - lead;
        const bool measured = c->nranks == 1 && c->pass_ms[1][0] > 0.0;
        const int d0 = c->dcur;                     // the direction of the first of them
        const double m4[2] = {measured ? c->pass_ms[1][d0] : 32.0, measured ? c->pass_ms[1][d0 ^ 1] : 32.0};
        const double m5[2] = {measured ? c->pass_ms[2][d0] : 48.0, measured ? c->pass_ms[2][d0 ^ 1] : 48.0};
        const double per_byte = 0.5 * (m4[0] + m4[1]) / 32.0;
        const double flush = 24.0 * per_byte, lead_u = 16.0 * per_byte;   // a u update on its own; u's share of the leading iteration
        const double deferred = (double)((n + 1) / 2) * m5[0] + (double)(n / 2) * m4[1] + ((n > 0 && !(n & 1)) ? flush : 0.0) - lead_u;
        const double at_once = (double)((n + 1) / 2) * m4[0] + (double)(n / 2) * m5[1] + ((n & 1) ? flush : 0.0);
        c->lead_defers_u = n > 0 && deferred < at_once;
        if (const char *e = getenv("CG_LEAD_DEFER_U")) c->lead_defers_u = n > 0 && atoi(e) != 0;      // diagnostic override
        if (getenv("CG_DEBUG_PLACEMENT"))
            fprintf(stderr, "pairing: %ld iterations behind the leading one from direction %d: deferred %.3f, at once %.3f -> %s\n", n, d0,
                    deferred, at_once, c->lead_defers_u ? "deferred" : "at once");
    }
    struct PendGuard {                      // the graph is chosen for the state the leading iteration will leave
        cg_ctx *c; int saved;
        ~PendGuard() { c->lean_pending = saved; }
    };

    // Graph of the steady-state iteration body (set-up, outside the timed region), for the buffer roles
    // the context will be in after the leading iteration (which works in place and leaves no u update owed).
    long glen = c->graph_iters;
    // The host-staged transport synchronises the stream and cannot be captured.  RCCL calls can: every rank
    // captures, then all ranks agree (one all-gather of a flag) that every capture succeeded, else all run eagerly.
    if (c->multi() && !c->comm->stream_ordered()) glen = 0;
    GraphSlot *graph = nullptr;
    if (glen > 0 && max_iters - lead >= glen) {
        PendGuard guard{c, c->lean_pending};
        if (lead && sched == CG_SCHED_LEAN) c->lean_pending = c->lead_defers_u ? 1 : 0;
        graph = find_graph(c, sched);
        bool agree = c->multi() && graph == nullptr;
        if (!graph) {
            const int rc = build_graph(c, glen, &graph);
            if (rc != CG_OK && !c->multi()) return rc;
            if (rc != CG_OK) graph = nullptr;
        }
        if (agree && c->nranks > 1 && !c->comm->self_only()) {
            const double mine = graph ? 1.0 : 0.0;
            std::vector<double> all((size_t)c->nranks, 0.0);
            HIPCHK(hipMemcpyAsync(c->comm->gather_buf() + c->rank, &mine, sizeof mine, hipMemcpyHostToDevice, s));
            HIPCHK(hipStreamSynchronize(s));
            CGCHK(comm_rc(c, c->comm->allgather_scalar(s)));
            HIPCHK(hipMemcpyAsync(all.data(), c->comm->gather_buf(), all.size() * sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            bool ok = true;
            for (double v : all) ok = ok && v == 1.0;
            if (!ok) { drop_graph(c); graph = nullptr; c->graph_iters = 0; }      // every rank takes the same decision
        }
    }
    glen = graph ? graph->len : 0;
    HIPCHK(hipStreamSynchronize(s));

    const auto t0 = std::chrono::steady_clock::now();
    if (cont && !c->resum_q0) {
        hipLaunchKernelGGL(cg::k_finish, dim3(1), dim3(kBlock), 0, s, (const double *)nullptr, 0L, (const double *)nullptr, 1,
                           c->sc, 5, rtol, c->hist, c->hist_cap);
        HIPCHK(hipGetLastError());
    } else {
        // q0 = g.g (cg.c:55); a continuing solve keeps its iteration count and what it owes (k_finish, what = 6)
        const TileGeom t = ctx_geom(c);
        const int fb = flat_blocks(c->count / t.vec, c->flat_blocks_opt);
        double *p0;
        CGCHK(rank_partials(c, c->partials, fb, &p0));
        if (c->reduce != CG_REDUCE_STRICT) {
            if (t.vec == 2) hipLaunchKernelGGL(cg::k_dot_partial<2>, dim3(fb), dim3(kBlock), 0, s, c->g, c->g, c->count, p0);
            else hipLaunchKernelGGL(cg::k_dot_partial<1>, dim3(fb), dim3(kBlock), 0, s, c->g, c->g, c->count, p0);
            HIPCHK(hipGetLastError());
        }
        CGCHK(enqueue_dot_finish(c, c->g, c->g, p0, fb, cont ? 6 : 2, rtol));
    }
    const int dcur_start = c->dcur, gcur_start = c->gcur;
    // Buffer roles after `executed` iterations of this solve (those skipped behind a fired stop rule or a fault moved
    // nothing): the leading iteration works in place, every other fused / lean iteration flips d (lean: and g).
    auto roles_after = [&](long executed, const Scalars &h) {
        if (sched != CG_SCHED_CLASSIC) {
            const long moved = executed - (lead && executed >= 1 ? 1 : 0);
            c->dcur = dcur_start ^ (int)(moved & 1);
            if (sched == CG_SCHED_LEAN) {
                c->set_gcur(gcur_start ^ (int)(moved & 1));
                c->lean_pending = (int)(h.iter - h.u_applied);
                c->q_valid = false;
            } else {
                c->set_gcur(gcur_start);
                c->lean_pending = 0;
                c->q_valid = (sched == CG_SCHED_FUSED);
            }
            if (executed >= 1) c->pending_dupdate = true;
        } else {
            c->dcur = dcur_start;
            c->set_gcur(gcur_start);
            if (c->nranks > 1 && executed >= 1) c->d_halo_stale = true;
        }
    };
    long launched = 0;
    bool stopped = false;
    if (sched != CG_SCHED_LEAN || c->reduce != CG_REDUCE_STRICT) c->band_left = 0;
    if (lead) { CGCHK(enqueue_iteration(c, true)); launched = 1; }
    for (;;) {
        while (launched < max_iters && !stopped) {
            if (glen > 0 && max_iters - launched >= glen && graph->dcur == c->dcur && graph->gcur == c->gcur &&
                graph->pend == c->lean_pending && c->band_left == 0) {     // (the first iterations behind a leading one have a step of their own: eager)
                HIPCHK(hipGraphLaunch(graph->exec, s));
                launched += glen;                           // an even number of iterations: the roles are as before
            } else {
                CGCHK(enqueue_iteration(c, false));
                launched += 1;
            }
            if (rtol > 0 && (launched % 64 < (glen > 0 ? glen : 1))) {
                // Stop rule: look at the device flag now and then.  Iterations launched
                // after it fired are no-ops on the device, so polling lazily costs
                // nothing but empty launches.
                HIPCHK(hipMemcpyAsync(c->host_sc, c->sc, sizeof(Scalars), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                stopped = c->host_sc->done != 0;
            }
        }
        HIPCHK(hipMemcpyAsync(c->host_sc, c->sc, sizeof(Scalars), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (c->host_sc->done != 2) break;
        // A published walk gave up (several ranks, strict order): the iteration it belongs to is completed with the ranks
        // walking in turn, and the solve goes on behind it.
        const Scalars h = *c->host_sc;
        const long executed = (long)(h.iter - h.iter_base);
        if (sched != CG_SCHED_LEAN || h.fault < 1 || h.fault > 2) return fail(CG_ERR_STATE, "unexpected fault state %d on the device", h.fault);
        roles_after(executed, h);
        CGCHK(recover_faulted_iteration(c, h.fault - 1));
        launched = executed + 1;
        stopped = false;
    }
    const auto t1 = std::chrono::steady_clock::now();

    const Scalars &h = *c->host_sc;
    const long executed = (long)(h.iter - h.iter_base);   // iterations of this solve that really ran
    roles_after(executed, h);
    if (c->lean_pending < 0 || c->lean_pending > 1) return fail(CG_ERR_STATE, "lean schedule: %d u updates owed", c->lean_pending);
    c->live = true;
    c->hist_len = executed < c->hist_cap ? executed : c->hist_cap;
    out->iters = executed;
    out->q1 = h.q1;
    out->q_init = h.q_init;
    out->seconds = std::chrono::duration<double>(t1 - t0).count();
    out->converged = h.done;
    return CG_OK;
}

int cg_device_ptr(cg_ctx *c, int which, double **out)
{
    if (!c || !out) return fail(CG_ERR_ARG, "cg_device_ptr: NULL argument");
    CGCHK(materialise(c));
    switch (which) {
    case CG_VEC_U: *out = c->u; break;
    case CG_VEC_G: *out = c->g; break;
    case CG_VEC_D: *out = c->d[c->dcur]; break;
    case CG_VEC_Q: *out = c->q; break;
    default: return fail(CG_ERR_ARG, "unknown vector id %d", which);
    }
    return CG_OK;
}

int cg_copy_out(cg_ctx *c, int which, double *host_dst, long count)
{
    if (!c || !host_dst) return fail(CG_ERR_ARG, "cg_copy_out: NULL argument");
    if (count != c->count) return fail(CG_ERR_ARG, "cg_copy_out: count %ld != block size %ld", count, c->count);
    if (c->needs_init) return fail(CG_ERR_STATE, "cg_time_kernel overwrote the vectors: call cg_init_problem first");
    CGCHK(select_device(c->device));
    if (which == CG_VEC_Q && !c->q_valid)
        return fail(CG_ERR_STATE, "q is not stored by the LEAN and FUSED_NOQ schedules; use CLASSIC or FUSED to read it");
    double *src;
    CGCHK(cg_device_ptr(c, which, &src));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(host_dst, src, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    return CG_OK;
}

int cg_get_history(cg_ctx *c, double *host_dst, long count, long *written)
{
    if (!c || !host_dst || count < 0) return fail(CG_ERR_ARG, "cg_get_history: bad arguments");
    CGCHK(select_device(c->device));
    const long m = count < c->hist_len ? count : c->hist_len;
    if (m > 0) HIPCHK(hipMemcpy(host_dst, c->hist, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
    if (written) *written = m;
    return CG_OK;
}

int cg_time_kernel(cg_ctx *c, int which, int reps, double *ms_per_launch, double *bytes_per_launch)
{
    if (!c || !ms_per_launch || reps < 1) return fail(CG_ERR_ARG, "cg_time_kernel: bad arguments");
    CGCHK(select_device(c->device));                    // local kernels only: no communication, any rank may call it
    const int sched = c->effective_schedule();
    const TileGeom t = ctx_geom(c), tq = dq_geom(c, t);
    hipStream_t s = c->stream;
    const long nvec = c->count / t.vec;
    const int fb = flat_blocks(nvec, c->flat_blocks_opt);
    // Give the prologue scalars harmless values: tau = 0 and beta = 1 keep the data bounded.
    if (sched == CG_SCHED_LEAN) CGCHK(lean_setup(c));
    if (c->effective_schedule() != sched) return fail(CG_ERR_HIP, "lean schedule: not enough free HBM");
    if (t.ntiles > c->npart || tq.ntiles > c->npart)
        return fail(CG_ERR_ARG, "%ld tiles exceed the %ld partial sums the context holds: use a larger CG_OPT_TILE_H",
                    t.ntiles > tq.ntiles ? t.ntiles : tq.ntiles, c->npart);
    // Lean schedule: the four- and the six-stream pass alternate, so each reads the same buffer pair in every one of its
    // launches of a solve; the timing keeps the pair the solve that ran last would have given the pass next (the read-only
    // pass runs in both directions in turn, as it does in a solve).
    int own_dir = -1;
    if (sched == CG_SCHED_LEAN && c->live) c->u_dir = (c->lean_pending == 1) ? c->dcur : c->dcur ^ 1;   // the pair the next pass with the u update reads
    if (sched == CG_SCHED_LEAN && c->time_dir < 0 && c->u_dir >= 0 && (which == 1 || which == 2))
        own_dir = (which == 2) ? c->u_dir : c->u_dir ^ 1;
    // From here on the vectors and the device scalars are scratch: nothing owed, nothing to continue, and
    // no solve or read-back until cg_init_problem has set the problem up again.
    c->pending_dupdate = false; c->lean_pending = 0; c->live = false; c->needs_init = true; c->hyp_valid = false;
    bool spec = false;
    if (sched == CG_SCHED_LEAN && c->reduce == CG_REDUCE_STRICT && c->strict_spec &&
        c->G % cg::exact::kChunk == 0 && use_parallel_strict(c->strict_impl, c->count)) {
        HIPCHK(c->xs.reserve(c->count));
        HIPCHK(c->xs.reserve_spec(c->count));
        drop_graph(c);
        spec = strict_speculative(c, t);
    }
    Scalars hs;
    memset(&hs, 0, sizeof hs);
    hs.q0 = 1.0; hs.tau = 0.0; hs.beta = 1.0; hs.rho[0] = 1.0; hs.rho[1] = 1.0;
    hs.iter = 1LL << 40;                                // every deferred u entry counts as "ran"
    HIPCHK(hipMemcpy(c->sc, &hs, sizeof hs, hipMemcpyHostToDevice));
    double per_point = 0.0;
    auto launch = [&]() {
        if (sched == CG_SCHED_LEAN) {
            const int fixed_dir = c->time_dir >= 0 ? c->time_dir : own_dir;
            if (fixed_dir >= 0) { c->dcur = fixed_dir; c->set_gcur(fixed_dir); }
            StencilArgs a = base_args(c, which == 0 ? tq : t);
            const int out = c->dcur ^ 1;
            a.partials = c->partials; a.d_in = c->d[c->dcur]; a.g = c->g;
            if (spec) {                                    // strict order: the passes with their span summaries
                a.sub = c->xs.sub; a.rec = c->xs.rec;
                a.hyp = c->xs.hyp[which == 0 ? 0 : 1]; a.hyp_which = which == 0 ? 0 : 1;
                a.partials2 = c->partials + 2 * c->npart;
            }
            if (which == 0) {
                if (spec) launch_stencil<3, false, 2>(tq, a, s);
                else launch_stencil<3, false>(tq, a, s);
            } else {
                a.partials = c->partials + c->npart; a.d_out = c->d[out]; a.g_out = c->g_buf[c->gcur ^ 1]; a.u = c->u;
                if (which == 1) { if (spec) launch_stencil<4, false, 2>(t, a, s); else launch_stencil<4, false>(t, a, s); }
                else { if (spec) launch_stencil<5, false, 2>(t, a, s); else launch_stencil<5, false>(t, a, s); }
                c->dcur = out; c->set_gcur(c->gcur ^ 1);
            }
        } else if (which == 0) {
            StencilArgs a = base_args(c, tq);
            a.partials = c->partials; a.q = c->q;
            if (sched == CG_SCHED_CLASSIC) {
                a.d_in = c->d[c->dcur];
                launch_stencil<0, true>(tq, a, s);
            } else {
                a.d_in = c->d[c->dcur]; a.g = c->g; a.d_out = c->d[c->dcur ^ 1];
                if (sched == CG_SCHED_FUSED) launch_stencil<1, true>(tq, a, s);
                else launch_stencil<1, false>(tq, a, s);
                c->dcur ^= 1;
            }
        } else if (which == 1) {
            if (sched == CG_SCHED_FUSED_NOQ) {
                StencilArgs b = base_args(c, t);
                b.d_in = c->d[c->dcur]; b.u = c->u; b.gw = c->g; b.partials = c->partials + c->npart;
                launch_stencil<2, false>(t, b, s);
            } else if (t.vec == 2) {
                hipLaunchKernelGGL(cg::k_update<2>, dim3(fb), dim3(kBlock), 0, s, c->u, c->g, c->d[c->dcur], c->q, c->count, c->sc, c->partials + c->npart, (double *)nullptr);
            } else {
                hipLaunchKernelGGL(cg::k_update<1>, dim3(fb), dim3(kBlock), 0, s, c->u, c->g, c->d[c->dcur], c->q, c->count, c->sc, c->partials + c->npart, (double *)nullptr);
            }
        } else {
            if (t.vec == 2) hipLaunchKernelGGL(cg::k_dupdate<2>, dim3(fb), dim3(kBlock), 0, s, c->d[c->dcur], c->g, c->count, c->sc, 0);
            else hipLaunchKernelGGL(cg::k_dupdate<1>, dim3(fb), dim3(kBlock), 0, s, c->d[c->dcur], c->g, c->count, c->sc, 0);
        }
    };
    if (sched == CG_SCHED_LEAN) {
        if (which == 0) per_point = 16.0;                               // R d, g
        else if (which == 1) per_point = 32.0;                          // R d, g  W d', g'
        else if (which == 2) per_point = 48.0;                          // ... and R u  W u (every second iteration)
        else return fail(CG_ERR_ARG, "unknown kernel id %d", which);
    } else if (which == 0) per_point = sched == CG_SCHED_CLASSIC ? 16.0 : (sched == CG_SCHED_FUSED ? 32.0 : 24.0);
    else if (which == 1) per_point = sched == CG_SCHED_FUSED_NOQ ? 40.0 : 48.0;
    else if (which == 2) {
        if (sched != CG_SCHED_CLASSIC) return fail(CG_ERR_STATE, "the fused schedules have no separate d-update kernel");
        per_point = 24.0;
    } else return fail(CG_ERR_ARG, "unknown kernel id %d", which);

    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < c->time_warm; ++i) launch();      // warm-up
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) launch();
    HIPCHK(hipEventRecord(e1, s));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipGetLastError());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *ms_per_launch = (double)ms / reps;
    if (bytes_per_launch) *bytes_per_launch = per_point * (double)c->count;
    return CG_OK;
}

// ------------------------------------------------------------- multi-GPU ---

int cg_comm_unique_id(void *id128)
{
    if (!id128) return fail(CG_ERR_ARG, "cg_comm_unique_id: NULL");
    char msg[256];
    if (cg::RcclComm::unique_id(id128, msg, sizeof msg) != 0) return fail(CG_ERR_COMM, "%s", msg);
    return CG_OK;
}

int cg_comm_init(cg_ctx *c, const void *id128)
{
    if (!c || !id128) return fail(CG_ERR_ARG, "cg_comm_init: NULL argument");
    CGCHK(select_device(c->device));                    // one rank is allowed: the solve ignores the communicator,
                                                        // cg_comm_selftest can still exercise it
    delete c->comm;
    c->comm = nullptr;
    char msg[256];
    cg::Comm *comm = cg::RcclComm::create(id128, c->rank, c->nranks, c->has_top(), c->has_bot(), msg, sizeof msg);
    if (!comm) return fail(CG_ERR_COMM, "%s", msg);
    c->comm = comm;
    drop_graph(c);
    return CG_OK;
}

int cg_comm_init_self(cg_ctx *c)
{
    if (!c) return fail(CG_ERR_ARG, "cg_comm_init_self: NULL context");
    CGCHK(select_device(c->device));
    delete c->comm;
    c->comm = nullptr;
    char msg[256], id[128];
    if (cg::RcclComm::unique_id(id, msg, sizeof msg) != 0) return fail(CG_ERR_COMM, "%s", msg);
    cg::Comm *comm = cg::RcclComm::create(id, c->rank, c->nranks, c->has_top(), c->has_bot(), msg, sizeof msg, true);
    if (!comm) return fail(CG_ERR_COMM, "%s", msg);
    c->comm = comm;
    drop_graph(c);
    return CG_OK;
}

int cg_comm_init_host(cg_ctx *c, cg_exchange_fn exchange, cg_allgather_fn allgather, void *user)
{
    if (!c || !exchange || !allgather) return fail(CG_ERR_ARG, "cg_comm_init_host: NULL argument");
    if (c->nranks < 2) return fail(CG_ERR_STATE, "cg_comm_init_host on a single-rank context");
    CGCHK(select_device(c->device));
    delete c->comm;
    c->comm = nullptr;
    char msg[256];
    cg::Comm *comm = cg::HostComm::create(exchange, allgather, user, c->rank, c->nranks, c->G, c->has_top(), c->has_bot(),
                                          msg, sizeof msg);
    if (!comm) return fail(CG_ERR_COMM, "%s", msg);
    c->comm = comm;
    drop_graph(c);
    return CG_OK;
}

// Transport check, collective over the ranks.  (1) every rank contributes 1000*(r+1)+0.5 to the
// all-gather and must read back every other rank's value; (2) a row exchange on scratch rows
// tagged with (rank, which end) must deliver the neighbours' tags; (3) one row is sent to this
// rank itself.  Uses the second direction buffer as scratch: call it before cg_init_problem.
int cg_comm_selftest(cg_ctx *c)
{
    if (!c) return fail(CG_ERR_ARG, "cg_comm_selftest: NULL context");
    if (!c->comm) return fail(CG_ERR_STATE, "cg_comm_selftest: no communicator (cg_comm_init / cg_comm_init_host first)");
    CGCHK(select_device(c->device));
    hipStream_t s = c->stream;
    const int P = c->nranks, r = c->rank;
    const long G = c->G;
    std::vector<double> vals((size_t)P, -1.0);
    const double mine = 1000.0 * (r + 1) + 0.5;
    HIPCHK(hipMemsetAsync(c->comm->gather_buf(), 0, (size_t)P * sizeof(double), s));
    HIPCHK(hipMemcpyAsync(c->comm->gather_buf() + r, &mine, sizeof mine, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    CGCHK(comm_rc(c, c->comm->allgather_scalar(s)));
    HIPCHK(hipMemcpyAsync(vals.data(), c->comm->gather_buf(), (size_t)P * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int k = 0; k < P; ++k)
        if (vals[(size_t)k] != 1000.0 * (k + 1) + 0.5)
            return fail(CG_ERR_COMM, "all-gather self-test: slot %d holds %g on rank %d", k, vals[(size_t)k], r);

    // rows: first row tagged 2r, last row tagged 2r+1 (one and the same row if the block has a single row)
    double *blk = c->d[1];
    std::vector<double> row((size_t)G);
    auto fill = [&](double *dst, double tag) -> hipError_t {
        for (long j = 0; j < G; ++j) row[(size_t)j] = tag + 1e-6 * (double)j;
        return hipMemcpy(dst, row.data(), (size_t)G * sizeof(double), hipMemcpyHostToDevice);
    };
    if (c->rows >= 2) {
        HIPCHK(fill(blk, 2.0 * r));
        HIPCHK(fill(blk + (c->rows - 1) * G, 2.0 * r + 1.0));
    } else {
        HIPCHK(fill(blk, 2.0 * r));
    }
    CGCHK(comm_rc(c, c->comm->exchange_rows(s, blk, c->rows, G, c->halo_d[1][0], c->halo_d[1][1])));
    HIPCHK(hipStreamSynchronize(s));
    auto expect = [&](const double *src, double tag, const char *what) -> int {
        if (hipMemcpy(row.data(), src, (size_t)G * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
            return fail(CG_ERR_HIP, "row self-test: read-back failed");
        for (long j = 0; j < G; ++j)
            if (row[(size_t)j] != tag + 1e-6 * (double)j)
                return fail(CG_ERR_COMM, "row self-test on rank %d: %s, column %ld holds %.9g, expected %.9g", r, what, j,
                            row[(size_t)j], tag + 1e-6 * (double)j);
        return CG_OK;
    };
    if (c->has_top()) CGCHK(expect(c->halo_d[1][0], 2.0 * (r - 1) + (c->rows >= 2 ? 1.0 : 0.0), "row from above"));   // equal blocks
    if (c->has_bot()) CGCHK(expect(c->halo_d[1][1], 2.0 * (r + 1), "row from below"));

    // loopback through the point-to-point primitives
    HIPCHK(fill(blk, 77.0));
    CGCHK(comm_rc(c, c->comm->loopback(s, blk, blk + G * (c->rows >= 2 ? 1 : 0), c->rows >= 2 ? G : 0)));
    HIPCHK(hipStreamSynchronize(s));
    if (c->rows >= 2) CGCHK(expect(blk + G, 77.0, "loopback row"));

    // leave the scratch rows as cg_create left them
    HIPCHK(hipMemset(c->d_alloc[1], 0, c->padded_bytes));
    HIPCHK(hipStreamSynchronize(nullptr));
    return CG_OK;
}

} // extern "C"
