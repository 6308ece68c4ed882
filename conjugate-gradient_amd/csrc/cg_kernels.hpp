// Device code of the MI355X (gfx950) conjugate-gradient hot path.
//
// Everything here is fp64 and is compiled with -ffp-contract=off: the reference
// rounds every product before the following add (cg.c:63-87, utils.c:27-31,
// utils.c:373-379 compiled as C99 without FMA contraction), so must we.
//
// Kernel inventory (reference lines each one replaces):
//   k_stencil          the tiled kernel; its MODE selects what rides on the stencil walk:
//                      0  q = A d, d.q partials                     utils.c:15-35, :74-104, :373-379
//                      1  d' = beta*d - g first (stored), then as 0 cg.c:81-87
//                      2  q recomputed, u += tau*d, g += tau*q, g.g partials      cg.c:63-76, :79
//                      3  as 1 without storing d': a pure read pass (lean schedule)
//                      4  d' again (stored), g' = g + tau*A d' (stored), g'.g' partials
//                      5  as 4 plus u = (u + tau_prev*d) + tau*d' for two iterations at once
//   k_update           u += tau*d ; g += tau*q ; g.g partials      cg.c:63-76,:79
//   k_dupdate          d = beta*d - g                              cg.c:81-87
//   k_uflush, k_umark  u updates the lean schedule still owes when a solve ends
//   k_dot_partial      block partials of v.w                       utils.c:373-379
//   k_dot_strict       v.w with ONE accumulator in index order     utils.c:373-379
//   k_finish           fixed-order sum of block partials + the scalar recurrences
//                      tau = q0/(d.q), beta = q1/q0, q0 = q1       cg.c:62,:80,:89
//   k_pack_block, k_tree_sum   the dot product in the layout of the reference on s*s ranks
//                      (per-block sequential sums, MPI_Allreduce's combination)   utils.c:368-381,:408-417
//   k_init_d, k_init_g problem set-up                              utils.c:319-350
//
// Reduction order of the fast (non-strict) dot products, fixed for a given grid
// size and launch geometry, hence run-to-run deterministic:
//   1. each thread adds the products of its own points in the order it visits
//      them (rows ascending; within a row its VEC columns left to right; for the
//      flat kernels ascending index with stride gridDim*256*VEC);
//   2. the 64 lanes of a wavefront are combined by the butterfly
//      v += shfl_down(v, 32), 16, 8, 4, 2, 1 (lane 0 holds the wave's sum);
//   3. thread 0 adds the four wave sums in wave order 0,1,2,3 -> one partial per
//      workgroup, stored at partials[tile or block index];
//   4. k_finish: thread t of 256 adds partials[t], partials[t+256], ... in that
//      order, then steps 2 and 3 again;
//   5. multi-GPU: every rank's partials (step 3) are concatenated in rank order 0..P-1 by an
//      all-gather, and step 4 runs over the concatenation on every rank.
#pragma once
#include <hip/hip_runtime.h>

namespace cg {

constexpr int kBlock = 256;           // 4 wavefronts of 64 lanes
constexpr int kWaves = kBlock / 64;

// Device-resident scalar state of one solve.  Only k_finish writes it (and k_umark its
// u_applied); the other kernels read tau / beta / done in their prologue, so the host
// never has to look at a scalar inside the iteration loop (graph-capturable).
struct Scalars {
    double q0;          // g.g of the previous iteration            (cg.c:55,:89)
    double q1;          // g.g of this iteration                    (cg.c:79)
    double dq;          // d.q                                      (cg.c:61)
    double tau;         // q0/dq                                    (cg.c:62)
    double beta;        // q1/q0                                    (cg.c:80)
    double q_init;      // g0.g0
    double stop;        // stop when q1 <= stop (0 = never)
    double dot;         // result slot of the stand-alone dot entry point
    long long iter;     // iterations completed
    int done;           // 1: set by k_finish when the stop rule fires; 2: a dot product could not be completed on the
                        // fast path (`fault`) -- either way every later kernel returns at once, and the host looks
    int fault;          // 1 + which dot product (0 = d.q, 1 = g.g) of iteration `iter` is waiting to be completed the slow way
    double tau_ring[8]; // tau of iteration i at [i & 7]: deferred u updates read them later
    long long u_applied;// iterations whose tau*d has been added into u (lean schedule defers them)
    long long iter_base;// value of `iter` when the current solve began: a solve that continues the previous one
                        // keeps counting (the ring and u_applied use absolute numbers), its history starts here
    // Strict order, speculative summaries: predicted ratio of the coming dot product to the previous one of its
    // kind, by which last iteration's prefix sums are scaled into this iteration's exponent hypotheses.
    //   [0] d.q: tau oscillates with period two while q0 is known, so d.q ~ q0 / tau(two iterations ago);
    //   [1] g.g: with g' = g + tau*q and g.q = -d.q, g'.g' = tau^2 q.q - q0 up to rounding; q.q rides on pass 1.
    // Predictions only steer the hypotheses: a bad one costs time, never a bit.
    double rho[2];
    double qq;          // q.q of this iteration (fast order; prediction input only)
};

// ---------------------------------------------------------------- reductions --

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;                         // valid in lane 0
}

// Sum over the wavefront in NO particular order, result valid in lane 63 only.  Data-parallel
// primitives (row shifts inside each row of 16 lanes, then two row broadcasts) instead of LDS
// permutes: a dozen and a half VALU instructions and nothing to wait for.  For approximate
// sums only -- the documented reduction order of the dot products uses wave_sum.
__device__ __forceinline__ double wave_sum_unordered(double v)
{
#define CG_DPP_ADD(CTRL, ROW_MASK)                                                                     \
    do {                                                                                               \
        const int lo_ = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true); \
        v += __hiloint2double(hi_, lo_);                                                               \
    } while (0)
    CG_DPP_ADD(0x111, 0xf);           // row_shr:1   (lanes shifted in from outside the row read 0)
    CG_DPP_ADD(0x112, 0xf);           // row_shr:2
    CG_DPP_ADD(0x114, 0xf);           // row_shr:4
    CG_DPP_ADD(0x118, 0xf);           // row_shr:8   -> lane 15 of each row holds the row's sum
    CG_DPP_ADD(0x142, 0xa);           // row_bcast:15 into rows 1 and 3
    CG_DPP_ADD(0x143, 0xc);           // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef CG_DPP_ADD
    return v;
}

// Sum over the workgroup in the documented order; result valid in thread 0.
__device__ __forceinline__ double block_sum(double v, double *wave_slots)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_slots[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
        s = wave_slots[0];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) s += wave_slots[w];
    }
    return s;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the
// vector-memory counter (s_waitcnt vmcnt(0)), which would retire every prefetched
// global load at each row step and serialise the kernel on HBM latency; here the
// global loads stay in flight across the barrier.
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// XCD-aware placement: workgroups b, b+8, b+16, ... share an XCD (and its L2),
// so hand each of those eight groups a contiguous run of tiles.  Bijective for
// any tile count.  Affects speed only, never results.
__device__ __forceinline__ long xcd_remap(long b, long ntiles)
{
    const long q = ntiles >> 3, r = ntiles & 7;
    const long x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ------------------------------------------------------------------- stencil --

// The 3x3 integer coefficient array of a stencil_struct (utils.h:3-7, cg.c:33), converted to double
// as utils.c:27-31 converts its entries per use.  The reference's kernels read entries 1, 3, 4, 5, 7
// (up, left, centre, right, down) only; the four corner entries are honoured here as well (`nine`
// = any of them is non-zero), which the reference's own code never does.
struct Coef9 {
    double ul, up, ur, left, centre, right, dl, down, dr;
    int nine;
};

struct StencilArgs {
    const double *d_in;               // rows x G block of the direction vector
    const double *g;                  // rows x G block of the gradient (fused d-update)
    const double *d_top, *d_bot;      // halo rows of d_in, nullptr at the global boundary
    const double *g_top, *g_bot;      // halo rows of g (fused d-update only)
    double *d_out;                    // updated direction (fused d-update only)
    double *d_top_out, *d_bot_out;    // updated halo rows of d (fused d-update, multi-GPU only)
    double *q;                        // A d (may be nullptr when q is recomputed later)
    double *u, *gw;                   // MODE 2: solution and gradient, updated in place
    double *g_out;                    // MODE 4: updated gradient (the other of two gradient buffers)
    double *partials;                 // one partial dot product per tile
    double *partials2;                // NH > 0, MODE 3: one partial of q.q per tile (prediction input for g.g)
    const Scalars *sc;
    long G;                           // grid side = row length
    long rows;                        // rows in this block
    long row0;                        // global index of the block's first row
    int nstrips;                      // column strips per row of tiles
    int tile_h;                       // rows per tile
    int remap;                        // 1 = XCD-aware tile order
    // A launch covers rows [row_lo, row_hi) of the block in tiles of tile_h rows counted from row_lo
    // (multi-GPU overlap: the rows next to a halo run in launches of their own, before or after the rest).
    long row_lo, row_hi;
    long part_base;                   // slot of this launch's first tile in `partials`
    long ntiles_launch;               // tiles this launch covers (>= its workgroup count)
    Coef9 coef;                       // stencil coefficients
    int store_products;               // WRITE_Q kernels: store d*q (rounded) where q would go
    int plain_d;                      // MODE 3: d_in holds the direction itself (the iteration that starts a solve): beta and g stay out of it
    // Strict order, speculative summaries (kernels instantiated with NH = 2 only).  The unit is one workgroup's part
    // of a row, kUnit = 512 consecutive products; unit (strip, row r) has index strip * rows + r.  hyp[unit] is last
    // iteration's prefix sum at the unit; scaled by the predicted growth sc->rho[hyp_which] it gives the LOWER of two
    // hypotheses e0, e0+1 (hyp_exponent) for the exponent of the sequential sum when it reaches the unit.  The kernel
    // stores one record of four doubles per unit in rec[4 * unit ...]: the plain sum of the unit's products (the
    // approximate sum), for each hypothesis the sum D of the products rounded to that binade's ulp (in ulps; a NaN
    // marks a unit with a half-way case), and for e0 an upper bound Q0 of the sum of their magnitudes (not written
    // where every product is >= 0, g.g: Q = D there).
    const double *hyp;
    int hyp_which;                    // 0 = d.q, 1 = g.g
    double *rec;
    double *sub;                      // VEC 2 and G % 128 == 0, or nullptr: sum of the products d.q of
                                      // (MODE 2, 4, 5: g.g) of every aligned 128-element span (one wavefront's part of a
                                      // row), at sub[(column/128)*rows + r] -- approximate sums for the strict-order dot
};

// Streaming accesses carry the non-temporal hint (global_load/store ... nt): every
// vector is touched once per kernel and is far larger than L2 + Infinity Cache, and
// measured on MI355X the hint is worth 8-10 % of HBM bandwidth on mixed read/write
// streams (5.3 -> 5.8 TB/s) and 15 % on pure reads (6.2 -> 7.2 TB/s).
typedef double v2d __attribute__((ext_vector_type(2)));

template <int VEC> struct Pack;
template <> struct Pack<1> {
    double v[1];
    __device__ __forceinline__ void load(const double *p) { v[0] = __builtin_nontemporal_load(p); }
    __device__ __forceinline__ void store(double *p) const { __builtin_nontemporal_store(v[0], p); }
};
template <> struct Pack<2> {
    double v[2];
    __device__ __forceinline__ void load(const double *p)
    {
        const v2d t = __builtin_nontemporal_load(reinterpret_cast<const v2d *>(p));
        v[0] = t.x; v[1] = t.y;
    }
    __device__ __forceinline__ void store(double *p) const
    {
        v2d t; t.x = v[0]; t.y = v[1];
        __builtin_nontemporal_store(t, reinterpret_cast<v2d *>(p));
    }
};

// The five-point sum in the reference's order: up, left, centre, right, down,
// added one by one to a zero accumulator (utils.c:25-31).
__device__ __forceinline__ double five_point(const Coef9 &c, double up, double left, double centre,
                                             double right, double down)
{
    double acc = 0.0;
    acc += c.up * up;
    acc += c.left * left;
    acc += c.centre * centre;
    acc += c.right * right;
    acc += c.down * down;
    return acc;
}

// All nine entries, the same loop continued in array order 0..8: the row above left to right, the
// point's own row, the row below.  With zero corners every extra term adds a (signed) zero, so the
// value equals five_point's bit for bit; the kernels still take the five-term path then.
__device__ __forceinline__ double nine_point(const Coef9 &c, double ul, double up, double ur, double left,
                                             double centre, double right, double dl, double down, double dr)
{
    double acc = 0.0;
    acc += c.ul * ul;
    acc += c.up * up;
    acc += c.ur * ur;
    acc += c.left * left;
    acc += c.centre * centre;
    acc += c.right * right;
    acc += c.dl * dl;
    acc += c.down * down;
    acc += c.dr * dr;
    return acc;
}

// ---- buffer addressing -------------------------------------------------------
// Every global access of the tiled kernel goes through a per-row buffer descriptor
// (base = first element of the row, num_records = row bytes, or 0 when the row does
// not exist / must not be written).  The hardware range check then does all the
// masking -- columns past the grid, halo rows beyond the global boundary, rows that
// belong to another tile, the strip-edge column that only two lanes fetch -- with
// no branch: an out-of-range lane loads 0 and generates no memory traffic, an
// out-of-range store is dropped.  The loop body stays one basic block, so the
// compiler can keep PF rows of loads in flight behind counted s_waitcnt vmcnt(N).
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
constexpr int kRsrcFlags = 0x00020000;    // gfx9 raw buffer, 32-bit data format
constexpr int kAuxNT = 2;                 // non-temporal (streaming) cache policy
constexpr int kOOB = 0x7fffff00;          // byte offset beyond any row: lane is masked off

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const double *row, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(row), 0, row != nullptr ? bytes : 0, kRsrcFlags);
}

// Same for a row pointer known to be valid (no null test, so no scalar branch).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc_nn(const double *row, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(row), 0, bytes, kRsrcFlags);
}

template <int VEC> struct BufIO;
template <> struct BufIO<1> {
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int off, double (&v)[1])
    {
        v[0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, kAuxNT));
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int off, const double (&v)[1])
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v[0]), r, off, 0, kAuxNT);
    }
};
template <> struct BufIO<2> {
    static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t r, int off, double (&v)[2])
    {
        const v2d t = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, kAuxNT));
        v[0] = t.x; v[1] = t.y;
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t r, int off, const double (&v)[2])
    {
        v2d t; t.x = v[0]; t.y = v[1];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), r, off, 0, kAuxNT);
    }
};
// The same with a scalar byte offset on top (one descriptor for a whole tile, the row picked by the scalar part).
__device__ __forceinline__ void buf_load2s(__amdgpu_buffer_rsrc_t r, int off, int soff, double (&v)[2])
{
    const v2d t = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(r, off, soff, kAuxNT));
    v[0] = t.x; v[1] = t.y;
}
__device__ __forceinline__ void buf_load2s(__amdgpu_buffer_rsrc_t r, int off, int soff, double (&v)[1])
{
    v[0] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, kAuxNT));
}
__device__ __forceinline__ double buf_load1s(__amdgpu_buffer_rsrc_t r, int off, int soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0));
}
__device__ __forceinline__ double buf_load1(__amdgpu_buffer_rsrc_t r, int off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0));
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, int off, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, off, 0, kAuxNT);
}

// Raw operands of one grid row as a thread sees them: its VEC own columns plus,
// for the first / last thread of the strip, the column just outside the strip.
template <int VEC>
struct RawRow {
    double d[VEC];                    // direction
    double g[VEC];                    // gradient (MODE 1: operand of the d update; MODE 2: updated)
    double u[VEC];                    // solution (MODE 2)
    double ed, eg;                    // edge column (threads 0 and kBlock-1 only)
    double hyp;                       // NH > 0: last iteration's prefix sum at this wavefront's span of the row
};

// One workgroup walks down a column strip of kBlock*VEC columns with a rolling
// three-row register window; left/right neighbours go through one LDS row.
// PF rows of global loads are kept in flight ahead of the row being consumed.
//
// MODE 0: q = A d, partials of d.q                      (stencil entry points)
// MODE 1: d' = beta*d - g, q = A d', partials of d'.q   (fused d-update)
// MODE 2: q = A d recomputed, u += tau*d, g += tau*q, partials of g.g
// MODE 3: MODE 1 without storing d' (reads only): partials of d'.q      (lean schedule, first pass;
//         strict order: WRITE_Q with store_products keeps the rounded products d'*q)
// MODE 4: d' = beta*d - g recomputed and stored, q = A d' recomputed, g' = g + tau*q stored to
//         g_out, partials of g'.g'                                       (lean schedule, second pass)
// MODE 5: MODE 4 plus the solution update of TWO iterations, u = (u + tau_prev*d) + tau*d':
//         the previous iteration's direction d is this pass's input anyway and d' is in
//         registers, so u costs one read and one write per two iterations
//
// CONTIG: the halo rows sit in memory directly above and below the block (row r of
// every array is base + r*G for r = -1 .. rows; the solver context allocates its
// vectors that way, with zero rows where there is no neighbour).  Row addresses are
// then a pure linear function and the per-row scalar work shrinks to an add; the
// general form (separate halo pointers, possibly null) serves the stand-alone entry.
// What a wavefront contributes per row to the strict-order dot product when the exponent of the running sum
// is only hypothesised (NH = 2): for e0 = the hypothesis and e0 + 1, the integer sum D of its 2 x 64 products
// rounded to the grid 2^(e-52); Q0, an upper bound of the sum of the magnitudes of the integers under e0
// (summed in single precision and inflated by more than its rounding error; under e0 + 1 the magnitudes sum to
// at most Q0/2 + 96); and the plain sum of the products (the approximate span sum).  Sums of integers are exact
// in any order as long as they stay below 2^53, which Q0 < 2^52 certifies afterwards (every partial sum of the
// magnitudes is at most their total, and |partial D| <= partial Q); a half-way case anywhere in the span turns
// that hypothesis' D into a NaN.
// The sums over the 64 lanes are NOT formed row by row (four dependent data-parallel reductions per row cost
// more than the rest of the pass): the entries of four consecutive rows are reduced together through LDS
// (span4_push_lds), and the four wavefronts' sums are added to one record per workgroup and row (unit_combine).
struct SpanEntry {
    double sub, D0, D1;
    float qf;
    int pad;
};

template <int VEC, bool NONNEG>
__device__ __forceinline__ SpanEntry span_entry(const double (&p)[VEC], int e0, bool &tie0, bool &tie1)
{
    SpanEntry s = {0.0, 0.0, 0.0, 0.0f, 0};
    tie0 = false; tie1 = false;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        s.sub += p[v];
        const double x0 = ldexp(p[v], 52 - e0);
        const double r0 = rint(x0);
        tie0 |= (fabs(x0 - r0) == 0.5);
        s.D0 += r0;
        if (!NONNEG) s.qf += (float)fabs(r0);
        const double x1 = x0 * 0.5;                   // exact (a power of two), or far below 1/2 where it is not
        const double r1 = rint(x1);
        tie1 |= (fabs(x1 - r1) == 0.5);
        s.D1 += r1;
    }
    return s;
}

// lanes trade halves (32 lanes) or rows of 16 lanes and add: the last two steps of a sum over the wavefront
__device__ __forceinline__ void swap_add32(double &a, double b)     // a <- [a.lower + a.upper | b.lower + b.upper]  (halves of 32 lanes)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap_add16(double &a, double b)     // rows of 16 lanes: a <- [a0 + a1 | b0 + b1 | a2 + a3 | b2 + b3]
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    a = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// Batches of four rows through LDS: every lane parks its entries of the four rows in its wavefront's own LDS block
// -- quantity-and-row major, one double per lane, so the stores are conflict-free and nothing is held in registers
// between rows -- and after the fourth row lane 16*part + j adds up sixteen lanes' worth of quantity-and-row j; two
// swaps between halves and quarters of the wavefront finish the sums (about 60 instructions per batch; the sums are of
// integers, or approximate by design, so their order is free).  Lanes 0..15 then hold the wavefront's sixteen sums and
// leave them in the workgroup's exchange block X; wavefront 0 adds the four wavefronts' sums to the unit's (the
// workgroup's 512 products of the row) one row later, behind the row's barrier (unit_combine).
//
// Nothing goes to memory row by row: a store sits in the wavefront's vector-memory queue in issue order with the
// loads, which retire in order, so every wait for a prefetched row behind it also waits for the store to be
// acknowledged -- and small writes trickling into a read stream cost the memory system far more than their bytes.
// (Measured at 16384^2, read-only pass: 0.91 ms with a 16-lane store of four 32-byte pieces per wavefront every four
// rows, 0.69 ms without any store, 0.64 in the fast order.)  The unit records are staged in LDS, 256 rows at a time
// (aligned in the row number), and written by wavefront 0 in one contiguous run of 8 KiB per window.
constexpr int kSpanStride = 65;                        // doubles per quantity-and-row in the LDS block (64 lanes + 1: the reads of the four parts fall on different banks)
constexpr int kSpanLds = 16 * kSpanStride;             // doubles per wavefront
constexpr int kUnitWindow = 256;                       // rows of unit records staged before they are written
constexpr int kUnitLds = kWaves * 16 + 4 * kUnitWindow; // doubles per workgroup: the exchange block X, then the staged records
template <bool NONNEG>
__device__ __forceinline__ void span4_push_lds(double *R, double *X, int P, const SpanEntry &e, bool t0, bool t1)
{
    const int lane = threadIdx.x & 63;
    R[(0 + P) * kSpanStride + lane] = e.sub;
    R[(4 + P) * kSpanStride + lane] = t0 ? __builtin_nan("") : e.D0;
    R[(8 + P) * kSpanStride + lane] = t1 ? __builtin_nan("") : e.D1;
    if (!NONNEG) R[(12 + P) * kSpanStride + lane] = (double)e.qf;
    if (P == 3) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int j = lane & 15, part = lane >> 4;
        const double *src = R + j * kSpanStride + part * 16;
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = src[i];
#pragma unroll
        for (int w = 8; w > 0; w >>= 1)
#pragma unroll
            for (int i = 0; i < w; ++i) v[i] += v[i + w];
        double y = v[0];
        swap_add16(y, y);                               // every row of 16 lanes: its own part plus its neighbour's
        swap_add32(y, y);                               // ... plus the other half's pair
        if ((j >> 2) == 3) y = y * (1.0 + 0x1p-12) + 1.0;       // the bound Q: above the rounding of its single-precision parts
        if (lane < 16) X[(threadIdx.x >> 6) * 16 + lane] = (NONNEG && lane >= 12) ? 0.0 : y;   // [quantity][row of the batch]
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // all reads done before the block's next use
        __builtin_amdgcn_wave_barrier();
    }
}

// Wavefront 0, one row after a batch was parked (the row's barrier lies in between): the four wavefronts' sums of the
// batch whose first row is m0 (a multiple of four) added to unit sums and staged; a complete window of kUnitWindow rows
// (or, with `last`, whatever the tile's last window holds) goes to the records of rows [m_lo, m_hi) at rec_col =
// rec + 4 * (first unit of this strip).  X and S: the workgroup's exchange and staging blocks.
__device__ __forceinline__ void unit_combine(const double *X, double *S, int m0, bool last, int m_lo, int m_hi, double *rec_col)
{
    const int lane = threadIdx.x & 63;
    if (lane < 16) {
        const double y = (X[lane] + X[16 + lane]) + (X[32 + lane] + X[48 + lane]);
        S[((m0 & (kUnitWindow - 1)) + (lane & 3)) * 4 + (lane >> 2)] = y;
    }
    const bool full = (m0 & (kUnitWindow - 1)) == kUnitWindow - 4;
    if (!full && !last) return;
    const int wbase = m0 & ~(kUnitWindow - 1);
    if (wbase + kUnitWindow <= m_lo) return;            // a window above the tile (the rows before its first one)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v2d *dst = reinterpret_cast<v2d *>(rec_col + (long)wbase * 4);
#pragma unroll
    for (int i = 0; i < 4 * kUnitWindow / 128; ++i) {
        const int pair = i * 64 + lane;                 // two doubles: half a record
        const int m = wbase + (pair >> 1);
        const v2d v = reinterpret_cast<const v2d *>(S)[pair];
        if (m >= m_lo && m < m_hi) __builtin_nontemporal_store(v, dst + pair);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // all reads done before the window's next use
    __builtin_amdgcn_wave_barrier();
}

constexpr int kUnit = 512;            // products per unit of the speculative summaries: one workgroup's part of a row (VEC = 2)

constexpr int kHypNone = -30000;      // "no hypothesis": every non-zero product overflows under it, the span is redone later

// The lower of the two exponent hypotheses for a span: `guess` is last iteration's prefix sum at the span, rho the
// predicted growth of the total.  The two binades nearest to guess*rho are e-1, e when it lies in the lower half
// (by ratio) of its binade, e, e+1 otherwise.  The kernel that summarises and the kernel that selects both call
// this with the same arguments.
__device__ __forceinline__ int hyp_exponent(double guess, double rho)
{
    // read off the bits of |guess * rho| = f * 2^e, f in [1, 2): binades e-1, e when f < sqrt 2 (judged by the
    // upper 20 bits of the fraction, which is all a heuristic needs), e, e+1 otherwise
    const int hi = __double2hiint(guess * rho);
    const int be = (hi >> 20) & 0x7ff;
    const int e = be - 1023;
    const int lower = ((hi & 0xfffff) < 0x6a09e) ? e - 1 : e;
    return (be >= 23 && be <= 2023) ? lower : kHypNone;
}

template <int VEC, int MODE, bool WRITE_Q, int PF, bool CONTIG, int NH = 0>
// Resident workgroups per CU the register budget is cut for: one for the writing passes, two for the read-only pass,
// three for the read-only pass that also forms the strict-order span sums (bound by its instruction stream, not by
// memory: every further wavefront per SIMD helps) when it keeps at most two rows in flight (147 registers).
__global__ __launch_bounds__(kBlock, (MODE == 3) ? ((NH > 0 && PF <= 2) ? 3 : 2) : 1) void k_stencil(const StencilArgs a)
{
    constexpr bool FUSE_D = (MODE == 1 || MODE >= 3);                // operands are d and g, d' is formed here
    constexpr bool STORE_D = (MODE == 1 || MODE == 4 || MODE == 5);
    constexpr int W = kBlock * VEC;
    if (a.sc != nullptr && a.sc->done) return;

    __shared__ double row_lds[2][W + 2];
    __shared__ __attribute__((aligned(16))) double span_lds[(NH > 0) ? kWaves * kSpanLds + kUnitLds : 2];
    static_assert(NH == 0 || (CONTIG && VEC == 2 && PF <= 4), "the span sums need the context's layout, 16-byte rows and batches of four rows");
    __shared__ double wave_slots[kWaves];
    const int tid = threadIdx.x;
    const long nwg = (long)gridDim.x;
    // A launch may have fewer workgroups than tiles: workgroup b then walks tiles
    // b', b' + nwg, b' + 2 nwg, ... one after the other (b' = its XCD-aware slot).
    for (long lt = a.remap ? xcd_remap(blockIdx.x, nwg) : (long)blockIdx.x; lt < a.ntiles_launch; lt += nwg) {
    const long strip = lt % a.nstrips, lrt = lt / a.nstrips;
    const long tile = a.part_base + lt;               // fixes the partial's slot
    const long G = a.G;
    const long c0 = strip * W;
    const long j = c0 + (long)tid * VEC;
    // row numbers are 32-bit (the scalar unit compares those directly; a block has far fewer than 2^31 rows)
    const int r0 = (int)(a.row_lo + lrt * a.tile_h);
    const int r1 = (r0 + a.tile_h < (int)a.row_hi) ? r0 + a.tile_h : (int)a.row_hi;
    const int nrows = (int)a.rows;
    const int rowbytes = (int)(G * 8);
    const bool left_edge = (tid == 0) && (c0 > 0);
    const bool right_edge = (tid == kBlock - 1) && (c0 + W < G);

    // Per-lane byte offsets inside a row; kOOB masks the lane off.
    const int off_main = (j < G) ? (int)(j * 8) : kOOB;            // VEC == 2 only with even G: all or nothing
    const int off_edge = left_edge ? (int)((c0 - 1) * 8) : (right_edge ? (int)((c0 + W) * 8) : kOOB);
    // q is written on interior columns only (the reference never touches the rest):
    // one vector store for lanes clear of the boundary, single-element stores for the
    // two lanes that hold a boundary column.
    int off_qv, off_q0 = kOOB, off_q1 = kOOB;
    if (VEC == 2) {
        off_qv = (j >= 1 && j + 1 <= G - 2) ? (int)(j * 8) : kOOB;
        if (j == 0 && G >= 3) off_q1 = 8;                          // lane holds columns 0,1: store column 1
        if (j == G - 2 && j >= 1) off_q0 = (int)(j * 8);           // lane holds columns G-2,G-1: store column G-2
    } else {
        off_qv = (j >= 1 && j <= G - 2) ? (int)(j * 8) : kOOB;
    }

    // NH > 0: this wavefront's column of the span arrays (span index = column block * rows + row)
    // NH > 0: this strip's column of the unit arrays (unit index = strip * rows + row)
    const long unit_col = (NH > 0) ? strip * a.rows : 0;
    const __amdgpu_buffer_rsrc_t hyp_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>((NH > 0) ? a.hyp + unit_col : nullptr), 0,
                                          (NH > 0) ? (int)(a.rows * 8) : 0, kRsrcFlags);
    const double rho = (NH > 0) ? a.sc->rho[a.hyp_which] : 1.0;

    double beta = 0.0, tau = 0.0;
    // plain_d: d' = 1*d - 0 = d, bit for bit (g is read through a descriptor of no bytes: zeros, no memory traffic)
    const bool plain_d = FUSE_D && a.plain_d != 0;
    if (FUSE_D) beta = plain_d ? 1.0 : a.sc->beta;
    double tau_prev = 0.0;
    if (MODE == 2 || MODE == 4 || MODE == 5) tau = a.sc->tau;
    if (MODE == 5) tau_prev = a.sc->tau_ring[(a.sc->iter - 1) & 7];   // iteration i-1's tau; this pass belongs to iteration i

    // Row r of the block, r in [-1, rows]; -1 and rows are the halo rows (nullptr at
    // the global boundary).  Everything here is wave-uniform scalar arithmetic.
    auto row_ptr = [&](const double *base, const double *top, const double *bot, int r) -> const double * {
        return (r < 0) ? top : ((r >= nrows) ? bot : base + (long)r * G);
    };
    // CONTIG: one descriptor per operand for the whole tile -- rows r0-1 .. r1 of the inputs, rows r0 .. r1-1 of what
    // only the tile's own rows need -- and the row chosen by a scalar byte offset, clamped to the descriptor's size:
    // a row outside the tile then fails the hardware range check like a column outside the grid (the host keeps
    // tiles below 2^31 bytes).  That is one scalar multiply and one minimum per row instead of a descriptor per
    // row and operand.
    const unsigned in_bytes = (unsigned)(r1 - r0 + 2) * (unsigned)rowbytes, own_bytes = (unsigned)(r1 - r0) * (unsigned)rowbytes;
    const __amdgpu_buffer_rsrc_t td = row_rsrc_nn(a.d_in + (long)(r0 - 1) * G, CONTIG ? (int)in_bytes : 0);
    const __amdgpu_buffer_rsrc_t tg = row_rsrc_nn((FUSE_D ? a.g : a.d_in) + (long)(r0 - 1) * G, (CONTIG && FUSE_D && !plain_d) ? (int)in_bytes : 0);
    // MODE 2 without u (a.u == nullptr: the update is left to the next iteration's paired one): descriptors of no
    // bytes, so the loads deliver zeros and the stores are dropped without touching memory
    const bool has_u = (MODE == 5) || (MODE == 2 && a.u != nullptr);
    const __amdgpu_buffer_rsrc_t tu = row_rsrc_nn(((MODE == 2 || MODE == 5) ? a.u : a.d_in) + (long)r0 * G,
                                                  (CONTIG && has_u) ? (int)own_bytes : 0);
    const __amdgpu_buffer_rsrc_t tgw = row_rsrc_nn(((MODE == 2) ? a.gw : a.d_in) + (long)r0 * G, (CONTIG && MODE == 2) ? (int)own_bytes : 0);
    auto umin = [](unsigned x, unsigned y) { return x < y ? x : y; };
    auto load_row = [&](int r, RawRow<VEC> &raw) {
        if constexpr (CONTIG) {
            const int so = (int)umin((unsigned)(r - r0 + 1) * (unsigned)rowbytes, in_bytes);      // r >= r0 - 1
            buf_load2s(td, off_main, so, raw.d);
            raw.ed = buf_load1s(td, off_edge, so);
            if (FUSE_D) {
                buf_load2s(tg, off_main, so, raw.g);
                raw.eg = buf_load1s(tg, off_edge, so);
            } else {
                raw.eg = 0.0;
            }
            if (NH > 0) raw.hyp = buf_load1(hyp_rsrc, (r >= 0 && r < nrows) ? r * 8 : kOOB);
            if (MODE == 2 || MODE == 5) {
                const int sou = (int)umin((unsigned)(r - r0) * (unsigned)rowbytes, own_bytes);    // r0 - 1 wraps round and is clamped
                buf_load2s(tu, off_main, sou, raw.u);
                if (MODE == 2) buf_load2s(tgw, off_main, sou, raw.g);
            }
        } else {
            const bool live = r <= r1;                    // rows past the tile's lower halo are not needed
            const double *pd = row_ptr(a.d_in, a.d_top, a.d_bot, r);
            const __amdgpu_buffer_rsrc_t rd = row_rsrc(pd, live ? rowbytes : 0);
            BufIO<VEC>::load(rd, off_main, raw.d);
            raw.ed = buf_load1(rd, off_edge);
            if (FUSE_D) {
                const double *pg = row_ptr(a.g, a.g_top, a.g_bot, r);
                const __amdgpu_buffer_rsrc_t rg = row_rsrc(pg, (live && !plain_d) ? rowbytes : 0);
                BufIO<VEC>::load(rg, off_main, raw.g);
                raw.eg = buf_load1(rg, off_edge);
            } else {
                raw.eg = 0.0;
            }
            if (NH > 0) raw.hyp = buf_load1(hyp_rsrc, (r >= 0 && r < nrows) ? r * 8 : kOOB);
            if (MODE == 2 || MODE == 5) {
                const bool own = (r >= r0) && (r < r1);
                BufIO<VEC>::load(row_rsrc_nn(a.u + (long)r * G, (own && has_u) ? rowbytes : 0), off_main, raw.u);
                if (MODE == 2) BufIO<VEC>::load(row_rsrc_nn(a.gw + (long)r * G, own ? rowbytes : 0), off_main, raw.g);
            }
        }
    };

    double up[VEC], mid[VEC], mid_u[VEC], mid_g[VEC], mid_dp[VEC], midL = 0.0, midR = 0.0, upL = 0.0, upR = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) { up[v] = 0.0; mid[v] = 0.0; mid_u[v] = 0.0; mid_g[v] = 0.0; mid_dp[v] = 0.0; }
    double acc = 0.0, acc2 = 0.0;
    double mid_hyp = 0.0;

    // NH > 0: the span sums of four rows are reduced over the wavefront together (span4_push_lds); a kernel that keeps
    // fewer than four rows in flight goes through its ring twice or four times per batch.  Batches begin at a multiple
    // of four rows, so that 64 of them fill an aligned window of the staged records; the up to three extra rows above
    // the tile's upper halo fail the range check like any row outside the tile.
    constexpr int BATCH = (NH > 0) ? 4 : PF;
    const int rb0 = (NH > 0) ? (((r0 - 2) & ~3) + 1) : (r0 - 1);
    double *const span_R = span_lds + (tid >> 6) * kSpanLds, *const span_X = span_lds + kWaves * kSpanLds, *const span_S = span_X + kWaves * 16;
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    RawRow<VEC> ring[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) load_row(rb0 + p, ring[p]);
    int buf = 0;
    // Rows r0-1 .. r1 in groups of PF; a group may run past r1, those rows are fully
    // masked (zero loads, dropped stores, zero contributions to the partial sum).
    int rb = rb0;
    for (; rb <= r1; rb += BATCH) {
#pragma unroll
        for (int p = 0; p < BATCH; ++p) {
            const int r = rb + p;
            const RawRow<VEC> raw = ring[p % PF];
            load_row(r + PF, ring[p % PF]);           // refill the slot: PF rows in flight

            double cur[VEC], edge;
            if (FUSE_D) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) cur[v] = beta * raw.d[v] - raw.g[v];
                edge = beta * raw.ed - raw.eg;
                // Owned rows go to d_out; a neighbour's boundary row (known here only as a
                // halo) is kept up to date locally, so that only g has to travel.
                if (!STORE_D) {
                } else if (CONTIG) {
                    const bool wr = (r >= r0 && r < r1) || (r == -1 && a.d_top_out != nullptr) ||
                                    (r >= nrows && r <= r1 && a.d_bot_out != nullptr);
                    BufIO<VEC>::store(row_rsrc_nn(a.d_out + (long)r * G, wr ? rowbytes : 0), off_main, cur);
                } else {
                    const double *dst = (r >= r0 && r < r1) ? a.d_out + (long)r * G
                                      : ((r < 0) ? a.d_top_out : ((r >= nrows && r <= r1) ? a.d_bot_out : nullptr));
                    BufIO<VEC>::store(row_rsrc(dst, rowbytes), off_main, cur);
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) cur[v] = raw.d[v];
                edge = raw.ed;
            }

            // Left/right neighbours travel through one LDS row (double-buffered, so a
            // single barrier per row orders both the writes before the reads and the
            // reads before the buffer's next reuse two rows later).
            double *L = row_lds[buf];
#pragma unroll
            for (int v = 0; v < VEC; ++v) L[1 + tid * VEC + v] = cur[v];
            if (tid == 0) L[0] = edge;
            if (tid == kBlock - 1) L[W + 1] = edge;
            lds_barrier();
            const double curL = L[tid * VEC];
            const double curR = L[tid * VEC + VEC + 1];
            buf ^= 1;
            // the batch parked one row ago (the barrier above lies in between): its unit sums, staged
            if (NH > 0 && p == 0 && VEC == 2 && a.sub != nullptr && rb > rb0 && wave0)
                unit_combine(span_X, span_S, rb - 5, false, r0, r1, a.rec + 4 * unit_col);

            // Rows r-2 (up), r-1 (mid) and r (cur) are now in registers: emit row r-1.
            const int m = r - 1;
            const long gi = a.row0 + m;
            const bool row_own = (m >= r0) && (m < r1);
            const bool row_in = row_own && (gi >= 1) && (gi <= G - 2);
            double qv[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double left = (v == 0) ? midL : mid[v - 1];
                const double right = (v == VEC - 1) ? midR : mid[v + 1];
                const bool in = row_in && (j + v >= 1) && (j + v <= G - 2);
                if (a.coef.nine) {                                         // wave-uniform
                    const double ul = (v == 0) ? upL : up[v - 1], ur = (v == VEC - 1) ? upR : up[v + 1];
                    const double dl = (v == 0) ? curL : cur[v - 1], dr = (v == VEC - 1) ? curR : cur[v + 1];
                    const double q9 = nine_point(a.coef, ul, up[v], ur, left, mid[v], right, dl, cur[v], dr);
                    qv[v] = in ? q9 : 0.0;
                } else {
                    const double q5 = five_point(a.coef, up[v], left, mid[v], right, cur[v]);     // always evaluated: a select
                    qv[v] = in ? q5 : 0.0;                                                        // costs less than a branch
                }
            }
            if (MODE == 2) {
                double pu[VEC], pg[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    pu[v] = mid_u[v] + tau * mid[v];
                    pg[v] = mid_g[v] + tau * qv[v];
                    acc += pg[v] * pg[v];             // masked rows/lanes hold g = 0
                }
                BufIO<VEC>::store(row_rsrc_nn(a.u +(long)m * G, (row_own && has_u) ? rowbytes : 0), off_main, pu);
                BufIO<VEC>::store(row_rsrc_nn(a.gw +(long)m * G, row_own ? rowbytes : 0), off_main, pg);
                if (VEC == 2 && a.sub != nullptr) {                                // wave-uniform
                    double prod[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) prod[v] = pg[v] * pg[v];
                    double span = prod[0];
                    span += prod[VEC - 1];
                    if (NH == 0) {
                        span = wave_sum_unordered(span);
                        if ((tid & 63) == 63 && row_own && j < G) a.sub[(j >> 7) * a.rows + m] = span;
                    } else {
                        bool t0, t1;
                        const SpanEntry e = span_entry<VEC, true>(prod, hyp_exponent(mid_hyp, rho), t0, t1);
                        span4_push_lds<true>(span_R, span_X, p, e, t0, t1);
                    }
                }
            } else if (MODE == 4 || MODE == 5) {
                double pg[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    pg[v] = mid_g[v] + tau * qv[v];
                    acc += row_own ? pg[v] * pg[v] : 0.0;   // halo rows carry g too here: not ours to count
                }
                BufIO<VEC>::store(row_rsrc_nn(a.g_out + (long)m * G, row_own ? rowbytes : 0), off_main, pg);
                if (VEC == 2 && a.sub != nullptr) {                                // wave-uniform
                    double prod[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) prod[v] = pg[v] * pg[v];
                    double span = prod[0];
                    span += prod[VEC - 1];
                    if (NH == 0) {
                        span = wave_sum_unordered(span);
                        if ((tid & 63) == 63 && row_own && j < G) a.sub[(j >> 7) * a.rows + m] = span;
                    } else {
                        bool t0, t1;
                        const SpanEntry e = span_entry<VEC, true>(prod, hyp_exponent(mid_hyp, rho), t0, t1);
                        span4_push_lds<true>(span_R, span_X, p, e, t0, t1);
                    }
                }
                if (MODE == 5) {
                    double pu[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        pu[v] = mid_u[v] + tau_prev * mid_dp[v];     // cg.c:63-69 of the previous iteration
                        pu[v] = pu[v] + tau * mid[v];                // ... and of this one
                    }
                    BufIO<VEC>::store(row_rsrc_nn(a.u + (long)m * G, row_own ? rowbytes : 0), off_main, pu);
                }
            } else {
                double span = 0.0, prod[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    prod[v] = mid[v] * qv[v];                          // q = 0 off the interior
                    acc += prod[v];
                    if (NH > 0 && MODE == 3) acc2 += qv[v] * qv[v];
                    span += prod[v];
                }
                if (VEC == 2 && a.sub != nullptr) {                                // MODE 0, 1, 3; wave-uniform
                    if (NH == 0) {
                        span = wave_sum_unordered(span);
                        if ((tid & 63) == 63 && row_own && j < G) a.sub[(j >> 7) * a.rows + m] = span;
                    } else {
                        bool t0, t1;
                        const SpanEntry e = span_entry<VEC, false>(prod, hyp_exponent(mid_hyp, rho), t0, t1);
                        span4_push_lds<false>(span_R, span_X, p, e, t0, t1);
                    }
                }
                if (WRITE_Q) {
                    // q itself, or (strict order without a stored q) the rounded products d*q that the
                    // sequential sum adds up -- the one thing the strict evaluation needs from q
                    double out[VEC];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) out[v] = a.store_products ? mid[v] * qv[v] : qv[v];
                    const __amdgpu_buffer_rsrc_t rq = row_rsrc_nn(a.q + (long)m * G, row_in ? rowbytes : 0);
                    BufIO<VEC>::store(rq, off_qv, out);
                    if (VEC == 2) {
                        buf_store1(rq, off_q0, out[0]);
                        buf_store1(rq, off_q1, out[VEC - 1]);
                    }
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                up[v] = mid[v]; mid[v] = cur[v]; mid_u[v] = raw.u[v]; mid_g[v] = raw.g[v]; mid_dp[v] = raw.d[v];
            }
            if (NH > 0) mid_hyp = raw.hyp;
            upL = midL; upR = midR;
            midL = curL; midR = curR;
        }
    }
    if (NH > 0 && VEC == 2 && a.sub != nullptr) {          // the last batch (its first row: rb - 5), and the tile's last window of records
        lds_barrier();
        if (wave0) unit_combine(span_X, span_S, rb - 5, true, r0, r1, a.rec + 4 * unit_col);
    }

    const double s = block_sum(acc, wave_slots);
    if (tid == 0) a.partials[tile] = s;
    if (NH > 0 && MODE == 3) {
        __syncthreads();
        const double s2 = block_sum(acc2, wave_slots);
        if (tid == 0) a.partials2[tile] = s2;
    }
    }
}

// ------------------------------------------------------------- flat kernels --

// u += tau*d ; g += tau*q ; partials of g.g           (cg.c:63-76, :79)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_update(double *__restrict__ u, double *__restrict__ g,
                                                   const double *__restrict__ d,
                                                   const double *__restrict__ q, long count,
                                                   const Scalars *sc, double *partials, double *sub)
{
    if (sc->done) return;
    __shared__ double wave_slots[kWaves];
    const double tau = sc->tau;
    const long nvec = count / VEC;
    const long stride = (long)gridDim.x * kBlock;
    const long rounds = (nvec + stride - 1) / stride;       // every lane makes the same number of trips (wave_sum below)
    double acc = 0.0;
    for (long it = 0; it < rounds; ++it) {
        const long k = it * stride + (long)blockIdx.x * kBlock + threadIdx.x;
        double span = 0.0;
        if (k < nvec) {
            Pack<VEC> pu, pg, pd, pq;
            pu.load(u + k * VEC); pg.load(g + k * VEC); pd.load(d + k * VEC); pq.load(q + k * VEC);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                pu.v[v] = pu.v[v] + tau * pd.v[v];
                pg.v[v] = pg.v[v] + tau * pq.v[v];
                acc += pg.v[v] * pg.v[v];
                span += pg.v[v] * pg.v[v];
            }
            pu.store(u + k * VEC); pg.store(g + k * VEC);
        }
        // VEC 2: a wavefront's 64 lanes hold 128 consecutive elements starting at a multiple of 128:
        // their g.g goes to sub[] as approximate input for the strict-order dot product
        if (VEC == 2 && sub != nullptr) {
            span = wave_sum_unordered(span);
            const long first = (k - (threadIdx.x & 63)) * 2;
            if ((threadIdx.x & 63) == 63 && first < count) sub[first >> 7] = span;
        }
    }
    const double s = block_sum(acc, wave_slots);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// d = beta*d - g                                       (cg.c:81-87)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_dupdate(double *__restrict__ d, const double *__restrict__ g,
                                                    long count, const Scalars *sc, int honour_done)
{
    if (honour_done && sc->done) return;
    const double beta = sc->beta;
    const long nvec = count / VEC;
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < nvec; k += stride) {
        Pack<VEC> pd, pg;
        pd.load(d + k * VEC); pg.load(g + k * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) pd.v[v] = beta * pd.v[v] - pg.v[v];
        pd.store(d + k * VEC);
    }
}

// Left-over solution update of the lean schedule (a solve that ends between two paired
// updates): u += tau_a*d_a, then + tau_b*d_b, in ascending iteration order -- the same
// additions, in the same order, that cg.c:63-69 performs one iteration at a time.  Entry e is iteration sc->u_applied + e; the
// host names the buffer that still holds that iteration's direction.  Only iterations that
// really ran (index < sc->iter) are applied: launches queued behind a fired stop rule find
// nothing to do.  k_umark, launched right after, advances sc->u_applied.
struct UFlushArgs {
    const double *d[2];               // direction of iteration sc->u_applied + e, e = 0 .. n-1
    int n;
};

template <int VEC, int N>
__device__ __forceinline__ void uflush_body(double *__restrict__ u, long count, const UFlushArgs &f, const double *tau)
{
    const long nvec = count / VEC;
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < nvec; k += stride) {
        Pack<VEC> pu, pd[N];
        pu.load(u + k * VEC);
#pragma unroll
        for (int e = 0; e < N; ++e) pd[e].load(f.d[e] + k * VEC);      // all N+1 loads in flight together
#pragma unroll
        for (int e = 0; e < N; ++e)
#pragma unroll
            for (int v = 0; v < VEC; ++v) pu.v[v] = pu.v[v] + tau[e] * pd[e].v[v];
        pu.store(u + k * VEC);
    }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_uflush(double *__restrict__ u, long count, const UFlushArgs f,
                                                   const Scalars *sc)
{
    const long long ran = sc->iter, first = sc->u_applied;
    double tau[2];
    int nv = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if (e < f.n && first + e < ran) { tau[e] = sc->tau_ring[(first + e) & 7]; nv = e + 1; }
        else tau[e] = 0.0;
    switch (nv) {                      // wave-uniform: one straight-line body per entry count
    case 1: uflush_body<VEC, 1>(u, count, f, tau); break;
    case 2: uflush_body<VEC, 2>(u, count, f, tau); break;
    default: break;
    }
}

__global__ void k_umark(Scalars *sc, int n)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long want = sc->u_applied + n;
    sc->u_applied = want < sc->iter ? want : sc->iter;
}

// y += a*x and d = b*d - g with host-supplied scalars (stand-alone entry points).
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_axpy(double *__restrict__ y, const double *__restrict__ x,
                                                 double a, long count)
{
    const long nvec = count / VEC;
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < nvec; k += stride) {
        Pack<VEC> py, px;
        py.load(y + k * VEC); px.load(x + k * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) py.v[v] = py.v[v] + a * px.v[v];
        py.store(y + k * VEC);
    }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_xpay(double *__restrict__ d, const double *__restrict__ g,
                                                 double b, long count)
{
    const long nvec = count / VEC;
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < nvec; k += stride) {
        Pack<VEC> pd, pg;
        pd.load(d + k * VEC); pg.load(g + k * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) pd.v[v] = b * pd.v[v] - pg.v[v];
        pd.store(d + k * VEC);
    }
}

// Block partials of v.w                                (utils.c:373-379, fast order)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_dot_partial(const double *__restrict__ v,
                                                        const double *__restrict__ w, long count,
                                                        double *partials)
{
    __shared__ double wave_slots[kWaves];
    const long nvec = count / VEC;
    const long stride = (long)gridDim.x * kBlock;
    double acc = 0.0;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < nvec; k += stride) {
        Pack<VEC> pv, pw;
        pv.load(v + k * VEC); pw.load(w + k * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += pv.v[e] * pw.v[e];
    }
    const double s = block_sum(acc, wave_slots);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// v.w with a single accumulator in strict index order -- the reference's exact
// summation (utils.c:372-379).  w == nullptr: v already holds the rounded products.  One wavefront: the lanes fetch 64 consecutive
// products at a time (coalesced), then every lane replays the same 64 dependent
// additions, reading product l from lane l.  The result is bit-identical to the
// CPU loop; the price is one dependent fp64 add per element.
__global__ __launch_bounds__(64) void k_dot_strict(const double *__restrict__ v,
                                                   const double *__restrict__ w, long count,
                                                   double start, const double *start_dev, double *result)
{
    const int lane = threadIdx.x;
    double acc = (start_dev != nullptr) ? *start_dev : start;     // device-resident seed: chained ranks
    long base = 0;
    for (; base + 64 <= count; base += 64) {
        const double p = (w != nullptr) ? v[base + lane] * w[base + lane] : v[base + lane];
#pragma unroll
        for (int l = 0; l < 64; ++l) acc += __shfl(p, l, 64);
    }
    const long rem = count - base;
    if (rem > 0) {
        const double p = (lane < rem) ? ((w != nullptr) ? v[base + lane] * w[base + lane] : v[base + lane]) : 0.0;
        for (int l = 0; l < rem; ++l) acc += __shfl(p, l, 64);
    }
    if (lane == 0) *result = acc;
}

// Fixed-order sum of `count` partials and the scalar recurrences.
//   what 0: dq   = sum ; tau  = q0/dq                                  (cg.c:61-62)
//   what 1: q1   = sum ; beta = q1/q0 ; q0 = q1 ; history ; stop rule  (cg.c:79-80,:89)
//   what 4: as 1, for an iteration whose u update is deferred (sc->u_applied stays)
//   what 2: q0   = sum ; q_init = q0 ; stop = rtol^2*q0                (cg.c:55)
//   what 3: dot  = sum                                                 (stand-alone dot)
//   what 5: a solve that continues the previous one: q0 is the q1 the last iteration left (the same
//           g.g, cg.c:55 and :89 hold the same value), so nothing is summed; q_init, stop and the
//           history origin are set for the new solve
//   what 6: a solve that continues the previous one but sums g.g once more, as the reference's timed
//           region does (cg.c:55): q0, q_init and stop from the sum; the iteration count, the deferred
//           u updates and beta (the d update still owed needs it) stay
// With `strict_src` the sum is taken from there instead of the partials: one value (the
// strict-order result), or `nsrc` values added in index order.
__device__ __forceinline__ double sum_partials(const double *partials, long count, double *wave_slots)
{
    double acc = 0.0;
    for (long k = threadIdx.x; k < count; k += kBlock) acc += partials[k];
    return block_sum(acc, wave_slots);
}

__device__ __forceinline__ double sane_ratio(double r)
{
    return (r > 0x1p-200 && r < 0x1p200) ? r : 1.0;
}

// The scalar step that follows a dot product (one thread): s = the dot product, s2 = q.q of the same pass where
// have_s2 (prediction input only; count2 < 0 without it: several ranks walking in turn fall back on the last beta).
__device__ __forceinline__ void finish_scalars(Scalars *sc, int what, double s, double s2, bool have_s2, long count2, double rtol,
                                               double *hist, long hist_cap)
{
    if (what == 0) {
        sc->dq = s;
        sc->tau = sc->q0 / s;
        sc->tau_ring[sc->iter & 7] = sc->tau;
        if (have_s2) {
            sc->qq = s2;
            sc->rho[1] = sane_ratio((sc->tau * sc->tau * s2 - sc->q0) / sc->q0);
        } else if (count2 < 0) {
            sc->rho[1] = sane_ratio(sc->beta);          // several ranks: q.q is not known globally; the last ratio will do
        }
    } else if (what == 1 || what == 4) {
        // next iteration's d.q, predicted from the new q0 and the step length of the iteration before this one (the
        // step lengths of CG alternate; the first two iterations use their own: the very first step is far off the rest)
        sc->rho[0] = sane_ratio(s / sc->tau_ring[(sc->iter >= 2 ? sc->iter - 1 : sc->iter) & 7] / sc->dq);
        sc->q1 = s;
        sc->beta = s / sc->q0;
        sc->q0 = s;
        const long long it = sc->iter, rel = it - sc->iter_base;
        if (hist != nullptr && rel < hist_cap) hist[rel] = s;
        sc->iter = it + 1;
        if (what == 1) sc->u_applied = it + 1;
        if (sc->stop > 0.0 && s <= sc->stop) sc->done = 1;
    } else if (what == 2) {
        sc->q0 = s;
        sc->q1 = s;
        sc->q_init = s;
        sc->stop = rtol * rtol * s;
        sc->iter = 0;
        sc->iter_base = 0;
        sc->u_applied = 0;
        sc->done = 0;
    } else if (what == 6) {
        sc->q0 = s;
        sc->q1 = s;
        sc->q_init = s;
        sc->stop = rtol * rtol * s;
        sc->iter_base = sc->iter;
        sc->done = 0;
    } else if (what == 5) {
        sc->q_init = sc->q0;
        sc->stop = rtol * rtol * sc->q0;
        sc->iter_base = sc->iter;
        sc->done = 0;
    } else {
        sc->dot = s;
    }
}

__global__ __launch_bounds__(kBlock) void k_finish(const double *partials, long count,
                                                   const double *strict_src, int nsrc, Scalars *sc, int what,
                                                   double rtol, double *hist, long hist_cap,
                                                   const double *partials2 = nullptr, long count2 = 0)
{
    if ((what < 2 || what == 4) && sc->done) return;
    __shared__ double wave_slots[kWaves];
    double s = sum_partials(partials, (strict_src == nullptr && what != 5) ? count : 0, wave_slots);
    double s2 = 0.0;
    if (partials2 != nullptr) {                        // q.q partials of the speculative strict pass
        __syncthreads();
        s2 = sum_partials(partials2, count2, wave_slots);
    }
    if (threadIdx.x != 0) return;
    if (strict_src != nullptr) {
        s = strict_src[0];
        for (int r = 1; r < nsrc; ++r) s += strict_src[r];
    }
    finish_scalars(sc, what, s, s2, partials2 != nullptr, count2, rtol, hist, hist_cap);
}

// The same fixed-order sum, written to a caller-supplied device location.
__global__ __launch_bounds__(kBlock) void k_sum_to(const double *partials, long count, double *out)
{
    __shared__ double wave_slots[kWaves];
    const double s = sum_partials(partials, count, wave_slots);
    if (threadIdx.x == 0) *out = s;
}

// Largest |u - x(1-x)y(1-y)| over a block of rows: the check against the closed-form solution of the
// reference's problem (right-hand side of utils.c:334).  x = (double)i*h, y = (double)j*h, the product
// evaluated left to right; a maximum does not depend on the order it is taken in.  One value per workgroup.
__global__ __launch_bounds__(kBlock) void k_max_err(const double *__restrict__ u, long G, long row0, long rows,
                                                    double h, double *__restrict__ block_max)
{
    __shared__ double wave_slots[kWaves];
    const long total = rows * G;
    const long stride = (long)gridDim.x * kBlock;
    double m = 0.0;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < total; k += stride) {
        const long r = k / G, j = k - r * G;
        const double x = (double)(row0 + r) * h, y = (double)j * h;
        const double e = fabs(u[k] - x * (1 - x) * y * (1 - y));
        m = (e > m || e != e) ? e : m;                 // a NaN in u must surface, not vanish in the comparison
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(m, off, 64);
        m = (o > m || o != o) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) wave_slots[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWaves; ++w) m = (wave_slots[w] > m || wave_slots[w] != wave_slots[w]) ? wave_slots[w] : m;
        block_max[blockIdx.x] = m;
    }
}

// The c x c block at (r0, c0) of a row-major G-column array, copied to contiguous storage in
// its own row-wise order: the local array of one rank of the reference's s x s decomposition
// (utils.c:408-417), so that its sequential sum (utils.c:372-379) can be evaluated.
__global__ __launch_bounds__(kBlock) void k_pack_block(const double *__restrict__ src, long G, long r0, long c0,
                                                       long c, double *__restrict__ dst)
{
    const long total = c * c;
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < total; k += stride) {
        const long i = k / c, j = k - i * c;
        dst[k] = src[(r0 + i) * G + c0 + j];
    }
}

// MPI_Allreduce of one double per rank as MPICH's recursive doubling forms it (utils.c:380):
// partners at distance 1, 2, 4, ... -- a balanced binary tree over the ranks.  count is a power of two <= 64.
__global__ void k_tree_sum(const double *vals, int count, double *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double part[64];
    for (int r = 0; r < count; ++r) part[r] = vals[r];
    for (int dist = 1; dist < count; dist *= 2)
        for (int r = 0; r + dist < count; r += 2 * dist) part[r] = part[r] + part[r + dist];
    *out = part[0];
}

// ---------------------------------------------------------------- set-up -----

// d0 on global rows [row0, row0+rows): utils.c:319-337.  h is passed in from
// the host ((double)1/n, utils.c:323) so that host and device agree on it.
__global__ __launch_bounds__(kBlock) void k_init_d(double *d, long G, long row0, long rows, double h)
{
    const long total = rows * G;
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < total; k += stride) {
        const long r = k / G, j = k - r * G;
        const long gi = row0 + r;
        double val = 0.0;
        if (!(gi == 0 || gi == G - 1 || j == 0 || j == G - 1)) {
            const double xi = (double)gi * h;
            const double yj = (double)j * h;
            val = 2 * h * h * (xi * (1 - xi) + yj * (1 - yj));
        }
        d[k] = val;
    }
}

// d0 on the rows x cols block whose first point is global (row0, col0) -- one rank's chunk of the reference's
// sqrt(P) x sqrt(P) decomposition (utils.c:319-337 with carti*chunklength = row0, cartj*chunklength = col0).
__global__ __launch_bounds__(kBlock) void k_init_d_block(double *d, long G, long row0, long col0, long rows, long cols, double h)
{
    const long total = rows * cols;
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < total; k += stride) {
        const long r = k / cols, c = k - r * cols;
        const long gi = row0 + r, gj = col0 + c;
        double val = 0.0;
        if (!(gi == 0 || gi == G - 1 || gj == 0 || gj == G - 1)) {
            const double xi = (double)gi * h;
            const double yj = (double)gj * h;
            val = 2 * h * h * (xi * (1 - xi) + yj * (1 - yj));
        }
        d[k] = val;
    }
}

// q = A d on one rank's rows x cols chunk of the reference's two-dimensional decomposition, the whole of
// apply_stencil_parallel's arithmetic in one kernel: interior points (utils.c:50 -> :15-35), first / last row
// with the pads received from above / below (utils.c:74-104), first / last column with the pads from the left
// / right (utils.c:121-151) and the four corner points with two pads each (utils.c:156-218).  A null pad means
// the chunk touches the global boundary on that side, where q is not written (the reference's loops skip those
// points the same way).  Five-point sum in the reference's order: up, left, centre, right, down.  With corner
// coefficients (nine-point sum, in array order) the chunk's four corner points also need the values diagonally
// across: corner[0..3] = the points up-left, up-right, down-left, down-right of the chunk (what the reference's
// d_struct reserves corner_pad for, utils.h:15, and never fills); an entry is read only where both adjacent
// pads exist, i.e. where there is a diagonal neighbour.
struct BlockArgs {
    const double *d;                  // rows x cols, row-major
    const double *top, *bot;          // cols doubles each, or nullptr
    const double *left, *right;       // rows doubles each, or nullptr
    const double *corner;             // four doubles, or nullptr (five-point stencils, chunks without diagonal neighbours)
    double *q;
    long rows, cols;
    Coef9 coef;
};

__global__ __launch_bounds__(kBlock) void k_stencil_block(const BlockArgs a)
{
    const long total = a.rows * a.cols;
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < total; k += stride) {
        const long r = k / a.cols, c = k - r * a.cols;
        if ((r == 0 && a.top == nullptr) || (r == a.rows - 1 && a.bot == nullptr) || (c == 0 && a.left == nullptr) ||
            (c == a.cols - 1 && a.right == nullptr))
            continue;                                  // a point of the global boundary
        const double up = (r > 0) ? a.d[k - a.cols] : a.top[c];
        const double down = (r < a.rows - 1) ? a.d[k + a.cols] : a.bot[c];
        const double lf = (c > 0) ? a.d[k - 1] : a.left[r];
        const double rt = (c < a.cols - 1) ? a.d[k + 1] : a.right[r];
        if (!a.coef.nine) {
            a.q[k] = five_point(a.coef, up, lf, a.d[k], rt, down);
            continue;
        }
        const bool r_lo = r > 0, r_hi = r < a.rows - 1, c_lo = c > 0, c_hi = c < a.cols - 1;
        const double ul = r_lo ? (c_lo ? a.d[k - a.cols - 1] : a.left[r - 1]) : (c_lo ? a.top[c - 1] : a.corner[0]);
        const double ur = r_lo ? (c_hi ? a.d[k - a.cols + 1] : a.right[r - 1]) : (c_hi ? a.top[c + 1] : a.corner[1]);
        const double dl = r_hi ? (c_lo ? a.d[k + a.cols - 1] : a.left[r + 1]) : (c_lo ? a.bot[c - 1] : a.corner[2]);
        const double dr = r_hi ? (c_hi ? a.d[k + a.cols + 1] : a.right[r + 1]) : (c_hi ? a.bot[c + 1] : a.corner[3]);
        a.q[k] = nine_point(a.coef, ul, up, ur, lf, a.d[k], rt, dl, down, dr);
    }
}

// Column `col` of a rows x cols row-major array, gathered into contiguous storage: what the reference's strided
// MPI datatype (MPI_Type_vector(chunklength, 1, chunklength), utils.c:392) does on the sending side.
__global__ __launch_bounds__(kBlock) void k_pack_column(const double *__restrict__ src, long rows, long cols, long col,
                                                        double *__restrict__ dst)
{
    const long stride = (long)gridDim.x * kBlock;
    for (long r = (long)blockIdx.x * kBlock + threadIdx.x; r < rows; r += stride) dst[r] = src[r * cols + col];
}

// g0 = -d0: utils.c:342-350.
__global__ __launch_bounds__(kBlock) void k_init_g(double *g, const double *d, long count)
{
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < count; k += stride) g[k] = -d[k];
}

} // namespace cg
