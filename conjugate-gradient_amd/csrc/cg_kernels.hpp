// Device code of the MI355X (gfx950) conjugate-gradient hot path.
//
// Everything here is fp64 and is compiled with -ffp-contract=off: the reference
// rounds every product before the following add (cg.c:63-87, utils.c:27-31,
// utils.c:373-379 compiled as C99 without FMA contraction), so must we.
//
// Kernel inventory (reference lines each one replaces):
//   k_stencil          q = A d (+ optional fused d = beta*d - g, + d.q partials)
//                      utils.c:15-35, utils.c:74-104, cg.c:81-87, utils.c:373-379
//   k_update           u += tau*d ; g += tau*q ; g.g partials      cg.c:63-76,:79
//   k_update_stencil   same, with q recomputed from d instead of loaded
//   k_dupdate          d = beta*d - g                              cg.c:81-87
//   k_dot_partial      block partials of v.w                       utils.c:373-379
//   k_dot_strict       v.w with ONE accumulator in index order     utils.c:373-379
//   k_finish           fixed-order sum of block partials + the scalar recurrences
//                      tau = q0/(d.q), beta = q1/q0, q0 = q1       cg.c:62,:80,:89
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
//   5. multi-GPU: the per-rank sums are added in rank order 0..P-1.
#pragma once
#include <hip/hip_runtime.h>

namespace cg {

constexpr int kBlock = 256;           // 4 wavefronts of 64 lanes
constexpr int kWaves = kBlock / 64;

// Device-resident scalar state of one solve.  Only k_finish writes it; the other
// kernels read tau / beta / done in their prologue, so the host never has to
// look at a scalar inside the iteration loop (graph-capturable).
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
    int done;           // set by k_finish when the stop rule fires
    int pad;
};

// ---------------------------------------------------------------- reductions --

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;                         // valid in lane 0
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

struct StencilArgs {
    const double *d_in;               // rows x G block of the direction vector
    const double *g;                  // rows x G block of the gradient (fused d-update)
    const double *d_top, *d_bot;      // halo rows of d_in, nullptr at the global boundary
    const double *g_top, *g_bot;      // halo rows of g (fused d-update only)
    double *d_out;                    // updated direction (fused d-update only)
    double *d_top_out, *d_bot_out;    // updated halo rows of d (fused d-update, multi-GPU only)
    double *q;                        // A d (may be nullptr when q is recomputed later)
    double *u, *gw;                   // k_update_stencil: solution and gradient, updated in place
    double *partials;                 // one partial dot product per tile
    const Scalars *sc;
    long G;                           // grid side = row length
    long rows;                        // rows in this block
    long row0;                        // global index of the block's first row
    int nstrips;                      // column strips per row of tiles
    int tile_h;                       // rows per tile
    int remap;                        // 1 = XCD-aware tile order
};

template <int VEC> struct Pack;
template <> struct Pack<1> {
    double v[1];
    __device__ __forceinline__ void load(const double *p) { v[0] = *p; }
    __device__ __forceinline__ void store(double *p) const { *p = v[0]; }
};
template <> struct Pack<2> {
    double v[2];
    __device__ __forceinline__ void load(const double *p)
    {
        const double2 t = *reinterpret_cast<const double2 *>(p);
        v[0] = t.x; v[1] = t.y;
    }
    __device__ __forceinline__ void store(double *p) const
    {
        *reinterpret_cast<double2 *>(p) = make_double2(v[0], v[1]);
    }
};

// The five-point sum in the reference's order: up, left, centre, right, down,
// added one by one to a zero accumulator (utils.c:25-31).
__device__ __forceinline__ double five_point(double up, double left, double centre,
                                             double right, double down)
{
    double acc = 0.0;
    acc += -1.0 * up;
    acc += -1.0 * left;
    acc += 4.0 * centre;
    acc += -1.0 * right;
    acc += -1.0 * down;
    return acc;
}

// Raw operands of one grid row as a thread sees them: its VEC own columns plus,
// for the first / last thread of the strip, the column just outside the strip.
template <int VEC, bool FUSE_D>
struct RawRow {
    Pack<VEC> d, g;
    double ed, eg;                    // edge column (threads 0 and kBlock-1 only)
};

// MODE 0: q = A d, partials of d.q                      (stencil entry points)
// MODE 1: d' = beta*d - g, q = A d', partials of d'.q   (fused d-update)
// MODE 2: q = A d recomputed, u += tau*d, g += tau*q, partials of g.g
template <int VEC, int MODE, bool WRITE_Q>
__global__ __launch_bounds__(kBlock) void k_stencil(const StencilArgs a)
{
    constexpr bool FUSE_D = (MODE == 1);
    constexpr int W = kBlock * VEC;
    if (a.sc != nullptr && a.sc->done) return;

    __shared__ double row_lds[2][W + 2];
    __shared__ double wave_slots[kWaves];

    const int tid = threadIdx.x;
    const long ntiles = (long)gridDim.x;
    const long tile = a.remap ? xcd_remap(blockIdx.x, ntiles) : (long)blockIdx.x;
    const long strip = tile % a.nstrips, rt = tile / a.nstrips;
    const long G = a.G;
    const long c0 = strip * W;
    const long j = c0 + (long)tid * VEC;
    const long r0 = rt * a.tile_h;
    const long r1 = (r0 + a.tile_h < a.rows) ? r0 + a.tile_h : a.rows;
    const bool col_ok = j < G;                       // VEC == 2 only with even G: all or nothing
    const bool left_edge = (tid == 0) && (c0 > 0);
    const bool right_edge = (tid == kBlock - 1) && (c0 + W < G);
    const long je = left_edge ? c0 - 1 : c0 + W;

    double beta = 0.0, tau = 0.0;
    if (MODE == 1) beta = a.sc->beta;
    if (MODE == 2) tau = a.sc->tau;

    // Row r of the block, r in [-1, rows]; -1 and rows are the halo rows.
    auto row_ptr = [&](const double *base, const double *top, const double *bot, long r) -> const double * {
        if (r < 0) return top;
        if (r >= a.rows) return bot;
        return base + r * G;
    };
    auto load_row = [&](long r, RawRow<VEC, FUSE_D> &raw) {
        const double *pd = row_ptr(a.d_in, a.d_top, a.d_bot, r);
#pragma unroll
        for (int v = 0; v < VEC; ++v) { raw.d.v[v] = 0.0; raw.g.v[v] = 0.0; }
        raw.ed = 0.0; raw.eg = 0.0;
        if (pd == nullptr) return;                    // beyond the global boundary
        if (col_ok) raw.d.load(pd + j);
        if (left_edge || right_edge) raw.ed = pd[je];
        if (FUSE_D) {
            const double *pg = row_ptr(a.g, a.g_top, a.g_bot, r);
            if (col_ok) raw.g.load(pg + j);
            if (left_edge || right_edge) raw.eg = pg[je];
        }
    };

    double up[VEC], mid[VEC], midL = 0.0, midR = 0.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) { up[v] = 0.0; mid[v] = 0.0; }
    double acc = 0.0;

    RawRow<VEC, FUSE_D> raw_next;
    load_row(r0 - 1, raw_next);
    int buf = 0;
    for (long r = r0 - 1; r <= r1; ++r) {
        RawRow<VEC, FUSE_D> raw = raw_next;
        if (r + 1 <= r1) load_row(r + 1, raw_next);  // one row of loads in flight ahead of use

        double cur[VEC], edge;
        if (FUSE_D) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) cur[v] = beta * raw.d.v[v] - raw.g.v[v];
            edge = beta * raw.ed - raw.eg;
            if (col_ok) {
                // Owned rows go to d_out; a neighbour's boundary row (known here only
                // as a halo) is kept up to date locally, so that only g has to travel.
                double *dst = nullptr;
                if (r >= r0 && r < r1) dst = a.d_out + r * G + j;
                else if (r < 0) dst = a.d_top_out ? a.d_top_out + j : nullptr;
                else if (r >= a.rows) dst = a.d_bot_out ? a.d_bot_out + j : nullptr;
                if (dst != nullptr) {
                    Pack<VEC> o;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) o.v[v] = cur[v];
                    o.store(dst);
                }
            }
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v) cur[v] = raw.d.v[v];
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
        __syncthreads();
        const double curL = L[tid * VEC];
        const double curR = L[tid * VEC + VEC + 1];
        buf ^= 1;

        // Rows r-2 (up), r-1 (mid) and r (cur) are now in registers: emit row r-1.
        const long m = r - 1;
        if (m >= r0) {
            const long gi = a.row0 + m;
            const bool row_in = (gi >= 1) && (gi <= G - 2);
            double qv[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const double left = (v == 0) ? midL : mid[v - 1];
                const double right = (v == VEC - 1) ? midR : mid[v + 1];
                const bool in = row_in && (j + v >= 1) && (j + v <= G - 2);
                qv[v] = in ? five_point(up[v], left, mid[v], right, cur[v]) : 0.0;
            }
            if (MODE == 2) {
                if (col_ok) {
                    Pack<VEC> pu, pg;
                    pu.load(a.u + m * G + j);
                    pg.load(a.gw + m * G + j);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        pu.v[v] = pu.v[v] + tau * mid[v];
                        pg.v[v] = pg.v[v] + tau * qv[v];
                        acc += pg.v[v] * pg.v[v];
                    }
                    pu.store(a.u + m * G + j);
                    pg.store(a.gw + m * G + j);
                }
            } else {
                if (col_ok) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc += mid[v] * qv[v];
                    if (WRITE_Q && row_in) {
                        // Boundary columns of an interior row are not written by the
                        // reference either; q there keeps its initial zero.
                        if (VEC == 2 && j >= 1 && j + 1 <= G - 2) {
                            Pack<VEC> o;
#pragma unroll
                            for (int v = 0; v < VEC; ++v) o.v[v] = qv[v];
                            o.store(a.q + m * G + j);
                        } else {
#pragma unroll
                            for (int v = 0; v < VEC; ++v)
                                if (j + v >= 1 && j + v <= G - 2) a.q[m * G + j + v] = qv[v];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VEC; ++v) { up[v] = mid[v]; mid[v] = cur[v]; }
        midL = curL; midR = curR;
    }

    const double s = block_sum(acc, wave_slots);
    if (tid == 0) a.partials[tile] = s;
}

// ------------------------------------------------------------- flat kernels --

// u += tau*d ; g += tau*q ; partials of g.g           (cg.c:63-76, :79)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_update(double *__restrict__ u, double *__restrict__ g,
                                                   const double *__restrict__ d,
                                                   const double *__restrict__ q, long count,
                                                   const Scalars *sc, double *partials)
{
    if (sc->done) return;
    __shared__ double wave_slots[kWaves];
    const double tau = sc->tau;
    const long nvec = count / VEC;
    const long stride = (long)gridDim.x * kBlock;
    double acc = 0.0;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < nvec; k += stride) {
        Pack<VEC> pu, pg, pd, pq;
        pu.load(u + k * VEC); pg.load(g + k * VEC); pd.load(d + k * VEC); pq.load(q + k * VEC);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            pu.v[v] = pu.v[v] + tau * pd.v[v];
            pg.v[v] = pg.v[v] + tau * pq.v[v];
            acc += pg.v[v] * pg.v[v];
        }
        pu.store(u + k * VEC); pg.store(g + k * VEC);
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
// summation (utils.c:372-379).  One wavefront: the lanes fetch 64 consecutive
// products at a time (coalesced), then every lane replays the same 64 dependent
// additions, reading product l from lane l.  The result is bit-identical to the
// CPU loop; the price is one dependent fp64 add per element.
__global__ __launch_bounds__(64) void k_dot_strict(const double *__restrict__ v,
                                                   const double *__restrict__ w, long count,
                                                   double start, double *result)
{
    const int lane = threadIdx.x;
    double acc = start;
    long base = 0;
    for (; base + 64 <= count; base += 64) {
        const double p = v[base + lane] * w[base + lane];
#pragma unroll
        for (int l = 0; l < 64; ++l) acc += __shfl(p, l, 64);
    }
    const long rem = count - base;
    if (rem > 0) {
        const double p = (lane < rem) ? v[base + lane] * w[base + lane] : 0.0;
        for (int l = 0; l < rem; ++l) acc += __shfl(p, l, 64);
    }
    if (lane == 0) *result = acc;
}

// Fixed-order sum of `count` partials and the scalar recurrences.
//   what 0: dq   = sum ; tau  = q0/dq                                  (cg.c:61-62)
//   what 1: q1   = sum ; beta = q1/q0 ; q0 = q1 ; history ; stop rule  (cg.c:79-80,:89)
//   what 2: q0   = sum ; q_init = q0 ; stop = rtol^2*q0                (cg.c:55)
//   what 3: dot  = sum                                                 (stand-alone dot)
// With `strict_src` the sum is taken from *strict_src (k_dot_strict's result)
// instead of the partials.  In a multi-rank solve the sum over ranks has been
// formed before this kernel runs and arrives through `strict_src` as well.
__global__ __launch_bounds__(kBlock) void k_finish(const double *partials, long count,
                                                   const double *strict_src, Scalars *sc, int what,
                                                   double rtol, double *hist, long hist_cap)
{
    if (what < 2 && sc->done) return;
    __shared__ double wave_slots[kWaves];
    double acc = 0.0;
    if (strict_src == nullptr)
        for (long k = threadIdx.x; k < count; k += kBlock) acc += partials[k];
    double s = block_sum(acc, wave_slots);
    if (threadIdx.x != 0) return;
    if (strict_src != nullptr) s = *strict_src;
    if (what == 0) {
        sc->dq = s;
        sc->tau = sc->q0 / s;
    } else if (what == 1) {
        sc->q1 = s;
        sc->beta = s / sc->q0;
        sc->q0 = s;
        const long long it = sc->iter;
        if (hist != nullptr && it < hist_cap) hist[it] = s;
        sc->iter = it + 1;
        if (sc->stop > 0.0 && s <= sc->stop) sc->done = 1;
    } else if (what == 2) {
        sc->q0 = s;
        sc->q1 = s;
        sc->q_init = s;
        sc->stop = rtol * rtol * s;
        sc->iter = 0;
        sc->done = 0;
    } else {
        sc->dot = s;
    }
}

// Sum of `count` per-rank values in rank order (multi-GPU step 5 of the order).
__global__ void k_rank_sum(const double *vals, int count, double *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = vals[0];
    for (int r = 1; r < count; ++r) s += vals[r];
    *out = s;
}

// Fixed-order sum of block partials into one slot (local part of a multi-rank dot).
__global__ __launch_bounds__(kBlock) void k_partial_sum(const double *partials, long count,
                                                        const Scalars *sc, double *out)
{
    if (sc != nullptr && sc->done) return;
    __shared__ double wave_slots[kWaves];
    double acc = 0.0;
    for (long k = threadIdx.x; k < count; k += kBlock) acc += partials[k];
    const double s = block_sum(acc, wave_slots);
    if (threadIdx.x == 0) *out = s;
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

// g0 = -d0: utils.c:342-350.
__global__ __launch_bounds__(kBlock) void k_init_g(double *g, const double *d, long count)
{
    const long stride = (long)gridDim.x * kBlock;
    for (long k = (long)blockIdx.x * kBlock + threadIdx.x; k < count; k += stride) g[k] = -d[k];
}

} // namespace cg
