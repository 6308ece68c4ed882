// Bit-exact parallel evaluation of the reference's dot product.
//
// The reference adds the products to ONE accumulator in index order
// (utils.c:372-379): s <- fl(s + fl(v[k]*w[k])), k = 0..N-1.  That recurrence is
// not associative, and replaying it literally costs one dependent fp64 add per
// element (k_dot_strict: ~1 s for 2.7e8 elements).  This file computes the very
// same bits in parallel.
//
// Idea.  While the accumulator stays inside one binade [2^e, 2^(e+1)) its ulp is
// the constant u = 2^(e-52), and with S = s/u (an integer in [2^52, 2^53)) and
// x = p/u (exact: a power-of-two scaling)
//        fl(s + p) / u  =  RN(S + x)  =  S + RN(x)        unless x is a tie (x = m + 1/2),
// i.e. inside a binade the recurrence adds INTEGERS, which is associative.  So:
//   A. approximate sums of chunks of 1024 consecutive products (any order), and of
//      groups of 64 chunks; an exclusive scan of the group sums;
//   B. per chunk: from those, a guess e of the accumulator's exponent when the
//      recurrence reaches the chunk; then, under the hypothesis "the accumulator stays
//      in binade e", the integer sum D of RN(x_k) and the min / max of its running
//      value (to verify the hypothesis later); ties, huge addends and NaNs mark the
//      chunk unusable;
//   C. groups of 64 chunks with one common guess are composed in order
//      ((D,mn,mx) o (D',mn',mx') = (D+D', min(mn, D+mn'), max(mx, D+mx')));
//      (and groups of 64 groups once more);
//   D. ONE wavefront then walks the top level in order with the true accumulator s:
//      it scans 64 summaries at a time, accepts the longest prefix whose hypothesis
//      holds for the actual s (right exponent, running value never leaves
//      [2^52+1, 2^53-1], no flag), and where that fails descends to the chunks of the
//      group, then to 16-element pieces of the failing chunk (summarised on the spot
//      under the accumulator's actual exponent), and finally to the literal
//      element-by-element recurrence for the one piece that does not fit.
//   E. that wavefront spends its time waiting unless everything it will need is laid out for it
//      beforehand.  The approximate prefix sums also say WHERE the walk will have to descend (where
//      the accumulator changes its binade), so the marked chunks are listed in order, their
//      products and the summaries of their 16-element pieces (under the three binades around the
//      predicted one, in prefix form) are staged, and each stretch between two marked chunks is
//      composed into ONE summary.  For each marked chunk the plan goes one step further (the
//      brief, kBrief): the approximation is good enough to tell which ONE piece the crossing falls
//      into, so the pieces before it and behind it are composed into two summaries and that
//      piece's sixteen products are copied next to them.  The walk applies a stretch in one step,
//      then the chunk by its brief -- a range test, sixteen additions, a range test, all on
//      wave-uniform values read from LDS -- and takes the chunk apart with the staged products and
//      piece summaries only where a test fails (k_xs_plan, k_xs_walk).
//      A plan that turns out wrong costs time, never a bit: every summary is judged by the true
//      accumulator, and the walk falls back to the hierarchical descent of D.
// Failures are rare: each binade crossing of the accumulator (a few dozen per dot
// product), the leading zeros, and ties (~2^-29 per element).  The margins of one ulp
// at both ends of the binade make the integer model exact: a running value of at
// least 2^52+1 means the unrounded sum was at least 2^52+1/2, still inside the binade.
// Every step is either exact integer arithmetic in fp64 (|values| < 2^53) or the
// literal recurrence, so the result is bit-identical to the CPU loop for ANY input.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <initializer_list>
#include <vector>

#include "cg_kernels.hpp"

namespace cg {
namespace exact {

constexpr int kChunk = 1024;          // elements per chunk = one wavefront x 16
constexpr int kPerLane = kChunk / 64;
constexpr int kGroup = 64;            // chunks per group = one wavefront of summaries
constexpr int kAnyExp = INT_MAX;      // summary valid under every hypothesis (all addends are zero)
constexpr int kNoExp = INT_MIN;       // no usable guess

typedef double xv2 __attribute__((ext_vector_type(2)));

// The scalar step of the solver to run right behind a walk, in the same kernel (sc == nullptr: none): what k_finish would do
// one launch later.  partials2 / count2: q.q partial sums to add up on the way (any order: prediction input), null or < 0 as there.
struct FinishArgs {
    Scalars *sc = nullptr;
    int what = 0;
    double rtol = 0.0;
    double *hist = nullptr;
    long hist_cap = 0;
    const double *partials2 = nullptr;
    long count2 = 0;
};

struct Summary {
    double D;         // sum of the rounded addends, in ulps of the hypothesised binade
    double mn, mx;    // extremes of the running value of D (the empty prefix, 0, included)
    int e;            // hypothesised exponent of the accumulator, or kAnyExp
    int bad;          // bit 0: the summary must not be used; bits 1..: for a chunk summary, 1 + the slot of the
                      // product dump that holds the chunk's products (0 = none)
};

__device__ __forceinline__ Summary identity()
{
    Summary s; s.D = 0.0; s.mn = 0.0; s.mx = 0.0; s.e = kAnyExp; s.bad = 0;
    return s;
}

// a happens before b
__device__ __forceinline__ Summary compose(const Summary &a, const Summary &b)
{
    Summary r;
    r.D = a.D + b.D;
    r.mn = fmin(a.mn, a.D + b.mn);
    r.mx = fmax(a.mx, a.D + b.mx);
    r.bad = ((a.bad | b.bad) & 1) | ((a.e != kAnyExp && b.e != kAnyExp && a.e != b.e) ? 1 : 0);
    r.e = (a.e != kAnyExp) ? a.e : b.e;
    return r;
}

__device__ __forceinline__ Summary shfl_down(const Summary &s, int off)
{
    Summary r;
    r.D = __shfl_down(s.D, off, 64); r.mn = __shfl_down(s.mn, off, 64); r.mx = __shfl_down(s.mx, off, 64);
    r.e = __shfl_down(s.e, off, 64); r.bad = __shfl_down(s.bad, off, 64);
    return r;
}
__device__ __forceinline__ Summary shfl_up(const Summary &s, int off)
{
    Summary r;
    r.D = __shfl_up(s.D, off, 64); r.mn = __shfl_up(s.mn, off, 64); r.mx = __shfl_up(s.mx, off, 64);
    r.e = __shfl_up(s.e, off, 64); r.bad = __shfl_up(s.bad, off, 64);
    return r;
}

// Ordered reduction over the wavefront, lane 0 first; result valid in lane 0.
__device__ __forceinline__ Summary wave_compose(Summary cur)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const Summary nxt = shfl_down(cur, off);
        if ((lane & (2 * off - 1)) == 0) cur = compose(cur, nxt);
    }
    return cur;
}

// Exponent of a finite double of magnitude at least 2^-1021, else kNoExp (zero, subnormal, the lowest normal
// binade, infinities, NaNs): the biased exponent field, 2 .. 0x7fe.
__device__ __forceinline__ int exponent_of(double s)
{
    const int be = (__double2hiint(s) >> 20) & 0x7ff;
    return (be >= 2 && be <= 0x7fe) ? be - 1023 : kNoExp;
}

// Summary of kPerLane consecutive addends under the hypothesis "exponent e".
// The per-element validity tests are folded into two running maxima and checked once:
// |x| < 2^51 for every addend <=> max|x| < 2^51; no tie <=> max|x - rint(x)| != 0.5 (it never
// exceeds 0.5); a NaN addend poisons D (v_max would skip it), which the final D == D catches.
// any_size: accept addends of any magnitude (the self-check recomputes summaries that were certified another way,
// by the sum of the magnitudes staying below 2^52).
__device__ __forceinline__ Summary piece_summary(const double (&p)[kPerLane], int e, bool any_size = false)
{
    const int sh = (e == kNoExp) ? 0 : 52 - e;
    Summary s = identity();
    double amax = 0.0, fmaxv = 0.0, pmax = 0.0;
#pragma unroll
    for (int i = 0; i < kPerLane; ++i) {
        const double x = ldexp(p[i], sh);
        const double r = rint(x);
        pmax = fmax(pmax, fabs(p[i]));
        amax = fmax(amax, fabs(x));
        fmaxv = fmax(fmaxv, fabs(x - r));
        s.D += r;
        s.mn = fmin(s.mn, s.D);
        s.mx = fmax(s.mx, s.D);
    }
    const bool nan = !(s.D == s.D) || !(pmax == pmax);
    if (!nan && pmax == 0.0) return identity();     // adding zeros leaves a non-zero accumulator unchanged
    s.bad = (nan || !(any_size || amax < 0x1p51) || fmaxv == 0.5 || e == kNoExp) ? 1 : 0;
    s.e = e;
    return s;
}

// Ordered composition over the wavefront of summaries that share ONE hypothesis e (or are
// identities): only D, mn, mx travel; flags and the exponent are combined by ballots.
__device__ __forceinline__ Summary wave_compose_same_e(Summary cur, int e)
{
    const int lane = threadIdx.x & 63;
    const bool any_bad = __ballot((cur.bad & 1) != 0) != 0ull;
    const bool any_real = __ballot(cur.e != kAnyExp) != 0ull;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double nD = __shfl_down(cur.D, off, 64), nmn = __shfl_down(cur.mn, off, 64),
                     nmx = __shfl_down(cur.mx, off, 64);
        if ((lane & (2 * off - 1)) == 0) {
            cur.mn = fmin(cur.mn, cur.D + nmn);
            cur.mx = fmax(cur.mx, cur.D + nmx);
            cur.D = cur.D + nD;
        }
    }
    cur.bad = any_bad ? 1 : 0;
    cur.e = any_real ? e : kAnyExp;
    return cur;
}

// Product k of the sequence that is summed; w == nullptr means v already holds the rounded
// products (the producing kernel formed fl(v*w) with the same single rounding).
__device__ __forceinline__ double product(const double *v, const double *w, long k)
{
    return (w != nullptr) ? v[k] * w[k] : v[k];
}

// Two consecutive products starting at even index i (zero beyond `count`).
__device__ __forceinline__ void product_pair(const double *__restrict__ v, const double *__restrict__ w, long i,
                                             long count, bool vec, double &p0, double &p1)
{
    if (w == nullptr) {
        if (vec && i + 1 < count) {
            const xv2 a = __builtin_nontemporal_load(reinterpret_cast<const xv2 *>(v + i));
            p0 = a.x; p1 = a.y;
        } else {
            p0 = (i < count) ? v[i] : 0.0;
            p1 = (i + 1 < count) ? v[i + 1] : 0.0;
        }
    } else if (vec && i + 1 < count) {
        const xv2 a = __builtin_nontemporal_load(reinterpret_cast<const xv2 *>(v + i));
        const xv2 b = __builtin_nontemporal_load(reinterpret_cast<const xv2 *>(w + i));
        p0 = a.x * b.x; p1 = a.y * b.y;
    } else {
        p0 = (i < count) ? v[i] * w[i] : 0.0;
        p1 = (i + 1 < count) ? v[i + 1] * w[i + 1] : 0.0;
    }
}

// The 16 products a lane fetches of one chunk (lane-interleaved: pair k of lane l starts at
// base + k*128 + 2*l).  A chunk that lies wholly inside the arrays, with 16-byte aligned
// operands, takes the straight-line path: eight (or sixteen) independent vector loads in
// flight at once.  Everything else -- the last, partial chunk; unaligned arrays -- goes
// through the guarded product_pair.  The choice is uniform over the wavefront.
template <bool HAS_W>
__device__ __forceinline__ void fetch_chunk(const double *__restrict__ v, const double *__restrict__ w, long base,
                                            long count, bool vec, int lane, double (&p)[kPerLane])
{
    if (vec && base + kChunk <= count) {
        xv2 a[kPerLane / 2], b[kPerLane / 2];
#pragma unroll
        for (int k = 0; k < kPerLane / 2; ++k) {
            a[k] = __builtin_nontemporal_load(reinterpret_cast<const xv2 *>(v + base + k * 128 + 2 * lane));
            if (HAS_W) b[k] = __builtin_nontemporal_load(reinterpret_cast<const xv2 *>(w + base + k * 128 + 2 * lane));
        }
#pragma unroll
        for (int k = 0; k < kPerLane / 2; ++k) {
            p[2 * k] = HAS_W ? a[k].x * b[k].x : a[k].x;
            p[2 * k + 1] = HAS_W ? a[k].y * b[k].y : a[k].y;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPerLane / 2; ++k)
            product_pair(v, HAS_W ? w : nullptr, base + k * 128 + 2 * lane, count, vec, p[2 * k], p[2 * k + 1]);
    }
}

// ---- where the products come from ---------------------------------------------------------------
// The ordered walk (D) and the repair of chunks whose speculation missed need single products and whole
// chunks of them.  Two sources: arrays (v*w, or v alone when it already holds the rounded products), and
// the lean schedule's first pass, which stores nothing: there product k is d'[k] * (A d')[k] with
// d' = beta*d - g, recomputed from d and g with the arithmetic of the tiled kernel (cg_kernels.hpp,
// k_stencil MODE 3: rounded multiply, then subtract; the stencil sum in the reference's order; q = 0 off
// the interior, so the product there is d' * 0).
struct SrcArrays {
    const double *v, *w;
    long count;
    int vec;                                            // both arrays 16-byte aligned
    static constexpr int kLds = 2;                      // doubles of LDS a wavefront needs for chunk16 (none)
    __device__ __forceinline__ void prepare() {}
    __device__ __forceinline__ double product(long k) const { return (w != nullptr) ? v[k] * w[k] : v[k]; }
    // The products of elements lo + 16*lane + i, i = 0..15 (zero from `hi` on).
    __device__ __forceinline__ void chunk16(long lo, long hi, int lane, double *, double (&p)[kPerLane]) const
    {
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            const long idx = lo + (long)lane * kPerLane + i;
            p[i] = (idx < hi) ? product(idx) : 0.0;
        }
    }
};

struct SrcStencil {
    const double *d, *g;                                // row r of either at base + r*G, r = -1 .. rows (halo rows in place)
    const Scalars *sc;
    Coef9 coef;
    long G, rows, row0, count;                          // G a multiple of kChunk: a chunk lies inside one row
    double beta;
    int plain;                                          // d holds d' itself (the iteration that starts a solve): beta and g stay out of it
    // d' of one row segment with its two outer columns.  Every lane reads 16 consecutive columns, so column x sits
    // at x + 2 + floor(x / 16): the lanes' blocks start 17 doubles apart and fall on different LDS banks (a stride
    // of 16 doubles would put 32 lanes on the same bank).
    static constexpr int kRow = kChunk + kChunk / 16 + 8;
    static constexpr int kLds = 3 * kRow;
    static __device__ __forceinline__ int at(int x) { return x + 2 + (x >> 4); }      // x = -1 .. kChunk
    __device__ __forceinline__ void prepare() { beta = sc->beta; }
    __device__ __forceinline__ double dp(long k) const { return plain ? d[k] : beta * d[k] - g[k]; }
    __device__ __forceinline__ double product(long k) const
    {
        const long m = k / G, j = k - m * G, gi = row0 + m;
        const double c = dp(k);
        if (gi < 1 || gi > G - 2 || j < 1 || j > G - 2) return c * 0.0;
        double q;
        if (coef.nine)
            q = nine_point(coef, dp(k - G - 1), dp(k - G), dp(k - G + 1), dp(k - 1), c, dp(k + 1), dp(k + G - 1), dp(k + G),
                           dp(k + G + 1));
        else
            q = five_point(coef, dp(k - G), dp(k - 1), c, dp(k + 1), dp(k + G));
        return c * q;
    }
    // The chunk that starts at element lo (a multiple of kChunk): the wavefront fetches the three rows of d and g
    // it needs with coalesced 16-byte loads, all in flight together, forms d' into LDS (T: kLds doubles of this
    // wavefront), and every lane then evaluates its 16 consecutive products from there.
    __device__ __forceinline__ void chunk16(long lo, long hi, int lane, double *T, double (&p)[kPerLane]) const
    {
        const long m = lo / G, j0 = lo - m * G, gi = row0 + m;
        xv2 dv[3][kChunk / 128], gv[3][kChunk / 128];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const long base = (m + rr - 1) * G + j0;
#pragma unroll
            for (int k = 0; k < kChunk / 128; ++k) {
                dv[rr][k] = *reinterpret_cast<const xv2 *>(d + base + k * 128 + 2 * lane);
                gv[rr][k] = *reinterpret_cast<const xv2 *>(g + base + k * 128 + 2 * lane);
            }
        }
        double edge = 0.0;                              // lanes 0..2: column j0-1 of the three rows; lanes 3..5: column j0+kChunk
        if (lane < 6) {
            const int rr = lane % 3;
            const long col = (lane < 3) ? j0 - 1 : j0 + kChunk;
            if (col >= 0 && col < G) edge = dp((m + rr - 1) * G + col);
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int k = 0; k < kChunk / 128; ++k) {
                double *dst = T + rr * kRow + at(k * 128 + 2 * lane);        // an even column and its neighbour: one block of 16
                dst[0] = plain ? dv[rr][k].x : beta * dv[rr][k].x - gv[rr][k].x;
                dst[1] = plain ? dv[rr][k].y : beta * dv[rr][k].y - gv[rr][k].y;
            }
        if (lane < 6) T[(lane % 3) * kRow + ((lane < 3) ? at(-1) : at(kChunk))] = edge;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool row_in = (gi >= 1) && (gi <= G - 2);
        static_assert(kPerLane == 16, "the LDS layout pads after every 16 columns");
        const double *U = T + at(lane * kPerLane), *M = U + kRow, *D = M + kRow;       // this lane's block: columns c .. c + 15 at [0 .. 15]
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            const int c = lane * kPerLane + i;
            const long j = j0 + c;
            const bool in = row_in && (j >= 1) && (j <= G - 2);
            const int l = (i == 0) ? -2 : i - 1, r = (i == kPerLane - 1) ? kPerLane + 1 : i + 1;   // the neighbours across the padding
            double q;
            if (coef.nine) q = nine_point(coef, U[l], U[i], U[r], M[l], M[i], M[r], D[l], D[i], D[r]);
            else q = five_point(coef, U[i], M[l], M[i], M[r], D[i]);
            p[i] = (lo + c < hi) ? M[i] * (in ? q : 0.0) : 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // all reads done before the buffer's next use
        __builtin_amdgcn_wave_barrier();
    }
};

// ---- A: approximate chunk sums ------------------------------------------------
// `vec` = both arrays are 16-byte aligned (chunk bases are even, so pairs are aligned too).
__global__ __launch_bounds__(256) void k_xs_approx(const double *__restrict__ v, const double *__restrict__ w,
                                                   long count, long nchunks, int vec, double *__restrict__ approx)
{
    const int lane = threadIdx.x & 63;
    const long c = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    double p[kPerLane];
    if (w != nullptr) fetch_chunk<true>(v, w, c * kChunk, count, vec != 0, lane, p);
    else fetch_chunk<false>(v, w, c * kChunk, count, vec != 0, lane, p);
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < kPerLane; ++i) acc += p[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) approx[c] = acc;
}

// A': the kernels that produce the operands can hand over approximate sums of every aligned
// span of kSub consecutive products (one wavefront's worth of a row); eight of them make a
// chunk sum, and pass A need not read the operands at all.
constexpr int kSub = 128;
// Two layouts of those span sums.  Linear: span i covers products [i*kSub, (i+1)*kSub).  By rows
// (G > 0: the operands are `rows` rows of G products, G a multiple of kSub): the span that starts
// at row m, column j sits at sub[(j/kSub)*rows + m], so that a wavefront walking down a column
// strip writes consecutive doubles (scattered 8-byte stores, one per row a kilobyte apart, cost
// the tiled kernels 20-35 % of their time).
// gsum != nullptr: the pass over the groups (k_xs_gsum) is done here as well -- thread t of a workgroup holds chunk
// 256 b + t, so each of its four wavefronts holds one whole group.
// span, stride: the sums are of `span` products each (kSub, or cg::kUnit for the records of the speculative
// path, by rows only) and lie `stride` doubles apart (4: the first member of each record).
__global__ __launch_bounds__(256) void k_xs_from_sub(const double *__restrict__ sub, long nsub, long nchunks,
                                                     long G, long rows, double *__restrict__ approx,
                                                     double *__restrict__ gsum, int span = kSub, int stride = 1,
                                                     const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    if (c < nchunks) {
        const int per = kChunk / span;
        for (int j = 0; j < per; ++j) {
            const long i = c * per + j;
            if (i >= nsub) break;
            if (G > 0) {
                const long k = i * span, m = k / G, col = k - m * G;
                acc += sub[((col / span) * rows + m) * stride];
            } else {
                acc += sub[i];
            }
        }
    }
    if (gsum == nullptr) {
        if (c < nchunks) approx[c] = acc;
        return;
    }
    const int lane = threadIdx.x & 63;
    double inc = acc;                                // chunks beyond the end count zero
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double up = __shfl_up(inc, off, 64);
        if (lane >= off) inc += up;
    }
    if (c < nchunks) approx[c] = inc - acc;
    if (lane == 63 && (c >> 6) * kGroup < nchunks) gsum[c >> 6] = inc;
}

// Sum of each group's 64 chunk sums; the chunk sums themselves are replaced by their exclusive
// prefix inside the group (what precedes the chunk in its group).
__global__ __launch_bounds__(256) void k_xs_gsum(double *__restrict__ approx, long nchunks,
                                                 double *__restrict__ gsum, long ngroups)
{
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= ngroups) return;
    const long i = g * kGroup + lane;
    const double mine = (i < nchunks) ? approx[i] : 0.0;
    double inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double up = __shfl_up(inc, off, 64);
        if (lane >= off) inc += up;
    }
    if (i < nchunks) approx[i] = inc - mine;
    if (lane == 63) gsum[g] = inc;
}

// In-place exclusive scan of the group sums, seeded with `start` (or, when `start_dev` is
// given, with the sum of the nstart device-resident values it points to).  One workgroup.
__global__ __launch_bounds__(1024) void k_xs_gscan(double *__restrict__ gsum, long ngroups, double start,
                                                   const double *start_dev, int nstart, int *zero_me = nullptr,
                                                   const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    if (zero_me != nullptr && threadIdx.x == 0) *zero_me = 0;      // the repair list of the speculative path starts empty
    if (start_dev != nullptr) {
        start = 0.0;
        for (int k = 0; k < nstart; ++k) start += start_dev[k];
    }
    __shared__ double wave_tot[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long per = (ngroups + 1023) / 1024;
    const long lo = (long)t * per, hi = (lo + per < ngroups) ? lo + per : ngroups;
    double mine = 0.0;
    for (long g = lo; g < hi; ++g) mine += gsum[g];
    double inc = mine;                               // inclusive scan inside the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double up = __shfl_up(inc, off, 64);
        if (lane >= off) inc += up;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    double before = start;
    for (int k = 0; k < wv; ++k) before += wave_tot[k];
    double run = before + (inc - mine);
    for (long g = lo; g < hi; ++g) {
        const double x = gsum[g];
        gsum[g] = run;
        run += x;
    }
}

// Sum of the (not yet scanned) group sums: this block's approximate total, for the ranks that come after it.
__global__ __launch_bounds__(1024) void k_xs_total(const double *__restrict__ gsum, long ngroups, double *__restrict__ out,
                                                   const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    __shared__ double wave_tot[16];
    double mine = 0.0;
    for (long g = threadIdx.x; g < ngroups; g += 1024) mine += gsum[g];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += wave_tot[k];
        *out = t;
    }
}

// ---- B: per-chunk integer summaries under the guessed exponent --------------------
// Persistent wavefronts: each walks its own run of chunks and keeps the NEXT chunk's products in
// flight while it summarises the current one.
// Summary of one chunk whose products `cur` a wavefront holds lane-interleaved (as fetch_chunk delivers them),
// under hypothesis e: the lanes trade them through LDS (T: 64 x 18 doubles of this wavefront) for 16
// CONSECUTIVE elements each; 18-double rows keep the 16-byte reads of the 16 lanes of a group on distinct banks.
__device__ __forceinline__ Summary chunk_summary(const double (&cur)[kPerLane], int e, double *T, int lane)
{
#pragma unroll
    for (int k = 0; k < kPerLane / 2; ++k) {
        const int pos = k * 128 + 2 * lane;
        double *dst = T + (pos >> 4) * 18 + (pos & 15);
        dst[0] = cur[2 * k]; dst[1] = cur[2 * k + 1];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double p[kPerLane];
#pragma unroll
    for (int i = 0; i < kPerLane; ++i) p[i] = T[lane * 18 + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // all reads done before the next round's writes
    __builtin_amdgcn_wave_barrier();
    return wave_compose_same_e(piece_summary(p, e), e);
}

template <bool HAS_W>
__global__ __launch_bounds__(256) void k_xs_chunk(const double *__restrict__ v, const double *__restrict__ w,
                                                  long count, long nchunks, int vec, const double *__restrict__ approx,
                                                  const double *__restrict__ goff, Summary *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) double tile[4][64 * 18];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Each wavefront owns a run of consecutive chunks: the chip's reads are then spread over as
    // many distant regions as there are wavefronts, which this memory system serves faster than one
    // dense moving window (same effect as in the tiled solver kernels).
    const long nwaves = (long)gridDim.x * 4, wid = (long)blockIdx.x * 4 + wv;
    const long per = (nchunks + nwaves - 1) / nwaves;
    long c = wid * per;
    const long cend = (c + per < nchunks) ? c + per : nchunks;
    double *T = tile[wv];
    if (c >= cend) return;                          // whole wavefront leaves together

    double nxt[kPerLane];
    auto fetch = [&](long chunk) { fetch_chunk<HAS_W>(v, w, chunk * kChunk, count, vec != 0, lane, nxt); };
    fetch(c);
    // guess of the accumulator at a chunk's start: scanned group offset + what precedes the chunk in its group
    double guess_nxt = goff[c >> 6] + approx[c];
    for (; c < cend; ++c) {
        double cur[kPerLane];
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) cur[i] = nxt[i];
        const int e = exponent_of(guess_nxt);
        if (c + 1 < cend) {                                   // in flight during the work below
            fetch(c + 1);
            guess_nxt = goff[(c + 1) >> 6] + approx[c + 1];
        }

        const Summary s = chunk_summary(cur, e, T, lane);
        if (lane == 0) out[c] = s;
    }
}

// ---- B': chunk summaries from the speculative unit sums of the producing kernel --------------------
// The tiled kernels of the lean schedule (cg_kernels.hpp, NH = 2) summarise every unit of cg::kUnit products (one
// workgroup's part of a row) under two hypothesised exponents e0, e0+1 while they produce them, so that the products
// need neither be stored nor read again.  Here the real guess -- the approximate prefix sum at the unit's start, known
// only now -- picks the hypothesis that matches it; a chunk with a unit for which neither does is put on
// the repair list and summarised from recomputed products by k_sp_fix.  The unit sums carry no running
// extremes; -N and +P (the sums of the negative and of the positive rounded products) bound the running
// value from below and above whatever the order, which only makes the walk descend a little more often
// next to a binade boundary.  The same pass writes the next iteration's hypotheses: this iteration's prefix sum at
// the unit.
// One wavefront per 64 rows of one chunk column (G is a multiple of kChunk here): lane = row, so every
// access to the by-rows unit arrays is a run of consecutive records.
struct UnitRec { double sub, D0, D1, Q0; };
__global__ __launch_bounds__(256) void k_sp_select(const UnitRec *__restrict__ rec, int nonneg,
                                                   double *__restrict__ hyp, double rho, const double *rho_dev, long G,
                                                   long rows, const double *__restrict__ approx,
                                                   const double *__restrict__ goff, Summary *__restrict__ chunks,
                                                   int *__restrict__ fixlist, int *__restrict__ nfix, const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    constexpr int kUnits = kChunk / kUnit;
    if (rho_dev != nullptr) rho = *rho_dev;                     // the prediction the producing kernel scaled the hypotheses by
    const int lane = threadIdx.x & 63;
    const long ccols = G / kChunk;                              // chunks per row
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // wavefront -> (row block, chunk column)
    const long mb = w / ccols, cc = w - mb * ccols;
    const long m = mb * 64 + lane;
    if (m >= rows) return;
    const long c = m * ccols + cc;
    double guess = goff[c >> 6] + approx[c];
    const double guess0 = guess;
    Summary cs = identity();
    bool miss = false;
#pragma unroll
    for (int j = 0; j < kUnits; ++j) {
        const long idx = (cc * kUnits + j) * rows + m;
        const xv2 ra = reinterpret_cast<const xv2 *>(rec + idx)[0], rb = reinterpret_cast<const xv2 *>(rec + idx)[1];
        const double usub = ra.x, D0 = ra.y, D1 = rb.x, Q0 = nonneg ? D0 : rb.y;
        const int e = exponent_of(guess);
        const int h = hyp_exponent(hyp[idx], rho);              // what the producing kernel worked with
        const int sel = (e == kNoExp) ? -1 : e - h;
        // D under the matching hypothesis; Q bounds the sum of the magnitudes under it (e0: the kernel's bound,
        // or D itself where no product is negative; e0 + 1: each magnitude at most half as large plus 3/4)
        double D, Q;
        if (sel == 0) { D = D0; Q = Q0; }
        else if (sel == 1) { D = D1; Q = nonneg ? D1 : 0.5 * Q0 + 0.75 * kUnit; }
        else { D = 0.0; Q = 0.0; miss = true; }
        Summary sp;
        sp.D = D;
        sp.mx = nonneg ? D : Q;                                 // the running value never exceeds the positive part ...
        sp.mn = nonneg ? 0.0 : -Q;                              // ... nor falls below minus the negative part
        sp.e = e;
        // exactness of the kernel's integer sums: the magnitudes under e0 stay below 2^52 in total
        sp.bad = (!(Q0 < 0x1p52) || D != D) ? 1 : 0;            // D is a NaN where the unit held a half-way case
        cs = compose(cs, sp);
        hyp[idx] = guess;                                       // next iteration's hypotheses start from this prefix sum
        guess += usub;
    }
    // The bounds above ignore the order of the products inside a unit.  Where products of both signs nearly cancel
    // they are far wider than the true extremes of the running value, and every chunk within that distance of a
    // binade boundary would send the walk down to the elements.  Judged by the guess (good to a part in 1e12; the
    // margin is a part in 1e8 of the binade), a chunk that does not clearly stay inside its binade is summarised
    // once more from recomputed products, with the true extremes, by k_sp_fix; the walk still verifies everything.
    bool unclear = false;
    if (!miss && !cs.bad && cs.e != kAnyExp) {
        const int es = exponent_of(guess0);
        const double S = (es == kNoExp) ? 0.0 : fabs(ldexp(guess0, 52 - es));
        const double lo = (guess0 > 0.0) ? S + cs.mn : S - cs.mx, hi = (guess0 > 0.0) ? S + cs.mx : S - cs.mn;
        unclear = (es != cs.e) || !(lo >= 0x1p52 + 0x1p26) || !(hi <= 0x1p53 - 0x1p26);
    }
    if (miss) cs.bad = 1;
    if (miss || unclear) fixlist[atomicAdd(nfix, 1)] = (int)c;
    chunks[c] = cs;
}

// First hypotheses for the kernels above, from a dot product that was evaluated the other way (products stored, span
// sums of kSub by rows in `sub`, approx / goff as k_xs_from_sub and k_xs_gscan left them): the approximate prefix sum
// at every unit's start, where k_sp_select would have put it.  One thread per unit, consecutive threads down a column.
__global__ __launch_bounds__(256) void k_sp_seed(const double *__restrict__ sub, long G, long rows,
                                                 const double *__restrict__ approx, const double *__restrict__ goff,
                                                 double *__restrict__ hyp, const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;
    constexpr int kUnits = kChunk / kUnit, kSpans = kUnit / kSub;
    const long ucols = G / kUnit, ccols = G / kChunk;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ucols * rows) return;
    const long j = idx / rows, m = idx - j * rows;
    const long cc = j / kUnits, c = m * ccols + cc;
    double guess = goff[c >> 6] + approx[c];
    for (long k = cc * kUnits * kSpans; k < j * kSpans; ++k) guess += sub[k * rows + m];    // the chunk's units before this one
    hyp[idx] = guess;
}

// What the hypotheses still lack after k_sp_seed is the growth of the total to the next iteration, and the scalar
// recurrences cannot predict it behind the very first iteration (the step length changes by orders of magnitude there).
// The SHAPE of the prefix sums hardly changes, though, so the ratio of the two iterations' sums over a band of rows
// stands in for the ratio of the totals: k_band_sum adds up this iteration's span sums over rows [m0, m1) (one
// workgroup), the next iteration evaluates the same rows ahead of its pass and divides (k_band_rho), and leaves its own
// sum for the iteration behind it.
__global__ __launch_bounds__(256) void k_band_sum(const double *__restrict__ sub, long G, long rows, long m0, long m1,
                                                  double *__restrict__ out, const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;
    __shared__ double wave_tot[4];
    const long nspans = G / kSub, width = m1 - m0;
    double acc = 0.0;
    for (long k = threadIdx.x; k < nspans * width; k += 256) acc += sub[(k / width) * rows + m0 + k % width];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) wave_tot[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (wave_tot[0] + wave_tot[1]) + (wave_tot[2] + wave_tot[3]);
}

__global__ __launch_bounds__(64) void k_band_rho(const double *__restrict__ partials, long count, double *__restrict__ before,
                                                 Scalars *sc, const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;
    double acc = 0.0;
    for (long k = threadIdx.x; k < count; k += 64) acc += partials[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (threadIdx.x == 0) {
        sc->rho[0] = sane_ratio(acc / *before);
        *before = acc;                                   // the next iteration's reference
    }
}

// Chunks on the repair list: their products are recomputed from the source and summarised under the guess at
// the chunk's start, exactly as k_xs_chunk does for stored products.  Persistent wavefronts over the list.
template <class Src>
__global__ __launch_bounds__(256) void k_sp_fix(const Src src_in, long count, const int *__restrict__ fixlist,
                                                const int *__restrict__ nfix, const double *__restrict__ approx,
                                                const double *__restrict__ goff, Summary *__restrict__ chunks,
                                                const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    __shared__ __attribute__((aligned(16))) double T[4][Src::kLds];
    Src src = src_in;
    src.prepare();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long nwaves = (long)gridDim.x * 4, wid = (long)blockIdx.x * 4 + wv;
    const long n = *nfix;
    for (long i = wid; i < n; i += nwaves) {
        const long c = fixlist[i];
        const long lo = c * kChunk, hi = (lo + kChunk < count) ? lo + kChunk : count;
        double p[kPerLane];
        src.chunk16(lo, hi, lane, T[wv], p);
        const int e = exponent_of(goff[c >> 6] + approx[c]);
        const Summary s = wave_compose_same_e(piece_summary(p, e), e);
        if (lane == 0) chunks[c] = s;
    }
}

// ---- C: groups of 64 chunk summaries -------------------------------------------------
// The walk (D) is one wavefront, and what it spends its time on is waiting: for the chunk summaries of a group
// it has to descend into, for the operands of a chunk it has to take apart.  Which groups and chunks those will
// be is known beforehand almost surely -- the approximate prefix sums that steered the hypotheses also say
// where the accumulator changes its binade -- so the kernels that produce the summaries mark them, the plan
// (k_xs_plan) lists the marked chunks in order and stages their products, and the walk fetches one entry ahead.
// A wrong mark costs time only: the walk judges every summary by the true accumulator.
constexpr int kPlanCap = 1024;                  // chunks whose products are staged for the walk (8 KiB each)
constexpr double kPlanMargin = 0x1p26;          // ulps; the approximate prefix sums are far more accurate than that
constexpr int kPlanBlocks = 64;                 // workgroups that stage them
constexpr int kMarkBit = 2;                     // Summary::bad of a group summary: the group is expected not to apply

// Will the walk, arriving with an accumulator close to `guess`, probably reject this summary?
__device__ __forceinline__ bool likely_fails(const Summary &c, double guess)
{
    if (c.bad & 1) return true;
    if (c.e == kAnyExp) return false;
    const int es = exponent_of(guess);
    if (es == kNoExp || c.e != es) return true;
    const double S = fabs(ldexp(guess, 52 - es));
    const double lo = (guess > 0.0) ? S + c.mn : S - c.mx, hi = (guess > 0.0) ? S + c.mx : S - c.mn;
    return !(lo >= 0x1p52 + kPlanMargin) || !(hi <= 0x1p53 - kPlanMargin);
}

// marks (may be null): per group, bit c set = chunk c of the group is expected to be taken apart; the group's own
// expectation goes into bit kMarkBit of its summary's `bad`.  approx / goff: exclusive prefix of the approximate
// chunk sums inside the group / approximate accumulator at the group's start.
__global__ __launch_bounds__(256) void k_xs_group(const Summary *__restrict__ in, long nin, Summary *__restrict__ out,
                                                  long nout, const double *__restrict__ approx,
                                                  const double *__restrict__ goff, unsigned long long *__restrict__ marks,
                                                  const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= nout) return;
    const long i = g * kGroup + lane;
    const Summary mine = (i < nin) ? in[i] : identity();
    Summary s = wave_compose(mine);
    if (marks != nullptr) {
        const double g0 = goff[g];
        const bool sus = (i < nin) && likely_fails(mine, g0 + approx[i]);
        unsigned long long m = __ballot(sus);
        if (lane == 0) {
            const bool gs = likely_fails(s, g0);
            if (gs && m == 0ull) m = 1ull;          // descended into for another reason: its summaries are worth fetching anyway
            marks[g] = gs ? m : 0ull;
            s.bad = (s.bad & 1) | (gs ? kMarkBit : 0);
        }
    }
    if (lane == 0) out[g] = s;
}

// ---- D: the ordered walk ----------------------------------------------------------------

// Does the hypothesis behind `c` hold for the accumulator s = S * 2^(es-52)?
__device__ __forceinline__ bool holds(const Summary &c, int es, double S, bool neg_zero)
{
    if (c.bad) return false;
    if (c.e == kAnyExp) return !neg_zero;           // zeros only: s + 0 = s, except (-0) + (+0) = +0
    if (es == kNoExp || c.e != es) return false;
    if (S > 0.0) return (S + c.mn >= 0x1p52 + 1.0) && (S + c.mx <= 0x1p53 - 1.0);
    return (S + c.mx <= -(0x1p52 + 1.0)) && (S + c.mn >= -(0x1p53 - 1.0));
}

// The literal recurrence over elements [lo, hi): 64 coalesced products at a time, then 64
// dependent additions replayed identically by every lane.
template <class Src>
__device__ __forceinline__ double replay(const Src &src, long lo, long hi, double s, int lane)
{
    for (long base = lo; base < hi; base += 64) {
        const long i = base + lane;
        const double p = (i < hi) ? src.product(i) : 0.0;
        const int m = (hi - base < 64) ? (int)(hi - base) : 64;
        if (m == 64) {
#pragma unroll
            for (int l = 0; l < 64; ++l) s += __shfl(p, l, 64);
        } else {
            for (int l = 0; l < m; ++l) s += __shfl(p, l, 64);
        }
    }
    return s;
}

// One step of an inclusive prefix sum over the wavefront by data-parallel moves: the lane the control word names
// contributes its running sum (zero where there is no such lane or the row is masked off).
#define CG_SCAN_STEP(CTRL, ROW_MASK)                                                                              \
    D += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(D), CTRL, ROW_MASK, 0xf, true),           \
                          __builtin_amdgcn_update_dpp(0, __double2loint(D), CTRL, ROW_MASK, 0xf, true))

// Walk up to 64 summaries held one per lane (`mine`; lanes >= n hold nothing), in order,
// starting at index `from` from accumulator s.  Returns the index of the first summary
// that could not be applied (n if all were) and leaves s as the accumulator just before it.
// Every summary is first judged on its own against the accumulator's exponent: one that is flagged, or
// hypothesises another binade, counts as (0, -inf, +inf) and fails the range test.  Only the sums D are scanned:
// with E the sum of the D before a lane, that lane's summary applies iff the running value stays in range
// throughout it, S + E + mn and S + E + mx, and the walk stops at the first lane where it does not.  Up to there
// every partial sum is a difference of two running values inside one binade, an integer below 2^52: exact.
__device__ __forceinline__ int walk(const Summary &mine, int n, int from, double &s, int lane)
{
    const int es = exponent_of(s);
    const double S = (es == kNoExp) ? 0.0 : ldexp(s, 52 - es);
    const bool neg_zero = (s == 0.0) && signbit(s);
    const bool act = (lane >= from) && (lane < n);
    const bool zeros = mine.e == kAnyExp;             // adding zeros: s + 0 = s, except (-0) + (+0) = +0
    const bool poison = act && ((mine.bad & 1) != 0 || (zeros ? neg_zero : (es == kNoExp || mine.e != es)));
    const bool real = act && !poison && !zeros;
    double D = real ? mine.D : 0.0;
    const double own = D;
    CG_SCAN_STEP(0x111, 0xf);                         // row_shr:1
    CG_SCAN_STEP(0x112, 0xf);                         // row_shr:2
    CG_SCAN_STEP(0x114, 0xf);                         // row_shr:4
    CG_SCAN_STEP(0x118, 0xf);                         // row_shr:8  -> inclusive inside each row of 16 lanes
    CG_SCAN_STEP(0x142, 0xa);                         // row_bcast:15 into rows 1 and 3
    CG_SCAN_STEP(0x143, 0xc);                         // row_bcast:31 into rows 2 and 3
    const double E = D - own;
    const double lo = S + (E + (real ? mine.mn : 0.0)), hi = S + (E + (real ? mine.mx : 0.0));
    bool ok;
    if (es == kNoExp) ok = !real;                         // only zeros can be added to a zero / subnormal / non-finite accumulator
    else if (S > 0.0) ok = (lo >= 0x1p52 + 1.0) && (hi <= 0x1p53 - 1.0);
    else ok = (hi <= -(0x1p52 + 1.0)) && (lo >= -(0x1p53 - 1.0));
    ok = ok && !poison;
    const unsigned long long okmask = __ballot(ok);
    const unsigned long long upto = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    const unsigned long long want = upto & ~((1ull << from) - 1ull);
    const unsigned long long failed = want & ~okmask;
    const int f = failed ? (int)__builtin_ctzll(failed) : n;
    if (f > from) {
        const unsigned long long any_real = __ballot(real) & ((f >= 64) ? ~0ull : ((1ull << f) - 1ull));
        const double Dt = __shfl(D, f - 1, 64);
        if (any_real != 0ull && es != kNoExp) s = ldexp(S + Dt, es - 52);   // exact: integers below 2^53
    }
    return f;
}
#undef CG_SCAN_STEP

// The pieces of a staged chunk (16 consecutive products per lane) under the three binades around the one its
// approximate prefix sum predicts (h = 0, 1, 2: e_g - 1, e_g, e_g + 1), as the plan prepares them, in PREFIX form:
// P = the sum of the pieces' D up to and including this lane's, lo / hi = the extremes of the running value inside
// this lane's piece counted from the chunk's start (the sum before the piece plus its mn / mx).  A run of pieces
// from piece `from` on then needs no scan at all: with base = P of piece from-1, piece l applies iff
// S + (lo - base) and S + (hi - base) stay inside the binade, a test every lane makes for itself.  A piece that
// cannot be summarised (a tie, a NaN, a huge addend) counts 0 in P and carries its flag: the walk replays it
// literally and goes on behind it with base = its P.  The plan vouches for a hypothesis (bit 4+h clear) only if
// every P, lo, hi stays below 2^51 in magnitude, so that all these differences are exact.
struct Pieces {
    double P0, lo0, hi0, P1, lo1, hi1, P2, lo2, hi2;
    int flags;                                        // bit h: this piece is unusable under hypothesis h; bit 4+h: hypothesis h is not prepared
};
constexpr int kPieceDoubles = 10;                     // per lane: 3 x (P, lo, hi) and the flags
constexpr int kEntry = kChunk + 64 * kPieceDoubles;   // doubles per entry of the plan: products, pieces

// What the walk needs of an entry when nothing unforeseen happens, 36 doubles that every lane loads alike: with the
// accumulator as close to the approximate prefix sum as it always is, the pieces of a marked chunk fall into a run
// that certainly applies under the predicted binade e_g (pieces [0, f)), ONE piece f that may leave the binade (or
// holds a tie), and a run behind it that certainly applies under whichever of e_g - 1, e_g, e_g + 1 the accumulator
// then has.  The plan composes the two runs into summaries and copies piece f's products, so the walk's work per
// marked chunk is three range tests on wave-uniform values and sixteen additions: no ballot, no lane reads.  The walk
// still judges every summary by the TRUE accumulator, and takes the chunk apart the general way (refine) if any
// test fails.
//   [0..2] D, mn, mx of pieces [0, f) under e_g   [3] low word: f | flags << 8, high word: e_g
//   [4..19] the products of piece f                [20..28] D, mn, mx of pieces (f, 64) under e_g - 1, e_g, e_g + 1
//   [32..35] the summary of the stretch of chunks before the entry's chunk (D, mn, mx, bad / e)
constexpr int kBrief = 36;
constexpr int kStretchEmpty = 2;       // high word of [35]: there are no chunks between the previous entry's chunk and this one
constexpr int kBriefNoFast = 1;        // flags: the runs are not prepared (no usable e_g, ragged chunk, huge sums)
constexpr int kBriefAfterBad = 2;      // << h: the run behind piece f is unusable under hypothesis h
constexpr int kBriefNoBefore = 16;     // f == 0
constexpr int kBriefNoAfter = 32;      // f >= 63
constexpr int kBriefNoPiece = 64;      // f == 64: every piece applies under e_g

__device__ __forceinline__ double read_lane(double v, int l)        // l wave-uniform
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// A value every lane holds alike, moved to a scalar register: the compiler cannot know that it is uniform, and
// would otherwise steer the walk's control flow with lane masks and select operands through memory.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long uniform(long v)
{
    return ((long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)v);
}

// Piece f of a chunk (the 16 products lane f holds) added one by one, as the reference adds them.
__device__ __forceinline__ double replay_piece(const double (&p)[kPerLane], int f, long first, long hi, double s)
{
    if (first + kPerLane <= hi) {                     // wave-uniform
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) s += read_lane(p[i], f);
    } else {
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            const double pi = read_lane(p[i], f);
            if (first + i < hi) s += pi;
        }
    }
    return s;
}

// A chunk that is to be taken apart, its products p held as 16 consecutive elements per lane (one piece each),
// from piece `piece` on: summarise the pieces under the accumulator's ACTUAL exponent, apply the longest valid run,
// replay the one offending piece literally, and repeat with the new exponent.
__device__ __forceinline__ double refine_from(const double (&p)[kPerLane], int piece, long lo, long hi, double s, int lane,
                                              long *tally)
{
    for (int round = 0; piece < 64; ++round) {
        int f = piece;                               // after many crossings or ties: stop summarising, one piece at a time
        if (round < 24) {
            const Summary mine = piece_summary(p, exponent_of(s));
            f = uniform(walk(mine, 64, piece, s, lane));
            if (f >= 64) break;
            ++tally[0];
        } else {
            ++tally[1];
        }
        s = replay_piece(p, f, lo + (long)f * kPerLane, hi, s);
        piece = f + 1;
    }
    return s;
}

// The same with the plan's prepared pieces: as long as the accumulator's exponent is one of the three the plan
// vouches for, a run of pieces costs one test per lane and a ballot; otherwise the rest of the chunk goes the
// general way.  e_g, lo, hi wave-uniform (scalar).
__device__ __forceinline__ double refine(const double (&p)[kPerLane], const Pieces &pre, int e_g, long lo, long hi,
                                         double s, int lane, long *tally)
{
    int piece = 0;
    const int prepared = uniform(pre.flags >> 4);     // the same in every lane
    // nine separate registers from here on (the compiler would otherwise pick the hypothesis by ADDRESS and read
    // it back from scratch memory, in the middle of the walk's critical path)
    double P0 = pre.P0, lo0 = pre.lo0, hi0 = pre.hi0, P1 = pre.P1, lo1 = pre.lo1, hi1 = pre.hi1, P2 = pre.P2, lo2 = pre.lo2,
           hi2 = pre.hi2;
    asm volatile("" : "+v"(P0), "+v"(lo0), "+v"(hi0), "+v"(P1), "+v"(lo1), "+v"(hi1), "+v"(P2), "+v"(lo2), "+v"(hi2));
    while (piece < 64) {
        const int es = uniform(exponent_of(s));
        const int h = es - e_g + 1;
        if (e_g == kNoExp || es == kNoExp || h < 0 || h > 2 || ((prepared >> h) & 1))
            return refine_from(p, piece, lo, hi, s, lane, tally);
        const double S = ldexp(s, 52 - es);
        const double P = (h == 0) ? P0 : (h == 1) ? P1 : P2;
        const double plo = (h == 0) ? lo0 : (h == 1) ? lo1 : lo2;
        const double phi = (h == 0) ? hi0 : (h == 1) ? hi1 : hi2;
        const double base = (piece > 0) ? read_lane(P, piece - 1) : 0.0;
        const double a = S + (plo - base), b = S + (phi - base);
        bool ok = (S > 0.0) ? (a >= 0x1p52 + 1.0) && (b <= 0x1p53 - 1.0) : (b <= -(0x1p52 + 1.0)) && (a >= -(0x1p53 - 1.0));
        ok = ok && !((pre.flags >> h) & 1);
        const unsigned long long failed = ~__ballot(ok) & ~((1ull << piece) - 1ull);
        const int f = failed ? (int)__builtin_ctzll(failed) : 64;
        if (f > piece) s = ldexp(S + (read_lane(P, f - 1) - base), es - 52);     // exact: integers below 2^53
        if (f >= 64) break;
        ++tally[0];
        s = replay_piece(p, f, lo + (long)f * kPerLane, hi, s);
        piece = f + 1;
    }
    return s;
}

__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Ordered composition of summaries arr[lo .. hi) by one wavefront (result in every lane).
__device__ __forceinline__ Summary range_compose(const Summary *__restrict__ arr, long lo, long hi, int lane)
{
    Summary acc = identity();
    for (long i = lo; i < hi; i += 64) {
        Summary m = (i + lane < hi) ? arr[i + lane] : identity();
        m = wave_compose(m);
        m.D = __shfl(m.D, 0, 64); m.mn = __shfl(m.mn, 0, 64); m.mx = __shfl(m.mx, 0, 64);
        m.e = __shfl(m.e, 0, 64); m.bad = __shfl(m.bad, 0, 64);
        acc = compose(acc, m);
    }
    return acc;
}

// Summary of the chunks [a, b) from the three levels: chunks at the ragged ends, whole groups, whole groups of
// groups in the middle.
__device__ __forceinline__ Summary stretch_summary(const Summary *__restrict__ chunks, const Summary *__restrict__ groups,
                                                   const Summary *__restrict__ supers, long a, long b, int lane)
{
    if (a >= b) return identity();
    const long ga = (a + kGroup - 1) / kGroup, gb = b / kGroup;         // whole groups [ga, gb)
    if (ga >= gb) return range_compose(chunks, a, b, lane);             // fewer than 128 chunks, no whole group among them
    Summary acc = range_compose(chunks, a, ga * kGroup, lane);
    const long sa = (ga + kGroup - 1) / kGroup, sb = gb / kGroup;       // whole groups of groups [sa, sb)
    if (sa >= sb) {
        acc = compose(acc, range_compose(groups, ga, gb, lane));
    } else {
        acc = compose(acc, range_compose(groups, ga, sa * kGroup, lane));
        acc = compose(acc, range_compose(supers, sa, sb, lane));
        acc = compose(acc, range_compose(groups, sb * kGroup, gb, lane));
    }
    return compose(acc, range_compose(chunks, gb * kGroup, b, lane));
}

// One entry of the plan applied to the accumulator s from its brief B (kBrief doubles, wave-uniform values, in LDS or in
// global memory): first the stretch of chunks before the entry's chunk, then the chunk the prepared way.  Either returns
// false -- and leaves s untouched -- when its test fails for the true accumulator; the caller then does that part with
// the chunk summaries / the staged products, which only the rank that owns the data can.
__device__ __forceinline__ bool brief_stretch(const double *B, double &s)
{
    Summary st;
    st.D = B[32]; st.mn = B[33]; st.mx = B[34];
    st.e = uniform(__double2loint(B[35])); st.bad = uniform(__double2hiint(B[35])) & 1;
    const int es = uniform(exponent_of(s));
    const double S = (es == kNoExp) ? 0.0 : ldexp(s, 52 - es);
    if (!uniform((int)holds(st, es, S, (s == 0.0) && signbit(s)))) return false;
    if (st.e != kAnyExp) s = ldexp(S + st.D, es - 52);      // exact: integers below 2^53
    return true;
}

__device__ __forceinline__ bool brief_chunk(const double *B, int e_g, double &s)
{
    const int meta = uniform(__double2loint(B[3])), fl = meta >> 8;
    const int es = uniform(exponent_of(s));
    if ((fl & kBriefNoFast) || es == kNoExp || es != e_g) return false;
    double S = ldexp(s, 52 - es);
    bool ok = true;
    if (!(fl & kBriefNoBefore)) {
        const double a = S + B[1], c = S + B[2];
        ok = (S > 0.0) ? (a >= 0x1p52 + 1.0) && (c <= 0x1p53 - 1.0) : (c <= -(0x1p52 + 1.0)) && (a >= -(0x1p53 - 1.0));
        S += B[0];                                            // exact if ok: integers below 2^53
    }
    if (!uniform((int)ok)) return false;
    double t = ldexp(S, es - 52);
    if (!(fl & kBriefNoPiece)) {
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) t += B[4 + i];
    }
    if (fl & kBriefNoAfter) { s = t; return true; }
    const int es2 = uniform(exponent_of(t)), h2 = es2 - e_g + 1;
    if (es2 == kNoExp || h2 < 0 || h2 > 2 || ((fl >> (1 + h2)) & 1)) return false;
    const double *ar = B + 20 + 3 * h2;
    const double D = ar[0], mn = ar[1], mx = ar[2];
    const double S2 = ldexp(t, 52 - es2), a = S2 + mn, c = S2 + mx;
    const bool ok2 = (S2 > 0.0) ? (a >= 0x1p52 + 1.0) && (c <= 0x1p53 - 1.0) : (c <= -(0x1p52 + 1.0)) && (a >= -(0x1p53 - 1.0));
    if (!uniform((int)ok2)) return false;
    s = ldexp(S2 + D, es2 - 52);
    return true;
}

// The plan: the marked chunks (k_xs_group) in index order, at most kPlanCap of them, and for each an entry of
// kEntry doubles in `dump`: the kChunk products of chunk plist[k], lane-major as chunk16 delivers them; per lane
// the prepared piece summaries (Pieces); and the summary of the whole stretch of chunks between the previous
// entry's chunk and this one (entry n: the stretch after the last marked chunk) -- what the walk applies in one
// step when nothing unforeseen happens in between.  pexp[k] is the entry's e_g.  Every workgroup compacts the
// marks for itself (a few microseconds; saves a launch and a grid-wide hand-over), workgroup 0 publishes the
// list, and all of them share the staging.
template <class Src>
__global__ __launch_bounds__(256) void k_xs_plan(const Src src_in, long count, const Summary *__restrict__ chunks,
                                                 long nchunks, const Summary *__restrict__ groups,
                                                 const Summary *__restrict__ supers,
                                                 const unsigned long long *__restrict__ marks,
                                                 long ngroups, const double *__restrict__ approx,
                                                 const double *__restrict__ goff, int *__restrict__ plist,
                                                 int *__restrict__ pexp, int *__restrict__ pn, double *__restrict__ dump,
                                                 const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    __shared__ __attribute__((aligned(16))) double T[4][Src::kLds];
    __shared__ int clist[kPlanCap];
    __shared__ int wave_tot[4];
    Src src = src_in;
    src.prepare();
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const long per = (ngroups + 255) / 256;
    const long g_lo = (long)t * per, g_hi = (g_lo + per < ngroups) ? g_lo + per : ngroups;
    int mine = 0;
    for (long g = g_lo; g < g_hi; ++g) mine += __popcll(marks[g]);
    int inc = mine;                                  // inclusive scan over the workgroup
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(inc, off, 64);
        if (lane >= off) inc += up;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    int pos = inc - mine, total = 0;
    for (int k = 0; k < 4; ++k) { if (k < wv) pos += wave_tot[k]; total += wave_tot[k]; }
    for (long g = g_lo; g < g_hi && pos < kPlanCap; ++g) {
        unsigned long long m = marks[g];
        while (m != 0ull && pos < kPlanCap) {
            clist[pos++] = (int)(g * kGroup + __builtin_ctzll(m));
            m &= m - 1ull;
        }
    }
    __syncthreads();
    const int n = total < kPlanCap ? total : kPlanCap;
    if (blockIdx.x == 0) {
        for (int k = t; k < n; k += 256) plist[k] = clist[k];
        if (t == 0) *pn = n;
    }
    // 2n + 1 pieces of work, one wavefront each: the n + 1 stretch summaries, then the n chunks
    for (int item = blockIdx.x * 4 + wv; item <= 2 * n; item += gridDim.x * 4) {
        const int k = (item <= n) ? item : item - n - 1;
        double *ent = dump + (long)k * kEntry;
        double *brief = dump + (long)(kPlanCap + 1) * kEntry + (long)k * kBrief;
        if (item <= n) {
            const long from = (k == 0) ? 0 : (long)clist[k - 1] + 1, to = (k < n) ? (long)clist[k] : nchunks;
            const Summary st = stretch_summary(chunks, groups, supers, from, to, lane);
            if (lane == 0) {
                xv2 *ps = reinterpret_cast<xv2 *>(brief + 32);
                xv2 v0, v1;
                v0.x = st.D; v0.y = st.mn; v1.x = st.mx; v1.y = __hiloint2double((st.bad & 1) | (from >= to ? kStretchEmpty : 0), st.e);
                ps[0] = v0; ps[1] = v1;
            }
            continue;
        }
        const long c = clist[k];
        const long lo = c * kChunk, hi = (lo + kChunk < count) ? lo + kChunk : count;
        double p[kPerLane];
        src.chunk16(lo, hi, lane, T[wv], p);
        xv2 *dst = reinterpret_cast<xv2 *>(ent + lane * kPerLane);
#pragma unroll
        for (int i = 0; i < kPerLane / 2; ++i) { xv2 v; v.x = p[2 * i]; v.y = p[2 * i + 1]; dst[i] = v; }
        const int e_g = exponent_of(goff[c / kGroup] + approx[c]);
        if (lane == 0) pexp[k] = e_g;
        double bf[30];                                   // the brief (wave-uniform values)
#pragma unroll
        for (int i = 0; i < 30; ++i) bf[i] = 0.0;
        int f = 0, bflags = kBriefNoFast;
        if (e_g != kNoExp) {
            double out[kPieceDoubles];
            int flags = 0;
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const Summary q = piece_summary(p, e_g + h - 1);
                const bool bad = (q.bad & 1) != 0;
                const double D = bad ? 0.0 : q.D;
                double P = D;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double up = __shfl_up(P, off, 64);
                    if (lane >= off) P += up;
                }
                const double E = P - D;
                const double plo = bad ? E : E + q.mn, phi = bad ? E : E + q.mx;
                const bool big = !(fabs(P) < 0x1p51) || !(fabs(plo) < 0x1p51) || !(fabs(phi) < 0x1p51);
                out[3 * h] = P; out[3 * h + 1] = plo; out[3 * h + 2] = phi;
                if (bad) flags |= 1 << h;
                if (__ballot(big) != 0ull) flags |= 16 << h;
            }
            out[9] = __hiloint2double(0, flags);
            xv2 *pd = reinterpret_cast<xv2 *>(ent + kChunk + lane * kPieceDoubles);
#pragma unroll
            for (int i = 0; i < kPieceDoubles / 2; ++i) { xv2 v; v.x = out[2 * i]; v.y = out[2 * i + 1]; pd[i] = v; }
            // The brief: which pieces certainly apply, given how close to the approximation the accumulator will be.
            const double A = goff[c / kGroup] + approx[c];
            const double Ag = ldexp(A, 52 - e_g);
            const double lo_m = Ag + out[4] - kPlanMargin, hi_m = Ag + out[5] + kPlanMargin;
            const bool certain = !(flags & 2) && ((A > 0.0) ? (lo_m >= 0x1p52 + 1.0) && (hi_m <= 0x1p53 - 1.0)
                                                            : (hi_m <= -(0x1p52 + 1.0)) && (lo_m >= -(0x1p53 - 1.0)));
            const unsigned long long unsure = ~__ballot(certain);
            f = unsure ? (int)__builtin_ctzll(unsure) : 64;
            bflags = ((flags & (16 << 1)) || hi - lo < kChunk) ? kBriefNoFast : 0;
            if (f == 0) bflags |= kBriefNoBefore;
            if (f >= 63) bflags |= kBriefNoAfter;
            if (f == 64) bflags |= kBriefNoPiece;
            if (f > 0) {
                bf[0] = read_lane(out[3], f - 1);
                bf[1] = wave_min(lane < f ? out[4] : HUGE_VAL);
                bf[2] = wave_max(lane < f ? out[5] : -HUGE_VAL);
            }
            if (f < 64) {
#pragma unroll
                for (int i = 0; i < kPerLane; ++i) bf[4 + i] = read_lane(p[i], f);
            }
            if (f < 63) {
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const double base = read_lane(out[3 * h], f);
                    bf[20 + 3 * h] = read_lane(out[3 * h], 63) - base;
                    bf[21 + 3 * h] = wave_min(lane > f ? out[3 * h + 1] : HUGE_VAL) - base;
                    bf[22 + 3 * h] = wave_max(lane > f ? out[3 * h + 2] : -HUGE_VAL) - base;
                    if (__ballot(lane > f && ((flags >> h) & 1)) != 0ull || (flags & (16 << h))) bflags |= kBriefAfterBad << h;
                }
            }
        }
        bf[3] = __hiloint2double(e_g, f | (bflags << 8));       // (high word: the predicted binade, for a walk that has only the briefs)
        if (lane < 15) {                                  // lane i stores doubles 2i, 2i + 1
            xv2 v;
            v.x = bf[0]; v.y = bf[1];
#pragma unroll
            for (int i = 1; i < 15; ++i) if (lane == i) { v.x = bf[2 * i]; v.y = bf[2 * i + 1]; }
            reinterpret_cast<xv2 *>(brief)[lane] = v;
        }
    }
}

constexpr int kBriefBatch = 64;                      // briefs the walk keeps in LDS at a time

// The hierarchical walk over the chunks [a, b), without a plan: group summaries 64 at a time where whole groups
// lie ahead, the chunks of a group that does not apply (or is only partly inside the range), the elements of a
// chunk that does not apply.  Everything it needs is fetched on the spot.
template <class Src>
__device__ __forceinline__ double walk_range(const Src &src, long count, const Summary *__restrict__ chunks,
                                             long nchunks, const Summary *__restrict__ groups, long a, long b, double s,
                                             int lane, double *T, long &n_groups, long &n_chunks, long *tally)
{
    long pos = a;
    bool descend = false;
    while (pos < b) {
        if (!descend && pos % kGroup == 0 && pos + kGroup <= b) {
            const long g0 = pos / kGroup, gw = b / kGroup - g0;
            const int ng = (gw < 64) ? (int)gw : 64;
            const Summary gmine = (lane < ng) ? groups[g0 + lane] : identity();
            const int gf = walk(gmine, ng, 0, s, lane);
            pos += (long)gf * kGroup;
            if (gf < ng) { descend = true; ++n_groups; }
        } else {
            const long cb = (pos / kGroup) * kGroup;
            const int from = (int)(pos - cb);
            const long room = ((b < nchunks) ? b : nchunks) - cb;
            const int nc = (room < 64) ? (int)room : 64;
            const Summary cmine = (lane < nc) ? chunks[cb + lane] : identity();
            const int cf = walk(cmine, nc, from, s, lane);
            if (cf < nc) {
                ++n_chunks;
                const long lo = (cb + cf) * kChunk, hi = (lo + kChunk < count) ? lo + kChunk : count;
                double p[kPerLane];
                src.chunk16(lo, hi, lane, T, p);
                s = refine_from(p, 0, lo, hi, s, lane, tally);
                pos = cb + cf + 1;
            } else {
                pos = cb + nc;
            }
            if (pos % kGroup == 0) descend = false;
        }
    }
    return s;
}

// The ordered walk, one wavefront -- so nothing hides a wait, and everything it will need is laid out for it
// beforehand by the plan (plist / pexp / pn / dump; plist may be null: then the whole range is walked
// hierarchically).  Per entry of the plan: the stretch of chunks before the entry's chunk is applied in one step if
// its summary holds for the true accumulator (walked hierarchically otherwise), then the chunk is taken apart: by
// its brief (two prepared runs of pieces around one piece added literally) if the brief's tests hold for the true
// accumulator, else with its staged products and prepared pieces, fetched then (taking apart a chunk that would
// have applied after all costs a few tests and is as exact as everything else).
// stats (may be null): [0] stretches walked hierarchically plus groups descended into, [1] chunks taken apart,
// [2] rounds of piece summaries inside them, [3] pieces replayed without summarising, [4] chunks taken apart
// according to the plan, [5] those of them whose brief did not hold.
template <class Src>
__global__ __launch_bounds__(64) void k_xs_walk(const Src src_in, long count,
                                                const Summary *__restrict__ chunks, long nchunks,
                                                const Summary *__restrict__ groups, long ngroups, double start,
                                                const double *start_dev, double *result, long *stats,
                                                const int *__restrict__ plist, const int *__restrict__ pexp,
                                                const int *__restrict__ pn, const double *__restrict__ dump,
                                                const int *skip = nullptr, const FinishArgs fin = FinishArgs())
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    __shared__ __attribute__((aligned(16))) double T[Src::kLds];
    __shared__ int lst[kPlanCap], lexp[kPlanCap];
    __shared__ __attribute__((aligned(16))) double lbrief[kBriefBatch * kBrief];
    Src src = src_in;
    src.prepare();
    const int lane = threadIdx.x;
    double s = (start_dev != nullptr) ? *start_dev : start;
    long n_groups = 0, n_chunks = 0, n_planned = 0, n_general = 0, tally[2] = {0, 0};
#ifdef CG_WALK_CLOCKS
    long clk_all = clock64();
#endif
    if (plist == nullptr) {
        s = walk_range(src, count, chunks, nchunks, groups, 0, nchunks, s, lane, T, n_groups, n_chunks, tally);
    } else {
        int n = uniform(*pn);
        if (n > kPlanCap) n = kPlanCap;
        const double *briefs = dump + (long)(kPlanCap + 1) * kEntry;
        // Briefs travel 64 at a time: lane-strided 16-byte loads into registers (the next batch is fetched while
        // this one is walked), then into LDS, where the walk reads them at wave-uniform addresses.
        constexpr int kPer = kBriefBatch * kBrief / 2 / 64;        // 16-byte pieces per lane and batch
        xv2 stage[kPer] = {};
        auto fetch_batch = [&](int first) {
            if (first > n) return;
            const xv2 *from = reinterpret_cast<const xv2 *>(briefs + (long)first * kBrief);
            const int have = ((n + 1 - first < kBriefBatch) ? n + 1 - first : kBriefBatch) * (kBrief / 2);
#pragma unroll
            for (int i = 0; i < kPer; ++i)
                if (i * 64 + lane < have) stage[i] = from[i * 64 + lane];
        };
        auto store_batch = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the last batch has been read
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < kPer; ++i) reinterpret_cast<xv2 *>(lbrief)[i * 64 + lane] = stage[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        fetch_batch(0);
        for (int k = lane; k < n; k += 64) { lst[k] = plist[k]; lexp[k] = pexp[k]; }
        long pos = 0;
        for (int k = 0; k <= n; ++k) {                 // entry k: the stretch before its chunk, then the chunk
            if ((k & (kBriefBatch - 1)) == 0) {
                store_batch();
                fetch_batch(k + kBriefBatch);
            }
            const long end = (k < n) ? (long)uniform(lst[k]) : nchunks;
            const double *B = lbrief + (k & (kBriefBatch - 1)) * kBrief;
            if (pos < end && !brief_stretch(B, s)) {
                ++n_groups;
                s = walk_range(src, count, chunks, nchunks, groups, pos, end, s, lane, T, n_groups, n_chunks, tally);
            }
            if (k == n) break;
            const long lo = end * kChunk, hi = (lo + kChunk < count) ? lo + kChunk : count;
            const int e_g = uniform(lexp[k]);
            ++n_chunks; ++n_planned;
            // The prepared way: a run of pieces under e_g, one piece literally, a run under the binade reached.
            const bool done = brief_chunk(B, e_g, s);
            if (!done) {                               // the general way, from the chunk's start: its staged products and pieces
                ++n_general;
                const double *base = dump + (long)k * kEntry;
                const xv2 *from = reinterpret_cast<const xv2 *>(base + lane * kPerLane);
                const xv2 *fq = reinterpret_cast<const xv2 *>(base + kChunk + lane * kPieceDoubles);
                double p[kPerLane];
                Pieces pre;
#pragma unroll
                for (int i = 0; i < kPerLane / 2; ++i) { const xv2 v = from[i]; p[2 * i] = v.x; p[2 * i + 1] = v.y; }
                const xv2 q0 = fq[0], q1 = fq[1], q2 = fq[2], q3 = fq[3], q4 = fq[4];
                pre.P0 = q0.x; pre.lo0 = q0.y; pre.hi0 = q1.x;
                pre.P1 = q1.y; pre.lo1 = q2.x; pre.hi1 = q2.y;
                pre.P2 = q3.x; pre.lo2 = q3.y; pre.hi2 = q4.x;
                pre.flags = __double2loint(q4.y);
                s = refine(p, pre, e_g, lo, hi, s, lane, tally);
            }
            pos = end + 1;
        }
    }
    double s2 = 0.0;
    if (fin.sc != nullptr && fin.partials2 != nullptr) {
        for (long k = lane; k < fin.count2; k += 64) s2 += fin.partials2[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s2 += __shfl_xor(s2, off, 64);
    }
    if (lane == 0) {
        *result = s;
        if (fin.sc != nullptr) finish_scalars(fin.sc, fin.what, s, s2, fin.partials2 != nullptr, fin.count2, fin.rtol, fin.hist, fin.hist_cap);
        if (stats != nullptr) {
            stats[0] = n_groups; stats[1] = n_chunks; stats[2] = tally[0]; stats[3] = tally[1]; stats[4] = n_planned;
            stats[5] = n_general;
#ifdef CG_WALK_CLOCKS
            stats[6] = clock64() - clk_all;
#endif
        }
    }
}

// ---- several ranks: the walk finished on every rank from published briefs ------------------------------------------
// Row blocks follow one another in the summation order, so the accumulator has to pass through the ranks' blocks in
// turn.  Handing it from rank to rank costs a collective per rank and dot product.  Instead every rank publishes what
// a walk over its block needs when nothing unforeseen happens -- its plan's briefs, a few hundred bytes -- in ONE
// all-gather; rank 0, whose block holds nearly all the binade crossings, walks its own block itself and publishes the
// exact accumulator it ends with; and every rank then continues that walk through the other ranks' briefs
// (k_xs_walk_remote), all of them with the same bits.  A brief whose test fails for the true accumulator, or a rank
// with more entries than fit, cannot be dealt with from afar: the kernel then raises the solve's fault flag, every
// later kernel returns at once, and the host completes that dot product with the ranks walking in turn (cg_solve).
constexpr int kRemoteCap = 32;                                   // plan entries a rank can publish
constexpr int kPayloadHead = 4;                                  // [0] exact accumulator after the block (the rank that walked it itself),
                                                                 // [1] entries published (-1: too many), [2] the block's q.q (prediction input), [3] spare
constexpr int kPayload = kPayloadHead + (kRemoteCap + 1) * kBrief;

__global__ __launch_bounds__(64) void k_xs_pack(const int *__restrict__ pn, const double *__restrict__ dump,
                                                const double *result, const double *__restrict__ partials2, long count2,
                                                double *__restrict__ slot, const int *skip)
{
    if (skip != nullptr && *skip) return;
    const int lane = threadIdx.x;
    int n = uniform(*pn);
    if (n > kPlanCap) n = kPlanCap;
    double qq = 0.0;
    for (long k = lane; k < count2; k += 64) qq += partials2[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) qq += __shfl_xor(qq, off, 64);
    if (lane == 0) {
        slot[0] = (result != nullptr) ? *result : 0.0;
        slot[1] = (n <= kRemoteCap) ? (double)n : -1.0;
        slot[2] = qq;
        slot[3] = 0.0;
    }
    if (n > kRemoteCap) return;
    const double *briefs = dump + (long)(kPlanCap + 1) * kEntry;
    for (int i = lane; i < (n + 1) * kBrief; i += 64) slot[kPayloadHead + i] = briefs[i];
}

// slots: nranks payloads in rank order, the first of them of the rank that walked its own block (slot[0] valid).
// quiet: a timing rehearsal with one rank standing in for all -- the other slots hold nothing, nothing is raised.
// fault_every > 0 (test hook): declare every fault_every-th dot product failed, counted over the solve's iterations.
__global__ __launch_bounds__(64) void k_xs_walk_remote(const double *__restrict__ slots, int nranks, int quiet, Scalars *sc, int phase,
                                                       long fault_every, double *result, double *qq_out, const FinishArgs fin)
{
    if (sc->done) return;
    double s = slots[0], qq = slots[2];
    bool ok = true;
    for (int j = 1; j < nranks && ok && !quiet; ++j) {
        const double *sl = slots + (long)j * kPayload;
        const int n = uniform((int)sl[1]);
        qq += sl[2];
        if (n < 0) { ok = false; break; }
        for (int k = 0; k <= n; ++k) {
            const double *B = sl + kPayloadHead + k * kBrief;
            const bool empty = (uniform(__double2hiint(B[35])) & kStretchEmpty) != 0;
            if (!empty && !brief_stretch(B, s)) { ok = false; break; }
            if (k == n) break;
            if (!brief_chunk(B, uniform(__double2hiint(B[3])), s)) { ok = false; break; }
        }
    }
    if (fault_every > 0 && ((2 * sc->iter + phase) % fault_every) == fault_every - 1) ok = false;
    if (threadIdx.x != 0) return;
    if (ok || quiet) {
        *result = s;
        *qq_out = qq;
        // the solver's scalar step right here (q.q of the whole grid rides on the d.q walk)
        if (fin.sc != nullptr) finish_scalars(fin.sc, fin.what, s, qq, phase == 0, 1, fin.rtol, fin.hist, fin.hist_cap);
    } else {
        sc->fault = 1 + phase;
        sc->done = 2;
    }
}

// Self-check (CG_OPT_STRICT_AUDIT): every usable chunk summary against one recomputed from the products under the
// same hypothesis -- the sum must be the same integer, the stored extremes must enclose the true ones -- and the
// result of the plan-driven walk against the hierarchical walk over the same summaries.  Failures are counted.
template <class Src>
__global__ __launch_bounds__(256) void k_xs_audit(const Src src_in, long count, const Summary *__restrict__ chunks, long nchunks,
                                                  unsigned long long *fails, const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    __shared__ __attribute__((aligned(16))) double T[4][Src::kLds];
    Src src = src_in;
    src.prepare();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (long c = (long)blockIdx.x * 4 + wv; c < nchunks; c += (long)gridDim.x * 4) {
        const Summary st = chunks[c];
        if ((st.bad & 1) || st.e == kAnyExp) continue;
        const long lo = c * kChunk, hi = (lo + kChunk < count) ? lo + kChunk : count;
        double p[kPerLane];
        src.chunk16(lo, hi, lane, T[wv], p);
        Summary t = wave_compose_same_e(piece_summary(p, st.e, true), st.e);
        t.D = __shfl(t.D, 0, 64); t.mn = __shfl(t.mn, 0, 64); t.mx = __shfl(t.mx, 0, 64); t.bad = __shfl(t.bad, 0, 64);
        if (lane == 0 && (t.bad || t.D != st.D || t.mn < st.mn || t.mx > st.mx)) {
            atomicAdd(fails, 1ull);
#ifdef CG_DEBUG_XS
            printf("audit: chunk %ld e %d: stored D %a mn %a mx %a | true D %a mn %a mx %a bad %d\n", c, st.e, st.D, st.mn, st.mx,
                   t.D, t.mn, t.mx, t.bad);
#endif
        }
    }
}
__global__ void k_xs_audit_result(const double *plan, const double *hier, unsigned long long *fails, const int *skip = nullptr)
{
    if (skip != nullptr && *skip) return;               // a stopped or faulted solve: the workspace stays as it is
    const double a = *plan, b = *hier;
    if (!(a == b || (a != a && b != b))) {
        atomicAdd(fails, 1ull);
#ifdef CG_DEBUG_XS
        printf("audit: plan-driven walk %a, hierarchical walk %a\n", a, b);
#endif
    }
}

#ifdef CG_DEBUG_XS
// Diagnostic build only: the same dot product by the literal recurrence, and a comparison that reports.
template <class Src>
__global__ __launch_bounds__(64) void k_xs_literal(const Src src_in, long count, double start, double *result)
{
    Src src = src_in;
    src.prepare();
    const double s = replay(src, 0, count, start, (int)threadIdx.x);
    if (threadIdx.x == 0) *result = s;
}
__global__ void k_xs_compare(const double *plan, const double *hier, const double *lit, int which)
{
    if (*plan != *lit || *hier != *lit)
        printf("strict dot %d: plan %a hierarchical %a literal %a\n", which, *plan, *hier, *lit);
}
#endif

// Workspace for `count` elements.
struct Workspace {
    double *approx = nullptr, *goff = nullptr;
    double *sub = nullptr;                // span sums handed over by the producing kernel (kSub products each)
    Summary *chunks = nullptr, *groups = nullptr, *supers = nullptr;   // three levels of summaries: 1024, 64 x 1024, 64 x 64 x 1024 products
    long *stats = nullptr;                // what the last walks did: [8*which + 0..4], see k_xs_walk
    unsigned long long *marks = nullptr;  // per group: the chunks the walk is expected to take apart
    int *plist = nullptr, *pexp = nullptr, *pn = nullptr;  // the plan: those chunks in order, their predicted binades, how many
    double *dump = nullptr;               // their products and prepared pieces (kEntry doubles each), then their briefs (kBrief each)
    unsigned long long *audit = nullptr;  // [0] failures counted by the self-check, [1] scratch (its second result)
    long cap_chunks = 0;
    const int *skip = nullptr;            // device flag (a solve's Scalars::done), or null: when set, the kernels of the speculative
                                          // path and the walks return at once and leave the workspace as it is
    // speculative path (reserve_spec): span sums under two hypotheses, the hypotheses of the two dot products of
    // an iteration (they persist from one iteration to the next), the repair list
    double *rec = nullptr;                            // one record of four doubles per unit of cg::kUnit products (UnitRec)
    double *hyp[2] = {nullptr, nullptr};              // last iteration's prefix sum at every unit (0 = none yet)
    int *fixlist = nullptr, *nfix = nullptr;
    long cap_spec = 0;

    static long chunks_for(long count) { return (count + kChunk - 1) / kChunk; }
    hipError_t reserve(long count)
    {
        const long nc = chunks_for(count);
        if (nc <= cap_chunks) return hipSuccess;
        release();
        const long ng = (nc + kGroup - 1) / kGroup;
        hipError_t e;
        if ((e = hipMalloc(&sub, (size_t)nc * (kChunk / kSub) * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&stats, 256)) != hipSuccess) return e;
        if ((e = hipMemset(stats, 0, 256)) != hipSuccess) return e;
        if ((e = hipMalloc(&approx, (size_t)nc * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&goff, (size_t)ng * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&chunks, (size_t)nc * sizeof(Summary))) != hipSuccess) return e;
        if ((e = hipMalloc(&groups, (size_t)ng * sizeof(Summary))) != hipSuccess) return e;
        if ((e = hipMalloc(&supers, (size_t)((ng + kGroup - 1) / kGroup) * sizeof(Summary))) != hipSuccess) return e;
        if ((e = hipMalloc(&marks, (size_t)ng * sizeof(unsigned long long))) != hipSuccess) return e;
        if ((e = hipMalloc(&plist, (size_t)kPlanCap * sizeof(int))) != hipSuccess) return e;
        if ((e = hipMalloc(&pexp, (size_t)kPlanCap * sizeof(int))) != hipSuccess) return e;
        if ((e = hipMalloc(&pn, 256)) != hipSuccess) return e;
        if ((e = hipMemset(pn, 0, 256)) != hipSuccess) return e;
        if ((e = hipMalloc(&audit, 256)) != hipSuccess) return e;
        if ((e = hipMemset(audit, 0, 256)) != hipSuccess) return e;
        if ((e = hipMalloc(&dump, (size_t)(kPlanCap + 1) * (kEntry + kBrief) * sizeof(double))) != hipSuccess) return e;
        cap_chunks = nc;
        return hipSuccess;
    }
    // Arrays of the speculative path for `count` elements; the hypotheses start as "none" (everything is
    // repaired the first time round, which also produces real hypotheses for the next).
    hipError_t reserve_spec(long count)
    {
        const long nc = chunks_for(count);
        if (nc <= cap_spec) return hipSuccess;
        release_spec();
        const size_t nspan = (size_t)nc * (kChunk / kUnit);
        hipError_t e;
        if ((e = hipMalloc(&rec, 4 * nspan * sizeof(double))) != hipSuccess) return e;
        for (int k = 0; k < 2; ++k) {
            if ((e = hipMalloc(&hyp[k], nspan * sizeof(double))) != hipSuccess) return e;
            if ((e = hipMemset(hyp[k], 0, nspan * sizeof(double))) != hipSuccess) return e;
        }
        if ((e = hipMalloc(&fixlist, (size_t)nc * sizeof(int))) != hipSuccess) return e;
        if ((e = hipMalloc(&nfix, 256)) != hipSuccess) return e;
        if ((e = hipMemset(nfix, 0, 256)) != hipSuccess) return e;
        if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
        cap_spec = nc;
        return hipSuccess;
    }
    void forget_hypotheses()                              // a new problem: last solve's prefix sums say nothing about it
    {
        if (cap_spec == 0) return;
        const size_t nspan = (size_t)cap_spec * (kChunk / kUnit);
        for (int k = 0; k < 2; ++k) (void)hipMemset(hyp[k], 0, nspan * sizeof(double));
        (void)hipDeviceSynchronize();
    }
    void release_spec()
    {
        (void)hipFree(rec);
        (void)hipFree(hyp[0]); (void)hipFree(hyp[1]); (void)hipFree(fixlist); (void)hipFree(nfix);
        rec = nullptr; hyp[0] = hyp[1] = nullptr; fixlist = nullptr; nfix = nullptr;
        cap_spec = 0;
    }
    void release()
    {
        release_spec();
        (void)hipFree(approx); (void)hipFree(goff); (void)hipFree(chunks); (void)hipFree(groups); (void)hipFree(supers);
        (void)hipFree(sub); (void)hipFree(stats); (void)hipFree(marks); (void)hipFree(plist); (void)hipFree(pexp); (void)hipFree(pn); (void)hipFree(audit);
        (void)hipFree(dump);
        approx = nullptr; goff = nullptr; chunks = nullptr; groups = nullptr; supers = nullptr; sub = nullptr; stats = nullptr;
        marks = nullptr; plist = nullptr; pexp = nullptr; pn = nullptr; dump = nullptr; audit = nullptr;
        cap_chunks = 0;
    }
};

// result <- the reference's sequential dot product of v and w started from `start`, or from
// *start_dev (read on the device when the kernels run: a value an earlier kernel or a
// neighbouring rank produced) when start_dev is not null.  `result` may alias `start_dev`.
// Stream-ordered; the workspace must have been reserved for `count`.  have_sub: ws.sub already
// holds approximate sums of every aligned span of kSub products (written by the kernel that
// produced v and w; sub_G > 0: in the by-rows layout of k_xs_from_sub), so the operands are read
// once instead of twice.  Those sums only steer
// the guesses; every guess is verified by the walk, so their accuracy affects speed, never bits.
//
// Three phases, so that several ranks whose blocks follow one another in the index order can do the first two
// side by side and only the walks one after the other:
//   approx     approximate chunk sums and group sums (total_out: their total, for the ranks that come after)
//   summaries  everything up to the plan, the guesses seeded with `start` or the sum of nseed device values (the
//              approximate totals of the ranks before this one: a seed steers, it need not be exact)
//   walk       the ordered walk from the exact accumulator
struct Arrays {
    const double *v, *w;
    long count;
    bool have_sub = false;
    long sub_G = 0, sub_rows = 0;
    int vec() const { return ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(w)) & 15) == 0 ? 1 : 0; }   // (null w: v alone)
    SrcArrays src() const { return SrcArrays{v, w, count > 0 ? count : 0, vec()}; }
};

inline void enqueue_approx(hipStream_t s, const Workspace &ws, const Arrays &a, double *total_out = nullptr)
{
    if (a.count <= 0) {
        if (total_out) hipLaunchKernelGGL(k_xs_total, dim3(1), dim3(1024), 0, s, ws.goff, 0L, total_out);
        return;
    }
    const long nc = Workspace::chunks_for(a.count);
    const long ng = (nc + kGroup - 1) / kGroup;
    if (a.have_sub) {
        hipLaunchKernelGGL(k_xs_from_sub, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, (const double *)ws.sub,
                           (a.count + kSub - 1) / kSub, nc, a.sub_G, a.sub_rows, ws.approx, ws.goff, (int)kSub, 1);
    } else {
        hipLaunchKernelGGL(k_xs_approx, dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, s, a.v, a.w, a.count, nc, a.vec(), ws.approx);
        hipLaunchKernelGGL(k_xs_gsum, dim3((unsigned)((ng + 3) / 4)), dim3(256), 0, s, ws.approx, nc, ws.goff, ng);
    }
    if (total_out) hipLaunchKernelGGL(k_xs_total, dim3(1), dim3(1024), 0, s, ws.goff, ng, total_out);
}

template <class Src>
inline void enqueue_levels_and_plan(hipStream_t s, const Workspace &ws, const Src &src, long count)
{
    const long nc = Workspace::chunks_for(count);
    const long ng = (nc + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(k_xs_group, dim3((unsigned)((ng + 3) / 4)), dim3(256), 0, s, ws.chunks, nc, ws.groups, ng, ws.approx, ws.goff, ws.marks, ws.skip);
    const long nsup = (ng + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(k_xs_group, dim3((unsigned)((nsup + 3) / 4)), dim3(256), 0, s, ws.groups, ng, ws.supers, nsup,
                       (const double *)nullptr, (const double *)nullptr, (unsigned long long *)nullptr, ws.skip);
    hipLaunchKernelGGL(k_xs_plan<Src>, dim3(kPlanBlocks), dim3(256), 0, s, src, count, ws.chunks, nc, ws.groups, ws.supers,
                       ws.marks, ng, ws.approx, ws.goff, ws.plist, ws.pexp, ws.pn, ws.dump, ws.skip);
}

inline void enqueue_summaries(hipStream_t s, const Workspace &ws, const Arrays &a, double start, const double *seed_dev = nullptr,
                              int nseed = 0)
{
    if (a.count <= 0) return;
    const long nc = Workspace::chunks_for(a.count);
    const long ng = (nc + kGroup - 1) / kGroup;
    const unsigned cb = (unsigned)((nc + 3) / 4);
    hipLaunchKernelGGL(k_xs_gscan, dim3(1), dim3(1024), 0, s, ws.goff, ng, start, seed_dev, nseed, (int *)nullptr);
    const unsigned cbp = cb < 1024u ? cb : 1024u;          // persistent: the four workgroups per CU that LDS allows
    if (a.w != nullptr)
        hipLaunchKernelGGL(k_xs_chunk<true>, dim3(cbp), dim3(256), 0, s, a.v, a.w, a.count, nc, a.vec(), ws.approx, ws.goff, ws.chunks);
    else
        hipLaunchKernelGGL(k_xs_chunk<false>, dim3(cbp), dim3(256), 0, s, a.v, a.w, a.count, nc, a.vec(), ws.approx, ws.goff, ws.chunks);
    enqueue_levels_and_plan(s, ws, a.src(), a.count);
}

template <class Src>
inline void enqueue_walk(hipStream_t s, const Workspace &ws, const Src &src, long count, double start, const double *start_dev,
                         double *result, long *stats, const FinishArgs &fin = FinishArgs())
{
    if (count <= 0) {
        hipLaunchKernelGGL(k_xs_walk<Src>, dim3(1), dim3(64), 0, s, src, 0L, ws.chunks, 0L, ws.groups, 0L,
                           start, start_dev, result, (long *)nullptr, (const int *)nullptr, (const int *)nullptr,
                           (const int *)nullptr, (const double *)nullptr, ws.skip, fin);
        return;
    }
    const long nc = Workspace::chunks_for(count);
    const long ng = (nc + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(k_xs_walk<Src>, dim3(1), dim3(64), 0, s, src, count, ws.chunks, nc, ws.groups, ng,
                       start, start_dev, result, stats, ws.plist, ws.pexp, ws.pn, ws.dump, ws.skip, fin);
}

inline void enqueue(hipStream_t s, const Workspace &ws, const double *v, const double *w, long count, double start,
                    double *result, const double *start_dev = nullptr, bool have_sub = false, long sub_G = 0,
                    long sub_rows = 0)
{
    Arrays a{v, w, count, have_sub, sub_G, sub_rows};
    enqueue_approx(s, ws, a);
    enqueue_summaries(s, ws, a, start, start_dev, start_dev ? 1 : 0);
    enqueue_walk(s, ws, a.src(), count, start, start_dev, result, ws.stats);
}

// The same result from the speculative span sums the producing kernel left in the workspace (sub, spD*, spQ*
// under the hypotheses ws.hyp[which]); `src` recomputes or re-reads single products where the walk or the
// repair of missed chunks needs them.  The operands are `rows` rows of G products, G a multiple of kChunk.
// nonneg: every product is >= 0 (a vector with itself), the kernel stored no Q arrays.
// The three phases for the span sums: spec_approx, spec_summaries, and enqueue_walk above.
inline void spec_approx(hipStream_t s, const Workspace &ws, long G, long rows, double *total_out = nullptr)
{
    const long count = G * rows;
    const long nc = Workspace::chunks_for(count);
    const long ng = (nc + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(k_xs_from_sub, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, (const double *)ws.rec, count / kUnit,
                       nc, G, rows, ws.approx, ws.goff, (int)kUnit, 4, ws.skip);
    if (total_out) hipLaunchKernelGGL(k_xs_total, dim3(1), dim3(1024), 0, s, ws.goff, ng, total_out, ws.skip);
}

// Seeds ws.hyp[which] from the dot product whose approximate sums (span sums by rows) are in the workspace.
inline void spec_seed(hipStream_t s, const Workspace &ws, long G, long rows, int which)
{
    const long units = (G / kUnit) * rows;
    hipLaunchKernelGGL(k_sp_seed, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, (const double *)ws.sub, G, rows,
                       (const double *)ws.approx, (const double *)ws.goff, ws.hyp[which], ws.skip);
}

template <class Src>
inline void spec_summaries(hipStream_t s, const Workspace &ws, const Src &src, long G, long rows, int which, bool nonneg,
                           const double *rho_dev, double start, const double *seed_dev = nullptr, int nseed = 0)
{
    const long count = G * rows;
    const long nc = Workspace::chunks_for(count);
    const long ng = (nc + kGroup - 1) / kGroup;
    hipLaunchKernelGGL(k_xs_gscan, dim3(1), dim3(1024), 0, s, ws.goff, ng, start, seed_dev, nseed, ws.nfix, ws.skip);
    const long nwave = ((rows + 63) / 64) * (G / kChunk);
    hipLaunchKernelGGL(k_sp_select, dim3((unsigned)((nwave + 3) / 4)), dim3(256), 0, s, reinterpret_cast<const UnitRec *>(ws.rec),
                       nonneg ? 1 : 0, ws.hyp[which], 1.0, rho_dev, G, rows, ws.approx, ws.goff, ws.chunks, ws.fixlist, ws.nfix, ws.skip);
    hipLaunchKernelGGL(k_sp_fix<Src>, dim3(256), dim3(256), 0, s, src, count, ws.fixlist, ws.nfix, ws.approx, ws.goff, ws.chunks, ws.skip);
    enqueue_levels_and_plan(s, ws, src, count);
}

// The self-check of CG_OPT_STRICT_AUDIT for a dot product whose walk has just written `result`.
template <class Src>
inline void enqueue_audit(hipStream_t s, const Workspace &ws, const Src &src, long count, double start, const double *start_dev,
                          const double *result)
{
    const long nc = Workspace::chunks_for(count);
    const long ng = (nc + kGroup - 1) / kGroup;
    double *second = reinterpret_cast<double *>(ws.audit + 1);
    hipLaunchKernelGGL(k_xs_walk<Src>, dim3(1), dim3(64), 0, s, src, count, ws.chunks, nc, ws.groups, ng,
                       start, start_dev, second, (long *)nullptr, (const int *)nullptr, (const int *)nullptr,
                       (const int *)nullptr, (const double *)nullptr, ws.skip);
    hipLaunchKernelGGL(k_xs_audit<Src>, dim3(256), dim3(256), 0, s, src, count, ws.chunks, nc, ws.audit, ws.skip);
    hipLaunchKernelGGL(k_xs_audit_result, dim3(1), dim3(1), 0, s, result, second, ws.audit, ws.skip);
}

// One rank: all three in a row.
template <class Src>
inline void enqueue_spec(hipStream_t s, const Workspace &ws, const Src &src, long G, long rows, int which, bool nonneg,
                         const double *rho_dev, double start, double *result, bool audit = false)
{
    spec_approx(s, ws, G, rows);
    spec_summaries(s, ws, src, G, rows, which, nonneg, rho_dev, start);
    enqueue_walk(s, ws, src, G * rows, start, (const double *)nullptr, result, ws.stats + 8 * which);
    if (audit) enqueue_audit(s, ws, src, G * rows, start, (const double *)nullptr, result);
#ifdef CG_DEBUG_XS
    double *dbg = reinterpret_cast<double *>(ws.stats + 24);
    hipLaunchKernelGGL(k_xs_literal<Src>, dim3(1), dim3(64), 0, s, src, G * rows, start, dbg + 1);
    hipLaunchKernelGGL(k_xs_compare, dim3(1), dim3(1), 0, s, result, dbg + 1, dbg + 1, which);
#endif
}

} // namespace exact
} // namespace cg
