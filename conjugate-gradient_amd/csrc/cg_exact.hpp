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
// Failures are rare: each binade crossing of the accumulator (a few dozen per dot
// product), the leading zeros, and ties (~2^-29 per element).  The margins of one ulp
// at both ends of the binade make the integer model exact: a running value of at
// least 2^52+1 means the unrounded sum was at least 2^52+1/2, still inside the binade.
// Every step is either exact integer arithmetic in fp64 (|values| < 2^53) or the
// literal recurrence, so the result is bit-identical to the CPU loop for ANY input.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>

namespace cg {
namespace exact {

constexpr int kChunk = 1024;          // elements per chunk = one wavefront x 16
constexpr int kPerLane = kChunk / 64;
constexpr int kGroup = 64;            // chunks per group = one wavefront of summaries
constexpr int kAnyExp = INT_MAX;      // summary valid under every hypothesis (all addends are zero)
constexpr int kNoExp = INT_MIN;       // no usable guess

typedef double xv2 __attribute__((ext_vector_type(2)));

struct Summary {
    double D;         // sum of the rounded addends, in ulps of the hypothesised binade
    double mn, mx;    // extremes of the running value of D (the empty prefix, 0, included)
    int e;            // hypothesised exponent of the accumulator, or kAnyExp
    int bad;          // the summary must not be used
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
    r.bad = a.bad | b.bad | ((a.e != kAnyExp && b.e != kAnyExp && a.e != b.e) ? 1 : 0);
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

// Exponent of a finite non-zero normal double, else kNoExp.
__device__ __forceinline__ int exponent_of(double s)
{
    if (!(fabs(s) >= 0x1p-1021) || !(fabs(s) <= 0x1.fffffffffffffp+1023)) return kNoExp;
    return ilogb(s);
}

// Summary of kPerLane consecutive addends under the hypothesis "exponent e".
// The per-element validity tests are folded into two running maxima and checked once:
// |x| < 2^51 for every addend <=> max|x| < 2^51; no tie <=> max|x - rint(x)| != 0.5 (it never
// exceeds 0.5); a NaN addend poisons D (v_max would skip it), which the final D == D catches.
__device__ __forceinline__ Summary piece_summary(const double (&p)[kPerLane], int e)
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
    s.bad = (nan || !(amax < 0x1p51) || fmaxv == 0.5 || e == kNoExp) ? 1 : 0;
    s.e = e;
    return s;
}

// Ordered composition over the wavefront of summaries that share ONE hypothesis e (or are
// identities): only D, mn, mx travel; flags and the exponent are combined by ballots.
__device__ __forceinline__ Summary wave_compose_same_e(Summary cur, int e)
{
    const int lane = threadIdx.x & 63;
    const bool any_bad = __ballot(cur.bad != 0) != 0ull;
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
__global__ __launch_bounds__(256) void k_xs_from_sub(const double *__restrict__ sub, long nsub, long nchunks,
                                                     long G, long rows, double *__restrict__ approx)
{
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= nchunks) return;
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < kChunk / kSub; ++j) {
        const long i = c * (kChunk / kSub) + j;
        if (i >= nsub) break;
        if (G > 0) {
            const long k = i * kSub, m = k / G, col = k - m * G;
            acc += sub[(col / kSub) * rows + m];
        } else {
            acc += sub[i];
        }
    }
    approx[c] = acc;
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
// given, with the device-resident value it points to).  One workgroup.
__global__ __launch_bounds__(1024) void k_xs_gscan(double *__restrict__ gsum, long ngroups, double start,
                                                   const double *start_dev)
{
    if (start_dev != nullptr) start = *start_dev;
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

// ---- B: per-chunk integer summaries under the guessed exponent --------------------
// Persistent wavefronts: each walks its own run of chunks and keeps the NEXT chunk's products in
// flight while it summarises the current one.
template <bool HAS_W>
__global__ __launch_bounds__(256) void k_xs_chunk(const double *__restrict__ v, const double *__restrict__ w,
                                                  long count, long nchunks, int vec, const double *__restrict__ approx,
                                                  const double *__restrict__ goff, Summary *__restrict__ out)
{
    // Products are fetched coalesced (lane-interleaved) and handed to the lanes as 16
    // CONSECUTIVE elements each through LDS; 18-double rows keep the 16-byte reads of the
    // 16 lanes of a group on distinct banks.
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

        const Summary s = wave_compose_same_e(piece_summary(p, e), e);
        if (lane == 0) out[c] = s;
    }
}

// ---- C: groups of 64 chunk summaries -------------------------------------------------
__global__ __launch_bounds__(256) void k_xs_group(const Summary *__restrict__ in, long nin, Summary *__restrict__ out,
                                                  long nout)
{
    const int lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= nout) return;
    const long i = g * kGroup + lane;
    Summary s = (i < nin) ? in[i] : identity();
    s = wave_compose(s);
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
__device__ __forceinline__ double replay(const double *v, const double *w, long lo, long hi, double s, int lane)
{
    for (long base = lo; base < hi; base += 64) {
        const long i = base + lane;
        const double p = (i < hi) ? product(v, w, i) : 0.0;
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

// Walk up to 64 summaries held one per lane (`mine`; lanes >= n hold nothing), in order,
// starting at index `from` from accumulator s.  Returns the index of the first summary
// that could not be applied (n if all were) and leaves s as the accumulator just before it.
__device__ __forceinline__ int walk(Summary mine, int n, int from, double &s, int lane)
{
    const int es = exponent_of(s);
    const double S = (es == kNoExp) ? 0.0 : ldexp(s, 52 - es);
    const bool neg_zero = (s == 0.0) && signbit(s);
    Summary cur = (lane >= from && lane < n) ? mine : identity();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {        // inclusive ordered scan
        const Summary prev = shfl_up(cur, off);
        if (lane >= off) cur = compose(prev, cur);
    }
    const bool ok = holds(cur, es, S, neg_zero);
    // validity is monotone along the scan (flags accumulate, extremes only widen)
    const unsigned long long okmask = __ballot(ok);
    const unsigned long long upto = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    const unsigned long long want = upto & ~((1ull << from) - 1ull);
    const unsigned long long failed = want & ~okmask;
    const int f = failed ? (int)__builtin_ctzll(failed) : n;
    if (f > from) {
        const double D = __shfl(cur.D, f - 1, 64);
        const int ce = __shfl(cur.e, f - 1, 64);
        if (ce != kAnyExp) s = ldexp(S + D, es - 52);   // exact: integers below 2^53
    }
    return f;
}

// A chunk whose stored summary does not apply: cut it into 64 pieces of 16 elements (one per
// lane), summarise the pieces under the accumulator's ACTUAL exponent, apply the longest
// valid run, replay the one offending piece literally, and repeat with the new exponent.
__device__ __forceinline__ double refine(const double *v, const double *w, long lo, long hi, double s, int lane)
{
    double p[kPerLane];
#pragma unroll
    for (int i = 0; i < kPerLane; ++i) {
        const long idx = lo + (long)lane * kPerLane + i;
        p[i] = (idx < hi) ? product(v, w, idx) : 0.0;
    }
    int piece = 0;
    for (int round = 0; piece < 64; ++round) {
        if (round >= 8)                              // many ties or crossings: not worth summarising
            return replay(v, w, lo + (long)piece * kPerLane, hi, s, lane);
        const Summary mine = piece_summary(p, exponent_of(s));
        const int f = walk(mine, 64, piece, s, lane);
        if (f >= 64) break;
        const long first = lo + (long)f * kPerLane;
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
            const double pi = __shfl(p[i], f, 64);
            if (first + i < hi) s += pi;
        }
        piece = f + 1;
    }
    return s;
}

// One group (64 chunk summaries) starting from accumulator s.
__device__ __forceinline__ double walk_group(const double *v, const double *w, long count, const Summary *chunks,
                                             long nchunks, long g, double s, int lane)
{
    const long cb = g * kGroup;
    const int nc = (nchunks - cb < 64) ? (int)(nchunks - cb) : 64;
    const Summary cmine = (lane < nc) ? chunks[cb + lane] : identity();
    int cpos = 0;
    while (cpos < nc) {
        const int cf = walk(cmine, nc, cpos, s, lane);
        if (cf >= nc) break;
        const long lo = (cb + cf) * kChunk;
        const long hi = (lo + kChunk < count) ? lo + kChunk : count;
        s = refine(v, w, lo, hi, s, lane);
        cpos = cf + 1;
    }
    return s;
}

// Three levels: 64 super-groups (of 64 groups of 64 chunks) are scanned per step; where a
// summary does not apply the walk descends one level, down to the elements.
__global__ __launch_bounds__(64) void k_xs_walk(const double *__restrict__ v, const double *__restrict__ w, long count,
                                                const Summary *__restrict__ chunks, long nchunks,
                                                const Summary *__restrict__ groups, long ngroups,
                                                const Summary *__restrict__ supers, long nsupers, double start,
                                                const double *start_dev, double *result)
{
    const int lane = threadIdx.x;
    double s = (start_dev != nullptr) ? *start_dev : start;
    for (long sb = 0; sb < nsupers; sb += 64) {
        const int ns = (nsupers - sb < 64) ? (int)(nsupers - sb) : 64;
        const Summary smine = (lane < ns) ? supers[sb + lane] : identity();
        int spos = 0;
        while (spos < ns) {
            const int sf = walk(smine, ns, spos, s, lane);
            if (sf >= ns) break;
            // super-group sb+sf does not apply as a whole: walk its groups
            const long gb = (sb + sf) * kGroup;
            const int ng = (ngroups - gb < 64) ? (int)(ngroups - gb) : 64;
            const Summary gmine = (lane < ng) ? groups[gb + lane] : identity();
            int gpos = 0;
            while (gpos < ng) {
                const int gf = walk(gmine, ng, gpos, s, lane);
                if (gf >= ng) break;
                s = walk_group(v, w, count, chunks, nchunks, gb + gf, s, lane);
                gpos = gf + 1;
            }
            spos = sf + 1;
        }
    }
    if (lane == 0) *result = s;
}

// Workspace for `count` elements.
struct Workspace {
    double *approx = nullptr, *goff = nullptr;
    double *sub = nullptr;                // span sums handed over by the producing kernel (kSub products each)
    Summary *chunks = nullptr, *groups = nullptr, *supers = nullptr;
    long cap_chunks = 0;

    static long chunks_for(long count) { return (count + kChunk - 1) / kChunk; }
    hipError_t reserve(long count)
    {
        const long nc = chunks_for(count);
        if (nc <= cap_chunks) return hipSuccess;
        release();
        const long ng = (nc + kGroup - 1) / kGroup, nsup = (ng + kGroup - 1) / kGroup;
        hipError_t e;
        if ((e = hipMalloc(&sub, (size_t)nc * (kChunk / kSub) * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&supers, (size_t)nsup * sizeof(Summary))) != hipSuccess) return e;
        if ((e = hipMalloc(&approx, (size_t)nc * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&goff, (size_t)ng * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&chunks, (size_t)nc * sizeof(Summary))) != hipSuccess) return e;
        if ((e = hipMalloc(&groups, (size_t)ng * sizeof(Summary))) != hipSuccess) return e;
        cap_chunks = nc;
        return hipSuccess;
    }
    void release()
    {
        (void)hipFree(approx); (void)hipFree(goff); (void)hipFree(chunks); (void)hipFree(groups);
        (void)hipFree(sub); (void)hipFree(supers);
        approx = nullptr; goff = nullptr; chunks = nullptr; groups = nullptr; sub = nullptr; supers = nullptr;
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
inline void enqueue(hipStream_t s, const Workspace &ws, const double *v, const double *w, long count, double start,
                    double *result, const double *start_dev = nullptr, bool have_sub = false, long sub_G = 0,
                    long sub_rows = 0)
{
    if (count <= 0) {
        hipLaunchKernelGGL(k_xs_walk, dim3(1), dim3(64), 0, s, v, w, 0L, ws.chunks, 0L, ws.groups, 0L, ws.supers, 0L,
                           start, start_dev, result);
        return;
    }
    const long nc = Workspace::chunks_for(count);
    const long ng = (nc + kGroup - 1) / kGroup, nsup = (ng + kGroup - 1) / kGroup;
    const unsigned cb = (unsigned)((nc + 3) / 4), gb = (unsigned)((ng + 3) / 4), sb = (unsigned)((nsup + 3) / 4);
    const int vec = ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(w)) & 15) == 0 ? 1 : 0;   // (null w: v alone)
    if (have_sub)
        hipLaunchKernelGGL(k_xs_from_sub, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, ws.sub,
                           (count + kSub - 1) / kSub, nc, sub_G, sub_rows, ws.approx);
    else
        hipLaunchKernelGGL(k_xs_approx, dim3(cb), dim3(256), 0, s, v, w, count, nc, vec, ws.approx);
    hipLaunchKernelGGL(k_xs_gsum, dim3(gb), dim3(256), 0, s, ws.approx, nc, ws.goff, ng);
    hipLaunchKernelGGL(k_xs_gscan, dim3(1), dim3(1024), 0, s, ws.goff, ng, start, start_dev);
    const unsigned cbp = cb < 1024u ? cb : 1024u;          // persistent: the four workgroups per CU that LDS allows
    if (w != nullptr)
        hipLaunchKernelGGL(k_xs_chunk<true>, dim3(cbp), dim3(256), 0, s, v, w, count, nc, vec, ws.approx, ws.goff, ws.chunks);
    else
        hipLaunchKernelGGL(k_xs_chunk<false>, dim3(cbp), dim3(256), 0, s, v, w, count, nc, vec, ws.approx, ws.goff, ws.chunks);
    hipLaunchKernelGGL(k_xs_group, dim3(gb), dim3(256), 0, s, ws.chunks, nc, ws.groups, ng);
    hipLaunchKernelGGL(k_xs_group, dim3(sb), dim3(256), 0, s, ws.groups, ng, ws.supers, nsup);
    hipLaunchKernelGGL(k_xs_walk, dim3(1), dim3(64), 0, s, v, w, count, ws.chunks, nc, ws.groups, ng, ws.supers, nsup,
                       start, start_dev, result);
}

} // namespace exact
} // namespace cg
