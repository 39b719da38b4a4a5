// nem_halfsum.hpp -- a d-ordered float chain  s <- RN24(s + x_d)  of NON-NEGATIVE MULTIPLES OF 1/2, evaluated by one
// wavefront in pieces instead of D dependent adds on one lane.
//
// InerToDispK_ (reference nem_mod.c:1054-1058) sums a class's inertia values over the organisms in index order into
// a float.  Under NCEM every value is a class count or half a class size (EstimLaplaceIner on one-hot memberships):
// a non-negative multiple of 1/2 below 2^24.  Work in half-units (h_d = 2 x_d, an integer; scaling by 2 commutes
// with every rounding here) and write S for the running float sum, P_i = h_0 + ... + h_i for the exact one.
//   * LEVEL j >= 1: S in [2^(23+j), 2^(24+j)) moves on the grid U = 2^j, S = M U with 2^23 <= M < 2^24; level 0:
//     S < 2^24, every add is exact (U = 1).
//   * A step at level j that stays below 2^(24+j): h = a U + r, 0 <= r < U, and
//         M <- M + a            (r < U/2)      M <- M + a + 1            (r > U/2)
//         M <- (M + a + 1) & ~1                (r = U/2: the tie goes to the even neighbour)
//     -- what is added depends on M through its parity only, so a run of such steps is a map M <- M + d[M & 1]; the
//     two increments are found by carrying an even and an odd significand through the run (two independent integer
//     chains of an add and a mask per step).
//   * Which level a step runs at follows from the EXACT prefix sums up to a bound: every step errs by at most half the
//     spacing of its result and no sum exceeds twice the exact total, so |S_i - P_i| <= E_i := (i + 1) Ucap with
//     Ucap = the spacing at P_n's level (induction: S_(t-1) + h_t <= P_t + t Ucap < 2 * 2^(24+L), L = P_n's level).
//     The steps i of a range [a, b) are CERTAIN at level j when
//         P_(a-1) - E_(b-1) >= 2^(23+j)  (no lower condition for j = 0)   and   P_(b-1) + E_(b-1) < 2^(24+j):
//     then every S_(i-1) lies in level j and every step's exact sum stays below the level's end (sums are monotone).
//     A range that is not certain -- one that holds a power of two the sum passes -- is taken with the reference's own
//     float adds.
// Procedure (W wavefronts of 64 lanes; thread t owns the contiguous elements [t e, (t+1) e), e = ceil(n / 64 W) made odd
// -- the lanes read their elements from LDS e floats apart --):
//   A  every thread sums its h; a scan over the lanes, then over the wavefronts' totals, gives each its exact prefix;
//      Ucap from the total;
//   B  every thread classifies its range and, if certain, walks it: (level, d[0], d[1]); else (first, count);
//   C  neighbouring certain ranges are at the same level (the one's last exact sum is the other's first), so a
//      segmented scan over the lanes of a wavefront -- six steps, segments = the runs of certain lanes between the
//      others -- composes every run into one map, and a wavefront without uncertain lanes into ONE map; what is left to
//      do in order (wavefront 0, all lanes carrying the same state) is, per wavefront: its map (an integer add on the
//      significand, kept as an integer while the level stays), or, for the few that hold a power of two the sum passes,
//      per uncertain lane: the map of the run before it, then its own float adds.
// Exactness does not depend on where the bound puts the stepped ranges, only the speed does (a chain that hovers around
// a power of two degrades to stepping).  halfsum_host() runs the same classification, walk, scan steps and piece
// application thread by thread on the CPU; tests/test_halfsum.py holds it (and the device procedure) against the plain
// loop.
#pragma once
#include <cstdint>
#include <cstring>

#include "nem_ff.hpp"

namespace nemk {

constexpr int kHsLanes = 64;
constexpr int kHsMaxWaves = 16;
NEMFF_HD inline int hs_segment(int n, int threads) { return ((n + threads - 1) / threads) | 1; }

NEMFF_HD inline int hs_log2_u64(unsigned long long v)      // floor(log2 v), v >= 1
{
#if defined(__HIP_DEVICE_COMPILE__)
    return 63 - __clzll((long long)v);
#else
    return 63 - __builtin_clzll(v);
#endif
}
// level of a value in half-units: 0 below 2^24, else floor(log2) - 23
NEMFF_HD inline int hs_level(unsigned long long v) { return v < (1ull << 24) ? 0 : hs_log2_u64(v) - 23; }

// exact half-units of a multiple of 1/2 in [0, 2^24)
NEMFF_HD inline uint32_t hs_half_units(float x) { return (uint32_t)(x + x); }

// step B, classification: the level at which every step of [a, b) certainly runs, or -1.
// Pa = exact sum before element a, Pb = through element b - 1 (half-units)
NEMFF_HD inline int hs_certain_level(int a, int b, unsigned long long Pa, unsigned long long Pb, unsigned long long ucap)
{
    if (a >= b) return -1;
    const unsigned long long E = (unsigned long long)b * ucap;      // E_(b-1)
    const int j = hs_level(Pb + E);
    if (j > 30) return -1;
    if (j == 0) return 0;
    return (Pa >= E && Pa - E >= (1ull << (23 + j))) ? j : -1;
}

// step B, walk: the two increments of a certain range at level j
NEMFF_HD inline void hs_map(const float* x, int a, int b, int j, uint32_t& d0, uint32_t& d1)
{
    const uint32_t rmask = (1u << j) - 1u, half = j ? (1u << (j - 1)) : 0xFFFFFFFFu;   // (level 0: r = 0, never a tie)
    uint32_t t = 0u, u = 1u;                                          // an even and an odd significand, relative
#pragma unroll 4
    for (int i = a; i < b; i++) {
        const uint32_t h = hs_half_units(x[i]);
        const uint32_t q = h >> j, r = h & rmask;
        const bool tie = r == half;
        const uint32_t add = q + ((tie || (j != 0 && r > half)) ? 1u : 0u);
        const uint32_t keep = tie ? ~1u : ~0u;
        t = (t + add) & keep;
        u = (u + add) & keep;
    }
    d0 = t; d1 = u - 1u;
}

// step C: the running sum, as a float or -- between map pieces of one level -- as its integer significand
struct HsState { float S; uint32_t M; int lev; };                     // lev < 0: S holds the value
NEMFF_HD inline void hs_to_float(HsState& st)
{
    if (st.lev < 0) return;
    float r = (float)st.M;                                            // M <= 2^24: exact
    if (st.lev > 0) { uint32_t bits; memcpy(&bits, &r, 4); bits += (uint32_t)st.lev << 23; memcpy(&r, &bits, 4); }
    st.S = r; st.lev = -1;
}
NEMFF_HD inline void hs_apply_map(HsState& st, int lev, uint32_t d0, uint32_t d1)
{
    if (st.lev != lev) {
        hs_to_float(st);
        // S is a multiple of 2^lev below 2^(24+lev) (certain at level lev): its significand as an integer
        if (lev == 0) st.M = (uint32_t)st.S;
        else { uint32_t bits; memcpy(&bits, &st.S, 4); st.M = (bits & 0x7FFFFFu) | 0x800000u; }   // exponent field 150 + lev
        st.lev = lev;
    }
    st.M += (st.M & 1u) ? d1 : d0;
}
NEMFF_HD inline void hs_apply_steps(HsState& st, const float* x, int first, int count)
{
    hs_to_float(st);
    float S = st.S;
    int t = 0;
    for (; t + 8 <= count; t += 8) {                                   // (the values ahead of the dependent adds)
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = x[first + t + e];
#pragma unroll
        for (int e = 0; e < 8; e++) S = S + (v[e] + v[e]);
    }
    for (; t < count; t++) { const float v = x[first + t]; S = S + (v + v); }
    st.S = S;
}

// first f, then g, both on significands of one level:  h[p] = f[p] + g[(p + f[p]) & 1]
NEMFF_HD inline void hs_compose(uint32_t f0, uint32_t f1, uint32_t g0, uint32_t g1, uint32_t& h0, uint32_t& h1)
{
    h0 = f0 + ((f0 & 1u) ? g1 : g0);
    h1 = f1 + (((1u + f1) & 1u) ? g1 : g0);
}
// one step (distance s) of the segmented scan, for the lane whose own values are (flag, lev, d0, d1) and whose
// neighbour s lanes below has (pflag, plev, p0, p1): the neighbour's map first, then the lane's own
NEMFF_HD inline void hs_scan_step(bool have, int& flag, int& lev, uint32_t& d0, uint32_t& d1, int pflag, int plev, uint32_t p0, uint32_t p1)
{
    if (!have || flag) return;
    uint32_t h0, h1;
    hs_compose(p0, p1, d0, d1, h0, h1);
    d0 = h0; d1 = h1;
    lev = plev > lev ? plev : lev;
    flag = pflag;
}

// what a thread leaves for the ordered pass: tag >= -1: a certain (or empty) range -- the run of such lanes that ends
// here is at level tag (-1: no elements in it) and is the map (x0, x1); tag = -2: a range to step, (first, count)
struct HsRecord { int tag; uint32_t x0, x1; };
// the ordered pass over `count` records (rec(l) for lane l); uncertain = bit l set when lane l is not a map: step(l)
// does what it stands for (a range's float adds; one level up: a wavefront's own ordered pass).  The same on the host
// and on the device.
template <class GetRec, class Step>
NEMFF_HD inline void hs_ordered_pass(HsState& st, unsigned long long uncertain, int count, GetRec&& rec, Step&& step)
{
    unsigned long long todo = uncertain;
    while (true) {
        int l = count;
        if (todo) {
#if defined(__HIP_DEVICE_COMPILE__)
            l = (int)__ffsll((long long)todo) - 1;
#else
            l = __builtin_ctzll(todo);
#endif
        }
        if (l > 0 && !((uncertain >> (l - 1)) & 1ull)) {
            const HsRecord r = rec(l - 1);
            if (r.tag >= 0) hs_apply_map(st, r.tag, r.x0, r.x1);         // the run of certain lanes before lane l
        }
        if (l == count) break;
        todo &= todo - 1ull;
        step(l);
    }
}

// the whole procedure, thread by thread, on the CPU (and the plain loop it must equal)
inline float halfsum_plain(const float* x, int n)
{
    float s = 0.0f;
    for (int i = 0; i < n; i++) s += x[i];
    return s;
}
inline float halfsum_host(const float* x, int n, int waves = kHsMaxWaves, int* n_stepped = nullptr)
{
    const int T = waves * kHsLanes, e = hs_segment(n, T);
    static thread_local unsigned long long P[kHsMaxWaves * kHsLanes + 1];
    static thread_local HsRecord R[kHsMaxWaves * kHsLanes];
    auto lo = [&](int t) { const long long v = (long long)t * e; return (int)(v < n ? v : n); };
    // A
    P[0] = 0ull;
    for (int t = 0; t < T; t++) {
        unsigned long long v = 0ull;
        for (int i = lo(t); i < lo(t + 1); i++) v += hs_half_units(x[i]);
        P[t + 1] = P[t] + v;
    }
    const unsigned long long ucap = 1ull << hs_level(P[T]);
    int stepped = 0;
    HsState st{0.0f, 0u, -1};
    unsigned long long unc_of[kHsMaxWaves];
    for (int w = 0; w < waves; w++) {
        HsRecord* W = R + w * kHsLanes;
        int flag[kHsLanes];
        unsigned long long uncertain = 0ull;
        // B
        for (int l = 0; l < kHsLanes; l++) {
            const int t = w * kHsLanes + l, a = lo(t), b = lo(t + 1);
            const int lev = hs_certain_level(a, b, P[t], P[t + 1], ucap);
            const bool unc = b > a && lev < 0;
            W[l].tag = lev; W[l].x0 = 0u; W[l].x1 = 0u;                    // (an empty range: the identity, level -1)
            if (lev >= 0) hs_map(x, a, b, lev, W[l].x0, W[l].x1);
            if (unc) { uncertain |= 1ull << l; stepped += b - a; }
            flag[l] = (l == 0 || unc) ? 1 : 0;
        }
        // C: the segmented scan (heads: lane 0 and the lanes to step, which carry the identity) ...
        for (int s = 1; s < kHsLanes; s <<= 1) {
            int nf[kHsLanes], nl[kHsLanes]; uint32_t n0[kHsLanes], n1[kHsLanes];
            for (int l = 0; l < kHsLanes; l++) {
                nf[l] = flag[l]; nl[l] = W[l].tag; n0[l] = W[l].x0; n1[l] = W[l].x1;
                if (l >= s) hs_scan_step(true, nf[l], nl[l], n0[l], n1[l], flag[l - s], W[l - s].tag, W[l - s].x0, W[l - s].x1);
            }
            for (int l = 0; l < kHsLanes; l++) { flag[l] = nf[l]; W[l].tag = nl[l]; W[l].x0 = n0[l]; W[l].x1 = n1[l]; }
        }
        for (int l = 0; l < kHsLanes; l++)
            if ((uncertain >> l) & 1ull) { const int t = w * kHsLanes + l; W[l].tag = -2; W[l].x0 = (uint32_t)lo(t); W[l].x1 = (uint32_t)(lo(t + 1) - lo(t)); }
        unc_of[w] = uncertain;
    }
    // ... a wavefront without lanes to step is the one map its last lane holds: the same scan over the wavefronts
    // (heads: wavefront 0 and those with lanes to step, which carry the identity) ...
    HsRecord S[kHsMaxWaves];
    int wflag[kHsMaxWaves];
    unsigned long long complex_waves = 0ull;
    for (int w = 0; w < waves; w++) {
        const bool cx = unc_of[w] != 0ull;
        if (cx) complex_waves |= 1ull << w;
        wflag[w] = (w == 0 || cx) ? 1 : 0;
        S[w] = cx ? HsRecord{-1, 0u, 0u} : R[w * kHsLanes + kHsLanes - 1];
    }
    for (int s = 1; s < waves; s <<= 1) {
        HsRecord N[kHsMaxWaves]; int nf[kHsMaxWaves];
        for (int w = 0; w < waves; w++) {
            N[w] = S[w]; nf[w] = wflag[w];
            if (w >= s) hs_scan_step(true, nf[w], N[w].tag, N[w].x0, N[w].x1, wflag[w - s], S[w - s].tag, S[w - s].x0, S[w - s].x1);
        }
        for (int w = 0; w < waves; w++) { S[w] = N[w]; wflag[w] = nf[w]; }
    }
    // ... and the ordered pass: per wavefront with lanes to step, the run of wavefronts before it, then its own pass
    hs_ordered_pass(st, complex_waves, waves, [&](int w) { return S[w]; }, [&](int w) {
        const HsRecord* W = R + w * kHsLanes;
        hs_ordered_pass(st, unc_of[w], kHsLanes, [&](int l) { return W[l]; },
                        [&](int l) { hs_apply_steps(st, x, (int)W[l].x0, (int)W[l].x1); });
    });
    hs_to_float(st);
    if (n_stepped) *n_stepped = stepped;
    return 0.5f * st.S;
}

#if defined(__HIPCC__)
// LDS a block needs for halfsum_block (besides the values themselves)
struct HsShared {
    unsigned long long wave_total[kHsMaxWaves];
    unsigned long long uncertain[kHsMaxWaves];
    HsRecord rec[kHsMaxWaves * kHsLanes];
    long long stamps[6];
};
// All 64 W threads of the block call it (W <= 16 wavefronts, tid = 0 .. 64 W - 1, x in LDS, n uniform; it contains
// block barriers).  The chain's value is returned to the threads of wavefront 0.
template <int W>
__device__ inline float halfsum_block(const float* x, int n, HsShared& sh, bool timing = false)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int e = hs_segment(n, W * kHsLanes);
    const long long lo_l = (long long)tid * e;
    const int a = (int)(lo_l < n ? lo_l : n), b = (int)(lo_l + e < n ? lo_l + e : n);
    if (timing && tid == 0) sh.stamps[0] = wall_clock64();
    // A: thread totals, scan over the lanes (two 32-bit shuffles per step), then over the wavefronts
    unsigned long long t = 0ull;
#pragma unroll 4
    for (int i = a; i < b; i++) t += hs_half_units(x[i]);
    unsigned long long inc = t;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t lo32 = (uint32_t)__shfl_up((int)(uint32_t)inc, s, 64);
        const uint32_t hi32 = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), s, 64);
        if (lane >= s) inc += ((unsigned long long)hi32 << 32) | lo32;
    }
    if (W > 1) {
        if (lane == 63) sh.wave_total[wv] = inc;
        __syncthreads();
    }
    unsigned long long before = 0ull, total = 0ull;
    if (W > 1) {
#pragma unroll
        for (int w = 0; w < W; w++) { const unsigned long long v = sh.wave_total[w]; if (w < wv) before += v; total += v; }
    } else {
        total = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(inc >> 32), 63, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)inc, 63, 64);
    }
    const unsigned long long ucap = 1ull << hs_level(total);
    if (timing && tid == 0) sh.stamps[1] = wall_clock64();
    // B: classify, walk
    const unsigned long long Pb = before + inc, Pa = Pb - t;
    const int lev = hs_certain_level(a, b, Pa, Pb, ucap);
    const bool unc = b > a && lev < 0;
    HsRecord r{lev, 0u, 0u};
    if (lev >= 0) hs_map(x, a, b, lev, r.x0, r.x1);
    if (timing && tid == 0) sh.stamps[2] = wall_clock64();
    // C: segmented scan over the lanes
    const unsigned long long uncertain = __ballot(unc);
    int flag = (lane == 0 || unc) ? 1 : 0;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int pflag = __shfl_up(flag, s, 64), plev = __shfl_up(r.tag, s, 64);
        const uint32_t p0 = (uint32_t)__shfl_up((int)r.x0, s, 64), p1 = (uint32_t)__shfl_up((int)r.x1, s, 64);
        hs_scan_step(lane >= s, flag, r.tag, r.x0, r.x1, pflag, plev, p0, p1);
    }
    if (unc) { r.tag = -2; r.x0 = (uint32_t)a; r.x1 = (uint32_t)(b - a); }
    if (W > 1) {
        // what the ordered pass reads: of a wavefront without lanes to step only its last lane's record
        if (uncertain != 0ull || lane == 63) sh.rec[tid] = r;
        if (lane == 0) sh.uncertain[wv] = uncertain;
        __syncthreads();
    }
    if (timing && tid == 0) sh.stamps[3] = wall_clock64();
    float out = 0.0f;
    if (wv == 0) {
        HsState st{0.0f, 0u, -1};
        auto lane_rec = [](const HsRecord& v, int l) {
            HsRecord q;
            q.tag = __builtin_amdgcn_readlane(v.tag, l);
            q.x0 = (uint32_t)__builtin_amdgcn_readlane((int)v.x0, l); q.x1 = (uint32_t)__builtin_amdgcn_readlane((int)v.x1, l);
            return q;
        };
        if (W > 1) {
            // lane w < W holds wavefront w's mask and last record; the same scan over those lanes (heads: wavefront 0
            // and the wavefronts with lanes to step, which carry the identity)
            const int wl = lane < W ? lane : 0;
            const unsigned long long um = lane < W ? sh.uncertain[wl] : 0ull;
            HsRecord sum = sh.rec[wl * kHsLanes + 63];
            const bool cx = um != 0ull;
            const unsigned long long complex_waves = __ballot(cx);
            if (cx || lane >= W) sum = HsRecord{-1, 0u, 0u};
            int wflag = (lane == 0 || cx) ? 1 : 0;
#pragma unroll
            for (int s = 1; s < W; s <<= 1) {
                const int pflag = __shfl_up(wflag, s, 64), plev = __shfl_up(sum.tag, s, 64);
                const uint32_t p0 = (uint32_t)__shfl_up((int)sum.x0, s, 64), p1 = (uint32_t)__shfl_up((int)sum.x1, s, 64);
                hs_scan_step(lane >= s, wflag, sum.tag, sum.x0, sum.x1, pflag, plev, p0, p1);
            }
            const int um_lo = (int)(uint32_t)um, um_hi = (int)(uint32_t)(um >> 32);
            hs_ordered_pass(st, complex_waves, W, [&](int w) { return lane_rec(sum, w); }, [&](int w) {
                // a wavefront with lanes to step: its 64 records, one per lane
                const unsigned long long u = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(um_hi, w) << 32) |
                                             (uint32_t)__builtin_amdgcn_readlane(um_lo, w);
                const HsRecord mine = sh.rec[w * kHsLanes + lane];
                hs_ordered_pass(st, u, kHsLanes, [&](int l) { return lane_rec(mine, l); }, [&](int l) {
                    const HsRecord q = lane_rec(mine, l);
                    hs_apply_steps(st, x, (int)q.x0, (int)q.x1);
                });
            });
        } else {
            hs_ordered_pass(st, uncertain, kHsLanes, [&](int l) { return lane_rec(r, l); }, [&](int l) {
                const HsRecord q = lane_rec(r, l);
                hs_apply_steps(st, x, (int)q.x0, (int)q.x1);
            });
        }
        hs_to_float(st);
        out = 0.5f * st.S;
    }
    if (timing && tid == 0) sh.stamps[4] = wall_clock64();
    return out;
}
#endif

}  // namespace nemk
