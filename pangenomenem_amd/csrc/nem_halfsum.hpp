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
//         M <- M + a + ((M + a) & 1)           (r = U/2: the tie goes to the even neighbour)
//     i.e. M <- M + d[M & 1] for two increments d[0], d[1] that do not depend on M otherwise.  Such maps compose into
//     maps of the same form:  (g o f)[p] = f[p] + g[(p + f[p]) & 1].
//   * Which level a step runs at follows from the EXACT prefix sums up to a bound: every step errs by at most half the
//     spacing of its result, the sums are monotone, so |S_i - P_i| <= (i + 1) Ucap / 2 <= E_i := (i + 1) Ucap with
//     Ucap = the spacing at P_n's level (S_i <= 2 P_i: its level is at most one above P_i's).  Step i is CERTAIN at level j when
//         P_(i-1) - E_(i-1) >= 2^(23+j)  (no lower condition for j = 0)   and   P_i + E_i < 2^(24+j):
//     then S_(i-1) lies in level j and the step's exact sum stays below the level's end.  All other steps (a few around
//     every power of two the sum passes) are taken with the reference's own float add.
// Procedure (64 lanes, lane l owns the contiguous elements [l m, (l+1) m), m = ceil(n / 64) made odd -- the lanes read
// their elements from LDS m floats apart --):
//   A  every lane sums its h; an exclusive scan over the lanes gives each its exact prefix; Ucap from the total;
//   B  every lane walks its elements once: runs of certain steps of one level become (level, d[0], d[1]) -- both
//      parities carried along, two independent integer chains --, runs of other steps become (first, count); a lane
//      keeps at most kHsPieces pieces (more: the whole segment is stepped);
//   C  the pieces are applied in lane order to S: a map is an integer add on the significand, a stepped run is its
//      float adds.
// Exactness does not depend on where the bound puts the uncertain zones, only the speed does (a chain of tiny addends
// degrades to stepping).  halfsum_host() runs the same walk and the same piece application lane by lane on the CPU;
// tests/test_halfsum.py holds it (and the device procedure) against the plain loop.
#pragma once
#include <cstdint>
#include <cstring>

#include "nem_ff.hpp"

namespace nemk {

constexpr int kHsLanes = 64;
constexpr int kHsPieces = 4;
NEMFF_HD inline int hs_segment(int n) { return ((n + kHsLanes - 1) / kHsLanes) | 1; }
enum { HS_NONE = 0, HS_MAP = 1, HS_STEPS = 2 };

struct HsLane {
    int np;
    int kind[kHsPieces];
    uint32_t a[kHsPieces];        // map: d[0]; steps: first element
    uint32_t b[kHsPieces];        // map: d[1]; steps: count
    int lev[kHsPieces];           // map: level
};

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

// step B for one lane: elements [lo, hi) of x, `base` = exact half-unit sum of the elements before lo
NEMFF_HD inline void hs_walk(const float* x, int lo, int hi, unsigned long long base, unsigned long long ucap, HsLane& L)
{
    L.np = 0;
#pragma unroll
    for (int q = 0; q < kHsPieces; q++) { L.kind[q] = HS_NONE; L.a[q] = 0u; L.b[q] = 0u; L.lev[q] = 0; }
    if (lo >= hi) return;
    int kind = HS_NONE, lev = 0;
    uint32_t pa = 0u, pb = 0u;
    bool overflow = false;
    auto flush = [&]() {
        if (kind == HS_NONE) return;
        if (L.np >= kHsPieces) { overflow = true; return; }
#pragma unroll
        for (int q = 0; q < kHsPieces; q++)
            if (q == L.np) { L.kind[q] = kind; L.a[q] = pa; L.b[q] = pb; L.lev[q] = lev; }
        L.np++;
    };
    unsigned long long P = base;
    unsigned long long E = (unsigned long long)lo * ucap;           // E_(i-1) = i * Ucap
    unsigned long long low_t = 0ull, up_t = 0ull;                   // the current map piece's level: [2^(23+j), 2^(24+j))
    for (int i = lo; i < hi; i++) {
        const uint32_t h = hs_half_units(x[i]);
        const unsigned long long Pprev = P, Eprev = E;
        P += h; E += ucap;
        bool certain;
        int j = lev;
        if (kind == HS_MAP && Pprev >= low_t + Eprev && P + E < up_t) certain = true;      // (the common case: same level)
        else {
            const int j_hi = hs_level(P + E);
            const int j_lo = Pprev >= Eprev ? hs_level(Pprev - Eprev) : 0;
            certain = j_hi == j_lo && (j_lo == 0 || Pprev - Eprev >= (1ull << (23 + j_lo))) && j_lo <= 30;
            j = j_lo;
        }
        if (certain) {
            if (!(kind == HS_MAP && j == lev)) {
                flush();
                kind = HS_MAP; lev = j; pa = 0u; pb = 0u;
                low_t = j ? (1ull << (23 + j)) : 0ull; up_t = 1ull << (24 + j);
            }
            const uint32_t a = h >> j, r = h & ((1u << j) - 1u), half = j ? (1u << (j - 1)) : 0u;
            const bool tie = j != 0 && r == half;
            const uint32_t c = (j != 0 && r > half) ? 1u : 0u;
            uint32_t t0 = pa + a, t1 = pb + a;                       // significand parity after + a: t0 & 1, (t1 + 1) & 1
            t0 += tie ? (t0 & 1u) : c;
            t1 += tie ? ((t1 + 1u) & 1u) : c;
            pa = t0; pb = t1;
        } else {
            if (kind == HS_STEPS) pb++;
            else { flush(); kind = HS_STEPS; lev = 0; pa = (uint32_t)i; pb = 1u; }
        }
    }
    flush();
    if (overflow) {                                                  // too many pieces: step the whole segment
        L.np = 1;
        L.kind[0] = HS_STEPS; L.a[0] = (uint32_t)lo; L.b[0] = (uint32_t)(hi - lo); L.lev[0] = 0;
#pragma unroll
        for (int q = 1; q < kHsPieces; q++) L.kind[q] = HS_NONE;
    }
}

// step C: one piece applied to the running float sum S (half-units)
NEMFF_HD inline float hs_apply(float S, int kind, uint32_t a, uint32_t b, int lev, const float* x)
{
    if (kind == HS_MAP) {
        // S is a multiple of 2^lev with S / 2^lev < 2^24 (certain at level lev): the significand as an integer
        uint32_t bits;
        memcpy(&bits, &S, 4);
        uint32_t M;
        if (lev == 0) M = (uint32_t)S;
        else M = (bits & 0x7FFFFFu) | 0x800000u;                     // exponent field = 150 + lev
        M += (M & 1u) ? b : a;
        float r = (float)M;                                          // M <= 2^24: exact
        if (lev) { memcpy(&bits, &r, 4); bits += (uint32_t)lev << 23; memcpy(&r, &bits, 4); }
        return r;
    }
    if (kind == HS_STEPS) {
        for (uint32_t t = 0; t < b; t++) { const float v = x[a + t]; S = S + (v + v); }
        return S;
    }
    return S;
}

// the whole procedure, lane by lane, on the CPU (and the plain loop it must equal)
inline float halfsum_plain(const float* x, int n)
{
    float s = 0.0f;
    for (int i = 0; i < n; i++) s += x[i];
    return s;
}
inline float halfsum_host(const float* x, int n, int* n_stepped = nullptr)
{
    const int m = hs_segment(n);
    unsigned long long base[kHsLanes + 1];
    base[0] = 0ull;
    for (int l = 0; l < kHsLanes; l++) {
        unsigned long long t = 0ull;
        for (int i = l * m; i < n && i < (l + 1) * m; i++) t += hs_half_units(x[i]);
        base[l + 1] = base[l] + t;
    }
    const unsigned long long ucap = 1ull << hs_level(base[kHsLanes]);
    float S = 0.0f;
    int stepped = 0;
    for (int l = 0; l < kHsLanes; l++) {
        HsLane L;
        const int lo = l * m < n ? l * m : n, hi = (l + 1) * m < n ? (l + 1) * m : n;
        hs_walk(x, lo, hi, base[l], ucap, L);
        for (int q = 0; q < L.np; q++) {
            S = hs_apply(S, L.kind[q], L.a[q], L.b[q], L.lev[q], x);
            if (L.kind[q] == HS_STEPS) stepped += (int)L.b[q];
        }
    }
    if (n_stepped) *n_stepped = stepped;
    return 0.5f * S;
}

#if defined(__HIPCC__)
// One wavefront (all 64 lanes call it, x in LDS or global memory, n uniform): returns the chain's value in every lane.
__device__ inline float halfsum_wave(const float* x, int n)
{
    const int lane = threadIdx.x & 63;
    const int m = hs_segment(n);
    const int lo = min(lane * m, n), hi = min(lane * m + m, n);
    // A: lane totals, exclusive scan over the lanes (two 32-bit shuffles per step)
    unsigned long long t = 0ull;
    for (int i = lo; i < hi; i++) t += hs_half_units(x[i]);
    unsigned long long inc = t;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const uint32_t lo32 = (uint32_t)__shfl_up((int)(uint32_t)inc, s, 64);
        const uint32_t hi32 = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), s, 64);
        if (lane >= s) inc += ((unsigned long long)hi32 << 32) | lo32;
    }
    const unsigned long long base = inc - t;
    const unsigned long long total = ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(inc >> 32), 63, 64) << 32) |
                                     (uint32_t)__shfl((int)(uint32_t)inc, 63, 64);
    const unsigned long long ucap = 1ull << hs_level(total);
    // B
    HsLane L;
    hs_walk(x, lo, hi, base, ucap, L);
    // C: the pieces in lane order (wave-uniform: every lane carries the same S)
    float S = 0.0f;
    for (int l = 0; l < kHsLanes; l++) {
        const int np = __builtin_amdgcn_readlane(L.np, l);
#pragma unroll
        for (int q = 0; q < kHsPieces; q++) {
            if (q < np) {
                const int kind = __builtin_amdgcn_readlane(L.kind[q], l);
                const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)L.a[q], l);
                const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)L.b[q], l);
                const int lev = __builtin_amdgcn_readlane(L.lev[q], l);
                S = hs_apply(S, kind, a, b, lev, x);
            }
        }
    }
    return 0.5f * S;
}
#endif

}  // namespace nemk
