// nem_ff.hpp -- exact "fast-forward" of the Bernoulli log-density chain inside one float binade.
//
// The chain of DensBernoulli (reference nem_mod.c:661) is, per organism,
//     dk <- RN24( RN53( RN53(dk + t) + b0 ) ),   t = 0 (match) or A (mismatch),
// with A = log((1-eps)/eps), b0 = -log(1-eps) doubles (uniform dispersion inside the class), RN53 / RN24 =
// round-to-nearest-even to double / float.  While dk stays inside ONE float binade [2^e, 2^(e+1)) and A, b0 >= 0:
//   * dk is a multiple M*u of the float spacing u = 2^(e-23); doubles in the same binade have spacing
//     v = 2^(e-52) = u * 2^-29, and dk / v = M * 2^29 is an even integer;
//   * RN53(dk + A) = dk + a*v with a = RNE(A / v): the tie rule sees the parity of dk/v + floor(A/v), i.e. of
//     floor(A/v) alone;
//   * RN53(dk + a*v + b0) = dk + (a + b)*v with b = RNE(b0 / v), where an exact tie of b0/v is broken towards an
//     even a + b (the parity of the accumulated multiple, not of b);
//   * RN24(dk + r*v) = dk + RNE(r / 2^29) * u, the same integer for every M -- unless r / 2^29 is an exact tie,
//     in which case the result depends on the parity of M and the binade is marked unusable.
// So inside the binade one organism adds a constant q0 (match) or q1 (mismatch) to the float's BIT PATTERN, and
// a run of n organisms with p mismatches adds (n-p)*q0 + p*q1 -- an integer multiply-add and a popcount
// instead of n dependent fma/add/cvt/cvt groups.  All intermediate sums are monotone (A, b0 >= 0), so if the
// pattern after the run is still below the binade's end, every step of the run was inside the binade and the
// identity holds step by step; the step that reaches or passes the end is redone with the reference's own
// arithmetic.  Nothing here is approximate: tests/test_fastforward.py checks the table against step-by-step
// float/double arithmetic on the CPU, tests/test_gpu_parity.py the kernels against the oracle.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define NEMFF_HD __host__ __device__
#else
#define NEMFF_HD
#endif

namespace nemk {

// q value that makes a step "leave the binade" (pattern + 2^23 >= next exponent): forces exact stepping.
// q0 and q1 are usable independently; all sums are taken modulo 2^32 and their true values stay below 2^32
// (pattern < 2^31, at most 32 increments of at most 2^23), so q1 - q0 may wrap without harm.
constexpr uint32_t kFFInvalid = 1u << 23;

// Increments of the float bit pattern for exponent field E (value in [2^(E-127), 2^(E-126))).
NEMFF_HD inline void ff_entry(double A, double b0, int E, uint32_t& q0, uint32_t& q1)
{
    q0 = q1 = kFFInvalid;
    if (E < 1 || E > 253) return;                       // zero / denormals / top binade: always exact stepping
    if (!(A >= 0.0) || !(b0 >= 0.0)) return;            // monotone chains only (also rejects NaN)
    const int s = 179 - E;                              // x / v = x * 2^s
    const double lim = 4503599627370496.0;              // 2^52: beyond it the increment is >= 2^23 anyway
    const double xa = ldexp(A, s), xb = ldexp(b0, s);   // exact scalings (A, b0 are 0 or far from under/overflow)
    if (!(xb < lim)) return;
    const double fb = floor(xb);
    const double frac = xb - fb;                        // exact (xb < 2^52)
    const int64_t bfl = (int64_t)fb;
    const int64_t half = (int64_t)1 << 28, mask = ((int64_t)1 << 29) - 1;
    // match step: one rounding on the even base dk / v
    const int64_t r0 = (frac > 0.5) ? bfl + 1 : (frac < 0.5) ? bfl : ((bfl & 1) ? bfl + 1 : bfl);
    if ((r0 & mask) != half) {                          // (an exact float tie would depend on the parity of M)
        const int64_t Q0 = (r0 + half) >> 29;
        if (Q0 < (int64_t)kFFInvalid) q0 = (uint32_t)Q0;
    }
    // mismatch step: a = RNE(A / v) on the even base, then b0 on base dk/v + a (parity of a)
    if (!(xa < lim)) return;
    const int64_t ai = (int64_t)rint(xa);
    const int64_t b1 = (frac > 0.5) ? bfl + 1 : (frac < 0.5) ? bfl : ((((ai + bfl) & 1) != 0) ? bfl + 1 : bfl);
    const int64_t r1 = ai + b1;
    if ((r1 & mask) != half) {
        const int64_t Q1 = (r1 + half) >> 29;
        if (Q1 < (int64_t)kFFInvalid) q1 = (uint32_t)Q1;
    }
}

// s = 0; repeat `times`: s = RN24(s + x) -- the N_KD chain of InerToDispK_ (reference nem_mod.c:1054-1058: the class
// size added once per organism) -- without the `times` dependent adds.  Inside one binade [2^(e-1), 2^e) with float
// spacing U the sum is m * U and a step adds the same integer c = RNE(x / U) to m every time; when x / U is an exact
// tie the rounding goes to the even neighbour, which leaves m even, and from an even m every tie step adds q rounded
// to even (q = floor(x / U)).  So a binade costs one division instead of its steps; the step that reaches or passes
// the end of the binade -- the only one whose rounding grid differs -- is a real float add.  All quantities are
// integers below 2^53 held in doubles (x / U is an exact power-of-two scaling).  x > 0 finite; anything else takes
// the plain loop.  tests/test_fastforward.py checks it against the loop.
NEMFF_HD inline float ff_repeat_add(float x, long long times)
{
    float s = 0.0f;
    if (!(x > 0.0f) || !(x <= 3.402823466e+38f)) {          // zero, negative, NaN, inf: nothing to gain, few steps matter
        for (long long j = 0; j < times && j < 4; j++) s += x;   // (0, NaN and inf are fixed points after one step ...
        if (x < 0.0f) for (long long j = 4; j < times; j++) s += x;   //  ... a negative x is not: the plain loop)
        return s;
    }
    long long left = times;
    while (left > 0) {
        if (!(s <= 3.402823466e+38f)) return s;                 // overflowed: inf + x = inf
        int e = 0;
        (void)frexp((double)s, &e);                             // s in [2^(e-1), 2^e)
        if (s == 0.0f || e - 24 < -149 + 1) { s += x; left--; continue; }   // zero / subnormal spacing: step for real
        const double U = ldexp(1.0, e - 24);
        double m = (double)s / U;                               // integer in [2^23, 2^24)
        const double xq = (double)x / U, q = floor(xq), fr = xq - q;
        double c;
        if (fr > 0.5) c = q + 1.0;
        else if (fr < 0.5) c = q;
        else {
            if (fmod(m, 2.0) != 0.0) { s += x; left--; continue; }   // a tie step from an odd m: for real (m becomes even)
            c = q + fmod(q, 2.0);
        }
        if (c == 0.0) return s;                                 // x vanishes against s: nothing changes any more
        double j = floor((16777215.0 - m) / c);                  // steps that stay below 2^24
        if (j > (double)left) j = (double)left;
        if (j > 0.0) { m += j * c; left -= (long long)j; s = (float)(m * U); }
        if (left > 0) { s += x; left--; }                       // the step that reaches the end of the binade
    }
    return s;
}

// The same chain for an integer addend 0 < H < 2^24 (an NCEM class size), in integer arithmetic: exact adds while the
// sum stays at or below 2^24, then per binade the constant increment of the 24-bit significand -- one 32-bit
// division per binade instead of its steps, no double-precision library calls (ff_repeat_add's frexp / ldexp / fmod
// cost as much as a thousand dependent adds on a lone GPU lane).
NEMFF_HD inline float ff_repeat_add_u24(uint32_t H, long long times)
{
    if (times <= 0 || H == 0u) return 0.0f;
    const uint32_t L = 1u << 24;
    long long j0 = (long long)(L / H);
    if (j0 > times) j0 = times;
    float s = (float)((uint32_t)j0 * H);                     // <= 2^24: exact
    long long left = times - j0;
    const float x = (float)H;
    while (left > 0) {
        uint32_t bits;
        memcpy(&bits, &s, 4);
        const int E = (int)((bits >> 23) & 255u);
        const int sh = E - 150;                                 // s = m * 2^sh, m in [2^23, 2^24)
        if (sh < 0 || E == 255) { s = s + x; left--; continue; }   // (cannot happen for a sum that left the exact range)
        if (sh >= 25) return s;                                 // H < 2^24 <= half a spacing: nothing changes any more
        uint32_t m = (bits & 0x7FFFFFu) | 0x800000u;
        const uint32_t q = H >> sh, r = H & ((1u << sh) - 1u), half = sh ? (1u << (sh - 1)) : 0u;
        uint32_t c;
        if (sh == 0 || r < half) c = q;
        else if (r > half) c = q + 1u;
        else {
            if (m & 1u) { s = s + x; left--; continue; }        // a tie step from an odd significand: for real (it becomes even)
            c = q + (q & 1u);
        }
        if (c == 0u) return s;
        long long j = (long long)((L - 1u - m) / c);            // steps that stay inside the binade
        if (j > left) j = left;
        if (j > 0) {
            m += (uint32_t)j * c;
            left -= j;
            bits = ((uint32_t)E << 23) | (m & 0x7FFFFFu);
            memcpy(&s, &bits, 4);
        }
        if (left > 0) { s = s + x; left--; }                    // the step that reaches the end of the binade
    }
    return s;
}

}  // namespace nemk
