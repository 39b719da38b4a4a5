// Exact parallel evaluation of the reference's sequential float accumulators.
//
// The criteria of ComputeCrit (nem_alg.c:2727-2745) are sums of the form
//     acc = (float)((double)acc + x_i),  i = 0 .. n-1,  acc a float, x_i a double
// (a plain `float += float` is the same thing: a double holds 2*24+2 bits, so rounding the exact sum of two floats
// to double first is innocuous).  Float addition does not associate, so the sum cannot be split -- but between two
// powers of two the accumulator moves on a FIXED grid, and there the chain is integer arithmetic:
//
//   let 2^e <= |acc| < 2^(e+1), u = 2^(e-23) (acc = s*M*u, 2^23 <= M < 2^24), and x = s*(a + r)*u with a >= 0 an
//   integer and 0 <= r < 1 (the sum grows in magnitude).  The exact sum is s*(M + a + r)*u; rounding it to double
//   moves it by at most 2^-30 u, rounding that to float picks a multiple of u (of 2u from 2^(e+1) on, where M + a
//   is 2^24: the two grids agree on it).  So unless r is within 2^-28 of 1/2,
//       M <- M + a + [r > 1/2]
//   whatever M is: the increments are independent of the state, a prefix sum of integers.
//
// A segment therefore ends only where (1) M would pass 2^24 (the next binade: at most ~30 times per chain),
// (2) r is (nearly) a tie -- then the parity of M and the double rounding decide -- or (3) x shrinks the sum, is not
// finite, or is out of range.  Those single steps are taken with the reference's own arithmetic (`step`).
// Everything here is shared by the device kernels (nem_kernels.hip) and the host emulation the CPU tests pin
// against the plain sequential loop (nemgpu_chain_host).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#ifndef NEM_HD
#if defined(__HIPCC__)
#define NEM_HD __host__ __device__
#else
#define NEM_HD
#endif
#endif

namespace nemchain {

constexpr long long kTop = 1ll << 24;

NEM_HD inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
NEM_HD inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// one step of the reference chain
NEM_HD inline float step(float acc, double x) { return (float)((double)acc + x); }

// can the chain leave `acc` in integer form?  (normal, finite, and room for one more binade)
NEM_HD inline bool ready(float acc, int& E, int& neg)
{
    const uint32_t b = f2u(acc);
    E = (int)((b >> 23) & 255u);
    neg = (int)(b >> 31);
    return E >= 1 && E <= 253;
}

// an accumulator no finite addend can move: +-inf (until `poison` arrives: a NaN or the opposite infinity) and NaN
NEM_HD inline bool absorbing(float acc) { return ((f2u(acc) >> 23) & 255u) == 255u; }
NEM_HD inline bool poison(float inf_acc, double x) { return x != x || (x - x != 0.0 && (x < 0.0) != (inf_acc < 0.0f)); }

// s / u as a double: (+-) 2^(150 - E), a normal double for every float exponent field E
NEM_HD inline double scale(int E, int neg)
{
    const uint64_t bits = ((uint64_t)neg << 63) | ((uint64_t)(1023 + 150 - E) << 52);
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

NEM_HD inline long long mantissa(float acc) { return (long long)((f2u(acc) & 0x7fffffu) | 0x800000u); }

// the float s * M * u for 2^23 <= M <= 2^24
NEM_HD inline float compose(long long M, int E, int neg)
{
    const uint32_t b = ((uint32_t)neg << 31) | (M >= kTop ? ((uint32_t)(E + 1) << 23) : (((uint32_t)E << 23) | ((uint32_t)M & 0x7fffffu)));
    return u2f(b);
}

// the state-independent increment of x (in units of u, signed like acc), or false when this step has to be taken
// with the reference's own arithmetic
NEM_HD inline bool increment(double x, double sc, long long& inc)
{
    const double q = x * sc;                             // exact: a power-of-two scaling (an overflow gives inf, a
    if (q == 0.0) { inc = 0; return true; }              //  result below 2^-1022 rounds, far from 1/2 either way)
    if (!(q > 0.0) || !(q < 33554432.0)) return false;   // shrinking sum, NaN, or 2^25 and more
    const double fl = floor(q);
    const double fr = q - fl;                            // exact
    if (fabs(fr - 0.5) <= 0x1p-28) return false;         // (near) tie
    inc = (long long)fl + (fr > 0.5 ? 1 : 0);
    return true;
}

// Sequential steps taken in a row after an element the integer form could not take: 16, doubling up to 256 while
// the integer form advances by fewer than 64 elements between two such events (the first binades of a sum, or ties
// on a coarse grid), back to 16 otherwise; an accumulator the integer form cannot leave at all (zero, subnormal,
// subnormal) quadruples it up to a whole window; an infinite or NaN accumulator skips to the value that changes it (see
// `absorbing`).  The result does not depend on it -- both forms are exact --
// only the time does.
constexpr int kBurst = 16, kBurstTies = 256, kBurstMax = 4096;
NEM_HD inline int next_burst(int burst, long long advanced)
{
    if (advanced < 0) return burst * 4 < kBurstMax ? burst * 4 : kBurstMax;          // accumulator not ready
    if (advanced < 64) return burst >= kBurstTies ? burst : burst * 2;
    return kBurst;
}

// Host emulation of the device procedure (same decisions in the same order; `lanes` x `per` elements per pass).
inline float run_segmented(const double* x, long long n, float acc, int lanes = 1024, int per = 4)
{
    const long long window = (long long)lanes * per;
    int burst = kBurst;
    for (long long base = 0; base < n; base += window) {
        const long long wn = (n - base < window) ? (n - base) : window;
        const double* xs = x + base;
        long long wpos = 0;
        while (wpos < wn) {
            int E, neg;
            if (!ready(acc, E, neg)) {
                if (absorbing(acc)) {
                    // an infinite accumulator stays what it is until a NaN or the opposite infinity arrives, a NaN for
                    // good: the steps in between change nothing and are not taken
                    long long p = wn;
                    if (!(acc != acc)) for (long long q = wpos; q < wn; q++) if (poison(acc, xs[q])) { p = q; break; }
                    if (p < wn) acc = step(acc, xs[p++]);
                    wpos = p;
                    continue;
                }
                for (int b = 0; b < burst && wpos < wn; b++) acc = step(acc, xs[wpos++]);
                burst = next_burst(burst, -1);
                continue;
            }
            const double sc = scale(E, neg);
            long long M = mantissa(acc);
            long long p = wpos;
            bool stopped = false;
            for (; p < wn; p++) {
                long long inc;
                if (M >= kTop || !increment(xs[p], sc, inc) || M + inc > kTop) { stopped = true; break; }
                M += inc;
            }
            acc = compose(M, E, neg);
            const long long advanced = p - wpos;
            wpos = p;
            if (stopped) for (int b = 0; b < burst && wpos < wn; b++) acc = step(acc, xs[wpos++]);
            burst = next_burst(burst, advanced);
        }
    }
    return acc;
}

inline float run_sequential(const double* x, long long n, float acc)
{
    for (long long i = 0; i < n; i++) acc = step(acc, x[i]);
    return acc;
}

}  // namespace nemchain
