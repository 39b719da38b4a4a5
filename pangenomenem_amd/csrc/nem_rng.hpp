// nem_rng.hpp -- the generator behind the reference's random draws, restated.
//
// The reference draws its random starts (MakeRandomPara, nem_alg.c:1381-1473, through RandomInteger,
// nem_rnd.c:40-63) from libc random() after srandom(NemPara.Seed) (nem_exe.c:621; the seed is time(NULL),
// nem_exe.c:353).  On glibc that is the TYPE_3 additive-feedback generator: r[i] = r[i-3] + r[i-31] over a
// 34-word state filled by the Lehmer recurrence r[i] = 16807 * r[i-1] mod (2^31 - 1), the first 310 outputs
// discarded, each output shifted right by one.  Restating it (instead of calling random()) keeps this library
// free of process-global state -- the reason the reference's nem() is not re-entrant -- and lets a caller who
// fixes the seed reproduce the reference's starts draw for draw (tests/test_capi.py checks it against libc).
#pragma once
#include <cstdint>

namespace nemk {

class GlibcRandom {
public:
    explicit GlibcRandom(uint32_t seed) { this->seed(seed); }
    void seed(uint32_t s)
    {
        if (s == 0) s = 1;                                   // srandom(0) behaves as srandom(1)
        int32_t word = (int32_t)s;
        r_[0] = (uint32_t)word;
        for (int i = 1; i < 31; i++) {
            // word = 16807 * word mod 2147483647 without overflow (Schrage)
            const int32_t hi = word / 127773, lo = word % 127773;
            word = 16807 * lo - 2836 * hi;
            if (word < 0) word += 2147483647;
            r_[i] = (uint32_t)word;
        }
        f_ = 3; b_ = 0;
        for (int i = 0; i < 310; i++) (void)next();
    }
    // one random(): 0 .. 2^31 - 1
    long next()
    {
        r_[f_] += r_[b_];
        const uint32_t out = r_[f_] >> 1;
        f_ = (f_ + 1 == 31) ? 0 : f_ + 1;
        b_ = (b_ + 1 == 31) ? 0 : b_ + 1;
        return (long)out;
    }
    // RandomInteger(mini, maxi), nem_rnd.c:40-63: no draw when mini >= maxi
    int integer(int mini, int maxi)
    {
        if (mini >= maxi) return maxi;
        const int span = maxi - mini + 1;
        return (int)(next() % span) + mini;
    }

private:
    uint32_t r_[31];
    int f_ = 3, b_ = 0;
};

}  // namespace nemk
