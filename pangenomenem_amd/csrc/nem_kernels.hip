// nem_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the NEM hot path.
//
// Reference stages (paths under /root/reference/ppanggolin/NEM/):
//   E1  ComputePkFkiM + DensBernoulli      nem_alg.c:2234-2289, nem_mod.c:619-690
//   E2  ComputePartitionNEM / ComputeLocalProba / ComputeMAP   nem_alg.c:2330-2405, 2546-2616, 590-664
//   M   EstimPara (Bernoulli = Laplace estimator)              nem_mod.c:415-469, 1180-1479, 1646-1704, 922-1174
//   C1  ComputeCrit                         nem_alg.c:2678-2757
//
// Arithmetic contract: every floating-point chain keeps the reference's operand types and
// order.  The file is compiled with -ffp-contract=off (no FMA contraction), no fast-math,
// denormals preserved.  Where a sum is order-independent (NCEM: c in {0,1} => integer
// counts < 2^24) it is computed with popcounts over bit-packed rows instead.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "nem_ff.hpp"
#include "nem_chain.hpp"
#include "nem_halfsum.hpp"
#include "nem_kernels.hpp"
#include "nem_sweep_dev.hpp"

namespace nemk {

// development probe (-DNEM_PHASE_PROF): block 0 / thread 0 stamps the 100 MHz wall clock at phase boundaries
#ifdef NEM_PHASE_PROF
__device__ unsigned long long g_phase[32];
#define NEM_PHASE(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase[i] = wall_clock64(); } while (0)
#else
#define NEM_PHASE(i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------

// NCEM statistic `idx`, summed over the ranks' partial arrays (integer sums: any order is exact)
__device__ __forceinline__ int stat_sum(const int* __restrict__ stats, int idx, int ranks, int stride)
{
    int s = stats[idx];
    for (int r = 1; r < ranks; r++) s += stats[idx + r * stride];
    return s;
}

// Sum over the 64 lanes, in every lane.  Six data-parallel-primitive adds (no LDS round trips: a __shfl_down
// reduction is six ds_bpermute latencies, ~0.3 us at the tail of kernels that last a few microseconds) and a
// readlane of lane 63, where the row_bcast steps leave the total.
__device__ inline int wave_reduce_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror: every lane holds its row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// the same for a non-negative 64-bit sum below 2^52: two 32-bit reductions of its low 20 and its high bits
__device__ inline long long wave_reduce_add_ll(long long v)
{
    const int lo = wave_reduce_add((int)(v & 0xFFFFF));
    const int hi = wave_reduce_add((int)(v >> 20));
    return ((long long)hi << 20) + (long long)lo;
}

// s + x0 + x1 + ... + x15, left to right, as sixteen v_add_f32 back to back: a d-ordered float chain on one lane is bound
// by what stands between two dependent adds, and left to itself the compiler puts a register copy there (it moves every
// loaded value out of the way of the next load: 1.5 instructions per add, 4.8 ns per organism)
__device__ __forceinline__ float chain_add16(float s, const float4& a, const float4& b, const float4& c, const float4& d)
{
    asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %0, %0, %4\n\t"
                 "v_add_f32 %0, %0, %5\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %0, %0, %7\n\tv_add_f32 %0, %0, %8\n\t"
                 "v_add_f32 %0, %0, %9\n\tv_add_f32 %0, %0, %10\n\tv_add_f32 %0, %0, %11\n\tv_add_f32 %0, %0, %12\n\t"
                 "v_add_f32 %0, %0, %13\n\tv_add_f32 %0, %0, %14\n\tv_add_f32 %0, %0, %15\n\tv_add_f32 %0, %0, %16"
                 : "+v"(s)
                 : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b.x), "v"(b.y), "v"(b.z), "v"(b.w),
                   "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w), "v"(d.x), "v"(d.y), "v"(d.z), "v"(d.w));
    return s;
}

// ------------------------------------------------------------------------------------------
// layout kernels (one-off, at upload time)
//   xf  [n][W]      family-major bit rows (host layout)
//   xw  [W][npad]   word-major: lane i reads word w of family i  -> coalesced E1 reads
//   xt  [d][nw64]   organism-major bit rows (bit j = local family j) -> popcount M-step
// ------------------------------------------------------------------------------------------
// words: block (bx, w) -- lane i <- word w of family i (xw) and of family perm[i] (xws, E1's sorted copy), one read pass
__device__ __forceinline__ void layout_words_body(const LayoutArgs& a, int bx)
{
    const int i = bx * 256 + (int)threadIdx.x;
    const int w = blockIdx.y;                                      // 0 .. 4 * ceil(W / 4) - 1
    if (i >= a.npad) return;
    uint32_t v = 0, vs = 0;
    if (i < a.n && w < a.wf) { v = a.xf[(size_t)i * a.wf + w]; vs = a.xf[(size_t)a.perm[i] * a.wf + w]; }
    if (w < a.W) a.xw[(size_t)w * a.npad + i] = v;
    a.xws[((size_t)(w >> 2) * a.npad + i) * 4 + (w & 3)] = vs;     // uint4[W4][npad]: words 4g..4g+3 of lane i together
}

// one wave = 64 families x one 32-organism word: 32 ballots transpose the 64x32 bit tile
__device__ __forceinline__ void layout_bits_body(const LayoutArgs& a, int bx)
{
    const int lane = threadIdx.x & 63;
    const int wave = (bx * 256 + (int)threadIdx.x) >> 6;           // 64-family block index
    const int w = blockIdx.y;
    if (wave >= a.nw64) return;
    const int i = wave * 64 + lane;
    const uint32_t v = (i < a.npad) ? a.xw[(size_t)w * a.npad + i] : 0u;
    uint64_t mine = 0;
#pragma unroll
    for (int b = 0; b < 32; b++) {
        const uint64_t m = __ballot((v >> b) & 1u);
        if (lane == b) mine = m;
    }
    const int org = w * 32 + lane;
    if (lane < 32 && org < a.d) a.xt[(size_t)org * a.nw64 + wave] = mine;
}

// ------------------------------------------------------------------------------------------
// per-(k,d) density tables.  DensBernoulli's two logs depend on (k,d) only (nem_mod.c:660-661):
//   t0 = |(int)(0 - mu)| * log((double)(float)((1-eps)/eps)),  t1 = same for x = 1,
//   l0 = log((double)(float)(1-eps));  eps <= EPSILON => t0 = t1 = l0 = 0 (dk unchanged) and a
//   "null density" bit when the mismatch is non-zero (nem_mod.c:662-666).
// Padding organisms (d >= D) get all-zero entries: the chain step is then an exact no-op.
// ------------------------------------------------------------------------------------------
constexpr int kDensMaskWords = 1024;   // class mismatch masks staged in LDS by k_density (2 x 4 KB): D <= 32768

// does organism dd force class k onto the general (per-organism constants) chain?  cheap: no logs
__device__ inline void table_flag_general(const FinishArgs& a, int t)
{
    const int D = a.D, dpad = a.dpad;
    const int k = t / dpad, dd = t - k * dpad;
    if (dd >= D) return;
    const float eps = a.disp[k * D + dd];
    const float mu = a.center[k * D + dd];
    const int ad0 = abs((int)(0.0f - mu)), ad1 = abs((int)(1.0f - mu));
    bool general = (ad0 > 1) || (ad1 > 1) || (__float_as_uint(eps) != __float_as_uint(a.disp[k * D])) ||
                   !((double)eps > kEpsilonD) || (dpad >> 5) > kDensMaskWords;   // (the uniform chain keeps its masks in LDS)
    if (!general && dd == 0) {
        const double l1 = log((double)((1.0f - eps) / eps)), l0 = log((double)(1.0f - eps));
        if (!isfinite(l1) || !isfinite(l0)) general = true;       // 0 * inf / NaN must propagate as in the reference
    }
    if (general) a.nonuni[k] = 1;
}

// one table entry t = k * dpad + dd; called by whole waves (dpad % 64 == 0).  Classes that stay on the
// uniform chain (nonuni[k] == 0, decided by table_flag_general) need the mismatch masks and uni[k] only.
__device__ inline void table_entry(const FinishArgs& a, int t)
{
    const int K = a.K, D = a.D, dpad = a.dpad;
    const int lane = threadIdx.x & 63;
    (void)K;
    const int k = t / dpad, dd = t - k * dpad;
    double t0 = 0.0, t1 = 0.0, l0 = 0.0;
    int n0 = 0, n1 = 0, a0 = 0, a1 = 0;
    if (dd < D) {
        const float eps = a.disp[k * D + dd];
        const float mu = a.center[k * D + dd];
        const int ad0 = abs((int)(0.0f - mu));
        const int ad1 = abs((int)(1.0f - mu));
        a0 = (ad0 != 0); a1 = (ad1 != 0);
        const bool need_tables = a.nonuni[k] != 0;
        if ((double)eps > kEpsilonD) {
            if (need_tables || dd == 0) {
                const double l1 = log((double)((1.0f - eps) / eps));
                l0 = log((double)(1.0f - eps));
                t0 = (double)ad0 * l1;
                t1 = (double)ad1 * l1;
                if (dd == 0) a.uni[k] = make_double2(l1, l0);
            }
        } else {
            n0 = a0; n1 = a1;
        }
    }
    if (dd >= D || a.nonuni[k] != 0) {                   // (padding entries are always written: zeros)
        a.tabT[t] = make_double2(t0, t1);
        a.tabL0[t] = l0;
    }
    const uint64_t m0 = __ballot(n0), m1 = __ballot(n1), b0 = __ballot(a0), b1 = __ballot(a1);
    if (lane == 0) {
        const int w = t >> 5;                            // word index inside [K][dpad/32]
        a.nz0[w] = (uint32_t)m0; a.nz0[w + 1] = (uint32_t)(m0 >> 32);
        a.nz1[w] = (uint32_t)m1; a.nz1[w + 1] = (uint32_t)(m1 >> 32);
        a.am0[w] = (uint32_t)b0; a.am0[w + 1] = (uint32_t)(b0 >> 32);
        a.am1[w] = (uint32_t)b1; a.am1[w + 1] = (uint32_t)(b1 >> 32);
    }
    if (dd == 0) {                                       // ComputePkFkiM, nem_alg.c:2262-2271
        const double p = (double)a.prop[k];
        a.pk[k] = p;
        if (p > kEpsilonD) a.logpk[k] = (float)log(p);
        else { a.logpk[k] = -INFINITY; atomicOr(&a.flags[FLAG_EMPTY_PROP], 1); }
    }
}

// ------------------------------------------------------------------------------------------
// E1: Bernoulli log-density chains.  One lane per (family, class): the chain over organisms is
// a sequential float accumulator with double intermediates and cannot be re-associated
// (SURVEY.md §0-2).  grid = (ceil(n/256), K); the class's table slice is staged through LDS in
// chunks of DCH organisms and read back as wave-uniform broadcasts.
// ------------------------------------------------------------------------------------------
constexpr int DCH = 512;    // organisms per general-path table chunk (small: LDS per block bounds the occupancy)

// XCD-aware tile mapping for the density kernels (1-D grid of ceil(tiles/8)*8*K blocks).  Blocks are dealt
// round-robin over the 8 XCDs, so blocks b, b+8, b+16, ... share an XCD and its L2: the K class-blocks of one
// family tile are placed 8 apart and pull the tile's matrix words through ONE L2 instead of K
// (speed only: any placement computes the same thing).
__device__ __forceinline__ bool density_tile(int K, int ntiles, int& tile, int& k)
{
    const int b = blockIdx.x;
    const int g = b / (8 * K), r = b - g * (8 * K);
    k = r >> 3;
    tile = g * 8 + (r & 7);
    return tile < ntiles;
}

// One organism of the uniform chain: dk <- (float)(((double)dk + m_b * l1) - l0)  (nem_mod.c:661), m_b = bit b
// of the mismatch word m.  m_b * l1 is formed inside an fma as {2.0 | 0.0} * (l1 / 2): the product is exact
// (a power-of-two scaling of l1, including the reference's 0 * l1 = -0 for negative l1), so the fma rounds
// once, exactly where the reference's add does.  The 2.0 / 0.0 factor is the mismatch bit moved to bit 30 of
// a double's high word -- a shift and an and instead of a compare and two 64-bit selects (6 VALU
// instructions per organism instead of 9; the chain itself stays fma, add, cvt, cvt).
__device__ __forceinline__ float bern_step(float dk, uint32_t m, int b, double l1h, double l0)
{
    const uint32_t hi = ((b <= 30) ? (m << (30 - b)) : (m >> 1)) & 0x40000000u;
    const double f = __hiloint2double((int)hi, 0);
    return (float)(fma(f, l1h, (double)dk) - l0);
}

// The uniform chain of one (family, class) lane.  am[w] = {am0, am1}: mismatch masks of the class per 32-organism word.
__device__ __forceinline__ uint32_t word_of(const uint4& v, int c)
{
    return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w;
}

// xw4: E1's matrix copy, uint4[W4][npad]: lane i reads words 4g .. 4g+3 of its family with one 16-byte load
// (1 KB per wave and request), two groups ahead of the one being consumed.
// (x & am1) | (~x & am0) is one bit-field insert
__device__ __forceinline__ uint32_t mismatch_word(uint32_t x, const uint2 am)
{
    uint32_t m;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(m) : "v"(x), "v"(am.y), "v"(am.x));
    return m;
}

__device__ __forceinline__ float chain_plain(const uint4* __restrict__ xw4, int npad, int i, int D,
                                    const uint2* am, double l1h, double l0)
{
    float dk = 0.0f;
    const int wlast = (D - 1) >> 5;                      // padding organisms must not take a step here
    const int glast = wlast >> 2;
    uint4 xn = xw4[i];
    for (int g = 0; g <= glast; g++) {
        const uint4 xv = xn;
        if (g < glast) xn = xw4[(size_t)(g + 1) * npad + i];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int w = 4 * g + c;
            if (w > wlast) break;
            const uint32_t m = mismatch_word(word_of(xv, c), am[w]);
            const int nb = (w < wlast) ? 32 : (D - (wlast << 5));
            if (nb == 32) {
#pragma unroll
                for (int b = 0; b < 32; b++) dk = bern_step(dk, m, b, l1h, l0);
            } else {
                for (int b = 0; b < nb; b++) dk = bern_step(dk, m, b, l1h, l0);
            }
        }
    }
    return dk;
}

// Same chain, fast-forwarded inside float binades (nem_ff.hpp): per 32-organism word ONE popcount and an integer
// multiply-add on the accumulator's bit pattern; a lane whose accumulator would leave its binade inside the
// word finds the last organism that still fits (5-step binary search over prefix popcounts), takes the
// boundary step with the reference's own arithmetic, reloads the increments of the new binade and goes on.
// sQ0 / sQ1: the class's 256-entry increment tables (LDS).  Bit-identical to chain_plain.
constexpr int kFFPlainWords = 1;
#ifndef NEM_FF_PLAIN_STEPS
#define NEM_FF_PLAIN_STEPS 32
#endif
constexpr int kFFPlainSteps = NEM_FF_PLAIN_STEPS;      // organisms of the first word that are stepped (the rest of it is fast-forwarded)

// sQ0[E] = q0, sDQ[E] = q1 - q0 (both at most 2^23: 24-bit multiplies), see ff_build
__device__ __forceinline__ void ff_load(const uint32_t* sQ0, const uint32_t* sDQ, uint32_t bits, uint32_t& q0, uint32_t& dq,
                                        uint32_t& end)
{
    const uint32_t E = (bits >> 23) & 255u;
    q0 = sQ0[E];
    dq = sDQ[E];
    end = (E + 1u) << 23;
}

// The integer side of the fast-forward is written as instructions: left to itself the compiler loses the 24-bit
// range of the increments (they travel through loop-carried registers) and emits v_mul_lo_u32 / v_mad_u64_u32 --
// quarter-rate multiplies -- where ONE full-rate v_mad_u32_u24 does.
// a * b + c on the low 24 bits of a and b
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// popc(m) * dq + base
__device__ __forceinline__ uint32_t ff_cand(uint32_t m, uint32_t dq, uint32_t base)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, 0\n\tv_mad_u32_u24 %0, %0, %2, %3" : "=&v"(r) : "v"(m), "v"(dq), "v"(base));
    return r;
}
// popc(low t bits of m) * dq + base
__device__ __forceinline__ uint32_t ff_probe(uint32_t m, uint32_t t, uint32_t dq, uint32_t base)
{
    uint32_t r;
    asm("v_bfe_u32 %0, %1, 0, %2\n\tv_bcnt_u32_b32 %0, %0, 0\n\tv_mad_u32_u24 %0, %0, %3, %4"
        : "=&v"(r) : "v"(m), "v"(t), "v"(dq), "v"(base));
    return r;
}

// the boundary organism (bit j of mm, j not a compile-time constant) with the reference's own arithmetic
__device__ __forceinline__ uint32_t ff_exact_step(uint32_t bits, uint32_t mm, uint32_t j, double l1h, double l0)
{
    const uint32_t hi = ((mm >> j) & 1u) << 30;          // 2.0 if the organism mismatches, else 0.0 (see bern_step)
    const double f = __hiloint2double((int)hi, 0);
    return __float_as_uint((float)(fma(f, l1h, (double)__uint_as_float(bits)) - l0));
}

// one word (nb organisms, mismatch bits m) of the fast-forwarded chain
__device__ __forceinline__ void ff_word(uint32_t& bits, uint32_t& q0, uint32_t& dq, uint32_t& end, uint32_t m, int nb,
                                        const uint32_t* sQ0, const uint32_t* sDQ, double l1h, double l0)
{
    // pattern after the nb organisms of which popc(m) mismatch: bits + nb*q0 + popc(m)*(q1 - q0)  (stays below
    // 2^32: pattern < 2^31, at most 32 increments of at most 2^23)
    const uint32_t cand = ff_cand(m, dq, (nb == 32) ? bits + (q0 << 5) : mad24((uint32_t)nb, q0, bits));
    const bool cross = cand >= end;
    bits = cross ? bits : cand;
    if (cross) {                                         // (a wave none of whose lanes cross skips this)
        uint32_t mm = m;
        int rem = nb;
        for (;;) {
            // organisms of this word that still fit in the binade: binary search over j, the pattern after the j
            // matching ones (at = bits + j * q0) carried along, so that a probe costs a shift-add and ONE multiply
            uint32_t j = 0, at = bits;
#pragma unroll
            for (int sh = 4; sh >= 0; sh--) {
                const uint32_t t = j + (1u << sh);
                const uint32_t at_t = at + (q0 << sh);
                const bool fits = ff_probe(mm, t, dq, at_t) < end;   // (f(t) >= f(rem) >= end for t >= rem: never accepted)
                j = fits ? t : j;
                at = fits ? at_t : at;
            }
            bits = ff_probe(mm, j, dq, at);
            bits = ff_exact_step(bits, mm, j, l1h, l0);
            ff_load(sQ0, sDQ, bits, q0, dq, end);        // the increments of the binade the step landed in
            rem -= (int)j + 1;
            if (rem <= 0) break;
            mm = mm >> (j + 1u);                         // j + 1 <= 31 here
            const uint32_t c2 = ff_cand(mm, dq, mad24((uint32_t)rem, q0, bits));
            if (c2 < end) { bits = c2; break; }          // (stays inside the binade just loaded)
        }
    }
}

// four full words (128 organisms) of the fast-forwarded chain
__device__ __forceinline__ void ff_group(const uint4& xv, int g, const uint2* am, uint32_t& bits,
                                         uint32_t& q0, uint32_t& q1, uint32_t& end, const uint32_t* sQ0,
                                         const uint32_t* sQ1, double l1h, double l0)   // (q1 / sQ1 carry q1 - q0)
{
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int w = 4 * g + c;
        const uint32_t m = mismatch_word(word_of(xv, c), am[w]);
        if (c < kFFPlainWords && g == 0) {
            // the first organisms run through small binades (a crossing every few steps): stepping is cheaper
            float dk0 = __uint_as_float(bits);
#pragma unroll
            for (int b = 0; b < kFFPlainSteps; b++) dk0 = bern_step(dk0, m, b, l1h, l0);
            bits = __float_as_uint(dk0);
            ff_load(sQ0, sQ1, bits, q0, q1, end);
            if (kFFPlainSteps < 32) ff_word(bits, q0, q1, end, m >> kFFPlainSteps, 32 - kFFPlainSteps, sQ0, sQ1, l1h, l0);
        } else {
            ff_word(bits, q0, q1, end, m, 32, sQ0, sQ1, l1h, l0);
        }
    }
}

__device__ __forceinline__ float chain_ff(const uint4* __restrict__ xw4, int npad, int i, int D,
                                          const uint2* am, const uint32_t* sQ0,
                                          const uint32_t* sQ1, double l1h, double l0)
{
    const int wlast = (D - 1) >> 5;
    const int gfull = wlast >> 2;                        // groups 0 .. gfull-1 hold four full words each
    uint32_t bits = 0u;                                  // +0.0f (exponent field 0 is always stepped exactly)
    uint32_t q0, q1, end;
    ff_load(sQ0, sQ1, bits, q0, q1, end);
    // four register buffers, refilled right after use: three loads stay in flight while a group is consumed.
    // Every load is issued unconditionally (past the last group it simply repeats the last one) so that the
    // compiler can count outstanding loads exactly and wait for the oldest one only.  The groups are read in
    // order through ONE running pointer (a 64-bit add per load; indexing would cost three quarter-rate multiplies).
    const uint4* pn = xw4 + i;
    int gi = 0;                                          // group *pn points at (wave-uniform)
    auto next = [&]() {
        const uint4 v = *pn;
        if (gi < gfull) { pn += npad; gi++; }
        return v;
    };
    uint4 x0 = next(), x1 = next(), x2 = next(), x3 = next();
    int g = 0;
    for (; g + 4 <= gfull; g += 4) {
        ff_group(x0, g, am, bits, q0, q1, end, sQ0, sQ1, l1h, l0);
        x0 = next();
        ff_group(x1, g + 1, am, bits, q0, q1, end, sQ0, sQ1, l1h, l0);
        x1 = next();
        ff_group(x2, g + 2, am, bits, q0, q1, end, sQ0, sQ1, l1h, l0);
        x2 = next();
        ff_group(x3, g + 3, am, bits, q0, q1, end, sQ0, sQ1, l1h, l0);
        x3 = next();
    }
    // 0..3 full groups left, then the last group (1..4 words, the last one possibly partial): after the loop
    // x0, x1, x2, x3 hold groups g, g+1, g+2, g+3 (clamped to gfull)
    uint4 xt = x0;
    if (g < gfull) { ff_group(x0, g, am, bits, q0, q1, end, sQ0, sQ1, l1h, l0); xt = x1; g++; }
    if (g < gfull) { ff_group(x1, g, am, bits, q0, q1, end, sQ0, sQ1, l1h, l0); xt = x2; g++; }
    if (g < gfull) { ff_group(x2, g, am, bits, q0, q1, end, sQ0, sQ1, l1h, l0); xt = x3; g++; }
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int w = 4 * gfull + c;
        if (w > wlast) break;
        uint32_t m = mismatch_word(word_of(xt, c), am[w]);
        const int nb = (w < wlast) ? 32 : (D - (wlast << 5));
        if (nb < 32) m &= (1u << nb) - 1u;
        ff_word(bits, q0, q1, end, m, nb, sQ0, sQ1, l1h, l0);
    }
    return __uint_as_float(bits);
}

// the class's increment tables, one exponent per thread of a 256-thread block (caller synchronises): q0 and
// q1 - q0.  An unusable q0 makes the whole binade unusable (q1 - q0 must stay a 24-bit non-negative number); an
// unusable q1 alone is kept: then q1 - q0 = 2^23 - q0 and a mismatch still "leaves the binade".
__device__ __forceinline__ void ff_build(uint32_t* sQ0, uint32_t* sDQ, double l1, double l0, int tid)
{
    uint32_t q0, q1;
    ff_entry(l1, -l0, tid, q0, q1);
    if (q0 == kFFInvalid) q1 = kFFInvalid;
    sQ0[tid] = q0; sDQ[tid] = q1 - q0;
}

// Epilogue of both density kernels: lane tid of the tile ran family perm[tile*256 + tid], which lies in the same
// tile; the block hands its 256 results back in family order through LDS and stores them contiguously.
__device__ __forceinline__ void density_store(const int* __restrict__ perm, int tile, int tid, int n, int npad, int k,
                                              float dk, uint32_t nul, double pk, float logpk,
                                              double* __restrict__ pkfki, float* __restrict__ logpkfki)
{
    __shared__ double sPk[256];
    __shared__ float sLp[256];
    float logfk; double fk;
    if (!nul) { logfk = -dk; fk = exp((double)logfk); }          // nem_mod.c:679-680
    else { logfk = -FLT_MAX; fk = 0.0; }                         // nem_mod.c:685-686
    const int slot = perm[tile * 256 + tid] - tile * 256;
    sPk[slot] = pk * fk;                                         // nem_alg.c:2282
    sLp[slot] = logpk + logfk;                                   // nem_alg.c:2283
    __syncthreads();
    const int io = tile * 256 + tid;
    if (io < n) {
        pkfki[(size_t)k * npad + io] = sPk[tid];
        logpkfki[(size_t)k * npad + io] = sLp[tid];
    }
}

struct DensityArgs {
    const uint4* xw; int n, npad, dpad, D, K;
    const double2* tabT; const double* tabL0; const uint32_t* nz0; const uint32_t* nz1;
    const uint32_t* am0; const uint32_t* am1; const double2* uni; const int* nonuni;
    const double* pk; const float* logpk;
    double* pkfki; float* logpkfki;
    int* zero_flags; int n_zero_flags;
    const int* stop;
    int use_ff;
    const int* perm;
    const uint2* ffq;                                    // per class: the 256 (q0, q1 - q0) of ff_build, made by k_finish
};

__device__ __forceinline__ void density_body(const DensityArgs& a)
{
    __shared__ double2 sT[DCH];
    __shared__ double sL[DCH];
    __shared__ uint32_t sQ0[256], sQ1[256];
    __shared__ uint2 sAm[kDensMaskWords];                // {am0, am1} of a word together: one 8-byte LDS read
    if (a.stop != nullptr && *a.stop) return;
    int k, tile;
    if (!density_tile(a.K, a.npad >> 8, tile, k)) return;
    const int tid = threadIdx.x;
    const int i = tile * 256 + tid;                      // i < npad by construction
    const int npad = a.npad, dpad = a.dpad;
    const int W = dpad >> 5;
    float dk = 0.0f;
    uint32_t nul = 0;
    // the sweep that follows this launch starts from clean flags (MOVED + the relaxation-round window)
    if (tile == 0 && k == 0)
        for (int t = tid; t < a.n_zero_flags; t += 256) a.zero_flags[t] = 0;

    if (a.nonuni[k] == 0) {
        // ---- uniform dispersion in this class (sk_, s__, the default .m): the step constants are two
        // wave-uniform doubles; which organisms mismatch comes from two bit masks per word.
        const double l1 = a.uni[k].x, l0 = a.uni[k].y;
        const double l1h = 0.5 * l1;                     // l1 is 0 or a normal double: halving is exact
        // the class's mismatch masks go through LDS: read from global inside the chain they would share the
        // vector-memory counter with the matrix prefetch and drain it at every word
        // (W <= kDensMaskWords here: table_flag_general sends wider matrices down the general path)
        for (int w = tid; w < W; w += 256) sAm[w] = make_uint2(a.am0[k * W + w], a.am1[k * W + w]);
        if (a.use_ff && l1 >= 0.0 && l0 <= 0.0) {        // block-uniform
            const uint2 q = a.ffq[k * 256 + tid];          // (k_finish built the class's increment table)
            sQ0[tid] = q.x; sQ1[tid] = q.y;
            __syncthreads();
            dk = chain_ff(a.xw, npad, i, a.D, sAm, sQ0, sQ1, l1h, l0);
        } else {
            __syncthreads();
            dk = chain_plain(a.xw, npad, i, a.D, sAm, l1h, l0);
        }
    } else {
        // ---- general case (skd, s_d, hand-written .m files): per-(k,d) constants staged through LDS
        for (int d0 = 0; d0 < dpad; d0 += DCH) {
            const int dn = min(DCH, dpad - d0);
            __syncthreads();
            for (int t = tid; t < dn; t += 256) {
                sT[t] = a.tabT[(size_t)k * dpad + d0 + t];
                sL[t] = a.tabL0[(size_t)k * dpad + d0 + t];
            }
            __syncthreads();
            const int w0 = d0 >> 5, wn = dn >> 5;        // DCH = 16 words: chunks start on a 4-word group
            const int g0 = w0 >> 2, gn = (wn + 3) >> 2;
            uint4 xnext = a.xw[(size_t)g0 * npad + i];
            for (int g = 0; g < gn; g++) {
                const uint4 xv = xnext;
                if (g + 1 < gn) xnext = a.xw[(size_t)(g0 + g + 1) * npad + i];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int w = 4 * g + c;
                    if (w >= wn) break;
                    const uint32_t x = word_of(xv, c);
                    nul |= (x & a.nz1[k * W + w0 + w]) | (~x & a.nz0[k * W + w0 + w]);
#pragma unroll 8
                    for (int b = 0; b < 32; b++) {
                        const double2 tt = sT[w * 32 + b];
                        const double l0 = sL[w * 32 + b];
                        const double add = ((x >> b) & 1u) ? tt.y : tt.x;
                        dk = (float)(((double)dk + add) - l0);   // nem_mod.c:661
                    }
                }
            }
        }
    }
    density_store(a.perm, tile, tid, a.n, npad, k, dk, nul, a.pk[k], a.logpk[k], a.pkfki, a.logpkfki);
}

// ------------------------------------------------------------------------------------------
// E1 with the NCEM parameter update folded in (sk_ and skd models): every block derives the
// constants of ITS class straight from the globally summed integer statistics -- centres by
// ComputeMedian's rule, inertia in closed form, dispersion, proportion, the two logs -- instead of
// waiting for a single-block "finish" launch to publish tables.  K*D is tiny, so the redundancy
// across blocks is cheaper than a kernel boundary.  Block (0, k) also publishes class k's new
// centre / dispersion / proportion / size; block (0, 0) the empty-class flag.
// ------------------------------------------------------------------------------------------
constexpr int FD_MAXD = kFusedMaxD; // organisms the fused kernel supports (iner/eps staged in LDS as floats)
constexpr int FD_CH = 256;        // organisms per general-path table chunk

struct FusedDensityArgs {
    const uint4* xw; int n, npad, dpad, D, K, n_total, disper, propor;
    const int* stats; int stats_ranks, stats_rank_stride;
    float* center; float* disp; float* prop; float* nbobs_k;
    int* iter_flags;
    double* pkfki; float* logpkfki;
    int* zero_flags; int n_zero_flags;
    const int* stop;
    int use_ff;
    const int* perm;
};

// returns the family tile the block worked on, -1 when it had nothing to do
__device__ __forceinline__ int density_fused_body(const FusedDensityArgs& a)
{
    __shared__ __align__(16) float sVal[FD_MAXD];        // inertia per organism, later epsilon per organism
    __shared__ double2 sT[FD_CH];
    __shared__ double sL[FD_CH];
    __shared__ uint2 sAm[FD_MAXD / 32];                   // {am0, am1} per word
    __shared__ uint32_t sNz0[FD_CH / 32], sNz1[FD_CH / 32];
    __shared__ uint32_t sQ0[256], sQ1[256];
    __shared__ unsigned long long sTot2;
    __shared__ float sEps, sChain[2];
    __shared__ int sGeneral;
    if (a.stop != nullptr && *a.stop) return -1;
    int k, tile;
    if (!density_tile(a.K, a.npad >> 8, tile, k)) return -1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = tile * 256 + tid;                      // i < npad by construction
    const int npad = a.npad, dpad = a.dpad, D = a.D, K = a.K;
    const bool writer = (tile == 0);                     // this block publishes class k's parameters
    if (tile == 0 && k == 0) {
        for (int t = tid; t < a.n_zero_flags; t += 256) a.zero_flags[t] = 0;
        if (tid == 0) {                                  // EstimLaplaceCenters, nem_mod.c:1404-1408
            int ek = 0;
            for (int c = 0; c < K; c++)
                if (!((double)(float)stat_sum(a.stats, c, a.stats_ranks, a.stats_rank_stride) > kEpsilonD)) ek = c + 1;
            a.iter_flags[FLAG_EMPTYK] = ek;
        }
    }
    if (tid == 0) { sTot2 = 0ull; sGeneral = 0; }
    __syncthreads();

    // ---- centres + inertia of class k from the counts (k_finish's centers_ncem_entry, per block)
    const int nkI = stat_sum(a.stats, k, a.stats_ranks, a.stats_rank_stride);
    const float nkf = (float)nkI;
    const bool nonempty = (double)nkf > kEpsilonD;
    long long acc2 = 0;
    int general = 0;
    for (int d = tid; d < dpad; d += 256) {              // dpad % 64 == 0: whole waves reach the ballots
        float mu = 0.0f, in = 0.0f;
        int a0 = 0, a1 = 0;
        if (d < D) {
            if (nonempty) {
                const float half = nkf / 2;
                const int s1 = stat_sum(a.stats, K + k * D + d, a.stats_ranks, a.stats_rank_stride);
                const float s0f = (float)(nkI - s1);
                if (s0f > half) { mu = 0.0f; in = (float)s1; }
                else if (s0f == half) { mu = 0.5f; in = 0.5f * nkf; }
                else { mu = 1.0f; in = s0f; }
                if (writer) a.center[k * D + d] = mu;
            } else {
                mu = a.center[k * D + d];                // empty class keeps its centre (nem_mod.c:1405)
            }
            const int ad0 = abs((int)(0.0f - mu)), ad1 = abs((int)(1.0f - mu));
            a0 = (ad0 != 0); a1 = (ad1 != 0);
            if (ad0 > 1 || ad1 > 1) general = 1;
            sVal[d] = in;
            acc2 += (long long)(2.0f * in);
        }
        const uint64_t b0 = __ballot(a0), b1 = __ballot(a1);
        if (lane == 0) {
            const int w = d >> 5;
            sAm[w] = make_uint2((uint32_t)b0, (uint32_t)b1);
            sAm[w + 1] = make_uint2((uint32_t)(b0 >> 32), (uint32_t)(b1 >> 32));
        }
    }
    acc2 = wave_reduce_add_ll(acc2);
    if (lane == 0 && acc2 != 0) atomicAdd(&sTot2, (unsigned long long)acc2);
    if (general) sGeneral = 1;
    __syncthreads();

    // ---- dispersion (InerToDispK_ / InerToDispKD with MISSING_IGNORE, nem_mod.c:1043-1073, 1152-1170)
    bool eps_per_d = false;                              // epsilon differs between organisms -> general chain
    if (a.disper == NEMGPU_DISP_K_) {
        if (nkf > 0) {
            const long long cap = 1ll << 24;
            if ((long long)sTot2 <= cap && nkI < (1 << 24)) {                              // (block-uniform)
                // the d-ordered inertia chain never rounds here (multiples of 1/2 below 2^23); the N_KD chain -- the
                // class size added D times -- is its closed form (nem_ff.hpp: exact while N_K * D <= 2^24, then one
                // division per binade)
                if (tid == 0) sEps = (0.5f * (float)(long long)sTot2) / ff_repeat_add_u24((uint32_t)nkI, D);
            } else {
                // the two d-ordered chains of InerToDispK_ (nem_mod.c:1054-1058), each on a wave of its own: the
                // inertia values sixteen at a time from LDS ahead of the dependent adds, the N_KD chain from a register
                if (tid == 0) {
                    float si = 0.0f;
                    const float4* v4 = reinterpret_cast<const float4*>(sVal);
                    int d = 0;
                    for (; d + 16 <= D; d += 16) si = chain_add16(si, v4[d >> 2], v4[(d >> 2) + 1], v4[(d >> 2) + 2], v4[(d >> 2) + 3]);
                    for (; d < D; d++) si += sVal[d];
                    sChain[0] = si;
                } else if (tid == 64) {
                    if (nkI < (1 << 24)) sChain[1] = ff_repeat_add_u24((uint32_t)nkI, D);
                    else {
                        float sn = 0.0f;
#pragma unroll 8
                        for (int d = 0; d < D; d++) sn += nkf;
                        sChain[1] = sn;
                    }
                }
                __syncthreads();
                if (tid == 0) sEps = sChain[0] / sChain[1];
            }
            __syncthreads();
            const float e = sEps;
            if (writer) for (int d = tid; d < D; d += 256) a.disp[k * D + d] = e;
        } else {
            // empty class keeps whatever dispersions it had (possibly different per organism)
            __syncthreads();
            for (int d = tid; d < D; d += 256) sVal[d] = a.disp[k * D + d];
            eps_per_d = true;
        }
    } else {                                             // NEMGPU_DISP_KD
        __syncthreads();
        for (int d = tid; d < D; d += 256) {
            float e;
            if (nonempty) { e = sVal[d] / nkf; if (writer) a.disp[k * D + d] = e; }
            else e = a.disp[k * D + d];
            sVal[d] = e;
        }
        eps_per_d = true;
    }
    // ---- proportion (nem_mod.c:456-465) and the class constants of ComputePkFkiM (nem_alg.c:2262-2271)
    const float propk = (a.propor == NEMGPU_PROP_K) ? nkf / (float)a.n_total : (float)(1.0 / K);
    if (writer && tid == 0) { a.prop[k] = propk; a.nbobs_k[k] = nkf; }
    const double pkd = (double)propk;
    const float logpk = (pkd > kEpsilonD) ? (float)log(pkd) : -INFINITY;
    __syncthreads();

    float dk = 0.0f;
    uint32_t nul = 0;
    const float eps_u = sEps;
    bool uniform = !eps_per_d && !sGeneral && ((double)eps_u > kEpsilonD);
    double l1 = 0.0, l0 = 0.0;
    if (uniform) {
        l1 = log((double)((1.0f - eps_u) / eps_u));
        l0 = log((double)(1.0f - eps_u));
        if (!isfinite(l1) || !isfinite(l0)) uniform = false;
    }
    if (uniform) {
        const double l1h = 0.5 * l1;                     // l1 is 0 or a normal double: halving is exact
        if (a.use_ff && l1 >= 0.0 && l0 <= 0.0) {        // block-uniform
            ff_build(sQ0, sQ1, l1, l0, tid);
            __syncthreads();
            dk = chain_ff(a.xw, npad, i, D, sAm, sQ0, sQ1, l1h, l0);
        } else {
            dk = chain_plain(a.xw, npad, i, D, sAm, l1h, l0);
        }
    } else {
        // general chain: per-organism constants built chunk by chunk in LDS (table_entry's arithmetic).
        // With every centre in {0, 1/2, 1} (sGeneral == 0; the class masks sAm are those of the new centres) an
        // organism's step is the uniform chain's, dk <- fma({2 | 0}, l1_d / 2, dk) - l0_d, with ITS two constants:
        // one 16-byte LDS read and six instructions instead of two reads, two selects and the same arithmetic.
        const bool by_mask = nonempty && !sGeneral;
        for (int d0 = 0; d0 < dpad; d0 += FD_CH) {
            const int dn = min(FD_CH, dpad - d0);
            __syncthreads();
            for (int t = tid; t < dn; t += 256) {        // dn % 64 == 0
                const int d = d0 + t;
                double t0 = 0.0, t1 = 0.0, ll0 = 0.0;
                int n0 = 0, n1 = 0;
                if (d < D) {
                    const float eps = eps_per_d ? sVal[d] : eps_u;
                    int ad0, ad1;
                    if (nonempty) { ad0 = (sAm[d >> 5].x >> (d & 31)) & 1; ad1 = (sAm[d >> 5].y >> (d & 31)) & 1; }
                    else { const float mu = a.center[k * D + d]; ad0 = abs((int)(0.0f - mu)); ad1 = abs((int)(1.0f - mu)); }
                    if ((double)eps > kEpsilonD) {
                        const double ll1 = log((double)((1.0f - eps) / eps));
                        ll0 = log((double)(1.0f - eps));
                        t0 = (double)ad0 * ll1;
                        t1 = (double)ad1 * ll1;
                        if (by_mask) { t0 = 0.5 * ll1; t1 = ll0; }           // (l1 / 2, l0): see the chain below
                    } else { n0 = (ad0 != 0); n1 = (ad1 != 0); }
                }
                sT[t] = make_double2(t0, t1);
                sL[t] = ll0;
                const uint64_t m0 = __ballot(n0), m1 = __ballot(n1);
                if (lane == 0) {
                    sNz0[t >> 5] = (uint32_t)m0; sNz0[(t >> 5) + 1] = (uint32_t)(m0 >> 32);
                    sNz1[t >> 5] = (uint32_t)m1; sNz1[(t >> 5) + 1] = (uint32_t)(m1 >> 32);
                }
            }
            __syncthreads();
            const int w0 = d0 >> 5, wn = dn >> 5;        // FD_CH = 8 words: chunks start on a 4-word group
            const int g0 = w0 >> 2, gn = (wn + 3) >> 2;
            uint4 xnext = a.xw[(size_t)g0 * npad + i];
            for (int g = 0; g < gn; g++) {
                const uint4 xv = xnext;
                if (g + 1 < gn) xnext = a.xw[(size_t)(g0 + g + 1) * npad + i];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int w = 4 * g + c;
                    if (w >= wn) break;
                    const uint32_t x = word_of(xv, c);
                    nul |= (x & sNz1[w]) | (~x & sNz0[w]);
                    if (by_mask) {                               // block-uniform
                        const uint32_t m = mismatch_word(x, sAm[w0 + w]);
#pragma unroll 8
                        for (int b = 0; b < 32; b++) {
                            const double2 cc = sT[w * 32 + b];   // (l1_d / 2, l0_d); zeros for padding and null dispersions
                            dk = bern_step(dk, m, b, cc.x, cc.y);
                        }
                    } else {
#pragma unroll 8
                        for (int b = 0; b < 32; b++) {
                            const double2 tt = sT[w * 32 + b];
                            const double c0 = sL[w * 32 + b];
                            const double add = ((x >> b) & 1u) ? tt.y : tt.x;
                            dk = (float)(((double)dk + add) - c0);   // nem_mod.c:661
                        }
                    }
                }
            }
        }
    }
    density_store(a.perm, tile, tid, a.n, npad, k, dk, nul, pkd, logpk, a.pkfki, a.logpkfki);
    return tile;
}


// ------------------------------------------------------------------------------------------
// NCEM bookkeeping after a sweep: per-class membership bitmasks (for the popcount M-step) and
// the CVTEST_CLAS flag (HasConverged, nem_alg.c:2075-2089: max|c - cold| is 1 iff a label moved).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void labels_post_body(int n_local, int lo, int K, int nw64, const uint8_t* __restrict__ lab_new,
                              const uint8_t* __restrict__ lab_old, uint64_t* __restrict__ mask,
                              int* __restrict__ flags, const int* __restrict__ stop, const CtrlArgs& ca, const int nblk)
{
    if (stop != nullptr && *stop) return;
    const int lane = threadIdx.x & 63;
    // grid-stride over 64-family groups: the grid is capped (launch_labels_post) so that the last-block ticket
    // and the "moved" flag cost a bounded number of same-address atomics however many families there are
    int any_moved = 0;
    for (int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave < nw64; wave += (nblk * blockDim.x) >> 6) {
        const int i = wave * 64 + lane;
        int lab = 255, moved = 0;
        if (i < n_local) {
            lab = lab_new[lo + i] & 0x7F;                    // (bit 7: the site drew, TIE_LIBC)
            if (lab_old != nullptr) moved = (lab != ((int)lab_old[lo + i] & 0x7F));
        }
        for (int k = 0; k < K; k++) {
            uint64_t m = __ballot(lab == k);
            if (lane == 0) mask[(size_t)k * nw64 + wave] = m;
        }
        any_moved |= __any(moved);
    }
    // (one device-scope atomic per block at most, behind a device-scope look at the flag: see sweep_body)
    const int blk_moved = __syncthreads_or(any_moved);
    if (blk_moved && threadIdx.x == 0 && __hip_atomic_load(&flags[FLAG_MOVED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&flags[FLAG_MOVED], 1);
    if (ca.ctrl != nullptr && last_block_ticket(ca.ticket, nblk)) ctrl_logic(ca);
}

// M1-M3 for NCEM as integer counts: S1[k][d] = #{i : label_i = k, x_id = 1}, N_k = #{label = k}.
// grid = d + 1 blocks (the last one counts class sizes); out: stats[0..K) = N_k, stats[K + k*d + j] = S1.
template <int R>
__device__ __forceinline__ void mstep_counts_body(int K, int D, int nw64, const uint64_t* __restrict__ xt,
                                                  const uint64_t* __restrict__ mask, int* __restrict__ stats,
                                                  const int* __restrict__ stop, const CtrlArgs& prev_ctrl, const int bx, const int nblk)
{
    // R organism rows per block (row D = the all-ones row that counts the class sizes): each class-mask word is
    // loaded once for R rows, so the masks' L2 traffic (K * N/8 bytes per block) shrinks by R
    __shared__ int red[4][R][4];
    // One block more than there are rows: the loop control of the iteration BEFORE this one (ctrl_logic), when its
    // last sweep round left it to us -- there it costs a last-block ticket, two device-wide atomic round trips at the
    // tail of the launch; here it runs beside the counting blocks.  They have read the stop word before it can be
    // raised, so the counts of an iteration that will not happen are computed once for nothing.
    if (prev_ctrl.ctrl != nullptr && bx == nblk - 1) {
        if (threadIdx.x == 0) ctrl_logic(prev_ctrl);
        return;
    }
    if (stop != nullptr && *stop) return;
    const int d0 = bx * R;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t* rows[R];
#pragma unroll
    for (int r = 0; r < R; r++) rows[r] = xt + (size_t)min(d0 + r, D - 1) * nw64;
    // four classes per pass over the bit rows (K = 3: the rows are read once)
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int kn = min(4, K - k0);
        int acc[4][R];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < R; r++) acc[c][r] = 0;
        // two words per thread and trip, all their loads requested before any is used (the masks were written by the
        // launch before this one: every trip of a one-word loop would wait for memory on its own)
        for (int j = threadIdx.x; j < nw64; j += 512) {
            const int j2 = j + 256;
            const bool two = j2 < nw64;
            uint64_t xv[R], xw2[R], m[4], m2[4];
#pragma unroll
            for (int r = 0; r < R; r++) {
                xv[r] = (d0 + r < D) ? rows[r][j] : ~0ull;
                xw2[r] = two ? ((d0 + r < D) ? rows[r][j2] : ~0ull) : 0ull;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                m[c] = (c < kn) ? mask[(size_t)(k0 + c) * nw64 + j] : 0ull;
                m2[c] = (c < kn && two) ? mask[(size_t)(k0 + c) * nw64 + j2] : 0ull;
            }
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int r = 0; r < R; r++) acc[c][r] += __popcll(xv[r] & m[c]) + __popcll(xw2[r] & m2[c]);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int v = wave_reduce_add(acc[c][r]);
                if (lane == 0) red[c][r][wv] = v;
            }
        __syncthreads();
        if (threadIdx.x < 4 * R) {
            const int c = threadIdx.x / R, r = threadIdx.x % R, k = k0 + c, d = d0 + r;
            if (c < kn && d <= D) {
                const int v = red[c][r][0] + red[c][r][1] + red[c][r][2] + red[c][r][3];
                if (d < D) stats[K + k * D + d] = v; else stats[k] = v;
            }
        }
    }
}

// centres and inertia from the (globally summed) counts.  With c in {0,1}:
//   N_K = count (EstimSizes, nem_mod.c:1293-1315);  halfwei = N_K/2;  zeros-side weight S0 = N_K - S1;
//   ComputeMedian (nem_mod.c:1439-1477) gives mu = 0 if S0 > half, 0.5 if S0 == half, 1 otherwise;
//   EstimLaplaceIner (nem_mod.c:1669-1686) gives S1, S0 or N_K/2 for mu = 0, 1, 0.5 (all exact).
__device__ inline void centers_ncem_entry(int K, int D, const int* __restrict__ stats, int ranks, int rstride,
                                          float* __restrict__ center, float* __restrict__ nbobs_k,
                                          float* __restrict__ iner, int t)
{
    const int k = t / D;
    const int nk = stat_sum(stats, k, ranks, rstride);
    const float nkf = (float)nk;
    if (t - k * D == 0) nbobs_k[k] = nkf;
    float in = 0.0f;
    if ((double)nkf > kEpsilonD) {
        const float half = nkf / 2;
        const int s1 = stat_sum(stats, K + t, ranks, rstride);
        const float s0f = (float)(nk - s1);
        float mu;
        if (s0f > half) { mu = 0.0f; in = (float)s1; }
        else if (s0f == half) { mu = 0.5f; in = 0.5f * nkf; }
        else { mu = 1.0f; in = s0f; }
        center[t] = mu;
    }
    iner[t] = in;
}

// M4/M5: dispersion model + proportions (InerToDisp*, nem_mod.c:965-1174; EstimPara :456-465).
// MissMode is MISSING_IGNORE for Bernoulli (nem_mod.c:446-448) and N_KD[k][d] == N_K[k] (no NaN).
// The d- / k-ordered float sums are order-dependent (values exceed 2^24) and stay sequential.
// `exact_half_ints`: the inertia values are non-negative multiples of 1/2 and N_K is an integer (NCEM).
// Then a d-ordered float chain whose total stays below 2^24 half-units never rounds, and equals the
// closed form; only larger totals (e.g. 50 000 x 1 000) take the sequential chain.
__device__ inline void disp_body(int K, int D, int n_total, int disper, int propor, int exact_half_ints,
                                 const float* __restrict__ nbobs_k, const float* __restrict__ iner,
                                 float* __restrict__ disp, float* __restrict__ prop, int* __restrict__ flags,
                                 int kb, int ke)
{
    // classes [kb, ke) are this block's (one block per class for the class-separable models sk_ / skd,
    // all classes in one block for s__ / s_d)
    constexpr int CAP = 16000;                           // staged inertia values (64 000 B)
    __shared__ float4 s_in4[CAP / 4];
    __shared__ float s_disp[kMaxKernelK], s_si[kMaxKernelK], s_sn[kMaxKernelK];
    __shared__ int s_valid[kMaxKernelK];
    __shared__ float s_vol;
    float* s_in = reinterpret_cast<float*>(s_in4);
    __shared__ unsigned long long s_tot2[kMaxKernelK];
    __shared__ int s_seq[kMaxKernelK];
    const int tid = threadIdx.x;
    if (disper == NEMGPU_DISP_K_) {
        // closed form where the chain provably never rounds (see above)
        if (tid < K) { s_tot2[tid] = 0ull; s_seq[tid] = 1; s_valid[tid] = 0; }
        __syncthreads();
        if (exact_half_ints) {
            for (int k = kb; k < ke; k++) {
                long long acc = 0;
                for (int d = tid; d < D; d += 1024) acc += (long long)(2.0f * iner[k * D + d]);
                acc = wave_reduce_add_ll(acc);
                if ((tid & 63) == 0 && acc != 0) atomicAdd(&s_tot2[k], (unsigned long long)acc);
            }
            __syncthreads();
            if (tid >= kb && tid < ke) {
                const float nk = nbobs_k[tid];
                const long long cap = 1ll << 24;
                if ((long long)s_tot2[tid] <= cap && (long long)nk * (long long)D <= cap) {
                    s_seq[tid] = 0;
                    s_valid[tid] = (nk > 0);
                    if (nk > 0) s_disp[tid] = (0.5f * (float)(long long)s_tot2[tid]) / (nk * (float)D);
                }
            }
            __syncthreads();
        }
        NEM_PHASE(8);
        bool any_seq = false;
        for (int k = kb; k < ke; k++) any_seq |= (s_seq[k] != 0);
        // per class: sn = sum_d N_KD, si = sum_d Iner, both d-ordered float chains (InerToDispK_,
        // nem_mod.c:1054-1058).  Every chain gets a wave of its own (lane 0): the class's D inertia values are
        // staged in LDS, the N_KD chain adds a constant and needs no memory at all.
        const int Dp = (D + 3) & ~3;
        const int per = max(1, min(ke - kb, CAP / max(Dp, 1)));    // classes staged per pass
        if (!any_seq) {
            // nothing left for the sequential chains
        } else if (Dp <= CAP) {
            for (int k0 = kb; k0 < ke; k0 += per) {
                const int kn = min(per, ke - k0);
                __syncthreads();
                for (int kk = 0; kk < kn; kk++)
                    for (int d = tid; d < D; d += 1024) s_in[kk * Dp + d] = iner[(k0 + kk) * D + d];
                __syncthreads();
                NEM_PHASE(9);
                if ((tid & 63) == 0) {
                    for (int c = tid >> 6; c < 2 * kn; c += 16) {
                        const int kk = c >> 1, k = k0 + kk;
                        if (!s_seq[k]) continue;
                        const float nk = nbobs_k[k];
                        if (!(nk > 0)) continue;
                        if (c & 1) {
                            float sn = 0.0f;
#pragma unroll 8
                            for (int d = 0; d < D; d++) sn += nk;
                            s_sn[k] = sn;
                        } else {
                            float si = 0.0f;
                            const float* base = s_in + kk * Dp;
                            const float4* b4 = reinterpret_cast<const float4*>(base);
                            const int dq = D & ~3;
#pragma unroll 8
                            for (int t = 0; t < (dq >> 2); t++) {
                                const float4 v = b4[t];
                                si = (((si + v.x) + v.y) + v.z) + v.w;
                            }
                            for (int d = dq; d < D; d++) si += base[d];
                            s_si[k] = si;
                        }
                    }
                }
                __syncthreads();
                NEM_PHASE(10);
                if (tid < kn && s_seq[k0 + tid]) {
                    const int k = k0 + tid;
                    const float nk = nbobs_k[k];
                    s_valid[k] = (nk > 0);
                    if (nk > 0) s_disp[k] = s_si[k] / s_sn[k];
                }
            }
        } else {                                                   // very wide matrices: straight from global
            if (tid >= kb && tid < ke && s_seq[tid]) {
                const int k = tid;
                const float nk = nbobs_k[k];
                s_valid[k] = (nk > 0);
                if (nk > 0) {
                    float sn = 0.0f, si = 0.0f;
                    for (int d = 0; d < D; d++) { sn += nk; si += iner[k * D + d]; }
                    s_disp[k] = si / sn;
                }
            }
        }
        __syncthreads();
        for (int t = kb * D + tid; t < ke * D; t += 1024) {
            const int k = t / D;
            if (s_valid[k]) disp[t] = s_disp[k];
        }
    } else if (disper == NEMGPU_DISP___) {
        // one (k, d)-ordered chain over all non-empty classes (InerToDisp__, nem_mod.c:988-1007)
        if (tid == 0) {
            float vol = 0.0f, nobs = 0.0f;
            for (int k = 0; k < K; k++) {
                const float nk = nbobs_k[k];
                if (nk > 0) {
#pragma unroll 8
                    for (int d = 0; d < D; d++) { vol += iner[k * D + d]; nobs += nk; }
                }
            }
            s_vol = vol / nobs;
        }
        __syncthreads();
        for (int t = tid; t < K * D; t += 1024) disp[t] = s_vol;
    } else if (disper == NEMGPU_DISP__D) {
        for (int d = tid; d < D; d += 1024) {
            float sn = 0.0f, si = 0.0f;
            for (int k = 0; k < K; k++) { sn += nbobs_k[k]; si += iner[k * D + d]; }
            const float dd = si / sn;
            for (int k = 0; k < K; k++) disp[k * D + d] = dd;
        }
    } else {
        for (int t = kb * D + tid; t < ke * D; t += 1024) {
            const int k = t / D;
            if ((double)nbobs_k[k] > kEpsilonD) disp[t] = iner[t] / nbobs_k[k];
        }
    }
    if (tid >= kb && tid < ke) {
        if (propor == NEMGPU_PROP_K) prop[tid] = nbobs_k[tid] / (float)n_total;
        else prop[tid] = (float)(1.0 / K);
    }
    if (tid == 0 && kb == 0) {                           // EstimLaplaceCenters :1404-1408
        int ek = 0;
        for (int k = 0; k < K; k++) if (!((double)nbobs_k[k] > kEpsilonD)) ek = k + 1;
        flags[FLAG_EMPTYK] = ek;
    }
}

// The same update for ONE class per block (the class-separable models sk_ / skd and tables-only launches) and
// dpad <= 1024 * NI: every thread keeps its organisms' centre, inertia and dispersion in registers from the counts
// to the tables, so no phase waits for the stores of the phase before it (each such store -> barrier -> load costs
// 1.5-2 us, and the generic form below has five of them); the class constants' logs are taken once; the N_KD chain
// of the sk_ model is its closed form (nem_ff.hpp) and the inertia chain -- the only sequential part left, D
// dependent float adds on one lane -- reads its values from LDS one group ahead of the adds.
template <int NI>
__device__ __forceinline__ void finish_lean(const FinishArgs& a)
{
    constexpr int CAP = 1024 * NI;
    __shared__ float4 s_in4[CAP / 4];
    __shared__ HsShared s_hs;
    __shared__ float s_si, s_sn, s_eps0;
    __shared__ double s_l1, s_l0;
    __shared__ unsigned long long s_tot2;
    float* s_in = reinterpret_cast<float*>(s_in4);
    const int tid = threadIdx.x, k = blockIdx.x, K = a.K, D = a.D, dpad = a.dpad;
    const int lane = tid & 63;
    const bool restart = a.reset_prop != nullptr;
    if (restart) {
        // initial parameters back in place, loop control cleared (the stop word may still be set from the run before;
        // every later kernel of the batch reads it after this launch)
        if (tid == 0) a.nbobs_k[k] = 0.0f;
        if (k == 0) {
            if (tid < a.reset_ctrl_words && tid != C_FOLD) a.reset_ctrl[tid] = 0;
            if (tid == 0) a.reset_sweep_next[0] = 0;
        }
    } else if (a.stop != nullptr && *a.stop) return;
    NEM_PHASE(0);
    float mu[NI], in[NI], eps[NI];
    float nkf = 0.0f;
    if (tid == 0) s_tot2 = 0ull;
    // ---- centres and inertia
    if (a.mode == 1) {
        // (every block needs every class size for the empty-class flag; its own class's entries otherwise)
        if (tid < K) a.nbobs_k[tid] = (float)stat_sum(a.stats, tid, a.stats_ranks, a.stats_rank_stride);
        const int nk = stat_sum(a.stats, k, a.stats_ranks, a.stats_rank_stride);
        nkf = (float)nk;
        const bool live = (double)nkf > kEpsilonD;
        const float half = nkf / 2;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int d = tid + 1024 * i;
            mu[i] = 0.0f; in[i] = 0.0f;
            if (d < D) {
                const int t = k * D + d;
                if (live) {                                      // centers_ncem_entry
                    const int s1 = stat_sum(a.stats, K + t, a.stats_ranks, a.stats_rank_stride);
                    const float s0f = (float)(nk - s1);
                    if (s0f > half) { mu[i] = 0.0f; in[i] = (float)s1; }
                    else if (s0f == half) { mu[i] = 0.5f; in[i] = 0.5f * nkf; }
                    else { mu[i] = 1.0f; in[i] = s0f; }
                    a.center[t] = mu[i];
                } else mu[i] = a.center[t];
                a.iner[t] = in[i];
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int d = tid + 1024 * i;
            mu[i] = 0.0f; in[i] = 0.0f;
            if (d < D) {
                const int t = k * D + d;
                if (restart) { mu[i] = a.reset_center[t]; a.center[t] = mu[i]; }
                else mu[i] = a.center[t];
                if (a.mode == 2) in[i] = a.iner[t];
            }
        }
        if (a.mode == 2) nkf = a.nbobs_k[k];
    }
    NEM_PHASE(1);
    // ---- dispersion (InerToDispKD / InerToDispK_) and proportions
    if (a.mode == 0) {
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int d = tid + 1024 * i;
            eps[i] = 0.0f;
            if (d < D) {
                if (restart) { eps[i] = a.reset_disp[k * D + d]; a.disp[k * D + d] = eps[i]; }
                else eps[i] = a.disp[k * D + d];
            }
        }
    } else if (a.disper == NEMGPU_DISP_KD) {
        const bool live = (double)nkf > kEpsilonD;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int d = tid + 1024 * i;
            eps[i] = 0.0f;
            if (d < D) {
                if (live) { eps[i] = in[i] / nkf; a.disp[k * D + d] = eps[i]; }
                else eps[i] = a.disp[k * D + d];
            }
        }
    } else {                                                     // sk_
        float dk = 0.0f;
        const bool valid = nkf > 0.0f;
        if (valid) {
            bool seq = true;
            if (a.mode == 1) {
                // the inertia values are non-negative multiples of 1/2 and N_K is an integer: a d-ordered float chain
                // whose total stays below 2^24 half-units never rounds and equals the closed form
                long long acc = 0;
#pragma unroll
                for (int i = 0; i < NI; i++) acc += (long long)(2.0f * in[i]);
                acc = wave_reduce_add_ll(acc);
                __syncthreads();                                 // (s_tot2 = 0 is in place)
                if (lane == 0 && acc != 0) atomicAdd(&s_tot2, (unsigned long long)acc);
                __syncthreads();
                const long long tot2 = (long long)s_tot2, cap = 1ll << 24;
                if (tot2 <= cap && nkf < 16777216.0f) {
                    // (the N_KD chain in closed form: exact while N_K * D <= 2^24, then one division per binade)
                    seq = false;
                    if (tid == 0) s_sn = ff_repeat_add_u24((uint32_t)nkf, D);
                    __syncthreads();
                    dk = (0.5f * (float)tot2) / s_sn;
                }
            }
            if (seq) {
                // sn = sum_d N_KD, si = sum_d Iner, both d-ordered float chains (InerToDispK_, nem_mod.c:1054-1058)
                __syncthreads();
#pragma unroll
                for (int i = 0; i < NI; i++) s_in[tid + 1024 * i] = in[i];        // (zeros beyond D: si + 0 = si)
                __syncthreads();
                NEM_PHASE(9);
                if (tid == 64) {
                    s_sn = (a.mode == 1 && nkf < 16777216.0f) ? ff_repeat_add_u24((uint32_t)nkf, D) : ff_repeat_add(nkf, D);
#ifdef NEM_PHASE_PROF
                    if (blockIdx.x == 0) g_phase[12] = wall_clock64();
#endif
                }
                if (a.mode == 1) {
                    // NCEM: non-negative multiples of 1/2 -- the chain in pieces, the whole block (nem_halfsum.hpp: exact
                    // prefix sums tell which float binade a step runs in, a run of steps inside one binade is a
                    // two-parity integer map, the few steps around a power of two are taken for real)
                    const float si = halfsum_block<16>(s_in, D, s_hs);
                    if (tid == 0) s_si = si;
                } else if (tid == 0) {
                    float si = 0.0f;
                    // groups of 32 values, two register sets: one is loaded from LDS while the other is added
                    // (measured and dropped, round 3: the chain as a block-wide scan of parity maps, binade by binade
                    //  -- exact, tested -- 42 us against 33.5: seven block-wide passes of ~5 us each)
                    const int ng = (D + 31) >> 5, last = CAP / 32 - 1;
                    float4 A[8], B[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) A[j] = s_in4[j];
                    for (int g = 0; g < ng; g += 2) {
                        const int g1 = min(g + 1, last), g2 = min(g + 2, last);
#pragma unroll
                        for (int j = 0; j < 8; j++) B[j] = s_in4[g1 * 8 + j];
                        si = chain_add16(si, A[0], A[1], A[2], A[3]);
                        si = chain_add16(si, A[4], A[5], A[6], A[7]);
#pragma unroll
                        for (int j = 0; j < 8; j++) A[j] = s_in4[g2 * 8 + j];
                        if (g + 1 < ng) {
                            si = chain_add16(si, B[0], B[1], B[2], B[3]);
                            si = chain_add16(si, B[4], B[5], B[6], B[7]);
                        }
                    }
                    s_si = si;
#ifdef NEM_PHASE_PROF
                    if (blockIdx.x == 0) g_phase[11] = wall_clock64();
#endif
                }
                __syncthreads();
                NEM_PHASE(10);
                dk = s_si / s_sn;
            }
        }
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int d = tid + 1024 * i;
            eps[i] = 0.0f;
            if (d < D) {
                if (valid) { eps[i] = dk; a.disp[k * D + d] = dk; }
                else eps[i] = a.disp[k * D + d];
            }
        }
    }
    float propk = 0.0f;
    if (tid == 0) {
        if (a.mode == 0) propk = restart ? a.reset_prop[k] : a.prop[k];
        else propk = (a.propor == NEMGPU_PROP_K) ? nkf / (float)a.n_total : (float)(1.0 / K);
        if (a.mode != 0 || restart) a.prop[k] = propk;
        if (a.mode != 0 && k == 0) {                             // EstimLaplaceCenters :1404-1408
            int ek = 0;
            for (int c = 0; c < K; c++) {
                const float nc = a.mode == 1 ? (float)stat_sum(a.stats, c, a.stats_ranks, a.stats_rank_stride) : a.nbobs_k[c];
                if (!((double)nc > kEpsilonD)) ek = c + 1;
            }
            a.flags[FLAG_EMPTYK] = ek;
        }
        s_eps0 = eps[0];
    }
    NEM_PHASE(2);
    __syncthreads();
    // ---- density tables (table_flag_general + table_entry, from the registers)
    const float eps0 = s_eps0;
    bool general = false;
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int d = tid + 1024 * i;
        if (d < D) {
            const int ad0 = abs((int)(0.0f - mu[i])), ad1 = abs((int)(1.0f - mu[i]));
            general |= (ad0 > 1) || (ad1 > 1) || (__float_as_uint(eps[i]) != __float_as_uint(eps0)) ||
                       !((double)eps[i] > kEpsilonD) || (dpad >> 5) > kDensMaskWords;
        }
    }
    if (tid == 0 && (double)eps0 > kEpsilonD) {
        const double l1 = log((double)((1.0f - eps0) / eps0)), l0 = log((double)(1.0f - eps0));
        if (!general && (!isfinite(l1) || !isfinite(l0))) general = true;   // 0 * inf / NaN must propagate as in the reference
        s_l1 = l1; s_l0 = l0;
        a.uni[k] = make_double2(l1, l0);
    }
    const bool nonuni = __syncthreads_or(general ? 1 : 0) != 0;
    NEM_PHASE(3);
    if (tid == 0) {
        a.nonuni[k] = nonuni ? 1 : 0;
        const double p = (double)propk;                          // ComputePkFkiM, nem_alg.c:2262-2271
        a.pk[k] = p;
        if (p > kEpsilonD) a.logpk[k] = (float)log(p);
        else { a.logpk[k] = -INFINITY; atomicOr(&a.flags[FLAG_EMPTY_PROP], 1); }
    }
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int dd = tid + 1024 * i;                           // dpad and 1024 are multiples of 64: whole waves
        if (dd < dpad) {
            double t0 = 0.0, t1 = 0.0, l0 = 0.0;
            int n0 = 0, n1 = 0, a0 = 0, a1 = 0;
            if (dd < D) {
                const int ad0 = abs((int)(0.0f - mu[i])), ad1 = abs((int)(1.0f - mu[i]));
                a0 = (ad0 != 0); a1 = (ad1 != 0);
                if ((double)eps[i] > kEpsilonD) {
                    if (nonuni) {
                        const double l1 = log((double)((1.0f - eps[i]) / eps[i]));
                        l0 = log((double)(1.0f - eps[i]));
                        t0 = (double)ad0 * l1;
                        t1 = (double)ad1 * l1;
                    }
                } else { n0 = a0; n1 = a1; }
            }
            const int t = k * dpad + dd;
            if (dd >= D || nonuni) {                             // (padding entries are always written: zeros)
                a.tabT[t] = make_double2(t0, t1);
                a.tabL0[t] = l0;
            }
            const uint64_t m0 = __ballot(n0), m1 = __ballot(n1), b0 = __ballot(a0), b1 = __ballot(a1);
            if (lane == 0) {
                const int w = t >> 5;                            // word index inside [K][dpad/32]
                a.nz0[w] = (uint32_t)m0; a.nz0[w + 1] = (uint32_t)(m0 >> 32);
                a.nz1[w] = (uint32_t)m1; a.nz1[w + 1] = (uint32_t)(m1 >> 32);
                a.am0[w] = (uint32_t)b0; a.am0[w + 1] = (uint32_t)(b0 >> 32);
                a.am1[w] = (uint32_t)b1; a.am1[w + 1] = (uint32_t)(b1 >> 32);
            }
        }
    }
    NEM_PHASE(4);
    if (a.use_ff && a.ffq != nullptr && !nonuni && tid < 256) {
        // the uniform chain's fast-forward increments (nem_ff.hpp) for a class that stays on that chain
        uint32_t q0, q1;
        ff_entry(s_l1, -s_l0, tid, q0, q1);
        if (q0 == kFFInvalid) q1 = kFFInvalid;
        a.ffq[k * 256 + tid] = make_uint2(q0, q1 - q0);
    }
    NEM_PHASE(5);
}

// The whole parameter update behind the sufficient statistics in ONE single-block launch:
// centres + inertia (NCEM, from the counts), dispersion + proportions, then the density tables
// E1 reads.  K*D is small (1 500 at configs[1], 15 000 at configs[3]); the only long part is the
// d-ordered float chain of the sk_/s__ models.
__device__ __forceinline__ void finish_body(const FinishArgs& a, const int nblk)
{
    const int tid = threadIdx.x;
    // grid = K blocks (one class each) for the class-separable dispersion models, else one block for all classes
    if (nblk > 1 && a.dpad <= 8192 && (a.mode == 0 || a.disper == NEMGPU_DISP_K_ || a.disper == NEMGPU_DISP_KD)) {
        finish_lean<8>(a);
        return;
    }
    const int kb = (nblk > 1) ? blockIdx.x : 0;
    const int ke = (nblk > 1) ? kb + 1 : a.K;
    if (a.reset_prop != nullptr) {
        // restart: initial parameters back in place, loop control cleared (the stop word may still be set from
        // the run before; every later kernel of the batch reads it after this launch)
        for (int t = kb * a.D + tid; t < ke * a.D; t += 1024) { a.center[t] = a.reset_center[t]; a.disp[t] = a.reset_disp[t]; }
        if (tid >= kb && tid < ke) { a.prop[tid] = a.reset_prop[tid]; a.nbobs_k[tid] = 0.0f; }
        if (blockIdx.x == 0) {
            if (tid < a.reset_ctrl_words && tid != C_FOLD) a.reset_ctrl[tid] = 0;
            if (tid == 0) a.reset_sweep_next[0] = 0;
        }
        __syncthreads();
    } else if (a.stop != nullptr && *a.stop) return;
    NEM_PHASE(0);
    if (a.mode == 1) {
        // (every block needs every class size for the empty-class flag; its own class's entries otherwise)
        if (tid < a.K) a.nbobs_k[tid] = (float)stat_sum(a.stats, tid, a.stats_ranks, a.stats_rank_stride);
        for (int t = kb * a.D + tid; t < ke * a.D; t += 1024)
            centers_ncem_entry(a.K, a.D, a.stats, a.stats_ranks, a.stats_rank_stride, a.center, a.nbobs_k, a.iner, t);
        __syncthreads();
    }
    NEM_PHASE(1);
    if (a.mode != 0) {
        disp_body(a.K, a.D, a.n_total, a.disper, a.propor, a.mode == 1, a.nbobs_k, a.iner, a.disp, a.prop, a.flags, kb, ke);
        __syncthreads();
    }
    NEM_PHASE(2);
    if (tid >= kb && tid < ke) a.nonuni[tid] = 0;
    __syncthreads();
    for (int t = kb * a.dpad + tid; t < ke * a.dpad; t += 1024) table_flag_general(a, t);
    __syncthreads();
    NEM_PHASE(3);
    for (int t = kb * a.dpad + tid; t < ke * a.dpad; t += 1024) table_entry(a, t);   // dpad and 1024 are multiples of 64
    NEM_PHASE(4);
    if (a.use_ff && a.ffq != nullptr) {
        // the uniform chain's fast-forward increments (nem_ff.hpp), one table per class that stays on that chain
        __syncthreads();                                 // (uni[k] / nonuni[k] of this block's classes are settled)
        for (int t = kb * 256 + tid; t < ke * 256; t += 1024) {
            const int k = t >> 8;
            if (a.nonuni[k] == 0) {
                const double l1 = a.uni[k].x, l0 = a.uni[k].y;
                uint32_t q0, q1;
                ff_entry(l1, -l0, t & 255, q0, q1);
                if (q0 == kFFInvalid) q1 = kFFInvalid;
                a.ffq[t] = make_uint2(q0, q1 - q0);
            }
        }
    }
    NEM_PHASE(5);
}

// ------------------------------------------------------------------------------------------
// Fuzzy NEM M-step.  The reference's sums are i-ordered float accumulators, so every (class, organism) chain
// stays on one lane and the K*D chains run in parallel; a wave holds 64 organisms of one class.  With so few
// waves (K*D/64) a wave is alone on its SIMD and the time is (instructions per family) x N, so:
//   * the N families are walked 64 at a time: one strided vector load brings the 64 memberships c_ik, two more
//     the 64 bit words of the wave's organisms, the next group's loads fly over this group's chain, and the
//     inner loop takes everything from registers (v_readlane -> SGPRs: c_ik is wave-uniform, and the two bit
//     words of a family ARE the wave's 64-bit lane mask "organism has the family");
//   * every chain gets its own wave (roles along blockIdx.x), each step = 3 v_readlane + an add + a select:
//       pass A  inertia for mu = 0 (sum of c_ik over the ones) / for mu = 1 (over the zeros) / the class total
//               N_k / the inertia for mu = 0.5 (the same for every organism) / the two order-free facts
//               ComputeMedian's tie rule needs (they do not depend on N_k, so they run here, beside the sums)
//       pass B  ComputeMedian's prefix scan over the zeros against N_k/2 (needs pass A's N_k).
//   (float)((double)a + (double)b) is the float sum a + b -- a double holds the exact sum of two floats' leading
//   2*24+2 bits, so the double rounding is innocuous -- which is how the mu = 0 / mu = 1 chains are float adds.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_f32(float v, int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); }
__device__ __forceinline__ uint64_t lane_mask(uint32_t lo, uint32_t hi, int j)
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)lo, j) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, j) << 32);
}

// one wave's view of the families: 64 memberships and the 64 lane masks at a time
struct FuzzyWalk {
    const float* c; const uint32_t *col_lo, *col_hi; int n, K, k, lane;
    float cv; uint32_t xl, xh;
    __device__ __forceinline__ void load(int i0, float& v, uint32_t& l, uint32_t& h) const
    {
        const int il = min(i0 + lane, n - 1);
        v = c[(size_t)il * K + k]; l = col_lo[il]; h = col_hi[il];
    }
    // group(cv, xl, xh, cnt): lane j of the three registers holds family i0 + j, cnt of them are valid
    template <typename G> __device__ __forceinline__ void groups(G&& group)
    {
        load(0, cv, xl, xh);
        int i0 = 0;
        for (; i0 + 64 <= n; i0 += 64) {
            float cn; uint32_t ln, hn;
            load(i0 + 64, cn, ln, hn);
            group(cv, xl, xh, 64);
            cv = cn; xl = ln; xh = hn;
        }
        if (i0 < n) group(cv, xl, xh, n - i0);
    }
    template <typename F> static __device__ __forceinline__ void lanes(int cnt, F&& f)
    {
        if (cnt == 64) {
#pragma unroll
            for (int j = 0; j < 64; j++) f(j);
        } else {
            for (int j = 0; j < cnt; j++) f(j);                          // j is wave-uniform
        }
    }
    template <typename F> __device__ __forceinline__ void run(F&& step)
    {
        groups([&](float v, uint32_t l, uint32_t h, int cnt) { lanes(cnt, [&](int j) { step(v, l, h, j); }); });
    }
};

__device__ __forceinline__ FuzzyWalk fuzzy_walk(int n, int npad, int K, int D, int k, int bx, const uint32_t* xw, const float* c)
{
    FuzzyWalk w;
    w.c = c; w.n = n; w.K = K; w.k = k; w.lane = threadIdx.x;
    w.col_lo = xw + (size_t)(min(bx * 64, D - 1) >> 5) * npad;           // organisms 64 bx .. +31 -> lanes 0..31
    w.col_hi = xw + (size_t)(min(bx * 64 + 32, D - 1) >> 5) * npad;      //           64 bx + 32 .. -> lanes 32..63
    return w;
}

// The sums that take c_ik from the families where the lane's organism has a given bit -- the inertia candidates, the
// median's prefix sum -- read that bit from the lane's OWN row of bits (`xt`, organism-major: one 64-bit word per 64
// families) and the 64 memberships from LDS, four per broadcast read: a step is a sign-extended bit, an AND and an
// add (the addend is c_ik or +0, and adding +0 leaves a sum that is >= +0 as it is), three instructions instead of
// three readlanes, an add and a select.
struct OwnWalk {
    const float* c; const uint64_t* row; int n, K, k, nw64, lane; float* sC;
    // group(i0, cnt, own): sC[0..cnt) holds the memberships of families i0 .. i0 + cnt - 1, bit j of `own` is the
    // lane's organism's bit for family i0 + j
    template <typename G> __device__ __forceinline__ void groups(G&& group)
    {
        // (two groups ahead: the bit words of a wave's 64 organisms are 64 different cache lines)
        float cn = c[(size_t)min(lane, n - 1) * K + k], cn2 = c[(size_t)min(64 + lane, n - 1) * K + k];
        uint64_t on = row[0], on2 = row[min(1, nw64 - 1)];
        int g = 0;
        for (int i0 = 0; i0 < n; i0 += 64, g++) {
            const float cv = cn;
            const uint64_t own = on;
            cn = cn2; on = on2;
            if (i0 + 128 < n) { cn2 = c[(size_t)min(i0 + 128 + lane, n - 1) * K + k]; on2 = row[g + 2]; }
            __syncthreads();                             // (one wave per block: the previous group's reads are done)
            sC[lane] = cv;
            __syncthreads();
            group(i0, min(64, n - i0), own);
        }
    }
    // acc += c for the families whose bit in `sel` is set, in family order
    __device__ __forceinline__ void masked_add(float& acc, int cnt, uint64_t sel) const
    {
        const uint32_t lo = (uint32_t)sel, hi = (uint32_t)(sel >> 32);
        if (cnt == 64) {
            const float4* s4 = reinterpret_cast<const float4*>(sC);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const float4 v = s4[q];
                const uint32_t w = q < 8 ? lo : hi;
                const int b = (4 * q) & 31;
                acc += __int_as_float(__float_as_int(v.x) & __builtin_amdgcn_sbfe((int)w, b, 1));
                acc += __int_as_float(__float_as_int(v.y) & __builtin_amdgcn_sbfe((int)w, b + 1, 1));
                acc += __int_as_float(__float_as_int(v.z) & __builtin_amdgcn_sbfe((int)w, b + 2, 1));
                acc += __int_as_float(__float_as_int(v.w) & __builtin_amdgcn_sbfe((int)w, b + 3, 1));
            }
        } else {
            for (int j = 0; j < cnt; j++)
                acc += ((sel >> j) & 1ull) ? sC[j] : 0.0f;
        }
    }
};

// roles along blockIdx.x: [0,DB) in0, [DB,2DB) in1, [2DB,3DB) last zero of weight >= EPSILON,
// [3DB,4DB) "some one has weight >= EPSILON", 4DB: N_k, 4DB+1: inertia for mu = 0.5
__device__ __forceinline__ void mstep_fuzzy_a_body(int n, int npad, int K, int D, const uint32_t* __restrict__ xw,
                                                      const uint64_t* __restrict__ xt, int nw64,
                                                      const float* __restrict__ c, float* __restrict__ nbobs_k,
                                                      float* __restrict__ in0_out, float* __restrict__ in1_out,
                                                      float* __restrict__ inh_k, int* __restrict__ lastz_out,
                                                      int* __restrict__ any1_out, const int* __restrict__ stop,
                                                      const int flags_only)
{
    if (stop != nullptr && *stop) return;
    const int DB = (D + 63) >> 6;
    const int role = blockIdx.x < 4 * DB ? blockIdx.x / DB : 4 + (int)blockIdx.x - 4 * DB;
    if (flags_only && role != 2 && role != 3) return;                    // (the sums are k_mstep_fuzzy_sums' then)
    const int bx = role < 4 ? blockIdx.x - role * DB : 0;
    const int k = blockIdx.y;
    const int d = bx * 64 + threadIdx.x;
    __shared__ float sC[64];
    OwnWalk ow{c, xt + (size_t)min(d, D - 1) * nw64, n, K, k, nw64, (int)threadIdx.x, sC};
    if (role == 4) {                                                     // N_k (nem_mod.c:1308): every membership
        float nk = 0.0f;
        ow.groups([&](int, int cnt, uint64_t) { ow.masked_add(nk, cnt, ~0ull); });
        if (threadIdx.x == 0) nbobs_k[k] = nk;
        return;
    }
    if (role == 5) {
        // nem_mod.c:1683 with |x - 0.5| = 0.5: inh = (float)((double)inh + (double)c * 0.5).  When c/2 is a float
        // (always, but for an odd subnormal) that is the float sum inh + c/2; groups holding such a c take the
        // double form.
        float inh = 0.0f;
        ow.groups([&](int, int cnt, uint64_t) {
            const float cv = sC[threadIdx.x];
            const float hv = cv * 0.5f;
            const bool halves = __ballot(threadIdx.x < (unsigned)cnt && hv * 2.0f != cv) == 0;
            __syncthreads();
            if (halves) {
                sC[threadIdx.x] = hv;
                __syncthreads();
                ow.masked_add(inh, cnt, ~0ull);
            } else {
                for (int j = 0; j < cnt; j++) inh = (float)((double)inh + (double)sC[j] * 0.5);
            }
        });
        if (threadIdx.x == 0) inh_k[k] = inh;
        return;
    }
    if (role == 2 || role == 3) {
        // The two facts ComputeMedian's tie rule needs (nem_mod.c:1470-1483) do not depend on the order of the
        // families: the index of the last zero whose weight is >= EPSILON (pass B compares it with where its chain
        // crossed) and whether some one has such a weight -- a ballot of the weights per 64 families against the
        // lane's own word of bits.
        int last = -1;
        bool some = false;
        ow.groups([&](int i0, int cnt, uint64_t own) {
            const float cv = sC[threadIdx.x];
            uint64_t big = __ballot(threadIdx.x < (unsigned)cnt && !((double)cv < kEpsilonD));
            const uint64_t z = ~own & big, o = own & big;
            if (z != 0) last = i0 + 63 - __clzll((long long)z);
            some |= (o != 0);
        });
        if (d < D) {
            if (role == 2) lastz_out[k * D + d] = last;
            else any1_out[k * D + d] = some ? 1 : 0;
        }
        return;
    }
    const uint64_t flip = role == 1 ? ~0ull : 0ull;                      // role 1 sums over the zeros
    float acc = 0.0f;
    ow.groups([&](int, int cnt, uint64_t own) { ow.masked_add(acc, cnt, own ^ flip); });   // nem_mod.c:1683, |x - mu| = 1
    if (d < D) (role == 0 ? in0_out : in1_out)[k * D + d] = acc;
}

__device__ __forceinline__ void mstep_fuzzy_b_body(int n, int npad, int K, int D, const uint32_t* __restrict__ xw,
                                                      const uint64_t* __restrict__ xt, int nw64,
                                                      const float* __restrict__ c, const float* __restrict__ nbobs_k,
                                                      const float* __restrict__ in0, const float* __restrict__ in1,
                                                      const float* __restrict__ inh_k, const int* __restrict__ lastz,
                                                      const int* __restrict__ any1, float* __restrict__ center,
                                                      float* __restrict__ iner, const int* __restrict__ stop)
{
    if (stop != nullptr && *stop) return;
    const int k = blockIdx.y;
    const int d = blockIdx.x * 64 + threadIdx.x;
    const int t = k * D + min(d, D - 1);
    FuzzyWalk w = fuzzy_walk(n, npad, K, D, k, blockIdx.x, xw, c);
    const float nk = nbobs_k[k];
    if (!((double)nk > kEpsilonD)) {
        // "empty" class (nem_mod.c:1404-1408): the centre is kept, and EstimLaplaceIner (:1669-1686) still runs
        // against that old centre -- the class may hold weights between 0 and EPSILON, which InerToDisp* then
        // turns into a dispersion (they test N_K > 0, not > EPSILON)
        const float mu = center[t];
        const double a1 = fabs((double)(1.0f - mu)), a0 = fabs((double)(0.0f - mu));
        float in = 0.0f;
        w.run([&](float cv, uint32_t xl, uint32_t xh, int j) {
            const float ci = lane_f32(cv, j);
            const bool one = __builtin_amdgcn_inverse_ballot_w64(lane_mask(xl, xh, j));
            in = (float)((double)in + (double)ci * (one ? a1 : a0));     // :1683
        });
        if (d < D) iner[t] = in;
        return;
    }
    // Only the zeros' chain decides (nem_mod.c:1439-1497): once the cumulated weight of the zeros reaches N_k/2
    // the median sits among them, otherwise it is a one (or midway between two ones) whatever the ones' chain
    // does.  crossed = lane mask "the chain has reached N_k/2" (it is frozen from there on).
    const float half = nk / 2;                           // nem_mod.c:1439
    const double half_eps = (double)half + kEpsilonD;    // nem_mod.c:1464
    // The weights are >= 0, so a chain that has reached N_k/2 stays there: the walk runs the bare chain (add, select
    // on the family's lane mask) over 64 families, tests the crossing once per group, and only a group in which
    // some lane crossed is walked again, from the saved start, to find that lane's family and value.
    float run = 0.0f, cum = 0.0f;
    int istar = n;                                       // family at which the chain crossed
    uint64_t crossed = 0;
    __shared__ float sC[64];
    OwnWalk ow{c, xt + (size_t)min(d, D - 1) * nw64, n, K, k, nw64, (int)threadIdx.x, sC};
    ow.groups([&](int i0, int cnt, uint64_t own) {
        const float start = run;
        ow.masked_add(run, cnt, ~own);                   // the zeros' memberships, in family order
        const uint64_t newly = __ballot(!(run < half)) & ~crossed;
        if (newly != 0) {
            float cv; uint32_t xl, xh;                   // this group again, family by family (lane j = family i0 + j)
            w.load(i0, cv, xl, xh);
            float r2 = start;
            int cj = -1;
            uint64_t seen = ~newly;
            FuzzyWalk::lanes(cnt, [&](int j) {
                const uint64_t one = lane_mask(xl, xh, j);
                const float nx = r2 + lane_f32(cv, j);
                const uint64_t now = __ballot(!(nx < half)) & ~one & ~seen;
                r2 = __builtin_amdgcn_inverse_ballot_w64(one) ? r2 : nx;
                cum = __builtin_amdgcn_inverse_ballot_w64(now) ? nx : cum;
                cj = __builtin_amdgcn_inverse_ballot_w64(now) ? j : cj;
                seen |= now;
            });
            istar = cj >= 0 ? i0 + cj : istar;
            crossed |= newly;
        }
    });
    const bool ph0 = __builtin_amdgcn_inverse_ballot_w64(crossed);
    const bool gt0 = (double)cum > half_eps;             // cum is the value at the crossing
    float mu;
    if (ph0) {                                           // median position among the zeros
        const bool next0 = lastz[t] > istar;             // a zero of weight >= EPSILON follows the crossing
        if (gt0 || next0) mu = 0.0f;                     // x_med = 0 (or midway to another 0)
        else if (any1[t]) mu = 0.5f;                     // midway to the first one with weight
        else mu = 0.0f;                                  // reference runs off the array here (UB)
    } else {
        mu = 1.0f;                                       // x_med = 1 (or midway to another 1)
    }
    if (d < D) {
        center[t] = mu;
        iner[t] = (mu == 0.0f) ? in0[t] : (mu == 1.0f ? in1[t] : inh_k[k]);
    }
}

// ------------------------------------------------------------------------------------------
// The fuzzy M-step's sums, split along the families.  Each of the K*(2D+2) sums of pass A and the K*D prefix scans
// of pass B is an i-ordered float accumulator over all N families (nem_mod.c:1303-1313, 1677-1686, 1448-1458) --
// N dependent adds, whoever runs them.  But between two powers of two a float accumulator moves on a fixed grid
// and the chain is a prefix sum of integers (nem_chain.hpp); the sum is then a matter of a scan.  Here ONE WAVE
// owns a chain and takes it 256 families at a time: every lane turns its four memberships into grid increments,
// a DPP scan places them, the first element the integer form cannot take (the next binade, an exact tie, a sum
// that is not ready) is found with a ballot and stepped -- with a short burst behind it -- by the reference's own
// float add.  Float inputs make this simpler than the criteria's chains: c * 2^s is exact, so "near a tie" is
// "exactly a tie".  (float)((double)a + (double)b) is the float sum, so all four kinds of sums are float adds.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);   // row_shr:8: prefix inside each row of 16
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// LDS written by some lanes of a wave and read by others of the SAME wave: the hardware keeps a wave's LDS
// operations in order; this only stops the compiler from moving them across (no block barrier: the chains of a
// block's waves advance independently)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kEPL = 4;                                  // families per lane and window of the chain kernels
constexpr int kWin = 64 * kEPL;                          // families per window

struct WaveChain {
    float* sx;                                           // LDS: the window's kWin addends, family order
    float acc;                                           // the accumulator (the same value in every lane)
    int burst;
    // sequential float adds of sx[p .. p + count): the reference's own arithmetic, sixteen addends fetched ahead
    __device__ __forceinline__ void steps(int p, int count)
    {
        int i = 0;
        for (; i + 16 <= count; i += 16) {
            float v[16];
#pragma unroll
            for (int t = 0; t < 16; t++) v[t] = sx[p + i + t];
#pragma unroll
            for (int t = 0; t < 16; t++) acc = acc + v[t];
        }
        for (; i < count; i++) acc = acc + sx[p + i];
    }
    // the window: wn (<= kWin) addends, lane l holds x[0 .. kEPL) = addends kEPL l .. (the same values are in sx).
    // Branch-free per element: what an element adds and whether the integer form can take it are selects.
    __device__ __forceinline__ void window(const float (&x)[kEPL], int wn)
    {
        const int e0 = kEPL * (int)(threadIdx.x & 63);
        int pos = 0;
        while (pos < wn) {
            const uint32_t ab = __float_as_uint(acc);
            const int E = (int)((ab >> 23) & 255u);
            if ((ab >> 31) != 0u || E < 24 || E > 253) {      // negative, tiny, zero or not finite: step
                const int cnt = min(burst, wn - pos);
                steps(pos, cnt);
                pos += cnt;
                burst = nemchain::next_burst(burst, -1);
                continue;
            }
            const float sc = __uint_as_float((uint32_t)(127 + 150 - E) << 23);    // 1 / ulp(acc): 2^(150 - E), a normal float
            const uint32_t M0 = (ab & 0x7fffffu) | 0x800000u;
            constexpr uint32_t top = 1u << 24;
            if (pos == 0) {
                // most windows hold nothing the integer form cannot take and end inside the binade they began in:
                // then all that is needed is the sum of the increments (no scan, no search for the first stop)
                uint32_t tot = 0; bool clean = true;
#pragma unroll
                for (int j = 0; j < kEPL; j++) {
                    const float q = x[j] * sc;           // (addends behind wn are -0: increment 0)
                    const float fl = floorf(q);
                    const float fr = q - fl;
                    clean = clean & (q >= 0.0f) & (q < 33554432.0f) & (fr != 0.5f);
                    tot += (uint32_t)fl + (fr > 0.5f ? 1u : 0u);
                }
                const bool small = tot < (1u << 17);     // (64 lanes of less than 2^17 cannot wrap 32 bits)
                if (__ballot(!(clean & small)) == 0ull) {
                    const uint32_t Mt = M0 + (uint32_t)wave_reduce_add((int)tot);
                    if (Mt <= top) {
                        acc = __uint_as_float(Mt >= top ? ((uint32_t)(E + 1) << 23) : (((uint32_t)E << 23) | (Mt & 0x7fffffu)));
                        burst = nemchain::next_burst(burst, wn);
                        return;
                    }
                }
            }
            uint32_t inc[kEPL];
            int stop = kEPL;                             // first of mine the integer form cannot take
            uint32_t lsum = 0;
#pragma unroll
            for (int j = 0; j < kEPL; j++) {
                const int e = e0 + j;
                const bool act = (e >= pos) & (e < wn);
                const float q = x[j] * sc;               // exact (a power-of-two scaling; an overflow gives inf)
                const float fl = floorf(q);
                const float fr = q - fl;                 // exact
                // takeable: 0 <= q < 2^25 (a shrinking sum, NaN or inf fail the compares) and not a tie (its rounding
                // depends on the parity of M)
                const bool ok = (q >= 0.0f) & (q < 33554432.0f) & (fr != 0.5f);
                const uint32_t v = (uint32_t)fl + (fr > 0.5f ? 1u : 0u);
                const bool first_bad = act & !ok & (stop == kEPL);
                stop = first_bad ? j : stop;
                inc[j] = (act & ok & (stop == kEPL)) ? v : 0u;
                lsum += inc[j];
            }
            const uint32_t incl = wave_scan_incl(lsum);
            uint32_t M = M0 + incl - lsum;
            int cand = INT_MAX; uint32_t candM = 0;
#pragma unroll
            for (int j = 0; j < kEPL; j++) {
                const int e = e0 + j;
                const bool act = (e >= pos) & (e < wn);
                const bool hit = act & (cand == INT_MAX) & ((j == stop) | (M >= top) | (M + inc[j] > top));
                candM = hit ? M : candM;
                cand = hit ? e : cand;
                M += inc[j];
            }
            const uint64_t who = __ballot(cand != INT_MAX);
            if (who == 0ull) {                           // the rest of the window went through in integer form
                const uint32_t Mt = (uint32_t)__builtin_amdgcn_readlane((int)(M0 + incl), 63);
                acc = __uint_as_float(Mt >= top ? ((uint32_t)(E + 1) << 23) : (((uint32_t)E << 23) | (Mt & 0x7fffffu)));
                burst = nemchain::next_burst(burst, wn - pos);
                pos = wn;
                break;
            }
            const int first = (int)__ffsll((long long)who) - 1;      // (a lane behind the first stop holds a wrong M: ignored)
            const int p = __builtin_amdgcn_readlane(cand, first);
            const uint32_t Mb = (uint32_t)__builtin_amdgcn_readlane((int)candM, first);
            acc = __uint_as_float(Mb >= top ? ((uint32_t)(E + 1) << 23) : (((uint32_t)E << 23) | (Mb & 0x7fffffu)));
            const int cnt = min(burst, wn - p);
            steps(p, cnt);
            burst = nemchain::next_burst(burst, p - pos);
            pos = p + cnt;
        }
    }
};

// One window's inputs of a lane: its kEPL memberships of class k (class-major copy ct[K][npad]) for families
// i0 + kEPL lane .. and the word of organism row `row` that holds their bits.  (The word is handed on as loaded:
// shifting it here would make the fetch wait for its own load, and the fetch runs ahead of its use.)
struct FuzzyIn { float4 c[kEPL / 4]; uint64_t word; };
__device__ __forceinline__ void fuzzy_fetch(const float* __restrict__ ctk, const uint64_t* __restrict__ row, int i0, int lane,
                                            int nw64, FuzzyIn& in)
{
    const float4* p = reinterpret_cast<const float4*>(ctk + i0 + kEPL * lane);
#pragma unroll
    for (int t = 0; t < kEPL / 4; t++) in.c[t] = p[t];
    const int w = min((i0 >> 6) + ((kEPL * lane) >> 6), nw64 - 1);
    in.word = row != nullptr ? row[w] : ~0ull;
}
__device__ __forceinline__ uint32_t fuzzy_bits(uint64_t word, int lane) { return (uint32_t)(word >> ((kEPL * lane) & 63)) & ((1u << kEPL) - 1u); }
__device__ __forceinline__ void fuzzy_unpack(const FuzzyIn& in, float (&x)[kEPL])
{
#pragma unroll
    for (int t = 0; t < kEPL / 4; t++) { x[4 * t] = in.c[t].x; x[4 * t + 1] = in.c[t].y; x[4 * t + 2] = in.c[t].z; x[4 * t + 3] = in.c[t].w; }
}
__device__ __forceinline__ void fuzzy_stage(float* sx, int lane, const float (&x)[kEPL])
{
    float4* d = reinterpret_cast<float4*>(sx + kEPL * lane);
#pragma unroll
    for (int t = 0; t < kEPL / 4; t++) d[t] = make_float4(x[4 * t], x[4 * t + 1], x[4 * t + 2], x[4 * t + 3]);
}

// c [n][K] -> ct [K][ctpad] (zeros behind n; ctpad: n rounded up to whole windows)
__device__ __forceinline__ void transpose_c_body(int n, int ctpad, int K, const float* __restrict__ c, float* __restrict__ ct,
                                                 const int* __restrict__ stop)
{
    if (stop != nullptr && *stop) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ctpad) return;
    for (int k = 0; k < K; k++) ct[(size_t)k * ctpad + i] = i < n ? c[(size_t)i * K + k] : 0.0f;
}

// ------------------------------------------------------------------------------------------
// The fuzzy sums with one LANE per chain again, at the chain's own speed.  A chain is N dependent float adds on one
// lane: ~10 cycles each whatever else the wave does -- 96 us for 20 000 families -- and the round-1 kernels spent
// three instructions per family on it (bit -> mask, and, add) at a lone wave's one instruction per ~8 cycles.  Here
// the lane that owns the chain does the adds and nothing else: eight PRODUCER waves of the block (on all the CU's
// SIMDs) turn the next 64 families into the addends -- c_ik where the organism's bit says so, +0 elsewhere; adding
// +0 leaves a sum that is >= +0 as it is -- and hand them over through LDS, 64 organisms x 256 families per hand-over,
// double-buffered, one barrier per hand-over; the CONSUMER wave reads its lanes' rows sixteen bytes at a time and adds
// in family order: 1.25 instructions per family.  The zeros' chains leave their value every 64 families (`chk`):
// ComputeMedian's prefix scan over the zeros IS that chain, so the median kernel looks up the window in which a
// chain reached N_k / 2 and walks those 64 families again instead of all of them.  The two order-free facts of the
// tie rule (last zero / some one of weight >= EPSILON) ride with the producers.
// roles along blockIdx.x: [0, DB) the ones, [DB, 2 DB) the zeros, 2 DB: N_k, 2 DB + 1: inertia for mu = 1/2
// ------------------------------------------------------------------------------------------
constexpr int kPcWaves = 10;                            // a consumer wave, eight producer waves and an idle one per block (see below)
constexpr int kPcStride = 260;                           // floats per organism row of a hand-over (256 + 4: rows stay 16-byte aligned)

struct PcIn { uint32_t lo, hi; float c; };

template <int J0, int J1>
__device__ __forceinline__ void pc_produce(float* __restrict__ rowA, const PcIn& in)
{
    // addends of families J0 .. J1-1 of the window for this lane's organism
#pragma unroll
    for (int j = J0; j < J1; j += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int jj = j + u;
            const int cj = __builtin_amdgcn_readlane(__float_as_int(in.c), jj);            // c of family jj: wave-uniform
            const int m = __builtin_amdgcn_sbfe((int)(jj < 32 ? in.lo : in.hi), jj & 31, 1);  // all ones where the bit is set
            v[u] = __int_as_float(cj & m);
        }
        *reinterpret_cast<float4*>(rowA + j) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

__device__ __forceinline__ void mstep_fuzzy_pc_body(const FuzzyArgs& a)
{
    if (a.stop != nullptr && *a.stop) return;
    const int n = a.n, D = a.D;
    const int DB = (D + 63) >> 6;
    const int k = blockIdx.y;
    const int role = (int)blockIdx.x < 2 * DB ? (int)blockIdx.x / DB : 2 + (int)blockIdx.x - 2 * DB;
    const int bx = role < 2 ? (int)blockIdx.x - role * DB : 0;
    // The block's waves go to the CU's four SIMDs in turn (0 4 8 | 1 5 9 | 2 6 | 3 7): the consumer's chain is a string
    // of dependent adds, and every instruction another wave issues on its SIMD can hold the next add back a few cycles.
    // So the consumer is hardware wave 2 and hardware wave 6, its only SIMD mate, does nothing but meet the barriers;
    // wv: 0 the consumer, 1..8 the producers, 9 the idle wave.
    const int lane = threadIdx.x & 63, hw = threadIdx.x >> 6;
    const int wv = hw == 2 ? 0 : hw == 6 ? 9 : hw < 2 ? hw + 1 : hw < 6 ? hw : hw - 1;   // hw 0 1 3 4 5 7 8 9 -> 1 .. 8
    const int nwin = (n + 63) >> 6;
    const int d = bx * 64 + lane;
    const float* __restrict__ ctk = a.ct + (size_t)k * a.ctpad;
    const uint64_t* __restrict__ row = a.xt + (size_t)min(d, D - 1) * a.nw64;
    const uint32_t flip = role == 1 ? ~0u : 0u;          // role 1 sums over the zeros
    __shared__ float4 sA4[2][64 * kPcStride / 4];
    auto fetch = [&](int w) -> PcIn {                    // window w as a producer lane sees it (zeros past the end)
        PcIn r{0u, 0u, 0.0f};
        if (w < nwin) {
            const uint64_t own = row[w];
            r.lo = (uint32_t)own ^ flip; r.hi = (uint32_t)(own >> 32) ^ flip;
            const int i = 64 * w + lane;
            r.c = i < n ? ctk[i] : 0.0f;
        }
        return r;
    };
    if (role >= 2) {
        // ---- N_k (nem_mod.c:1308: every membership) and the inertia for mu = 1/2 (:1683 with |x - 0.5| = 0.5:
        // inh = (float)((double)inh + (double)c * 0.5) -- the float sum inh + c/2 whenever c/2 is a float, i.e. always
        // but for an odd subnormal c).  One chain per class: the consumer wave alone, 256 memberships per step (one
        // 16-byte load per lane, requested a step ahead), through LDS to every lane, added in family order.
        if (wv != 0) return;
        float acc = 0.0f;
        const int nstep = (n + 255) >> 8;                // (ct is zero from n up to its padded length, a multiple of 256)
        const float4* __restrict__ c4 = reinterpret_cast<const float4*>(ctk);
        __shared__ float4 sC4[2][64];
        float4 cur4 = nstep > 0 ? c4[lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int H = 0; H < nstep; H++) {
            const float4 nxt4 = H + 1 < nstep ? c4[64 * (H + 1) + lane] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const int cnt = min(256, n - 256 * H);
            bool plain = true;
            float4 put = cur4;
            if (role == 3) {
                // (c/2 is not a float only for an odd subnormal c, less than 2^-127: against a sum of 2^-100 or more
                //  -- spacing 2^-123 at least -- it is below an eighth of the spacing, the double form returns the
                //  sum unchanged, and so does the float add of the rounded half: the double form matters only while
                //  the sum itself is tiny)
                const float4 hv = make_float4(cur4.x * 0.5f, cur4.y * 0.5f, cur4.z * 0.5f, cur4.w * 0.5f);
                const bool odd = hv.x * 2.0f != cur4.x || hv.y * 2.0f != cur4.y || hv.z * 2.0f != cur4.z || hv.w * 2.0f != cur4.w;
                plain = acc >= 0x1p-100f || __ballot(odd) == 0;           // (the padding is zeros: never odd)
                if (plain) put = hv;
            }
            sC4[H & 1][lane] = put;
            __builtin_amdgcn_s_waitcnt(0xc07f);          // (lgkmcnt(0): one wave writes and reads its own LDS lines)
            __builtin_amdgcn_wave_barrier();
            if (plain) {
#pragma unroll 16
                for (int q = 0; q < 64; q++) {
                    const float4 v = sC4[H & 1][q];
                    acc += v.x; acc += v.y; acc += v.z; acc += v.w;
                }
            } else {
                const float* sCf = reinterpret_cast<const float*>(sC4[H & 1]);
                for (int j = 0; j < cnt; j++) acc = (float)((double)acc + (double)sCf[j] * 0.5);
            }
            cur4 = nxt4;
        }
        if (lane == 0) (role == 2 ? a.nbobs_k : a.inh_k)[k] = acc;
        return;
    }
    // ---- the ones' / the zeros' chains of 64 organisms.  A hand-over is 256 families (four 64-family sub-windows):
    // a barrier of the block's nine waves costs ~0.3 us, as much as a sub-window's work.  Producer wave p makes
    // families [32 h, 32 h + 32) of sub-window s, (s, h) = ((p - 1) / 2, (p - 1) % 2): one word of bits and one vector
    // of memberships per hand-over, requested a hand-over ahead.
    const int DW = DB * 64;
    float* __restrict__ chk = (role == 1 && a.chk != nullptr) ? a.chk + (size_t)k * (nwin + 1) * DW + d : nullptr;
    float acc = 0.0f;
    int last = -1; int some = 0;                         // (producers with h = 0: the tie rule's two facts of their sub-windows)
    const bool prod = wv >= 1 && wv <= 8;
    const int ps = prod ? (wv - 1) >> 1 : 0, ph = prod ? (wv - 1) & 1 : 0;
    const int nhand = (nwin + 3) >> 2;
    auto produce = [&](int buf, int w, const PcIn& in) {  // sub-window w of the families, into hand-over buffer buf
        if (w >= nwin) return;                           // (nothing there: the consumer does not read it)
        float* rowA = reinterpret_cast<float*>(sA4[buf]) + lane * kPcStride + 64 * ps;
        if (ph == 0) {
            // last zero of weight >= EPSILON / some one of such weight (nem_mod.c:1470-1483): order-free, from the
            // sub-window's 64 weights against the lane's own bits (held flipped in the zeros' role)
            const int cnt = min(64, n - 64 * w);
            const uint64_t big = __ballot(lane < cnt && !((double)in.c < kEpsilonD));
            const uint64_t sel = (((uint64_t)in.hi << 32) | in.lo) & big;  // zeros' role: the zeros; ones' role: the ones
            if (role == 1) { if (sel != 0) last = 64 * w + 63 - __clzll((long long)sel); }
            else some |= (sel != 0);
            pc_produce<0, 32>(rowA, in);
        } else pc_produce<32, 64>(rowA, in);
    };
    // Hand-overs without a barrier of the whole block (one costs the consumer ~0.3 us, a quarter of a hand-over's adds):
    // two counters in LDS.  s_ready[b]: producer waves that have finished writing buffer b, over all its tenants;
    // s_done: hand-overs the consumer has finished reading.  A producer writes its slice, waits for its LDS writes
    // (workgroup-scope release), bumps s_ready; the consumer reads a buffer once all eight producers of that tenant
    // are in (acquire), and says so in s_done when its last read has come back; a producer does not touch a buffer
    // before the consumer is done with its previous tenant.  The producers run ahead and sleep on s_done; the
    // consumer's check is one LDS read.  Spins are bounded (a wait that runs out goes on with whatever is there), and a
    // hand-over that was not delivered is a FAULT, not a result: the producers' counters are cumulative, so the consumer
    // sees at the end whether every hand-over arrived and, if not, raises the engine's FLAG_FAULT word, which the host
    // turns into NEMGPU_E_INTERNAL -- a broken hand-over ends the kernel, reported, not hung and not silently wrong.
    __shared__ int s_ready[2], s_done;
    if (threadIdx.x == 0) { s_ready[0] = 0; s_ready[1] = 0; s_done = -1; }
    __syncthreads();
    constexpr int kSpinCap = 1 << 16;                    // (producers, with s_sleep: ~10 ms per hand-over)
    constexpr int kConsumerSpinCap = 1 << 18;            // (per hand-over: ~10 ms, a thousand times the longest legitimate wait)
    auto raise_fault = [&]() { if (lane == 0 && a.fault != nullptr) atomicOr(a.fault, 1); };
#ifdef NEM_PHASE_PROF
    unsigned long long t_work = 0, t_begin = wall_clock64();
#endif
    if (prod) {
        PcIn cur = fetch(ps), nxt = fetch(4 + ps);
        for (int H = 0; H < nhand; H++) {
            // buffer H & 1 held hand-over H - 2
            // (round 2's wait, to the letter.  Round 3 had the producers look at a fault flag here and leave the loop
            //  on it: a producer's hand-over takes about as long as the consumer's 256 adds, so every instruction in
            //  this loop is on the consumer's critical path -- the M-step went from 139 to 153 us.  A wait that runs
            //  out now just goes on -- everything is bounded -- and the consumer reports the fault at the end.)
            for (int spin = 0; spin < kSpinCap && __hip_atomic_load(&s_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < H - 2; spin++)
                __builtin_amdgcn_s_sleep(4);
            if (a.inject == 1 && wv == 1 && H == 1 && blockIdx.x == 0 && blockIdx.y == 0) continue;   // (test hook: a producer that never delivers)
            produce(H & 1, 4 * H + ps, cur);
            if (lane == 0) __hip_atomic_fetch_add(&s_ready[H & 1], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            cur = nxt;
            nxt = fetch(4 * (H + 2) + ps);               // (for the hand-over after the next)
        }
    } else if (wv == 0) {
        for (int H = 0; H < nhand; H++) {
#ifdef NEM_PHASE_PROF
            const unsigned long long t_a = wall_clock64();
#endif
            const int need = 8 * ((H >> 1) + 1);         // all eight producers of this tenant of the buffer
            // (This wait is round 2's, to the letter: nothing of it is looked at afterwards.  With the spin count tested
            //  behind the loop -- round 3's fault report -- the compiler kept an LDS operation outstanding across the
            //  loop's exit and throttled the sixteen reads below with four `s_waitcnt lgkmcnt(14)`: +0.16 us per
            //  hand-over, 139 -> 153 us per M-step.  A hand-over that never completes is found behind the loop: the
            //  producers' counters are cumulative.)
            for (int spin = 0; spin < kConsumerSpinCap && __hip_atomic_load(&s_ready[H & 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need; spin++) { }
            // two register sets: the next sub-window's 64 addends are read from LDS while this one's are added
            const float4* r4 = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(sA4[H & 1]) + lane * kPcStride);
            float4 vA[16], vB[16];
#pragma unroll
            for (int q = 0; q < 16; q++) vA[q] = r4[q];
#pragma unroll
            for (int sw = 0; sw < 4; sw++) {
                const int g = 4 * H + sw;
                if (g < nwin) {
                    if (chk != nullptr) chk[(size_t)g * DW] = acc;
                    if (sw < 3 && g + 1 < nwin) {
#pragma unroll
                        for (int q = 0; q < 16; q++) (sw & 1 ? vA : vB)[q] = r4[16 * (sw + 1) + q];
                    }
#pragma unroll
                    for (int q = 0; q < 16; q++) {                       // family order
                        const float4 v = (sw & 1 ? vB : vA)[q];
                        acc += v.x; acc += v.y; acc += v.z; acc += v.w;
                    }
                }
            }
            // (every read of the buffer has come back: its values are in the sums)
            if (lane == 0) __hip_atomic_store(&s_done, H, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef NEM_PHASE_PROF
            t_work += wall_clock64() - t_a;
#endif
        }
        // every hand-over delivered?  (tenants of buffer 0: hand-overs 0, 2, ...; of buffer 1: 1, 3, ...; eight producers each)
        if (__hip_atomic_load(&s_ready[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 8 * ((nhand + 1) >> 1) ||
            __hip_atomic_load(&s_ready[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 8 * (nhand >> 1)) raise_fault();
    }
#ifdef NEM_PHASE_PROF
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && wv == 0) { g_phase[10] = t_work; g_phase[24] = wall_clock64() - t_begin; }
#endif
    // the facts of the four sub-window series: the latest zero, any one
    __shared__ int sFact[4][64];
    if (prod && ph == 0) sFact[ps][lane] = role == 1 ? last : some;
    __syncthreads();
    if (wv == 0) {
        if (chk != nullptr) chk[(size_t)nwin * DW] = acc;
        if (d < D) {
            (role == 0 ? a.in0 : a.in1)[k * D + d] = acc;
            const int f0 = sFact[0][lane], f1 = sFact[1][lane], f2 = sFact[2][lane], f3 = sFact[3][lane];
            if (role == 1) a.lastz[k * D + d] = max(max(f0, f1), max(f2, f3));
            else a.any1[k * D + d] = (f0 | f1 | f2 | f3) ? 1 : 0;
        }
    }
}

// the medians from the zeros' checkpoints: which window reached N_k / 2, then that window's families again
__device__ __forceinline__ void mstep_fuzzy_med2_body(const FuzzyArgs& a)
{
    if (a.stop != nullptr && *a.stop) return;
    const int n = a.n, K = a.K, D = a.D;
    const int k = blockIdx.y;
    const int lane = threadIdx.x;
    const int d = blockIdx.x * 64 + lane;
    const int t = k * D + min(d, D - 1);
    const float nk = a.nbobs_k[k];
    if (!((double)nk > kEpsilonD)) {
        // "empty" class (nem_mod.c:1404-1408): the centre is kept, and EstimLaplaceIner (:1669-1686) still runs
        // against that old centre (see mstep_fuzzy_b_body)
        FuzzyWalk w = fuzzy_walk(n, a.npad, K, D, k, blockIdx.x, a.xw, a.c);
        const float mu = a.center[t];
        const double a1 = fabs((double)(1.0f - mu)), a0 = fabs((double)(0.0f - mu));
        float in = 0.0f;
        w.run([&](float cv, uint32_t xl, uint32_t xh, int j) {
            const float ci = lane_f32(cv, j);
            const bool one = __builtin_amdgcn_inverse_ballot_w64(lane_mask(xl, xh, j));
            in = (float)((double)in + (double)ci * (one ? a1 : a0));     // :1683
        });
        if (d < D) a.iner[t] = in;
        return;
    }
    const int DB = (D + 63) >> 6, DW = DB * 64, nwin = (n + 63) >> 6;
    const float half = nk / 2;                           // nem_mod.c:1439
    const double half_eps = (double)half + kEpsilonD;    // nem_mod.c:1464
    const float* __restrict__ ck = a.chk + (size_t)k * (nwin + 1) * DW + d;
    // the weights are >= 0: a chain that has reached N_k / 2 stays there -- the first window at whose end it has, by
    // bisection over the (non-decreasing) checkpoints; a NaN anywhere counts as reached, as in the walk
    int gstar = -1;
    if (!(ck[(size_t)nwin * DW] < half)) {
        int lo = 0, hi = nwin - 1;                       // the answer is in [lo, hi]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (!(ck[(size_t)(mid + 1) * DW] < half)) hi = mid; else lo = mid + 1;
        }
        gstar = lo;
    }
    float cum = 0.0f;
    int istar = n;
    const bool ph0 = gstar >= 0;
    if (ph0) {
        float r2 = ck[(size_t)gstar * DW];
        const uint64_t own = a.xt[(size_t)min(d, D - 1) * a.nw64 + gstar];
        const float4* __restrict__ cw4 = reinterpret_cast<const float4*>(a.ct + (size_t)k * a.ctpad + 64 * gstar);
        const int cnt = min(64, n - 64 * gstar);
        // the window's 64 memberships first (sixteen independent loads), then the walk from registers
        float4 cv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) cv[q] = cw4[q];
        bool found = false;
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const float cj = j % 4 == 0 ? cv[j / 4].x : j % 4 == 1 ? cv[j / 4].y : j % 4 == 2 ? cv[j / 4].z : cv[j / 4].w;
            const bool zero = j < cnt && !((own >> j) & 1ull);           // (a one is not in this chain)
            const float nx = r2 + cj;
            const bool hit = zero && !found && !(nx < half);
            cum = hit ? nx : cum;
            istar = hit ? 64 * gstar + j : istar;
            found = found || hit;
            r2 = (zero && !found) ? nx : r2;
        }
    }
    const bool gt0 = (double)cum > half_eps;             // cum is the value at the crossing
    float mu;
    if (ph0) {                                           // median position among the zeros
        const bool next0 = a.lastz[t] > istar;           // a zero of weight >= EPSILON follows the crossing
        if (gt0 || next0) mu = 0.0f;                     // x_med = 0 (or midway to another 0)
        else if (a.any1[t]) mu = 0.5f;                   // midway to the first one with weight
        else mu = 0.0f;                                  // reference runs off the array here (UB)
    } else {
        mu = 1.0f;                                       // x_med = 1 (or midway to another 1)
    }
    if (d < D) {
        a.center[t] = mu;
        a.iner[t] = (mu == 0.0f) ? a.in0[t] : (mu == 1.0f ? a.in1[t] : a.inh_k[k]);
    }
}

// pass A: chains [0, D) inertia for mu = 0 (the ones), [D, 2D) for mu = 1 (the zeros), 2D: N_k, 2D + 1: mu = 1/2
__device__ __forceinline__ void mstep_fuzzy_sums_body(const FuzzyArgs& a)
{
    if (a.stop != nullptr && *a.stop) return;
    __shared__ float sx_all[kFuzzyWaves][kWin];
    const int k = blockIdx.y, D = a.D, lane = threadIdx.x & 63;
    const int chain = blockIdx.x * kFuzzyWaves + (threadIdx.x >> 6);             // one chain per wave
    if (chain >= 2 * D + 2) return;
    float* sx = sx_all[threadIdx.x >> 6];
    const int role = chain < D ? 0 : chain < 2 * D ? 1 : chain - 2 * D + 2;      // 0 ones, 1 zeros, 2 all, 3 halves
    const int d = role == 0 ? chain : role == 1 ? chain - D : 0;
    const uint64_t* row = role < 2 ? a.xt + (size_t)d * a.nw64 : nullptr;
    const float* ctk = a.ct + (size_t)k * a.ctpad;
    WaveChain wc{sx, 0.0f, nemchain::kBurst};
    int fact = -1;                                       // this lane's last family of the sum with a weight >= EPSILON
    // Four windows are in flight ahead of the one being summed (a window's arithmetic is shorter than a trip to
    // memory): four register sets, each refilled right after use.  Every fetch is issued unconditionally (behind the
    // last window it repeats the last one), so that the compiler counts outstanding loads exactly and waits for the
    // oldest only.
    const int nwin = (a.n + kWin - 1) / kWin;
    int gi = 0;
    auto fetch = [&](FuzzyIn& dst) { fuzzy_fetch(ctk, row, min(gi, nwin - 1) * kWin, lane, a.nw64, dst); gi++; };
    auto take = [&](const FuzzyIn& cur, const int i0) {
        uint32_t b = fuzzy_bits(cur.word, lane);
        if (role == 1) b = ~b;
        float x[kEPL];
        fuzzy_unpack(cur, x);
        const int wn = min(kWin, a.n - i0);
        bool halves = true;
        if (role < 2) {
            // the two order-free facts ComputeMedian's tie rule needs (nem_mod.c:1470-1483), beside the sums: whether
            // some ONE has a weight >= EPSILON (with the ones' sum), the last ZERO that has (with the zeros' sum)
#pragma unroll
            for (int j = 0; j < kEPL; j++) {
                const bool in = ((b >> j) & 1u) != 0 && kEPL * lane + j < wn;
                if (in && !((double)x[j] < kEpsilonD)) fact = i0 + kEPL * lane + j;
            }
        }
#pragma unroll
        for (int j = 0; j < kEPL; j++) {
            if (role == 3) {                             // nem_mod.c:1683 with |x - 1/2| = 1/2: (float)((double)s + (double)c * 0.5)
                const float h = x[j] * 0.5f;             //  = the float sum s + c/2 whenever c/2 is a float
                halves = halves && (h * 2.0f == x[j]);
                x[j] = h;
            } else if (!((b >> j) & 1u) || kEPL * lane + j >= wn) x[j] = -0.0f;   // not in this sum: the additive identity
        }
        if (role == 3 && __ballot(!halves) != 0ull) {    // an odd subnormal membership: this window in the double form
            float raw[kEPL];
            fuzzy_unpack(cur, raw);
            wave_lds_sync();
            fuzzy_stage(sx, lane, raw);
            wave_lds_sync();
            for (int j = 0; j < wn; j++) wc.acc = (float)((double)wc.acc + (double)sx[j] * 0.5);
            return;
        }
        wave_lds_sync();                                 // (the previous window's reads are done)
        fuzzy_stage(sx, lane, x);
        wave_lds_sync();
        wc.window(x, wn);
    };
    FuzzyIn r0, r1, r2, r3;
    fetch(r0); fetch(r1); fetch(r2); fetch(r3);
    for (int w = 0; w < nwin; w += 4) {
        take(r0, w * kWin); fetch(r0);
        if (w + 1 < nwin) { take(r1, (w + 1) * kWin); fetch(r1); }
        if (w + 2 < nwin) { take(r2, (w + 2) * kWin); fetch(r2); }
        if (w + 3 < nwin) { take(r3, (w + 3) * kWin); fetch(r3); }
    }
    if (role < 2) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) fact = max(fact, __shfl_xor(fact, o));
    }
    if (lane == 0) {
        if (role == 0) { a.in0[k * D + d] = wc.acc; a.any1[k * D + d] = fact >= 0 ? 1 : 0; }
        else if (role == 1) { a.in1[k * D + d] = wc.acc; a.lastz[k * D + d] = fact; }
        else if (role == 2) a.nbobs_k[k] = wc.acc;
        else a.inh_k[k] = wc.acc;
    }
}

// pass B: ComputeMedian's prefix scan over the zeros of organism d against N_k / 2 (nem_mod.c:1439-1497), then the
// centre and its inertia; a class without weight keeps its centres and gets its inertia against them (:1404-1408, 1669-1686)
__device__ __forceinline__ void mstep_fuzzy_median_body(const FuzzyArgs& a)
{
    if (a.stop != nullptr && *a.stop) return;
    __shared__ float sx_all[kFuzzyWaves][kWin];
    const int k = blockIdx.y, D = a.D, lane = threadIdx.x & 63, d = blockIdx.x * kFuzzyWaves + (threadIdx.x >> 6);
    if (d >= D) return;
    float* sx = sx_all[threadIdx.x >> 6];
    const int t = k * D + d;
    const uint64_t* row = a.xt + (size_t)d * a.nw64;
    const float* ctk = a.ct + (size_t)k * a.ctpad;
    const float nk = a.nbobs_k[k];
    const bool empty = !((double)nk > kEpsilonD);
    const float mu_old = a.center[t];
    const float half = nk / 2;                           // nem_mod.c:1439
    const double half_eps = (double)half + kEpsilonD;    // nem_mod.c:1464
    WaveChain wc{sx, 0.0f, nemchain::kBurst};
    int istar = a.n; float cum = 0.0f; bool crossed = false;
    const int nwin = (a.n + kWin - 1) / kWin;            // (four windows in flight: see mstep_fuzzy_sums_body)
    int gi = 0;
    auto fetch = [&](FuzzyIn& dst) { fuzzy_fetch(ctk, row, min(gi, nwin - 1) * kWin, lane, a.nw64, dst); gi++; };
    auto take = [&](const FuzzyIn& cur, const int i0) {
        const uint32_t b = fuzzy_bits(cur.word, lane);
        float x[kEPL];
        fuzzy_unpack(cur, x);
        const int wn = min(kWin, a.n - i0);
        bool exact = true;
#pragma unroll
        for (int j = 0; j < kEPL; j++) {
            const bool one = ((b >> j) & 1u) != 0;
            if (kEPL * lane + j >= wn) x[j] = -0.0f;
            else if (empty) {                            // c * |x - mu_old|, |.| in {0, 1/2, 1} for the centres the M-step makes
                const float ad = fabsf((one ? 1.0f : 0.0f) - mu_old);
                const float v = x[j] * ad;
                exact = exact && (ad == 0.0f || ad == 1.0f || (ad == 0.5f && v * 2.0f == x[j]));
                x[j] = ad == 0.0f ? -0.0f : v;
            } else if (one) x[j] = -0.0f;                // the scan runs over the zeros
        }
        if (empty && __ballot(!exact) != 0ull) {         // a centre outside {0, 1/2, 1} (hand-made .m): the double form
            float raw[kEPL];
            fuzzy_unpack(cur, raw);
            wave_lds_sync();
            fuzzy_stage(sx, lane, raw);
            wave_lds_sync();
            for (int j = 0; j < wn; j++) {
                const int i = i0 + j;
                const bool one = ((row[i >> 6] >> (i & 63)) & 1ull) != 0;
                wc.acc = (float)((double)wc.acc + (double)sx[j] * fabs((double)((one ? 1.0f : 0.0f) - mu_old)));   // :1683
            }
            return;
        }
        wave_lds_sync();
        fuzzy_stage(sx, lane, x);
        wave_lds_sync();
        const float before = wc.acc;
        wc.window(x, wn);
        if (!empty && !(wc.acc < half)) {
            // the weights are >= 0: the scan reached N_k / 2 inside this window -- walk it again for the family
            float run = before;
            for (int j = 0; j < wn; j++) {
                run = run + sx[j];
                if (!(run < half)) { istar = i0 + j; cum = run; break; }
            }
            crossed = true;
        }
    };
    FuzzyIn r0, r1, r2, r3;
    fetch(r0); fetch(r1); fetch(r2); fetch(r3);
    for (int w = 0; w < nwin && !crossed; w += 4) {
        take(r0, w * kWin); fetch(r0);
        if (w + 1 < nwin && !crossed) { take(r1, (w + 1) * kWin); fetch(r1); }
        if (w + 2 < nwin && !crossed) { take(r2, (w + 2) * kWin); fetch(r2); }
        if (w + 3 < nwin && !crossed) { take(r3, (w + 3) * kWin); fetch(r3); }
    }
    if (lane != 0) return;
    if (empty) { a.iner[t] = wc.acc; return; }
    float mu;
    if (crossed) {                                       // median position among the zeros
        const bool gt0 = (double)cum > half_eps;         // cum is the value at the crossing
        const bool next0 = a.lastz[t] > istar;           // a zero of weight >= EPSILON follows the crossing
        if (gt0 || next0) mu = 0.0f;                     // x_med = 0 (or midway to another 0)
        else if (a.any1[t]) mu = 0.5f;                   // midway to the first one with weight
        else mu = 0.0f;                                  // reference runs off the array here (UB)
    } else mu = 1.0f;                                    // x_med = 1 (or midway to another 1)
    a.center[t] = mu;
    a.iner[t] = (mu == 0.0f) ? a.in0[t] : (mu == 1.0f ? a.in1[t] : a.inh_k[k]);
}

// CVTEST_CLAS for float partitions (nem_alg.c:2077-2088): converged iff no |c - cold| >= thres
__device__ __forceinline__ void conv_fuzzy_body(size_t m, const float* __restrict__ c, const float* __restrict__ cold, float thres,
                             int* __restrict__ flags, const int* __restrict__ stop, const CtrlArgs& ca, const int nblk)
{
    if (stop != nullptr && *stop) return;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int bad = 0;
    if (t < m) {
        float dif = c[t] - cold[t];
        if (dif < 0) dif = -dif;
        bad = (dif >= thres);
    }
    // (one device-scope atomic per block at most, behind a device-scope look at the flag: a fuzzy partition moves
    //  everywhere, and 938 same-address atomics were the whole of this launch's 4.7 us at 20 000 x 500)
    const int blk_bad = __syncthreads_or(bad);
    if (blk_bad && threadIdx.x == 0 && __hip_atomic_load(&flags[FLAG_MOVED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&flags[FLAG_MOVED], 1);
    if (ca.ctrl != nullptr && last_block_ticket(ca.ticket, nblk)) ctrl_logic(ca);
}

// labels -> one-hot float rows (LabelToClassVector, nem_alg.c:649-664)
__device__ __forceinline__ void onehot_body(int n, int K, const uint8_t* __restrict__ lab, float* __restrict__ c)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * K) return;
    const int i = (int)(t / K), k = (int)(t - (size_t)i * K);
    c[t] = ((lab[i] & 0x7F) == k) ? 1.0f : 0.0f;
}

// ------------------------------------------------------------------------------------------
// C1: criteria (ComputeCrit, nem_alg.c:2702-2751).  Per-site terms in parallel, then the four
// i-ordered float accumulators (D, G, L, Z) on four lanes reading LDS-staged terms.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void crit_terms_body(int n, int K, int npad, const int* __restrict__ nei_ptr,
                                                    const int* __restrict__ nei_idx, const float* __restrict__ nei_w,
                                                    int use_nei, float beta, const float* __restrict__ c,
                                                    const double* __restrict__ pkfki,
                                                    const float* __restrict__ logpkfki, float* __restrict__ dik,
                                                    float* __restrict__ gik, double* __restrict__ lfi,
                                                    double* __restrict__ lzi, int hard)
{
    // hard (NCEM): rows are one-hot, so a site adds exactly one term to D and to G -- stored as ONE entry per site,
    // which makes the i-ordered chains of k_crit_reduce K times shorter (same adds, same order)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double fi = 0.0;
    float zi = 0.0f;
    float dsel = -0.0f, gsel = -0.0f;
    const int b = use_nei ? nei_ptr[i] : 0, e = use_nei ? nei_ptr[i + 1] : 0;
    for (int k = 0; k < K; k++) {
        const float cik = c[(size_t)i * K + k];
        float pik = 0.0f;
        for (int t = b; t < e; t++) pik = pik + (nei_w[t] * c[(size_t)nei_idx[t] * K + k]);
        // entries with cik <= MINFLOAT take no part in D/G (:2727): store -0.0f, the exact additive
        // identity of IEEE addition (x + -0 == x for every x, including +0, -0, inf and NaN)
        float dv = -0.0f, gv = -0.0f;
        if (cik > FLT_MIN) {                                              // MINFLOAT, :2727
            const float lp = logpkfki[(size_t)k * npad + i];
            dv = (float)((double)cik * ((double)lp - log((double)cik)));  // :2731
            gv = cik * pik;                                               // :2732
        }
        if (hard) { if (cik > FLT_MIN) { dsel = dv; gsel = gv; } }
        else { dik[(size_t)i * K + k] = dv; gik[(size_t)i * K + k] = gv; }
        fi = fi + pkfki[(size_t)k * npad + i];                            // :2739
        zi = (float)((double)zi + exp((double)(beta * pik)));             // :2740
    }
    if (hard) { dik[i] = dsel; gik[i] = gsel; }
    lfi[i] = log(fi);                                                     // :2744
    lzi[i] = log((double)zi);                                             // :2745
}

// ------------------------------------------------------------------------------------------
// The four accumulators are i-ordered (k inner) float chains, acc = (float)((double)acc + x).  They are evaluated
// exactly, but not one element after the other: between two powers of two the accumulator moves on a fixed grid and
// the chain is a prefix sum of integers (nem_chain.hpp).  One 1024-thread block per chain: a window of 4096 values
// is staged in LDS as doubles; each pass gives every thread 4 consecutive values, scans the increments, finds the
// first element the integer form cannot take (next binade, (near) tie, shrinking sum, non-finite) and takes that
// one -- and a burst after it, short at first, doubling while the integer form keeps making no headway (sums that
// change binade at every step, or all non-finite: then the chain is simply stepped) -- with the reference's
// arithmetic.
// Entries with cik <= MINFLOAT were stored as -0.0f by k_crit_terms, so adding every entry reproduces the
// reference's conditional adds (nem_alg.c:2727-2736) bit for bit.
// ------------------------------------------------------------------------------------------
constexpr int CH_T = 1024, CH_C = 4, CH_W = CH_T * CH_C;
static_assert(CH_T == kCritReduceThreads, "launchers outside this file use kCritReduceThreads");

struct ChainShared {
    double x[CH_W];
    uint32_t wave_sum[CH_T / 64];
    uint32_t m_before;
    float acc;
    int pstar, next;
};

// `count` reference steps from xs[p] on, sixteen values fetched at a time ahead of the dependent chain
__device__ inline float chain_steps(float acc, const double* xs, int p, int count)
{
    int i = 0;
    for (; i + 16 <= count; i += 16) {
        double v[16];
#pragma unroll
        for (int t = 0; t < 16; t++) v[t] = xs[p + i + t];
#pragma unroll
        for (int t = 0; t < 16; t++) acc = nemchain::step(acc, v[t]);
    }
    for (; i < count; i++) acc = nemchain::step(acc, xs[p + i]);
    return acc;
}

// consume the wn staged values; every thread of the block calls this (uniform control flow: all decisions are taken
// on shared values read after a barrier)
__device__ inline void chain_window(ChainShared& s, int wn, int& burst)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int wpos = 0;
    while (wpos < wn) {
        __syncthreads();                                 // s.acc (and, the first time round, s.x) is settled
        const float acc = s.acc;
        int E, neg;
        if (!nemchain::ready(acc, E, neg)) {
            if (nemchain::absorbing(acc)) {                  // (block-uniform)
                // +-inf stays what it is until a NaN or the opposite infinity arrives, a NaN for good: the first such
                // value of the window is found by all threads at once and only that step is taken (at 200 000 x 5 000
                // every criterion is -inf from the first family on: 2.2 ms of stepping per evaluation otherwise)
                if (acc != acc) { wpos = wn; continue; }
                if (tid == 0) s.pstar = INT_MAX;
                __syncthreads();
                int cand = INT_MAX;
                for (int j = wpos + tid; j < wn; j += CH_T) if (nemchain::poison(acc, s.x[j])) { cand = j; break; }
                if (cand != INT_MAX) atomicMin(&s.pstar, cand);
                __syncthreads();
                const int p = s.pstar;
                if (p == INT_MAX) { wpos = wn; continue; }
                if (tid == 0) s.acc = nemchain::step(acc, s.x[p]);
                wpos = p + 1;
                continue;                                    // (the barrier at the loop head publishes s.acc)
            }
            if (tid == 0) {
                const int cnt = min(burst, wn - wpos);
                s.acc = chain_steps(acc, s.x, wpos, cnt); s.next = wpos + cnt;
            }
            burst = nemchain::next_burst(burst, -1);
            __syncthreads();
            wpos = s.next;
            continue;
        }
        if (tid == 0) s.pstar = INT_MAX;
        const double sc = nemchain::scale(E, neg);
        const long long M0 = nemchain::mantissa(acc);
        const int j0 = wpos + tid * CH_C;
        // Increments are below 2^25 and every prefix that matters is at most 2^24, so the scan runs on 32-bit
        // unsigned integers: a prefix that wraps belongs to a thread behind the first stop, whose candidate (always
        // at or behind its own first element) loses to the true one.
        uint32_t inc[CH_C];
        int stop = CH_C;                                 // first element of my four that needs the exact step
        uint32_t lsum = 0;
#pragma unroll
        for (int c = 0; c < CH_C; c++) {
            inc[c] = 0;
            if (j0 + c < wn && c < stop) {
                long long v;
                if (nemchain::increment(s.x[j0 + c], sc, v)) { inc[c] = (uint32_t)v; lsum += inc[c]; }
                else stop = c;
            }
        }
        uint32_t incl = lsum;                            // inclusive scan over the wave, then over the waves
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) s.wave_sum[wave] = incl;
        __syncthreads();
        uint32_t M = (uint32_t)M0 + incl - lsum;
#pragma unroll
        for (int w = 0; w < CH_T / 64; w++) {            // (independent loads, one LDS latency)
            const uint32_t v = s.wave_sum[w];
            M += (w < wave) ? v : 0u;
        }
        int cand = INT_MAX;
        uint32_t candM = 0;
        constexpr uint32_t top = (uint32_t)nemchain::kTop;
#pragma unroll
        for (int c = 0; c < CH_C; c++) {
            const int j = j0 + c;
            if (j < wn && cand == INT_MAX) {
                if (c == stop || M >= top || M + inc[c] > top) { cand = j; candM = M; }
                else M += inc[c];
            }
        }
        if (cand != INT_MAX) atomicMin(&s.pstar, cand);
        __syncthreads();
        const int p = s.pstar;
        if (p == INT_MAX) {                              // the rest of the window went through in integer form
            if (j0 <= wn - 1 && wn - 1 < j0 + CH_C) s.acc = nemchain::compose((long long)M, E, neg);
            burst = nemchain::next_burst(burst, wn - wpos);
            wpos = wn;
            continue;                                    // (the barrier at the loop head / the caller's publishes s.acc)
        }
        if (cand == p) s.m_before = candM;
        __syncthreads();
        if (tid == 0) {
            const int cnt = min(burst, wn - p);
            s.acc = chain_steps(nemchain::compose((long long)s.m_before, E, neg), s.x, p, cnt); s.next = p + cnt;
        }
        burst = nemchain::next_burst(burst, p - wpos);
        __syncthreads();
        wpos = s.next;
    }
    __syncthreads();
}

// chain 0: D (sum of dik), 1: G (sum of gik), 2: L (sum of lfi), 3: Z (minus the sum of lzi); result to part[chain]
__device__ __forceinline__ void crit_reduce_body(int n, int K, const float* __restrict__ dik,
                                                      const float* __restrict__ gik, const double* __restrict__ lfi,
                                                      const double* __restrict__ lzi, float* __restrict__ part)
{
    __shared__ ChainShared s;
    const int chain = blockIdx.x;
    const long long total = chain < 2 ? (long long)n * K : (long long)n;     // row-major (i, k): i outer, k inner
    if (threadIdx.x == 0) s.acc = 0.0f;
    int burst = nemchain::kBurst;
    for (long long base = 0; base < total; base += CH_W) {
        const int wn = (int)min((long long)CH_W, total - base);
        __syncthreads();
        for (int t = threadIdx.x; t < wn; t += CH_T) {
            const long long g = base + t;
            s.x[t] = chain == 0 ? (double)dik[g] : chain == 1 ? (double)gik[g] : chain == 2 ? lfi[g] : -lzi[g];
        }
        chain_window(s, wn, burst);
    }
    __syncthreads();
    if (threadIdx.x == 0) part[chain] = s.acc;
}

__device__ __forceinline__ void crit_final_body(float beta, const float* __restrict__ part, float* __restrict__ crit6)
{
    const float D = part[0], G = part[1], L = part[2], Z = part[3];
    crit6[0] = D; crit6[1] = G;
    crit6[2] = (float)((double)D + (0.5 * (double)beta) * (double)G);   // :2750
    crit6[3] = (D + (beta * G)) + Z;                                    // :2751
    crit6[4] = L; crit6[5] = Z;
}

// test hook: the chain procedure on arbitrary doubles (one block)
__global__ __launch_bounds__(CH_T) void k_chain_debug(const double* __restrict__ x, long long n, float init, float* out)
{
    __shared__ ChainShared s;
    if (threadIdx.x == 0) s.acc = init;
    int burst = nemchain::kBurst;
    for (long long base = 0; base < n; base += CH_W) {
        const int wn = (int)min((long long)CH_W, n - base);
        __syncthreads();
        for (int t = threadIdx.x; t < wn; t += CH_T) s.x[t] = x[base + t];
        chain_window(s, wn, burst);
    }
    __syncthreads();
    if (threadIdx.x == 0) *out = s.acc;
}

void launch_chain_debug(const double* x, long long n, float init, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_chain_debug, dim3(1), dim3(CH_T), 0, s, x, n, init, out);
}

// test hook: the piecewise d-ordered chain of nem_halfsum.hpp on `n` values staged in LDS (as k_finish has them)
constexpr int kHalfsumDebugCap = 8192;
template <int W>
__global__ void __launch_bounds__(64 * W) k_halfsum_debug(const float* __restrict__ x, int n, float* __restrict__ out,
                                                          long long* __restrict__ stamps)
{
    __shared__ float s_x[kHalfsumDebugCap];
    __shared__ HsShared s_hs;
    for (int i = threadIdx.x; i < n; i += 64 * W) s_x[i] = x[i];
    __syncthreads();
    const float r = halfsum_block<W>(s_x, n, s_hs, stamps != nullptr);
    if (threadIdx.x == 0) {
        *out = r;
        if (stamps) {                                                // + the plain chain on one lane, for comparison
            float s = 0.0f;
            for (int i = 0; i < n; i++) s += s_x[i];
            s_hs.stamps[5] = wall_clock64();
            out[1] = s;
            for (int i = 0; i < 6; i++) stamps[i] = s_hs.stamps[i];
        }
    }
}
bool launch_halfsum_debug(const float* x, int n, int waves, float* out, hipStream_t s, long long* stamps)
{
    if (n < 0 || n > kHalfsumDebugCap) return false;
    if (waves == 16) hipLaunchKernelGGL(k_halfsum_debug<16>, dim3(1), dim3(1024), 0, s, x, n, out, stamps);
    else if (waves == 1) hipLaunchKernelGGL(k_halfsum_debug<1>, dim3(1), dim3(64), 0, s, x, n, out, stamps);
    else return false;
    return true;
}

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Kernels = the bodies above, twice: once with the argument block as a kernel parameter (one problem), once with an
// array of argument blocks in device memory and the problem in blockIdx.z (B problems per launch).
// ------------------------------------------------------------------------------------------
// (NEM_B_HEAD: nem_kernels.hpp)

__global__ __launch_bounds__(1024) void k_finish(FinishArgs a) { finish_body(a, gridDim.x); }
__global__ __launch_bounds__(1024) void k_finish_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(FinishArgs) finish_body(a, nblk); }
__global__ __launch_bounds__(256) void k_density(DensityArgs a) { density_body(a); }
__global__ __launch_bounds__(256) void k_density_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(DensityArgs) density_body(a); }
__global__ __launch_bounds__(256) void k_density_fused(FusedDensityArgs a) { (void)density_fused_body(a); }
__global__ __launch_bounds__(256) void k_density_fused_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(FusedDensityArgs) (void)density_fused_body(a); }
// (k_sweep / k_sweep_b: nem_sweep.hip)
// One relaxation round (blocks [0, nsweep)) and the M-step counts of the partition it verifies (the other blocks: its
// class masks were made by the round before) side by side in ONE launch -- the sharded iteration's second half, where
// the counts are taken from round 0's labels while round 1, the verifying round, runs.  The two do not depend on each
// other; as two launches the second waited for the first.  (The round's last-block ticket -- it publishes the rank's
// flag bytes -- is among the round's blocks only.)
template <int KT, int R>
__global__ __launch_bounds__(256) void k_sweep_counts(SweepArgs s, CountsArgs c, int nsweep)
{
    if ((int)blockIdx.x < nsweep) { sweep_body<KT, true, 256, false>(s, blockIdx.x, nsweep); return; }   // (the sharded path has no TIE_LIBC)
    mstep_counts_body<R>(c.K, c.D, c.nw64, c.xt, c.mask, c.stats, c.stop, CtrlArgs{}, (int)blockIdx.x - nsweep, (int)gridDim.x - nsweep);
}

__global__ void k_ctrl(CtrlArgs a) { ctrl_logic(a); }
__global__ void k_ctrl_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(CtrlArgs) ctrl_logic(a); }
__global__ void k_labels_post(LabelsPostArgs a)
{
    labels_post_body(a.n_local, a.lo, a.K, a.nw64, a.lab_new, a.lab_old, a.mask, a.flags, a.stop, a.ca, gridDim.x);
}
__global__ void k_labels_post_b(const void* arr, int stride, const int* gx)
{
    NEM_B_HEAD(LabelsPostArgs)
    labels_post_body(a.n_local, a.lo, a.K, a.nw64, a.lab_new, a.lab_old, a.mask, a.flags, a.stop, a.ca, nblk);
}
template <int R>
__global__ __launch_bounds__(256) void k_mstep_counts(CountsArgs a)
{
    mstep_counts_body<R>(a.K, a.D, a.nw64, a.xt, a.mask, a.stats, a.stop, a.prev_ctrl, blockIdx.x, gridDim.x);
}
template <int R>
__global__ __launch_bounds__(256) void k_mstep_counts_b(const void* arr, int stride, const int* gx)
{
    NEM_B_HEAD(CountsArgs)
    mstep_counts_body<R>(a.K, a.D, a.nw64, a.xt, a.mask, a.stats, a.stop, a.prev_ctrl, blockIdx.x, nblk);
}
__global__ __launch_bounds__(64) void k_mstep_fuzzy_a(FuzzyArgs a)
{
    mstep_fuzzy_a_body(a.n, a.npad, a.K, a.D, a.xw, a.xt, a.nw64, a.c, a.nbobs_k, a.in0, a.in1, a.inh_k, a.lastz, a.any1, a.stop,
                       a.ct != nullptr);
}
__global__ __launch_bounds__(64) void k_mstep_fuzzy_a_b(const void* arr, int stride, const int* gx)
{
    NEM_B_HEAD(FuzzyArgs)
    mstep_fuzzy_a_body(a.n, a.npad, a.K, a.D, a.xw, a.xt, a.nw64, a.c, a.nbobs_k, a.in0, a.in1, a.inh_k, a.lastz, a.any1, a.stop,
                       a.ct != nullptr);
}
__global__ __launch_bounds__(64) void k_mstep_fuzzy_b(FuzzyArgs a)
{
    mstep_fuzzy_b_body(a.n, a.npad, a.K, a.D, a.xw, a.xt, a.nw64, a.c, a.nbobs_k, a.in0, a.in1, a.inh_k, a.lastz, a.any1, a.center,
                       a.iner, a.stop);
}
__global__ __launch_bounds__(64) void k_mstep_fuzzy_b_b(const void* arr, int stride, const int* gx)
{
    NEM_B_HEAD(FuzzyArgs)
    mstep_fuzzy_b_body(a.n, a.npad, a.K, a.D, a.xw, a.xt, a.nw64, a.c, a.nbobs_k, a.in0, a.in1, a.inh_k, a.lastz, a.any1, a.center,
                       a.iner, a.stop);
}
__global__ __launch_bounds__(64 * kPcWaves) void k_mstep_fuzzy_pc(FuzzyArgs a) { mstep_fuzzy_pc_body(a); }
__global__ __launch_bounds__(64 * kPcWaves) void k_mstep_fuzzy_pc_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(FuzzyArgs) mstep_fuzzy_pc_body(a); }
__global__ __launch_bounds__(64) void k_mstep_fuzzy_med2(FuzzyArgs a) { mstep_fuzzy_med2_body(a); }
__global__ __launch_bounds__(64) void k_mstep_fuzzy_med2_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(FuzzyArgs) mstep_fuzzy_med2_body(a); }
__global__ void k_transpose_c(FuzzyArgs a) { transpose_c_body(a.n, a.ctpad, a.K, a.c, a.ct, a.stop); }
__global__ void k_transpose_c_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(FuzzyArgs) transpose_c_body(a.n, a.ctpad, a.K, a.c, a.ct, a.stop); }
__global__ __launch_bounds__(64 * kFuzzyWaves) void k_mstep_fuzzy_sums(FuzzyArgs a) { mstep_fuzzy_sums_body(a); }
__global__ __launch_bounds__(64 * kFuzzyWaves) void k_mstep_fuzzy_sums_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(FuzzyArgs) mstep_fuzzy_sums_body(a); }
__global__ __launch_bounds__(64 * kFuzzyWaves) void k_mstep_fuzzy_median(FuzzyArgs a) { mstep_fuzzy_median_body(a); }
__global__ __launch_bounds__(64 * kFuzzyWaves) void k_mstep_fuzzy_median_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(FuzzyArgs) mstep_fuzzy_median_body(a); }
__global__ void k_conv_fuzzy(ConvFuzzyArgs a) { conv_fuzzy_body(a.m, a.c, a.cold, a.thres, a.flags, a.stop, a.ca, gridDim.x); }
__global__ void k_conv_fuzzy_b(const void* arr, int stride, const int* gx)
{
    NEM_B_HEAD(ConvFuzzyArgs)
    conv_fuzzy_body(a.m, a.c, a.cold, a.thres, a.flags, a.stop, a.ca, nblk);
}
__global__ void k_onehot(OnehotArgs a) { onehot_body(a.n, a.K, a.lab, a.c); }
__global__ void k_onehot_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(OnehotArgs) onehot_body(a.n, a.K, a.lab, a.c); }
__global__ __launch_bounds__(256) void k_crit_terms(CritArgs a)
{
    crit_terms_body(a.n, a.K, a.npad, a.nei_ptr, a.nei_idx, a.nei_w, a.use_nei, a.beta, a.c, a.pkfki, a.logpkfki, a.dik, a.gik,
                    a.lfi, a.lzi, a.hard);
}
__global__ __launch_bounds__(256) void k_crit_terms_b(const void* arr, int stride, const int* gx)
{
    NEM_B_HEAD(CritArgs)
    crit_terms_body(a.n, a.K, a.npad, a.nei_ptr, a.nei_idx, a.nei_w, a.use_nei, a.beta, a.c, a.pkfki, a.logpkfki, a.dik, a.gik,
                    a.lfi, a.lzi, a.hard);
}
// (crit6 has room for the four partial results behind the six criteria)
__global__ __launch_bounds__(CH_T) void k_crit_reduce(CritArgs a)
{
    crit_reduce_body(a.n, a.hard ? 1 : a.K, a.dik, a.gik, a.lfi, a.lzi, a.crit6 + 6);
}
__global__ __launch_bounds__(CH_T) void k_crit_reduce_b(const void* arr, int stride, const int* gx)
{
    NEM_B_HEAD(CritArgs)
    crit_reduce_body(a.n, a.hard ? 1 : a.K, a.dik, a.gik, a.lfi, a.lzi, a.crit6 + 6);
}
__global__ void k_crit_final(CritArgs a) { crit_final_body(a.beta, a.crit6 + 6, a.crit6); }
__global__ void k_crit_final_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(CritArgs) crit_final_body(a.beta, a.crit6 + 6, a.crit6); }
__device__ __forceinline__ void fill_body(const FillArgs& a)
{
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < a.words; t += 256) a.ptr[t] = a.value;   // (one block)
}
__global__ void k_fill(FillArgs a) { fill_body(a); }
__global__ void k_fill_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(FillArgs) fill_body(a); }

__device__ __forceinline__ void copy_body(const CopyArgs& a, int bx, int nblk)
{
    for (int t = bx * 256 + (int)threadIdx.x; t < a.words; t += nblk * 256) a.dst[t] = a.src[t];    // (one block per 4096 words, at most 64)
}
__global__ __launch_bounds__(256) void k_layout_words(LayoutArgs a) { layout_words_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void k_layout_words_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(LayoutArgs) layout_words_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void k_layout_bits(LayoutArgs a) { layout_bits_body(a, blockIdx.x); }
__global__ __launch_bounds__(256) void k_layout_bits_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(LayoutArgs) layout_bits_body(a, blockIdx.x); }
__global__ void k_copy_words(CopyArgs a) { copy_body(a, blockIdx.x, gridDim.x); }
__global__ void k_copy_words_b(const void* arr, int stride, const int* gx) { NEM_B_HEAD(CopyArgs) copy_body(a, blockIdx.x, nblk); }

// ---- the recorder ----
static thread_local Recorder* g_recorder = nullptr;
void set_recorder(Recorder* r) { g_recorder = r; }
Recorder* current_recorder() { return g_recorder; }

void launch_ctrl(const CtrlArgs& a, hipStream_t s)
{
    if (record_op(OP_CTRL, 0, dim3(1), 1, a)) return;
    hipLaunchKernelGGL(k_ctrl, dim3(1), dim3(1), 0, s, a);
}

void launch_fill(int* ptr, int words, int value, hipStream_t s)
{
    FillArgs a{ptr, words, value};
    if (record_op(OP_FILL, 0, dim3(1), 256, a)) return;
    hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, s, a);
}

__global__ __launch_bounds__(256) void k_copy_segments(CopySegsArgs a)
{
    const int g = blockIdx.y;
    if (g >= a.n) return;
    const int words = a.words[g];
    const int4* __restrict__ s4 = reinterpret_cast<const int4*>(a.src[g]);
    int4* __restrict__ d4 = reinterpret_cast<int4*>(a.dst[g]);
    const int quads = words >> 2;                                   // (every buffer starts on a 256-byte boundary)
    for (int t = blockIdx.x * 256 + (int)threadIdx.x; t < quads; t += gridDim.x * 256) d4[t] = s4[t];
    for (int t = (quads << 2) + blockIdx.x * 256 + (int)threadIdx.x; t < words; t += gridDim.x * 256) a.dst[g][t] = a.src[g][t];
}
void launch_copy_segments(const CopySegsArgs& a, hipStream_t s)
{
    int most = 0;
    for (int g = 0; g < a.n; g++) most = std::max(most, a.words[g]);
    const dim3 grid((unsigned)std::max(1, std::min(32, (most / 4 + 1023) / 1024)), (unsigned)std::max(1, a.n));
    hipLaunchKernelGGL(k_copy_segments, grid, dim3(256), 0, s, a);
}

void launch_copy_words(const int* src, int* dst, int words, hipStream_t s)
{
    CopyArgs a{src, dst, words};
    const dim3 grid((unsigned)std::max(1, std::min(64, (words + 4095) / 4096)));   // (a fuzzy member's memberships: n x k words)
    if (record_op(OP_COPY, 0, grid, 256, a)) return;
    hipLaunchKernelGGL(k_copy_words, grid, dim3(256), 0, s, a);
}

// the device layouts of an uploaded matrix (recordable: nemgpu_solve_many leaves them to the group's run, one launch each
// for all members instead of three small launches per engine next to the running group)
void launch_layout(const uint32_t* xf, int n, int wf, int W, int npad, int d, int nw64, uint32_t* xw, uint64_t* xt,
                   const int* perm, uint32_t* xws, hipStream_t s)
{
    const LayoutArgs a{xf, perm, n, wf, W, npad, d, nw64, xw, xws, xt};
    const dim3 gw((npad + 255) / 256, ((W + 3) / 4) * 4), gb((nw64 * 64 + 255) / 256, W);
    if (!record_op(OP_LAYOUT_WORDS, 0, gw, 256, a)) hipLaunchKernelGGL(k_layout_words, gw, dim3(256), 0, s, a);
    if (!record_op(OP_LAYOUT_BITS, 0, gb, 256, a)) hipLaunchKernelGGL(k_layout_bits, gb, dim3(256), 0, s, a);
}

void launch_finish(const FinishArgs& a, hipStream_t s)
{
    // tables only (mode 0) and the models whose dispersion is per class (sk_, skd): one block per class
    const bool separable = a.mode == 0 || a.disper == NEMGPU_DISP_K_ || a.disper == NEMGPU_DISP_KD;
    const dim3 grid(separable ? a.K : 1);
    if (record_op(OP_FINISH, separable ? 1 : 0, grid, 1024, a)) return;
    hipLaunchKernelGGL(k_finish, grid, dim3(1024), 0, s, a);
}

void launch_density(const FinishArgs& t, const uint32_t* xw, int n, int npad, double* pkfki, float* logpkfki,
                    int* zero_flags, int n_zero_flags, hipStream_t s)
{
    DensityArgs a;
    a.xw = (const uint4*)xw; a.n = n; a.npad = npad; a.dpad = t.dpad; a.D = t.D; a.K = t.K;
    a.tabT = t.tabT; a.tabL0 = t.tabL0; a.nz0 = t.nz0; a.nz1 = t.nz1; a.am0 = t.am0; a.am1 = t.am1;
    a.uni = t.uni; a.nonuni = t.nonuni; a.pk = t.pk; a.logpk = t.logpk;
    a.pkfki = pkfki; a.logpkfki = logpkfki; a.zero_flags = zero_flags; a.n_zero_flags = n_zero_flags;
    a.stop = t.stop; a.use_ff = t.use_ff; a.perm = t.perm; a.ffq = t.ffq;
    const dim3 grid(((npad / 256 + 7) / 8) * 8 * t.K);
    if (record_op(OP_DENSITY, 0, grid, 256, a)) return;
    hipLaunchKernelGGL(k_density, grid, dim3(256), 0, s, a);
}

void launch_density_fused(const FinishArgs& t, const uint32_t* xw, int n, int npad, double* pkfki, float* logpkfki,
                          int* zero_flags, int n_zero_flags, hipStream_t s)
{
    FusedDensityArgs a;
    a.xw = (const uint4*)xw; a.n = n; a.npad = npad; a.dpad = t.dpad; a.D = t.D; a.K = t.K; a.n_total = t.n_total;
    a.disper = t.disper; a.propor = t.propor; a.stats = t.stats;
    a.stats_ranks = t.stats_ranks; a.stats_rank_stride = t.stats_rank_stride;
    a.center = t.center; a.disp = t.disp; a.prop = t.prop; a.nbobs_k = t.nbobs_k; a.iter_flags = t.flags;
    a.pkfki = pkfki; a.logpkfki = logpkfki; a.zero_flags = zero_flags; a.n_zero_flags = n_zero_flags; a.stop = t.stop;
    a.use_ff = t.use_ff; a.perm = t.perm;
    const dim3 grid(((npad / 256 + 7) / 8) * 8 * t.K);
    if (record_op(OP_DENSITY_FUSED, 0, grid, 256, a)) return;
    hipLaunchKernelGGL(k_density_fused, grid, dim3(256), 0, s, a);
}

// an NCEM relaxation round and M-step counts in one launch (k_sweep_counts); false: not for this shape (the caller
// launches the two separately)
bool launch_sweep_counts(const SweepArgs& sw, int K, int D, int nw64, const uint64_t* xt, const uint64_t* mask, int* stats,
                         const int* stop, hipStream_t s)
{
    if (current_recorder() != nullptr || sw.n_local >= 65536 || K < 2 || K > 5 || sw.tie_rule == NEMGPU_TIE_LIBC) return false;
    CountsArgs c{K, D, nw64, xt, mask, stats, stop, CtrlArgs{}};
    const int nsweep = (sw.n_local + 255) / 256;
    const bool wide = D + 1 >= 1024;
    const dim3 grid(nsweep + (wide ? (D + 1 + 3) / 4 : D + 1));
#define NEM_SC(KT_) case KT_:                                                                                        \
        if (wide) hipLaunchKernelGGL((k_sweep_counts<KT_, 4>), grid, dim3(256), 0, s, sw, c, nsweep);                \
        else hipLaunchKernelGGL((k_sweep_counts<KT_, 1>), grid, dim3(256), 0, s, sw, c, nsweep);                     \
        break;
    switch (K) { NEM_SC(2) NEM_SC(3) NEM_SC(4) NEM_SC(5) default: return false; }
#undef NEM_SC
    return true;
}

void launch_labels_post(int n_local, int lo, int K, int nw64, const uint8_t* lab_new, const uint8_t* lab_old,
                        uint64_t* mask, int* flags, const int* stop, const CtrlArgs* ctrl, hipStream_t s)
{
    LabelsPostArgs a{n_local, lo, K, nw64, lab_new, lab_old, mask, flags, stop, CtrlArgs{}};
    if (ctrl != nullptr) a.ca = *ctrl;
    const dim3 grid(std::min((nw64 * 64 + 255) / 256, 128));
    if (record_op(OP_LABELS_POST, 0, grid, 256, a)) return;
    hipLaunchKernelGGL(k_labels_post, grid, dim3(256), 0, s, a);
}

void launch_mstep_counts(int K, int D, int nw64, const uint64_t* xt, const uint64_t* mask, int* stats,
                         const int* stop, const CtrlArgs* prev_ctrl, hipStream_t s)
{
    // wide matrices: 4 organism rows per block (fewer re-reads of the class masks); narrow ones keep one row per
    // block so that the launch still spreads over the CUs
    CountsArgs a{K, D, nw64, xt, mask, stats, stop, CtrlArgs{}};
    if (prev_ctrl != nullptr) a.prev_ctrl = *prev_ctrl;
    const int extra = a.prev_ctrl.ctrl != nullptr ? 1 : 0;
    const bool wide = D + 1 >= 1024;
    const dim3 grid(wide ? (D + 1 + 3) / 4 + extra : D + 1 + extra);
    if (record_op(OP_COUNTS, wide ? 4 : 1, grid, 256, a)) return;
    if (wide) hipLaunchKernelGGL(k_mstep_counts<4>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_mstep_counts<1>, grid, dim3(256), 0, s, a);
}

void launch_mstep_fuzzy(int n, int npad, int K, int D, const uint32_t* xw, const uint64_t* xt, int nw64, const float* c,
                        float* ct, float* nbobs_k, float* in0, float* in1, float* inh_k, int* lastz, int* any1, float* center,
                        float* iner, const int* stop, hipStream_t s, float* chk, int* fault, int inject)
{
    const int DB = (D + 63) / 64;
    const int ctpad = (n + kWin - 1) / kWin * kWin;
    FuzzyArgs a{n, npad, K, D, xw, xt, nw64, c, nbobs_k, in0, in1, inh_k, lastz, any1, center, iner, stop, ct, ctpad, chk, fault, inject};
    if (ct != nullptr && chk != nullptr) {
        // one lane per chain at the chain's own speed: producer waves make the addends (see mstep_fuzzy_pc_body)
        if (!record_op(OP_FUZZY_T, 0, dim3(ctpad / 256), 256, a))
            hipLaunchKernelGGL(k_transpose_c, dim3(ctpad / 256), dim3(256), 0, s, a);
        if (!record_op(OP_FUZZY_PC, 0, dim3(2 * DB + 2, K), 64 * kPcWaves, a))
            hipLaunchKernelGGL(k_mstep_fuzzy_pc, dim3(2 * DB + 2, K), dim3(64 * kPcWaves), 0, s, a);
        if (!record_op(OP_FUZZY_MED2, 0, dim3(DB, K), 64, a))
            hipLaunchKernelGGL(k_mstep_fuzzy_med2, dim3(DB, K), dim3(64), 0, s, a);
        return;
    }
    if (ct == nullptr) {                                 // one lane per chain (kept for comparison: NEM_MI355X_FUZZY_CHAINS=0)
        if (!record_op(OP_FUZZY_A, 0, dim3(4 * DB + 2, K), 64, a))
            hipLaunchKernelGGL(k_mstep_fuzzy_a, dim3(4 * DB + 2, K), dim3(64), 0, s, a);
        if (!record_op(OP_FUZZY_B, 0, dim3(DB, K), 64, a))
            hipLaunchKernelGGL(k_mstep_fuzzy_b, dim3(DB, K), dim3(64), 0, s, a);
        return;
    }
    // one wave per chain, 256 families per step: class-major copy of the memberships, the sums (and with them the two
    // order-free facts of ComputeMedian's tie rule), the medians
    if (!record_op(OP_FUZZY_T, 0, dim3(ctpad / 256), 256, a))
        hipLaunchKernelGGL(k_transpose_c, dim3(ctpad / 256), dim3(256), 0, s, a);
    const dim3 gs((2 * D + 2 + kFuzzyWaves - 1) / kFuzzyWaves, K), gm((D + kFuzzyWaves - 1) / kFuzzyWaves, K);
    if (!record_op(OP_FUZZY_SUMS, 0, gs, 64 * kFuzzyWaves, a))
        hipLaunchKernelGGL(k_mstep_fuzzy_sums, gs, dim3(64 * kFuzzyWaves), 0, s, a);
    if (!record_op(OP_FUZZY_MED, 0, gm, 64 * kFuzzyWaves, a))
        hipLaunchKernelGGL(k_mstep_fuzzy_median, gm, dim3(64 * kFuzzyWaves), 0, s, a);
}

void launch_conv_fuzzy(size_t m, const float* c, const float* cold, float thres, int* flags, const int* stop,
                       const CtrlArgs* ctrl, hipStream_t s)
{
    ConvFuzzyArgs a{m, c, cold, thres, flags, stop, CtrlArgs{}};
    if (ctrl != nullptr) a.ca = *ctrl;
    const dim3 grid((unsigned)((m + 255) / 256));
    if (record_op(OP_CONV_FUZZY, 0, grid, 256, a)) return;
    hipLaunchKernelGGL(k_conv_fuzzy, grid, dim3(256), 0, s, a);
}

void launch_onehot(int n, int K, const uint8_t* lab, float* c, hipStream_t s)
{
    const size_t m = (size_t)n * K;
    OnehotArgs a{n, K, lab, c};
    const dim3 grid((unsigned)((m + 255) / 256));
    if (record_op(OP_ONEHOT, 0, grid, 256, a)) return;
    hipLaunchKernelGGL(k_onehot, grid, dim3(256), 0, s, a);
}

void launch_criteria(int n, int K, int npad, const int* nei_ptr, const int* nei_idx, const float* nei_w, int use_nei,
                     float beta, const float* c, const double* pkfki, const float* logpkfki, float* dik, float* gik,
                     double* lfi, double* lzi, float* crit6, int hard, hipStream_t s)
{
    CritArgs a{n, K, npad, nei_ptr, nei_idx, nei_w, use_nei, beta, c, pkfki, logpkfki, dik, gik, lfi, lzi, crit6, hard};
    if (!record_op(OP_CRIT_TERMS, 0, dim3((n + 255) / 256), 256, a))
        hipLaunchKernelGGL(k_crit_terms, dim3((n + 255) / 256), dim3(256), 0, s, a);
    if (!record_op(OP_CRIT_REDUCE, 0, dim3(4), CH_T, a))
        hipLaunchKernelGGL(k_crit_reduce, dim3(4), dim3(CH_T), 0, s, a);
    if (!record_op(OP_CRIT_FINAL, 0, dim3(1), 1, a))
        hipLaunchKernelGGL(k_crit_final, dim3(1), dim3(1), 0, s, a);
}

void launch_zipped(int kind, int variant, int B, const void* arr, int stride, const int* gx, unsigned max_gx, unsigned gy,
                   unsigned block, hipStream_t s)
{
    const dim3 grid(max_gx, gy, (unsigned)B), blk(block);
    switch (kind) {
    case OP_FINISH: hipLaunchKernelGGL(k_finish_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_DENSITY: hipLaunchKernelGGL(k_density_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_DENSITY_FUSED: hipLaunchKernelGGL(k_density_fused_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_SWEEP: sweep_dispatch_batched(variant, grid, block, s, arr, stride, gx); break;
    case OP_COUNTS:
        if (variant == 4) hipLaunchKernelGGL(k_mstep_counts_b<4>, grid, blk, 0, s, arr, stride, gx);
        else hipLaunchKernelGGL(k_mstep_counts_b<1>, grid, blk, 0, s, arr, stride, gx);
        break;
    case OP_LABELS_POST: hipLaunchKernelGGL(k_labels_post_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_CTRL: hipLaunchKernelGGL(k_ctrl_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_FUZZY_A: hipLaunchKernelGGL(k_mstep_fuzzy_a_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_FUZZY_B: hipLaunchKernelGGL(k_mstep_fuzzy_b_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_FUZZY_T: hipLaunchKernelGGL(k_transpose_c_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_FUZZY_SUMS: hipLaunchKernelGGL(k_mstep_fuzzy_sums_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_FUZZY_MED: hipLaunchKernelGGL(k_mstep_fuzzy_median_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_FUZZY_PC: hipLaunchKernelGGL(k_mstep_fuzzy_pc_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_FUZZY_MED2: hipLaunchKernelGGL(k_mstep_fuzzy_med2_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_CONV_FUZZY: hipLaunchKernelGGL(k_conv_fuzzy_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_ONEHOT: hipLaunchKernelGGL(k_onehot_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_CRIT_TERMS: hipLaunchKernelGGL(k_crit_terms_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_CRIT_REDUCE: hipLaunchKernelGGL(k_crit_reduce_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_CRIT_FINAL: hipLaunchKernelGGL(k_crit_final_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_FILL: hipLaunchKernelGGL(k_fill_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_COPY: hipLaunchKernelGGL(k_copy_words_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_LAYOUT_WORDS: hipLaunchKernelGGL(k_layout_words_b, grid, blk, 0, s, arr, stride, gx); break;
    case OP_LAYOUT_BITS: hipLaunchKernelGGL(k_layout_bits_b, grid, blk, 0, s, arr, stride, gx); break;
    default: break;
    }
}

// FETCH_SIZE calibration (MI355X_MICROARCH.md, HBM section): a read of a KNOWN byte count with E1's access
// pattern -- 16 bytes per lane, lanes consecutive, row after row -- so that the counter can be scaled.
__global__ __launch_bounds__(256) void k_calib_read16(const uint4* __restrict__ buf, size_t quads,
                                                      uint32_t* __restrict__ sink)
{
    uint32_t acc = 0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < quads; t += (size_t)gridDim.x * 256) {
        const uint4 v = buf[t];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9E3779B9u) sink[0] = acc;               // keeps the loads alive; practically never taken
}

void launch_calib_read(const uint32_t* buf, size_t words, uint32_t* sink, hipStream_t s)
{
    hipLaunchKernelGGL(k_calib_read16, dim3(256 * 16), dim3(256), 0, s, (const uint4*)buf, words / 4, sink);
}

#ifdef NEM_PHASE_PROF
extern "C" int nemgpu_debug_phases(unsigned long long* out32)
{
    return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 32);
}
#endif

}  // namespace nemk
