// nem_sweep_dev.hpp -- device code shared by the translation units that hold sweep kernels (nem_kernels.hip: the
// sweep + counts launch of the sharded path, the loop-control and bookkeeping kernels; nem_sweep.hip: every k_sweep
// instance): the tie-break hash, the device-side loop control, the last-block ticket, one site's row
// (ComputeLocalProba, nem_alg.c:2546-2616) and the relaxation round itself (ComputePartitionNEM's sweep,
// nem_alg.c:2330-2405, with ComputeMAP, nem_alg.c:590-645).  Compiled without relocatable device code: every
// translation unit instantiates what it launches.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "nem_kernels.hpp"

namespace nemk {

__host__ __device__ inline uint32_t mix32(uint32_t seed, uint32_t sweep, uint32_t site)
{
    // counter-based stand-in for the reference's time-seeded random() (nem_rnd.c:40-63);
    // identical to orc_mix32() in oracle/nem_oracle.c
    uint32_t h = seed * 0x9E3779B1u + sweep * 0x85EBCA77u + site * 0xC2B2AE3Du + 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

// ------------------------------------------------------------------------------------------
// Device-side loop control (one thread, once per EM iteration).  The host enqueues several
// iterations ahead; every loop kernel returns at once when ctrl[C_STOP] is set, so the host only
// has to look at `ctrl` once per batch instead of once per iteration (NemAlgo's loop test,
// nem_alg.c:1789-1840: convergence, empty class; plus "the sweep needs more relaxation rounds").
// Runs in the last block to finish of the iteration's bookkeeping kernel (k_labels_post /
// k_conv_fuzzy), or as its own tiny launch when there is no such kernel.
// ------------------------------------------------------------------------------------------
// (measured, round 3: out of line -- __noinline__, the block passed by value -- the kernels that call it need a private
//  segment, 176 bytes per lane, and every launch of theirs got 3-4 us longer; it stays inline)
__device__ inline void ctrl_logic(const CtrlArgs& a)
{
    int* c = a.ctrl;
    // One thread runs this at the tail of a launch (or as a launch of its own): a string of dependent read-modify-writes
    // of device memory would BE that tail (each ~0.1-0.4 us).  So everything the decision reads is requested up front
    // -- independent loads, one memory latency -- the tests run on registers, and what changed is stored at the end.
    constexpr int kMaxRounds = 8;                                 // (kRoundsMax in nem_engine.hip: the fuzzy loop's learned counts go up to 7)
    const int nr = a.n_rounds > 0 ? (a.n_rounds < kMaxRounds ? a.n_rounds : kMaxRounds) : 2;
    const int stop = c[C_STOP];
    int iters = c[C_ITERS], commits = c[C_COMMITS], sweep_rounds = c[C_SWEEP_ROUNDS], nzero = c[C_NZERO], firstzero = c[C_FIRSTZERO];
    int draws = c[C_DRAWS];
    const int sweep_next = *a.sweep_next;
    const int emptyk = a.iter_flags[FLAG_EMPTYK];
    int moved = __hip_atomic_load(&a.iter_flags[FLAG_MOVED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    static_assert(FLAG_ROUND_STRIDE == 4 && FLAG_CHANGED == 0 && FLAG_NZERO == 1 && FLAG_FIRSTZERO == 2 && FLAG_NTIES == 3, "a round's slot is one int4");
    int4 rf[kMaxRounds];
#pragma unroll
    for (int r = 0; r < kMaxRounds; r++) rf[r] = r < nr ? reinterpret_cast<const int4*>(a.round0)[r] : make_int4(1, 0, 0, 0);
    int4 bf = make_int4(0, 0, 0, 0);
    bool blind_ok = true;
    int blind_needed = 1;
    if (a.is_init && a.blind != nullptr) {
        if (a.blind_rounds > 1) {
            // TIE_LIBC: the blind sweep's sites are coupled through the draw counter, it ran verified rounds like any
            // sweep: its result is the first round that changed nothing (the rounds behind it did not run)
            blind_ok = false;
            for (int q = 0; q < a.blind_rounds && !blind_ok; q++) {
                const int4 f = reinterpret_cast<const int4*>(a.blind)[q];
                if (f.x == 0) { bf = f; blind_ok = true; blind_needed = q + 1; }
            }
        } else bf = *reinterpret_cast<const int4*>(a.blind);
    }
    int draw0 = 0, rank_draws = 0;
    if (a.draw_ctl != nullptr) {
        draw0 = a.draw_ctl[0];
        if (a.q_tot != nullptr)                                   // sharded: every rank's draws of the sweep's final round
            for (int r = 0; r < a.n_ranks; r++) rank_draws += *reinterpret_cast<const int*>(a.q_tot + (size_t)r * a.flag_stride);
    }
    int ch0 = 0, ch1 = 0, mv = 0;
    if (a.q_flags != nullptr) {                                   // sharded: every rank's flag bytes (all-gathered)
        for (int r = 0; r < a.n_ranks; r++) {
            const size_t o = (size_t)r * a.flag_stride;
            ch0 |= a.q_flags[o];
            ch1 |= a.r_flags[o];
            // every rank's 'one of MY labels moved' byte, next to its flag byte of the sweep's last round (1: round 1's,
            // 2: round 0's): every rank takes the same decision without another collective, and without a pass over
            // the other ranks' labels
            if (a.moved_bytes) mv |= (a.moved_bytes == 2 ? a.q_flags : a.r_flags)[o + 1];
        }
        if (a.moved_bytes) moved = mv;
    }
    if (stop) return;

    // which of the enqueued relaxation rounds changed nothing (= the sweep's fixed point)?  -1: none of them
    int last = -1;
    if (!a.use_nei) last = 0;                                     // one round, nothing to verify
    else if (a.q_flags != nullptr) last = !ch0 ? 0 : (!ch1 ? 1 : -1);   // sharded: two rounds
    else {
#pragma unroll
        for (int r = kMaxRounds - 1; r >= 0; r--) if (r < nr && rf[r].x == 0) last = r;
    }
    int4 f = rf[0];
#pragma unroll
    for (int r = 1; r < kMaxRounds; r++) if (last == r) f = rf[r];
    // the tie-break hash is keyed by the sweep number; sweeps 0 and 1 are the two initial ones
    if (a.is_init) *a.sweep_next = 2;
    else { iters += 1; c[C_ITERS] = iters; *a.sweep_next = sweep_next + 1; }
    if (a.is_init) {                                              // ComputePartitionFromPara(Needinit=1): no iteration counted
        if (!blind_ok) { c[C_NEED_ROUNDS] = 3; c[C_STOP] = 1; return; }   // (what ran behind the blind sweep is void: the host starts over)
        if (bf.y > 0) {                                           // the blind sweep's zero-density sites come first
            nzero += bf.y;
            if (firstzero == 0) firstzero = bf.z;
        }
        if (last >= 0) {
            sweep_rounds += last + 2;                             // blind sweep + this one
            if (f.y > 0) { nzero += f.y; if (firstzero == 0) firstzero = f.z; }
            c[C_FOLD] = f.y > 0;                                  // how the next sweeps tally such sites, see k_sweep
            c[C_SWEEP_ROUNDS] = sweep_rounds;
            c[C_INIT_ROUNDS] = (blind_needed << 8) | (last + 1);     // (what the two sweeps needed: the host enqueues accordingly)
            if (a.draw_ctl != nullptr && a.blind_rounds > 1) {    // TIE_LIBC: both sweeps' draws move the stream on
                const int nt = (bf.w & ((1 << 30) - 1)) + (f.w & ((1 << 30) - 1));
                a.draw_ctl[0] = draw0 + nt;
                c[C_DRAWS] = draws + nt;
                c[C_DRAWS_INIT] = nt;
            }
        }
        c[C_NZERO] = nzero; c[C_FIRSTZERO] = firstzero;
        if (last < 0) { c[C_NEED_ROUNDS] = 2; c[C_STOP] = 1; }
        return;
    }
    if (!a.param_fix && emptyk != 0) {                            // nem_alg.c:1831-1838: E-step "not run"
        c[C_STATUS] = NEMGPU_W_EMPTYCLASS;
        c[C_EMPTYK] = emptyk;
        c[C_STOP] = 1;
        return;
    }
    if (last < 0) { c[C_NEED_ROUNDS] = 1; c[C_STOP] = 1; return; }
    c[C_SWEEP_ROUNDS] = sweep_rounds + last + 1;
    if (last >= 2) c[C_DEEP] = iters;
    if (f.y > 0) {
        c[C_NZERO] = nzero + f.y;
        if (firstzero == 0) c[C_FIRSTZERO] = f.z;
    }
    c[C_FOLD] = f.y > 0;
    if (a.draw_ctl != nullptr) {                                  // TIE_LIBC: the sweep's draws move the stream on
        const int nt = a.q_tot != nullptr ? rank_draws : (f.w & ((1 << 30) - 1));
        a.draw_ctl[0] = draw0 + nt;
        c[C_DRAWS] = draws + nt;
    }
    c[C_COMMITS] = commits + 1;
    if (a.cvtest == NEMGPU_CV_CLAS) {                             // HasConverged, nem_alg.c:2075-2089
        const int conv = a.ncem ? (moved ? (1.0f < a.cvthres) : (0.0f < a.cvthres)) : !moved;
        if (conv) { c[C_CONVERGED] = 1; c[C_STOP] = 1; }
    }
}

// last-block-done ticket: returns true in exactly one thread of the grid, after every block's global
// writes (made before its call) are visible to it
// Two levels (groups of blocks, then the groups) once the grid is larger than 32 blocks: same-address atomics
// are served one after the other (~45 ns each on MI355X), so a flat counter costs 9 us at 200 blocks; with at most
// 32 groups on counters 128 bytes apart it is ~sqrt of that.  `ticket` points at kTicketWords zeroed ints.
// The counters are 64-bit: arrivals in the low word, and in the high word an optional grid-wide sum (`tally`, this
// block's share; the last block stores the total to *tally_out) that rides on the same atomics -- a sum that would
// otherwise cost one same-address atomic per block.
__device__ inline bool last_block_ticket(int* ticket, int nblocks, int tally = 0, int* tally_out = nullptr)
{
    __shared__ int s_last;
    // What the last block reads of the others are FLAGS, all of them updated by device-scope atomics: every wave waits
    // until its own are performed (vmcnt(0)), the block meets, one thread adds the arrival.  (A device-scope release
    // fence here -- `buffer_wbl2`, the whole L2 written back -- cost microseconds per launch once every block did one;
    // the last block's acquire fence below stays: it reads the flags with plain loads.)
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0)
    __syncthreads();
    if (threadIdx.x == 0 && tally_out == nullptr) {                // plain ticket: 32-bit counters
        int last = 0;
        if (nblocks <= 32) {
            const int t = atomicAdd(ticket, 1);
            last = (t == nblocks - 1);
            if (last) *ticket = 0;
        } else {
            const int gsz = (nblocks + 31) / 32;
            const int g = blockIdx.x / gsz;
            const int ng = (nblocks + gsz - 1) / gsz;
            const int members = min(gsz, nblocks - g * gsz);
            int* gc = ticket + 32 * (1 + g);
            if (atomicAdd(gc, 1) == members - 1) {                 // (the top arrival is issued once this one has returned)
                *gc = 0;
                if (atomicAdd(ticket, 1) == ng - 1) { *ticket = 0; last = 1; }
            }
        }
        if (last) __threadfence();
        s_last = last;
    } else if (threadIdx.x == 0) {
        using u64 = unsigned long long;
        const u64 mine = 1ull | ((u64)(unsigned)tally << 32);
        int last = 0;
        u64 total = 0;
        if (nblocks <= 32) {
            u64* tc = reinterpret_cast<u64*>(ticket);
            const u64 t = atomicAdd(tc, mine);
            last = ((int)(t & 0xffffffffull) == nblocks - 1);
            if (last) { *tc = 0; total = (t >> 32) + (u64)(unsigned)tally; }
        } else {
            const int gsz = (nblocks + 31) / 32;                   // blocks per group; at most 32 groups
            const int g = blockIdx.x / gsz;
            const int ng = (nblocks + gsz - 1) / gsz;
            const int members = min(gsz, nblocks - g * gsz);
            u64* gc = reinterpret_cast<u64*>(ticket + 32 * (1 + g));
            const u64 t = atomicAdd(gc, mine);
            if ((int)(t & 0xffffffffull) == members - 1) {
                *gc = 0;
                const u64 gsum = (t >> 32) + (u64)(unsigned)tally;
                u64* tc = reinterpret_cast<u64*>(ticket);
                const u64 tt = atomicAdd(tc, 1ull | (gsum << 32));
                if ((int)(tt & 0xffffffffull) == ng - 1) { *tc = 0; last = 1; total = (tt >> 32) + gsum; }
            }
        }
        if (last) {
            if (tally_out != nullptr) *tally_out = (int)total;
            __threadfence();
        }
        s_last = last;
    }
    __syncthreads();
    return s_last && threadIdx.x == 0;
}

// ------------------------------------------------------------------------------------------
// E2: one relaxation round of the Gauss-Seidel site sweep.
//
// The reference updates sites in index order, in place (UPDATE_SEQ): site i sees the NEW rows
// of neighbours j < i and the OLD rows of neighbours j >= i.  That is a lower-triangular
// system  c = F(c_{<i}, old_{>=i})  with a unique solution.  Each round evaluates every site
// in parallel against a guess of the new rows; when a round changes nothing the guess IS that
// solution, bit for bit.  Round r reads guess_r and writes out_r; rounds after the first
// unchanged one exit immediately (prev_changed == 0).
// ------------------------------------------------------------------------------------------
// One site's row: ComputeLocalProba (nem_alg.c:2576-2613) from the class contexts.  cf = the normalised row;
// returns true when the site hit the "density = 0" branch.
template <int KA>
__device__ __forceinline__ bool local_proba(const SweepArgs& a, int K, const double* pkf, const float* ctx, float* cf,
                                            const double* exp_tab = nullptr, int exp_tab_len = 0)
{
    double cinum[KA];
    double cum = 0.0;
    // (exp_tab[m] = exp((double)beta * (double)(float)m) from the same device exp: a context that is such an integer
    //  takes its factor from the table, every other one runs exp -- decided for the whole row, so that a row costs
    //  either K table reads or K calls)
    bool tab = exp_tab != nullptr;
    int mi[KA];
    if (tab) {
#pragma unroll
        for (int k = 0; k < KA; k++) {
            if (k < K) {
                const float c = ctx[k];
                mi[k] = (int)c;
                tab = tab && c >= 0.0f && c < (float)exp_tab_len && (float)mi[k] == c;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KA; k++) {
        if (k < K) {                                     // nem_alg.c:2581-2584
            double v = pkf[k];
            if (a.use_nei) v = v * (tab ? exp_tab[mi[k]] : exp((double)a.beta * (double)ctx[k]));
            cinum[k] = v;
            cum = cum + v;
        }
    }
    if (cum > 0) {                                       // nem_alg.c:2589-2601
        if (cum > kEpsilonD) {
            const double invz = 1 / cum;
#pragma unroll
            for (int k = 0; k < KA; k++) if (k < K) cf[k] = (float)(invz * cinum[k]);
        } else {
            const double invz = 1 / (cum / kEpsilonD);
#pragma unroll
            for (int k = 0; k < KA; k++) if (k < K) cf[k] = (float)(invz * (cinum[k] / kEpsilonD));
        }
        return false;
    }
    const float u = (float)(1.0 / K);                    // nem_alg.c:2603-2607
#pragma unroll
    for (int k = 0; k < KA; k++) if (k < K) cf[k] = u;
    return true;                                         // counted by the caller, once per block
}

// Labels are bytes: class in the low 7 bits; bit 7 = "this site's C-step drew a random number" (TIE_LIBC only), so
// that a guess of the new partition carries the guess of who draws with it.
constexpr int kLabMask = 0x7F, kLabDrew = 0x80;
constexpr int kTabShort = 1 << 30;                       // in a round's FLAG_NTIES word: the draw table was too short
constexpr int kFuzzyWaves = 4;                          // chains (waves) per block of the fuzzy M-step's chain kernels
constexpr int kInnerCap = 64;                            // block-local iterations per round (any cap is exact)

// sum over the 64 lanes, in every lane (data-parallel-primitive adds, no LDS)
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);   // row_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast:15
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);   // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}

// A label byte of another block as that block published it between two rounds of ONE launch (k_sweep_fused): the
// aligned word it sits in, by an agent-scope load that no cache of this CU answers (MI355X_MICROARCH.md, inter-workgroup
// visibility: the producers' stores are agent-scope word stores, drained before their block's arrival on the barrier)
__device__ __forceinline__ int label_coherent(const uint8_t* buf, int j)
{
    const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(buf + (j & ~3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int)((w >> (8 * (j & 3))) & 0xFFu);
}

// The meeting of a fused launch's blocks between two relaxation rounds.  No counter: 79 arrivals on one word are 79
// same-address atomics, served one after the other (~45 ns each: 3-4.5 us per meeting, measured with the in-kernel
// clock) -- every block has a word of its own instead, words[parity of the round][block] = (round + 1) << 1 | "this
// block changed a label", written by one agent-scope store once every wave of the block has waited for its label
// stores (vmcnt(0)) and the block has met; the block's first wave then polls ALL blocks' words, one load instruction
// per 64 blocks, until every word carries this round.  Two parities: a block can reach the next meeting while a slow
// one is still reading this one's words, never the one after (it would need the slow block's arrival).  The words are
// zero before the sweep.  Bounded (100 MHz clock): a block that waits longer than kFusedWaitTicks gives up and
// reports it (-1), so that the grid always drains.  Returns the number of blocks that changed something, the same
// value in every block.
constexpr unsigned long long kFusedWaitTicks = 400000ull;      // 4 ms
__device__ __forceinline__ int fused_meet(unsigned* words, int nblk, int rnd, bool blk_changed, int bx)
{
    __shared__ int s_meet;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        unsigned* w = words + (rnd & 1) * kFusedMaxBlocks;
        const unsigned want = (unsigned)(rnd + 1);
        if (threadIdx.x == 0) __hip_atomic_store(&w[bx], (want << 1) | (blk_changed ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        int total = 0;
        for (;;) {
            bool ok = true;
            int chg = 0;
            for (int b = threadIdx.x; b < nblk; b += 64) {
                const unsigned v = __hip_atomic_load(&w[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (v >> 1) == want;
                chg += (int)(v & 1u);
            }
            if (__all(ok)) { total = wave_sum_i32(chg); break; }
            if (__any(wall_clock64() - t0 > kFusedWaitTicks)) { total = -1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (threadIdx.x == 0) s_meet = total;
    }
    __syncthreads();
    return s_meet;
}

// LIBC: the reference's tie stream (TIE_LIBC) -- its bookkeeping (who drew, per wave and block, at every block-local
// step) is compiled into the instances that need it only
// FUSED: up to a.fused_rounds relaxation rounds in this one launch (see SweepArgs); NCEM, no tie stream, every block
// of the grid resident
template <int KT, bool NCEM, int BS, bool LIBC = false, bool FUSED = false>
__device__ __forceinline__ void sweep_body(const SweepArgs& a, const int bx, const int nblk)
{
    static_assert(!FUSED || (NCEM && !LIBC), "the fused rounds exist for NCEM without the libc tie stream");
#define NEM_SWEEP_STAMP(i_) do { if (FUSED && a.prof != nullptr && threadIdx.x == 0 && (bx == 0 || bx == nblk - 1) && (i_) < 32) \
        a.prof[(bx == 0 ? 0 : 32) + (i_)] = wall_clock64(); } while (0)
    NEM_SWEEP_STAMP(0);
    int fold_hint = 0;
    if (a.stop != nullptr) {
        static_assert(C_STOP == 0 && C_FOLD == 1, "one 8-byte load");
        const int2 sf = *reinterpret_cast<const int2*>(a.stop);
        if (sf.x) return;
        fold_hint = sf.y;
    }
    bool skip = (a.prev_changed != nullptr && *a.prev_changed == 0);   // the previous round was already the fixed point
    if (a.flags_in != nullptr) {                         // sharded: did ANY rank change a label last round?
        int any = 0;
        for (int r = 0; r < a.n_ranks; r++) any |= a.flags_in[(size_t)r * a.slot_stride];
        if (!any) skip = true;
    }
    if (skip && !(NCEM && a.post_on)) {
        if (a.publish_byte != nullptr && bx == 0 && threadIdx.x == 0) *a.publish_byte = 0;
        // TIE_LIBC: the sweep's draw count rides on through the rounds that have nothing left to do, so that whoever was
        // enqueued behind the LAST of them finds it there (SweepArgs::draw_extra of the sweep that follows)
        if (LIBC && a.prev_changed != nullptr && bx == 0 && threadIdx.x == 0) a.flags[FLAG_NTIES] = a.prev_changed[FLAG_NTIES - FLAG_CHANGED];
        return;
    }
    __shared__ int s_nzero, s_first;
    // "a label changed" / "a label moved": per block, then ONE device-scope atomic -- behind a device-scope look at the
    // flag -- instead of one per wave behind a plain load (which another XCD's L2 answers with a stale 0 long after the
    // flag was set: at 200 000 x 5 000, where every wave has something to report, a round's 3 136 same-address atomics
    // were 10-40 us of its 24-52)
    __shared__ int s_chg, s_mov;
    // NCEM: the MRF factor exp(beta * context) takes few distinct arguments -- the context of a class is a sum of edge
    // weights, small integers in PPanGGOLiN's graphs (numbers of organisms sharing an adjacency): the block fills a table
    // exp((double)beta * (double)(float)m), m = 0 .. kExpTab - 1, with the SAME device exp on the same argument the site
    // would pass (bit-identical by construction), one entry per thread, and a site whose context is such an integer
    // reads it instead of running three double-precision exp (a third of the round's vector instructions)
    // Graphs with PPanGGOLiN's own edge weights -- counts of organisms, up to D -- have contexts in the hundreds: there the
    // table comes from SweepArgs::exp_tab (made once per beta by the same exp on the same arguments, k_exp_table) and is
    // copied to LDS, as many entries as the graph's largest weight sum can index: up to 1024 in a round of its own, 4096
    // in a fused launch.  (64 entries or fewer: the block computes them itself, no dependent load at its head.)
    constexpr int kExpTab = FUSED ? kExpTabGlobal : 1024;
    __shared__ double s_exp[NCEM ? kExpTab : 1];
    __shared__ __attribute__((aligned(16))) uint8_t s_lab[NCEM ? BS : 1];   // the block's labels while it iterates
    __shared__ uint64_t s_drew[BS / 64];                 // TIE_LIBC: per wave, which of its sites drew
    const int spb = (BS > 256 && a.spb > 0) ? a.spb : BS;   // sites of this block (large shards: fewer than the launch bound)
    const int i = bx * spb + threadIdx.x;
    const bool active = (int)threadIdx.x < spb && i < a.n_local;
    const int gi = a.lo + (active ? i : 0);
    const int K = KT > 0 ? KT : a.K;
    constexpr int KA = KT > 0 ? KT : kMaxKernelK;
    bool zero_density = false;
    bool changed = false;
    // this site's own labels, requested with everything else at the head of the block: asked for where they are
    // used -- behind the label store, which they might alias -- they would be one more memory latency at the tail
    int my_guess = 0, my_old = 0, my_new = 255;           // my_guess: the whole byte; my_old: the class
    if (NCEM && active) { my_guess = a.lab_guess[gi]; my_old = a.lab_old[gi] & kLabMask; }
    // (TIE_LIBC: the block's draw count in the guess, compared at the round's end -- asked for here, not there: a load at
    //  the tail is a memory latency added to the block's run time)
    int blk_cnt_guess = 0;
    if (LIBC && threadIdx.x == 0 && a.tie_cnt_guess != nullptr) blk_cnt_guess = a.tie_cnt_guess[bx];
    __shared__ int s_anydrew;                            // TIE_LIBC: a site of the block drew at some local step of this launch
    if (threadIdx.x == 0) { s_nzero = 0; s_first = 0; s_chg = 0; s_mov = 0; s_anydrew = 0; }
    if (NCEM) s_lab[threadIdx.x] = (uint8_t)my_guess;
    int exp_len = 64;                                    // usable entries of s_exp
    if (NCEM && a.exp_tab != nullptr && (FUSED || a.exp_tab_len > 64)) {
        exp_len = a.exp_tab_len < kExpTab ? a.exp_tab_len : kExpTab;
        if (a.use_nei && !skip) for (int m = threadIdx.x; m < exp_len; m += BS) s_exp[m] = a.exp_tab[m];
    } else if (NCEM && a.use_nei && !skip && threadIdx.x < 64) s_exp[threadIdx.x] = exp((double)a.beta * (double)(float)threadIdx.x);
    int* rflags = a.flags;                               // this round's flag slot (a fused launch moves on slot by slot)
    // TIE_LIBC: draws of the blocks below this one, as the guess has them -- summed once by the block's first wave (a
    // lane that ties used to add the counts up itself: up to 78 loads in a row per tying lane, 2-3 us of a round on
    // data that tie in every sweep, the initial sweeps of a random start)
    __shared__ int s_lower;
    if (LIBC && !skip && threadIdx.x < 64 && a.tie_cnt_guess != nullptr) {
        int part = 0;
        for (int b = threadIdx.x; b < bx; b += 64) part += a.tie_cnt_guess[b];
        part = wave_sum_i32(part);
        if (threadIdx.x == 0) {
            if (a.rank_tot_in != nullptr)                // sharded: the ranks below, as the guess has them
                for (int r = 0; r < a.rank_index; r++) part += *reinterpret_cast<const int*>(a.rank_tot_in + (size_t)r * a.slot_stride);
            s_lower = part;
        }
    }
    // TIE_LIBC: does any site of the block carry "drew" in the guess?  Blocks without ties (most blocks, most rounds) skip
    // the per-step tally of who drew -- a ballot and a barrier per local step, and another pair at the round's end
    bool blk_drew = false;
    if (LIBC) blk_drew = __syncthreads_or((my_guess & kLabDrew) != 0) != 0;
    else __syncthreads();
    NEM_SWEEP_STAMP(1);

    if (NCEM) {
    // ------------------------------------------------------------------------------------------
    // NCEM.  A plain round would evaluate every site once against the guess.  Most Gauss-Seidel dependencies are
    // index-adjacent (the contiguity path), i.e. inside the block: so the block goes on evaluating its sites against
    // ITS OWN latest labels for the lower-indexed neighbours it holds (the guess for all others) until nothing in
    // the block changes -- Jacobi steps on a lower-triangular system, final for the block's first t sites after t
    // steps.  Exactness is untouched: the first step is the plain round, so the block reproduces its guess iff
    // the plain round would, and the sweep still ends with a round that changes nothing anywhere.  What it buys is
    // rounds: label changes run down the path inside one launch instead of one launch per hop.
    // ------------------------------------------------------------------------------------------
    constexpr bool libc = LIBC;
    const int blk_lo = a.lo + bx * spb;          // first label slot of this block
    double pkf[KA];
#pragma unroll
    for (int k = 0; k < KA; k++) if (k < K) pkf[k] = active && !skip ? a.pkfki[(size_t)k * a.npad + i] : 0.0;
    // the first four neighbours live in registers: index, weight, and the label when it cannot change in here
    int nb = 0, ne = 0;
    int dyn[4]; float wn[4]; int fl[4];
    int jlow[4];                                         // FUSED: label slot of a lower neighbour in another block (re-read between rounds), else -1
#pragma unroll
    for (int u = 0; u < 4; u++) { dyn[u] = -1; wn[u] = 0.0f; fl[u] = 255; jlow[u] = -1; }
    if (active && !skip && a.use_nei) {
        nb = a.nei_ptr[i]; ne = a.nei_ptr[i + 1];
        int jn[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bool ok = nb + u < ne;
            jn[u] = ok ? a.nei_idx[nb + u] : gi;
            wn[u] = ok ? a.nei_w[nb + u] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (nb + u < ne) {
                if (jn[u] < gi && jn[u] >= blk_lo) dyn[u] = jn[u] - blk_lo;
                else {
                    fl[u] = ((jn[u] < gi) ? a.lab_guess[jn[u]] : a.lab_old[jn[u]]) & kLabMask;
                    if (FUSED && jn[u] < gi) jlow[u] = jn[u];
                }
            }
        }
    }
    const bool long_row = ne - nb > 4;
    int lower_draws = -1;                                // (read from s_lower, written ahead of the barrier at the block's head)
    bool tab_short = false;                              // TIE_LIBC: a draw fell outside the table: the round is void
    int cur = my_guess;                                  // this site's byte in s_lab
    int seen[4] = {-1, -1, -1, -1};                      // labels the last evaluation used for the dyn neighbours
    uint64_t seen_drew = ~0ull; int seen_wave_draws = -1;
    // FUSED: the rounds of this launch.  Round `rnd` reads the labels of other blocks from guess_buf -- the old
    // partition in round 0 (a fused launch starts a sweep), the buffer every block published before the meeting
    // afterwards -- and publishes this block's labels to out_buf.  `dirty`: this site must be evaluated at the round's
    // first local step (round 0: every site; later: the sites one of whose inputs from outside the block changed).
    const uint8_t* guess_buf = a.lab_guess;
    uint8_t* out_buf = a.lab_out;
    const int n_rounds = FUSED ? a.fused_rounds : 1;
    bool dirty = true;
    bool fused_failed = false;
    int rnd = 0;
    for (;;) {
    for (int it = 0; it < kInnerCap; it++) {
        int below_in_block = 0;
        uint64_t drew_lt = 0;
        if (libc && blk_drew) {                          // who drew, per wave (block-uniform branch)
            const uint64_t bal = __ballot((s_lab[threadIdx.x] & kLabDrew) != 0);
            if ((threadIdx.x & 63) == 0) s_drew[threadIdx.x >> 6] = bal;
            __syncthreads();
            for (int w = 0; w < (int)(threadIdx.x >> 6); w++) below_in_block += (int)__popcll(s_drew[w]);
            drew_lt = s_drew[threadIdx.x >> 6] & ((1ull << (threadIdx.x & 63)) - 1ull);
        }
        int nxt = cur;
        if (active && !skip) {
            int lab[4];
            bool same = !(it == 0 && dirty) && !long_row;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                lab[u] = dyn[u] >= 0 ? (s_lab[dyn[u]] & kLabMask) : fl[u];
                same = same && (dyn[u] < 0 || lab[u] == seen[u]);
            }
            if (libc) same = same && drew_lt == seen_drew && below_in_block == seen_wave_draws;
            if (!same) {
                float ctx[KA];
#pragma unroll
                for (int k = 0; k < KA; k++) ctx[k] = 0.0f;
                // (the adds keep the .nei order, SumNeighsOfClass nem_alg.c:2865-2875; w*1 = w, w*0: additive identity)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    seen[u] = lab[u];
                    if (nb + u < ne) {
#pragma unroll
                        for (int k = 0; k < KA; k++)
                            if (k < K) ctx[k] = ctx[k] + ((lab[u] == k) ? wn[u] : -0.0f);
                    }
                }
                for (int t = nb + 4; t < ne; t++) {      // rows longer than four neighbours: from memory
                    const int j = a.nei_idx[t];
                    const float wt = a.nei_w[t];
                    int l;
                    if (j < gi && j >= blk_lo) l = s_lab[j - blk_lo] & kLabMask;
                    else if (FUSED && rnd > 0 && j < gi) l = label_coherent(guess_buf, j) & kLabMask;
                    else l = ((j < gi) ? a.lab_guess[j] : a.lab_old[j]) & kLabMask;
#pragma unroll
                    for (int k = 0; k < KA; k++)
                        if (k < K) ctx[k] = ctx[k] + ((l == k) ? wt : -0.0f);
                }
                float cf[KA];
                zero_density = local_proba<KA>(a, K, pkf, ctx, cf, s_exp, exp_len);
                // ComputeMAP, nem_alg.c:603-640
                int kmax = 0; float ukmax = cf[0];
#pragma unroll
                for (int k = 1; k < KA; k++) if (k < K && cf[k] > ukmax) { ukmax = cf[k]; kmax = k; }
                int drew = 0;
                if (a.tie_rule != NEMGPU_TIE_FIRST) {
                    int nequal = 0;
#pragma unroll
                    for (int k = 1; k < KA; k++) if (k < K && k > kmax && cf[k] == ukmax) nequal++;
                    if (nequal > 0) {
                        uint32_t r;
                        if (libc) {
                            // the reference's stream (nem_rnd.c:53-61): this site's draw is number
                            //   draws before the sweep + sites below it that drew in this sweep
                            if (lower_draws < 0) lower_draws = s_lower;
                            int base = a.draw_base, tab0 = a.draw_tab0;
                            if (a.draw_ctl != nullptr) { base = a.draw_ctl[0]; tab0 = a.draw_ctl[1]; }
                            if (a.draw_extra != nullptr) base += *a.draw_extra & (kTabShort - 1);
                            const int at = base + lower_draws + below_in_block + (int)__popcll(drew_lt) - tab0;
                            if (at >= 0 && at < a.draw_tab_len) r = a.draw_tab[at];
                            else { r = 0; tab_short = true; }
                            drew = kLabDrew;
                        } else {
                            const uint32_t sid = a.sweep_id_ptr != nullptr ? (uint32_t)*a.sweep_id_ptr : a.sweep_id;
                            // the hash is keyed by the TRUE family index (label slots of a sharded run carry a flag tail per rank)
                            const uint32_t site = a.slot_stride > 0 ? (uint32_t)(gi - (gi / a.slot_stride) * a.slot_pad) : (uint32_t)gi;
                            r = mix32(a.tie_seed, sid, site);
                        }
                        const int pick = (int)(r % (uint32_t)(nequal + 1));
                        int seen_eq = 0, chosen = kmax;
#pragma unroll
                        for (int k = 1; k < KA; k++)
                            if (k < K && k > kmax && cf[k] == ukmax) { seen_eq++; if (seen_eq == pick) chosen = k; }
                        kmax = chosen;
                    }
                }
                nxt = kmax | drew;
                seen_drew = drew_lt; seen_wave_draws = below_in_block;
            }
        }
        __syncthreads();                                 // every read of s_lab / s_drew of this step is done
        const bool moved_now = nxt != cur;
        cur = nxt;
        s_lab[threadIdx.x] = (uint8_t)cur;
        // (__syncthreads_or hands back a truth value, not the OR of the arguments: "somebody drew" travels through LDS,
        //  written ahead of the barrier, and sticks -- a block that has met a tie keeps the tally for the rest of the launch)
        if (libc && !blk_drew && (cur & kLabDrew) != 0) s_anydrew = 1;
        const int any_moved = __syncthreads_or(moved_now);
        if (libc && !blk_drew) blk_drew = s_anydrew != 0;
        if (!any_moved) break;
    }
    if (active && !skip) {
        changed = (cur != my_guess) || tab_short;         // (a void round never passes for the fixed point)
        my_new = cur & kLabMask;
        if (!FUSED) a.lab_out[gi] = (uint8_t)cur;
        if (tab_short) atomicOr(&rflags[FLAG_NTIES], kTabShort);
    }
    if (!FUSED) break;
    NEM_SWEEP_STAMP(2 + 4 * rnd);
    // ---- fused launch: publish this round's labels, meet the other blocks, look at what they changed
    // (the block's labels go out as whole words by agent-scope stores: what label_coherent reads on the other side)
    if ((int)threadIdx.x < spb / 4 && bx * spb + 4 * (int)threadIdx.x < a.n_local)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(out_buf + blk_lo) + threadIdx.x,
                           reinterpret_cast<const uint32_t*>(s_lab)[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (rnd == n_rounds - 1) break;                      // the launch's last round reports like a round of its own (below)
    {
        const int blk_changed = __syncthreads_or(changed ? 1 : 0);
        NEM_SWEEP_STAMP(3 + 4 * rnd);
        const int total = fused_meet(a.bar, nblk, rnd, blk_changed != 0, bx);
        NEM_SWEEP_STAMP(4 + 4 * rnd);
        if (total < 0) { fused_failed = true; break; }
        if (total == 0) break;                           // nothing changed anywhere: this round's output is the fixed point
        // the round changed something: its slot says so; the next one takes the next slot and the other buffer
        if (bx == 0 && threadIdx.x == 0) __hip_atomic_store(&rflags[FLAG_CHANGED], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rflags += FLAG_ROUND_STRIDE;
        guess_buf = out_buf;
        out_buf = (rnd & 1) ? a.lab_out : a.lab_out2;
        rnd++;
        my_guess = cur;
        changed = false;
        dirty = false;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (jlow[u] >= 0) {
                const int nl = label_coherent(guess_buf, jlow[u]) & kLabMask;
                if (nl != fl[u]) { fl[u] = nl; dirty = true; }
            }
        }
        NEM_SWEEP_STAMP(1 + 4 * rnd);                    // (rnd has moved on: the slot behind the meeting's)
    }
    }
    if (FUSED && fused_failed) {
        // a block gave up waiting: every slot from this round on says "changed" and why -- the loop control stops the
        // pipeline, the host redoes the sweep with one launch per round
        changed = true;
        if (threadIdx.x == 0)
            for (int q = 0; q < n_rounds - rnd; q++) atomicOr(&rflags[q * FLAG_ROUND_STRIDE + FLAG_CHANGED], 1 | kFusedFailed);
    }
    if (libc && !skip) {
        // the block's draws of this round, next to the labels they belong to: a guess is (labels, counts)
        if (blk_drew) {
            const uint64_t bal = __ballot(active && (cur & kLabDrew) != 0);
            if ((threadIdx.x & 63) == 0) s_drew[threadIdx.x >> 6] = bal;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            int cnt = 0;
            if (blk_drew) for (int w = 0; w < (int)(blockDim.x >> 6); w++) cnt += (int)__popcll(s_drew[w]);
            if (cnt != blk_cnt_guess) changed = true;
            a.tie_cnt_out[bx] = cnt;
            if (cnt > 0) atomicAdd(&rflags[FLAG_NTIES], cnt);
        }
    }
    } else if (active && !skip) {
    // ------------------------------------------------------------------------------------------
    // fuzzy NEM: one evaluation of every site against the guess
    // ------------------------------------------------------------------------------------------
    float ctx[KA];
    double pkf[KA];
#pragma unroll
    for (int k = 0; k < KA; k++) ctx[k] = 0.0f;
    // the densities do not depend on the graph: requested first, so that they travel while the neighbour lists do
#pragma unroll
    for (int k = 0; k < KA; k++) if (k < K) pkf[k] = a.pkfki[(size_t)k * a.npad + i];
    if (a.use_nei) {
        const int b = a.nei_ptr[i], e = a.nei_ptr[i + 1];
        for (int t = b; t < e; t++) {
            const int j = a.nei_idx[t];
            const float wt = a.nei_w[t];
            const float* row = ((j < gi) ? a.c_guess : a.c_old) + (size_t)j * K;
#pragma unroll
            for (int k = 0; k < KA; k++)
                if (k < K) ctx[k] = ctx[k] + (wt * row[k]);
        }
    }
    float cf[KA];
    zero_density = local_proba<KA>(a, K, pkf, ctx, cf);
    const float* g = a.c_guess + (size_t)gi * K;
    float* o = a.c_out + (size_t)gi * K;
#pragma unroll
    for (int k = 0; k < KA; k++) {
        if (k < K) {
            changed |= (__float_as_uint(cf[k]) != __float_as_uint(g[k]));
            o[k] = cf[k];
        }
    }
    }
    if (__any(changed) && (threadIdx.x & 63) == 0) s_chg = 1;
    // the iteration's bookkeeping, when it rides in this round (see SweepArgs): the site's label, "moved"
    int post_lab = 255;
    if (NCEM && a.post_on) {
        int moved = 0;
        if (active) {
            // (a round that skipped its sites and posts its own output reads that output back: not a case the
            //  engine enqueues, the verification round posts its guess)
            // (a round that skipped its sites -- a round before it changed nothing -- has no output of its own: the
            //  partition is in the sweep's first buffer, the guess of odd rounds and the output of even ones)
            if (skip) post_lab = (a.post_skip_guess ? my_guess : (int)a.lab_out[gi]) & kLabMask;
            else post_lab = a.post_from_guess ? (my_guess & kLabMask) : (my_new != 255 ? my_new : ((int)a.lab_out[gi] & kLabMask));
            if (a.post_moved) moved = (post_lab != my_old);
        }
        if (__any(moved) && (threadIdx.x & 63) == 0) s_mov = 1;
    }
    // zero-density sites (nem_alg.c:2603-2613): count and first index, one pair of atomics per block
    const uint64_t zmask = __ballot(zero_density);
    if (zmask != 0ull && (threadIdx.x & 63) == 0) {
        atomicAdd(&s_nzero, (int)__popcll(zmask));
        atomicMax(&s_first, a.n_total - (gi + (int)__ffsll((long long)zmask) - 1));   // lanes are consecutive sites
    }
    __syncthreads();
    // One same-address atomic per block is what a round costs when every site reports (all densities underflow at
    // D = 5000: 196 blocks x ~45 ns, 12 of a round's 26 us at 200 000 x 5 000).  Once the loop control has seen
    // such a sweep (ctrl[C_FOLD], set by ctrl_logic, read with the stop word) the tally rides on the last-block
    // counters instead.
    const bool fold = fold_hint != 0 && a.fold_ticket != nullptr && nblk > 32;
    if (threadIdx.x == 0 && s_nzero > 0) {
        if (!fold) atomicAdd(&rflags[FLAG_NZERO], s_nzero);
        if (rflags[FLAG_FIRSTZERO] < s_first) atomicMax(&rflags[FLAG_FIRSTZERO], s_first);   // first site = n_total - max
    }
    if (threadIdx.x == 0) {
        if (s_chg && __hip_atomic_load(&rflags[FLAG_CHANGED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(&rflags[FLAG_CHANGED], 1);
        if (NCEM && a.post_on && s_mov && __hip_atomic_load(&a.post_flags[FLAG_MOVED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
            atomicOr(&a.post_flags[FLAG_MOVED], 1);
    }
    if (NCEM && a.post_on) {                             // k_labels_post's work, see SweepArgs
        const int wave = i >> 6;
        if (wave < a.post_nw64 && !a.post_no_masks) {
            for (int k = 0; k < K; k++) {
                const uint64_t m = __ballot(post_lab == k);
                if ((threadIdx.x & 63) == 0) a.post_mask[(size_t)k * a.post_nw64 + wave] = m;
            }
        }
    }
    const bool post_ctrl = NCEM && a.post_on && a.post_ctrl.ctrl != nullptr;
    if (a.publish_byte != nullptr || post_ctrl || fold) {
        int* ticket = a.publish_byte != nullptr ? a.publish_ticket : (post_ctrl ? a.post_ctrl.ticket : a.fold_ticket);
        if (last_block_ticket(ticket, nblk, fold ? s_nzero : 0, fold ? &rflags[FLAG_NZERO] : nullptr)) {
            if (a.publish_byte != nullptr) {
                *a.publish_byte = (uint8_t)(__hip_atomic_load(&rflags[FLAG_CHANGED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
                if (NCEM && a.post_on && a.post_moved)          // (sharded: this rank's 'a label moved' byte rides next to it)
                    a.publish_byte[1] = (uint8_t)(__hip_atomic_load(&a.post_flags[FLAG_MOVED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0);
                if (LIBC && a.rank_tot_out != nullptr)           // (... and its draws of this round)
                    *reinterpret_cast<int*>(a.rank_tot_out) =
                        __hip_atomic_load(&rflags[FLAG_NTIES], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (kTabShort - 1);
            }
            if (post_ctrl) ctrl_logic(a.post_ctrl);
        }
    }
    NEM_SWEEP_STAMP(31);
#undef NEM_SWEEP_STAMP
}

}  // namespace nemk
