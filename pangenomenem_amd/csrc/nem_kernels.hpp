// nem_kernels.hpp -- kernel launch interface (device pointers only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/nem_mi355x.h"

namespace nemk {

constexpr int kMaxKernelK = 32;
constexpr int kTicketWords = 32 * 33;   // last-block ticket: top counter + 32 group counters, 128 bytes apart
constexpr double kEpsilonD = 1e-20;   // EPSILON, reference nem_typ.h:63

// per-round flag slot of an E2 sweep
// FLAG_NTIES (TIE_LIBC): random draws the round's sites made; bit 30: a draw fell outside the draw table
enum { FLAG_CHANGED = 0, FLAG_NZERO = 1, FLAG_FIRSTZERO = 2, FLAG_NTIES = 3, FLAG_ROUND_STRIDE = 4 };
// per-iteration flag block.  FLAG_EMPTYK is overwritten by every k_mstep_disp; FLAG_MOVED and the
// relaxation-round window that follows this block are zeroed by k_density, which precedes every
// sweep of the EM loop (so no memset launches are needed).
// FLAG_FAULT: a kernel could not finish its work (bit 0: a hand-over of k_mstep_fuzzy_pc never completed).  Sticky: only
// a reset clears it; the host turns it into NEMGPU_E_INTERNAL wherever it reads the flags back.
enum { FLAG_EMPTYK = 0, FLAG_EMPTY_PROP = 1, FLAG_FAULT = 2, FLAG_MOVED = 3, FLAG_ITER_STRIDE = 4 };

// device-side loop control block (ints), see k_ctrl
// C_FOLD sits next to C_STOP so that a sweep block reads both with one 8-byte scalar load: "the last sweep met
// zero-density sites" (how the next sweeps tally them, see k_sweep); it survives a restart.
enum { C_STOP = 0, C_FOLD = 1, C_ITERS = 10, C_COMMITS = 2, C_STATUS = 3, C_EMPTYK = 4, C_CONVERGED = 5, C_NEED_ROUNDS = 6,
       C_SWEEP_ROUNDS = 7, C_NZERO = 8, C_FIRSTZERO = 9, C_DRAWS = 11, C_DEEP = 12, C_DRAWS_INIT = 13, C_INIT_ROUNDS = 14, C_WORDS = 16 };
// C_DRAWS_INIT (TIE_LIBC): of C_DRAWS, what the run's two initial sweeps drew; C_INIT_ROUNDS: the rounds they needed (blind << 8 | beta)
// C_DEEP: the last iteration of the batch (1-based) whose sweep used more than two of its relaxation rounds

struct CtrlArgs {
    int* ctrl; const int* iter_flags; const int* round0;   // round0: flag slot of relaxation round 0 (round r: r slots on)
    int n_rounds;                  // relaxation rounds enqueued per sweep (0: two)
    int param_fix, use_nei, cvtest, ncem; float cvthres;
    int* sweep_next;               // device word: number of the next sweep (tie-break hash key)
    int* ticket;                   // last-block-done counter (self-resetting)
    // sharded runs: every rank's "some label changed" byte for round 0 / round 1, as all-gathered at the tail
    // of each rank's block of the label arrays (nullptr on a single GPU: the int flags above are used)
    const uint8_t* q_flags; const uint8_t* r_flags; int n_ranks, flag_stride;
    int moved_bytes;               // sharded: the byte behind a rank's flag byte says 'one of its labels moved' (1: in r_flags, 2: in q_flags)
    int is_init;                   // the two initial sweeps: no iteration is counted, the sweep number becomes 2
    const int* blind;              // is_init: flag slot of the blind beta = 0 sweep (its zero-density tally), or nullptr
    int blind_rounds;              // is_init, TIE_LIBC: the blind sweep ran this many verified rounds, their slots from `blind` on (0: one slot, nothing to verify)
    int* draw_ctl;                 // TIE_LIBC: {draws made so far, first draw of the table} (device), else nullptr
    const uint8_t* q_tot;          // TIE_LIBC, sharded: rank 0's draw total in the tail of the sweep's first buffer (flag_stride apart)
};
void launch_ctrl(const CtrlArgs& a, hipStream_t s);

struct SweepArgs {
    int n_local, lo, n_total, K, npad, use_nei;
    const int* nei_ptr; const int* nei_idx; const float* nei_w;
    float beta;
    const double* pkfki;                       // [K][npad], local families
    // NCEM state: labels, GLOBAL family indexing
    const uint8_t* lab_old; const uint8_t* lab_guess; uint8_t* lab_out;
    // fuzzy state: float rows [n_total][K], GLOBAL family indexing
    const float* c_old; const float* c_guess; float* c_out;
    int tie_rule; uint32_t tie_seed; uint32_t sweep_id;
    const int* sweep_id_ptr;                   // when set, the sweep number is read from the device instead
    int* flags;                                // this round's slot
    const int* prev_changed;                   // previous round's FLAG_CHANGED (nullptr for round 0)
    const int* stop;                           // loop-control stop word (nullptr outside the pipelined loop)
    // sharded runs: label arrays are laid out in per-rank blocks of `slot_stride` bytes (labels + flag tail);
    // flags_in = previous round's all-gathered flag bytes (round exits when none is set)
    const uint8_t* flags_in; int n_ranks, slot_stride, slot_pad;
    // NCEM, pipelined loop: the iteration's bookkeeping (k_labels_post: class masks, "moved" flag, loop control in
    // the last block) folded into the last enqueued round.  post_from_guess: the final labels are this round's
    // guess (a verification round: if it changes anything the loop control stops the pipeline and the host redoes
    // the bookkeeping), else this round's output.
    // post_skip_guess: where the partition is when this round skips its sites (a round before it changed nothing):
    // the sweep's first buffer -- this round's guess if it is an odd round, its output buffer if it is an even one
    int post_on, post_from_guess, post_moved, post_nw64, post_skip_guess;
    int post_no_masks;             // the bookkeeping without the class masks (a verifying round whose guess has them already)
    uint64_t* post_mask; int* post_flags;
    CtrlArgs post_ctrl;
    // sharded runs: the last block to finish stores this rank's "some label changed in this round" byte behind
    // its label block (the label all-gather then carries it to every rank)
    uint8_t* publish_byte; int* publish_ticket;
    // zero-density tally of a large grid (see k_sweep): last-block counters to fold it through, or nullptr
    int* fold_ticket;
    // TIE_LIBC (the reference's own tie stream, nem_alg.c:617-637 -> nem_rnd.c:53-61): the site that ties draws
    // random() number  draws before the sweep + sites below it that drew in this sweep.  draw_tab holds draws
    // draw_tab0 .. draw_tab0 + draw_tab_len - 1 of the stream; draws-before-the-sweep and draw_tab0 come from
    // draw_ctl[0..1] (device, pipelined loop) or by value; draw_extra: a count to add (a preceding sweep's draws
    // that no loop control has booked yet).  tie_cnt_*: draws per block of the guess / of this round's output.
    const uint32_t* draw_tab; int draw_tab_len, draw_base, draw_tab0;
    const int* draw_ctl; const int* draw_extra;
    const int* tie_cnt_guess; int* tie_cnt_out;
    // ... sharded: the sites of the ranks below come first in the draw order.  Every rank's draws of the round that
    // produced a label buffer ride in its block's tail (an int32), all-gathered with the labels: rank_tot_in = rank 0's
    // in the GUESS buffer (the others slot_stride bytes apart), rank_index = this rank, rank_tot_out = this rank's slot
    // in the output buffer (written by the last block of the round)
    const uint8_t* rank_tot_in; int rank_index; uint8_t* rank_tot_out;
    // sites per block of the large-shard kernel instances (set by launch_sweep; 0: the kernel's block size)
    int spb;
    // ---- k_sweep_fused: up to `fused_rounds` relaxation rounds in ONE launch (NCEM, one engine, every block resident).
    // Round t of the launch (t = 0: this block's `flags`, `lab_guess`, `lab_out` as above) writes flag slot t
    // (FLAG_ROUND_STRIDE ints further on each) and the label buffer lab_out (t even) / lab_out2 (t odd) -- exactly what
    // `fused_rounds` separate launches would have written, so that the loop control and the host go on from there if the
    // rounds were not enough.  Between two rounds the blocks meet through bar[2][kFusedMaxBlocks] (one word per block
    // and round parity, see fused_meet; zero before the sweep) and re-read only the labels of their sites'
    // lower-indexed neighbours in OTHER blocks; a site whose inputs did not change keeps its label without an evaluation.
    int fused_rounds;
    uint8_t* lab_out2;
    unsigned* bar;
    // exp((double)beta * (double)(float)m), m = 0 .. exp_tab_len - 1, made once per beta by k_exp_table with the device
    // exp the sweep itself would call (nullptr: the block computes the first entries itself)
    const double* exp_tab; int exp_tab_len;
    // development probe (NEM_MI355X_SWEEP_PROF=1): the 100 MHz clock at the phase boundaries of a fused launch, first and
    // last block (nemgpu_sweep_phases)
    unsigned long long* prof;
};
// bits of a round's FLAG_CHANGED word besides bit 0
constexpr int kFusedFailed = 1 << 28;   // k_sweep_fused: a block gave up waiting for the others (not every block resident?):
                                        // the launch's rounds are void, the host redoes the sweep with one launch per round
constexpr int kFusedMaxBlocks = 256;    // one block per CU at most: every block of a fused launch must be resident
constexpr int kFusedMaxRounds = 16;     // (the pipelined loop uses 4: ctrl_logic's window)
constexpr int kExpTabGlobal = 4096;     // entries of SweepArgs::exp_tab
// argument blocks of the kernels whose launch wrappers take scalars (the batched launches need them as structs)
struct LabelsPostArgs { int n_local, lo, K, nw64; const uint8_t* lab_new; const uint8_t* lab_old; uint64_t* mask; int* flags;
                        const int* stop; CtrlArgs ca; };
struct CountsArgs { int K, D, nw64; const uint64_t* xt; const uint64_t* mask; int* stats; const int* stop; CtrlArgs prev_ctrl; };
struct FuzzyArgs { int n, npad, K, D; const uint32_t* xw; const uint64_t* xt; int nw64; const float* c; float* nbobs_k;
                   float* in0; float* in1; float* inh_k; int* lastz; int* any1; float* center; float* iner; const int* stop;
                   float* ct; int ctpad;      // ct: class-major copy of c, [K][ctpad] (nullptr: the one-lane-per-chain kernels)
                   float* chk;                // producer/consumer kernels: the zeros' chains every 64 families, [K][nwin + 1][64 DB]
                   int* fault;                // the iteration flags' FLAG_FAULT word (nullptr: faults go unreported)
                   int inject; };             // test hook (NEM_MI355X_FAULT_INJECT=fuzzy_pc): one producer skips a hand-over
struct ConvFuzzyArgs { size_t m; const float* c; const float* cold; float thres; int* flags; const int* stop; CtrlArgs ca; };
struct OnehotArgs { int n, K; const uint8_t* lab; float* c; };
struct CritArgs { int n, K, npad; const int* nei_ptr; const int* nei_idx; const float* nei_w; int use_nei; float beta;
                  const float* c; const double* pkfki; const float* logpkfki; float* dik; float* gik; double* lfi; double* lzi;
                  float* crit6; int hard; };
struct FillArgs { int* ptr; int words; int value; };
struct CopyArgs { const int* src; int* dst; int words; };
// up to eight device-to-device copies in one launch (the logged run's set-aside behind an iteration; not recordable)
struct CopySegsArgs { const int* src[8]; int* dst[8]; int words[8]; int n; };
void launch_copy_segments(const CopySegsArgs& a, hipStream_t s);
struct LayoutArgs { const uint32_t* xf; const int* perm; int n, wf, W, npad, d, nw64; uint32_t* xw; uint32_t* xws; uint64_t* xt; };

// ---- batched launches: B independent problems per launch ---------------------------------------------------------
// Every loop kernel has a twin that takes an ARRAY of argument blocks in device memory and runs problem blockIdx.z
// of it.  A host thread that sets a Recorder gets the launches of the code it then runs RECORDED instead of issued
// (same argument blocks, same grids); the batch driver (nem_engine.hip) records one sequence per problem, checks that
// the sequences agree launch for launch, and issues each position once for all problems with launch_zipped.
enum OpKind { OP_FINISH = 1, OP_DENSITY, OP_DENSITY_FUSED, OP_SWEEP, OP_COUNTS, OP_LABELS_POST, OP_CTRL, OP_FUZZY_A, OP_FUZZY_B,
              OP_CONV_FUZZY, OP_ONEHOT, OP_CRIT_TERMS, OP_CRIT_REDUCE, OP_CRIT_FINAL, OP_FILL, OP_COPY, OP_FUZZY_T, OP_FUZZY_SUMS, OP_FUZZY_MED,
              OP_FUZZY_PC, OP_FUZZY_MED2, OP_LAYOUT_WORDS, OP_LAYOUT_BITS };
constexpr int kOpArgBytes = 512;
struct OpRecord {
    int kind, variant;             // variant: template instance / block size, part of what must agree across problems
    unsigned gx, gy, block;
    int nbytes;
    alignas(16) unsigned char args[kOpArgBytes];
};
struct Recorder { std::vector<OpRecord> ops; };
void set_recorder(Recorder* r);    // thread-local; nullptr: launches are issued
Recorder* current_recorder();
// position `kind`/`variant` for B problems: dev_args = B argument blocks `stride` bytes apart, dev_gx[B] = each
// problem's own grid width (blocks beyond it return at once)
void launch_zipped(int kind, int variant, int B, const void* dev_args, int stride, const int* dev_gx, unsigned max_gx,
                   unsigned gy, unsigned block, hipStream_t s);
void launch_fill(int* ptr, int words, int value, hipStream_t s);   // recordable memset of 32-bit words
void launch_copy_words(const int* src, int* dst, int words, hipStream_t s);   // recordable device-to-device copy


void launch_layout(const uint32_t* xf, int n, int wf, int W, int npad, int d, int nw64, uint32_t* xw, uint64_t* xt,
                   const int* perm, uint32_t* xws, hipStream_t s);
// parameter update + density tables (k_finish); also carries the table pointers k_density reads
struct FinishArgs {
    int mode;                      // 0: tables only; 1: NCEM centres from counts + dispersion + tables; 2: dispersion + tables
    int K, D, dpad, n_total, disper, propor;
    const int* stats;              // NCEM counts {N_k, S1[k][d]}: stats_ranks partial arrays, stats_rank_stride ints apart
    int stats_ranks, stats_rank_stride;   // (1, 0 on a single GPU; sharded: every rank's partial counts, summed on read)
    float* prop; float* center; float* disp; float* nbobs_k; float* iner;
    double2* tabT; double* tabL0; uint32_t* nz0; uint32_t* nz1;
    uint32_t* am0; uint32_t* am1; double2* uni; int* nonuni;
    double* pk; float* logpk; int* flags;
    const int* stop;
    // head of a restart batch (mode 0, reset_prop != nullptr): reload the initial parameters and clear the loop
    // control words in this launch instead of five copy / fill operations
    const float* reset_prop; const float* reset_center; const float* reset_disp;
    int* reset_ctrl; int reset_ctrl_words; int* reset_sweep_next;
    const int* perm;               // density kernels: lane i of the (sorted) matrix copy is family perm[i]
    int use_ff;                    // density kernels: fast-forward the uniform chain inside float binades (nem_ff.hpp)
    uint2* ffq;                    // [K][256] (q0, q1 - q0): the fast-forward increments per class, built next to the tables
};
void launch_finish(const FinishArgs& a, hipStream_t s);
void launch_density(const FinishArgs& t, const uint32_t* xw, int n, int npad, double* pkfki, float* logpkfki,
                    int* zero_flags, int n_zero_flags, hipStream_t s);
// NCEM, sk_/skd, D <= kFusedMaxD: parameter update from t.stats folded into the density kernel (no k_finish)
constexpr int kFusedMaxD = 1024;   // beyond this the per-block parameter derivation costs more than a k_finish launch
void launch_density_fused(const FinishArgs& t, const uint32_t* xw, int n, int npad, double* pkfki, float* logpkfki,
                          int* zero_flags, int n_zero_flags, hipStream_t s);
void launch_sweep(const SweepArgs& a, bool ncem, hipStream_t s);
// blocks a launch of launch_sweep would use for n_local sites (what decides whether the fused form may be used)
int sweep_grid_blocks(int n_local, int K);
bool sweep_fused_has_instance(int K);
int sweep_phases_read(unsigned long long* out64);       // the probe's stamps of the last fused launch (64 words), or -1
void launch_exp_table(float beta, double* tab, int len, hipStream_t s);   // SweepArgs::exp_tab (recordable)
// one NCEM relaxation round and the M-step counts (of the partition whose class masks exist already) in ONE launch;
// returns false when the shape has no such kernel (2 <= K <= 5, fewer than 65 536 families, not recordable)
bool launch_sweep_counts(const SweepArgs& sw, int K, int D, int nw64, const uint64_t* xt, const uint64_t* mask, int* stats,
                         const int* stop, hipStream_t s);
void launch_labels_post(int n_local, int lo, int K, int nw64, const uint8_t* lab_new, const uint8_t* lab_old,
                        uint64_t* mask, int* flags, const int* stop, const CtrlArgs* ctrl, hipStream_t s);
void launch_mstep_counts(int K, int D, int nw64, const uint64_t* xt, const uint64_t* mask, int* stats,
                         const int* stop, const CtrlArgs* prev_ctrl, hipStream_t s);
void launch_mstep_fuzzy(int n, int npad, int K, int D, const uint32_t* xw, const uint64_t* xt, int nw64, const float* c,
                        float* ct, float* nbobs_k, float* in0, float* in1, float* inh_k, int* lastz, int* any1, float* center,
                        float* iner, const int* stop, hipStream_t s, float* chk = nullptr, int* fault = nullptr, int inject = 0);
void launch_conv_fuzzy(size_t m, const float* c, const float* cold, float thres, int* flags, const int* stop,
                       const CtrlArgs* ctrl, hipStream_t s);
void launch_chain_debug(const double* x, long long n, float init, float* out, hipStream_t s);
bool launch_halfsum_debug(const float* x, int n, int waves, float* out, hipStream_t s, long long* stamps = nullptr);   // n <= 8192, waves 1 | 16; out[2], stamps[6] when timing
void launch_calib_read(const uint32_t* buf, size_t words, uint32_t* sink, hipStream_t s);
void launch_onehot(int n, int K, const uint8_t* lab, float* c, hipStream_t s);
constexpr int kCritReduceThreads = 1024;   // block size of k_crit_reduce (CH_T in nem_kernels.hip)
void launch_criteria(int n, int K, int npad, const int* nei_ptr, const int* nei_idx, const float* nei_w, int use_nei,
                     float beta, const float* c, const double* pkfki, const float* logpkfki, float* dik, float* gik,
                     double* lfi, double* lzi, float* crit6, int hard, hipStream_t s);   // hard: one-hot rows (NCEM)

// the head of every batched kernel twin: problem = blockIdx.z, its own grid width, its argument block
#define NEM_B_HEAD(Args)                                                   \
    const int p__ = blockIdx.z;                                            \
    const int nblk = gx[p__];                                              \
    if ((int)blockIdx.x >= nblk) return;                                   \
    const Args a = *reinterpret_cast<const Args*>(reinterpret_cast<const char*>(arr) + (size_t)p__ * stride);

// a launch wrapper's first step: with a recorder set on this thread the launch is stored, not issued
template <typename Args>
inline bool record_op(int kind, int variant, dim3 grid, unsigned block, const Args& a)
{
    static_assert(sizeof(Args) <= kOpArgBytes, "argument block too large for an OpRecord");
    static_assert(std::is_trivially_copyable<Args>::value, "argument blocks are copied byte for byte");
    Recorder* r = current_recorder();
    if (r == nullptr) return false;
    r->ops.emplace_back();
    OpRecord& o = r->ops.back();
    o.kind = kind; o.variant = variant; o.gx = grid.x; o.gy = grid.y; o.block = block; o.nbytes = (int)sizeof(Args);
    memcpy(o.args, &a, sizeof(Args));
    return true;
}
// the batched twin of a recorded k_sweep position (nem_sweep.hip)
void sweep_dispatch_batched(int variant, dim3 grid, unsigned bdim, hipStream_t s, const void* arr, int stride, const int* gx);

}  // namespace nemk
