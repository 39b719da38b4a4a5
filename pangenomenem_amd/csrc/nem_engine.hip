// nem_engine.hip -- the in-memory NEM engine: device buffers, the EM driver and the nemgpu_* C ABI.
//
// Mirrors the reference's in-memory path ClassifyByNem -> ClassifyByNemOneBeta(INIT_PARAM_FILE)
// -> NemAlgo (/root/reference/ppanggolin/NEM/nem_alg.c:546-584, 1151-1169, 1746-1879) with all
// numeric work in the kernels of nem_kernels.hip.  There is no CPU compute path here: every
// entry point returns NEMGPU_E_DEVICE when HIP is unusable.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <pthread.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "nem_internal.hpp"
#include "nem_ff.hpp"
#include "nem_chain.hpp"
#include "nem_halfsum.hpp"
#include "nem_rng.hpp"
#include "nem_kernels.hpp"
#include "nem_chunks.hpp"

using namespace nemk;

namespace nemk {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
}  // namespace nemk

#define HIPCHK(call)                                                                            \
    do {                                                                                        \
        hipError_t err__ = (call);                                                              \
        if (err__ != hipSuccess) {                                                              \
            set_error(std::string(#call) + " failed: " + hipGetErrorString(err__));             \
            return NEMGPU_E_DEVICE;                                                             \
        }                                                                                       \
    } while (0)

namespace {
constexpr int kRoundCap = 64;     // relaxation rounds per flag window
// what is zeroed ahead of every sweep (by the density launch, or a fill): FLAG_MOVED, the round window, the meeting words
constexpr int kMeetWords = 2 * kFusedMaxBlocks;
constexpr int kSweepFlagWords = 1 + kRoundCap * FLAG_ROUND_STRIDE + kMeetWords;
constexpr int kRoundBatchMax = 4;
constexpr int kRoundsMax = 7;       // most rounds a sweep of the pipelined loop is given ahead (ctrl_logic reads 8 slots)
constexpr int kFzPositions = 16;    // fuzzy NEM: leading iterations of a run whose round counts are learned
}  // namespace

struct nemgpu_engine {
    int n_total = 0, d = 0, k = 0, lo = 0, hi = 0, n = 0;
    int n_true = 0;               // families of the whole problem (= n_total unless label slots carry padding)
    void* rccl_comm = nullptr;           // native RCCL communicator of a sharded engine (nemgpu_rccl_attach)
    int sh_world = 1, sh_rank = 0, sh_blk = 0, sh_stride = 0;   // sharded label-slot layout (stride 0 = plain)
    const uint8_t* shard_lab[3] = {nullptr, nullptr, nullptr};   // the driver's three all-gathered label arrays (TIE_LIBC: tie_cnt[b] goes with shard_lab[b])
    int sh_tot_off() const { return ((sh_blk + 3) & ~3) + 4; }   // a rank's int32 draw total inside its block (behind the flag bytes)
    int npad = 0, dpad = 0, W = 0, wf = 0, nw64 = 0, device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    nemgpu_config cfg{};
    bool have_matrix = false, have_params = false, has_graph = false;
    bool reset_pending = false;          // set_params / configure: the device half of the reset is still to do
                                         // (a run that starts with the restart launch does it there)

    uint32_t* xw = nullptr;
    // E1's own copy of the matrix: uint4[ceil(W/4)][npad], lane i holds family perm[i].  Inside every 256-family
    // tile (one block of the density kernels) the families are ordered by their number of present organisms, so
    // the lanes of a wave run similar chains and cross float binades in the same words (what keeps the
    // fast-forward of nem_ff.hpp coherent across a wave; a global sort gains almost nothing over the tile-local
    // one).  The block puts its 256 results back in family order through LDS, so its stores are contiguous and
    // nothing outside the density kernels sees the order.
    uint32_t* xws = nullptr;
    int* perm = nullptr;
    // the family-major bit rows as uploaded (random starts pick centres from them): pinned when they fit the pool's
    // staging limit -- then the upload is a true asynchronous copy out of them -- else host_bits_own's
    uint32_t* host_bits = nullptr; size_t host_bits_words = 0, host_bits_pin = 0;
    std::vector<uint32_t> host_bits_own;
    // pinned sources of uploads that may still be in flight on the stream (matrix order, graph, parameters): they go
    // back to the pool when the engine is destroyed or an input is replaced (after a wait)
    struct Staged { char* p; size_t size; };
    std::vector<Staged> staging;
    bool use_sort = true;
    uint64_t* xt = nullptr;
    int *nei_ptr = nullptr, *nei_idx = nullptr;
    float* nei_w = nullptr;
    int nnz = 0;

    float *prop = nullptr, *center = nullptr, *disp = nullptr;
    float *prop0 = nullptr, *center0 = nullptr, *disp0 = nullptr;
    // prop | center | disp | nbobs_k live in ONE block (sections 256-byte aligned), and so do the initial values
    // (with a zero tail where nbobs_k is): a reset, an upload and a download are one copy each
    size_t par_o_center = 0, par_o_disp = 0, par_o_nb = 0, par_words = 0;
    float *nbobs_k = nullptr, *iner = nullptr;
    int *fz_lastz = nullptr, *fz_any1 = nullptr;
    float *fz_in0 = nullptr, *fz_in1 = nullptr, *fz_inh = nullptr;
    float* fz_ct = nullptr;       // class-major copy of the memberships, [k][npad] (the wave-per-chain M-step)
    float* fz_chk = nullptr;      // the zeros' chains every 64 families, [k][ceil(n/64) + 1][64 ceil(d/64)] (producer/consumer M-step)
    int fuzzy_chains = 2;         // NEM_MI355X_FUZZY_CHAINS: 2 producer/consumer (lane per chain, adds only), 1 wave per chain, 0 lane per chain as in round 1
    double2* tabT = nullptr;
    double* tabL0 = nullptr;
    uint32_t *nz0 = nullptr, *nz1 = nullptr, *am0 = nullptr, *am1 = nullptr;
    double2* uni = nullptr;
    int* nonuni = nullptr;
    uint2* ffq = nullptr;         // [k][256] fast-forward increments of the classes on the uniform chain (k_finish)
    int* sweep_next = nullptr;    // device word: number of the next E-step sweep (tie-break hash key)
    bool tables_fresh = false;    // density tables match prop/center/disp
    bool density_fresh = false;   // pkfki / logpkfki match the tables
    double* pk = nullptr;
    float* logpk = nullptr;
    double* pkfki = nullptr;
    float* logpkfki = nullptr;

    uint8_t* lab[3] = {nullptr, nullptr, nullptr};   // NCEM partitions (labels), n_total each
    // TIE_LIBC (the reference's tie stream): draws per sweep block of each label buffer; the stream's table on the
    // device (draws draw_tab0 .. +draw_cap-1 of glibc random() after srandom(tie_seed)); {draws made, draw_tab0} for
    // the pipelined loop (device words); draws made so far (host)
    int* tie_cnt[3] = {nullptr, nullptr, nullptr};
    uint32_t* draw_tab = nullptr; int draw_cap = 0; long draw_tab0 = 0; bool draw_valid = false;
    std::vector<uint32_t> draw_host; nemk::GlibcRandom draw_gen{1}; long draw_gen_pos = 0;
    int* draw_ctl = nullptr;
    float crit_ref = 0.0f;               // CVTEST_CRIT: the criterion the next iteration's is compared with
    int draws = 0; bool tie_heavy = false;
    bool draw_borrowed = false;          // the table is the parent's (lock-step random starts): never written through this engine
    int draws_after_init = 0;                  // TIE_LIBC: the stream's position behind a run's two initial sweeps
    char* rs_par_host = nullptr; size_t rs_par_host_size = 0;   // random starts under TIE_LIBC: the starts' parameters, pinned
    // NCEM, stateless tie rules: rounds enqueued for a start's beta sweep.  round_batch (three) until three starts in a row
    // were through in two -- the third launch then only finds out that it has nothing to do, 4.4 us of every restart;
    // a start that needs the third round after that is finished from the host and the count goes back up for good.
    int init_rounds_ncem = 0; int init_two_streak = 0; bool init_rounds_locked = false;
    int libc_ra = 3, libc_rb = 3;              // TIE_LIBC, pipelined start: rounds enqueued for the blind / the beta initial sweep (batch_plan sets them; part of a first batch's graph)
    const int* draw_extra_once = nullptr;      // TIE_LIBC: the next sweep set up adds this device word to the draws before it (a sweep enqueued behind one whose count the host has not seen)
    int libc_init_hist[2][17] = {};            // TIE_LIBC random starts: how many relaxation rounds the blind / the beta initial sweep of the starts so far needed (16: more)
    int rs_two_waits = 0;                      // ... starts of phase A whose blind sweep was not through in the rounds enqueued (done again, sweep by sweep)
    int rs_rounds = 0, rs_lockstep = 0, rs_alone = 0, rs_redone = 0;   // the last nemgpu_run_random: lock-step rounds, starts that stood in them, starts run alone, starts thrown away
    bool libc() const { return cfg.algo == NEMGPU_ALGO_NCEM && cfg.tie_rule == NEMGPU_TIE_LIBC; }
    float* cbuf[3] = {nullptr, nullptr, nullptr};    // fuzzy partitions, n_total*k each
    int cur = 0;
    uint64_t* mask = nullptr;
    int* stats = nullptr;
    int* flags_dev = nullptr;     // [C_WORDS loop control] [FLAG_ITER_STRIDE] [kRoundCap * FLAG_ROUND_STRIDE] [2 * kFusedMaxBlocks meeting words of the fused sweep]
    int* flags_host = nullptr;    // pinned mirror
    const int* stop_ptr = nullptr;   // &ctrl[C_STOP] while the pipelined loop is being enqueued, else nullptr

    float* c_onehot = nullptr;                        // lazily allocated (criteria / NCEM)
    float *crit_dik = nullptr, *crit_gik = nullptr, *crit6_dev = nullptr;
    double *crit_lfi = nullptr, *crit_lzi = nullptr;
    // a second set: the logged run evaluates the criteria of two partitions in the same launches (criteria_pair_enqueue)
    float *crit2_dik = nullptr, *crit2_gik = nullptr, *crit2_6 = nullptr, *c_onehot2 = nullptr;
    double *crit2_lfi = nullptr, *crit2_lzi = nullptr;
    char* crit_pair_dev = nullptr;                   // the two argument blocks of each of those launches

    // run state
    uint32_t sweep_counter = 0;
    int iters = 0, converged = 0, emptyk = 0, status = NEMGPU_OK;
    int zero_density = 0, first_zero = -1, sweep_rounds = 0;
    bool masks_valid = false;
    bool flags_clean = false;     // MOVED + round window are zero (set by k_density, consumed by a sweep)

    // captured batches of the pipelined loop, keyed by (current buffer, iterations in the batch)
    hipGraphExec_t graphs[2][3][8][8] = {};   // [with initial sweeps][current buffer][iterations][leading iterations with one round more]
    uint8_t graph_asked[2][3][8][8] = {};   // how often a batch shape was enqueued before it got a graph
    bool use_graphs = true;
    // relaxation rounds enqueued per sweep before anybody looks (round 0, its verification, and one more that costs
    // an early-exit launch when it is not needed and a host round trip when it is missing); NEM_MI355X_ROUNDS=2..4
    int round_batch = 3;
    // ... per sweep of an ITERATION: two (the round and its verification) until a sweep of this engine needed more --
    // then every later one gets round_batch.  (Labels are sticky: on the bench's pre-convergence data only the sweep
    // that starts from the blind partition needs a third round; an early-exit launch per iteration costs 2.5 us.)
    int rounds_iter = 2;
    // k_sweep_fused (NEM_MI355X_FUSED_SWEEP=1 turns it on): the first rounds of a sweep in ONE launch, the blocks
    // meeting between rounds (nem_sweep_dev.hpp).  OFF by default -- measured, round 4 (profiles/r04_fused_sweep_*.json):
    // a meeting of 79-241 blocks costs 3-3.5 us on the in-kernel clock whichever way it is built (one counter: 79
    // same-address atomics; one word per block + a wave-wide poll: a store's way to the other XCDs and two polls), a
    // 2-round sweep needs two of them, and a launch boundary with its prologue is no dearer: 20 000 x 500 0.0391 vs
    // 0.0370 ms per EM iteration, 50 000 x 1 000 0.0512 vs 0.0477, 200 000 x 5 000 0.1227 vs 0.1247.
    // fused_rounds: rounds per such launch (NEM_MI355X_FUSED_ROUNDS, at most kFusedMaxRounds; the pipelined loop takes
    // min(4, .): its loop control looks at four slots).  fused_sweep goes off for good when a launch's blocks did not all
    // meet (kFusedFailed: the grid was not resident as a whole).
    bool fused_sweep = false; int fused_rounds = 4; int n_fused_failed = 0, n_fused = 0;
    double* exp_tab = nullptr; float exp_beta = 0.0f; bool exp_ready = false;   // SweepArgs::exp_tab for cfg.beta
    int exp_need = kExpTabGlobal;        // entries a context of this graph can index: 1 + the largest row sum of |weights| (set_graph)
    // ... but not all iterations are alike: the first ones after a start move many labels and tend to need the
    // extra round, the later ones almost never do.  The first deep_iters iterations of a run get round_batch rounds;
    // an iteration further on whose sweep the host had to finish moves the mark (it is kept across restarts: the
    // bench and PPanGGOLiN's repeated solves of similar problems find the same pattern every time).
    int deep_iters = 2;
    // Fuzzy NEM: the sweeps of a run's FIRST iterations need more rounds than the later ones (memberships move
    // everywhere after a start), and a sweep that runs out of enqueued rounds stops the pipeline for a host round trip
    // (round 3: 162 of them in 2 700 timed steps).  fz_need[p]: rounds to enqueue for iteration p of a run (0: the
    // default rule), raised to what the host had to add when it finished that iteration's sweep; fz_init_need: the
    // same for the start's beta sweep.  Kept across restarts, like deep_iters.
    uint8_t fz_need[kFzPositions] = {}; uint8_t fz_init_need = 0;
    int run_deep_used = 0; bool run_tracked = false;   // this run: the last iteration that used the extra round
    bool capture_first = false;          // capture a batch shape the first time it is enqueued (nemgpu_set_graph_policy)
    int n_plain = 0, n_captured = 0, n_replayed = 0, n_host_rounds = 0;   // nemgpu_graph_counters
    int ff_mode = -1;                    // density: binade fast-forward of the uniform chain (nem_ff.hpp): 0 off, 1 on, -1 auto
    // auto: on from 256 organisms (below that the chain is mostly the small-binade prefix that is stepped anyway
    // and the table build is pure overhead).  Measured on MI355X: 20k x 500 on par with plain stepping
    // (latency-bound, one wave per SIMD), 50k x 1000 1.7x, 200k x 5000 9x faster.
    // parameter update folded into the density launch: per-organism dispersions (skd) and one dispersion per class
    // (sk_).  Where InerToDispK_'s d-ordered sums can round (N*D > 2^24) every block redoes the two D-step chains
    // -- at most 1024 steps, 4.4 us, in parallel in all blocks -- which is still cheaper than k_finish's launch.
    bool fused_update() const
    {
        return d <= kFusedMaxD && (cfg.disper == NEMGPU_DISP_KD || cfg.disper == NEMGPU_DISP_K_);
    }
    bool use_ff() const { return ff_mode < 0 ? d >= 256 : ff_mode != 0; }
    // a kernel's FLAG_FAULT seen on the host (cleared, device word included, by the next reset / restart);
    // fault_inject: NEM_MI355X_FAULT_INJECT=fuzzy_pc makes one producer of k_mstep_fuzzy_pc skip a hand-over (tests)
    bool fault_seen = false; int fault_inject = 0;

    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // nemgpu_profile_density
    hipEvent_t ev_flags = nullptr;             // libc_init_one_wait: the flag block has reached the host
    bool defer_layout = false, layout_pending = false;   // nemgpu_solve_many: the device layouts are made by the group's run (zipped)
    hipEvent_t ready_ev = nullptr;             // nemgpu_solve_many: recorded behind the engine's uploads on its builder's stream,
    bool ready_pending = false;                // which carries other engines' uploads too -- a run waits for the event, not the stream

    // Device memory comes from a few large zeroed chunks (one hipMalloc + one fill each) that buffers are carved
    // from and that live as long as the engine: a nem() call creates and destroys an engine, and ~50 hipMalloc /
    // fill / hipFree triples were a fifth of its time at configs[1].
    struct Chunk { char* base; size_t size, used; bool owned = true; };
    std::vector<Chunk> chunks;
    int shared_chunk = -1;                     // the chunk small buffers are carved from
    uint8_t* best_lab = nullptr; float* best_c = nullptr;   // nemgpu_run_random: the best start's partition
    uint32_t* xf_stage = nullptr;                            // nemgpu_set_matrix_bits: upload staging (small matrices)
    bool ctrl_pending = false; CtrlArgs ctrl_deferred{};     // loop control left to the next iteration's counts launch
    // lock-step batches (iterate_many): a state-only twin of another engine -- matrix, graph and draw table are the
    // parent's, everything it allocates is carved from a slab the parent owns
    nemgpu_engine* parent = nullptr;
    bool carve_all = false;
    bool dry_run = false; size_t dry_bytes = 0;               // dev_alloc only adds up what it would carve
    std::vector<nemgpu_engine*> clones;                      // random starts in lock step: state-only twins of this engine
    char* clone_slab = nullptr; int* clone_flags_host = nullptr; float* clone_par0 = nullptr; size_t clone_bytes = 0;
    bool flags_host_borrowed = false;
    size_t clone_slab_size = 0, clone_flags_size = 0, flags_host_size = 0;   // real sizes of the pooled blocks
    // What the lead engine of a lock-step batch works with: the slab the members' argument blocks go through, the
    // area their flag blocks are gathered in, and the launch sequences captured so far (they hold the slabs'
    // addresses).  It belongs to the device, not to the engine: a lead takes one from the device's pool and hands it
    // back when it is destroyed, so the next group of problems of the same shape REPLAYS the graphs this one captured.
    // desc: everything the captured launches depend on (kernel, variant, members, strides, grids, slab offsets),
    // compared on a key hit -- a 64-bit hash alone would replay the wrong graph on a collision
    struct ZipGraph { uint64_t key; int asked; hipGraphExec_t exec; std::vector<uint64_t> desc; };
    struct ZipContext {
        char* zip_host = nullptr; char* zip_dev = nullptr; size_t zip_cap = 0;
        int* zip_flags_host = nullptr; int* zip_flags_dev = nullptr; size_t zip_flags_cap = 0;
        size_t zip_host_size = 0, zip_dev_size = 0, zip_flags_host_size = 0, zip_flags_dev_size = 0;
        std::vector<ZipGraph> zip_graphs;
    };
    ZipContext* zc = nullptr;
    // a rank alone in the sharded EM (no collective inside a batch): its batches as hipGraphs of the library's own,
    // captured the second time a shape is enqueued (nemgpu_shard_enqueue_batch)
    // (post_*: the host flags the captured body left behind -- a replay launches the graph only, so it sets them itself)
    struct ShardGraph { uint64_t key; std::vector<uint64_t> desc; int asked; hipGraphExec_t exec;
                        bool post_tables_fresh = false, post_density_fresh = false, post_flags_clean = false, post_masks_valid = false; };
    std::vector<ShardGraph> shard_graphs;

    bool ncem() const { return cfg.algo == NEMGPU_ALGO_NCEM; }
    int* ctrl() const { return flags_dev; }
    int* iter_flags() const { return flags_dev + C_WORDS; }
    int* round_flags(int r) const { return flags_dev + C_WORDS + FLAG_ITER_STRIDE + (r % kRoundCap) * FLAG_ROUND_STRIDE; }
    size_t flag_words() const { return C_WORDS + FLAG_ITER_STRIDE + (size_t)kRoundCap * FLAG_ROUND_STRIDE; }   // what the host mirrors
    // k_sweep_fused: the words through which the blocks of a launch meet between rounds (behind the mirrored part;
    // zeroed with the round flags ahead of every sweep)
    size_t flag_alloc_words() const { return flag_words() + 2 * (size_t)kFusedMaxBlocks; }
    unsigned* meet_words() const { return reinterpret_cast<unsigned*>(flags_dev + flag_words()); }
    const int* h_ctrl() const { return flags_host; }
    const int* h_iter() const { return flags_host + C_WORDS; }
    const int* h_round(int r) const { return flags_host + C_WORDS + FLAG_ITER_STRIDE + (r % kRoundCap) * FLAG_ROUND_STRIDE; }
};

// A master pangenome on the device (nemgpu_master_create): what the chunks of PPanGGOLiN's voting loop are formed from
// (nem_chunks.hpp).  One allocation; the stream carries the formation's phase 1.
struct nemgpu_master {
    int device = 0, n = 0, d = 0, wf = 0, nw64 = 0, nnz = 0;
    hipStream_t stream = nullptr;
    char* block = nullptr;
    nemk::MasterDev dev{};
};
// what nemgpu_solve_chunks hands to the builders of nemgpu_solve_many's pipeline instead of host matrices and graphs
struct ChunkSource {
    const nemgpu_master* master = nullptr;
    const nemk::ChunkPlan* plans = nullptr;           // host copies (device pointers inside), one per problem
    const int* nnz = nullptr;                         // directed edges of each chunk's graph
    uint8_t* const* labels = nullptr;                 // per problem: where the NCEM labels go (or null)
};

namespace {

// Every memory operation of an engine is ordered on ITS stream (a non-blocking one) and never touches the legacy
// default stream: several engines may run on different host threads at once (pangenomenem_amd/batch.py), and a
// legacy-stream operation of one thread would collide with a graph capture in progress on another.
thread_local hipStream_t g_alloc_stream = nullptr;      // stream of the engine whose buffers are being allocated
thread_local nemgpu_engine* g_alloc_engine = nullptr;   // ... and the engine itself (owner of the chunks)
void alloc_for(nemgpu_engine* e) { g_alloc_engine = e; g_alloc_stream = e->stream; }

// What an engine needs before it can do anything -- a stream, device memory, the pinned block the loop control is
// copied to -- costs ~2 ms to create and up to 2 ms to release (hipFree waits for the whole device, other engines'
// work included): what a destroyed engine held is kept in a per-device pool and handed to the next engines on that
// device (PPanGGOLiN calls nem() once per chunk of organisms, each call creating and destroying an engine; a
// lock-step batch has tens of engines alive at once).  The pool keeps up to NEM_MI355X_POOL_MB (default 8192) of
// device memory per device; nemgpu_release_cached() frees everything it holds.
struct ResourcePool {
    std::mutex m;
    std::vector<hipStream_t> streams;
    std::vector<hipStream_t> run_streams;                   // highest priority (nemgpu_solve_many's lock-step runs)
    std::multimap<size_t, char*> dev, pinned;
    size_t dev_bytes = 0, pinned_bytes = 0;
    std::vector<nemgpu_engine::ZipContext*> zips;           // lock-step contexts (slabs + captured graphs), most recent last
};
constexpr int kPoolDevices = 64;
constexpr size_t kPoolStreams = 256, kPoolPinnedBytes = (size_t)512 << 20;
ResourcePool* g_pools = new ResourcePool[kPoolDevices];     // (never destroyed: engines may outlive static destructors)
size_t pool_dev_cap()
{
    static const size_t cap = [] {
        const char* g = getenv("NEM_MI355X_POOL_MB");
        return (size_t)(g ? std::max(0ll, atoll(g)) : 8192ll) << 20;
    }();
    return cap;
}
size_t pool_round(size_t bytes)
{
    const size_t g = bytes >= ((size_t)1 << 20) ? ((size_t)1 << 20) : ((size_t)64 << 10);
    return (std::max<size_t>(bytes, 1) + g - 1) / g * g;
}
static std::atomic<long long> g_pool_miss[2], g_pool_spill[2];   // NEM_MI355X_BATCH_PROF: blocks allocated / freed for real (device, pinned)
// a block of at least `bytes` (at most a quarter more) from the pool, else a new one; *got = its real size
hipError_t pool_get(int device, bool pinned, size_t bytes, char** out, size_t* got)
{
    const size_t want = pool_round(bytes);
    if (device >= 0 && device < kPoolDevices) {
        ResourcePool& P = g_pools[device];
        std::lock_guard<std::mutex> lock(P.m);
        auto& M = pinned ? P.pinned : P.dev;
        auto it = M.lower_bound(want);
        if (it != M.end() && it->first <= want + want / 4) {
            *out = it->second; *got = it->first;
            (pinned ? P.pinned_bytes : P.dev_bytes) -= it->first;
            M.erase(it);
            return hipSuccess;
        }
    }
    *got = want;
    g_pool_miss[pinned ? 1 : 0]++;
    hipError_t err = pinned ? hipHostMalloc((void**)out, want) : hipMalloc((void**)out, want);
    if (err != hipSuccess) {                             // out of memory with blocks idle in the pool: give them back
        (void)hipGetLastError();
        nemgpu_release_cached();
        err = pinned ? hipHostMalloc((void**)out, want) : hipMalloc((void**)out, want);
    }
    return err;
}
void pool_put(int device, bool pinned, char* ptr, size_t size)
{
    if (!ptr) return;
    if (device >= 0 && device < kPoolDevices && size > 0) {
        ResourcePool& P = g_pools[device];
        std::lock_guard<std::mutex> lock(P.m);
        size_t& held = pinned ? P.pinned_bytes : P.dev_bytes;
        if (held + size <= (pinned ? kPoolPinnedBytes : pool_dev_cap())) {
            (pinned ? P.pinned : P.dev).emplace(size, ptr);
            held += size;
            return;
        }
    }
    g_pool_spill[pinned ? 1 : 0]++;
    if (pinned) (void)hipHostFree(ptr); else (void)hipFree(ptr);
}
// an engine's blocks in one visit to the pool (a destroyed engine gives back about ten: with 8-16 threads destroying
// the members of a finished group at once, a lock per block was 0.1-0.25 ms per engine in lock hand-overs)
struct PoolBlock { bool pinned; char* ptr; size_t size; };
void pool_put_many(int device, std::vector<PoolBlock>& blocks)
{
    std::vector<PoolBlock> spill;
    if (device >= 0 && device < kPoolDevices) {
        ResourcePool& P = g_pools[device];
        std::lock_guard<std::mutex> lock(P.m);
        for (const PoolBlock& b : blocks) {
            if (!b.ptr) continue;
            size_t& held = b.pinned ? P.pinned_bytes : P.dev_bytes;
            if (b.size > 0 && held + b.size <= (b.pinned ? kPoolPinnedBytes : pool_dev_cap())) {
                (b.pinned ? P.pinned : P.dev).emplace(b.size, b.ptr);
                held += b.size;
            } else spill.push_back(b);
        }
    } else spill = blocks;
    for (const PoolBlock& b : spill) if (b.ptr) { g_pool_spill[b.pinned ? 1 : 0]++; if (b.pinned) (void)hipHostFree(b.ptr); else (void)hipFree(b.ptr); }
    blocks.clear();
}
void release_staging(nemgpu_engine* e);
void zip_context_release(nemgpu_engine* lead);
// a non-blocking stream from the device's pool (creating one costs ~6 ms on this stack, destroying one ~2 ms)
hipError_t pool_stream_get(int device, hipStream_t* out)
{
    *out = nullptr;
    if (device >= 0 && device < kPoolDevices) {
        std::lock_guard<std::mutex> lock(g_pools[device].m);
        if (!g_pools[device].streams.empty()) { *out = g_pools[device].streams.back(); g_pools[device].streams.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
void pool_stream_put(int device, hipStream_t s)             // (idle)
{
    if (!s) return;
    if (device >= 0 && device < kPoolDevices) {
        std::lock_guard<std::mutex> lock(g_pools[device].m);
        if (g_pools[device].streams.size() < kPoolStreams) { g_pools[device].streams.push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}
// The stream of nemgpu_solve_many's lock-step runs: of the highest priority the device offers.  A run is a chain of ~45
// dependent launches; on a plain stream every one of them queued behind whatever the builders' streams had in flight
// (16 MB fills, layout kernels, uploads) and a group took 2-2.4 ms next to the builders against 0.9 ms alone.
hipError_t pool_run_stream_get(int device, hipStream_t* out)
{
    *out = nullptr;
    if (device >= 0 && device < kPoolDevices) {
        std::lock_guard<std::mutex> lock(g_pools[device].m);
        auto& v = g_pools[device].run_streams;
        if (!v.empty()) { *out = v.back(); v.pop_back(); return hipSuccess; }
    }
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest);
}
void pool_run_stream_put(int device, hipStream_t s)         // (idle)
{
    if (!s) return;
    if (device >= 0 && device < kPoolDevices) {
        std::lock_guard<std::mutex> lock(g_pools[device].m);
        if (g_pools[device].run_streams.size() < 16) { g_pools[device].run_streams.push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}
constexpr size_t kStageMax = (size_t)64 << 20;          // larger uploads take the blocking path from the caller's memory
// a pinned block the engine owns until its next release_staging() -- the source of an asynchronous upload
char* stage(nemgpu_engine* e, size_t bytes)
{
    if (bytes > kStageMax) return nullptr;
    if (e->staging.size() >= 32) {                       // (an engine whose inputs are replaced over and over)
        if (hipStreamSynchronize(e->stream) != hipSuccess) return nullptr;
        release_staging(e);
    }
    char* p = nullptr; size_t got = 0;
    if (pool_get(e->device, true, bytes, &p, &got) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    e->staging.push_back({p, got});
    return p;
}
nemgpu_engine::ZipContext* zip_context(nemgpu_engine* lead)
{
    if (lead->zc) return lead->zc;
    if (lead->device >= 0 && lead->device < kPoolDevices) {
        std::lock_guard<std::mutex> lock(g_pools[lead->device].m);
        auto& Z = g_pools[lead->device].zips;
        if (!Z.empty()) { lead->zc = Z.back(); Z.pop_back(); }
    }
    if (!lead->zc) lead->zc = new nemgpu_engine::ZipContext();
    return lead->zc;
}
void zip_context_free(int device, nemgpu_engine::ZipContext* z)
{
    for (auto& g : z->zip_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    pool_put(device, true, z->zip_host, z->zip_host_size);
    pool_put(device, false, z->zip_dev, z->zip_dev_size);
    pool_put(device, true, (char*)z->zip_flags_host, z->zip_flags_host_size);
    pool_put(device, false, (char*)z->zip_flags_dev, z->zip_flags_dev_size);
    delete z;
}
void zip_context_release(nemgpu_engine* lead)           // (the lead's stream is idle)
{
    nemgpu_engine::ZipContext* z = lead->zc;
    if (!z) return;
    lead->zc = nullptr;
    if (lead->device >= 0 && lead->device < kPoolDevices) {
        std::lock_guard<std::mutex> lock(g_pools[lead->device].m);
        if (g_pools[lead->device].zips.size() < 8) { g_pools[lead->device].zips.push_back(z); return; }
    }
    zip_context_free(lead->device, z);
}
void release_staging(nemgpu_engine* e)                  // (the caller has waited for the stream)
{
    for (const nemgpu_engine::Staged& st : e->staging) pool_put(e->device, true, st.p, st.size);
    e->staging.clear();
}

// A forked child inherits a HIP runtime it cannot use (and handles parked by its parent that mean nothing to it).
// PPanGGOLiN's multiprocessing.Pool forks (ppanggolin.py:1039): if the parent has already run nem(), the workers must
// be started with the spawn / forkserver method -- a child of a GPU-using parent fails fast here instead of hanging.
std::atomic<bool> g_hip_used{false}, g_forked_after_hip{false};
std::once_flag g_atfork_once;
void atfork_child()
{
    if (g_hip_used.load()) g_forked_after_hip.store(true);
    g_pools = new ResourcePool[kPoolDevices];               // the parent's handles: leaked on purpose, never touched
}

constexpr size_t kChunkShared = (size_t)16 << 20;       // small buffers share 16 MB chunks
constexpr size_t kChunkOwn = (size_t)4 << 20;           // from 4 MB on a buffer gets a chunk of its own

int chunk_new(nemgpu_engine* e, size_t bytes)
{
    char* base = nullptr;
    size_t got = 0;
    HIPCHK(pool_get(e->device, false, bytes, &base, &got));
    e->chunks.push_back({base, got, 0});
    HIPCHK(hipMemsetAsync(base, 0, got, g_alloc_stream));     // (every later use is on the same stream)
    return NEMGPU_OK;
}

// zeroed device memory that lives until the engine is destroyed
template <typename T>
int dev_alloc(T** p, size_t count)
{
    *p = nullptr;
    nemgpu_engine* e = g_alloc_engine;
    if (e == nullptr) { set_error("dev_alloc outside an engine"); return NEMGPU_E_FUNCARG; }
    if (count == 0) count = 1;
    const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    if (e->dry_run) { e->dry_bytes += bytes; *p = reinterpret_cast<T*>((uintptr_t)256); return NEMGPU_OK; }
    int r;
    if (e->carve_all && e->shared_chunk >= 0 && e->chunks[e->shared_chunk].size - e->chunks[e->shared_chunk].used >= bytes) {
        nemgpu_engine::Chunk& c = e->chunks[e->shared_chunk];
        *p = reinterpret_cast<T*>(c.base + c.used);
        c.used += bytes;
        return NEMGPU_OK;
    }
    if (bytes >= kChunkOwn) {
        if ((r = chunk_new(e, bytes))) return r;
        e->chunks.back().used = bytes;
        *p = reinterpret_cast<T*>(e->chunks.back().base);
        return NEMGPU_OK;
    }
    if (e->shared_chunk < 0 || e->chunks[e->shared_chunk].size - e->chunks[e->shared_chunk].used < bytes) {
        if ((r = chunk_new(e, kChunkShared))) return r;
        e->shared_chunk = (int)e->chunks.size() - 1;
    }
    nemgpu_engine::Chunk& c = e->chunks[e->shared_chunk];
    *p = reinterpret_cast<T*>(c.base + c.used);
    c.used += bytes;
    return NEMGPU_OK;
}

// blocking copy on the engine's stream
hipError_t copy_sync(nemgpu_engine* e, void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
void drop_graphs(nemgpu_engine* e);

// everything an engine owns besides the matrix layouts and the graph: parameters, tables, densities, masks, flags
int alloc_model_buffers(nemgpu_engine* e)
{
    const int k = e->k;
    const size_t kd = (size_t)k * e->d, kdp = (size_t)k * e->dpad;
    int r = NEMGPU_OK;
    auto A = [&](int rr) { if (r == NEMGPU_OK) r = rr; };
    {
        auto a64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
        e->par_o_center = a64((size_t)k); e->par_o_disp = e->par_o_center + a64(kd); e->par_o_nb = e->par_o_disp + a64(kd);
        e->par_words = e->par_o_nb + a64((size_t)k);
        float* pb = nullptr;
        A(dev_alloc(&pb, e->par_words));
        e->prop = pb; e->center = pb + e->par_o_center; e->disp = pb + e->par_o_disp; e->nbobs_k = pb + e->par_o_nb;
        if (e->parent == nullptr) {
            float* p0 = nullptr;
            A(dev_alloc(&p0, e->par_words));
            e->prop0 = p0; e->center0 = p0 + e->par_o_center; e->disp0 = p0 + e->par_o_disp;
        }
    }
    A(dev_alloc(&e->iner, kd));
    A(dev_alloc(&e->tabT, kdp)); A(dev_alloc(&e->tabL0, kdp));
    A(dev_alloc(&e->nz0, (size_t)k * e->W)); A(dev_alloc(&e->nz1, (size_t)k * e->W));
    A(dev_alloc(&e->am0, (size_t)k * e->W)); A(dev_alloc(&e->am1, (size_t)k * e->W));
    A(dev_alloc(&e->ffq, (size_t)k * 256));
    A(dev_alloc(&e->uni, (size_t)k)); A(dev_alloc(&e->nonuni, (size_t)k)); A(dev_alloc(&e->sweep_next, (size_t)32 + kTicketWords));   // [0] sweep number, [32..] last-block ticket counters
    A(dev_alloc(&e->pk, (size_t)k)); A(dev_alloc(&e->logpk, (size_t)k));
    A(dev_alloc(&e->pkfki, (size_t)k * e->npad)); A(dev_alloc(&e->logpkfki, (size_t)k * e->npad));
    A(dev_alloc(&e->mask, (size_t)k * e->nw64));
    A(dev_alloc(&e->stats, (size_t)k + kd));
    A(dev_alloc(&e->flags_dev, e->flag_alloc_words()));
    return r;
}

int ensure_state_buffers(nemgpu_engine* e)
{
    alloc_for(e);
    if (e->ncem()) {
        for (int b = 0; b < 3; b++)
            if (!e->lab[b]) { int r = dev_alloc(&e->lab[b], (size_t)e->n_total); if (r) return r; }
        for (int b = 0; b < 3; b++)
            if (!e->tie_cnt[b]) { int r = dev_alloc(&e->tie_cnt[b], (size_t)e->n / 256 + 2); if (r) return r; }
    } else {
        for (int b = 0; b < 3; b++)
            if (!e->cbuf[b]) { int r = dev_alloc(&e->cbuf[b], (size_t)e->n_total * e->k); if (r) return r; }
        if (!e->fz_in0) {
            size_t kd = (size_t)e->k * e->d;
            int r;
            if ((r = dev_alloc(&e->fz_in0, kd))) return r;
            if ((r = dev_alloc(&e->fz_in1, kd))) return r;
            if ((r = dev_alloc(&e->fz_inh, (size_t)e->k))) return r;
            if ((r = dev_alloc(&e->fz_lastz, kd))) return r;
            if ((r = dev_alloc(&e->fz_any1, kd))) return r;
            if ((r = dev_alloc(&e->fz_ct, (size_t)e->k * ((size_t)(e->n + 1023) / 1024 * 1024)))) return r;   // [k][n up to whole 1024-family windows]
            if ((r = dev_alloc(&e->fz_chk, (size_t)e->k * ((size_t)(e->n + 63) / 64 + 1) * ((size_t)(e->d + 63) / 64 * 64)))) return r;
        }
    }
    return NEMGPU_OK;
}

hipError_t copy_sync(nemgpu_engine* e, void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
    hipError_t err = hipMemcpyAsync(dst, src, bytes, kind, e->stream);
    if (err != hipSuccess) return err;
    return hipStreamSynchronize(e->stream);
}

// the iteration flags are on the host: did a kernel report that it could not finish its work?
int check_fault(nemgpu_engine* e)
{
    if (e->h_iter()[FLAG_FAULT] == 0) return NEMGPU_OK;
    e->fault_seen = true;
    set_error("internal: a kernel reported a fault (a producer/consumer hand-over of the fuzzy M-step never completed); "
              "the results of this call are not to be used");
    return NEMGPU_E_INTERNAL;
}
int clear_fault(nemgpu_engine* e)
{
    if (!e->fault_seen) return NEMGPU_OK;
    e->fault_seen = false;
    e->flags_host[C_WORDS + FLAG_FAULT] = 0;
    HIPCHK(hipMemsetAsync(e->iter_flags() + FLAG_FAULT, 0, sizeof(int), e->stream));
    return NEMGPU_OK;
}

FinishArgs finish_args(nemgpu_engine* e, int mode, const int* stats)
{
    FinishArgs t;
    t.mode = mode;
    t.K = e->k; t.D = e->d; t.dpad = e->dpad; t.n_total = e->n_true; t.disper = e->cfg.disper; t.propor = e->cfg.propor;
    t.stats = stats; t.stats_ranks = 1; t.stats_rank_stride = 0;
    t.prop = e->prop; t.center = e->center; t.disp = e->disp; t.nbobs_k = e->nbobs_k; t.iner = e->iner;
    t.tabT = e->tabT; t.tabL0 = e->tabL0; t.nz0 = e->nz0; t.nz1 = e->nz1;
    t.am0 = e->am0; t.am1 = e->am1; t.uni = e->uni; t.nonuni = e->nonuni;
    t.pk = e->pk; t.logpk = e->logpk; t.flags = e->iter_flags();
    t.stop = e->stop_ptr;
    t.reset_prop = nullptr; t.reset_center = nullptr; t.reset_disp = nullptr;
    t.reset_ctrl = nullptr; t.reset_ctrl_words = 0; t.reset_sweep_next = nullptr;
    t.use_ff = e->use_ff() ? 1 : 0;
    t.perm = e->perm;
    t.ffq = e->ffq;
    return t;
}

// density tables from the current parameters (k_finish mode 0); a no-op when they are up to date
int do_tables(nemgpu_engine* e)
{
    if (e->tables_fresh) return NEMGPU_OK;
    launch_finish(finish_args(e, 0, nullptr), e->stream);
    HIPCHK(hipGetLastError());
    e->tables_fresh = true;
    e->density_fresh = false;
    return NEMGPU_OK;
}

int do_density(nemgpu_engine* e)
{
    launch_density(finish_args(e, 0, nullptr), e->xws, e->n, e->npad, e->pkfki, e->logpkfki, e->iter_flags() + FLAG_MOVED,
                   kSweepFlagWords, e->stream);
    HIPCHK(hipGetLastError());
    e->flags_clean = true;
    e->density_fresh = true;
    return NEMGPU_OK;
}

// One full Gauss-Seidel sweep == relaxation rounds until a round changes nothing.
// sweep_enqueue() launches the first batch of rounds without waiting; sweep_complete() reads the
// flags back (one D2H copy + sync), launches more rounds if the batch did not reach the fixed
// point, and reports whether it had to.  The new partition ends up in buffer (cur+1)%3 and `cur`
// is NOT advanced (the caller commits).
struct SweepCtx {
    SweepArgs a{};
    bool use_nei = false;
    bool multi = false;  // the sweep needs verified relaxation rounds: it reads neighbours, or its ties share a draw stream
    int r = 0;           // rounds launched so far
    int checked = 0;     // rounds whose flags the host has examined
    int slot_base = 0;   // first flag slot of the round window (the blind initial sweep takes a slot of its own)
    // NCEM pipelined loop: fold the iteration's bookkeeping into the last round of the first batch
    bool post = false; bool post_moved = false; CtrlArgs post_ctrl{};
};

int clear_sweep_flags(nemgpu_engine* e)
{
    if (current_recorder()) launch_fill(e->iter_flags() + FLAG_MOVED, kSweepFlagWords, 0, e->stream);
    else HIPCHK(hipMemsetAsync(e->iter_flags() + FLAG_MOVED, 0, kSweepFlagWords * sizeof(int), e->stream));
    e->flags_clean = true;
    return NEMGPU_OK;
}

// ---- TIE_LIBC: the reference's random() stream ------------------------------------------------
// The device reads draws from a table that covers a window of the stream; the host keeps the generator behind it.
// Makes draws [lo, lo + count) available (regenerating / sliding / growing the window as needed).
int ensure_draw_window(nemgpu_engine* e, long lo, long count)
{
    if (e->draw_valid && e->draw_tab0 <= lo && lo + count <= e->draw_tab0 + e->draw_cap) return NEMGPU_OK;
    if (e->draw_borrowed) {                                  // a twin that outgrew its parent's window gets a table of its own
        e->draw_borrowed = false; e->draw_valid = false; e->draw_tab = nullptr; e->draw_cap = 0;
    }
    long cap = std::max<long>(e->draw_cap, 1024);
    while (cap < 2 * count) cap *= 2;
    if (!e->draw_valid || lo < e->draw_tab0) {               // a new stream (seed / restart below the window)
        e->draw_gen.seed(e->cfg.tie_seed); e->draw_gen_pos = 0; e->draw_host.clear(); e->draw_tab0 = 0;
    }
    // slide the host window to start at lo
    const long have_end = e->draw_tab0 + (long)e->draw_host.size();
    if (lo >= have_end) {
        while (e->draw_gen_pos < lo) { (void)e->draw_gen.next(); e->draw_gen_pos++; }
        e->draw_host.clear();
    } else if (lo > e->draw_tab0) {
        e->draw_host.erase(e->draw_host.begin(), e->draw_host.begin() + (lo - e->draw_tab0));
    }
    e->draw_tab0 = lo;
    while ((long)e->draw_host.size() < cap) { e->draw_host.push_back((uint32_t)e->draw_gen.next()); e->draw_gen_pos++; }
    if (cap > e->draw_cap) {
        HIPCHK(hipStreamSynchronize(e->stream));
        alloc_for(e);
        int r = dev_alloc(&e->draw_tab, (size_t)cap);        // (a smaller table stays in its chunk until the engine goes)
        if (r) return r;
        e->draw_cap = (int)cap;
        drop_graphs(e);                                      // table pointer and length are kernel arguments
    }
    HIPCHK(hipMemcpyAsync(e->draw_tab, e->draw_host.data(), (size_t)e->draw_cap * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->draw_valid = true;
    return NEMGPU_OK;
}
long draw_need(const nemgpu_engine* e) { return e->tie_heavy ? 8l * e->n + 1024 : 1024; }

// one draw on the host (RandNemAlgo's MakeRandomPara shares the stream with the ties)
int host_draw(nemgpu_engine* e, uint32_t* out)
{
    int r = ensure_draw_window(e, e->draws, draw_need(e));
    if (r) return r;
    *out = e->draw_host[(size_t)(e->draws - e->draw_tab0)];
    e->draws++;
    return NEMGPU_OK;
}

// the draw-stream side of a sweep's arguments; by_value: the host knows how many draws were made (else the device does)
void sweep_draw_args(nemgpu_engine* e, SweepArgs& a, bool by_value)
{
    a.draw_tab = e->draw_tab; a.draw_tab_len = e->draw_cap;
    a.draw_base = e->draws; a.draw_tab0 = (int)e->draw_tab0;
    a.draw_ctl = by_value ? nullptr : e->draw_ctl;
    a.draw_extra = e->draw_extra_once; e->draw_extra_once = nullptr;
}
// the device words the pipelined loop reads the stream position from (outside any graph capture)
int publish_draw_ctl(nemgpu_engine* e)
{
    if (current_recorder()) {
        launch_fill(e->draw_ctl, 1, e->draws, e->stream);
        launch_fill(e->draw_ctl + 1, 1, (int)e->draw_tab0, e->stream);
        return NEMGPU_OK;
    }
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)e->draw_ctl, e->draws, 1, e->stream));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(e->draw_ctl + 1), (int)e->draw_tab0, 1, e->stream));
    return NEMGPU_OK;
}

// May the first rounds of a sweep go out as ONE launch whose blocks meet between rounds (k_sweep_fused)?  NCEM without
// the libc tie stream, the whole problem on this engine, issued for real (a lock-step batch zips the classic rounds), a
// kernel instance for K, and a grid that is resident as a whole: at most one block per CU.
bool sweep_fused_ok(const nemgpu_engine* e, bool use_nei)
{
    return e->fused_sweep && use_nei && e->ncem() && !e->libc() && e->sh_world == 1 && e->sh_stride == 0 && e->lo == 0 &&
           e->n == e->n_total && e->parent == nullptr && current_recorder() == nullptr && sweep_fused_has_instance(e->k) &&
           sweep_grid_blocks(e->n, e->k) <= kFusedMaxBlocks;
}
// rounds of a sweep's first batch: what the caller wants, or the fused launch's rounds (inside the pipelined loop at
// most the four slots its loop control reads)
int sweep_first_rounds(const nemgpu_engine* e, float beta, int wanted, int pipelined = -1)
{
    if (!sweep_fused_ok(e, e->has_graph && beta != 0.0f)) return wanted;
    if (pipelined < 0) pipelined = e->stop_ptr != nullptr ? 1 : 0;
    return std::max(wanted, pipelined ? std::min(e->fused_rounds, 4) : e->fused_rounds);
}
// SweepArgs::exp_tab for the configured beta (outside any capture / recording: loop_begin, init_partition)
int ensure_exp_table(nemgpu_engine* e)
{
    if (current_recorder() != nullptr || !e->ncem() || !e->has_graph || e->cfg.beta == 0.0f || e->parent != nullptr) return NEMGPU_OK;
    if (e->exp_need <= 64) return NEMGPU_OK;              // (small weights: every block computes its 64 entries itself)
    // (measured neutral, round 4 -- profiles/r04_exp_table_ab.json: adjacency weights, one engine 0.03450 vs 0.03469 ms per
    //  iteration, 64 problems in lock step 5.02 vs 4.98 us per problem-iteration: the exponentials are not what bounds a
    //  round, alone or in a batch -- so the table is opt-in, NEM_MI355X_EXP_TABLE=1; the fused sweep uses it when both are on)
    static const bool on = getenv("NEM_MI355X_EXP_TABLE") && getenv("NEM_MI355X_EXP_TABLE")[0] == '1';
    if (!on) return NEMGPU_OK;
    if (e->exp_ready && e->exp_beta == e->cfg.beta) return NEMGPU_OK;
    if (e->exp_tab == nullptr) { alloc_for(e); int r = dev_alloc(&e->exp_tab, (size_t)kExpTabGlobal); if (r) return r; }
    launch_exp_table(e->cfg.beta, e->exp_tab, kExpTabGlobal, e->stream);
    HIPCHK(hipGetLastError());
    e->exp_beta = e->cfg.beta; e->exp_ready = true;
    return NEMGPU_OK;
}

int sweep_launch_rounds(nemgpu_engine* e, SweepCtx& c, int count)
{
    const bool ncem = e->ncem();
    const int P = e->cur, Q = (e->cur + 1) % 3, R = (e->cur + 2) % 3;
    const int r0 = c.r;
    c.a.fused_rounds = 0;
    if (r0 == 0 && count >= 2 && c.multi && sweep_fused_ok(e, c.use_nei) && c.slot_base + count <= kRoundCap) {
        // the sweep's first `count` rounds in one launch: round t writes flag slot t and buffer Q (t even) / R (t odd),
        // as `count` launches would -- whoever reads the flags afterwards goes on from there
        count = std::min(count, kFusedMaxRounds);
        SweepArgs& a = c.a;
        a.lab_old = e->lab[P]; a.lab_guess = e->lab[P]; a.lab_out = e->lab[Q]; a.lab_out2 = e->lab[R];
        a.tie_cnt_guess = e->tie_cnt[P]; a.tie_cnt_out = e->tie_cnt[Q];
        a.flags = e->round_flags(c.slot_base);
        a.bar = e->meet_words();
        a.fused_rounds = count;
        a.exp_tab = (e->exp_ready && e->exp_beta == a.beta) ? e->exp_tab : nullptr; a.exp_tab_len = e->exp_need;
        a.fold_ticket = e->sweep_next + 32;
        a.prev_changed = nullptr;
        a.stop = e->stop_ptr;
        a.post_on = 0;
        if (c.post) {
            a.post_on = 1; a.post_from_guess = 0; a.post_moved = c.post_moved ? 1 : 0; a.post_skip_guess = 0;
            a.post_nw64 = e->nw64; a.post_mask = e->mask; a.post_flags = e->iter_flags(); a.post_ctrl = c.post_ctrl;
        }
        launch_sweep(a, true, e->stream);
        HIPCHK(hipGetLastError());
        e->n_fused++;
        c.r += count;
        a.fused_rounds = 0;
        return NEMGPU_OK;
    }
    for (int b = 0; b < count; b++, c.r++) {
        const int r = c.r;
        const int gb = (r == 0) ? P : ((r - 1) % 2 == 0 ? Q : R);
        const int ob = (r % 2 == 0) ? Q : R;
        if (ncem) {
            c.a.lab_old = e->lab[P]; c.a.lab_guess = e->lab[gb]; c.a.lab_out = e->lab[ob];
            c.a.tie_cnt_guess = e->tie_cnt[gb]; c.a.tie_cnt_out = e->tie_cnt[ob];
        } else { c.a.c_old = e->cbuf[P]; c.a.c_guess = e->cbuf[gb]; c.a.c_out = e->cbuf[ob]; }
        c.a.flags = e->round_flags(c.slot_base + r);
        c.a.fold_ticket = e->sweep_next + 32;
        c.a.prev_changed = (r == r0) ? nullptr : (e->round_flags(c.slot_base + r - 1) + FLAG_CHANGED);
        c.a.stop = e->stop_ptr;
        c.a.post_on = 0;
        if (c.post && ncem && r0 == 0 && b == count - 1) {
            // (the sweep's final labels are in buffer Q, the out buffer of even rounds)
            c.a.post_on = 1; c.a.post_from_guess = (r % 2 == 1) ? 1 : 0; c.a.post_moved = c.post_moved ? 1 : 0;
            c.a.post_skip_guess = c.a.post_from_guess;
            c.a.post_nw64 = e->nw64; c.a.post_mask = e->mask; c.a.post_flags = e->iter_flags(); c.a.post_ctrl = c.post_ctrl;
        }
        launch_sweep(c.a, ncem, e->stream);
    }
    HIPCHK(hipGetLastError());
    return NEMGPU_OK;
}

// the arguments every round of one sweep shares (takes the sweep's number)
int sweep_setup(nemgpu_engine* e, float beta, SweepCtx& c, bool id_by_value)
{
    c.use_nei = e->has_graph && beta != 0.0f;
    c.multi = c.use_nei || e->libc();
    SweepArgs& a = c.a;
    a.n_local = e->n; a.lo = e->lo; a.n_total = e->n_total; a.K = e->k; a.npad = e->npad;
    a.use_nei = c.use_nei ? 1 : 0;
    a.nei_ptr = e->nei_ptr; a.nei_idx = e->nei_idx; a.nei_w = e->nei_w;
    a.beta = beta;
    a.pkfki = e->pkfki;
    a.tie_rule = e->cfg.tie_rule; a.tie_seed = e->cfg.tie_seed; a.sweep_id = e->sweep_counter++;
    a.sweep_id_ptr = (e->stop_ptr != nullptr && !id_by_value) ? e->sweep_next : nullptr;   // pipelined loop: the device keeps count
    if (e->libc()) {
        if (e->stop_ptr == nullptr) { int r = ensure_draw_window(e, e->draws, draw_need(e)); if (r) return r; }
        sweep_draw_args(e, a, e->stop_ptr == nullptr);
    }
    // exp(beta * context) from the per-beta table where the graph's weights reach beyond the 64 entries a block computes
    // itself (sweep_body); the table was made by ensure_exp_table, ahead of any capture or recording
    a.exp_tab = (c.use_nei && e->ncem() && e->exp_ready && e->exp_beta == beta) ? e->exp_tab : nullptr;
    a.exp_tab_len = e->exp_need;
    return NEMGPU_OK;
}

int sweep_enqueue(nemgpu_engine* e, float beta, SweepCtx& c, bool id_by_value = false, const CtrlArgs* post_ctrl = nullptr,
                  bool post_moved = false, int slot_base = 0, int rounds = 0)
{
    c = SweepCtx();
    c.slot_base = slot_base;
    if (post_ctrl != nullptr && e->ncem()) { c.post = true; c.post_moved = post_moved; c.post_ctrl = *post_ctrl; }
    { int r = sweep_setup(e, beta, c, id_by_value); if (r) return r; }
    if (!e->flags_clean) { int r = clear_sweep_flags(e); if (r) return r; }
    e->flags_clean = false;
    return sweep_launch_rounds(e, c, c.multi ? sweep_first_rounds(e, beta, rounds > 0 ? rounds : e->round_batch) : 1);
}

// Did a fused launch among the rounds [from, to) of this sweep fail to meet (kFusedFailed)?  Then its rounds are void.
bool fused_failed_seen(const nemgpu_engine* e, int slot_base, int from, int to)
{
    for (int q = from; q < to; q++) if (e->h_round(slot_base + q)[FLAG_CHANGED] & kFusedFailed) return true;
    return false;
}
// ... the sweep starts again from round 0, one launch per round from now on (the old partition and the densities are
// untouched; the flag window is cleared)
int fused_fallback(nemgpu_engine* e, SweepCtx& c)
{
    e->fused_sweep = false;
    e->n_fused_failed++;
    drop_graphs(e);                                      // (they hold fused launches)
    HIPCHK(hipMemsetAsync(e->round_flags(0), 0, (kRoundCap * FLAG_ROUND_STRIDE + kMeetWords) * sizeof(int), e->stream));
    c.r = 0; c.checked = 0;
    return NEMGPU_OK;
}

// `extra` is set when rounds beyond the first batch were needed (work enqueued after the first
// batch read a partition that was not final yet and must be redone by the caller).
// `grow`: every further batch of rounds is as long as all the rounds before it (up to 16) instead of round_batch -- for
// sweeps whose need is long-tailed (the initial sweeps of a random start that ties at thousands of families).
int sweep_complete(nemgpu_engine* e, SweepCtx& c, int* rounds_out, bool* extra, bool flags_ready = false, bool grow = false)
{
    int done_at = -1;
    if (extra) *extra = false;
    for (;;) {
        if (!flags_ready) {                                  // (a lock-step batch has fetched every member's flags already)
            HIPCHK(hipMemcpyAsync(e->flags_host, e->flags_dev, e->flag_words() * sizeof(int), hipMemcpyDeviceToHost,
                                  e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
        }
        flags_ready = false;
        { const int fr = check_fault(e); if (fr) return fr; }
        if (!c.multi) { done_at = 0; break; }
        bool tab_short = false;
        if (fused_failed_seen(e, c.slot_base, c.checked, c.r)) {
            int rr = fused_fallback(e, c);
            if (rr) return rr;
            if (extra) *extra = true;
            if ((rr = sweep_launch_rounds(e, c, e->round_batch))) return rr;
            continue;
        }
        for (int q = c.checked; q < c.r; q++) {
            if (e->h_round(c.slot_base + q)[FLAG_CHANGED] == 0) { done_at = q; break; }
            if (e->h_round(c.slot_base + q)[FLAG_NTIES] & (1 << 30)) tab_short = true;
        }
        c.checked = c.r;
        if (done_at >= 0) break;
        if (extra) *extra = true;
        if (tab_short) {
            // a site needed a draw beyond the table: those rounds were void (they flagged a change).  From now on
            // the window covers whatever a batch of sweeps can draw.
            e->tie_heavy = true;
            int rr = ensure_draw_window(e, e->draws, draw_need(e));
            if (rr) return rr;
            sweep_draw_args(e, c.a, true);
        }
        const int more = grow ? std::min(16, std::max(e->round_batch, c.r % kRoundCap)) : e->round_batch;
        if (c.r % kRoundCap == 0 || c.r % kRoundCap + more > kRoundCap) {
            // the flag window is about to wrap: every earlier round has been examined, start a clean window
            // (keeps the parity of r, which selects the ping-pong buffers)
            HIPCHK(hipMemsetAsync(e->round_flags(0), 0, (kRoundCap * FLAG_ROUND_STRIDE + kMeetWords) * sizeof(int), e->stream));
            while (c.r % kRoundCap != 0) c.r += 2;       // skip to the window start, same parity
            c.checked = c.r;
        }
        int rr = sweep_launch_rounds(e, c, more);
        if (rr) return rr;
    }
    // the round that changed nothing recomputed every site: its zero-density tally is the sweep's
    const int* f = e->h_round(c.slot_base + done_at);
    if (e->libc()) e->draws += f[FLAG_NTIES] & ((1 << 30) - 1);
    if (f[FLAG_NZERO] > 0) {
        e->zero_density += f[FLAG_NZERO];
        if (e->first_zero < 0) e->first_zero = e->n_total - f[FLAG_FIRSTZERO];
    }
    // result: the out buffer of round done_at; when that is R it equals Q bit for bit (see k_sweep)
    e->sweep_rounds += done_at + 1;
    if (rounds_out) *rounds_out = done_at + 1;
    return NEMGPU_OK;
}

int do_sweep(nemgpu_engine* e, float beta, int* rounds_out)
{
    SweepCtx c;
    int r = sweep_enqueue(e, beta, c);
    if (r) return r;
    return sweep_complete(e, c, rounds_out, nullptr);
}

int do_labels_post(nemgpu_engine* e, int newbuf, int oldbuf, const CtrlArgs* ctrl = nullptr)
{
    launch_labels_post(e->n, e->lo, e->k, e->nw64, e->lab[newbuf], oldbuf >= 0 ? e->lab[oldbuf] : nullptr, e->mask,
                       e->iter_flags(), e->stop_ptr, ctrl, e->stream);
    HIPCHK(hipGetLastError());
    e->masks_valid = true;
    return NEMGPU_OK;
}

// EstimPara (nem_mod.c:415-469) on the current partition; leaves FLAG_EMPTYK in the iteration flags.
int do_mstep(nemgpu_engine* e, const CtrlArgs* prev_ctrl = nullptr)
{
    if (e->ncem()) {
        if (!e->masks_valid) { int r = do_labels_post(e, e->cur, -1); if (r) return r; }
        launch_mstep_counts(e->k, e->d, e->nw64, e->xt, e->mask, e->stats, e->stop_ptr, prev_ctrl, e->stream);
        launch_finish(finish_args(e, 1, e->stats), e->stream);
    } else {
        launch_mstep_fuzzy(e->n, e->npad, e->k, e->d, e->xw, e->xt, e->nw64, e->cbuf[e->cur] + (size_t)e->lo * e->k,
                           e->fuzzy_chains ? e->fz_ct : nullptr, e->nbobs_k,
                           e->fz_in0, e->fz_in1, e->fz_inh, e->fz_lastz, e->fz_any1, e->center, e->iner, e->stop_ptr, e->stream,
                           e->fuzzy_chains == 2 ? e->fz_chk : nullptr, e->iter_flags() + FLAG_FAULT, e->fault_inject);
        launch_finish(finish_args(e, 2, nullptr), e->stream);
    }
    HIPCHK(hipGetLastError());
    e->tables_fresh = true;                                        // k_finish rebuilt them from the new parameters
    e->density_fresh = false;
    return NEMGPU_OK;
}

int read_iter_flags(nemgpu_engine* e)
{
    HIPCHK(hipMemcpyAsync(e->flags_host, e->flags_dev, (C_WORDS + FLAG_ITER_STRIDE) * sizeof(int),
                          hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return check_fault(e);
}

int reset_device(nemgpu_engine* e);
int reset_state(nemgpu_engine* e, bool lazy = false);
int flush_reset(nemgpu_engine* e);
void drop_graphs(nemgpu_engine* e);
int criteria_enqueue(nemgpu_engine* e, int buf);
void fill_result(nemgpu_engine* e, nemgpu_result* res);

// ComputePartitionFromPara(Needinit = 1), nem_alg.c:1967-1981
int init_partition(nemgpu_engine* e)
{
    int r;
    if (!e->have_matrix || !e->have_params) { set_error("matrix and parameters must be set first"); return NEMGPU_E_FUNCARG; }
    if ((r = flush_reset(e))) return r;
    if ((r = ensure_state_buffers(e))) return r;
    if ((r = ensure_exp_table(e))) return r;
    if ((r = do_tables(e))) return r;
    if ((r = do_density(e))) return r;
    // ClassifM starts as zeros (calloc, nem_exe.c:524-526): the blind beta = 0 sweep never reads it
    if ((r = do_sweep(e, 0.0f, nullptr))) return r;                // blind sweep: one round, no neighbour reads
    e->cur = (e->cur + 1) % 3;
    if ((r = do_sweep(e, e->cfg.beta, nullptr))) return r;
    e->cur = (e->cur + 1) % 3;
    e->masks_valid = false;
    if (e->ncem()) { if ((r = do_labels_post(e, e->cur, -1))) return r; }
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)e->sweep_next, (int)e->sweep_counter, 1, e->stream));
    return NEMGPU_OK;
}

// NemAlgo's loop body (nem_alg.c:1789-1840), up to n_iters iterations, as a device-controlled
// pipeline: the host enqueues up to kPipeDepth whole iterations back to back; k_ctrl (one thread, at
// the end of each iteration) applies the loop tests on the device and raises ctrl[C_STOP]; every loop
// kernel returns at once when it is set.  The host synchronises once per batch.  The bookkeeping of
// an iteration (label masks + "moved" flag, or the fuzzy convergence test) is enqueued right behind
// the first two relaxation rounds; if a sweep needs more rounds than that, the pipeline stops and the
// host finishes that iteration round by round (rare: labels are sticky).
// `ctrl` (pipelined loop only): the loop tests run in the last block of the bookkeeping kernel
int post_sweep(nemgpu_engine* e, int newbuf, int oldbuf, const CtrlArgs* ctrl = nullptr)
{
    if (e->ncem()) return do_labels_post(e, newbuf, oldbuf, ctrl);
    if (e->cfg.cvtest == NEMGPU_CV_CLAS) {
        launch_conv_fuzzy((size_t)e->n * e->k, e->cbuf[newbuf] + (size_t)e->lo * e->k,
                          e->cbuf[oldbuf] + (size_t)e->lo * e->k, e->cfg.cvthres, e->iter_flags(), e->stop_ptr, ctrl,
                          e->stream);
        HIPCHK(hipGetLastError());
    } else if (ctrl != nullptr) {
        launch_ctrl(*ctrl, e->stream);
        HIPCHK(hipGetLastError());
    }
    return NEMGPU_OK;
}

constexpr int kPipeDepth = 7;       // (the graph table holds batches of up to 7 iterations)

// A sweep of the pipelined loop whose two enqueued rounds did not reach the fixed point (or, TIE_LIBC, ran out of
// the draw table): the context from which the host goes on with rounds 2, 3, ... (current partition = e->cur)
int host_rounds_ctx(nemgpu_engine* e, SweepCtx& sc, uint32_t sweep_id, int launched)
{
    sc = SweepCtx();
    sc.use_nei = e->has_graph && e->cfg.beta != 0.0f;
    sc.multi = true;
    SweepArgs& a = sc.a;
    a.n_local = e->n; a.lo = e->lo; a.n_total = e->n_total; a.K = e->k; a.npad = e->npad; a.use_nei = sc.use_nei ? 1 : 0;
    a.nei_ptr = e->nei_ptr; a.nei_idx = e->nei_idx; a.nei_w = e->nei_w; a.beta = e->cfg.beta; a.pkfki = e->pkfki;
    a.tie_rule = e->cfg.tie_rule; a.tie_seed = e->cfg.tie_seed; a.sweep_id = sweep_id; a.sweep_id_ptr = nullptr;
    a.exp_tab = (sc.use_nei && e->ncem() && e->exp_ready && e->exp_beta == a.beta) ? e->exp_tab : nullptr;
    a.exp_tab_len = e->exp_need;
    sc.r = launched; sc.checked = launched;               // (the enqueued rounds all changed something)
    if (e->libc()) {
        for (int q = 0; q < launched; q++) if (e->h_round(q)[FLAG_NTIES] & (1 << 30)) e->tie_heavy = true;
        int r = ensure_draw_window(e, e->draws, draw_need(e));
        if (r) return r;
        sweep_draw_args(e, a, true);
    }
    return NEMGPU_OK;
}

// enqueue one whole iteration whose current partition is buffer `cur`.  defer_ctrl: another iteration follows in
// the same batch -- an NCEM iteration's loop control then runs in that iteration's counts launch (k_mstep_counts)
// instead of in a last-block ticket at the tail of the last sweep round.
// rounds enqueued for the sweep of iteration `pos` of a run (pos < 0: not known)
int iteration_rounds(const nemgpu_engine* e, int pos, bool deep)
{
    // (fuzzy sweeps need their third round nearly always -- memberships keep moving in the last digits -- : round_batch
    //  everywhere, and what the host had to add for the leading iterations of a run)
    int want = (current_recorder() != nullptr || deep || !e->ncem()) ? e->round_batch : e->rounds_iter;
    if (!e->ncem() && current_recorder() == nullptr && pos >= 0 && pos < kFzPositions && e->fz_need[pos] > want) want = e->fz_need[pos];
    return want;
}

int enqueue_iteration(nemgpu_engine* e, int cur, uint32_t sweep_id, bool defer_ctrl, bool deep, int pos = -1)
{
    int r;
    const int saved = e->cur;
    e->cur = cur;
    // one engine alone: the parameter update rides in the density launch (a launch boundary costs more than the
    // redundant per-block derivation).  In a lock-step batch the launch is shared by all members and the kernels are
    // bound by instruction issue: there the update runs once per class (k_finish) and the density blocks only load it
    const bool fused = !e->cfg.param_fix && e->ncem() && e->fused_update() && current_recorder() == nullptr;
    const bool counts_first = !e->cfg.param_fix && e->ncem();      // the iteration starts with k_mstep_counts
    if (e->ctrl_pending && !counts_first) {                        // (not reached: the mode is fixed within a batch)
        launch_ctrl(e->ctrl_deferred, e->stream);
        e->ctrl_pending = false;
    }
    if (fused) {
        // M-step counts, then ONE kernel: parameter update (per block, from the counts) + density
        if (!e->masks_valid) { if ((r = do_labels_post(e, e->cur, -1))) { e->cur = saved; return r; } }
        launch_mstep_counts(e->k, e->d, e->nw64, e->xt, e->mask, e->stats, e->stop_ptr,
                            e->ctrl_pending ? &e->ctrl_deferred : nullptr, e->stream);
        e->ctrl_pending = false;
        launch_density_fused(finish_args(e, 1, e->stats), e->xws, e->n, e->npad, e->pkfki, e->logpkfki,
                             e->iter_flags() + FLAG_MOVED, kSweepFlagWords, e->stream);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess) { e->cur = saved; set_error(std::string("launch failed: ") + hipGetErrorString(le)); return NEMGPU_E_DEVICE; }
        e->flags_clean = true;
        e->tables_fresh = false;                                   // the table buffers were not rebuilt
        e->density_fresh = true;
    } else {
        if (!e->cfg.param_fix) {                                   // nem_alg.c:1806
            if ((r = do_mstep(e, e->ctrl_pending ? &e->ctrl_deferred : nullptr))) { e->cur = saved; return r; }
            e->ctrl_pending = false;
            if ((r = do_tables(e))) { e->cur = saved; return r; }
        }
        if ((r = do_density(e))) { e->cur = saved; return r; }
    }
    SweepCtx c;
    e->sweep_counter = sweep_id;
    CtrlArgs ca{};
    // (members of a lock-step batch all take the same number of rounds: a member with a sequence of its own would
    //  need launches of its own)
    const int it_rounds = sweep_first_rounds(e, e->cfg.beta, iteration_rounds(e, pos, deep));
    ca.ctrl = e->ctrl(); ca.iter_flags = e->iter_flags(); ca.round0 = e->round_flags(0); ca.n_rounds = it_rounds;
    ca.param_fix = e->cfg.param_fix; ca.use_nei = ((e->has_graph && e->cfg.beta != 0.0f) || e->libc()) ? 1 : 0; ca.cvtest = e->cfg.cvtest;
    ca.ncem = e->ncem() ? 1 : 0; ca.cvthres = e->cfg.cvthres; ca.sweep_next = e->sweep_next; ca.ticket = e->sweep_next + 32;
    ca.draw_ctl = e->libc() ? e->draw_ctl : nullptr;
    // NCEM: the bookkeeping (masks, "moved", loop tests) rides in the last relaxation round's launch -- the loop
    // tests in the next iteration's counts launch when there is one
    const bool defer = defer_ctrl && counts_first;
    CtrlArgs none{};
    if ((r = sweep_enqueue(e, e->cfg.beta, c, false, e->ncem() ? (defer ? &none : &ca) : nullptr, true, 0, it_rounds))) { e->cur = saved; return r; }
    if (defer) { e->ctrl_deferred = ca; e->ctrl_pending = true; }
    if (!e->ncem()) { if ((r = post_sweep(e, (cur + 1) % 3, cur, &ca))) { e->cur = saved; return r; } }
    else e->masks_valid = true;
    e->cur = saved;
    return NEMGPU_OK;
}

// the restart + the two initial sweeps as the head of a pipelined batch (buffers 0 -> 1 -> 2)
// defer_ctrl: an iteration follows in the same batch (see enqueue_iteration)
int libc_init_usual(const nemgpu_engine* e, int which, int percent);
// TIE_LIBC: is a run's start part of its first pipelined batch (verified on the device, see ctrl_logic), or completed
// from the host before it (NEM_MI355X_LIBC_INIT=host)?
static bool libc_init_pipelined()
{
    const char* v = getenv("NEM_MI355X_LIBC_INIT");          // (read every time: a switch for tests and A/B runs)
    return !(v && strcmp(v, "host") == 0);
}

// relaxation rounds a pipelined start enqueues for its beta sweep (enqueue_init; batch_finish continues from there)
static int init_beta_rounds(const nemgpu_engine* e, int pipelined = -1)
{
    int n = sweep_first_rounds(e, e->cfg.beta, std::max<int>(e->round_batch, (!e->ncem() && current_recorder() == nullptr) ? e->fz_init_need : 0), pipelined);
    if (e->libc()) n = sweep_first_rounds(e, e->cfg.beta, e->libc_rb, pipelined);      // (what 95 % of the starts so far needed: batch_plan)
    else if (e->ncem() && current_recorder() == nullptr && e->init_rounds_ncem >= 2) n = sweep_first_rounds(e, e->cfg.beta, e->init_rounds_ncem, pipelined);
    return n;
}

int enqueue_init(nemgpu_engine* e, bool defer_ctrl)
{
    int r;
    // one launch: initial parameters back in place, loop control cleared, density tables built
    {
        FinishArgs t = finish_args(e, 0, nullptr);
        t.reset_prop = e->prop0; t.reset_center = e->center0; t.reset_disp = e->disp0;
        t.reset_ctrl = e->ctrl(); t.reset_ctrl_words = C_WORDS; t.reset_sweep_next = e->sweep_next;
        launch_finish(t, e->stream);
        HIPCHK(hipGetLastError());
    }
    e->tables_fresh = true; e->density_fresh = false;
    e->cur = 0;
    if ((r = do_density(e))) return r;                             // (also clears every sweep flag slot)
    SweepCtx c0, c1;
    e->sweep_counter = 0;
    // blind sweep: one round, 0 -> 1, on a flag slot of its own so that no clear is needed before the next sweep.
    // TIE_LIBC: its sites are coupled through the draw counter -- `ra` verified rounds on the window's last slots; the
    // beta sweep behind it takes the blind sweep's draw count from the last of those slots on the device (the rounds
    // behind a fixed point carry it along), the loop control checks both sweeps and books their draws (ctrl_logic).
    const bool libc = e->libc();
    const int ra = libc ? e->libc_ra : 1;
    if ((r = sweep_enqueue(e, 0.0f, c0, true, nullptr, false, kRoundCap - ra, libc ? ra : 0))) return r;
    e->flags_clean = true;
    e->cur = 1;
    CtrlArgs ca{};
    ca.ctrl = e->ctrl(); ca.iter_flags = e->iter_flags(); ca.round0 = e->round_flags(0); ca.n_rounds = init_beta_rounds(e);
    ca.param_fix = e->cfg.param_fix; ca.use_nei = ((e->has_graph && e->cfg.beta != 0.0f) || e->libc()) ? 1 : 0; ca.cvtest = e->cfg.cvtest;
    ca.ncem = e->ncem() ? 1 : 0; ca.cvthres = e->cfg.cvthres; ca.sweep_next = e->sweep_next; ca.ticket = e->sweep_next + 32;
    ca.draw_ctl = e->libc() ? e->draw_ctl : nullptr;
    ca.is_init = 1;
    ca.blind = e->round_flags(kRoundCap - ra);
    ca.blind_rounds = libc ? ra : 0;
    const bool defer = defer_ctrl && e->ncem() && !e->cfg.param_fix;
    CtrlArgs none{};
    if (libc) e->draw_extra_once = e->round_flags(kRoundCap - 1) + FLAG_NTIES;
    if ((r = sweep_enqueue(e, e->cfg.beta, c1, true, e->ncem() ? (defer ? &none : &ca) : nullptr, false, 0, ca.n_rounds))) return r;   // 1 -> 2 (and 0 as the pong buffer)
    e->draw_extra_once = nullptr;
    if (defer) { e->ctrl_deferred = ca; e->ctrl_pending = true; }
    if (e->ncem()) e->masks_valid = true;
    else { launch_ctrl(ca, e->stream); HIPCHK(hipGetLastError()); }
    e->cur = 2;
    return NEMGPU_OK;
}

// ---- TIE_LIBC start: the two initial sweeps draw from ONE stream, the blind one first, so each is completed (its
// rounds verified from the host, its draws booked) before the next is enqueued.  Three enqueue steps with a host step
// after the first two; recordable like everything else.
int libc_init_a(nemgpu_engine* e, SweepCtx& c, int rounds = 0)
{
    int r;
    FinishArgs t = finish_args(e, 0, nullptr);                 // initial parameters back in place, loop control cleared, tables
    t.reset_prop = e->prop0; t.reset_center = e->center0; t.reset_disp = e->disp0;
    t.reset_ctrl = e->ctrl(); t.reset_ctrl_words = C_WORDS; t.reset_sweep_next = e->sweep_next;
    launch_finish(t, e->stream);
    e->tables_fresh = true; e->density_fresh = false;
    e->cur = 0; e->sweep_counter = 0;
    if ((r = do_density(e))) return r;
    return sweep_enqueue(e, 0.0f, c, false, nullptr, false, 0, rounds);          // blind sweep 0 -> 1
}
int libc_init_b(nemgpu_engine* e, SweepCtx& c, int rounds = 0)
{
    e->cur = 1;
    return sweep_enqueue(e, e->cfg.beta, c, false, nullptr, false, 0, rounds);   // 1 -> 2
}
int libc_init_c(nemgpu_engine* e)
{
    int r;
    e->cur = 2;
    e->masks_valid = false;
    if ((r = do_labels_post(e, 2, -1))) return r;
    launch_fill(e->sweep_next, 1, (int)e->sweep_counter, e->stream);
    return NEMGPU_OK;
}

// The same start with ONE wait: the beta sweep's rounds go out behind the blind sweep's, taking the blind sweep's draw
// count from the flag slot of its last round on the device (rounds behind a sweep's fixed point carry the count along),
// and the class masks behind them.  `ra` blind rounds on the last slots of the flag window, `rb` beta rounds from slot 0.
// Then the host looks once: blind sweep at its fixed point within ra rounds?  (else what ran behind it is void:
// *redo = true, nothing booked, the caller starts over with libc_init_a)  beta sweep within rb?  (else it is continued
// from the host as any sweep is, and the masks are made again).  rounds[0..1]: what the two sweeps needed.
int libc_init_one_wait(nemgpu_engine* e, int ra, int rb, bool* redo, int rounds[2])
{
    int r;
    *redo = false;
    ra = std::max(1, std::min(ra, kRoundCap / 4)); rb = std::max(1, std::min(rb, kRoundCap / 2));
    FinishArgs t = finish_args(e, 0, nullptr);
    t.reset_prop = e->prop0; t.reset_center = e->center0; t.reset_disp = e->disp0;
    t.reset_ctrl = e->ctrl(); t.reset_ctrl_words = C_WORDS; t.reset_sweep_next = e->sweep_next;
    launch_finish(t, e->stream);
    e->tables_fresh = true; e->density_fresh = false;
    e->cur = 0; e->sweep_counter = 0;
    if ((r = do_density(e))) return r;                             // (also clears every sweep flag slot)
    SweepCtx ca, cb;
    const int base_a = kRoundCap - ra;
    if ((r = sweep_enqueue(e, 0.0f, ca, false, nullptr, false, base_a, ra))) return r;          // blind sweep 0 -> 1
    e->flags_clean = true;                                         // (the beta sweep's slots have not been touched)
    e->cur = 1;
    e->draw_extra_once = e->round_flags(kRoundCap - 1) + FLAG_NTIES;
    if ((r = sweep_enqueue(e, e->cfg.beta, cb, false, nullptr, false, 0, rb))) return r;         // 1 -> 2
    e->draw_extra_once = nullptr;
    // the flags leave as soon as the rounds are through (an event marks the copy); the class masks and the sweep counter
    // run while the host wakes up and draws the next start's centres
    HIPCHK(hipMemcpyAsync(e->flags_host, e->flags_dev, e->flag_words() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    nemgpu_engine* ev_owner = e->parent ? e->parent : e;     // (the twins of a random-start round share their parent's event: one start at a time)
    if (!ev_owner->ev_flags) HIPCHK(hipEventCreateWithFlags(&ev_owner->ev_flags, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev_owner->ev_flags, e->stream));
    e->cur = 2;
    e->masks_valid = false;
    if ((r = do_labels_post(e, 2, -1))) return r;
    launch_fill(e->sweep_next, 1, (int)e->sweep_counter, e->stream);
    // (polled: the wait is tens of microseconds and its end is on the critical path of every start; a blocking wait's
    //  wake-up cost 5-10 us more)
    for (;;) {
        const hipError_t q = hipEventQuery(ev_owner->ev_flags);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) { HIPCHK(q); }
        __builtin_ia32_pause();
    }
    if ((r = check_fault(e))) return r;
    auto book = [&](const int* f, int done_at) {
        e->draws += f[FLAG_NTIES] & ((1 << 30) - 1);
        if (f[FLAG_NZERO] > 0) { e->zero_density += f[FLAG_NZERO]; if (e->first_zero < 0) e->first_zero = e->n_total - f[FLAG_FIRSTZERO]; }
        e->sweep_rounds += done_at + 1;
    };
    int done_a = -1; bool tab_short = false;
    for (int q = 0; q < ra && done_a < 0; q++) {
        const int* f = e->h_round(base_a + q);
        if (f[FLAG_NTIES] & (1 << 30)) tab_short = true;
        if (f[FLAG_CHANGED] == 0) done_a = q;
    }
    if (done_a < 0 || tab_short) { *redo = true; e->cur = 0; e->sweep_counter = 0; e->masks_valid = false; e->flags_clean = false; return NEMGPU_OK; }
    const int blind_draws = e->h_round(base_a + done_a)[FLAG_NTIES] & ((1 << 30) - 1);
    book(e->h_round(base_a + done_a), done_a);
    rounds[0] = done_a + 1;
    int done_b = -1;
    for (int q = 0; q < rb && done_b < 0; q++) {
        const int* f = e->h_round(q);
        if (f[FLAG_NTIES] & (1 << 30)) tab_short = true;
        if (f[FLAG_CHANGED] == 0) done_b = q;
    }
    if (done_b >= 0 && !tab_short) { book(e->h_round(done_b), done_b); rounds[1] = done_b + 1; return NEMGPU_OK; }
    // the beta sweep goes on from the host: its draws-before by value now (the blind sweep's are booked)
    e->cur = 1;
    cb.a.draw_base += blind_draws; cb.a.draw_extra = nullptr;
    if ((r = sweep_complete(e, cb, &rounds[1], nullptr, true, true))) return r;
    e->cur = 2;
    e->masks_valid = false;
    if ((r = do_labels_post(e, 2, -1))) return r;
    return NEMGPU_OK;
}

// How many rounds of an initial sweep go out at once: what `percent` % of the runs so far got by with (which: 0 the blind
// sweep, 1 the beta sweep; the tally lives on the engine that owns the twins)
int libc_init_usual(const nemgpu_engine* e, int which, int percent)
{
    const nemgpu_engine* o = e->parent ? e->parent : e;
    const int* h = o->libc_init_hist[which];
    int total = 0; for (int q = 0; q <= 16; q++) total += h[q];
    int acc = 0, q = 0;
    for (; q < 16; q++) { acc += h[q]; if (total > 0 && 100 * acc >= percent * total) break; }
    return total > 0 ? std::max(2, q) : std::max(3, e->round_batch);      // (nothing seen yet: three)
}
void libc_init_tally(nemgpu_engine* e, const int rounds[2])
{
    nemgpu_engine* o = e->parent ? e->parent : e;
    o->libc_init_hist[0][std::max(0, std::min(16, rounds[0]))]++;
    o->libc_init_hist[1][std::max(0, std::min(16, rounds[1]))]++;
}
// a TIE_LIBC run's start: the one-wait form, or (asked for, or its blind sweep was not through) sweep by sweep
int libc_init(nemgpu_engine* e, int* rounds_beta = nullptr, bool* was_redone = nullptr)
{
    static const bool one_wait = !(getenv("NEM_MI355X_STARTS_ONE_WAIT") && getenv("NEM_MI355X_STARTS_ONE_WAIT")[0] == '0');
    int r;
    int rounds[2] = {0, 0};
    bool redo = false;
    const int ra = std::min(8, libc_init_usual(e, 0, 90)), rb = libc_init_usual(e, 1, 70);
    if (one_wait && current_recorder() == nullptr) { if ((r = libc_init_one_wait(e, ra, rb, &redo, rounds))) return r; }
    if (was_redone) *was_redone = redo;
    if (!one_wait || redo || current_recorder() != nullptr) {
        SweepCtx sc;
        if ((r = libc_init_a(e, sc, ra)) || (r = sweep_complete(e, sc, &rounds[0], nullptr, false, true))) return r;
        if ((r = libc_init_b(e, sc, rb)) || (r = sweep_complete(e, sc, &rounds[1], nullptr, false, true))) return r;
        if ((r = libc_init_c(e))) return r;
    }
    HIPCHK(hipGetLastError());
    libc_init_tally(e, rounds);
    if (rounds_beta) *rounds_beta = rounds[1];
    e->draws_after_init = e->draws;
    return NEMGPU_OK;
}

// ---- one batch of the pipelined loop, in three pieces so that the same code drives ONE engine (iterate) or SEVERAL
// in lock step (iterate_many: the pieces' launches are recorded per engine and issued once for all of them) ----------
struct LoopCursor {
    int remaining = 0;            // iterations still allowed
    bool first = false;           // the next batch starts with the restart + the two initial sweeps
    // the batch in flight
    int g = 0, base = 0; uint32_t sweep0 = 0; bool batch_first = false;
    int deep = 0;                 // leading iterations of the batch that get round_batch relaxation rounds
    int pos0 = 0;                 // run-relative number of the batch's first iteration (fuzzy: selects the learned round counts)
    bool active() const { return remaining > 0 || first; }
};

// host half of a restart (the device half is the head of the first batch)
int loop_begin(nemgpu_engine* e, LoopCursor& lc, int n_iters, bool with_init)
{
    int r;
    lc = LoopCursor();
    lc.remaining = n_iters; lc.first = with_init;
    if (!with_init) { if ((r = flush_reset(e))) return r; }
    if (with_init) {
        if (!e->have_matrix || !e->have_params) { set_error("matrix and parameters must be set first"); return NEMGPU_E_FUNCARG; }
        if ((r = ensure_state_buffers(e))) return r;
        if ((r = ensure_exp_table(e))) return r;
        if ((r = clear_fault(e))) return r;
        e->reset_pending = false;                          // (the head of the first batch is the device half of a reset)
        e->run_deep_used = 0; e->run_tracked = true;
        e->cur = 0; e->sweep_counter = 0;
        e->iters = 0; e->converged = 0; e->emptyk = 0; e->status = NEMGPU_OK;
        e->zero_density = 0; e->first_zero = -1; e->sweep_rounds = 0; e->masks_valid = false;
    }
    return NEMGPU_OK;
}
bool loop_wants_batch(const nemgpu_engine* e, const LoopCursor& lc)
{
    return lc.active() && !e->converged && e->status == NEMGPU_OK;
}

// what the next batch is; the state every batch may rely on (issued / recorded ahead of it)
int batch_plan(nemgpu_engine* e, LoopCursor& lc)
{
    int r;
    lc.g = std::min(lc.remaining, kPipeDepth);
    lc.batch_first = lc.first;
    lc.base = lc.first ? 2 : e->cur;
    lc.sweep0 = lc.first ? 2u : e->sweep_counter;
    lc.deep = current_recorder() != nullptr ? lc.g : std::max(0, std::min(lc.g, e->deep_iters - (lc.first ? 0 : e->iters)));
    lc.pos0 = lc.first ? 0 : e->iters;
    // (fuzzy: the round counts of a batch's iterations depend on where in the run it starts; the graph table's last
    //  index tells the classes apart -- a stale class would only enqueue another number of rounds, never change a result)
    if (!e->ncem() && current_recorder() == nullptr) lc.deep = lc.pos0 >= kFzPositions ? 0 : std::min(7, 1 + lc.pos0 / kPipeDepth);
    // class masks of the current labels, fresh tables when the parameters are fixed (otherwise k_finish rebuilds them
    // inside the batch)
    if (!lc.first) {
        if (e->ncem() && !e->cfg.param_fix && !e->masks_valid) { if ((r = do_labels_post(e, e->cur, -1))) return r; }
        if (e->cfg.param_fix) { if ((r = do_tables(e))) return r; }
    }
    if (e->libc()) {
        if (lc.first) {
            // the rounds of the two initial sweeps: what most runs so far got by with (at most 8: the loop control's window)
            // (95 %: a start that needs more costs a whole batch and a second start from the host here, not one more wait)
            const int ra = std::max(2, std::min(8, libc_init_usual(e, 0, 95))), rb = std::max(2, std::min(8, libc_init_usual(e, 1, 95)));
            if (ra != e->libc_ra || rb != e->libc_rb) { e->libc_ra = ra; e->libc_rb = rb; drop_graphs(e); }
        }
        if ((r = ensure_draw_window(e, e->draws, draw_need(e)))) return r;      // (may drop the graphs)
        if ((r = publish_draw_ctl(e))) return r;
    }
    return NEMGPU_OK;
}

// the batch's launches (issued, captured or recorded by the caller's choice), ending with the copy of the control block
// after_iter (optional): called behind the launches of every iteration of the batch (the logged run's snapshots)
int batch_enqueue(nemgpu_engine* e, LoopCursor& lc, bool with_copy, const std::function<int(int)>* after_iter = nullptr)
{
    int r = NEMGPU_OK;
    e->ctrl_pending = false;                               // (a batch never inherits a deferred loop control)
    hipError_t herr = hipSuccess;
    if (!lc.batch_first) {                                 // (the restart launch clears the loop control itself)
        if (current_recorder()) launch_fill(e->ctrl(), C_WORDS, 0, e->stream);
        else herr = hipMemsetAsync(e->ctrl(), 0, C_WORDS * sizeof(int), e->stream);
    }
    e->stop_ptr = e->ctrl() + C_STOP;
    if (lc.batch_first && herr == hipSuccess) r = enqueue_init(e, lc.g > 0);
    for (int j = 0; j < lc.g && r == NEMGPU_OK && herr == hipSuccess; j++) {
        r = enqueue_iteration(e, (lc.base + j) % 3, lc.sweep0 + j, j + 1 < lc.g && after_iter == nullptr, e->ncem() ? j < lc.deep : false, lc.pos0 + j);
        if (r == NEMGPU_OK && after_iter != nullptr) r = (*after_iter)(j);
    }
    e->stop_ptr = nullptr;
    if (herr == hipSuccess && r == NEMGPU_OK && with_copy)
        herr = hipMemcpyAsync(e->flags_host, e->flags_dev, e->flag_words() * sizeof(int), hipMemcpyDeviceToHost, e->stream);
    if (r) return r;
    HIPCHK(herr);
    return NEMGPU_OK;
}

// after the batch has run (stream synchronised, control block on the host): account for it, finish from the host what
// the pipeline could not
int batch_finish(nemgpu_engine* e, LoopCursor& lc)
{
    int r;
    if ((r = check_fault(e))) return r;
    const int* c = e->h_ctrl();
    const int done = c[C_ITERS], commits = c[C_COMMITS];
    const bool first = lc.batch_first;
    const int base = lc.base;
    const uint32_t sweep0 = lc.sweep0;
    if (first && e->libc()) {
        if (c[C_NEED_ROUNDS] >= 2) {
            // TIE_LIBC: one of the two initial sweeps was not through in the rounds enqueued (or a draw left the table):
            // nothing was booked, everything behind returned at the stop word -- the start is done again from the host
            e->n_host_rounds++;
            e->flags_clean = false;
            if ((r = libc_init(e))) return r;
            lc.first = false;
            return NEMGPU_OK;
        }
        const int got[2] = {c[C_INIT_ROUNDS] >> 8, c[C_INIT_ROUNDS] & 0xFF};
        libc_init_tally(e, got);                               // (what the two sweeps needed, as the loop control saw it)
        e->draws_after_init = e->draws + c[C_DRAWS_INIT];
    }
    e->iters += done;
    e->draws += c[C_DRAWS];
    e->sweep_rounds += c[C_SWEEP_ROUNDS];
    if (c[C_NZERO] > 0) {
        e->zero_density += c[C_NZERO];
        if (e->first_zero < 0) e->first_zero = e->n_total - c[C_FIRSTZERO];
    }
    e->flags_clean = false;
    e->tables_fresh = e->cfg.param_fix;                        // (the fused density kernel does not rebuild the table buffers)
    if (e->ncem()) e->masks_valid = true;
    const int init_launched = first ? init_beta_rounds(e, 1) : 0;    // (before the count below is touched)
    if (first && !e->libc() && e->ncem() && current_recorder() == nullptr && e->has_graph && e->cfg.beta != 0.0f) {
        if (c[C_NEED_ROUNDS] == 2) {
            if (e->init_rounds_ncem != 0) { e->init_rounds_ncem = 0; drop_graphs(e); }       // (back to round_batch)
            e->init_rounds_locked = true; e->init_two_streak = 0;
        } else if (!e->init_rounds_locked && e->init_rounds_ncem == 0) {
            e->init_two_streak = (c[C_INIT_ROUNDS] & 0xFF) <= 2 ? e->init_two_streak + 1 : 0;
            if (e->init_two_streak >= 3 && e->round_batch > 2) { e->init_rounds_ncem = 2; drop_graphs(e); }
        }
    }
    if (first && c[C_NEED_ROUNDS] == 2) {
        // the initial beta sweep (buffers 1 -> 2/0) is not at its fixed point after the enqueued rounds; every
        // iteration behind it returned at the stop word.  Finish it from the host, then go on.
        e->cur = 1;
        e->n_host_rounds++;
        SweepCtx sc;
        const int launched = init_launched;
        if ((r = host_rounds_ctx(e, sc, 1u, launched))) return r;
        if (fused_failed_seen(e, 0, 0, launched)) { if ((r = fused_fallback(e, sc))) return r; }
        if ((r = sweep_launch_rounds(e, sc, e->round_batch))) return r;
        int init_rounds = 0;
        if ((r = sweep_complete(e, sc, &init_rounds, nullptr))) return r;
        if (!e->ncem() && current_recorder() == nullptr && std::min(kRoundsMax, init_rounds) > e->fz_init_need) {
            e->fz_init_need = (uint8_t)std::min(kRoundsMax, init_rounds);     // the next starts get them enqueued
            drop_graphs(e);
        }
        e->sweep_rounds += 1;                                  // + the blind sweep
        e->cur = 2; e->sweep_counter = 2;
        e->masks_valid = false;
        if (e->ncem()) { if ((r = do_labels_post(e, 2, -1))) return r; }
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)e->sweep_next, 2, 1, e->stream));
        lc.first = false;
        return NEMGPU_OK;
    }
    lc.first = false;
    e->cur = (base + commits) % 3;
    e->sweep_counter = sweep0 + (uint32_t)done;
    lc.remaining -= done;
    if (c[C_DEEP] > 0) e->run_deep_used = std::max(e->run_deep_used, e->iters - done + c[C_DEEP]);
    // a whole run in which the leading iterations did not use their extra round: one of them loses it next time
    auto run_ends = [&]() {
        if (e->run_tracked && current_recorder() == nullptr && (e->converged || e->iters >= e->deep_iters) &&
            e->run_deep_used < e->deep_iters) e->deep_iters--;
        e->run_tracked = false;
    };
    if (c[C_STATUS] == NEMGPU_W_EMPTYCLASS) {                  // nem_alg.c:1831-1838
        e->status = NEMGPU_W_EMPTYCLASS;
        e->emptyk = c[C_EMPTYK];
        e->masks_valid = false;                                // the speculative E-step rebuilt them for a discarded partition
        run_ends();
        return NEMGPU_OK;
    }
    if (c[C_CONVERGED]) { e->converged = 1; run_ends(); return NEMGPU_OK; }
    if (c[C_NEED_ROUNDS]) {
        // iteration #commits of this batch ran its M-step, density and the enqueued relaxation rounds and is
        // not at the fixed point yet: continue its rounds from the host, then redo the bookkeeping.
        const int oldbuf = e->cur, newbuf = (e->cur + 1) % 3;
        e->n_host_rounds++;
        SweepCtx sc;
        const int pos = lc.pos0 + done - 1;                          // the run-relative number of that iteration
        const int launched = sweep_first_rounds(e, e->cfg.beta, iteration_rounds(e, pos, e->ncem() && done - 1 < lc.deep), 1);
        if ((r = host_rounds_ctx(e, sc, sweep0 + (uint32_t)(done - 1), launched))) return r;
        if (fused_failed_seen(e, 0, 0, launched)) { if ((r = fused_fallback(e, sc))) return r; }
        e->deep_iters = std::max(e->deep_iters, e->iters);          // from now on: one round more up to this iteration of a run
        e->run_deep_used = std::max(e->run_deep_used, e->iters);
        if ((r = sweep_launch_rounds(e, sc, e->round_batch))) return r;
        int rounds = 0;
        if ((r = sweep_complete(e, sc, &rounds, nullptr))) return r;
        if (!e->ncem() && current_recorder() == nullptr && pos >= 0 && pos < kFzPositions && std::min(kRoundsMax, rounds) > e->fz_need[pos]) {
            e->fz_need[pos] = (uint8_t)std::min(kRoundsMax, rounds);       // this position gets them enqueued from now on
            drop_graphs(e);
        }
        HIPCHK(hipMemsetAsync(e->iter_flags() + FLAG_MOVED, 0, sizeof(int), e->stream));
        if ((r = post_sweep(e, newbuf, oldbuf))) return r;
        if ((r = read_iter_flags(e))) return r;
        e->cur = newbuf;
        if (e->cfg.cvtest == NEMGPU_CV_CLAS) {
            const int moved = e->h_iter()[FLAG_MOVED];
            if (e->ncem()) e->converged = moved ? (1.0f < e->cfg.cvthres) : (0.0f < e->cfg.cvthres);
            else e->converged = !moved;
        }
    }
    if (e->converged || lc.remaining <= 0) run_ends();
    return NEMGPU_OK;
}

int iterate_pipelined(nemgpu_engine* e, int n_iters, bool with_init)
{
    int r;
    LoopCursor lc;
    if ((r = loop_begin(e, lc, n_iters, with_init))) return r;
    if (lc.first && e->libc() && !libc_init_pipelined()) {
        // TIE_LIBC, NEM_MI355X_LIBC_INIT=host: the two initial sweeps are completed from the host (one wait:
        // libc_init_one_wait); the iterations behind them are pipelined
        if ((r = libc_init(e))) return r;
        lc.first = false;
    }
    while (loop_wants_batch(e, lc)) {
        if ((r = batch_plan(e, lc))) return r;
        const int g = lc.g, base = lc.base;
        const bool first = lc.batch_first;
        bool graphed = e->use_graphs && g < 8;
        const int deep = lc.deep;
        hipGraphExec_t exec = graphed ? e->graphs[first ? 1 : 0][base][g][deep] : nullptr;
        // the first batch of a shape goes out as plain launches: capturing and instantiating a graph costs more than
        // it saves unless the batch is replayed, and a nem() call's engine enqueues most shapes once
        if (graphed && exec == nullptr && e->graph_asked[first ? 1 : 0][base][g][deep]++ == 0 && !e->capture_first) graphed = false;
        if (exec == nullptr) {
            if (graphed) e->n_captured++; else e->n_plain++;
            if (graphed) HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
            r = batch_enqueue(e, lc, true);
            if (graphed) {
                hipGraph_t graph = nullptr;
                hipError_t cerr = hipStreamEndCapture(e->stream, &graph);
                if (r == NEMGPU_OK && cerr == hipSuccess && graph != nullptr &&
                    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    e->graphs[first ? 1 : 0][base][g][deep] = exec;
                } else {
                    exec = nullptr;
                    e->n_captured--;
                    e->use_graphs = false;                         // fall back to plain launches for good
                    if (r == NEMGPU_OK && cerr != hipSuccess) { set_error(std::string("graph capture failed: ") + hipGetErrorString(cerr)); }
                }
                if (graph) (void)hipGraphDestroy(graph);
                if (exec == nullptr && r == NEMGPU_OK) {           // redo this batch ungraphed
                    e->sweep_counter = lc.sweep0;
                    if (first) e->cur = 0;
                    continue;
                }
            }
            if (r) return r;
        }
        if (exec != nullptr) { HIPCHK(hipGraphLaunch(exec, e->stream)); e->n_replayed++; }
        HIPCHK(hipStreamSynchronize(e->stream));
        if ((r = batch_finish(e, lc))) return r;
    }
    return NEMGPU_OK;
}

int criteria(nemgpu_engine* e, float crit6[6], int buf);

// CVTEST_CRIT (HasConverged, nem_alg.c:2090-2105): iterate until the chosen criterion (M, DEFAULT_CRIT nem_typ.h:80)
// moves by less than the threshold, relatively.  The criterion is an i-ordered float sum over all families that the
// loop does not otherwise need, so this test runs one iteration per host round trip: iteration, criteria, decision.
// The value the first iteration is compared with: 0 (Criteria = {0}, nem_exe.c:264) or, for a run that logs, the
// criterion of the initial partition (WriteLogCrit at the end of the initial sweep, nem_alg.c:1980, 2398).
int iterate_crit(nemgpu_engine* e, int n_iters, bool with_init)
{
    int r;
    if (with_init) {
        if ((r = iterate_pipelined(e, 0, true))) return r;
        e->crit_ref = 0.0f;
        if (e->cfg.cvtest == NEMGPU_CV_CRIT_LOGGED) {
            float c6[6];
            if ((r = criteria(e, c6, -1))) return r;
            e->crit_ref = c6[3];
        }
    }
    for (int it = 0; it < n_iters && !e->converged && e->status == NEMGPU_OK; it++) {
        const float oldcrit = e->crit_ref;
        if ((r = iterate_pipelined(e, 1, false))) return r;
        if (e->status != NEMGPU_OK) break;
        float c6[6];
        if ((r = criteria(e, c6, -1))) return r;
        const float curcrit = c6[3];
        e->crit_ref = curcrit;
        float critdif;
        if (curcrit != 0) critdif = (float)fabs((curcrit - oldcrit) / curcrit);
        else critdif = FLT_MAX;                                    // MAXFLOAT
        if (critdif < e->cfg.cvthres) e->converged = 1;
    }
    return NEMGPU_OK;
}

bool crit_test(const nemgpu_engine* e) { return e->cfg.cvtest == NEMGPU_CV_CRIT || e->cfg.cvtest == NEMGPU_CV_CRIT_LOGGED; }

int iterate(nemgpu_engine* e, int n_iters, bool with_init = false)
{
    return crit_test(e) ? iterate_crit(e, n_iters, with_init) : iterate_pipelined(e, n_iters, with_init);
}

// ============================================================================================
// Lock-step batches: B independent problems, ONE launch per step for all of them.
// PPanGGOLiN solves many NEM problems of the same kind -- one per 500-organism chunk (ppanggolin.py:1045-1086),
// fifty random starts of one problem in RandNemAlgo (nem_alg.c:1574-1742) -- and one problem of that size leaves
// the chip >95 % empty.  Every engine of a batch runs the code it would run alone, with its launches RECORDED
// (nem_kernels.hpp); the records of all members agree position by position (same kernels, own arguments), so each
// position goes out once, problem = blockIdx.z, and the host synchronises once per batch for everybody.  Members
// whose sequences differ (another batch length, a missing bookkeeping step) form groups of their own.
// ============================================================================================
int zip_reserve(nemgpu_engine* lead, size_t bytes)
{
    nemgpu_engine::ZipContext* z = zip_context(lead);
    if (z->zip_cap >= bytes) return NEMGPU_OK;
    HIPCHK(hipStreamSynchronize(lead->stream));
    for (auto& g : z->zip_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);   // (they hold the old slab's addresses)
    z->zip_graphs.clear();
    pool_put(lead->device, true, z->zip_host, z->zip_host_size);
    pool_put(lead->device, false, z->zip_dev, z->zip_dev_size);
    z->zip_host = nullptr; z->zip_dev = nullptr; z->zip_cap = 0;
    size_t cap = (size_t)1 << 20, got = 0;
    while (cap < bytes) cap *= 2;
    HIPCHK(pool_get(lead->device, true, cap, &z->zip_host, &got)); z->zip_host_size = got;
    HIPCHK(pool_get(lead->device, false, cap, &z->zip_dev, &got)); z->zip_dev_size = got;
    z->zip_cap = cap;
    return NEMGPU_OK;
}

// issue the recorded sequences of `members` (indices into recs / E) on the lead's stream; flags_words > 0: behind them,
// one copy of every member's flag block to the host (gathered on the device first)
int zip_and_launch(nemgpu_engine* lead, const std::vector<Recorder>& recs, const std::vector<int>& members, size_t flags_words)
{
    if (members.empty()) return NEMGPU_OK;
    nemgpu_engine::ZipContext* z = zip_context(lead);
    // groups of members with the same sequence of (kernel, variant, block, grid height)
    std::vector<std::vector<int>> groups;
    for (int m : members) {
        bool placed = false;
        for (auto& g : groups) {
            const std::vector<OpRecord>& a = recs[g[0]].ops; const std::vector<OpRecord>& b = recs[m].ops;
            bool same = a.size() == b.size();
            for (size_t t = 0; same && t < a.size(); t++)
                same = a[t].kind == b[t].kind && a[t].variant == b[t].variant && a[t].block == b[t].block && a[t].gy == b[t].gy &&
                       a[t].nbytes == b[t].nbytes;
            if (same) { g.push_back(m); placed = true; break; }
        }
        if (!placed) groups.push_back({m});
    }
    size_t total = 0;
    for (const auto& g : groups)
        for (const OpRecord& o : recs[g[0]].ops) total += ((size_t)((o.nbytes + 15) & ~15) + 16) * g.size() + 64;
    int r = zip_reserve(lead, total);
    if (r) return r;
    struct Launch { int kind, variant, B, stride; size_t args_off, gx_off; unsigned max_gx, gy, block; };
    std::vector<Launch> launches;
    size_t off = 0;
    uint64_t key = 1469598103934665603ull;                        // FNV-1a over everything a captured sequence depends on
    std::vector<uint64_t> desc;                                   // ... and the list itself (compared on a key hit)
    auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    for (const auto& g : groups) {
        const int B = (int)g.size();
        for (size_t t = 0; t < recs[g[0]].ops.size(); t++) {
            const OpRecord& o0 = recs[g[0]].ops[t];
            const int stride = (o0.nbytes + 15) & ~15;
            Launch L{o0.kind, o0.variant, B, stride, off, 0, 0, o0.gy, o0.block};
            for (int b = 0; b < B; b++) memcpy(z->zip_host + off + (size_t)b * stride, recs[g[b]].ops[t].args, (size_t)o0.nbytes);
            off += (size_t)B * stride;
            L.gx_off = off;
            int* gx = reinterpret_cast<int*>(z->zip_host + off);
            for (int b = 0; b < B; b++) { gx[b] = (int)recs[g[b]].ops[t].gx; L.max_gx = std::max(L.max_gx, recs[g[b]].ops[t].gx); }
            off += ((size_t)B * sizeof(int) + 15) & ~(size_t)15;
            launches.push_back(L);
            for (uint64_t v : {(uint64_t)L.kind, (uint64_t)L.variant, (uint64_t)B, (uint64_t)stride, (uint64_t)L.max_gx, (uint64_t)L.gy,
                               (uint64_t)L.block, (uint64_t)L.args_off, (uint64_t)L.gx_off}) { mix(v); desc.push_back(v); }
        }
    }
    mix(flags_words); desc.push_back(flags_words);
    if (off == 0 && flags_words == 0) return NEMGPU_OK;
    if (off) HIPCHK(hipMemcpyAsync(z->zip_dev, z->zip_host, off, hipMemcpyHostToDevice, lead->stream));
    // The launches themselves depend only on the shape (kernels, grids, slab offsets), not on the argument blocks'
    // content: a shape seen before is replayed from its captured graph
    nemgpu_engine::ZipGraph* slot = nullptr;
    for (auto& g : z->zip_graphs) if (g.key == key && g.desc == desc) { slot = &g; break; }
    if (slot == nullptr && lead->use_graphs) {
        if (z->zip_graphs.size() >= 64) {
            for (auto& g : z->zip_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
            z->zip_graphs.clear();
        }
        z->zip_graphs.push_back({key, 0, nullptr, desc});
        slot = &z->zip_graphs.back();
    }
    auto issue = [&]() -> int {
        for (const Launch& L : launches)
            launch_zipped(L.kind, L.variant, L.B, z->zip_dev + L.args_off, L.stride, reinterpret_cast<const int*>(z->zip_dev + L.gx_off),
                          L.max_gx, L.gy, L.block, lead->stream);
        if (flags_words)
            HIPCHK(hipMemcpyAsync(z->zip_flags_host, z->zip_flags_dev, flags_words * sizeof(int), hipMemcpyDeviceToHost, lead->stream));
        return NEMGPU_OK;
    };
    if (slot != nullptr && slot->exec != nullptr) {
        HIPCHK(hipGraphLaunch(slot->exec, lead->stream));
        return NEMGPU_OK;
    }
    // Capturing and instantiating costs ~160 us per node (8 ms for a batch of 7 iterations) and a replay saves
    // 0.15-0.35 ms of issue time: a shape pays for its graph after some thirty replays.  The contexts live as long as
    // the process (a pangenome's chunks are hundreds of batches of one shape), so: from the fourth sighting on.
    const bool capture = slot != nullptr && (slot->asked++ >= 3 || lead->capture_first);
    if (capture) {
        HIPCHK(hipStreamBeginCapture(lead->stream, hipStreamCaptureModeThreadLocal));
        r = issue();
        hipGraph_t graph = nullptr;
        const hipError_t cerr = hipStreamEndCapture(lead->stream, &graph);
        hipGraphExec_t exec = nullptr;
        if (r == NEMGPU_OK && cerr == hipSuccess && graph != nullptr && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            slot->exec = exec;
            if (graph) (void)hipGraphDestroy(graph);
            HIPCHK(hipGraphLaunch(exec, lead->stream));
            return NEMGPU_OK;
        }
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        lead->use_graphs = false;                                  // plain launches from now on
    }
    if ((r = issue())) return r;
    HIPCHK(hipGetLastError());
    return NEMGPU_OK;
}

// record `fn(engine)` for every member, issue the records zipped, fetch every member's flag block, wait
template <typename F>
int lockstep(std::vector<nemgpu_engine*>& E, const std::vector<int>& members, std::vector<Recorder>& recs, F&& fn, bool fetch_flags)
{
    if (members.empty()) return NEMGPU_OK;
    nemgpu_engine* lead = E[0];
    HIPCHK(hipStreamSynchronize(lead->stream));                // (the argument slab of the previous step is free again)
    static const bool prof = getenv("NEM_MI355X_BATCH_PROF") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    int r = NEMGPU_OK;
    for (int m : members) {
        recs[m].ops.clear();
        set_recorder(&recs[m]);
        r = fn(m);
        set_recorder(nullptr);
        if (r) return r;
    }
    const auto t1 = std::chrono::steady_clock::now();
    const size_t fw = lead->flag_words();
    nemgpu_engine::ZipContext* z = zip_context(lead);
    if (fetch_flags) {
        // every member's flag block goes to one staging area on the device (a last zipped launch) and from there
        // to the host in ONE copy
        const size_t need = fw * E.size();
        if (z->zip_flags_cap < need) {
            pool_put(lead->device, true, (char*)z->zip_flags_host, z->zip_flags_host_size);
            pool_put(lead->device, false, (char*)z->zip_flags_dev, z->zip_flags_dev_size);
            z->zip_flags_host = nullptr; z->zip_flags_dev = nullptr; z->zip_flags_cap = 0;
            for (auto& g : z->zip_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
            z->zip_graphs.clear();
            HIPCHK(pool_get(lead->device, true, need * sizeof(int), (char**)&z->zip_flags_host, &z->zip_flags_host_size));
            HIPCHK(pool_get(lead->device, false, need * sizeof(int), (char**)&z->zip_flags_dev, &z->zip_flags_dev_size));
            z->zip_flags_cap = need;
        }
        size_t slot = 0;
        for (int m : members) {
            set_recorder(&recs[m]);
            launch_copy_words(E[m]->flags_dev, z->zip_flags_dev + slot * fw, (int)fw, lead->stream);
            set_recorder(nullptr);
            slot++;
        }
    }
    if ((r = zip_and_launch(lead, recs, members, fetch_flags ? fw * members.size() : 0))) return r;
    const auto t2 = std::chrono::steady_clock::now();
    HIPCHK(hipStreamSynchronize(lead->stream));
    if (fetch_flags) {
        size_t slot = 0;
        for (int m : members) { memcpy(E[m]->flags_host, z->zip_flags_host + slot * fw, fw * sizeof(int)); slot++; }
    }
    if (prof) {
        const auto t3 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[lockstep] members %zu ops %zu: record %.0f us, zip+launch+copies %.0f us, wait %.0f us\n", members.size(),
                recs[members[0]].ops.size(), us(t0, t1), us(t1, t2), us(t2, t3));
    }
    return NEMGPU_OK;
}

// the EM loops of several engines in lock step (all on E[0]'s stream); L: their cursors after loop_begin
// a member whose matrix was uploaded by nemgpu_solve_many's builders: its device layouts ride at the head of the group's
// first step (recorded: one launch each for all members)
static void pending_layout(nemgpu_engine* e)
{
    if (!e->layout_pending) return;
    launch_layout(e->xf_stage, e->n, e->wf, e->W, e->npad, e->d, e->nw64, e->xw, e->xt, e->perm, e->xws, e->stream);
    e->layout_pending = false;
}

int iterate_many(std::vector<nemgpu_engine*>& E, std::vector<LoopCursor>& L)
{
    int r;
    const int B = (int)E.size();
    std::vector<Recorder> recs((size_t)B);
    std::vector<int> members;
    // TIE_LIBC members: the two initial sweeps, completed one after the other (see libc_init_a)
    members.clear();
    for (int i = 0; i < B; i++) if (L[i].first && E[i]->libc() && !libc_init_pipelined()) members.push_back(i);
    if (!members.empty()) {
        std::vector<SweepCtx> ctx((size_t)B);
        if ((r = lockstep(E, members, recs, [&](int m) { pending_layout(E[m]); return libc_init_a(E[m], ctx[m]); }, true))) return r;
        for (int m : members) if ((r = sweep_complete(E[m], ctx[m], nullptr, nullptr, true))) return r;
        if ((r = lockstep(E, members, recs, [&](int m) { return libc_init_b(E[m], ctx[m]); }, true))) return r;
        for (int m : members) if ((r = sweep_complete(E[m], ctx[m], nullptr, nullptr, true))) return r;
        if ((r = lockstep(E, members, recs, [&](int m) { return libc_init_c(E[m]); }, false))) return r;
        for (int m : members) { L[m].first = false; E[m]->draws_after_init = E[m]->draws; }
    }
    for (;;) {
        members.clear();
        for (int i = 0; i < B; i++) if (loop_wants_batch(E[i], L[i])) members.push_back(i);
        if (members.empty()) break;
        r = lockstep(E, members, recs, [&](int m) {
            pending_layout(E[m]);
            int rr = batch_plan(E[m], L[m]);
            if (rr == NEMGPU_OK) rr = batch_enqueue(E[m], L[m], false);
            return rr;
        }, true);
        if (r) return r;
        for (int m : members) { E[m]->n_plain++; if ((r = batch_finish(E[m], L[m]))) return r; }
    }
    return NEMGPU_OK;
}

// ---- RandNemAlgo's per-iteration log with the starts in lock step ------------------------------------------------
// What one start writes to <Fname>.log (nem_alg.c:1662-1669, 2361, 2398): line 0 for its initial partition, then a line
// per iteration -- the six criteria of the partition the E-step started from and of the one it left, both on the new
// densities, and the iteration's parameters.  The criteria are i-ordered float sums (four lone blocks per partition,
// 75-100 us at configs[1] size): one start after the other they are most of a logged iteration; with the starts in lock
// step the blocks of all starts run side by side.  The lines are kept per start and handed to the caller's writer in the
// reference's order afterwards.
struct StartLogLine {
    int kind = NEMGPU_LOG_LINE, iter = 0, emptyk = 0;
    float cb[6] = {0, 0, 0, 0, 0, 0}, ca[6] = {0, 0, 0, 0, 0, 0};
    std::vector<float> prop, center, disp, nb;
    bool has_nb = false;
};
struct StartLog { std::vector<StartLogLine> lines; };

int criteria(nemgpu_engine* e, float crit6[6], int buf);

// E: the starts' engines behind their initial partitions (cur = 2; blind partition in buffer 1); L: their cursors with
// first = false; host_par: the starts' own parameters, `par` floats apart (prop | center | disp)
static int iterate_many_logged(std::vector<nemgpu_engine*>& E, std::vector<LoopCursor>& L, const float* host_par, size_t par,
                               std::vector<StartLog>& logs)
{
    int r;
    const int B = (int)E.size();
    nemgpu_engine* lead = E[0];
    const int k = lead->k, d = lead->d;
    const size_t kd = (size_t)k * d, W = 12 + lead->par_words;
    std::vector<Recorder> recs((size_t)B);
    std::vector<int> members((size_t)B), left((size_t)B), it_no((size_t)B, 0), oldbuf((size_t)B, 0), newbuf((size_t)B, 0);
    for (int i = 0; i < B; i++) { members[i] = i; left[i] = L[i].remaining; }
    char* st_dev = nullptr; char* st_host = nullptr; size_t dev_size = 0, host_size = 0;
    HIPCHK(pool_get(lead->device, false, (size_t)B * W * sizeof(float), &st_dev, &dev_size));
    if (pool_get(lead->device, true, (size_t)B * W * sizeof(float), &st_host, &host_size) != hipSuccess) {
        pool_put(lead->device, false, st_dev, dev_size);
        set_error("no pinned memory for the starts' log lines"); return NEMGPU_E_DEVICE;
    }
    struct Back { int dev; char* a; size_t as; char* b; size_t bs; hipStream_t s; ~Back() { (void)hipStreamSynchronize(s); pool_put(dev, false, a, as); pool_put(dev, true, b, bs); } }
        back{lead->device, st_dev, dev_size, st_host, host_size, lead->stream};
    auto slot_dev = [&](int m) { return reinterpret_cast<int*>(st_dev) + (size_t)m * W; };
    auto slot_host = [&](int m) { return reinterpret_cast<const float*>(st_host) + (size_t)m * W; };
    auto fetch = [&]() -> int {
        HIPCHK(hipMemcpyAsync(st_host, st_dev, (size_t)B * W * sizeof(float), hipMemcpyDeviceToHost, lead->stream));
        HIPCHK(hipStreamSynchronize(lead->stream));
        return NEMGPU_OK;
    };
    auto two_criteria = [&](int m, int before, int after) -> int {   // (recorded: zipped over the members)
        nemgpu_engine* c = E[m];
        int rr = criteria_enqueue(c, before);
        if (rr == NEMGPU_OK) launch_copy_words(reinterpret_cast<const int*>(c->crit6_dev), slot_dev(m), 6, lead->stream);
        if (rr == NEMGPU_OK) rr = criteria_enqueue(c, after);
        if (rr == NEMGPU_OK) launch_copy_words(reinterpret_cast<const int*>(c->crit6_dev), slot_dev(m) + 6, 6, lead->stream);
        return rr;
    };
    // ---- line 0: the initial partition (before: the blind sweep's, buffer 1; after: the beta sweep's, buffer 2)
    if ((r = lockstep(E, members, recs, [&](int m) { return two_criteria(m, 1, 2); }, false))) return r;
    if ((r = fetch())) return r;
    for (int m = 0; m < B; m++) {
        StartLogLine ln;
        ln.iter = 0;
        memcpy(ln.cb, slot_host(m), sizeof ln.cb); memcpy(ln.ca, slot_host(m) + 6, sizeof ln.ca);
        const float* p0 = host_par + (size_t)m * par;
        ln.prop.assign(p0, p0 + k); ln.center.assign(p0 + k, p0 + k + kd); ln.disp.assign(p0 + k + kd, p0 + k + 2 * kd);
        logs[(size_t)m].lines.push_back(std::move(ln));
    }
    // ---- one iteration per step for every start that still runs
    for (;;) {
        members.clear();
        for (int i = 0; i < B; i++) if (left[i] > 0 && !E[i]->converged && E[i]->status == NEMGPU_OK) members.push_back(i);
        if (members.empty()) break;
        r = lockstep(E, members, recs, [&](int m) {
            nemgpu_engine* c = E[m];
            L[m].remaining = 1;
            int rr = batch_plan(c, L[m]);
            oldbuf[m] = L[m].base; newbuf[m] = (L[m].base + 1) % 3;
            if (rr == NEMGPU_OK) rr = batch_enqueue(c, L[m], false);
            if (rr == NEMGPU_OK) rr = two_criteria(m, oldbuf[m], newbuf[m]);     // (speculated: void if the sweep was not through)
            if (rr == NEMGPU_OK) launch_copy_words(reinterpret_cast<const int*>(c->prop), slot_dev(m) + 12, (int)c->par_words, lead->stream);
            return rr;
        }, true);
        if (r) return r;
        if ((r = fetch())) return r;
        for (int m : members) {
            nemgpu_engine* c = E[m];
            const bool clean = c->h_ctrl()[C_NEED_ROUNDS] == 0 && c->h_ctrl()[C_STATUS] != NEMGPU_W_EMPTYCLASS;
            c->n_plain++;
            if ((r = batch_finish(c, L[m]))) return r;
            left[m] -= 1; it_no[m] += 1;
            StartLogLine ln;
            ln.iter = it_no[m];
            if (c->status == NEMGPU_W_EMPTYCLASS) {
                // EstimSizes (nem_mod.c:1275-1317) ran before the empty class was found: i-ordered float sums per class
                std::vector<float> part((size_t)c->n * k);
                if ((r = nemgpu_get_partition(c, part.data()))) return r;
                ln.kind = NEMGPU_LOG_EMPTY; ln.emptyk = c->emptyk; ln.nb.assign((size_t)k, 0.0f); ln.has_nb = true;
                for (int h = 0; h < k; h++) {
                    float acc = 0.0f;
                    for (int i = 0; i < c->n; i++) acc += part[(size_t)i * k + h];
                    ln.nb[(size_t)h] = acc;
                }
                logs[(size_t)m].lines.push_back(std::move(ln));
                continue;
            }
            ln.prop.resize((size_t)k); ln.center.resize(kd); ln.disp.resize(kd); ln.nb.resize((size_t)k); ln.has_nb = true;
            if (clean) {
                const float* f = slot_host(m);
                memcpy(ln.cb, f, sizeof ln.cb); memcpy(ln.ca, f + 6, sizeof ln.ca);
                memcpy(ln.prop.data(), f + 12, sizeof(float) * k);
                memcpy(ln.center.data(), f + 12 + c->par_o_center, sizeof(float) * kd);
                memcpy(ln.disp.data(), f + 12 + c->par_o_disp, sizeof(float) * kd);
                memcpy(ln.nb.data(), f + 12 + c->par_o_nb, sizeof(float) * k);
            } else {
                // (the sweep was finished from the host: the plain sequence of calls, this start alone)
                if ((r = criteria(c, ln.cb, (c->cur + 2) % 3)) || (r = criteria(c, ln.ca, -1))) return r;
                if ((r = nemgpu_get_results(c, ln.prop.data(), ln.center.data(), ln.disp.data(), ln.nb.data(), nullptr))) return r;
            }
            logs[(size_t)m].lines.push_back(std::move(ln));
        }
    }
    for (int i = 0; i < B; i++) L[i].remaining = 0;
    return NEMGPU_OK;
}

// several whole runs (nemgpu_run) in lock step; every engine complete (matrix, graph, parameters, configuration), all on
// one device.  They are run on E[0]'s stream for the duration.
// A group's results in one block (nemgpu_solve_many): every member's partition and parameter block (the layout of
// nemgpu_get_results' staging block) are copied by zipped launches behind the criteria into `dev`, `stride` bytes apart,
// and reach `host` (pinned) in ONE copy before run_many returns.  Per-member copies from the workers' threads cost
// 0.2-0.7 ms each once 8-16 threads were inside the runtime together.
struct GroupFetch { char* dev = nullptr; char* host = nullptr; size_t stride = 0; bool filled = false; };
static size_t result_part_bytes(const nemgpu_engine* e) { return e->ncem() ? ((size_t)e->n + 3) & ~(size_t)3 : sizeof(float) * (size_t)e->n * e->k; }
static size_t result_block_bytes(const nemgpu_engine* e) { return result_part_bytes(e) + sizeof(float) * e->par_words; }

// a staging block [partition (b_part bytes) | prop | center | disp | nbobs_k] into the caller's arrays
static void result_unpack(const nemgpu_engine* e, const char* st, size_t b_part, float* prop, float* center, float* disp, float* nbobs_k, float* c_nk)
{
    const size_t kd = (size_t)e->k * e->d, n = (size_t)e->n, k = (size_t)e->k;
    const char* par = st + b_part;
    if (c_nk) {
        if (e->ncem()) {
            const uint8_t* lab = (const uint8_t*)st;
            for (size_t i = 0; i < n; i++)                          // LabelToClassVector, nem_alg.c:649-664
                for (size_t h = 0; h < k; h++) c_nk[i * k + h] = ((size_t)(lab[i] & 0x7F) == h) ? 1.0f : 0.0f;
        } else memcpy(c_nk, st, sizeof(float) * n * k);
    }
    if (prop) memcpy(prop, par, sizeof(float) * k);
    if (center) memcpy(center, par + sizeof(float) * e->par_o_center, sizeof(float) * kd);
    if (disp) memcpy(disp, par + sizeof(float) * e->par_o_disp, sizeof(float) * kd);
    if (nbobs_k) memcpy(nbobs_k, par + sizeof(float) * e->par_o_nb, sizeof(float) * k);
}

int run_many(std::vector<nemgpu_engine*>& E, nemgpu_result* results, GroupFetch* gf = nullptr)
{
    int r;
    const int B = (int)E.size();
    nemgpu_engine* lead = E[0];
    for (int i = 0; i < B; i++) {
        if (E[i]->device != lead->device) { set_error("a batch lives on one device"); return NEMGPU_E_ARG; }
        if (E[i]->lo != 0 || E[i]->hi != E[i]->n_total) { set_error("sharded engines cannot join a batch"); return NEMGPU_E_ARG; }
    }
    // The members run on the lead's stream for the duration.  Whatever way this function is left, the lead's stream is
    // waited for BEFORE the members get their own streams back: a later call on a member's stream must not race with
    // work still queued on the lead's.
    struct StreamLoan {
        std::vector<nemgpu_engine*>& E; nemgpu_engine* lead; std::vector<hipStream_t> own; int taken = 0;
        StreamLoan(std::vector<nemgpu_engine*>& E_, nemgpu_engine* l) : E(E_), lead(l), own(E_.size()) {}
        void give_back() {
            if (taken == 0) return;
            (void)hipStreamSynchronize(lead->stream);
            for (int i = 0; i < taken; i++) E[i]->stream = own[i];
            taken = 0;
        }
        ~StreamLoan() { give_back(); }
    } loan(E, lead);
    static const bool prof = getenv("NEM_MI355X_BATCH_PROF") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    double laps[6] = {0, 0, 0, 0, 0, 0};
    auto lap = [&](int j) { auto t1 = std::chrono::steady_clock::now(); laps[j] += std::chrono::duration<double>(t1 - tp).count() * 1e6; tp = t1; };
    for (int i = 0; i < B; i++) {
        if (E[i]->ready_pending) {                             // (built by nemgpu_solve_many: see ready_ev)
            HIPCHK(hipStreamWaitEvent(lead->stream, E[i]->ready_ev, 0));
            E[i]->ready_pending = false;
        } else HIPCHK(hipStreamSynchronize(E[i]->stream));     // (an early return gives back what was taken so far)
        loan.own[i] = E[i]->stream;
        E[i]->stream = lead->stream;
        loan.taken = i + 1;
    }
    auto restore = [&]() { loan.give_back(); };
    for (int i = 0; i < B; i++) {
        if (crit_test(E[i])) {                                     // (a host round trip per iteration: nothing to share)
            for (int j = 0; j < B; j++) pending_layout(E[j]);      // (still on the lead's stream, behind the waits for the uploads)
            restore();
            for (int j = 0; j < B; j++) { int rr = nemgpu_run(E[j], results ? &results[j] : nullptr); if (rr) return rr; }
            return NEMGPU_OK;
        }
    }
    lap(0);
    auto t0 = std::chrono::steady_clock::now();
    std::vector<LoopCursor> L((size_t)B);
    r = NEMGPU_OK;
    for (int i = 0; i < B && r == NEMGPU_OK; i++) { E[i]->draws = 0; r = loop_begin(E[i], L[i], E[i]->cfg.it_max, true); }
    lap(1);
    if (r == NEMGPU_OK) r = iterate_many(E, L);
    for (int i = 0; i < B; i++) pending_layout(E[i]);             // (it_max 0: no step has carried them)
    lap(2);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < B && r == NEMGPU_OK; i++) {
        if (E[i]->iters == 0) {                                    // nem_alg.c:1845-1851
            if ((r = do_mstep(E[i])) || (r = do_tables(E[i])) || (r = do_density(E[i]))) break;
        }
    }
    lap(3);
    if (r == NEMGPU_OK && results != nullptr) {
        std::vector<Recorder> recs((size_t)B);
        std::vector<int> all((size_t)B);
        for (int i = 0; i < B; i++) all[i] = i;
        // every member's six criteria go to the batch's staging area (zipped copies behind the zipped criteria launches)
        // and reach the host in ONE copy: a copy per member into the caller's pageable result blocks cost ~20 us each
        // (16 problems: 0.3 ms of a 2 ms batch)
        nemgpu_engine::ZipContext* z = zip_context(lead);
        const bool staged = z != nullptr && z->zip_flags_dev != nullptr && z->zip_flags_host != nullptr && z->zip_flags_cap >= (size_t)8 * B;
        r = lockstep(E, all, recs, [&](int m) {
            const int rr = criteria_enqueue(E[m], -1);
            if (rr == NEMGPU_OK && staged) launch_copy_words(reinterpret_cast<const int*>(E[m]->crit6_dev), z->zip_flags_dev + (size_t)8 * m, 6, lead->stream);
            if (rr == NEMGPU_OK && gf != nullptr) {
                nemgpu_engine* e = E[m];
                char* dst = gf->dev + gf->stride * (size_t)m;
                const size_t b_part = result_part_bytes(e);
                const void* part = e->ncem() ? (const void*)(e->lab[e->cur] + e->lo) : (const void*)(e->cbuf[e->cur] + (size_t)e->lo * e->k);
                launch_copy_words(reinterpret_cast<const int*>(part), reinterpret_cast<int*>(dst), (int)(b_part / 4), lead->stream);
                launch_copy_words(reinterpret_cast<const int*>(e->prop), reinterpret_cast<int*>(dst + b_part), (int)e->par_words, lead->stream);
            }
            return rr;
        }, false);
        if (r == NEMGPU_OK && gf != nullptr) {
            if (hipMemcpyAsync(gf->host, gf->dev, gf->stride * (size_t)B, hipMemcpyDeviceToHost, lead->stream) != hipSuccess) {
                set_error("results copy failed"); r = NEMGPU_E_DEVICE;
            } else gf->filled = true;                              // (once the wait below is over)
        }
        if (r == NEMGPU_OK && staged &&
            hipMemcpyAsync(z->zip_flags_host, z->zip_flags_dev, (size_t)8 * B * sizeof(int), hipMemcpyDeviceToHost, lead->stream) != hipSuccess) {
            set_error("criteria copy failed"); r = NEMGPU_E_DEVICE;
        }
        for (int i = 0; i < B && r == NEMGPU_OK; i++) {
            fill_result(E[i], &results[i]);
            results[i].loop_seconds = secs;
            if (!staged && hipMemcpyAsync(results[i].crit, E[i]->crit6_dev, 6 * sizeof(float), hipMemcpyDeviceToHost, lead->stream) != hipSuccess) {
                set_error("criteria copy failed"); r = NEMGPU_E_DEVICE;
            }
        }
        if (r == NEMGPU_OK && hipStreamSynchronize(lead->stream) != hipSuccess) { set_error("synchronisation failed"); r = NEMGPU_E_DEVICE; }
        if (r == NEMGPU_OK && staged)
            for (int i = 0; i < B; i++) memcpy(results[i].crit, z->zip_flags_host + (size_t)8 * i, 6 * sizeof(float));
    }
    lap(4);
    restore();
    lap(5);
    if (prof) fprintf(stderr, "[run_many] %d members: waits for the uploads %.0f us, loop_begin %.0f, iterations %.0f, it_max 0 %.0f, criteria + results %.0f, streams back %.0f\n",
                      B, laps[0], laps[1], laps[2], laps[3], laps[4], laps[5]);
    return r;
}

void drop_graphs(nemgpu_engine* e)
{
    for (auto& g : e->shard_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    e->shard_graphs.clear();
    for (auto& plane : e->graphs)
        for (auto& row : plane)
            for (auto& col : row)
                for (auto& g : col)
                    if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
}

int ensure_crit_buffers(nemgpu_engine* e)
{
    alloc_for(e);
    int r;
    size_t nk = (size_t)e->n * e->k;
    if (!e->crit_dik) {
        if ((r = dev_alloc(&e->crit_dik, nk))) return r;
        if ((r = dev_alloc(&e->crit_gik, nk))) return r;
        if ((r = dev_alloc(&e->crit_lfi, (size_t)e->n))) return r;
        if ((r = dev_alloc(&e->crit_lzi, (size_t)e->n))) return r;
        if ((r = dev_alloc(&e->crit6_dev, 16))) return r;     // six criteria + the four chains' results
    }
    if (e->ncem() && !e->c_onehot) { if ((r = dev_alloc(&e->c_onehot, (size_t)e->n_total * e->k))) return r; }
    return NEMGPU_OK;
}

const float* float_partition(nemgpu_engine* e, int buf)
{
    if (!e->ncem()) return e->cbuf[buf];
    launch_onehot(e->n_total, e->k, e->lab[buf], e->c_onehot, e->stream);
    return e->c_onehot;
}

// ComputeCrit on the partition in buffer `buf` (default: the current one) with the current densities: the launches
int criteria_enqueue(nemgpu_engine* e, int buf = -1)
{
    int r;
    if (e->lo != 0 || e->hi != e->n_total) { set_error("criteria need the whole partition on one engine"); return NEMGPU_E_FUNCARG; }
    if ((r = ensure_crit_buffers(e))) return r;
    const float* c = float_partition(e, buf < 0 ? e->cur : buf);
    launch_criteria(e->n, e->k, e->npad, e->nei_ptr, e->nei_idx, e->nei_w, e->has_graph ? 1 : 0, e->cfg.beta, c,
                    e->pkfki, e->logpkfki, e->crit_dik, e->crit_gik, e->crit_lfi, e->crit_lzi, e->crit6_dev,
                    e->ncem() ? 1 : 0, e->stream);
    if (!current_recorder()) HIPCHK(hipGetLastError());
    return NEMGPU_OK;
}
// The criteria of TWO partitions of the engine (label / membership buffers buf_a and buf_b) in the same launches --
// problem = blockIdx.z, through the kernels' batched twins: the i-ordered reductions are four lone blocks each and
// take as long for two partitions as for one.  Results: crit6_dev (buf_a), crit2_6 (buf_b).
int criteria_pair_enqueue(nemgpu_engine* e, int buf_a, int buf_b)
{
    int r;
    if (e->lo != 0 || e->hi != e->n_total) { set_error("criteria need the whole partition on one engine"); return NEMGPU_E_FUNCARG; }
    if ((r = ensure_crit_buffers(e))) return r;
    const size_t nk = (size_t)e->n * e->k;
    if (!e->crit2_dik) {
        alloc_for(e);
        if ((r = dev_alloc(&e->crit2_dik, nk))) return r;
        if ((r = dev_alloc(&e->crit2_gik, nk))) return r;
        if ((r = dev_alloc(&e->crit2_lfi, (size_t)e->n))) return r;
        if ((r = dev_alloc(&e->crit2_lzi, (size_t)e->n))) return r;
        if ((r = dev_alloc(&e->crit2_6, 16))) return r;
        if (e->ncem()) { if ((r = dev_alloc(&e->c_onehot2, (size_t)e->n_total * e->k))) return r; }
        if ((r = dev_alloc(&e->crit_pair_dev, 2048))) return r;
    }
    // argument blocks: [2 x OnehotArgs | gx] [2 x CritArgs | gx terms | gx reduce | gx final]
    constexpr int so = (sizeof(OnehotArgs) + 15) & ~15, sc = (sizeof(CritArgs) + 15) & ~15;
    static_assert(2 * so + 16 + 2 * sc + 48 <= 2048, "argument slab of the paired criteria");
    char* st = stage(e, 2048);
    if (!st) { set_error("no staging memory for the paired criteria"); return NEMGPU_E_DEVICE; }
    memset(st, 0, 2048);
    const int o_oh = 0, o_ohgx = 2 * so, o_cr = o_ohgx + 16, o_gxt = o_cr + 2 * sc, o_gxr = o_gxt + 16, o_gxf = o_gxr + 16;
    const float* cpart[2];
    const int bufs[2] = {buf_a, buf_b};
    float* onehots[2] = {e->c_onehot, e->c_onehot2};
    const unsigned g_oh = (unsigned)(((size_t)e->n_total * e->k + 255) / 256), g_t = (unsigned)((e->n + 255) / 256);
    for (int p = 0; p < 2; p++) {
        if (e->ncem()) {
            OnehotArgs oa{e->n_total, e->k, e->lab[bufs[p]], onehots[p]};
            memcpy(st + o_oh + p * so, &oa, sizeof oa);
            cpart[p] = onehots[p];
        } else cpart[p] = e->cbuf[bufs[p]];
        CritArgs ca{e->n, e->k, e->npad, e->nei_ptr, e->nei_idx, e->nei_w, e->has_graph ? 1 : 0, e->cfg.beta, cpart[p], e->pkfki, e->logpkfki,
                    p ? e->crit2_dik : e->crit_dik, p ? e->crit2_gik : e->crit_gik, p ? e->crit2_lfi : e->crit_lfi,
                    p ? e->crit2_lzi : e->crit_lzi, p ? e->crit2_6 : e->crit6_dev, e->ncem() ? 1 : 0};
        memcpy(st + o_cr + p * sc, &ca, sizeof ca);
        reinterpret_cast<int*>(st + o_ohgx)[p] = (int)g_oh;
        reinterpret_cast<int*>(st + o_gxt)[p] = (int)g_t;
        reinterpret_cast<int*>(st + o_gxr)[p] = 4;
        reinterpret_cast<int*>(st + o_gxf)[p] = 1;
    }
    HIPCHK(hipMemcpyAsync(e->crit_pair_dev, st, 2048, hipMemcpyHostToDevice, e->stream));
    const char* dv = e->crit_pair_dev;
    if (e->ncem()) launch_zipped(OP_ONEHOT, 0, 2, dv + o_oh, so, reinterpret_cast<const int*>(dv + o_ohgx), g_oh, 1, 256, e->stream);
    launch_zipped(OP_CRIT_TERMS, 0, 2, dv + o_cr, sc, reinterpret_cast<const int*>(dv + o_gxt), g_t, 1, 256, e->stream);
    launch_zipped(OP_CRIT_REDUCE, 0, 2, dv + o_cr, sc, reinterpret_cast<const int*>(dv + o_gxr), 4, 1, (unsigned)kCritReduceThreads, e->stream);
    launch_zipped(OP_CRIT_FINAL, 0, 2, dv + o_cr, sc, reinterpret_cast<const int*>(dv + o_gxf), 1, 1, 1, e->stream);
    HIPCHK(hipGetLastError());
    return NEMGPU_OK;
}

int criteria(nemgpu_engine* e, float crit6[6], int buf = -1)
{
    int r;
    if ((r = criteria_enqueue(e, buf))) return r;
    HIPCHK(hipMemcpyAsync(crit6, e->crit6_dev, 6 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return NEMGPU_OK;
}

int reset_device(nemgpu_engine* e)
{
    if (e->have_params && e->parent == nullptr) {
        // (the initial block's tail, where nbobs_k is, stays zero)
        HIPCHK(hipMemcpyAsync(e->prop, e->prop0, sizeof(float) * e->par_words, hipMemcpyDeviceToDevice, e->stream));
    } else {
        if (e->have_params) {
            HIPCHK(hipMemcpyAsync(e->prop, e->prop0, sizeof(float) * e->k, hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->center, e->center0, sizeof(float) * e->k * e->d, hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->disp, e->disp0, sizeof(float) * e->k * e->d, hipMemcpyDeviceToDevice, e->stream));
        }
        HIPCHK(hipMemsetAsync(e->nbobs_k, 0, sizeof(float) * e->k, e->stream));
    }
    HIPCHK(hipMemsetAsync(e->sweep_next, 0, sizeof(int), e->stream));
    return clear_fault(e);
}

int flush_reset(nemgpu_engine* e)
{
    if (!e->reset_pending) return NEMGPU_OK;
    e->reset_pending = false;
    return reset_device(e);
}

// host half now; the device half (initial parameters back in place, counters cleared) before the next thing that
// looks at the device state -- or never, when that is a run whose first launch does it anyway (enqueue_init)
int reset_state(nemgpu_engine* e, bool lazy)
{
    if (lazy) e->reset_pending = true;
    else { e->reset_pending = false; int r0 = reset_device(e); if (r0) return r0; }
    e->tables_fresh = false;
    e->density_fresh = false;
    e->cur = 0; e->sweep_counter = 0;
    e->iters = 0; e->converged = 0; e->emptyk = 0; e->status = NEMGPU_OK;
    e->zero_density = 0; e->first_zero = -1; e->sweep_rounds = 0; e->masks_valid = false;
    e->draws = 0;                                                  // srandom(seed): the stream starts over
    return NEMGPU_OK;
}

void fill_result(nemgpu_engine* e, nemgpu_result* res)
{
    if (!res) return;
    res->status = e->status; res->iters = e->iters; res->converged = e->converged; res->emptyk = e->emptyk;
    res->zero_density_sites = e->zero_density; res->first_zero_density_site = e->first_zero;
    res->sweep_rounds = e->sweep_rounds;
    res->tie_draws = e->draws;
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" {

const char* nemgpu_last_error(void) { return g_last_error.c_str(); }

static const char kForkedMsg[] =
    "this process was forked from one that had already used the GPU: HIP is unusable in a forked child "
    "(start the workers with multiprocessing's 'spawn' or 'forkserver' method)";

int nemgpu_device_count(void)
{
    if (g_forked_after_hip.load()) return 0;                       // (no HIP call in such a child)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int nemgpu_default_device(int* device)
{
    if (!device) return NEMGPU_E_FUNCARG;
    *device = 0;
    std::call_once(g_atfork_once, [] { pthread_atfork(nullptr, nullptr, atfork_child); });
    if (g_forked_after_hip.load()) { set_error(kForkedMsg); return NEMGPU_E_DEVICE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        set_error("no usable HIP device: this library has no CPU fallback");
        return NEMGPU_E_DEVICE;
    }
    g_hip_used.store(true);
    const char* s = getenv("NEM_MI355X_DEVICE");
    if (s == nullptr || s[0] == '\0') {
        // unset: a launcher's LOCAL_RANK when there is one (one process per GPU), else the calling thread's current HIP
        // device -- the embedding application's choice.  A pool of workers that selects nothing lands on device 0:
        // NEM_MI355X_DEVICE=auto spreads it (INTEGRATION.md says so first thing in its multi-GPU paragraph)
        if (const char* lr = getenv("LOCAL_RANK")) {
            char* end = nullptr;
            const long v = strtol(lr, &end, 10);
            if (end != lr && *end == '\0' && v >= 0) { *device = (int)((unsigned long)v % (unsigned long)ndev); return NEMGPU_OK; }
        }
        int cur = 0;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = 0; }
        *device = cur;
        return NEMGPU_OK;
    }
    if (!strcmp(s, "auto")) {
        unsigned long key = (unsigned long)getpid();
        if (const char* lr = getenv("LOCAL_RANK")) {
            char* end = nullptr;
            const long v = strtol(lr, &end, 10);
            if (end != lr && *end == '\0' && v >= 0) key = (unsigned long)v;
        }
        *device = (int)(key % (unsigned long)ndev);
        return NEMGPU_OK;
    }
    char* end = nullptr;
    const long v = strtol(s, &end, 10);
    if (end == s || *end != '\0' || v < 0 || v >= ndev) {
        set_error(std::string("NEM_MI355X_DEVICE=") + s + ": not a device index below " + std::to_string(ndev) + " (or \"auto\")");
        return NEMGPU_E_ARG;
    }
    *device = (int)v;
    return NEMGPU_OK;
}

int nemgpu_create(nemgpu_engine** out, int n_total, int d, int k, int site_lo, int site_hi, int device,
                  void* hip_stream)
{
    if (!out) return NEMGPU_E_FUNCARG;
    *out = nullptr;
    if (n_total <= 0 || d <= 0 || k <= 0 || k > kMaxKernelK || site_lo < 0 || site_hi > n_total || site_lo >= site_hi) {
        set_error("nemgpu_create: bad sizes (need n,d > 0, 1 <= k <= 32, 0 <= lo < hi <= n)");
        return NEMGPU_E_ARG;
    }
    if (n_total > (1 << 24)) {
        // class sizes are float sums of 0/1 memberships in the reference (EstimSizes, nem_mod.c:1293-1315): exact
        // integers only up to 2^24, which is what the popcount M-step reproduces
        set_error("nemgpu_create: more than 2^24 families (the reference's float class sizes stop being integers)");
        return NEMGPU_E_ARG;
    }
    std::call_once(g_atfork_once, [] { pthread_atfork(nullptr, nullptr, atfork_child); });
    if (g_forked_after_hip.load()) { set_error(kForkedMsg); return NEMGPU_E_DEVICE; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no usable HIP device: this library has no CPU fallback");
        return NEMGPU_E_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("nemgpu_create: bad device index"); return NEMGPU_E_ARG; }
    g_hip_used.store(true);
    HIPCHK(hipSetDevice(device));
    nemgpu_engine* e = new nemgpu_engine();
    e->n_total = n_total; e->n_true = n_total; e->d = d; e->k = k; e->lo = site_lo; e->hi = site_hi; e->n = site_hi - site_lo;
    e->device = device;
    e->npad = (e->n + 255) / 256 * 256;
    e->dpad = (d + 63) / 64 * 64;
    e->W = e->dpad / 32;
    e->wf = (d + 31) / 32;
    e->nw64 = (e->n + 63) / 64;
    e->cfg.algo = NEMGPU_ALGO_NCEM; e->cfg.beta = 0.5f; e->cfg.disper = NEMGPU_DISP_K_; e->cfg.propor = NEMGPU_PROP_K;
    e->cfg.cvtest = NEMGPU_CV_CLAS; e->cfg.cvthres = 1e-8f; e->cfg.it_max = 100; e->cfg.param_fix = 0;
    e->cfg.tie_rule = NEMGPU_TIE_HASH; e->cfg.tie_seed = 0;
    if (const char* g = getenv("NEM_MI355X_GRAPHS")) e->use_graphs = (g[0] != '0');   // 0: plain launches only
    if (const char* g = getenv("NEM_MI355X_ROUNDS")) e->round_batch = std::max(2, std::min(kRoundBatchMax, atoi(g)));
    if (const char* g = getenv("NEM_MI355X_ROUNDS_ITER")) e->rounds_iter = std::max(2, std::min(e->round_batch, atoi(g)));
    e->rounds_iter = std::min(e->rounds_iter, e->round_batch);
    if (const char* g = getenv("NEM_MI355X_FUSED_SWEEP")) e->fused_sweep = (g[0] == '1');   // 1: the first rounds of a sweep in one launch
    if (const char* g = getenv("NEM_MI355X_FUSED_ROUNDS")) e->fused_rounds = std::max(2, std::min(kFusedMaxRounds, atoi(g)));
    if (const char* g = getenv("NEM_MI355X_FUZZY_CHAINS")) e->fuzzy_chains = std::max(0, std::min(2, atoi(g)));
    if (const char* g = getenv("NEM_MI355X_FAULT_INJECT")) e->fault_inject = !strcmp(g, "fuzzy_pc") ? 1 : 0;
    if (const char* g = getenv("NEM_MI355X_FF")) e->ff_mode = (g[0] == '0') ? 0 : (g[0] == '1') ? 1 : -1;   // 0 plain chain, 1 always
    if (const char* g = getenv("NEM_MI355X_SORT")) e->use_sort = (g[0] != '0');       // 0: E1 lanes in family order
    if (hip_stream) { e->stream = (hipStream_t)hip_stream; e->own_stream = false; }
    else {
        if (pool_stream_get(device, &e->stream) != hipSuccess) { delete e; set_error("hipStreamCreate failed"); return NEMGPU_E_DEVICE; }
        e->own_stream = true;
    }
    int r = NEMGPU_OK;
    alloc_for(e);
    auto A = [&](int rr) { if (r == NEMGPU_OK) r = rr; };
    A(dev_alloc(&e->xw, (size_t)e->W * e->npad));
    A(dev_alloc(&e->xws, (size_t)((e->W + 3) / 4) * 4 * e->npad));   // uint4[ceil(W/4)][npad]
    A(dev_alloc(&e->perm, (size_t)e->npad));
    A(dev_alloc(&e->xt, (size_t)d * e->nw64));
    A(alloc_model_buffers(e));
    if (r == NEMGPU_OK && !e->flags_host &&
        pool_get(device, true, e->flag_words() * sizeof(int), (char**)&e->flags_host, &e->flags_host_size) != hipSuccess) {
        set_error("hipHostMalloc failed"); r = NEMGPU_E_DEVICE;
    }
    if (r != NEMGPU_OK) { nemgpu_destroy(e); return r; }
    e->draw_ctl = e->sweep_next + 8;                               // two of the spare words ahead of the ticket counters
    *out = e;
    return NEMGPU_OK;
}

void rccl_release(nemgpu_engine* e);

static std::atomic<long long> g_destroy_ns[5];               // NEM_MI355X_BATCH_PROF: where nemgpu_destroy's time goes
void nemgpu_destroy(nemgpu_engine* e)
{
    if (!e) return;
    for (nemgpu_engine* c : e->clones) nemgpu_destroy(c);
    e->clones.clear();
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](int j) { auto t1 = std::chrono::steady_clock::now(); g_destroy_ns[j] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count(); t0 = t1; };
    (void)hipSetDevice(e->device);
    // (nemgpu_solve_many: an engine whose group never ran -- a failed job -- may still have uploads queued on its
    //  builder's stream, out of pinned blocks that go back to the pool below: wait for them)
    if (e->ready_pending && e->ready_ev) { (void)hipEventSynchronize(e->ready_ev); e->ready_pending = false; }
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    lap(0);
    std::vector<PoolBlock> blocks;
    blocks.push_back({false, e->clone_slab, e->clone_slab_size});
    blocks.push_back({true, (char*)e->clone_flags_host, e->clone_flags_size});
    blocks.push_back({true, e->rs_par_host, e->rs_par_host_size});
    e->clone_slab = nullptr; e->clone_flags_host = nullptr; e->rs_par_host = nullptr; e->rs_par_host_size = 0;
    rccl_release(e);
    drop_graphs(e);
    if (g_alloc_engine == e) g_alloc_engine = nullptr;
    lap(1);
    // (the stream is idle: everything the engine held goes back to the device's pool for the next engines)
    for (const nemgpu_engine::Staged& st : e->staging) blocks.push_back({true, st.p, st.size});
    e->staging.clear();
    if (e->host_bits_pin) blocks.push_back({true, (char*)e->host_bits, e->host_bits_pin});
    for (const nemgpu_engine::Chunk& c : e->chunks) if (c.owned) blocks.push_back({false, c.base, c.size});
    if (e->flags_host && !e->flags_host_borrowed) blocks.push_back({true, (char*)e->flags_host, e->flags_host_size});
    pool_put_many(e->device, blocks);
    zip_context_release(e);
    if (e->own_stream && e->stream) pool_stream_put(e->device, e->stream);
    lap(2);
    if (e->ev_flags) (void)hipEventDestroy(e->ev_flags);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->ready_ev) (void)hipEventDestroy(e->ready_ev);
    lap(3);
    delete e;
    lap(4);
}

void nemgpu_release_cached(void)
{
    for (int dev = 0; dev < kPoolDevices; dev++) {
        ResourcePool& P = g_pools[dev];
        std::vector<hipStream_t> streams;
        std::multimap<size_t, char*> devm, pin;
        std::vector<nemgpu_engine::ZipContext*> zips;
        { std::lock_guard<std::mutex> lock(P.m); zips.swap(P.zips); }
        if (!zips.empty()) (void)hipSetDevice(dev);
        for (nemgpu_engine::ZipContext* z : zips) zip_context_free(dev, z);      // (its blocks go through the pool ...)
        {
            std::lock_guard<std::mutex> lock(P.m);                                // (... which is emptied here)
            streams.swap(P.streams); devm.swap(P.dev); pin.swap(P.pinned);
            streams.insert(streams.end(), P.run_streams.begin(), P.run_streams.end()); P.run_streams.clear();
            P.dev_bytes = P.pinned_bytes = 0;
        }
        if (streams.empty() && devm.empty() && pin.empty()) continue;
        (void)hipSetDevice(dev);
        for (auto& kv : devm) (void)hipFree(kv.second);
        for (auto& kv : pin) (void)hipHostFree(kv.second);
        for (hipStream_t st : streams) (void)hipStreamDestroy(st);
    }
}

// room for the bit rows on the host: pinned (pooled) when they fit, else the engine's own vector
static uint32_t* host_bits_reserve(nemgpu_engine* e, size_t words)
{
    if (e->host_bits_pin) {
        (void)hipStreamSynchronize(e->stream);             // (a replaced matrix: its upload may still be reading them)
        pool_put(e->device, true, (char*)e->host_bits, e->host_bits_pin);
    }
    e->host_bits = nullptr; e->host_bits_pin = 0; e->host_bits_words = 0;
    // (behind the rows: room for the lanes' family order, npad ints -- upload_bits sends both in one copy)
    const size_t bytes = (std::max<size_t>(words, 1) + (size_t)e->npad) * sizeof(uint32_t);
    if (bytes <= kStageMax) {
        char* p = nullptr; size_t got = 0;
        if (pool_get(e->device, true, bytes, &p, &got) == hipSuccess) { e->host_bits = (uint32_t*)p; e->host_bits_pin = got; }
        else (void)hipGetLastError();
    }
    if (!e->host_bits) { e->host_bits_own.resize(words); e->host_bits = e->host_bits_own.data(); }
    else { std::vector<uint32_t>().swap(e->host_bits_own); }
    e->host_bits_words = words;
    return e->host_bits;
}

// bit rows (already in e->host_bits) -> device layouts.  pc: the rows' popcounts when the caller has them, else null.
// With pinned bit rows nothing here waits for the device: the copies and the layout kernels are ordered on the
// engine's stream ahead of everything that reads the layouts.
// Row popcounts of the family-major bit rows (the density kernels' lane order): with the CPU's popcnt instruction when
// it has one (20 000 x 500: 180 -> 45 us of a 0.36 ms engine build -- the builders are what bounds nemgpu_solve_many
// once the runs overlap them).  Host code only.
__attribute__((visibility("hidden"))) void nem_row_popcounts(const uint32_t* bits, int n, int wf, int* pc);
#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("popcnt"))) static void row_popcounts_hw(const uint32_t* bits, int n, int wf, int* pc)
{
    for (int i = 0; i < n; i++) {
        const uint32_t* row = bits + (size_t)i * wf;
        int c = 0, w = 0;
        for (; w + 2 <= wf; w += 2) { uint64_t v; memcpy(&v, row + w, 8); c += (int)__builtin_popcountll(v); }
        if (w < wf) c += __builtin_popcount(row[w]);
        pc[i] = c;
    }
}
static void row_popcounts_sw(const uint32_t* bits, int n, int wf, int* pc)
{
    for (int i = 0; i < n; i++) {
        const uint32_t* row = bits + (size_t)i * wf;
        int c = 0;
        for (int w = 0; w < wf; w++) c += __builtin_popcount(row[w]);
        pc[i] = c;
    }
}
void nem_row_popcounts(const uint32_t* bits, int n, int wf, int* pc)
{
    static const bool hw = __builtin_cpu_supports("popcnt") != 0;
    if (hw) row_popcounts_hw(bits, n, wf, pc); else row_popcounts_sw(bits, n, wf, pc);
}
#endif

static int upload_bits(nemgpu_engine* e, const int* pc_in)
{
    HIPCHK(hipSetDevice(e->device));
    const uint32_t* xbits_host = e->host_bits;
    uint32_t* xf = nullptr;
    size_t xf_size = 0;
    const size_t words = (size_t)e->n * e->wf;
    // lane order of the density kernels: inside each 256-family tile, families by popcount (stable)
    std::vector<int> perm_own;
    int* perm = e->host_bits_pin ? reinterpret_cast<int*>(e->host_bits + words) : nullptr;   // (the tail of the pinned block)
    const bool async = perm != nullptr;
    if (!perm) { perm_own.resize((size_t)e->npad); perm = perm_own.data(); }
    for (int i = 0; i < e->npad; i++) perm[i] = i;
    if (e->use_sort) {
        std::vector<int> pc_own;
        if (pc_in == nullptr) {
            pc_own.resize((size_t)e->n);
            nem_row_popcounts(xbits_host, e->n, e->wf, pc_own.data());
            pc_in = pc_own.data();
        }
        const int* pc = pc_in;
        // stable counting sort of every 256-family tile by popcount (0 .. d)
        std::vector<int> slot((size_t)e->d + 2);
        for (int t0 = 0; t0 < e->n; t0 += 256) {
            const int t1 = std::min(t0 + 256, e->n);
            int lo = e->d, hi = 0;
            for (int i = t0; i < t1; i++) { lo = std::min(lo, pc[i]); hi = std::max(hi, pc[i]); }
            std::fill(slot.begin() + lo, slot.begin() + hi + 2, 0);
            for (int i = t0; i < t1; i++) slot[pc[i] + 1]++;
            for (int c = lo + 1; c <= hi; c++) slot[c + 1] += slot[c];       // slot[c] = first position of popcount c
            for (int i = t0; i < t1; i++) perm[(size_t)t0 + slot[pc[i]]++] = i;
        }
    }
    // staging copy of the family-major bit rows: small ones live in the engine's chunk, large ones come and go
    const bool staged_in_chunk = words * sizeof(uint32_t) < kChunkOwn;
    if (staged_in_chunk) {
        if (!e->xf_stage) { alloc_for(e); int r = dev_alloc(&e->xf_stage, words + (size_t)e->npad); if (r) return r; }
        xf = e->xf_stage;
    } else {
        HIPCHK(pool_get(e->device, false, words * sizeof(uint32_t), (char**)&xf, &xf_size));
    }
    hipError_t err;
    if (async && staged_in_chunk) {
        // rows and lane order in ONE copy (host: the pinned block and its tail; device: the staging rows and theirs): a
        // copy's fixed cost on the copy engine -- 16 us median for the four copies of an engine in nemgpu_solve_many's
        // trace -- is what bounds that job
        int* dperm = reinterpret_cast<int*>(e->xf_stage + words);
        if (e->perm != dperm) { e->perm = dperm; drop_graphs(e); }      // (captured batches hold the old address)
        err = hipMemcpyAsync(xf, xbits_host, (words + (size_t)e->npad) * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream);
    } else {
        err = hipMemcpyAsync(xf, xbits_host, words * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream);
        if (err == hipSuccess)
            err = hipMemcpyAsync(e->perm, perm, (size_t)e->npad * sizeof(int), hipMemcpyHostToDevice, e->stream);
    }
    e->layout_pending = false;
    if (err == hipSuccess && e->defer_layout && async && staged_in_chunk) e->layout_pending = true;   // (run_many: pending_layout)
    else if (err == hipSuccess) {
        launch_layout(xf, e->n, e->wf, e->W, e->npad, e->d, e->nw64, e->xw, e->xt, e->perm, e->xws, e->stream);
        err = hipGetLastError();
    }
    if (err == hipSuccess && !(async && staged_in_chunk)) err = hipStreamSynchronize(e->stream);
    if (!staged_in_chunk) pool_put(e->device, false, (char*)xf, xf_size);
    if (err != hipSuccess) { set_error(std::string("matrix upload failed: ") + hipGetErrorString(err)); return NEMGPU_E_DEVICE; }
    e->have_matrix = true;
    return NEMGPU_OK;
}

int nemgpu_set_matrix_bits(nemgpu_engine* e, const uint32_t* xbits_host)
{
    if (!e || !xbits_host) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    const size_t words = (size_t)e->n * e->wf;
    uint32_t* bits = host_bits_reserve(e, words);
    memcpy(bits, xbits_host, words * sizeof(uint32_t));
    // bits above organism d-1 in a row's last word are not data: cleared here, so that a caller's dirty padding can
    // neither count as organisms in the popcount M-step nor push a row's popcount past d in the lane ordering
    if (e->d & 31) {
        const uint32_t keep = (1u << (e->d & 31)) - 1u;
        for (size_t i = 0; i < (size_t)e->n; i++) bits[i * e->wf + (e->wf - 1)] &= keep;
    }
    return upload_bits(e, nullptr);
}

// engines packing a byte matrix right now, in all threads of the process: a lone caller takes helper threads, callers
// that already run side by side (pangenomenem_amd.batch) pack their own rows
static std::atomic<int> g_packers{0};

int nemgpu_set_matrix_bytes(nemgpu_engine* e, const uint8_t* x_host)
{
    if (!e || !x_host) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    const int d = e->d, wf = e->wf;
    uint32_t* bits = host_bits_reserve(e, (size_t)e->n * wf);
    std::vector<int> pc((size_t)e->n);
    // 8 values per load (bit 0 of each byte -> one byte), a whole 32-organism word at a time; then the rows' popcounts
    // (the density kernels' lane order)
    auto pack_rows = [&](int r0, int r1, uint64_t* bad_out) {
        uint64_t bad = 0;
        for (int i = r0; i < r1; i++) {
            const uint8_t* row = x_host + (size_t)i * d;
            uint32_t* out = bits + (size_t)i * wf;
            int j = 0;
            for (; j + 32 <= d; j += 32) {
                uint64_t w[4];
                memcpy(w, row + j, 32);
                bad |= (w[0] | w[1]) | (w[2] | w[3]);
                uint32_t v = 0;
                for (int q = 0; q < 4; q++)
                    v |= (uint32_t)(((w[q] & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << (8 * q);
                out[j >> 5] = v;
            }
            if (j < d) {
                uint32_t v = 0;
                int b = 0;
                for (; j + 8 <= d; j += 8, b += 8) {
                    uint64_t w8;
                    memcpy(&w8, row + j, 8);
                    bad |= w8;
                    v |= (uint32_t)(((w8 & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << b;
                }
                for (; j < d; j++, b++) { bad |= row[j]; v |= (uint32_t)(row[j] & 1u) << b; }
                out[wf - 1] = v;
            }
        }
        nem_row_popcounts(bits + (size_t)r0 * wf, r1 - r0, wf, pc.data() + r0);
        *bad_out = bad;
    };
    // rows are independent: matrices of 4 MB and more are packed by up to 4 threads
    const size_t bytes = (size_t)e->n * d;
    const unsigned hw = std::thread::hardware_concurrency();
    const int others = g_packers.fetch_add(1);
    int nt = (int)std::max(1u, std::min({4u, hw ? hw : 1u, (unsigned)(bytes >> 21)}));
    if (others > 0) nt = 1;
    uint64_t badv[4] = {0, 0, 0, 0};
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; t++)
            th.emplace_back(pack_rows, (int)((long long)e->n * t / nt), (int)((long long)e->n * (t + 1) / nt), &badv[t]);
        pack_rows(0, (int)((long long)e->n / nt), &badv[0]);
        for (std::thread& x : th) x.join();
    }
    g_packers.fetch_sub(1);
    const uint64_t bad = badv[0] | badv[1] | badv[2] | badv[3];
    if (bad & 0xFEFEFEFEFEFEFEFEull) { e->host_bits_words = 0; set_error("presence/absence matrix must hold 0/1 only"); return NEMGPU_E_ARG; }
    return upload_bits(e, pc.data());
}

int nemgpu_set_graph(nemgpu_engine* e, const int32_t* ptr, const int32_t* idx, const float* w)
{
    if (!e || !ptr) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    const int nnz = ptr[e->n] - ptr[0];
    if (ptr[0] != 0 || nnz < 0) { set_error("graph: ptr[0] must be 0 and ptr non-decreasing"); return NEMGPU_E_ARG; }
    for (int i = 0; i < e->n; i++) if (ptr[i + 1] < ptr[i]) { set_error("graph: ptr not monotone"); return NEMGPU_E_ARG; }
    for (int t = 0; t < nnz; t++)
        if (idx[t] < 0 || idx[t] >= e->n_total) { set_error("graph: neighbour index out of range"); return NEMGPU_E_ARG; }
    e->nei_ptr = nullptr; e->nei_idx = nullptr; e->nei_w = nullptr;   // (a replaced graph's arrays stay in their chunk)
    int r;
    alloc_for(e);
    // ptr | idx | w in one device block and one pinned block of the engine's: one copy, nothing waits (sources too
    // large for a pinned block: three copies from the caller's memory and one wait)
    const size_t b_ptr = sizeof(int) * ((size_t)e->n + 1), b_nz = sizeof(int) * (size_t)nnz;
    const size_t o_idx = (b_ptr + 255) & ~(size_t)255, o_w = (o_idx + b_nz + 255) & ~(size_t)255, total = o_w + b_nz;
    char* dev = nullptr;
    if ((r = dev_alloc(&dev, total))) return r;
    e->nei_ptr = (int*)dev; e->nei_idx = (int*)(dev + o_idx); e->nei_w = (float*)(dev + o_w);
    char* st = stage(e, total);
    if (st) {
        memcpy(st, ptr, b_ptr);
        if (nnz > 0) { memcpy(st + o_idx, idx, b_nz); memcpy(st + o_w, w, b_nz); }
        HIPCHK(hipMemcpyAsync(dev, st, total, hipMemcpyHostToDevice, e->stream));
    } else {
        HIPCHK(hipMemcpyAsync(e->nei_ptr, ptr, b_ptr, hipMemcpyHostToDevice, e->stream));
        if (nnz > 0) {
            HIPCHK(hipMemcpyAsync(e->nei_idx, idx, b_nz, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->nei_w, w, b_nz, hipMemcpyHostToDevice, e->stream));
        }
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    e->nnz = nnz;
    e->has_graph = nnz > 0;
    {
        // how far a class context can reach: the largest row sum (what the fused sweep copies of its exp table)
        double top = 0.0;
        for (int i = 0; i < e->n; i++) {
            double sum = 0.0;
            for (int t = ptr[i]; t < ptr[i + 1]; t++) sum += std::fabs((double)w[t]);
            if (!(sum <= top)) top = sum;                          // (a NaN weight: the whole table)
        }
        e->exp_need = (top < (double)(kExpTabGlobal - 1)) ? std::max(1, (int)top + 2) : kExpTabGlobal;
    }
    drop_graphs(e);
    return NEMGPU_OK;
}

int nemgpu_set_params(nemgpu_engine* e, const float* prop, const float* center, const float* disp)
{
    if (!e || !prop || !center || !disp) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    const size_t kd = (size_t)e->k * e->d;
    float* st = e->parent == nullptr ? (float*)stage(e, sizeof(float) * e->par_words) : nullptr;
    if (st) {
        // one copy of the whole block of initial values (its gaps and its tail are zeros)
        memset(st, 0, sizeof(float) * e->par_words);
        memcpy(st, prop, sizeof(float) * e->k); memcpy(st + e->par_o_center, center, sizeof(float) * kd);
        memcpy(st + e->par_o_disp, disp, sizeof(float) * kd);
        HIPCHK(hipMemcpyAsync(e->prop0, st, sizeof(float) * e->par_words, hipMemcpyHostToDevice, e->stream));
    } else {
        HIPCHK(copy_sync(e, e->prop0, prop, sizeof(float) * e->k, hipMemcpyHostToDevice));
        HIPCHK(copy_sync(e, e->center0, center, sizeof(float) * kd, hipMemcpyHostToDevice));
        HIPCHK(copy_sync(e, e->disp0, disp, sizeof(float) * kd, hipMemcpyHostToDevice));
    }
    e->have_params = true;
    return reset_state(e, true);
}

int nemgpu_configure(nemgpu_engine* e, const nemgpu_config* cfg)
{
    if (!e || !cfg) return NEMGPU_E_FUNCARG;
    if (cfg->algo != NEMGPU_ALGO_NEM && cfg->algo != NEMGPU_ALGO_NCEM) { set_error("algo must be nem or ncem"); return NEMGPU_E_ARG; }
    if (cfg->disper < 0 || cfg->disper > 3 || cfg->propor < 0 || cfg->propor > 1) { set_error("bad dispersion/proportion model"); return NEMGPU_E_ARG; }
    if (cfg->cvtest < NEMGPU_CV_NONE || cfg->cvtest > NEMGPU_CV_CRIT_LOGGED) { set_error("convergence must be none, clas or crit"); return NEMGPU_E_ARG; }
    if (cfg->cvtest != NEMGPU_CV_NONE && !(cfg->cvthres > 0)) { set_error("convergence threshold must be > 0"); return NEMGPU_E_ARG; }
    if (cfg->it_max < 0) { set_error("it_max must be >= 0"); return NEMGPU_E_ARG; }
    if (cfg->tie_rule != NEMGPU_TIE_FIRST && cfg->tie_rule != NEMGPU_TIE_HASH && cfg->tie_rule != NEMGPU_TIE_LIBC) { set_error("bad tie rule"); return NEMGPU_E_ARG; }
    if (cfg->tie_seed != e->cfg.tie_seed || cfg->tie_rule != e->cfg.tie_rule) e->draw_valid = false;
    e->cfg = *cfg;
    HIPCHK(hipSetDevice(e->device));
    drop_graphs(e);                                                // kernel arguments are baked into captured batches
    { const int xr = ensure_exp_table(e); if (xr) return xr; }     // (the graph is normally set by now; loop_begin looks again)
    return reset_state(e, true);
}

int nemgpu_reset(nemgpu_engine* e)
{
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    return reset_state(e);
}

int nemgpu_init_partition(nemgpu_engine* e)
{
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    int r = init_partition(e);
    if (r) return r;
    e->crit_ref = 0.0f;
    if (e->cfg.cvtest == NEMGPU_CV_CRIT_LOGGED) {                  // (what the first iteration's criterion is compared with)
        float c6[6];
        if ((r = criteria(e, c6))) return r;
        e->crit_ref = c6[3];
    }
    return NEMGPU_OK;
}

int nemgpu_iterate(nemgpu_engine* e, int n_iters, nemgpu_result* res)
{
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    auto t0 = std::chrono::steady_clock::now();
    int r = iterate(e, n_iters);
    if (r) return r;
    HIPCHK(hipStreamSynchronize(e->stream));
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fill_result(e, res);
    if (res) res->loop_seconds = secs;
    return NEMGPU_OK;
}

// One EM iteration with everything a line of the reference's log needs -- the criteria of the partition the sweep
// started from and of the new one (WriteLogCrit, nem_alg.c:2361, 2398), the parameters (WriteLogClasses) -- in ONE
// stream submission and one wait: the two criteria passes and the copies are enqueued behind the iteration before
// anyone knows how it ended.  An iteration the host has to finish (more relaxation rounds) or a criterion-driven
// convergence test takes the plain sequence of calls instead.
// keep_draws: a start of RandNemAlgo -- the tie stream goes on where the start before left it (one srandom per nem() call)
static int iterate_logged_impl(nemgpu_engine* e, int with_init, bool keep_draws, nemgpu_result* res, float crit_before[6],
                               float crit_after[6], float* prop, float* center, float* disp, float* nbobs_k)
{
    if (!e || !crit_before || !crit_after) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipStreamSynchronize(e->stream));
    const auto t0 = std::chrono::steady_clock::now();
    int r;
    auto plain_tail = [&]() -> int {
        if (e->status == NEMGPU_W_EMPTYCLASS) return NEMGPU_OK;          // (the E-step did not run: nothing to log)
        int rr;
        if ((rr = criteria(e, crit_before, (e->cur + 2) % 3))) return rr;
        if ((rr = criteria(e, crit_after, -1))) return rr;
        return nemgpu_get_results(e, prop, center, disp, nbobs_k, nullptr);
    };
    bool speculated = false;
    char* st = nullptr; size_t got = 0;
    if (with_init && (crit_test(e) || e->libc() || !(e->lo == 0 && e->hi == e->n_total))) {
        // (the start whose two sweeps the host completes one after the other: the plain sequence of calls)
        const int draws = e->draws;
        if ((r = reset_state(e))) return r;
        if (keep_draws) e->draws = draws;
        if ((r = init_partition(e))) return r;
        e->crit_ref = 0.0f;
        if (e->cfg.cvtest == NEMGPU_CV_CRIT_LOGGED) {              // (what the first iteration's criterion is compared with)
            float c6[6];
            if ((r = criteria(e, c6))) return r;
            e->crit_ref = c6[3];
        }
    } else if (!crit_test(e) && e->lo == 0 && e->hi == e->n_total) {
        LoopCursor lc;
        if (with_init && !keep_draws) e->draws = 0;
        if ((r = loop_begin(e, lc, with_init ? 0 : 1, with_init != 0))) return r;
        if (loop_wants_batch(e, lc)) {
            if ((r = batch_plan(e, lc))) return r;
            // (a start: the blind sweep's partition is in buffer 1, the beta sweep's in 2)
            const int oldbuf = with_init ? 1 : lc.base, newbuf = with_init ? 2 : (lc.base + 1) % 3;
            const size_t words = 12 + e->par_words;
            if (pool_get(e->device, true, words * sizeof(float), &st, &got) != hipSuccess) { (void)hipGetLastError(); st = nullptr; }
            e->n_plain++;
            if ((r = batch_enqueue(e, lc, true))) { if (st) pool_put(e->device, true, st, got); return r; }
            if (st) {
                float* f = reinterpret_cast<float*>(st);
                r = criteria_pair_enqueue(e, newbuf, oldbuf);                  // both partitions in the same launches
                if (r == NEMGPU_OK) HIPCHK(hipMemcpyAsync(f, e->crit2_6, 6 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
                if (r == NEMGPU_OK) HIPCHK(hipMemcpyAsync(f + 6, e->crit6_dev, 6 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
                if (r == NEMGPU_OK) HIPCHK(hipMemcpyAsync(f + 12, e->prop, e->par_words * sizeof(float), hipMemcpyDeviceToHost, e->stream));
                if (r) { (void)hipStreamSynchronize(e->stream); pool_put(e->device, true, st, got); return r; }
            }
            HIPCHK(hipStreamSynchronize(e->stream));
            const bool clean = e->h_ctrl()[C_NEED_ROUNDS] == 0 && e->h_ctrl()[C_STATUS] != NEMGPU_W_EMPTYCLASS;
            if ((r = batch_finish(e, lc))) { if (st) pool_put(e->device, true, st, got); return r; }
            speculated = st != nullptr && clean;
        }
    } else {
        if ((r = iterate(e, 1))) return r;
    }
    if (speculated) {
        const float* f = reinterpret_cast<const float*>(st);
        memcpy(crit_before, f, 6 * sizeof(float));
        memcpy(crit_after, f + 6, 6 * sizeof(float));
        const size_t kd = (size_t)e->k * e->d;
        if (prop) memcpy(prop, f + 12, sizeof(float) * e->k);
        if (center) memcpy(center, f + 12 + e->par_o_center, sizeof(float) * kd);
        if (disp) memcpy(disp, f + 12 + e->par_o_disp, sizeof(float) * kd);
        if (nbobs_k) memcpy(nbobs_k, f + 12 + e->par_o_nb, sizeof(float) * e->k);
    }
    if (st) pool_put(e->device, true, st, got);
    if (!speculated && (r = plain_tail())) return r;
    HIPCHK(hipStreamSynchronize(e->stream));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    fill_result(e, res);
    if (res) res->loop_seconds = secs;
    return NEMGPU_OK;
}

int nemgpu_iterate_logged(nemgpu_engine* e, int with_init, nemgpu_result* res, float crit_before[6], float crit_after[6],
                          float* prop, float* center, float* disp, float* nbobs_k)
{
    return iterate_logged_impl(e, with_init, false, res, crit_before, crit_after, prop, center, disp, nbobs_k);
}

// ---- one run with the reference's per-iteration log, the iterations pipelined ------------------------------------------
// nemgpu_iterate_logged is one EM iteration, two criteria evaluations and a wait per call: 0.3 ms per logged iteration at
// configs[1] size, of which the criteria's i-ordered sums (four lone blocks per partition) are 0.1-0.2 and the iteration
// itself 0.04.  Here a batch of up to seven iterations runs as the unlogged loop runs it (device-side loop control, one
// wait), and behind every iteration what its log line needs is set aside in a twin of the engine -- the partition the
// E-step started from, the one it left, the densities, the parameters (0.75 MB of device copies) -- so that the criteria
// of ALL the batch's iterations are one lock-step pass over the twins afterwards: their lone blocks side by side.
static int ensure_clones(nemgpu_engine* e, int count);

int nemgpu_run_logged(nemgpu_engine* e, nemgpu_result* res, nemgpu_log_fn fn, void* user)
{
    if (!e || !fn) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    int r;
    const int k = e->k, d = e->d;
    const size_t kd = (size_t)k * d;
    std::vector<float> prop((size_t)k), center(kd), disp(kd), nb((size_t)k);
    float cb[6], ca[6];
    nemgpu_result r1{};
    auto emit_line = [&](int iter, const float* b, const float* a, const float* p, const float* c, const float* s, const float* sizes) {
        nemgpu_log_event ev{};
        ev.kind = NEMGPU_LOG_LINE; ev.start = 0; ev.iter = iter; ev.crit_before = b; ev.crit_after = a;
        ev.prop = p; ev.center = c; ev.disp = s; ev.nbobs_k = sizes;
        fn(&ev, user);
    };
    const auto t0 = std::chrono::steady_clock::now();
    static const bool prof = getenv("NEM_MI355X_BATCH_PROF") != nullptr;
    auto lapt = t0;
    auto lap = [&](const char* what) { if (prof) { const auto now = std::chrono::steady_clock::now(); fprintf(stderr, "[logged run] %s: %.0f us\n", what, std::chrono::duration<double, std::micro>(now - lapt).count()); lapt = now; } };
    // ---- line 0: the start (reset, the two initial sweeps), both criteria, the parameters
    if ((r = iterate_logged_impl(e, 1, false, &r1, cb, ca, prop.data(), center.data(), disp.data(), nb.data()))) return r;
    emit_line(0, cb, ca, prop.data(), center.data(), disp.data(), nb.data());
    lap("line 0 (start, two criteria)");
    static const bool batched_on = !(getenv("NEM_MI355X_LOG_BATCHED") && getenv("NEM_MI355X_LOG_BATCHED")[0] == '0');
    const bool batched = batched_on && e->ncem() && !crit_test(e) && e->lo == 0 && e->hi == e->n_total && e->parent == nullptr;
    bool last_logged = false;
    if (!batched) {
        // (the fuzzy algorithm, the `crit` test, a sharded engine: one logged iteration per call, as nemgpu_iterate_logged)
        for (int iter = 1; iter <= e->cfg.it_max && !e->converged && e->status == NEMGPU_OK; iter++) {
            if ((r = iterate_logged_impl(e, 0, false, &r1, cb, ca, prop.data(), center.data(), disp.data(), nb.data()))) return r;
            if (e->status == NEMGPU_W_EMPTYCLASS) {
                nemgpu_log_event ev{}; ev.kind = NEMGPU_LOG_EMPTY; ev.iter = iter; ev.emptyk = e->emptyk; fn(&ev, user);
                last_logged = false;
                break;
            }
            emit_line(iter, cb, ca, prop.data(), center.data(), disp.data(), nb.data());
            last_logged = true;
        }
    } else {
        // two twins per iteration of a batch: the partition the E-step started from and the one it left, each with the
        // iteration's densities -- 14 criteria evaluations in one pass
        if ((r = ensure_clones(e, 2 * kPipeDepth))) return r;
        for (nemgpu_engine* c : e->clones) { c->cfg = e->cfg; c->stream = e->stream; }
        std::vector<nemgpu_engine*> T(e->clones.begin(), e->clones.begin() + 2 * kPipeDepth);
        const size_t W = 12 + e->par_words;
        char* st_dev = nullptr; char* st_host = nullptr; size_t dev_size = 0, host_size = 0;
        HIPCHK(pool_get(e->device, false, (size_t)kPipeDepth * W * sizeof(float), &st_dev, &dev_size));
        if (pool_get(e->device, true, (size_t)kPipeDepth * W * sizeof(float), &st_host, &host_size) != hipSuccess) {
            pool_put(e->device, false, st_dev, dev_size);
            set_error("no pinned memory for the log lines"); return NEMGPU_E_DEVICE;
        }
        struct Back { int dev; char* a; size_t as; char* b; size_t bs; hipStream_t s; ~Back() { (void)hipStreamSynchronize(s); pool_put(dev, false, a, as); pool_put(dev, true, b, bs); } }
            back{e->device, st_dev, dev_size, st_host, host_size, e->stream};
        std::vector<Recorder> recs((size_t)2 * kPipeDepth);
        lap("twins and staging");
        int iter = 0;
        while (iter < e->cfg.it_max && !e->converged && e->status == NEMGPU_OK) {
            LoopCursor lc;
            // (a run that is through in one or two iterations -- chunks near their fixed point -- should not pay for seven
            //  enqueued ones: three first, then seven at a time)
            const int g = std::min(e->cfg.it_max - iter, iter == 0 ? 3 : kPipeDepth);
            if ((r = loop_begin(e, lc, g, false))) return r;
            if ((r = batch_plan(e, lc))) return r;
            const int base = lc.base;
            // behind iteration j: the partition it started from and the one it left, the densities, the parameters
            const std::function<int(int)> set_aside = [&](int j) -> int {
                const int P = (base + j) % 3, Q = (base + j + 1) % 3;
                CopySegsArgs cs{};
                int q = 0;
                auto seg = [&](const void* src, void* dst, size_t bytes) {
                    cs.src[q] = static_cast<const int*>(src); cs.dst[q] = static_cast<int*>(dst); cs.words[q] = (int)((bytes + 3) / 4); q++;
                };
                for (int side = 0; side < 2; side++) {
                    nemgpu_engine* c = T[(size_t)(2 * j + side)];
                    seg(e->lab[side ? Q : P], c->lab[0], (size_t)e->n_total);
                    seg(e->pkfki, c->pkfki, sizeof(double) * (size_t)k * e->npad);
                    seg(e->logpkfki, c->logpkfki, sizeof(float) * (size_t)k * e->npad);
                }
                seg(e->prop, reinterpret_cast<float*>(st_dev) + (size_t)j * W + 12, sizeof(float) * e->par_words);
                cs.n = q;
                launch_copy_segments(cs, e->stream);                // (one launch: seven copy commands were 28 us of issue time per iteration)
                HIPCHK(hipGetLastError());
                return NEMGPU_OK;
            };
            e->n_plain++;
            if ((r = batch_enqueue(e, lc, true, &set_aside))) return r;
            HIPCHK(hipStreamSynchronize(e->stream));
            const int* hc = e->h_ctrl();
            const int done = hc[C_ITERS];
            const bool tail_open = hc[C_NEED_ROUNDS] != 0 || hc[C_STATUS] == NEMGPU_W_EMPTYCLASS;   // the last counted iteration did not end in the pipeline
            if ((r = batch_finish(e, lc))) return r;
            if (prof) fprintf(stderr, "[logged run]   g %d done %d need_rounds %d status %d converged %d deep %d\n", g, done, hc[C_NEED_ROUNDS], hc[C_STATUS], hc[C_CONVERGED], lc.deep);
            lap("a batch of iterations");
            const int whole = tail_open ? std::max(0, done - 1) : done;
            if (whole > 0) {
                // the criteria of the `whole` iterations in one lock-step pass over the twins
                std::vector<nemgpu_engine*> TT(T.begin(), T.begin() + 2 * whole);
                std::vector<int> all((size_t)2 * whole);
                for (int m = 0; m < 2 * whole; m++) all[(size_t)m] = m;
                if ((r = lockstep(TT, all, recs, [&](int m) {
                        nemgpu_engine* c = TT[(size_t)m];
                        int* slot = reinterpret_cast<int*>(st_dev) + (size_t)(m / 2) * W + (m % 2) * 6;
                        const int rr = criteria_enqueue(c, 0);
                        if (rr == NEMGPU_OK) launch_copy_words(reinterpret_cast<const int*>(c->crit6_dev), slot, 6, e->stream);
                        return rr;
                    }, false))) return r;
                HIPCHK(hipMemcpyAsync(st_host, st_dev, (size_t)whole * W * sizeof(float), hipMemcpyDeviceToHost, e->stream));
                HIPCHK(hipStreamSynchronize(e->stream));
                for (int m = 0; m < whole; m++) {
                    const float* f = reinterpret_cast<const float*>(st_host) + (size_t)m * W;
                    emit_line(iter + m + 1, f, f + 6, f + 12, f + 12 + e->par_o_center, f + 12 + e->par_o_disp, f + 12 + e->par_o_nb);
                    if (m == whole - 1) memcpy(ca, f + 6, sizeof ca);
                }
                last_logged = true;
                lap("its criteria and lines");
            }
            iter += whole;
            if (tail_open && done > 0) {
                iter += 1;
                if (e->status == NEMGPU_W_EMPTYCLASS) {
                    nemgpu_log_event ev{}; ev.kind = NEMGPU_LOG_EMPTY; ev.iter = iter; ev.emptyk = e->emptyk; fn(&ev, user);
                    last_logged = false;
                    break;
                }
                // (its sweep was finished from the host: the plain sequence of calls)
                if ((r = criteria(e, cb, (e->cur + 2) % 3)) || (r = criteria(e, ca, -1))) return r;
                if ((r = nemgpu_get_results(e, prop.data(), center.data(), disp.data(), nb.data(), nullptr))) return r;
                emit_line(iter, cb, ca, prop.data(), center.data(), disp.data(), nb.data());
                last_logged = true;
            }
            if (done == 0) break;                                  // (nothing ran: the loop tests above end it)
        }
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    fill_result(e, res);
    if (res) {
        res->loop_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        // (the last logged iteration's second criteria evaluation was of the final partition, on the final densities)
        if (last_logged && e->iters > 0) memcpy(res->crit, ca, sizeof ca);
        else res->crit[0] = NAN;                                   // the caller asks nemgpu_criteria (after its own iters == 0 steps)
    }
    return NEMGPU_OK;
}


int nemgpu_restart_iterate(nemgpu_engine* e, int n_iters, nemgpu_result* res)
{
    // reset + ComputePartitionFromPara(Needinit=1) + up to n_iters EM iterations as ONE pipelined batch sequence
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    e->draws = 0;
    int r = iterate(e, n_iters, true);
    if (r) return r;
    fill_result(e, res);
    return NEMGPU_OK;
}

int nemgpu_run(nemgpu_engine* e, nemgpu_result* res)
{
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    int r;
    HIPCHK(hipStreamSynchronize(e->stream));
    auto t0 = std::chrono::steady_clock::now();
    e->draws = 0;                                                  // one nem() call = one srandom(seed), nem_exe.c:621
    if ((r = iterate(e, e->cfg.it_max, true))) return r;           // restart + initial sweeps + EM loop, pipelined
    HIPCHK(hipStreamSynchronize(e->stream));
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (e->iters == 0) {                                           // nem_alg.c:1845-1851
        if ((r = do_mstep(e))) return r;
        if (!e->ncem() && (r = read_iter_flags(e))) return r;      // (a fuzzy M-step can report a fault)
        if ((r = do_tables(e))) return r;
        if ((r = do_density(e))) return r;
    }
    fill_result(e, res);
    if (res) {
        res->loop_seconds = secs;
        if ((r = criteria(e, res->crit))) return r;                // nem_alg.c:1852
    }
    return NEMGPU_OK;
}

// Several whole runs in lock step: one launch per step of the EM for ALL of them (see "Lock-step batches" above).
int nemgpu_run_many(nemgpu_engine** engines, int count, nemgpu_result* results)
{
    if (!engines || count <= 0) return NEMGPU_E_FUNCARG;
    std::vector<nemgpu_engine*> E((size_t)count);
    for (int i = 0; i < count; i++) {
        if (!engines[i]) return NEMGPU_E_FUNCARG;
        for (int j = 0; j < i; j++) if (engines[j] == engines[i]) { set_error("nemgpu_run_many: the same engine twice"); return NEMGPU_E_ARG; }
        E[i] = engines[i];
    }
    HIPCHK(hipSetDevice(E[0]->device));
    return run_many(E, results);
}

// INIT_RANDOM: RandNemAlgo (nem_alg.c:1574-1742).  InitPara's whole-sample dispersion (:1200-1281) is one M-step
// with every family in class 0; every start draws its centres from the data (MakeRandomPara, :1381-1473), runs the
// INIT_PARAM_FILE pipeline on them and is ranked by criterion M (DEFAULT_CRIT, nem_typ.h:80); the best partition is
// restored and EstimPara run on it (:1703-1713).
constexpr int kStartsNotInLockStep = -7001;   // run_random_lockstep, logged: a start drew behind its initial sweeps -- nothing was handed to the writer, the caller runs the starts one after the other
static int run_random_lockstep(nemgpu_engine* e, int n_starts, uint32_t seed, nemgpu_result* res, int* best_start, nemgpu_log_fn fn = nullptr, void* user = nullptr);

// Many whole problems, from host arrays to host arrays, in one call: `workers` threads of the library build the
// engines (bit packing, uploads out of pinned blocks -- nothing waits for the device), the calling thread runs every
// `group` of them in lock step as soon as it is complete (nemgpu_run_many), the workers fetch the results and recycle
// the engines while later groups are being built.  What PPanGGOLiN's chunk loop is when its chunks are arrays.
static int solve_many_one(nemgpu_problem* P, int count, const nemgpu_config* cfg, int device, int workers, int group, const ChunkSource* src = nullptr);
static int adopt_chunk(nemgpu_engine* e, const nemgpu_master* M, const nemk::ChunkPlan& plan, int nnz_c);
static thread_local bool tl_runner = false;                  // this thread is one of nemgpu_solve_many_devices' runners

// Four groups and more, six workers and more: two runners on the device (nemgpu_solve_many_devices with the device named
// twice), each with half of the workers.  Between the steps of a group its runner reads flags and records the next step
// -- 0.1-0.3 ms in which its stream is empty; the other runner's group fills them (256 problems of 20 000 x 500,
// 12 workers: 8 400 -> 9 900 problems/s).  NEM_MI355X_RUNNERS=n fixes the number (1: never split).
int nemgpu_solve_many(nemgpu_problem* P, int count, const nemgpu_config* cfg, int device, int workers, int group)
{
    if (!P || count <= 0 || !cfg) return NEMGPU_E_FUNCARG;
    static const int fixed = [] { const char* g = getenv("NEM_MI355X_RUNNERS"); return g ? std::max(0, std::min(atoi(g), 4)) : 0; }();
    const int g = std::max(1, std::min(group, 256));
    const int runners = tl_runner ? 1 : fixed > 0 ? fixed : (count >= 4 * g && workers >= 6 ? 2 : 1);
    if (runners <= 1 || count <= g) return solve_many_one(P, count, cfg, device, workers, group);
    const int devs[4] = {device, device, device, device};
    return nemgpu_solve_many_devices(P, count, cfg, devs, runners, workers, group);
}

static int solve_many_one(nemgpu_problem* P, int count, const nemgpu_config* cfg, int device, int workers, int group, const ChunkSource* src)
{
    if (!P || count <= 0 || !cfg) return NEMGPU_E_FUNCARG;
    workers = std::max(1, std::min(workers, 64));
    group = std::max(1, std::min(group, 256));
    for (int i = 0; i < count; i++) {
        nemgpu_problem& q = P[i];
        q.rc = NEMGPU_OK; q.result = nemgpu_result{};
        if (q.n <= 0 || q.d <= 0 || q.k <= 0 || (src == nullptr && (!q.x_bytes == !q.x_bits)) || !q.prop || !q.center || !q.disp ||
            (q.nei_ptr && q.nei_ptr[q.n] > 0 && (!q.nei_idx || !q.nei_w))) {
            set_error("nemgpu_solve_many: problem " + std::to_string(i) + " is incomplete (sizes, exactly one of x_bytes / x_bits, parameters)");
            return NEMGPU_E_FUNCARG;
        }
    }
    const int G = (count + group - 1) / group;
    std::vector<nemgpu_engine*> eng((size_t)count, nullptr);
    std::vector<std::string> errs((size_t)count);
    std::vector<int> built((size_t)G, 0);
    std::mutex m;
    std::condition_variable cv;
    std::deque<int> fetchq;
    struct GroupSlab { char* host = nullptr; size_t size = 0, stride = 0; int left = 0; };
    std::vector<GroupSlab> gslab((size_t)G);                     // a group's results on the host (GroupFetch), until its last fetch
    std::vector<int> gslot((size_t)count, -1);
    int build_next = 0, build_limit = std::min(count, 2 * group), fetched = 0;
    bool quit = false;

    // Streams are the expensive resource of this stack (6 ms to create one): the engines a worker builds share that
    // worker's stream, the lock-step runs have one of their own, so that a group's run and the next groups' uploads
    // are not ordered behind each other.
    std::vector<hipStream_t> wstream((size_t)workers, nullptr);
    hipStream_t rstream = nullptr, fstream = nullptr;
    (void)hipSetDevice(device);
    if (pool_run_stream_get(device, &rstream) != hipSuccess) { set_error("hipStreamCreate failed"); return NEMGPU_E_DEVICE; }
    if (pool_stream_get(device, &fstream) != hipSuccess) { (void)hipGetLastError(); fstream = nullptr; }
    const bool prof = getenv("NEM_MI355X_BATCH_PROF") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double>(now() - t0).count(); };
    double slow_build[6] = {0, 0, 0, 0, 0, 0}, slow_fetch[3] = {0, 0, 0};      // (profile: the slowest build / fetch, by stage)
    std::mutex pm;
    auto build = [&](int i, hipStream_t st) {
        nemgpu_problem& q = P[i];
        nemgpu_engine* e = nullptr;
        double t[6] = {0, 0, 0, 0, 0, 0};
        auto t0 = now();
        int r = nemgpu_create(&e, q.n, q.d, q.k, 0, q.n, device, st);
        if (r == NEMGPU_OK && st != nullptr) e->defer_layout = true;
        t[1] = since(t0);
        if (src != nullptr) {
            // a chunk of a master that lives on the device: matrix rows, lane order and graph are made THERE, into this
            // engine's own buffers (nem_chunks.hip) -- nothing of them crosses PCIe
            if (r == NEMGPU_OK) r = adopt_chunk(e, src->master, src->plans[i], src->nnz[i]);
            t[2] = t[3] = since(t0);
        } else {
        if (r == NEMGPU_OK) r = q.x_bits ? nemgpu_set_matrix_bits(e, q.x_bits) : nemgpu_set_matrix_bytes(e, q.x_bytes);
        t[2] = since(t0);
        if (r == NEMGPU_OK) {
            if (q.nei_ptr) r = nemgpu_set_graph(e, q.nei_ptr, q.nei_idx, q.nei_w);
            else { std::vector<int32_t> z((size_t)q.n + 1, 0); r = nemgpu_set_graph(e, z.data(), nullptr, nullptr); }
        }
        t[3] = since(t0);
        }
        if (r == NEMGPU_OK) r = nemgpu_set_params(e, q.prop, q.center, q.disp);
        if (r == NEMGPU_OK) r = nemgpu_configure(e, cfg);
        t[4] = since(t0);
        if (r == NEMGPU_OK && st != nullptr) {
            // the builder's stream goes on to other engines' uploads: the run waits for THIS engine's, by event
            if (hipEventCreateWithFlags(&e->ready_ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(e->ready_ev, st) == hipSuccess)
                e->ready_pending = true;
            else (void)hipGetLastError();                          // (no event: the run waits for the stream, as before)
        }
        t[5] = since(t0);
        if (prof) { std::lock_guard<std::mutex> lock(pm); if (t[5] > slow_build[5]) for (int j = 1; j < 6; j++) slow_build[j] = t[j]; slow_build[0] += t[5]; }
        if (r != NEMGPU_OK) { errs[(size_t)i] = g_last_error; if (e) nemgpu_destroy(e); e = nullptr; }
        q.rc = r;
        eng[(size_t)i] = e;
    };
    auto fetch = [&](int i) {
        nemgpu_problem& q = P[i];
        nemgpu_engine* e = eng[(size_t)i];
        if (!e) return;
        auto t0 = now();
        // (its run is over and was waited for: whatever it still copies goes to the fetch stream, not behind the uploads
        // its builder's stream has taken in the meantime)
        if (!e->own_stream && fstream != nullptr) e->stream = fstream;
        GroupSlab& gs = gslab[(size_t)(i / group)];
        uint8_t* lab_out = (src != nullptr && src->labels != nullptr) ? src->labels[i] : nullptr;
        if (q.rc == NEMGPU_OK && lab_out != nullptr && e->ncem()) {        // (chunks: the class of every kept family)
            if (gs.host != nullptr && gslot[(size_t)i] >= 0) {
                const uint8_t* lab = (const uint8_t*)(gs.host + gs.stride * (size_t)gslot[(size_t)i]);
                for (int f = 0; f < e->n; f++) lab_out[f] = lab[f] & 0x7F;
            } else {
                q.rc = nemgpu_get_labels(e, lab_out);
                if (q.rc != NEMGPU_OK) errs[(size_t)i] = g_last_error;
            }
        }
        if (q.rc == NEMGPU_OK && (q.out_prop || q.out_center || q.out_disp || q.out_nbobs_k || q.out_c)) {
            if (gs.host != nullptr && gslot[(size_t)i] >= 0)
                result_unpack(e, gs.host + gs.stride * (size_t)gslot[(size_t)i], result_part_bytes(e), q.out_prop, q.out_center, q.out_disp, q.out_nbobs_k, q.out_c);
            else {
                q.rc = nemgpu_get_results(e, q.out_prop, q.out_center, q.out_disp, q.out_nbobs_k, q.out_c);
                if (q.rc != NEMGPU_OK) errs[(size_t)i] = g_last_error;
            }
        }
        const double t1 = since(t0);
        nemgpu_destroy(e);
        const double t2 = since(t0);
        if (prof) { std::lock_guard<std::mutex> lock(pm); if (t2 > slow_fetch[2]) { slow_fetch[1] = t1; slow_fetch[2] = t2; } slow_fetch[0] += t2; }
        eng[(size_t)i] = nullptr;
    };
    auto worker = [&](int w) {
        (void)hipSetDevice(device);
        (void)pool_stream_get(device, &wstream[(size_t)w]);        // (nullptr: the engines take streams of their own)
        std::unique_lock<std::mutex> lock(m);
        for (;;) {
            cv.wait(lock, [&] { return quit || !fetchq.empty() || build_next < build_limit; });
            if (!fetchq.empty()) {                                 // results first: they free engines for the builders
                const int i = fetchq.front(); fetchq.pop_front();
                lock.unlock(); fetch(i); lock.lock();
                GroupSlab& gs = gslab[(size_t)(i / group)];
                if (gs.host != nullptr && --gs.left == 0) { pool_put(device, true, gs.host, gs.size); gs.host = nullptr; }
                fetched++;
                cv.notify_all();
            } else if (build_next < build_limit) {
                const int i = build_next++;
                lock.unlock(); build(i, wstream[(size_t)w]); lock.lock();
                built[(size_t)(i / group)]++;
                cv.notify_all();
            } else if (quit) return;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < workers; t++) pool.emplace_back(worker, t);
    int rc = NEMGPU_OK;
    std::string first_err;
    double t_wait = 0, t_run = 0, t_tail = 0;
    for (int g = 0; g < G; g++) {
        const int i0 = g * group, i1 = std::min(count, i0 + group);
        {
            const auto t0 = now();
            std::unique_lock<std::mutex> lock(m);
            cv.wait(lock, [&] { return built[(size_t)g] == i1 - i0; });
            t_wait += since(t0);
        }
        std::vector<nemgpu_engine*> E;
        std::vector<int> who;
        for (int i = i0; i < i1; i++) if (eng[(size_t)i]) { E.push_back(eng[(size_t)i]); who.push_back(i); }
        if (!E.empty()) {
            std::vector<nemgpu_result> R(E.size());
            (void)hipSetDevice(device);
            const auto t0 = now();
            // (the lead's own stream is a worker's: what it still has to upload is waited for, then the run moves)
            hipStream_t lead_own = E[0]->stream;
            int r = E[0]->ready_pending || hipStreamSynchronize(lead_own) == hipSuccess ? NEMGPU_OK : NEMGPU_E_DEVICE;
            if (r == NEMGPU_OK) {
                if (!E[0]->own_stream) E[0]->stream = rstream;
                // the group's partitions and parameters come back with the run, in one block
                GroupFetch gf;
                size_t dev_size = 0, host_size = 0;
                for (nemgpu_engine* e : E) gf.stride = std::max(gf.stride, (result_block_bytes(e) + 63) & ~(size_t)63);
                const size_t total = gf.stride * E.size();
                bool want = false;
                for (int i : who) want = want || P[i].out_prop || P[i].out_center || P[i].out_disp || P[i].out_nbobs_k || P[i].out_c ||
                                         (src != nullptr && src->labels != nullptr && src->labels[i] != nullptr);
                if (want && total <= kStageMax) {
                    if (pool_get(device, false, total, &gf.dev, &dev_size) != hipSuccess) { (void)hipGetLastError(); gf.dev = nullptr; }
                    if (gf.dev && pool_get(device, true, total, &gf.host, &host_size) != hipSuccess) { (void)hipGetLastError(); gf.host = nullptr; }
                }
                const bool grouped = gf.dev != nullptr && gf.host != nullptr;
                r = run_many(E, R.data(), grouped ? &gf : nullptr);
                if (!E[0]->own_stream) { (void)hipStreamSynchronize(rstream); E[0]->stream = lead_own; }
                if (gf.dev) pool_put(device, false, gf.dev, dev_size);
                if (grouped && r == NEMGPU_OK && gf.filled) {
                    std::lock_guard<std::mutex> lock(m);
                    GroupSlab& gs = gslab[(size_t)g];
                    gs.host = gf.host; gs.size = host_size; gs.stride = gf.stride; gs.left = (int)E.size();
                    for (size_t j = 0; j < E.size(); j++) gslot[(size_t)who[j]] = (int)j;
                } else if (gf.host) pool_put(device, true, gf.host, host_size);
            }
            t_run += since(t0);
            for (size_t j = 0; j < E.size(); j++) { P[who[j]].result = R[j]; if (r != NEMGPU_OK) P[who[j]].rc = r; }
            if (r != NEMGPU_OK && rc == NEMGPU_OK) { rc = r; first_err = g_last_error; }
        }
        {
            std::lock_guard<std::mutex> lock(m);
            for (int i = i0; i < i1; i++) {
                if (eng[(size_t)i]) fetchq.push_back(i);
                else fetched++;
            }
            build_limit = std::min(count, (g + 3) * group);
        }
        cv.notify_all();
    }
    {
        const auto t0 = now();
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return fetched == count; });
        quit = true;
        t_tail = since(t0);
    }
    if (prof) {
        fprintf(stderr, "[destroy] stream wait %.2f ms, graphs %.2f, pool %.2f, events %.2f, delete %.2f (all engines so far)\n", g_destroy_ns[0] * 1e-6,
                g_destroy_ns[1] * 1e-6, g_destroy_ns[2] * 1e-6, g_destroy_ns[3] * 1e-6, g_destroy_ns[4] * 1e-6);
        for (auto& a : g_destroy_ns) a = 0;
        fprintf(stderr, "[pool] idle: device %.0f MB, pinned %.0f MB; since the last report: allocated %lld device / %lld pinned blocks, freed %lld / %lld\n",
                g_pools[device].dev_bytes / 1048576.0, g_pools[device].pinned_bytes / 1048576.0, (long long)g_pool_miss[0], (long long)g_pool_miss[1],
                (long long)g_pool_spill[0], (long long)g_pool_spill[1]);
        for (int j = 0; j < 2; j++) { g_pool_miss[j] = 0; g_pool_spill[j] = 0; }
    }
    if (prof)
        fprintf(stderr, "[solve_many] %d problems, %d workers, groups of %d: waited for builds %.2f ms, lock-step runs %.2f ms, "
                        "waited for the last results %.2f ms\n"
                        "             builds %.2f ms in all, the slowest: create %.0f us, matrix %.0f, graph %.0f, parameters %.0f, event %.0f; "
                        "fetches %.2f ms in all, the slowest: results %.0f us, destroy %.0f\n",
                count, workers, group, t_wait * 1e3, t_run * 1e3, t_tail * 1e3, slow_build[0] * 1e3, slow_build[1] * 1e6,
                (slow_build[2] - slow_build[1]) * 1e6, (slow_build[3] - slow_build[2]) * 1e6, (slow_build[4] - slow_build[3]) * 1e6,
                (slow_build[5] - slow_build[4]) * 1e6, slow_fetch[0] * 1e3, slow_fetch[1] * 1e6, (slow_fetch[2] - slow_fetch[1]) * 1e6);
    cv.notify_all();
    for (std::thread& t : pool) t.join();
    for (hipStream_t st : wstream) if (st) { (void)hipStreamSynchronize(st); pool_stream_put(device, st); }
    (void)hipStreamSynchronize(rstream);
    pool_run_stream_put(device, rstream);
    if (fstream) { (void)hipStreamSynchronize(fstream); pool_stream_put(device, fstream); }
    for (int i = 0; i < count; i++)
        if (P[i].rc != NEMGPU_OK && rc == NEMGPU_OK) { rc = P[i].rc; first_err = errs[(size_t)i]; }
    if (rc != NEMGPU_OK && !first_err.empty()) set_error(first_err);
    return rc;
}


// ============================================================================================
// Chunks of a master pangenome, formed on the device (nem_chunks.hpp / nem_chunks.hip): the samples of partition()'s
// voting loop (ppanggolin.py:1045-1086), each the input files of __write_nem_input_files (ppanggolin.py:821-930) for
// that sample -- without the files, and without a matrix or a graph crossing PCIe per sample.
// ============================================================================================
int nemgpu_master_create(nemgpu_master** out, int device, int n, int d, const uint32_t* xbits, const int32_t* nei_ptr,
                         const int32_t* nei_idx, const uint32_t* edge_bits)
{
    if (!out) return NEMGPU_E_FUNCARG;
    *out = nullptr;
    if (n <= 0 || d <= 0 || !xbits || !nei_ptr) { set_error("nemgpu_master_create: sizes, bit rows and row pointers are needed"); return NEMGPU_E_FUNCARG; }
    const int wf = (d + 31) / 32, nw64 = (n + 63) / 64;
    if (wf > nemk::chunk_mask_words_max()) { set_error("nemgpu_master_create: more than 131 072 organisms"); return NEMGPU_E_ARG; }
    const long long nnz_ll = (long long)nei_ptr[n] - nei_ptr[0];
    if (nei_ptr[0] != 0 || nnz_ll < 0 || nnz_ll > 0x7fffffff) { set_error("graph: ptr[0] must be 0 and ptr non-decreasing"); return NEMGPU_E_ARG; }
    const int nnz = (int)nnz_ll;
    for (int i = 0; i < n; i++) if (nei_ptr[i + 1] < nei_ptr[i]) { set_error("graph: ptr not monotone"); return NEMGPU_E_ARG; }
    if (nnz > 0 && (!nei_idx || !edge_bits)) { set_error("nemgpu_master_create: a graph needs neighbour indices and edge organism sets"); return NEMGPU_E_FUNCARG; }
    for (int t = 0; t < nnz; t++) if (nei_idx[t] < 0 || nei_idx[t] >= n) { set_error("graph: neighbour index out of range"); return NEMGPU_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no usable HIP device: this library has no CPU fallback"); return NEMGPU_E_DEVICE; }
    if (device < 0 || device >= ndev) { set_error("nemgpu_master_create: bad device index"); return NEMGPU_E_ARG; }
    g_hip_used.store(true);
    HIPCHK(hipSetDevice(device));
    nemgpu_master* m = new nemgpu_master();
    m->device = device; m->n = n; m->d = d; m->wf = wf; m->nw64 = nw64; m->nnz = nnz;
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_xt = a256((size_t)d * nw64 * 8), b_ptr = a256(((size_t)n + 1) * 4), b_idx = a256((size_t)std::max(nnz, 1) * 4),
                 b_eb = a256((size_t)std::max(nnz, 1) * wf * 4), b_xf = a256((size_t)n * wf * 4);
    uint32_t* xf_tmp = nullptr;
    auto fail = [&](const char* what) { if (xf_tmp) (void)hipFree(xf_tmp); if (m->block) (void)hipFree(m->block);
                                        if (m->stream) (void)hipStreamDestroy(m->stream); delete m; set_error(what); return NEMGPU_E_DEVICE; };
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    if (hipMalloc(&m->block, b_xt + b_ptr + b_idx + b_eb) != hipSuccess) return fail("nemgpu_master_create: device memory");
    if (hipMalloc(&xf_tmp, b_xf) != hipSuccess) return fail("nemgpu_master_create: device memory");
    uint64_t* xt = (uint64_t*)m->block;
    int* dptr = (int*)(m->block + b_xt);
    int* didx = (int*)(m->block + b_xt + b_ptr);
    uint32_t* deb = (uint32_t*)(m->block + b_xt + b_ptr + b_idx);
    // the bits above organism d - 1 in a row's last word are not data
    std::vector<uint32_t> rows(xbits, xbits + (size_t)n * wf);
    if (d & 31) { const uint32_t keep = (1u << (d & 31)) - 1u; for (int i = 0; i < n; i++) rows[(size_t)i * wf + wf - 1] &= keep; }
    hipError_t err = hipMemcpyAsync(xf_tmp, rows.data(), (size_t)n * wf * 4, hipMemcpyHostToDevice, m->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(dptr, nei_ptr, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, m->stream);
    if (err == hipSuccess && nnz > 0) err = hipMemcpyAsync(didx, nei_idx, (size_t)nnz * 4, hipMemcpyHostToDevice, m->stream);
    if (err == hipSuccess && nnz > 0) err = hipMemcpyAsync(deb, edge_bits, (size_t)nnz * wf * 4, hipMemcpyHostToDevice, m->stream);
    if (err == hipSuccess) { nemk::launch_master_transpose(xf_tmp, n, wf, d, nw64, xt, m->stream); err = hipGetLastError(); }
    if (err == hipSuccess) err = hipStreamSynchronize(m->stream);
    if (err != hipSuccess) return fail("nemgpu_master_create: upload failed");
    (void)hipFree(xf_tmp);
    m->dev = nemk::MasterDev{n, d, wf, nw64, nnz, xt, dptr, didx, deb};
    *out = m;
    return NEMGPU_OK;
}

void nemgpu_master_destroy(nemgpu_master* m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    if (m->block) (void)hipFree(m->block);
    delete m;
}

// the engine's matrix staging rows, lane order and graph block filled by the device from the master (the engine was
// created with n = the chunk's kept families, d = the sample's organisms)
static int adopt_chunk(nemgpu_engine* e, const nemgpu_master* M, const nemk::ChunkPlan& plan, int nnz_c)
{
    if (e->device != M->device) { set_error("a chunk's engine lives on its master's device"); return NEMGPU_E_ARG; }
    HIPCHK(hipSetDevice(e->device));
    alloc_for(e);
    int r;
    const size_t words = (size_t)e->n * e->wf;
    if ((r = dev_alloc(&e->xf_stage, words + (size_t)e->npad))) return r;
    e->perm = reinterpret_cast<int*>(e->xf_stage + words);
    const int nnz = std::max(nnz_c, 0);
    const size_t b_ptr = sizeof(int) * ((size_t)e->n + 1), b_nz = sizeof(int) * (size_t)std::max(nnz, 1);
    const size_t o_idx = (b_ptr + 255) & ~(size_t)255, o_w = (o_idx + b_nz + 255) & ~(size_t)255, total = o_w + b_nz;
    char* dev = nullptr;
    if ((r = dev_alloc(&dev, total))) return r;
    e->nei_ptr = (int*)dev; e->nei_idx = (int*)(dev + o_idx); e->nei_w = (float*)(dev + o_w);
    nemk::ChunkFill f{e->n, e->d, e->wf, e->npad, e->xf_stage, e->perm, e->nei_ptr, e->nei_idx, e->nei_w};
    nemk::launch_chunk_fill(M->dev, plan, f, e->stream);
    HIPCHK(hipGetLastError());
    e->nnz = nnz; e->has_graph = nnz > 0; e->exp_need = kExpTabGlobal;
    e->host_bits = nullptr; e->host_bits_words = 0;              // (no host copy of the rows: random starts are not for chunk engines)
    if (e->defer_layout) e->layout_pending = true;               // (the group's first step carries the layouts, zipped)
    else { launch_layout(e->xf_stage, e->n, e->wf, e->W, e->npad, e->d, e->nw64, e->xw, e->xt, e->perm, e->xws, e->stream); HIPCHK(hipGetLastError()); }
    e->have_matrix = true;
    return NEMGPU_OK;
}

int nemgpu_solve_chunks(nemgpu_master* M, nemgpu_chunk* chunks, int count, int k, const float* prop, const float* center_k,
                        const float* disp_k, const nemgpu_config* cfg, int workers, int group)
{
    if (!M || !chunks || count <= 0 || k <= 0 || k > kMaxKernelK || !prop || !center_k || !disp_k || !cfg) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(M->device));
    const int n = M->n, nw64 = M->nw64, wf = M->wf, nnz = M->nnz;
    size_t org_total = 0;
    int max_dc = 0;
    for (int c = 0; c < count; c++) {
        nemgpu_chunk& q = chunks[c];
        q.rc = NEMGPU_OK; q.n = 0; q.nnz = 0; q.result = nemgpu_result{};
        if (!q.organisms || q.dc <= 0 || q.dc > M->d) { set_error("nemgpu_solve_chunks: chunk " + std::to_string(c) + ": 1 .. d organisms are needed"); return NEMGPU_E_FUNCARG; }
        for (int t = 0; t < q.dc; t++)
            if (q.organisms[t] < 0 || q.organisms[t] >= M->d) { set_error("nemgpu_solve_chunks: chunk " + std::to_string(c) + ": organism index out of range"); return NEMGPU_E_ARG; }
        org_total += ((size_t)q.dc + 63) & ~(size_t)63;
        max_dc = std::max(max_dc, q.dc);
    }
    // ---- phase 1, all chunks: which families, which edges, how many of each
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_mask = a256((size_t)wf * 4), b_keep = a256((size_t)nw64 * 8), b_list = a256((size_t)n * 4), b_cov = a256((size_t)std::max(nnz, 1) * 2),
                 b_ptr = a256(((size_t)n + 1) * 4), b_cnt = 256;
    (void)b_cnt;
    const size_t per = b_mask + 2 * b_list + b_cov + b_ptr;
    // (every chunk's two counters and kept-family bits sit together: two copies bring them all to the host)
    const size_t b_plans = a256((size_t)count * sizeof(nemk::ChunkPlan)), b_org = a256(org_total * 4), b_counts = a256((size_t)count * 8),
                 b_keeps = b_keep * (size_t)count;
    const size_t o_counts = b_plans + b_org, o_keeps = o_counts + b_counts, o_per = o_keeps + b_keeps;
    static const bool prof = getenv("NEM_MI355X_BATCH_PROF") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    char* slab = nullptr;
    if (hipMalloc(&slab, o_per + per * (size_t)count) != hipSuccess) { (void)hipGetLastError(); set_error("nemgpu_solve_chunks: device memory for the chunk plans"); return NEMGPU_E_MEMORY; }
    struct SlabFree { char* p; ~SlabFree() { if (p) (void)hipFree(p); } } slab_free{slab};
    std::vector<nemk::ChunkPlan> plans((size_t)count);
    std::vector<int> org_host(org_total, 0);
    {
        size_t oo = 0;
        for (int c = 0; c < count; c++) {
            char* b = slab + o_per + per * (size_t)c;
            nemk::ChunkPlan& p = plans[(size_t)c];
            p.organisms = reinterpret_cast<const int*>(slab + b_plans) + oo; p.dc = chunks[c].dc;
            memcpy(org_host.data() + oo, chunks[c].organisms, (size_t)chunks[c].dc * 4);
            oo += ((size_t)chunks[c].dc + 63) & ~(size_t)63;
            p.mask = (uint32_t*)b; b += b_mask;
            p.keep = (uint64_t*)(slab + o_keeps + b_keep * (size_t)c);
            p.list = (int*)b; b += b_list;
            p.map = (int*)b; b += b_list;
            p.cov = (uint16_t*)b; b += b_cov;
            p.ptr = (int*)b;
            p.counts = (int*)(slab + o_counts) + 2 * (size_t)c;
        }
    }
    HIPCHK(hipMemcpyAsync(slab, plans.data(), (size_t)count * sizeof(nemk::ChunkPlan), hipMemcpyHostToDevice, M->stream));
    HIPCHK(hipMemcpyAsync(slab + b_plans, org_host.data(), org_total * 4, hipMemcpyHostToDevice, M->stream));
    nemk::launch_chunk_plan(M->dev, reinterpret_cast<const nemk::ChunkPlan*>(slab), count, max_dc, M->stream);
    HIPCHK(hipGetLastError());
    std::vector<int> counts((size_t)count * 2);
    std::vector<char> keeps;
    bool want_keep = false;
    for (int c = 0; c < count; c++) want_keep = want_keep || chunks[c].keep != nullptr;
    HIPCHK(hipMemcpyAsync(counts.data(), slab + o_counts, (size_t)count * 8, hipMemcpyDeviceToHost, M->stream));
    if (want_keep) { keeps.resize(b_keeps); HIPCHK(hipMemcpyAsync(keeps.data(), slab + o_keeps, b_keeps, hipMemcpyDeviceToHost, M->stream)); }
    HIPCHK(hipStreamSynchronize(M->stream));
    if (want_keep)
        for (int c = 0; c < count; c++) if (chunks[c].keep) memcpy(chunks[c].keep, keeps.data() + b_keep * (size_t)c, (size_t)nw64 * 8);
    const auto t_plan = std::chrono::steady_clock::now();
    // ---- the problems: sizes from phase 1, PPanGGOLiN-style initial parameters (one value per class and kind)
    std::vector<nemgpu_problem> P((size_t)count);
    std::vector<int> nnzc((size_t)count);
    std::vector<uint8_t*> labels((size_t)count, nullptr);
    std::map<int, std::vector<float>> init;                   // sample size -> prop | center | disp
    for (int c = 0; c < count; c++) {
        nemgpu_chunk& q = chunks[c];
        q.n = counts[(size_t)c * 2]; q.nnz = counts[(size_t)c * 2 + 1];
        if (q.n <= 0) { set_error("nemgpu_solve_chunks: chunk " + std::to_string(c) + " holds no family"); q.rc = NEMGPU_E_ARG; return NEMGPU_E_ARG; }
        nnzc[(size_t)c] = q.nnz;
        labels[(size_t)c] = q.labels;
        std::vector<float>& v = init[q.dc];
        if (v.empty()) {
            v.resize((size_t)k + 2 * (size_t)k * q.dc);
            for (int h = 0; h < k; h++) {
                v[(size_t)h] = prop[h];
                for (int o = 0; o < q.dc; o++) { v[(size_t)k + (size_t)h * q.dc + o] = center_k[h]; v[(size_t)k + (size_t)k * q.dc + (size_t)h * q.dc + o] = disp_k[h]; }
            }
        }
        nemgpu_problem& p = P[(size_t)c];
        p = nemgpu_problem{};
        p.n = q.n; p.d = q.dc; p.k = k;
        p.prop = v.data(); p.center = v.data() + k; p.disp = v.data() + k + (size_t)k * q.dc;
        p.out_prop = q.out_prop; p.out_center = q.out_center; p.out_disp = q.out_disp; p.out_nbobs_k = q.out_nbobs_k; p.out_c = nullptr;
    }
    ChunkSource src;
    src.master = M; src.plans = plans.data(); src.nnz = nnzc.data(); src.labels = labels.data();
    const auto t_prep = std::chrono::steady_clock::now();
    const int rc = solve_many_one(P.data(), count, cfg, M->device, workers, group, &src);
    for (int c = 0; c < count; c++) { chunks[c].rc = P[(size_t)c].rc; chunks[c].result = P[(size_t)c].result; }
    if (prof) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[solve_chunks] %d chunks: plans (which families, which edges; one wait) %.2f ms, problem records %.2f ms, pipeline %.2f ms\n",
                count, ms(t_begin, t_plan), ms(t_plan, t_prep), ms(t_prep, std::chrono::steady_clock::now()));
    }
    return rc;
}

// Which device solves which problem when nemgpu_solve_many_devices deals the lock-step groups: group g (problems
// g*group .. g*group + group - 1) goes to slot g % n_devices of the device list.  Host arithmetic only.
int nemgpu_deal_groups(int count, int group, int n_devices, int* slot_of_problem)
{
    if (count < 0 || n_devices <= 0 || !slot_of_problem) return NEMGPU_E_FUNCARG;
    group = std::max(1, std::min(group, 256));
    for (int i = 0; i < count; i++) slot_of_problem[i] = (i / group) % n_devices;
    return NEMGPU_OK;
}

// nemgpu_solve_many over SEVERAL devices of this process: the reference's own form of parallelism -- independent chunk
// problems side by side (multiprocessing.Pool over 500-organism chunks, ppanggolin.py:1039-1095) -- with the devices in
// the role of the pool's workers.  The lock-step groups are dealt round-robin over `devices` (an entry may repeat: two
// pipelines on one device); every device runs its share through nemgpu_solve_many on a thread of its own, with its own
// worker threads, streams, lock-step contexts and resource pool; no device waits for another.  Every problem's result
// equals its own nemgpu_run.  Returns the first failure (each problem's own status is in its rc).
int nemgpu_solve_many_devices(nemgpu_problem* P, int count, const nemgpu_config* cfg, const int* devices, int n_devices,
                              int workers, int group)
{
    if (!P || count <= 0 || !cfg || !devices || n_devices <= 0) return NEMGPU_E_FUNCARG;
    if (n_devices == 1) return nemgpu_solve_many(P, count, cfg, devices[0], workers, group);
    const int ndev = nemgpu_device_count();
    for (int s = 0; s < n_devices; s++)
        if (devices[s] < 0 || devices[s] >= ndev) { set_error("nemgpu_solve_many_devices: bad device index in the list"); return NEMGPU_E_ARG; }
    group = std::max(1, std::min(group, 256));
    std::vector<int> slot((size_t)count);
    (void)nemgpu_deal_groups(count, group, n_devices, slot.data());
    std::vector<std::vector<nemgpu_problem>> sub((size_t)n_devices);
    std::vector<std::vector<int>> who((size_t)n_devices);
    for (int i = 0; i < count; i++) { sub[(size_t)slot[i]].push_back(P[i]); who[(size_t)slot[i]].push_back(i); }
    std::vector<int> rcs((size_t)n_devices, NEMGPU_OK);
    std::vector<std::string> errs((size_t)n_devices);
    const int wk = std::max(1, workers / n_devices);
    std::vector<std::thread> th;
    for (int s = 0; s < n_devices; s++) {
        if (sub[(size_t)s].empty()) continue;
        th.emplace_back([&, s] {
            tl_runner = true;
            rcs[(size_t)s] = nemgpu_solve_many(sub[(size_t)s].data(), (int)sub[(size_t)s].size(), cfg, devices[s], wk, group);
            if (rcs[(size_t)s] != NEMGPU_OK) errs[(size_t)s] = g_last_error;       // (the error text is per thread)
        });
    }
    for (std::thread& t : th) t.join();
    int rc = NEMGPU_OK;
    for (int s = 0; s < n_devices; s++) {
        for (size_t j = 0; j < sub[(size_t)s].size(); j++) {
            P[who[(size_t)s][j]].result = sub[(size_t)s][j].result;
            P[who[(size_t)s][j]].rc = sub[(size_t)s][j].rc;
        }
        if (rcs[(size_t)s] != NEMGPU_OK && rc == NEMGPU_OK) { rc = rcs[(size_t)s]; set_error(errs[(size_t)s]); }
    }
    return rc;
}

static int run_random_impl(nemgpu_engine* e, int n_starts, uint32_t seed, nemgpu_result* res, int* best_start,
                           nemgpu_log_fn fn, void* user);

int nemgpu_run_random(nemgpu_engine* e, int n_starts, uint32_t seed, nemgpu_result* res, int* best_start)
{
    return run_random_impl(e, n_starts, seed, res, best_start, nullptr, nullptr);
}

int nemgpu_run_random_logged(nemgpu_engine* e, int n_starts, uint32_t seed, nemgpu_result* res, int* best_start,
                             nemgpu_log_fn fn, void* user)
{
    if (!fn) return NEMGPU_E_FUNCARG;
    return run_random_impl(e, n_starts, seed, res, best_start, fn, user);
}

// fn != nullptr: the starts one after the other, one logged step per host round trip (nemgpu_iterate_logged's form)
static int run_random_impl(nemgpu_engine* e, int n_starts, uint32_t seed, nemgpu_result* res, int* best_start,
                           nemgpu_log_fn fn, void* user)
{
    if (!e || n_starts <= 0) return NEMGPU_E_FUNCARG;
    if (e->lo != 0 || e->hi != e->n_total) { set_error("random starts need the whole problem on one engine"); return NEMGPU_E_FUNCARG; }
    if (!e->have_matrix || e->host_bits_words == 0) { set_error("the matrix must be set first"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    {
        // the starts are independent EM runs on ONE matrix: by default they run in lock step, every launch serving all
        // of them (NEM_MI355X_BATCH_STARTS=0: one after the other on this engine)
        const char* g = getenv("NEM_MI355X_BATCH_STARTS");
        const bool lock_ok = n_starts > 1 && !(g && g[0] == '0') && !crit_test(e);
        if (!fn && lock_ok) return run_random_lockstep(e, n_starts, seed, res, best_start);
        // With the per-iteration log (fn): one iteration per lock-step step, the criteria of all starts side by side, the
        // lines handed over start by start afterwards (NEM_MI355X_BATCH_STARTS_LOGGED=0: one start after the other).  One
        // round only -- under TIE_LIBC a start that draws behind its initial sweeps sends the call to the sequential form
        // below before a line has been handed over.
        const char* gl = getenv("NEM_MI355X_BATCH_STARTS_LOGGED");
        if (fn && lock_ok && n_starts <= 64 && !(gl && gl[0] == '0')) {
            const int rr = run_random_lockstep(e, n_starts, seed, res, best_start, fn, user);
            if (rr != kStartsNotInLockStep) return rr;
        }
    }
    int r;
    const int n = e->n, d = e->d, k = e->k, wf = e->wf;
    const size_t kd = (size_t)k * d;
    if ((r = ensure_state_buffers(e))) return r;
    // ---- InitPara: dispersion of the whole sample
    e->have_params = true;                                         // (the slots are (re)written below)
    if ((r = reset_state(e))) return r;
    if (e->ncem()) {
        HIPCHK(hipMemsetAsync(e->lab[0], 0, (size_t)e->n_total, e->stream));
    } else {
        std::vector<float> c1((size_t)n * k, 0.0f);
        for (int i = 0; i < n; i++) c1[(size_t)i * k] = 1.0f;
        HIPCHK(copy_sync(e, e->cbuf[0], c1.data(), c1.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    e->cur = 0; e->masks_valid = false;
    if ((r = do_mstep(e))) return r;
    if (!e->ncem() && (r = read_iter_flags(e))) return r;         // (a fuzzy M-step can report a fault)
    std::vector<float> dispsam((size_t)d);
    HIPCHK(copy_sync(e, dispsam.data(), e->disp, sizeof(float) * d, hipMemcpyDeviceToHost));

    GlibcRandom rng(seed);
    if (e->libc() && e->cfg.tie_seed != seed) { e->cfg.tie_seed = seed; e->draw_valid = false; }   // one stream (nem_exe.c:621)
    // RandomInteger (nem_rnd.c:40-63); with TIE_LIBC the starts' draws and the sweeps' tie draws are ONE stream
    auto draw_integer = [&](int mini, int maxi, int* out) -> int {
        if (mini >= maxi) { *out = maxi; return NEMGPU_OK; }
        if (!e->libc()) { *out = rng.integer(mini, maxi); return NEMGPU_OK; }
        uint32_t v = 0;
        int rr = host_draw(e, &v);
        *out = (int)(v % (uint32_t)(maxi - mini + 1)) + mini;
        return rr;
    };
    std::vector<float> prop((size_t)k), center(kd), disp(kd);
    auto bit = [&](int i, int j) { return (float)((e->host_bits[(size_t)i * wf + (j >> 5)] >> (j & 31)) & 1u); };
    alloc_for(e);
    if (e->ncem()) { if (!e->best_lab && (r = dev_alloc(&e->best_lab, (size_t)e->n_total))) return r; }
    else { if (!e->best_c && (r = dev_alloc(&e->best_c, (size_t)e->n_total * k))) return r; }
    uint8_t* best_lab = e->best_lab; float* best_c = e->best_c;
    auto cleanup = [&]() { (void)hipStreamSynchronize(e->stream); };
    int nbsucc = 0, best = -1, last_status = NEMGPU_OK;
    float best_crit[6] = {0, 0, 0, 0, 0, 0};
    nemgpu_result best_res{};
    e->rs_rounds = e->rs_lockstep = e->rs_redone = e->rs_two_waits = 0; e->rs_alone = n_starts;   // (one after the other)
    for (int s = 0; s < n_starts; s++) {
        // MakeRandomPara
        for (int h = 0; h < k; h++) for (int j = 0; j < d; j++) disp[(size_t)h * d + j] = dispsam[j] / k;      // :1400
        for (int h = 0; h < k; h++) prop[h] = (float)(1.0 / k);                                              // :1405
        for (int h = 0; h < k; h++) {
            int ipt = 0;
            bool again = true;
            for (int ndraw = 0; again && ndraw < 100; ndraw++) {                                             // :1419
                if ((r = draw_integer(0, n - 1, &ipt))) { cleanup(); return r; }
                again = false;
                for (int g = 0; g < h && !again; g++) {
                    bool different = false;
                    for (int j = 0; j < d && !different; j++) different = center[(size_t)g * d + j] != bit(ipt, j);
                    if (!different) again = true;
                }
            }
            for (int j = 0; j < d; j++) center[(size_t)h * d + j] = bit(ipt, j);                             // :1457
        }
        HIPCHK(copy_sync(e, e->prop0, prop.data(), sizeof(float) * k, hipMemcpyHostToDevice));
        HIPCHK(copy_sync(e, e->center0, center.data(), sizeof(float) * kd, hipMemcpyHostToDevice));
        HIPCHK(copy_sync(e, e->disp0, disp.data(), sizeof(float) * kd, hipMemcpyHostToDevice));
        // ComputePartitionFromPara + NemAlgo + criteria, as nemgpu_run does
        float crit[6];
        bool have_crit = false;
        if (!fn) {
            if ((r = iterate(e, e->cfg.it_max, true))) { cleanup(); return r; }
        } else {
            nemgpu_log_event ev{};
            ev.kind = NEMGPU_LOG_START; ev.start = s;
            fn(&ev, user);
            float cb[6], ca[6];
            std::vector<float> nb((size_t)k);
            nemgpu_result r1{};
            if ((r = iterate_logged_impl(e, 1, true, &r1, cb, ca, prop.data(), center.data(), disp.data(), nullptr))) { cleanup(); return r; }
            ev.kind = NEMGPU_LOG_LINE; ev.iter = 0; ev.crit_before = cb; ev.crit_after = ca;
            ev.prop = prop.data(); ev.center = center.data(); ev.disp = disp.data(); ev.nbobs_k = nullptr;
            fn(&ev, user);
            for (int iter = 1; iter <= e->cfg.it_max && !e->converged && e->status == NEMGPU_OK; iter++) {
                if ((r = iterate_logged_impl(e, 0, true, &r1, cb, ca, prop.data(), center.data(), disp.data(), nb.data()))) { cleanup(); return r; }
                ev.iter = iter; ev.nbobs_k = nb.data();
                if (e->status == NEMGPU_W_EMPTYCLASS) {
                    // EstimSizes (nem_mod.c:1275-1317) ran before the empty class was found: i-ordered float sums per class
                    std::vector<float> part((size_t)n * k);
                    if ((r = nemgpu_get_partition(e, part.data()))) { cleanup(); return r; }
                    for (int h = 0; h < k; h++) {
                        float acc = 0.0f;
                        for (int i = 0; i < n; i++) acc += part[(size_t)i * k + h];
                        nb[(size_t)h] = acc;
                    }
                    ev.kind = NEMGPU_LOG_EMPTY; ev.emptyk = e->emptyk;
                    fn(&ev, user);
                    break;
                }
                ev.kind = NEMGPU_LOG_LINE;
                fn(&ev, user);
                memcpy(crit, ca, sizeof crit);                     // (of the final partition, on the final densities, if the run ends here)
                have_crit = true;
            }
            if (e->status == NEMGPU_W_EMPTYCLASS) have_crit = false;
        }
        if (e->iters == 0) {
            if ((r = do_mstep(e)) || (r = do_tables(e)) || (r = do_density(e))) { cleanup(); return r; }
            have_crit = false;
        }
        if (!have_crit && (r = criteria(e, crit))) { cleanup(); return r; }
        last_status = e->status;
        if (e->status == NEMGPU_OK) {
            nbsucc++;
            if (nbsucc == 1 || crit[3] > best_crit[3]) {                                                     // :1676-1697
                if (e->ncem()) HIPCHK(hipMemcpyAsync(best_lab, e->lab[e->cur], (size_t)e->n_total, hipMemcpyDeviceToDevice, e->stream));
                else HIPCHK(hipMemcpyAsync(best_c, e->cbuf[e->cur], sizeof(float) * (size_t)e->n_total * k, hipMemcpyDeviceToDevice, e->stream));
                for (int t = 0; t < 6; t++) best_crit[t] = crit[t];
                best = s;
                fill_result(e, &best_res);
            }
        }
    }
    if (nbsucc > 0) {
        if (e->ncem()) HIPCHK(hipMemcpyAsync(e->lab[e->cur], best_lab, (size_t)e->n_total, hipMemcpyDeviceToDevice, e->stream));
        else HIPCHK(hipMemcpyAsync(e->cbuf[e->cur], best_c, sizeof(float) * (size_t)e->n_total * k, hipMemcpyDeviceToDevice, e->stream));
        e->masks_valid = false; e->tables_fresh = false; e->density_fresh = false;
        if ((r = do_mstep(e)) || (!e->ncem() && (r = read_iter_flags(e)))) { cleanup(); return r; }             // :1711
        e->status = NEMGPU_OK; e->emptyk = 0;
        e->iters = best_res.iters; e->converged = best_res.converged;
        if (res) { *res = best_res; res->status = NEMGPU_OK; res->tie_draws = e->draws; for (int t = 0; t < 6; t++) res->crit[t] = best_crit[t]; }
    } else if (res) {
        fill_result(e, res);
        res->status = last_status;
    }
    if (best_start) *best_start = best;
    cleanup();
    return NEMGPU_OK;
}

// ---- random starts in lock step -------------------------------------------------------------------------------
// A state-only twin of `p`: matrix layouts, graph and draw table are p's, everything else is carved from `slab`.
static int make_clone(nemgpu_engine* p, nemgpu_engine** out, char* slab, size_t slab_bytes, int* flags_host, float* par0)
{
    nemgpu_engine* c = new nemgpu_engine();
    c->n_total = p->n_total; c->n_true = p->n_true; c->d = p->d; c->k = p->k; c->lo = p->lo; c->hi = p->hi; c->n = p->n;
    c->npad = p->npad; c->dpad = p->dpad; c->W = p->W; c->wf = p->wf; c->nw64 = p->nw64; c->device = p->device;
    c->stream = p->stream; c->own_stream = false;
    c->cfg = p->cfg; c->have_matrix = true; c->have_params = true; c->has_graph = p->has_graph;
    c->xw = p->xw; c->xws = p->xws; c->perm = p->perm; c->xt = p->xt; c->use_sort = p->use_sort;
    c->nei_ptr = p->nei_ptr; c->nei_idx = p->nei_idx; c->nei_w = p->nei_w; c->nnz = p->nnz;
    c->use_graphs = p->use_graphs; c->ff_mode = p->ff_mode; c->round_batch = p->round_batch; c->rounds_iter = p->rounds_iter;
    c->fuzzy_chains = p->fuzzy_chains;   // (graphs: the zipped sequences')
    c->parent = p; c->carve_all = true;
    c->chunks.push_back({slab, slab_bytes, 0, false});
    c->shared_chunk = 0;
    c->flags_host = flags_host; c->flags_host_borrowed = true;
    const size_t kd = (size_t)p->k * p->d;
    c->prop0 = par0; c->center0 = par0 + p->k; c->disp0 = par0 + p->k + kd;
    alloc_for(c);
    int r = alloc_model_buffers(c);
    if (r == NEMGPU_OK) r = ensure_state_buffers(c);
    if (r == NEMGPU_OK) r = ensure_crit_buffers(c);
    alloc_for(p);
    if (r != NEMGPU_OK) { nemgpu_destroy(c); return r; }
    c->draw_ctl = c->sweep_next + 8;
    *out = c;
    return NEMGPU_OK;
}

// `count` twins of e (kept for the next call)
static int ensure_clones(nemgpu_engine* e, int count)
{
    if ((int)e->clones.size() >= count) return NEMGPU_OK;
    HIPCHK(hipStreamSynchronize(e->stream));
    for (nemgpu_engine* c : e->clones) nemgpu_destroy(c);
    e->clones.clear();
    pool_put(e->device, false, e->clone_slab, e->clone_slab_size); e->clone_slab = nullptr;
    pool_put(e->device, true, (char*)e->clone_flags_host, e->clone_flags_size); e->clone_flags_host = nullptr;
    // what one twin carves: a dry run of its allocations
    {
        nemgpu_engine probe;
        probe.n_total = e->n_total; probe.d = e->d; probe.k = e->k; probe.n = e->n; probe.npad = e->npad; probe.dpad = e->dpad;
        probe.W = e->W; probe.nw64 = e->nw64; probe.cfg = e->cfg; probe.parent = e; probe.dry_run = true;
        alloc_for(&probe);
        int r = alloc_model_buffers(&probe);
        if (r == NEMGPU_OK) r = ensure_state_buffers(&probe);
        if (r == NEMGPU_OK) r = ensure_crit_buffers(&probe);
        alloc_for(e);
        if (r) return r;
        e->clone_bytes = probe.dry_bytes + 4096;
    }
    const size_t kd = (size_t)e->k * e->d;
    const size_t par_bytes = (((size_t)count * (e->k + 2 * kd) * sizeof(float)) + 255) & ~(size_t)255;
    HIPCHK(pool_get(e->device, false, par_bytes + (size_t)count * e->clone_bytes, &e->clone_slab, &e->clone_slab_size));
    HIPCHK(hipMemsetAsync(e->clone_slab, 0, par_bytes + (size_t)count * e->clone_bytes, e->stream));
    HIPCHK(pool_get(e->device, true, (size_t)count * e->flag_words() * sizeof(int), (char**)&e->clone_flags_host, &e->clone_flags_size));
    e->clone_par0 = reinterpret_cast<float*>(e->clone_slab);
    for (int i = 0; i < count; i++) {
        nemgpu_engine* c = nullptr;
        int r = make_clone(e, &c, e->clone_slab + par_bytes + (size_t)i * e->clone_bytes, e->clone_bytes,
                           e->clone_flags_host + (size_t)i * e->flag_words(), e->clone_par0 + (size_t)i * (e->k + 2 * kd));
        if (r) return r;
        e->clones.push_back(c);
    }
    return NEMGPU_OK;
}

// RandNemAlgo (nem_alg.c:1574-1742) with the starts in lock step.  TIE_LIBC: the starts' centre draws and the sweeps'
// tie draws are one stream, so where start s begins depends on the ties of the starts before it.  Where the ties ARE
// is what makes lock step possible: a random start gives every class the same dispersion (the whole sample's over K,
// nem_alg.c:1400) and the same proportion, so in the two initial sweeps every family that is as far from two centres
// ties exactly -- hundreds to thousands of draws per start -- while from the first M-step on the classes' parameters
// differ and a tie is an accident (profiles/r04_random_starts_draws.json: 100 starts, 0 draws behind the initial sweeps).
// So a round runs in two phases.  A: start by start, in stream order -- centres drawn, parameters up, density, the blind
// sweep and the beta sweep completed from the host (what the sequential form does for them), which leaves the stream
// where the next start's centres are drawn.  B: the EM iterations of all the round's starts in lock step, each from its
// own initial partition, on the assumption that none of them draws.  The assumption is checked: the first start whose
// iterations did draw is still right (its draws came from where they should) and everything behind it is redone from
// the position it left the stream at.
static int run_random_lockstep(nemgpu_engine* e, int n_starts, uint32_t seed, nemgpu_result* res, int* best_start, nemgpu_log_fn fn, void* user)
{
    int r;
    const int n = e->n, d = e->d, k = e->k, wf = e->wf;
    const size_t kd = (size_t)k * d, par = (size_t)k + 2 * kd;
    if ((r = ensure_state_buffers(e))) return r;
    // ---- InitPara: dispersion of the whole sample
    e->have_params = true;
    if ((r = reset_state(e))) return r;
    if (e->ncem()) {
        HIPCHK(hipMemsetAsync(e->lab[0], 0, (size_t)e->n_total, e->stream));
    } else {
        std::vector<float> c1((size_t)n * k, 0.0f);
        for (int i = 0; i < n; i++) c1[(size_t)i * k] = 1.0f;
        HIPCHK(copy_sync(e, e->cbuf[0], c1.data(), c1.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    e->cur = 0; e->masks_valid = false;
    if ((r = do_mstep(e))) return r;
    if (!e->ncem() && (r = read_iter_flags(e))) return r;         // (a fuzzy M-step can report a fault)
    std::vector<float> dispsam((size_t)d);
    HIPCHK(copy_sync(e, dispsam.data(), e->disp, sizeof(float) * d, hipMemcpyDeviceToHost));

    GlibcRandom rng(seed);
    if (e->libc() && e->cfg.tie_seed != seed) { e->cfg.tie_seed = seed; e->draw_valid = false; }
    auto draw_integer = [&](int mini, int maxi, int* out) -> int {
        if (mini >= maxi) { *out = maxi; return NEMGPU_OK; }
        if (!e->libc()) { *out = rng.integer(mini, maxi); return NEMGPU_OK; }
        uint32_t v = 0;
        int rr = host_draw(e, &v);
        *out = (int)(v % (uint32_t)(maxi - mini + 1)) + mini;
        return rr;
    };
    auto bit = [&](int i, int j) { return (float)((e->host_bits[(size_t)i * wf + (j >> 5)] >> (j & 31)) & 1u); };
    alloc_for(e);
    if (e->ncem()) { if (!e->best_lab && (r = dev_alloc(&e->best_lab, (size_t)e->n_total))) return r; }
    else { if (!e->best_c && (r = dev_alloc(&e->best_c, (size_t)e->n_total * k))) return r; }
    const int group = std::min(n_starts, 64);
    // Starts per round.  TIE_LIBC: phase B bets that no start draws behind its initial sweeps.  Whether these data keep
    // the bet is not known ahead: start 0 runs ALONE, on this engine's own pipelined path (what the sequential form
    // does), and tells; if its iterations drew nothing the rest go as wide as the group allows.  After a start whose
    // iterations drew (alone, or one that voided the round behind it) the next runs alone, the width grows with the run
    // of clean starts behind it (4, 20, the rest), and after two voided rounds in a row every remaining start runs alone:
    // data whose every start draws in its iterations cost what the sequential form costs.
    const char* fa = getenv("NEM_MI355X_STARTS_FIRST_ALONE");      // (0: bet from the first start on -- tests of the losing path)
    int width = (e->libc() && !(fa && fa[0] == '0') && fn == nullptr) ? 1 : group;   // (a logged run bets from the first start on: a lost bet sends the whole call to the sequential form)
    int voided_in_a_row = 0, clean_run = 0; bool alone_for_good = false;
    if ((r = ensure_clones(e, group))) return r;
    for (nemgpu_engine* c : e->clones) { c->cfg = e->cfg; c->stream = e->stream; c->round_batch = e->round_batch; c->ff_mode = e->ff_mode; }

    e->rs_rounds = e->rs_lockstep = e->rs_alone = e->rs_redone = e->rs_two_waits = 0;
    // (random starts tie by the hundreds in their initial sweeps: the draw window is sized for that from the first start on,
    //  instead of growing -- and sliding away from the earlier starts' positions -- when the first heavy start is met)
    if (e->libc()) e->tie_heavy = true;
    int nbsucc = 0, best = -1, last_status = NEMGPU_OK;
    float best_crit[6] = {0, 0, 0, 0, 0, 0};
    nemgpu_result best_res{};
    std::vector<float> host_par;
    std::vector<int> start_pos;
    std::vector<GlibcRandom> rng_after;                           // (hash / first tie rules: the generator after each start's draws)
    int next = 0;
    static const bool prof = getenv("NEM_MI355X_BATCH_PROF") != nullptr;
    while (next < n_starts) {
        const int M = std::min(width, n_starts - next);
        std::vector<nemgpu_engine*> E(e->clones.begin(), e->clones.begin() + M);
        host_par.assign((size_t)M * par, 0.0f);
        start_pos.assign((size_t)M, 0);
        // MakeRandomPara, nem_alg.c:1381-1473, for start j of the round (TIE_LIBC: from where the stream stands)
        auto make_random_para = [&](int j) -> int {
            float* prop = host_par.data() + (size_t)j * par; float* center = prop + k; float* disp = center + kd;
            for (int h = 0; h < k; h++) for (int t = 0; t < d; t++) disp[(size_t)h * d + t] = dispsam[t] / k;     // :1400
            for (int h = 0; h < k; h++) prop[h] = (float)(1.0 / k);                                              // :1405
            for (int h = 0; h < k; h++) {
                int ipt = 0;
                bool again = true;
                for (int ndraw = 0; again && ndraw < 100; ndraw++) {                                             // :1419
                    int rr = draw_integer(0, n - 1, &ipt);
                    if (rr) return rr;
                    again = false;
                    for (int g = 0; g < h && !again; g++) {
                        bool different = false;
                        for (int t = 0; t < d && !different; t++) different = center[(size_t)g * d + t] != bit(ipt, t);
                        if (!different) again = true;
                    }
                }
                for (int t = 0; t < d; t++) center[(size_t)h * d + t] = bit(ipt, t);                             // :1457
            }
            start_pos[j] = e->draws;                                 // where this start's sweeps begin to draw
            return NEMGPU_OK;
        };
        if (e->libc() && M == 1) {
            // ---- one start alone: this engine's own run (nemgpu_run_random's sequential body)
            if ((r = make_random_para(0))) return r;
            const float* prop = host_par.data(); const float* center = prop + k; const float* disp = center + kd;
            HIPCHK(copy_sync(e, e->prop0, prop, sizeof(float) * k, hipMemcpyHostToDevice));
            HIPCHK(copy_sync(e, e->center0, center, sizeof(float) * kd, hipMemcpyHostToDevice));
            HIPCHK(copy_sync(e, e->disp0, disp, sizeof(float) * kd, hipMemcpyHostToDevice));
            e->draws_after_init = e->draws;
            if ((r = iterate(e, e->cfg.it_max, true))) return r;
            if (e->iters == 0) { if ((r = do_mstep(e)) || (r = do_tables(e)) || (r = do_density(e))) return r; }
            float crit[6];
            if ((r = criteria(e, crit))) return r;
            const bool drew_late = e->draws != e->draws_after_init;  // (its iterations drew: phase B's bet would have been lost)
            last_status = e->status;
            if (e->status == NEMGPU_OK) {
                nbsucc++;
                if (nbsucc == 1 || crit[3] > best_crit[3]) {                                                     // :1676-1697
                    HIPCHK(hipMemcpyAsync(e->best_lab, e->lab[e->cur], (size_t)e->n_total, hipMemcpyDeviceToDevice, e->stream));
                    for (int t = 0; t < 6; t++) best_crit[t] = crit[t];
                    best = next;
                    fill_result(e, &best_res);
                }
            }
            next += 1; e->rs_alone++;
            clean_run = drew_late ? 0 : clean_run + 1;
            width = (alone_for_good || clean_run == 0) ? 1 : (voided_in_a_row == 0 ? group : std::min(group, 4 * clean_run));
            continue;
        }
        std::vector<LoopCursor> L((size_t)M);
        std::vector<int> after_init((size_t)M, 0);
        if (e->libc()) {
            // ---- phase A: the starts' initial partitions, one after the other in stream order
            const int pos_round_first = e->draws;
            using clk = std::chrono::steady_clock;
            const auto ta = clk::now();
            double seg[5] = {0, 0, 0, 0, 0};
            auto lap = [&](int which, clk::time_point& t) { if (prof) { const auto now = clk::now(); seg[which] += std::chrono::duration<double, std::micro>(now - t).count(); t = now; } };
            auto lend_window = [&](nemgpu_engine* c) {
                c->draw_tab = e->draw_tab; c->draw_cap = e->draw_cap; c->draw_tab0 = e->draw_tab0; c->draw_valid = true;
                c->draw_borrowed = true;                           // (a twin that outgrows the shared window builds its own)
            };
            // the starts' parameters go up from a pinned block, one slot per start: no wait for a copy
            const size_t par_bytes = par * sizeof(float);
            if (e->rs_par_host_size < (size_t)M * par_bytes) {
                HIPCHK(hipStreamSynchronize(e->stream));
                pool_put(e->device, true, e->rs_par_host, e->rs_par_host_size); e->rs_par_host = nullptr; e->rs_par_host_size = 0;
                HIPCHK(pool_get(e->device, true, (size_t)group * par_bytes, &e->rs_par_host, &e->rs_par_host_size));
            }
            for (int j = 0; j < M; j++) {
                nemgpu_engine* c = E[j];
                auto t = clk::now();
                if ((r = make_random_para(j))) return r;
                // (k_finish reads the start's parameters where the host wrote them: a pinned slot is device-visible, 12 KB
                //  cross the bus inside the launch; a copy command of its own was 2 us + two queue gaps of 4-5 us in the chain)
                float* slot = reinterpret_cast<float*>(e->rs_par_host + (size_t)j * par_bytes);
                memcpy(slot, host_par.data() + (size_t)j * par, par_bytes);
                float* const own_par[3] = {c->prop0, c->center0, c->disp0};
                c->prop0 = slot; c->center0 = slot + k; c->disp0 = slot + k + kd;
                struct ParBack { nemgpu_engine* c; float* const* own; ~ParBack() { c->prop0 = own[0]; c->center0 = own[1]; c->disp0 = own[2]; } } par_back{c, own_par};
                if ((r = ensure_draw_window(e, e->draws, draw_need(e)))) return r;
                lend_window(c);
                c->draws = e->draws; c->tie_heavy = e->tie_heavy;
                if ((r = loop_begin(c, L[j], e->cfg.it_max, true))) return r;
                lap(0, t);
                // The rounds a sweep needs are long-tailed: a start that ties at a few families is through in three or
                // four, one whose centres tie at thousands needs dozens (a draw's number depends on every tie before
                // it).  What goes out at once: what 90 % of the starts so far got by with for the blind sweep (three: a
                // round behind the fixed point still costs 4 us), 70 % for the beta sweep; a sweep that needs more gets
                // batches that double (libc_init).
                int rounds_beta = 0;
                bool redone = false;
                if ((r = libc_init(c, &rounds_beta, &redone))) return r;
                if (redone) e->rs_two_waits++;
                lap(1, t);
                if (prof && rounds_beta >= 16) fprintf(stderr, "[random starts]   start %d: beta sweep %d rounds, %d draws in its initial sweeps\n", next + j, rounds_beta, c->draws - start_pos[j]);
                HIPCHK(hipGetLastError());
                lap(4, t);
                L[j].first = false;
                c->draws_after_init = after_init[j] = c->draws;
                e->draws = c->draws; e->tie_heavy = e->tie_heavy || c->tie_heavy;
            }
            // (the parent's window may have slid or grown since a twin borrowed it: it is made to reach from the round's
            //  first start to what the last one may still draw, and every twin gets it as it is now -- a twin whose
            //  position it did not cover would build a table of its own at its first batch, generator run from the seed
            //  and 2 MB up: 0.3 ms per twin, 14.7 ms of a cold engine's first round)
            if ((r = ensure_draw_window(e, pos_round_first, (long)(e->draws - pos_round_first) + draw_need(e)))) return r;
            for (int j = 0; j < M; j++) { lend_window(E[j]); E[j]->tie_heavy = e->tie_heavy; }
            if (prof) {
                HIPCHK(hipStreamSynchronize(e->stream));
                fprintf(stderr, "[random starts] phase A, %d starts: %.0f us (centres + upload %.0f, blind sweep out %.0f + wait %.0f, beta sweep out %.0f + wait and masks %.0f), "
                                "%d of them sweep by sweep, stream at %d\n", M,
                        std::chrono::duration<double, std::micro>(clk::now() - ta).count(), seg[0], seg[1], seg[2], seg[3], seg[4], e->rs_two_waits, e->draws);
                for (int w = 0; w < 2; w++) {
                    fprintf(stderr, "[random starts]   rounds needed by the %s sweeps so far:", w ? "beta" : "blind");
                    for (int q = 1; q <= 16; q++) if (e->libc_init_hist[w][q]) fprintf(stderr, " %d%s x%d", q, q == 16 ? "+" : "", e->libc_init_hist[w][q]);
                    fprintf(stderr, "\n");
                }
            }
        } else {
            for (int j = 0; j < M; j++) if ((r = make_random_para(j))) return r;
            HIPCHK(copy_sync(e, e->clone_par0, host_par.data(), host_par.size() * sizeof(float), hipMemcpyHostToDevice));
            for (int j = 0; j < M; j++) if ((r = loop_begin(E[j], L[j], e->cfg.it_max, true))) return r;
        }
        // ---- the EM iterations in lock step (TIE_LIBC: phase B)
        std::vector<StartLog> logs((size_t)(fn ? M : 0));
        if (fn) {
            if (!e->libc()) {                                       // (the initial partitions first: a batch of no iterations)
                for (int j = 0; j < M; j++) L[j].remaining = 0;
                if ((r = iterate_many(E, L))) return r;
                for (int j = 0; j < M; j++) L[j].remaining = e->cfg.it_max;
            }
            if ((r = iterate_many_logged(E, L, host_par.data(), par, logs))) return r;
        } else if ((r = iterate_many(E, L))) return r;
        for (int j = 0; j < M; j++)
            if (E[j]->iters == 0) { if ((r = do_mstep(E[j])) || (r = do_tables(E[j])) || (r = do_density(E[j]))) return r; }
        std::vector<float> crits((size_t)M * 6);
        bool staged = false;
        {
            std::vector<Recorder> recs((size_t)M);
            std::vector<int> all((size_t)M);
            for (int j = 0; j < M; j++) all[j] = j;
            // (the starts' criteria to the batch's staging area behind the zipped criteria launches, ONE copy to the host:
            //  see run_many)
            nemgpu_engine::ZipContext* z = zip_context(E[0]);
            staged = z != nullptr && z->zip_flags_dev != nullptr && z->zip_flags_host != nullptr && z->zip_flags_cap >= (size_t)8 * M;
            if ((r = lockstep(E, all, recs, [&](int m) {
                    const int rr = criteria_enqueue(E[m], -1);
                    if (rr == NEMGPU_OK && staged) launch_copy_words(reinterpret_cast<const int*>(E[m]->crit6_dev), z->zip_flags_dev + (size_t)8 * m, 6, E[0]->stream);
                    return rr;
                }, false))) return r;
            if (staged) {
                HIPCHK(hipMemcpyAsync(z->zip_flags_host, z->zip_flags_dev, (size_t)8 * M * sizeof(int), hipMemcpyDeviceToHost, E[0]->stream));
                HIPCHK(hipStreamSynchronize(E[0]->stream));
                for (int j = 0; j < M; j++) memcpy(crits.data() + (size_t)j * 6, z->zip_flags_host + (size_t)8 * j, 6 * sizeof(float));
            }
        }
        if (!staged) {
            for (int j = 0; j < M; j++)
                HIPCHK(hipMemcpyAsync(crits.data() + (size_t)j * 6, E[j]->crit6_dev, 6 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        }
        HIPCHK(hipStreamSynchronize(e->stream));
        // the starts of this round that stand: all of them, or (TIE_LIBC) up to the first one whose iterations drew
        int valid = M;
        if (e->libc())
            for (int j = 0; j < M; j++) if (E[j]->draws != after_init[j]) { valid = j + 1; break; }
        if (fn && (valid < M || (e->libc() && E[M - 1]->draws != after_init[M - 1]))) return kStartsNotInLockStep;
        if (fn) {
            // the round's lines to the caller's writer, start by start in the reference's order
            for (int j = 0; j < M; j++) {
                nemgpu_log_event ev{};
                ev.kind = NEMGPU_LOG_START; ev.start = next + j;
                fn(&ev, user);
                for (const StartLogLine& ln : logs[(size_t)j].lines) {
                    ev = nemgpu_log_event{};
                    ev.kind = ln.kind; ev.start = next + j; ev.iter = ln.iter; ev.emptyk = ln.emptyk;
                    ev.crit_before = ln.cb; ev.crit_after = ln.ca;
                    ev.prop = ln.prop.data(); ev.center = ln.center.data(); ev.disp = ln.disp.data();
                    ev.nbobs_k = ln.has_nb ? ln.nb.data() : nullptr;
                    fn(&ev, user);
                }
            }
        }
        for (int j = 0; j < valid; j++) {
            nemgpu_engine* c = E[j];
            const float* crit = crits.data() + (size_t)j * 6;
            last_status = c->status;
            if (c->status == NEMGPU_OK) {
                nbsucc++;
                if (nbsucc == 1 || crit[3] > best_crit[3]) {                                                     // :1676-1697
                    if (e->ncem()) HIPCHK(hipMemcpyAsync(e->best_lab, c->lab[c->cur], (size_t)e->n_total, hipMemcpyDeviceToDevice, e->stream));
                    else HIPCHK(hipMemcpyAsync(e->best_c, c->cbuf[c->cur], sizeof(float) * (size_t)e->n_total * k, hipMemcpyDeviceToDevice, e->stream));
                    for (int t = 0; t < 6; t++) best_crit[t] = crit[t];
                    best = next + j;
                    fill_result(c, &best_res);
                }
            }
        }
        if (e->libc()) {
            e->draws = E[valid - 1]->draws;                        // the stream as start `valid - 1` left it
            e->tie_heavy = e->tie_heavy || E[valid - 1]->tie_heavy;
        } else if (valid < M) {
            set_error("internal: lock-step starts out of order"); return NEMGPU_E_FUNCARG;
        }
        next += valid; e->rs_rounds++; e->rs_lockstep += valid; e->rs_redone += M - valid;
        if (e->libc()) {
            if (valid < M || E[M - 1]->draws != after_init[M - 1]) { voided_in_a_row++; clean_run = 0; if (voided_in_a_row >= 2) alone_for_good = true; width = 1; }
            else { voided_in_a_row = 0; clean_run += M; width = group; }
        }
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (nbsucc > 0) {
        if (e->ncem()) HIPCHK(hipMemcpyAsync(e->lab[e->cur], e->best_lab, (size_t)e->n_total, hipMemcpyDeviceToDevice, e->stream));
        else HIPCHK(hipMemcpyAsync(e->cbuf[e->cur], e->best_c, sizeof(float) * (size_t)e->n_total * k, hipMemcpyDeviceToDevice, e->stream));
        e->masks_valid = false; e->tables_fresh = false; e->density_fresh = false;
        if ((r = do_mstep(e)) || (!e->ncem() && (r = read_iter_flags(e)))) return r;                              // :1711
        e->status = NEMGPU_OK; e->emptyk = 0;
        e->iters = best_res.iters; e->converged = best_res.converged;
        if (res) { *res = best_res; res->status = NEMGPU_OK; res->tie_draws = e->draws; for (int t = 0; t < 6; t++) res->crit[t] = best_crit[t]; }
    } else if (res) {
        fill_result(e, res);
        res->status = last_status;
    }
    if (best_start) *best_start = best;
    HIPCHK(hipStreamSynchronize(e->stream));
    return NEMGPU_OK;
}

int nemgpu_glibc_random(uint32_t seed, int count, int32_t* out)
{
    if (!out || count < 0) return NEMGPU_E_FUNCARG;
    nemk::GlibcRandom rng(seed);
    for (int i = 0; i < count; i++) out[i] = (int32_t)rng.next();
    return NEMGPU_OK;
}

int nemgpu_density(nemgpu_engine* e)
{
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    if ((r = do_tables(e))) return r;
    if ((r = do_density(e))) return r;
    HIPCHK(hipStreamSynchronize(e->stream));
    return NEMGPU_OK;
}

int nemgpu_sweep(nemgpu_engine* e, float beta, int* rounds)
{
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    if ((r = ensure_state_buffers(e))) return r;
    if ((r = do_sweep(e, beta, rounds))) return r;
    e->cur = (e->cur + 1) % 3;
    e->masks_valid = false;
    return NEMGPU_OK;
}

int nemgpu_mstep(nemgpu_engine* e, int* emptyk)
{
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    if ((r = ensure_state_buffers(e))) return r;
    if ((r = do_mstep(e))) return r;
    if ((r = read_iter_flags(e))) return r;
    if (emptyk) *emptyk = e->h_iter()[FLAG_EMPTYK];
    return e->h_iter()[FLAG_EMPTYK] ? NEMGPU_W_EMPTYCLASS : NEMGPU_OK;
}

int nemgpu_criteria(nemgpu_engine* e, float crit6[6])
{
    if (!e || !crit6) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    return criteria(e, crit6);
}

// ---- multi-GPU (sharded) step pieces ----------------------------------------------------------
// Same kernels; the host driver (pangenomenem_amd/distributed.py) owns the all-gathered label arrays
// and the statistics buffer and runs the collectives between these calls.  Everything is
// asynchronous on the engine's stream; between nemgpu_shard_begin and nemgpu_shard_end the loop
// control lives on the device (k_ctrl), exactly like the single-GPU pipelined loop.
int nemgpu_stats_words(const nemgpu_engine* e) { return e ? e->k + e->k * e->d : 0; }

int nemgpu_shard_layout(nemgpu_engine* e, int world, int rank, int blk, int stride, int n_true)
{
    if (!e || world <= 0 || rank < 0 || rank >= world || blk <= 0 || stride < blk + 1 || n_true <= 0) return NEMGPU_E_FUNCARG;
    if (e->lo != rank * stride || e->n > blk || e->n_total != world * stride) {
        set_error("shard layout does not match the engine's slot range");
        return NEMGPU_E_ARG;
    }
    if (stride % 4 != 0) { set_error("shard layout: stride must be a multiple of 4 bytes"); return NEMGPU_E_ARG; }
    e->sh_world = world; e->sh_rank = rank; e->sh_blk = blk; e->sh_stride = stride; e->n_true = n_true;
    return NEMGPU_OK;
}

namespace {
void shard_sweep_args(nemgpu_engine* e, SweepArgs& a, float beta, int sweep_id)
{
    a = SweepArgs{};
    a.n_local = e->n; a.lo = e->lo; a.n_total = e->n_total; a.K = e->k; a.npad = e->npad;
    a.use_nei = (e->has_graph && beta != 0.0f) ? 1 : 0;
    a.nei_ptr = e->nei_ptr; a.nei_idx = e->nei_idx; a.nei_w = e->nei_w;
    a.beta = beta; a.pkfki = e->pkfki;
    a.tie_rule = e->cfg.tie_rule; a.tie_seed = e->cfg.tie_seed;
    a.sweep_id = sweep_id >= 0 ? (uint32_t)sweep_id : 0u;
    a.sweep_id_ptr = sweep_id >= 0 ? nullptr : e->sweep_next;     // -1: the device keeps count (pipelined loop)
    a.stop = e->stop_ptr;
    a.n_ranks = e->sh_world; a.slot_stride = e->sh_stride; a.slot_pad = e->sh_stride - e->sh_blk;
    a.fold_ticket = e->sweep_next + 32;
    a.exp_tab = (a.use_nei && e->ncem() && e->exp_ready && e->exp_beta == beta) ? e->exp_tab : nullptr;
    a.exp_tab_len = e->exp_need;
    if (e->libc()) { sweep_draw_args(e, a, e->stop_ptr == nullptr); a.rank_index = e->sh_rank; }
}
uint8_t* own_flag_byte(nemgpu_engine* e, uint8_t* labels) { return labels + (size_t)e->sh_rank * e->sh_stride + e->sh_blk; }
int shard_buf(const nemgpu_engine* e, const uint8_t* p) { for (int b = 0; b < 3; b++) if (e->shard_lab[b] == p) return b; return -1; }
// TIE_LIBC: the draw counts that go with a round's guess and output buffers -- per block (the engine's, by buffer) and
// per rank (in the buffers' block tails, all-gathered with the labels)
int shard_tie_args(nemgpu_engine* e, SweepArgs& a, const uint8_t* guess, uint8_t* out)
{
    if (!e->libc()) return NEMGPU_OK;
    const int g = shard_buf(e, guess), o = shard_buf(e, out);
    if (g < 0 || o < 0) { set_error("TIE_LIBC in the sharded path: a label array that nemgpu_shard_set_labels was not told about"); return NEMGPU_E_FUNCARG; }
    a.tie_cnt_guess = e->tie_cnt[g]; a.tie_cnt_out = e->tie_cnt[o];
    a.rank_tot_in = guess + e->sh_tot_off();
    a.rank_tot_out = out + (size_t)e->sh_rank * e->sh_stride + e->sh_tot_off();
    return NEMGPU_OK;
}
void shard_round1_args(nemgpu_engine* e, SweepArgs& a, float beta, int sweep_id, const uint8_t* labels_old_dev,
                       const uint8_t* labels_guess_dev, uint8_t* labels_out_dev)
{
    shard_sweep_args(e, a, beta, sweep_id);
    a.lab_old = labels_old_dev; a.lab_guess = labels_guess_dev; a.lab_out = labels_out_dev;
    a.flags = e->round_flags(1);
    a.flags_in = labels_guess_dev + e->sh_blk;                     // rank 0's flag byte; stride = slot_stride
    a.publish_byte = own_flag_byte(e, labels_out_dev); a.publish_ticket = e->sweep_next + 32;
    // the verifying round also says whether one of THIS rank's labels moved in the sweep (its guess against the old
    // partition): the byte behind its flag byte, gathered with it -- the convergence test then needs no pass over the
    // whole label array on every rank
    a.post_on = 1; a.post_from_guess = 1; a.post_skip_guess = 1; a.post_moved = 1; a.post_no_masks = 1;
    a.post_nw64 = e->nw64; a.post_mask = e->mask; a.post_flags = e->iter_flags();
}
}  // namespace

static int shard_check(nemgpu_engine* e)
{
    if (!e) return NEMGPU_E_FUNCARG;
    if (!e->ncem()) { set_error("the sharded path is NCEM-only (SURVEY.md 8e)"); return NEMGPU_E_FUNCARG; }
    if (e->sh_stride == 0) { set_error("nemgpu_shard_layout must be called first"); return NEMGPU_E_FUNCARG; }
    if (e->libc() && (e->shard_lab[0] == nullptr || e->sh_tot_off() + 4 > e->sh_stride)) {
        set_error("TIE_LIBC in the sharded path: nemgpu_shard_set_labels first (and a block tail of at least 12 bytes)");
        return NEMGPU_E_FUNCARG;
    }
    if (crit_test(e)) {
        // (the criterion is an i-ordered float sum over ALL families: no sharded reduction reproduces it, and the loop
        //  control of this path only knows the clas test -- a run configured with crit would never converge)
        set_error("the family-sharded path implements the convergence tests none and clas only");
        return NEMGPU_E_FUNCARG;
    }
    return NEMGPU_OK;
}

int nemgpu_shard_begin(nemgpu_engine* e)
{
    { const int cr = shard_check(e); if (cr) return cr; }
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    HIPCHK(hipMemsetAsync(e->ctrl(), 0, C_WORDS * sizeof(int), e->stream));
    if (e->libc()) {                                               // the stream's window and position, as every rank has them
        int r;
        if ((r = ensure_draw_window(e, e->draws, draw_need(e)))) return r;
        if ((r = publish_draw_ctl(e))) return r;
    }
    e->stop_ptr = e->ctrl() + C_STOP;
    return NEMGPU_OK;
}

// nemgpu_shard_begin for a batch that starts a run over: the reset (initial parameters back in place), the clearing of
// the loop control and the density tables in ONE launch (the head of the single engine's restart batches, enqueue_init)
// instead of a device-to-device copy, three fills and a tables launch
int nemgpu_shard_begin_restart(nemgpu_engine* e)
{
    { const int cr = shard_check(e); if (cr) return cr; }
    if (!e->have_matrix || !e->have_params) { set_error("matrix and parameters must be set first"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(e->device));
    int r;
    if ((r = reset_state(e, true))) return r;                      // host half; the device half is the launch below
    e->reset_pending = false;
    if ((r = clear_fault(e))) return r;
    FinishArgs t = finish_args(e, 0, nullptr);
    t.reset_prop = e->prop0; t.reset_center = e->center0; t.reset_disp = e->disp0;
    t.reset_ctrl = e->ctrl(); t.reset_ctrl_words = C_WORDS; t.reset_sweep_next = e->sweep_next;
    launch_finish(t, e->stream);
    HIPCHK(hipGetLastError());
    e->tables_fresh = true; e->density_fresh = false;
    if (e->libc()) {                                               // (srandom(seed): the stream starts over)
        if ((r = ensure_draw_window(e, e->draws, draw_need(e)))) return r;
        if ((r = publish_draw_ctl(e))) return r;
    }
    e->stop_ptr = e->ctrl() + C_STOP;
    return NEMGPU_OK;
}

// this rank's partial counts of the labels relaxation round 0 just produced (its class masks were built by that
// launch) -> stats_dev, normally the statistics tail of this rank's block in the label array gathered next
int nemgpu_shard_counts(nemgpu_engine* e, int32_t* stats_dev)
{
    if (!e || !stats_dev) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    launch_mstep_counts(e->k, e->d, e->nw64, e->xt, e->mask, stats_dev, e->stop_ptr, nullptr, e->stream);
    HIPCHK(hipGetLastError());
    return NEMGPU_OK;
}

// local class masks of the given labels + popcounts -> stats_dev (the explicit form of the above)
int nemgpu_shard_mstep_partial(nemgpu_engine* e, const uint8_t* labels_cur_dev, int32_t* stats_dev)
{
    if (!e || !labels_cur_dev || !stats_dev) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    launch_labels_post(e->n, e->lo, e->k, e->nw64, labels_cur_dev, nullptr, e->mask, e->iter_flags(), e->stop_ptr,
                       nullptr, e->stream);
    launch_mstep_counts(e->k, e->d, e->nw64, e->xt, e->mask, stats_dev, e->stop_ptr, nullptr, e->stream);
    HIPCHK(hipGetLastError());
    e->masks_valid = false;
    return NEMGPU_OK;
}

// [parameter update from the summed statistics] + density + relaxation round 0 (guess = old) + this rank's
// changed byte behind its label block (then: all-gather of labels_out)
int nemgpu_shard_estep_round0(nemgpu_engine* e, const int32_t* stats_dev, float beta, int sweep_id,
                              const uint8_t* labels_old_dev, uint8_t* labels_out_dev)
{
    if (!e || !labels_old_dev || !labels_out_dev) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    const bool fused = stats_dev != nullptr && e->fused_update();
    if (fused) {
        // parameter update (per block, from the summed counts) + density in one launch
        FinishArgs t = finish_args(e, 1, stats_dev);
        t.stats_ranks = e->sh_world; t.stats_rank_stride = e->sh_stride / 4;
        launch_density_fused(t, e->xws, e->n, e->npad, e->pkfki, e->logpkfki,
                             e->iter_flags() + FLAG_MOVED, kSweepFlagWords, e->stream);
        HIPCHK(hipGetLastError());
        e->tables_fresh = false;
        e->density_fresh = true;
        e->flags_clean = true;
    } else {
        if (stats_dev != nullptr) {
            FinishArgs t = finish_args(e, 1, stats_dev);
            t.stats_ranks = e->sh_world; t.stats_rank_stride = e->sh_stride / 4;
            launch_finish(t, e->stream);
            HIPCHK(hipGetLastError());
            e->tables_fresh = true;
            e->density_fresh = false;
        } else if ((r = do_tables(e))) return r;
        if (e->density_fresh) { if (!e->flags_clean && (r = clear_sweep_flags(e))) return r; }   // same parameters as the last E1: keep pkfki
        else if ((r = do_density(e))) return r;                    // also clears MOVED + the round flag window
    }
    // the blind sweep of a start (beta = 0, sweep 0, the parameters as given): a flag slot of its own and no "moved"
    // flag, so that the sweep behind it -- same densities -- needs no clearing of the flags in between
    const bool blind = stats_dev == nullptr && sweep_id == 0 && beta == 0.0f;
    SweepArgs a;
    shard_sweep_args(e, a, beta, sweep_id);
    a.lab_old = labels_old_dev; a.lab_guess = labels_old_dev; a.lab_out = labels_out_dev;
    a.flags = e->round_flags(blind ? kRoundCap - 1 : 0);
    // the same launch builds the class masks of its output (for nemgpu_shard_counts) and publishes the flag byte
    a.post_on = 1; a.post_from_guess = 0; a.post_moved = blind ? 0 : 1; a.post_nw64 = e->nw64; a.post_mask = e->mask;
    a.post_flags = e->iter_flags();                                // ("moved" byte: final when this is the sweep's only round)
    a.publish_byte = own_flag_byte(e, labels_out_dev); a.publish_ticket = e->sweep_next + 32;
    if ((r = shard_tie_args(e, a, labels_old_dev, labels_out_dev))) return r;
    launch_sweep(a, true, e->stream);
    HIPCHK(hipGetLastError());
    e->flags_clean = blind && e->flags_clean;
    return NEMGPU_OK;
}

// relaxation round 1 (guess = round 0's all-gathered output); returns at once on every rank when no rank
// changed a label in round 0 (then: all-gather of labels_out, whose content is unused in that case)
int nemgpu_shard_estep_round1(nemgpu_engine* e, float beta, int sweep_id, const uint8_t* labels_old_dev,
                              const uint8_t* labels_guess_dev, uint8_t* labels_out_dev)
{
    if (!e || !labels_old_dev || !labels_guess_dev || !labels_out_dev) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    SweepArgs a;
    shard_round1_args(e, a, beta, sweep_id, labels_old_dev, labels_guess_dev, labels_out_dev);
    { const int tr = shard_tie_args(e, a, labels_guess_dev, labels_out_dev); if (tr) return tr; }
    launch_sweep(a, true, e->stream);
    HIPCHK(hipGetLastError());
    return NEMGPU_OK;
}

// relaxation round 1 and this rank's partial counts of round 0's labels (nemgpu_shard_counts) in ONE launch where the
// shape has such a kernel (k_sweep_counts), else one after the other: the two do not depend on each other
int nemgpu_shard_estep_round1_counts(nemgpu_engine* e, float beta, int sweep_id, const uint8_t* labels_old_dev,
                                     const uint8_t* labels_guess_dev, uint8_t* labels_out_dev, int32_t* stats_dev)
{
    if (!e || !labels_old_dev || !labels_guess_dev || !labels_out_dev || !stats_dev) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    SweepArgs a;
    shard_round1_args(e, a, beta, sweep_id, labels_old_dev, labels_guess_dev, labels_out_dev);
    { const int tr = shard_tie_args(e, a, labels_guess_dev, labels_out_dev); if (tr) return tr; }
    static const bool merge = !(getenv("NEM_DIST_MERGE") && getenv("NEM_DIST_MERGE")[0] == '0');
    if (!merge || !launch_sweep_counts(a, e->k, e->d, e->nw64, e->xt, e->mask, stats_dev, e->stop_ptr, e->stream)) {
        launch_sweep(a, true, e->stream);
        launch_mstep_counts(e->k, e->d, e->nw64, e->xt, e->mask, stats_dev, e->stop_ptr, nullptr, e->stream);
    }
    HIPCHK(hipGetLastError());
    return NEMGPU_OK;
}

// ---- fuzzy NEM on several GPUs (SURVEY.md 8e's exact alternative) ----------------------------------------------
// The fuzzy M-step's sums are i-ordered float accumulators (nem_mod.c:1303-1313, 1677-1686, 1448-1458): no reduction
// over family shards reproduces them.  So the E-step is sharded over FAMILIES (this engine: rows [lo, hi), the
// memberships of every family in caller-owned arrays float[n_total][K] that the caller all-gathers between relaxation
// rounds) and the M-step over ORGANISMS (a second engine per rank that holds ALL families x its slice of the
// organisms: every chain is summed on one device over the whole, gathered membership matrix, in family order).
// Host-driven, one step per call; pangenomenem_amd/distributed.py:ShardedFuzzyNem drives it.

// families of the whole problem when n_total carries padding (proportions divide by it, nem_mod.c:456-465)
int nemgpu_shard_fuzzy_layout(nemgpu_engine* e, int n_true)
{
    if (!e || n_true <= 0 || n_true > e->n_total) return NEMGPU_E_FUNCARG;
    if (e->ncem()) { set_error("nemgpu_shard_fuzzy_*: the engine must be configured with algo = nem"); return NEMGPU_E_ARG; }
    e->n_true = n_true;
    return NEMGPU_OK;
}

// one relaxation round of the E-step sweep over this engine's families: c_out[lo..hi) from the guess c_guess (round 0:
// the old partition) -- the other families' rows of c_out are left alone (the caller's all-gather fills them).
// changed / nzero / firstzero (host): the round's flags (firstzero: global index of the first family whose densities
// all vanished, -1 if none).  Synchronises.
int nemgpu_shard_fuzzy_round(nemgpu_engine* e, float beta, int sweep_id, int round, const float* c_old_dev, const float* c_guess_dev,
                             float* c_out_dev, int* changed, int* nzero, int* firstzero)
{
    if (!e || !c_old_dev || !c_guess_dev || !c_out_dev || round < 0) return NEMGPU_E_FUNCARG;
    if (e->ncem()) { set_error("nemgpu_shard_fuzzy_round: the engine must be configured with algo = nem"); return NEMGPU_E_ARG; }
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    if ((r = ensure_state_buffers(e))) return r;
    const int slot = round % kRoundCap;
    HIPCHK(hipMemsetAsync(e->round_flags(slot), 0, FLAG_ROUND_STRIDE * sizeof(int), e->stream));
    SweepArgs a{};
    a.n_local = e->n; a.lo = e->lo; a.n_total = e->n_total; a.K = e->k; a.npad = e->npad;
    a.use_nei = (e->has_graph && beta != 0.0f) ? 1 : 0;
    a.nei_ptr = e->nei_ptr; a.nei_idx = e->nei_idx; a.nei_w = e->nei_w;
    a.beta = beta; a.pkfki = e->pkfki;
    a.tie_rule = e->cfg.tie_rule; a.tie_seed = e->cfg.tie_seed; a.sweep_id = (uint32_t)sweep_id;
    a.c_old = c_old_dev; a.c_guess = c_guess_dev; a.c_out = c_out_dev;
    a.flags = e->round_flags(slot);
    a.fold_ticket = e->sweep_next + 32;
    launch_sweep(a, false, e->stream);
    HIPCHK(hipGetLastError());
    e->flags_clean = false;
    HIPCHK(hipMemcpyAsync(e->flags_host, e->flags_dev, e->flag_words() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    if ((r = check_fault(e))) return r;
    const int* f = e->h_round(slot);
    if (changed) *changed = f[FLAG_CHANGED];
    if (nzero) *nzero = f[FLAG_NZERO];
    if (firstzero) *firstzero = f[FLAG_NZERO] > 0 ? e->n_total - f[FLAG_FIRSTZERO] : -1;
    return NEMGPU_OK;
}

// EstimSizes / EstimLaplaceCenters / EstimLaplaceIner for THIS engine's organisms on the memberships of all its
// families (c_dev: float[n][K], the whole gathered matrix for an organism-slice engine): class sizes, centres and
// inertia to device arrays of the caller's (K, K * d, K * d floats).  Dispersions are not derived here: InerToDisp
// needs every organism (nemgpu_shard_fuzzy_finish).
int nemgpu_shard_fuzzy_mstep_cols(nemgpu_engine* e, const float* c_dev, float* nbobs_out_dev, float* center_out_dev, float* iner_out_dev)
{
    if (!e || !c_dev || !nbobs_out_dev || !center_out_dev || !iner_out_dev) return NEMGPU_E_FUNCARG;
    if (e->ncem()) { set_error("nemgpu_shard_fuzzy_mstep_cols: the engine must be configured with algo = nem"); return NEMGPU_E_ARG; }
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    if ((r = ensure_state_buffers(e))) return r;
    launch_mstep_fuzzy(e->n, e->npad, e->k, e->d, e->xw, e->xt, e->nw64, c_dev + (size_t)e->lo * e->k,
                       e->fuzzy_chains ? e->fz_ct : nullptr, e->nbobs_k,
                       e->fz_in0, e->fz_in1, e->fz_inh, e->fz_lastz, e->fz_any1, e->center, e->iner, nullptr, e->stream,
                       e->fuzzy_chains == 2 ? e->fz_chk : nullptr, e->iter_flags() + FLAG_FAULT, e->fault_inject);
    HIPCHK(hipGetLastError());
    const size_t kd = (size_t)e->k * e->d;
    HIPCHK(hipMemcpyAsync(nbobs_out_dev, e->nbobs_k, sizeof(float) * e->k, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(center_out_dev, e->center, sizeof(float) * kd, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(iner_out_dev, e->iner, sizeof(float) * kd, hipMemcpyDeviceToDevice, e->stream));
    e->tables_fresh = false; e->density_fresh = false;
    return read_iter_flags(e);                                     // (waits; a stalled hand-over of the M-step is an error)
}

// the rest of EstimPara on the gathered statistics of ALL organisms (device arrays: K, K * D, K * D floats): InerToDisp*
// (nem_mod.c:922-1174), proportions, the empty-class test, the density tables -- then E1 on this engine's families.
// emptyk (host): 0, or the class (1..K) EstimLaplaceCenters found empty (nothing else was updated for it).
int nemgpu_shard_fuzzy_finish(nemgpu_engine* e, const float* nbobs_dev, const float* center_dev, const float* iner_dev, int* emptyk)
{
    if (!e || !nbobs_dev || !center_dev || !iner_dev) return NEMGPU_E_FUNCARG;
    if (e->ncem()) { set_error("nemgpu_shard_fuzzy_finish: the engine must be configured with algo = nem"); return NEMGPU_E_ARG; }
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    if ((r = ensure_state_buffers(e))) return r;
    const size_t kd = (size_t)e->k * e->d;
    HIPCHK(hipMemcpyAsync(e->nbobs_k, nbobs_dev, sizeof(float) * e->k, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->center, center_dev, sizeof(float) * kd, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->iner, iner_dev, sizeof(float) * kd, hipMemcpyDeviceToDevice, e->stream));
    launch_finish(finish_args(e, 2, nullptr), e->stream);
    HIPCHK(hipGetLastError());
    e->tables_fresh = true; e->density_fresh = false;
    if ((r = read_iter_flags(e))) return r;
    const int ek = e->h_iter()[FLAG_EMPTYK];
    if (emptyk) *emptyk = ek;
    if (ek != 0) return NEMGPU_W_EMPTYCLASS;                       // nem_alg.c:1831-1838: the E-step does not run
    return do_density(e);
}

// HasConverged's CVTEST_CLAS (nem_alg.c:2075-2089) over this engine's families: moved (host) = some membership moved
// by the threshold or more
int nemgpu_shard_fuzzy_moved(nemgpu_engine* e, const float* c_new_dev, const float* c_old_dev, int* moved)
{
    if (!e || !c_new_dev || !c_old_dev || !moved) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemsetAsync(e->iter_flags() + FLAG_MOVED, 0, sizeof(int), e->stream));
    launch_conv_fuzzy((size_t)e->n * e->k, c_new_dev + (size_t)e->lo * e->k, c_old_dev + (size_t)e->lo * e->k, e->cfg.cvthres,
                      e->iter_flags(), nullptr, nullptr, e->stream);
    HIPCHK(hipGetLastError());
    int r;
    if ((r = read_iter_flags(e))) return r;
    *moved = e->h_iter()[FLAG_MOVED];
    return NEMGPU_OK;
}

// convergence test over the whole label array + the device-side loop tests (k_ctrl logic)
int nemgpu_shard_finish_iteration(nemgpu_engine* e, float beta, int is_init, const uint8_t* labels_old_dev,
                                  const uint8_t* labels_q_dev, const uint8_t* labels_r_dev)
{
    if (!e || !labels_old_dev || !labels_q_dev || !labels_r_dev) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    CtrlArgs ca{};
    ca.ctrl = e->ctrl(); ca.iter_flags = e->iter_flags(); ca.round0 = e->round_flags(0); ca.n_rounds = 2;
    ca.param_fix = e->cfg.param_fix; ca.use_nei = beta != 0.0f ? 1 : 0; ca.cvtest = e->cfg.cvtest; ca.ncem = 1;
    ca.cvthres = e->cfg.cvthres; ca.sweep_next = e->sweep_next; ca.ticket = e->sweep_next + 32;
    ca.q_flags = labels_q_dev + e->sh_blk; ca.r_flags = labels_r_dev + e->sh_blk;
    ca.n_ranks = e->sh_world; ca.flag_stride = e->sh_stride; ca.is_init = is_init;
    if (is_init) ca.blind = e->round_flags(kRoundCap - 1);         // (the blind sweep's zero-density tally)
    // every rank's "one of my labels moved" byte came with the gather of the sweep's last round: round 1's, or -- no
    // neighbours to verify against (beta = 0) -- round 0's
    const bool two_rounds = beta != 0.0f || e->libc();             // (the reference's tie stream couples the sites of a sweep without neighbours too)
    ca.use_nei = two_rounds ? 1 : 0;
    ca.moved_bytes = is_init ? 0 : (two_rounds ? 1 : 2);
    if (e->libc()) { ca.draw_ctl = e->draw_ctl; ca.q_tot = labels_q_dev + e->sh_tot_off(); }
    launch_ctrl(ca, e->stream);
    HIPCHK(hipGetLastError());
    return NEMGPU_OK;
}

// one host-checked relaxation round (fallback when a sweep needs more than two rounds): returns whether
// THIS rank changed a label; the caller all-gathers labels_out and max-reduces the flag
int nemgpu_shard_round_sync(nemgpu_engine* e, float beta, int sweep_id, const uint8_t* labels_old_dev,
                            const uint8_t* labels_guess_dev, uint8_t* labels_out_dev, int* changed)
{
    if (!e || !labels_old_dev || !labels_guess_dev || !labels_out_dev || !changed) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    const int* saved = e->stop_ptr;
    e->stop_ptr = nullptr;
    SweepArgs a;
    shard_sweep_args(e, a, beta, sweep_id);
    e->stop_ptr = saved;
    a.lab_old = labels_old_dev; a.lab_guess = labels_guess_dev; a.lab_out = labels_out_dev;
    a.flags = e->round_flags(2);
    if (e->libc()) {
        // (the window of the stream at the host's position; the rank's draws of this round go to its block's tail:
        //  the last block of the round writes them with the flag byte)
        int r;
        if ((r = ensure_draw_window(e, e->draws, draw_need(e)))) return r;
        sweep_draw_args(e, a, true);
        a.publish_byte = own_flag_byte(e, labels_out_dev); a.publish_ticket = e->sweep_next + 32;
        if ((r = shard_tie_args(e, a, labels_guess_dev, labels_out_dev))) return r;
    }
    HIPCHK(hipMemsetAsync(e->round_flags(2), 0, FLAG_ROUND_STRIDE * sizeof(int), e->stream));
    launch_sweep(a, true, e->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(e->flags_host, e->flags_dev, e->flag_words() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    *changed = e->h_round(2)[FLAG_CHANGED] != 0;
    return NEMGPU_OK;
}

// TIE_LIBC in the sharded path, host side.  nemgpu_shard_set_labels: the driver's three all-gathered label arrays (the
// engine keeps each one's per-block draw counts).  nemgpu_shard_round_draws: after nemgpu_shard_round_sync -- this
// rank's draws in that round, and whether one of them fell outside the draw table (then the round is void on every
// rank: nemgpu_shard_grow_draws on all of them and the round again).  nemgpu_shard_book_draws: a sweep the host
// completed drew n numbers (the sum of the ranks' draws of its final round): the stream moves on.
int nemgpu_shard_set_labels(nemgpu_engine* e, const uint8_t* lab0, const uint8_t* lab1, const uint8_t* lab2)
{
    if (!e || !lab0 || !lab1 || !lab2) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    e->shard_lab[0] = lab0; e->shard_lab[1] = lab1; e->shard_lab[2] = lab2;
    alloc_for(e);
    for (int b = 0; b < 3; b++)
        if (!e->tie_cnt[b]) { const int r = dev_alloc(&e->tie_cnt[b], (size_t)e->n / 256 + 2); if (r) return r; }
    drop_graphs(e);
    return NEMGPU_OK;
}
int nemgpu_shard_round_draws(nemgpu_engine* e, int* draws, int* table_short)
{
    if (!e) return NEMGPU_E_FUNCARG;
    const int v = e->h_round(2)[FLAG_NTIES];
    if (draws) *draws = v & ((1 << 30) - 1);
    if (table_short) *table_short = (v >> 30) & 1;
    return NEMGPU_OK;
}
int nemgpu_shard_grow_draws(nemgpu_engine* e) { if (!e) return NEMGPU_E_FUNCARG; e->tie_heavy = true; return NEMGPU_OK; }
int nemgpu_shard_book_draws(nemgpu_engine* e, int n) { if (!e || n < 0) return NEMGPU_E_FUNCARG; e->draws += n; return NEMGPU_OK; }

// end of a batch: read the loop-control block back (one sync).  res->iters etc. are the batch's increments.
int nemgpu_shard_end_enqueue(nemgpu_engine* e)
{
    // last call of a batch's enqueue phase (may be inside a graph capture): copy the control block to the host
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    e->stop_ptr = nullptr;
    HIPCHK(hipMemcpyAsync(e->flags_host, e->flags_dev, e->flag_words() * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    return NEMGPU_OK;
}

int nemgpu_shard_end(nemgpu_engine* e, nemgpu_result* res, int* commits, int* need_rounds)
{
    // wait for the batch and report; call after nemgpu_shard_end_enqueue (or after replaying a captured batch)
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    e->stop_ptr = nullptr;
    HIPCHK(hipStreamSynchronize(e->stream));
    const int* c = e->h_ctrl();
    if (res) {
        res->status = c[C_STATUS]; res->iters = c[C_ITERS]; res->converged = c[C_CONVERGED]; res->emptyk = c[C_EMPTYK];
        res->zero_density_sites = c[C_NZERO];
        res->first_zero_density_site = c[C_NZERO] > 0 ? e->n_total - c[C_FIRSTZERO] : -1;
        res->sweep_rounds = c[C_SWEEP_ROUNDS];
    }
    if (commits) *commits = c[C_COMMITS];
    if (need_rounds) *need_rounds = c[C_NEED_ROUNDS];
    if (e->libc()) {
        e->draws += c[C_DRAWS];                                    // (the committed sweeps' draws: where the next one starts)
        for (int q = 0; q < 2; q++) if (e->h_round(q)[FLAG_NTIES] & (1 << 30)) e->tie_heavy = true;
    }
    e->flags_clean = false;
    return NEMGPU_OK;
}

// ---- RCCL called directly (no Python between the launches of a batch) ------------------------------------------
// The sharded EM needs one collective, an in-place all-gather of label blocks.  Issued through torch.distributed
// every call costs tens of microseconds of host time, more than the iteration's kernels; issued from here the whole
// batch -- kernels and all-gathers -- is enqueued by one C call.  librccl.so (the one PyTorch ships and has already
// loaded) is bound at run time; torch.distributed stays the bootstrap (it carries the ncclUniqueId) and the fallback.
struct nccl_uid_t { char internal[128]; };                                     // ncclUniqueId, rccl.h:40-43
namespace {
struct RcclApi {
    void* handle = nullptr;
    int (*get_unique_id)(nccl_uid_t*) = nullptr;                               // ncclGetUniqueId
    int (*comm_init_rank)(void**, int, nccl_uid_t, int) = nullptr;             // ncclCommInitRank(comm*, nranks, id, rank)
    int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*comm_destroy)(void*) = nullptr;
    int (*comm_count)(const void*, int*) = nullptr;                            // ncclCommCount
    int (*comm_abort)(void*) = nullptr;                                        // ncclCommAbort
    const char* (*error_string)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_fail(const char* what, int code)
{
    set_error(std::string(what) + " failed: " + (g_rccl.error_string ? g_rccl.error_string(code) : "?"));
    return NEMGPU_E_DEVICE;
}
}  // namespace

int nemgpu_rccl_open(const char* librccl_path)
{
    if (g_rccl.handle) return NEMGPU_OK;
    void* h = dlopen(librccl_path && librccl_path[0] ? librccl_path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { set_error(std::string("dlopen(librccl) failed: ") + dlerror()); return NEMGPU_E_DEVICE; }
    g_rccl.get_unique_id = (int (*)(nccl_uid_t*))dlsym(h, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (int (*)(void**, int, nccl_uid_t, int))dlsym(h, "ncclCommInitRank");
    g_rccl.all_gather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.comm_destroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.error_string = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.comm_count = (int (*)(const void*, int*))dlsym(h, "ncclCommCount");
    g_rccl.comm_abort = (int (*)(void*))dlsym(h, "ncclCommAbort");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.all_gather || !g_rccl.comm_destroy) {
        set_error("librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
        dlclose(h);
        return NEMGPU_E_DEVICE;
    }
    g_rccl.handle = h;
    return NEMGPU_OK;
}

int nemgpu_rccl_unique_id(uint8_t id128[128])
{
    if (!id128) return NEMGPU_E_FUNCARG;
    if (!g_rccl.handle) { set_error("nemgpu_rccl_open first"); return NEMGPU_E_FUNCARG; }
    nccl_uid_t id;
    int rc = g_rccl.get_unique_id(&id);
    if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, 128);
    return NEMGPU_OK;
}

void rccl_release(nemgpu_engine* e)
{
    if (e->rccl_comm && g_rccl.comm_destroy) (void)g_rccl.comm_destroy(e->rccl_comm);
    e->rccl_comm = nullptr;
}

// collective: every rank of the job calls it with the id rank 0 produced
int nemgpu_rccl_attach(nemgpu_engine* e, const uint8_t id128[128], int world, int rank)
{
    if (!e || !id128 || world <= 0 || rank < 0 || rank >= world) return NEMGPU_E_FUNCARG;
    if (!g_rccl.handle) { set_error("nemgpu_rccl_open first"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(e->device));
    nccl_uid_t id;
    memcpy(id.internal, id128, 128);
    void* comm = nullptr;
    int rc = g_rccl.comm_init_rank(&comm, world, id, rank);
    if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
    e->rccl_comm = comm;
    return NEMGPU_OK;
}

// ranks of the engine's native communicator as RCCL itself reports them (ncclCommCount); 0 when none is attached
int nemgpu_rccl_ranks(const nemgpu_engine* e)
{
    if (!e || !e->rccl_comm || !g_rccl.comm_count) return 0;
    int n = 0;
    if (g_rccl.comm_count(e->rccl_comm, &n) != 0) return 0;
    return n;
}

namespace {
int rccl_allgather_blocks(nemgpu_engine* e, uint8_t* buf)
{
    if (e->sh_world == 1) return NEMGPU_OK;                        // (a rank alone holds every block already)
    const size_t stride = (size_t)e->sh_stride;
    int rc = g_rccl.all_gather(buf + (size_t)e->sh_rank * stride, buf, stride, /*ncclUint8*/ 1, e->rccl_comm, e->stream);
    if (rc != 0) return rccl_fail("ncclAllGather", rc);
    return NEMGPU_OK;
}
}  // namespace

// One in-place all-gather of `stride`-byte blocks through the engine's own communicator, waited for with a deadline:
// the caller (every rank, collectively) compares the result with the same gather through torch.distributed before
// trusting the native path with the EM.  On a timeout the communicator is aborted and detached (the engine then has
// no native path) and NEMGPU_E_DEVICE is returned.
int nemgpu_rccl_selftest(nemgpu_engine* e, uint8_t* buf_dev, int stride, int timeout_ms)
{
    if (!e || !buf_dev || stride <= 0) return NEMGPU_E_FUNCARG;
    if (!e->rccl_comm) { set_error("nemgpu_rccl_attach first"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(e->device));
    int rc = g_rccl.all_gather(buf_dev + (size_t)e->sh_rank * stride, buf_dev, (size_t)stride, /*ncclUint8*/ 1,
                               e->rccl_comm, e->stream);
    if (rc != 0) return rccl_fail("ncclAllGather (self-test)", rc);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(e->stream);
        if (q == hipSuccess) return NEMGPU_OK;
        if (q != hipErrorNotReady) { set_error(std::string("self-test: ") + hipGetErrorString(q)); return NEMGPU_E_DEVICE; }
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (g_rccl.comm_abort) (void)g_rccl.comm_abort(e->rccl_comm);
    e->rccl_comm = nullptr;
    set_error("native RCCL self-test timed out: communicator aborted");
    return NEMGPU_E_DEVICE;
}

// One whole batch of the sharded EM -- [the two initial sweeps +] n_iters iterations, all-gathers included --
// enqueued on the engine's stream without returning to the caller in between (what ShardedNem._enqueue_batch
// does through torch.distributed).  lab[3]: the all-gathered label arrays; stats_off: byte offset of a rank's
// statistics inside its block; base: buffer of the current partition (2 after an init).
static int shard_batch_body(nemgpu_engine* e, int with_init, int n_iters, int base, float beta, int want_stats,
                            uint8_t* lab0, uint8_t* lab1, uint8_t* lab2, int stats_off)
{
    uint8_t* L[3] = {lab0, lab1, lab2};
    const bool use_nei = beta != 0.0f || e->libc();                // two rounds: neighbours to verify against, or the shared draw stream
    auto own_stats = [&](uint8_t* buf) { return (int32_t*)(buf + (size_t)e->sh_rank * e->sh_stride + stats_off); };
    auto all_stats = [&](uint8_t* buf) { return (const int32_t*)(buf + stats_off); };
    int r;
#define NEM_TRY(call) do { if ((r = (call)) != NEMGPU_OK) return r; } while (0)
    NEM_TRY(with_init ? nemgpu_shard_begin_restart(e) : nemgpu_shard_begin(e));
    if (with_init) {
        NEM_TRY(nemgpu_shard_estep_round0(e, nullptr, 0.0f, 0, L[0], L[1]));       // blind sweep
        NEM_TRY(rccl_allgather_blocks(e, L[1]));
        NEM_TRY(nemgpu_shard_estep_round0(e, nullptr, beta, 1, L[1], L[2]));
        if (want_stats && !use_nei) NEM_TRY(nemgpu_shard_counts(e, own_stats(L[2])));
        NEM_TRY(rccl_allgather_blocks(e, L[2]));
        if (use_nei) {
            if (want_stats) NEM_TRY(nemgpu_shard_estep_round1_counts(e, beta, 1, L[1], L[2], L[0], own_stats(L[0])));
            else NEM_TRY(nemgpu_shard_estep_round1(e, beta, 1, L[1], L[2], L[0]));
            NEM_TRY(rccl_allgather_blocks(e, L[0]));
        }
        NEM_TRY(nemgpu_shard_finish_iteration(e, beta, 1, L[1], L[2], L[0]));
    }
    for (int j = 0; j < n_iters; j++) {
        const int P = (base + j) % 3, Q = (P + 1) % 3, R = (P + 2) % 3;
        const int S = use_nei ? Q : P;                                             // where P's statistics were gathered
        NEM_TRY(nemgpu_shard_estep_round0(e, want_stats ? all_stats(L[S]) : nullptr, beta, -1, L[P], L[Q]));
        if (want_stats && !use_nei) NEM_TRY(nemgpu_shard_counts(e, own_stats(L[Q])));
        NEM_TRY(rccl_allgather_blocks(e, L[Q]));
        if (use_nei) {
            if (want_stats) NEM_TRY(nemgpu_shard_estep_round1_counts(e, beta, -1, L[P], L[Q], L[R], own_stats(L[R])));
            else NEM_TRY(nemgpu_shard_estep_round1(e, beta, -1, L[P], L[Q], L[R]));
            NEM_TRY(rccl_allgather_blocks(e, L[R]));
        }
        NEM_TRY(nemgpu_shard_finish_iteration(e, beta, 0, L[P], L[Q], L[R]));
    }
#undef NEM_TRY
    return nemgpu_shard_end_enqueue(e);
}

int nemgpu_shard_enqueue_batch(nemgpu_engine* e, int with_init, int n_iters, int base, float beta, int want_stats,
                               uint8_t* lab0, uint8_t* lab1, uint8_t* lab2, int stats_off)
{
    if (!e || !lab0 || !lab1 || !lab2 || n_iters < 0 || base < 0 || base > 2) return NEMGPU_E_FUNCARG;
    if (!e->rccl_comm) { set_error("nemgpu_rccl_attach first"); return NEMGPU_E_FUNCARG; }
    // A rank alone has no collective inside its batches: they go out as hipGraphs of the library's own, captured the
    // second time a shape is enqueued (as the single engine's batches are).  With more ranks the batch holds RCCL
    // calls and is issued launch by launch -- one C call either way.
    nemgpu_engine::ShardGraph* slot = nullptr;
    if (e->sh_world == 1 && e->use_graphs && !e->libc()) {        // (TIE_LIBC: the stream's position is a launch argument)
        uint32_t bb; memcpy(&bb, &beta, 4);
        //  Everything that decides WHICH launches the body issues is part of the key: besides the call's
        // arguments, the host state the body branches on.  A batch that does not start with the restart launch first
        // flushes a pending reset -- outside the capture, below -- so that state is the same at capture and at replay.
        if (with_init == 0) { const int fr = flush_reset(e); if (fr) return fr; }
        else { HIPCHK(hipSetDevice(e->device)); const int cf = clear_fault(e); if (cf) return cf; }   // (a restart clears a reported fault: not inside a capture)
        const std::vector<uint64_t> desc = {(uint64_t)with_init, (uint64_t)n_iters, (uint64_t)base, (uint64_t)bb, (uint64_t)want_stats,
                                            (uint64_t)(uintptr_t)lab0, (uint64_t)(uintptr_t)lab1, (uint64_t)(uintptr_t)lab2, (uint64_t)stats_off,
                                            (uint64_t)((with_init ? 0 : ((e->tables_fresh ? 1 : 0) | (e->density_fresh ? 2 : 0) | (e->flags_clean ? 4 : 0) |
                                                                         (e->masks_valid ? 8 : 0))))};
        uint64_t key = 1469598103934665603ull;
        for (uint64_t v : desc) key = (key ^ v) * 1099511628211ull;
        for (auto& g : e->shard_graphs) if (g.key == key && g.desc == desc) { slot = &g; break; }
        if (slot == nullptr) {
            if (e->shard_graphs.size() >= 64) drop_graphs(e);
            e->shard_graphs.push_back({key, desc, 0, nullptr});
            slot = &e->shard_graphs.back();
        }
        if (slot->exec != nullptr) {
            { const int cr = shard_check(e); if (cr) return cr; }
            HIPCHK(hipSetDevice(e->device));
            // the host half of what the captured body did (nemgpu_shard_begin_restart / the round functions): a replay
            // only launches the graph -- a reset left pending here would later copy the initial parameters over the
            // estimated ones
            if (with_init) {
                if (!e->have_matrix || !e->have_params) { set_error("matrix and parameters must be set first"); return NEMGPU_E_FUNCARG; }
                const int rr = reset_state(e, true);
                if (rr) return rr;
                e->reset_pending = false;                          // (the device half is the graph's first launch)
            }
            HIPCHK(hipGraphLaunch(slot->exec, e->stream));
            e->tables_fresh = slot->post_tables_fresh; e->density_fresh = slot->post_density_fresh;
            e->flags_clean = slot->post_flags_clean; e->masks_valid = slot->post_masks_valid;
            e->n_replayed++;
            e->stop_ptr = nullptr;
            return NEMGPU_OK;
        }
        if (slot->asked++ >= 1 || e->capture_first) {
            HIPCHK(hipSetDevice(e->device));
            HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
            const int r = shard_batch_body(e, with_init, n_iters, base, beta, want_stats, lab0, lab1, lab2, stats_off);
            hipGraph_t graph = nullptr;
            const hipError_t cerr = hipStreamEndCapture(e->stream, &graph);
            hipGraphExec_t exec = nullptr;
            if (r == NEMGPU_OK && cerr == hipSuccess && graph != nullptr && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                if (graph) (void)hipGraphDestroy(graph);
                slot->exec = exec;
                slot->post_tables_fresh = e->tables_fresh; slot->post_density_fresh = e->density_fresh;
                slot->post_flags_clean = e->flags_clean; slot->post_masks_valid = e->masks_valid;
                e->n_captured++;
                HIPCHK(hipGraphLaunch(exec, e->stream));
                return NEMGPU_OK;
            }
            if (graph) (void)hipGraphDestroy(graph);
            if (getenv("NEM_MI355X_DEBUG"))
                fprintf(stderr, "[nem] sharded batch capture failed: body rc %d (%s), end-capture %s\n", r, g_last_error.c_str(), hipGetErrorString(cerr));
            (void)hipGetLastError();
            e->use_graphs = false;                                 // plain launches from now on
            if (r != NEMGPU_OK) return r;
        }
    }
    e->n_plain++;
    return shard_batch_body(e, with_init, n_iters, base, beta, want_stats, lab0, lab1, lab2, stats_off);
}

// host-side completion of an iteration whose sweep needed extra rounds: count it like k_ctrl would have
int nemgpu_shard_set_sweep_number(nemgpu_engine* e, int next_sweep)
{
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)e->sweep_next, next_sweep, 1, e->stream));
    return NEMGPU_OK;
}

// Criteria of the partition the last sweep started from (buffer cur+2: the sweep's ping/pong leave it intact),
// evaluated with the current densities -- what the reference logs "after the M-step" of an iteration
// (WriteLogCrit at the top of ComputePartitionNEM, nem_alg.c:2361).
int nemgpu_criteria_previous(nemgpu_engine* e, float crit6[6])
{
    if (!e || !crit6) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    return criteria(e, crit6, (e->cur + 2) % 3);
}

int nemgpu_set_partition(nemgpu_engine* e, const float* c_nk)
{
    // test hook: load a partition (row-major [n_total x k], HOST) as the current state
    if (!e || !c_nk) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    if ((r = ensure_state_buffers(e))) return r;
    if (e->ncem()) {
        std::vector<uint8_t> lab((size_t)e->n_total);
        for (int i = 0; i < e->n_total; i++) {
            int best = 0;
            for (int k = 1; k < e->k; k++) if (c_nk[(size_t)i * e->k + k] > c_nk[(size_t)i * e->k + best]) best = k;
            lab[i] = (uint8_t)best;
        }
        HIPCHK(copy_sync(e, e->lab[e->cur], lab.data(), lab.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemsetAsync(e->tie_cnt[e->cur], 0, ((size_t)e->n / 256 + 2) * sizeof(int), e->stream));
    } else {
        HIPCHK(copy_sync(e, e->cbuf[e->cur], c_nk, sizeof(float) * (size_t)e->n_total * e->k, hipMemcpyHostToDevice));
    }
    e->masks_valid = false;
    return NEMGPU_OK;
}

int nemgpu_get_labels(nemgpu_engine* e, uint8_t* labels)
{
    if (!e || !labels) return NEMGPU_E_FUNCARG;
    if (!e->ncem() || !e->lab[e->cur]) { set_error("labels exist only for ncem runs"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(copy_sync(e, labels, e->lab[e->cur] + e->lo, (size_t)e->n, hipMemcpyDeviceToHost));
    for (int i = 0; i < e->n; i++) labels[i] &= 0x7F;              // (bit 7: the site drew, TIE_LIBC)
    return NEMGPU_OK;
}

// the final partition and parameters in one round trip: every piece is copied into one pinned block, one wait
int nemgpu_get_results(nemgpu_engine* e, float* prop, float* center, float* disp, float* nbobs_k, float* c_nk)
{
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    const size_t n = (size_t)e->n, k = (size_t)e->k;
    if (c_nk) {
        if (e->ncem() ? !e->lab[e->cur] : !e->cbuf[e->cur]) { set_error("no partition yet"); return NEMGPU_E_FUNCARG; }
    }
    const size_t b_part = c_nk ? (e->ncem() ? (n + 3) & ~(size_t)3 : sizeof(float) * n * k) : 0;
    const bool want_par = prop || center || disp || nbobs_k;
    const size_t o_prop = b_part, total = o_prop + (want_par ? sizeof(float) * e->par_words : 0);
    char* st = nullptr; size_t got = 0;
    std::vector<char> own;
    const bool pinned = total <= kStageMax && pool_get(e->device, true, total, &st, &got) == hipSuccess;
    if (!pinned) { (void)hipGetLastError(); own.resize(total); st = own.data(); }
    hipError_t err = hipSuccess;
    auto D2H = [&](size_t off, const void* src, size_t bytes) {
        if (err == hipSuccess && bytes) err = hipMemcpyAsync(st + off, src, bytes, hipMemcpyDeviceToHost, e->stream);
    };
    if (c_nk) {
        if (e->ncem()) D2H(0, e->lab[e->cur] + e->lo, n);
        else D2H(0, e->cbuf[e->cur] + (size_t)e->lo * k, sizeof(float) * n * k);
    }
    if (want_par) D2H(o_prop, e->prop, sizeof(float) * e->par_words);   // prop | center | disp | nbobs_k: one block
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err == hipSuccess) result_unpack(e, st, b_part, prop, center, disp, nbobs_k, c_nk);
    if (pinned) pool_put(e->device, true, st, got);
    HIPCHK(err);
    return NEMGPU_OK;
}

int nemgpu_get_partition(nemgpu_engine* e, float* c_nk)
{
    if (!e || !c_nk) return NEMGPU_E_FUNCARG;
    return nemgpu_get_results(e, nullptr, nullptr, nullptr, nullptr, c_nk);
}

int nemgpu_get_params(nemgpu_engine* e, float* prop, float* center, float* disp, float* nbobs_k)
{
    if (!e) return NEMGPU_E_FUNCARG;
    return nemgpu_get_results(e, prop, center, disp, nbobs_k, nullptr);
}

int nemgpu_get_density(nemgpu_engine* e, double* pkfki_nk, float* logpkfki_nk)
{
    // device layout is class-major [k][npad]; hand back the reference's row-major [n][k]
    if (!e) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t m = (size_t)e->k * e->npad;
    if (pkfki_nk) {
        std::vector<double> t(m);
        HIPCHK(copy_sync(e, t.data(), e->pkfki, sizeof(double) * m, hipMemcpyDeviceToHost));
        for (int i = 0; i < e->n; i++) for (int k = 0; k < e->k; k++) pkfki_nk[(size_t)i * e->k + k] = t[(size_t)k * e->npad + i];
    }
    if (logpkfki_nk) {
        std::vector<float> t(m);
        HIPCHK(copy_sync(e, t.data(), e->logpkfki, sizeof(float) * m, hipMemcpyDeviceToHost));
        for (int i = 0; i < e->n; i++) for (int k = 0; k < e->k; k++) logpkfki_nk[(size_t)i * e->k + k] = t[(size_t)k * e->npad + i];
    }
    return NEMGPU_OK;
}

// Kernel-duration probe for bench.py: `reps` launches of the E1 density kernel on the current parameters,
// each bracketed by HIP events recorded on the engine's stream; returns the average duration and the
// algorithmic bytes one launch moves.
int nemgpu_profile_density(nemgpu_engine* e, int reps, double* avg_ms, double* algorithmic_bytes_per_launch,
                           int* used_fused_kernel)
{
    if (!e || reps <= 0) return NEMGPU_E_FUNCARG;
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    if ((r = do_tables(e))) return r;
    if (!e->ev0) { HIPCHK(hipEventCreate(&e->ev0)); HIPCHK(hipEventCreate(&e->ev1)); }
    double total = 0.0;
    // measure the kernel the EM loop actually launches: the fused one (parameter update + density) when it applies
    const bool fused = !e->cfg.param_fix && e->ncem() && e->fused_update() && e->iters > 0;
    for (int i = 0; i < reps; i++) {
        HIPCHK(hipEventRecord(e->ev0, e->stream));
        if (fused) {
            launch_density_fused(finish_args(e, 1, e->stats), e->xws, e->n, e->npad, e->pkfki, e->logpkfki,
                                 e->iter_flags() + FLAG_MOVED, kSweepFlagWords, e->stream);
            HIPCHK(hipGetLastError());
        } else if ((r = do_density(e))) return r;
        HIPCHK(hipEventRecord(e->ev1, e->stream));
        HIPCHK(hipEventSynchronize(e->ev1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        total += ms;
    }
    if (avg_ms) *avg_ms = total / reps;
    if (used_fused_kernel) *used_fused_kernel = fused ? 1 : 0;
    if (algorithmic_bytes_per_launch) {
        // E1 per launch: the bit-packed matrix once, the (k,d) tables once, pk*fk (f64) and log (f32) out
        *algorithmic_bytes_per_launch = (double)e->n * e->wf * 4.0 + (double)e->k * e->d * 24.0 +
                                        (double)e->n * e->k * 12.0;
    }
    return NEMGPU_OK;
}

// E1 as a lock-step batch launches it: the density kernels of `count` engines (same shape) zipped into ONE launch
// (k_density_b, problem = blockIdx.z), `reps` such launches back to back between one pair of HIP events on the first
// engine's stream.  avg_ms: per launch (all members); bytes: algorithmic bytes of one launch (all members).  The engines
// must have run (their parameters and tables are what the last M-step left).
int nemgpu_profile_density_many(nemgpu_engine** engines, int count, int reps, double* avg_ms, double* algorithmic_bytes_per_launch)
{
    if (!engines || count <= 0 || reps <= 0) return NEMGPU_E_FUNCARG;
    nemgpu_engine* lead = engines[0];
    HIPCHK(hipSetDevice(lead->device));
    int r;
    std::vector<Recorder> recs((size_t)count);
    double bytes = 0.0;
    for (int m = 0; m < count; m++) {
        nemgpu_engine* e = engines[m];
        if (!e || e->device != lead->device) { set_error("profile_density_many: engines of one device"); return NEMGPU_E_FUNCARG; }
        if ((r = flush_reset(e)) || (r = do_tables(e))) return r;
        HIPCHK(hipStreamSynchronize(e->stream));
        set_recorder(&recs[m]);
        r = do_density(e);
        set_recorder(nullptr);
        if (r) return r;
        if (recs[m].ops.size() != 1) { set_error("profile_density_many: one density launch per engine expected"); return NEMGPU_E_INTERNAL; }
        const OpRecord& o = recs[m].ops[0]; const OpRecord& o0 = recs[0].ops[0];
        if (o.kind != o0.kind || o.variant != o0.variant || o.block != o0.block || o.gy != o0.gy || o.nbytes != o0.nbytes) {
            set_error("profile_density_many: the engines' density launches differ in shape"); return NEMGPU_E_FUNCARG;
        }
        bytes += (double)e->n * e->wf * 4.0 + (double)e->k * e->d * 24.0 + (double)e->n * e->k * 12.0;
    }
    const OpRecord& o0 = recs[0].ops[0];
    const int stride = (o0.nbytes + 15) & ~15;
    const size_t gx_off = (size_t)count * stride, total = gx_off + (((size_t)count * sizeof(int) + 15) & ~(size_t)15);
    if ((r = zip_reserve(lead, total))) return r;
    nemgpu_engine::ZipContext* z = zip_context(lead);
    unsigned max_gx = 0;
    for (int m = 0; m < count; m++) {
        memcpy(z->zip_host + (size_t)m * stride, recs[m].ops[0].args, (size_t)o0.nbytes);
        reinterpret_cast<int*>(z->zip_host + gx_off)[m] = (int)recs[m].ops[0].gx;
        max_gx = std::max(max_gx, recs[m].ops[0].gx);
    }
    HIPCHK(hipMemcpyAsync(z->zip_dev, z->zip_host, total, hipMemcpyHostToDevice, lead->stream));
    if (!lead->ev0) { HIPCHK(hipEventCreate(&lead->ev0)); HIPCHK(hipEventCreate(&lead->ev1)); }
    auto once = [&]() {
        launch_zipped(o0.kind, o0.variant, count, z->zip_dev, stride, reinterpret_cast<const int*>(z->zip_dev + gx_off), max_gx, o0.gy,
                      o0.block, lead->stream);
    };
    once();                                                    // (warm)
    HIPCHK(hipEventRecord(lead->ev0, lead->stream));
    for (int i = 0; i < reps; i++) once();
    HIPCHK(hipEventRecord(lead->ev1, lead->stream));
    HIPCHK(hipEventSynchronize(lead->ev1));
    HIPCHK(hipGetLastError());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, lead->ev0, lead->ev1));
    if (avg_ms) *avg_ms = (double)ms / reps;
    if (algorithmic_bytes_per_launch) *algorithmic_bytes_per_launch = bytes;
    return NEMGPU_OK;
}

// Kernel-duration probe for bench.py's roofline.kernels[]: the three kernels of a solo NCEM iteration -- E1 (with the
// parameter update when the loop launches it that way), one relaxation round of the E-step sweep, the M-step counts --
// each launched `reps` times back to back between ONE pair of HIP events on the engine's stream (an event pair per
// launch costs ~3 us, as much as the shorter kernels), on the state the engine is in (the launches are idempotent: same
// inputs, same outputs; the partition is not advanced).  avg_ms[3], bytes[3]: average duration and algorithmic bytes
// per launch of {E1, sweep round, counts}; which[0]: 1 when E1 is the fused kernel.
int nemgpu_profile_kernels(nemgpu_engine* e, int reps, double avg_ms[3], double bytes[3], int which[1])
{
    if (!e || reps <= 0 || !avg_ms || !bytes) return NEMGPU_E_FUNCARG;
    if (!e->ncem() || e->lo != 0 || e->hi != e->n_total) { set_error("the kernel probe times a whole NCEM problem on one engine"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(e->device));
    { const int fr = flush_reset(e); if (fr) return fr; }
    int r;
    if ((r = ensure_state_buffers(e))) return r;
    if ((r = do_tables(e))) return r;
    if (!e->masks_valid) { if ((r = do_labels_post(e, e->cur, -1))) return r; }
    if (!e->ev0) { HIPCHK(hipEventCreate(&e->ev0)); HIPCHK(hipEventCreate(&e->ev1)); }
    const bool fused = !e->cfg.param_fix && e->fused_update() && e->iters > 0;
    auto timed = [&](auto&& launch, double* out) -> int {
        launch();                                                  // (one untimed launch: code object, caches)
        HIPCHK(hipEventRecord(e->ev0, e->stream));
        for (int i = 0; i < reps; i++) launch();
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e->ev1, e->stream));
        HIPCHK(hipEventSynchronize(e->ev1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
        *out = (double)ms / reps;
        return NEMGPU_OK;
    };
    // M-step counts first: they leave the statistics the fused E1 derives its parameters from
    if ((r = timed([&] { launch_mstep_counts(e->k, e->d, e->nw64, e->xt, e->mask, e->stats, nullptr, nullptr, e->stream); }, &avg_ms[2]))) return r;
    if ((r = timed([&] {
            if (fused) launch_density_fused(finish_args(e, 1, e->stats), e->xws, e->n, e->npad, e->pkfki, e->logpkfki,
                                            e->iter_flags() + FLAG_MOVED, kSweepFlagWords, e->stream);
            else launch_density(finish_args(e, 0, nullptr), e->xws, e->n, e->npad, e->pkfki, e->logpkfki, e->iter_flags() + FLAG_MOVED,
                                kSweepFlagWords, e->stream);
        }, &avg_ms[0]))) return r;
    e->density_fresh = true; e->flags_clean = true;
    if (fused) e->tables_fresh = false;
    // one relaxation round: round 0 of a sweep from the current partition (guess = old), into the next buffer
    {
        SweepCtx c;
        const uint32_t keep = e->sweep_counter;
        if ((r = sweep_setup(e, e->cfg.beta, c, true))) return r;
        e->sweep_counter = keep;
        const int P = e->cur, Q = (e->cur + 1) % 3;
        c.a.lab_old = e->lab[P]; c.a.lab_guess = e->lab[P]; c.a.lab_out = e->lab[Q];
        c.a.tie_cnt_guess = e->tie_cnt[P]; c.a.tie_cnt_out = e->tie_cnt[Q];
        c.a.flags = e->round_flags(0); c.a.fold_ticket = e->sweep_next + 32;
        if ((r = timed([&] { launch_sweep(c.a, true, e->stream); }, &avg_ms[1]))) return r;
        e->flags_clean = false;
    }
    const double n = e->n, d = e->d, k = e->k, nnz = e->nnz;
    bytes[0] = n * e->wf * 4.0 + k * d * 24.0 + n * k * 12.0;             // bit matrix once, (k,d) tables, pk*fk f64 + log f32 out
    bytes[1] = 8.0 * nnz + 4.0 * (n + 1) + nnz + 8.0 * n * k + 2.0 * n;   // CSR idx + w, row pointers, neighbour labels, densities, labels in/out
    bytes[2] = d * e->nw64 * 8.0 + k * e->nw64 * 8.0 + 4.0 * (k + k * d); // organism bit rows once, class masks once, counts out
    if (which) which[0] = fused ? 1 : 0;
    return NEMGPU_OK;
}

// `reps` in-place all-gathers of the sharded EM's label blocks (stride bytes per rank) back to back through the engine's
// own communicator, between one pair of HIP events: what ONE of the iteration's two collectives costs.  Collective.
int nemgpu_rccl_time_allgather(nemgpu_engine* e, uint8_t* buf_dev, int reps, double* avg_ms)
{
    if (!e || !buf_dev || reps <= 0 || !avg_ms) return NEMGPU_E_FUNCARG;
    if (!e->rccl_comm) { set_error("nemgpu_rccl_attach first"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(e->device));
    if (!e->ev0) { HIPCHK(hipEventCreate(&e->ev0)); HIPCHK(hipEventCreate(&e->ev1)); }
    const size_t stride = (size_t)e->sh_stride;
    auto gather = [&]() { return g_rccl.all_gather(buf_dev + (size_t)e->sh_rank * stride, buf_dev, stride, /*ncclUint8*/ 1, e->rccl_comm, e->stream); };
    for (int i = 0; i < 3; i++) { const int rc = gather(); if (rc != 0) return rccl_fail("ncclAllGather", rc); }
    HIPCHK(hipEventRecord(e->ev0, e->stream));
    for (int i = 0; i < reps; i++) { const int rc = gather(); if (rc != 0) return rccl_fail("ncclAllGather", rc); }
    HIPCHK(hipEventRecord(e->ev1, e->stream));
    HIPCHK(hipEventSynchronize(e->ev1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->ev0, e->ev1));
    *avg_ms = (double)ms / reps;
    return NEMGPU_OK;
}

// FETCH_SIZE calibration helper (profiles/README.md): one launch that reads `bytes` of freshly written device
// memory with E1's dword-per-lane pattern.  Run it under `rocprofv3 --pmc FETCH_SIZE` and compare.
int nemgpu_calibrate_fetch(size_t bytes, int reps)
{
    if (bytes < 4096 || reps <= 0) return NEMGPU_E_FUNCARG;
    uint32_t* buf = nullptr; uint32_t* sink = nullptr;
    HIPCHK(hipMalloc((void**)&buf, bytes));
    HIPCHK(hipMalloc((void**)&sink, 64));
    HIPCHK(hipMemset(buf, 0x5A, bytes));
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < reps; i++) launch_calib_read(buf, bytes / 4, sink, nullptr);
    hipError_t err = hipDeviceSynchronize();
    (void)hipFree(buf); (void)hipFree(sink);
    HIPCHK(err);
    return NEMGPU_OK;
}

// Re-target the engine to another HIP stream (e.g. the capturing stream of a torch.cuda.graph).
int nemgpu_set_fast_forward(nemgpu_engine* e, int on)
{
    if (!e) return NEMGPU_E_FUNCARG;
    const int mode = on < 0 ? -1 : (on != 0);
    if (e->ff_mode != mode) {
        HIPCHK(hipStreamSynchronize(e->stream));
        drop_graphs(e);                                            // captured launches carry the old setting
        e->ff_mode = mode;
        e->density_fresh = false;
        e->tables_fresh = false;                                   // (the increment tables are built with the density tables)
    }
    return NEMGPU_OK;
}

int nemgpu_ff_table(double l1, double l0, uint32_t* q0_256, uint32_t* q1_256)
{
    if (!q0_256 || !q1_256) return NEMGPU_E_FUNCARG;
    for (int E = 0; E < 256; E++) nemk::ff_entry(l1, -l0, E, q0_256[E], q1_256[E]);
    return NEMGPU_OK;
}

// capture_on_first != 0: a batch shape of the pipelined loop is captured into a hipGraph the FIRST time it is
// enqueued (default: the second time -- a nem() call's engine enqueues most shapes once and capturing costs more
// than it saves unless the batch is replayed).  Benchmarks prime every shape of their timed region with it.
int nemgpu_set_graph_policy(nemgpu_engine* e, int capture_on_first)
{
    if (!e) return NEMGPU_E_FUNCARG;
    e->capture_first = capture_on_first != 0;
    return NEMGPU_OK;
}

// out[0] batches enqueued as plain launches, out[1] batches captured + instantiated (one-off cost), out[2] graph
// replays, out[3] sweeps the host had to finish round by round (more than the two enqueued relaxation rounds)
int nemgpu_graph_counters(const nemgpu_engine* e, int out[4])
{
    if (!e || !out) return NEMGPU_E_FUNCARG;
    out[0] = e->n_plain; out[1] = e->n_captured; out[2] = e->n_replayed; out[3] = e->n_host_rounds;
    return NEMGPU_OK;
}

int nemgpu_sweep_counters(const nemgpu_engine* e, int out[4])
{
    if (!e || !out) return NEMGPU_E_FUNCARG;
    out[0] = e->n_fused; out[1] = e->n_fused_failed; out[2] = e->fused_sweep ? 1 : 0; out[3] = 0;
    return NEMGPU_OK;
}

int nemgpu_random_start_counters(const nemgpu_engine* e, int out[4])
{
    if (!e || !out) return NEMGPU_E_FUNCARG;
    out[0] = e->rs_rounds; out[1] = e->rs_lockstep; out[2] = e->rs_alone; out[3] = e->rs_redone;
    return NEMGPU_OK;
}

int nemgpu_sweep_phases(unsigned long long out64[64]) { return out64 && nemk::sweep_phases_read(out64) == 0 ? NEMGPU_OK : NEMGPU_E_FUNCARG; }

int nemgpu_set_stream(nemgpu_engine* e, void* hip_stream)
{
    if (!e || !hip_stream) return NEMGPU_E_FUNCARG;
    if (e->own_stream && e->stream) { (void)hipStreamSynchronize(e->stream); (void)hipStreamDestroy(e->stream); }
    e->stream = (hipStream_t)hip_stream;
    e->own_stream = false;
    drop_graphs(e);
    return NEMGPU_OK;
}

}  // extern "C"

// ---- test hooks for the exact segmented chains (nem_chain.hpp) ----
// mode 0: plain sequential loop; 1: the host emulation of the device procedure.  No GPU needed.
extern "C" float nemgpu_repeat_add_host(float x, long long times, int mode)
{
    if (mode == 2) return (x > 0.0f && x < 16777216.0f && x == (float)(uint32_t)x) ? nemk::ff_repeat_add_u24((uint32_t)x, times) : nemk::ff_repeat_add(x, times);
    if (mode != 0) return nemk::ff_repeat_add(x, times);
    volatile float s = 0.0f;
    for (long long j = 0; j < times; j++) s = s + x;
    return s;
}

extern "C" float nemgpu_chain_host(const double* x, long long n, float init, int mode)
{
    return mode == 0 ? nemchain::run_sequential(x, n, init) : nemchain::run_segmented(x, n, init);
}

// the device procedure on device `device`; returns NEMGPU_OK or an error code
extern "C" int nemgpu_chain_device(const double* x, long long n, float init, int device, float* out)
{
    if (x == nullptr || out == nullptr || n < 0) { set_error("nemgpu_chain_device: bad argument"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(device));
    double* dx = nullptr; float* dout = nullptr;
    HIPCHK(hipMalloc(&dx, (size_t)std::max<long long>(n, 1) * sizeof(double)));
    HIPCHK(hipMalloc(&dout, sizeof(float)));
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (n > 0) HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice, st));
    nemk::launch_chain_debug(dx, n, init, dout, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout, sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    (void)hipStreamDestroy(st); (void)hipFree(dx); (void)hipFree(dout);
    return NEMGPU_OK;
}

// ---- test hooks for the piecewise d-ordered chain of non-negative multiples of 1/2 (nem_halfsum.hpp) ----
// mode 0: the plain sequential loop; mode w = 1 .. 16: the host emulation of the device procedure with w wavefronts
// (stepped: how many of the n adds it took for real).  No GPU needed.
extern "C" float nemgpu_halfsum_host(const float* x, int n, int mode, int* stepped)
{
    if (stepped) *stepped = mode == 0 ? n : 0;
    if (x == nullptr || n <= 0 || mode < 0 || mode > nemk::kHsMaxWaves) return 0.0f;
    return mode == 0 ? nemk::halfsum_plain(x, n) : nemk::halfsum_host(x, n, mode, stepped);
}

// the device procedure (n <= 8192; waves = 16: the block of k_finish, 1: one wavefront) on device `device`
extern "C" int nemgpu_halfsum_device(const float* x, int n, int waves, int device, float* out)
{
    if (x == nullptr || out == nullptr || n < 0 || n > 8192 || (waves != 1 && waves != 16)) { set_error("nemgpu_halfsum_device: bad argument"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(device));
    float* dx = nullptr; float* dout = nullptr;
    HIPCHK(hipMalloc(&dx, (size_t)std::max(n, 1) * sizeof(float)));
    HIPCHK(hipMalloc(&dout, 2 * sizeof(float)));
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (n > 0) HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    nemk::launch_halfsum_debug(dx, n, waves, dout, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout, sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    (void)hipStreamDestroy(st); (void)hipFree(dx); (void)hipFree(dout);
    return NEMGPU_OK;
}

// where the device procedure spends its time (in-kernel clock, 100 MHz): us[0..3] = prefix sums, classification + walk,
// scan + hand-over, ordered pass; us[4] = the same chain as plain dependent adds on one lane
extern "C" int nemgpu_halfsum_profile(const float* x, int n, int waves, int device, double us[5])
{
    if (x == nullptr || us == nullptr || n < 0 || n > 8192 || (waves != 1 && waves != 16)) { set_error("nemgpu_halfsum_profile: bad argument"); return NEMGPU_E_FUNCARG; }
    HIPCHK(hipSetDevice(device));
    float* dx = nullptr; float* dout = nullptr; long long* dst = nullptr;
    HIPCHK(hipMalloc(&dx, (size_t)std::max(n, 1) * sizeof(float)));
    HIPCHK(hipMalloc(&dout, 2 * sizeof(float)));
    HIPCHK(hipMalloc(&dst, 6 * sizeof(long long)));
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (n > 0) HIPCHK(hipMemcpyAsync(dx, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice, st));
    long long h[6];
    for (int rep = 0; rep < 3; rep++) nemk::launch_halfsum_debug(dx, n, waves, dout, st, dst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(h, dst, sizeof h, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < 5; i++) us[i] = (double)(h[i + 1] - h[i]) * 0.01;
    (void)hipStreamDestroy(st); (void)hipFree(dx); (void)hipFree(dout); (void)hipFree(dst);
    return NEMGPU_OK;
}
