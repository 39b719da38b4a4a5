/*
 * nem_mi355x.h -- C ABI of the MI355X-native NEM partitioning engine.
 *
 * Two layers, both plain C (pointers and sizes, no torch/C++ types):
 *
 *  1. The DROP-IN entry `nem()`: same symbol, signature, argument meaning, file formats and
 *     return codes as the reference's only FFI entry point
 *         /root/reference/ppanggolin/NEM/nem_exe.h:23-35   (declaration)
 *         /root/reference/ppanggolin/NEM/nem_exe.c:239-704 (definition)
 *         /root/reference/ppanggolin/NEM/nem.pyx:1-14      (the Cython `cpdef` binding PPanGGOLiN uses)
 *     It reads <Fname>.str/.dat/.nei/.m, runs the EM loop on the GPU, writes <Fname>.uf|.cf, .mf,
 *     .stderr and (dolog) .log.
 *
 *  2. The in-memory engine `nemgpu_*`: the same EM loop without the ASCII files, for callers that
 *     already hold the presence/absence matrix (benchmarks, tests, the multi-GPU host driver).
 *     It replaces the in-memory path  ClassifyByNem()  nem_alg.h:10-18 / nem_alg.c:546-584.
 *
 * Every entry point fails loudly (non-zero return + message in nemgpu_last_error()) when no HIP
 * device is usable: there is no CPU fallback in this library.
 */
#ifndef NEM_MI355X_H
#define NEM_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * 1. Drop-in entry (reference: nem_exe.h:23-35).
 *
 *   Fname          base path; inputs <Fname>.str .dat .nei .m, outputs <Fname>.uf|.cf .mf .stderr .log
 *   nk             number of classes K (> 0)
 *   algo           "nem" | "ncem"            ("gem" is not reachable from PPanGGOLiN; rejected)
 *   beta           MRF weight of the neighbourhood term
 *   convergence    "none" | "clas" | "crit"
 *   convergence_th threshold: "clas" max |c - c_old| < th; "crit" |(M - M_old) / M| < th (nem_alg.c:2075-2105)
 *   format         "hard" (.cf) | "fuzzy" (.uf)
 *   it_max         maximum number of EM iterations (>= 0)
 *   dolog          non-zero: messages to <Fname>.stderr and the reference's iteration log <Fname>.log (criteria
 *                  before / after every E-step and all parameters per iteration; NEM_MI355X_LOG=0: header only)
 *   model_family   "bern" (the only family PPanGGOLiN uses; "norm"/"lapl" rejected)
 *   proportion     "p_" | "pk"
 *   dispersion     "s__" | "sk_" | "s_d" | "skd"
 *   init_mode      2 (INIT_PARAM_FILE, nem_typ.h:218: start from <Fname>.m) or 1 (INIT_RANDOM, :217: 50 random
 *                  starts, best by criterion M, <Fname>.m not read; what PPanGGOLiN's partition_shell uses,
 *                  ppanggolin.py:1207); modes 0, 3, 4 rejected
 *
 * Return value (ExitET, lib_io.h:22-34): 0 ok, 1 empty class (no output files, like the
 * reference), 2 bad arguments, 3 file error, 4 memory, 5 GPU/system error, 6 internal error.
 * ------------------------------------------------------------------------------------------ */
int nem(const char* Fname,
        const int nk,
        const char* algo,
        const float beta,
        const char* convergence,
        const float convergence_th,
        const char* format,
        const int it_max,
        const int dolog,
        const char* model_family,
        const char* proportion,
        const char* dispersion,
        const int init_mode);

/* ------------------------------------------------------------------------------------------
 * 2. In-memory engine.
 * ------------------------------------------------------------------------------------------ */
typedef struct nemgpu_engine nemgpu_engine;

/* numeric values follow the reference enums (nem_typ.h:121-128, 191-205, 271-277) */
enum { NEMGPU_ALGO_NEM = 0, NEMGPU_ALGO_NCEM = 1 };
enum { NEMGPU_DISP___ = 0, NEMGPU_DISP_K_ = 1, NEMGPU_DISP__D = 2, NEMGPU_DISP_KD = 3 };
enum { NEMGPU_PROP__ = 0, NEMGPU_PROP_K = 1 };
/* CRIT: HasConverged's CVTEST_CRIT (nem_alg.c:2090-2105) on criterion M, starting from 0 like a reference run
   without a log (Criteria = {0}, nem_exe.c:264); CRIT_LOGGED: starting from the criterion of the initial partition,
   like a reference run with dolog (WriteLogCrit, nem_alg.c:1980, 2398) */
enum { NEMGPU_CV_NONE = 0, NEMGPU_CV_CLAS = 1, NEMGPU_CV_CRIT = 2, NEMGPU_CV_CRIT_LOGGED = 3 };
/* NCEM tie rule (ComputeMAP, nem_alg.c:590-645; numeric values of TIE_RANDOM / TIE_FIRST follow TieET).
   LIBC = the reference's TIE_RANDOM: kmaxes[random() % (nequal+1)] on glibc's random() after srandom(tie_seed)
   (nem_exe.c:621; the reference seeds with time(NULL), :353), drawn in site order exactly as the sequential
   sweep draws them -- with the seed fixed, labels equal the reference's even where classes tie.
   FIRST = TIE_FIRST.  HASH: counter-based, kmaxes[mix32(seed, sweep, site) % (nequal+1)] (stateless; the
   family-sharded multi-GPU path uses it). */
enum { NEMGPU_TIE_LIBC = 0, NEMGPU_TIE_FIRST = 1, NEMGPU_TIE_HASH = 2 };
/* status codes = StatusET (nem_typ.h:106-117) */
enum { NEMGPU_OK = 0, NEMGPU_W_EMPTYCLASS = 2, NEMGPU_E_ARG = 3, NEMGPU_E_MEMORY = 4,
       NEMGPU_E_FILEIN = 5, NEMGPU_E_FILEOUT = 6, NEMGPU_E_FILE = 7, NEMGPU_E_FUNCARG = 8,
       NEMGPU_E_DEVICE = 9,
       /* a kernel reported that it could not finish its work (e.g. a producer/consumer hand-over of the fuzzy M-step
          that never completed): the results of the call are not to be used.  nem() returns 6 (EXIT_E_BUG). */
       NEMGPU_E_INTERNAL = 10 };

typedef struct {
    int   algo;        /* NEMGPU_ALGO_*  */
    float beta;
    int   disper;      /* NEMGPU_DISP_*  */
    int   propor;      /* NEMGPU_PROP_*  */
    int   cvtest;      /* NEMGPU_CV_*    */
    float cvthres;
    int   it_max;
    int   param_fix;   /* .m flag 2: never re-estimate parameters (nem_alg.c:1806) */
    int   tie_rule;    /* NEMGPU_TIE_*   */
    uint32_t tie_seed;
} nemgpu_config;

typedef struct {
    int   status;      /* NEMGPU_OK or NEMGPU_W_EMPTYCLASS */
    int   iters;       /* completed EM iterations */
    int   converged;
    int   emptyk;      /* 1..K, 0 if none */
    int   zero_density_sites;   /* E-step updates that hit cumnum == 0 (nem_alg.c:2603-2613) */
    int   first_zero_density_site;
    int   sweep_rounds;         /* total relaxation rounds spent in E-step sweeps */
    float crit[6];     /* D G U M L Z (ComputeCrit, nem_alg.c:2678-2757) */
    double loop_seconds;        /* wall time of the EM iteration loop (host clock, synchronised) */
    int   tie_draws;            /* TIE_LIBC: random() draws consumed so far (a .cf writer continues the stream there) */
} nemgpu_result;

const char* nemgpu_last_error(void);

/* Number of usable HIP devices (0 when none, and in a process forked from one that had already used the GPU). */
int nemgpu_device_count(void);
/* The device a call that was not given one runs on -- what the drop-in nem() uses.  NEM_MI355X_DEVICE = <index>: that
   device (anything but a valid index is NEMGPU_E_ARG); NEM_MI355X_DEVICE = auto: processes spread over the node's
   devices, LOCAL_RANK (a launcher's) when set, else the process id, modulo the device count (PPanGGOLiN runs its
   chunks in a multiprocessing.Pool, ppanggolin.py:1039-1095); unset: the calling thread's current HIP device, i.e.
   the one the embedding application selected (device 0 when it selected none).  Fails with NEMGPU_E_DEVICE, before
   any HIP call, in a forked child of a process that had used the GPU. */
int nemgpu_default_device(int* device);

/* Create an engine for n_total families x d organisms, k classes on `device`.
   [site_lo, site_hi) is the family shard this engine owns (0, n_total for a single GPU).
   `hip_stream` is a hipStream_t to run on, or NULL for an engine-owned stream. */
int nemgpu_create(nemgpu_engine** out, int n_total, int d, int k, int site_lo, int site_hi,
                  int device, void* hip_stream);
void nemgpu_destroy(nemgpu_engine* e);

/* Presence/absence rows of the owned shard, one byte per cell (0/1), row-major
   [(site_hi-site_lo) x d], HOST memory.  Bit-packs, uploads and builds both device layouts. */
int nemgpu_set_matrix_bytes(nemgpu_engine* e, const uint8_t* x_host);
/* Same, family-major bit-packed rows: word w of row i holds organisms 32w..32w+31 (bit b =
   organism 32w+b), ceil(d/32) words per row, HOST memory.  Bits above organism d-1 in a row's last word are
   padding: the library clears them in its own copy (the caller's buffer is not written), whatever they hold. */
int nemgpu_set_matrix_bits(nemgpu_engine* e, const uint32_t* xbits_host);
/* Neighbourhood graph of the owned shard in CSR, .nei file order inside a row; neighbour
   indices are GLOBAL family indices (0-based).  ptr has (site_hi-site_lo)+1 entries.  HOST memory. */
int nemgpu_set_graph(nemgpu_engine* e, const int32_t* ptr, const int32_t* idx, const float* w);
/* Initial parameters (the .m file content): prop[k], center[k*d], disp[k*d].  HOST memory. */
int nemgpu_set_params(nemgpu_engine* e, const float* prop, const float* center, const float* disp);
int nemgpu_configure(nemgpu_engine* e, const nemgpu_config* cfg);

/* Whole run: INIT_PARAM_FILE start + EM loop + final criteria (single-GPU engines only). */
int nemgpu_run(nemgpu_engine* e, nemgpu_result* res);

/* Several whole runs (nemgpu_run each) in LOCK STEP: every step of the EM -- M-step counts, density, relaxation
   rounds, criteria -- is ONE launch for all `count` problems (problem = blockIdx.z; every loop kernel has a twin that
   reads an array of argument blocks), and the host synchronises once per batch of iterations for everybody.  What
   PPanGGOLiN's chunk loop (ppanggolin.py:1045-1086) needs: many independent problems of one kind, each far too small
   to fill the chip.  The engines must be complete (matrix, graph, parameters, configuration), whole problems, on one
   device; sizes and configurations may differ (members whose launch sequences differ are grouped).  Each problem's
   result is bit-identical to its nemgpu_run.  results: `count` entries. */
int nemgpu_run_many(nemgpu_engine** engines, int count, nemgpu_result* results);

/* Many whole problems from host arrays to host arrays in ONE call -- PPanGGOLiN's chunk loop
   (ppanggolin.py:1045-1086) when the chunks are arrays: `workers` threads of the library build the engines (bit
   packing, asynchronous uploads), the calling thread runs each `group` of them in lock step (nemgpu_run_many) as soon
   as it is complete -- the run waits for its members' uploads by event, their device layouts ride in its first step,
   their partitions and parameters come back in one block with it -- and the workers unpack the results and recycle the
   engines while later groups are being built.  From four groups on (and six workers) the groups are dealt to two such
   runners on the device (NEM_MI355X_RUNNERS=n: n runners; 1: never split).  Every problem's result equals its own
   nemgpu_run; every element of a problem's out_* arrays is written when its rc is NEMGPU_OK.  Returns the first
   failure (each problem's own status is in rc). */
typedef struct {
    int n, d, k;
    const uint8_t*  x_bytes;     /* [n][d] values 0/1 ...                                   */
    const uint32_t* x_bits;      /* ... or [n][ceil(d/32)] bit rows: exactly one of the two */
    const int32_t*  nei_ptr;     /* CSR neighbourhood graph [n+1] (NULL: none) */
    const int32_t*  nei_idx;
    const float*    nei_w;
    const float *prop, *center, *disp;                                  /* initial parameters [k], [k][d], [k][d] */
    float *out_prop, *out_center, *out_disp, *out_nbobs_k, *out_c;      /* results, any may be NULL; out_c [n][k] */
    nemgpu_result result;
    int rc;
} nemgpu_problem;
int nemgpu_solve_many(nemgpu_problem* problems, int count, const nemgpu_config* cfg, int device, int workers, int group);

/* The same over several devices of this process -- independent chunk problems side by side, the reference's own form of
   parallelism (multiprocessing.Pool over 500-organism chunks, ppanggolin.py:1039-1095) with the devices in the role of
   the pool's workers: the lock-step groups are dealt round-robin over `devices` (nemgpu_deal_groups; an entry may
   repeat), every device runs its share on a thread of its own with its own worker threads (workers / n_devices each),
   streams, lock-step contexts and resource pool.  Every problem's result equals its own nemgpu_run. */
int nemgpu_solve_many_devices(nemgpu_problem* problems, int count, const nemgpu_config* cfg, const int* devices, int n_devices,
                              int workers, int group);
/* slot_of_problem[i] = index into the device list of the device that solves problem i (host arithmetic only) */
int nemgpu_deal_groups(int count, int group, int n_devices, int* slot_of_problem);

/* ---- The chunks of partition()'s voting loop, formed on the device ----------------------------------------------
   The reference solves a pangenome of more than 500 organisms as many NEM problems, each on a random sample of the
   organisms (ppanggolin.py:1045-1086: `orgs = sample(organisms, chunck_size)`), and writes every sample's input files
   from ONE graph (__write_nem_input_files, ppanggolin.py:821-930): columns = the sampled organisms in sample order
   (:850); families without a sampled organism dropped, the others numbered in the graph's order (:849-852); an edge's
   weight = the number of sampled organisms that carry the adjacency, an edge none carries dropped (:866-878), a
   family's neighbours in the order the master lists them.  nemgpu_master_create puts that ONE pangenome on the device:
   the presence/absence bit rows xbits[n][ceil(d/32)] (family-major, bit o of a row = organism o), the graph in CSR
   (nei_ptr[n + 1], nei_idx[nnz]) and, per directed edge, the bit set of the organisms that carry it
   (edge_bits[nnz][ceil(d/32)]).  HOST memory, copied once. */
typedef struct nemgpu_master nemgpu_master;
int nemgpu_master_create(nemgpu_master** out, int device, int n, int d, const uint32_t* xbits, const int32_t* nei_ptr,
                         const int32_t* nei_idx, const uint32_t* edge_bits);
void nemgpu_master_destroy(nemgpu_master* m);
/* One sample.  in: organisms[dc] (indices into the master's, the chunk's column order).  out: n = families with at least
   one sampled organism, nnz = directed edges of the chunk's graph; optional arrays (NULL: not wanted): keep[ceil(n_master
   / 64)] (bit i: master family i is in the chunk -- the chunk's family j is the j-th set bit), labels[n] (NCEM: class of
   the chunk's family j; room for n_master bytes), out_prop[k], out_center[k][dc], out_disp[k][dc], out_nbobs_k[k]. */
typedef struct {
    const int32_t* organisms;
    int dc;
    int n, nnz;
    uint64_t* keep;
    uint8_t* labels;
    float *out_prop, *out_center, *out_disp, *out_nbobs_k;
    nemgpu_result result;
    int rc;
} nemgpu_chunk;
/* All samples in ONE call: the device decides which families and edges each sample keeps (one pass over all samples,
   one wait), then every sample goes through nemgpu_solve_many's pipeline -- `workers` builder threads, lock-step groups
   of `group` problems, results unpacked while later groups are built -- with its matrix rows, lane order and graph
   written by the device straight into its engine's buffers: per sample only the initial parameters (k values per kind:
   prop[k], and ONE centre and ONE dispersion per class for every organism, as PPanGGOLiN's default .m has them,
   ppanggolin.py:893-901) and the results cross PCIe.  Every sample's result equals nemgpu_solve_many on the same
   sample formed on the host (tests/test_gpu_chunks.py). */
int nemgpu_solve_chunks(nemgpu_master* m, nemgpu_chunk* chunks, int count, int k, const float* prop, const float* center_k,
                        const float* disp_k, const nemgpu_config* cfg, int workers, int group);

/* Whole run from random starts (the reference's init_mode = INIT_RANDOM, RandNemAlgo nem_alg.c:1574-1742): n_starts
   starts (the reference uses 50), centres drawn from the data with the reference's generator -- glibc random()
   after srandom(seed), restated in csrc/nem_rng.hpp -- best start by criterion M, EstimPara on the best partition.
   With the same seed and tie-free data the result equals the reference's (which seeds with time(NULL)).
   best_start: 0-based index of the chosen start, -1 if every start ended with an empty class. */
int nemgpu_run_random(nemgpu_engine* e, int n_starts, uint32_t seed, nemgpu_result* res, int* best_start);
/* The same run with what the reference writes to <Fname>.log for INIT_RANDOM (nem_alg.c:1632-1636, 1662-1669: "Random
   initialization %d :", the start's line 0, then NemAlgo's line per iteration): `fn` is called on the calling thread for
   every event, in the reference's order -- start by start, though up to 64 starts run in lock step underneath (their
   lines are kept and handed over when the round is through; more starts, the `crit` convergence test or
   NEM_MI355X_BATCH_STARTS_LOGGED=0: one start after the other, events as they happen).  The pointers of an event are
   host arrays that are valid during the call only. */
typedef struct nemgpu_log_event {
    int kind;                     /* NEMGPU_LOG_START | _LINE | _EMPTY */
    int start;                    /* 0-based start */
    int iter;                     /* _LINE: 0 = the start's initial partition, else the iteration; _EMPTY: the iteration */
    int emptyk;                   /* _EMPTY: the class (1..K) */
    const float* crit_before;     /* _LINE: D,G,U,M,L,Z of the partition the E-step sweep started from (WriteLogCrit, :2361) */
    const float* crit_after;      /* _LINE: ... of the partition it left (:2398) */
    const float* prop;            /* _LINE: K */
    const float* center;          /* _LINE: K*D */
    const float* disp;            /* _LINE: K*D */
    const float* nbobs_k;         /* _LINE, iter > 0: K class sizes of this iteration's EstimPara; _EMPTY: EstimSizes'
                                     sizes of the partition that emptied a class (the next start's line 0 prints them,
                                     nothing resets NbObs_KD between starts); _LINE, iter 0: NULL */
} nemgpu_log_event;
enum { NEMGPU_LOG_START = 0, NEMGPU_LOG_LINE = 1, NEMGPU_LOG_EMPTY = 2 };
typedef void (*nemgpu_log_fn)(const nemgpu_log_event* ev, void* user);
/* nemgpu_run (INIT_PARAM_FILE: ComputePartitionFromPara + NemAlgo, nem_alg.c:1160, 1746-1879) with what the reference writes
   to <Fname>.log when dolog is set (WriteLogCrit, nem_alg.c:2361, 2398): `fn` gets a _LINE event for the initial partition
   (iter 0) and one per EM iteration, or _EMPTY for the iteration that emptied a class; `start` is 0.  The iterations run
   pipelined, seven per wait, and the criteria of a batch's iterations are evaluated together afterwards (NCEM with the
   `clas` / `none` tests; otherwise, or with NEM_MI355X_LOG_BATCHED=0, one iteration per wait as nemgpu_iterate_logged).
   res->crit: the criteria of the final partition when the last line carried them, else crit[0] = NaN (the caller asks
   nemgpu_criteria, after nemgpu_mstep + nemgpu_density if no iteration ran, nem_alg.c:1845-1852). */
int nemgpu_run_logged(nemgpu_engine* e, nemgpu_result* res, nemgpu_log_fn fn, void* user);
int nemgpu_run_random_logged(nemgpu_engine* e, int n_starts, uint32_t seed, nemgpu_result* res, int* best_start,
                             nemgpu_log_fn fn, void* user);
/* Test hook: the first `count` values random() returns after srandom(seed), from the restated generator. */
int nemgpu_glibc_random(uint32_t seed, int count, int32_t* out);

/* Step-level entry points (same kernels; used by tests, bench.py and the multi-GPU host driver). */
int nemgpu_init_partition(nemgpu_engine* e);              /* ComputePartitionFromPara(Needinit=1) */
int nemgpu_iterate(nemgpu_engine* e, int n_iters, nemgpu_result* res);  /* up to n_iters EM iterations */
int nemgpu_reset(nemgpu_engine* e);                       /* back to the initial parameters, zero partition */
/* One EM iteration -- or, with_init != 0, the start: reset + the two initial sweeps, no iteration -- plus what a line
   of the reference's log needs (the criteria of the partition the sweep started from and of the new one, the
   parameters), all in one stream submission and one wait (HOST buffers; the parameter pointers may be NULL).
   res->loop_seconds covers the criteria as well. */
int nemgpu_iterate_logged(nemgpu_engine* e, int with_init, nemgpu_result* res, float crit_before[6], float crit_after[6],
                          float* prop, float* center, float* disp, float* nbobs_k);
int nemgpu_restart_iterate(nemgpu_engine* e, int n_iters, nemgpu_result* res);  /* reset + init + n iterations, one pipeline */
int nemgpu_density(nemgpu_engine* e);                     /* E1 only */
int nemgpu_sweep(nemgpu_engine* e, float beta, int* rounds);  /* E2 only (one full Gauss-Seidel sweep) */
int nemgpu_mstep(nemgpu_engine* e, int* emptyk);          /* M only */
int nemgpu_criteria(nemgpu_engine* e, float crit6[6]);    /* C1 only */
/* C1 on the partition the last E-step sweep started from, with the current densities: the "after the M-step"
   criteria of the reference's <Fname>.log (WriteLogCrit, nem_alg.c:2361, 2620-2646) */
int nemgpu_criteria_previous(nemgpu_engine* e, float crit6[6]);

/* Multi-GPU step pieces (NCEM; families sharded across engines in contiguous blocks).  The host
   driver (pangenomenem_amd/distributed.py) owns the all-gathered label arrays (device memory) and
   the statistics buffer and runs the collectives (RCCL through torch.distributed) between these
   calls.  All calls are asynchronous on the engine's stream; between nemgpu_shard_begin and
   nemgpu_shard_end the loop tests run on the device, as in the single-GPU pipelined loop.

   Label arrays are uint8[world * stride]: rank r owns slots [r*stride, r*stride + blk) (its families,
   in order); the byte at r*stride + blk carries its "a label changed in this round" flag, the byte behind it "one
   of my labels moved in this sweep" (the verifying round's, or round 0's when there are no neighbours to verify
   against: every rank then runs the convergence test on the gathered bytes) and, further
   behind (4-byte aligned, the driver picks the offset), its partial M-step statistics, so ONE
   all-gather per relaxation round moves labels, flags and statistics together -- there is no separate
   all-reduce.  The engine is created with n_total = world*stride, site_lo = rank*stride,
   site_hi = site_lo + (families of the shard) and the graph's neighbour indices are slot indices.
   stats: int32[k + k*d] = { N_k, S1[k][j] = #{i in shard : label_i = k, x_ij = 1} } per rank; readers
   sum the world partial arrays (stride bytes apart), integer sums being exact in any order.          */
int nemgpu_stats_words(const nemgpu_engine* e);
int nemgpu_shard_layout(nemgpu_engine* e, int world, int rank, int blk, int stride, int n_families_total);
int nemgpu_shard_begin(nemgpu_engine* e);
/* partial counts of this rank's families under the given labels -> stats_dev (own statistics tail) */
int nemgpu_shard_mstep_partial(nemgpu_engine* e, const uint8_t* labels_cur_dev, int32_t* stats_dev);
/* the same for the labels the last nemgpu_shard_estep_round0 produced (their class masks already exist) */
int nemgpu_shard_counts(nemgpu_engine* e, int32_t* stats_dev);
/* stats_dev = rank 0's partial statistics inside an all-gathered label array (rank r's are r*stride bytes
   further) -> parameters, or NULL to keep them; density; round 0 (+ class masks of its output, + this rank's
   flag byte); sweep_id < 0 = device counter */
int nemgpu_shard_estep_round0(nemgpu_engine* e, const int32_t* stats_dev, float beta, int sweep_id,
                              const uint8_t* labels_old_dev, uint8_t* labels_out_dev);
int nemgpu_shard_estep_round1(nemgpu_engine* e, float beta, int sweep_id, const uint8_t* labels_old_dev,
                              const uint8_t* labels_guess_dev, uint8_t* labels_out_dev);
/* round 1 and nemgpu_shard_counts(stats_dev) together: one launch where the shape has a kernel for it (the two do not
   depend on each other), else the two launches */
int nemgpu_shard_estep_round1_counts(nemgpu_engine* e, float beta, int sweep_id, const uint8_t* labels_old_dev,
                                     const uint8_t* labels_guess_dev, uint8_t* labels_out_dev, int32_t* stats_dev);
/* Fuzzy NEM (algo = nem) on several GPUs -- SURVEY.md 8e's exact alternative to an all-reduce, which cannot reproduce
   the reference's i-ordered float sums (nem_mod.c:1303-1313, 1677-1686): the E-step sharded over FAMILIES, the M-step
   over ORGANISMS.  Per rank two engines: a row engine (its families x all organisms; site range [lo, hi) of n_total)
   and a column engine (all families x its organisms).  The membership matrix lives in caller-owned device arrays
   float[n_total][K] that the caller all-gathers after every relaxation round; the statistics of the organism slices
   are all-gathered into arrays of K, K*D, K*D floats.  One step per call, every call synchronises
   (pangenomenem_amd/distributed.py: ShardedFuzzyNem).  Results equal the single engine's bit for bit. */
int nemgpu_shard_fuzzy_layout(nemgpu_engine* e, int n_true);      /* families of the whole problem (n_total may be padded) */
int nemgpu_shard_fuzzy_round(nemgpu_engine* e, float beta, int sweep_id, int round, const float* c_old_dev, const float* c_guess_dev,
                             float* c_out_dev, int* changed, int* nzero, int* firstzero);
int nemgpu_shard_fuzzy_mstep_cols(nemgpu_engine* e, const float* c_dev, float* nbobs_out_dev, float* center_out_dev, float* iner_out_dev);
int nemgpu_shard_fuzzy_finish(nemgpu_engine* e, const float* nbobs_dev, const float* center_dev, const float* iner_dev, int* emptyk);
int nemgpu_shard_fuzzy_moved(nemgpu_engine* e, const float* c_new_dev, const float* c_old_dev, int* moved);

/* nemgpu_shard_begin for a batch that starts the run over: nemgpu_reset, the cleared loop control and the density
   tables in one launch */
int nemgpu_shard_begin_restart(nemgpu_engine* e);
int nemgpu_shard_finish_iteration(nemgpu_engine* e, float beta, int is_init, const uint8_t* labels_old_dev,
                                  const uint8_t* labels_q_dev, const uint8_t* labels_r_dev);
int nemgpu_shard_round_sync(nemgpu_engine* e, float beta, int sweep_id, const uint8_t* labels_old_dev,
                            const uint8_t* labels_guess_dev, uint8_t* labels_out_dev, int* changed);
/* NEMGPU_TIE_LIBC in the sharded path (nem_alg.c:617-637 -> nem_rnd.c:53-61): a tied site draws random() number
   `draws before the sweep + sites below it that drew in this sweep`, and the sites of the ranks below come first.  Every
   rank's draws of the round that produced a label array ride in its block's tail (an int32, nemgpu_shard_layout's
   blk rounded up to 4, + 4), all-gathered with the labels; every rank holds the same window of the same stream.
   nemgpu_shard_set_labels: the driver's three label arrays (the engine keeps each one's per-block draw counts).
   A start under this rule is completed from the host, sweep by sweep (the blind sweep's ties come first in the
   stream): nemgpu_shard_round_sync rounds until no rank changes anything, nemgpu_shard_round_draws after each (this
   rank's draws; table_short: a draw fell outside the table -- the round is void everywhere: nemgpu_shard_grow_draws on
   every rank, then the round again), nemgpu_shard_book_draws(sum over the ranks of the final round's draws). */
int nemgpu_shard_set_labels(nemgpu_engine* e, const uint8_t* lab0, const uint8_t* lab1, const uint8_t* lab2);
int nemgpu_shard_round_draws(nemgpu_engine* e, int* draws, int* table_short);
int nemgpu_shard_grow_draws(nemgpu_engine* e);
int nemgpu_shard_book_draws(nemgpu_engine* e, int n);
int nemgpu_shard_end_enqueue(nemgpu_engine* e);   /* async copy of the control block; last call of a batch */
int nemgpu_shard_end(nemgpu_engine* e, nemgpu_result* res, int* commits, int* need_rounds);   /* sync + report */
int nemgpu_shard_set_sweep_number(nemgpu_engine* e, int next_sweep);

/* RCCL called directly: the sharded EM's one collective (the in-place all-gather of label blocks) issued from C, so
   that a whole batch -- kernels and all-gathers -- is enqueued by ONE call instead of a Python round trip per
   launch.  librccl.so is bound at run time (pass the path of the library PyTorch ships; NULL: the loader's default);
   torch.distributed remains the bootstrap (it carries the 128-byte ncclUniqueId from rank 0) and the fallback.
   nemgpu_rccl_attach is collective over the job's ranks.  nemgpu_shard_enqueue_batch = nemgpu_shard_begin,
   [the two initial sweeps], n_iters iterations (round 0, all-gather, round 1 + counts, all-gather, convergence +
   loop control), nemgpu_shard_end_enqueue; follow with nemgpu_shard_end.  lab0..2: the three all-gathered label
   arrays; stats_off: byte offset of a rank's statistics inside its block; base: buffer of the current partition. */
int nemgpu_rccl_open(const char* librccl_path);
int nemgpu_rccl_unique_id(uint8_t id128[128]);
int nemgpu_rccl_attach(nemgpu_engine* e, const uint8_t id128[128], int world, int rank);
int nemgpu_shard_enqueue_batch(nemgpu_engine* e, int with_init, int n_iters, int base, float beta, int want_stats,
                               uint8_t* lab0, uint8_t* lab1, uint8_t* lab2, int stats_off);
/* One in-place all-gather of `stride`-byte blocks of buf_dev (world * stride bytes, device memory) through the
   engine's own communicator, waited for with a deadline; collective.  The caller compares the result with the same
   gather through torch.distributed before trusting the native path.  A timeout aborts and detaches the communicator. */
int nemgpu_rccl_selftest(nemgpu_engine* e, uint8_t* buf_dev, int stride, int timeout_ms);
/* ranks of the engine's native communicator as RCCL reports them (ncclCommCount); 0 when none is attached */
int nemgpu_rccl_ranks(const nemgpu_engine* e);

/* Test hook: load a partition (row-major [n_total x k], HOST) as the current state
   (argmax labels for ncem engines). */
int nemgpu_set_partition(nemgpu_engine* e, const float* c_nk);

/* Results (HOST buffers; any pointer may be NULL).  c_nk is row-major [(site_hi-site_lo) x k].
   nemgpu_get_results: parameters and partition in one device round trip. */
int nemgpu_get_results(nemgpu_engine* e, float* prop, float* center, float* disp, float* nbobs_k, float* c_nk);
int nemgpu_get_partition(nemgpu_engine* e, float* c_nk);
int nemgpu_get_labels(nemgpu_engine* e, uint8_t* labels);
int nemgpu_get_params(nemgpu_engine* e, float* prop, float* center, float* disp, float* nbobs_k);
int nemgpu_get_density(nemgpu_engine* e, double* pkfki_nk, float* logpkfki_nk);
/* ------------------------------------------------------------------------------------------
 * 3. File layer (host only; what nem() uses on either side of the engine).  Readers follow
 *    ReadStrFile / ReadMatrixFile / ReadPtsNeighs / ReadParamFile (nem_exe.c:739-1091, 1278-1478),
 *    writers follow SaveResults (nem_exe.c:1596-1781).  Return values are StatusET codes.
 * ------------------------------------------------------------------------------------------ */
typedef struct nemio_inputs nemio_inputs;
/* Parse <Fname>.str/.dat/.nei/.m for nk classes. */
int nemio_read(const char* Fname, int nk, nemio_inputs** out);
void nemio_free(nemio_inputs* in);
/* sizes: n families, d organisms, nnz neighbour entries, max_neighs, param_mode (1 init / 2 fixed), type 'S'|'N' */
int nemio_sizes(const nemio_inputs* in, int* n, int* d, int* nnz, int* max_neighs, int* param_mode, int* type);
/* copy out: xbits [n * ceil(d/32)], nei_ptr [n+1], nei_idx/nei_w [nnz], prop [k], center/disp [k*d]; NULL skips */
int nemio_copy(const nemio_inputs* in, uint32_t* xbits, int32_t* nei_ptr, int32_t* nei_idx, float* nei_w,
               float* prop, float* center, float* disp);
int nemio_write_uf(const char* path, const float* c_nk, int n, int k);
/* Test hook: " %<width>.<dec>f" (dec <= 3) of a float exactly as printf prints it -- the formatter behind the
   per-iteration <Fname>.log lines (WriteLogClasses' " %5.3f", " %7.3f", " %7.1f").  out: at least 64 bytes;
   returns the length written, -1 on bad arguments. */
int nemio_format_fixed(float v, int width, int dec, char* out);
int nemio_write_cf(const char* path, const float* c_nk, int n, int k, int tie_rule, uint32_t seed);
int nemio_write_mf(const char* path, const float crit6[6], float beta, int d, int k, const float* center,
                   const float* prop, const float* disp);

/* Kernel-duration probe for bench.py: `reps` launches of the E1 density kernel on the current
   parameters, each bracketed by HIP events on the engine's stream; average duration (ms) and the
   algorithmic bytes one launch moves. */
int nemgpu_profile_density(nemgpu_engine* e, int reps, double* avg_ms, double* algorithmic_bytes_per_launch,
                           int* used_fused_kernel);
/* The same for the three kernels of a solo NCEM iteration -- {E1, one relaxation round of the E-step sweep, the M-step
   counts} -- each launched `reps` times back to back between ONE pair of HIP events (an event pair per launch costs as
   much as the shorter kernels).  avg_ms[3], bytes[3]; which[0] = 1 when E1 is the fused kernel.  The engine's partition
   is not advanced. */
int nemgpu_profile_kernels(nemgpu_engine* e, int reps, double avg_ms[3], double bytes[3], int which[1]);
/* E1 (DensBernoulli over the whole matrix, nem_mod.c:619-690) as a lock-step batch launches it: the density kernels of
   `count` same-shaped engines of one device in ONE launch, `reps` launches back to back between one pair of HIP events.
   avg_ms: one launch (all members); algorithmic_bytes_per_launch: all members' bit-packed matrices, tables and outputs. */
int nemgpu_profile_density_many(nemgpu_engine** engines, int count, int reps, double* avg_ms, double* algorithmic_bytes_per_launch);
/* `reps` in-place all-gathers of the sharded EM's label blocks through the engine's own communicator between one pair
   of HIP events: the cost of ONE of an iteration's two collectives.  Collective over the job's ranks. */
int nemgpu_rccl_time_allgather(nemgpu_engine* e, uint8_t* buf_dev, int reps, double* avg_ms);
/* FETCH_SIZE calibration helper: `reps` launches reading `bytes` of device memory 16 bytes per lane (E1's
   pattern); run under `rocprofv3 --pmc FETCH_SIZE` and compare with the known byte count (profiles/README.md). */
int nemgpu_calibrate_fetch(size_t bytes, int reps);

/* E1 fast-forward (pangenomenem_amd/csrc/nem_ff.hpp): inside one float binade the reference's chain
   dk = (float)(((double)dk + |x-mu|*L1) - L0)  (nem_mod.c:661) adds a constant to the accumulator's bit pattern, so
   whole 32-organism words advance with one popcount; bit-identical to stepping.  on: 1 always, 0 never,
   -1 (default) automatic -- on from 256 organisms.  The
   environment variable NEM_MI355X_FF=0|1 sets the mode at creation.  The tests compare the two code paths. */
int nemgpu_set_fast_forward(nemgpu_engine* e, int on);
/* The increment tables that fast-forward uses for class constants L1 = log((1-eps)/eps), L0 = log(1-eps):
   q0[E] / q1[E] = bit-pattern increment of a match / mismatch step while the accumulator's exponent field is E
   (2^23 = "step exactly").  Host-only, needs no GPU. */
int nemgpu_ff_table(double l1, double l0, uint32_t* q0_256, uint32_t* q1_256);

/* The criteria of ComputeCrit (nem_alg.c:2727-2745) are i-ordered float accumulators
   acc = (float)((double)acc + x_i); the library evaluates them exactly but in parallel: between two powers of two
   the accumulator moves on a fixed grid and the chain is a prefix sum of integers
   (pangenomenem_amd/csrc/nem_chain.hpp).  Test hooks for that procedure on arbitrary doubles:
   nemgpu_chain_host -- mode 0: the plain sequential loop, mode 1: the host emulation of the device procedure
   (no GPU needed); nemgpu_chain_device -- the device procedure itself. */
float nemgpu_chain_host(const double* x, long long n, float init, int mode);
/* s = 0; `times` times s += x in float: mode 0 the loop, mode 1 its closed form per binade, mode 2 the integer form of
   that for integer x below 2^24 (csrc/nem_ff.hpp) */
float nemgpu_repeat_add_host(float x, long long times, int mode);
int nemgpu_chain_device(const double* x, long long n, float init, int device, float* out);
/* InerToDispK_'s d-ordered float sum of a class's inertia values (nem_mod.c:1054-1058): under NCEM they are
   non-negative multiples of 1/2 below 2^24, and the library evaluates the chain in pieces, all threads of a block at
   once (pangenomenem_amd/csrc/nem_halfsum.hpp).  Test hooks on arbitrary such values: nemgpu_halfsum_host -- mode 0:
   the plain loop, mode w = 1..16: the host emulation of the device procedure with w wavefronts (stepped: adds it took
   for real; no GPU needed); nemgpu_halfsum_device -- the device procedure itself (n <= 8192, waves 1 or 16);
   nemgpu_halfsum_profile -- its in-kernel timing: us[0..3] = prefix sums, classification + walk, scan + hand-over,
   ordered pass; us[4] = the same chain as plain dependent adds on one lane. */
float nemgpu_halfsum_host(const float* x, int n, int mode, int* stepped);
int nemgpu_halfsum_device(const float* x, int n, int waves, int device, float* out);
int nemgpu_halfsum_profile(const float* x, int n, int waves, int device, double us[5]);

/* nemgpu_destroy parks an engine's stream, first 16 MB of device memory and pinned control block (up to 16 sets
   per process) for the next nemgpu_create on the same device -- a nem() call creates and destroys an engine, and
   creating these costs as much as a small EM run.  This frees whatever is parked. */
void nemgpu_release_cached(void);

/* hipGraph policy of the pipelined EM loop.  By default a batch shape (initial sweeps or not, buffer phase, number of
   iterations) goes out as plain launches the first time and is captured + instantiated the second time it is
   enqueued; capture_on_first != 0 captures at once (benchmarks prime every shape of their timed region this way). */
int nemgpu_set_graph_policy(nemgpu_engine* e, int capture_on_first);
/* out[0] batches sent as plain launches, out[1] batches captured + instantiated, out[2] graph replays,
   out[3] sweeps the host had to finish round by round */
int nemgpu_graph_counters(const nemgpu_engine* e, int out[4]);
/* The fused form of the E2 sweep (the reference's ComputePartitionNEM, nem_alg.c:2330-2405, as relaxation rounds): one
   engine alone can run the first rounds of a sweep in ONE launch whose blocks meet between rounds (NCEM, hash / first tie
   rule, at most one block per CU; NEM_MI355X_FUSED_SWEEP=1 turns it on -- measured slower than one launch per round
   except at 200 000 x 5 000, DESIGN.md -- NEM_MI355X_FUSED_ROUNDS sets the rounds per launch).  out[0] such launches issued or captured so far, out[1] launches whose blocks failed to meet (the sweep was
   redone with one launch per round and the engine keeps to that form), out[2] 1 while the form is in use, out[3] 0. */
int nemgpu_sweep_counters(const nemgpu_engine* e, int out[4]);
/* how the last nemgpu_run_random went (RandNemAlgo, nem_alg.c:1574-1742, starts in lock step): out = {lock-step rounds,
   starts that stood in them, starts run alone on the engine's own path, starts thrown away and redone because a start
   before them drew tie-breaks behind its initial sweeps (TIE_LIBC: one random() stream for all starts)} */
int nemgpu_random_start_counters(const nemgpu_engine* e, int out[4]);
/* development probe (NEM_MI355X_SWEEP_PROF=1): the device's 100 MHz clock at the phase boundaries of the last fused
   launch, first block in out[0..31], last block in out[32..63] (0: phase not reached) */
int nemgpu_sweep_phases(unsigned long long out64[64]);

/* Re-target the engine to another HIP stream (e.g. the capturing stream of a torch.cuda.graph). */
int nemgpu_set_stream(nemgpu_engine* e, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* NEM_MI355X_H */
