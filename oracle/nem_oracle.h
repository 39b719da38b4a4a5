/*
 * nem_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference NEM hot path
 * (labgem/pangenomeNEM, ppanggolin/NEM/nem_alg.c + nem_mod.c) for Bernoulli
 * mixtures on 0/1 data.  It exists to CHECK the HIP engine; it is never
 * imported, linked or executed by the product path (pangenomenem_amd/).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED.  Every stage and the full loop are checked
 * bit-for-bit against the compiled, unmodified reference (oracle/_ref, built by
 * oracle/Makefile from /root/reference) in tests/test_oracle_vs_reference.py
 * (runs where /root/reference exists) and against the committed golden vectors
 * in tests/golden/ (generated from the reference by tests/golden/make_golden.py).
 */
#ifndef NEM_ORACLE_H
#define NEM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* enums keep the reference's numeric values (nem_typ.h:121-128, 191-205, 271-277) */
enum { ORC_ALGO_NEM = 0, ORC_ALGO_NCEM = 1 };
enum { ORC_DISP___ = 0, ORC_DISP_K_ = 1, ORC_DISP__D = 2, ORC_DISP_KD = 3 };
enum { ORC_PROP__ = 0, ORC_PROP_K = 1 };
/* CvemET (nem_typ.h) + one: CVTEST_CRIT compares the chosen criterion (M, DEFAULT_CRIT) of consecutive iterations
   (HasConverged, nem_alg.c:2090-2105); the value it starts from is 0 in a run without a log (Criteria = {0},
   nem_exe.c:264) and the criterion of the initial partition in a logging run (WriteLogCrit, nem_alg.c:1980, 2398):
   CRIT = the former, CRIT_LOGGED = the latter */
enum { ORC_CV_NONE = 0, ORC_CV_CLAS = 1, ORC_CV_CRIT = 2, ORC_CV_CRIT_LOGGED = 3 };
/* tie rule of the NCEM C-step (ComputeMAP, nem_alg.c:590-645):
   LIBC  = reference behaviour: kmaxes[random() % (nequal+1)] in site order (srandom(seed) first)
   FIRST = TIE_FIRST (keep the first maximum)
   HASH  = the HIP engine's reproducible stand-in for the time-seeded random():
           kmaxes[mix32(seed, sweep, site) % (nequal+1)]                                  */
enum { ORC_TIE_LIBC = 0, ORC_TIE_FIRST = 1, ORC_TIE_HASH = 2 };
/* status = StatusET (nem_typ.h:106-117) */
enum { ORC_STS_OK = 0, ORC_STS_W_EMPTYCLASS = 2 };

typedef struct {
    int n, d, k;                 /* families, organisms, classes */
    const unsigned char* x;      /* n*d bytes, 0/1, row-major (PointsM) */
    const int* nei_ptr;          /* n+1 */
    const int* nei_idx;          /* 0-based, .nei file order */
    const float* nei_w;
    int algo, disper, propor, cvtest;
    float beta, cvthres;
    int it_max;
    int param_fix;               /* .m flag 2: parameters never re-estimated (nem_alg.c:1806) */
    int tie_rule;
    unsigned tie_seed;
} orc_problem;

typedef struct {
    float* c_nk;                 /* n*k  posteriors (ClassifM) */
    float* prop_k;               /* in: initial, out: final */
    float* center_kd;
    float* disp_kd;
    float* nbobs_k;              /* k */
    float* nbobs_kd;             /* k*d */
    float* iner_kd;              /* k*d */
    double* pkfki_nk;            /* n*k (may be NULL) */
    float* logpkfki_nk;          /* n*k (may be NULL) */
    float crit[6];               /* D G U M L Z */
    int iters;                   /* completed EM iterations */
    int converged;
    int emptyk;                  /* 1..K or 0 */
    int n_zero_density;          /* sites that took the cumnum==0 branch at least once */
} orc_state;

unsigned orc_mix32(unsigned seed, unsigned sweep, unsigned site);

/* E1: ComputePkFkiM + DensBernoulli (nem_alg.c:2234-2289, nem_mod.c:619-690) */
int orc_density(int n, int d, int k, const unsigned char* x,
                const float* prop_k, const float* center_kd, const float* disp_kd,
                double* pkfki_nk, float* logpkfki_nk);

/* E2: one ComputePartitionNEM sweep, UPDATE_SEQ, ORDER_DIRECT (nem_alg.c:2330-2405,
   2546-2616, 2850-2884) + optional NCEM C-step (nem_alg.c:590-664).  c_nk in/out. */
int orc_sweep(int n, int k, const int* nei_ptr, const int* nei_idx, const float* nei_w,
              float beta, const double* pkfki_nk, int ncem,
              int tie_rule, unsigned tie_seed, unsigned sweep_id, float* c_nk);

/* One RELAXATION ROUND of the same sweep for sites [lo, hi): site i reads c_guess for neighbours j < i
   and c_old for neighbours j >= i (rows are GLOBAL), writes c_out[i].  The unique fixed point of these
   rounds (c_out == c_guess) is exactly orc_sweep()'s result; this is how the HIP engine parallelises
   the Gauss-Seidel order (tests check the equivalence on CPU).  pkfki_local has (hi-lo) rows.
   Returns the number of sites of [lo, hi) whose output differs (bitwise) from its guess. */
int orc_relax_round(int lo, int hi, int k, const int* nei_ptr_local, const int* nei_idx, const float* nei_w,
                    float beta, const double* pkfki_local, int ncem, int tie_rule, unsigned tie_seed,
                    unsigned sweep_id, const float* c_old, const float* c_guess, float* c_out);
int orc_relax_round_keyed(int lo, int hi, int k, const int* nei_ptr_local, const int* nei_idx, const float* nei_w,
                    float beta, const double* pkfki_local, int ncem, int tie_rule, unsigned tie_seed,
                    unsigned sweep_id, const float* c_old, const float* c_guess, float* c_out, int key_bias);

/* M: EstimPara for FAMILY_BERNOULLI (nem_mod.c:415-469, 1180-1479, 1646-1704, 922-1174) */
int orc_mstep(int n, int d, int k, const unsigned char* x, const float* c_nk,
              int disper, int propor,
              float* prop_k, float* center_kd, float* disp_kd,
              float* nbobs_k, float* nbobs_kd, float* iner_kd, int* emptyk);

/* C1: ComputeCrit (nem_alg.c:2678-2757) */
void orc_crit(int n, int k, const int* nei_ptr, const int* nei_idx, const float* nei_w,
              float beta, const float* c_nk, const double* pkfki_nk,
              const float* logpkfki_nk, float crit6[6]);

/* HasConverged, CVTEST_CLAS (nem_alg.c:2075-2089) */
int orc_converged(int n, int k, const float* c_nk, const float* cold_nk, float thres);

/* ClassifyByNemOneBeta/INIT_PARAM_FILE + NemAlgo (nem_alg.c:1151-1169, 1746-1879) */
int orc_run(const orc_problem* p, orc_state* s);

/* INIT_RANDOM: ClassifyByNemOneBeta's default branch -> RandNemAlgo (nem_alg.c:1185-1189, 1574-1742): n_starts
   random starts (centres = distinct data rows drawn with libc random() after srandom(seed)), best start by
   criterion M, final EstimPara on the best partition.  best_start: 0-based index or -1. */
int orc_run_random(const orc_problem* p, orc_state* s, int n_starts, unsigned seed, int* best_start);

/* seconds of wall time spent in the EM iteration loop of the last orc_run (bench only) */
double orc_last_loop_seconds(void);

#ifdef __cplusplus
}
#endif
#endif
