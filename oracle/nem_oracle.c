/*
 * nem_oracle.c -- TEST INFRASTRUCTURE ONLY (see nem_oracle.h).
 *
 * CPU restatement of the reference NEM E-step / M-step loop for Bernoulli
 * mixtures on 0/1 data.  Every function cites the reference lines it follows
 * (paths relative to /root/reference/ppanggolin/NEM/).  The arithmetic keeps the
 * reference's exact operand types and evaluation order (float storage, double
 * temporaries where C's usual arithmetic conversions produce them); loop nests
 * are re-ordered only where each accumulation chain keeps its own order.
 *
 * Build: make -C oracle oracle   (gcc -O2 -ffp-contract=off, no fast-math)
 */
#include "nem_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_EPSILON 1e-20 /* nem_typ.h:63 (a double constant) */

static double g_loop_seconds = 0.0;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_last_loop_seconds(void) { return g_loop_seconds; }
/* tie draws made by the sweeps since the library was loaded (a diagnostic: ORC_DRAW_LOG=1 prints it per sweep) */
static long g_draws = 0;
long orc_tie_draws(void) { return g_draws; }
static void draw_log(const char* what, int iter)
{
    static int on = -1;
    if (on < 0) on = getenv("ORC_DRAW_LOG") != NULL;
    if (on) fprintf(stderr, "orc draws %s %d %ld\n", what, iter, g_draws);
}

/* The HIP engine's counter-based stand-in for the reference's time-seeded
   random() (nem_exe.c:353,621; nem_rnd.c:40-63).  Must stay identical to
   mix32() in pangenomenem_amd/csrc/nem_kernels.hip. */
unsigned orc_mix32(unsigned seed, unsigned sweep, unsigned site)
{
    unsigned h = seed * 0x9E3779B1u + sweep * 0x85EBCA77u + site * 0xC2B2AE3Du + 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x7FEB352Du;
    h ^= h >> 15; h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

/* ------------------------------------------------------------------ E1 */
/* ComputePkFkiM (nem_alg.c:2260-2285) calling DensBernoulli (nem_mod.c:649-688).
   For 0/1 data there is no NaN, so every variable is "observed". */
int orc_density(int n, int d, int k, const unsigned char* x,
                const float* prop_k, const float* center_kd, const float* disp_kd,
                double* pkfki_nk, float* logpkfki_nk)
{
    int sts = ORC_STS_OK;
    double* l1 = (double*)malloc(sizeof(double) * (size_t)d);
    double* l0 = (double*)malloc(sizeof(double) * (size_t)d);
    unsigned char* pos = (unsigned char*)malloc((size_t)d);
    int* ad0 = (int*)malloc(sizeof(int) * (size_t)d);
    int* ad1 = (int*)malloc(sizeof(int) * (size_t)d);
    int ik, i, j;

    for (ik = 0; ik < k; ik++) {
        double pk = prop_k[ik];                         /* nem_alg.c:2262 */
        float logpk;
        if (pk > ORC_EPSILON) logpk = (float)log(pk);   /* :2265-2266 */
        else { logpk = -INFINITY; sts = ORC_STS_W_EMPTYCLASS; } /* :2269-2270 */

        for (j = 0; j < d; j++) {
            float disp = disp_kd[ik * d + j];           /* nem_mod.c:656 */
            float cen = center_kd[ik * d + j];
            pos[j] = ((double)disp > ORC_EPSILON);      /* :660 */
            /* the two logs of :661 depend on (k,d) only */
            l1[j] = log((double)((1 - disp) / disp));
            l0[j] = log((double)(1 - disp));
            ad0[j] = abs((int)(0.0f - cen));            /* :657-658, x = 0 */
            ad1[j] = abs((int)(1.0f - cen));            /*           x = 1 */
        }
        for (i = 0; i < n; i++) {
            const unsigned char* xi = x + (size_t)i * d;
            float dk = 0.0f;
            int nuldens = 0;
            double fki; float logfki;
            for (j = 0; j < d; j++) {
                int absdif = xi[j] ? ad1[j] : ad0[j];
                if (pos[j])
                    dk = (float)(((double)dk + absdif * l1[j]) - l0[j]);  /* :661 */
                else if (absdif != 0)
                    nuldens = 1;                                          /* :664-666 */
            }
            if (!nuldens) { logfki = -dk; fki = exp((double)logfki); }    /* :679-680 */
            else { logfki = -FLT_MAX; fki = 0.0; }                        /* :685-686 */
            pkfki_nk[(size_t)i * k + ik] = pk * fki;                      /* nem_alg.c:2282 */
            if (logpkfki_nk) logpkfki_nk[(size_t)i * k + ik] = logpk + logfki; /* :2283 */
        }
    }
    free(l1); free(l0); free(pos); free(ad0); free(ad1);
    return sts;
}

/* ------------------------------------------------------------------ E2 */
/* ComputeMAP (nem_alg.c:590-645) on one row */
static int orc_map(const float* row, int k, int tie_rule, unsigned seed, unsigned sweep, unsigned site, int* kmaxes)
{
    int kk, kmax = 0, nequal = 0;
    float ukmax = row[0];
    for (kk = 1; kk < k; kk++) if (row[kk] > ukmax) { ukmax = row[kk]; kmax = kk; }  /* :607-615 */
    if (tie_rule == ORC_TIE_FIRST) return kmax;                                       /* :641 */
    kmaxes[0] = kmax;
    for (kk = kmax + 1; kk < k; kk++) if (row[kk] == ukmax) kmaxes[++nequal] = kk;    /* :620-628 */
    if (nequal > 0) {
        if (tie_rule == ORC_TIE_LIBC) { g_draws++; return kmaxes[(int)(random() % (nequal + 1))]; } /* nem_rnd.c:53-61 */
        return kmaxes[orc_mix32(seed, sweep, site) % (unsigned)(nequal + 1)];
    }
    return kmax;
}

int orc_sweep(int n, int k, const int* nei_ptr, const int* nei_idx, const float* nei_w,
              float beta, const double* pkfki_nk, int ncem,
              int tie_rule, unsigned tie_seed, unsigned sweep_id, float* c_nk)
{
    double* cinum = (double*)malloc(sizeof(double) * (size_t)k);
    int* kmaxes = (int*)malloc(sizeof(int) * (size_t)k);
    int ipt, kk, nzero = 0;

    for (ipt = 0; ipt < n; ipt++) {                       /* nem_alg.c:2370, ORDER_DIRECT */
        int b = nei_ptr ? nei_ptr[ipt] : 0, e = nei_ptr ? nei_ptr[ipt + 1] : 0, t;
        double cumnum = 0.0;
        float* cout = c_nk + (size_t)ipt * k;
        for (kk = 0; kk < k; kk++) {                      /* :2576-2586 */
            float context = 0.0f;                         /* SumNeighsOfClass :2865-2875 */
            for (t = b; t < e; t++)
                context = context + (nei_w[t] * c_nk[(size_t)nei_idx[t] * k + kk]); /* Cin = CM (UPDATE_SEQ, :2380) */
            cinum[kk] = pkfki_nk[(size_t)ipt * k + kk] * exp((double)beta * context); /* :2581-2582 */
            cumnum = cumnum + cinum[kk];
        }
        if (cumnum > 0) {                                 /* :2589 */
            if (cumnum > ORC_EPSILON) {
                double invz = 1 / cumnum;
                for (kk = 0; kk < k; kk++) cout[kk] = (float)(invz * cinum[kk]);      /* :2594 */
            } else {
                double invz = 1 / (cumnum / ORC_EPSILON);
                for (kk = 0; kk < k; kk++) cout[kk] = (float)(invz * (cinum[kk] / ORC_EPSILON)); /* :2600 */
            }
        } else {
            double invz = 1.0 / k;                        /* :2604-2607 */
            for (kk = 0; kk < k; kk++) cout[kk] = (float)invz;
            nzero++;
        }
        if (ncem) {                                       /* :2386-2391 */
            int kmap = orc_map(cout, k, tie_rule, tie_seed, sweep_id, (unsigned)ipt, kmaxes);
            for (kk = 0; kk < k; kk++) cout[kk] = 0.0f;   /* LabelToClassVector :659-663 */
            cout[kmap] = 1.0f;
        }
    }
    free(cinum); free(kmaxes);
    return nzero;
}

/* One relaxation round (see nem_oracle.h).  Same arithmetic as orc_sweep, different data flow. */
int orc_relax_round(int lo, int hi, int k, const int* nei_ptr_local, const int* nei_idx, const float* nei_w,
                    float beta, const double* pkfki_local, int ncem, int tie_rule, unsigned tie_seed,
                    unsigned sweep_id, const float* c_old, const float* c_guess, float* c_out)
{
    return orc_relax_round_keyed(lo, hi, k, nei_ptr_local, nei_idx, nei_w, beta, pkfki_local, ncem, tie_rule, tie_seed, sweep_id,
                                 c_old, c_guess, c_out, 0);
}

/* ... with the tie hash keyed by (site - key_bias): a rank of the sharded driver works in label SLOTS (every rank's block
   is followed by a flag tail), the hash rule is keyed by the family's true index -- key_bias = the slots skipped below */
int orc_relax_round_keyed(int lo, int hi, int k, const int* nei_ptr_local, const int* nei_idx, const float* nei_w,
                          float beta, const double* pkfki_local, int ncem, int tie_rule, unsigned tie_seed,
                          unsigned sweep_id, const float* c_old, const float* c_guess, float* c_out, int key_bias)
{
    double* cinum = (double*)malloc(sizeof(double) * (size_t)k);
    float* row = (float*)malloc(sizeof(float) * (size_t)k);
    int* kmaxes = (int*)malloc(sizeof(int) * (size_t)k);
    int gi, kk, changed = 0;
    for (gi = lo; gi < hi; gi++) {
        int il = gi - lo;
        int b = nei_ptr_local ? nei_ptr_local[il] : 0, e = nei_ptr_local ? nei_ptr_local[il + 1] : 0, t;
        double cumnum = 0.0;
        for (kk = 0; kk < k; kk++) {
            float context = 0.0f;
            for (t = b; t < e; t++) {
                int j = nei_idx[t];
                const float* src = (j < gi) ? c_guess : c_old;
                context = context + (nei_w[t] * src[(size_t)j * k + kk]);
            }
            cinum[kk] = pkfki_local[(size_t)il * k + kk] * exp((double)beta * context);
            cumnum = cumnum + cinum[kk];
        }
        if (cumnum > 0) {
            if (cumnum > ORC_EPSILON) { double invz = 1 / cumnum; for (kk = 0; kk < k; kk++) row[kk] = (float)(invz * cinum[kk]); }
            else { double invz = 1 / (cumnum / ORC_EPSILON); for (kk = 0; kk < k; kk++) row[kk] = (float)(invz * (cinum[kk] / ORC_EPSILON)); }
        } else {
            double invz = 1.0 / k;
            for (kk = 0; kk < k; kk++) row[kk] = (float)invz;
        }
        if (ncem) {
            int kmap = orc_map(row, k, tie_rule, tie_seed, sweep_id, (unsigned)(gi - key_bias), kmaxes);
            for (kk = 0; kk < k; kk++) row[kk] = 0.0f;
            row[kmap] = 1.0f;
        }
        if (memcmp(row, c_guess + (size_t)gi * k, sizeof(float) * (size_t)k) != 0) changed++;
        memcpy(c_out + (size_t)gi * k, row, sizeof(float) * (size_t)k);
    }
    free(cinum); free(row); free(kmaxes);
    return changed;
}

/* ------------------------------------------------------------------- M */
int orc_mstep(int n, int d, int k, const unsigned char* x, const float* c_nk,
              int disper, int propor,
              float* prop_k, float* center_kd, float* disp_kd,
              float* nbobs_k, float* nbobs_kd, float* iner_kd, int* emptyk)
{
    int sts = ORC_STS_OK;
    int h, i, j;
    float* cum = (float*)malloc(sizeof(float) * (size_t)d);
    float* xmed = (float*)malloc(sizeof(float) * (size_t)d);
    float* med = (float*)malloc(sizeof(float) * (size_t)d);
    unsigned char* phase = (unsigned char*)malloc((size_t)d);

    *emptyk = 0;
    for (h = 0; h < k; h++) {
        /* EstimSizes (nem_mod.c:1293-1315): for every j the same i-ordered float sum;
           without NaN N_KD[h,j] is bitwise N_K[h] */
        float nk = 0.0f;
        for (i = 0; i < n; i++) nk += c_nk[(size_t)i * k + h];
        nbobs_k[h] = nk;
        for (j = 0; j < d; j++) nbobs_kd[h * d + j] = nk;

        /* EstimLaplaceCenters (:1358-1412) + ComputeMedian (:1439-1477).
           Sort_ND[:,j] for 0/1 data = zeros in index order, then ones in index order
           (ModelPreprocess nem_alg.c:701-716 with glibc's stable qsort). */
        if ((double)nk > ORC_EPSILON) {                    /* :1363 */
            float halfwei = nk / 2;                        /* :1439 */
            int pass;
            for (j = 0; j < d; j++) { cum[j] = 0.0f; phase[j] = 0; xmed[j] = 0.0f; med[j] = 0.0f; }
            for (pass = 0; pass < 2; pass++) {
                float xv = (float)pass;
                for (i = 0; i < n; i++) {
                    const unsigned char* xi = x + (size_t)i * d;
                    float c = c_nk[(size_t)i * k + h];
                    for (j = 0; j < d; j++) {
                        if (xi[j] != pass || phase[j] == 2) continue;
                        if (phase[j] == 0) {
                            cum[j] += c;                                   /* :1456 */
                            if (!(cum[j] < halfwei)) {                     /* loop test :1451 */
                                xmed[j] = xv;
                                if (cum[j] > halfwei + ORC_EPSILON) { med[j] = xv; phase[j] = 2; } /* :1464-1466 */
                                else phase[j] = 1;
                            }
                        } else {                                           /* :1470-1476 */
                            if (!((double)c < ORC_EPSILON)) {
                                med[j] = (float)(0.5 * (xmed[j] + xv));
                                phase[j] = 2;
                            }
                        }
                    }
                }
            }
            for (j = 0; j < d; j++) {
                if (phase[j] != 2) med[j] = xmed[j];       /* reference reads out of bounds here (UB) */
                center_kd[h * d + j] = med[j];             /* :1375-1376 (N_KD == N_K) */
            }
        } else {                                           /* :1404-1408 */
            sts = ORC_STS_W_EMPTYCLASS;
            *emptyk = h + 1;
        }

        /* EstimLaplaceIner (:1669-1686), also run for an empty class (old centre) */
        for (j = 0; j < d; j++) iner_kd[h * d + j] = 0.0f;
        for (i = 0; i < n; i++) {
            const unsigned char* xi = x + (size_t)i * d;
            float c = c_nk[(size_t)i * k + h];
            for (j = 0; j < d; j++) {
                float xij = (float)xi[j];
                iner_kd[h * d + j] = (float)((double)iner_kd[h * d + j] +
                                             c * fabs((double)(xij - center_kd[h * d + j]))); /* :1683 */
            }
        }
    }

    /* InerToDisp with MissMode == MISSING_IGNORE (forced for Bernoulli, nem_mod.c:446-448) */
    switch (disper) {
    case ORC_DISP___: {                                    /* :988-1015 */
        float vol = 0.0f, nobs = 0.0f;
        for (h = 0; h < k; h++) if (nbobs_k[h] > 0)
            for (j = 0; j < d; j++) { vol += iner_kd[h * d + j]; nobs += nbobs_kd[h * d + j]; }
        vol /= nobs;
        for (h = 0; h < k; h++) for (j = 0; j < d; j++) disp_kd[h * d + j] = vol;
        break; }
    case ORC_DISP_K_:                                      /* :1043-1073 */
        for (h = 0; h < k; h++) if (nbobs_k[h] > 0) {
            float sn = 0.0f, si = 0.0f, dispk;
            for (j = 0; j < d; j++) { sn += nbobs_kd[h * d + j]; si += iner_kd[h * d + j]; }
            dispk = si / sn;
            for (j = 0; j < d; j++) disp_kd[h * d + j] = dispk;
        }
        break;
    case ORC_DISP__D:                                      /* :1104-1126 */
        for (j = 0; j < d; j++) {
            float sn = 0.0f, si = 0.0f, dispd;
            for (h = 0; h < k; h++) { sn += nbobs_kd[h * d + j]; si += iner_kd[h * d + j]; }
            dispd = si / sn;
            for (h = 0; h < k; h++) disp_kd[h * d + j] = dispd;
        }
        break;
    default:                                               /* DISPER_KD :1152-1170 */
        for (h = 0; h < k; h++) for (j = 0; j < d; j++)
            if ((double)nbobs_kd[h * d + j] > ORC_EPSILON)
                disp_kd[h * d + j] = iner_kd[h * d + j] / nbobs_kd[h * d + j];
    }

    /* proportions (nem_mod.c:456-465) */
    if (propor == ORC_PROP_K) for (h = 0; h < k; h++) prop_k[h] = nbobs_k[h] / n;
    else for (h = 0; h < k; h++) prop_k[h] = (float)(1.0 / k);

    free(cum); free(xmed); free(med); free(phase);
    return sts;
}

/* ------------------------------------------------------------------ C1 */
void orc_crit(int n, int k, const int* nei_ptr, const int* nei_idx, const float* nei_w,
              float beta, const float* c_nk, const double* pkfki_nk,
              const float* logpkfki_nk, float crit6[6])
{
    float cd = 0.0f, cg = 0.0f, cl = 0.0f, cz = 0.0f, cu, cm;  /* nem_alg.c:2702-2707 */
    int i, kk, t;
    for (i = 0; i < n; i++) {
        int b = nei_ptr ? nei_ptr[i] : 0, e = nei_ptr ? nei_ptr[i + 1] : 0;
        double fi = 0.0; float zi = 0.0f;
        for (kk = 0; kk < k; kk++) {
            float cik = c_nk[(size_t)i * k + kk];
            float pik = 0.0f;
            for (t = b; t < e; t++) pik = pik + (nei_w[t] * c_nk[(size_t)nei_idx[t] * k + kk]);
            if (cik > FLT_MIN) {                                   /* MINFLOAT, :2727 */
                float logpkfki = logpkfki_nk[(size_t)i * k + kk];
                float dik = (float)(cik * (logpkfki - log((double)cik)));  /* :2731 */
                float gik = cik * pik;                                     /* :2732 */
                cd = cd + dik; cg = cg + gik;
            }
            fi = fi + pkfki_nk[(size_t)i * k + kk];                /* :2739 */
            zi = (float)(zi + exp((double)(beta * pik)));          /* :2740 */
        }
        cl = (float)(cl + log(fi));                                /* :2744 */
        cz = (float)(cz - log((double)zi));                        /* :2745 */
    }
    cu = (float)(cd + 0.5 * beta * cg);                            /* :2750 */
    cm = cd + beta * cg + cz;                                      /* :2751 */
    crit6[0] = cd; crit6[1] = cg; crit6[2] = cu; crit6[3] = cm; crit6[4] = cl; crit6[5] = cz;
}

int orc_converged(int n, int k, const float* c_nk, const float* cold_nk, float thres)
{
    float maxdif = 0.0f;                                           /* nem_alg.c:2077-2088 */
    size_t t, m = (size_t)n * k;
    for (t = 0; t < m; t++) {
        float dif = c_nk[t] - cold_nk[t];
        if (dif < 0) dif = -dif;
        if (dif > maxdif) maxdif = dif;
    }
    return maxdif < thres;
}

/* ----------------------------------------------------------------- loop */
static int run_from_params(const orc_problem* p, orc_state* s, int reseed);

int orc_run(const orc_problem* p, orc_state* s)
{
    return run_from_params(p, s, 1);
}

/* INIT_RANDOM: RandNemAlgo (nem_alg.c:1574-1742) = InitPara (:1200-1281) + n_starts x { MakeRandomPara
   (:1381-1473), ComputePartitionFromPara, NemAlgo }, best start by the chosen criterion (DEFAULT_CRIT = M,
   nem_typ.h:80), then EstimPara on the best partition.  The draws are libc random() after ONE srandom(seed)
   (nem_exe.c:621), shared with the TIE_RANDOM tie-breaks exactly as in the reference. */
int orc_run_random(const orc_problem* p, orc_state* s, int n_starts, unsigned seed, int* best_start)
{
    int n = p->n, d = p->d, k = p->k, h, j, i, irandom, nbsucc = 0, best = -1, err = ORC_STS_OK;
    size_t nk = (size_t)n * k;
    float* dispsam = (float*)malloc(sizeof(float) * d);
    float* c1 = (float*)calloc(nk, sizeof(float));
    float* bestc = (float*)malloc(sizeof(float) * nk);
    float bestcrit[6] = {0, 0, 0, 0, 0, 0};
    int best_iters = 0, best_conv = 0, ek = 0;

    srandom(seed);
    /* InitPara: dispersion of the whole sample = class 0 of an M-step with everything in class 0 */
    for (i = 0; i < n; i++) c1[(size_t)i * k] = 1.0f;
    orc_mstep(n, d, k, p->x, c1, p->disper, p->propor, s->prop_k, s->center_kd, s->disp_kd, s->nbobs_k,
              s->nbobs_kd, s->iner_kd, &ek);
    for (j = 0; j < d; j++) dispsam[j] = s->disp_kd[j];
    free(c1);

    for (irandom = 0; irandom < n_starts; irandom++) {
        /* MakeRandomPara */
        for (h = 0; h < k; h++) for (j = 0; j < d; j++) s->disp_kd[h * d + j] = dispsam[j] / k;      /* :1400 */
        for (h = 0; h < k; h++) s->prop_k[h] = (float)(1.0 / k);                                     /* :1405 */
        for (h = 0; h < k; h++) {
            int ipt = 0, again = 1, ndraw;
            for (ndraw = 0; again && ndraw < 100; ndraw++) {                                        /* :1419 */
                int g;
                /* RandomInteger(0, npt-1), nem_rnd.c:40-63: no draw is made when Mini >= Maxi */
                if (0 >= n - 1) ipt = n - 1;
                else ipt = (int)(random() % (long)n);
                for (g = 0, again = 0; g < h && !again; g++) {
                    int different = 0;
                    for (j = 0; j < d; j++)
                        if (s->center_kd[g * d + j] != (float)p->x[(size_t)ipt * d + j]) different = 1;
                    if (!different) again = 1;
                }
            }
            for (j = 0; j < d; j++) s->center_kd[h * d + j] = (float)p->x[(size_t)ipt * d + j];       /* :1457 */
        }
        err = run_from_params(p, s, 0);
        if (err == ORC_STS_OK) {
            nbsucc++;
            if (nbsucc == 1 || s->crit[3] > bestcrit[3]) {                                          /* :1676-1697 */
                memcpy(bestc, s->c_nk, sizeof(float) * nk);
                memcpy(bestcrit, s->crit, sizeof bestcrit);
                best = irandom; best_iters = s->iters; best_conv = s->converged;
            }
        }
    }
    if (nbsucc > 0) {
        err = ORC_STS_OK;
        memcpy(s->c_nk, bestc, sizeof(float) * nk);
        /* (the reference restores saved centres / dispersions first; they only matter for missing data) */
        orc_mstep(n, d, k, p->x, s->c_nk, p->disper, p->propor, s->prop_k, s->center_kd, s->disp_kd, s->nbobs_k,
                  s->nbobs_kd, s->iner_kd, &ek);                                                    /* :1711 */
        memcpy(s->crit, bestcrit, sizeof bestcrit);
        s->iters = best_iters; s->converged = best_conv;
    }
    if (best_start) *best_start = best;
    free(dispsam); free(bestc);
    return err;
}

static int run_from_params(const orc_problem* p, orc_state* s, int reseed)
{
    int n = p->n, d = p->d, k = p->k;
    int ncem = (p->algo == ORC_ALGO_NCEM);
    int err = ORC_STS_OK, iter, converged = 0, own_pk = 0, own_lp = 0;
    unsigned sweep = 0;
    size_t nk = (size_t)n * k;
    float* cold = (float*)malloc(sizeof(float) * nk);
    double t0;

    if (!s->pkfki_nk) { s->pkfki_nk = (double*)malloc(sizeof(double) * nk); own_pk = 1; }
    if (!s->logpkfki_nk) { s->logpkfki_nk = (float*)malloc(sizeof(float) * nk); own_lp = 1; }
    if (reseed && p->tie_rule == ORC_TIE_LIBC) srandom(p->tie_seed);   /* nem_exe.c:621 */
    s->n_zero_density = 0; s->emptyk = 0;

    memset(s->c_nk, 0, sizeof(float) * nk);                        /* calloc, nem_exe.c:524-526 */

    /* INIT_PARAM_FILE: ComputePartitionFromPara(Needinit=1) (nem_alg.c:1160, 1967-1981):
       densities, a "blind" beta=0 sweep, then a sweep with the real beta */
    orc_density(n, d, k, p->x, s->prop_k, s->center_kd, s->disp_kd, s->pkfki_nk, s->logpkfki_nk);
    s->n_zero_density += orc_sweep(n, k, p->nei_ptr, p->nei_idx, p->nei_w, 0.0f, s->pkfki_nk, ncem,
                                   p->tie_rule, p->tie_seed, sweep++, s->c_nk);
    s->n_zero_density += orc_sweep(n, k, p->nei_ptr, p->nei_idx, p->nei_w, p->beta, s->pkfki_nk, ncem,
                                   p->tie_rule, p->tie_seed, sweep++, s->c_nk);
    draw_log("init", 0);

    /* NemAlgo (nem_alg.c:1789-1840) */
    {
        int t;
        for (t = 0; t < 6; t++) s->crit[t] = 0.0f;                 /* Criteria = {0}, nem_exe.c:264 */
    }
    if (p->cvtest == ORC_CV_CRIT_LOGGED)                           /* a logging run: WriteLogCrit after the initial sweep */
        orc_crit(n, k, p->nei_ptr, p->nei_idx, p->nei_w, p->beta, s->c_nk, s->pkfki_nk, s->logpkfki_nk, s->crit);
    t0 = now_s();
    for (iter = 1; iter <= p->it_max && !converged && err == ORC_STS_OK; iter++) {
        const float oldcrit = s->crit[3];                          /* ChosenCrit(CRIT_M), :1802 */
        memcpy(cold, s->c_nk, sizeof(float) * nk);                 /* :1801 */
        if (!p->param_fix)                                         /* :1806 */
            err = orc_mstep(n, d, k, p->x, s->c_nk, p->disper, p->propor, s->prop_k, s->center_kd,
                            s->disp_kd, s->nbobs_k, s->nbobs_kd, s->iner_kd, &s->emptyk);
        if (err == ORC_STS_OK) {                                   /* :1815-1829 */
            orc_density(n, d, k, p->x, s->prop_k, s->center_kd, s->disp_kd, s->pkfki_nk, s->logpkfki_nk);
            s->n_zero_density += orc_sweep(n, k, p->nei_ptr, p->nei_idx, p->nei_w, p->beta, s->pkfki_nk,
                                           ncem, p->tie_rule, p->tie_seed, sweep++, s->c_nk);
            draw_log("iter", iter);
            if (p->cvtest == ORC_CV_CLAS) converged = orc_converged(n, k, s->c_nk, cold, p->cvthres);
            else if (p->cvtest == ORC_CV_CRIT || p->cvtest == ORC_CV_CRIT_LOGGED) {     /* nem_alg.c:2090-2105 */
                float curcrit, critdif;
                orc_crit(n, k, p->nei_ptr, p->nei_idx, p->nei_w, p->beta, s->c_nk, s->pkfki_nk, s->logpkfki_nk, s->crit);
                curcrit = s->crit[3];
                if (curcrit != 0) critdif = (float)fabs((curcrit - oldcrit) / curcrit);
                else critdif = FLT_MAX;                            /* MAXFLOAT */
                converged = (critdif < p->cvthres);
            }
        }
    }
    iter = iter - 1;                                               /* :1842 */
    g_loop_seconds = now_s() - t0;

    if (iter == 0) {                                               /* :1845-1851 */
        orc_mstep(n, d, k, p->x, s->c_nk, p->disper, p->propor, s->prop_k, s->center_kd,
                  s->disp_kd, s->nbobs_k, s->nbobs_kd, s->iner_kd, &s->emptyk);
        orc_density(n, d, k, p->x, s->prop_k, s->center_kd, s->disp_kd, s->pkfki_nk, s->logpkfki_nk);
    }
    orc_crit(n, k, p->nei_ptr, p->nei_idx, p->nei_w, p->beta, s->c_nk, s->pkfki_nk, s->logpkfki_nk, s->crit); /* :1852 */

    s->iters = iter;
    s->converged = converged;
    free(cold);
    if (own_pk) { free(s->pkfki_nk); s->pkfki_nk = NULL; }
    if (own_lp) { free(s->logpkfki_nk); s->logpkfki_nk = NULL; }
    return err;
}
