/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never part of the product path).
 *
 * A thin in-memory driver around the *unmodified* reference NEM sources.  It is
 * compiled TOGETHER with the reference C files where they lie under
 * /root/reference/ppanggolin/NEM (see oracle/Makefile, target `ref`) into
 * oracle/_ref/libnem_ref.so.  Nothing from the reference is copied here: this
 * file only #includes the reference's public headers at build time and calls
 * its public symbols
 *     nem()            nem_exe.h:23-35   (file-in / file-out entry)
 *     ClassifyByNem()  nem_alg.h:10-18   (in-memory EM driver)
 *
 * Why it exists
 *   - `.uf` keeps only 3 decimals (nem_exe.c:1677) and ClassifM is a
 *     function-local static (nem_exe.c:263), so 1e-6 posterior checks need the
 *     full-precision matrices: we fill DataT/SpatialT/StatModelT/NemParaT the
 *     way nem() does (nem_exe.c:296-435, 469-574) and call ClassifyByNem.
 *   - it gives the CPU baseline timing of the real reference EM loop
 *     (bench.py, cpu_baseline.kind == "reference").
 *
 * The struct filling mirrors nem() with these deliberate differences:
 *   - srandom(seed) with a caller-provided seed instead of time(NULL)
 *     (nem_exe.c:353,621) so tie-breaks are reproducible;
 *   - out_stderr goes to an in-memory stream whose text is returned, so the
 *     iteration count ("NEM converged after %d iterations",
 *     nem_alg.c:1868-1873) can be parsed by the caller.
 */
#define _GNU_SOURCE
#include "nem_typ.h"
#include "nem_alg.h"
#include "nem_mod.h"
#include "genmemo.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/*
 * Run the reference EM driver on in-memory inputs.
 *
 *  x_nd        N*D floats (0/1), row-major, as ReadMatrixFile would produce
 *  nei_ptr     N+1 CSR row pointers; nei_idx 0-based neighbour index,
 *              nei_w weights, in .nei file order (after ReadPtsNeighs filtering)
 *  algo        AlgoET (0 nem, 1 ncem)            disper DisperET   propor ProporET
 *  cvtest      CvemET (0 none, 1 clas, 2 crit)   param_mode ParamFileET (0 fix, 1 init)
 *  prop_k, center_kd, disp_kd   in: initial parameters (.m contents); out: final
 *  classif_nk  out: N*K posteriors (full precision)
 *  nbobs_k     out: K
 *  crit6       out: D,G,U,M,L,Z
 *  log_text/log_cap  out: text the reference wrote to out_stderr
 *  secs        out: wall seconds spent inside ClassifyByNem
 * returns the StatusET of ClassifyByNem.
 */
int ref_classify_ex(int n, int d, int k,
                    const float* x_nd,
                    const int* nei_ptr, const int* nei_idx, const float* nei_w,
                    int algo, float beta, int disper, int propor,
                    int cvtest, float cvthres, int nbiters, int param_mode,
                    long seed, int init_mode, int nb_random_inits,
                    float* prop_k, float* center_kd, float* disp_kd,
                    float* classif_nk, float* nbobs_k, float* crit6,
                    char* log_text, int log_cap, double* secs);

/* The reference's own per-iteration log (<Fname>.log, StartLogFile nem_alg.c:1478-1498): the next ref_classify*
   call writes it to `path` (DoLog = TRUE, LogName = path, as nem_exe.c:446-447 does); NULL or "" switches it off. */
static char g_log_name[LEN_FILENAME + 1] = "";
void ref_set_log(const char* path)
{
    g_log_name[0] = '\0';
    if (path) strncat(g_log_name, path, LEN_FILENAME);
}

int ref_classify(int n, int d, int k,
                 const float* x_nd,
                 const int* nei_ptr, const int* nei_idx, const float* nei_w,
                 int algo, float beta, int disper, int propor,
                 int cvtest, float cvthres, int nbiters, int param_mode,
                 long seed,
                 float* prop_k, float* center_kd, float* disp_kd,
                 float* classif_nk, float* nbobs_k, float* crit6,
                 char* log_text, int log_cap, double* secs)
{
    return ref_classify_ex(n, d, k, x_nd, nei_ptr, nei_idx, nei_w, algo, beta, disper, propor, cvtest, cvthres,
                           nbiters, param_mode, seed, INIT_PARAM_FILE, DEFAULT_NBRANDINITS, prop_k, center_kd,
                           disp_kd, classif_nk, nbobs_k, crit6, log_text, log_cap, secs);
}

/* Same with the start mode of nem()'s init_mode argument (INIT_PARAM_FILE = 2, INIT_RANDOM = 1, nem_typ.h) and the
   number of random starts (NbRandomInits, DEFAULT_NBRANDINITS = 50 in nem()). */
int ref_classify_ex(int n, int d, int k,
                    const float* x_nd,
                    const int* nei_ptr, const int* nei_idx, const float* nei_w,
                    int algo, float beta, int disper, int propor,
                    int cvtest, float cvthres, int nbiters, int param_mode,
                    long seed, int init_mode, int nb_random_inits,
                    float* prop_k, float* center_kd, float* disp_kd,
                    float* classif_nk, float* nbobs_k, float* crit6,
                    char* log_text, int log_cap, double* secs)
{
    DataT      data;
    NemParaT   para;
    SpatialT   spatial;
    StatModelT model;
    CriterT    crit;
    int        i, sts, maxnei = 0;
    char*      membuf = NULL;
    size_t     memlen = 0;
    double     t0;

    memset(&data, 0, sizeof data);
    memset(&para, 0, sizeof para);
    memset(&spatial, 0, sizeof spatial);
    memset(&model, 0, sizeof model);
    memset(&crit, 0, sizeof crit);

    out_stderr = open_memstream(&membuf, &memlen);

    /* DataT -- nem_exe.c:472-496 */
    data.NbPts = n;
    data.NbVars = d;
    data.NbMiss = 0;
    data.PointsM = (float*)malloc(sizeof(float) * (size_t)n * d);
    memcpy(data.PointsM, x_nd, sizeof(float) * (size_t)n * d);
    data.LabelV = NULL;
    data.SiteVisitV = (int*)malloc(sizeof(int) * (size_t)n);
    for (i = 0; i < n; i++) data.SiteVisitV[i] = i;
    data.SortPos_ND = NULL;

    /* SpatialT -- nem_exe.c:563, ReadPtsNeighs nem_exe.c:1342-1478 */
    spatial.Type = TYPE_SPATIAL;
    spatial.NeighData.PtsNeighsV = (PtNeighsT*)calloc((size_t)n, sizeof(PtNeighsT));
    for (i = 0; i < n; i++) {
        int nb = nei_ptr[i + 1] - nei_ptr[i], j;
        spatial.NeighData.PtsNeighsV[i].NbNeigh = nb;
        spatial.NeighData.PtsNeighsV[i].NeighsV = (NeighT*)calloc((size_t)(nb > 0 ? nb : 1), sizeof(NeighT));
        for (j = 0; j < nb; j++) {
            spatial.NeighData.PtsNeighsV[i].NeighsV[j].Index = nei_idx[nei_ptr[i] + j];
            spatial.NeighData.PtsNeighsV[i].NeighsV[j].Weight = nei_w[nei_ptr[i] + j];
        }
        if (nb > maxnei) maxnei = nb;
    }
    spatial.MaxNeighs = maxnei;

    /* StatModelT -- nem_exe.c:296, 312-336, 378, 412-431 */
    model.Spec.K = k;
    model.Spec.ClassFamily = FAMILY_BERNOULLI;
    model.Spec.ClassDisper = (DisperET)disper;
    model.Spec.ClassPropor = (ProporET)propor;
    model.Spec.BetaModel = BETA_FIX;
    model.Para.Beta = beta;
    model.Para.Prop_K = (float*)calloc((size_t)k, sizeof(float));
    model.Para.Disp_KD = (float*)calloc((size_t)k * d, sizeof(float));
    model.Para.Center_KD = (float*)calloc((size_t)k * d, sizeof(float));
    model.Para.NbObs_K = (float*)calloc((size_t)k, sizeof(float));
    model.Para.NbObs_KD = (float*)calloc((size_t)k * d, sizeof(float));
    model.Para.Iner_KD = (float*)calloc((size_t)k * d, sizeof(float));
    model.Desc.DispSam_D = (float*)calloc((size_t)d, sizeof(float));
    model.Desc.MiniSam_D = (float*)calloc((size_t)d, sizeof(float));
    model.Desc.MaxiSam_D = (float*)calloc((size_t)d, sizeof(float));
    memcpy(model.Para.Prop_K, prop_k, sizeof(float) * (size_t)k);
    memcpy(model.Para.Center_KD, center_kd, sizeof(float) * (size_t)k * d);
    memcpy(model.Para.Disp_KD, disp_kd, sizeof(float) * (size_t)k * d);

    /* NemParaT -- nem_exe.c:334-362, 371-435 */
    para.Algo = (AlgoET)algo;
    para.BtaHeuStep = DEFAULT_BTAHEUSTEP;
    para.BtaHeuMax = DEFAULT_BTAHEUMAX;
    para.BtaHeuDDrop = DEFAULT_BTAHEUDDROP;
    para.BtaHeuDLoss = DEFAULT_BTAHEUDLOSS;
    para.BtaHeuLLoss = DEFAULT_BTAHEULLOSS;
    para.BtaPsGrad.NbIter = DEFAULT_BTAGRADNIT;
    para.BtaPsGrad.ConvThres = DEFAULT_BTAGRADCVTH;
    para.BtaPsGrad.Step = DEFAULT_BTAGRADSTEP;
    para.BtaPsGrad.RandInit = DEFAULT_BTAGRADRAND;
    para.Crit = DEFAULT_CRIT;
    para.CvThres = cvthres;
    para.CvTest = (CvemET)cvtest;
    para.DoLog = g_log_name[0] ? TRUE : FALSE;
    para.NbIters = nbiters;
    para.NbEIters = DEFAULT_NBEITERS;
    para.NbRandomInits = nb_random_inits;
    para.Seed = seed;
    para.Format = FORMAT_FUZZY;
    para.InitMode = (InitET)init_mode;
    para.MissMode = MISSING_REPLACE;      /* NemPara is zero-initialised in nem(): nem_exe.c:260 */
    para.ParamFileMode = (ParamFileET)param_mode;
    para.SortedVar = DEFAULT_SORTEDVAR;
    para.NeighSpec = NEIGH_FILE;
    para.VisitOrder = DEFAULT_ORDER;
    para.SiteUpdate = DEFAULT_UPDATE;
    para.TieRule = DEFAULT_TIE;
    para.Debug = FALSE;
    strcpy(para.LogName, g_log_name);

    /* CriterT -- MakeErrinfo("") => Kr = 0 (nem_exe.c:1157-1161) */
    crit.Errinfo.Kc = k;
    crit.Errinfo.Kr = 0;
    crit.Errinfo.Km = k;
    crit.Errinfo.TieRule = para.TieRule;

    /* ClassifM comes from calloc in nem() (nem_exe.c:524-526, genmemo.c:30) */
    memset(classif_nk, 0, sizeof(float) * (size_t)n * k);
    srandom((unsigned)seed);              /* nem_exe.c:621, fixed instead of time() */

    t0 = now_s();
    sts = ClassifyByNem(&para, &spatial, &data, &model, classif_nk, &crit);
    if (secs) *secs = now_s() - t0;

    memcpy(prop_k, model.Para.Prop_K, sizeof(float) * (size_t)k);
    memcpy(center_kd, model.Para.Center_KD, sizeof(float) * (size_t)k * d);
    memcpy(disp_kd, model.Para.Disp_KD, sizeof(float) * (size_t)k * d);
    if (nbobs_k) memcpy(nbobs_k, model.Para.NbObs_K, sizeof(float) * (size_t)k);
    if (crit6) {
        crit6[0] = crit.D; crit6[1] = crit.G; crit6[2] = crit.U;
        crit6[3] = crit.M; crit6[4] = crit.L; crit6[5] = crit.Z;
    }

    fflush(out_stderr);
    fclose(out_stderr);
    if (log_text && log_cap > 0) {
        size_t m = memlen < (size_t)(log_cap - 1) ? memlen : (size_t)(log_cap - 1);
        memcpy(log_text, membuf, m);
        log_text[m] = '\0';
    }
    free(membuf);

    for (i = 0; i < n; i++) free(spatial.NeighData.PtsNeighsV[i].NeighsV);
    free(spatial.NeighData.PtsNeighsV);
    free(data.PointsM); free(data.SiteVisitV); free(data.SortPos_ND);
    free(model.Para.Prop_K); free(model.Para.Disp_KD); free(model.Para.Center_KD);
    free(model.Para.NbObs_K); free(model.Para.NbObs_KD); free(model.Para.Iner_KD);
    free(model.Desc.DispSam_D); free(model.Desc.MiniSam_D); free(model.Desc.MaxiSam_D);
    return sts;
}

/* One reference M-step (EstimPara, nem_mod.h:21-31) on an in-memory partition:
   used to pin the oracle's M-step restatement in isolation. */
int ref_estim_para(int n, int d, int k, const float* x_nd, const float* c_nk,
                   int disper, int propor,
                   float* prop_k, float* center_kd, float* disp_kd,
                   float* nbobs_k, float* nbobs_kd, float* iner_kd, int* emptyk)
{
    DataT      data;
    ModelSpecT spec;
    ModelParaT mp;
    int        i, j, sts;

    memset(&data, 0, sizeof data);
    memset(&spec, 0, sizeof spec);
    memset(&mp, 0, sizeof mp);
    out_stderr = stderr;
    data.NbPts = n; data.NbVars = d; data.NbMiss = 0;
    data.PointsM = (float*)x_nd;
    /* zeros (index order) then ones (index order) == glibc 2.35 qsort result of
       ModelPreprocess (nem_alg.c:672-726) on 0/1 data */
    data.SortPos_ND = (int*)malloc(sizeof(int) * (size_t)n * d);
    for (j = 0; j < d; j++) {
        int p = 0;
        for (i = 0; i < n; i++) if (x_nd[(size_t)i * d + j] == 0.0f) data.SortPos_ND[(size_t)(p++) * d + j] = i;
        for (i = 0; i < n; i++) if (x_nd[(size_t)i * d + j] != 0.0f) data.SortPos_ND[(size_t)(p++) * d + j] = i;
    }
    spec.K = k; spec.ClassFamily = FAMILY_BERNOULLI;
    spec.ClassDisper = (DisperET)disper; spec.ClassPropor = (ProporET)propor;
    spec.BetaModel = BETA_FIX;
    mp.Prop_K = prop_k; mp.Center_KD = center_kd; mp.Disp_KD = disp_kd;
    mp.NbObs_K = nbobs_k; mp.NbObs_KD = nbobs_kd; mp.Iner_KD = iner_kd;
    sts = EstimPara(c_nk, &data, k, MISSING_REPLACE, &spec, emptyk, &mp);
    free(data.SortPos_ND);
    return sts;
}
