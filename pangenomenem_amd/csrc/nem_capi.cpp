// nem_capi.cpp -- the drop-in `nem()` entry point (reference: nem_exe.h:23-35, nem_exe.c:239-704).
//
// Same symbol, argument list, file formats and return codes as the reference's only FFI entry;
// the EM loop runs on the GPU through the nemgpu_* engine.  Text written to <Fname>.stderr keeps
// the reference's wording where a caller could plausibly grep it.
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <future>
#include <deque>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <memory>
#include <unistd.h>
#include <string>
#include <vector>

#include "nem_internal.hpp"

using namespace nemk;

namespace {

// ExitET, lib_io.h:22-34
enum { EXIT_OK_ = 0, EXIT_W_RESULT_ = 1, EXIT_E_ARGS_ = 2, EXIT_E_FILE_ = 3, EXIT_E_MEMORY_ = 4, EXIT_E_SYSTEM_ = 5, EXIT_E_BUG_ = 6 };

const char* kVersion = "1.08-a";   // NemVersionStrC, nem_exe.c:233 (banner compatibility)

int get_enum(const char* s, const char* const* tab, int n)   // GetEnum, nem_exe.c:1784-1809
{
    if (!s) return -1;
    for (int i = 0; i < n; i++) if (!strcmp(s, tab[i])) return i;
    return -1;
}

struct Log {
    FILE* fp = nullptr;
    bool owned = false;
    void open(const std::string& base, int dolog)
    {
        if (dolog) {
            std::string name = std::string(base).substr(0, 200) + ".stderr";   // LEN_FILENAME, nem_typ.h:50
            fp = fopen(name.c_str(), "w");
            owned = fp != nullptr;
        }
        if (!fp) { fp = stderr; owned = false; }
    }
    void close() { if (owned && fp) fclose(fp); fp = nullptr; }   // the reference also fclose()s stderr; we do not
    void pr(const char* fmt, ...)
    {
        va_list ap; va_start(ap, fmt); vfprintf(fp, fmt, ap); va_end(ap);
    }
};

}  // namespace

namespace {

// ---- <Fname>.log, the reference's per-iteration log (dolog != 0) ---------------------------------------------
// Line layout: StartLogFile (nem_alg.c:1478-1498), the INIT_PARAM_FILE preamble (:1151-1158), WriteLogHeader
// (:1883-1945), and per iteration "%4d " + criteria before and after the E-step sweep (WriteLogCrit, :2620-2646,
// called at :2361 and :2398) + WriteLogClasses (:1993-2052).
// asctime(localtime()) share static buffers; nem() may run on several threads at once (nem_many)
std::string date_line()
{
    time_t timer = time(nullptr);
    struct tm tmv;
    char buf[64];
    if (localtime_r(&timer, &tmv) == nullptr || asctime_r(&tmv, buf) == nullptr) return "\n";
    return buf;                                                              // ends with '\n' like asctime()
}

float log_mult(int npt)
{
    return (float)exp(-((int)(log(npt / 1000.) / log(10))) * log(10));      // :1495, :2638
}

void log_crit(std::string& out, const float crit6[6], float mult)
{
    char buf[128];
    const int m = snprintf(buf, sizeof buf, " %5.0f %5.0f %5.3f", (double)(float)(crit6[2] * mult), (double)(float)(crit6[3] * mult),
                           (double)std::nanf(""));                           // U, M, error rate (no reference partition)
    out.append(buf, (size_t)m);
}
void log_header(FILE* fl, int k, int d)
{
    fprintf(fl, "%4s  %5s %5s %5s", "It", "UM", "PM", "Er");
    fprintf(fl, " %3s%-2d %3s%-2d %3s%-2d", "UE", 1, "PE", 1, "Er", 1);       // NbEIters = 1
    fprintf(fl, " ");
    fprintf(fl, " %5s", "Beta");
    fprintf(fl, " ");
    for (int h = 0; h < k; h++) fprintf(fl, " %3s%02d", "P", h + 1);
    fprintf(fl, " ");
    // 3 K D column names " %3s%02d_%1d": put together by hand (thousands of fprintf calls cost more than the EM loop)
    std::string line;
    line.reserve((size_t)3 * k * d * 12 + 16);
    auto names = [&](char letter) {
        for (int h = 0; h < k; h++)
            for (int j = 0; j < d; j++) {
                char buf[32];
                int m = 0;
                buf[m++] = ' '; buf[m++] = ' '; buf[m++] = ' '; buf[m++] = letter;
                if (h + 1 < 100) { buf[m++] = (char)('0' + (h + 1) / 10); buf[m++] = (char)('0' + (h + 1) % 10); }
                else m += snprintf(buf + m, sizeof buf - (size_t)m, "%02d", h + 1);
                buf[m++] = '_';
                char dg[12];
                int nd = 0;
                for (int v = j + 1; v > 0; v /= 10) dg[nd++] = (char)('0' + v % 10);
                while (nd > 0) buf[m++] = dg[--nd];
                line.append(buf, (size_t)m);
            }
    };
    names('M'); line += " "; names('D'); line += " "; names('n'); line += "\n";
    fwrite(line.data(), 1, line.size(), fl);
}

// " %<width>.<dec>f" of a float, as printf prints it (dec <= 3).  A float times 10^dec is exact in a double (24 + 10
// bits), so rint() -- to nearest, ties to even, printf's rule in the default rounding mode -- of that product is the
// decimal rounding of the exact binary value; negative, -0.0, huge and non-finite values go to snprintf.
static inline void put_fixed(std::string& out, float v, int width, int dec)
{
    static const double p10[4] = {1.0, 10.0, 100.0, 1000.0};
    char buf[48];
    if (v >= 0.0f && v < 1.0e9f && !std::signbit(v)) {
        unsigned long long r = (unsigned long long)rint((double)v * p10[dec]);
        char tmp[32];
        int n = 0;
        for (int i = 0; i < dec; i++) { tmp[n++] = (char)('0' + r % 10); r /= 10; }
        if (dec > 0) tmp[n++] = '.';
        do { tmp[n++] = (char)('0' + r % 10); r /= 10; } while (r != 0);
        int m = 0;
        buf[m++] = ' ';
        for (int pad = width - n; pad > 0; pad--) buf[m++] = ' ';
        while (n > 0) buf[m++] = tmp[--n];
        out.append(buf, (size_t)m);
    } else {
        const int m = snprintf(buf, sizeof buf, " %*.*f", width, dec, (double)v);
        out.append(buf, (size_t)m);
    }
}

struct LogParams { std::vector<float> prop, center, disp, nk; };

void log_classes(std::string& line, const LogParams& P, float beta, int k, int d, bool sizes_known)
{
    const std::vector<float>&prop = P.prop, &center = P.center, &disp = P.disp, &nk = P.nk;
    // WriteLogClasses' formats: " %5.3f" beta and proportions, " %7.3f" centres and dispersions, " %7.1f" NbObs_KD
    line.reserve(line.size() + (size_t)k * d * 3 * 9 + 64);
    line += " ";
    put_fixed(line, beta, 5, 3);
    line += " ";
    for (int h = 0; h < k; h++) put_fixed(line, prop[h], 5, 3);
    line += " ";
    for (size_t t = 0; t < (size_t)k * d; t++) put_fixed(line, center[t], 7, 3);
    line += " ";
    for (size_t t = 0; t < (size_t)k * d; t++) put_fixed(line, disp[t], 7, 3);
    line += " ";
    // NbObs_KD: zero until the first EstimPara (calloc, nem_exe.c:320), then N_K for every organism (no missing data)
    for (int h = 0; h < k; h++) for (int j = 0; j < d; j++) put_fixed(line, sizes_known ? nk[h] : 0.0f, 7, 1);
    line += "\n";
}

// The INIT_RANDOM run with the reference's log (RandNemAlgo, nem_alg.c:1632-1636, 1662-1669, 1730-1732): per start
// "Random initialization %d :", line 0 (the start's own parameters; NbObs_KD is whatever the run so far left there --
// NaN from InitPara, :1270-1276, before the first start, then the sizes of the last EstimPara, which nothing resets),
// WriteLogHeader and NemAlgo's line per iteration; "Best start was %d (U = %g)" at the end.
// The text is put together and written by a helper thread, in order, while the device runs the next logged step (a
// line is 9 000 numbers at configs[1] size, a run of 50 starts some 30 MB).
struct LogJob {
    enum Kind { TEXT, LINE, HEADER } kind = TEXT;
    std::string text;                                                       // TEXT: as it is; LINE: what precedes the criteria
    float cb[6], ca[6];
    LogParams P;
    bool blank_after = false;                                               // LINE: Needinit's empty line (:1985-1986)
};
// The lines are formatted by a few threads and written by one, in the order they were pushed: with the starts in lock step
// all 1 350 lines of a 50-start run (30 MB of text at configs[1] size) arrive when the last step is through, and one thread
// formatting them was 30 ms at the end of an 80 ms call.
class LogWriter {
public:
    LogWriter(FILE* fl, float mult, float beta, int k, int d, int formatters = 3) : fl_(fl), mult_(mult), beta_(beta), k_(k), d_(d)
    {
        for (int i = 0; i < formatters; i++) fmt_.emplace_back([this] { format_loop(); });
        writer_ = std::thread([this] { write_loop(); });
    }
    ~LogWriter() { finish(); }
    void push(LogJob&& j)
    {
        auto sl = std::make_shared<Slot>();
        sl->j = std::move(j);
        sl->ready = sl->j.kind != LogJob::LINE;                              // (TEXT and HEADER need no formatting)
        if (sl->j.kind == LogJob::TEXT) sl->out = sl->j.text;
        {
            std::lock_guard<std::mutex> g(m_);
            order_.push_back(sl);
            if (!sl->ready) work_.push_back(sl);
        }
        cv_work_.notify_one(); cv_ready_.notify_one();
    }
    void finish()
    {
        if (!writer_.joinable()) return;
        { std::lock_guard<std::mutex> g(m_); done_ = true; }
        cv_work_.notify_all(); cv_ready_.notify_all();
        for (std::thread& t : fmt_) t.join();
        writer_.join();
    }
private:
    struct Slot { LogJob j; std::string out; bool ready = false; };
    void format_loop()
    {
        for (;;) {
            std::shared_ptr<Slot> sl;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_work_.wait(g, [this] { return done_ || !work_.empty(); });
                if (work_.empty()) return;
                sl = work_.front(); work_.pop_front();
            }
            std::string out = sl->j.text;
            log_crit(out, sl->j.cb, mult_); log_crit(out, sl->j.ca, mult_);
            log_classes(out, sl->j.P, beta_, k_, d_, true);
            if (sl->j.blank_after) out += "\n";
            { std::lock_guard<std::mutex> g(m_); sl->out = std::move(out); sl->ready = true; }
            cv_ready_.notify_all();
        }
    }
    void write_loop()
    {
        for (;;) {
            std::shared_ptr<Slot> sl;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_ready_.wait(g, [this] { return (!order_.empty() && order_.front()->ready) || (done_ && order_.empty()); });
                if (order_.empty()) return;
                sl = order_.front(); order_.pop_front();
            }
            if (sl->j.kind == LogJob::HEADER) log_header(fl_, k_, d_);
            else fwrite(sl->out.data(), 1, sl->out.size(), fl_);
        }
    }
    FILE* fl_; float mult_, beta_; int k_, d_;
    std::mutex m_; std::condition_variable cv_work_, cv_ready_;
    std::deque<std::shared_ptr<Slot>> order_, work_; bool done_ = false;
    std::vector<std::thread> fmt_; std::thread writer_;
};

struct RandomLog {
    LogWriter* w; int k, d;
    std::vector<float> nk_left;                                              // NbObs_K as the run so far left it
};

void random_log_event(const nemgpu_log_event* ev, void* user)
{
    RandomLog& L = *static_cast<RandomLog*>(user);
    char t[96];
    LogJob j;
    if (ev->kind == NEMGPU_LOG_START) {
        snprintf(t, sizeof t, "\nRandom initialization %d :\n%4d ", ev->start + 1, 0);
        j.text = t;
        L.w->push(std::move(j));
        return;
    }
    if (ev->kind == NEMGPU_LOG_EMPTY) {
        snprintf(t, sizeof t, "%4d  Class %d empty at iteration %d\n", ev->iter, ev->emptyk, ev->iter);     // :1798, :1835-1837
        j.text = t;
        L.w->push(std::move(j));
        L.nk_left.assign(ev->nbobs_k, ev->nbobs_k + L.k);
        return;
    }
    const size_t kd = (size_t)L.k * L.d;
    j.kind = LogJob::LINE;
    j.P = LogParams{std::vector<float>(ev->prop, ev->prop + L.k), std::vector<float>(ev->center, ev->center + kd),
                    std::vector<float>(ev->disp, ev->disp + kd), L.nk_left};
    memcpy(j.cb, ev->crit_before, sizeof j.cb); memcpy(j.ca, ev->crit_after, sizeof j.ca);
    if (ev->iter > 0) {
        j.P.nk.assign(ev->nbobs_k, ev->nbobs_k + L.k);
        L.nk_left = j.P.nk;
        snprintf(t, sizeof t, "%4d ", ev->iter);
        j.text = t;
    }
    j.blank_after = ev->iter == 0;                                           // Needinit, :1985-1986
    L.w->push(std::move(j));
    if (ev->iter == 0) { LogJob h; h.kind = LogJob::HEADER; L.w->push(std::move(h)); }   // NemAlgo's header, :1783
}

// The INIT_PARAM_FILE run with the reference's log through nemgpu_run_logged: the iterations pipelined, the criteria of a
// batch's iterations evaluated together, the lines formatted by the writer's threads.
struct RunLog { LogWriter* w; int k, d; bool sizes; };

void run_log_event(const nemgpu_log_event* ev, void* user)
{
    RunLog& L = *static_cast<RunLog*>(user);
    char t[96];
    LogJob j;
    if (ev->kind == NEMGPU_LOG_EMPTY) {
        snprintf(t, sizeof t, "%4d  Class %d empty at iteration %d\n", ev->iter, ev->emptyk, ev->iter);     // :1798, :1835-1837
        j.text = t;
        L.w->push(std::move(j));
        return;
    }
    const size_t kd = (size_t)L.k * L.d;
    j.kind = LogJob::LINE;
    // NbObs_KD: zero until the first EstimPara (calloc, nem_exe.c:320), then N_K for every organism (no missing data)
    const bool sizes = ev->iter > 0 && L.sizes && ev->nbobs_k != nullptr;
    j.P = LogParams{std::vector<float>(ev->prop, ev->prop + L.k), std::vector<float>(ev->center, ev->center + kd),
                    std::vector<float>(ev->disp, ev->disp + kd),
                    sizes ? std::vector<float>(ev->nbobs_k, ev->nbobs_k + L.k) : std::vector<float>((size_t)L.k, 0.0f)};
    memcpy(j.cb, ev->crit_before, sizeof j.cb); memcpy(j.ca, ev->crit_after, sizeof j.ca);
    snprintf(t, sizeof t, "%4d ", ev->iter);
    j.text = t;
    j.blank_after = ev->iter == 0;                                           // Needinit, :1985-1986
    L.w->push(std::move(j));
    if (ev->iter == 0) { LogJob h; h.kind = LogJob::HEADER; L.w->push(std::move(h)); }   // NemAlgo's header, :1783
}

int run_logged_pipelined(nemgpu_engine* e, const nemgpu_config& cfg, int n, int d, int k, FILE* fl, nemgpu_result* res)
{
    const float mult = log_mult(n);
    fprintf(fl, "NEM log file  -  %s\n", date_line().c_str());
    fprintf(fl, "  Criteria are multiplied by %f\n\n", (double)mult);
    fprintf(fl, "Initializing parameters from given value :\n");
    fflush(fl);
    int rc;
    {
        LogWriter w(fl, mult, cfg.beta, k, d, 2);
        RunLog L{&w, k, d, !cfg.param_fix};
        rc = nemgpu_run_logged(e, res, run_log_event, &L);
        w.finish();
    }
    if (rc) return rc;
    if (res->iters == 0) {                                                  // :1845-1851
        int ek = 0;
        rc = nemgpu_mstep(e, &ek);
        if (rc != NEMGPU_OK && rc != NEMGPU_W_EMPTYCLASS) return rc;
        if ((rc = nemgpu_density(e))) return rc;
    }
    if (std::isnan(res->crit[0])) return nemgpu_criteria(e, res->crit);
    return NEMGPU_OK;
}

int run_random_logged(nemgpu_engine* e, const nemgpu_config& cfg, int n, int d, int k, FILE* fl, nemgpu_result* res, int* best)
{
    const float mult = log_mult(n);
    fprintf(fl, "NEM log file  -  %s\n", date_line().c_str());
    fprintf(fl, "  Criteria are multiplied by %f\n\n", (double)mult);
    fflush(fl);
    int rc;
    {
        LogWriter w(fl, mult, cfg.beta, k, d);
        RandomLog L{&w, k, d, std::vector<float>((size_t)k, std::nanf(""))};
        rc = nemgpu_run_random_logged(e, 50, cfg.tie_seed, res, best, random_log_event, &L);
        w.finish();
    }
    if (rc == NEMGPU_OK && *best >= 0) fprintf(fl, "Best start was %d (U = %g)\n", *best + 1, (double)res->crit[3]);
    return rc;
}

}  // namespace

extern "C" int nem(const char* Fname, const int nk, const char* algo, const float beta, const char* convergence,
                   const float convergence_th, const char* format, const int it_max, const int dolog,
                   const char* model_family, const char* proportion, const char* dispersion, const int init_mode)
{
    static const char* const AlgoStr[] = {"nem", "ncem", "gem"};            // nem_typ.h:556
    static const char* const CvStr[] = {"none", "clas", "crit"};            // :560
    static const char* const FormatStr[] = {"hard", "fuzzy"};               // :561
    static const char* const FamilyStr[] = {"norm", "lapl", "bern"};        // :575
    static const char* const DisperStr[] = {"s__", "sk_", "s_d", "skd"};    // :576
    static const char* const ProporStr[] = {"p_", "pk"};                    // :577
    static const char* const AlgoDes[] = {"NEM", "NCEM (C-step)", "GEM (Monte-Carlo at E-step)"};
    static const char* const DisperDes[] = {"S__", "SK_", "S_D", "S_KD"};
    static const char* const ProporDes[] = {"P_", "Pk"};

    if (!Fname) return EXIT_E_ARGS_;
    const std::string base = std::string(Fname).substr(0, 200);             // strncpy(..., LEN_FILENAME)
    Log lg;
    lg.open(base, dolog);
    lg.pr(" * * * NEM (spatial data clustering) v%s * * *\n", kVersion);
    lg.pr(" * * * MI355X-native engine (HIP, gfx950) behind the reference nem() interface * * *\n");

    if (nk <= 0) {                                                          // nem_exe.c:297-302
        lg.pr("Nb of classes must be > 0 (here %d)\n", nk);
        lg.close();
        return NEMGPU_E_ARG;
    }
    if (nk > kMaxK) {
        lg.pr("Nb of classes must be <= %d in this engine (here %d)\n", kMaxK, nk);
        lg.close();
        return EXIT_E_ARGS_;
    }

    // wall-clock of the call's phases, reported on one "[engine]" line of the text output
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    const clk::time_point t_start = clk::now();

    NemInputs in;
    std::string err;
    int sts;
    if ((sts = read_str_file(base, in, err)) != NEMGPU_OK) {                // :304-309 (raw StatusET returned)
        fprintf(stderr, "%s\n", err.c_str());
        lg.close();
        return sts;
    }

    // ---- argument decoding, nem_exe.c:371-435.  The reference records bad strings in `err` and then
    // overwrites it (:472), so most of them are not fatal there; the same fall-backs apply here.
    nemgpu_config cfg{};
    int a = get_enum(algo, AlgoStr, 3);
    if (a == -1) lg.pr(" Unknown type of algorithm %s\n", algo ? algo : "(null)");
    if (a == 2) {
        lg.pr(" Algorithm gem (Monte-Carlo E-step) is not supported by this engine\n");
        lg.pr("*** NEM error status : bad arguments\n");
        lg.close();
        return EXIT_E_ARGS_;
    }
    cfg.algo = (a == 1) ? NEMGPU_ALGO_NCEM : NEMGPU_ALGO_NEM;               // Algo == -1 behaves like NEM
    cfg.beta = beta;
    int cv = get_enum(convergence, CvStr, 3);
    cfg.cvtest = NEMGPU_CV_NONE;
    cfg.cvthres = 1.0f;
    if (cv == -1) lg.pr(" Unknown convergence test %s\n", convergence ? convergence : "(null)");
    else if (cv != 0) {
        if (convergence_th <= 0) lg.pr(" Conv threshold must be > 0 (here %f)\n", convergence_th);   // never satisfied
        else {
            // crit (nem_alg.c:2090-2105): a run that logs starts from the initial partition's criterion, one that
            // does not from Criteria = {0} (nem_exe.c:264)
            cfg.cvtest = cv == 1 ? NEMGPU_CV_CLAS : (dolog ? NEMGPU_CV_CRIT_LOGGED : NEMGPU_CV_CRIT);
            cfg.cvthres = convergence_th;
        }
    }
    int fm = get_enum(format, FormatStr, 2);
    if (fm == -1) lg.pr(" Unknown format %s\n", format ? format : "(null)");
    const bool hard = (fm == 0);
    cfg.it_max = it_max;
    if (it_max < 0) { lg.pr("Nb iterations must be >= 0 (here %d)\n", it_max); cfg.it_max = 0; }
    int fam = get_enum(model_family, FamilyStr, 3);
    if (fam != 2) {
        lg.pr(fam == -1 ? " Unknown family %s\n" : " Family %s is not supported by this engine (only bern)\n",
              model_family ? model_family : "(null)");
        lg.pr("*** NEM error status : bad arguments\n");
        lg.close();
        return EXIT_E_ARGS_;
    }
    int pr = get_enum(proportion, ProporStr, 2);
    if (pr == -1) lg.pr(" Unknown proportion %s\n", proportion ? proportion : "(null)");
    cfg.propor = (pr == 1) ? NEMGPU_PROP_K : NEMGPU_PROP__;                 // only PROPOR_K re-estimates (nem_mod.c:456)
    int dp = get_enum(dispersion, DisperStr, 4);
    if (dp == -1) lg.pr(" Unknown dispersion %s\n", dispersion ? dispersion : "(null)");
    cfg.disper = dp;
    // TieRule = TIE_RANDOM on random() after srandom(NemPara.Seed), Seed = time(NULL) (:353, :621): the same stream here,
    // drawn in the same order; NEM_MI355X_SEED fixes the seed, NEM_MI355X_TIE=hash|first selects a stateless rule
    cfg.tie_rule = NEMGPU_TIE_LIBC;
    if (const char* s = getenv("NEM_MI355X_TIE")) {
        if (!strcmp(s, "hash")) cfg.tie_rule = NEMGPU_TIE_HASH;
        else if (!strcmp(s, "first")) cfg.tie_rule = NEMGPU_TIE_FIRST;
    }
    cfg.tie_seed = (uint32_t)time(nullptr);
    if (const char* s = getenv("NEM_MI355X_SEED")) cfg.tie_seed = (uint32_t)strtoul(s, nullptr, 10);

    const std::string outname = base + (hard ? ".cf" : ".uf");              // :437-440

    // ---- inputs, nem_exe.c:469-574
    lg.pr("Reading points ...\n");
    if ((sts = read_dat_file(base, in, err)) != NEMGPU_OK) { fprintf(stderr, "%s\n", err.c_str()); lg.close(); return sts; }
    const bool random_init = (init_mode == 1);                              // INIT_RANDOM, nem_typ.h:217
    if (init_mode != 2 && !random_init) {                                   // INIT_PARAM_FILE, nem_typ.h:218
        lg.pr("Initialization mode %d is not supported by this engine (2 = parameter file, 1 = random starts)\n", init_mode);
        lg.pr("*** NEM error status : bad arguments\n");
        lg.close();
        return EXIT_E_ARGS_;
    }
    if (!random_init) {                                                     // nem_exe.c:513-519: only this mode reads <Fname>.m
        lg.pr("Reading parameter file ...\n");
        if ((sts = read_param_file(base, nk, in, err)) != NEMGPU_OK) { fprintf(stderr, "%s\n", err.c_str()); lg.close(); return sts; }
    }
    cfg.param_fix = (!random_init && in.param_mode == 2);
    if (in.type != 'N') {
        lg.pr("Reading neighborhood information ...\n");
        if ((sts = read_nei_file(base, in, err)) != NEMGPU_OK) { fprintf(stderr, "%s\n", err.c_str()); lg.close(); return sts; }
    } else {
        cfg.beta = 0.0f;                                                    // :572
        in.nei_ptr.assign(in.n + 1, 0);
    }

    lg.pr("\nData : ");
    if (!in.str_comment.empty()) lg.pr("%s\n", in.str_comment.c_str()); else lg.pr("\n");
    lg.pr("  file names =  %10s   |   nb points   = %10d\n", Fname, in.n);
    lg.pr("  type       =  %10s   |   dim         = %10d\n", in.type == 'N' ? "NoSpatial" : "Spatial", in.d);
    if (in.type != 'N') {
        lg.pr("Neighborhood system :\n  max neighb =  %10d\n", in.max_neighs);
        lg.pr("%s\n", in.nei_comment.c_str());
    }
    lg.pr("\nNEM parameters :\n");
    lg.pr("Type of algorithm : '%s'\n", AlgoDes[cfg.algo]);
    lg.pr("  beta       =  %10.2f   |   nk                    = %3d\n", cfg.beta, nk);
    lg.pr("                %10s   |   model                 = %s, %s %s\n", " ", "Bernoulli",
          ProporDes[pr == 1 ? 1 : 0], dp >= 0 ? DisperDes[dp] : "?");
    lg.pr("\n");

    if (dp == -1) {                                                         // InerToDisp default -> STS_E_FUNCARG
        lg.pr("*** NEM internal error : bad arguments\n");
        lg.close();
        return EXIT_E_BUG_;
    }

    // ---- the EM run on the GPU (ClassifyByNem, :624)
    const clk::time_point t_read = clk::now();
    clk::time_point t_up = t_read;
    nemgpu_engine* e = nullptr;
    nemgpu_result res{};
    bool full_log = false;
    // which GPU: the calling thread's current HIP device unless NEM_MI355X_DEVICE says otherwise (an index, or "auto":
    // processes spread over the node's GPUs -- PPanGGOLiN runs its chunks in a multiprocessing.Pool,
    // ppanggolin.py:1039-1095); decided behind the library's fork guard (nemgpu_default_device)
    int device = 0;
    int rc = nemgpu_default_device(&device);
    if (rc == NEMGPU_OK) rc = nemgpu_create(&e, in.n, in.d, nk, 0, in.n, device, nullptr);
    if (rc == NEMGPU_OK) rc = nemgpu_set_matrix_bits(e, in.xbits.data());
    if (rc == NEMGPU_OK) rc = nemgpu_set_graph(e, in.nei_ptr.data(), in.nei_idx.data(), in.nei_w.data());
    if (rc == NEMGPU_OK && !random_init) rc = nemgpu_set_params(e, in.prop.data(), in.center.data(), in.disp.data());
    if (rc == NEMGPU_OK) rc = nemgpu_configure(e, &cfg);
    t_up = clk::now();
    if (rc == NEMGPU_OK) {
        if (random_init) {
            // RandNemAlgo with the reference's 50 starts (DEFAULT_NBRANDINITS, nem_typ.h:94); the draws come from the
            // reference's generator seeded like its NemPara.Seed (time(NULL), or NEM_MI355X_SEED)
            int best = -1;
            // dolog: the reference's log of every start (the starts then run one after the other, one logged step per
            // host round trip); NEM_MI355X_LOG=0 keeps the lock-step run and writes a header-only log
            const char* lenv = getenv("NEM_MI355X_LOG");
            FILE* fl = (dolog && !(lenv && lenv[0] == '0')) ? fopen((base + ".log").c_str(), "w") : nullptr;
            if (fl) { rc = run_random_logged(e, cfg, in.n, in.d, nk, fl, &res, &best); fclose(fl); full_log = true; }
            else rc = nemgpu_run_random(e, 50, cfg.tie_seed, &res, &best);
            if (rc == NEMGPU_OK && best >= 0)
                lg.pr("Best start was %d (%s = %g)\n", best + 1, "M", (double)res.crit[3]);                  // nem_alg.c:1722-1725
        } else {
            lg.pr("Initializing parameters from given value\n");
            // dolog: the reference's per-iteration <Fname>.log (two criteria passes per iteration, one host round
            // trip per iteration); NEM_MI355X_LOG=0 keeps the pipelined run and writes a header-only log
            const char* lenv = getenv("NEM_MI355X_LOG");
            FILE* fl = (dolog && !(lenv && lenv[0] == '0')) ? fopen((base + ".log").c_str(), "w") : nullptr;
            if (fl) { rc = run_logged_pipelined(e, cfg, in.n, in.d, nk, fl, &res); fclose(fl); full_log = true; }
            else rc = nemgpu_run(e, &res);
        }
    }
    if (rc != NEMGPU_OK) {
        lg.pr("*** NEM GPU engine error : %s\n", nemgpu_last_error());
        fprintf(stderr, "nem (MI355X engine): %s\n", nemgpu_last_error());
        if (e) nemgpu_destroy(e);
        lg.close();
        return rc == NEMGPU_E_ARG ? EXIT_E_ARGS_ : (rc == NEMGPU_E_MEMORY ? EXIT_E_MEMORY_ : (rc == NEMGPU_E_INTERNAL ? EXIT_E_BUG_ : EXIT_E_SYSTEM_));
    }

    lg.pr("  Iterations : %4d \n", res.iters);
    if (res.zero_density_sites > 0) lg.pr("Warning : pt %d density = 0\n", res.first_zero_density_site);
    if (res.status == NEMGPU_W_EMPTYCLASS) lg.pr("Class %d empty at iteration %d\n", res.emptyk, res.iters);
    lg.pr("  criterion NEM = %6.3f / Ps-Like = %6.3f / Lmix = %6.3f\n", res.crit[2], res.crit[3], res.crit[4]);
    if (cfg.cvtest != NEMGPU_CV_NONE && res.status == NEMGPU_OK) {          // nem_alg.c:1863-1875
        if (res.converged) lg.pr("  NEM converged after %d iterations\n", res.iters);
        else lg.pr("  NEM did not converge after %d iterations\n", res.iters);
    }
    lg.pr("  [engine] EM loop %.6f s, %d sweep relaxation rounds\n", res.loop_seconds, res.sweep_rounds);
    const clk::time_point t_run = clk::now();

    int ret = EXIT_OK_;
    if (res.status == NEMGPU_OK) {
        lg.pr("Saving results ...\n");                                      // nem_exe.c:628-630
        std::vector<float> c((size_t)in.n * nk), prop(nk), center((size_t)nk * in.d), disp((size_t)nk * in.d);
        nemgpu_get_partition(e, c.data());
        nemgpu_get_params(e, prop.data(), center.data(), disp.data(), nullptr);
        int w1 = hard ? write_cf_file(outname, c.data(), in.n, nk, cfg.tie_rule, cfg.tie_seed, res.tie_draws)
                      : write_uf_file(outname, c.data(), in.n, nk);
        int w2 = write_mf_file(base + ".mf", res.crit, cfg.beta, in.d, nk, center.data(), prop.data(), disp.data());
        if (w1 != NEMGPU_OK) fprintf(stderr, "Could not open file '%s' in write mode\n", outname.c_str());
        if (w2 != NEMGPU_OK) fprintf(stderr, "Could not open file '%s.mf' in write mode\n", base.c_str());
        if (dolog && !full_log) {                                           // header-only <Fname>.log (never parsed by PPanGGOLiN)
            FILE* fl = fopen((base + ".log").c_str(), "w");
            if (fl) {
                fprintf(fl, "NEM log file  -  %s\n", date_line().c_str());
                fprintf(fl, "  MI355X engine: per-iteration criteria are not logged; iterations = %d\n", res.iters);
                fclose(fl);
            }
        }
        lg.pr("  [engine] phases: read %.4f s, engine + upload %.4f s, run%s %.4f s, results %.4f s\n", secs(t_start, t_read),
              secs(t_read, t_up), full_log ? " (logged)" : "", secs(t_up, t_run), secs(t_run, clk::now()));
        lg.pr("NEM completed, classification in %s\n", outname.c_str());    // :642-645
        lg.pr(" criteria and parameters in %s%s\n", base.c_str(), ".mf");
        if (dolog) lg.pr("Log of detailed running in %s.log\n", base.c_str());
    } else {
        lg.pr("*** NEM warning status : empty class\n");                    // :660-663
        ret = EXIT_W_RESULT_;
    }
    nemgpu_destroy(e);
    lg.close();
    return ret;
}

// ============================================================================================
// file layer C ABI (host only)
// ============================================================================================
struct nemio_inputs {
    NemInputs in;
    int k = 0;
};

extern "C" {

int nemio_read(const char* Fname, int nk, nemio_inputs** out)
{
    if (!Fname || !out || nk <= 0) return NEMGPU_E_FUNCARG;
    *out = nullptr;
    nemio_inputs* h = new nemio_inputs();
    h->k = nk;
    const std::string base = std::string(Fname).substr(0, 200);
    std::string err;
    int sts = read_str_file(base, h->in, err);
    if (sts == NEMGPU_OK) sts = read_dat_file(base, h->in, err);
    if (sts == NEMGPU_OK) sts = read_param_file(base, nk, h->in, err);
    if (sts == NEMGPU_OK) {
        if (h->in.type != 'N') sts = read_nei_file(base, h->in, err);
        else h->in.nei_ptr.assign(h->in.n + 1, 0);
    }
    if (sts != NEMGPU_OK) { set_error(err); delete h; return sts; }
    *out = h;
    return NEMGPU_OK;
}

void nemio_free(nemio_inputs* in) { delete in; }

int nemio_sizes(const nemio_inputs* h, int* n, int* d, int* nnz, int* max_neighs, int* param_mode, int* type)
{
    if (!h) return NEMGPU_E_FUNCARG;
    if (n) *n = h->in.n;
    if (d) *d = h->in.d;
    if (nnz) *nnz = (int)h->in.nei_idx.size();
    if (max_neighs) *max_neighs = h->in.max_neighs;
    if (param_mode) *param_mode = h->in.param_mode;
    if (type) *type = h->in.type;
    return NEMGPU_OK;
}

int nemio_copy(const nemio_inputs* h, uint32_t* xbits, int32_t* nei_ptr, int32_t* nei_idx, float* nei_w, float* prop,
               float* center, float* disp)
{
    if (!h) return NEMGPU_E_FUNCARG;
    const NemInputs& in = h->in;
    if (xbits) memcpy(xbits, in.xbits.data(), in.xbits.size() * sizeof(uint32_t));
    if (nei_ptr) memcpy(nei_ptr, in.nei_ptr.data(), in.nei_ptr.size() * sizeof(int32_t));
    if (nei_idx && !in.nei_idx.empty()) memcpy(nei_idx, in.nei_idx.data(), in.nei_idx.size() * sizeof(int32_t));
    if (nei_w && !in.nei_w.empty()) memcpy(nei_w, in.nei_w.data(), in.nei_w.size() * sizeof(float));
    if (prop) memcpy(prop, in.prop.data(), in.prop.size() * sizeof(float));
    if (center) memcpy(center, in.center.data(), in.center.size() * sizeof(float));
    if (disp) memcpy(disp, in.disp.data(), in.disp.size() * sizeof(float));
    return NEMGPU_OK;
}

int nemio_write_uf(const char* path, const float* c, int n, int k) { return write_uf_file(path, c, n, k); }

// test hook: the log writer's " %<width>.<dec>f" formatter; returns the length written (out: at least 64 bytes)
int nemio_format_fixed(float v, int width, int dec, char* out)
{
    if (!out || dec < 0 || dec > 3 || width < 0 || width > 24) return -1;
    std::string s;
    put_fixed(s, v, width, dec);
    memcpy(out, s.data(), s.size());
    out[s.size()] = '\0';
    return (int)s.size();
}
int nemio_write_cf(const char* path, const float* c, int n, int k, int tie_rule, uint32_t seed)
{
    return write_cf_file(path, c, n, k, tie_rule, seed, 0);
}
int nemio_write_mf(const char* path, const float crit6[6], float beta, int d, int k, const float* center,
                   const float* prop, const float* disp)
{
    return write_mf_file(path, crit6, beta, d, k, center, prop, disp);
}

}  // extern "C"
