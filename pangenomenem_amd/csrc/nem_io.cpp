// nem_io.cpp -- the reference's ASCII file formats, host side only.
//
// Readers accept what the reference's fscanf-based readers accept for the files PPanGGOLiN
// writes (ppanggolin/ppanggolin.py:821-930) and return the same StatusET codes; writers emit
// byte-identical text for identical values (same printf formats as SaveResults,
// /root/reference/ppanggolin/NEM/nem_exe.c:1596-1781).
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <immintrin.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

#include "nem_internal.hpp"
#include "nem_rng.hpp"

namespace nemk {

namespace {

bool slurp(const std::string& path, std::vector<char>& buf)
{
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    buf.resize((size_t)(sz > 0 ? sz : 0) + 1);
    size_t got = sz > 0 ? fread(buf.data(), 1, (size_t)sz, fp) : 0;
    fclose(fp);
    buf[got] = '\0';
    buf.resize(got + 1);
    return true;
}

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// Skip the opening comment lines the way ReadOpeningComments does (lib_io.c:32-87): leading
// lines that START with the marker are comments; their text (marker stripped) is collected.
const char* skip_opening_comments(const char* p, std::string& comment)
{
    comment.clear();
    while (*p == '#') {
        const char* eol = strchr(p, '\n');
        size_t len = eol ? (size_t)(eol - p + 1) : strlen(p);
        comment.append(p + 1, len - 1);
        p += len;
    }
    return p;
}

struct Tok {
    const char* p;
    explicit Tok(const char* s) : p(s) {}
    void ws() { while (*p && is_space(*p)) p++; }
    bool eof() { ws(); return *p == '\0'; }
    // fscanf("%d"): optional sign + digits; stops at the first non-digit
    bool next_int(long& v)
    {
        ws();
        if (!*p) return false;
        // up to 18 digits by hand (strtol's result for them), anything longer through strtol
        const char* q = p;
        const bool neg = (*q == '-');
        if (*q == '-' || *q == '+') q++;
        const char* d0 = q;
        long acc = 0;
        while (*q >= '0' && *q <= '9' && q - d0 < 18) acc = acc * 10 + (*q++ - '0');
        if (q != d0 && !(*q >= '0' && *q <= '9')) { v = neg ? -acc : acc; p = q; return true; }
        char* end = nullptr;
        errno = 0;
        v = strtol(p, &end, 10);
        if (end == p) return false;
        p = end;
        return true;
    }
    // fscanf("%f" / "%g").  Plain decimals "[sign]digits[.digits]" with at most 15 significant and 8 fractional digits
    // are converted by hand: digits / 10^k is one correctly rounded double operation on exact operands, and such a
    // short decimal is either exactly a float midpoint (then the double holds it exactly and the cast rounds to even,
    // like strtof) or at least 2^-24 / 10^8 > 2^-53 (relative) away from one, so the cast to float cannot land on the
    // wrong side.  Exponents, hex, inf/nan, longer numbers: strtof.
    bool next_float(float& v)
    {
        ws();
        if (!*p) return false;
        {
            const char* q = p;
            const bool neg = (*q == '-');
            if (*q == '-' || *q == '+') q++;
            unsigned long long digits = 0;
            int nd = 0, nf = 0;
            const char* d0 = q;
            while (*q >= '0' && *q <= '9' && nd < 15) { digits = digits * 10 + (unsigned)(*q++ - '0'); nd++; }
            bool ok = !(*q >= '0' && *q <= '9');
            if (ok && *q == '.') {
                q++;
                while (*q >= '0' && *q <= '9' && nd < 15 && nf < 8) { digits = digits * 10 + (unsigned)(*q++ - '0'); nd++; nf++; }
                ok = !(*q >= '0' && *q <= '9');
            }
            (void)d0;
            if (ok && nd > 0 && *q != 'e' && *q != 'E' && *q != 'x' && *q != 'X' && *q != 'p' && *q != 'P') {
                static const double p10[9] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8};
                const double dv = (double)digits / p10[nf];
                v = (float)(neg ? -dv : dv);
                p = q;
                return true;
            }
        }
        char* end = nullptr;
        v = strtof(p, &end);
        if (end == p) return false;
        p = end;
        return true;
    }
};

}  // namespace

// ReadStrFile, nem_exe.c:739-830
int read_str_file(const std::string& base, NemInputs& in, std::string& err)
{
    std::vector<char> buf;
    const std::string path = base + ".str";
    if (!slurp(path, buf)) { err = "File str " + path + " does not exist"; return NEMGPU_E_FILEIN; }
    const char* p = skip_opening_comments(buf.data(), in.str_comment);
    char type[100] = {0};
    long n1 = 0, n2 = 0, n3 = 0;
    int got = sscanf(p, "%99s %ld %ld %ld", type, &n1, &n2, &n3);
    if (got < 3) { err = "Structure file (" + path + ") not enough fields"; return NEMGPU_E_FILE; }
    for (char* s = type; *s; s++) if (*s > 'a' && *s < 'z') *s = (char)(*s + ('A' - 'a'));   // my_strupr, :1813-1824
    if (!strcmp(type, "N") || !strcmp(type, "S")) {
        in.type = type[0];
        in.n = (int)n1; in.d = (int)n2;
    } else if (!strcmp(type, "I")) {
        err = "Data type I (image) is not supported by this engine (PPanGGOLiN always writes S)";
        return NEMGPU_E_FILE;
    } else {
        err = std::string("Data type ") + type + " unknown in file " + path;
        return NEMGPU_E_FILE;
    }
    if (in.n <= 0 || in.d <= 0) { err = "Structure file (" + path + ") needs positive sizes"; return NEMGPU_E_FILE; }
    return NEMGPU_OK;
}

// ReadMatrixFile, nem_exe.c:834-898: N*D whitespace-separated numbers.  This engine handles
// presence/absence data: every value must be 0 or 1 (what ppanggolin.py:850 writes).
// One row of the text ppanggolin.py:850 writes -- one character per value, one tab between values, a newline at
// the end ("0\t1\t...\t1\n", 2*d bytes): 4 values per 8-byte load, validated and packed with integer arithmetic.
// Returns how many values (a multiple of 4, or d) were taken.
// `at`: index of the first value of p within the row (a multiple of 4), for rows whose head was packed elsewhere.
static inline int pack_regular_row(const char* p, int d, uint32_t* row, int at = 0)
{
    int j = 0;
    for (; j + 4 <= d; j += 4) {
        uint64_t w;
        memcpy(&w, p + 2 * (size_t)j, 8);
        const bool last = (j + 4 == d);
        const uint64_t seps = last ? 0x0A00090009000900ull : 0x0900090009000900ull;
        if ((w & 0xFFFEFFFEFFFEFFFEull) != (0x0030003000300030ull | seps)) break;
        const uint32_t nib = (uint32_t)(((w & 0x0001000100010001ull) * 0x0001000200040008ull) >> 48) & 0xFu;
        row[(at + j) >> 5] |= nib << ((at + j) & 31);   // (at + j) % 4 == 0: the nibble never straddles a word
    }
    return j;
}

// The same row with 32-byte loads where the CPU has AVX2 + BMI2: 16 values per load -- the value bytes' low bits
// moved to the sign positions, one movemask, the even bits squeezed together -- for all but the last 1..16 values
// of the row (whose final separator is the newline); those are left to pack_regular_row.  Returns the number of
// values taken (a multiple of 16), or -1 when the text is not of the regular form.
__attribute__((target("avx2,bmi2"))) static int pack_regular_row_avx2(const char* p, int d, uint32_t* row)
{
    const __m256i keep = _mm256_set1_epi16((short)0xFFFE);          // value byte without its low bit, separator whole
    const __m256i want = _mm256_set1_epi16((short)0x0930);          // '0' | '1' then a tab
    int j = 0;
    for (; j + 16 < d; j += 16) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + 2 * (size_t)j));
        if (_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_and_si256(v, keep), want)) != -1) return -1;
        const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(v, 7));   // bit 2i = value i
        row[j >> 5] |= _pext_u32(m, 0x55555555u) << (j & 31);       // j % 16 == 0: never straddles a word
    }
    return j;
}

static bool cpu_has_avx2_bmi2()
{
    static const bool yes = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    return yes;
}

// A file of exactly n rows of 2*d bytes (d a multiple of 4) is, if every row has the regular form, n rows at known
// offsets: a few threads pread() their share of rows through a small buffer each (the text never sits in memory as
// a whole) and pack it.  false = some row is not regular (nothing is reported; the general reader takes over).
static bool read_dat_regular(const std::string& path, int n, int d, int wf, uint32_t* xbits)
{
    if (n <= 0 || d <= 0 || (d & 3)) return false;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    const size_t rowb = 2 * (size_t)d;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size != rowb * (size_t)n) { close(fd); return false; }
    unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::max(1u, std::min({4u, hw ? hw : 1u, (unsigned)(((size_t)st.st_size >> 21) + 1)}));   // 2 MB and more per thread
    std::vector<char> ok((size_t)nt, 1);
    const bool wide = cpu_has_avx2_bmi2();
    auto work = [&](int t) {
        const int r0 = (int)((long long)n * t / nt), r1 = (int)((long long)n * (t + 1) / nt);
        const int rows_per = (int)std::max<size_t>(1, ((size_t)256 << 10) / rowb);              // ~256 KB per pread
        std::vector<char> buf((size_t)rows_per * rowb + 8);                                     // (+8: the 8-byte loads)
        for (int r = r0; r < r1; r += rows_per) {
            const int nr = std::min(rows_per, r1 - r);
            const size_t want = (size_t)nr * rowb;
            size_t got = 0;
            while (got < want) {
                const ssize_t g = pread(fd, buf.data() + got, want - got, (off_t)((size_t)r * rowb + got));
                if (g <= 0) { ok[(size_t)t] = 0; return; }
                got += (size_t)g;
            }
            for (int i = 0; i < nr; i++) {
                const char* text = buf.data() + (size_t)i * rowb;
                uint32_t* row = xbits + (size_t)(r + i) * wf;
                int j0 = 0;
                if (wide) { j0 = pack_regular_row_avx2(text, d, row); if (j0 < 0) { ok[(size_t)t] = 0; return; } }
                if (j0 + pack_regular_row(text + 2 * (size_t)j0, d - j0, row, j0) != d) { ok[(size_t)t] = 0; return; }
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work, t);
    work(0);
    for (std::thread& x : th) x.join();
    close(fd);
    for (char c : ok) if (!c) return false;
    return true;
}

int read_dat_file(const std::string& base, NemInputs& in, std::string& err)
{
    std::vector<char> buf;
    const std::string path = base + ".dat";
    {
        const int wf0 = (in.d + 31) / 32;
        in.xbits.assign((size_t)in.n * wf0, 0u);
        if (read_dat_regular(path, in.n, in.d, wf0, in.xbits.data())) return NEMGPU_OK;
    }
    if (!slurp(path, buf)) { err = "File matrix " + path + " does not exist"; return NEMGPU_E_FILEIN; }
    const int n = in.n, d = in.d, wf = (d + 31) / 32;
    in.xbits.assign((size_t)n * wf, 0u);
    const char* p = buf.data();
    const char* const bend = buf.data() + buf.size() - 1;        // the terminating NUL slurp() appended
    for (int i = 0; i < n; i++) {
        uint32_t* row = in.xbits.data() + (size_t)i * wf;
        int j = 0;
        // Row fast path (pack_regular_row); anything else in the row (spaces, "1.0", blank lines) leaves the rest
        // to the tokeniser below.
        if ((size_t)(bend - p) >= 2 * (size_t)d) {
            j = pack_regular_row(p, d, row);
            if (j == d) { p += 2 * (size_t)d; continue; }
            p += 2 * (size_t)j;                                  // values [0, j) are consumed (each "v\t")
        }
        for (; j < d; j++) {
            while (is_space(*p)) p++;
            if (!*p) {
                char msg[256];
                snprintf(msg, sizeof msg, "%s : short file (%d/%d lines and %d/%d columns)", path.c_str(),
                         j == 0 ? i - 1 : i, n, j == 0 ? d : j, d);
                err = msg;
                return NEMGPU_E_FILE;
            }
            // fast path: a lone '0' or '1'
            if ((p[0] == '0' || p[0] == '1') && (is_space(p[1]) || p[1] == '\0')) {
                if (p[0] == '1') row[j >> 5] |= 1u << (j & 31);
                p++;
                continue;
            }
            char* end = nullptr;
            float v = strtof(p, &end);
            if (end == p) { err = path + " : unreadable value"; return NEMGPU_E_FILE; }
            p = end;
            if (v == 1.0f) row[j >> 5] |= 1u << (j & 31);
            else if (v != 0.0f) {
                err = path + " : only 0/1 presence/absence values are supported by this engine";
                return NEMGPU_E_FILE;
            }
        }
    }
    return NEMGPU_OK;
}

// ReadNeiFile / ReadPtsNeighs for TYPE_SPATIAL, nem_exe.c:1278-1478.
// Quirks kept: neighbours outside [1,N] are dropped and zero weights are dropped, each list is
// compacted on its own (so a dropped index shifts the weights), NbNeigh = number of non-zero
// weights when weighted; a point listed twice keeps its last record.
int read_nei_file(const std::string& base, NemInputs& in, std::string& err)
{
    std::vector<char> buf;
    const std::string path = base + ".nei";
    if (!slurp(path, buf)) { err = "File Neigh " + path + " File does not exist"; return NEMGPU_E_FILEIN; }
    const char* p = skip_opening_comments(buf.data(), in.nei_comment);
    Tok tk(p);
    long weighted = 0;
    tk.next_int(weighted);
    const int n = in.n;
    // every record's lists go behind each other into two flat arrays; where[i] / nb[i] = the LAST record of point i
    std::vector<int32_t> flat_idx;
    std::vector<float> flat_w;
    flat_idx.reserve(buf.size() / 4);
    flat_w.reserve(buf.size() / 4);
    std::vector<long long> where((size_t)n, -1);
    std::vector<int> nb(n, 0);
    int nmax = 0;
    int line = 0;
    while (!tk.eof()) {
        long ipt = 0, nbv = 0;
        if (!tk.next_int(ipt)) { err = "Error in neighb. file : unreadable point index"; return NEMGPU_E_FILE; }
        if (!tk.next_int(nbv)) break;
        if (ipt < 1 || ipt > n || nbv < 0) { err = "Error in neighb. file : point index out of range"; return NEMGPU_E_FILE; }
        const size_t at = flat_idx.size();
        flat_idx.resize(at + (size_t)nbv, 0);           // calloc'ed NeighT[nbv] (genmemo.c:30)
        flat_w.resize(at + (size_t)nbv, 0.0f);
        int32_t* iv = flat_idx.data() + at;
        float* wv = flat_w.data() + at;
        int nv = 0;
        for (long t = 0; t < nbv && !tk.eof(); t++) {
            long j = 0;
            if (!tk.next_int(j)) {
                char msg[128]; snprintf(msg, sizeof msg, "Error in neighb. file l.%d : neighbor %ld", line, t);
                err = msg; return NEMGPU_E_FILE;
            }
            if (j >= 1 && j <= n) iv[nv++] = (int32_t)(j - 1);
        }
        if (weighted) {
            nv = 0;
            for (long t = 0; t < nbv && !tk.eof(); t++) {
                float w = 0.f;
                if (!tk.next_float(w)) {
                    char msg[128]; snprintf(msg, sizeof msg, "Error in neighb. file l.%d : weight %ld", line, t);
                    err = msg; return NEMGPU_E_FILE;
                }
                if (w != 0.0f) wv[nv++] = w;
            }
        } else {
            for (int t = 0; t < nv; t++) wv[t] = 1.0f;
        }
        flat_idx.resize(at + (size_t)nv); flat_w.resize(at + (size_t)nv);
        where[(size_t)ipt - 1] = (long long)at; nb[ipt - 1] = nv;
        if (nv > nmax) nmax = nv;
        line++;
    }
    in.max_neighs = nmax;
    in.nei_ptr.assign(n + 1, 0);
    for (int i = 0; i < n; i++) in.nei_ptr[i + 1] = in.nei_ptr[i] + nb[i];
    in.nei_idx.resize(in.nei_ptr[n]);
    in.nei_w.resize(in.nei_ptr[n]);
    for (int i = 0; i < n; i++) {
        if (nb[i] == 0) continue;
        std::copy(flat_idx.begin() + where[i], flat_idx.begin() + where[i] + nb[i], in.nei_idx.begin() + in.nei_ptr[i]);
        std::copy(flat_w.begin() + where[i], flat_w.begin() + where[i] + nb[i], in.nei_w.begin() + in.nei_ptr[i]);
    }
    return NEMGPU_OK;
}

// ReadParamFile, nem_exe.c:973-1091 (Bernoulli: dispersions taken as written)
int read_param_file(const std::string& base, int k, NemInputs& in, std::string& err)
{
    std::vector<char> buf;
    const std::string path = base + ".m";
    if (!slurp(path, buf)) { err = "File param " + path + " does not exist"; return NEMGPU_E_FILEIN; }
    const int d = in.d;
    const int need = 1 + (k - 1) + k * d + k * d;
    Tok tk(buf.data());
    auto missing = [&](int have) {
        char msg[256];
        snprintf(msg, sizeof msg, "The file %s needs at least %d values (%d missing)", path.c_str(), need, need - have);
        err = msg;
        return NEMGPU_E_FILEIN;
    };
    long flag = 0;
    if (!tk.next_int(flag)) return missing(0);
    if (flag == 1) in.param_mode = 1;
    else if (flag == 2) in.param_mode = 2;
    else {
        err = "First line of file " + path + " must be 1 (parameters at beginning) or 2 (fixed parameters throughout the clustering process)";
        return NEMGPU_E_FILEIN;
    }
    int have = 1, sts = NEMGPU_OK;
    in.prop.assign(k, 0.f); in.center.assign((size_t)k * d, 0.f); in.disp.assign((size_t)k * d, 0.f);
    float pK = 1;
    for (int c = 0; c < k - 1; c++) {
        float v;
        if (!tk.next_float(v)) return missing(have);
        have++;
        in.prop[c] = v;
        pK = pK - in.prop[c];
    }
    in.prop[k - 1] = pK;
    if (pK <= 0.0) { char msg[64]; snprintf(msg, sizeof msg, "Last class has pK = %5.2f <= 0", pK); err = msg; sts = NEMGPU_E_FILE; }
    for (size_t t = 0; t < (size_t)k * d; t++) {
        float v;
        if (!tk.next_float(v)) return missing(have);
        have++;
        in.center[t] = v;
    }
    for (size_t t = 0; t < (size_t)k * d; t++) {
        float v;
        if (!tk.next_float(v)) return missing(have);
        have++;
        in.disp[t] = v;
        if (v <= 0) {
            char msg[96];
            snprintf(msg, sizeof msg, "Dispersion(k=%d, d=%d) = %5.3f <= 0", (int)(t / d) + 1, (int)(t % d) + 1, v);
            err = msg; sts = NEMGPU_E_FILE;
        }
    }
    if (!tk.eof()) {
        char msg[256];
        snprintf(msg, sizeof msg, "The file %s needs do not need more than %d values", path.c_str(), need);
        err = msg; sts = NEMGPU_E_FILEIN;
    }
    return sts;
}

// SaveResults, fuzzy branch: nem_exe.c:1673-1687
// " %5.3f " of one membership, as printf prints it.  Memberships live in [0, 1]: v * 1000 is exact in a double (24 + 10
// bits), so rint() -- round to nearest, ties to even, printf's rule in the default rounding mode -- of that product
// is the three-decimal rounding of the exact binary value; anything else (negative, >= 10, NaN) goes to snprintf.
static inline char* put_membership(char* o, float v)
{
    if (v >= 0.0f && v < 9.9994f && !std::signbit(v)) {    // (-0.0 prints as -0.000)
        const int r = (int)rint((double)v * 1000.0);     // 0 .. 9999
        o[0] = ' ';
        o[1] = (char)('0' + r / 1000);
        o[2] = '.';
        o[3] = (char)('0' + (r / 100) % 10);
        o[4] = (char)('0' + (r / 10) % 10);
        o[5] = (char)('0' + r % 10);
        o[6] = ' ';
        return o + 7;
    }
    return o + snprintf(o, 64, " %5.3f ", v);
}

int write_uf_file(const std::string& path, const float* c, int n, int k)
{
    FILE* fp = fopen(path.c_str(), "w");
    if (!fp) return NEMGPU_E_FILEOUT;
    const size_t row_max = (size_t)64 * k + 2;
    std::vector<char> buf(std::max(row_max, (size_t)1 << 20));
    char* o = buf.data();
    for (int i = 0; i < n; i++) {
        if ((size_t)(buf.data() + buf.size() - o) < row_max) { fwrite(buf.data(), 1, (size_t)(o - buf.data()), fp); o = buf.data(); }
        for (int kk = 0; kk < k; kk++) o = put_membership(o, c[(size_t)i * k + kk]);
        *o++ = '\n';
    }
    fwrite(buf.data(), 1, (size_t)(o - buf.data()), fp);
    fclose(fp);
    return NEMGPU_OK;
}

// SaveResults, hard branch: nem_exe.c:1634-1670 (MAP label + 1 per point, one line)
// TIE_LIBC: the ties go on drawing from the run's stream, `draws_before` draws into random() after srandom(seed)
int write_cf_file(const std::string& path, const float* c, int n, int k, int tie_rule, uint32_t seed, long draws_before)
{
    FILE* fp = fopen(path.c_str(), "w");
    if (!fp) return NEMGPU_E_FILEOUT;
    GlibcRandom rng(seed);
    bool rng_ready = false;
    for (int i = 0; i < n; i++) {
        const float* row = c + (size_t)i * k;
        int kmax = 0; float u = row[0];
        for (int kk = 1; kk < k; kk++) if (row[kk] > u) { u = row[kk]; kmax = kk; }
        if (tie_rule == NEMGPU_TIE_HASH) {
            int eq[kMaxK]; int ne = 0; eq[0] = kmax;
            for (int kk = kmax + 1; kk < k; kk++) if (row[kk] == u) eq[++ne] = kk;
            if (ne > 0) {
                kmax = eq[mix32_host(seed, 0xFFFFFFFFu, (uint32_t)i) % (uint32_t)(ne + 1)];
            }
        } else if (tie_rule == NEMGPU_TIE_LIBC) {
            int eq[kMaxK]; int ne = 0; eq[0] = kmax;
            for (int kk = kmax + 1; kk < k; kk++) if (row[kk] == u) eq[++ne] = kk;
            if (ne > 0) {
                if (!rng_ready) { for (long t = 0; t < draws_before; t++) (void)rng.next(); rng_ready = true; }
                kmax = eq[rng.integer(0, ne)];                       // RandomInteger(0, nequal), nem_alg.c:635
            }
        }
        fprintf(fp, "%d ", kmax + 1);
    }
    fputc('\n', fp);
    fclose(fp);
    return NEMGPU_OK;
}

// SaveResults, .mf: nem_exe.c:1708-1774 (Bernoulli family)
int write_mf_file(const std::string& path, const float crit[6], float beta, int d, int k, const float* center,
                  const float* prop, const float* disp)
{
    FILE* fp = fopen(path.c_str(), "w");
    if (!fp) return NEMGPU_E_FILEOUT;
    const float errorrate = nanf("");                    // CalcError with Kr == 0, nem_alg.c:2790-2792
    fprintf(fp, "Criteria U=NEM, D=Hathaway, L=mixture, M=markov ps-like, error\n\n");
    fprintf(fp, "  %g    %g    %g    %g   %g\n\n", crit[2], crit[0], crit[4], crit[3], errorrate);
    fprintf(fp, "Beta (%s)\n", "fixed");
    fprintf(fp, "  %6.4f\n", beta);
    fprintf(fp, "Mu (%d), Pk, and disp (%d) of the %d classes\n\n", d, d, k);
    // (a value equal to the one before it -- centres are 0, 0.5 or 1, a class's dispersions are often all the same --
    //  reuses that one's text instead of going through printf again)
    char txt[64];
    int len = 0;
    uint32_t last = 0;
    bool have = false;
    auto put = [&](const char* fmt, float v) {
        uint32_t bits;
        memcpy(&bits, &v, 4);
        if (!have || bits != last) { len = snprintf(txt, sizeof txt, fmt, v); last = bits; have = true; }
        fwrite(txt, 1, (size_t)len, fp);
    };
    for (int kk = 0; kk < k; kk++) {
        have = false;
        for (int j = 0; j < d; j++) put(" %10.3g ", center[(size_t)kk * d + j]);
        fprintf(fp, "  %5.3g  ", prop[kk]);
        have = false;
        for (int j = 0; j < d; j++) put(" %10g ", disp[(size_t)kk * d + j]);
        fputc('\n', fp);
    }
    fclose(fp);
    return NEMGPU_OK;
}

}  // namespace nemk
