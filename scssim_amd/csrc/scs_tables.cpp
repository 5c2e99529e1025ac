// scs_tables.cpp -- .profile loader, table builder and FASTA staging (host).
// Reference behaviour followed: lib/profile/Profile.cpp:930-1234 (load), 832-863 + 897-927
// (normParas(true)), 1363-1430 (initCDFs), lib/matrix/Matrix.h:483-522 (normalize, cumsum),
// lib/mydefine/MyDefine.cpp:54-57 (normpdf), 274-282 (randIndx), 337-349 (getNextLine);
// lib/fastahack/Fasta.cpp:45-215 + lib/genome/Genome.cpp:176-195,272-278 (FASTA).
#include "scs_tables.h"
#include "scs_seams.h"
#include "scs_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace scs {

static const double kZeroFinal = 2.2204e-16;      // MyDefine.cpp:20 / Matrix.h:86

// ---------------------------------------------------------------- exact thresholds
template <class Pred>   // pred(x) monotone: true for x < T, false from T on; returns T clamped to 2^32-1
static uint32_t count_true(Pred pred) {
    uint64_t lo = 0, hi = 1ull << 32;           // invariant: pred true below lo, false from hi on
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (pred((uint32_t)mid)) lo = mid + 1; else hi = mid;
    }
    return lo > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)lo;
}
uint32_t threshold_lt(double c) { return count_true([c](uint32_t x) { return (x / 4294967296.0) < c; }); }
uint32_t threshold_le(double c) { return count_true([c](uint32_t x) { return (x / 4294967296.0) <= c; }); }
uint32_t threshold_cdf(double c) {
    return count_true([c](uint32_t x) { double r = kZeroFinal + (1 - kZeroFinal) * (x / 4294967296.0); return r <= c; });
}

std::vector<uint64_t> binom_table(double ber, int n_min, int n_max) {
    std::vector<uint64_t> T((size_t)(n_max - n_min + 1) * BINOM_KMAX);
    const double q = 1 - ber, ratio = ber / q;
    for (int n = n_min; n <= n_max; ++n) {
        double pmf = 1, b = q;                                   // q^n by repeated squaring
        for (unsigned e = (unsigned)n; e; e >>= 1) { if (e & 1) pmf = pmf * b; b = b * b; }
        double cdf = 0;
        uint64_t* row = &T[(size_t)(n - n_min) * BINOM_KMAX];
        for (int k = 0; k < BINOM_KMAX; ++k) {
            cdf += pmf;
            row[k] = cdf >= 1.0 ? ~0ull : (uint64_t)(cdf * 18446744073709551616.0);
            pmf = pmf * (double)(n - k) / (double)(k + 1) * ratio;
        }
    }
    return T;
}

// ---------------------------------------------------------------- small text helpers
static std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && strchr(" \t\r\n", s[a])) ++a;
    while (b > a && strchr(" \t\r\n", s[b - 1])) --b;
    return s.substr(a, b - a);
}
static void split_on(const std::string& s, char d, std::vector<std::string>& out) {
    out.clear();
    size_t p = 0;
    if (s.empty()) return;
    for (;;) {
        size_t q = s.find(d, p);
        if (q == std::string::npos) { out.push_back(s.substr(p)); break; }
        out.push_back(s.substr(p, q - p));
        p = q + 1;
        if (p == s.size()) break;               // std::getline semantics: no trailing empty field
    }
}

namespace {
struct ProfileReader {
    std::ifstream ifs; std::string path; int line_no = 0;
    explicit ProfileReader(const std::string& p) : ifs(p), path(p) {
        if (!ifs.is_open()) throw std::runtime_error("can not open file " + p);
    }
    bool next(std::string& line) {               // skips blank lines and '#' comments
        while (std::getline(ifs, line)) { ++line_no; if (!line.empty() && line[0] != '#') return true; }
        line.clear(); return false;
    }
    std::string need() { std::string l; if (!next(l)) bad("unexpected end of file"); return l; }
    [[noreturn]] void bad(const std::string& why) {
        throw std::runtime_error("Error: malformed model file " + path + " @line " + std::to_string(line_no) + ": " + why);
    }
    void row(std::vector<double>& dst, size_t expect) {
        std::string l = need(); std::vector<std::string> f; split_on(l, '\t', f);
        if (expect && f.size() != expect) bad("wrong number of columns");
        dst.resize(f.size());
        for (size_t i = 0; i < f.size(); ++i) dst[i] = atof(strip(f[i]).c_str());
    }
};
}  // namespace

static void normalize_rows(double* m, size_t rows, size_t cols) {
    for (size_t r = 0; r < rows; ++r) {
        double* p = m + r * cols; double sum = 0;
        for (size_t c = 0; c < cols; ++c) sum += p[c];
        const double den = kZeroFinal + sum;
        for (size_t c = 0; c < cols; ++c) p[c] /= den;
    }
}
static void cumsum_rows(double* m, size_t rows, size_t cols) {
    for (size_t r = 0; r < rows; ++r) { double* p = m + r * cols; for (size_t c = 1; c < cols; ++c) p[c] = p[c] + p[c - 1]; }
}
static int kmer_row(const std::string& k) {       // order of Profile::initKmers (Profile.cpp:69-123)
    auto code = [](char c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; case 'X': return 5; } return -1; };
    if (k.size() != 3) return -1;
    int a = code(k[0]), b = code(k[1]), c = code(k[2]);
    if (c < 0 || c > 3) return -1;
    if (a == 5 && b == 5) return c;
    if (a == 5 && b >= 0 && b < 4) return 4 + b * 4 + c;
    if (a >= 0 && a < 4 && b >= 0 && b < 4) return 20 + a * 16 + b * 4 + c;
    return -1;
}

void load_profile(const std::string& path, bool paired, int isize, ProfileTables& T) {
    ProfileReader rd(path);
    std::string bases, line; int bin_count = -1, kmer = -1, read_len = -1;
    std::vector<std::string> f;
    while (rd.next(line)) {                        // header: four "key: value" lines in any order
        split_on(line, ':', f);
        if (f.size() != 2) rd.bad(line);
        const std::string k = strip(f[0]), v = strip(f[1]);
        if (k == "bases") bases = v; else if (k == "binCount") bin_count = atoi(v.c_str());
        else if (k == "kmer") kmer = atoi(v.c_str()); else if (k == "readLength") read_len = atoi(v.c_str());
        else rd.bad(line);
        if (!bases.empty() && bin_count > 0 && kmer > 0 && read_len > 0) break;
    }
    if (bases.empty() || bin_count <= 0 || kmer <= 0 || read_len <= 0) throw std::runtime_error("Error: malformed model file " + path);
    if (bases != "ACGT" || kmer != 3) throw std::runtime_error("Error: only bases ACGT / kmer 3 profiles are supported: " + path);
    // Profile::init sets bins := readLength (Profile.cpp:183) while load() reads binCount rows; the
    // reference is only well-defined when both agree (all shipped profiles do).
    if (bin_count != read_len) throw std::runtime_error("Error: profile binCount must equal readLength: " + path);
    const size_t B = (size_t)read_len;
    T.read_length = read_len; T.bins = read_len;
    T.subs1.assign(NKMER * B * 4, 0.0); T.subs2.assign(NKMER * B * 4, 0.0); T.qual.assign(16 * B * 94, 0.0);
    std::vector<double> ins_f(1, 0.0), del_f(1, 0.0), tmp;
    int sections = 0;
    while (rd.next(line)) {
        if (line == "[Insert Rate]") { T.insert_rate = atof(strip(rd.need()).c_str()); ++sections; }
        else if (line == "[Insert Frequency]") { rd.row(ins_f, 0); ++sections; }
        else if (line == "[Deletion Rate]") { T.del_rate = atof(strip(rd.need()).c_str()); ++sections; }
        else if (line == "[Deletion Frequency]") { rd.row(del_f, 0); ++sections; }
        else if (line == "[Substitution Probs]") {
            for (int i = 0; i < NKMER; ++i) {
                split_on(rd.need(), ':', f);
                if (f.size() != 2 || strip(f[0]) != "kmer") rd.bad("expected kmer header");
                const int row = kmer_row(strip(f[1]));
                if (row < 0) rd.bad("unrecognized kmer");
                for (size_t j = 0; j < 2 * B; ++j) {
                    rd.row(tmp, 4);
                    double* dst = (j < B ? T.subs1.data() : T.subs2.data()) + ((size_t)row * B + (j < B ? j : j - B)) * 4;
                    memcpy(dst, tmp.data(), 4 * sizeof(double));
                }
            }
            ++sections;
        }
        else if (line == "[Base Quality Distribution]") {
            for (int i = 0; i < 16; ++i) {
                split_on(rd.need(), ':', f);
                if (f.size() != 2 || strip(f[0]) != "basePairIndx") rd.bad("expected basePairIndx header");
                const int bp = atoi(strip(f[1]).c_str());
                if (bp < 0 || bp > 15) rd.bad("unrecognized basePairIndx");
                for (size_t j = 0; j < B; ++j) { rd.row(tmp, 94); memcpy(T.qual.data() + ((size_t)bp * B + j) * 94, tmp.data(), 94 * sizeof(double)); }
            }
            ++sections;
        }
        else if (line == "[Insert Size Standard Deviation]") { T.std_isize = atof(strip(rd.need()).c_str()); ++sections; }
        else if (line == "[Log Ratio Mean Value]") {
            for (int j = 0; j < 101; ++j) {
                split_on(rd.need(), '\t', f);
                if (f.size() != 2) rd.bad("expected gc<TAB>mean");
                const int gc = atoi(f[0].c_str());
                if (gc < 0 || gc > 100) rd.bad("gc out of range");
                T.gc_means[gc] = atof(f[1].c_str());
            }
            ++sections;
        }
        else if (line == "[Log Ratio Standard Deviation]") { T.gc_std = atof(strip(rd.need()).c_str()); ++sections; }
    }
    if (sections < 9) throw std::runtime_error("Error: corrupted model file " + path + ", failed to load some parameters!");

    // normParas(true): row-normalise; all-zero rows get probability 1 on the k-mer's own last base
    for (int i = 0; i < NKMER; ++i) {
        const int own = i < 4 ? i : (i - 4) % 4;     // (i-20)%4 == (i-4)%4
        for (std::vector<double>* M : {&T.subs1, &T.subs2}) {
            double* m = M->data() + (size_t)i * B * 4;
            normalize_rows(m, B, 4);
            for (size_t j = 0; j < B; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += m[j * 4 + k]; if (s < kZeroFinal) m[j * 4 + own] = 1; }
        }
    }
    normalize_rows(T.qual.data(), 16 * B, 94);       // normParas ...
    normalize_rows(T.qual.data(), 16 * B, 94);       // ... and again in initCDFs (Profile.cpp:1393)
    cumsum_rows(T.qual.data(), 16 * B, 94);
    cumsum_rows(T.subs1.data(), NKMER * B, 4);
    T.have_cdf2 = paired && T.std_isize > 0;
    if (T.have_cdf2) cumsum_rows(T.subs2.data(), NKMER * B, 4); else T.subs2.clear();
    T.ins_cdf = ins_f; cumsum_rows(T.ins_cdf.data(), 1, T.ins_cdf.size());
    T.del_cdf = del_f; cumsum_rows(T.del_cdf.data(), 1, T.del_cdf.size());
    T.isize_cdf.clear(); T.isize_min = 0;
    if (paired && T.std_isize > 0) {                 // insert-size alphabet + discretised normal (Profile.cpp:908-926)
        const int mean = isize + 1, interval = (int)(6 * T.std_isize);
        const int lo = std::max(mean - interval / 2, read_len), hi = 2 * mean - lo;
        const double PI = 3.1415926;
        for (int x = lo; x <= hi; ++x)
            T.isize_cdf.push_back(exp(-pow((double)x - mean, 2) / (2 * pow(T.std_isize, 2))) / (sqrt(2 * PI) * T.std_isize));
        if (T.isize_cdf.empty()) throw std::runtime_error("Error: empty insert size range (isize too small for this profile)");
        normalize_rows(T.isize_cdf.data(), 1, T.isize_cdf.size());
        cumsum_rows(T.isize_cdf.data(), 1, T.isize_cdf.size());
        T.isize_min = lo;
    }
    auto to_thr = [](const std::vector<double>& c, std::vector<uint32_t>& t) { t.resize(c.size()); for (size_t i = 0; i < c.size(); ++i) t[i] = threshold_cdf(c[i]); };
    to_thr(T.subs1, T.subs1_t); to_thr(T.subs2, T.subs2_t); to_thr(T.qual, T.qual_t);
    to_thr(T.ins_cdf, T.ins_t); to_thr(T.del_cdf, T.del_t); to_thr(T.isize_cdf, T.isize_t);
    {   // alias rows of the quality tables ([REMAP], scs_tables.h)
        std::vector<uint8_t> sym((size_t)16 * B * 94); std::vector<uint64_t> w((size_t)16 * B * 94); std::vector<int> ns((size_t)16 * B); int max_syms = 1;
        for (size_t row = 0; row < (size_t)16 * B; ++row) { ns[row] = quality_row_weights(&T.qual[row * 94], &sym[row * 94], &w[row * 94]); max_syms = std::max(max_syms, ns[row]); }
        T.qual_k = max_syms <= 16 ? 16 : max_syms <= 64 ? 64 : 128;
        if (const char* f = seam_env("SCS_TEST_QK")) T.qual_k = std::max(T.qual_k, atoi(f) >= 128 ? 128 : atoi(f) >= 64 ? 64 : 16);   // tests: more columns than needed (no shipped model needs the 128-column kernels; the oracle reads the same variable)
        const size_t RW = (size_t)T.qual_k + T.qual_k / 4;
        T.qual_alias.assign((size_t)16 * B * RW, 0u);
        for (size_t row = 0; row < (size_t)16 * B; ++row) quality_alias_row(&sym[row * 94], &w[row * 94], ns[row], T.qual_k, &T.qual_alias[row * RW]);
    }
    T.t_insert = threshold_le(T.insert_rate);
    T.t_delete = threshold_lt(T.del_rate / (1 - T.insert_rate));
    T.t_indel = T.t_insert + (uint32_t)((((1ull << 32) - T.t_insert) * (uint64_t)T.t_delete) >> 32);
    T.gap_t = indel_gap_table(T.t_indel, T.read_length); T.t_kind = indel_kind_threshold(T.t_insert, T.t_indel);
}

// ---------------------------------------------------------------- FASTA
// Staging for whole-genome inputs: the file is mmap'ed and cut at line ends with memchr; sequence
// lines are appended as raw ASCII (bulk memcpy), the base-code conversion happens on the GPU after the upload.
static std::string index_name(const std::string& header) {      // Fasta.cpp:56-68: first token, "chrom"/"chr" prefix dropped
    std::string nm = header;
    size_t e = nm.find_first_of(" \t");
    if (e != std::string::npos) nm.resize(e);
    size_t i = nm.find("chrom");
    if (i != std::string::npos) return nm.substr(i + 5);
    i = nm.find("chr");
    if (i != std::string::npos) return nm.substr(i + 3);
    return nm;
}
template <class Pred>   // as count_true, without the clamp: 0 .. 2^32
static uint64_t count_true64(Pred pred) {
    uint64_t lo = 0, hi = 1ull << 32;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (pred((uint32_t)mid)) lo = mid + 1; else hi = mid; }
    return lo;
}
int quality_row_weights(const double* cdf, uint8_t sym[94], uint64_t w[94]) {
    int n = 0; uint64_t prev = 0;
    for (int k = 0; k < 94; ++k) {
        const double c = cdf[k];
        uint64_t cnt = k == 93 ? (1ull << 32)                                       // randIndx falls through to the last symbol (MyDefine.cpp:281)
                               : count_true64([c](uint32_t x) { const double r = kZeroFinal + (1 - kZeroFinal) * (x / 4294967296.0); return r <= c; });
        if (cnt < prev) cnt = prev;
        if (cnt > prev) { sym[n] = (uint8_t)k; w[n] = cnt - prev; ++n; prev = cnt; }
    }
    return n;
}
void quality_alias_row(const uint8_t* sym, const uint64_t* w, int n, int K, uint32_t* row) {
    const int abits = K == 16 ? 4 : K == 64 ? 6 : 7;
    const uint64_t C = (1ull << 32) / (uint64_t)K;                                 // draws per column
    std::vector<uint64_t> m((size_t)K, 0); std::vector<uint32_t> t((size_t)K, 0), al((size_t)K, 0);
    for (int j = 0; j < n; ++j) m[(size_t)j] = w[j];
    std::vector<int> small, large;
    for (int j = 0; j < K; ++j) { al[(size_t)j] = (uint32_t)j; (m[(size_t)j] < C ? small : large).push_back(j); }
    while (!small.empty() && !large.empty()) {                                     // Vose, in integers: the sums stay exact
        const int sj = small.back(); small.pop_back(); const int lj = large.back(); large.pop_back();
        t[(size_t)sj] = (uint32_t)m[(size_t)sj]; al[(size_t)sj] = (uint32_t)lj;
        m[(size_t)lj] -= C - m[(size_t)sj];
        (m[(size_t)lj] < C ? small : large).push_back(lj);
    }
    // what is left holds exactly a column's worth: "always my own symbol" = threshold 0 with myself as the alias
    for (int j = 0; j < K; ++j) row[j] = (t[(size_t)j] << abits) | al[(size_t)j];
    uint8_t* sb = reinterpret_cast<uint8_t*>(row + K);
    for (int j = 0; j < K; ++j) sb[j] = j < n ? sym[j] : (uint8_t)0;
}
std::vector<uint32_t> indel_gap_table(uint32_t t_indel, int read_length) {
    std::vector<uint32_t> t((size_t)read_length + 1, 0xFFFFFFFFu);
    const double q = 1.0 - (double)t_indel / 4294967296.0; double pw = 1.0;
    for (int g = 1; g <= read_length; ++g) { pw = pw * q; const double v = std::floor(pw * 4294967296.0); t[(size_t)g] = v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v; }
    return t;
}
uint32_t indel_kind_threshold(uint32_t t_insert, uint32_t t_indel) {
    if (t_indel == 0) return 0;
    const double v = std::floor(4294967296.0 * ((double)t_insert / (double)t_indel));
    return v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v;
}
void encode_record(const char* name, const char* seq, uint64_t len, FastaRecord& out) {
    out.name = index_name(name);
    out.code.assign((const uint8_t*)seq, (const uint8_t*)seq + len);
}
}  // namespace scs
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
namespace scs {
std::string fasta_index_name(const std::string& header_text) { return index_name(header_text); }
// a path as ONE word of a /bin/sh command line: single-quoted, embedded single quotes closed, escaped and reopened
static std::string sh_quote(const std::string& p) {
    std::string q = "'";
    for (char ch : p) { if (ch == '\'') q += "'\\''"; else q += ch; }
    return q + "'";
}
// "<name>.gz" is inflated beside itself with `gzip -cd`, as the reference does (Genome.cpp:183-187: the same command through
// system(), there with the path unquoted -- a name with a blank or a shell character is a word of its own here)
std::string fasta_plain_path(const std::string& path_in) {
    std::string path = path_in;
    if (path.empty()) throw std::runtime_error("reference sequence file not specified!");
    if (path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0) {
        const std::string plain = path.substr(0, path.size() - 3);
        const std::string cmd = "gzip -cd " + sh_quote(path) + " > " + sh_quote(plain);
        if (system(cmd.c_str()) != 0) throw std::runtime_error("could not inflate " + path);
        path = plain;
    }
    return path;
}
// fastahack's index entry (Fasta.cpp:241-249): first word of the name, length, offset of the first sequence byte, bases per
// line, bytes per line -- the last three from the record's first sequence line
void fasta_write_fai(const std::string& path, const char* base, size_t size, const std::vector<uint64_t>& hdr_off, const std::vector<uint64_t>& lens) {
    const std::string fname = path + ".fai"; struct stat fst;
    if (stat(fname.c_str(), &fst) == 0) return;
    fprintf(stderr, "index file %s not found, generating...\n", fname.c_str());
    // (written under a name of its own and renamed: the ranks of a sharded job stage side by side, and one that looks for the index
    // while another writes it must find a whole index or none)
    const std::string tmp = fname + ".tmp." + std::to_string((long)getpid());
    FILE* fai = fopen(tmp.c_str(), "w");
    if (!fai) { fprintf(stderr, "could not open index file %s for writing! (continuing without it)\n", fname.c_str()); return; }
    for (size_t r = 0; r < hdr_off.size(); ++r) {
        const size_t hdr = (size_t)hdr_off[r], end = r + 1 < hdr_off.size() ? (size_t)hdr_off[r + 1] : size;
        const char* nl = (const char*)memchr(base + hdr, '\n', end - hdr);
        const size_t heol = nl ? (size_t)(nl - base) : end; size_t hend = heol; if (hend > hdr && base[hend - 1] == '\r') --hend;
        size_t fl = heol + 1 < end ? heol + 1 : end;
        while (fl < end && base[fl] == ';') { const char* e = (const char*)memchr(base + fl, '\n', end - fl); fl = e ? (size_t)(e - base) + 1 : end; }
        const char* e = fl < end ? (const char*)memchr(base + fl, '\n', end - fl) : nullptr;
        const size_t line_len = fl < end ? (e ? (size_t)(e - base) + 1 - fl : end - fl) : 0;
        size_t line_blen = e ? line_len - 1 : line_len; if (line_blen && base[fl + line_blen - 1] == '\r') --line_blen;
        const std::string full(base + hdr + 1, base + hend);
        const std::string first = full.substr(0, full.find(' '));
        fprintf(fai, "%s\t%llu\t%zu\t%zu\t%zu\n", first.c_str(), (unsigned long long)lens[r], fl, line_blen, line_len);
    }
    if (fclose(fai) != 0 || rename(tmp.c_str(), fname.c_str()) != 0) { (void)unlink(tmp.c_str()); fprintf(stderr, "could not write index file %s! (continuing without it)\n", fname.c_str()); }
}
void load_fasta(const std::string& path_in, std::vector<FastaRecord>& out, bool make_index) {
    const std::string path = fasta_plain_path(path_in);                          // ".gz": inflated first (Genome.cpp:183-187)
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("could not open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("could not stat " + path); }
    const size_t size = (size_t)st.st_size;
    out.clear();
    if (size == 0) { close(fd); throw std::runtime_error("ERROR: reference sequence cannot be empty!"); }
    const char* base = (const char*)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (base == MAP_FAILED) { close(fd); throw std::runtime_error("could not map " + path); }
    (void)madvise((void*)base, size, MADV_SEQUENTIAL);
    // pass 1: record boundaries (headers) so each record's buffer is reserved once
    struct Span { size_t hdr, body, end; };
    std::vector<Span> spans;
    for (size_t pos = 0; pos < size;) {
        const char* nl = (const char*)memchr(base + pos, '\n', size - pos);
        const size_t eol = nl ? (size_t)(nl - base) : size;
        if (base[pos] == '>') { if (!spans.empty()) spans.back().end = pos; spans.push_back(Span{pos, eol + 1 < size ? eol + 1 : size, size}); }
        else if (spans.empty() && base[pos] != ';' && eol > pos && base[pos] != '\r') { munmap((void*)base, size); close(fd); throw std::runtime_error("malformed FASTA (sequence before header): " + path); }
        if (base[pos] == '>') {
            // jump: headers are rare; look for the next one directly
            const char* nx = (const char*)memmem(base + eol, size - eol, "\n>", 2);
            pos = nx ? (size_t)(nx - base) + 1 : size;
            continue;
        }
        pos = eol + 1;
    }
    out.resize(spans.size());
    // the reference indexes the FASTA through fastahack and leaves <file>.fai beside it when there is none
    // (lib/fastahack/Fasta.cpp:241-249; entry = first word of the name, length, offset, bases per line, bytes per line)
    FILE* fai = nullptr; const std::string fai_name = path + ".fai", fai_tmp = fai_name + ".tmp." + std::to_string((long)getpid());   // (renamed when whole: fasta_write_fai)
    if (make_index) {
        struct stat fst;
        if (stat(fai_name.c_str(), &fst) != 0) {
            fprintf(stderr, "index file %s not found, generating...\n", fai_name.c_str());
            if (!(fai = fopen(fai_tmp.c_str(), "w"))) fprintf(stderr, "could not open index file %s for writing! (continuing without it)\n", fai_name.c_str());
        }
    }
    for (size_t r = 0; r < spans.size(); ++r) {
        const Span& sp = spans[r];
        const char* nl = (const char*)memchr(base + sp.hdr, '\n', sp.end - sp.hdr);
        size_t heol = nl ? (size_t)(nl - base) : sp.end;
        size_t hend = heol; if (hend > sp.hdr && base[hend - 1] == '\r') --hend;
        out[r].name = index_name(std::string(base + sp.hdr + 1, base + hend));
        std::vector<uint8_t>& dst = out[r].code;
        dst.resize(sp.end > sp.body ? sp.end - sp.body : 0);
        size_t w = 0;
        for (size_t pos = sp.body; pos < sp.end;) {
            const char* e = (const char*)memchr(base + pos, '\n', sp.end - pos);
            size_t eol = e ? (size_t)(e - base) : sp.end, le = eol;
            if (le > pos && base[le - 1] == '\r') --le;
            if (base[pos] != ';' && le > pos) { memcpy(dst.data() + w, base + pos, le - pos); w += le - pos; }
            pos = eol + 1;
        }
        dst.resize(w);
        if (fai) {
            size_t fl = sp.body; while (fl < sp.end && base[fl] == ';') { const char* e = (const char*)memchr(base + fl, '\n', sp.end - fl); fl = e ? (size_t)(e - base) + 1 : sp.end; }
            const char* e = fl < sp.end ? (const char*)memchr(base + fl, '\n', sp.end - fl) : nullptr;
            const size_t line_len = fl < sp.end ? (e ? (size_t)(e - base) + 1 - fl : sp.end - fl) : 0;
            size_t line_blen = e ? line_len - 1 : line_len; if (line_blen && base[fl + line_blen - 1] == '\r') --line_blen;
            const std::string full(base + sp.hdr + 1, base + hend);
            const std::string first = full.substr(0, full.find(' '));
            fprintf(fai, "%s\t%zu\t%zu\t%zu\t%zu\n", first.c_str(), w, fl, line_blen, line_len);
        }
    }
    if (fai && (fclose(fai) != 0 || rename(fai_tmp.c_str(), fai_name.c_str()) != 0)) { (void)unlink(fai_tmp.c_str()); fprintf(stderr, "could not write index file %s! (continuing without it)\n", fai_name.c_str()); }
    munmap((void*)base, size); close(fd);
    if (out.empty()) throw std::runtime_error("ERROR: reference sequence cannot be empty!");
}

}  // namespace scs
