// ORACLE -- TEST INFRASTRUCTURE ONLY (see scs_oracle.h for the contract).
//
// CPU restatement of the SCSsim `genreads` hot path.  Every function cites the
// reference file:line it follows (paths relative to the reference root,
// qasimyu/scssim).  Two random back-ends:
//   ref     : the reference's own streams, consumed in the reference's order
//             (lib/threadpool/ThreadPool.cpp:41-47,203-212; glibc rand();
//             libstdc++ normal_distribution over minstd_rand0) -> byte-identical
//             FASTQ to the compiled reference at -t 1 under oracle/seedshim.cpp.
//   counter : Philox4x32-10 keyed by logical ids (DESIGN.md "RNG remapping").
// Apart from where a draw comes from, the two modes differ only in the places marked [REMAP] below -- eight, DESIGN.md
// section 4 numbers them: (2) amplification errors as a binomial count + distinct positions, (3) the Poisson sampler's
// software logarithm / product form, (4) the GC factor's normal sampler, (5) fixed-SHAPE sums and scans of the read
// allocation, (6) attach tries by geometric skip + a uniform draw over the feasible pairs, (7) a read's two xoshiro
// streams with one two-word step per output position, (8) indel events by geometric gaps, (9) quality symbols by the
// alias method.  Each is order-free / libm-free, so that any thread, shard or GPU schedule gives the same bytes, and is
// pinned statistically against the reference streams by tests/test_oracle_stats.py.  The primer stock is NOT among
// them: counter mode hands a primer type out exactly as the reference's live decrement does at -t 1 (amplify_pass).
// --rng ref is never touched by such changes: it reproduces the compiled reference's FASTQ byte for byte
// (tests/test_oracle_golden.py).
#include "scs_oracle.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace {

std::string g_err;
double g_timings[6];

[[noreturn]] void fail(const std::string& m) { throw std::runtime_error(m); }

const double ZERO_FINAL = 2.2204e-16;   // lib/mydefine/MyDefine.cpp:20, lib/matrix/Matrix.h:86

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11).  Published algorithm.
// ---------------------------------------------------------------------------
inline void philox(const uint32_t c_in[4], const uint32_t k_in[2], uint32_t out[4]) {
    uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3];
    uint32_t k0 = k_in[0], k1 = k_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---------------------------------------------------------------------------
// xoshiro128++ (Blackman & Vigna, public domain): the per-read sequential streams of the counter mode.
// ---------------------------------------------------------------------------
struct Xoshiro {
    uint32_t s[4];
    static inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    void seed(const uint32_t w[4]) { for (int i = 0; i < 4; ++i) s[i] = w[i]; if (!(s[0] | s[1] | s[2] | s[3])) s[0] = 1; }
    inline uint32_t next() {
        const uint32_t result = rotl(s[0] + s[3], 7) + s[0];
        const uint32_t t = s[1] << 9;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 11);
        return result;
    }
    // [REMAP] one step, two words: the xoshiro128++ output and the same scrambler on the other two state words (stream B of a read)
    inline void next2(uint32_t& a, uint32_t& b) {
        a = rotl(s[0] + s[3], 7) + s[0]; b = rotl(s[1] + s[2], 7) + s[1];
        const uint32_t t = s[1] << 9;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t; s[3] = rotl(s[3], 11);
    }
};

// ---------------------------------------------------------------------------
// Deterministic log: classic argument reduction x = 2^k (1+f), s = f/(2+f),
// degree-14 minimax in s (the fdlibm e_log scheme and coefficients).  Uses only
// IEEE +,-,*,/ so that x86 and gfx950 give the same bits (no contraction).
// ---------------------------------------------------------------------------
double det_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x < 0) return std::nan("");
    if (x == 0) return -HUGE_VAL;
    uint64_t b; memcpy(&b, &x, 8);
    int k = 0;
    if ((b >> 52) == 0) { x *= 18014398509481984.0; memcpy(&b, &x, 8); k = -54; }   // subnormal: * 2^54
    if ((b >> 52) == 0x7ff) return x;
    k += (int)(b >> 52) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m; memcpy(&m, &b, 8);
    if (m >= 1.4142135623730951) { m = m * 0.5; k += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s, w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    double dk = (double)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// exp(x) for -256 <= x <= 0 (fdlibm's argument reduction and polynomial; IEEE + - * / only): the GPU computes the same bits
double det_exp(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x >= 0) return 1.0;
    if (x < -256.0) return 0.0;
    const int k = (int)(invln2 * x - 0.5);                                         // x < 0: nearest integer
    const double dk = (double)k;
    const double hi = x - dk * ln2_hi, lo = dk * ln2_lo, r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    uint64_t b; memcpy(&b, &y, 8);
    b += (uint64_t)(int64_t)k << 52;                                               // * 2^k: k >= -370, y in [0.5, 2): no underflow
    double out; memcpy(&out, &b, 8);
    return out;
}

// ---------------------------------------------------------------------------
// Random source.  Every draw site names its counter-mode key; in ref mode the
// key is ignored and the reference's sequential streams are advanced instead.
// ---------------------------------------------------------------------------
enum Stage : uint32_t {
    ST_FRAGSPLIT = 1, ST_POISSON = 2, ST_ATTACH = 3, ST_ERR = 4, ST_ERRALT = 5, ST_WEIGHT = 6,
    ST_ALLOC_TOP = 7, ST_ALLOC_CHUNK = 8, ST_PAIR = 9, ST_READ = 10, ST_INDEL_INS = 11, ST_INDEL_LEN = 12, ST_INDEL = 13
};
// ST_READ blocks 0 and 1 of a read seed its two xoshiro128++ streams (indel tests / base-pass draws).

struct Key { uint32_t idx; uint64_t uid; uint32_t stage_word; int word; };
inline Key mk(uint32_t stage, uint32_t aux, uint64_t uid, uint32_t idx, int word) {
    return Key{idx, uid, stage | (aux << 8), word};
}

struct RefStreams {           // one worker thread (-t 1) + the main thread
    std::mt19937 w_real, w_int;     // ThreadPool.cpp:41-47: two copies of one seeded generator
    std::mt19937 m_real, m_int;     // main thread: map::operator[] default-constructs (seed 5489)
};

struct Rng {
    bool counter = false;
    uint32_t key[2] = {0, 0};
    RefStreams* ref = nullptr;

    inline uint32_t word(const Key& k) const {
        uint32_t c[4] = {k.idx, (uint32_t)k.uid, (uint32_t)(k.uid >> 32), k.stage_word}, o[4];
        philox(c, key, o);
        return o[k.word];
    }
    // ThreadPool::randomDouble / randomInteger both map x -> x / 2^32  (ThreadPool.cpp:203-212)
    inline double real(const Key& k) { return (counter ? word(k) : (uint32_t)ref->w_real()) / 4294967296.0; }
    inline double integer(const Key& k) { return (counter ? word(k) : (uint32_t)ref->w_int()) / 4294967296.0; }
    inline double main_real(const Key& k) { return (counter ? word(k) : (uint32_t)ref->m_real()) / 4294967296.0; }
    // glibc: rand()/(RAND_MAX+1.0)  (MyDefine.cpp:285-292)
    inline double grand(const Key& k) { return counter ? word(k) / 4294967296.0 : rand() / (RAND_MAX + 1.0); }
};

// randIndx(double* cdf, ac): MyDefine.cpp:274-282
inline unsigned rand_indx(const double* cdf, unsigned ac, double u) {
    double r = ZERO_FINAL + (1 - ZERO_FINAL) * u;
    for (unsigned k = 0; k < ac; ++k) if (r <= cdf[k]) return k;
    return ac - 1;
}

// ---------------------------------------------------------------------------
// Bases.  codes 0..3 = ACGT (config "bases", lib/config/Config.cpp:24), 4 = N / any other
// ---------------------------------------------------------------------------
const char BASES[] = "ACGTN";
inline uint8_t code_of(char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
                 case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
inline uint8_t comp_code(uint8_t c) { return c < 4 ? (uint8_t)(3 - c) : (uint8_t)4; }   // MyDefine.cpp:352-367
inline bool is_gc(uint8_t c) { return c == 1 || c == 2; }

// ---------------------------------------------------------------------------
// FASTA (lib/fastahack/Fasta.cpp:45-215,304-334; lib/genome/Genome.cpp:176-195,272-278)
// Records in file order; index key = first token of the header with a leading
// "chrom"/"chr" (first occurrence) stripped (Fasta.cpp:59-68).  Sequence upper-cased.
// ---------------------------------------------------------------------------
struct Record { std::string name; std::vector<uint8_t> code; };

std::vector<Record> load_fasta(const std::string& path) {
    std::ifstream ifs(path);
    if (!ifs.is_open()) fail("could not open " + path);
    std::vector<Record> recs;
    std::string line;
    while (std::getline(ifs, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == ';') continue;
        if (line[0] == '>') {
            std::string nm = line.substr(1);
            size_t e = nm.find_first_of(" \t");
            if (e != std::string::npos) nm = nm.substr(0, e);
            size_t i = nm.find("chrom");
            if (i == std::string::npos) { i = nm.find("chr"); if (i != std::string::npos) nm = nm.substr(i + 3); }
            else nm = nm.substr(i + 5);
            recs.push_back(Record{nm, {}});
        } else {
            if (recs.empty()) fail("FASTA sequence before header in " + path);
            auto& v = recs.back().code;
            for (char c : line) v.push_back(code_of(c));
        }
    }
    if (recs.empty()) fail("ERROR: reference sequence cannot be empty!");
    return recs;
}

// ---------------------------------------------------------------------------
// Profile (lib/profile/Profile.cpp:69-123 kmers, 171-214 init, 930-1234 load,
// 832-863/897-927 normParas(true), 1363-1430 initCDFs; Matrix.h:328-336,483-522)
// ---------------------------------------------------------------------------
// TEST-ONLY: SCSO_TEST_BIAS=<branch>[:<factor>] makes ONE counter-mode remap wrong by 2 % (or by the factor given): errors | attach | indel | alias | gc, so that
// tests/test_oracle_stats.py can show that its statistics would catch such a remap (test_a_biased_remap_is_caught).  Read at every
// use, never set by anything but that test.
static double test_bias(const char* branch) {                                       // SCSO_TEST_BIAS=<branch>[:<factor>], factor 1.02 by default
    const char* e = getenv("SCSO_TEST_BIAS"); const size_t n = strlen(branch);
    if (!e || strncmp(e, branch, n) != 0 || (e[n] != 0 && e[n] != ':')) return 1.0;
    return e[n] == ':' ? atof(e + n + 1) : 1.02;
}

struct Profile {
    int L = 0, bins = 0, kmer = 3, N = 4, kmerCount = 84, nq = 94;
    double insertRate = 0, delRate = 0, stdISize = 0, gcStd = 0;
    // [REMAP] counter mode decides insertion / deletion / neither at a base from ONE 32-bit draw x:
    //   x < tIns (#{x : x/2^32 <= insertRate}) -> insertion; else x < tIndel -> deletion, where the deletion test's own
    //   threshold tDel = #{x : x/2^32 < delRate/(1-insertRate)} is rescaled to the draws left: tIndel = tIns + ((2^32-tIns)*tDel >> 32)
    uint32_t tIns = 0, tIndel = 0;
    // [REMAP] counter mode walks a read from indel event to indel event: the per-base tests are i.i.d. with probability
    // p = tIndel / 2^32, so the number of event-free bases before the next event is geometric: gap >= g <=> x < tGap[g],
    // tGap[g] = floor((1-p)^g 2^32); the event is an insertion when a second draw y < tKind = floor(2^32 tIns / tIndel)
    std::vector<uint32_t> tGap; uint32_t tKind = 0;
    std::vector<double> insCdf, delCdf;
    std::vector<double> subs1, subs2;     // [84][bins][4]  (dist, then cdf in place)
    bool haveCdf2 = false;
    std::vector<double> qual;             // [16][bins][94]
    // [REMAP] counter mode draws a quality symbol by the ALIAS method: of the 2^32 draws, symbol k of a row is hit by
    // w_k = #{x : r(x) <= cdf[k]} - #{x : r(x) <= cdf[k-1]} in the reference's comparison (MyDefine.cpp:274-282; the last
    // symbol takes the rest), and the alias row -- qualK columns (16, 64 or 128), column j = x >> (32 - log2 K) owning
    // 2^32 / K draws of which the lowest t_j go to the row's j-th drawable symbol and the others to column alias_j's --
    // hits it with exactly w_k draws as well (integer Vose construction).  Row = K words (t_j << log2 K | alias_j) + K bytes.
    int qualK = 16; std::vector<uint32_t> qualAlias;
    double biasAlias = 1.0;               // test_bias("alias") when the model was loaded (1.0 outside the bias test: the top 2 % of a column's draws go to the next column's symbol)
    std::vector<int> isizeAlphabet;
    std::vector<double> isizeCdf;
    double gcMeans[101];
    std::map<std::string, int> kmerIndex;
};

std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}
std::vector<std::string> split(const std::string& s, char d) {       // lib/split/split.cpp:3-16
    std::vector<std::string> out; std::stringstream ss(s); std::string it;
    while (std::getline(ss, it, d)) out.push_back(it);
    return out;
}
bool next_line(std::ifstream& ifs, std::string& line) {             // MyDefine.cpp:337-349
    line.clear();
    while (std::getline(ifs, line)) { if (!line.empty() && line[0] != '#') break; }
    return !line.empty();
}

void row_normalize(double* m, int rows, int cols) {                   // Matrix::normalize(0)
    for (int i = 0; i < rows; ++i) {
        double s = 0; for (int j = 0; j < cols; ++j) s += m[i * cols + j];
        for (int j = 0; j < cols; ++j) m[i * cols + j] /= (ZERO_FINAL + s);
    }
}
void row_cumsum(double* m, int rows, int cols) {                      // Matrix::cumsum
    for (int i = 0; i < rows; ++i)
        for (int j = 1; j < cols; ++j) m[i * cols + j] = m[i * cols + j] + m[i * cols + j - 1];
}
double normpdf(double x, double mu, double sigma) {                   // MyDefine.cpp:54-57
    double PI = 3.1415926;
    return exp(-pow(x - mu, 2) / (2 * pow(sigma, 2))) / (sqrt(2 * PI) * sigma);
}

Profile* load_profile(const std::string& path, bool paired, int isize) {
    std::ifstream ifs(path);
    if (!ifs.is_open()) fail("can not open file " + path);
    auto P = std::make_unique<Profile>();
    std::string line, bases;
    int binCount = -1, kmer = -1, readLength = -1;
    while (next_line(ifs, line)) {                                    // Profile.cpp:946-990
        auto f = split(line, ':');
        if (f.size() != 2) fail("malformed model file header: " + line);
        std::string k = trim(f[0]), v = trim(f[1]);
        if (k == "bases") bases = v; else if (k == "binCount") binCount = atoi(v.c_str());
        else if (k == "kmer") kmer = atoi(v.c_str()); else if (k == "readLength") readLength = atoi(v.c_str());
        else fail("malformed model file header: " + line);
        if (!bases.empty() && binCount > 0 && kmer > 0 && readLength > 0) break;
    }
    if (bases != "ACGT" || kmer != 3 || binCount <= 0 || readLength <= 0) fail("unsupported profile header in " + path);
    if (binCount != readLength) fail("profile needs binCount == readLength (Profile.cpp:183 sets bins := readLength)");
    P->L = readLength; P->bins = readLength;
    const int B = P->bins;
    // k-mer order (Profile.cpp:69-123): 4 x "XXb", 16 x "Xab", 64 x "abc", each ACGT-lexicographic
    {
        int k = 0; const char* b = "ACGT";
        for (int c = 0; c < 4; ++c) P->kmerIndex[std::string("XX") + b[c]] = k++;
        for (int a = 0; a < 4; ++a) for (int c = 0; c < 4; ++c) P->kmerIndex[std::string("X") + b[a] + b[c]] = k++;
        for (int a = 0; a < 4; ++a) for (int c2 = 0; c2 < 4; ++c2) for (int c = 0; c < 4; ++c)
            P->kmerIndex[std::string() + b[a] + b[c2] + b[c]] = k++;
    }
    P->subs1.assign((size_t)84 * B * 4, 0.0); P->subs2.assign((size_t)84 * B * 4, 0.0);
    P->qual.assign((size_t)16 * B * 94, 0.0);
    std::vector<double> insF(1, 0.0), delF(1, 0.0);
    int loaded = 0;
    while (next_line(ifs, line)) {
        if (line == "[Insert Rate]") { if (!next_line(ifs, line)) fail("malformed profile"); P->insertRate = atof(trim(line).c_str()); loaded++; }
        else if (line == "[Insert Frequency]") {
            if (!next_line(ifs, line)) fail("malformed profile");
            auto f = split(line, '\t'); insF.clear(); for (auto& s : f) insF.push_back(atof(trim(s).c_str())); loaded++;
        }
        else if (line == "[Deletion Rate]") { if (!next_line(ifs, line)) fail("malformed profile"); P->delRate = atof(trim(line).c_str()); loaded++; }
        else if (line == "[Deletion Frequency]") {
            if (!next_line(ifs, line)) fail("malformed profile");
            auto f = split(line, '\t'); delF.clear(); for (auto& s : f) delF.push_back(atof(trim(s).c_str())); loaded++;
        }
        else if (line == "[Substitution Probs]") {
            for (int i = 0; i < 84; ++i) {
                if (!next_line(ifs, line)) fail("malformed profile");
                auto f = split(line, ':');
                if (f.size() != 2 || trim(f[0]) != "kmer") fail("malformed profile: " + line);
                auto it = P->kmerIndex.find(trim(f[1]));
                if (it == P->kmerIndex.end()) fail("unrecognized kmer " + line);
                int ki = it->second;
                for (int j = 0; j < 2 * B; ++j) {
                    if (!next_line(ifs, line)) fail("malformed profile");
                    auto g = split(line, '\t');
                    if (g.size() != 4) fail("malformed profile: " + line);
                    for (int k = 0; k < 4; ++k) {
                        double p = atof(trim(g[k]).c_str());
                        if (j < B) P->subs1[((size_t)ki * B + j) * 4 + k] = p;
                        else P->subs2[((size_t)ki * B + (j - B)) * 4 + k] = p;
                    }
                }
            }
            loaded++;
        }
        else if (line == "[Base Quality Distribution]") {
            for (int i = 0; i < 16; ++i) {
                if (!next_line(ifs, line)) fail("malformed profile");
                auto f = split(line, ':');
                if (f.size() != 2 || trim(f[0]) != "basePairIndx") fail("malformed profile: " + line);
                int bp = atoi(trim(f[1]).c_str());
                if (bp < 0 || bp > 15) fail("unrecognized basePairIndx");
                for (int j = 0; j < B; ++j) {
                    if (!next_line(ifs, line)) fail("malformed profile");
                    auto g = split(line, '\t');
                    if (g.size() != 94) fail("malformed profile (quality row)");
                    for (int k = 0; k < 94; ++k) P->qual[((size_t)bp * B + j) * 94 + k] = atof(trim(g[k]).c_str());
                }
            }
            loaded++;
        }
        else if (line == "[Insert Size Standard Deviation]") { if (!next_line(ifs, line)) fail("malformed profile"); P->stdISize = atof(trim(line).c_str()); loaded++; }
        else if (line == "[Log Ratio Mean Value]") {
            for (int j = 0; j < 101; ++j) {
                if (!next_line(ifs, line)) fail("malformed profile");
                auto g = split(line, '\t'); if (g.size() != 2) fail("malformed profile: " + line);
                int gc = atoi(g[0].c_str()); if (gc < 0 || gc > 100) fail("bad gc row");
                P->gcMeans[gc] = atof(g[1].c_str());
            }
            loaded++;
        }
        else if (line == "[Log Ratio Standard Deviation]") { if (!next_line(ifs, line)) fail("malformed profile"); P->gcStd = atof(trim(line).c_str()); loaded++; }
    }
    if (loaded < 9) fail("Error: corrupted model file " + path + ", failed to load some parameters!");

    // ---- normParas(true): Profile.cpp:832-863 -------------------------------------------------
    for (int i = 0; i < 84; ++i) {
        int last = i < 4 ? i : (i < 20 ? (i - 4) % 4 : (i - 20) % 4);          // index of kmers[i][kmer-1]
        for (auto* M : {&P->subs1, &P->subs2}) {
            double* m = M->data() + (size_t)i * B * 4;
            row_normalize(m, B, 4);
            for (int j = 0; j < B; ++j) {                                     // Profile.cpp:844-856 (tmp.get(0,j) reads row j's sum)
                double s = 0; for (int k = 0; k < 4; ++k) s += m[j * 4 + k];
                if (s < ZERO_FINAL) m[j * 4 + last] = 1;
            }
        }
    }
    for (int i = 0; i < 16; ++i) row_normalize(P->qual.data() + (size_t)i * B * 94, B, 94);   // Profile.cpp:860-862
    if (paired && P->stdISize > 0) {                                           // Profile.cpp:908-926
        int mean = isize + 1;
        int interval = (int)(6 * P->stdISize);
        int lo = std::max(mean - interval / 2, readLength);
        int hi = 2 * mean - lo;
        std::vector<double> d;
        for (int x = lo; x <= hi; ++x) { P->isizeAlphabet.push_back(x); d.push_back(normpdf(x, mean, P->stdISize)); }
        if (d.empty()) fail("empty insert size range");
        row_normalize(d.data(), 1, (int)d.size());
        row_cumsum(d.data(), 1, (int)d.size());                                // initCDFs Profile.cpp:1399-1402
        P->isizeCdf = d;
    }
    // ---- initCDFs: Profile.cpp:1363-1430 -------------------------------------------------------
    P->insCdf = insF; row_cumsum(P->insCdf.data(), 1, (int)P->insCdf.size());
    P->delCdf = delF; row_cumsum(P->delCdf.data(), 1, (int)P->delCdf.size());
    for (int i = 0; i < 16; ++i) {                                             // quality normalised a second time (1393)
        double* m = P->qual.data() + (size_t)i * B * 94;
        row_normalize(m, B, 94); row_cumsum(m, B, 94);
    }
    for (int i = 0; i < 84; ++i) row_cumsum(P->subs1.data() + (size_t)i * B * 4, B, 4);
    P->haveCdf2 = paired && P->stdISize > 0;                                   // Profile.cpp:1416-1428
    if (P->haveCdf2) for (int i = 0; i < 84; ++i) row_cumsum(P->subs2.data() + (size_t)i * B * 4, B, 4);
    {   // alias rows of the quality tables (counter mode)
        const int Bq = P->bins;
        auto count64 = [](double c) { uint64_t lo = 0, hi = 1ull << 32; while (lo < hi) { const uint64_t mid = (lo + hi) >> 1;
            const double r = ZERO_FINAL + (1 - ZERO_FINAL) * ((uint32_t)mid / 4294967296.0); if (r <= c) lo = mid + 1; else hi = mid; } return lo; };
        std::vector<std::vector<uint8_t>> sym((size_t)16 * Bq); std::vector<std::vector<uint64_t>> w((size_t)16 * Bq); size_t maxs = 1;
        for (size_t row = 0; row < (size_t)16 * Bq; ++row) {
            uint64_t prev = 0;
            for (int k = 0; k < 94; ++k) {
                uint64_t cnt = k == 93 ? (1ull << 32) : count64(P->qual[row * 94 + k]);
                if (cnt < prev) cnt = prev;
                if (cnt > prev) { sym[row].push_back((uint8_t)k); w[row].push_back(cnt - prev); prev = cnt; }
            }
            maxs = std::max(maxs, sym[row].size());
        }
        int K = maxs <= 16 ? 16 : maxs <= 64 ? 64 : 128;
        if (const char* f = getenv("SCS_TEST_QK")) K = std::max(K, atoi(f) >= 128 ? 128 : atoi(f) >= 64 ? 64 : 16);   // tests: more columns than needed (the 128-column kernels; the product reads the same variable)
        const int abits = K == 16 ? 4 : K == 64 ? 6 : 7;
        const uint64_t C = (1ull << 32) / (uint64_t)K; const size_t RW = (size_t)K + K / 4;
        P->qualK = K; P->qualAlias.assign((size_t)16 * Bq * RW, 0u); P->biasAlias = test_bias("alias");
        for (size_t row = 0; row < (size_t)16 * Bq; ++row) {
            std::vector<uint64_t> m((size_t)K, 0); std::vector<uint32_t> t((size_t)K, 0), al((size_t)K, 0); std::vector<int> small, large;
            for (size_t j = 0; j < w[row].size(); ++j) m[j] = w[row][j];
            for (int j = 0; j < K; ++j) { al[(size_t)j] = (uint32_t)j; (m[(size_t)j] < C ? small : large).push_back(j); }
            while (!small.empty() && !large.empty()) {
                const int sj = small.back(); small.pop_back(); const int lj = large.back(); large.pop_back();
                t[(size_t)sj] = (uint32_t)m[(size_t)sj]; al[(size_t)sj] = (uint32_t)lj;
                m[(size_t)lj] -= C - m[(size_t)sj];
                (m[(size_t)lj] < C ? small : large).push_back(lj);
            }
            uint32_t* r = &P->qualAlias[row * RW];
            for (int j = 0; j < K; ++j) r[j] = (t[(size_t)j] << abits) | al[(size_t)j];
            uint8_t* sb = reinterpret_cast<uint8_t*>(r + K);
            for (int j = 0; j < K; ++j) sb[j] = (size_t)j < sym[row].size() ? sym[row][(size_t)j] : (uint8_t)0;
        }
    }
    {   // integer thresholds of the two indel tests (monotone in the 32-bit draw: found by bisection on the reference's
        // own double comparisons), combined for the one-draw counter mode
        auto count_true = [](auto pred) { uint64_t lo = 0, hi = 1ull << 32; while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (pred((uint32_t)mid)) lo = mid + 1; else hi = mid; } return (uint32_t)std::min<uint64_t>(lo, 0xFFFFFFFFull); };
        const double ir = P->insertRate, dr = P->delRate / (1 - P->insertRate);
        const uint32_t tIns = count_true([ir](uint32_t x) { return (x / 4294967296.0) <= ir; });
        const uint32_t tDel = count_true([dr](uint32_t x) { return (x / 4294967296.0) < dr; });
        P->tIns = tIns; P->tIndel = tIns + (uint32_t)((((1ull << 32) - tIns) * (uint64_t)tDel) >> 32);
        P->tGap.assign((size_t)P->L + 1, 0xFFFFFFFFu);
        { const double q = 1.0 - (double)P->tIndel / 4294967296.0 * test_bias("indel"); double pw = 1.0;
          for (int g = 1; g <= P->L; ++g) { pw = pw * q; const double v = std::floor(pw * 4294967296.0); P->tGap[(size_t)g] = v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v; } }
        if (P->tIndel) { const double v = std::floor(4294967296.0 * ((double)P->tIns / (double)P->tIndel)); P->tKind = v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v; }
    }
    return P.release();
}

// k-mer index of the 3 context codes (5 = 'X'); -1 when not in the table (Profile.cpp:216-222)
inline int kmer_index(uint8_t a, uint8_t b, uint8_t c) {
    if (c > 3) return -1;
    if (a == 5 && b == 5) return c;
    if (a == 5 && b < 4) return 4 + b * 4 + c;
    if (a < 4 && b < 4) return 20 + a * 16 + b * 4 + c;
    return -1;
}

// ---------------------------------------------------------------------------
// predict(): Profile.cpp:1582-1697 (+1515-1580).  window = n codes; outputs n' bases/quals.
// ---------------------------------------------------------------------------
int predict(const Profile& P, Rng& rng, const uint8_t* win, int n, bool isRead1,
            uint64_t uid, uint32_t attempt, char* out_b, char* out_q) {
    const uint32_t rd = isRead1 ? 0u : 1u;
    const uint32_t aux = rd | (attempt << 1);
    // [REMAP] counter mode: each read owns two xoshiro128++ streams seeded by Philox blocks 0 and 1 of ST_READ --
    // A feeds the indel tests, B the base pass.  B advances in STEPS of two words (Xoshiro::next2): every output position
    // takes one step -- its first word is the substitution draw (unused where the k-mer has no row), its second the
    // quality draw (quality symbol, or the quality of an 'N') -- and an inserted base takes a step of its own, ahead of its
    // position's (first word = the base).  Indel lengths (rare) stay keyed Philox draws.
    Xoshiro xa, xb;
    if (rng.counter) {
        uint32_t c[4] = {0, (uint32_t)uid, (uint32_t)(uid >> 32), ST_READ | (aux << 8)}, o[4];
        philox(c, rng.key, o); xa.seed(o);
        c[0] = 1; philox(c, rng.key, o); xb.seed(o);
    }
    auto drawA = [&]() { return rng.counter ? xa.next() / 4294967296.0 : rng.real(Key{}); };
    auto drawB = [&]() { return rng.counter ? xb.next() / 4294967296.0 : rng.real(Key{}); };
    auto drawBi = [&]() { return rng.counter ? xb.next() / 4294967296.0 : rng.integer(Key{}); };
    std::vector<int> indelLens; indelLens.reserve(n);
    std::vector<std::vector<uint8_t>> ins(n);
    int indelLength = 0;
    auto gap = [&](uint32_t x, uint32_t rem) -> uint32_t {                       // [REMAP] event-free bases before the next event among `rem`
        const std::vector<uint32_t>& T = P.tGap;
        if (x < T[rem]) return rem;
        uint32_t lo = 1, hi = rem;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (x >= T[mid]) hi = mid; else lo = mid + 1; }
        return lo - 1;
    };
    if (rng.counter) {
        // [REMAP] stream A = gap, kind, gap, kind, ...: the walk jumps from event to event (same distribution as one test per base)
        indelLens.assign(n, 0);
        if (P.tIndel) for (int j = 0; j < n;) {
            j += (int)gap(xa.next(), (uint32_t)(n - j));
            if (j >= n) break;
            const uint32_t y = xa.next();
            const double u = rng.real(mk(ST_INDEL_LEN, aux, uid, j, 0));
            if (y < P.tKind) {
                int k = (int)rand_indx(P.insCdf.data(), P.insCdf.size(), u);
                for (int t = 0; t < k; ++t) ins[j].push_back(0);                 // the inserted base is drawn from stream B when it is emitted (base pass below)
                indelLength += k; indelLens[j] = k; j++;
            } else {
                int k = (int)rand_indx(P.delCdf.data(), P.delCdf.size(), u);
                if (k > 0) { k = std::min(n - j, k); indelLength -= k; indelLens[j] = k; j += k; }
                else j++;
            }
        }
    } else
    for (int j = 0; j < n;) {                                                  // 1606-1622
        int k = 0; bool isIns = false;
        double p = drawA();                                                     // getIndelSeq 1552-1570
        if (p <= P.insertRate) {
            k = rand_indx(P.insCdf.data(), P.insCdf.size(), rng.real(mk(ST_INDEL_LEN, aux, uid, j, 0)));
            for (int t = 0; t < k; ++t) {
                double u = rng.integer(Key{});
                ins[j].push_back((uint8_t)(long)(0 + (P.N - 1 - 0) * u));       // randomInteger(0, N-1): never 'T'
            }
            isIns = !ins[j].empty();
        } else {
            p = drawA();
            if (p < P.delRate / (1 - P.insertRate))
                k = rand_indx(P.delCdf.data(), P.delCdf.size(), rng.real(mk(ST_INDEL_LEN, aux, uid, j, 0)));
        }
        if (!isIns && k > 0) {                                                  // deletion
            k = std::min(n - j, k);
            indelLength -= k;
            indelLens.push_back(k);
            for (int i = 1; i < k; ++i) indelLens.push_back(0);
            j += k;
        } else {
            indelLength += k; j++; indelLens.push_back(k);
        }
    }
    if (n + indelLength < 50) {                                                 // 1623-1630
        indelLength = 0;
        for (auto& v : ins) v.clear();
        indelLens.assign(n, 0);
    }
    std::vector<uint8_t> src, inserted; src.reserve(n + indelLength + 2); inserted.reserve(n + indelLength + 2);
    for (int j = 0; j < n;) {                                                   // 1632-1654
        if (ins[j].empty() && indelLens[j] > 0) { j += indelLens[j]; continue; }
        src.push_back(win[j]); inserted.push_back(0);
        for (uint8_t b : ins[j]) { src.push_back(b); inserted.push_back(1); }
        j++;
    }
    const int m = n + indelLength;
    if ((int)src.size() != m) fail("predict: length bookkeeping mismatch");
    const int B = P.bins;
    const std::vector<double>& subs = (isRead1 || !P.haveCdf2) ? P.subs1 : P.subs2;   // 1523-1550
    for (int j = 0; j < m; ++j) {                                               // 1666-1694
        uint32_t xs = 0, xq = 0;
        if (rng.counter) {
            if (inserted[j]) { xb.next2(xs, xq); src[j] = (uint8_t)(long)(0 + (P.N - 1 - 0) * (xs / 4294967296.0)); }   // [REMAP] a step of its own, before the position's
            xb.next2(xs, xq);
        }
        uint8_t c0 = j >= 2 ? src[j - 2] : 5, c1 = j >= 1 ? src[j - 1] : 5, c2 = src[j];
        int refIndx = c2 < 4 ? c2 : -1;
        int bin = j * B / m;
        int ki = kmer_index(c0, c1, c2);
        int k;
        if (ki < 0) k = refIndx;
        else k = rand_indx(&subs[((size_t)ki * B + bin) * 4], 4, rng.counter ? xs / 4294967296.0 : drawB());
        if (k < 0) {
            out_b[j] = 'N';
            out_q[j] = (char)(long)(33 + (53 - 33) * (rng.counter ? xq / 4294967296.0 : drawBi()));   // getRandBaseQuality 1578-1580
        } else {
            out_b[j] = BASES[k];
            int bp = refIndx * 4 + k;
            if (rng.counter) {                                                  // [REMAP] alias lookup on the raw 32-bit draw
                const int K = P.qualK, abits = K == 16 ? 4 : K == 64 ? 6 : 7;
                const uint32_t* row = &P.qualAlias[((size_t)bp * B + bin) * ((size_t)K + K / 4)];
                const uint32_t x = xq, col = x >> (32 - abits), e = row[col];
                uint32_t pick = (x & ((1u << (32 - abits)) - 1u)) < (e >> abits) ? col : (e & (uint32_t)(K - 1));
                if (P.biasAlias != 1.0 && (double)(x & ((1u << (32 - abits)) - 1u)) >= (double)(1u << (32 - abits)) / P.biasAlias) pick = (col + 1u) & (uint32_t)(K - 1);   // (bias test only)
                out_q[j] = (char)(33 + reinterpret_cast<const uint8_t*>(row + K)[pick]);
            } else out_q[j] = (char)(33 + rand_indx(&P.qual[((size_t)bp * B + bin) * 94], 94, drawB()));
        }
    }
    return m;
}

// ---------------------------------------------------------------------------
// Pipeline state
// ---------------------------------------------------------------------------
struct Frag {                 // lib/fragment/Fragment.h:20-31
    int rec; long start; int len; int strand; int primers;
    std::vector<uint8_t> T;   // template strand c(F) used by amplify (Fragment.cpp:65-68)
};
struct Amp {                  // lib/amplicon/Amplicon.h:47-53 (packed fields unpacked)
    uint32_t parent;          // semis: fragment index; fulls: index into semis
    uint32_t spos, len, gc;
    uint32_t primers;         // 12-bit (Amplicon.cpp:76-79)
    uint64_t uid;             // lineage id (counter-mode key)
    uint32_t err_off; uint32_t err_cnt;
};
struct AmpList { std::vector<Amp> a; std::vector<uint32_t> errs; };   // err = pos<<3 | alt  (AmpError, Amplicon.cpp:13-45)

struct Params {
    long primers = 100000; double gamma = 1e-9, coverage = 5, ber = 3.4e-4;
    int isize = 260; bool paired = true; int threads = 1;
    int ampMin = 1000, ampMax = 2000, fragSize = 1000, fragMin = 10000, fragMax = 100000;   // Config.cpp:35-48, Fragment.cpp:15-16
    bool counter = false; uint64_t seed = 1; long long fixed_time = 1234567890LL; bool verbose = true;
    int shard_rank = 0, shard_count = 1; scso_allreduce_fn allreduce = nullptr; scso_allgatherv_fn allgatherv = nullptr; void* coll_user = nullptr;
};

inline uint64_t semi_uid(uint64_t frag, uint32_t pass, uint32_t i) { return (frag << 23) | ((uint64_t)pass << 20) | i; }
inline uint64_t full_uid(uint64_t semi, uint32_t cyc, uint32_t i) { return (semi << 15) | ((uint64_t)cyc << 12) | i; }

struct Sim {
    Params prm; Rng rng; RefStreams streams;
    std::vector<Record> recs; Profile* prof = nullptr;
    std::vector<Frag> frags; AmpList semis, fulls;
    std::vector<long> primerCount;          // 65536 counters (Malbac.cpp:36-81; flat instead of trie)
    int exhausted_passes = 0;               // counter mode: passes in which a primer type ran dry (run again sequentially, amplify_pass)
    std::vector<long> primerUsed;           // attachments per primer type over the whole run (dump / statistics only)
    std::vector<uint64_t> binom;            // [REMAP] counter mode: error-count thresholds
    double bias_attach = 1.0, bias_gc = 1.0; // test_bias() of the two per-unit branches (1.0 outside the bias test)
    // sharded mode: this shard owns fragments [frag_gbase, frag_gbase + frags.size()) of the global list
    uint64_t frag_gbase = 0;
    std::vector<size_t> semi_block_end;     // local semi count after each fragment pass (block p = semis made in pass p)
    struct Seg { int c, p; size_t count; };
    std::vector<Seg> full_segs;             // local fulls list = concatenation of these (cycle c asc, p desc)
    std::vector<uint64_t> gidx;             // local full index -> global list index (empty = identity)
    void allreduce(uint64_t* v, uint64_t n) {
        if (prm.shard_count <= 1) return;
        if (!prm.allreduce || prm.allreduce(prm.coll_user, v, n)) fail("sharded run: allreduce hook missing or failed");
    }
    unsigned long totalPrimers = 0;
    std::vector<unsigned> readNumbers;
    // GC-factor engines (ref mode): Profile.cpp:1405-1411
    std::vector<std::default_random_engine> gcGen; std::vector<std::normal_distribution<double>> gcDist;
    ~Sim() { delete prof; }
};

template <class F>
void parallel_blocks(size_t n, int threads, size_t min_block, F f) {
    if (n == 0) return;
    size_t nb = std::max<size_t>(1, std::min<size_t>((size_t)threads * 4, (n + min_block - 1) / min_block));
    if (threads <= 1) nb = 1;
    std::vector<std::thread> th; size_t next = 0; std::mutex* mu = new std::mutex;
    auto worker = [&]() {
        for (;;) { size_t b; { std::lock_guard<std::mutex> g(*mu); b = next++; } if (b >= nb) break;
                   f(b, n * b / nb, n * (b + 1) / nb); }
    };
    if (threads <= 1) worker(); else { for (int t = 0; t < threads; ++t) th.emplace_back(worker); for (auto& t : th) t.join(); }
    delete mu;
}

// ---- a1: Genome::splitToFrags (Genome.cpp:753-782) + Fragment::createSequence (Fragment.cpp:40-50)
void split_to_frags(Sim& S) {
    const Params& p = S.prm;
    for (size_t r = 0; r < S.recs.size(); ++r) {
        long chrLen = (long)S.recs[r].code.size(), pos = 1; uint32_t k = 0;
        while (pos <= chrLen) {
            double u = S.rng.grand(mk(ST_FRAGSPLIT, 0, r, k++, 0));
            int fl = (int)(long)(p.fragMin + (p.fragMax + 1 - p.fragMin) * u);   // randomInteger(minSize, maxSize+1)
            if (pos + fl - 1 > chrLen) break;
            S.frags.push_back(Frag{(int)r, pos, fl, 1, 0, {}});
            S.frags.push_back(Frag{(int)r, pos, fl, -1, 0, {}});
            pos += fl;
        }
        if (pos <= chrLen) {                                                    // tail emitted twice with strand +1 (quirk kept)
            S.frags.push_back(Frag{(int)r, pos, (int)(chrLen - pos + 1), 1, 0, {}});
            S.frags.push_back(Frag{(int)r, pos, (int)(chrLen - pos + 1), 1, 0, {}});
        }
    }
    if (p.shard_count > 1) {                 // contiguous fragment ranges balanced by bases (same rule as the product)
        uint64_t tot = 0; for (auto& f : S.frags) tot += (uint64_t)f.len;
        std::vector<size_t> cut(p.shard_count + 1, S.frags.size()); cut[0] = 0;
        uint64_t acc = 0; int sh = 1;
        for (size_t i = 0; i < S.frags.size() && sh < p.shard_count; ++i) { acc += (uint64_t)S.frags[i].len; while (sh < p.shard_count && acc * p.shard_count >= tot * (uint64_t)sh) cut[sh++] = i + 1; }
        const size_t lo = cut[p.shard_rank], hi = cut[p.shard_rank + 1];
        S.frags = std::vector<Frag>(S.frags.begin() + lo, S.frags.begin() + hi); S.frag_gbase = lo;
    }
    for (auto& f : S.frags) {
        const uint8_t* g = S.recs[f.rec].code.data() + (f.start - 1);
        f.T.resize(f.len);
        // stored F: strand +1 -> reversed slice, -1 -> complemented slice; template strand T = c(F)
        if (f.strand == 1) for (int i = 0; i < f.len; ++i) f.T[i] = comp_code(g[f.len - 1 - i]);
        else for (int i = 0; i < f.len; ++i) f.T[i] = comp_code(comp_code(g[i]));
    }
}

// ---- a2: primer pool (Malbac.cpp:36-103) ----------------------------------------------------
inline int primer_index(const uint8_t* t) {
    int idx = 0; for (int k = 0; k < 8; ++k) { if (t[k] > 3) return -1; idx = (idx << 2) | t[k]; } return idx;
}
struct PrimerPool {
    Sim& S; std::vector<long>* pending;           // what this worker took during the pass (counter mode)
    bool live;                                    // decrement the shared stock at once (the reference's way), or look at the pass's start
    // counter mode, a pass in which a type runs dry (amplify_pass): the types in `dry_set` are handed out from `left` (live), the
    // others as of the pass's start; sig = a 64-bin filter of the types the template in hand has asked for, asked_dry = one of dry_set
    const std::vector<uint8_t>* dry_set = nullptr; std::vector<long>* left = nullptr; uint64_t sig = 0; bool asked_dry = false;
    static uint64_t bin(int idx) { return 1ull << (((uint32_t)idx * 0x9E3779B1u) >> 26); }
    bool take(const uint8_t* t) {                 // updatePrimerCount(s, -1)
        int idx = primer_index(t);
        if (idx < 0) return false;                // N-containing 8-mer: node with no stock (A.8 D-item)
        sig |= bin(idx);
        if (dry_set && (*dry_set)[idx]) {
            asked_dry = true;
            if (!left) return S.primerCount[idx] > 0 ? ((*pending)[idx]++, true) : false;   // (the dry run that recovers the first run's takes)
            if ((*left)[idx] <= 0) return false;
            (*left)[idx]--; (*pending)[idx]++; return true;
        }
        if (!live) {                              // counter mode: the stock as of the pass's start
            if (S.primerCount[idx] <= 0) return false;
            (*pending)[idx]++; return true;
        }
        if (S.primerCount[idx] - 1 < 0) return false;
        S.primerCount[idx] -= 1; S.primerUsed[idx] += 1; if (pending) (*pending)[idx]++; return true;
    }
};

// ---- a3: Malbac::setPrimers (Malbac.cpp:236-283) + poissRand (MyDefine.cpp:69-80) ------------
long poiss_rand(Sim& S, double lambda, uint32_t call, uint32_t kind, uint64_t tuid) {
    if (S.prm.counter && kind == 1 && lambda <= 256.0) {
        // [REMAP] semi amplicons (lambda of a few units, 10^8 of them per cycle at whole-genome size): Knuth's loop in its
        // product form -- p *= u until p < exp(-lambda) -- one multiply per draw instead of one logarithm
        const double L = det_exp(-lambda); long x = -1; double pr = 1.0; uint32_t t = 0;
        do { const double u = S.rng.grand(mk(ST_POISSON, kind | (call << 1), tuid, t / 4, t % 4)); t++; pr = pr * u; x++; } while (pr >= L);
        return x;
    }
    long x = -1; double log1 = 0, log2 = -lambda; uint32_t t = 0;
    do {
        double u = S.rng.grand(mk(ST_POISSON, kind | (call << 1), tuid, t / 4, t % 4)); t++;
        log1 += S.prm.counter ? det_log(u) : log(u);      // [REMAP] counter mode: software log a GPU reproduces bit for bit
        x++;
    } while (log1 >= log2);
    return x;
}
void set_primers(Sim& S, bool onlyFrags, uint32_t call) {
    uint64_t tot[2] = {0, 0};                                                      // {templateNum, totalLen}: integer sums, shard-order free
    for (auto& f : S.frags) tot[1] += (unsigned)f.len;
    tot[0] += S.frags.size();
    if (!onlyFrags) { tot[0] += S.semis.a.size(); for (auto& a : S.semis.a) tot[1] += a.len; }
    S.allreduce(tot, 2);
    const unsigned long templateNum = tot[0]; const double totalLen = (double)tot[1];
    unsigned long expected = (unsigned long)(S.totalPrimers * S.prm.gamma * templateNum);
    uint64_t count = 0;
    // (counter mode: the draws are keyed by template, so the templates are handed to the workers in blocks -- same budgets)
    const int th = S.prm.counter ? S.prm.threads : 1;
    std::vector<uint64_t> part((size_t)std::max(1, th) * 4, 0);
    parallel_blocks(S.frags.size(), th, 16, [&](size_t b, size_t lo, size_t hi) {
        uint64_t c = 0;
        for (size_t i = lo; i < hi; ++i) {
            double lambda = expected * (1.0 * (unsigned)S.frags[i].len / totalLen);
            unsigned long k = (unsigned long)poiss_rand(S, lambda, call, 0, S.frag_gbase + i);
            c += k; S.frags[i].primers = (int)k;
        }
        part[b] += c;
    });
    if (!onlyFrags) parallel_blocks(S.semis.a.size(), th, 4096, [&](size_t b, size_t lo, size_t hi) {
        uint64_t c = 0;
        for (size_t i = lo; i < hi; ++i) {
            Amp& a = S.semis.a[i];
            double lambda = expected * (1.0 * a.len / totalLen);
            unsigned long k = (unsigned long)poiss_rand(S, lambda, call, 1, a.uid);
            c += k; a.primers = (uint32_t)(k & 0xFFF);
        }
        part[b] += c;
    });
    for (uint64_t v : part) count += v;
    S.allreduce(&count, 1);
    S.totalPrimers -= count;
}

// [REMAP] counter mode draws the amplification errors of one new amplicon as K ~ Binomial(l-8, ber)
// followed by K distinct uniform positions -- the same distribution as the reference's one
// Bernoulli(ber) draw per base (Fragment.cpp:100-104, Amplicon.cpp:203-207) at ~4 draws per amplicon
// instead of ~1500.  T[n][k] = floor(CDF_n(k) * 2^64); IEEE * + / only, so every build gets the same table.
const int BINOM_KMAX = 16;
std::vector<uint64_t> binom_table(double ber, int n_min, int n_max) {
    std::vector<uint64_t> T((size_t)(n_max - n_min + 1) * BINOM_KMAX);
    const double q = 1 - ber, ratio = ber / q;
    for (int n = n_min; n <= n_max; ++n) {
        double pmf = 1, b = q; for (unsigned e = (unsigned)n; e; e >>= 1) { if (e & 1) pmf = pmf * b; b = b * b; }
        double cdf = 0;
        for (int k = 0; k < BINOM_KMAX; ++k) {
            cdf += pmf;
            T[(size_t)(n - n_min) * BINOM_KMAX + k] = cdf >= 1.0 ? ~0ull : (uint64_t)(cdf * 18446744073709551616.0);
            pmf = pmf * (double)(n - k) / (double)(k + 1) * ratio;
        }
    }
    return T;
}

// [REMAP] attach tries in counter mode.  A try of the reference (Fragment.cpp:76-82) draws spos uniformly from [27, len) and
// the length from [amin, amax]; it fails outright when spos + alen > len.  Such tries are i.i.d., so (a) the number of them
// before a try that fits is geometric with failure probability 1 - N/M and (b) the try that fits is uniform over the N
// feasible pairs: every spos <= len - amax admits all W lengths (A positions), then position len - amin + 1 - d admits d
// lengths, d = 1 .. D.
struct AttachFit { uint32_t A, W, D, N; };
AttachFit attach_fit_count(uint32_t len, uint32_t amin, uint32_t amax) {
    AttachFit f; f.W = amax - amin + 1u;
    f.A = len >= amax + 27u ? len - amax - 26u : 0u;
    f.D = (len - amin - 26u) - f.A;
    f.N = f.A * f.W + f.D * (f.D + 1u) / 2u;
    return f;
}
void attach_fit_decode(const AttachFit& f, uint32_t len, uint32_t amin, uint32_t r, unsigned& spos, unsigned& alen) {
    const uint32_t full = f.A * f.W;
    if (r < full) { spos = 27u + r / f.W; alen = amin + r % f.W; return; }
    const uint32_t q = r - full;
    uint32_t d = (uint32_t)((1.0 + sqrt(1.0 + 8.0 * (double)q)) * 0.5);
    while (d > 1u && d * (d - 1u) / 2u > q) --d;
    while (d * (d + 1u) / 2u <= q) ++d;
    alen = amin + (q - d * (d - 1u) / 2u);
    spos = len - amin + 1u - d;
}
uint32_t attach_gap(double u, double qfail) {                                       // P(gap >= g) = qfail^g, capped at 51
    double acc = qfail; uint32_t g = 0;
    while (g < 51u && u < acc) { acc = acc * qfail; ++g; }
    return g;
}

// ---- a4/a5: Fragment::amplify (Fragment.cpp:52-137) / Amplicon::amplify (Amplicon.cpp:156-240) --
// T = template strand (c(F) or c(S)); appends created amplicons in creation order.
void amplify_template(Sim& S, Rng& rng, PrimerPool& pool, bool fromFrag, uint64_t tuid, uint32_t parent,
                      const uint8_t* T, unsigned length, unsigned primerNum, uint32_t pass,
                      std::vector<uint8_t>& posAttached, AmpList& out) {
    const Params& p = S.prm;
    if ((int)length < p.ampMin + 27) return;
    posAttached.assign(length, 0);
    const uint32_t kind = fromFrag ? 0u : 1u, aux = kind | (pass << 1);
    for (unsigned i = 0; i < primerNum; ++i) {
        int tryTimes = 0; unsigned spos = 0, alen = 0;
        // [REMAP] counter mode: primer i of this template owns a xoshiro128++ stream seeded by Philox block i of ST_ATTACH;
        // every try takes two draws from it (position, then length)
        Xoshiro xt;
        if (rng.counter) { uint32_t c[4] = {i, (uint32_t)tuid, (uint32_t)(tuid >> 32), ST_ATTACH | (aux << 8)}, o[4]; philox(c, rng.key, o); xt.seed(o); }
        if (rng.counter) {
            // [REMAP] the tries that do not fit are i.i.d.: their number before a fitting try is geometric (one draw), and the
            // fitting try is uniform over the feasible (position, length) pairs (a 64-bit draw); tries that fit but land on
            // an attached position or a primer type without stock fail as in the reference and count as tries
            const AttachFit fit = attach_fit_count(length, (uint32_t)p.ampMin, (uint32_t)p.ampMax);
            const double qfail = 1.0 - (double)fit.N / ((double)(length - 27) * (double)fit.W) / S.bias_attach;
            do {
                tryTimes += (int)attach_gap(((double)xt.next() + 0.5) / 4294967296.0, qfail) + 1;
                if (tryTimes > 50) break;
                const uint64_t hi = xt.next(), lo = xt.next(), x64 = (hi << 32) | lo;
                attach_fit_decode(fit, length, (uint32_t)p.ampMin, (uint32_t)(((unsigned __int128)x64 * fit.N) >> 64), spos, alen);
                if (posAttached[spos] == 1) continue;
                if (pool.take(T + spos)) break;
            } while (1);
        } else
        do {
            const double u1 = rng.integer(Key{});
            const double u2 = rng.real(Key{});
            spos = (unsigned)(long)(27 + ((long)length - 27) * u1);
            alen = (unsigned)(p.ampMin + (double)(p.ampMax + 1 - p.ampMin) * u2);
            tryTimes++;
            if (tryTimes > 50) break;
            if (spos + alen > length || posAttached[spos] == 1) continue;
            if (pool.take(T + spos)) break;
        } while (1);
        if (tryTimes > 50) break;
        posAttached[spos] = 1;
        int gc = 0; bool hasN = false;                                           // countGC MyDefine.cpp:434-452
        for (unsigned j = 0; j < alen; ++j) { uint8_t c = T[spos + j]; if (is_gc(c)) gc++; else if (c > 3) hasN = true; }
        if (hasN) gc = 0;
        const uint64_t nuid = fromFrag ? semi_uid(tuid, pass, i) : full_uid(tuid, pass, i);
        uint32_t off = (uint32_t)out.errs.size();
        auto add_error = [&](unsigned j) {
            uint8_t base = T[spos + j]; unsigned n; uint32_t a = 0;
            do {
                Key k = mk(ST_ERRALT, kind, nuid, j | ((a / 4) << 16), a % 4); a++;
                // Fragment.cpp:110 draws from the int stream, Amplicon.cpp:213 from the real stream
                n = fromFrag ? (unsigned)(long)(0 + (4 - 0) * rng.integer(k)) : (unsigned)(0 + (4 - 0) * rng.real(k));
            } while (n == base);
            if (is_gc((uint8_t)n)) gc++;
            if (is_gc(base)) gc--;
            out.errs.push_back((j << 3) | n);
        };
        if (!p.counter) {
            for (unsigned j = 8; j < alen; ++j) if (rng.real(mk(ST_ERR, kind, nuid, 0, 0)) < p.ber) add_error(j);
        } else {                                                                   // [REMAP] binomial count + distinct positions
            const unsigned ntr = alen - 8;
            uint32_t c[4] = {0, (uint32_t)nuid, (uint32_t)(nuid >> 32), ST_ERR | (kind << 8)}, o[4];
            philox(c, rng.key, o);
            const uint64_t x64 = ((uint64_t)o[0] << 32) | o[1];
            const uint64_t* Tn = &S.binom[(size_t)(ntr - (p.ampMin - 8)) * BINOM_KMAX];
            unsigned K = 0; while (K < (unsigned)BINOM_KMAX && x64 >= Tn[K]) ++K;
            unsigned pos[BINOM_KMAX], cnt = 0, q = 0;
            while (cnt < K) {
                if ((q & 3) == 0) { c[0] = 1 + (q >> 2); philox(c, rng.key, o); }
                const unsigned cand = 8 + (unsigned)(((uint64_t)ntr * o[q & 3]) >> 32); ++q;
                bool dup = false; for (unsigned z = 0; z < cnt; ++z) dup |= pos[z] == cand;
                if (!dup) pos[cnt++] = cand;
            }
            std::sort(pos, pos + cnt);
            for (unsigned z = 0; z < cnt; ++z) add_error(pos[z]);
        }
        out.a.push_back(Amp{parent, spos, alen, (uint32_t)std::max(0, gc), 0, nuid, off, (uint32_t)out.errs.size() - off});
    }
}

// ---- a7: Amplicon::getSequence (Amplicon.cpp:255-382), closed form ----------------
// semi S[t] = (T_frag[s..s+l) with subs)[l-1-t]
void semi_sequence(const Sim& S, const Amp& a, std::vector<uint8_t>& out) {
    const Frag& f = S.frags[a.parent];
    out.resize(a.len);
    std::vector<uint8_t> tmp(f.T.begin() + a.spos, f.T.begin() + a.spos + a.len);
    for (uint32_t e = 0; e < a.err_cnt; ++e) { uint32_t v = S.semis.errs[a.err_off + e]; tmp[v >> 3] = (uint8_t)(v & 7); }
    for (uint32_t t = 0; t < a.len; ++t) out[t] = tmp[a.len - 1 - t];
}
// full U[t] = c(S)[s2+t] with subs
void full_sequence(const Sim& S, const Amp& a, std::vector<uint8_t>& scratch, std::vector<uint8_t>& out) {
    semi_sequence(S, S.semis.a[a.parent], scratch);
    out.resize(a.len);
    for (uint32_t t = 0; t < a.len; ++t) out[t] = comp_code(scratch[a.spos + t]);
    for (uint32_t e = 0; e < a.err_cnt; ++e) { uint32_t v = S.fulls.errs[a.err_off + e]; out[v >> 3] = (uint8_t)(v & 7); }
}

// append `add` to `dst`; ref order = reversed creation order per pool task (insertLinkList prepends,
// Amplicon.cpp:574-585; one task at -t 1, Malbac.cpp:324,351).  Counter mode keeps the same order.
void append_reversed(AmpList& dst, AmpList& add) {
    dst.a.reserve(dst.a.size() + add.a.size()); dst.errs.reserve(dst.errs.size() + add.errs.size());
    for (size_t i = add.a.size(); i-- > 0;) {
        Amp a = add.a[i]; uint32_t off = (uint32_t)dst.errs.size();
        for (uint32_t e = 0; e < a.err_cnt; ++e) dst.errs.push_back(add.errs[a.err_off + e]);
        a.err_off = off; dst.a.push_back(a);
    }
}

// ---- one pass over the templates [0, n) of a list (Malbac::amplifyFrags 318-343, amplifySemiAmplicons 345-368).
// The reference decrements the stock of a primer type at every attachment (Malbac.cpp:91-103) and walks the templates
// in list order (one pool task at -t 1): a type is used exactly `stock` times, by the first `stock` attachments that ask
// for it.  Counter mode keeps exactly that, free of the thread / shard schedule:
//   1. every template is evaluated against the stock as of the pass's START, in parallel.  If no type was then asked for
//      more often than it has stock, its availability never changed during the pass: this IS the sequential loop's result.
//   2. Otherwise let D be the over-demanded ("dry") types.  A template that never asks for a type of D evaluates in the
//      sequential loop exactly as in step 1 (every type it asks for is to be had throughout -- checked in step 3).  The others
//      -- few: found by a 64-bin filter of the types each template asked for, then by evaluating the candidates again -- are
//      evaluated once more ONE AFTER THE OTHER in list order, the types of D handed out from a live count; in a sharded
//      job segment by segment in the whole job's list order (`segs`: local template ranges, the order of the shards
//      inside each), the live counts handed from shard to shard by an all-reduce to which only the owner contributes.
//   3. What the templates of step 2 now take elsewhere may push a further type over its stock: it joins D and step 2 is
//      done again from step 1's results.  (A whole-genome pass runs dry in a handful of types -- poly-A / poly-T 8-mers --
//      and step 2 touches one template in a thousand; the plain sequential loop over 10^8 templates took a quarter of an hour.)
struct PassSeg { size_t lo, hi; bool shards_descending; };
struct EvalScratch { std::vector<uint8_t> pa, seq, tc; };
template <class Eval>
void amplify_pass(Sim& S, size_t n, size_t min_block, const std::vector<PassSeg>& segs, Eval eval, AmpList& all) {
    const bool ctr = S.prm.counter; const int th = ctr ? S.prm.threads : 1;
    auto append = [&](const AmpList& src, size_t first, size_t count) {
        for (size_t k = first; k < first + count; ++k) { Amp a = src.a[k]; uint32_t off = (uint32_t)all.errs.size();
            for (uint32_t e = 0; e < a.err_cnt; ++e) all.errs.push_back(src.errs[a.err_off + e]); a.err_off = off; all.a.push_back(a); }
    };
    if (!ctr) {
        AmpList part; PrimerPool pool{S, nullptr, true}; EvalScratch sc;
        for (size_t i = 0; i < n; ++i) eval(i, pool, part, sc);
        append(part, 0, part.a.size()); return;
    }
    const size_t nbMax = (size_t)std::max(1, th) * 4;
    std::vector<AmpList> parts(nbMax); std::vector<std::vector<long>> pend(nbMax); std::vector<size_t> blo(nbMax, 0), bhi(nbMax, 0);
    std::vector<uint32_t> cnt(n, 0); std::vector<uint64_t> sig(n, 0);
    parallel_blocks(n, th, min_block, [&](size_t b, size_t lo, size_t hi) {
        pend[b].assign(65536, 0); PrimerPool pool{S, &pend[b], false}; EvalScratch sc; blo[b] = lo; bhi[b] = hi;
        for (size_t i = lo; i < hi; ++i) { pool.sig = 0; const size_t before = parts[b].a.size(); eval(i, pool, parts[b], sc); cnt[i] = (uint32_t)(parts[b].a.size() - before); sig[i] = pool.sig; }
    });
    std::vector<uint64_t> sum1(65536, 0);
    for (auto& v : pend) for (size_t i = 0; i < v.size(); ++i) sum1[i] += (uint64_t)v[i];
    S.allreduce(sum1.data(), sum1.size());                                          // sharded: the demand of all shards
    std::vector<uint8_t> dry(65536, 0); bool over = false;
    for (size_t i = 0; i < sum1.size(); ++i) if ((long)sum1[i] > S.primerCount[i]) { dry[i] = 1; over = true; }
    std::vector<uint64_t> total = sum1;
    std::vector<size_t> touch; std::vector<AmpList> repl;                           // step 2's templates (ascending) and what they make
    if (over) S.exhausted_passes++;
    if (over && getenv("SCSO_PLAIN_SEQUENTIAL_PASS")) {
        // the reference's loop to the letter -- one worker, list order, live decrement of every type -- for the test that steps 2 and 3
        // give its result (tests/test_oracle_stats.py); unsharded only
        if (S.prm.shard_count > 1) fail("SCSO_PLAIN_SEQUENTIAL_PASS: unsharded runs only");
        AmpList part; PrimerPool pool{S, nullptr, true}; EvalScratch sc;
        for (size_t i = 0; i < n; ++i) eval(i, pool, part, sc);
        append(part, 0, part.a.size()); return;
    }
    while (over) {
        uint64_t sig_dry = 0; for (size_t i = 0; i < 65536; ++i) if (dry[i]) sig_dry |= PrimerPool::bin((int)i);
        // the templates that ask for a dry type in step 1's evaluation, and what they took then
        std::vector<size_t> cand; for (size_t i = 0; i < n; ++i) if (sig[i] & sig_dry) cand.push_back(i);
        std::vector<uint8_t> asks(cand.size(), 0); std::vector<std::vector<long>> old(nbMax);
        parallel_blocks(cand.size(), th, 16, [&](size_t b, size_t lo, size_t hi) {
            if (old[b].empty()) old[b].assign(65536, 0);
            std::vector<long> mine(65536, 0); AmpList scratch; EvalScratch sc;
            for (size_t k = lo; k < hi; ++k) {
                PrimerPool pool{S, &mine, false}; pool.dry_set = &dry; scratch.a.clear(); scratch.errs.clear();
                eval(cand[k], pool, scratch, sc);
                if (pool.asked_dry) asks[k] = 1;
            }
            // second sweep over the block's true ones, counting their takes alone
            std::fill(mine.begin(), mine.end(), 0);
            for (size_t k = lo; k < hi; ++k) if (asks[k]) { PrimerPool pool{S, &mine, false}; pool.dry_set = &dry; scratch.a.clear(); scratch.errs.clear(); eval(cand[k], pool, scratch, sc); }
            for (size_t i = 0; i < 65536; ++i) old[b][i] += mine[i];
        });
        touch.clear(); for (size_t k = 0; k < cand.size(); ++k) if (asks[k]) touch.push_back(cand[k]);
        repl.assign(touch.size(), AmpList());
        std::vector<long> left(65536, 0); for (size_t i = 0; i < 65536; ++i) left[i] = S.primerCount[i];
        std::vector<long> took(65536, 0);
        size_t tk = 0;
        for (size_t sg = 0; sg < segs.size(); ++sg) for (int k = 0; k < S.prm.shard_count; ++k) {
            const int owner = segs[sg].shards_descending ? S.prm.shard_count - 1 - k : k;
            std::vector<long> mine(65536, 0);
            if (owner == S.prm.shard_rank) {
                EvalScratch sc;
                for (; tk < touch.size() && touch[tk] < segs[sg].hi; ++tk) {
                    if (touch[tk] < segs[sg].lo) continue;
                    PrimerPool pool{S, &mine, false}; pool.dry_set = &dry; pool.left = &left;
                    eval(touch[tk], pool, repl[tk], sc);
                }
                for (size_t i = 0; i < 65536; ++i) took[i] += mine[i];
            }
            if (S.prm.shard_count > 1) {                                             // the live counts of the dry types travel on
                std::vector<uint64_t> d(65536, 0); for (size_t i = 0; i < 65536; ++i) if (dry[i]) d[i] = (uint64_t)mine[i];
                S.allreduce(d.data(), d.size());
                if (owner != S.prm.shard_rank) for (size_t i = 0; i < 65536; ++i) left[i] -= (long)d[i];
            }
        }
        // the pass's demand now: step 1's, less what step 2's templates took then, plus what they take now
        std::vector<uint64_t> delta(2 * 65536, 0);
        for (auto& v : old) if (!v.empty()) for (size_t i = 0; i < 65536; ++i) delta[i] += (uint64_t)v[i];
        for (size_t i = 0; i < 65536; ++i) delta[65536 + i] = (uint64_t)took[i];
        S.allreduce(delta.data(), delta.size());
        over = false;
        for (size_t i = 0; i < 65536; ++i) {
            total[i] = sum1[i] - delta[i] + delta[65536 + i];
            if ((long)total[i] > S.primerCount[i]) { if (dry[i]) fail("internal: a dry primer type was handed out beyond its stock"); dry[i] = 1; over = true; }
        }
    }
    for (size_t i = 0; i < 65536; ++i) { S.primerUsed[i] += (long)total[i]; S.primerCount[i] -= (long)total[i]; }
    // the pass's amplicons in template order: step 1's, step 2's in their templates' places
    size_t na = 0, ne = 0; for (auto& pt : parts) { na += pt.a.size(); ne += pt.errs.size(); }
    all.a.reserve(all.a.size() + na + 1024); all.errs.reserve(all.errs.size() + ne + 1024);
    size_t tk = 0;
    for (size_t b = 0; b < nbMax; ++b) {
        size_t at = 0;
        if (touch.empty() || tk >= touch.size() || touch[tk] >= bhi[b]) { append(parts[b], 0, parts[b].a.size()); }
        else for (size_t i = blo[b]; i < bhi[b]; ++i) {
            if (tk < touch.size() && touch[tk] == i) { append(repl[tk], 0, repl[tk].a.size()); ++tk; }
            else append(parts[b], at, cnt[i]);
            at += cnt[i];
        }
        parts[b] = AmpList();                                                       // (a whole-genome pass holds tens of GB here)
    }
}

// Malbac::amplifyFrags (Malbac.cpp:318-343)
void amplify_frags(Sim& S, uint32_t pass) {
    const size_t n = S.frags.size();
    AmpList all; Rng rng = S.rng;
    amplify_pass(S, n, 1, {PassSeg{0, n, false}}, [&](size_t i, PrimerPool& pool, AmpList& out, EvalScratch& sc) {
        Frag& f = S.frags[i];
        amplify_template(S, rng, pool, true, S.frag_gbase + i, (uint32_t)i, f.T.data(), f.len, f.primers, pass, sc.pa, out);
    }, all);
    append_reversed(S.semis, all);
    S.semi_block_end.push_back(S.semis.a.size());
}
// Malbac::amplifySemiAmplicons (Malbac.cpp:345-368)
void amplify_semis(Sim& S, uint32_t cyc) {
    const size_t n = S.semis.a.size();
    // the whole job's semi list: block p = the semis of fragment pass p, inside it the fragments DEscending (append_reversed),
    // hence the shards of a sharded job (contiguous fragment ranges) descending too
    std::vector<PassSeg> segs;
    for (size_t pb = 0; pb < S.semi_block_end.size(); ++pb) segs.push_back(PassSeg{pb ? S.semi_block_end[pb - 1] : 0, S.semi_block_end[pb], true});
    AmpList all; Rng rng = S.rng;
    amplify_pass(S, n, 64, segs, [&](size_t i, PrimerPool& pool, AmpList& out, EvalScratch& sc) {
        const Amp& a = S.semis.a[i];
        if ((int)a.len < S.prm.ampMin + 27) return;
        semi_sequence(S, a, sc.seq); sc.tc.resize(sc.seq.size());
        for (size_t t = 0; t < sc.seq.size(); ++t) sc.tc[t] = comp_code(sc.seq[t]);     // Amplicon.cpp:171-172
        amplify_template(S, rng, pool, false, a.uid, (uint32_t)i, sc.tc.data(), a.len, a.primers, cyc, sc.pa, out);
    }, all);
    {   // segments of this cycle in stored (reversed) order: semis made in fragment pass p, p descending
        std::vector<size_t> cnt(S.semi_block_end.size(), 0);
        for (auto& a : all.a) { size_t pblk = 0; while (a.parent >= S.semi_block_end[pblk]) ++pblk; cnt[pblk]++; }
        for (int pb = (int)cnt.size() - 1; pb >= 0; --pb) S.full_segs.push_back(Sim::Seg{(int)cyc, pb, cnt[pb]});
    }
    append_reversed(S.fulls, all);
}

// Malbac::amplify (Malbac.cpp:173-201)
void amplify(Sim& S) {
    if (S.prm.verbose) fprintf(stderr, "\nMALBAC amplification...\n");
    S.primerCount.assign(65536, S.prm.primers); S.primerUsed.assign(65536, 0);
    if (S.prm.counter) S.binom = binom_table(S.prm.ber * test_bias("errors"), S.prm.ampMin - 8, S.prm.ampMax - 8);
    S.bias_attach = test_bias("attach"); S.bias_gc = test_bias("gc");
    S.totalPrimers = 65536UL * (unsigned long)S.prm.primers;
    set_primers(S, true, 0);
    amplify_frags(S, 0);
    for (int i = 0; i < 5; ++i) {
        if (S.totalPrimers == 0) break;
        if (S.prm.verbose) fprintf(stderr, "cycle number: %d\n", i + 1);
        set_primers(S, false, i + 1);
        amplify_semis(S, i);
        if (S.prm.verbose) fprintf(stderr, "semi amplicon amplification done!\n");
        if (i < 4) { amplify_frags(S, i + 1); if (S.prm.verbose) fprintf(stderr, "fragment amplification done!\n"); }
    }
}

// ---- a8: Amplicon::getWeightedLength (Amplicon.cpp:396-400) + Profile::getGCFactor (1503-1513) ---
double gc_factor(Sim& S, int gc, uint64_t uid) {
    if (gc < 0 || gc > 100) return 0;
    if (!S.prm.counter) {
        double v = S.gcDist[gc](S.gcGen[gc]);
        while (v < 0) v = S.gcDist[gc](S.gcGen[gc]);
        return v;
    }
    // [REMAP] Marsaglia polar normal from keyed uniforms; log via det_log so a GPU can reproduce the bits.
    uint32_t o[4] = {0, 0, 0, 0};
    for (uint32_t a = 0;; ++a) {                                                   // attempt a = words 2(a&1), 2(a&1)+1 of Philox block a/2
        if ((a & 1u) == 0) { uint32_t c[4] = {a >> 1, (uint32_t)uid, (uint32_t)(uid >> 32), ST_WEIGHT}; philox(c, S.rng.key, o); }
        const uint32_t w0 = (a & 1u) ? o[2] : o[0], w1 = (a & 1u) ? o[3] : o[1];
        double x = 2.0 * ((w0 + 0.5) / 4294967296.0) - 1.0, y = 2.0 * ((w1 + 0.5) / 4294967296.0) - 1.0;
        double r2 = x * x + y * y;
        if (r2 > 1.0 || r2 == 0.0) continue;
        double mult = sqrt(-2.0 * det_log(r2) / r2);
        double v = S.prof->gcMeans[gc] + S.prof->gcStd * S.bias_gc * (y * mult);
        if (v < 0) continue;
        return v;
    }
}

// ---- a9: Malbac::setReadCounts (Malbac.cpp:370-408) + randIndx_hp/batchSampling (MyDefine.cpp:191-272)
void compute_weights(Sim& S, std::vector<double>& w) {
    const size_t ac = S.fulls.a.size();
    w.assign(ac, 0.0);
    const unsigned fragSize = S.prm.fragSize;
    if (S.prm.counter) {
        parallel_blocks(ac, S.prm.threads, 4096, [&](size_t, size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) { const Amp& a = S.fulls.a[i]; int gc = 100 * a.gc / a.len;
                w[i] = gc_factor(S, gc, a.uid) * a.len / (fragSize * fragSize); } });
    } else {
        for (size_t i = 0; i < ac; ++i) { const Amp& a = S.fulls.a[i]; int gc = 100 * a.gc / a.len;
            w[i] = gc_factor(S, gc, a.uid) * a.len / (fragSize * fragSize); }
    }
}

// ---- [REMAP] fixed-shape sums of the counter mode's read allocation.  The reference adds 10^9 weights in one serial
// loop (wls.normalize, Malbac.cpp:384) and builds its CDFs by serial prefix sums (MyDefine.cpp:203-253); a serial order
// cannot be computed in parallel or by shards, so counter mode fixes a SHAPE instead (same terms, other association):
//   tree1000(v, n <= 1000): 64 accumulators, a[l] = v[l] + v[l+64] + ... in index order, then the butterfly
//       a[l] = a[l] + a[l ^ d], d = 32, 16, .. 1  (what 64 lanes with shuffle-xor compute); result a[0]
//   tree_sum(v, n): tree1000 over consecutive groups of 1000, applied again to the group sums until one value is left
//   scan1000(v, n <= 1000): rows of 64; inside a row the Hillis-Steele inclusive scan s[l] += s[l-d], d = 1, 2, .. 32
//       (shuffle-up); out[64k + l] = carry + s[l], carry = out of the row's last lane (0 before row 0)
//   scan_all(v, n): scan1000 inside groups of 1000; the groups' last entries are scanned the same way (recursively) and
//       the entry before a group is added to every element of it
// A scan of this shape may differ from a monotone sequence in the last bit, so the lookups are the BINARY search
// first_le below (on a monotone CDF it returns what randIndx's linear scan returns, MyDefine.cpp:274-282).
double tree1000(const double* v, size_t n) {
    double a[64];
    for (int l = 0; l < 64; ++l) { double acc = 0; for (size_t i = (size_t)l; i < n; i += 64) acc += v[i]; a[l] = acc; }
    for (int d = 32; d >= 1; d >>= 1) { double b[64]; for (int l = 0; l < 64; ++l) b[l] = a[l] + a[l ^ d]; memcpy(a, b, sizeof a); }
    return a[0];
}
double tree_sum(const double* v, size_t n) {
    if (n == 0) return 0;
    std::vector<double> cur, nxt;
    for (size_t s = 0; s < n; s += 1000) cur.push_back(tree1000(v + s, std::min<size_t>(1000, n - s)));
    while (cur.size() > 1) {
        nxt.clear();
        for (size_t s = 0; s < cur.size(); s += 1000) nxt.push_back(tree1000(cur.data() + s, std::min<size_t>(1000, cur.size() - s)));
        cur.swap(nxt);
    }
    return cur[0];
}
void scan1000(const double* v, size_t n, double* out) {
    double carry = 0;
    for (size_t r = 0; r < n; r += 64) {
        double s[64];
        for (int l = 0; l < 64; ++l) s[l] = r + l < n ? v[r + l] : 0.0;
        for (int d = 1; d < 64; d <<= 1) { double t[64]; for (int l = 0; l < 64; ++l) t[l] = l >= d ? s[l] + s[l - d] : s[l]; memcpy(s, t, sizeof s); }
        for (int l = 0; l < 64 && r + l < n; ++l) out[r + l] = carry + s[l];
        carry = carry + s[63];
    }
}
void scan_all(const double* v, size_t n, double* out) {
    if (n == 0) return;
    const size_t ng = (n + 999) / 1000;
    for (size_t g = 0; g < ng; ++g) scan1000(v + g * 1000, std::min<size_t>(1000, n - g * 1000), out + g * 1000);
    if (ng == 1) return;
    std::vector<double> last(ng), pre(ng);
    for (size_t g = 0; g < ng; ++g) last[g] = out[std::min(n, (g + 1) * 1000) - 1];
    scan_all(last.data(), ng, pre.data());
    for (size_t g = 1; g < ng; ++g) for (size_t i = g * 1000; i < std::min(n, (g + 1) * 1000); ++i) out[i] = pre[g - 1] + out[i];
}
inline unsigned first_le(const double* cdf, unsigned n, double u) {                 // first k with r <= cdf[k] by bisection, else n-1
    const double r = ZERO_FINAL + (1 - ZERO_FINAL) * u;
    unsigned lo = 0, hi = n;
    while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (r <= cdf[mid]) hi = mid; else lo = mid + 1; }
    return lo < n ? lo : n - 1;
}

// allocation over a weight vector in (global) list order -> read numbers
void allocate_reads(Sim& S, std::vector<double>& w, long reads, std::vector<unsigned>& readNumbers) {
    const size_t ac = w.size();
    const unsigned chunk = (unsigned)std::max<size_t>(1, std::min<size_t>(1000, ac / 1));   // loadPerThread at -t 1
    const bool ctr = S.prm.counter;
    const int th = ctr ? S.prm.threads : 1;                                       // counter mode: every loop below is per element or per chunk
    double total = 0;
    if (ctr) total = tree_sum(w.data(), ac);                                      // [REMAP] fixed-shape sum
    else for (size_t i = 0; i < ac; ++i) total += w[i];
    readNumbers.assign(ac, 0);
    unsigned long sum = 0;
    {   std::vector<unsigned long> part((size_t)std::max(1, th) * 4, 0);
        parallel_blocks(ac, th, 1 << 16, [&](size_t b, size_t lo, size_t hi) {
            unsigned long acc = 0;
            for (size_t i = lo; i < hi; ++i) { w[i] /= (ZERO_FINAL + total);      // wls.normalize(0)
                                               unsigned rc = (unsigned)(w[i] * reads); readNumbers[i] = rc; acc += rc; }
            part[b] += acc;
        });
        for (unsigned long v : part) sum += v; }
    reads -= (long)sum;
    // randIndx_hp(wls, reads, readNumbers, true)
    unsigned long n = (unsigned long)reads;
    const size_t nch = (ac + chunk - 1) / chunk;
    // a chunk's probability, and its CDF (the reference keeps every chunk's CDF from here to the sampling; a chunk's CDF is a
    // function of the chunk alone, so it is made again where it is used: 8 bytes per amplicon less)
    auto chunk_cdf = [&](size_t c, std::vector<double>& cdf, std::vector<double>& q) -> double {
        const size_t s0 = c * chunk, e = std::min(ac, s0 + chunk) - 1;
        double tp = 0;
        cdf.resize(e - s0 + 1);
        if (ctr) {                                                                // [REMAP] fixed-shape sum and scan
            tp = tree1000(&w[s0], e - s0 + 1);
            q.resize(e - s0 + 1);
            for (size_t i = s0; i <= e; ++i) q[i - s0] = w[i] / tp;
            scan1000(q.data(), q.size(), cdf.data());
        } else {
            for (size_t i = s0; i <= e; ++i) tp += w[i];
            double run = 0;
            for (size_t i = s0; i <= e; ++i) { run = run + w[i] / tp; cdf[i - s0] = run; }
        }
        return tp;
    };
    std::vector<double> totalProbs(nch); std::vector<unsigned> quota(nch); unsigned long count = 0;
    parallel_blocks(nch, th, 64, [&](size_t, size_t lo, size_t hi) {
        for (size_t c = lo; c < hi; ++c) {
            const size_t s0 = c * chunk, e = std::min(ac, s0 + chunk) - 1;
            double tp = 0;
            if (ctr) tp = tree1000(&w[s0], e - s0 + 1); else for (size_t i = s0; i <= e; ++i) tp += w[i];
            totalProbs[c] = tp; quota[c] = (unsigned)(tp * n);
        }
    });
    for (size_t c = 0; c < nch; ++c) count += quota[c];
    n -= count;
    if (n > 0 && nch) {
        std::vector<double> probs(nch);
        if (ctr) scan_all(totalProbs.data(), totalProbs.size(), probs.data());
        else { probs[0] = totalProbs[0]; for (size_t i = 1; i < probs.size(); ++i) probs[i] = probs[i - 1] + totalProbs[i]; }
        uint32_t t = 0;
        while (n-- > 0) {
            const double u = S.rng.main_real(mk(ST_ALLOC_TOP, 0, 0, t++, 0));
            unsigned j = ctr ? first_le(probs.data(), (unsigned)probs.size(), u) : rand_indx(probs.data(), probs.size(), u);
            quota[j] += 1;
        }
    }
    parallel_blocks(nch, th, 64, [&](size_t, size_t lo, size_t hi) {              // batchSampling, one task per chunk, FIFO
        std::vector<double> cdf, q; Rng rng = S.rng;
        for (size_t c = lo; c < hi; ++c) {
            if (!quota[c]) continue;
            chunk_cdf(c, cdf, q);
            for (unsigned t = 0; t < quota[c]; ++t) {
                const double u = rng.real(mk(ST_ALLOC_CHUNK, 0, c, t >> 2, (int)(t & 3)));   // [REMAP] draw t = word t & 3 of block t >> 2
                unsigned j = ctr ? first_le(cdf.data(), (unsigned)cdf.size(), u) : rand_indx(cdf.data(), cdf.size(), u);
                readNumbers[c * chunk + j] += 1;
            }
        }
    });
    if (S.prm.paired) { int k = 1; for (size_t i = 0; i < ac; ++i) if (readNumbers[i] % 2 == 1) { readNumbers[i] += k; k *= -1; } }
}

void set_read_counts(Sim& S, long reads) {
    std::vector<double> w;
    compute_weights(S, w);
    if (S.prm.shard_count <= 1) { allocate_reads(S, w, reads, S.readNumbers); return; }
    // ---- sharded: assemble the GLOBAL weight vector in the reference's list order, allocate identically on every shard
    const int R = S.prm.shard_count, NSEG = 5 * 6;
    std::vector<uint64_t> segc((size_t)R * NSEG, 0);                                 // counts[r][c][p]
    for (auto& sg : S.full_segs) segc[(size_t)S.prm.shard_rank * NSEG + sg.c * 6 + sg.p] = sg.count;
    S.allreduce(segc.data(), segc.size());
    uint64_t maxn = 0; std::vector<uint64_t> nloc(R, 0);
    for (int r = 0; r < R; ++r) { for (int k = 0; k < NSEG; ++k) nloc[r] += segc[(size_t)r * NSEG + k]; maxn = std::max(maxn, nloc[r]); }
    std::vector<double> all((size_t)R * std::max<uint64_t>(maxn, 1)); std::vector<uint64_t> sizes(R, 0);
    if (!S.prm.allgatherv || S.prm.allgatherv(S.prm.coll_user, w.data(), w.size() * 8, all.data(), std::max<uint64_t>(maxn, 1) * 8, sizes.data()))
        fail("sharded run: allgatherv hook missing or failed");
    std::vector<uint64_t> loff(R, 0);                                                // running local offsets per shard
    std::vector<double> gw; std::vector<std::pair<uint64_t, uint64_t>> mine;          // (global offset, count) of my segments, in local order
    for (int c = 0; c < 5; ++c) for (int pb = 5; pb >= 0; --pb) for (int r = 0; r < R; ++r) {
        const uint64_t n = segc[(size_t)r * NSEG + c * 6 + pb];
        if (!n) continue;
        if (r == S.prm.shard_rank) mine.push_back({gw.size(), n});
        gw.insert(gw.end(), all.begin() + (size_t)r * std::max<uint64_t>(maxn, 1) + loff[r], all.begin() + (size_t)r * std::max<uint64_t>(maxn, 1) + loff[r] + n);
        loff[r] += n;
    }
    std::vector<unsigned> grn;
    allocate_reads(S, gw, reads, grn);
    S.readNumbers.clear(); S.gidx.clear();
    for (auto& m : mine) for (uint64_t k = 0; k < m.second; ++k) { S.readNumbers.push_back(grn[m.first + k]); S.gidx.push_back(m.first + k); }
    if (S.readNumbers.size() != S.fulls.a.size()) fail("sharded allocation: segment bookkeeping mismatch");
}

// ---- a11: Amplicon::yieldReads (Amplicon.cpp:402-565) -----------------------------------------
struct ReadOut { std::string f1, f2; unsigned long pairs = 0; std::string dump; };

// poff (checksum mode, below): planned pairs (SE: reads) before every amplicon; only the records whose slot -- poff[i] + their
// number among the amplicon's produced ones -- lies in [slot_lo, slot_hi) are made
void yield_reads_range(Sim& S, Rng& rng, size_t lo, size_t hi, ReadOut& out, bool dump, const uint64_t* poff = nullptr, uint64_t slot_lo = 0, uint64_t slot_hi = ~0ull) {
    const Profile& P = *S.prof; const int L = P.L; const bool paired = S.prm.paired;
    std::vector<uint8_t> scratch, seq, win(L);
    std::vector<char> ob(2 * L + 128), oq(2 * L + 128);
    char name[64];
    for (size_t i = lo; i < hi; ++i) {
        int n = (int)S.readNumbers[i];
        if (n == 0) continue;
        const Amp& a = S.fulls.a[i];
        const uint64_t gi = S.gidx.empty() ? i : S.gidx[i];                          // list index in the whole (unsharded) job
        full_sequence(S, a, scratch, seq);
        const int ampLen = (int)a.len;
        if (ampLen < L) continue;
        int fragCount = 0, failCount = 0; uint64_t made = 0;
        while (n > 0) {
            fragCount++;
            const uint32_t att = (uint32_t)(fragCount - 1);
            if (!paired) {
                if (poff) { const uint64_t slot = poff[i] + made++; if (slot < slot_lo || slot >= slot_hi) { n--; continue; } }
                long pos = (long)(0 + (double)(ampLen - L + 1 - 0) * rng.integer(mk(ST_PAIR, 0, a.uid, att, 1)));
                int m = predict(P, rng, &seq[pos], L, true, a.uid, att, ob.data(), oq.data());
                int k = snprintf(name, sizeof name, "@%d#%d\n", (int)gi, fragCount);
                out.f1.append(name, k); out.f1.append(ob.data(), m); out.f1.append("\n+\n"); out.f1.append(oq.data(), m); out.f1.push_back('\n');
                out.pairs++; n--;
                continue;
            }
            int isz = P.isizeAlphabet.empty() ? -1 :
                      P.isizeAlphabet[rand_indx(P.isizeCdf.data(), P.isizeCdf.size(), rng.real(mk(ST_PAIR, 0, a.uid, att, 0)))];
            if (isz < 0) fail("Error: unrecognized parameter name \"insertSize\"");      // Profile.cpp:1483-1485 (exit(1))
            if (isz < L || isz > ampLen) { failCount++; if (failCount > 1000) break; continue; }
            if (poff) { const uint64_t slot = poff[i] + made++; if (slot < slot_lo || slot >= slot_hi) { n -= 2; continue; } }
            long pos = (long)(0 + (double)(ampLen - isz + 1 - 0) * rng.integer(mk(ST_PAIR, 0, a.uid, att, 1)));
            int m1 = predict(P, rng, &seq[pos], L, true, a.uid, att, ob.data(), oq.data());
            int k = snprintf(name, sizeof name, "@%d#%d/1\n", (int)gi, fragCount);
            out.f1.append(name, k); out.f1.append(ob.data(), m1); out.f1.append("\n+\n"); out.f1.append(oq.data(), m1); out.f1.push_back('\n');
            for (int t = 0; t < L; ++t) win[t] = comp_code(seq[pos + isz - 1 - t]);    // revcomp of the far end
            int m2 = predict(P, rng, win.data(), L, false, a.uid, att, ob.data(), oq.data());
            k = snprintf(name, sizeof name, "@%d#%d/2\n", (int)gi, fragCount);
            out.f2.append(name, k); out.f2.append(ob.data(), m2); out.f2.append("\n+\n"); out.f2.append(oq.data(), m2); out.f2.push_back('\n');
            if (dump) { char b[128]; int q = snprintf(b, sizeof b, "%zu\t%d\t%ld\t%d\t%d\t%d\n", i, fragCount, pos, isz, m1, m2); out.dump.append(b, q); }
            out.pairs++; n -= 2;
        }
    }
}

void dump_amps(const std::string& path, const AmpList& L) {
    FILE* f = fopen(path.c_str(), "w"); if (!f) fail("cannot write " + path);
    for (size_t i = 0; i < L.a.size(); ++i) { const Amp& a = L.a[i];
        fprintf(f, "%zu\t%u\t%u\t%u\t%u\t%u\t%llu\t", i, a.parent, a.spos, a.len, a.gc, a.primers, (unsigned long long)a.uid);
        for (uint32_t e = 0; e < a.err_cnt; ++e) fprintf(f, "%s%u:%u", e ? "," : "", L.errs[a.err_off + e] >> 3, L.errs[a.err_off + e] & 7);
        fputc('\n', f); }
    fclose(f);
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int genreads(const scso_params& q) {
    Sim S; Params& p = S.prm;
    p.primers = q.primers; p.gamma = q.gamma; p.coverage = q.coverage; p.isize = q.isize; p.paired = q.paired != 0;
    p.threads = std::max(1, q.threads); p.counter = q.rng_mode == 1; p.seed = q.seed; p.fixed_time = q.fixed_time; p.verbose = q.verbose != 0;
    if (!p.counter) p.threads = 1;
    p.shard_rank = q.shard_rank; p.shard_count = std::max(1, q.shard_count); p.allreduce = q.allreduce; p.allgatherv = q.allgatherv; p.coll_user = q.coll_user;
    if (p.shard_count > 1 && !p.counter) fail("sharding needs --rng counter (the reference streams are sequential)");
    S.rng.counter = p.counter; S.rng.key[0] = (uint32_t)p.seed; S.rng.key[1] = (uint32_t)(p.seed >> 32); S.rng.ref = &S.streams;
    if (!p.counter) {
        srand((unsigned)p.fixed_time);                                                    // scssim.cpp:47
        unsigned seed = (unsigned)(p.fixed_time * 1000000000LL);                          // ThreadPool.cpp:41
        S.streams.w_real = std::mt19937(seed); S.streams.w_int = std::mt19937(seed);
    }
    double t0 = now_s();
    S.recs = load_fasta(q.input_fasta);
    if (p.verbose) fprintf(stderr, "\nReference sequence was loaded from file %s\n", q.input_fasta);
    S.prof = load_profile(q.profile, p.paired, p.isize);
    if (p.verbose) fprintf(stderr, "profile was loaded from file %s\n", q.profile);
    if (!p.counter) {                                                                     // Profile.cpp:1405-1411
        unsigned seed = (unsigned)(p.fixed_time * 1000000000LL);
        for (int l = 0; l < 101; ++l) { S.gcGen.emplace_back(seed); S.gcDist.emplace_back(S.prof->gcMeans[l], S.prof->gcStd); }
    }
    double t1 = now_s();
    split_to_frags(S);
    double t2 = now_s();
    amplify(S);
    double t3 = now_s();
    // Malbac::yieldReads (Malbac.cpp:410-460)
    unsigned long refLen = 0;
    for (auto& r : S.recs) { auto f = split(r.name, '_'); refLen += atoi(f.back().c_str()); }
    refLen /= 2;
    unsigned long reads = (unsigned long)(refLen * p.coverage / (long)S.prof->L);
    if (p.verbose) fprintf(stderr, "\nNumber of reads to generate: %lu\n", reads);
    set_read_counts(S, (long)reads);
    double t4 = now_s();
    std::string pre = q.output_prefix ? q.output_prefix : "";
    FILE *o1 = nullptr, *o2 = nullptr;
    if (!pre.empty()) {
        o1 = fopen((pre + (p.paired ? "_1.fq" : ".fq")).c_str(), "w");
        if (!o1) fail("Error: can not open fastq file to save results:\n" + pre);
        if (p.paired) { o2 = fopen((pre + "_2.fq").c_str(), "w"); if (!o2) fail("Error: can not open fastq file to save results:\n" + pre); }
    }
    if (p.verbose) fprintf(stderr, "\n*****Producing reads*****\n");
    // Profile.cpp:1483-1485: the first yieldInsertSize of a paired-end job on a model without insert-size alphabet exit(1)s; said here, on the
    // main thread (the same failure inside a worker of the range loop below would end in std::terminate)
    if (p.paired && S.prof->isizeAlphabet.empty())
        for (size_t i = 0; i < S.fulls.a.size(); ++i) if (S.readNumbers[i] > 0 && S.fulls.a[i].len >= (unsigned)S.prof->L) fail("Error: unrecognized parameter name \"insertSize\"");
    if (q.checksum_file) {
        // checksum mode (tools/whole_genome_golden.py: BASELINE configs[3] at full size, 197 GB of text): the FASTQ is not written
        // but summed batch by batch the way the library sums its batches (scs_set_batch_checksums, include/scssim_hip.h): batch b
        // = the records of the planned pairs [b N, (b + 1) N) -- an amplicon plans readNumbers / 2 pairs (SE: readNumbers
        // reads), the k-th one it produces takes its k-th slot, the slots of pairs it gives up on stay empty -- and its sum =
        // sum_i fmix64(w_i + (i + 1) phi) over the batch's text as little-endian 64-bit words (the last zero-padded).
        const uint64_t B = q.batch_pairs ? q.batch_pairs : (1ull << 23);
        const size_t ac = S.fulls.a.size();
        std::vector<uint64_t> poff(ac + 1, 0);
        for (size_t i = 0; i < ac; ++i) poff[i + 1] = poff[i] + (p.paired ? S.readNumbers[i] / 2 : S.readNumbers[i]);
        const uint64_t P = poff[ac], nb = (P + B - 1) / B;
        FILE* cf = fopen(q.checksum_file, "w"); if (!cf) fail(std::string("cannot write ") + q.checksum_file);
        fprintf(cf, "# frags %zu semis %zu fulls %zu planned %llu batch %llu\n", S.frags.size(), S.semis.a.size(), ac, (unsigned long long)P, (unsigned long long)B);
        auto text_sum = [&](const std::string& t) -> uint64_t {
            const size_t nw = (t.size() + 7) / 8; std::vector<uint64_t> part((size_t)p.threads * 4, 0);
            parallel_blocks(nw, p.threads, 1 << 16, [&](size_t b, size_t lo, size_t hi) {
                uint64_t acc = 0;
                for (size_t i = lo; i < hi; ++i) {
                    uint64_t w = 0; const size_t o = i * 8, n = std::min<size_t>(8, t.size() - o); memcpy(&w, t.data() + o, n);
                    uint64_t x = w + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull;
                    x ^= x >> 33; x *= 0xFF51AFD7ED558CCDull; x ^= x >> 33; x *= 0xC4CEB9FE1A85EC53ull; x ^= x >> 33;
                    acc += x;
                }
                part[b] = acc;
            });
            uint64_t sum = 0; for (uint64_t v : part) sum += v;
            return sum;
        };
        unsigned long pairs = 0; size_t i0 = 0; std::string x1, x2;
        std::vector<uint64_t> only;                                                  // checksum_batches "0,1,35": those batches alone (negative: from the end)
        if (q.checksum_batches) for (auto& f : split(q.checksum_batches, ',')) { long long v = atoll(f.c_str()); if (v < 0) v += (long long)nb; if (v >= 0 && (uint64_t)v < nb) only.push_back((uint64_t)v); }
        for (uint64_t b = 0; b < nb; ++b) {
            if (q.checksum_batches && std::find(only.begin(), only.end(), b) == only.end()) continue;
            const uint64_t lo = b * B, hi = std::min(P, lo + B);
            while (poff[i0 + 1] <= lo) ++i0;                                          // first amplicon with a slot in the batch
            size_t i1 = i0; while (i1 < ac && poff[i1] < hi) ++i1;
            size_t nbMax = (size_t)p.threads * 4; std::vector<ReadOut> outs(nbMax);
            parallel_blocks(i1 - i0, p.threads, 256, [&](size_t k, size_t a, size_t e) {
                Rng rng = S.rng;
                const size_t room = (size_t)(std::min(hi, poff[i0 + e]) - std::max(lo, poff[i0 + a])) * (size_t)(2 * S.prof->L + 40);
                outs[k].f1.reserve(room); if (p.paired) outs[k].f2.reserve(room);
                yield_reads_range(S, rng, i0 + a, i0 + e, outs[k], false, poff.data(), lo, hi);
            });
            x1.clear(); x2.clear(); unsigned long bp = 0;
            { size_t n1 = 0, n2 = 0; for (auto& o : outs) { n1 += o.f1.size(); n2 += o.f2.size(); } x1.reserve(n1); x2.reserve(n2); }
            for (auto& o : outs) { x1 += o.f1; x2 += o.f2; bp += o.pairs; o = ReadOut(); }
            pairs += bp;
            fprintf(cf, "%llu\t%016llx\t%016llx\t%zu\t%zu\t%lu\n", (unsigned long long)b, (unsigned long long)text_sum(x1), (unsigned long long)text_sum(x2), x1.size(), x2.size(), bp);
            fflush(cf);
            if (p.verbose) fprintf(stderr, "[oracle] batch %llu / %llu: %lu pairs, %.1f s\n", (unsigned long long)b + 1, (unsigned long long)nb, bp, now_s() - t4);
        }
        fclose(cf);
        double t5 = now_s();
        if (p.verbose) fprintf(stderr, "[oracle] frags=%zu semis=%zu fulls=%zu primers_left=%lu pairs=%lu | load %.2fs frag %.2fs amplify %.2fs alloc %.2fs readgen %.2fs\n",
                               S.frags.size(), S.semis.a.size(), S.fulls.a.size(), S.totalPrimers, pairs, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
        g_timings[0] = t1 - t0; g_timings[1] = t2 - t1; g_timings[2] = t3 - t2; g_timings[3] = t4 - t3; g_timings[4] = t5 - t4; g_timings[5] = (double)pairs;
        return 0;
    }
    const bool dump = q.dump_prefix != nullptr;
    FILE* dr = dump ? fopen((std::string(q.dump_prefix) + ".reads.tsv").c_str(), "w") : nullptr;
    unsigned long pairs = 0;
    const size_t ac = S.fulls.a.size(), step = (size_t)p.threads * 4 * 2048;
    for (size_t base = 0; base < ac; base += step) {                                       // bounded memory: wave of blocks, written in order
        size_t end = std::min(ac, base + step);
        size_t nbMax = (size_t)p.threads * 4; std::vector<ReadOut> outs(nbMax);
        parallel_blocks(end - base, p.threads, 256, [&](size_t b, size_t lo, size_t hi) { Rng rng = S.rng; yield_reads_range(S, rng, base + lo, base + hi, outs[b], dump); });
        for (auto& o : outs) { if (o1) fwrite(o.f1.data(), 1, o.f1.size(), o1); if (o2) fwrite(o.f2.data(), 1, o.f2.size(), o2);
                               if (dr) fwrite(o.dump.data(), 1, o.dump.size(), dr); pairs += o.pairs; }
    }
    if (o1) fclose(o1); if (o2) fclose(o2); if (dr) fclose(dr);
    double t5 = now_s();
    if (dump) {
        std::string d = q.dump_prefix;
        FILE* f = fopen((d + ".frags.tsv").c_str(), "w");
        for (size_t i = 0; i < S.frags.size(); ++i) fprintf(f, "%zu\t%d\t%ld\t%d\t%d\n", i, S.frags[i].rec, S.frags[i].start, S.frags[i].len, S.frags[i].strand);
        fclose(f);
        dump_amps(d + ".semis.tsv", S.semis); dump_amps(d + ".fulls.tsv", S.fulls);
        f = fopen((d + ".primers.tsv").c_str(), "w");                                       // primer type, attachments, stock left
        for (size_t i = 0; i < S.primerCount.size(); ++i) if (S.primerUsed[i]) fprintf(f, "%zu\t%ld\t%ld\n", i, S.primerUsed[i], S.primerCount[i]);
        fclose(f);
        f = fopen((d + ".readnum.tsv").c_str(), "w");
        for (size_t i = 0; i < S.readNumbers.size(); ++i) if (S.readNumbers[i]) fprintf(f, "%zu\t%u\n", i, S.readNumbers[i]);
        fclose(f);
    }
    if (p.verbose) {
        fprintf(stderr, "\nReads generation done!\n");
        fprintf(stderr, "[oracle] frags=%zu semis=%zu fulls=%zu primers_left=%lu pairs=%lu | load %.2fs frag %.2fs amplify %.2fs alloc %.2fs readgen %.2fs\n",
                S.frags.size(), S.semis.a.size(), S.fulls.a.size(), S.totalPrimers, pairs, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
    }
    g_timings[0] = t1 - t0; g_timings[1] = t2 - t1; g_timings[2] = t3 - t2; g_timings[3] = t4 - t3; g_timings[4] = t5 - t4; g_timings[5] = (double)pairs;
    return 0;
}

// ===========================================================================
// simuvars: Genome::loadAbers (lib/genome/Genome.cpp:35-165), SNPOnChr::readSNPs (lib/snp/snp.cpp:147-203),
// Genome::saveSequence (329-384), Genome::generateSegment (386-691), restated with the reference's own std::string
// operations.  The reference's simuvars branch never calls srand (src/scssim.cpp:33-38), so its glibc rand() runs from
// the default seed: the output is a pure function of the input files, and this restatement draws from rand() the same way.
// ===========================================================================
namespace simuvars {
struct Cnv { long spos, epos; float cn, mcn; };
struct Snv { long pos; char alt; bool het; };
struct Ins { long pos; std::string seq; bool het; };
struct Del { long pos; int len; bool het; };
struct Snp { long pos; char nuc; };
struct RawRec { std::string name, seq; };

std::string abbr_chr(std::string c) {                                              // MyDefine.cpp:310-323 / snp.cpp aberOfChr
    size_t i = c.find("chrom");
    if (i == std::string::npos) { i = c.find("chr"); if (i != std::string::npos) c = c.substr(i + 3); }
    else c = c.substr(i + 5);
    return c;
}
char snp_complement(char n) {                                                      // snp.cpp SNP::getComplement
    switch (n) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
                 case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c'; default: return 'N'; }
}
long rand_int(long start, long end) { return (long)(start + (end - start) * (rand() / (RAND_MAX + 1.0))); }   // MyDefine.cpp:290-292

std::vector<RawRec> load_raw_fasta(const std::string& path) {
    std::ifstream ifs(path);
    if (!ifs.is_open()) fail("could not open " + path);
    std::vector<RawRec> recs; std::string line;
    while (std::getline(ifs, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == ';') continue;
        if (line[0] == '>') { std::string nm = line.substr(1); size_t e = nm.find_first_of(" \t"); if (e != std::string::npos) nm = nm.substr(0, e); recs.push_back(RawRec{abbr_chr(nm), ""}); }
        else { if (recs.empty()) fail("FASTA sequence before header in " + path); recs.back().seq += line; }
    }
    if (recs.empty()) fail("ERROR: reference sequence cannot be empty!");
    return recs;
}

struct Vars {
    std::map<std::string, std::vector<Cnv>> cnvs; std::map<std::string, std::vector<Snv>> snvs;
    std::map<std::string, std::vector<Ins>> inss; std::map<std::string, std::vector<Del>> dels; std::map<std::string, std::vector<Snp>> snps;
};
void load_vars(const std::string& path, Vars& V) {                                 // Genome.cpp:35-165
    if (path.empty()) return;
    std::ifstream ifs(path);
    if (!ifs.is_open()) fail("can not open file " + path);
    std::string line; int ln = 0;
    auto bad = [&](const std::string& m) { fail("ERROR: " + m + " at line " + std::to_string(ln) + " in file " + path); };
    while (std::getline(ifs, line)) {
        ++ln;
        if (line.empty() || line[0] == '#') continue;
        auto f = split(line, '\t');
        const std::string t = f[0];
        auto typ = [&](const std::string& c) { if (c != "homo" && c != "het") bad("unrecognized variant type"); return c == "het"; };
        if (t == "c") {
            if (f.size() != 6) bad("wrong number of fields");
            float cn = atof(f[4].c_str()), mcn = atof(f[5].c_str());
            if (cn < mcn) bad("total copy number should be not lower than major copy number");
            if (cn - mcn > mcn) mcn = cn - mcn;
            V.cnvs[abbr_chr(f[1])].push_back(Cnv{atol(f[2].c_str()), atol(f[3].c_str()), cn, mcn});
        } else if (t == "s") {
            if (f.size() != 6) bad("wrong number of fields");
            if (f[3].at(0) == f[4].at(0)) bad("the mutated allele should be not same as the reference allele");
            V.snvs[abbr_chr(f[1])].push_back(Snv{atol(f[2].c_str()), f[4].at(0), typ(f[5])});
        } else if (t == "i") {
            if (f.size() != 5) bad("wrong number of fields");
            V.inss[abbr_chr(f[1])].push_back(Ins{atol(f[2].c_str()), f[3], typ(f[4])});
        } else if (t == "d") {
            if (f.size() != 5) bad("wrong number of fields");
            V.dels[abbr_chr(f[1])].push_back(Del{atol(f[2].c_str()), atoi(f[3].c_str()), typ(f[4])});
        } else bad("unrecognized aberraton type");
    }
}
void load_snps(const std::string& path, Vars& V) {                                 // snp.cpp:147-203 + SNP::SNP (12-36)
    if (path.empty()) return;
    FILE* f = fopen(path.c_str(), "r");
    if (!f) fail("can not open SNP file " + path);
    char buf[1000];
    while (fgets(buf, 1000, f)) {
        std::string line(buf);
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        auto e = split(line, '\t');
        if (e.size() != 6) continue;                                               // the reference warns and skips
        const std::string observed = e[3]; const char strand = e[4].empty() ? '+' : e[4][0]; char ref = e[5].empty() ? 'N' : e[5][0];
        auto ob = split(observed, '/');
        if (strand == '-') ref = snp_complement(ref);
        char nuc = (ob[0].at(0) == ref) ? ob[1].at(0) : ob[0].at(0);
        if (strand == '-') nuc = snp_complement(nuc);
        V.snps[abbr_chr(e[1])].push_back(Snp{atol(e[2].c_str()), nuc});
    }
    fclose(f);
}

// Genome::generateSegment (Genome.cpp:386-691), ploidy = 2
void generate_segment(std::vector<std::string>& seqs, const std::string& chrseq, const Vars& V, const std::string& chr, long s, long e, int CN, int mCN) {
    if (CN == 0) return;
    const int ploidy = 2;
    std::string refSeq = chrseq.substr((size_t)(s - 1), (size_t)(e - s + 1));
    for (auto& c : refSeq) c = (char)toupper(c);
    const size_t refSize = refSeq.size();
    std::vector<int> mIndx, seqReps; int i, j, k, n;
    auto has = [](const std::vector<int>& v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
    if (CN < ploidy) {
        for (i = 0; i < CN; i++) for (;;) { j = (int)rand_int(0, ploidy); if (!has(seqReps, j)) { seqReps.push_back(j); break; } }
        for (i = 0; i < mCN; i++) mIndx.push_back(seqReps[i]);
    } else {
        for (i = 0; i < ploidy; i++) seqReps.push_back(1);
        n = CN - ploidy; k = (int)rand_int(0, ploidy);
        for (i = n; i >= 0; i--) {
            if (seqReps[k] + i == mCN) { seqReps[k] += i; mIndx.push_back(k); break; }
            else if (seqReps[k] + i == CN - mCN) { seqReps[k] += i; for (j = 0; j < ploidy; j++) if (j != k) mIndx.push_back(j); break; }
        }
        if (i >= 0) { n -= i; while (n > 0) { j = (int)rand_int(0, ploidy); if (j != k) { seqReps[j]++; n--; } } }
        else { while (n > 0) { j = (int)rand_int(0, ploidy); seqReps[j]++; n--; } for (i = 0; i < ploidy; i++) mIndx.push_back(i); }
    }
    std::vector<std::string> seg;
    if (CN < ploidy) for (i = 0; i < ploidy; i++) seg.push_back(has(seqReps, i) ? refSeq : std::string());
    else for (i = 0; i < ploidy; i++) { std::string t; for (j = 0; j < seqReps[i]; j++) t += refSeq; seg.push_back(t); }
    auto get = [](auto& m, const std::string& c) -> decltype(m.begin()->second)& { static decltype(m.begin()->second) empty; auto it = m.find(c); return it == m.end() ? empty : it->second; };
    auto skip = [&](int kk, int hap) { const bool in = has(mIndx, hap); return (kk == 0 && !in) || (kk == 1 && in); };
    k = 0;                                                                         // SNPs: alternately on the major / the other haplotypes
    for (const Snp& sp : get(const_cast<Vars&>(V).snps, chr)) if (sp.pos >= s && sp.pos <= e) {
        const size_t sindx = (size_t)(sp.pos - s);
        for (j = 0; j < ploidy; j++) { if (skip(k, j)) continue; std::string& q = seg[j]; for (size_t t = 0; t < q.size() / refSize; t++) q[sindx + t * refSize] = sp.nuc; }
        k = (k + 1) % 2;
    }
    k = 0;
    for (const Snv& sv : get(const_cast<Vars&>(V).snvs, chr)) if (sv.pos >= s && sv.pos <= e) {
        const size_t sindx = (size_t)(sv.pos - s);
        for (j = 0; j < ploidy; j++) { if (sv.het && skip(k, j)) continue; std::string& q = seg[j]; for (size_t t = 0; t < q.size() / refSize; t++) q[sindx + t * refSize] = sv.alt; }
        if (sv.het) k = (k + 1) % 2;
    }
    std::map<int, std::map<int, int>> insPer, delPer; std::vector<int> insLens(ploidy, 0), delLens(ploidy, 0);
    k = 0;
    for (const Ins& in : get(const_cast<Vars&>(V).inss, chr)) if (in.pos >= s && in.pos <= e) {
        const int sindx = (int)(in.pos - s);
        for (j = 0; j < ploidy; j++) {
            if (in.het && skip(k, j)) continue;
            int offset = 0; auto& done = insPer[j];
            for (auto& m : done) if (m.first <= sindx) offset += m.second;
            std::string& q = seg[j]; n = (int)(q.size() / (refSize + insLens[j])); const int len = (int)in.seq.size();
            for (int t = 0; t < n; t++) q.insert((size_t)(sindx + offset) + (size_t)t * (refSize + insLens[j] + len), in.seq);
            insLens[j] += len; done.insert(std::make_pair(sindx, len));
        }
        if (in.het) k = (k + 1) % 2;
    }
    for (const Del& dl : get(const_cast<Vars&>(V).dels, chr)) if (dl.pos >= s && dl.pos <= e) {   // k carries over from the insertions (Genome.cpp:613,658)
        const int sindx = (int)(dl.pos - s);
        for (j = 0; j < ploidy; j++) {
            if (dl.het && skip(k, j)) continue;
            int offset = 0;
            for (auto& m : insPer[j]) if (m.first <= sindx) offset += m.second;
            auto& done = delPer[j];
            for (auto& m : done) if (m.first <= sindx) offset -= m.second;
            if (sindx + offset < 0) continue;
            std::string& q = seg[j]; n = (int)(q.size() / (refSize + insLens[j] - delLens[j]));
            for (int t = 0; t < n; t++) q.erase((size_t)(sindx + offset) + (size_t)t * (refSize + insLens[j] - delLens[j] - dl.len), (size_t)dl.len);
            delLens[j] += dl.len; done.insert(std::make_pair(sindx, dl.len));
        }
        if (dl.het) k = (k + 1) % 2;
    }
    for (i = 0; i < ploidy; i++) { for (auto& c : seg[i]) c = (char)toupper(c); seqs[i] += seg[i]; }
}

int run(const std::string& ref, const std::string& snp, const std::string& var, const std::string& out) {   // Genome::saveSequence (329-384)
    Vars V; load_vars(var, V); load_snps(snp, V);
    std::vector<RawRec> recs = load_raw_fasta(ref);
    FILE* o = fopen(out.c_str(), "w");
    if (!o) fail("can not open file " + out);
    const int ploidy = 2, mCN = 1;
    for (auto& r : recs) {
        const long L = (long)r.seq.size(); long segStart = 1;
        std::vector<std::string> hs(ploidy);
        auto it = V.cnvs.find(r.name);
        if (it != V.cnvs.end()) for (Cnv cv : it->second) {
            if (segStart > L) break;
            cv.epos = std::min(cv.epos, L);
            if (segStart < cv.spos) generate_segment(hs, r.seq, V, r.name, segStart, cv.spos - 1, ploidy, mCN);
            generate_segment(hs, r.seq, V, r.name, cv.spos, cv.epos, (int)cv.cn, (int)cv.mcn);
            segStart = cv.epos + 1;
        }
        if (segStart <= L) generate_segment(hs, r.seq, V, r.name, segStart, L, ploidy, mCN);
        for (int j = 0; j < ploidy; j++) {
            fprintf(o, ">%s_%d_%ld\n", r.name.c_str(), j + 1, L);
            for (size_t x = 0; x < hs[j].size(); x += 100) { fwrite(hs[j].data() + x, 1, std::min<size_t>(100, hs[j].size() - x), o); fputc('\n', o); }
        }
    }
    fclose(o);
    return 0;
}
}  // namespace simuvars

}  // namespace

extern "C" {

void scso_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox(ctr, key, out); }
double scso_det_log(double x) { return det_log(x); }
double scso_det_exp(double x) { return det_exp(x); }
const char* scso_last_error(void) { return g_err.c_str(); }
void scso_last_timings(double out[6]) { for (int i = 0; i < 6; ++i) out[i] = g_timings[i]; }

void scso_default_params(scso_params* p) {
    memset(p, 0, sizeof *p);
    p->primers = 100000; p->gamma = 1e-9; p->coverage = 5; p->isize = 260; p->paired = 1; p->threads = 1;
    p->rng_mode = 1; p->seed = 1; p->fixed_time = 1234567890LL; p->verbose = 1; p->shard_rank = 0; p->shard_count = 1;
}

int scso_genreads(const scso_params* p) {
    try { return genreads(*p); }
    catch (const std::exception& e) { g_err = e.what(); return 1; }
}

void* scso_profile_load(const char* path, int paired, int isize) {
    try { return load_profile(path, paired != 0, isize); }
    catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void scso_profile_free(void* h) { delete (Profile*)h; }
int scso_profile_read_length(void* h) { return ((Profile*)h)->L; }
int scso_profile_kmer_count(void* h) { return ((Profile*)h)->kmerCount; }
size_t scso_profile_table(void* h, int which, const double** data) {
    Profile* P = (Profile*)h;
    const std::vector<double>* v = nullptr;
    switch (which) {
        case 0: v = &P->subs1; break; case 1: v = &P->subs2; break; case 2: v = &P->qual; break;
        case 3: v = &P->insCdf; break; case 4: v = &P->delCdf; break; case 5: v = &P->isizeCdf; break;
        case 6: *data = P->gcMeans; return 101;
        default: *data = nullptr; return 0;
    }
    *data = v->data(); return v->size();
}
void scso_profile_scalars(void* h, double out[8]) {
    Profile* P = (Profile*)h;
    out[0] = P->insertRate; out[1] = P->delRate; out[2] = P->stdISize; out[3] = P->gcStd;
    out[4] = P->isizeAlphabet.empty() ? -1 : P->isizeAlphabet[0]; out[5] = (double)P->isizeAlphabet.size();
    out[6] = P->haveCdf2 ? 1 : 0; out[7] = P->bins;
}
int scso_predict_counter(void* h, const uint8_t* window, int n, int is_read1, uint64_t seed, uint64_t uid,
                         uint32_t attempt, char* out_bases, char* out_quals) {
    try {
        Rng rng; rng.counter = true; rng.key[0] = (uint32_t)seed; rng.key[1] = (uint32_t)(seed >> 32);
        return predict(*(Profile*)h, rng, window, n, is_read1 != 0, uid, attempt, out_bases, out_quals);
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// predict() on one window drawing from the REFERENCE's streams (two mt19937 copies of one seed, ThreadPool.cpp:41-47,203-212): a
// batch of windows in one call, the streams running on from read to read as they do in a worker thread.  For the statistical
// comparison of the counter-mode remaps with the reference's draws (tests/test_oracle_stats.py); lens[i] = n' of read i,
// outputs at stride 2 n + 64.
int scso_predict_ref_batch(void* h, const uint8_t* windows, int n, int count, const uint8_t* is_read1, unsigned seed,
                           char* out_bases, char* out_quals, int* lens) {
    try {
        RefStreams st; st.w_real = std::mt19937(seed); st.w_int = std::mt19937(seed);
        Rng rng; rng.counter = false; rng.ref = &st;
        const size_t stride = 2 * (size_t)n + 64;
        for (int i = 0; i < count; ++i)
            lens[i] = predict(*(Profile*)h, rng, windows + (size_t)i * n, n, is_read1[i] != 0, 0, 0, out_bases + (size_t)i * stride, out_quals + (size_t)i * stride);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// the same batch in counter mode (uid = first_uid + i, attempt 0)
int scso_predict_counter_batch(void* h, const uint8_t* windows, int n, int count, const uint8_t* is_read1, uint64_t seed, uint64_t first_uid,
                               char* out_bases, char* out_quals, int* lens) {
    try {
        Rng rng; rng.counter = true; rng.key[0] = (uint32_t)seed; rng.key[1] = (uint32_t)(seed >> 32);
        const size_t stride = 2 * (size_t)n + 64;
        for (int i = 0; i < count; ++i)
            lens[i] = predict(*(Profile*)h, rng, windows + (size_t)i * n, n, is_read1[i] != 0, first_uid + (uint64_t)i, 0, out_bases + (size_t)i * stride, out_quals + (size_t)i * stride);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// ---- probes of two more [REMAP] branches, for tests/test_oracle_stats.py: the same unit through the reference's draws and through counter mode
// The attach tries of ONE primer on an EMPTY template of `length` bases (nothing attached, every primer type in stock: only the "does not
// fit" failures of Fragment.cpp:76-82 / Amplicon.cpp:179-185 are left), `count` times: tries[i] = tries used (51 = gave up), spos / alen
// of the try that fit.  counter = 0: try by try from two mt19937 streams (int stream: position, real stream: length); counter = 1: the
// geometric gap + the closed-form fitting try of amplify_template, primer i = Philox block i of ST_ATTACH.
int scso_attach_tries_batch(unsigned length, int amin, int amax, int count, int counter, uint64_t seed, uint32_t* tries, uint32_t* spos_out, uint32_t* alen_out) {
    try {
        if ((int)length < amin + 27) fail("scso_attach_tries_batch: template too short");
        // (a worker's two generators start from one seed, ThreadPool.cpp:41-47, and run apart as soon as the per-base error draws consume the
        // real stream alone; in lockstep position and length of a try would be one draw -- the probe starts them a million draws apart)
        RefStreams st; st.w_real = std::mt19937((unsigned)seed); st.w_int = std::mt19937((unsigned)seed); st.w_real.discard(1000003);
        Rng rng; rng.counter = counter != 0; rng.ref = &st; rng.key[0] = (uint32_t)seed; rng.key[1] = (uint32_t)(seed >> 32);
        const AttachFit fit = attach_fit_count(length, (uint32_t)amin, (uint32_t)amax);
        const double qfail = 1.0 - (double)fit.N / ((double)(length - 27) * (double)fit.W) / test_bias("attach");
        for (int i = 0; i < count; ++i) {
            int tryTimes = 0; unsigned spos = 0, alen = 0;
            if (counter) {
                Xoshiro xt; { uint32_t c[4] = {(uint32_t)i, 77u, 0u, ST_ATTACH}, o[4]; philox(c, rng.key, o); xt.seed(o); }
                tryTimes += (int)attach_gap(((double)xt.next() + 0.5) / 4294967296.0, qfail) + 1;
                if (tryTimes <= 50) {
                    const uint64_t hi = xt.next(), lo = xt.next(), x64 = (hi << 32) | lo;
                    attach_fit_decode(fit, length, (uint32_t)amin, (uint32_t)(((unsigned __int128)x64 * fit.N) >> 64), spos, alen);
                }
            } else do {
                const double u1 = rng.integer(Key{}), u2 = rng.real(Key{});
                spos = (unsigned)(long)(27 + ((long)length - 27) * u1);
                alen = (unsigned)(amin + (double)(amax + 1 - amin) * u2);
                tryTimes++;
                if (tryTimes > 50) break;
            } while (spos + alen > length);
            const bool gave_up = tryTimes > 50;
            tries[i] = gave_up ? 51u : (uint32_t)tryTimes; spos_out[i] = gave_up ? 0u : spos; alen_out[i] = gave_up ? 0u : alen;
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// Profile::getGCFactor (Profile.cpp:1503-1513) for one GC percentage, `count` times: counter = 0 from the reference's engine of that
// percentage (minstd_rand0 + libstdc++ normal_distribution, Profile.cpp:1405-1411), counter = 1 the keyed polar method (uid = i).
int scso_gc_factor_batch(void* h, int gc, int count, int counter, uint64_t seed, double* out) {
    try {
        Sim S; S.prof = (Profile*)h; S.prm.counter = counter != 0; S.rng.counter = counter != 0;
        S.rng.key[0] = (uint32_t)seed; S.rng.key[1] = (uint32_t)(seed >> 32); S.bias_gc = test_bias("gc");
        struct Lend { Sim& s; ~Lend() { s.prof = nullptr; } } lend{S};               // (the profile stays the caller's)
        for (int l = 0; l < 101; ++l) { S.gcGen.emplace_back((unsigned)seed); S.gcDist.emplace_back(S.prof->gcMeans[l], S.prof->gcStd); }
        for (int i = 0; i < count; ++i) out[i] = gc_factor(S, gc, (uint64_t)i);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"

#ifdef SCS_ORACLE_MAIN
// scs_oracle genreads: the reference's genreads CLI (src/scssim.cpp:285-404) + oracle extras
//   --rng ref|counter   --seed N   --fixed-time T   --dump PREFIX   -q (quiet)
//   --checksums FILE [--batch-pairs N] [--checksum-batches 0,1,-1]   no FASTQ files: the text's checksum per batch of N planned pairs (default 2^23) and mate
#include <getopt.h>
int main(int argc, char** argv) {
    if (argc >= 2 && strcmp(argv[1], "simuvars") == 0) {                          // the reference's simuvars CLI (src/scssim.cpp:108-170)
        std::string ref, snp, var, out; static option so[] = {{"ref", 1, 0, 'r'}, {"snp", 1, 0, 's'}, {"var", 1, 0, 'v'}, {"output", 1, 0, 'o'}, {0, 0, 0, 0}};
        int c; argc--; argv++;
        while ((c = getopt_long(argc, argv, "r:s:v:o:", so, nullptr)) != -1) switch (c) { case 'r': ref = optarg; break; case 's': snp = optarg; break; case 'v': var = optarg; break; case 'o': out = optarg; break; default: return 1; }
        if (ref.empty() || out.empty()) { fprintf(stderr, "Error: -r and -o are required\n"); return 1; }
        try { return simuvars::run(ref, snp, var, out); } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); return 1; }
    }
    if (argc < 2 || strcmp(argv[1], "genreads") != 0) { fprintf(stderr, "usage: scs_oracle genreads -i simu.fa -m model.profile -o prefix [options] | scs_oracle simuvars -r ref.fa [-s snp.txt] [-v vars.txt] -o simu.fa\n"); return 1; }
    scso_params p; scso_default_params(&p);
    static option lo[] = {{"input", 1, 0, 'i'}, {"primers", 1, 0, 'p'}, {"gamma", 1, 0, 'r'}, {"model", 1, 0, 'm'}, {"layout", 1, 0, 'l'},
                          {"coverage", 1, 0, 'c'}, {"isize", 1, 0, 's'}, {"threads", 1, 0, 't'}, {"output", 1, 0, 'o'},
                          {"rng", 1, 0, 1000}, {"seed", 1, 0, 1001}, {"fixed-time", 1, 0, 1002}, {"dump", 1, 0, 1003},
                          {"checksums", 1, 0, 1004}, {"batch-pairs", 1, 0, 1005}, {"checksum-batches", 1, 0, 1006}, {0, 0, 0, 0}};
    int c; argc--; argv++;
    while ((c = getopt_long(argc, argv, "i:p:r:m:l:c:s:t:o:q", lo, nullptr)) != -1) switch (c) {
        case 'i': p.input_fasta = optarg; break; case 'p': p.primers = atol(optarg); break; case 'r': p.gamma = atof(optarg); break;
        case 'm': p.profile = optarg; break; case 'l': p.paired = strcmp(optarg, "SE") != 0; break; case 'c': p.coverage = atof(optarg); break;
        case 's': p.isize = atoi(optarg); break; case 't': p.threads = atoi(optarg); break; case 'o': p.output_prefix = optarg; break;
        case 'q': p.verbose = 0; break;
        case 1000: p.rng_mode = strcmp(optarg, "ref") == 0 ? 0 : 1; break; case 1001: p.seed = strtoull(optarg, 0, 10); break;
        case 1002: p.fixed_time = atoll(optarg); break; case 1003: p.dump_prefix = optarg; break;
        case 1004: p.checksum_file = optarg; break; case 1005: p.batch_pairs = strtoull(optarg, 0, 10); break; case 1006: p.checksum_batches = optarg; break;
        default: return 1;
    }
    if (!p.input_fasta || !p.profile || (!p.output_prefix && !p.checksum_file)) { fprintf(stderr, "Error: -i, -m and -o are required\n"); return 1; }
    int rc = scso_genreads(&p);
    if (rc) fprintf(stderr, "%s\n", scso_last_error());
    return rc;
}
#endif
