// scs_simuvars.cpp -- host planner of `simuvars` (see scs_simuvars.h).
#include "scs_simuvars.h"

#include <stdlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace scs {
namespace {

struct Cnv { long spos, epos; float cn, mcn; };
struct Snv { long pos; char alt; bool het; };
struct Ins { long pos; std::string seq; bool het; };
struct Del { long pos; int len; bool het; };
struct Snp { long pos; char nuc; };

[[noreturn]] void fail(const std::string& m) { throw std::runtime_error(m); }

std::vector<std::string> split_tab(const std::string& s) {
    std::vector<std::string> f; size_t b = 0;
    for (;;) { const size_t e = s.find('\t', b); if (e == std::string::npos) { f.push_back(s.substr(b)); break; } f.push_back(s.substr(b, e - b)); b = e + 1; }
    return f;
}
std::string abbr_chr(std::string c) {                                              // abbrOfChr, lib/mydefine/MyDefine.cpp:310-323
    size_t i = c.find("chrom");
    if (i == std::string::npos) { i = c.find("chr"); if (i != std::string::npos) c = c.substr(i + 3); }
    else c = c.substr(i + 5);
    return c;
}
char snp_complement(char n) {                                                      // SNP::getComplement, lib/snp/snp.cpp:88-102
    switch (n) { case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
                 case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c'; default: return 'N'; }
}
uint32_t upper(char c) { return (uint32_t)(unsigned char)((c >= 'a' && c <= 'z') ? c - 32 : c); }

// The reference draws from glibc rand() and its simuvars branch never calls srand (src/scssim.cpp:33-38): the default
// seed.  random_r over a private state initialised with seed 1 is that generator (rand() is random() in glibc), without
// touching the process-wide state of the host application.
struct GlibcRand {
    struct random_data rd; char state[128];
    GlibcRand() { memset(&rd, 0, sizeof rd); memset(state, 0, sizeof state); initstate_r(1u, state, sizeof state, &rd); }
    long integer(long start, long end) {                                           // randomInteger, MyDefine.cpp:290-292
        int32_t r = 0; random_r(&rd, &r);
        return (long)(start + (end - start) * (r / (RAND_MAX + 1.0)));
    }
};

struct Vars {
    std::map<std::string, std::vector<Cnv>> cnvs; std::map<std::string, std::vector<Snv>> snvs;
    std::map<std::string, std::vector<Ins>> inss; std::map<std::string, std::vector<Del>> dels; std::map<std::string, std::vector<Snp>> snps;
};

void load_vars(const std::string& path, Vars& V, SvPlan& P, bool verbose) {       // Genome::loadAbers, Genome.cpp:35-165
    if (path.empty()) return;
    std::ifstream ifs(path);
    if (!ifs.is_open()) fail("can not open file " + path);
    std::string line; int ln = 0;
    while (std::getline(ifs, line)) {
        ++ln;
        if (line.empty() || line[0] == '#') continue;
        const std::vector<std::string> f = split_tab(line);
        auto bad = [&](const std::string& m) { fail("ERROR: " + m + " " + std::to_string(ln) + " in file " + path + "\n" + line); };
        auto nfields = [&](size_t n) { if (f.size() != n) fail("ERROR: line " + std::to_string(ln) + " has wrong number of fields in file " + path + "\n" + line); };
        auto het = [&](const std::string& c, const char* what) { if (c != "homo" && c != "het") bad(std::string("unrecognized ") + what + " type at line"); return c == "het"; };
        const std::string& t = f[0];
        if (t == "c") {
            nfields(6);
            float cn = (float)atof(f[4].c_str()), mcn = (float)atof(f[5].c_str());
            if (cn < mcn) bad("total copy number should be not lower than major copy number at line");
            if (cn - mcn > mcn) mcn = cn - mcn;
            V.cnvs[abbr_chr(f[1])].push_back(Cnv{atol(f[2].c_str()), atol(f[3].c_str()), cn, mcn}); ++P.n_cnv;
        } else if (t == "s") {
            nfields(6);
            if (f[3].empty() || f[4].empty()) bad("empty allele at line");
            if (f[3][0] == f[4][0]) bad("the mutated allele should be not same as the reference allele at line");
            V.snvs[abbr_chr(f[1])].push_back(Snv{atol(f[2].c_str()), f[4][0], het(f[5], "SNV")}); ++P.n_snv;
        } else if (t == "i") {
            nfields(5);
            V.inss[abbr_chr(f[1])].push_back(Ins{atol(f[2].c_str()), f[3], het(f[4], "insert")}); ++P.n_ins;
        } else if (t == "d") {
            nfields(5);
            V.dels[abbr_chr(f[1])].push_back(Del{atol(f[2].c_str()), atoi(f[3].c_str()), het(f[4], "deletion")}); ++P.n_del;
        } else bad("unrecognized aberraton type at line");
    }
    if (verbose) fprintf(stderr, "\nDetails of the aberrations loaded from file %s are as follows:\nCNV: %d\nSNV: %d\nInsert: %d\nDeletion: %d\n", path.c_str(), P.n_cnv, P.n_snv, P.n_ins, P.n_del);
}

void load_snps(const std::string& path, Vars& V, SvPlan& P, bool verbose) {        // SNPOnChr::readSNPs + SNP::SNP, snp.cpp:12-36,147-203
    if (path.empty()) return;
    FILE* f = fopen(path.c_str(), "r");
    if (!f) fail("can not open SNP file " + path);
    char buf[1000]; long ln = 0;
    while (fgets(buf, 1000, f)) {
        ++ln;
        std::string line(buf);
        while (!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
        const std::vector<std::string> e = split_tab(line);
        if (e.size() != 6) { fprintf(stderr, "Warning: malformed snp file %s, there should be 6 fields @line %ld\n%s\n", path.c_str(), ln, buf); continue; }
        char ref = e[5].empty() ? 'N' : e[5][0]; const char strand = e[4].empty() ? '+' : e[4][0];
        const size_t slash = e[3].find('/');
        const std::string a0 = e[3].substr(0, slash), a1 = slash == std::string::npos ? std::string() : e[3].substr(slash + 1);
        if (a0.empty() || a1.empty()) fail("malformed observed alleles in SNP file " + path + " @line " + std::to_string(ln));
        if (strand == '-') ref = snp_complement(ref);
        char nuc = a0[0] == ref ? a1[0] : a0[0];
        if (strand == '-') nuc = snp_complement(nuc);
        V.snps[abbr_chr(e[1])].push_back(Snp{atol(e[2].c_str()), nuc}); ++P.n_snp;
    }
    fclose(f);
    if (verbose) fprintf(stderr, "\n%ld SNPs to simulate were loaded from file %s\n", P.n_snp, path.c_str());
}

// ---- the rope of one (segment, haplotype): what generateSegment keeps in a std::string
struct RopePiece { uint64_t src; uint32_t len; bool lit; };
struct Rope {
    std::vector<RopePiece> p; uint64_t size = 0;
    size_t split_at(uint64_t x) {                                                  // index of the piece that starts at x (pieces are cut so that one does)
        uint64_t acc = 0;
        for (size_t i = 0; i < p.size(); ++i) {
            if (acc == x) return i;
            if (x < acc + p[i].len) {
                const uint32_t left = (uint32_t)(x - acc);
                RopePiece r = p[i]; r.src += left; r.len -= left; p[i].len = left;
                p.insert(p.begin() + (long)i + 1, r);
                return i + 1;
            }
            acc += p[i].len;
        }
        return p.size();                                                           // x == size
    }
    void insert(uint64_t x, uint64_t lit_off, uint32_t len) {                      // std::string::insert(x, seq): x > size() throws
        if (x > size) fail("ERROR: an insertion falls outside its segment (std::string::insert out of range in the reference)");
        if (!len) return;
        const size_t i = split_at(x);
        p.insert(p.begin() + (long)i, RopePiece{lit_off, len, true}); size += len;
    }
    void erase(uint64_t x, uint64_t n) {                                           // std::string::erase(x, n): x > size() throws, n is clamped
        if (x > size) fail("ERROR: a deletion falls outside its segment (std::string::erase out of range in the reference)");
        n = std::min<uint64_t>(n, size - x);
        if (!n) return;
        const size_t a = split_at(x), b = split_at(x + n);
        p.erase(p.begin() + (long)a, p.begin() + (long)b); size -= n;
    }
};

struct Builder {
    SvPlan& P; uint64_t hap_len[2] = {0, 0};
    std::vector<SvPiece> hp[2]; std::vector<SvSubst> hs[2];                       // per haplotype of the current chromosome, dst relative to the haplotype
    explicit Builder(SvPlan& p) : P(p) {}
};

// Genome::generateSegment (Genome.cpp:386-691) for ploidy 2; U0 = offset of the chromosome in the reference buffer
void generate_segment(Builder& B, GlibcRand& rng, const Vars& V, const std::string& chr, uint64_t U0, uint64_t chr_len, long s, long e, int CN, int mCN) {
    if (CN == 0) return;
    const int ploidy = 2;
    if (e < s) return;                                                             // getSubSequence of a non-positive length: empty -> nothing appended
    if (s < 1 || (uint64_t)e > chr_len) fail("ERROR: a copy-number interval of chromosome " + chr + " lies outside the reference sequence");
    const uint64_t refSize = (uint64_t)(e - s + 1), unit = U0 + (uint64_t)(s - 1);
    std::vector<int> mIndx, seqReps; int i, j, k, n;
    auto has = [](const std::vector<int>& v, int x) { return std::find(v.begin(), v.end(), x) != v.end(); };
    if (CN < ploidy) {
        for (i = 0; i < CN; i++) for (;;) { j = (int)rng.integer(0, ploidy); if (!has(seqReps, j)) { seqReps.push_back(j); break; } }
        for (i = 0; i < mCN; i++) mIndx.push_back(seqReps[(size_t)i]);
    } else {
        for (i = 0; i < ploidy; i++) seqReps.push_back(1);
        n = CN - ploidy; k = (int)rng.integer(0, ploidy);
        for (i = n; i >= 0; i--) {
            if (seqReps[(size_t)k] + i == mCN) { seqReps[(size_t)k] += i; mIndx.push_back(k); break; }
            else if (seqReps[(size_t)k] + i == CN - mCN) { seqReps[(size_t)k] += i; for (j = 0; j < ploidy; j++) if (j != k) mIndx.push_back(j); break; }
        }
        if (i >= 0) { n -= i; while (n > 0) { j = (int)rng.integer(0, ploidy); if (j != k) { seqReps[(size_t)j]++; n--; } } }
        else { while (n > 0) { j = (int)rng.integer(0, ploidy); seqReps[(size_t)j]++; n--; } for (i = 0; i < ploidy; i++) mIndx.push_back(i); }
    }
    Rope rope[2];
    for (i = 0; i < ploidy; i++) {
        const int copies = CN < ploidy ? (has(seqReps, i) ? 1 : 0) : seqReps[(size_t)i];
        for (j = 0; j < copies; j++) rope[i].p.push_back(RopePiece{unit, (uint32_t)refSize, false});
        rope[i].size = (uint64_t)copies * refSize;
    }
    if (refSize > 0xFFFFFFFFull) fail("segment longer than 4 Gb");
    auto skip = [&](int kk, int hap) { const bool in = has(mIndx, hap); return (kk == 0 && !in) || (kk == 1 && in); };
    auto list = [&](const auto& m) -> const typename std::decay<decltype(m)>::type::mapped_type* { auto it = m.find(chr); return it == m.end() ? nullptr : &it->second; };
    // substitutions act on the un-indel'd copies (segSeq[sindx + t*refSize]): kept per haplotype in unit coordinates, the
    // last write to a position wins
    std::map<uint64_t, uint32_t> sub[2];
    k = 0;
    if (auto* L = list(V.snps)) for (const Snp& sp : *L) if (sp.pos >= s && sp.pos <= e) {
        for (j = 0; j < ploidy; j++) { if (skip(k, j)) continue; if (rope[j].size) sub[j][(uint64_t)(sp.pos - s)] = upper(sp.nuc); }
        k = (k + 1) % 2;
    }
    k = 0;
    if (auto* L = list(V.snvs)) for (const Snv& sv : *L) if (sv.pos >= s && sv.pos <= e) {
        for (j = 0; j < ploidy; j++) { if (sv.het && skip(k, j)) continue; if (rope[j].size) sub[j][(uint64_t)(sv.pos - s)] = upper(sv.alt); }
        if (sv.het) k = (k + 1) % 2;
    }
    std::map<int, int> insDone[2], delDone[2]; long insLens[2] = {0, 0}, delLens[2] = {0, 0};
    k = 0;
    if (auto* L = list(V.inss)) for (const Ins& in : *L) if (in.pos >= s && in.pos <= e) {
        const int sindx = (int)(in.pos - s); const long len = (long)in.seq.size();
        uint64_t lit_off = 0; bool have_lit = false;
        for (j = 0; j < ploidy; j++) {
            if (in.het && skip(k, j)) continue;
            long offset = 0;
            for (auto& m : insDone[j]) if (m.first <= sindx) offset += m.second;
            n = (int)(rope[j].size / (refSize + (uint64_t)insLens[j]));
            if (n > 0 && !have_lit) { lit_off = B.P.literals.size(); B.P.literals += in.seq; have_lit = true; }
            for (int t = 0; t < n; t++) rope[j].insert((uint64_t)(sindx + offset) + (uint64_t)t * (refSize + (uint64_t)insLens[j] + (uint64_t)len), lit_off, (uint32_t)len);
            insLens[j] += len; insDone[j].insert(std::make_pair(sindx, (int)len));
        }
        if (in.het) k = (k + 1) % 2;
    }
    if (auto* L = list(V.dels)) for (const Del& dl : *L) if (dl.pos >= s && dl.pos <= e) {   // k carries over from the insertions (Genome.cpp:613,658)
        const int sindx = (int)(dl.pos - s);
        for (j = 0; j < ploidy; j++) {
            if (dl.het && skip(k, j)) continue;
            long offset = 0;
            for (auto& m : insDone[j]) if (m.first <= sindx) offset += m.second;
            for (auto& m : delDone[j]) if (m.first <= sindx) offset -= m.second;
            if (sindx + offset < 0) continue;
            const long copy = (long)refSize + insLens[j] - delLens[j];
            if (copy <= 0) fail("ERROR: deletions consume a whole segment copy (division by zero in the reference)");
            n = (int)(rope[j].size / (uint64_t)copy);
            for (int t = 0; t < n; t++) rope[j].erase((uint64_t)(sindx + offset) + (uint64_t)t * (uint64_t)(copy - dl.len), (uint64_t)std::max(0, dl.len));
            delLens[j] += dl.len; delDone[j].insert(std::make_pair(sindx, dl.len));
        }
        if (dl.het) k = (k + 1) % 2;
    }
    for (i = 0; i < ploidy; i++) {                                                 // sequences[i] += segSeqs[i]
        uint64_t dst = B.hap_len[i];
        for (const RopePiece& rp : rope[i].p) {
            if (!rp.len) continue;
            B.hp[i].push_back(SvPiece{dst, rp.src, rp.len, rp.lit ? 1u : 0u});
            if (!rp.lit && !sub[i].empty()) {                                      // the substitutions that fall into this stretch of the unit
                const uint64_t a = rp.src - unit;
                for (auto it = sub[i].lower_bound(a); it != sub[i].end() && it->first < a + rp.len; ++it) B.hs[i].push_back(SvSubst{dst + (it->first - a), it->second, 0});
            }
            dst += rp.len;
        }
        B.hap_len[i] = dst;
    }
}

}  // namespace

void simuvars_plan(const std::vector<SvChrom>& chroms, const std::string& snp_file, const std::string& var_file, bool verbose, SvPlan& P) {
    P = SvPlan();
    Vars V;
    load_vars(var_file, V, P, verbose);                                            // Genome::loadData: loadAbers, loadSNPs, loadRefSeq (Genome.cpp:18-25)
    load_snps(snp_file, V, P, verbose);
    GlibcRand rng;
    const int ploidy = 2, mCN = 1;                                                 // Config "ploidy" = 2 (Config.cpp:35); mCN = ceil(ploidy / 2)
    uint64_t out_off = 0;
    for (const SvChrom& c : chroms) {                                              // Genome::saveSequence, Genome.cpp:329-384
        Builder B(P);
        const long L = (long)c.len; long segStart = 1;
        auto it = V.cnvs.find(c.name);
        if (it != V.cnvs.end()) for (Cnv cv : it->second) {
            if (segStart > L) break;
            cv.epos = std::min(cv.epos, L);
            if (segStart < cv.spos) generate_segment(B, rng, V, c.name, c.off, c.len, segStart, cv.spos - 1, ploidy, mCN);
            generate_segment(B, rng, V, c.name, c.off, c.len, cv.spos, cv.epos, (int)cv.cn, (int)cv.mcn);
            segStart = cv.epos + 1;
        }
        if (segStart <= L) generate_segment(B, rng, V, c.name, c.off, c.len, segStart, L, ploidy, mCN);
        for (int h = 0; h < ploidy; ++h) {
            P.rec_names.push_back(c.name + "_" + std::to_string(h + 1) + "_" + std::to_string(L));
            P.rec_lens.push_back(B.hap_len[h]);
            for (SvPiece pc : B.hp[h]) { pc.dst += out_off; P.pieces.push_back(pc); }
            for (SvSubst sb : B.hs[h]) { sb.dst += out_off; P.substs.push_back(sb); }
            out_off += B.hap_len[h];
        }
    }
    P.total = out_off;
}

}  // namespace scs
