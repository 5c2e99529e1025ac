// scs_simuvars.h -- `scssim simuvars` on the data plane: the host plans, the GPU builds.
// Reference: Genome::loadAbers (lib/genome/Genome.cpp:35-165), SNPOnChr::readSNPs (lib/snp/snp.cpp:147-203),
// Genome::saveSequence (329-384), Genome::generateSegment (386-691).
// The planner walks the chromosomes and segments exactly as saveSequence does and applies every edit of
// generateSegment -- copy-number replication, SNP / SNV substitutions, insertions, deletions, in the reference's order
// and with its index arithmetic -- to a ROPE of pieces instead of a std::string: a piece is a range of the reference
// (resident in HBM) or of a small literal pool.  Nothing is copied on the host; the haplotype sequences are materialised
// on the device (k_sv_build / k_sv_subst) straight into the buffer that genreads stages its genome from.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace scs {

struct SvPiece { uint64_t dst; uint64_t src; uint32_t len; uint32_t lit; };      // dst: offset in the output; src: offset in the reference (lit = 0) or the literal pool (lit = 1)
struct SvSubst { uint64_t dst; uint32_t ch; uint32_t pad; };                      // output[dst] = ch (SNP / SNV alleles; upper-cased)
struct SvPlan {
    std::vector<std::string> rec_names;               // <chr>_<hap>_<reference length> (Genome.cpp:368), two per chromosome
    std::vector<uint64_t> rec_lens;                   // haplotype lengths, same order; their concatenation is the output
    std::vector<SvPiece> pieces;                      // ascending dst, covering the output exactly
    std::vector<SvSubst> substs;
    std::string literals;                             // inserted sequences as given in the variation file
    uint64_t total = 0;
    int n_cnv = 0, n_snv = 0, n_ins = 0, n_del = 0; long n_snp = 0;
};
struct SvChrom { std::string name; uint64_t off, len; };                          // reference records as staged: index name, offset in the device buffer

// Throws std::runtime_error with the reference's message where it has one.  ref_text: host view of one reference record is
// NOT needed: every decision of the planner depends on positions and lengths only (the bases are touched on the device).
void simuvars_plan(const std::vector<SvChrom>& chroms, const std::string& snp_file, const std::string& var_file, bool verbose, SvPlan& out);

}  // namespace scs
