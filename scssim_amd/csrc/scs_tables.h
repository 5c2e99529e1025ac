// scs_tables.h -- host side of the .profile model: parse, normalise, CDFs, exact integer thresholds.
// Mirrors Profile::train(file) = load + normParas(true) + initCDFs (reference lib/profile/Profile.cpp:
// 930-1234, 832-863/897-927, 1363-1430) and Matrix::normalize/cumsum (lib/matrix/Matrix.h:483-522).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace scs {

struct ProfileTables {
    int read_length = 0, bins = 0;
    double insert_rate = 0, del_rate = 0, std_isize = 0, gc_std = 0;
    double gc_means[101];
    bool have_cdf2 = false;              // read 2 has its own substitution table (PE and sigma_isize > 0)
    // CDFs in double, exactly as the reference holds them (ground truth; also the device slow path)
    std::vector<double> subs1, subs2;    // [84][bins][4]
    std::vector<double> qual;            // [16][bins][94]
    std::vector<double> ins_cdf, del_cdf, isize_cdf;
    int isize_min = 0;                   // iSizeAlphabet[k] = isize_min + k
    // Every CDF lookup in the reference is `r <= cdf[k]` with r = 2.2204e-16 + (1-2.2204e-16) * x/2^32
    // (MyDefine.cpp:274-282), monotone in the 32-bit draw x.  T[k] = #{x : r(x) <= cdf[k]} clamped to
    // 2^32-1, so the lookup is `x < T[k]`; exact for every x except x == 0xFFFFFFFF, which the kernels
    // route to the double tables.
    std::vector<uint32_t> subs1_t, subs2_t, qual_t, ins_t, del_t, isize_t;
    // [REMAP] quality symbols by the ALIAS method (Walker / Vose) instead of a CDF search.  Of the 2^32 draws, symbol k of a
    // row is hit by w_k = #{x : r(x) <= cdf[k]} - #{x : r(x) <= cdf[k-1]} of them in the reference's comparison (the last
    // symbol takes what is left, randIndx's `return ac-1`).  The alias row hits symbol k with EXACTLY w_k draws too: K =
    // qual_k columns (16, 64 or 128: the smallest that holds the most varied row), column j = x >> (32 - log2 K) owns
    // 2^32 / K draws, of which the lowest t_j go to the row's j-th drawable symbol and the rest to column alias_j's
    // (integer Vose construction: no rounding anywhere).  Row = K words (t_j << log2 K | alias_j) + K symbol bytes.
    int qual_k = 16;
    std::vector<uint32_t> qual_alias;    // [16*bins][qual_k + qual_k/4] words
    uint32_t t_insert = 0;               // p <= insertRate            (Profile.cpp:1557)
    uint32_t t_delete = 0;               // p <  delRate/(1-insertRate) (Profile.cpp:1565-1566)
    // [REMAP] both tests from ONE draw x: x < t_insert -> insertion, else x < t_indel -> deletion, with the deletion
    // threshold rescaled to the draws left: t_indel = t_insert + ((2^32 - t_insert) * t_delete >> 32)
    uint32_t t_indel = 0;
    // [REMAP] the indel tests of a read are i.i.d. Bernoulli(p = t_indel / 2^32) per visited base, so the number of
    // event-free bases before the next event is geometric: gap >= g  <=>  x < gap_t[g], gap_t[g] = floor((1-p)^g * 2^32),
    // g = 0 .. read_length (gap_t[0] = 2^32 - 1); one draw per EVENT instead of one per base.  Given an event it is an
    // insertion when a second draw y < t_kind = floor(2^32 * t_insert / t_indel), else a deletion.
    std::vector<uint32_t> gap_t; uint32_t t_kind = 0;
};
// shared with the oracle: the same loop in the same IEEE operations
std::vector<uint32_t> indel_gap_table(uint32_t t_indel, int read_length);
uint32_t indel_kind_threshold(uint32_t t_insert, uint32_t t_indel);
// the drawable symbols of a quality CDF row and their exact draw counts (sum 2^32); returns their number
int quality_row_weights(const double* cdf94, uint8_t sym[94], uint64_t w[94]);
// alias row of K columns (words: K entries, then K symbol bytes) from those weights; n <= K
void quality_alias_row(const uint8_t* sym, const uint64_t* w, int n, int K, uint32_t* row_words);

// count of 32-bit draws x with (x / 2^32) < c  resp. <= c, clamped to 2^32-1
uint32_t threshold_lt(double c);
uint32_t threshold_le(double c);
// count of x with r(x) <= c for the randIndx mapping, clamped to 2^32-1
uint32_t threshold_cdf(double c);

// [REMAP] thresholds of K ~ Binomial(n, ber), n = n_min..n_max: T[(n-n_min)*BINOM_KMAX + k] = floor(CDF_n(k) * 2^64).
// Replaces the reference's one Bernoulli(ber) draw per amplified base (Fragment.cpp:100-104) by a count draw
// plus K distinct positions (same distribution).  IEEE * + / only: every build computes the same table.
std::vector<uint64_t> binom_table(double ber, int n_min, int n_max);

// Throws std::runtime_error with the reference's message where it has one.
void load_profile(const std::string& path, bool paired, int isize, ProfileTables& out);

// One FASTA record as staged on the host: `code` holds the sequence as RAW ASCII (line breaks removed); the
// ASCII -> base-code conversion (0..3 ACGT, 4 other; upper-casing of Genome::getSubSequence, Genome.cpp:272-278)
// runs on the device after the upload (k_encode_bases).
struct FastaRecord { std::string name; std::vector<uint8_t> code; };
// make_index: also leave <path>.fai beside the file when there is none, as the reference does (fastahack, Fasta.cpp:241-249)
void load_fasta(const std::string& path, std::vector<FastaRecord>& out, bool make_index = false);
void encode_record(const char* name, const char* seq, uint64_t len, FastaRecord& out);
// pieces of load_fasta for the GPU-side parser (scs_stage.cpp stage_fasta_on_device): the plain path of a possibly
// gzip'ed input (inflated beside itself with `gzip -cd`, Genome.cpp:183-187), the index name of a header line, and the
// .fai beside the file (written only when absent) from the header offsets and record lengths the device found
std::string fasta_plain_path(const std::string& path);
std::string fasta_index_name(const std::string& header_text);
void fasta_write_fai(const std::string& path, const char* base, size_t size, const std::vector<uint64_t>& hdr_off, const std::vector<uint64_t>& lens);

}  // namespace scs
