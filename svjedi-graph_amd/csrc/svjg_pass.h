// What a finished fused pass (svjg_run_begin / svjg_run_end) means for the ranks of a job — host logic shared by libsvjg_hip.so
// and the CPU harness (tests/hostsim), so that the decision the ranks must take TOGETHER is tested where there is no GPU.
//
// A pass enqueues, with no host round trip: reset, k_classify_main, both exact-path kernels (one wave per line for up to
// `wave_limit` deferred lines, one lane per line beyond it; each reads the number on the device and works only in its range),
// the guard kernel, the all-reduce of [ counts | guard words ], the genotypes.  The only thing the device cannot repair by itself
// is a list that overflowed (deferred lines, lines for the host): such a pass has to be repeated with larger lists.  Under a
// communicator that decision is COLLECTIVE: every rank puts "I must repeat" into guard word 2, the words travel through the
// pass's own all-reduce, and every rank repeats (classify step by step + ONE more all-reduce) iff the sum is not zero — so all
// ranks issue the same number of collectives in the same order and none hands out a sum that lacks a rank's deferred lines.
#pragma once
#include <stdint.h>

namespace svjg {

constexpr uint32_t GUARD_WORDS = 3;          // behind the count vector: largest ref field, largest alt field, ranks that must repeat the pass
constexpr uint32_t GUARD_MAX_REF = 0, GUARD_MAX_ALT = 1, GUARD_REPEAT = 2;

// what this rank contributes to guard word 2 (computed on the device by k_counts_guard from the pass's status block)
inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint64_t pass_repeat_word(uint32_t overflow_bits) { return overflow_bits ? 1u : 0u; }

// does the pass have to be repeated?  has_comm: the guard words went through the all-reduce (their sum over the ranks is at hand);
// otherwise the rank is alone and its own status decides.
inline bool pass_repeats(bool has_comm, uint32_t own_overflow_bits, uint64_t guard_repeat_sum) {
    return has_comm ? guard_repeat_sum != 0 : own_overflow_bits != 0;
}

// 32-bit halves of the packed ref | alt << 32 counters cannot have carried into each other iff the SUMS of the ranks' maxima fit
inline bool pass_counts_overflowed(uint64_t guard_max_ref_sum, uint64_t guard_max_alt_sum) {
    return guard_max_ref_sum >= (1ull << 32) || guard_max_alt_sum >= (1ull << 32);
}

}  // namespace svjg
