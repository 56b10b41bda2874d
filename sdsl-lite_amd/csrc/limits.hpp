// limits.hpp — every size gate of the library, named once.  The gates in the .hip files test these constants, sdsl_hip_limit() (common.cpp)
// reports them, INTEGRATION.md 5 tabulates them and tests/test_size_limits.py holds the table to what the library says.  The reference
// has no such gates: its structures are size_type (64-bit) throughout and bounded by host memory alone (int_vector.hpp, wt_pc.hpp:366-474).
#pragma once
#include <cstdint>

namespace sdslhip {

// plain bit vector (bv.hip): the select directory stores positions >> sel_pshift in 32 bits, sel_pshift <= 8
constexpr uint64_t kLimBvBits = UINT64_C(1) << 40;
// bucketed rank / select on a plain vector (bv_sorted.hip, bv_swc.hip): 2^16 slices of up to 2^(10 + 3) rank lines of 448 bits
constexpr unsigned kLimBvBucketedLinesLog = 29;
// rrr_vector<63> (rrr.hip): the stream parser's bound; the device records address 2^32 - 1 records and 2^48 offset bits
constexpr uint64_t kLimRrrBits = UINT64_C(1) << 40;
// bucketed rank / select on an rrr vector (rrr_sorted.hip): 2^16 slices of 256 records (34 blocks of 63 bits each; slim records: 42)
constexpr uint64_t kLimRrrBucketedRecords = UINT64_C(65536) * 256;
// wavelet tree (wt.hip): sequences below this get the fused lines (rank / access / LF / count / select on them); from it on the binary
// levels answer everything (any size the bit vector admits)
constexpr uint64_t kLimWtFusedSymbols = UINT64_C(1) << 36;
// bucketed wt.select (wt_sorted.hip): 32-bit keys; larger trees take the direct fused / binary select
constexpr uint64_t kLimWtSelectBucketedSymbols = UINT64_C(1) << 32;
// FM-index: the flat count kernels' tables, the k-mer table and restore_suffix_array keep intervals in 40 bits with a spare
constexpr uint64_t kLimFmFastSymbols = UINT64_C(1) << 39;
// device suffix sorter (sa.hip): 32-bit suffixes below this many symbols (text + sentinel), 64-bit ones from it on
constexpr uint64_t kLimSorter32Symbols = UINT64_C(0xFFFFFFFE);
// ... and the 64-bit sorter's own bound (its working memory, 40 bytes per symbol, ends at about 6 * 10^9 symbols on 288 GB)
constexpr uint64_t kLimSorter64Symbols = UINT64_C(1) << 40;
// steps tables of the flat walkers (fm_count2.hip, wt.hip): the first line of a fused node in 28 bits; a layout with more lines
// walks through the node tables instead (same answers)
constexpr unsigned kLimStepTableLineBits = 28;

} // namespace sdslhip
