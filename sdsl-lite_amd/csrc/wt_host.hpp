// wt_host.hpp — host-side owner of a device wavelet tree and its builders.
#pragma once
#include "bv_host.hpp"
#include "rrr_host.hpp"
#include "sdsl_stream.hpp"
#include "wt_device.hpp"

namespace sdslhip {

struct WtHost
{
    int device = 0;
    uint64_t size = 0, sigma = 0;
    uint32_t n_nodes = 0;
    uint32_t backend = 0; // 0: plain bit_vector + rank_support_v5 (rank lines); 1: rrr_vector<63>
    BvHost bv;          // the concatenated WT bit vector as rank lines (+ select directories)
    RrrHost rrr;        // ... or as an rrr_vector<63>
    DevBuf d_tables;    // WtTables image in HBM
    DevBuf d_fused;     // the fused layout (16-ary lines; 8-ary in a SDSL_HIP_FUSED_K=3 build) walked by rank / access / LF / count / select (wt_device.hpp), optional
    DevBuf d_ftables;   // its node tables (WtFusedTables)
    DevBuf d_fsuper;    // 16-ary lines: the superblocks' counts (wt_device.hpp), 32-bit or — 2^32 symbols and more — 64-bit records
    DevBuf d_fwalk;     // the layout by fused node (WtFusedWalk), optional
    DevBuf d_fsteps;    // the layout by symbol (WtStepTab), optional
    DevBuf d_tables_f;  // node table of the fused layout's OWN tree shape (2^kFK-ary Huffman written as a binary tree); empty
                        // when the fused layout was derived from the SDSL-shaped tree itself
    WtTables tables_f;  // host copy of it
    DevBuf d_fsel, d_fsel_tables; // its select directory (wt_device.hpp: WtFusedSelTables), optional
    WtTables tables;    // host copy (code lengths, alphabet queries)
    uint64_t occ[256];  // occurrences of every byte (== wt.rank(size(), c))
    bool binary_dropped = false; // wt_drop_binary: bv holds no lines (only n_bits); everything walks the fused layout
    BvView bv_dropped_view{};    // the view as it was (counts of the select directories etc.), for diagnostics
    // what the query kernels see: with a fused layout, the node table of the tree the fused layout was built from
    WtView view() const
    {
        WtView v = view_binary();
        v.f_lines = d_fused.as<uint64_t>();
        v.f_tables = d_ftables.as<WtFusedTables>();
        v.f_super = d_fsuper.as<uint32_t>();
        v.f_walk = d_fwalk.as<WtFusedWalk>();
        v.f_steps = d_fsteps.as<WtStepTab>();
        v.f_sel = d_fsel.as<uint32_t>();
        v.f_sel_tables = d_fsel_tables.as<WtFusedSelTables>();
        if (v.f_lines && d_tables_f.p)
            v.tables = d_tables_f.as<WtTables>();
        return v;
    }
    // SDSL's tree only (binary levels): construction passes, select without the fused directory
    WtView view_binary() const
    {
        WtView v;
        v.bv = bv.view;
        v.rrr = rrr.view;
        v.backend = backend;
        v.tables = d_tables.as<WtTables>();
        v.size = size;
        v.sigma = sigma;
        v.n_nodes = n_nodes;
        v.f_lines = nullptr;
        v.f_tables = nullptr;
        v.f_super = nullptr;
        v.f_walk = nullptr;
        v.f_steps = nullptr;
        v.f_sel = nullptr;
        v.f_sel_tables = nullptr;
        return v;
    }
    size_t device_bytes() const
    {
        return bv.device_bytes() + rrr.device_bytes() + d_tables.bytes + d_fused.bytes + d_ftables.bytes + d_fsuper.bytes + d_fwalk.bytes + d_fsteps.bytes + d_fsel.bytes + d_fsel_tables.bytes + d_tables_f.bytes;
    }
};

// Builds the tree shape on the host (a 256-entry histogram decides it) and the bit vector on the device, one
// stable radix sort per tree level, from a symbol sequence that already lives in device memory.
// flags: SDSL_HIP_WT_RRR63 (bit vector as rrr_vector<63> instead of rank lines + select directories),
// SDSL_HIP_WT_BLCD (balanced shape instead of Huffman); internal: kWtShapeHuff8, kWtNoSelect
constexpr uint32_t kWtShapeHuff8 = 0x100u; // 2^kFK-ary Huffman tree written as a binary tree (the fused layout's own shape)
constexpr uint32_t kWtNoSelect = 0x200u;   // no select directories on the bit vector
constexpr uint32_t kWtShapeGiven = 0x400u; // wt.tables / n_nodes / sigma are set by the caller: build the bits of THAT tree
// words_out: stop after the level builder and hand back the tree's bits as SDSL's words (no rank lines, no directories)
sdsl_hip_status wt_build_from_device_text(WtHost & wt, const uint8_t * d_text, uint64_t n, int device, uint32_t flags = 0,
                                          DevBuf * words_out = nullptr);
// Parses wt_pc::serialize output (wt_pc.hpp:713-726) and uploads; advances the reader.
// layout: 0 = plain bv + select_support_scan (zero bytes), 1 = plain bv + select_support_mcl, 2 = rrr_vector<63> with
// its own rank/select supports (zero bytes)
sdsl_hip_status wt_build_from_stream(WtHost & wt, StreamReader & rd, int layout, int device);
uint64_t wt_bv_bits(const WtHost & wt);
// SDSL's binary levels released / rebuilt from the fused lines (wt.hip; plain backend with the fused layout only)
sdsl_hip_status wt_drop_binary(WtHost & wt);
sdsl_hip_status wt_restore_binary(WtHost & wt);
// Derives the fused layout from the finished binary tree (plain backend, fewer than kLimWtFusedSymbols = 2^36 symbols; SDSL_HIP_WT_FUSED=0
// in the environment turns it off).  A no-op otherwise.
sdsl_hip_status wt_build_fused(WtHost & wt);

// kernels over the rrr backend (wt_rrr.hip)
sdsl_hip_status wt_rrr_launch_rank(const WtHost & wt, const uint64_t * d_i, const uint8_t * d_c, uint64_t n,
                                   uint64_t * d_out, hipStream_t s);
sdsl_hip_status wt_rrr_launch_select(const WtHost & wt, const uint64_t * d_occ, const uint64_t * d_i, const uint8_t * d_c,
                                     uint64_t n, uint64_t * d_out, hipStream_t s);
sdsl_hip_status wt_rrr_launch_inverse_select(const WtHost & wt, const uint64_t * d_i, uint64_t n, uint64_t * d_rank,
                                             uint8_t * d_c, hipStream_t s);

// select of a large batch, bucketed by the argument's place in symbol order (wt_sorted.hip)
bool wt_select_sorted_applicable(const WtHost & wt, uint64_t n);
size_t wt_select_sorted_scratch_bytes(uint64_t n);
sdsl_hip_status wt_launch_select_sorted(const WtHost & wt, const uint64_t * d_occ, const uint64_t * d_i, const uint8_t * d_c, uint64_t n,
                                        uint64_t * d_out, hipStream_t s, void * scratch, size_t scratch_bytes);

sdsl_hip_status wt_launch_rank(const WtHost & wt, const uint64_t * d_i, const uint8_t * d_c, uint64_t n,
                               uint64_t * d_out, hipStream_t s);

} // namespace sdslhip
