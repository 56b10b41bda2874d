// wt_sorted.hip — wt_pc::select (wt_pc.hpp:443-474) for LARGE batches, bucketed by the argument's place in symbol order.
//
// The direct kernel (wt.hip: k_wt_select_fused) answers a query with about five fabric requests — per fused tree step a directory
// bracket and one or two window probes, each a random 128-byte line — and the fabric's request rate is what bounds it
// (DESIGN.md §4.0).  select is monotone: the k-th and the (k+1)-th occurrence of a symbol lie next to each other in EVERY node
// on the symbol's path.  So the batch is ordered by v = (occurrences of smaller symbols) + k — the place of (c, k) in the sorted
// symbol sequence — with the write-combined passes of bv_swc.hip (the machine behind the batched rank / select of a plain
// vector, keyed by v / B exactly like its select: sw_run_with + SwCallbacks), a bucket of B consecutive places is answered by ONE
// block, whose quads then walk lines and directory entries their neighbours have just pulled into the cache, and the answers
// travel back through the same two un-permute passes.  No directory of its own, no second layout: the walk is the fused select.
#include "bv_sorted_dev.hpp"
#include "wt_host.hpp"

namespace sdslhip {

namespace {

// v[q] = 1 + place of (c, k) among all occurrences in symbol order, 0 for a query that needs no walk (the fix-up answers it)
__global__ __launch_bounds__(256) void k_wt_sel_places(const uint64_t * __restrict__ occ, const uint64_t * __restrict__ iq,
                                                       const uint8_t * __restrict__ cq, uint64_t n, uint64_t * __restrict__ v)
{
    __shared__ uint64_t first[256], cnt[256];
    if (threadIdx.x == 0)
    {
        uint64_t run = 0;
        for (int c = 0; c < 256; ++c)
        {
            first[c] = run;
            cnt[c] = occ[c];
            run += occ[c];
        }
    }
    __syncthreads();
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t k = iq[q];
        const unsigned c = cq[q];
        v[q] = k >= 1 && k <= cnt[c] ? first[c] + k : 0;
    }
}

// what the passes answered with NPOS: a symbol that does not occur -> size() (wt_pc.hpp:447-450); anything else outside
// select's precondition stays NPOS
__global__ __launch_bounds__(256) void k_wt_sel_fixup(const uint64_t * __restrict__ occ, const uint64_t * __restrict__ iq,
                                                      const uint8_t * __restrict__ cq, uint64_t n, uint64_t size, uint64_t * __restrict__ out,
                                                      const uint32_t * __restrict__ go)
{
    if (go && !*go)
        return;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x)
        if (out[q] == SDSL_HIP_NPOS && occ[cq[q]] == 0)
            out[q] = size;
}

__global__ __launch_bounds__(256) void k_wt_sel_zero_bases(unsigned nf, uint64_t * __restrict__ hf)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f < nf)
        hf[f] = 0; // answers are positions inside the sequence (below 2^32: the fused layout's limit), absolute as they are
}

// One block per work item (the keys of one bucket, at most kItemKeys of them).  The walk is k_wt_select_fused's: a flat loop, one
// iteration = one window probe of whatever fused step of whatever key the quad is at.
__global__ __launch_bounds__(kBlock) void k_wt_select_sorted(WtView wt, const uint64_t * __restrict__ occ, unsigned nf,
                                                             unsigned kb, unsigned B, const uint32_t * __restrict__ fstart,
                                                             const uint32_t * __restrict__ ioff, uint32_t * __restrict__ keys,
                                                             const uint32_t * __restrict__ go)
{
    if (go && !*go)
        return;
    __shared__ struct
    {
        uint64_t path[256];
        uint16_t parent[kWtMaxNodes];
        uint16_t c_to_leaf[256];
    } T;
    __shared__ WtFusedTables FT;
    __shared__ WtFusedSelTables FS;
    __shared__ uint32_t first[257]; // occurrences of smaller symbols (the sequence has fewer than 2^32)
    __shared__ unsigned sh_f;
    {
        const uint64_t * src = reinterpret_cast<const uint64_t *>(wt.f_sel_tables);
        uint64_t * dst = reinterpret_cast<uint64_t *>(&FS);
        for (unsigned i = threadIdx.x; i < sizeof(WtFusedSelTables) / 8; i += blockDim.x)
            dst[i] = src[i];
        for (unsigned i = threadIdx.x; i < 256; i += blockDim.x)
        {
            T.path[i] = wt.tables->path[i];
            T.c_to_leaf[i] = wt.tables->c_to_leaf[i];
        }
        for (unsigned i = threadIdx.x; i < kWtMaxNodes; i += blockDim.x)
            T.parent[i] = wt.tables->parent[i];
        if (threadIdx.x == 0)
        {
            uint64_t run = 0;
            for (int c = 0; c < 256; ++c)
            {
                first[c] = (uint32_t)run;
                run += occ[c];
            }
            first[256] = (uint32_t)run; // == size
        }
    }
    wt_stage_fused(&FT, wt); // ends with __syncthreads()
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    const unsigned n_items = ioff[nf];
    for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x)
    {
        if (threadIdx.x == 0)
        {
            unsigned a = 0, z = nf;
            while (a + 1 < z)
            {
                const unsigned m = (a + z) >> 1;
                if (ioff[m] <= item)
                    a = m;
                else
                    z = m;
            }
            sh_f = a;
        }
        __syncthreads();
        const unsigned f = sh_f;
        const uint64_t lo = (uint64_t)fstart[f] + (uint64_t)(item - ioff[f]) * kItemKeys;
        const uint64_t fend = fstart[f + 1];
        const unsigned cnt = (unsigned)(lo + kItemKeys < fend ? kItemKeys : fend - lo);
        uint32_t * kp = keys + lo;
        const uint64_t v0 = (uint64_t)f * B; // the bucket's first place (the write-combined passes keep their tables in bucket order)
        const uint32_t kmask = (1u << kb) - 1u;
        unsigned c0 = 0; // the symbol of the bucket's first place (block-uniform; 8 LDS steps once per item, not per key)
        {
            unsigned z = 256;
            while (c0 + 1 < z)
            {
                const unsigned m = (c0 + z) >> 1;
                if (first[m] <= (uint32_t)v0)
                    c0 = m;
                else
                    z = m;
            }
        }
        unsigned nxt = gq; // this quad's next key of the item
        uint32_t key_nxt = nxt < cnt ? kp[nxt] : kBad;
        bool have = false;
        unsigned mine = 0, groups = 0, len = 0, cur = 0, t = 0;
        uint64_t p = 0, base_line = 0;
        uint32_t res = 0;
        int tries = 0;
        FselBracket br{};
        auto start_group = [&]() {
            const unsigned g = groups - 1, nlev = len - kFK * g < kFK ? len - kFK * g : kFK;
            t = (unsigned)(p >> (kFK * g)) & ((1u << nlev) - 1u);
            unsigned u = cur;
            for (unsigned k = 0; k < nlev; ++k)
                u = T.parent[u];
            cur = u;
            len = kFK * g;
            base_line = FT.fline[u];
            const unsigned rid = FS.root_id[u];
            br = fsel_bracket(wt.f_sel, FS.off[rid][t], res, FS.cnt[rid][t]);
            tries = 0;
        };
        for (;;)
        {
            while (!have && nxt < cnt)
            {
                mine = nxt;
                const uint32_t key = key_nxt;
                nxt += kQPB;
                if (nxt < cnt)
                    key_nxt = kp[nxt];
                if (key >= kMark)
                    continue; // (kBad: no walk — the un-permute passes turn it into NPOS)
                const uint32_t v = (uint32_t)(v0 + (key & kmask)); // place in symbol order, 0-based
                // the symbol: first[c] <= v < first[c + 1] — nearly always the one the bucket starts in, or the next present one
                unsigned c = c0;
                while (first[c + 1] <= v)
                    ++c;
                res = v - first[c];
                cur = T.c_to_leaf[c];
                p = T.path[c];
                len = (unsigned)(p >> 56);
                groups = (len + kFK - 1) / kFK;
                have = true;
                start_group();
            }
            if (__ballot(have) == 0)
                break;
            if (have)
            {
                uint64_t pos;
                if (quad_fsel_probe<false>(wt, base_line, s, t, res, br, tries, pos))
                {
                    res = (uint32_t)pos;
                    if (--groups == 0)
                    {
                        if (s == 0)
                            kp[mine] = res;
                        have = false;
                    }
                    else
                        start_group();
                }
                else
                    ++tries;
            }
        }
        __syncthreads(); // (thread 0 rewrites sh_f at the top of the next item)
    }
}

// ---- one LANE per key ------------------------------------------------------------------------------------------------
// The quad form spends four lanes on every scalar step of a key's walk (bracket, interpolation, compare, the bookkeeping of the
// flat loop) and the kernel was bound by exactly that: VALU share of issue 0.86, 7 of the call's 9 ms.  Inside a bucket the lines a
// key needs are in the cache whoever fetches them, so here a lane walks its key alone: it reads the probed 128-byte line itself
// (eight 16-byte loads), counts its four sections and finds the occurrence — about 2.4 times fewer lane-instructions per key.
__device__ __forceinline__ bool lane_fsel_probe(const uint64_t * __restrict__ f_lines, const uint32_t * __restrict__ f_super,
                                                uint64_t base_line, unsigned t, uint32_t k, FselBracket & b, int tries, uint32_t & pos_out)
{
    const uint32_t span = b.phi - b.plo; // > 0
    uint32_t pe;
    if (tries >= 3 && (tries & 1))
        pe = b.plo + (span >> 1);
    else
    {
        const float f = (float)(k - b.lo_cnt) * __builtin_amdgcn_rcpf((float)(b.hi_cnt - b.lo_cnt));
        const uint32_t o = (uint32_t)(f * (float)span);
        pe = b.plo + (o >= span ? span - 1 : o);
    }
    const uint32_t g = (uint32_t)fused_line(pe);
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    const v2u64 * ln = reinterpret_cast<const v2u64 *>(f_lines + (base_line + g) * kFusedWords);
    v2u64 w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        w[i] = ln[i]; // section s: w[2s] = (header, plane 0), w[2s + 1] = (plane 1, plane 2)  [16-ary: header, three words of 16 positions]
    const uint32_t sup = (uint32_t)fused_super(f_super, false, base_line, base_line + g, t);
    const uint64_t x0 = (t & 1) ? 0 : ~UINT64_C(0), x1 = (t & 2) ? 0 : ~UINT64_C(0), x2 = (t & 4) ? 0 : ~UINT64_C(0);
    uint64_t m[4];
    unsigned c[4];
#pragma unroll
    for (int sct = 0; sct < 4; ++sct)
    {
        if constexpr (kFK == 3)
            m[sct] = (w[2 * sct].y ^ x0) & (w[2 * sct + 1].x ^ x1) & (w[2 * sct + 1].y ^ x2);
        else
            m[sct] = fsec_match_words(w[2 * sct].y, w[2 * sct + 1].x, w[2 * sct + 1].y, t);
        c[sct] = popc64(m[sct]);
    }
    uint32_t c0;
    {
        const unsigned hs = kFK == 3 ? t >> 1 : t >> 2;
        const uint64_t h = hs == 0 ? w[0].x : (hs == 1 ? w[2].x : (hs == 2 ? w[4].x : w[6].x));
        if constexpr (kFK == 3)
            c0 = (uint32_t)(h >> (32 * (t & 1)));
        else
        {
            const uint64_t w2 = hs == 0 ? w[1].y : (hs == 1 ? w[3].y : (hs == 2 ? w[5].y : w[7].y)); // the section's third word: the count's top bits
            c0 = sup + fsec16_count_field(h, w2, t & 3);
        }
    }
    const uint32_t c_in = c[0] + c[1] + c[2] + c[3];
    if (k < c0)
    {
        b.phi = g * kFusedPos;
        b.hi_cnt = c0;
        return false;
    }
    if (k >= c0 + c_in)
    {
        b.plo = (g + 1) * kFusedPos;
        b.lo_cnt = c0 + c_in;
        return false;
    }
    unsigned r = k - c0, sct = 0;
    uint64_t mm = m[0];
    if (r >= c[0])
    {
        r -= c[0];
        sct = 1;
        mm = m[1];
        if (r >= c[1])
        {
            r -= c[1];
            sct = 2;
            mm = m[2];
            if (r >= c[2])
            {
                r -= c[2];
                sct = 3;
                mm = m[3];
            }
        }
    }
    pos_out = g * kFusedPos + kFLane * sct + sel64(mm, r + 1);
    return true;
}

template <unsigned THREADS>
__global__ __launch_bounds__(THREADS) void k_wt_select_sorted_lane(WtView wt, const uint64_t * __restrict__ occ, unsigned nf, unsigned kb,
                                                                  unsigned B, const uint32_t * __restrict__ fstart,
                                                                  const uint32_t * __restrict__ ioff, uint32_t * __restrict__ keys,
                                                                  const uint32_t * __restrict__ go)
{
    if (go && !*go)
        return;
    __shared__ struct
    {
        uint64_t path[256];
        uint16_t parent[kWtMaxNodes];
        uint16_t c_to_leaf[256];
    } T;
    __shared__ WtFusedTables FT;
    __shared__ WtFusedSelTables FS;
    __shared__ uint32_t first[257];
    __shared__ unsigned sh_f;
    {
        const uint64_t * src = reinterpret_cast<const uint64_t *>(wt.f_sel_tables);
        uint64_t * dst = reinterpret_cast<uint64_t *>(&FS);
        for (unsigned i = threadIdx.x; i < sizeof(WtFusedSelTables) / 8; i += blockDim.x)
            dst[i] = src[i];
        for (unsigned i = threadIdx.x; i < 256; i += blockDim.x)
        {
            T.path[i] = wt.tables->path[i];
            T.c_to_leaf[i] = wt.tables->c_to_leaf[i];
        }
        for (unsigned i = threadIdx.x; i < kWtMaxNodes; i += blockDim.x)
            T.parent[i] = wt.tables->parent[i];
        if (threadIdx.x == 0)
        {
            uint64_t run = 0;
            for (int c = 0; c < 256; ++c)
            {
                first[c] = (uint32_t)run;
                run += occ[c];
            }
            first[256] = (uint32_t)run;
        }
    }
    wt_stage_fused(&FT, wt); // ends with __syncthreads()
    // One block per item: the lanes of a block share the lines of ONE bucket — with an item per wave four times as many buckets
    // are in flight and their lines push each other out of the L2 (13.9 against 15.2 G/s).
    const unsigned n_items = ioff[nf];
    for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x)
    {
        if (threadIdx.x == 0)
        {
            unsigned a = 0, z = nf;
            while (a + 1 < z)
            {
                const unsigned mid = (a + z) >> 1;
                if (ioff[mid] <= item)
                    a = mid;
                else
                    z = mid;
            }
            sh_f = a;
        }
        __syncthreads();
        const unsigned f = sh_f;
        const uint64_t lo = (uint64_t)fstart[f] + (uint64_t)(item - ioff[f]) * kItemKeys;
        const uint64_t fend = fstart[f + 1];
        const unsigned cnt = (unsigned)(lo + kItemKeys < fend ? kItemKeys : fend - lo);
        uint32_t * kp = keys + lo;
        const uint64_t v0 = (uint64_t)f * B;
        const uint32_t kmask = (1u << kb) - 1u;
        unsigned c0 = 0;
        {
            unsigned z = 256;
            while (c0 + 1 < z)
            {
                const unsigned mid = (c0 + z) >> 1;
                if (first[mid] <= (uint32_t)v0)
                    c0 = mid;
                else
                    z = mid;
            }
        }
        unsigned nxt = threadIdx.x; // this lane's next key of the item
        uint32_t key_nxt = nxt < cnt ? kp[nxt] : kBad;
        bool have = false;
        unsigned mine = 0, groups = 0, len = 0, cur = 0, t = 0;
        uint64_t p = 0, base_line = 0;
        uint32_t res = 0;
        int tries = 0;
        FselBracket br{};
        for (;;)
        {
            if (!have && nxt < cnt)
            { // (one key per iteration: a kBad key costs its lane one round)
                mine = nxt;
                const uint32_t key = key_nxt;
                nxt += THREADS;
                if (nxt < cnt)
                    key_nxt = kp[nxt];
                if (key < kMark)
                {
                    const uint32_t v = (uint32_t)(v0 + (key & kmask));
                    unsigned c = c0;
                    while (first[c + 1] <= v)
                        ++c;
                    res = v - first[c];
                    cur = T.c_to_leaf[c];
                    p = T.path[c];
                    len = (unsigned)(p >> 56);
                    groups = (len + kFK - 1) / kFK;
                    have = true;
                    tries = -1; // the group below is new
                }
            }
            if (__ballot(have || nxt < cnt) == 0)
                break;
            if (!have)
                continue;
            if (tries < 0)
            { // start of a fused step: the node three levels up (or the root's remainder), its slot, the directory bracket
                const unsigned gq = groups - 1, nlev = len - kFK * gq < kFK ? len - kFK * gq : kFK;
                t = (unsigned)(p >> (kFK * gq)) & ((1u << nlev) - 1u);
                unsigned u = cur;
                for (unsigned j = 0; j < nlev; ++j)
                    u = T.parent[u];
                cur = u;
                len = kFK * gq;
                base_line = FT.fline[u];
                const unsigned rid = FS.root_id[u];
                br = fsel_bracket(wt.f_sel, FS.off[rid][t], res, FS.cnt[rid][t]);
                tries = 0;
            }
            uint32_t pos;
            if (lane_fsel_probe(wt.f_lines, wt.f_super, base_line, t, res, br, tries, pos))
            {
                res = pos;
                tries = -1;
                if (--groups == 0)
                {
                    kp[mine] = res;
                    have = false;
                }
            }
            else
                ++tries;
        }
        __syncthreads(); // (thread 0 rewrites sh_f at the top of the next item)
    }
}

} // namespace

bool wt_select_sorted_applicable(const WtHost & wt, uint64_t n)
{
    const int mode = g_wt_select_sorted_mode.load();
    if (mode == 0 || wt.backend != 0 || !wt.d_fused.p || !wt.d_fsel.p || wt.size < 2 || wt.size >= kLimWtSelectBucketedSymbols)
        return false;
    return mode > 0 ? n >= 4096 : n >= (UINT64_C(1) << 23);
}

size_t wt_select_sorted_scratch_bytes(uint64_t n)
{
    const uint64_t pass = n < (UINT64_C(1) << 30) ? n : (UINT64_C(1) << 30);
    return bv_swc_scratch_bytes(BvView{}, pass) + 256 + n * 8; // the passes' working memory + the places
}

sdsl_hip_status wt_launch_select_sorted(const WtHost & wt, const uint64_t * d_occ, const uint64_t * d_i, const uint8_t * d_c, uint64_t n,
                                        uint64_t * d_out, hipStream_t s, void * scratch, size_t scratch_bytes)
{
    const uint64_t pass = n < (UINT64_C(1) << 30) ? n : (UINT64_C(1) << 30);
    const size_t pass_bytes = (bv_swc_scratch_bytes(BvView{}, pass) + 255) & ~(size_t)255;
    if (scratch_bytes < pass_bytes + n * 8)
    {
        set_error("wt select_sorted: scratch too small");
        return SDSL_HIP_ERR_INVALID;
    }
    uint64_t * places = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(scratch) + pass_bytes);
    hipLaunchKernelGGL(k_wt_sel_places, dim3(grid_for(n, 256, 256u * 8u)), dim3(256), 0, s, d_occ, d_i, d_c, n, places);
    // buckets of B = m << bs consecutive places (m in 8..15), at most 2^16 of them
    SelectPlan sp;
    sp.total = wt.size;
    {
        const uint64_t b_min = std::max<uint64_t>(64, (wt.size + 65535) >> 16);
        unsigned bs = 0;
        while ((b_min >> bs) > 15)
            ++bs;
        unsigned bm = (unsigned)(b_min >> bs);
        if (((uint64_t)bm << bs) < b_min && ++bm == 16)
            bm = 8, ++bs;
        if (bm < 8)
        { // (b_min == 64 exactly fits 8 << 3)
            bm = 8;
            bs = 3;
        }
        sp.bm = bm;
        sp.bs = bs;
        sp.nf = (unsigned)((wt.size + ((uint64_t)bm << bs) - 1) / ((uint64_t)bm << bs));
    }
    const unsigned B = sp.bm << sp.bs;
    const WtView view = wt.view();
    SwCallbacks cb;
    cb.what = "bucketed wt select";
    cb.fill = [&](SrGeom & g, uint64_t cnt)
    {
        BvView none{};
        none.n_bits = wt.size;
        none.n_lines = 2;
        sr_fill_geom(g, none, 1, sp, cnt);
    };
    cb.answers = [&](const SrGeom & g, unsigned nf, const uint32_t * fstart, const uint32_t * ioff, uint32_t * keys2, uint64_t * hf, uint32_t * marked,
                     hipStream_t st) -> sdsl_hip_status
    {
        SH_TRY(fill_u32_async(marked, 0u, 4, st));
        hipLaunchKernelGGL(k_wt_sel_zero_bases, dim3((nf + 255) / 256), dim3(256), 0, st, nf, hf);
        static const bool quad_form = getenv("SDSL_HIP_WT_SEL_LANE") && atoi(getenv("SDSL_HIP_WT_SEL_LANE")) == 0; // (A/B knob)
        static const unsigned lane_grid = getenv("SDSL_HIP_WT_SEL_GRID") ? (unsigned)atoi(getenv("SDSL_HIP_WT_SEL_GRID")) : 256u * 8u;      // measured: 256 x 4096 15.8,
        static const unsigned lane_threads = getenv("SDSL_HIP_WT_SEL_THREADS") ? (unsigned)atoi(getenv("SDSL_HIP_WT_SEL_THREADS")) : 512u; // 512 x 2048 16.2, 1024 x 1024 15.8 G/s
        if (quad_form)
            hipLaunchKernelGGL(k_wt_select_sorted, dim3(256u * 8u), dim3(kBlock), 0, st, view, d_occ, nf, g.kb, B, fstart, ioff, keys2, g.go);
        else if (lane_threads == 1024)
            hipLaunchKernelGGL(k_wt_select_sorted_lane<1024>, dim3(lane_grid), dim3(1024), 0, st, view, d_occ, nf, g.kb, B, fstart, ioff, keys2, g.go);
        else if (lane_threads == 512)
            hipLaunchKernelGGL(k_wt_select_sorted_lane<512>, dim3(lane_grid), dim3(512), 0, st, view, d_occ, nf, g.kb, B, fstart, ioff, keys2, g.go);
        else
            hipLaunchKernelGGL(k_wt_select_sorted_lane<256>, dim3(lane_grid), dim3(256), 0, st, view, d_occ, nf, g.kb, B, fstart, ioff, keys2, g.go);
        SH_HIP(hipGetLastError());
        return SDSL_HIP_OK;
    };
    cb.fixup = [&](const uint32_t *, const uint64_t *, uint64_t * out, uint64_t cnt, hipStream_t st)
    { // (the passes hand over their slice of the batch: `out` is d_out + done, so are the arguments)
        const uint64_t done = (uint64_t)(out - d_out);
        hipLaunchKernelGGL(k_wt_sel_fixup, dim3(grid_for(cnt, 256, 256u * 8u)), dim3(256), 0, st, d_occ, d_i + done, d_c + done, cnt, wt.size, out,
                           (const uint32_t *)nullptr);
    };
    return sw_run_with(cb, 1, places, n, d_out, s, scratch, pass_bytes, nullptr);
}

} // namespace sdslhip
