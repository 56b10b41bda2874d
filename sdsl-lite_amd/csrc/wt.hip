// wt.hip — byte wavelet tree (wt_huff<bit_vector, rank_support_v5<>>) on the device:
// host-side construction of the Huffman-shaped tree and its bit vector, the loader for SDSL's
// serialised form, and the batched rank / access / inverse_select / select kernels.
//
// Reference semantics reproduced:
//   wt_pc::rank            wt_pc.hpp:371-399      wt_pc::operator[]       wt_pc.hpp:336-357
//   wt_pc::inverse_select  wt_pc.hpp:411-430      wt_pc::select           wt_pc.hpp:443-474
//   Huffman shape          wt_huff.hpp:83-115     BFS byte tree + paths   wt_helper.hpp:230-327
#include <algorithm>
#include <chrono>
#include <functional>
#include <queue>

#include "bv_serialize.hpp"
#include "wt_host.hpp"

namespace sdslhip {

// =========================================================================================
// host construction
// =========================================================================================

static uint64_t popcount_range(const uint64_t * w, uint64_t from, uint64_t to)
{ // ones in bit positions [from, to)
    if (from >= to)
        return 0;
    uint64_t fw = from >> 6, tw = to >> 6, c = 0;
    if (fw == tw)
        return popc64((w[fw] >> (from & 63)) & lo_set((unsigned)(to - from)));
    c += popc64(w[fw] >> (from & 63));
    for (uint64_t i = fw + 1; i < tw; ++i)
        c += popc64(w[i]);
    if (to & 63)
        c += popc64(w[tw] & lo_set((unsigned)(to & 63)));
    return c;
}

static void tables_clear(WtTables & T)
{
    memset(&T, 0, sizeof T);
    for (int v = 0; v < kWtMaxNodes; ++v)
        T.child[v][0] = T.child[v][1] = T.parent[v] = kWtUndef;
    for (int c = 0; c < 256; ++c)
        T.c_to_leaf[c] = kWtUndef;
}

// Code trees in SDSL's node numbering.  A shape builder fills `tmp` (root last or anywhere, `root` says where);
// nodes are then renumbered breadth-first from the root (wt_helper.hpp:230-300) and every inner node gets the running
// sum of the preceding inner-node frequencies as the start of its slice.
struct ShapeTmp
{
    uint64_t freq;
    int sym, left, right;
};

// wt_huff (wt_huff.hpp:83-115): leaves in symbol order, the two smallest (frequency, creation index) pairs are
// merged, first popped = left
static int shape_huffman(const uint64_t occ[256], std::vector<ShapeTmp> & tmp)
{
    typedef std::pair<uint64_t, int> Item;
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
    for (int c = 0; c < 256; ++c)
        if (occ[c])
        {
            heap.push(Item(occ[c], (int)tmp.size()));
            tmp.push_back(ShapeTmp{occ[c], c, -1, -1});
        }
    while (heap.size() > 1)
    {
        Item a = heap.top();
        heap.pop();
        Item b = heap.top();
        heap.pop();
        heap.push(Item(a.first + b.first, (int)tmp.size()));
        tmp.push_back(ShapeTmp{a.first + b.first, -1, a.second, b.second});
    }
    return (int)tmp.size() - 1;
}

// wt_blcd (wt_blcd.hpp:86-127): the symbols in order, split into halves of ceil(sigma/2) and the rest, recursively
static int shape_balanced_rec(const std::vector<int> & syms, size_t lb, size_t sigma, const uint64_t occ[256],
                              std::vector<ShapeTmp> & tmp)
{
    if (sigma == 1)
    {
        tmp.push_back(ShapeTmp{occ[syms[lb]], syms[lb], -1, -1});
        return (int)tmp.size() - 1;
    }
    const int id = (int)tmp.size();
    tmp.push_back(ShapeTmp{0, -1, -1, -1});
    const size_t l_sigma = (sigma + 1) / 2;
    const int l = shape_balanced_rec(syms, lb, l_sigma, occ, tmp);
    const int r = shape_balanced_rec(syms, lb + l_sigma, sigma - l_sigma, occ, tmp);
    tmp[id].freq = tmp[l].freq + tmp[r].freq;
    tmp[id].left = l;
    tmp[id].right = r;
    return id;
}

// wt_hutu (wt_hutu.hpp:355-676): Hu-Tucker, the optimal ALPHABETIC code.  Phase 1 works on a sequence of nodes; two
// nodes are compatible if no leaf lies strictly between them, so the sequence falls into groups delimited by leaves.
// Every group offers its two smallest nodes by (weight, position); the offer with the smallest (weight sum, left
// position, right position) is merged into an inner node at the left position (the rules of the reference's master
// queue, :299-311, 336-343, 605-625).  Only the leaf LEVELS of that merge tree matter; phase 2 rebuilds the
// alphabetic tree from them with the stack algorithm (:641-676).  sigma <= 256, so the quadratic search is free.
static int shape_hutu(const std::vector<int> & syms, const uint64_t occ[256], std::vector<ShapeTmp> & tmp)
{
    const size_t sigma = syms.size();
    struct Seq
    {
        uint64_t w;
        bool alive, leaf;
        int left, right; // indices into `merge` (inner) or -1
    };
    struct M
    {
        int l_is_leaf, l, r_is_leaf, r; // children: leaf position or merge index
    };
    std::vector<Seq> A(sigma);
    std::vector<M> merge;
    std::vector<int> node_of(sigma); // position -> merge index of the inner node living there (if !leaf)
    for (size_t i = 0; i < sigma; ++i)
        A[i] = Seq{occ[syms[i]], true, true, -1, -1};
    std::vector<unsigned> level(sigma, 0);
    for (size_t step = 1; step < sigma; ++step)
    {
        bool have = false;
        uint64_t best_sum = 0;
        size_t best_i = 0, best_j = 0;
        // scan the groups: [.. inner .. leaf] [leaf .. inner .. leaf] ...
        size_t g0 = 0;
        auto offer = [&](size_t lo, size_t hi)
        { // two smallest (w, pos) among the alive nodes at positions [lo, hi]
            size_t a = SIZE_MAX, b = SIZE_MAX;
            for (size_t p = lo; p <= hi; ++p)
            {
                if (!A[p].alive)
                    continue;
                if (a == SIZE_MAX || A[p].w < A[a].w)
                {
                    b = a;
                    a = p;
                }
                else if (b == SIZE_MAX || A[p].w < A[b].w)
                    b = p;
            }
            if (b == SIZE_MAX)
                return;
            const uint64_t sum = A[a].w + A[b].w;
            const size_t i = std::min(a, b), j = std::max(a, b);
            if (!have || sum < best_sum || (sum == best_sum && (i < best_i || (i == best_i && j < best_j))))
            {
                have = true;
                best_sum = sum;
                best_i = i;
                best_j = j;
            }
        };
        for (size_t p = 0; p < sigma; ++p)
            if (A[p].alive && A[p].leaf)
            {
                offer(g0, p);
                g0 = p;
            }
        offer(g0, sigma - 1);
        M m;
        m.l_is_leaf = A[best_i].leaf;
        m.l = A[best_i].leaf ? (int)best_i : node_of[best_i];
        m.r_is_leaf = A[best_j].leaf;
        m.r = A[best_j].leaf ? (int)best_j : node_of[best_j];
        merge.push_back(m);
        A[best_i].w = best_sum;
        A[best_i].leaf = false;
        node_of[best_i] = (int)merge.size() - 1;
        A[best_j].alive = false;
    }
    // leaf levels: depth in the merge tree (root = the last merge)
    if (!merge.empty())
    {
        std::vector<std::pair<int, unsigned>> st; // (merge index, depth)
        st.push_back({(int)merge.size() - 1, 0u});
        while (!st.empty())
        {
            auto [mi, d] = st.back();
            st.pop_back();
            const M & m = merge[mi];
            if (m.l_is_leaf)
                level[m.l] = d + 1;
            else
                st.push_back({m.l, d + 1});
            if (m.r_is_leaf)
                level[m.r] = d + 1;
            else
                st.push_back({m.r, d + 1});
        }
    }
    // phase 2: the alphabetic tree with these leaf levels
    std::vector<std::pair<int, unsigned>> stack; // (tmp index, level)
    size_t q = 0;
    while (q < sigma || stack.size() > 1)
    {
        const size_t sp = stack.size();
        if (sp >= 2 && stack[sp - 1].second == stack[sp - 2].second)
        {
            const int l = stack[sp - 2].first, r = stack[sp - 1].first;
            const unsigned lv = stack[sp - 1].second - 1;
            tmp.push_back(ShapeTmp{tmp[l].freq + tmp[r].freq, -1, l, r});
            stack.pop_back();
            stack.back() = {(int)tmp.size() - 1, lv};
        }
        else
        {
            tmp.push_back(ShapeTmp{occ[syms[q]], syms[q], -1, -1});
            stack.push_back({(int)tmp.size() - 1, level[q]});
            ++q;
        }
    }
    return stack[0].first;
}

// The fused layout's own shape: an 8-ary Huffman tree (the expected number of fused steps per symbol is what it
// minimises), written as a binary tree in which every 8-ary node is a complete subtree of depth 3.  Zero-weight dummies
// pad the alphabet to 7k + 1 leaves as usual; they all end up in the first merged node, whose real children are then
// leaves and form a balanced subtree of depth <= 3.  Returns -1 if the tree cannot be written that way.
static int shape_huff8(const uint64_t occ[256], std::vector<ShapeTmp> & tmp)
{
    struct N8
    {
        uint64_t w;
        int sym; // >= 0 leaf, -1 inner, -2 dummy
        std::vector<int> kids;
    };
    std::vector<N8> nodes;
    typedef std::pair<uint64_t, int> Item;
    std::priority_queue<Item, std::vector<Item>, std::greater<Item>> heap;
    for (int c = 0; c < 256; ++c)
        if (occ[c])
        {
            heap.push(Item(occ[c], (int)nodes.size()));
            nodes.push_back(N8{occ[c], c, {}});
        }
    if (heap.size() < 2)
        return -1;
    constexpr int ARY = (int)kFSlots; // 8 or 16 (kFK levels of the binary tree per fused step)
    while ((heap.size() - 1) % (ARY - 1) != 0)
    {
        heap.push(Item(0, (int)nodes.size()));
        nodes.push_back(N8{0, -2, {}});
    }
    while (heap.size() > 1)
    {
        N8 in{0, -1, {}};
        for (int k = 0; k < ARY; ++k)
        {
            const Item it = heap.top();
            heap.pop();
            in.w += it.first;
            if (nodes[it.second].sym != -2)
                in.kids.push_back(it.second);
        }
        heap.push(Item(in.w, (int)nodes.size()));
        nodes.push_back(in);
    }
    const int root8 = heap.top().second;
    bool ok = true;
    // binary subtree over kids[lo, hi) of an 8-ary node
    std::function<int(int)> expand;
    std::function<int(const std::vector<int> &, size_t, size_t)> over = [&](const std::vector<int> & kids, size_t lo,
                                                                            size_t hi) -> int {
        if (hi - lo == 1)
            return expand(kids[lo]);
        const size_t mid = lo + (hi - lo + 1) / 2;
        const int l = over(kids, lo, mid), r = over(kids, mid, hi);
        tmp.push_back(ShapeTmp{tmp[l].freq + tmp[r].freq, -1, l, r});
        return (int)tmp.size() - 1;
    };
    expand = [&](int id) -> int {
        const N8 & nd = nodes[id];
        if (nd.sym >= 0)
        {
            tmp.push_back(ShapeTmp{nd.w, nd.sym, -1, -1});
            return (int)tmp.size() - 1;
        }
        if (nd.kids.size() < 2)
            ok = false;
        if (nd.kids.size() < (size_t)ARY) // fewer than ARY children: they must all be leaves (depth <= kFK is then enough)
            for (int k : nd.kids)
                if (nodes[k].sym < 0)
                    ok = false;
        if (!ok)
        {
            tmp.push_back(ShapeTmp{nd.w, 0, -1, -1});
            return (int)tmp.size() - 1;
        }
        return over(nd.kids, 0, nd.kids.size());
    };
    const int root = expand(root8);
    return ok ? root : -1;
}

static sdsl_hip_status build_shape(const uint64_t occ[256], uint32_t shape, WtTables & T, uint32_t & n_nodes,
                                   uint64_t & bv_size, uint64_t & sigma)
{
    std::vector<ShapeTmp> tmp;
    std::vector<int> syms;
    for (int c = 0; c < 256; ++c)
        if (occ[c])
            syms.push_back(c);
    sigma = syms.size();
    tables_clear(T);
    n_nodes = 0;
    bv_size = 0;
    if (syms.empty())
        return SDSL_HIP_OK;
    const int root = shape == 1 ? shape_balanced_rec(syms, 0, syms.size(), occ, tmp)
                                : (shape == 2 ? shape_hutu(syms, occ, tmp)
                                              : (shape == 3 ? shape_huff8(occ, tmp) : shape_huffman(occ, tmp)));
    if (root < 0)
    {
        set_error("internal: no 8-ary shape for this alphabet");
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    n_nodes = (uint32_t)tmp.size();
    // breadth-first renumbering
    std::vector<int> order; // order[bfs id] = tmp id
    order.reserve(n_nodes);
    order.push_back(root);
    for (size_t head = 0; head < order.size(); ++head)
    {
        const ShapeTmp & t = tmp[order[head]];
        T.bv_pos[head] = bv_size;
        if (t.left >= 0)
        {
            bv_size += t.freq;
            for (int k = 0; k < 2; ++k)
            {
                uint16_t id = (uint16_t)order.size();
                order.push_back(k ? t.right : t.left);
                T.child[head][k] = id;
                T.parent[id] = (uint16_t)head;
            }
        }
        else
        {
            T.bv_pos_rank[head] = (uint64_t)t.sym;
            T.c_to_leaf[t.sym] = (uint16_t)head;
        }
    }
    // root-to-leaf paths, LSB = edge leaving the root; absent symbols: length 0, payload = previous
    // present symbol (wt_helper.hpp:311-315)
    uint64_t prev = 0;
    for (int c = 0; c < 256; ++c)
    {
        if (T.c_to_leaf[c] == kWtUndef)
        {
            T.path[c] = prev;
            continue;
        }
        uint64_t bits = 0, len = 0;
        for (uint16_t v = T.c_to_leaf[c]; v != 0; v = T.parent[v])
        {
            bits = (bits << 1) | (T.child[T.parent[v]][1] == v ? 1u : 0u);
            ++len;
        }
        if (len > 56)
        {
            set_error("wavelet tree code depth %llu exceeds 56 (SDSL throws here too, wt_helper.hpp:304-307)",
                      (unsigned long long)len);
            return SDSL_HIP_ERR_UNSUPPORTED;
        }
        T.path[c] = bits | (len << 56);
        prev = (uint64_t)c;
    }
    return SDSL_HIP_OK;
}

static sdsl_hip_status upload(WtHost & wt, const std::vector<uint64_t> & words, uint64_t bv_size, int device)
{
    wt.device = device;
    DevBuf d_words;
    uint64_t nw = (bv_size + 63) >> 6;
    SH_TRY(d_words.alloc(nw * 8));
    if (nw)
        SH_HIP(hipMemcpy(d_words.p, words.data(), nw * 8, hipMemcpyHostToDevice));
    wt.bv.device = device;
    SH_TRY(bv_build_from_device_words(wt.bv, d_words.as<uint64_t>(), bv_size,
                                      SDSL_HIP_BV_SELECT1 | SDSL_HIP_BV_SELECT0, default_sel_shift()));
    SH_TRY(wt.d_tables.alloc(sizeof(WtTables)));
    SH_HIP(hipMemcpy(wt.d_tables.p, &wt.tables, sizeof(WtTables), hipMemcpyHostToDevice));
    return SDSL_HIP_OK;
}

// ---- device-side construction of the bit vector ---------------------------------------------------------------
// SDSL appends one bit per symbol per level at the cursor of the node on the symbol's path (wt_pc.hpp:97-111,
// 218-242).  The result has a closed form that maps onto sorts: inner nodes are laid out in BFS order, i.e. level by
// level, and inside a level a node's slice lists its symbols in text order.  So the bits of level d are: take the
// text, keep the symbols whose code is longer than d, stable-sort them by the BFS index of their depth-d ancestor,
// and emit bit d of each code.  One 9-bit stable radix sort per level, independent of the other levels.
struct WtLevelKeys
{
    uint16_t key[256]; // (BFS index of the depth-d ancestor << 1) | code bit d; 0xFFFF if the code is shorter
};

__global__ __launch_bounds__(256) void k_wt_hist(const uint8_t * __restrict__ text, uint64_t n,
                                                 unsigned long long * __restrict__ occ)
{
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[text[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x])
        atomicAdd(&occ[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void k_wt_level_keys(const uint8_t * __restrict__ text, uint64_t n, WtLevelKeys tab,
                                                       uint16_t * __restrict__ keys)
{
    __shared__ uint16_t k[256];
    k[threadIdx.x] = tab.key[threadIdx.x];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        keys[i] = k[text[i]];
}

// thread t packs the code bits of sorted elements [64t, 64t+64) into bit positions level_start + 64t ...
__global__ __launch_bounds__(256) void k_wt_pack_level(const uint16_t * __restrict__ sorted, uint64_t alive,
                                                       uint64_t level_start, unsigned long long * __restrict__ words)
{
    const uint64_t chunks = (alive + 63) >> 6;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < chunks; t += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t v = 0;
        const uint64_t base = t << 6;
        const unsigned cnt = (unsigned)(alive - base < 64 ? alive - base : 64);
        for (unsigned j = 0; j < cnt; ++j)
            v |= (uint64_t)(sorted[base + j] & 1u) << j;
        const uint64_t pos = level_start + base;
        const unsigned off = (unsigned)(pos & 63);
        if (v)
        {
            atomicOr(&words[pos >> 6], (unsigned long long)(v << off));
            if (off && (v >> (64 - off)))
                atomicOr(&words[(pos >> 6) + 1], (unsigned long long)(v >> (64 - off)));
        }
    }
}

__global__ void k_wt_node_positions(const WtTables * __restrict__ T, uint32_t n_nodes, uint64_t * __restrict__ pos)
{
    for (uint32_t v = threadIdx.x; v < n_nodes; v += blockDim.x)
        pos[v] = T->bv_pos[v];
}

size_t sort_keys_u16_temp_bytes(uint64_t n, unsigned begin_bit, unsigned end_bit);
sdsl_hip_status sort_keys_u16(uint16_t * keys, uint16_t * other, uint64_t n, unsigned begin_bit, unsigned end_bit, hipStream_t s, void * tmp,
                              size_t tmp_bytes, uint16_t ** sorted_out);

sdsl_hip_status wt_build_from_device_text(WtHost & wt, const uint8_t * d_text, uint64_t n, int device, uint32_t flags, DevBuf * words_out)
{
    const uint32_t backend = (flags & SDSL_HIP_WT_RRR63) ? 1u : 0u;
    SH_HIP(hipSetDevice(device));
    wt.device = device;
    wt.size = n;
    // 1. symbol histogram
    {
        DevBuf d_occ;
        SH_TRY(d_occ.alloc(256 * 8, true));
        if (n)
            hipLaunchKernelGGL(k_wt_hist, dim3(grid_for(n, 256 * 16, 256u * 8u)), dim3(256), 0, 0, d_text, n,
                               d_occ.as<unsigned long long>());
        SH_HIP(hipGetLastError());
        SH_HIP(hipMemcpy(wt.occ, d_occ.p, 256 * 8, hipMemcpyDeviceToHost));
    }
    // 2. shape
    uint64_t bv_size = 0;
    if ((flags & SDSL_HIP_WT_BLCD) && (flags & SDSL_HIP_WT_HUTU))
    {
        set_error("wt_create: SDSL_HIP_WT_BLCD and SDSL_HIP_WT_HUTU exclude each other");
        return SDSL_HIP_ERR_INVALID;
    }
    if (flags & kWtShapeGiven)
    { // the caller has put the node table, n_nodes and sigma into `wt` (wt_restore_binary: whatever shape the tree has)
        for (int c = 0; c < 256; ++c)
            if (wt.tables.c_to_leaf[c] != kWtUndef)
                bv_size += wt.occ[c] * (wt.tables.path[c] >> 56);
            else if (wt.occ[c])
            {
                set_error("internal: symbol %d occurs in the sequence but has no leaf in the given tree shape", c);
                return SDSL_HIP_ERR_HIP;
            }
    }
    else
        SH_TRY(build_shape(wt.occ,
                           (flags & kWtShapeHuff8) ? 3u : ((flags & SDSL_HIP_WT_BLCD) ? 1u : ((flags & SDSL_HIP_WT_HUTU) ? 2u : 0u)),
                           wt.tables, wt.n_nodes, bv_size, wt.sigma));
    WtTables & T = wt.tables;
    const bool tr_b = getenv("SDSL_HIP_TRACE_BUILD") != nullptr;
    auto mark_b = [&](const char * what, unsigned d) {
        if (tr_b)
        {
            const hipError_t e = hipDeviceSynchronize();
            fprintf(stderr, "[sdsl_hip] wt build: %s %u (%s)\n", what, d, hipGetErrorString(e));
        }
    };
    mark_b("shape, nodes", wt.n_nodes);
    // 3. bits, level by level
    const uint64_t nw = (bv_size + 63) >> 6;
    DevBuf d_words;
    SH_TRY(d_words.alloc((nw + 2) * 8, true));
    if (bv_size)
    {
        DevBuf k0, k1, sort_tmp; // (the sort's working memory: one plain allocation for all levels, sa.hip: sort_pairs_u64_u32)
        SH_TRY(k0.alloc(n * 2));
        SH_TRY(k1.alloc(n * 2));
        SH_TRY(sort_tmp.alloc(sort_keys_u16_temp_bytes(n, 1u, 10u)));
        uint64_t level_start = 0;
        for (unsigned d = 0; d < 57; ++d)
        {
            WtLevelKeys tab;
            uint64_t alive = 0;
            for (int c = 0; c < 256; ++c)
            {
                tab.key[c] = 0xFFFF;
                if (T.c_to_leaf[c] == kWtUndef)
                    continue;
                uint64_t p = T.path[c];
                unsigned len = (unsigned)(p >> 56);
                if (len <= d)
                    continue;
                unsigned v = 0;
                for (unsigned l = 0; l < d; ++l)
                    v = T.child[v][(p >> l) & 1];
                tab.key[c] = (uint16_t)((v << 1) | ((p >> d) & 1));
                alive += wt.occ[c];
            }
            if (alive == 0)
                break;
            hipLaunchKernelGGL(k_wt_level_keys, dim3(grid_for(n, 256 * 8, 256u * 8u)), dim3(256), 0, 0, d_text, n, tab,
                               k0.as<uint16_t>());
            SH_HIP(hipGetLastError());
            mark_b("level keys", d);
            uint16_t * sorted = nullptr;
            SH_TRY(sort_keys_u16(k0.as<uint16_t>(), k1.as<uint16_t>(), n, 1u, 10u, nullptr, sort_tmp.p, sort_tmp.bytes, &sorted)); // dead keys sort last
            mark_b("level sorted", d);
            hipLaunchKernelGGL(k_wt_pack_level, dim3(grid_for((alive + 63) >> 6, 256, 256u * 8u)), dim3(256), 0, 0,
                               sorted, alive, level_start, d_words.as<unsigned long long>());
            SH_HIP(hipGetLastError());
            mark_b("level packed", d);
            level_start += alive;
        }
        if (level_start != bv_size)
        {
            set_error("internal: wavelet-tree levels cover %llu of %llu bits", (unsigned long long)level_start,
                      (unsigned long long)bv_size);
            return SDSL_HIP_ERR_HIP;
        }
    }
    if (words_out)
    { // the caller wants the tree's bits in SDSL's word layout and nothing else (the serialiser of a tree without its binary levels)
        wt.bv.view.n_bits = bv_size;
        *words_out = std::move(d_words);
        return SDSL_HIP_OK;
    }
    // 4. rank lines + select directories
    wt.bv.device = device;
    SH_TRY(bv_build_from_device_words(wt.bv, d_words.as<uint64_t>(), bv_size,
                                      (flags & kWtNoSelect) ? 0u : (SDSL_HIP_BV_SELECT1 | SDSL_HIP_BV_SELECT0),
                                      default_sel_shift()));
    mark_b("rank lines built, bits", (unsigned)bv_size);
    // 5. bv_pos_rank of the inner nodes = rank_1 at the start of their slices (wt_helper.hpp:320-327)
    SH_TRY(wt.d_tables.alloc(sizeof(WtTables)));
    SH_HIP(hipMemcpy(wt.d_tables.p, &T, sizeof(WtTables), hipMemcpyHostToDevice));
    if (wt.n_nodes)
    {
        DevBuf d_pos, d_rank;
        SH_TRY(d_pos.alloc(wt.n_nodes * 8));
        SH_TRY(d_rank.alloc(wt.n_nodes * 8));
        hipLaunchKernelGGL(k_wt_node_positions, dim3(1), dim3(256), 0, 0, wt.d_tables.as<WtTables>(), wt.n_nodes,
                           d_pos.as<uint64_t>());
        SH_HIP(hipGetLastError());
        SH_TRY(bv_launch_rank(wt.bv.view, 1, d_pos.as<uint64_t>(), wt.n_nodes, d_rank.as<uint64_t>(), nullptr));
        std::vector<uint64_t> ranks(wt.n_nodes);
        SH_HIP(hipMemcpy(ranks.data(), d_rank.p, wt.n_nodes * 8, hipMemcpyDeviceToHost));
        for (uint32_t v = 0; v < wt.n_nodes; ++v)
            if (T.child[v][0] != kWtUndef)
                T.bv_pos_rank[v] = ranks[v];
        SH_HIP(hipMemcpy(wt.d_tables.p, &T, sizeof(WtTables), hipMemcpyHostToDevice));
    }
    // 6. wt_huff<rrr_vector<63>>: encode the same bits as an rrr vector and drop the plain lines
    wt.backend = backend;
    if (backend == 1)
    {
        SH_TRY(rrr_build_device(wt.rrr, d_words.as<uint64_t>(), bv_size, device));
        wt.bv.lines.release();
        wt.bv.sel[0].release();
        wt.bv.sel[1].release();
        wt.bv.view.lines = nullptr;
        wt.bv.view.sel[0] = wt.bv.view.sel[1] = nullptr;
    }
    return SDSL_HIP_OK;
}

uint64_t wt_bv_bits(const WtHost & wt)
{
    return wt.backend == 1 ? wt.rrr.view.n_bits : wt.bv.view.n_bits;
}

sdsl_hip_status wt_build_from_stream(WtHost & wt, StreamReader & rd, int layout, int device)
{
    HostIntVec bv;
    uint64_t n_nodes64 = 0;
    if (layout < 0 || layout > 3)
    {
        set_error("wt stream layout must be 0 (scan selects), 1 or 3 (mcl selects) or 2 (rrr_vector<63>)");
        return SDSL_HIP_ERR_INVALID;
    }
    if (layout == SDSL_HIP_LAYOUT_BV_DEFAULT)
        layout = SDSL_HIP_LAYOUT_BV_MCL; // the rank support (v or v5) is skipped either way
    wt.backend = layout == 2 ? 1u : 0u;
    if (!rd.u64(wt.size) || !rd.u64(wt.sigma))
        goto bad;
    if (layout == 2)
    { // wt_pc<…, rrr_vector<63>, rank_support_rrr, select_support_rrr<1>, select_support_rrr<0>>: the supports
      // serialise to nothing (rrr_vector.hpp:580-585,769-774)
        wt.rrr.device = device;
        sdsl_hip_status st = rrr_parse_and_upload(wt.rrr, rd, device, &bv.words);
        if (st != SDSL_HIP_OK)
            return st;
        bv.bit_size = wt.rrr.view.n_bits;
        bv.width = 1;
    }
    else
    {
        if (!rd.int_vector(bv, 1) || !rd.skip_int_vector() /* bv_rank */)
            goto bad;
        if (layout == 1 && (!rd.skip_select_mcl() || !rd.skip_select_mcl()))
            goto bad;
    }
    if (!rd.u64(n_nodes64) || n_nodes64 >= (uint64_t)kWtMaxNodes)
        goto bad;
    {
        WtTables & T = wt.tables;
        tables_clear(T);
        wt.n_nodes = (uint32_t)n_nodes64;
        for (uint32_t v = 0; v < wt.n_nodes; ++v)
        { // _node::serialize wt_helper.hpp:139-150 — 22 bytes
            uint16_t par, ch[2];
            if (!rd.u64(T.bv_pos[v]) || !rd.u64(T.bv_pos_rank[v]) || !rd.u16(par) || !rd.raw(ch, 4))
                goto bad;
            T.parent[v] = par;
            T.child[v][0] = ch[0];
            T.child[v][1] = ch[1];
            bool leaf = ch[0] == kWtUndef;
            if ((!leaf && (ch[0] >= wt.n_nodes || ch[1] >= wt.n_nodes)) || (v > 0 && par >= wt.n_nodes)
                || T.bv_pos[v] > bv.bit_size)
                goto bad;
        }
        uint16_t c2l[256];
        uint64_t path[256];
        if (!rd.raw(c2l, sizeof c2l) || !rd.raw(path, sizeof path))
            goto bad;
        if (wt.size != 0)
        { // a default-constructed (empty) wt_pc serialises uninitialised tables: ignore them
            for (int c = 0; c < 256; ++c)
            {
                if (c2l[c] != kWtUndef && (c2l[c] >= wt.n_nodes || (path[c] >> 56) > 56))
                    goto bad;
                T.c_to_leaf[c] = c2l[c];
                T.path[c] = path[c];
            }
        }
        // Structural validation (the kernels trust these tables): BFS numbering (children come after their
        // parent, so every walk terminates), contiguous slices covering the bit vector, every child's slice as
        // long as the number of zeros/ones of its parent's slice, stored prefix ranks equal to the real ones.
        // Along the way: occurrences per symbol (zeros go left, ones go right).
        memset(wt.occ, 0, sizeof wt.occ);
        if (wt.size != 0 && wt.n_nodes == 0)
            goto bad;
        if (wt.size != 0)
        {
            std::vector<uint64_t> slice(wt.n_nodes, 0), expect(wt.n_nodes, UINT64_MAX);
            expect[0] = wt.size;
            uint64_t run_pos = 0, run_rank = 0, leaves = 0;
            for (uint32_t v = 0; v < wt.n_nodes; ++v)
            {
                const bool leaf = T.child[v][0] == kWtUndef;
                if (T.bv_pos[v] != run_pos || expect[v] == UINT64_MAX)
                    goto bad;
                if (leaf)
                {
                    if (T.child[v][1] != kWtUndef || T.bv_pos_rank[v] > 255)
                        goto bad;
                    uint8_t sym = (uint8_t)T.bv_pos_rank[v];
                    if (T.c_to_leaf[sym] != v)
                        goto bad;
                    wt.occ[sym] = expect[v];
                    ++leaves;
                    continue;
                }
                if (T.child[v][0] <= v || T.child[v][1] <= v || T.child[v][1] == kWtUndef
                    || T.parent[T.child[v][0]] != v || T.parent[T.child[v][1]] != v)
                    goto bad;
                slice[v] = expect[v];
                if (slice[v] > bv.bit_size - run_pos || T.bv_pos_rank[v] != run_rank)
                    goto bad;
                uint64_t ones = popcount_range(bv.words.data(), run_pos, run_pos + slice[v]);
                expect[T.child[v][0]] = slice[v] - ones;
                expect[T.child[v][1]] = ones;
                run_pos += slice[v];
                run_rank += ones;
            }
            if (run_pos != bv.bit_size || leaves != wt.sigma)
                goto bad;
            for (int c = 0; c < 256; ++c)
            { // paths must lead from the root to the symbol's leaf
                if (T.c_to_leaf[c] == kWtUndef)
                    continue;
                uint64_t p = T.path[c];
                unsigned len = (unsigned)(p >> 56), v = 0;
                for (unsigned l = 0; l < len; ++l, p >>= 1)
                {
                    if (T.child[v][0] == kWtUndef)
                        goto bad;
                    v = T.child[v][p & 1];
                }
                if (v != T.c_to_leaf[c] || T.bv_pos_rank[v] != (uint64_t)c)
                    goto bad;
            }
        }
        if (wt.backend == 1)
        { // the rrr vector is already on the device; only the node tables remain
            wt.device = device;
            SH_TRY(wt.d_tables.alloc(sizeof(WtTables)));
            SH_HIP(hipMemcpy(wt.d_tables.p, &wt.tables, sizeof(WtTables), hipMemcpyHostToDevice));
            return SDSL_HIP_OK;
        }
        return upload(wt, bv.words, bv.bit_size, device);
    }
bad:
    set_error("malformed wt_huff stream (offset %zu of %zu)", rd.pos, rd.len);
    return SDSL_HIP_ERR_FORMAT;
}

// =========================================================================================
// kernels
// =========================================================================================

template <bool NT>
__global__ __launch_bounds__(kBlock) void k_wt_rank(WtView wt, const uint64_t * __restrict__ iq,
                                                    const uint8_t * __restrict__ cq, uint64_t * __restrict__ out,
                                                    uint64_t n)
{
    __shared__ WtTables T;
    __shared__ WtFusedTables FT;
    wt_stage_tables(&T, wt.tables);
    wt_stage_fused(&FT, wt);
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n; base += (uint64_t)gridDim.x * kQPB)
    {
        uint64_t q = base + gq;
        if (q >= n)
            continue;
        uint64_t i = __builtin_nontemporal_load(iq + q); // streamed once: keep the caches for the tree
        unsigned c = __builtin_nontemporal_load(cq + q);
        uint64_t r = SDSL_HIP_NPOS;
        if (i <= wt.size)
            r = wt.f_lines ? quad_wt8_rank<NT>(wt, &T, &FT, s, i, c) : quad_wt_rank<NT>(wt, &T, s, i, c);
        if (s == 0)
            __builtin_nontemporal_store(r, out + q);
    }
}

// ---- flat loops over a batch: which query a quad takes next --------------------------------------------------------------
// A wave works through chunks of kFlatChunkQ consecutive queries (chunk w, w + waves, ...); the quads that are out of work take the
// chunk's next queries in quad order, so what the quads of one iteration fetch are neighbours in the argument arrays, and a line of
// arguments is used up by the wave that first touched it.  (Handing every quad its own strided sequence reads each line of arguments
// once per quad: 10 G/s instead of 25.)
constexpr uint32_t kFlatChunkQ = 256;
struct WaveChunks
{
    uint64_t chunk, n_chunks, step, n, cur = 0, cur_end = 0;
    uint64_t quads_below;
    bool drained = false;
    __device__ __forceinline__ WaveChunks(uint64_t n_)
    {
        n = n_;
        n_chunks = (n_ + kFlatChunkQ - 1) / kFlatChunkQ;
        step = (uint64_t)gridDim.x * (blockDim.x / 64);
        chunk = (uint64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        const int lane = threadIdx.x & 63;
        quads_below = UINT64_C(0x1111111111111111) & ((UINT64_C(1) << (lane & ~3)) - 1); // leaders of the quads in front
    }
    // quads with `wants` (quad-uniform) get the index of their next query; false: none (for now, or for good when drained)
    __device__ __forceinline__ bool claim(bool wants, int s, uint64_t & q_out)
    {
        const uint64_t want = __ballot(wants && s == 0);
        bool got = false;
        if (want && !drained)
        {
            if (cur == cur_end)
            {
                if (chunk >= n_chunks)
                    drained = true;
                else
                {
                    cur = chunk * kFlatChunkQ;
                    cur_end = cur + kFlatChunkQ < n ? cur + kFlatChunkQ : n;
                    chunk += step;
                }
            }
            if (!drained)
            {
                const uint64_t avail = cur_end - cur, rank = (uint64_t)__popcll(want & quads_below);
                if (wants && rank < avail)
                {
                    q_out = cur + rank;
                    got = true;
                }
                const uint64_t wanted = (uint64_t)__popcll(want);
                cur += wanted < avail ? wanted : avail;
            }
        }
        return got;
    }
};

// rank(i, c) on the fused lines as a FLAT loop: one iteration is one fused step of whatever query the quad holds, and a quad that is
// done takes its next query at once (its arguments fetched an iteration ahead).  Code lengths differ — the bench text: 1.23 steps on
// average, three at most — and the loop above makes a wave wait for the longest of its sixteen; here nobody waits.  The path of c comes
// from the per-symbol step table (wt_device.hpp: WtStepTab, 6 KiB of LDS instead of the node tables' 16).
__global__ __launch_bounds__(kBlock) void k_wt_rank_flat(WtView wt, const uint64_t * __restrict__ iq, const uint8_t * __restrict__ cq,
                                                         uint64_t * __restrict__ out, uint64_t n)
{
    __shared__ WtStepTab ST;
    {
        const uint64_t * src = reinterpret_cast<const uint64_t *>(wt.f_steps);
        uint64_t * dst = reinterpret_cast<uint64_t *>(&ST);
        for (unsigned k = threadIdx.x; k < sizeof(WtStepTab) / 8; k += blockDim.x)
            dst[k] = src[k];
        __syncthreads();
    }
    const int s = threadIdx.x & (kG - 1);
    const bool wide = (wt.size >> 32) != 0; // kernel-uniform
    WaveChunks wc(n);
    bool have = false, nx = false;
    uint64_t q = 0, res = 0, q_n = 0, i_n = 0;
    unsigned c_n = 0;
    uint32_t si = 0, left = 0;
    for (;;)
    {
        // 1. quads without a next query claim one and request its arguments
        if (wc.claim(!nx, s, q_n))
        {
            i_n = iq[q_n];
            c_n = cq[q_n];
            nx = true;
        }
        // 2. one fused step
        if (have)
        {
            const uint32_t st = ST.steps[si];
            ++si;
            --left;
            const uint32_t base = st & 0x0FFFFFFFu, t = st >> 28;
            const uint64_t li = fused_line(res), L = base + li;
            const FSec x = load_fsec<false>(wt.f_lines, L, s);
            const uint64_t sup = fused_super(wt.f_super, wide, base, L, t);
            res = sup + quad_sum(fsec_count(x, s, fused_off(res, li), t)); // (8-ary lines: fewer than 2^32 symbols here, sup = 0)
            if (left == 0 || res == 0)
            { // (wt_pc.hpp:386: a count of 0 stays 0)
                if (s == 0)
                    __builtin_nontemporal_store(res, out + q);
                have = false;
            }
        }
        // 3. a quad that is free takes the query it has requested; the ones that need no walk are answered on the spot
        if (!have && nx)
        {
            nx = false;
            q = q_n;
            const uint32_t meta = ST.meta[c_n];
            if (i_n <= wt.size && meta != 0 && i_n != 0)
            {
                res = i_n;
                si = meta & 0xFFFFu;
                left = meta >> 16;
                have = true;
            }
            else if (s == 0) // past the end: NPOS; c does not occur (wt_pc.hpp:374-377) or nothing in front of position 0: 0
                __builtin_nontemporal_store(i_n > wt.size ? (uint64_t)SDSL_HIP_NPOS : UINT64_C(0), out + q);
        }
        if (wc.drained && !__any(have || nx))
            break;
    }
}

// inverse_select / operator[] on the fused lines as the same flat loop, walked by fused node (wt_device.hpp: WtFusedWalk)
template <bool WITH_RANK>
__global__ __launch_bounds__(kBlock) void k_wt_inverse_select_flat(WtView wt, const uint64_t * __restrict__ iq, uint64_t * __restrict__ out_rank,
                                                                   uint8_t * __restrict__ out_c, uint64_t n)
{
    __shared__ WtFusedWalk W;
    {
        const uint64_t * src = reinterpret_cast<const uint64_t *>(wt.f_walk);
        uint64_t * dst = reinterpret_cast<uint64_t *>(&W);
        for (unsigned k = threadIdx.x; k < sizeof(WtFusedWalk) / 8; k += blockDim.x)
            dst[k] = src[k];
        __syncthreads();
    }
    const int s = threadIdx.x & (kG - 1);
    WaveChunks wc(n);
    bool have = false, nx = false;
    uint64_t q = 0, i = 0, q_n = 0, i_n = 0;
    unsigned r = 0;
    for (;;)
    {
        if (wc.claim(!nx, s, q_n))
        {
            i_n = iq[q_n];
            nx = true;
        }
        if (have)
        {
            const unsigned e = quad_wtf_invsel_step<false>(wt, &W, s, r, i);
            r = e;
            if (e & kFWalkLeaf)
            {
                if (s == 0)
                {
                    __builtin_nontemporal_store((uint8_t)(e & 0xFFu), out_c + q);
                    if (WITH_RANK)
                        __builtin_nontemporal_store(i, out_rank + q);
                }
                have = false;
            }
        }
        if (!have && nx)
        {
            nx = false;
            q = q_n;
            i = i_n;
            if (i < wt.size)
            {
                r = 0;
                have = true;
            }
            else if (s == 0)
            { // outside the sequence
                __builtin_nontemporal_store((uint8_t)0xFF, out_c + q);
                if (WITH_RANK)
                    __builtin_nontemporal_store((uint64_t)SDSL_HIP_NPOS, out_rank + q);
            }
        }
        if (wc.drained && !__any(have || nx))
            break;
    }
}

// operator[] and inverse_select share one traversal
template <bool NT, bool WITH_RANK>
__global__ __launch_bounds__(kBlock) void k_wt_inverse_select(WtView wt, const uint64_t * __restrict__ iq,
                                                              uint64_t * __restrict__ out_rank,
                                                              uint8_t * __restrict__ out_c, uint64_t n)
{
    __shared__ WtTables T;
    __shared__ WtFusedTables FT;
    wt_stage_tables(&T, wt.tables);
    wt_stage_fused(&FT, wt);
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n; base += (uint64_t)gridDim.x * kQPB)
    {
        uint64_t q = base + gq;
        if (q >= n)
            continue;
        uint64_t i = __builtin_nontemporal_load(iq + q);
        unsigned c = 0xFF;
        uint64_t r = SDSL_HIP_NPOS;
        if (i < wt.size)
            r = quad_wt_inverse_select<NT>(wt, &T, &FT, s, i, c);
        if (s == 0)
        {
            __builtin_nontemporal_store((uint8_t)c, out_c + q);
            if (WITH_RANK)
                __builtin_nontemporal_store(r, out_rank + q);
        }
    }
}


// wt_pc::select (wt_pc.hpp:443-474): leaf-to-root, one select on the parent's slice per level (§3.2 of DESIGN.md).
// A query is a chain of L(c) dependent selects, each a directory read plus one or more window probes, and both L(c)
// and the number of probes differ from quad to quad.  The loop is therefore FLAT and persistent: one iteration is ONE
// window probe of whatever level of whatever query the quad is at; a quad that finishes a query takes its next one at
// once instead of waiting for the slowest of the wave's 16 (nested loops cost max(levels) x max(probes) latencies per
// query instead of their sum).  The kernel is VALU-bound (90 % busy), hence sel_eval_rt: one evaluation for ones and
// zeros instead of two divergent instantiations.  1 GiB text: 1.9 -> 4.05 G select/s.
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_wt_select(WtView wt, const uint64_t * __restrict__ occ,
                                                      const uint64_t * __restrict__ iq,
                                                      const uint8_t * __restrict__ cq, uint64_t * __restrict__ out,
                                                      uint64_t n)
{
    __shared__ WtTables T;
    __shared__ uint64_t occ_s[256];
    occ_s[threadIdx.x & 255] = occ[threadIdx.x & 255];
    wt_stage_tables(&T, wt.tables); // ends with __syncthreads()
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    const uint64_t stride = (uint64_t)gridDim.x * kQPB;
    uint64_t q_next = (uint64_t)blockIdx.x * kQPB + gq; // this quad's next query, loaded one query ahead
    uint64_t i_nxt = q_next < n ? iq[q_next] : 0;
    unsigned c_nxt = q_next < n ? cq[q_next] : 0;
    bool have = false;
    uint64_t q = 0, res = 0, p = 0, k = 0;
    unsigned left = 0, v = 0, par = 0, bit = 0;
    int tries = 0;
    SelBracket br{};
    // the select that takes child v's offset `res` up into its parent: argument index and directory bracket
    auto start_level = [&]() {
        par = T.parent[v];
        bit = (unsigned)(p >> 63); // v is a right child: ones of the parent's slice; left child: zeros
        k = bit ? T.bv_pos_rank[par] + res : T.bv_pos[par] - T.bv_pos_rank[par] + res;
        const uint64_t j = k >> wt.bv.sel_shift;
        const uint32_t * smp = wt.bv.sel[bit];
        const uint32_t s0 = smp[j], s1 = smp[j + 1];
        br = sel_bracket<1>(wt.bv, k, s0, s1); // positions and counts of the two samples ...
        const uint64_t total = bit ? wt.bv.ones : wt.bv.n_bits - wt.bv.ones;
        const uint64_t top = ((k >> wt.bv.sel_shift) + 1) << wt.bv.sel_shift;
        br.hi_cnt = top > total ? total : top; // ... with the last bracket clamped to this bit value's argument count
        tries = 0;
    };
    for (;;)
    {
        while (!have && q_next < n)
        { // next query of this quad; the ones that need no walk are answered on the spot
            q = q_next;
            q_next += stride;
            const uint64_t i = i_nxt;
            const unsigned c = c_nxt;
            if (q_next < n)
            { // in flight while this query walks
                i_nxt = iq[q_next];
                c_nxt = cq[q_next];
            }
            v = T.c_to_leaf[c];
            uint64_t direct = 0;
            bool walk = false;
            if (v == kWtUndef)
                direct = wt.size; // c not in the text (wt_pc.hpp:447-450)
            else if (i == 0 || i > occ_s[c])
                direct = SDSL_HIP_NPOS; // outside SDSL's precondition
            else if (wt.sigma == 1)
                direct = i - 1 < wt.size ? i - 1 : wt.size;
            else
                walk = true;
            if (!walk)
            {
                if (s == 0)
                    out[q] = direct;
                continue;
            }
            res = i - 1;
            p = T.path[c];
            left = (unsigned)(p >> 56);
            p <<= (64 - left); // deepest level first
            have = true;
            start_level();
        }
        if (__ballot(have) == 0)
            break; // every quad of the wave is out of queries
        if (have)
        {
            const uint64_t W = sel_guess(wt.bv, br, k, tries);
            const Pair wa = load_pair<NT>(wt.bv.lines, 2 * W, s);
            const Pair wb = load_pair<NT>(wt.bv.lines, 2 * W + 1, s);
            bool mine = false;
            uint64_t pos = 0;
            const bool hit = sel_eval_rt(wt.bv, s, bit != 0, k, W, wa, wb, br, mine, pos);
            if (hit)
            {
                res = quad_gather_u64(pos, mine) - T.bv_pos[par];
                v = par;
                p <<= 1;
                if (--left == 0)
                {
                    if (s == 0)
                        out[q] = res;
                    have = false;
                }
                else
                    start_level();
            }
            else
                ++tries;
        }
    }
}

// wt_pc::select on the fused layout: the same flat persistent loop, but one iteration is one probe of a FUSED step
// (three tree levels): the path of c is cut into groups of three levels from the root, and the walk goes from the
// deepest group up, each group one select of its slot inside the fused node above it.
// (WIDE: a sequence of 2^32 symbols and more on 16-ary lines — positions, counts and the directory's entries are 64-bit)
template <bool NT, bool WIDE>
__global__ __launch_bounds__(kBlock) void k_wt_select_fused(WtView wt, const uint64_t * __restrict__ occ,
                                                            const uint64_t * __restrict__ iq,
                                                            const uint8_t * __restrict__ cq, uint64_t * __restrict__ out,
                                                            uint64_t n)
{
    // only the parts of the node table this walk reads (parent links, paths, leaves): 3.5 KiB instead of 13.5 — LDS is
    // what decides how many waves a CU holds here
    __shared__ struct
    {
        uint64_t path[256];
        uint16_t parent[kWtMaxNodes];
        uint16_t c_to_leaf[256];
    } T;
    __shared__ WtFusedTables FT;
    __shared__ WtFusedSelTables FS;
    __shared__ uint64_t occ_s[256];
    occ_s[threadIdx.x & 255] = occ[threadIdx.x & 255];
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
    }
    wt_stage_fused(&FT, wt); // ends with __syncthreads()
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    const uint64_t stride = (uint64_t)gridDim.x * kQPB;
    uint64_t q_next = (uint64_t)blockIdx.x * kQPB + gq; // this quad's next query, loaded one query ahead
    uint64_t i_nxt = q_next < n ? iq[q_next] : 0;
    unsigned c_nxt = q_next < n ? cq[q_next] : 0;
    bool have = false;
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type pos_t;
    const pos_t * dir = reinterpret_cast<const pos_t *>(wt.f_sel);
    uint64_t q = 0, p = 0, base_line = 0;
    pos_t res = 0;
    unsigned groups = 0, len = 0, cur = 0, t = 0;
    int tries = 0;
    FselBracketT<pos_t> br{};
    // the fused step that takes node `cur`'s offset `res` up into the fused node above it
    auto start_group = [&]() {
        const unsigned g = groups - 1, nlev = len - kFK * g < kFK ? len - kFK * g : kFK;
        t = (unsigned)(p >> (kFK * g)) & ((1u << nlev) - 1u);
        unsigned u = cur;
        for (unsigned k = 0; k < nlev; ++k)
            u = T.parent[u];
        cur = u;
        len = kFK * g; // levels above u
        base_line = FT.fline[u];
        const unsigned rid = FS.root_id[u];
        pos_t total = FS.cnt[rid][t];
        if constexpr (WIDE)
            total |= (pos_t)FS.cnt_hi[rid][t] << 32;
        br = fsel_bracket_t<pos_t>(dir, FS.off[rid][t], res, total);
        tries = 0;
    };
    for (;;)
    {
        while (!have && q_next < n)
        { // next query of this quad; the ones that need no walk are answered on the spot
            q = q_next;
            q_next += stride;
            const uint64_t i = i_nxt;
            const unsigned c = c_nxt;
            if (q_next < n)
            { // in flight while this query walks
                i_nxt = iq[q_next];
                c_nxt = cq[q_next];
            }
            cur = T.c_to_leaf[c];
            uint64_t direct = 0;
            bool walk = false;
            if (cur == kWtUndef)
                direct = wt.size; // c not in the text (wt_pc.hpp:447-450)
            else if (i == 0 || i > occ_s[c])
                direct = SDSL_HIP_NPOS; // outside SDSL's precondition
            else if (wt.sigma == 1)
                direct = i - 1 < wt.size ? i - 1 : wt.size;
            else
                walk = true;
            if (!walk)
            {
                if (s == 0)
                    out[q] = direct;
                continue;
            }
            res = (pos_t)(i - 1);
            p = T.path[c];
            len = (unsigned)(p >> 56);
            groups = (len + kFK - 1) / kFK;
            have = true;
            start_group();
        }
        if (__ballot(have) == 0)
            break; // every quad of the wave is out of queries
        if (have)
        {
            uint64_t pos;
            if (quad_fsel_probe_t<NT, pos_t>(wt, base_line, s, t, res, br, tries, pos))
            {
                res = (pos_t)pos;
                if (--groups == 0)
                {
                    if (s == 0)
                        out[q] = res;
                    have = false;
                }
                else
                    start_group();
            }
            else
                ++tries;
        }
    }
}

// ---- construction of the fused layout (wt_device.hpp) from the binary tree ---------------------------------------
// planes: a wave handles 64 consecutive positions of node u's sequence, one lane each.  Consecutive positions of a node
// stay consecutive inside each child, so a lane's offset one level down = (rank at the group's first position, the
// same for all lanes that took the same branch) + (number of earlier lanes on that branch, from ballots): two ranks at
// wave-uniform-per-branch positions and three single-bit reads per lane instead of three full ranks.
__device__ __forceinline__ unsigned bv_bit(const uint64_t * lines, uint64_t pos)
{
    const uint64_t L = pos / kDB;
    const unsigned off = (unsigned)(pos - L * kDB);
    return (unsigned)((lines[L * kLW + 1 + (off >> 6)] >> (off & 63)) & 1);
}

__global__ __launch_bounds__(256) void k_wt8_planes(WtView wt, unsigned u, uint64_t size_u, uint64_t * __restrict__ fl)
{
    __shared__ WtTables T;
    wt_stage_tables(&T, wt.tables);
    const unsigned lane = threadIdx.x & 63;
    const uint64_t below = (UINT64_C(1) << lane) - 1; // lanes before this one
    // a wave takes one section of a line: kFLane consecutive positions (64, or 46 of the 16-ary form: the other lanes idle)
    const uint64_t n_groups = (size_u + kFLane - 1) / kFLane;
    const uint64_t * lines = wt.bv.lines;
    for (uint64_t g = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6; g < n_groups; g += (uint64_t)gridDim.x * 4)
    { // wave-uniform trip count
        const uint64_t j0 = g * kFLane, j = j0 + lane;
        bool in = lane < kFLane && j < size_u; // still at an inner node
        uint64_t same = __ballot(in);           // the lanes that took the same branches as this one
        uint64_t start = j0, i = j;             // offset inside node v of the group's first symbol on this lane's path / of this lane's symbol
        unsigned v = u;
        uint64_t planes[kFK];
#pragma unroll
        for (unsigned k = 0; k < kFK; ++k)
        {
            unsigned bit = 0;
            if (in)
                bit = bv_bit(lines, T.bv_pos[v] + i);
            const uint64_t m = __ballot(bit);
            planes[k] = m;
            if (k + 1 < kFK && in)
            { // ones in front of the path's first symbol: the same for all lanes on this path
                const uint64_t r = lane_rank1(lines, T.bv_pos[v] + start, nullptr) - T.bv_pos_rank[v];
                start = bit ? r : start - r;
                same &= bit ? m : ~m;
                i = start + (uint64_t)__popcll(same & below);
                v = T.child[v][bit];
                in = T.child[v][0] != kWtUndef;
            }
        }
        if (lane == 0)
        {
            uint64_t * sec = fl + (g >> 2) * kFusedWords + (g & 3) * 4;
            if constexpr (kFK == 3)
            {
                sec[1] = planes[0];
                sec[2] = planes[1];
                sec[3] = planes[2];
            }
            else
            { // three words of 16 positions, a 16-bit field per plane (wt_device.hpp: fsec16_pack)
                uint64_t pl[4] = {planes[0], planes[1], planes[2], planes[kFK - 1]};
                sec[1] = fsec16_pack(pl, 0);
                sec[2] = fsec16_pack(pl, 1);
                sec[3] = fsec16_pack(pl, 2);
            }
        }
    }
}

// position i of node u taken down slot t's bits: the slot's count in front of i (ok = false: no such slot under u)
__device__ __forceinline__ uint64_t wt8_cascade(const WtView & wt, const WtTables * T, unsigned u, uint64_t i, unsigned t, bool & ok)
{
    unsigned v = u;
    ok = true;
    for (unsigned k = 0; k < kFK; ++k)
    {
        if (T->child[v][0] == kWtUndef)
        { // a leaf above the last level: its slot is the path padded with zeros
            ok = (t >> k) == 0;
            break;
        }
        const unsigned bit = (t >> k) & 1;
        const uint64_t r = lane_rank1(wt.bv.lines, T->bv_pos[v] + i, nullptr) - T->bv_pos_rank[v];
        i = bit ? r : i - r;
        v = T->child[v][bit];
    }
    return i;
}

// 16-ary lines, first: the superblock records of node u — thread (superblock, t) cascades the first position of every superblock that
// starts inside the node (the one the node starts in counts from the node's start and has no record of its own)
__global__ __launch_bounds__(256) void k_wt8_supers(WtView wt, unsigned u, uint64_t n_lines_u, uint64_t first_line, uint32_t * __restrict__ sup, bool wide)
{
    const WtTables * T = wt.tables;
    const uint64_t sb0 = (first_line >> kFSuperLog) + 1, sb1 = (first_line + n_lines_u - 1) >> kFSuperLog; // records sb0 .. sb1
    if (sb1 < sb0)
        return;
    const uint64_t n = (sb1 - sb0 + 1) * kFSlots;
    for (uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (uint64_t)gridDim.x * 256)
    {
        const uint64_t sb = sb0 + id / kFSlots;
        const unsigned t = (unsigned)id & (kFSlots - 1);
        bool ok;
        const uint64_t c = wt8_cascade(wt, T, u, ((sb << kFSuperLog) - first_line) * kFusedPos, t, ok);
        const uint64_t at = sb * kFSlots + t;
        if (wide)
            reinterpret_cast<uint64_t *>(sup)[at] = ok ? c : UINT64_C(0);
        else
            sup[at] = ok ? (uint32_t)c : 0u;
    }
}

// counts: thread (line, t) cascades the line's first position of node u down slot t's bits.  16-ary: the header holds the count
// relative to the first line of the line's superblock (wt_device.hpp: fused_super; superblocks are counted in absolute lines,
// `first_line` = the node's first), whose own count k_wt8_supers has written; in the superblock
// the node starts in, the counts are the node's own.  fl = the node's first line, sup = the whole table (64-bit records if `wide`).
__global__ __launch_bounds__(256) void k_wt8_counts(WtView wt, unsigned u, uint64_t n_lines_u, uint64_t * __restrict__ fl, uint64_t first_line,
                                                    uint32_t * __restrict__ sup, bool wide)
{
    const WtTables * T = wt.tables;
    // grid-stride: the launch caps its grid (2^20 blocks = 2^25 lines = 2^33 symbols per pass), a node of a 2^36-symbol sequence has more
    for (uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x; id < n_lines_u * kFSlots; id += (uint64_t)gridDim.x * 256)
    {
        const uint64_t line = id / kFSlots;
        const unsigned t = (unsigned)id & (kFSlots - 1);
        bool ok;
        const uint64_t c = wt8_cascade(wt, T, u, line * kFusedPos, t, ok);
        if constexpr (kFK == 3)
            reinterpret_cast<uint32_t *>(fl + line * kFusedWords + 4 * (t >> 1))[t & 1] = ok ? (uint32_t)c : 0u;
        else
        {
            const uint64_t abs_line = first_line + line, sabs = abs_line & ~(uint64_t)((1u << kFSuperLog) - 1); // the superblock's first line
            uint64_t cs = 0; // (the superblock the node starts in: counted from the node's start)
            if (sabs > first_line)
            { // the superblock's own count: k_wt8_supers has written it (round 5 cascaded it again in every one of the superblock's
              // 2^kFSuperLog lines: twice the rank walks of the whole pass)
                const uint64_t at = (abs_line >> kFSuperLog) * kFSlots + t;
                cs = wide ? reinterpret_cast<const uint64_t *>(sup)[at] : (uint64_t)sup[at];
            }
            const uint32_t rel = ok ? (uint32_t)(c - cs) : 0u; // < 2^(16 + kFSpare); modulo 2^32 when the record holds the low word only
            uint64_t * sec = fl + line * kFusedWords + 4 * (t >> 2);
            reinterpret_cast<uint16_t *>(sec)[t & 3] = (uint16_t)rel;
            if (kFSpare != 0 && (rel >> 16)) // the top bits of field t & 3 of the section's third word (k_wt8_planes has written the word)
                atomicOr(reinterpret_cast<unsigned long long *>(sec + 3), (unsigned long long)(rel >> 16) << (16 * (t & 3) + 16 - kFSpare));
        }
    }
}

// Sequences of 2^32 symbols and more: the headers hold the low 32 bits of their counts; a count that reaches a multiple of 2^32
// inside a line shows as a next header (for the node's last line: the slot's total) smaller than this one — a line adds at most 256.
// Thread (line, t) of node u notes the PLACE (absolute line << 8 | offset) at which the count is first j * 2^32 or more: one past the
// occurrence that completes the multiple (WtFusedTables::cross_*).
struct FcrossArgs
{
    uint64_t total[kFSlots]; // occurrences of every slot in the node
};
__global__ __launch_bounds__(256) void k_wt8_cross(const uint64_t * __restrict__ fl, uint64_t n_lines_u, uint32_t first_line, uint32_t u,
                                                   FcrossArgs a, uint32_t * __restrict__ n_out, uint64_t * __restrict__ pos_out,
                                                   uint32_t * __restrict__ key_out)
{
    for (uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x; id < n_lines_u * 8; id += (uint64_t)gridDim.x * 256)
    { // (grid-stride, as k_wt8_counts)
        const uint64_t line = id >> 3;
        const unsigned t = (unsigned)id & 7u;
        const uint64_t * ln = fl + line * kFusedWords;
        const uint32_t c0 = reinterpret_cast<const uint32_t *>(ln + 4 * (t >> 1))[t & 1];
        const uint32_t c1 = line + 1 < n_lines_u ? reinterpret_cast<const uint32_t *>(ln + kFusedWords + 4 * (t >> 1))[t & 1] : (uint32_t)a.total[t];
        if (c1 >= c0 || (line + 1 == n_lines_u && a.total[t] < (UINT64_C(1) << 32)))
            continue; // (c1 < c0 in the last line of a slot with fewer than 2^32 occurrences cannot happen; the test keeps junk totals out)
        uint32_t r = 0u - c0; // occurrences of t the line must add to reach the multiple: 1 .. 256
        unsigned off = 256;
        for (unsigned g = 0; g < 4; ++g)
        {
            const uint64_t p0 = ln[4 * g + 1], p1 = ln[4 * g + 2], p2 = ln[4 * g + 3];
            const uint64_t m = ((t & 1) ? p0 : ~p0) & ((t & 2) ? p1 : ~p1) & ((t & 4) ? p2 : ~p2);
            const uint32_t c = popc64(m);
            if (r <= c)
            {
                off = 64u * g + sel64(m, r) + 1u;
                break;
            }
            r -= c;
        }
        const uint32_t e = atomicAdd(n_out, 1u);
        if (e < kFusedMaxCross)
        {
            pos_out[e] = (((uint64_t)first_line + line) << 8) + off; // (8-ary lines only: 256 positions)
            key_out[e] = (u << 3) | t;
        }
    }
}

// select directory of one fused node: thread (line, t) knows which occurrences of t fall into its line from two
// neighbouring headers; if occurrence 256 * j is among them it finds its position in the line's match masks
struct FselNodeArgs
{
    uint64_t cnt[kFSlots];
    uint32_t off[kFSlots], n_samples[kFSlots];
    uint64_t size;
};
// the absolute count of slot t in front of line `line` of a node (fl: the node's first line, `first_line` its index, sup: the whole table)
template <bool WIDE>
__device__ __forceinline__ uint64_t wt8_header_abs(const uint64_t * fl, const uint32_t * sup, uint64_t first_line, uint64_t line, unsigned t)
{
    if constexpr (kFK == 3)
        return reinterpret_cast<const uint32_t *>(fl + line * kFusedWords + 4 * (t >> 1))[t & 1];
    else
    {
        const uint64_t * sec = fl + line * kFusedWords + 4 * (t >> 2);
        return fused_super(sup, WIDE, first_line, first_line + line, t) + fsec16_count_field(sec[0], sec[3], t & 3);
    }
}
// the positions of section g of a line that hold slot t
__device__ __forceinline__ uint64_t wt8_section_match(const uint64_t * ln, unsigned g, unsigned t)
{
    return fsec_match_words(ln[4 * g + 1], ln[4 * g + 2], ln[4 * g + 3], t);
}
// (WIDE: 2^32 symbols and more on 16-ary lines: 64-bit directory entries; grid-stride: a node of such a sequence has more lines than a grid covers)
template <bool WIDE>
__global__ __launch_bounds__(256) void k_wt8_sel_dir(const uint64_t * __restrict__ fl, const uint32_t * __restrict__ sup, uint64_t first_line,
                                                     uint64_t n_lines, FselNodeArgs a, void * __restrict__ dir_)
{
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type pos_t;
    pos_t * dir = reinterpret_cast<pos_t *>(dir_);
    for (uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x; id < n_lines * kFSlots; id += (uint64_t)gridDim.x * 256)
    {
        const uint64_t line = id / kFSlots;
        const unsigned t = (unsigned)id & (kFSlots - 1);
        if (a.off[t] == kFselNone)
            continue;
        const uint64_t * ln = fl + line * kFusedWords;
        const pos_t c0 = (pos_t)wt8_header_abs<WIDE>(fl, sup, first_line, line, t);
        const pos_t c1 = line + 1 < n_lines ? (pos_t)wt8_header_abs<WIDE>(fl, sup, first_line, line + 1, t) : (pos_t)a.cnt[t];
        constexpr pos_t S = (pos_t)1 << kFselLog;
        for (pos_t j = (c0 + S - 1) >> kFselLog; c1 > c0 && (j << kFselLog) < c1; ++j)
        {
            uint32_t r = (uint32_t)((j << kFselLog) - c0); // rank of the wanted occurrence inside the line
            for (unsigned g = 0; g < 4; ++g)
            {
                const uint64_t m = wt8_section_match(ln, g, t);
                const uint32_t c = popc64(m);
                if (r < c)
                {
                    dir[a.off[t] + j] = (pos_t)(line * kFusedPos) + kFLane * g + sel64(m, r + 1);
                    break;
                }
                r -= c;
            }
        }
        if (line + 1 == n_lines)
            dir[a.off[t] + a.n_samples[t] - 1] = (pos_t)a.size;
    }
}

// the symbol sequence of a tree (wt[0 .. size)), read back through the binary levels (view_binary()) or the fused lines (view())
__global__ __launch_bounds__(kBlock) void k_wt_export_symbols(WtView wt, uint8_t * __restrict__ out, uint64_t n)
{
    __shared__ WtTables T;
    __shared__ WtFusedTables FT;
    wt_stage_tables(&T, wt.tables);
    wt_stage_fused(&FT, wt);
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n; base += (uint64_t)gridDim.x * kQPB)
    {
        const uint64_t q = base + gq;
        if (q >= n)
            continue;
        unsigned c = 0;
        quad_wt_inverse_select<false>(wt, &T, &FT, s, q, c);
        if (s == 0)
            out[q] = (uint8_t)c;
    }
}

// ---- the tree without its binary levels (sdsl_hip_fm_set_footprint) --------------------------------------------------
// rank, access, inverse_select, the LF walks, count and — with its directory — select walk the fused lines only; SDSL's binary
// levels (rank lines + both select directories: 1.27 bits per tree bit) are needed for serialisation and for select on a tree
// without the fused directory.  They can be released and are rebuilt from the fused lines when one of those is asked for: the
// symbol sequence is read back (k_wt_export_symbols over view()), and the level builder runs on it with the tree's own node
// table (kWtShapeGiven), so trees of any shape (loaded wt_blcd / wt_hutu streams too) come back bit for bit.
sdsl_hip_status wt_drop_binary(WtHost & wt)
{
    if (wt.binary_dropped)
        return SDSL_HIP_OK;
    if (wt.backend != 0 || !wt.d_fused.p || !wt.d_ftables.p)
    {
        set_error("the binary levels can only be released on a plain tree that has its fused layout (fewer than 2^36 symbols, "
                  "SDSL_HIP_WT_FUSED not 0)");
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    BvHost & b = wt.bv;
    std::lock_guard<std::mutex> lk(b.scratch_mutex);
    b.lines.release();
    b.cnts.release();
    for (int i = 0; i < 2; ++i)
    {
        b.sel[i].release();
        b.lmask[i].release();
        b.lidx[i].release();
        b.lpos[i].release();
        b.sel_plan[i].bnd.release();
        b.sel_plan[i].ready = b.sel_plan[i].ok = false;
    }
    b.spread_probe.release();
    b.capture_scratch.release();
    wt.bv_dropped_view = b.view; // n_bits, ones, ... stay readable (wt_bv_bits)
    BvView v{};
    v.n_bits = b.view.n_bits;
    v.n_lines = b.view.n_lines;
    v.ones = b.view.ones;
    b.view = v;
    wt.binary_dropped = true;
    return SDSL_HIP_OK;
}

sdsl_hip_status wt_restore_binary(WtHost & wt)
{
    if (!wt.binary_dropped)
        return SDSL_HIP_OK;
    SH_HIP(hipSetDevice(wt.device));
    WtHost tmp;
    {
        DevBuf d_sym;
        SH_TRY(d_sym.alloc(std::max<uint64_t>(wt.size, 1)));
        if (wt.size)
            hipLaunchKernelGGL(k_wt_export_symbols, dim3(grid_for(wt.size, kQPB, 256u * 16u)), dim3(kBlock), 0, 0, wt.view(),
                               d_sym.as<uint8_t>(), wt.size);
        SH_HIP(hipGetLastError());
        tmp.tables = wt.tables;
        tmp.n_nodes = wt.n_nodes;
        tmp.sigma = wt.sigma;
        SH_TRY(wt_build_from_device_text(tmp, d_sym.as<uint8_t>(), wt.size, wt.device, kWtShapeGiven));
    }
    if (tmp.bv.view.n_bits != wt.bv.view.n_bits || memcmp(tmp.tables.bv_pos_rank, wt.tables.bv_pos_rank, sizeof tmp.tables.bv_pos_rank) != 0)
    {
        set_error("internal: the binary levels rebuilt from the fused lines do not match the tree's node table");
        return SDSL_HIP_ERR_HIP;
    }
    BvHost & b = wt.bv;
    std::lock_guard<std::mutex> lk(b.scratch_mutex);
    b.view = tmp.bv.view;
    b.lines = std::move(tmp.bv.lines);
    for (int i = 0; i < 2; ++i)
    {
        b.sel[i] = std::move(tmp.bv.sel[i]);
        b.lmask[i] = std::move(tmp.bv.lmask[i]);
        b.lidx[i] = std::move(tmp.bv.lidx[i]);
        b.lpos[i] = std::move(tmp.bv.lpos[i]);
    }
    wt.binary_dropped = false;
    return SDSL_HIP_OK;
}

// blocks per launch of k_wt8_counts / k_wt8_cross (they stride over what the grid does not cover; SDSL_HIP_WT8_GRID_CAP lets a test
// reach the striding with a small tree: tests/test_gpu_wt_layouts.py)
static unsigned wt8_grid_cap()
{
    const char * e = getenv("SDSL_HIP_WT8_GRID_CAP");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? (unsigned)v : (1u << 20);
}

// Builds the fused layout (lines, node tables, select directory) of the tree `src` into `dst`.  Leaves dst.d_fused
// empty when the tree does not qualify.
static sdsl_hip_status fused_from(const WtHost & src, WtHost & dst)
{
    const WtHost & wt = src;
    const WtTables & T = wt.tables;
    std::vector<WtFusedTables> ft_store(1);
    WtFusedTables & FT = ft_store[0];
    memset(&FT, 0, sizeof FT);
    const uint32_t N = wt.n_nodes;
    // breadth-first order, depths and subtree sizes
    std::vector<uint32_t> order, depth(N, 0);
    std::vector<uint64_t> size(N, 0);
    std::vector<char> seen(N, 0);
    order.reserve(N);
    order.push_back(0);
    seen[0] = 1;
    for (size_t h = 0; h < order.size(); ++h)
    {
        const uint32_t v = order[h];
        if (T.child[v][0] == kWtUndef)
            continue;
        for (int b = 0; b < 2; ++b)
        {
            const uint32_t c = T.child[v][b];
            if (c >= N || seen[c])
                return SDSL_HIP_OK; // not a tree the fused walk understands: keep the binary layout only
            seen[c] = 1;
            depth[c] = depth[v] + 1;
            order.push_back(c);
        }
    }
    for (size_t h = order.size(); h-- > 0;)
    {
        const uint32_t v = order[h];
        if (T.child[v][0] == kWtUndef)
            size[v] = wt.occ[T.bv_pos_rank[v] & 0xFF];
        else
            size[v] = size[T.child[v][0]] + size[T.child[v][1]];
    }
    if (size[0] != wt.size)
        return SDSL_HIP_OK;
    uint64_t total = 0;
    std::vector<uint32_t> roots;
    for (uint32_t v : order)
    {
        if (T.child[v][0] == kWtUndef || depth[v] % kFK != 0)
            continue;
        FT.fline[v] = (uint32_t)total;
        total += fused_lines_for(size[v]);
        roots.push_back(v);
    }
    if (total >= (UINT64_C(1) << 32))
        return SDSL_HIP_OK;
    const uint64_t n_super = kFK == 4 ? (total >> kFSuperLog) + 1 : 0; // (16-ary lines: wt_device.hpp, fused_super)
    SH_HIP(hipSetDevice(wt.device));
    const bool trace = getenv("SDSL_HIP_TRACE_BUILD") != nullptr;
    auto now = [&]() {
        if (trace)
            (void)hipDeviceSynchronize();
        return std::chrono::steady_clock::now();
    };
    const auto t0 = now();
    SH_TRY(dst.d_fused.alloc(total * kFusedWords * 8));
    SH_HIP(hipMemsetAsync(dst.d_fused.p, 0, total * kFusedWords * 8, 0));
    const bool sup_wide = (wt.size >> 32) != 0; // 64-bit records
    if (n_super)
    {
        SH_TRY(dst.d_fsuper.alloc(n_super * kFSlots * (sup_wide ? 8 : 4)));
        SH_HIP(hipMemsetAsync(dst.d_fsuper.p, 0, n_super * kFSlots * (sup_wide ? 8 : 4), 0));
    }
    const WtView view = wt.view_binary();
    uint64_t * fl = dst.d_fused.as<uint64_t>();
    uint32_t * sup_lo = dst.d_fsuper.as<uint32_t>();
    const auto t1 = now();
    for (uint32_t v : roots)
    {
        const uint64_t lines_v = fused_lines_for(size[v]);
        uint64_t * at = fl + (uint64_t)FT.fline[v] * kFusedWords;
        if (size[v])
            hipLaunchKernelGGL(k_wt8_planes, dim3(grid_for((size[v] + kFLane - 1) / kFLane, 4, 256u * 8u)), dim3(256), 0, 0, view, v,
                               size[v], at);
        if (kFK == 4)
            hipLaunchKernelGGL(k_wt8_supers, dim3(grid_for(((lines_v >> kFSuperLog) + 2) * kFSlots, 256, wt8_grid_cap())), dim3(256), 0, 0, view, v, lines_v,
                               (uint64_t)FT.fline[v], sup_lo, sup_wide);
        hipLaunchKernelGGL(k_wt8_counts, dim3(grid_for(lines_v * kFSlots, 256, wt8_grid_cap())), dim3(256), 0, 0, view, v, lines_v, at,
                           (uint64_t)FT.fline[v], sup_lo, sup_wide);
    }
    SH_HIP(hipGetLastError());
    if (kFK == 3 && (wt.size >> 32))
    { // where the counts reach multiples of 2^32 (wt_device.hpp: WtFusedTables)
        DevBuf d_n, d_pos, d_key;
        SH_TRY(d_n.alloc(4, true));
        SH_TRY(d_pos.alloc(kFusedMaxCross * 8, true));
        SH_TRY(d_key.alloc(kFusedMaxCross * 4, true));
        for (uint32_t v : roots)
            if (size[v] >> 32)
            {
                FcrossArgs ca;
                for (unsigned t = 0; t < kFSlots; ++t)
                { // the node three levels down along t, or the leaf met earlier (then the rest of t must be zero)
                    uint32_t x = v;
                    bool ok = true;
                    for (unsigned k = 0; k < kFK; ++k)
                    {
                        if (T.child[x][0] == kWtUndef)
                        {
                            ok = (t >> k) == 0;
                            break;
                        }
                        x = T.child[x][(t >> k) & 1];
                    }
                    ca.total[t] = ok ? size[x] : 0;
                }
                const uint64_t lines_v = fused_lines_for(size[v]);
                hipLaunchKernelGGL(k_wt8_cross, dim3(grid_for(lines_v * 8, 256, wt8_grid_cap())), dim3(256), 0, 0,
                                   fl + (uint64_t)FT.fline[v] * kFusedWords, lines_v, FT.fline[v], v, ca, d_n.as<uint32_t>(), d_pos.as<uint64_t>(),
                                   d_key.as<uint32_t>());
            }
        SH_HIP(hipGetLastError());
        uint32_t n_cross = 0;
        std::vector<uint64_t> hp(kFusedMaxCross);
        std::vector<uint32_t> hk(kFusedMaxCross);
        SH_HIP(hipMemcpy(&n_cross, d_n.p, 4, hipMemcpyDeviceToHost));
        SH_HIP(hipMemcpy(hp.data(), d_pos.p, kFusedMaxCross * 8, hipMemcpyDeviceToHost));
        SH_HIP(hipMemcpy(hk.data(), d_key.p, kFusedMaxCross * 4, hipMemcpyDeviceToHost));
        if (n_cross > kFusedMaxCross)
        { // (a sequence of 2^36 symbols and more: the binary levels answer)
            dst.d_fused.release();
            return SDSL_HIP_OK;
        }
        FT.n_cross = n_cross;
        for (uint32_t e = 0; e < n_cross; ++e)
        {
            FT.cross_pos[e] = hp[e];
            FT.cross_key[e] = (uint16_t)hk[e];
        }
        if (trace)
            fprintf(stderr, "[sdsl_hip] fused layout: %u places where a count reaches a multiple of 2^32\n", n_cross);
    }
    SH_TRY(dst.d_ftables.alloc(sizeof(WtFusedTables)));
    SH_HIP(hipMemcpy(dst.d_ftables.p, &FT, sizeof(WtFusedTables), hipMemcpyHostToDevice));
    // the layout by fused node (wt_device.hpp: WtFusedWalk) for the walks of inverse_select
    if (roots.size() <= (size_t)kFselMaxRoots && (kFK == 4 || !(wt.size >> 32)))
    {
        std::vector<WtFusedWalk> fw_store(1);
        WtFusedWalk & FW = fw_store[0];
        memset(&FW, 0xFF, sizeof FW);
        FW.n_roots = (uint32_t)roots.size();
        std::vector<uint32_t> rid(N, 0);
        for (size_t r = 0; r < roots.size(); ++r)
            rid[roots[r]] = (uint32_t)r;
        for (size_t r = 0; r < roots.size(); ++r)
        {
            FW.rline[r] = FT.fline[roots[r]];
            for (unsigned t = 0; t < kFSlots; ++t)
            { // the node kFK levels down along t, or the leaf met earlier (then the rest of t must be zero)
                uint32_t x = roots[r];
                bool ok = true;
                for (unsigned k = 0; k < kFK; ++k)
                {
                    if (T.child[x][0] == kWtUndef)
                    {
                        ok = (t >> k) == 0;
                        break;
                    }
                    x = T.child[x][(t >> k) & 1];
                }
                if (!ok)
                    continue;
                FW.succ[r][t] = T.child[x][0] == kWtUndef ? (uint16_t)(kFWalkLeaf | (T.bv_pos_rank[x] & 0xFF)) : (uint16_t)rid[x];
            }
        }
        SH_TRY(dst.d_fwalk.alloc(sizeof(WtFusedWalk)));
        SH_HIP(hipMemcpy(dst.d_fwalk.p, &FW, sizeof(WtFusedWalk), hipMemcpyHostToDevice));
    }
    // the layout by symbol (wt_device.hpp: WtStepTab) for rank(i, c)
    if (kFK == 4 || !(wt.size >> 32))
    {
        std::vector<WtStepTab> st_store(1);
        WtStepTab & ST = st_store[0];
        memset(&ST, 0, sizeof ST);
        uint32_t used = 0;
        bool fits = true;
        for (unsigned c = 0; c < 256 && fits; ++c)
        {
            if (T.c_to_leaf[c] == kWtUndef)
                continue;
            uint64_t p = T.path[c];
            unsigned left = (unsigned)(p >> 56), v = 0, steps = 0;
            const uint32_t first = used;
            while (left && fits)
            {
                const unsigned k = left < kFK ? left : kFK, t = (unsigned)p & ((1u << k) - 1u);
                if (used >= kWtMaxSteps || FT.fline[v] >= (1u << kLimStepTableLineBits))
                {
                    fits = false;
                    break;
                }
                ST.steps[used++] = FT.fline[v] | (t << 28);
                for (unsigned j = 0, tt = t; j < kFK; ++j, tt >>= 1)
                { // wt_descend
                    const unsigned nv = T.child[v][tt & 1];
                    v = nv == kWtUndef ? v : nv;
                }
                p >>= k;
                left -= k;
                ++steps;
            }
            if (steps == 0 || steps > 255)
                fits = false;
            ST.meta[c] = first | (steps << 16);
        }
        if (fits)
        {
            SH_TRY(dst.d_fsteps.alloc(sizeof(WtStepTab)));
            SH_HIP(hipMemcpy(dst.d_fsteps.p, &ST, sizeof(WtStepTab), hipMemcpyHostToDevice));
        }
    }
    // select directory (skipped for trees with more fused nodes than its table holds: select then walks the binary levels)
    const char * env_sel = getenv("SDSL_HIP_WT_FUSED_SELECT"); // 0: select keeps walking the binary levels
    // (8-ary lines of 2^32 symbols and more: the directory holds 32-bit positions there; 16-ary lines get 64-bit entries)
    if (roots.size() <= (size_t)kFselMaxRoots && !(env_sel && atoi(env_sel) == 0) && (kFK == 4 || !(wt.size >> 32)))
    {
        std::vector<WtFusedSelTables> fs_store(1);
        WtFusedSelTables & FS = fs_store[0];
        memset(&FS, 0xFF, sizeof FS);
        std::vector<FselNodeArgs> args(roots.size());
        uint64_t n_dir = 0;
        for (size_t r = 0; r < roots.size(); ++r)
        {
            const uint32_t u = roots[r];
            FS.root_id[u] = (uint16_t)r;
            args[r].size = size[u];
            for (unsigned t = 0; t < kFSlots; ++t)
            { // the node kFK levels down along t, or the leaf met earlier (then the rest of t must be zero)
                uint32_t x = u;
                bool ok = true;
                for (unsigned k = 0; k < kFK; ++k)
                {
                    if (T.child[x][0] == kWtUndef)
                    {
                        ok = (t >> k) == 0;
                        break;
                    }
                    x = T.child[x][(t >> k) & 1];
                }
                args[r].cnt[t] = args[r].n_samples[t] = 0;
                args[r].off[t] = kFselNone;
                if (!ok || size[x] == 0)
                    continue;
                args[r].cnt[t] = size[x];
                FS.cnt[r][t] = (uint32_t)size[x];
                FS.cnt_hi[r][t] = (uint8_t)(size[x] >> 32);
                args[r].n_samples[t] = (uint32_t)((size[x] + (1u << kFselLog) - 1) >> kFselLog) + 1;
                args[r].off[t] = FS.off[r][t] = (uint32_t)n_dir;
                n_dir += args[r].n_samples[t];
            }
        }
        if (n_dir < (UINT64_C(1) << 32))
        {
            SH_TRY(dst.d_fsel.alloc((n_dir + 1) * (sup_wide ? 8 : 4)));
            for (size_t r = 0; r < roots.size(); ++r)
            {
                const uint32_t u = roots[r];
                const uint64_t lines_u = fused_lines_for(size[u]);
                const dim3 grid(grid_for(lines_u * kFSlots, 256, wt8_grid_cap()));
                if (sup_wide)
                    hipLaunchKernelGGL(k_wt8_sel_dir<true>, grid, dim3(256), 0, 0, fl + (uint64_t)FT.fline[u] * kFusedWords, sup_lo,
                                       (uint64_t)FT.fline[u], lines_u, args[r], dst.d_fsel.p);
                else
                    hipLaunchKernelGGL(k_wt8_sel_dir<false>, grid, dim3(256), 0, 0, fl + (uint64_t)FT.fline[u] * kFusedWords, sup_lo,
                                       (uint64_t)FT.fline[u], lines_u, args[r], dst.d_fsel.p);
            }
            SH_HIP(hipGetLastError());
            SH_TRY(dst.d_fsel_tables.alloc(sizeof(WtFusedSelTables)));
            SH_HIP(hipMemcpy(dst.d_fsel_tables.p, &FS, sizeof(WtFusedSelTables), hipMemcpyHostToDevice));
        }
    }
    SH_HIP(hipStreamSynchronize(0));
    if (trace)
    {
        const auto t2 = now();
        fprintf(stderr, "[sdsl_hip] fused layout: %zu nodes, %llu lines, alloc+clear %.1f ms, kernels %.1f ms\n", roots.size(),
                (unsigned long long)total, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    return SDSL_HIP_OK;
}

sdsl_hip_status wt_build_fused(WtHost & wt)
{
    const char * env = getenv("SDSL_HIP_WT_FUSED");
    if (env && atoi(env) == 0)
        return SDSL_HIP_OK;
    if (wt.backend != 0 || wt.sigma < 2 || wt.n_nodes < 3 || wt.size == 0 || wt.size >= kLimWtFusedSymbols ||
        !wt.d_tables.p)
        return SDSL_HIP_OK; // (2^32 .. 2^36 symbols: rank / access / LF on the fused lines, select on the binary levels)
    // The fused layout serves rank / access / select, whose answers do not depend on the tree's shape, so it gets the
    // shape that suits it: an 8-ary Huffman tree (fewest fused steps per symbol) instead of SDSL's binary tree cut
    // into groups of three levels.  The symbols are read back from the binary levels, a throw-away binary tree of the
    // new shape is built from them, and the fused layout is derived from that.  SDSL_HIP_WT_FUSED_SHAPE=binary keeps
    // the derived-from-SDSL's-tree form (also the fallback).
    const char * shape_env = getenv("SDSL_HIP_WT_FUSED_SHAPE");
    { // the walk below follows child links until it meets a leaf: only on a proper tree (a loaded stream may be anything)
        const WtTables & T = wt.tables;
        std::vector<char> seen(wt.n_nodes, 0);
        std::vector<uint32_t> stack(1, 0);
        seen[0] = 1;
        size_t visited = 0;
        while (!stack.empty())
        {
            const uint32_t v = stack.back();
            stack.pop_back();
            ++visited;
            if (T.child[v][0] == kWtUndef)
                continue;
            for (int b = 0; b < 2; ++b)
            {
                const uint32_t c = T.child[v][b];
                if (c >= wt.n_nodes || seen[c])
                    return SDSL_HIP_OK; // keep the binary layout only
                seen[c] = 1;
                stack.push_back(c);
            }
        }
        if (visited != wt.n_nodes)
            return SDSL_HIP_OK;
    }
    if (!(shape_env && shape_env[0] == 'b'))
    {
        SH_HIP(hipSetDevice(wt.device));
        DevBuf d_sym;
        WtHost own;
        sdsl_hip_status st = d_sym.alloc(wt.size);
        const bool tr = getenv("SDSL_HIP_TRACE_BUILD") != nullptr;
        auto mark = [&](const char * what) {
            if (tr)
            {
                const hipError_t e = hipDeviceSynchronize();
                fprintf(stderr, "[sdsl_hip] fused build: %s (%s)\n", what, hipGetErrorString(e));
            }
        };
        if (st == SDSL_HIP_OK)
        {
            hipLaunchKernelGGL(k_wt_export_symbols, dim3(grid_for(wt.size, kQPB, 256u * 16u)), dim3(kBlock), 0, 0,
                               wt.view_binary(), d_sym.as<uint8_t>(), wt.size);
            mark("symbols exported");
            st = wt_build_from_device_text(own, d_sym.as<uint8_t>(), wt.size, wt.device, kWtShapeHuff8 | kWtNoSelect);
            mark("own-shape tree built");
        }
        if (st == SDSL_HIP_OK)
            st = fused_from(own, wt);
        if (tr && st != SDSL_HIP_OK)
            fprintf(stderr, "[sdsl_hip] fused build in its own shape failed (%s): deriving it from the binary tree\n", last_error_message());
        mark("fused layout derived");
        if (st == SDSL_HIP_OK && wt.d_fused.p)
        {
            wt.d_tables_f = std::move(own.d_tables);
            wt.tables_f = own.tables;
            return SDSL_HIP_OK;
        }
        wt.d_fused.release();
        wt.d_ftables.release();
        wt.d_fsuper.release();
        wt.d_fwalk.release();
        wt.d_fsteps.release();
        wt.d_fsel.release();
        wt.d_fsel_tables.release();
        (void)hipGetLastError(); // (a failed allocation stays the runtime's "last error" until it is read: the launches below check it)
    }
    // the fused layout is an accelerator: if it cannot be built (memory), the handle works on its binary levels
    if (fused_from(wt, wt) != SDSL_HIP_OK)
    {
        if (getenv("SDSL_HIP_TRACE_BUILD"))
            fprintf(stderr, "[sdsl_hip] fused layout not built: %s\n", last_error_message());
        wt.d_fused.release();
        wt.d_ftables.release();
        wt.d_fsuper.release();
        wt.d_fwalk.release();
        wt.d_fsteps.release();
        wt.d_fsel.release();
        wt.d_fsel_tables.release();
        (void)hipGetLastError();
    }
    return SDSL_HIP_OK;
}

static unsigned wt_grid(uint64_t n)
{
    return grid_for(n, kQPB, 256u * 8u);
}

sdsl_hip_status wt_launch_rank(const WtHost & wt, const uint64_t * d_i, const uint8_t * d_c, uint64_t n,
                               uint64_t * d_out, hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    KernelTimer t(s);
    static const bool flat = !(getenv("SDSL_HIP_WT_RANK_FLAT") && atoi(getenv("SDSL_HIP_WT_RANK_FLAT")) == 0);
    if (wt.d_fused.p && wt.d_fsteps.p && wt.sigma >= 2 && flat)
        hipLaunchKernelGGL(k_wt_rank_flat, dim3(wt_grid(n)), dim3(kBlock), 0, s, wt.view(), d_i, d_c, d_out, n);
    else
        hipLaunchKernelGGL((k_wt_rank<false>), dim3(wt_grid(n)), dim3(kBlock), 0, s, wt.view(), d_i, d_c, d_out, n);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

} // namespace sdslhip

using namespace sdslhip;

struct sdsl_hip_wt_s
{
    WtHost h;
    DevBuf d_occ;
    uint64_t uid = next_handle_uid();
};

// shared with fm.hip
sdsl_hip_wt_s * sdsl_hip_wt_alloc()
{
    return new (std::nothrow) sdsl_hip_wt_s();
}
WtHost & sdsl_hip_wt_host(sdsl_hip_wt_s * w)
{
    return w->h;
}
const uint64_t * sdsl_hip_wt_device_occ(sdsl_hip_wt_s * w)
{
    return w->d_occ.as<uint64_t>();
}
sdsl_hip_status sdsl_hip_wt_finish(sdsl_hip_wt_s * w)
{
    SH_TRY(wt_build_fused(w->h));
    SH_TRY(w->d_occ.alloc(sizeof w->h.occ));
    SH_HIP(hipMemcpy(w->d_occ.p, w->h.occ, sizeof w->h.occ, hipMemcpyHostToDevice));
    return SDSL_HIP_OK;
}

extern "C" {

static sdsl_hip_status sdsl_hip_wt_create_ex_impl(const uint8_t * text, uint64_t n, int32_t device, uint32_t flags, sdsl_hip_wt_t * out)
{
    if (!out || (!text && n))
    {
        set_error("wt_create: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_wt_s * w = sdsl_hip_wt_alloc();
    if (!w)
        return SDSL_HIP_ERR_NOMEM;
    Staged t;
    sdsl_hip_status st = t.in(text, n, nullptr); // host bytes are uploaded; device bytes are used where they are
    if (st == SDSL_HIP_OK)
        st = wt_build_from_device_text(w->h, (const uint8_t *)t.dev, n, device, flags);
    if (st == SDSL_HIP_OK)
        st = sdsl_hip_wt_finish(w);
    if (st != SDSL_HIP_OK)
    {
        delete w;
        return st;
    }
    *out = w;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_wt_create_ex(const uint8_t * text, uint64_t n, int32_t device, uint32_t flags, sdsl_hip_wt_t * out)
{
    return guarded("wt_create_ex", [&] { return sdsl_hip_wt_create_ex_impl(text, n, device, flags, out); });
}

static sdsl_hip_status sdsl_hip_wt_create_impl(const uint8_t * text, uint64_t n, int32_t device, sdsl_hip_wt_t * out)
{
    return sdsl_hip_wt_create_ex(text, n, device, 0, out);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_wt_create(const uint8_t * text, uint64_t n, int32_t device, sdsl_hip_wt_t * out)
{
    return guarded("wt_create", [&] { return sdsl_hip_wt_create_impl(text, n, device, out); });
}

static sdsl_hip_status sdsl_hip_wt_create_from_sdsl_impl(const void * bytes, size_t len, int32_t layout, int32_t device,
                                             sdsl_hip_wt_t * out, size_t * consumed)
{
    if (!out || !bytes)
    {
        set_error("wt_create_from_sdsl: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_wt_s * w = sdsl_hip_wt_alloc();
    if (!w)
        return SDSL_HIP_ERR_NOMEM;
    StreamReader rd(bytes, len);
    sdsl_hip_status st = wt_build_from_stream(w->h, rd, layout, device);
    if (st == SDSL_HIP_OK)
        st = sdsl_hip_wt_finish(w);
    if (st != SDSL_HIP_OK)
    {
        delete w;
        return st;
    }
    if (consumed)
        *consumed = rd.pos;
    *out = w;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_wt_create_from_sdsl(const void * bytes, size_t len, int32_t layout, int32_t device,
                                             sdsl_hip_wt_t * out, size_t * consumed)
{
    return guarded("wt_create_from_sdsl", [&] { return sdsl_hip_wt_create_from_sdsl_impl(bytes, len, layout, device, out, consumed); });
}

static sdsl_hip_status sdsl_hip_wt_serialize_impl(sdsl_hip_wt_t wt, void * buf, size_t cap, size_t * written)
{
    return sdsl_hip_wt_serialize_ex(wt, SDSL_HIP_LAYOUT_BV_SCAN, buf, cap, written);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_wt_serialize(sdsl_hip_wt_t wt, void * buf, size_t cap, size_t * written)
{
    return guarded("wt_serialize", [&] { return sdsl_hip_wt_serialize_impl(wt, buf, cap, written); });
}

} // extern "C"

// appends wt_pc::serialize of the tree in the requested flavour to w (shared with fm.hip)
sdsl_hip_status sdsl_hip_wt_serialize_into(sdsl_hip_wt_s * wt, int32_t layout, StreamWriter & w)
{
    if (!wt || (wt->h.backend == 0 && layout != SDSL_HIP_LAYOUT_BV_SCAN && layout != SDSL_HIP_LAYOUT_BV_MCL
                && layout != SDSL_HIP_LAYOUT_BV_DEFAULT))
    {
        set_error("wt_serialize: null handle or unknown layout");
        return SDSL_HIP_ERR_INVALID;
    }
    const WtHost & h = wt->h;
    SH_HIP(hipSetDevice(h.device));
    if (h.backend == 1)
    { // wt_huff<rrr_vector<63>>: size, sigma, the rrr vector, (supports: nothing), tree
        w.u64(h.size);
        w.u64(h.sigma);
        SH_TRY(rrr_serialize_host(h.rrr, w));
        w.u64(h.n_nodes);
        for (uint32_t v = 0; v < h.n_nodes; ++v)
        {
            w.u64(h.tables.bv_pos[v]);
            w.u64(h.tables.bv_pos_rank[v]);
            w.u16(h.tables.parent[v]);
            w.u16(h.tables.child[v][0]);
            w.u16(h.tables.child[v][1]);
        }
        w.raw(h.tables.c_to_leaf, sizeof h.tables.c_to_leaf);
        w.raw(h.tables.path, sizeof h.tables.path);
        return SDSL_HIP_OK;
    }
    const uint64_t nb = h.bv.view.n_bits, W = (nb + 63) >> 6;
    // the bit vector back in SDSL's word layout
    std::vector<uint64_t> words(W + 1, 0);
    if (W && h.binary_dropped)
    { // a tree whose binary levels were released (wt_drop_binary): the symbols are read out of the fused lines and the level builder
      // writes the bits of the tree's own node table — into a buffer of this call; the handle is not touched (readers may run beside it)
        DevBuf d_sym, d;
        SH_TRY(d_sym.alloc(h.size));
        hipLaunchKernelGGL(k_wt_export_symbols, dim3(grid_for(h.size, kQPB, 256u * 16u)), dim3(kBlock), 0, 0, h.view(), d_sym.as<uint8_t>(),
                           h.size);
        SH_HIP(hipGetLastError());
        WtHost tmp;
        tmp.tables = h.tables;
        tmp.n_nodes = h.n_nodes;
        tmp.sigma = h.sigma;
        SH_TRY(wt_build_from_device_text(tmp, d_sym.as<uint8_t>(), h.size, h.device, kWtShapeGiven, &d));
        if (tmp.bv.view.n_bits != nb)
        {
            set_error("internal: the levels rebuilt from the fused lines hold %llu bits, the tree %llu", (unsigned long long)tmp.bv.view.n_bits,
                      (unsigned long long)nb);
            return SDSL_HIP_ERR_HIP;
        }
        SH_HIP(hipMemcpy(words.data(), d.p, W * 8, hipMemcpyDeviceToHost));
    }
    else if (W)
    {
        DevBuf d;
        SH_TRY(d.alloc(W * 8));
        SH_TRY(bv_export_words_device(h.bv.view, d.as<uint64_t>(), W, nullptr));
        SH_HIP(hipMemcpy(words.data(), d.p, W * 8, hipMemcpyDeviceToHost));
    }
    w.u64(h.size);
    w.u64(h.sigma);
    w.int_vector(words.data(), nb, 1);
    if (h.size == 0)
        w.int_vector(nullptr, 0, 64); // default-constructed support: empty int_vector<64>
    else if (layout == SDSL_HIP_LAYOUT_BV_DEFAULT)
        rank_v_serialize_host(words.data(), nb, 1, w);
    else
        rank_v5_serialize_host(words.data(), nb, 1, w);
    if (layout != SDSL_HIP_LAYOUT_BV_SCAN)
    { // select_support_mcl<1>, select_support_mcl<0> of the tree's bit vector
        if (h.size == 0)
        {
            w.u64(0);
            w.u64(0);
        }
        else
        {
            select_mcl_serialize_host(words.data(), nb, 1, w);
            select_mcl_serialize_host(words.data(), nb, 0, w);
        }
    }
    // (select_support_scan<1>, select_support_scan<0>: nothing)
    w.u64(h.n_nodes);
    for (uint32_t v = 0; v < h.n_nodes; ++v)
    { // _node::serialize (wt_helper.hpp:139-150)
        w.u64(h.tables.bv_pos[v]);
        w.u64(h.tables.bv_pos_rank[v]);
        w.u16(h.tables.parent[v]);
        w.u16(h.tables.child[v][0]);
        w.u16(h.tables.child[v][1]);
    }
    w.raw(h.tables.c_to_leaf, sizeof h.tables.c_to_leaf);
    w.raw(h.tables.path, sizeof h.tables.path);
    return SDSL_HIP_OK;
}

extern "C" {

static sdsl_hip_status sdsl_hip_wt_serialize_ex_impl(sdsl_hip_wt_t wt, int32_t layout, void * buf, size_t cap, size_t * written)
{
    if (!wt)
    {
        set_error("wt_serialize: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    sdsl_hip_status st;
    if (deliver_cached(wt->uid, (uint64_t)(uint32_t)layout, buf, cap, written, st))
        return st;
    StreamWriter w;
    SH_TRY(sdsl_hip_wt_serialize_into(wt, layout, w));
    return deliver_and_cache(wt->uid, (uint64_t)(uint32_t)layout, w, buf, cap, written);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_wt_serialize_ex(sdsl_hip_wt_t wt, int32_t layout, void * buf, size_t cap, size_t * written)
{
    return guarded("wt_serialize_ex", [&] { return sdsl_hip_wt_serialize_ex_impl(wt, layout, buf, cap, written); });
}

sdsl_hip_status sdsl_hip_wt_destroy(sdsl_hip_wt_t wt)
{
    if (!wt)
        return SDSL_HIP_OK;
    (void)hipSetDevice(wt->h.device);
    delete wt;
    return SDSL_HIP_OK;
}

uint64_t sdsl_hip_wt_size(sdsl_hip_wt_t wt)
{
    return wt ? wt->h.size : 0;
}
uint64_t sdsl_hip_wt_sigma(sdsl_hip_wt_t wt)
{
    return wt ? wt->h.sigma : 0;
}
uint64_t sdsl_hip_wt_bv_size(sdsl_hip_wt_t wt)
{
    return wt ? wt_bv_bits(wt->h) : 0;
}
uint64_t sdsl_hip_wt_device_bytes(sdsl_hip_wt_t wt)
{
    return wt ? wt->h.device_bytes() : 0;
}

sdsl_hip_status sdsl_hip_wt_release_binary_levels(sdsl_hip_wt_t wt)
{
    if (!wt)
    {
        set_error("wt_release_binary_levels: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    SH_HIP(hipSetDevice(wt->h.device));
    SH_HIP(hipDeviceSynchronize()); // nothing in flight may still read the levels
    return wt_drop_binary(wt->h);
}

void sdsl_hip_wt_fused_geometry(uint32_t * levels, uint32_t * positions_per_line, uint32_t * lines_per_superblock)
{
    if (levels)
        *levels = kFK;
    if (positions_per_line)
        *positions_per_line = kFusedPos;
    if (lines_per_superblock)
        *lines_per_superblock = kFK == 4 ? 1u << kFSuperLog : 0u;
}

sdsl_hip_status sdsl_hip_wt_fused_steps(sdsl_hip_wt_t wt, uint8_t steps_out[256])
{
    if (!wt || !steps_out)
        return SDSL_HIP_ERR_INVALID;
    const WtTables & T = wt->h.d_tables_f.p ? wt->h.tables_f : wt->h.tables;
    for (int c = 0; c < 256; ++c)
        steps_out[c] = !wt->h.d_fused.p || T.c_to_leaf[c] == kWtUndef ? 0 : (uint8_t)(((T.path[c] >> 56) + kFK - 1) / kFK);
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_wt_code_lengths(sdsl_hip_wt_t wt, uint8_t len_out[256])
{
    if (!wt || !len_out)
        return SDSL_HIP_ERR_INVALID;
    for (int c = 0; c < 256; ++c)
        len_out[c] = wt->h.tables.c_to_leaf[c] == kWtUndef ? 0 : (uint8_t)(wt->h.tables.path[c] >> 56);
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_wt_rank_batch(sdsl_hip_wt_t wt, const uint64_t * i, const uint8_t * c, uint64_t n,
                                       uint64_t * out, void * stream)
{
    if (!wt || (n && (!i || !c || !out)))
    {
        set_error("wt_rank_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(wt->h.device));
    if (n >= kPipelineMinQueries && !is_device_ptr(i) && !is_device_ptr(c) && !is_device_ptr(out))
    { // all three arrays in host memory: chunked over several streams (common.hpp host_pipeline_bytes2)
        const WtHost * h = &wt->h;
        return host_pipeline_bytes2(h->device, (const uint8_t *)i, 8, c, 1, (uint8_t *)out, 8, n, pipeline_chunk(),
                                    [h](const void * di, const void * dc, void * dout, uint64_t cnt,
                                        hipStream_t st) -> sdsl_hip_status
                                    {
                                        return h->backend == 1
                                                   ? wt_rrr_launch_rank(*h, (const uint64_t *)di, (const uint8_t *)dc, cnt,
                                                                        (uint64_t *)dout, st)
                                                   : wt_launch_rank(*h, (const uint64_t *)di, (const uint8_t *)dc, cnt,
                                                                    (uint64_t *)dout, st);
                                    });
    }
    Staged si, sc, so;
    SH_TRY(si.in(i, n * 8, s));
    SH_TRY(sc.in(c, n, s));
    SH_TRY(so.out(out, n * 8));
    if (wt->h.backend == 1)
        SH_TRY(wt_rrr_launch_rank(wt->h, (const uint64_t *)si.dev, (const uint8_t *)sc.dev, n, (uint64_t *)so.dev, s));
    else
        SH_TRY(wt_launch_rank(wt->h, (const uint64_t *)si.dev, (const uint8_t *)sc.dev, n, (uint64_t *)so.dev, s));
    SH_TRY(so.finish(s));
    if ((si.host || sc.host) && !so.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_wt_inverse_select_batch(sdsl_hip_wt_t wt, const uint64_t * i, uint64_t n,
                                                 uint64_t * out_rank, uint8_t * out_c, void * stream)
{
    if (!wt || (n && (!i || !out_c)))
    {
        set_error("wt_inverse_select_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(wt->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    Staged si, sr, sc;
    SH_TRY(si.in(i, n * 8, s));
    SH_TRY(sc.out(out_c, n));
    if (out_rank)
        SH_TRY(sr.out(out_rank, n * 8));
    if (wt->h.backend == 1)
        SH_TRY(wt_rrr_launch_inverse_select(wt->h, (const uint64_t *)si.dev, n, out_rank ? (uint64_t *)sr.dev : nullptr,
                                            (uint8_t *)sc.dev, s));
    else
    {
        KernelTimer t(s);
        unsigned grid = grid_for(n, kQPB, 256u * 8u);
        static const bool flat = !(getenv("SDSL_HIP_WT_RANK_FLAT") && atoi(getenv("SDSL_HIP_WT_RANK_FLAT")) == 0);
        if (wt->h.d_fused.p && wt->h.d_fwalk.p && wt->h.sigma >= 2 && flat)
        {
            if (out_rank)
                hipLaunchKernelGGL(k_wt_inverse_select_flat<true>, dim3(grid), dim3(kBlock), 0, s, wt->h.view(), (const uint64_t *)si.dev,
                                   (uint64_t *)sr.dev, (uint8_t *)sc.dev, n);
            else
                hipLaunchKernelGGL(k_wt_inverse_select_flat<false>, dim3(grid), dim3(kBlock), 0, s, wt->h.view(), (const uint64_t *)si.dev,
                                   (uint64_t *)nullptr, (uint8_t *)sc.dev, n);
        }
        else if (out_rank)
            hipLaunchKernelGGL((k_wt_inverse_select<false, true>), dim3(grid), dim3(kBlock), 0, s, wt->h.view(),
                               (const uint64_t *)si.dev, (uint64_t *)sr.dev, (uint8_t *)sc.dev, n);
        else
            hipLaunchKernelGGL((k_wt_inverse_select<false, false>), dim3(grid), dim3(kBlock), 0, s, wt->h.view(),
                               (const uint64_t *)si.dev, (uint64_t *)nullptr, (uint8_t *)sc.dev, n);
    }
    SH_HIP(hipGetLastError());
    SH_TRY(sc.finish(s));
    if (out_rank)
        SH_TRY(sr.finish(s));
    if (si.host && !sc.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_wt_access_batch(sdsl_hip_wt_t wt, const uint64_t * i, uint64_t n, uint8_t * out_c,
                                         void * stream)
{
    return sdsl_hip_wt_inverse_select_batch(wt, i, n, nullptr, out_c, stream);
}

sdsl_hip_status sdsl_hip_wt_select_batch(sdsl_hip_wt_t wt, const uint64_t * i, const uint8_t * c, uint64_t n,
                                         uint64_t * out, void * stream)
{
    if (!wt || (n && (!i || !c || !out)))
    {
        set_error("wt_select_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(wt->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    Staged si, sc, so;
    SH_TRY(si.in(i, n * 8, s));
    SH_TRY(sc.in(c, n, s));
    SH_TRY(so.out(out, n * 8));
    if (wt->h.backend == 1)
        SH_TRY(wt_rrr_launch_select(wt->h, wt->d_occ.as<uint64_t>(), (const uint64_t *)si.dev, (const uint8_t *)sc.dev, n,
                                    (uint64_t *)so.dev, s));
    else
    {
        if (wt->h.binary_dropped && !(wt->h.d_fused.p && wt->h.d_fsel.p))
            SH_TRY(wt_restore_binary(wt->h)); // (a tree without the fused select directory selects on its binary levels)
        const WtView view = wt->h.view();
        bool done = false;
        if (view.f_lines && view.f_sel && wt_select_sorted_applicable(wt->h, n))
        { // a large batch: ordered by place in symbol order, answered bucket by bucket (wt_sorted.hip); working memory from the
          // device's pool — none during a stream capture (ScratchLease), then the direct kernel below answers
            static DevBuf no_capture_scratch;
            ScratchLease L;
            SH_TRY(L.acquire(wt->h.device, no_capture_scratch, wt_select_sorted_scratch_bytes(n), s));
            if (L.p)
            {
                KernelTimer t(s);
                SH_TRY(wt_launch_select_sorted(wt->h, wt->d_occ.as<uint64_t>(), (const uint64_t *)si.dev, (const uint8_t *)sc.dev, n,
                                               (uint64_t *)so.dev, s, L.p, L.bytes));
                done = true;
            }
        }
        if (!done)
        {
            KernelTimer t(s);
            if (view.f_lines && view.f_sel && (view.size >> 32))
                hipLaunchKernelGGL((k_wt_select_fused<false, true>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(kBlock), 0, s, view,
                                   wt->d_occ.as<uint64_t>(), (const uint64_t *)si.dev, (const uint8_t *)sc.dev, (uint64_t *)so.dev, n);
            else if (view.f_lines && view.f_sel)
                hipLaunchKernelGGL((k_wt_select_fused<false, false>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(kBlock), 0, s, view,
                                   wt->d_occ.as<uint64_t>(), (const uint64_t *)si.dev, (const uint8_t *)sc.dev,
                                   (uint64_t *)so.dev, n);
            else // SDSL's tree and its binary levels
                hipLaunchKernelGGL((k_wt_select<false>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(kBlock), 0, s,
                                   wt->h.view_binary(), wt->d_occ.as<uint64_t>(), (const uint64_t *)si.dev,
                                   (const uint8_t *)sc.dev, (uint64_t *)so.dev, n);
        }
    }
    SH_HIP(hipGetLastError());
    SH_TRY(so.finish(s));
    if ((si.host || sc.host) && !so.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}
}
