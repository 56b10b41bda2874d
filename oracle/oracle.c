/* oracle.c — CPU restatement of SDSL's rank/select/rrr/wt_huff/csa_wt-count algorithms.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain C11, single-threaded, no dependency on the
 * product library.  Each routine follows the reference routine cited above it, keeping the
 * reference's data layout and control flow so that it is an independent check of the device
 * implementation (which uses a different layout).
 *
 * Parity status: pinned.  tests/test_oracle_*.py compare (a) the serialised bytes produced here
 * with files written by the real SDSL (tests/golden/*.sdsl and, when oracle/_ref is built, fresh
 * ones), and (b) query answers with the real library's on seeded inputs.
 */
#include "oracle.h"

#include <assert.h>
#include <stdio.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================== */
/* bits.hpp                                                                                 */
/* ======================================================================================== */

uint32_t orc_cnt(uint64_t x) /* bits.hpp:486-502 */
{
    return (uint32_t)__builtin_popcountll(x);
}

uint32_t orc_sel(uint64_t x, uint32_t i) /* bits.hpp:586-612: position of the i-th set bit, i>=1 */
{
    /* clear the i-1 lowest set bits, then count trailing zeros */
    for (uint32_t k = 1; k < i; ++k)
        x &= x - 1;
    return (uint32_t)__builtin_ctzll(x);
}

uint32_t orc_hi(uint64_t x) /* bits.hpp:653-684: hi(0) = 0 */
{
    return x ? 63u - (uint32_t)__builtin_clzll(x) : 0u;
}

static inline uint64_t lo_set(unsigned k) /* bits.hpp:194 lo_set[k] */
{
    return k >= 64 ? ~UINT64_C(0) : ((UINT64_C(1) << k) - 1);
}
static inline uint64_t lo_unset(unsigned k) /* bits.hpp:216 lo_unset[k] */
{
    return ~lo_set(k);
}

uint64_t orc_read_int(const uint64_t * d, uint64_t pos, uint8_t len) /* bits.hpp:777-790 */
{
    if (len == 0)
        return 0;
    const uint64_t * w = d + (pos >> 6);
    unsigned off = (unsigned)(pos & 63);
    uint64_t v = w[0] >> off;
    if (off + len > 64)
        v |= w[1] << (64 - off);
    return v & lo_set(len);
}

static void write_int(uint64_t * d, uint64_t pos, uint64_t x, uint8_t len) /* bits.hpp:724-746 */
{
    if (len == 0)
        return;
    x &= lo_set(len);
    uint64_t * w = d + (pos >> 6);
    unsigned off = (unsigned)(pos & 63);
    w[0] = (w[0] & ~(lo_set(len) << off)) | (x << off);
    if (off + len > 64)
    {
        unsigned done = 64 - off;
        w[1] = (w[1] & ~lo_set(len - done)) | (x >> done);
    }
}

/* std::mt19937_64 (ISO C++ [rand.predef]): w=64 n=312 m=156 r=31 a=0xB5026F5AA96619E9 u=29
 * d=0x5555555555555555 s=17 b=0x71D67FFFEDA60000 t=37 c=0xFFF7EEE000000000 l=43 f=6364136223846793005 */
typedef struct
{
    uint64_t mt[312];
    int idx;
} mt64;
static void mt64_seed(mt64 * g, uint64_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 312; ++i)
        g->mt[i] = UINT64_C(6364136223846793005) * (g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) + (uint64_t)i;
    g->idx = 312;
}
static uint64_t mt64_next(mt64 * g)
{
    if (g->idx >= 312)
    {
        for (int i = 0; i < 312; ++i)
        {
            uint64_t x = (g->mt[i] & UINT64_C(0xFFFFFFFF80000000)) | (g->mt[(i + 1) % 312] & UINT64_C(0x7FFFFFFF));
            uint64_t xa = x >> 1;
            if (x & 1)
                xa ^= UINT64_C(0xB5026F5AA96619E9);
            g->mt[i] = g->mt[(i + 156) % 312] ^ xa;
        }
        g->idx = 0;
    }
    uint64_t y = g->mt[g->idx++];
    y ^= (y >> 29) & UINT64_C(0x5555555555555555);
    y ^= (y << 17) & UINT64_C(0x71D67FFFEDA60000);
    y ^= (y << 37) & UINT64_C(0xFFF7EEE000000000);
    y ^= y >> 43;
    return y;
}

void orc_mt19937_64_fill(uint64_t * out, uint64_t n, uint64_t seed)
{
    mt64 g;
    mt64_seed(&g, seed);
    for (uint64_t i = 0; i < n; ++i)
        out[i] = mt64_next(&g);
}

void orc_set_random_bits(uint64_t * words, uint64_t n_bits, uint64_t seed) /* util.hpp:467-485 */
{
    orc_mt19937_64_fill(words, (n_bits + 63) >> 6, seed);
}

/* ======================================================================================== */
/* byte sink + int_vector serialisation                                                     */
/* ======================================================================================== */

void orc_buf_free(orc_buf * b)
{
    free(b->p);
    b->p = NULL;
    b->len = b->cap = 0;
}
static void buf_put(orc_buf * b, const void * src, size_t n)
{
    if (b->len + n > b->cap)
    {
        size_t nc = b->cap ? b->cap * 2 : 4096;
        while (nc < b->len + n)
            nc *= 2;
        b->p = (uint8_t *)realloc(b->p, nc);
        b->cap = nc;
    }
    memcpy(b->p + b->len, src, n);
    b->len += n;
}
static void buf_u64(orc_buf * b, uint64_t v)
{
    buf_put(b, &v, 8);
}

/* packed vector of `width`-bit integers in a u64 array = sdsl::int_vector<0> / bit_vector
 * (int_vector.hpp:211,254-257); one spare word is kept like memory_management.hpp:892-918 */
typedef struct
{
    uint64_t * d;
    uint64_t size; /* elements */
    uint8_t width;
} iv;

static iv iv_make(uint64_t size, uint8_t width)
{
    iv v;
    v.size = size;
    v.width = width;
    uint64_t words = ((size * width + 63) >> 6) + 1;
    v.d = (uint64_t *)calloc(words, 8);
    return v;
}
static iv iv_empty(void) /* default int_vector<0>: size 0, width 64 (int_vector.hpp width()) */
{
    return iv_make(0, 64);
}
static void iv_free(iv * v)
{
    free(v->d);
    v->d = NULL;
    v->size = 0;
}
static inline uint64_t iv_get(const iv * v, uint64_t i)
{
    return orc_read_int(v->d, i * v->width, v->width);
}
static inline void iv_set(iv * v, uint64_t i, uint64_t x)
{
    write_int(v->d, i * v->width, x, v->width);
}
/* int_vector<w>::serialize: u64 (width<<56 | bit_size) then ceil(bit_size/64) words
 * (int_vector.hpp:904-916,1978-2004) */
static size_t iv_serialize(const iv * v, orc_buf * out)
{
    uint64_t bits = v->size * v->width;
    buf_u64(out, ((uint64_t)v->width << 56) | bits);
    size_t nb = (size_t)((bits + 63) >> 6) * 8;
    buf_put(out, v->d, nb);
    return 8 + nb;
}
static size_t words_serialize(const uint64_t * w, uint64_t n_bits, uint8_t width, orc_buf * out)
{
    buf_u64(out, ((uint64_t)width << 56) | n_bits);
    size_t nb = (size_t)((n_bits + 63) >> 6) * 8;
    buf_put(out, w, nb);
    return 8 + nb;
}

static inline int bv_get(const uint64_t * w, uint64_t i) /* int_vector.hpp:1900-1904 */
{
    return (int)((w[i >> 6] >> (i & 63)) & 1);
}

/* util::cnt_one_bits (util.hpp:671-686): ones among the first n_bits */
static uint64_t cnt_one_bits(const uint64_t * w, uint64_t n_bits)
{
    uint64_t nw = (n_bits + 63) >> 6, r = 0;
    for (uint64_t i = 0; i < nw; ++i)
        r += orc_cnt(w[i]);
    if (n_bits & 63)
        r -= orc_cnt(w[nw - 1] & ~lo_set((unsigned)(n_bits & 63)));
    return r;
}

/* ======================================================================================== */
/* rank_support_v5<b,1>                                                                     */
/* ======================================================================================== */

struct orc_rank_v5
{
    const uint64_t * data; /* supported vector (not owned) */
    uint64_t n_bits;
    int bit;
    uint64_t * bb; /* m_basic_block */
    uint64_t bb_words;
    int has_vector; /* v != nullptr */
};

static inline uint64_t v5_word(const orc_rank_v5 * r, uint64_t w) /* trait: w or ~w (rank_support.hpp:111-160) */
{
    return r->bit ? w : ~w;
}

/* rank_support_v5.hpp:68-124 */
orc_rank_v5 * orc_rank_v5_build(const uint64_t * words, uint64_t n_bits, int bit)
{
    orc_rank_v5 * r = (orc_rank_v5 *)calloc(1, sizeof *r);
    r->data = words;
    r->n_bits = n_bits;
    r->bit = bit;
    r->has_vector = 1;
    if (n_bits == 0)
    { /* v->empty(): m_basic_block = int_vector<64>(2,0)  (:74-78) */
        r->bb_words = 2;
        r->bb = (uint64_t *)calloc(2, 8);
        return r;
    }
    uint64_t bbs = (((n_bits + 63) >> 11) + 1) << 1; /* :80 */
    r->bb_words = bbs;
    r->bb = (uint64_t *)calloc(bbs + 2, 8);
    const uint64_t * data = words;
    uint64_t j = 0;
    r->bb[0] = r->bb[1] = 0;
    uint64_t sum = orc_cnt(v5_word(r, *data));
    uint64_t second_level_cnt = 0, cnt_words = 1;
    for (uint64_t i = 1; i < ((n_bits + 63) >> 6); ++i, ++cnt_words)
    { /* :92-108 */
        if (cnt_words == 32)
        {
            j += 2;
            r->bb[j - 1] = second_level_cnt;
            r->bb[j] = r->bb[j - 2] + sum;
            second_level_cnt = sum = cnt_words = 0;
        }
        else if ((cnt_words % 6) == 0)
        {
            second_level_cnt |= sum << (60 - 12 * (cnt_words / 6));
        }
        sum += orc_cnt(v5_word(r, *(++data)));
    }
    if ((cnt_words % 6) == 0) /* :109-112 */
        second_level_cnt |= sum << (60 - 12 * (cnt_words / 6));
    if (cnt_words == 32)
    { /* :113-119 */
        j += 2;
        r->bb[j - 1] = second_level_cnt;
        r->bb[j] = r->bb[j - 2] + sum;
        r->bb[j + 1] = 0;
    }
    else
    {
        r->bb[j + 1] = second_level_cnt; /* :120-123 */
    }
    return r;
}

void orc_rank_v5_free(orc_rank_v5 * r)
{
    if (!r)
        return;
    free(r->bb);
    free(r);
}

/* rank_support_v5.hpp:131-149 with rank_support_trait<b,1>::word_rank / full_word_rank
 * (rank_support.hpp:121-129,146-154) */
uint64_t orc_rank_v5_rank(const orc_rank_v5 * r, uint64_t idx)
{
    const uint64_t * p = r->bb + ((idx >> 10) & UINT64_C(0xFFFFFFFFFFFFFFFE));
    uint64_t result = *p + ((*(p + 1) >> (60 - 12 * ((idx & 0x7FF) / (64 * 6)))) & UINT64_C(0x7FF))
                    + orc_cnt(v5_word(r, r->data[idx >> 6]) & lo_set((unsigned)(idx & 0x3F)));
    idx -= (idx & 0x3F);
    uint8_t to_do = (uint8_t)(((idx >> 6) & UINT64_C(0x1F)) % 6);
    --idx;
    while (to_do)
    {
        result += orc_cnt(v5_word(r, r->data[idx >> 6]));
        --to_do;
        idx -= 64;
    }
    return result;
}

void orc_rank_v5_batch(const orc_rank_v5 * r, const uint64_t * idx, uint64_t n, uint64_t * out)
{
    for (uint64_t q = 0; q < n; ++q)
        out[q] = orc_rank_v5_rank(r, idx[q]);
}

/* the same scalar loop on `threads` host threads, contiguous slices (queries are const-safe) */
typedef struct
{
    const orc_rank_v5 * r;
    const uint64_t * idx;
    uint64_t n;
    uint64_t * out;
} v5_job;
static void * v5_worker(void * a)
{
    v5_job * j = (v5_job *)a;
    orc_rank_v5_batch(j->r, j->idx, j->n, j->out);
    return NULL;
}
void orc_rank_v5_batch_mt(const orc_rank_v5 * r, const uint64_t * idx, uint64_t n, uint64_t * out, int threads)
{
    if (threads < 1)
        threads = 1;
    if (threads > 1024)
        threads = 1024;
    pthread_t * th = (pthread_t *)malloc((size_t)threads * sizeof *th);
    v5_job * jobs = (v5_job *)malloc((size_t)threads * sizeof *jobs);
    for (int t = 0; t < threads; ++t)
    {
        uint64_t lo = n * (uint64_t)t / (uint64_t)threads, hi = n * (uint64_t)(t + 1) / (uint64_t)threads;
        jobs[t].r = r;
        jobs[t].idx = idx + lo;
        jobs[t].n = hi - lo;
        jobs[t].out = out + lo;
        pthread_create(&th[t], NULL, v5_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t)
        pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

size_t orc_rank_v5_serialize(const orc_rank_v5 * r, orc_buf * out) /* :160-167 — int_vector<64> */
{
    return words_serialize(r->bb, r->bb_words * 64, 64, out);
}

/* ======================================================================================== */
/* select_support_mcl<b,1>                                                                  */
/* ======================================================================================== */

struct orc_select_mcl
{
    const uint64_t * data;
    uint64_t n_bits;
    int bit;
    uint32_t logn, logn2, logn4;
    uint64_t arg_cnt;
    iv superblock;
    iv * longsb; /* NULL or array[sb(+1)] */
    iv * mini;   /* array[sb] */
    uint64_t sb;
};

static inline uint64_t sel_word(const orc_select_mcl * s, uint64_t w)
{
    return s->bit ? w : ~w;
}
static inline int sel_found(const orc_select_mcl * s, uint64_t i) /* found_arg select_support.hpp:149-152,188-191 */
{
    return bv_get(s->data, i) == s->bit;
}

static void mcl_init_data(orc_select_mcl * s) /* select_support_mcl.hpp:448-465 */
{
    s->arg_cnt = 0;
    s->logn = orc_hi(((s->n_bits + 63) >> 6) << 6) + 1;
    s->logn2 = s->logn * s->logn;
    s->logn4 = s->logn2 * s->logn2;
    s->longsb = NULL;
    s->mini = NULL;
    s->superblock = iv_empty();
}

static uint64_t mcl_arg_cnt(const orc_select_mcl * s) /* select_support.hpp:132-135,171-174 */
{
    uint64_t ones = cnt_one_bits(s->data, s->n_bits);
    return s->bit ? ones : s->n_bits - ones;
}

/* select_support_mcl.hpp:207-266 */
static void mcl_init_slow(orc_select_mcl * s)
{
    mcl_init_data(s);
    s->arg_cnt = mcl_arg_cnt(s);
    const uint64_t SB = 4096;
    if (s->arg_cnt == 0)
        return;
    uint64_t sb = (s->arg_cnt + SB - 1) / SB;
    s->sb = sb;
    s->mini = (iv *)calloc(sb, sizeof(iv));
    for (uint64_t i = 0; i < sb; ++i)
        s->mini[i] = iv_empty();
    iv_free(&s->superblock);
    s->superblock = iv_make(sb, (uint8_t)s->logn);
    uint64_t * arg_position = (uint64_t *)malloc(SB * 8);
    uint64_t arg_cnt = 0, sb_cnt = 0;
    for (uint64_t i = 0; i < s->n_bits; ++i)
    {
        if (!sel_found(s, i))
            continue;
        arg_position[arg_cnt % SB] = i;
        ++arg_cnt;
        if (arg_cnt % SB == 0 || arg_cnt == s->arg_cnt)
        {
            iv_set(&s->superblock, sb_cnt, arg_position[0]);
            uint64_t last = arg_position[(arg_cnt - 1) % SB];
            uint64_t pos_diff = last - arg_position[0];
            if (pos_diff > s->logn4)
            { /* long block :242-252 */
                if (!s->longsb)
                {
                    s->longsb = (iv *)calloc(sb, sizeof(iv));
                    for (uint64_t k = 0; k < sb; ++k)
                        s->longsb[k] = iv_empty();
                }
                iv_free(&s->longsb[sb_cnt]);
                s->longsb[sb_cnt] = iv_make(SB, (uint8_t)(orc_hi(last) + 1));
                for (uint64_t j = 0; j <= (arg_cnt - 1) % SB; ++j)
                    iv_set(&s->longsb[sb_cnt], j, arg_position[j]);
            }
            else
            { /* short block :254-261 */
                iv_free(&s->mini[sb_cnt]);
                s->mini[sb_cnt] = iv_make(64, (uint8_t)(orc_hi(pos_diff) + 1));
                for (uint64_t j = 0; j <= (arg_cnt - 1) % SB; j += 64)
                    iv_set(&s->mini[sb_cnt], j / 64, arg_position[j] - arg_position[0]);
            }
            ++sb_cnt;
        }
    }
    free(arg_position);
}

/* select_support_mcl.hpp:269-381 */
static void mcl_init_fast(orc_select_mcl * s)
{
    mcl_init_data(s);
    s->arg_cnt = mcl_arg_cnt(s);
    const uint64_t SB = 64 * 64;
    if (s->arg_cnt == 0)
        return;
    uint64_t sb = (s->arg_cnt + SB - 1) / SB;
    s->sb = sb;
    s->mini = (iv *)calloc(sb, sizeof(iv));
    for (uint64_t i = 0; i < sb; ++i)
        s->mini[i] = iv_empty();
    iv_free(&s->superblock);
    s->superblock = iv_make(sb, (uint8_t)s->logn);
    uint64_t * arg_position = (uint64_t *)calloc(SB, 8);
    const uint64_t * data = s->data;
    uint64_t last_k64 = 1, sb_cnt = 0;
    uint64_t cnt_old = 0, cnt_new = 0, last_k64_sum = 1;
    const uint64_t n = s->n_bits;
    for (uint64_t i = 0; i < (((n + 63) >> 6) << 6); i += 64, ++data)
    {
        cnt_new += orc_cnt(sel_word(s, *data));
        if (cnt_new > s->arg_cnt) /* :299-300 padding clamp for (0,1) */
            cnt_new = s->arg_cnt;
        if (cnt_new >= last_k64_sum)
        {
            arg_position[last_k64 - 1] = i + orc_sel(sel_word(s, *data), (uint32_t)(last_k64_sum - cnt_old));
            last_k64 += 64;
            last_k64_sum += 64;
            if (last_k64 == SB + 1)
            {
                iv_set(&s->superblock, sb_cnt, arg_position[0]);
                uint64_t pos_of_last = arg_position[last_k64 - 65];
                for (uint64_t ii = arg_position[last_k64 - 65] + 1, j = last_k64 - 65; ii < n && j < SB; ++ii)
                    if (sel_found(s, ii))
                    {
                        pos_of_last = ii;
                        ++j;
                    }
                uint64_t pos_diff = pos_of_last - arg_position[0];
                if (pos_diff > s->logn4)
                { /* long block :325-350 */
                    if (!s->longsb)
                    {
                        s->longsb = (iv *)calloc(sb + 1, sizeof(iv));
                        for (uint64_t k = 0; k < sb + 1; ++k)
                            s->longsb[k] = iv_empty();
                    }
                    iv_free(&s->longsb[sb_cnt]);
                    s->longsb[sb_cnt] = iv_make(SB, (uint8_t)(orc_hi(pos_of_last) + 1));
                    for (uint64_t j = arg_position[0], k = 0; k < SB && j <= pos_of_last; ++j)
                        if (sel_found(s, j))
                            iv_set(&s->longsb[sb_cnt], k++, j);
                }
                else
                { /* :352-359 */
                    iv_free(&s->mini[sb_cnt]);
                    s->mini[sb_cnt] = iv_make(64, (uint8_t)(orc_hi(pos_diff) + 1));
                    for (uint64_t j = 0; j < SB; j += 64)
                        iv_set(&s->mini[sb_cnt], j / 64, arg_position[j] - arg_position[0]);
                }
                ++sb_cnt;
                last_k64 = 1;
            }
        }
        cnt_old = cnt_new;
    }
    if (last_k64 > 1)
    { /* handle last block: append long superblock :365-380 */
        if (!s->longsb)
        {
            s->longsb = (iv *)calloc(sb + 1, sizeof(iv));
            for (uint64_t k = 0; k < sb + 1; ++k)
                s->longsb[k] = iv_empty();
        }
        iv_free(&s->longsb[sb_cnt]);
        s->longsb[sb_cnt] = iv_make(SB, (uint8_t)(orc_hi(n - 1) + 1));
        for (uint64_t i = arg_position[0], k = 0; i < n; ++i)
            if (sel_found(s, i))
                iv_set(&s->longsb[sb_cnt], k++, i);
        ++sb_cnt;
    }
    free(arg_position);
}

orc_select_mcl * orc_select_mcl_build(const uint64_t * words, uint64_t n_bits, int bit)
{
    orc_select_mcl * s = (orc_select_mcl *)calloc(1, sizeof *s);
    s->data = words;
    s->n_bits = n_bits;
    s->bit = bit;
    /* constructor dispatch select_support_mcl.hpp:121-128 (t_pat_len == 1 here) */
    if (n_bits < 100000)
        mcl_init_slow(s);
    else
        mcl_init_fast(s);
    return s;
}

void orc_select_mcl_free(orc_select_mcl * s)
{
    if (!s)
        return;
    iv_free(&s->superblock);
    if (s->mini)
    {
        for (uint64_t i = 0; i < s->sb; ++i)
            iv_free(&s->mini[i]);
        free(s->mini);
    }
    if (s->longsb)
    {
        for (uint64_t i = 0; i < s->sb; ++i)
            iv_free(&s->longsb[i]);
        free(s->longsb);
    }
    free(s);
}

uint64_t orc_select_mcl_arg_cnt(const orc_select_mcl * s)
{
    return s->arg_cnt;
}

/* select_support_mcl.hpp:384-439 with select_support_trait<b,1> (select_support.hpp:128-203) */
uint64_t orc_select_mcl_select(const orc_select_mcl * s, uint64_t i)
{
    i = i - 1;
    uint64_t sb_idx = i >> 12, offset = i & 0xFFF;
    if (s->longsb != NULL && s->longsb[sb_idx].size != 0)
        return iv_get(&s->longsb[sb_idx], offset);
    if ((offset & 0x3F) == 0)
        return iv_get(&s->superblock, sb_idx) + iv_get(&s->mini[sb_idx], offset >> 6);
    i = i - (sb_idx << 12) - ((offset >> 6) << 6);
    uint64_t pos = iv_get(&s->superblock, sb_idx) + iv_get(&s->mini[sb_idx], offset >> 6) + 1;
    uint64_t word_pos = pos >> 6, word_off = pos & 0x3F;
    const uint64_t * data = s->data + word_pos;
    uint64_t w = sel_word(s, *data) & lo_unset((unsigned)word_off);
    uint64_t args = orc_cnt(w);
    if (args >= i)
        return (word_pos << 6) + orc_sel(w, (uint32_t)i);
    word_pos += 1;
    uint64_t sum_args = args;
    args = orc_cnt(sel_word(s, *(++data)));
    while (sum_args + args < i)
    {
        sum_args += args;
        args = orc_cnt(sel_word(s, *(++data)));
        word_pos += 1;
    }
    return (word_pos << 6) + orc_sel(sel_word(s, *data), (uint32_t)(i - sum_args));
}

void orc_select_mcl_batch(const orc_select_mcl * s, const uint64_t * i, uint64_t n, uint64_t * out)
{
    for (uint64_t q = 0; q < n; ++q)
        out[q] = orc_select_mcl_select(s, i[q]);
}

/* select_support_mcl.hpp:474-518 */
size_t orc_select_mcl_serialize(const orc_select_mcl * s, orc_buf * out)
{
    size_t w = 8;
    buf_u64(out, s->arg_cnt);
    uint64_t sb = (s->arg_cnt + 4095) >> 12;
    if (s->arg_cnt)
    {
        w += iv_serialize(&s->superblock, out);
        iv mol = iv_make(s->longsb ? sb : 0, 1); /* bit_vector mini_or_long */
        if (s->longsb)
            for (uint64_t i = 0; i < sb; ++i)
                iv_set(&mol, i, s->mini[i].size != 0);
        w += iv_serialize(&mol, out);
        for (uint64_t i = 0; i < sb; ++i)
        {
            if (mol.size != 0 && !iv_get(&mol, i))
                w += iv_serialize(&s->longsb[i], out);
            else
                w += iv_serialize(&s->mini[i], out);
        }
        iv_free(&mol);
    }
    return w;
}

/* ======================================================================================== */
/* sd_vector<> (Elias-Fano)                                                                 */
/* ======================================================================================== */

struct orc_sd
{
    uint64_t size;      /* length of the original bit vector */
    uint8_t wl;         /* width of the low parts */
    iv low;             /* m entries of wl bits */
    uint64_t * high;    /* m ones, 2^logm zeros */
    uint64_t high_bits;
    orc_select_mcl *sel1, *sel0; /* select supports of high */
};

/* the common tail of the constructors (sd_vector.hpp:223-231,271-279): widths and empty arrays */
static orc_sd * sd_alloc(uint64_t size, uint64_t m)
{
    orc_sd * v = (orc_sd *)calloc(1, sizeof *v);
    v->size = size;
    uint8_t logm = (uint8_t)(orc_hi(m) + 1), logn = (uint8_t)(orc_hi(size) + 1);
    if (logm == logn)
        --logm; /* to ensure logn - logm > 0 */
    v->wl = (uint8_t)(logn - logm);
    v->low = iv_make(m, v->wl);
    v->high_bits = m + (UINT64_C(1) << logm);
    v->high = (uint64_t *)calloc(((v->high_bits + 63) >> 6) + 2, 8);
    return v;
}
static void sd_init_supports(orc_sd * v)
{
    v->sel1 = orc_select_mcl_build(v->high, v->high_bits, 1);
    v->sel0 = orc_select_mcl_build(v->high, v->high_bits, 0);
}

/* sd_vector(bit_vector const&) sd_vector.hpp:217-257 */
orc_sd * orc_sd_build(const uint64_t * words, uint64_t n_bits)
{
    uint64_t m = cnt_one_bits(words, n_bits);
    orc_sd * v = sd_alloc(n_bits, m);
    uint64_t mm = 0, last_high = 0, highpos = 0;
    for (uint64_t i = 0; i < n_bits; ++i)
    {
        if (!bv_get(words, i))
            continue;
        uint64_t cur_high = i >> v->wl;
        highpos += cur_high - last_high; /* write cur_high - last_high zeros */
        last_high = cur_high;
        iv_set(&v->low, mm++, i & lo_set(v->wl)); /* int_vector truncates the most significant bits */
        v->high[highpos >> 6] |= UINT64_C(1) << (highpos & 63);
        ++highpos;
    }
    sd_init_supports(v);
    return v;
}

/* sd_vector(begin, end) sd_vector.hpp:259-305: size = last position + 1; an empty list gives an empty vector */
orc_sd * orc_sd_build_from_positions(const uint64_t * pos, uint64_t m)
{
    if (m == 0)
    {
        orc_sd * v = (orc_sd *)calloc(1, sizeof *v);
        v->low = iv_empty();
        v->high = (uint64_t *)calloc(2, 8);
        sd_init_supports(v);
        return v;
    }
    orc_sd * v = sd_alloc(pos[m - 1] + 1, m);
    uint64_t last_high = 0, highpos = 0;
    for (uint64_t mm = 0; mm < m; ++mm)
    {
        uint64_t cur_high = pos[mm] >> v->wl;
        highpos += cur_high - last_high;
        last_high = cur_high;
        iv_set(&v->low, mm, pos[mm] & lo_set(v->wl));
        v->high[highpos >> 6] |= UINT64_C(1) << (highpos & 63);
        ++highpos;
    }
    sd_init_supports(v);
    return v;
}

void orc_sd_free(orc_sd * v)
{
    if (!v)
        return;
    iv_free(&v->low);
    free(v->high);
    orc_select_mcl_free(v->sel1);
    orc_select_mcl_free(v->sel0);
    free(v);
}
uint64_t orc_sd_size(const orc_sd * v)
{
    return v->size;
}
uint64_t orc_sd_ones(const orc_sd * v)
{
    return v->low.size;
}
uint32_t orc_sd_wl(const orc_sd * v)
{
    return v->wl;
}

/* sd_vector::operator[] sd_vector.hpp:328-349 */
int orc_sd_access(const orc_sd * v, uint64_t i)
{
    uint64_t high_val = i >> v->wl;
    uint64_t sel_high = orc_select_mcl_select(v->sel0, high_val + 1);
    uint64_t rank_low = sel_high - high_val;
    if (rank_low == 0)
        return 0;
    uint64_t val_low = i & lo_set(v->wl);
    --sel_high;
    --rank_low;
    while (bv_get(v->high, sel_high) && iv_get(&v->low, rank_low) > val_low)
    {
        if (sel_high > 0)
        {
            --sel_high;
            --rank_low;
        }
        else
            return 0;
    }
    return bv_get(v->high, sel_high) && iv_get(&v->low, rank_low) == val_low;
}

/* rank_support_sd<b>::rank sd_vector.hpp:553-575 (+ adjust_rank :505-517) */
uint64_t orc_sd_rank(const orc_sd * v, uint64_t i, int bit)
{
    uint64_t r1;
    uint64_t high_val = i >> v->wl;
    uint64_t sel_high = orc_select_mcl_select(v->sel0, high_val + 1);
    uint64_t rank_low = sel_high - high_val;
    if (rank_low == 0)
        r1 = 0;
    else
    {
        uint64_t val_low = i & lo_set(v->wl);
        r1 = UINT64_MAX;
        do
        {
            if (!sel_high)
            {
                r1 = 0;
                break;
            }
            --sel_high;
            --rank_low;
        }
        while (bv_get(v->high, sel_high) && iv_get(&v->low, rank_low) >= val_low);
        if (r1 == UINT64_MAX)
            r1 = rank_low + 1;
    }
    return bit ? r1 : i - r1;
}

/* select_support_sd_trait<1>::select sd_vector.hpp:621-631 */
static uint64_t sd_select1(const orc_sd * v, uint64_t i)
{
    return iv_get(&v->low, i - 1) + ((orc_select_mcl_select(v->sel1, i) + 1 - i) << v->wl);
}
/* select_support_sd_trait<0>::select sd_vector.hpp:633-664 */
uint64_t orc_sd_select(const orc_sd * v, uint64_t i, int bit)
{
    if (bit)
        return sd_select1(v, i);
    uint64_t ones = v->low.size;
    uint64_t lb = 1, rb = ones + 1, r0 = 0, pos = UINT64_MAX;
    while (lb < rb)
    {
        uint64_t mid = lb + (rb - lb) / 2;
        uint64_t x = sd_select1(v, mid);
        uint64_t rank0 = x + 1 - mid;
        if (rank0 >= i)
            rb = mid;
        else
        {
            r0 = rank0;
            pos = x;
            lb = mid + 1;
        }
    }
    return pos + i - r0;
}

/* sd_vector::serialize sd_vector.hpp:435-445: size, wl, low, high, high_1_select, high_0_select */
size_t orc_sd_serialize(const orc_sd * v, orc_buf * out)
{
    size_t w = 9;
    buf_u64(out, v->size);
    buf_put(out, &v->wl, 1);
    w += iv_serialize(&v->low, out);
    w += words_serialize(v->high, v->high_bits, 1, out);
    w += orc_select_mcl_serialize(v->sel1, out);
    w += orc_select_mcl_serialize(v->sel0, out);
    return w;
}

/* ======================================================================================== */
/* rrr_vector<63, int_vector<>, 32>                                                         */
/* ======================================================================================== */

enum
{
    RRR_BS = 63,
    RRR_K = 32
};

/* binomial_table<64,uint64_t> (rrr_helper.hpp:194-237) and binomial_coefficients<63>::space
 * (rrr_helper.hpp:262-295); BINARY_SEARCH_THRESHOLD = n / MAX_LOG = 63/6 = 10 (:279) */
static uint64_t g_binom[65][65];
static uint16_t g_space[64];
static int g_binom_ready = 0;
enum
{
    RRR_BS_THRESHOLD = RRR_BS / 6
};

static void binom_init(void)
{
    if (g_binom_ready)
        return;
    const int n = 64;
    for (int k = 0; k <= n; ++k)
        g_binom[k][k] = 1;
    for (int k = 0; k <= n; ++k)
        g_binom[0][k] = 0;
    for (int nn = 0; nn <= n; ++nn)
        g_binom[nn][0] = 1;
    for (int nn = 1; nn <= n; ++nn)
        for (int k = 1; k <= n; ++k)
            g_binom[nn][k] = g_binom[nn - 1][k - 1] + g_binom[nn - 1][k];
    for (int k = 0; k <= RRR_BS; ++k)
        g_space[k] = (g_binom[RRR_BS][k] == 1) ? 0 : (uint16_t)(orc_hi(g_binom[RRR_BS][k]) + 1);
    g_binom_ready = 1;
}

static uint64_t rrr_bin_to_nr(uint64_t bin) /* rrr_helper.hpp:346-366 */
{
    if (bin == 0 || bin == lo_set(RRR_BS))
        return 0;
    uint64_t nr = 0;
    uint16_t k = (uint16_t)orc_cnt(bin);
    uint16_t nn = RRR_BS;
    while (bin != 0)
    {
        if (bin & 1)
        {
            nr += g_binom[nn - 1][k];
            --k;
        }
        bin >>= 1;
        --nn;
    }
    return nr;
}

static int rrr_decode_bit(uint16_t k, uint64_t nr, uint16_t off) /* rrr_helper.hpp:369-436 */
{
    const uint16_t n = RRR_BS;
    if (k == n)
        return 1;
    else if (k == 0)
        return 0;
    else if (k == 1)
        return (uint64_t)(n - nr - 1) == off;
    uint16_t nn = n;
    if (k + 1 < RRR_BS_THRESHOLD + 1)
    {
        while (k > 1)
        {
            uint16_t nn_lb = k, nn_rb = nn + 1;
            while (nn_lb < nn_rb)
            {
                uint16_t nn_mid = (nn_lb + nn_rb) / 2;
                if (nr >= g_binom[nn_mid - 1][k])
                    nn_lb = nn_mid + 1;
                else
                    nn_rb = nn_mid;
            }
            nn = nn_lb - 1;
            if (n - nn >= off)
                return (n - nn) == off;
            nr -= g_binom[nn - 1][k];
            --k;
            --nn;
        }
    }
    else
    {
        int i = 0;
        while (k > 1)
        {
            if (i > off)
                return 0;
            if (nr >= g_binom[nn - 1][k])
            {
                nr -= g_binom[nn - 1][k];
                --k;
                if (i == off)
                    return 1;
            }
            --nn;
            ++i;
        }
    }
    return (uint64_t)(n - nr - 1) == off;
}

static uint16_t rrr_decode_popcount(uint16_t k, uint64_t nr, uint16_t off) /* rrr_helper.hpp:487-555 */
{
    const uint16_t n = RRR_BS;
    if (k == n)
        return off;
    else if (k == 0)
        return 0;
    else if (k == 1)
        return (uint64_t)(n - nr - 1) < off;
    uint16_t result = 0;
    uint16_t nn = n;
    if (k + 1 < RRR_BS_THRESHOLD + 1)
    {
        while (k > 1)
        {
            uint16_t nn_lb = k, nn_rb = nn + 1;
            while (nn_lb < nn_rb)
            {
                uint16_t nn_mid = (nn_lb + nn_rb) / 2;
                if (nr >= g_binom[nn_mid - 1][k])
                    nn_lb = nn_mid + 1;
                else
                    nn_rb = nn_mid;
            }
            nn = nn_lb - 1;
            if (n - nn >= off)
                return result;
            ++result;
            nr -= g_binom[nn - 1][k];
            --k;
            --nn;
        }
    }
    else
    {
        int i = 0;
        while (k > 1)
        {
            if (i >= off)
                return result;
            if (nr >= g_binom[nn - 1][k])
            {
                nr -= g_binom[nn - 1][k];
                --k;
                ++result;
            }
            --nn;
            ++i;
        }
    }
    return result + ((uint64_t)(n - nr - 1) < off);
}

static uint16_t rrr_decode_select(uint16_t k, uint64_t nr, uint16_t sel) /* rrr_helper.hpp:559-614 */
{
    const uint16_t n = RRR_BS;
    if (k == n)
        return sel - 1;
    else if (k == 1 && sel == 1)
        return (uint16_t)(n - nr - 1);
    uint16_t nn = n;
    if (sel + 1 < RRR_BS_THRESHOLD + 1)
    {
        while (sel > 0)
        {
            uint16_t nn_lb = k, nn_rb = nn + 1;
            while (nn_lb < nn_rb)
            {
                uint16_t nn_mid = (nn_lb + nn_rb) / 2;
                if (nr >= g_binom[nn_mid - 1][k])
                    nn_lb = nn_mid + 1;
                else
                    nn_rb = nn_mid;
            }
            nn = nn_lb - 1;
            nr -= g_binom[nn - 1][k];
            --sel;
            --nn;
            --k;
        }
        return n - nn - 1;
    }
    else
    {
        int i = 0;
        while (sel > 0)
        {
            if (nr >= g_binom[nn - 1][k])
            {
                nr -= g_binom[nn - 1][k];
                --sel;
                --k;
            }
            --nn;
            ++i;
        }
        return (uint16_t)(i - 1);
    }
}

/* decode_select_bitpattern<0,1> rrr_helper.hpp:619-649 (pattern "0" of length 1) */
static uint16_t rrr_decode_select0(uint16_t k, uint64_t nr, uint16_t sel)
{
    int i = 0;
    uint16_t nn = RRR_BS;
    while (sel > 0)
    {
        int one = 0;
        if (nr >= g_binom[nn - 1][k])
        {
            nr -= g_binom[nn - 1][k];
            one = 1;
            --k;
        }
        --nn;
        ++i;
        if (!one)
            --sel;
    }
    return (uint16_t)(i - 1);
}

struct orc_rrr
{
    uint64_t size;
    iv bt;     /* width 6 */
    iv btnr;   /* bit_vector */
    iv btnrp;  /* int_vector<> */
    iv rank;   /* int_vector<> */
    iv invert; /* bit_vector */
};

static uint64_t bv_get_int(const uint64_t * w, uint64_t pos, uint64_t len)
{
    return orc_read_int(w, pos, (uint8_t)len);
}

/* rrr_vector.hpp:158-270 */
orc_rrr * orc_rrr_build(const uint64_t * words, uint64_t n_bits)
{
    binom_init();
    orc_rrr * r = (orc_rrr *)calloc(1, sizeof *r);
    const uint64_t bs = RRR_BS, tk = RRR_K;
    r->size = n_bits;
    /* a copy padded with two zero words so that 63-bit reads at the tail stay in bounds and bits
     * beyond n_bits read as the reference's get_int would (it never reads past size) */
    uint64_t nw = (n_bits + 63) >> 6;
    uint64_t * bv = (uint64_t *)calloc(nw + 2, 8);
    memcpy(bv, words, nw * 8);
    iv bt_array = iv_make((n_bits + bs) / bs, (uint8_t)(orc_hi(bs) + 1));
    uint64_t pos = 0, i = 0, x;
    uint64_t btnr_pos = 0, sum_rank = 0;
    while (pos + bs <= n_bits)
    { /* :169-175 */
        x = orc_cnt(bv_get_int(bv, pos, bs));
        iv_set(&bt_array, i++, x);
        sum_rank += x;
        btnr_pos += g_space[x];
        pos += bs;
    }
    if (pos < n_bits)
    { /* :176-181 */
        x = orc_cnt(bv_get_int(bv, pos, n_bits - pos));
        iv_set(&bt_array, i++, x);
        sum_rank += x;
        btnr_pos += g_space[x];
    }
    uint64_t nsb = (bt_array.size + tk - 1) / tk;
    r->btnr = iv_make(btnr_pos > 64 ? btnr_pos : 64, 1);                               /* :183 */
    r->btnrp = iv_make(nsb, (uint8_t)(orc_hi(btnr_pos) + 1));                          /* :184 */
    r->rank = iv_make(nsb + ((n_bits % (tk * bs)) > 0), (uint8_t)(orc_hi(sum_rank) + 1)); /* :185-186 */
    r->invert = iv_make(nsb, 1);                                                       /* :189 */
    pos = 0;
    i = 0;
    btnr_pos = 0;
    sum_rank = 0;
    int invert = 0;
    while (pos + bs <= n_bits)
    { /* :196-241 */
        if ((i % tk) == 0)
        {
            iv_set(&r->btnrp, i / tk, btnr_pos);
            iv_set(&r->rank, i / tk, sum_rank);
            if (i + tk <= bt_array.size)
            {
                uint64_t gt_half = 0;
                for (uint64_t j = i; j < i + tk; ++j)
                    if (iv_get(&bt_array, j) > bs / 2)
                        ++gt_half;
                if (gt_half > (tk / 2))
                {
                    iv_set(&r->invert, i / tk, 1);
                    for (uint64_t j = i; j < i + tk; ++j)
                        iv_set(&bt_array, j, bs - iv_get(&bt_array, j));
                    invert = 1;
                }
                else
                    invert = 0;
            }
            else
                invert = 0;
        }
        x = iv_get(&bt_array, i++);
        uint16_t space = g_space[x];
        sum_rank += invert ? (bs - x) : x;
        if (space)
        {
            uint64_t bin = bv_get_int(bv, pos, bs);
            write_int(r->btnr.d, btnr_pos, rrr_bin_to_nr(bin), (uint8_t)space);
        }
        btnr_pos += space;
        pos += bs;
    }
    if (pos < n_bits)
    { /* :242-261 */
        if ((i % tk) == 0)
        {
            iv_set(&r->btnrp, i / tk, btnr_pos);
            iv_set(&r->rank, i / tk, sum_rank);
            iv_set(&r->invert, i / tk, 0);
            invert = 0;
        }
        x = iv_get(&bt_array, i++);
        uint16_t space = g_space[x];
        sum_rank += invert ? (bs - x) : x;
        if (space)
        {
            uint64_t bin = bv_get_int(bv, pos, n_bits - pos);
            write_int(r->btnr.d, btnr_pos, rrr_bin_to_nr(bin), (uint8_t)space);
        }
        btnr_pos += space;
    }
    iv_set(&r->rank, r->rank.size - 1, sum_rank); /* :268 */
    r->bt = bt_array;                             /* :269 */
    free(bv);
    return r;
}

void orc_rrr_free(orc_rrr * r)
{
    if (!r)
        return;
    iv_free(&r->bt);
    iv_free(&r->btnr);
    iv_free(&r->btnrp);
    iv_free(&r->rank);
    iv_free(&r->invert);
    free(r);
}

uint64_t orc_rrr_size(const orc_rrr * r)
{
    return r->size;
}

/* rank_support_rrr<1,63>::rank rrr_vector.hpp:503-544; for t_b=0 adjust_rank = i - rank (:446-458) */
uint64_t orc_rrr_rank(const orc_rrr * r, uint64_t i, int bit)
{
    const uint64_t bs = RRR_BS, tk = RRR_K;
    uint64_t bt_idx = i / bs;
    uint64_t sample_pos = bt_idx / tk;
    uint64_t btnrp = iv_get(&r->btnrp, sample_pos);
    uint64_t rank = iv_get(&r->rank, sample_pos);
    uint64_t res;
    if (sample_pos + 1 < r->rank.size)
    {
        uint64_t diff_rank = iv_get(&r->rank, sample_pos + 1) - rank;
        if (diff_rank == 0)
        {
            res = rank;
            goto done;
        }
        else if (diff_rank == bs * tk)
        {
            res = rank + i - sample_pos * tk * bs;
            goto done;
        }
    }
    {
        int inv = (int)iv_get(&r->invert, sample_pos);
        for (uint64_t j = sample_pos * tk; j < bt_idx; ++j)
        {
            uint16_t b = (uint16_t)iv_get(&r->bt, j);
            rank += (inv ? bs - b : b);
            btnrp += g_space[b];
        }
        uint16_t off = (uint16_t)(i % bs);
        if (!off)
        {
            res = rank;
            goto done;
        }
        uint16_t bt = (uint16_t)(inv ? bs - iv_get(&r->bt, bt_idx) : iv_get(&r->bt, bt_idx));
        uint16_t btnrlen = g_space[bt];
        uint64_t btnr = orc_read_int(r->btnr.d, btnrp, (uint8_t)btnrlen);
        res = rank + rrr_decode_popcount(bt, btnr, off);
    }
done:
    return bit ? res : i - res;
}

/* select_support_rrr<1,63>::select1 rrr_vector.hpp:639-682 */
static uint64_t rrr_select1(const orc_rrr * r, uint64_t i)
{
    const uint64_t bs = RRR_BS, tk = RRR_K;
    if (iv_get(&r->rank, r->rank.size - 1) < i)
        return r->size;
    uint64_t begin = 0, end = r->rank.size - 1;
    uint64_t idx, rank;
    while (end - begin > 1)
    {
        idx = (begin + end) >> 1;
        rank = iv_get(&r->rank, idx);
        if (rank >= i)
            end = idx;
        else
            begin = idx;
    }
    rank = iv_get(&r->rank, begin);
    idx = begin * tk;
    uint64_t diff_rank = iv_get(&r->rank, end) - rank;
    if (diff_rank == bs * tk)
        return idx * bs + i - rank - 1;
    int inv = (int)iv_get(&r->invert, begin);
    uint64_t btnrp = iv_get(&r->btnrp, begin);
    uint16_t bt = 0, btnrlen = 0;
    while (i > rank)
    {
        bt = (uint16_t)iv_get(&r->bt, idx++);
        bt = inv ? (uint16_t)(bs - bt) : bt;
        rank += bt;
        btnrp += (btnrlen = g_space[bt]);
    }
    rank -= bt;
    uint64_t btnr = orc_read_int(r->btnr.d, btnrp - btnrlen, (uint8_t)btnrlen);
    return (idx - 1) * bs + rrr_decode_select(bt, btnr, (uint16_t)(i - rank));
}

/* select_support_rrr<0,63>::select0 rrr_vector.hpp:684-726 */
static uint64_t rrr_select0(const orc_rrr * r, uint64_t i)
{
    const uint64_t bs = RRR_BS, tk = RRR_K;
    if ((r->size - iv_get(&r->rank, r->rank.size - 1)) < i)
        return r->size;
    uint64_t begin = 0, end = r->rank.size - 1;
    uint64_t idx, rank;
    while (end - begin > 1)
    {
        idx = (begin + end) >> 1;
        rank = idx * bs * tk - iv_get(&r->rank, idx);
        if (rank >= i)
            end = idx;
        else
            begin = idx;
    }
    rank = begin * bs * tk - iv_get(&r->rank, begin);
    idx = begin * tk;
    if (iv_get(&r->rank, end) == iv_get(&r->rank, begin))
        return idx * bs + i - rank - 1;
    int inv = (int)iv_get(&r->invert, begin);
    uint64_t btnrp = iv_get(&r->btnrp, begin);
    uint16_t bt = 0, btnrlen = 0;
    while (i > rank)
    {
        bt = (uint16_t)iv_get(&r->bt, idx++);
        bt = inv ? (uint16_t)(bs - bt) : bt;
        rank += (bs - bt);
        btnrp += (btnrlen = g_space[bt]);
    }
    rank -= (bs - bt);
    uint64_t btnr = orc_read_int(r->btnr.d, btnrp - btnrlen, (uint8_t)btnrlen);
    return (idx - 1) * bs + rrr_decode_select0(bt, btnr, (uint16_t)(i - rank));
}

uint64_t orc_rrr_select(const orc_rrr * r, uint64_t i, int bit)
{
    return bit ? rrr_select1(r, i) : rrr_select0(r, i);
}

/* rrr_vector::operator[] rrr_vector.hpp:276-298 */
int orc_rrr_access(const orc_rrr * r, uint64_t i)
{
    const uint64_t bs = RRR_BS, tk = RRR_K;
    uint64_t bt_idx = i / bs;
    uint16_t bt = (uint16_t)iv_get(&r->bt, bt_idx);
    uint64_t sample_pos = bt_idx / tk;
    if (iv_get(&r->invert, sample_pos))
        bt = (uint16_t)(bs - bt);
    if (bt == 0 || bt == bs)
        return bt > 0;
    uint16_t off = (uint16_t)(i % bs);
    uint64_t btnrp = iv_get(&r->btnrp, sample_pos);
    for (uint64_t j = sample_pos * tk; j < bt_idx; ++j)
        btnrp += g_space[iv_get(&r->bt, j)];
    uint16_t btnrlen = g_space[bt];
    uint64_t btnr = orc_read_int(r->btnr.d, btnrp, (uint8_t)btnrlen);
    return rrr_decode_bit(bt, btnr, off);
}

void orc_rrr_rank_batch(const orc_rrr * r, int bit, const uint64_t * i, uint64_t n, uint64_t * out)
{
    for (uint64_t q = 0; q < n; ++q)
        out[q] = orc_rrr_rank(r, i[q], bit);
}
void orc_rrr_select_batch(const orc_rrr * r, int bit, const uint64_t * i, uint64_t n, uint64_t * out)
{
    for (uint64_t q = 0; q < n; ++q)
        out[q] = orc_rrr_select(r, i[q], bit);
}

size_t orc_rrr_serialize(const orc_rrr * r, orc_buf * out) /* rrr_vector.hpp:366-378 */
{
    size_t w = 8;
    buf_u64(out, r->size);
    w += iv_serialize(&r->bt, out);
    w += iv_serialize(&r->btnr, out);
    w += iv_serialize(&r->btnrp, out);
    w += iv_serialize(&r->rank, out);
    w += iv_serialize(&r->invert, out);
    return w;
}

/* ======================================================================================== */
/* wt_huff over bytes: wt_pc<huff_shape, bit_vector, rank_support_v5<>, ...>                 */
/* ======================================================================================== */

#define WT_UNDEF16 0xFFFFu
#define PC_UNDEF UINT64_MAX

typedef struct /* _node<_byte_tree> wt_helper.hpp:92-112 */
{
    uint64_t bv_pos, bv_pos_rank;
    uint16_t parent, child[2];
} wt_node;

struct orc_wt
{
    uint64_t size, sigma;
    uint64_t * bv; /* m_bv words (+1 padding) */
    uint64_t bv_size;
    orc_rank_v5 * bv_rank;
    orc_select_mcl * sel1;
    orc_select_mcl * sel0;
    wt_node * nodes;
    uint64_t n_nodes;
    uint16_t c_to_leaf[256];
    uint64_t path[256];
};

typedef struct /* pc_node wt_helper.hpp:72-89 */
{
    uint64_t freq, sym, parent, child[2];
} pc_node;

/* min-heap of (freq, node) pairs = std::priority_queue<tPII, vector<tPII>, greater<tPII>>
 * (wt_huff.hpp:70-80): ordering is lexicographic on the pair */
typedef struct
{
    uint64_t f, v;
} hp;
static int hp_less(hp a, hp b)
{
    return a.f < b.f || (a.f == b.f && a.v < b.v);
}
static void heap_push(hp * h, size_t * n, hp x)
{
    size_t i = (*n)++;
    h[i] = x;
    while (i > 0)
    {
        size_t p = (i - 1) / 2;
        if (!hp_less(h[i], h[p]))
            break;
        hp t = h[i];
        h[i] = h[p];
        h[p] = t;
        i = p;
    }
}
static hp heap_pop(hp * h, size_t * n)
{
    hp top = h[0];
    h[0] = h[--(*n)];
    size_t i = 0;
    for (;;)
    {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && hp_less(h[l], h[m]))
            m = l;
        if (r < *n && hp_less(h[r], h[m]))
            m = r;
        if (m == i)
            break;
        hp t = h[i];
        h[i] = h[m];
        h[m] = t;
        i = m;
    }
    return top;
}

/* wt_pc::insert_char wt_pc.hpp:97-111 (set_int of `times` one bits) */
static void wt_insert_char(orc_wt * wt, uint8_t chr, uint64_t * bv_node_pos, uint64_t times)
{
    uint64_t p = wt->path[chr];
    uint32_t path_len = (uint32_t)(p >> 56);
    uint16_t v = 0;
    for (uint32_t l = 0; l < path_len; ++l, p >>= 1)
    {
        if (p & 1)
            write_int(wt->bv, bv_node_pos[v], UINT64_MAX, (uint8_t)times);
        bv_node_pos[v] += times;
        v = wt->nodes[v].child[p & 1];
    }
}

orc_wt * orc_wt_build(const uint8_t * text, uint64_t n)
{
    orc_wt * wt = (orc_wt *)calloc(1, sizeof *wt);
    wt->size = n;
    for (int c = 0; c < 256; ++c)
        wt->c_to_leaf[c] = WT_UNDEF16;
    if (n == 0)
    { /* wt_pc.hpp:196-197: default members; rank/select supports of an empty bit vector */
        wt->bv = (uint64_t *)calloc(2, 8);
        wt->bv_rank = orc_rank_v5_build(wt->bv, 0, 1);
        wt->sel1 = orc_select_mcl_build(wt->bv, 0, 1);
        wt->sel0 = orc_select_mcl_build(wt->bv, 0, 0);
        return wt;
    }
    /* 1. calculate_character_occurences wt_helper.hpp:45-60 */
    uint64_t C[256];
    memset(C, 0, sizeof C);
    for (uint64_t i = 0; i < n; ++i)
        ++C[text[i]];
    /* 2. effective alphabet size :62-71 */
    for (int c = 0; c < 256; ++c)
        wt->sigma += C[c] > 0;
    /* 3a. _huff_shape::construct_tree wt_huff.hpp:83-115 */
    pc_node temp[512];
    size_t nt = 0;
    hp heap[512];
    size_t hn = 0;
    for (uint64_t c = 0; c < 256; ++c)
        if (C[c] > 0)
        {
            hp e = {C[c], nt};
            heap_push(heap, &hn, e);
            pc_node nd = {C[c], c, PC_UNDEF, {PC_UNDEF, PC_UNDEF}};
            temp[nt++] = nd;
        }
    while (hn > 1)
    {
        hp v1 = heap_pop(heap, &hn), v2 = heap_pop(heap, &hn);
        temp[v1.v].parent = nt;
        temp[v2.v].parent = nt;
        uint64_t frq_sum = v1.f + v2.f;
        hp e = {frq_sum, nt};
        heap_push(heap, &hn, e);
        pc_node nd = {frq_sum, 0, PC_UNDEF, {v1.v, v2.v}};
        temp[nt++] = nd;
    }
    /* 3b. _byte_tree(temp_nodes, bv_size) BFS layout wt_helper.hpp:230-275 */
    wt->n_nodes = nt;
    wt->nodes = (wt_node *)calloc(nt, sizeof(wt_node));
    /* "m_nodes[i] = pc_node" copies freq->bv_pos, sym->bv_pos_rank, parent, children (:114-122) */
    uint64_t tmp_child[512][2];
#define ASSIGN_NODE(dst, src)                                                                                      \
    do {                                                                                                           \
        wt->nodes[dst].bv_pos = temp[src].freq;                                                                    \
        wt->nodes[dst].bv_pos_rank = temp[src].sym;                                                                \
        wt->nodes[dst].parent = (uint16_t)temp[src].parent;                                                        \
        tmp_child[dst][0] = temp[src].child[0];                                                                    \
        tmp_child[dst][1] = temp[src].child[1];                                                                    \
    } while (0)
    ASSIGN_NODE(0, nt - 1);
    uint64_t bv_size = 0;
    size_t node_cnt = 1;
    uint16_t last_parent = WT_UNDEF16;
    uint16_t queue[512];
    size_t qh = 0, qt = 0;
    queue[qt++] = 0;
    while (qh < qt)
    {
        uint16_t idx = queue[qh++];
        uint64_t frq = wt->nodes[idx].bv_pos;
        wt->nodes[idx].bv_pos = bv_size;
        int is_leaf = (tmp_child[idx][0] == PC_UNDEF);
        if (!is_leaf)
            bv_size += frq;
        if (idx > 0)
        {
            uint16_t par = wt->nodes[idx].parent;
            if (last_parent != par)
                wt->nodes[par].child[0] = idx;
            else
                wt->nodes[par].child[1] = idx;
            last_parent = par;
        }
        if (!is_leaf)
        {
            for (uint32_t k = 0; k < 2; ++k)
            {
                uint64_t src = tmp_child[idx][k];
                ASSIGN_NODE(node_cnt, src);
                wt->nodes[node_cnt].parent = idx;
                queue[qt++] = (uint16_t)node_cnt;
                tmp_child[idx][k] = node_cnt; /* m_nodes[idx].child[k] = node_cnt++ */
                ++node_cnt;
            }
        }
        else
        {
            wt->nodes[idx].child[0] = wt->nodes[idx].child[1] = WT_UNDEF16;
        }
    }
#undef ASSIGN_NODE
    /* inner nodes: child[] were fixed by their children's visits above (:252-259); make sure leaves stay undef */
    for (size_t v = 0; v < nt; ++v)
        if (tmp_child[v][0] != PC_UNDEF)
        {
            wt->nodes[v].child[0] = (uint16_t)tmp_child[v][0];
            wt->nodes[v].child[1] = (uint16_t)tmp_child[v][1];
        }
    wt->nodes[0].parent = WT_UNDEF16;
    /* m_c_to_leaf :276-284 */
    for (size_t v = 0; v < nt; ++v)
        if (wt->nodes[v].child[0] == WT_UNDEF16)
            wt->c_to_leaf[(uint8_t)wt->nodes[v].bv_pos_rank] = (uint16_t)v;
    /* m_path :289-316 */
    for (uint32_t c = 0, prev_c = 0; c < 256; ++c)
    {
        if (wt->c_to_leaf[c] != WT_UNDEF16)
        {
            uint16_t v = wt->c_to_leaf[c];
            uint64_t pw = 0, pl = 0;
            while (v != 0)
            {
                pw <<= 1;
                if (wt->nodes[wt->nodes[v].parent].child[1] == v)
                    pw |= 1;
                ++pl;
                v = wt->nodes[v].parent;
            }
            wt->path[c] = pw | (pl << 56);
            prev_c = c;
        }
        else
        {
            wt->path[c] = prev_c;
        }
    }
    /* 4. fill the bit vector wt_pc.hpp:211-242 */
    wt->bv_size = bv_size;
    wt->bv = (uint64_t *)calloc(((bv_size + 63) >> 6) + 2, 8);
    uint64_t bv_node_pos[512];
    for (size_t v = 0; v < nt; ++v)
        bv_node_pos[v] = wt->nodes[v].bv_pos;
    uint8_t old_chr = text[0];
    uint32_t times = 0;
    for (uint64_t i = 0; i < n; ++i)
    {
        uint8_t chr = text[i];
        if (chr != old_chr)
        {
            wt_insert_char(wt, old_chr, bv_node_pos, times);
            times = 1;
            old_chr = chr;
        }
        else
        {
            ++times;
            if (times == 64)
            {
                wt_insert_char(wt, old_chr, bv_node_pos, times);
                times = 0;
            }
        }
    }
    if (times > 0)
        wt_insert_char(wt, old_chr, bv_node_pos, times);
    /* 5. rank/select supports (:244-245) */
    wt->bv_rank = orc_rank_v5_build(wt->bv, bv_size, 1);
    wt->sel1 = orc_select_mcl_build(wt->bv, bv_size, 1);
    wt->sel0 = orc_select_mcl_build(wt->bv, bv_size, 0);
    /* 6. init_node_ranks wt_helper.hpp:320-327 */
    for (size_t v = 0; v < nt; ++v)
        if (wt->nodes[v].child[0] != WT_UNDEF16)
            wt->nodes[v].bv_pos_rank = orc_rank_v5_rank(wt->bv_rank, wt->nodes[v].bv_pos);
    return wt;
}

void orc_wt_free(orc_wt * wt)
{
    if (!wt)
        return;
    orc_rank_v5_free(wt->bv_rank);
    orc_select_mcl_free(wt->sel1);
    orc_select_mcl_free(wt->sel0);
    free(wt->bv);
    free(wt->nodes);
    free(wt);
}

uint64_t orc_wt_size(const orc_wt * wt)
{
    return wt->size;
}
uint64_t orc_wt_sigma(const orc_wt * wt)
{
    return wt->sigma;
}
uint64_t orc_wt_bv_size(const orc_wt * wt)
{
    return wt->bv_size;
}
const uint64_t * orc_wt_bv_words(const orc_wt * wt)
{
    return wt->bv;
}

void orc_wt_code_lengths(const orc_wt * wt, uint8_t len_out[256])
{
    for (int c = 0; c < 256; ++c)
        len_out[c] = wt->c_to_leaf[c] == WT_UNDEF16 ? 0 : (uint8_t)(wt->path[c] >> 56);
}

/* wt_pc::rank wt_pc.hpp:371-399 */
uint64_t orc_wt_rank(const orc_wt * wt, uint64_t i, uint8_t c)
{
    if (wt->c_to_leaf[c] == WT_UNDEF16)
        return 0;
    if (wt->sigma == 1)
        return i;
    uint64_t p = wt->path[c];
    uint32_t path_len = (uint32_t)(p >> 56);
    uint64_t result = i;
    uint16_t v = 0;
    for (uint32_t l = 0; l < path_len && result; ++l, p >>= 1)
    {
        uint64_t r1 = orc_rank_v5_rank(wt->bv_rank, wt->nodes[v].bv_pos + result) - wt->nodes[v].bv_pos_rank;
        if (p & 1)
            result = r1;
        else
            result -= r1;
        v = wt->nodes[v].child[p & 1];
    }
    return result;
}

void orc_wt_rank_batch(const orc_wt * wt, const uint64_t * i, const uint8_t * c, uint64_t n, uint64_t * out)
{
    for (uint64_t q = 0; q < n; ++q)
        out[q] = orc_wt_rank(wt, i[q], c[q]);
}

/* wt_pc::operator[] wt_pc.hpp:336-357 */
uint8_t orc_wt_access(const orc_wt * wt, uint64_t i)
{
    uint16_t v = 0;
    if (wt->n_nodes == 0)
        return 0;
    while (wt->nodes[v].child[0] != WT_UNDEF16)
    {
        uint64_t pos = wt->nodes[v].bv_pos + i;
        if (bv_get(wt->bv, pos))
        {
            i = orc_rank_v5_rank(wt->bv_rank, pos) - wt->nodes[v].bv_pos_rank;
            v = wt->nodes[v].child[1];
        }
        else
        {
            i -= orc_rank_v5_rank(wt->bv_rank, pos) - wt->nodes[v].bv_pos_rank;
            v = wt->nodes[v].child[0];
        }
    }
    return (uint8_t)wt->nodes[v].bv_pos_rank;
}

/* wt_pc::inverse_select wt_pc.hpp:411-430 */
uint64_t orc_wt_inverse_select(const orc_wt * wt, uint64_t i, uint8_t * c_out)
{
    uint16_t v = 0;
    while (wt->nodes[v].child[0] != WT_UNDEF16)
    {
        uint64_t pos = wt->nodes[v].bv_pos + i;
        if (bv_get(wt->bv, pos))
        {
            i = orc_rank_v5_rank(wt->bv_rank, pos) - wt->nodes[v].bv_pos_rank;
            v = wt->nodes[v].child[1];
        }
        else
        {
            i -= orc_rank_v5_rank(wt->bv_rank, pos) - wt->nodes[v].bv_pos_rank;
            v = wt->nodes[v].child[0];
        }
    }
    *c_out = (uint8_t)wt->nodes[v].bv_pos_rank;
    return i;
}

/* wt_pc::select wt_pc.hpp:443-474 */
uint64_t orc_wt_select(const orc_wt * wt, uint64_t i, uint8_t c)
{
    uint16_t v = wt->c_to_leaf[c];
    if (v == WT_UNDEF16)
        return wt->size; /* c not in the text -> a position right of the end (:447-450) */
    if (wt->sigma == 1)
        return i - 1 < wt->size ? i - 1 : wt->size;
    uint64_t result = i - 1;
    uint64_t p = wt->path[c];
    uint32_t path_len = (uint32_t)(p >> 56);
    p <<= (64 - path_len);
    for (uint32_t l = 0; l < path_len; ++l, p <<= 1)
    {
        uint16_t par = wt->nodes[v].parent;
        if ((p & UINT64_C(0x8000000000000000)) == 0)
            result = orc_select_mcl_select(wt->sel0, wt->nodes[par].bv_pos - wt->nodes[par].bv_pos_rank + result + 1)
                   - wt->nodes[par].bv_pos;
        else
            result = orc_select_mcl_select(wt->sel1, wt->nodes[par].bv_pos_rank + result + 1) - wt->nodes[par].bv_pos;
        v = par;
    }
    return result;
}

/* _byte_tree::serialize wt_helper.hpp:362-375 with _node::serialize :139-150 (22 bytes per node) */
static size_t wt_tree_serialize(const orc_wt * wt, orc_buf * out)
{
    size_t w = 8;
    buf_u64(out, wt->n_nodes);
    for (uint64_t v = 0; v < wt->n_nodes; ++v)
    {
        buf_u64(out, wt->nodes[v].bv_pos);
        buf_u64(out, wt->nodes[v].bv_pos_rank);
        buf_put(out, &wt->nodes[v].parent, 2);
        buf_put(out, wt->nodes[v].child, 4);
        w += 22;
    }
    buf_put(out, wt->c_to_leaf, 512);
    buf_put(out, wt->path, 2048);
    return w + 512 + 2048;
}

size_t orc_wt_serialize(const orc_wt * wt, int select_is_mcl, orc_buf * out) /* wt_pc.hpp:713-726 */
{
    size_t w = 16;
    buf_u64(out, wt->size);
    buf_u64(out, wt->sigma);
    w += words_serialize(wt->bv, wt->bv_size, 1, out);
    if (wt->size == 0)
    { /* default-constructed supports: rank_support_v5(nullptr) has an EMPTY int_vector<64>,
         select_support_mcl(nullptr) writes arg_cnt = 0 */
        buf_u64(out, (uint64_t)64 << 56);
        w += 8;
        if (select_is_mcl)
        {
            buf_u64(out, 0);
            buf_u64(out, 0);
            w += 16;
        }
    }
    else
    {
        w += orc_rank_v5_serialize(wt->bv_rank, out);
        if (select_is_mcl)
        {
            w += orc_select_mcl_serialize(wt->sel1, out);
            w += orc_select_mcl_serialize(wt->sel0, out);
        }
    }
    w += wt_tree_serialize(wt, out);
    return w;
}

/* ======================================================================================== */
/* csa_wt: alphabet, BWT, backward search, count                                            */
/* ======================================================================================== */

struct orc_csa
{
    uint64_t size; /* text length + 1 */
    uint8_t * bwt;
    orc_wt * wt;
    uint8_t char2comp[256], comp2char[256];
    uint64_t C[257];
    uint16_t sigma;
    /* samples of a csa built from text: SA[k*sa_dens] (sa_order_sa_sampling, csa_sampling_strategy.hpp:97-114) and
     * ISA[k*isa_dens] (isa_sampling, :755-777); csa_wt's defaults are 32 and 64 (csa_wt.hpp:56-64) */
    uint64_t sa_dens, isa_dens, n_sa_s, n_isa_s;
    uint64_t *sa_s, *isa_s;
};

/* byte_alphabet(text_buf, len) csa_alphabet_strategy.hpp:175-212; the symbol histogram of the BWT
 * equals the one of the text it permutes */
static void csa_alphabet(orc_csa * c, const uint8_t * seq, uint64_t len)
{
    uint64_t cnt[257];
    memset(cnt, 0, sizeof cnt);
    memset(c->char2comp, 0, 256);
    memset(c->comp2char, 0, 256);
    for (uint64_t i = 0; i < len; ++i)
        ++cnt[seq[i]];
    c->sigma = 0;
    for (int i = 0; i < 256; ++i)
        if (cnt[i])
        {
            c->char2comp[i] = (uint8_t)c->sigma;
            c->comp2char[c->sigma] = (uint8_t)i;
            cnt[c->sigma] = cnt[i];
            ++c->sigma;
        }
    memset(c->C, 0, sizeof c->C);
    for (int i = c->sigma; i > 0; --i)
        c->C[i] = cnt[i - 1];
    c->C[0] = 0;
    for (int i = 1; i <= c->sigma; ++i)
        c->C[i] += c->C[i - 1];
}

orc_csa * orc_csa_build_from_bwt(const uint8_t * bwt, uint64_t n)
{
    orc_csa * c = (orc_csa *)calloc(1, sizeof *c);
    c->size = n;
    c->bwt = (uint8_t *)malloc(n ? n : 1);
    memcpy(c->bwt, bwt, n);
    csa_alphabet(c, bwt, n);
    c->wt = orc_wt_build(bwt, n); /* csa_wt.hpp:337-343: wavelet tree over the BWT */
    return c;
}

/* suffix array of s[0..n) by prefix doubling with qsort (O(n log^2 n)); s must end with a unique
 * smallest symbol.  The reference uses divsufsort (construct_sa.hpp:120-153); any correct suffix
 * sorter yields the same SA, so the restatement anchors on the definition. */
static const uint64_t * g_rk;
static uint64_t g_n, g_k;
static int sa_cmp(const void * a, const void * b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    if (g_rk[x] != g_rk[y])
        return g_rk[x] < g_rk[y] ? -1 : 1;
    uint64_t rx = x + g_k < g_n ? g_rk[x + g_k] + 1 : 0;
    uint64_t ry = y + g_k < g_n ? g_rk[y + g_k] + 1 : 0;
    return rx < ry ? -1 : (rx > ry ? 1 : 0);
}
static uint64_t * build_sa(const uint8_t * s, uint64_t n)
{
    uint64_t * sa = (uint64_t *)malloc(n * 8);
    uint64_t * rk = (uint64_t *)malloc(n * 8);
    uint64_t * tmp = (uint64_t *)malloc(n * 8);
    for (uint64_t i = 0; i < n; ++i)
    {
        sa[i] = i;
        rk[i] = s[i];
    }
    for (uint64_t k = 1;; k <<= 1)
    {
        g_rk = rk;
        g_n = n;
        g_k = k;
        qsort(sa, n, 8, sa_cmp);
        tmp[sa[0]] = 0;
        for (uint64_t i = 1; i < n; ++i)
            tmp[sa[i]] = tmp[sa[i - 1]] + (sa_cmp(&sa[i - 1], &sa[i]) < 0);
        memcpy(rk, tmp, n * 8);
        if (rk[sa[n - 1]] == n - 1)
            break;
    }
    free(rk);
    free(tmp);
    return sa;
}

orc_csa * orc_csa_build(const uint8_t * text, uint64_t n_text)
{
    return orc_csa_build_ex(text, n_text, 32, 64);
}

orc_csa * orc_csa_build_ex(const uint8_t * text, uint64_t n_text, uint64_t sa_dens, uint64_t isa_dens)
{
    uint64_t n = n_text + 1; /* construct.hpp:100-108 appends the 0 sentinel */
    uint8_t * s = (uint8_t *)malloc(n);
    memcpy(s, text, n_text);
    s[n_text] = 0;
    uint64_t * sa = build_sa(s, n);
    uint8_t * bwt = (uint8_t *)malloc(n);
    for (uint64_t i = 0; i < n; ++i) /* construct_bwt.hpp:59-77: bwt[i] = text[sa[i]-1], wrapping */
        bwt[i] = s[(sa[i] + n - 1) % n];
    orc_csa * c = orc_csa_build_from_bwt(bwt, n);
    c->sa_dens = sa_dens;
    c->isa_dens = isa_dens;
    c->n_sa_s = (n + sa_dens - 1) / sa_dens;
    c->n_isa_s = (n + isa_dens - 1) / isa_dens;
    c->sa_s = (uint64_t *)calloc(c->n_sa_s ? c->n_sa_s : 1, 8);
    c->isa_s = (uint64_t *)calloc(c->n_isa_s ? c->n_isa_s : 1, 8);
    for (uint64_t i = 0; i < n; ++i)
    {
        if (i % sa_dens == 0)
            c->sa_s[i / sa_dens] = sa[i];
        if (sa[i] % isa_dens == 0)
            c->isa_s[sa[i] / isa_dens] = i;
    }
    free(bwt);
    free(sa);
    free(s);
    return c;
}

/* csa.lf[i] (suffix_array_helper.hpp:346-360) */
uint64_t orc_csa_lf(const orc_csa * c, uint64_t i)
{
    uint8_t ch;
    uint64_t j = orc_wt_inverse_select(c->wt, i, &ch);
    return c->C[c->char2comp[ch]] + j;
}

/* first_row_symbol (suffix_array_helper.hpp:28-48) and csa.psi[i] (:330-342) */
static uint8_t csa_first_row_symbol(const orc_csa * c, uint64_t i)
{
    unsigned cc = 0;
    while (cc + 1 < c->sigma && c->C[cc + 1] <= i)
        ++cc;
    return c->comp2char[cc];
}
uint64_t orc_csa_psi(const orc_csa * c, uint64_t i)
{
    uint8_t ch = csa_first_row_symbol(c, i);
    return orc_wt_select(c->wt, i - c->C[c->char2comp[ch]] + 1, ch);
}

/* csa_wt::operator[] (csa_wt.hpp:363-381) */
uint64_t orc_csa_sa(const orc_csa * c, uint64_t i)
{
    uint64_t off = 0;
    while (i % c->sa_dens != 0)
    {
        i = orc_csa_lf(c, i);
        ++off;
    }
    uint64_t result = c->sa_s[i / c->sa_dens];
    return result + off < c->size ? result + off : result + off - c->size;
}

/* csa.isa[i] (suffix_array_helper.hpp:519-537, sample_qeq csa_sampling_strategy.hpp:795-799) */
uint64_t orc_csa_isa(const orc_csa * c, uint64_t i)
{
    uint64_t ci = (i / c->isa_dens + 1) % c->n_isa_s;
    uint64_t result = c->isa_s[ci], pos = ci * c->isa_dens;
    uint64_t steps = pos < i ? pos + c->size - i : pos - i;
    while (steps--)
        result = orc_csa_lf(c, result);
    return result;
}

/* extract(csa, begin, end, text) for LF-based CSAs (suffix_array_algorithm.hpp:578-600); end inclusive */
uint64_t orc_csa_extract(const orc_csa * c, uint64_t begin, uint64_t end, uint8_t * text)
{
    uint64_t steps = end - begin + 1;
    uint64_t order = orc_csa_isa(c, end);
    text[--steps] = csa_first_row_symbol(c, order);
    while (steps != 0)
    {
        uint8_t ch;
        uint64_t j = orc_wt_inverse_select(c->wt, order, &ch);
        order = c->C[c->char2comp[ch]] + j;
        text[--steps] = ch;
    }
    return end - begin + 1;
}

/* locate(csa, begin, end) (suffix_array_algorithm.hpp:505-523): occurrences in SA order; returns their number and
 * writes at most cap of them */
uint64_t orc_csa_locate(const orc_csa * c, const uint8_t * pat, uint64_t m, uint64_t * out, uint64_t cap)
{
    uint64_t l, r;
    uint64_t occs = orc_csa_interval(c, pat, m, &l, &r);
    for (uint64_t i = 0; i < occs && i < cap; ++i)
        out[i] = orc_csa_sa(c, l + i);
    return occs;
}

void orc_csa_free(orc_csa * c)
{
    if (!c)
        return;
    orc_wt_free(c->wt);
    free(c->bwt);
    free(c->sa_s);
    free(c->isa_s);
    free(c);
}
uint64_t orc_csa_size(const orc_csa * c)
{
    return c->size;
}
uint64_t orc_csa_sigma(const orc_csa * c)
{
    return c->sigma;
}
const uint8_t * orc_csa_bwt(const orc_csa * c)
{
    return c->bwt;
}
const orc_wt * orc_csa_wt(const orc_csa * c)
{
    return c->wt;
}
void orc_csa_alphabet(const orc_csa * c, uint8_t char2comp[256], uint64_t C[257])
{
    memcpy(char2comp, c->char2comp, 256);
    memcpy(C, c->C, sizeof c->C);
}

/* suffix_array_algorithm.hpp:167-201 */
uint64_t orc_csa_backward_search_char(const orc_csa * csa, uint64_t l, uint64_t r, uint8_t c, uint64_t * l_res,
                                      uint64_t * r_res)
{
    uint64_t cc = csa->char2comp[c];
    if (cc == 0 && c > 0)
    {
        *l_res = 1;
        *r_res = 0;
    }
    else
    {
        uint64_t c_begin = csa->C[cc];
        if (l == 0 && r + 1 == csa->size)
        {
            *l_res = c_begin;
            *r_res = csa->C[cc + 1] - 1;
        }
        else
        {
            *l_res = c_begin + orc_wt_rank(csa->wt, l, c);
            *r_res = c_begin + orc_wt_rank(csa->wt, r + 1, c) - 1;
        }
    }
    return *r_res + 1 - *l_res;
}

/* suffix_array_algorithm.hpp:228-248 */
uint64_t orc_csa_interval(const orc_csa * csa, const uint8_t * pat, uint64_t m, uint64_t * l_res, uint64_t * r_res)
{
    uint64_t l = 0, r = csa->size - 1;
    const uint8_t * it = pat + m;
    while (pat < it && r + 1 - l > 0)
    {
        --it;
        orc_csa_backward_search_char(csa, l, r, *it, &l, &r);
    }
    *l_res = l;
    *r_res = r;
    return r + 1 - l;
}

/* suffix_array_algorithm.hpp:464-471 */
uint64_t orc_csa_count(const orc_csa * csa, const uint8_t * pat, uint64_t m)
{
    if (m > csa->size)
        return 0;
    uint64_t l, r;
    return orc_csa_interval(csa, pat, m, &l, &r);
}

void orc_csa_count_batch(const orc_csa * csa, const uint8_t * pats, uint32_t m, uint64_t n_pat, uint64_t * out)
{
    for (uint64_t p = 0; p < n_pat; ++p)
        out[p] = orc_csa_count(csa, pats + p * (uint64_t)m, m);
}

/* byte_alphabet::serialize csa_alphabet_strategy.hpp:258-268 */
size_t orc_csa_serialize_alphabet(const orc_csa * c, orc_buf * out)
{
    size_t w = 0;
    uint64_t tmp[32];
    memset(tmp, 0, sizeof tmp);
    memcpy(tmp, c->char2comp, 256);
    w += words_serialize(tmp, 256 * 8, 8, out);
    memset(tmp, 0, sizeof tmp);
    memcpy(tmp, c->comp2char, c->sigma);
    w += words_serialize(tmp, (uint64_t)c->sigma * 8, 8, out);
    w += words_serialize(c->C, ((uint64_t)c->sigma + 1) * 64, 64, out);
    buf_put(out, &c->sigma, 2);
    return w + 2;
}
