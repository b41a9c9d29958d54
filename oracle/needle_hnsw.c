/*
 * CPU restatement of the reference's HNSW index -- TEST / BASELINE INFRASTRUCTURE ONLY.
 *
 * This is what the reference's Collection::batch_search runs today (src/collection/batch.rs:32-52: one
 * HnswIndex::search per query, rayon-parallel over the queries, ef_search = 50), restated so that its
 * throughput and recall can be reported beside the exact GPU search.  Nothing in the product imports it.
 *
 * Follows src/indexing/hnsw.rs:
 *   HnswConfig::default          :249-259   m 16, m_max_0 32, ef_construction 200, ef_search 50, ml 1/ln(m)
 *   Layer::add_connection        :390-405   append below 32 entries, else binary-search insert
 *   random_level                 :578-585   geometric: level += 1 while u < ml, capped at 32
 *   insert                       :612-694
 *   search_with_ef_stats         :734-771
 *   search_layer_core            :957-1017
 *   select_neighbors             :1026-1036 closest max_neighbors, no diversity heuristic
 *   delete (mark only)           :1039-1047
 *
 * Parity status: the reference draws levels from rand::thread_rng (unseeded) and orders equal distances
 * with sort_unstable_by, so its graphs are not reproducible even by itself: GRAPH STRUCTURE IS UNPINNED.
 * What the reference's own tests pin, and tests here check, are properties: self-query first at distance
 * ~0 (:1450-1473), recall@10 > 0.8 on 500 x 32 (:1486-1534), empty index (:1476-1483), ef < k and
 * ef = 0 (:1756-1784), deleted rows never returned (:1537-1567).  Levels here come from splitmix64 under a
 * caller-given seed; equal distances are ordered by (distance, position) in select_neighbors and by
 * (distance, id) in the returned list.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NDL_API __attribute__((visibility("default")))

float ndl_compute(int metric, const float *a, const float *b, size_t n); /* needle_oracle.c */

typedef struct { float d; uint32_t id; } hn_item;

/* OrderedFloat total order on the distance (NaN greatest), then the id: the order of the (OrderedFloat<f32>,
 * VectorId) tuples in both heaps (:968-970). */
static inline int item_less(hn_item a, hn_item b) {
    int an = a.d != a.d, bn = b.d != b.d;
    if (an || bn) { if (an != bn) return bn; return a.id < b.id; }
    if (a.d != b.d) return a.d < b.d;
    return a.id < b.id;
}

typedef struct { hn_item *v; size_t n, cap; int max_heap; } hn_heap;

static inline int heap_before(const hn_heap *h, hn_item a, hn_item b) {
    return h->max_heap ? item_less(b, a) : item_less(a, b);
}
static void heap_push(hn_heap *h, hn_item x) {
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 256; h->v = realloc(h->v, h->cap * sizeof(hn_item)); }
    size_t i = h->n++;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!heap_before(h, x, h->v[p])) break;
        h->v[i] = h->v[p]; i = p;
    }
    h->v[i] = x;
}
static hn_item heap_pop(hn_heap *h) {
    hn_item top = h->v[0], x = h->v[--h->n];
    size_t i = 0;
    for (;;) {
        size_t c = 2 * i + 1;
        if (c >= h->n) break;
        if (c + 1 < h->n && heap_before(h, h->v[c + 1], h->v[c])) c++;
        if (!heap_before(h, h->v[c], x)) break;
        h->v[i] = h->v[c]; i = c;
    }
    if (h->n) h->v[i] = x;
    return top;
}

typedef struct {
    uint32_t m, m_max_0, ef_construction, ef_search;
    double ml;
    int metric;
    size_t dim, n_cap;
    const float *vectors; /* borrowed, row-major [n_cap][dim] */
    /* layer 0: fixed blocks of (m_max_0 + 1) ids per node; upper layers: (m + 1) ids per (node, layer >= 1),
     * at up_off[id] + (layer - 1) blocks.  One spare entry: a list exceeds its bound by one before pruning. */
    uint32_t *adj0; uint16_t *cnt0;
    uint32_t *adj_up; uint16_t *cnt_up; size_t up_blocks, up_cap;
    uint64_t *up_off; /* UINT64_MAX: node has no upper layers */
    uint8_t *level, *present, *deleted;
    int64_t entry_point; uint32_t entry_level;
    size_t count;
    uint64_t rng;
    int fast_build; /* the build scores several rows at a time (same bits, see dist_many) */
    float *bb; uint8_t *bb_known; /* fast build, cosine: b.b of a row, summed once (the sum every pair would repeat) */
} ndl_hnsw;

static inline uint64_t hn_splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint32_t max_conn(const ndl_hnsw *h, uint32_t layer) { return layer == 0 ? h->m_max_0 : h->m; }

static inline uint32_t *conns(const ndl_hnsw *h, uint32_t id, uint32_t layer, uint16_t **cnt) {
    static const uint16_t zero = 0; /* never written: only lists of present nodes are */
    if (layer == 0) { *cnt = &h->cnt0[id]; return h->adj0 + (size_t)id * (h->m_max_0 + 1); }
    /* Layer::get_connections (:378-382): a node absent from the layer has no connections */
    if (h->up_off[id] == UINT64_MAX || layer > h->level[id]) { *cnt = (uint16_t *)&zero; return NULL; }
    size_t b = h->up_off[id] + (layer - 1);
    *cnt = &h->cnt_up[b];
    return h->adj_up + b * (h->m + 1);
}

NDL_API ndl_hnsw *ndl_hnsw_new(const float *vectors, size_t n_cap, size_t dim, int metric, uint32_t m,
                               uint32_t m_max_0, uint32_t ef_construction, uint32_t ef_search, uint64_t seed) {
    if (m == 0 || m_max_0 < m || ef_construction == 0 || ef_search == 0 || m_max_0 > 60000) return NULL; /* validate, :326-357 */
    ndl_hnsw *h = calloc(1, sizeof *h);
    h->m = m; h->m_max_0 = m_max_0; h->ef_construction = ef_construction; h->ef_search = ef_search;
    h->ml = 1.0 / log((double)m);
    h->metric = metric; h->dim = dim; h->n_cap = n_cap; h->vectors = vectors;
    h->adj0 = malloc((n_cap ? n_cap : 1) * (size_t)(m_max_0 + 1) * sizeof(uint32_t));
    h->cnt0 = calloc(n_cap ? n_cap : 1, sizeof(uint16_t));
    h->up_off = malloc((n_cap ? n_cap : 1) * sizeof(uint64_t));
    for (size_t i = 0; i < n_cap; i++) h->up_off[i] = UINT64_MAX;
    h->level = calloc(n_cap ? n_cap : 1, 1);
    h->present = calloc(n_cap ? n_cap : 1, 1);
    h->deleted = calloc(n_cap ? n_cap : 1, 1);
    h->entry_point = -1;
    h->rng = seed;
    h->fast_build = 1;
    h->bb = malloc((n_cap ? n_cap : 1) * sizeof(float));
    h->bb_known = calloc(n_cap ? n_cap : 1, 1);
    return h;
}

NDL_API void ndl_hnsw_free(ndl_hnsw *h) {
    if (!h) return;
    free(h->adj0); free(h->cnt0); free(h->adj_up); free(h->cnt_up); free(h->up_off);
    free(h->level); free(h->present); free(h->deleted); free(h->bb); free(h->bb_known); free(h);
}

/* The BUILD's distances, several rows at a time.  Every row's sums are the sequences ndl_compute runs -- the same
 * operations in the same order per row, so the same bits -- but the chains of eight rows are interleaved, which fills
 * the adder's latency instead of waiting it out once per element.  Of the cosine's three sums, q.q is one and the same
 * for every row (the caller passes it in) and b.b is the same whenever row b is scored (kept after its first use).  The timed searches do not come through here:
 * they pay what the reference pays (one DistanceFunction::compute per neighbour). */
float ndl_dot(const float *a, const float *b, size_t n); /* needle_oracle.c */
enum { HN_COSINE = 0, HN_COSINE_NORMALIZED = 1, HN_EUCLIDEAN = 2, HN_DOT_PRODUCT = 3 };

static void dist_many(const ndl_hnsw *h, const float *q, float qq, const uint32_t *ids, size_t n, float *out) {
    const size_t dim = h->dim;
    if (h->metric == HN_COSINE) {
        const float na = sqrtf(qq);
        for (size_t j = 0; j < n; j++)
            if (!h->bb_known[ids[j]]) {  /* (only the single-threaded build comes here) */
                const float *r = h->vectors + (size_t)ids[j] * dim;
                h->bb[ids[j]] = ndl_dot(r, r, dim);
                h->bb_known[ids[j]] = 1;
            }
        for (size_t g = 0; g < n; g += 8) {
            const float *r[8];
            float s[8];
            for (int j = 0; j < 8; j++) {
                r[j] = h->vectors + (size_t)ids[g + (size_t)j < n ? g + (size_t)j : n - 1] * dim;
                s[j] = -0.0f;
            }
            for (size_t i = 0; i < dim; i++) {
                const float x = q[i];
                for (int j = 0; j < 8; j++) { const float p = x * r[j][i]; s[j] = s[j] + p; }
            }
            for (int j = 0; j < 8 && g + (size_t)j < n; j++) {
                const float nb = sqrtf(h->bb[ids[g + (size_t)j]]);
                out[g + (size_t)j] = (na == 0.0f || nb == 0.0f) ? 1.0f : 1.0f - (s[j] / (na * nb));
            }
        }
        return;
    }
    if (h->metric == HN_COSINE_NORMALIZED || h->metric == HN_EUCLIDEAN || h->metric == HN_DOT_PRODUCT) {
        const int l2 = h->metric == HN_EUCLIDEAN;
        for (size_t g = 0; g < n; g += 8) {
            const float *r[8];
            float s[8];
            for (int j = 0; j < 8; j++) {
                r[j] = h->vectors + (size_t)ids[g + (size_t)j < n ? g + (size_t)j : n - 1] * dim;
                s[j] = -0.0f;
            }
            if (l2) {
                for (size_t i = 0; i < dim; i++) {
                    const float x = q[i];
                    for (int j = 0; j < 8; j++) { const float d = x - r[j][i]; const float p = d * d; s[j] = s[j] + p; }
                }
            } else {
                for (size_t i = 0; i < dim; i++) {
                    const float x = q[i];
                    for (int j = 0; j < 8; j++) { const float p = x * r[j][i]; s[j] = s[j] + p; }
                }
            }
            for (int j = 0; j < 8 && g + (size_t)j < n; j++)
                out[g + (size_t)j] = l2 ? sqrtf(s[j]) : (h->metric == HN_DOT_PRODUCT ? -s[j] : 1.0f - s[j]);
        }
        return;
    }
    for (size_t j = 0; j < n; j++) out[j] = ndl_compute(h->metric, q, h->vectors + (size_t)ids[j] * dim, dim);
}

/* search_layer_core, :957-1017.  `visited` is a zeroed n_cap-byte array, as the reference allocates one per call
 * (:966); the caller provides it zeroed and gets it back zeroed (the touched entries are cleared from a log), which
 * costs less than the reference's fresh allocation -- the timing baseline passes NULL to get the reference's cost.
 * Results: closest first; returns the count, at most ef (at least the entry when it is not deleted). */
static size_t search_layer(const ndl_hnsw *h, const float *query, uint32_t entry, size_t ef, uint32_t layer,
                           uint8_t *visited_in, hn_item **out, size_t *out_cap, size_t *visited_count, int fast) {
    uint8_t *visited = visited_in ? visited_in : calloc(h->n_cap, 1);
    /* fast (the build): the unvisited neighbours of a candidate are scored together; they are marked, and afterwards
     * offered to the heaps, in list order, against the `worst` read before -- what the loop below does one by one */
    uint32_t *fresh = fast ? malloc(((size_t)h->m_max_0 + 1) * sizeof(uint32_t)) : NULL;
    float *fresh_d = fast ? malloc(((size_t)h->m_max_0 + 1) * sizeof(float)) : NULL;
    const float qq = (fast && h->metric == HN_COSINE) ? ndl_dot(query, query, h->dim) : 0.0f;
    uint32_t *log = NULL; size_t nlog = 0, caplog = 0;
    hn_heap cand = {0}, res = {0};
    res.max_heap = 1;
#define MARK(x) do { visited[x] = 1; if (visited_in) { if (nlog == caplog) { caplog = caplog ? caplog * 2 : 1024; \
        log = realloc(log, caplog * sizeof(uint32_t)); } log[nlog++] = (x); } } while (0)
    float entry_dist = ndl_compute(h->metric, query, h->vectors + (size_t)entry * h->dim, h->dim);
    heap_push(&cand, (hn_item){entry_dist, entry});
    if (!h->deleted[entry]) heap_push(&res, (hn_item){entry_dist, entry});
    MARK(entry);
    size_t nvis = 1;
    while (cand.n) {
        hn_item c = heap_pop(&cand);
        float worst = res.n ? res.v[0].d : INFINITY;
        if (c.d > worst && res.n >= ef) break;
        uint16_t *cnt; const uint32_t *nb = conns(h, c.id, layer, &cnt);
        uint16_t nnb = *cnt;
        size_t nfresh = 0;
        if (fast) {
            for (uint16_t j = 0; j < nnb; j++) {
                uint32_t x = nb[j];
                if (visited[x]) continue;
                MARK(x); nvis++;
                fresh[nfresh++] = x;
            }
            dist_many(h, query, qq, fresh, nfresh, fresh_d);
        }
        for (size_t j = 0; j < (fast ? nfresh : (size_t)nnb); j++) {
            uint32_t x = fast ? fresh[j] : nb[j];
            float d;
            if (fast) {
                d = fresh_d[j];
            } else {
                if (visited[x]) continue;
                MARK(x); nvis++;
                d = ndl_compute(h->metric, query, h->vectors + (size_t)x * h->dim, h->dim);
            }
            /* worst_dist is read once per popped candidate (:982), not per neighbour */
            if (d < worst || res.n < ef) {
                heap_push(&cand, (hn_item){d, x});
                if (!h->deleted[x]) {
                    heap_push(&res, (hn_item){d, x});
                    if (res.n > ef) heap_pop(&res);
                }
            }
        }
    }
#undef MARK
    /* into_iter + sort_unstable_by distance (:1013-1015): popping the max-heap and filling from the back gives
     * ascending (distance, id) */
    size_t n = res.n;
    if (*out_cap < n) { *out_cap = n; *out = realloc(*out, (n ? n : 1) * sizeof(hn_item)); }
    for (size_t i = n; i-- > 0;) (*out)[i] = heap_pop(&res);
    if (visited_in) { for (size_t i = 0; i < nlog; i++) visited[log[i]] = 0; free(log); }
    else free(visited);
    free(cand.v); free(res.v); free(fresh); free(fresh_d);
    if (visited_count) *visited_count += nvis;
    return n;
}

/* select_neighbors, :1026-1036: sort by distance, truncate.  Ties (unspecified under sort_unstable_by) keep their
 * input order here.  Sorts `c` in place (insertion merge sort by distance, stable) and returns the kept count. */
static int cmp_stable(const void *a, const void *b) {
    const hn_item *x = a, *y = b; /* id field carries the input position during the sort */
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}
static size_t select_neighbors(hn_item *c, size_t n, size_t max_neighbors, hn_item *tmp) {
    for (size_t i = 0; i < n; i++) { tmp[i].d = c[i].d; tmp[i].id = (uint32_t)i; }
    qsort(tmp, n, sizeof(hn_item), cmp_stable); /* NaN distances do not occur: vectors are validated finite */
    size_t keep = n < max_neighbors ? n : max_neighbors;
    /* gather through the positions; c is overwritten, so go through a second buffer half */
    hn_item *g = tmp + n;
    for (size_t i = 0; i < keep; i++) g[i] = c[tmp[i].id];
    memcpy(c, g, keep * sizeof(hn_item));
    return keep;
}

/* Layer::add_connection, :390-405 (with slice::binary_search's probe sequence, which decides where an id lands in a
 * list of 32 or more entries that is not sorted). */
static void add_connection(uint32_t *list, uint16_t *cnt, uint32_t to) {
    uint16_t n = *cnt;
    if (n < 32) {
        for (uint16_t i = 0; i < n; i++) if (list[i] == to) return;
        list[n] = to; *cnt = n + 1;
        return;
    }
    size_t size = n, base = 0;
    while (size > 1) {
        size_t half = size / 2, mid = base + half;
        if (!(list[mid] > to)) base = mid;
        size -= half;
    }
    if (list[base] == to) return;
    size_t pos = base + (list[base] < to);
    memmove(list + pos + 1, list + pos, (n - pos) * sizeof(uint32_t));
    list[pos] = to; *cnt = n + 1;
}

/* insert, :612-694.  Ids are row numbers of the vectors array, each inserted once.  Returns 0, or -1 for an id out of
 * range. */
NDL_API int ndl_hnsw_insert(ndl_hnsw *h, uint32_t id) {
    if (id >= h->n_cap) return -1;
    uint32_t level = 0;
    while ((double)(hn_splitmix64(&h->rng) >> 11) * (1.0 / 9007199254740992.0) < h->ml && level < 32) level++;
    h->level[id] = (uint8_t)level;
    h->present[id] = 1;
    if (level > 0) {
        if (h->up_blocks + level > h->up_cap) {
            h->up_cap = (h->up_cap ? h->up_cap * 2 : 4096) + level;
            h->adj_up = realloc(h->adj_up, h->up_cap * (size_t)(h->m + 1) * sizeof(uint32_t));
            h->cnt_up = realloc(h->cnt_up, h->up_cap * sizeof(uint16_t));
        }
        h->up_off[id] = h->up_blocks;
        memset(h->cnt_up + h->up_blocks, 0, level * sizeof(uint16_t));
        h->up_blocks += level;
    }
    if (h->entry_point < 0) {
        h->entry_point = id; h->entry_level = level; h->count++;
        return 0;
    }
    const float *vec = h->vectors + (size_t)id * h->dim;
    uint32_t current = (uint32_t)h->entry_point;
    hn_item *cand = NULL; size_t cand_cap = 0;
    uint8_t *visited = calloc(h->n_cap, 1);
    for (uint32_t l = h->entry_level; l > level; l--) {
        size_t n = search_layer(h, vec, current, 1, l, visited, &cand, &cand_cap, NULL, h->fast_build);
        if (n) current = cand[0].id;
    }
    uint32_t start = level < h->entry_level ? level : h->entry_level;
    size_t tmp_cap = 2 * ((size_t)h->ef_construction + h->m_max_0 + 2);
    hn_item *tmp = malloc(tmp_cap * sizeof(hn_item)), *scored = malloc((h->m_max_0 + 2) * sizeof(hn_item));
    hn_item *nbrs = malloc(((size_t)h->ef_construction + 1) * sizeof(hn_item));
    float *prune_d = malloc(((size_t)h->m_max_0 + 2) * sizeof(float));
    for (uint32_t l = start + 1; l-- > 0;) {
        size_t nc = search_layer(h, vec, current, h->ef_construction, l, visited, &cand, &cand_cap, NULL, h->fast_build);
        uint32_t first = nc ? cand[0].id : current;
        memcpy(nbrs, cand, nc * sizeof(hn_item));
        size_t nn = select_neighbors(nbrs, nc, max_conn(h, l), tmp);
        uint16_t *cnt; uint32_t *mine = conns(h, id, l, &cnt);
        for (size_t i = 0; i < nn; i++) mine[i] = nbrs[i].id;
        *cnt = (uint16_t)nn;
        for (size_t i = 0; i < nn; i++) {
            uint32_t nb = nbrs[i].id;
            uint16_t *ncnt; uint32_t *nl = conns(h, nb, l, &ncnt);
            if (!nl) continue; /* cannot happen: a result of layer l is a node of layer l */
            add_connection(nl, ncnt, id);
            uint32_t mc = max_conn(h, l);
            if (*ncnt > mc) {
                const float *nv = h->vectors + (size_t)nb * h->dim;
                size_t ns = *ncnt;
                if (h->fast_build) {
                    dist_many(h, nv, h->metric == HN_COSINE ? ndl_dot(nv, nv, h->dim) : 0.0f, nl, ns, prune_d);
                    for (size_t j = 0; j < ns; j++) { scored[j].id = nl[j]; scored[j].d = prune_d[j]; }
                } else {
                    for (size_t j = 0; j < ns; j++) {
                        scored[j].id = nl[j];
                        scored[j].d = ndl_compute(h->metric, nv, h->vectors + (size_t)nl[j] * h->dim, h->dim);
                    }
                }
                size_t kept = select_neighbors(scored, ns, mc, tmp);
                for (size_t j = 0; j < kept; j++) nl[j] = scored[j].id;
                *ncnt = (uint16_t)kept;
            }
        }
        if (nc) current = first;
    }
    free(tmp); free(scored); free(nbrs); free(prune_d); free(cand); free(visited);
    if (level > h->entry_level) { h->entry_point = id; h->entry_level = level; }
    h->count++;
    return 0;
}

NDL_API void ndl_hnsw_insert_range(ndl_hnsw *h, uint32_t first, uint32_t count) {
    for (uint32_t i = 0; i < count; i++) ndl_hnsw_insert(h, first + i);
}

NDL_API int ndl_hnsw_delete(ndl_hnsw *h, uint32_t id) { /* :1039-1047: mark only */
    if (id >= h->n_cap || !h->present[id] || h->deleted[id]) return 0;
    h->deleted[id] = 1;
    return 1;
}

NDL_API size_t ndl_hnsw_len(const ndl_hnsw *h) { return h->count; }
NDL_API void ndl_hnsw_set_fast_build(ndl_hnsw *h, int on) { h->fast_build = on; } /* 0: the build scores row by row too */
NDL_API void ndl_hnsw_set_ef_search(ndl_hnsw *h, uint32_t ef) { h->ef_search = ef; } /* :566-568 */
NDL_API uint32_t ndl_hnsw_entry_level(const ndl_hnsw *h) { return h->entry_level; }
NDL_API size_t ndl_hnsw_degree(const ndl_hnsw *h, uint32_t id, uint32_t layer) {
    uint16_t *cnt; conns(h, id, layer, &cnt); return *cnt;
}

/* search_with_ef_stats, :734-771.  Returns the number of results written (at most k). */
NDL_API size_t ndl_hnsw_search(const ndl_hnsw *h, const float *query, size_t k, size_t ef, uint64_t *out_ids,
                               float *out_dists, size_t *visited_nodes) {
    if (h->entry_point < 0) return 0;
    uint32_t current = (uint32_t)h->entry_point;
    hn_item *cand = NULL; size_t cap = 0, vis = 0;
    for (uint32_t l = h->entry_level; l >= 1; l--) {
        size_t n = search_layer(h, query, current, 1, l, NULL, &cand, &cap, &vis, 0);
        if (n) current = cand[0].id;
    }
    size_t n = search_layer(h, query, current, ef, 0, NULL, &cand, &cap, &vis, 0);
    if (n > k) n = k;
    for (size_t i = 0; i < n; i++) { out_ids[i] = cand[i].id; out_dists[i] = cand[i].d; }
    free(cand);
    if (visited_nodes) *visited_nodes = vis;
    return n;
}

/* Collection::batch_search, src/collection/batch.rs:32-52: one search per query at the configured ef_search,
 * parallel over the queries.  out_ids / out_dists are [nq][k]; out_counts [nq]. */
typedef struct { const ndl_hnsw *h; const float *q; size_t nq, k, first, stride; uint64_t *ids; float *dists; uint32_t *counts; } hn_job;
static void *hn_worker(void *p) {
    hn_job *j = p;
    for (size_t i = j->first; i < j->nq; i += j->stride)
        j->counts[i] = (uint32_t)ndl_hnsw_search(j->h, j->q + i * j->h->dim, j->k, j->h->ef_search,
                                                 j->ids + i * j->k, j->dists + i * j->k, NULL);
    return NULL;
}
NDL_API void ndl_hnsw_batch_search(const ndl_hnsw *h, const float *queries, size_t nq, size_t k, uint64_t *out_ids,
                                   float *out_dists, uint32_t *out_counts, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256]; hn_job jobs[256];
    for (int t = 0; t < threads; t++) {
        jobs[t] = (hn_job){h, queries, nq, k, (size_t)t, (size_t)threads, out_ids, out_dists, out_counts};
        pthread_create(&th[t], NULL, hn_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
}
