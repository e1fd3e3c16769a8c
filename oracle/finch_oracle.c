/*
 * finch_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see finch_oracle.h).
 *
 * Plain-C restatement of the reference algorithm; every function cites the reference
 * file:line (relative to /root/reference) or the third-party crate whose published
 * algorithm it restates.  Data structures deliberately mirror the reference (binary
 * max-heap of owned k-mer byte vectors + hash map of counts, per-record normalize /
 * reverse-complement allocations) so that it doubles as the CPU baseline ("port").
 */
#include "finch_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * murmurhash3 0.0.5 (Cargo.lock:471-474): murmurhash3_x64_128(bytes, seed: u64) -> (u64, u64)
 * == Austin Appleby's MurmurHash3_x64_128 with h1 = h2 = seed (64-bit seed), little-endian
 * block reads.  Call site: lib/src/sketch_schemes/hashing.rs:10-12 (keeps .0 == h1).
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

static inline uint64_t load_le64(const uint8_t *p) {
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | p[i];
    return v;
}

void fo_murmur3_x64_128(const uint8_t *data, size_t len, uint64_t seed, uint64_t out[2]) {
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
    uint64_t h1 = seed, h2 = seed;
    const size_t nblocks = len / 16;
    for (size_t i = 0; i < nblocks; ++i) {
        uint64_t k1 = load_le64(data + 16 * i);
        uint64_t k2 = load_le64(data + 16 * i + 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint8_t *tail = data + nblocks * 16;
    uint64_t k1 = 0, k2 = 0;
    switch (len & 15) {
    case 15: k2 ^= (uint64_t)tail[14] << 48; /* fallthrough */
    case 14: k2 ^= (uint64_t)tail[13] << 40; /* fallthrough */
    case 13: k2 ^= (uint64_t)tail[12] << 32; /* fallthrough */
    case 12: k2 ^= (uint64_t)tail[11] << 24; /* fallthrough */
    case 11: k2 ^= (uint64_t)tail[10] << 16; /* fallthrough */
    case 10: k2 ^= (uint64_t)tail[9] << 8;   /* fallthrough */
    case 9:  k2 ^= (uint64_t)tail[8];
             k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; /* fallthrough */
    case 8:  k1 ^= (uint64_t)tail[7] << 56; /* fallthrough */
    case 7:  k1 ^= (uint64_t)tail[6] << 48; /* fallthrough */
    case 6:  k1 ^= (uint64_t)tail[5] << 40; /* fallthrough */
    case 5:  k1 ^= (uint64_t)tail[4] << 32; /* fallthrough */
    case 4:  k1 ^= (uint64_t)tail[3] << 24; /* fallthrough */
    case 3:  k1 ^= (uint64_t)tail[2] << 16; /* fallthrough */
    case 2:  k1 ^= (uint64_t)tail[1] << 8;  /* fallthrough */
    case 1:  k1 ^= (uint64_t)tail[0];
             k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
    }
    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}

/* lib/src/sketch_schemes/hashing.rs:10-12 */
uint64_t fo_hash_f(const uint8_t *item, size_t len, uint64_t seed) {
    uint64_t h[2];
    fo_murmur3_x64_128(item, len, seed, h);
    return h[0];
}

/* ------------------------------------------------------------------------------------------
 * needletail 0.5.0 (Cargo.lock:490-494) sequence helpers; call sites mash.rs:72-76.
 * normalize(iupac=false):  ACGT kept; acg -> upper; t,u,U -> T; '.', '-', '~' -> '-';
 * space, tab, CR, LF dropped; everything else (N, IUPAC codes, ...) -> 'N'.
 * ---------------------------------------------------------------------------------------- */
size_t fo_normalize(const uint8_t *in, size_t n, uint8_t *out) {
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        uint8_t c = in[i], o;
        switch (c) {
        case 'A': case 'C': case 'G': case 'T': o = c; break;
        case 'a': o = 'A'; break;
        case 'c': o = 'C'; break;
        case 'g': o = 'G'; break;
        case 't': case 'u': case 'U': o = 'T'; break;
        case '-': case '.': case '~': o = '-'; break;
        case ' ': case '\t': case '\r': case '\n': continue; /* removed */
        default: o = 'N'; break;
        }
        out[m++] = o;
    }
    return m;
}

/* needletail complement(): A<->T, C<->G (case kept), IUPAC pairs, everything else unchanged.
 * Only the ACGT rows matter here: after normalize(false) the alphabet is {A,C,G,T,N,-}. */
static inline uint8_t complement(uint8_t c) {
    switch (c) {
    case 'A': return 'T'; case 'T': return 'A'; case 'C': return 'G'; case 'G': return 'C';
    case 'a': return 't'; case 't': return 'a'; case 'c': return 'g'; case 'g': return 'c';
    default: return c;
    }
}

void fo_reverse_complement(const uint8_t *in, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) out[i] = complement(in[n - 1 - i]);
}

static inline int is_good_base(uint8_t c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

/* ------------------------------------------------------------------------------------------
 * Sketcher state: BinaryHeap<HashedItem<Vec<u8>>> + HashMap<ItemHash,(u32,u32)>
 * (mash.rs:10-18, scaled.rs:10-19, hashing.rs:15-38: ordering by hash only).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t hash;
    uint8_t *item; /* owned copy of the k-mer bytes (kmer.to_owned()) */
} heap_item;

typedef struct {
    uint64_t key;
    uint32_t count, extra;
    uint8_t used;
} map_slot;

struct fo_sketcher {
    int kind;
    heap_item *heap; size_t heap_len, heap_cap;
    map_slot *map; size_t map_cap, map_len; /* open addressing, linear probing, pow2 */
    uint8_t k;
    uint64_t total_kmers, total_bases;
    size_t size;
    uint64_t max_hash; /* scaled only */
    uint64_t seed;
    uint64_t hash_mask; /* test hook */
    uint8_t *norm_buf, *rc_buf; size_t buf_cap;
};

/* --- binary max-heap (std::collections::BinaryHeap semantics) --- */
static void heap_push(fo_sketcher *s, uint64_t hash, const uint8_t *kmer, size_t len) {
    if (s->heap_len == s->heap_cap) {
        s->heap_cap = s->heap_cap ? s->heap_cap * 2 : 16;
        s->heap = (heap_item *)realloc(s->heap, s->heap_cap * sizeof(heap_item));
    }
    heap_item it;
    it.hash = hash;
    it.item = (uint8_t *)malloc(len ? len : 1);
    memcpy(it.item, kmer, len);
    size_t i = s->heap_len++;
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (s->heap[p].hash >= it.hash) break;
        s->heap[i] = s->heap[p];
        i = p;
    }
    s->heap[i] = it;
}

static heap_item heap_pop(fo_sketcher *s) {
    heap_item top = s->heap[0];
    heap_item last = s->heap[--s->heap_len];
    size_t n = s->heap_len, i = 0;
    if (n > 0) {
        for (;;) {
            size_t c = 2 * i + 1;
            if (c >= n) break;
            if (c + 1 < n && s->heap[c + 1].hash > s->heap[c].hash) c++;
            if (s->heap[c].hash <= last.hash) break;
            s->heap[i] = s->heap[c];
            i = c;
        }
        s->heap[i] = last;
    }
    return top;
}

/* --- hash map keyed by the item hash itself (NoHashHasher, hashing.rs:43-64) --- */
static inline size_t map_home(const fo_sketcher *s, uint64_t key) {
    return (size_t)((key * 0x9E3779B97F4A7C15ULL) >> 17) & (s->map_cap - 1);
}

static map_slot *map_find(const fo_sketcher *s, uint64_t key) {
    size_t i = map_home(s, key);
    while (s->map[i].used) {
        if (s->map[i].key == key) return &s->map[i];
        i = (i + 1) & (s->map_cap - 1);
    }
    return NULL;
}

static void map_insert_nogrow(fo_sketcher *s, uint64_t key, uint32_t count, uint32_t extra) {
    size_t i = map_home(s, key);
    while (s->map[i].used) i = (i + 1) & (s->map_cap - 1);
    s->map[i].used = 1; s->map[i].key = key; s->map[i].count = count; s->map[i].extra = extra;
    s->map_len++;
}

static void map_insert(fo_sketcher *s, uint64_t key, uint32_t count, uint32_t extra) {
    if ((s->map_len + 1) * 2 > s->map_cap) {
        map_slot *old = s->map; size_t oc = s->map_cap;
        s->map_cap = oc * 2; s->map_len = 0;
        s->map = (map_slot *)calloc(s->map_cap, sizeof(map_slot));
        for (size_t j = 0; j < oc; ++j)
            if (old[j].used) map_insert_nogrow(s, old[j].key, old[j].count, old[j].extra);
        free(old);
    }
    map_insert_nogrow(s, key, count, extra);
}

static void map_remove(fo_sketcher *s, uint64_t key) {
    size_t mask = s->map_cap - 1, i = map_home(s, key);
    while (s->map[i].used && s->map[i].key != key) i = (i + 1) & mask;
    if (!s->map[i].used) return; /* reference would panic on unwrap(); unreachable */
    /* backward-shift deletion */
    size_t j = i;
    for (;;) {
        j = (j + 1) & mask;
        if (!s->map[j].used) break;
        size_t h = map_home(s, s->map[j].key);
        /* can slot j move to i?  yes iff its home is cyclically outside (i, j] */
        int between = (i <= j) ? (i < h && h <= j) : (i < h || h <= j);
        if (!between) { s->map[i] = s->map[j]; i = j; }
    }
    s->map[i].used = 0;
    s->map_len--;
}

static inline uint32_t sat_add_u32(uint32_t a, uint32_t b) {
    uint32_t r = a + b;
    return r < a ? UINT32_MAX : r;
}

fo_sketcher *fo_new(int kind, size_t size, double scale, uint8_t k, uint64_t seed) {
    fo_sketcher *s = (fo_sketcher *)calloc(1, sizeof(fo_sketcher));
    s->kind = kind; s->k = k; s->size = size; s->seed = seed; s->hash_mask = ~0ULL;
    s->map_cap = 1024; s->map = (map_slot *)calloc(s->map_cap, sizeof(map_slot));
    if (kind == FO_SCALED) {
        /* scaled.rs:23,31: iscale = (1. / scale) as u64 ; max_hash = u64::MAX / iscale
         * (Rust `as u64` saturates; scale > 1 gives iscale 0 and the reference would panic on
         * division by zero -- callers must not do that.) */
        double inv = 1.0 / scale;
        uint64_t iscale;
        if (!(inv == inv) || inv <= 0.0) iscale = 0;
        else if (inv >= 18446744073709551616.0) iscale = UINT64_MAX;
        else iscale = (uint64_t)inv;
        s->max_hash = iscale ? UINT64_MAX / iscale : UINT64_MAX;
    }
    return s;
}

void fo_free(fo_sketcher *s) {
    if (!s) return;
    for (size_t i = 0; i < s->heap_len; ++i) free(s->heap[i].item);
    free(s->heap); free(s->map); free(s->norm_buf); free(s->rc_buf); free(s);
}

void fo_set_hash_mask(fo_sketcher *s, uint64_t mask) { s->hash_mask = mask; }
uint64_t fo_max_hash(const fo_sketcher *s) { return s->max_hash; }

void fo_push(fo_sketcher *s, const uint8_t *kmer, size_t len, uint8_t extra_count) {
    s->total_kmers += 1;                                        /* mash.rs:35 / scaled.rs:38 */
    uint64_t new_hash = fo_hash_f(kmer, len, s->seed) & s->hash_mask;
    int add_hash;
    if (s->kind == FO_MASH) {                                   /* mash.rs:37-42 */
        if (s->heap_len == 0) add_hash = 1;
        else add_hash = (new_hash <= s->heap[0].hash) || (s->heap_len < s->size);
    } else {                                                    /* scaled.rs:41 */
        add_hash = (new_hash <= s->max_hash) || (s->heap_len <= s->size && s->size != 0);
    }
    if (!add_hash) return;
    map_slot *e = map_find(s, new_hash);
    if (e) {                                                    /* mash.rs:45-50 */
        e->count = sat_add_u32(e->count, 1);
        e->extra = sat_add_u32(e->extra, (uint32_t)extra_count);
        return;
    }
    heap_push(s, new_hash, kmer, len);                          /* mash.rs:52-56 */
    map_insert(s, new_hash, 1, (uint32_t)extra_count);
    if (s->kind == FO_MASH) {                                   /* mash.rs:57-60 */
        if (s->heap_len > s->size) {
            heap_item h = heap_pop(s);
            map_remove(s, h.hash);
            free(h.item);
        }
    } else {                                                    /* scaled.rs:54-58 */
        if (s->heap_len > s->size && s->heap[0].hash > s->max_hash) {
            heap_item h = heap_pop(s);
            map_remove(s, h.hash);
            free(h.item);
        }
    }
}

/* needletail CanonicalKmers (kmer.rs): every window of k consecutive good bases of the
 * normalized sequence; yields (pos, fwd, false) if fwd < rc lexicographically else (pos, rc, true). */
void fo_process(fo_sketcher *s, const uint8_t *seq, size_t len) {
    s->total_bases += (uint64_t)len;                            /* mash.rs:72 */
    if (len > s->buf_cap) {
        s->buf_cap = len * 2 + 64;
        s->norm_buf = (uint8_t *)realloc(s->norm_buf, s->buf_cap);
        s->rc_buf = (uint8_t *)realloc(s->rc_buf, s->buf_cap);
    }
    size_t n = fo_normalize(seq, len, s->norm_buf);             /* mash.rs:73 */
    fo_reverse_complement(s->norm_buf, n, s->rc_buf);           /* mash.rs:75 */
    const size_t k = s->k;
    if (k == 0 || n < k) return;
    const uint8_t *buf = s->norm_buf, *rc = s->rc_buf;
    size_t good = 0; /* length of the current run of good bases ending at i */
    for (size_t i = 0; i < n; ++i) {
        good = is_good_base(buf[i]) ? good + 1 : 0;
        if (good >= k) {
            size_t pos = i + 1 - k;
            const uint8_t *fwd = buf + pos;
            const uint8_t *rcw = rc + (n - pos - k);
            if (memcmp(fwd, rcw, k) < 0) fo_push(s, fwd, k, 0); /* mash.rs:76-78 */
            else fo_push(s, rcw, k, 1);
        }
    }
}

void fo_process_packed(fo_sketcher *s, const uint8_t *buf, size_t len, uint8_t sep) {
    size_t start = 0;
    for (size_t i = 0; i <= len; ++i) {
        if (i == len || buf[i] == sep) {
            if (i > start) fo_process(s, buf + start, i - start);
            start = i + 1;
        }
    }
}

void fo_totals(const fo_sketcher *s, uint64_t *total_bases, uint64_t *total_kmers) {
    if (total_bases) *total_bases = s->total_bases;
    if (total_kmers) *total_kmers = s->total_kmers;
}

size_t fo_len(const fo_sketcher *s) { return s->heap_len; }

static int cmp_heap_item(const void *a, const void *b) {
    uint64_t x = ((const heap_item *)a)->hash, y = ((const heap_item *)b)->hash;
    return (x > y) - (x < y);
}

/* mash.rs:86-102 / scaled.rs:84-100: into_sorted_vec (ascending) + counts lookup */
size_t fo_to_vec(const fo_sketcher *s, fo_kmercount *out, uint8_t *kmers) {
    size_t n = s->heap_len;
    heap_item *tmp = (heap_item *)malloc((n ? n : 1) * sizeof(heap_item));
    memcpy(tmp, s->heap, n * sizeof(heap_item));
    qsort(tmp, n, sizeof(heap_item), cmp_heap_item);
    for (size_t i = 0; i < n; ++i) {
        const map_slot *e = map_find(s, tmp[i].hash);
        out[i].hash = tmp[i].hash;
        out[i].count = e ? e->count : 0;
        out[i].extra_count = e ? e->extra : 0;
        if (kmers) memcpy(kmers + i * s->k, tmp[i].item, s->k);
    }
    free(tmp);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * sketch_stream record loop (lib/src/lib.rs:60-68) over needletail 0.5.0's FASTX reader,
 * restated: the first byte of the file picks the format ('>' FASTA, '@' FASTQ).
 *  FASTA: header line, then sequence = bytes up to (not incl.) the newline before the next
 *         line starting with '>' (or EOF); internal newlines are part of sequence() and are
 *         dropped later by normalize.  A trailing CR before that final newline is trimmed.
 *  FASTQ: strict 4-line records (@id / seq / + / qual), seq and qual same length, CR trimmed.
 * ---------------------------------------------------------------------------------------- */
static const uint8_t *find_nl(const uint8_t *p, const uint8_t *end) {
    return (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
}

int fo_sketch_stream(fo_sketcher *s, const uint8_t *buf, size_t len) {
    const uint8_t *p = buf, *end = buf + len;
    if (len == 0) return -1;
    if (*p == '>') {
        while (p < end) {
            if (*p != '>') return -2;
            const uint8_t *nl = find_nl(p, end);
            if (!nl) break; /* header without sequence at EOF */
            const uint8_t *seq = nl + 1;
            /* sequence lines run until the next line that starts with '>' (or EOF) */
            const uint8_t *q = seq;
            while (q < end && *q != '>') {
                const uint8_t *l = find_nl(q, end);
                q = l ? l + 1 : end;
            }
            const uint8_t *se = q;
            if (se > seq && se[-1] == '\n') se--;
            if (se > seq && se[-1] == '\r') se--;
            fo_process(s, seq, (size_t)(se - seq));
            p = q;
        }
        return FO_FMT_FASTA;
    } else if (*p == '@') {
        while (p < end) {
            if (*p == '\n' || *p == '\r') { p++; continue; } /* trailing blank lines */
            if (*p != '@') return -3;
            const uint8_t *l1 = find_nl(p, end); if (!l1) return -4;
            const uint8_t *seq = l1 + 1;
            const uint8_t *l2 = find_nl(seq, end); if (!l2) return -4;
            const uint8_t *plus = l2 + 1;
            if (plus >= end || *plus != '+') return -5;
            const uint8_t *l3 = find_nl(plus, end); if (!l3) return -4;
            const uint8_t *qual = l3 + 1;
            const uint8_t *l4 = find_nl(qual, end);
            const uint8_t *qe = l4 ? l4 : end;
            const uint8_t *se = l2;
            if (se > seq && se[-1] == '\r') se--;
            const uint8_t *qe2 = qe;
            if (qe2 > qual && qe2[-1] == '\r') qe2--;
            if ((se - seq) != (qe2 - qual)) return -6;
            fo_process(s, seq, (size_t)(se - seq));
            p = l4 ? l4 + 1 : end;
        }
        return FO_FMT_FASTQ;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * Host post-processing, restated from lib/src/filtering.rs and lib/src/statistics.rs
 * ---------------------------------------------------------------------------------------- */

/* filtering.rs:413-432 */
size_t fo_filter_strands(const fo_kmercount *in, const uint8_t *kin, size_t n, size_t k, double ratio,
                         fo_kmercount *out, uint8_t *kout) {
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        int keep;
        if (in[i].count < 16) keep = 1;
        else {
            uint32_t other = in[i].count - in[i].extra_count;
            uint32_t lowest = in[i].extra_count < other ? in[i].extra_count : other;
            keep = ((double)lowest / (double)in[i].count) >= ratio;
        }
        if (keep) {
            out[m] = in[i];
            if (kin && kout) memcpy(kout + m * k, kin + i * k, k);
            m++;
        }
    }
    return m;
}

/* statistics.rs:30-47 (hist) + filtering.rs:154-195 (guess_filter_threshold) */
uint32_t fo_guess_filter_threshold(const fo_kmercount *in, size_t n, double filter_level) {
    uint64_t max_count = 0;
    for (size_t i = 0; i < n; ++i)
        if (in[i].count > max_count) max_count = in[i].count;
    uint64_t *hist = (uint64_t *)calloc(max_count ? max_count : 1, sizeof(uint64_t));
    for (size_t i = 0; i < n; ++i) hist[in[i].count - 1] += 1;
    uint64_t total = 0;
    for (uint64_t i = 0; i < max_count; ++i) total += (i + 1) * hist[i];
    double total_counts = (double)total;
    double cutoff_amt = filter_level * total_counts;

    size_t wgt_cutoff = 0;
    uint64_t cum_count = 0;
    for (uint64_t i = 0; i < max_count; ++i) {
        cum_count += (uint64_t)wgt_cutoff * hist[i];
        if ((double)cum_count > cutoff_amt) break;
        wgt_cutoff += 1;
    }
    if (wgt_cutoff == 0) { free(hist); return 1; }

    size_t win_size = wgt_cutoff / 20 > 1 ? wgt_cutoff / 20 : 1;
    uint64_t sum = 0;
    for (size_t i = 0; i < win_size; ++i) sum += hist[i];
    uint64_t lowest_val = sum;
    size_t lowest_idx = win_size - 1;
    for (size_t i = 0, j = win_size; j < wgt_cutoff; ++i, ++j) {
        if (sum <= lowest_val) { lowest_val = sum; lowest_idx = j; }
        sum -= hist[i];
        sum += hist[j];
    }
    free(hist);
    return (uint32_t)lowest_idx + 1;
}

/* filtering.rs:329-343 */
size_t fo_filter_abundance(const fo_kmercount *in, const uint8_t *kin, size_t n, size_t k, int has_lo, uint32_t lo,
                           int has_hi, uint32_t hi, fo_kmercount *out, uint8_t *kout) {
    uint32_t lo_t = has_lo ? lo : 0u, hi_t = has_hi ? hi : UINT32_MAX;
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        if (lo_t <= in[i].count && in[i].count <= hi_t) {
            out[m] = in[i];
            if (kin && kout) memcpy(kout + m * k, kin + i * k, k);
            m++;
        }
    }
    return m;
}
