/*
 * ORACLE -- test infrastructure only.  NOT part of the product.
 *
 * Plain-C, single-threaded restatement on packed words of the hot path of jimpo/quantum-css-codes, used
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg to check libgf2hip.so at sizes
 * where the NumPy restatement (oracle/cpu_ref.py) would take minutes.  Parity status: PINNED by the CPU suite --
 * tests/test_oracle_golden.py::test_c_oracle_{rref,normalize,syndromes,full_size_digests,swap_columns} compare orc_rref,
 * orc_normalize (matrices, swap lists, both error codes), orc_syndrome_batch, orc_histogram and orc_swap_columns with every
 * array the reference itself produced (tests/golden/reference_golden.npz, incl. the three 2048 x 4096 SHA-256 digests);
 * test_c_oracle_{nullspace,sampler,sampler_n4096,monte_carlo} and test_decode_tally_c_vs_numpy compare the build-defined
 * pieces with oracle/cpu_ref.py (multi-word rows with a ragged last word included).
 *
 * Packed layout as in include/gf2hip.h: row-major uint64 words, column j at word j>>6 bit j&63.
 * Nothing here shares code with quantum_css_codes_amd/csrc.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;

static inline int bit(const u64* row, int64_t j) { return (int)((row[j >> 6] >> (j & 63)) & 1u); }
static inline void xor_row(u64* dst, const u64* src, int64_t ld) {
    for (int64_t w = 0; w < ld; ++w) dst[w] ^= src[w];
}

/* bin_matrix.py:8-34.  The reference adds the found row into position r (bin_matrix.py:23-24) and then
 * clears the column in every other row (bin_matrix.py:27-29); on bits, addition is XOR. */
int orc_rref(u64* a, int64_t m, int64_t n, int64_t ld, int64_t* pivots, int64_t* rank_out) {
    int64_t lead = 0;
    for (int64_t c = 0; c < n; ++c) {
        int64_t found = -1;
        for (int64_t i = lead; i < m; ++i)
            if (bit(a + i * ld, c)) { found = i; break; }
        if (found < 0) continue;
        if (!bit(a + lead * ld, c)) xor_row(a + lead * ld, a + found * ld, ld);
        for (int64_t i = 0; i < m; ++i)
            if (i != lead && bit(a + i * ld, c)) xor_row(a + i * ld, a + lead * ld, ld);
        if (pivots) pivots[lead] = c;
        lead += 1;
    }
    *rank_out = lead;
    return 0;
}

static void swap_cols(u64* a, int64_t m, int64_t ld, int64_t i, int64_t j) {
    for (int64_t r = 0; r < m; ++r) {
        u64* row = a + r * ld;
        int bi = bit(row, i), bj = bit(row, j);
        if (bi != bj) {
            row[i >> 6] ^= 1ull << (i & 63);
            row[j >> 6] ^= 1ull << (j & 63);
        }
    }
}

/* css_code.py:783-785 */
int orc_swap_columns(u64* a, int64_t m, int64_t ld, int64_t i, int64_t j) {
    swap_cols(a, m, ld, i, j);
    return 0;
}

/* css_code.py:809-836.  Returns 0, -2 ("not enough columns", :811-812) or -3 ("rows are not
 * independent", :825-826).  swaps receives (i + offset, col) pairs (:828). */
int orc_normalize(u64* h, int64_t r, int64_t n, int64_t ld, int64_t offset, int64_t* swaps, int64_t* nswaps) {
    *nswaps = 0;
    if (n < offset + r) return -2;
    for (int64_t i = 0; i < r; ++i) {
        const int64_t c = i + offset;
        int64_t row = -1;
        for (int64_t j = i; j < r; ++j)
            if (bit(h + j * ld, c)) { row = j; break; }
        if (row >= 0) {
            if (!bit(h + i * ld, c)) xor_row(h + i * ld, h + row * ld, ld);
        } else {
            int64_t col = -1;
            for (int64_t j = c; j < n; ++j)
                if (bit(h + i * ld, j)) { col = j; break; }
            if (col < 0) return -3;
            swaps[2 * (*nswaps)] = c;
            swaps[2 * (*nswaps) + 1] = col;
            *nswaps += 1;
            swap_cols(h, r, ld, c, col);
        }
        for (int64_t j = 0; j < r; ++j)
            if (j != i && bit(h + j * ld, c)) xor_row(h + j * ld, h + i * ld, ld);
    }
    return 0;
}

/* [build-defined x1] nullspace from the RREF, as oracle/cpu_ref.py:nullspace.  out: (n - rank) x ldn. */
int orc_nullspace(const u64* a, int64_t m, int64_t n, int64_t ld, u64* out, int64_t ldn, int64_t* rows_out) {
    u64* red = (u64*)malloc((size_t)(m > 0 ? m : 1) * ld * 8);
    int64_t* piv = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * 8);
    int64_t* piv_row = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * 8);
    int64_t rank = 0;
    memcpy(red, a, (size_t)m * ld * 8);
    orc_rref(red, m, n, ld, piv, &rank);
    for (int64_t c = 0; c < n; ++c) piv_row[c] = -1;
    for (int64_t i = 0; i < rank; ++i) piv_row[piv[i]] = i;
    int64_t t = 0;
    for (int64_t fc = 0; fc < n; ++fc) {
        if (piv_row[fc] >= 0) continue;
        u64* dst = out + t * ldn;
        memset(dst, 0, (size_t)ldn * 8);
        dst[fc >> 6] |= 1ull << (fc & 63);
        for (int64_t i = 0; i < rank; ++i)
            if (bit(red + i * ld, fc)) dst[piv[i] >> 6] |= 1ull << (piv[i] & 63);
        t += 1;
    }
    *rows_out = t;
    free(red);
    free(piv);
    free(piv_row);
    return 0;
}

/* Threads the batch loops below run on (bench.py's cpu_baseline reports it next to the rate). */
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* css_code.py:728 for a batch, sample-major: s[b] bit i = parity(h[i] & e[b]). */
int orc_syndrome_batch(const u64* h, int64_t r, int64_t n, int64_t ldh, const u64* e, int64_t batch, int64_t lde,
                       u64* s, int64_t lds) {
    const int64_t words = (n + 63) >> 6;
#pragma omp parallel for schedule(static) if (batch * r * words > (1 << 22))      /* samples are independent */
    for (int64_t b = 0; b < batch; ++b) {
        u64* out = s + b * lds;
        memset(out, 0, (size_t)lds * 8);
        const u64* err = e + b * lde;
        for (int64_t i = 0; i < r; ++i) {
            const u64* row = h + i * ldh;
            u64 acc = 0;
            for (int64_t w = 0; w < words; ++w) acc ^= row[w] & err[w];
            out[i >> 6] |= (u64)(__builtin_popcountll(acc) & 1) << (i & 63);
        }
    }
    return 0;
}

/* mode 0: key = big-endian integer of the r syndrome bits (bin_matrix.py:36-43, css_code.py:729);
 * mode 1: key = Hamming weight.  hist is accumulated. */
int orc_histogram(const u64* s, int64_t batch, int64_t lds, int64_t r, int mode, u64* hist) {
    for (int64_t b = 0; b < batch; ++b) {
        const u64* row = s + b * lds;
        u64 key = 0;
        if (mode == 0) {
            for (int64_t i = 0; i < r; ++i) key = (key << 1) | (u64)bit(row, i);
        } else {
            for (int64_t i = 0; i < r; ++i) key += (u64)bit(row, i);
        }
        hist[key] += 1;
    }
    return 0;
}

/* ---- [build-defined x3] sampler: DESIGN.md "Sampler", plain evaluation ------------------------------- */

#define GOLDEN 0x9E3779B97F4A7C15ull
#define STREAM_MULT 0xD1B54A32D192ED03ull

static u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static u64 quantise(double x) {
    double t = __builtin_floor(x * 4294967296.0 + 0.5);
    if (!(t > 0.0)) return 0;
    if (t >= 4294967296.0) return 4294967296ull;
    return (u64)t;
}

/* Inverse binomial CDF table: cdf[k] = round-to-nearest(2^32 * P(Bin(nb, q) <= k)) clamped to 2^32, q = t_any / 2^32, IEEE
 * doubles in exactly this operation order (DESIGN.md "Sampler").  K = number of k < nb with u >= cdf[k]. */
#define SEGMENT 512
static void binomial_cdf_direct(u64 t_any, int nb, u64* cdf) {
    for (int k = 0; k <= SEGMENT; ++k) cdf[k] = 4294967296ull;
    if (nb <= 0) return;
    if (t_any >= 4294967296ull) {
        for (int k = 0; k < nb; ++k) cdf[k] = 0;
        return;
    }
    const double q = (double)t_any / 4294967296.0, om = 1.0 - q;
    double pmf = 1.0;
    for (int i = 0; i < nb; ++i) pmf *= om;
    double cum = 0.0;
    for (int k = 0; k < nb; ++k) {
        cum += pmf;
        double c = __builtin_floor(cum * 4294967296.0 + 0.5);
        if (c > 4294967296.0) c = 4294967296.0;
        cdf[k] = (u64)c;
        pmf = pmf * (double)(nb - k) / (double)(k + 1) * q / om;
    }
}
/* Segments of more than 64 qubits at q > 1/2: from the complementary count nb - K ~ Bin(nb, 1 - q), whose (1 - q')^nb does
 * not underflow. */
static void binomial_cdf(u64 t_any, int nb, u64* cdf) {
    if (nb <= 64 || t_any <= 2147483648ull || t_any >= 4294967296ull) {
        binomial_cdf_direct(t_any, nb, cdf);
        return;
    }
    u64 other[SEGMENT + 1];
    binomial_cdf_direct(4294967296ull - t_any, nb, other);
    for (int k = 0; k <= SEGMENT; ++k) cdf[k] = 4294967296ull;
    for (int k = 0; k < nb; ++k) cdf[k] = 4294967296ull - other[nb - k - 1];
}

/* Sample-major rows of lde words; per SEGMENT of 512 qubits one draw for the number of errors, one per erroneous qubit. */
int orc_sample_errors(int64_t n, u64 seed, int64_t first, int64_t count, double p_x, double p_y, double p_z,
                      u64* ex, u64* ez, int64_t lde) {
    const double p_t = p_x + p_y + p_z, p_xy = p_x + p_y;
    const u64 t_any = quantise(p_t);
    const u64 t_1 = p_t > 0.0 ? quantise(p_x / p_t) : 0;          /* X only below t_1, Y below t_2, else Z */
    const u64 t_2 = p_t > 0.0 ? quantise(p_xy / p_t) : 0;
    const int64_t segments = (n + SEGMENT - 1) / SEGMENT;
    const int nb_last = n > 0 ? (int)(n - (segments - 1) * SEGMENT) : 0;
    static u64 cdf_full[SEGMENT + 1], cdf_last[SEGMENT + 1], made_t = ~0ull;   /* (kept between calls: the decode tally asks per sample) */
    static int made_nb = -1;
    if (made_t != t_any || made_nb != nb_last) {
        binomial_cdf(t_any, SEGMENT, cdf_full);
        binomial_cdf(t_any, nb_last, cdf_last);
        made_t = t_any, made_nb = nb_last;
    }
    for (int64_t i = 0; i < count; ++i) {
        const u64 ks = mix64(seed + GOLDEN * ((u64)(first + i) + 1));
        for (int64_t w = 0; w < lde; ++w) ex[i * lde + w] = 0, ez[i * lde + w] = 0;
        for (int64_t s = 0; s < segments; ++s) {
            const int nb = s == segments - 1 ? nb_last : SEGMENT;
            const u64* cdf = s == segments - 1 ? cdf_last : cdf_full;
            const u64 d = mix64(ks + STREAM_MULT * ((u64)s + 1));          /* the segment's draw */
            int k_err = 0;
            while (k_err < nb && (d >> 32) >= cdf[k_err]) k_err += 1;      /* the table is non-decreasing */
            u64 chosen[SEGMENT / 64] = {0};
            for (int k = 0; k < k_err; ++k) {                             /* Floyd: k_err distinct positions, one draw each */
                const u64 v = mix64(d + GOLDEN * ((u64)k + 1));
                const int j = nb - k_err + k;
                const int t = (int)(((v >> 32) * (u64)(j + 1)) >> 32);
                const int pos = ((chosen[t >> 6] >> (t & 63)) & 1ull) ? j : t;
                const u64 kind = v & 0xFFFFFFFFull;
                const int64_t word = s * (SEGMENT / 64) + (pos >> 6);
                chosen[pos >> 6] |= 1ull << (pos & 63);
                if (kind < t_2) ex[i * lde + word] |= 1ull << (pos & 63);  /* X or Y */
                if (kind >= t_1) ez[i * lde + word] |= 1ull << (pos & 63); /* Y or Z */
            }
        }
    }
    return 0;
}

/* Whole Monte-Carlo: X errors against h2, Z errors against h1 (css_code.py:457-470).  Histograms are
 * overwritten.  Works in chunks of 4096 samples. */
int orc_mc(const u64* h1, int64_t r1, int64_t ld1, const u64* h2, int64_t r2, int64_t ld2, int64_t n, u64 seed,
           int64_t first, int64_t count, double p_x, double p_y, double p_z, int mode, u64* hist_z, int64_t nbins_z,
           u64* hist_x, int64_t nbins_x) {
    const int64_t lde = (n + 63) >> 6 ? (n + 63) >> 6 : 1;
    const int64_t ls1 = (r1 + 63) >> 6 ? (r1 + 63) >> 6 : 1, ls2 = (r2 + 63) >> 6 ? (r2 + 63) >> 6 : 1;
    const int64_t chunk = 4096;
    u64* ex = (u64*)malloc((size_t)chunk * lde * 8);
    u64* ez = (u64*)malloc((size_t)chunk * lde * 8);
    u64* s1 = (u64*)malloc((size_t)chunk * ls1 * 8);
    u64* s2 = (u64*)malloc((size_t)chunk * ls2 * 8);
    memset(hist_z, 0, (size_t)nbins_z * 8);
    memset(hist_x, 0, (size_t)nbins_x * 8);
    for (int64_t done = 0; done < count; done += chunk) {
        const int64_t now = count - done < chunk ? count - done : chunk;
        orc_sample_errors(n, seed, first + done, now, p_x, p_y, p_z, ex, ez, lde);
        orc_syndrome_batch(h1, r1, n, ld1, ez, now, lde, s1, ls1);
        orc_syndrome_batch(h2, r2, n, ld2, ex, now, lde, s2, ls2);
        orc_histogram(s1, now, ls1, r1, mode, hist_z);
        orc_histogram(s2, now, ls2, r2, mode, hist_x);
    }
    free(ex);
    free(ez);
    free(s1);
    free(s2);
    return 0;
}

/* [build-defined, SURVEY.md 8f item 1]  Table decode + logical tally for codes with n <= 63: tables are 2^r words
 * indexed by the big-endian syndrome key (css_code.py:729), ~0 = no entry (css_code.py:655-657: error unchanged);
 * logical flips per css_code.py:640-646.  counts[5] as in include/gf2hip.h, overwritten. */
int orc_mc_decode(const u64* h1, int64_t r1, const u64* h2, int64_t r2, int64_t n, const u64* t1, const u64* t2, u64 xop,
                  u64 zop, u64 seed, int64_t first, int64_t count, double p_x, double p_y, double p_z, u64* counts) {
    for (int k = 0; k < 5; ++k) counts[k] = 0;
    for (int64_t i = 0; i < count; ++i) {
        u64 ex, ez;
        orc_sample_errors(n, seed, first + i, 1, p_x, p_y, p_z, &ex, &ez, 1);
        u64 kx = 0, kz = 0;
        for (int64_t k = 0; k < r2; ++k) kx = (kx << 1) | (u64)(__builtin_popcountll(h2[k] & ex) & 1);
        for (int64_t k = 0; k < r1; ++k) kz = (kz << 1) | (u64)(__builtin_popcountll(h1[k] & ez) & 1);
        const int miss_x = t2[kx] == ~0ull, miss_z = t1[kz] == ~0ull;
        const u64 res_x = miss_x ? ex : ex ^ t2[kx], res_z = miss_z ? ez : ez ^ t1[kz];
        const int fx = __builtin_popcountll(zop & res_x) & 1, fz = __builtin_popcountll(xop & res_z) & 1;
        counts[0] += (u64)fx;
        counts[1] += (u64)fz;
        counts[2] += (u64)(fx | fz);
        counts[3] += (u64)miss_x;
        counts[4] += (u64)miss_z;
    }
    return 0;
}

/* ---- syndrome_table (css_code.py:715-735) on packed words, for checks of up to 128 rows --------------------------------------
 * for w = 0, 1, 2, ...: every error of weight w in bin_matrix.weight_w_vectors order (bin_matrix.py:57-72: supports ascending
 * lexicographically), key = vec_to_int(H e mod 2) (bin_matrix.py:36-43: row 0 is the most significant bit; exact here, two
 * words, where the reference's int64 wraps beyond 63 bits); a key seen before -- in an earlier class or earlier in this one --
 * ends the search: return w - 1 and the entries of the classes before w (css_code.py:730-731).  No collision: t = n.
 * max_weight >= 0 [build-defined cap]: stop before class max_weight + 1 with t = max_weight.
 * Outputs in insertion order: keys_out[2 i], keys_out[2 i + 1] = low / high word of entry i's key, errs_out[i * lde ..] its
 * packed error; at most `capacity` entries are written (*entries_out says how many there are).  Returns 0. */
static u64 mixkey(u64 lo, u64 hi) {
    u64 z = lo ^ (hi * 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

typedef struct {
    u64 *lo, *hi;
    unsigned char* used;
    u64 slots, count;
} keyset;

static int keyset_init(keyset* s, u64 slots) {
    s->lo = (u64*)malloc(slots * 8);
    s->hi = (u64*)malloc(slots * 8);
    s->used = (unsigned char*)calloc(slots, 1);
    s->slots = slots;
    s->count = 0;
    return s->lo && s->hi && s->used ? 0 : -1;
}
static void keyset_free(keyset* s) {
    free(s->lo);
    free(s->hi);
    free(s->used);
}
static int keyset_has(const keyset* s, u64 lo, u64 hi) {
    for (u64 at = mixkey(lo, hi) & (s->slots - 1); s->used[at]; at = (at + 1) & (s->slots - 1))
        if (s->lo[at] == lo && s->hi[at] == hi) return 1;
    return 0;
}
static void keyset_put(keyset* s, u64 lo, u64 hi) {
    u64 at = mixkey(lo, hi) & (s->slots - 1);
    while (s->used[at]) at = (at + 1) & (s->slots - 1);
    s->used[at] = 1;
    s->lo[at] = lo;
    s->hi[at] = hi;
    s->count += 1;
}
static int keyset_grow(keyset* s) {
    keyset bigger;
    if (keyset_init(&bigger, s->slots * 2)) return -1;
    for (u64 at = 0; at < s->slots; ++at)
        if (s->used[at]) keyset_put(&bigger, s->lo[at], s->hi[at]);
    keyset_free(s);
    *s = bigger;
    return 0;
}

int orc_syndrome_table(const u64* h, int64_t r, int64_t n, int64_t ld, int64_t max_weight, int64_t capacity, u64* keys_out,
                       u64* errs_out, int64_t lde, int64_t* t_out, int64_t* entries_out) {
    if (r > 128 || r < 0 || n < 0) return -1;
    /* column keys: bit r - 1 - i of column j = H[i][j] */
    u64* ck = (u64*)calloc((size_t)(n > 0 ? n : 1) * 2, 8);
    for (int64_t i = 0; i < r; ++i) {
        const int64_t b = r - 1 - i;
        for (int64_t j = 0; j < n; ++j)
            if (bit(h + i * ld, j)) ck[2 * j + (b >> 6)] |= 1ull << (b & 63);
    }
    keyset all, layer;
    if (keyset_init(&all, 1024) || keyset_init(&layer, 1024)) return -2;
    int64_t* sup = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
    int64_t entries = 0, t = n;
    for (int64_t w = 0; w <= n; ++w) {
        if (max_weight >= 0 && w > max_weight) {
            t = max_weight;
            break;
        }
        /* the class's keys go into `layer` first: a collision discards the class as a whole */
        memset(layer.used, 0, layer.slots);
        layer.count = 0;
        const int64_t first_entry = entries;
        int collided = 0;
        for (int64_t k = 0; k < w; ++k) sup[k] = k;                 /* first support in lexicographic order */
        for (;;) {
            u64 lo = 0, hi = 0;
            for (int64_t k = 0; k < w; ++k) {
                lo ^= ck[2 * sup[k]];
                hi ^= ck[2 * sup[k] + 1];
            }
            if (keyset_has(&all, lo, hi) || keyset_has(&layer, lo, hi)) {
                collided = 1;
                break;
            }
            if ((layer.count + 1) * 2 > layer.slots && keyset_grow(&layer)) return -2;
            keyset_put(&layer, lo, hi);
            if (entries < capacity) {
                keys_out[2 * entries] = lo;
                keys_out[2 * entries + 1] = hi;
                memset(errs_out + entries * lde, 0, (size_t)lde * 8);
                for (int64_t k = 0; k < w; ++k) errs_out[entries * lde + (sup[k] >> 6)] |= 1ull << (sup[k] & 63);
            }
            entries += 1;
            /* next support, ascending lexicographically: the last position that can still move up does, the ones behind it follow */
            int64_t k = w - 1;
            while (k >= 0 && sup[k] == n - w + k) --k;
            if (k < 0) break;
            sup[k] += 1;
            for (int64_t q = k + 1; q < w; ++q) sup[q] = sup[q - 1] + 1;
        }
        if (collided) {
            entries = first_entry;
            t = w - 1;
            break;
        }
        for (u64 at = 0; at < layer.slots; ++at)
            if (layer.used[at]) {
                if ((all.count + 1) * 2 > all.slots && keyset_grow(&all)) return -2;
                keyset_put(&all, layer.lo[at], layer.hi[at]);
            }
    }
    keyset_free(&all);
    keyset_free(&layer);
    free(sup);
    free(ck);
    *t_out = t;
    *entries_out = entries;
    return 0;
}

/* [build-defined, SURVEY.md 8f item 1] orc_mc_decode for codes of up to 128 qubits and checks of up to 128 rows, the tables given
 * by their entries (keys: two words each, low first; corr: two words each) as orc_syndrome_table returns them.  Same rules:
 * css_code.py:655-657 (no entry: the error stays), css_code.py:640-646 (logical flips).  h1 / h2: packed rows of ld words;
 * xop / zop: two words each.  counts[5] as in include/gf2hip.h, overwritten. */
int orc_mc_decode_wide(const u64* h1, int64_t r1, const u64* h2, int64_t r2, int64_t n, int64_t ld, const u64* keys1,
                       const u64* corr1, int64_t entries1, const u64* keys2, const u64* corr2, int64_t entries2, const u64* xop,
                       const u64* zop, u64 seed, int64_t first, int64_t count, double p_x, double p_y, double p_z, u64* counts) {
    if (n > 128 || ld > 2 || ld < 1) return -1;
    for (int k = 0; k < 5; ++k) counts[k] = 0;
    /* index of each table: slot -> entry + 1 */
    const u64* keys[2] = {keys2, keys1};                            /* side 0: X errors against h2, side 1: Z errors against h1 */
    const u64* corr[2] = {corr2, corr1};
    const int64_t ents[2] = {entries2, entries1};
    const u64* hs[2] = {h2, h1};
    const int64_t rs[2] = {r2, r1};
    const u64* ops[2] = {zop, xop};
    u64 slots[2];
    int64_t* index[2];
    for (int c = 0; c < 2; ++c) {
        slots[c] = 1024;
        while (slots[c] < (u64)ents[c] * 2 + 2) slots[c] <<= 1;
        index[c] = (int64_t*)calloc(slots[c], sizeof(int64_t));
        for (int64_t i = 0; i < ents[c]; ++i) {
            u64 at = mixkey(keys[c][2 * i], keys[c][2 * i + 1]) & (slots[c] - 1);
            while (index[c][at]) at = (at + 1) & (slots[c] - 1);
            index[c][at] = i + 1;
        }
    }
    for (int64_t i = 0; i < count; ++i) {
        u64 e[2][2] = {{0, 0}, {0, 0}};
        orc_sample_errors(n, seed, first + i, 1, p_x, p_y, p_z, e[0], e[1], ld);
        int flip[2], miss[2];
        for (int c = 0; c < 2; ++c) {
            u64 lo = 0, hi = 0;                                     /* vec_to_int of the syndrome: row 0 most significant */
            for (int64_t k = 0; k < rs[c]; ++k) {
                int par = 0;
                for (int64_t w = 0; w < ld; ++w) par ^= __builtin_popcountll(hs[c][k * ld + w] & e[c][w]) & 1;
                const int64_t b = rs[c] - 1 - k;
                if (par) { if (b >> 6) hi |= 1ull << (b & 63); else lo |= 1ull << (b & 63); }
            }
            int64_t found = 0;
            for (u64 at = mixkey(lo, hi) & (slots[c] - 1); index[c][at]; at = (at + 1) & (slots[c] - 1)) {
                const int64_t ent = index[c][at] - 1;
                if (keys[c][2 * ent] == lo && keys[c][2 * ent + 1] == hi) { found = ent + 1; break; }
            }
            miss[c] = found == 0;
            u64 res[2] = {e[c][0], e[c][1]};
            if (found) {
                res[0] ^= corr[c][2 * (found - 1)];
                res[1] ^= corr[c][2 * (found - 1) + 1];
            }
            flip[c] = (__builtin_popcountll(ops[c][0] & res[0]) + __builtin_popcountll(ops[c][1] & res[1])) & 1;
        }
        counts[0] += (u64)flip[0];
        counts[1] += (u64)flip[1];
        counts[2] += (u64)(flip[0] | flip[1]);
        counts[3] += (u64)miss[0];
        counts[4] += (u64)miss[1];
    }
    free(index[0]);
    free(index[1]);
    return 0;
}
