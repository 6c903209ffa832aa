/*
 * Deterministic synthetic corpora for the benchmark configurations (bench.py and the large-size
 * property tests).  Not part of the shipped library and not part of the oracle.
 *
 *   kind 0 ("B")  ASCII English-like prose: 4096-word vocabulary (a-z, 1-12 letters, Zipf s=1 rank
 *                 sampling), single spaces, ". " + capital every 8-20 words, '\n' every 80-120 bytes,
 *                 a blank line about every 600 bytes, 3 % of the words replaced by 1-6 digit numbers.
 *   kind 1 ("C")  OCR-noisy markdown with multi-byte UTF-8.  Lines by count: 35 % prose (10 % of the
 *                 words multi-byte: Latin-1 / Greek / Cyrillic / CJK / emoji; 1 % of the separators
 *                 U+00A0 or U+3000), 25 % OCR rows (2-4 upper-case words + 8-14 numeric columns), 15 %
 *                 numeric-only, 10 % GFM table rows (header / separator / data), 5 % blank, 5 %
 *                 symbol-only, 5 % other; 5 % of all lines end with CRLF.
 *   kind 2 ("D")  kind-0 prose with 20 % kind-1 lines (used for the skewed-length configuration).
 *   kind 3 ("W")  kind 1 with half of the multi-byte words taken from scripts that the GPU classifier has no
 *                 bit-plane rule for (Georgian, Ethiopic, Bengali, Tamil, Armenian, Khmer, Tibetan, Cherokee):
 *                 every such character goes through the per-character table path.
 *
 * Record r is generated from splitmix64 seeded with seed ^ (r+1)*0x9E3779B97F4A7C15 and depends on
 * nothing else, so any record range can be produced independently (sharding, threads).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;
static uint64_t next64(rng_t* r) {
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint32_t below(rng_t* r, uint32_t n) { return (uint32_t)((next64(r) >> 11) % n); }
static double unit(rng_t* r) { return (double)(next64(r) >> 11) * (1.0 / 9007199254740992.0); }

#define VOCAB 4096
static char g_vocab[VOCAB][13];
static uint8_t g_vlen[VOCAB];
static double g_cdf[VOCAB];
static int g_init = 0;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

static const char* MB_WORDS[] = {"Société", "Ünternehmen", "año", "façade", "αγορά", "κέρδος", "β",
                                 "прибыль", "рынок", "北京工业有限公司", "東京", "収益", "市场", "🚀", "🌏",
                                 "naïve", "Ελλάδα", "доход"};
#define N_MB (sizeof(MB_WORDS) / sizeof(MB_WORDS[0]))
static const char* WIDE_WORDS[] = {"საქართველო", "ბაზარი", "ኢትዮጵያ", "ገበያ", "বাংলা", "বাজার", "தமிழ்", "சந்தை",
                                   "հայերեն", "շուկա", "ខ្មែរ", "བོད་ཡིག", "ᏣᎳᎩ", "ગુજરાતી", "ਪੰਜਾਬੀ", "മലയാളം"};
#define N_WIDE (sizeof(WIDE_WORDS) / sizeof(WIDE_WORDS[0]))
static const char* SYMS[] = {"—", "—", "(0.8)", "(203.5)", "***", "--", "&", "€", "·", "=="};
#define N_SYMS (sizeof(SYMS) / sizeof(SYMS[0]))

static void init_vocab(void) {
    pthread_mutex_lock(&g_mu);
    if (!g_init) {
        rng_t r = {42};
        double h = 0;
        for (int i = 0; i < VOCAB; i++) {
            int len = 1 + (int)below(&r, 12);
            if (i < 64 && len > 5) len = 1 + len % 5; /* frequent words are short */
            for (int k = 0; k < len; k++) g_vocab[i][k] = (char)('a' + below(&r, 26));
            g_vocab[i][len] = 0;
            g_vlen[i] = (uint8_t)len;
            h += 1.0 / (double)(i + 1);
            g_cdf[i] = h;
        }
        for (int i = 0; i < VOCAB; i++) g_cdf[i] /= h;
        g_init = 1;
    }
    pthread_mutex_unlock(&g_mu);
}

typedef struct { uint8_t* p; uint64_t n; } sink_t; /* p == NULL: count only */
static void put(sink_t* s, const void* d, uint64_t n) {
    if (s->p) memcpy(s->p + s->n, d, n);
    s->n += n;
}
static void putc1(sink_t* s, char c) { put(s, &c, 1); }

static int zipf_word(rng_t* r) {
    double u = unit(r);
    int lo = 0, hi = VOCAB - 1;
    while (lo < hi) { int mid = (lo + hi) / 2; if (g_cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return lo;
}
static void put_number(sink_t* s, rng_t* r, int maxd) {
    int nd = 1 + (int)below(r, (uint32_t)maxd);
    char b[8];
    for (int i = 0; i < nd; i++) b[i] = (char)('0' + below(r, 10));
    if (nd > 1 && b[0] == '0') b[0] = '1';
    put(s, b, (uint64_t)nd);
}
static void put_word(sink_t* s, rng_t* r, int cap, int upper) {
    int w = zipf_word(r);
    char b[13];
    memcpy(b, g_vocab[w], g_vlen[w]);
    if (upper) for (int i = 0; i < g_vlen[w]; i++) b[i] = (char)(b[i] - 32);
    else if (cap) b[0] = (char)(b[0] - 32);
    put(s, b, g_vlen[w]);
}

/* one prose line of about `width` bytes (no terminator) */
static void prose_line(sink_t* s, rng_t* r, uint32_t width, int multibyte, uint32_t* sentence_left) {
    uint64_t start = s->n;
    int first = 1;
    while (s->n - start < width) {
        if (!first) {
            if (multibyte && below(r, 100) == 0) put(s, below(r, 2) ? "\xC2\xA0" : "\xE3\x80\x80", below(r, 2) ? 2 : 3);
            else putc1(s, ' ');
        }
        first = 0;
        int cap = 0;
        if (*sentence_left == 0) { cap = 1; *sentence_left = 8 + below(r, 13); }
        if (below(r, 100) < 3) put_number(s, r, 6);
        else if (multibyte && below(r, 10) == 0) {
            const char* w = (multibyte == 2 && below(r, 2)) ? WIDE_WORDS[below(r, N_WIDE)] : MB_WORDS[below(r, N_MB)];
            put(s, w, strlen(w));
        }
        else put_word(s, r, cap, 0);
        if (--*sentence_left == 0) putc1(s, '.');
    }
}
static void ocr_row(sink_t* s, rng_t* r) {
    int lead = (int)below(r, 3);
    for (int i = 0; i < lead; i++) { put_number(s, r, 3); putc1(s, ' '); }
    int names = 2 + (int)below(r, 3);
    for (int i = 0; i < names; i++) { put_word(s, r, 0, 1); putc1(s, ' '); }
    int cols = 8 + (int)below(r, 7);
    for (int i = 0; i < cols; i++) {
        if (i) putc1(s, ' ');
        uint32_t k = below(r, 20);
        if (k == 0) put(s, "—", 3);
        else if (k == 1) { putc1(s, '('); put_number(s, r, 3); putc1(s, '.'); put_number(s, r, 1); putc1(s, ')'); }
        else if (k == 2) { put_number(s, r, 2); putc1(s, ','); put_number(s, r, 3); }
        else put_number(s, r, 5);
    }
}
static void numeric_row(sink_t* s, rng_t* r) {
    int cols = 3 + (int)below(r, 12);
    for (int i = 0; i < cols; i++) { if (i) putc1(s, ' '); put_number(s, r, 6); }
}
static void symbol_row(sink_t* s, rng_t* r) {
    int cols = 1 + (int)below(r, 8);
    for (int i = 0; i < cols; i++) { if (i) putc1(s, ' '); const char* w = SYMS[below(r, N_SYMS)]; put(s, w, strlen(w)); }
}
static void table_row(sink_t* s, rng_t* r) {
    int cols = 2 + (int)below(r, 4);
    uint32_t kind = below(r, 10);
    putc1(s, '|');
    for (int i = 0; i < cols; i++) {
        if (kind < 2) { put(s, below(r, 2) ? "------" : ":----:", 6); }
        else {
            putc1(s, ' ');
            if (kind < 5 || i == 0) { put_word(s, r, 1, 0); if (below(r, 3) == 0) { putc1(s, ' '); put_word(s, r, 0, 0); } }
            else {
                uint32_t k = below(r, 3);
                if (k == 0) { putc1(s, '$'); put_number(s, r, 1); putc1(s, '.'); put_number(s, r, 1); putc1(s, 'B'); }
                else if (k == 1) { putc1(s, '+'); put_number(s, r, 2); putc1(s, '%'); }
                else put_number(s, r, 4);
            }
            int pad = 1 + (int)below(r, 4);
            for (int k = 0; k < pad; k++) putc1(s, ' ');
        }
        putc1(s, '|');
    }
}
static void noisy_line(sink_t* s, rng_t* r, uint32_t* sentence_left, int multibyte) {
    uint32_t k = below(r, 100);
    if (k < 35) prose_line(s, r, 60 + below(r, 60), multibyte, sentence_left);
    else if (k < 60) ocr_row(s, r);
    else if (k < 75) numeric_row(s, r);
    else if (k < 85) table_row(s, r);
    else if (k < 90) { /* blank */ }
    else if (k < 95) symbol_row(s, r);
    else { put_word(s, r, 1, 0); putc1(s, ' '); numeric_row(s, r); }
}

/* One record of at most `target` bytes (it stops at the last line that fits). */
static void gen_record(int kind, uint64_t seed, uint64_t r, uint64_t target, sink_t* s) {
    rng_t g = {seed ^ ((r + 1) * 0x9E3779B97F4A7C15ull)};
    uint64_t start = s->n;
    uint32_t sentence_left = 0;
    uint64_t next_blank = 500 + below(&g, 200);
    while (s->n - start + 140 < target) {
        int noisy = kind == 1 || kind == 3 || (kind == 2 && below(&g, 5) == 0);
        if (noisy) noisy_line(s, &g, &sentence_left, kind == 3 ? 2 : 1);
        else prose_line(s, &g, 80 + below(&g, 41), 0, &sentence_left);
        if (kind != 0 && below(&g, 20) == 0) put(s, "\r\n", 2);
        else putc1(s, '\n');
        if (s->n - start > next_blank) { putc1(s, '\n'); next_blank += 500 + below(&g, 200); }
    }
    /* fill the remainder with a short last line without terminator */
    while (s->n - start + 14 < target) { put_word(s, &g, 0, 0); putc1(s, ' '); }
}

typedef struct {
    int kind; uint64_t seed, r0, r1; const uint64_t* targets; uint64_t* lens; const uint64_t* offsets; uint8_t* out;
} job_t;
static void* job_main(void* p) {
    job_t* j = (job_t*)p;
    for (uint64_t r = j->r0; r < j->r1; r++) {
        sink_t s = {j->out ? j->out + j->offsets[r] : NULL, 0};
        gen_record(j->kind, j->seed, r, j->targets[r], &s);
        if (j->lens) j->lens[r] = s.n;
    }
    return NULL;
}
static void run_jobs(int kind, uint64_t seed, uint64_t R, const uint64_t* targets, uint64_t* lens,
                     const uint64_t* offsets, uint8_t* out, int threads) {
    if (threads < 1) threads = 1;
    job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        job_t jb = {kind, seed, R * (uint64_t)t / (uint64_t)threads, R * (uint64_t)(t + 1) / (uint64_t)threads,
                    targets, lens, offsets, out};
        jobs[t] = jb;
        pthread_create(&th[t], NULL, job_main, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(jobs); free(th);
}

/* Record length targets.  mode 0: uniform in [lo, hi].  mode 1 (skewed): each record is `lo` bytes or
 * `hi` bytes, chosen so that about half of the bytes fall in each class. */
uint64_t cg_plan(uint64_t seed, uint64_t total_bytes, uint64_t lo, uint64_t hi, int mode, uint64_t* targets,
                 uint64_t max_records) {
    rng_t g = {seed ^ 0xD1B54A32D192ED03ull};
    uint64_t n = 0, sum = 0;
    double p_big = (double)lo / (double)(lo + hi); /* P(record is big) for equal byte shares */
    while (sum < total_bytes && n < max_records) {
        uint64_t t;
        if (mode == 0) t = lo + (next64(&g) >> 11) % (hi - lo + 1);
        else t = unit(&g) < p_big ? hi : lo;
        if (sum + t > total_bytes) t = total_bytes - sum;
        targets[n++] = t;
        sum += t;
    }
    return n;
}

/* Generates R records; offsets has R+1 entries (written); returns the total byte count.  Pass
 * out == NULL to size the buffer first (offsets are filled in either case). */
uint64_t cg_generate(int kind, uint64_t seed, uint64_t R, const uint64_t* targets, uint64_t* offsets, uint8_t* out,
                     int threads) {
    init_vocab();
    uint64_t* lens = (uint64_t*)calloc(R ? R : 1, sizeof(uint64_t));
    run_jobs(kind, seed, R, targets, lens, NULL, NULL, threads);
    uint64_t sum = 0;
    for (uint64_t r = 0; r < R; r++) { offsets[r] = sum; sum += lens[r]; }
    offsets[R] = sum;
    free(lens);
    if (out) run_jobs(kind, seed, R, targets, NULL, offsets, out, threads);
    return sum;
}
