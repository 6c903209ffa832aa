/*
 * CPU oracle (test infrastructure only - see oracle.h).  Plain C restatement of the
 * reference's denoise -> tokenise -> chunk path, kept deliberately close to the reference's
 * own control flow (line by line, token vectors, literal wave loop) so that it can be read
 * side by side with it.  It is NOT optimised and NOT part of the shipped library.
 */
#include "oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define TRP_TABLE_QUAL static const
#include "unicode_tables.inc"

/* ------------------------------------------------------------------------------------------
 * Character model: Rust `char` predicates over UTF-8 text.
 * ---------------------------------------------------------------------------------------- */

/* char::is_alphabetic - Unicode derived property Alphabetic. */
int orc_is_alphabetic(uint32_t cp) {
    if (cp < 0x80) return ((cp | 0x20) - 'a') < 26u;
    uint32_t blk = cp >> 8;
    if (blk >= TRP_ALPHA_STAGE1_LEN) return 0;
    uint32_t row = trp_alpha_stage1[blk];
    uint32_t o = cp & 255u;
    return (trp_alpha_stage2[row * 8 + (o >> 5)] >> (o & 31u)) & 1u;
}

/* char::is_whitespace - Unicode property White_Space (25 code points). */
int orc_is_whitespace(uint32_t cp) {
    if (cp == 0x20 || (cp >= 0x09 && cp <= 0x0D)) return 1;
    if (cp < 0x80) return 0;
    return cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) ||
           cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}

/* Decode one scalar value starting at p (p < end).  Input is required to be valid UTF-8
 * (Rust &str); malformed or truncated sequences decode as one opaque byte so that the
 * function is total. */
static size_t decode(const uint8_t* p, const uint8_t* end, uint32_t* cp) {
    uint32_t b0 = p[0];
    size_t n;
    if (b0 < 0x80) { *cp = b0; return 1; }
    if (b0 < 0xC0) { *cp = 0xFFFD; return 1; }
    if (b0 < 0xE0) { n = 2; *cp = b0 & 0x1F; }
    else if (b0 < 0xF0) { n = 3; *cp = b0 & 0x0F; }
    else { n = 4; *cp = b0 & 0x07; }
    if ((size_t)(end - p) < n) { *cp = 0xFFFD; return 1; }
    for (size_t i = 1; i < n; i++) {
        if ((p[i] & 0xC0) != 0x80) { *cp = 0xFFFD; return 1; }
        *cp = (*cp << 6) | (p[i] & 0x3F);
    }
    return n;
}

/* Growable byte buffer (Rust String). */
typedef struct { uint8_t* p; size_t len, cap; } buf_t;
static void buf_reserve(buf_t* b, size_t extra) {
    if (b->len + extra <= b->cap) return;
    size_t c = b->cap ? b->cap * 2 : 64;
    while (c < b->len + extra) c *= 2;
    b->p = (uint8_t*)realloc(b->p, c);
    b->cap = c;
}
static void buf_push(buf_t* b, const uint8_t* s, size_t n) {
    buf_reserve(b, n);
    if (n) memcpy(b->p + b->len, s, n);
    b->len += n;
}
static void buf_push1(buf_t* b, uint8_t c) { buf_push(b, &c, 1); }

/* A borrowed slice (Rust &str). */
typedef struct { const uint8_t* p; size_t len; } str_t;

/* str::trim - strip White_Space chars at both ends. */
static str_t str_trim(str_t s) {
    const uint8_t* b = s.p;
    const uint8_t* e = s.p + s.len;
    while (b < e) {
        uint32_t cp;
        size_t n = decode(b, e, &cp);
        if (!orc_is_whitespace(cp)) break;
        b += n;
    }
    while (e > b) {
        const uint8_t* q = e - 1;
        while (q > b && (*q & 0xC0) == 0x80) q--;
        uint32_t cp;
        size_t n = decode(q, e, &cp);
        if (q + n != e || !orc_is_whitespace(cp)) break;
        e = q;
    }
    str_t r = { b, (size_t)(e - b) };
    return r;
}

/* str::split_whitespace collected into a Vec<&str>. */
typedef struct { str_t* v; size_t n, cap; } strvec_t;
static void sv_push(strvec_t* v, str_t s) {
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16;
        v->v = (str_t*)realloc(v->v, v->cap * sizeof(str_t));
    }
    v->v[v->n++] = s;
}
static void split_whitespace(str_t s, strvec_t* out) {
    const uint8_t* p = s.p;
    const uint8_t* e = s.p + s.len;
    const uint8_t* tok = NULL;
    while (p < e) {
        uint32_t cp;
        size_t n = decode(p, e, &cp);
        if (orc_is_whitespace(cp)) {
            if (tok) { str_t t = { tok, (size_t)(p - tok) }; sv_push(out, t); tok = NULL; }
        } else if (!tok) {
            tok = p;
        }
        p += n;
    }
    if (tok) { str_t t = { tok, (size_t)(e - tok) }; sv_push(out, t); }
}

/* count_digit_alpha (denoiser_preprocessor.rs:21-32). */
static void count_digit_alpha(str_t s, size_t* digits, size_t* alpha) {
    size_t d = 0, a = 0;
    const uint8_t* p = s.p;
    const uint8_t* e = s.p + s.len;
    while (p < e) {
        uint32_t cp;
        p += decode(p, e, &cp);
        if (cp >= '0' && cp <= '9') d++;
        else if (orc_is_alphabetic(cp)) a++;
    }
    *digits = d;
    *alpha = a;
}

/* digit_ratio (denoiser_preprocessor.rs:37-45): f32 arithmetic, usize -> f32 conversion. */
static float ratio_f32(size_t d, size_t a) {
    size_t total = d + a;
    if (total == 0) return 0.0f;
    volatile float fd = (float)d;
    volatile float ft = (float)total;
    volatile float r = fd / ft;
    return r;
}

/* strip_digit_tokens (denoiser_preprocessor.rs:70-140): literal wave expansion. */
static void strip_digit_tokens(str_t line, float max_digit_ratio, buf_t* out) {
    strvec_t toks = { 0, 0, 0 };
    split_whitespace(line, &toks);
    size_t n = toks.n;
    if (n == 0) { free(toks.v); return; }
    uint8_t* keep = (uint8_t*)calloc(n, 1);
    size_t* cd = (size_t*)calloc(n, sizeof(size_t));
    size_t* ca = (size_t*)calloc(n, sizeof(size_t));
    size_t* wave = (size_t*)calloc(n, sizeof(size_t));
    size_t d = 0, a = 0;
    for (size_t i = 0; i < n; i++) {
        count_digit_alpha(toks.v[i], &cd[i], &ca[i]);
        keep[i] = ca[i] > 0; /* has_alpha: any alphabetic char */
        if (keep[i]) { d += cd[i]; a += ca[i]; }
    }
    for (;;) {
        size_t nw = 0;
        for (size_t i = 0; i < n; i++)
            if (!keep[i] && ((i > 0 && keep[i - 1]) || (i + 1 < n && keep[i + 1]))) wave[nw++] = i;
        if (nw == 0) break;
        size_t wd = 0, wa = 0;
        for (size_t k = 0; k < nw; k++) { wd += cd[wave[k]]; wa += ca[wave[k]]; }
        size_t new_d = d + wd, new_a = a + wa;
        float new_ratio = ratio_f32(new_d, new_a);
        if (new_ratio > max_digit_ratio) break;
        for (size_t k = 0; k < nw; k++) keep[wave[k]] = 1;
        d = new_d;
        a = new_a;
    }
    int first = 1;
    for (size_t i = 0; i < n; i++) {
        if (!keep[i]) continue;
        if (!first) buf_push1(out, ' ');
        buf_push(out, toks.v[i].p, toks.v[i].len);
        first = 0;
    }
    free(keep); free(cd); free(ca); free(wave); free(toks.v);
}

static size_t count_byte(str_t s, uint8_t c) {
    size_t k = 0;
    for (size_t i = 0; i < s.len; i++) k += s.p[i] == c;
    return k;
}

/* is_markdown_table_line (denoiser_preprocessor.rs:151-154). */
static int is_markdown_table_line(str_t line) {
    str_t t = str_trim(line);
    return t.len > 0 && t.p[0] == '|' && count_byte(t, '|') >= 2;
}

/* is_markdown_table_separator (denoiser_preprocessor.rs:161-168). */
static int is_markdown_table_separator(str_t line) {
    str_t t = str_trim(line);
    if (!(t.len > 0 && t.p[0] == '|' && count_byte(t, '|') >= 2)) return 0;
    for (size_t i = 0; i < t.len; i++) {
        uint8_t c = t.p[i];
        if (!(c == '|' || c == '-' || c == ':' || c == ' ' || c == '\t')) return 0;
    }
    return 1;
}

/* strip_table_pipes (denoiser_preprocessor.rs:177-183). */
static void strip_table_pipes(str_t line, buf_t* out) {
    size_t start = 0;
    int first = 1;
    for (size_t i = 0; i <= line.len; i++) {
        if (i == line.len || line.p[i] == '|') {
            str_t cell = { line.p + start, i - start };
            cell = str_trim(cell);
            if (cell.len) {
                if (!first) buf_push1(out, ' ');
                buf_push(out, cell.p, cell.len);
                first = 0;
            }
            start = i + 1;
        }
    }
}

/* LineEnding::normalize of crate line-ending 1.5.1 as documented:
 * replace("\r\n", "\n") then replace("\r", "\n").  (Parity unpinned, see oracle.h.) */
static void normalize_line_endings(const uint8_t* text, size_t len, buf_t* out) {
    for (size_t i = 0; i < len; i++) {
        if (text[i] == '\r') {
            if (i + 1 < len && text[i + 1] == '\n') continue; /* "\r\n" -> "\n" */
            buf_push1(out, '\n');                               /* lone "\r" -> "\n" */
        } else {
            buf_push1(out, text[i]);
        }
    }
}

int orc_denoise(const uint8_t* text, size_t len, int enabled, float max_digit_ratio,
                int strip_markdown, uint8_t** out, size_t* out_len) {
    buf_t res = { 0, 0, 0 };
    if (!enabled) {
        buf_push(&res, text, len);
        *out = res.p;
        *out_len = res.len;
        return 1;
    }
    buf_t norm = { 0, 0, 0 };
    normalize_line_endings(text, len, &norm);
    size_t kept_lines = 0;
    /* str::lines(): split on '\n', a final empty piece is not a line, one trailing '\r'
     * stripped (none can remain after normalisation). */
    size_t pos = 0;
    while (pos < norm.len) {
        size_t e = pos;
        while (e < norm.len && norm.p[e] != '\n') e++;
        str_t line = { norm.p + pos, e - pos };
        if (line.len && line.p[line.len - 1] == '\r') line.len--;
        pos = e + 1; /* skip the '\n' (or run past the end) */

        buf_t stripped = { 0, 0, 0 };
        str_t effective = line;
        if (strip_markdown && is_markdown_table_line(line)) {
            if (is_markdown_table_separator(line)) continue;
            strip_table_pipes(line, &stripped);
            effective.p = stripped.p;
            effective.len = stripped.len;
        }
        size_t d, a;
        count_digit_alpha(effective, &d, &a);
        if (a == 0) { free(stripped.p); continue; }
        if (ratio_f32(d, a) > max_digit_ratio) {
            buf_t retained = { 0, 0, 0 };
            strip_digit_tokens(effective, max_digit_ratio, &retained);
            if (retained.len) {
                if (kept_lines) buf_push1(&res, '\n');
                buf_push(&res, retained.p, retained.len);
                kept_lines++;
            }
            free(retained.p);
        } else {
            if (kept_lines) buf_push1(&res, '\n');
            buf_push(&res, effective.p, effective.len);
            kept_lines++;
        }
        free(stripped.p);
    }
    free(norm.p);
    if (kept_lines == 0) {
        free(res.p);
        *out = NULL;
        *out_len = 0;
        return 0;
    }
    *out = res.p;
    *out_len = res.len;
    return 1;
}

size_t orc_tokenize(const uint8_t* text, size_t len, uint32_t* starts, uint32_t* lens, size_t cap) {
    strvec_t toks = { 0, 0, 0 };
    str_t s = { text, len };
    split_whitespace(s, &toks);
    for (size_t i = 0; i < toks.n && i < cap; i++) {
        starts[i] = (uint32_t)(toks.v[i].p - text);
        lens[i] = (uint32_t)toks.v[i].len;
    }
    size_t n = toks.n;
    free(toks.v);
    return n;
}

static uint64_t min_u64(uint64_t a, uint64_t b) { return a < b ? a : b; }
static uint64_t max_u64(uint64_t a, uint64_t b) { return a > b ? a : b; }

size_t orc_chunk_views(uint64_t n_tokens, const orc_config* cfg, uint32_t record, orc_chunk* out,
                       size_t cap) {
    size_t n = 0;
    if (n_tokens == 0) return 0;
    uint64_t total = n_tokens;
    uint64_t span = min_u64(cfg->max_window_tokens, total);
#define EMIT(K, I, O, S, B, E)                                                          \
    do {                                                                                \
        if (n < cap) {                                                                  \
            orc_chunk c = { record, (K), (uint32_t)(I), (uint32_t)(O), (uint32_t)(S),   \
                            (uint32_t)(B), (uint32_t)(E), 0 };                          \
            out[n] = c;                                                                 \
        }                                                                               \
        n++;                                                                            \
    } while (0)
    if (span == total) {
        EMIT(0, 0, 0, span, 0, total);
        return n;
    }
    for (uint32_t k = 0; k < cfg->n_overlaps; k++) {
        uint64_t overlap = cfg->overlaps[k];
        uint64_t stride = max_u64(span > overlap ? span - overlap : 0, 1);
        uint64_t start = 0, index = 0;
        while (start < total) {
            uint64_t end = min_u64(start + span, total);
            EMIT(0, index, overlap, span, start, end);
            if (end == total) break;
            start += stride;
            index += 1;
        }
    }
    if (total > cfg->max_window_tokens && cfg->summary_fallback_tokens > 0) {
        uint64_t fcap = max_u64(min_u64(cfg->summary_fallback_tokens, cfg->max_window_tokens), 1);
        uint64_t flen = min_u64(total, fcap);
        EMIT(1, 0, 0, 0, 0, flen);
    }
#undef EMIT
    return n;
}

/* ------------------------------------------------------------------------------------------
 * Batch driver.
 * ---------------------------------------------------------------------------------------- */

typedef struct {
    buf_t den;
    uint64_t* den_off;
    uint8_t* dropped;
    uint32_t* ts; uint32_t* tl; size_t nt, tcap;
    uint64_t* tok_off;
    orc_chunk* ch; size_t nc, ccap;
    uint64_t* ch_off;
} acc_t;

static void process_range(const uint8_t* bytes, const uint64_t* offsets, uint64_t r0, uint64_t r1,
                          const orc_config* cfg, acc_t* A) {
    uint64_t nr = r1 - r0;
    memset(A, 0, sizeof(*A));
    A->den_off = (uint64_t*)calloc(nr + 1, sizeof(uint64_t));
    A->tok_off = (uint64_t*)calloc(nr + 1, sizeof(uint64_t));
    A->ch_off = (uint64_t*)calloc(nr + 1, sizeof(uint64_t));
    A->dropped = (uint8_t*)calloc(nr ? nr : 1, 1);
    for (uint64_t r = r0; r < r1; r++) {
        uint64_t i = r - r0;
        const uint8_t* text = bytes + offsets[r];
        size_t len = (size_t)(offsets[r + 1] - offsets[r]);
        A->den_off[i] = A->den.len;
        A->tok_off[i] = A->nt;
        A->ch_off[i] = A->nc;
        uint8_t* den = NULL;
        size_t dlen = 0;
        int some = orc_denoise(text, len, cfg->denoise_enabled, cfg->max_digit_ratio,
                               cfg->strip_markdown, &den, &dlen);
        if (!some) { A->dropped[i] = 1; continue; }
        size_t base = A->den.len;
        buf_push(&A->den, den, dlen);
        size_t need = dlen / 2 + 2;
        if (A->nt + need > A->tcap) {
            A->tcap = (A->nt + need) * 2;
            A->ts = (uint32_t*)realloc(A->ts, A->tcap * sizeof(uint32_t));
            A->tl = (uint32_t*)realloc(A->tl, A->tcap * sizeof(uint32_t));
        }
        size_t nt = orc_tokenize(den, dlen, A->ts + A->nt, A->tl + A->nt, need);
        for (size_t k = 0; k < nt; k++) A->ts[A->nt + k] += (uint32_t)base;
        A->nt += nt;
        free(den);
        size_t ncks = orc_chunk_views(nt, cfg, (uint32_t)r, NULL, 0);
        if (A->nc + ncks > A->ccap) {
            A->ccap = (A->nc + ncks) * 2 + 16;
            A->ch = (orc_chunk*)realloc(A->ch, A->ccap * sizeof(orc_chunk));
        }
        orc_chunk_views(nt, cfg, (uint32_t)r, A->ch + A->nc, ncks);
        A->nc += ncks;
    }
    A->den_off[nr] = A->den.len;
    A->tok_off[nr] = A->nt;
    A->ch_off[nr] = A->nc;
}

int orc_batch(const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
              const orc_config* cfg, orc_result* out) {
    return orc_batch_mt(bytes, offsets, n_records, cfg, 1, out);
}

typedef struct {
    const uint8_t* bytes; const uint64_t* offsets; uint64_t r0, r1; const orc_config* cfg; acc_t A;
} job_t;
static void* job_main(void* p) {
    job_t* j = (job_t*)p;
    process_range(j->bytes, j->offsets, j->r0, j->r1, j->cfg, &j->A);
    return NULL;
}

int orc_batch_mt(const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
                 const orc_config* cfg, int n_threads, orc_result* out) {
    if (n_threads < 1) n_threads = 1;
    job_t* jobs = (job_t*)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    uint64_t total_bytes = offsets[n_records] - offsets[0];
    uint64_t r = 0;
    for (int t = 0; t < n_threads; t++) {
        /* byte-balanced static partition: thread t ends at the first record whose start
         * offset reaches (t+1)/n of the bytes */
        uint64_t target = offsets[0] + (total_bytes / (uint64_t)n_threads) * (uint64_t)(t + 1);
        uint64_t r1 = r;
        if (t == n_threads - 1) r1 = n_records;
        else while (r1 < n_records && offsets[r1] < target) r1++;
        jobs[t].bytes = bytes; jobs[t].offsets = offsets; jobs[t].r0 = r; jobs[t].r1 = r1;
        jobs[t].cfg = cfg;
        r = r1;
    }
    if (n_threads == 1) job_main(&jobs[0]);
    else {
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, job_main, &jobs[t]);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    uint64_t D = 0, T = 0, C = 0;
    for (int t = 0; t < n_threads; t++) { D += jobs[t].A.den.len; T += jobs[t].A.nt; C += jobs[t].A.nc; }
    memset(out, 0, sizeof(*out));
    out->n_records = n_records; out->n_denoised = D; out->n_tokens = T; out->n_chunks = C;
    out->denoised = (uint8_t*)malloc(D ? D : 1);
    out->denoised_offsets = (uint64_t*)malloc((n_records + 1) * sizeof(uint64_t));
    out->dropped = (uint8_t*)malloc(n_records ? n_records : 1);
    out->tok_start = (uint32_t*)malloc((T ? T : 1) * sizeof(uint32_t));
    out->tok_len = (uint32_t*)malloc((T ? T : 1) * sizeof(uint32_t));
    out->tok_offsets = (uint64_t*)malloc((n_records + 1) * sizeof(uint64_t));
    out->chunks = (orc_chunk*)malloc((C ? C : 1) * sizeof(orc_chunk));
    out->chunk_offsets = (uint64_t*)malloc((n_records + 1) * sizeof(uint64_t));
    uint64_t d0 = 0, t0 = 0, c0 = 0;
    for (int t = 0; t < n_threads; t++) {
        acc_t* A = &jobs[t].A;
        uint64_t nr = jobs[t].r1 - jobs[t].r0;
        if (A->den.len) memcpy(out->denoised + d0, A->den.p, A->den.len);
        for (uint64_t i = 0; i < nr; i++) {
            out->denoised_offsets[jobs[t].r0 + i] = A->den_off[i] + d0;
            out->tok_offsets[jobs[t].r0 + i] = A->tok_off[i] + t0;
            out->chunk_offsets[jobs[t].r0 + i] = A->ch_off[i] + c0;
            out->dropped[jobs[t].r0 + i] = A->dropped[i];
        }
        for (size_t k = 0; k < A->nt; k++) {
            out->tok_start[t0 + k] = A->ts[k] + (uint32_t)d0;
            out->tok_len[t0 + k] = A->tl[k];
        }
        if (A->nc) memcpy(out->chunks + c0, A->ch, A->nc * sizeof(orc_chunk));
        d0 += A->den.len; t0 += A->nt; c0 += A->nc;
        free(A->den.p); free(A->den_off); free(A->dropped); free(A->ts); free(A->tl);
        free(A->tok_off); free(A->ch); free(A->ch_off);
    }
    out->denoised_offsets[n_records] = D;
    out->tok_offsets[n_records] = T;
    out->chunk_offsets[n_records] = C;
    free(jobs); free(th);
    return 0;
}

void orc_result_free(orc_result* r) {
    free(r->denoised); free(r->denoised_offsets); free(r->dropped); free(r->tok_start);
    free(r->tok_len); free(r->tok_offsets); free(r->chunks); free(r->chunk_offsets);
    memset(r, 0, sizeof(*r));
}

size_t orc_chunk_text(const orc_result* r, uint64_t chunk_idx, uint8_t* out, size_t cap) {
    const orc_chunk* c = &r->chunks[chunk_idx];
    uint64_t rec = c->record;
    uint64_t t0 = r->tok_offsets[rec];
    uint64_t nt = r->tok_offsets[rec + 1] - t0;
    size_t n = 0;
    if (c->kind == 0 && c->tok_begin == 0 && c->tok_end == nt && c->span == nt && c->index == 0 &&
        c->overlap == 0 && r->chunk_offsets[rec + 1] - r->chunk_offsets[rec] == 1) {
        /* single window: the preprocessed text verbatim (sliding_window.rs:43-58) */
        uint64_t b = r->denoised_offsets[rec], e = r->denoised_offsets[rec + 1];
        for (uint64_t i = b; i < e; i++) { if (n < cap) out[n] = r->denoised[i]; n++; }
        return n;
    }
    for (uint64_t k = c->tok_begin; k < c->tok_end; k++) {
        if (k > c->tok_begin) { if (n < cap) out[n] = ' '; n++; }
        uint32_t s = r->tok_start[t0 + k], l = r->tok_len[t0 + k];
        for (uint32_t i = 0; i < l; i++) { if (n < cap) out[n] = r->denoised[s + i]; n++; }
    }
    return n;
}

/* Test hooks for the private helpers (the reference tests them directly:
 * denoiser_preprocessor.rs:370-411, :419-446, :454-465). */
int orc_is_markdown_table_line(const uint8_t* line, size_t len) {
    str_t s = { line, len };
    return is_markdown_table_line(s);
}
int orc_is_markdown_table_separator(const uint8_t* line, size_t len) {
    str_t s = { line, len };
    return is_markdown_table_separator(s);
}
size_t orc_strip_table_pipes(const uint8_t* line, size_t len, uint8_t* out) {
    str_t s = { line, len };
    buf_t b = { 0, 0, 0 };
    strip_table_pipes(s, &b);
    if (b.len) memcpy(out, b.p, b.len);
    size_t n = b.len;
    free(b.p);
    return n;
}

size_t orc_normalize_inline_whitespace(const uint8_t* text, size_t len, uint8_t* out) {
    buf_t b = { 0, 0, 0 };
    int seen_space = 0;
    const uint8_t* p = text;
    const uint8_t* e = text + len;
    while (p < e) {
        uint32_t cp;
        size_t n = decode(p, e, &cp);
        if (orc_is_whitespace(cp)) {
            if (!seen_space) { buf_push1(&b, ' '); seen_space = 1; }
        } else {
            buf_push(&b, p, n);
            seen_space = 0;
        }
        p += n;
    }
    str_t s = { b.p, b.len };
    s = str_trim(s);
    if (s.len) memmove(out, s.p, s.len);
    size_t r = s.len;
    free(b.p);
    return r;
}

/* ---- the steps right after the path -------------------------------------------------------------- */

static void push_joined(buf_t* b, const strvec_t* v, size_t n) {   /* tokens.take(n).join(" ") */
    for (size_t i = 0; i < n && i < v->n; i++) {
        if (i) buf_push1(b, ' ');
        buf_push(b, v->v[i].p, v->v[i].len);
    }
}

/* The body of decorate_chunk once a prefix was sampled (src/sampler/mod.rs:1484-1506, the rng-free twin
 * :1596-1618): the chunk text behind the prefix on its own line, cut to max_window_tokens in all. */
size_t orc_decorate_chunk(const uint8_t* text, size_t text_len, const uint8_t* prefix, size_t prefix_len,
                          uint32_t max_window_tokens, const uint8_t* newline, size_t newline_len,
                          uint8_t** out, uint64_t* tokens_estimate) {
    str_t body = { text, text_len }, pre = { prefix, prefix_len };
    strvec_t body_tokens = { 0, 0, 0 }, prefix_tokens = { 0, 0, 0 };
    split_whitespace(body, &body_tokens);                       /* :1484 */
    split_whitespace(pre, &prefix_tokens);                      /* :1485 */
    size_t total_tokens = prefix_tokens.n + body_tokens.n;      /* :1486 */
    size_t max_window = max_window_tokens;                      /* :1487 */
    buf_t b = { 0, 0, 0 };
    if (max_window > 0 && total_tokens > max_window) {          /* :1488 */
        if (prefix_tokens.n >= max_window) {                    /* :1489 */
            push_joined(&b, &prefix_tokens, max_window);        /* :1490-1494 */
            *tokens_estimate = max_window;                      /* :1495 */
        } else {
            size_t remaining = max_window - prefix_tokens.n;    /* :1497 */
            buf_push(&b, prefix, prefix_len);                   /* :1499-1500 */
            buf_push(&b, newline, newline_len);
            push_joined(&b, &body_tokens, remaining);
            *tokens_estimate = max_window;                      /* :1501 */
        }
    } else {
        buf_push(&b, prefix, prefix_len);                       /* :1504 */
        buf_push(&b, newline, newline_len);
        buf_push(&b, text, text_len);
        *tokens_estimate = total_tokens;                        /* :1505 */
    }
    free(body_tokens.v);
    free(prefix_tokens.v);
    *out = b.p;
    return b.len;
}

/* record_bm25_text (src/sampler/backends/bm25_backend.rs:503-532).  The record's sections arrive as 2 * n
 * strings: heading (has_heading[i] = 0 for None) and text of section i. */
size_t orc_record_bm25_text(const uint8_t* const* headings, const size_t* heading_len, const uint8_t* has_heading,
                            const uint8_t* const* texts, const size_t* text_len, size_t n_sections,
                            const uint8_t* record_id, size_t id_len, size_t max_tokens,
                            const uint8_t* newline, size_t newline_len, uint8_t** out) {
    buf_t b = { 0, 0, 0 };
    for (size_t i = 0; i < n_sections; i++) {                   /* :505 */
        str_t h = { headings[i], heading_len[i] }, t = { texts[i], text_len[i] };
        if (has_heading[i] && str_trim(h).len) {                /* :506-507 */
            buf_push(&b, h.p, h.len);
            buf_push(&b, newline, newline_len);
        }
        if (str_trim(t).len) {                                  /* :512 */
            buf_push(&b, t.p, t.len);
            buf_push(&b, newline, newline_len);
        }
    }
    str_t all = { b.p, b.len };
    if (str_trim(all).len == 0) {                               /* :517-518 */
        b.len = 0;
        buf_push(&b, record_id, id_len);
        *out = b.p;
        return b.len;
    }
    if (max_tokens == 0) { *out = b.p; return b.len; }          /* :520-521 */
    strvec_t tokens = { 0, 0, 0 };
    split_whitespace(all, &tokens);                             /* :523 */
    if (tokens.n <= max_tokens) {                               /* :524-525 */
        free(tokens.v);
        *out = b.p;
        return b.len;
    }
    buf_t j = { 0, 0, 0 };
    push_joined(&j, &tokens, max_tokens);                       /* :527-531 */
    free(tokens.v);
    free(b.p);
    *out = j.p;
    return j.len;
}

/* BM25 query capping (bm25_backend.rs:219-225): the text itself when it has at most `limit` tokens, else
 * its first `limit` tokens joined by one space. */
size_t orc_cap_tokens(const uint8_t* text, size_t len, size_t limit, uint8_t** out) {
    str_t s = { text, len };
    strvec_t tokens = { 0, 0, 0 };
    split_whitespace(s, &tokens);
    buf_t b = { 0, 0, 0 };
    if (tokens.n <= limit) buf_push(&b, text, len);
    else push_joined(&b, &tokens, limit);
    free(tokens.v);
    *out = b.p;
    return b.len;
}
