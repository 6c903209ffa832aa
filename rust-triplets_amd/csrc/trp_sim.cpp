// Host simulator of the GPU algorithm (development/test aid, not part of the shipped library).
//
// Runs the per-block functions of trp_lane.h - the very code the HIP kernels execute per lane - and
// replaces the wave/workgroup/tile scans by sequential folds with the same monoid operators, tile by
// tile, so that the bitmask formulation (summaries, carries, plan, gate-3 solver) can be compared
// with the CPU oracle on machines without a GPU.  The interpretation of the finished plan (byte
// compaction, token table) is written here in the most direct way possible.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "trp_lane.h"

#define TRP_TABLE_QUAL static const
#include "unicode_tables.inc"

using namespace trp;

extern "C" {

struct sim_config {
    uint8_t denoise_enabled;
    uint8_t strip_markdown;
    float max_digit_ratio;
    uint32_t max_window_tokens;
    const uint32_t* overlaps;
    uint32_t n_overlaps;
    uint32_t summary_fallback_tokens;
};

struct sim_result {
    uint8_t* denoised;
    uint64_t* denoised_offsets;
    uint8_t* dropped;
    uint32_t* tok_start;
    uint32_t* tok_len;
    uint64_t* tok_offsets;
    ChunkDesc* chunks;
    uint64_t* chunk_offsets;
    uint64_t n_records, n_denoised, n_tokens, n_chunks;
};

int trp_sim_batch(const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const sim_config* scfg,
                  sim_result* out, int tile_blocks);
void trp_sim_free(sim_result* r);
int trp_sim_normalize(const uint8_t* bytes, const uint64_t* offsets, uint64_t R, sim_result* out, int tile_blocks);
int trp_sim_check_monoids(unsigned seed, int iters);
}

static DevCfg to_dev(const sim_config* c) {
    DevCfg d;
    memset(&d, 0, sizeof(d));
    d.denoise_enabled = c->denoise_enabled;
    d.strip_markdown = c->strip_markdown;
    d.max_digit_ratio = c->max_digit_ratio;
    d.max_window_tokens = c->max_window_tokens;
    d.n_overlaps = c->n_overlaps > MAX_OVERLAPS ? MAX_OVERLAPS : c->n_overlaps;
    d.summary_fallback_tokens = c->summary_fallback_tokens;
    for (u32 i = 0; i < d.n_overlaps; i++) d.overlaps[i] = c->overlaps[i];
    return d;
}

namespace {
struct PlanBits {
    std::vector<Plan>* plan;
    void set(u64 q, bool c, bool tok, int prefix) {   // prefix: 0 none, 1 ' ', 2 '\n'
        Plan& p = (*plan)[q / 64];
        u64 b = bit((int)(q % 64));
        if (c) p.C |= b;
        if (tok) p.TOK |= b;
        if (prefix == 1) p.PSP |= b;
        if (prefix == 2) p.PNL |= b;
    }
    void clear_marker(u64 q) {
        Plan& p = (*plan)[q / 64];
        u64 b = ~bit((int)(q % 64));
        p.TOK &= b;
        p.PSP &= b;
        p.PNL &= b;
    }
};
}  // namespace

static int sim_run(const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const DevCfg& cfg, sim_result* out,
                   int tile_blocks);

int trp_sim_batch(const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const sim_config* scfg,
                  sim_result* out, int tile_blocks) {
    return sim_run(bytes, offsets, R, to_dev(scfg), out, tile_blocks);
}

// normalize_inline_whitespace of every record, with the reference's default chunking on the result
int trp_sim_normalize(const uint8_t* bytes, const uint64_t* offsets, uint64_t R, sim_result* out, int tile_blocks) {
    DevCfg d;
    memset(&d, 0, sizeof(d));
    d.denoise_enabled = 1;
    d.normalize = 1;
    d.max_digit_ratio = 0.35f;
    d.max_window_tokens = 1024;
    d.n_overlaps = 1;
    d.overlaps[0] = 64;
    d.summary_fallback_tokens = 512;
    return sim_run(bytes, offsets, R, d, out, tile_blocks);
}

static int sim_run(const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const DevCfg& cfg, sim_result* out,
                   int tile_blocks) {
    UniTables ut = {trp_alpha_stage1, trp_alpha_stage2, TRP_ALPHA_STAGE1_LEN, nullptr};
    const u64 N = offsets[R];
    const u64 nB = (N + 63) / 64;
    if (tile_blocks <= 0) tile_blocks = TILE_BLOCKS;
    std::vector<u8> pad(N + 2 * HALO + 64, 0);
    if (N) memcpy(pad.data() + HALO, bytes, N);
    const u8* base = pad.data() + HALO;
    std::vector<u64> RSw(nB + 1, 0);
    for (u64 r = 0; r < R; r++)
        if (offsets[r] < N) RSw[offsets[r] / 64] |= bit((int)(offsets[r] % 64));
    auto rs_at = [&](u64 q) -> bool { return q < N && ((RSw[q / 64] >> (q % 64)) & 1); };

    std::vector<Plan> plan(nB + 1, Plan{0, 0, 0, 0});
    std::vector<Cls> cls(nB);
    std::vector<u64> Vm(nB);
    SeqNeighbours nb;   // groups of 64 blocks, like the lanes of a wavefront
    for (u64 b = 0; b < nB; b++) {
        u64 p = b * 64;
        int nvalid = (int)((N - p) < 64 ? (N - p) : 64);
        nb.first = b % 64 == 0;
        nb.last = b % 64 == 63 || b + 1 == nB;
        cls[b] = classify_block_fast(PlainBlk{base + p}, nvalid, ut, nb);
        {   // the bit-sliced classifier must agree with the byte-by-byte one
            Cls ref = classify_block(PlainBlk{base + p}, nvalid, ut);
            if (ref.ws != cls[b].ws || ref.nl != cls[b].nl || ref.cr != cls[b].cr || ref.pipe != cls[b].pipe ||
                ref.dig != cls[b].dig || ref.alp != cls[b].alp)
                return -100;
        }
        Vm[b] = below(nvalid);
    }

    if (cfg.denoise_enabled) {
        // the digit-ratio test through its table, as on the device
        uint16_t thr1[RATIO_TAB];
        ratio_table_fill(cfg.max_digit_ratio, thr1);
        const RatioTab rt = {cfg.max_digit_ratio, thr1};
        std::vector<Derived> der(nB);
        std::vector<J> jl(nB);
        std::vector<u64> spm(nB, 0);
        for (u64 b = 0; b < nB; b++) {
            u64 p = b * 64;
            bool next_nl = (p + 64 < N) && base[p + 64] == '\n' && !rs_at(p + 64);
            bool pt = prev_is_term(p > 0, p > 0 ? base[p - 1] : 0, true, base[p], rs_at(p));
            if (cfg.normalize) {
                u64 space = 0;
                Cls zw = {classify_ws_fast(PlainBlk{base + p}, (int)(N - p < 64 ? N - p : 64), &space), 0, 0, 0, 0, 0, 0};
                cls[b] = normalize_cls(zw, Vm[b]);
                spm[b] = space;
                next_nl = pt = false;
            }
            der[b] = derive(cls[b], Vm[b], RSw[b], next_nl, pt);
            jl[b] = lane_J(cls[b], der[b]);
        }
        // tile summaries, then per-lane prefixes/suffixes through the tile level (as on the device)
        u64 nT = (nB + tile_blocks - 1) / tile_blocks;
        std::vector<J> SA(nT);
        for (u64 t = 0; t < nT; t++) {
            J acc = j_identity();
            for (u64 b = t * tile_blocks; b < nB && b < (t + 1) * tile_blocks; b++) acc = j_combine(acc, jl[b]);
            SA[t] = acc;
        }
        for (u64 t = 0; t < nT; t++) {
            // As on the device (k_seed): first try to settle both sides from windows of blocks next to the
            // tile (12 before, 4 after, moved outwards up to 16 times); a window that does not reach the
            // record start is enough once it holds a line start with a letter before it.
            J P = j_identity();
            S X = s_identity();
            const long long tb0 = (long long)(t * tile_blocks), tb1 = tb0 + (long long)tile_blocks;
            bool back_ok = t == 0, ahead_ok = cfg.normalize || (u64)tb1 >= nB;
            for (int round = 0; round < 16 && !(back_ok && ahead_ok); round++) {
                if (!back_ok) {
                    long long lo = tb0 - 12LL * (round + 1), hi = tb0 - 12LL * round;
                    J Wj = j_identity();
                    for (long long b = lo < 0 ? 0 : lo; b < hi; b++) Wj = j_combine(Wj, jl[b]);
                    P = j_combine(Wj, P);
                    bool settled = cfg.normalize ? (P.g & (G_RS | G_AH)) != 0
                                                 : (P.g & G_RS) != 0 || ((P.g & G_LS) && (P.g & G_C0));
                    back_ok = settled || lo <= 0;
                }
                if (!ahead_ok) {
                    long long lo = tb1 + 4LL * round, hi = tb1 + 4LL * (round + 1);
                    J Wj = j_identity();
                    for (long long b = lo; b < hi && (u64)b < nB; b++) Wj = j_combine(Wj, jl[b]);
                    X = s_combine(X, s_from_J(Wj));
                    ahead_ok = X.has_ls != 0 || (u64)hi >= nB;
                }
            }
            if (!(back_ok && ahead_ok)) {
                // look-back to the start of the record that covers the first byte of the tile
                P = j_identity();
                for (u64 u = t; u-- > 0;) {
                    P = j_combine(SA[u], P);
                    if (P.g & G_RS) break;
                }
                // look-ahead to the next line start
                X = s_identity();
                for (u64 u = t + 1; u < nT; u++) {
                    X = s_combine(X, s_from_J(SA[u]));
                    if (X.has_ls) break;
                }
            }
            u64 b0 = t * tile_blocks, b1 = (t + 1) * tile_blocks < nB ? (t + 1) * tile_blocks : nB;
            std::vector<S> suf(b1 - b0);
            S acc = X;
            for (u64 b = b1; b-- > b0;) {
                suf[b - b0] = acc;
                acc = s_combine(s_from_J(jl[b]), acc);
            }
            J pre = P;
            for (u64 b = b0; b < b1; b++) {
                bool next_rs = ((b + 1) * 64 >= N) || rs_at((b + 1) * 64);
                LaneCarry k = carry_from(pre, suf[b - b0], next_rs);
                if (cfg.normalize) {
                    u64 p = b * 64;
                    bool prev_tok = p > 0 && !rs_at(p) && !byte_is_ws(PlainBlk{base + p}, -1);
                    plan[b] = plan_normalize(cls[b], der[b], k.ar_in, prev_tok, spm[b]);
                } else {
                    plan[b] = plan_block(cls[b], der[b], k, cfg, rt);
                }
                pre = j_combine(pre, jl[b]);
            }
        }
        // pending digit-heavy lines (markers): solve serially
        ByteSrc src = make_src(base, N);
        PlanBits pb = {&plan};
        for (u64 b = 0; b < nB; b++) {
            u64 marks = plan[b].TOK & ~plan[b].C;
            while (marks) {
                int j = ctz(marks);
                marks &= marks - 1;
                u64 p = b * 64 + j;
                bool table = (plan[b].PSP >> j) & 1, rab = (plan[b].PNL >> j) & 1;
                // record end
                u64 lo = 0, hi = R;   // last r with offsets[r] <= p
                while (lo < hi) {
                    u64 mid = (lo + hi) / 2;
                    if (offsets[mid + 1] <= p) lo = mid + 1; else hi = mid;
                }
                u64 rec_end = offsets[lo + 1];
                u64 e = p;
                while (e < rec_end && base[e] != '\n' && base[e] != '\r') e++;
                pb.clear_marker(p);
                bool first = true;
                if (e > p && e - p <= 64u * G3_WORDS) {   // the bit-mask solvers, as on the device
                    const u64 w0 = p >> 6;
                    auto masks = [&](u32 w) -> G3Src {
                        G3Src r = {0, 0, 0};
                        const u64 wb = w0 + w;
                        if (wb < nB) {
                            r.x = ~(cls[wb].ws | (table ? cls[wb].pipe : 0ull));
                            r.dg = cls[wb].dig;
                            r.al = cls[wb].alp;
                        }
                        return r;
                    };
                    u64 Kw[G3_WORDS + 1] = {0, 0, 0, 0, 0}, KSw[G3_WORDS + 1] = {0, 0, 0, 0, 0};
                    const u32 s0 = (u32)(p & 63), e0 = (u32)(e - w0 * 64);
                    if (e - p <= 64u) {   // one-word masks (the device's list of short lines)
                        u64 K1[2], S1[2];
                        g3_solve_shifted<1>(masks, s0, e0, rt, K1, S1);
                        for (int i = 0; i <= 1; i++) { Kw[i] = K1[i]; KSw[i] = S1[i]; }
                    } else if (e - p <= 64u * G3_SHORT) {
                        u64 K2[G3_SHORT + 1], S2[G3_SHORT + 1];
                        g3_solve_shifted<G3_SHORT>(masks, s0, e0, rt, K2, S2);
                        for (int i = 0; i <= G3_SHORT; i++) { Kw[i] = K2[i]; KSw[i] = S2[i]; }
                    } else {
                        g3_solve_shifted<G3_WORDS>(masks, s0, e0, rt, Kw, KSw);
                    }
                    for (u64 q = p; q < e; q++) {
                        const u64 i = q - w0 * 64;
                        if (!((Kw[i >> 6] >> (i & 63)) & 1)) continue;
                        const bool start = (KSw[i >> 6] >> (i & 63)) & 1;
                        pb.set(q, true, true, start ? (first ? (rab ? 2 : 0) : 1) : 0);
                        if (start) first = false;
                    }
                    continue;
                }
                u64 K = g3_solve_K(src, p, e, table, cfg.max_digit_ratio, ut);
                g3_walk(src, p, e, table, ut,
                        [&](const TokInfo& t, u64 dist) {
                            if (dist > K) return;
                            for (u64 q = t.ts; q < t.te; q++)
                                pb.set(q, true, true, q == t.ts ? (first ? (rab ? 2 : 0) : 1) : 0);
                            first = false;
                        });
            }
        }
    } else {
        for (u64 b = 0; b < nB; b++) {   // whitespace from the mask-only classifier, as on the device
            Cls cw = cls[b];
            const u64 p = b * 64;
            cw.ws = classify_ws_fast(PlainBlk{base + p}, (int)(N - p < 64 ? N - p : 64));
            plan[b] = plan_passthrough(cw, Vm[b]);
        }
    }

    // interpret the plan
    std::vector<u8> den;
    std::vector<u32> ts, tl;
    std::vector<u64> doff(R + 1), toff(R + 1);
    u64 r = 0;
    bool prev_tok = false;
    u64 cur_start_out = 0, cur_start_in = 0;
    for (u64 q = 0; q <= N; q++) {
        bool c = false, tok = false;
        int j = (int)(q % 64);
        if (q < N) {
            const Plan& p = plan[q / 64];
            c = (p.C >> j) & 1;
            tok = c && ((p.TOK >> j) & 1);
        }
        if (prev_tok && (q == N || !tok || rs_at(q))) {   // the open token ended at q-1
            ts.push_back((u32)cur_start_out);
            tl.push_back((u32)(q - cur_start_in));
            prev_tok = false;
        }
        while (r < R && offsets[r] == q) {
            doff[r] = cfg.denoise_enabled ? den.size() : q;
            toff[r] = ts.size();
            r++;
        }
        if (q == N) break;
        if (c) {
            const Plan& p = plan[q / 64];
            if (cfg.denoise_enabled) {
                if ((p.PSP >> j) & 1) den.push_back(' ');
                else if ((p.PNL >> j) & 1) den.push_back('\n');
            }
            if (tok && !prev_tok) {
                prev_tok = true;
                cur_start_in = q;
                cur_start_out = cfg.denoise_enabled ? den.size() : q;
            }
            if (cfg.denoise_enabled) den.push_back(base[q]);
        }
    }
    while (r < R) { doff[r] = cfg.denoise_enabled ? den.size() : N; toff[r] = ts.size(); r++; }
    doff[R] = cfg.denoise_enabled ? den.size() : N;
    toff[R] = ts.size();

    memset(out, 0, sizeof(*out));
    out->n_records = R;
    u64 D = cfg.denoise_enabled ? den.size() : N;
    out->n_denoised = D;
    out->n_tokens = ts.size();
    out->denoised = (u8*)malloc(D ? D : 1);
    if (D) memcpy(out->denoised, cfg.denoise_enabled ? den.data() : bytes, D);
    out->denoised_offsets = (u64*)malloc((R + 1) * sizeof(u64));
    out->tok_offsets = (u64*)malloc((R + 1) * sizeof(u64));
    out->chunk_offsets = (u64*)malloc((R + 1) * sizeof(u64));
    out->dropped = (u8*)malloc(R ? R : 1);
    memcpy(out->denoised_offsets, doff.data(), (R + 1) * sizeof(u64));
    memcpy(out->tok_offsets, toff.data(), (R + 1) * sizeof(u64));
    out->tok_start = (u32*)malloc((ts.size() ? ts.size() : 1) * sizeof(u32));
    out->tok_len = (u32*)malloc((ts.size() ? ts.size() : 1) * sizeof(u32));
    if (!ts.empty()) {
        memcpy(out->tok_start, ts.data(), ts.size() * sizeof(u32));
        memcpy(out->tok_len, tl.data(), tl.size() * sizeof(u32));
    }
    u64 C = 0;
    for (u64 i = 0; i < R; i++) {
        out->chunk_offsets[i] = C;
        C += chunk_count(toff[i + 1] - toff[i], cfg);
        out->dropped[i] = cfg.denoise_enabled && !cfg.normalize && doff[i + 1] == doff[i];
    }
    out->chunk_offsets[R] = C;
    out->n_chunks = C;
    out->chunks = (ChunkDesc*)malloc((C ? C : 1) * sizeof(ChunkDesc));
    for (u64 i = 0; i < R; i++) {
        u64 T = toff[i + 1] - toff[i];
        u64 n = out->chunk_offsets[i + 1] - out->chunk_offsets[i];
        for (u64 j = 0; j < n; j++) out->chunks[out->chunk_offsets[i] + j] = chunk_at(T, j, (u32)i, cfg);
    }
    return 0;
}

void trp_sim_free(sim_result* r) {
    free(r->denoised); free(r->denoised_offsets); free(r->dropped); free(r->tok_start);
    free(r->tok_len); free(r->tok_offsets); free(r->chunks); free(r->chunk_offsets);
    memset(r, 0, sizeof(*r));
}

// The range shortcuts of cp_is_alpha_high must agree with the table: returns the number of code
// points where the function differs from a plain table lookup.
extern "C" int trp_sim_check_alpha_shortcut() {
    int bad = 0;
    for (u32 cp = 0x80; cp < 0x110000; cp++) {
        u32 blk = cp >> 8;
        bool t = false;
        if (blk < TRP_ALPHA_STAGE1_LEN) {
            u32 row = trp_alpha_stage1[blk], o = cp & 255u;
            t = (trp_alpha_stage2[row * 8 + (o >> 5)] >> (o & 31u)) & 1u;
        }
        UniTables ut = {trp_alpha_stage1, trp_alpha_stage2, TRP_ALPHA_STAGE1_LEN, nullptr};
        if (t != cp_is_alpha_high(cp, ut)) bad++;
        u32 low[64];
        for (int i = 0; i < 64; i++) low[i] = trp_alpha_stage2[trp_alpha_stage1[i >> 3] * 8 + (i & 7)];
        UniTables ul = {trp_alpha_stage1, trp_alpha_stage2, TRP_ALPHA_STAGE1_LEN, low};
        if (t != cp_is_alpha_high(cp, ul)) bad++;
    }
    return bad;
}

// classify_block_fast (bit planes, range rules for the common scripts) against classify_block (one byte
// at a time, table only) for every code point at several positions of a block, including positions where
// the sequence crosses the end of the block.  Returns the number of differing blocks.
extern "C" int trp_sim_check_classify_codepoints() {
    UniTables ut = {trp_alpha_stage1, trp_alpha_stage2, TRP_ALPHA_STAGE1_LEN, nullptr};
    int bad = 0;
    u8 buf[8 + 128 + 8];
    static const int positions[] = {0, 33, 58, 60, 61, 62, 63};
    for (u32 cp = 0x80; cp < 0x110000; cp++) {
        if (cp >= 0xD800 && cp < 0xE000) continue;
        if (cp >= 0x40000 && cp % 7u) continue;   // nothing is assigned there but the tag and selector blocks
        u8 enc[4];
        int n;
        if (cp < 0x800) { enc[0] = 0xC0 | (cp >> 6); enc[1] = 0x80 | (cp & 63); n = 2; }
        else if (cp < 0x10000) { enc[0] = 0xE0 | (cp >> 12); enc[1] = 0x80 | ((cp >> 6) & 63); enc[2] = 0x80 | (cp & 63); n = 3; }
        else { enc[0] = 0xF0 | (cp >> 18); enc[1] = 0x80 | ((cp >> 12) & 63); enc[2] = 0x80 | ((cp >> 6) & 63); enc[3] = 0x80 | (cp & 63); n = 4; }
        for (int pos : positions) {
            for (int fill = 0; fill < 2; fill++) {
                // surroundings: ASCII, or copies of the character itself (adjacent multi-byte sequences)
                for (int i = 0; i < (int)sizeof(buf); i++) buf[i] = fill ? enc[(i + 4 * n - (8 + pos) % n) % n] : (u8)('a' + i % 3);
                for (int i = 0; i < n; i++) buf[8 + pos + i] = enc[i];
                const PlainBlk blk = {buf + 8};
                const Cls f = classify_block_fast(blk, 64, ut), r = classify_block(blk, 64, ut);
                if (f.ws != r.ws || f.alp != r.alp || f.nl != r.nl || f.cr != r.cr || f.pipe != r.pipe || f.dig != r.dig)
                    bad++;
                if (classify_ws_fast(blk, 64) != r.ws) bad++;
                // the same two blocks classified as neighbours of one group
                SeqNeighbours nb;
                for (int k = 0; k < 2; k++) {
                    const PlainBlk bk = {buf + 8 + 64 * k};
                    nb.first = k == 0;
                    nb.last = k == 1;
                    const Cls g = classify_block_fast(bk, 64, ut, nb), h = classify_block(bk, 64, ut);
                    if (g.ws != h.ws || g.alp != h.alp || g.nl != h.nl || g.cr != h.cr || g.pipe != h.pipe || g.dig != h.dig)
                        bad++;
                }
            }
        }
    }
    return bad;
}

// Brute-force associativity check of the two monoids on random elements; returns the number of
// violations.
static u32 rnd(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static LineSum rnd_ls(unsigned* s) {
    LineSum l;
    l.d = rnd(s) % 4; l.a = rnd(s) % 3; l.p = rnd(s) % 3;
    u32 first = rnd(s) % 3, last = first ? 1 + rnd(s) % 2 : 0;
    l.f = first | (last << 2) | ((first == 2 || last == 2 || (first && rnd(s) % 2)) ? 16u : 0u);
    return l;
}
static J rnd_J(unsigned* s) {
    J j;
    j.head = rnd_ls(s);
    j.g = 0;
    if (rnd(s) % 2) { j.g |= G_LS; j.tail = rnd_ls(s); j.g |= (rnd(s) % 2) ? G_C0 : 0; j.g |= (rnd(s) % 2) ? G_C1 : 0; }
    else j.tail = ls_empty();
    if (rnd(s) % 2) { j.g |= G_RS | G_LS; j.g &= ~G_C1; if (rnd(s) % 2) j.g |= G_AH; if (rnd(s) % 2) j.g |= G_AT; }
    else if (rnd(s) % 2) j.g |= G_AH | G_AT;
    return j;
}
static bool eq_ls(const LineSum& a, const LineSum& b) { return a.d == b.d && a.a == b.a && a.p == b.p && a.f == b.f; }
int trp_sim_check_monoids(unsigned seed, int iters) {
    int bad = 0;
    for (int i = 0; i < iters; i++) {
        J a = rnd_J(&seed), b = rnd_J(&seed), c = rnd_J(&seed);
        J l = j_combine(j_combine(a, b), c), r = j_combine(a, j_combine(b, c));
        if (!(eq_ls(l.head, r.head) && eq_ls(l.tail, r.tail) && l.g == r.g)) bad++;
        S x = s_from_J(a), y = s_from_J(b), z = s_from_J(c);
        S sl = s_combine(s_combine(x, y), z), sr = s_combine(x, s_combine(y, z));
        if (!(eq_ls(sl.head, sr.head) && sl.has_ls == sr.has_ls)) bad++;
        J ia = j_combine(j_identity(), a), ai = j_combine(a, j_identity());
        if (!(eq_ls(ia.head, a.head) && eq_ls(ia.tail, a.tail) && ia.g == a.g)) bad++;
        if (!(eq_ls(ai.head, a.head) && eq_ls(ai.tail, a.tail) && ai.g == a.g)) bad++;
    }
    return bad;
}
