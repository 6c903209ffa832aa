// Per-block ("lane") logic of the denoise -> tokenise -> chunk path.
//
// The packed corpus is cut into 64-byte blocks; one GPU lane owns one block and represents
// every per-byte property as a 64-bit mask (bit j = byte j of the block).  A wavefront
// therefore covers 4 KiB and a 256-thread workgroup a 16 KiB tile.  Everything that crosses a
// block boundary travels as a small associative summary (monoid) that is combined by
// wave/workgroup scans on the device and by a plain sequential fold in the host simulator
// (trp_sim.cpp), which runs exactly these functions so the algorithm can be checked against the
// CPU oracle without a GPU.
//
// Reference semantics restated here (paths relative to the reference checkout):
//   src/preprocessor/backends/denoiser_preprocessor.rs:21-268  (gates 1-3, table stripping)
//   src/tokenizer.rs:55-65                                      (split_whitespace)
//   src/chunking/sliding_window.rs:40-115                       (window arithmetic)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TRP_HD __host__ __device__ inline
#else
#define TRP_HD inline
#endif
#if defined(__clang__)
#define TRP_UNROLL _Pragma("unroll")
#else
#define TRP_UNROLL
#endif

namespace trp {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

constexpr int BLOCK = 64;          // bytes per lane
constexpr int TILE_BLOCKS = 256;   // lanes per workgroup
constexpr int TILE = BLOCK * TILE_BLOCKS;
constexpr int HALO = 16;           // staged bytes before and after a tile (UTF-8 look-around)
constexpr int MAX_OVERLAPS = 16;

// ---- character classes (bits of the per-byte class code) -------------------------------------
enum : u32 { C_WS = 1, C_NL = 2, C_CR = 4, C_PIPE = 8, C_DIGIT = 16, C_ALPHA = 32, C_HIGH = 64 };

// Class code of an ASCII byte; bytes >= 0x80 only get C_HIGH (resolved by decoding).
TRP_HD u32 ascii_class(u32 b) {
    u32 c = 0;
    if (b >= 0x80) return C_HIGH;
    if (b == 0x20 || (b - 9u) < 5u) c |= C_WS;
    if (b == 0x0A) c |= C_NL;
    if (b == 0x0D) c |= C_CR;
    if (b == 0x7C) c |= C_PIPE;
    if ((b - 0x30u) < 10u) c |= C_DIGIT;
    if (((b | 0x20u) - 0x61u) < 26u) c |= C_ALPHA;
    return c;
}

// Unicode White_Space beyond ASCII (char::is_whitespace).
TRP_HD bool cp_is_ws_high(u32 cp) {
    return cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A) ||
           cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}

struct UniTables {
    const u8* stage1;
    const u32* stage2;
    u32 stage1_len;
    const u32* low8;   // optional: the bitmap rows of code point blocks 0..7 in block order (64 words)
};

// Unicode Alphabetic (char::is_alphabetic) through the generated two-stage table.  The big uniform
// ranges (CJK ideographs, Hangul syllables: every code point Alphabetic) are answered without a
// table access; tests check this shortcut against the table for every code point.
TRP_HD bool cp_is_alpha_high(u32 cp, const UniTables& ut) {
    if ((cp >= 0x3400 && cp <= 0x4DBF) || (cp >= 0x4E00 && cp <= 0x9FFF) || (cp >= 0xAC00 && cp <= 0xD7A3))
        return true;
    u32 blk = cp >> 8;
    u32 o = cp & 255u;
    if (ut.low8 && blk < 8) return (ut.low8[blk * 8 + (o >> 5)] >> (o & 31u)) & 1u;
    if (blk >= ut.stage1_len) return false;
    u32 row = ut.stage1[blk];
    return (ut.stage2[row * 8 + (o >> 5)] >> (o & 31u)) & 1u;
}

// Block accessors.  A block is 64 bytes with at least 4 readable bytes before and after it (indices
// -4..67).  PlainBlk: the bytes are contiguous in memory.  PadBlk: the staged LDS image, where every
// 64-byte block is followed by 4 unused bytes so that lanes that touch the same byte index of their
// blocks fall into different LDS banks (block stride 68 bytes = 17 banks).
struct PlainBlk {
    const u8* p;
    TRP_HD u32 operator[](int j) const { return p[j]; }
    TRP_HD u32 word(int k) const {
        return (u32)p[4 * k] | ((u32)p[4 * k + 1] << 8) | ((u32)p[4 * k + 2] << 16) | ((u32)p[4 * k + 3] << 24);
    }
};
constexpr int PAD_STRIDE = 68;
struct PadBlk {
    const u8* p;
    TRP_HD u32 operator[](int j) const { return p[j + 4 * (j >> 6)]; }
    TRP_HD u32 word(int k) const { return *reinterpret_cast<const u32*>(p + 4 * k); }
};

// Decode the multi-byte sequence whose lead byte is q[at] (>= 0xC0).  q[at+1..at+3] must be
// readable.  Returns the sequence length (2..4) and the scalar value, or 0 when malformed.
template <class B>
TRP_HD int decode_lead(const B& q, int at, u32* cp) {
    u32 b0 = q[at];
    int n;
    u32 v;
    if (b0 < 0xE0) { n = 2; v = b0 & 0x1F; }
    else if (b0 < 0xF0) { n = 3; v = b0 & 0x0F; }
    else { n = 4; v = b0 & 0x07; }
    for (int i = 1; i < n; i++) {
        u32 c = q[at + i];
        if ((c & 0xC0) != 0x80) return 0;
        v = (v << 6) | (c & 0x3F);
    }
    *cp = v;
    return n;
}

// ---- bit helpers ---------------------------------------------------------------------------------
TRP_HD int popc(u64 x) { return __builtin_popcountll(x); }
TRP_HD int ctz(u64 x) { return __builtin_ctzll(x); }          // x != 0
TRP_HD int top(u64 x) { return 63 - __builtin_clzll(x); }     // x != 0: index of highest set bit
TRP_HD u64 below(int j) { return j >= 64 ? ~0ull : ((1ull << j) - 1ull); }   // bits < j
TRP_HD u64 bit(int j) { return 1ull << j; }
TRP_HD u64 brev(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#elif defined(__has_builtin) && __has_builtin(__builtin_bitreverse64)
    return __builtin_bitreverse64(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}

// ---- per-block class masks -----------------------------------------------------------------------
struct Cls {
    u64 ws, nl, cr, pipe, dig, alp;
    u64 sp;   // the byte is exactly 0x20
};

// Classify `nvalid` bytes starting at blk[0].  blk[-4..-1] and blk[64..67] must be readable
// (staged halo; zero where the corpus has no byte).  `ws` covers every byte of a whitespace
// character; `alp` is set on the lead byte of an alphabetic character only.
template <class B>
TRP_HD Cls classify_block(const B& blk, int nvalid, const UniTables& ut) {
    Cls c = {0, 0, 0, 0, 0, 0, 0};
    u64 high = 0;
    for (int j = 0; j < nvalid; j++) {
        u32 k = ascii_class(blk[j]);
        u64 b = 1ull << j;
        if (blk[j] == 0x20) c.sp |= b;
        if (k & C_WS) c.ws |= b;
        if (k & C_NL) c.nl |= b;
        if (k & C_CR) c.cr |= b;
        if (k & C_PIPE) c.pipe |= b;
        if (k & C_DIGIT) c.dig |= b;
        if (k & C_ALPHA) c.alp |= b;
        if (k & C_HIGH) high |= b;
    }
    while (high) {
        int j = ctz(high);
        high &= high - 1;
        u32 b0 = blk[j];
        u32 cp;
        if (b0 >= 0xC0) {
            if (decode_lead(blk, j, &cp)) {
                if (cp_is_ws_high(cp)) c.ws |= bit(j);
                else if (cp_is_alpha_high(cp, ut)) c.alp |= bit(j);
            }
        } else {
            // continuation byte: it is whitespace iff its character is
            for (int k = 1; k <= 3; k++) {
                u32 l = blk[j - k];
                if (l >= 0xC0) {
                    int n = decode_lead(blk, j - k, &cp);
                    if (n > k && cp_is_ws_high(cp)) c.ws |= bit(j);
                    break;
                }
                if (l < 0x80) break;
            }
        }
    }
    return c;
}

// ---- bit-sliced classification ------------------------------------------------------------------
// The 64 bytes of a block are transposed into the eight bit planes of the raw byte values (8x8 bit
// transposes of byte groups + byte gathers), after which every ASCII class is a short boolean formula
// evaluated for all 64 bytes at once.  Bytes >= 0x80 are resolved one by one as in classify_block.
TRP_HD u32 load_u32(const u8* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *reinterpret_cast<const u32*>(p);
#else
    return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
#endif
}
// (a & m) | (b & ~m)
TRP_HD u32 bfi(u32 m, u32 a, u32 b) { return (a & m) | (b & ~m); }

// Byte select from the eight bytes {hi:lo}: result byte i is byte sel.byte[i] of the pair, where 0..3
// name the bytes of lo and 4..7 those of hi.
TRP_HD u32 byte_perm(u32 hi, u32 lo, u32 sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    u64 pair = ((u64)hi << 32) | lo;
    u32 r = 0;
    for (int i = 0; i < 4; i++) r |= (u32)((pair >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
    return r;
#endif
}

// Transpose the 64 bytes held in w[16] (byte i of the block = byte i%4 of w[i/4]) into the eight bit
// planes of the byte values: bit i of plane[c] = bit c of byte i.
// Index algebra: a bit of the input sits at word (i2,i3,i4,i5), in-word position (c0,c1,c2,i0,i1).
//   1. 4x4 byte transposes among the words that differ in (i3,i4) exchange the byte position with
//      those two word-index bits: word (i2,i0,i1,i5), position (c0,c1,c2,i3,i4).
//   2. three rounds of bit exchanges between word pairs swap c0<->i0, c1<->i1, c2<->i2:
//      word (c2,c0,c1,i5), position (i0,i1,i2,i3,i4) - the natural bit order of a plane half.
TRP_HD void transpose_block(u32 (&w)[16], u64 (&plane)[8]) {
TRP_UNROLL
    for (int base = 0; base < 16; base += 8) {
TRP_UNROLL
        for (int k0 = 0; k0 < 2; k0++) {
            const u32 A = w[base + k0], B = w[base + k0 + 2], C2 = w[base + k0 + 4], D = w[base + k0 + 6];
            const u32 t0 = byte_perm(B, A, 0x05010400u);    // A0 B0 A1 B1
            const u32 t1 = byte_perm(B, A, 0x07030602u);    // A2 B2 A3 B3
            const u32 t2 = byte_perm(D, C2, 0x05010400u);   // C0 D0 C1 D1
            const u32 t3 = byte_perm(D, C2, 0x07030602u);   // C2 D2 C3 D3
            w[base + k0] = byte_perm(t2, t0, 0x05040100u);      // A0 B0 C0 D0
            w[base + k0 + 2] = byte_perm(t2, t0, 0x07060302u);  // A1 B1 C1 D1
            w[base + k0 + 4] = byte_perm(t3, t1, 0x05040100u);  // A2 B2 C2 D2
            w[base + k0 + 6] = byte_perm(t3, t1, 0x07060302u);  // A3 B3 C3 D3
        }
    }
TRP_UNROLL
    for (int k = 0; k < 16; k++) {
        if (k & 2) continue;   // c0 <-> word bit 1
        const u32 a = w[k], b = w[k | 2];
        w[k] = bfi(0x55555555u, a, b << 1);
        w[k | 2] = bfi(0x55555555u, a >> 1, b);
    }
TRP_UNROLL
    for (int k = 0; k < 16; k++) {
        if (k & 4) continue;   // c1 <-> word bit 2
        const u32 a = w[k], b = w[k | 4];
        w[k] = bfi(0x33333333u, a, b << 2);
        w[k | 4] = bfi(0x33333333u, a >> 2, b);
    }
TRP_UNROLL
    for (int k = 0; k < 16; k++) {
        if (k & 1) continue;   // c2 <-> word bit 0
        const u32 a = w[k], b = w[k | 1];
        w[k] = bfi(0x0F0F0F0Fu, a, b << 4);
        w[k | 1] = bfi(0x0F0F0F0Fu, a >> 4, b);
    }
TRP_UNROLL
    for (int c = 0; c < 8; c++) {
        const int k = ((c >> 2) & 1) | ((c & 1) << 1) | (((c >> 1) & 1) << 2);
        plane[c] = (u64)w[k] | ((u64)w[k | 8] << 32);
    }
}

// Multi-byte characters of the common scripts, classified for half a block at once from the bit planes
// (no byte access, no table).  p[k]: bit plane k of the 32 bytes of the half; q[k]: the same plane of the
// 32 bytes that follow (sequences may run over).  `lead`: lead bytes (11xxxxxx) of the half whose sequence
// ends inside the block.  ORs the alphabetic ones into *alp and returns the ones it has settled
// (alphabetic, or neither alphabetic nor whitespace).  Everything else - other scripts, whitespace
// characters, sequences that cross the block edge, malformed bytes - stays for the per-character path.
// Each rule is a range of code points that is uniformly Alphabetic (or uniformly neither class) in the
// table; a test compares the two paths for every code point.
TRP_HD u32 funnel_shr(u32 hi, u32 lo, int n) { return (u32)((((u64)hi << 32) | lo) >> n); }
// Bit-sliced comparison of 32 six-bit values (plane k in b[k]) with a constant: bit i of the result says
// whether value i is >= c.  From the lowest bit up: a one in c needs a one in the value and the lower part
// to hold, a zero in c is satisfied by a one in the value or by the lower part.
TRP_HD u32 planes_ge(const u32 (&b)[6], u32 c) {
    if (c == 0) return ~0u;
    if (c > 63) return 0u;
    u32 r = ~0u;
    TRP_UNROLL
    for (int k = 0; k < 6; k++) r = ((c >> k) & 1u) ? (b[k] & r) : (b[k] | r);
    return r;
}
TRP_HD u32 planes_in(const u32 (&b)[6], u32 lo, u32 hi) { return planes_ge(b, lo) & ~planes_ge(b, hi + 1u); }
// *ws2 / *ws3 receive the lead bytes of the two- and three-byte White_Space characters among the settled ones.
// The planes that follow a half: a full word per plane (the other half of the block) ...
struct PlanesAfter {
    const u32* q;
    TRP_HD u32 operator()(int k) const { return q[k]; }
};
// ... or only the first three bytes of the next block, plane k in bits 3k .. 3k+2 of one word.
struct PlanesAfter3 {
    u32 packed;
    TRP_HD u32 operator()(int k) const { return (packed >> (3 * k)) & 7u; }
};
TRP_HD u32 pack_planes3(const u32 (&lo)[8]) {
    u32 r = lo[0] & 7u;
    TRP_UNROLL
    for (int k = 1; k < 8; k++) r |= (lo[k] & 7u) << (3 * k);
    return r;
}
template <class Q>
TRP_HD u32 classify_leads_planes(const u32 (&p)[8], const Q& q, u32 lead, u32* alp, u32* ws2, u32* ws3) {
    if (!lead) return 0;
    const u32 cont = p[7] & ~p[6], contq = q(7) & ~q(6);
    const u32 v1 = funnel_shr(contq, cont, 1), v2 = funnel_shr(contq, cont, 2), v3 = funnel_shr(contq, cont, 3);
    const u32 l0 = p[0], l1 = p[1], l2 = p[2], l3 = p[3], l4 = p[4];
    // low six bits of the second (x) and third (y) byte, aligned to the lead byte
    const u32 x0 = funnel_shr(q(0), p[0], 1), x1 = funnel_shr(q(1), p[1], 1), x2 = funnel_shr(q(2), p[2], 1),
              x3 = funnel_shr(q(3), p[3], 1), x4 = funnel_shr(q(4), p[4], 1), x5 = funnel_shr(q(5), p[5], 1);
    const u32 x_ge16 = x5 | x4;
    u32 settled = 0, alpha = 0;
    const u32 L2 = lead & ~p[5] & v1;   // 110xxxxx 10xxxxxx
    if (L2) {
        const u32 lo = L2 & ~l4, hi = L2 & l4 & ~l3;                                // leads C0-CF, D0-D7
        const u32 C3 = lo & ~l3 & ~l2 & l1 & l0;                                    // U+00C0-00FF
        const u32 C4_CA = lo & ((~l3 & l2) | (l3 & ~l2 & ~(l1 & l0)));              // U+0100-02BF
        const u32 CE = lo & l3 & l2 & l1 & ~l0, CF = lo & l3 & l2 & l1 & l0;        // U+0380-03BF, 03C0-03FF
        const u32 D013 = hi & ~l2 & (~l1 | l0);                                     // U+0400-047F, 04C0-04FF
        const u32 D2 = hi & ~l2 & l1 & ~l0;                                         // U+0480-04BF
        const u32 D4 = hi & l2 & ~l1 & ~l0;                                         // U+0500-053F
        const u32 x17 = x4 & ~x3 & x2 & x1 & x0;                                    // U+00D7, U+00F7
        const u32 x22 = x5 & ~x4 & ~x3 & ~x2 & x1 & ~x0;                            // U+03A2
        const u32 x36 = x5 & x4 & ~x3 & x2 & x1 & ~x0;                              // U+03F6
        const u32 x30 = x5 & x4 & ~x3 & ~x2 & ~x1 & ~x0;                            // U+0530
        settled |= C3 | C4_CA | (CE & x_ge16) | CF | D013 | (D2 & x_ge16) | D4;
        alpha |= (C3 & ~x17) | C4_CA | (CE & x_ge16 & ~x22) | (CF & ~x36) | D013 | (D2 & x_ge16) | (D4 & ~x30);
        const u32 C2 = lo & ~l3 & ~l2 & l1 & ~l0;                                   // U+0080-00BF
        if (C2) {
            // letters U+00AA, 00B5, 00BA; whitespace U+0085, 00A0; the rest of the row is in neither class
            const u32 x2A = x5 & ~x4 & x3 & ~x2 & x1 & ~x0, x35 = x5 & x4 & ~x3 & x2 & ~x1 & x0,
                      x3A = x5 & x4 & x3 & ~x2 & x1 & ~x0;
            const u32 x05 = ~(x5 | x4 | x3 | x1) & x2 & x0, x20 = x5 & ~(x4 | x3 | x2 | x1 | x0);
            settled |= C2;
            alpha |= C2 & (x2A | x35 | x3A);
            *ws2 |= C2 & (x05 | x20);
        }
        const u32 semitic = L2 & l4 & (l3 | (l2 & l1 & l0));                        // leads D7-DF
        if (semitic) {
            const u32 xb[6] = {x0, x1, x2, x3, x4, x5};
            const u32 D7 = hi & l2 & l1 & l0;                                       // U+05C0-05FF
            const u32 hi2 = L2 & l4 & l3 & ~l2;                                     // leads D8-DB: U+0600-06FF
            const u32 D8 = hi2 & ~l1 & ~l0, D9 = hi2 & ~l1 & l0, DA = hi2 & l1 & ~l0, DB = hi2 & l1 & l0;
            const u32 heb = D7 & planes_in(xb, 0x10, 0x2A);                         // U+05D0-05EA
            const u32 ar8 = D8 & x5;                                                // U+0620-063F
            const u32 ar9_not = planes_in(xb, 0x18, 0x18) | planes_in(xb, 0x20, 0x2D);   // U+0658, U+0660-066D
            const u32 arB = DB & ~planes_ge(xb, 0x14);                              // U+06C0-06D3
            settled |= heb | ar8 | D9 | DA | arB;
            alpha |= heb | ar8 | (D9 & ~ar9_not) | DA | arB;
        }
    }
    const u32 L3 = lead & p[5] & ~p[4] & v1 & v2;   // 1110xxxx 10xxxxxx 10xxxxxx
    const u32 L4 = lead & p[5] & p[4] & ~p[3] & v1 & v2 & v3;
    if (L3 | L4) {
        const u32 y0 = funnel_shr(q(0), p[0], 2), y1 = funnel_shr(q(1), p[1], 2), y2 = funnel_shr(q(2), p[2], 2),
                  y3 = funnel_shr(q(3), p[3], 2), y4 = funnel_shr(q(4), p[4], 2), y5 = funnel_shr(q(5), p[5], 2);
        const u32 E5_E9 = L3 & ((~l3 & l2 & (l1 | l0)) | (l3 & ~l2 & ~l1));         // U+5000-9FFF
        const u32 E4 = L3 & ~l3 & l2 & ~l1 & ~l0;                                   // U+4000-4FFF
        const u32 E3 = L3 & ~l3 & ~l2 & l1 & l0;                                    // U+3000-3FFF
        const u32 EB_EC = L3 & l3 & ((~l2 & l1 & l0) | (l2 & ~l1 & ~l0));           // U+B000-CFFF
        const u32 EA = L3 & l3 & ~l2 & l1 & ~l0;                                    // U+A000-AFFF
        const u32 ED = L3 & l3 & l2 & ~l1 & l0;                                     // U+D000-DFFF
        const u32 E2 = L3 & ~l3 & ~l2 & l1 & ~l0;                                   // U+2000-2FFF
        const u32 x37 = x5 & x4 & ~x3 & x2 & x1 & x0;                               // U+4DC0-4DFF
        const u32 x_lt1E = ~x5 & ~(x4 & x3 & x2 & x1);                              // U+D000-D77F
        const u32 x00 = ~(x5 | x4 | x3 | x2 | x1 | x0), x02 = ~(x5 | x4 | x3 | x2 | x0) & x1;
        const u32 y_ws = y5 & ~y4 & y3 & ((~y2 & ~y1) | (y2 & y1 & y0));            // U+2028, 2029, 202F
        // U+2010-203F without the three whitespace characters, U+20A0-20BF: punctuation and currency signs
        const u32 E2_none = E2 & ((x00 & (y5 | y4) & ~y_ws) | (x02 & y5));
        // U+2000-200F: eleven spaces (U+2000-200A), then format characters in neither class
        const u32 E2_00 = E2 & x00 & ~(y5 | y4);
        *ws3 |= (E2 & x00 & y_ws) | (E2_00 & ~(y3 & (y2 | (y1 & y0))));
        // U+1F300-1FFFF: pictographs, no letters
        const u32 F0_9F = L4 & ~l2 & ~l1 & ~l0 & ~x5 & x4 & x3 & x2 & x1 & x0 & (y5 | y4 | (y3 & y2));
        u32 a3 = E5_E9 | (E4 & ~x37) | (E3 & x_ge16) | EB_EC | (EA & x5 & x4) | (ED & x_lt1E);
        settled |= a3 | E4 | E2_none | (E2 & x00 & y_ws) | E2_00 | F0_9F;
        // kana (E3 81-83), Devanagari (E0 A4-A5), Thai (E0 B8-B9): every code point of these rows is either
        // alphabetic or in neither class
        const u32 E0 = L3 & ~(l3 | l2 | l1 | l0);
        const u32 E3_low = E3 & ~x_ge16 & ~(x3 | x2) & (x1 | x0);                   // second byte 81, 82, 83
        // U+3000-301F without U+3005-3007: the ideographic space, then punctuation in neither class
        const u32 E3_00 = E3 & x00 & ~y5 & ~(~y4 & ~y3 & y2 & (y1 | y0));
        settled |= E3_00;
        *ws3 |= E3_00 & ~(y4 | y3 | y2 | y1 | y0);
        if (E0 | E3_low) {
            const u32 yb[6] = {y0, y1, y2, y3, y4, y5};
            const u32 x_hi = x5 & ~x4, x_lo3 = ~x3 & x2 & ~x1;                      // second byte A4/A5: 10 010x
            const u32 A4 = E0 & x_hi & x_lo3 & ~x0, A5 = E0 & x_hi & x_lo3 & x0;    // U+0900-093F, 0940-097F
            const u32 xB8 = x5 & x4 & x3 & ~x2 & ~x1;                               // second byte B8/B9: 11 100x
            const u32 B8 = E0 & xB8 & ~x0, B9 = E0 & xB8 & x0;                      // U+0E00-0E3F, 0E40-0E7F
            const u32 K1 = E3_low & ~x1 & x0, K2 = E3_low & x1 & ~x0, K3 = E3_low & x1 & x0;
            const u32 a_A4 = ~planes_in(yb, 0x3C, 0x3C);
            const u32 a_A5 = ~planes_ge(yb, 0x0D) | planes_in(yb, 0x0E, 0x10) | planes_in(yb, 0x15, 0x23) | planes_ge(yb, 0x31);
            const u32 a_B8 = planes_in(yb, 0x01, 0x3A);
            const u32 a_B9 = ~planes_ge(yb, 0x07) | planes_in(yb, 0x0D, 0x0D);
            const u32 a_K1 = planes_ge(yb, 0x01);
            const u32 a_K2 = ~planes_ge(yb, 0x17) | planes_in(yb, 0x1D, 0x1F) | planes_ge(yb, 0x21);
            const u32 a_K3 = ~planes_in(yb, 0x3B, 0x3B);
            settled |= A4 | A5 | B8 | B9 | K1 | K2 | K3;
            a3 |= (A4 & a_A4) | (A5 & a_A5) | (B8 & a_B8) | (B9 & a_B9) | (K1 & a_K1) | (K2 & a_K2) | (K3 & a_K3);
        }
        alpha |= a3;
    }
    *alp |= alpha;
    return settled;
}

// What a block may learn from the blocks next to it while it is being classified.  The blocks of a group are
// classified together (the lanes of a wavefront) or one after the other in ascending order (the host simulator);
// a block at the edge of its group has no neighbour on that side and falls back on reading the bytes itself.
//   any(x)            x for any block of the group (every block of the group makes the calls below, or none)
//   has_next()        the next block belongs to the group; next_planes3(mine, blk): the bit planes of the first
//                     three bytes of the next block as pack_planes3() packs them (mine = this block's own)
//   has_prev()        the previous block belongs to the group; swap_spill(v) hands v to the next block and
//                     returns what the previous block handed over (0 without one)
//   leftover(...)     the group classifies the lead bytes its blocks could not settle by rule (every block of the
//                     group calls it, with its own lead bytes) and ORs the results into *ws / *alp; false: the
//                     group has no way to do so and every block decodes its own
struct NoNeighbours {
    TRP_HD bool any(bool x) const { return x; }
    TRP_HD bool has_next() const { return false; }
    TRP_HD bool has_prev() const { return false; }
    template <class B>
    TRP_HD u32 next_planes3(u32, const B&) const { return 0; }
    TRP_HD u32 swap_spill(u32) const { return 0; }
    template <class B>
    TRP_HD bool leftover(const B&, u64, u64, const UniTables&, u64*, u64*) const { return false; }
};
// Blocks classified one after the other in ascending order (host code): the planes of the next block are built
// from its first three bytes, the spill is kept from one call to the next.
struct SeqNeighbours {
    bool first = true, last = false;   // set by the caller before each block
    u32 kept = 0;
    bool any(bool) const { return true; }
    bool has_next() const { return !last; }
    bool has_prev() const { return !first; }
    template <class B>
    u32 next_planes3(u32, const B& blk) const {
        u32 r = 0;
        for (int k = 0; k < 8; k++)
            for (int j = 0; j < 3; j++) r |= (((u32)blk[64 + j] >> k) & 1u) << (3 * k + j);
        return r;
    }
    u32 swap_spill(u32 v) {
        const u32 r = first ? 0u : kept;
        kept = v;
        return r;
    }
    template <class B>
    bool leftover(const B&, u64, u64, const UniTables&, u64*, u64*) const { return false; }
};

// One multi-byte character by its lead byte q[j] (11xxxxxx): straight-line decode by sequence length (same
// acceptance rules as decode_lead).  Returns 1 when the character is alphabetic, 0 when it is in neither class or
// malformed, and its length (2 or 3) negated when it is whitespace; only the lead bytes C2, E1, E2, E3 can begin
// a whitespace character.  q[j + 1 .. j + 3] must be readable.
template <class B>
TRP_HD int classify_lead_at(const B& q, int j, const UniTables& ut) {
    const u32 b0 = q[j], b1 = q[j + 1];
    if ((b1 & 0xC0u) != 0x80u) return 0;
    if (b0 < 0xE0u) {
        const u32 cp = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu);
        if (cp == 0x85u || cp == 0xA0u) return -2;
        return cp_is_alpha_high(cp, ut) ? 1 : 0;
    }
    const u32 b2 = q[j + 2];
    if ((b2 & 0xC0u) != 0x80u) return 0;
    if (b0 < 0xF0u) {
        const u32 cp = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu);
        if (b0 <= 0xE3u && b0 != 0xE0u && cp_is_ws_high(cp)) return -3;
        return cp_is_alpha_high(cp, ut) ? 1 : 0;
    }
    const u32 b3 = q[j + 3];
    if ((b3 & 0xC0u) != 0x80u) return 0;
    const u32 cp = ((b0 & 0x07u) << 18) | ((b1 & 0x3Fu) << 12) | ((b2 & 0x3Fu) << 6) | (b3 & 0x3Fu);
    return cp_is_alpha_high(cp, ut) ? 1 : 0;
}

// Continuation bytes at the very start of a block whose character began in an earlier block: the ones that
// belong to a whitespace character (found by looking back for the lead byte).  `high`: the bytes >= 0x80 of the
// block, `cont6`: bit plane 6 of the block (clear on continuation bytes).
template <class B>
TRP_HD u64 head_ws_by_lookback(const B& blk, u64 high, u64 plane6) {
    u64 ws = 0;
    const u64 head = high & ~plane6 & 7ull;   // continuation bytes among the first three
    for (int j = 0; j < 3 && ((head >> j) & 1ull); j++) {
        u32 cp;
        for (int k = j + 1; k <= 3; k++) {   // the lead byte is before the block
            u32 l = blk[j - k];
            if (l >= 0xC0) {
                int n = decode_lead(blk, j - k, &cp);
                if (n > k && cp_is_ws_high(cp)) ws |= bit(j);
                break;
            }
            if (l < 0x80) break;
        }
    }
    return ws;
}

// The ASCII classes as boolean formulas over the bit planes; mb_ws / mb_alp: what the multi-byte characters add.
TRP_HD Cls classes_from_planes(const u64 (&b)[8], u64 V, u64 mb_ws, u64 mb_alp) {
    const u64 n7 = ~b[7] & V;
    const u64 lowq = n7 & ~b[6] & ~b[5] & ~b[4];                 // 0x00-0x0F
    const u64 ctl = lowq & b[3];                                 // 0x08-0x0F
    Cls c;
    c.nl = ctl & ~b[2] & b[1] & ~b[0];                           // 0x0A
    c.cr = ctl & b[2] & ~b[1] & b[0];                            // 0x0D
    const u64 wsc = ctl & (b[2] | b[1] | b[0]) & ~(b[2] & b[1]); // 0x09-0x0D
    const u64 sp = n7 & ~b[6] & b[5] & ~(b[4] | b[3] | b[2] | b[1] | b[0]);   // 0x20
    c.ws = wsc | sp | mb_ws;
    c.sp = sp;
    c.dig = n7 & ~b[6] & b[5] & b[4] & ~(b[3] & (b[2] | b[1]));  // 0x30-0x39
    const u64 nz5 = b[4] | b[3] | b[2] | b[1] | b[0];
    const u64 gt26 = b[4] & b[3] & (b[2] | (b[1] & b[0]));
    c.alp = (n7 & b[6] & nz5 & ~gt26) | mb_alp;                  // A-Z a-z
    c.pipe = n7 & b[6] & b[5] & b[4] & b[3] & b[2] & ~b[1] & ~b[0];   // 0x7C
    return c;
}

// The multi-byte characters of a block from its bit planes b: the bytes of its whitespace characters (*mb_ws)
// and the lead bytes of its alphabetic characters (*mb_alp).
template <class B, class X>
TRP_HD void mb_classify_planes(const B& blk, const u64 (&b)[8], u64 V, const UniTables& ut, X& nb, u64* out_ws,
                               u64* out_alp) {
    // Bytes >= 0x80 first (few values are live here).  Only lead bytes (11xxxxxx) are decoded; a
    // whitespace character marks all its bytes, an alphabetic one its lead byte.  Continuation bytes at
    // the very start of the block belong to a character that began in the previous block: the previous
    // block says whether they are whitespace when it has settled that character, otherwise they are
    // resolved by looking back.
    u64 mb_ws = 0, mb_alp = 0;
    const u64 high = b[7] & V;
    u64 lead = high & b[6];
    bool look_back = true;
    const bool any_high = nb.any(high != 0);
    if (any_high) {
        u32 lo[8], hi[8];
TRP_UNROLL
        for (int k = 0; k < 8; k++) { lo[k] = (u32)b[k]; hi[k] = (u32)(b[k] >> 32); }
        // with the planes of the next block's first bytes every sequence that begins in this block is seen whole,
        // without them the second half takes the sequences that end by byte 63
        const PlanesAfter3 after = {nb.next_planes3(pack_planes3(lo), blk)};
        const bool has_next = nb.has_next();
        u32 spill = 0;
        if (lead) {
            // the common scripts from the bit planes, half a block at a time
            u32 alp_lo = 0, alp_hi = 0, ws2_lo = 0, ws2_hi = 0, ws3_lo = 0, ws3_hi = 0;
            const PlanesAfter second = {hi};
            const u32 s_lo = classify_leads_planes(lo, second, (u32)lead, &alp_lo, &ws2_lo, &ws3_lo);
            const u32 s_hi = classify_leads_planes(hi, after, (u32)(lead >> 32) & (has_next ? ~0u : 0x1FFFFFFFu), &alp_hi,
                                                   &ws2_hi, &ws3_hi);
            mb_alp = (u64)alp_lo | ((u64)alp_hi << 32);
            lead &= ~((u64)s_lo | ((u64)s_hi << 32));
            // a whitespace character marks all its bytes
            const u64 w3 = (u64)ws3_lo | ((u64)ws3_hi << 32), w23 = w3 | (u64)ws2_lo | ((u64)ws2_hi << 32);
            mb_ws = (w23 | (w23 << 1) | (w3 << 2)) & V;
            // for the next block: its first / second byte continue a whitespace character of this block;
            // bit 2: a character that may run over is still unsettled (the next block has to look back)
            spill = (u32)(w23 >> 63) | ((u32)(w3 >> 62) & 1u) | ((u32)(w3 >> 63) << 1) | ((lead >> 61) ? 4u : 0u);
        }
        const u32 from_prev = nb.swap_spill(spill);
        if (nb.has_prev()) {
            mb_ws |= (u64)(from_prev & 3u) & V;
            look_back = (from_prev & 4u) != 0;
        }
    }
    // what the rules have not settled: the group may take the characters on together (leftover), otherwise the
    // block decodes its own, one after the other
    if (!any_high || !nb.leftover(blk, lead, V, ut, &mb_ws, &mb_alp)) {
        while (lead) {
            const int j = ctz(lead);
            lead &= lead - 1;
            const int k = classify_lead_at(blk, j, ut);
            if (k > 0) mb_alp |= bit(j);
            else if (k < 0) mb_ws |= ((k == -2 ? 3ull : 7ull) << j) & V;
        }
    }
    if (look_back) mb_ws |= head_ws_by_lookback(blk, high, b[6]);
    *out_ws = mb_ws;
    *out_alp = mb_alp;
}
template <class B, class X>
TRP_HD Cls classify_block_fast(const B& blk, int nvalid, const UniTables& ut, X& nb) {
    u32 w[16];
TRP_UNROLL
    for (int k = 0; k < 16; k++) w[k] = blk.word(k);
    u64 b[8];
    transpose_block(w, b);
    const u64 V = below(nvalid);
    u64 mb_ws, mb_alp;
    mb_classify_planes(blk, b, V, ut, nb, &mb_ws, &mb_alp);
    return classes_from_planes(b, V, mb_ws, mb_alp);
}
template <class B>
TRP_HD Cls classify_block_fast(const B& blk, int nvalid, const UniTables& ut) {
    NoNeighbours nb;
    return classify_block_fast(blk, nvalid, ut, nb);
}

// The whitespace mask alone (what the pass-through configuration needs): no Alphabetic lookups at all.
// Only the lead bytes C2, E1, E2 and E3 can begin a whitespace character.
template <class B>
TRP_HD u64 classify_ws_fast(const B& blk, int nvalid, u64* sp = nullptr) {
    u32 w[16];
TRP_UNROLL
    for (int k = 0; k < 16; k++) w[k] = blk.word(k);
    u64 b[8];
    transpose_block(w, b);
    const u64 V = below(nvalid);
    const u64 n7 = ~b[7] & V;
    const u64 ctl = n7 & ~b[6] & ~b[5] & ~b[4] & b[3];
    const u64 space = n7 & ~b[6] & b[5] & ~(b[4] | b[3] | b[2] | b[1] | b[0]);   // 0x20
    if (sp) *sp = space;
    u64 ws = (ctl & (b[2] | b[1] | b[0]) & ~(b[2] & b[1])) | space;
    const u64 high = b[7] & V;
    const u64 z432 = ~(b[4] | b[3] | b[2]);
    u64 lead = high & b[6] & z432 & ((~b[5] & b[1] & ~b[0]) | (b[5] & (b[1] | b[0])));   // C2 | E1..E3
    while (lead) {
        const int j = ctz(lead);
        lead &= lead - 1;
        const u32 b0 = blk[j], b1 = blk[j + 1];
        if ((b1 & 0xC0u) != 0x80u) continue;
        if (b0 < 0xE0u) {
            const u32 cp = ((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu);
            if (cp == 0x85u || cp == 0xA0u) ws |= (3ull << j) & V;
            continue;
        }
        const u32 b2 = blk[j + 2];
        if ((b2 & 0xC0u) != 0x80u) continue;
        const u32 cp = ((b0 & 0x0Fu) << 12) | ((b1 & 0x3Fu) << 6) | (b2 & 0x3Fu);
        if (cp_is_ws_high(cp)) ws |= (7ull << j) & V;
    }
    u64 head = high & ~b[6] & 7ull;   // continuation bytes among the first three
    for (int j = 0; j < 3 && ((head >> j) & 1ull); j++) {
        u32 cp;
        for (int k = j + 1; k <= 3; k++) {   // the lead byte is before the block
            u32 l = blk[j - k];
            if (l >= 0xC0) {
                int n = decode_lead(blk, j - k, &cp);
                if (n > k && cp_is_ws_high(cp)) ws |= bit(j);
                break;
            }
            if (l < 0x80) break;
        }
    }
    return ws;
}

// Is the byte q[at] part of a whitespace character?  q[at-3..at+3] must be readable.
template <class B>
TRP_HD bool byte_is_ws(const B& q, int at) {
    u32 b = q[at];
    if (b < 0x80) return (ascii_class(b) & C_WS) != 0;
    u32 cp;
    if (b >= 0xC0) return decode_lead(q, at, &cp) && cp_is_ws_high(cp);
    for (int k = 1; k <= 3; k++) {
        u32 l = q[at - k];
        if (l >= 0xC0) return decode_lead(q, at - k, &cp) > k && cp_is_ws_high(cp);
        if (l < 0x80) return false;
    }
    return false;
}

// ---- line structure ------------------------------------------------------------------------------
struct Derived {
    u64 V;     // bytes that exist (below the end of the corpus)
    u64 RS;    // first byte of a record
    u64 LS;    // first byte of a line (record start, or the byte after a terminator)
    u64 TERM;  // line terminators after normalisation: '\n', and '\r' not followed by '\n'
    u64 DEL;   // '\r' of a "\r\n" pair (removed by the normalisation)
    u64 LC;    // line content: V minus TERM and DEL
};

// next_nl_same_rec: the byte after the block exists, is '\n' and is not a record start.
// prev_term: the byte before the block exists and is a line terminator.
TRP_HD Derived derive(const Cls& c, u64 V, u64 RS, bool next_nl_same_rec, bool prev_term) {
    Derived d;
    d.V = V;
    d.RS = RS & V;
    d.DEL = c.cr & (((c.nl & ~RS) >> 1) | (next_nl_same_rec ? bit(63) : 0)) & V;
    d.TERM = (c.nl | (c.cr & ~d.DEL)) & V;
    d.LS = (d.RS | (d.TERM << 1) | (prev_term ? 1ull : 0ull)) & V;
    d.LC = V & ~(d.TERM | d.DEL);
    return d;
}

// Is the byte before the block a terminator?  prev = that byte, cur0 = first byte of the block,
// rs0 = first byte of the block starts a record.
TRP_HD bool prev_is_term(bool has_prev, u32 prev, bool has_cur, u32 cur0, bool rs0) {
    if (!has_prev) return false;
    if (prev == 0x0A) return true;
    if (prev == 0x0D) return !(has_cur && cur0 == 0x0A && !rs0);
    return false;
}

// Statistics of (a part of) one line.  f: [1:0] class of first non-whitespace content byte
// (0 none, 1 pipe, 2 text), [3:2] class of the last one, [4] a text byte was seen.
struct LineSum {
    u32 d, a, p, f;
};
TRP_HD LineSum ls_empty() { LineSum s = {0, 0, 0, 0}; return s; }
TRP_HD u32 ls_first(const LineSum& s) { return s.f & 3u; }
TRP_HD u32 ls_last(const LineSum& s) { return (s.f >> 2) & 3u; }
TRP_HD bool ls_seen_text(const LineSum& s) { return (s.f >> 4) & 1u; }
// A = earlier part, B = later part of the same line.
TRP_HD LineSum ls_add(const LineSum& A, const LineSum& B) {
    LineSum s;
    s.d = A.d + B.d;
    s.a = A.a + B.a;
    s.p = A.p + B.p;
    u32 first = ls_first(A) ? ls_first(A) : ls_first(B);
    u32 last = ls_last(B) ? ls_last(B) : ls_last(A);
    s.f = first | (last << 2) | ((A.f | B.f) & 16u);
    return s;
}
TRP_HD LineSum range_sum(const Cls& c, const Derived& d, u64 R) {
    LineSum s;
    u64 lc = d.LC & R;
    s.d = popc(c.dig & lc);
    s.a = popc(c.alp & lc);
    s.p = popc(c.pipe & lc);
    u64 nw = lc & ~c.ws;
    u32 f = 0;
    if (nw) {
        u32 first = (c.pipe >> ctz(nw)) & 1u ? 1u : 2u;
        u32 last = (c.pipe >> top(nw)) & 1u ? 1u : 2u;
        f = first | (last << 2);
        if (nw & ~c.pipe) f |= 16u;
    }
    s.f = f;
    return s;
}

// Forward summary of a run of blocks.  g: bit0 has_ls, bit1 has_rs, bit2 alpha before the first
// record start (or anywhere when there is none), bit3 alpha at/after the last record start,
// bit4/bit5: "the record has an alphabetic char before the last line start" = c0 | (c1 & AR_in)
// where AR_in is the same statement about the bytes before the run.
struct J {
    LineSum head;   // bytes before the first line start (whole run when has_ls == 0)
    LineSum tail;   // bytes from the last line start on (valid when has_ls)
    u32 g;
};
enum : u32 { G_LS = 1, G_RS = 2, G_AH = 4, G_AT = 8, G_C0 = 16, G_C1 = 32 };
TRP_HD J j_identity() { J j; j.head = ls_empty(); j.tail = ls_empty(); j.g = 0; return j; }

TRP_HD J j_combine(const J& A, const J& B) {
    J C;
    bool als = A.g & G_LS, bls = B.g & G_LS;
    if (!als && !bls) { C.head = ls_add(A.head, B.head); C.tail = ls_empty(); }
    else if (als && !bls) { C.head = A.head; C.tail = ls_add(A.tail, B.head); }
    else if (!als && bls) { C.head = ls_add(A.head, B.head); C.tail = B.tail; }
    else { C.head = A.head; C.tail = B.tail; }
    u32 g = (A.g | B.g) & (G_LS | G_RS);
    bool ars = A.g & G_RS, brs = B.g & G_RS;
    bool ah = ars ? (A.g & G_AH) != 0 : ((A.g | B.g) & G_AH) != 0;
    bool at = brs ? (B.g & G_AT) != 0 : ((A.g | B.g) & G_AT) != 0;
    if (ah) g |= G_AH;
    if (at) g |= G_AT;
    if (bls) {
        bool known = ars ? (A.g & G_AT) != 0 : (A.g & G_AH) != 0;
        bool dep = !ars;
        if ((B.g & G_C0) || ((B.g & G_C1) && known)) g |= G_C0;
        if ((B.g & G_C1) && dep) g |= G_C1;
    } else {
        g |= A.g & (G_C0 | G_C1);
    }
    C.g = g;
    return C;
}

// Statistics of the bytes R of one line (R lies inside the valid bytes; digits, letters and pipes are never
// terminators).  nws: the bytes of the block that are not whitespace (terminators are whitespace).  Without
// branches or bit searches: with np / nt the pipes / the other non-whitespace bytes of R, the last one is a pipe
// when np > nt as numbers (the highest bit decides), the first one when the same holds for the mirror images.
TRP_HD LineSum line_part_sum(const Cls& c, u64 nws, u64 R) {
    LineSum s;
    s.d = (u32)popc(c.dig & R);
    s.a = (u32)popc(c.alp & R);
    s.p = (u32)popc(c.pipe & R);
    const u64 nw = R & nws, np = nw & c.pipe, nt = nw ^ np;
    const u32 first = nw ? (brev(np) > brev(nt) ? 1u : 2u) : 0u;
    const u32 last = nw ? (np > nt ? 1u : 2u) : 0u;
    s.f = first | (last << 2) | (nt ? 16u : 0u);
    return s;
}

TRP_HD J lane_J(const Cls& c, const Derived& d) {
    J j;
    const bool has_ls = d.LS != 0;
    const u64 nws = d.V & ~c.ws;
    // before the first line start (the whole block when it has none); from the last line start on
    const u64 Rh = ((d.LS & (0ull - d.LS)) - 1ull) & d.V;
    const u64 Rt = has_ls ? (~below(top(d.LS)) & d.V) : 0ull;
    j.head = line_part_sum(c, nws, Rh);
    j.tail = line_part_sum(c, nws, Rt);
    // alphabetic characters since the last record start (or the start of the block) before each byte; the carry
    // out of the last byte is "an alphabetic character at or after the last record start"
    const u64 alp = c.alp & d.V;
    const u64 K = d.RS >> 1, X = ~K, Y = alp & X;
    const u64 sum = X + Y;
    const u64 FA = sum ^ X ^ Y;
    const bool at_after = sum < X;
    const u64 first_rs_below = (d.RS & (0ull - d.RS)) - 1ull;      // bytes before the first record start (all without one)
    u32 g = 0;
    if (has_ls) {
        g |= G_LS;
        const u64 last_ls = d.LS & Rt;                               // the last line start alone
        if (FA & last_ls) g |= G_C0;
        if (!(d.RS & (last_ls | (last_ls - 1ull)))) g |= G_C1;       // no record start at or before it
    }
    if (d.RS) g |= G_RS;
    if (alp & first_rs_below) g |= G_AH;
    if (at_after) g |= G_AT;
    j.g = g;
    return j;
}

// Backward summary: the bytes of a run that belong to the line entering it from the left.
struct S {
    LineSum head;
    u32 has_ls;
};
TRP_HD S s_identity() { S s; s.head = ls_empty(); s.has_ls = 0; return s; }
// A = earlier run, B = later run.
TRP_HD S s_combine(const S& A, const S& B) {
    S c;
    c.has_ls = A.has_ls | B.has_ls;
    c.head = A.has_ls ? A.head : ls_add(A.head, B.head);
    return c;
}
TRP_HD S s_from_J(const J& j) { S s; s.head = j.head; s.has_ls = j.g & G_LS; return s; }

// ---- configuration ---------------------------------------------------------------------------------
struct DevCfg {
    u32 denoise_enabled;
    u32 strip_markdown;
    float max_digit_ratio;
    u32 max_window_tokens;
    u32 n_overlaps;
    u32 summary_fallback_tokens;
    u32 overlaps[MAX_OVERLAPS];
    u32 normalize;   // 1: normalize_inline_whitespace instead of the denoiser (see plan_normalize)
};

// digits as f32 / (digits + alpha) as f32 > max   (denoiser_preprocessor.rs:37-45, :115-121)
TRP_HD bool ratio_exceeds(u32 d, u32 a, float max_ratio) {
    u32 t = d + a;
    if (t == 0) return 0.0f > max_ratio;
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __fdiv_rn((float)d, (float)t);
#else
    volatile float fd = (float)d, ft = (float)t;
    volatile float r = fd / ft;
#endif
    return r > max_ratio;
}
TRP_HD bool ratio_exceeds64(u64 d, u64 a, float max_ratio) {
    u64 t = d + a;
    if (t == 0) return 0.0f > max_ratio;
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __fdiv_rn((float)d, (float)t);
#else
    volatile float fd = (float)d, ft = (float)t;
    volatile float r = fd / ft;
#endif
    return r > max_ratio;
}

// The same test through a table: for a given total t = digits + alpha the f32 quotient digits / t grows with the
// number of digits, so "digits / t > max" is "digits >= thr1[t]", thr1[t] being the smallest number of digits that
// exceeds (t + 1 when none does).  The table covers the totals below RATIO_TAB (lines of ordinary length) and is
// filled on the host with the very division it replaces; larger totals take the division.
constexpr u32 RATIO_TAB = 256;
inline void ratio_table_fill(float max_ratio, uint16_t* thr1) {
    for (u32 t = 0; t < RATIO_TAB; t++) {
        u32 d = 0;
        while (d <= t && !ratio_exceeds(d, t - d, max_ratio)) d++;
        thr1[t] = (uint16_t)d;
    }
}
struct RatioTab {
    float max;
    const uint16_t* thr1;   // RATIO_TAB entries (LDS on the device)
    TRP_HD bool exceeds(u32 d, u32 a) const {
        const u32 t = d + a;
        if (t < RATIO_TAB) return d >= (u32)thr1[t];
        return ratio_exceeds(d, a, max);
    }
};
struct RatioDiv {   // no table at hand: always the division
    float max;
    TRP_HD bool exceeds(u32 d, u32 a) const { return ratio_exceeds(d, a, max); }
};

// ---- the emit plan -----------------------------------------------------------------------------------
// Four bit planes per block.  C: the byte is copied.  TOK: the byte belongs to a token of the output
// (copied and not whitespace).  PSP / PNL: one ' ' / '\n' is emitted in front of the (copied) byte.
// A set TOK bit on a byte that is not copied marks the first byte of a digit-heavy ("gate 3") line
// that is still to be solved; there PSP means "table line" and PNL "the record already has a kept
// line before this one".
struct Plan {
    u64 C, TOK, PSP, PNL;
};

struct LaneCarry {
    LineSum ci;     // the current line, from its start to the start of this block
    bool ci_rab;    // the record has an alphabetic char before that line's start
    bool ar_in;     // the record has an alphabetic char before this block
    LineSum co;     // the rest of the block's last line after the end of this block
    bool next_rs;   // the byte after this block starts a record (or does not exist)
};

// ---- normalize_inline_whitespace (src/utils.rs:18-33) on the same machinery ------------------------
// Canonical form of a plan: a separator that is emitted in front of byte j while byte j-1 is dropped and
// happens to be that very character ('\n' terminator before a kept line, single space before a table
// cell or a kept token) is expressed as "copy byte j-1" instead.  Same output bytes, longer copy runs.
// Only where the prefixed byte itself is copied (not the marker of an unsolved digit-heavy line).
TRP_HD void plan_fold_prefixes(Plan& p, u64 sp, u64 nl, u64 V) {
    u64 cn = (p.PNL >> 1) & ~p.C & nl & V;
    u64 cs = (p.PSP >> 1) & ~p.C & sp & V & ~(p.PNL >> 1);
    cn &= (p.C >> 1);
    cs &= (p.C >> 1);
    p.C |= cn | cs;
    p.PNL &= ~(cn << 1);
    p.PSP &= ~(cs << 1);
}

// The reference collapses every run of White_Space characters into one ' ' and trims the result, i.e.
// the output is the record's whitespace-separated tokens joined by single spaces.  As an emit plan:
// copy the token bytes, and put one ' ' in front of every token start except the first of its record.
// "The record has a token before this block" is the record-level flag the denoiser machinery already
// carries for alphabetic characters, so this mode classifies every non-whitespace byte as alphabetic
// and knows no line terminators.
TRP_HD Cls normalize_cls(const Cls& c, u64 V) {
    Cls n = {c.ws, 0, 0, 0, 0, V & ~c.ws, 0};
    return n;
}
// prev_tok: the byte before the block exists, belongs to the same record and is not whitespace.
// sp: the bytes that are exactly 0x20.
TRP_HD Plan plan_normalize(const Cls& c, const Derived& d, bool ar_in, bool prev_tok, u64 sp) {
    const u64 T = d.V & ~c.ws;
    const u64 TS = T & (~((T << 1) | (prev_tok ? 1ull : 0ull)) | d.RS);
    u64 first = 0;   // the first token start of every record that starts in, or reaches, this block
    if (!ar_in) {
        const u64 x = TS & (d.RS ? below(ctz(d.RS)) : ~0ull);
        first |= x & (0ull - x);
    }
    for (u64 rs = d.RS; rs;) {
        const int r = ctz(rs);
        rs &= rs - 1;
        const u64 x = TS & ~below(r) & (rs ? below(ctz(rs)) : ~0ull);
        first |= x & (0ull - x);
    }
    Plan p = {T, T, TS & ~first, 0};
    plan_fold_prefixes(p, sp, 0, d.V);
    return p;
}

TRP_HD LaneCarry carry_from(const J& P, const S& X, bool next_rs) {
    LaneCarry k;
    k.ci = (P.g & G_LS) ? P.tail : P.head;
    k.ci_rab = (P.g & G_C0) != 0;
    k.ar_in = (P.g & G_RS) ? (P.g & G_AT) != 0 : (P.g & G_AH) != 0;
    k.co = X.head;
    k.next_rs = next_rs;
    return k;
}

// Segmented "seen before": bit i of the result is the OR of x over the positions j < i of the same segment, a
// segment beginning at every set bit of `starts`; c0 is the value that enters bit 0 (ignored when bit 0 starts a
// segment).  One addition: with K = positions whose successor starts a segment, the operands X = ~K and Y = x & ~K
// generate a carry at every x, pass a carry on everywhere else and kill it in front of a segment start, and the
// carry into each position is what was asked for.
TRP_HD u64 seen_before(u64 x, u64 starts, bool c0) {
    const u64 K = starts >> 1;
    const u64 X = ~K, Y = x & ~K;
    const u64 sum = X + Y + ((c0 && !(starts & 1ull)) ? 1ull : 0ull);
    return sum ^ X ^ Y;
}

// own_head / own_tail: the block's own statistics before its first and from its last line start on, as
// lane_J computes them (head and tail of the block's summary).
// *head_g3 (optional): the line that runs into the block from an earlier one is a digit-heavy line; it is
// marked in the block that holds its first byte, nothing of it is planned here.
// fold = false leaves plan_fold_prefixes() to the caller.
// The gates (denoiser_preprocessor.rs:228-262) are decided line by line - drop (no alphabetic character), table
// row, digit-heavy, plain - and collected as masks of the lines' bytes; the plan then follows from the masks for
// all lines of the block at once.
template <class RT>
TRP_HD Plan plan_block(const Cls& c, const Derived& d, const LaneCarry& k, const DevCfg& cfg, const RT& rt,
                       const LineSum& own_head, const LineSum& own_tail, bool* head_g3 = nullptr, bool fold = true) {
    const bool has_ls = d.LS != 0;
    const bool head0 = !(d.LS & 1ull);      // the block starts inside a line begun earlier
    const int first_ls = has_ls ? ctz(d.LS) : 64, last_ls = has_ls ? top(d.LS) : 0;
    const u64 empty = d.TERM | d.DEL;
    // bytes of the plain lines / of the table rows that are kept; first bytes of the digit-heavy lines that start
    // in the block, and which of those are table rows
    u64 PLm = 0, TBm = 0, G3S = 0, G3T = 0;
    bool head_pending = false;
    auto verdict = [&](u32 dd, u32 aa, u32 pp, bool first_pipe, u64 R, u64 sbit, bool is_head) {
        const bool keep = aa != 0;                                   // gate 2
        const bool table = cfg.strip_markdown && first_pipe && pp >= 2u;
        const bool heavy = keep && rt.exceeds(dd, aa);               // gate 3
        const bool plain = keep && !heavy && !table, tbl = keep && !heavy && table;
        PLm |= plain ? R : 0ull;
        TBm |= tbl ? R : 0ull;
        G3S |= heavy ? sbit : 0ull;
        G3T |= (heavy && table) ? sbit : 0ull;
        if (is_head && heavy) head_pending = true;
    };
    // The line that runs into the block and the block's last line take their own part from the block's
    // summary; the lines in between (they start and end inside the block) are counted here.  Empty lines (a
    // line start that sits on its own terminator) are skipped: nothing of them is ever emitted.
    if (head0) {
        LineSum tot = ls_add(k.ci, own_head);
        if (!has_ls) tot = ls_add(tot, k.co);
        verdict(tot.d, tot.a, tot.p, ls_first(tot) == 1u, below(first_ls) & d.V, 0ull, true);
    }
    if (has_ls && !((empty >> last_ls) & 1ull)) {
        const LineSum tot = ls_add(own_tail, k.co);
        verdict(tot.d, tot.a, tot.p, ls_first(tot) == 1u, ~below(last_ls) & d.V, bit(last_ls), false);
    }
    u64 todo = has_ls ? (d.LS & ~empty & ~bit(last_ls)) : 0ull;
    while (todo) {
        const u64 sb = todo & (0ull - todo);                 // the line's first byte
        todo ^= sb;
        const u64 above = d.LS & ~(sb | (sb - 1ull));        // a later line start exists
        const u64 R = (above & (0ull - above)) - sb;         // [s, next line start)
        const u64 nw = R & ~c.ws;                            // (terminators are whitespace)
        verdict((u32)popc(c.dig & R), (u32)popc(c.alp & R), (u32)popc(c.pipe & R),
                (c.pipe & nw & (0ull - nw)) != 0, R, sb, false);
    }
    Plan out;
    // the record has an alphabetic character (hence a kept line) before this byte
    const u64 FA = seen_before(c.alp & d.V, d.RS, k.ar_in);
    out.C = d.LC & PLm;
    out.TOK = (d.LC & PLm & ~c.ws) | G3S;
    out.PSP = G3T;
    out.PNL = ((PLm & d.LS) | G3S) & FA;
    if (TBm) {
        // Cell trimming of table rows (strip_table_pipes): a whitespace byte survives only when the nearest
        // non-whitespace byte on either side, within the line, is text (not a pipe).
        const u64 T = d.LC & ~c.ws & ~c.pipe;     // text bytes
        const u64 W = d.LC & c.ws;                // whitespace inside lines
        const u64 cin_text = (head0 && ls_last(k.ci) == 2u) ? 1ull : 0ull;
        const u64 Wp = W & ~d.LS;
        const u64 SL = ((T << 1) | cin_text) & Wp;
        const u64 LT = Wp & ~(Wp + SL);
        const u64 Wq = W & ~((d.LS >> 1) | (k.next_rs ? bit(63) : 0ull));
        const u64 cin_right = (ls_first(k.co) == 2u) ? bit(63) : 0ull;
        const u64 SR = ((T >> 1) | cin_right) & Wq;
        const u64 rW = brev(Wq), rS = brev(SR);
        const u64 RT2 = brev(rW & ~(rW + rS));
        const u64 IW = LT & RT2;
        const u64 prevtext = (((T | LT) << 1) | cin_text) & ~d.LS;
        const u64 CS = T & ~prevtext & TBm;       // first bytes of the trimmed cells of the kept rows
        // the first text byte of every line (of a line that runs in: only when it had no text before the block);
        // no byte of a line before its first text byte is alphabetic, so FA there is the line's own "the record
        // has a kept line before this one"
        const u64 FT = T & ~seen_before(T, d.LS, head0 && ls_seen_text(k.ci));
        out.C |= TBm & (T | IW);
        out.TOK |= TBm & T;
        out.PSP |= CS & ~FT;
        out.PNL |= CS & FT & FA;
    }
    if (fold) plan_fold_prefixes(out, c.sp, c.nl, d.V);
    if (head_g3) *head_g3 = head_pending;
    return out;
}
template <class RT>
TRP_HD Plan plan_block(const Cls& c, const Derived& d, const LaneCarry& k, const DevCfg& cfg, const RT& rt) {
    const J own = lane_J(c, d);
    return plan_block(c, d, k, cfg, rt, own.head, own.tail);
}

// Plan of the pass-through configuration (denoiser disabled): every byte is "copied" in place and
// tokens are the runs of non-whitespace bytes.
TRP_HD Plan plan_passthrough(const Cls& c, u64 V) {
    Plan p;
    p.C = V;
    p.TOK = V & ~c.ws;
    p.PSP = 0;
    p.PNL = 0;
    return p;
}

// Token starts / ends of a block from its plan.  prev_tok: the byte before the block is a token
// byte; next_tok: the byte after it is; next_rs: the byte after the block starts a record.
TRP_HD u64 token_starts(u64 TOKC, u64 RS, bool prev_tok) {
    return TOKC & (~((TOKC << 1) | (prev_tok ? 1ull : 0ull)) | RS);
}
TRP_HD u64 token_ends(u64 TOKC, u64 RS, bool next_tok, bool next_rs) {
    u64 nt = (TOKC >> 1) | (next_tok ? bit(63) : 0ull);
    u64 nr = (RS >> 1) | (next_rs ? bit(63) : 0ull);
    return TOKC & (~nt | nr);
}

// ---- chunk windows (sliding_window.rs:40-115) --------------------------------------------------------
struct ChunkDesc {
    u32 record, kind, index, overlap, span, tok_begin, tok_end, reserved;
};

TRP_HD u64 chunk_windows(u64 T, u64 span, u64 overlap) {
    if (span == 0) return T;
    u64 stride = span > overlap ? span - overlap : 0;
    if (stride < 1) stride = 1;
    return 1 + (T - span + stride - 1) / stride;
}
TRP_HD u64 chunk_count(u64 T, const DevCfg& cfg) {
    if (T == 0) return 0;
    if (T <= cfg.max_window_tokens) return 1;
    u64 n = 0;
    for (u32 k = 0; k < cfg.n_overlaps; k++) n += chunk_windows(T, cfg.max_window_tokens, cfg.overlaps[k]);
    if (cfg.summary_fallback_tokens > 0) n += 1;
    return n;
}
TRP_HD ChunkDesc chunk_at(u64 T, u64 j, u32 record, const DevCfg& cfg) {
    ChunkDesc c = {record, 0, 0, 0, 0, 0, 0, 0};
    if (T <= cfg.max_window_tokens) {
        c.span = (u32)T;
        c.tok_end = (u32)T;
        return c;
    }
    u64 span = cfg.max_window_tokens;
    for (u32 k = 0; k < cfg.n_overlaps; k++) {
        u64 ov = cfg.overlaps[k];
        u64 w = chunk_windows(T, span, ov);
        if (j < w) {
            u64 stride = span > ov ? span - ov : 0;
            if (stride < 1) stride = 1;
            u64 start = j * stride;
            u64 end = start + span < T ? start + span : T;
            c.index = (u32)j;
            c.overlap = (u32)ov;
            c.span = (u32)span;
            c.tok_begin = (u32)start;
            c.tok_end = (u32)end;
            return c;
        }
        j -= w;
    }
    u64 cap = cfg.summary_fallback_tokens < cfg.max_window_tokens ? cfg.summary_fallback_tokens
                                                                   : cfg.max_window_tokens;
    if (cap < 1) cap = 1;
    c.kind = 1;
    c.tok_end = (u32)(T < cap ? T : cap);
    return c;
}

// ---- serial solver for one digit-heavy line (gate 3, denoiser_preprocessor.rs:70-140) ---------------
// Memory-free restatement: a token is kept when its distance (in tokens) to the nearest
// alphabetic-bearing token is <= K, K being the number of accepted waves.  The line is walked gap
// by gap (a gap = the tokens between two alphabetic-bearing tokens); each gap is walked twice, first
// for its length, then with every token's distance known.
struct ByteSrc {
    const u8* p;   // corpus (16-byte aligned on the device)
    u64 n;         // corpus length
    // The device keeps the 16-byte chunk it read last in registers: the walks below are sequential,
    // so this turns one memory round trip per byte into one per 16 bytes.
    u64 cbase, clo, chi;
    TRP_HD u32 get(u64 q) {
#if defined(__HIP_DEVICE_COMPILE__)
        u64 base = q & ~15ull;
        if (base != cbase) {
            if (base + 16 <= n) {
                uint4 v = *reinterpret_cast<const uint4*>(p + base);
                clo = (u64)v.x | ((u64)v.y << 32);
                chi = (u64)v.z | ((u64)v.w << 32);
            } else {
                clo = chi = 0;
                for (u64 i = base; i < n; i++) {
                    u64 sh = 8 * (i & 7);
                    if ((i & 8) == 0) clo |= (u64)p[i] << sh; else chi |= (u64)p[i] << sh;
                }
            }
            cbase = base;
        }
        u64 w = (q & 8) ? chi : clo;
        return (u32)(w >> (8 * (q & 7))) & 0xFFu;
#else
        return p[q];
#endif
    }
};
TRP_HD ByteSrc make_src(const u8* p, u64 n) {
    ByteSrc s;
    s.p = p; s.n = n; s.cbase = ~0ull; s.clo = 0; s.chi = 0;
    return s;
}
struct TokInfo {
    u64 ts, te;   // [ts, te) byte range, te == ts: no token
    u32 digits;
    bool alpha;
};
TRP_HD u32 char_at(ByteSrc& s, u64 q, u64 end, int* len, const UniTables& ut) {
    // class code of the character starting at q (q < end); *len = its byte length
    u32 b = s.get(q);
    if (b < 0x80) { *len = 1; return ascii_class(b); }
    *len = 1;
    if (b < 0xC0) return 0;
    u8 tmp[4] = {(u8)b, 0, 0, 0};
    for (int i = 1; i < 4; i++) tmp[i] = (q + i < end) ? (u8)s.get(q + i) : 0;
    u32 cp;
    PlainBlk tb = {tmp};
    int n = decode_lead(tb, 0, &cp);
    if (!n) return 0;
    *len = n;
    if (cp_is_ws_high(cp)) return C_WS;
    if (cp_is_alpha_high(cp, ut)) return C_ALPHA;
    return 0;
}
TRP_HD TokInfo next_token(ByteSrc& s, u64 pos, u64 end, bool table, const UniTables& ut) {
    TokInfo t;
    t.digits = 0;
    t.alpha = false;
    const u32 sep = C_WS | (table ? C_PIPE : 0u);
    int len;
    while (pos < end) {
        u32 k = char_at(s, pos, end, &len, ut);
        if (!(k & sep)) break;
        pos += len;
    }
    t.ts = pos;
    while (pos < end) {
        u32 k = char_at(s, pos, end, &len, ut);
        if (k & sep) break;
        if (k & C_DIGIT) t.digits++;
        else if (k & C_ALPHA) t.alpha = true;
        pos += len;
    }
    t.te = pos;
    return t;
}

// Calls f(token, keep) for every token of the line [p, e) in order.  Pass K = ~0 to get the digit
// sum / count of the tokens at distance exactly `probe` instead (no callback use).
template <class F>
TRP_HD void g3_walk(ByteSrc& s, u64 p, u64 e, bool table, const UniTables& ut, F f) {
    u64 pos = p;
    bool left_alpha = false;
    for (;;) {
        // first walk: length of the gap and what closes it
        u64 g = 0, q = pos;
        TokInfo closer;
        bool right_alpha = false;
        for (;;) {
            closer = next_token(s, q, e, table, ut);
            if (closer.te == closer.ts) break;
            if (closer.alpha) { right_alpha = true; break; }
            g++;
            q = closer.te;
        }
        // second walk: every gap token with its distance to the nearest token that has a letter
        q = pos;
        for (u64 i = 1; i <= g; i++) {
            TokInfo t = next_token(s, q, e, table, ut);
            q = t.te;
            u64 df = left_alpha ? i : ~0ull;
            u64 db = right_alpha ? g + 1 - i : ~0ull;
            f(t, df < db ? df : db);
        }
        if (!right_alpha) break;
        f(closer, 0ull);
        left_alpha = true;
        pos = closer.te;
    }
}

// Number of accepted waves for the line [p, e) (denoiser_preprocessor.rs:94-131).  The digits of the
// tokens at each distance are gathered for G3_WINDOW distances per walk over the line.
constexpr int G3_WINDOW = 32;
TRP_HD u64 g3_solve_K(ByteSrc& s, u64 p, u64 e, bool table, float max_ratio, const UniTables& ut) {
    u64 d = 0, a = 0, K = 0;
    for (u64 k0 = 0;; k0 += G3_WINDOW) {
        u64 hd[G3_WINDOW], hc[G3_WINDOW];
        for (int i = 0; i < G3_WINDOW; i++) hd[i] = hc[i] = 0;
        g3_walk(s, p, e, table, ut, [&](const TokInfo& t, u64 dist) {
            if (dist == 0 && k0 == 0) {   // seed: digits and letters of the tokens that have a letter
                d += t.digits;
                int len;
                for (u64 q = t.ts; q < t.te; q += len)
                    if (char_at(s, q, t.te, &len, ut) & C_ALPHA) a++;
            } else if (dist >= k0 && dist < k0 + G3_WINDOW) {
                hd[dist - k0] += t.digits;
                hc[dist - k0]++;
            }
        });
        for (int i = (k0 == 0 ? 1 : 0); i < G3_WINDOW; i++) {
            if (hc[i] == 0) return K;
            if (ratio_exceeds64(d + hd[i], a, max_ratio)) return K;
            d += hd[i];
            K = k0 + (u64)i;
        }
    }
}

// ---- gate 3 on bit masks -----------------------------------------------------------------------------
// A line of at most 64 * W bytes solved without walking its tokens: every set of tokens is the mask of its
// bytes (bit i = byte i of the line, W x 64 bits), and "the token after / before a kept token" is found
// with carry chains.  Adding a point to a mask of runs of ones clears the run from the point upwards and
// sets the first zero above it; the mirror image (bits reversed) does the same downwards.
constexpr int G3_WORDS = 4;    // longest line the mask solver takes: 256 bytes (longer ones are walked serially)
constexpr int G3_SHORT = 2;    // lines of at most 128 bytes (nearly all) take the solver of half the width
template <int W>
struct G3BitsT {
    u64 w[W];
};
typedef G3BitsT<G3_WORDS> G3Bits;
TRP_HD u64 bitrev64(u64 x) {
#if defined(__clang__)
    return __builtin_bitreverse64(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}
template <int W>
TRP_HD G3BitsT<W> g3b_rev(const G3BitsT<W>& a) {
    G3BitsT<W> r;
    TRP_UNROLL
    for (int j = 0; j < W; j++) r.w[j] = bitrev64(a.w[W - 1 - j]);
    return r;
}
template <int W>
TRP_HD G3BitsT<W> g3b_add(const G3BitsT<W>& a, const G3BitsT<W>& b) {
    G3BitsT<W> r;
#if defined(__clang__)
    unsigned long long c = 0;
    TRP_UNROLL
    for (int j = 0; j < W; j++) r.w[j] = __builtin_addcll(a.w[j], b.w[j], c, &c);
#else
    u64 c = 0;
    for (int j = 0; j < W; j++) {
        const u64 s1 = a.w[j] + b.w[j], s2 = s1 + c;
        c = (u64)(s1 < a.w[j]) | (u64)(s2 < s1);
        r.w[j] = s2;
    }
#endif
    return r;
}
template <int W>
TRP_HD G3BitsT<W> g3b_and(const G3BitsT<W>& a, const G3BitsT<W>& b) {
    G3BitsT<W> r;
    TRP_UNROLL
    for (int j = 0; j < W; j++) r.w[j] = a.w[j] & b.w[j];
    return r;
}
template <int W>
TRP_HD G3BitsT<W> g3b_andn(const G3BitsT<W>& a, const G3BitsT<W>& b) {   // a & ~b
    G3BitsT<W> r;
    TRP_UNROLL
    for (int j = 0; j < W; j++) r.w[j] = a.w[j] & ~b.w[j];
    return r;
}
template <int W>
TRP_HD G3BitsT<W> g3b_or(const G3BitsT<W>& a, const G3BitsT<W>& b) {
    G3BitsT<W> r;
    TRP_UNROLL
    for (int j = 0; j < W; j++) r.w[j] = a.w[j] | b.w[j];
    return r;
}
template <int W>
TRP_HD G3BitsT<W> g3b_xnor(const G3BitsT<W>& a, const G3BitsT<W>& b) {   // ~(a ^ b)
    G3BitsT<W> r;
    TRP_UNROLL
    for (int j = 0; j < W; j++) r.w[j] = ~(a.w[j] ^ b.w[j]);
    return r;
}
template <int W>
TRP_HD G3BitsT<W> g3b_orn(const G3BitsT<W>& a, const G3BitsT<W>& b) {   // a | ~b
    G3BitsT<W> r;
    TRP_UNROLL
    for (int j = 0; j < W; j++) r.w[j] = a.w[j] | ~b.w[j];
    return r;
}
template <int W>
TRP_HD bool g3b_any(const G3BitsT<W>& a) {
    u64 x = 0;
    TRP_UNROLL
    for (int j = 0; j < W; j++) x |= a.w[j];
    return x != 0;
}
template <int W>
TRP_HD u32 g3b_popc_and(const G3BitsT<W>& a, const G3BitsT<W>& b) {
    u32 n = 0;
    TRP_UNROLL
    for (int j = 0; j < W; j++) n += (u32)popc(a.w[j] & b.w[j]);
    return n;
}
// first bytes / last bytes of the runs of ones
template <int W>
TRP_HD G3BitsT<W> g3b_run_starts(const G3BitsT<W>& x) {
    G3BitsT<W> r;
    u64 c = 0;
    TRP_UNROLL
    for (int j = 0; j < W; j++) {
        r.w[j] = x.w[j] & ~((x.w[j] << 1) | c);
        c = x.w[j] >> 63;
    }
    return r;
}
template <int W>
TRP_HD G3BitsT<W> g3b_run_ends(const G3BitsT<W>& x) {
    G3BitsT<W> r;
    TRP_UNROLL
    for (int j = 0; j < W; j++) {
        const u64 nx = j + 1 < W ? x.w[j + 1] << 63 : 0ull;
        r.w[j] = x.w[j] & ~((x.w[j] >> 1) | nx);
    }
    return r;
}
// the bytes from each point (at most one per run, inside a run of x) to the end of its run
template <int W>
TRP_HD G3BitsT<W> g3b_run_from(const G3BitsT<W>& x, const G3BitsT<W>& points) { return g3b_andn(x, g3b_add(x, points)); }

// Keeps the compiler from carrying loop-invariant derived masks (mirror images, run starts) in registers
// across the expansion loop: they are cheap to recompute and the kernel's occupancy depends on it.
template <int W>
TRP_HD void g3b_opaque(G3BitsT<W>& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    TRP_UNROLL
    for (int j = 0; j < W; j++) asm volatile("" : "+v"(x.w[j]));
#else
    (void)x;
#endif
}
// The bytes of the tokens that strip_digit_tokens keeps (denoiser_preprocessor.rs:70-140).  X: bytes of
// the line's tokens; DG / AL: its ASCII digits / alphabetic characters (one bit per character).
// V is a bit vector type with the g3b_* operations: G3BitsT<W> (a line in the registers of one lane), or a
// vector spread over the lanes of a wavefront.
template <class V, class RT>
TRP_HD V g3_bits_solve(V X, const V& DG, const V& AL, const RT& rt) {
    // seed: the tokens with a letter.  P = the first letter of each such token (the carry from the token
    // start runs through the letterless bytes Y and stops on a letter or on the separator after the token)
    V K;
    {
        const V Y = g3b_andn(X, AL);
        const V P = g3b_and(g3b_andn(g3b_add(Y, g3b_run_starts(X)), Y), AL);
        K = g3b_or(g3b_run_from(X, P), g3b_rev(g3b_run_from(g3b_rev(X), g3b_rev(P))));
    }
    const u32 a0 = g3b_popc_and(AL, X);
    u32 d0 = g3b_popc_and(DG, K);
    for (;;) {
        g3b_opaque(X);
        // Z: kept bytes and everything that is not a token (K is made of whole tokens of X, so K | ~X is the
        // complement of X ^ K).  Adding K to Z sends a carry up from every kept token through the rest of its run
        // of Z - the token itself, the separators after it, further kept tokens - onto the first byte of the next
        // token that is not kept; the mirror image finds the last byte of the previous one.
        const V Z = g3b_xnor(X, K);
        const V NS = g3b_andn(g3b_add(Z, K), Z);
        V Wv = g3b_run_from(X, NS);
        {
            const V rZ = g3b_rev(Z);
            const V rPE = g3b_andn(g3b_add(rZ, g3b_rev(K)), rZ);
            Wv = g3b_or(Wv, g3b_rev(g3b_run_from(g3b_rev(X), rPE)));
        }
        if (!g3b_any(Wv)) break;
        const u32 wd = g3b_popc_and(DG, Wv);
        if (rt.exceeds(d0 + wd, a0)) break;
        d0 += wd;
        K = g3b_or(K, Wv);
    }
    return K;
}

// The same for a line given by its place in a row of block words: the line occupies bits [s, e) of the row
// (word k = bits 64k .. 64k+63), 0 < e - s <= 64 * W.  src(k) returns the masks of word k of the row (token
// bytes, digits, letters; anything outside the line is ignored, words the caller does not have are zero).  The
// line is moved down to bit 0, solved in W words, and the answer moved back: K[j] / KS[j] are the kept bytes /
// the first bytes of the kept tokens in word (s >> 6) + j of the row.
struct G3Src {
    u64 x, dg, al;
};
template <int W, class F, class RT>
TRP_HD void g3_solve_shifted(F src, u32 s, u32 e, const RT& rt, u64 (&K)[W + 1], u64 (&KS)[W + 1]) {
    const u32 w0 = s >> 6, len = e - s;
    const int o = (int)(s & 63u);
    G3BitsT<W> X, DG, AL;
    G3Src cur = src(w0);
    TRP_UNROLL
    for (int j = 0; j < W; j++) {
        const G3Src nx = src(w0 + (u32)j + 1u);
        const u32 lo = 64u * (u32)j;
        const u64 m = len >= lo + 64u ? ~0ull : (len > lo ? below((int)(len - lo)) : 0ull);
        X.w[j] = (o ? (cur.x >> o) | (nx.x << (64 - o)) : cur.x) & m;
        DG.w[j] = (o ? (cur.dg >> o) | (nx.dg << (64 - o)) : cur.dg) & m;
        AL.w[j] = (o ? (cur.al >> o) | (nx.al << (64 - o)) : cur.al) & m;
        cur = nx;
    }
    const G3BitsT<W> Kb = g3_bits_solve(X, DG, AL, rt);
    const G3BitsT<W> Sb = g3b_run_starts(Kb);   // kept tokens are whole tokens: the runs of Kb
    TRP_UNROLL
    for (int j = 0; j <= W; j++) {
        const u64 a = j < W ? Kb.w[j < W ? j : 0] : 0ull, b = j > 0 ? Kb.w[j > 0 ? j - 1 : 0] : 0ull;
        const u64 sa = j < W ? Sb.w[j < W ? j : 0] : 0ull, sb = j > 0 ? Sb.w[j > 0 ? j - 1 : 0] : 0ull;
        K[j] = o ? (a << o) | (b >> (64 - o)) : a;
        KS[j] = o ? (sa << o) | (sb >> (64 - o)) : sa;
    }
}

}  // namespace trp
