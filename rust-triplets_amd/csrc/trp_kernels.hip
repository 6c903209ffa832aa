// HIP kernels (gfx950) and C ABI of the denoise -> tokenise -> chunk path.
//
// Data layout in HBM (one batch = N bytes, R records, nB = ceil(N/64) blocks, nT = ceil(N/16384) tiles)
//   bytes[N], offsets[R+1]        caller's packed corpus
//   rec_lo[nT+2]                  first record that starts at/after each tile start (+ one for N)
//   seedJ[nT], seedS[nT]          what enters each tile from the left / right (k_seed, or k_tile_carry)
//   SA[nT]                        per-tile forward summary (struct J) written by k_summary (fallback path)
//   plan: ct[nB] (2 x u64)        emit plan bit planes C and TOK of every block, written by k_plan / k_gate3
//         px[nB] (2 x u64)        the planes PSP and PNL, only of the blocks that have or may get a separator
//         pfx_mask[nB/64]         which blocks those are (one bit per block)
//   g3list[]                      digit-heavy lines that k_plan could not solve inside a tile (position | flags)
//   rec_lo_w[nW+2]                the same per 2 KiB sub-tile (nW = ceil(N/2048)): k_emit works one wavefront per sub-tile
//   tile_bytes[nW], tile_toks[nW], tile_flags[nW]   emitted bytes / token starts per sub-tile
//   tile_base[nW+1]               exclusive prefix of (bytes | toks<<32)
//   denoised[D], tok_start[T], tok_len[T], *_offsets[R+1], chunks[C]   results
//
// Kernel sequence with the denoiser enabled:
//   k_rec_lo -> k_seed (-> k_summary -> k_tile_carry when k_seed left tiles open) -> k_plan -> k_gate3_solve
//   -> scan of the sub-tile counts (2) -> k_emit
//   -> scan of the chunk counts (2; totals, trailing offsets and dropped flags on the way) -> k_chunk_emit
// and with it disabled: k_rec_lo -> k_count_tokens -> scan(2) -> k_emit<passthrough> ...
//
// The per-lane logic lives in trp_lane.h (shared with the host simulator).
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "triplets_amd.h"
#include "trp_lane.h"

#define TRP_TABLE_QUAL static __device__ const
#include "unicode_tables.inc"

using namespace trp;

namespace {

constexpr int NT = TILE_BLOCKS;   // threads per workgroup
// k_emit works on sub-tiles of 32 blocks (2 KiB): one wavefront per sub-tile, one lane per half block (32 bytes,
// 32-bit masks), which keeps a wave's LDS and registers small enough for eight waves per SIMD.  The emitted
// byte / token counts and their scanned bases are kept per sub-tile (= per half wavefront of k_plan).
constexpr int SUB_BLOCKS = 32;
constexpr int SUB = SUB_BLOCKS * BLOCK;
constexpr int HALF = 32;   // bytes per lane of k_emit

__device__ inline UniTables dev_tables() {
    UniTables ut = {trp_alpha_stage1, trp_alpha_stage2, TRP_ALPHA_STAGE1_LEN, nullptr};
    return ut;
}

// Alphabetic lookups: code points below U+0800 (Latin-1, Greek, Cyrillic, Hebrew, Arabic ...) are
// served from a 256-byte LDS copy of their eight bitmap rows, the uniform CJK / Hangul ranges by rule
// (cp_is_alpha_high); only the rest goes to the table in global memory.
struct UniShared {
    u32 low[8 * 8];   // bitmap rows of code point blocks 0..7, in block order
};
__device__ inline void stage_tables(UniShared& u) {   // needs a barrier before use
    if (threadIdx.x < 64) {
        u32 row = trp_alpha_stage1[threadIdx.x >> 3];
        u.low[threadIdx.x] = trp_alpha_stage2[row * 8 + (threadIdx.x & 7)];
    }
}
__device__ inline UniTables lds_tables(const UniShared& u) {
    UniTables ut = {trp_alpha_stage1, trp_alpha_stage2, TRP_ALPHA_STAGE1_LEN, u.low};
    return ut;
}

// ---------------------------------------------------------------------------------------------------
// wave / workgroup primitives
// ---------------------------------------------------------------------------------------------------
template <class T>
__device__ inline T shfl_up_pod(const T& v, int delta) {
    static_assert(sizeof(T) % 4 == 0, "pod size");
    T r;
    const u32* s = reinterpret_cast<const u32*>(&v);
    u32* d = reinterpret_cast<u32*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); i++) d[i] = (u32)__shfl_up((int)s[i], delta, 64);
    return r;
}
template <class T>
__device__ inline T shfl_down_pod(const T& v, int delta) {
    static_assert(sizeof(T) % 4 == 0, "pod size");
    T r;
    const u32* s = reinterpret_cast<const u32*>(&v);
    u32* d = reinterpret_cast<u32*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); i++) d[i] = (u32)__shfl_down((int)s[i], delta, 64);
    return r;
}

// Ordered reduction over the 64 lanes of a wave: lane 0 receives op(v0, v1, ..., v63).
template <class T, class Op>
__device__ inline T wave_reduce_ordered(T v, Op op) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T o = shfl_down_pod(v, off);
        v = op(v, o);
    }
    return v;
}

// Exclusive forward scan over the workgroup (thread order), seeded with `seed` on the left.
// smem: NT/64 slots.  Every thread must call it.  *total receives op(seed, all values).
template <class T, class Op>
__device__ inline T block_scan_excl_fwd(T v, Op op, T identity, T seed, T* smem, T* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T o = shfl_up_pod(inc, off);
        if (lane >= off) inc = op(o, inc);
    }
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    T pre = seed;
    for (int w = 0; w < wave; w++) pre = op(pre, smem[w]);
    T ex = shfl_up_pod(inc, 1);
    if (lane == 0) ex = identity;
    T res = op(pre, ex);
    if (total) {
        T t = seed;
        for (int w = 0; w < NT / 64; w++) t = op(t, smem[w]);
        *total = t;
    }
    __syncthreads();
    return res;
}

// Orders the LDS accesses of the lanes of one wavefront: everything the wave did before this point (in any
// lane, whatever branch it sat in) is complete and visible before anything after it starts.  The wave
// barrier is a reconvergence point the compiler may not move memory accesses across; the fence makes the
// hardware wait for the outstanding LDS operations.
__device__ inline void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ inline u32 readlane_u32(u32 v, int lane) { return (u32)__builtin_amdgcn_readlane((int)v, lane); }
// This lane's bit of a mask that is the same in every lane (one bit per lane, as a ballot returns it).
__device__ inline u32 lane_bit(u64 uniform_mask) { return (u32)(uniform_mask >> (threadIdx.x & 63)) & 1u; }

// ---- line carries of a tile from wave prefix sums and ballots ----------------------------------------
// What a lane has to know about the blocks around it (LaneCarry) is a segmented sum over the lanes back
// to the last block with a line start, and forward to the next one.  With x = the part of a block after
// its last line start (the whole block when it has none) and h = the part before its first line start,
// both are differences of plain prefix sums over the wave (DPP adds), the flag part of a LineSum and the
// record-level flags are read off ballots, and only the four wave totals go through LDS.

// Inclusive prefix sum over the 64 lanes of a wave (row shifts inside the 16-lane rows, then the row
// totals are broadcast into the following rows).
__device__ inline u32 wave_incl_scan_add(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ inline u32 lane_read(u32 v, int src_lane) {   // any lane reads any lane
    return (u32)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

struct FlagBallots { u64 nz, first_pipe, last_pipe, seen; };
__device__ inline FlagBallots flag_ballots(const LineSum& x) {
    FlagBallots B = {__ballot(ls_first(x) != 0), __ballot(ls_first(x) == 1u), __ballot(ls_last(x) == 1u),
                     __ballot(ls_seen_text(x))};
    return B;
}
// f field of the sum of the LineSums of the lanes in m
__device__ inline u32 flags_of(const FlagBallots& B, u64 m) {
    const u64 nz = B.nz & m;
    if (!nz) return 0;
    const u32 first = ((B.first_pipe >> ctz(nz)) & 1ull) ? 1u : 2u;
    const u32 last = ((B.last_pipe >> top(nz)) & 1ull) ? 1u : 2u;
    return first | (last << 2) | ((B.seen & m) ? 16u : 0u);
}
__device__ inline LineSum unpack_ls(u32 da, u32 p, u32 f) {
    LineSum r = {da & 0xFFFFu, da >> 16, p, f};
    return r;
}

struct WaveLines {
    u32 Ix_da, Ix_p;                // inclusive prefix sums of x over the lanes: digits | alpha << 16, pipes
    u32 Z_da, Z_p;                  // exclusive prefix sum of x plus the lane's own h: the sum of h over the lanes
                                    // up to this one when no lane before it has a line start (then x = h there)
    u32 x_da, x_p;                  // the lane's own x
    u64 lsm, rsm, gah, gat, c0m, c1m;
    FlagBallots xb, hb;
};

// The ballots of the record / line flags alone (what the short-line path below needs of WaveLines).
__device__ inline void wave_ballots(const J& j, WaveLines& W) {
    W.lsm = __ballot((j.g & G_LS) != 0);
    W.rsm = __ballot((j.g & G_RS) != 0);
    W.gah = __ballot((j.g & G_AH) != 0);
    W.gat = __ballot((j.g & G_AT) != 0);
    W.c0m = __ballot((j.g & G_C0) != 0);
    W.c1m = __ballot((j.g & G_C1) != 0);
}

__device__ inline WaveLines wave_lines(const J& j) {
    WaveLines W;
    const bool has_ls = (j.g & G_LS) != 0;
    LineSum x;   // field by field: a struct select goes through scratch memory
    x.d = has_ls ? j.tail.d : j.head.d;
    x.a = has_ls ? j.tail.a : j.head.a;
    x.p = has_ls ? j.tail.p : j.head.p;
    x.f = has_ls ? j.tail.f : j.head.f;
    W.x_da = x.d | (x.a << 16);
    W.x_p = x.p;
    W.Ix_da = wave_incl_scan_add(W.x_da);
    W.Ix_p = wave_incl_scan_add(W.x_p);
    W.Z_da = W.Ix_da - W.x_da + (j.head.d | (j.head.a << 16));
    W.Z_p = W.Ix_p - W.x_p + j.head.p;
    W.lsm = __ballot(has_ls);
    W.rsm = __ballot((j.g & G_RS) != 0);
    W.gah = __ballot((j.g & G_AH) != 0);
    W.gat = __ballot((j.g & G_AT) != 0);
    W.c0m = __ballot((j.g & G_C0) != 0);
    W.c1m = __ballot((j.g & G_C1) != 0);
    W.xb = flag_ballots(x);
    W.hb = flag_ballots(j.head);
    return W;
}

// The summary of the lanes [lo, hi) of the wave (uniform); equals the ordered fold of j_combine over them.
__device__ inline J wave_range_total(const WaveLines& W, int lo, int hi) {
    J r;
    u32 g = 0;
    const u64 m = below(hi) & ~below(lo);
    const u64 lsm = W.lsm & m, rsm = W.rsm & m, gah = W.gah & m, gat = W.gat & m;
    // the lanes of the range before its first line start have x = h, so sums of h are differences of Ix / Z
    const u32 h_lo_da = lo > 0 ? readlane_u32(W.Ix_da, lo - 1) : 0u, h_lo_p = lo > 0 ? readlane_u32(W.Ix_p, lo - 1) : 0u;
    if (!lsm) {
        r.head = unpack_ls(readlane_u32(W.Ix_da, hi - 1) - h_lo_da, readlane_u32(W.Ix_p, hi - 1) - h_lo_p,
                           flags_of(W.hb, m));
        r.tail = ls_empty();
    } else {
        const int f = ctz(lsm), l = top(lsm);
        r.head = unpack_ls(readlane_u32(W.Z_da, f) - h_lo_da, readlane_u32(W.Z_p, f) - h_lo_p,
                           flags_of(W.hb, m & below(f + 1)));
        const u32 before_da = l > 0 ? readlane_u32(W.Ix_da, l - 1) : 0u, before_p = l > 0 ? readlane_u32(W.Ix_p, l - 1) : 0u;
        r.tail = unpack_ls(readlane_u32(W.Ix_da, hi - 1) - before_da, readlane_u32(W.Ix_p, hi - 1) - before_p,
                           flags_of(W.xb, m & ~below(l)));
        g |= G_LS;
        const bool c0l = (W.c0m >> l) & 1ull, c1l = (W.c1m >> l) & 1ull;
        const u64 rs_before = rsm & below(l);
        bool known;
        if (rs_before) {
            const int r0 = top(rs_before);
            known = ((W.gat >> r0) & 1ull) || (gah & below(l) & ~below(r0 + 1)) != 0;
        } else {
            known = (gah & below(l)) != 0;
            if (c1l) g |= G_C1;
        }
        if (c0l || (c1l && known)) g |= G_C0;
    }
    if (!rsm) {
        if (gah) g |= G_AH | G_AT;
    } else {
        g |= G_RS;
        if (gah & below(ctz(rsm) + 1)) g |= G_AH;
        if (gat & ~below(top(rsm))) g |= G_AT;
    }
    r.g = g;
    return r;
}
__device__ inline J wave_total(const WaveLines& W) { return wave_range_total(W, 0, 64); }

// ---- the same carries for a wave of short lines -------------------------------------------------------
// When every lane has a line start among the three lanes before it and among the three lanes after it (lines
// of at most about 250 bytes: the usual case), a lane reaches everything it needs with three DPP shifts in each
// direction: no prefix sums, no flag ballots.  A LineSum travels in one register: d | a << 8 | p << 16 | f << 24
// (sums of up to three blocks stay below 256), with the block's "has a line start" flag at bit 29.
constexpr u32 PK_LS = 1u << 29, PK_SUMS = 0x00FFFFFFu;
__device__ inline u32 from_prev_lane(u32 v) {   // lane L reads lane L - 1, lane 0 reads 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);   // wave_shr:1
}
__device__ inline u32 from_next_lane(u32 v) {   // lane L reads lane L + 1, lane 63 reads 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);   // wave_shl:1
}
// ls_add over three packed parts of one line, a the earliest (absent parts are 0)
__device__ inline u32 pk_join3(u32 a, u32 b, u32 c) {
    constexpr u32 F = 3u << 24, LA = 12u << 24, SEEN = 16u << 24;
    const u32 sums = (a & PK_SUMS) + (b & PK_SUMS) + (c & PK_SUMS);
    const u32 first = (a & F) ? a : ((b & F) ? b : c);
    const u32 last = (c & LA) ? c : ((b & LA) ? b : a);
    return sums | (first & F) | (last & LA) | ((a | b | c) & SEEN);
}
__device__ inline LineSum unpack_pk(u32 v) {
    LineSum r = {v & 0xFFu, (v >> 8) & 0xFFu, (v >> 16) & 0xFFu, (v >> 24) & 31u};
    return r;
}
__device__ inline bool lines_are_near(u64 lsm) {   // uniform
    const u64 back = (lsm << 1) | (lsm << 2) | (lsm << 3) | 0xFull;
    const u64 fwd = (lsm >> 1) | (lsm >> 2) | (lsm >> 3) | (0xFull << 60);
    return (back & fwd) == ~0ull;
}
struct NearLines {
    u32 ci, co;        // the sums over the lanes of this wave only (packed)
    u32 prev_ls;       // bit k: lane L - 1 - k has a line start (k = 0, 1)
};
__device__ inline NearLines near_lines(bool has_ls, u32 own_head, u32 own_tail) {
    NearLines Q;
    const u32 lsbit = has_ls ? PK_LS : 0u;
    const u32 xp = (has_ls ? own_tail : own_head) | lsbit, hp = own_head | lsbit;
    const u32 v1 = from_prev_lane(xp), v2 = from_prev_lane(v1), v3 = from_prev_lane(v2);
    const bool b2 = !(v1 & PK_LS), b3 = b2 && !(v2 & PK_LS);
    Q.ci = pk_join3(b3 ? v3 : 0u, b2 ? v2 : 0u, v1);
    Q.prev_ls = ((v1 >> 29) & 1u) | ((v2 >> 28) & 2u);
    const u32 u1 = from_next_lane(hp), u2 = from_next_lane(u1), u3 = from_next_lane(u2);
    const bool f2 = !(u1 & PK_LS), f3 = f2 && !(u2 & PK_LS);
    Q.co = pk_join3(u1, f2 ? u2 : 0u, f3 ? u3 : 0u);
    return Q;
}
// the wave's summary (uniform); equals wave_total() of the general path
__device__ inline J near_total(const WaveLines& W, const NearLines& Q, u32 own_head, u32 own_tail) {
    J r;
    const int f = ctz(W.lsm), l = top(W.lsm);
    r.head = ls_add(unpack_pk(readlane_u32(Q.ci, f)), unpack_pk(readlane_u32(own_head, f)));
    r.tail = ls_add(unpack_pk(readlane_u32(own_tail, l)), unpack_pk(readlane_u32(Q.co, l)));
    u32 g = G_LS;
    const bool c0l = (W.c0m >> l) & 1ull, c1l = (W.c1m >> l) & 1ull;
    const u64 rs_before = W.rsm & below(l);
    bool known;
    if (rs_before) {
        const int r0 = top(rs_before);
        known = ((W.gat >> r0) & 1ull) || (W.gah & below(l) & ~below(r0 + 1)) != 0;
    } else {
        known = (W.gah & below(l)) != 0;
        if (c1l) g |= G_C1;
    }
    if (c0l || (c1l && known)) g |= G_C0;
    if (!W.rsm) {
        if (W.gah) g |= G_AH | G_AT;
    } else {
        g |= G_RS;
        if (W.gah & below(ctz(W.rsm) + 1)) g |= G_AH;
        if (W.gat & ~below(top(W.rsm))) g |= G_AT;
    }
    r.g = g;
    return r;
}

// Field-wise select on values (a select between two members of one struct would otherwise be turned
// into an indexed load from a stack copy of the struct).
__device__ inline u32 sel32(bool c, u32 a, u32 b) {
    const u32 m = 0u - (u32)c;
    return (a & m) | (b & ~m);
}
__device__ inline LineSum ls_select(bool c, LineSum a, LineSum b) {
    LineSum r = {sel32(c, a.d, b.d), sel32(c, a.a, b.a), sel32(c, a.p, b.p), sel32(c, a.f, b.f)};
    return r;
}
// The lane's carries.  Pw = summary of everything before the wave back to the start of the record that
// covers the wave's first byte, Xw = backward summary of what follows the wave (both uniform).
__device__ inline LaneCarry wave_carry(const WaveLines& W, const J& Pw, const S& Xw, bool next_rs) {
    const int L = threadIdx.x & 63;
    LaneCarry k;
    const u64 beforeL = below(L);
    // -- the current line from its start to this block
    const u64 sb = W.lsm & beforeL;
    const int s = sb ? top(sb) : 0;
    const u32 Ex_da = W.Ix_da - W.x_da, Ex_p = W.Ix_p - W.x_p;   // exclusive prefix
    const u32 at_s_da = lane_read(Ex_da, s), at_s_p = lane_read(Ex_p, s);
    const bool pw_ar = (Pw.g & G_RS) ? (Pw.g & G_AT) != 0 : (Pw.g & G_AH) != 0;
    auto ar_at = [&](int n) -> bool {   // the record has an alphabetic char before block n of the wave
        const u64 bn = below(n);
        const u64 rb = W.rsm & bn;
        if (rb) {
            const int r = top(rb);
            return ((W.gat >> r) & 1ull) || (W.gah & bn & ~below(r + 1)) != 0;
        }
        return pw_ar || (W.gah & bn) != 0;
    };
    if (sb) {
        k.ci = unpack_ls(Ex_da - at_s_da, Ex_p - at_s_p, flags_of(W.xb, beforeL & ~below(s)));
        k.ci_rab = ((W.c0m >> s) & 1ull) || (((W.c1m >> s) & 1ull) && ar_at(s));
    } else {
        const LineSum own = unpack_ls(Ex_da, Ex_p, flags_of(W.xb, beforeL));
        k.ci = ls_add(ls_select((Pw.g & G_LS) != 0, Pw.tail, Pw.head), own);
        k.ci_rab = (Pw.g & G_C0) != 0;
    }
    k.ar_in = ar_at(L);
    // -- the rest of the block's last line after this block
    const u64 afterL = ~below(L + 1);
    const u64 eb = W.lsm & afterL;
    const int e = eb ? ctz(eb) : 63;
    // sum of h over the lanes (L, e]: the lanes between have no line start (x = h), lane e adds its own h
    // (with no line start after this lane it is the sum of x over the remaining lanes)
    // (the cross-lane reads are made by every lane, outside the selects: a lane may be another lane's source)
    const u32 ze_da = lane_read(W.Z_da, e), ze_p = lane_read(W.Z_p, e);
    const u32 all_da = readlane_u32(W.Ix_da, 63), all_p = readlane_u32(W.Ix_p, 63);
    const u32 at_e_da = sel32(eb != 0, ze_da, all_da), at_e_p = sel32(eb != 0, ze_p, all_p);
    const LineSum fwd = unpack_ls(at_e_da - W.Ix_da, at_e_p - W.Ix_p, flags_of(W.hb, afterL & below(e + 1)));
    k.co = eb ? fwd : ls_add(fwd, Xw.head);
    k.next_rs = next_rs;
    return k;
}

// wave_carry() for a wave of short lines (lines_are_near): own_g = the lane's own J::g
__device__ inline LaneCarry near_carry(const WaveLines& W, const NearLines& Q, u32 own_g, const J& Pw, const S& Xw,
                                       bool next_rs) {
    const int L = threadIdx.x & 63;
    LaneCarry k;
    const int f = ctz(W.lsm), l = top(W.lsm);
    const bool pw_ar = (Pw.g & G_RS) ? (Pw.g & G_AT) != 0 : (Pw.g & G_AH) != 0;
    // The record has an alphabetic char before this block: a prefix OR over the lanes that starts afresh at every
    // lane with a record start, for all lanes at once as the carries of one addition on the ballots (uniform
    // values, scalar instructions).  A lane with a record start hands on "alpha at or after my last record start"
    // and nothing of what came before; any other lane hands on what it got, or that it has an alpha itself.
    {
        const u64 kill = W.rsm & ~W.gat, gen = (W.rsm & W.gat) | (~W.rsm & W.gah);
        const u64 X = ~kill, Y = gen;
        const u64 sum = X + Y + (pw_ar ? 1ull : 0ull);
        k.ar_in = lane_bit(sum ^ X ^ Y) != 0;   // the carry into this lane
    }
    // "the record has an alphabetic char before this line's start", as the lane that holds the line start sees it
    const u32 rab_own = ((own_g & G_C0) != 0 || ((own_g & G_C1) != 0 && k.ar_in)) ? 1u : 0u;
    const u32 r1 = from_prev_lane(rab_own), r2 = from_prev_lane(r1), r3 = from_prev_lane(r2);
    const LineSum ci = unpack_pk(Q.ci);
    if (L <= f) {   // no line start in this wave before the lane: the line comes in from the left
        k.ci = ls_add(ls_select((Pw.g & G_LS) != 0, Pw.tail, Pw.head), ci);
        k.ci_rab = (Pw.g & G_C0) != 0;
    } else {
        k.ci = ci;
        k.ci_rab = ((Q.prev_ls & 1u) ? r1 : (Q.prev_ls & 2u) ? r2 : r3) != 0;
    }
    const LineSum co = unpack_pk(Q.co);
    k.co = L >= l ? ls_add(co, Xw.head) : co;
    k.next_rs = next_rs;
    return k;
}

// A block's own LineSum in one register (d, a, p <= 64 and f < 32).
__device__ inline u32 pack_ls(const LineSum& x) { return x.d | (x.a << 8) | (x.p << 16) | (x.f << 24); }
__device__ inline LineSum unpack_ls8(u32 v) {
    LineSum r = {v & 0xFFu, (v >> 8) & 0xFFu, (v >> 16) & 0xFFu, v >> 24};
    return r;
}
// A summary that is the same in every lane (read from LDS at a uniform address), moved to scalar registers.
__device__ inline u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline LineSum uniform_ls(const LineSum& x) {
    LineSum r = {uni(x.d), uni(x.a), uni(x.p), uni(x.f)};
    return r;
}
__device__ inline J uniform_J(const J& j) {
    J r;
    r.head = uniform_ls(j.head);
    r.tail = uniform_ls(j.tail);
    r.g = uni(j.g);
    return r;
}
__device__ inline S uniform_S(const S& x) {
    S r;
    r.head = uniform_ls(x.head);
    r.has_ls = uni(x.has_ls);
    return r;
}

struct JOp { __device__ J operator()(const J& a, const J& b) const { return j_combine(a, b); } };
struct SOp { __device__ S operator()(const S& a, const S& b) const { return s_combine(a, b); } };

// ---------------------------------------------------------------------------------------------------
// tile staging
// ---------------------------------------------------------------------------------------------------
// LDS image of a tile: block i of the tile lives at (i + 1) * PAD_STRIDE (64 data bytes + 4 unused, see
// PadBlk); "block -1" and "block TILE_BLOCKS" hold the 16 halo bytes before / after the tile.
struct TileShared {
    alignas(16) u8 bytes[(TILE_BLOCKS + 2) * PAD_STRIDE];
};

__device__ inline void lds_store16(u8* dst, uint4 v) {   // dst is 4-byte aligned
    u32* d = reinterpret_cast<u32*>(dst);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}

__device__ inline uint4 load16_guarded(const u8* __restrict__ g, u64 N, u64 q) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (q + 16 <= N) v = *reinterpret_cast<const uint4*>(g + q);
    else if (q < N) {
        u32 w[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++)
            if (q + i < N) w[i >> 2] |= (u32)g[q + i] << (8 * (i & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return v;
}

// A tile [tS, tS+TILE) plus HALO bytes on both sides travels from global memory to the LDS image in two
// steps, so that the loads of the next tile can be in flight while the current one is processed.
struct TileRegs {
    uint4 v[TILE / 16 / NT];
    uint4 halo;   // thread 0: the 16 bytes before the tile; thread 1: the 16 bytes after it
};
__device__ inline void tile_load(const u8* __restrict__ g, u64 N, u64 tS, TileRegs& r) {
    const int tid = threadIdx.x;
    if (tS >= HALO && tS + TILE + HALO <= N) {
        // interior tile (uniform branch): straight-line loads, all in flight together
        const uint4* src = reinterpret_cast<const uint4*>(g + tS);
#pragma unroll
        for (int k = 0; k < TILE / 16 / NT; k++) r.v[k] = src[k * NT + tid];
        r.halo = make_uint4(0, 0, 0, 0);
        if (tid < 2) r.halo = *reinterpret_cast<const uint4*>(g + (tid == 0 ? tS - HALO : tS + TILE));
        return;
    }
#pragma unroll
    for (int k = 0; k < TILE / 16 / NT; k++) r.v[k] = load16_guarded(g, N, tS + (u64)(k * NT + tid) * 16);
    r.halo = make_uint4(0, 0, 0, 0);
    if (tid == 0) { if (tS >= HALO) r.halo = load16_guarded(g, N, tS - HALO); }
    else if (tid == 1) r.halo = load16_guarded(g, N, tS + TILE);
}
__device__ inline void tile_store(const TileRegs& r, u8* sb) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < TILE / 16 / NT; k++) {
        const int c = k * NT + tid;
        lds_store16(sb + ((c >> 2) + 1) * PAD_STRIDE + (c & 3) * 16, r.v[k]);
    }
    if (tid == 0) lds_store16(sb + 48, r.halo);                                  // last 16 bytes of block -1
    else if (tid == 1) lds_store16(sb + (TILE_BLOCKS + 1) * PAD_STRIDE, r.halo);
}
__device__ inline PadBlk lane_block(const TileShared& sh, int i) {
    PadBlk b = {sh.bytes + (i + 1) * PAD_STRIDE};
    return b;
}

// Record starts of a tile, one bit per byte, built in LDS from the offsets array: records
// [rec_lo[t], rec_lo[t+1]) start inside tile t.  w[i] covers block i of the tile; w[TILE_BLOCKS] bit 0
// says whether the byte right after the tile starts a record.  Run by wave 0 only (LDS operations of one
// wave complete in order: clear, then set), before the barrier that publishes the tile image.
struct RsShared {
    u64 w[TILE_BLOCKS + 1];
};
// Two consecutive entries of rec_lo / rec_lo_w as a range of records.  With offsets that are not
// non-decreasing (invalid input) k_rec_lo leaves entries unwritten or out of order: whatever they hold, the range
// stays inside [0, R] and is never negative, so that no loop over it runs away.
struct RecRange { u32 r0, r1; };
__device__ inline RecRange rec_range(const u32* __restrict__ lo, u64 i, u64 R) {
    u32 r0 = lo[i], r1 = lo[i + 1];
    if (r1 > (u32)R) r1 = (u32)R;
    if (r0 > r1) r0 = r1;
    RecRange g = {r0, r1};
    return g;
}
struct RsFetch {
    u32 q;   // wave 0: start offset of record r0 + lane, r1 included (or ~0; a batch is shorter than 4 GiB)
    u32 n;   // wave 0: r1 - r0
};
// Global loads of the record starts of tile t, issued together with the tile's bytes.  Returns r0.
__device__ inline u32 rs_fetch(const u64* __restrict__ offsets, const u32* __restrict__ rec_lo, u64 R, u64 t,
                               RsFetch& f) {
    f.q = ~0u;
    f.n = 0;
    if (threadIdx.x >= 64) return 0;
    const RecRange g = rec_range(rec_lo, t, R);
    const u32 r0 = g.r0, r1 = g.r1;
    const u64 r = (u64)r0 + threadIdx.x;
    f.n = r1 - r0;
    if (r <= r1 && r < R) f.q = (u32)offsets[r];
    return r0;
}
// A start at tS + TILE (the byte after the tile) lands in bit 0 of the extra word.
__device__ inline void rs_set(RsShared& rs, u64 q, u64 N, u64 tS) {
    const u64 d = q - tS;
    if (q < N && d <= (u64)TILE) atomicOr(reinterpret_cast<unsigned long long*>(&rs.w[d >> 6]), 1ull << (d & 63));
}
__device__ inline void rs_store(const RsFetch& f, const u64* __restrict__ offsets, const u32* __restrict__ rec_lo,
                                u64 R, u64 N, u64 t, RsShared& rs) {
    if (threadIdx.x >= 64) return;
    const u64 tS = t * (u64)TILE;
    for (int i = threadIdx.x; i <= TILE_BLOCKS; i += 64) rs.w[i] = 0;
    if (f.q != ~0u) rs_set(rs, f.q, N, tS);
    if (f.n >= 64) {   // crowded tile: the rest of its records
        const u64 r0 = rec_range(rec_lo, t, R).r0, r1 = (u64)r0 + f.n;
        for (u64 r = r0 + 64 + threadIdx.x; r <= r1 && r < R; r += 64) rs_set(rs, offsets[r], N, tS);
    }
}

// Record starts inside the 64-byte block wb (wb < nB), straight from the offsets array (binary search; equal
// offsets of empty records are skipped in one step); *next says whether the byte after the block starts
// a record.  For the rare lines solved outside their tile.
__device__ inline u64 block_record_starts(const u64* __restrict__ offsets, const u32* __restrict__ rec_lo, u64 R,
                                          u64 N, u64 wb, bool* next) {
    const u64 lo = wb * 64;
    const RecRange g = rec_range(rec_lo, wb / TILE_BLOCKS, R);   // the block's tile narrows the search
    u64 a = g.r0, b = g.r1;
    while (a < b) {
        const u64 m = (a + b) >> 1;
        if (offsets[m] < lo) a = m + 1; else b = m;
    }
    u64 bits = 0;
    *next = false;
    while (a < R) {
        const u64 q = offsets[a];
        if (q >= N || q > lo + 64) break;
        if (q == lo + 64) { *next = true; break; }
        bits |= 1ull << (q - lo);
        u64 x = a + 1, y = R;   // first index whose offset is larger
        while (x < y) {
            const u64 m = (x + y) >> 1;
            if (offsets[m] <= q) x = m + 1; else y = m;
        }
        a = x;
    }
    return bits;
}

// The blocks next to a lane's block are the blocks of the lanes next to it (trp_lane.h, NoNeighbours).
struct WaveNeighbours {
    __device__ bool any(bool x) const { return __ballot(x) != 0; }
    __device__ bool has_next() const { return (threadIdx.x & 63) < 63; }
    __device__ bool has_prev() const { return (threadIdx.x & 63) > 0; }
    template <class B>
    __device__ u32 next_planes3(u32 mine, const B&) const { return from_next_lane(mine); }
    __device__ u32 swap_spill(u32 v) const { return from_prev_lane(v); }
    template <class B>
    __device__ bool leftover(const B&, u64, u64, const UniTables&, u64*, u64*) const { return false; }
};

// Lanes that hold consecutive blocks in groups (the windows of k_seed): the caller says whether the lanes next
// to this one belong to its group.
struct WindowNeighbours {
    bool next_in_group, prev_in_group;
    __device__ bool any(bool x) const { return __ballot(x) != 0; }
    __device__ bool has_next() const { return next_in_group; }
    __device__ bool has_prev() const { return prev_in_group; }
    template <class B>
    __device__ u32 next_planes3(u32 mine, const B&) const { return from_next_lane(mine); }
    __device__ u32 swap_spill(u32 v) const { return from_prev_lane(v); }
    template <class B>
    __device__ bool leftover(const B&, u64, u64, const UniTables&, u64*, u64*) const { return false; }
};

// ---- multi-byte characters handed out over the lanes --------------------------------------------------------
// Characters outside the bit-plane rules of classify_leads_planes (the less common scripts) would be decoded and
// looked up one after the other by the lane that owns them, the whole wavefront waiting for the lane with the
// most.  Instead the positions of the wave's unsettled lead bytes are listed, lane k decodes and looks up the k-th
// of them (64 at a time), and the results go back to the owning blocks through per-block masks in LDS.
constexpr u32 MB_LIST = 256;
struct MbScratch {
    u64 alp[64];                // lead bytes of the alphabetic characters of each block
    u64 ws[64];                 // bytes of the whitespace characters of each block
    uint16_t list[MB_LIST];     // block << 6 | byte of every listed lead byte of the wave, in order
};
// The lanes of a wavefront, lane i with block i of the wave's 64, sharing a scratch area.
struct WaveHandOut {
    MbScratch* ms;
    __device__ bool any(bool x) const { return __ballot(x) != 0; }
    __device__ bool has_next() const { return (threadIdx.x & 63) < 63; }
    __device__ bool has_prev() const { return (threadIdx.x & 63) > 0; }
    template <class B>
    __device__ u32 next_planes3(u32 mine, const B&) const { return from_next_lane(mine); }
    __device__ u32 swap_spill(u32 v) const { return from_prev_lane(v); }
    __device__ bool leftover(const PadBlk& blk, u64 lead, u64 V, const UniTables& ut, u64* ws, u64* alp) const {
        const int lane = threadIdx.x & 63;
        const u32 n = (u32)popc(lead);
        // uniform: with at most two characters in any lane the lanes are done sooner on their own (about 45
        // instructions per character there; listing, two synchronisations and a round here)
        if (__ballot(n > 2u) == 0) return false;
        const u32 inc = wave_incl_scan_add(n);
        const u32 M = readlane_u32(inc, 63);
        if (M > MB_LIST) return false;            // uniform: too many for the list, every lane takes its own
        ms->alp[lane] = 0;
        ms->ws[lane] = 0;
        u32 at = inc - n;
        for (u32 m = (u32)lead; m; m &= m - 1) ms->list[at++] = (uint16_t)((u32)(lane << 6) | (u32)__builtin_ctz(m));
        for (u32 m = (u32)(lead >> 32); m; m &= m - 1)
            ms->list[at++] = (uint16_t)((u32)(lane << 6) | 32u | (u32)__builtin_ctz(m));
        wave_lds_sync();
        const PadBlk wblk = {blk.p - lane * PAD_STRIDE};   // the wave's 64 blocks (and what follows) as one run of bytes
        for (u32 s0 = 0; s0 < M; s0 += 64) {   // uniform
            const u32 slot = s0 + (u32)lane;
            if (slot < M) {
                const u32 pos = ms->list[slot], ob = pos >> 6, bp = pos & 63u;
                const int k = classify_lead_at(wblk, (int)pos, ut);
                if (k > 0) {
                    atomicOr(reinterpret_cast<unsigned long long*>(&ms->alp[ob]), 1ull << bp);
                } else if (k < 0) {   // a whitespace character marks all its bytes, in the next block too
                    const u64 m = k == -2 ? 3ull : 7ull;
                    atomicOr(reinterpret_cast<unsigned long long*>(&ms->ws[ob]), m << bp);
                    if (bp > 61u && ob < 63u && (m >> (64u - bp)))
                        atomicOr(reinterpret_cast<unsigned long long*>(&ms->ws[ob + 1]), m >> (64u - bp));
                }
            }
        }
        wave_lds_sync();
        *alp |= ms->alp[lane];
        *ws |= ms->ws[lane] & V;
        return true;
    }
};

struct LaneCtx {
    Cls c;
    Derived d;
    u64 p;          // byte offset of the block in the corpus
    int nvalid;
    bool next_rs;   // the byte after the block starts a record or does not exist
    bool prev_tok;  // normalize mode: the byte before the block is a token byte of the same record
    u64 sp, nl;     // bytes that are exactly ' ' / '\n' (for plan_fold_prefixes)
};

// rs_words: record-start words of the tile's blocks (RsShared), complete before the call.
// mb: per-wavefront scratch for handing out the characters the bit-plane rules leave open (WaveHandOut); nullptr:
// every lane decodes its own.
__device__ inline LaneCtx lane_setup(const TileShared& sh, const UniTables& ut, u64 N, u64 tS, bool want_lines,
                                     const u64* rs_words, bool normalize = false, MbScratch* mb = nullptr) {
    LaneCtx L;
    const int tid = threadIdx.x;
    const PadBlk blk = lane_block(sh, tid);
    L.p = tS + (u64)tid * BLOCK;
    L.nvalid = L.p >= N ? 0 : (int)((N - L.p) < (u64)BLOCK ? (N - L.p) : (u64)BLOCK);
    if (normalize || !want_lines) {   // these modes look at whitespace only
        Cls z = {classify_ws_fast(blk, L.nvalid, &L.sp), 0, 0, 0, 0, 0, 0};
        L.c = z;
        L.nl = 0;
    } else {
        if (mb) {
            WaveHandOut nb = {mb};
            L.c = classify_block_fast(blk, L.nvalid, ut, nb);
        } else {
            WaveNeighbours nb;   // every lane of the wavefront is here, lane i with block i of its 64
            L.c = classify_block_fast(blk, L.nvalid, ut, nb);
        }
        L.sp = L.c.sp;
        L.nl = L.c.nl;
    }
    // read only now: the classifier needs every register it can get
    asm volatile("" ::: "memory");
    const u64 rs_here = rs_words[tid], rs_next = rs_words[tid + 1];
    u64 V = below(L.nvalid);
    u64 RS = rs_here & V;
    bool nrs = (rs_next & 1ull) != 0;
    bool has_next = L.p + BLOCK < N;
    L.next_rs = !has_next || nrs;
    L.prev_tok = false;
    if (normalize) {
        L.prev_tok = L.p > 0 && L.nvalid > 0 && !(RS & 1ull) && !byte_is_ws(blk, -1);
        L.c = normalize_cls(L.c, V);
        L.d = derive(L.c, V, RS, false, false);
    } else if (want_lines) {
        bool next_nl = has_next && blk[BLOCK] == '\n' && !nrs;
        bool pt = prev_is_term(L.p > 0, blk[-1], L.nvalid > 0, blk[0], (RS & 1ull) != 0);
        L.d = derive(L.c, V, RS, next_nl, pt);
    } else {
        L.d.V = V;
        L.d.RS = RS;
        L.d.LS = L.d.TERM = L.d.DEL = 0;
        L.d.LC = V;
    }
    return L;
}

// ---------------------------------------------------------------------------------------------------
// k_rec_lo: rec_lo[t] = number of records whose start offset is < t*TILE (t = 0..nT), and
// rec_lo[nT+1] = number of records whose start offset is < N.
// ---------------------------------------------------------------------------------------------------
// rec_lo_w: the same for sub-tiles (w = 0..nW, and rec_lo_w[nW+1] for N).
// The first kernel of a batch: it also clears the batch's totals and counters (totals[0..15]).
// One thread per gap between consecutive record starts: thread r (0..R) owns the keys in (offsets[r-1], offsets[r]]
// (everything up to offsets[0] for r = 0, everything above offsets[R-1] for r = R) and writes r, the number of
// record starts below such a key, to the entries whose key falls there.  Most gaps hold a few entries; a long
// one (a record of many tiles) is written by the whole wavefront.  Entries are clamped to the arrays, whatever
// the offsets hold.
__device__ inline void rec_lo_fill(u32* __restrict__ dst, u64 unit, u64 n_units, u64 N, bool has_lo, u64 lo, bool has_hi,
                                   u64 hi, u32 r) {
    // entries i in [first, last]: lo < i * unit <= hi; entry n_units + 1 belongs to the key N
    u64 first = has_lo ? lo / unit + 1 : 0;
    u64 last = has_hi ? hi / unit : n_units;
    if (last > n_units) last = n_units;
    if ((!has_lo || lo < N) && (!has_hi || N <= hi)) dst[n_units + 1] = r;
    constexpr u64 OWN = 8;
    const u64 count = first <= last ? last - first + 1 : 0;
    for (u64 i = 0; i < (count < OWN ? count : OWN); i++) dst[first + i] = r;
    u64 wide = __ballot(count > OWN);
    const int lane = threadIdx.x & 63;
    while (wide) {   // uniform
        const int l = ctz(wide);
        wide &= wide - 1;
        const u64 f = ((u64)readlane_u32((u32)(first >> 32), l) << 32) | readlane_u32((u32)first, l);
        const u64 e = ((u64)readlane_u32((u32)(last >> 32), l) << 32) | readlane_u32((u32)last, l);
        const u32 v = readlane_u32(r, l);
        for (u64 i = f + OWN + (u64)lane; i <= e; i += 64) dst[i] = v;
    }
}
// The offsets are checked on the way (offsets[0] = 0, non-decreasing, offsets[R] = N): a thread that finds them
// wrong writes the batch's number `seq` to totals[9], which is never cleared - an older number means nothing.
// (Every kernel of the path is a thin wrapper around a body function that takes its workgroup number as an
// argument: k_small_batch runs the same bodies one after the other in a single workgroup.)
__device__ __forceinline__ void rec_lo_body(u32 vb, const u64* __restrict__ offsets, u64 R, u64 N, u64 nT, u64 nW,
                                            u32* __restrict__ rec_lo, u32* __restrict__ rec_lo_w,
                                            u64* __restrict__ totals, u64 seq) {
    const u64 r = (u64)vb * 256 + threadIdx.x;
    if (r < 16 && r != 9) totals[r] = 0;
    const bool live = r <= R;   // whole wavefronts stay: the long gaps are written together
    const bool has_lo = live && r > 0, has_hi = live && r < R;
    const u64 lo = has_lo ? offsets[r - 1] : 0, hi = has_hi ? offsets[r] : 0;
    if ((has_lo && has_hi && lo > hi) || (r == 0 && offsets[0] != 0) || (r == R && offsets[R] != N)) totals[9] = seq;
    // a thread beyond R owns nothing: an empty gap
    rec_lo_fill(rec_lo, TILE, nT, N, live ? has_lo : true, live ? lo : ~0ull, live ? has_hi : true, live ? hi : 0, (u32)r);
    rec_lo_fill(rec_lo_w, SUB, nW, N, live ? has_lo : true, live ? lo : ~0ull, live ? has_hi : true, live ? hi : 0, (u32)r);
}
__global__ __launch_bounds__(256) void k_rec_lo(const u64* __restrict__ offsets, u64 R, u64 N, u64 nT, u64 nW,
                                                u32* __restrict__ rec_lo, u32* __restrict__ rec_lo_w,
                                                u64* __restrict__ totals, u64 seq) {
    rec_lo_body(blockIdx.x, offsets, R, N, nT, nW, rec_lo, rec_lo_w, totals, seq);
}

// ---------------------------------------------------------------------------------------------------
// k_summary: per-tile forward summary (line statistics at the tile edges, record-level alpha flags).
// ---------------------------------------------------------------------------------------------------
// k_summary: per-tile forward summary.  Runs only when k_seed left tiles open (*open_count != 0); the grid is
// a fixed number of workgroups that stride over the tiles, so that the usual case costs an empty launch.
constexpr unsigned SUMMARY_GRID = 4096;
template <bool NORMALIZE>
__global__ __launch_bounds__(NT) void k_summary(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                                 const u32* __restrict__ rec_lo, u64 R, u64 N,
                                                 u64 nB, u64 nT, const u32* __restrict__ open_count,
                                                 J* __restrict__ SA) {
    __shared__ TileShared sh;
    __shared__ UniShared uni;
    __shared__ J wsum[NT / 64];
    __shared__ RsShared rs;
    if (*open_count == 0) return;
    stage_tables(uni);
#pragma unroll 1
    for (u64 t = blockIdx.x; t < nT; t += gridDim.x) {
        const u64 tS = t * (u64)TILE;
        {
            TileRegs regs;
            RsFetch rf;
            tile_load(bytes, N, tS, regs);
            rs_fetch(offsets, rec_lo, R, t, rf);
            tile_store(regs, sh.bytes);
            rs_store(rf, offsets, rec_lo, R, N, t, rs);
        }
        __syncthreads();
        LaneCtx L = lane_setup(sh, lds_tables(uni), N, tS, true, rs.w, NORMALIZE);
        J j = wave_total(wave_lines(lane_J(L.c, L.d)));
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = j;
        __syncthreads();   // also: every lane has finished reading the image and the record starts
        if (threadIdx.x == 0) {
            J acc = wsum[0];
            for (int w = 1; w < NT / 64; w++) acc = j_combine(acc, wsum[w]);
            SA[t] = acc;
        }
        __syncthreads();   // wsum is free again
    }
}

// ---------------------------------------------------------------------------------------------------
// k_plan: emit plan of every block + per-tile counts.
// ---------------------------------------------------------------------------------------------------
enum : u32 { TF_FIRST_CONT = 1, TF_LAST_TOK = 2 };

// Look-back over the tile summaries to the tile that holds the start of the record covering the
// first byte of tile t (wave 0 only; result valid in lane 0).
__device__ inline J tile_lookback(const J* __restrict__ SA, u64 t) {
    const int lane = threadIdx.x & 63;
    J acc = j_identity();
    u64 hi = t;   // tiles [.., hi) still to be folded in front of acc
    while (hi > 0) {
        // lane l holds tile hi-64+l so that lane order is tile order
        long long idx = (long long)hi - 64 + lane;
        J v = idx >= 0 ? SA[idx] : j_identity();
        u64 m = __ballot((v.g & G_RS) != 0);
        int from = m ? top(m) : 0;
        if (lane < from) v = j_identity();
        v = wave_reduce_ordered(v, JOp());
        acc = j_combine(v, acc);   // meaningful in lane 0
        if (m) break;
        hi = hi > 64 ? hi - 64 : 0;
    }
    return acc;
}

// Look-ahead to the next line start after tile t (wave 0 only; result valid in lane 0).
__device__ inline S tile_lookahead(const J* __restrict__ SA, u64 t, u64 nT) {
    const int lane = threadIdx.x & 63;
    S acc = s_identity();
    u64 lo = t + 1;
    while (lo < nT) {
        u64 idx = lo + lane;
        S v = idx < nT ? s_from_J(SA[idx]) : s_identity();
        u64 m = __ballot(v.has_ls != 0);
        int upto = m ? ctz(m) : 63;
        if (lane > upto) v = s_identity();
        v = wave_reduce_ordered(v, SOp());
        acc = s_combine(acc, v);
        if (m) break;
        lo += 64;
    }
    return acc;
}

// Guarded view of a block of the corpus in global memory (zeros outside [0, n)).
struct GuardBlk {
    const u8* p;
    u64 base, n;
    __device__ u32 operator[](int j) const {
        long long q = (long long)base + j;
        return (q >= 0 && (u64)q < n) ? p[q] : 0u;
    }
    __device__ u32 word(int k) const {
        u64 q = base + 4 * (u64)k;
        if (q + 4 <= n) return *reinterpret_cast<const u32*>(p + q);
        u32 v = 0;
        for (int i = 0; i < 4; i++) if (q + i < n) v |= (u32)p[q + i] << (8 * i);
        return v;
    }
};

// k_seed: what enters a tile from the left and from the right, from the bytes next to it.  One wavefront
// per tile classifies SEED_BACK blocks before the tile and SEED_AHEAD blocks after it, straight from
// global memory, and folds them; the window is moved outwards while the answer is open, at most
// SEED_ROUNDS times.  The fold of a window that does not reach back to the record start is as good as
// the full one once it holds a line start with a letter somewhere before it (then "the record has a
// letter / a kept line before this line" is true whatever lies further back; normalize mode only asks for
// a token byte), and the right-hand side is settled by the first line start.  Tiles that stay open are
// counted; if there are any, k_summary / k_tile_carry redo the seeds of the whole batch from per-tile
// summaries (long lines, long stretches without a letter).
constexpr int SEED_TPW = 4;                       // tiles per wavefront
constexpr int SEED_LPT = 64 / SEED_TPW;           // lanes (= window blocks) per tile
constexpr int SEED_BACK = 12, SEED_AHEAD = SEED_LPT - SEED_BACK, SEED_ROUNDS = 16;
// Per tile of a wave: byte images of the two windows (one block of halo on both sides, PadBlk layout) and
// the record starts of their blocks (one extra word each for "the next block starts a record").
struct SeedTile {
    alignas(16) u8 back[(SEED_BACK + 2) * PAD_STRIDE];
    alignas(16) u8 ahead[(SEED_AHEAD + 2) * PAD_STRIDE];
    u64 rs_back[SEED_BACK + 1];
    u64 rs_ahead[SEED_AHEAD + 1];
};
struct SeedWave {
    SeedTile tile[SEED_TPW];
    // the folds so far (uniform values: kept here, not in registers, of which the classifier needs every one)
    J P[SEED_TPW];
    S X[SEED_TPW];
};

struct SeedShared {
    SeedWave shw[NT / 64];
};
// uni: the staged tables (stage_tables, complete before the call).
template <bool NORMALIZE>
__device__ __forceinline__ void seed_body(SeedShared& ssh, const UniShared& uni, u32 vb, const u8* __restrict__ bytes,
                                          const u64* __restrict__ offsets, const u32* __restrict__ rec_lo, u64 R, u64 N,
                                          u64 nB, u64 nT, u32 force_open, J* __restrict__ seedJ, S* __restrict__ seedS,
                                          u32* __restrict__ open_count) {
    const int lane = threadIdx.x & 63;
    SeedWave& sw = ssh.shw[threadIdx.x >> 6];
    const u64 t0 = ((u64)vb * (NT / 64) + (threadIdx.x >> 6)) * SEED_TPW;
    if (t0 >= nT) return;   // whole waves; no workgroup barrier below
    const UniTables ut = lds_tables(uni);
    u32 open_mask = 0;   // bit 2k: the back window of tile k is still open, bit 2k+1: its ahead window
#pragma unroll
    for (int k = 0; k < SEED_TPW; k++) {
        const u64 t = t0 + k;
        if (!(t == 0 || t >= nT)) open_mask |= 1u << (2 * k);
        // normalize mode has no lines: nothing of what follows the tile matters to it
        if (!(NORMALIZE || (t + 1) * TILE_BLOCKS >= nB)) open_mask |= 2u << (2 * k);   // closed also for t >= nT
    }
    if (lane < SEED_TPW) { sw.P[lane] = j_identity(); sw.X[lane] = s_identity(); }
    // this lane's place: tile k, window block li of the back or of the ahead window
    const int my_k = lane / SEED_LPT, l16 = lane % SEED_LPT;
    const bool is_back = l16 < SEED_BACK;
    const int li = is_back ? l16 : l16 - SEED_BACK;
    const long long my_tb0 = (long long)((t0 + my_k) * TILE_BLOCKS);
    // for the record starts: lane -> window (tile, side), SEED_RG_LANES lanes per window
    constexpr int SEED_RG_LANES = 64 / (2 * SEED_TPW);
    const int rg_k = lane / SEED_RG_LANES, rg_sub = lane % SEED_RG_LANES;
#pragma unroll 1
    for (int round = 0; round < SEED_ROUNDS; round++) {
        if (open_mask == 0) break;
        // ---- stage the open windows: bytes (16-byte chunks, all lanes) ...
        constexpr int CH_BACK = (SEED_BACK + 2) * 4, CH_TILE = CH_BACK + (SEED_AHEAD + 2) * 4;
        for (int c = lane; c < SEED_TPW * CH_TILE; c += 64) {
            const int k = c / CH_TILE, cc = c % CH_TILE;
            const bool bk = cc < CH_BACK;
            const int ci = bk ? cc : cc - CH_BACK;
            if (!((open_mask >> (2 * k + (bk ? 0 : 1))) & 1u)) continue;
            const long long tb0 = (long long)((t0 + k) * TILE_BLOCKS);
            const long long first = bk ? tb0 - (long long)SEED_BACK * (round + 1)
                                       : tb0 + TILE_BLOCKS + (long long)SEED_AHEAD * round;
            const long long q = (first - 1) * 64 + (long long)ci * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q >= 0) v = load16_guarded(bytes, N, (u64)q);
            u8* img = bk ? sw.tile[k].back : sw.tile[k].ahead;
            lds_store16(img + (ci >> 2) * PAD_STRIDE + (ci & 3) * 16, v);
        }
        // ... and record starts (cleared, then set by the lanes of each window)
        for (int i = lane; i < SEED_TPW * (SEED_BACK + 1 + SEED_AHEAD + 1); i += 64) {
            const int k = i / (SEED_BACK + 1 + SEED_AHEAD + 1), ii = i % (SEED_BACK + 1 + SEED_AHEAD + 1);
            if (ii <= SEED_BACK) sw.tile[k].rs_back[ii] = 0; else sw.tile[k].rs_ahead[ii - SEED_BACK - 1] = 0;
        }
        {
            const int k = rg_k >> 1;
            const bool bk = (rg_k & 1) == 0;
            if ((open_mask >> rg_k) & 1u) {
                const long long tb0 = (long long)((t0 + k) * TILE_BLOCKS);
                const long long first = bk ? tb0 - (long long)SEED_BACK * (round + 1)
                                           : tb0 + TILE_BLOCKS + (long long)SEED_AHEAD * round;
                const int n = bk ? SEED_BACK : SEED_AHEAD;
                const long long lo_blk = first < 0 ? 0 : first, hi_blk = first + n;   // starts in [lo_blk*64, (hi_blk+1)*64)
                if (hi_blk >= 0) {
                    const u64 T0 = (u64)lo_blk / TILE_BLOCKS, T1 = (u64)hi_blk / TILE_BLOCKS;
                    u64 r0 = rec_lo[T0 < nT ? T0 : nT], r1 = rec_lo[T1 + 1 < nT ? T1 + 1 : nT];
                    if (r1 > R) r1 = R;   // (entries left behind by invalid offsets)
                    if (r0 > r1) r0 = r1;
                    const u64 w0 = (u64)lo_blk * 64, w1 = ((u64)hi_blk + 1) * 64;
                    u64* rs = bk ? sw.tile[k].rs_back : sw.tile[k].rs_ahead;
                    for (u64 r = r0 + rg_sub; r < r1; r += SEED_RG_LANES) {
                        const u64 q = offsets[r];
                        if (q >= w0 && q < w1 && q < N) {
                            const u64 d = q - (u64)(first * 64);
                            atomicOr(reinterpret_cast<unsigned long long*>(&rs[d >> 6]), 1ull << (d & 63));
                        }
                    }
                }
            }
        }
        wave_lds_sync();   // the wave's own LDS writes are complete before any lane reads them
        // ---- classify this lane's window block
        const long long first = is_back ? my_tb0 - (long long)SEED_BACK * (round + 1)
                                        : my_tb0 + TILE_BLOCKS + (long long)SEED_AHEAD * round;
        const long long wb = first + li;
        const bool live = wb >= 0 && (u64)wb < nB && ((open_mask >> (2 * my_k + (is_back ? 0 : 1))) & 1u);
        J j = j_identity();
        const SeedTile& st = sw.tile[my_k];
        const PadBlk blk = {(is_back ? st.back : st.ahead) + (li + 1) * PAD_STRIDE};
        const u64 q = live ? (u64)wb * 64 : N;
        const int nvalid = (int)((N - q) < 64 ? (N - q) : 64);
        Cls c;
        if (!NORMALIZE) {
            // every lane classifies (a lane without a live block sees no valid byte): the lanes of a window
            // hold consecutive blocks and pass what runs over a block edge to each other
            WindowNeighbours nb = {l16 != SEED_BACK - 1 && l16 != SEED_LPT - 1, l16 != 0 && l16 != SEED_BACK};
            c = classify_block_fast(blk, nvalid, ut, nb);
        }
        if (live) {
            const u64* rsw = is_back ? st.rs_back : st.rs_ahead;
            const u64 V = below(nvalid);
            const u64 RS = rsw[li] & V;
            const bool nrs = (rsw[li + 1] & 1ull) != 0;
            Derived d;
            if (NORMALIZE) {
                Cls z = {classify_ws_fast(blk, nvalid), 0, 0, 0, 0, 0, 0};
                c = normalize_cls(z, V);
                d = derive(c, V, RS, false, false);
            } else {
                const bool has_next = q + 64 < N;
                const bool next_nl = has_next && blk[64] == '\n' && !nrs;
                const bool pt = prev_is_term(q > 0, blk[-1], nvalid > 0, blk[0], (RS & 1ull) != 0);
                d = derive(c, V, RS, next_nl, pt);
            }
            j = lane_J(c, d);
        }
        const WaveLines W = wave_lines(j);
        u32 still_open = 0;
#pragma unroll
        for (int k = 0; k < SEED_TPW; k++) {
            const long long tb0 = (long long)((t0 + k) * TILE_BLOCKS);
            if ((open_mask >> (2 * k)) & 1u) {
                J p = wave_range_total(W, k * SEED_LPT, k * SEED_LPT + SEED_BACK);
                if (round > 0) p = j_combine(p, uniform_J(sw.P[k]));
                if (lane == 0) sw.P[k] = p;
                const bool settled = NORMALIZE ? (p.g & (G_RS | G_AH)) != 0
                                               : (p.g & G_RS) != 0 || ((p.g & G_LS) && (p.g & G_C0));
                if (!(settled || tb0 - (long long)SEED_BACK * (round + 1) <= 0))   // or the corpus starts here
                    still_open |= 1u << (2 * k);
            }
            if ((open_mask >> (2 * k + 1)) & 1u) {
                S x = s_from_J(wave_range_total(W, k * SEED_LPT + SEED_BACK, (k + 1) * SEED_LPT));
                if (round > 0) x = s_combine(uniform_S(sw.X[k]), x);
                if (lane == 0) sw.X[k] = x;
                if (!(x.has_ls != 0 || (u64)(tb0 + TILE_BLOCKS + (long long)SEED_AHEAD * (round + 1)) >= nB))
                    still_open |= 2u << (2 * k);
            }
        }
        open_mask = still_open;
        wave_lds_sync();   // every lane has read the images before the next round overwrites them; the folds are stored
    }
    wave_lds_sync();
    if (lane < SEED_TPW && t0 + lane < nT) {
        seedJ[t0 + lane] = sw.P[lane];
        seedS[t0 + lane] = sw.X[lane];
    }
    if ((open_mask != 0 || force_open != 0) && lane == 0) *open_count = 1u;   // same value from every writer
}
template <bool NORMALIZE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4))) void k_seed(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                              const u32* __restrict__ rec_lo, u64 R, u64 N, u64 nB, u64 nT,
                                              u32 force_open, J* __restrict__ seedJ, S* __restrict__ seedS,
                                              u32* __restrict__ open_count) {
    __shared__ SeedShared ssh;
    __shared__ UniShared uni;
    stage_tables(uni);
    __syncthreads();
    seed_body<NORMALIZE>(ssh, uni, blockIdx.x, bytes, offsets, rec_lo, R, N, nB, nT, force_open, seedJ, seedS, open_count);
}

// k_tile_carry: what enters tile t from the left (fold of the summaries back to the start of the
// record that covers the tile's first byte) and from the right (rest of the line that leaves the tile).
// Usually the neighbouring tiles settle both (a record start within the previous few tiles, a line start
// in the next one): one thread per tile looks at most TC_NEAR tiles each way and lists the tiles that
// need a longer walk; those are then folded by one wavefront each, 64 summaries at a time.
constexpr int TC_NEAR = 4;
__global__ void k_tile_carry_near(const J* __restrict__ SA, u64 nT, const u32* __restrict__ open_count,
                                  J* __restrict__ seedJ, S* __restrict__ seedS,
                                  u32* __restrict__ far_list, u32* __restrict__ far_count) {
    if (*open_count == 0) return;   // k_seed has settled every tile
    const u64 nthreads = (u64)gridDim.x * blockDim.x;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < nT; t += nthreads) {
        J P = j_identity();
        bool ok = false;
        for (int step = 1; step <= TC_NEAR; step++) {
            if (t < (u64)step) { ok = true; break; }
            const J v = SA[t - step];
            P = j_combine(v, P);
            if (v.g & G_RS) { ok = true; break; }
        }
        S X = s_identity();
        bool ok2 = false;
        for (int step = 1; step <= TC_NEAR; step++) {
            if (t + step >= nT) { ok2 = true; break; }
            X = s_combine(X, s_from_J(SA[t + step]));
            if (X.has_ls) { ok2 = true; break; }
        }
        if (ok && ok2) { seedJ[t] = P; seedS[t] = X; }
        else far_list[atomicAdd(far_count, 1u)] = (u32)t;
    }
}

__global__ __launch_bounds__(NT) void k_tile_carry(const J* __restrict__ SA, u64 nT, J* __restrict__ seedJ,
                                                    S* __restrict__ seedS, const u32* __restrict__ far_list,
                                                    const u32* __restrict__ far_count) {
    const u32 n = *far_count;
    const u32 nwaves = gridDim.x * (NT / 64);
    for (u32 i = blockIdx.x * (NT / 64) + (threadIdx.x >> 6); i < n; i += nwaves) {   // whole waves
        const u64 t = far_list[i];
        J P = tile_lookback(SA, t);
        S X = tile_lookahead(SA, t, nT);
        if ((threadIdx.x & 63) == 0) { seedJ[t] = P; seedS[t] = X; }
    }
}

// ---- digit-heavy lines solved inside the tile (one wavefront per line) ---------------------------
constexpr u32 G3_LCAP = 1024;   // lines per tile that can be listed in LDS
constexpr u32 G3_MAXW = G3_WORDS;   // one lane's mask solver takes lines of up to 64 * G3_MAXW bytes; longer ones are solved by lanes together


// Per-block masks of the tile, kept in the LDS region of the (no longer needed) byte image.
struct G3Masks {
    u64 *ws, *pipe, *dig, *alp, *ls;   // inputs
    u64 *G, *gPSP, *gPNL;              // outputs: kept token bytes, ' ' / '\n' separators
};

// One digit-heavy line (gate 3, reference strip_digit_tokens, denoiser_preprocessor.rs:70-140) solved by a
// single lane (64 lines per wave at a time) with the mask solver of W words: no walk over the tokens, no
// limit on their number.  The mask arrays are indexed by block number within a view of n_words blocks; s is
// the position of the line's first byte in the view and corpus_end the position where the corpus ends (or ~0
// when that is outside the view).  Returns false when the line does not fit (its end is not in the view, or
// it is longer than 64 * W bytes).
template <int W, class RT>
__device__ inline bool g3_lane_solve(const G3Masks& a, u32 s, bool table, bool rab, u32 n_words, u32 corpus_end,
                                     const RT& rt) {
    const u32 w0 = s >> 6;
    u32 e = 0xFFFFFFFFu;
    for (u32 w = w0; w <= w0 + W && w < n_words; w++) {   // the next line start, if it is close enough
        u64 m = a.ls[w];
        if (w == w0) m &= ~below((int)(s & 63u) + 1);
        if (m) { e = w * 64u + (u32)ctz(m); break; }
    }
    if (e == 0xFFFFFFFFu) {
        if (corpus_end != 0xFFFFFFFFu && corpus_end <= (w0 + W + 1) * 64u) e = corpus_end;
        else return false;
    }
    if (e - s > 64u * W) return false;
    u64 K[W + 1], KS[W + 1];
    g3_solve_shifted<W>(
        [&](u32 w) -> G3Src {
            G3Src r = {0, 0, 0};
            if (w < n_words) {
                r.x = ~(a.ws[w] | (table ? a.pipe[w] : 0ull));
                r.dg = a.dig[w];
                r.al = a.alp[w];
            }
            return r;
        },
        s, e, rt, K, KS);
    bool first = true;
#pragma unroll
    for (u32 j = 0; j <= (u32)W; j++) {
        const u64 g = K[j];
        if (g) {
            u64 psp = KS[j], pnl = 0;
            if (first && psp) {
                const u64 low = psp & (0ull - psp);
                psp &= ~low;
                if (rab) pnl = low;
                first = false;
            }
            atomicOr(reinterpret_cast<unsigned long long*>(&a.G[w0 + j]), g);
            if (psp) atomicOr(reinterpret_cast<unsigned long long*>(&a.gPSP[w0 + j]), psp);
            if (pnl) atomicOr(reinterpret_cast<unsigned long long*>(&a.gPNL[w0 + j]), pnl);
        }
    }
    return true;
}

struct U32Add { __device__ u32 operator()(u32 a, u32 b) const { return a + b; } };

// The emit plan in global memory.  The copy and token planes exist for every block.  The separator planes
// (PSP / PNL) are zero for most blocks of most corpora: they are stored only for the blocks that have a
// separator bit, hold the marker of a digit-heavy line that is still to be solved, or lie inside such a line
// (k_gate3_solve ORs the separators of the kept tokens into them); pfx_mask says which blocks those are.
struct PlanCT { u64 C, TOK; };
struct PlanPX { u64 PSP, PNL; };
struct PlanMem {
    PlanCT* ct;
    PlanPX* px;
    u64* pfx_mask;   // word s: blocks 64 s .. 64 s + 63
};
__host__ __device__ constexpr size_t plan_bytes(u64 nB) { return (nB + 1) * (sizeof(PlanCT) + sizeof(PlanPX)) + (nB / 64 + 2) * sizeof(u64); }
__host__ __device__ inline PlanMem plan_view(void* base, u64 nB) {
    PlanMem m;
    m.ct = reinterpret_cast<PlanCT*>(base);
    m.px = reinterpret_cast<PlanPX*>(m.ct + (nB + 1));
    m.pfx_mask = reinterpret_cast<u64*>(m.px + (nB + 1));
    return m;
}

struct PlanWork {
    u64 tokc[TILE_BLOCKS];
    u64 marks[TILE_BLOCKS];
    uint16_t list[G3_LCAP];
};
struct PlanShared {
    TileShared tile;   // byte image; after classification reused for the eight G3Masks arrays
    J jscan[NT / 64];
    union {            // the classifier's per-wave scratch is dead before the first of the others is written
        PlanWork w;
        MbScratch mb[NT / 64];
    };
    u32 scan32[NT / 64];
    u32 nlist;
    u64 g3base;
    u32 red[3][NT / 64];
    RsShared rs;
    uint16_t thr1[RATIO_TAB];   // the digit-ratio test as a table (RatioTab)
};
static_assert(RATIO_TAB == NT, "one table entry per thread of k_plan");
// The table of the digit-ratio test as a kernel argument (filled on the host, ratio_table_fill).
struct RatioArg {
    uint16_t thr1[RATIO_TAB];
};
static_assert(sizeof(TileShared) >= 8 * TILE_BLOCKS * sizeof(u64), "mask arrays must fit in the byte image");
static_assert(sizeof(MbScratch) * (NT / 64) <= sizeof(PlanWork), "the classifier's scratch must fit under the planning arrays");

// uni: written here (stage_tables) before the first barrier.
template <bool NORMALIZE>
__device__ __forceinline__ void plan_body(PlanShared& sh, UniShared& uni, u32 vb, const u8* __restrict__ bytes,
                                          const u64* __restrict__ offsets, const u32* __restrict__ rec_lo, u64 R, u64 N,
                                          u64 nB, const J* __restrict__ seedJ, const S* __restrict__ seedS,
                                          const DevCfg& cfg, const RatioArg& ratio, const PlanMem& plan,
                                          u64* __restrict__ g3list, u64 g3cap, u64* __restrict__ g3count,
                                          u32* __restrict__ tile_bytes, u32* __restrict__ tile_toks,
                                          u32* __restrict__ tile_flags, u32 stop_after) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 t = vb, tS = t * (u64)TILE;
    const u64 b = t * (u64)TILE_BLOCKS + tid;
    {
        TileRegs regs;
        RsFetch rf;
        tile_load(bytes, N, tS, regs);
        rs_fetch(offsets, rec_lo, R, t, rf);
        stage_tables(uni);
        if (!NORMALIZE) sh.thr1[tid] = ratio.thr1[tid];
        tile_store(regs, sh.tile.bytes);
        rs_store(rf, offsets, rec_lo, R, N, t, sh.rs);
    }
    const RatioTab rt = {cfg.max_digit_ratio, sh.thr1};
    // what enters the tile from its neighbours: the same address for every lane, loaded into scalar registers
    const J seed_j = seedJ[t];
    const S seed_s = seedS[t];
    __syncthreads();
    if (stop_after == 1) return;   // profiling aid (see trp_ctx::plan_stop): end after the named phase
    LaneCtx L = lane_setup(sh.tile, lds_tables(uni), N, tS, true, sh.rs.w, NORMALIZE, &sh.mb[wave]);
    // (the classifier's scratch shares its space with the arrays written from here on: every wave is through with
    // it when it gets here, and the first of those arrays that another wave reads is read after the next barrier)
    if (stop_after == 2) { if (L.c.ws == 0x123456789ull) g3list[0] = L.d.LS ^ L.c.alp ^ L.c.dig; return; }
    // The ' ' and '\n' bytes are needed once more, at the very end (plan_fold_prefixes): parked in LDS in one
    // mask until then.  A '\n' is a terminator and a ' ' is line content, which tells them apart again.
    // Parked in the middle of the lane's own block of the byte image: nobody else reads those bytes (neighbours
    // look at most three bytes into a block, from either side), and the planning arrays still hold other
    // wavefronts' classifier scratch until the next barrier.
    u32* const park = reinterpret_cast<u32*>(sh.tile.bytes + (tid + 1) * PAD_STRIDE + 16);
    if (!NORMALIZE) { const u64 v = L.c.sp | L.c.nl; park[0] = (u32)v; park[1] = (u32)(v >> 32); }
    const J own = lane_J(L.c, L.d);
    // the block's own head / tail statistics are used again by plan_block: parked in one register each
    const u32 own_head = pack_ls(own.head), own_tail = pack_ls(own.tail);
#ifdef TRP_FINE_STOPS
    if (stop_after == 21) { if (own_head == 0x12345678u) g3list[0] = own_tail + own.g; return; }
#endif
    WaveLines W;
    NearLines Q = {};
    wave_ballots(own, W);
    const bool near = lines_are_near(W.lsm);   // uniform: the wave's lines are short
    {
        J wt;
        if (near) {
            Q = near_lines((own.g & G_LS) != 0, own_head, own_tail);
            wt = near_total(W, Q, own_head, own_tail);
        } else {
            W = wave_lines(own);
            wt = wave_total(W);
        }
        if (lane == 0) sh.jscan[wave] = wt;
    }
#ifdef TRP_FINE_STOPS
    if (stop_after == 22) { if (Q.ci == 0x12345678u) g3list[0] = Q.co + W.lsm; return; }
#endif
    __syncthreads();   // also: every lane has classified its block, the byte image is free from here on
    // what enters this wave from the left and from the right: folds over the other waves' totals.  The
    // summaries are the same for every lane: moved to scalar registers, so that the folds run on the scalar unit
    J Pw = uniform_J(seed_j);
    S Xw = uniform_S(seed_s);
    {
        const int wu = __builtin_amdgcn_readfirstlane(wave);   // scalar: the branches below are uniform
#pragma unroll
        for (int w = 0; w < NT / 64 - 1; w++) {
            if (w < wu) Pw = j_combine(Pw, uniform_J(sh.jscan[w]));
            const int v = NT / 64 - 1 - w;
            if (v > wu) Xw = s_combine(uniform_S(s_from_J(sh.jscan[v])), Xw);
        }
    }
    LaneCarry k = near ? near_carry(W, Q, own.g, Pw, Xw, L.next_rs) : wave_carry(W, Pw, Xw, L.next_rs);
    if (stop_after == 3) { if (k.ci.d == 0x12345678u) g3list[0] = k.co.d + k.ci.a + (k.ci_rab ? 1 : 0); return; }
    bool head_g3 = false;   // the line that runs into this block is a digit-heavy line (marked at its first byte)
    Plan pl = NORMALIZE ? plan_normalize(L.c, L.d, k.ar_in, L.prev_tok, L.sp)
                        : plan_block(L.c, L.d, k, cfg, rt, unpack_ls8(own_head), unpack_ls8(own_tail), &head_g3, false);
    // (the plan is brought into its canonical form, plan_fold_prefixes, once the digit-heavy lines of the tile are
    // in it: the markers and what is known so far are not touched by that step)
    u64 sp_nl = 0;
    if (!NORMALIZE) sp_nl = (u64)park[0] | ((u64)park[1] << 32);
    // Digit-heavy lines (gate 3) are marked by plan_block at their first byte.  Lines that fit are solved
    // right here, one wavefront per line, on the per-block masks (the byte image is overwritten).
    if (stop_after == 4) { if (pl.C == 0x123456789ull) g3list[0] = pl.TOK ^ pl.PSP ^ pl.PNL; return; }
    u64 marks = pl.TOK & ~pl.C;
    pl.TOK &= pl.C;
    // most tiles of ordinary text have no digit-heavy line at all: one ballot per wave tells
    {
        const u64 bm = __ballot(marks != 0);
        if (lane == 0) sh.red[0][wave] = bm != 0 ? 1u : 0u;
    }
    __syncthreads();   // also: nobody reads the byte image any more
    u32 any_marks = 0;
    for (int w = 0; w < NT / 64; w++) any_marks |= sh.red[0][w];
    if (any_marks) {   // uniform
        u64* base = reinterpret_cast<u64*>(sh.tile.bytes);
        G3Masks gm = {base, base + TILE_BLOCKS, base + 2 * TILE_BLOCKS, base + 3 * TILE_BLOCKS,
                      base + 4 * TILE_BLOCKS, base + 5 * TILE_BLOCKS, base + 6 * TILE_BLOCKS,
                      base + 7 * TILE_BLOCKS};
        gm.ws[tid] = L.c.ws; gm.pipe[tid] = L.c.pipe; gm.dig[tid] = L.c.dig; gm.alp[tid] = L.c.alp;
        gm.ls[tid] = L.d.LS;
        gm.G[tid] = 0; gm.gPSP[tid] = 0; gm.gPNL[tid] = 0;
        sh.w.marks[tid] = marks;
        sh.w.tokc[tid] = sp_nl;   // (the byte image, where it was parked, now holds the mask arrays)
        // Two work lists: lines that end within 64 bytes of their first byte (more than half of the digit-heavy
        // lines of OCR text) are solved with masks of one word, the others with two words, at twice the price per
        // operation.  A marker below the block's last line start belongs to a line that ends inside the block; the
        // line of the last line start is short when the next block has a line start no further in than the marker
        // is in this block (the last lane of a wavefront cannot tell and says "long", which is always right).
        const u32 nls_lo = from_next_lane((u32)L.d.LS), nls_hi = from_next_lane((u32)(L.d.LS >> 32));
        u64 shortm = 0;
        if (marks) {
            const int last = top(L.d.LS);   // markers are line starts
            const u64 next_ls = (u64)nls_lo | ((u64)nls_hi << 32);
            shortm = marks & below(last);
            if (((marks >> last) & 1ull) && (next_ls & below(last + 1))) shortm |= bit(last);
        }
        const u64 longm = marks & ~shortm;
        // positions in the two lists: one packed prefix sum (short | long << 16) over the tile
        const u32 cnt2 = (u32)popc(shortm) | ((u32)popc(longm) << 16);
        const u32 inc2 = wave_incl_scan_add(cnt2);
        if (lane == 63) sh.scan32[wave] = inc2;
        __syncthreads();   // also: the mask arrays are complete
        u32 pre2 = 0, tot2 = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) {
            const u32 tw = sh.scan32[w];
            if (w < wave) pre2 += tw;
            tot2 += tw;
        }
        if (tot2 > 0) {   // uniform
            uint16_t* list_s = sh.w.list;
            uint16_t* list_l = sh.w.list + G3_LCAP / 2;
            u32 at_s = (pre2 + inc2 - cnt2) & 0xFFFFu, at_l = (pre2 + inc2 - cnt2) >> 16;
            // a tile whose lines all fit one pass of the two-word solver takes that one pass
            const bool one_list = (tot2 & 0xFFFFu) + (tot2 >> 16) <= 64u;   // uniform
            if (one_list) { at_l += at_s; shortm = 0; }
            for (u64 m = marks; m; m &= m - 1) {
                const int jb = ctz(m);
                const uint16_t ent = (uint16_t)((u32)(tid * BLOCK + jb) | (((pl.PSP >> jb) & 1ull) ? 0x4000u : 0u) |
                                                (((pl.PNL >> jb) & 1ull) ? 0x8000u : 0u));
                if ((shortm >> jb) & 1ull) { if (at_s < G3_LCAP / 2) list_s[at_s] = ent; at_s++; }
                else { if (at_l < G3_LCAP / 2) list_l[at_l] = ent; at_l++; }
            }
            __syncthreads();
#ifdef TRP_FINE_STOPS
            if (stop_after == 51) return;
#endif
            const u32 n_s = one_list ? 0u : min(tot2 & 0xFFFFu, G3_LCAP / 2);
            const u32 n_l = one_list ? (tot2 & 0xFFFFu) + (tot2 >> 16) : min(tot2 >> 16, G3_LCAP / 2);
            const u32 tile_valid = (u32)(N - tS < (u64)TILE ? N - tS : (u64)TILE);
            const u32 cend = tS + TILE >= N ? tile_valid : 0xFFFFFFFFu;
            // one line per lane; the short list is handed out from the first lane of the workgroup upwards, the
            // long one from the last lane downwards, so that the two usually occupy different wavefronts
            for (u32 i = tid; i < n_s; i += NT) {
                const u32 ent = list_s[i], s0 = ent & 0x3FFFu;
                if (g3_lane_solve<1>(gm, s0, (ent >> 14) & 1u, (ent >> 15) & 1u, TILE_BLOCKS, cend, rt))
                    atomicAnd(reinterpret_cast<unsigned long long*>(&sh.w.marks[s0 >> 6]), ~(1ull << (s0 & 63u)));
            }
            for (u32 i = NT - 1 - tid; i < n_l; i += NT) {
                const u32 ent = list_l[i], s0 = ent & 0x3FFFu;
                // lines of more than 128 bytes are left for k_gate3_solve
                if (g3_lane_solve<G3_SHORT>(gm, s0, (ent >> 14) & 1u, (ent >> 15) & 1u, TILE_BLOCKS, cend, rt))
                    atomicAnd(reinterpret_cast<unsigned long long*>(&sh.w.marks[s0 >> 6]), ~(1ull << (s0 & 63u)));
            }
            __syncthreads();
#ifdef TRP_FINE_STOPS
            if (stop_after == 52) return;
#endif
            const u64 left = sh.w.marks[tid];
            const u64 solved = marks & ~left;
            pl.PSP = (pl.PSP & ~solved) | gm.gPSP[tid];
            pl.PNL = (pl.PNL & ~solved) | gm.gPNL[tid];
            pl.C |= gm.G[tid];
            pl.TOK |= gm.G[tid];
            marks = left;
        }
        sp_nl = sh.w.tokc[tid];
    }
    if (!NORMALIZE) plan_fold_prefixes(pl, sp_nl & L.d.LC, sp_nl & ~L.d.LC, L.d.V);
    if (stop_after == 5) { if (pl.C == 0x123456789ull) g3list[0] = pl.TOK ^ pl.PSP ^ pl.PNL ^ marks; return; }
    {
        // separator planes: for blocks with a separator, with the marker of a line left for k_gate3_solve, or
        // inside a digit-heavy line that began in an earlier block (k_gate3_solve may add separators there)
        const bool need_px = L.nvalid > 0 && ((pl.PSP | pl.PNL) != 0 || marks != 0 || head_g3);
        const u64 pm = __ballot(need_px);
        if (L.nvalid > 0) {
            const PlanCT ct = {pl.C, pl.TOK};
            plan.ct[b] = ct;
            if (need_px) { const PlanPX px = {pl.PSP, pl.PNL}; plan.px[b] = px; }
        }
        if (lane == 0 && (b >> 6) * (u64)SUB < N) plan.pfx_mask[b >> 6] = pm;
    }
    if (stop_after == 6) return;
    u64 TOKC = pl.TOK;
    u64 PFX = (pl.PSP | pl.PNL) & pl.C;
    // The counts are kept per sub-tile (= per half wavefront: 32 blocks).  A token that runs into a sub-tile from
    // the left is counted as a start by the sub-tile (GenTiles takes it back, from the flags).
    const u32 left_tok = from_prev_lane((u32)(TOKC >> 63));   // read by every lane: a lane outside a branch is no source
    const bool prev_tok = (lane & (SUB_BLOCKS - 1)) > 0 && left_tok != 0;
    const u64 TS = token_starts(TOKC, L.d.RS, prev_tok);
    // bytes (<= 128 per block) and token starts (<= 32 per block) summed over each half in one register:
    // row prefix sums, then the totals of rows 0 and 2 added into rows 1 and 3 - lanes 31 and 63 hold the sums
    u32 cnt = (popc(pl.C) + popc(PFX)) | (popc(TS) << 16);
    cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x111, 0xf, 0xf, false);   // row_shr:1
    cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x112, 0xf, 0xf, false);   // row_shr:2
    cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x114, 0xf, 0xf, false);   // row_shr:4
    cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x118, 0xf, 0xf, false);   // row_shr:8
    cnt += (u32)__builtin_amdgcn_update_dpp(0, (int)cnt, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    const u64 first_cont = __ballot((TOKC & 1ull) && !(L.d.LS & 1ull));   // bits 0 and 32: the halves' first blocks
    const u64 left_m = __ballot(marks != 0);
    if (lane == 0) sh.red[2][wave] = left_m != 0 ? 1u : 0u;
    if ((lane & (SUB_BLOCKS - 1)) == SUB_BLOCKS - 1) {
        const u64 sub = b / SUB_BLOCKS;
        if (sub * (u64)SUB < N) {
            tile_bytes[sub] = cnt & 0xFFFFu;
            tile_toks[sub] = cnt >> 16;
            u32 f = 0;
            if ((first_cont >> (lane - (SUB_BLOCKS - 1))) & 1ull) f |= TF_FIRST_CONT;
            if (TOKC >> 63) f |= TF_LAST_TOK;
            tile_flags[sub] = f;
        }
    }
    __syncthreads();
    u32 sm = 0;
    for (int w = 0; w < NT / 64; w++) sm |= sh.red[2][w];
    if (sm > 0) {   // uniform, rare: lines left for k_gate3_solve go to the global work list
        u32 tot = 0;
        const u32 at = block_scan_excl_fwd((u32)popc(marks), U32Add(), 0u, 0u, sh.scan32, &tot);   // barriers inside
        if (tid == 0) sh.g3base = atomicAdd(reinterpret_cast<unsigned long long*>(g3count), (unsigned long long)tot);
        __syncthreads();
        u64 pos = sh.g3base + at;
        for (u64 m = marks; m; m &= m - 1, pos++) {
            const int jb = ctz(m);
            if (pos < g3cap)
                g3list[pos] = (b * 64 + (u64)jb) | (((pl.PSP >> jb) & 1ull) << 32) | (((pl.PNL >> jb) & 1ull) << 33);
        }
    }
}
template <bool NORMALIZE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(6))) void k_plan(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                              const u32* __restrict__ rec_lo, u64 R, u64 N, u64 nB,
                                              const J* __restrict__ seedJ, const S* __restrict__ seedS, DevCfg cfg,
                                              RatioArg ratio, PlanMem plan, u64* __restrict__ g3list, u64 g3cap,
                                              u64* __restrict__ g3count,
                                              u32* __restrict__ tile_bytes, u32* __restrict__ tile_toks,
                                              u32* __restrict__ tile_flags, u32 stop_after) {
    __shared__ PlanShared sh;
    __shared__ UniShared uni;
    plan_body<NORMALIZE>(sh, uni, blockIdx.x, bytes, offsets, rec_lo, R, N, nB, seedJ, seedS, cfg, ratio, plan, g3list, g3cap,
                         g3count, tile_bytes, tile_toks, tile_flags, stop_after);
}

struct CountShared {
    TileShared tile;
    RsShared rs;
};
__device__ __forceinline__ void count_tokens_body(CountShared& csh, u32 vb, const u8* __restrict__ bytes,
                                                  const u64* __restrict__ offsets, const u32* __restrict__ rec_lo, u64 R,
                                                  u64 N, u64 nB, const PlanMem& plan, u32* __restrict__ tile_bytes,
                                                  u32* __restrict__ tile_toks, u32* __restrict__ tile_flags) {
    TileShared& sh = csh.tile;
    RsShared& rs = csh.rs;
    const int tid = threadIdx.x, lane = tid & 63;
    const u64 t = vb, tS = t * (u64)TILE;
    {
        TileRegs regs;
        RsFetch rf;
        tile_load(bytes, N, tS, regs);
        rs_fetch(offsets, rec_lo, R, t, rf);
        tile_store(regs, sh.bytes);
        rs_store(rf, offsets, rec_lo, R, N, t, rs);
    }
    __syncthreads();
    const UniTables no_tables = {nullptr, nullptr, 0, nullptr};   // the whitespace classifier uses none
    LaneCtx L = lane_setup(sh, no_tables, N, tS, false, rs.w);
    Plan pl = plan_passthrough(L.c, L.d.V);
    // the copy and token planes go to the plan, so that k_emit need not classify the bytes again
    if (L.nvalid > 0) { const PlanCT ct = {pl.C, pl.TOK}; plan.ct[t * (u64)TILE_BLOCKS + tid] = ct; }
    // counts per sub-tile (half a wavefront), as in k_plan: one row-wise DPP sum, lanes 31 and 63 hold the totals
    const u32 left_tok = from_prev_lane((u32)(pl.TOK >> 63));
    const bool prev_tok = (lane & (SUB_BLOCKS - 1)) > 0 && left_tok != 0;
    u32 ntk = popc(token_starts(pl.TOK, L.d.RS, prev_tok));
    ntk += (u32)__builtin_amdgcn_update_dpp(0, (int)ntk, 0x111, 0xf, 0xf, false);   // row_shr:1
    ntk += (u32)__builtin_amdgcn_update_dpp(0, (int)ntk, 0x112, 0xf, 0xf, false);   // row_shr:2
    ntk += (u32)__builtin_amdgcn_update_dpp(0, (int)ntk, 0x114, 0xf, 0xf, false);   // row_shr:4
    ntk += (u32)__builtin_amdgcn_update_dpp(0, (int)ntk, 0x118, 0xf, 0xf, false);   // row_shr:8
    ntk += (u32)__builtin_amdgcn_update_dpp(0, (int)ntk, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    const u64 first_cont = __ballot((pl.TOK & 1ull) && !(L.d.RS & 1ull));   // bits 0 and 32: the halves' first blocks
    if ((lane & (SUB_BLOCKS - 1)) == SUB_BLOCKS - 1) {
        const u64 sub = (t * (u64)TILE_BLOCKS + tid) / SUB_BLOCKS, sS = sub * (u64)SUB;
        if (sS < N) {
            tile_bytes[sub] = (u32)(N - sS < (u64)SUB ? N - sS : (u64)SUB);
            tile_toks[sub] = ntk;
            u32 f = 0;
            if ((first_cont >> (lane - (SUB_BLOCKS - 1))) & 1ull) f |= TF_FIRST_CONT;
            if (pl.TOK >> 63) f |= TF_LAST_TOK;
            tile_flags[sub] = f;
        }
    }
}
__global__ __launch_bounds__(NT) void k_count_tokens(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                                      const u32* __restrict__ rec_lo, u64 R, u64 N,
                                                      u64 nB, PlanMem plan, u32* __restrict__ tile_bytes,
                                                      u32* __restrict__ tile_toks, u32* __restrict__ tile_flags) {
    __shared__ CountShared csh;
    count_tokens_body(csh, blockIdx.x, bytes, offsets, rec_lo, R, N, nB, plan, tile_bytes, tile_toks, tile_flags);
}

// ---------------------------------------------------------------------------------------------------
// k_gate3: solve the digit-heavy lines that k_plan left on the work list.
// ---------------------------------------------------------------------------------------------------
__device__ inline void plan_or(const PlanMem& plan, u64 word, u64 c, u64 tok, u64 psp, u64 pnl) {
    if (c) atomicOr(reinterpret_cast<unsigned long long*>(&plan.ct[word].C), c);
    if (tok) atomicOr(reinterpret_cast<unsigned long long*>(&plan.ct[word].TOK), tok);
    if (psp) atomicOr(reinterpret_cast<unsigned long long*>(&plan.px[word].PSP), psp);
    if (pnl) atomicOr(reinterpret_cast<unsigned long long*>(&plan.px[word].PNL), pnl);
}

// Serial solver for one listed line (any length): lane-local walk over the bytes, results OR-ed into
// the plan with atomics.  Slow (microseconds per byte); only used when there is no output buffer to hold the
// vectors of g3_huge_line.
__device__ inline void g3_serial_solve(const u8* __restrict__ bytes, const u64* __restrict__ offsets, u64 R, u64 N,
                                       const DevCfg& cfg, const PlanMem& plan, u64 p, bool table, bool rab,
                                       u32* tile_bytes, u32* tile_toks) {
    const UniTables ut = dev_tables();
    ByteSrc src = make_src(bytes, N);
    u64 lo = 0, hi = R;   // end of the record that holds p: first offset > p
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (offsets[mid + 1] <= p) lo = mid + 1; else hi = mid;
    }
    u64 rec_end = lo < R ? offsets[lo + 1] : N;   // (lo == R only with offsets that break the contract)
    if (rec_end > N) rec_end = N;
    u64 e = p;
    while (e < rec_end) {
        u32 ch = src.get(e);
        if (ch == '\n' || ch == '\r') break;
        e++;
    }
    u64 K = g3_solve_K(src, p, e, table, cfg.max_digit_ratio, ut);
    bool first = true;
    g3_walk(src, p, e, table, ut,
            [&](const TokInfo& tk, u64 dist) {
                if (dist > K) return;
                u64 w0 = tk.ts >> 6, w1 = (tk.te - 1) >> 6;
                for (u64 w = w0; w <= w1; w++) {
                    u64 lo_b = w == w0 ? (tk.ts & 63) : 0;
                    u64 hi_b = w == w1 ? ((tk.te - 1) & 63) + 1 : 64;
                    u64 m = below((int)hi_b) & ~below((int)lo_b);
                    u64 pre = (w == w0) ? (1ull << lo_b) : 0;
                    plan_or(plan, w, m, m, (pre && !first) ? pre : 0, (pre && first && rab) ? pre : 0);
                    atomicAdd(&tile_bytes[w / SUB_BLOCKS], (u32)popc(m));
                }
                if (!first || rab) atomicAdd(&tile_bytes[w0 / SUB_BLOCKS], 1u);
                atomicAdd(&tile_toks[w0 / SUB_BLOCKS], 1u);
                first = false;
            });
}

}  // namespace

// ---- gate 3 for long lines: one wavefront per line ---------------------------------------------------
// The bit vectors of the mask solver spread over the lanes of a wavefront: lane L holds the words
// L * W .. L * W + W - 1 (W * 4 KiB of text per vector).  The operations g3_bits_solve needs (trp_lane.h):
// carries, shifted-in bits and the mirror image cross the lanes through ballots, DPP and ds_bpermute.
namespace trp {
template <int W>
struct WaveBits {
    u64 w[W];
};
template <int W>
__device__ inline WaveBits<W> g3b_and(const WaveBits<W>& a, const WaveBits<W>& b) {
    WaveBits<W> r;
#pragma unroll
    for (int j = 0; j < W; j++) r.w[j] = a.w[j] & b.w[j];
    return r;
}
template <int W>
__device__ inline WaveBits<W> g3b_andn(const WaveBits<W>& a, const WaveBits<W>& b) {
    WaveBits<W> r;
#pragma unroll
    for (int j = 0; j < W; j++) r.w[j] = a.w[j] & ~b.w[j];
    return r;
}
template <int W>
__device__ inline WaveBits<W> g3b_or(const WaveBits<W>& a, const WaveBits<W>& b) {
    WaveBits<W> r;
#pragma unroll
    for (int j = 0; j < W; j++) r.w[j] = a.w[j] | b.w[j];
    return r;
}
template <int W>
__device__ inline WaveBits<W> g3b_orn(const WaveBits<W>& a, const WaveBits<W>& b) {
    WaveBits<W> r;
#pragma unroll
    for (int j = 0; j < W; j++) r.w[j] = a.w[j] | ~b.w[j];
    return r;
}
template <int W>
__device__ inline WaveBits<W> g3b_xnor(const WaveBits<W>& a, const WaveBits<W>& b) {
    WaveBits<W> r;
#pragma unroll
    for (int j = 0; j < W; j++) r.w[j] = ~(a.w[j] ^ b.w[j]);
    return r;
}
template <int W>
__device__ inline bool g3b_any(const WaveBits<W>& a) {
    u64 x = 0;
#pragma unroll
    for (int j = 0; j < W; j++) x |= a.w[j];
    return __ballot(x != 0) != 0;
}
template <int W>
__device__ inline u32 g3b_popc_and(const WaveBits<W>& a, const WaveBits<W>& b) {
    u32 n = 0;
#pragma unroll
    for (int j = 0; j < W; j++) n += (u32)popc(a.w[j] & b.w[j]);
    return readlane_u32(wave_incl_scan_add(n), 63);
}
// Sum of two vectors.  A lane's own words are added with their carries; g = the lane carries out, p = the
// lane would pass a carry on (all its sum bits are ones).  The carry into every lane follows from the two
// ballots like the carries of a binary addition: c[i+1] = g[i] | (p[i] & c[i]).
template <int W>
__device__ inline WaveBits<W> g3b_add(const WaveBits<W>& a, const WaveBits<W>& b) {
    WaveBits<W> r;
    unsigned long long c = 0;
    u64 all = ~0ull;
#pragma unroll
    for (int j = 0; j < W; j++) { r.w[j] = __builtin_addcll(a.w[j], b.w[j], c, &c); all &= r.w[j]; }
    const u64 G = __ballot(c != 0), P = __ballot(all == ~0ull);
    const u64 V = G << 1;
    const u64 C = V | ((P + V) ^ P ^ V);
    u64 cin = (C >> (threadIdx.x & 63)) & 1ull;
#pragma unroll
    for (int j = 0; j < W; j++) { r.w[j] += cin; cin = (cin && r.w[j] == 0) ? 1ull : 0ull; }
    return r;
}
template <int W>
__device__ inline WaveBits<W> g3b_run_starts(const WaveBits<W>& x) {
    WaveBits<W> r;
    u64 c = from_prev_lane((u32)(x.w[W - 1] >> 63));   // the last bit of the lane before (lane 0: none)
#pragma unroll
    for (int j = 0; j < W; j++) { r.w[j] = x.w[j] & ~((x.w[j] << 1) | c); c = x.w[j] >> 63; }
    return r;
}
template <int W>
__device__ inline WaveBits<W> g3b_run_ends(const WaveBits<W>& x) {
    WaveBits<W> r;
    const u64 nl = from_next_lane((u32)(x.w[0] & 1ull));   // the first bit of the lane after (lane 63: none)
#pragma unroll
    for (int j = 0; j < W; j++) {
        const u64 nx = j + 1 < W ? (x.w[j + 1 < W ? j + 1 : 0] & 1ull) : nl;
        r.w[j] = x.w[j] & ~((x.w[j] >> 1) | (nx << 63));
    }
    return r;
}
template <int W>
__device__ inline WaveBits<W> g3b_run_from(const WaveBits<W>& x, const WaveBits<W>& points) { return g3b_andn(x, g3b_add(x, points)); }
template <int W>
__device__ inline WaveBits<W> g3b_rev(const WaveBits<W>& a) {
    WaveBits<W> r;
    const int src = 63 - (int)(threadIdx.x & 63);
#pragma unroll
    for (int j = 0; j < W; j++) {
        const u64 v = a.w[W - 1 - j];
        const u32 lo = (u32)__shfl((int)(u32)v, src, 64), hi = (u32)__shfl((int)(u32)(v >> 32), src, 64);
        r.w[j] = bitrev64((u64)lo | ((u64)hi << 32));
    }
    return r;
}
template <int W>
__device__ inline void g3b_opaque(WaveBits<W>&) {}

// The same with one vector per DPP row: the 16 lanes of a row hold one 1 KiB vector, a wavefront four of them
// (four lines solved side by side).  Carries, shifted-in bits and the mirror image stay inside the row.
struct RowBits {
    u64 w;
};
__device__ inline u32 row_prev_lane(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true); }   // row_shr:1
__device__ inline u32 row_next_lane(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true); }   // row_shl:1
__device__ inline u32 row_bits(u64 ballot) { return (u32)(ballot >> (threadIdx.x & 48)) & 0xFFFFu; }   // the row's part of a ballot
__device__ inline RowBits g3b_and(const RowBits& a, const RowBits& b) { RowBits r = {a.w & b.w}; return r; }
__device__ inline RowBits g3b_andn(const RowBits& a, const RowBits& b) { RowBits r = {a.w & ~b.w}; return r; }
__device__ inline RowBits g3b_or(const RowBits& a, const RowBits& b) { RowBits r = {a.w | b.w}; return r; }
__device__ inline RowBits g3b_orn(const RowBits& a, const RowBits& b) { RowBits r = {a.w | ~b.w}; return r; }
__device__ inline bool g3b_any(const RowBits& a) { return row_bits(__ballot(a.w != 0)) != 0; }   // the same in the lanes of a row
__device__ inline u32 g3b_popc_and(const RowBits& a, const RowBits& b) {   // the row's total, in every lane of the row
    u32 n = (u32)popc(a.w & b.w);
    n += (u32)__builtin_amdgcn_update_dpp(0, (int)n, 0x111, 0xf, 0xf, false);   // row_shr:1
    n += (u32)__builtin_amdgcn_update_dpp(0, (int)n, 0x112, 0xf, 0xf, false);   // row_shr:2
    n += (u32)__builtin_amdgcn_update_dpp(0, (int)n, 0x114, 0xf, 0xf, false);   // row_shr:4
    n += (u32)__builtin_amdgcn_update_dpp(0, (int)n, 0x118, 0xf, 0xf, false);   // row_shr:8
    return (u32)__shfl((int)n, (int)(threadIdx.x & 63) | 15, 64);
}
__device__ inline RowBits g3b_add(const RowBits& a, const RowBits& b) {
    unsigned long long c = 0;
    RowBits r;
    r.w = __builtin_addcll(a.w, b.w, c, &c);
    const u32 G = row_bits(__ballot(c != 0)), P = row_bits(__ballot(r.w == ~0ull));
    const u32 V = (G << 1) & 0xFFFFu;
    const u32 C = V | ((P + V) ^ P ^ V);
    r.w += (C >> (threadIdx.x & 15)) & 1u;
    return r;
}
__device__ inline RowBits g3b_run_starts(const RowBits& x) {
    RowBits r = {x.w & ~((x.w << 1) | (u64)row_prev_lane((u32)(x.w >> 63)))};
    return r;
}
__device__ inline RowBits g3b_run_ends(const RowBits& x) {
    RowBits r = {x.w & ~((x.w >> 1) | ((u64)row_next_lane((u32)(x.w & 1ull)) << 63))};
    return r;
}
__device__ inline RowBits g3b_run_from(const RowBits& x, const RowBits& points) { return g3b_andn(x, g3b_add(x, points)); }
__device__ inline RowBits g3b_rev(const RowBits& a) {
    const int lane = (int)(threadIdx.x & 63), src = (lane & 48) | (15 - (lane & 15));
    const u32 lo = (u32)__shfl((int)(u32)a.w, src, 64), hi = (u32)__shfl((int)(u32)(a.w >> 32), src, 64);
    RowBits r = {bitrev64((u64)lo | ((u64)hi << 32))};
    return r;
}
// g3_bits_solve for four lines at once: the rows leave the expansion loop one by one (`active`), the wavefront
// when the last one has.
__device__ inline RowBits g3_rows_solve(const RowBits& X, const RowBits& DG, const RowBits& AL, float max_ratio, bool active) {
    RowBits K;
    {
        const RowBits Y = g3b_andn(X, AL);
        const RowBits P = g3b_and(g3b_andn(g3b_add(Y, g3b_run_starts(X)), Y), AL);
        K = g3b_or(g3b_run_from(X, P), g3b_rev(g3b_run_from(g3b_rev(X), g3b_rev(P))));
    }
    const u32 a0 = g3b_popc_and(AL, X);
    u32 d0 = g3b_popc_and(DG, K);
    for (;;) {
        const RowBits Z = g3b_orn(K, X);
        const RowBits NS = g3b_andn(g3b_add(Z, g3b_run_starts(K)), Z);
        RowBits Wv = g3b_run_from(X, NS);
        {
            const RowBits rZ = g3b_rev(Z);
            const RowBits rPE = g3b_andn(g3b_add(rZ, g3b_rev(g3b_run_ends(K))), rZ);
            Wv = g3b_or(Wv, g3b_rev(g3b_run_from(g3b_rev(X), rPE)));
        }
        const bool has = g3b_any(Wv);
        const u32 wd = g3b_popc_and(DG, Wv);
        active = active && has && !ratio_exceeds(d0 + wd, a0, max_ratio);
        if (__ballot(active) == 0) break;   // uniform
        if (active) { d0 += wd; K = g3b_or(K, Wv); }
    }
    return K;
}
}  // namespace trp

namespace {
// One digit-heavy line of up to 64 * W blocks solved by the whole wavefront (every lane takes part): lane L
// classifies blocks L * W .. L * W + W - 1 of the view that begins with the line's first block, the line's
// end is the first line start after its first byte, and the solver of the short lines runs on WaveBits.
// Returns false when the line does not end inside the view.
template <int W>
__device__ inline bool g3_wave_line(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                    const u32* __restrict__ rec_lo, u64 R, u64 N, u64 nB, const DevCfg& cfg,
                                    const PlanMem& plan, u64 p, bool table, bool rab, u32* tile_bytes, u32* tile_toks,
                                    const UniTables& ut) {
    const int lane = threadIdx.x & 63;
    const u64 w0 = p >> 6;
    // The first record start after p ends the line at the latest, and no other record start matters to it: one
    // search by the whole wavefront (the records that start in the tiles of the view, 64 at a time) instead
    // of one per block.
    u64 next_rec = ~0ull;
    {
        const u64 nT = (N + TILE - 1) / TILE;
        const u64 t0 = w0 / TILE_BLOCKS, last_blk = w0 + 64ull * W - 1;
        u64 t1 = last_blk / TILE_BLOCKS + 1;
        if (t1 > nT) t1 = nT;
        u64 a = rec_lo[t0], b = rec_lo[t1];   // records [a, b) start in those tiles (all of them before N)
        if (b > R) b = R;                     // (entries left behind by invalid offsets)
        for (; a < b; a += 64) {              // uniform
            const u64 r = a + (u64)lane;
            const u64 q = r < b ? offsets[r] : ~0ull;
            const u64 m = __ballot(r < b && q > p);
            if (m) {
                const int f = ctz(m);
                next_rec = ((u64)readlane_u32((u32)(q >> 32), f) << 32) | readlane_u32((u32)q, f);
                break;
            }
        }
    }
    WaveBits<W> X, DG, AL;
    u32 my_end = 0xFFFFFFFFu;   // view-relative position of the first line start in this lane's blocks
#pragma unroll
    for (int j = 0; j < W; j++) {
        const u64 wb = w0 + (u64)(lane * W + j);
        X.w[j] = DG.w[j] = AL.w[j] = 0;
        if (wb < nB) {
            const GuardBlk blk = {bytes, wb * 64, N};
            const u64 q = wb * 64;
            const int nvalid = (int)((N - q) < 64 ? (N - q) : 64);
            const Cls c = classify_block_fast(blk, nvalid, ut);
            const u64 V = below(nvalid);
            const bool nrs = next_rec == q + 64;
            const u64 RS = (next_rec >= q && next_rec < q + 64 ? 1ull << (next_rec - q) : 0ull) & V;
            const bool has_next = q + 64 < N;
            const bool next_nl = has_next && blk[64] == '\n' && !nrs;
            const bool pt = prev_is_term(q > 0, blk[-1], nvalid > 0, blk[0], (RS & 1ull) != 0);
            u64 LS = derive(c, V, RS, next_nl, pt).LS;
            if (lane == 0 && j == 0) LS &= ~below((int)(p & 63) + 1);
            if (LS && my_end == 0xFFFFFFFFu) my_end = (u32)((lane * W + j) * 64 + ctz(LS));
            X.w[j] = ~(c.ws | (table ? c.pipe : 0ull)) & V;
            DG.w[j] = c.dig;
            AL.w[j] = c.alp;
        }
    }
    const u64 em = __ballot(my_end != 0xFFFFFFFFu);
    u32 e;
    if (em) e = readlane_u32(my_end, ctz(em));
    else if (N <= (w0 + 64ull * W) * 64) e = (u32)(N - w0 * 64);
    else return false;
    const u32 s0 = (u32)(p & 63);
#pragma unroll
    for (int j = 0; j < W; j++) {   // the line's bytes: [s0, e) of the view
        const u32 lo = (u32)((lane * W + j) * 64);
        const u32 a = s0 > lo ? (s0 - lo < 64u ? s0 - lo : 64u) : 0u, b = e > lo ? (e - lo < 64u ? e - lo : 64u) : 0u;
        const u64 m = below((int)b) & ~below((int)a);
        X.w[j] &= m; DG.w[j] &= m; AL.w[j] &= m;
    }
    const RatioDiv rt = {cfg.max_digit_ratio};
    const WaveBits<W> K = g3_bits_solve(X, DG, AL, rt);
    const WaveBits<W> KS = g3b_run_starts(K);   // kept tokens are whole tokens: the runs of K
    int first_j = -1;
#pragma unroll
    for (int j = W - 1; j >= 0; j--) if (KS.w[j]) first_j = j;
    const u64 fm = __ballot(first_j >= 0);
    const int first_lane = fm ? ctz(fm) : -1;
#pragma unroll
    for (int j = 0; j < W; j++) {
        const u64 g = K.w[j];
        if (g) {
            const u64 wb = w0 + (u64)(lane * W + j);
            u64 psp = KS.w[j], pnl = 0;
            if (lane == first_lane && j == first_j) {   // the line's first kept token: a line break, if anything
                const u64 low = psp & (0ull - psp);
                psp &= ~low;
                if (rab) pnl = low;
            }
            plan_or(plan, wb, g, g, psp, pnl);
            atomicAdd(&tile_bytes[wb / SUB_BLOCKS], (u32)(popc(g) + popc(psp | pnl)));
            atomicAdd(&tile_toks[wb / SUB_BLOCKS], (u32)popc(KS.w[j]));
        }
    }
    return true;
}

// ---- digit-heavy lines of any length -----------------------------------------------------------------
// The same solver streamed over the line in pieces of 64 blocks (4 KiB), lane i taking block i of the piece.
// Its bit vectors live in memory, four words per block (token bytes X, digits DG, kept bytes K, the bytes the
// current expansion round would add W): in the part of the output buffer that belongs to the line's own bytes
// [p, e) - k_emit fills the output later, nothing else touches that part now, and 32 bytes per 64-byte block
// fit.  A multi-word addition runs through the pieces with its carry; the mirrored additions run through them
// from the last piece to the first.  Seeding takes one pass each way, every expansion round one pass each way
// (the round's W is merged into K by the next round's first pass), the result one more.
struct HugeWords {
    u64 X, DG, K, W;
};
// a + b + cin over the 64 words of the lanes (lane 0 lowest); *cout = the carry out of lane 63
__device__ inline u64 wave_add_carry(u64 a, u64 b, u32 cin, u32* cout) {
    unsigned long long c = 0;
    u64 r = __builtin_addcll(a, b, c, &c);
    const u64 G = __ballot(c != 0), P = __ballot(r == ~0ull);
    const u64 V = (G << 1) | cin;
    const u64 C = V | ((P + V) ^ P ^ V);
    *cout = (u32)(((G >> 63) | ((P >> 63) & (C >> 63))) & 1ull);
    r += (C >> (threadIdx.x & 63)) & 1ull;
    return r;
}
__device__ inline u64 wave_mirror(u64 v) {   // the 4096-bit vector of the wavefront, end for end
    const int src = 63 - (int)(threadIdx.x & 63);
    const u32 lo = (u32)__shfl((int)(u32)v, src, 64), hi = (u32)__shfl((int)(u32)(v >> 32), src, 64);
    return bitrev64((u64)lo | ((u64)hi << 32));
}
__device__ inline u32 wave_popc_total(u64 v) { return readlane_u32(wave_incl_scan_add((u32)popc(v)), 63); }

__device__ inline void g3_huge_line(const u8* __restrict__ bytes, const u64* __restrict__ offsets, u64 R, u64 N, u64 nB,
                                    const DevCfg& cfg, const PlanMem& plan, u8* __restrict__ scratch_text, u64 p, bool table,
                                    bool rab, u32* tile_bytes, u32* tile_toks, const UniTables& ut) {
    const int lane = threadIdx.x & 63;
    const u64 w0 = p >> 6;
    const u32 s0 = (u32)(p & 63);
    HugeWords* H = reinterpret_cast<HugeWords*>(scratch_text + ((p + 7) & ~7ull));
    // end of the record that holds p (the first record start after p ends the line at the latest)
    u64 rec_end;
    {
        u64 lo = 0, hi = R;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if (offsets[mid + 1] <= p) lo = mid + 1; else hi = mid;
        }
        rec_end = lo < R ? offsets[lo + 1] : N;   // (lo == R only with offsets that break the contract)
        if (rec_end > N) rec_end = N;
    }
    // ---- first pass: classify, find the line's end, seed (the part of every token with a letter from its first
    // letter on), store X / DG / K; W holds the first letters for the way back
    u64 nchunks = 0, e = 0;   // e: end of the line relative to the view's first byte
    u32 a0 = 0;
    {
        u32 c1 = 0, c2 = 0, last_x = 0;
        for (u64 c = 0;; c++) {   // uniform
            const u64 bi = c * 64 + (u64)lane, wb = w0 + bi, q = wb * 64;
            Cls cl = {0, 0, 0, 0, 0, 0, 0};
            u64 LS = 0, V = 0;
            if (wb < nB) {
                const GuardBlk blk = {bytes, q, N};
                const int nvalid = (int)((N - q) < 64 ? (N - q) : 64);
                cl = classify_block_fast(blk, nvalid, ut);
                V = below(nvalid);
                const bool nrs = rec_end == q + 64;
                const u64 RS = (rec_end >= q && rec_end < q + 64 ? 1ull << (rec_end - q) : 0ull) & V;
                const bool next_nl = q + 64 < N && blk[64] == '\n' && !nrs;
                const bool pt = prev_is_term(q > 0, blk[-1], nvalid > 0, blk[0], (RS & 1ull) != 0);
                LS = derive(cl, V, RS, next_nl, pt).LS;
                if (bi == 0) LS &= ~below((int)s0 + 1);
            }
            const u64 em = __ballot(LS != 0);
            bool last = false;
            if (em) {
                const int f = ctz(em);
                e = (c * 64 + (u64)f) * 64 + (u64)readlane_u32((u32)ctz(LS | (1ull << 63)), f);
                last = true;
            } else if ((w0 + (c + 1) * 64) * 64 >= N) {
                e = N - w0 * 64;
                last = true;
            }
            const u64 lo = bi * 64;
            const u64 from = (u64)s0 > lo ? ((u64)s0 - lo < 64 ? (u64)s0 - lo : 64) : 0;
            const u64 to = last ? (e > lo ? (e - lo < 64 ? e - lo : 64) : 0) : 64;
            const u64 m = below((int)to) & ~below((int)from);
            const u64 X = ~(cl.ws | (table ? cl.pipe : 0ull)) & V & m, DG = cl.dig & m, AL = cl.alp & m;
            a0 += wave_popc_total(AL & X);
            const u64 Y = X & ~AL;
            const u32 x_before = from_prev_lane((u32)(X >> 63));   // (read by every lane, outside any select)
            const u64 S = X & ~((X << 1) | (lane == 0 ? (u64)last_x : (u64)x_before));
            const u64 P = wave_add_carry(Y, S, c1, &c1) & ~Y & AL;
            const u64 Kf = X & ~wave_add_carry(X, P, c2, &c2);
            if (!last || lo < e) { const HugeWords hw = {X, DG, Kf, P}; H[bi] = hw; }
            last_x = readlane_u32((u32)(X >> 63), 63);
            nchunks = c + 1;
            if (last) break;
        }
    }
    const u64 nblk = (e + 63) / 64;   // blocks of the view that hold bytes of the line
    // ---- way back: the part of every token with a letter before its first letter; the digits kept so far
    u32 d0 = 0;
    {
        u32 cb = 0;
        for (u64 c = nchunks; c-- > 0;) {
            const u64 bi = c * 64 + (u64)lane;
            HugeWords hw = {0, 0, 0, 0};
            if (bi < nblk) hw = H[bi];
            const u64 rX = wave_mirror(hw.X);
            const u64 Kb = wave_mirror(rX & ~wave_add_carry(rX, wave_mirror(hw.W), cb, &cb));
            hw.K |= Kb;
            hw.W = 0;
            if (bi < nblk) H[bi] = hw;
            d0 += wave_popc_total(hw.DG & hw.K);
        }
    }
    // ---- expansion rounds (denoiser_preprocessor.rs:94-131)
    bool merge = false;   // the previous round was accepted: its W belongs to K
    for (;;) {
        {
            u32 c1 = 0, c2 = 0, last_k = 0;
            for (u64 c = 0; c < nchunks; c++) {
                const u64 bi = c * 64 + (u64)lane;
                HugeWords hw = {0, 0, 0, 0};
                if (bi < nblk) hw = H[bi];
                if (merge) hw.K |= hw.W;
                const u64 Z = hw.K | ~hw.X;
                const u32 k_before = from_prev_lane((u32)(hw.K >> 63));
                const u64 S = hw.K & ~((hw.K << 1) | (lane == 0 ? (u64)last_k : (u64)k_before));
                const u64 NS = wave_add_carry(Z, S, c1, &c1) & ~Z;
                hw.W = hw.X & ~wave_add_carry(hw.X, NS, c2, &c2);
                if (bi < nblk) H[bi] = hw;
                last_k = readlane_u32((u32)(hw.K >> 63), 63);
            }
        }
        u32 wd = 0;
        u64 any = 0;
        {
            u32 c1 = 0, c2 = 0, next_k = 0;
            for (u64 c = nchunks; c-- > 0;) {
                const u64 bi = c * 64 + (u64)lane;
                HugeWords hw = {0, 0, 0, 0};
                if (bi < nblk) hw = H[bi];
                const u32 k_after = from_next_lane((u32)(hw.K & 1ull));
                const u64 E = hw.K & ~((hw.K >> 1) | ((lane == 63 ? (u64)next_k : (u64)k_after) << 63));
                const u64 rZ = wave_mirror(hw.K | ~hw.X);
                const u64 rPE = wave_add_carry(rZ, wave_mirror(E), c1, &c1) & ~rZ;
                const u64 rX = wave_mirror(hw.X);
                hw.W |= wave_mirror(rX & ~wave_add_carry(rX, rPE, c2, &c2));
                if (bi < nblk) H[bi].W = hw.W;
                any |= __ballot(hw.W != 0);
                wd += wave_popc_total(hw.DG & hw.W);
                next_k = readlane_u32((u32)(hw.K & 1ull), 0);
            }
        }
        if (!any) break;
        if (ratio_exceeds(d0 + wd, a0, cfg.max_digit_ratio)) break;
        d0 += wd;
        merge = true;
    }
    // ---- the kept tokens into the plan
    {
        u32 last_k = 0;
        bool first_done = false;
        for (u64 c = 0; c < nchunks; c++) {
            const u64 bi = c * 64 + (u64)lane;
            u64 K = 0;
            if (bi < nblk) K = H[bi].K;
            const u32 k_before = from_prev_lane((u32)(K >> 63));
            const u64 KS = K & ~((K << 1) | (lane == 0 ? (u64)last_k : (u64)k_before));
            const u64 fm = __ballot(KS != 0);
            u64 psp = KS, pnl = 0;
            if (!first_done && fm) {
                if (lane == ctz(fm)) {   // the line's first kept token: a line break, if anything
                    const u64 low = psp & (0ull - psp);
                    psp &= ~low;
                    if (rab) pnl = low;
                }
                first_done = true;
            }
            if (K) {
                const u64 wb = w0 + bi;
                plan_or(plan, wb, K, K, psp, pnl);
                atomicAdd(&tile_bytes[wb / SUB_BLOCKS], (u32)(popc(K) + popc(psp | pnl)));
                atomicAdd(&tile_toks[wb / SUB_BLOCKS], (u32)popc(KS));
            }
            last_k = readlane_u32((u32)(K >> 63), 63);
        }
    }
}

// Up to four listed lines at once, one per DPP row: lane 16 r + l classifies block l of the 16-block view of line
// ent[k0 + r].  Returns the rows whose line ended inside its view and was solved (bit r); the others are left
// to g3_wave_line.  Every lane takes part.
__device__ inline u32 g3_rows_lines(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                    const u32* __restrict__ rec_lo, u64 R, u64 N, u64 nB, const DevCfg& cfg,
                                    const PlanMem& plan, const u64* ent, u32 k0, u32 np, u32* tile_bytes, u32* tile_toks,
                                    const UniTables& ut) {
    const int lane = threadIdx.x & 63, row = lane >> 4, l = lane & 15, rowbase = lane & 48;
    const bool have = k0 + (u32)row < np;
    const u64 e64 = have ? ent[k0 + row] : 0;
    const u64 p = e64 & 0xFFFFFFFFull;
    const bool table = (e64 >> 32) & 1, rab = (e64 >> 33) & 1;
    const u64 w0 = p >> 6, wb = w0 + (u64)l;
    // the first record start after p (see g3_wave_line), searched by the row
    u64 next_rec = ~0ull;
    {
        const u64 nT = (N + TILE - 1) / TILE;
        u64 t1 = (w0 + 15) / TILE_BLOCKS + 1;
        if (t1 > nT) t1 = nT;
        u64 a = have ? rec_lo[w0 / TILE_BLOCKS] : 0, b = have ? rec_lo[t1] : 0;
        if (b > R) b = R;
        for (;;) {   // uniform
            const u64 r = a + (u64)l;
            const u64 q = r < b ? offsets[r] : ~0ull;
            const u32 m = row_bits(__ballot(r < b && q > p));
            const int f = rowbase + (m ? ctz((u64)m) : 0);
            const u64 qf = ((u64)(u32)__shfl((int)(u32)(q >> 32), f, 64) << 32) | (u32)__shfl((int)(u32)q, f, 64);
            if (m && next_rec == ~0ull) next_rec = qf;
            a += 16;
            if (__ballot(next_rec == ~0ull && a < b) == 0) break;
        }
    }
    RowBits X = {0}, DG = {0}, AL = {0};
    u32 my_end = 0xFFFFFFFFu;
    if (have && wb < nB) {
        const GuardBlk blk = {bytes, wb * 64, N};
        const u64 q = wb * 64;
        const int nvalid = (int)((N - q) < 64 ? (N - q) : 64);
        const Cls c = classify_block_fast(blk, nvalid, ut);
        const u64 V = below(nvalid);
        const bool nrs = next_rec == q + 64;
        const u64 RS = (next_rec >= q && next_rec < q + 64 ? 1ull << (next_rec - q) : 0ull) & V;
        const bool has_next = q + 64 < N;
        const bool next_nl = has_next && blk[64] == '\n' && !nrs;
        const bool pt = prev_is_term(q > 0, blk[-1], nvalid > 0, blk[0], (RS & 1ull) != 0);
        u64 LS = derive(c, V, RS, next_nl, pt).LS;
        if (l == 0) LS &= ~below((int)(p & 63) + 1);
        if (LS) my_end = (u32)(l * 64 + ctz(LS));
        X.w = ~(c.ws | (table ? c.pipe : 0ull)) & V;
        DG.w = c.dig;
        AL.w = c.alp;
    }
    const u32 em = row_bits(__ballot(my_end != 0xFFFFFFFFu));
    const u32 e_found = (u32)__shfl((int)my_end, rowbase + (em ? ctz((u64)em) : 0), 64);
    u32 e = 0;
    bool ok = have;
    if (em) e = e_found;
    else if (N <= (w0 + 16) * 64) e = (u32)(N - w0 * 64);
    else ok = false;
    {
        const u32 s0 = (u32)(p & 63), lo = (u32)(l * 64);
        const u32 a = s0 > lo ? (s0 - lo < 64u ? s0 - lo : 64u) : 0u, b = e > lo ? (e - lo < 64u ? e - lo : 64u) : 0u;
        const u64 m = ok ? below((int)b) & ~below((int)a) : 0ull;
        X.w &= m; DG.w &= m; AL.w &= m;
    }
    const RowBits K = g3_rows_solve(X, DG, AL, cfg.max_digit_ratio, ok);
    const RowBits KS = g3b_run_starts(K);
    const u32 fm = row_bits(__ballot(KS.w != 0));
    if (ok && K.w) {
        u64 psp = KS.w, pnl = 0;
        if (fm && l == ctz((u64)fm)) {   // the line's first kept token: a line break, if anything
            const u64 low = psp & (0ull - psp);
            psp &= ~low;
            if (rab) pnl = low;
        }
        plan_or(plan, wb, K.w, K.w, psp, pnl);
        atomicAdd(&tile_bytes[wb / SUB_BLOCKS], (u32)(popc(K.w) + popc(psp | pnl)));
        atomicAdd(&tile_toks[wb / SUB_BLOCKS], (u32)popc(KS.w));
    }
    const u64 okm = __ballot(ok);
    return (u32)((okm & 1) | ((okm >> 15) & 2) | ((okm >> 30) & 4) | ((okm >> 45) & 8));
}

// Lines that k_plan could not solve inside a tile (they cross a tile edge or are too long): twelve listed
// lines per wavefront (one wavefront per workgroup).  Lane 5 * l + j re-classifies block j of line l's first
// five blocks from global memory into the line's mask arrays in LDS; lane 5 * l then runs the same solver
// as k_plan; lines of more than G3_MAXW blocks are then solved by the lanes together (g3_rows_lines, g3_wave_line,
// g3_huge_line).
constexpr u32 G3_VIEW = G3_MAXW + 1;
constexpr int G3S_LINES = 64 / G3_VIEW;             // lines per wavefront
constexpr int G3S_STRIDE = 8 * (G3_VIEW + 1) + 1;   // u64 per line; odd, so that the lines spread over the banks
// One wavefront's working space (the kernel's workgroups are single wavefronts, synchronised as such).
struct G3SolveShared {
    u64 masks[G3S_LINES * G3S_STRIDE];
    u64 pend[G3S_LINES];   // lines of the group that are too long for one lane: entries of the list
    u32 npend;
};
// vb / vgrid: this wavefront's number and the number of wavefronts that share the list.
__device__ __forceinline__ void gate3_solve_body(G3SolveShared& gsh, u32 vb, u32 vgrid, const u8* __restrict__ bytes,
                                                 const u64* __restrict__ offsets, const u32* __restrict__ rec_lo, u64 R,
                                                 u64 N, u64 nB, const DevCfg& cfg, const PlanMem& plan,
                                                 const u64* __restrict__ list, const u64* __restrict__ mark_total, u64 cap,
                                                 u32* tile_bytes, u32* tile_toks, u64* __restrict__ totals,
                                                 u8* __restrict__ scratch_text) {
    u64* const masks = gsh.masks;
    u64* const pend = gsh.pend;
    u32& npend = gsh.npend;
    const int lane = threadIdx.x & 63;
    u64 n = *mark_total;
    if (vb == 0 && lane == 0) { totals[4] = n; totals[5] = n > cap ? 1 : 0; }
    if (n > cap) n = cap;
    const int l = lane / (int)G3_VIEW, j = lane % (int)G3_VIEW;   // line of the group, block of its view
    u64* m = masks + (l < G3S_LINES ? l : 0) * G3S_STRIDE;        // eight arrays of G3_VIEW + 1 words
    const G3Masks gm = {m, m + (G3_VIEW + 1), m + 2 * (G3_VIEW + 1), m + 3 * (G3_VIEW + 1), m + 4 * (G3_VIEW + 1),
                        m + 5 * (G3_VIEW + 1), m + 6 * (G3_VIEW + 1), m + 7 * (G3_VIEW + 1)};
    const UniTables ut = dev_tables();
    for (u64 i0 = (u64)vb * G3S_LINES; i0 < n; i0 += (u64)vgrid * G3S_LINES) {   // uniform
        const u64 i = i0 + (u64)l;
        const bool have = l < G3S_LINES && i < n;
        u64 p = 0;
        bool table = false, rab = false;
        if (lane == 0) npend = 0;
        if (have) {
            const u64 ent = list[i];
            p = ent & 0xFFFFFFFFull;
            table = (ent >> 32) & 1;
            rab = (ent >> 33) & 1;
        }
        const u64 w0 = p >> 6;
        if (have) {
            const u64 wb = w0 + (u64)j;
            Cls c = {0, 0, 0, 0, 0, 0, 0};
            u64 LS = 0;
            if (wb < nB) {
                const GuardBlk blk = {bytes, wb * 64, N};
                const u64 q = wb * 64;
                const int nvalid = (int)((N - q) < 64 ? (N - q) : 64);
                c = classify_block_fast(blk, nvalid, ut);
                const u64 V = below(nvalid);
                bool nrs;
                const u64 RS = block_record_starts(offsets, rec_lo, R, N, wb, &nrs) & V;
                const bool has_next = q + 64 < N;
                const bool next_nl = has_next && blk[64] == '\n' && !nrs;
                const bool pt = prev_is_term(q > 0, blk[-1], nvalid > 0, blk[0], (RS & 1ull) != 0);
                LS = derive(c, V, RS, next_nl, pt).LS;
            }
            gm.ws[j] = c.ws; gm.pipe[j] = c.pipe; gm.dig[j] = c.dig; gm.alp[j] = c.alp; gm.ls[j] = LS;
            gm.G[j] = 0; gm.gPSP[j] = 0; gm.gPNL[j] = 0;
        }
        wave_lds_sync();   // the masks of a line are written by five lanes and read by the first of them
        if (have && j == 0) {
            const u64 view0 = w0 * 64;
            const u32 cend = (N - view0 <= (u64)G3_VIEW * 64) ? (u32)(N - view0) : 0xFFFFFFFFu;
            const RatioDiv rt = {cfg.max_digit_ratio};
            const bool ok = g3_lane_solve<G3_WORDS>(gm, (u32)(p & 63), table, rab, G3_VIEW, cend, rt);
            const u64 mb = ~(1ull << (p & 63));   // the marker's flag bits
            atomicAnd(reinterpret_cast<unsigned long long*>(&plan.px[w0].PSP), mb);
            atomicAnd(reinterpret_cast<unsigned long long*>(&plan.px[w0].PNL), mb);
            if (ok) {
                u64 prevg = 0;
#pragma unroll 1
                for (u32 jj = 0; jj < G3_VIEW; jj++) {
                    const u64 g = gm.G[jj];
                    if (g) {
                        const u64 wb = w0 + jj, sp = gm.gPSP[jj], nl = gm.gPNL[jj];
                        plan_or(plan, wb, g, g, sp, nl);
                        atomicAdd(&tile_bytes[wb / SUB_BLOCKS], (u32)(popc(g) + popc(sp | nl)));
                        // token starts of this block: kept bytes whose predecessor is not kept
                        atomicAdd(&tile_toks[wb / SUB_BLOCKS], (u32)popc(g & ~((g << 1) | prevg)));
                    }
                    prevg = g >> 63;
                }
            } else {
                pend[atomicAdd(&npend, 1u)] = list[i];
            }
        }
        wave_lds_sync();   // the line's lane has read the masks before the next group overwrites them
        // the long lines of the group, one after the other, by the whole wavefront (up to 4 KiB, then up to 16 KiB);
        // what is longer still is walked by one lane
        const u32 np = npend;
        for (u32 k0 = 0; k0 < np; k0 += 4) {   // uniform: four lines at a time while they fit 1 KiB views
            const u32 solved = g3_rows_lines(bytes, offsets, rec_lo, R, N, nB, cfg, plan, pend, k0, np, tile_bytes, tile_toks, ut);
            for (u32 k = k0; k < np && k < k0 + 4; k++) {
                if ((solved >> (k - k0)) & 1u) continue;
                const u64 ent = pend[k];
                const u64 lp = ent & 0xFFFFFFFFull;
                const bool ltable = (ent >> 32) & 1, lrab = (ent >> 33) & 1;
                bool done = g3_wave_line<1>(bytes, offsets, rec_lo, R, N, nB, cfg, plan, lp, ltable, lrab, tile_bytes, tile_toks, ut);
                if (!done) done = g3_wave_line<4>(bytes, offsets, rec_lo, R, N, nB, cfg, plan, lp, ltable, lrab, tile_bytes, tile_toks, ut);
                if (!done) {
                    if (scratch_text) g3_huge_line(bytes, offsets, R, N, nB, cfg, plan, scratch_text, lp, ltable, lrab, tile_bytes, tile_toks, ut);
                    else if (lane == 0) g3_serial_solve(bytes, offsets, R, N, cfg, plan, lp, ltable, lrab, tile_bytes, tile_toks);
                }
            }
        }
        wave_lds_sync();   // npend and pend are free again
    }
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_gate3_solve(const u8* __restrict__ bytes,
                                                     const u64* __restrict__ offsets, const u32* __restrict__ rec_lo, u64 R, u64 N, u64 nB, DevCfg cfg,
                                                     PlanMem plan, const u64* __restrict__ list,
                                                     const u64* __restrict__ mark_total, u64 cap, u32* tile_bytes,
                                                     u32* tile_toks, u64* __restrict__ totals, u8* __restrict__ scratch_text) {
    __shared__ G3SolveShared gsh;
    gate3_solve_body(gsh, blockIdx.x, gridDim.x, bytes, offsets, rec_lo, R, N, nB, cfg, plan, list, mark_total, cap, tile_bytes,
                     tile_toks, totals, scratch_text);
}

// ---------------------------------------------------------------------------------------------------
// generic exclusive scan of u64 (3 kernels), out has n+1 entries
// ---------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 2;    // small chunks: these kernels are short and latency-bound, more workgroups finish sooner
constexpr int SCAN_CHUNK = NT * SCAN_ITEMS;

__global__ __launch_bounds__(NT) void k_scan_reduce(const u64* __restrict__ in, u64 n, u64* __restrict__ partial) {
    __shared__ u64 red[NT / 64];
    u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x * SCAN_ITEMS;
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) if (base + i < n) s += in[base + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 tot = 0;
        for (int w = 0; w < NT / 64; w++) tot += red[w];
        partial[blockIdx.x] = tot;
    }
}

__device__ inline u64 block_excl_sum_u64(u64 v, u64* smem, u64* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    u64 inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u64 o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    u64 pre = 0, tot = 0;
    for (int w = 0; w < nw; w++) { if (w < wave) pre += smem[w]; tot += smem[w]; }
    __syncthreads();
    *total = tot;
    return pre + inc - v;
}

// single workgroup: exclusive scan of partial[0..nb) in place
__global__ __launch_bounds__(1024) void k_scan_partials(u64* partial, u64 nb) {
    __shared__ u64 smem[16];
    u64 carry = 0;
    for (u64 base = 0; base < nb; base += 1024) {
        u64 i = base + threadIdx.x;
        u64 v = i < nb ? partial[i] : 0;
        u64 tot;
        u64 ex = block_excl_sum_u64(v, smem, &tot);
        if (i < nb) partial[i] = carry + ex;
        carry += tot;
    }
}

__global__ __launch_bounds__(NT) void k_scan_final(const u64* __restrict__ in, u64 n, const u64* __restrict__ partial,
                                                    u64* __restrict__ out) {
    __shared__ u64 smem[NT / 64];
    u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x * SCAN_ITEMS;
    u64 v[SCAN_ITEMS];
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    u64 tot;
    u64 ex = block_excl_sum_u64(s, smem, &tot) + partial[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
        if (base + i + 1 == n) out[n] = ex;
    }
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}

// The two scans of the path in two kernels each.  k_scan_reduce_gen computes the values to be scanned with a
// generator (which also stores them for the second kernel) and reduces them per workgroup; k_scan_final_sum
// adds up the partial sums before its own workgroup itself instead of waiting for a single-workgroup scan of
// them in between (a few loads per thread: there are n / SCAN_CHUNK partial sums; beyond SCAN_SUM_MAX of them the
// three-kernel form is used).  Letting the last workgroup of the first kernel scan the partial sums instead was
// measured slower: the device-wide fence it needs writes back every workgroup's L2 lines.
//
// GenTiles: sub-tile counts -> packed u64 (bytes | toks << 32); a token that continues across a sub-tile edge
// was counted as a start by the right-hand sub-tile: taken back here.
struct GenTiles {
    const u32 *bytes, *toks, *flags;
    u64* packed;
    __device__ u64 operator()(u64 t) const {
        u32 tk = toks[t];
        if (t > 0 && (flags[t] & TF_FIRST_CONT) && (flags[t - 1] & TF_LAST_TOK)) tk -= 1;
        const u64 v = (u64)bytes[t] | ((u64)tk << 32);
        packed[t] = v;
        return v;
    }
    struct State {};
    __device__ State begin() const { return State(); }
    __device__ u64 at(const State&, u64 t) const { return (*this)(t); }
    __device__ void once() const {}
};
// GenChunkCounts: per record the number of chunks; on the way the totals, the offsets of records that start at
// the very end of the corpus (k_emit only sees records that start inside a sub-tile) and the dropped flags.
struct GenChunkCounts {
    const u64* tile_base;
    u64 nW;
    const u32* rec_lo_w;
    u64 R, N;
    int passthrough;
    DevCfg cfg;
    u64 *den_off, *tok_off, *totals, *counts;
    u8* dropped;
    struct State { u64 D, T, first; };
    __device__ State begin() const {
        State e;
        e.D = nW ? (tile_base[nW] & 0xFFFFFFFFull) : 0;
        e.T = nW ? (tile_base[nW] >> 32) : 0;
        if (passthrough) e.D = N;
        e.first = rec_lo_w[nW + 1];   // first record whose start offset is >= N
        return e;
    }
    __device__ u64 at(const State& e, u64 r) const {
        const u64 D = e.D, T = e.T, first = e.first;
        // entries at and after `first` are written here, by their own thread: neighbours never read them
        const u64 d0 = r >= first ? D : den_off[r], t0 = r >= first ? T : tok_off[r];
        const u64 d1 = r + 1 >= first ? D : den_off[r + 1], t1 = r + 1 >= first ? T : tok_off[r + 1];
        if (r >= first) { den_off[r] = D; tok_off[r] = T; }
        const u64 n = chunk_count(t1 - t0, cfg);
        counts[r] = n;
        dropped[r] = cfg.denoise_enabled && !cfg.normalize && d1 == d0;
        return n;
    }
    __device__ void once() const {
        const State e = begin();
        totals[0] = e.D;
        totals[1] = e.T;
        den_off[R] = e.D;
        tok_off[R] = e.T;
    }
};

constexpr u64 SCAN_SUM_MAX = 8192;
// Inputs of at most SCAN_SMALL_MAX values (batches of a megabyte or two, a thousand records) are scanned by one
// workgroup in one launch: at that size a batch is bound by the number of dependent launches, not by bytes.
constexpr u64 SCAN_SMALL_MAX = 1024;   // (2048 values: 12.7 us in one workgroup against 9.8 us for the two kernels)
template <class Gen>
__global__ __launch_bounds__(NT) void k_scan_reduce_gen(Gen gen, u64 n, u64* __restrict__ partial) {
    __shared__ u64 red[NT / 64];
    // consecutive threads take consecutive elements (the order within the workgroup's chunk does not matter here)
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x;
    const typename Gen::State st = gen.begin();
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) if (base + (u64)i * NT < n) s += gen.at(st, base + (u64)i * NT);
    if (blockIdx.x == 0 && threadIdx.x == 0) gen.once();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 tot = 0;
        for (int w = 0; w < NT / 64; w++) tot += red[w];
        partial[blockIdx.x] = tot;
    }
}
template <class Gen>
__device__ __forceinline__ void scan_small_body(u64* smem, const Gen& gen, u64 n, u64* __restrict__ out) {
    const typename Gen::State st = gen.begin();
    u64 carry = 0;
    for (u64 base = 0; base < n; base += NT) {   // uniform
        const u64 i = base + threadIdx.x;
        const u64 v = i < n ? gen.at(st, i) : 0;
        u64 tot;
        const u64 ex = block_excl_sum_u64(v, smem, &tot);   // barriers inside
        if (i < n) out[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        out[n] = carry;
        gen.once();
    }
}
template <class Gen>
__global__ __launch_bounds__(NT) void k_scan_small_gen(Gen gen, u64 n, u64* __restrict__ out) {
    __shared__ u64 smem[NT / 64];
    scan_small_body(smem, gen, n, out);
}
// k_scan_final with the workgroup's base summed from the (unscanned) partial sums of the workgroups before it
__global__ __launch_bounds__(NT) void k_scan_final_sum(const u64* __restrict__ in, u64 n, const u64* __restrict__ partial,
                                                        u64* __restrict__ out) {
    __shared__ u64 smem[NT / 64];
    __shared__ u64 bsum[NT / 64];
    u64 pre = 0;
    for (u64 i = threadIdx.x; i < blockIdx.x; i += NT) pre += partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_down(pre, off, 64);
    if ((threadIdx.x & 63) == 0) bsum[threadIdx.x >> 6] = pre;
    u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x * SCAN_ITEMS;
    u64 v[SCAN_ITEMS];
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) { v[i] = base + i < n ? in[base + i] : 0; s += v[i]; }
    u64 tot;
    u64 ex = block_excl_sum_u64(s, smem, &tot);   // barriers inside: bsum is complete afterwards
    for (int w = 0; w < NT / 64; w++) ex += bsum[w];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) out[base + i] = ex;
        ex += v[i];
        if (base + i + 1 == n) out[n] = ex;
    }
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) out[0] = 0;
}

// ---------------------------------------------------------------------------------------------------
// k_emit: stream compaction of the tile per its plan -> denoised bytes, token table, per-record offsets
// ---------------------------------------------------------------------------------------------------

// four u32 that are only 4-byte aligned in memory
struct __attribute__((packed, aligned(4))) Quad32 { u32 x, y, z, w; };
struct Pair32 { u32 lo, hi; };   // two consecutive dwords at a 4-byte aligned address

// ---- sub-tile staging (one wavefront, 64 half blocks of 32 bytes) -------------------------------------
// LDS image of a sub-tile: the 32 bytes of lane l live at l * HALF_STRIDE (32 data bytes + 4 unused, so that
// lanes that touch the same dword of their half blocks fall into different banks: 9 dwords per lane).  One
// spare slot at the end: the aligned reads of a run copy may touch the dword after a half block.
constexpr int HALF_STRIDE = 36;
struct SubShared {
    alignas(16) u8 bytes[(64 + 1) * HALF_STRIDE];
};
struct SubRs {
    u32 w[64 + 1];   // record starts, one bit per byte: w[l] covers lane l; bit 0 of w[64]: the byte after the sub-tile
};
struct SubRegs {
    uint4 v[SUB / 16 / 64];
};
__device__ inline void sub_load(const u8* __restrict__ g, u64 N, u64 sS, SubRegs& r) {
    const int lane = threadIdx.x & 63;
    if (sS + SUB <= N) {
        const uint4* src = reinterpret_cast<const uint4*>(g + sS);
#pragma unroll
        for (int k = 0; k < SUB / 16 / 64; k++) r.v[k] = src[k * 64 + lane];
        return;
    }
#pragma unroll
    for (int k = 0; k < SUB / 16 / 64; k++) r.v[k] = load16_guarded(g, N, sS + (u64)(k * 64 + lane) * 16);
}
__device__ inline void sub_store(const SubRegs& r, u8* sb) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < SUB / 16 / 64; k++) {
        const int c = k * 64 + lane;   // 16-byte chunk c: half block c / 2, its first or second 16 bytes
        lds_store16(sb + (c >> 1) * HALF_STRIDE + (c & 1) * 16, r.v[k]);
    }
}
// Record starts of sub-tile s: records [rec_lo_w[s], rec_lo_w[s+1]) start inside it (see rs_fetch / rs_store).
__device__ inline u32 sub_rs_fetch(const u64* __restrict__ offsets, const u32* __restrict__ rec_lo_w, u64 R, u64 s,
                                   RsFetch& f) {
    const RecRange g = rec_range(rec_lo_w, s, R);
    const u32 r0 = g.r0, r1 = g.r1;
    const u64 r = (u64)r0 + (threadIdx.x & 63);
    f.q = ~0u;
    f.n = r1 - r0;
    if (r <= r1 && r < R) f.q = (u32)offsets[r];
    return r0;
}
__device__ inline void sub_rs_set(SubRs& rs, u64 q, u64 N, u64 sS) {
    const u64 d = q - sS;
    if (q < N && d <= (u64)SUB) atomicOr(&rs.w[d >> 5], 1u << (d & 31));
}
__device__ inline void sub_rs_store(const RsFetch& f, const u64* __restrict__ offsets, const u32* __restrict__ rec_lo_w,
                                    u64 R, u64 N, u64 s, SubRs& rs) {
    const u64 sS = s * (u64)SUB;
    for (int i = threadIdx.x & 63; i <= 64; i += 64) rs.w[i] = 0;
    if (f.q != ~0u) sub_rs_set(rs, f.q, N, sS);
    if (f.n >= 64) {   // crowded sub-tile: the rest of its records
        const u64 r0 = rec_range(rec_lo_w, s, R).r0, r1 = (u64)r0 + f.n;
        for (u64 r = r0 + 64 + (threadIdx.x & 63); r <= r1 && r < R; r += 64) sub_rs_set(rs, offsets[r], N, sS);
    }
}

// Once the run copies are done the input image is dead; its space then holds the staged token table: 18 tokens
// per 64 bytes on average (English text has 11); the tokens of a denser sub-tile beyond that are stored directly.
constexpr u32 TOK_CAP = sizeof(SubShared) / sizeof(u32);

struct EmitShared {
    SubShared tile;
    // output image of the sub-tile (linear).  Before the first byte is copied into it, its space holds the
    // record-start words (SubRs), which every lane has moved to registers by then.
    alignas(16) u8 out[SUB + 64];
    u64 cin_start;      // corpus position where the token that runs into this sub-tile starts
};
static_assert(sizeof(SubRs) <= SUB + 64, "record-start words must fit in the output image");

// Position where the token covering byte tS-1 starts (the sub-tile's first byte continues it): a walk back over
// the token planes of the blocks before, at most to the start of the record.  The whole wavefront walks, 64
// blocks at a time (every lane calls this): the nearest block with a byte that is not a token byte, or with the
// record start, answers.  (One lane walking alone made every sub-tile inside a token of megabytes - a base64
// blob - take time in proportion to its distance from the token's start.)
__device__ inline u64 find_token_start_before(const PlanCT* __restrict__ plan, u64 tS, u64 rec_start) {
    const int lane = threadIdx.x & 63;
    long long nearest = (long long)(tS / 64) - 1;
    while (nearest >= 0) {   // uniform
        const long long bb = nearest - lane;
        bool cand = false;
        u64 ans = 0;
        if (bb >= 0) {
            const u64 w = plan[bb].TOK & plan[bb].C;
            const u64 lo = (u64)bb * 64;
            // bits at/after the record start (a lane beyond the block that holds it is never the nearest answer)
            const u64 allow = rec_start > lo ? (rec_start - lo < 64 ? ~below((int)(rec_start - lo)) : 0ull) : ~0ull;
            const u64 zero = ~w & allow;
            if (zero) { cand = true; ans = lo + (u64)top(zero) + 1; }
            else if (rec_start >= lo) { cand = true; ans = rec_start; }
        }
        const u64 m = __ballot(cand);
        if (m) {
            const int f = ctz(m);
            return ((u64)readlane_u32((u32)(ans >> 32), f) << 32) | readlane_u32((u32)ans, f);
        }
        nearest -= 64;
    }
    return rec_start;
}

// What a wavefront fetches from global memory for its sub-tile.  Lane l owns half h = l & 1 of block
// 32 s + (l >> 1): the 32 bits of the block's plan words that cover its bytes.
struct EmitIn {
    SubRegs regs;
    PlanCT ct;            // the block's copy and token planes (both halves)
    RsFetch rf;           // record starts inside the sub-tile
    u32 rec0;             // first record that starts inside the sub-tile
    u64 prev_w;           // lane 0: token bytes of the block before the sub-tile
    u32 prev_rec;         // lane 0: start of the last record before the sub-tile
    u64 next_w;           // last lane: token bytes of the block after the sub-tile
    u64 base;             // scanned output byte / token base of the sub-tile
    u32 pfx;              // blocks of the sub-tile whose separator planes are stored (bit i: block 32 s + i)
};
__device__ inline void emit_fetch(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                  const u32* __restrict__ rec_lo_w, u64 R, const PlanMem& plan,
                                  const u64* __restrict__ tile_base, u64 N, u64 nB, u64 s, bool passthrough, EmitIn& in) {
    const int lane = threadIdx.x & 63;
    const u64 b = s * (u64)SUB_BLOCKS + (lane >> 1);
    if (!passthrough) sub_load(bytes, N, s * (u64)SUB, in.regs);   // nothing is copied when the denoiser is off
    const u32 r0 = sub_rs_fetch(offsets, rec_lo_w, R, s, in.rf);
    in.rec0 = r0;
    const PlanCT z = {0, 0};
    in.ct = z;
    if (b < nB) in.ct = plan.ct[b];
    in.pfx = passthrough ? 0u : (u32)(plan.pfx_mask[s >> 1] >> (32 * (s & 1)));
    in.prev_w = in.next_w = 0;
    in.prev_rec = 0;
    if (lane == 0 && b > 0) {
        if (r0 > 0) in.prev_rec = (u32)offsets[r0 - 1];
        in.prev_w = plan.ct[b - 1].TOK & plan.ct[b - 1].C;
    }
    if (lane == 63 && b + 1 < nB) in.next_w = plan.ct[b + 1].TOK & plan.ct[b + 1].C;
    in.base = tile_base[s];
}

// Inclusive maximum over the lanes up to this one (values >= 0; same lane pattern as wave_incl_scan_add).
__device__ inline u32 wave_incl_scan_max(u32 v) {
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}
__device__ inline u32 below32(u32 j) { return j >= 32 ? ~0u : ((1u << j) - 1u); }   // bits < j

// One wavefront per sub-tile (2 KiB), one lane per 32 bytes.  Nothing crosses wavefronts, so there are no
// workgroup barriers: a wavefront only orders its own LDS traffic, and the waves of a CU drift apart and overlap
// each other's load, compute and store phases.  A workgroup is WAVES such wavefronts that share nothing but
// their launch: one-wave workgroups left the kernel waiting for the workgroup dispatcher (487,000 workgroups per
// GB: 0.72 / 0.69 / 0.67 / 0.68 / 0.72 ms for 1 / 2 / 4 / 8 / 16 waves per workgroup on 1 GB of English text).  The kernel is latency-bound: its time falls with the
// number of resident waves (1.05 / 0.83 / 0.75 / 0.69 ms at 9 / 12 / 15 / 18 waves per CU with 64-byte lanes on
// 1 GB of English text), hence the small lanes: 4.4 KB of LDS and about 50 VGPRs per wave allow 8 per SIMD.
// (Several sub-tiles per wavefront with the loads of the next one in flight were measured slower: fresh
// wavefronts hide the load latency better than the registers a prefetch holds.)
// PASSTHROUGH: the denoiser is off, the text is the input itself: nothing is copied, k_count_tokens has left the
// token plane in the plan.
constexpr int EMIT_WAVES = 4;
// Sub-tile t by the calling wavefront (sh: its own LDS).
template <bool PASSTHROUGH>
__device__ __forceinline__ void emit_sub(EmitShared& sh, u64 t, const u8* __restrict__ bytes,
                                         const u64* __restrict__ offsets, const u32* __restrict__ rec_lo_w, u64 R, u64 N,
                                         u64 nB, u64 nW, const PlanMem& plan, const u64* __restrict__ tile_base,
                                         u8* __restrict__ denoised, u32* __restrict__ tok_start, u32* __restrict__ tok_len,
                                         u64* __restrict__ den_off, u64* __restrict__ tok_off, u32 stop_after) {
    const int tid = threadIdx.x & 63;   // = lane
    if (t >= nW) return;
    SubRs& srs = *reinterpret_cast<SubRs*>(sh.out);
    EmitIn in;
    emit_fetch(bytes, offsets, rec_lo_w, R, plan, tile_base, N, nB, t, PASSTHROUGH, in);
    const u32 rec0 = uni(in.rec0), rec_n = uni(in.rf.n);   // records [rec0, rec0 + rec_n) start in this sub-tile
    const u64 tS = t * (u64)SUB;
    const u64 b = t * (u64)SUB_BLOCKS + (tid >> 1);   // the lane's block; it owns half `hi` of it
    const int hi = tid & 1;
    // the separator planes of the few blocks that have them: a second, dependent load, issued before the
    // sub-tile is stored to LDS
    PlanPX px = {0, 0};
    if (!PASSTHROUGH && ((in.pfx >> (tid >> 1)) & 1u) && b < nB) px = plan.px[b];
    if (!PASSTHROUGH) sub_store(in.regs, sh.tile.bytes);
    sub_rs_store(in.rf, offsets, rec_lo_w, R, N, t, srs);
    const u8* blk = sh.tile.bytes + tid * HALF_STRIDE;
    const u64 p = tS + (u64)tid * HALF;
    const int nvalid = p >= N ? 0 : (int)((N - p) < (u64)HALF ? (N - p) : (u64)HALF);
    const u32 V = below32((u32)nvalid);
    const u32 Cm = (u32)(in.ct.C >> (32 * hi)) & V;
    const u32 TOKC = (u32)(in.ct.TOK >> (32 * hi)) & Cm;
    const u32 PSP = (u32)(px.PSP >> (32 * hi)), PNL = (u32)(px.PNL >> (32 * hi));
    const u32 PFX = (PSP | PNL) & Cm;
    const u64 prev_w = in.prev_w, prev_rec = in.prev_rec;
    const bool edge_prev = tid == 0 && (prev_w >> 63) != 0, edge_next = (in.next_w & 1ull) != 0;
    const u64 tb64 = (u64)uni((u32)in.base) | ((u64)uni((u32)(in.base >> 32)) << 32);   // the same in every lane
    wave_lds_sync();
    const u32 RS = srs.w[tid] & V;
    const bool next_rs = !(p + HALF < N) || (srs.w[tid + 1] & 1u);
    wave_lds_sync();   // the record-start words are in registers: their space is the output image from here on
    if (!PASSTHROUGH) {   // the image starts out as zeros (the run copies OR their partial dwords into it)
        uint4* z = reinterpret_cast<uint4*>(sh.out);
        const uint4 zero = make_uint4(0, 0, 0, 0);
        z[tid] = zero;
        z[tid + 64] = zero;
        if (tid < (int)((SUB + 64) / 16 - 128)) z[tid + 128] = zero;
    }
    // stop_after != 0 (profiling aid, see trp_ctx::emit_stop) ends every sub-tile after that phase
    if (stop_after == 1) return;
    // the neighbours' edge bits through the cross-lane network
    const u32 up = from_prev_lane(TOKC >> 31), down = from_next_lane(TOKC & 1u);   // (lanes 0 / 63 get 0 and use the edge words)
    const bool prev_tok = tid > 0 ? up != 0 : edge_prev;
    const bool next_tok = tid < 63 ? down != 0 : edge_next;
    const u32 TS = TOKC & (~((TOKC << 1) | (prev_tok ? 1u : 0u)) | RS);
    const u32 TE = TOKC & (~((TOKC >> 1) | (next_tok ? 0x80000000u : 0u)) | ((RS >> 1) | (next_rs ? 0x80000000u : 0u)));
    // wave prefix sums: emitted bytes | token starts << 16, token ends; prefix maximum: last token start + 1
    const u32 me_nb = (u32)__builtin_popcount(Cm) + (u32)__builtin_popcount(PFX), me_ntk = (u32)__builtin_popcount(TS);
    const u32 me_a = me_nb | (me_ntk << 16);
    const u32 inc_a = wave_incl_scan_add(me_a);
    const u32 inc_l = wave_incl_scan_max(TS ? (u32)(tid * HALF + 31 - __builtin_clz(TS) + 1) : 0u);
    const u32 out0 = PASSTHROUGH ? (u32)tS : (u32)(tb64 & 0xFFFFFFFFull);   // output position of the sub-tile's first byte
    const u32 tok0 = (u32)(tb64 >> 32);                                      // index of the first token that starts in it
    const u32 ex_a = inc_a - me_a;
    const u32 ex_nb = out0 + (ex_a & 0xFFFFu), ex_ntk = tok0 + (ex_a >> 16);
    const u32 ex_l = from_prev_lane(inc_l);   // (0 in lane 0)
    const int ex_lts = (int)ex_l - 1;   // sub-tile-relative position of the last token start before this lane, -1 none
    const u32 tot_a = readlane_u32(inc_a, 63);
    const u32 total_nb = out0 + (tot_a & 0xFFFFu);
    const u32 shift = PASSTHROUGH ? 0 : (out0 & 15u);
    // does a token run into this sub-tile from the left?  Where it starts matters only when it also ends here
    // (the sub-tile with the end writes the token's table entry): the first token end of the sub-tile comes
    // before its first token start.
    const bool runs_in = readlane_u32((u32)(((TOKC & 1u) && prev_tok && !(RS & 1u)) ? 1u : 0u), 0) != 0;
    // token ends of the sub-tile: every token that starts or runs in also ends here, unless it runs on
    const bool runs_out = readlane_u32((u32)(((TOKC >> 31) && next_tok && !next_rs) ? 1u : 0u), 63) != 0;
    const u32 tot_nte = (tot_a >> 16) + (runs_in ? 1u : 0u) - (runs_out ? 1u : 0u);
    bool need_start = false;
    if (runs_in) {   // uniform
        const u64 bte = __ballot(TE != 0), bts = __ballot(TS != 0);
        if (bte) {
            const int la = ctz(bte);
            if (!bts || la < ctz(bts)) need_start = true;
            else if (la == ctz(bts)) need_start = readlane_u32((u32)(__builtin_ctz(TE | 0x80000000u) < __builtin_ctz(TS | 0x80000000u) ? 1u : 0u), la) != 0;
        }
    }
    if (need_start) {   // uniform
        // lane 0: the last record that starts before the sub-tile, as a position in the previous block (or -1)
        const long long rp = (prev_rec + 64 >= tS && prev_rec < tS) ? (long long)(prev_rec + 64 - tS) : -1;
        const u64 zeros = ~prev_w;   // bit 63 of prev_w is set
        const long long z = zeros ? top(zeros) + 1 : -1;
        const long long at = z > rp ? z : rp;   // start position inside the previous block, if it can be told from there
        const int at0 = (int)readlane_u32((u32)(int)at, 0);
        if (at0 >= 0) {
            if (tid == 0) sh.cin_start = tS - 64 + (u64)at0;
        } else {
            const u64 rec0_start = readlane_u32((u32)prev_rec, 0);   // a batch is shorter than 4 GiB
            const u64 start = find_token_start_before(plan.ct, tS, rec0_start);
            if (tid == 0) sh.cin_start = start;
        }
    }
    if (stop_after == 2) return;
    wave_lds_sync();   // the image is zero in every lane's view before anything is ORed into it
    // bytes -> LDS output image, one copy run at a time.  A run is a maximal stretch of copied bytes with no
    // separator inside; its source is contiguous in the lane's half block and its destination is contiguous in
    // the output image.  The image starts out as zeros and a run moves as whole dwords of the image: the dwords
    // it covers completely are stored, the one or two it shares with its neighbours (other runs, separators,
    // other lanes) are ORed in with the bytes outside the run masked off - no byte-sized accesses, no loops over
    // bytes.  The source of image dword k is the unaligned dword at a fixed distance in the half block: two
    // aligned dwords shifted in registers (LDS accesses that straddle a dword stall the whole LDS pipeline).
    if (!PASSTHROUGH) {
        u32* img = reinterpret_cast<u32*>(sh.out);
        const u32* srcw = reinterpret_cast<const u32*>(blk);   // 8 dwords and a spare one; the slot is dword aligned
        u32 o = ex_nb - out0 + shift;
        u32 brk = Cm & (PFX | ~(Cm << 1));
        while (brk) {
            const u32 a0 = (u32)__builtin_ctz(brk);
            brk &= brk - 1;
            const u32 rest = ~Cm >> a0;
            u32 len = rest ? (u32)__builtin_ctz(rest) : 32u - a0;
            if (brk) { const u32 nbk = (u32)__builtin_ctz(brk) - a0; if (nbk < len) len = nbk; }
            if ((PFX >> a0) & 1u) {
                atomicOr(&img[o >> 2], (((PNL >> a0) & 1u) ? (u32)'\n' : (u32)' ') << (8u * (o & 3u)));
                o++;
            }
            const u32 e = o + len - 1u;                    // last image byte of the run
            const u32 d = o & 3u, D0 = o >> 2, nd = (e >> 2) - D0;   // nd + 1 image dwords
            const int s0 = (int)a0 - (int)d;               // source byte of image dword D0's first byte: -3 .. 31
            const u32 sa = (u32)s0 & 3u;
            const int q0 = s0 >> 2;                        // aligned source dword of that byte: -1 .. 7
            const u32 mf = 0xFFFFFFFFu << (8u * d), ml = 0xFFFFFFFFu >> (8u * (3u - (e & 3u)));
            // (source bytes before the half block only ever land in the bytes that mf masks off: dword -1 is not read)
            const u32* sp = srcw + q0;
            u32* dp = img + D0;
            u32 w0 = srcw[q0 < 0 ? 0 : q0], w1 = sp[1];
            const u32 v0 = __builtin_amdgcn_alignbyte(w1, w0, sa) & mf;
            if (nd == 0) {
                atomicOr(dp, v0 & ml);
            } else {
                atomicOr(dp, v0);
                u32 k = 1;   // next image dword; its source dwords are sp[k], sp[k + 1]
                if (nd >= 5) {   // four whole dwords at once
                    const u32 x1 = sp[1], x2 = sp[2], x3 = sp[3], x4 = sp[4], x5 = sp[5];
                    dp[1] = __builtin_amdgcn_alignbyte(x2, x1, sa);
                    dp[2] = __builtin_amdgcn_alignbyte(x3, x2, sa);
                    dp[3] = __builtin_amdgcn_alignbyte(x4, x3, sa);
                    dp[4] = __builtin_amdgcn_alignbyte(x5, x4, sa);
                    k = 5;
                }
                for (; k < nd; k++) dp[k] = __builtin_amdgcn_alignbyte(sp[k + 1], sp[k], sa);   // at most three
                atomicOr(dp + nd, __builtin_amdgcn_alignbyte(sp[nd + 1], sp[nd], sa) & ml);
            }
            o += len;
        }
    }
    wave_lds_sync();   // the input image is dead from here on, the output image and cin_start are complete
    if (stop_after == 3) return;
    // tokens: one (start, len) record per token end.  Tokens that start and end inside the sub-tile are
    // staged in LDS (the input image is no longer needed) and stored with coalesced writes below.
    u32* tokstage = reinterpret_cast<u32*>(sh.tile.bytes);
    {
        // output position of the (copied) byte at bit j: the lane's base plus the bit index, minus the bytes
        // dropped below j, plus the separators inserted up to j
        const u32 base = PASSTHROUGH ? (u32)(p - tS) : ex_nb - out0;   // relative to out0
        // (bit j is a copied byte, so the bits below it are all valid: j minus the dropped bytes below it is the
        // number of copied bytes below it)
        const u32 pfx = PASSTHROUGH ? 0u : PFX;
        auto pos = [&](u32 j) -> u32 {
            if (PASSTHROUGH) return base + j;
            return base + (u32)__builtin_popcount(Cm & ((1u << j) - 1u)) + (u32)__builtin_popcount(pfx & ((2u << j) - 1u));
        };
        u32 ts = TS, te = TE;
        u32 idx = ex_ntk;   // index of the next token that starts in this lane
        if (te) {
            const u32 x0 = (u32)__builtin_ctz(te);
            if (!ts || x0 < (u32)__builtin_ctz(ts)) {
                // the token that runs into this lane from the left ends at x0
                const bool inside = ex_lts >= 0;
                const u64 start_abs = inside ? tS + (u64)ex_lts : sh.cin_start;
                const u32 len = (u32)(p + x0 - start_abs + 1);
                const u32 start_rel = pos(x0) - (len - 1);   // relative to out0 (wraps below it when not inside)
                if (inside && idx - 1 - tok0 < TOK_CAP) tokstage[idx - 1 - tok0] = start_rel | (len << 16);
                else { tok_start[idx - 1] = out0 + start_rel; tok_len[idx - 1] = len; }
                te &= te - 1;
            }
        }
        // from here on starts and ends alternate; a start without an end left belongs to a token that runs on
        const u32 slot0 = idx - tok0;
        if (slot0 + (u32)__builtin_popcount(ts) <= TOK_CAP) {
            u32* slot = tokstage + slot0;
            while (te) {
                const u32 x = (u32)__builtin_ctz(te), s0 = (u32)__builtin_ctz(ts);
                te &= te - 1;
                ts &= ts - 1;
                *slot++ = pos(s0) | ((x - s0 + 1u) << 16);
            }
        } else {
            while (te) {
                const u32 x = (u32)__builtin_ctz(te), s0 = (u32)__builtin_ctz(ts);
                te &= te - 1;
                ts &= ts - 1;
                if (idx - tok0 < TOK_CAP) tokstage[idx - tok0] = pos(s0) | ((x - s0 + 1u) << 16);
                else { tok_start[idx] = out0 + pos(s0); tok_len[idx] = x - s0 + 1u; }
                idx++;
            }
        }
    }
    wave_lds_sync();
    {
        u32 n_ss = tot_nte - ((runs_in && tot_nte > 0) ? 1u : 0u);
        if (n_ss > TOK_CAP) n_ss = TOK_CAP;
        if (stop_after == 6) n_ss = 0;   // profiling aid: everything but the token table stores
        // four tokens per lane and store (the token arrays are only 4-byte aligned at tok0)
        for (u32 i = tid * 4; i < n_ss; i += 64 * 4) {
            if (i + 4 <= n_ss) {
                const uint4 v = *reinterpret_cast<const uint4*>(tokstage + i);
                const Quad32 st = {out0 + (v.x & 0xFFFFu), out0 + (v.y & 0xFFFFu), out0 + (v.z & 0xFFFFu),
                                   out0 + (v.w & 0xFFFFu)};
                const Quad32 ln = {v.x >> 16, v.y >> 16, v.z >> 16, v.w >> 16};
                *reinterpret_cast<Quad32*>(tok_start + tok0 + i) = st;
                *reinterpret_cast<Quad32*>(tok_len + tok0 + i) = ln;
            } else {
                for (u32 k = i; k < n_ss; k++) {
                    const u32 v = tokstage[k];
                    tok_start[tok0 + k] = out0 + (v & 0xFFFFu);
                    tok_len[tok0 + k] = v >> 16;
                }
            }
        }
    }
    if (stop_after == 4) return;
    // per-record offsets
    {
        // lane i fetched the start of record rec0 + i with the sub-tile; a crowded sub-tile reads the rest.
        // The record's half block is another lane's: its words come through the cross-lane network (every lane
        // takes part in every round).
        for (u32 k0 = 0; k0 < rec_n; k0 += 64) {   // uniform
            const u32 k = k0 + (u32)tid;
            const u64 r = (u64)rec0 + k;
            const bool have = k < rec_n;
            u64 q = k0 == 0 ? (u64)in.rf.q : 0;
            if (k0 != 0 && have) q = offsets[r];
            const bool live = have && q < N && q >= tS;
            const u32 l = live ? (u32)((q - tS) >> 5) & 63u : 0u, j = (u32)((q - tS) & 31);
            const u32 lowm = (1u << j) - 1u;
            const u32 eb = (u32)__shfl((int)ex_nb, (int)l, 64), tb = (u32)__shfl((int)ex_ntk, (int)l, 64);
            const u32 cw = (u32)__shfl((int)Cm, (int)l, 64), pxm = (u32)__shfl((int)PFX, (int)l, 64),
                      tsm = (u32)__shfl((int)TS, (int)l, 64);
            if (live) {
                den_off[r] = PASSTHROUGH ? q : (u64)(eb + (u32)__builtin_popcount(cw & lowm) + (u32)__builtin_popcount(pxm & lowm));
                tok_off[r] = (u64)(tb + (u32)__builtin_popcount(tsm & lowm));
            }
        }
    }
    if (stop_after == 5) return;
    // LDS output image -> global, 16-byte stores where the destination is aligned
    if (!PASSTHROUGH) {
        const u32 n = total_nb - out0;   // bytes this sub-tile emits
        const u64 g0 = (u64)out0 - shift;   // 16-aligned
        const u32 end = shift + n;
        for (u32 c16 = tid; c16 * 16 < end; c16 += 64) {
            const u32 lo = c16 * 16, hi16 = lo + 16;
            if (lo >= shift && hi16 <= end)
                *reinterpret_cast<uint4*>(denoised + g0 + lo) = *reinterpret_cast<const uint4*>(sh.out + lo);
        }
        // the partial chunks at the two ends, one byte per lane: lanes 0-15 the first chunk, lanes 16-31 the last
        const u32 cl = (end - 1) >> 4;   // end > 0 whenever a lane gets here with something to do
        if (tid < 16) {
            const u32 i = (u32)tid;
            if ((shift > 0 || end < 16) && i >= shift && i < end) denoised[g0 + i] = sh.out[i];
        } else if (tid < 32) {
            const u32 i = cl * 16 + (u32)tid - 16;
            if (end > 0 && cl > 0 && (end & 15u) != 0 && i < end) denoised[g0 + i] = sh.out[i];
        }
    }
}
template <bool PASSTHROUGH, int WAVES>
__device__ __forceinline__ void emit_body(EmitShared* shw, u32 vb, const u8* __restrict__ bytes,
                                          const u64* __restrict__ offsets, const u32* __restrict__ rec_lo_w, u64 R, u64 N,
                                          u64 nB, u64 nW, const PlanMem& plan, const u64* __restrict__ tile_base,
                                          u8* __restrict__ denoised, u32* __restrict__ tok_start, u32* __restrict__ tok_len,
                                          u64* __restrict__ den_off, u64* __restrict__ tok_off, u32 stop_after) {
    emit_sub<PASSTHROUGH>(shw[threadIdx.x >> 6], (u64)vb * WAVES + (threadIdx.x >> 6), bytes, offsets, rec_lo_w, R, N, nB, nW,
                          plan, tile_base, denoised, tok_start, tok_len, den_off, tok_off, stop_after);
}
template <bool PASSTHROUGH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(8))) void k_emit(const u8* __restrict__ bytes,
                                              const u64* __restrict__ offsets, const u32* __restrict__ rec_lo_w, u64 R,
                                              u64 N, u64 nB, u64 nW, PlanMem plan,
                                              const u64* __restrict__ tile_base, u8* __restrict__ denoised,
                                              u32* __restrict__ tok_start, u32* __restrict__ tok_len,
                                              u64* __restrict__ den_off, u64* __restrict__ tok_off, u32 stop_after) {
    __shared__ EmitShared shw[WAVES];
    emit_body<PASSTHROUGH, WAVES>(shw, blockIdx.x, bytes, offsets, rec_lo_w, R, N, nB, nW, plan, tile_base, denoised, tok_start,
                                  tok_len, den_off, tok_off, stop_after);
}

// One lane per record (wavefronts stride over the records): a record's first CHUNK_INLINE descriptors are
// written by its lane, the rest of a record with many windows by the whole wavefront.
constexpr u32 CHUNK_INLINE = 4;
// The last kernel of a batch: its first thread also copies the batch's totals and flags (totals[0..9]) into the
// page-locked host words `host_totals`, which spares a copy command behind the kernels.
__device__ __forceinline__ void chunk_emit_body(u32 vb, u32 vgrid, const u64* __restrict__ chunk_off,
                                                const u64* __restrict__ tok_off, u64 R, const DevCfg& cfg, u64 capacity,
                                                ChunkDesc* __restrict__ chunks, u64* __restrict__ totals,
                                                u64* __restrict__ host_totals) {
    const u64 C = chunk_off[R];
    if (vb == 0 && threadIdx.x == 0) {
        totals[2] = C;
        totals[3] = C > capacity ? 1 : 0;
        if (host_totals) {
            for (int i = 0; i < 10; i++) host_totals[i] = i == 2 ? C : i == 3 ? (u64)(C > capacity ? 1 : 0) : totals[i];
            __threadfence_system();
        }
    }
    if (C > capacity) return;
    const int lane = threadIdx.x & 63;
    const u64 wave = ((u64)vb * blockDim.x + threadIdx.x) >> 6, nwaves = ((u64)vgrid * blockDim.x) >> 6;
    for (u64 rb = wave * 64; rb < R; rb += nwaves * 64) {   // uniform
        const u64 r = rb + (u64)lane;
        u64 n = 0, T = 0, base = 0;
        if (r < R) {
            base = chunk_off[r];
            n = chunk_off[r + 1] - base;
            T = tok_off[r + 1] - tok_off[r];
        }
        for (u64 i = 0; i < n && i < CHUNK_INLINE; i++) chunks[base + i] = chunk_at(T, i, (u32)r, cfg);
        u64 big = __ballot(n > CHUNK_INLINE);
        while (big) {   // uniform
            const int L = ctz(big);
            big &= big - 1;
            const u64 nn = __shfl(n, L, 64), TT = __shfl(T, L, 64), bb = __shfl(base, L, 64);
            for (u64 i = CHUNK_INLINE + (u64)lane; i < nn; i += 64) chunks[bb + i] = chunk_at(TT, i, (u32)(rb + (u64)L), cfg);
        }
    }
}
__global__ void k_chunk_emit(const u64* __restrict__ chunk_off, const u64* __restrict__ tok_off, u64 R, DevCfg cfg,
                             u64 capacity, ChunkDesc* __restrict__ chunks, u64* __restrict__ totals,
                             u64* __restrict__ host_totals) {
    chunk_emit_body(blockIdx.x, gridDim.x, chunk_off, tok_off, R, cfg, capacity, chunks, totals, host_totals);
}

// The chunk-count scan and the descriptors in one launch, for batches of few records and few descriptors (a
// dependent launch costs more than this work).
constexpr u64 CHUNKS_SMALL_MAX_DESC = 65536;
__global__ __launch_bounds__(NT) void k_chunks_small(GenChunkCounts gen, u64 R, u64* __restrict__ chunk_off, DevCfg cfg,
                                                      u64 capacity, ChunkDesc* __restrict__ chunks,
                                                      u64* __restrict__ host_totals) {
    __shared__ u64 smem[NT / 64];
    scan_small_body(smem, gen, R, chunk_off);
    __syncthreads();
    chunk_emit_body(0u, 1u, chunk_off, gen.tok_off, R, cfg, capacity, chunks, gen.totals, host_totals);
}

// n bytes from a 16-byte aligned device array to a 16-byte aligned place in host memory, by the whole workgroup.
__device__ inline void pack_copy(u8* __restrict__ dst, const void* __restrict__ src_, u64 n) {
    const u8* src = reinterpret_cast<const u8*>(src_);
    const u64 q = n >> 4;
    for (u64 i = threadIdx.x; i < q; i += NT) reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (u64 i = (q << 4) + threadIdx.x; i < n; i += NT) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------------
// chunk text (optional): tokens joined by one space, or the record text for a single window
// ---------------------------------------------------------------------------------------------------
// A chunk is rendered as the record's text verbatim when it is the record's only window
// (sliding_window.rs:43-58).  chunk_off == nullptr: "head" descriptors (k_head_desc), verbatim whenever
// they cover all tokens of the record.
__device__ inline bool chunk_is_whole_record(const ChunkDesc& c, const u64* chunk_off, const u64* tok_off) {
    u64 r = c.record;
    const bool all_tokens = (u64)(c.tok_end - c.tok_begin) == tok_off[r + 1] - tok_off[r];
    if (!chunk_off) return all_tokens;
    return c.kind == 0 && chunk_off[r + 1] - chunk_off[r] == 1 && all_tokens;
}

// One descriptor per record: its first min(T, k) tokens.
__global__ void k_head_desc(const u64* __restrict__ tok_off, u64 R, u32 k, ChunkDesc* __restrict__ out) {
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const u64 T = tok_off[r + 1] - tok_off[r];
    ChunkDesc d = {(u32)r, 0u, 0u, 0u, 0u, 0u, (u32)(T < (u64)k ? T : (u64)k), 0u};
    out[r] = d;
}

// one wave per chunk: chunk i by the calling wavefront
__device__ inline void chunk_text_len_body(u64 i, const ChunkDesc* __restrict__ chunks, const u64* __restrict__ chunk_off,
                                           const u64* __restrict__ tok_off, const u64* __restrict__ den_off,
                                           const u32* __restrict__ tok_len, u64* __restrict__ lens) {
    const int lane = threadIdx.x & 63;
    ChunkDesc c = chunks[i];
    u64 r = c.record;
    if (chunk_is_whole_record(c, chunk_off, tok_off)) {
        if (lane == 0) lens[i] = den_off[r + 1] - den_off[r];
        return;
    }
    u64 t0 = tok_off[r];
    u64 s = 0;
    for (u32 k = c.tok_begin + lane; k < c.tok_end; k += 64) s += tok_len[t0 + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) lens[i] = c.tok_end > c.tok_begin ? s + (c.tok_end - c.tok_begin - 1) : 0;
}
__global__ __launch_bounds__(NT) void k_chunk_text_len(const ChunkDesc* __restrict__ chunks, u64 C,
                                                        const u64* __restrict__ chunk_off, const u64* __restrict__ tok_off,
                                                        const u64* __restrict__ den_off, const u32* __restrict__ tok_len,
                                                        u64* __restrict__ lens) {
    const u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i < C) chunk_text_len_body(i, chunks, chunk_off, tok_off, den_off, tok_len, lens);
}

__device__ inline void chunk_text_emit_body(u64 i, const ChunkDesc* __restrict__ chunks, const u64* __restrict__ chunk_off,
                                            const u64* __restrict__ tok_off, const u64* __restrict__ den_off,
                                            const u32* __restrict__ tok_start, const u32* __restrict__ tok_len,
                                            const u8* __restrict__ denoised, const u64* __restrict__ text_off,
                                            u8* __restrict__ text) {
    const int lane = threadIdx.x & 63;
    ChunkDesc c = chunks[i];
    u64 r = c.record;
    u8* dst = text + text_off[i];
    if (chunk_is_whole_record(c, chunk_off, tok_off)) {
        const u8* src = denoised + den_off[r];
        u64 n = den_off[r + 1] - den_off[r];
        for (u64 k = lane; k < n; k += 64) dst[k] = src[k];
        return;
    }
    u64 t0 = tok_off[r];
    u64 base = 0;
    for (u32 k0 = c.tok_begin; k0 < c.tok_end; k0 += 64) {
        u32 k = k0 + lane;
        bool have = k < c.tok_end;
        u64 len = have ? tok_len[t0 + k] : 0;
        u64 span = have ? len + (k + 1 < c.tok_end ? 1 : 0) : 0;   // token plus its trailing space
        u64 inc = span;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            u64 o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (have) {
            u8* d = dst + base + inc - span;
            const u8* s = denoised + tok_start[t0 + k];
            for (u64 q = 0; q < len; q++) d[q] = s[q];
            if (span > len) d[len] = ' ';
        }
        base += __shfl(inc, 63, 64);
    }
}
__global__ __launch_bounds__(NT) void k_chunk_text_emit(const ChunkDesc* __restrict__ chunks, u64 C,
                                                         const u64* __restrict__ chunk_off, const u64* __restrict__ tok_off,
                                                         const u64* __restrict__ den_off, const u32* __restrict__ tok_start,
                                                         const u32* __restrict__ tok_len, const u8* __restrict__ denoised,
                                                         const u64* __restrict__ text_off, u8* __restrict__ text) {
    const u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i < C) chunk_text_emit_body(i, chunks, chunk_off, tok_off, den_off, tok_start, tok_len, denoised, text_off, text);
}
// The texts of a few descriptors in ONE launch (lengths, their scan, the texts: otherwise five launches and a
// synchronise between them to size the text buffer).  One workgroup, a wavefront per descriptor at a time; the text
// goes into a buffer of text_cap bytes, and the kernel tells the host the size and whether it fitted.
constexpr u64 RENDER_SMALL_MAX = 256;
struct GenLoadU64 {
    const u64* v;
    struct State {};
    __device__ State begin() const { return State(); }
    __device__ u64 at(const State&, u64 i) const { return v[i]; }
    __device__ void once() const {}
};
__global__ __launch_bounds__(NT) void k_render_text_small(const ChunkDesc* __restrict__ chunks, u64 C, const u64* chunk_off,
                                                           const u64* tok_off, const u64* den_off, const u32* tok_start,
                                                           const u32* tok_len, const u8* denoised, u64* lens, u64* text_off,
                                                           u8* text, u64 text_cap, u64* host_totals, u8* host_pack,
                                                           u64 host_cap) {
    __shared__ u64 smem[NT / 64];
    const u64 wave = threadIdx.x >> 6;
    for (u64 i = wave; i < C; i += NT / 64) chunk_text_len_body(i, chunks, chunk_off, tok_off, den_off, tok_len, lens);
    __syncthreads();
    const GenLoadU64 gen = {lens};
    scan_small_body(smem, gen, C, text_off);
    __syncthreads();
    const u64 total = text_off[C];
    const bool fits = total <= text_cap;
    if (fits)
        for (u64 i = wave; i < C; i += NT / 64)
            chunk_text_emit_body(i, chunks, chunk_off, tok_off, den_off, tok_start, tok_len, denoised, text_off, text);
    // for a caller in host memory: offsets and text into its page-locked block as well, when they fit
    const u64 at_text = ((C + 1) * sizeof(u64) + 15) & ~(u64)15;
    const bool packed = host_pack && fits && at_text + total <= host_cap;
    if (packed) {
        __syncthreads();
        pack_copy(host_pack, text_off, (C + 1) * sizeof(u64));
        pack_copy(host_pack + at_text, text, total);
    }
    if (threadIdx.x == 0) {
        host_totals[4] = total;
        host_totals[14] = packed ? 3 : fits ? 1 : 2;
        __threadfence_system();
    }
}

// ---------------------------------------------------------------------------------------------------
// k_small_batch: the whole path in one launch, for batches of a few tiles
// ---------------------------------------------------------------------------------------------------
// A batch of a few kilobytes is bound by the number of dependent launches (eight of them, each a few
// microseconds, for microseconds of work).  Up to SMALL_MAX_TILES tiles and SMALL_MAX_RECORDS records are
// therefore handled by ONE workgroup that runs the bodies of the kernels above one after the other, with a
// workgroup barrier where the full path has a kernel boundary.  Same code, same arrays, same results; when the
// tile seeds cannot be settled from the windows next to the tiles the batch says so in its totals exactly as the
// full path does, and the host runs it again there.
constexpr u64 SMALL_MAX_TILES = 2;        // 32 KiB of text: one workgroup takes ~13 us a tile, the kernel sequence ~50 us in all
// (A batch from host memory has the same limit: beyond two tiles the kernel sequence with its short result arrays
// packed by one launch, k_pack_small, is as fast - 88 us against 121 us for 64 KiB.)
constexpr u64 SMALL_HOST_MAX_TILES = SMALL_MAX_TILES;
constexpr u64 SMALL_MAX_RECORDS = 4096;
struct SmallArgs {
    const u8* bytes;
    const u64* offsets;
    u64 R, N;
    DevCfg cfg;
    RatioArg ratio;
    u32 force_open;
    // working arrays (see PathBufs)
    u32 *rec_lo, *rec_lo_w;
    J* seedJ;
    S* seedS;
    void* plan;
    u64* g3list;
    u64 g3cap;
    u32 *tile_bytes, *tile_toks, *tile_flags;
    u64 *packed, *tile_base, *counts, *totals;
    // results
    u8* denoised;
    u64 *den_off, *tok_off, *chunk_off;
    u8* dropped;
    u32 *tok_start, *tok_len;
    ChunkDesc* chunks;
    u64 chunk_cap;
    u64* host_totals;
    u64 seq;
    // A batch that comes from host memory (null otherwise): its text and offsets lie in page-locked host memory and
    // are first copied to `bytes` / `offsets` (stage_bytes / stage_offsets are the same arrays, writable); at the
    // end the results are gathered into one page-locked block (pack_layout) when they fit.
    const u8* src_bytes;
    const u64* src_offsets;
    u8* stage_bytes;
    u64* stage_offsets;
    u8* host_out;
    u64 host_out_cap;
    u32 outputs;
    // with TRP_OUT_CHUNK_TEXT among the outputs: the chunk texts as well (lengths, their offsets, the text buffer and
    // its capacity), rendered behind the descriptors when there are at most RENDER_SMALL_MAX of them
    u64 *text_len, *text_off;
    u8* text;
    u64 text_cap;
};
// Where the result arrays of a small host batch lie in the page-locked block, every array on a 16-byte boundary.
struct PackLayout {
    u64 den_off, tok_off, chunk_off, chunks, tok_start, tok_len, denoised, dropped, total;
};
__host__ __device__ inline PackLayout pack_layout(u64 R, u64 n_den, u64 n_tok, u64 n_chunks, u32 outputs) {
    PackLayout L;
    u64 p = 0;
    auto take = [&p](u64 bytes) { const u64 at = p; p += (bytes + 15) & ~(u64)15; return at; };
    L.den_off = take((R + 1) * sizeof(u64));
    L.tok_off = take((R + 1) * sizeof(u64));
    L.chunk_off = take((R + 1) * sizeof(u64));
    L.chunks = take((outputs & TRP_OUT_CHUNKS) ? n_chunks * sizeof(ChunkDesc) : 0);
    L.tok_start = take((outputs & TRP_OUT_TOKENS) ? n_tok * sizeof(u32) : 0);
    L.tok_len = take((outputs & TRP_OUT_TOKENS) ? n_tok * sizeof(u32) : 0);
    L.denoised = take((outputs & TRP_OUT_TEXT) ? n_den : 0);
    L.dropped = take(R);
    L.total = p;
    return L;
}
constexpr u64 SMALL_OUT_CAP = (u64)1 << 20;   // page-locked block for the results of a small host batch ...
constexpr u64 H_OUT_BYTES = 2 * SMALL_OUT_CAP;  // ... and as much again behind it for rendered texts (k_render_text_small)
// Several short device arrays into one page-locked host block with one launch: the per-record arrays of a host
// batch are a few kilobytes each, and a copy command to pageable memory costs 20 us apiece.
constexpr int PACK_MAX = 12;
struct PackArgs {
    const void* src[PACK_MAX];
    u64 bytes[PACK_MAX];
    u64 at[PACK_MAX];     // 16-byte aligned offsets into dst
    int n;
    u8* dst;
};
__global__ __launch_bounds__(NT) void k_pack_small(PackArgs a) {
    for (int i = blockIdx.x; i < a.n; i += gridDim.x) pack_copy(a.dst + a.at[i], a.src[i], a.bytes[i]);
}
constexpr u64 SMALL_IN_CAP = (u64)256 << 10;   // page-locked block for a small host batch, or for the offsets of a larger one
static_assert(SMALL_IN_CAP >= SMALL_HOST_MAX_TILES * TILE + 64 + (SMALL_MAX_RECORDS + 1) * sizeof(u64), "small batch must fit");
union SmallShared {
    SeedShared seed;
    PlanShared plan;
    CountShared count;
    EmitShared emit[EMIT_WAVES];
    G3SolveShared g3[NT / 64];
    u64 scan[NT / 64];
};
static_assert(EMIT_WAVES == NT / 64, "the emit body takes one sub-tile per wavefront of the workgroup");

template <int MODE>   // 0: denoiser, 1: normalize_inline_whitespace, 2: denoiser off (pass-through)
__global__ __launch_bounds__(NT) void k_small_batch(SmallArgs a) {
    __shared__ SmallShared sh;
    __shared__ UniShared uni;
    const u64 R = a.R, N = a.N;
    const u64 nB = (N + BLOCK - 1) / BLOCK, nT = (N + TILE - 1) / TILE, nW = (N + SUB - 1) / SUB;
    const PlanMem plan = plan_view(a.plan, nB);
    const int wave = threadIdx.x >> 6;
    if (a.src_offsets) {   // uniform: the batch comes from page-locked host memory (the block has slack behind the text)
        for (u64 i = threadIdx.x; i * 16 < N; i += NT)
            reinterpret_cast<uint4*>(a.stage_bytes)[i] = reinterpret_cast<const uint4*>(a.src_bytes)[i];
        for (u64 i = threadIdx.x; i <= R; i += NT) a.stage_offsets[i] = a.src_offsets[i];
        __syncthreads();
    }
    for (u32 vb = 0; (u64)vb * NT <= R; vb++)
        rec_lo_body(vb, a.offsets, R, N, nT, nW, a.rec_lo, a.rec_lo_w, a.totals, a.seq);
    stage_tables(uni);
    __syncthreads();
    if (nT) {   // uniform
        if (MODE != 2) {
            u32* const open_count = reinterpret_cast<u32*>(a.totals + 8);   // zeroed with the totals
            for (u32 vb = 0; (u64)vb * SEED_TPW * (NT / 64) < nT; vb++)   // (a wavefront shares nothing with the others)
                seed_body<MODE == 1>(sh.seed, uni, vb, a.bytes, a.offsets, a.rec_lo, R, N, nB, nT, a.force_open, a.seedJ,
                                     a.seedS, open_count);
            __syncthreads();
            for (u32 vb = 0; vb < nT; vb++) {
                plan_body<MODE == 1>(sh.plan, uni, vb, a.bytes, a.offsets, a.rec_lo, R, N, nB, a.seedJ, a.seedS, a.cfg, a.ratio,
                                     plan, a.g3list, a.g3cap, a.totals + 6, a.tile_bytes, a.tile_toks, a.tile_flags, 0u);
                __syncthreads();
            }
            gate3_solve_body(sh.g3[wave], (u32)wave, NT / 64, a.bytes, a.offsets, a.rec_lo, R, N, nB, a.cfg, plan, a.g3list,
                             a.totals + 6, a.g3cap, a.tile_bytes, a.tile_toks, a.totals, a.denoised);
        } else {
            for (u32 vb = 0; vb < nT; vb++) {
                count_tokens_body(sh.count, vb, a.bytes, a.offsets, a.rec_lo, R, N, nB, plan, a.tile_bytes, a.tile_toks,
                                  a.tile_flags);
                __syncthreads();
            }
        }
        __syncthreads();
        {
            const GenTiles gen = {a.tile_bytes, a.tile_toks, a.tile_flags, a.packed};
            scan_small_body(sh.scan, gen, nW, a.tile_base);
        }
        __syncthreads();
        for (u32 vb = 0; (u64)vb * EMIT_WAVES < nW; vb++)   // (wavefronts are independent of each other here too)
            emit_body<MODE == 2, EMIT_WAVES>(sh.emit, vb, a.bytes, a.offsets, a.rec_lo_w, R, N, nB, nW, plan, a.tile_base,
                                             MODE == 2 ? (u8*)nullptr : a.denoised, a.tok_start, a.tok_len, a.den_off,
                                             a.tok_off, 0u);
        __syncthreads();
    }
    {
        GenChunkCounts gen;
        gen.tile_base = a.tile_base; gen.nW = nW; gen.rec_lo_w = a.rec_lo_w; gen.R = R; gen.N = N;
        gen.passthrough = MODE == 2 ? 1 : 0; gen.cfg = a.cfg;
        gen.den_off = a.den_off; gen.tok_off = a.tok_off; gen.totals = a.totals; gen.counts = a.counts; gen.dropped = a.dropped;
        scan_small_body(sh.scan, gen, R, a.chunk_off);
    }
    __syncthreads();
    chunk_emit_body(0u, 1u, a.chunk_off, a.tok_off, R, a.cfg, a.chunk_cap, a.chunks, a.totals, a.host_totals);
    if (a.host_out) {   // uniform
        __syncthreads();
        // nothing to gather from a batch that the host will run again (open tiles, a list or the descriptor array too
        // small) or reject (offsets)
        const u64 n_den = a.totals[0], n_tok = a.totals[1], C = a.chunk_off[R];
        const bool again = a.totals[5] != 0 || (u32)a.totals[8] != 0 || C > a.chunk_cap || a.totals[9] == a.seq;
        const PackLayout L = pack_layout(R, n_den, n_tok, C, a.outputs);
        if (!again && L.total <= a.host_out_cap) {
            u8* const o = a.host_out;
            pack_copy(o + L.den_off, a.den_off, (R + 1) * sizeof(u64));
            pack_copy(o + L.tok_off, a.tok_off, (R + 1) * sizeof(u64));
            pack_copy(o + L.chunk_off, a.chunk_off, (R + 1) * sizeof(u64));
            if (a.outputs & TRP_OUT_CHUNKS) pack_copy(o + L.chunks, a.chunks, C * sizeof(ChunkDesc));
            if (a.outputs & TRP_OUT_TOKENS) {
                pack_copy(o + L.tok_start, a.tok_start, n_tok * sizeof(u32));
                pack_copy(o + L.tok_len, a.tok_len, n_tok * sizeof(u32));
            }
            if (a.outputs & TRP_OUT_TEXT) pack_copy(o + L.denoised, MODE == 2 ? a.bytes : a.denoised, n_den);
            pack_copy(o + L.dropped, a.dropped, R);
            if (threadIdx.x == 0) a.host_totals[10] = a.seq;   // "gathered": read by the host once the kernel has ended
            if ((a.outputs & TRP_OUT_CHUNK_TEXT) && C <= RENDER_SMALL_MAX) {   // uniform
                // the chunk texts behind the arrays (k_render_text_small's steps)
                const u8* const den = MODE == 2 ? a.bytes : a.denoised;
                __syncthreads();
                for (u64 i = (u64)wave; i < C; i += NT / 64)
                    chunk_text_len_body(i, a.chunks, a.chunk_off, a.tok_off, a.den_off, a.tok_len, a.text_len);
                __syncthreads();
                const GenLoadU64 gen = {a.text_len};
                scan_small_body(sh.scan, gen, C, a.text_off);
                __syncthreads();
                const u64 total = a.text_off[C];
                const u64 at_text = ((C + 1) * sizeof(u64) + 15) & ~(u64)15;
                if (total <= a.text_cap && L.total + at_text + total <= a.host_out_cap) {
                    for (u64 i = (u64)wave; i < C; i += NT / 64)
                        chunk_text_emit_body(i, a.chunks, a.chunk_off, a.tok_off, a.den_off, a.tok_start, a.tok_len, den, a.text_off,
                                             a.text);
                    __syncthreads();
                    pack_copy(o + L.total, a.text_off, (C + 1) * sizeof(u64));
                    pack_copy(o + L.total + at_text, a.text, total);
                    if (threadIdx.x == 0) { a.host_totals[12] = a.seq; a.host_totals[13] = total; }
                }
            }
        }
    }
}

// Body of a decorated chunk (src/sampler/mod.rs:1484-1506): descriptor i is chunk index[i] of the batch cut to
// its first budget[i] tokens (unchanged when the budget covers it, which keeps the verbatim rendering of a
// record's only window).
__global__ void k_body_desc(const ChunkDesc* __restrict__ chunks, u64 C, const u32* __restrict__ index,
                            const u32* __restrict__ budget, u64 n, ChunkDesc* __restrict__ out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ChunkDesc z = {0, 0, 0, 0, 0, 0, 0, 0};
    ChunkDesc d = index[i] < C ? chunks[index[i]] : z;
    const u32 have = d.tok_end - d.tok_begin;
    if (budget[i] < have) d.tok_end = d.tok_begin + budget[i];
    out[i] = d;
}

// record_bm25_text (src/sampler/backends/bm25_backend.rs:503-532) over the records of the batch: document i is
// the concatenation of records [doc_off[i], doc_off[i+1]) (headings and section texts in order).  A record
// without a token is blank (`trim().is_empty()`) and left out; the others are followed by the newline.  With
// more than max_tokens tokens (max_tokens > 0) the document is its first max_tokens tokens joined by one
// space; the tokens of a document are consecutive in the token table.  A document without any token has
// length 0 (the caller substitutes the record id).  One wavefront per document.
__global__ __launch_bounds__(NT) void k_doc_text(const u64* __restrict__ doc_off, u64 D, u32 max_tokens, u32 nl_len,
                                                  u32 nl_bytes, const u64* __restrict__ tok_off,
                                                  const u64* __restrict__ den_off, const u32* __restrict__ tok_start,
                                                  const u32* __restrict__ tok_len, const u8* __restrict__ denoised,
                                                  u64* __restrict__ lens, const u64* __restrict__ text_off,
                                                  u8* __restrict__ text) {
    const int lane = threadIdx.x & 63;
    const u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= D) return;
    const u64 r0 = doc_off[i], r1 = doc_off[i + 1];
    const u64 t0 = tok_off[r0], T = tok_off[r1] - t0;
    u8* dst = text ? text + text_off[i] : nullptr;   // first call: lengths only
    if (T == 0) {
        if (!text && lane == 0) lens[i] = 0;
        return;
    }
    if (max_tokens == 0 || T <= (u64)max_tokens) {
        u64 at = 0;
        for (u64 r = r0; r < r1; r++) {   // uniform
            if (tok_off[r + 1] == tok_off[r]) continue;
            const u64 n = den_off[r + 1] - den_off[r];
            if (dst) {
                const u8* src = denoised + den_off[r];
                for (u64 k = lane; k < n; k += 64) dst[at + k] = src[k];
                if (lane < (int)nl_len) dst[at + n + lane] = (u8)(nl_bytes >> (8 * lane));
            }
            at += n + nl_len;
        }
        if (!text && lane == 0) lens[i] = at;
        return;
    }
    u64 base = 0;
    for (u32 k0 = 0; k0 < max_tokens; k0 += 64) {
        const u32 k = k0 + (u32)lane;
        const bool have = k < max_tokens;
        const u64 len = have ? tok_len[t0 + k] : 0;
        const u64 span = have ? len + (k + 1 < max_tokens ? 1 : 0) : 0;   // token plus its trailing space
        u64 inc = span;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u64 o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (have && dst) {
            u8* d = dst + base + inc - span;
            const u8* sp = denoised + tok_start[t0 + k];
            for (u64 q = 0; q < len; q++) d[q] = sp[q];
            if (span > len) d[len] = ' ';
        }
        base += __shfl(inc, 63, 64);
    }
    if (!text && lane == 0) lens[i] = base;
}

}  // namespace

// =====================================================================================================
// host side: context, buffers, C ABI
// =====================================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
            return fail(TRP_E_HIP, std::string(#x) + ": " + hipGetErrorString(e_));            \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return TRP_OK;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return fail(TRP_E_HIP, "hipFree failed"); }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(TRP_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
        cap = want;
        return TRP_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- fixed scratch memory of the entry point that allocates nothing (trp_preprocess_batch_device_noalloc) ----
// Every GPU the library runs on holds one copy of this array from the moment the code object is loaded until
// the process ends.  Host code never uses the array's name as an address (on the host it is only a handle for
// hipGetSymbolAddress): scratch_of_device() asks the runtime once per device, with that device current, where
// the device's copy lives, and everything else (kernel arguments, memsets) goes through that pointer.
constexpr u64 NOALLOC_MAX_BYTES = TRP_NOALLOC_MAX_BYTES, NOALLOC_MAX_RECORDS = TRP_NOALLOC_MAX_RECORDS;
constexpr u64 NOALLOC_G3_CAP = NOALLOC_MAX_BYTES / 64;   // one listed digit-heavy line per 64 bytes of text

struct ScratchLayout {
    size_t totals, rec_lo, rec_lo_w, SA, seedJ, seedS, far_list, tile_bytes, tile_toks, tile_flags, packed, tile_base, partial,
        counts, g3list, plan, end;
};
constexpr size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr ScratchLayout scratch_layout() {
    constexpr size_t nT = (NOALLOC_MAX_BYTES + TILE - 1) / TILE, nB = (NOALLOC_MAX_BYTES + BLOCK - 1) / BLOCK;
    constexpr size_t nW = (NOALLOC_MAX_BYTES + SUB - 1) / SUB;
    constexpr size_t R = NOALLOC_MAX_RECORDS;
    constexpr size_t nPart = ((nW > R ? nW : R) + SCAN_CHUNK - 1) / SCAN_CHUNK + 1;
    ScratchLayout L = {};
    size_t at = 0;
    L.totals = at;     at = align256(at + 16 * sizeof(u64));
    L.rec_lo = at;     at = align256(at + (nT + 2) * sizeof(u32));
    L.rec_lo_w = at;   at = align256(at + (nW + 2) * sizeof(u32));
    L.SA = at;         at = align256(at + (nT + 1) * sizeof(J));
    L.seedJ = at;      at = align256(at + (nT + 1) * sizeof(J));
    L.seedS = at;      at = align256(at + (nT + 1) * sizeof(S));
    L.far_list = at;   at = align256(at + (nT + 1) * sizeof(u32));
    L.tile_bytes = at; at = align256(at + (nW + 1) * sizeof(u32));
    L.tile_toks = at;  at = align256(at + (nW + 1) * sizeof(u32));
    L.tile_flags = at; at = align256(at + (nW + 1) * sizeof(u32));
    L.packed = at;     at = align256(at + (nW + 1) * sizeof(u64));
    L.tile_base = at;  at = align256(at + (nW + 2) * sizeof(u64));
    L.partial = at;    at = align256(at + nPart * sizeof(u64));
    L.counts = at;     at = align256(at + (R + 1) * sizeof(u64));
    L.g3list = at;     at = align256(at + NOALLOC_G3_CAP * sizeof(u64));
    L.plan = at;       at = align256(at + plan_bytes(nB));
    L.end = at;
    return L;
}
constexpr size_t SCRATCH_BYTES = scratch_layout().end;   // about 25 MiB
static_assert(SCRATCH_BYTES <= ((size_t)32 << 20), "the fixed scratch array is meant to stay modest");
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

constexpr int SCRATCH_MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static unsigned char* g_scratch_ptr[SCRATCH_MAX_DEVICES];   // device address of `scratch` on each GPU, once known

// Device address of this GPU's copy of `scratch`.  `device` must be the current device.
static int scratch_of_device(int device, unsigned char** out) {
    if (device < 0 || device >= SCRATCH_MAX_DEVICES) return fail(TRP_E_ARG, "device ordinal out of range");
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        unsigned char* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(TRP_E_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return TRP_OK;
}

// Numbers the batches of the whole process (PathBufs::seq): contexts on one GPU share the fixed scratch array,
// so a number must never come back on another context.
static std::atomic<u64> g_batch_seq{0};

struct trp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;            // false: the caller's stream (trp_create_on_stream)
    DevBuf rec_lo, rec_lo_w, SA, seedJ, seedS, plan, g3list, tile_bytes, tile_toks, tile_flags, packed, tile_base, partial;
    DevBuf denoised, den_off, dropped, tok_start, tok_len, tok_off, counts, chunk_off, chunks, totals;
    DevBuf in_bytes, in_offsets;        // staging of the host-memory entry point
    DevBuf fold_bytes, fold_offsets;    // text between two stages of a preprocessor fold (run_device_fold)
    DevBuf text, text_off, text_len;    // chunk text
    DevBuf far_list;                    // tiles whose carries need a long walk (k_tile_carry)
    DevBuf heads;                       // head-of-record / chunk-body descriptors (trp_head_tokens_text_device, trp_chunk_bodies_device)
    DevBuf sel;                         // caller's selection arrays (chunk indices + budgets, document offsets)
    u64* h_totals = nullptr;            // pinned
    // Small batches from host memory: text and offsets go into the page-locked block h_in, from where the single
    // launch fetches them itself, and the results come back gathered in the page-locked block h_out (both made at
    // the first such batch).  small_io is set around the device call of such a batch only.
    u8* h_in = nullptr;
    u8* h_out = nullptr;
    // Host entry points that want rendered texts in host memory set want_host_pack; the single-launch renderer then
    // also leaves the offsets and the text in the upper half of h_out and says so in text_host_packed.
    bool want_host_pack = false;
    bool text_host_packed = false;
    struct SmallIO {
        bool on = false;
        bool gather = false;
        u32 outputs = 0;
    } small_io;
    unsigned char* scratch_ptr = nullptr;   // this GPU's copy of the fixed scratch array (see scratch_of_device)
    // the digit-ratio test as a table (ratio_table_fill), kept for the max_digit_ratio it was made for
    RatioArg ratio_arg = {};
    u32 ratio_bits = 0;
    bool ratio_valid = false;
    bool profiling = false;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> events;
    std::vector<trp_kernel_time> last_profile;
    // last device batch (for chunk text)
    trp_result last = {};
    u64 last_R = 0;
    bool have_last = false;
    trp_ctx* aux = nullptr;             // second set of buffers + stream for the pipelined host path
    // Test aid: TRIPLETS_AMD_FORCE_SUMMARY=1 in the environment when the context is created makes every
    // batch take the per-tile-summary path for the tile seeds (normally only batches that k_seed cannot settle).
    u32 force_summary = 0;
    // Test aid: TRIPLETS_AMD_NO_SMALL=1 in the environment when the context is created keeps batches of a few tiles
    // on the full kernel sequence (normally they take the single launch of k_small_batch).
    bool no_small = false;
    // The summary kernels (k_summary, k_tile_carry_near, k_tile_carry) are launched only while batches need them:
    // a batch whose tiles k_seed could not all settle is run again with them, and so are the batches after it
    // until one comes along that k_seed settles alone.
    bool summary_path = false;
    // batch in flight between submit_device and collect_device
    struct Pending {
        bool active = false;
        const uint8_t* d_bytes = nullptr;
        const uint64_t* d_offsets = nullptr;
        u64 R = 0, N = 0, nT = 0, g3_cap = 0, chunk_cap = 0;
        bool with_summary = false;
        u64 seq = 0;
        DevCfg dc = {};
    } pending;
    // Profiling aid: with profiling on and TRP_PROFILE_EMIT_STOP=k (1..5) in the environment, k_emit ends
    // every tile after phase k, so that the phases can be timed one by one.  The results of such a call
    // are incomplete; calls without profiling are never affected.
    u32 plan_stop() const {   // the same for k_plan (TRP_PROFILE_PLAN_STOP=1..6)
        if (!profiling) return 0;
        const char* e = getenv("TRP_PROFILE_PLAN_STOP");
        return e ? (u32)atoi(e) : 0u;
    }
    u32 emit_stop() const {
        if (!profiling) return 0;
        const char* e = getenv("TRP_PROFILE_EMIT_STOP");
        return e ? (u32)atoi(e) : 0u;
    }
};

namespace {
struct Prof {
    trp_ctx* c;
    const char* name;
    hipEvent_t a = nullptr, b = nullptr;
    Prof(trp_ctx* ctx, const char* n) : c(ctx), name(n) {
        if (c->profiling) {
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            (void)hipEventRecord(a, c->stream);
        }
    }
    ~Prof() {
        if (c->profiling) {
            (void)hipEventRecord(b, c->stream);
            c->events.push_back({name, {a, b}});
        }
    }
};

// Event pairs of a profiled submit that never reached collect_device (failed midway, or repeated).
void drop_events(trp_ctx* c) {
    for (auto& ev : c->events) {
        (void)hipEventDestroy(ev.second.first);
        (void)hipEventDestroy(ev.second.second);
    }
    c->events.clear();
}

int make_devcfg(const trp_config* cfg, DevCfg* d) {
    if (!cfg) return fail(TRP_E_ARG, "config is null");
    if (cfg->n_overlaps > TRP_MAX_OVERLAPS) return fail(TRP_E_ARG, "more than TRP_MAX_OVERLAPS overlaps");
    if (cfg->n_overlaps && !cfg->overlaps) return fail(TRP_E_ARG, "overlaps is null");
    memset(d, 0, sizeof(*d));
    d->denoise_enabled = cfg->denoise_enabled ? 1 : 0;
    d->strip_markdown = cfg->strip_markdown ? 1 : 0;
    d->max_digit_ratio = cfg->max_digit_ratio;
    d->max_window_tokens = cfg->max_window_tokens;
    d->n_overlaps = cfg->n_overlaps;
    d->summary_fallback_tokens = cfg->summary_fallback_tokens;
    for (u32 i = 0; i < cfg->n_overlaps; i++) d->overlaps[i] = cfg->overlaps[i];
    return TRP_OK;
}

// Number of u64 partial sums scan_u64 needs for n inputs.
u64 scan_partials(u64 n) {
    const u64 nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    return nb ? nb : 1;
}
int scan_u64_with(hipStream_t st, u64* partial, const u64* in, u64 n, u64* out) {
    const u64 nb = scan_partials(n);
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(NT), 0, st, in, n, partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, st, partial, nb);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(NT), 0, st, in, n, partial, out);
    HIPCHK(hipGetLastError());
    return TRP_OK;
}
// The path's own scans: values from a generator, two kernels (see k_scan_reduce_gen).  `values` is where the
// generator stores what it computes.
template <class Gen>
int scan_gen(hipStream_t st, u64* partial, const Gen& gen, const u64* values, u64 n, u64* out) {
    const u64 nb = scan_partials(n);
    if (n <= SCAN_SMALL_MAX) {
        hipLaunchKernelGGL(k_scan_small_gen<Gen>, dim3(1), dim3(NT), 0, st, gen, n, out);
        HIPCHK(hipGetLastError());
        return TRP_OK;
    }
    hipLaunchKernelGGL(k_scan_reduce_gen<Gen>, dim3((unsigned)nb), dim3(NT), 0, st, gen, n, partial);
    if (nb <= SCAN_SUM_MAX) {
        hipLaunchKernelGGL(k_scan_final_sum, dim3((unsigned)nb), dim3(NT), 0, st, values, n, partial, out);
    } else {
        hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, st, partial, nb);
        hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(NT), 0, st, values, n, partial, out);
    }
    HIPCHK(hipGetLastError());
    return TRP_OK;
}
int scan_u64(trp_ctx* c, const u64* in, u64 n, u64* out) {
    int rc = c->partial.ensure(scan_partials(n) * sizeof(u64));
    if (rc) return rc;
    return scan_u64_with(c->stream, c->partial.as<u64>(), in, n, out);
}

// Upper bound of the number of chunks of a batch with N bytes and R records.
u64 chunk_bound(u64 N, u64 R, const DevCfg& cfg) {
    u64 Tmax = (N + 1) / 2 + 1;
    u64 per_ov = R;   // one (possibly short) window per record ...
    u64 span = cfg.max_window_tokens;
    u64 bound = R;    // single windows / fallbacks
    for (u32 k = 0; k < cfg.n_overlaps; k++) {
        u64 ov = cfg.overlaps[k];
        u64 stride = span > ov ? span - ov : 0;
        if (stride < 1) stride = 1;
        bound += per_ov + Tmax / stride + R;
    }
    return bound + 16;
}
}  // namespace

extern "C" {

const char* trp_last_error(void) { return g_err.c_str(); }

static int create_ctx(int device, hipStream_t stream, bool own, trp_ctx** out) {
    if (!out) return fail(TRP_E_ARG, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(TRP_E_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(TRP_E_ARG, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    unsigned char* sp = nullptr;
    {
        int rc = scratch_of_device(device, &sp);
        if (rc) return rc;
    }
    trp_ctx* c = new trp_ctx();
    c->scratch_ptr = sp;
    { const char* e2 = getenv("TRIPLETS_AMD_FORCE_SUMMARY"); c->force_summary = (e2 && atoi(e2)) ? 1u : 0u; }
    c->summary_path = c->force_summary != 0;
    { const char* e3 = getenv("TRIPLETS_AMD_NO_SMALL"); c->no_small = e3 && atoi(e3); }
    c->device = device;
    c->owns_stream = own;
    c->stream = stream;
    if (own) {
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(TRP_E_HIP, "hipStreamCreate failed"); }
    }
    e = hipHostMalloc((void**)&c->h_totals, 16 * sizeof(u64), hipHostMallocDefault);
    if (e == hipSuccess) memset(c->h_totals, 0, 16 * sizeof(u64));
    if (e != hipSuccess) {
        if (own) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(TRP_E_HIP, "hipHostMalloc failed");
    }
    *out = c;
    return TRP_OK;
}

int trp_create(int device, trp_ctx** out) { return create_ctx(device, nullptr, true, out); }

int trp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n < 0 ? 0 : n;
}

int trp_create_on_stream(int device, void* hip_stream, trp_ctx** out) {
    return create_ctx(device, (hipStream_t)hip_stream, false, out);
}

int trp_wait_event(trp_ctx* c, void* hip_event) {
    if (!c || !hip_event) return fail(TRP_E_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamWaitEvent(c->stream, (hipEvent_t)hip_event, 0));
    return TRP_OK;
}

void trp_destroy(trp_ctx* c) {
    if (!c) return;
    if (c->aux) { trp_destroy(c->aux); c->aux = nullptr; }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->rec_lo, &c->rec_lo_w, &c->SA, &c->seedJ, &c->seedS, &c->plan,
                      &c->g3list, &c->tile_bytes, &c->tile_toks, &c->tile_flags, &c->packed,
                      &c->tile_base, &c->partial, &c->denoised, &c->den_off, &c->dropped, &c->tok_start,
                      &c->tok_len, &c->tok_off, &c->counts, &c->chunk_off, &c->chunks, &c->totals,
                      &c->in_bytes, &c->in_offsets, &c->fold_bytes, &c->fold_offsets, &c->text, &c->text_off, &c->text_len, &c->heads, &c->far_list, &c->sel};
    for (DevBuf* b : bufs) b->release();
    if (c->h_totals) (void)hipHostFree(c->h_totals);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void* trp_stream(trp_ctx* c) { return c ? (void*)c->stream : nullptr; }

int trp_set_profiling(trp_ctx* c, int on) {
    if (!c) return fail(TRP_E_ARG, "ctx is null");
    c->profiling = on != 0;
    return TRP_OK;
}

int trp_get_profile(trp_ctx* c, trp_kernel_time* out, int cap) {
    if (!c) return fail(TRP_E_ARG, "ctx is null");
    if (cap < 0 || (cap > 0 && !out)) return fail(TRP_E_ARG, "out is null");
    const int n = std::min((int)c->last_profile.size(), cap);
    for (int i = 0; i < n; i++) out[i] = c->last_profile[i];
    return n;
}

namespace {
// Shift the result arrays of one sub-batch to their place in the whole batch (pipelined host path).
__global__ void k_add_u32(u32* __restrict__ a, u64 n, u32 add) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += add;
}
__global__ void k_add_u64(u64* __restrict__ a, u64 n, u64 add) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += add;
}
__global__ void k_add_chunk_record(ChunkDesc* __restrict__ c, u64 n, u32 add) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i].record += add;
}
}  // namespace

// Device memory one pass of the path works in: scratch arrays (sized by the batch) and the result arrays.
struct PathBufs {
    // scratch
    u32* rec_lo;                  // nT + 2
    u32* rec_lo_w;                // nW + 2
    J *SA, *seedJ;                // nT + 1 each
    S* seedS;                     // nT + 1
    u32* far_list;                // nT + 1
    void* plan;                   // plan_bytes(nB): see PlanMem
    u64* g3list; u64 g3_cap;      // work list of digit-heavy lines
    u32 *tile_bytes, *tile_toks, *tile_flags;   // nW + 1 each (per sub-tile)
    u64 *packed, *tile_base;      // nW + 1, nW + 2
    u64* partial;                 // scan_partials(max(nW, R))
    u64* counts;                  // R + 1
    u64* totals;                  // 16
    // results
    u8* denoised;
    u64 *den_off, *tok_off, *chunk_off;
    u8* dropped;
    u32 *tok_start, *tok_len;
    ChunkDesc* chunks; u64 chunk_cap;
    u64 seq = 0;   // number of the batch (never 0): how k_rec_lo reports offsets that are not what the header asks for
};

// Token table entries a batch can need.  A token needs one byte, and two tokens of one record a separator byte
// between them: a record of n bytes holds at most (n + 1) / 2 tokens, the batch at most
// (N + non-empty records) / 2; a few entries of slack for the four-wide stores.
static u64 token_bound(u64 N, u64 R) { return (N + std::min<u64>(R, N)) / 2 + 4; }

// Enqueues the kernels of the whole path for one batch on `st`; nothing is allocated here.  `c` supplies the
// profiling switches only.
// Whether a batch of N bytes in R records takes the single launch when the context is not on the summary path.
static bool small_batch_ok(const trp_ctx* c, u64 N, u64 R, u64 max_tiles = SMALL_MAX_TILES) {
    return (N + TILE - 1) / TILE <= max_tiles && R <= SMALL_MAX_RECORDS && !c->force_summary && !c->no_small &&
           !c->plan_stop() && !c->emit_stop();
}

static int enqueue_path(trp_ctx* c, hipStream_t st, const PathBufs& B, const uint8_t* d_bytes, const uint64_t* d_offsets,
                        uint64_t R, uint64_t N, const DevCfg& dc, bool with_summary) {
    int rc;
    const u64 nB = (N + BLOCK - 1) / BLOCK, nT = (N + TILE - 1) / TILE, nW = (N + SUB - 1) / SUB;
    const bool pass = !dc.denoise_enabled;
    u32* rec_lo = B.rec_lo;
    u32 *tbytes = B.tile_bytes, *ttoks = B.tile_toks, *tflags = B.tile_flags;
    u64* tile_base = B.tile_base;
    u64 *den_off = B.den_off, *tok_off = B.tok_off, *chunk_off = B.chunk_off;
    u64* totals = B.totals;
    const PlanMem plan = plan_view(B.plan, nB);
    J* SA = B.SA;
    if (!pass && !dc.normalize) {
        u32 bits;
        memcpy(&bits, &dc.max_digit_ratio, 4);
        if (!c->ratio_valid || c->ratio_bits != bits) {
            ratio_table_fill(dc.max_digit_ratio, c->ratio_arg.thr1);
            c->ratio_bits = bits;
            c->ratio_valid = true;
        }
    }
    if (small_batch_ok(c, N, R, c->small_io.on ? SMALL_HOST_MAX_TILES : SMALL_MAX_TILES) && !with_summary) {
        Prof pr(c, "k_small_batch");
        SmallArgs a;
        memset(&a, 0, sizeof(a));
        a.bytes = d_bytes; a.offsets = d_offsets; a.R = R; a.N = N;
        a.cfg = dc; a.ratio = c->ratio_arg; a.force_open = 0;
        a.rec_lo = B.rec_lo; a.rec_lo_w = B.rec_lo_w; a.seedJ = B.seedJ; a.seedS = B.seedS; a.plan = B.plan;
        a.g3list = B.g3list; a.g3cap = B.g3_cap;
        a.tile_bytes = B.tile_bytes; a.tile_toks = B.tile_toks; a.tile_flags = B.tile_flags;
        a.packed = B.packed; a.tile_base = B.tile_base; a.counts = B.counts; a.totals = B.totals;
        a.denoised = B.denoised; a.den_off = B.den_off; a.tok_off = B.tok_off; a.chunk_off = B.chunk_off;
        a.dropped = B.dropped; a.tok_start = B.tok_start; a.tok_len = B.tok_len; a.chunks = B.chunks;
        a.chunk_cap = B.chunk_cap; a.host_totals = c->h_totals; a.seq = B.seq;
        if (c->small_io.on) {
            a.src_bytes = c->h_in;
            a.src_offsets = reinterpret_cast<const u64*>(c->h_in + SMALL_HOST_MAX_TILES * TILE + 64);
            a.stage_bytes = const_cast<u8*>(d_bytes);          // the context's own in_bytes / in_offsets
            a.stage_offsets = const_cast<u64*>(d_offsets);
            if (c->small_io.gather) {
                a.host_out = c->h_out; a.host_out_cap = SMALL_OUT_CAP; a.outputs = c->small_io.outputs;
                if (a.outputs & TRP_OUT_CHUNK_TEXT) {   // (sized by run_small_host)
                    a.text_len = c->text_len.as<u64>(); a.text_off = c->text_off.as<u64>();
                    a.text = c->text.as<u8>(); a.text_cap = c->text.cap - 16;
                }
            }
        }
        if (pass) hipLaunchKernelGGL(k_small_batch<2>, dim3(1), dim3(NT), 0, st, a);
        else if (dc.normalize) hipLaunchKernelGGL(k_small_batch<1>, dim3(1), dim3(NT), 0, st, a);
        else hipLaunchKernelGGL(k_small_batch<0>, dim3(1), dim3(NT), 0, st, a);
        HIPCHK(hipGetLastError());
        return TRP_OK;
    }
    {
        Prof pr(c, "k_rec_lo");   // also clears the totals and counters
        hipLaunchKernelGGL(k_rec_lo, dim3((unsigned)((R + 1 + 255) / 256)), dim3(256), 0, st, d_offsets, R, N, nT, nW, rec_lo,
                           B.rec_lo_w, totals, B.seq);
    }
    if (nT) {
        if (!pass) {
            u32* open_count = reinterpret_cast<u32*>(totals + 8);   // zeroed with the totals
            { Prof pr(c, "k_seed");
              const dim3 grid((unsigned)((nT + SEED_TPW * (NT / 64) - 1) / (SEED_TPW * (NT / 64))));
              if (dc.normalize)
                  hipLaunchKernelGGL(k_seed<true>, grid, dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nB, nT, c->force_summary,
                                     B.seedJ, B.seedS, open_count);
              else
                  hipLaunchKernelGGL(k_seed<false>, grid, dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nB, nT, c->force_summary,
                                     B.seedJ, B.seedS, open_count); }
            // the tile seeds again from per-tile summaries, for the tiles k_seed left open (the three kernels return
            // at once when there are none; they are not even launched while batches do not need them: totals[8]
            // tells the host, which then runs the batch again)
            if (with_summary) {
            { Prof pr(c, "k_summary");
              const dim3 grid((unsigned)std::min<u64>(nT, SUMMARY_GRID));
              if (dc.normalize) hipLaunchKernelGGL(k_summary<true>, grid, dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nB, nT, open_count, SA);
              else hipLaunchKernelGGL(k_summary<false>, grid, dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nB, nT, open_count, SA); }
            { Prof pr(c, "k_tile_carry");
              u32* far_count = reinterpret_cast<u32*>(totals + 7);   // zeroed with the totals
              hipLaunchKernelGGL(k_tile_carry_near, dim3((unsigned)std::min<u64>((nT + 255) / 256, 1024)), dim3(256), 0, st, SA, nT,
                                 open_count, B.seedJ, B.seedS, B.far_list, far_count);
              hipLaunchKernelGGL(k_tile_carry, dim3((unsigned)std::min<u64>((nT + NT / 64 - 1) / (NT / 64), 1024)), dim3(NT), 0, st,
                                 SA, nT, B.seedJ, B.seedS, B.far_list, far_count); }
            }
            { Prof pr(c, "k_plan");
              if (dc.normalize)
                  hipLaunchKernelGGL(k_plan<true>, dim3((unsigned)nT), dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nB,
                                     B.seedJ, B.seedS, dc, c->ratio_arg, plan, B.g3list, B.g3_cap, totals + 6, tbytes,
                                     ttoks, tflags, c->plan_stop());
              else
                  hipLaunchKernelGGL(k_plan<false>, dim3((unsigned)nT), dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nB,
                                     B.seedJ, B.seedS, dc, c->ratio_arg, plan, B.g3list, B.g3_cap, totals + 6, tbytes,
                                     ttoks, tflags, c->plan_stop()); }
            { Prof pr(c, "k_gate3");
              // (any grid is right: the workgroups stride over the list; a small batch spares the dispatch of idle ones)
              hipLaunchKernelGGL(k_gate3_solve, dim3((unsigned)std::min<u64>(4096, std::max<u64>(64, 8 * nT))), dim3(64), 0, st, d_bytes, d_offsets, rec_lo, R, N, nB, dc, plan,
                                 B.g3list, totals + 6, B.g3_cap, tbytes, ttoks, totals, B.denoised); }
        } else {
            Prof pr(c, "k_count_tokens");
            hipLaunchKernelGGL(k_count_tokens, dim3((unsigned)nT), dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nB, plan, tbytes,
                               ttoks, tflags);
        }
        { Prof pr(c, "scan_tiles");
          const GenTiles gen = {tbytes, ttoks, tflags, B.packed};
          if ((rc = scan_gen(st, B.partial, gen, B.packed, nW, tile_base))) return rc; }
        { Prof pr(c, "k_emit");
          const dim3 grid((unsigned)((nW + EMIT_WAVES - 1) / EMIT_WAVES)), block(64 * EMIT_WAVES);
          if (pass)
              hipLaunchKernelGGL((k_emit<true, EMIT_WAVES>), grid, block, 0, st, d_bytes,
                                 d_offsets, B.rec_lo_w, R, N, nB, nW, plan, tile_base, (u8*)nullptr, B.tok_start, B.tok_len,
                                 den_off, tok_off, c->emit_stop());
          else
              hipLaunchKernelGGL((k_emit<false, EMIT_WAVES>), grid, block, 0, st, d_bytes,
                                 d_offsets, B.rec_lo_w, R, N, nB, nW, plan, tile_base, B.denoised, B.tok_start, B.tok_len,
                                 den_off, tok_off, c->emit_stop()); }
    }
    GenChunkCounts gen;
    gen.tile_base = tile_base; gen.nW = nW; gen.rec_lo_w = B.rec_lo_w; gen.R = R; gen.N = N;
    gen.passthrough = pass ? 1 : 0; gen.cfg = dc;
    gen.den_off = den_off; gen.tok_off = tok_off; gen.totals = totals; gen.counts = B.counts; gen.dropped = B.dropped;
    if (R <= SCAN_SMALL_MAX && chunk_bound(N, R, dc) <= CHUNKS_SMALL_MAX_DESC) {   // (the bound, not the buffer, which only grows)
        Prof pr(c, "k_chunks_small");
        hipLaunchKernelGGL(k_chunks_small, dim3(1), dim3(NT), 0, st, gen, R, chunk_off, dc, B.chunk_cap, B.chunks, c->h_totals);
        HIPCHK(hipGetLastError());
        return TRP_OK;
    }
    { Prof pr(c, "scan_chunks");   // with the totals, the trailing offsets and the dropped flags on the way
      if ((rc = scan_gen(st, B.partial, gen, B.counts, R, chunk_off))) return rc; }
    { Prof pr(c, "k_chunk_emit");
      hipLaunchKernelGGL(k_chunk_emit, dim3((unsigned)std::min<u64>(1024, R / 256 + 1)), dim3(256), 0, st, chunk_off, tok_off, R, dc, B.chunk_cap,
                         B.chunks, totals, c->h_totals); }   // the totals reach the host with this kernel
    HIPCHK(hipGetLastError());
    return TRP_OK;
}

static int check_batch_args(const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R, uint64_t N) {
    if (!d_offsets) return fail(TRP_E_ARG, "offsets is null");
    if (N && !d_bytes) return fail(TRP_E_ARG, "bytes is null");
    if (((uintptr_t)d_bytes & 15u) != 0) return fail(TRP_E_ARG, "bytes must be 16-byte aligned");
    if (N > TRP_MAX_BATCH_BYTES || R >= 0xFFFFFFFEull) return fail(TRP_E_TOO_LARGE, "batch too large: split it by record range");
    return TRP_OK;
}

// Enqueues the whole path for one batch on the context's stream and returns without waiting.
static int submit_device(trp_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R, uint64_t N,
                         const DevCfg& dc) {
    if (!c) return fail(TRP_E_ARG, "ctx is null");
    if (c->pending.active) return fail(TRP_E_ARG, "a batch is already in flight on this context");
    int rc;
    if ((rc = check_batch_args(d_bytes, d_offsets, R, N))) return rc;
    HIPCHK(hipSetDevice(c->device));
    c->have_last = false;
    const u64 nB = (N + BLOCK - 1) / BLOCK, nT = (N + TILE - 1) / TILE, nW = (N + SUB - 1) / SUB;
    const bool pass = !dc.denoise_enabled;
    const u64 tok_cap = token_bound(N, R);
    u64 chunk_cap = chunk_bound(N, R, dc);
    const u64 chunk_cap_limit = (u64)1 << 28;   // do not pre-allocate more than 8 GiB of descriptors
    if (chunk_cap > chunk_cap_limit) chunk_cap = chunk_cap_limit;
    if (c->chunks.cap / sizeof(ChunkDesc) > chunk_cap) chunk_cap = c->chunks.cap / sizeof(ChunkDesc);

    // list of digit-heavy lines: sized for one line per 256 bytes; a batch that needs more is run again
    // with the exact size (the context keeps the larger buffer)
    u64 g3_cap = N / 256 + 4096;
    if (c->g3list.cap / sizeof(u64) > g3_cap) g3_cap = c->g3list.cap / sizeof(u64);
    if ((rc = c->rec_lo.ensure((nT + 2) * sizeof(u32)))) return rc;
    if ((rc = c->rec_lo_w.ensure((nW + 2) * sizeof(u32)))) return rc;
    if ((rc = c->tile_bytes.ensure((nW + 1) * sizeof(u32)))) return rc;
    if ((rc = c->tile_toks.ensure((nW + 1) * sizeof(u32)))) return rc;
    if ((rc = c->tile_flags.ensure((nW + 1) * sizeof(u32)))) return rc;
    if ((rc = c->packed.ensure((nW + 1) * sizeof(u64)))) return rc;
    if ((rc = c->tile_base.ensure((nW + 2) * sizeof(u64)))) return rc;
    if ((rc = c->partial.ensure(scan_partials(std::max(nW, R)) * sizeof(u64)))) return rc;
    if ((rc = c->den_off.ensure((R + 1) * sizeof(u64)))) return rc;
    if ((rc = c->tok_off.ensure((R + 1) * sizeof(u64)))) return rc;
    if ((rc = c->chunk_off.ensure((R + 2) * sizeof(u64)))) return rc;
    if ((rc = c->counts.ensure((R + 1) * sizeof(u64)))) return rc;
    if ((rc = c->dropped.ensure(R + 1))) return rc;
    if ((rc = c->tok_start.ensure(tok_cap * sizeof(u32)))) return rc;
    if ((rc = c->tok_len.ensure(tok_cap * sizeof(u32)))) return rc;
    if ((rc = c->chunks.ensure(chunk_cap * sizeof(ChunkDesc)))) return rc;
    if (!c->totals.p) {
        if ((rc = c->totals.ensure(16 * sizeof(u64)))) return rc;
        HIPCHK(hipMemsetAsync(c->totals.p, 0, 16 * sizeof(u64), c->stream));   // totals[9] is compared, never cleared
    }
    if ((rc = c->plan.ensure(plan_bytes(nB)))) return rc;
    if (!pass) {
        if ((rc = c->SA.ensure((nT + 1) * sizeof(J)))) return rc;
        if ((rc = c->denoised.ensure(N + 64))) return rc;
        if ((rc = c->seedJ.ensure((nT + 1) * sizeof(J)))) return rc;
        if ((rc = c->seedS.ensure((nT + 1) * sizeof(S)))) return rc;
        if ((rc = c->far_list.ensure((nT + 1) * sizeof(u32)))) return rc;
        if ((rc = c->g3list.ensure(g3_cap * sizeof(u64)))) return rc;
    }
    PathBufs B;
    B.rec_lo = c->rec_lo.as<u32>();
    B.rec_lo_w = c->rec_lo_w.as<u32>();
    B.SA = c->SA.as<J>(); B.seedJ = c->seedJ.as<J>(); B.seedS = c->seedS.as<S>();
    B.far_list = c->far_list.as<u32>();
    B.plan = c->plan.p;
    B.g3list = c->g3list.as<u64>(); B.g3_cap = g3_cap;
    B.tile_bytes = c->tile_bytes.as<u32>(); B.tile_toks = c->tile_toks.as<u32>(); B.tile_flags = c->tile_flags.as<u32>();
    B.packed = c->packed.as<u64>(); B.tile_base = c->tile_base.as<u64>();
    B.partial = c->partial.as<u64>();
    B.counts = c->counts.as<u64>();
    B.totals = c->totals.as<u64>();
    B.denoised = c->denoised.as<u8>();
    B.den_off = c->den_off.as<u64>(); B.tok_off = c->tok_off.as<u64>(); B.chunk_off = c->chunk_off.as<u64>();
    B.dropped = c->dropped.as<u8>();
    B.tok_start = c->tok_start.as<u32>(); B.tok_len = c->tok_len.as<u32>();
    B.chunks = c->chunks.as<ChunkDesc>(); B.chunk_cap = chunk_cap;
    B.seq = ++g_batch_seq;
    drop_events(c);
    const bool with_summary = c->summary_path;
    if ((rc = enqueue_path(c, c->stream, B, d_bytes, d_offsets, R, N, dc, with_summary))) return rc;
    c->pending.active = true;
    c->pending.d_bytes = d_bytes;
    c->pending.d_offsets = d_offsets;
    c->pending.R = R;
    c->pending.N = N;
    c->pending.nT = nT;
    c->pending.g3_cap = g3_cap;
    c->pending.chunk_cap = chunk_cap;
    c->pending.with_summary = with_summary;
    c->pending.seq = B.seq;
    c->pending.dc = dc;
    return TRP_OK;
}

// Waits for the batch in flight, repeats what overflowed an optimistic buffer, and describes the result.
static int collect_device(trp_ctx* c, trp_result* out) {
    if (!c || !out) return fail(TRP_E_ARG, "ctx/out is null");
    if (!c->pending.active) return fail(TRP_E_ARG, "no batch in flight on this context");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    const trp_ctx::Pending pd = c->pending;
    c->pending.active = false;
    hipStream_t st = c->stream;
    const DevCfg& dc = pd.dc;
    const bool pass = !dc.denoise_enabled;
    const u64 R = pd.R, nT = pd.nT;
    u64* totals = c->totals.as<u64>();
    u64 *den_off = c->den_off.as<u64>(), *tok_off = c->tok_off.as<u64>(), *chunk_off = c->chunk_off.as<u64>();
    HIPCHK(hipStreamSynchronize(st));
    if (c->h_totals[9] == pd.seq)
        return fail(TRP_E_ARG, "offsets must start at 0, be non-decreasing and end at n_bytes");
    if (!pass && nT) {
        const bool open = (u32)c->h_totals[8] != 0;   // k_seed left tiles open (or the test switch forces the path)
        if (open && !pd.with_summary) {
            // the tile seeds of this batch are incomplete: run it again with the summary kernels
            c->summary_path = true;
            if ((rc = submit_device(c, pd.d_bytes, pd.d_offsets, pd.R, pd.N, dc))) return rc;
            return collect_device(c, out);
        }
        if (!open && !c->force_summary) c->summary_path = false;
    }
    if (!pass && nT && c->h_totals[5]) {
        // more digit-heavy lines than the work list holds: grow it to the exact size and run the batch again
        if (c->h_totals[4] <= pd.g3_cap) return fail(TRP_E_HIP, "inconsistent gate-3 list size");
        if ((rc = c->g3list.ensure(c->h_totals[4] * sizeof(u64)))) return rc;
        if ((rc = submit_device(c, pd.d_bytes, pd.d_offsets, pd.R, pd.N, dc))) return rc;
        return collect_device(c, out);
    }
    if (c->h_totals[3]) {
        // the optimistic descriptor buffer was too small: grow to the exact size and emit again
        const u64 chunk_cap = c->h_totals[2];
        if ((rc = c->chunks.ensure(chunk_cap * sizeof(ChunkDesc)))) return rc;
        hipLaunchKernelGGL(k_chunk_emit, dim3(1024), dim3(256), 0, st, chunk_off, tok_off, R, dc, chunk_cap,
                           c->chunks.as<ChunkDesc>(), totals, c->h_totals);
        HIPCHK(hipStreamSynchronize(st));
        if (c->h_totals[3]) return fail(TRP_E_HIP, "chunk descriptor buffer overflow");
    }
    if (c->profiling) {
        c->last_profile.clear();
        for (auto& ev : c->events) {
            trp_kernel_time kt;
            memset(&kt, 0, sizeof(kt));
            snprintf(kt.name, sizeof(kt.name), "%s", ev.first.c_str());
            (void)hipEventElapsedTime(&kt.ms, ev.second.first, ev.second.second);
            (void)hipEventDestroy(ev.second.first);
            (void)hipEventDestroy(ev.second.second);
            c->last_profile.push_back(kt);
        }
        c->events.clear();
    }
    memset(out, 0, sizeof(*out));
    out->denoised = pass ? const_cast<uint8_t*>(pd.d_bytes) : c->denoised.as<u8>();
    out->denoised_offsets = den_off;
    out->dropped = c->dropped.as<u8>();
    out->tok_start = c->tok_start.as<u32>();
    out->tok_len = c->tok_len.as<u32>();
    out->tok_offsets = tok_off;
    out->chunks = reinterpret_cast<trp_chunk*>(c->chunks.as<ChunkDesc>());
    out->chunk_offsets = chunk_off;
    out->n_records = R;
    out->n_denoised = c->h_totals[0];
    out->n_tokens = c->h_totals[1];
    out->n_chunks = c->h_totals[2];
    c->last = *out;
    c->last_R = R;
    c->have_last = true;
    return TRP_OK;
}

uint64_t trp_token_bound(uint64_t n_bytes, uint64_t n_records) { return token_bound(n_bytes, n_records); }

// The path with every working array in this GPU's fixed scratch array and every result array the caller's: no
// hipMalloc / hipFree / hipHostMalloc / event creation, here or in anything it calls.
int trp_preprocess_batch_device_noalloc(trp_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R,
                                        uint64_t N, const trp_config* cfg, const trp_out_buffers* ob, trp_result* out) {
    if (!c || !ob || !out) return fail(TRP_E_ARG, "ctx/buffers/out is null");
    if (c->pending.active) return fail(TRP_E_ARG, "a batch is already in flight on this context");
    DevCfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    if ((rc = check_batch_args(d_bytes, d_offsets, R, N))) return rc;
    if (N > NOALLOC_MAX_BYTES) return fail(TRP_E_TOO_LARGE, "batch exceeds TRP_NOALLOC_MAX_BYTES: split it by record range");
    if (R > NOALLOC_MAX_RECORDS) return fail(TRP_E_TOO_LARGE, "batch exceeds TRP_NOALLOC_MAX_RECORDS: split it by record range");
    const bool pass = !dc.denoise_enabled;
    if (!ob->denoised_offsets || !ob->tok_offsets || !ob->chunk_offsets || (R && !ob->dropped))
        return fail(TRP_E_ARG, "a per-record output array is null");
    if (!pass && N && (!ob->denoised || ob->denoised_cap < N))
        return fail(TRP_E_ARG, "denoised buffer must hold n_bytes bytes");
    if (((uintptr_t)ob->denoised & 15u) != 0) return fail(TRP_E_ARG, "denoised buffer must be 16-byte aligned");
    if (N && (!ob->tok_start || !ob->tok_len || ob->tok_cap < token_bound(N, R)))
        return fail(TRP_E_ARG, "token arrays must hold trp_token_bound(n_bytes, n_records) entries");
    if (ob->chunk_cap && !ob->chunks) return fail(TRP_E_ARG, "chunks is null");
    HIPCHK(hipSetDevice(c->device));
    c->have_last = false;   // whatever happens, the previous batch is no longer "the last batch" of this context
    unsigned char* sp = c->scratch_ptr;   // never the array's name: that is not a device address on the host
    if (!sp) return fail(TRP_E_HIP, "scratch array address unknown");
    constexpr ScratchLayout L = scratch_layout();
    PathBufs B;
    B.totals = reinterpret_cast<u64*>(sp + L.totals);
    B.rec_lo = reinterpret_cast<u32*>(sp + L.rec_lo);
    B.rec_lo_w = reinterpret_cast<u32*>(sp + L.rec_lo_w);
    B.SA = reinterpret_cast<J*>(sp + L.SA);
    B.seedJ = reinterpret_cast<J*>(sp + L.seedJ);
    B.seedS = reinterpret_cast<S*>(sp + L.seedS);
    B.far_list = reinterpret_cast<u32*>(sp + L.far_list);
    B.tile_bytes = reinterpret_cast<u32*>(sp + L.tile_bytes);
    B.tile_toks = reinterpret_cast<u32*>(sp + L.tile_toks);
    B.tile_flags = reinterpret_cast<u32*>(sp + L.tile_flags);
    B.packed = reinterpret_cast<u64*>(sp + L.packed);
    B.tile_base = reinterpret_cast<u64*>(sp + L.tile_base);
    B.partial = reinterpret_cast<u64*>(sp + L.partial);
    B.counts = reinterpret_cast<u64*>(sp + L.counts);
    B.g3list = reinterpret_cast<u64*>(sp + L.g3list);
    B.g3_cap = NOALLOC_G3_CAP;
    B.plan = sp + L.plan;
    B.denoised = ob->denoised;
    B.den_off = ob->denoised_offsets; B.tok_off = ob->tok_offsets; B.chunk_off = ob->chunk_offsets;
    B.dropped = ob->dropped;
    B.tok_start = ob->tok_start; B.tok_len = ob->tok_len;
    B.chunks = reinterpret_cast<ChunkDesc*>(ob->chunks);
    B.chunk_cap = ob->chunk_cap;
    // per-kernel event pairs would have to be created: this entry point is never profiled
    const bool was_profiling = c->profiling;
    c->profiling = false;
    for (int attempt = 0; attempt < 2; attempt++) {
        const bool with_summary = c->summary_path;
        B.seq = ++g_batch_seq;
        rc = enqueue_path(c, c->stream, B, d_bytes, d_offsets, R, N, dc, with_summary);   // starts by clearing the totals and counters
        if (rc) break;
        hipError_t e = hipStreamSynchronize(c->stream);   // the last kernel has written the totals into pinned memory
        if (e != hipSuccess) { rc = fail(TRP_E_HIP, std::string("noalloc batch: ") + hipGetErrorString(e)); break; }
        if (c->h_totals[9] == B.seq) { rc = fail(TRP_E_ARG, "offsets must start at 0, be non-decreasing and end at n_bytes"); break; }
        const bool open = !pass && N && (u32)c->h_totals[8] != 0;
        if (open && !with_summary) { c->summary_path = true; continue; }   // once more, with the summary kernels
        if (!pass && N && !open && !c->force_summary) c->summary_path = false;
        break;
    }
    c->profiling = was_profiling;
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    out->n_records = R;
    out->n_denoised = c->h_totals[0];
    out->n_tokens = c->h_totals[1];
    out->n_chunks = c->h_totals[2];
    if (!pass && N && c->h_totals[5])
        return fail(TRP_E_TOO_LARGE, "more digit-heavy lines outside the in-tile solver than the fixed work list holds");
    if (c->h_totals[3])
        return fail(TRP_E_TOO_LARGE, "chunk descriptor buffer too small: " + std::to_string(c->h_totals[2]) + " needed");
    out->denoised = pass ? const_cast<uint8_t*>(d_bytes) : ob->denoised;
    out->denoised_offsets = ob->denoised_offsets;
    out->dropped = ob->dropped;
    out->tok_start = ob->tok_start;
    out->tok_len = ob->tok_len;
    out->tok_offsets = ob->tok_offsets;
    out->chunks = ob->chunks;
    out->chunk_offsets = ob->chunk_offsets;
    // the text renderers (trp_materialize_chunk_text_device, ...) work on the last batch of the context: this one,
    // through the caller's buffers (they allocate their own output; only the batch call itself allocates nothing)
    c->last = *out;
    c->last_R = R;
    c->have_last = true;
    return TRP_OK;
}

static int run_device(trp_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R, uint64_t N,
                      const DevCfg& dc, trp_result* out) {
    if (!c || !out) return fail(TRP_E_ARG, "ctx/out is null");
    int rc = submit_device(c, d_bytes, d_offsets, R, N, dc);
    if (rc) return rc;
    return collect_device(c, out);
}

// The preprocessor fold of a strategy with several denoisers (sliding_window.rs:20-34): the stages that do
// something run one after the other as whole batches, the text of one being the input of the next.  A record a
// stage returns None for has no text; the next stage returns None for an empty text as well, so the flags of the
// last stage say "some stage returned None" - as long as every stage that runs is an enabled one, which is why the
// disabled ones (identity, denoiser_preprocessor.rs:222-224) are left out rather than run as pass-through stages.
static int run_device_fold(trp_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R, uint64_t N,
                           const DevCfg& dc, const trp_denoiser* more, uint32_t n_more, trp_result* out) {
    if (!c || !out) return fail(TRP_E_ARG, "ctx/out is null");
    if (n_more && !more) return fail(TRP_E_ARG, "more is null");
    if (dc.normalize) return fail(TRP_E_ARG, "the fold takes denoisers only");
    std::vector<DevCfg> stages;
    if (dc.denoise_enabled) stages.push_back(dc);
    for (uint32_t i = 0; i < n_more; i++) {
        if (!more[i].denoise_enabled) continue;
        DevCfg x = dc;
        x.denoise_enabled = 1;
        x.strip_markdown = more[i].strip_markdown ? 1 : 0;
        x.max_digit_ratio = more[i].max_digit_ratio;
        stages.push_back(x);
    }
    if (stages.empty()) {
        DevCfg x = dc;
        x.denoise_enabled = 0;
        return run_device(c, d_bytes, d_offsets, R, N, x, out);
    }
    HIPCHK(hipSetDevice(c->device));
    const uint8_t* in_b = d_bytes;
    const uint64_t* in_o = d_offsets;
    uint64_t n = N;
    for (size_t k = 0; k < stages.size(); k++) {
        int rc = run_device(c, in_b, in_o, R, n, stages[k], out);
        if (rc || k + 1 == stages.size()) return rc;
        // (the context's result arrays are overwritten by the next stage: its input goes into arrays of its own)
        if ((rc = c->fold_bytes.ensure(out->n_denoised + 64))) return rc;
        if ((rc = c->fold_offsets.ensure((R + 1) * sizeof(u64)))) return rc;
        if (out->n_denoised)
            HIPCHK(hipMemcpyAsync(c->fold_bytes.p, out->denoised, out->n_denoised, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->fold_offsets.p, out->denoised_offsets, (R + 1) * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        in_b = c->fold_bytes.as<u8>();
        in_o = c->fold_offsets.as<u64>();
        n = out->n_denoised;
    }
    return TRP_OK;
}

// Text of `C` descriptors over the last device batch: lengths, scan, emit.  chunk_off == nullptr for head
// descriptors.
static int render_text(trp_ctx* c, const ChunkDesc* chunks, u64 C, const u64* chunk_off, uint8_t** d_text,
                       uint64_t** d_text_offsets, uint64_t* n_text_bytes) {
    const trp_result& L = c->last;
    int rc;
    c->text_host_packed = false;
    if ((rc = c->text_len.ensure((C + 1) * sizeof(u64)))) return rc;
    if ((rc = c->text_off.ensure((C + 2) * sizeof(u64)))) return rc;
    if (C <= RENDER_SMALL_MAX && L.n_denoised <= ((u64)1 << 20) && !c->no_small) {
        // a few descriptors over a small batch: one launch into a buffer that is usually large enough
        if ((rc = c->text.ensure(((u64)4 << 20) + 16))) return rc;
        c->h_totals[14] = 0;
        u8* host_pack = nullptr;
        if (c->want_host_pack) {
            if (!c->h_out) HIPCHK(hipHostMalloc((void**)&c->h_out, H_OUT_BYTES, hipHostMallocDefault));
            host_pack = c->h_out + SMALL_OUT_CAP;
        }
        hipLaunchKernelGGL(k_render_text_small, dim3(1), dim3(NT), 0, c->stream, chunks, C, chunk_off, L.tok_offsets,
                           L.denoised_offsets, L.tok_start, L.tok_len, L.denoised, c->text_len.as<u64>(), c->text_off.as<u64>(),
                           c->text.as<u8>(), c->text.cap - 16, c->h_totals, host_pack, SMALL_OUT_CAP);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        c->text_host_packed = c->h_totals[14] == 3;
        if (c->h_totals[14] == 1 || c->h_totals[14] == 3) {
            *d_text = c->text.as<u8>();
            *d_text_offsets = c->text_off.as<u64>();
            *n_text_bytes = c->h_totals[4];
            return TRP_OK;
        }
        // (the text did not fit: the general way below sizes the buffer first)
    }
    if (C) hipLaunchKernelGGL(k_chunk_text_len, dim3((unsigned)((C * 64 + NT - 1) / NT)), dim3(NT), 0, c->stream, chunks, C,
                              chunk_off, L.tok_offsets, L.denoised_offsets, L.tok_len, c->text_len.as<u64>());
    if ((rc = scan_u64(c, c->text_len.as<u64>(), C, c->text_off.as<u64>()))) return rc;
    HIPCHK(hipMemcpyAsync(c->h_totals + 4, c->text_off.as<u64>() + C, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    u64 nbytes = c->h_totals[4];
    if ((rc = c->text.ensure(nbytes + 16))) return rc;
    if (C) hipLaunchKernelGGL(k_chunk_text_emit, dim3((unsigned)((C * 64 + NT - 1) / NT)), dim3(NT), 0, c->stream, chunks, C,
                              chunk_off, L.tok_offsets, L.denoised_offsets, L.tok_start, L.tok_len, L.denoised,
                              c->text_off.as<u64>(), c->text.as<u8>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    *d_text = c->text.as<u8>();
    *d_text_offsets = c->text_off.as<u64>();
    *n_text_bytes = nbytes;
    return TRP_OK;
}

int trp_materialize_chunk_text_device(trp_ctx* c, uint8_t** d_text, uint64_t** d_text_offsets, uint64_t* n_text_bytes) {
    if (!c || !d_text || !d_text_offsets || !n_text_bytes) return fail(TRP_E_ARG, "null argument");
    if (!c->have_last) return fail(TRP_E_ARG, "no device batch result to render");
    HIPCHK(hipSetDevice(c->device));
    const trp_result& L = c->last;
    return render_text(c, reinterpret_cast<const ChunkDesc*>(L.chunks), L.n_chunks, L.chunk_offsets, d_text,
                       d_text_offsets, n_text_bytes);
}

int trp_head_tokens_text_device(trp_ctx* c, uint32_t max_tokens, uint8_t** d_text, uint64_t** d_text_offsets,
                                uint64_t* n_text_bytes) {
    if (!c || !d_text || !d_text_offsets || !n_text_bytes) return fail(TRP_E_ARG, "null argument");
    if (!c->have_last) return fail(TRP_E_ARG, "no device batch result to render");
    HIPCHK(hipSetDevice(c->device));
    const u64 R = c->last_R;
    int rc;
    if ((rc = c->heads.ensure((R + 1) * sizeof(ChunkDesc)))) return rc;
    if (R) hipLaunchKernelGGL(k_head_desc, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, c->stream, c->last.tok_offsets, R,
                              max_tokens, c->heads.as<ChunkDesc>());
    return render_text(c, c->heads.as<ChunkDesc>(), R, (const u64*)nullptr, d_text, d_text_offsets, n_text_bytes);
}

int trp_chunk_bodies_device(trp_ctx* c, const uint32_t* chunk_index, const uint32_t* budget, uint64_t n,
                            uint8_t** d_text, uint64_t** d_text_offsets, uint64_t* n_text_bytes) {
    if (!c || !d_text || !d_text_offsets || !n_text_bytes || (n && (!chunk_index || !budget)))
        return fail(TRP_E_ARG, "null argument");
    if (!c->have_last) return fail(TRP_E_ARG, "no device batch result to render");
    HIPCHK(hipSetDevice(c->device));
    const trp_result& L = c->last;
    for (uint64_t i = 0; i < n; i++)
        if (chunk_index[i] >= L.n_chunks) return fail(TRP_E_ARG, "chunk index out of range");
    int rc;
    if ((rc = c->heads.ensure((n + 1) * sizeof(ChunkDesc)))) return rc;
    if ((rc = c->sel.ensure((2 * n + 2) * sizeof(u32)))) return rc;
    u32* d_idx = c->sel.as<u32>();
    u32* d_bud = d_idx + n;
    if (n && 2 * n * sizeof(u32) <= SMALL_IN_CAP) {
        // a short selection: through the page-locked block, which the kernel reads itself (no copy commands, and
        // nothing to wait for: the block is not touched again before this call's own synchronise)
        if (!c->h_in) HIPCHK(hipHostMalloc((void**)&c->h_in, SMALL_IN_CAP, hipHostMallocDefault));
        u32* h = reinterpret_cast<u32*>(c->h_in);
        memcpy(h, chunk_index, n * sizeof(u32));
        memcpy(h + n, budget, n * sizeof(u32));
        hipLaunchKernelGGL(k_body_desc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           reinterpret_cast<const ChunkDesc*>(L.chunks), L.n_chunks, h, h + n, n, c->heads.as<ChunkDesc>());
    } else if (n) {
        HIPCHK(hipMemcpyAsync(d_idx, chunk_index, n * sizeof(u32), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_bud, budget, n * sizeof(u32), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_body_desc, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           reinterpret_cast<const ChunkDesc*>(L.chunks), L.n_chunks, d_idx, d_bud, n, c->heads.as<ChunkDesc>());
        HIPCHK(hipStreamSynchronize(c->stream));   // the index arrays are the caller's (pageable) memory
    }
    return render_text(c, c->heads.as<ChunkDesc>(), n, L.chunk_offsets, d_text, d_text_offsets, n_text_bytes);
}

int trp_bm25_texts_device(trp_ctx* c, const uint64_t* doc_offsets, uint64_t n_docs, uint32_t max_tokens,
                          const char* newline, uint8_t** d_text, uint64_t** d_text_offsets, uint64_t* n_text_bytes) {
    if (!c || !d_text || !d_text_offsets || !n_text_bytes || !doc_offsets || !newline)
        return fail(TRP_E_ARG, "null argument");
    if (!c->have_last) return fail(TRP_E_ARG, "no device batch result to render");
    const size_t nl_len = strlen(newline);
    if (nl_len > 4) return fail(TRP_E_ARG, "newline longer than 4 bytes");
    const u64 R = c->last_R;
    for (uint64_t i = 0; i < n_docs; i++)
        if (doc_offsets[i] > doc_offsets[i + 1] || doc_offsets[i + 1] > R)
            return fail(TRP_E_ARG, "document offsets must be non-decreasing record indices of the last batch");
    HIPCHK(hipSetDevice(c->device));
    u32 nl_bytes = 0;
    for (size_t i = 0; i < nl_len; i++) nl_bytes |= (u32)(u8)newline[i] << (8 * i);
    const trp_result& L = c->last;
    int rc;
    if ((rc = c->sel.ensure((n_docs + 1) * sizeof(u64)))) return rc;
    if ((rc = c->text_len.ensure((n_docs + 1) * sizeof(u64)))) return rc;
    if ((rc = c->text_off.ensure((n_docs + 2) * sizeof(u64)))) return rc;
    u64* d_doc = c->sel.as<u64>();
    HIPCHK(hipMemcpyAsync(d_doc, doc_offsets, (n_docs + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    const dim3 grid((unsigned)((n_docs * 64 + NT - 1) / NT));
    if (n_docs) hipLaunchKernelGGL(k_doc_text, grid, dim3(NT), 0, c->stream, d_doc, n_docs, max_tokens, (u32)nl_len, nl_bytes,
                                   L.tok_offsets, L.denoised_offsets, L.tok_start, L.tok_len, L.denoised,
                                   c->text_len.as<u64>(), (const u64*)nullptr, (u8*)nullptr);
    if ((rc = scan_u64(c, c->text_len.as<u64>(), n_docs, c->text_off.as<u64>()))) return rc;
    HIPCHK(hipMemcpyAsync(c->h_totals + 4, c->text_off.as<u64>() + n_docs, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const u64 nbytes = c->h_totals[4];
    if ((rc = c->text.ensure(nbytes + 16))) return rc;
    if (n_docs) hipLaunchKernelGGL(k_doc_text, grid, dim3(NT), 0, c->stream, d_doc, n_docs, max_tokens, (u32)nl_len, nl_bytes,
                                   L.tok_offsets, L.denoised_offsets, L.tok_start, L.tok_len, L.denoised,
                                   (u64*)nullptr, c->text_off.as<u64>(), c->text.as<u8>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    *d_text = c->text.as<u8>();
    *d_text_offsets = c->text_off.as<u64>();
    *n_text_bytes = nbytes;
    return TRP_OK;
}

static DevCfg normalize_devcfg() {
    DevCfg dc;
    memset(&dc, 0, sizeof(dc));
    dc.denoise_enabled = 1;        // runs the planning kernels ...
    dc.normalize = 1;              // ... with the whitespace-normalising plan instead of the denoiser gates
    dc.max_digit_ratio = 0.35f;
    dc.max_window_tokens = 1024;   // reference defaults for the chunk descriptors that come with the result
    dc.n_overlaps = 1;
    dc.overlaps[0] = 64;
    dc.summary_fallback_tokens = 512;
    return dc;
}

int trp_preprocess_batch_device(trp_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R,
                                uint64_t N, const trp_config* cfg, trp_result* out) {
    DevCfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    return run_device(c, d_bytes, d_offsets, R, N, dc, out);
}

int trp_submit_batch_device(trp_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R, uint64_t N,
                            const trp_config* cfg) {
    DevCfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    return submit_device(c, d_bytes, d_offsets, R, N, dc);
}

int trp_wait_batch(trp_ctx* c, trp_result* out) { return collect_device(c, out); }

int trp_normalize_batch_device(trp_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R,
                               uint64_t N, trp_result* out) {
    return run_device(c, d_bytes, d_offsets, R, N, normalize_devcfg(), out);
}

int trp_copy_to_host(trp_ctx* c, void* dst, const void* d_src, size_t bytes) {
    if (!c || (bytes && (!dst || !d_src))) return fail(TRP_E_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TRP_OK;
}

}  // extern "C"

// Host result arrays.  Large arrays live in page-locked memory so that the device-to-host copies run
// at link speed; the blocks are recycled through a process-wide pool (pinning is expensive), small
// arrays come from malloc.  host_release accepts pointers of either kind.
namespace {
struct HostPool {
    static constexpr size_t PIN_MIN = 256u << 10;          // smaller arrays are not worth pinning
    static constexpr size_t KEEP_MAX = (size_t)16 << 30;   // free pinned bytes kept for reuse
    std::mutex mu;
    std::unordered_map<void*, size_t> live;                // pinned blocks handed out -> capacity
    std::vector<std::pair<void*, size_t>> idle;            // pinned blocks ready for reuse
    size_t idle_bytes = 0;

    void* alloc(size_t n) {
        if (n < PIN_MIN) return malloc(n ? n : 1);
        {
            std::lock_guard<std::mutex> g(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].second >= n && idle[i].second <= 2 * n + (1u << 20) &&
                    (best == idle.size() || idle[i].second < idle[best].second))
                    best = i;
            if (best != idle.size()) {
                auto blk = idle[best];
                idle.erase(idle.begin() + best);
                idle_bytes -= blk.second;
                live[blk.first] = blk.second;
                return blk.first;
            }
        }
        const size_t cap = (n + n / 8 + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        void* p = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            return malloc(n);   // pageable memory still works, only slower
        }
        std::lock_guard<std::mutex> g(mu);
        live[p] = cap;
        return p;
    }
    void release(void* p) {
        if (!p) return;
        bool drop = false;
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = live.find(p);
            if (it == live.end()) { free(p); return; }
            const size_t cap = it->second;
            live.erase(it);
            if (idle_bytes + cap <= KEEP_MAX) { idle.emplace_back(p, cap); idle_bytes += cap; }
            else drop = true;
        }
        if (drop) (void)hipHostFree(p);
    }
    void trim() {
        std::vector<std::pair<void*, size_t>> blocks;
        {
            std::lock_guard<std::mutex> g(mu);
            blocks.swap(idle);
            idle_bytes = 0;
        }
        for (auto& b : blocks) (void)hipHostFree(b.first);
    }
};
HostPool& host_pool() {
    static HostPool* pool = new HostPool();   // never destroyed: results may outlive static teardown
    return *pool;
}
}  // namespace

extern "C" {

void trp_result_free(trp_result* r) {
    if (!r) return;
    HostPool& hp = host_pool();
    hp.release(r->denoised); hp.release(r->denoised_offsets); hp.release(r->dropped); hp.release(r->tok_start);
    hp.release(r->tok_len); hp.release(r->tok_offsets); hp.release(r->chunks); hp.release(r->chunk_offsets);
    memset(r, 0, sizeof(*r));
}

void trp_free(void* p) { host_pool().release(p); }

void trp_trim_host_pool(void) { host_pool().trim(); }

// Host batch in sub-batches of consecutive records, alternating between two sets of device buffers and two
// streams: while the results of one sub-batch travel to the host, the text of the next travels to the
// device (the two directions of the link run concurrently; the kernels are a small fraction of the time).
// The result is assembled in place: every sub-batch's arrays are shifted on the device and copied straight
// to their final position in the (page-locked) result arrays.
constexpr u64 PIPE_SUB_BYTES = 96ull << 20;
constexpr u64 PIPE_MIN_BYTES = 256ull << 20;

static int run_host_pipelined(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const DevCfg& dc,
                              uint32_t outputs, trp_result* out, trp_chunk_text* ctext) {
    const u64 N = offsets[R];
    int rc;
    if (!c->aux && (rc = trp_create(c->device, &c->aux))) return rc;
    trp_ctx* cx[2] = {c, c->aux};
    // sub-batch boundaries: first record at or after each multiple of the sub-batch size
    std::vector<u64> cut(1, 0);
    for (u64 target = PIPE_SUB_BYTES; target < N; target += PIPE_SUB_BYTES) {
        u64 r = (u64)(std::lower_bound(offsets, offsets + R, target) - offsets);
        if (r > cut.back() && r < R) cut.push_back(r);
    }
    cut.push_back(R);
    const size_t S = cut.size() - 1;
    u64 max_bytes = 0, max_recs = 0;
    for (size_t i = 0; i < S; i++) {
        max_bytes = std::max(max_bytes, offsets[cut[i + 1]] - offsets[cut[i]]);
        max_recs = std::max(max_recs, cut[i + 1] - cut[i]);
    }
    for (trp_ctx* x : cx) {
        if ((rc = x->in_bytes.ensure(max_bytes + 64))) return rc;
        if ((rc = x->in_offsets.ensure((max_recs + 1) * sizeof(u64)))) return rc;
    }
    std::vector<u64> rebased[2];
    auto send = [&](size_t i) -> int {   // text and offsets of sub-batch i to its device buffers (asynchronous)
        trp_ctx* x = cx[i & 1];
        const u64 r0 = cut[i], r1 = cut[i + 1], b0 = offsets[r0], nb = offsets[r1] - b0;
        std::vector<u64>& rb = rebased[i & 1];
        rb.resize(r1 - r0 + 1);
        for (u64 j = 0; j <= r1 - r0; j++) rb[j] = offsets[r0 + j] - b0;
        if (nb) HIPCHK(hipMemcpyAsync(x->in_bytes.p, bytes + b0, nb, hipMemcpyHostToDevice, x->stream));
        HIPCHK(hipMemcpyAsync(x->in_offsets.p, rb.data(), rb.size() * sizeof(u64), hipMemcpyHostToDevice, x->stream));
        return TRP_OK;
    };

    memset(out, 0, sizeof(*out));
    HostPool& hp = host_pool();
    const bool want_den = outputs & TRP_OUT_TEXT, want_tok = outputs & TRP_OUT_TOKENS, want_ch = outputs & TRP_OUT_CHUNKS,
               want_ct = (outputs & TRP_OUT_CHUNK_TEXT) && ctext;
    u64 cap_den = 0, cap_tok = 0, cap_ch = 0, cap_ct = 0, cap_cto = 0;   // capacities of the data-dependent arrays
    u64 base_d = 0, base_t = 0, base_c = 0, base_x = 0;
    // on failure: copies into the result arrays may still be in flight on either stream
    auto fail_free = [&](int code) {
        for (trp_ctx* x : cx) (void)hipStreamSynchronize(x->stream);
        trp_result_free(out);
        if (ctext) { hp.release(ctext->text); hp.release(ctext->offsets); memset(ctext, 0, sizeof(*ctext)); }
        return code;
    };
    auto sync_all = [&]() -> int {
        for (trp_ctx* x : cx) HIPCHK(hipStreamSynchronize(x->stream));
        return TRP_OK;
    };
    // grow one array to hold `need` elements (rare: only when the first sub-batch under-estimated the rest)
    auto grow = [&](void** p, u64* cap, u64 need, u64 used, size_t elem) -> int {
        if (need <= *cap) return TRP_OK;
        int r2 = sync_all();
        if (r2) return r2;
        const u64 ncap = need + need / 2 + 1024;
        void* q = hp.alloc(ncap * elem);
        if (!q) return fail(TRP_E_ARG, "out of host memory");
        if (*p && used) memcpy(q, *p, used * elem);
        hp.release(*p);
        *p = q;
        *cap = ncap;
        return TRP_OK;
    };
    out->denoised_offsets = (u64*)hp.alloc((R + 1) * sizeof(u64));
    out->tok_offsets = (u64*)hp.alloc((R + 1) * sizeof(u64));
    out->chunk_offsets = (u64*)hp.alloc((R + 1) * sizeof(u64));
    out->dropped = (u8*)hp.alloc(R ? R : 1);
    if (!out->denoised_offsets || !out->tok_offsets || !out->chunk_offsets || !out->dropped)
        return fail_free(fail(TRP_E_ARG, "out of host memory"));

    if ((rc = send(0))) return fail_free(rc);
    for (size_t i = 0; i < S; i++) {
        trp_ctx* x = cx[i & 1];
        const u64 r0 = cut[i], r1 = cut[i + 1], Rs = r1 - r0, Ns = offsets[r1] - offsets[r0];
        if (i + 1 < S && (rc = send(i + 1))) return fail_free(rc);
        trp_result d;
        if ((rc = run_device(x, x->in_bytes.as<u8>(), x->in_offsets.as<u64>(), Rs, Ns, dc, &d))) return fail_free(rc);
        // chunk text is rendered on the device from this sub-batch's own arrays, before they are shifted
        uint8_t* d_ct = nullptr;
        uint64_t* d_cto = nullptr;
        uint64_t n_ct = 0;
        if (want_ct && (rc = render_text(x, reinterpret_cast<const ChunkDesc*>(d.chunks), d.n_chunks, d.chunk_offsets, &d_ct,
                                         &d_cto, &n_ct))) return fail_free(rc);
        if (i == 0) {   // size the result arrays from the first sub-batch's yield, with a margin
            const double scale = Ns ? (double)N / (double)Ns * 1.08 : 1.0;
            u64 want_d = std::min<u64>(N + 64, (u64)((double)d.n_denoised * scale) + (1u << 20));
            if (want_den && (rc = grow((void**)&out->denoised, &cap_den, S == 1 ? d.n_denoised : want_d, 0, 1))) return fail_free(rc);
            u64 want_t = (u64)((double)d.n_tokens * scale) + (1u << 18);
            u64 tcap = 0;
            if (want_tok && (rc = grow((void**)&out->tok_start, &tcap, S == 1 ? d.n_tokens : want_t, 0, sizeof(u32)))) return fail_free(rc);
            if (want_tok && (rc = grow((void**)&out->tok_len, &cap_tok, S == 1 ? d.n_tokens : want_t, 0, sizeof(u32)))) return fail_free(rc);
            u64 want_c = (u64)((double)d.n_chunks * scale) + R / 8 + 4096;
            if (want_ch && (rc = grow((void**)&out->chunks, &cap_ch, S == 1 ? d.n_chunks : want_c, 0, sizeof(trp_chunk)))) return fail_free(rc);
            if (want_ct) {
                if ((rc = grow((void**)&ctext->text, &cap_ct, S == 1 ? n_ct : (u64)((double)n_ct * scale) + (1u << 20), 0, 1))) return fail_free(rc);
                if ((rc = grow((void**)&ctext->offsets, &cap_cto, (S == 1 ? d.n_chunks : want_c) + 1, 0, sizeof(u64)))) return fail_free(rc);
            }
        }
        if (want_den && base_d + d.n_denoised > cap_den &&
            (rc = grow((void**)&out->denoised, &cap_den, base_d + d.n_denoised, base_d, 1))) return fail_free(rc);
        if (want_tok && base_t + d.n_tokens > cap_tok) {
            u64 tcap = cap_tok;
            if ((rc = grow((void**)&out->tok_start, &tcap, base_t + d.n_tokens, base_t, sizeof(u32)))) return fail_free(rc);
            if ((rc = grow((void**)&out->tok_len, &cap_tok, base_t + d.n_tokens, base_t, sizeof(u32)))) return fail_free(rc);
        }
        if (want_ch && base_c + d.n_chunks > cap_ch &&
            (rc = grow((void**)&out->chunks, &cap_ch, base_c + d.n_chunks, base_c, sizeof(trp_chunk)))) return fail_free(rc);
        if (want_ct) {
            if (base_x + n_ct > cap_ct && (rc = grow((void**)&ctext->text, &cap_ct, base_x + n_ct, base_x, 1))) return fail_free(rc);
            if (base_c + d.n_chunks + 1 > cap_cto &&
                (rc = grow((void**)&ctext->offsets, &cap_cto, base_c + d.n_chunks + 1, base_c + 1, sizeof(u64)))) return fail_free(rc);
        }
        // shift on the device, then copy to the final position
        hipStream_t st = x->stream;
        auto blocks = [](u64 n) { return dim3((unsigned)((n + 255) / 256)); };
        if (want_ct && base_x) hipLaunchKernelGGL(k_add_u64, blocks(d.n_chunks + 1), dim3(256), 0, st, d_cto, d.n_chunks + 1, base_x);
        if (base_d && d.n_tokens) hipLaunchKernelGGL(k_add_u32, blocks(d.n_tokens), dim3(256), 0, st, d.tok_start, d.n_tokens, (u32)base_d);
        if (base_d) hipLaunchKernelGGL(k_add_u64, blocks(Rs + 1), dim3(256), 0, st, d.denoised_offsets, Rs + 1, base_d);
        if (base_t) hipLaunchKernelGGL(k_add_u64, blocks(Rs + 1), dim3(256), 0, st, d.tok_offsets, Rs + 1, base_t);
        if (base_c) hipLaunchKernelGGL(k_add_u64, blocks(Rs + 1), dim3(256), 0, st, d.chunk_offsets, Rs + 1, base_c);
        if (r0 && d.n_chunks)
            hipLaunchKernelGGL(k_add_chunk_record, blocks(d.n_chunks), dim3(256), 0, st, reinterpret_cast<ChunkDesc*>(d.chunks),
                               d.n_chunks, (u32)r0);
        auto back = [&](void* dst, const void* src, size_t n) -> int {
            if (n) HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st));
            return TRP_OK;
        };
        if (want_den && (rc = back(out->denoised + base_d, d.denoised, d.n_denoised))) return fail_free(rc);
        if (want_tok && (rc = back(out->tok_start + base_t, d.tok_start, d.n_tokens * sizeof(u32)))) return fail_free(rc);
        if (want_tok && (rc = back(out->tok_len + base_t, d.tok_len, d.n_tokens * sizeof(u32)))) return fail_free(rc);
        if (want_ch && (rc = back(out->chunks + base_c, d.chunks, d.n_chunks * sizeof(trp_chunk)))) return fail_free(rc);
        if (want_ct && (rc = back(ctext->text + base_x, d_ct, n_ct))) return fail_free(rc);
        if (want_ct && (rc = back(ctext->offsets + base_c, d_cto, (d.n_chunks + 1) * sizeof(u64)))) return fail_free(rc);
        if ((rc = back(out->denoised_offsets + r0, d.denoised_offsets, (Rs + 1) * sizeof(u64)))) return fail_free(rc);
        if ((rc = back(out->tok_offsets + r0, d.tok_offsets, (Rs + 1) * sizeof(u64)))) return fail_free(rc);
        if ((rc = back(out->chunk_offsets + r0, d.chunk_offsets, (Rs + 1) * sizeof(u64)))) return fail_free(rc);
        if ((rc = back(out->dropped + r0, d.dropped, Rs))) return fail_free(rc);
        base_d += d.n_denoised;
        base_t += d.n_tokens;
        base_c += d.n_chunks;
        base_x += n_ct;
    }
    if ((rc = sync_all())) return fail_free(rc);
    if (want_ct) ctext->n_bytes = base_x;
    out->n_records = R;
    out->n_denoised = base_d;
    out->n_tokens = base_t;
    out->n_chunks = base_c;
    return TRP_OK;
}

// A small batch from host memory onto the device and through the path in ONE launch: text and offsets are put
// into the context's page-locked block, the kernel fetches them from there (no copy commands in front of it) and,
// with `gather`, leaves the selected result arrays in the other page-locked block (no copy commands behind it
// either).  *gathered tells whether it did: not when the batch had to be run again by the kernel sequence or the
// arrays exceed the block; the device arrays in *d are valid either way.  Call only when small_host_ok() holds.
static bool small_host_ok(const trp_ctx* c, u64 N, u64 R) {
    return small_batch_ok(c, N, R, SMALL_HOST_MAX_TILES) && !c->summary_path;
}
static int run_small_host(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, uint64_t N,
                          const DevCfg& dc, bool gather, uint32_t outputs, trp_result* d, bool* gathered) {
    int rc;
    *gathered = false;
    if (!c->h_in) HIPCHK(hipHostMalloc((void**)&c->h_in, SMALL_IN_CAP, hipHostMallocDefault));
    if (gather && !c->h_out) HIPCHK(hipHostMalloc((void**)&c->h_out, H_OUT_BYTES, hipHostMallocDefault));
    if ((rc = c->in_bytes.ensure(N + 64))) return rc;
    if ((rc = c->in_offsets.ensure((R + 1) * sizeof(u64)))) return rc;
    if (gather && (outputs & TRP_OUT_CHUNK_TEXT)) {
        if ((rc = c->text_len.ensure((RENDER_SMALL_MAX + 1) * sizeof(u64)))) return rc;
        if ((rc = c->text_off.ensure((RENDER_SMALL_MAX + 2) * sizeof(u64)))) return rc;
        if ((rc = c->text.ensure(((u64)4 << 20) + 16))) return rc;
    }
    if (N) memcpy(c->h_in, bytes, N);
    memcpy(c->h_in + SMALL_HOST_MAX_TILES * TILE + 64, offsets, (R + 1) * sizeof(u64));
    c->h_totals[10] = 0;
    c->h_totals[12] = 0;
    c->small_io.on = true;
    c->small_io.gather = gather;
    c->small_io.outputs = outputs;
    rc = run_device(c, c->in_bytes.as<u8>(), c->in_offsets.as<u64>(), R, N, dc, d);
    c->small_io = trp_ctx::SmallIO();
    if (rc) return rc;
    *gathered = gather && c->h_totals[10] != 0;
    return TRP_OK;
}

static int run_host(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const DevCfg& dc,
                    trp_result* out, uint32_t outputs = TRP_OUT_ALL, trp_chunk_text* ctext = nullptr,
                    const trp_denoiser* more = nullptr, uint32_t n_more = 0) {
    if (!c || !out || !offsets) return fail(TRP_E_ARG, "null argument");
    if ((outputs & TRP_OUT_CHUNK_TEXT) && !ctext) return fail(TRP_E_ARG, "chunk text requested without a place for it");
    if (ctext) memset(ctext, 0, sizeof(*ctext));
    if (offsets[0] != 0) return fail(TRP_E_ARG, "offsets[0] must be 0");
    const u64 N = offsets[R];
    if (N && !bytes) return fail(TRP_E_ARG, "bytes is null");
    HIPCHK(hipSetDevice(c->device));
    if (N > TRP_MAX_BATCH_BYTES || R >= 0xFFFFFFFEull) return fail(TRP_E_TOO_LARGE, "batch too large: split it by record range");
    for (u64 r = 0; r < R; r++)
        if (offsets[r + 1] < offsets[r]) return fail(TRP_E_ARG, "offsets must be non-decreasing");
    const char* no_pipe = getenv("TRP_NO_PIPELINE");
    if (N >= PIPE_MIN_BYTES && R >= 2 && dc.denoise_enabled && !(no_pipe && atoi(no_pipe)) && n_more == 0)
        return run_host_pipelined(c, bytes, offsets, R, dc, outputs, out, ctext);
    int rc;
    trp_result d;
    bool gathered = false;
    if (n_more) {   // a fold of several denoisers: the stages run on the device one after the other
        if ((rc = c->in_bytes.ensure(N + 64))) return rc;
        if ((rc = c->in_offsets.ensure((R + 1) * sizeof(u64)))) return rc;
        if (N) HIPCHK(hipMemcpyAsync(c->in_bytes.p, bytes, N, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->in_offsets.p, offsets, (R + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
        if ((rc = run_device_fold(c, c->in_bytes.as<u8>(), c->in_offsets.as<u64>(), R, N, dc, more, n_more, &d))) return rc;
    } else if (small_host_ok(c, N, R)) {
        if ((rc = run_small_host(c, bytes, offsets, R, N, dc, true, outputs, &d, &gathered))) return rc;
    } else {
        if ((rc = c->in_bytes.ensure(N + 64))) return rc;
        if ((rc = c->in_offsets.ensure((R + 1) * sizeof(u64)))) return rc;
        if (N) HIPCHK(hipMemcpyAsync(c->in_bytes.p, bytes, N, hipMemcpyHostToDevice, c->stream));
        if ((R + 1) * sizeof(u64) <= SMALL_IN_CAP) {
            // a short offsets array through the page-locked block: a copy command that needs no staging
            if (!c->h_in) HIPCHK(hipHostMalloc((void**)&c->h_in, SMALL_IN_CAP, hipHostMallocDefault));
            memcpy(c->h_in, offsets, (R + 1) * sizeof(u64));
            HIPCHK(hipMemcpyAsync(c->in_offsets.p, c->h_in, (R + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
        } else {
            HIPCHK(hipMemcpyAsync(c->in_offsets.p, offsets, (R + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
        }
        if ((rc = run_device(c, c->in_bytes.as<u8>(), c->in_offsets.as<u64>(), R, N, dc, &d))) return rc;
    }
    memset(out, 0, sizeof(*out));
    out->n_records = R; out->n_denoised = d.n_denoised; out->n_tokens = d.n_tokens; out->n_chunks = d.n_chunks;
    const PackLayout L = pack_layout(R, d.n_denoised, d.n_tokens, d.n_chunks, outputs);
    // one result array: out of the gathered block when the kernel has filled it (`at` = its place there), else by a
    // copy command from the device array
    // (arrays too short for a page-locked result block, when the single launch has not gathered them already, are
    // packed into the context's page-locked block by one more launch and copied from there after the synchronise)
    PackArgs pk;
    pk.n = 0;
    void* pk_dst[PACK_MAX];
    u64 pk_used = 0;
    auto grab = [&](void** dst, const void* src, size_t bytes_, u64 at = ~(u64)0) -> int {
        *dst = host_pool().alloc(bytes_);
        if (!*dst) return fail(TRP_E_ARG, "out of host memory");
        if (bytes_ && gathered && at != ~(u64)0) {
            memcpy(*dst, c->h_out + at, bytes_);
        } else if (bytes_ && bytes_ < HostPool::PIN_MIN && pk.n < PACK_MAX && pk_used + bytes_ + 16 <= SMALL_OUT_CAP &&
                   ((uintptr_t)src & 15u) == 0) {
            pk.src[pk.n] = src; pk.bytes[pk.n] = bytes_; pk.at[pk.n] = pk_used; pk_dst[pk.n] = *dst;
            pk.n++;
            pk_used += (bytes_ + 15) & ~(u64)15;
        } else if (bytes_) {
            hipError_t e = hipMemcpyAsync(*dst, src, bytes_, hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) return fail(TRP_E_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
        }
        return TRP_OK;
    };
    // on failure: wait for the copies already enqueued, then hand the arrays taken so far back to the pool
    auto fail_free = [&](int code) {
        (void)hipStreamSynchronize(c->stream);
        trp_result_free(out);
        if (ctext) { host_pool().release(ctext->text); host_pool().release(ctext->offsets); memset(ctext, 0, sizeof(*ctext)); }
        return code;
    };
    if ((outputs & TRP_OUT_TEXT) && (rc = grab((void**)&out->denoised, d.denoised, d.n_denoised, L.denoised))) return fail_free(rc);
    if ((rc = grab((void**)&out->denoised_offsets, d.denoised_offsets, (R + 1) * sizeof(u64), L.den_off))) return fail_free(rc);
    if ((rc = grab((void**)&out->dropped, d.dropped, R, L.dropped))) return fail_free(rc);
    if ((outputs & TRP_OUT_TOKENS) && (rc = grab((void**)&out->tok_start, d.tok_start, d.n_tokens * sizeof(u32), L.tok_start))) return fail_free(rc);
    if ((outputs & TRP_OUT_TOKENS) && (rc = grab((void**)&out->tok_len, d.tok_len, d.n_tokens * sizeof(u32), L.tok_len))) return fail_free(rc);
    if ((rc = grab((void**)&out->tok_offsets, d.tok_offsets, (R + 1) * sizeof(u64), L.tok_off))) return fail_free(rc);
    if ((outputs & TRP_OUT_CHUNKS) && (rc = grab((void**)&out->chunks, d.chunks, d.n_chunks * sizeof(trp_chunk), L.chunks))) return fail_free(rc);
    if ((rc = grab((void**)&out->chunk_offsets, d.chunk_offsets, (R + 1) * sizeof(u64), L.chunk_off))) return fail_free(rc);
    if (outputs & TRP_OUT_CHUNK_TEXT) {
        uint8_t* d_ct = nullptr;
        uint64_t* d_cto = nullptr;
        uint64_t n_ct = 0;
        const u64 nb_off0 = (d.n_chunks + 1) * sizeof(u64);
        if (gathered && c->h_totals[12] != 0) {
            // the single launch has rendered the texts as well: they lie behind the gathered arrays
            n_ct = c->h_totals[13];
            ctext->text = (uint8_t*)host_pool().alloc(n_ct ? n_ct : 1);
            ctext->offsets = (uint64_t*)host_pool().alloc(nb_off0);
            if (!ctext->text || !ctext->offsets) return fail_free(fail(TRP_E_ARG, "out of host memory"));
            const u8* hp = c->h_out + L.total;
            memcpy(ctext->offsets, hp, nb_off0);
            if (n_ct) memcpy(ctext->text, hp + ((nb_off0 + 15) & ~(u64)15), n_ct);
            ctext->n_bytes = n_ct;
        } else {
        c->want_host_pack = true;
        rc = render_text(c, reinterpret_cast<const ChunkDesc*>(d.chunks), d.n_chunks, d.chunk_offsets, &d_ct, &d_cto, &n_ct);
        c->want_host_pack = false;
        if (rc) return fail_free(rc);
        if (c->text_host_packed) {   // both arrays lie in the page-locked block already
            c->text_host_packed = false;
            const u64 nb_off = (d.n_chunks + 1) * sizeof(u64);
            ctext->text = (uint8_t*)host_pool().alloc(n_ct ? n_ct : 1);
            ctext->offsets = (uint64_t*)host_pool().alloc(nb_off);
            if (!ctext->text || !ctext->offsets) return fail_free(fail(TRP_E_ARG, "out of host memory"));
            const u8* hp = c->h_out + SMALL_OUT_CAP;
            memcpy(ctext->offsets, hp, nb_off);
            if (n_ct) memcpy(ctext->text, hp + ((nb_off + 15) & ~(u64)15), n_ct);
        } else {
            if ((rc = grab((void**)&ctext->text, d_ct, n_ct))) return fail_free(rc);
            if ((rc = grab((void**)&ctext->offsets, d_cto, (d.n_chunks + 1) * sizeof(u64)))) return fail_free(rc);
        }
        ctext->n_bytes = n_ct;
        }
    }
    if (pk.n) {
        if (!c->h_out) {
            hipError_t e = hipHostMalloc((void**)&c->h_out, H_OUT_BYTES, hipHostMallocDefault);
            if (e != hipSuccess) return fail_free(fail(TRP_E_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)));
        }
        pk.dst = c->h_out;
        hipLaunchKernelGGL(k_pack_small, dim3((unsigned)pk.n), dim3(NT), 0, c->stream, pk);
    }
    {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail_free(fail(TRP_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)));
    }
    for (int i = 0; i < pk.n; i++) memcpy(pk_dst[i], c->h_out + pk.at[i], pk.bytes[i]);
    return TRP_OK;
}

int trp_preprocess_batch(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const trp_config* cfg,
                         trp_result* out) {
    DevCfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    return run_host(c, bytes, offsets, R, dc, out);
}

int trp_preprocess_batch_select(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R,
                                const trp_config* cfg, uint32_t outputs, trp_result* out, trp_chunk_text* chunk_text) {
    DevCfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    if (outputs & ~(TRP_OUT_ALL | TRP_OUT_CHUNK_TEXT)) return fail(TRP_E_ARG, "unknown output selector");
    return run_host(c, bytes, offsets, R, dc, out, outputs, chunk_text);
}

int trp_preprocess_batch_fold(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const trp_config* cfg,
                              const trp_denoiser* more, uint32_t n_more, uint32_t outputs, trp_result* out,
                              trp_chunk_text* chunk_text) {
    DevCfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    if (n_more && !more) return fail(TRP_E_ARG, "more is null");
    if (outputs & ~(TRP_OUT_ALL | TRP_OUT_CHUNK_TEXT)) return fail(TRP_E_ARG, "unknown output selector");
    return run_host(c, bytes, offsets, R, dc, out, outputs, chunk_text, more, n_more);
}

int trp_preprocess_batch_device_fold(trp_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R, uint64_t N,
                                     const trp_config* cfg, const trp_denoiser* more, uint32_t n_more, trp_result* out) {
    DevCfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    return run_device_fold(c, d_bytes, d_offsets, R, N, dc, more, n_more, out);
}

// ---- the steps after the path over host memory ------------------------------------------------------------
// A packed host batch onto the device, run as one device batch (no pipelining: the text renderers work on the
// last batch of the context, which must be the whole batch).
static int stage_and_run(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const DevCfg& dc,
                         trp_result* d) {
    if (!c || !offsets) return fail(TRP_E_ARG, "null argument");
    if (offsets[0] != 0) return fail(TRP_E_ARG, "offsets[0] must be 0");
    const u64 N = offsets[R];
    if (N && !bytes) return fail(TRP_E_ARG, "bytes is null");
    if (N > TRP_MAX_BATCH_BYTES || R >= 0xFFFFFFFEull) return fail(TRP_E_TOO_LARGE, "batch too large: split it by record range");
    for (u64 r = 0; r < R; r++)
        if (offsets[r + 1] < offsets[r]) return fail(TRP_E_ARG, "offsets must be non-decreasing");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if (small_host_ok(c, N, R)) {
        bool gathered;
        return run_small_host(c, bytes, offsets, R, N, dc, false, 0, d, &gathered);
    }
    if ((rc = c->in_bytes.ensure(N + 64))) return rc;
    if ((rc = c->in_offsets.ensure((R + 1) * sizeof(u64)))) return rc;
    if (N) HIPCHK(hipMemcpyAsync(c->in_bytes.p, bytes, N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->in_offsets.p, offsets, (R + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    return run_device(c, c->in_bytes.as<u8>(), c->in_offsets.as<u64>(), R, N, dc, d);
}
// Rendered texts (device) -> library-owned host arrays.  Short arrays travel through the context's page-locked block
// by one launch (a copy command to pageable memory costs 20 us), long ones by copy commands into page-locked arrays.
static int texts_to_host(trp_ctx* c, const uint8_t* d_text, const uint64_t* d_off, uint64_t n, uint64_t nbytes,
                         trp_chunk_text* out) {
    memset(out, 0, sizeof(*out));
    out->offsets = (uint64_t*)host_pool().alloc((n + 1) * sizeof(u64));
    out->text = (uint8_t*)host_pool().alloc(nbytes ? nbytes : 1);
    if (!out->offsets || !out->text) { trp_chunk_text_free(out); return fail(TRP_E_ARG, "out of host memory"); }
    if (c->text_host_packed) {   // the renderer has left both arrays in the page-locked block already
        c->text_host_packed = false;
        const u8* hp = c->h_out + SMALL_OUT_CAP;
        memcpy(out->offsets, hp, (n + 1) * sizeof(u64));
        if (nbytes) memcpy(out->text, hp + (((n + 1) * sizeof(u64) + 15) & ~(u64)15), nbytes);
        out->n_bytes = nbytes;
        return TRP_OK;
    }
    PackArgs pk;
    pk.n = 0;
    void* pk_dst[2];
    u64 used = 0;
    hipError_t e = hipSuccess;
    auto fetch = [&](void* dst, const void* src, u64 bytes_) {
        if (!bytes_ || e != hipSuccess) return;
        if (bytes_ < HostPool::PIN_MIN && used + bytes_ + 16 <= SMALL_OUT_CAP && ((uintptr_t)src & 15u) == 0) {
            pk.src[pk.n] = src; pk.bytes[pk.n] = bytes_; pk.at[pk.n] = used; pk_dst[pk.n] = dst;
            pk.n++;
            used += (bytes_ + 15) & ~(u64)15;
        } else {
            e = hipMemcpyAsync(dst, src, bytes_, hipMemcpyDeviceToHost, c->stream);
        }
    };
    fetch(out->offsets, d_off, (n + 1) * sizeof(u64));
    fetch(out->text, d_text, nbytes);
    if (e == hipSuccess && pk.n) {
        if (!c->h_out) e = hipHostMalloc((void**)&c->h_out, H_OUT_BYTES, hipHostMallocDefault);
        if (e == hipSuccess) {
            pk.dst = c->h_out;
            hipLaunchKernelGGL(k_pack_small, dim3((unsigned)pk.n), dim3(NT), 0, c->stream, pk);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { trp_chunk_text_free(out); return fail(TRP_E_HIP, std::string("copy of rendered texts: ") + hipGetErrorString(e)); }
    for (int i = 0; i < pk.n; i++) memcpy(pk_dst[i], c->h_out + pk.at[i], pk.bytes[i]);
    out->n_bytes = nbytes;
    return TRP_OK;
}

int trp_bm25_texts(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const uint64_t* doc_offsets,
                   uint64_t n_docs, uint32_t max_tokens, const char* newline, trp_chunk_text* out) {
    if (!out) return fail(TRP_E_ARG, "out is null");
    memset(out, 0, sizeof(*out));
    DevCfg dc = normalize_devcfg();   // the reference's default chunking; the text itself is passed through untouched
    dc.denoise_enabled = 0;
    dc.normalize = 0;
    trp_result d;
    int rc = stage_and_run(c, bytes, offsets, R, dc, &d);
    if (rc) return rc;
    uint8_t* d_text = nullptr;
    uint64_t* d_off = nullptr;
    uint64_t nbytes = 0;
    c->text_host_packed = false;
    if ((rc = trp_bm25_texts_device(c, doc_offsets, n_docs, max_tokens, newline, &d_text, &d_off, &nbytes))) return rc;
    return texts_to_host(c, d_text, d_off, n_docs, nbytes, out);
}

int trp_head_tokens_text(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, uint32_t max_tokens,
                         trp_chunk_text* out) {
    if (!out) return fail(TRP_E_ARG, "out is null");
    memset(out, 0, sizeof(*out));
    DevCfg dc = normalize_devcfg();
    dc.denoise_enabled = 0;
    dc.normalize = 0;
    trp_result d;
    int rc = stage_and_run(c, bytes, offsets, R, dc, &d);
    if (rc) return rc;
    uint8_t* d_text = nullptr;
    uint64_t* d_off = nullptr;
    uint64_t nbytes = 0;
    c->want_host_pack = true;
    rc = trp_head_tokens_text_device(c, max_tokens, &d_text, &d_off, &nbytes);
    c->want_host_pack = false;
    if (rc) return rc;
    return texts_to_host(c, d_text, d_off, R, nbytes, out);
}

int trp_chunk_bodies(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const trp_config* cfg,
                     const uint32_t* chunk_index, const uint32_t* budget, uint64_t n, trp_chunk_text* out) {
    if (!out) return fail(TRP_E_ARG, "out is null");
    memset(out, 0, sizeof(*out));
    DevCfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    trp_result d;
    if ((rc = stage_and_run(c, bytes, offsets, R, dc, &d))) return rc;
    uint8_t* d_text = nullptr;
    uint64_t* d_off = nullptr;
    uint64_t nbytes = 0;
    c->want_host_pack = true;
    rc = trp_chunk_bodies_device(c, chunk_index, budget, n, &d_text, &d_off, &nbytes);
    c->want_host_pack = false;
    if (rc) return rc;
    return texts_to_host(c, d_text, d_off, n, nbytes, out);
}

void trp_chunk_text_free(trp_chunk_text* t) {
    if (!t) return;
    host_pool().release(t->text);
    host_pool().release(t->offsets);
    memset(t, 0, sizeof(*t));
}

int trp_normalize_batch(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, trp_result* out) {
    return run_host(c, bytes, offsets, R, normalize_devcfg(), out);
}

int trp_normalize_inline_whitespace(trp_ctx* c, const uint8_t* text, size_t len, uint8_t** out, size_t* out_len) {
    if (!out || !out_len) return fail(TRP_E_ARG, "null argument");
    uint64_t offs[2] = {0, (uint64_t)len};
    trp_result r;
    int rc = trp_normalize_batch(c, text, offs, 1, &r);
    if (rc) return rc;
    *out_len = (size_t)r.n_denoised;
    *out = r.denoised;
    r.denoised = nullptr;
    trp_result_free(&r);
    return TRP_OK;
}

static int one_record(trp_ctx* c, const uint8_t* text, size_t len, const trp_config* cfg, trp_result* out) {
    uint64_t offs[2] = {0, (uint64_t)len};
    return trp_preprocess_batch(c, text, offs, 1, cfg, out);
}

int trp_denoise(trp_ctx* c, const uint8_t* text, size_t len, const trp_config* cfg, uint8_t** out, size_t* out_len) {
    if (!out || !out_len) return fail(TRP_E_ARG, "null argument");
    trp_result r;
    int rc = one_record(c, text, len, cfg, &r);
    if (rc) return rc;
    int some = r.dropped[0] ? 0 : 1;
    *out = nullptr;
    *out_len = 0;
    if (some) {
        *out_len = (size_t)r.n_denoised;
        *out = r.denoised;
        r.denoised = nullptr;
    }
    trp_result_free(&r);
    return some;
}

int trp_tokenize(trp_ctx* c, const uint8_t* text, size_t len, uint32_t** starts, uint32_t** lens, size_t* n_tokens) {
    if (!starts || !lens || !n_tokens) return fail(TRP_E_ARG, "null argument");
    uint32_t none = 0;
    trp_config cfg = {0, 1, 0.35f, 1024, &none, 0, 0};
    trp_result r;
    int rc = one_record(c, text, len, &cfg, &r);
    if (rc) return rc;
    *starts = r.tok_start; *lens = r.tok_len; *n_tokens = (size_t)r.n_tokens;
    r.tok_start = nullptr; r.tok_len = nullptr;
    trp_result_free(&r);
    return TRP_OK;
}

int trp_token_count(trp_ctx* c, const uint8_t* text, size_t len, size_t* n_tokens) {
    uint32_t *s = nullptr, *l = nullptr;
    int rc = trp_tokenize(c, text, len, &s, &l, n_tokens);
    trp_free(s); trp_free(l);
    return rc;
}

int trp_chunk_section(trp_ctx* c, const uint8_t* text, size_t len, const trp_config* cfg, trp_result* out) {
    return one_record(c, text, len, cfg, out);
}

}  // extern "C"
