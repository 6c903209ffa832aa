// HIP kernels (gfx950) and C ABI of the denoise -> tokenise -> chunk path.
//
// Data layout in HBM (one batch = N bytes, R records, nB = ceil(N/64) blocks, nT = ceil(N/16384) tiles)
//   bytes[N], offsets[R+1]        caller's packed corpus
//   rec_lo[nT+2]                  first record that starts at/after each tile start (+ one for N)
//   SA[nT]                        per-tile forward summary (struct J) written by k_summary
//   plan[nB] (4 x u64)            emit plan bit planes (struct Plan) written by k_plan / k_gate3
//   g3m[nB]                       first bytes of the digit-heavy lines k_gate3 has to solve
//   tile_bytes[nT], tile_toks[nT], tile_flags[nT]   emitted bytes / token starts per tile
//   tile_base[nT+1]               exclusive prefix of (bytes | toks<<32)
//   denoised[D], tok_start[T], tok_len[T], *_offsets[R+1], chunks[C]   results
//
// Kernel sequence with the denoiser enabled:
//   k_rec_lo -> k_summary -> k_plan -> k_gate3 -> k_pack_tiles -> scan(3) -> k_emit -> k_finish
//   -> k_chunk_count -> scan(3) -> k_chunk_emit
// and with it disabled: k_rec_lo -> k_count_tokens -> k_pack_tiles -> scan(3) -> k_emit<passthrough> ...
//
// The per-lane logic lives in trp_lane.h (shared with the host simulator).
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "triplets_amd.h"
#include "trp_lane.h"

#define TRP_TABLE_QUAL static __device__ const
#include "unicode_tables.inc"

using namespace trp;

namespace {

constexpr int NT = TILE_BLOCKS;   // threads per workgroup

__device__ inline UniTables dev_tables() {
    UniTables ut = {trp_alpha_stage1, trp_alpha_stage2, TRP_ALPHA_STAGE1_LEN};
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

// Exclusive backward scan: thread i receives op(v[i+1], ..., v[NT-1], seed).
template <class T, class Op>
__device__ inline T block_scan_excl_bwd(T v, Op op, T identity, T seed, T* smem) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        T o = shfl_down_pod(inc, off);
        if (lane + off < 64) inc = op(inc, o);
    }
    if (lane == 0) smem[wave] = inc;
    __syncthreads();
    T post = seed;
    for (int w = NT / 64 - 1; w > wave; w--) post = op(smem[w], post);
    T ex = shfl_down_pod(inc, 1);
    if (lane == 63) ex = identity;
    T res = op(ex, post);
    __syncthreads();
    return res;
}

struct JOp { __device__ J operator()(const J& a, const J& b) const { return j_combine(a, b); } };
struct SOp { __device__ S operator()(const S& a, const S& b) const { return s_combine(a, b); } };

// ---------------------------------------------------------------------------------------------------
// tile staging
// ---------------------------------------------------------------------------------------------------
struct TileShared {
    alignas(16) u8 bytes[HALO + TILE + HALO];
    u64 rsw[TILE_BLOCKS + 1];
};

// Copy the tile [tS, tS+TILE) plus HALO bytes on both sides into LDS (zeros outside the corpus).
__device__ inline void stage_tile(const u8* __restrict__ g, u64 N, u64 tS, u8* sb) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < TILE / 16 / NT; k++) {
        int c = k * NT + tid;
        u64 q = tS + (u64)c * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q + 16 <= N) v = *reinterpret_cast<const uint4*>(g + q);
        else if (q < N) {
            u8 tmp[16];
            for (int i = 0; i < 16; i++) tmp[i] = (q + i < N) ? g[q + i] : 0;
            v = *reinterpret_cast<uint4*>(tmp);
        }
        *reinterpret_cast<uint4*>(sb + HALO + c * 16) = v;
    }
    if (tid < 2) {
        u64 q = tid == 0 ? tS - HALO : tS + TILE;
        uint4 v = make_uint4(0, 0, 0, 0);
        bool ok = tid == 0 ? (tS >= HALO) : true;
        if (ok && q + 16 <= N) v = *reinterpret_cast<const uint4*>(g + q);
        else if (ok && q < N) {
            u8 tmp[16];
            for (int i = 0; i < 16; i++) tmp[i] = (q + i < N) ? g[q + i] : 0;
            v = *reinterpret_cast<uint4*>(tmp);
        }
        *reinterpret_cast<uint4*>(sb + (tid == 0 ? 0 : HALO + TILE)) = v;
    }
}

// Record-start bit masks of the tile's blocks; word TILE_BLOCKS bit 0 tells whether the byte right
// after the tile starts a record.
__device__ inline void stage_record_starts(const u64* __restrict__ offsets, const u32* __restrict__ rec_lo,
                                           u64 R, u64 N, u64 t, u64* rsw) {
    const int tid = threadIdx.x;
    const u64 tS = t * (u64)TILE;
    rsw[tid] = 0;
    if (tid == 0) rsw[TILE_BLOCKS] = 0;
    __syncthreads();
    u32 r0 = rec_lo[t], r1 = rec_lo[t + 1];
    for (u64 r = (u64)r0 + tid; r < r1; r += NT) {
        u64 p = offsets[r];
        if (p < N) atomicOr(reinterpret_cast<unsigned long long*>(&rsw[(p - tS) >> 6]), 1ull << ((p - tS) & 63));
    }
    if (tid == 0) {
        u64 tE = tS + TILE;
        if (tE < N && r1 < R && offsets[r1] == tE) rsw[TILE_BLOCKS] = 1;
    }
    __syncthreads();
}

struct LaneCtx {
    Cls c;
    Derived d;
    u64 p;          // byte offset of the block in the corpus
    int nvalid;
    bool next_rs;   // the byte after the block starts a record or does not exist
};

__device__ inline LaneCtx lane_setup(const TileShared& sh, u64 N, u64 tS, bool want_lines) {
    LaneCtx L;
    const int tid = threadIdx.x;
    const u8* blk = sh.bytes + HALO + tid * BLOCK;
    L.p = tS + (u64)tid * BLOCK;
    L.nvalid = L.p >= N ? 0 : (int)((N - L.p) < (u64)BLOCK ? (N - L.p) : (u64)BLOCK);
    L.c = classify_block(blk, L.nvalid, dev_tables());
    u64 V = below(L.nvalid);
    u64 RS = sh.rsw[tid] & V;
    bool nrs = (sh.rsw[tid + 1] & 1ull) != 0;
    bool has_next = L.p + BLOCK < N;
    L.next_rs = !has_next || nrs;
    if (want_lines) {
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
__global__ void k_rec_lo(const u64* __restrict__ offsets, u64 R, u64 N, u64 nT, u32* __restrict__ rec_lo) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > nT + 1) return;
    u64 key = t <= nT ? t * (u64)TILE : N;
    u64 lo = 0, hi = R;
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if (offsets[mid] < key) lo = mid + 1; else hi = mid;
    }
    rec_lo[t] = (u32)lo;
}

// ---------------------------------------------------------------------------------------------------
// k_summary: per-tile forward summary (line statistics at the tile edges, record-level alpha flags).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_summary(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                                 const u32* __restrict__ rec_lo, u64 R, u64 N,
                                                 J* __restrict__ SA) {
    __shared__ TileShared sh;
    __shared__ J wsum[NT / 64];
    const u64 t = blockIdx.x, tS = t * (u64)TILE;
    stage_tile(bytes, N, tS, sh.bytes);
    stage_record_starts(offsets, rec_lo, R, N, t, sh.rsw);
    LaneCtx L = lane_setup(sh, N, tS, true);
    J j = lane_J(L.c, L.d);
    j = wave_reduce_ordered(j, JOp());
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = j;
    __syncthreads();
    if (threadIdx.x == 0) {
        J acc = wsum[0];
        for (int w = 1; w < NT / 64; w++) acc = j_combine(acc, wsum[w]);
        SA[t] = acc;
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

struct PlanShared {
    TileShared tile;
    J jscan[NT / 64];
    S sscan[NT / 64];
    J seedJ;
    S seedS;
    u64 tokc[TILE_BLOCKS];
    u32 red[2][NT / 64];
};

__global__ __launch_bounds__(NT) void k_plan(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                              const u32* __restrict__ rec_lo, u64 R, u64 N, u64 nT,
                                              const J* __restrict__ SA, DevCfg cfg, Plan* __restrict__ plan,
                                              u64* __restrict__ g3m, u32* __restrict__ tile_bytes,
                                              u32* __restrict__ tile_toks, u32* __restrict__ tile_flags) {
    __shared__ PlanShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 t = blockIdx.x, tS = t * (u64)TILE;
    stage_tile(bytes, N, tS, sh.tile.bytes);
    if (wave == 0) {
        J P = tile_lookback(SA, t);
        S X = tile_lookahead(SA, t, nT);
        if (lane == 0) { sh.seedJ = P; sh.seedS = X; }
    }
    stage_record_starts(offsets, rec_lo, R, N, t, sh.tile.rsw);   // contains barriers
    LaneCtx L = lane_setup(sh.tile, N, tS, true);
    J j = lane_J(L.c, L.d);
    J P = block_scan_excl_fwd(j, JOp(), j_identity(), sh.seedJ, sh.jscan, (J*)nullptr);
    S X = block_scan_excl_bwd(s_from_J(j), SOp(), s_identity(), sh.seedS, sh.sscan);
    LaneCarry k = carry_from(P, X, L.next_rs);
    Plan pl = plan_block(L.c, L.d, k, cfg);
    const u64 b = t * (u64)TILE_BLOCKS + tid;
    // Lines still to be solved (gate 3) are listed in a plane of their own: k_gate3 finds its work
    // there while other lanes of that kernel are already writing token bits into the plan.
    const u64 marks = pl.TOK & ~pl.C;
    pl.TOK &= pl.C;
    if (L.nvalid > 0) { plan[b] = pl; g3m[b] = marks; }
    u64 TOKC = pl.TOK;
    u64 PFX = (pl.PSP | pl.PNL) & pl.C;
    sh.tokc[tid] = TOKC;
    __syncthreads();
    bool prev_tok = tid > 0 && (sh.tokc[tid - 1] >> 63);
    u64 TS = token_starts(TOKC, L.d.RS, prev_tok);
    u32 nb = popc(pl.C) + popc(PFX), ntk = popc(TS);
    // workgroup sums
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nb += __shfl_down((int)nb, off, 64);
        ntk += __shfl_down((int)ntk, off, 64);
    }
    if (lane == 0) { sh.red[0][wave] = nb; sh.red[1][wave] = ntk; }
    __syncthreads();
    if (tid == 0) {
        u32 sb = 0, st = 0;
        for (int w = 0; w < NT / 64; w++) { sb += sh.red[0][w]; st += sh.red[1][w]; }
        tile_bytes[t] = sb;
        tile_toks[t] = st;
        u32 f = 0;
        if ((sh.tokc[0] & 1ull) && !(L.d.LS & 1ull)) f |= TF_FIRST_CONT;   // thread 0's own block
        if (sh.tokc[TILE_BLOCKS - 1] >> 63) f |= TF_LAST_TOK;
        tile_flags[t] = f;
    }
}

// Pass-through configuration: token starts per tile (no denoising, the text stays in place).
__global__ __launch_bounds__(NT) void k_count_tokens(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                                      const u32* __restrict__ rec_lo, u64 R, u64 N,
                                                      u32* __restrict__ tile_bytes, u32* __restrict__ tile_toks,
                                                      u32* __restrict__ tile_flags) {
    __shared__ TileShared sh;
    __shared__ u64 tokc[TILE_BLOCKS];
    __shared__ u32 red[NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 t = blockIdx.x, tS = t * (u64)TILE;
    stage_tile(bytes, N, tS, sh.bytes);
    stage_record_starts(offsets, rec_lo, R, N, t, sh.rsw);
    LaneCtx L = lane_setup(sh, N, tS, false);
    Plan pl = plan_passthrough(L.c, L.d.V);
    tokc[tid] = pl.TOK;
    __syncthreads();
    bool prev_tok = tid > 0 && (tokc[tid - 1] >> 63);
    u32 ntk = popc(token_starts(pl.TOK, L.d.RS, prev_tok));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ntk += __shfl_down((int)ntk, off, 64);
    if (lane == 0) red[wave] = ntk;
    __syncthreads();
    if (tid == 0) {
        u32 st = 0;
        for (int w = 0; w < NT / 64; w++) st += red[w];
        u64 nv = N - tS < (u64)TILE ? N - tS : (u64)TILE;
        tile_bytes[t] = (u32)nv;
        tile_toks[t] = st;
        u32 f = 0;
        if ((tokc[0] & 1ull) && !(L.d.RS & 1ull)) f |= TF_FIRST_CONT;
        if (tokc[TILE_BLOCKS - 1] >> 63) f |= TF_LAST_TOK;
        tile_flags[t] = f;
    }
}

// ---------------------------------------------------------------------------------------------------
// k_gate3: solve the digit-heavy lines that k_plan marked (one lane per marked line, serial walk).
// ---------------------------------------------------------------------------------------------------
__device__ inline void plan_or(Plan* plan, u64 word, u64 c, u64 tok, u64 psp, u64 pnl) {
    if (c) atomicOr(reinterpret_cast<unsigned long long*>(&plan[word].C), c);
    if (tok) atomicOr(reinterpret_cast<unsigned long long*>(&plan[word].TOK), tok);
    if (psp) atomicOr(reinterpret_cast<unsigned long long*>(&plan[word].PSP), psp);
    if (pnl) atomicOr(reinterpret_cast<unsigned long long*>(&plan[word].PNL), pnl);
}

__global__ __launch_bounds__(NT) void k_gate3(const u8* __restrict__ bytes, const u64* __restrict__ offsets, u64 R,
                                               u64 N, u64 nB, DevCfg cfg, Plan* plan,
                                               const u64* __restrict__ g3m, u32* tile_bytes, u32* tile_toks) {
    u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nB) return;
    u64 marks = g3m[b];
    if (!marks) return;
    // flag bits at a marked position are only ever changed by the lane that owns the mark
    u64 pspw = plan[b].PSP, pnlw = plan[b].PNL;
    UniTables ut = dev_tables();
    ByteSrc src = {bytes, N};
    while (marks) {
        int jb = ctz(marks);
        marks &= marks - 1;
        u64 p = b * 64 + jb;
        bool table = (pspw >> jb) & 1, rab = (pnlw >> jb) & 1;
        // end of the record that holds p: first offset > p
        u64 lo = 0, hi = R;
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if (offsets[mid + 1] <= p) lo = mid + 1; else hi = mid;
        }
        u64 rec_end = offsets[lo + 1];
        u64 e = p;
        while (e < rec_end) {
            u8 ch = bytes[e];
            if (ch == '\n' || ch == '\r') break;
            e++;
        }
        u64 K = g3_solve_K(src, p, e, table, cfg.max_digit_ratio, ut);
        // clear the marker, then write the kept tokens
        u64 mb = ~(1ull << jb);
        atomicAnd(reinterpret_cast<unsigned long long*>(&plan[b].PSP), mb);
        atomicAnd(reinterpret_cast<unsigned long long*>(&plan[b].PNL), mb);
        bool first = true;
        g3_walk(src, p, e, table, ut, K, ~0ull - 1, (u64*)nullptr, (u64*)nullptr,
                [&](const TokInfo& tk, bool keep) {
                    if (!keep) return;
                    u64 w0 = tk.ts >> 6, w1 = (tk.te - 1) >> 6;
                    for (u64 w = w0; w <= w1; w++) {
                        u64 lo_b = w == w0 ? (tk.ts & 63) : 0;
                        u64 hi_b = w == w1 ? ((tk.te - 1) & 63) + 1 : 64;
                        u64 m = below((int)hi_b) & ~below((int)lo_b);
                        u64 pre = (w == w0) ? (1ull << lo_b) : 0;
                        plan_or(plan, w, m, m, (pre && !first) ? pre : 0, (pre && first && rab) ? pre : 0);
                        atomicAdd(&tile_bytes[w / TILE_BLOCKS], (u32)popc(m));
                    }
                    if (!first || rab) atomicAdd(&tile_bytes[w0 / TILE_BLOCKS], 1u);
                    atomicAdd(&tile_toks[w0 / TILE_BLOCKS], 1u);
                    first = false;
                });
    }
}

// ---------------------------------------------------------------------------------------------------
// generic exclusive scan of u64 (3 kernels), out has n+1 entries
// ---------------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8;
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

// tile counts -> packed u64 (bytes | toks << 32); a token that continues across a tile edge was
// counted as a start by the right-hand tile: take it back here.
__global__ void k_pack_tiles(const u32* __restrict__ tile_bytes, const u32* __restrict__ tile_toks,
                             const u32* __restrict__ tile_flags, u64 nT, u64* __restrict__ packed) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nT) return;
    u32 tk = tile_toks[t];
    if (t > 0 && (tile_flags[t] & TF_FIRST_CONT) && (tile_flags[t - 1] & TF_LAST_TOK)) tk -= 1;
    packed[t] = (u64)tile_bytes[t] | ((u64)tk << 32);
}

// ---------------------------------------------------------------------------------------------------
// k_emit: stream compaction of the tile per its plan -> denoised bytes, token table, per-record offsets
// ---------------------------------------------------------------------------------------------------
struct EmitScan {
    u32 nb, ntk;
    int lts;   // tile-relative byte position of the last token start so far, -1 none
};
struct EmitOp {
    __device__ EmitScan operator()(const EmitScan& a, const EmitScan& b) const {
        EmitScan r;
        r.nb = a.nb + b.nb;
        r.ntk = a.ntk + b.ntk;
        r.lts = b.lts >= 0 ? b.lts : a.lts;
        return r;
    }
};

struct EmitShared {
    TileShared tile;
    alignas(16) u8 out[TILE + 48];
    u64 tokc[TILE_BLOCKS + 1];
    u64 cw[TILE_BLOCKS], pfx[TILE_BLOCKS], ts[TILE_BLOCKS];
    u32 eb[TILE_BLOCKS], tb[TILE_BLOCKS];
    EmitScan scan[NT / 64];
    EmitScan total;
    u64 cin_start;      // corpus position where the token that runs into this tile starts
    u32 flags;          // bit0: byte before the tile is a token byte of the same record, bit1: byte after is
};

// Position where the token covering byte tS-1 starts (the tile's first byte continues it).
template <bool PASSTHROUGH>
__device__ inline u64 find_token_start_before(const u8* __restrict__ bytes, u64 N, const Plan* __restrict__ plan,
                                              u64 tS, u64 rec_start) {
    if (PASSTHROUGH) {
        u64 q = tS;   // bytes [q, tS) are token bytes
        while (q > rec_start) {
            u8 w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            u64 c = q - 1;
            for (int i = -3; i <= 3; i++) {
                long long a = (long long)c + i;
                w[i + 3] = (a >= 0 && (u64)a < N) ? bytes[a] : 0;
            }
            if (byte_is_ws(w + 3)) break;
            q--;
        }
        return q;
    } else {
        u64 bb = tS / 64;
        while (bb > 0) {
            bb--;
            u64 w = plan[bb].TOK & plan[bb].C;
            u64 lo = bb * 64;
            u64 allow = rec_start > lo ? ~below((int)(rec_start - lo)) : ~0ull;   // bits at/after the record start
            u64 zero = ~w & allow;
            if (zero) return lo + top(zero) + 1;
            if (rec_start >= lo) return rec_start;
        }
        return rec_start;
    }
}

template <bool PASSTHROUGH>
__global__ __launch_bounds__(NT) void k_emit(const u8* __restrict__ bytes, const u64* __restrict__ offsets,
                                              const u32* __restrict__ rec_lo, u64 R, u64 N, u64 nT,
                                              const Plan* __restrict__ plan, const u64* __restrict__ tile_base,
                                              u8* __restrict__ denoised, u32* __restrict__ tok_start,
                                              u32* __restrict__ tok_len, u64* __restrict__ den_off,
                                              u64* __restrict__ tok_off) {
    __shared__ EmitShared sh;
    const int tid = threadIdx.x;
    const u64 t = blockIdx.x, tS = t * (u64)TILE;
    const u64 b = t * (u64)TILE_BLOCKS + tid;
    stage_tile(bytes, N, tS, sh.tile.bytes);
    stage_record_starts(offsets, rec_lo, R, N, t, sh.tile.rsw);
    const u8* blk = sh.tile.bytes + HALO + tid * BLOCK;
    const u64 p = tS + (u64)tid * BLOCK;
    const int nvalid = p >= N ? 0 : (int)((N - p) < (u64)BLOCK ? (N - p) : (u64)BLOCK);
    const u64 V = below(nvalid);
    const u64 RS = sh.tile.rsw[tid] & V;
    const bool next_rs = !(p + BLOCK < N) || (sh.tile.rsw[tid + 1] & 1ull);
    Plan pl = {0, 0, 0, 0};
    if (PASSTHROUGH) {
        Cls c = classify_block(blk, nvalid, dev_tables());
        pl = plan_passthrough(c, V);
    } else if (nvalid > 0) {
        pl = plan[b];
    }
    const u64 TOKC = pl.TOK & pl.C;
    const u64 PFX = (pl.PSP | pl.PNL) & pl.C;
    sh.tokc[tid] = TOKC;
    if (tid == 0) {
        // neighbours across the tile edges
        u32 f = 0;
        if (tS > 0) {
            bool pt = PASSTHROUGH ? !byte_is_ws(blk - 1) : (((plan[b - 1].TOK & plan[b - 1].C) >> 63) & 1);
            if (pt) f |= 1;
        }
        u64 tE = tS + TILE;
        if (tE < N) {
            bool nt = PASSTHROUGH ? !byte_is_ws(sh.tile.bytes + HALO + TILE)
                                  : ((plan[b + TILE_BLOCKS].TOK & plan[b + TILE_BLOCKS].C) & 1);
            if (nt) f |= 2;
        }
        sh.flags = f;
    }
    __syncthreads();
    if (tid == 0) sh.tokc[TILE_BLOCKS] = (sh.flags & 2) ? 1 : 0;
    const bool prev_tok = tid > 0 ? (sh.tokc[tid - 1] >> 63) != 0 : (sh.flags & 1) != 0;
    __syncthreads();
    const bool next_tok = (sh.tokc[tid + 1] & 1ull) != 0;
    const u64 TS = token_starts(TOKC, RS, prev_tok);
    const u64 TE = token_ends(TOKC, RS, next_tok, next_rs);
    EmitScan me;
    me.nb = popc(pl.C) + popc(PFX);
    me.ntk = popc(TS);
    me.lts = TS ? tid * BLOCK + top(TS) : -1;
    EmitScan ident = {0, 0, -1};
    const u64 tb64 = tile_base[t];
    EmitScan seed = {(u32)(tb64 & 0xFFFFFFFFull), (u32)(tb64 >> 32), -1};
    EmitScan tot;
    EmitScan ex = block_scan_excl_fwd(me, EmitOp(), ident, seed, sh.scan, &tot);
    const u32 out0 = seed.nb;                 // first output byte of this tile
    const u32 shift = PASSTHROUGH ? 0 : (out0 & 15u);
    sh.cw[tid] = pl.C;
    sh.pfx[tid] = PFX;
    sh.ts[tid] = TS;
    sh.eb[tid] = ex.nb;
    sh.tb[tid] = ex.ntk;
    if (tid == 0) {
        sh.total = tot;
        // does a token run into this tile from the left?
        sh.cin_start = 0;
        if ((TOKC & 1ull) && prev_tok && !(RS & 1ull)) {
            u32 r0 = rec_lo[t];   // first record starting at/after tS; the covering record is r0-1
            u64 rec_start = r0 > 0 ? offsets[r0 - 1] : 0;
            sh.cin_start = find_token_start_before<PASSTHROUGH>(bytes, N, plan, tS, rec_start);
        }
    }
    // bytes -> LDS output image
    if (!PASSTHROUGH) {
        u32 pos = ex.nb - out0 + shift;
        u64 c = pl.C;
        while (c) {
            int j = ctz(c);
            c &= c - 1;
            if ((PFX >> j) & 1) sh.out[pos++] = ((pl.PNL >> j) & 1) ? '\n' : ' ';
            sh.out[pos++] = blk[j];
        }
    }
    __syncthreads();
    // tokens: one record per token end
    {
        u64 te = TE;
        while (te) {
            int x = ctz(te);
            te &= te - 1;
            u64 sb = TS & below(x + 1);
            u64 start_abs;   // corpus position of the token's first byte
            if (sb) start_abs = p + top(sb);
            else start_abs = ex.lts >= 0 ? tS + (u64)ex.lts : sh.cin_start;
            u64 end_abs = p + x;
            u32 idx = ex.ntk + popc(sb) - 1;
            u32 out_end = PASSTHROUGH ? (u32)end_abs
                                      : ex.nb + popc(pl.C & below(x)) + popc(PFX & below(x + 1));
            u32 len = (u32)(end_abs - start_abs + 1);
            tok_start[idx] = out_end - (len - 1);
            tok_len[idx] = len;
        }
    }
    // per-record offsets
    {
        u32 r0 = rec_lo[t], r1 = rec_lo[t + 1];
        for (u64 r = (u64)r0 + tid; r < r1; r += NT) {
            u64 q = offsets[r];
            if (q >= N) continue;
            u32 l = (u32)((q - tS) >> 6), j = (u32)((q - tS) & 63);
            u64 lowm = below((int)j);
            den_off[r] = PASSTHROUGH ? q : (u64)(sh.eb[l] + popc(sh.cw[l] & lowm) + popc(sh.pfx[l] & lowm));
            tok_off[r] = (u64)(sh.tb[l] + popc(sh.ts[l] & lowm));
        }
    }
    // LDS output image -> global, 16-byte stores where the destination is aligned
    if (!PASSTHROUGH) {
        const u32 n = sh.total.nb - out0;   // bytes this tile emits
        const u64 g0 = (u64)out0 - shift;   // 16-aligned
        const u32 end = shift + n;
        for (u32 c16 = tid; c16 * 16 < end; c16 += NT) {
            u32 lo = c16 * 16, hi = lo + 16;
            if (lo >= shift && hi <= end) {
                *reinterpret_cast<uint4*>(denoised + g0 + lo) = *reinterpret_cast<const uint4*>(sh.out + lo);
            } else {
                for (u32 i = lo > shift ? lo : shift; i < hi && i < end; i++) denoised[g0 + i] = sh.out[i];
            }
        }
    }
}

// Totals, and the offsets of records that start at the very end of the corpus.
__global__ void k_finish(const u64* __restrict__ tile_base, u64 nT, const u32* __restrict__ rec_lo, u64 R,
                         u64 N, int passthrough, u64* __restrict__ den_off, u64* __restrict__ tok_off,
                         u64* __restrict__ totals) {
    u64 D = nT ? (tile_base[nT] & 0xFFFFFFFFull) : 0, T = nT ? (tile_base[nT] >> 32) : 0;
    if (passthrough) D = N;
    u64 first = rec_lo[nT + 1];   // first record whose start offset is >= N
    u64 i = first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= R) { den_off[i] = D; tok_off[i] = T; }
    if (blockIdx.x == 0 && threadIdx.x == 0) { totals[0] = D; totals[1] = T; }
}

// ---------------------------------------------------------------------------------------------------
// chunk windows
// ---------------------------------------------------------------------------------------------------
__global__ void k_chunk_count(const u64* __restrict__ den_off, const u64* __restrict__ tok_off, u64 R, DevCfg cfg,
                              u64* __restrict__ counts, u8* __restrict__ dropped) {
    u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    counts[r] = chunk_count(tok_off[r + 1] - tok_off[r], cfg);
    dropped[r] = cfg.denoise_enabled && den_off[r + 1] == den_off[r];
}

// grid-stride over all chunks; the record of a chunk is found by binary search in chunk_offsets
__global__ void k_chunk_emit(const u64* __restrict__ chunk_off, const u64* __restrict__ tok_off, u64 R, DevCfg cfg,
                             u64 capacity, ChunkDesc* __restrict__ chunks, u64* __restrict__ totals) {
    const u64 C = chunk_off[R];
    if (blockIdx.x == 0 && threadIdx.x == 0) { totals[2] = C; totals[3] = C > capacity ? 1 : 0; }
    if (C > capacity) return;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < C; i += (u64)gridDim.x * blockDim.x) {
        u64 lo = 0, hi = R;   // last r with chunk_off[r] <= i
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if (chunk_off[mid + 1] <= i) lo = mid + 1; else hi = mid;
        }
        u64 r = lo;
        chunks[i] = chunk_at(tok_off[r + 1] - tok_off[r], i - chunk_off[r], (u32)r, cfg);
    }
}

// ---------------------------------------------------------------------------------------------------
// chunk text (optional): tokens joined by one space, or the record text for a single window
// ---------------------------------------------------------------------------------------------------
__device__ inline bool chunk_is_whole_record(const ChunkDesc& c, const u64* chunk_off, const u64* tok_off) {
    u64 r = c.record;
    return c.kind == 0 && chunk_off[r + 1] - chunk_off[r] == 1 && (u64)(c.tok_end - c.tok_begin) == tok_off[r + 1] - tok_off[r];
}

// one wave per chunk
__global__ __launch_bounds__(NT) void k_chunk_text_len(const ChunkDesc* __restrict__ chunks, u64 C,
                                                        const u64* __restrict__ chunk_off, const u64* __restrict__ tok_off,
                                                        const u64* __restrict__ den_off, const u32* __restrict__ tok_len,
                                                        u64* __restrict__ lens) {
    const int lane = threadIdx.x & 63;
    u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= C) return;
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

__global__ __launch_bounds__(NT) void k_chunk_text_emit(const ChunkDesc* __restrict__ chunks, u64 C,
                                                         const u64* __restrict__ chunk_off, const u64* __restrict__ tok_off,
                                                         const u64* __restrict__ den_off, const u32* __restrict__ tok_start,
                                                         const u32* __restrict__ tok_len, const u8* __restrict__ denoised,
                                                         const u64* __restrict__ text_off, u8* __restrict__ text) {
    const int lane = threadIdx.x & 63;
    u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= C) return;
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

struct trp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf rec_lo, SA, plan, g3m, tile_bytes, tile_toks, tile_flags, packed, tile_base, partial;
    DevBuf denoised, den_off, dropped, tok_start, tok_len, tok_off, counts, chunk_off, chunks, totals;
    DevBuf in_bytes, in_offsets;        // staging of the host-memory entry point
    DevBuf text, text_off, text_len;    // chunk text
    u64* h_totals = nullptr;            // pinned
    bool profiling = false;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> events;
    std::vector<trp_kernel_time> last_profile;
    // last device batch (for chunk text)
    trp_result last = {};
    u64 last_R = 0;
    bool have_last = false;
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

int scan_u64(trp_ctx* c, const u64* in, u64 n, u64* out) {
    u64 nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (nb == 0) nb = 1;
    int rc = c->partial.ensure(nb * sizeof(u64));
    if (rc) return rc;
    u64* partial = c->partial.as<u64>();
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(NT), 0, c->stream, in, n, partial);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, c->stream, partial, nb);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(NT), 0, c->stream, in, n, partial, out);
    HIPCHK(hipGetLastError());
    return TRP_OK;
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

int trp_create(int device, trp_ctx** out) {
    if (!out) return fail(TRP_E_ARG, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(TRP_E_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(TRP_E_ARG, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    trp_ctx* c = new trp_ctx();
    c->device = device;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(TRP_E_HIP, "hipStreamCreate failed"); }
    e = hipHostMalloc((void**)&c->h_totals, 8 * sizeof(u64), hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return fail(TRP_E_HIP, "hipHostMalloc failed"); }
    *out = c;
    return TRP_OK;
}

void trp_destroy(trp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    DevBuf* bufs[] = {&c->rec_lo, &c->SA, &c->plan, &c->g3m, &c->tile_bytes, &c->tile_toks, &c->tile_flags, &c->packed,
                      &c->tile_base, &c->partial, &c->denoised, &c->den_off, &c->dropped, &c->tok_start,
                      &c->tok_len, &c->tok_off, &c->counts, &c->chunk_off, &c->chunks, &c->totals,
                      &c->in_bytes, &c->in_offsets, &c->text, &c->text_off, &c->text_len};
    for (DevBuf* b : bufs) b->release();
    if (c->h_totals) (void)hipHostFree(c->h_totals);
    (void)hipStreamDestroy(c->stream);
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
    int n = (int)c->last_profile.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = c->last_profile[i];
    return n;
}

int trp_preprocess_batch_device(trp_ctx* c, const uint8_t* d_bytes, const uint64_t* d_offsets, uint64_t R,
                                uint64_t N, const trp_config* cfg, trp_result* out) {
    if (!c || !out) return fail(TRP_E_ARG, "ctx/out is null");
    if (!d_offsets) return fail(TRP_E_ARG, "offsets is null");
    if (N && !d_bytes) return fail(TRP_E_ARG, "bytes is null");
    if (((uintptr_t)d_bytes & 15u) != 0) return fail(TRP_E_ARG, "bytes must be 16-byte aligned");
    if (N > TRP_MAX_BATCH_BYTES || R >= 0xFFFFFFFEull) return fail(TRP_E_TOO_LARGE, "batch too large: split it by record range");
    DevCfg dc;
    int rc = make_devcfg(cfg, &dc);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    c->have_last = false;
    const u64 nB = (N + BLOCK - 1) / BLOCK, nT = (N + TILE - 1) / TILE;
    const bool pass = !dc.denoise_enabled;
    const u64 tok_cap = (N + 1) / 2 + 2;
    u64 chunk_cap = chunk_bound(N, R, dc);
    const u64 chunk_cap_limit = (u64)1 << 28;   // do not pre-allocate more than 8 GiB of descriptors
    if (chunk_cap > chunk_cap_limit) chunk_cap = chunk_cap_limit;
    if (c->chunks.cap / sizeof(ChunkDesc) > chunk_cap) chunk_cap = c->chunks.cap / sizeof(ChunkDesc);

    if ((rc = c->rec_lo.ensure((nT + 2) * sizeof(u32)))) return rc;
    if ((rc = c->tile_bytes.ensure((nT + 1) * sizeof(u32)))) return rc;
    if ((rc = c->tile_toks.ensure((nT + 1) * sizeof(u32)))) return rc;
    if ((rc = c->tile_flags.ensure((nT + 1) * sizeof(u32)))) return rc;
    if ((rc = c->packed.ensure((nT + 1) * sizeof(u64)))) return rc;
    if ((rc = c->tile_base.ensure((nT + 2) * sizeof(u64)))) return rc;
    if ((rc = c->den_off.ensure((R + 1) * sizeof(u64)))) return rc;
    if ((rc = c->tok_off.ensure((R + 1) * sizeof(u64)))) return rc;
    if ((rc = c->chunk_off.ensure((R + 2) * sizeof(u64)))) return rc;
    if ((rc = c->counts.ensure((R + 1) * sizeof(u64)))) return rc;
    if ((rc = c->dropped.ensure(R + 1))) return rc;
    if ((rc = c->tok_start.ensure(tok_cap * sizeof(u32)))) return rc;
    if ((rc = c->tok_len.ensure(tok_cap * sizeof(u32)))) return rc;
    if ((rc = c->chunks.ensure(chunk_cap * sizeof(ChunkDesc)))) return rc;
    if ((rc = c->totals.ensure(8 * sizeof(u64)))) return rc;
    if (!pass) {
        if ((rc = c->SA.ensure((nT + 1) * sizeof(J)))) return rc;
        if ((rc = c->plan.ensure((nB + 1) * sizeof(Plan)))) return rc;
        if ((rc = c->g3m.ensure((nB + 1) * sizeof(u64)))) return rc;
        if ((rc = c->denoised.ensure(N + 64))) return rc;
    }
    hipStream_t st = c->stream;
    u32* rec_lo = c->rec_lo.as<u32>();
    u32 *tbytes = c->tile_bytes.as<u32>(), *ttoks = c->tile_toks.as<u32>(), *tflags = c->tile_flags.as<u32>();
    u64* tile_base = c->tile_base.as<u64>();
    u64 *den_off = c->den_off.as<u64>(), *tok_off = c->tok_off.as<u64>(), *chunk_off = c->chunk_off.as<u64>();
    u64* totals = c->totals.as<u64>();
    Plan* plan = c->plan.as<Plan>();
    J* SA = c->SA.as<J>();
    c->events.clear();

    {
        Prof pr(c, "k_rec_lo");
        hipLaunchKernelGGL(k_rec_lo, dim3((unsigned)((nT + 2 + 255) / 256)), dim3(256), 0, st, d_offsets, R, N, nT, rec_lo);
    }
    if (nT) {
        if (!pass) {
            { Prof pr(c, "k_summary");
              hipLaunchKernelGGL(k_summary, dim3((unsigned)nT), dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, SA); }
            { Prof pr(c, "k_plan");
              hipLaunchKernelGGL(k_plan, dim3((unsigned)nT), dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nT, SA, dc,
                                 plan, c->g3m.as<u64>(), tbytes, ttoks, tflags); }
            { Prof pr(c, "k_gate3");
              hipLaunchKernelGGL(k_gate3, dim3((unsigned)((nB + NT - 1) / NT)), dim3(NT), 0, st, d_bytes, d_offsets, R, N, nB,
                                 dc, plan, c->g3m.as<u64>(), tbytes, ttoks); }
        } else {
            Prof pr(c, "k_count_tokens");
            hipLaunchKernelGGL(k_count_tokens, dim3((unsigned)nT), dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, tbytes,
                               ttoks, tflags);
        }
        { Prof pr(c, "scan_tiles");
          hipLaunchKernelGGL(k_pack_tiles, dim3((unsigned)((nT + 255) / 256)), dim3(256), 0, st, tbytes, ttoks, tflags, nT,
                             c->packed.as<u64>());
          if ((rc = scan_u64(c, c->packed.as<u64>(), nT, tile_base))) return rc; }
        { Prof pr(c, "k_emit");
          if (pass)
              hipLaunchKernelGGL(k_emit<true>, dim3((unsigned)nT), dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nT,
                                 (const Plan*)nullptr, tile_base, (u8*)nullptr, c->tok_start.as<u32>(), c->tok_len.as<u32>(),
                                 den_off, tok_off);
          else
              hipLaunchKernelGGL(k_emit<false>, dim3((unsigned)nT), dim3(NT), 0, st, d_bytes, d_offsets, rec_lo, R, N, nT,
                                 plan, tile_base, c->denoised.as<u8>(), c->tok_start.as<u32>(), c->tok_len.as<u32>(),
                                 den_off, tok_off); }
    }
    { Prof pr(c, "k_finish");
      // all records at/after N (at least the sentinel entry R)
      hipLaunchKernelGGL(k_finish, dim3((unsigned)((R + 1 + 255) / 256)), dim3(256), 0, st, tile_base, nT, rec_lo, R, N,
                         pass ? 1 : 0, den_off, tok_off, totals); }
    { Prof pr(c, "k_chunk_count");
      if (R) hipLaunchKernelGGL(k_chunk_count, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, den_off, tok_off, R, dc,
                                c->counts.as<u64>(), c->dropped.as<u8>()); }
    { Prof pr(c, "scan_chunks");
      if ((rc = scan_u64(c, c->counts.as<u64>(), R, chunk_off))) return rc; }
    { Prof pr(c, "k_chunk_emit");
      hipLaunchKernelGGL(k_chunk_emit, dim3(1024), dim3(256), 0, st, chunk_off, tok_off, R, dc, chunk_cap,
                         c->chunks.as<ChunkDesc>(), totals); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_totals, totals, 4 * sizeof(u64), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (c->h_totals[3]) {
        // the optimistic descriptor buffer was too small: grow to the exact size and emit again
        chunk_cap = c->h_totals[2];
        if ((rc = c->chunks.ensure(chunk_cap * sizeof(ChunkDesc)))) return rc;
        hipLaunchKernelGGL(k_chunk_emit, dim3(1024), dim3(256), 0, st, chunk_off, tok_off, R, dc, chunk_cap,
                           c->chunks.as<ChunkDesc>(), totals);
        HIPCHK(hipMemcpyAsync(c->h_totals, totals, 4 * sizeof(u64), hipMemcpyDeviceToHost, st));
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
    out->denoised = pass ? const_cast<uint8_t*>(d_bytes) : c->denoised.as<u8>();
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

int trp_materialize_chunk_text_device(trp_ctx* c, uint8_t** d_text, uint64_t** d_text_offsets, uint64_t* n_text_bytes) {
    if (!c || !d_text || !d_text_offsets || !n_text_bytes) return fail(TRP_E_ARG, "null argument");
    if (!c->have_last) return fail(TRP_E_ARG, "no device batch result to render");
    HIPCHK(hipSetDevice(c->device));
    const trp_result& L = c->last;
    const u64 C = L.n_chunks, R = c->last_R;
    int rc;
    if ((rc = c->text_len.ensure((C + 1) * sizeof(u64)))) return rc;
    if ((rc = c->text_off.ensure((C + 2) * sizeof(u64)))) return rc;
    const ChunkDesc* chunks = reinterpret_cast<const ChunkDesc*>(L.chunks);
    (void)R;
    if (C) hipLaunchKernelGGL(k_chunk_text_len, dim3((unsigned)((C * 64 + NT - 1) / NT)), dim3(NT), 0, c->stream, chunks, C,
                              L.chunk_offsets, L.tok_offsets, L.denoised_offsets, L.tok_len, c->text_len.as<u64>());
    if ((rc = scan_u64(c, c->text_len.as<u64>(), C, c->text_off.as<u64>()))) return rc;
    HIPCHK(hipMemcpyAsync(c->h_totals + 4, c->text_off.as<u64>() + C, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    u64 nbytes = c->h_totals[4];
    if ((rc = c->text.ensure(nbytes + 16))) return rc;
    if (C) hipLaunchKernelGGL(k_chunk_text_emit, dim3((unsigned)((C * 64 + NT - 1) / NT)), dim3(NT), 0, c->stream, chunks, C,
                              L.chunk_offsets, L.tok_offsets, L.denoised_offsets, L.tok_start, L.tok_len, L.denoised,
                              c->text_off.as<u64>(), c->text.as<u8>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    *d_text = c->text.as<u8>();
    *d_text_offsets = c->text_off.as<u64>();
    *n_text_bytes = nbytes;
    return TRP_OK;
}

int trp_copy_to_host(trp_ctx* c, void* dst, const void* d_src, size_t bytes) {
    if (!c || (bytes && (!dst || !d_src))) return fail(TRP_E_ARG, "null argument");
    HIPCHK(hipSetDevice(c->device));
    if (bytes) HIPCHK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TRP_OK;
}

void trp_result_free(trp_result* r) {
    if (!r) return;
    free(r->denoised); free(r->denoised_offsets); free(r->dropped); free(r->tok_start); free(r->tok_len);
    free(r->tok_offsets); free(r->chunks); free(r->chunk_offsets);
    memset(r, 0, sizeof(*r));
}

void trp_free(void* p) { free(p); }

int trp_preprocess_batch(trp_ctx* c, const uint8_t* bytes, const uint64_t* offsets, uint64_t R, const trp_config* cfg,
                         trp_result* out) {
    if (!c || !out || !offsets) return fail(TRP_E_ARG, "null argument");
    if (offsets[0] != 0) return fail(TRP_E_ARG, "offsets[0] must be 0");
    const u64 N = offsets[R];
    if (N && !bytes) return fail(TRP_E_ARG, "bytes is null");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if ((rc = c->in_bytes.ensure(N + 64))) return rc;
    if ((rc = c->in_offsets.ensure((R + 1) * sizeof(u64)))) return rc;
    if (N) HIPCHK(hipMemcpyAsync(c->in_bytes.p, bytes, N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->in_offsets.p, offsets, (R + 1) * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    trp_result d;
    rc = trp_preprocess_batch_device(c, c->in_bytes.as<u8>(), c->in_offsets.as<u64>(), R, N, cfg, &d);
    if (rc) return rc;
    memset(out, 0, sizeof(*out));
    out->n_records = R; out->n_denoised = d.n_denoised; out->n_tokens = d.n_tokens; out->n_chunks = d.n_chunks;
    auto grab = [&](void** dst, const void* src, size_t bytes_) -> int {
        *dst = malloc(bytes_ ? bytes_ : 1);
        if (!*dst) return fail(TRP_E_ARG, "out of host memory");
        if (bytes_) {
            hipError_t e = hipMemcpyAsync(*dst, src, bytes_, hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) return fail(TRP_E_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
        }
        return TRP_OK;
    };
    if ((rc = grab((void**)&out->denoised, d.denoised, d.n_denoised))) return rc;
    if ((rc = grab((void**)&out->denoised_offsets, d.denoised_offsets, (R + 1) * sizeof(u64)))) return rc;
    if ((rc = grab((void**)&out->dropped, d.dropped, R))) return rc;
    if ((rc = grab((void**)&out->tok_start, d.tok_start, d.n_tokens * sizeof(u32)))) return rc;
    if ((rc = grab((void**)&out->tok_len, d.tok_len, d.n_tokens * sizeof(u32)))) return rc;
    if ((rc = grab((void**)&out->tok_offsets, d.tok_offsets, (R + 1) * sizeof(u64)))) return rc;
    if ((rc = grab((void**)&out->chunks, d.chunks, d.n_chunks * sizeof(trp_chunk)))) return rc;
    if ((rc = grab((void**)&out->chunk_offsets, d.chunk_offsets, (R + 1) * sizeof(u64)))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
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
    free(s); free(l);
    return rc;
}

int trp_chunk_section(trp_ctx* c, const uint8_t* text, size_t len, const trp_config* cfg, trp_result* out) {
    return one_record(c, text, len, cfg, out);
}

}  // extern "C"
