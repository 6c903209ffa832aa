// Where does k_summary's time go?  Variants that stop after each stage, timed on 1 GB of ASCII text.
#include "../../rust-triplets_amd/csrc/trp_kernels.hip"
#include <vector>

namespace {
template <int STAGE>
__global__ __launch_bounds__(NT) void k_exp(const u8* __restrict__ bytes, const u64* __restrict__ rsmap, u64 N,
                                             u64 nB, J* __restrict__ SA, u32* sink) {
    __shared__ TileShared sh;
    __shared__ UniShared uni;
    __shared__ J wsum[NT / 64];
    const u64 t = blockIdx.x, tS = t * (u64)TILE;
    const u64 b = t * (u64)TILE_BLOCKS + threadIdx.x;
    { TileRegs regs; tile_load(bytes, N, tS, regs); tile_store(regs, sh.bytes); }
    stage_tables(uni);
    const u64 rs_here = b < nB ? rsmap[b] : 0, rs_next = b + 1 < nB ? rsmap[b + 1] : 0;
    __syncthreads();
    if (STAGE == 0) {   // read the lane's words only
        const PadBlk blk = lane_block(sh, threadIdx.x);
        u32 acc = 0;
        for (int k = 0; k < 16; k++) acc ^= blk.word(k);
        if (acc == 0x12345u) sink[0] = acc + (u32)rs_here + (u32)rs_next;
        return;
    }
    if (STAGE == 1) {   // + transposition + class formulas
        const PadBlk blk = lane_block(sh, threadIdx.x);
        Cls c = classify_block_fast(blk, 64, lds_tables(uni));
        u64 x = c.ws ^ c.nl ^ c.cr ^ c.pipe ^ c.dig ^ c.alp ^ c.sp;
        if (x == 0x12345u) sink[0] = (u32)x + (u32)rs_here + (u32)rs_next;
        return;
    }
    LaneCtx L = lane_setup(sh, lds_tables(uni), N, tS, true, rs_here, rs_next);
    if (STAGE == 2) {
        u64 x = L.d.LS ^ L.d.TERM ^ L.d.LC ^ L.c.alp;
        if (x == 0x12345u) sink[0] = (u32)x;
        return;
    }
    J j = lane_J(L.c, L.d);
    if (STAGE == 3) {
        u32 x = j.g ^ j.head.d ^ j.head.a ^ j.head.p ^ j.head.f ^ j.tail.d ^ j.tail.a ^ j.tail.p ^ j.tail.f;
        if (x == 0x12345u) sink[0] = x;
        return;
    }
    if (STAGE == 5) {   // check the ballot formulation against the ordered fold
        J a = wave_reduce_ordered(j, JOp());
        J w = wave_total(wave_lines(j));
        if ((threadIdx.x & 63) == 0) {
            bool same = a.g == w.g && a.head.d == w.head.d && a.head.a == w.head.a && a.head.p == w.head.p &&
                        a.head.f == w.head.f;
            if (a.g & G_LS) same = same && a.tail.d == w.tail.d && a.tail.a == w.tail.a && a.tail.p == w.tail.p && a.tail.f == w.tail.f;
            if (!same) atomicAdd(&sink[1], 1u);
            atomicAdd(&sink[2], 1u);
        }
        return;
    }
    j = STAGE == 4 ? wave_total(wave_lines(j)) : wave_reduce_ordered(j, JOp());
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = j;
    __syncthreads();
    if (threadIdx.x == 0) {
        J acc = wsum[0];
        for (int w = 1; w < NT / 64; w++) acc = j_combine(acc, wsum[w]);
        SA[t] = acc;
    }
}
}  // namespace

int main() {
    const u64 N = 1000000000ull;
    std::vector<u8> h(N);
    u64 x = 42;
    for (u64 i = 0; i < N; i++) {   // words of lowercase letters, spaces, a newline every ~100 bytes
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        u32 r = (u32)(x >> 33);
        // three regimes: prose, sparse lines (many blocks without a line start), dense short lines
        u32 regime = (u32)((i >> 22) % 3);
        u32 nlp = regime == 0 ? 100 : (regime == 1 ? 3000 : 7);
        h[i] = (r % nlp == 0) ? '\n' : (r % 6 == 0 ? ' ' : (r % 37 == 1 ? '0' + r % 10 : (r % 41 == 2 ? '|' : (r % 5 == 3 ? '-' : 'a' + r % 26))));
    }
    std::vector<u64> hrs((N + 63) / 64 + 2, 0);
    for (u64 q = 0; q < N; ) {   // record starts at random distances (short and long records)
        hrs[q >> 6] |= 1ull << (q & 63);
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        u32 r = (u32)(x >> 33);
        q += (r % 7 == 0) ? 1 + r % 50 : 1 + r % 9000;
    }
    u8* d; u64* rs; J* sa; u32* sink;
    const u64 nB = (N + 63) / 64, nT = (N + TILE - 1) / TILE;
    if (hipMalloc(&d, N + 64) != hipSuccess) return 1;
    (void)hipMalloc(&rs, (nB + 2) * 8); (void)hipMalloc(&sa, nT * sizeof(J)); (void)hipMalloc(&sink, 64);
    (void)hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    (void)hipMemcpy(rs, hrs.data(), (nB + 2) * 8, hipMemcpyHostToDevice); (void)hipMemset(sink, 0, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int s = 0; s <= 6; s++) {
        float best = 1e9;
        for (int r = 0; r < 5; r++) {
            (void)hipEventRecord(e0);
            switch (s) {
                case 0: hipLaunchKernelGGL(k_exp<0>, dim3(nT), dim3(NT), 0, 0, d, rs, N, nB, sa, sink); break;
                case 1: hipLaunchKernelGGL(k_exp<1>, dim3(nT), dim3(NT), 0, 0, d, rs, N, nB, sa, sink); break;
                case 2: hipLaunchKernelGGL(k_exp<2>, dim3(nT), dim3(NT), 0, 0, d, rs, N, nB, sa, sink); break;
                case 3: hipLaunchKernelGGL(k_exp<3>, dim3(nT), dim3(NT), 0, 0, d, rs, N, nB, sa, sink); break;
                case 4: hipLaunchKernelGGL(k_exp<4>, dim3(nT), dim3(NT), 0, 0, d, rs, N, nB, sa, sink); break;
                case 5: hipLaunchKernelGGL(k_exp<5>, dim3(nT), dim3(NT), 0, 0, d, rs, N, nB, sa, sink); break;
                default: hipLaunchKernelGGL(k_exp<6>, dim3(nT), dim3(NT), 0, 0, d, rs, N, nB, sa, sink); break;
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const char* nm[] = {"stage + lane word reads", "+ classify", "+ derive", "+ lane_J", "+ prefix-sum wave total + store", "check wave total == fold", "+ shuffle fold + store (old)"};
        printf("%-34s %.3f ms\n", nm[s], best);
    }
    u32 hs[4] = {0, 0, 0, 0};
    (void)hipMemcpy(hs, sink, 16, hipMemcpyDeviceToHost);
    printf("wave total check: %u mismatches in %u waves\n", hs[1], hs[2]);
    return hipDeviceSynchronize() == hipSuccess && hs[1] == 0 ? 0 : 1;
}
