// Time per blocking call of the C ABI on one record of ordinary prose, measured from C (no interpreter between
// the calls): the device-resident batch call and the host-memory single-record conveniences.
//   hipcc -O2 -I include -o /tmp/call_latency tools/microbench/call_latency.cpp -Lrust-triplets_amd/csrc \
//         -ltriplets_amd -Wl,-rpath,$PWD/rust-triplets_amd/csrc && /tmp/call_latency
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "triplets_amd.h"

template <class F>
static double median_us(F f) {
    for (int i = 0; i < 50; i++) f();
    std::vector<double> runs;
    for (int r = 0; r < 9; r++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; i++) f();
        runs.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200);
    }
    std::sort(runs.begin(), runs.end());
    return runs[runs.size() / 2];
}

int main() {
    trp_ctx* ctx = nullptr;
    if (trp_create(0, &ctx) != TRP_OK) { std::fprintf(stderr, "no GPU context\n"); return 2; }
    const char* words[] = {"the", "sampler", "draws", "anchor", "and", "positive", "sections", "from", "records",
                           "in", "2024", "with", "window", "overlap", "of", "tokens,", "then", "joins", "them."};
    const uint32_t ov[1] = {64};
    trp_config cfg = {1, 1, 0.35f, 1024, ov, 1, 512};
    for (size_t want : {256u, 1024u, 4096u, 16000u, 32000u}) {
        std::string text;
        for (size_t i = 0; text.size() < want; i++) {
            text += words[(i * 7 + i / 19) % 19];
            text += (i % 13 == 12) ? '\n' : ' ';
        }
        text.resize(want);
        uint8_t* d_bytes; uint64_t* d_off;
        (void)hipMalloc(&d_bytes, want + 64); (void)hipMalloc(&d_off, 16);
        (void)hipMemset(d_bytes, 0, want + 64);
        const uint64_t off[2] = {0, want};
        (void)hipMemcpy(d_bytes, text.data(), want, hipMemcpyHostToDevice);
        (void)hipMemcpy(d_off, off, 16, hipMemcpyHostToDevice);
        trp_result res;
        int bad = 0;
        const double dev = median_us([&] { bad |= trp_preprocess_batch_device(ctx, d_bytes, d_off, 1, want, &cfg, &res); });
        size_t n = 0;
        const double cnt = median_us([&] { bad |= trp_token_count(ctx, (const uint8_t*)text.data(), want, &n); });
        const double den = median_us([&] {
            uint8_t* out = nullptr; size_t len = 0;
            bad |= trp_denoise(ctx, (const uint8_t*)text.data(), want, &cfg, &out, &len) < 0;
            trp_free(out);
        });
        const double chk = median_us([&] { bad |= trp_chunk_section(ctx, (const uint8_t*)text.data(), want, &cfg, &res); });
        const uint64_t hoff[2] = {0, want};
        const uint32_t pick[1] = {0}, budget[1] = {40};
        const double cap = median_us([&] {
            trp_chunk_text t;
            bad |= trp_head_tokens_text(ctx, (const uint8_t*)text.data(), hoff, 1, 64, &t);
            trp_chunk_text_free(&t);
        });
        const double body = median_us([&] {
            trp_chunk_text t;
            bad |= trp_chunk_bodies(ctx, (const uint8_t*)text.data(), hoff, 1, &cfg, pick, budget, 1, &t);
            trp_chunk_text_free(&t);
        });
        std::printf("record of %6zu bytes (%zu tokens): device batch call %.1f us, trp_token_count %.1f us, trp_denoise %.1f us, "
                    "trp_chunk_section %.1f us, trp_head_tokens_text %.1f us, trp_chunk_bodies %.1f us%s\n", want, n, dev, cnt, den, chk,
                    cap, body, bad ? "  (a call FAILED)" : "");
        (void)hipFree(d_bytes); (void)hipFree(d_off);
    }
    // batches of 4 KB records in host memory: every array back (trp_preprocess_batch) or descriptors only
    for (size_t want : {16u << 10, 32u << 10, 64u << 10, 120u << 10, 256u << 10, 1u << 20, 4u << 20, 16u << 20}) {
        std::string text;
        std::vector<uint64_t> offs{0};
        for (size_t i = 0; text.size() < want; i++) {
            text += words[(i * 7 + i / 19) % 19];
            text += (i % 13 == 12) ? '\n' : ' ';
            if (text.size() - offs.back() >= 4096) offs.push_back(text.size());
        }
        if (offs.back() != text.size()) offs.push_back(text.size());
        const uint64_t R = offs.size() - 1;
        trp_result res;
        int bad = 0;
        const double all = median_us([&] {
            bad |= trp_preprocess_batch(ctx, (const uint8_t*)text.data(), offs.data(), R, &cfg, &res);
            trp_result_free(&res);
        });
        const double chunks = median_us([&] {
            bad |= trp_preprocess_batch_select(ctx, (const uint8_t*)text.data(), offs.data(), R, &cfg, TRP_OUT_CHUNKS, &res, nullptr);
            trp_result_free(&res);
        });
        std::printf("host batch of %9zu bytes in %5llu records: all arrays %.1f us (%.2f GB/s), descriptors only %.1f us (%.2f GB/s)%s\n",
                    text.size(), (unsigned long long)R, all, text.size() / all / 1e3, chunks, text.size() / chunks / 1e3,
                    bad ? "  (a call FAILED)" : "");
    }
    trp_destroy(ctx);
    return 0;
}
