// quantize_probe.hip — diagnostic build of k_quantize with phase stamps (SBM_QSTAMP): where does a tile's
// time go, how many tiles does a CU hold at once, and what clock does the kernel run at?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/quantize_probe.hip -o gpurun_out/quantize_probe
#ifndef SBM_NO_QSTAMP
#define SBM_QSTAMP
#endif
#ifdef SBM_QSTAMP_B
#define STAMP_BASE 10 /* per-wave stamps sit at the end of phase B: relative to the start of B */
#else
#define STAMP_BASE 11
#endif
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include "../shape_based_matching_amd/csrc/sbm_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace sbm;
#ifndef PROBE_QN
#define PROBE_QN QN_LATENCY
#endif
int main(int argc, char** argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 1024, cols = argc > 2 ? atoi(argv[2]) : 1024;
    std::vector<uint8_t> img((size_t)rows * cols * 3);
    srand(7);
    for (auto& b : img) b = (uint8_t)(rand() >> 7);
    if (getenv("PROBE_BLACK")) std::fill(img.begin(), img.end(), (uint8_t)0); // every tile takes the flat-tile exit
    uint8_t *d_img, *d_out, *d_pyr;
    CK(hipMalloc(&d_img, img.size())); CK(hipMalloc(&d_out, (size_t)rows * cols)); CK(hipMalloc(&d_pyr, img.size() / 4));
    CK(hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice));
    const dim3 grid((cols + QT_C - 1) / QT_C, (rows + QT_R - 1) / QT_R);
    const int nt = grid.x * grid.y;
    unsigned long long* d_st; CK(hipMalloc(&d_st, (size_t)nt * 64 * 8)); CK(hipMemset(d_st, 0, (size_t)nt * 64 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch = [&] { hipLaunchKernelGGL((k_quantize<3, false, PROBE_QN>), grid, dim3(PROBE_QN), 0, s, d_img, rows, cols, cols * 3, (const uint8_t*)nullptr, 900.f, d_out, (float*)nullptr, (float*)nullptr, d_pyr, (int64_t)0, (int64_t)0, (int64_t)0); };
    for (int i = 0; i < 300; ++i) launch();
    CK(hipEventRecord(a, s));
    for (int i = 0; i < 300; ++i) launch();
    CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%dx%d: %d tiles, back-to-back %.2f us/launch (stamps off)\n", rows, cols, nt, ms * 1e3 / 300);
#ifdef SBM_NO_QSTAMP
    // production kernel, timing only: noise in the 8 tiles of one (block % 8, block / 8 % 4) class (= one shader
    // engine of one XCD when blocks are dealt round-robin), black elsewhere
    if (argc > 3) {
        printf("launch time with noise only in tiles of class (b%%8, b/8%%4):\n");
        for (int x = 0; x < 8; ++x) {
            for (int y = 0; y < 4; ++y) {
                std::vector<uint8_t> im(img.size(), 0);
                for (int t = 0; t < nt; ++t)
                    if (t % 8 == x && (t / 8) % 4 == y) {
                        const int by = t / grid.x, bx = t % grid.x;
                        for (int r = by * QT_R; r < std::min(rows, (by + 1) * QT_R); ++r)
                            for (int c = bx * QT_C * 3; c < std::min(cols, (bx + 1) * QT_C) * 3; ++c) im[(size_t)r * cols * 3 + c] = img[(size_t)r * cols * 3 + c];
                    }
                CK(hipMemcpy(d_img, im.data(), im.size(), hipMemcpyHostToDevice));
                for (int i = 0; i < 50; ++i) launch();
                CK(hipEventRecord(a, s));
                for (int i = 0; i < 200; ++i) launch();
                CK(hipEventRecord(b, s)); CK(hipStreamSynchronize(s));
                CK(hipEventElapsedTime(&ms, a, b));
                printf(" %d.%d:%.2f", x, y, ms * 1e3 / 200);
            }
            printf("\n");
        }
    }
    return 0;
#else
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_qstamp), &d_st, sizeof(d_st)));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> st((size_t)nt * 64);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int t = 0; t < nt; ++t) { t0 = std::min(t0, st[t * 64]); t1 = std::max(t1, st[t * 64 + 6]); }
    printf("kernel span (first tile start -> last tile end) %.2f us\n", (t1 - t0) / 100.0);
    const char* names[6] = {"A load", "P pyr+flat", "B horiz", "C vert", "D sobel", "E vote"};
    for (int p = 0; p < 6; ++p) {
        std::vector<double> us, cyc;
        for (int t = 0; t < nt; ++t) { us.push_back((st[t * 64 + p + 1] - st[t * 64 + p]) / 100.0); cyc.push_back((double)(st[t * 64 + 8 + p + 1] - st[t * 64 + 8 + p])); }
        std::sort(us.begin(), us.end()); std::sort(cyc.begin(), cyc.end());
        printf("  %-12s median %.2f us  p90 %.2f us   median %.0f cycles\n", names[p], us[nt / 2], us[nt * 9 / 10], cyc[nt / 2]);
    }
    std::vector<double> life, clk;
    for (int t = 0; t < nt; ++t) {
        life.push_back((st[t * 64 + 6] - st[t * 64]) / 100.0);
        clk.push_back((double)(st[t * 64 + 14] - st[t * 64 + 8]) / (double)(st[t * 64 + 6] - st[t * 64]) * 0.1);
    }
    std::sort(life.begin(), life.end()); std::sort(clk.begin(), clk.end());
    printf("tile lifetime median %.2f us p90 %.2f us; in-kernel clock median %.2f GHz\n", life[nt / 2], life[nt * 9 / 10], clk[nt / 2]);
    // per-CU schedule: how many tiles overlap
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> cu;
    std::map<unsigned long long, std::vector<int>> cu_ids;
    for (int t = 0; t < nt; ++t) {
        const unsigned long long hw = st[t * 64 + 16], xcc = st[t * 64 + 17] & 0xf;
        const unsigned long long key = (xcc << 16) | (hw & 0xff00); // se, sh, cu
        cu[key].push_back({st[t * 64], st[t * 64 + 6]});
        cu_ids[key].push_back(t);
    }
    std::map<int, int> hist; int maxov = 0; std::map<int,int> ovh;
    for (auto& kv : cu) {
        hist[(int)kv.second.size()]++;
        int best = 0;
        for (auto& x : kv.second) { int ov = 0; for (auto& y : kv.second) if (y.first <= x.first && y.second > x.first) ++ov; best = std::max(best, ov); }
        ovh[best]++;
    }
    printf("CUs seen %zu; tiles per CU histogram:", cu.size());
    for (auto& h : hist) printf(" %d:%d", h.first, h.second);
    printf("; max concurrently resident tiles per CU:");
    for (auto& h : ovh) printf(" %d:%d", h.first, h.second);
    printf("\n");
    auto it = cu.begin();
    for (int k = 0; k < (argc > 3 ? atoi(argv[3]) : 3) && it != cu.end(); ++k, ++it) {
        printf("  CU %05llx:", it->first);
        std::sort(it->second.begin(), it->second.end());
        for (auto& x : it->second) printf(" [%.2f..%.2f]", (x.first - t0) / 100.0, (x.second - t0) / 100.0);
        printf("  ids:"); for (int id : cu_ids[it->first]) printf(" %d", id);
        printf("\n");
    }
    // first-round tiles (start < 1 us, the older of the two on the CU): phase medians of the fast vs slow ones
    for (int pass = 0; pass < 2; ++pass) {
        std::vector<std::vector<double>> ph(6);
        int cnt = 0;
        for (int t = 0; t < nt; ++t) {
            if (st[t * 64] - t0 > 30) continue; // started within the first 0.3 us -> the older tile
            const double lt = (st[t * 64 + 6] - st[t * 64]) / 100.0;
            if ((pass == 0) != (lt < 6.3)) continue;
            ++cnt;
            for (int p = 0; p < 6; ++p) ph[p].push_back((double)(st[t * 64 + 8 + p + 1] - st[t * 64 + 8 + p]));
        }
        {
            std::vector<double> own, wait;
            for (int t = 0; t < nt; ++t) {
                if (st[t * 64] - t0 > 30) continue;
                const double lt = (st[t * 64 + 6] - st[t * 64]) / 100.0;
                if ((pass == 0) != (lt < 6.3)) continue;
                own.push_back((double)(st[t * 64 + 15] - st[t * 64 + 11])); wait.push_back((double)(st[t * 64 + 12] - st[t * 64 + 15]));
            }
            // per-wave arrival at the end of C relative to the start-of-C stamp of wave 0
            std::vector<std::vector<double>> wv(16);
            for (int t = 0; t < nt; ++t) {
                if (st[t * 64] - t0 > 30) continue;
                const double lt = (st[t * 64 + 6] - st[t * 64]) / 100.0;
                if ((pass == 0) != (lt < 6.3)) continue;
                for (int w = 0; w < 16; ++w) wv[w].push_back((double)st[t * 64 + 24 + w] - (double)st[t * 64 + STAMP_BASE]);
            }
            if (!wv[0].empty()) { printf("   C done, per wave (cyc after C start):"); for (int w = 0; w < 16; ++w) { std::sort(wv[w].begin(), wv[w].end()); printf(" %.0f", wv[w][wv[w].size() / 2]); } printf("\n"); }
            std::vector<std::vector<double>> wd(16);
            for (int t = 0; t < nt; ++t) {
                if (st[t * 64] - t0 > 30) continue;
                const double lt = (st[t * 64 + 6] - st[t * 64]) / 100.0;
                if ((pass == 0) != (lt < 6.3)) continue;
                for (int w = 0; w < 16; ++w) wd[w].push_back((double)st[t * 64 + 40 + w] - (double)st[t * 64 + 11]);
            }
            if (!wd[0].empty()) { printf("   LDS drained, per wave (cyc after C start):"); for (int w = 0; w < 16; ++w) { std::sort(wd[w].begin(), wd[w].end()); printf(" %.0f", wd[w][wd[w].size() / 2]); } printf("\n"); }
            if (!own.empty()) { std::sort(own.begin(), own.end()); std::sort(wait.begin(), wait.end()); printf("   C: wave 0 own work %.0f cyc, then barrier wait %.0f cyc\n", own[own.size() / 2], wait[wait.size() / 2]); }
        }
        printf("%s first tiles (%d):", pass == 0 ? "fast" : "slow", cnt);
        for (int p = 0; p < 6 && cnt; ++p) { std::sort(ph[p].begin(), ph[p].end()); printf(" %s %.0f cyc", names[p], ph[p][cnt / 2]); }
        printf("\n");
    }
    { // per shader engine: median lifetime of the first tile of each CU
        std::map<int, std::vector<double>> se;
        for (auto& kv : cu) {
            auto v = kv.second; std::sort(v.begin(), v.end());
            se[(int)(kv.first >> 13)].push_back((v[0].second - v[0].first) / 100.0);
        }
        printf("first-tile lifetime by (xcc,se):");
        for (auto& kv : se) { std::sort(kv.second.begin(), kv.second.end()); printf(" %d.%d:%.1f", kv.first >> 3, kv.first & 7, kv.second[kv.second.size() / 2]); }
        printf("\n");
    }
    // when do tiles start / end (0.5 us buckets), and which block ids start when
    std::map<int, int> sh, eh; std::map<int, std::pair<int,int>> ids;
    for (int t = 0; t < nt; ++t) {
        const int bs = (int)((st[t * 64] - t0) / 50), be = (int)((st[t * 64 + 6] - t0) / 50);
        sh[bs]++; eh[be]++;
        if (!ids.count(bs)) ids[bs] = {t, t}; else { ids[bs].first = std::min(ids[bs].first, t); ids[bs].second = std::max(ids[bs].second, t); }
    }
#endif
#ifndef SBM_NO_QSTAMP
    printf("starts per 0.5us:"); for (auto& h : sh) printf(" %.1f:%d(id %d..%d)", h.first * 0.5, h.second, ids[h.first].first, ids[h.first].second); printf("\n");
    printf("ends per 0.5us:"); for (auto& h : eh) printf(" %.1f:%d", h.first * 0.5, h.second); printf("\n");
#endif
    return 0;
}
