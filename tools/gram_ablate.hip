// Timing harness of the f64 whitening-learning GEMMs (mdx_gram_f64 / mdx_project_f64) at D = 2048, n = 20 000 with the
// kernel's compile-time shape and timing switches:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DMDX_GRAM_ABL=5] [-DMDX_GRAM_LW=4 -DMDX_GRAM_LC_GK=16 -DMDX_GRAM_LC_NSTAGE=4]
//         [-DMDX_GRAM_PLAIN_MAP] -I mdir_amd/csrc -I include tools/gram_ablate.hip -o tools/gram_ablate_bin_<name>
//   MDX_GRAM_ABL=5: the loaders issue nothing inside the loop (what the operand supply costs; results wrong by construction);
//   MDX_GRAM_LW / _LC_GK / _LC_NSTAGE: loader waves, k per stage, stages; MDX_GRAM_PLAIN_MAP: workgroup id = tile position.
// (Round 6: MDX_GRAM_ABL / MDX_GRAM_PLAIN_MAP left mdx_gram.hip; apply tools/ablate/scores_kernel_ablate.patch to a scratch copy of
// mdir_amd/csrc and include that copy to get them back.)
// Only the default build's result is checked (a host sum over a few entries).  Numbers: the header of mdx_gram.hip.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../mdir_amd/csrc/mdx_gram.hip"
namespace mdx { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }

int main(int argc, char **argv)
{
    const int64_t d = argc > 1 ? atoll(argv[1]) : 2048, n = argc > 2 ? atoll(argv[2]) : 20000;
    std::vector<double> ha((size_t)d * n), hp((size_t)d * d);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(int64_t)(s >> 11) / 9007199254740992.0 - 0.5; };
    for (auto &v : ha) v = rnd();
    for (auto &v : hp) v = rnd();
    double *a, *p, *g, *y; void *ws;
    const int64_t wsb = std::max(mdx_gram_f64_workspace(d, n), mdx_project_f64_workspace(d, d));
    hipMalloc(&a, ha.size() * 8); hipMalloc(&p, hp.size() * 8); hipMalloc(&g, d * d * 8); hipMalloc(&y, d * n * 8); hipMalloc(&ws, wsb);
    hipMemcpy(a, ha.data(), ha.size() * 8, hipMemcpyHostToDevice); hipMemcpy(p, hp.data(), hp.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timed = [&](auto fn) { fn(); fn(); hipEventRecord(e0, 0); for (int i = 0; i < 10; ++i) fn(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10; };
    const float tg = timed([&]() { mdx_gram_f64(a, d, n, nullptr, g, ws, wsb, nullptr); });
    const float tp = timed([&]() { mdx_project_f64(p, d, d, a, n, nullptr, y, ws, wsb, nullptr); });
    const int64_t tile = d >= 1024 ? 128 : 64, t = (d + tile - 1) / tile;
    printf("ABL %d  d %lld n %lld  gram %.3f ms (%.3f of 78.6 TF)  project %.3f ms (%.3f)\n", MDX_GRAM_ABL, (long long)d, (long long)n, tg,
           2.0 * n * tile * tile * (t * (t + 1) / 2) / tg / 1e9 / 78.6, tp, 2.0 * d * d * n / tp / 1e9 / 78.6);
    if (MDX_GRAM_ABL == 0) {
        std::vector<double> hg((size_t)d * d), hy(64);
        hipMemcpy(hg.data(), g, hg.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int e = 0; e < 64; ++e) {
            const int64_t i = (e * 131) % d, j = (e * 977 + 5) % d;
            double ref = 0; for (int64_t k = 0; k < n; ++k) ref += ha[i * n + k] * ha[j * n + k];
            worst = std::max(worst, std::abs(ref - hg[i * d + j]));
        }
        printf("gram max abs diff vs host on 64 entries: %.3e\n", worst);
    }
    return 0;
}
