// Host-side sanitizer run of the library's host code (no GPU needed, none touched): the layout / sizing functions that
// decide every buffer offset of a view, and the host walk of merge_nearby_gaussians.  Built by `make san_host` with
// AddressSanitizer + UBSan on the HOST side only (-Xarch_host; GPU sanitizers are not available on this pool), linked
// against the same objects as the library.  Any report fails the run.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/gsx.h"
#include "../../include/gsx_constants.h"

static uint32_t lcg = 777u;
static uint32_t rnd() { lcg = lcg * 1664525u + 1013904223u; return lcg >> 8; }

int main()
{
    int bad = 0;
    // every buffer of a view lies inside the block, 256-byte aligned, no two persistent ones overlap
    const int sizes[][4] = { {1, 1, 1, 1}, {5000, 300, 400, 20000}, {1000000, 1024, 1024, 4000000}, {6000000, 1080, 1920, 300000000}, {7, 65535, 3, 100} };
    for (const auto& c : sizes)
        for (int need_bwd = 0; need_bwd <= 1; need_bwd++)
            for (int force : {0, 64, 128, 256}) {
                int64_t off[GSX_VIEW_SLOTS + 1]; int64_t max_chunks = 0;
                const int64_t total = gsx_view_layout(c[0], c[1], c[2], c[3], force, need_bwd, off, &max_chunks);
                if (total <= 0 || max_chunks <= 0) { std::printf("layout refused %d %d %d %d\n", c[0], c[1], c[2], c[3]); bad = 1; continue; }
                for (int k = 0; k < GSX_VIEW_SLOTS; k++) {
                    if (off[k] == -1) continue;                      // not present in this configuration
                    if (off[k] < 0 || off[k] > total || (off[k] & 255)) { std::printf("slot %d at %lld of %lld\n", k, (long long)off[k], (long long)total); bad = 1; }
                }
                for (int k = 0; k + 1 <= GSX_VIEW_TOTAL; k++) if (off[k] >= 0 && off[k + 1] >= 0 && off[k + 1] < off[k]) { std::printf("persistent slots out of order at %d\n", k); bad = 1; }
                if (gsx_max_chunks(c[3], c[1], c[2], force) != max_chunks) { std::printf("max_chunks disagree\n"); bad = 1; }
            }
    for (int64_t n : {0ll, 1ll, 4096ll, 1000000ll, 300000000ll}) {
        if (gsx_sort_workspace_bytes(n) == 0 || gsx_scan_workspace_bytes(n) == 0 || gsx_merge_workspace_bytes((int32_t)n) == 0) { std::printf("zero workspace at %lld\n", (long long)n); bad = 1; }
    }
    // the host walk of the merge on random pair lists (exactly-sized vectors: an overrun is a report)
    for (int trial = 0; trial < 50; trial++) {
        const int N = 1 + (int)(rnd() % 300);
        std::vector<uint32_t> offsets(N + 1), partners;
        for (int i = 0; i < N; i++) {
            offsets[i] = (uint32_t)partners.size();
            uint32_t j = (uint32_t)i + (rnd() % 3);                  // may start at i itself: self pairs are legal input
            for (int k = rnd() % 4; k > 0 && j < (uint32_t)N; k--, j += 1 + rnd() % 5) partners.push_back(j);
        }
        offsets[N] = (uint32_t)partners.size();
        std::vector<float> pos(3 * N), fea(3 * N), opa(N); std::vector<uint8_t> removed(N);
        for (auto& v : pos) v = (float)rnd() / 1e6f; for (auto& v : fea) v = (float)rnd() / 1.7e7f; for (auto& v : opa) v = (float)rnd() / 1.7e7f;
        const int st = gsx_merge_greedy_host(N, offsets.data(), partners.empty() ? nullptr : partners.data(), pos.data(), fea.data(), opa.data(), removed.data());
        if (st != 0 && !(partners.empty())) { std::printf("walk status %d\n", st); bad = 1; }
        if (trial == 0) {                                              // a bad pair is refused, nothing is read past the lists
            std::vector<uint32_t> o2 = {0, 1}, p2 = {9};
            if (gsx_merge_greedy_host(1, o2.data(), p2.data(), pos.data(), fea.data(), opa.data(), removed.data()) == 0) { std::printf("bad pair accepted\n"); bad = 1; }
        }
    }
    std::printf("san_host: %s\n", bad ? "FAILED" : "ok");
    return bad;
}
