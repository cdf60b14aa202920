// layout_probe.cpp -- runs the BP kernels' host-side layout searches on the CPU and prints the modelled LDS cost of the
// bit pass under the measured banking rules (tools/microbench/lds_scatter_probe.hip):
//   ds_read_b64  = sum over half-waves of the largest number of distinct slots on one 8-byte column mod 32
//   ds_write_b64 = max(6, sum over quarter-waves of the largest number of distinct slots on one column mod 16)
//   g++ -O2 -std=c++17 -pthread tools/layout_probe.cpp -o build/layout_probe && build/layout_probe pcm.csr [local|class DC DVLO DVHI MP]
// pcm.csr: "m n nnz", indptr, indices (text).
#include <cstdio>
#include <cstring>
#include <chrono>
#include "../bp_osd_amd/csrc/local_layout.h"
#include "../bp_osd_amd/csrc/class_layout.h"

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "r");
    int m, n, nnz;
    if (!f || fscanf(f, "%d %d %d", &m, &n, &nnz) != 3) return 1;
    std::vector<int> rp(m + 1), ci(nnz);
    for (auto& v : rp) if (fscanf(f, "%d", &v) != 1) return 1;
    for (auto& v : ci) if (fscanf(f, "%d", &v) != 1) return 1;
    fclose(f);
    const auto t0 = std::chrono::steady_clock::now();
    if (argc >= 7 && !strcmp(argv[2], "class")) {
        const int DC = atoi(argv[3]), DVLO = atoi(argv[4]), DVHI = atoi(argv[5]), MP = atoi(argv[6]);
        const int iters = argc > 7 ? atoi(argv[7]) : 200000;
        class_layout::Tables T;
        if (!class_layout::build(rp, ci, m, n, getenv("DCLO") ? atoi(getenv("DCLO")) : DC, DC, DVLO, DVHI, 2, MP, MP, iters, T)) { printf("no class layout\n"); return 1; }
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("class layout: %.2f s, %d threads, read cycles %ld (floor %ld), write cycles %ld (floor %ld); group degrees:", sec, T.NT,
               T.read_cycles, T.read_floor, T.write_cycles, T.write_floor);
        for (int r = 0; r < T.VPT; ++r) { printf(" |"); for (int w = 0; w < T.NT / 64; ++w) printf(" %d", T.grp_deg[r * (T.NTMAX / 64) + w]); }
        printf("\n");
        return 0;
    }
    const int MP = m <= 1024 ? 1024 : 2048;
    local_layout::Graph g;
    local_layout::Layout best;
    if (!local_layout_host(rp, ci, m, n, MP, g, best)) { printf("no layout\n"); return 1; }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const local_layout::LdsCost c = local_layout::lds_cost(g, best);
    printf("layout: %.2f s, read cycles %d (ideal %d), write cycles %d (ideal %d), mixed (group, slot) pairs %d, uniform positions %d\n", sec,
           c.read_cycles, 4 * (MP / 32), c.write_cycles, 6 * 4 * (MP / 64), c.mixed, best.nfull);
    return 0;
}
