// Where does the single-launch compaction spend its time?  Wall-clock stamps (s_memrealtime, 100 MHz)
// at the phase boundaries of k_compact_fused, per workgroup.  Includes the product source with the probe
// macro defined; nothing here ships.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -o /tmp/compact_phases tools/compact_phases.hip
//   /tmp/compact_phases [n] [R] [items] [by_ticket]
#include <hip/hip_runtime.h>
__device__ long long *g_probe;
#define ZRK_PROBE(slot) do { if (threadIdx.x == 0 && g_probe) g_probe[(long long)S.ticket * 8 + (slot)] = wall_clock64(); } while (0)
#include "../zrk_modulation_amd/csrc/zrk_hot.hip"
#include <random>
#include <vector>

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int R = argc > 2 ? atoi(argv[2]) : 16;
    const int items = argc > 3 ? atoi(argv[3]) : 2;
    const int by_ticket = argc > 4 ? atoi(argv[4]) : 1;
    const int nb = (int)((n + 1024LL * items - 1) / (1024LL * items));
    std::mt19937_64 g(1);
    std::vector<uint32_t> vis(n, 0);
    for (int64_t i = 0; i < n; ++i)
        if (g() % 100 < 17) {                       // list order is spatially random: detections are spread evenly
            vis[i] = (uint32_t)(g() & g() & g() & g()) & ((R < 32) ? ((1u << R) - 1) : ~0u);
            if (!vis[i]) vis[i] = 1u << (g() % R);
        }
    uint32_t *dvis, *dzero; int32_t *det, *cnt; int64_t *packed; void *ws; long long *probe;
    hipMalloc(&dvis, n * 4); hipMalloc(&dzero, n * 4); hipMalloc(&det, n * R * 4); hipMalloc(&cnt, 256); hipMalloc(&packed, (n + 1) * 8);
    const int64_t wsb = zrk_workspace_bytes(n);
    hipMalloc(&ws, wsb); hipMemset(ws, 0, wsb);
    hipMalloc(&probe, (int64_t)nb * 8 * 8); hipMemset(probe, 0, (int64_t)nb * 8 * 8);
    hipMemcpy(dvis, vis.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    Workspace w = carve(ws, 0, n);
    int lanes = 1; while (lanes < R + 1) lanes <<= 1;
    MissileArgs M = no_missiles();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 30; ++rep) {
        if (rep == 29) hipEventRecord(e0, 0);
        CompactArgs C;
        C.vis = dvis; C.zero_next = dzero; C.n = n; C.R = R; C.nb = nb; C.items = items; C.lanes = lanes; C.epoch = (uint32_t)(rep + 1);
        C.base_index = 0; C.ctl = w.ctl; C.agg = w.agg; C.det_idx = det; C.det_stride = n; C.det_cnt = cnt; C.packed = packed;
        C.packed_capacity = n + 1; C.gid0 = 0;
        EnsembleArgs E; std::memset(&E, 0, sizeof(E));
        static PutArgs U;
        C.seg_blocks = 0; C.zero_own = 0; C.seg_slots = 0;
        hipLaunchKernelGGL(k_compact_fused, dim3(nb), dim3(kCompBlock), 0, 0, C, by_ticket, M, E, U);
        if (rep == 29) hipEventRecord(e1, 0);
    }
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> t((size_t)nb * 8);
    hipMemcpy(t.data(), probe, t.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = t[0];
    for (int b = 0; b < nb; ++b) t0 = std::min(t0, t[b * 8]);
    printf("n=%lld R=%d items=%d nb=%d by_ticket=%d: last launch %.2f us by events\n", (long long)n, R, items, nb, by_ticket, ms * 1000);
    const char *name[6] = {"ticket taken", "masks loaded", "squeezed", "counted", "predecessors summed", "scattered"};
    for (int k = 0; k < 6; ++k) {
        std::vector<double> v(nb);
        for (int b = 0; b < nb; ++b) v[b] = (t[b * 8 + k] - t0) * 0.01;
        std::vector<double> s = v; std::sort(s.begin(), s.end());
        printf("  %-22s min %6.2f  median %6.2f  max %6.2f us   (wg 0: %6.2f, wg %d: %6.2f)\n", name[k], s.front(), s[nb / 2], s.back(), v[0], nb - 1, v[nb - 1]);
    }
    return 0;
}
