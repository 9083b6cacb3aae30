// What does the table layout cost the sweep's streaming part?  Same bytes per row (advance: read the
// trajectory, write the position), three layouts, no radar work:
//   A  planar 8-byte columns: alive u8, lidx i32, t0, vel[3], sp[3] -> pos[3]      (12 loads, 3 stores / row)
//   B  16-byte pair planes: (sx,sy) (sz,vx) (vy,vz) (t0,meta) -> pos[3] planar      (4 loads, 3 stores / row)
//   C  pair planes -> (x,y) pair plane + z plane                                  (4 loads, 2 stores / row)
//   D  B with two rows per lane
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o stream_layouts stream_layouts.hip ; run: ./stream_layouts [n]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_planar(const double *__restrict__ sp, const double *__restrict__ vel,
                                                const double *__restrict__ t0, const uint8_t *__restrict__ alive,
                                                const int32_t *__restrict__ lidx, double *__restrict__ pos,
                                                uint32_t *__restrict__ vis, int64_t n, int64_t cap, double t)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t ic = i < n ? i : 0;
    const uint8_t al = alive[ic];
    const int32_t li = lidx[ic];
    const double t0v = t0[ic], vx = vel[ic], vy = vel[cap + ic], vz = vel[2 * cap + ic];
    const double sx = sp[ic], sy = sp[cap + ic], sz = sp[2 * cap + ic];
    const double d = t - t0v;
    const double x = sx + vx * d, y = sy + vy * d, z = sz + vz * d;
    if (i < n && al) {
        pos[i] = x; pos[cap + i] = y; pos[2 * cap + i] = z;
        if (x == 1.2345e300) vis[li] = 1u;     // keeps li alive
    }
}

struct alignas(16) D2 { double a, b; };

template <int NR, bool PAIR_OUT>
__global__ __launch_bounds__(256) void k_pairs(const D2 *__restrict__ p0, const D2 *__restrict__ p1,
                                               const D2 *__restrict__ p2, const D2 *__restrict__ p3,
                                               double *__restrict__ pos, D2 *__restrict__ posxy,
                                               uint32_t *__restrict__ vis, int64_t n, int64_t cap, double t)
{
    D2 a[NR], b[NR], c[NR], m[NR];
    int64_t i[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        i[j] = ((int64_t)blockIdx.x * NR + j) * 256 + threadIdx.x;
        const int64_t ic = i[j] < n ? i[j] : 0;
        a[j] = p0[ic]; b[j] = p1[ic]; c[j] = p2[ic]; m[j] = p3[ic];
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const unsigned long long meta = __builtin_bit_cast(unsigned long long, m[j].b);
        const int32_t li = (int32_t)(meta & 0xFFFFFFFFull);
        const bool al = ((meta >> 32) & 0xFF) != 0;
        const double d = t - m[j].a;
        const double x = a[j].a + b[j].b * d, y = a[j].b + c[j].a * d, z = b[j].a + c[j].b * d;
        if (i[j] < n && al) {
            if (PAIR_OUT) { posxy[i[j]] = D2{x, y}; pos[2 * cap + i[j]] = z; }
            else { pos[i[j]] = x; pos[cap + i[j]] = y; pos[2 * cap + i[j]] = z; }
            if (x == 1.2345e300) vis[li] = 1u;
        }
    }
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int64_t cap = (n + 255) / 256 * 256;
    double *sp, *vel, *t0, *pos[2];
    uint8_t *alive; int32_t *lidx; uint32_t *vis;
    D2 *p[4], *posxy[2];
    CK(hipMalloc(&sp, 24 * cap)); CK(hipMalloc(&vel, 24 * cap)); CK(hipMalloc(&t0, 8 * cap));
    CK(hipMalloc(&alive, cap)); CK(hipMalloc(&lidx, 4 * cap)); CK(hipMalloc(&vis, 4 * cap));
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&pos[k], 24 * cap)); CK(hipMalloc(&posxy[k], 16 * cap)); }
    for (int k = 0; k < 4; ++k) CK(hipMalloc(&p[k], 16 * cap));
    CK(hipMemset(sp, 0, 24 * cap)); CK(hipMemset(vel, 0, 24 * cap)); CK(hipMemset(t0, 0, 8 * cap));
    CK(hipMemset(alive, 1, cap)); CK(hipMemset(lidx, 0, 4 * cap));
    {
        std::vector<D2> h(cap);
        for (int64_t i = 0; i < cap; ++i) { h[i].a = 0.0; unsigned long long meta = (1ull << 32) | (unsigned)i; h[i].b = __builtin_bit_cast(double, meta); }
        CK(hipMemcpy(p[3], h.data(), 16 * cap, hipMemcpyHostToDevice));
        for (int k = 0; k < 3; ++k) CK(hipMemset(p[k], 0, 16 * cap));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 400;
    auto time = [&](const char *name, auto launch) {
        for (int k = 0; k < 50; ++k) launch(k);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int k = 0; k < reps; ++k) launch(k);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %7.2f us / launch (back to back, boundaries included)\n", name, ms / reps * 1e3);
    };
    const int nb = (int)(cap / 256);
    time("A planar 8-byte columns, 12 loads 3 stores", [&](int k) { hipLaunchKernelGGL(k_planar, dim3(nb), dim3(256), 0, 0, sp, vel, t0, alive, lidx, pos[k & 1], vis, n, cap, 0.01 * k); });
    time("B pair planes, 4 loads 3 stores", [&](int k) { hipLaunchKernelGGL((k_pairs<1, false>), dim3(nb), dim3(256), 0, 0, p[0], p[1], p[2], p[3], pos[k & 1], posxy[k & 1], vis, n, cap, 0.01 * k); });
    time("C pair planes, 4 loads, (x,y) pair + z stores", [&](int k) { hipLaunchKernelGGL((k_pairs<1, true>), dim3(nb), dim3(256), 0, 0, p[0], p[1], p[2], p[3], pos[k & 1], posxy[k & 1], vis, n, cap, 0.01 * k); });
    time("D pair planes, two rows per lane, 3 stores", [&](int k) { hipLaunchKernelGGL((k_pairs<2, false>), dim3((nb + 1) / 2), dim3(256), 0, 0, p[0], p[1], p[2], p[3], pos[k & 1], posxy[k & 1], vis, n, cap, 0.01 * k); });
    time("E pair planes, two rows per lane, pair stores", [&](int k) { hipLaunchKernelGGL((k_pairs<2, true>), dim3((nb + 1) / 2), dim3(256), 0, 0, p[0], p[1], p[2], p[3], pos[k & 1], posxy[k & 1], vis, n, cap, 0.01 * k); });
    time("F pair planes, four rows per lane, pair stores", [&](int k) { hipLaunchKernelGGL((k_pairs<4, true>), dim3((nb + 3) / 4), dim3(256), 0, 0, p[0], p[1], p[2], p[3], pos[k & 1], posxy[k & 1], vis, n, cap, 0.01 * k); });
    return 0;
}
