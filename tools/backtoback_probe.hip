// What stands between two kernels that follow each other on ONE stream (the overlapped loop's side stream): the same
// 17-us kernel launched N times back to back, (a) plainly, (b) with a hipEventRecord behind every launch (what the side
// stream's thread does for the ring's back-pressure), (c) with 640 bytes of arguments (k_compact_pair's), (d) both, (e) the
// launches released one by one by a host thread that polls a word (as the side stream's are).  Prints the stream's time
// per launch minus the kernel's own duration.
//
//   make -C tools backtoback_probe && tools/backtoback_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Fat { unsigned long long w[80]; };

__global__ __launch_bounds__(1024) void k_work(const uint32_t *__restrict__ src, uint32_t *dst, long n, int rounds)
{
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r)
        for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n; i += (long)gridDim.x * 1024) acc += src[i] ^ (uint32_t)r;
    if (acc == 0x12345678u) dst[0] = acc;
}

__global__ __launch_bounds__(1024) void k_work_fat(const uint32_t *__restrict__ src, uint32_t *dst, long n, int rounds, const Fat f)
{
    uint32_t acc = (uint32_t)f.w[threadIdx.x % 80];
    for (int r = 0; r < rounds; ++r)
        for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n; i += (long)gridDim.x * 1024) acc += src[i] ^ (uint32_t)r;
    if (acc == 0x12345678u) dst[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const long n = 2 << 20;                                    // 8 MB
    uint32_t *src, *dst;
    CK(hipMalloc(&src, n * 4)); CK(hipMalloc(&dst, 64)); CK(hipMemset(src, 1, n * 4));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int N = 40, grid = 247;
    int rounds = 4;
    std::vector<hipEvent_t> ev(N);
    for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    Fat f{};
    // the kernel's own duration
    float one = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(t0, s));
        hipLaunchKernelGGL(k_work, dim3(grid), dim3(1024), 0, s, src, dst, n, rounds);
        CK(hipEventRecord(t1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&one, t0, t1));
    }
    std::printf("one launch between two timing events: %.1f us\n", one * 1e3);
    for (int variant = 0; variant < 4; ++variant) {
        const bool with_event = variant & 1, fat = variant & 2;
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipStreamSynchronize(s));
            const auto h0 = std::chrono::steady_clock::now();
            for (int k = 0; k < N; ++k) {
                if (fat) hipLaunchKernelGGL(k_work_fat, dim3(grid), dim3(1024), 0, s, src, dst, n, rounds, f);
                else hipLaunchKernelGGL(k_work, dim3(grid), dim3(1024), 0, s, src, dst, n, rounds);
                if (with_event) CK(hipEventRecord(ev[k], s));
            }
            CK(hipStreamSynchronize(s));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
            best = us < best ? us : best;
        }
        std::printf("%-28s %6.1f us per launch (%d launches back to back, host clock around launch..sync)\n",
                    variant == 0 ? "plain" : variant == 1 ? "event behind each" : variant == 2 ? "640 B of arguments" : "both", best / N, N);
    }
    return 0;
}
