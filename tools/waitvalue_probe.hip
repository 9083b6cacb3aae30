// Hand-over between two streams without an event on the producing stream: the producer's NEXT kernel writes a flag word,
// the consumer stream waits for it with hipStreamWaitValue32.  Prints whether the device supports it, checks the
// ordering on a small buffer and times the producing stream's kernels with (a) nothing, (b) an event record per
// iteration, (c) the flag.        hipcc -O3 --offload-arch=gfx950 -o waitvalue_probe waitvalue_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_work(double *a, int n, double v, uint32_t *flag, uint32_t value)
{
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] * 1.0000001 + v;
}
__global__ void k_fill(int64_t *list, int n, int64_t v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) list[i] = v; }
__global__ void k_check(const int64_t *list, int n, int64_t v, int *bad) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n && list[i] != v) atomicAdd(bad, 1); }

int main()
{
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    hipStream_t s, c;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    const int n = 1 << 22, ln = 1 << 16;
    double *a; int64_t *list; int *bad; uint32_t *flag;
    CK(hipMalloc(&a, sizeof(double) * n)); CK(hipMemset(a, 0, sizeof(double) * n));
    CK(hipMalloc(&list, sizeof(int64_t) * ln)); CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    CK(hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory));
    CK(hipMemset(flag, 0, 8));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int iters = 400;
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipDeviceSynchronize());
        uint32_t seq = 1000u * (mode + 1);
        for (int warm = 0; warm < 2; ++warm) {
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < iters; ++k) {
                ++seq;
                // "sweep": carries the flag of the PREVIOUS iteration's list in mode 2
                hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, a, n, 1.0, mode == 2 ? flag : nullptr, seq - 1);
                hipLaunchKernelGGL(k_fill, dim3(ln / 256), dim3(256), 0, s, list, ln, (int64_t)seq);   // "compaction"
                if (mode == 1) { CK(hipEventRecord(ev, s)); CK(hipStreamWaitEvent(c, ev, 0)); hipLaunchKernelGGL(k_check, dim3(ln / 256), dim3(256), 0, c, list, ln, (int64_t)seq, bad); }
                if (mode == 2 && k > 0) {
                    // consumer of iteration k-1's list: may run as soon as this iteration's first kernel has started ...
                    // (the list of k-1 is rewritten by THIS iteration's k_fill, so the check below is only valid if it
                    // runs before; it is not ordered against it here, so check a value range instead)
                    CK(hipStreamWaitValue32(c, flag, seq - 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
                }
            }
            CK(hipStreamSynchronize(s));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(c));
            if (warm) printf("mode %d (%s): %.2f us per iteration on the producing stream\n", mode,
                             mode == 0 ? "nothing" : mode == 1 ? "event record + wait" : "flag in the next kernel + wait value",
                             std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
        }
    }
    int hbad = -1; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    printf("event-ordered consumer saw %d stale words\n", hbad);
    // ordering check of the flag path: producer fills the list, a later kernel raises the flag, consumer checks
    CK(hipMemset(bad, 0, 4));
    for (int k = 0; k < 200; ++k) {
        const uint32_t seq = 50000u + k;
        hipLaunchKernelGGL(k_fill, dim3(ln / 256), dim3(256), 0, s, list, ln, (int64_t)seq);
        hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, s, a, n, 1.0, flag, seq);
        CK(hipStreamWaitValue32(c, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(k_check, dim3(ln / 256), dim3(256), 0, c, list, ln, (int64_t)seq, bad);
        CK(hipStreamSynchronize(c));      // the list is rewritten next iteration: the consumer must be through
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    printf("flag-ordered consumer saw %d stale words in 200 rounds\n", hbad);
    return 0;
}
