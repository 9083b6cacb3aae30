// How fast does the dispatcher refill a device that is full?  A grid of W waves in workgroups of T threads, every wave holding
// `regs` vector registers and staying resident for `us` microseconds (it polls the wall clock): with S wave slots the launch lasts
// ceil(W / S) * us at best.  What it lasts beyond that is the dispatcher's.   usage: dispatch_rate_probe [waves] [us]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned long long wall() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

template <int T, int REGS>
__global__ __launch_bounds__(T) void k_stay(int ticks_short, int ticks_long, int long_every, float *sink)
{
    // hold REGS vector registers
    float v[REGS];
#pragma unroll
    for (int k = 0; k < REGS; ++k) v[k] = (float)(threadIdx.x + k);
    // long_every > 0: every long_every-th workgroup stays long;  long_every < 0: in EVERY workgroup the first half of the waves stays
    // long, the second half short (is a finished wave's slot given to the next workgroup before its own workgroup has ended?)
    const int ticks = long_every > 0 ? ((blockIdx.x % long_every) == 0 ? ticks_long : ticks_short)
                    : long_every < 0 ? ((threadIdx.x < T / 2) ? ticks_long : ticks_short) : ticks_short;
    const unsigned long long t0 = wall();
    while (wall() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < REGS; ++k) { asm volatile("" : "+v"(v[k])); acc += v[k]; }
    if (acc == -1.f) sink[0] = acc;
}

template <int T>
static float run(int waves, int us_short, int us_long, int long_every, float *sink)
{
    const int wgs = waves / (T / 64);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_stay<T, 56>), dim3(wgs), dim3(T), 0, 0, us_short * 100, us_long * 100, long_every, sink);
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 10; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_stay<T, 56>), dim3(wgs), dim3(T), 0, 0, us_short * 100, us_long * 100, long_every, sink);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best; sum += ms;
    }
    (void)sum;
    return best * 1000.f;
}

int main(int argc, char **argv)
{
    const int waves = argc > 1 ? atoi(argv[1]) : 15744;
    float *sink; hipMalloc(&sink, 4);
    int dev = 0; hipDeviceProp_t pr; hipGetDeviceProperties(&pr, dev);
    printf("%s: %d CUs\n", pr.name, pr.multiProcessorCount);
    for (int us : {2, 4, 7, 10}) {
        printf("waves %d, every wave resident %d us (56+ VGPRs):  T=64 %.1f us   T=256 %.1f us   T=512 %.1f us   T=1024 %.1f us\n", waves, us,
               run<64>(waves, us, us, 0, sink), run<256>(waves, us, us, 0, sink), run<512>(waves, us, us, 0, sink), run<1024>(waves, us, us, 0, sink));
    }
    // a quarter of the workgroups stay 16 us, the others 6 us (the sweep's mix)
    printf("mix: every 4th workgroup 16 us, the others 6 us:  T=256 %.1f us   T=512 %.1f us\n", run<256>(waves, 6, 16, 4, sink), run<512>(waves, 6, 16, 4, sink));
    printf("every workgroup: half its waves 16 us, half 6 us (slot time says %.1f us + launch; whole-workgroup release %.1f us + launch):  T=256 %.1f us   T=512 %.1f us\n",
           waves * 11.0 / 8192, waves * 16.0 / 8192, run<256>(waves, 6, 16, -1, sink), run<512>(waves, 6, 16, -1, sink));
    printf("all waves 11 us:  T=256 %.1f us;  all waves 16 us:  T=256 %.1f us\n", run<256>(waves, 11, 11, 0, sink), run<256>(waves, 16, 16, 0, sink));
    printf("empty-ish (1 us):  T=256 %.1f us   T=512 %.1f us   T=1024 %.1f us\n", run<256>(waves, 1, 1, 0, sink), run<512>(waves, 1, 1, 0, sink), run<1024>(waves, 1, 1, 0, sink));
    return 0;
}
