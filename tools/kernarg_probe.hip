// What does a large kernel-argument segment cost a launch?  Empty kernels with 64 B / 4.4 KB / 8.8 KB of arguments,
// back to back on one stream, wall time per launch; and the same with a busy second stream beside them.
//   hipcc -O3 --offload-arch=gfx950 -o kernarg_probe kernarg_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <int N> struct Blob { unsigned w[N]; };
template <int N> __global__ void k_args(const Blob<N> b, unsigned *out) { if (b.w[threadIdx.x % N] == 0xdeadbeefu) out[0] = 1; }
template <int N> __global__ void k_noread(const Blob<N> b, unsigned *out) { if (threadIdx.x == 5000) out[0] = b.w[0]; }
template <int N> __global__ void k_copy(const Blob<N> b, unsigned *out) { for (int k = threadIdx.x; k < N; k += blockDim.x) out[k] = b.w[k]; }
__global__ void k_busy(double *a, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i] * 1.0000001 + 1.0; }
template <class F> double per_launch(hipStream_t s, F f, int reps = 2000)
{
    for (int k = 0; k < 100; ++k) f();
    hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; ++k) f();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}
int main()
{
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *out; hipMalloc(&out, 1 << 16);
    static Blob<16> b0; static Blob<1100> b1; static Blob<2200> b2;
    printf("empty kernel, 12 workgroups of 1024: %5.2f us (64 B args) %5.2f us (4.4 KB) %5.2f us (8.8 KB)\n",
           per_launch(s, [&] { hipLaunchKernelGGL(k_args<16>, dim3(12), dim3(1024), 0, s, b0, out); }),
           per_launch(s, [&] { hipLaunchKernelGGL(k_args<1100>, dim3(12), dim3(1024), 0, s, b1, out); }),
           per_launch(s, [&] { hipLaunchKernelGGL(k_args<2200>, dim3(12), dim3(1024), 0, s, b2, out); }));
    printf("empty kernel that never looks at them:       %5.2f us (64 B args) %5.2f us (4.4 KB) %5.2f us (8.8 KB)\n",
           per_launch(s, [&] { hipLaunchKernelGGL(k_noread<16>, dim3(12), dim3(1024), 0, s, b0, out); }),
           per_launch(s, [&] { hipLaunchKernelGGL(k_noread<1100>, dim3(12), dim3(1024), 0, s, b1, out); }),
           per_launch(s, [&] { hipLaunchKernelGGL(k_noread<2200>, dim3(12), dim3(1024), 0, s, b2, out); }));
    printf("copying the arguments to memory, 1 workgroup:  %5.2f us (4.4 KB) %5.2f us (8.8 KB)\n",
           per_launch(s, [&] { hipLaunchKernelGGL(k_copy<1100>, dim3(1), dim3(1024), 0, s, b1, out); }),
           per_launch(s, [&] { hipLaunchKernelGGL(k_copy<2200>, dim3(1), dim3(1024), 0, s, b2, out); }));
    return 0;
}
