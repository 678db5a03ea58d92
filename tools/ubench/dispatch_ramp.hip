// Microbenchmark: how long does the dispatcher take to hand out the waves of a launch, as a function of the workgroup's size?
// Every wave stamps s_memrealtime (100 MHz, device-wide) when it starts and spins for `spin_us`; 4096 waves with 7936 B of LDS and
// ~86 VGPRs each (the step kernel's footprint) as 4096 x 64, 2048 x 128, 1024 x 256 threads.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench/dispatch_ramp tools/ubench/dispatch_ramp.hip ; run: tools/ubench/dispatch_ramp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(unsigned long long* out, int spin_ticks, float* sink) {
    extern __shared__ char lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int wave = threadIdx.x >> 6, gw = blockIdx.x * WAVES + wave;
    float acc[40];  // (register footprint)
#pragma unroll
    for (int i = 0; i < 40; i++) acc[i] = (float)(threadIdx.x + i);
    lds[wave * 7936 + (threadIdx.x & 63)] = (char)t0;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) {
#pragma unroll
        for (int i = 0; i < 40; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 40; i++) s += acc[i];
    if (s == 12345.678f) sink[0] = s + lds[wave * 7936];
    if ((threadIdx.x & 63) == 0) { out[2 * gw] = t0; out[2 * gw + 1] = __builtin_amdgcn_s_memrealtime(); }
}
template <int WAVES> void run(int n_waves, int spin_us, unsigned long long* d, float* sink) {
    std::vector<unsigned long long> h(2 * (size_t)n_waves);
    double ramp99 = 0, ramp = 0, span = 0;
    const int reps = 50;
    for (int r = 0; r < reps + 5; r++) {
        hipLaunchKernelGGL(k<WAVES>, dim3(n_waves / WAVES), dim3(64 * WAVES), 7936 * WAVES, 0, d, spin_us * 100, sink);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        if (r < 5) continue;
        std::vector<unsigned long long> st(n_waves);
        unsigned long long mn = ~0ull, mx = 0;
        for (int i = 0; i < n_waves; i++) { st[i] = h[2 * i]; mn = std::min(mn, st[i]); mx = std::max(mx, h[2 * i + 1]); }
        std::sort(st.begin(), st.end());
        ramp99 += (st[(size_t)(0.99 * n_waves)] - mn) * 0.01; ramp += (st.back() - mn) * 0.01; span += (mx - mn) * 0.01;
    }
    printf("%5d waves as %4d x %3d threads, %d us of work each: 99 %% of the waves started after %.2f us, all after %.2f us; first start -> last end %.2f us\n",
           n_waves, n_waves / WAVES, 64 * WAVES, spin_us, ramp99 / reps, ramp / reps, span / reps);
}
int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 2 * 8192 * 8); hipMalloc(&sink, 4);
    for (int spin : {0, 20}) {
        run<1>(4096, spin, d, sink); run<2>(4096, spin, d, sink); run<4>(4096, spin, d, sink); run<8>(4096, spin, d, sink);
    }
    return 0;
}
