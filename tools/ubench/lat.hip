// Latency probe (diagnostic): one wave measures, with s_memtime, (a) a dependent global load that misses every cache,
// (b) one that hits in L2, (c) a global store followed by s_waitcnt vmcnt(0), (d) an LDS read, each averaged over N.
// Optionally `bg` background waves stream through memory to load the fabric.  hipcc --offload-arch=gfx950 lat.hip -o lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void probe(const uint32_t* chase, uint32_t* scratch, uint64_t* out, int n, size_t stride_words, uint32_t* sink, size_t bg_words) {
    if (blockIdx.x > 0) {  // background traffic
        uint32_t acc = 0;
        for (int rep = 0; rep < 64; rep++)
            for (size_t i = (size_t)(blockIdx.x - 1) * 64 + threadIdx.x + rep * 7919; i < bg_words; i += (size_t)(gridDim.x - 1) * 64 * 13) acc += sink[i];
        if (acc == 0x12345678) sink[0] = acc;
        return;
    }
    __shared__ uint32_t lds[256];
    lds[threadIdx.x] = threadIdx.x;
    uint32_t idx = threadIdx.x == 0 ? 0 : 1;
    // (a) pointer chase, HBM misses
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) idx = chase[idx];
    __builtin_amdgcn_s_waitcnt(0);
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    // (b) the same chase again over the first 256 entries only (now in L2; L1 is bypassed? no - first touch per CU of a line misses L1)
    uint32_t j = idx & 1;
    uint64_t t2 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) j = scratch[j];
    __builtin_amdgcn_s_waitcnt(0);
    uint64_t t3 = __builtin_amdgcn_s_memtime();
    // (c) store + wait
    uint64_t t4 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        scratch[4096 + (size_t)i * stride_words + threadIdx.x] = j + i;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_sched_barrier(0);
    }
    uint64_t t5 = __builtin_amdgcn_s_memtime();
    // (d) LDS dependent reads
    uint32_t k = threadIdx.x;
    uint64_t t6 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) k = lds[k & 255];
    __builtin_amdgcn_s_waitcnt(0);
    uint64_t t7 = __builtin_amdgcn_s_memtime();
    // (e) store then load of another line, wait for both (the store-then-load pattern of a serial order)
    uint64_t t8 = __builtin_amdgcn_s_memtime();
    uint32_t q = 0;
    for (int i = 0; i < n; i++) {
        scratch[4096 + (size_t)(i + n) * stride_words + threadIdx.x] = q + i;
        q = chase[idx]; idx = q;
        __builtin_amdgcn_s_waitcnt(0);
    }
    uint64_t t9 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t3 - t2; out[2] = t5 - t4; out[3] = t7 - t6; out[4] = t9 - t8; out[5] = idx + j + k + q; }
}

int main(int argc, char** argv) {
    int n = 256, bg = argc > 1 ? atoi(argv[1]) : 0;
    size_t words = (size_t)1 << 28;  // 1 GiB chase buffer
    std::vector<uint32_t> h(words, 0);
    // chase: a random cycle over n+ entries spread 1 MiB apart
    std::vector<uint32_t> pos(2 * n + 2);
    std::mt19937 rng(1);
    for (size_t i = 0; i < pos.size(); i++) pos[i] = (uint32_t)((i * 262144 + (rng() % 4096) * 16) % words);
    pos[0] = 0; pos[1] = 1;
    for (size_t i = 0; i + 1 < pos.size(); i++) h[pos[i]] = pos[i + 1];
    h[1] = pos[2];
    uint32_t *d, *s, *sink; uint64_t* o;
    hipMalloc(&d, words * 4); hipMemcpy(d, h.data(), words * 4, hipMemcpyHostToDevice);
    size_t sw = (size_t)1 << 26; hipMalloc(&s, sw * 4); hipMemset(s, 0, sw * 4);
    std::vector<uint32_t> hs(4096); for (int i = 0; i < 4096; i++) hs[i] = (i * 97 + 64) % 4096; hipMemcpy(s, hs.data(), 4096 * 4, hipMemcpyHostToDevice);
    size_t bgw = (size_t)1 << 28; hipMalloc(&sink, bgw * 4); hipMemset(sink, 1, bgw * 4);
    hipMalloc(&o, 64);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(probe, dim3(1 + bg), dim3(64), 0, 0, d, s, o, n, (size_t)16384, sink, bgw);
        uint64_t r[6]; hipMemcpy(r, o, 48, hipMemcpyDeviceToHost);
        printf("bg=%d rep %d: HBM-miss load %.0f cyc | second chase (L2/MALL?) %.0f | store+wait %.0f | LDS read %.0f | store+load+wait %.0f   (s_memtime ticks per op)\n", bg, rep, r[0] / (double)n, r[1] / (double)n, r[2] / (double)n, r[3] / (double)n, r[4] / (double)n);
    }
    return 0;
}
