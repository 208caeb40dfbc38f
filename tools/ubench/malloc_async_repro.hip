// malloc_async_repro.hip — is "hipMallocAsync -> kernel A writes -> kernel B reads -> hipFreeAsync", all on ONE
// stream, reliable on this runtime?  Round 1 built smooth == false palettes into stream-ordered memory exactly
// like that and a 1500-configuration soak intermittently read them back as zeros (commit d65c2b5 replaced the
// allocator with a library-owned slot ring).  The sequence is legal as written, so this is the minimal form of
// it, with the things the library did around it (a synchronous hipFree/hipMalloc of a large scratch buffer
// between calls, a D2H copy + stream synchronise after each) and an optional second thread doing the same on
// its own stream.  Every read is checked on the device.
//   hipcc --offload-arch=gfx950 -O2 -o malloc_async_repro malloc_async_repro.hip -lpthread
//   ./malloc_async_repro [iterations] [threads] [regrow_scratch 0/1]
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                         \
            exit(2);                                                                      \
        }                                                                                 \
    } while (0)

__global__ void fill(uint32_t *pal, uint32_t n, uint32_t seed) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) pal[i] = (seed * 2654435761u + i * 40503u) | 1u; /* never zero */
}

/* every workgroup stages the palette in LDS (as the render kernels did) and checks it */
__global__ void check(const uint32_t *pal, uint32_t n, uint32_t seed, unsigned long long *bad, unsigned long long *zeros,
                      uint8_t *out) {
    __shared__ uint32_t s[1280];
    for (uint32_t k = threadIdx.x; k < n; k += 64) s[k] = pal[k];
    __syncthreads();
    unsigned long long b = 0, z = 0;
    for (uint32_t k = threadIdx.x; k < n; k += 64) {
        const uint32_t want = (seed * 2654435761u + k * 40503u) | 1u;
        b += s[k] != want;
        z += s[k] == 0u;
    }
    if (b) atomicAdd(bad, b);
    if (z) atomicAdd(zeros, z);
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = (uint8_t)s[threadIdx.x % n];
}

std::atomic<unsigned long long> g_bad{0}, g_zero{0}, g_calls_bad{0};

void worker(int tid, int iterations, bool regrow) {
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *d_cnt;
    CK(hipMalloc(&d_cnt, 16));
    uint8_t *scratch = nullptr;
    size_t scratch_cap = 0;
    std::vector<uint8_t> host(1 << 24);
    uint32_t rng = 12345u + 977u * (uint32_t)tid;
    for (int it = 0; it < iterations; it++) {
        rng = rng * 1664525u + 1013904223u;
        const uint32_t n = 2 + (rng >> 8) % 1278;            /* palette entries: iterations + 1 < 1280 */
        const uint32_t blocks = 1 + (rng >> 20) % 4096;      /* render-kernel workgroups */
        const size_t need = (size_t)blocks * 64;
        if (regrow && need > scratch_cap) { /* reserve_locked(): synchronous free + malloc of a bigger buffer */
            if (scratch) CK(hipFree(scratch));
            CK(hipMalloc(&scratch, need));
            scratch_cap = need;
        } else if (!scratch) {
            CK(hipMalloc(&scratch, (size_t)4096 * 64));
            scratch_cap = (size_t)4096 * 64;
        }
        CK(hipMemsetAsync(d_cnt, 0, 16, s));
        uint32_t *pal = nullptr;
        CK(hipMallocAsync(reinterpret_cast<void **>(&pal), sizeof(uint32_t) * n, s));
        hipLaunchKernelGGL(fill, dim3((n + 255) / 256), dim3(256), 0, s, pal, n, (uint32_t)it);
        hipLaunchKernelGGL(check, dim3(blocks), dim3(64), 0, s, pal, n, (uint32_t)it, d_cnt, d_cnt + 1, scratch);
        CK(hipFreeAsync(pal, s));
        unsigned long long cnt[2];
        CK(hipMemcpyAsync(host.data(), scratch, need, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(cnt, d_cnt, 16, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        if (cnt[0]) {
            g_bad += cnt[0];
            g_zero += cnt[1];
            if (g_calls_bad++ < 10)
                printf("thread %d call %d: %llu wrong words (%llu of them zero) of %u x %u reads\n", tid, it, cnt[0], cnt[1], n, blocks);
        }
    }
    CK(hipFree(scratch));
    CK(hipFree(d_cnt));
    CK(hipStreamDestroy(s));
}

int main(int argc, char **argv) {
    const int iterations = argc > 1 ? atoi(argv[1]) : 20000;
    const int threads = argc > 2 ? atoi(argv[2]) : 1;
    const bool regrow = argc > 3 ? atoi(argv[3]) != 0 : true;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(worker, t, iterations, regrow);
    for (auto &t : th) t.join();
    printf("malloc_async_repro: %d calls x %d threads, regrow=%d: %llu calls with wrong reads, %llu wrong words, %llu zeros\n",
           iterations, threads, (int)regrow, g_calls_bad.load(), g_bad.load(), g_zero.load());
    return g_calls_bad.load() ? 1 : 0;
}
