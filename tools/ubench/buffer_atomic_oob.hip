// How many atomics a second does ONE address take, against 64 addresses on 64 different 128-byte lines?  (Why the
// work-queue kernel claims work through 64 counters; it issues its claim with a plain `if (lane == 0) atomicAdd`.)
//   hipcc --offload-arch=gfx950 -O3 -o buffer_atomic_oob buffer_atomic_oob.hip && ./buffer_atomic_oob
//
// Opt-in second experiment (`./buffer_atomic_oob oob`, NOT run by default): is a raw-buffer atomic whose offset
// lies past NUM_RECORDS dropped on gfx950 (which would let lane 0 claim without a branch)?  Round 2 ran this with
// the masked lanes at byte offset 0x7FFFFFFF and the device faulted — but that offset is not dword-aligned, so the
// fault may have been the misaligned atomic and says nothing about range checking.  The offset is now the aligned
// 0x7FFFFFFC; that form has NOT been run (a fault on this pool can reset the host's GPUs), so the question is open
// and the product does not depend on it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void oob_kernel(unsigned *counters, unsigned *results) {
    const unsigned lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(counters, 0, 64 * 128, 0x00020000);
    const int r = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rsrc, lane == 0 ? (int)(blockIdx.x % 64) * 128 : 0x7FFFFFFC, 0, 0);
    results[blockIdx.x * 64 + lane] = (unsigned)r;
}

__global__ void rate_kernel(unsigned *counters, unsigned spread_mask, int reps, unsigned *sink) {
    unsigned acc = 0;
    unsigned *addr = counters + ((blockIdx.x & spread_mask) * 32);
    for (int i = 0; i < reps; i++)
        if (threadIdx.x == 0) acc += atomicAdd(addr, 1u);
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const bool do_oob = argc >= 2 && argv[1][0] == 'o', do_rate = !do_oob; /* default: the rate study only */
    unsigned *counters, *results, *sink;
    const int blocks = 256;
    CHECK(hipMalloc(&counters, 64 * 128 + 4096));
    CHECK(hipMalloc(&results, blocks * 64 * 4));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(counters, 0, 64 * 128 + 4096));
    printf("device ready\n");
    if (do_oob) {
    hipLaunchKernelGGL(oob_kernel, dim3(blocks), dim3(64), 0, 0, counters, results);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> c((64 * 128 + 4096) / 4), r(blocks * 64);
    CHECK(hipMemcpy(c.data(), counters, c.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(r.data(), results, r.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 0, stray = 0, nonzero_oob = 0;
    for (size_t i = 0; i < c.size(); i++) {
        if (i < 64 * 32 && i % 32 == 0) sum += c[i]; else stray += c[i];
    }
    for (int b = 0; b < blocks; b++) for (int l = 1; l < 64; l++) nonzero_oob += r[b * 64 + l] != 0;
    printf("bounds-masked buffer atomic: %llu claims counted (expected %d), %llu stray increments, %llu out-of-range lanes with a non-zero return\n",
           sum, blocks, stray, nonzero_oob);
    }
    if (!do_rate) return 0;

    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (unsigned mask : {0u, 63u}) {
        const int waves = 6144, reps = 64;
        CHECK(hipMemset(counters, 0, 64 * 128));
        hipLaunchKernelGGL(rate_kernel, dim3(waves), dim3(64), 0, 0, counters, mask, 4, sink);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(rate_kernel, dim3(waves), dim3(64), 0, 0, counters, mask, reps, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%d waves x %d dependent atomicAdd-with-return on %d address(es): %.3f ms = %.1f M atomics/s\n", waves, reps,
               mask + 1, ms, waves * (double)reps / ms / 1e3);
    }
    return 0;
}
