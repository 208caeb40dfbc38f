// host_path.hip — what does getting an 805 MB image from HBM into a caller's host buffer cost?
// Measures, on the GPU box, the pieces the host-buffer entry points are built from:
//   1. hipHostRegister / hipHostUnregister of a resident (already touched) and of a fresh buffer
//   2. first-touch of a fresh buffer by T threads (MADV_POPULATE_WRITE, and a plain touch loop)
//   3. D2H into the registered caller buffer (one copy, and 64-MiB bands)
//   4. D2H into a library-owned pinned staging ring + T memcpy threads into the (fresh / resident) buffer
// Build: hipcc --offload-arch=gfx950 -O2 -o host_path host_path.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

static uint8_t *fresh(size_t n) {
    void *p = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (p == MAP_FAILED) {
        perror("mmap");
        exit(1);
    }
    return static_cast<uint8_t *>(p);
}

template <typename F>
static void par(int T, size_t n, F f) {
    std::vector<std::thread> th;
    const size_t chunk = ((n + T - 1) / T + 4095) & ~size_t(4095);
    for (int t = 0; t < T; t++) {
        size_t a = (size_t)t * chunk, b = a + chunk < n ? a + chunk : n;
        if (a >= n) break;
        th.emplace_back([=] { f(a, b); });
    }
    for (auto &t : th) t.join();
}

int main(int argc, char **argv) {
    const size_t N = argc > 1 ? strtoull(argv[1], nullptr, 0) : (size_t)805306368;
    CK(hipSetDevice(0));
    uint8_t *dev;
    CK(hipMalloc(&dev, N));
    CK(hipMemset(dev, 0x5A, N));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipDeviceSynchronize());
    printf("bytes %zu, hw threads %u\n", N, std::thread::hardware_concurrency());

    // 1. register / unregister
    for (int rep = 0; rep < 2; rep++) {
        uint8_t *h = fresh(N);
        double t0 = now_ms();
        CK(hipHostRegister(h, N, hipHostRegisterDefault));
        double t1 = now_ms();
        CK(hipHostUnregister(h));
        double t2 = now_ms();
        CK(hipHostRegister(h, N, hipHostRegisterDefault));
        double t3 = now_ms();
        CK(hipMemcpyAsync(h, dev, N, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t4 = now_ms();
        CK(hipHostUnregister(h));
        double t5 = now_ms();
        printf("register fresh %.2f ms, unregister %.2f, register resident %.2f, D2H one copy %.2f (%.1f GB/s), unregister %.2f\n",
               t1 - t0, t2 - t1, t3 - t2, t4 - t3, N / (t4 - t3) / 1e6, t5 - t4);
        munmap(h, N);
    }
    // 1b. register per 64 MiB band
    {
        uint8_t *h = fresh(N);
        memset(h, 1, N);
        const size_t band = (size_t)64 << 20;
        double t0 = now_ms();
        for (size_t a = 0; a < N; a += band) CK(hipHostRegister(h + a, a + band < N ? band : N - a, hipHostRegisterDefault));
        double t1 = now_ms();
        for (size_t a = 0; a < N; a += band) CK(hipHostUnregister(h + a));
        double t2 = now_ms();
        printf("register resident in 64 MiB bands %.2f ms, unregister %.2f\n", t1 - t0, t2 - t1);
        munmap(h, N);
    }
    // 1c. (round 3) hipHostRegister of RESIDENT memory: how does it scale with the chunk size and with the number
    //     of threads registering different chunks at the same time?  (The multi-device host sink pins the caller's
    //     buffer while N devices wait for it: one thread's 72 GB/s is 1.3 PCIe links' worth.)  `regonly` = only this.
    //     A NEW buffer per measurement: registering memory that was registered before costs next to nothing (the
    //     first run of this study showed 14.6 ms for the first pass over a buffer and 0.05-0.4 ms for every later
    //     one, whatever the chunk size or thread count) — the driver keeps what it built for those pages.
    for (size_t mib : {16, 64, 256}) {
        const size_t band = mib << 20;
        const size_t nb = (N + band - 1) / band;
        for (int T : {1, 2, 4, 8}) {
            for (unsigned flags : {(unsigned)hipHostRegisterDefault, (unsigned)hipHostRegisterPortable}) {
                uint8_t *h = fresh(N);
                madvise(h, N, MADV_HUGEPAGE);
                par(8, N, [&](size_t a, size_t b) { memset(h + a, 1, b - a); });
                std::atomic<size_t> next{0};
                std::atomic<int> failed{0};
                double t0 = now_ms();
                {
                    std::vector<std::thread> th;
                    for (int t = 0; t < T; t++)
                        th.emplace_back([&] {
                            for (;;) {
                                const size_t b = next.fetch_add(1);
                                if (b >= nb) break;
                                const size_t a = b * band;
                                if (hipHostRegister(h + a, a + band < N ? band : N - a, flags) != hipSuccess) failed++;
                            }
                        });
                    for (auto &t : th) t.join();
                }
                double t1 = now_ms();
                // again, the same (still resident) pages after an unregister: what a caller that reuses its buffer pays
                for (size_t a = 0; a < N; a += band) (void)hipHostUnregister(h + a);
                double t2 = now_ms();
                for (size_t a = 0; a < N; a += band) (void)hipHostRegister(h + a, a + band < N ? band : N - a, flags);
                double t3 = now_ms();
                for (size_t a = 0; a < N; a += band) (void)hipHostUnregister(h + a);
                printf("register resident, %3zu MiB chunks, %d thread(s), %s: first %.2f ms (%.1f GB/s)  unregister %.2f ms  again (1 thread) %.2f ms %s\n",
                       mib, T, flags == hipHostRegisterPortable ? "portable" : "default ", t1 - t0, N / (t1 - t0) / 1e6, t2 - t1, t3 - t2,
                       failed.load() ? "FAILED" : "");
                munmap(h, N);
            }
        }
    }
    // 1d. the same for memory the CALLER touched without a huge-page hint (4-KiB pages unless THP is "always"):
    //     what a buffer that already exists costs to pin, by thread count (64 MiB chunks)
    for (int T : {1, 2, 4, 8}) {
        uint8_t *h = fresh(N);
        par(8, N, [&](size_t a, size_t b) { memset(h + a, 1, b - a); });
        const size_t band = (size_t)64 << 20, nb = (N + band - 1) / band;
        std::atomic<size_t> next{0};
        double t0 = now_ms();
        {
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&] {
                    for (;;) {
                        const size_t b = next.fetch_add(1);
                        if (b >= nb) break;
                        const size_t a = b * band;
                        (void)hipHostRegister(h + a, a + band < N ? band : N - a, hipHostRegisterPortable);
                    }
                });
            for (auto &t : th) t.join();
        }
        double t1 = now_ms();
        for (size_t a = 0; a < N; a += band) (void)hipHostUnregister(h + a);
        printf("register resident 4-KiB-page memory, 64 MiB chunks, %d thread(s): %.2f ms (%.1f GB/s)\n", T, t1 - t0, N / (t1 - t0) / 1e6);
        munmap(h, N);
    }
    if (argc > 2 && !strcmp(argv[2], "regonly")) return 0;
    // 2. first touch by T threads
    for (int T : {1, 2, 4, 8, 16}) {
        uint8_t *h = fresh(N);
        double t0 = now_ms();
        std::atomic<int> bad{0};
        par(T, N, [&](size_t a, size_t b) {
            if (madvise(h + a, b - a, MADV_POPULATE_WRITE) != 0) bad++;
        });
        double t1 = now_ms();
        munmap(h, N);
        uint8_t *g = fresh(N);
        double t2 = now_ms();
        par(T, N, [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k += 4096) g[k] = 0;
        });
        double t3 = now_ms();
        munmap(g, N);
        uint8_t *q = fresh(N);
        madvise(q, N, MADV_HUGEPAGE);
        double t4 = now_ms();
        par(T, N, [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k += 4096) q[k] = 0;
        });
        double t5 = now_ms();
        munmap(q, N);
        printf("T=%2d  MADV_POPULATE_WRITE %.2f ms (%d failed)   touch loop %.2f ms   touch after MADV_HUGEPAGE %.2f ms\n", T,
               t1 - t0, bad.load(), t3 - t2, t5 - t4);
    }
    // 3. banded D2H into a registered resident buffer
    {
        uint8_t *h = fresh(N);
        memset(h, 1, N);
        CK(hipHostRegister(h, N, hipHostRegisterDefault));
        const size_t band = (size_t)64 << 20;
        double t0 = now_ms();
        for (size_t a = 0; a < N; a += band)
            CK(hipMemcpyAsync(h + a, dev + a, a + band < N ? band : N - a, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        double t1 = now_ms();
        printf("D2H in 64 MiB bands into registered buffer %.2f ms (%.1f GB/s)\n", t1 - t0, N / (t1 - t0) / 1e6);
        CK(hipHostUnregister(h));
        munmap(h, N);
    }
    // 4. staging ring + memcpy threads
    {
        const size_t band = (size_t)32 << 20;
        const int R = 4;
        uint8_t *stage[R];
        hipEvent_t ev[R];
        for (int r = 0; r < R; r++) {
            CK(hipHostMalloc(reinterpret_cast<void **>(&stage[r]), band, hipHostMallocDefault));
            CK(hipEventCreateWithFlags(&ev[r], hipEventDisableTiming));
        }
        for (int T : {2, 4, 8, 16}) {
            for (int freshbuf = 0; freshbuf < 2; freshbuf++) {
                uint8_t *h = fresh(N);
                if (!freshbuf) memset(h, 1, N);
                double t0 = now_ms();
                const size_t nb = (N + band - 1) / band;
                // T copier threads: band b is handled by thread pool slices once its D2H event completed
                std::vector<std::thread> th;
                std::atomic<size_t> issued{0};
                std::atomic<size_t> drained[64];
                for (auto &d : drained) d = 0;
                // simple scheme: the main thread issues D2H for band b into slot b % R once the slot is drained,
                // copier threads each take a 1/T slice of every band.
                std::atomic<size_t> ready{0};  // number of bands whose D2H completed
                std::vector<std::atomic<int>> slices(nb);
                for (auto &x : slices) x = 0;
                for (int t = 0; t < T; t++) {
                    th.emplace_back([&, t] {
                        for (size_t b = 0; b < nb; b++) {
                            while (ready.load(std::memory_order_acquire) <= b) std::this_thread::yield();
                            const size_t len = b * band + band < N ? band : N - b * band;
                            const size_t sl = ((len + T - 1) / T + 63) & ~size_t(63);
                            const size_t a = (size_t)t * sl, e = a + sl < len ? a + sl : len;
                            if (a < len) memcpy(h + b * band + a, stage[b % R] + a, e - a);
                            slices[b].fetch_add(1, std::memory_order_release);
                        }
                    });
                }
                for (size_t b = 0; b < nb; b++) {
                    if (b >= (size_t)R)
                        while (slices[b - R].load(std::memory_order_acquire) < T) std::this_thread::yield();
                    const size_t len = b * band + band < N ? band : N - b * band;
                    CK(hipMemcpyAsync(stage[b % R], dev + b * band, len, hipMemcpyDeviceToHost, s));
                    CK(hipEventRecord(ev[b % R], s));
                    // completion is observed in order: wait for the previous band here so `ready` advances
                    if (b >= 1) {
                        CK(hipEventSynchronize(ev[(b - 1) % R]));
                        ready.store(b, std::memory_order_release);
                    }
                }
                CK(hipEventSynchronize(ev[(nb - 1) % R]));
                ready.store(nb, std::memory_order_release);
                for (auto &t : th) t.join();
                double t1 = now_ms();
                bool ok = h[0] == 0x5A && h[N - 1] == 0x5A && h[N / 2] == 0x5A;
                printf("staging ring (4 x 32 MiB) + %2d memcpy threads into %s buffer: %.2f ms (%.1f GB/s) %s\n", T,
                       freshbuf ? "FRESH" : "resident", t1 - t0, N / (t1 - t0) / 1e6, ok ? "ok" : "BAD");
                munmap(h, N);
            }
        }
    }
    return 0;
}
