// What does v_cvt_pk_u8_f32 do with every f32, next to Rust's `as u8` (truncate toward zero, saturate, NaN -> 0:
// v_cvt_u32_f32 + min 255, what the colour map uses)?  And what does it cost?
//   hipcc --offload-arch=gfx950 -O3 -o cvt_pk_u8 cvt_pk_u8.hip && ./cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned sat_u8(float v) {
    unsigned u;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(u) : "v"(v));
    return u < 255u ? u : 255u;
}
__device__ __forceinline__ unsigned pk_u8(float v) {
    unsigned u;
    asm("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(u) : "v"(v));
    return u;
}
__device__ __forceinline__ unsigned pk_u8_floor(float v) {
    float f;
    unsigned u;
    asm("v_floor_f32 %0, %1" : "=v"(f) : "v"(v));
    asm("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(u) : "v"(f));
    return u;
}

// out[0..1]: mismatches of the raw / floored form; out[2..3]: a first mismatching bit pattern of each
__global__ void scan(unsigned long long *out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long bad0 = 0, bad1 = 0, ex0 = ~0ull, ex1 = ~0ull;
    for (unsigned long long b = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += stride) {
        const float f = __builtin_bit_cast(float, (unsigned)b);
        const unsigned want = sat_u8(f);
        if (pk_u8(f) != want) { bad0++; if (b < ex0) ex0 = b; }
        if (pk_u8_floor(f) != want) { bad1++; if (b < ex1) ex1 = b; }
    }
    if (bad0) { atomicAdd(out + 0, bad0); atomicMin(out + 2, ex0); }
    if (bad1) { atomicAdd(out + 1, bad1); atomicMin(out + 3, ex1); }
}

__global__ void show(const float *in, unsigned *o, int n) {
    int i = threadIdx.x;
    if (i < n) { o[3 * i] = sat_u8(in[i]); o[3 * i + 1] = pk_u8(in[i]); o[3 * i + 2] = pk_u8_floor(in[i]); }
}

int main() {
    unsigned long long *d, h[4] = {0, 0, ~0ull, ~0ull};
    CHECK(hipMalloc(&d, sizeof h));
    CHECK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scan, dim3(256 * 32), dim3(256), 0, 0, d);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("every f32 against trunc+saturate+NaN->0:  v_cvt_pk_u8_f32 differs on %llu patterns (first 0x%08llx);  v_floor_f32 + v_cvt_pk_u8_f32 on %llu (first 0x%08llx)\n",
           h[0], h[2], h[1], h[3]);
    const float samples[] = {0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 254.5f, 254.9f, 255.0f, 255.5f, 256.0f, 1e9f, -0.4f, -0.6f, -1.0f, -1e9f,
                             __builtin_nanf(""), __builtin_inff(), -__builtin_inff(), 127.99999f};
    const int n = sizeof samples / sizeof samples[0];
    float *di; unsigned *dout, ho[3 * n];
    CHECK(hipMalloc(&di, sizeof samples)); CHECK(hipMalloc(&dout, sizeof ho));
    CHECK(hipMemcpy(di, samples, sizeof samples, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(show, dim3(1), dim3(64), 0, 0, di, dout, n);
    CHECK(hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) printf("  %-14g as-u8 %3u   cvt_pk %3u   floor+cvt_pk %3u\n", samples[i], ho[3 * i], ho[3 * i + 1], ho[3 * i + 2]);
    return 0;
}
