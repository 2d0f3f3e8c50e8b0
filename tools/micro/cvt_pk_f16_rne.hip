// Is v_cvt_pk_f16_f32 (new on gfx950) round-to-nearest-even under hipcc's default float mode?
// Compares it with the (_Float16) cast (v_cvt_f16_f32, RNE) on 2^24 random floats incl. ties and
// subnormal results.  topk_gemm.h relies on it for the bound of its fp16 operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
__global__ void k(const float* a, const float* b, uint32_t* o1, uint32_t* o2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a[i]), "v"(b[i]));
    o1[i] = r;
    const uint16_t lo = __builtin_bit_cast(uint16_t, (_Float16)a[i]), hi = __builtin_bit_cast(uint16_t, (_Float16)b[i]);
    o2[i] = (uint32_t)lo | ((uint32_t)hi << 16);
}
int main() {
    const int n = 1 << 24;
    std::vector<float> a(n), b(n);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> u(-2.f, 2.f);
    for (int i = 0; i < n; ++i) {
        a[i] = std::ldexp(u(g), (int)(g() % 40) - 28);                     // 2^-28 .. 2^12: normals, subnormals, ties rare
        uint32_t bits = (g() & 0x807FE000u) | ((100u + g() % 30u) << 23) | 0x1000u;   // exact ties of the 13 dropped bits
        b[i] = __builtin_bit_cast(float, bits);
    }
    float *da, *db; uint32_t *d1, *d2;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, d1, d2, n);
    std::vector<uint32_t> o1(n), o2(n);
    hipMemcpy(o1.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(o2.data(), d2, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) bad += o1[i] != o2[i];
    std::printf("v_cvt_pk_f16_f32 vs (_Float16) cast on %d pairs (ties and subnormals included): %ld differ\n", n, bad);
    return bad != 0;
}
