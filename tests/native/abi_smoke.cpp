// The C ABI of librange_hip.so used from a program that knows nothing of Python or torch: plain
// pointers and sizes, device memory from hipMalloc, a stream from hipStreamCreate.  Built and run by
// tests/test_gpu_tools.py::test_c_abi_from_a_plain_program on the GPU box:
//   hipcc -O1 -o abi_smoke tests/native/abi_smoke.cpp -Iinclude -Lrange_amd -lrange_hip -Wl,-rpath,$PWD/range_amd
// A tiny RANGE+ engine (L = 6, H = 64, N = 300 rows) is fed with deterministic pseudo-random weights
// and bank rows prepared as range/range.py:78-95 prepares them; the program checks the contract every
// caller relies on - status codes and messages, unit e-hat, retrieval columns inside the value range,
// the numpy-contract entry equal to the device-resident one, top-k ordered and in range - and prints
// 64 output values for the Python side to compare with the oracle.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "range_hip.h"

#define REQUIRE(cond)                                                                \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "abi_smoke FAILED %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, \
                         range_last_error());                                        \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double uni() {   // splitmix64 -> (-1, 1)
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) / 4503599627370496.0 - 1.0;
}

int main() {
    REQUIRE(range_abi_version() == RANGE_ABI_VERSION);
    const int L = 6, H = 64, E = 256, N = 300, B = 40;
    // ---- encoder weights (torch (out,in) row-major, float64): Siren-like ranges
    std::vector<double> w0((size_t)H * L * L), b0(H), w1((size_t)H * H), b1(H), w2((size_t)E * H), b2(E);
    for (auto& v : w0) v = uni() / (L * L);
    for (auto& v : b0) v = uni() / (L * L);
    const double c = std::sqrt(6.0 / H);
    for (auto& v : w1) v = uni() * c;
    for (auto& v : b1) v = uni() * c;
    for (auto& v : w2) v = uni() * c;
    for (auto& v : b2) v = uni() * c;
    // ---- bank: keys normalised in float32, values as they are, unit xyz from float32 locations
    std::vector<float> keys((size_t)N * 256), values((size_t)N * 1024), xyz((size_t)N * 3);
    float vmin = 1e30f, vmax = -1e30f;
    for (int r = 0; r < N; ++r) {
        double n2 = 0.0;
        for (int k = 0; k < 256; ++k) { keys[(size_t)r * 256 + k] = (float)uni(); n2 += (double)keys[(size_t)r * 256 + k] * keys[(size_t)r * 256 + k]; }
        const float inv = (float)(1.0 / std::sqrt(n2));
        for (int k = 0; k < 256; ++k) keys[(size_t)r * 256 + k] *= inv;
        for (int k = 0; k < 1024; ++k) {
            const float v = (float)(3.0 * uni());
            values[(size_t)r * 1024 + k] = v;
            vmin = std::fmin(vmin, v);
            vmax = std::fmax(vmax, v);
        }
        const float lon = (float)(180.0 * uni()) * 0.017453292f, lat = (float)(std::asin(uni())) ;
        xyz[(size_t)r * 3 + 0] = std::cos(lat) * std::cos(lon);
        xyz[(size_t)r * 3 + 1] = std::cos(lat) * std::sin(lon);
        xyz[(size_t)r * 3 + 2] = std::sin(lat);
    }
    std::vector<double> lonlat((size_t)B * 2);
    for (int q = 0; q < B; ++q) { lonlat[2 * q] = 180.0 * uni(); lonlat[2 * q + 1] = 60.0 * uni(); }

    // ---- errors come back as codes + messages, never as exceptions
    range_ctx* ctx = nullptr;
    REQUIRE(range_create(99, &ctx) == RANGE_ERR_INVALID && ctx == nullptr && range_last_error()[0] != 0);
    REQUIRE(range_create(0, &ctx) == RANGE_OK && ctx != nullptr);
    double* d_lonlat = nullptr; double* d_out = nullptr; double* d_e64 = nullptr; float* d_e32 = nullptr; float* d_xq = nullptr;
    float* d_tv = nullptr; int64_t* d_ti = nullptr;
    REQUIRE(hipMalloc((void**)&d_lonlat, sizeof(double) * 2 * B) == hipSuccess);
    REQUIRE(hipMalloc((void**)&d_out, sizeof(double) * 1280 * B) == hipSuccess);
    REQUIRE(hipMalloc((void**)&d_e64, sizeof(double) * 256 * B) == hipSuccess);
    REQUIRE(hipMalloc((void**)&d_e32, sizeof(float) * 256 * B) == hipSuccess);
    REQUIRE(hipMalloc((void**)&d_xq, sizeof(float) * 4 * B) == hipSuccess);
    REQUIRE(hipMalloc((void**)&d_tv, sizeof(float) * 8 * B) == hipSuccess);
    REQUIRE(hipMalloc((void**)&d_ti, sizeof(int64_t) * 8 * B) == hipSuccess);
    REQUIRE(hipMemcpy(d_lonlat, lonlat.data(), sizeof(double) * 2 * B, hipMemcpyHostToDevice) == hipSuccess);
    hipStream_t stream = nullptr;
    REQUIRE(hipStreamCreate(&stream) == hipSuccess);
    REQUIRE(range_forward(ctx, d_lonlat, B, RANGE_MODEL_RANGE_PLUS, 0.5f, d_out, stream) == RANGE_ERR_STATE);   // nothing set yet

    range_encoder_desc desc{L, H, 2, E, RANGE_SH_CLOSED_FORM};
    const double* W[3] = {w0.data(), w1.data(), w2.data()};
    const double* Bv[3] = {b0.data(), b1.data(), b2.data()};
    REQUIRE(range_set_encoder(ctx, &desc, W, Bv) == RANGE_OK);
    REQUIRE(range_set_bank(ctx, keys.data(), values.data(), xyz.data(), N, 0) == RANGE_OK && range_bank_rows(ctx) == N);

    // ---- the forward, device-resident, and through the numpy-contract entry
    REQUIRE(range_forward(ctx, d_lonlat, B, RANGE_MODEL_RANGE_PLUS, 0.5f, d_out, stream) == RANGE_OK);
    REQUIRE(hipStreamSynchronize(stream) == hipSuccess);
    REQUIRE(range_check_async_error(ctx) == RANGE_OK);
    std::vector<double> out((size_t)B * 1280), out_host((size_t)B * 1280);
    REQUIRE(hipMemcpy(out.data(), d_out, sizeof(double) * 1280 * B, hipMemcpyDeviceToHost) == hipSuccess);
    REQUIRE(range_forward_host(ctx, d_lonlat, B, RANGE_MODEL_RANGE_PLUS, 0.5f, out_host.data(), stream) == RANGE_OK);
    for (int q = 0; q < B; ++q) {
        double n2 = 0.0;
        for (int k = 1024; k < 1280; ++k) n2 += out[(size_t)q * 1280 + k] * out[(size_t)q * 1280 + k];
        REQUIRE(std::fabs(std::sqrt(n2) - 1.0) < 1e-12);
        for (int k = 0; k < 1024; ++k) {
            const double v = out[(size_t)q * 1280 + k];
            REQUIRE(std::isfinite(v) && v >= vmin && v <= vmax);                 // a convex combination of bank values
        }
        for (int k = 0; k < 1280; ++k) REQUIRE(out[(size_t)q * 1280 + k] == out_host[(size_t)q * 1280 + k]);
    }
    // ---- the pieces: encode, then the top-k side channel
    REQUIRE(range_encode(ctx, d_lonlat, B, d_e64, d_e32, d_xq, stream) == RANGE_OK);
    REQUIRE(range_topk_stream(ctx, d_e32, B, 8, d_tv, d_ti, stream) == RANGE_OK);
    REQUIRE(hipStreamSynchronize(stream) == hipSuccess);
    std::vector<float> tv((size_t)B * 8);
    std::vector<int64_t> ti((size_t)B * 8);
    REQUIRE(hipMemcpy(tv.data(), d_tv, sizeof(float) * 8 * B, hipMemcpyDeviceToHost) == hipSuccess);
    REQUIRE(hipMemcpy(ti.data(), d_ti, sizeof(int64_t) * 8 * B, hipMemcpyDeviceToHost) == hipSuccess);
    for (int q = 0; q < B; ++q)
        for (int k = 0; k < 8; ++k) {
            REQUIRE(ti[(size_t)q * 8 + k] >= 0 && ti[(size_t)q * 8 + k] < N && std::fabs(tv[(size_t)q * 8 + k]) <= 1.0001f);
            if (k) REQUIRE(tv[(size_t)q * 8 + k] <= tv[(size_t)q * 8 + k - 1]);
        }
    REQUIRE(range_forward(ctx, d_lonlat, B, 7, 0.5f, d_out, stream) == RANGE_ERR_INVALID);   // unknown model id
    std::printf("abi_smoke ok: B=%d N=%d out[0][0..3] = %.9g %.9g %.9g %.9g top1[0] = %lld (%.7f)\n", B, N, out[0], out[1], out[2],
                out[3], (long long)ti[0], (double)tv[0]);
    range_destroy(ctx);
    (void)hipFree(d_lonlat); (void)hipFree(d_out); (void)hipFree(d_e64); (void)hipFree(d_e32); (void)hipFree(d_xq);
    (void)hipFree(d_tv); (void)hipFree(d_ti);
    (void)hipStreamDestroy(stream);
    return 0;
}
