// Times mlpk_dwconv_nhwc on the ConvMixer-1536/20 layer shape (256 x 32 x 32 x 1536 bf16, k = 9) and prints a checksum.
// Built per kernel variant against jittor-mlp_amd/csrc/mlpk_dwconv.hip (tools/gpu_dwconv_variants.sh): an experiment harness.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
extern "C" int mlpk_dwconv_nhwc(int dtype, const void* x, void* out, int B, int H, int W, int C, int k, const float* w,
                                const float* bias, const float* bn_scale, const float* bn_shift, void* stream);
int mlpk_dwconv_direct(int, const void*, void*, int, int, int, int, int, const float*, const float*, const float*, const float*, void*) { return -1; }
static uint16_t bf16(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, H = 32, W = 32, C = 1536, K = 9;
    const size_t n = (size_t)B * H * W * C;
    std::vector<uint16_t> hx(n);
    uint32_t s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = bf16(((s >> 8) & 0xffff) / 65536.f - 0.5f); }
    std::vector<float> hw((size_t)K * K * C), hb(C), hs(C), hh(C);
    for (size_t i = 0; i < hw.size(); ++i) { s = s * 1664525u + 1013904223u; hw[i] = (((s >> 8) & 0xffff) / 65536.f - 0.5f) * 0.2f; }
    for (int c = 0; c < C; ++c) { hb[c] = 0.01f * (c % 7); hs[c] = 1.f + 0.001f * (c % 5); hh[c] = 0.02f * (c % 3); }
    void *x, *o; float *w, *b, *sc, *sh;
    hipMalloc(&x, n * 2); hipMalloc(&o, n * 2); hipMalloc(&w, hw.size() * 4); hipMalloc(&b, C * 4); hipMalloc(&sc, C * 4); hipMalloc(&sh, C * 4);
    hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(sc, hs.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(sh, hh.data(), C * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) { int rc = mlpk_dwconv_nhwc(2, x, o, B, H, W, C, K, w, b, sc, sh, nullptr); if (rc) { printf("rc=%d\n", rc); return 1; } }
    hipDeviceSynchronize();
    const int it = 20;
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < it; ++i) mlpk_dwconv_nhwc(2, x, o, B, H, W, C, K, w, b, sc, sh, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint16_t> ho(n);
    hipMemcpy(ho.data(), o, n * 2, hipMemcpyDeviceToHost);
    uint64_t ck = 0; for (size_t i = 0; i < n; ++i) ck = ck * 1099511628211ull + ho[i];
    printf("%s: %.1f us per launch, checksum %016llx\n", argv[0], ms * 1000.f / it, (unsigned long long)ck);
    return 0;
}
