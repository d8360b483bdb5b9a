// Round 5: semantics of the packed-f16 / mixed-precision instructions the GELU epilogue relies on, checked on the device:
//   * an inline constant (0.5, -1.0) as a source of v_pk_fma_f16 with op_sel_hi 0 for that source feeds BOTH halves;
//   * the clamp modifier of v_pk_fma_f16 clamps both halves to [0, 1] (+-inf included);
//   * v_fma_mix_f32 with op_sel / op_sel_hi reads the low / high half of a packed f16 register as an fp32 source;
//   * v_cvt_pk_f16_f32 rounds to nearest even and overflows to inf.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
__global__ void probe(const float* in, unsigned* out) {
    float a = in[0], b = in[1], c = in[2], d = in[3];
    unsigned h, r0, r1, r2; float m0, m1;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
    asm volatile("v_pk_fma_f16 %0, %1, %1, -1.0 op_sel_hi:[1,1,0]" : "=v"(r0) : "v"(h));
    asm volatile("v_pk_fma_f16 %0, %1, %1, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(r1) : "v"(h));
    unsigned ch; asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ch) : "v"(c), "v"(d));
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r2) : "v"(h), "v"(ch));
    asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(m0) : "v"(c), "v"(h));
    asm volatile("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(m1) : "v"(c), "v"(h));
    out[0] = h; out[1] = r0; out[2] = r1; out[3] = r2; out[4] = __float_as_uint(m0); out[5] = __float_as_uint(m1);
}
static float h2f(unsigned short x) { _Float16 v; memcpy(&v, &x, 2); return (float)v; }
int main() {
    float* in; unsigned* out; hipMalloc(&in, 16); hipMalloc(&out, 32);
    const float cases[][4] = {{0.75f, -1.5f, 3.0f, 0.25f}, {1.00048828125f, 70000.0f, 2.0f, 2.0f}, {300.0f, -300.0f, 1.0f, 1.0f}, {0.3f, 2.0009765625f + 0.00048828125f, -2.0f, 1.0f}};
    for (auto& cs : cases) {
        hipMemcpy(in, cs, 16, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(in, out);
        unsigned r[6]; hipMemcpy(r, out, 24, hipMemcpyDeviceToHost);
        float m0, m1; memcpy(&m0, &r[4], 4); memcpy(&m1, &r[5], 4);
        printf("a=%g b=%g c=%g d=%g | h=(%g, %g)  h*h-1=(%g, %g)  clamp(h*h+0.5)=(%g, %g)  h*(c,d)=(%g, %g)  c*h.lo=%g c*h.hi=%g\n", cs[0], cs[1], cs[2], cs[3],
               h2f(r[0] & 0xFFFF), h2f(r[0] >> 16), h2f(r[1] & 0xFFFF), h2f(r[1] >> 16), h2f(r[2] & 0xFFFF), h2f(r[2] >> 16), h2f(r[3] & 0xFFFF), h2f(r[3] >> 16), m0, m1);
    }
    return 0;
}
