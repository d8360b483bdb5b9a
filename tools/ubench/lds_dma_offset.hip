// Where does the immediate offset of global_load_lds_dwordx4 go?  (gfx950; round 4)
// Question: with `global_load_lds_dwordx4 voff, s[base:base+1] offset:N`, is N added to the GLOBAL address only, to the LDS
// destination only, or to both?  The q4 GEMM issues 12 LDS-DMA pieces per K slab and writes m0 in front of each one; if the
// immediate moves the LDS side too, one m0 write serves four pieces (offsets 0 / 1024 / 2048 / 3072).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_dma_offset.hip -o lds_dma_offset.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int OFF>
__global__ void probe(const unsigned* src, unsigned* out, int m0v) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xDEAD0000u + i;
    __syncthreads();
    unsigned voff = threadIdx.x * 16 + 8192;                 // byte offset of this lane's 16 bytes (src + 8 KiB: room for negative N)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(src), "s"(m0v), "n"(OFF) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}

template <int OFF>
static void run(const unsigned* src, unsigned* out, int m0v) {
    hipLaunchKernelGGL(probe<OFF>, dim3(1), dim3(64), 16384, 0, src, out, m0v);
    std::vector<unsigned> h(4096);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, 16384, hipMemcpyDeviceToHost);
    int first = -1, n = 0;
    for (int i = 0; i < 4096; ++i)
        if (h[i] != 0xDEAD0000u + i) { if (first < 0) first = i; ++n; }
    if (first < 0) { printf("offset %6d m0 %5d: nothing landed\n", OFF, m0v); return; }
    // src[i] = i, so the value names the global dword that was read
    printf("offset %6d m0 %5d: %d dwords landed, first at LDS byte %d (m0 %+d), holding global byte %u (lane base 8192 %+d)\n", OFF, m0v, n,
           first * 4, first * 4 - m0v, h[first] * 4, (int)(h[first] * 4) - 8192);
}

int main() {
    unsigned *src, *out;
    hipMalloc(&src, 1 << 16);
    hipMalloc(&out, 16384);
    std::vector<unsigned> h(1 << 14);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i;
    hipMemcpy(src, h.data(), 1 << 16, hipMemcpyHostToDevice);
    run<0>(src, out, 0);
    run<1024>(src, out, 0);
    run<3072>(src, out, 1024);
    run<-1024>(src, out, 4096);
    run<-4096>(src, out, 8192);
    run<4080>(src, out, 0);
    return 0;
}
