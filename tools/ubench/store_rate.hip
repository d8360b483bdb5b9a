// Tuning aid: how fast can one CU ISSUE global stores?  512 threads per workgroup, every thread issues NST 16-byte stores;
// cycles from first issue to last issue (no wait for completion) and to completion (s_waitcnt vmcnt(0)).
// Patterns: 0 = each wave writes 1 KiB contiguous per instruction; 1 = two 512-byte row pieces (row stride LD bytes), the
// p8 epilogue's pattern; 2 = pattern 1 with non-temporal stores; 3 = pattern 1 with 8-byte stores (twice as many).
// Build: hipcc --offload-arch=gfx950 -O3 store_rate.hip -o store_rate.bin ; run: ./store_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define NST 8
template <int PAT>
__global__ void __launch_bounds__(512) k(char* out, unsigned long long* cyc, long long wg_stride, int ld, int rounds) {
    const int tid = threadIdx.x;
    char* base = out + (long long)blockIdx.x * wg_stride;
    unsigned long long t_issue = 0, t_done = 0;
    for (int r = 0; r < rounds; ++r) {
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
        char* rb = base + (long long)r * (PAT == 0 ? 512 * 16 * NST : 0);
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const u32x4 v = {(unsigned)tid, (unsigned)q, (unsigned)r, 1u};
            if (PAT == 0) {
                *reinterpret_cast<u32x4*>(rb + ((long long)q * 512 + tid) * 16) = v;
            } else {
                const int row = q * 16 + (tid >> 5) + r * 128, c16 = tid & 31;
                char* p = rb + (long long)row * ld + c16 * 16;
                if (PAT == 1) *reinterpret_cast<u32x4*>(p) = v;
                else if (PAT == 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
                else { *reinterpret_cast<u32x2*>(p) = u32x2{v.x, v.y}; *reinterpret_cast<u32x2*>(p + 8) = u32x2{v.z, v.w}; }
            }
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        t_issue += t1 - t0; t_done += t2 - t0;
    }
    if ((tid & 63) == 0) { cyc[(blockIdx.x * 8 + (tid >> 6)) * 2] = t_issue; cyc[(blockIdx.x * 8 + (tid >> 6)) * 2 + 1] = t_done; }
}
template <int PAT> void run(const char* name, int grid, char* buf, unsigned long long* cyc) {
    const int rounds = 16, ld = 6144;
    const long long wg_stride = PAT == 0 ? (long long)rounds * 512 * 16 * NST : (long long)rounds * 128 * ld;   // disjoint regions
    hipMemset(cyc, 0, 256 * 8 * 2 * 8);
    k<PAT><<<grid, 512>>>(buf, cyc, wg_stride, ld, rounds);
    hipDeviceSynchronize();
    unsigned long long h[256 * 16];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double si = 0, sd = 0;
    for (int i = 0; i < grid * 8; ++i) { si += h[2 * i]; sd += h[2 * i + 1]; }
    const double bytes = 512.0 * 16 * NST;
    printf("%-44s grid %3d: issue %7.0f cycles (%.1f B/clk/CU)   complete %7.0f cycles (%.1f B/clk/CU)\n", name, grid, si / (grid * 8) / rounds,
           bytes / (si / (grid * 8) / rounds), sd / (grid * 8) / rounds, bytes / (sd / (grid * 8) / rounds));
}
int main() {
    char* buf; unsigned long long* cyc;
    const size_t sz = (size_t)256 * 16 * 128 * 6144;
    hipMalloc(&buf, sz); hipMalloc(&cyc, 256 * 8 * 2 * 8);
    for (int grid : {1, 32, 256}) {
        run<0>("1 KiB contiguous per wave-instruction", grid, buf, cyc);
        run<1>("2 x 512-byte row pieces (p8 epilogue)", grid, buf, cyc);
        run<2>("  same, non-temporal", grid, buf, cyc);
        run<3>("  same, 8-byte stores", grid, buf, cyc);
    }
    return 0;
}
