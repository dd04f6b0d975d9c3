// Probe (lab): LDS-DMA (global_load_lds_dwordx4) addressing on gfx950.
//   1. does an M0 base >= 64 KB work?                           (yes: all six destinations below land in place)
//   2. is the instruction's immediate offset added to BOTH the global source address and the LDS destination?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OFF>
__global__ void k(const unsigned *src, unsigned *out, int dst_bytes, int total_words) {
    extern __shared__ unsigned sm[];
    for (int i = threadIdx.x; i < total_words; i += 64) sm[i] = 0;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + 2048 + threadIdx.x * 4),
                                     (void __attribute__((address_space(3))) *)((char *)sm + dst_bytes), 16, OFF, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < total_words; i += 64) out[i] = sm[i];
}
template <int OFF>
static void run(const unsigned *src, unsigned *out, int dst, int total) {
    hipFuncSetAttribute((const void *)k<OFF>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    hipLaunchKernelGGL(k<OFF>, dim3(1), dim3(64), 81920, 0, src, out, dst, total);
    std::vector<unsigned> o(total);
    hipMemcpy(o.data(), out, total * 4, hipMemcpyDeviceToHost);
    int first = -1, cnt = 0;
    for (int i = 0; i < total; i++) if (o[i]) { if (first < 0) first = i; cnt++; }
    printf("M0 base %6d, imm offset %5d: %d nonzero words, first at LDS byte %d, holding source word %d (source base word 2048)\n", dst, OFF, cnt,
           first * 4, first >= 0 ? (int)(o[first] - 0xA0000000u) : -1);
}
int main() {
    const int total = 81920 / 4;
    unsigned *src, *out;
    hipMalloc(&src, 16384); hipMalloc(&out, total * 4);
    std::vector<unsigned> h(4096); for (int i = 0; i < 4096; i++) h[i] = 0xA0000000u + i;
    hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
    for (int dst : {4096, 61440, 65536, 66560, 73728, 79872}) run<0>(src, out, dst, total);
    run<1024>(src, out, 8192, total);
    run<4080>(src, out, 8192, total);
    run<-1024>(src, out, 8192, total);
    run<-4096>(src, out, 8192, total);
    run<-3072>(src, out, 70000 / 16 * 16, total);
    return 0;
}
