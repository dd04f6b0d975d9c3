// Probe: where does an LDS-DMA (global_load_lds_dwordx4) land when its M0 base is >= 64 KB?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const unsigned *src, unsigned *out, int dst_bytes, int total_words) {
    extern __shared__ unsigned sm[];
    for (int i = threadIdx.x; i < total_words; i += 64) sm[i] = 0;
    __syncthreads();
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + threadIdx.x * 4),
                                     (void __attribute__((address_space(3))) *)((char *)sm + dst_bytes), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < total_words; i += 64) out[i] = sm[i];
}
int main() {
    const int total = 81920 / 4;
    unsigned *src, *out;
    hipMalloc(&src, 1024); hipMalloc(&out, total * 4);
    std::vector<unsigned> h(256); for (int i = 0; i < 256; i++) h[i] = 0xA0000000u + i;
    hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    for (int dst : {4096, 61440, 65536, 66560, 73728, 79872}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 81920, 0, src, out, dst, total);
        std::vector<unsigned> o(total);
        hipMemcpy(o.data(), out, total * 4, hipMemcpyDeviceToHost);
        int first = -1, cnt = 0, ok = 0;
        for (int i = 0; i < total; i++) if (o[i]) { if (first < 0) first = i; cnt++; }
        for (int i = 0; i < 256; i++) ok += o[dst / 4 + i] == h[i];
        printf("dst %6d: %d nonzero words, first at byte %d, %d/256 correct at dst\n", dst, cnt, first * 4, ok);
    }
    return 0;
}
