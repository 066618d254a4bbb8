// hwid_probe.hip -- where do the workgroups and wavefronts of a (1024 x 256-thread, 40 KB LDS) launch land?  Prints, for the first blocks, the XCC / SE / CU of the
// workgroup and the SIMD of each of its four wavefronts, then the histogram "wavefront index -> SIMD" and which block indices share a CU.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/hwid_probe.hip -o /tmp/hwid_probe && /tmp/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 4) void k(unsigned* out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(8);      // 200 us: the whole grid is resident together
    if ((threadIdx.x & 63) == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
    if (lds[255 - threadIdx.x] == 0.0f) out[0] = 0;
}
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1024;      // grid size: 4 = the four segment workgroups of ONE chunk (tools/phase_latency.py 1)
    unsigned* d;
    hipMalloc(&d, B * 4 * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
    hipLaunchKernelGGL(k, dim3(B), dim3(256), 40 * 1024, 0, d);
    std::vector<unsigned> h(B * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    std::map<unsigned, std::vector<int>> cu_blocks;
    for (int b = 0; b < B; ++b) {
        unsigned key = 0;
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 15;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            hist[w][simd]++;
            key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            if (b < 12) printf("block %4d wave %d: xcc %u se %u sh %u cu %2u simd %u wave_slot %u\n", b, w, xcc, se, sh, cu, simd, hw & 15);
        }
        cu_blocks[key].push_back(b);
    }
    printf("wave index -> SIMD histogram (rows: wave 0..3, columns: SIMD 0..3)\n");
    for (int w = 0; w < 4; ++w) printf("  wave %d: %4d %4d %4d %4d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("distinct CUs: %zu\n", cu_blocks.size());
    int shown = 0;
    for (auto& kv : cu_blocks) {
        if (shown++ >= 12) break;
        printf("  cu key %05x:", kv.first);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
    }
    return 0;
}
