// Probe: lane permutation of v_permlane16_swap_b32 / v_permlane32_swap_b32 on gfx950 (semantics check for the TRA GRU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned* out) {
    const unsigned lane = threadIdx.x;
    u2 a = __builtin_amdgcn_permlane16_swap(lane, 100 + lane, false, false);
    u2 b = __builtin_amdgcn_permlane32_swap(lane, 100 + lane, false, false);
    out[lane] = a[0]; out[64 + lane] = a[1]; out[128 + lane] = b[0]; out[192 + lane] = b[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    probe<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[4] = {"p16 vdst", "p16 src0", "p32 vdst", "p32 src0"};
    for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; i += 4) printf(" %u", h[r * 64 + i]); printf("\n"); }
    return 0;
}
