#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(rep == 2 ? 40 : 32), dim3(256), 0, 0, d);
        unsigned h[64]; hipMemcpy(h, d, 64 * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) printf("%u ", h[i]);
        printf("\n");
    }
    return 0;
}
