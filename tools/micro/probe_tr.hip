// probe_tr.hip -- what ds_read_b64_tr_b16 returns on gfx950: LDS holds element index as value; every lane supplies the address
// of the i-th 8-byte piece of a [4 rows][16 cols] bf16 block (row pitch PITCH bytes) of its 16-lane group and prints what it got.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/probe_tr.hip -o tools/micro/bin/probe_tr && tools/micro/bin/probe_tr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int PITCH = 288;   // bytes per LDS row (128 elements + 16 pad)
__global__ void probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * PITCH / 2];
    for (int i = threadIdx.x; i < 64 * PITCH / 2; i += 64) lds[i] = (unsigned short)((i / (PITCH / 2)) * 256 + (i % (PITCH / 2)));   // row*256 + col
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    // group g: rows g*8 .. g*8+3, cols 32 .. 47
    const unsigned addr = (unsigned)((g * 8 + (i >> 2)) * PITCH + (32 + (i & 3) * 4) * 2);
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)((__attribute__((address_space(3))) char*)lds + addr));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf("  (r%2d,c%3d)", h[l * 4 + j] >> 8, h[l * 4 + j] & 255);
        printf("\n");
    }
    return 0;
}
