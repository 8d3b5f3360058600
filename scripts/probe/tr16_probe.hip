// Probe of ds_read_b64_tr_b16 (gfx950): which 16-bit element of the LDS lands in which (lane, element) of the result.
// LDS is filled with its own element index; every lane supplies the byte address addr[lane]; the four returned halves are printed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) bf4 lds_bf4;
    bf4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf4*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
    uint16_t r[4];
    __builtin_memcpy(r, &v, 8);
    for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = r[e];
}
int main(int argc, char** argv) {
    const int stride = argc > 1 ? atoi(argv[1]) : 96;      // row stride in bytes
    int h_addr[64]; uint16_t h_out[256];
    // lane l = 16 g + i: row 4 g + (i >> 2), 4 elements starting at column 4 (i & 3)
    for (int l = 0; l < 64; ++l) { const int g = l >> 4, i = l & 15; h_addr[l] = (4 * g + (i >> 2)) * stride + 8 * (i & 3); }
    int* d_addr; uint16_t* d_out;
    hipMalloc(&d_addr, sizeof h_addr); hipMalloc(&d_out, sizeof h_out);
    hipMemcpy(d_addr, h_addr, sizeof h_addr, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost);
    printf("stride %d bytes = %d elements\n", stride, stride / 2);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            const int idx = h_out[l * 4 + e], row = idx / (stride / 2), col = idx % (stride / 2);
            printf("  (r%2d,c%2d)", row, col);
            // expectation: lane (g, i) element e = row 4 g + e, column i
            const int g = l >> 4, i = l & 15;
            if (row != 4 * g + e || col != i) ++bad;
        }
        printf("\n");
    }
    printf("expectation [row 4g+e][col i]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return 0;
}
