// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in which lane / slot.  LDS holds element index i at element i
// (u16); every lane passes its own byte address; prints, per lane, the four u16 it receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(const int* addr_bytes, unsigned* out) {
    __shared__ uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v;
    const unsigned a = (unsigned)(size_t)lds + (unsigned)addr_bytes[threadIdx.x];   // LDS byte address (shared aperture offset)
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a));
    out[threadIdx.x * 2] = v.x;
    out[threadIdx.x * 2 + 1] = v.y;
}
int main() {
    int h_addr[64];
    unsigned h_out[128];
    int *d_addr; unsigned* d_out;
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: canonical row-major 4 x 16 block per 16-lane group, row pitch 16 el: lane i -> row i/4, cols 4(i%4); groups 64 el apart
        // mode 1: same but row pitch 160 el (a padded [p][128 ch] image), groups 16 columns apart
        // mode 2: every lane the same address (uniform)
        for (int l = 0; l < 64; ++l) {
            const int g = l >> 4, i = l & 15;
            int el = mode == 0 ? g * 64 + (i / 4) * 16 + (i % 4) * 4 : mode == 1 ? (i / 4) * 160 + g * 16 + (i % 4) * 4 : 100;
            h_addr[l] = el * 2;
        }
        hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
        hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l)
            printf("lane %2d addr_el %4d -> %4u %4u %4u %4u\n", l, h_addr[l] / 2, h_out[2 * l] & 0xffff, h_out[2 * l] >> 16, h_out[2 * l + 1] & 0xffff, h_out[2 * l + 1] >> 16);
    }
    return 0;
}
