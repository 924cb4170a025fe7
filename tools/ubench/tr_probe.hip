// Probe of ds_read_b64_tr_b16 on gfx950: which LDS elements does lane l receive?
// LDS holds u16 values = their own element index; every lane passes the address of 4 contiguous elements
// (row = lane>>2 within its 16-lane group, column quad = lane&3, row stride RS elements).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(unsigned short *out, int rs) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, grp = l >> 4, i = l & 15;
    // group g reads the 4x16 block whose first row is 4*g, rows rs apart
    const unsigned addr = (unsigned)(size_t)(&lds[(4 * grp + (i >> 2)) * rs + 4 * (i & 3)]);   // low 32 bits = LDS offset
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
    unsigned short *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int rs : {16, 40}) {
        probe<<<1, 64>>>(d, rs);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row stride %d elements: lane -> 4 received element indices (row*rs+col => printed as row:col)\n", rs);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %d:%d", h[l * 4 + j] / rs, h[l * 4 + j] % rs);
            printf("\n");
        }
    }
    return 0;
}
