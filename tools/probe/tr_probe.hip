// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds lds[i] = i (16-bit elements); every lane supplies its own byte address; the four
// 16-bit values each lane receives are printed for several address patterns.  Build: hipcc --offload-arch=gfx950 -O2 tr_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(int pattern, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];      // the only LDS object: byte offset 0
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    uint32_t addr;                                                     // byte address
    if (pattern == 0) addr = lane * 8;                                 // consecutive 8-byte chunks
    else if (pattern == 1) addr = (lane & 15) * 128 + (lane >> 4) * 8; // lane&15 -> row of a [*][64] image, lane>>4 -> 4-element chunk
    else if (pattern == 2) addr = ((lane & 15) >> 2) * 128 + (lane & 3) * 8 + (lane >> 4) * 512;   // 4 rows x 4 chunks per 16 lanes
    else addr = (lane & 3) * 128 + ((lane & 15) >> 2) * 8 + (lane >> 4) * 512;                     // same block, lanes walk rows first
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[lane * 4 + 0] = (uint16_t)(r.x & 0xffff); out[lane * 4 + 1] = (uint16_t)(r.x >> 16);
    out[lane * 4 + 2] = (uint16_t)(r.y & 0xffff); out[lane * 4 + 3] = (uint16_t)(r.y >> 16);
    if (lane == 0) out[256] = (uint16_t)(uint32_t)(uintptr_t)(&lds[0]);
}
int main() {
    uint16_t* d; hipMalloc(&d, 1024);
    uint16_t h[512];
    for (int p = 0; p < 4; ++p) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, p, d);
        hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
        printf("pattern %d (lds base lo16 = %u)\n", p, h[256]);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5u %5u %5u %5u%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : " |");
    }
    return 0;
}
