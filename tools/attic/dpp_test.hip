// Lane-permute primitives of uia_common.h on the GPU: prints what each of them delivers to every lane (gpurun: hipcc this file, run it).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../nextgen-uia_amd/csrc/uia_common.h"
__global__ void k(float* out) {
    const int l = threadIdx.x;
    float v = (float)(1 << (l & 15)) + (l >> 4) * 100000.f;        // distinguishable per lane
    out[l] = uia_dpp_quad_xor1((float)l);
    out[64 + l] = uia_dpp_quad_xor2((float)l);
    out[128 + l] = uia_dpp_half_mirror((float)l);
    out[192 + l] = uia_dpp_row_mirror((float)l);
    { float a = (float)l, b = (float)(l + 1000); uia_swap16(a, b); out[256 + l] = a; out[320 + l] = b; }
    { float a = (float)l, b = (float)(l + 1000); uia_swap32(a, b); out[384 + l] = a; out[448 + l] = b; }
    out[640 + l] = rows_sum((float)l);
    out[704 + l] = rows_max((float)l);
    out[512 + l] = wave_sum((float)l);
    out[576 + l] = wave_max((float)((l * 37) % 64));
    (void)v;
}
int main() {
    float* d; hipMalloc(&d, 768 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[768]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"quad_xor1", "quad_xor2", "half_mirror", "row_mirror", "swap16.a", "swap16.b", "swap32.a", "swap32.b", "wave_sum", "wave_max", "rows_sum", "rows_max"};
    for (int s = 0; s < 12; ++s) { printf("%-12s", names[s]); for (int l = 0; l < 64; ++l) printf(" %g", h[64 * s + l]); printf("\n"); }
    return 0;
}
