// uia_kernels.h — internal launcher declarations shared by the .hip files and capi.cpp.
// The public contract is include/uia_hip.h; nothing here is exported.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

enum { UIA_F32 = 0, UIA_BF16 = 1 };
enum { UIA_ACT_NONE = 0, UIA_ACT_GELU = 1, UIA_ACT_QUICKGELU = 2, UIA_ACT_RELU = 3 };
enum { UIA_MASK_NONE = 0, UIA_MASK_CAUSAL = 1, UIA_MASK_KEYPAD = 2 };
enum { UIA_MONA_BASELINE = 0, UIA_MONA_NOISE_AWARE = 1, UIA_MONA_FREQ_ENHANCED = 2, UIA_MONA_HYBRID = 3 };

struct UiaGemmParams {
    const void* A; long lda;        // [M,K] of T
    const void* W; long ldw;        // [N,K] of T
    int M, N, K;
    float alpha;                    // scale on the accumulator (before bias)
    const float* bias;              // [N] fp32 or null
    int act;                        // UIA_ACT_* applied after bias
    int dact;                       // UIA_ACT_*: multiply by act'(aux_in) (GELU backward fused into dgrad)
    const void* aux_in; long ldaux_in;    // T [M,N]: pre-activation read by dact
    void* aux_out; long ldaux_out;        // T [M,N]: pre-activation stash (value before act)
    const float* resid; long ldr;   // fp32 residual added last, or null
    int resid_mod, resid_row_off;   // if resid_mod>0: residual row = m % resid_mod + resid_row_off (pos-embed)
    const void* residT; long ldrT;  // T residual (accumulate into a T tensor), or null
    int out_group;                  // if >0: output row = m + m/out_group + 1 (patch rows → token rows)
    void* outT; long ldo;           // T output or null
    float* out32; long ldo32;       // fp32 output or null
};

int uia_gemm_launch(hipStream_t stream, int dtype, const UiaGemmParams& p, int cfg);
