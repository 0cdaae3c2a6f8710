// jdsp_internal.h -- shared declarations of libjdsp (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/jdsp.h"

struct jdsp_ctx {
    int device = 0;
    int n_cu = 0;
    size_t hbm_bytes = 0;
    char name[64] = {0};
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // the stream work is enqueued on
    std::string error;
    int opt_stft_fpw = 0;              // 0 = auto
    // device tables, created on first use
    float2 *stft1024_table = nullptr;
};

namespace jdsp {

int fail(jdsp_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess);

#define JDSP_HIP(ctx, call)                                             \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return ::jdsp::fail((ctx), JDSP_EHIP, #call, e_); \
    } while (0)

// stft_kernels.hip
int stft1024_table_count();
void fill_stft1024_table(float2 *host_table);
int launch_stft1024(hipStream_t stream, int n_cu, int fpw_opt, const short *pcm, long n_frames, long hop, float2 *spec,
                    const float2 *table);

}  // namespace jdsp
