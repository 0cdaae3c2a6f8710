// fft_pipe.hip -- what one 512-point wave transform costs when NOTHING else is in the way (GPU box only):
// every wave runs N back-to-back transforms on registers + its private LDS scratch, no global memory in the loop.
//   full      wave_fft512 as the kernels use it
//   valu      the three radix-8 passes and twiddles only (no LDS instructions)
//   lds       the two exchanges only (no arithmetic)
//   stag x2   two transforms per iteration through wave_fft512_x2_staggered
// at 1..8 waves per SIMD.  Printed: cycles per transform per SIMD and per CU-LDS, so that "full" can be compared with
// max(valu, lds) and with valu + lds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ijeicyboodsp_amd/csrc tools/fft_pipe.hip -o build/fft_pipe && build/fft_pipe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include "wave_fft512.h"
using namespace jdsp;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
constexpr int N_ITER = 2048;

template <bool INV> __device__ __forceinline__ void valu_only(float2 (&v)[8], const WaveTwiddles &tw)
{
    fft512_pass1<INV>(v, tw);
    fft512_pass2<INV>(v, tw);
    dft8<INV>(v);
}
__device__ __forceinline__ void lds_only(float2 (&v)[8], float2 *lds, int lane)
{
    fft512_xchg1(v, lds, lane);
    wave_lds_fence();
    fft512_xchg2(v, lds, lane);
    wave_lds_fence();
}

template <int MODE, int MINW>
__global__ __launch_bounds__(64, MINW) void pipe_kernel(const float2 *__restrict__ table, float *out, float seed)
{
    __shared__ __attribute__((aligned(16))) float2 lds[(MODE == 3 ? 2 : 1) * kWaveLdsComplex];
    const int lane = threadIdx.x;
    WaveTwiddles tw;
#pragma unroll
    for (int k = 0; k < 7; k++) { tw.t1[k] = table[k * 64 + lane]; tw.t2[k] = table[448 + k * 8 + (lane & 7)]; }
    float2 a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { a[r] = make_float2(seed * (lane + r), seed - r); b[r] = make_float2(seed + lane, seed * r); }
#pragma unroll 1
    for (int it = 0; it < N_ITER; it++) {
        if (MODE == 0) { wave_fft512<false>(a, lds, lane, tw); wave_lds_fence(); }
        if (MODE == 1) valu_only<false>(a, tw);
        if (MODE == 2) lds_only(a, lds, lane);
        if (MODE == 3) wave_fft512_x2_staggered<false>(a, b, lds, lds + kWaveLdsComplex, lane, tw);
#pragma unroll
        for (int r = 0; r < 8; r++) { a[r].x *= 1e-3f; a[r].y *= 1e-3f; if (MODE == 3) { b[r].x *= 1e-3f; b[r].y *= 1e-3f; } }
    }
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 8; r++) acc += a[r].x + a[r].y + b[r].x + b[r].y;
    if (acc == 1234.5678f) out[lane] = acc;
}

template <int MODE, int MINW>
static float run(int waves_per_simd, int n_cu, const float2 *table, float *d_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = n_cu * 4 * waves_per_simd;
    pipe_kernel<MODE, MINW><<<grid, 64>>>(table, d_out, 1.0f);
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        pipe_kernel<MODE, MINW><<<grid, 64>>>(table, d_out, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[2];
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    std::vector<float2> host(512);
    for (int i = 0; i < 512; i++) host[i] = make_float2(cosf(0.01f * i), sinf(0.01f * i));
    float2 *table; CK(hipMalloc(&table, sizeof(float2) * 512));
    CK(hipMemcpy(table, host.data(), sizeof(float2) * 512, hipMemcpyHostToDevice));
    float *d_out; CK(hipMalloc(&d_out, 4096));
    for (int i = 0; i < 20; i++) run<1, 1>(4, n_cu, table, d_out);
    printf("%d CUs; %d transforms per wave per launch.  ns per transform per SIMD (= launch time / (N x waves per SIMD)); x2 counts two\n", n_cu, N_ITER);
    printf("%-10s %9s %9s %9s %9s\n", "waves/SIMD", "full", "valu", "lds", "stag x2");
    const int ws[] = {1, 2, 3, 4, 6, 8};
    for (int w : ws) {
        const float f = run<0, 1>(w, n_cu, table, d_out), v = run<1, 1>(w, n_cu, table, d_out), l = run<2, 1>(w, n_cu, table, d_out);
        const float s = w <= 4 ? run<3, 1>(w, n_cu, table, d_out) : 0.f;
        const double k = 1e6 / (double)N_ITER / w;
        printf("%-10d %9.1f %9.1f %9.1f %9.1f\n", w, f * k, v * k, l * k, s * k / 2);
    }
    return 0;
}
