// fft_c2c_kernels.hip -- FFTAlgorithm_ver2.cpp on the device: the Bitrev table
// (:186-202, bit-exact) and batched FFTProcess (:94-149) in double precision.
//
// One workgroup owns one n_fft-point transform held entirely in LDS
// (16 B per point: 128 KB at n_fft = 8192).  The structure follows the
// reference: bit-reversed gather through the Bitrev table, then log2(n)
// decimation-in-time stages.  The reference multiplies the upper half of each
// next-size group by its twiddle in a separate pass after the butterflies
// (:128-145); here that multiply is fused into the following stage's butterfly,
// which is the same arithmetic.  Twiddles come from a host-built table of the
// true pi (the reference truncates PI to 3.14159265358, a 1e-11 effect).
#include "jdsp_internal.h"

namespace jdsp {

// FFTAlgorithm_ver2.cpp:191-202 with the reference's 16-bit variables.
__device__ __forceinline__ short bitrev16(int k, int bits, int n_fft)
{
    short walk = (short)k;
    short rev = walk;
    for (int i = 1; i < bits; i++) {
        walk = (short)(walk >> 1);
        rev = (short)(unsigned short)(((unsigned)(unsigned short)rev) << 1);    // <<= 1 on the 16-bit pattern
        rev = (short)(rev | (walk & 1));
    }
    return (short)(rev & (n_fft - 1));
}

__global__ void bitrev_table_kernel(short *table, int n_fft, int bits)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_fft) table[k] = bitrev16(k, bits, n_fft);
}

// tw[j] = exp(-2*pi*i*j/n_fft), j < n_fft/2
__global__ __launch_bounds__(256) void fft_process_f64_kernel(const double2 *__restrict__ in, double2 *__restrict__ out,
                                                             int n_fft, int log2n, int forward,
                                                             const double2 *__restrict__ tw)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double2 *x = reinterpret_cast<double2 *>(smem_raw);
    const double2 *src = in + (size_t)blockIdx.x * n_fft;
    double2 *dst = out + (size_t)blockIdx.x * n_fft;
    // Bitrev gather (:204-205), done as a coalesced read + scattered LDS write: the table is an
    // involution for n_fft = 2^log2n, so x[rev[k]] = src[k] is the same permutation
    for (int k = threadIdx.x; k < n_fft; k += blockDim.x)
        x[(unsigned short)bitrev16(k, log2n, n_fft)] = src[k];
    __syncthreads();
    const int half_n = n_fft >> 1;
    for (int s = 0; s < log2n; s++) {
        const int half = 1 << s;
        const int tstep = half_n >> s;                              // twiddle stride: n_fft / (2*half)
        for (int b = threadIdx.x; b < half_n; b += blockDim.x) {
            const int j = b & (half - 1);
            const int i0 = ((b >> s) << (s + 1)) + j;
            double2 w = tw[j * tstep];
            if (!forward) w.y = -w.y;
            const double2 u = x[i0], v = x[i0 + half];
            const double tr = w.x * v.x - w.y * v.y;
            const double ti = w.x * v.y + w.y * v.x;
            x[i0] = make_double2(u.x + tr, u.y + ti);
            x[i0 + half] = make_double2(u.x - tr, u.y - ti);
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < n_fft; k += blockDim.x) dst[k] = x[k];
}

int launch_bitrev_table(hipStream_t stream, short *table_dev, int n_fft, int bits)
{
    hipLaunchKernelGGL(bitrev_table_kernel, dim3((n_fft + 255) / 256), dim3(256), 0, stream, table_dev, n_fft, bits);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_fft_process_f64(hipStream_t stream, const double2 *in, double2 *out, int n_fft, int log2n, long batch,
                           int forward, const double2 *tw)
{
    if (batch <= 0) return 0;
    const size_t lds = sizeof(double2) * (size_t)n_fft;
    int threads = n_fft / 2 < 256 ? (n_fft / 2 < 64 ? 64 : n_fft / 2) : 256;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)fft_process_f64_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -1;
    }
    hipLaunchKernelGGL(fft_process_f64_kernel, dim3((unsigned)batch), dim3(threads), lds, stream, in, out, n_fft, log2n,
                       forward, tw);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

void fill_c2c_twiddles(double2 *t, int n_fft)
{
    const double two_pi = 6.283185307179586476925286766559;
    for (int j = 0; j < n_fft / 2; j++) {
        double a = -two_pi * (double)j / (double)n_fft;
        t[j] = make_double2(cos(a), sin(a));
    }
}

}  // namespace jdsp
